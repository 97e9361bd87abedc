// Is the binned (atom) kernel's skeleton bound by vector-memory INSTRUCTIONS?  One wave walks a patch of 64 * V adjacent
// columns down the rows of p, t, c (fp32) + a one-byte atom id per point, V = 1 (dword + ubyte loads: what det_atoms_kernel
// does) or V = 2 (dwordx2 + ushort loads: two adjacent columns per lane), with the kernel's arithmetic per point (DET6
// terms, two accumulator sets selected by atom id: 12 fp64 FMAs).  PD rows are in flight.
// grid: rows x 1440 columns per plane, nplane planes; patch = (plane, column tile, row range of RR rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
// -DCW_RAGGED: the latitude-fastest chunk instead (rows of 721 floats: not a whole number of 128-byte lines, so a wave load
// touches three lines and neighbouring tiles share their boundary lines): 1440 rows of 721 points, 9 row ranges per plane,
// with / without the non-temporal hint and with / without the XCD-contiguous block order of patch_decode (wbx_patch.hpp).
#ifdef CW_RAGGED
constexpr int NX = 721, NROW = 1440, RR = 160, NRANGE = 9;
#else
constexpr int NX = 1440, NROW = 721, RR = 103, NRANGE = 7;  // 7 row ranges per plane
#endif
template <bool NT, typename Q>
__device__ __forceinline__ Q ld(const Q* q) {
  if constexpr (NT) return __builtin_nontemporal_load(q);
  else return *q;
}

template <int V>
struct Vec;
template <>
struct Vec<1> { using F = float; using B = uint8_t; };
typedef float float2v __attribute__((ext_vector_type(2)));
template <>
struct Vec<2> { using F = float2v; using B = uint16_t; };

template <int V, int PD, int WAVES, bool NT = true, bool XCD = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, WAVES))) walk_kernel(const float* __restrict__ p, const float* __restrict__ t, const float* __restrict__ c,
                                                  const uint8_t* __restrict__ aid, int ntile, int nplane, double* out) {
  using F = typename Vec<V>::F;
  using B = typename Vec<V>::B;
  const int lane = threadIdx.x;
  int b = blockIdx.x;
  if constexpr (XCD) {  // blocks that follow each other in b run on ONE XCD (blockIdx.x round-robins over the eight)
    const int per_xcd = (gridDim.x + 7) >> 3;
    b = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (b >= (int)gridDim.x) return;
  }
  const int tile = b % ntile; b /= ntile;
  const int rr = b % NRANGE; const int plane = b / NRANGE;
  if (plane >= nplane) return;  // (the grid is rounded up to a multiple of 8)
  const int x = tile * 64 * V + lane * V;
  if (x >= NX) return;
  const int r0 = rr * RR, r1 = r0 + RR < NROW ? r0 + RR : NROW;
  const size_t base = (size_t)plane * NROW * NX;
  double acc[2][6] = {};
  const int key0 = 3, key1 = 7;
  F fp[PD], ft[PD], fc[PD]; B fa[PD];
  auto load = [&](int r, int s) {
    const size_t o = base + (size_t)(r < r1 ? r : r1 - 1) * NX + x;
    fp[s] = ld<NT>(reinterpret_cast<const F*>(p + o));
    ft[s] = ld<NT>(reinterpret_cast<const F*>(t + o));
    fc[s] = ld<NT>(reinterpret_cast<const F*>(c + o));  // a climatology plane per (lead, level): 12 B/point from HBM
    fa[s] = *reinterpret_cast<const B*>(aid + (o % ((size_t)NROW * NX)));
  };
#pragma unroll
  for (int s = 0; s < PD; ++s) load(r0 + s, s);
  for (int r = r0; r < r1; r += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      const F vp = fp[s], vt = ft[s], vc = fc[s]; const B va = fa[s];
      load(r + PD + s, s);
      if (r + s < r1) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const double dp = (double)reinterpret_cast<const float*>(&vp)[v], dt = (double)reinterpret_cast<const float*>(&vt)[v],
                       dc = (double)reinterpret_cast<const float*>(&vc)[v];
          const int a = (va >> (8 * v)) & 255;
          const double e = dp - dt, pa = dp - dc, ta = dt - dc;
          const double w0 = a == key0 ? 1.0 : 0.0, w1 = a == key1 ? 1.0 : 0.0;
          const double s6[6] = {e, e * e, fabs(e), pa * ta, pa * pa, ta * ta};
#pragma unroll
          for (int l = 0; l < 6; ++l) { acc[0][l] = fma(s6[l], w0, acc[0][l]); acc[1][l] = fma(s6[l], w1, acc[1][l]); }
        }
      }
    }
  }
  double s = 0;
  for (int l = 0; l < 6; ++l) s += acc[0][l] + acc[1][l];
  if (s == 1234.5) out[0] = s;
}

template <int V, int PD, int WAVES = 8, bool NT = true, bool XCD = false>
void run(const char* name, const float* p, const float* t, const float* c, const uint8_t* aid, int nplane, double* out) {
  const int ntile = (NX + 64 * V - 1) / (64 * V);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto launch = [&] { hipLaunchKernelGGL((walk_kernel<V, PD, WAVES, NT, XCD>), dim3((nplane * NRANGE * ntile + 7) / 8 * 8), dim3(64), 0, 0, p, t, c, aid, ntile, nplane, out); };
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a); for (int i = 0; i < 10; ++i) launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 10;
  const double bytes = (double)nplane * NROW * NX * 12;
  printf("%-34s %.4f ms  %.0f GB/s (12 B/point)  %.1f %% of 8 TB/s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80);
}

int main() {
  const int nplane = 156;  // 12 leads x 13 levels
  const size_t n = (size_t)nplane * NROW * NX;
  float *p, *t, *c; uint8_t* aid; double* out;
  (void)hipMalloc(&p, n * 4 + 64); (void)hipMalloc(&t, n * 4 + 64); (void)hipMalloc(&c, n * 4 + 64);
  (void)hipMalloc(&aid, (size_t)NROW * NX + 64); (void)hipMalloc(&out, 8);
  (void)hipMemset(p, 0, n * 4); (void)hipMemset(t, 0, n * 4); (void)hipMemset(c, 0, n * 4); (void)hipMemset(aid, 3, (size_t)NROW * NX);
#ifdef CW_RAGGED
  run<1, 4, 4, true, false>("721: nt loads, launch order", p, t, c, aid, nplane, out);
  run<1, 4, 4, false, false>("721: plain loads, launch order", p, t, c, aid, nplane, out);
  run<1, 4, 4, true, true>("721: nt loads, XCD-contiguous", p, t, c, aid, nplane, out);
  run<1, 4, 4, false, true>("721: plain loads, XCD-contiguous", p, t, c, aid, nplane, out);
  run<1, 8, 4, false, true>("721: plain, XCD, 8 rows ahead", p, t, c, aid, nplane, out);
  run<1, 4, 6, false, true>("721: plain, XCD, <= 6 waves/SIMD", p, t, c, aid, nplane, out);
  run<2, 4, 4, false, true>("721: plain, XCD, 2 columns/lane", p, t, c, aid, nplane, out);
  run<2, 4, 4, true, true>("721: nt, XCD, 2 columns/lane", p, t, c, aid, nplane, out);
  return 0;
#endif
  run<1, 4>("1 column per lane, 4 rows ahead", p, t, c, aid, nplane, out);
  run<1, 8>("1 column per lane, 8 rows ahead", p, t, c, aid, nplane, out);
  run<2, 2>("2 columns per lane, 2 rows ahead", p, t, c, aid, nplane, out);
  run<2, 4>("2 columns per lane, 4 rows ahead", p, t, c, aid, nplane, out);
  run<1, 4, 6>("1 column, 4 rows, <= 6 waves/SIMD", p, t, c, aid, nplane, out);
  run<1, 4, 5>("1 column, 4 rows, <= 5 waves/SIMD", p, t, c, aid, nplane, out);
  run<1, 4, 4>("1 column, 4 rows, <= 4 waves/SIMD", p, t, c, aid, nplane, out);
  run<1, 4, 3>("1 column, 4 rows, <= 3 waves/SIMD", p, t, c, aid, nplane, out);
  run<1, 8, 4>("1 column, 8 rows, <= 4 waves/SIMD", p, t, c, aid, nplane, out);
  run<2, 4, 4>("2 columns, 4 rows, <= 4 waves/SIMD", p, t, c, aid, nplane, out);
  return 0;
}
