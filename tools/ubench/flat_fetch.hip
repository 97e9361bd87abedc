// How many 128-byte line requests does the flat one-point-per-lane ensemble sweep send to the L2?
// 52 streams (51 members + target) of NPLANE floats each, read as 64-element pieces by every wave, non-temporal dword loads.
//   rows<T>   : one block of T threads per row of 1440 floats (the longitude-fastest geometry)
//   flat<T,MASKED> : one block of T threads per chunk of CH rows of 721 floats, lanes started on a 64-element boundary of
//               the plane; the lanes in front of the chunk either run one trip ahead (MASKED = false: what s1_xf1_kernel
//               did until round 3) or are masked in the first trip
// Measured (MI355X, line requests per launch / expected): rows<64> 1.003; flat<64,false> 1.49, flat<256,false> 1.12;
// a buffer shifted by 64 bytes: rows<64> 1.51.
// run under `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace`; expected = bytes / 128 per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int M = 52;
constexpr long NPLANE = 721L * 1440L;

template <int T>
__global__ void __launch_bounds__(T) rows_kernel(const float* __restrict__ p, long mstride, float* out) {
  const long base = (long)blockIdx.x * 1440;
  float s = 0.f;
  for (int x = threadIdx.x; x < 1440; x += T) {
    float v[M];
#pragma unroll
    for (int m = 0; m < M; ++m) v[m] = __builtin_nontemporal_load(p + base + x + m * mstride);
#pragma unroll
    for (int m = 0; m < M; ++m) s += v[m];
  }
  if (s == 12345.f) out[0] = s;
}

template <int T, bool MASKED>
__global__ void __launch_bounds__(T) flat_kernel(const float* __restrict__ p, long mstride, int chunk_rows, float* out) {
  const long e0 = (long)blockIdx.x * chunk_rows * 721;
  long e1 = e0 + (long)chunk_rows * 721;
  if (e1 > NPLANE) e1 = NPLANE;
  float s = 0.f;
  long e = (e0 & ~63L) + threadIdx.x;
  if (!MASKED && e < e0) e += T;  // "sit the first trip out" by running one trip AHEAD: the wave's lanes split over two pieces
  for (; e < e1; e += T) {
    if (!MASKED || e >= e0) {     // the lanes in front of e0 masked in the first trip: one piece per wave load
      float v[M];
#pragma unroll
      for (int m = 0; m < M; ++m) v[m] = __builtin_nontemporal_load(p + e + m * mstride);
#pragma unroll
      for (int m = 0; m < M; ++m) s += v[m];
    }
  }
  if (s == 12345.f) out[0] = s;
}

int main(int argc, char** argv) {
  const long shift = argc > 1 ? atol(argv[1]) : 0;  // floats the whole buffer is shifted by (16 = 64 bytes)
  const int nlev = 8;
  float *buf, *out;
  const size_t n = (size_t)nlev * M * NPLANE + 1024;
  hipMalloc(&buf, n * 4);
  hipMalloc(&out, 4);
  hipMemset(buf, 0, n * 4);
  const float* p = buf + shift;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  auto time = [&](const char* name, auto launch) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-28s shift %3ld floats: %.4f ms per launch, (8 kernel launches, one per level) expected 128-byte line requests per kernel launch %.0f\n", name, shift, ms / 5,
           (double)M * NPLANE * 4 / 128);
  };
  const int ch = 13, nchunk = (1440 + ch - 1) / ch;
  // one launch per level keeps the geometry of one plane; levels are back to back in the buffer
  time("rows_kernel<64>", [&] { for (int l = 0; l < nlev; ++l) hipLaunchKernelGGL(rows_kernel<64>, dim3(721), dim3(64), 0, 0, p + (size_t)l * M * NPLANE, NPLANE, out); });
  time("flat_kernel<64,ahead>", [&] { for (int l = 0; l < nlev; ++l) hipLaunchKernelGGL((flat_kernel<64, false>), dim3(nchunk), dim3(64), 0, 0, p + (size_t)l * M * NPLANE, NPLANE, ch, out); });
  time("flat_kernel<256,ahead>", [&] { for (int l = 0; l < nlev; ++l) hipLaunchKernelGGL((flat_kernel<256, false>), dim3(nchunk), dim3(256), 0, 0, p + (size_t)l * M * NPLANE, NPLANE, ch, out); });
  time("flat_kernel<64,masked>", [&] { for (int l = 0; l < nlev; ++l) hipLaunchKernelGGL((flat_kernel<64, true>), dim3(nchunk), dim3(64), 0, 0, p + (size_t)l * M * NPLANE, NPLANE, ch, out); });
  time("flat_kernel<256,masked>", [&] { for (int l = 0; l < nlev; ++l) hipLaunchKernelGGL((flat_kernel<256, true>), dim3(nchunk), dim3(256), 0, 0, p + (size_t)l * M * NPLANE, NPLANE, ch, out); });
  return 0;
}
