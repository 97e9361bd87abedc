// Stage-1 skeleton: fused per-point statistic + unweighted fp64 partial sums.
//
// Index space (built by the host planner, weatherbenchx_amd/planner.py):
//   key   k in [0,nkey)   one partial per key            -> blockIdx
//   depth d in [0,D)      summed here, cut in nchunk chunks -> blockIdx
//   x     in [0,nx)       innermost, lane-mapped dimension
// Two kernels share every Op (statistic family):
//   s1_xr : x is summed away.   One wave sweeps one row at a time with V-wide loads,
//           the waves of a block interleave over the chunk's rows, fp64 lane partials
//           are folded with wave shuffles + LDS, one write per (key,chunk,lane).
//   s1_xk : x is kept (latitude-fastest real data, or x is a surviving dim).  One lane
//           owns V x-positions and walks the chunk's rows; no cross-lane traffic at all.
// Both are pure streaming reads: algorithmic bytes = the inputs, once.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "wbx_common.hpp"

namespace wbx {

struct S1Args {
  const void* in[WBX_MAX_INPUTS];
  const int64_t* key_off[WBX_MAX_INPUTS];
  const int64_t* depth_off[WBX_MAX_INPUTS];
  int64_t xstride[WBX_MAX_INPUTS];
  const int32_t* gk;
  const int32_t* gd;
  const int64_t* gtab;
  int32_t ngd;
  int64_t nkey, D, nx, dchunk;
  int32_t nchunk, nxtile;
  uint32_t flags;
  double* out;
  const double* xw;  // per-x weights folded into stage 1 (plan->x_weights) or NULL
  // ensemble
  int32_t M;
  int64_t mstride;
  // map kernels (wbx_ens2_partial: the target ensemble size)
  int32_t lane;
  int64_t ngd_t;  // wbx_ens2_partial: element stride between target members
};

// Row base offsets (elements) of every input for (key, depth row).
template <int NIN>
__device__ __forceinline__ void key_bases(const S1Args& a, int64_t key, int64_t (&kb)[WBX_MAX_INPUTS]) {
#pragma unroll
  for (int i = 0; i < WBX_MAX_INPUTS; ++i) kb[i] = (i < NIN || i == 3) && a.key_off[i] ? a.key_off[i][key] : 0;
}

template <int NIN>
__device__ __forceinline__ void row_bases(const S1Args& a, const int64_t (&kb)[WBX_MAX_INPUTS], int64_t key,
                                          int64_t d, int64_t (&ro)[WBX_MAX_INPUTS]) {
#pragma unroll
  for (int i = 0; i < WBX_MAX_INPUTS; ++i)
    ro[i] = kb[i] + (((i < NIN || i == 3) && a.depth_off[i]) ? a.depth_off[i][d] : 0);
  if (NIN > 2 && a.gtab) ro[2] += a.gtab[(int64_t)(a.gk ? a.gk[key] : 0) * a.ngd + (a.gd ? a.gd[d] : 0)];
}

// ---------------------------------------------------------------------------------------------
// x summed away.  grid = nkey * nchunk blocks, block = 64..256 threads.
// ---------------------------------------------------------------------------------------------
// Weights that depend on the innermost dim only (area weights on latitude-fastest data) can be applied INSIDE stage 1
// (plan->x_weights), so that x is summed here like any other reduced dim; s1_xf_kernel below does that for contiguous
// planes.  (A generic per-row variant -- every op wrapped so that its lanes are multiplied by w[x] on the x-summed
// kernel with dword loads -- measured slower than keeping x: 52 % vs 76 % for DET6 on 721-float rows, 0.50 vs 0.46 ms
// for the 51-member ensemble, 43 % vs 69 % with a mask; it was removed.)
constexpr int WBX_XW_MAX = 2048;
static __shared__ double wbx_xw_lds[WBX_XW_MAX];

// MROW: the validity mask does not depend on the depth dims (a (lat, lon) mask under an (init) reduction): the key's
// mask row is staged in LDS once per block instead of being re-read from L2 for every depth row.
constexpr int WBX_MROW_MAX = 8192;

// Ops that provide accum_mrow() declare `static constexpr bool MROW_OK = true`.
template <class Op, class = void>
struct op_has_mrow : std::false_type {};
template <class Op>
struct op_has_mrow<Op, std::void_t<decltype(Op::MROW_OK)>> : std::integral_constant<bool, Op::MROW_OK> {};

template <class Op, int V, bool MROW = false>
__global__ void __launch_bounds__(256, Op::MIN_WAVES) s1_xr_kernel(S1Args a) {
  constexpr int NA = Op::NACC;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nwave = blockDim.x >> 6;
  const int64_t b = blockIdx.x;
  const int64_t key = b / a.nchunk;
  const int chunk = (int)(b - key * a.nchunk);
  const int64_t d0 = (int64_t)chunk * a.dchunk;
  const int64_t d1 = d0 + a.dchunk < a.D ? d0 + a.dchunk : a.D;

  int64_t kb[WBX_MAX_INPUTS];
  key_bases<Op::NIN>(a, key, kb);
  double acc[1][NA];
#pragma unroll
  for (int l = 0; l < NA; ++l) acc[0][l] = 0.0;

  __shared__ __attribute__((aligned(16))) uint8_t smask[MROW ? WBX_MROW_MAX : 16];
  if constexpr (MROW) {
    const uint8_t* src = reinterpret_cast<const uint8_t*>(a.in[3]) + kb[3];
    for (int64_t i = threadIdx.x; i < a.nx; i += blockDim.x) smask[i] = src[i];
    __syncthreads();
  }

  // Row offsets come from tables (depth offsets per input, the climatology gather: two dependent lookups).  Resolving
  // them row by row puts that latency in front of every row's loads (a row is only ~6 loads per lane); instead lane l
  // resolves the wave's l-th row up front and the sweep broadcasts the results (configs[1]: 4.05 -> 3.9 ms).
  for (int64_t dbatch = d0 + wave; dbatch < d1; dbatch += (int64_t)64 * nwave) {
    const int64_t dmine = dbatch + (int64_t)lane * nwave;
    int64_t rov[WBX_MAX_INPUTS];
    row_bases<Op::NIN>(a, kb, key, dmine < d1 ? dmine : d1 - 1, rov);
    const int64_t left = (d1 - dbatch + nwave - 1) / nwave;
    const int nrow = (int)(left < 64 ? left : 64);
    for (int l = 0; l < nrow; ++l) {
      int64_t ro[WBX_MAX_INPUTS];
#pragma unroll
      for (int i = 0; i < WBX_MAX_INPUTS; ++i) ro[i] = (i < Op::NIN || i == 3) ? readlane64(rov[i], l) : 0;
#pragma unroll Op::XR_UNROLL
      for (int64_t x = (int64_t)lane * V; x < a.nx; x += 64 * V) {
        if constexpr (MROW)
          Op::template accum_mrow<V>(a, ro, x, acc, smask);
        else
          Op::template accum<V, false>(a, ro, x, acc);
      }
    }
  }

  __shared__ double red[4][NA];
#pragma unroll
  for (int l = 0; l < NA; ++l) {
    double v = wave_sum(acc[0][l]);
    if (lane == 0) red[wave][l] = v;
  }
  __syncthreads();
  if (threadIdx.x < NA) {
    double s = 0.0;
    for (int w = 0; w < nwave; ++w) s += red[w][threadIdx.x];
    a.out[(key * a.nchunk + chunk) * NA + threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// x kept.  grid = nkey * nxtile * nchunk blocks; a block covers blockDim*V consecutive x.
template <class Op, int V>
__global__ void __launch_bounds__(256, Op::MIN_WAVES) s1_xk_kernel(S1Args a) {
  constexpr int NA = Op::NACC;
  int64_t b = blockIdx.x;
  const int chunk = (int)(b % a.nchunk);
  b /= a.nchunk;
  const int xt = (int)(b % a.nxtile);
  const int64_t key = b / a.nxtile;
  const int64_t x = ((int64_t)xt * blockDim.x + threadIdx.x) * V;
  if (x >= a.nx) return;
  const int64_t d0 = (int64_t)chunk * a.dchunk;
  const int64_t d1 = d0 + a.dchunk < a.D ? d0 + a.dchunk : a.D;

  int64_t kb[WBX_MAX_INPUTS];
  key_bases<Op::NIN>(a, key, kb);
  double acc[V][NA];
#pragma unroll
  for (int k = 0; k < V; ++k)
#pragma unroll
    for (int l = 0; l < NA; ++l) acc[k][l] = 0.0;

  const int nvalid = a.nx - x < V ? (int)(a.nx - x) : V;  // < V only for the last lane of a ragged row
  // (row offsets are resolved row by row here: the 4x unrolled loop already batches the table lookups of four rows;
  // the lane-parallel resolution of s1_xr_kernel defeats that unrolling and measured slower: 0.39 -> 0.44 ms on the
  // latitude-fastest ensemble chunk)
  if (V == 1 || nvalid == V) {
#pragma unroll Op::XK_UNROLL
    for (int64_t d = d0; d < d1; ++d) {
      int64_t ro[WBX_MAX_INPUTS];
      row_bases<Op::NIN>(a, kb, key, d, ro);
      Op::template accum<V, true>(a, ro, x, acc);
    }
  } else {
    for (int64_t d = d0; d < d1; ++d) {
      int64_t ro[WBX_MAX_INPUTS];
      row_bases<Op::NIN>(a, kb, key, d, ro);
      for (int k = 0; k < nvalid; ++k)
        Op::template accum<1, true>(a, ro, x + k, reinterpret_cast<double(&)[1][NA]>(acc[k]));
    }
  }
  double* o = a.out + ((key * a.nchunk + chunk) * NA) * a.nx + x;
#pragma unroll
  for (int l = 0; l < NA; ++l)
#pragma unroll
    for (int k = 0; k < V; ++k)
      if (k < nvalid) o[(int64_t)l * a.nx + k] = acc[k][l];
}

// ---------------------------------------------------------------------------------------------
// x kept, "plane mode": latitude-fastest chunks (real WeatherBench data) have rows of nx = 721 floats,
// so per-row loads can never be 16-B aligned.  But consecutive depth rows (longitudes) are adjacent in memory: a
// group of R rows is one contiguous span of R*nx floats.  The block fetches that span with ALIGNED dwordx4 loads
// (span start rounded down to 16 B, next group prefetched into registers while this one is reduced), parks it in
// LDS, and every lane then reads its own x column from LDS (consecutive lanes -> consecutive banks, conflict free).
// grid = nkey * nchunk, block = round_up(nx, 64) threads, dynamic LDS = NIN * (R*nx + 8) floats.
template <class Op>
__global__ void __launch_bounds__(1024, 6) s1_xp_kernel(S1Args a, int R) {  // <= 80 VGPRs: two 768-thread blocks per CU
  constexpr int NA = Op::NACC;
  constexpr int NIN = Op::NIN;
  extern __shared__ __attribute__((aligned(16))) float lds_raw[];
  const int T = blockDim.x;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t key = b / a.nchunk;
  const int chunk = (int)(b - key * a.nchunk);
  const int64_t d0 = (int64_t)chunk * a.dchunk;
  const int64_t d1 = d0 + a.dchunk < a.D ? d0 + a.dchunk : a.D;
  const int nx = (int)a.nx;
  const int slot = R * nx + 8;  // floats per input in LDS (span + alignment slack)
  const int nvec = (R * nx + 3 + 3) / 4;  // float4s that cover any span of R*nx floats with a lead of <= 3
  constexpr int MAXV = 2;  // float4s per thread per input (host guarantees nvec <= MAXV * T)

  int64_t kb[WBX_MAX_INPUTS];
  key_bases<NIN>(a, key, kb);
  double acc[NA];
#pragma unroll
  for (int l = 0; l < NA; ++l) acc[l] = 0.0;

  float4 regs[NIN][MAXV];
  int lead[NIN], lead_next[NIN];
  // validity mask (Op::MROW_OK ops with WBX_FLAG_MASKED): the span's R * nx mask bytes ride along as dwords
  constexpr bool MASKED = op_has_mrow<Op>::value;
  uint8_t* lds_mask = reinterpret_cast<uint8_t*>(lds_raw + NIN * slot);
  const int nvec_m = (R * nx + 3 + 3) / 4;  // dwords covering any span of R*nx bytes with a lead of <= 3
  uint32_t mregs[MAXV] = {};
  int mlead = 0, mlead_next = 0;

  auto prefetch = [&](int64_t d, int (&ld)[NIN], int& ml) {
    int64_t ro[WBX_MAX_INPUTS];
    row_bases<NIN>(a, kb, key, d, ro);
    const int rows = (int)(d1 - d < R ? d1 - d : R);
    const int n = rows * nx;
    if constexpr (MASKED) {
      const uint8_t* mbase = reinterpret_cast<const uint8_t*>(a.in[3]);
      const int64_t start = ro[3];
      const int64_t a0 = start & ~(int64_t)3;
      ml = (int)(start - a0);
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int k = tid + v * T;
        uint32_t q = 0;
        if (k < nvec_m) {
          const int e0 = 4 * k;
          if (e0 + 3 < ml + n) {
            q = *reinterpret_cast<const uint32_t*>(mbase + a0 + e0);
          } else if (e0 < ml + n) {  // straddles the span end: never read past the last byte
            for (int c = 0; c < 4; ++c)
              if (e0 + c < ml + n) q |= (uint32_t)mbase[a0 + e0 + c] << (8 * c);
          }
        }
        mregs[v] = q;
      }
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const float* base = reinterpret_cast<const float*>(a.in[i]);
      const int64_t start = ro[i];
      const int64_t a0 = start & ~(int64_t)3;
      ld[i] = (int)(start - a0);
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int k = tid + v * T;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < nvec) {
          const int e0 = 4 * k;  // first element of this float4, relative to a0
          if (e0 + 3 < ld[i] + n) {
            typedef float f4_t __attribute__((ext_vector_type(4)));
            const f4_t w = ld_stream(reinterpret_cast<const f4_t*>(base + a0 + e0));  // inside [a0, span end): aligned 16 B
            q = make_float4(w.x, w.y, w.z, w.w);
          } else if (e0 < ld[i] + n) {  // straddles the span end: never read past the last element
            float tmp[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < 4; ++c)
              if (e0 + c < ld[i] + n) tmp[c] = base[a0 + e0 + c];
            q = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
          }
        }
        regs[i][v] = q;
      }
    }
  };

  if (d0 < d1) prefetch(d0, lead, mlead);
  for (int64_t d = d0; d < d1; d += R) {
    // registers -> LDS
#pragma unroll
    for (int i = 0; i < NIN; ++i)
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int k = tid + v * T;
        if (k < nvec) *reinterpret_cast<float4*>(lds_raw + i * slot + 4 * k) = regs[i][v];
      }
    if constexpr (MASKED) {
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int k = tid + v * T;
        if (k < nvec_m) *reinterpret_cast<uint32_t*>(lds_mask + 4 * k) = mregs[v];
      }
    }
    int cur_lead[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) cur_lead[i] = lead[i];
    const int cur_mlead = mlead;
    __syncthreads();
    if (d + R < d1) {  // next group's loads fly while this group is reduced
      prefetch(d + R, lead_next, mlead_next);
#pragma unroll
      for (int i = 0; i < NIN; ++i) lead[i] = lead_next[i];
      mlead = mlead_next;
    }
    if (tid < nx) {
      const int rows = (int)(d1 - d < R ? d1 - d : R);
      for (int r = 0; r < rows; ++r) {
        float pv = lds_raw[0 * slot + cur_lead[0] + r * nx + tid];
        float tv = NIN > 1 ? lds_raw[1 * slot + cur_lead[NIN > 1 ? 1 : 0] + r * nx + tid] : 0.f;
        float cv = NIN > 2 ? lds_raw[2 * slot + cur_lead[NIN > 2 ? 2 : 0] + r * nx + tid] : 0.f;
        if constexpr (MASKED) {
          // masked-out points contribute exactly 0 whatever they hold (aggregation.py:339-357); count lane = validity
          const bool valid = lds_mask[cur_mlead + r * nx + tid] != 0;
          pv = valid ? pv : 0.f;
          tv = valid ? tv : 0.f;
          cv = valid ? cv : 0.f;
          acc[Op::NLANE] += valid ? 1.0 : 0.0;
        }
        double val[Op::NLANE];
        Op::lanes((double)pv, (double)tv, (double)cv, val);
#pragma unroll
        for (int l = 0; l < Op::NLANE; ++l) acc[l] += val[l];
      }
    }
    __syncthreads();
  }
  if (tid < nx) {
    double* o = a.out + ((key * a.nchunk + chunk) * NA) * a.nx + tid;
#pragma unroll
    for (int l = 0; l < NA; ++l) o[(int64_t)l * a.nx] = acc[l];
  }
}

template <class Op>
int launch_plane(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a) {
  const int R = plan->plane_rows;
  const int threads = (int)((plan->nx + 63) / 64) * 64;
  WBX_REQUIRE(plan->x_kept && R > 0 && threads <= 1024, "plane mode needs x kept and nx <= 1024");
  WBX_REQUIRE(plan->depth_chunk % R == 0 || plan->nchunk == 1, "plane mode needs depth_chunk %% plane_rows == 0");
  const int nvec = (int)((R * plan->nx + 6) / 4);
  WBX_REQUIRE(nvec <= 2 * threads, "plane_rows too large for the block (R*nx/4 > 2*threads)");
  const size_t lds = (size_t)Op::NIN * (size_t)(R * plan->nx + 8) * sizeof(float) +
                     (op_has_mrow<Op>::value ? (size_t)(R * plan->nx + 16) : 0);
  WBX_REQUIRE(lds <= 80 * 1024, "plane mode LDS footprint %zu exceeds 80 KiB (two blocks per CU)", lds);
  if (lds > 48 * 1024)
    WBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&s1_xp_kernel<Op>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int64_t grid = plan->nkey * plan->nchunk;
  WBX_REQUIRE(grid < (int64_t)1 << 31, "grid too large (%lld blocks)", (long long)grid);
  hipLaunchKernelGGL((s1_xp_kernel<Op>), dim3((unsigned)grid), dim3(threads), lds, ctx->stream, a, R);
  WBX_HIP(hipGetLastError());
  return 0;
}


// ---------------------------------------------------------------------------------------------
// x summed with folded x-weights, "flat" variant for latitude-fastest planes: R consecutive depth rows (longitudes) of
// nx = 721 floats are one contiguous, 16-B aligned span, so the x-summed sweep does not have to respect row boundaries
// at all: the span is streamed as float4s (aligned non-temporal dwordx4, exactly like the lon-fastest kernel) and the
// element at flat position e simply takes the weight w[e mod nx] from the LDS copy (padded by 3 for the wrap).
// Needs R % 4 == 0 (4 rows = nx float4s exactly, chunks cut at row quads) and no mask.  grid = nkey * nchunk.
template <class Op>
__global__ void __launch_bounds__(256) s1_xf_kernel(S1Args a, int R) {
  constexpr int NA = Op::NACC;
  constexpr int NIN = Op::NIN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nx = (int)a.nx;
  const int64_t b = blockIdx.x;
  const int64_t key = b / a.nchunk;
  const int chunk = (int)(b - key * a.nchunk);
  const int64_t d0 = (int64_t)chunk * a.dchunk;
  const int64_t d1 = d0 + a.dchunk < a.D ? d0 + a.dchunk : a.D;
  const int nt = blockDim.x, nwave = nt >> 6;
  for (int i = tid; i < nx + 3; i += nt) wbx_xw_lds[i] = a.xw[i < nx ? i : i - nx];
  __syncthreads();

  int64_t kb[WBX_MAX_INPUTS];
  key_bases<NIN>(a, key, kb);
  double acc[NA];
#pragma unroll
  for (int l = 0; l < NA; ++l) acc[l] = 0.0;

  typedef float f4_t __attribute__((ext_vector_type(4)));
  const int qpp = R / 4;  // row quads per plane; one quad = nx float4s
  int64_t g = d0 / 4;
  const int64_t g1 = d1 / 4;
  while (g < g1) {
    const int64_t plane = g / qpp;
    const int j0 = (int)(g - plane * qpp);
    const int64_t left = g1 - g;
    const int nj = (int)(left < qpp - j0 ? left : qpp - j0);
    int64_t ro[WBX_MAX_INPUTS];
    row_bases<NIN>(a, kb, key, plane * R, ro);
    const int64_t q0 = (int64_t)j0 * nx, q1 = q0 + (int64_t)nj * nx;  // float4 range of this plane segment
    const f4_t* pp = reinterpret_cast<const f4_t*>(reinterpret_cast<const float*>(a.in[0]) + ro[0]);
    const f4_t* pt = NIN > 1 ? reinterpret_cast<const f4_t*>(reinterpret_cast<const float*>(a.in[1]) + ro[1]) : nullptr;
    const f4_t* pc = NIN > 2 ? reinterpret_cast<const f4_t*>(reinterpret_cast<const float*>(a.in[2]) + ro[2]) : nullptr;
    const uint32_t* pm = nullptr;  // validity mask: the same flat walk over its (contiguous) plane, 4 bytes per float4
    if constexpr (op_has_mrow<Op>::value) pm = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(a.in[3]) + ro[3]);
    int m = (int)((4 * (q0 + tid)) % nx);  // latitude of the thread's first element
    const int step = (4 * nt) % nx;
#pragma unroll 2
    for (int64_t q = q0 + tid; q < q1; q += nt) {
      const f4_t p4 = ld_stream(pp + q);
      f4_t t4 = {0.f, 0.f, 0.f, 0.f}, c4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (NIN > 1) t4 = ld_stream(pt + q);
      if constexpr (NIN > 2) c4 = ld_stream(pc + q);
      uint32_t mq = 0xffffffffu;
      if constexpr (op_has_mrow<Op>::value) mq = pm[q];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pv = p4[k], tv = t4[k], cv = c4[k];
        const double w = wbx_xw_lds[m + k];
        if constexpr (op_has_mrow<Op>::value) {
          // masked-out points contribute exactly 0 whatever they hold (aggregation.py:339-357); count lane = sum of w
          const bool valid = ((mq >> (8 * k)) & 0xffu) != 0;
          pv = valid ? pv : 0.f;
          tv = valid ? tv : 0.f;
          cv = valid ? cv : 0.f;
          acc[Op::NLANE] += valid ? w : 0.0;
        }
        double val[Op::NLANE];
        Op::lanes((double)pv, (double)tv, (double)cv, val);
#pragma unroll
        for (int l = 0; l < Op::NLANE; ++l) acc[l] = fma(val[l], w, acc[l]);
      }
      m += step;
      m = m >= nx ? m - nx : m;
    }
    g += nj;
  }
  __shared__ double red[4][NA];
#pragma unroll
  for (int l = 0; l < NA; ++l) {
    const double v = wave_sum(acc[l]);
    if (lane == 0) red[wave][l] = v;
  }
  __syncthreads();
  if (tid < NA) {
    double sum = 0.0;
    for (int w = 0; w < nwave; ++w) sum += red[w][tid];
    a.out[(key * a.nchunk + chunk) * NA + tid] = sum;
  }
}

template <class Op>
int launch_flat_weighted(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a) {
  const int R = plan->plane_rows;
  WBX_REQUIRE(!plan->x_kept && plan->x_weights && R > 0 && R % 4 == 0 && plan->nx + 3 <= WBX_XW_MAX,
              "flat x-weighted mode needs x summed, plane_rows %% 4 == 0 and nx <= %d", WBX_XW_MAX - 3);
  WBX_REQUIRE(plan->depth_chunk % 4 == 0 && plan->ndepth % R == 0, "flat x-weighted mode needs depth_chunk %% 4 == 0 and whole planes");
  const int64_t grid = plan->nkey * plan->nchunk;
  WBX_REQUIRE(grid < (int64_t)1 << 31, "grid too large (%lld blocks)", (long long)grid);
  hipLaunchKernelGGL((s1_xf_kernel<Op>), dim3((unsigned)grid), dim3(plan->block_threads), 0, ctx->stream, a, R);
  WBX_HIP(hipGetLastError());
  return 0;
}


// The same flat sweep for ops that work one point at a time (the ensemble family: a point's M members sit at
// ro[0] + e + m * mstride, so walking e over the contiguous plane needs no row structure either).  Any nx, any row
// count; chunks are cut at rows.  grid = nkey * nchunk, block = plan->block_threads.
template <class Op>
__global__ void __launch_bounds__(256, Op::MIN_WAVES) s1_xf1_kernel(S1Args a, int R) {
  constexpr int NA = Op::NACC;  // value lanes + the count lanes of the masked / skipna wrappers: all take the weight
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x, nwave = nt >> 6;
  const int nx = (int)a.nx;
  const int64_t b = blockIdx.x;
  const int64_t key = b / a.nchunk;
  const int chunk = (int)(b - key * a.nchunk);
  const int64_t d0 = (int64_t)chunk * a.dchunk;
  const int64_t d1 = d0 + a.dchunk < a.D ? d0 + a.dchunk : a.D;
  for (int i = tid; i < nx; i += nt) wbx_xw_lds[i] = a.xw[i];
  __syncthreads();
  int64_t kb[WBX_MAX_INPUTS];
  key_bases<Op::NIN>(a, key, kb);
  double acc[NA];
#pragma unroll
  for (int l = 0; l < NA; ++l) acc[l] = 0.0;
  const int step = nt % nx;
  int64_t d = d0;
  while (d < d1) {
    const int64_t plane = d / R;
    const int64_t j0 = d - plane * R;
    const int64_t nj = d1 - d < R - j0 ? d1 - d : R - j0;
    int64_t ro[WBX_MAX_INPUTS];
    row_bases<Op::NIN>(a, kb, key, plane * R, ro);
    const int64_t e0 = j0 * nx, e1 = (j0 + nj) * nx;
    // lanes start on a 64-element boundary of the plane (planes are normally 128-B aligned, rows of 721 are not), so
    // every wave load covers whole cache lines.  The lanes in front of e0 are DROPPED in the first trip -- they must not run
    // one trip ahead of the others instead: a wave whose lanes sit in two different 256-byte pieces touches three or four
    // lines per load, every trip (round 2 did that: FETCH_SIZE 1.11 x the algorithmic bytes with 256-thread blocks, whose
    // first wave is the split one, 1.44 x with one-wave blocks; tools/ubench/flat_fetch.hip counts the line requests).
    int64_t e = (e0 & ~(int64_t)63) + tid;
    int m = (int)((e - e0) % nx);  // the weight index of e (a lane in front of e0 counts back from the row's end)
    m = m < 0 ? m + nx : m;
    for (; e < e1; e += nt) {
      // (branch-free: the masked lanes read their own element -- it lies in the same plane, in the previous chunk -- and
      // drop the values; a divergent `if` around the body costs 200 instead of 117 VGPRs)
      double one[1][NA];
#pragma unroll
      for (int l = 0; l < NA; ++l) one[0][l] = 0.0;
      Op::template accum<1, false>(a, ro, e, one);
      const double w = wbx_xw_lds[m];
      const bool mine = e >= e0;
#pragma unroll
      for (int l = 0; l < NA; ++l) acc[l] = fma(mine ? one[0][l] : 0.0, w, acc[l]);
      m += step;
      m = m >= nx ? m - nx : m;
    }
    d += nj;
  }
  __shared__ double red[4][NA];
#pragma unroll
  for (int l = 0; l < NA; ++l) {
    const double v = wave_sum(acc[l]);
    if (lane == 0) red[wave][l] = v;
  }
  __syncthreads();
  if (tid < NA) {
    double sum = 0.0;
    for (int w = 0; w < nwave; ++w) sum += red[w][tid];
    a.out[(key * a.nchunk + chunk) * NA + tid] = sum;
  }
}

template <class Op>
int launch_flat_weighted1(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a) {
  const int R = plan->plane_rows;
  WBX_REQUIRE(!plan->x_kept && plan->x_weights && R > 0 && plan->nx <= WBX_XW_MAX && plan->ndepth % R == 0,
              "flat x-weighted mode needs x summed, whole planes and nx <= %d", WBX_XW_MAX);
  WBX_REQUIRE(plan->xstride[0] == 1 && plan->xstride[1] == 1, "flat x-weighted mode needs unit x stride");
  if (plan->flags & WBX_FLAG_MASKED) WBX_REQUIRE(plan->xstride[3] == 1, "flat x-weighted mode needs a mask stored like the data");
  const int64_t grid = plan->nkey * plan->nchunk;
  WBX_REQUIRE(grid < (int64_t)1 << 31, "grid too large (%lld blocks)", (long long)grid);
  hipLaunchKernelGGL((s1_xf1_kernel<Op>), dim3((unsigned)grid), dim3(plan->block_threads), 0, ctx->stream, a, R);
  WBX_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// materialise one lane: out[key][d][x].  grid = nkey * D * nxtile.
template <class Op>
__global__ void __launch_bounds__(256) s1_map_kernel(S1Args a) {
  int64_t b = blockIdx.x;
  const int xt = (int)(b % a.nxtile);
  b /= a.nxtile;
  const int64_t d = b % a.D;
  const int64_t key = b / a.D;
  const int64_t x = (int64_t)xt * blockDim.x + threadIdx.x;
  if (x >= a.nx) return;
  int64_t kb[WBX_MAX_INPUTS], ro[WBX_MAX_INPUTS];
  key_bases<Op::NIN>(a, key, kb);
  row_bases<Op::NIN>(a, kb, key, d, ro);
  double val[Op::NLANE];
  Op::values(a, ro, x, val);
  double r = 0.0;
#pragma unroll
  for (int l = 0; l < Op::NLANE; ++l)
    if (l == a.lane) r = val[l];
  a.out[(key * a.D + d) * a.nx + x] = r;
}

inline int fill_args(const wbx_s1_plan* plan, S1Args& a) {
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < WBX_MAX_INPUTS; ++i) {
    a.key_off[i] = plan->key_off[i];
    a.depth_off[i] = plan->depth_off[i];
    a.xstride[i] = plan->xstride[i];
  }
  a.gk = plan->gather_key;
  a.gd = plan->gather_depth;
  a.gtab = plan->gather_tab;
  a.ngd = plan->n_gather_depth > 0 ? plan->n_gather_depth : 1;
  a.nkey = plan->nkey;
  a.D = plan->ndepth;
  a.nx = plan->nx;
  a.dchunk = plan->depth_chunk;
  a.nchunk = plan->nchunk;
  a.flags = plan->flags;
  a.xw = plan->x_weights;
  return 0;
}

inline int check_plan(const wbx_s1_plan* p) {
  WBX_REQUIRE(p != nullptr, "plan is NULL");
  WBX_REQUIRE(p->nkey >= 0 && p->ndepth >= 0 && p->nx >= 0, "negative plan extent");
  WBX_REQUIRE(p->nchunk >= 1 && p->depth_chunk >= 1, "nchunk/depth_chunk must be >= 1");
  WBX_REQUIRE((int64_t)p->nchunk * p->depth_chunk >= p->ndepth, "chunks do not cover depth");
  WBX_REQUIRE(p->block_threads == 64 || p->block_threads == 128 || p->block_threads == 256,
              "block_threads must be 64, 128 or 256 (got %d)", p->block_threads);
  WBX_REQUIRE(p->vec == 1 || p->vec == 4, "vec must be 1 or 4 (got %d)", p->vec);
  if (p->vec == 4) {
    WBX_REQUIRE(p->x_kept || p->nx % 4 == 0, "vec=4 with x summed needs nx %% 4 == 0");
    for (int i = 0; i < WBX_MAX_INPUTS; ++i)
      WBX_REQUIRE(p->xstride[i] == 0 || p->xstride[i] == 1, "vec=4 needs unit/zero x strides");
  }
  return 0;
}

// Launch helpers -----------------------------------------------------------------------------
template <class Op, int V>
int launch_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, bool mask_row_in_lds = false) {
  if (plan->nkey == 0) return 0;
  const int64_t nj = plan->x_kept ? plan->nx : 1;
  if (plan->ndepth == 0 || plan->nx == 0) {
    // empty reduction: sums are zero
    size_t n = (size_t)plan->nkey * plan->nchunk * Op::NACC * (size_t)nj;
    if (n) WBX_HIP(hipMemsetAsync(a.out, 0, n * sizeof(double), ctx->stream));
    return 0;
  }
  if (plan->x_kept) {
    const int64_t per_block = (int64_t)plan->block_threads * V;
    a.nxtile = (int)((plan->nx + per_block - 1) / per_block);
    const int64_t grid = plan->nkey * a.nxtile * plan->nchunk;
    WBX_REQUIRE(grid < (int64_t)1 << 31, "grid too large (%lld blocks)", (long long)grid);
    hipLaunchKernelGGL((s1_xk_kernel<Op, V>), dim3((unsigned)grid), dim3(plan->block_threads), 0, ctx->stream, a);
  } else {
    const int64_t grid = plan->nkey * plan->nchunk;
    WBX_REQUIRE(grid < (int64_t)1 << 31, "grid too large (%lld blocks)", (long long)grid);
    if constexpr (op_has_mrow<Op>::value) {
      if (mask_row_in_lds) {
        hipLaunchKernelGGL((s1_xr_kernel<Op, V, true>), dim3((unsigned)grid), dim3(plan->block_threads), 0, ctx->stream, a);
        WBX_HIP(hipGetLastError());
        return 0;
      }
    }
    hipLaunchKernelGGL((s1_xr_kernel<Op, V>), dim3((unsigned)grid), dim3(plan->block_threads), 0, ctx->stream, a);
  }
  WBX_HIP(hipGetLastError());
  return 0;
}

template <class Op>
int launch_map(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a) {
  if (plan->nkey == 0 || plan->ndepth == 0 || plan->nx == 0) return 0;
  a.nxtile = (int)((plan->nx + 255) / 256);
  const int64_t grid = plan->nkey * plan->ndepth * a.nxtile;
  WBX_REQUIRE(grid < (int64_t)1 << 31, "grid too large (%lld blocks)", (long long)grid);
  hipLaunchKernelGGL((s1_map_kernel<Op>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, a);
  WBX_HIP(hipGetLastError());
  return 0;
}

}  // namespace wbx
