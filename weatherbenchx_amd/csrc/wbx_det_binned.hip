// Fused "binned" deterministic reduction: per-point statistics, weights and boolean bin membership in ONE pass.
//
// The two-stage path (wbx_det_partial + wbx_contract_bits) has to keep every dimension the bin masks depend on
// (latitude AND longitude for Regions x land/sea) in the stage-1 partials.  When little is reduced before them --
// the public benchmark's chunks are 1 init x 12 leads (public_benchmark/run_benchmark_evaluation.py:97-101) -- those
// partials are 6 lanes x 8 B per grid point, 4x the inputs, and stage 2 re-reads them per lane.  This kernel instead
// re-derives the statistics per bin group straight from p, t, c (L2 / Infinity-Cache served on the repeats) and
// accumulates  acc[bin][lane] = fma(w(point) * stat(point), member(bin, point), acc)  in registers:
//   grid  = nA * nBk * ngroup * nsplit     (bin group of BG bins, split of the nBr*D reduced rows)
//   block = 4 waves; waves interleave over rows, lanes over x (any stride); block fold -> tmp[..][split][lane][bin]
// A second tiny kernel sums the splits.  Same semantics as aggregation.py:297-366: NaN * 0 = NaN poisons every bin of
// a lane (member is 0.0 / 1.0 and the FMA propagates it), mask / skipna count lanes follow the conventions of wbx_det_partial.
#include <type_traits>

#include "wbx_s1.hpp"

namespace wbx {

struct BinnedArgs {
  const double* wt;                  // [nBk][nBr][nj]
  const unsigned long long* bits;    // [nBk][nBr][nj]
  int64_t nBk, nBr, nj;              // nj = nx if W depends on x, else 1
  int32_t nbin, ngroup, nsplit;
  int64_t rows_per_split;            // rows = nBr * D
  double* tmp;                       // [nA][nBk][ngroup][nsplit][NACC][BG]
};

template <typename T>
__device__ __forceinline__ T ld1(const void* base, int64_t off, int64_t x, int64_t xs) {
  return reinterpret_cast<const T*>(base)[off + x * xs];
}

// MM: 0 none, 1 mask only (one shared count lane), 2 skipna (count lane per value lane)
template <typename T, int FUNC, int MM, int BG>
__global__ void __launch_bounds__(256) det_binned_kernel(S1Args a, BinnedArgs g) {
  constexpr int NIN = FUNC == WBX_DET6 ? 3 : (FUNC == WBX_DET3 ? 2 : 1);
  constexpr int NL = FUNC == WBX_DET6 ? 6 : (FUNC == WBX_DET3 ? 3 : 1);
  constexpr int NC = MM == 1 ? 1 : (MM == 2 ? NL : 0);
  constexpr int NA = NL + NC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  int64_t b = blockIdx.x;
  const int split = (int)(b % g.nsplit);
  b /= g.nsplit;
  const int grp = (int)(b % g.ngroup);
  b /= g.ngroup;
  const int64_t bk = b % g.nBk;
  const int64_t A = b / g.nBk;
  const int bin0 = grp * BG;

  double acc[NA][BG];
#pragma unroll
  for (int l = 0; l < NA; ++l)
#pragma unroll
    for (int q = 0; q < BG; ++q) acc[l][q] = 0.0;
  const int64_t R = g.nBr * a.D;
  const int64_t r0 = (int64_t)split * g.rows_per_split;
  const int64_t r1 = r0 + g.rows_per_split < R ? r0 + g.rows_per_split : R;
  for (int64_t r = r0 + wave; r < r1; r += nwave) {
    const int64_t br = r / a.D;
    const int64_t d = r - br * a.D;
    const int64_t key = (A * g.nBk + bk) * g.nBr + br;
    int64_t kb[WBX_MAX_INPUTS], ro[WBX_MAX_INPUTS];
    key_bases<NIN>(a, key, kb);
    row_bases<NIN>(a, kb, key, d, ro);
    const int64_t wrow = (bk * g.nBr + br) * g.nj;
    for (int64_t x = lane; x < a.nx; x += 64) {
      const double p = (double)ld1<T>(a.in[0], ro[0], x, a.xstride[0]);
      const double t = NIN > 1 ? (double)ld1<T>(a.in[1], ro[1], x, a.xstride[1]) : 0.0;
      const double c = NIN > 2 ? (double)ld1<T>(a.in[2], ro[2], x, a.xstride[2]) : 0.0;
      double val[NA];
      if constexpr (FUNC == WBX_PASS1) {
        val[0] = p;
      } else {
        const double e = p - t;
        val[0] = e;
        val[1] = fabs(e);
        val[2] = e * e;
        if constexpr (FUNC == WBX_DET6) {
          const double pa = p - c, ta = t - c;
          val[3] = pa * pa;
          val[4] = ta * ta;
          val[5] = pa * ta;
        }
      }
      if constexpr (MM != 0) {
        const bool valid = (a.flags & WBX_FLAG_MASKED) ? ld1<uint8_t>(a.in[3], ro[3], x, a.xstride[3]) != 0 : true;
        if constexpr (MM == 1) {
#pragma unroll
          for (int l = 0; l < NL; ++l) val[l] = valid ? val[l] : 0.0;
          val[NL] = valid ? 1.0 : 0.0;
        } else {
#pragma unroll
          for (int l = 0; l < NL; ++l) {
            const bool ok = valid && !(val[l] != val[l]);
            val[NL + l] = ok ? 1.0 : 0.0;
            val[l] = ok ? val[l] : 0.0;
          }
        }
      }
      const int64_t wi = wrow + (g.nj > 1 ? x : 0);
      const double w = g.wt[wi];
      const unsigned long long bw = g.bits[wi] >> bin0;
      const unsigned word = (unsigned)bw;
      double m[NA];
#pragma unroll
      for (int l = 0; l < NA; ++l) m[l] = val[l] * w;
#pragma unroll
      for (int q = 0; q < BG; ++q) {
        // membership as 0.0 / 1.0: the FMA keeps IEEE NaN * 0 = NaN, i.e. a NaN statistic poisons every bin exactly
        // as (stat * weights * mask).sum() does in the reference (aggregation.py:335)
        const double f = __hiloint2double(((word >> q) & 1u) ? 0x3FF00000 : 0, 0);
#pragma unroll
        for (int l = 0; l < NA; ++l) acc[l][q] = fma(m[l], f, acc[l][q]);
      }
    }
  }
  __shared__ double red[4][NA * BG];
#pragma unroll
  for (int l = 0; l < NA; ++l)
#pragma unroll
    for (int q = 0; q < BG; ++q) {
      const double s = wave_sum(acc[l][q]);
      if (lane == 0) red[wave][l * BG + q] = s;
    }
  __syncthreads();
  if (threadIdx.x < NA * BG) {
    double s = 0.0;
    for (int w2 = 0; w2 < nwave; ++w2) s += red[w2][threadIdx.x];
    g.tmp[((((A * g.nBk + bk) * g.ngroup + grp) * g.nsplit + split) * (NA * BG)) + threadIdx.x] = s;
  }
}

// tmp[cell][group][split][lane][q] -> out[cell][lane][bin]
__global__ void __launch_bounds__(256) det_binned_finish(int64_t ncell, int ngroup, int nsplit, int nacc, int bg, int nbin,
                                                         const double* __restrict__ tmp, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ncell * nacc * nbin) return;
  const int bin = (int)(i % nbin);
  const int l = (int)((i / nbin) % nacc);
  const int64_t cell = i / ((int64_t)nbin * nacc);
  const int grp = bin / bg, q = bin - grp * bg;
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += tmp[(((cell * ngroup + grp) * nsplit + k) * nacc + l) * bg + q];
  out[i] = s;
}

template <typename T, int FUNC, int MM>
static int launch_binned(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, const double* wt, const uint64_t* bits,
                         int64_t nA, int64_t nBk, int64_t nBr, int64_t nj, int nbin, double* out) {
  constexpr int NL = FUNC == WBX_DET6 ? 6 : (FUNC == WBX_DET3 ? 3 : 1);
  constexpr int NA = NL + (MM == 1 ? 1 : (MM == 2 ? NL : 0));
  constexpr int BG = NA <= 4 ? 8 : (NA <= 7 ? 6 : 4);  // accumulators per lane: NA * BG fp64 (<= 48)
  BinnedArgs g;
  g.wt = wt;
  g.bits = reinterpret_cast<const unsigned long long*>(bits);
  g.nBk = nBk;
  g.nBr = nBr;
  g.nj = nj;
  g.nbin = nbin;
  g.ngroup = (nbin + BG - 1) / BG;
  const int64_t rows = nBr * plan->ndepth;
  const int64_t cells = nA * nBk;
  int64_t want = (8192 + cells * g.ngroup - 1) / (cells * g.ngroup);  // blocks per (cell, group)
  if (want > (rows + 3) / 4) want = (rows + 3) / 4;                   // >= 4 rows per block (one per wave)
  if (want < 1) want = 1;
  g.rows_per_split = (rows + want - 1) / want;
  g.nsplit = (int)((rows + g.rows_per_split - 1) / g.rows_per_split);
  const size_t need = (size_t)cells * g.ngroup * g.nsplit * NA * BG * sizeof(double);
  if (ctx->s2_scratch_size < need) {
    if (ctx->s2_scratch) {
      WBX_HIP(hipStreamSynchronize(ctx->stream));
      WBX_HIP(hipFree(ctx->s2_scratch));
    }
    WBX_HIP(hipMalloc(&ctx->s2_scratch, need));
    ctx->s2_scratch_size = need;
  }
  g.tmp = reinterpret_cast<double*>(ctx->s2_scratch);
  const int64_t grid = cells * g.ngroup * g.nsplit;
  WBX_REQUIRE(grid < (int64_t)1 << 31, "binned grid too large");
  hipLaunchKernelGGL((det_binned_kernel<T, FUNC, MM, BG>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, a, g);
  WBX_HIP(hipGetLastError());
  const int64_t n = cells * NA * nbin;
  hipLaunchKernelGGL(det_binned_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, cells, g.ngroup,
                     g.nsplit, NA, BG, nbin, g.tmp, out);
  WBX_HIP(hipGetLastError());
  return 0;
}

template <typename T, int FUNC>
static int binned_mm(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, const double* wt, const uint64_t* bits, int64_t nA,
                     int64_t nBk, int64_t nBr, int64_t nj, int nbin, double* out) {
  if (plan->flags & WBX_FLAG_SKIPNA) return launch_binned<T, FUNC, 2>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out);
  if (plan->flags & WBX_FLAG_MASKED) return launch_binned<T, FUNC, 1>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out);
  return launch_binned<T, FUNC, 0>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out);
}

template <typename T>
static int binned_func(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, S1Args& a, const double* wt,
                       const uint64_t* bits, int64_t nA, int64_t nBk, int64_t nBr, int64_t nj, int nbin, double* out) {
  switch (func) {
    case WBX_DET3:
      return binned_mm<T, WBX_DET3>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out);
    case WBX_DET6:
      return binned_mm<T, WBX_DET6>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out);
    case WBX_PASS1:
      return binned_mm<T, WBX_PASS1>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out);
  }
  return fail(WBX_ERR_INVALID, "unknown deterministic family %d", func);
}

}  // namespace wbx

extern "C" int wbx_det_binned(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, const void* p, const void* t,
                              const void* c, const uint8_t* mask, const double* wt, const uint64_t* bits, int64_t nA,
                              int64_t nBk, int64_t nBr, int32_t w_on_x, int32_t nbin, double* out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(nbin >= 1 && nbin <= 64, "wbx_det_binned handles 1..64 bins (got %d)", nbin);
  WBX_REQUIRE(nA >= 0 && nBk >= 0 && nBr >= 0 && nA * nBk * nBr == plan->nkey, "nA*nBk*nBr must equal plan->nkey");
  const int nl = func == WBX_DET6 ? 6 : (func == WBX_DET3 ? 3 : 1);
  const int na = nl + ((plan->flags & WBX_FLAG_SKIPNA) ? nl : ((plan->flags & WBX_FLAG_MASKED) ? 1 : 0));
  const int64_t nout = nA * nBk * na * nbin;
  if (nout == 0) return 0;
  WBX_REQUIRE(out != nullptr, "out is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  if (nBr * plan->ndepth * plan->nx == 0) {
    WBX_HIP(hipMemsetAsync(out, 0, (size_t)nout * sizeof(double), ctx->stream));
    return 0;
  }
  WBX_REQUIRE(p != nullptr && wt != nullptr && bits != nullptr, "p/wt/bits is NULL");
  if (func != WBX_PASS1) WBX_REQUIRE(t != nullptr, "targets pointer is NULL");
  if (func == WBX_DET6) WBX_REQUIRE(c != nullptr, "climatology pointer is NULL");
  if (plan->flags & WBX_FLAG_MASKED) WBX_REQUIRE(mask != nullptr, "WBX_FLAG_MASKED set but mask is NULL");
  S1Args a;
  fill_args(plan, a);
  a.in[0] = p;
  a.in[1] = t;
  a.in[2] = c;
  a.in[3] = mask;
  const int64_t nj = w_on_x ? plan->nx : 1;
  if (dtype == WBX_F32) return binned_func<float>(ctx, plan, func, a, wt, bits, nA, nBk, nBr, nj, nbin, out);
  if (dtype == WBX_F64) return binned_func<double>(ctx, plan, func, a, wt, bits, nA, nBk, nBr, nj, nbin, out);
  return fail(WBX_ERR_INVALID, "unknown dtype %d", dtype);
}
