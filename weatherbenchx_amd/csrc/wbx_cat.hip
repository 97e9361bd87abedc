// Indicator ("categorical") statistics: per point a 0/1 vector along a NEW dimension, summed like any other statistic.
//
// Reference semantics restated:
//   ErrorExceedance (weatherbenchX/metrics/deterministic.py:262-295)
//       out[.., k] = float(|p - t| > thr_k), NaN where |p - t| or thr_k is NaN
//   EnsembleErrorExceedance (weatherbenchX/metrics/probabilistic.py:836-861)
//       the mean over members of the above, NaN members skipped (xarray's default skipna), NaN if every member is NaN
//   RankHistogram (weatherbenchX/metrics/probabilistic.py:1306-1343)
//       out[.., r] = float(#{m : p_m < t} == r), r = 0..M; comparisons with NaN are False, the result is never NaN
//
// The category index is data dependent, so the per-thread accumulators cannot live in registers: each thread owns one
// column col[cat][thread] of counters in LDS (conflict-free: the bank depends on the thread only) -- fp64 for the exceedance
// fractions, uint32 for the rank histogram, whose sums are plain counts (r2: 52 categories x 64 threads x 8 B = 26.6 KB per
// one-wave block held a CU to six waves; with 4-byte counters and ds_add_u32 it is twelve).  One wave per
// block; same plan / partial layout as the other stage-1 kernels (partial[key][chunk][lane][j]), so stage 2 -- weights,
// bins, the patch contraction -- is shared.  Count lanes follow wbx_det_partial: one shared lane for WBX_FLAG_MASKED,
// one per category for WBX_FLAG_SKIPNA.
#include <cmath>
#include <type_traits>

#include "wbx_s1.hpp"

namespace wbx {

struct CatArgs {
  const double* thr;  // [ncat] thresholds (exceedance families), or (FIELD) the threshold field: input 2 of the plan
  int64_t cstride;    // FIELD: element stride of the category axis of the threshold field
  int32_t ncat;       // value lanes
  int32_t func;
};

template <typename T>
__device__ __forceinline__ T cat_ld(const void* base, int64_t off) {
  return reinterpret_cast<const T*>(base)[off];
}

// One point: adds its indicator vector (and count lanes) to this thread's LDS column.  MF > 0: the ensemble size is
// known at compile time (the 50 / 51-member archives): all member loads of a point are issued back to back into
// registers before the compares (the generic loop keeps 4-8 in flight); MF == 0: any M.
// FIELD: the thresholds depend on the statistic's own dims (per level, per latitude ...: deterministic.py:262-295 compares
// against any DataArray that broadcasts): threshold k of the point is thr[ro[2] + x * xstride[2] + k * cstride], float64.
template <typename T, int MF, typename C, bool FIELD = false>
__device__ __forceinline__ void cat_point(const S1Args& a, const CatArgs& c, const int64_t (&ro)[WBX_MAX_INPUTS],
                                          int64_t x, C* col, int stride) {
  constexpr bool RANK = std::is_same<C, uint32_t>::value;  // the column type says which family this instantiation serves
  const bool masked = a.flags & WBX_FLAG_MASKED, skipna = a.flags & WBX_FLAG_SKIPNA;
  const bool valid = masked ? cat_ld<uint8_t>(a.in[3], ro[3] + x * a.xstride[3]) != 0 : true;
  const int nc = c.ncat;
  const T tnat = cat_ld<T>(a.in[1], ro[1] + x * a.xstride[1]);
  const double t = (double)tnat;
  const T* pm = reinterpret_cast<const T*>(a.in[0]) + ro[0] + x * a.xstride[0];  // members: pm[m * mstride]
  T xm[MF > 0 ? MF : 1];
  if constexpr (MF > 0) {
#pragma unroll
    for (int m = 0; m < MF; ++m) xm[m] = ld_stream(pm + (int64_t)m * a.mstride);
  }
  if constexpr (RANK) {
    int r = 0;
    if constexpr (MF > 0) {
#pragma unroll
      for (int m = 0; m < MF; ++m) r += (xm[m] < tnat) ? 1 : 0;  // NaN compares false
    } else {
#pragma unroll 8
      for (int m = 0; m < a.M; ++m) r += (ld_stream(pm + (int64_t)m * a.mstride) < tnat) ? 1 : 0;
    }
    if (valid) col[r * stride] += 1u;
    if (skipna) {
      for (int k = 0; k < nc; ++k) col[(nc + k) * stride] += valid ? 1u : 0u;
    } else if (masked) {
      col[nc * stride] += valid ? 1u : 0u;
    }
    return;
  } else {
  // exceedance: per threshold the fraction of (non-NaN) members whose absolute error exceeds it; M = 1 without ensemble
  for (int k0 = 0; k0 < nc; k0 += 8) {
    int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int n = 0;
    double thr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if constexpr (FIELD)
        thr[q] = k0 + q < nc ? c.thr[ro[2] + x * a.xstride[2] + (int64_t)(k0 + q) * c.cstride] : INFINITY;
      else
        thr[q] = k0 + q < nc ? c.thr[k0 + q] : INFINITY;
    }
    if constexpr (MF > 0) {
#pragma unroll
      for (int m = 0; m < MF; ++m) {
        const double ae = fabs((double)xm[m] - t);
        n += (ae == ae) ? 1 : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) cnt[q] += (ae > thr[q]) ? 1 : 0;  // NaN > x is false
      }
    } else {
#pragma unroll 4
      for (int m = 0; m < a.M; ++m) {
        const double ae = fabs((double)ld_stream(pm + (int64_t)m * a.mstride) - t);
        n += (ae == ae) ? 1 : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) cnt[q] += (ae > thr[q]) ? 1 : 0;  // NaN > x is false
      }
    }
    const double inv = n > 0 ? 1.0 / (double)n : NAN;  // every member NaN -> NaN
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (k0 + q < nc) {
        // deterministic.py:293-294: a NaN threshold makes the indicator NaN -- at valid points only, so a cell whose
        // points are all masked out still sums to 0 (aggregation.py:339-352)
        const bool thr_ok = thr[q] == thr[q];
        const double v = thr_ok ? (double)cnt[q] * inv : NAN;
        if (skipna) {  // aggregation.py:353-355: NaN statistics are counted out, lane by lane
          const bool ok = valid && n > 0 && thr_ok;
          col[(k0 + q) * stride] += ok ? v : 0.0;
          col[(nc + k0 + q) * stride] += ok ? 1.0 : 0.0;
        } else {       // aggregation.py:339-352: masked-out points contribute 0; a NaN under a valid point poisons
          col[(k0 + q) * stride] += valid ? v : 0.0;
        }
      }
    }
  }
  if (masked && !skipna) col[nc * stride] += valid ? 1.0 : 0.0;
  }
}

// grid = nkey * nchunk (x summed) or nkey * nxtile * nchunk (x kept); block = 64 threads; dynamic LDS = nacc * 64 * 8 B
template <typename T, int MF, typename C, bool FIELD = false>
__global__ void __launch_bounds__(64) s1_cat_kernel(S1Args a, CatArgs c, int nacc, int x_kept) {
  constexpr int NIN = FIELD ? 3 : 2;
  extern __shared__ double cols_raw[];  // [nacc][64] of C
  C* const cols = reinterpret_cast<C*>(cols_raw);
  const int lane = threadIdx.x;
  for (int i = 0; i < nacc; ++i) cols[i * 64 + lane] = C(0);
  int64_t b = blockIdx.x;
  const int chunk = (int)(b % a.nchunk);
  b /= a.nchunk;
  int xt = 0;
  if (x_kept) {
    xt = (int)(b % a.nxtile);
    b /= a.nxtile;
  }
  const int64_t key = b;
  const int64_t d0 = (int64_t)chunk * a.dchunk;
  const int64_t d1 = d0 + a.dchunk < a.D ? d0 + a.dchunk : a.D;
  int64_t kb[WBX_MAX_INPUTS];
  key_bases<NIN>(a, key, kb);
  C* col = cols + lane;
  if (x_kept) {
    const int64_t x = (int64_t)xt * 64 + lane;
    if (x < a.nx) {
      for (int64_t d = d0; d < d1; ++d) {
        int64_t ro[WBX_MAX_INPUTS];
        row_bases<NIN>(a, kb, key, d, ro);
        cat_point<T, MF, C, FIELD>(a, c, ro, x, col, 64);
      }
      for (int i = 0; i < nacc; ++i) a.out[((key * a.nchunk + chunk) * nacc + i) * a.nx + x] = (double)col[i * 64];
    }
  } else {
    for (int64_t d = d0; d < d1; ++d) {
      int64_t ro[WBX_MAX_INPUTS];
      row_bases<NIN>(a, kb, key, d, ro);
      for (int64_t x = lane; x < a.nx; x += 64) cat_point<T, MF, C, FIELD>(a, c, ro, x, col, 64);
    }
    __syncthreads();
    for (int i = 0; i < nacc; ++i) {
      const double s = wave_sum((double)cols[i * 64 + lane]);
      if (lane == 0) a.out[(key * a.nchunk + chunk) * nacc + i] = s;
    }
  }
}

}  // namespace wbx

namespace wbx {
static int cat_common(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, int ncat, int M, int64_t member_stride, const void* p,
                      const void* t, const double* thresholds, bool field, int64_t cat_stride, const uint8_t* mask, double* partial_out) {
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(func == WBX_CAT_EXCEED || func == WBX_CAT_RANK, "unknown categorical family %d", func);
  WBX_REQUIRE(ncat >= 1 && ncat <= 256, "1..256 categories (got %d)", ncat);
  WBX_REQUIRE(M >= 1, "M must be >= 1");
  if (func == WBX_CAT_RANK) WBX_REQUIRE(ncat == M + 1, "rank histogram needs ncat == M + 1");
  const int nacc = ncat + ((plan->flags & WBX_FLAG_SKIPNA) ? ncat : ((plan->flags & WBX_FLAG_MASKED) ? 1 : 0));
  const bool rank = func == WBX_CAT_RANK;
  const size_t counter = rank ? sizeof(uint32_t) : sizeof(double);
  WBX_REQUIRE((size_t)nacc * 64 * counter <= 64 * 1024, "too many categories for the LDS columns (%d lanes)", nacc);
  if (rank)  // a thread's counters are uint32: it meets at most depth_chunk * ceil(nx / 64) points
    WBX_REQUIRE((double)plan->depth_chunk * (double)((plan->nx + 63) / 64) < 4.0e9, "rank histogram: more than 2^32 points per thread");
  if (plan->nkey == 0) return 0;
  WBX_REQUIRE(partial_out != nullptr, "partial_out is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  const int64_t nj = plan->x_kept ? plan->nx : 1;
  if (plan->ndepth == 0 || plan->nx == 0) {
    const size_t n = (size_t)plan->nkey * plan->nchunk * nacc * (size_t)nj;
    if (n) WBX_HIP(hipMemsetAsync(partial_out, 0, n * sizeof(double), ctx->stream));
    return 0;
  }
  WBX_REQUIRE(p != nullptr && t != nullptr, "p/t is NULL");
  if (func == WBX_CAT_EXCEED) WBX_REQUIRE(thresholds != nullptr, "thresholds is NULL");
  if (plan->flags & WBX_FLAG_MASKED) WBX_REQUIRE(mask != nullptr, "WBX_FLAG_MASKED set but mask is NULL");
  S1Args a;
  fill_args(plan, a);
  a.in[0] = p;
  a.in[1] = t;
  a.in[3] = mask;
  a.M = M;
  a.mstride = member_stride;
  a.out = partial_out;
  CatArgs c;
  c.thr = thresholds;
  c.cstride = cat_stride;
  c.ncat = ncat;
  c.func = func;
  a.nxtile = (int)((plan->nx + 63) / 64);
  const int64_t grid = plan->nkey * plan->nchunk * (plan->x_kept ? a.nxtile : 1);
  WBX_REQUIRE(grid < (int64_t)1 << 31, "grid too large (%lld blocks)", (long long)grid);
  const size_t lds = (size_t)nacc * 64 * counter;
#define WBX_LAUNCH_CAT(TT, MFIX)                                                                                              \
  do {                                                                                                                        \
    if (rank)                                                                                                                 \
      hipLaunchKernelGGL((s1_cat_kernel<TT, MFIX, uint32_t>), dim3((unsigned)grid), dim3(64), lds, ctx->stream, a, c, nacc,   \
                         (int)plan->x_kept);                                                                                  \
    else                                                                                                                      \
      hipLaunchKernelGGL((s1_cat_kernel<TT, MFIX, double>), dim3((unsigned)grid), dim3(64), lds, ctx->stream, a, c, nacc,     \
                         (int)plan->x_kept);                                                                                  \
  } while (0)
  if (field) {  // thresholds that depend on the statistic's dims: any M through the generic member loop
    if (dtype == WBX_F32)
      hipLaunchKernelGGL((s1_cat_kernel<float, 0, double, true>), dim3((unsigned)grid), dim3(64), lds, ctx->stream, a, c, nacc, (int)plan->x_kept);
    else if (dtype == WBX_F64)
      hipLaunchKernelGGL((s1_cat_kernel<double, 0, double, true>), dim3((unsigned)grid), dim3(64), lds, ctx->stream, a, c, nacc, (int)plan->x_kept);
    else
      return fail(WBX_ERR_INVALID, "unknown dtype %d", dtype);
    WBX_HIP(hipGetLastError());
    return 0;
  }
  if (dtype == WBX_F32) {
    if (M == 51) WBX_LAUNCH_CAT(float, 51);
    else if (M == 50) WBX_LAUNCH_CAT(float, 50);
    else WBX_LAUNCH_CAT(float, 0);
  } else if (dtype == WBX_F64) {
    WBX_LAUNCH_CAT(double, 0);
  } else {
    return fail(WBX_ERR_INVALID, "unknown dtype %d", dtype);
  }
#undef WBX_LAUNCH_CAT
  WBX_HIP(hipGetLastError());
  return 0;
}
}  // namespace wbx

extern "C" int wbx_cat_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, int ncat, int M,
                               int64_t member_stride, const void* p, const void* t, const double* thresholds,
                               const uint8_t* mask, double* partial_out) {
  return wbx::cat_common(ctx, plan, func, dtype, ncat, M, member_stride, p, t, thresholds, false, 0, mask, partial_out);
}

extern "C" int wbx_cat_exceed_field(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int ncat, int M, int64_t member_stride,
                                    const void* p, const void* t, const double* threshold_field, int64_t category_stride,
                                    const uint8_t* mask, double* partial_out) {
  return wbx::cat_common(ctx, plan, WBX_CAT_EXCEED, dtype, ncat, M, member_stride, p, t, threshold_field, true, category_stride, mask,
                         partial_out);
}
