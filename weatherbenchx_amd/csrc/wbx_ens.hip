// Ensemble entry points + the generic (any M, fp32/fp64) ensemble op.
// The register-bucket fast paths live in wbx_ens_m*.hip (see wbx_ens_impl.hpp).
#include "wbx_ens_impl.hpp"

namespace wbx {

// Generic op: members are re-read from memory (L1/L2-served) instead of living in VGPRs.
// Always uses the O(M^2) pair form in fp64 -- algebraically identical to the rank form
// (probabilistic.py:214-247) -- so it serves M > 64 and float64 inputs (the reference's
// mock test data is float64, test_utils.py:36-48).
template <typename T>
struct EnsOpGeneric {
  static constexpr int NIN = 2;
  static constexpr int NLANE = WBX_ENS_LANES;
  static constexpr int NACC = WBX_ENS_LANES;
  static constexpr int XR_UNROLL = 1, XK_UNROLL = 1, MIN_WAVES = 1;

  __device__ __forceinline__ static void values(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                                double (&val)[NLANE]) {
    const int M = a.M;
    const T* pp = reinterpret_cast<const T*>(a.in[0]) + ro[0] + x * a.xstride[0];
    const double td = (double)(reinterpret_cast<const T*>(a.in[1])[ro[1] + x * a.xstride[1]]);
    // skipna_ensemble (probabilistic.py:139-145, 206-216, 271-273, 303-336): NaN members are missing members; the
    // ensemble size becomes the per-point count of non-NaN values.
    const bool skip = a.flags & WBX_FLAG_SKIPNA_ENS;
    double se = 0.0, sq = 0.0, sabs = 0.0, pair_total = 0.0, x0 = 0.0;
    int n = 0;
    for (int i = 0; i < M; ++i) {
      const double xi = (double)pp[(int64_t)i * a.mstride];
      if (skip && xi != xi) continue;
      if (n == 0) x0 = xi;  // member-only lanes use e = x - x0 and stay finite for a NaN target
      ++n;
      const double e = xi - x0;
      se += e;
      sq = fma(e, e, sq);
      sabs += fabs(xi - td);
      double row = 0.0;
      for (int j = 0; j < i; ++j) {
        const double xj = (double)pp[(int64_t)j * a.mstride];
        if (skip && xj != xj) continue;
        row += fabs(xi - xj);
      }
      pair_total += row;
    }
    const double dM = (double)n;
    const double fair = (a.flags & WBX_FLAG_FAIR) ? 1.0 : 0.0;
    const double mean_e = se / dM;
    const double mean_d = (x0 - td) + mean_e;
    const double var = (sq - se * mean_e) / (dM - 1.0);
    val[0] = sabs / dM;
    val[1] = 2.0 * pair_total / (dM * (dM - fair));
    val[2] = var;
    val[3] = mean_d * mean_d - var / dM;
    val[4] = mean_d * mean_d;
  }

  template <int V, bool XK>
  __device__ __forceinline__ static void accum(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                               double (&acc)[XK ? V : 1][NACC]) {
    static_assert(V == 1, "ensemble op is one point per lane");
    double val[NLANE];
    values(a, ro, x, val);
#pragma unroll
    for (int l = 0; l < NLANE; ++l) acc[0][l] += val[l];
  }
};

static int ens_common(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride, int algo,
                      const void* p, const void* t, const uint8_t* mask, double* out, bool map, int lane) {
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(plan->vec == 1, "ensemble kernels use vec=1");
  if (plan->flags & WBX_FLAG_MASKED) WBX_REQUIRE(mask != nullptr, "WBX_FLAG_MASKED set but mask is NULL");
  WBX_REQUIRE(M >= 1, "ensemble size must be >= 1 (got %d)", M);
  WBX_REQUIRE(algo == WBX_ENS_SORT || algo == WBX_ENS_PAIRWISE || (algo == WBX_ENS_DIAG_LOADONLY && (M == 50 || M == 51)) ||
                  (algo == WBX_ENS_DIAG_PAIRWISE_LDS && M == 51),
              "unknown ensemble algorithm %d", algo);
  const bool empty = plan->nkey * plan->ndepth * plan->nx == 0;
  WBX_REQUIRE(empty || (p != nullptr && t != nullptr), "predictions/targets pointer is NULL");
  WBX_REQUIRE(out != nullptr || plan->nkey == 0, "output pointer is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  S1Args a;
  fill_args(plan, a);
  a.in[0] = p;
  a.in[1] = t;
  a.in[3] = mask;
  a.out = out;
  a.M = M;
  a.mstride = member_stride;
  a.lane = lane;
  if (dtype == WBX_F64) return launch_ens_op<EnsOpGeneric<double>>(ctx, plan, a, map);
  if (plan->flags & WBX_FLAG_SKIPNA_ENS) return launch_ens_op<EnsOpGeneric<float>>(ctx, plan, a, map);
  WBX_REQUIRE(dtype == WBX_F32, "unknown dtype %d", dtype);
  if (M == 51) return launch_ens_m51(ctx, plan, a, algo, map);
  if (M == 50) return launch_ens_m50(ctx, plan, a, algo, map);
  if (M <= 4) return launch_ens_m4(ctx, plan, a, algo, map);
  if (M <= 8) return launch_ens_m8(ctx, plan, a, algo, map);
  if (M <= 16) return launch_ens_m16(ctx, plan, a, algo, map);
  if (M <= 32) return launch_ens_m32(ctx, plan, a, algo, map);
  if (M <= 64) return launch_ens_m64(ctx, plan, a, algo, map);
  return launch_ens_op<EnsOpGeneric<float>>(ctx, plan, a, map);
}

}  // namespace wbx

extern "C" int wbx_ens_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride,
                               int algo, const void* p, const void* t, const uint8_t* mask, double* partial_out) {
  return wbx::ens_common(ctx, plan, dtype, M, member_stride, algo, p, t, mask, partial_out, false, 0);
}

extern "C" int wbx_ens_map(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride, int algo,
                           int lane, const void* p, const void* t, double* out) {
  if (lane < 0 || lane >= WBX_ENS_LANES) return wbx::fail(WBX_ERR_INVALID, "ensemble lane %d out of range", lane);
  return wbx::ens_common(ctx, plan, dtype, M, member_stride, algo, p, t, nullptr, out, true, lane);
}
