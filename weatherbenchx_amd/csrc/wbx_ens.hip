// Ensemble entry points + the generic (any M, fp32/fp64) ensemble op.
// The register-bucket fast paths live in wbx_ens_m*.hip (see wbx_ens_impl.hpp).
#include "wbx_ens_impl.hpp"

namespace wbx {

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
