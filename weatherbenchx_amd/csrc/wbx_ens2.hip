// Ensemble-valued predictions AND targets with per-point member counts on both sides: the statistics the reference writes as
//   CRPSSkill                          mean over the (prediction member, target member) pairs of |p_i - t_j|
//                                      (weatherbenchX/metrics/probabilistic.py:133-145)
//   UnbiasedEnsembleMeanSquaredError   (mean_i p - mean_j t)^2 - var(p) / n_p - var(t) / n_t, ddof = 1
//                                      (weatherbenchX/metrics/probabilistic.py:304-336)
// with skipna_ensemble=True: NaN members on either side are missing members, n_p / n_t are the per-point counts of the
// non-NaN ones, and a point without a valid pair (resp. with fewer than two members on a side) is NaN.  Neither is linear in
// the target member any more, so the one-launch-per-target-member route of the fused rank-form kernels does not apply; this
// is the from-memory fp64 op for them (members re-read from L1 / L2, O(M N) per point), on the stage-1 skeleton: same plan,
// same partial layout, same mask / skipna wrappers, so weights, bins and stage 2 are shared.
// Without WBX_FLAG_SKIPNA_ENS a NaN member makes the point NaN (the reference's skipna=False).
#include "wbx_ens_impl.hpp"

namespace wbx {

template <typename T>
struct Ens2Op {
  static constexpr int NIN = 2;
  static constexpr int NLANE = WBX_ENS2_LANES;
  static constexpr int NACC = WBX_ENS2_LANES;
  static constexpr int XR_UNROLL = 1, XK_UNROLL = 1, MIN_WAVES = 1;

  __device__ __forceinline__ static void values(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                                double (&val)[NLANE]) {
    const int M = a.M, N = a.lane;  // (S1Args::lane carries the target ensemble size here)
    const T* pp = reinterpret_cast<const T*>(a.in[0]) + ro[0] + x * a.xstride[0];
    const T* tp = reinterpret_cast<const T*>(a.in[1]) + ro[1] + x * a.xstride[1];
    const int64_t ps = a.mstride, ts = (int64_t)a.ngd_t;
    const bool skip = a.flags & WBX_FLAG_SKIPNA_ENS;
    // one-pass moments on e = x - first valid member (shift invariant, no cancellation for tight ensembles)
    double p0 = 0.0, t0 = 0.0, pse = 0.0, psq = 0.0, tse = 0.0, tsq = 0.0, poison = 0.0;
    int np = 0, nt = 0;
    for (int i = 0; i < M; ++i) {
      const double v = (double)pp[i * ps];
      if (v != v) {
        if (!skip) poison = v;
        continue;
      }
      if (np == 0) p0 = v;
      ++np;
      const double e = v - p0;
      pse += e;
      psq = fma(e, e, psq);
    }
    for (int j = 0; j < N; ++j) {
      const double v = (double)tp[j * ts];
      if (v != v) {
        if (!skip) poison = v;
        continue;
      }
      if (nt == 0) t0 = v;
      ++nt;
      const double e = v - t0;
      tse += e;
      tsq = fma(e, e, tsq);
    }
    double pairs = 0.0;
    for (int j = 0; j < N; ++j) {
      const double tj = (double)tp[j * ts];
      if (tj != tj) continue;
      double row = 0.0;
      for (int i = 0; i < M; ++i) {
        const double d = fabs((double)pp[i * ps] - tj);
        row += (d != d) ? 0.0 : d;
      }
      pairs += row;
    }
    const double dnp = (double)np, dnt = (double)nt;
    const double nan = __builtin_nan("");
    val[0] = (np > 0 && nt > 0) ? pairs / (dnp * dnt) : nan;
    const double pm = pse / dnp, tm = tse / dnt;                     // means of e (0 / 0 = NaN without members)
    const double pvar = np > 1 ? (psq - pse * pm) / (dnp - 1.0) : nan;  // ddof = 1: NaN with fewer than two members
    const double tvar = nt > 1 ? (tsq - tse * tm) / (dnt - 1.0) : nan;
    const double md = (p0 - t0) + (pm - tm);
    val[1] = md * md - pvar / dnp - tvar / dnt;
    if (poison != poison) val[0] = val[1] = poison;
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

template <typename T>
static int launch_ens2(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a) {
  if (plan->flags & WBX_FLAG_SKIPNA) return launch_partial<EnsMasked<Ens2Op<T>, true>, 1>(ctx, plan, a);
  if (plan->flags & WBX_FLAG_MASKED) return launch_partial<EnsMasked<Ens2Op<T>, false>, 1>(ctx, plan, a);
  return launch_partial<Ens2Op<T>, 1>(ctx, plan, a);
}

}  // namespace wbx

extern "C" int wbx_ens2_partial(wbx_ctx* ctx, const wbx_s1_plan* plan, int dtype, int M, int64_t member_stride, int N,
                                int64_t target_member_stride, const void* p, const void* t, const uint8_t* mask,
                                double* partial_out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(plan->vec == 1 && plan->x_weights == nullptr, "wbx_ens2_partial uses vec = 1 and no folded weights");
  WBX_REQUIRE(M >= 1 && N >= 1, "ensemble sizes must be >= 1 (got %d, %d)", M, N);
  WBX_REQUIRE(target_member_stride == (int64_t)(int32_t)target_member_stride || target_member_stride >= 0, "bad target member stride");
  if (plan->flags & WBX_FLAG_MASKED) WBX_REQUIRE(mask != nullptr, "WBX_FLAG_MASKED set but mask is NULL");
  const bool empty = plan->nkey * plan->ndepth * plan->nx == 0;
  WBX_REQUIRE(empty || (p != nullptr && t != nullptr), "predictions/targets pointer is NULL");
  WBX_REQUIRE(partial_out != nullptr || plan->nkey == 0, "output pointer is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  S1Args a;
  fill_args(plan, a);
  a.in[0] = p;
  a.in[1] = t;
  a.in[3] = mask;
  a.out = partial_out;
  a.M = M;
  a.mstride = member_stride;
  a.lane = N;
  a.ngd_t = target_member_stride;
  if (dtype == WBX_F32) return launch_ens2<float>(ctx, plan, a);
  if (dtype == WBX_F64) return launch_ens2<double>(ctx, plan, a);
  return fail(WBX_ERR_INVALID, "unknown dtype %d", dtype);
}
