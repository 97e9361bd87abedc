// Ensemble statistic family for the stage-1 skeleton (one lane owns one grid point's M members).
//
// Reference semantics restated (weatherbenchX/metrics/probabilistic.py):
//   CRPSSkill :116-145                 mean_m |p_m - t|
//   CRPSSpread :165-247                rank form  2 * mean_m((2 r_m - M - 1) p_m) / (M - fair)      (:231-240)
//                                      pair form  sum_{m,m'} |p_m - p_m'| / (M (M - fair))          (:241-247)
//   EnsembleVariance :250-273          var_m(p, ddof=1)
//   UnbiasedEnsembleMeanSquaredError :276-336   (mean_m p - t)^2 - var/M
//   EnsembleMean + SquaredError (wrappers.py:116-148, deterministic.py:115-123)   (mean_m p - t)^2
//
// fp32 members are sorted exactly with a v_min/v_max network in VGPRs; every sum is fp64 on the
// widened values, so each per-point value matches the float64 restatement to ~1e-15 (rank form)
// or ~1e-7 (pair form: the |x_i - x_j| row sums are fp32, their total fp64).
// Members beyond the runtime M (padded buckets) are +inf for the network and skipped in the sums.
#pragma once
#include <cmath>

#include "wbx_s1.hpp"
#include "wbx_sortnet_gen.hpp"

#ifndef WBX_ENS_MIN_WAVES
#define WBX_ENS_MIN_WAVES 4  // waves per SIMD the ensemble kernels are register-budgeted for (<= 128 VGPRs)
#endif

namespace wbx {

struct EnsLanes {
  double skill, spread, var, uemse, emse;
};

// MP: register bucket (compile time).  EXACT: M == MP known at compile time.
template <int MP, bool EXACT, int ALGO>
struct EnsOpF32 {
  static constexpr int NIN = 2;
  static constexpr int NLANE = WBX_ENS_LANES;
  static constexpr int NACC = WBX_ENS_LANES;
  static constexpr int XR_UNROLL = 1, XK_UNROLL = 1, MIN_WAVES = WBX_ENS_MIN_WAVES;

  __device__ __forceinline__ static void values(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                                double (&val)[NLANE]) {
    const int M = EXACT ? MP : a.M;
    const float* pp = reinterpret_cast<const float*>(a.in[0]) + ro[0] + x * a.xstride[0];
    const double td = (double)(reinterpret_cast<const float*>(a.in[1])[ro[1] + x * a.xstride[1]]);
    float xm[MP];
#pragma unroll
    for (int m = 0; m < MP; ++m) xm[m] = (EXACT || m < M) ? pp[(int64_t)m * a.mstride] : INFINITY;

    double pair_total = 0.0;
    float poison = 0.f;  // NaN iff any member is NaN/inf (v_min/v_max would silently drop a NaN)
    if constexpr (ALGO == WBX_ENS_PAIRWISE) {
#pragma unroll
      for (int i = 1; i < MP; ++i) {
        if (EXACT || i < M) {
          float row = 0.f;
#pragma unroll
          for (int j = 0; j < i; ++j) row += fabsf(xm[i] - xm[j]);
          pair_total += (double)row;
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < MP; ++m)
        if (EXACT || m < M) poison = fmaf(xm[m], 0.f, poison);
      SortNet<MP>::sort(
          xm, [](float u, float v) { return fminf(u, v); }, [](float u, float v) { return fmaxf(u, v); });
    }

    double sum = 0.0, sq = 0.0, sabs = 0.0, dot = 0.0;
#pragma unroll
    for (int m = 0; m < MP; ++m) {
      if (EXACT || m < M) {
        const double d = (double)xm[m] - td;
        sum += d;
        sq = fma(d, d, sq);
        sabs += fabs(d);
        if constexpr (ALGO == WBX_ENS_SORT) dot = fma((double)(2 * (m + 1) - M - 1), d, dot);
      }
    }
    const double dM = (double)M;
    const double fair = (a.flags & WBX_FLAG_FAIR) ? 1.0 : 0.0;
    const double mean_d = sum / dM;                              // mean_m p - t
    const double var = (sq - sum * mean_d) / (dM - 1.0);         // ddof = 1
    double spread;
    if constexpr (ALGO == WBX_ENS_SORT) {
      spread = 2.0 * dot / (dM * (dM - fair));
    } else {
      spread = 2.0 * pair_total / (dM * (dM - fair));
    }
    val[0] = sabs / dM;
    val[1] = spread;
    val[2] = var;
    val[3] = mean_d * mean_d - var / dM;
    val[4] = mean_d * mean_d;
    if constexpr (ALGO == WBX_ENS_SORT) {
      if (poison != poison) {  // reference: a NaN member makes every ensemble statistic NaN
#pragma unroll
        for (int l = 0; l < NLANE; ++l) val[l] = (double)poison;
      }
    }
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

template <int MP, bool EXACT>
int launch_ens_bucket(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, int algo, bool map) {
  if (algo == WBX_ENS_SORT) {
    using Op = EnsOpF32<MP, EXACT, WBX_ENS_SORT>;
    return map ? launch_map<Op>(ctx, plan, a) : launch_partial<Op, 1>(ctx, plan, a);
  }
  using Op = EnsOpF32<MP, EXACT, WBX_ENS_PAIRWISE>;
  return map ? launch_map<Op>(ctx, plan, a) : launch_partial<Op, 1>(ctx, plan, a);
}

// one translation unit per bucket (parallel build)
int launch_ens_m4(wbx_ctx*, const wbx_s1_plan*, S1Args&, int, bool);
int launch_ens_m8(wbx_ctx*, const wbx_s1_plan*, S1Args&, int, bool);
int launch_ens_m16(wbx_ctx*, const wbx_s1_plan*, S1Args&, int, bool);
int launch_ens_m32(wbx_ctx*, const wbx_s1_plan*, S1Args&, int, bool);
int launch_ens_m64(wbx_ctx*, const wbx_s1_plan*, S1Args&, int, bool);
int launch_ens_m50(wbx_ctx*, const wbx_s1_plan*, S1Args&, int, bool);
int launch_ens_m51(wbx_ctx*, const wbx_s1_plan*, S1Args&, int, bool);

}  // namespace wbx
