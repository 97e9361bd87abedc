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
// fp32 members are sorted exactly with a network of compare-exchanges (v_min / v_max) and 3-sorters (v_min3 / v_med3 /
// v_max3) in VGPRs (gen_sortnet3.py); every sum is fp64 on the
// widened values, so each per-point value matches the float64 restatement to ~1e-15 (rank form)
// or ~1e-7 (pair form: the |x_i - x_j| row sums are fp32, their total fp64).
// Members beyond the runtime M (padded buckets) are +inf for the network and skipped in the sums.
#pragma once
#include <cmath>

#include "wbx_s1.hpp"
#include "wbx_sortnet_gen.hpp"
#include "wbx_sortnet3_gen.hpp"

#ifndef WBX_ENS_MIN_WAVES
#define WBX_ENS_MIN_WAVES 1  // no forced occupancy: 106 VGPRs / 4 waves per SIMD on its own since the fp64 divisions went (before: 140-155
                             // VGPRs / 3 waves, and forcing 4 spilled 132 B: 0.63 ms vs 0.37 ms)
#endif

#ifndef WBX_ENS_MIN_WAVES_PLAIN
#define WBX_ENS_MIN_WAVES_PLAIN WBX_ENS_MIN_WAVES  // the unmasked op alone (the masked / skipna wrappers keep WBX_ENS_MIN_WAVES)
#endif
#ifndef WBX_ENS_SORTNET3
#define WBX_ENS_SORTNET3 1  // 0: Batcher's compare-exchange network for every size (A/B timing: make EXTRA=-DWBX_ENS_SORTNET3=0)
#endif

namespace wbx {

// not part of the ABI enum: stage-1 memory-pattern diagnostic used by tools/kbench.py (lane 0 = sum_m p - t)
constexpr int WBX_ENS_DIAG_LOADONLY = 99;
// not part of the ABI enum either: the north_star's "LDS-tiled pairwise" form, built to be MEASURED (tools/kbench.py) next to
// the register-tiled pair form that `use_sort=False` runs: a point's members are parked in the LDS (one column per lane: bank =
// lane, conflict free, no barrier -- a lane reads back only what it wrote) after the member sums have consumed them, and the
// 1275 |x_i - x_j| terms are formed on register blocks of 17 members read back from there (34 live member registers instead of
// 51).  M == 51 only.
constexpr int WBX_ENS_DIAG_PAIRWISE_LDS = 98;

// MP: register bucket (compile time).  EXACT: M == MP known at compile time.
template <int MP, bool EXACT, int ALGO>
struct EnsOpF32 {
  static constexpr int NIN = 2;
  static constexpr int NLANE = WBX_ENS_LANES;
  static constexpr int NACC = WBX_ENS_LANES;
  static constexpr int XR_UNROLL = 1, XK_UNROLL = 1, MIN_WAVES = WBX_ENS_MIN_WAVES_PLAIN;

  // One grid point's inputs in VGPRs.  load() only issues the (coalesced-across-lanes) member loads; compute() is
  // pure register work, so the skeleton can keep the NEXT point's loads in flight while this one is reduced.
  struct Regs {
    float xm[MP];
    float t;
  };
  // (Register double-buffering of the next point was tried on MI355X: 255 VGPRs, 2 waves/SIMD, 0.49 ms vs 0.39 ms -- dropped.)

  // Measured alternatives on MI355X (tools/kbench.py, M = 51, 8 x 721 x 1440 points, 0.375 ms baseline):
  //   * SGPR buffer descriptors + scalar member offsets (no 64-bit VGPR addresses): 0.373 ms rank form (noise),
  //     but 0.60 ms vs 0.52 ms for the pair form -> not kept;
  //   * __launch_bounds__ for 4 waves/SIMD (128 VGPRs): 88-164 B of spills, 0.60-0.63 ms -> not kept;
  //   * register double-buffering of the next point: 255 VGPRs, 2 waves/SIMD, 0.49 ms -> not kept;
  //   * 2 / 4 interleaved fp64 accumulation chains instead of one: 0.39 ms either way -> not kept.
  // rocprofv3 SQ counters (tools/pmc_ens.sh): WAIT_ANY (memory) 12 %, the VALU pipe is ~saturated: 1453 VALU
  // instructions per 64 points (877 v_min/v_max for the 415-comparator network, ~430 fp64), i.e. VALU-, not HBM-bound.
  // Instruction diet since then (ISA counts of the inner loop, per 64 points): 1524 -> 1395: reciprocal multiplies for
  // the five divisions by M-derived constants, packed NaN probe, bare v_min/v_max (no canonicalisation of the loaded
  // members), statistics accumulated on x - t (one fp64 add per member less): 0.404 -> 0.363 ms.
  // The load-only diagnostic (WBX_ENS_DIAG_LOADONLY) streams the same 52 dword streams at 6.0 TB/s, so what is
  // left is VALU time (~2000 instructions per 64 points) that 3 waves/SIMD only partly overlap with the loads.
  // (r2) The network itself: Batcher's 415 compare-exchanges (830 instructions) -> 54 compare-exchanges + 166 3-sorters (606):
  // 0.349 -> 0.316 ms on one box (tools/gpu_ens_ab.sh), 133 VGPRs / 3 waves per SIMD; forcing 4 waves (5 spilled registers)
  // changes nothing.  A wave-uniform row base + one 32-bit VGPR offset for all 52 streams (to drop the 53 v_lshl_add_u64 per
  // point): the compiler reassociates the sum back into 64-bit VGPR addresses, 107 VGPRs, 0.329 against 0.312 ms -- not kept.
  __device__ __forceinline__ static void load(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x, Regs& r) {
    const int M = EXACT ? MP : a.M;
    const float* pp = reinterpret_cast<const float*>(a.in[0]) + ro[0] + x * a.xstride[0];
    r.t = ld_stream(reinterpret_cast<const float*>(a.in[1]) + ro[1] + x * a.xstride[1]);
#pragma unroll
    for (int m = 0; m < MP; ++m) r.xm[m] = (EXACT || m < M) ? ld_stream(pp + (int64_t)m * a.mstride) : INFINITY;
  }

  __device__ __forceinline__ static void values(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                                double (&val)[NLANE]) {
    Regs r;
    load(a, ro, x, r);
    compute(a, r, val);
  }

  __device__ __forceinline__ static void compute(const S1Args& a, Regs& r, double (&val)[NLANE]) {
    const int M = EXACT ? MP : a.M;
    const double td = (double)r.t;
    float(&xm)[MP] = r.xm;

    if constexpr (ALGO == WBX_ENS_DIAG_LOADONLY) {  // diagnostic: the member loads with (almost) no arithmetic
      float s = 0.f;
#pragma unroll
      for (int m = 0; m < MP; ++m)
        if (EXACT || m < M) s += xm[m];
      val[0] = (double)s - td;
      val[1] = val[2] = val[3] = val[4] = 0.0;
      return;
    }
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
    } else if constexpr (ALGO == WBX_ENS_SORT) {
      {  // two members per instruction (v_pk_fma_f32): x * 0 + p is NaN iff x is NaN or +-inf
        typedef float pk2 __attribute__((ext_vector_type(2)));
        pk2 pz = {0.f, 0.f};
#pragma unroll
        for (int m = 0; m + 1 < MP; m += 2) {
          const pk2 x2 = {xm[m], (EXACT || m + 1 < M) ? xm[m + 1] : 0.f};
          if (EXACT || m < M) pz = x2 * (pk2){0.f, 0.f} + pz;
        }
        if ((MP & 1) && (EXACT || MP - 1 < M)) pz.x = fmaf(xm[MP - 1], 0.f, pz.x);
        poison = pz.x + pz.y;
      }
      // the bare instructions: fminf / fmaxf make the compiler canonicalise every loaded member first (one extra
      // v_max_f32 x, x each) to quiet signalling NaNs, which v_min / v_max in IEEE mode do themselves; NaN members
      // are caught by the probe above, not by the network
      auto mn = [](float u, float v) {
        float r;
        asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(u), "v"(v));
        return r;
      };
      auto mx = [](float u, float v) {
        float r;
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(u), "v"(v));
        return r;
      };
      if constexpr (WBX_ENS_SORTNET3) {
        // merge sort of compare-exchanges and 3-sorters (gen_sortnet3.py): v_min3 / v_med3 / v_max3 cost what v_min / v_max
        // do, so M = 51 sorts in 606 instructions instead of the 830 of Batcher's 415 comparators (32: 301 / 382, 64: 850 / 1086)
        SortNet3<MP>::sort(
            xm, mn, mx,
            [](float u, float v, float w) {
              float r;
              asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(u), "v"(v), "v"(w));
              return r;
            },
            [](float u, float v, float w) {
              float r;
              asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(u), "v"(v), "v"(w));
              return r;
            },
            [](float u, float v, float w) {
              float r;
              asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(u), "v"(v), "v"(w));
              return r;
            });
      } else {
        SortNet<MP>::sort(xm, mn, mx);
      }
    }

    // Everything is accumulated on e = x - shift (variance and spread are shift invariant).  shift = the target when it
    // is finite, so |x - t| is |e| itself (one fp64 add per member less than a separate x0 shift); with a NaN / infinite
    // target the shift falls back to the first register member, so that the member-only quantities (spread, variance)
    // stay finite -- the reference's spread / variance never look at the target -- and the target's NaN / inf reaches
    // the skill lanes through x0t below.
    const double x0 = (double)xm[0];
    const bool tfin = (td - td) == 0.0;
    const double shift = tfin ? td : x0;
    const double x0t = shift - td;  // 0 for a finite target, else NaN / -+inf
    double se = 0.0, sq = 0.0, sabs = 0.0, dot = 0.0;
#pragma unroll
    for (int m = 0; m < MP; ++m) {
      if (EXACT || m < M) {
        const double e = (double)xm[m] - shift;
        se += e;
        sq = fma(e, e, sq);
        sabs += fabs(e);
        if constexpr (ALGO == WBX_ENS_SORT) dot = fma((double)(2 * (m + 1) - M - 1), e, dot);
      }
    }
    sabs += fabs(x0t);  // mean |x - t| is NaN / inf with the target
    // M and `fair` are the same for every point of the launch: the three reciprocals are loop invariant (hoisted by the
    // compiler), and the five fp64 divisions per point (~10 instructions each) become multiplications -- within 1 ulp
    // of the divisions of the float64 restatement.
    const double dM = (double)M;
    const double fair = (a.flags & WBX_FLAG_FAIR) ? 1.0 : 0.0;
    const double inv_m = 1.0 / dM, inv_m1 = 1.0 / (dM - 1.0), spread_scale = 2.0 / (dM * (dM - fair));
    const double mean_e = se * inv_m;
    const double mean_d = x0t + mean_e;                       // mean_m p - t
    const double var = (sq - se * mean_e) * inv_m1;           // ddof = 1
    double spread;
    if constexpr (ALGO == WBX_ENS_SORT) {
      spread = dot * spread_scale;
    } else {
      spread = pair_total * spread_scale;
    }
    if constexpr (ALGO == WBX_ENS_DIAG_PAIRWISE_LDS) {
      static_assert(EXACT && MP % 17 == 0, "the LDS-tiled diagnostic is instantiated for M = 51");
      constexpr int B = 17, NB = MP / B;
      __shared__ float tile[MP][64];  // one-wave blocks (the launcher checks): 13 KB each, 12 blocks = 3 waves per SIMD on a CU
      float* col = &tile[0][threadIdx.x];
#pragma unroll
      for (int m = 0; m < MP; ++m) col[m * 64] = xm[m];
      asm volatile("" ::: "memory");  // the members are read back from the LDS, not forwarded from the registers
      double total = 0.0;
#pragma unroll
      for (int bi = 0; bi < NB; ++bi) {
        float xi[B];
#pragma unroll
        for (int k = 0; k < B; ++k) xi[k] = col[(bi * B + k) * 64];
#pragma unroll
        for (int k = 1; k < B; ++k) {
          float row = 0.f;
#pragma unroll
          for (int l = 0; l < k; ++l) row += fabsf(xi[k] - xi[l]);
          total += (double)row;
        }
#pragma unroll
        for (int bj = 0; bj < bi; ++bj) {
          float xj[B];
#pragma unroll
          for (int l = 0; l < B; ++l) xj[l] = col[(bj * B + l) * 64];
#pragma unroll
          for (int k = 0; k < B; ++k) {
            float row = 0.f;
#pragma unroll
            for (int l = 0; l < B; ++l) row += fabsf(xi[k] - xj[l]);
            total += (double)row;
          }
        }
      }
      spread = total * spread_scale;
    }
    val[0] = sabs * inv_m;
    val[1] = spread;
    val[2] = var;
    val[3] = mean_d * mean_d - var * inv_m;
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

// Mask / skipna handling of Aggregator.aggregate_stat_var (aggregation.py:339-357) around any ensemble core:
// masked-out or (skipna) NaN statistic values become 0 and are counted out through the paired count lanes.
template <class Core, bool SKIPNA>
struct EnsMasked {
  static constexpr int NIN = Core::NIN;
  static constexpr int NLANE = Core::NLANE;
  static constexpr int NACC = Core::NLANE + (SKIPNA ? Core::NLANE : 1);  // same count-lane convention as DetOp
  static constexpr int XR_UNROLL = 1, XK_UNROLL = 1, MIN_WAVES = WBX_ENS_MIN_WAVES;

  __device__ __forceinline__ static void values(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                                double (&val)[NLANE]) {
    Core::values(a, ro, x, val);
  }

  template <int V, bool XK>
  __device__ __forceinline__ static void accum(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x,
                                               double (&acc)[XK ? V : 1][NACC]) {
    static_assert(V == 1, "ensemble op is one point per lane");
    double val[NLANE];
    Core::values(a, ro, x, val);
    const bool valid =
        (a.flags & WBX_FLAG_MASKED) ? reinterpret_cast<const uint8_t*>(a.in[3])[ro[3] + x * a.xstride[3]] != 0 : true;
    if constexpr (!SKIPNA) {
#pragma unroll
      for (int l = 0; l < NLANE; ++l) acc[0][l] += valid ? val[l] : 0.0;
      acc[0][NLANE] += valid ? 1.0 : 0.0;
    } else {
#pragma unroll
      for (int l = 0; l < NLANE; ++l) {
        const bool ok = valid && !(val[l] != val[l]);
        acc[0][l] += ok ? val[l] : 0.0;
        acc[0][NLANE + l] += ok ? 1.0 : 0.0;
      }
    }
  }
};

template <class Op>
int launch_ens_op(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, bool map) {
  if (map) return launch_map<Op>(ctx, plan, a);
  if (plan->x_weights != nullptr) {  // latitude weights folded into stage 1, flat sweep over contiguous planes
    WBX_REQUIRE(!(plan->flags & WBX_FLAG_SKIPNA_ENS), "flat x-weighted mode does not take skipna_ensemble");
    if (plan->flags & WBX_FLAG_SKIPNA) return launch_flat_weighted1<EnsMasked<Op, true>>(ctx, plan, a);
    if (plan->flags & WBX_FLAG_MASKED) return launch_flat_weighted1<EnsMasked<Op, false>>(ctx, plan, a);
    return launch_flat_weighted1<Op>(ctx, plan, a);
  }
  if (plan->flags & WBX_FLAG_SKIPNA) return launch_partial<EnsMasked<Op, true>, 1>(ctx, plan, a);
  if (plan->flags & WBX_FLAG_MASKED) return launch_partial<EnsMasked<Op, false>, 1>(ctx, plan, a);
  return launch_partial<Op, 1>(ctx, plan, a);
}

template <int MP, bool EXACT>
int launch_ens_bucket(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, int algo, bool map) {
  if (algo == WBX_ENS_SORT) return launch_ens_op<EnsOpF32<MP, EXACT, WBX_ENS_SORT>>(ctx, plan, a, map);
  if (algo == WBX_ENS_DIAG_LOADONLY) {
    if (!EXACT || map) return fail(WBX_ERR_INVALID, "the load-only diagnostic exists for the exact-M partial kernels only");
    return launch_partial<EnsOpF32<MP, EXACT, WBX_ENS_DIAG_LOADONLY>, 1>(ctx, plan, a);
  }
  if (algo == WBX_ENS_DIAG_PAIRWISE_LDS) {
    if constexpr (EXACT && MP == 51) {
      if (map || plan->x_weights != nullptr || (plan->flags & (WBX_FLAG_MASKED | WBX_FLAG_SKIPNA)) || plan->block_threads != 64)
        return fail(WBX_ERR_INVALID, "the LDS-tiled pair-form diagnostic exists for the plain partial kernels with one-wave blocks only");
      return launch_partial<EnsOpF32<MP, EXACT, WBX_ENS_DIAG_PAIRWISE_LDS>, 1>(ctx, plan, a);
    } else {
      return fail(WBX_ERR_INVALID, "the LDS-tiled pair-form diagnostic exists for M = 51 only");
    }
  }
  return launch_ens_op<EnsOpF32<MP, EXACT, WBX_ENS_PAIRWISE>>(ctx, plan, a, map);
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
