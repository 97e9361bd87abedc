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
#include <cstdlib>

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

#ifndef WBX_ENS_NOFALLBACK
#define WBX_ENS_NOFALLBACK 0  // diagnostic builds only
#endif
#ifndef WBX_ENS_STATS32
#define WBX_ENS_STATS32 1  // 0: fp64 sums in the exact-M rank-form kernels too (A/B timing / exactness checks: make EXTRA=-DWBX_ENS_STATS32=0)
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
// not part of the ABI enum: what WBX_FLAG_SKIPNA_ENS runs for float32 ensembles of up to 64 members (r4).  skipna_ensemble
// (probabilistic.py:139-145, 206-216, 271-273, 303-336) makes the ensemble size a per-point count, which the generic operator
// below serves with an O(M^2) loop over members re-read from memory: 10.7 ms per 1.73 GB variable = 2 % of the HBM peak
// (profiles/r04_bench_n1.json, ensemble.skipna_ensemble).  Here the members sit in registers like in the rank-form kernels: a
// NaN member becomes +inf, the same sorting network moves the n valid members to the front, and the rank form runs over the
// first n of them with per-lane n (coefficients 2 i - (n - 1), divisions by n-derived numbers per point).  A point that holds
// an infinite member goes to the generic operator (EnsOpF32::finish), so +inf can stand for "missing".
constexpr int WBX_ENS_SKIPNA_SORT = 97;

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
    float poison;  // NaN iff a member is NaN / inf (set by compute())
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
    finish(a, ro, x, r, val);
  }

  // -> true: the point has to be redone by the generic fp64 op (see finish())
  // FAST32: the fp32 chain sums (stats32) -- the pipelined kernel only.  In s1_xr / s1_xk / s1_xf1 (no prefetch, 64-bit vector
  // addresses) they measured SLOWER than the fp64 sums (37-level field, same box: 1.57 against 1.40 ms on longitude-fastest,
  // 1.73 against 1.44 ms on latitude-fastest data: 162 instead of 133 VGPRs), while ens_pipe_kernel gains 9 % (1.39 -> 1.27 ms).
  template <bool FAST32 = false>
  __device__ __forceinline__ static bool compute(const S1Args& a, Regs& r, double (&val)[NLANE]) {
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
      return false;
    }
    double pair_total = 0.0;
    float poison = 0.f;  // NaN iff any member is NaN/inf (v_min/v_max would silently drop a NaN)
    int nvalid = 0;      // (WBX_ENS_SKIPNA_SORT) members that are not NaN
    bool weird = false;  // ... and whether one of them is infinite
    float shift32 = 0.f; // ... and what the missing ones are replaced by
    if constexpr (ALGO == WBX_ENS_SKIPNA_SORT) {
      // (r5, second step) A missing member becomes the SHIFT of the sums below -- the target, e = x - t = 0: it adds nothing to
      // any sum, so no sum has to select "the first n" of the sorted registers; it only sits in the sorted order, k = M - n of
      // them at the shift, and lifts the rank of every valid member above the shift by k: stats_skipna takes that out again
      // (sum (2 idx + 1) e = sum (2 rank + 1) e + 2 k sum max(e, 0), and sum max(e, 0) = (sum |e| + sum e) / 2).  Against
      // NaN -> +inf and a per-member `m < n` select: 51 compares + 51 selects less; infinite members are found from the sorted
      // extremes (exact sizes) or by one max3 fold over |x| (bucketed sizes) instead of a class test per member.
      const bool tfin32 = (r.t - r.t) == 0.f;
      shift32 = r.t;
      if (__builtin_amdgcn_ballot_w64(!tfin32)) {  // wave-uniform, rare: a NaN / infinite target -- the smallest valid member instead
        float lo = INFINITY;
#pragma unroll
        for (int m = 0; m < MP; ++m)
          if (EXACT || m < M) lo = fminf(lo, xm[m]);  // (fminf ignores NaN; no valid member: +inf, and every output is NaN anyway)
        shift32 = tfin32 ? r.t : lo;
      }
#pragma unroll
      for (int m = 0; m < MP; ++m) {
        if (EXACT || m < M) {
          const float x = xm[m];
          const bool missing = x != x;
          nvalid += missing ? 0 : 1;
          xm[m] = missing ? shift32 : x;
        }
      }
      if constexpr (!EXACT) {  // (bucketed sizes: the padding behind the M members is +inf, so the sorted extremes say nothing)
        float big = 0.f;  // max |x| over the members as they are now (no NaN among them unless the shift is one)
        int m = 0;
        for (; m + 1 < MP; m += 2) {
          const float a = m < M ? xm[m] : 0.f, b = m + 1 < M ? xm[m + 1] : 0.f;
          asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(big) : "v"(big), "v"(a), "v"(b));
        }
        if (m < MP) {
          const float a = m < M ? xm[m] : 0.f;
          asm("v_max3_f32 %0, %1, |%2|, |%2|" : "=v"(big) : "v"(big), "v"(a));
        }
        weird = big == INFINITY;
      }
    }
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
    } else if constexpr (ALGO == WBX_ENS_SORT || ALGO == WBX_ENS_SKIPNA_SORT) {
      if constexpr (ALGO == WBX_ENS_SORT) {  // two members per instruction (v_pk_fma_f32): x * 0 + p is NaN iff x is NaN or +-inf
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

    bool redo = false;
    if constexpr (ALGO == WBX_ENS_SKIPNA_SORT) {
      // (exact sizes: every register is a member or a copy of the shift, all finite unless a member is infinite -- and then the
      //  sorted extremes are: two compares instead of a fold over the members)
      if constexpr (EXACT) weird = xm[0] == -INFINITY || xm[MP - 1] == INFINITY;
      stats_skipna(a, xm, nvalid, td, shift32, val);
      r.poison = 0.f;
      return weird;  // per lane: the caller redoes such a point with the generic operator
    }
    if constexpr (FAST32 && EXACT && ALGO == WBX_ENS_SORT && WBX_ENS_STATS32) {
      // the hot instantiations (M = 50 / 51, rank form): fp32 chain sums on median-centred members (stats32), unless a lane
      // of the wave holds a point whose magnitudes could overflow / underflow an fp32 square or sum.  Then the CALLER redoes
      // the point with the generic fp64 op (members re-read from memory, a rolled loop: no registers of the hot path are
      // spent on the escape; a second set of unrolled fp64 sums here cost 20-40 VGPRs and a wave per SIMD).  Wave-uniform;
      // never taken on physical fields.  The comparisons are false for NaN: a NaN target goes the same way.
      const float range = xm[MP - 1] - xm[0];
      const float big = fmaxf(fmaxf(fabsf(xm[0]), fabsf(xm[MP - 1])), fabsf(r.t));
      const bool fast_ok = (range == 0.f || (range >= 0x1p-50f && range <= 0x1p60f)) && big <= 0x1p100f;
      redo = !WBX_ENS_NOFALLBACK && __builtin_amdgcn_ballot_w64(!fast_ok) != 0;
      if (!redo) stats32(a, xm, r.t, poison, val);
      r.poison = poison;
    } else {
      stats64(a, xm, td, pair_total, val);
      r.poison = poison;
      apply_poison(r, val);
    }
    return redo;
  }

  __device__ __forceinline__ static void apply_poison(const Regs& r, double (&val)[NLANE]) {
    if constexpr (ALGO == WBX_ENS_SORT) {
      if (r.poison != r.poison) {  // reference: a NaN member makes every ensemble statistic NaN
#pragma unroll
        for (int l = 0; l < NLANE; ++l) val[l] = (double)r.poison;
      }
    }
  }

  // compute() + the escape, for callers that know where the point lives
  template <bool FAST32 = false>
  __device__ __forceinline__ static void finish(const S1Args& a, const int64_t (&ro)[WBX_MAX_INPUTS], int64_t x, Regs& r,
                                                double (&val)[NLANE]) {
    if (compute<FAST32>(a, r, val)) {
      EnsOpGeneric<float>::values(a, ro, x, val);
      apply_poison(r, val);
    }
  }

  // The rank form over the n valid members of a point, n per lane; the formulas of EnsOpGeneric::values on e = x - shift.
  // sum_{i<j} |x_i - x_j| = sum_i (2 i - (n - 1)) x_(i): the coefficients add up to 0 over i < n, so the shift drops out.
  // (r5) The registers hold the n valid members AND k = M - n copies of the shift itself (compute(): what the missing members
  // were replaced by) in sorted order: e = 0 for the copies, so every sum runs over all registers with compile-time
  // coefficients; a valid member above the shift sits k places too high, which  sum (2 idx + 1) e  pays back as
  // 2 k sum max(e, 0) = k (sum |e| + sum e).  1 / n and 1 / (n - 1) come from v_rcp_f64 + two Newton steps (n is a small
  // integer: no scaling, no fix-up) instead of five full fp64 divisions per point (~35 instructions each).
  __device__ __forceinline__ static double recip_small(double n) {
    double r = __builtin_amdgcn_rcp(n);
    r = fma(fma(-n, r, 1.0), r, r);
    r = fma(fma(-n, r, 1.0), r, r);
    return r;
  }
  __device__ __forceinline__ static void stats_skipna(const S1Args& a, const float (&xm)[MP], const int n, const double td,
                                                      const float shift32, double (&val)[NLANE]) {
    const int M = EXACT ? MP : a.M;
    const double shift = (double)shift32;
    const double x0t = shift - td;  // 0 for a finite target, else NaN / -+inf
    const double dM = (double)n, dK = (double)(M - n);
    double se = 0.0, sq = 0.0, sabs = 0.0, dot1 = 0.0;
#pragma unroll
    for (int m = 0; m < MP; ++m) {
      if (EXACT || m < M) {
        const double e = (double)xm[m] - shift;
        se += e;
        sq = fma(e, e, sq);
        sabs += fabs(e);
        dot1 = fma((double)(2 * m + 1), e, dot1);
      }
    }
    const double dot = fma(-dM, se, fma(-dK, sabs + se, dot1));
    sabs += n > 0 ? fabs(x0t) : 0.0;  // mean |x - t| is NaN / inf with the target
    const double nan = __builtin_nan("");
    const double inv_n = n > 0 ? recip_small(dM) : nan;          // (0 / 0 in the reference's means: NaN)
    const double inv_n1 = n > 1 ? recip_small(dM - 1.0) : nan;   // ddof = 1 / the fair divisor of a single member: NaN
    const double inv_pairs = (a.flags & WBX_FLAG_FAIR) ? inv_n * inv_n1 : inv_n * inv_n;
    const double mean_e = se * inv_n;
    const double mean_d = x0t + mean_e;
    const double var = (sq - se * mean_e) * inv_n1;
    val[0] = sabs * inv_n;
    val[1] = 2.0 * dot * inv_pairs;
    val[2] = var;
    val[3] = mean_d * mean_d - var * inv_n;
    val[4] = mean_d * mean_d;
  }

  __device__ __forceinline__ static void stats64(const S1Args& a, float (&xm)[MP], const double td, const double pair_total,
                                                 double (&val)[NLANE]) {
    const int M = EXACT ? MP : a.M;
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
  }

  // The fp32 chain sums of the exact-M rank form (M = MP known at compile time, xm ascending).
  //
  // Per member the fp64 sums above cost six fp64-class instructions (convert, x - t, and the four accumulations) at 2.1 ns
  // per wave instruction; plain fp32 adds / FMAs issue at 1.18 ns (tools/ubench/valu_rates).  Here everything per member is
  // fp32, arranged so that no sum cancels and every chain is short:
  //   * centre c = the sorted median x_(MP/2).  e_i = fl(x_i - c): one rounding, |error| <= u |e_i|, u = 2^-24, and exact
  //     whenever x_i and c are within a factor two of each other (Sterbenz): temperatures, geopotential, pressure ...
  //   * spread:  sum_i (2i - M - 1) x_(i) = sum_{i < M/2} (M - 1 - 2i) (x_(M-1-i) - x_(i)): every term is a non-negative
  //     difference times a positive integer <= 50, so the relative error of the sum is bounded by the chain length: no term
  //     can cancel another.
  //   * skill:   sum_i |x_i - t|, terms non-negative.
  //   * sum e^2: terms non-negative.  With the median as centre sum e^2 / M <= 2 var_pop (|mean - median| <= sigma), so
  //     var = (sum e^2 - (sum e)^2 / M) / (M - 1) loses at most a factor ~2-3 to the subtraction, whatever the bias to the
  //     target is (the fp64 path's shift is the target: exact in fp64, hopeless in fp32 when |bias| >> spread).
  //   * sum e:   accumulated as pairs e_i + e_(M-1-i), which have opposite signs around the median: partial sums stay small.
  // K = 8 interleaved chains per sum (<= 8 terms each), chains added in pairs in fp32, the four pair sums widened and added
  // in fp64; the rest of the point (mean, variance, squares) is fp64 as before.  Worst-case relative error of each
  // non-negative sum: (1 rounding of the term + 7 chain adds + 1 pair add) u = 9 u = 5.4e-7; typical (random rounding) ~2 u;
  // measured against the float64 oracle in tests/test_metrics.py and tests/test_gpu_round3.py.  The north_star tolerance
  // is 1e-6 on the aggregated value (a weighted mean of >= 10^3 such points, whose errors do not add coherently).
  // `poison` (0, or NaN when a member is NaN / inf) enters the first chain of every sum: all five values turn NaN with it.
  __device__ __forceinline__ static void stats32(const S1Args& a, float (&xm)[MP], const float t, const float poison,
                                                 double (&val)[NLANE]) {
    static_assert(EXACT, "fp32 chain sums are instantiated for the exact-M buckets");
    constexpr int K = 8, MID = MP / 2, NPAIR = MP / 2;
    const float c = xm[MID];
    float se[K], sq[K], sa[K], dt[K];
#pragma unroll
    for (int k = 0; k < K; ++k) se[k] = sq[k] = sa[k] = dt[k] = k == 0 ? poison : 0.f;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      const int j = MP - 1 - i, k = i % K;
      const float ei = xm[i] - c, ej = xm[j] - c;
      se[k] += ei + ej;
      sq[k] = fmaf(ei, ei, sq[k]);
      sq[k] = fmaf(ej, ej, sq[k]);
      sa[k] += fabsf(xm[i] - t);
      sa[k] += fabsf(xm[j] - t);
      dt[k] = fmaf((float)(MP - 1 - 2 * i), xm[j] - xm[i], dt[k]);
    }
    if constexpr (MP & 1) {
      sa[NPAIR % K] += fabsf(c - t);  // the median itself: e = 0, rank weight 0
    } else {
      // even M: x_(MID) is the upper median and already went through the loop as j = MID of the pair i = MID - 1
    }
    auto fold = [](const float (&v)[K]) {
      return ((double)(v[0] + v[4]) + (double)(v[1] + v[5])) + ((double)(v[2] + v[6]) + (double)(v[3] + v[7]));
    };
    const double dse = fold(se), dsq = fold(sq), dsa = fold(sa), ddt = fold(dt);
    constexpr double dM = (double)MP;
    const double fair = (a.flags & WBX_FLAG_FAIR) ? 1.0 : 0.0;
    constexpr double inv_m = 1.0 / dM, inv_m1 = 1.0 / (dM - 1.0);
    const double spread_scale = 2.0 / (dM * (dM - fair));
    const double mean_e = dse * inv_m;
    const double mean_d = ((double)c - (double)t) + mean_e;  // mean_m p - t
    const double var = (dsq - dse * mean_e) * inv_m1;         // ddof = 1
    val[0] = dsa * inv_m;
    val[1] = ddt * spread_scale;
    val[2] = var;
    val[3] = mean_d * mean_d - var * inv_m;
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


// ---------------------------------------------------------------------------------------------
// The x-summed sweep of the unmasked fp32 ensemble ops with the NEXT tile's members in flight while this one is reduced.
//
// s1_xr_kernel reduces a 64-point tile as  [52 loads] -> wait -> [~1100 VALU instructions]: a wave's own loads and arithmetic
// never overlap, and three waves per SIMD (133 VGPRs) cover only part of the ~5 us a tile's loads take at this bandwidth
// (load-only 0.256 ms, arithmetic ~0.25 ms, kernel 0.31 ms per 1.73 GB: the two add up to 80 % instead of overlapping).
// Register double-buffering needs 255 VGPRs (tried in round 1: slower).  Here the next tile goes through the LDS instead:
//   * `global_load_lds_dword` (LDS-DMA) writes a member's 64 dwords straight into the wave's staging buffer
//     stage[member][lane] -- no VGPR is held while the load is in flight;
//   * the address is  SGPR base (row + member * stride, scalar adds) + one shared 32-bit VGPR offset (x):  no 64-bit vector
//     address arithmetic at all (the 68 v_lshl_add_u64 per tile of the register version);
//   * per tile: wait vmcnt(0) -> 50 members from the LDS into VGPRs (ds_read2st64_b32, bank = lane) -> wait lgkmcnt(0) ->
//     issue the next tile's 50 LDS-DMA loads -> sort + sums of this tile.
// LDS is allocated in 1280-byte granules on gfx950: 50 members x 256 B = 12 800 B = exactly ten, so twelve one-wave blocks
// (3 per SIMD) fit the 160 KB of a CU; the 51st member and the target ride in two VGPRs loaded one tile ahead.
// One wave per block; rows (depth) and x tiles of the block's (key, chunk) form one tile sequence.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // m0 is written by the LDS-DMA statements and listed as clobbered
// FLAT: the same sweep over contiguous latitude-fastest planes with the latitude weights folded in (s1_xf1_kernel's
// geometry, plan->x_weights): a chunk's rows are one run [e0, e1) of the plane, walked in 64-element tiles that start on a
// 64-element boundary of the PLANE (whole cache lines; the lanes of the first tile in front of e0 are dropped, never moved
// to another tile: a wave whose lanes sit in two 256-byte pieces fetches three or four lines per load).  A lane's weight
// w[e mod nx] comes from the fp64 table in global memory (5.7 KB, cache resident), fetched one tile ahead like the target.
// (A first version of this flavour lost to s1_xf1_kernel in round 3 -- 1.48-1.52 against 1.44 ms for the 37-level field --
// because both started their lanes one trip ahead instead of dropping them: 12 % more line requests, see wbx_s1.hpp.)
#ifndef WBX_ENS_PIPE_NLDS
#define WBX_ENS_PIPE_NLDS 50   // members staged through the LDS (x 256 B per one-wave block)
#endif
#ifndef WBX_ENS_PIPE_WAVES
#define WBX_ENS_PIPE_WAVES 3   // waves per SIMD the register budget is cut for (12 800-byte blocks: 12 per CU)
#endif
// (the skipna_ensemble flavour -- per-lane member counts, fp64 sums over the valid members -- needed 209 registers = two waves per
//  SIMD while its sums selected and weighted every member in fp64 by the per-lane count; with stats_skipna's compile-time
//  coefficients it is 151 registers and runs three: 0.423 -> 0.378 ms on the 1.73 GB variable, 51 -> 57 % of the HBM peak; with the
//  missing members replaced by the shift itself -- no per-member select, no class tests -- 113 registers, 0.336 ms = 64 %)
#ifndef WBX_ENS_PIPE_SKIPNA_WAVES
#define WBX_ENS_PIPE_SKIPNA_WAVES 3
#endif
template <int MP, bool EXACT, int ALGO, bool FLAT>
__global__ void __launch_bounds__(64, ALGO == WBX_ENS_SKIPNA_SORT ? WBX_ENS_PIPE_SKIPNA_WAVES : WBX_ENS_PIPE_WAVES)
ens_pipe_kernel(S1Args a, int R) {
  using Op = EnsOpF32<MP, EXACT, ALGO>;
  constexpr int NA = Op::NACC;
  constexpr int NLDS = MP < WBX_ENS_PIPE_NLDS ? MP : WBX_ENS_PIPE_NLDS;  // members staged through the LDS
  constexpr int NREG = MP - NLDS;          // members prefetched into VGPRs
  __shared__ float stage[NLDS][64];
  const int lane = threadIdx.x;
  const int M = EXACT ? MP : a.M;
  const int64_t b = blockIdx.x;
  const int64_t key = b / a.nchunk;
  const int chunk = (int)(b - key * a.nchunk);
  const int64_t d0 = (int64_t)chunk * a.dchunk;
  const int64_t d1 = d0 + a.dchunk < a.D ? d0 + a.dchunk : a.D;
  const int nx = (int)a.nx;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&stage[0][0];
  const int64_t mstride_b = a.mstride * 4;

  int64_t kb[WBX_MAX_INPUTS], ro[WBX_MAX_INPUTS];
  key_bases<2>(a, key, kb);
  double acc[NA];
#pragma unroll
  for (int l = 0; l < NA; ++l) acc[l] = 0.0;

  // the current segment: elements [e0, e1) of the depth row (FLAT: of the plane) at `ro`, tiles of 64 from et; d = the next
  // depth row
  int64_t d = d0, e0 = 0, e1 = 0, et = 0;
  int wm = 0;  // FLAT: this lane's weight index in the tile at et
  const int wstep = 64 % nx;
  auto open_segment = [&]() {  // d < d1
    if constexpr (FLAT) {
      const int64_t plane = d / R;
      const int64_t j0 = d - plane * R;
      const int64_t nj = d1 - d < R - j0 ? d1 - d : R - j0;
      row_bases<2>(a, kb, key, plane * R, ro);
      e0 = j0 * nx;
      e1 = (j0 + nj) * nx;
      et = e0 & ~(int64_t)63;
      wm = (int)((et + lane) % nx);
      d += nj;
    } else {
      row_bases<2>(a, kb, key, d, ro);
      e0 = 0;
      e1 = nx;
      et = 0;
      d += 1;
    }
  };

  float xn[NREG > 0 ? NREG : 1], tn = 0.f;
  double wn = 0.0;
  auto issue = [&]() {  // the tile at et of the open segment
    int64_t e = et + lane;
    e = e < e0 ? e0 : (e < e1 ? e : e1 - 1);  // lanes outside the segment re-read its edge (counted out below): no EXEC games
    const uint32_t voff = (uint32_t)e * (uint32_t)a.xstride[0] * 4u;
    const char* um = uniform_ptr(reinterpret_cast<const char*>(a.in[0]) + ro[0] * 4);
#pragma unroll
    for (int m = 0; m < NLDS; ++m) {
      if (EXACT || m < M)
        asm volatile("s_add_u32 m0, %2, %3\n\tglobal_load_lds_dword %0, %1 nt" ::"v"(voff), "s"(um), "s"(lds0), "i"(m * 256)
                     : "memory", "scc", "m0");
      um += mstride_b;
      asm volatile("" : "+s"(um));  // one s_add_u32 / s_addc_u32 per member, not a table of 50 hoisted products
    }
    const float* pr = reinterpret_cast<const float*>(a.in[0]) + ro[0] + e * a.xstride[0];
#pragma unroll
    for (int m = NLDS; m < MP; ++m) xn[m - NLDS] = (EXACT || m < M) ? ld_stream(pr + (int64_t)m * a.mstride) : INFINITY;
    tn = ld_stream(reinterpret_cast<const float*>(a.in[1]) + ro[1] + e * a.xstride[1]);
    if constexpr (FLAT) wn = a.xw[wm];
  };

  bool more = d < d1;
  if (more) {
    open_segment();
    issue();
  }
  while (more) {
    typename Op::Regs r;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int m = 0; m < NLDS; ++m) r.xm[m] = (EXACT || m < M) ? stage[m][lane] : INFINITY;
#pragma unroll
    for (int m = NLDS; m < MP; ++m) r.xm[m] = xn[m - NLDS];
    r.t = tn;
    const double wcur = wn;
    const int64_t ecur = et + lane;
    const bool valid = ecur >= e0 && ecur < e1;
    const int64_t xcur = ecur < e0 ? e0 : (ecur < e1 ? ecur : e1 - 1);
    int64_t rocur[WBX_MAX_INPUTS];
#pragma unroll
    for (int i = 0; i < WBX_MAX_INPUTS; ++i) rocur[i] = ro[i];
    const bool full = et >= e0 && et + 64 <= e1;  // every lane of this tile holds a point of the segment (wave-uniform)
    et += 64;
    if (et >= e1) {
      more = d < d1;
      if (more) open_segment();
    } else if constexpr (FLAT) {
      wm += wstep;
      wm = wm >= nx ? wm - nx : wm;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile has left the staging buffer
    if (more) issue();
    double val[Op::NLANE];
    Op::template finish<true>(a, rocur, xcur, r, val);
    if constexpr (FLAT) {
      if (full) {
#pragma unroll
        for (int l = 0; l < NA; ++l) acc[l] = fma(val[l], wcur, acc[l]);
      } else {
#pragma unroll
        for (int l = 0; l < NA; ++l) acc[l] = fma(valid ? val[l] : 0.0, wcur, acc[l]);
      }
    } else if (full) {
#pragma unroll
      for (int l = 0; l < NA; ++l) acc[l] += val[l];
    } else {
#pragma unroll
      for (int l = 0; l < NA; ++l) acc[l] += valid ? val[l] : 0.0;
    }
  }
#pragma unroll
  for (int l = 0; l < NA; ++l) {
    const double v = wave_sum(acc[l]);
    if (lane == l) a.out[(key * a.nchunk + chunk) * NA + l] = v;
  }
}

#pragma clang diagnostic pop

// Eligibility of the pipelined sweep: plain (no mask / skipna wrappers), x summed without folded weights, one-wave blocks,
// offsets inside a row within 32 bits.
// `skipna_ens`: the per-point member counts of skipna_ensemble (EnsOpF32<.., SKIPNA_SORT>) ride the same sweep (r5).
inline bool ens_pipe_ok(const wbx_s1_plan* plan, const S1Args& a, bool skipna_ens = false) {
  static const bool off = getenv("WBX_ENS_PIPE") && atoi(getenv("WBX_ENS_PIPE")) == 0;  // A/B against s1_xr_kernel
  static const bool off_skipna = getenv("WBX_ENS_PIPE_SKIPNA") && atoi(getenv("WBX_ENS_PIPE_SKIPNA")) == 0;
  if (off || (skipna_ens && off_skipna)) return false;
  if (plan->x_kept || plan->x_weights != nullptr || plan->block_threads != 64) return false;
  if (plan->flags & (WBX_FLAG_MASKED | WBX_FLAG_SKIPNA | (skipna_ens ? 0u : (unsigned)WBX_FLAG_SKIPNA_ENS))) return false;
  if (plan->nx <= 0 || plan->ndepth <= 0 || plan->nkey <= 0) return false;
  if (a.xstride[0] < 0 || a.xstride[1] < 0) return false;
  return (double)plan->nx * (double)a.xstride[0] * 4.0 < 4294967296.0;
}

// ... and of its FLAT flavour: latitude weights folded into stage 1 over whole contiguous planes (what launch_flat_weighted1
// asks for), one-wave blocks (the planner's 'point64' geometry), offsets inside a plane within 32 bits.
inline bool ens_pipe_flat_ok(const wbx_s1_plan* plan, const S1Args& a) {
  static const bool off = getenv("WBX_ENS_PIPE") && atoi(getenv("WBX_ENS_PIPE")) == 0;
  if (off) return false;
  if (plan->x_kept || plan->x_weights == nullptr || plan->block_threads != 64 || plan->plane_rows <= 0) return false;
  if (plan->flags & (WBX_FLAG_MASKED | WBX_FLAG_SKIPNA | WBX_FLAG_SKIPNA_ENS)) return false;
  if (plan->nx <= 0 || plan->nx > WBX_XW_MAX || plan->ndepth <= 0 || plan->nkey <= 0 || plan->ndepth % plan->plane_rows != 0) return false;
  if (a.xstride[0] != 1 || a.xstride[1] != 1) return false;
  return (double)plan->nx * (double)plan->plane_rows * 4.0 < 4294967296.0;
}

template <int MP, bool EXACT, int ALGO, bool FLAT = false>
int launch_ens_pipe(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a) {
  const int64_t grid = plan->nkey * plan->nchunk;
  WBX_REQUIRE(grid < (int64_t)1 << 31, "grid too large (%lld blocks)", (long long)grid);
  hipLaunchKernelGGL((ens_pipe_kernel<MP, EXACT, ALGO, FLAT>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, a,
                     FLAT ? plan->plane_rows : 0);
  WBX_HIP(hipGetLastError());
  return 0;
}

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
  // (the pair form stays on s1_xr_kernel: pipelined it takes 168 VGPRs + scratch, and it is a diagnostic since round 3)
  if (!map && algo == WBX_ENS_SORT && ens_pipe_ok(plan, a)) return launch_ens_pipe<MP, EXACT, WBX_ENS_SORT>(ctx, plan, a);
  if (!map && algo == WBX_ENS_SORT && ens_pipe_flat_ok(plan, a)) return launch_ens_pipe<MP, EXACT, WBX_ENS_SORT, true>(ctx, plan, a);
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
  if (!map && algo == WBX_ENS_SKIPNA_SORT && ens_pipe_ok(plan, a, true)) return launch_ens_pipe<MP, EXACT, WBX_ENS_SKIPNA_SORT>(ctx, plan, a);
  if (algo == WBX_ENS_SKIPNA_SORT) return launch_ens_op<EnsOpF32<MP, EXACT, WBX_ENS_SKIPNA_SORT>>(ctx, plan, a, map);
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

// Measured and not kept (r4), EnsOpF32::stats32: the two members of a pair (x_(i), x_(M-1-i)) in one register pair -- x - c, the
// sum of e and the sum of e^2 as v_pk_add_f32 / v_pk_fma_f32 -- is 83 vector instructions less per point (1320 -> 1237 in
// ens_pipe_kernel<51>, no extra moves) and 1 % SLOWER on the same box (north_star kernel 1.244 -> 1.258 ms, three alternating
// runs of bench.py): a packed fp32 instruction takes two issue passes here, so nothing is saved and the chains get longer.
