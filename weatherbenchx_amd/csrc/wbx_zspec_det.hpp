// Zonal spectra of predictions AND targets plus the deterministic lanes of the same (p, t[, c]) rows in ONE sweep.
//
// configs[3] / configs[4] ask for "spectra of p and t + the full deterministic suite in the same sweep" (SURVEY 8d).  As
// separate launches the z fields cross the HBM twice: wbx_det_partial reads p, t, c (12 B/point), the two spectrum launches
// read p and t again (8 B/point).  zspec1440_kernel already transforms TWO rows per wave (every quantity is a (row A, row B)
// pair of packed fp32): here the pair is (p[row], t[row]) instead of two consecutive rows of one field, so
//   * the 48 floats a lane holds of the pair in front of pass 1 are exactly the p and t values of its 24 points: with the
//     climatology row fetched alongside (one more 8-byte load per 2 points) the DET6 lanes e, |e|, e^2, (p-c)^2, (t-c)^2,
//     (p-c)(t-c) are accumulated there in fp64 -- the same arithmetic as DetOp (wbx_det.hip) -- folded over the wave with DPP
//     row operations and written as the row's entry of the stage-1 partial buffer (one key = one row: stage 2 is unchanged);
//   * the spectra of row A go to the predictions' accumulators, those of row B to the targets' (z14_pair<.., PT = true>).
// Algorithmic bytes: 12 B/point (8 without a climatology) for BOTH families.
// Registers: +24 (climatology prefetch) +24 (second set of spectrum sums) +12 (deterministic sums) on a kernel that sat at
// 168: eight one-wave teams per block = two waves per SIMD and up to 256 VGPRs (the spectrum kernel's throughput is flat
// from 8 to 12 teams per CU: 0.311 vs 0.291 ms per field).  What it took to fit (round 3, configs[4] chunk
// = 533 540 rows, `tools/kbench_det_spectrum.py`; the three separate launches: 2.59 ms):
//   * first version: 287 VGPRs needed, 31 spilled -- among them ten of the 24 PREFETCH loads, each stored to scratch behind a
//     full `s_waitcnt vmcnt(0)`: 3.75 ms.  The six deterministic sums are serial fma chains over a lane's 24 points, so the
//     compiler widened all 72 inputs and formed all 48 anomalies up front (~100 fp64 temporaries) to feed the chains;
//     scheduling barriers do not help (the hoisting happens on the IR).  An opaque asm redefinition of a point pair's inputs
//     AND of the running sums makes pair i start when pair i - 1 is done: 214 VGPRs, no scratch, 1.87 ms = 4.95 TB/s = 62 % of
//     the HBM peak on 12 B/point.
//   * the climatology row staged through the LDS instead of 24 registers (24 `global_load_lds_dword` per row into a 6 KB slot
//     per team, read back with ds_read2st64_b32; WBX_ZD_C_IN_REGISTERS = 0): 192 VGPRs, 1.98 ms -- twice the vector-memory
//     instructions for the same bytes; kept as the A/B build (`make ab-zdlds`).
//   * without a climatology (DET3): 190 VGPRs, 1.44-1.51 ms (the two spectra alone: 1.22-1.26 ms).
//   * (r6) where the time went next: the kernel's time follows its vector-ALU time almost one to one at two waves per SIMD
//     (knock-out 16: 144 fp64 instructions fewer per row = -10.4 %), and every row began with four dependent VECTOR-load round
//     trips for its three base pointers.  Scalar lookups one row ahead + global instead of flat loads: 1.83 -> 1.68 ms (folded,
//     5.0 -> 5.5 TB/s), 232 VGPRs (profiles/r06_det_spectrum_scalar_lookup_ab.txt).
//
// Rows = the keys of the deterministic plan (wbx_s1_plan with x = longitude summed, nx = 1440, unit x strides, ndepth = 1,
// nchunk = 1), addressed through the plan's own offset / gather tables; `group` / `scale` are per key.
// Included by wbx_spectrum.hip (inside namespace wbx, after wbx_zspec1440.hpp).
#pragma once

#ifndef WBX_ZD_TEAMS
#define WBX_ZD_TEAMS 8  // one-wave teams per block = per CU (the block's LDS fills it): 8 = two waves per SIMD at 214 VGPRs.  12 =
                        // three waves per SIMD needs <= 168 VGPRs, which the kernel only fits with WBX_ZD_FETCH_AT = 5 (161, no
                        // scratch): make ab-zd12f5, measured in round 6 and NOT adopted (below)
#endif
#ifndef WBX_ZD_KNOCK
#define WBX_ZD_KNOCK 0  // timing diagnostics (WRONG results; make ab-zdk1 ...): 1 = every row re-reads the team's FIRST row (cache
                        // hits, no HBM stream), 2 = no deterministic lanes, 4 = no loads after the first row, 8 = no wave sums / stores of the deterministic lanes, 16 = the deterministic lanes over every other point pair
#endif
#ifndef WBX_ZD_FETCH_AT
#define WBX_ZD_FETCH_AT 0  // 0: the next row's p (+ c) behind pass 1's stores, t behind pass 2's (72 registers live across the
                           // transform's peak); 4 / 5: all of it behind the mirror exchange / at the end of the pair -- the rows'
                           // registers are then live only across the unpack, and the latency is left to the other waves of the SIMD.
                           // (r6, profiles/r06_det_spectrum_3waves_ab.txt) 12 teams + 5 against 8 teams + 0, same box, the kernel
                           // alone on a configs[4] chunk: DET6 1.96-1.98 against 2.01-2.03 ms (-1 ... -2 %), DET3 -3 %; but in the
                           // configs[4] chunk loop 2.92 against 2.85 ms per chunk: a 155 KB block owns its CU's LDS, and the
                           // ensemble kernel of the other stream no longer fits beside it.  Not adopted.
#endif
constexpr int ZD_TEAMS = WBX_ZD_TEAMS;
#ifndef WBX_ZD_F32_CHAINS
#define WBX_ZD_F32_CHAINS 0  // 1: the deterministic lanes' per-point statistics in fp32, sums of <= 8 non-negative terms as fp32
                             // chains (A/B: make ab-zdf32).  Measured in round 6 (tools/gpu_r6_kernels_a.sh, same box, configs[4]
                             // chunk): 1.976 / 1.994 against 2.009 / 2.012 ms -- 168 of the row's 433 fp64-rate instructions become
                             // fp32 ones and the kernel gains < 1 %: its vector ALU is not what bounds it.  Not adopted (the fp64
                             // lanes equal wbx_det_partial's bit for bit, tests/test_gpu_round3.py).
#endif
#ifndef WBX_ZD_FLAT_LOADS
#define WBX_ZD_FLAT_LOADS 0  // 1: the rows through generic pointers = flat_load, what the kernel did up to round 6 (A/B: make ab-zdflat)
#endif
#ifndef WBX_ZD_PRIO
#define WBX_ZD_PRIO 1  // the two waves of a SIMD alternate their user priority row by row (0: A/B, make ab-zdnoprio)
#endif
#ifndef WBX_ZD_TW_EARLY
#define WBX_ZD_TW_EARLY 1  // z14_pair<.., TW_EARLY>: a stage's LDS reads grouped ahead of the arithmetic before their use (-1 %; 0: A/B, make ab-zdtwlate)
#endif
#ifndef WBX_ZD_C_IN_REGISTERS
#define WBX_ZD_C_IN_REGISTERS 1  // 0: the climatology row staged through the LDS (24 LDS-DMA dwords per row) instead of 24 VGPRs (A/B: make ab-zdlds)
#endif

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // m0 is written by the LDS-DMA statements and listed as clobbered
// FOLD (r6, wbx_det_spectrum_folded): the rows' deterministic sums are not stored row by row for stage 2 -- each is multiplied by
// its row's weight `dscale[row]` (what stage 2's W holds for the row: the latitude weight) and added to the team's running sums
// PER LANE; the six wave sums and stores happen once per RECORD instead of once per row (6 % of the kernel,
// profiles/r06_det_spectrum_knockouts.txt), and the record carries them behind the two spectra (ZD_TAIL values: spec_close_kernel
// adds them over the records of a group in key order like every wavenumber).
constexpr int ZD_TAIL = 8;  // doubles behind the 2 x 721 spectrum values of a folded record (6 used)
template <bool HAS_C, bool FOLD = false>
__global__ void __launch_bounds__(64 * ZD_TEAMS) zspec1440_det_kernel(S1Args a, int64_t nrows, int rows_per_team,
                                                                      const float2* __restrict__ tables_g,
                                                                      const int32_t* __restrict__ group,
                                                                      const double* __restrict__ scale, SpecRecs recs,
                                                                      const double* __restrict__ dscale = nullptr) {
  constexpr int NA = HAS_C ? 6 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float2* const tw1 = reinterpret_cast<float2*>(lds_raw);
  float2* const tw2 = tw1 + Z14_TW1;
  float2* const twr = tw2 + Z14_TW2;
  const int lane = (int)(threadIdx.x & 63);
  const int team = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nteam = (int)(blockDim.x >> 6);
  v4* const buf = reinterpret_cast<v4*>(twr + Z14_TWR) + team * Z14_BUF;
  // the climatology staging slots behind the teams' transform buffers: [team][24 dwords of a lane][64 lanes]
  float* const cbuf = reinterpret_cast<float*>(reinterpret_cast<v4*>(twr + Z14_TWR) + nteam * Z14_BUF) + team * (24 * 64);
  const uint32_t cbuf_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)cbuf;
  for (int i = threadIdx.x; i < Z14_TABLES; i += blockDim.x) tw1[i] = tables_g[i];
  __syncthreads();
  const int64_t r0 = ((int64_t)blockIdx.x * nteam + team) * rows_per_team;
  const int64_t r1 = r0 + rows_per_team < nrows ? r0 + rows_per_team : nrows;
  const int64_t team_id = (int64_t)blockIdx.x * nteam + team;
  if (r0 >= r1) {  // only wave-level ordering below
    spec_rec_static(recs, (unsigned int)team_id, -1, 0ull, lane == 0);  // (the team's reserved slot stays empty)
    return;
  }
  constexpr int nk = Z14_N2 + 1;
  const Z14Lane c = z14_lane(lane, buf, tw2);
  const int L = c.L;
  const double quarter_inv_nn = 0.25 / ((double)Z14_N * (double)Z14_N);  // E and O are used without their factor 1/2

  double accp[6], accmp[6], acct[6], accmt[6];  // sums of k = L + 60 s and of 720 - k, predictions / targets
#pragma unroll
  for (int s = 0; s < 6; ++s) accp[s] = accmp[s] = acct[s] = accmt[s] = 0.0;
  double dsum[FOLD ? NA : 1];  // FOLD: this lane's weighted deterministic sums over the rows of the current record
#pragma unroll
  for (int l = 0; l < (FOLD ? NA : 1); ++l) dsum[l] = 0.0;
  int32_t cur = group[r0];
  unsigned int seq = 0;
  // (r5) one record of 2 x 721 values: the predictions' sums, then the targets'; the team's LAST record sits in its reserved slot
  auto flush = [&](int32_t next, bool last = false) {
    double* const rec = last ? spec_rec_static(recs, (unsigned int)team_id, cur, spec_key(team_id, 0xffffffu), lane == 0)
                             : spec_rec_open(recs, cur, spec_key(team_id, seq++), lane);
    z14_send<true>(rec, c, accp, accmp);
    z14_send<true>(rec + nk, c, acct, accmt);
#pragma unroll
    for (int s = 0; s < 6; ++s) accp[s] = accmp[s] = acct[s] = accmt[s] = 0.0;
    if constexpr (FOLD) {
#pragma unroll
      for (int l = 0; l < NA; ++l) {
        const double tot = wave_sum_uniform(lane < Z14_LANES ? dsum[l] : 0.0);  // (lanes 60..63 shadow lane 59)
        if (lane == 0) rec[2 * nk + l] = tot;
        dsum[l] = 0.0;
      }
      if (lane < ZD_TAIL - NA) rec[2 * nk + NA + lane] = 0.0;
    }
    cur = next;
  };

  v2 pa[12], pb[12];  // the row's p and t values of this lane's 12 packed points, fetched one row ahead (c: through cbuf)
  v2 pc[WBX_ZD_C_IN_REGISTERS ? 12 : 1];
  const uint32_t voff_c = (uint32_t)L * 8u;  // this lane's first packed point, bytes into the row
  // A row's three base pointers come out of the plan's offset / gather tables, one row ahead.
  // (r6) Through the CONSTANT address space: scalar loads with SGPR results.  Up to round 6 `resolve` went through the generic
  // helpers (key_bases / row_bases): behind the asm statements' memory clobbers those are VECTOR loads, and the compiler had
  // laid them out as four dependent round trips at the top of every row -- six table entries, `s_waitcnt vmcnt(0)`, gk[row],
  // wait, gd[0], wait, gtab[..], wait -- each an L2 latency that the one other wave of the SIMD cannot cover.  Now: the
  // entries that do not depend on the row (depth_off[i][0], gd[0]) are read once; `lookup` issues the row's four scalar loads
  // at the TOP of the previous row and `pointers` consumes them behind its deterministic lanes (~300 instructions later), where
  // it issues the one dependent load (the climatology's gather table, a few hundred hot bytes), consumed behind pass 1.
  // NULL tables are stood in for by a one-element table of zeros (index mask 0): no control flow around a load.
  constexpr int NIN = HAS_C ? 3 : 2;
  const const_ptr<int64_t> zero64 = (const_ptr<int64_t>)wbx_zero_i64;
  const_ptr<int64_t> tk[NIN];
  int64_t mk[NIN], dep[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    tk[i] = a.key_off[i] ? (const_ptr<int64_t>)a.key_off[i] : zero64;
    mk[i] = a.key_off[i] ? -1 : 0;
    dep[i] = a.depth_off[i] ? ((const_ptr<int64_t>)a.depth_off[i])[0] : 0;
  }
  const bool has_g = HAS_C && a.gtab != nullptr;
  const const_ptr<int32_t> tgk = (has_g && a.gk) ? (const_ptr<int32_t>)a.gk : (const_ptr<int32_t>)wbx_zero_i64;
  const int64_t mgk = (has_g && a.gk) ? -1 : 0;
  const const_ptr<int64_t> tg = has_g ? (const_ptr<int64_t>)a.gtab : zero64;
  const int64_t mg = has_g ? -1 : 0;
  const int64_t gd0 = (has_g && a.gd) ? (int64_t)((const_ptr<int32_t>)a.gd)[0] : 0;
  int64_t kbn[NIN];    // the looked-up row's key offsets, as loaded
  int32_t gkn = 0;     // ... its row of the gather table
  int64_t gvn = 0;     // ... its gather-table entry (elements), as loaded: added where the climatology row is fetched
  const const_ptr<int32_t> tgroup = (const_ptr<int32_t>)group;
  const const_ptr<double> tscale = (const_ptr<double>)scale;
  const const_ptr<double> tdscale = FOLD ? (const_ptr<double>)dscale : (const_ptr<double>)wbx_one_f64;
  auto lookup = [&](int64_t r) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) kbn[i] = tk[i][r & mk[i]];
    if constexpr (HAS_C) gkn = tgk[r & mgk];
  };
  auto pointers = [&](const char*& up, const char*& ut, const char*& uc) {
    up = reinterpret_cast<const char*>(reinterpret_cast<const float*>(a.in[0]) + (kbn[0] + dep[0]));
    ut = reinterpret_cast<const char*>(reinterpret_cast<const float*>(a.in[1]) + (kbn[1] + dep[1]));
    if constexpr (HAS_C) {
      uc = reinterpret_cast<const char*>(reinterpret_cast<const float*>(a.in[2]) + (kbn[2] + dep[2]));
      gvn = tg[((int64_t)gkn * a.ngd + gd0) & mg];
    } else {
      uc = nullptr;
    }
  };
  // The next row's loads are spread over the transform so that they never sit on top of its register peak (pass 2 holds 60
  // registers of butterflies): p and the climatology's LDS-DMA (no registers) behind pass 1's stores, t behind pass 2's.
  // (r6) The row pointers went through integers (uniform_ptr) and lost their address space: as generic pointers the 36 loads
  // of a row came out as flat_load -- issued to the LDS pipeline as well and counted on lgkmcnt, the counter every exchange of
  // the transform waits on.  `gv2` says "global" again.
#if WBX_ZD_FLAT_LOADS
  using gv2 = const v2*;
#else
  using gv2 = const __attribute__((address_space(1))) v2*;
#endif
  auto fetch_p = [&](const char* up, const char* uc) {
    // (the twelve loads of a row address it from its MIDDLE: offsets -2880 .. +2400 bytes all fit the instruction's 13-bit
    // immediate, so a row is one SGPR base + one 32-bit lane offset; from its start the last three needed a 64-bit base of their own)
    gv2 rowp = (gv2)(reinterpret_cast<const v2*>(up + 2880) + L);
#pragma unroll
    for (int i = 0; i < 12; ++i) pa[i] = __builtin_nontemporal_load(rowp + 60 * (i - 6));
    if constexpr (HAS_C && WBX_ZD_C_IN_REGISTERS) {
      gv2 rowc = (gv2)(reinterpret_cast<const v2*>(reinterpret_cast<const char*>(reinterpret_cast<const float*>(uc) + gvn) + 2880) + L);
#pragma unroll
      for (int i = 0; i < 12; ++i) pc[i] = __builtin_nontemporal_load(rowc + 60 * (i - 6));
    } else if constexpr (HAS_C) {
      const char* um = reinterpret_cast<const char*>(reinterpret_cast<const float*>(uc) + gvn);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // dword (2 i + h) of the lane: point 60 i + L, component h
          asm volatile("s_add_u32 m0, %2, %3\n\tglobal_load_lds_dword %0, %1 nt" ::"v"(voff_c), "s"(um), "s"(cbuf_lds), "i"((2 * i + h) * 256)
                       : "memory", "scc", "m0");
          um += h == 0 ? 4 : 476;  // -> the next dword of the pair, then 60 packed points on
          asm volatile("" : "+s"(um));
        }
      }
    }
  };
  auto fetch_t = [&](const char* ut) {
    gv2 rowt = (gv2)(reinterpret_cast<const v2*>(ut + 2880) + L);
#pragma unroll
    for (int i = 0; i < 12; ++i) pb[i] = __builtin_nontemporal_load(rowt + 60 * (i - 6));
  };
  const char *np = nullptr, *nt = nullptr, *nc = nullptr;
  lookup(r0);
  pointers(np, nt, nc);
  fetch_p(np, nc);
  fetch_t(nt);
  int turn = team >> 2;  // waves t and t + 4 of a block share a SIMD: the user priority alternates row by row (see zspec1440_kernel)
  for (int64_t r = r0; r < r1; ++r) {
#if WBX_ZD_PRIO
    turn ^= 1;
    if (turn == 0) __builtin_amdgcn_s_setprio(0);
    else __builtin_amdgcn_s_setprio(1);
#endif
    // (the row's own scalars; loading them a row ahead with the lookups measured the same -- 1.808 against 1.809-1.817 ms -- at
    // 250 instead of 232 registers and with three prefetched registers copied behind an `s_waitcnt vmcnt(0)` at the row's end)
    const int32_t g = tgroup[r];
    const double sc = tscale[r] * quarter_inv_nn;
    const double dw = FOLD ? tdscale[r] : 1.0;
    lookup(r + 1 < r1 ? r + 1 : r);  // (unconditional: the last row looks itself up again)
    if constexpr (HAS_C && !WBX_ZD_C_IN_REGISTERS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the row's LDS-DMA has landed in cbuf
    // ---- the deterministic lanes of this row, on the raw values (lanes 60..63 shadow lane 59: counted out)
    __builtin_amdgcn_sched_barrier(0);
    double d[NA];
    if constexpr ((WBX_ZD_KNOCK & 2) == 0) {
#pragma unroll
    for (int l = 0; l < NA; ++l) d[l] = 0.0;
#if WBX_ZD_F32_CHAINS >= 2
    // (r6, A/B only: make ab-zdf32p2 / ab-zdf32p3)  PACKED fp32: the differences e = p - t, p - c, t - c of a packed point (two
    // longitudes) are one v_pk_add_f32 each -- float32 differences are what the reference forms (deterministic.py:91-123) --, the
    // sums of non-negative terms (|e|, e^2, (p-c)^2, (t-c)^2) run as two fp32 chains of the lane's 12 packed points and are widened
    // once per row.  2: the two sums that cancel (e, (p-c)(t-c)) stay fp64 per point (9 instructions per point instead of 12);
    // 3: they are fp32 chains as well (5.75 per point).
    {
      v2 s_ee = {0.f, 0.f}, s_pp = {0.f, 0.f}, s_tt = {0.f, 0.f}, s_ab = {0.f, 0.f}, s_e = {0.f, 0.f}, s_pt = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const v2 e = pa[i] - pb[i];
        s_ab.x += fabsf(e.x);
        s_ab.y += fabsf(e.y);
        s_ee = __builtin_elementwise_fma(e, e, s_ee);
        if constexpr (WBX_ZD_F32_CHAINS >= 3) s_e += e;
        else d[0] += (double)e.x + (double)e.y;
        if constexpr (HAS_C) {
          const v2 ap = pa[i] - pc[i], at = pb[i] - pc[i];
          s_pp = __builtin_elementwise_fma(ap, ap, s_pp);
          s_tt = __builtin_elementwise_fma(at, at, s_tt);
          if constexpr (WBX_ZD_F32_CHAINS >= 3) s_pt = __builtin_elementwise_fma(ap, at, s_pt);
          else d[5] = fma((double)ap.y, (double)at.y, fma((double)ap.x, (double)at.x, d[5]));
        }
      }
      if constexpr (WBX_ZD_F32_CHAINS >= 3) d[0] = (double)s_e.x + (double)s_e.y;
      d[1] = (double)s_ab.x + (double)s_ab.y;
      d[2] = (double)s_ee.x + (double)s_ee.y;
      if constexpr (HAS_C) {
        d[3] = (double)s_pp.x + (double)s_pp.y;
        d[4] = (double)s_tt.x + (double)s_tt.y;
        if constexpr (WBX_ZD_F32_CHAINS >= 3) d[5] = (double)s_pt.x + (double)s_pt.y;
      }
    }
#elif WBX_ZD_F32_CHAINS
    // (r6, A/B only) The statistics of a POINT in fp32, as the reference forms them -- `predictions - targets`, `(p - t)**2`,
    // `(p - c) * (t - c)` of float32 fields are float32 arrays (deterministic.py:91-123, 222-259); only the weighted
    // dot promotes to float64 (aggregation.py:335).  The four sums of non-negative terms (|e|, e^2, pa^2, ta^2) run as fp32
    // CHAINS of 8 points and are widened per chain -- the bound of the ensemble kernels' chain sums (include/wbx.h: <= 8
    // non-negative terms, <= 4.8e-7 relative per chain, ~1e-7 typical, random in sign across the ~10^5 chains of an output) --;
    // the two sums that cancel (e, pa * ta) are widened per point.  Per point 8 fp32 + 4 half-rate instructions instead of 12
    // half-rate ones: the deterministic lanes were 0.7 of the kernel's 1.9 ms, the vector ALU being what it is bound by (the
    // spectra alone: 1.24 ms), 12 fp64-rate instructions per point against the transform's ~12 fp32 flops per point.
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
      // (an ordering point per chain: without it the compiler forms all 48 anomalies up front, see below)
      if constexpr (HAS_C)
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]));
      else
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]));
#pragma unroll
      for (int i = 4 * ch; i < 4 * ch + 4; ++i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float p = h ? pa[i].y : pa[i].x, t = h ? pb[i].y : pb[i].x;
          const float e = p - t;
          d[0] += (double)e;
          s1 += fabsf(e);
          s2 = fmaf(e, e, s2);
          if constexpr (HAS_C) {
            const float cv = WBX_ZD_C_IN_REGISTERS ? (h ? pc[i].y : pc[i].x) : cbuf[(2 * i + h) * 64 + lane];
            const float ap = p - cv, at = t - cv;
            s3 = fmaf(ap, ap, s3);
            s4 = fmaf(at, at, s4);
            d[5] += (double)(ap * at);
          }
        }
      }
      d[1] += (double)s1;
      d[2] += (double)s2;
      if constexpr (HAS_C) {
        d[3] += (double)s3;
        d[4] += (double)s4;
      }
    }
#else
#pragma unroll
    for (int i = 0; i < 12; i += (WBX_ZD_KNOCK & 16) ? 2 : 1) {  // (knock-out 16: every other point pair -- half the lanes' instructions)
      // The six sums are serial fma chains over the lane's 24 points, so the compiler widens all 72 inputs and forms all 48
      // anomalies up front to have independent work for the chains' latency: ~100 fp64 temporaries (v160..v253 in the ISA), 287
      // VGPRs, the prefetch loads spilled behind vmcnt(0) waits.  An opaque redefinition of the point pair's inputs AND of the
      // running sums right where they are used makes point pair i start when pair i - 1 is done; the other wave of the SIMD
      // covers the chains' latency (an ordering point every second pair instead: 1.93 ms either way).
      if constexpr (HAS_C && WBX_ZD_C_IN_REGISTERS)
        asm volatile("" : "+v"(pa[i]), "+v"(pb[i]), "+v"(pc[i]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]));
      else if constexpr (HAS_C)
        asm volatile("" : "+v"(pa[i]), "+v"(pb[i]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]));
      else
        asm volatile("" : "+v"(pa[i]), "+v"(pb[i]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double p = (double)(h ? pa[i].y : pa[i].x), t = (double)(h ? pb[i].y : pb[i].x);
        const double e = p - t;
        d[0] += e;
        d[1] += fabs(e);
        d[2] = fma(e, e, d[2]);
        if constexpr (HAS_C) {
          const double cv = WBX_ZD_C_IN_REGISTERS ? (double)(h ? pc[i].y : pc[i].x) : (double)cbuf[(2 * i + h) * 64 + lane];
          const double ap = p - cv, at = t - cv;
          d[3] = fma(ap, ap, d[3]);
          d[4] = fma(at, at, d[4]);
          d[5] = fma(ap, at, d[5]);
        }
      }
    }
#endif
    if constexpr (FOLD) {
      if (g != cur) flush(g);  // (the row belongs to the next record: close the running one first; wave-uniform)
#pragma unroll
      for (int l = 0; l < NA; ++l) dsum[l] = fma(d[l], dw, dsum[l]);
    } else if constexpr ((WBX_ZD_KNOCK & 8) != 0) {  // (diagnostic: the lanes' sums are formed but not added over the wave)
      double any = 0.0;
#pragma unroll
      for (int l = 0; l < NA; ++l) any += d[l];
      if (any == 1.2345e300) a.out[r * NA] = any;
    } else {
#pragma unroll
    for (int l = 0; l < NA; ++l) {
      const double tot = wave_sum_uniform(lane < Z14_LANES ? d[l] : 0.0);
      if (lane == 0) a.out[r * NA + l] = tot;
    }
    }
    }
    __builtin_amdgcn_sched_barrier(0);  // the row's deterministic sums are done before the transform starts: their temporaries die here
    if constexpr ((WBX_ZD_KNOCK & 1) == 0) pointers(np, nt, nc);  // the next row's: `lookup`'s loads have had the lanes above to land
    __builtin_amdgcn_sched_barrier(0);  // (or the gather table's load sinks to its use behind pass 1 and is waited for on the spot)
    C2 v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = {{pa[i].x, pb[i].x}, {pa[i].y, pb[i].y}};
    // (r6, measured and dropped: forming x - m with 48 opaque scalar subtractions that ARE this interleave instead of 48 moves + 24
    // packed subtractions -- bit-identical, 9 instructions fewer per row: 1.7226-1.7260 against 1.7260-1.7289 ms here, and the
    // three-wave spectrum kernel, at its 168 registers, reloads one more spilled value inside its loop: 0.355-0.357 -> 0.361 ms)
    const v2 msh = z14_demean(v);  // (the deterministic lanes above took the raw values)
    if (g != cur) flush(g);  // wave-uniform
    z14_pair<0, true, WBX_ZD_TW_EARLY>(v, buf, c, tw1, twr, sc, sc, false, g, accp, accmp, nullptr, [&](int i) {
      // (r6) Unconditional: behind a team's last row the row is asked for once more (`lookup` clamps; 1 row in ~260, found in the
      // L2 / Infinity Cache).  Under `r + 1 < r1` the loads sat in a block of their own: the compiler sank the gather-table
      // load into it (issued and waited for on the spot) and, at 254 registers, joined the two paths with copies of three
      // prefetched registers behind an `s_waitcnt vmcnt(0)` at the END of every row.
      if constexpr ((WBX_ZD_KNOCK & 4) == 0) {
        if constexpr (WBX_ZD_FETCH_AT == 99) {  // spread: a fifth of the row's loads behind each of the first five exchanges
          if (i < 5) {
            constexpr int NLD = HAS_C ? 36 : 24;
            gv2 rowp = (gv2)(reinterpret_cast<const v2*>(np) + L);
            gv2 rowt = (gv2)(reinterpret_cast<const v2*>(nt) + L);
            gv2 rowc = (gv2)(reinterpret_cast<const v2*>(reinterpret_cast<const float*>(nc) + gvn) + L);
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
              if (j * 5 / NLD != i) continue;
              if (j < 12) pa[j] = __builtin_nontemporal_load(rowp + 60 * j);
              else if (HAS_C && j < 24) pc[WBX_ZD_C_IN_REGISTERS ? j - 12 : 0] = __builtin_nontemporal_load(rowc + 60 * (j - 12));
              else pb[j - (HAS_C ? 24 : 12)] = __builtin_nontemporal_load(rowt + 60 * (j - (HAS_C ? 24 : 12)));
            }
          }
        } else if constexpr (WBX_ZD_FETCH_AT == 0) {
          if (i == 0) fetch_p(np, nc);
          if (i == 2) fetch_t(nt);
        } else if constexpr (WBX_ZD_FETCH_AT == 45) {  // t behind the mirror exchange, p (+ c) at the end
          if (i == 4) fetch_t(nt);
          if (i == 5) fetch_p(np, nc);
        } else if (i == WBX_ZD_FETCH_AT) {
          fetch_p(np, nc);
          fetch_t(nt);
        }
      }
    }, acct, accmt, msh);
  }
  flush(cur, true);
}
#pragma clang diagnostic pop

// Measured and not kept (r4): the row's work split between two kinds of waves -- a block of eight transform teams that never
// touch global memory plus four loader waves (three waves per SIMD, 160 VGPRs) that stream p, t, c into registers, form the
// deterministic sums and hand the (p, t) pairs to the teams through their LDS buffers (one flag word per team, s_sleep polling,
// the next row of both its teams ready in a loader's registers).  Correct, and SLOWER on the same box: 1.07-1.08 ms per
// configs[3] chunk against 0.91-0.93 ms for one wave per row (configs[4]: 3.23 against 2.86 ms per chunk) -- the teams pay
// twelve more 16-byte LDS writes and reads per row and the hand-over, and the loaders' fp64 chains take vector-ALU slots from
// the transforms instead of filling idle ones.
