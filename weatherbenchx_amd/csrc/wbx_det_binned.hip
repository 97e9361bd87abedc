// Fused "binned" deterministic reduction: per-point statistics, weights and boolean bin membership in ONE pass.
//
// The two-stage path (wbx_det_partial + wbx_contract_bits) has to keep every dimension the bin masks depend on
// (latitude AND longitude for Regions x land/sea) in the stage-1 partials.  When little is reduced before them --
// the public benchmark's chunks are 1 init x 12 leads (public_benchmark/run_benchmark_evaluation.py:97-101) -- those
// partials are 6 lanes x 8 B per grid point, 4x the inputs, and stage 2 re-reads them per lane.  This kernel reads
// p, t, c once and accumulates  acc[bin][lane] = fma(w(point) * stat(point), member(bin, point), acc)  in registers.
//
// There are too many (bin, lane) accumulators for one register file (34 bins x 7 lanes), but bins are regions: a
// spatially compact patch of the grid touches only a few of them.  So one WAVE owns one patch
//     patch = (cell (A, Bk), 64 consecutive x, a range of the nBr * D reduced rows)
// reads the union of the patch's bins (one 64-bit word per patch from binned_union_kernel, wbx_patch.hpp), deals its
// set bits to K register "slots" (bit masks in SGPRs), and sweeps its rows accumulating only those; a patch with more
// than K bins takes another sweep for the next K.  Per 64-point tile the wave ORs the membership words across lanes
// (DPP) and skips the slots no point of the tile is in.  Waves write tmp[cell][patch][lane][bin] (pre-zeroed) and a
// second kernel sums the patches.  No LDS, no block barrier: the block is one wave.
//
// Semantics as aggregation.py:297-366: NaN * 0 = NaN poisons every bin of a lane (`poison`, added to all bins by the
// second kernel; inside a visited bin the 0.0 / 1.0 membership FMA propagates it), mask / skipna count lanes follow
// the conventions of wbx_det_partial.
#include <cstdlib>
#include <set>
#include <type_traits>

#include "wbx_aidm.hpp"
#include "wbx_patch.hpp"
#include "wbx_s1.hpp"

namespace wbx {

// waves per block = adjacent x tiles walked side by side (see patch_decode).  Measured on the public-benchmark chunk:
// 4 waves per block 0.86 / 0.90 ms (latitude- / longitude-fastest) against 0.80 / 0.77 ms for lone waves -- the wider
// block spreads an XCD's L2 over 4x the weight / membership rows and buys nothing at the DRAM, so 1 it is.
#ifndef WBX_BINNED_WPB
#define WBX_BINNED_WPB 1
#endif
constexpr int BINNED_WPB = WBX_BINNED_WPB;


// MM: 0 none, 1 mask only (one shared count lane), 2 skipna (count lane per value lane), 3 skipna + mask.
// K = accumulator slots, PD = rows of p, t, c in flight.  WM = how the weights are stored: 0 wt[nBk][nBr][nj] (one 8-byte
// load per point next to the membership word), 1 wt[nBk][nj] (they depend on x only -- latitude weights on
// latitude-fastest data: one register per lane for the whole patch), 2 wt[nBk][nBr] (rows only -- latitude weights on
// longitude-fastest data: resolved 64 rows at a time like the row offsets, broadcast per row).
template <typename T, int FUNC, int MM, int K, int PD, int WM>
__global__ void __launch_bounds__(64 * BINNED_WPB) det_binned_kernel(S1Args a, BinnedArgs g) {
  constexpr int NIN = FUNC == WBX_DET6 ? 3 : (FUNC == WBX_DET3 ? 2 : 1);
  constexpr int NL = FUNC == WBX_DET6 ? 6 : (FUNC == WBX_DET3 ? 3 : 1);
  constexpr int NC = MM == 1 ? 1 : (MM >= 2 ? NL : 0);
  constexpr int NA = NL + NC;
  const int lane = threadIdx.x & 63;
  int64_t cell;
  int xt, rs;
  if (!patch_decode<BINNED_WPB>(g, cell, xt, rs)) return;
  const int64_t bk = cell % g.nBk;
  const int64_t A = cell / g.nBk;
  const int64_t R = g.nBr * a.D;
  const int64_t rbeg = (int64_t)rs * g.rows_per_split;
  const int64_t rend = rbeg + g.rows_per_split < R ? rbeg + g.rows_per_split : R;
  constexpr bool has_mask = MM == 1 || MM == 3;

  // lanes beyond a ragged nx re-read the last element and never accumulate
  const bool live = (int64_t)xt * 64 + lane < a.nx;
  const int64_t x = live ? (int64_t)xt * 64 + lane : a.nx - 1;
  const int64_t xw = g.nj > 1 ? x : 0;
  int64_t xoff[WBX_MAX_INPUTS];
#pragma unroll
  for (int i = 0; i < WBX_MAX_INPUTS; ++i) xoff[i] = x * a.xstride[i];

  const int64_t patch = (int64_t)rs * g.nxt + xt;
  if (g.nwords[bk * ((int64_t)g.nrs * g.nxt) + patch] >= 0) return;  // det_atoms_kernel owns this patch
  // union of the patch's bins (binned_atoms_kernel: it does not depend on A, so it is computed once per launch)
  unsigned long long todo = g.uni[bk * ((int64_t)g.nrs * g.nxt) + patch];
  todo = (unsigned long long)readlane64((int64_t)todo, 0);
  double* const out = g.tmp + (cell * ((int64_t)g.nrs * g.nxt) + patch) * (NA * (int64_t)g.nbin);
  double w_lane = 1.0;
  if constexpr (WM == 1) w_lane = g.wt[bk * g.nj + xw];
  bool first = true;
  do {
    // ---- deal the next K bins of the union to the slots (wave-uniform)
    int sbin[K];
    unsigned long long smask[K];  // one bit per dealt slot, 0 for the unused ones
#pragma unroll
    for (int q = 0; q < K; ++q) {
      sbin[q] = todo ? __builtin_ctzll(todo) : 0;
      smask[q] = todo & (~todo + 1ull);
      todo &= todo - 1ull;
    }
    unsigned long long sweep = 0ull;  // the bins of this sweep
#pragma unroll
    for (int q = 0; q < K; ++q) sweep |= smask[q];
    double acc[NA][K];
    double poison[NA];
#pragma unroll
    for (int l = 0; l < NA; ++l) {
      poison[l] = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) acc[l][q] = 0.0;
    }

    for (int64_t rb = rbeg; rb < rend; rb += 64) {
      // lane j resolves row rb + j through the plan's tables; the sweep below broadcasts them one by one
      const int64_t rmine = rb + lane < rend ? rb + lane : rend - 1;
      const int64_t br = rmine < ((int64_t)1 << 31) && a.D < ((int64_t)1 << 31) ? (int64_t)((uint32_t)rmine / (uint32_t)a.D) : rmine / a.D;
      const int64_t d = rmine - br * a.D;
      const int64_t key = (A * g.nBk + bk) * g.nBr + br;
      int64_t kb[WBX_MAX_INPUTS], ro[WBX_MAX_INPUTS];
      key_bases<NIN>(a, key, kb);
      row_bases<NIN>(a, kb, key, d, ro);
      const int64_t wrow_v = (bk * g.nBr + br) * g.nj;
      double wrow_w = 0.0;
      if constexpr (WM == 2) wrow_w = g.wt[bk * g.nBr + br];
      const int nrow = (int)(rend - rb < 64 ? rend - rb : 64);

      // p, t, c (and the mask) stream from HBM: PD rows in flight per wave.  wt / bits are L2 hits: one row ahead.
      T rp[PD], rt[PD], rc[PD];
      uint8_t rv[PD];
      auto fetch_ptc = [&](int j, int u) {
        rp[u] = ld_stream(reinterpret_cast<const T*>(a.in[0]) + readlane64(ro[0], j) + xoff[0]);
        if constexpr (NIN > 1) rt[u] = ld_stream(reinterpret_cast<const T*>(a.in[1]) + readlane64(ro[1], j) + xoff[1]);
        if constexpr (NIN > 2) rc[u] = ld_stream(reinterpret_cast<const T*>(a.in[2]) + readlane64(ro[2], j) + xoff[2]);
        rv[u] = 1;
        if constexpr (has_mask) rv[u] = reinterpret_cast<const uint8_t*>(a.in[3])[readlane64(ro[3], j) + xoff[3]];
      };
      double w_cur, w_nxt = 0.0;
      unsigned long long bits_cur, bits_nxt = 0ull;
      auto fetch_bw = [&](int j, double& w, unsigned long long& bw) {
        const int64_t wi = readlane64(wrow_v, j) + xw;
        bw = g.bits[wi];
        if constexpr (WM == 0) w = g.wt[wi];
        if constexpr (WM == 1) w = w_lane;
        if constexpr (WM == 2) w = __longlong_as_double(readlane64(__double_as_longlong(wrow_w), j));
      };
      auto accumulate = [&](T tp, T tt, T tc, uint8_t tv, double w, unsigned long long bw) {
        const bool ok = live && tv != 0;
        if (ok) {
          const double p = (double)tp, t = (double)tt, c = (double)tc;
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
          if constexpr (MM == 1) val[NL] = 1.0;
          if constexpr (MM >= 2) {
#pragma unroll
            for (int l = 0; l < NL; ++l) {
              const bool fin = !(val[l] != val[l]);
              val[NL + l] = fin ? 1.0 : 0.0;
              val[l] = fin ? val[l] : 0.0;
            }
          }
          double m[NA];
#pragma unroll
          for (int l = 0; l < NA; ++l) {
            m[l] = val[l] * w;
            poison[l] = fma(m[l], 0.0, poison[l]);
          }
#pragma unroll
          for (int q = 0; q < K; ++q) {
            if (smask[q]) {  // wave-uniform: the slot holds a bin
              // The membership compare doubles as the tile test: its lane mask is all zero when no point of the tile is
              // in the bin, and the slot is skipped (a DPP OR-reduction of the tile's words for that cost 26 issue
              // slots per tile, more than the skipped slots saved: 0.83 -> 0.82 / 0.80 -> 0.77 ms per public chunk).
              const bool mem = (bw & smask[q]) != 0ull;
              if (__builtin_amdgcn_ballot_w64(mem)) {
                // membership as 0.0 / 1.0: exact, and NaN * 0 stays NaN
                const double f = __hiloint2double(mem ? 0x3FF00000 : 0, 0);
#pragma unroll
                for (int l = 0; l < NA; ++l) acc[l][q] = fma(m[l], f, acc[l][q]);
              }
            }
          }
        }
      };
      // Every load below is unconditional (row indices are clamped, the surplus ones re-read the block's last row):
      // the compiler can then count the loads in flight and wait for exactly the oldest (s_waitcnt vmcnt(n)); a load
      // under a branch would make it drain the whole queue, prefetches included, once per tile.
      const int last = nrow - 1;
#pragma unroll
      for (int u = 0; u < PD; ++u) fetch_ptc(u < last ? u : last, u);
      fetch_bw(0, w_cur, bits_cur);
      for (int j = 0; j < nrow; j += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
          const int jj = j + u;
          const T tp = rp[u], tt = NIN > 1 ? rt[u] : T(0), tc = NIN > 2 ? rc[u] : T(0);
          const uint8_t tv = rv[u];
          fetch_ptc(jj + PD < last ? jj + PD : last, u);
          fetch_bw(jj + 1 < last ? jj + 1 : last, w_nxt, bits_nxt);
          if (jj < nrow) accumulate(tp, tt, tc, tv, w_cur, bits_cur);  // wave-uniform
          w_cur = w_nxt;
          bits_cur = bits_nxt;
        }
      }
    }

    // ---- fold the lanes; one write per (visited bin, lane)
#pragma unroll
    for (int q = 0; q < K; ++q) {
      if (smask[q]) {
#pragma unroll
        for (int l = 0; l < NA; ++l) {
          const double sum = wave_sum(acc[l][q]);
          if (lane == 0) out[(int64_t)l * g.nbin + sbin[q]] = sum;
        }
      }
    }
    if (first) {
#pragma unroll
      for (int l = 0; l < NA; ++l) {
        const double sum = wave_sum(poison[l]);
        if (lane == 0) g.tmp_poison[(cell * ((int64_t)g.nrs * g.nxt) + patch) * NA + l] = sum;
      }
      first = false;
    }
  } while (todo);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same reduction by ATOMS instead of bins.  A point belongs to several bins (its region, the hemisphere, 'global',
// each again as land or sea: ~8 of the 34), so the slot kernel above spends ~8 slots x NA fp64 FMAs per point, most of
// them with a 0.0 factor -- it is VALU-bound at 30 % of the HBM peak.  But a point belongs to exactly ONE atom (= distinct
// membership word; binned_atoms_kernel lists a patch's atoms and stores every point's index as one byte), and a lane
// that walks down its column stays in the same one or two atoms (land / sea of one region pattern) for long runs.  So
// every lane keeps TWO private accumulator sets keyed by atom id (a 2-entry cache: 2 x NA FMAs per point, whatever the
// number of bins); when some lane with both entries taken meets a third atom -- a region edge, the same row for most
// lanes -- the wave flushes ALL sets, lanes grouped by atom id, one DPP wave sum per (group, statistic), into a
// wave-private [atom][statistic] table in LDS and starts over with empty caches.  At the end of the patch the
// table is expanded to the patch's bins, out[statistic][bin] = sum over atoms that have the bin's bit, and written in the
// slot kernel's tmp layout, so the finish kernel and the NaN rule are shared: a non-finite term makes its atom's sum
// non-finite, `poison` = sum over atoms of (sum * 0) then turns every bin of that statistic into NaN, exactly like the
// reference's xr.dot (aggregation.py:272-277).  Per point the membership costs 1 byte from L2 instead of 8.
// Patches with more than ATOM_MAX atoms (arbitrary user masks) stay with the slot kernel.
// NT: the operands are fetched with the non-temporal hint (see ld_stream).  Off for rows that are not a whole number of
// 128-byte lines (721 floats): neighbouring x tiles share their boundary lines, and a streamed line is gone from L2 before
// the neighbour asks for it -- measured on the latitude-fastest public chunk: 1.53x -> 1.08x the algorithmic bytes, 8 % faster.
// MERGED (MM == 1 with a mask that depends on the W dims only -- a (latitude, longitude) validity mask like the atom ids
// themselves): the mask byte and the atom-id byte of a point are ONE byte, 255 = masked out (aid_merge_kernel, a 1 MB
// pre-pass per launch).  The kernel is bound by the texture addresser -- a wave64 load costs it 16 cycles whatever its width,
// and a row was five of them (p, t, c, mask, atom id): four now.
template <typename T, int FUNC, int MM, int PD, int WM, bool NT, bool MERGED = false>
#ifndef WBX_ATOMS_WAVES
#define WBX_ATOMS_WAVES 4  // waves per SIMD the register budget is cut for (111 VGPRs as it falls)
#endif
// On such rows (!NT) a block is FOUR waves on adjacent x tiles of one row range, meeting at a barrier every 64 rows: the line
// two tiles share is asked for by both within a few rows and from one CU.  Same-box A/B on the latitude-fastest public chunk,
// 1 / 2 / 3 / 4 / 6 / 12 waves per block (`make ab-wpb1 ...`, tools/gpu_r3_ragged_wpb.sh): 0.449 / 0.427 (one box: 0.434 /)
// 0.427 / 0.440 / 0.428 / 0.493 / 0.465 ms -- 4 is 4-5 % faster than 1 on both boxes, 4 without the barrier 3.7 %; 6 and 12
// leave wave slots of the CU empty (16 per CU at 111 VGPRs).
#ifndef WBX_ATOMS_RAGGED_WPB
#define WBX_ATOMS_RAGGED_WPB 4
#endif
#ifndef WBX_ATOMS_PD
#define WBX_ATOMS_PD 4  // rows of p, t, c in flight per wave
#endif
#ifndef WBX_ATOMS_WAVES_COUNTS
#define WBX_ATOMS_WAVES_COUNTS 3  // waves per SIMD of the flavours with one count lane per statistic (MM >= 2: Aggregator(skipna=True)): twice the
                                  // accumulators; at four waves (128 VGPRs) they spilled 152-184 B per lane
#endif
#ifndef WBX_ATOMS_KNOCK
#define WBX_ATOMS_KNOCK 0  // 1: timing diagnostic, see `accumulate`
#endif
#ifndef WBX_ATOMS_VPTR
#define WBX_ATOMS_VPTR 1   // 0: row offsets stepped in scalar registers (rounds 3-5; A/B: make ab-novptr)
#endif
#ifndef WBX_ATOMS_SKIP
#define WBX_ATOMS_SKIP 1   // 0: both entries' FMAs issued every row under EXEC masks (A/B: make ab-atoms6)
#endif
__global__ void __launch_bounds__(64 * (NT ? 1 : WBX_ATOMS_RAGGED_WPB))
__attribute__((amdgpu_waves_per_eu(MM >= 2 ? WBX_ATOMS_WAVES_COUNTS : WBX_ATOMS_WAVES, MM >= 2 ? WBX_ATOMS_WAVES_COUNTS : WBX_ATOMS_WAVES)))
det_atoms_kernel(S1Args a, BinnedArgs g) {
  constexpr int W = NT ? 1 : WBX_ATOMS_RAGGED_WPB;
  constexpr int NIN = FUNC == WBX_DET6 ? 3 : (FUNC == WBX_DET3 ? 2 : 1);
  constexpr int NL = FUNC == WBX_DET6 ? 6 : (FUNC == WBX_DET3 ? 3 : 1);
  constexpr int NC = MM == 1 ? 1 : (MM >= 2 ? NL : 0);
  constexpr int NA = NL + NC;
  constexpr bool has_mask = MM == 1 || MM == 3;
  constexpr int NONE = 255;
  __shared__ double tab_all[W][ATOM_MAX * NA];
  __shared__ unsigned long long wlist_all[W][ATOM_MAX];
  const int lane = threadIdx.x & 63;
  const int wave_in_block = W > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  double* const tab = tab_all[wave_in_block];
  unsigned long long* const wlist = wlist_all[wave_in_block];
  int64_t cell;
  int xt, rs;
  if (!patch_decode<W>(g, cell, xt, rs)) return;
  const int64_t bk = cell % g.nBk;
  const int64_t A = cell / g.nBk;
  const int64_t npatch = (int64_t)g.nrs * g.nxt;
  const int64_t patch = (int64_t)rs * g.nxt + xt;
  const int nw = g.nwords[bk * npatch + patch];
  double* const out = g.tmp + (cell * npatch + patch) * (NA * (int64_t)g.nbin);
  if (nw < 0) {  // too many atoms: det_binned_kernel takes this patch (it adds into zeros: nobody memsets tmp in this mode)
    for (int pr = lane; pr < NA * g.nbin; pr += 64) out[pr] = 0.0;
    return;
  }
  for (int i = lane; i < ATOM_MAX * NA; i += 64) tab[i] = 0.0;
  if (lane < ATOM_MAX) wlist[lane] = g.words[(bk * npatch + patch) * ATOM_MAX + lane];
  __syncthreads();  // (a wave only touches its own table: the barriers pin the order of its LDS accesses for the compiler;
                    //  with W > 1 every wave still in the kernel passes each of them exactly once)
  const int64_t R = g.nBr * a.D;
  const int64_t rbeg = (int64_t)rs * g.rows_per_split;
  const int64_t rend = rbeg + g.rows_per_split < R ? rbeg + g.rows_per_split : R;

  // lanes beyond a ragged nx re-read the last element and never accumulate
  const bool live = (int64_t)xt * 64 + lane < a.nx;
  const int64_t x = live ? (int64_t)xt * 64 + lane : a.nx - 1;
  // 32-bit lane offsets (the launcher checked they fit): row base in SGPRs + one VGPR, no 64-bit VALU adds per load
  uint32_t xo[WBX_MAX_INPUTS];
#pragma unroll
  for (int i = 0; i < WBX_MAX_INPUTS; ++i) xo[i] = (uint32_t)(x * a.xstride[i]);
  const uint32_t xw = g.nj > 1 ? (uint32_t)x : 0u;
  double w_lane = 1.0;
  if constexpr (WM == 1) w_lane = g.wt[bk * g.nj + xw];

  int c0 = NONE, c1 = NONE;  // the atoms this lane is accumulating
  double acc0[NA], acc1[NA];
#pragma unroll
  for (int l = 0; l < NA; ++l) acc0[l] = acc1[l] = 0.0;

  // Flush BOTH entries of EVERY lane into the LDS table and empty the caches: lanes are grouped by atom id, one DPP wave
  // sum per (group, statistic).  Evicting lane by lane was measured 1.3x slower overall: after a region edge every lane
  // drops its stale entries at a different row (when it next crosses a coast), and each of those rows paid a flush.
  auto flush_all = [&](bool at_end = true) {
#ifdef WBX_DIAG_NOFLUSH  // timing diagnostic (wrong sums): what do the flushes in the middle of a sweep cost?
    if (!at_end) {
      c0 = c1 = NONE;
      return;
    }
#endif
#pragma unroll 1
    for (int e = 0; e < 2; ++e) {  // (not unrolled: this code is inlined at every row position of the sweeps)
      const int id = e ? c1 : c0;
      const bool go = id != NONE;
      unsigned long long todo = __builtin_amdgcn_ballot_w64(go);
      while (todo) {
        const int gid = __builtin_amdgcn_readlane(id, __builtin_ctzll(todo));
        const bool sel = go && id == gid;
        double mine = 0.0;
#pragma unroll
        for (int l = 0; l < NA; ++l) {
          const double sum = wave_sum_uniform(sel ? (e ? acc1[l] : acc0[l]) : 0.0);
          if (lane == l) mine = sum;
        }
        if (lane < NA) tab[gid * NA + lane] += mine;
        todo &= ~__builtin_amdgcn_ballot_w64(sel);
      }
    }
#pragma unroll
    for (int l = 0; l < NA; ++l) acc0[l] = acc1[l] = 0.0;
    c0 = c1 = NONE;
  };

  // One row of the patch: hit / miss bookkeeping of the lane's two accumulator sets, then 2 x NA FMAs.
  // The lane predicates of a row are kept as wave masks in scalar registers (ballot of each compare, combined with s_and /
  // s_not, handed back to the vector unit with inverse_ballot): `if (ballot(a && !b && !c))` materialises the combined
  // predicate in a VGPR and compares it again, and the two `id == c` compares for the weights come for free as ~(id != c)
  // (same-box A/B on the public chunk: 0.394 -> 0.384 ms).
  const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);
  auto accumulate = [&](T tp, T tt, T tc, uint8_t tv, double w, int id) {
    const unsigned long long m_ok = live_mask & (MERGED ? __builtin_amdgcn_ballot_w64(id != NONE) : __builtin_amdgcn_ballot_w64(tv != 0));
    unsigned long long n0 = __builtin_amdgcn_ballot_w64(id != c0), n1 = __builtin_amdgcn_ballot_w64(id != c1);
#if WBX_ATOMS_KNOCK >= 1  // timing diagnostic (wrong sums): every point goes to entry 0 as atom 0, no hit / miss bookkeeping
    c0 = 0;
    n0 = 0ull;
    n1 = ~0ull;
#endif
    const bool ok = __builtin_amdgcn_inverse_ballot_w64(m_ok);
    if (WBX_ATOMS_KNOCK == 0 && (m_ok & n0 & n1)) {  // wave-uniform: a lane meets an atom it is not accumulating
      const bool miss = __builtin_amdgcn_inverse_ballot_w64(m_ok & n0 & n1);
      // a lane with both entries taken meets a third atom (a region edge: the same row for most lanes): start over
      bool place = miss;
      if (__builtin_amdgcn_ballot_w64(miss && c0 != NONE && c1 != NONE)) {
        flush_all(false);
        place = ok;  // every entry is empty now: the lanes that had a hit re-enter their atom too
      }
      if (place && c0 == NONE) c0 = id;
      else if (place) c1 = id;
      n0 = __builtin_amdgcn_ballot_w64(id != c0);
      n1 = __builtin_amdgcn_ballot_w64(id != c1);
    }
#if WBX_ATOMS_SKIP
    // (r6) the entries' masks carry `ok` themselves and the values are formed on every lane (clamped loads: any lane holds a real
    // element): one exec save / restore and one branch per row less than `if (ok) { ... }` around it all
    const bool hit0 = __builtin_amdgcn_inverse_ballot_w64(m_ok & ~n0), hit1 = __builtin_amdgcn_inverse_ballot_w64(m_ok & ~n1);
    {
#else
    const bool hit0 = __builtin_amdgcn_inverse_ballot_w64(~n0), hit1 = __builtin_amdgcn_inverse_ballot_w64(~n1);
    if (ok) {
#endif
      const double p = (double)tp, t = (double)tt, c = (double)tc;
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
      if constexpr (MM == 1) val[NL] = 1.0;
      if constexpr (MM >= 2) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          const bool fin = !(val[l] != val[l]);
          val[NL + l] = fin ? 1.0 : 0.0;
          val[l] = fin ? val[l] : 0.0;
        }
      }
      // each entry's FMAs run under its own lane mask (EXEC = the lanes whose point belongs to that entry) with the weight
      // as it is -- a scalar register for row weights: no `hit ? w : 0` selects (four v_cndmask + two v_mov per row), and a
      // row in which no lane uses an entry skips that entry's FMAs
#if WBX_ATOMS_SKIP
      // (r6) ... and the skip is real: the compiler predicates so short a block with EXEC and drops the `s_cbranch_execz` around
      // it (SIInsertSkips keeps a branch only over 12+ instructions, or over something with side effects), so the six fp64 FMAs
      // of an entry no lane is in were ISSUED row after row -- in the interior of a region every lane sits in one entry.  The
      // empty asm statement is that side effect: six fp64 instructions less per row whenever an entry is idle.
      if (hit0) {
        asm volatile("");
#pragma unroll
        for (int l = 0; l < NA; ++l) acc0[l] = fma(val[l], w, acc0[l]);
      }
      if (hit1) {
        asm volatile("");
#pragma unroll
        for (int l = 0; l < NA; ++l) acc1[l] = fma(val[l], w, acc1[l]);
      }
#else
      if (hit0) {
#pragma unroll
        for (int l = 0; l < NA; ++l) acc0[l] = fma(val[l], w, acc0[l]);
      }
      if (hit1) {
#pragma unroll
        for (int l = 0; l < NA; ++l) acc1[l] = fma(val[l], w, acc1[l]);
      }
#endif
    }
  };

  for (int64_t rb = rbeg; rb < rend; rb += 64) {
    // W > 1: the waves of a block walk ADJACENT x tiles of the same rows and meet here every 64 rows, so a boundary line two
    // tiles share is asked for by both within a few rows of each other (waves that have left the kernel do not count)
#ifndef WBX_ATOMS_RAGGED_NOBARRIER  // (A/B: side by side on one CU, free-running)
    if constexpr (W > 1) __builtin_amdgcn_s_barrier();
#endif
    // lane j resolves row rb + j through the plan's tables (key / depth offsets, the climatology gather)
    const int64_t rmine = rb + lane < rend ? rb + lane : rend - 1;
    // (a 64-bit divide is a ~250-instruction sequence, paid per 64 rows: the row count fits 31 bits on every real chunk)
    const int64_t br = R < ((int64_t)1 << 31) ? (int64_t)((uint32_t)rmine / (uint32_t)a.D) : rmine / a.D;
    const int64_t d = rmine - br * a.D;
    const int64_t key = (A * g.nBk + bk) * g.nBr + br;
    int64_t kb[WBX_MAX_INPUTS], ro[WBX_MAX_INPUTS];
    key_bases<NIN>(a, key, kb);
    row_bases<NIN>(a, kb, key, d, ro);
    const int64_t wrow_v = (bk * g.nBr + br) * g.nj;
    double wrow_w = 0.0;
    if constexpr (WM == 2) wrow_w = g.wt[bk * g.nBr + br];
    const int nrow = (int)(rend - rb < 64 ? rend - rb : 64);
    const int last = nrow - 1;

    // Usually the rows of a batch are evenly spaced in every input (a patch walks one dim of the chunk): then row j is
    // base + j * step in SCALAR registers and every load is `global_load v, v_lane_offset, s[row]` -- no broadcast of
    // lane j's offsets (12 v_readlane per row) and no 64-bit vector address arithmetic, in a kernel that is VALU-bound.
    int64_t step[WBX_MAX_INPUTS + 1];
    bool even = true;
    {
      const int up = lane + 1 < nrow ? lane + 1 : lane;  // (the last row compares with itself: difference 0, ignored)
#pragma unroll
      for (int i = 0; i <= WBX_MAX_INPUTS; ++i) {
        const bool used = i == WBX_MAX_INPUTS || i < NIN || (i == 3 && has_mask);
        if (!used) {
          step[i] = 0;
          continue;
        }
        const int64_t mine = i == WBX_MAX_INPUTS ? wrow_v : ro[i];
        const int64_t next = (int64_t)(((uint64_t)(uint32_t)__shfl((int)((uint64_t)mine >> 32), up, 64) << 32) |
                                       (uint32_t)__shfl((int)(uint32_t)(uint64_t)mine, up, 64));
        step[i] = readlane64(next - mine, 0);
        even = even && !__builtin_amdgcn_ballot_w64(lane + 1 < nrow && next - mine != step[i]);
      }
    }

    // EVERYTHING a row needs (p, t, c, mask, atom id, weight) is requested together, PD rows ahead.  The memory counter
    // retires in issue order: an operand fetched "one row ahead" (as the slot kernel does with its membership words)
    // sits behind the data loads of PD - 1 later rows in the queue, so waiting for it drains all of them and the real
    // prefetch depth collapses to one row.
    auto sweep = [&](auto even_tag, auto depth_tag) {
      constexpr bool EVEN = decltype(even_tag)::value;
      constexpr int RD = decltype(depth_tag)::value;
      struct Slots {  // the rows in flight
        T p[RD], t[RD], c[RD];
        uint8_t v[RD], id[RD];
        double w[RD];
      };
      // EVEN: the row offsets of the NEXT fetch, stepped by scalar additions.  Fetches ask for rows 0, 1, 2, ... in order and
      // stay on the last row once they reach it, so `base + j * step` (a 64-bit scalar multiplication per input and row:
      // 24 of the 41 scalar instructions a row cost in round 2; public chunk 0.443 -> 0.415 ms) is never needed.
#if WBX_ATOMS_VPTR
      // (r6) EVEN: the address of the NEXT fetch as a per-lane 64-bit pointer per operand, advanced by ONE vector add (the step sits
      // in a scalar pair).  Round 3 kept the row offsets as scalars -- two scalar adds per operand and row -- and the compiler
      // formed every load's address with a 64-bit vector add on top of that anyway (`v_lshl_add_u64 v, s[row], 2, v[lane base]`):
      // eight of the ~21 scalar instructions a row cost, in a kernel whose gap to its load + arithmetic skeleton is scalar
      // instructions and branches (profiles/r03_binned_vs_skeleton.txt: 52.5 M against 10.4 M, 366 against 297 us).
      const T* vp[3] = {nullptr, nullptr, nullptr};
      const uint8_t *vmask = nullptr, *vid = nullptr;
      const double* vwt = nullptr;
      if constexpr (EVEN) {
#pragma unroll
        for (int i = 0; i < NIN; ++i) vp[i] = reinterpret_cast<const T*>(a.in[i]) + readlane64(ro[i], 0) + xo[i];
        if constexpr (has_mask && !MERGED) vmask = reinterpret_cast<const uint8_t*>(a.in[3]) + readlane64(ro[3], 0) + xo[3];
        vid = (MERGED ? g.aidm : g.aid) + readlane64(wrow_v, 0) + xw;
        if constexpr (WM == 0) vwt = g.wt + readlane64(wrow_v, 0) + xw;
      }
      auto row_of = [&](int i, int j) -> int64_t { return readlane64(i == WBX_MAX_INPUTS ? wrow_v : ro[i], j); };  // (!EVEN)
      auto ld = [&](const T* q) -> T {
        if constexpr (NT) return ld_stream(q);
        return *q;
      };
      auto fetch = [&](Slots& S, int j, int u, auto inside_tag) {
        if constexpr (EVEN) {
          S.p[u] = ld(vp[0]);
          if constexpr (NIN > 1) S.t[u] = ld(vp[1]);
          if constexpr (NIN > 2) S.c[u] = ld(vp[2]);
          S.v[u] = 1;
          if constexpr (has_mask && !MERGED) S.v[u] = *vmask;
          S.id[u] = *vid;
          if constexpr (WM == 0) S.w[u] = *vwt;
          if constexpr (WM == 1) S.w[u] = w_lane;
          if constexpr (WM == 2) S.w[u] = __longlong_as_double(readlane64(__double_as_longlong(wrow_w), j));
          if (decltype(inside_tag)::value || j < last) {  // (wave-uniform; inside: row j + 1 exists, no test)
#pragma unroll
            for (int i = 0; i < NIN; ++i) vp[i] += step[i];
            if constexpr (has_mask && !MERGED) vmask += step[3];
            vid += step[WBX_MAX_INPUTS];
            if constexpr (WM == 0) vwt += step[WBX_MAX_INPUTS];
          }
        } else {
          S.p[u] = ld(reinterpret_cast<const T*>(a.in[0]) + row_of(0, j) + xo[0]);
          if constexpr (NIN > 1) S.t[u] = ld(reinterpret_cast<const T*>(a.in[1]) + row_of(1, j) + xo[1]);
          if constexpr (NIN > 2) S.c[u] = ld(reinterpret_cast<const T*>(a.in[2]) + row_of(2, j) + xo[2]);
          S.v[u] = 1;
          if constexpr (has_mask && !MERGED) S.v[u] = (reinterpret_cast<const uint8_t*>(a.in[3]) + row_of(3, j))[xo[3]];
          const int64_t wi = row_of(WBX_MAX_INPUTS, j);
          S.id[u] = ((MERGED ? g.aidm : g.aid) + wi)[xw];
          if constexpr (WM == 0) S.w[u] = (g.wt + wi)[xw];
          if constexpr (WM == 1) S.w[u] = w_lane;
          if constexpr (WM == 2) S.w[u] = __longlong_as_double(readlane64(__double_as_longlong(wrow_w), j));
        }
      };
#else
      int64_t cur[WBX_MAX_INPUTS + 1];
#pragma unroll
      for (int i = 0; i <= WBX_MAX_INPUTS; ++i) cur[i] = EVEN ? readlane64(i == WBX_MAX_INPUTS ? wrow_v : ro[i], 0) : 0;
      auto row_of = [&](int i, int j) -> int64_t {
        if constexpr (EVEN) return cur[i];
        return readlane64(i == WBX_MAX_INPUTS ? wrow_v : ro[i], j);
      };
      auto fetched = [&](int j, auto inside_tag) {  // row j has just been asked for
        if constexpr (EVEN) {
          const bool more = decltype(inside_tag)::value || j < last;  // inside: row j + 1 exists, no test
#pragma unroll
          for (int i = 0; i <= WBX_MAX_INPUTS; ++i) cur[i] += more ? step[i] : 0;
        }
      };
      auto operand = [&](int i, int j) -> T {
        const T* q = (reinterpret_cast<const T*>(a.in[i]) + row_of(i, j)) + xo[i];
        if constexpr (NT) return ld_stream(q);
        return *q;
      };
      auto fetch = [&](Slots& S, int j, int u, auto inside_tag) {
        S.p[u] = operand(0, j);
        if constexpr (NIN > 1) S.t[u] = operand(1, j);
        if constexpr (NIN > 2) S.c[u] = operand(2, j);
        S.v[u] = 1;
        if constexpr (has_mask && !MERGED) S.v[u] = (reinterpret_cast<const uint8_t*>(a.in[3]) + row_of(3, j))[xo[3]];
        const int64_t wi = row_of(WBX_MAX_INPUTS, j);
        S.id[u] = ((MERGED ? g.aidm : g.aid) + wi)[xw];
        if constexpr (WM == 0) S.w[u] = (g.wt + wi)[xw];
        if constexpr (WM == 1) S.w[u] = w_lane;
        if constexpr (WM == 2) S.w[u] = __longlong_as_double(readlane64(__double_as_longlong(wrow_w), j));
        fetched(j, inside_tag);
      };
#endif
      // every load is unconditional (clamped row indices), see det_binned_kernel
      Slots A;
#pragma unroll
      for (int u = 0; u < RD; ++u) fetch(A, u < last ? u : last, u, std::false_type{});
      int j = 0;
      if constexpr (EVEN) {
        // The rows whose prefetch (RD rows ahead) and its successor both exist need no clamping and no `row < nrow` test: the
        // scalar unit is what this kernel has too much work for (32 scalar instructions + 5.5 branches per row against 4 + 1.3
        // in its load + arithmetic skeleton, tools/gpu_r3_binned_vs_skeleton.sh), and the clamp was 6 of them and a branch.
        const int nmain = nrow > RD + 1 ? ((nrow - RD - 1) / RD) * RD : 0;
        for (; j < nmain; j += RD) {
#pragma unroll
          for (int u = 0; u < RD; ++u) {
            const T tp = A.p[u], tt = NIN > 1 ? A.t[u] : T(0), tc = NIN > 2 ? A.c[u] : T(0);
            const uint8_t tv = A.v[u];
            const double tw = A.w[u];
            const int tid = A.id[u];
            // (consumed first, asked for again second: the load then lands in the registers it replaces -- the other order made
            // the compiler rotate the slots with v_mov at the back edge, behind an `s_waitcnt vmcnt(1)` that drained the queue)
            accumulate(tp, tt, tc, tv, tw, tid);
            fetch(A, j + u + RD, u, std::true_type{});  // j + u + RD + 1 <= last
          }
        }
      }
      Slots B = A;  // (the tail's own copy: its loop-carried registers are not tied to the loop above)
      for (; j < nrow; j += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
          const int jj = j + u;
          const T tp = B.p[u], tt = NIN > 1 ? B.t[u] : T(0), tc = NIN > 2 ? B.c[u] : T(0);
          const uint8_t tv = B.v[u];
          const double tw = B.w[u];
          const int tid = B.id[u];
          fetch(B, jj + RD < last ? jj + RD : last, u, std::false_type{});
          if (jj < nrow) accumulate(tp, tt, tc, tv, tw, tid);  // wave-uniform
        }
      }
    };
    if (even)
      sweep(std::true_type{}, std::integral_constant<int, PD>{});
    else
      sweep(std::false_type{}, std::integral_constant<int, 2>{});  // (gathers that jump between rows, several inits)
  }
  flush_all();
  __syncthreads();

  // ---- atoms -> bins: thread (bin of the union, statistic) sums the atoms that carry the bin's bit
  unsigned long long uni = 0ull;
  for (int k = 0; k < nw; ++k) uni |= wlist[k];
  // every (statistic, bin) of the patch is written, zeros for the bins outside the union: det_binned_finish sums plain rows
  for (int pr = lane; pr < NA * g.nbin; pr += 64) {
    const int l = pr / g.nbin, bit = pr - l * g.nbin;
    double s = 0.0;
    if ((uni >> bit) & 1ull)
      for (int k = 0; k < nw; ++k)
        if ((wlist[k] >> bit) & 1ull) s += tab[k * NA + l];
    out[pr] = s;
  }
  if (lane < NA) {
    double ps = 0.0;
    for (int k = 0; k < nw; ++k) ps = fma(tab[k * NA + lane], 0.0, ps);
    g.tmp_poison[(cell * npatch + patch) * NA + lane] = ps;
  }
}

// WBX_BINNED_ATOMS=0 sends every patch to the slot kernel (A/B timing); WBX_ATOMS_NT=0/1 pins the non-temporal hint
static int atoms_setting(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

template <typename T, int FUNC, int MM, int K, int PD>
static int launch_binned_k(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, const double* wt, const uint64_t* bits,
                         int64_t nA, int64_t nBk, int64_t nBr, int64_t nj, int nbin, double* out, int wmode, const void* prepared,
                         bool mask_on_w, bool accumulate) {
  constexpr int NL = FUNC == WBX_DET6 ? 6 : (FUNC == WBX_DET3 ? 3 : 1);
  constexpr int NA = NL + (MM == 1 ? 1 : (MM >= 2 ? NL : 0));
  static const int use_atoms = atoms_setting("WBX_BINNED_ATOMS", 1);
  static const int use_merged = atoms_setting("WBX_BINNED_MERGED_MASK", 1);  // A/B timing
  // the atom kernel addresses a row as (uniform base) + (32-bit lane offset)
  bool atoms = use_atoms != 0;
  for (int i = 0; i < WBX_MAX_INPUTS; ++i)
    if ((plan->nx - 1) * plan->xstride[i] >= ((int64_t)1 << 31) / (int64_t)sizeof(double) || plan->xstride[i] < 0) atoms = false;
  BinnedArgs g;
  if (int rc = patch_setup(ctx, g, wt, bits, nA * nBk, nBk, nBr, nj, plan->ndepth, plan->nx, NA, nbin, atoms, atoms ? prepared : nullptr, atoms))
    return rc;
  const int64_t grid = patch_grid<BINNED_WPB>(g);
  if (atoms) {
    constexpr int RW = WBX_ATOMS_RAGGED_WPB;
    const int64_t agrid1 = patch_grid<1>(g), agridw = patch_grid<RW>(g);
    static const int order_env = atoms_setting("WBX_PATCH_ORDER", -1);
    static const int nt_env = atoms_setting("WBX_ATOMS_NT", -1);
    const bool ragged_lines = (plan->nx * (int64_t)sizeof(T)) % 128 != 0;
    const bool nt = nt_env >= 0 ? nt_env != 0 : !ragged_lines;
    BinnedArgs ga = g;
    ga.order = order_env >= 0 ? order_env : (ragged_lines ? 1 : 0);
    bool merged = false;
    if constexpr (MM == 1) {
      if (mask_on_w && use_merged != 0 && nj == plan->nx && nj > 1) {  // the mask lives on the atom ids' own index space
        if (int rc = merge_mask_into_atom_ids(ctx, a, ga)) return rc;
        merged = true;
      }
    }
#define g ga
#define WBX_ATOMS_LAUNCH_NT(PDV, WMV, NTV)                                                                                                   \
    do {                                                                                                                                       \
      if constexpr (MM == 1) {                                                                                                                 \
        if (merged) {                                                                                                                          \
          hipLaunchKernelGGL((det_atoms_kernel<T, FUNC, MM, PDV, WMV, NTV, true>), dim3((unsigned)(NTV ? agrid1 : agridw)),                  \
                             dim3(64 * (NTV ? 1 : RW)), 0, ctx->stream, a, g);                                                                \
          break;                                                                                                                               \
        }                                                                                                                                      \
      }                                                                                                                                        \
      hipLaunchKernelGGL((det_atoms_kernel<T, FUNC, MM, PDV, WMV, NTV>), dim3((unsigned)(NTV ? agrid1 : agridw)),                            \
                         dim3(64 * (NTV ? 1 : RW)), 0, ctx->stream, a, g);                                                                    \
    } while (0)
#define WBX_ATOMS_LAUNCH(PDV, WMV)                                  \
    do {                                                              \
      if (nt) WBX_ATOMS_LAUNCH_NT(PDV, WMV, true);                    \
      else WBX_ATOMS_LAUNCH_NT(PDV, WMV, false);                      \
    } while (0)
    // (4 rows in flight per wave; 2 / 4 / 8 were measured 0.61 / 0.59 / 0.63 ms and the variants dropped: they doubled the
    // 144 instantiations of this kernel and the build time of this file)
    if (wmode == 1) WBX_ATOMS_LAUNCH(WBX_ATOMS_PD, 1); else if (wmode == 2) WBX_ATOMS_LAUNCH(WBX_ATOMS_PD, 2); else WBX_ATOMS_LAUNCH(WBX_ATOMS_PD, 0);
#undef WBX_ATOMS_LAUNCH_NT
#undef WBX_ATOMS_LAUNCH
#undef g
    WBX_HIP(hipGetLastError());
  }
  // the slot kernel takes the patches the atom kernel declined (more than ATOM_MAX distinct membership words); its
  // waves return at once everywhere else -- and it is not launched at all behind atom tables that wbx_binned_atoms found
  // free of such patches (an empty launch is 5 us of GPU time plus a dependency gap in a 0.35 ms chunk: 0.394 -> 0.374 ms per
  // public chunk).  (Summing the patches inside the atom kernel as ens_atoms_kernel does -- last arriver per group of 16
  // patches, then per cell -- was measured too and is SLOWER than det_binned_finish here: 0.385 against 0.381 ms per chunk; a
  // patch's table is NA x nbin = 200-400 doubles, and one wave adding sixteen of them waits for its loads batch after batch
  // where the finish kernel spreads them over a block per (cell, statistic).)
  const bool no_overflow = atoms && prepared && ctx->atoms_clean &&
                           static_cast<std::set<const void*>*>(ctx->atoms_clean)->count(prepared) != 0;
  if (no_overflow) return patch_finish(ctx, g, NA, out, accumulate);
  if (wmode == 1)
    hipLaunchKernelGGL((det_binned_kernel<T, FUNC, MM, K, PD, 1>), dim3((unsigned)grid), dim3(64 * BINNED_WPB), 0, ctx->stream, a, g);
  else if (wmode == 2)
    hipLaunchKernelGGL((det_binned_kernel<T, FUNC, MM, K, PD, 2>), dim3((unsigned)grid), dim3(64 * BINNED_WPB), 0, ctx->stream, a, g);
  else
    hipLaunchKernelGGL((det_binned_kernel<T, FUNC, MM, K, PD, 0>), dim3((unsigned)grid), dim3(64 * BINNED_WPB), 0, ctx->stream, a, g);
  WBX_HIP(hipGetLastError());
  return patch_finish(ctx, g, NA, out, accumulate);
}

template <typename T, int FUNC, int MM>
static int launch_binned(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, const double* wt, const uint64_t* bits,
                         int64_t nA, int64_t nBk, int64_t nBr, int64_t nj, int nbin, double* out, int wmode, const void* prepared, bool mask_on_w,
                         bool accumulate) {
  constexpr int NL = FUNC == WBX_DET6 ? 6 : (FUNC == WBX_DET3 ? 3 : 1);
  constexpr int NA = NL + (MM == 1 ? 1 : (MM >= 2 ? NL : 0));
  // Slots: 2 * NA * K accumulator VGPRs + ~70 working registers must stay <= 168 for 3 waves / SIMD.  Measured on the
  // public-benchmark chunk (DET6, 34 bins): K = 6 / 8 / 12 -> 0.92 / 0.85 / 0.95 ms; 2 rows of p, t, c in flight are
  // enough (4: 1.01 ms, the extra registers cost a wave).
  constexpr int K = NA <= 1 ? 32 : (NA <= 2 ? 24 : (NA <= 3 ? 16 : (NA <= 4 ? 12 : (NA <= 6 ? 8 : (NA <= 7 ? 6 : 3)))));
  return launch_binned_k<T, FUNC, MM, K, 2>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
}

template <typename T, int FUNC>
static int binned_mm(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, const double* wt, const uint64_t* bits, int64_t nA,
                     int64_t nBk, int64_t nBr, int64_t nj, int nbin, double* out, int wmode, const void* prepared, bool mask_on_w,
                     bool accumulate) {
  if ((plan->flags & WBX_FLAG_SKIPNA) && (plan->flags & WBX_FLAG_MASKED))
    return launch_binned<T, FUNC, 3>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
  if (plan->flags & WBX_FLAG_SKIPNA) return launch_binned<T, FUNC, 2>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
  if (plan->flags & WBX_FLAG_MASKED) return launch_binned<T, FUNC, 1>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
  return launch_binned<T, FUNC, 0>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
}

template <typename T>
static int binned_func(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, S1Args& a, const double* wt,
                       const uint64_t* bits, int64_t nA, int64_t nBk, int64_t nBr, int64_t nj, int nbin, double* out, int wmode, const void* prepared,
                       bool mask_on_w, bool accumulate) {
  switch (func) {
    case WBX_DET3:
      return binned_mm<T, WBX_DET3>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
    case WBX_DET6:
      return binned_mm<T, WBX_DET6>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
    case WBX_PASS1:
      return binned_mm<T, WBX_PASS1>(ctx, plan, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
  }
  return fail(WBX_ERR_INVALID, "unknown deterministic family %d", func);
}

}  // namespace wbx

extern "C" int wbx_binned_atoms_size(const wbx_s1_plan* plan, int64_t nA, int64_t nBk, int64_t nBr, int32_t w_on_x,
                                     int64_t* bytes_out) {
  using namespace wbx;
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(bytes_out != nullptr, "bytes_out is NULL");
  WBX_REQUIRE(nA >= 1 && nBk >= 1 && nBr >= 1 && plan->nx >= 1 && plan->ndepth >= 1, "empty geometry");
  BinnedArgs g;
  patch_geometry(g, nA * nBk, nBk, nBr, (w_on_x & WBX_BINNED_W_ON_X) ? plan->nx : 1, plan->ndepth, plan->nx);
  *bytes_out = (int64_t)atoms_carve(g, nullptr);
  return 0;
}

extern "C" int wbx_binned_atoms(wbx_ctx* ctx, const wbx_s1_plan* plan, int64_t nA, int64_t nBk, int64_t nBr,
                                int32_t w_on_x, const uint64_t* bits, void* atoms_out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (int rc = check_plan(plan)) return rc;
  WBX_REQUIRE(bits != nullptr && atoms_out != nullptr, "NULL pointer");
  WBX_REQUIRE(nA >= 1 && nBk >= 1 && nBr >= 1 && plan->nx >= 1 && plan->ndepth >= 1, "empty geometry");
  WBX_HIP(hipSetDevice(ctx->device));
  static const int use_atoms = atoms_setting("WBX_BINNED_ATOMS", 1);
  BinnedArgs g;
  patch_geometry(g, nA * nBk, nBk, nBr, (w_on_x & WBX_BINNED_W_ON_X) ? plan->nx : 1, plan->ndepth, plan->nx);
  atoms_carve(g, atoms_out);
  if (int rc = atoms_launch(ctx, g, bits, plan->ndepth, plan->nx, use_atoms != 0)) return rc;
  // tables without a patch that overflows the atom list need no slot-kernel launch behind them: remembered by address (this
  // call runs once per (bins, geometry); a later call that fills the same address again replaces the entry)
  auto* clean = static_cast<std::set<const void*>*>(ctx->atoms_clean);
  if (!clean) ctx->atoms_clean = clean = new std::set<const void*>();
  clean->erase(atoms_out);
  const size_t n = (size_t)g.nBk * g.nrs * g.nxt;
  std::vector<int32_t> host(n);
  WBX_HIP(hipMemcpyAsync(host.data(), g.nwords, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  WBX_HIP(hipStreamSynchronize(ctx->stream));
  bool any = false;
  for (size_t i = 0; i < n; ++i) any = any || host[i] < 0;
  if (!any && use_atoms != 0) {
    if (clean->size() > 64) clean->clear();
    clean->insert(atoms_out);
  }
  return 0;
}

extern "C" int wbx_det_binned(wbx_ctx* ctx, const wbx_s1_plan* plan, int func, int dtype, const void* p, const void* t,
                              const void* c, const uint8_t* mask, const double* wt, const uint64_t* bits, int64_t nA,
                              int64_t nBk, int64_t nBr, int32_t w_on_x, int32_t nbin, const void* prepared, double* out) {
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
  const bool accumulate = (w_on_x & WBX_BINNED_ACCUMULATE) != 0;
  if (nBr * plan->ndepth * plan->nx == 0) {
    if (!accumulate) WBX_HIP(hipMemsetAsync(out, 0, (size_t)nout * sizeof(double), ctx->stream));
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
  WBX_REQUIRE((w_on_x & ~(15 | WBX_BINNED_ACCUMULATE)) == 0 && (w_on_x & 6) != 6, "w_on_x: unknown or contradictory WBX_BINNED_* flags (%d)", w_on_x);
  const bool mask_on_w = (w_on_x & WBX_BINNED_MASK_ON_W) != 0 && (plan->flags & WBX_FLAG_MASKED);
  const int64_t nj = (w_on_x & WBX_BINNED_W_ON_X) ? plan->nx : 1;
  const int wmode = (w_on_x & WBX_BINNED_WT_X_ONLY) ? 1 : ((w_on_x & WBX_BINNED_WT_ROW_ONLY) ? 2 : 0);
  if (dtype == WBX_F32) return binned_func<float>(ctx, plan, func, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
  if (dtype == WBX_F64) return binned_func<double>(ctx, plan, func, a, wt, bits, nA, nBk, nBr, nj, nbin, out, wmode, prepared, mask_on_w, accumulate);
  return fail(WBX_ERR_INVALID, "unknown dtype %d", dtype);
}
