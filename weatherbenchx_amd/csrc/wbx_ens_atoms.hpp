// The ensemble family with weights, boolean bin masks and a validity mask in ONE pass over the members: the public
// benchmark's probabilistic configuration -- CRPS / spread-skill / ensemble-mean RMSE per region x land/sea (34 bins),
// GridAreaWeighting, masked=True (public_benchmark/run_benchmark_evaluation.py:341-354, 365-382; aggregation.py:297-366).
//
// The two-stage route (wbx_ens_partial with x kept + wbx_contract_bits) writes a full-map partial of 5 lanes x 8 B per point
// and reads it back: 43 % of the HBM peak where the un-binned sweep reaches 73 %.  Here the pipelined member sweep of
// ens_pipe_kernel (the next tile's members travel HBM -> LDS by LDS-DMA while this tile is sorted in VGPRs) is combined with
// the atom scheme of det_atoms_kernel (wbx_det_binned.hip):
//   * one WAVE owns one patch = (cell (A, Bk), 64 consecutive x, a short range of the nBr * D reduced rows); a tile of the
//     sweep is one ROW of the patch, so a lane stays on its x column and walks down the rows;
//   * a point belongs to exactly one atom (= distinct 64-bit membership word of the patch; the index comes as one byte per
//     point from binned_atoms_kernel, with the validity mask folded in as 255).  A lane keeps TWO private accumulator sets
//     keyed by atom id (land / sea of one region pattern alternate for long runs): 2 x 5 fp64 FMAs per point, whatever the
//     number of bins -- against ~1100 instructions for the sort and the member sums of that point;
//   * when a lane with both sets taken meets a third atom the wave flushes ALL sets, lanes grouped by atom id, one DPP wave sum
//     per (group, statistic), into the patch's [atom][statistic] table.  That table lives in GLOBAL memory (1.3 KB per patch,
//     cache resident, touched by this wave only): the LDS is the staging buffer -- twelve one-wave blocks of 12.8 KB fill a
//     CU's 160 KB -- and the register file is what the kernel is short of.  The bookkeeping of a row runs right after the
//     staged members have been copied into VGPRs and BEFORE the next row's loads are issued, so a flush waits for nothing but
//     its own accesses;
//   * four value lanes are accumulated, not five: lane 3 = (mean - t)^2 - var / M = lane 4 - lane 2 / M at every point, and
//     the weighted sums are linear, so it is formed once per (patch, bin) from the sums of lanes 2 and 4 (a NaN in either
//     poisons it exactly like a NaN of its own).  The fifth accumulator is the count lane (sum of weights of the valid points);
//   * at the end of the patch the table is expanded to the patch's bins in det_atoms_kernel's tmp layout: det_binned_finish and
//     the NaN rule (poison = sum over atoms of sum * 0: a non-finite term anywhere turns every bin of that statistic NaN, like
//     the reference's xr.dot, aggregation.py:272-277) are shared.
// Weights come factored (WBX_BINNED_WT_X_ONLY / _ROW_ONLY: GridAreaWeighting on latitude- / longitude-fastest chunks):
// w = w_x[x] * w_row[row], one of the two factors being 1 (exact).
// Masks (r5).  A mask that lives on the W dims (WBX_BINNED_MASK_ON_W) is folded into the [bk][br][x] atom-id bytes; a mask with
// strides along A / the depth dims -- what add_nan_mask_to_data builds, data_loaders/base.py:25-56 -- into one id byte per POINT of
// the chunk (aid_merge_points_kernel), and the sweep reads its id byte from row (cell, r) instead of (bk, br): same kernel, same
// instruction count, 209 instead of 208 bytes per point.
// Aggregator(skipna=True) (r5, aggregation.py:339-357: NaN statistics are left out value by value and counted out of that
// statistic's weights): flavour SKIPNA.  The statistics that look at the targets (skill, unbiased MSE, MSE of the mean) are NaN
// where the target or a member is, the statistics of the predictions alone (spread, variance) only where a member is.  A point
// with a NaN member gets weight 0 and values 0 (it is in no sum and no count); a point with a NaN TARGET is accumulated under its
// atom's TWIN with skill = squared error = 0: the atom rows then hold the sums and the count of the target statistics, atom +
// twin rows those of the member-only statistics, and lane 3 = lane 4 - lane 2 / M is formed from the atom rows alone.  No extra
// accumulators (the kernel has three registers to spare).
// Everything else (dense weights, skipna_ensemble, float64 members, M > 64) stays on the two-stage route.
#pragma once
#include <type_traits>
#include "wbx_aidm.hpp"
#include "wbx_ens_impl.hpp"
#include "wbx_patch.hpp"

namespace wbx {

constexpr int ENS_ATOMS_NQ = 5;    // accumulated per atom: skill, spread, variance, squared error of the mean, count
constexpr int ENS_ATOMS_NOUT = 6;  // written per (cell, bin): the five ensemble lanes + the count lane ...
constexpr int ENS_ATOMS_NOUT2 = 12;  // ... and, in twin mode, the same six again over ALL points (mask ignored)
constexpr int ENS_ATOMS_ROWS2 = 2 * ATOM_MAX;  // table rows per patch: an atom and its twin (the masked-out points of the atom)

#ifndef WBX_EA_KNOCK
#define WBX_EA_KNOCK 0  // diagnostic builds only (make ab-eak1 / ab-eak2)
#endif
#ifndef WBX_ENS_ATOMS_ROWS
#define WBX_ENS_ATOMS_ROWS 16  // rows (= 64-point tiles) per patch; WBX_ENS_ATOMS_ROWS in the environment overrides
#endif

// (global_ptr / const_ptr and the stand-in tables wbx_zero_i64 / wbx_one_f64: wbx_common.hpp)

// The sum over a cell's patches happens INSIDE the kernel, in three levels, each done by whichever wave finishes last:
//   level 1: the last wave of a group of G1 consecutive patches expands the group's atom tables to bins -> part1[cell][group]
//   level 2: the last level-1 finisher of G2 consecutive groups adds their records                      -> part2[cell][group2]
//   level 3: the last level-2 finisher of the cell adds those                                            -> out[cell]
// ("last" = an atomic counter per set, reset by the wave that sees it full; every sum runs in index order, so the result does
// not depend on who does it: deterministic, and no atomics on any VALUE).  As separate kernels behind the sweep the same sums
// cost two launches, two dependency gaps and a 14 MB round trip of per-patch bin tables: 0.48 ms per call around a 0.35 ms kernel.
constexpr int ENS_ATOMS_G1 = 4;    // patches per level-1 group (their tables fit the idle staging buffer: 4 x (2560 + 256) B)
constexpr int ENS_ATOMS_G2 = 16;   // level-1 records per level-2 group

// What one wave publishes for another (atom tables, flags, records) is written and read with device-scope relaxed atomics:
// they go to / come from the point where the eight XCDs' L2 caches agree (sc1), so no cache-wide write-back or invalidation is
// needed to hand data from a wave on one XCD to a wave on another -- __threadfence() in front of every patch's arrival (a
// release at agent scope writes the XCD's whole L2 back) took the kernel from 0.35 to 0.83 ms.  Order: a wave's stores, then
// s_waitcnt vmcnt(0) (they are acknowledged), then its arrival (an atomic add); the wave that sees the set complete reads after
// that.
template <typename T>
__device__ __forceinline__ void st_dev(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ T ld_dev(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the wave's LDS stores above are complete (and visible to its own loads below): what __syncthreads() gave a one-wave block,
// without the block barrier that the waves of a wider block -- each on its own way through the finish levels -- must not meet at
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a load that other XCDs' stores are visible to, as a plain (not atomic) instruction: buffer_load ... sc1
__device__ __forceinline__ double ld_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t byte_offset) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)byte_offset, 0, 1 << 4);
  return __hiloint2double((int)v.y, (int)v.x);
}

struct EnsAtomsArgs {
  const double* wx;    // [nBk][nj] or NULL (all ones)
  const double* wrow;  // [nBk][nBr] or NULL (all ones)
  double* tab;         // [cell][patch][ATOM_MAX][NQ]: the patches' atom tables (rows k < nwords cleared by their own wave)
  double* part1;       // [cell][ng1][NOUT][64]
  double* part2;       // [cell][ng2][NP]
  uint32_t* counters;  // [cell][ng1] | [cell][ng2] | [cell], zero before the first launch and after every launch
  uint32_t* queue;     // persistent launches: [64] patch tickets, one queue per (XCD, eighth of its patches); zero at launch
  int32_t queue_static;  // A/B (WBX_ENS_ATOMS_STATIC=1): no tickets, a fixed share of the patches per wave
  uint32_t* queue_next;  // the next launch's [64]: zeroed by this one (nobody else touches it meanwhile: stream order)
  double* out;         // [cell][NOUT][nbin]
  int32_t ng1, ng2;
  int32_t masked;      // 0: no mask; 1: the atom ids are g.aidm, 255 = masked out; 2 (twin): g.aidm, id | 0x80 = masked out -- such
                       // points are accumulated under the atom's TWIN, so that one launch yields the masked sums (atoms only) AND
                       // the unmasked ones (atoms + twins): the reference masks skill / unbiased MSE / mean MSE of a variable
                       // whose targets carry a mask but not its spread / variance (statistics of the predictions alone)
  int32_t twin_rows;   // the twin half of the atom tables is in use: masked == 2, or the SKIPNA flavour (NaN targets)
  int32_t out_mode;    // 0: out[cell][6][nbin]; 1 (twin): [12]; 2 (SKIPNA): [10] = five values + their five counts; 3 (SKIPNA +
                       // twin): [20] = the masked ten, then the ten over all points (see wbx.h)
  int32_t accumulate;  // WBX_BINNED_ACCUMULATE: level 3 adds the cell's sums into `out` (a chunk loop's accumulator) instead of storing them
  int64_t id_cell_rows;  // 0: the id bytes are [bk][br][nj] (bins / a mask on the W dims); R = nBr * D: one id byte per point,
                         // [cell][r][nj] (a mask with strides along A / the depth dims)
  int64_t br_per_split;  // g.rows_per_split / D
  unsigned long long* prof;  // diagnostic builds (WBX_EA_PROF): eight time stamps per patch, else NULL
};

#ifndef WBX_ENS_ATOMS_ROW_BARRIER
#define WBX_ENS_ATOMS_ROW_BARRIER 4  // rows between two block barriers of the ragged-row flavour (power of two; 0 = none: make
                                     // ab-eabar0).  (r6) 4: the four waves of a block -- adjacent x tiles of the same rows -- ask for
                                     // the 128-byte lines two tiles share within the few microseconds a streamed line survives in the
                                     // XCD's L2: FETCH_SIZE x 2 = 1.076 x the algorithmic bytes instead of 1.141 x (NaN mask 1.10
                                     // instead of 1.17) at the same time per chunk (profiles/r06_ens_atoms_rowbarrier.txt)
#endif
#ifndef WBX_EA_PERSIST
#define WBX_EA_PERSIST 0  // make ab-eapersist: persistent waves on one-wave blocks (see ens_atoms_kernel<.., PERSIST>); measured
#endif                    // SLOWER than one block per patch (profiles/r05_ens_atoms_persistent_ab.txt) and not in the shipped library
#ifndef WBX_EA_PERSIST_RELOAD
#define WBX_EA_PERSIST_RELOAD 1  // make ab-eanoreload: the persistent kernel keeps its arguments in registers across patches
#endif
#ifndef WBX_EA_PROF
#define WBX_EA_PROF 0  // make ab-eaprof: phase stamps of every wave, dumped to $WBX_EA_PROF_DUMP after each launch
#endif
#if WBX_EA_PROF
#define WBX_EA_STAMP(slot) \
  do { if (lane == 0) e.prof[(cell * npatch + patch) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define WBX_EA_STAMP(slot) do {} while (0)
#endif

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // m0 is written by the LDS-DMA statements and listed as clobbered
// WPB waves per block: WPB ADJACENT x tiles of one (cell, row split), one wave each -- independent (no barrier), but they start
// together and walk the same rows, so the 128-byte lines two tiles of a 721-point row share are asked for within a row or two
// of each other and the second request finds them in the CU's L1 / the XCD's L2 (alone, the neighbouring tile is another wave
// somewhere on the XCD, up to a patch length away in time, and the line has left the L2 by then: FETCH_SIZE 1.28 x the
// algorithmic bytes on the latitude-fastest public chunk).  Used with !NT (rows that are not whole lines), like det_atoms_kernel.
#ifndef WBX_ENS_ATOMS_RAGGED_WPB
#define WBX_ENS_ATOMS_RAGGED_WPB 4
#endif
// MODE 0: the id bytes are [bk][br][x] (bins, or bins + a mask on the W dims) -- the round-4 kernel, nothing added to its loop;
//      1: one id byte per point of the chunk (a mask with strides along A / the depth dims);
//      2: Aggregator(skipna=True), either kind of id table (run-time)
// The kernel's arguments field by field out of the kernel-argument segment (scalar loads; whole-struct copies through a word
// pointer stay in scratch memory)
__device__ __forceinline__ void load_args(S1Args& d, const_ptr<S1Args> s) {
#pragma unroll
  for (int i = 0; i < WBX_MAX_INPUTS; ++i) {
    d.in[i] = s->in[i];
    d.key_off[i] = s->key_off[i];
    d.depth_off[i] = s->depth_off[i];
    d.xstride[i] = s->xstride[i];
  }
  d.gk = s->gk, d.gd = s->gd, d.gtab = s->gtab, d.ngd = s->ngd;
  d.nkey = s->nkey, d.D = s->D, d.nx = s->nx, d.dchunk = s->dchunk;
  d.nchunk = s->nchunk, d.nxtile = s->nxtile, d.flags = s->flags;
  d.out = s->out, d.xw = s->xw, d.M = s->M, d.mstride = s->mstride, d.lane = s->lane, d.ngd_t = s->ngd_t;
}
__device__ __forceinline__ void load_args(BinnedArgs& d, const_ptr<BinnedArgs> s) {  // (all but split_br: this kernel reads split_tab)
  d.wt = s->wt, d.bits = s->bits, d.nBk = s->nBk, d.nBr = s->nBr, d.nj = s->nj;
  d.nbin = s->nbin, d.nxt = s->nxt, d.nrs = s->nrs, d.rows_per_split = s->rows_per_split;
  d.ncell = s->ncell, d.nblocks = s->nblocks, d.tmp = s->tmp, d.tmp_poison = s->tmp_poison, d.uni = s->uni;
  d.aid = s->aid, d.aidm = s->aidm, d.words = s->words, d.nwords = s->nwords, d.atoms = s->atoms, d.order = s->order;
  d.taper = s->taper, d.split_tab = s->split_tab;
}
__device__ __forceinline__ void load_args(EnsAtomsArgs& d, const_ptr<EnsAtomsArgs> s) {
  d.wx = s->wx, d.wrow = s->wrow, d.tab = s->tab, d.part1 = s->part1, d.part2 = s->part2;
  d.counters = s->counters, d.queue = s->queue, d.queue_next = s->queue_next, d.queue_static = s->queue_static, d.out = s->out, d.ng1 = s->ng1, d.ng2 = s->ng2;
  d.masked = s->masked, d.twin_rows = s->twin_rows, d.out_mode = s->out_mode, d.accumulate = s->accumulate;
  d.id_cell_rows = s->id_cell_rows, d.br_per_split = s->br_per_split, d.prof = s->prof;
}

// One patch, by one wave: (virtual) block `vb` of patch_grid<WPB>(g); lds_raw = the wave's staging slice.
template <int MP, bool EXACT, bool NT, int MODE>
__device__ __forceinline__ void ens_atoms_patch(const S1Args& a, const BinnedArgs& g, const EnsAtomsArgs& e, uint32_t vb,
                                                unsigned char* const lds_raw) {
  constexpr bool SKIPNA = MODE == 2;
  constexpr int WPB = NT ? 1 : WBX_ENS_ATOMS_RAGGED_WPB;
  using Op = EnsOpF32<MP, EXACT, WBX_ENS_SORT>;
  constexpr int NQ = ENS_ATOMS_NQ, NOUT = ENS_ATOMS_NOUT;
  constexpr int NLDS = MP < WBX_ENS_PIPE_NLDS ? MP : WBX_ENS_PIPE_NLDS;  // members staged through the LDS
  constexpr int NREG = MP - NLDS;                                        // members prefetched into VGPRs
  constexpr int NONE = 255;
  float(*stage)[64] = reinterpret_cast<float(*)[64]>(lds_raw);
  const int lane = threadIdx.x & 63;
  const int M = EXACT ? MP : a.M;
  int64_t cell;
  int xt, rs;
  if (!patch_decode<WPB>(g, cell, xt, rs, vb)) return;
  const int64_t bk = cell % g.nBk;
  const int64_t A = cell / g.nBk;
  const int64_t npatch = (int64_t)g.nrs * g.nxt;
  const int64_t patch = (int64_t)rs * g.nxt + xt;
  const int nw = g.nwords[bk * npatch + patch];
  WBX_EA_STAMP(0);
  constexpr int TR = ENS_ATOMS_ROWS2;  // table rows per patch (the twin half is only touched in twin mode)
  const bool twin = SKIPNA || e.twin_rows != 0;
  double* const tab = e.tab + (cell * npatch + patch) * (TR * NQ);
  // nw < 0: more than ATOM_MAX distinct membership words in one patch (arbitrary user masks).  This kernel has no slot
  // fallback: the host checks the tables before it chooses this route (wbx_ens_binned_atoms reports such patches); a caller
  // that did not gets NaN in every bin of the cell instead of silently wrong sums.
  int64_t rbeg, rend;
  int br_first = 0;
  if (g.taper) {  // (scalar loads: the table sits next to the atom tables)
    const const_ptr<int32_t> tab = (const_ptr<int32_t>)g.split_tab;
    br_first = tab[rs];
    rbeg = (int64_t)br_first * a.D;
    rend = (int64_t)tab[rs + 1] * a.D;
  } else {
    const int64_t R = g.nBr * a.D;
    rbeg = (int64_t)rs * g.rows_per_split;
    rend = rbeg + g.rows_per_split < R ? rbeg + g.rows_per_split : R;
  }

  // lanes beyond a ragged nx re-read the last element and never accumulate
  const bool live = (int64_t)xt * 64 + lane < a.nx;
  const uint32_t x = live ? (uint32_t)xt * 64u + (uint32_t)lane : (uint32_t)a.nx - 1u;
  const uint32_t xw = g.nj > 1 ? x : 0u;
  const double w_lane = e.wx ? e.wx[bk * g.nj + xw] : 1.0;
  const uint8_t* const ids = e.masked ? g.aidm : g.aid;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&stage[0][0];
  const int64_t mstride_b = a.mstride * 4;
  const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(live);

  int c0 = NONE, c1 = NONE;  // the atoms this lane is accumulating
  double acc0[NQ], acc1[NQ];
#pragma unroll
  for (int l = 0; l < NQ; ++l) acc0[l] = acc1[l] = 0.0;
  unsigned long long touched = 0ull;  // table rows that have been written (wave-uniform): the first flush of a row stores
  // ... and the rows no flush ever writes must read as zeros for the wave that sums the group: lane l < NQ clears its column
  // (the lane that writes it later: same lane, same address, in order), fire and forget
  if (lane < NQ)
    for (int k = 0; k < nw; ++k) {
      st_dev(tab + k * NQ + lane, 0.0);
      if (twin) st_dev(tab + (ATOM_MAX + k) * NQ + lane, 0.0);
    }

  // Flush BOTH sets of EVERY lane into the patch's table and empty them: lanes grouped by atom id, one DPP wave sum per
  // (group, statistic); lane l < NQ owns column l of the table (the only lane that ever reads or writes it).
  auto flush_all = [&]() {
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
      const int id = s ? c1 : c0;
      const bool go = id != NONE;
      unsigned long long todo = __builtin_amdgcn_ballot_w64(go);
      while (todo) {
        const int gid = __builtin_amdgcn_readlane(id, __builtin_ctzll(todo));
        const bool sel = go && id == gid;
        double mine = 0.0;
#pragma unroll
        for (int l = 0; l < NQ; ++l) {
          const double sum = wave_sum_uniform(sel ? (s ? acc1[l] : acc0[l]) : 0.0);
          if (lane == l) mine = sum;
        }
        const int row = (gid & (ATOM_MAX - 1)) + ((gid >> 7) ? ATOM_MAX : 0);  // ids 128 + k are the twins
        if (lane < NQ) {
          double* q = tab + row * NQ + lane;
          st_dev(q, ((touched >> row) & 1ull) ? ld_dev(q) + mine : mine);
        }
        touched |= 1ull << row;
        todo &= ~__builtin_amdgcn_ballot_w64(sel);
      }
    }
#pragma unroll
    for (int l = 0; l < NQ; ++l) acc0[l] = acc1[l] = 0.0;
    c0 = c1 = NONE;
  };

  // ---- rows.  Everything about a row is wave-uniform and lives in SCALAR registers: the offsets of row i + 2 are looked up
  // in the plan's tables (scalar loads) right after the loads of row i + 1 have been issued, and arrive while row i is sorted.
  // (A first version resolved 64 rows at a time lane-parallel like det_atoms_kernel: eight vector registers this kernel does
  // not have.)  Row r of the cell = (br, d) = (r / D, r mod D), stepped without a division.
  const int nrows = nw < 0 ? 0 : (int)(rend - rbeg);  // (the launcher checked that the row counts fit 31 bits)
  const int nD = (int)a.D;
  int br_a = g.taper ? br_first : (int)((int64_t)rs * e.br_per_split);  // (row splits are whole Br rows: rbeg = br_a * D)
  int d_a = 0;
  const int br_last = (int)((rend - 1) / a.D);
  const int64_t key0 = (A * g.nBk + bk) * g.nBr, wrow0 = bk * g.nBr;
  // id bytes: row (bk, br) of the [bk][br][nj] table, or row (cell, r = br * D + d) of the per-point table
  // (the per-point rows of a patch are consecutive: a row counter that runs with the look-ahead, clamped like it)
  const bool point_ids = MODE == 1 || (MODE == 2 && e.id_cell_rows != 0);
  const int64_t idrow0 = point_ids ? cell * e.id_cell_rows : wrow0;
  int r_a = (int)rbeg;
  const int r_last = (int)(rend - 1);
  int64_t ra0k = 0, ra0d = 0, ra1k = 0, ra1d = 0, wra = 0;  // the row looked up ahead (the table entries as they were loaded:
  double wwa = 1.0;                                          // adding them here would wait for the loads on the spot)
  // Unconditional (past the patch's last row it looks the last row up again): under a branch the scalar loads would be
  // waited for at the join, in front of the sort.
  const const_ptr<int64_t> zero = (const_ptr<int64_t>)wbx_zero_i64;
  const const_ptr<int64_t> tk0 = a.key_off[0] ? (const_ptr<int64_t>)a.key_off[0] : zero, td0 = a.depth_off[0] ? (const_ptr<int64_t>)a.depth_off[0] : zero;
  const const_ptr<int64_t> tk1 = a.key_off[1] ? (const_ptr<int64_t>)a.key_off[1] : zero, td1 = a.depth_off[1] ? (const_ptr<int64_t>)a.depth_off[1] : zero;
  const const_ptr<double> twr = e.wrow ? (const_ptr<double>)e.wrow : (const_ptr<double>)wbx_one_f64;
  const int64_t mk0 = a.key_off[0] ? -1 : 0, md0 = a.depth_off[0] ? -1 : 0, mk1 = a.key_off[1] ? -1 : 0, md1 = a.depth_off[1] ? -1 : 0,
                mwr = e.wrow ? -1 : 0;  // index masks: a stand-in table has one element
  auto resolve_ahead = [&]() {
    const int64_t key = key0 + br_a;
    ra0k = tk0[key & mk0];
    ra0d = td0[(int64_t)d_a & md0];
    ra1k = tk1[key & mk1];
    ra1d = td1[(int64_t)d_a & md1];
    if constexpr (MODE == 0) {
      wra = (wrow0 + br_a) * g.nj;
    } else {
      wra = (idrow0 + (int64_t)__builtin_amdgcn_readfirstlane(point_ids ? r_a : br_a)) * g.nj;  // (the compiler keeps r_a on the VALU)
      r_a = r_a < r_last ? r_a + 1 : r_a;
    }
    wwa = twr[(wrow0 + br_a) & mwr];
    const bool wrap = d_a + 1 == nD;
    d_a = wrap ? 0 : d_a + 1;
    br_a = (wrap && br_a < br_last) ? br_a + 1 : br_a;
  };

  // the tile (row) in flight
  float xn[NREG > 0 ? NREG : 1], tn = 0.f;
  int idn = NONE;
  int64_t ron0 = 0, ron1 = 0;
  double wrn = 1.0;
  const uint32_t sx0 = (uint32_t)a.xstride[0] * 4u, sx1 = (uint32_t)a.xstride[1] * 4u, sxw = g.nj > 1 ? 1u : 0u;
  auto issue = [&]() {  // the row that was looked up ahead
    ron0 = ra0k + ra0d;
    ron1 = ra1k + ra1d;
    wrn = wwa;
    // 32-bit BYTE offsets of the lane (the launcher checked they fit), formed here from x: every load of the sweep is
    //  SGPR row base + one VGPR offset.  Kept as loop-invariant registers they were three more than the kernel has, hoisted
    // zero extensions of them are 64-bit pairs, and the instruction selector only folds  base + zext(offset)  into one load
    // when it sees the extension in the same block: hence the opaque statement.
    uint32_t o0 = x * sx0, o1 = x * sx1, ow = x * sxw;
    asm volatile("" : "+v"(o0), "+v"(o1), "+v"(ow));
    // (opaque SGPR row pointers: left alone, the compiler re-associates  (base + row) + lane offset  into a hoisted 64-bit
    //  VECTOR pointer per operand plus a scalar)
    const float* tr = reinterpret_cast<const float*>(a.in[1]) + ron1;
    asm volatile("" : "+s"(tr));
    const global_ptr<float> tq = (global_ptr<float>)((global_ptr<char>)tr + o1);
    tn = NT ? __builtin_nontemporal_load(tq) : *tq;
    const uint8_t* ir = ids + wra;
    asm volatile("" : "+s"(ir));
    idn = ((global_ptr<uint8_t>)ir)[ow];
    const char* um = uniform_ptr(reinterpret_cast<const char*>(a.in[0]) + ron0 * 4);
#pragma unroll
    for (int m = 0; m < NLDS; ++m) {
      if (EXACT || m < M) {
        if constexpr (NT)
          asm volatile("s_add_u32 m0, %2, %3\n\tglobal_load_lds_dword %0, %1 nt" ::"v"(o0), "s"(um), "s"(lds0), "i"(m * 256)
                       : "memory", "scc", "m0");
        else
          asm volatile("s_add_u32 m0, %2, %3\n\tglobal_load_lds_dword %0, %1" ::"v"(o0), "s"(um), "s"(lds0), "i"(m * 256)
                       : "memory", "scc", "m0");
      }
      um += mstride_b;
      asm volatile("" : "+s"(um));  // one s_add_u32 / s_addc_u32 per member, not a table of 50 hoisted products
    }
#pragma unroll
    for (int m = NLDS; m < MP; ++m) {  // the members beyond the staging buffer ride in registers
      const global_ptr<float> pr = (global_ptr<float>)((global_ptr<char>)um + o0);
      if constexpr (NT)
        xn[m - NLDS] = (EXACT || m < M) ? __builtin_nontemporal_load(pr) : INFINITY;
      else
        xn[m - NLDS] = (EXACT || m < M) ? *pr : INFINITY;
      um += mstride_b;
      asm volatile("" : "+s"(um));
    }
  };

  WBX_EA_STAMP(8);  // prologue done
  if (nrows > 0) {
    resolve_ahead();
    issue();
    resolve_ahead();
  }
  WBX_EA_STAMP(9);  // first row asked for
  for (int i = 0; i < nrows; ++i) {
    typename Op::Regs r;
#if WBX_ENS_ATOMS_ROW_BARRIER > 0
    // (A/B, make ab-eabar*) the waves of a block -- adjacent x tiles of the same rows -- meet every few rows, so that the boundary
    // lines two tiles share are asked for within the few microseconds a streamed line survives in the XCD's L2
    if constexpr (WPB > 1) {
      if ((i & (WBX_ENS_ATOMS_ROW_BARRIER - 1)) == 0) __builtin_amdgcn_s_barrier();
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if WBX_EA_PROF
    if (i == 0) WBX_EA_STAMP(10);  // first row has landed
    if (i == 1) WBX_EA_STAMP(11);  // ... the second
#endif
#pragma unroll
    for (int m = 0; m < NLDS; ++m) r.xm[m] = (EXACT || m < M) ? stage[m][lane] : INFINITY;
#pragma unroll
    for (int m = NLDS; m < MP; ++m) r.xm[m] = xn[m - NLDS];
    r.t = tn;
    int id = idn;
    if constexpr (SKIPNA) id = (tn != tn && id != NONE) ? (id | 0x80) : id;  // a NaN target: the point goes to its atom's twin
    double w = w_lane * wrn;
    int64_t rocur[WBX_MAX_INPUTS];
    rocur[0] = ron0;
    rocur[1] = ron1;
    rocur[2] = rocur[3] = 0;

    // ---- hit / miss bookkeeping of the lane's two accumulator sets (wave masks in scalar registers, see det_atoms_kernel);
    // no load of the next row is in flight here
    const unsigned long long m_ok = live_mask & __builtin_amdgcn_ballot_w64(id != NONE);
    unsigned long long n0 = __builtin_amdgcn_ballot_w64(id != c0), n1 = __builtin_amdgcn_ballot_w64(id != c1);
#if WBX_EA_KNOCK >= 1  // timing diagnostic (wrong sums): every point goes to set 0 as atom 0, no hit / miss bookkeeping
    c0 = 0;
    n0 = 0ull;
    n1 = ~0ull;
#endif
    if (WBX_EA_KNOCK == 0 && (m_ok & n0 & n1)) {  // wave-uniform: a lane meets an atom it is not accumulating
      const bool ok = __builtin_amdgcn_inverse_ballot_w64(m_ok);
      const bool miss = __builtin_amdgcn_inverse_ballot_w64(m_ok & n0 & n1);
      bool place = miss;
      if (__builtin_amdgcn_ballot_w64(miss && c0 != NONE && c1 != NONE)) {
        flush_all();
        place = ok;  // every set is empty now: the lanes that had a hit re-enter their atom too
      }
      if (place && c0 == NONE) c0 = id;
      else if (place) c1 = id;
      n0 = __builtin_amdgcn_ballot_w64(id != c0);
      n1 = __builtin_amdgcn_ballot_w64(id != c1);
    }
    const unsigned long long h0 = m_ok & ~n0, h1 = m_ok & ~n1;

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile has left the staging buffer
    if (i + 1 < nrows) issue();
    resolve_ahead();

    double val[Op::NLANE];
    Op::template finish<true>(a, rocur, (int64_t)x, r, val);
    double q[4] = {val[0], val[1], val[2], val[4]};
    if constexpr (SKIPNA) {
      // spread is NaN iff a member is NaN / infinite (it never looks at the target): such a point is in no sum and no count;
      // a twin point (NaN target, or masked out) adds to the member-only statistics alone
      const bool bad = q[1] != q[1], tw = (id & 0x80) != 0;
      w = bad ? 0.0 : w;
      q[0] = (bad || tw) ? 0.0 : q[0];
      q[3] = (bad || tw) ? 0.0 : q[3];
      q[1] = bad ? 0.0 : q[1];
      q[2] = bad ? 0.0 : q[2];
    }
    if (__builtin_amdgcn_inverse_ballot_w64(h0)) {
#pragma unroll
      for (int l = 0; l < 4; ++l) acc0[l] = fma(q[l], w, acc0[l]);
      acc0[4] += w;
    }
    if (__builtin_amdgcn_inverse_ballot_w64(h1)) {
#pragma unroll
      for (int l = 0; l < 4; ++l) acc1[l] = fma(q[l], w, acc1[l]);
      acc1[4] += w;
    }
  }
  WBX_EA_STAMP(1);
  flush_all();
  WBX_EA_STAMP(2);
#if WBX_EA_KNOCK >= 2  // timing diagnostic: no sums over patches
  return;
#endif

  // ---- the patch is done: publish its table, then the sums over patches (see EnsAtomsArgs).  A record is [NOUT][64]: lane =
  // bin, so every lane of the wave adds the same six statistics and only the membership factor differs.
  constexpr int NOUT2 = ENS_ATOMS_NOUT2;
  const int NP = (twin ? NOUT2 : NOUT) * 64;  // lanes of a record between the levels
  // -> true for the wave that completes the set (exactly one): everything the others wrote before their arrival is visible to it
  auto last_of = [&](uint32_t* counter, uint32_t total) -> bool {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's table / record has arrived where every XCD sees it
    uint32_t seen = 0;
    if (lane == 0) seen = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)seen);
    if (seen != total - 1) return false;
    if (lane == 0) st_dev(counter, 0u);  // ready for the next launch
    return true;
  };
  // sum of n consecutive records at src (index order); sc1 BUFFER loads: the compiler keeps a batch of them in flight, where
  // it waits for every atomic load on its own (16 records = 96 loads one after the other took 18 us of the kernel's tail)
  // (the number of lanes is a compile-time constant of each flavour: under the run-time `twin` test the loads of the upper six
  //  lanes were issued and waited for one by one -- 192 of them in a level-2 sum: 28-100 us of a masked launch's tail,
  //  tools/gpu_r4_ens_prof_mask.sh)
  auto add_records_n = [&](auto nl_tag, const double* src, int n, double (&sum)[NOUT2]) {
    constexpr int NL = decltype(nl_tag)::value;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(src), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int l = 0; l < NOUT2; ++l) sum[l] = 0.0;
#pragma unroll 4
    for (int q = 0; q < n; ++q) {
#pragma unroll
      for (int l = 0; l < NL; ++l) sum[l] += ld_sc1(rs, (uint32_t)((q * NP + l * 64 + lane) * 8));
    }
  };
  auto add_records = [&](const double* src, int n, double (&sum)[NOUT2]) {
    if (twin)
      add_records_n(std::integral_constant<int, NOUT2>{}, src, n, sum);
    else
      add_records_n(std::integral_constant<int, NOUT>{}, src, n, sum);
  };
  auto put_record = [&](double* dst, const double (&sum)[NOUT2]) {
#pragma unroll
    for (int l = 0; l < NOUT2; ++l)
      if (l < NOUT || twin) st_dev(dst + l * 64 + lane, sum[l]);
  };

  const int g1 = (int)(patch / ENS_ATOMS_G1);
  const int k0 = g1 * ENS_ATOMS_G1;
  const int gn = npatch - k0 < ENS_ATOMS_G1 ? (int)(npatch - k0) : ENS_ATOMS_G1;
  uint32_t* const cnt1 = e.counters + cell * e.ng1 + g1;
  const bool last1 = last_of(cnt1, (uint32_t)gn);
  WBX_EA_STAMP(3);
  if (!last1) return;

  // ---- level 1: atoms -> bins for the gn patches of the group.  Their tables and membership words go through the idle
  // staging buffer: ltab[p][row][l], wl[p][k]; every load of the group is asked for before any of them is used.
  double* const ltab = reinterpret_cast<double*>(lds_raw);
  unsigned long long* const wl = reinterpret_cast<unsigned long long*>(lds_raw + ENS_ATOMS_G1 * TR * NQ * sizeof(double));
  int nwv[ENS_ATOMS_G1];
#pragma unroll
  for (int p = 0; p < ENS_ATOMS_G1; ++p) nwv[p] = p < gn ? ((const_ptr<int32_t>)g.nwords)[bk * npatch + k0 + p] : 0;
  bool overflowed = false;
  {
    constexpr int NJ = (TR * NQ + 63) / 64;
    double v[ENS_ATOMS_G1][NJ];
    unsigned long long wv[ENS_ATOMS_G1];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(e.tab + (cell * npatch + k0) * (TR * NQ), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int p = 0; p < ENS_ATOMS_G1; ++p) {
      overflowed = overflowed || nwv[p] < 0;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int i = lane + 64 * jj;  // (rows past the patch's atoms -- never cleared -- are not used below)
        const bool mine = i < TR * NQ && p < gn && (twin || i < ATOM_MAX * NQ);
        v[p][jj] = ld_sc1(rs, (uint32_t)((p * (TR * NQ) + (mine ? i : 0)) * 8));
      }
      wv[p] = (p < gn && lane < ATOM_MAX) ? g.words[(bk * npatch + k0 + p) * ATOM_MAX + lane] : 0ull;
    }
    wave_lds_order();
#pragma unroll
    for (int p = 0; p < ENS_ATOMS_G1; ++p) {
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj)
        if (lane + 64 * jj < TR * NQ) ltab[p * (TR * NQ) + lane + 64 * jj] = v[p][jj];
      if (lane < ATOM_MAX) wl[p * ATOM_MAX + lane] = wv[p];
    }
    wave_lds_order();
  }
  const double inv_m = 1.0 / (double)M;
  double sum[NOUT2];
#pragma unroll
  for (int l = 0; l < NOUT2; ++l) sum[l] = 0.0;
#pragma unroll
  for (int p = 0; p < ENS_ATOMS_G1; ++p) {
    for (int k = 0; k < nwv[p]; ++k) {
      const double* row = ltab + (p * TR + k) * NQ;
      // membership of the lane's bin as 0.0 / 1.0: NaN * 0 = NaN, so a non-finite sum reaches every bin of its statistic
      // like in the reference's xr.dot (aggregation.py:272-277)
      const double f = ((wl[p * ATOM_MAX + k] >> lane) & 1ull) ? 1.0 : 0.0;
      // output lanes: 0 skill, 1 spread, 2 variance, 3 = (4) - (2) / M, 4 squared error of the mean, 5 count
      double t0 = row[0], t1 = row[1], t2 = row[2], t4 = row[3], t5 = row[4];
      sum[0] = fma(t0, f, sum[0]);
      sum[1] = fma(t1, f, sum[1]);
      sum[2] = fma(t2, f, sum[2]);
      sum[3] = fma(t4 - t2 * inv_m, f, sum[3]);
      sum[4] = fma(t4, f, sum[4]);
      sum[5] = fma(t5, f, sum[5]);
      if (twin) {  // the same over ALL points of the atom: its own rows + those of its twin (what the mask left out)
        const double* rt = row + ATOM_MAX * NQ;
        t0 += rt[0], t1 += rt[1], t2 += rt[2], t4 += rt[3], t5 += rt[4];
        sum[6] = fma(t0, f, sum[6]);
        sum[7] = fma(t1, f, sum[7]);
        sum[8] = fma(t2, f, sum[8]);
        sum[9] = fma(t4 - t2 * inv_m, f, sum[9]);
        sum[10] = fma(t4, f, sum[10]);
        sum[11] = fma(t5, f, sum[11]);
      }
    }
  }
  if (overflowed) {
#pragma unroll
    for (int l = 0; l < NOUT2; ++l) sum[l] = __builtin_nan("");
  }
  put_record(e.part1 + (cell * e.ng1 + g1) * NP, sum);
  WBX_EA_STAMP(4);

  // ---- level 2
  const int g2 = g1 / ENS_ATOMS_G2;
  const int q0 = g2 * ENS_ATOMS_G2;
  const int qn = e.ng1 - q0 < ENS_ATOMS_G2 ? e.ng1 - q0 : ENS_ATOMS_G2;
  uint32_t* const cnt2 = e.counters + (int64_t)g.ncell * e.ng1 + cell * e.ng2 + g2;
  const bool last2 = last_of(cnt2, (uint32_t)qn);
  WBX_EA_STAMP(5);
  if (!last2) return;
  add_records(e.part1 + (cell * e.ng1 + q0) * NP, qn, sum);
  put_record(e.part2 + (cell * e.ng2 + g2) * NP, sum);
  WBX_EA_STAMP(6);

  // ---- level 3: the cell's result
  uint32_t* const cnt3 = e.counters + (int64_t)g.ncell * (e.ng1 + e.ng2) + cell;
  if (!last_of(cnt3, (uint32_t)e.ng2)) return;
  add_records(e.part2 + cell * e.ng2 * NP, e.ng2, sum);
  if (lane < g.nbin) {
    // (one lane per output element: with `accumulate` it reads, adds and writes the element of the caller's accumulator itself)
    const bool add = e.accumulate != 0;
    auto put = [&](int64_t i, double v) { e.out[i] = add ? e.out[i] + v : v; };
    if constexpr (!SKIPNA) {
      const int nout = twin ? NOUT2 : NOUT;
#pragma unroll
      for (int l = 0; l < NOUT2; ++l)
        if (l < nout) put((cell * nout + l) * g.nbin + lane, sum[l]);
    } else {
      // five values, then their five counts (the layout of every skipna reduction in this library); sums 0-5 are over the atom
      // rows (targets valid), 6-11 over atom + twin rows
      const double nan = __builtin_nan("");
      if (e.out_mode == 2) {
        // no twin output: the member-only statistics of THIS group are the sums over atoms + twins (NaN-target points included)
        const double v[10] = {sum[0], sum[7], sum[8], sum[3], sum[4], sum[5], sum[11], sum[11], sum[5], sum[5]};
#pragma unroll
        for (int l = 0; l < 10; ++l) put((cell * 10 + l) * g.nbin + lane, v[l]);
      } else {
        // twin output: the twins also hold the masked-out points, so the masked spread / variance (valid mask AND valid members,
        // whatever the target) is not among the sums: NaN, and nobody asks for it -- the member-only statistics of such a variable
        // carry no mask and read the second ten; there the target statistics over all points are not formed either
        const double v[20] = {sum[0], nan, nan, sum[3], sum[4], sum[5], sum[5], sum[5], sum[5], sum[5],
                              nan, sum[7], sum[8], nan, nan, sum[11], sum[11], sum[11], sum[11], sum[11]};
#pragma unroll
        for (int l = 0; l < 20; ++l) put((cell * 20 + l) * g.nbin + lane, v[l]);
      }
    }
  }
  WBX_EA_STAMP(7);
}

// PERSIST (r5; one-wave blocks): the launch is as many waves as the device holds at once (or fewer), and a wave that has
// finished a patch draws the next one from a queue instead of exiting and leaving its slot to a new wave (2 800 of 3 072 slots
// were occupied in the middle of a launch, profiles/r04_ens_atoms_phases.txt).  64 queues = (XCD y, eighth s of its patches):
// ticket t of queue (y, s) is block 8 (8 t + s) + y of the plain launch, so an XCD still walks its contiguous eighth of the patches
// in order; a wave's home queue is blockIdx & 63.  (One queue per XCD was 2 x SLOWER than the plain launch: atomics on one address
// are served one after the other, ~100 ns each -- 5 000 draws per address and a departures counter that every wave hit once.
// Now ~280 draws per address.)  A wave whose queue has run dry reads all 64 tickets with ONE load (lane = queue), prefers what is
// left on its own XCD, and leaves when every queue is dry.  The tickets are not reset at the end (no departures counter): the
// launches of a context alternate between two sets, and every launch clears the other one.
template <int MP, bool EXACT, bool NT, int MODE = 0, bool PERSIST = false>
__global__ void __launch_bounds__(64 * (NT ? 1 : WBX_ENS_ATOMS_RAGGED_WPB), WBX_ENS_PIPE_WAVES)
ens_atoms_kernel(S1Args a, BinnedArgs g, EnsAtomsArgs e) {
  constexpr int WPB = NT ? 1 : WBX_ENS_ATOMS_RAGGED_WPB;
  constexpr int NLDS = MP < WBX_ENS_PIPE_NLDS ? MP : WBX_ENS_PIPE_NLDS;
  constexpr int NST = NLDS < 48 ? 48 : NLDS;  // (the level-1 finisher borrows 12 KB of it)
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[WPB][NST * 256];
  const int wave_in_block = WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  unsigned char* const lds_raw = lds_all[wave_in_block];  // (every wave works in its own slice: no block-level sync anywhere)
  if constexpr (!PERSIST) {
    ens_atoms_patch<MP, EXACT, NT, MODE>(a, g, e, blockIdx.x, lds_raw);
  } else {
    static_assert(!PERSIST || WPB == 1, "persistent waves are one-wave blocks");
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0) st_dev(e.queue_next + lane, 0u);
    const uint32_t per_xcd = patch_per_xcd<WPB>(g);
    const uint32_t home = blockIdx.x & 63u;
    uint32_t q = home;
    uint32_t static_round = 0;
    const uint32_t mine_n = (per_xcd + 7u - ((uint32_t)lane >> 3)) >> 3;  // tickets of queue `lane`
#pragma unroll 1
    for (;;) {
      uint32_t t = 0;
      if (e.queue_static) {  // (A/B) no queue: block b takes the tickets b / 64, b / 64 + gridDim / 64, .. of queue b & 63
        t = (blockIdx.x >> 6) + static_round * (gridDim.x >> 6);
        ++static_round;
        if (t >= ((per_xcd + 7u - (q >> 3)) >> 3)) break;
      } else {
        if (lane == 0) t = __hip_atomic_fetch_add(e.queue + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
      }
      if (t < ((per_xcd + 7u - (q >> 3)) >> 3)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging slice is this patch's now
        // The patch reads its arguments from the kernel-argument segment again (scalar loads through a pointer the compiler
        // cannot see through): hoisted out of this loop, everything the prologue and the finish levels derive from them stays
        // live across the sweep -- 154 spilled scalar and 21 spilled vector registers against 63 / 0 of the one-patch kernel.
        const_ptr<char> kp = (const_ptr<char>)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        constexpr size_t off_g = (sizeof(S1Args) + alignof(BinnedArgs) - 1) / alignof(BinnedArgs) * alignof(BinnedArgs);
        [[maybe_unused]] constexpr size_t off_e = (off_g + sizeof(BinnedArgs) + alignof(EnsAtomsArgs) - 1) / alignof(EnsAtomsArgs) * alignof(EnsAtomsArgs);
#if WBX_EA_PERSIST_RELOAD
        S1Args a2;
        BinnedArgs g2;
        EnsAtomsArgs e2;
        load_args(a2, (const_ptr<S1Args>)kp);
        load_args(g2, (const_ptr<BinnedArgs>)(kp + off_g));
        load_args(e2, (const_ptr<EnsAtomsArgs>)(kp + off_e));
        ens_atoms_patch<MP, EXACT, NT, MODE>(a2, g2, e2, ((t << 3) + (q >> 3)) * 8u + (q & 7u), lds_raw);
#else
        ens_atoms_patch<MP, EXACT, NT, MODE>(a, g, e, ((t << 3) + (q >> 3)) * 8u + (q & 7u), lds_raw);
#endif
        continue;
      }
      // dry: what is left anywhere?  (a queue seen non-empty may be dry by the time of the draw: then once more)
      const uint32_t seen = ld_dev(e.queue + lane);
      const unsigned long long left = __builtin_amdgcn_ballot_w64(seen < mine_n);
      if (!left) break;
      const unsigned long long near = left & (0x0101010101010101ull << (home & 7u));
      q = (uint32_t)__builtin_ctzll(near ? near : left);
    }
  }
}
#pragma clang diagnostic pop

// What wbx_ens_binned was called with, besides the plan and the inputs in S1Args.
struct EnsBinnedCall {
  const double* wt;      // factored weights (see w_on_x) or NULL
  const uint64_t* bits;  // [nBk][nBr][nj]
  int64_t nA, nBk, nBr, nj;
  int32_t nbin, w_on_x;
  const void* prepared;  // atom tables of wbx_ens_binned_atoms, or NULL (computed inside the call)
  double* out;           // [nA][nBk][ENS_ATOMS_NOUT (twin mode: NOUT2)][nbin]
};

// WBX_ENS_ATOMS_TAPER=0: uniform row splits (A/B timing); default: the tapered table of patch_taper
inline bool ens_atoms_taper() {
  static const bool on = !(getenv("WBX_ENS_ATOMS_TAPER") && atoi(getenv("WBX_ENS_ATOMS_TAPER")) == 0);
  return on;
}

inline int64_t ens_atoms_rows() {
  static const int64_t rows = getenv("WBX_ENS_ATOMS_ROWS") && atol(getenv("WBX_ENS_ATOMS_ROWS")) > 0 ? atol(getenv("WBX_ENS_ATOMS_ROWS")) : WBX_ENS_ATOMS_ROWS;
  return rows;
}

// Counters of the in-kernel sums (EnsAtomsArgs): zero when allocated, and every launch leaves them zero.
inline int ens_atoms_counters(wbx_ctx* ctx, size_t n, uint32_t** out) {
  if (ctx->patch_counters_size < n) {
    if (ctx->patch_counters) {
      WBX_HIP(hipStreamSynchronize(ctx->stream));
      WBX_HIP(hipFree(ctx->patch_counters));
      ctx->patch_counters = nullptr;
      ctx->patch_counters_size = 0;
    }
    const size_t cap = n < 4096 ? 4096 : n * 2;
    WBX_HIP(hipMalloc(&ctx->patch_counters, cap * sizeof(uint32_t)));
    WBX_HIP(hipMemsetAsync(ctx->patch_counters, 0, cap * sizeof(uint32_t), ctx->stream));
    ctx->patch_counters_size = cap;
  }
  *out = reinterpret_cast<uint32_t*>(ctx->patch_counters);
  return 0;
}

template <int MP, bool EXACT>
int launch_ens_atoms(wbx_ctx* ctx, const wbx_s1_plan* plan, S1Args& a, const EnsBinnedCall& c) {
  BinnedArgs g;
  double* extra = nullptr;
  const int64_t cells = c.nA * c.nBk;
  // (the scratch is sized for the geometry patch_setup is about to choose: same call as inside it)
  BinnedArgs probe;
  patch_geometry(probe, cells, c.nBk, c.nBr, c.nj, plan->ndepth, plan->nx, ens_atoms_rows(), ens_atoms_taper());
  const int64_t npatch = (int64_t)probe.nrs * probe.nxt;
  const int ng1 = (int)((npatch + ENS_ATOMS_G1 - 1) / ENS_ATOMS_G1), ng2 = (ng1 + ENS_ATOMS_G2 - 1) / ENS_ATOMS_G2;
  const bool masked = (plan->flags & WBX_FLAG_MASKED) != 0, skipna = (plan->flags & WBX_FLAG_SKIPNA) != 0;
  const bool twin_out = masked && (c.w_on_x & WBX_BINNED_TWIN_MASK);  // the caller wants the sums over all points too
  const bool twin = twin_out || skipna;                                // the twin half of the atom tables is in use
  const size_t NP = (size_t)(twin ? ENS_ATOMS_NOUT2 : ENS_ATOMS_NOUT) * 64;
  const size_t n_tab = (size_t)probe.nblocks * ENS_ATOMS_ROWS2 * ENS_ATOMS_NQ, n_p1 = (size_t)cells * ng1 * NP, n_p2 = (size_t)cells * ng2 * NP;
  // (nacc = 0: no per-patch bin tables -- the sums over patches happen inside the kernel)
  if (int rc = patch_setup(ctx, g, nullptr, c.bits, cells, c.nBk, c.nBr, c.nj, plan->ndepth, plan->nx, 0, c.nbin, true, c.prepared,
                           true, ens_atoms_rows(), n_tab + n_p1 + n_p2, &extra, ens_atoms_taper()))
    return rc;
  EnsAtomsArgs e;
  e.wx = (c.w_on_x & WBX_BINNED_WT_X_ONLY) ? c.wt : nullptr;
  e.wrow = (c.w_on_x & WBX_BINNED_WT_ROW_ONLY) ? c.wt : nullptr;
  e.tab = extra;
  e.part1 = extra + n_tab;
  e.part2 = e.part1 + n_p1;
  e.out = c.out;
  e.ng1 = ng1;
  e.ng2 = ng2;
  if (int rc = ens_atoms_counters(ctx, (size_t)cells * (ng1 + ng2 + 1) + 128, &e.counters)) return rc;
  e.queue = e.counters;  // (the two ticket sets sit at the head of the buffer: the same addresses whatever the geometry)
  e.queue_next = e.counters + 64;
  static const int static_env = getenv("WBX_ENS_ATOMS_STATIC") ? atoi(getenv("WBX_ENS_ATOMS_STATIC")) : 0;
  e.queue_static = static_env;
  e.counters += 128;
  e.masked = 0;
  e.twin_rows = twin ? 1 : 0;
  e.out_mode = skipna ? (twin_out ? 3 : 2) : (twin_out ? 1 : 0);
  e.accumulate = (c.w_on_x & WBX_BINNED_ACCUMULATE) ? 1 : 0;
  e.id_cell_rows = 0;
  e.br_per_split = g.rows_per_split / plan->ndepth;
  if (masked) {
    if (c.w_on_x & WBX_BINNED_MASK_ON_W) {
      if (int rc = merge_mask_into_atom_ids(ctx, a, g, twin_out)) return rc;
    } else {  // strides along A / the depth dims: one id byte per point of the chunk
      if (int rc = merge_point_mask_into_atom_ids(ctx, a, g, plan->ndepth, twin_out)) return rc;
      e.id_cell_rows = c.nBr * plan->ndepth;
    }
    e.masked = twin_out ? 2 : 1;
  }
  static const int nt_env = getenv("WBX_ENS_ATOMS_NT") ? atoi(getenv("WBX_ENS_ATOMS_NT")) : -1;
  static const int order_env = getenv("WBX_PATCH_ORDER") ? atoi(getenv("WBX_PATCH_ORDER")) : -1;
  // rows that are not whole 128-byte lines (721 latitudes): neighbouring x tiles share their boundary lines -- x tile fastest
  // block order and no non-temporal hint, so that the second request of a line finds it in L2 (see det_atoms_kernel)
  const bool ragged_lines = (plan->nx * plan->xstride[0] * 4) % 128 != 0 || plan->xstride[0] != 1;
  const bool nt = nt_env >= 0 ? nt_env != 0 : !ragged_lines;
  g.order = order_env >= 0 ? order_env : (ragged_lines ? 1 : 0);
  const int64_t grid = nt ? patch_grid<1>(g) : patch_grid<WBX_ENS_ATOMS_RAGGED_WPB>(g);
  e.prof = nullptr;
#if WBX_EA_PROF
  const size_t prof_bytes = (size_t)g.nblocks * 16 * sizeof(unsigned long long);
  WBX_HIP(hipMalloc(reinterpret_cast<void**>(&e.prof), prof_bytes));
  WBX_HIP(hipMemsetAsync(e.prof, 0, prof_bytes, ctx->stream));
#endif
  const int mode = skipna ? 2 : (e.id_cell_rows ? 1 : 0);
  const dim3 block(nt ? 64 : 64 * WBX_ENS_ATOMS_RAGGED_WPB);
#define WBX_EA_LAUNCH(NTV, MODEV) hipLaunchKernelGGL((ens_atoms_kernel<MP, EXACT, NTV, MODEV>), dim3((unsigned)grid), block, 0, ctx->stream, a, g, e)
#define WBX_EA_LAUNCH_P(MODEV) hipLaunchKernelGGL((ens_atoms_kernel<MP, EXACT, true, MODEV, true>), dim3((unsigned)pgrid), block, 0, ctx->stream, a, g, e)
#if WBX_EA_PERSIST
  // (diagnostic build, make ab-eapersist) one-wave blocks: persistent waves, as many as the device holds at once
  // (WBX_ENS_ATOMS_PERSIST=0: one block per patch; = N > 1: N waves; WBX_ENS_ATOMS_STATIC=1: fixed shares instead of tickets)
  static const int persist_env = getenv("WBX_ENS_ATOMS_PERSIST") ? atoi(getenv("WBX_ENS_ATOMS_PERSIST")) : 1;
  const int64_t slots = (int64_t)ctx->num_cus * 4 * WBX_ENS_PIPE_WAVES / 8 * 8;
  const bool persist = nt && persist_env != 0 && slots >= 8;
  const int64_t pgrid = persist_env > 1 ? (grid < (int64_t)persist_env ? grid : (int64_t)persist_env / 8 * 8) : (grid < slots ? grid : slots);
  if (persist) {
    if (ctx->ens_queue_parity) {
      uint32_t* const t = e.queue;
      e.queue = e.queue_next;
      e.queue_next = t;
    }
    ctx->ens_queue_parity ^= 1u;
    if (mode == 0) WBX_EA_LAUNCH_P(0);
    else if (mode == 1) WBX_EA_LAUNCH_P(1);
    else WBX_EA_LAUNCH_P(2);
  } else
#endif
  if (nt) {
    if (mode == 0) WBX_EA_LAUNCH(true, 0);
    else if (mode == 1) WBX_EA_LAUNCH(true, 1);
    else WBX_EA_LAUNCH(true, 2);
  } else {
    if (mode == 0) WBX_EA_LAUNCH(false, 0);
    else if (mode == 1) WBX_EA_LAUNCH(false, 1);
    else WBX_EA_LAUNCH(false, 2);
  }
#undef WBX_EA_LAUNCH
#undef WBX_EA_LAUNCH_P
  WBX_HIP(hipGetLastError());
#if WBX_EA_PROF
  {
    WBX_HIP(hipStreamSynchronize(ctx->stream));
    unsigned long long* host = (unsigned long long*)malloc(prof_bytes);
    WBX_HIP(hipMemcpy(host, e.prof, prof_bytes, hipMemcpyDeviceToHost));
    if (const char* path = getenv("WBX_EA_PROF_DUMP")) {
      if (FILE* f = fopen(path, "wb")) {
        const long long hdr[4] = {(long long)g.ncell, (long long)g.nrs, (long long)g.nxt, 16};
        fwrite(hdr, sizeof(hdr), 1, f);
        fwrite(host, 1, prof_bytes, f);
        fclose(f);
      }
    }
    free(host);
    WBX_HIP(hipFree(e.prof));
  }
#endif
  return 0;
}

// one translation unit per bucket (parallel build)
int launch_ens_atoms_m4(wbx_ctx*, const wbx_s1_plan*, S1Args&, const EnsBinnedCall&);
int launch_ens_atoms_m8(wbx_ctx*, const wbx_s1_plan*, S1Args&, const EnsBinnedCall&);
int launch_ens_atoms_m16(wbx_ctx*, const wbx_s1_plan*, S1Args&, const EnsBinnedCall&);
int launch_ens_atoms_m32(wbx_ctx*, const wbx_s1_plan*, S1Args&, const EnsBinnedCall&);
int launch_ens_atoms_m64(wbx_ctx*, const wbx_s1_plan*, S1Args&, const EnsBinnedCall&);
int launch_ens_atoms_m50(wbx_ctx*, const wbx_s1_plan*, S1Args&, const EnsBinnedCall&);
int launch_ens_atoms_m51(wbx_ctx*, const wbx_s1_plan*, S1Args&, const EnsBinnedCall&);

}  // namespace wbx
