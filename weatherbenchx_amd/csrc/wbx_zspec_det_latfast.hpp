// Zonal spectra of predictions AND targets plus the deterministic lanes of the same (p, t[, c]) rows in ONE sweep over
// LATITUDE-FASTEST fields (the public ERA5 / WeatherBench layout [.., longitude, latitude]: the rows of a slab are adjacent
// floats, longitude is strided) -- the counterpart of zspec1440_det_kernel (wbx_zspec_det.hpp) for the layout real archives
// have.  Before, such a chunk crossed the HBM as three launches at 20 B/point (wbx_det_partial on p, t, c + one staged-run
// spectrum launch per field); here p, t and c are read once: 12 B/point (8 without a climatology).
//
// A block of ZL_TEAMS = 8 one-wave teams takes a run of <= 8 adjacent rows of one slab.  Its 512 threads are LOADERS first:
// thread -> (row pair q of the run, longitudes j0 + 128 n): one 8-byte load per field delivers the values of rows 2q and
// 2q + 1 at one longitude, fetched one step ahead into registers (spread behind the passes of the current step like
// zspec1440_latfast_kernel's).  At the start of the next step a loader
//   * accumulates the DET3 / DET6 lanes of its 24 points in fp64 -- the arithmetic of DetOp (wbx_det.hip) -- per row; every
//     fourth lane of a DPP row holds the same row pair, so two row rotations leave 12 sums per pair and DPP row in a 12 KB LDS
//     table and 8 x NA threads add the 32 entries of a row in a fixed order: the row's entry of the stage-1 partial buffer
//     (one key = one row, stage 2 unchanged), no atomics;
//   * stores (p, t) of row 2q at longitude j into team 2q's buffer and (p, t) of row 2q + 1 into team 2q + 1's: element j
//     of a team's buffer is the 8 bytes (p[j], t[j]) = one half of the packed point j / 2 of the row PAIR (p row, t row),
//     already the pass-1 layout of z14_pair<.., PT = true>.
// Then the teams run the three passes; the spectra of row A go to the predictions' sums, those of row B to the targets'
// (fp64 registers per team -> two block tables in the LDS when the team's group changes -> one global atomic per
// wavenumber when the block's group changes, as in the spectrum-only kernel).
// Eight teams = two waves per SIMD and up to 256 VGPRs, like the longitude-fastest fused kernel (whose throughput is flat
// from 8 to 12 teams per CU); 32-byte segments per longitude and field: the four runs that share a 128-byte line go to
// neighbouring blocks of ONE XCD in the same step (the block schedule of zspec1440_latfast_kernel).
//
// Rows = the keys of the deterministic plan (x = longitude summed, nx = 1440, ndepth = 1, nchunk = 1); key o * rps + r is
// row r of slab o, and the rows of a slab are adjacent elements of every input (the caller's plan: wbx_det_spectrum_slabs).
// Included by wbx_spectrum.hip (inside namespace wbx, after wbx_zspec_det.hpp).
#pragma once

constexpr int ZL_TEAMS = 8;                                // teams = rows of a run
constexpr int ZL_THREADS = 64 * ZL_TEAMS;
constexpr int ZL_ITEMS = 12;                               // longitudes per loader: j0 + 128 n (n = 11: j0 < 32 only)
constexpr int ZL_GROUPS = ZL_THREADS / 16;                 // DPP rows of the block: 32, each with four lanes per row pair
#ifndef WBX_ZL_BUFL
#define WBX_ZL_BUFL 732                                    // v4 elements between the team buffers (>= Z14_BUF; = 4 mod 8: the staging stores of a wave -- four row pairs x 16 longitudes -- then fall on all 64 banks twice)
#endif

// KNOCK (diagnostic instantiations, wrong results; WBX_ZL_KNOCK): 1 = no deterministic arithmetic, 2 = no passes, 4 = no global
// loads, 8 = no staging stores
template <bool HAS_C, int KNOCK = 0>
__global__ void __launch_bounds__(ZL_THREADS) zspec1440_det_latfast_kernel(
    S1Args a, int64_t rps, int64_t nslab, int64_t slabs_per_xcd, int runs_per_slab, int run_base, int run_rem,
    const float2* __restrict__ tables_g, const int32_t* __restrict__ group, const double* __restrict__ scale, SpecRecs recs) {
  __shared__ int team_in_table[ZL_TEAMS];   // the team's sums of the last step sit in its buffer, for the block's tables
  constexpr int NA = HAS_C ? 6 : 3;
  constexpr int NIN = HAS_C ? 3 : 2;
  constexpr int BUFL = WBX_ZL_BUFL;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  float2* const tw1 = reinterpret_cast<float2*>(lds_raw);
  float2* const tw2 = tw1 + Z14_TW1;
  float2* const twr = tw2 + Z14_TW2;
  v4* const bufs = reinterpret_cast<v4*>(twr + Z14_TWR);
  constexpr int nk = Z14_N2 + 1;
  double* const blkp = reinterpret_cast<double*>(bufs + ZL_TEAMS * BUFL);
  double* const blkt = blkp + nk + 1;
  double* const dsum = blkt + nk + 1;  // [ZL_GROUPS][4 row pairs][2 rows][NA]
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int team = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the younger wave of a SIMD (teams 4-7) gets the higher user priority: both reach the step's barriers together
  if (team >> 2) __builtin_amdgcn_s_setprio(1);
  v4* const buf = bufs + team * BUFL;
  for (int i = tid; i < Z14_TABLES; i += ZL_THREADS) tw1[i] = tables_g[i];
  for (int k = tid; k < 2 * (nk + 1); k += ZL_THREADS) blkp[k] = 0.0;
  const Z14Lane c = z14_lane(lane, buf, tw2);
  const int L = c.L;
  const double quarter_inv_nn = 0.25 / ((double)Z14_N * (double)Z14_N);

  // the block's steps (see zspec1440_latfast_kernel): XCD x owns a contiguous eighth of the slabs; its (slab, run) pairs,
  // slab-major, are dealt out to its blocks round-robin
  const int xcd = (int)(blockIdx.x & 7u), local = (int)(blockIdx.x >> 3), nlocal = (int)(gridDim.x >> 3);
  const int64_t o_end = (xcd + 1) * slabs_per_xcd < nslab ? (xcd + 1) * slabs_per_xcd : nslab;
  int64_t o = xcd * slabs_per_xcd;
  int run = local;
  auto normalise = [&](int64_t& oo, int& rr) {
    while (rr >= runs_per_slab) {
      rr -= runs_per_slab;
      oo += 1;
    }
  };
  normalise(o, run);
  auto run_rows = [&](int rr, int64_t& rbeg, int64_t& rend) {  // the first `rem` runs of a slab are one row longer
    rbeg = (int64_t)rr * run_base + (rr < run_rem ? rr : run_rem);
    rend = rbeg + run_base + (rr < run_rem ? 1 : 0);
    rend = rend < rps ? rend : rps;
  };

  // loader role: four consecutive lanes = the four row pairs of ONE longitude (32 adjacent bytes: a quad of lanes asks for one
  // line), a wave covers 16 consecutive longitudes
  const int grp = tid >> 4, q = tid & 3, j0 = tid >> 2;
  v2* const dst0 = reinterpret_cast<v2*>(bufs + (2 * q) * BUFL) + j0;
  v2* const dst1 = reinterpret_cast<v2*>(bufs + (2 * q + 1) * BUFL) + j0;
  v2 hp[ZL_ITEMS], ht[ZL_ITEMS], hc[HAS_C ? ZL_ITEMS : 1];
#pragma unroll
  for (int n = 0; n < ZL_ITEMS; ++n) {
    hp[n] = ht[n] = (v2){0.f, 0.f};
    if constexpr (HAS_C) hc[n] = (v2){0.f, 0.f};
  }
  // the next run's row-pair bases (elements), resolved through the plan's tables one step ahead; rows_next: 0, 1 or 2 rows
  const float *sp = nullptr, *st = nullptr, *sc_ = nullptr;
  int rows_next = 0, rows_held = 0;
  auto resolve = [&](int64_t oo, int rr) {
    int64_t rbeg, rend;
    run_rows(rr, rbeg, rend);
    const int64_t ra = rbeg + 2 * q;
    rows_next = ra + 1 < rend ? 2 : (ra < rend ? 1 : 0);
    if (rows_next) {
      const int64_t key = oo * rps + ra;
      int64_t kb[WBX_MAX_INPUTS], ro[WBX_MAX_INPUTS];
      key_bases<NIN>(a, key, kb);
      row_bases<NIN>(a, kb, key, 0, ro);
      sp = reinterpret_cast<const float*>(a.in[0]) + ro[0] + (int64_t)j0 * a.xstride[0];
      st = reinterpret_cast<const float*>(a.in[1]) + ro[1] + (int64_t)j0 * a.xstride[1];
      if constexpr (HAS_C) sc_ = reinterpret_cast<const float*>(a.in[2]) + ro[2] + (int64_t)j0 * a.xstride[2];
    }
  };
  // (no non-temporal hint: the other three runs of a 128-byte line want it from L2)
  auto load_part = [&](int part) {  // items [3 part, 3 part + 3)
    if constexpr (KNOCK & 4) return;
    if (rows_next == 2) {
#pragma unroll
      for (int n = 0; n < ZL_ITEMS; ++n)
        if (n / 3 == part && (n < ZL_ITEMS - 1 || j0 < Z14_N - 128 * (ZL_ITEMS - 1))) {
          hp[n] = *reinterpret_cast<const v2u*>(sp + (int64_t)(128 * n) * a.xstride[0]);
          ht[n] = *reinterpret_cast<const v2u*>(st + (int64_t)(128 * n) * a.xstride[1]);
          if constexpr (HAS_C) hc[n] = *reinterpret_cast<const v2u*>(sc_ + (int64_t)(128 * n) * a.xstride[2]);
        }
    } else if (rows_next == 1) {  // a lone last row: nothing may be read behind it
#pragma unroll
      for (int n = 0; n < ZL_ITEMS; ++n)
        if (n / 3 == part && (n < ZL_ITEMS - 1 || j0 < Z14_N - 128 * (ZL_ITEMS - 1))) {
          hp[n] = (v2){sp[(int64_t)(128 * n) * a.xstride[0]], 0.f};
          ht[n] = (v2){st[(int64_t)(128 * n) * a.xstride[1]], 0.f};
          if constexpr (HAS_C) hc[n] = (v2){sc_[(int64_t)(128 * n) * a.xstride[2]], 0.f};
        }
    } else {  // (a short run: zeros add nothing to the sums below)
#pragma unroll
      for (int n = 0; n < ZL_ITEMS; ++n)
        if (n / 3 == part) {
          hp[n] = ht[n] = (v2){0.f, 0.f};
          if constexpr (HAS_C) hc[n] = (v2){0.f, 0.f};
        }
    }
  };

  double accp[6], accmp[6], acct[6], accmt[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) accp[s] = accmp[s] = acct[s] = accmt[s] = 0.0;
  // (r5: no atomics, fixed order -- see zspec1440_latfast_kernel)  A team's sums of ONE step go into its own staging buffer (2 x 721
  // doubles: the predictions', then the targets') when its passes are done; behind the next barrier every thread adds the
  // teams' buffers, team 0 first, into the block's two tables; the tables go out as ONE record of 2 x 721 values when the
  // step's group changes.  A team whose row is of another group than the step's writes a record of its own.
  int32_t blk_group = -1;  // block-uniform
  if (tid < ZL_TEAMS) team_in_table[tid] = 0;
  const int64_t block_id = (int64_t)blockIdx.x;
  unsigned int seq_team = 0;
  int64_t table_slot = -1;  // the slot (= step number, o * runs_per_slab + run) of the last step whose sums are in the tables
  double* const own = reinterpret_cast<double*>(buf);
  auto gather_teams = [&]() {
    for (int k = tid; k < 2 * nk; k += ZL_THREADS) {
      double* tab = k >= nk ? blkt + (k - nk) : blkp + k;
      double sum = *tab;
#pragma unroll
      for (int t = 0; t < ZL_TEAMS; ++t)
        if (team_in_table[t]) sum += reinterpret_cast<const double*>(bufs + t * BUFL)[k];
      *tab = sum;
    }
  };
  auto flush_block = [&](int32_t next) {  // every thread of the block, block-uniformly; contains block barriers
    if (blk_group >= 0) {
      double* const rec = spec_rec_static(recs, (unsigned int)table_slot, blk_group, (unsigned long long)table_slot, tid == 0);
      for (int k = tid; k < 2 * nk; k += ZL_THREADS) {
        const bool second = k >= nk;
        const int kk = second ? k - nk : k;
        double* tab = second ? blkt : blkp;
        const double sum = tab[kk];
        rec[k] = kk == 0 ? sum : 2.0 * sum;  // S_k, include/wbx.h
        tab[kk] = 0.0;
      }
    }
    blk_group = next;
  };
  if (o < o_end) {
    resolve(o, run);
#pragma unroll
    for (int part = 0; part < 4; ++part) load_part(part);
  }
  rows_held = rows_next;
  while (o < o_end) {  // block-uniform
    __syncthreads();  // every team is done with its buffer (and, the first time, the tables are in place)
    gather_teams();   // the sums of the step that just ended (in the teams' buffers) -> the block's tables
    __syncthreads();  // ... before the buffers are filled again
    if (tid < ZL_TEAMS) team_in_table[tid] = 0;
    int64_t rbeg, rend;
    run_rows(run, rbeg, rend);
    const int64_t row0 = o * rps;
    const int64_t ra = rbeg + team;
    const bool active = ra < rend;  // team-uniform
    const int32_t g0 = group[row0 + rbeg];  // the step's group (block-uniform)
    int32_t g = g0;
    double sc = 0.0;
    if (active) {
      g = group[row0 + ra];
      sc = scale[row0 + ra] * quarter_inv_nn;
    }
    // the next run's bases: two dependent table lookups (key -> climatology slot -> offset), asked for here so that their
    // latency passes under the arithmetic below instead of in front of the first loads behind pass 1
    int64_t on = o;
    int rn = run + nlocal;
    normalise(on, rn);
    const bool more = on < o_end;
    if (more) resolve(on, rn);
    else rows_next = 0;
    // ---- loaders: the deterministic lanes of the held points, then the staging stores
    {
      double d[2][NA];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int l = 0; l < NA; ++l) d[h][l] = 0.0;
#pragma unroll
      for (int n = 0; n < ZL_ITEMS; ++n) {
        // (an ordering point per item: without it the compiler widens every input up front to feed the twelve fma chains --
        // see zspec1440_det_kernel)
        if constexpr (HAS_C)
          asm volatile("" : "+v"(hp[n]), "+v"(ht[n]), "+v"(hc[n]), "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[0][3]), "+v"(d[0][4]),
                       "+v"(d[0][5]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[1][2]), "+v"(d[1][3]), "+v"(d[1][4]), "+v"(d[1][5]));
        else
          asm volatile("" : "+v"(hp[n]), "+v"(ht[n]), "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[0][2]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[1][2]));
#pragma unroll
        for (int h = 0; h < ((KNOCK & 1) ? 0 : 2); ++h) {
          const double p = (double)(h ? hp[n].y : hp[n].x), t = (double)(h ? ht[n].y : ht[n].x);
          const double e = p - t;
          d[h][0] += e;
          d[h][1] += fabs(e);
          d[h][2] = fma(e, e, d[h][2]);
          if constexpr (HAS_C) {
            const double cv = (double)(h ? hc[n].y : hc[n].x);
            const double ap = p - cv, at = t - cv;
            d[h][3] = fma(ap, ap, d[h][3]);
            d[h][4] = fma(at, at, d[h][4]);
            d[h][5] = fma(ap, at, d[h][5]);
          }
        }
        if (!(KNOCK & 8) && (n < ZL_ITEMS - 1 || j0 < Z14_N - 128 * (ZL_ITEMS - 1))) {
          dst0[128 * n] = (v2){hp[n].x, ht[n].x};
          dst1[128 * n] = (v2){hp[n].y, ht[n].y};
        }
      }
      // lanes q, q + 4, q + 8, q + 12 of a DPP row hold the same row pair: after two rotations each of them has the four's sum
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int l = 0; l < NA; ++l) {
          double v = d[h][l];
          v += dpp_moved<0x124, 0xf>(v);  // row_ror:4
          v += dpp_moved<0x128, 0xf>(v);  // row_ror:8
          if ((h * NA + l) % 4 == ((tid >> 2) & 3)) dsum[((grp * 4 + q) * 2 + h) * NA + l] = v;
        }
    }
    __syncthreads();
    // row i of the run = row i & 1 of pair i >> 1: the 32 DPP rows' sums of that pair, in a fixed order
    if (tid < ZL_TEAMS * NA) {
      const int i = tid / NA, l = tid - i * NA;
      if (rbeg + i < rend) {
        double tot = 0.0;
#pragma unroll 8
        for (int gg = 0; gg < ZL_GROUPS; ++gg) tot += dsum[((gg * 4 + (i >> 1)) * 2 + (i & 1)) * NA + l];
        a.out[(row0 + rbeg + i) * NA + l] = tot;
      }
    }
    // block-uniform: the tables hold the sums of the earlier steps -- of another group: out they go (into the last of those steps'
    // slot); of this step's group: they stay, and that step's slot is marked empty
    if (g0 != blk_group) flush_block(g0);
    else if (table_slot >= 0) spec_rec_static(recs, (unsigned int)table_slot, -1, 0ull, tid == 0);
    table_slot = o * runs_per_slab + run;
    if (active) {
      C2 v[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) v[(6 * i) % 12 + i / 2] = ld_c2(buf + 60 * ((6 * i) % 12 + i / 2) + L);  // 0, 6, 1, 7, ..: z14_demean
      __builtin_amdgcn_wave_barrier();
      const v2 msh = z14_demean<true>(v);
      if constexpr (KNOCK & 2) {
#pragma unroll
        for (int i = 0; i < 6; ++i) accp[i] += (double)(v[i].re.x + v[i + 6].im.y) * sc;
#pragma unroll
        for (int part = 0; part < 4; ++part)
          if (more) load_part(part);
      } else
      z14_pair<0, true>(v, buf, c, tw1, twr, sc, sc, false, g, accp, accmp, nullptr, [&](int i) {
        if (more && (!(i & 1) || i == 5)) load_part(i == 5 ? 3 : i / 2);
      }, acct, accmt, msh);
      // this step's sums leave the registers: into the team's own buffer for the block's tables, or -- a row of another group
      // than the step's -- into a record of its own
      __builtin_amdgcn_wave_barrier();
      if (g == g0) {
        z14_send<false>(own, c, accp, accmp);
        z14_send<false>(own + nk, c, acct, accmt);
        if (lane == 0) team_in_table[team] = 1;
      } else {
        double* const rec = spec_rec_open(recs, g, (1ull << 62) | spec_key(block_id * ZL_TEAMS + team, seq_team++), lane);
        z14_send<true>(rec, c, accp, accmp);
        z14_send<true>(rec + nk, c, acct, accmt);
      }
#pragma unroll
      for (int s = 0; s < 6; ++s) accp[s] = accmp[s] = acct[s] = accmt[s] = 0.0;
    } else if (more) {
#pragma unroll
      for (int part = 0; part < 4; ++part) load_part(part);  // (a team without a row in this run still loads its share of the next one)
    }
    rows_held = rows_next;
    o = on;
    run = rn;
  }
  (void)rows_held;
  __syncthreads();
  gather_teams();  // the last step's sums
  __syncthreads();
  flush_block(-1);
}
