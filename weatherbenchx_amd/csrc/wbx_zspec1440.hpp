// Zonal spectrum of 1440-point rows (0.25 degree grids): ONE WAVE per row pair, 720 = 12 x 5 x 12.
//
// The generic fused kernel (zspec_fused_kernel) runs five Stockham passes (4, 4, 5, 3, 3) over a 256-thread team: five
// LDS round trips, a block barrier between the read and the write half of every pass, 60-94 % of the lanes busy, and the
// butterfly addresses / twiddle indices recomputed per pass (SQ counters of configs[3]: 1150 VALU + 630 SALU + 172 LDS
// instructions per row pair, every unit ~45 % busy, 2500 cycles per pair and CU).  Here the length-720 complex transform
// of the packed row (z[m] = x[2m] + i x[2m+1]) is split as
//     m = 60 a + b          pass 1: lane b (60 of 64 lanes) takes z[60 a + b], a = 0..11, straight from global memory
//                                   (consecutive lanes = consecutive points), a 12-point DFT in registers, then the
//                                   twiddle W720^(b k1)
//     b = 12 c + d          pass 2: butterfly (k1, d) -- 144 of them, 3 per lane on 48 lanes -- a 5-point DFT over c,
//                                   then W60^(d q)
//     k = k1 + 12 q + 60 s  pass 3: lane L = k1 + 12 q (60 lanes) a 12-point DFT over d: Z[L + 60 s], s = 0..11
// so a row crosses the LDS three times (two transposes between the passes and one mirror exchange for the Hermitian
// unpack) instead of five, the 12-point DFTs are prime-factor (3 x 4) butterflies without internal twiddles, every LDS
// address and twiddle index of a lane is a loop-invariant register or an immediate offset, and there is no barrier: a
// wave's LDS instructions execute in order.  As in the generic kernel two rows share every instruction (C2 = the same
// point of rows A and B in one packed-fp32 register pair), the next pair's 24 loads are issued behind pass 1's stores,
// and |X_k|^2 * scale is accumulated in fp64 registers (lane L owns k = L + 60 s, s < 6, and their mirrors 720 - k: one
// Hermitian unpack yields both) and flushed with fp64 atomics when the row group changes.
//
// Included by wbx_spectrum.hip (inside namespace wbx, after C2 / butterfly<R>).
#pragma once

constexpr int Z14_N = 1440, Z14_N2 = 720, Z14_LANES = 60, Z14_LANES5 = 48;
constexpr int Z14_S1 = 61;                 // row stride of the first transpose (odd: the strided side is the READ)
constexpr int Z14_BUF = 12 * Z14_S1;       // v4 elements of LDS per wave (>= 720)
constexpr int Z14_TW1 = 11 * 60, Z14_TW2 = 4 * 12, Z14_TWR = 720;  // float2 entries: W720^(b k1) | W60^(d q) | W1440^k
constexpr int Z14_TABLES = Z14_TW1 + Z14_TW2 + Z14_TWR;

// The transform is fp32: every output carries ~eps |F|_max of its row, and on a physical field |F|_max is the mean (F_0 = n x
// mean: T ~ 280 K against anomalies of a few K, geopotential 5e4 against 1e2..1e3), so the wavenumbers above it were held to
// eps x mean / anomaly instead of eps (measured on N(280, 1) rows: median relative error of S_k 1.0e-5 per row, up to 48 % on
// single coefficients, against 1.4e-7 on N(0, 1) rows).  With WBX_SPECTRUM_DEMEAN a row is shifted by m = an estimate of its
// mean (fp32; any m is valid, it only has to be close) before the transform (z14_demean) -- x - m is exact in fp32 for x
// within a factor two of m -- which changes nothing but F_0 (n is even: the Nyquist term keeps its value), and
// F_0 = F'_0 + n m is put back in fp64 where |X_0|^2 is formed (z14_pair).
// (the macro and the DPP reduction: wbx_spectrum.hip, in front of the generic kernel, whose one-wave teams use them too)

// -> (m_A, m_B) of the rows A and B whose pass-1 inputs v holds; subtracted from v in place.  Every instruction here is
// paid three times per SIMD (its waves run the passes in step), so the estimate takes what is cheap: the real parts of every
// other pass-1 input (x[120 a + 2 b], a even: 5 packed additions) of all 64 lanes -- lanes 60..63 shadow lane 59, whose
// samples therefore count five times: a slightly noisier mean, no select -- i.e. 360 evenly spread longitudes of the row.
// FEW (the latitude-fastest kernel, whose pass-1 inputs are LDS loads still in flight here): only v[0] and v[6] -- two
// antipodal arcs of 32 degrees, 128 samples -- so that the reduction runs under the latency of the other ten loads.
template <bool FEW = false>
__device__ __forceinline__ v2 z14_demean(C2 (&v)[12]) {
  if constexpr (!WBX_SPECTRUM_DEMEAN) return (v2){0.f, 0.f};
  v2 s = v[0].re + v[6].re;
  if constexpr (!FEW) {
#pragma unroll
    for (int a = 2; a < 12; a += 2)
      if (a != 6) s += v[a].re;
  }
  constexpr float inv = FEW ? 1.0f / 128.0f : 1.0f / 384.0f;
  const v2 m = {wave_sum_uniform_f32(s.x) * inv, wave_sum_uniform_f32(s.y) * inv};
#pragma unroll
  for (int a = 0; a < 12; ++a) {
    v[a].re -= m;
    v[a].im -= m;
  }
  return m;
}

// 12-point DFT, Good-Thomas: input n = (4 n1 + 3 n2) mod 12, output k = (4 k1 + 9 k2) mod 12; four 3-point and three
// 4-point butterflies, no twiddles in between.
__device__ __forceinline__ void dft12(C2 (&v)[12]) {
  C2 t[3][4];
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) {
    C2 u[3] = {v[(3 * n2) % 12], v[(4 + 3 * n2) % 12], v[(8 + 3 * n2) % 12]};
    butterfly<3>(u);
    t[0][n2] = u[0];
    t[1][n2] = u[1];
    t[2][n2] = u[2];
  }
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    butterfly<4>(t[k1]);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) v[(4 * k1 + 9 * k2) % 12] = t[k1][k2];
  }
}

struct Z14Lane {  // what a lane keeps for the whole launch: every LDS address / twiddle index is this + an immediate
  int lane, L;        // L = the lane's butterfly in passes 1 and 3 (lanes 60..63 shadow lane 59: same addresses, same values)
  const v4* rd5;      // pass 2 (lanes 48..63 shadow lane 47), butterfly i = 0..2: loads at + 4 i + 12 c
  v4* wr5;            //         stores at + 240 i + 12 q
  const float2* tw5;  //         twiddles at + 4 i + 12 (q - 1)
  int mir0;           // mirror of k = L + 60 s is mir0 - 60 s (s = 0: 0 for L = 0)
  int k0;             // the lane's first wavenumber in the unpack: L, or 360 on lane 60 (see z14_pair)
};

__device__ __forceinline__ Z14Lane z14_lane(int lane, v4* buf, const float2* tw2) {
  Z14Lane c;
  c.lane = lane;
  c.L = lane < Z14_LANES ? lane : Z14_LANES - 1;
  const int l5 = lane < Z14_LANES5 ? lane : Z14_LANES5 - 1;
  const int k1_5 = l5 % 12, d0_5 = l5 / 12;  // butterfly i of pass 2: (k1, d) = (k1_5, d0_5 + 4 i)
  c.rd5 = buf + k1_5 * Z14_S1 + d0_5;
  c.wr5 = buf + d0_5 * 60 + k1_5;
  c.tw5 = tw2 + d0_5;
  c.mir0 = c.L == 0 ? 0 : Z14_N2 - c.L;
  c.k0 = lane == Z14_LANES ? Z14_N2 / 2 : c.L;
  return c;
}

// (KNOCK & 2: the value is computed -- pinned by an empty asm -- but not stored)
template <bool DROP>
__device__ __forceinline__ void z14_store(v4* p, C2 a) {
  if constexpr (DROP)
    asm volatile("" ::"v"(a.re.x), "v"(a.re.y), "v"(a.im.x), "v"(a.im.y));
  else
    st_c2(p, a);
}

// A lane's twelve sums go to k = L + 60 s (acc) and 720 - L - 60 s (accm), s < 6; lane 60 owns k = 360 (acc[0]).
// weight = true applies S_k = |F_k|^2 * (k == 0 ? 1 : 2) (include/wbx.h); the block table of the latitude-fastest kernel
// takes the raw sums.
// (r5) `out` is a RECORD (spec_rec_open) or a table of the caller's own: every wavenumber is written exactly once, with a plain
// store -- the sums over teams happen in spec_close_kernel, in an order that does not depend on who arrives first.
// (k = 0 of lane 0 is written by the first line; its "mirror" 720 is the Nyquist term.)
template <bool WEIGHT>
__device__ __forceinline__ void z14_send(double* out, const Z14Lane& c, const double (&acc)[6], const double (&accm)[6]) {
  if (c.lane < Z14_LANES) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      out[c.L + 60 * s] = (!WEIGHT || (s == 0 && c.L == 0)) ? acc[s] : 2.0 * acc[s];
      out[Z14_N2 - c.L - 60 * s] = WEIGHT ? 2.0 * accm[s] : accm[s];
    }
  } else if (c.lane == Z14_LANES) {
    out[Z14_N2 / 2] = WEIGHT ? 2.0 * acc[0] : acc[0];
  }
}

// One row pair: passes 1-3 on v (the pass-1 inputs), the mirror exchange and the Hermitian unpack; adds scale * |X_k|^2
// into acc (k = L + 60 s, s < 6; lane 60: k = 360 in slot 0) and accm (720 - k) -- row B's values straight into `power`, a
// record of its own, when the pair straddles two groups (`split`).
// at(i) is called at six points: 0 pass 1 has issued its stores | 1 pass 2 has its loads | 2 pass 2 has issued its stores |
// 3 pass 3 has its loads | 4 the mirror stores are issued | 5 done.  The longitude-fastest kernel stamps the clock there
// (PROF), the latitude-fastest one issues a part of the next run's global loads at 0, 2, 4.
// KNOCK (diagnostic instantiations, wrong results): 2 = the exchange stores are dropped, 4 = their loads too (register
// values are passed on), 8 = no unpack arithmetic.
// (`buf` must NOT be __restrict__: c.rd5 / c.wr5 point into it -- with the qualifier the compiler moved pass-2 loads above
// the pass-1 stores.)
// PT (wbx_zspec_det.hpp): rows A and B are the predictions' and the targets' row of ONE location; B's sums go to their own
// accumulators accb / accmb instead of joining A's (no `split` then).
// TW_EARLY (r6; 1 in the fused det + spectra sweep): the lane's twiddles of a stage (and the unpack's mirror partners) are ASKED FOR
// ahead of the arithmetic in front of their use -- left alone the compiler reads them from the LDS just in time, two or three at a
// stretch, each stretch behind its own `s_waitcnt lgkmcnt(0)` (~30 waits per row; with the reads grouped ~12): -1 % at unchanged
// registers (profiles/r06_det_spectrum_twearly_ab.txt).  The first attempt changed the last fp32 bit of the sweep's spectra -- the
// other source shape made the compiler fuse other products with their sums -- until the ambiguous sums were pinned (`prod`,
// wbx_spectrum.hip): with them the plain kernel (TW_EARLY = 0: no registers for the grouping at three waves per SIMD) and the sweep
// return the same bits again, whatever their schedules.
template <int KNOCK, bool PT = false, int TW_EARLY = 0, typename At>
__device__ __forceinline__ void z14_pair(C2 (&v)[12], v4* buf, const Z14Lane& c, const float2* __restrict__ tw1,
                                         const float2* __restrict__ twr, double sca, double scb, bool split, int32_t gb,
                                         double (&acc)[6], double (&accm)[6], double* __restrict__ power, At&& at,
                                         double (&accb)[6], double (&accmb)[6], v2 msh = (v2){0.f, 0.f}) {
  constexpr bool DROP = (KNOCK & 2) != 0;
  const int L = c.L;
  // ---- pass 1: 12-point DFT over a, twiddle W720^(b k1), transpose 1: buf[k1 * S1 + b]
  if constexpr (TW_EARLY) {
    float2 w[11];
#pragma unroll
    for (int k1 = 1; k1 < 12; ++k1) w[k1 - 1] = tw1[(k1 - 1) * 60 + L];
    if constexpr (TW_EARLY == 1) __builtin_amdgcn_sched_barrier(0);  // the eleven reads are issued here ...
    dft12(v);
#pragma unroll
    for (int k1 = 1; k1 < 12; ++k1) v[k1] = ctw(v[k1], w[k1 - 1]);  // ... and waited for once, here
  } else {
    dft12(v);
#pragma unroll
    for (int k1 = 1; k1 < 12; ++k1) v[k1] = ctw(v[k1], tw1[(k1 - 1) * 60 + L]);
  }
#pragma unroll
  for (int k1 = 0; k1 < 12; ++k1) z14_store<DROP>(buf + k1 * Z14_S1 + L, v[k1]);
  __builtin_amdgcn_wave_barrier();
  at(0);
  // ---- pass 2: 5-point DFT over c, twiddle W60^(d q), transpose 2: buf[d * 60 + k1 + 12 q]
  C2 u[3][5];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) u[i][cc] = (KNOCK & 4) ? v[(5 * i + cc) % 12] : ld_c2(c.rd5 + 4 * i + 12 * cc);
  }
  float2 w5[3][4];  // (TW_EARLY) pass 2's twelve twiddles, asked for with its data
  if constexpr (TW_EARLY) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int q = 1; q < 5; ++q) w5[i][q - 1] = c.tw5[4 * i + 12 * (q - 1)];
    }
    if constexpr (TW_EARLY == 1) __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_wave_barrier();  // every load of the first layout precedes the stores of the second
  at(1);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    butterfly<5>(u[i]);
#pragma unroll
    for (int q = 1; q < 5; ++q) u[i][q] = ctw(u[i][q], TW_EARLY ? w5[i][q - 1] : c.tw5[4 * i + 12 * (q - 1)]);
#pragma unroll
    for (int q = 0; q < 5; ++q) z14_store<DROP>(c.wr5 + 240 * i + 12 * q, u[i][q]);
  }
  __builtin_amdgcn_wave_barrier();
  at(2);
  // ---- pass 3: 12-point DFT over d: v[s] = Z[L + 60 s]
#pragma unroll
  for (int d = 0; d < 12; ++d) v[d] = (KNOCK & 4) ? u[d % 3][d % 5] : ld_c2(buf + d * 60 + L);
  __builtin_amdgcn_wave_barrier();
  at(3);
  dft12(v);
  // ---- mirror exchange: Z[k] goes to buf[k], the partner Z[720 - k] comes back
#pragma unroll
  for (int s = 0; s < 12; ++s) z14_store<DROP>(buf + L + 60 * s, v[s]);
  __builtin_amdgcn_wave_barrier();
  at(4);
  // Hermitian unpack: X_k = E_k + W^k O_k, W = exp(-2 pi i / 1440), 2 E_k = Z_k + conj Z_{720-k},
  // 2 O_k = -i (Z_k - conj Z_{720-k}), and the mirror from the same two points: X_{720-k} = conj(E_k - W^k O_k).  A lane
  // therefore takes only the FIRST six of its wavenumbers, k = L + 60 s (s < 6), with their mirrors 720 - k -- which are
  // the last six wavenumbers of lane 60 - L (of lane 0 itself, and the Nyquist wavenumber 720 for k = 0): half the mirror
  // loads, twiddle loads and twiddle products of taking all twelve.  Only k = 360 (its own mirror, lane 0's s = 6) is
  // left over: the otherwise idle lane 60 takes it in its s = 0 slot.  |X|^2 in packed fp32 (as the transform), fp64 sums.
  const bool self360 = c.lane == Z14_LANES;
  const C2 z360 = ld_c2(buf + Z14_N2 / 2);
  C2 zm[6];      // (TW_EARLY) the six mirror partners and the six W1440^k of the unpack, asked for together
  float2 wu[6];
  if constexpr (TW_EARLY) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      zm[s] = ld_c2(s == 0 ? buf + c.mir0 : buf + (Z14_N2 - 60 * s) - L);
      wu[s] = twr[s == 0 ? c.k0 : L + 60 * s];
    }
    if constexpr (TW_EARLY == 1) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    C2 zk = v[s];
    C2 zc = (KNOCK & 4) ? v[11 - s] : (TW_EARLY ? zm[s] : ld_c2(s == 0 ? buf + c.mir0 : buf + (Z14_N2 - 60 * s) - L));
    if (s == 0) {
      zk.re = self360 ? z360.re : zk.re;
      zk.im = self360 ? z360.im : zk.im;
      zc.re = self360 ? z360.re : zc.re;
      zc.im = self360 ? z360.im : zc.im;
    }
    const C2 e = {zk.re + zc.re, zk.im - zc.im};
    const C2 o = {zk.im + zc.im, zc.re - zk.re};
    const C2 wo = (KNOCK & 8) ? o : ctw(o, TW_EARLY ? wu[s] : twr[s == 0 ? c.k0 : L + 60 * s]);
    const C2 x = cadd(e, wo), xm = csub(e, wo);
    const v2 p = (KNOCK & 8) ? x.re : norm2(x);     // (row A, row B) of k
    const v2 pm = (KNOCK & 8) ? xm.re : norm2(xm);  // ... of 720 - k
    double pxd = (double)p.x, pyd = (double)p.y;
    if constexpr (WBX_SPECTRUM_DEMEAN && !(KNOCK & 8)) {
      // k = 0 (lane 0, s = 0): x.re = 2 F'_0 of the shifted rows and x.im = 0 exactly; F_0 = F'_0 + n m (2 F_0 here: E and O
      // are used without their factor 1/2), formed and squared in fp64 -- the mean is not squeezed through fp32 again
      if (s == 0 && c.lane == 0) {
        const double fa = (double)x.re.x + 2.0 * Z14_N * (double)msh.x, fb = (double)x.re.y + 2.0 * Z14_N * (double)msh.y;
        pxd = fa * fa;
        pyd = fb * fb;
      }
    }
    if constexpr (PT) {
      acc[s] = fma(pxd, sca, acc[s]);
      accm[s] = fma((double)pm.x, sca, accm[s]);
      accb[s] = fma(pyd, scb, accb[s]);
      accmb[s] = fma((double)pm.y, scb, accmb[s]);
    } else if (split) {
      acc[s] = fma(pxd, sca, acc[s]);
      accm[s] = fma((double)pm.x, sca, accm[s]);
      // (r5) `power` is row B's OWN record here (the caller opened it for group gb): one plain store per wavenumber
      if (c.lane < Z14_LANES) {
        power[L + 60 * s] = pyd * scb * ((s == 0 && L == 0) ? 1.0 : 2.0);
        power[Z14_N2 - L - 60 * s] = (double)pm.y * scb * 2.0;
      } else if (self360 && s == 0) {
        power[Z14_N2 / 2] = pyd * scb * 2.0;
      }
    } else {
      if constexpr (KNOCK & 8) {
        acc[s] += (double)(p.x + p.y);
        accm[s] += (double)(pm.x + pm.y);
      } else {
        acc[s] = fma(pxd, sca, fma(pyd, scb, acc[s]));
        accm[s] = fma((double)pm.x, sca, fma((double)pm.y, scb, accm[s]));
      }
    }
  }
  __builtin_amdgcn_wave_barrier();  // buf is overwritten by the next pair
  at(5);
}

// blockDim.x = 64 * (teams per block); dynamic LDS = Z14_TABLES * 8 + teams * Z14_BUF * 16 bytes.
// PROF (tools/spec_phase_profile.py): wave 0 of every block adds the cycles it spent in each phase of a pair to prof[0..7], in its prologue to prof[8] and in its closing flush to prof[9]
// (s_memtime stamps; the stamps after a transpose first wait for the LDS, which the production kernel does not).
// KNOCK (diagnostic, wrong results; tools/kbench_spectrum_raw.py): 1 = every pair re-reads the team's first rows (L2 hits,
// no HBM stream), 2 = the LDS stores of the three exchanges are dropped, 4 = their loads too, 8 = no unpack arithmetic.
// FETCH_EARLY (A/B): the next pair's 24 loads in front of pass 1 instead of behind its stores (0.285 vs 0.275 ms per field)
#ifndef WBX_Z14_TW_EARLY
#define WBX_Z14_TW_EARLY 0  // z14_pair<.., TW_EARLY> in the three-wave spectrum kernel (1: +45 %, 116 B of scratch; 2 = the source shape without the
                            // scheduling barriers: the same)
#endif
template <bool PROF, int KNOCK, bool ROTATE = true, bool FETCH_EARLY = false>
__global__ void __launch_bounds__(768) zspec1440_kernel(const float* __restrict__ field, int64_t row_stride, int64_t nrows,
                                                        int rows_per_team, int skew, const float2* __restrict__ tables_g,
                                                        const int32_t* __restrict__ group,
                                                        const double* __restrict__ scale, SpecRecs recs,
                                                        unsigned long long* __restrict__ prof) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const unsigned long long t_start = PROF ? __builtin_readcyclecounter() : 0ull;
  const unsigned long long w_start = PROF ? wall_clock64() : 0ull;  // 100 MHz
  float2* const tw1 = reinterpret_cast<float2*>(lds_raw);
  float2* const tw2 = tw1 + Z14_TW1;
  float2* const twr = tw2 + Z14_TW2;
  const int lane = (int)(threadIdx.x & 63);
  const int team = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int nteam = (int)(blockDim.x >> 6);
  v4* const buf = reinterpret_cast<v4*>(twr + Z14_TWR) + team * Z14_BUF;
  for (int i = threadIdx.x; i < Z14_TABLES; i += blockDim.x) tw1[i] = tables_g[i];
  __syncthreads();
  // A block's 12 * rows_per_team rows are dealt to its teams in launch order.  The instruction arbiter serves the OLDEST wave
  // of a SIMD first (teams 0-3 before 4-7 before 8-11: lifetimes 196 / 217 / 231 us with equal shares, even with the priority
  // rotation below), so with `skew` > 0 (twelve-team blocks only) the first four teams take `skew` rows more and the last four
  // `skew` rows less: the waves of a SIMD finish together and the kernel's tail shrinks.
  const int g = team >> 2;
  const int64_t lead = g == 0 ? (int64_t)team * skew : (g == 1 ? 4 * (int64_t)skew : (int64_t)(12 - team) * skew);
  const int64_t r0 = ((int64_t)blockIdx.x * nteam + team) * rows_per_team + lead;
  const int64_t mine = rows_per_team + (int64_t)skew * (1 - g);
  const int64_t r1 = r0 + mine < nrows ? r0 + mine : nrows;
  const int64_t team_id = (int64_t)blockIdx.x * nteam + team;  // the records' keys: (team, sequence number) -- fixed by the launch
  if (r0 >= r1) {  // only wave-level ordering below
    spec_rec_static(recs, (unsigned int)team_id, -1, 0ull, lane == 0);  // (the team's reserved slot stays empty)
    return;
  }
  const Z14Lane c = z14_lane(lane, buf, tw2);
  const int L = c.L;
  const double quarter_inv_nn = 0.25 / ((double)Z14_N * (double)Z14_N);  // E and O are used without their factor 1/2

  double acc[6], accm[6];  // sums of k = L + 60 s and of 720 - k (z14_send)
#pragma unroll
  for (int s = 0; s < 6; ++s) acc[s] = accm[s] = 0.0;
  int32_t cur = group[r0];
  unsigned int seq = 0;
  // a team's LAST record (for most teams the only one) goes into the slot the launch reserved for it: no counter to wait for
  auto flush = [&](int32_t next, bool last = false) {
    double* const rec = last ? spec_rec_static(recs, (unsigned int)team_id, cur, spec_key(team_id, 0xffffffu), lane == 0)
                             : spec_rec_open(recs, cur, spec_key(team_id, seq++), lane);
    z14_send<true>(rec, c, acc, accm);
#pragma unroll
    for (int s = 0; s < 6; ++s) acc[s] = accm[s] = 0.0;
    cur = next;
  };

  v2 pa[12], pb[12];  // the pair's pass-1 inputs, fetched one pair ahead
  auto fetch = [&](int64_t r) {
    const bool two = r + 1 < r1;
    const int64_t ra = (KNOCK & 1) ? r0 : r;
    const v2* rowa = reinterpret_cast<const v2*>(field + ra * row_stride) + L;
    const v2* rowb = reinterpret_cast<const v2*>(field + (two ? ra + 1 : ra) * row_stride) + L;
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      pa[a] = __builtin_nontemporal_load(rowa + 60 * a);
      pb[a] = __builtin_nontemporal_load(rowb + 60 * a);
    }
  };
  fetch(r0);
  unsigned long long stamp[8], spent[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (PROF) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    spent[8] = __builtin_readcyclecounter() - t_start;  // tables, first rows
  }
  auto mark = [&](int i, bool drain) {
    if constexpr (PROF) {
      if (drain) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp[i] = __builtin_readcyclecounter();
    }
  };
  // The instruction arbiter serves the oldest wave of a SIMD first: with equal shares of rows the three waves of a SIMD
  // finished after 175 / 226 / 275 us (configs[3]) and the SIMD ran its last 100 us under-occupied.  Rotating the user
  // priority pair by pair gives every wave the same share of the issue slots.
  int turn = team >> 2;  // waves t, t + 4, t + 8 of a block share a SIMD
  for (int64_t r = r0; r < r1;) {
    if constexpr (ROTATE) {
      turn = turn == 2 ? 0 : turn + 1;
      if (turn == 0) __builtin_amdgcn_s_setprio(0);
      else if (turn == 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(2);
    }
    mark(0, false);
    // a missing second row is a row of zeros with scale 0.  (r5) So is the second row of a pair that would straddle two
    // groups: row r goes alone and row r + 1 opens the next pair (its loads are issued again; ~0.6 % of configs[3]'s pairs)
    // -- the sums in flight always belong to ONE group and a record is only ever written from the accumulators.
    const int32_t ga = group[r];
    const bool two = r + 1 < r1 && group[r + 1] == ga;
    const int64_t rn = r + (two ? 2 : 1);
    const double sca = scale[r] * quarter_inv_nn, scb = two ? scale[r + 1] * quarter_inv_nn : 0.0;
    C2 v[12];
#pragma unroll
    for (int a = 0; a < 12; ++a) v[a] = {{pa[a].x, two ? pb[a].x : 0.f}, {pa[a].y, two ? pb[a].y : 0.f}};
    const v2 msh = z14_demean(v);
    if constexpr (PROF) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mark(1, false);  // 0 -> 1: scalar bookkeeping + the wait for the prefetched rows
    if constexpr (FETCH_EARLY) {
      if (rn < r1) fetch(rn);
    }

    if (ga != cur) flush(ga);  // wave-uniform
    // stamps (PROF): 1 -> 2 pass 1 | 2 -> 3 transpose 1 round trip (stores drain, 15 loads return) | 3 -> 4 pass 2 |
    // 4 -> 5 transpose 2 round trip | 5 -> 6 pass 3 | 6 -> 7 mirror exchange + unpack + fp64 sums
    z14_pair<(KNOCK & 14), false, WBX_Z14_TW_EARLY>(v, buf, c, tw1, twr, sca, scb, false, ga, acc, accm, nullptr,
                           [&](int i) {
                             mark(i + 2, i == 1 || i == 3 || i == 5);
                             if constexpr (!FETCH_EARLY) {  // the registers of pass 1's inputs are free: the next pair's loads
                               if (i == 0 && rn < r1) fetch(rn);
                             }
                           }, acc, accm, msh);
    r = rn;
    if constexpr (PROF) {
#pragma unroll
      for (int i = 1; i < 8; ++i) spent[i] += stamp[i] - stamp[i - 1];
      spent[0] += 1;
    }
  }
  const unsigned long long t_flush = PROF ? __builtin_readcyclecounter() : 0ull;
  flush(cur, true);
  if constexpr (PROF) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    spent[9] = __builtin_readcyclecounter() - t_flush;  // the closing atomics
    if (team == 0 && lane == 0) {
#pragma unroll
      for (int i = 0; i < 10; ++i) atomicAdd(prof + i, spent[i]);
      const unsigned long long w_end = wall_clock64();
      atomicAdd(prof + 10, w_end - w_start);  // lifetime of the wave
      atomicMin(prof + 11, w_start);
      atomicMax(prof + 12, w_end);
      atomicMax(prof + 13, w_start);  // the last block to start
    }
    if (lane == 0) atomicAdd(prof + 14 + team, wall_clock64() - w_start);  // lifetime by team (= launch order on its SIMD)
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Latitude-fastest fields (the public ERA5 / WeatherBench layout [.., longitude, latitude]): element (lon j, row r) of a
// slab sits at slab + j * lon_stride + r, so the two rows of a pair are ADJACENT floats and one 8-byte load delivers
// (A[j], B[j]) -- exactly one half (re or im, rows A and B) of the packed point m = j / 2.  A block takes a run of <= 24
// adjacent rows of one slab = one row pair per team: its 768 threads read the 1440 x (<= 96 byte) segments with
// consecutive threads on consecutive pairs of a longitude, one step ahead into registers (a quarter of the loads behind
// each pass's stores and behind the unpack: all twelve waves leave a barrier together, and one burst of 23 loads per
// thread is 9 K cycles of texture-addresser time in front of the passes), and store them into the teams' LDS buffers at
// the start of the next step (element j of team t's buffer = the 8 bytes of longitude j: already the pass-1 layout);
// the teams then run the same three passes with pass 1 reading the LDS instead of global memory.  One pass over the
// field, no transposed scratch (the generic route: transpose_rows_kernel + fused kernel = three passes).
// The 96-byte segments do not line up with the 128-byte lines, so a line is shared with the neighbouring runs of rows:
// the runs of a slab go to the blocks of ONE XCD (blockIdx & 7; an XCD owns a contiguous eighth of the slabs) at about
// the same time, so the line's other users find it in that XCD's L2 (FETCH_SIZE x 2 = 1.06-1.10 x the field).
constexpr int Z14_TEAMS = 12;                 // teams (row pairs) of a latitude-fastest block
constexpr int Z14_RUN = 2 * Z14_TEAMS;        // rows of a run
constexpr int Z14_BUFL = 733;                 // v4 elements between the team buffers (odd: the staging stores of one
                                              // longitude go to 12 buffers at once)
constexpr int Z14_STAGE = (Z14_N * Z14_TEAMS + 64 * Z14_TEAMS - 1) / (64 * Z14_TEAMS);  // 8-byte items per thread: 23

typedef float v2u __attribute__((ext_vector_type(2), aligned(4)));

// grid = 8 * (blocks per XCD); runs_per_slab >= ceil(rps / 24) and rps = run_base * runs_per_slab + run_rem; row i of slab o
// is row o * rps + i of group / scale
// PROF: wave 0 of every block adds its cycles per step to prof[1..6] (wait at the first barrier | staging stores + second
// barrier | issue of the next run's loads | pass-1 loads from the LDS | the three passes and the unpack), steps to prof[0].
// KNOCK (diagnostic, wrong results): 1 = no global loads, 2 = no passes (the staged data is only summed)
template <bool PROF, int KNOCK = 0, int SPREAD = 2>
__global__ void __launch_bounds__(64 * Z14_TEAMS) zspec1440_latfast_kernel(
    const float* __restrict__ field, int64_t lon_stride, const int64_t* __restrict__ slab_off, int64_t rps, int64_t nslab,
    int64_t slabs_per_xcd, int runs_per_slab, int run_base, int run_rem, int prio, const float2* __restrict__ tables_g,
    const int32_t* __restrict__ group, const double* __restrict__ scale, SpecRecs recs, unsigned long long* __restrict__ prof) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int team_in_table[Z14_TEAMS];  // the team's sums of the last step sit in its buffer, for the block's table
  float2* const tw1 = reinterpret_cast<float2*>(lds_raw);
  float2* const tw2 = tw1 + Z14_TW1;
  float2* const twr = tw2 + Z14_TW2;
  v4* const bufs = reinterpret_cast<v4*>(twr + Z14_TWR);
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int team = __builtin_amdgcn_readfirstlane(tid >> 6);
  // every step ends in a block barrier, and the arbiter serves the oldest wave of a SIMD first: with `prio` = 1 the younger
  // waves of a SIMD (teams 4-7, 8-11) get the higher user priority, so that the three reach the barrier together (2: the reverse)
  if (prio != 0) {
    const int rank = prio == 1 ? team >> 2 : 2 - (team >> 2);
    if (rank == 0) __builtin_amdgcn_s_setprio(0);
    else if (rank == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(2);
  }
  v4* const buf = bufs + team * Z14_BUFL;
  for (int i = tid; i < Z14_TABLES; i += 64 * Z14_TEAMS) tw1[i] = tables_g[i];
  constexpr int nk = Z14_N2 + 1;
  const Z14Lane c = z14_lane(lane, buf, tw2);
  const int L = c.L;
  const double quarter_inv_nn = 0.25 / ((double)Z14_N * (double)Z14_N);

  // the block's steps: XCD x owns a contiguous eighth of the slabs (callers list the slabs of one group next to each other:
  // a block then changes group -- and empties its table of sums -- only every few steps); its (slab, run) pairs, slab-major,
  // are dealt out to its blocks round-robin, so the blocks of an XCD work on one or two neighbouring slabs at any time
  const int xcd = (int)(blockIdx.x & 7u), local = (int)(blockIdx.x >> 3), nlocal = (int)(gridDim.x >> 3);
  // (no integer division in here: a 64-bit divide is a ~1000-cycle routine on this ISA; the host sends rows = base * runs + rem)
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
  };
  // loader role: thread -> (team t_ld, longitudes j0 + 64 n); its 8 bytes of longitude j belong at element j of that team's buffer
  const int t_ld = tid % Z14_TEAMS, j0 = tid / Z14_TEAMS;
  v2* const stage_dst = reinterpret_cast<v2*>(bufs + t_ld * Z14_BUFL) + j0;
  v2 held[Z14_STAGE];  // rows a short last run does not have stay at their previous (first: zero) value; nobody reads them
#pragma unroll
  for (int n = 0; n < Z14_STAGE; ++n) held[n] = (KNOCK & 1) ? (v2){1.f, 2.f} : (v2){0.f, 0.f};
  // part p of the run's loads: items [6 p, 6 p + 6) (p = 3: the rest).  All twelve waves of the block leave the barrier
  // together; 23 loads each at once is 7 us of texture-addresser time in front of the passes, spread over the passes they
  // run under the other waves' arithmetic.
  auto load_part = [&](int64_t oo, int rr, int part) {
    if constexpr (KNOCK & 1) return;
    int64_t rbeg, rend;
    run_rows(rr, rbeg, rend);
    const int64_t ra = rbeg + 2 * t_ld;
    const float* src = field + (slab_off ? slab_off[oo] : 0) + ra + (int64_t)j0 * lon_stride;
    if (ra + 1 < rend) {
#pragma unroll
      for (int n = 0; n < Z14_STAGE; ++n)
        if ((n / 6 == part || (part == 3 && n >= 18)) && (n < Z14_STAGE - 1 || j0 + 64 * n < Z14_N))
          held[n] = *reinterpret_cast<const v2u*>(src + (int64_t)(64 * n) * lon_stride);
    } else if (ra < rend) {  // a lone last row: nothing may be read behind it
#pragma unroll
      for (int n = 0; n < Z14_STAGE; ++n)
        if ((n / 6 == part || (part == 3 && n >= 18)) && (n < Z14_STAGE - 1 || j0 + 64 * n < Z14_N))
          held[n] = (v2){src[(int64_t)(64 * n) * lon_stride], 0.f};
    }
  };
  auto load_run = [&](int64_t oo, int rr) {
#pragma unroll
    for (int part = 0; part < 4; ++part) load_part(oo, rr, part);
  };
  // Sums (r5: no atomics, in a fixed order).  A team's sums of ONE step (fp64 registers, k = L + 60 s and the mirrors) go into
  // its own staging buffer when the step's passes are done -- the buffer is idle then -- and behind the next barrier every
  // thread adds the twelve buffers, team 0 first, into the BLOCK's [721] table in the LDS.  The table goes out as ONE RECORD
  // (plain stores; spec_close_kernel adds the records of a group in key order) when the block's group changes: consecutive
  // steps of a block are different slabs, i.e. normally different groups, so a record holds the 24 rows of a step -- 5.8 KB
  // written per 138 KB read.  A team whose rows are not of the step's group (a run that crosses a group boundary) writes
  // records of its own.
  double* const blk = reinterpret_cast<double*>(bufs + Z14_TEAMS * Z14_BUFL);
  for (int k = tid; k < nk; k += 64 * Z14_TEAMS) blk[k] = 0.0;
  if (tid < Z14_TEAMS) team_in_table[tid] = 0;
  int32_t blk_group = -1;  // block-uniform
  double acc[6], accm[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) acc[s] = accm[s] = 0.0;
  const int64_t block_id = (int64_t)blockIdx.x;
  unsigned int seq_team = 0;  // keys: a step's slot number for the table; (1 << 30) + (block, team, n) for a team's own records
  auto team_key = [&]() { return (1ull << 62) | spec_key(block_id * Z14_TEAMS + team, seq_team++); };
  double* const own = reinterpret_cast<double*>(buf);  // the team's buffer as 721 doubles (5.8 of its 11.7 KB)
  // the teams' sums of the step that just ended -> the block's table, in team order (every thread; between two block barriers)
  auto gather_teams = [&]() {
    for (int k = tid; k < nk; k += 64 * Z14_TEAMS) {
      double sum = blk[k];
#pragma unroll
      for (int t = 0; t < Z14_TEAMS; ++t)
        if (team_in_table[t]) sum += reinterpret_cast<const double*>(bufs + t * Z14_BUFL)[k];
      blk[k] = sum;
    }
  };
  // Every step (slab o, run) of the launch has a record slot of its own (slot = o * runs_per_slab + run: fixed by the geometry,
  // no counter, no barrier).  The table is written into the slot of the LAST step that added to it; the slots of the steps in
  // between stay empty (header group -1).
  int64_t table_slot = -1;  // the slot of the step whose sums are in the table (block-uniform)
  auto flush_block = [&](int32_t next) {  // every thread of the block, block-uniformly
    if (blk_group >= 0) {
      double* const rec = spec_rec_static(recs, (unsigned int)table_slot, blk_group, (unsigned long long)table_slot, tid == 0);
      for (int k = tid; k < nk; k += 64 * Z14_TEAMS) {
        const double sum = blk[k];
        rec[k] = k == 0 ? sum : 2.0 * sum;  // S_k, include/wbx.h
        blk[k] = 0.0;
      }
    }
    blk_group = next;
  };
  if (o < o_end) load_run(o, run);
  unsigned long long stamp[6], spent[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto mark = [&](int i) {
    if constexpr (PROF) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp[i] = __builtin_readcyclecounter();
    }
  };
  while (o < o_end) {  // block-uniform
    mark(0);
    __syncthreads();  // every team is done with its buffer (and, the first time, the tables are in place)
    mark(1);
    gather_teams();   // the sums of the step that just ended (they sit in the teams' buffers) -> the block's table
    __syncthreads();  // ... before the buffers are filled again
    if (tid < Z14_TEAMS) team_in_table[tid] = 0;
    int64_t rbeg, rend;
    run_rows(run, rbeg, rend);
    const int64_t row0 = o * rps;
    const int64_t ra = rbeg + 2 * team;
    const bool active = ra < rend, two = ra + 1 < rend;  // team-uniform
    const int32_t g0 = group[row0 + rbeg];               // the step's group (block-uniform)
    int32_t ga = g0, gb = g0;
    double sca = 0.0, scb = 0.0;
    if (active) {
      ga = group[row0 + ra];
      gb = two ? group[row0 + ra + 1] : ga;
      sca = scale[row0 + ra] * quarter_inv_nn;
      scb = two ? scale[row0 + ra + 1] * quarter_inv_nn : 0.0;
    }
#pragma unroll
    for (int n = 0; n < Z14_STAGE; ++n)
      if (n < Z14_STAGE - 1 || j0 + 64 * n < Z14_N) stage_dst[64 * n] = held[n];
    __syncthreads();
    mark(2);
    // block-uniform: the table holds the sums of the earlier steps -- of another group: out they go (into the last of those steps'
    // slot); of this step's group: they stay, and that step's slot is marked empty
    if (g0 != blk_group) flush_block(g0);
    else if (table_slot >= 0) spec_rec_static(recs, (unsigned int)table_slot, -1, 0ull, tid == 0);
    table_slot = o * runs_per_slab + run;
    int64_t on = o;
    int rn = run + nlocal;
    normalise(on, rn);
    const bool more = on < o_end;
    mark(3);
    if (active) {
      C2 v[12];
#pragma unroll
      for (int a = 0; a < 12; ++a) v[(6 * a) % 12 + a / 2] = ld_c2(buf + 60 * ((6 * a) % 12 + a / 2) + L);  // 0, 6, 1, 7, ..: z14_demean
      __builtin_amdgcn_wave_barrier();
      mark(4);
      const v2 msh = (KNOCK & 2) ? (v2){0.f, 0.f} : z14_demean<true>(v);
      if constexpr (KNOCK & 2) {
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[a] += (double)(v[a].re.x + v[a + 6].im.y) * sca;
        if (more) load_run(on, rn);
      } else {
        // SPREAD: where the next run's loads are issued -- 0 one burst in front of the passes, 1 a quarter in front and one
        // behind the stores of each pass, 2 a quarter behind each pass's stores and the last behind the unpack (production);
        // configs[3]: 0.498 / 0.390 / 0.382 ms per field
        if constexpr (SPREAD == 0) {
          if (more) load_run(on, rn);
        } else if constexpr (SPREAD == 1) {
          if (more) load_part(on, rn, 0);
        }
        // (a pair that straddles two groups: row B's values are a record of their own)
        double* const rec_b = gb != ga ? spec_rec_open(recs, gb, team_key(), lane) : nullptr;
        z14_pair<0>(v, buf, c, tw1, twr, sca, scb, gb != ga, gb, acc, accm, rec_b, [&](int i) {
          if constexpr (SPREAD == 1) {
            if (more && !(i & 1) && i < 6) load_part(on, rn, i / 2 + 1);
          } else if constexpr (SPREAD == 2) {
            if (more && (!(i & 1) || i == 5)) load_part(on, rn, i == 5 ? 3 : i / 2);
          }
        }, acc, accm, msh);
      }
      // this step's sums leave the registers: into the team's own buffer for the block's table (the buffer is idle until the
      // next step's staging stores, two barriers away), or -- rows of another group than the step's -- into a record
      __builtin_amdgcn_wave_barrier();
      if (ga == g0) {
        z14_send<false>(own, c, acc, accm);
        if (lane == 0) team_in_table[team] = 1;
      } else {
        z14_send<true>(spec_rec_open(recs, ga, team_key(), lane), c, acc, accm);
      }
#pragma unroll
      for (int s = 0; s < 6; ++s) acc[s] = accm[s] = 0.0;
      mark(5);
      if constexpr (PROF) {
#pragma unroll
        for (int i = 1; i < 6; ++i) spent[i] += stamp[i] - stamp[i - 1];
        spent[0] += 1;
      }
    } else if (more) {
      load_run(on, rn);  // (a team without rows in this run still loads its share of the next one)
    }
    o = on;
    run = rn;
  }
  if constexpr (PROF) {
    if (team == 0 && lane == 0) {
#pragma unroll
      for (int i = 0; i < 6; ++i) atomicAdd(prof + i, spent[i]);
    }
  }
  __syncthreads();
  gather_teams();  // the last step's sums
  __syncthreads();
  flush_block(-1);
}

// host side: W720^(b k1) at (k1 - 1) * 60 + b | W60^(d q) at (q - 1) * 12 + d | W1440^k, k < 720
static void zspec1440_tables(std::vector<float2>& host) {
  host.resize(Z14_TABLES);
  auto unit = [](double num, double den) {
    const double a = -2.0 * M_PI * num / den;
    return make_float2((float)cos(a), (float)sin(a));
  };
  for (int k1 = 1; k1 < 12; ++k1)
    for (int b = 0; b < 60; ++b) host[(k1 - 1) * 60 + b] = unit((double)((b * k1) % 720), 720.0);
  for (int q = 1; q < 5; ++q)
    for (int d = 0; d < 12; ++d) host[Z14_TW1 + (q - 1) * 12 + d] = unit((double)((d * q) % 60), 60.0);
  for (int k = 0; k < Z14_TWR; ++k) host[Z14_TW1 + Z14_TW2 + k] = unit((double)k, 1440.0);
}
