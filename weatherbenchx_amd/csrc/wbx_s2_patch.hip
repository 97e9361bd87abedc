// Stage 2 over FULL-MAP partials with boolean bins: the patch / slot reduction of wbx_det_binned.hip applied to
// partial[A][Bk][Br][chunk][lane][j] (fp64 lane sums from any stage-1 kernel -- ensemble CRPS lanes, PASS1 statistics,
// deterministic lanes) instead of p, t, c.  It replaces s2_bits_kernel when x is W-dependent and summed here and the
// partial is large: that kernel re-reads the partial once per lane with 256-thread blocks per (A, Bk, lane) and reaches
// ~0.2 TB/s on the public benchmark's 1-init chunks (7.5 GB of partials -> 33-46 ms).
//
// One wave = (cell (A, Bk), 64 consecutive j, a range of Br rows); per row it sums the chunks, multiplies by wt once and
// feeds only the bin slots some point of the tile is a member of.  Same NaN * 0 = NaN semantics (poison lane).
#include <cstdlib>

#include "wbx_patch.hpp"

namespace wbx {

template <int NL, int K>
__global__ void __launch_bounds__(64) s2_patch_kernel(const double* __restrict__ partial, int64_t nchunk, BinnedArgs g) {
  const int lane = threadIdx.x;
  int64_t cell;
  int xt, rs;
  if (!patch_decode<1>(g, cell, xt, rs)) return;
  const int64_t bk = cell % g.nBk;
  const int64_t rbeg = (int64_t)rs * g.rows_per_split;
  const int64_t rend = rbeg + g.rows_per_split < g.nBr ? rbeg + g.rows_per_split : g.nBr;
  const bool live = (int64_t)xt * 64 + lane < g.nj;
  const int64_t x = live ? (int64_t)xt * 64 + lane : g.nj - 1;
  const int64_t patch = (int64_t)rs * g.nxt + xt;
  unsigned long long todo = g.uni[bk * ((int64_t)g.nrs * g.nxt) + patch];
  todo = (unsigned long long)readlane64((int64_t)todo, 0);
  double* const out = g.tmp + (cell * ((int64_t)g.nrs * g.nxt) + patch) * (NL * (int64_t)g.nbin);
  const int64_t row_stride = nchunk * NL * g.nj;  // elements between consecutive Br rows of one cell
  const double* const base = partial + cell * g.nBr * row_stride + x;
  const double* const wbase = g.wt + bk * g.nBr * g.nj + x;
  const unsigned long long* const bbase = g.bits + bk * g.nBr * g.nj + x;
  bool first = true;
  do {
    unsigned long long smask[K];
    int sbin[K];
#pragma unroll
    for (int q = 0; q < K; ++q) {
      sbin[q] = todo ? __builtin_ctzll(todo) : 0;
      smask[q] = todo & (~todo + 1ull);
      todo &= todo - 1ull;
    }
    unsigned long long sweep = 0ull;  // the bins of this sweep
#pragma unroll
    for (int q = 0; q < K; ++q) sweep |= smask[q];
    double acc[NL][K];
    double poison[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      poison[l] = 0.0;
#pragma unroll
      for (int q = 0; q < K; ++q) acc[l][q] = 0.0;
    }
    auto accumulate = [&](const double (&v)[NL], double w, unsigned long long bw) {
      const unsigned long long tile = wave_or64_of(live ? bw : 0ull, sweep);
      if (live) {
        double m[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          m[l] = v[l] * w;
          poison[l] = fma(m[l], 0.0, poison[l]);
        }
#pragma unroll
        for (int q = 0; q < K; ++q) {
          if (tile & smask[q]) {  // wave-uniform
            const double f = __hiloint2double((bw & smask[q]) ? 0x3FF00000 : 0, 0);
#pragma unroll
            for (int l = 0; l < NL; ++l) acc[l][q] = fma(m[l], f, acc[l][q]);
          }
        }
      }
    };
    if (nchunk == 1) {
      // one row ahead, every load unconditional (see wbx_det_binned.hip)
      double v0[NL], v1[NL], w0, w1;
      unsigned long long b0, b1;
      auto fetch = [&](int64_t br, double (&v)[NL], double& w, unsigned long long& bw) {
        const double* p = base + br * row_stride;
#pragma unroll
        for (int l = 0; l < NL; ++l) v[l] = p[(int64_t)l * g.nj];
        w = wbase[br * g.nj];
        bw = bbase[br * g.nj];
      };
      const int64_t last = rend - 1;
      fetch(rbeg, v0, w0, b0);
      for (int64_t br = rbeg; br < rend; br += 2) {
        fetch(br + 1 < last ? br + 1 : last, v1, w1, b1);
        accumulate(v0, w0, b0);
        fetch(br + 2 < last ? br + 2 : last, v0, w0, b0);
        if (br + 1 < rend) accumulate(v1, w1, b1);
      }
    } else {
      for (int64_t br = rbeg; br < rend; ++br) {
        double v[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) v[l] = 0.0;
        const double* p = base + br * row_stride;
        for (int64_t c = 0; c < nchunk; ++c)
#pragma unroll
          for (int l = 0; l < NL; ++l) v[l] += p[(c * NL + l) * g.nj];
        accumulate(v, wbase[br * g.nj], bbase[br * g.nj]);
      }
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
      if (smask[q]) {
#pragma unroll
        for (int l = 0; l < NL; ++l) {
          const double sum = wave_sum(acc[l][q]);
          if (lane == 0) out[(int64_t)l * g.nbin + sbin[q]] = sum;
        }
      }
    }
    if (first) {
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const double sum = wave_sum(poison[l]);
        if (lane == 0) g.tmp_poison[(cell * ((int64_t)g.nrs * g.nxt) + patch) * NL + l] = sum;
      }
      first = false;
    }
  } while (todo);
}

template <int NL>
static int launch_s2_patch(wbx_ctx* ctx, const wbx_s2_plan& p, const double* partial, const double* wt,
                           const uint64_t* bits, double* out) {
  // 2 * NL * K accumulator VGPRs + 6 * NL (two rows in flight + products) + ~40 <= 168: 3 waves / SIMD
  constexpr int K = NL <= 1 ? 32 : (NL <= 2 ? 24 : (NL <= 3 ? 16 : (NL <= 4 ? 12 : (NL <= 5 ? 8 : (NL <= 6 ? 7 : (NL <= 7 ? 5 : 2))))));
  BinnedArgs g;
  if (int rc = patch_setup(ctx, g, wt, bits, p.nA * p.nBk, p.nBk, p.nBr, p.nj, 1, p.nj, NL, (int)p.nbin)) return rc;
  const int64_t grid = (g.nblocks + 7) / 8 * 8;
  hipLaunchKernelGGL((s2_patch_kernel<NL, K>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, partial, (int64_t)p.nchunk, g);
  WBX_HIP(hipGetLastError());
  return patch_finish(ctx, g, NL, out);
}

bool s2_patch_eligible(const wbx_s2_plan& p) {
  if (!p.sum_j || p.nj < 64) return false;
  switch (p.nlane) {
    case 1: case 2: case 3: case 4: case 5: case 6: case 7: case 10: case 12:
      break;
    default:
      return false;
  }
  // worth the three launches only when the partial is big (>= 8 MB); WBX_S2_PATCH_MIN (elements) lets the tests drive
  // small cases through this kernel
  const char* e = getenv("WBX_S2_PATCH_MIN");
  const int64_t min_elems = e ? atoll(e) : (int64_t)1 << 20;
  return p.nA * p.nBk * p.nBr * p.nchunk * p.nlane * p.nj >= min_elems;
}

int s2_patch(wbx_ctx* ctx, const wbx_s2_plan& p, const double* partial, const double* wt, const uint64_t* bits,
             double* out) {
  switch (p.nlane) {
#define WBX_CASE(N) \
    case N:         \
      return launch_s2_patch<N>(ctx, p, partial, wt, bits, out);
    WBX_CASE(1) WBX_CASE(2) WBX_CASE(3) WBX_CASE(4) WBX_CASE(5) WBX_CASE(6) WBX_CASE(7) WBX_CASE(10) WBX_CASE(12)
#undef WBX_CASE
  }
  return fail(WBX_ERR_INVALID, "s2_patch: unsupported lane count %lld", (long long)p.nlane);
}

}  // namespace wbx
