// Shared pieces of the "patch" reductions (wbx_det_binned.hip: fused statistics + bins; wbx_s2_patch.hip: stage 2 over
// full-map partials): per-wave patches of the (row, x) plane, the union of a patch's bins, slot dealing, the final sum
// over patches.  See the header comment of wbx_det_binned.hip for the design.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

#include "wbx_common.hpp"

namespace wbx {

constexpr int WBX_SPLIT_MAX = 255;

struct BinnedArgs {
  const double* wt;                  // [nBk][nBr][nj]
  const unsigned long long* bits;    // [nBk][nBr][nj]
  int64_t nBk, nBr, nj;              // nj = nx if W depends on x, else 1
  int32_t nbin, nxt, nrs;            // x tiles, row splits: npatch = nrs * nxt
  int64_t rows_per_split;            // rows = nBr * D
  int64_t ncell, nblocks;            // nA * nBk, ncell * npatch
  double* tmp;                       // [cell][patch][NA][nbin], zeroed
  double* tmp_poison;                // [cell][patch][NA]
  unsigned long long* uni;           // [nBk][patch]: union of the patch's membership words
  // "atoms" of a patch = its distinct membership words (regions are boxes: a patch of 64 x ~150 rows sees a handful)
  uint8_t* aid;                      // [nBk][nBr][nj]: index of the point's word in its patch's list
  const uint8_t* aidm;               // the same with masked-out points set to 255 (det_atoms_kernel<.., MERGED>), or NULL
  unsigned long long* words;         // [nBk][patch][ATOM_MAX]
  int32_t* nwords;                   // [nBk][patch]: entries in the list, -1 = more than ATOM_MAX (slot kernel takes it)
  int32_t atoms;                     // 0: every patch goes to the slot kernel (A/B timing: WBX_BINNED_ATOMS=0)
  int32_t order;                     // block order: 0 cell fastest, 1 x tile fastest (see patch_decode)
  // Tapered row splits (patch_taper; the ensemble atom kernel): split rs = Br rows [split_br[rs], split_br[rs + 1]) instead of
  // rs * rows_per_split ..: an XCD walks a contiguous eighth of the splits in order (patch_decode), and the last splits of each
  // eighth are half and quarter size, so that its 384 wave slots run dry within a short patch instead of a long one
  int32_t taper;
  int32_t* split_tab;                // the same table in device memory, next to the atom tables (written by binned_atoms_kernel):
  int32_t split_br[WBX_SPLIT_MAX + 1];  // the main kernel reads it with scalar loads
};

constexpr int ATOM_MAX = 32;

// Rows [rbeg, rend) of row split rs (rows = Br rows x D depth rows).
__device__ __forceinline__ void patch_rows(const BinnedArgs& g, int rs, int64_t D, int64_t& rbeg, int64_t& rend) {
  if (g.taper) {
    rbeg = (int64_t)g.split_br[rs] * D;
    rend = (int64_t)g.split_br[rs + 1] * D;
  } else {
    const int64_t R = g.nBr * D;
    rbeg = (int64_t)rs * g.rows_per_split;
    rend = rbeg + g.rows_per_split < R ? rbeg + g.rows_per_split : R;
  }
}

// OR of a 32-bit value over the 64 lanes (all lanes must be active); the result is wave-uniform (SGPR).
__device__ __forceinline__ uint32_t wave_or32(uint32_t v) {
  int x = (int)v;
  x |= __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
  x |= __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
  x |= __builtin_amdgcn_update_dpp(0, x, 0x124, 0xf, 0xf, true);  // row_ror:4
  x |= __builtin_amdgcn_update_dpp(0, x, 0x128, 0xf, 0xf, true);  // row_ror:8  -> every lane holds its row's OR
  x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);  // row_bcast:15 into rows 1, 3
  x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);  // row_bcast:31 into rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}

__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
  return ((unsigned long long)wave_or32((uint32_t)(v >> 32)) << 32) | wave_or32((uint32_t)v);
}

// wave_or64(v) & want, reducing only the 32-bit halves `want` (wave-uniform) has bits in: the slots of one sweep
// usually sit in one half of the membership word, which halves the DPP work per tile.
__device__ __forceinline__ unsigned long long wave_or64_of(unsigned long long v, unsigned long long want) {
  unsigned long long r = 0ull;
  if ((uint32_t)want) r = wave_or32((uint32_t)v);
  if ((uint32_t)(want >> 32)) r |= (unsigned long long)wave_or32((uint32_t)(v >> 32)) << 32;
  return r;
}

// uni[bk][patch] |= OR of bits over the patch's (rows, 64 x); the 4 waves of a block interleave over the rows
static __global__ void __launch_bounds__(256) binned_union_kernel(BinnedArgs g, int64_t D, int64_t nx) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t b = blockIdx.x;
  const int xt = (int)(b % g.nxt);
  b /= g.nxt;
  const int rs = (int)(b % g.nrs);
  const int64_t bk = b / g.nrs;
  const int64_t R = g.nBr * D;
  const int64_t rbeg = (int64_t)rs * g.rows_per_split;
  const int64_t rend = rbeg + g.rows_per_split < R ? rbeg + g.rows_per_split : R;
  const bool live = (int64_t)xt * 64 + lane < nx;
  const int64_t xw = g.nj > 1 ? (live ? (int64_t)xt * 64 + lane : nx - 1) : 0;
  unsigned long long mine = 0ull;
  for (int64_t br = rbeg / D + wave; br <= (rend - 1) / D; br += 4) mine |= g.bits[(bk * g.nBr + br) * g.nj + xw];
  const unsigned long long all = wave_or64(live ? mine : 0ull);
  if (lane == 0 && all) atomicOr(&g.uni[bk * ((int64_t)g.nrs * g.nxt) + (int64_t)rs * g.nxt + xt], all);
}

// One wave per (bk, patch): the list of the patch's distinct membership words ("atoms"), every point's index in it, and
// the union of the words.  Row splits are aligned to Br (patch_setup), so a (bk, br, x) point belongs to one patch.
// Rows are fetched RB at a time (the loop is latency bound: one wave, dependent list updates); a lane whose word equals
// the one it saw in the previous row keeps that index, so the list is searched only where region / land-sea borders
// are crossed.
static __global__ void __launch_bounds__(64) binned_atoms_kernel(BinnedArgs g, int64_t D, int64_t nx) {
  constexpr int RB = 16;
  __shared__ unsigned long long list[ATOM_MAX];
  const int lane = threadIdx.x;
  int64_t b = blockIdx.x;
  const int xt = (int)(b % g.nxt);
  b /= g.nxt;
  const int rs = (int)(b % g.nrs);
  const int64_t bk = b / g.nrs;
  int64_t rbeg, rend;
  patch_rows(g, rs, D, rbeg, rend);
  const int64_t br0 = rbeg / D, br1 = (rend - 1) / D;  // inclusive
  const bool live = (int64_t)xt * 64 + lane < nx;
  const int64_t xw = g.nj > 1 ? (live ? (int64_t)xt * 64 + lane : nx - 1) : 0;
  const bool writer = live && (g.nj > 1 || lane == 0);  // W independent of x: one byte per row, written once
  int count = 0;
  bool overflow = false;
  unsigned long long prev = 0ull;
  int prev_idx = -1;
  for (int64_t brb = br0; brb <= br1; brb += RB) {
    unsigned long long w[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int64_t br = brb + u <= br1 ? brb + u : br1;
      w[u] = g.bits[(bk * g.nBr + br) * g.nj + xw];
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      if (brb + u > br1) break;  // wave-uniform
      const unsigned long long me = w[u];
      int idx = (prev_idx >= 0 && me == prev) ? prev_idx : -1;
      bool pending = live && idx < 0;
      if (__builtin_amdgcn_ballot_w64(pending)) {
        for (int k = 0; k < count; ++k)
          if (pending && list[k] == me) {
            idx = k;
            pending = false;
          }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(pending);
        while (todo) {
          const int leader = __builtin_ctzll(todo);
          const unsigned long long wl = (unsigned long long)readlane64((int64_t)me, leader);
          int slot = count;
          if (count < ATOM_MAX) {
            if (lane == 0) list[count] = wl;
            ++count;
          } else {
            overflow = true;
            slot = 0;
          }
          if (pending && me == wl) {
            idx = slot;
            pending = false;
          }
          todo = __builtin_amdgcn_ballot_w64(pending);
        }
        __syncthreads();  // list[] was written by lane 0 and is read by every lane (one wave per block: free)
      }
      prev = me;
      prev_idx = idx;
      if (writer) g.aid[(bk * g.nBr + brb + u) * g.nj + xw] = (uint8_t)idx;
    }
  }
  const int64_t pidx = bk * ((int64_t)g.nrs * g.nxt) + (int64_t)rs * g.nxt + xt;
  if (g.taper && blockIdx.x == 0)
    for (int i = lane; i <= g.nrs; i += 64) g.split_tab[i] = g.split_br[i];
  __syncthreads();
  unsigned long long all = 0ull;
  for (int k = 0; k < count; ++k) all |= list[k];
  if (overflow) {  // the list is incomplete: take the union the slow way
    unsigned long long mine = 0ull;
    for (int64_t br = br0; br <= br1; ++br) mine |= g.bits[(bk * g.nBr + br) * g.nj + xw];
    all = wave_or64(live ? mine : 0ull);
  }
  if (lane < ATOM_MAX) g.words[pidx * ATOM_MAX + lane] = lane < count ? list[lane] : 0ull;
  if (lane == 0) {
    g.uni[pidx] = all;
    g.nwords[pidx] = (overflow || !g.atoms) ? -1 : count;
  }
}

// out[cell][lane][bin] = sum over patches of tmp[cell][patch][lane][bin] + poison[cell][patch][lane].
// One block per (cell, lane): thread (pg, bin) sums every (256 / nbin)-th patch, LDS folds the pg.
// accumulate: out += the sums (WBX_BINNED_ACCUMULATE: `out` is a chunk loop's accumulator)
static __global__ void __launch_bounds__(256) det_binned_finish(int64_t npatch, int nacc, int nbin,
                                                         const double* __restrict__ tmp,
                                                         const double* __restrict__ poison, double* __restrict__ out,
                                                         int accumulate) {
  __shared__ double red[256];
  const int64_t cell = blockIdx.x / nacc;
  const int l = (int)(blockIdx.x % nacc);
  const int ng = 256 / nbin;
  const int bin = threadIdx.x % nbin, pg = threadIdx.x / nbin;
  double s = 0.0;
  if (pg < ng)
    for (int64_t k = pg; k < npatch; k += ng)
      s += tmp[((cell * npatch + k) * nacc + l) * nbin + bin] + poison[(cell * npatch + k) * nacc + l];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < nbin) {
    for (int q = 1; q < ng; ++q) s += red[q * nbin + threadIdx.x];
    double* const dst = out + (cell * nacc + l) * nbin + threadIdx.x;
    *dst = accumulate ? *dst + s : s;
  }
}


// Patch geometry: rows = nBr * D reduced rows of nx points per cell, cut into nrs row splits x nxt tiles of 64 x.
// rows_hint > 0: patches of about that many rows instead (the ensemble atom kernel, wbx_ens_atoms.hpp: a row of a patch is a
// 64-point tile of ~2.4 us there, so its patches are short).
// The tapered split table: the Br rows are dealt to eight segments (one per XCD, see BinnedArgs::split_br) with the SAME number
// of splits each -- patches of `s0` Br rows for the first ~70 % of a segment, then half, then quarter size.  -> false when the
// table would not fit (the caller keeps uniform splits).
inline bool patch_taper(BinnedArgs& g, int64_t nBr, int64_t s0) {
  if (s0 < 4 || nBr < 8 * 2 * s0) return false;
  const int64_t longest = (nBr + 7) / 8;
  std::vector<int64_t> sizes;  // of the longest segment
  int64_t done = 0;
  const int64_t half = s0 / 2, quarter = s0 / 4;
  while ((done + s0) * 10 <= longest * 7) sizes.push_back(s0), done += s0;
  while ((done + half) * 10 <= longest * 9) sizes.push_back(half), done += half;
  while (done + quarter <= longest) sizes.push_back(quarter), done += quarter;
  if (done < longest) sizes.push_back(longest - done);
  if (sizes.empty() || sizes[0] < 2) return false;
  const int64_t per = (int64_t)sizes.size();
  if (per * 8 > WBX_SPLIT_MAX) return false;
  int64_t at = 0;
  for (int seg = 0; seg < 8; ++seg) {
    int64_t len = nBr / 8 + (seg < nBr % 8 ? 1 : 0);
    int64_t shave = longest - len;  // 0 or 1 (or more when nBr % 8 == 0 ... never: longest = ceil)
    for (int64_t i = 0; i < per; ++i) {
      int64_t n = sizes[(size_t)i];
      if (i == 0) n -= shave;
      g.split_br[seg * per + i] = (int32_t)at;
      at += n;
    }
  }
  g.split_br[8 * per] = (int32_t)at;
  if (at != nBr) return false;
  g.nrs = (int)(8 * per);
  g.taper = 1;
  return true;
}

inline void patch_geometry(BinnedArgs& g, int64_t cells, int64_t nBk, int64_t nBr, int64_t nj, int64_t D, int64_t nx,
                           int64_t rows_hint = 0, bool taper = false) {
  g.taper = 0;
  g.nBk = nBk;
  g.nBr = nBr;
  g.nj = nj;
  const int64_t rows = nBr * D;
  g.nxt = (int)((nx + 63) / 64);
  // row splits: enough waves to fill the chip several times over, but >= 64 rows per patch where the data allows it
  // (the lane fold at the end of a patch costs about as much as 10 rows)
  // (measured on the public-benchmark chunk: 3 splits of 241 rows 0.39 ms, 5 of 145 rows 0.41-0.42 ms on 1440-point rows --
  // every patch and every 64-row batch pays a setup; rows that are not whole 128-byte lines (721 points) keep the shorter
  // patches: their neighbouring x tiles share boundary lines, which only hit in L2 while the tiles walk the same rows at
  // about the same time -- 5 splits of 288 rows fetched 1.21 x the algorithmic bytes, 9 of 160 rows 1.11 x)
  static const int64_t target_env = getenv("WBX_BINNED_TARGET_WAVES") ? atol(getenv("WBX_BINNED_TARGET_WAVES")) : 0;
  const int64_t target = target_env > 0 ? target_env : (nx % 32 == 0 ? 8192 : 16384);
  int64_t want = (target + cells * g.nxt - 1) / (cells * g.nxt);
  if (want > (rows + 63) / 64) want = (rows + 63) / 64;
  if (want < 1) want = 1;
  if (rows_hint > 0) want = (rows + rows_hint - 1) / rows_hint;
  g.rows_per_split = (rows + want - 1) / want;
  g.rows_per_split = (g.rows_per_split + D - 1) / D * D;  // whole Br rows per split: a (bk, br, x) point has ONE patch
  g.nrs = (int)((rows + g.rows_per_split - 1) / g.rows_per_split);
  if (taper && rows_hint > 0) patch_taper(g, nBr, g.rows_per_split / D);
  g.ncell = cells;
  g.nblocks = cells * (int64_t)g.nrs * g.nxt;
  g.order = 0;
}

// The atom tables of a geometry: uni | words | nwords | aid, each padded to 8 bytes.  They depend on the membership
// bits and the geometry only -- not on the data -- so a caller that runs many chunks with the same bins computes them
// once (wbx_binned_atoms) and hands them to every wbx_det_binned call.
inline size_t atoms_carve(BinnedArgs& g, void* base) {
  const size_t npatch = (size_t)g.nrs * g.nxt;
  const size_t n_uni = (size_t)g.nBk * npatch;
  const size_t n_words = (size_t)g.nBk * npatch * ATOM_MAX;
  const size_t n_nwords = ((size_t)g.nBk * npatch + 1) / 2;          // int32 pairs, in 8-byte units
  const size_t n_aid = ((size_t)g.nBk * g.nBr * g.nj + 7) / 8;       // bytes, in 8-byte units
  const size_t n_split = g.taper ? (WBX_SPLIT_MAX + 2) / 2 : 0;      // int32 pairs
  if (base) {
    g.uni = reinterpret_cast<unsigned long long*>(base);
    g.words = g.uni + n_uni;
    g.nwords = reinterpret_cast<int32_t*>(g.words + n_words);
    g.aid = reinterpret_cast<uint8_t*>(g.words + n_words + n_nwords);
    g.split_tab = g.taper ? reinterpret_cast<int32_t*>(g.words + n_words + n_nwords + n_aid) : nullptr;
  }
  return (n_uni + n_words + n_nwords + n_aid + n_split) * 8;
}

inline int atoms_launch(wbx_ctx* ctx, BinnedArgs& g, const uint64_t* bits, int64_t D, int64_t nx, bool atoms) {
  g.bits = reinterpret_cast<const unsigned long long*>(bits);
  g.atoms = atoms ? 1 : 0;
  hipLaunchKernelGGL(binned_atoms_kernel, dim3((unsigned)(g.nBk * g.nrs * g.nxt)), dim3(64), 0, ctx->stream, g, D, nx);
  WBX_HIP(hipGetLastError());
  return 0;
}

// Scratch (tmp | poison [| atom tables]) and, unless the caller brought prepared atom tables, the atom pre-kernel.
// nacc = accumulated lanes.
// extra_doubles > 0: that many more doubles of scratch per patch launch, handed back through extra_out (not initialised).
inline int patch_setup(wbx_ctx* ctx, BinnedArgs& g, const double* wt, const uint64_t* bits, int64_t cells, int64_t nBk,
                       int64_t nBr, int64_t nj, int64_t D, int64_t nx, int nacc, int nbin, bool atoms = false,
                       const void* prepared = nullptr, bool tmp_written_by_kernels = false, int64_t rows_hint = 0,
                       size_t extra_doubles = 0, double** extra_out = nullptr, bool taper = false) {
  g.wt = wt;
  g.bits = reinterpret_cast<const unsigned long long*>(bits);
  g.nbin = nbin;
  g.aidm = nullptr;
  patch_geometry(g, cells, nBk, nBr, nj, D, nx, rows_hint, taper);
  const int64_t npatch = (int64_t)g.nrs * g.nxt;
  const size_t n_tmp = (size_t)cells * npatch * nacc * nbin, n_poison = (size_t)cells * npatch * nacc;
  const size_t need = (n_tmp + n_poison + extra_doubles) * sizeof(double) + (prepared ? 0 : atoms_carve(g, nullptr));
  if (ctx->s2_scratch_size < need) {
    if (ctx->s2_scratch) {
      WBX_HIP(hipStreamSynchronize(ctx->stream));
      WBX_HIP(hipFree(ctx->s2_scratch));
      ctx->s2_scratch = nullptr;
      ctx->s2_scratch_size = 0;
    }
    WBX_HIP(hipMalloc(&ctx->s2_scratch, need));
    ctx->s2_scratch_size = need;
  }
  g.tmp = reinterpret_cast<double*>(ctx->s2_scratch);
  g.tmp_poison = g.tmp + n_tmp;
  if (extra_out) *extra_out = g.tmp_poison + n_poison;
  // (det_atoms_kernel writes every bin of every patch itself -- zeros for the patches it leaves to the slot kernel --, so
  // the 29 MB memset of a public-benchmark chunk, 12 us + a dependency gap in front of a 0.39 ms kernel, is not needed there)
  if (!tmp_written_by_kernels) WBX_HIP(hipMemsetAsync(g.tmp, 0, n_tmp * sizeof(double), ctx->stream));
  WBX_REQUIRE((g.nblocks + 7) / 8 * 8 < (int64_t)1 << 31, "patch grid too large");
  if (prepared) {
    atoms_carve(g, const_cast<void*>(prepared));
    g.atoms = 1;
    return 0;
  }
  atoms_carve(g, g.tmp_poison + n_poison + extra_doubles);
  return atoms_launch(ctx, g, bits, D, nx, atoms);
}

inline int patch_finish(wbx_ctx* ctx, const BinnedArgs& g, int nacc, double* out, bool accumulate = false) {
  hipLaunchKernelGGL(det_binned_finish, dim3((unsigned)(g.ncell * nacc)), dim3(256), 0, ctx->stream,
                     (int64_t)g.nrs * g.nxt, nacc, (int)g.nbin, g.tmp, g.tmp_poison, out, accumulate ? 1 : 0);
  WBX_HIP(hipGetLastError());
  return 0;
}

// Block -> (cell, x tile, row split).  Workgroup i runs on XCD i % 8.  Logical ids are dealt so that each XCD walks a
// contiguous range in order, with the cell as the fastest index: the waves resident on one XCD at a time are the same
// patch of many cells, so the patch's wt / bits rows (which do not depend on A) are fetched into that XCD's L2 once
// and hit by the others.  Without this the 16 B / point of wt + bits miss L2 for every cell and cost as much fabric
// bandwidth as the data.
// WPB waves per block: the block is WPB ADJACENT x tiles of the same cell and rows, one wave each (xt = WPB * xq + wave).
// The waves are independent (own slots, own sweeps) but start together and walk the same rows, so a row's WPB x 256 B
// are requested at about the same time.  (Tried for DRAM page locality with WPB = 4; measured slower than lone waves,
// see wbx_det_binned.hip -- both patch kernels run WPB = 1.)  grid = patch_grid<WPB>(g).
// blocks per XCD of patch_grid<WPB>(g): the tickets of a persistent kernel's per-XCD queue (ens_atoms_kernel)
template <int WPB>
__device__ __forceinline__ uint32_t patch_per_xcd(const BinnedArgs& g) {
  const uint32_t nxq = ((uint32_t)g.nxt + WPB - 1) / WPB;
  return ((uint32_t)g.ncell * nxq * (uint32_t)g.nrs + 7u) >> 3;
}

// `vb`: the block index to decode -- blockIdx.x, or the virtual one a persistent wave drew from a queue
template <int WPB>
__device__ __forceinline__ bool patch_decode(const BinnedArgs& g, int64_t& cell, int& xt, int& rs, uint32_t vb) {
  // 32-bit arithmetic (the launcher checks nblocks < 2^31): a 64-bit divide is a ~100-instruction sequence on this ISA,
  // and a patch is only a few thousand instructions long
  const uint32_t ncell = (uint32_t)g.ncell, nxt = (uint32_t)g.nxt;
  const uint32_t nxq = (nxt + WPB - 1) / WPB;
  const uint32_t nblocks = ncell * nxq * (uint32_t)g.nrs;
  const uint32_t per_xcd = (nblocks + 7u) >> 3;
  uint32_t b = (vb & 7u) * per_xcd + (vb >> 3);
  if (b >= nblocks) return false;
  const uint32_t wave = WPB > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0u;
  if (g.order == 1) {
    // x tile fastest: the tiles of one (cell, row range) run side by side on one XCD.  Rows that are not a multiple of
    // the 128-byte line long (721 floats) make neighbouring tiles share their boundary lines; with the cell-fastest
    // order the neighbour comes ~20 MB of streamed data later and fetches the line again (measured: 1.53x the
    // algorithmic bytes on latitude-fastest chunks).  The atom kernel reads 1 byte of membership per point, so it does
    // not need the cell-fastest order for its W operand any more.
    const uint32_t q = b / nxq;
    const uint32_t x = (b - q * nxq) * WPB + wave;
    const uint32_t q2 = q / ncell;
    cell = (int64_t)(q - q2 * ncell);
    if (x >= nxt) return false;
    xt = (int)x;
    rs = (int)q2;
    return true;
  }
  const uint32_t q = b / ncell;
  cell = (int64_t)(b - q * ncell);
  const uint32_t q2 = q / nxq;
  const uint32_t x = (q - q2 * nxq) * WPB + wave;
  if (x >= nxt) return false;
  xt = (int)x;
  rs = (int)q2;
  return true;
}

template <int WPB>
__device__ __forceinline__ bool patch_decode(const BinnedArgs& g, int64_t& cell, int& xt, int& rs) {
  return patch_decode<WPB>(g, cell, xt, rs, blockIdx.x);
}

template <int WPB>
inline int64_t patch_grid(const BinnedArgs& g) {
  const int64_t nxq = (g.nxt + WPB - 1) / WPB;
  return (g.ncell * nxq * g.nrs + 7) / 8 * 8;
}

}  // namespace wbx
