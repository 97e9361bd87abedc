// A validity mask that lives on the W dims folded into the atom-id bytes of the patch kernels (det_atoms_kernel<.., MERGED>,
// ens_atoms_kernel): one byte per point carries both, 255 = masked out.
#pragma once
#include "wbx_patch.hpp"
#include "wbx_s1.hpp"

namespace wbx {

// aidm[bk][br][x] = mask(bk, br, x) ? aid[bk][br][x] : 255, for a mask that does not depend on A or the depth dims (the caller
// says so: WBX_BINNED_MASK_ON_W): addressed through the plan's tables at A = 0, depth row 0.
// twin: masked-out points keep their atom as its TWIN (id | 0x80) instead of 255 (ens_atoms_kernel's twin mode).
static __global__ void __launch_bounds__(256) aid_merge_kernel(S1Args a, BinnedArgs g, uint8_t* __restrict__ aidm, int twin) {
  const int64_t row = blockIdx.x;  // (bk, br)
  const int64_t bk = row / g.nBr, br = row - bk * g.nBr;
  const int64_t key = bk * g.nBr + br;  // A = 0
  const int64_t base = (a.key_off[3] ? a.key_off[3][key] : 0) + (a.depth_off[3] ? a.depth_off[3][0] : 0);
  const uint8_t* m = reinterpret_cast<const uint8_t*>(a.in[3]) + base;
  for (int64_t x = threadIdx.x; x < g.nj; x += blockDim.x) {
    const int64_t i = row * g.nj + x;
    aidm[i] = m[x * a.xstride[3]] != 0 ? g.aid[i] : (twin ? (uint8_t)(g.aid[i] | 0x80) : (uint8_t)255);
  }
}

// The same for a mask with strides along A and / or the depth dims -- what add_nan_mask_to_data builds
// (data_loaders/base.py:25-56: `mask = ~isnan(data)`, every dim of the data): ONE id byte per POINT of the chunk,
// idp[cell][r = br * D + d][x], written once per call in front of the sweep (1 mask byte + 1 id byte read, 1 byte written per
// point beside the 208 the sweep reads).  The sweep then reads its id byte from row (cell, r) instead of row (bk, br).
static __global__ void __launch_bounds__(256) aid_merge_points_kernel(S1Args a, BinnedArgs g, int64_t D, uint8_t* __restrict__ idp, int twin) {
  const int64_t R = g.nBr * D;
  const int64_t rowi = blockIdx.x;  // (cell, br, d)
  const int64_t cell = rowi / R, r = rowi - cell * R;
  const int64_t br = r / D, d = r - br * D;
  const int64_t bk = cell % g.nBk;
  const int64_t key = cell * g.nBr + br;  // ((A * nBk) + bk) * nBr + br
  const int64_t base = (a.key_off[3] ? a.key_off[3][key] : 0) + (a.depth_off[3] ? a.depth_off[3][d] : 0);
  const uint8_t* m = reinterpret_cast<const uint8_t*>(a.in[3]) + base;
  const uint8_t* src = g.aid + (bk * g.nBr + br) * g.nj;
  uint8_t* dst = idp + rowi * g.nj;
  for (int64_t x = threadIdx.x; x < g.nj; x += blockDim.x) {
    const uint8_t id = src[x];
    dst[x] = m[x * a.xstride[3]] != 0 ? id : (twin ? (uint8_t)(id | 0x80) : (uint8_t)255);
  }
}

inline int aidm_scratch(wbx_ctx* ctx, size_t need) {
  if (ctx->aidm_scratch_size < need) {
    if (ctx->aidm_scratch) {
      WBX_HIP(hipStreamSynchronize(ctx->stream));
      WBX_HIP(hipFree(ctx->aidm_scratch));
      ctx->aidm_scratch = nullptr;
      ctx->aidm_scratch_size = 0;
    }
    WBX_HIP(hipMalloc(&ctx->aidm_scratch, need));
    ctx->aidm_scratch_size = need;
  }
  return 0;
}

// -> g.aidm = idp[cell][r][x] (see aid_merge_points_kernel)
inline int merge_point_mask_into_atom_ids(wbx_ctx* ctx, const S1Args& a, BinnedArgs& g, int64_t D, bool twin) {
  const int64_t rows = g.ncell * g.nBr * D;
  WBX_REQUIRE(rows < ((int64_t)1 << 31), "too many rows for the mask merge (%lld)", (long long)rows);
  if (int rc = aidm_scratch(ctx, (size_t)rows * (size_t)g.nj)) return rc;
  hipLaunchKernelGGL(aid_merge_points_kernel, dim3((unsigned)rows), dim3(256), 0, ctx->stream, a, g, D,
                     reinterpret_cast<uint8_t*>(ctx->aidm_scratch), twin ? 1 : 0);
  WBX_HIP(hipGetLastError());
  g.aidm = reinterpret_cast<const uint8_t*>(ctx->aidm_scratch);
  return 0;
}

// Launches the merge into the context's scratch (grown on demand) and points g.aidm at it.
inline int merge_mask_into_atom_ids(wbx_ctx* ctx, const S1Args& a, BinnedArgs& g, bool twin = false) {
  const size_t need = (size_t)(g.nBk * g.nBr * g.nj);
  if (ctx->aidm_scratch_size < need) {
    if (ctx->aidm_scratch) {
      WBX_HIP(hipStreamSynchronize(ctx->stream));
      WBX_HIP(hipFree(ctx->aidm_scratch));
      ctx->aidm_scratch = nullptr;
      ctx->aidm_scratch_size = 0;
    }
    WBX_HIP(hipMalloc(&ctx->aidm_scratch, need));
    ctx->aidm_scratch_size = need;
  }
  hipLaunchKernelGGL(aid_merge_kernel, dim3((unsigned)(g.nBk * g.nBr)), dim3(256), 0, ctx->stream, a, g,
                     reinterpret_cast<uint8_t*>(ctx->aidm_scratch), twin ? 1 : 0);
  WBX_HIP(hipGetLastError());
  g.aidm = reinterpret_cast<const uint8_t*>(ctx->aidm_scratch);
  return 0;
}

}  // namespace wbx
