// Stage 2: weighted / binned contraction of the stage-1 partial sums.
//
// Replaces the weight- and bin-mask operands of the reference's single xr.dot
// (weatherbenchX/aggregation.py:311-335): W = prod(weights) * prod(bin masks) is built once per
// (aggregator, grid) on the host in float64 (O(lat*lon*bins), data independent) and contracted here.
// The products are plain multiply-adds with no zero skipping, so a NaN partial poisons every bin
// (NaN * 0 = NaN) exactly as documented at aggregation.py:272-277.
//
//   partial [nA][nBk][nBr][nchunk][nlane][nj]      W [nBk][nBr][nj][nbin]
//   sum_j=1: out[nA][nBk][nlane][nbin]     = sum_{Br,chunk,j}
//   sum_j=0: out[nA][nBk][nlane][nj][nbin] = sum_{Br,chunk}
// Summation order is fixed by the launch geometry, so results are run-to-run reproducible.
#include "wbx_s1.hpp"

namespace wbx {

constexpr int BG = 8;  // bins per register group

// one block per (A, Bk, lane[, j]); threads stride over the flattened (Br, chunk[, j]) contraction.
// fixed_j = 0: j is contracted (sum_j).  fixed_j = 1: j is kept and few (< 64): one block per j.
// nsplit > 1 (long rows only): the (Br, chunk) rows of one output are dealt to nsplit blocks, out is
// tmp[output][split][bin] and s2_bits_finish_kernel adds the splits -- a handful of outputs (8 leads x 5 ensemble lanes)
// over a 118 MB partial otherwise ran on 40 of the 256 CUs.
__global__ void __launch_bounds__(256) s2_reduce_kernel(wbx_s2_plan p, int fixed_j, int nsplit,
                                                        const double* __restrict__ partial,
                                                        const double* __restrict__ W, double* __restrict__ out) {
  int64_t b = blockIdx.x;
  const int split = (int)(b % nsplit);
  b /= nsplit;
  int64_t jf = 0;
  if (fixed_j) {
    jf = b % p.nj;
    b /= p.nj;
  }
  const int64_t lane = b % p.nlane;
  b /= p.nlane;
  const int64_t bk = b % p.nBk;
  const int64_t A = b / p.nBk;
  const int64_t njc = fixed_j ? 1 : p.nj;
  const int64_t ncj = p.nchunk * njc;
  const int64_t ncontr = p.nBr * ncj;
  const int tl = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __shared__ double red[4][BG];
  const double* pbase = partial + (A * p.nBk + bk) * p.nBr * p.nchunk * p.nlane * p.nj;
  const double* wbase = W + bk * p.nBr * p.nj * p.nbin;
  for (int64_t b0 = 0; b0 < p.nbin; b0 += BG) {
    double acc[BG];
#pragma unroll
    for (int g = 0; g < BG; ++g) acc[g] = 0.0;
    if (!fixed_j && p.nj >= 64) {
      // long rows (x kept in stage 1, summed here): lanes walk j, rows/chunks are plain loops -- no 64-bit divides
      const int64_t nrc = p.nBr * p.nchunk, per = (nrc + nsplit - 1) / nsplit;
      const int64_t rc0 = split * per, rc1 = rc0 + per < nrc ? rc0 + per : nrc;
      int64_t br = rc0 / p.nchunk, ch = rc0 - br * p.nchunk;  // one divide per block, then carried
      for (int64_t rc = rc0; rc < rc1; ++rc) {
        const double* wrow = wbase + br * p.nj * p.nbin + b0;
        const double* prow = pbase + (rc * p.nlane + lane) * p.nj;
        for (int64_t j = threadIdx.x; j < p.nj; j += blockDim.x) {
          const double v = prow[j];
          const double* w = wrow + j * p.nbin;
#pragma unroll
          for (int g = 0; g < BG; ++g)
            if (b0 + g < p.nbin) acc[g] += v * w[g];
        }
        if (++ch == p.nchunk) {
          ch = 0;
          ++br;
        }
      }
    } else {
      for (int64_t c = threadIdx.x; c < ncontr; c += blockDim.x) {
        const int64_t br = c / ncj;
        const int64_t r = c - br * ncj;
        const int64_t ch = r / njc;
        const int64_t j = fixed_j ? jf : r - ch * njc;
        const double v = pbase[((br * p.nchunk + ch) * p.nlane + lane) * p.nj + j];
        const double* w = wbase + (br * p.nj + j) * p.nbin + b0;
#pragma unroll
        for (int g = 0; g < BG; ++g)
          if (b0 + g < p.nbin) acc[g] += v * w[g];
      }
    }
#pragma unroll
    for (int g = 0; g < BG; ++g) {
      double s = wave_sum(acc[g]);
      if (tl == 0) red[wv][g] = s;
    }
    __syncthreads();
    if (threadIdx.x < BG && b0 + threadIdx.x < p.nbin) {
      double s = 0.0;
      for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) s += red[w2][threadIdx.x];
      const int64_t o = fixed_j ? (((A * p.nBk + bk) * p.nlane + lane) * p.nj + jf) : ((A * p.nBk + bk) * p.nlane + lane);
      out[(o * nsplit + split) * p.nbin + b0 + threadIdx.x] = s;
    }
    __syncthreads();
  }
}

// thread per j; grid = nA * nBk * nlane * njtile.
__global__ void __launch_bounds__(256) s2_keepj_kernel(wbx_s2_plan p, int njtile, const double* __restrict__ partial,
                                                       const double* __restrict__ W, double* __restrict__ out) {
  int64_t b = blockIdx.x;
  const int64_t jt = b % njtile;
  b /= njtile;
  const int64_t lane = b % p.nlane;
  b /= p.nlane;
  const int64_t bk = b % p.nBk;
  const int64_t A = b / p.nBk;
  const int64_t j = jt * blockDim.x + threadIdx.x;
  if (j >= p.nj) return;
  const double* pbase = partial + (A * p.nBk + bk) * p.nBr * p.nchunk * p.nlane * p.nj;
  const double* wbase = W + bk * p.nBr * p.nj * p.nbin;
  double* obase = out + ((((A * p.nBk + bk) * p.nlane + lane) * p.nj) + j) * p.nbin;
  for (int64_t b0 = 0; b0 < p.nbin; b0 += BG) {
    double acc[BG];
#pragma unroll
    for (int g = 0; g < BG; ++g) acc[g] = 0.0;
    for (int64_t br = 0; br < p.nBr; ++br) {
      const double* w = wbase + (br * p.nj + j) * p.nbin + b0;
      for (int64_t ch = 0; ch < p.nchunk; ++ch) {
        const double v = pbase[((br * p.nchunk + ch) * p.nlane + lane) * p.nj + j];
#pragma unroll
        for (int g = 0; g < BG; ++g)
          if (b0 + g < p.nbin) acc[g] += v * w[g];
      }
    }
#pragma unroll
    for (int g = 0; g < BG; ++g)
      if (b0 + g < p.nbin) obase[b0 + g] = acc[g];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Bit-mask form of the same contraction, for W = wt[Bk][Br][j] * mask[Bk][Br][j][bin] with boolean masks and at most
// 64 bins (Regions / LandSea binning of weatherbenchX/binning.py:92-201 times GridAreaWeighting): the dense W of the
// public benchmark's 34 bins at 0.25 degree would be 282 MB and cost 34 multiply-adds per element; here one uint64
// carries a point's membership in every bin.  Per element: m = p * wt once, then per bin 4 integer/fp ops
// (v_bfe_i32 -> all-ones/zero, two v_and, v_add_f64).  NaN * 0 = NaN is kept by a poison term (m * 0) that is
// added to EVERY bin, exactly like the reference's xr.dot (aggregation.py:272-277).
template <int NB>
__global__ void __launch_bounds__(256) s2_bits_kernel(wbx_s2_plan p, int rows_per_block, int nsplit,
                                                      const double* __restrict__ partial,
                                                      const double* __restrict__ wt,
                                                      const unsigned long long* __restrict__ bits,
                                                      double* __restrict__ out) {
  // sum_j = 1: grid = nA*nBk*nlane*nsplit; block = rows [split*rows_per_block, ...) x all j; out = tmp[row][split][bin]
  // sum_j = 0: grid = nA*nBk*nlane*njtile (nsplit = njtile); block = all rows x 256 j;       out = final [..][j][bin]
  int64_t b = blockIdx.x;
  const int split = (int)(b % nsplit);
  b /= nsplit;
  const int64_t lane = b % p.nlane;
  b /= p.nlane;
  const int64_t bk = b % p.nBk;
  const int64_t A = b / p.nBk;
  const double* pbase = partial + (A * p.nBk + bk) * p.nBr * p.nchunk * p.nlane * p.nj;
  const double* wbase = wt + bk * p.nBr * p.nj;
  const unsigned long long* bbase = bits + bk * p.nBr * p.nj;
  double acc[NB];
#pragma unroll
  for (int g = 0; g < NB; ++g) acc[g] = 0.0;
  double poison = 0.0;
  const int64_t r0 = p.sum_j ? (int64_t)split * rows_per_block : 0;
  const int64_t r1 = p.sum_j ? (r0 + rows_per_block < p.nBr ? r0 + rows_per_block : p.nBr) : p.nBr;
  const int64_t j0 = p.sum_j ? threadIdx.x : (int64_t)split * blockDim.x + threadIdx.x;
  const int64_t jstep = p.sum_j ? blockDim.x : p.nj;  // sum_j = 0: exactly one j per thread
  for (int64_t br = r0; br < r1; ++br) {
    for (int64_t j = j0; j < p.nj; j += jstep) {
      double v = 0.0;
      for (int64_t ch = 0; ch < p.nchunk; ++ch) v += pbase[((br * p.nchunk + ch) * p.nlane + lane) * p.nj + j];
      const double m = v * wbase[br * p.nj + j];
      poison = fma(m, 0.0, poison);
      const unsigned long long bw = bbase[br * p.nj + j];
      const int lo = (int)(unsigned)bw, hi = (int)(unsigned)(bw >> 32);
      const long long mb = __double_as_longlong(m);
#pragma unroll
      for (int g = 0; g < NB; ++g) {
        const int word = g < 32 ? lo : hi;
        const long long sel = (long long)((word << (31 - (g & 31))) >> 31);  // 0 or -1 (all ones)
        acc[g] += __longlong_as_double(mb & sel);
      }
    }
  }
  if (!p.sum_j) {
    if (j0 < p.nj) {
      double* o = out + ((((A * p.nBk + bk) * p.nlane + lane) * p.nj) + j0) * p.nbin;
      for (int g = 0; g < NB && g < p.nbin; ++g) o[g] = acc[g] + poison;
    }
    return;
  }
  __shared__ double red[4][NB];
  const int tl = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int g = 0; g < NB; ++g) {
    const double s = wave_sum(acc[g] + poison);
    if (tl == 0) red[wv][g] = s;
  }
  __syncthreads();
  if (threadIdx.x < NB && threadIdx.x < p.nbin) {
    double s = 0.0;
    for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) s += red[w2][threadIdx.x];
    out[(((A * p.nBk + bk) * p.nlane + lane) * nsplit + split) * p.nbin + threadIdx.x] = s;
  }
}

// tmp[row][split][bin] -> out[row][bin]
__global__ void __launch_bounds__(256) s2_bits_finish_kernel(int64_t nrow, int nsplit, int nbin,
                                                             const double* __restrict__ tmp, double* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrow * nbin) return;
  const int64_t row = i / nbin;
  const int bin = (int)(i - row * nbin);
  double s = 0.0;
  for (int k = 0; k < nsplit; ++k) s += tmp[(row * nsplit + k) * nbin + bin];
  out[i] = s;
}

bool s2_patch_eligible(const wbx_s2_plan& p);
int s2_patch(wbx_ctx* ctx, const wbx_s2_plan& p, const double* partial, const double* wt, const uint64_t* bits,
             double* out);

}  // namespace wbx

extern "C" int wbx_contract_bits(wbx_ctx* ctx, const wbx_s2_plan* plan, const double* partial, const double* wt,
                                 const uint64_t* bits, double* out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr && plan != nullptr, "ctx/plan is NULL");
  const wbx_s2_plan& p = *plan;
  WBX_REQUIRE(p.nA >= 0 && p.nBk >= 0 && p.nBr >= 0 && p.nchunk >= 0 && p.nlane >= 0 && p.nj >= 0, "negative extent");
  WBX_REQUIRE(p.nbin >= 1 && p.nbin <= 64, "wbx_contract_bits handles 1..64 bins (got %lld)", (long long)p.nbin);
  const int64_t nrow = p.nA * p.nBk * p.nlane;
  const int64_t nout = nrow * (p.sum_j ? 1 : p.nj) * p.nbin;
  if (nout == 0) return 0;
  WBX_REQUIRE(out != nullptr, "out is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  if (p.nBr * p.nchunk * (p.sum_j ? p.nj : 1) == 0) {
    WBX_HIP(hipMemsetAsync(out, 0, (size_t)nout * sizeof(double), ctx->stream));
    return 0;
  }
  WBX_REQUIRE(partial != nullptr && wt != nullptr && bits != nullptr, "partial/wt/bits is NULL");
  if (s2_patch_eligible(p)) return s2_patch(ctx, p, partial, wt, bits, out);  // full-map partials: wbx_s2_patch.hip
  const unsigned long long* b64 = reinterpret_cast<const unsigned long long*>(bits);
  int rows_per_block = 1, nsplit = 1;
  double* dst = out;
  if (p.sum_j) {
    // enough blocks to fill the chip: split the Br rows; each block covers whole rows of nj
    int64_t want = (4096 + nrow - 1) / nrow;
    if (want > p.nBr) want = p.nBr;
    if (want < 1) want = 1;
    rows_per_block = (int)((p.nBr + want - 1) / want);
    nsplit = (int)((p.nBr + rows_per_block - 1) / rows_per_block);
    const size_t need = (size_t)nrow * nsplit * p.nbin * sizeof(double);
    if (ctx->s2_scratch_size < need) {
      if (ctx->s2_scratch) {
        WBX_HIP(hipStreamSynchronize(ctx->stream));
        WBX_HIP(hipFree(ctx->s2_scratch));
      }
      WBX_HIP(hipMalloc(&ctx->s2_scratch, need));
      ctx->s2_scratch_size = need;
    }
    dst = reinterpret_cast<double*>(ctx->s2_scratch);
  } else {
    nsplit = (int)((p.nj + 255) / 256);
  }
  const int64_t grid = nrow * nsplit;
  WBX_REQUIRE(grid < (int64_t)1 << 31, "stage-2 grid too large");
  if (p.nbin <= 16)
    hipLaunchKernelGGL((s2_bits_kernel<16>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, p, rows_per_block, nsplit,
                       partial, wt, b64, dst);
  else if (p.nbin <= 32)
    hipLaunchKernelGGL((s2_bits_kernel<32>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, p, rows_per_block, nsplit,
                       partial, wt, b64, dst);
  else
    hipLaunchKernelGGL((s2_bits_kernel<64>), dim3((unsigned)grid), dim3(256), 0, ctx->stream, p, rows_per_block, nsplit,
                       partial, wt, b64, dst);
  WBX_HIP(hipGetLastError());
  if (p.sum_j) {
    const int64_t n = nrow * p.nbin;
    hipLaunchKernelGGL(s2_bits_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, nrow, nsplit,
                       (int)p.nbin, dst, out);
    WBX_HIP(hipGetLastError());
  }
  return 0;
}

extern "C" int wbx_contract(wbx_ctx* ctx, const wbx_s2_plan* plan, const double* partial, const double* W,
                            double* out) {
  WBX_REQUIRE(ctx != nullptr && plan != nullptr, "ctx/plan is NULL");
  const wbx_s2_plan& p = *plan;
  WBX_REQUIRE(p.nA >= 0 && p.nBk >= 0 && p.nBr >= 0 && p.nchunk >= 0 && p.nlane >= 0 && p.nj >= 0 && p.nbin >= 0,
              "negative extent in stage-2 plan");
  const int64_t nout = p.nA * p.nBk * p.nlane * (p.sum_j ? 1 : p.nj) * p.nbin;
  if (nout == 0) return 0;
  WBX_REQUIRE(out != nullptr, "out is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  if (p.nBr * p.nchunk * (p.sum_j ? p.nj : 1) == 0) {
    WBX_HIP(hipMemsetAsync(out, 0, (size_t)nout * sizeof(double), ctx->stream));
    return 0;
  }
  WBX_REQUIRE(partial != nullptr && W != nullptr, "partial/W is NULL");
  if (p.sum_j || p.nj < 64) {
    const int fixed_j = p.sum_j ? 0 : 1;
    const int64_t grid = p.nA * p.nBk * p.nlane * (fixed_j ? p.nj : 1);
    WBX_REQUIRE(grid < (int64_t)1 << 31, "stage-2 grid too large");
    const int64_t ncontr = p.nBr * p.nchunk * (fixed_j ? 1 : p.nj);
    const int threads = ncontr >= 256 ? 256 : (ncontr >= 128 ? 128 : 64);
    int nsplit = 1;
    double* dst = out;
    const int64_t nrc = p.nBr * p.nchunk;
    if (!fixed_j && p.nj >= 64 && grid < 1024 && nrc * p.nj >= ((int64_t)1 << 16)) {
      int64_t want = (2048 + grid - 1) / grid;
      if (want > nrc) want = nrc;
      nsplit = (int)want;
      const size_t need = (size_t)grid * nsplit * p.nbin * sizeof(double);
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
      dst = reinterpret_cast<double*>(ctx->s2_scratch);
    }
    hipLaunchKernelGGL(wbx::s2_reduce_kernel, dim3((unsigned)(grid * nsplit)), dim3(threads), 0, ctx->stream, p, fixed_j,
                       nsplit, partial, W, dst);
    if (nsplit > 1) {
      WBX_HIP(hipGetLastError());
      const int64_t n = grid * p.nbin;
      hipLaunchKernelGGL(wbx::s2_bits_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, grid,
                         nsplit, (int)p.nbin, dst, out);
    }
  } else {
    const int threads = p.nj >= 256 ? 256 : (p.nj >= 128 ? 128 : 64);
    const int njtile = (int)((p.nj + threads - 1) / threads);
    const int64_t grid = p.nA * p.nBk * p.nlane * njtile;
    WBX_REQUIRE(grid < (int64_t)1 << 31, "stage-2 grid too large");
    hipLaunchKernelGGL(wbx::s2_keepj_kernel, dim3((unsigned)grid), dim3(threads), 0, ctx->stream, p, njtile, partial,
                       W, out);
  }
  WBX_HIP(hipGetLastError());
  return 0;
}
