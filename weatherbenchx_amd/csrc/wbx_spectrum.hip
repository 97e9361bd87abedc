// Zonal energy spectrum: batched 1-D R2C rocFFT along longitude + HIP |F|^2 reduction.
//
// No reference implementation exists in this snapshot -> "parity unpinned"; the definition is
// fixed in include/wbx.h and pinned by analytic tests (Parseval, single sinusoid, constant field).
// Rows are transformed in place through rocFFT's strided layout (no gather copy), in tiles of rows so the
// complex scratch stays bounded; each tile's |F_k|^2 is scaled and added to its group's spectrum.
#include <rocfft/rocfft.h>

#include <map>
#include <tuple>

#include "wbx_common.hpp"

namespace wbx {

struct FftPlan {
  rocfft_plan plan = nullptr;
  rocfft_execution_info info = nullptr;
  void* work = nullptr;
  size_t work_size = 0;
};

struct FftState {
  std::map<std::tuple<int, int64_t, int64_t, int64_t>, FftPlan> plans;  // (nlon, lon_stride, row_stride, batch)
  void* scratch = nullptr;  // complex tile
  size_t scratch_size = 0;
  bool setup = false;
};

static void destroy_plan(FftPlan& p) {
  if (p.info) rocfft_execution_info_destroy(p.info);
  if (p.plan) rocfft_plan_destroy(p.plan);
  if (p.work) (void)hipFree(p.work);
  p = FftPlan();
}

void spectrum_release(wbx_ctx* ctx) {
  auto* st = reinterpret_cast<FftState*>(ctx->fft_state);
  if (!st) return;
  for (auto& kv : st->plans) destroy_plan(kv.second);
  if (st->scratch) (void)hipFree(st->scratch);
  if (st->setup) rocfft_cleanup();
  delete st;
  ctx->fft_state = nullptr;
}

#define WBX_FFT(expr)                                                                              \
  do {                                                                                             \
    rocfft_status _s = (expr);                                                                     \
    if (_s != rocfft_status_success) return fail(WBX_ERR_FFT, "%s failed with rocfft_status %d", #expr, (int)_s); \
  } while (0)

static int get_plan(wbx_ctx* ctx, FftState* st, int nlon, int64_t lon_stride, int64_t row_stride, int64_t batch,
                    FftPlan** out) {
  auto key = std::make_tuple(nlon, lon_stride, row_stride, batch);
  auto it = st->plans.find(key);
  if (it != st->plans.end()) {
    *out = &it->second;
    return 0;
  }
  FftPlan p;
  rocfft_plan_description desc = nullptr;
  WBX_FFT(rocfft_plan_description_create(&desc));
  const size_t in_stride = (size_t)lon_stride, out_stride = 1;
  const size_t nk = (size_t)nlon / 2 + 1;
  WBX_FFT(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                  nullptr, nullptr, 1, &in_stride, (size_t)row_stride, 1, &out_stride, nk));
  const size_t length = (size_t)nlon;
  WBX_FFT(rocfft_plan_create(&p.plan, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                             rocfft_precision_single, 1, &length, (size_t)batch, desc));
  WBX_FFT(rocfft_plan_description_destroy(desc));
  WBX_FFT(rocfft_execution_info_create(&p.info));
  WBX_FFT(rocfft_plan_get_work_buffer_size(p.plan, &p.work_size));
  if (p.work_size) {
    WBX_HIP(hipMalloc(&p.work, p.work_size));
    WBX_FFT(rocfft_execution_info_set_work_buffer(p.info, p.work, p.work_size));
  }
  WBX_FFT(rocfft_execution_info_set_stream(p.info, ctx->stream));
  st->plans[key] = p;
  *out = &st->plans[key];
  return 0;
}

// One block = 256 wavenumbers x a run of rows; consecutive rows of the same group are summed in registers and
// flushed with one fp64 atomic per (group change, k).  F is [rows][nk] interleaved complex (coalesced along k).
__global__ void __launch_bounds__(256) power_kernel(const float2* __restrict__ F, int64_t row0, int64_t nrows_tile,
                                                    int rows_per_block, int nk, int nlon,
                                                    const int32_t* __restrict__ group, const double* __restrict__ scale,
                                                    double* __restrict__ power) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r_begin = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r_end = r_begin + rows_per_block < nrows_tile ? r_begin + rows_per_block : nrows_tile;
  if (k >= nk || r_begin >= r_end) return;
  const double norm = 1.0 / ((double)nlon * (double)nlon) * (k == 0 ? 1.0 : 2.0);
  int32_t cur = group[row0 + r_begin];
  double acc = 0.0;
  for (int64_t r = r_begin; r < r_end; ++r) {
    const int32_t g = group[row0 + r];
    if (g != cur) {
      unsafeAtomicAdd(&power[(int64_t)cur * nk + k], acc);
      acc = 0.0;
      cur = g;
    }
    const float2 f = F[r * nk + k];
    const double re = (double)f.x, im = (double)f.y;
    acc += (re * re + im * im) * norm * scale[row0 + r];
  }
  unsafeAtomicAdd(&power[(int64_t)cur * nk + k], acc);
}

}  // namespace wbx

extern "C" int wbx_zonal_spectrum(wbx_ctx* ctx, const float* field, int64_t lon_stride, int64_t row_stride,
                                  int64_t nrows, int32_t nlon, const int32_t* group, const double* scale,
                                  int32_t ngroup, int32_t accumulate, double* power_out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  WBX_REQUIRE(nlon >= 2 && nrows >= 0 && ngroup >= 0, "bad spectrum extents (nlon=%d, nrows=%lld)", nlon, (long long)nrows);
  WBX_REQUIRE(lon_stride >= 1 && row_stride >= 1, "strides must be positive");
  const int nk = nlon / 2 + 1;
  WBX_REQUIRE(power_out != nullptr || ngroup == 0, "power_out is NULL");
  WBX_HIP(hipSetDevice(ctx->device));
  if (!accumulate && ngroup > 0)
    WBX_HIP(hipMemsetAsync(power_out, 0, (size_t)ngroup * nk * sizeof(double), ctx->stream));
  if (nrows == 0) return 0;
  WBX_REQUIRE(field && group && scale, "field/group/scale is NULL");
  auto* st = reinterpret_cast<FftState*>(ctx->fft_state);
  if (!st) {
    st = new FftState();
    ctx->fft_state = st;
    WBX_FFT(rocfft_setup());
    st->setup = true;
  }
  // tile of rows: <= 256 MiB of complex scratch
  int64_t tile = ((int64_t)256 << 20) / ((int64_t)nk * 8);
  if (tile < 1) tile = 1;
  if (tile > nrows) tile = nrows;
  // a strided batch only tiles cleanly when rows are uniformly spaced, which they are by construction
  const size_t need = (size_t)tile * nk * 8;
  if (st->scratch_size < need) {
    if (st->scratch) {
      WBX_HIP(hipStreamSynchronize(ctx->stream));
      WBX_HIP(hipFree(st->scratch));
    }
    WBX_HIP(hipMalloc(&st->scratch, need));
    st->scratch_size = need;
  }
  for (int64_t r0 = 0; r0 < nrows; r0 += tile) {
    const int64_t n = r0 + tile <= nrows ? tile : nrows - r0;
    FftPlan* plan = nullptr;
    if (int rc = get_plan(ctx, st, nlon, lon_stride, row_stride, n, &plan)) return rc;
    void* in = const_cast<float*>(field + r0 * row_stride);
    void* out = st->scratch;
    WBX_FFT(rocfft_execute(plan->plan, &in, &out, plan->info));
    const int rows_per_block = 64;
    dim3 grid((nk + 255) / 256, (unsigned)((n + rows_per_block - 1) / rows_per_block));
    hipLaunchKernelGGL(power_kernel, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const float2*>(st->scratch), r0, n,
                       rows_per_block, nk, nlon, group, scale, power_out);
    WBX_HIP(hipGetLastError());
  }
  return 0;
}
