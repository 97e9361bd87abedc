// Device-resident accumulators and the small stream plumbing around them (see include/wbx.h, "accumulators").
//
// The reference sums per-chunk AggregationStates with beam.CombinePerKey(CombiningSum()) on the workers' hosts
// (beam_pipeline.py:509-510, beam_utils.py:30-50).  Here a chunk's stage-2 / binned output never leaves HBM: it is
// added into a persistent fp64 accumulator right behind the kernel that produced it, the accumulators of a rank are
// all-reduced in place (RCCL) and read back once per job.
#include "wbx_common.hpp"

namespace wbx {

// acc[i] = (overwrite ? 0 : acc[i]) + src[i]; n is KBs..MBs: one element per thread, grid-stride
static __global__ void __launch_bounds__(256) acc_add_kernel(double* __restrict__ acc, const double* __restrict__ src,
                                                             int64_t n, int overwrite) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double s = src[i];
    acc[i] = overwrite ? s : acc[i] + s;
  }
}

// valid[i] = !isnan(x[i]) as one byte per element (data_loaders/base.py:25-56: the `mask` coordinate, True = valid);
// four elements per thread so that the mask is written as dwords
template <typename T>
static __global__ void __launch_bounds__(256) notnan_kernel(const T* __restrict__ x, int64_t n, uint8_t* __restrict__ valid) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const T v = x[4 * i + k];
      w |= (v == v ? 1u : 0u) << (8 * k);
    }
    reinterpret_cast<uint32_t*>(valid)[i] = w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (n4 << 2) + threadIdx.x;
    const T v = x[i];
    valid[i] = v == v ? 1 : 0;
  }
}

}  // namespace wbx

extern "C" int wbx_acc_add(wbx_ctx* ctx, double* acc, const double* src, int64_t n, int32_t overwrite) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  WBX_REQUIRE(n >= 0, "n must be >= 0");
  if (n == 0) return 0;
  WBX_REQUIRE(acc != nullptr && src != nullptr, "NULL pointer");
  WBX_HIP(hipSetDevice(ctx->device));
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(acc_add_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, acc, src, n, (int)overwrite);
  WBX_HIP(hipGetLastError());
  return 0;
}

extern "C" int wbx_notnan_mask(wbx_ctx* ctx, const void* data, int dtype, int64_t n, uint8_t* valid_out) {
  using namespace wbx;
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  WBX_REQUIRE(n >= 0, "n must be >= 0");
  if (n == 0) return 0;
  WBX_REQUIRE(data != nullptr && valid_out != nullptr, "NULL pointer");
  WBX_REQUIRE(reinterpret_cast<uintptr_t>(valid_out) % 4 == 0, "valid_out must be 4-byte aligned");
  WBX_HIP(hipSetDevice(ctx->device));
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 16384) blocks = 16384;
  if (dtype == WBX_F32)
    hipLaunchKernelGGL((notnan_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float*>(data), n, valid_out);
  else if (dtype == WBX_F64)
    hipLaunchKernelGGL((notnan_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const double*>(data), n, valid_out);
  else
    return fail(WBX_ERR_INVALID, "unknown dtype %d", dtype);
  WBX_HIP(hipGetLastError());
  return 0;
}

namespace wbx {
void spectrum_note_write(const void* dst, size_t bytes);  // wbx_spectrum.hip
}

extern "C" int wbx_memcpy_d2d(wbx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (bytes == 0) return 0;
  WBX_REQUIRE(dst != nullptr && src != nullptr, "NULL pointer");
  WBX_HIP(hipSetDevice(ctx->device));
  wbx::spectrum_note_write(dst, bytes);
  WBX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

extern "C" int wbx_memcpy_h2d_async(wbx_ctx* ctx, void* dptr, const void* h_pinned, size_t bytes) {
  WBX_REQUIRE(ctx != nullptr, "ctx is NULL");
  if (bytes == 0) return 0;
  WBX_REQUIRE(dptr != nullptr && h_pinned != nullptr, "NULL pointer");
  WBX_HIP(hipSetDevice(ctx->device));
  wbx::spectrum_note_write(dptr, bytes);
  WBX_HIP(hipMemcpyAsync(dptr, h_pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
  return 0;
}
