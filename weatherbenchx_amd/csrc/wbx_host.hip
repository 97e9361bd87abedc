// Host side of the chunk feeder: a chunk's fields change their innermost dim on the way into page-locked memory.
//
// Real archives store [.., longitude, latitude] (weatherbenchX/data_loaders/xarray_loaders.py:185-188, 236-239: the loaders
// hand on whatever order the store has); the zonal transforms want longitude contiguous (csrc/wbx_zspec1440.hpp: 0.60 of the HBM
// peak against 0.38 on latitude-fastest rows, which are bound by L1 line requests whatever the kernel does).  The loader threads
// copy every chunk from the page cache into page-locked memory anyway (loaders.FileLoader._gather): with
// `device_layout='lon_fastest'` that copy is this blocked transposition instead of a memcpy -- the H2D DMA and every kernel then
// see longitude-fastest fields, and no transposed copy is ever made on the device.
//
// dst[b][c][r] = src[b][r][c]: 32 x 32 tiles (a tile's source rows and destination rows are whole 128-byte runs: every cache
// line is touched once per side), 8 x 8 in registers with AVX2 when the CPU has it, scalar edges.  Pure host code: no context,
// no stream; callers run it on several threads over disjoint planes (ctypes releases the GIL).
#include <immintrin.h>

#include <cstdint>
#include <cstring>

#include "wbx_common.hpp"

namespace {

constexpr int64_t TILE = 32;

template <typename T>
void tile_scalar(T* dst, const T* src, int64_t rows, int64_t cols, int64_t r0, int64_t r1, int64_t c0, int64_t c1) {
  for (int64_t c = c0; c < c1; ++c) {
    T* d = dst + c * rows;
    const T* s = src + c;
    for (int64_t r = r0; r < r1; ++r) d[r] = s[r * cols];
  }
}

__attribute__((target("avx2"))) inline void block8x8_avx2(float* dst, const float* src, int64_t rows, int64_t cols) {
  // 8 source rows of 8 floats -> 8 destination rows of 8 floats
  __m256 r0 = _mm256_loadu_ps(src + 0 * cols), r1 = _mm256_loadu_ps(src + 1 * cols);
  __m256 r2 = _mm256_loadu_ps(src + 2 * cols), r3 = _mm256_loadu_ps(src + 3 * cols);
  __m256 r4 = _mm256_loadu_ps(src + 4 * cols), r5 = _mm256_loadu_ps(src + 5 * cols);
  __m256 r6 = _mm256_loadu_ps(src + 6 * cols), r7 = _mm256_loadu_ps(src + 7 * cols);
  __m256 t0 = _mm256_unpacklo_ps(r0, r1), t1 = _mm256_unpackhi_ps(r0, r1);
  __m256 t2 = _mm256_unpacklo_ps(r2, r3), t3 = _mm256_unpackhi_ps(r2, r3);
  __m256 t4 = _mm256_unpacklo_ps(r4, r5), t5 = _mm256_unpackhi_ps(r4, r5);
  __m256 t6 = _mm256_unpacklo_ps(r6, r7), t7 = _mm256_unpackhi_ps(r6, r7);
  __m256 u0 = _mm256_shuffle_ps(t0, t2, 0x44), u1 = _mm256_shuffle_ps(t0, t2, 0xEE);
  __m256 u2 = _mm256_shuffle_ps(t1, t3, 0x44), u3 = _mm256_shuffle_ps(t1, t3, 0xEE);
  __m256 u4 = _mm256_shuffle_ps(t4, t6, 0x44), u5 = _mm256_shuffle_ps(t4, t6, 0xEE);
  __m256 u6 = _mm256_shuffle_ps(t5, t7, 0x44), u7 = _mm256_shuffle_ps(t5, t7, 0xEE);
  _mm256_storeu_ps(dst + 0 * rows, _mm256_permute2f128_ps(u0, u4, 0x20));
  _mm256_storeu_ps(dst + 1 * rows, _mm256_permute2f128_ps(u1, u5, 0x20));
  _mm256_storeu_ps(dst + 2 * rows, _mm256_permute2f128_ps(u2, u6, 0x20));
  _mm256_storeu_ps(dst + 3 * rows, _mm256_permute2f128_ps(u3, u7, 0x20));
  _mm256_storeu_ps(dst + 4 * rows, _mm256_permute2f128_ps(u0, u4, 0x31));
  _mm256_storeu_ps(dst + 5 * rows, _mm256_permute2f128_ps(u1, u5, 0x31));
  _mm256_storeu_ps(dst + 6 * rows, _mm256_permute2f128_ps(u2, u6, 0x31));
  _mm256_storeu_ps(dst + 7 * rows, _mm256_permute2f128_ps(u3, u7, 0x31));
}

__attribute__((target("avx2"))) void plane_f32_avx2(float* dst, const float* src, int64_t rows, int64_t cols) {
  const int64_t rfull = rows / TILE * TILE, cfull = cols / 8 * 8;
  for (int64_t r0 = 0; r0 < rfull; r0 += TILE) {
    for (int64_t c0 = 0; c0 < cfull; c0 += TILE) {
      const int64_t c1 = c0 + TILE < cfull ? c0 + TILE : cfull;
      for (int64_t c = c0; c < c1; c += 8)
        for (int64_t r = r0; r < r0 + TILE; r += 8) block8x8_avx2(dst + c * rows + r, src + r * cols + c, rows, cols);
    }
    if (cfull < cols) tile_scalar(dst, src, rows, cols, r0, r0 + TILE, cfull, cols);
  }
  if (rfull < rows) {
    // the last rows: 8-row blocks while they last, then scalars
    const int64_t r8 = rfull + (rows - rfull) / 8 * 8;
    for (int64_t r = rfull; r < r8; r += 8) {
      for (int64_t c = 0; c < cfull; c += 8) block8x8_avx2(dst + c * rows + r, src + r * cols + c, rows, cols);
      if (cfull < cols) tile_scalar(dst, src, rows, cols, r, r + 8, cfull, cols);
    }
    if (r8 < rows) tile_scalar(dst, src, rows, cols, r8, rows, 0, cols);
  }
}

template <typename T>
void plane_scalar(T* dst, const T* src, int64_t rows, int64_t cols) {
  for (int64_t r0 = 0; r0 < rows; r0 += TILE)
    for (int64_t c0 = 0; c0 < cols; c0 += TILE)
      tile_scalar(dst, src, rows, cols, r0, r0 + TILE < rows ? r0 + TILE : rows, c0, c0 + TILE < cols ? c0 + TILE : cols);
}

}  // namespace

extern "C" int wbx_host_transpose(void* dst, const void* src, int64_t batch, int64_t rows, int64_t cols, int32_t elem_bytes) {
  WBX_REQUIRE(batch >= 0 && rows >= 0 && cols >= 0, "negative extent");
  if (batch == 0 || rows == 0 || cols == 0) return 0;
  WBX_REQUIRE(dst != nullptr && src != nullptr, "NULL pointer");
  WBX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "elem_bytes %d: 4 or 8", elem_bytes);
  const int64_t plane = rows * cols;
  const char* s = static_cast<const char*>(src);
  char* d = static_cast<char*>(dst);
  const int64_t bytes = plane * elem_bytes;
  WBX_REQUIRE(d + batch * bytes <= s || s + batch * bytes <= d, "source and destination overlap");
  static const bool avx2 = __builtin_cpu_supports("avx2");
  for (int64_t b = 0; b < batch; ++b, s += bytes, d += bytes) {
    if (elem_bytes == 4) {
      if (avx2)
        plane_f32_avx2(reinterpret_cast<float*>(d), reinterpret_cast<const float*>(s), rows, cols);
      else
        plane_scalar(reinterpret_cast<uint32_t*>(d), reinterpret_cast<const uint32_t*>(s), rows, cols);
    } else {
      plane_scalar(reinterpret_cast<uint64_t*>(d), reinterpret_cast<const uint64_t*>(s), rows, cols);
    }
  }
  return 0;
}
