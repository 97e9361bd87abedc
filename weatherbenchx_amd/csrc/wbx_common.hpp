// Shared host-side plumbing of libwbx_hip.so: context object, error reporting, HIP checks.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/wbx.h"

struct wbx_ctx {
  int device = 0;
  int num_cus = 256;  // hipDeviceProp_t::multiProcessorCount
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev_start = nullptr;
  hipEvent_t ev_stop = nullptr;
  // scratch for rocFFT (spectrum) and plans live in wbx_spectrum.hip
  void* fft_state = nullptr;
  void* s2_scratch = nullptr;  // split partials of wbx_contract_bits
  size_t s2_scratch_size = 0;
  void* aidm_scratch = nullptr;  // wbx_det_binned: atom ids with the validity mask folded in (one byte per point)
  size_t aidm_scratch_size = 0;
  void* atoms_clean = nullptr;  // std::set<const void*>*: prepared atom tables (wbx_binned_atoms) without an overflowing patch
  void* patch_counters = nullptr;  // wbx_ens_binned: arrival counters of its in-kernel sums over patches (uint32, kept zero)
  size_t patch_counters_size = 0;
  uint32_t ens_queue_parity = 0;   // which of the two ticket sets at the head of patch_counters the next persistent launch uses
  hipEvent_t* marks = nullptr;  // wbx_mark: timing events, created on demand and recycled by wbx_marks_reset
  int marks_used = 0, marks_made = 0, marks_cap = 0;
};

namespace wbx {

char* last_error_buf();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// Streaming load: p, t, c and the ensemble members are read exactly once, so they are fetched with the non-temporal
// hint (global_load ... nt) and do not displace the re-used lines (weights, membership bits, masks, offset tables) in
// L2 / Infinity Cache.  Measured on the configs[1] kernel: 6.12 -> 6.45 TB/s (76.5 % -> 80.6 % of the HBM peak).
template <typename T>
__device__ __forceinline__ T ld_stream(const T* p) {
#ifdef WBX_LD_STREAM_PLAIN  // A/B build (make ab-plainld): the same loads without the hint
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}

// A wave-uniform pointer the compiler also KNOWS to be uniform (SGPR pair): row offsets come out of tables through vector
// loads once an asm statement clobbers memory.
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const char*)(((uint64_t)hi << 32) | lo);
}

// a pointer that went through an opaque asm statement or an integer has lost its address space: say "global" again, or the
// loads come out as flat_load (which also counts on lgkmcnt)
template <typename T>
using global_ptr = const __attribute__((address_space(1))) T*;

// read-only tables (the plan's offset tables, the row weights) addressed with wave-uniform indices: through the constant
// address space these are scalar loads (s_load, SGPR results, counted on lgkmcnt) -- plain global loads are vector loads even
// when every lane asks for the same element
template <typename T>
using const_ptr = const __attribute__((address_space(4))) T*;

// what a NULL table stands for (all zeros / all ones): selecting one of these instead of branching around a load keeps the
// row lookups of the sweep free of control flow (the compiler waits for outstanding scalar loads wherever two paths join)
static __constant__ int64_t wbx_zero_i64[1] = {0};
static __constant__ double wbx_one_f64[1] = {1.0};

// Broadcast of lane j's 64-bit value to the whole wave (result in SGPRs).
__device__ __forceinline__ int64_t readlane64(int64_t v, int j) {
  const int lo = __builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, j);
  const int hi = __builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), j);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// Sum over the 64 lanes of a wave (every lane gets the total; all lanes must be active).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over the 64 lanes with DPP row operations (no LDS crossbar round trips, unlike the ds_bpermute shuffles above):
// the total is valid in LANE 63 ONLY.  All lanes must be active.  Same step pattern as wave_or32 (wbx_patch.hpp).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_moved(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);  // rows masked off receive +0.0
}

__device__ __forceinline__ double wave_sum_lane63(double v) {
  v += dpp_moved<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_moved<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_moved<0x124, 0xf>(v);  // row_ror:4
  v += dpp_moved<0x128, 0xf>(v);  // row_ror:8  -> every lane holds its row's sum
  v += dpp_moved<0x142, 0xa>(v);  // row_bcast:15 into rows 1, 3
  v += dpp_moved<0x143, 0xc>(v);  // row_bcast:31 into rows 2, 3
  return v;
}

// The wave total as a wave-uniform value (SGPRs).
__device__ __forceinline__ double wave_sum_uniform(double v) {
  const double t = wave_sum_lane63(v);
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(t), 63), __builtin_amdgcn_readlane(__double2loint(t), 63));
}

}  // namespace wbx

#define WBX_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess)                                                               \
      return wbx::fail(WBX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                             \
  } while (0)

#define WBX_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return wbx::fail(WBX_ERR_INVALID, __VA_ARGS__); \
  } while (0)
