// Micro-benchmark: issue cost of single VALU instructions, written as inline assembly so that the compiler adds nothing
// (fminf() may come with canonicalising v_max x, x).  8 independent chains per lane; 1, 2, 4 and 8 waves per SIMD.
// Question it answers: which 32-bit instructions run at the v_add_f32 rate and which at the fp64 / packed rate -- the
// ensemble kernel's sorting network is 830 v_min_f32 / v_max_f32 per 64 points.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define OP2(name, text)                                                                                       \
  struct name {                                                                                               \
    static constexpr const char* label = text;                                                                \
    static __device__ __forceinline__ void op(float& a, float b) { asm volatile(text " %0, %0, %1" : "+v"(a) : "v"(b)); } \
  };
#define OP3(name, text)                                                                                       \
  struct name {                                                                                               \
    static constexpr const char* label = text;                                                                \
    static __device__ __forceinline__ void op(float& a, float b) { asm volatile(text " %0, %0, %1, %1" : "+v"(a) : "v"(b)); } \
  };

OP2(AddF32, "v_add_f32")
OP2(MulF32, "v_mul_f32")
OP2(MinF32, "v_min_f32")
OP2(MaxF32, "v_max_f32")
OP2(MinI32, "v_min_i32")
OP2(MaxI32, "v_max_i32")
OP2(MinU32, "v_min_u32")
OP2(AndB32, "v_and_b32")
OP2(XorB32, "v_xor_b32")
OP2(AddU32, "v_add_u32")
OP2(SubF32, "v_sub_f32")
OP3(Min3F32, "v_min3_f32")
OP3(Med3F32, "v_med3_f32")
OP3(Max3F32, "v_max3_f32")
OP3(Min3I32, "v_min3_i32")
OP3(Med3I32, "v_med3_i32")
OP3(FmaF32, "v_fma_f32")
OP3(Add3U32, "v_add3_u32")

// a comparator as the kernel has it: lo = min(a, b), hi = max(a, b) on two chains
struct Comparator {
  static constexpr const char* label = "v_min_f32 + v_max_f32 (comparator)";
};
struct ComparatorI {
  static constexpr const char* label = "v_min_i32 + v_max_i32 (comparator)";
};

template <typename O>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  const float b = seed * 1.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) O::op(a[i], b);
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678f) out[0] = s;
}

template <bool INT>
__global__ void __launch_bounds__(256) kcmp(float* out, int iters, float seed) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + (i * 5 % 8);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {  // two comparators back to back, so that the values return to their registers
      float lo, hi;
      if (INT) {
        asm volatile("v_min_i32 %0, %2, %3\n\tv_max_i32 %1, %2, %3\n\tv_min_i32 %2, %0, %1\n\tv_max_i32 %3, %0, %1"
                     : "=&v"(lo), "=&v"(hi), "+v"(a[i]), "+v"(a[i + 1]));
      } else {
        asm volatile("v_min_f32 %0, %2, %3\n\tv_max_f32 %1, %2, %3\n\tv_min_f32 %2, %0, %1\n\tv_max_f32 %3, %0, %1"
                     : "=&v"(lo), "=&v"(hi), "+v"(a[i]), "+v"(a[i + 1]));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678f) out[0] = s;
}

template <typename F>
int time_it(const char* label, int ops_per_iter, F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%-36s", label);
  for (int waves : {1, 2, 4, 8}) {  // waves per SIMD: blocks of 4 waves, `waves` blocks per CU
    const int iters = 4096, blocks = 256 * waves;
    launch(blocks, 16);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(blocks, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = (double)blocks * 4 * iters * ops_per_iter / 1024.0;
    printf("  %d w/SIMD %5.2f ns", waves, ms * 1e6 / per_simd);
  }
  printf("   per wave-instruction per SIMD\n");
  return 0;
}

template <typename O>
int run(float* out) {
  return time_it(O::label, 8, [&](int blocks, int iters) { hipLaunchKernelGGL(k<O>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); });
}

int main() {
  float* out;
  CHECK(hipMalloc(&out, 4));
  run<AddF32>(out); run<SubF32>(out); run<MulF32>(out); run<FmaF32>(out);
  run<MinF32>(out); run<MaxF32>(out); run<Min3F32>(out); run<Med3F32>(out); run<Max3F32>(out);
  run<MinI32>(out); run<MaxI32>(out); run<MinU32>(out); run<Min3I32>(out); run<Med3I32>(out);
  run<AndB32>(out); run<XorB32>(out); run<AddU32>(out); run<Add3U32>(out);
  time_it(Comparator::label, 16, [&](int blocks, int iters) { hipLaunchKernelGGL(kcmp<false>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); });
  time_it(ComparatorI::label, 16, [&](int blocks, int iters) { hipLaunchKernelGGL(kcmp<true>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f); });
  return 0;
}
