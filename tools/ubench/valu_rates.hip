// Micro-benchmark: VALU issue cost (cycles per wave64 instruction per SIMD) of the instructions the ensemble kernel
// is made of.  8 independent chains per lane, 16 waves per CU-SIMD group; build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  typedef float v2 __attribute__((ext_vector_type(2)));
  float a[8]; double d[8]; v2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; d[i] = a[i]; p[i] = (v2){a[i], a[i] + 0.5f}; }
  const v2 c1 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = a[i] + 1.0001f;                       // v_add_f32
      if (OP == 1) a[i] = fmaf(a[i], 1.0001f, 0.5f);            // v_fma_f32
      if (OP == 2) a[i] = fminf(a[i], a[(i + 1) & 7] + 1.f);    // v_add + v_min
      if (OP == 3) d[i] = d[i] + 1.0001;                        // v_add_f64
      if (OP == 4) d[i] = fma(d[i], 1.0001, 0.5);               // v_fma_f64
      if (OP == 5) d[i] = d[i] + (double)a[i];                  // v_cvt_f64_f32 + v_add_f64
      if (OP == 6) d[i] = d[i] * 1.0001;                        // v_mul_f64
      if (OP == 7) d[i] = d[i] + fabs(d[(i + 1) & 7]);          // v_add_f64 with |.| modifier
      if (OP == 8) p[i] = __builtin_elementwise_fma(p[i], c1, c2);  // v_pk_fma_f32
      if (OP == 9) p[i] = p[i] + c1;                            // v_pk_add_f32
      if (OP == 10) p[i] = p[i] * c1;                           // v_pk_mul_f32
      if (OP == 11) { a[i] = fmaf(a[i], 1.0001f, 0.5f); d[i] = fma(d[i], 1.0001, 0.5); }  // f32 + f64 interleaved
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <int OP>
int run(const char* name, int ops_per_iter_per_chain, float* out) {
  const int iters = 4096, blocks = 256 * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 16, 1.f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_instr = (double)blocks * 4 * iters * 8 * ops_per_iter_per_chain;  // per whole chip
  const double per_simd = wave_instr / 1024.0;
  printf("%-28s %8.3f ms  -> %6.2f ns per wave-instruction per SIMD (= %5.2f cycles at 2.4 GHz)\n", name, ms,
         ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  return 0;
}

int main() {
  float* out; CHECK(hipMalloc(&out, 4));
  run<0>("v_add_f32", 1, out); run<1>("v_fma_f32", 1, out); run<2>("v_add_f32+v_min_f32", 2, out);
  run<3>("v_add_f64", 1, out); run<4>("v_fma_f64", 1, out); run<5>("v_cvt_f64_f32+v_add_f64", 2, out);
  run<6>("v_mul_f64", 1, out); run<7>("v_add_f64 |src|", 1, out);
  run<8>("v_pk_fma_f32", 1, out); run<9>("v_pk_add_f32", 1, out); run<10>("v_pk_mul_f32", 1, out);
  run<11>("v_fma_f32+v_fma_f64", 2, out);
  return 0;
}
