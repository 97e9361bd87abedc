// Micro-benchmark: LDS cost (cycles per wave64 instruction per CU) of ds_read / ds_write at 4 / 8 / 16 bytes per lane,
// consecutive lanes on consecutive elements (the conflict-free pattern of the spectrum kernels' transposes), 16 waves
// per CU each on a private 8 KB region.  build: hipcc --offload-arch=gfx950 -O3 lds_rates.hip -o lds_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2 __attribute__((ext_vector_type(2)));
typedef float v4 __attribute__((ext_vector_type(4)));

// MODE 0 write-only, 1 read-only, 2 write then read (a transpose round trip); T = float / v2 / v4; STRIDE in elements
template <typename T, int MODE, int STRIDE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* buf = reinterpret_cast<T*>(raw + wave * 8192);
  T v[8];
  for (int i = 0; i < 8; ++i) { T t; __builtin_memset(&t, 0, sizeof(T)); v[i] = t; }
  float acc = lane;
  for (int i = 0; i < 8; ++i) buf[lane * STRIDE % 64 + i * 64] = v[i];
  for (int it = 0; it < iters; ++it) {
    if (MODE != 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) buf[(lane * STRIDE) % 64 + i * 64] = v[i];
      __builtin_amdgcn_wave_barrier();
    }
    if (MODE != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = buf[(lane * STRIDE) % 64 + i * 64];
      __builtin_amdgcn_wave_barrier();
      asm volatile("" ::: "memory");
    }
    if (MODE == 0) asm volatile("" ::: "memory");
  }
  for (int i = 0; i < 8; ++i) acc += reinterpret_cast<float*>(&v[i])[0];
  if (acc == 12345.678f) out[0] = acc;
}

template <typename T, int MODE, int STRIDE>
int run(const char* name, float* out) {
  const int iters = 2048, blocks = 256, threads = 1024;  // one block of 16 waves per CU
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<T, MODE, STRIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 8192));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<T, MODE, STRIDE>), dim3(blocks), dim3(threads), 16 * 8192, 0, out, 16);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<T, MODE, STRIDE>), dim3(blocks), dim3(threads), 16 * 8192, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_cu = 16.0 * iters * 8 * (MODE == 2 ? 2 : 1);
  const double ns = ms * 1e6 / instr_per_cu;
  printf("%-34s %8.3f ms -> %6.2f ns per wave-instruction per CU (%5.1f B/ns per CU)\n", name, ms, ns, 64.0 * sizeof(T) / ns);
  return 0;
}

int main() {
  float* out; CHECK(hipMalloc(&out, 4));
  run<float, 0, 1>("ds_write_b32", out); run<v2, 0, 1>("ds_write_b64", out); run<v4, 0, 1>("ds_write_b128", out);
  run<float, 1, 1>("ds_read_b32", out); run<v2, 1, 1>("ds_read_b64", out); run<v4, 1, 1>("ds_read_b128", out);
  run<v2, 2, 1>("write+read b64", out); run<v4, 2, 1>("write+read b128", out);
  return 0;
}
