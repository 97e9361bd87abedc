// Which clock does s_memtime (clock64) tick at under a VALU-heavy load?  Compares it with s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int iters, float seed) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x + i;
  for (int it = 0; it < iters; ++it)
    for (int i = 0; i < 8; ++i) a[i] = fmaf(a[i], 1.0001f, 0.5f);
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  if (s == 1234.5f) out[2] = 1;
}
int main() {
  unsigned long long *d, h[3];
  hipMalloc(&d, 24);
  int wall_khz = 0, core_khz = 0;
  hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  hipDeviceGetAttribute(&core_khz, hipDeviceAttributeClockRate, 0);
  printf("hipDeviceAttributeWallClockRate = %d kHz, hipDeviceAttributeClockRate = %d kHz\n", wall_khz, core_khz);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {1, 2048}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1 << 16, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1 << 16, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("blocks=%d: %.3f ms by HIP events\n", blocks, ms);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("blocks=%d: s_memtime ticks %llu, s_memrealtime ticks %llu (100 MHz) -> s_memtime = %.1f MHz; v_fma_f32 = %.2f s_memtime ticks per wave instruction\n",
           blocks, h[0], h[1], 100.0 * (double)h[0] / (double)h[1], (double)h[0] / ((double)(1 << 16) * 8 * (blocks == 1 ? 1 : 2)));
  }
  return 0;
}
