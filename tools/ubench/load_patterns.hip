// Micro-benchmark: cost of a wave64 global load on one CU for the access patterns a latitude-fastest spectrum loader can
// use.  The field is [lon][721] floats; a block wants 24 adjacent latitudes of every longitude (96-byte segments 2884 bytes
// apart, arbitrary 4-byte alignment).  12 waves per block, 1 block per CU, data small enough to stay in L2.
//   0  dwordx2, 12 lanes per longitude (unaligned 8-byte pieces)       -- the first version of zspec1440_latfast_kernel
//   1  dword,   24 lanes per longitude
//   2  dwordx4, 7 lanes per longitude on a 16-byte-aligned 112-byte window
//   3  dwordx4, fully contiguous aligned rows (the longitude-fastest kernel's kind of load), for reference
//   4  dwordx2, fully contiguous aligned
//   5  dword,   fully contiguous
// build: hipcc --offload-arch=gfx950 -O3 load_patterns.hip -o load_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float v4a __attribute__((ext_vector_type(4)));
typedef float v2a __attribute__((ext_vector_type(2)));

template <int P>
__global__ void __launch_bounds__(768) k(const float* __restrict__ f, float* out, int iters, int nlat) {
  const int tid = threadIdx.x;
  const float* base = f + (size_t)blockIdx.x * 24 % (nlat - 40);  // the block's first latitude
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float* b = base + (size_t)(it & 3) * 360 * nlat;  // move around inside a 4 MB window (L2)
#pragma unroll 4
    for (int n = 0; n < 8; ++n) {
      if (P == 0) {
        const int t = tid % 12, j = tid / 12 + 64 * n;
        const v2u v = *reinterpret_cast<const v2u*>(b + (size_t)j * nlat + 2 * t);
        acc += v.x + v.y;
      } else if (P == 1) {
        const int t = tid % 24, j = tid / 24 + 32 * n;
        acc += b[(size_t)j * nlat + t];
      } else if (P == 2) {
        const int p = tid % 7, j = tid / 7 + 109 * n;  // 763 of 768 threads
        const float* s = b + (size_t)j * nlat;
        const v4a* w = reinterpret_cast<const v4a*>(reinterpret_cast<uintptr_t>(s) & ~(uintptr_t)15) + p;
        const v4a v = *w;
        acc += v.x + v.y + v.z + v.w;
      } else if (P == 3) {
        const v4a v = *reinterpret_cast<const v4a*>(b + ((size_t)n * 768 + tid) * 4);
        acc += v.x + v.y + v.z + v.w;
      } else if (P == 4) {
        const v2a v = *reinterpret_cast<const v2a*>(b + ((size_t)n * 768 + tid) * 2);
        acc += v.x + v.y;
      } else {
        acc += b[(size_t)n * 768 + tid];
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int P>
int run(const char* name, const float* f, float* out, int bytes_per_lane) {
  const int iters = 2000;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<P>, dim3(256), dim3(768), 0, 0, f, out, 10, 721);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<P>, dim3(256), dim3(768), 0, 0, f, out, iters, 721);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double wave_loads_per_cu = 12.0 * iters * 8;
  printf("%-58s %8.3f ms -> %7.1f ns per wave load per CU, %6.1f useful B/ns per CU\n", name, ms, ms * 1e6 / wave_loads_per_cu,
         wave_loads_per_cu * 64 * bytes_per_lane / (ms * 1e6));
  return 0;
}

int main() {
  float *f, *out;
  CHECK(hipMalloc(&f, (size_t)3200 * 721 * 4 + 4096)); CHECK(hipMemset(f, 0, (size_t)3200 * 721 * 4 + 4096)); CHECK(hipMalloc(&out, 4));
  run<0>("0 dwordx2, 12 lanes / longitude, unaligned", f, out, 8);
  run<1>("1 dword, 24 lanes / longitude", f, out, 4);
  run<2>("2 dwordx4, 7 lanes / longitude, 16-byte aligned window", f, out, 16);
  run<3>("3 dwordx4 contiguous", f, out, 16);
  run<4>("4 dwordx2 contiguous", f, out, 8);
  run<5>("5 dword contiguous", f, out, 4);
  return 0;
}
