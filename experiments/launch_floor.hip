// launch_floor.hip — what does a short MFMA kernel cost beyond its MFMAs?  Back-to-back launches on one stream of
// (a) an empty kernel, (b) 256 MFMAs per wave in a rolled loop, (c) the same 256 MFMAs fully unrolled (2 KB of
// straight-line code), (d) unrolled + 6 KB of never-taken code in front, for grids of 256..1024 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_empty(float* out, float a0) {
  if (a0 == 12345.f) out[threadIdx.x] = a0;
}
template <bool UNROLL>
__global__ __launch_bounds__(256) void k_mfma(float* out, float a0, int n) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = a0 * 0.5f + threadIdx.x * 1e-3f;
  if (UNROLL) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + u, b, acc[i], 0, 0, 0);
    }
  } else {
#pragma unroll 1
    for (int u = 0; u < n; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + u, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <typename F>
float timeit(F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipEventRecord(e0));
  for (int i = 0; i < 100; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 10.0f;  // us per launch
}

int main() {
  float* d; CK(hipMalloc(&d, 1 << 20));
  printf("256 MFMAs per wave = 8192 cycles = 3.4 us at 1 wave/SIMD (wgs=256), x2 at 512, x3 at 768, x4 at 1024\n");
  for (int wgs : {256, 512, 768, 1024}) {
    float te = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, 0, d, 1.0f); });
    float tr = timeit([&] { hipLaunchKernelGGL(k_mfma<false>, dim3(wgs), dim3(256), 0, 0, d, 1.0f, 64); });
    float tu = timeit([&] { hipLaunchKernelGGL(k_mfma<true>, dim3(wgs), dim3(256), 0, 0, d, 1.0f, 64); });
    printf("wgs=%4d: empty %5.2f us | rolled %5.2f us | unrolled %5.2f us | ideal MFMA %5.2f us\n", wgs, te, tr, tu,
           3.413f * wgs / 256);
  }
  return 0;
}
