// mfma_rate.hip — sustained issue rate of v_mfma_f32_16x16x4_f32 on gfx950: ACC independent accumulators per wave,
// WPS waves per SIMD, every CU busy.  Prints cycles per MFMA per SIMD assuming 2.4 GHz and the implied TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[ACC];
#pragma unroll
  for (int i = 0; i < ACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}

template <int ACC>
void run(float* d, int wgs, float av, float bv, const char* label) {
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<ACC>, dim3(wgs), dim3(256), 0, 0, d, 10, av, bv);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<ACC>, dim3(wgs), dim3(256), 0, 0, d, iters, av, bv);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_wave = (double)iters * 16 * ACC;
  const double waves_per_simd = wgs / 256.0;
  const double cyc = ms * 1e-3 * 2.4e9 / (mfma_per_wave * waves_per_simd);
  const double tf = mfma_per_wave * wgs * 4 * 2048 / (ms * 1e-3) / 1e12;
  printf("%-14s ACC=%d wgs=%4d: %8.3f ms  %.1f cyc/MFMA/SIMD @2.4GHz  %.1f TFLOP/s\n", label, ACC, wgs, ms, cyc, tf);
}

int main() {
  float* d; CK(hipMalloc(&d, 4096));
  for (int pass = 0; pass < 2; ++pass) {
    const float av = pass ? 1.2345f : 0.0f, bv = pass ? 0.9876f : 0.0f;
    const char* l = pass ? "nonzero data" : "zeros";
    run<1>(d, 256, av, bv, l); run<2>(d, 256, av, bv, l); run<4>(d, 256, av, bv, l); run<8>(d, 256, av, bv, l);
    run<1>(d, 512, av, bv, l); run<4>(d, 512, av, bv, l); run<4>(d, 1024, av, bv, l);
  }
  return 0;
}
