// gemm_ablate.hip — where does the tiled f32-MFMA GEMM lose time?  QKV shape of stt-1b (M=64, N=6144, K=2048,
// bf16 weights).  ABL bit 0: skip X global loads; bit 1: skip W global loads; bit 2: skip LDS write+read (xb from
// registers); bit 3: skip barrier; bit 4: skip MFMAs.  Timing only — ablated variants compute garbage.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dsm_kernels.h"
#define DSM_XS_LD_R01 36  // the padded LDS rows these ablations were measured with (the product kernels swizzle since r02)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int MT, int ABL, int KCH>
__global__ __launch_bounds__(256, 2) void abl_kernel(const float* __restrict__ X, const uint16_t* __restrict__ W, float* __restrict__ ws,
                                                     int M, int Kpad, int ldx) {
  __shared__ __attribute__((aligned(16))) float Xs[2][16 * MT][DSM_XS_LD_R01];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int chunk = blockIdx.y;
  const int n_base = blockIdx.x * 64 + 16 * wave;
  const int k0 = chunk * KCH, k1 = k0 + KCH;
  const uint16_t* wrow = W + (long)(n_base + r) * Kpad + 8 * q;
  const int row0 = tid >> 3, part = tid & 7;
  const float* xsrc0 = X + (long)row0 * ldx + 4 * part;
  const float* xsrc1 = X + (long)(row0 + 32) * ldx + 4 * part;
  const int xdst0 = row0 * DSM_XS_LD_R01 + 4 * part, xdst1 = (row0 + 32) * DSM_XS_LD_R01 + 4 * part;
  f32x4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 xg0 = make_float4(1.f, 2.f, 3.f, 4.f), xg1 = xg0;
  float wa[8], wn[8];
  for (int j = 0; j < 8; ++j) wn[j] = 1.0f + j;
  uint4 wq[8];
  if (ABL & 32) {
#pragma unroll
    for (int i = 0; i < 8; ++i) wq[i] = *reinterpret_cast<const uint4*>(wrow + k0 + 32 * (i < KCH / 32 ? i : 0));
  }
  if (!(ABL & 1)) { xg0 = *reinterpret_cast<const float4*>(xsrc0 + k0); xg1 = *reinterpret_cast<const float4*>(xsrc1 + k0); }
  if (!(ABL & 2) && !(ABL & 32)) load_w8<uint16_t>(wrow + k0, wn);
  int buf = 0;
#pragma unroll
  for (int it = 0; it < KCH / 32; ++it) {
    const int kb = k0 + 32 * it;
    float* xs = &Xs[buf][0][0];
    if (!(ABL & 4)) { *reinterpret_cast<float4*>(xs + xdst0) = xg0; *reinterpret_cast<float4*>(xs + xdst1) = xg1; }
    if (ABL & 32) {
      const uint4 v = wq[it & 7];
      wa[0] = __uint_as_float(v.x << 16); wa[1] = __uint_as_float(v.x & 0xFFFF0000u);
      wa[2] = __uint_as_float(v.y << 16); wa[3] = __uint_as_float(v.y & 0xFFFF0000u);
      wa[4] = __uint_as_float(v.z << 16); wa[5] = __uint_as_float(v.z & 0xFFFF0000u);
      wa[6] = __uint_as_float(v.w << 16); wa[7] = __uint_as_float(v.w & 0xFFFF0000u);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) wa[j] = wn[j];
    }
    if (!(ABL & 8)) __syncthreads();
    const int kn = kb + 32;
    if (kn < k1) {
      if (!(ABL & 1)) { xg0 = *reinterpret_cast<const float4*>(xsrc0 + kn); xg1 = *reinterpret_cast<const float4*>(xsrc1 + kn); }
      if (!(ABL & 2) && !(ABL & 32)) load_w8<uint16_t>(wrow + kn, wn);
    }
    float xb[MT][8];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (!(ABL & 4)) {
        const float* fp = xs + (16 * mt + r) * DSM_XS_LD_R01 + 8 * q;
        float4 f0 = *reinterpret_cast<const float4*>(fp), f1 = *reinterpret_cast<const float4*>(fp + 4);
        xb[mt][0] = f0.x; xb[mt][1] = f0.y; xb[mt][2] = f0.z; xb[mt][3] = f0.w;
        xb[mt][4] = f1.x; xb[mt][5] = f1.y; xb[mt][6] = f1.z; xb[mt][7] = f1.w;
      } else {
        xb[mt][0] = xg0.x + mt; xb[mt][1] = xg0.y; xb[mt][2] = xg0.z; xb[mt][3] = xg0.w;
        xb[mt][4] = xg1.x; xb[mt][5] = xg1.y; xb[mt][6] = xg1.z; xb[mt][7] = xg1.w;
      }
    }
    if (!(ABL & 16)) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], xb[mt][s], acc[mt], 0, 0, 0);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][0] += wa[mt & 7] * xb[mt][mt & 7];
    }
    buf ^= 1;
  }
  const int mtiles = (M + 15) >> 4, ntiles = gridDim.x * 4;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    *reinterpret_cast<f32x4*>(ws + ((((long)chunk * mtiles + mt) * ntiles + (n_base >> 4)) * 64 + lane) * 4) = acc[mt];
}

static int g_nbuf = 1;
static size_t g_wstride = 0;
template <int ABL, int KCH>
float run(const float* X, const uint16_t* W0, float* ws, int M, int N, int K, int iters) {
  const uint16_t* W = W0;
  dim3 grid(N / 64, K / KCH, 1);
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((abl_kernel<4, ABL, KCH>), grid, dim3(256), 0, 0, X, W, ws, M, K, K);
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((abl_kernel<4, ABL, KCH>), grid, dim3(256), 0, 0, X, W0 + (size_t)(i % g_nbuf) * g_wstride, ws, M, K, K);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.0f / iters;
}

int main() {
  const int M = 64, N = 6144, K = 2048;
  float *X, *ws; uint16_t* W;
  const int NBUF = 16;
  g_wstride = (size_t)(N + 64) * K;
  CK(hipMalloc(&X, (size_t)M * K * 4)); CK(hipMalloc(&W, g_wstride * 2 * NBUF)); CK(hipMalloc(&ws, (size_t)16 * M * N * 4));
  std::vector<float> hx((size_t)M * K); for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.0f;
  std::vector<uint16_t> hw((size_t)(N + 64) * K); for (auto& v : hw) v = dsm_f32_to_bf16((rand() % 2001 - 1000) / 1000.0f);
  CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < NBUF; ++i) CK(hipMemcpy(W + (size_t)i * g_wstride, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  const double mfma = (double)(N / 16) * (M / 16) * (K / 4);
  printf("QKV-shaped GEMM M=%d N=%d K=%d: %.0f MFMAs; at 32 cyc/SIMD on 1024 SIMDs @2.4GHz = %.1f us\n", M, N, K, mfma, mfma * 32 / 1024 / 2400.0);
#define R(ABL, KCH, name) printf("  KC=%4d %-44s %7.1f us\n", KCH, name, run<ABL, KCH>(X, W, ws, M, N, K, 50));
  R(0, 256, "full (weights warm in cache)");
  R(32, 256, "full, W chunk preloaded up front (warm)");
  g_nbuf = NBUF;
  printf(" -- cold weights: %d rotating 25 MB weight buffers --\n", NBUF);
  R(0, 256, "full (cold)");
  R(32, 256, "full, W chunk preloaded up front (cold)");
  R(2, 256, "no W loads (cold)");
  R(0, 512, "full KC=512 (cold)");
  g_nbuf = 1;
  printf(" -- warm again --\n");
  R(1, 256, "no X global loads");
  R(2, 256, "no W global loads");
  R(3, 256, "no global loads at all");
  R(4, 256, "no LDS write/read");
  R(8, 256, "no barrier");
  R(12, 256, "no LDS, no barrier");
  R(15, 256, "MFMA only");
  R(16, 256, "everything but MFMA");
  R(0, 512, "full");
  R(15, 512, "MFMA only");
  R(0, 1024, "full");
  R(0, 2048, "full (no split-K: 96 blocks)");
  R(15, 2048, "MFMA only (96 blocks)");
  return 0;
}
