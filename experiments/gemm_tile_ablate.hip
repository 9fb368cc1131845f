// gemm_tile_ablate.hip — where does gemm_tile_kernel (the product GEMM) lose time?  LM shapes of stt-1b, bf16 weights,
// M = 64 and 32, cold weights (12 rotating buffers).  Build once per mask: hipcc -DDSM_TILE_ABL=<mask>
//   bit 0 no X global loads, 1 all W loads hit one row (cache), 2 no LDS write/read, 3 no barrier, 4 no MFMA, 5 no stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dsm_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int MT, int NT, int EPI>
float run(GemmArgs a, int iters, const uint16_t* W0, size_t wstride, int nbuf) {
  const int chunks = a.Kpad / DSM_KC;
  dim3 grid((a.N + 63) / 64, chunks, (a.M + 16 * MT - 1) / (16 * MT));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_tile_kernel<uint16_t, uint16_t, MT, NT, EPI>), grid, dim3(256), 0, 0, a);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) {
    a.W = W0 + (size_t)(i % nbuf) * wstride;
    hipLaunchKernelGGL((gemm_tile_kernel<uint16_t, uint16_t, MT, NT, EPI>), grid, dim3(256), 0, 0, a);
  }
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0f / iters;
}

int main() {
  const int NBUF = 12;
  struct Shape { const char* name; int N, K, NT, stride; } shapes[] = {
      {"qkv", 6144, 2048, 1, 16}, {"gate", 5632, 2048, 2, 5632}, {"out_proj", 2048, 2048, 1, 16}, {"ff_out", 2048, 5632, 1, 16}};
  const size_t wmax = (size_t)(11264 + 128) * 2048;
  float *X, *ws; uint16_t* W;
  const int MMAX = 512;
  CK(hipMalloc(&X, (size_t)MMAX * 5632 * 4)); CK(hipMalloc(&W, wmax * 2 * NBUF)); CK(hipMalloc(&ws, (size_t)22 * MMAX * 11264 * 4 + (1 << 20)));
  std::vector<float> hx((size_t)MMAX * 5632); for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.0f;
  std::vector<uint16_t> hw(wmax); for (auto& v : hw) v = dsm_f32_to_bf16((rand() % 2001 - 1000) / 1000.0f);
  CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < NBUF; ++i) CK(hipMemcpy(W + (size_t)i * wmax, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  printf("DSM_TILE_ABL=%d\n", DSM_TILE_ABL);
  for (int M : {64, 32, 512}) {
    for (auto& s : shapes) {
      GemmArgs a; memset(&a, 0, sizeof a);
      a.X = X; a.xmap.bstride = 0; a.xmap.rpb = M; a.xmap.ld = s.K; a.xmap.toff = 0;
      a.W = W; a.Kpad = s.K; a.K = s.K; a.N = s.N; a.M = M; a.nt_stride = s.stride; a.ws = ws;
      a.ws_ntiles = (((s.NT - 1) * s.stride) >> 4) + ((s.N + 63) / 64) * 4;
      const double mfma = (double)(s.N * s.NT / 16) * (M / 16) * (s.K / 4);
      float t;
      if (M >= 64) t = s.NT == 2 ? run<4, 2, EPI_GATE>(a, 40, W, wmax, NBUF) : run<4, 1, EPI_STORE>(a, 40, W, wmax, NBUF);
      else t = s.NT == 2 ? run<2, 2, EPI_GATE>(a, 40, W, wmax, NBUF) : run<2, 1, EPI_STORE>(a, 40, W, wmax, NBUF);
      printf("  M=%2d %-9s ideal MFMA %5.1f us | %6.1f us\n", M, s.name, mfma * 32 / 1024 / 2400.0, t);
    }
  }
  return 0;
}
