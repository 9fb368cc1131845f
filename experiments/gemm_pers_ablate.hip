// gemm_pers_ablate.hip — where does gemm_pers_kernel lose time?  LM shapes of stt-1b at M = 64, bf16 weights.
// Build once per ablation mask:  hipcc -DDSM_GEMM_ABL=<mask> ...   bit 0: no X global loads, 1: no W refills,
// 2: no slab stores, 3: no MFMAs.  Timing only — ablated variants compute garbage.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_pers_kernel.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int NT, int EPI>
float run(GemmArgs a, int wgs, int iters, const uint16_t* W0, size_t wstride, int nbuf) {
  const int chunks = a.Kpad / DSM_KC, mtiles = (a.M + 15) / 16;
  PersGeom gm;
  gm.chunks = chunks; gm.mper = mtiles < 4 ? mtiles : 4; gm.mgroups = (mtiles + gm.mper - 1) / gm.mper;
  gm.ntiles64 = (a.N + 63) / 64; gm.atoms = chunks * gm.mgroups * gm.ntiles64 * gm.mper;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_pers_kernel<uint16_t, uint16_t, NT, EPI>), dim3(wgs), dim3(256), 0, 0, a, gm);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) {
    a.W = W0 + (size_t)(i % nbuf) * wstride;
    hipLaunchKernelGGL((gemm_pers_kernel<uint16_t, uint16_t, NT, EPI>), dim3(wgs), dim3(256), 0, 0, a, gm);
  }
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0f / iters;
}

int main() {
  const int M = 64, NBUF = 12;
  struct Shape { const char* name; int N, K, NT, stride; } shapes[] = {
      {"qkv", 6144, 2048, 1, 16}, {"gate", 5632, 2048, 2, 5632}, {"out_proj", 2048, 2048, 1, 16}, {"ff_out", 2048, 5632, 1, 16}};
  const size_t wmax = (size_t)(11264 + 128) * 2048;
  float *X, *ws; uint16_t* W;
  CK(hipMalloc(&X, (size_t)M * 5632 * 4)); CK(hipMalloc(&W, wmax * 2 * NBUF)); CK(hipMalloc(&ws, (size_t)22 * M * 11264 * 4 + (1 << 20)));
  std::vector<float> hx((size_t)M * 5632); for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.0f;
  std::vector<uint16_t> hw(wmax); for (auto& v : hw) v = dsm_f32_to_bf16((rand() % 2001 - 1000) / 1000.0f);
  CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < NBUF; ++i) CK(hipMemcpy(W + (size_t)i * wmax, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  printf("DSM_GEMM_ABL=%d\n", DSM_GEMM_ABL);
  for (auto& s : shapes) {
    GemmArgs a; memset(&a, 0, sizeof a);
    a.X = X; a.xmap.bstride = 0; a.xmap.rpb = M; a.xmap.ld = s.K; a.xmap.toff = 0;
    a.W = W; a.Kpad = s.K; a.K = s.K; a.N = s.N; a.M = M; a.nt_stride = s.stride; a.ws = ws;
    const int gx = (s.N + 63) / 64;
    a.ws_ntiles = (((s.NT - 1) * s.stride) >> 4) + gx * 4;
    const double mfma = (double)(s.N * s.NT / 16) * (M / 16) * (s.K / 4);
    printf("%-9s ideal %5.1f us |", s.name, mfma * 32 / 1024 / 2400.0);
    for (int wgs : {256, 512, 768, 1024}) {
      float t = s.NT == 2 ? run<2, EPI_GATE>(a, wgs, 40, W, wmax, NBUF) : run<1, EPI_STORE>(a, wgs, 40, W, wmax, NBUF);
      printf("  wgs=%4d %6.1f us", wgs, t);
    }
    printf("\n");
  }
  return 0;
}
