// gemm_loop_ablate.hip — where does the large-batch GEMM (gemm_loop_kernel's structure: 64n x 64m tile, whole K, rolling
// four-block load window, activations through LDS with one barrier per block) lose its MFMA cycles?  A local copy of the
// loop with parts compiled out by ABL bits (timing only, results are wrong by construction):
//   1 no global loads (operands stay whatever was loaded first)   2 no LDS traffic (fragments from registers)
//   4 no barrier   8 no MFMA (one FMA per block keeps the dependency)   16 accumulate into ONE chunk (no tot += acc)
// Shapes: stt-1b QKV / out_proj / ff_out at M = 512 (bf16 weights, f32 activations).  Build: see experiments/README.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "dsm_kernels.h"
#define DSM_XS_LD_R01 36  // the padded LDS rows these ablations were measured with (the product kernels swizzle since r02)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

template <int ABL, int OCC>
__global__ __launch_bounds__(256, OCC) void loop_abl(const float* __restrict__ X, const uint16_t* __restrict__ W, float* __restrict__ Y,
                                                      int M, int N, int K) {
  constexpr int MT = 4, D = 4;
  __shared__ __attribute__((aligned(16))) float Xs[2][16 * MT][DSM_XS_LD_R01];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int m_base = blockIdx.z * 64, n_base = blockIdx.x * 64 + 16 * wave;
  const uint16_t* wrow = W + (long)(n_base + r) * K + 8 * q;
  const int row0 = tid >> 3, part = tid & 7;
  const float* xsrc0 = X + (long)(m_base + row0) * K + 4 * part;
  const float* xsrc1 = X + (long)(m_base + row0 + 32) * K + 4 * part;
  const int xdst0 = row0 * DSM_XS_LD_R01 + 4 * part, xdst1 = (row0 + 32) * DSM_XS_LD_R01 + 4 * part;
  const int nb = K >> 5;
  f32x4 acc[MT], tot[MT];
  for (int mt = 0; mt < MT; ++mt) { acc[mt] = (f32x4){0, 0, 0, 0}; tot[mt] = (f32x4){0, 0, 0, 0}; }
  float4 xp0, xp1, xp2, xp3, xq0, xq1, xq2, xq3;
  Raw8<uint16_t> rw0, rw1, rw2, rw3;
#define LLOAD(S, G) { const int kb_ = 32 * min((G), nb - 1); xp##S = *reinterpret_cast<const float4*>(xsrc0 + kb_); xq##S = *reinterpret_cast<const float4*>(xsrc1 + kb_); rw##S.load(wrow + kb_); }
  LLOAD(0, 0) LLOAD(1, 1) LLOAD(2, 2) LLOAD(3, 3)
  __builtin_amdgcn_sched_barrier(0);
  float xbA[MT][8], xbB[MT][8];
#define LSTORE(S, BUF) if (!(ABL & 2)) { float* xs_ = &Xs[BUF][0][0]; *reinterpret_cast<float4*>(xs_ + xdst0) = xp##S; *reinterpret_cast<float4*>(xs_ + xdst1) = xq##S; }
#define LFRAG(XB, BUF, S) for (int mt = 0; mt < MT; ++mt) { \
    if (ABL & 2) { for (int j = 0; j < 8; ++j) XB[mt][j] = xp##S.x + (float)(mt + j); } else { \
    const float* fp = &Xs[BUF][0][0] + (16 * mt + r) * DSM_XS_LD_R01 + 8 * q; \
    const float4 f0 = *reinterpret_cast<const float4*>(fp), f1 = *reinterpret_cast<const float4*>(fp + 4); \
    XB[mt][0] = f0.x; XB[mt][1] = f0.y; XB[mt][2] = f0.z; XB[mt][3] = f0.w; XB[mt][4] = f1.x; XB[mt][5] = f1.y; XB[mt][6] = f1.z; XB[mt][7] = f1.w; } }
#define LMFMA(CUR, S0, S1) for (int s = (S0); s < (S1); ++s) { \
    for (int mt = 0; mt < MT; ++mt) { if (ABL & 8) { if (s == 0) acc[mt][0] += wa[s] * CUR[mt][s]; } else acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s], CUR[mt][s], acc[mt], 0, 0, 0); } \
    __builtin_amdgcn_sched_barrier(0); }
  LSTORE(0, 0)
  if (!(ABL & 4)) __syncthreads();
  LFRAG(xbA, 0, 0)
#define LSTEP(S, SN, CUR, NXT) { const int gb = g + (S); float wa[8]; rw##S.unpack(wa); \
    if (gb < nb) { if (gb + 1 < nb) LSTORE(SN, ((S) + 1) & 1) if (!(ABL & 4)) __syncthreads(); LMFMA(CUR, 0, 1) if (gb + 1 < nb) { LFRAG(NXT, ((S) + 1) & 1, SN) } __builtin_amdgcn_sched_barrier(0); } \
    if (!(ABL & 1)) LLOAD(S, gb + D) __builtin_amdgcn_sched_barrier(0); \
    if (gb < nb) { LMFMA(CUR, 1, 8) if (!(ABL & 16) && ((gb & 7) == 7 || gb == nb - 1)) { for (int mt = 0; mt < MT; ++mt) { tot[mt] = tot[mt] + acc[mt]; acc[mt] = (f32x4){0, 0, 0, 0}; } } } }
#pragma clang loop unroll(disable)
  for (int g = 0; g < nb; g += D) { LSTEP(0, 1, xbA, xbB) LSTEP(1, 2, xbB, xbA) LSTEP(2, 3, xbA, xbB) LSTEP(3, 0, xbB, xbA) }
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 v = (ABL & 16) ? acc[mt] : tot[mt];
    *reinterpret_cast<f32x4*>(Y + (long)(m_base + 16 * mt + r) * N + n_base + 4 * q) = v;
  }
}

template <int ABL, int OCC>
float run(const float* X, const uint16_t* W, float* Y, int M, int N, int K, size_t wstride, int nbuf) {
  dim3 grid(N / 64, 1, M / 64);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((loop_abl<ABL, OCC>), grid, dim3(256), 0, 0, X, W, Y, M, N, K);
  CK(hipEventRecord(e0));
  const int iters = 20;
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((loop_abl<ABL, OCC>), grid, dim3(256), 0, 0, X, W + (size_t)(i % nbuf) * wstride, Y, M, N, K);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1000.0f / iters;
}

int main() {
  const int NBUF = 6, M = 512;
  struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 6144, 2048}, {"out_proj", 2048, 2048}, {"ff_out", 2048, 5632}, {"gate_as_nt1", 11264, 2048}};
  const size_t wmax = (size_t)11264 * 2048;
  float *X, *Y; uint16_t* W;
  CK(hipMalloc(&X, (size_t)M * 5632 * 4)); CK(hipMalloc(&Y, (size_t)M * 11264 * 4)); CK(hipMalloc(&W, wmax * 2 * NBUF));
  std::vector<float> hx((size_t)M * 5632); for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.0f;
  std::vector<uint16_t> hw(wmax); for (auto& v : hw) v = dsm_f32_to_bf16((rand() % 2001 - 1000) / 1000.0f);
  CK(hipMemcpy(X, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < NBUF; ++i) CK(hipMemcpy(W + (size_t)i * wmax, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  for (auto& s : shapes) {
    const double ideal = (double)(s.N / 16) * (M / 16) * (s.K / 4) * 32 / 1024 / 2400.0;  // us at 2.4 GHz
    printf("%-12s M=%d N=%d K=%d ideal MFMA %.1f us @2.4GHz | full(occ2) %.1f | full(occ3) %.1f | no-global %.1f | no-LDS %.1f | no-barrier %.1f | no-LDS+no-barrier %.1f | MFMA only %.1f | no-MFMA %.1f | one-chunk %.1f\n",
           s.name, M, s.N, s.K, ideal,
           run<0, 2>(X, W, Y, M, s.N, s.K, wmax, NBUF), run<0, 3>(X, W, Y, M, s.N, s.K, wmax, NBUF), run<1, 3>(X, W, Y, M, s.N, s.K, wmax, NBUF),
           run<2, 3>(X, W, Y, M, s.N, s.K, wmax, NBUF), run<4, 3>(X, W, Y, M, s.N, s.K, wmax, NBUF), run<6, 3>(X, W, Y, M, s.N, s.K, wmax, NBUF),
           run<7, 3>(X, W, Y, M, s.N, s.K, wmax, NBUF), run<8, 3>(X, W, Y, M, s.N, s.K, wmax, NBUF), run<16, 3>(X, W, Y, M, s.N, s.K, wmax, NBUF));
  }
  return 0;
}
