// bf16_group_probe.hip — which of the 32 products of v_mfma_f32_16x16x32_bf16 are added together, and in which order?
// For every (i, j, l): +X at k = i, -X at k = j, a tiny s at k = l (X = 1, s = 2^-40).  The result is s only if X and -X have
// cancelled before s meets them (s sits in a LATER addition step than both); it is 0 when s was aligned against X first.
// Prints, for every l, the set of (i, j) classes — enough to read off the grouping and the order of the groups.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ void mfma_tiles(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, const float* __restrict__ C,
                           float* __restrict__ D, int ntiles) {
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const int l = threadIdx.x & 63, r = l & 15, q = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)A[(size_t)tile * 512 + r * 32 + 8 * q + j]; b[j] = (short)B[(size_t)tile * 512 + (8 * q + j) * 16 + r]; }
  f32x4 acc;
  for (int i = 0; i < 4; ++i) acc[i] = C[(size_t)tile * 256 + (q * 4 + i) * 16 + r];
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(size_t)tile * 256 + (q * 4 + i) * 16 + r] = acc[i];
}
int main() {
  const uint16_t ONE = 0x3F80, NEG1 = 0xBF80, TINY = 0x3580 /* 2^-20 */;
  std::vector<int> I, J, L;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int l = 0; l < 32; ++l) if (i != j && l != i && l != j) { I.push_back(i); J.push_back(j); L.push_back(l); }
  const size_t n = I.size(), ntiles = (n + 15) / 16;
  std::vector<uint16_t> A(ntiles * 512, 0), B(ntiles * 512, 0);
  std::vector<float> C(ntiles * 256, 0.f), D(ntiles * 256);
  for (size_t s = 0; s < n; ++s) {
    const size_t t = s / 16; const int d = (int)(s % 16);
    A[t * 512 + d * 32 + I[s]] = ONE; B[t * 512 + I[s] * 16 + d] = ONE;
    A[t * 512 + d * 32 + J[s]] = NEG1; B[t * 512 + J[s] * 16 + d] = ONE;
    A[t * 512 + d * 32 + L[s]] = TINY; B[t * 512 + L[s] * 16 + d] = TINY;
  }
  uint16_t *dA, *dB; float *dC, *dD;
  CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mfma_tiles, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, 0, dA, dB, dC, dD, (int)ntiles);
  CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
  // survive[l][i][j] = result == s
  static char surv[32][32][32];
  memset(surv, '.', sizeof surv);
  for (size_t s = 0; s < n; ++s) { const float v = D[(s / 16) * 256 + (s % 16) * 17]; surv[L[s]][I[s]][J[s]] = v != 0.0f ? '1' : '0'; }
  // "later[l][i]" : s at l survives against the pair (i, j) for EVERY j in i's own candidate group?  print the compact relation
  // R[l][i] = '1' if s at l survives with the pair (i, i^1)   (partner next to it: certainly the same addition step)
  printf("rows l = position of the tiny term, columns i = position of +X (with -X at i ^ 1); 1 = the tiny term survived\n");
  for (int l = 0; l < 32; ++l) { printf("l=%2d ", l); for (int i = 0; i < 32; ++i) putchar(surv[l][i][i ^ 1]); putchar('\n'); }
  printf("pairs (i, j) far apart, tiny term at l = 31 / l = 0: rows i, columns j\n");
  for (int l : {31, 0, 15, 16}) { printf("l = %d\n", l); for (int i = 0; i < 32; ++i) { printf("i=%2d ", i); for (int j = 0; j < 32; ++j) putchar(surv[l][i][j]); putchar('\n'); } }
  return 0;
}
