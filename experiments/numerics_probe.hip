// numerics_probe.hip — hardware facts the numerics contract rests on (run once per round on gfx950):
//  (A) v_mfma_f32_16x16x4_f32 == k-ordered f32 fmaf chain (bit-for-bit)?
//  (B) raw dump of v_mfma_f32_16x16x32_bf16 inputs/outputs for offline characterisation
//  (C) device results of dsm_numerics.h functions + IEEE div/sqrt, dumped for a bitwise
//      compare against the gcc build of the same header (experiments/numerics_check.c).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I delayed-streams-modeling_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "dsm_numerics.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd32() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 16);
}
static float rnd_unit() { return (float)((rnd32() >> 8) * (1.0 / 16777216.0)) * 2.0f - 1.0f; }
// wide dynamic range so summation order matters
static float rnd_wide() {
  int e = (int)(rnd32() % 41) - 20;
  return ldexpf(rnd_unit(), e);
}

// (A) K = 4*NSTEP chained f32 MFMAs.  A is [16][K] row-major, B is [K][16] row-major.
template <int NSTEP>
__global__ void mfma_f32_chain(const float* __restrict__ A, const float* __restrict__ B,
                               const float* __restrict__ C, float* __restrict__ D) {
  int l = threadIdx.x;
  int r = l & 15, q = l >> 4;
  f32x4 acc;
  for (int i = 0; i < 4; ++i) acc[i] = C[(q * 4 + i) * 16 + r];
  const int K = 4 * NSTEP;
  for (int t = 0; t < NSTEP; ++t) {
    float a = A[r * K + 4 * t + q];
    float b = B[(4 * t + q) * 16 + r];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 4; ++i) D[(q * 4 + i) * 16 + r] = acc[i];
}

// (B) one bf16 MFMA 16x16x32.  A [16][32] bf16, B [32][16] bf16, C/D [16][16] f32.
__global__ void mfma_bf16_one(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                              const float* __restrict__ C, float* __restrict__ D) {
  int l = threadIdx.x;
  int r = l & 15, q = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)A[r * 32 + 8 * q + j];
    b[j] = (short)B[(8 * q + j) * 16 + r];
  }
  f32x4 acc;
  for (int i = 0; i < 4; ++i) acc[i] = C[(q * 4 + i) * 16 + r];
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(q * 4 + i) * 16 + r] = acc[i];
}

// (C) numerics functions on device
__global__ void numerics_kernel(const float* __restrict__ x, const float* __restrict__ y, int n,
                                float* o_exp, float* o_elu, float* o_silu, float* o_gelu,
                                float* o_sin, float* o_cos, float* o_div, float* o_rsqrt, float* o_fma) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = x[i], b = y[i];
  o_exp[i] = dsm_expf(a);
  o_elu[i] = dsm_elu(a);
  o_silu[i] = dsm_silu(a);
  o_gelu[i] = dsm_gelu_erf(a);
  float s, c;
  dsm_sincosf(fabsf(a) * 1000.0f, &s, &c);
  o_sin[i] = s;
  o_cos[i] = c;
  o_div[i] = a / b;
  o_rsqrt[i] = 1.0f / sqrtf(fabsf(b) + 1e-8f);
  o_fma[i] = a * b + a;  // must stay un-contracted: mul then add
}

static void dump(const char* path, const void* p, size_t bytes) {
  FILE* f = fopen(path, "wb");
  if (!f) { perror(path); exit(3); }
  fwrite(p, 1, bytes, f);
  fclose(f);
}

int main(int argc, char** argv) {
  const char* outdir = argc > 1 ? argv[1] : "gpurun_out";
  char path[512];
  // ---------------- (A) ----------------
  {
    const int NSTEP = 16, K = 64, TRIALS = 256;
    std::vector<float> A(16 * K), B(K * 16), C(256), D(256);
    float *dA, *dB, *dC, *dD;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4));
    CK(hipMalloc(&dC, 1024)); CK(hipMalloc(&dD, 1024));
    long mism_chain = 0, mism_rev = 0, total = 0;
    for (int t = 0; t < TRIALS; ++t) {
      for (auto& v : A) v = (t & 1) ? rnd_wide() : rnd_unit();
      for (auto& v : B) v = (t & 1) ? rnd_wide() : rnd_unit();
      for (auto& v : C) v = (t & 2) ? rnd_unit() : 0.0f;
      CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(mfma_f32_chain<NSTEP>, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
      CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          float acc = C[i * 16 + j], rev = C[i * 16 + j];
          for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 16 + j], acc);
          for (int tt = 0; tt < NSTEP; ++tt)
            for (int k = 3; k >= 0; --k) rev = fmaf(A[i * K + 4 * tt + k], B[(4 * tt + k) * 16 + j], rev);
          float d = D[i * 16 + j];
          total++;
          if (memcmp(&d, &acc, 4)) mism_chain++;
          if (memcmp(&d, &rev, 4)) mism_rev++;
        }
    }
    printf("[A] f32 mfma 16x16x4 x%d: total=%ld mismatch_vs_k-ordered_fmaf_chain=%ld mismatch_vs_reversed=%ld\n",
           NSTEP, total, mism_chain, mism_rev);
  }
  // ---------------- (B) ----------------
  {
    const int TRIALS = 128;
    std::vector<uint16_t> A(TRIALS * 512), B(TRIALS * 512);
    std::vector<float> C(TRIALS * 256), D(TRIALS * 256);
    uint16_t *dA, *dB; float *dC, *dD;
    CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024)); CK(hipMalloc(&dD, 1024));
    for (int t = 0; t < TRIALS; ++t) {
      for (int i = 0; i < 512; ++i) {
        float a = (t % 3 == 0) ? rnd_unit() : rnd_wide(), b = (t % 3 == 0) ? rnd_unit() : rnd_wide();
        A[t * 512 + i] = dsm_f32_to_bf16(a);
        B[t * 512 + i] = dsm_f32_to_bf16(b);
      }
      for (int i = 0; i < 256; ++i) C[t * 256 + i] = (t % 2) ? rnd_wide() : 0.0f;
      CK(hipMemcpy(dA, &A[t * 512], 1024, hipMemcpyHostToDevice));
      CK(hipMemcpy(dB, &B[t * 512], 1024, hipMemcpyHostToDevice));
      CK(hipMemcpy(dC, &C[t * 256], 1024, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(mfma_bf16_one, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
      CK(hipMemcpy(&D[t * 256], dD, 1024, hipMemcpyDeviceToHost));
    }
    snprintf(path, sizeof path, "%s/probe_bf16_A.bin", outdir); dump(path, A.data(), A.size() * 2);
    snprintf(path, sizeof path, "%s/probe_bf16_B.bin", outdir); dump(path, B.data(), B.size() * 2);
    snprintf(path, sizeof path, "%s/probe_bf16_C.bin", outdir); dump(path, C.data(), C.size() * 4);
    snprintf(path, sizeof path, "%s/probe_bf16_D.bin", outdir); dump(path, D.data(), D.size() * 4);
    // quick in-program hypothesis: k-ordered fmaf chain
    long mism = 0, tot = 0;
    for (int t = 0; t < TRIALS; ++t)
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          float acc = C[t * 256 + i * 16 + j];
          for (int k = 0; k < 32; ++k)
            acc = fmaf(dsm_bf16_to_f32(A[t * 512 + i * 32 + k]), dsm_bf16_to_f32(B[t * 512 + k * 16 + j]), acc);
          float d = D[t * 256 + i * 16 + j];
          tot++;
          if (memcmp(&d, &acc, 4)) mism++;
        }
    printf("[B] bf16 mfma 16x16x32: total=%ld mismatch_vs_k-ordered_fmaf_chain=%ld (raw dumped)\n", tot, mism);
  }
  // ---------------- (C) ----------------
  {
    const int N = 1 << 18;
    std::vector<float> x(N), y(N);
    for (int i = 0; i < N; ++i) {
      int m = i & 7;
      float u = rnd_unit();
      x[i] = m == 0 ? u * 100.0f : m == 1 ? u * 10.0f : m == 2 ? u : m == 3 ? u * 4.0f : m == 4 ? u * 0.01f : m == 5 ? u * 88.0f : m == 6 ? u * 6.0f : rnd_wide();
      y[i] = rnd_wide();
      if (y[i] == 0.0f) y[i] = 1.0f;
    }
    float *dx, *dy, *o[9];
    CK(hipMalloc(&dx, N * 4)); CK(hipMalloc(&dy, N * 4));
    for (int k = 0; k < 9; ++k) CK(hipMalloc(&o[k], N * 4));
    CK(hipMemcpy(dx, x.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, y.data(), N * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(numerics_kernel, dim3(N / 256), dim3(256), 0, 0, dx, dy, N, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8]);
    CK(hipDeviceSynchronize());
    snprintf(path, sizeof path, "%s/probe_num_x.bin", outdir); dump(path, x.data(), N * 4);
    snprintf(path, sizeof path, "%s/probe_num_y.bin", outdir); dump(path, y.data(), N * 4);
    const char* names[9] = {"exp", "elu", "silu", "gelu", "sin", "cos", "div", "rsqrt", "muladd"};
    std::vector<float> h(N);
    for (int k = 0; k < 9; ++k) {
      CK(hipMemcpy(h.data(), o[k], N * 4, hipMemcpyDeviceToHost));
      snprintf(path, sizeof path, "%s/probe_num_%s.bin", outdir, names[k]);
      dump(path, h.data(), N * 4);
      // in-program: device vs host-clang build of the same functions
      long mism = 0;
      for (int i = 0; i < N; ++i) {
        float a = x[i], b = y[i], ref = 0, s, c;
        switch (k) {
          case 0: ref = dsm_expf(a); break;
          case 1: ref = dsm_elu(a); break;
          case 2: ref = dsm_silu(a); break;
          case 3: ref = dsm_gelu_erf(a); break;
          case 4: dsm_sincosf(fabsf(a) * 1000.0f, &s, &c); ref = s; break;
          case 5: dsm_sincosf(fabsf(a) * 1000.0f, &s, &c); ref = c; break;
          case 6: ref = a / b; break;
          case 7: ref = 1.0f / sqrtf(fabsf(b) + 1e-8f); break;
          case 8: { volatile float m = a * b; ref = m + a; } break;
        }
        if (memcmp(&ref, &h[i], 4) && !(ref != ref && h[i] != h[i])) mism++;
      }
      printf("[C] %-7s device-vs-host(clang) mismatches: %ld / %d\n", names[k], mism, N);
    }
  }
  return 0;
}
