// bf16_adder_probe.hip — data for a bit-exact model of v_mfma_f32_16x16x32_bf16's internal accumulation.
//
// Every product of two bf16 values is exact in f32 (8 x 8 significand bits), so what the instruction adds up is known
// exactly; what is not documented is HOW it adds: grouping of the 32 products, alignment width, truncation or rounding of
// the aligned terms, where the accumulator enters, the final rounding.  This program records N independent dot products
//     d = mfma(a[0..32), b[0..32), c)
// (the 16 diagonal outputs of a tile have free rows / columns: 16 dot products per instruction) under several input
// families, and dumps (a, b, c, d) for experiments/bf16_adder_fit.py, which tests closed-form adder models offline.
//   family 0  unit-range random values               family 1  wide exponents (2^-20 .. 2^20)
//   family 2  sparse: 2-4 non-zero products at random k, exponent gaps 0..40, accumulator zero / random
//   family 3  cancellation: products in +/- pairs that almost cancel, small survivors
//   family 4  one large product + 31 small ones just below its half-ulp (sticky / truncation behaviour)
// usage: bf16_adder_probe <outdir> [tiles per family = 4096]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../delayed-streams-modeling_amd/csrc/dsm_bf16_mfma_model.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP %s @%d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

// one wave per tile: A [16][32] bf16 row-major, B [32][16], C / D [16][16]
__global__ void mfma_tiles(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, const float* __restrict__ C,
                           float* __restrict__ D, int ntiles) {
  const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const int l = threadIdx.x & 63, r = l & 15, q = l >> 4;
  const uint16_t* a_ = A + (size_t)tile * 512;
  const uint16_t* b_ = B + (size_t)tile * 512;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)a_[r * 32 + 8 * q + j];
    b[j] = (short)b_[(8 * q + j) * 16 + r];
  }
  f32x4 acc;
  for (int i = 0; i < 4; ++i) acc[i] = C[(size_t)tile * 256 + (q * 4 + i) * 16 + r];
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[(size_t)tile * 256 + (q * 4 + i) * 16 + r] = acc[i];
}

static uint64_t rs = 0x9E3779B97F4A7C15ull;
static uint32_t rnd32() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }
static float rnd_unit() { return (float)((rnd32() >> 8) * (1.0 / 16777216.0)) * 2.0f - 1.0f; }
static uint16_t bf(float x) {  // round to nearest even
  uint32_t u; memcpy(&u, &x, 4);
  u = u + 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t rnd_bf_exp(int e) {  // random sign and 8-bit significand, exponent e
  const uint32_t sig = 0x80u | (rnd32() & 0x7Fu);
  return bf(ldexpf((float)sig, e - 7) * ((rnd32() & 1) ? -1.0f : 1.0f));
}

static int run_round(const char* outdir, int per, unsigned long long seed, bool dump_raw) {
  rs = seed;
  const int NF = 11, ntiles = NF * per;
  std::vector<uint16_t> A((size_t)ntiles * 512, 0), B((size_t)ntiles * 512, 0);
  std::vector<float> C((size_t)ntiles * 256, 0.0f), D((size_t)ntiles * 256);
  for (int t = 0; t < ntiles; ++t) {
    const int fam = t / per;
    uint16_t* a = &A[(size_t)t * 512];
    uint16_t* b = &B[(size_t)t * 512];
    float* c = &C[(size_t)t * 256];
    for (int i = 0; i < 16; ++i) {  // the dot product that lands on output (i, i): row i of A, column i of B
      auto setk = [&](int k, uint16_t av, uint16_t bv) { a[i * 32 + k] = av; b[k * 16 + i] = bv; };
      if (fam == 0) {
        for (int k = 0; k < 32; ++k) setk(k, bf(rnd_unit()), bf(rnd_unit()));
        c[i * 16 + i] = (t & 1) ? rnd_unit() * 4.0f : 0.0f;
      } else if (fam == 1) {
        for (int k = 0; k < 32; ++k) setk(k, rnd_bf_exp((int)(rnd32() % 21) - 10), rnd_bf_exp((int)(rnd32() % 21) - 10));
        c[i * 16 + i] = (t & 1) ? ldexpf(rnd_unit(), (int)(rnd32() % 41) - 20) : 0.0f;
      } else if (fam == 2) {
        const int m = 2 + (int)(rnd32() % 3);
        const int e0 = (int)(rnd32() % 9) - 4;
        for (int j = 0; j < m; ++j) {
          const int k = (int)(rnd32() % 32);
          const int gap = j == 0 ? 0 : (int)(rnd32() % 41);
          setk(k, rnd_bf_exp(e0 - gap / 2), rnd_bf_exp(-(gap - gap / 2)));
        }
        const int mode = (int)(rnd32() % 3);
        c[i * 16 + i] = mode == 0 ? 0.0f : ldexpf(rnd_unit(), e0 - (int)(rnd32() % 41) + 8);
      } else if (fam == 3) {
        for (int k = 0; k < 32; k += 2) {
          const int e = (int)(rnd32() % 7) - 3;
          const uint16_t x = rnd_bf_exp(e), y = rnd_bf_exp(0);
          setk(k, x, y);
          // partner: nearly the negative (one significand step away now and then)
          uint16_t x2 = x ^ 0x8000u;
          if (rnd32() & 1) x2 = (uint16_t)(x2 + ((rnd32() & 1) ? 1 : -1));
          setk(k + 1 < 32 ? k + 1 : k, x2, y);
        }
        c[i * 16 + i] = (t & 1) ? ldexpf(rnd_unit(), -(int)(rnd32() % 20)) : 0.0f;
      } else if (fam == 5) {  // single product + accumulator, the accumulator the larger by 0..45 binades
        const int k = (int)(rnd32() % 32), gap = (int)(rnd32() % 46);
        setk(k, rnd_bf_exp(-gap / 2), rnd_bf_exp(-(gap - gap / 2)));
        c[i * 16 + i] = ldexpf((float)(0x800000u | (rnd32() & 0x7FFFFFu)), -23 + (int)(rnd32() % 3)) * ((rnd32() & 1) ? -1.f : 1.f);
      } else if (fam == 6) {  // single product + accumulator, the product the larger by 0..45 binades
        const int k = (int)(rnd32() % 32), gap = (int)(rnd32() % 46);
        setk(k, rnd_bf_exp(0), rnd_bf_exp((int)(rnd32() % 3)));
        c[i * 16 + i] = ldexpf((float)(0x800000u | (rnd32() & 0x7FFFFFu)), -23 - gap) * ((rnd32() & 1) ? -1.f : 1.f);
      } else if (fam == 7) {  // two products in ONE group of 8, no accumulator: product-product alignment
        const int g = (int)(rnd32() % 4), k0 = 8 * g + (int)(rnd32() % 8);
        int k1 = 8 * g + (int)(rnd32() % 8);
        if (k1 == k0) k1 = 8 * g + ((k0 + 1) & 7);
        const int gap = (int)(rnd32() % 36);
        setk(k0, rnd_bf_exp(0), rnd_bf_exp((int)(rnd32() % 2)));
        setk(k1, rnd_bf_exp(-gap / 2), rnd_bf_exp(-(gap - gap / 2)));
      } else if (fam == 8) {  // group 0: p0 + p1 (more than 24 bits between them); group 1..3: -p0: which low bits of p1 survived the hand-over?
        const int gap = 4 + (int)(rnd32() % 24), g1 = 1 + (int)(rnd32() % 3);
        const uint16_t x = rnd_bf_exp(0), y = rnd_bf_exp(0);
        setk((int)(rnd32() % 4), x, y);
        setk(4 + (int)(rnd32() % 4), rnd_bf_exp(-gap / 2), rnd_bf_exp(-(gap - gap / 2)));
        setk(8 * g1 + (int)(rnd32() % 8), (uint16_t)(x ^ 0x8000u), y);
        c[i * 16 + i] = (t & 1) ? 0.0f : ldexpf(rnd_unit(), -(int)(rnd32() % 30));
      } else if (fam == 10) {  // bf16 subnormal / tiny inputs among ordinary ones, tiny accumulators
        for (int k = 0; k < 32; ++k) {
          const uint32_t r = rnd32() % 4;
          const uint16_t x = r == 0 ? (uint16_t)((rnd32() & 0x807Fu)) /* subnormal or zero */ : r == 1 ? rnd_bf_exp(-120 + (int)(rnd32() % 10)) : rnd_bf_exp((int)(rnd32() % 9) - 4);
          const uint16_t y = (rnd32() % 3 == 0) ? rnd_bf_exp(100 + (int)(rnd32() % 20)) : rnd_bf_exp((int)(rnd32() % 9) - 4);
          setk(k, x, y);
        }
        c[i * 16 + i] = (t & 1) ? ldexpf(rnd_unit(), -20 - (int)(rnd32() % 20)) : 0.0f;
      } else if (fam == 9) {  // eight random products in one group + accumulator, moderate exponent spread
        const int g = (int)(rnd32() % 4);
        for (int k = 8 * g; k < 8 * g + 8; ++k) setk(k, rnd_bf_exp((int)(rnd32() % 9) - 4), rnd_bf_exp((int)(rnd32() % 9) - 4));
        c[i * 16 + i] = (t % 3 == 0) ? 0.0f : ldexpf(rnd_unit(), (int)(rnd32() % 25) - 12);
      } else {
        const int kbig = (int)(rnd32() % 32);
        for (int k = 0; k < 32; ++k) {
          if (k == kbig) setk(k, rnd_bf_exp(0), rnd_bf_exp(0));
          else if (rnd32() % 3) setk(k, rnd_bf_exp(-12 - (int)(rnd32() % 8)), rnd_bf_exp(-(int)(rnd32() % 8) - 5));
        }
        c[i * 16 + i] = (t % 3 == 0) ? 0.0f : ldexpf(rnd_unit(), -(int)(rnd32() % 30));
      }
    }
  }
  uint16_t *dA, *dB; float *dC, *dD;
  CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mfma_tiles, dim3((ntiles + 3) / 4), dim3(256), 0, 0, dA, dB, dC, dD, ntiles);
  CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
  // keep only the diagonal dot products: per sample a[32], b[32] (bf16 bits), c, d
  const size_t n = (size_t)ntiles * 16;
  std::vector<uint16_t> sa(n * 32), sb(n * 32);
  std::vector<float> sc(n), sd(n);
  for (int t = 0; t < ntiles; ++t)
    for (int i = 0; i < 16; ++i) {
      const size_t s = (size_t)t * 16 + i;
      for (int k = 0; k < 32; ++k) { sa[s * 32 + k] = A[(size_t)t * 512 + i * 32 + k]; sb[s * 32 + k] = B[(size_t)t * 512 + k * 16 + i]; }
      sc[s] = C[(size_t)t * 256 + i * 16 + i];
      sd[s] = D[(size_t)t * 256 + i * 16 + i];
    }
  // every OFF-diagonal output is a dot product too (row i of A with column j of B, shared operands): check them all against
  // the model as well — 256 per tile
  long mism_model[16] = {0}, tot_model[16] = {0}, mism_exact = 0, skipped = 0;
  for (int t = 0; t < ntiles; ++t) {
    const int fam = t / per;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        uint16_t bv[32];
        for (int k = 0; k < 32; ++k) bv[k] = B[(size_t)t * 512 + k * 16 + j];
        uint32_t cbits, dbits;
        memcpy(&cbits, &C[(size_t)t * 256 + i * 16 + j], 4);
        memcpy(&dbits, &D[(size_t)t * 256 + i * 16 + j], 4);
        const uint32_t ex = (dbits >> 23) & 0xFF;
        if (ex == 0 || ex == 255) { ++skipped; continue; }  // subnormal / overflowed results: outside the model's range
        const uint32_t want = dsm_bfm_mfma32(cbits, &A[(size_t)t * 512 + i * 32], bv);
        tot_model[fam] += 1;
        if (want != dbits) {
          if (mism_model[fam]++ < 50) {
            fprintf(stderr, "MISMATCH family %d hw %08x model %08x c %08x a", fam, dbits, want, cbits);
            for (int k = 0; k < 32; ++k) fprintf(stderr, " %04x", A[(size_t)t * 512 + i * 32 + k]);
            fprintf(stderr, " b");
            for (int k = 0; k < 32; ++k) fprintf(stderr, " %04x", bv[k]);
            fprintf(stderr, "\n");
          }
        }
      }
  }
  for (size_t s = 0; s < n; ++s) {
    double acc = sc[s];
    for (int k = 0; k < 32; ++k) acc += (double)bf2f(sa[s * 32 + k]) * (double)bf2f(sb[s * 32 + k]);
    const float want = (float)acc;
    if (memcmp(&want, &sd[s], 4)) ++mism_exact;
  }
  long tm = 0, mm = 0;
  for (int f = 0; f < NF; ++f) { tm += tot_model[f]; mm += mism_model[f]; }
  printf("bf16_adder_probe seed %llu: %ld dot products checked against dsm_bf16_mfma_model.h: %ld mismatches (%ld results outside the normal f32 range skipped); per family:",
         (unsigned long long)seed, tm, mm, skipped);
  for (int f = 0; f < NF; ++f) printf(" %ld/%ld", mism_model[f], tot_model[f]);
  printf("\n   (for scale: %ld of the %zu diagonal dot products differ from the double-precision sum rounded once)\n", mism_exact, n);
  if (dump_raw) {
    char path[512];
    auto dump = [&](const char* name, const void* p, size_t bytes) {
      snprintf(path, sizeof path, "%s/%s", outdir, name);
      FILE* f = fopen(path, "wb");
      if (!f) { perror(path); exit(3); }
      fwrite(p, 1, bytes, f);
      fclose(f);
    };
    dump("bf16adder_a.bin", sa.data(), sa.size() * 2);
    dump("bf16adder_b.bin", sb.data(), sb.size() * 2);
    dump("bf16adder_c.bin", sc.data(), sc.size() * 4);
    dump("bf16adder_d.bin", sd.data(), sd.size() * 4);
  }
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dD));
  return mm == 0 ? 0 : 1;
}

int main(int argc, char** argv) {
  const char* outdir = argc > 1 ? argv[1] : "gpurun_out";
  const int per = argc > 2 ? atoi(argv[2]) : 1024;
  const int rounds = argc > 3 ? atoi(argv[3]) : 1;
  int bad = 0;
  for (int r = 0; r < rounds; ++r) bad += run_round(outdir, per, 0x9E3779B97F4A7C15ull + 0x1234567ull * (unsigned long long)r, r == 0);
  printf("%s\n", bad ? "MODEL DOES NOT HOLD" : "model holds on every checked dot product");
  return bad;
}
