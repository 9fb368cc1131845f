/*
 * dsm_numerics.h — the numerics contract shared by the HIP kernels (hipcc, gfx950)
 * and the CPU oracle (gcc).  Everything here is written with explicit fmaf/fma and
 * IEEE +,-,*,/ only (no libm, no fast-math, build with -ffp-contract=off), so the
 * two compilers produce bit-identical floats.  That is what lets the parity tests
 * compare Mimi codes / text tokens (integer outputs of a float pipeline) with
 * assert-equal instead of a tolerance.
 *
 * What each function restates (reference = /root/reference, Candle 0.9.1 ops it calls):
 *   dsm_expf      exp() inside softmax_last_dim (core/batched_transformer.rs:111, core/asr.rs:200),
 *                 ELU (core/seanet.rs:144,299,301) and SiLU (core/batched_transformer.rs:174)
 *   dsm_gelu_erf  Tensor::gelu_erf (core/batched_transformer.rs:169) — Candle evaluates it in f64
 *   dsm_sincos    freqs.sin()/cos() of RotaryEmbedding::rope (core/transformer.rs:394-402)
 *   bf16 helpers  dtype casts of the bf16 LM checkpoint (srv/batched_asr.rs:738-745)
 */
#ifndef DSM_NUMERICS_H
#define DSM_NUMERICS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define DSM_HD __host__ __device__ __forceinline__
#define DSM_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#define DSM_FMA(a, b, c) __builtin_fma((a), (b), (c))
#else
#define DSM_HD static inline __attribute__((always_inline))
#define DSM_FMAF(a, b, c) __builtin_fmaf((a), (b), (c))
#define DSM_FMA(a, b, c) __builtin_fma((a), (b), (c))
#endif

#include "dsm_numerics_tables.h"

DSM_HD float dsm_u32_as_f32(uint32_t u) {
  union { uint32_t u; float f; } c;
  c.u = u;
  return c.f;
}
DSM_HD uint32_t dsm_f32_as_u32(float f) {
  union { uint32_t u; float f; } c;
  c.f = f;
  return c.u;
}

/* bf16 <-> f32.  Upcast is exact; downcast is round-to-nearest-even on the bit pattern
 * (inputs are finite in this engine; NaN payloads are not preserved). */
DSM_HD float dsm_bf16_to_f32(uint16_t h) { return dsm_u32_as_f32(((uint32_t)h) << 16); }
DSM_HD uint16_t dsm_f32_to_bf16(float f) {
  uint32_t u = dsm_f32_as_u32(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

#define DSM_INF_F (dsm_u32_as_f32(0x7F800000u))

/* dot_mode 1 ("bx3"): x = hi + mid + lo exactly, three bf16 pieces of 8 significand bits each — hi = x with the low 16 bits
 * of its pattern cleared, mid the same of x - hi (exact), lo = x - hi - mid (exact, at most 8 significant bits left).
 * Returned as bf16 bit patterns.  (Pieces of a value below 2^-110 may be bf16 subnormals; the matrix instruction takes
 * them at face value, csrc/dsm_bf16_mfma_model.h.) */
DSM_HD void dsm_split3(float x, uint16_t* hi, uint16_t* mid, uint16_t* lo) {
  const uint32_t u = dsm_f32_as_u32(x);
  const float r1 = x - dsm_u32_as_f32(u & 0xFFFF0000u);
  const uint32_t u1 = dsm_f32_as_u32(r1);
  const float r2 = r1 - dsm_u32_as_f32(u1 & 0xFFFF0000u);
  *hi = (uint16_t)(u >> 16);
  *mid = (uint16_t)(u1 >> 16);
  *lo = (uint16_t)(dsm_f32_as_u32(r2) >> 16);
}

/* e^x, ~1 ulp, flushes to 0 below -87.3 (keeps every result normal or zero, so the
 * function does not depend on the denormal mode of either machine). */
DSM_HD float dsm_expf(float x) {
  if (!(x > -87.3f)) return 0.0f;
  if (x > 88.72f) return DSM_INF_F;
  float t = x * 1.44269504088896341f;
  float n = (t + 12582912.0f) - 12582912.0f; /* round-to-nearest-even, |t| < 2^22 */
  float r = DSM_FMAF(n, -0.693145751953125f, x);
  r = DSM_FMAF(n, -1.428606765330187045e-06f, r);
  float p = 1.9875691500e-4f;
  p = DSM_FMAF(p, r, 1.3981999507e-3f);
  p = DSM_FMAF(p, r, 8.3334519073e-3f);
  p = DSM_FMAF(p, r, 4.1665795894e-2f);
  p = DSM_FMAF(p, r, 1.6666665459e-1f);
  p = DSM_FMAF(p, r, 5.0000001201e-1f);
  float z = r * r;
  float y = DSM_FMAF(p, z, r) + 1.0f;
  int ni = (int)n;
  int h = ni >> 1; /* arithmetic shift: floor(ni/2) */
  float s1 = dsm_u32_as_f32((uint32_t)(127 + h) << 23);
  float s2 = dsm_u32_as_f32((uint32_t)(127 + (ni - h)) << 23);
  return (y * s1) * s2;
}

/* ln(x) for finite x > 0 (normal or subnormal), < 1 ulp: x = 2^e m with m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f),
 * ln(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with R the degree-4 minimax polynomial in s^2 that fdlibm's logf publishes
 * (Lg1..Lg4).  IEEE +, -, *, / only, in a fixed order: the same bits from gcc and hipcc (r04: the Gumbel noise of
 * candle_nn::sampling::gumbel_softmax, core/asr.rs:211-215). */
DSM_HD float dsm_logf(float x) {
  uint32_t ix = dsm_f32_as_u32(x);
  int k = 0;
  if (ix < 0x00800000u) { /* subnormal: scale up by 2^25 */
    x = x * 33554432.0f;
    ix = dsm_f32_as_u32(x);
    k = -25;
  }
  k += (int)(ix >> 23) - 127;
  ix &= 0x007FFFFFu;
  const uint32_t i = (ix + (0x95f64u << 3)) & 0x800000u; /* m >= sqrt(2): halve it */
  const float m = dsm_u32_as_f32(ix | (i ^ 0x3F800000u));
  k += (int)(i >> 23);
  const float f = m - 1.0f;
  const float s = f / (2.0f + f);
  const float dk = (float)k;
  const float z = s * s;
  const float w = z * z;
  const float t1 = w * (0.40000972152f + w * 0.24279078841f);
  const float t2 = z * (0.66666662693f + w * 0.28498786688f);
  const float R = t2 + t1;
  const float hfsq = (0.5f * f) * f;
  return dk * 6.9313812256e-01f - ((hfsq - (s * (hfsq + R) + dk * 9.0580006145e-06f)) - f);
}

/* candle Activation::Elu(1.0):  x >= 0 ? x : exp(x) - 1 */
DSM_HD float dsm_elu(float x) { return x >= 0.0f ? x : dsm_expf(x) - 1.0f; }

/* candle silu:  x / (1 + exp(-x)) */
DSM_HD float dsm_silu(float x) { return x / (1.0f + dsm_expf(-x)); }

/* erf in f64 by the Maclaurin series (64 terms, Horner in x^2); |x| >= 4 saturates to +-1
 * (1 - erf(4) = 1.5e-8 is below half an f32 ulp of 1).  Series cancellation error at
 * |x| = 4 is ~1e-11 absolute — invisible after the final rounding to f32. */
DSM_HD double dsm_erf_d(double x) {
  if (x >= 4.0) return 1.0;
  if (x <= -4.0) return -1.0;
  double z = x * x;
  double p;
  DSM_ERF_HORNER(p, z);
  return (DSM_TWO_OVER_SQRTPI * x) * p;
}

/* candle gelu_erf for f32: evaluated in f64, (erf(v/sqrt2) + 1) * 0.5 * v, rounded once. */
DSM_HD float dsm_gelu_erf(float v) {
  double d = (double)v;
  double e = dsm_erf_d(d * DSM_INV_SQRT2);
  return (float)(((e + 1.0) * 0.5) * d);
}

/* sin/cos of a non-negative f32 angle (RoPE: position * inv_freq, both f32), evaluated in
 * f64 with a 3-term Cody-Waite reduction (exact k*PIO2_1 for k < 2^20, i.e. angles < 1.6e6)
 * and rounded to f32 once. */
DSM_HD void dsm_sincosf(float xf, float* s_out, float* c_out) {
  double x = (double)xf;
  double kf = (x * DSM_TWO_OVER_PI + 6755399441055744.0) - 6755399441055744.0;
  double r = DSM_FMA(-kf, DSM_PIO2_1, x);
  r = DSM_FMA(-kf, DSM_PIO2_2, r);
  r = DSM_FMA(-kf, DSM_PIO2_3, r);
  double z = r * r;
  double ps, pc;
  DSM_SIN_HORNER(ps, z);
  DSM_COS_HORNER(pc, z);
  double sn = DSM_FMA(ps * z, r, r);
  double cs = DSM_FMA(pc, z, 1.0);
  long long q = (long long)kf;
  double s, c;
  switch (q & 3) {
    case 0: s = sn; c = cs; break;
    case 1: s = cs; c = -sn; break;
    case 2: s = -sn; c = -cs; break;
    default: s = -cs; c = sn; break;
  }
  *s_out = (float)s;
  *c_out = (float)c;
}

/* ---------------------------------------------------------------------------------
 * Canonical reduction orders (see DESIGN.md "Numerics contract").
 *
 * DOT ORDER ("mfma order"): a K-long dot product is accumulated as ONE fmaf chain per
 * K-chunk of DSM_KC elements; inside a chunk the elements are visited 32 at a time in
 * the order  k = 32*blk + 8*q + s  for s in 0..7 (outer), q in 0..3 (inner)  — which is
 * exactly what consecutive v_mfma_f32_16x16x4_f32 instructions compute when lane group
 * q = lane>>4 holds elements 8q..8q+7 of the block.  K is zero-padded to a multiple of
 * 32 (fmaf(0,0,acc) == acc).  Chunk partial sums are then added left to right:
 * ((c0 + c1) + c2) + ...  and the bias / residual are added last.
 * --------------------------------------------------------------------------------- */
#define DSM_KC 256

/* ROW SUM (norms): 256 threads per row, thread t chains elements 1024*it + 4*t + j; per-wave butterfly;
 * the 4 wave totals are added left to right.
 * WAVE SUM: 64 lane partials combined by an xor butterfly, offsets 32,16,8,4,2,1
 * (== __shfl_xor all-reduce).  In-place; every entry ends up holding the total. */
DSM_HD void dsm_butterfly_sum(float* p, int width /* power of two <= 64 */) {
  for (int off = width >> 1; off >= 1; off >>= 1) {
    float t[64];
    for (int l = 0; l < width; ++l) t[l] = p[l] + p[l ^ off];
    for (int l = 0; l < width; ++l) p[l] = t[l];
  }
}

#endif /* DSM_NUMERICS_H */
