/*
 * dsm_sampling.h — seeded top-k sampling shared by the HIP kernels and the CPU oracle (same source, both compilers):
 * what `candle_transformers::generation::LogitsProcessor` does for `Sampling::TopK { k, temperature }`
 * (reference call sites: core/lm.rs:674 `lp.sample(&logits)` in DepFormer::sample, core/tts_streaming.rs:188
 * `text_lp.sample`, configured at srv/tts.rs:401-415).  candle-transformers 0.9.1 and rand 0.9 are un-vendored crates,
 * absent offline, and the reference holds no vector of theirs: what follows restates their PUBLISHED algorithms; it is
 * "parity unpinned" against Candle, pinned only between this engine and its oracle.
 *
 *   prs    = softmax_last_dim(logits / temperature)            (over the WHOLE vocabulary, then the k largest are kept)
 *   top-k  = the k largest probabilities.  candle picks them with `select_nth_unstable_by`, whose order among the k
 *            survivors is an implementation detail of the Rust standard library; here the order is fixed:
 *            probability descending, token id ascending.  k >= vocabulary: all tokens in id order (candle's
 *            `sample_multinomial(prs)` — that case IS order-defined).
 *   draw   = rand::distr::weighted::WeightedIndex: running f32 sums c_i = p_0 + ... + p_i, total = c_{k-1},
 *            x = Uniform[0, total) from ONE u32 of the generator: ((u >> 9) as mantissa of [1,2)) - 1, times total;
 *            result = number of c_i (i < k-1) with c_i <= x.
 *   rng    = rand 0.9 `StdRng` = ChaCha12, 64-bit block counter from 0, stream 0, key = `seed_from_u64(seed)` (eight u32
 *            from a PCG32 walk).  The reference seeds text_lp and audio_lp with the same query seed (two equal streams).
 */
#ifndef DSM_SAMPLING_H
#define DSM_SAMPLING_H

#include <stdint.h>
#include "dsm_numerics.h"

/* rand_core::SeedableRng::seed_from_u64: PCG32 (XSH-RR) output words, little endian, into the 32-byte seed */
DSM_HD void dsm_seed_from_u64(uint64_t state, uint32_t key[8]) {
  for (int i = 0; i < 8; ++i) {
    state = state * 6364136223846793005ull + 11634580027462260723ull;
    const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
    const uint32_t rot = (uint32_t)(state >> 59);
    key[i] = (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
  }
}

#define DSM_ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define DSM_QR(a, b, c, d)                                  \
  a += b; d ^= a; d = DSM_ROTL32(d, 16);                    \
  c += d; b ^= c; b = DSM_ROTL32(b, 12);                    \
  a += b; d ^= a; d = DSM_ROTL32(d, 8);                     \
  c += d; b ^= c; b = DSM_ROTL32(b, 7);

/* word `index` (0-based) of the ChaCha stream with `rounds` rounds (12 for StdRng), 64-bit counter, stream id 0 */
DSM_HD uint32_t dsm_chacha_word(const uint32_t key[8], uint64_t index, int rounds) {
  const uint64_t block = index >> 4;
  uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                     key[4], key[5], key[6], key[7], (uint32_t)block, (uint32_t)(block >> 32), 0u, 0u};
  uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7];
  uint32_t x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
  for (int r = 0; r < rounds; r += 2) {
    DSM_QR(x0, x4, x8, x12) DSM_QR(x1, x5, x9, x13) DSM_QR(x2, x6, x10, x14) DSM_QR(x3, x7, x11, x15)
    DSM_QR(x0, x5, x10, x15) DSM_QR(x1, x6, x11, x12) DSM_QR(x2, x7, x8, x13) DSM_QR(x3, x4, x9, x14)
  }
  const uint32_t out[16] = {x0 + in[0], x1 + in[1], x2 + in[2], x3 + in[3], x4 + in[4], x5 + in[5], x6 + in[6], x7 + in[7],
                            x8 + in[8], x9 + in[9], x10 + in[10], x11 + in[11], x12 + in[12], x13 + in[13], x14 + in[14], x15 + in[15]};
  return out[index & 15];
}

/* the sixteen words of block `block` of the same stream (word 16 block + i = out[i]) */
DSM_HD void dsm_chacha_block(const uint32_t key[8], uint64_t block, int rounds, uint32_t out[16]) {
  const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                           key[4], key[5], key[6], key[7], (uint32_t)block, (uint32_t)(block >> 32), 0u, 0u};
  uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7];
  uint32_t x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
  for (int r = 0; r < rounds; r += 2) {
    DSM_QR(x0, x4, x8, x12) DSM_QR(x1, x5, x9, x13) DSM_QR(x2, x6, x10, x14) DSM_QR(x3, x7, x11, x15)
    DSM_QR(x0, x5, x10, x15) DSM_QR(x1, x6, x11, x12) DSM_QR(x2, x7, x8, x13) DSM_QR(x3, x4, x9, x14)
  }
  out[0] = x0 + in[0]; out[1] = x1 + in[1]; out[2] = x2 + in[2]; out[3] = x3 + in[3];
  out[4] = x4 + in[4]; out[5] = x5 + in[5]; out[6] = x6 + in[6]; out[7] = x7 + in[7];
  out[8] = x8 + in[8]; out[9] = x9 + in[9]; out[10] = x10 + in[10]; out[11] = x11 + in[11];
  out[12] = x12 + in[12]; out[13] = x13 + in[13]; out[14] = x14 + in[14]; out[15] = x15 + in[15];
}

/* rand UniformFloat<f32>::sample for Uniform::new(0, total): one u32 -> [0, total) */
DSM_HD float dsm_uniform_f32(uint32_t u, float total) {
  const float value1_2 = dsm_u32_as_f32((u >> 9) | 0x3F800000u);
  const float value0_1 = value1_2 - 1.0f;
  return value0_1 * total + 0.0f;
}

/* candle_nn::sampling::gumbel_softmax(logits, temperature) — core/asr.rs:211-215 (temperature > 0):
 *   minus_g = logits.rand_like(1e-7, 0.999).log().neg().log();  sampled = argmax(logits - minus_g)            (temperature == 1)
 *                                                                 sampled = argmax(logits + minus_g * (-temperature))   (otherwise)
 * The reference draws from the device's UNSEEDED generator, one stream for the whole [batch, vocab] tensor: nothing to pin and
 * slot-coupled.  Here every slot has its own ChaCha12 stream (dsm_asr_set_seed), word j of the step's V words gives entry j:
 * u = 1e-7 + (0.999 - 1e-7) * U[0, 1) with U from the word's 23 high bits (rand's UniformFloat), the rest as candle writes it. */
DSM_HD float dsm_gumbel_value(float logit, uint32_t word, float temperature) {
  const float u01 = dsm_u32_as_f32((word >> 9) | 0x3F800000u) - 1.0f;
  const float u = u01 * (0.999f - 1e-7f) + 1e-7f;
  const float minus_g = dsm_logf(-dsm_logf(u));
  if (temperature == 1.0f) return logit - minus_g;
  return logit + minus_g * (-temperature);
}

/* sort key of (probability, token): descending key order = probability descending, token id ascending.
 * p >= 0, so its bit pattern orders like the float. */
DSM_HD uint64_t dsm_sample_key(float p, uint32_t token) { return ((uint64_t)dsm_f32_as_u32(p) << 32) | (uint64_t)(0xFFFFFFFFu - token); }
DSM_HD float dsm_sample_key_p(uint64_t key) { return dsm_u32_as_f32((uint32_t)(key >> 32)); }
DSM_HD uint32_t dsm_sample_key_token(uint64_t key) { return 0xFFFFFFFFu - (uint32_t)key; }

/* WeightedIndex draw over the first k entries of `keys` (already in the order they are offered to WeightedIndex).
 * Serial by definition (running f32 sums). */
DSM_HD uint32_t dsm_weighted_draw(const uint64_t* keys, int k, uint32_t u) {
  float total = 0.0f;
  for (int i = 0; i < k; ++i) total = total + dsm_sample_key_p(keys[i]);
  const float x = dsm_uniform_f32(u, total);
  float c = 0.0f;
  int idx = 0;
  for (int i = 0; i + 1 < k; ++i) { /* cumulative_weights holds k-1 entries; partition_point(|w| w <= x) */
    c = c + dsm_sample_key_p(keys[i]);
    if (c <= x) idx = i + 1; else break;
  }
  return dsm_sample_key_token(keys[idx]);
}

#endif /* DSM_SAMPLING_H */
