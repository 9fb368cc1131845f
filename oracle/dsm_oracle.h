/*
 * dsm_oracle.h — CPU oracle for the batched STT hot path.  TEST INFRASTRUCTURE ONLY:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libdsm_mi355x.so) never links or calls it.
 *
 * It is a plain-C restatement of the reference's algorithm (Rust, moshi-core), function by
 * function, with the canonical reduction orders of csrc/dsm_numerics.h so that the HIP
 * kernels can be compared bit-for-bit.  Pinning: the kv-cache known-answer vectors of
 * core/kv_cache.rs:339-405 and the streaming==batch conv property of core/conv.rs:698-723
 * are reproduced in tests/; model outputs are "parity unpinned" against Candle itself (the
 * reference cannot be built offline and ships no golden outputs) — see DESIGN.md.
 */
#ifndef DSM_ORACLE_H
#define DSM_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/dsm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_asr orc_asr;

/* Same contract as dsm_asr_create / dsm_mimi_encode_step / dsm_asr_step_tokens / ... */
orc_asr* orc_asr_create(const dsm_asr_config* cfg, int batch_size, const char* lm_safetensors,
                        const char* mimi_safetensors, char* err, size_t errcap);
void orc_asr_destroy(orc_asr*);
/* side: 0 = encoder-thread Mimi clone, 1 = model-side Mimi (asr::State::audio_tokenizer). */
int orc_mimi_encode_step(orc_asr*, int side, const float* pcm, const uint8_t* mask, uint32_t* codes_out);
/* Mimi::decode_step: codes [B*n_q] -> pcm_out [B*1920]; returns samples per slot */
int orc_mimi_decode_step(orc_asr*, int side, const uint32_t* codes, const uint8_t* mask, float* pcm_out);
int orc_asr_step_tokens(orc_asr*, const uint32_t* codes, const uint8_t* mask, uint32_t* text_tokens_out,
                        float* vad_prs_out);
int orc_asr_reset_slot(orc_asr*, int slot);
int orc_asr_set_seed(orc_asr*, int slot, uint64_t seed); /* temperature > 0: dsm_asr_set_seed's counterpart */
int orc_mimi_reset_slot(orc_asr*, int side, int slot);
int orc_asr_poll_msgs(orc_asr*, dsm_asr_msg* msgs, int cap, uint32_t* tokens_out, int tokens_cap);
int orc_debug_read(orc_asr*, const char* name, float* out, size_t cap);

void orc_set_num_threads(int n); /* OpenMP team size (small models: few threads) */

/* ---- unit-level entry points (tests of single ops; same canonical orders) ---- */
float orc_dot(const float* x, const float* w, int K);
void orc_linear(float* y, int ldy, const float* x, int ldx, const float* W, int ldw, const float* bias, int M,
                int N, int K);
void orc_rmsnorm(float* y, const float* x, const float* alpha, int rows, int d, float eps);
void orc_layernorm(float* y, const float* x, const float* w, const float* b, int rows, int d, float eps);
/* q [T][hd]; K,V [ctx][hd]; maskf [T][ctx] additive (0 / -inf); out [T][hd] */
void orc_attention_head(const float* q, int T, const float* K, const float* V, int ctx, int hd, const float* maskf,
                        float* out);
void orc_rope_table(int hd, int max_period, float* inv_freq /* [hd/2] */);
void orc_rope_apply(float* x /* [hd] in place */, int hd, const float* inv_freq, uint32_t pos);

/* ScatteredCacheBuilder (core/kv_cache.rs:54-295) */
typedef struct orc_kvb orc_kvb;
orc_kvb* orc_kvb_new(int batch_size, int context);
void orc_kvb_free(orc_kvb*);
void orc_kvb_reset_batch_index(orc_kvb*, int b);
/* indices_out [B*T] u32, mask_out [B*T*ctx] f32 (0 / -inf) */
void orc_kvb_indices_and_mask(orc_kvb*, int seq_len, const uint8_t* batch_mask, uint32_t* indices_out,
                              float* mask_out);
void orc_kvb_get(const orc_kvb*, uint32_t* positions, uint32_t* indices);

/* StreamableConv1d / StreamableConvTranspose1d (core/conv.rs:226-502), channels-last [B][T][C].
 * weight is the checkpoint layout: conv [out_c][in_c][k], convtr [in_c][out_c][k]. */
typedef struct orc_conv1d orc_conv1d;
orc_conv1d* orc_conv1d_new(int batch, int in_c, int out_c, int k, int stride, int dilation, int replicate_pad,
                           const float* weight, const float* bias);
void orc_conv1d_free(orc_conv1d*);
/* returns number of output frames (may be 0); y must hold B*max_frames*out_c */
int orc_conv1d_step(orc_conv1d*, const float* x, int T, const uint8_t* mask, float* y, int y_cap_frames);
int orc_conv1d_forward(orc_conv1d*, const float* x, int T, float* y, int y_cap_frames); /* non-streaming (conv.rs:284-304) */
void orc_conv1d_reset_state(orc_conv1d*);
void orc_conv1d_reset_batch_idx(orc_conv1d*, int b);

typedef struct orc_convtr1d orc_convtr1d;
orc_convtr1d* orc_convtr1d_new(int batch, int in_c, int out_c, int k, int stride, int depthwise,
                               const float* weight, const float* bias);
void orc_convtr1d_free(orc_convtr1d*);
int orc_convtr1d_step(orc_convtr1d*, const float* x, int T, const uint8_t* mask, float* y, int y_cap_frames);
int orc_convtr1d_forward(orc_convtr1d*, const float* x, int T, float* y, int y_cap_frames);
void orc_convtr1d_reset_batch_idx(orc_convtr1d*, int b);

#ifdef __cplusplus
}
#endif
#endif

#ifdef __cplusplus
extern "C" {
#endif
float orc_expf(float x);
float orc_logf(float x);
float orc_elu(float x);
float orc_silu(float x);
float orc_gelu_erf(float x);
void orc_sincosf(float x, float* s, float* c);
uint16_t orc_f32_to_bf16(float x);
float orc_bf16_to_f32(uint16_t h);
#ifdef __cplusplus
}
#endif

/* dot_mode 1 ("bx3"): the header's reference statement of v_mfma_f32_16x16x32_bf16 and the oracle's fast form of one group */
uint32_t orc_bf16_mfma32(uint32_t c_bits, const uint16_t* a /*[32]*/, const uint16_t* b /*[32]*/);
uint32_t orc_bx3_group8(uint32_t v_bits, const uint16_t* a /*[8]*/, const uint16_t* b /*[8]*/);
void orc_linear_mode(int mode);


/* ---- TTS step (config 5): see oracle/dsm_oracle_tts.inc ---- */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orc_tts orc_tts;
orc_tts* orc_tts_create(const dsm_tts_config* cfg, int batch_size, const char* lm_safetensors, char* err, size_t errcap);
void orc_tts_destroy(orc_tts*);
int orc_tts_step(orc_tts*, const uint32_t* prev_text_token, const int32_t* allowed, const uint8_t* mask,
                 uint32_t* text_token_out, uint32_t* audio_out);
int orc_tts_audio_tokens(orc_tts*, int slot, int step, uint32_t* out);
int orc_tts_step_idx(orc_tts*, int slot);
int orc_tts_reset_slot(orc_tts*, int slot);
int orc_tts_set_sampling(orc_tts*, int slot, int top_k, float temperature, uint64_t seed);
int orc_tts_set_ca_src(orc_tts*, int slot, const float* ca_src, int n, const float* ca_src_uncond, int n_uncond, double cfg_alpha);
uint32_t orc_sample_topk(const float* logits, int V, int k, float inv_t, const uint32_t* key, uint32_t* pos);
uint32_t orc_chacha_word(const uint32_t* key, uint64_t index, int rounds);
void orc_seed_from_u64(uint64_t seed, uint32_t* key);
float orc_uniform_f32(uint32_t u, float total);
int orc_tts_debug_read(orc_tts*, const char* name, float* out, size_t cap);
#ifdef __cplusplus
}
#endif
