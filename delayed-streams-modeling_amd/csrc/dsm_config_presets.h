/*
 * dsm_config_presets.h — the shipped configurations as C structs.
 * Sources: core/mimi.rs:32-93 (Mimi v0_1), configs/stt/config-stt-en_fr-hf.toml:18-56,
 * configs/stt/config-stt-en-hf.toml:18-49 (reference tree /root/reference).
 */
#ifndef DSM_CONFIG_PRESETS_H
#define DSM_CONFIG_PRESETS_H
#include <string.h>
#include "../../include/dsm.h"

static inline void dsm_preset_mimi_v0_1(dsm_mimi_config* m, int num_codebooks) {
  memset(m, 0, sizeof *m);
  m->channels = 1;
  m->dimension = 512;
  m->n_filters = 64;
  m->n_residual_layers = 1;
  m->n_ratios = 4;
  m->ratios[0] = 8; m->ratios[1] = 6; m->ratios[2] = 5; m->ratios[3] = 4;
  m->kernel_size = 7;
  m->residual_kernel_size = 3;
  m->last_kernel_size = 3;
  m->dilation_base = 2;
  m->compress = 2;
  m->transformer.d_model = 512;
  m->transformer.num_heads = 8;
  m->transformer.num_layers = 8;
  m->transformer.dim_feedforward = 2048;
  m->transformer.context = 250;
  m->transformer.max_period = 10000;
  m->transformer.gating = 0;
  m->transformer.norm = 0;
  m->transformer.positional_embedding = 1;
  m->transformer.layer_scale = 1;
  m->transformer.conv_layout = 1;
  m->quantizer_n_q = num_codebooks > 0 ? num_codebooks : 16;
  m->quantizer_bins = 2048;
  m->quantizer_dim = 256;
  m->downsample_stride = 2; /* 24000 / (8*6*5*4) = 25 Hz encoder rate over 12.5 Hz frame rate */
}

static inline void dsm_preset_stt_common(dsm_asr_config* c) {
  memset(c, 0, sizeof *c);
  c->lm.d_model = 2048;
  c->lm.dim_feedforward = 8192;
  c->lm.max_period = 100000;
  c->lm.gating = 1;
  c->lm.norm = 1;
  c->lm.positional_embedding = 1;
  c->lm.layer_scale = 0;
  c->lm.conv_layout = 0;
  c->audio_vocab_size = 2049;
  c->audio_codebooks = 32;
  c->temperature = 0.0f;
  c->kv_bf16 = 1;
  c->dot_mode = 1; /* r04: the shipped presets select the bf16 matrix instruction ("bx3", include/dsm.h) — the mode bench.py times;
                      against dot_mode 0 at these dimensions: logits within 1.4e-6 (f32 ring) / 1.8e-4 (bf16 ring) of the row's
                      largest, 480 / 480 text tokens equal (tests/test_dot_mode_agreement_gpu.py, profiles/r04/dot_mode_agreement.json) */
  dsm_preset_mimi_v0_1(&c->mimi, 32); /* srv/batched_asr.rs:754-757: Config::v0_1(Some(audio_codebooks)) */
}

static inline void dsm_preset_stt_1b_en_fr(dsm_asr_config* c) {
  dsm_preset_stt_common(c);
  c->lm.num_heads = 16;
  c->lm.num_layers = 16;
  c->lm.context = 750;
  c->text_in_vocab_size = 8001;
  c->text_out_vocab_size = 8000;
  c->extra_heads_num = 4;
  c->extra_heads_dim = 6;
  c->asr_delay_in_tokens = 6;
}

static inline void dsm_preset_stt_2_6b_en(dsm_asr_config* c) {
  dsm_preset_stt_common(c);
  c->lm.num_heads = 32;
  c->lm.num_layers = 48;
  c->lm.context = 375;
  c->text_in_vocab_size = 4001;
  c->text_out_vocab_size = 4000;
  c->extra_heads_num = 0;
  c->extra_heads_dim = 0;
  c->asr_delay_in_tokens = 32;
}

/* tts_streaming::Config::v202501 (core/tts_streaming.rs:29-44) + configs/tts/config-tts.toml:35-79.  The toml's depformer
 * transformer (num_heads = 11, head_dim = 1024 with d_model = 1024) cannot run in the reference (core/transformer.rs:538-541
 * reshape fails); "11" is the number of weight groups.  A consistent shape is used instead:
 * d 1024, 16 heads x 64, 4 layers, dim_feedforward 3072 (hidden 2048), 32 slices, low-rank 128, 11 weight groups. */
static inline void dsm_preset_tts_v202501(dsm_tts_config* c) {
  memset(c, 0, sizeof *c);
  c->lm.d_model = 2048; c->lm.num_heads = 16; c->lm.num_layers = 16; c->lm.dim_feedforward = 8192;
  c->lm.context = 1024; c->lm.max_period = 100000; c->lm.gating = 1; c->lm.norm = 1; c->lm.positional_embedding = 1;
  c->text_in_vocab_size = 8001; c->text_out_vocab_size = 8000; c->audio_vocab_size = 2049; c->audio_codebooks = 32;
  c->depformer.d_model = 1024; c->depformer.num_heads = 16; c->depformer.num_layers = 4; c->depformer.dim_feedforward = 3072;
  c->depformer.context = 32; c->depformer.max_period = 10000; c->depformer.gating = 1; c->depformer.norm = 1;
  c->depformer.positional_embedding = 0;
  c->dep_num_slices = 32; c->dep_low_rank = 128; c->dep_weight_groups = 11;
  c->acoustic_delay = 2; c->text_eop_token = 0; c->text_bos_token = 1; c->text_eos_token = 2; c->text_pad_token = 3;
  c->text_start_token = 8000; c->text_audio_delay_in_tokens = 25; c->max_consecutive_pads = 10; c->max_steps = 4096;
  c->kv_bf16 = 1;
  /* core/lm.rs:392-396 tts_202501: cross_attention = Some((CrossAttentionGating::Normal, NormType::LayerNorm, None)), and the
   * server always builds its State with Some(CaSrc::Tokens(..)) (srv/tts.rs:426-441): the preset carries the branch the
   * reference runs.  Sources: speaker_cond_n_speakers = 5 x 25 rows at 12.5 Hz -> 125 rows; 128 fit.  Guidance
   * (two batch rows per slot, cfg_alpha) stays opt-in through cfg_rows, as it is per request in the reference. */
  c->cross_attention = 1; c->ca_norm = 0; c->ca_dim = 0; c->ca_max_len = 128; c->cfg_rows = 0;
  c->dot_mode = 1; /* as the STT presets */
}
#endif
