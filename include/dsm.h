/*
 * dsm.h — C ABI of libdsm_mi355x.so: the MI355X-native replacement for the per-80 ms-frame
 * batched step of moshi-server's BatchedAsr worker (reference: /root/reference,
 * server/rust/moshi/moshi-server/src/batched_asr.rs calling into moshi-core).
 *
 * The reference boundary is a Rust API (no FFI exists upstream).  Each entry point below
 * names the Rust call it replaces; INTEGRATION.md shows the `extern "C"` shim a maintainer
 * adds on the Rust side.  Plain pointers and sizes only — no torch / candle types.
 *
 * Conventions
 *   - return 0 on success, <0 on error (DSM_ERR_*); dsm_last_error() gives the message.
 *     The reference bubbles candle::Error up and kills the loop thread
 *     (srv/utils.rs:376-384); here the caller decides.
 *   - the caller owns every host buffer; the engine owns all device memory.
 *   - B = batch_size = number of stream slots; one frame = 1920 f32 samples @ 24 kHz = 80 ms
 *     (srv/batched_asr.rs:26).
 *   - mask[b] != 0  <=>  slot b is active this step (core/streaming.rs:20-68 StreamMask).
 *   - threading: like the reference, dsm_mimi_encode_step may run on an "encoder" thread
 *     while dsm_asr_step_tokens / dsm_asr_reset_slot run on a "model" thread
 *     (srv/batched_asr.rs:314-522); the two sides use separate HIP streams and state.
 */
#ifndef DSM_H
#define DSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSM_FRAME_SIZE 1920 /* srv/batched_asr.rs:26 FRAME_SIZE */
#define DSM_MAX_RATIOS 8
#define DSM_MAX_EXTRA_HEADS 8

enum {
  DSM_OK = 0,
  DSM_ERR_INVALID = -1,  /* bad argument / config (candle::bail! paths) */
  DSM_ERR_IO = -2,       /* weight file missing / malformed / tensor key or shape mismatch */
  DSM_ERR_DEVICE = -3,   /* HIP error */
  DSM_ERR_STATE = -4     /* call sequence the reference would reject */
};

/* transformer::Config — core/transformer.rs:21-53 (fields the batched path reads). */
typedef struct dsm_transformer_config {
  int d_model;
  int num_heads;
  int num_layers;
  int dim_feedforward;
  int context;          /* ring KV-cache length (core/kv_cache.rs:88-93) */
  int max_period;       /* RoPE theta (core/batched_transformer.rs:385-389) */
  int gating;           /* 0 = None (linear1/gelu_erf/linear2), 1 = silu gating (core/batched_transformer.rs:142-179) */
  int norm;             /* 0 = LayerNorm eps 1e-5, 1 = RmsNorm eps 1e-8 (core/batched_transformer.rs:236-252) */
  int positional_embedding; /* 0 = None, 1 = Rope */
  int layer_scale;      /* 0 = absent, 1 = layer_scale_{1,2}.scale present (core/batched_transformer.rs:291-304) */
  int conv_layout;      /* 1 = caller layout is [B,C,T] (Mimi), 0 = [B,T,C] (core/batched_transformer.rs:584-602) */
} dsm_transformer_config;

/* seanet::Config + mimi::Config — core/seanet.rs:12-31, core/mimi.rs:17-93 (v0_1 values in comments). */
typedef struct dsm_mimi_config {
  int channels;              /* 1 */
  int dimension;             /* 512 */
  int n_filters;             /* 64 */
  int n_residual_layers;     /* 1 */
  int n_ratios;              /* 4 */
  int ratios[DSM_MAX_RATIOS];/* {8,6,5,4} — the ENCODER applies them reversed (core/seanet.rs:194) */
  int kernel_size;           /* 7 */
  int residual_kernel_size;  /* 3 */
  int last_kernel_size;      /* 3 */
  int dilation_base;         /* 2 */
  int compress;              /* 2 */
  dsm_transformer_config transformer; /* d 512, 8 heads, 8 layers, ffn 2048, ctx 250, theta 10000, LayerNorm, layer_scale, conv_layout */
  int quantizer_n_q;         /* 32 for STT (srv/batched_asr.rs:754-757), 16 default */
  int quantizer_bins;        /* 2048 */
  int quantizer_dim;         /* 256 */
  int downsample_stride;     /* encoder_frame_rate / frame_rate = 25 / 12.5 = 2 (core/mimi.rs:141-144) */
} dsm_mimi_config;

/* ModuleConfig::BatchedAsr + lm::Config — srv/main.rs:110-128,179-184, core/lm.rs:36-46. */
typedef struct dsm_asr_config {
  dsm_transformer_config lm;
  int text_in_vocab_size;
  int text_out_vocab_size;
  int audio_vocab_size;
  int audio_codebooks;
  int extra_heads_num;   /* [modules.asr.model.extra_heads] num_heads (0 if absent) */
  int extra_heads_dim;
  int asr_delay_in_tokens;
  float temperature;     /* <= 0: argmax; > 0 (r04): candle_nn::sampling::gumbel_softmax (core/asr.rs:211-215) with a seeded ChaCha12
                            stream per slot (dsm_asr_set_seed) — the reference draws from an unseeded device generator: unpinnable */
  dsm_mimi_config mimi;
  /* engine numerics: 1 = K/V ring cache stored as bf16 (GPU target, BASELINE.md roofline),
   * 0 = f32 cache (the Candle CPU path's dtype, srv/utils.rs:386-395). */
  int kv_bf16;
  /* how the bf16-weight linear layers (the LM's GEMMs) form their dot products — both orders are restated bit for bit by the
   * oracle (oracle/dsm_oracle.c orc_dot), neither is Candle's (whose gemm order is implementation-defined):
   *   0  v_mfma_f32_16x16x4_f32: one f32 fmaf chain per 256-wide K-chunk (csrc/dsm_numerics.h)
   *   1  "bx3" (r03): the f32 activation split exactly into three bf16 pieces, v_mfma_f32_16x16x32_bf16 on (weight, piece)
   *      per 32-wide block in the order lo, mid, hi; the instruction's internal adder as modelled in
   *      csrc/dsm_bf16_mfma_model.h (validated on the hardware: experiments/bf16_adder_probe.hip).  Every product is exact; the
   *      matrix pipe does a third of the cycles per k ... at 16 x the rate.
   * The presets below (and config_toml's default) select 1, the mode bench.py times; tests/test_dot_mode_agreement_gpu.py bounds
   * its distance from mode 0 at the real dimensions. */
  int dot_mode;
} dsm_asr_config;

/* Presets matching configs/stt/config-stt-en_fr-hf.toml and config-stt-en-hf.toml. */
void dsm_mimi_config_v0_1(dsm_mimi_config* out, int num_codebooks);
void dsm_asr_config_stt_1b_en_fr(dsm_asr_config* out);
void dsm_asr_config_stt_2_6b_en(dsm_asr_config* out);

typedef struct dsm_engine dsm_engine; /* opaque: one per (device, BatchedAsr module) */

/* AsrMsg — core/asr.rs:9-13.  Word tokens are returned through a side buffer. */
enum { DSM_MSG_STEP = 0, DSM_MSG_WORD = 1, DSM_MSG_END_WORD = 2 };
typedef struct dsm_asr_msg {
  int kind;
  int batch_idx;       /* Word / EndWord */
  int step_idx;        /* Step: model_step_idx after the step */
  double time;         /* Word: start_time, EndWord: stop_time (seconds) */
  int tokens_offset;   /* Word: tokens live in tokens_out[offset .. offset+n_tokens) */
  int n_tokens;
  float prs[DSM_MAX_EXTRA_HEADS]; /* unused here; Step probabilities come from vad_prs_out */
} dsm_asr_msg;

typedef struct dsm_metrics {
  double last_encode_us;     /* device time of the last dsm_mimi_encode_step (HIP events) */
  double last_lm_us;         /* device time of the last dsm_asr_step_tokens */
  double algorithmic_bytes_encode; /* BASELINE.md bytes_step() terms, for roofline.achieved */
  double algorithmic_bytes_lm;
  uint64_t steps_encode;
  uint64_t steps_lm;
  uint64_t graph_launches;   /* launch sequences replayed as one hipGraphLaunch (Mimi encode / decode, one LM stream group, one TTS step) */
  uint64_t eager_bodies;     /* the same sequences enqueued launch by launch (warm-up runs, DSM_GRAPHS=0, profiling on) */
  uint64_t capture_failures; /* stream captures that were begun and did not end in a launchable graph (the body then ran launch by
                                launch; the capture is retried up to three times before that sequence stays eager).  Expected: 0 */
  char capture_error[96];    /* what the FIRST such failure reported (hipGetErrorString / the engine's own reason), "" if none */
} dsm_metrics;

/*
 * Create.  Replaces LmModel::batched + Mimi::batched + asr::State::new
 * (srv/batched_asr.rs:738-758, 272-278; core/lm.rs:816; core/mimi.rs:113; core/asr.rs:65).
 * Reads the same safetensors keys as the reference; LM tensors must be
 * BF16 or F32 (stored as bf16 on device), Mimi tensors F32/BF16 (stored as f32,
 * core/mimi.rs:250-255).  Fails with DSM_ERR_DEVICE if no gfx950 device is usable — there
 * is no CPU fallback in this library.
 */
int dsm_asr_create(const dsm_asr_config* cfg, int device_id, int batch_size,
                   const char* lm_safetensors, const char* mimi_safetensors, dsm_engine** out);
void dsm_destroy(dsm_engine*);
const char* dsm_last_error(const dsm_engine*); /* NULL engine -> last create error */

/*
 * Multi-GPU load (the reference is single-device: srv/main.rs:317-327).  Every immutable weight
 * tensor of an engine lives in ONE contiguous device allocation (the "weight arena": packed bf16 LM matrices, f32 Mimi
 * matrices, folded codebooks, norms — 2.0 GB for stt-1b-en_fr), carved in load order.  The rank that read the
 * safetensors exposes it with dsm_asr_weight_arena; the launcher broadcasts those bytes once (RCCL over xGMI: the only
 * collective of the whole path) together with the small manifest (the answers of the loader's optional-key probes, one
 * byte each), and every other rank attaches with dsm_asr_create_from_arena: no file, no conversion, no packing, no
 * host copy.  The arena handed to dsm_asr_create_from_arena stays owned by the caller and must outlive the engine; it
 * must sit on device `device_id`.  Configuration and batch size may differ per rank only in batch_size.
 */
int dsm_asr_weight_arena(dsm_engine*, void** d_arena, size_t* arena_bytes, const uint8_t** manifest, size_t* manifest_bytes);
int dsm_asr_create_from_arena(const dsm_asr_config* cfg, int device_id, int batch_size, const void* d_arena,
                              size_t arena_bytes, const uint8_t* manifest, size_t manifest_bytes, dsm_engine** out);
/* r04 — one process, several devices (the reference is one process: srv/main.rs:317-327): a second engine on `device_id` whose
 * weights are a device-to-device copy of `src`'s arena (hipMemcpyPeerAsync over xGMI in 256 MB pieces; a plain device copy when
 * device_id is src's own device).  The replica owns its copy; `cfg` NULL = src's configuration.  INTEGRATION.md shows the Rust
 * server creating one engine + one worker per GPU this way and routing a new socket to the first worker with a free slot. */
int dsm_asr_create_replica(dsm_engine* src, const dsm_asr_config* cfg, int device_id, int batch_size, dsm_engine** out);

/*
 * Mimi::encode_step(&StreamTensor, &StreamMask) — core/mimi.rs:195-206, called at
 * srv/batched_asr.rs:362 (encoder thread).  pcm: host [B*1920] f32; mask: host [B] u8;
 * codes_out: host [B*n_q] u32 (slot-major, [B, n_q, 1] like the reference) or NULL to keep
 * the codes on the device for a following dsm_asr_step_tokens(codes == NULL).
 * *produced = number of frames emitted (always 1 for 1920-sample frames).
 * Codes of inactive slots are unspecified (the reference computes garbage for them and
 * ignores it: core/asr.rs:177-183).
 */
int dsm_mimi_encode_step(dsm_engine*, const float* pcm, const uint8_t* mask, uint32_t* codes_out,
                         int* produced);

/*
 * Mimi::decode_step(&StreamTensor codes [B,n_q,1], &StreamMask) — core/mimi.rs:217-225 (called by the TTS worker at
 * srv/tts.rs:538): quantizer decode -> ConvTr upsample -> decoder transformer -> SEANet decoder.
 * codes: host [B*n_q] u32 (values < quantizer_bins); pcm_out: host [B*1920] f32.  *produced = samples per slot.
 * The decode state is allocated on first use and belongs to the encoder-side Mimi (dsm_mimi_reset_slot resets its conv
 * carries; like the reference it does NOT reset the decoder transformer: core/mimi.rs:237-238).
 */
int dsm_mimi_decode_step(dsm_engine*, const uint32_t* codes, const uint8_t* mask, float* pcm_out, int* produced);
int dsm_mimi_decode_step_dev(dsm_engine*, const uint32_t* d_codes, const uint8_t* d_mask, float* d_pcm_out);

/*
 * asr::State::step_tokens(&Tensor[B,n_q,1], None, &StreamMask, f) — core/asr.rs:147-255,
 * called at srv/batched_asr.rs:476 (model thread).  codes: host [B*n_q] u32 or NULL (use the
 * device-resident output of the last encode step).  text_tokens_out: host [B] u32 (argmax
 * per slot, valid for active slots).  vad_prs_out: host [extra_heads_num*B] f32 (class-0
 * probability of every extra head, layout [head][slot], core/asr.rs:195-203) or NULL.
 * Word/EndWord/Step messages produced by this step are queued for dsm_asr_poll_msgs.
 */
int dsm_asr_step_tokens(dsm_engine*, const uint32_t* codes, const uint8_t* mask,
                        uint32_t* text_tokens_out, float* vad_prs_out);

/* asr::State::step_pcm — core/asr.rs:115-131 (warmup path, srv/batched_asr.rs:236):
 * encode_step on the MODEL side's own Mimi state followed by step_tokens. */
int dsm_asr_step_pcm(dsm_engine*, const float* pcm, const uint8_t* mask, uint32_t* codes_out,
                     uint32_t* text_tokens_out, float* vad_prs_out);

/* The same two calls for callers that run the reference's thread pair with run-ahead (encoder_loop feeds model_loop through a
 * bounded channel, srv/batched_asr.rs:291,314-522): dsm_mimi_encode_step_async copies the frame, enqueues upload + encode on the
 * encoder stream and returns at once with a ticket (ring of 4 frames: DSM_ERR_STATE while all four wait for the model side);
 * dsm_asr_step_tokens_ticket makes the model stream wait for that frame's codes (kept in a per-ticket device copy), runs the
 * LM step and returns tokens like dsm_asr_step_tokens.  Encode of frame n+1 then overlaps the LM step of frame n on the GPU.
 * Tickets must be consumed in the order they were issued. */
int dsm_mimi_encode_step_async(dsm_engine*, const float* pcm, const uint8_t* mask, int* ticket);
int dsm_asr_step_tokens_ticket(dsm_engine*, int ticket, const uint8_t* mask, uint32_t* text_tokens_out, float* vad_prs_out);

/* Read the Vec<AsrMsg> the last step returned (the list lives until the next step: polling does not consume it).
 * Returns the number of messages written (<= cap; a step yields at most 2*B + 1); word tokens are copied to
 * tokens_out, truncated to tokens_cap entries — a Word whose tokens_offset + n_tokens exceeds tokens_cap was cut:
 * poll again with a larger buffer (dsm_worker_step does exactly that). */
int dsm_asr_poll_msgs(dsm_engine*, dsm_asr_msg* msgs, int cap, uint32_t* tokens_out, int tokens_cap);

/* asr::State::reset_batch_idx(slot) — core/asr.rs:257-266, srv/batched_asr.rs:467-471.
 * Resets the LM slot (KV index/position, item state) and the MODEL side's Mimi slot.
 * The reference never resets the ENCODER thread's Mimi clone;
 * dsm_mimi_reset_slot exposes that missing reset explicitly instead of hiding a
 * behaviour change inside dsm_asr_reset_slot. */
int dsm_asr_reset_slot(dsm_engine*, int slot);
int dsm_mimi_reset_slot(dsm_engine*, int slot);
/* temperature > 0 only: (re)start the slot's Gumbel-noise stream at word 0 of ChaCha12 keyed by rand's seed_from_u64(seed)
 * (slot b starts with seed 0x5EED0000 + b).  A reset of the slot does not touch the stream, like the reference's global generator. */
int dsm_asr_set_seed(dsm_engine*, int slot, uint64_t seed);

/* Device::synchronize — srv/batched_asr.rs:238. */
int dsm_sync(dsm_engine*);

int dsm_get_metrics(dsm_engine*, dsm_metrics* out);
int dsm_batch_size(const dsm_engine*);
int dsm_n_q(const dsm_engine*); /* mimi.config().quantizer_n_q — srv/batched_asr.rs:379 */

/*
 * Device-resident variants for callers that already hold HBM buffers (bench.py's timed
 * region; a Rust caller with hipMalloc'd staging).  Pointers are DEVICE pointers, nothing
 * is synchronised; work is enqueued on the engine's encoder / model / group streams, which are
 * NON-BLOCKING: they are not ordered against the null stream or the caller's streams.  The caller
 * makes its inputs visible before the call (stream/device synchronise after filling them) and
 * reads outputs only after dsm_sync; inputs must stay unmodified until then.
 */
int dsm_mimi_encode_step_dev(dsm_engine*, const float* d_pcm, const uint8_t* d_mask, uint32_t* d_codes_out);
int dsm_asr_step_tokens_dev(dsm_engine*, const uint32_t* d_codes, const uint8_t* d_mask,
                            uint32_t* d_text_tokens_out, float* d_vad_prs_out);
/* Make the model stream wait for everything enqueued so far on the encoder stream. */
int dsm_streams_join(dsm_engine*);
/* step_pcm with device pointers: Mimi encode (model-side state) + LM step, all enqueued on the
 * model stream, nothing synchronised.  d_codes_out / d_text_tokens_out / d_vad_prs_out may be NULL. */
int dsm_asr_step_pcm_dev(dsm_engine*, const float* d_pcm, const uint8_t* d_mask, uint32_t* d_codes_out,
                         uint32_t* d_text_tokens_out, float* d_vad_prs_out);

/* Per-kernel-class device timing with HIP events recorded on the stream each kernel is launched on
 * (feeds bench.py's roofline object).  tag_mask selects classes; dsm_prof_read synchronises, adds
 * up the elapsed time of every bracketed launch since the last read, and resets. */
enum { DSM_PROF_ATTN_LM = 0, DSM_PROF_GEMM_LM = 1, DSM_PROF_ATTN_MIMI = 2, DSM_PROF_GEMM_MIMI = 3,
       DSM_PROF_RVQ = 4, DSM_PROF_OTHER = 5, DSM_PROF_NTAGS = 6 };
int dsm_prof_enable(dsm_engine*, unsigned tag_mask);
int dsm_prof_read(dsm_engine*, double* total_us /*[DSM_PROF_NTAGS]*/, uint64_t* launches /*[DSM_PROF_NTAGS]*/);
/* Same classes, but bracketed INSIDE the kernel with the device wall clock (first workgroup in, last workgroup out):
 * the duration rocprofv3's kernel trace reports.  With several streams in flight a HIP-event bracket also counts the
 * time a launch queues behind other streams' kernels.  Instrumented classes: ATTN_LM, ATTN_MIMI (others read 0). */
int dsm_prof_read_device(dsm_engine*, double* total_us /*[DSM_PROF_NTAGS]*/, uint64_t* launches /*[DSM_PROF_NTAGS]*/);
/* Timeline diagnostics (tools/timeline.py): with `on`, the tiled GEMM launches and their reduce launches are bracketed in the
 * kernel like the attention launches; dsm_prof_timeline_read returns every bracketed launch since the last read as (stream id:
 * 0 encoder, 1 + g LM group g; kind: 0 attention, 1 GEMM, 2 reduce; class tag; start / end in us from the earliest start). */
int dsm_prof_timeline(dsm_engine*, int on);
int dsm_prof_timeline_read(dsm_engine*, int* sid, int* kind, int* tag, double* start_us, double* end_us, int cap);

/* ------------------------------------------------------------------------------------------------
 * TTS (BASELINE.json configs[4]): tts_streaming::State::step + LmModel::forward_cond + DepFormer::sample
 * (core/tts_streaming.rs:117-242, core/lm.rs:957-1008, :640-684); Sampling::ArgMax by default (what the reference
 * selects when temperature <= 0, srv/tts.rs:402), seeded top-k per slot with dsm_tts_set_sampling.  The reference serves one generation at a
 * time behind a mutex (srv/tts.rs:374); here B independent generations advance together, each slot with its
 * own step index (batching the TTS path is new capability).
 * ------------------------------------------------------------------------------------------------ */
#define DSM_TTS_UNGENERATED 0xFFFFFFFFu /* tts_streaming::UNGENERATED */
enum { DSM_TTS_ALLOW_PAD = -1, DSM_TTS_ALLOW_PAD_OR_EPAD = -2 }; /* AllowedTokens::{Pad, PadOrEpad}; >= 0: Text(v) */

typedef struct dsm_tts_config {
  dsm_transformer_config lm;        /* [modules.tts.model.transformer] */
  int text_in_vocab_size, text_out_vocab_size, audio_vocab_size, audio_codebooks;
  dsm_transformer_config depformer; /* [modules.tts.model.depformer.transformer]; positional_embedding = None */
  int dep_num_slices;               /* DepFormerConfig::num_slices = generated audio codebooks */
  int dep_low_rank;                 /* low_rank_embeddings (0 = full-rank tables) */
  int dep_weight_groups;            /* distinct linear_in / gating weight sets: idx*groups/num_slices (core/lm.rs:527,558) */
  /* tts_streaming::Config — core/tts_streaming.rs:12-44 */
  int acoustic_delay, text_pad_token, text_bos_token, text_eos_token, text_eop_token, text_start_token;
  int text_audio_delay_in_tokens, max_consecutive_pads, max_steps;
  int kv_bf16;
  /* The branch the reference's TTS server runs (srv/tts.rs:426-441: State::new(.., Some(CaSrc::Tokens(ca_src)), .., cfg_alpha, ..)):
   * cross-attention to a speaker-conditioning source in every main-LM layer (core/lm.rs:392-396 tts_202501:
   * cross_attention = Some((Normal, LayerNorm, None)); core/transformer.rs:205-330,747-763) and classifier-free guidance
   * (core/tts_streaming.rs:164-173,207-214).  All zero = the `ca_src = None`, `cfg_alpha = None` path. */
  int cross_attention;   /* 1: every main-LM layer carries norm_cross + cross_attention.{in_proj_weight_q, in_proj_weight_kv, out_proj}
                            (or the legacy single in_proj_weight); gating must be CrossAttentionGating::Normal */
  int ca_norm;           /* norm_cross: 0 LayerNorm (eps 1e-5, weight|alpha + bias), 1 RmsNorm (eps 1e-8, alpha) — cfg.cross_attention.1 */
  int ca_dim;            /* width of a cross-attention source row (kv_in_dim, cfg.cross_attention.2); 0 = d_model */
  int ca_max_len;        /* longest source (rows of ca_src) a slot may be given with dsm_tts_set_ca_src */
  int cfg_rows;          /* 1: every slot owns TWO batch rows (conditional, unconditional) so that it can run with a cfg_alpha */
  int dot_mode;          /* as dsm_asr_config.dot_mode */
} dsm_tts_config;
void dsm_tts_config_v202501(dsm_tts_config* out); /* Config::v202501 + configs/tts/config-tts.toml with a consistent depformer */

typedef struct dsm_tts dsm_tts;
int dsm_tts_create(const dsm_tts_config* cfg, int device_id, int batch_size, const char* lm_safetensors, dsm_tts** out);
void dsm_tts_destroy(dsm_tts*);
const char* dsm_tts_last_error(const dsm_tts*);
/* State::step for every active slot.  prev_text_token [B]; allowed [B] (>= 0 Text(v), DSM_TTS_ALLOW_*);
 * text_token_out [B]; audio_out [B*num_slices]: this step's last_audio_tokens (depformer samples) or
 * DSM_TTS_UNGENERATED while step_idx < text_audio_delay_in_tokens.  */
int dsm_tts_step(dsm_tts*, const uint32_t* prev_text_token, const int32_t* allowed, const uint8_t* mask,
                 uint32_t* text_token_out, uint32_t* audio_out);
/* audio_tokens[step] of a slot (delayed write-back applied, core/tts_streaming.rs:220-236); entries may be UNGENERATED */
int dsm_tts_audio_tokens(dsm_tts*, int slot, int step, uint32_t* out /* [num_slices] */);
int dsm_tts_step_idx(dsm_tts*, int slot);
int dsm_tts_reset_slot(dsm_tts*, int slot);
/* The slot's text and audio LogitsProcessor (srv/tts.rs:401-415): Sampling::TopK{k, temperature} seeded with `seed` when
 * temperature > 0 and top_k > 1, Sampling::ArgMax otherwise (the default after create and after dsm_tts_reset_slot).
 * Softmax over the whole vocabulary at the temperature, the k most probable tokens (probability descending, token id
 * ascending), one weighted draw from a ChaCha12 stream (rand 0.9 StdRng, seed_from_u64): csrc/dsm_sampling.h.  The
 * reference's order among the k survivors is an implementation detail of Rust's select_nth_unstable_by and no vector of
 * candle's / rand's exists offline: sampled tokens are pinned engine-vs-oracle only ("parity unpinned" vs Candle). */
int dsm_tts_set_sampling(dsm_tts*, int slot, int top_k, float temperature, uint64_t seed);
/* The slot's cross-attention source and guidance — the `ca_src` / `cfg_alpha` arguments of tts_streaming::State::new
 * (core/tts_streaming.rs:70-100; built at srv/tts.rs:426-441 from the voice's `ca_src` tensor, with the speaker encoder's
 * empty conditioning appended as a second batch row when the query carries a cfg_alpha).
 *   ca_src [n][ca_dim] f32, host: CaSrc::Tokens of batch row 0.  Its keys and values (in_proj_kv, core/transformer.rs:299-318)
 *     are projected ONCE here into a per-slot device buffer; the reference recomputes the same values every step.
 *     NULL / n = 0: ca_src = None (cross-attention is skipped for the slot: core/transformer.rs:753-760).
 *   ca_src_uncond [n_uncond][ca_dim] or NULL: batch row 1 (needs cfg_rows = 1).  With it the slot runs both rows and mixes
 *     text logits and every depformer slice's logits as l0 * a - l1 * (a - 1), a = cfg_alpha (core/tts_streaming.rs:166-172,
 *     core/lm.rs:718-721: Tensor * f64 = affine(mul as f32, 0)), sampling once per slot.
 * Needs cfg.cross_attention; n, n_uncond <= ca_max_len.  dsm_tts_reset_slot clears both (a fresh State). */
int dsm_tts_set_ca_src(dsm_tts*, int slot, const float* ca_src, int n, const float* ca_src_uncond, int n_uncond, double cfg_alpha);
int dsm_tts_debug_read(dsm_tts*, const char* name, float* out, size_t cap); /* "lm.hidden", "lm.logits": one row per BATCH ROW (2 per slot with cfg_rows) */
int dsm_tts_get_metrics(dsm_tts*, dsm_metrics* out); /* graph_launches / eager_bodies only */

/* The LM step splits the batch into stream groups (slots [first, first+n) each on its own HIP stream) so that one
 * group's HBM-bound attention overlaps another's MFMA-bound GEMMs; results do not depend on the split.  Returns the
 * number of groups and fills up to `cap` entries.  Environment DSM_LM_GROUPS (1..4) overrides the default at create. */
int dsm_lm_stream_groups(dsm_engine*, int* first_slot, int* n_slots, int cap);
/* Profiling aid: run every group on the model stream, one after the other (per-kernel timings without overlap). */
int dsm_debug_serialize_groups(dsm_engine*, int on);

/* Profiling aid: pretend every slot already streamed `pos` frames (ring index = pos mod ctx, cache content
 * untouched) so that short profiler runs see steady-state attention traffic.  Never used for `value`. */
int dsm_debug_set_positions(dsm_engine*, uint32_t lm_pos, uint32_t mimi_pos);

/* Test aid (teacher forcing): overwrite the text token every slot feeds back into its next step (core/asr.rs:147-160, the
 * `text_token` of State) with `tokens` [batch].  tests/test_dot_mode_agreement_gpu.py drives a dot_mode 1 engine along a
 * dot_mode 0 engine's token history with it, so that one flipped argmax does not cascade. */
int dsm_debug_set_text_tokens(dsm_engine*, const uint32_t* tokens);

/* Debug taps for the parity tests: copy a named intermediate of the last step to the host.
 * Names: "lm.hidden" [B,d], "lm.logits" [B,V], "mimi.seanet_out" [B,T,dim], "mimi.latent" [B,dim] ...
 * Returns the number of floats written, or <0. */
int dsm_debug_read(dsm_engine*, const char* name, float* out, size_t cap);

/* ------------------------------------------------------------------------------------------------
 * Audio ingest helpers (host only; SURVEY.md §8(f) rank 3, first part).
 * dsm_wav_decode: channel 0 of a RIFF/WAVE body as f32 — what srv/utils.rs:263-305 `pcm_decode` hands the worker
 * (srv/batched_asr.rs:834-842) for such a body.  *pcm_out is malloc'd; release it with dsm_free.
 * dsm_linear_resampler_*: the clients' default resampler, client/rust/kyutai-client-core/src/audio.rs:133-183
 * (`LinearResampler::process_into`), streaming.  The Opus codec is the host's callback; the Ogg container and mp3 are built (below).
 * ------------------------------------------------------------------------------------------------ */
int dsm_wav_decode(const uint8_t* bytes, size_t len, float** pcm_out, size_t* n_out, int* sample_rate_out);

/* r04 — the rest of srv/utils.rs:263-305 `pcm_decode` for the bodies the reference's own samples use (audio/bria.mp3 of
 * BASELINE.json configs[0], loona.mp3, sample_fr_hibiki_*.mp3) and srv/batched_asr.rs:834-842 `kaudio::resample(&pcm, rate, 24000)`:
 *   dsm_mp3_decode   MPEG-1 Audio Layer III (ISO/IEC 11172-3; mono, stereo, joint stereo; bit reservoir; ID3v2 / Xing / ID3v1
 *                    skipped as symphonia's reader does), channel 0 as f32 in [-1, 1], no gapless trimming (FormatOptions::default()).
 *                    symphonia is an un-vendored crate and no decoder is importable offline: PARITY UNPINNED against it; the
 *                    tables and the filter bank are pinned by their defining properties (tests/test_mp3.py).  MPEG-2 / 2.5 and
 *                    Layers I / II: DSM_ERR_IO.
 *   dsm_mp3_probe    the frame walk alone (headers + side information): counts, bit rate, sync losses.
 *   dsm_resample     whole-buffer polyphase windowed-sinc resampler for a rational ratio (Kaiser, 32 zero crossings, cut-off
 *                    0.95 of the lower Nyquist) in the place of kaudio::resample (un-vendored; parity unpinned).
 *   dsm_pcm_decode   RIFF/WAVE or mp3 by magic — the call the worker front end makes on a request body.
 * Outputs are malloc'd (dsm_free). */
typedef struct dsm_mp3_info {
  int sample_rate, channels, bitrate_kbps /* first frame */, vbr;
  int frames;       /* audio frames (1152 samples each); a leading Xing / Info / VBRI frame is metadata and not counted */
  int info_frames, resyncs, huffman_overruns, frames_without_reservoir;
  uint64_t id3v2_bytes, junk_bytes; /* junk: bytes skipped between frames after the first one (sync losses) */
} dsm_mp3_info;
int dsm_mp3_decode(const uint8_t* bytes, size_t len, float** pcm_out, size_t* n_out, int* sample_rate_out);
int dsm_mp3_decode_info(const uint8_t* bytes, size_t len, float** pcm_out, size_t* n_out, dsm_mp3_info* info);
int dsm_mp3_probe(const uint8_t* bytes, size_t len, dsm_mp3_info* out);
int dsm_resample(const float* in, size_t n, int rate_in, int rate_out, float** out, size_t* n_out);
int dsm_pcm_decode(const uint8_t* bytes, size_t len, float** pcm_out, size_t* n_out, int* sample_rate_out);
/* test aids (tests/test_mp3.py): the synthesis bank, the Huffman tables and the IMDCT on their own */
int dsm_mp3_test_synth(const float* subbands, int slots, float* pcm);
int dsm_mp3_test_tables(int table, int* xlen, int* linbits, uint16_t* codes, uint8_t* lens, int cap);
int dsm_mp3_test_imdct(const float* spectrum, int block_type, float* out);
void dsm_free(void*);
/* Ogg container demultiplexer (RFC 3533 pages -> RFC 7845 Opus packets) for InMsg::OggOpus bodies, the container half of
 * kaudio::ogg_opus::Decoder (srv/batched_asr.rs:894,941-949): streaming (bytes may be split anywhere), CRC-checked,
 * resynchronises after damage, drops packets whose beginning or middle was lost.  No Opus codec is built here: the
 * packets go to the decoder the host already has (dsm_worker_set_opus_decoder, or the caller's own loop). */
typedef struct dsm_ogg_demux dsm_ogg_demux;
dsm_ogg_demux* dsm_ogg_demux_new(void);
void dsm_ogg_demux_free(dsm_ogg_demux*);
int dsm_ogg_demux_push(dsm_ogg_demux*, const uint8_t* bytes, size_t len); /* -> packets waiting, or <0 */
int dsm_ogg_demux_next(dsm_ogg_demux*, const uint8_t** packet, size_t* len, int* is_header); /* 1 / 0; pointer valid until the next call */
int dsm_ogg_demux_info(const dsm_ogg_demux*, int* channels, int* pre_skip, uint32_t* input_rate, uint64_t* pages_ok, uint64_t* pages_bad);
/* Ogg multiplexer: the container half of kaudio::ogg_opus::Encoder on the TTS output side (srv/tts.rs:75-76,188-260):
 * dsm_ogg_mux_header = `header_data()` (OpusHead page + OpusTags page), dsm_ogg_mux_page = the page `encode_page(pcm)`
 * emits for one 80 ms frame, from the Opus packets of the host's encoder.  Both return the page bytes' count and write
 * them when they fit in cap (a page that does not fit is not consumed: call again with a larger buffer). */
typedef struct dsm_ogg_mux dsm_ogg_mux;
dsm_ogg_mux* dsm_ogg_mux_new(uint32_t serial, int channels, uint32_t input_sample_rate, int pre_skip);
void dsm_ogg_mux_free(dsm_ogg_mux*);
int dsm_ogg_mux_header(dsm_ogg_mux*, uint8_t* out, size_t cap);
int dsm_ogg_mux_page(dsm_ogg_mux*, const uint8_t* const* packets, const size_t* lens, int n_packets, uint64_t samples_48k,
                     int end_of_stream, uint8_t* out, size_t cap);
typedef struct dsm_resampler dsm_resampler;
dsm_resampler* dsm_linear_resampler_new(uint32_t in_rate_hz, uint32_t out_rate_hz);
size_t dsm_linear_resampler_process(dsm_resampler*, const float* in, size_t n_in, float* out, size_t out_cap);
void dsm_linear_resampler_free(dsm_resampler*);

/* ------------------------------------------------------------------------------------------------
 * Host worker above the engine: moshi-server's BatchedAsr module minus the sockets.
 * Wire format: rmp_serde "struct map" encoding of the internally tagged enums InMsg / OutMsg (srv/asr.rs:15-34):
 * a msgpack map whose first entry is "type": <variant>, followed by the variant's fields in declaration order;
 * usize / i64 in their smallest msgpack integer form, f64 as 0xcb, f32 as 0xca (srv/batched_asr.rs:877-880,969-975;
 * client/rust/kyutai-client/src/stt/protocol.rs:47-62, whose test vector at :82-97 pins the encoding).
 * ------------------------------------------------------------------------------------------------ */
enum { DSM_IN_INIT = 0, DSM_IN_MARKER = 1, DSM_IN_AUDIO = 2, DSM_IN_OGGOPUS = 3, DSM_IN_PING = 4 };
typedef struct dsm_in_msg {
  int kind;
  int64_t id;          /* Marker */
  const float* pcm;    /* Audio */
  size_t n_pcm;
  const uint8_t* data; /* OggOpus */
  size_t n_data;
} dsm_in_msg;
enum { DSM_OUT_WORD = 0, DSM_OUT_ENDWORD = 1, DSM_OUT_MARKER = 2, DSM_OUT_STEP = 3, DSM_OUT_ERROR = 4, DSM_OUT_READY = 5 };
typedef struct dsm_out_msg {
  int kind;
  const char* text;      /* Word.text / Error.message (UTF-8, NUL terminated) */
  double time;           /* Word.start_time / EndWord.stop_time */
  int64_t id;            /* Marker */
  uint64_t step_idx;     /* Step */
  const float* prs;
  size_t n_prs;
  uint64_t buffered_pcm;
} dsm_out_msg;
/* Encoders return the encoded size (and write it when it fits in cap); decoders return 0 or DSM_ERR_IO. */
int dsm_inmsg_encode(const dsm_in_msg*, uint8_t* buf, size_t cap);   /* client: protocol.rs encode_in_msg */
int dsm_outmsg_encode(const dsm_out_msg*, uint8_t* buf, size_t cap); /* server: send_loop serialisation */
int dsm_inmsg_decode(const uint8_t* bytes, size_t len, dsm_in_msg* out, float* pcm_buf, size_t pcm_cap,
                     uint8_t* data_buf, size_t data_cap);            /* server: rmp_serde::from_slice::<InMsg> */
int dsm_outmsg_decode(const uint8_t* bytes, size_t len, dsm_out_msg* out, char* text_buf, size_t text_cap,
                      float* prs_buf, size_t prs_cap);               /* client: protocol.rs decode_out_msg */

/* text_tokenizer.decode_piece_ids (srv/batched_asr.rs:667): write the UTF-8 text of the pieces, return its length or <0.
 * Without one a Word carries the decimal piece ids separated by spaces. */
typedef int (*dsm_detok_fn)(void* user, const uint32_t* tokens, int n_tokens, char* out, size_t cap);

typedef struct dsm_worker dsm_worker;
int dsm_worker_create(dsm_engine*, dsm_worker** out);  /* slot table of batch_size channels (srv/batched_asr.rs:762-790) */
/* The four moshi-core calls the worker makes (encode_step :362, reset_batch_idx :468, step_tokens :476 + its
 * Vec<AsrMsg>), as a table, so that the worker logic can be driven by something other than the HIP engine (the
 * test-suite plugs the CPU oracle in here; the product path is dsm_worker_create). */
typedef struct dsm_worker_backend {
  void* self;
  int batch_size, asr_delay_in_tokens, extra_heads_num;
  int (*encode_step)(void* self, const float* pcm /*[B*1920]*/, const uint8_t* mask /*[B]*/); /* codes stay inside */
  int (*reset_slot)(void* self, int slot);
  int (*step_tokens)(void* self, const uint8_t* mask, uint32_t* text_tokens_out /*[B]*/, float* prs_out /*[heads*B]*/);
  int (*poll_msgs)(void* self, dsm_asr_msg* msgs, int cap, uint32_t* tokens_out, int tokens_cap);
  const char* (*last_error)(void* self); /* may be NULL */
  /* optional pair (both or neither): the run-ahead forms — encode returns at once with a ticket naming the frame's codes,
   * the model side consumes tickets in order (dsm_mimi_encode_step_async / dsm_asr_step_tokens_ticket).  With them the
   * worker lets dsm_worker_step_encode run up to three frames ahead of dsm_worker_step_model; without them one. */
  int (*encode_async)(void* self, const float* pcm, const uint8_t* mask, int* ticket);
  int (*step_ticket)(void* self, int ticket, const uint8_t* mask, uint32_t* text_tokens_out, float* prs_out);
} dsm_worker_backend;
int dsm_worker_create_with_backend(const dsm_worker_backend*, dsm_worker** out);
void dsm_worker_destroy(dsm_worker*);
const char* dsm_worker_last_error(const dsm_worker*);
void dsm_worker_set_detokenizer(dsm_worker*, dsm_detok_fn, void* user);
/* BatchedAsr::channels (:796-808) + the Init that handle_socket / handle_query send (:833,893).  Returns the slot
 * (batch_idx) or DSM_ERR_STATE when no slot is free ("Server at capacity - no free channels available", :875). */
int dsm_worker_open(dsm_worker*, uint64_t* channel_id);
int dsm_worker_close(dsm_worker*, int slot);           /* the socket went away: the slot is recycled by the next step */
/* One binary websocket message from the client (recv_loop, :927-951): 0 queued, 1 undecodable and skipped (malformed,
 * larger than 64 MiB, or nested deeper than rmp_serde's limit of 1024: the reference logs and carries on),
 * DSM_ERR_STATE closed channel / OggOpus (no Opus decoder in this build). */
int dsm_worker_send(dsm_worker*, int slot, const uint8_t* msgpack, size_t len);
/* handle_query (srv/batched_asr.rs:811-851): a whole audio FILE (RIFF/WAVE or MPEG-1 Layer III) as the request on a freshly opened channel
 * (dsm_worker_open queued the Init) — queues the decoded clip resampled to 24 kHz, Marker { id: 0 } and ten seconds of silence; the transcript is complete when the Marker returns. */
int dsm_worker_send_body(dsm_worker*, int slot, const uint8_t* body, size_t len);
/* One pass of encoder_loop -> model_loop -> post_process (:314-522) on the calling thread: 1 a step ran, 0 idle, <0 engine error. */
int dsm_worker_step(dsm_worker*);
/* The same work as the reference's threads split it (:314 encoder_loop, :432 model_loop + :414 post_process), for a host
 * that runs them on two threads: dsm_worker_step_encode cuts the next frame from the sockets' queues, starts its Mimi encode and
 * queues a PipelineMsg (1 produced, 0 idle or queue full); dsm_worker_step_model takes the oldest one through slot resets,
 * step_tokens and the message fan-out (1 ran, 0 nothing queued).  With the engine as backend the encode of frame n+1 overlaps
 * the LM step of frame n on the GPU; every client receives the same messages in the same order as with dsm_worker_step. */
int dsm_worker_step_encode(dsm_worker*);
int dsm_worker_step_model(dsm_worker*);
/* Next serialised OutMsg for the slot's socket (send_loop, :960-985): 1 written, 0 none; *len = its size. */
int dsm_worker_recv(dsm_worker*, int slot, uint8_t* buf, size_t cap, size_t* len);
int dsm_worker_buffered(dsm_worker*, int slot);        /* samples waiting in the channel's queue (Step.buffered_pcm) */
/* The codec half of kaudio::ogg_opus::Decoder::new(24000, FRAME_SIZE) (srv/batched_asr.rs:894): decode ONE Opus packet of
 * the slot's stream to 24 kHz mono f32 (libopus on the host side: opus_decode_float at 24 kHz), return the sample count or
 * <0.  With a decoder set, InMsg::OggOpus bodies are demultiplexed per channel (dsm_ogg_demux) and every audio packet is
 * decoded and queued like InMsg::Audio (:941-947); a decode error is reported and skipped like the reference logs it.
 * Without one, dsm_worker_send keeps refusing OggOpus messages (DSM_ERR_STATE). */
typedef int (*dsm_opus_decode_fn)(void* user, int slot, const uint8_t* packet, size_t len, float* pcm_out, size_t cap);
/* Threading contract of the decoder callback: it is called from the thread that called dsm_worker_send, OUTSIDE
 * the worker lock and under the CHANNEL's own lock only — concurrently for different slots, never concurrently for one slot;
 * `slot` tells the callback which per-socket decoder state to use.  It must not call back into dsm_worker_* for the same slot. */
void dsm_worker_set_opus_decoder(dsm_worker*, dsm_opus_decode_fn, void* user);

#ifdef __cplusplus
}
#endif
#endif /* DSM_H */
