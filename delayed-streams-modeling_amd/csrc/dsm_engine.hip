// dsm_engine.hip — libdsm_mi355x.so: the engine behind include/dsm.h.
//
// Host orchestration of the per-80 ms-frame batched step on one MI355X:
//   encoder side  (HIP stream "enc")   Mimi::encode_step      core/mimi.rs:195-206
//   model side    (HIP stream "model") asr::State::step_tokens core/asr.rs:147-255
// All per-slot state (conv carries, ring KV caches, ring index/position, item state) lives in
// HBM and is advanced by the kernels themselves; the only per-step host<->device traffic is the
// PCM/mask upload and the token/VAD download, like the reference (SURVEY.md §2.1 last paragraph).
// There is no CPU fallback: creation fails if no HIP device is usable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <set>
#include <vector>

#include "../../include/dsm.h"
#include "dsm_config_presets.h"
#include "dsm_kernels.h"
#include "dsm_numerics.h"
#include "dsm_safetensors.h"

static thread_local std::string g_create_error;

#define HIPCHK_E(eng, expr)                                                                         \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      (eng)->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
      return DSM_ERR_DEVICE;                                                                        \
    }                                                                                               \
  } while (0)
#define HIPCHK(expr) HIPCHK_E(e, expr)

namespace {

struct Linear {  // packed weight [Npad][Kpad] (+ optional bias) on the device
  void* w = nullptr;
  float* bias = nullptr;
  int N = 0, K = 0, Npad = 0, Kpad = 0;
  bool bf16 = false;
  bool packed = false;  // bf16 only (r04): fragment-major [Npad / 16][Kpad / 32][64 lanes][8] instead of row-major (dsm_wbase)
};

struct ConvGeom {
  Linear lin;
  int in_c = 0, out_c = 0, k = 0, stride = 1;
  int S = 0;     // carried frames = k_eff - stride (dilation 1)
  int T_in = 0;  // input frames per step
  int T_out = 0;
  bool replicate = false;
};

struct TLayerW {
  Linear in_proj, out_proj, ff_in, ff_out;
  float *n1w = nullptr, *n1b = nullptr, *n2w = nullptr, *n2b = nullptr, *ls1 = nullptr, *ls2 = nullptr;
  // cross attention (TTS main LM, core/transformer.rs:205-330,747-763): in_proj_q, in_proj_kv, out_proj, norm_cross
  Linear ca_q, ca_kv, ca_out;
  float *ncw = nullptr, *ncb = nullptr;
};

struct TransformerW {
  dsm_transformer_config cfg{};
  int hidden = 0;
  std::vector<TLayerW> layers;
  float* inv_freq = nullptr;
  bool rope_pos_before = false;  // non-batched transformer semantics (TTS main LM)
  bool has_ca = false;           // every layer carries norm_cross + cross_attention
  int ca_norm_rms = 0;           // norm_cross: RmsNorm (eps 1e-8) instead of LayerNorm (eps 1e-5)
};

// Per-row cross-attention sources of a transformer_forward call: the projected keys / values of every batch row's ca_src
// (compute_kv, core/transformer.rs:299-318) laid out like a ring cache, [rows][H][smax][hd] per layer, so that attn_kernel
// serves them: `last` = source length - 1 plays start_pos (every one of the `len` rows is visible, no causal mask), `act`
// = the row attends this step (it is active and has a source).  `att` [rows][d] holds the attention output; rows that never
// attend keep the zeros they were created with, so their out_proj term is +0 (ca_src = None: the layer adds nothing, :753-760).
struct CaState {
  std::vector<void*> k, v;
  uint32_t* last = nullptr;
  uint8_t* act = nullptr;
  float* att = nullptr;
  int smax = 0;
};

struct TransformerState {  // per (side): ring caches + ScatteredCacheBuilder state
  std::vector<void*> k, v;  // per layer [B][H][ctx][hd]
  uint32_t *pos = nullptr, *idx = nullptr, *start_pos = nullptr, *widx = nullptr;
  float* rope_cs = nullptr;
};

struct RvqW {
  Linear input_proj, output_proj;
  int n_q = 0;
  std::vector<Linear> codebooks;  // W = embedding [bins][dim] f32, bias = c2 [bins]
};

struct MimiW {
  dsm_mimi_config cfg{};
  ConvGeom init_conv, final_conv, downsample;
  struct Stage {
    ConvGeom ra, rb, down;
  };
  std::vector<Stage> stages;
  TransformerW tr;
  RvqW rvq_first, rvq_rest;
  // decode side (core/mimi.rs:99-103, core/seanet.rs:305-468)
  struct DecStage {
    Linear up;       // convtr as a GEMM: rows kk*out_c + co, K = in_c
    float* up_bias = nullptr;
    int in_c = 0, out_c = 0, k = 0, stride = 0, T_in = 0;
    ConvGeom ra, rb;
  };
  bool has_decoder = false;
  ConvGeom dec_init, dec_final;
  std::vector<DecStage> dec_stages;
  TransformerW dec_tr;
  float* upsample_w = nullptr;  // [k][dim]
  const float** emb_ptrs = nullptr;  // device array of n_q codebook pointers
};

struct MimiState {  // one per side (encoder-thread clone / model side)
  float* pcm = nullptr;  // == cat_init + S0 (the PCM frame is uploaded straight into the concat buffer)
  float* cat_init = nullptr;
  struct Stage {
    float *y = nullptr, *cat_ra = nullptr, *cat_rb = nullptr, *cat_down = nullptr;
  };
  std::vector<Stage> stages;
  float* cat_final = nullptr;
  float *x_tr = nullptr, *xn = nullptr, *q = nullptr, *att = nullptr, *ff = nullptr;
  float* cat_ds = nullptr;
  float* latent = nullptr;
  float *res_first = nullptr, *res_rest = nullptr, *pval = nullptr;
  uint32_t* pidx = nullptr;
  uint32_t* codes = nullptr;  // [B][n_q]
  TransformerState tr;
  ConvStateDesc* descs = nullptr;  // device table for the state-shift kernel
  std::vector<ConvStateDesc> h_descs;
  int ds_desc = -1;  // index of the downsample conv in descs
  bool first_call = true;
  uint8_t* mask = nullptr;  // device [B]
};

struct MimiDecState {  // Mimi::decode_step state, allocated on first use
  bool ready = false, first_call = true;
  uint32_t* codes = nullptr;
  uint8_t* mask = nullptr;
  float *q_first = nullptr, *q_rest = nullptr, *emb = nullptr, *up_carry = nullptr;
  float *x_tr = nullptr, *xn = nullptr, *q = nullptr, *att = nullptr, *ff = nullptr;
  TransformerState tr;
  float* cat_init = nullptr;
  struct Stage {
    float *x = nullptr, *z = nullptr, *carry = nullptr, *y = nullptr, *cat_ra = nullptr, *cat_rb = nullptr;
  };
  std::vector<Stage> stages;
  float* cat_final = nullptr;
  float* pcm = nullptr;
  ConvStateDesc* descs = nullptr;
  std::vector<ConvStateDesc> h_descs;
};

struct LmW {
  uint16_t* text_emb = nullptr;
  uint16_t* audio_emb = nullptr;
  TransformerW tr;
  float* out_norm = nullptr;
  Linear text_linear, extra_heads;
};

struct LmState {
  float *x = nullptr, *xn = nullptr, *q = nullptr, *att = nullptr, *g = nullptr, *hidden = nullptr, *logits = nullptr,
        *eh = nullptr, *prs = nullptr;
  uint32_t *next_cb = nullptr, *text_token = nullptr, *text_out = nullptr, *codes_in = nullptr;
  uint8_t *first_step = nullptr, *mask = nullptr;
  uint32_t* rng_key = nullptr;             // [B][8] ChaCha12 key per slot (temperature > 0: lm_gumbel_kernel; dsm_asr_set_seed)
  unsigned long long* rng_pos = nullptr;   // [B] words drawn so far (multiples of 16)
  uint8_t* gmask = nullptr;  // per-group private copy of the step's mask (groups free-run against each other)
  TransformerState tr;
  // stream groups: slots [b0, b0+nb) of group g step on their own HIP stream so that one group's HBM-bound attention
  // overlaps another's MFMA-bound GEMMs; rows are independent, so the split changes no result
  struct Group {
    int b0 = 0, nb = 0;
    TransformerState view;  // tr with every per-slot pointer advanced to b0
  };
  std::vector<Group> groups;
};

struct HostItem {  // ItemState — core/asr.rs:15-51 (word assembly stays on the host)
  size_t step_idx = 0;
  std::vector<uint32_t> word_tokens;
  bool unended_word = false;
  double last_stop_time = 0.0;
};

}  // namespace

struct dsm_engine {
  dsm_asr_config cfg{};
  int B = 0, device = 0;
  hipStream_t s_enc = nullptr, s_model = nullptr;
  static constexpr int kMaxGroups = 4;
  hipStream_t s_grp[kMaxGroups] = {nullptr, nullptr, nullptr, nullptr};  // s_grp[0] is unused (group 0 runs on s_model)
  hipEvent_t ev_fork = nullptr, ev_grp_in[kMaxGroups] = {}, ev_grp_done[kMaxGroups] = {}, ev_stagger[kMaxGroups] = {};
  bool stagger = true;  // DSM_STAGGER=0: every group starts its step at once (r01); 2: staggered at every batch size
  bool stagger_force = false;
  bool grp_busy = false;
  int enc_cus = 0;            // DSM_ENC_CUS: CUs per XCD for the encoder stream (0: no CU mask)
  bool lm_cus_excl = false;   // DSM_LM_CUS_EXCL=1: the LM streams get the complement
  bool fuse_qkv = true;  // DSM_FUSE_QKV=0: keep the separate QKV reduce launch
  bool chunk_loop = true;  // DSM_CHUNK_LOOP=0: always split K across workgroups
  bool roll_prefetch = true;  // DSM_ROLL=0: the chunk loop requests a chunk only after finishing the previous one (r01 behaviour)
  size_t gemm_lds_pad = 0;    // DSM_GEMM_LDS_PAD (bytes)
  bool gate_occ3 = false;     // DSM_GATE_OCC3=1: the gate's whole-K kernel squeezed to 168 VGPRs (three waves per SIMD, 80 B of spills)
  int loop_depth = 4;         // DSM_LOOP_DEPTH=2: two-block rolling window (fewer registers, three waves per SIMD) where four is the default
  size_t attn_lds_pad = 60000;  // DSM_ATTN_LDS_PAD: extra dynamic LDS per attention workgroup of a large launch (2 per CU)
  int bx3_nt2_min = 256;      // ... from this many workgroups on (DSM_BX3_NT2_MIN)
  bool bx3_nt2 = true;        // DSM_BX3_NT2=0: one n-tile per wave in the whole-K bx3 kernel's plain-epilogue launches
  int dot_mode = 0;           // dsm_asr_config.dot_mode / dsm_tts_config.dot_mode: 1 = the bf16-weight GEMMs in "bx3" (gemm_bx3_kernel)
  int attn_nt = 1;            // DSM_ATTN_NT: ring-cache rows with non-temporal loads: 0 never, 1 bf16 rings (default), 2 every ring
  bool fuse_front = true;     // DSM_FUSE_FRONT=0: the SEANet front end as three GEMM launches (r01)
  int smallk_loop = 1;        // DSM_SMALLK_LOOP=0: one-chunk GEMMs (K <= 256) over many m-tiles stay on gemm_tile_kernel (r01)
  int smallk_min_tiles = 1024;  // DSM_SMALLK_MIN: from how many 64-row tiles on
  int smallk_mt = 4;          // DSM_SMALLK_MT: 16-row tiles per workgroup of those launches
  int chunk_loop_min_tiles = 384;  // DSM_CHUNK_LOOP_MIN (swept at B = 512 / 1024: 384 best)
  bool bx3u = true;           // DSM_BX3U=0: split-K bx3 launches at M <= 32 keep r03's one-block look-ahead (gemm_bx3_kernel)
  bool wk_norm = false;       // DSM_WK_NORM=1: where the MLP input GEMM runs whole-K, norm2 moves into its prologue (gemm_wkn_kernel) and out_proj
                              // stores the residual stream itself (two launches fewer per layer; measured 6.56 against 6.45 ms per TTS step: off)
  int wk_gate_max_chunks = 4; // DSM_WK_GATE_CHUNKS: gated-MLP input GEMMs with at most this many K-chunks run whole-K-in-the-workgroup
  bool bx3u_m64 = true;  // DSM_BX3U_M64=0: 33..64-row narrow GEMMs stay on one 64-row tile
  bool bx3u_late = true;  // DSM_BX3U_LATE=0: the 32-row gemm_bx3u_kernel requests all eight weight blocks up front (129-140 VGPRs)
  int attn_unr = 0;  // DSM_ATTN_UNR=4 / 8: force the bf16-ring attention kernel's keys per lane group and batch (0: by shape)
  bool attn_small = true;  // DSM_ATTN_SMALL=0: rings of at most 32 positions use attn_kernel too
                              // (gemm_wk_kernel: no slabs, no reduce launch); 0: never
  int prio_hi = 0;
  bool serialize_groups = false;  // dsm_debug_serialize_groups: every group on the model stream (profiling aid)
  hipEvent_t ev_codes_consumed = nullptr;
  bool codes_consumed_valid = false;
  hipEvent_t ev_join = nullptr, ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_d = nullptr;
  MimiW mimi_w;
  MimiState mimi[2];
  // mimi[1] — the Mimi clone of asr::State (core/asr.rs:58), used only by dsm_asr_step_pcm[_dev] — is allocated on first use
  // (r04): a server drives the encoder-thread clone (mimi[0]) and step_tokens, and at the capacity batch the second state set is
  // 11 GB of HBM that 256 more streams' ring caches fit into
  std::atomic<bool> mimi1_ready{false};
  std::mutex mimi1_mu;
  MimiDecState dec;
  float* h_pcm_out = nullptr;
  LmW lm_w;
  LmState lm;
  std::vector<void*> allocs;
  std::string err;
  // host staging (pinned)
  float* h_pcm = nullptr;
  float* h_pcm1 = nullptr;
  uint8_t* h_mask = nullptr;
  uint32_t *h_codes = nullptr, *h_text = nullptr;
  float* h_prs = nullptr;
  // host item state + message queue
  std::vector<HostItem> items;
  size_t model_step_idx = 0;
  std::vector<dsm_asr_msg> msgs;
  std::vector<uint32_t> msg_tokens;
  dsm_metrics metrics{};
  // split-K workspaces of the tiled GEMM, one per stream (0 = encoder, 1 = model); grown on first use
  static constexpr int kStreams = 1 + kMaxGroups;
  float* gemm_ws[kStreams] = {};
  size_t gemm_ws_cap[kStreams] = {};
  std::atomic<uint64_t> ws_gen{0};  // bumped whenever a workspace moves: captured graphs hold the old pointer
  // hipGraph replay of the launch-bound inner loops: the kernel sequence of one Mimi encode / decode,
  // of one LM stream group's transformer + heads, of one TTS step is captured once its shapes, pointers and first-call
  // branches have settled (two eager runs), then replayed with ONE hipGraphLaunch — ~150 kernel nodes for ~12 us of
  // host time instead of ~3.5 us each.  Every per-step variable already lives in device buffers, so the captured
  // arguments never change.  DSM_GRAPHS=0 keeps the eager path; profiling brackets force it too.
  struct GraphSlot {
    hipGraphExec_t exec = nullptr;
    uint64_t key = 0, ws_gen = 0;
    int warm = 0;
    int failures = 0;  // captures of this sequence that did not end in a graph; after kMaxCaptureTries it stays eager
    bool disabled = false;
  };
  static constexpr int kMaxCaptureTries = 3;
  bool use_graphs = true;
  // capturing: per host thread (the encoder thread may capture while the model thread launches eagerly)
  static thread_local bool capturing;
  bool capture_failed = false;
  GraphSlot g_enc[2], g_grp[kMaxGroups], g_dec, g_ttsg[kMaxGroups][2];
  std::atomic<uint64_t> graph_launches{0}, eager_bodies{0}, capture_failures{0};
  std::string capture_error;  // what the first failed capture reported (err_mu)
  // a capture that did not produce a graph is never silent: counted, its first reason kept for dsm_metrics
  void note_capture_failure(const char* what, hipError_t he) {
    capture_failures += 1;
    std::lock_guard<std::mutex> lk(err_mu);
    if (capture_error.empty()) {
      capture_error = what;
      if (he != hipSuccess) { capture_error += ": "; capture_error += hipGetErrorString(he); }
    }
  }
  // run-ahead pipeline between the encoder thread and the model thread (dsm_mimi_encode_step_async /
  // dsm_asr_step_tokens_ticket): the reference's sync_channel(100) of PipelineMsg (srv/batched_asr.rs:291), here a ring of
  // kPipe frames: pinned staging for the PCM + mask, a private device copy of the frame's codes, one event "encoded" and
  // one "consumed" per entry.
  static constexpr int kPipe = 4;
  struct PipeSlot {
    float* h_pcm = nullptr;
    uint8_t* h_mask = nullptr;
    uint32_t* d_codes = nullptr;
    hipEvent_t ev_done = nullptr, ev_consumed = nullptr;
    bool in_flight = false;  // encoded (or being encoded) and not yet handed to the model side
  };
  PipeSlot pipe[kPipe];
  int pipe_next = 0;
  bool pipe_ready = false;
  std::mutex pipe_mu;
  // Two host threads may drive one engine (encoder thread || model thread, srv/batched_asr.rs:314,432).  A stream capture
  // must not see the OTHER thread touch the capturing stream or an event of it: the model thread's dsm_streams_join records
  // ev_join ON the encoder stream and then makes the model stream wait for it — issued while the encoder thread is between
  // Begin and EndCapture on that stream, the record lands inside the capture and the wait pulls the model stream into it
  // (EndCapture then fails as "unjoined" / "invalidated", which is what r02 saw now and then); the same goes for
  // hipEventSynchronize / hipStreamWaitEvent on ev_done / ev_consumed against a capture on the stream of their last record.
  // A capture therefore runs alone: every entry point that issues HIP work holds api_mu shared for the whole call,
  // run_captured trades that for the exclusive side around Begin..Instantiate (a few times per engine lifetime).
  std::shared_mutex api_mu;
  static thread_local std::shared_lock<std::shared_mutex>* api_held;
  // per-kernel-class event timing (dsm_prof_*)
  unsigned prof_mask = 0;
  // one slot per stream (0 = encoder, 1 = model): the two host threads of the worker never share a slot
  int tag_gemm[kStreams] = {DSM_PROF_OTHER, DSM_PROF_OTHER, DSM_PROF_OTHER, DSM_PROF_OTHER, DSM_PROF_OTHER};
  int tag_attn[kStreams] = {DSM_PROF_OTHER, DSM_PROF_OTHER, DSM_PROF_OTHER, DSM_PROF_OTHER, DSM_PROF_OTHER};
  std::mutex prof_mu, err_mu;
  int sid(hipStream_t st) const {  // 0 = encoder, 1 = model (= group 0), 1 + g = group g
    if (st == s_enc) return 0;
    for (int g = 1; g < kMaxGroups; ++g)
      if (st == s_grp[g]) return 1 + g;
    return 1;
  }
  struct ProfRec {
    int tag;
    hipEvent_t a, b;
  };
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> prof_pool;
  double prof_total_us[DSM_PROF_NTAGS] = {0};
  uint64_t prof_launches[DSM_PROF_NTAGS] = {0};

  // in-kernel launch brackets (attention kernels): device buffer of (min start, max end) wall-clock pairs
  static constexpr size_t kDevTsCap = 1 << 16;
  unsigned long long* dev_ts = nullptr;
  std::vector<int> dev_ts_tags;  // tag of record i (records are handed out in launch order)
  std::vector<int> dev_ts_info;  // (stream id << 8) | kind of record i: 0 attention, 1 GEMM, 2 its reduce launch
  bool timeline = false;         // dsm_prof_timeline: GEMM launches take records too (two each: the GEMM and its reduce)
  unsigned long long* dev_ts_slot(int tag, int sid_ = 0, int kind = 0, int n = 1) {
    if (!dev_ts || !(prof_mask & (1u << tag))) return nullptr;
    std::lock_guard<std::mutex> lk(prof_mu);
    if (dev_ts_tags.size() + n > kDevTsCap) return nullptr;
    unsigned long long* p = dev_ts + 2 * dev_ts_tags.size();
    for (int i = 0; i < n; ++i) {
      dev_ts_tags.push_back(tag);
      dev_ts_info.push_back((sid_ << 8) | (kind + i));
    }
    return p;
  }

  hipEvent_t prof_event() {
    if (!prof_pool.empty()) {
      hipEvent_t ev = prof_pool.back();
      prof_pool.pop_back();
      return ev;
    }
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    return ev;
  }
  // bracket one launch: returns an index to close with prof_end, or -1 when the class is not selected
  int prof_begin(int tag, hipStream_t st) {
    if (!(prof_mask & (1u << tag))) return -1;
    std::lock_guard<std::mutex> lk(prof_mu);
    ProfRec r{tag, prof_event(), prof_event()};
    (void)hipEventRecord(r.a, st);
    prof_recs.push_back(r);
    return (int)prof_recs.size() - 1;
  }
  void prof_end(int h, hipStream_t st) {
    if (h < 0) return;
    std::lock_guard<std::mutex> lk(prof_mu);
    (void)hipEventRecord(prof_recs[h].b, st);
  }

  void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> lk(err_mu);
    err = buf;
  }

  template <typename T>
  int dalloc(T** out, size_t count, bool zero = true) {
    void* p = nullptr;
    size_t bytes = count * sizeof(T) + 256;  // slack: K-padding reads of the GEMM may run past a row
    HIPCHK_E(this, hipMalloc(&p, bytes));
    // hipMemset / hipMemcpy run on the null stream and may return before the device side is done (a pageable H2D copy
    // returns once the data is staged); the engine's streams are non-blocking, i.e. NOT ordered against the null
    // stream, so a kernel launched right after (load-time table folds, state fills) could read or be overwritten by
    // them.  Load time only: wait.
    if (zero) {
      HIPCHK_E(this, hipMemset(p, 0, bytes));
      HIPCHK_E(this, hipStreamSynchronize(nullptr));
    }
    allocs.push_back(p);
    *out = reinterpret_cast<T*>(p);
    return 0;
  }
  template <typename T>
  int upload(T** out, const T* host, size_t count) {
    if (int rc = dalloc(out, count)) return rc;
    HIPCHK_E(this, hipMemcpy(*out, host, count * sizeof(T), hipMemcpyHostToDevice));
    HIPCHK_E(this, hipStreamSynchronize(nullptr));
    return 0;
  }

  // ---- weight arena: every immutable weight tensor of the STT engine lives in ONE contiguous device
  // allocation, carved in load order, so that a multi-GPU launcher can fan the packed weights out with a single RCCL
  // broadcast and the other ranks attach to the received bytes without reading, converting or packing anything.
  //   W_PLAIN    no arena: upload_w == upload (the TTS engine)
  //   W_MEASURE  first pass over the checkpoint: only adds up the carve sizes
  //   W_LOAD     second pass: carve + host-to-device copy; the answers of the optional-key probes go to `manifest`
  //   W_ATTACH   carve only: the bytes are already there (received arena); probes replay `manifest`
  enum WeightMode { W_PLAIN = 0, W_MEASURE, W_LOAD, W_ATTACH };
  WeightMode wmode = W_PLAIN;
  char* arena = nullptr;
  size_t arena_size = 0, arena_off = 0;
  bool arena_owned = false;
  std::vector<uint8_t> manifest;
  size_t manifest_pos = 0;
  template <typename T>
  int upload_w(T** out, const T* host, size_t count) {
    if (wmode == W_PLAIN) return upload(out, host, count);
    const size_t bytes = (count * sizeof(T) + 256 + 255) & ~(size_t)255;  // same slack as dalloc, 256-byte aligned carves
    if (wmode != W_MEASURE) {
      if (arena_off + bytes > arena_size) {
        set_error("weight arena too small: need %zu bytes at offset %zu of %zu (config / manifest mismatch?)", bytes, arena_off, arena_size);
        return DSM_ERR_INVALID;
      }
      *out = reinterpret_cast<T*>(arena + arena_off);
      if (wmode == W_LOAD) HIPCHK_E(this, hipMemcpy(*out, host, count * sizeof(T), hipMemcpyHostToDevice));
    } else {
      *out = nullptr;
    }
    arena_off += bytes;
    return 0;
  }
  bool skip_host_weights() const { return wmode == W_MEASURE || wmode == W_ATTACH; }
};

thread_local bool dsm_engine::capturing = false;
thread_local std::shared_lock<std::shared_mutex>* dsm_engine::api_held = nullptr;

// shared side of dsm_engine::api_mu for the length of one API call: EVERY entry point that issues HIP work holds it (r03;
// r02 had it on the two ticket entry points only, and the synchronous pair dsm_mimi_encode_step || dsm_asr_step_tokens of
// tests/harness ran unprotected).  Nested entry points (one public call inside another on the same thread) share the outer hold.
struct ApiShared {
  std::shared_lock<std::shared_mutex> lk;
  bool outer;
  explicit ApiShared(dsm_engine* e) : outer(dsm_engine::api_held == nullptr) {
    if (outer) {
      lk = std::shared_lock<std::shared_mutex>(e->api_mu);
      dsm_engine::api_held = &lk;
    }
  }
  ~ApiShared() { if (outer) dsm_engine::api_held = nullptr; }
};
// exclusive side, for a capture: gives up this thread's shared hold first (two threads upgrading at once would deadlock)
struct ApiExclusive {
  std::shared_lock<std::shared_mutex>* held;
  std::unique_lock<std::shared_mutex> lk;
  explicit ApiExclusive(dsm_engine* e) : held(dsm_engine::api_held) {
    if (held) held->unlock();
    lk = std::unique_lock<std::shared_mutex>(e->api_mu);
  }
  ~ApiExclusive() {
    lk.unlock();
    if (held) held->lock();
  }
};

// ----------------------------------------------------------------------------------------------
// weight loading
// ----------------------------------------------------------------------------------------------
namespace {

struct Loader {
  dsm_engine* e;
  dsm_st_file* f;
  bool failed = false;
  std::vector<float> get(int64_t numel, const char* fmt, ...) __attribute__((format(printf, 3, 4))) {
    char name[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    std::vector<float> out((size_t)numel);
    if (failed || e->skip_host_weights()) return out;  // measure / attach: sizes only, nothing is read
    char err[512];
    if (dsm_st_read_f32(f, name, numel, out.data(), err, sizeof err)) {
      e->set_error("%s", err);
      failed = true;
    }
    return out;
  }
  bool has(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
    char name[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name, sizeof name, fmt, ap);
    va_end(ap);
    if (e->wmode == dsm_engine::W_ATTACH) {  // no checkpoint on this rank: replay the loading rank's answers, in order
      if (e->manifest_pos >= e->manifest.size()) { failed = true; e->set_error("weight manifest exhausted"); return false; }
      return e->manifest[e->manifest_pos++] != 0;
    }
    const bool found = dsm_st_find(f, name) != nullptr;
    if (e->wmode == dsm_engine::W_LOAD) e->manifest.push_back(found ? 1 : 0);
    return found;
  }
};

int round_up(int x, int m) { return (x + m - 1) / m * m; }

// pack [N][K] f32 row-major into the GEMM layout [Npad][Kpad] (zero padded)
// bf16 weights (the LM's) are stored FRAGMENT-MAJOR since r04: the 16 rows x 32 k block that one wave loads as its MFMA A operand
// (lane (r, q) = 8 consecutive k of row r) is 1 KB contiguous, tile (n / 16, k / 32) at ((n / 16) * (Kpad / 32) + k / 32) * 512
// elements, lane q * 16 + r inside it — every weight load instruction covers eight full 128-byte lines instead of sixteen half
// lines 4 KB apart, and a wave's chunk is 8 KB of one DRAM page run (experiments/gemm_wk_probe: QKV / gate launches -10 %).
// split_row: a row offset the kernels address as a tile origin (the gate's up half at +hidden): must be a multiple of 16, else
// the matrix stays row-major.  DSM_WPACK=0: row-major everywhere.
int pack_linear(dsm_engine* e, Linear* L, const float* w, int N, int K, bool bf16, const float* bias, int split_row = 0) {
  static const bool wpack = !(getenv("DSM_WPACK") && atoi(getenv("DSM_WPACK")) == 0);
  L->packed = bf16 && wpack && (split_row % 16 == 0);
  L->N = N;
  L->K = K;
  L->Npad = round_up(N, 64) + 64;  // the tiled kernel reads whole 64-row tiles (and the gate's up-tile at +hidden)
  L->Kpad = round_up(K, 32);
  L->bf16 = bf16;
  size_t n = (size_t)L->Npad * L->Kpad;
  const bool skip = e->skip_host_weights();
  if (bf16) {
    std::vector<uint16_t> p(skip ? 0 : n, 0);
    if (!skip) {
      const size_t nblk = (size_t)L->Kpad >> 5;
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < K; ++j) {
          const size_t at = L->packed ? ((((size_t)i >> 4) * nblk + ((size_t)j >> 5)) * 64 + (size_t)(((j >> 3) & 3) * 16 + (i & 15))) * 8 + (j & 7)
                                      : (size_t)i * L->Kpad + j;
          p[at] = dsm_f32_to_bf16(w[(size_t)i * K + j]);
        }
    }
    uint16_t* d = nullptr;
    if (int rc = e->upload_w(&d, p.data(), n)) return rc;
    L->w = d;
  } else {
    std::vector<float> p(skip ? 0 : n, 0.0f);
    if (!skip)
      for (int i = 0; i < N; ++i) memcpy(&p[(size_t)i * L->Kpad], &w[(size_t)i * K], sizeof(float) * K);
    float* d = nullptr;
    if (int rc = e->upload_w(&d, p.data(), n)) return rc;
    L->w = d;
  }
  if (bias) {
    std::vector<float> pb((size_t)L->Npad, 0.0f);
    memcpy(pb.data(), bias, sizeof(float) * N);
    if (int rc = e->upload_w(&L->bias, pb.data(), pb.size())) return rc;
  }
  return 0;
}

// conv1d_weight_norm — core/conv.rs:27-45
std::vector<float> load_conv_weight(Loader& ld, const char* prefix, int out_c, int in_c, int k) {
  if (ld.has("%s.weight", prefix)) return ld.get((int64_t)out_c * in_c * k, "%s.weight", prefix);
  std::vector<float> g = ld.get(out_c, "%s.weight_g", prefix);
  std::vector<float> v = ld.get((int64_t)out_c * in_c * k, "%s.weight_v", prefix);
  for (int o = 0; o < out_c; ++o) {
    float ss = 0.0f;
    for (int i = 0; i < in_c * k; ++i) ss = ss + v[(size_t)o * in_c * k + i] * v[(size_t)o * in_c * k + i];
    float nrm = sqrtf(ss);
    for (int i = 0; i < in_c * k; ++i) v[(size_t)o * in_c * k + i] = v[(size_t)o * in_c * k + i] * g[o] / nrm;
  }
  return v;
}

int load_conv(dsm_engine* e, Loader& ld, ConvGeom* c, const char* prefix, int in_c, int out_c, int k, int stride,
              bool bias, bool replicate, const char* wkey = nullptr) {
  char p[256];
  if (wkey)
    snprintf(p, sizeof p, "%s", wkey);
  else
    snprintf(p, sizeof p, "%s.conv.conv", prefix);
  std::vector<float> w = load_conv_weight(ld, p, out_c, in_c, k);
  std::vector<float> b;
  if (bias) b = ld.get(out_c, "%s.bias", p);
  if (ld.failed) return DSM_ERR_IO;
  // [out_c][in_c][k] -> [out_c][k*in_c] (kk-major reduction index)
  std::vector<float> r((size_t)out_c * k * in_c);
  for (int o = 0; o < out_c; ++o)
    for (int ci = 0; ci < in_c; ++ci)
      for (int kk = 0; kk < k; ++kk) r[((size_t)o * k + kk) * in_c + ci] = w[((size_t)o * in_c + ci) * k + kk];
  c->in_c = in_c;
  c->out_c = out_c;
  c->k = k;
  c->stride = stride;
  c->S = k - stride;
  c->replicate = replicate;
  return pack_linear(e, &c->lin, r.data(), out_c, k * in_c, false, bias ? b.data() : nullptr);
}

int gating_hidden(const dsm_transformer_config& c) {  // core/batched_transformer.rs:153-157
  return c.dim_feedforward == 4 * c.d_model ? 11 * c.d_model / 4 : 2 * c.dim_feedforward / 3;
}

int load_transformer(dsm_engine* e, Loader& ld, TransformerW* t, const dsm_transformer_config& cfg,
                     const char* prefix, bool bf16) {
  t->cfg = cfg;
  const int d = cfg.d_model, hd = d / cfg.num_heads;
  t->hidden = cfg.gating ? gating_hidden(cfg) : cfg.dim_feedforward;
  t->layers.resize(cfg.num_layers);
  std::vector<float> inv(hd / 2);
  for (int i = 0; i < hd / 2; ++i)  // RotaryEmbedding::new — core/transformer.rs:386-392
    inv[i] = (float)(1.0 / pow((double)cfg.max_period, (double)(2 * i) / (double)hd));
  if (int rc = e->upload_w(&t->inv_freq, inv.data(), inv.size())) return rc;
  for (int l = 0; l < cfg.num_layers; ++l) {
    TLayerW& L = t->layers[l];
    {
      auto w = ld.get((int64_t)3 * d * d, "%s.layers.%d.self_attn.in_proj_weight", prefix, l);
      if (ld.failed) return DSM_ERR_IO;
      if (int rc = pack_linear(e, &L.in_proj, w.data(), 3 * d, d, bf16, nullptr)) return rc;
    }
    {
      auto w = ld.get((int64_t)d * d, "%s.layers.%d.self_attn.out_proj.weight", prefix, l);
      if (ld.failed) return DSM_ERR_IO;
      if (int rc = pack_linear(e, &L.out_proj, w.data(), d, d, bf16, nullptr)) return rc;
    }
    for (int which = 1; which <= 2; ++which) {
      float** w = which == 1 ? &L.n1w : &L.n2w;
      float** b = which == 1 ? &L.n1b : &L.n2b;
      std::vector<float> wv, bv;
      if (cfg.norm == 1) {
        wv = ld.get(d, "%s.layers.%d.norm%d.alpha", prefix, l, which);
      } else {
        bv = ld.get(d, "%s.layers.%d.norm%d.bias", prefix, l, which);
        if (ld.has("%s.layers.%d.norm%d.alpha", prefix, l, which))
          wv = ld.get(d, "%s.layers.%d.norm%d.alpha", prefix, l, which);
        else
          wv = ld.get(d, "%s.layers.%d.norm%d.weight", prefix, l, which);
      }
      if (ld.failed) return DSM_ERR_IO;
      if (int rc = e->upload_w(w, wv.data(), wv.size())) return rc;
      if (!bv.empty())
        if (int rc = e->upload_w(b, bv.data(), bv.size())) return rc;
    }
    if (cfg.gating) {
      auto wi = ld.get((int64_t)2 * t->hidden * d, "%s.layers.%d.gating.linear_in.weight", prefix, l);
      if (ld.failed) return DSM_ERR_IO;
      if (int rc = pack_linear(e, &L.ff_in, wi.data(), 2 * t->hidden, d, bf16, nullptr, t->hidden)) return rc;
      auto wo = ld.get((int64_t)d * t->hidden, "%s.layers.%d.gating.linear_out.weight", prefix, l);
      if (ld.failed) return DSM_ERR_IO;
      if (int rc = pack_linear(e, &L.ff_out, wo.data(), d, t->hidden, bf16, nullptr)) return rc;
    } else {
      auto wi = ld.get((int64_t)t->hidden * d, "%s.layers.%d.linear1.weight", prefix, l);
      if (ld.failed) return DSM_ERR_IO;
      if (int rc = pack_linear(e, &L.ff_in, wi.data(), t->hidden, d, bf16, nullptr)) return rc;
      auto wo = ld.get((int64_t)d * t->hidden, "%s.layers.%d.linear2.weight", prefix, l);
      if (ld.failed) return DSM_ERR_IO;
      if (int rc = pack_linear(e, &L.ff_out, wo.data(), d, t->hidden, bf16, nullptr)) return rc;
    }
    if (cfg.layer_scale) {
      auto s1 = ld.get(d, "%s.layers.%d.layer_scale_1.scale", prefix, l);
      auto s2 = ld.get(d, "%s.layers.%d.layer_scale_2.scale", prefix, l);
      if (ld.failed) return DSM_ERR_IO;
      if (int rc = e->upload_w(&L.ls1, s1.data(), s1.size())) return rc;
      if (int rc = e->upload_w(&L.ls2, s2.data(), s2.size())) return rc;
    }
  }
  return 0;
}

int alloc_transformer_state(dsm_engine* e, TransformerState* st, const dsm_transformer_config& cfg, int B, int T,
                            bool kv_bf16) {
  const int H = cfg.num_heads, hd = cfg.d_model / H;
  size_t per = (size_t)B * H * cfg.context * hd;
  st->k.resize(cfg.num_layers);
  st->v.resize(cfg.num_layers);
  for (int l = 0; l < cfg.num_layers; ++l) {
    if (kv_bf16) {
      uint16_t *k = nullptr, *v = nullptr;
      if (int rc = e->dalloc(&k, per)) return rc;
      if (int rc = e->dalloc(&v, per)) return rc;
      st->k[l] = k;
      st->v[l] = v;
    } else {
      float *k = nullptr, *v = nullptr;
      if (int rc = e->dalloc(&k, per)) return rc;
      if (int rc = e->dalloc(&v, per)) return rc;
      st->k[l] = k;
      st->v[l] = v;
    }
  }
  if (int rc = e->dalloc(&st->pos, B)) return rc;
  if (int rc = e->dalloc(&st->idx, B)) return rc;
  if (int rc = e->dalloc(&st->start_pos, B)) return rc;
  if (int rc = e->dalloc(&st->widx, (size_t)B * T)) return rc;
  if (int rc = e->dalloc(&st->rope_cs, (size_t)B * T * hd)) return rc;
  return 0;
}

int load_rvq(dsm_engine* e, Loader& ld, RvqW* r, const char* prefix, int n_q, const dsm_mimi_config& m) {
  const int bins = m.quantizer_bins, dim = m.quantizer_dim;
  r->n_q = n_q;
  auto ip = ld.get((int64_t)dim * m.dimension, "%s.input_proj.weight", prefix);
  if (ld.failed) return DSM_ERR_IO;
  if (int rc = pack_linear(e, &r->input_proj, ip.data(), dim, m.dimension, false, nullptr)) return rc;
  if (ld.has("%s.output_proj.weight", prefix)) {  // decode side; STT-only checkpoints may omit nothing, but be lenient
    auto op = ld.get((int64_t)m.dimension * dim, "%s.output_proj.weight", prefix);
    if (ld.failed) return DSM_ERR_IO;
    if (int rc = pack_linear(e, &r->output_proj, op.data(), m.dimension, dim, false, nullptr)) return rc;
  }
  r->codebooks.resize(n_q);
  for (int i = 0; i < n_q; ++i) {
    auto usage = ld.get(bins, "%s.vq.layers.%d._codebook.cluster_usage", prefix, i);
    auto esum = ld.get((int64_t)bins * dim, "%s.vq.layers.%d._codebook.embedding_sum", prefix, i);
    if (ld.failed) return DSM_ERR_IO;
    std::vector<float> c2(bins);
    for (int j = 0; j < bins; ++j) {  // EuclideanCodebook::new — core/quantization.rs:86-95
      float u = usage[j] > 1e-5f ? usage[j] : 1e-5f;
      float ss = 0.0f;
      for (int dd = 0; dd < dim; ++dd) {
        float v = esum[(size_t)j * dim + dd] / u;
        esum[(size_t)j * dim + dd] = v;
        ss = ss + v * v;
      }
      c2[j] = ss / 2.0f;
    }
    if (int rc = pack_linear(e, &r->codebooks[i], esum.data(), bins, dim, false, c2.data())) return rc;
  }
  return 0;
}

int load_mimi(dsm_engine* e, Loader& ld, MimiW* m, const dsm_mimi_config& cfg) {
  m->cfg = cfg;
  char p[128];
  int mult = 1, idx = 0, T = DSM_FRAME_SIZE;
  snprintf(p, sizeof p, "encoder.model.%d", idx);
  if (int rc = load_conv(e, ld, &m->init_conv, p, cfg.channels, mult * cfg.n_filters, cfg.kernel_size, 1, true, false)) return rc;
  m->init_conv.T_in = T;
  m->init_conv.T_out = T;
  idx += 1;
  m->stages.resize(cfg.n_ratios);
  for (int i = 0; i < cfg.n_ratios; ++i) {
    int ratio = cfg.ratios[cfg.n_ratios - 1 - i];  // core/seanet.rs:194 ratios.iter().rev()
    int dim = mult * cfg.n_filters, hidden = dim / cfg.compress;
    MimiW::Stage& st = m->stages[i];
    snprintf(p, sizeof p, "encoder.model.%d.block.1", idx);
    if (int rc = load_conv(e, ld, &st.ra, p, dim, hidden, cfg.residual_kernel_size, 1, true, false)) return rc;
    snprintf(p, sizeof p, "encoder.model.%d.block.3", idx);
    if (int rc = load_conv(e, ld, &st.rb, p, hidden, dim, 1, 1, true, false)) return rc;
    idx += 1;
    snprintf(p, sizeof p, "encoder.model.%d", idx + 1);
    if (int rc = load_conv(e, ld, &st.down, p, dim, dim * 2, ratio * 2, ratio, true, false)) return rc;
    idx += 2;
    st.ra.T_in = st.ra.T_out = st.rb.T_in = st.rb.T_out = st.down.T_in = T;
    if (T % ratio) {
      e->set_error("frame of %d samples does not divide by the encoder ratios", DSM_FRAME_SIZE);
      return DSM_ERR_INVALID;
    }
    T /= ratio;
    st.down.T_out = T;
    mult *= 2;
  }
  snprintf(p, sizeof p, "encoder.model.%d", idx + 1);
  if (int rc = load_conv(e, ld, &m->final_conv, p, mult * cfg.n_filters, cfg.dimension, cfg.last_kernel_size, 1, true, false)) return rc;
  m->final_conv.T_in = m->final_conv.T_out = T;
  if (int rc = load_transformer(e, ld, &m->tr, cfg.transformer, "encoder_transformer.transformer", false)) return rc;
  if (int rc = load_conv(e, ld, &m->downsample, nullptr, cfg.dimension, cfg.dimension, 2 * cfg.downsample_stride,
                         cfg.downsample_stride, false, true, "downsample.conv.conv.conv"))
    return rc;
  m->downsample.T_in = T;
  if (T % cfg.downsample_stride) {
    e->set_error("encoder frames per step (%d) do not divide by the downsample stride", T);
    return DSM_ERR_INVALID;
  }
  m->downsample.T_out = T / cfg.downsample_stride;
  if (m->downsample.T_out != 1) {
    e->set_error("expected exactly one latent frame per step, got %d", m->downsample.T_out);
    return DSM_ERR_INVALID;
  }
  if (int rc = load_rvq(e, ld, &m->rvq_first, "quantizer.rvq_first", 1, cfg)) return rc;
  if (cfg.quantizer_n_q > 1)
    if (int rc = load_rvq(e, ld, &m->rvq_rest, "quantizer.rvq_rest", cfg.quantizer_n_q - 1, cfg)) return rc;
  // ---- decode side: only if the checkpoint carries it (the STT worker never calls decode_step) ----
  m->has_decoder = ld.has("decoder.model.0.conv.conv.weight") || ld.has("decoder.model.0.conv.conv.weight_v");
  if (!m->has_decoder) return 0;
  {
    int mult = 1 << cfg.n_ratios, idx = 0, T = m->final_conv.T_out;  // frames per step at the encoder rate (2)
    char p[128];
    snprintf(p, sizeof p, "decoder.model.%d", idx);
    if (int rc = load_conv(e, ld, &m->dec_init, p, cfg.dimension, mult * cfg.n_filters, cfg.kernel_size, 1, true, false)) return rc;
    m->dec_init.T_in = m->dec_init.T_out = T;
    idx += 1;
    m->dec_stages.resize(cfg.n_ratios);
    for (int i = 0; i < cfg.n_ratios; ++i) {
      const int ratio = cfg.ratios[i];
      MimiW::DecStage& st = m->dec_stages[i];
      st.in_c = mult * cfg.n_filters;
      st.out_c = st.in_c / 2;
      st.k = 2 * ratio;
      st.stride = ratio;
      st.T_in = T;
      snprintf(p, sizeof p, "decoder.model.%d.convtr.convtr", idx + 1);
      std::vector<float> w;
      if (ld.has("%s.weight", p)) {
        w = ld.get((int64_t)st.in_c * st.out_c * st.k, "%s.weight", p);
      } else {  // weight norm over dims (1,2) of [in_c, out_c, k] — core/conv.rs:136-139
        auto g = ld.get(st.in_c, "%s.weight_g", p);
        w = ld.get((int64_t)st.in_c * st.out_c * st.k, "%s.weight_v", p);
        for (int ci = 0; ci < st.in_c && !ld.failed; ++ci) {
          float ss = 0.0f;
          const size_t n = (size_t)st.out_c * st.k;
          for (size_t j = 0; j < n; ++j) ss = ss + w[ci * n + j] * w[ci * n + j];
          float nrm = sqrtf(ss);
          for (size_t j = 0; j < n; ++j) w[ci * n + j] = w[ci * n + j] * g[ci] / nrm;
        }
      }
      auto b = ld.get(st.out_c, "%s.bias", p);
      if (ld.failed) return DSM_ERR_IO;
      // [in_c][out_c][k] -> rows (kk*out_c + co), K = in_c
      std::vector<float> r((size_t)st.k * st.out_c * st.in_c);
      for (int ci = 0; ci < st.in_c; ++ci)
        for (int co = 0; co < st.out_c; ++co)
          for (int kk = 0; kk < st.k; ++kk)
            r[((size_t)kk * st.out_c + co) * st.in_c + ci] = w[((size_t)ci * st.out_c + co) * st.k + kk];
      if (int rc = pack_linear(e, &st.up, r.data(), st.k * st.out_c, st.in_c, false, nullptr)) return rc;
      if (int rc = e->upload_w(&st.up_bias, b.data(), b.size())) return rc;
      idx += 2;
      T *= ratio;
      const int dim = st.out_c, hidden = dim / cfg.compress;
      snprintf(p, sizeof p, "decoder.model.%d.block.1", idx);
      if (int rc = load_conv(e, ld, &st.ra, p, dim, hidden, cfg.residual_kernel_size, 1, true, false)) return rc;
      snprintf(p, sizeof p, "decoder.model.%d.block.3", idx);
      if (int rc = load_conv(e, ld, &st.rb, p, hidden, dim, 1, 1, true, false)) return rc;
      st.ra.T_in = st.ra.T_out = st.rb.T_in = st.rb.T_out = T;
      idx += 1;
      mult /= 2;
    }
    snprintf(p, sizeof p, "decoder.model.%d", idx + 1);
    if (int rc = load_conv(e, ld, &m->dec_final, p, cfg.n_filters, cfg.channels, cfg.last_kernel_size, 1, true, false)) return rc;
    m->dec_final.T_in = m->dec_final.T_out = T;
    if (T != DSM_FRAME_SIZE) {
      e->set_error("decoder emits %d samples per step, expected %d", T, DSM_FRAME_SIZE);
      return DSM_ERR_INVALID;
    }
    if (int rc = load_transformer(e, ld, &m->dec_tr, cfg.transformer, "decoder_transformer.transformer", false)) return rc;
    const int st_ = cfg.downsample_stride, dim = cfg.dimension;
    auto uw = ld.get((int64_t)dim * 2 * st_, "upsample.convtr.convtr.convtr.weight");  // [dim][1][k]
    if (ld.failed) return DSM_ERR_IO;
    std::vector<float> ur((size_t)2 * st_ * dim);
    for (int c = 0; c < dim; ++c)
      for (int kk = 0; kk < 2 * st_; ++kk) ur[(size_t)kk * dim + c] = uw[(size_t)c * 2 * st_ + kk];
    if (int rc = e->upload_w(&m->upsample_w, ur.data(), ur.size())) return rc;
    std::vector<const float*> ptrs;
    ptrs.push_back(reinterpret_cast<const float*>(m->rvq_first.codebooks[0].w));
    for (auto& cb : m->rvq_rest.codebooks) ptrs.push_back(reinterpret_cast<const float*>(cb.w));
    if (e->wmode != dsm_engine::W_MEASURE)  // a table of device pointers: per engine, never part of the arena
      if (int rc = e->upload(&m->emb_ptrs, ptrs.data(), ptrs.size())) return rc;
  }
  return 0;
}

int alloc_cat(dsm_engine* e, float** out, const ConvGeom& c, int B) {
  return e->dalloc(out, (size_t)B * (c.S + c.T_in) * c.in_c);
}

void add_desc(MimiState* s, float* cat, const ConvGeom& c) {
  if (c.S == 0) return;
  ConvStateDesc d;
  d.cat = cat;
  d.bstride = (long)(c.S + c.T_in) * c.in_c;
  d.S = c.S;
  d.T = c.T_in;
  d.C = c.in_c;
  d.replicate = c.replicate ? 1 : 0;
  s->h_descs.push_back(d);
}

int alloc_mimi_state(dsm_engine* e, MimiState* s, const MimiW& w, int B) {
  const dsm_mimi_config& cfg = w.cfg;
  if (int rc = alloc_cat(e, &s->cat_init, w.init_conv, B)) return rc;
  add_desc(s, s->cat_init, w.init_conv);
  s->stages.resize(w.stages.size());
  for (size_t i = 0; i < w.stages.size(); ++i) {
    const MimiW::Stage& st = w.stages[i];
    if (int rc = e->dalloc(&s->stages[i].y, (size_t)B * st.ra.T_in * st.ra.in_c)) return rc;
    if (int rc = alloc_cat(e, &s->stages[i].cat_ra, st.ra, B)) return rc;
    if (int rc = alloc_cat(e, &s->stages[i].cat_rb, st.rb, B)) return rc;
    if (int rc = alloc_cat(e, &s->stages[i].cat_down, st.down, B)) return rc;
    add_desc(s, s->stages[i].cat_ra, st.ra);
    add_desc(s, s->stages[i].cat_rb, st.rb);
    add_desc(s, s->stages[i].cat_down, st.down);
  }
  if (int rc = alloc_cat(e, &s->cat_final, w.final_conv, B)) return rc;
  add_desc(s, s->cat_final, w.final_conv);
  const int Tt = w.final_conv.T_out, d = cfg.dimension;
  if (int rc = e->dalloc(&s->x_tr, (size_t)B * Tt * d)) return rc;
  if (int rc = e->dalloc(&s->xn, (size_t)B * Tt * d)) return rc;
  if (int rc = e->dalloc(&s->q, (size_t)B * Tt * d)) return rc;
  if (int rc = e->dalloc(&s->att, (size_t)B * Tt * d)) return rc;
  if (int rc = e->dalloc(&s->ff, (size_t)B * Tt * cfg.transformer.dim_feedforward)) return rc;
  if (int rc = alloc_cat(e, &s->cat_ds, w.downsample, B)) return rc;
  s->ds_desc = (int)s->h_descs.size();
  add_desc(s, s->cat_ds, w.downsample);
  if (int rc = e->dalloc(&s->latent, (size_t)B * d)) return rc;
  if (int rc = e->dalloc(&s->res_first, (size_t)B * cfg.quantizer_dim)) return rc;
  if (int rc = e->dalloc(&s->res_rest, (size_t)B * cfg.quantizer_dim)) return rc;
  const int n_tiles = (cfg.quantizer_bins + 15) / 16;
  if (int rc = e->dalloc(&s->pval, (size_t)n_tiles * B)) return rc;
  if (int rc = e->dalloc(&s->pidx, (size_t)n_tiles * B)) return rc;
  if (int rc = e->dalloc(&s->codes, (size_t)B * cfg.quantizer_n_q)) return rc;
  if (int rc = e->dalloc(&s->mask, B)) return rc;
  if (int rc = alloc_transformer_state(e, &s->tr, cfg.transformer, B, Tt, false)) return rc;
  if (int rc = e->upload(&s->descs, s->h_descs.data(), s->h_descs.size())) return rc;
  s->pcm = s->cat_init + (size_t)w.init_conv.S * w.init_conv.in_c;  // slot 0's frame; slots are bstride apart
  return 0;
}

// ----------------------------------------------------------------------------------------------
// GEMM launch
// ----------------------------------------------------------------------------------------------
// whole-K-in-the-workgroup GEMMs (dsm_gemm_wk.h): dot_mode 1, bf16 weights, at most four K-chunks (DSM_WK_GATE_CHUNKS), M <= 64
bool wk_applicable(const dsm_engine* e, bool bf16_weights, int Kpad, int K, int M) {
  const int chunks = (Kpad + DSM_KC - 1) / DSM_KC;
  return bf16_weights && e->dot_mode == 1 && K % 32 == 0 && Kpad == K && chunks <= e->wk_gate_max_chunks && chunks <= 4 && M <= 64;
}

template <typename WT, typename KVT, int EPI, int NT>
int launch_gemm_tiled(dsm_engine* e, hipStream_t st, GemmArgs& a) {
  int chunks = (a.Kpad + DSM_KC - 1) / DSM_KC;
  const int gx = (a.N + 63) / 64;
  // Enough (n, m) tiles to fill the chip (large batches; the Mimi convs, whose M is B x frames): no
  // split-K across workgroups — each walks the chunks itself and sums them in order in registers, so the slabs
  // (chunks x M x N floats written, then read back by a reduce launch) disappear.
  a.chunk_loop = 0;
  if (chunks > 1 && e->chunk_loop && (long)gx * ((a.M + 63) / 64) >= e->chunk_loop_min_tiles) {
    a.chunk_loop = chunks;
    a.defer_reduce = 0;
    chunks = 1;
  }
  // r04: short reductions (the DepFormer's: K = 1024) keep the whole K inside the workgroup — four waves, one chunk each,
  // 16 rows x one (gate, up) tile pair (or one tile) per workgroup, the epilogue behind the ordered LDS sum: no slabs, no
  // reduce launch.  experiments/gemm_wk_probe: gate 7.5 us against 13.4 (10.9 with gemm_bx3u_kernel) at M = 32; at K = 2048 the
  // activation re-read (every workgroup reads 16 x K x 4 bytes from L2) makes it lose (27 against 20 us).  With
  // a.pre_norm_w the row norm of the input runs in the kernel's prologue (gemm_wkn_kernel) and the norm launch goes too.
  {
    const bool wk_can = wk_applicable(e, sizeof(WT) == 2, a.Kpad, a.K, a.M) && a.chunk_loop == 0 && chunks > 1 && a.N % 16 == 0;
    const bool wk = wk_can && ((EPI == EPI_GATE && NT == 2) ? true : (EPI == EPI_STORE && NT == 1 && a.wk_hint));
    if (a.pre_norm_w && !(wk && EPI == EPI_GATE)) {
      e->set_error("internal: a norm prologue was requested for a GEMM that does not run whole-K (K=%d M=%d)", a.K, a.M);
      return DSM_ERR_STATE;
    }
    if (wk) {
      auto ok4 = [](const RowMap& m) { return m.ld % 4 == 0 && m.bstride % 4 == 0; };
      a.vec = (a.N % 4 == 0) && (!a.Y || ok4(a.ymap)) && (!a.Y2 || ok4(a.y2map)) && (!a.res || ok4(a.rmap));
      a.ts = e->timeline ? e->dev_ts_slot(e->tag_gemm[e->sid(st)], e->sid(st), 1, 2) : nullptr;
      const int ph = e->prof_begin(e->tag_gemm[e->sid(st)], st);
      const dim3 grid(a.N / 16, 1, (a.M + 15) / 16);
      if (EPI == EPI_GATE && a.pre_norm_w && a.pre_norm_rms)
        hipLaunchKernelGGL((gemm_wkn_kernel<KVT, 2, EPI_GATE, true>), grid, dim3(256), (size_t)chunks * 2 * 1024 + 512 + 8192, st, a);
      else if (EPI == EPI_GATE && a.pre_norm_w)
        hipLaunchKernelGGL((gemm_wkn_kernel<KVT, 2, EPI_GATE, false>), grid, dim3(256), (size_t)chunks * 2 * 1024 + 512 + 8192, st, a);
      else if (EPI == EPI_GATE)
        hipLaunchKernelGGL((gemm_wk_kernel<KVT, 1, 2, EPI_GATE, 1, 4, 2, true, 4>), grid, dim3(256), (size_t)chunks * 2 * 1024, st, a);
      else
        hipLaunchKernelGGL((gemm_wk_kernel<KVT, 1, 1, EPI_STORE, 1, 4, 2, true, 4>), grid, dim3(256), (size_t)chunks * 1024, st, a);
      e->prof_end(ph, st);
      HIPCHK(hipGetLastError());
      if (a.norm_out) {
        hipLaunchKernelGGL(row_norm_kernel, dim3(a.M), dim3(256), 0, st, a.norm_out, a.Y, a.norm_w, a.norm_b, a.M, a.N, a.norm_eps, a.norm_rms);
        HIPCHK(hipGetLastError());
      }
      return 0;
    }
  }
  int MT = a.M > 32 ? 4 : (a.M > 16 ? 2 : 1);
  while (MT > 1 && (long)gx * chunks * ((a.M + 16 * MT - 1) / (16 * MT)) < 256) MT /= 2;  // cover the 256 CUs
  // 33..64 rows, dot_mode 1, a launch of at most 256 workgroups (out_proj of a 2048-wide model): two 32-row z-tiles on
  // gemm_bx3u_kernel instead of one 64-row tile on gemm_bx3_kernel — 12.8 against 15.4 us with its reduce (experiments/gemm_wk_probe 3,
  // form 5); the wider launches (QKV, gate, ff_out) tie or lose that way and keep MT = 4.  DSM_BX3U_M64=0: off.
  if (MT == 4 && e->bx3u && e->bx3u_m64 && e->dot_mode == 1 && sizeof(WT) == 2 && chunks > 1 && a.chunk_loop == 0 && a.M <= 64 &&
      (long)gx * chunks <= 256 && EPI == EPI_STORE)
    MT = 2;
  // one K-chunk and thousands of m-tiles (the first SEANet layers at large batches: K = 32..192, M = B x 1920): a
  // workgroup is one short dependent chain — loads, one to six MFMA blocks, residual load, store — so what counts is how
  // many of them a CU holds; gemm_tile_kernel's up-front window of eight blocks costs 200-230 VGPRs (two workgroups per CU),
  // gemm_loop_kernel's two-block window 150 (three).  Mimi encode alone at B = 2048: 17.7 -> 16.6 ms; 8-row tiles no better.
  const bool smallk = e->smallk_loop && chunks == 1 && a.chunk_loop == 0 && (long)gx * ((a.M + 63) / 64) >= e->smallk_min_tiles &&
                      !(e->dot_mode == 1 && sizeof(WT) == 2);
  if (smallk && e->smallk_mt < MT) MT = e->smallk_mt;
  auto ok4 = [](const RowMap& m) { return m.ld % 4 == 0 && m.bstride % 4 == 0; };
  a.vec = (a.N % 4 == 0) && (!a.Y || ok4(a.ymap)) && (!a.Y2 || ok4(a.y2map)) && (!a.res || ok4(a.rmap));
  a.ws_ntiles = (((NT - 1) * a.nt_stride) >> 4) + gx * 4;
  const int mtiles = (a.M + 15) / 16;
  if (chunks > 1) {
    const int wsid = e->sid(st);
    size_t need = (size_t)chunks * mtiles * a.ws_ntiles * 256 * sizeof(float);
    if (need > e->gemm_ws_cap[wsid]) {  // first use of a bigger shape: grow (never happens in steady state)
      if (e->capturing) {  // a graph capture cannot allocate: give up on this capture, the caller reruns the body eagerly
        e->capture_failed = true;
        e->set_error("split-K workspace grew during a graph capture");
        return DSM_ERR_STATE;
      }
      e->ws_gen += 1;
      HIPCHK(hipStreamSynchronize(st));
      if (e->gemm_ws[wsid]) HIPCHK(hipFree(e->gemm_ws[wsid]));
      e->gemm_ws[wsid] = nullptr;
      e->gemm_ws_cap[wsid] = 0;
      void* p = nullptr;
      HIPCHK(hipMalloc(&p, need));
      e->gemm_ws[wsid] = reinterpret_cast<float*>(p);
      e->gemm_ws_cap[wsid] = need;
    }
    a.ws = e->gemm_ws[wsid];
  }
  dim3 grid(gx, chunks, (a.M + 16 * MT - 1) / (16 * MT));
  // dot_mode 1, whole-K form, plain epilogues: two n-tiles per wave (128 weight rows per workgroup).  With one n-tile a wave
  // reads 12 LDS fragments (12 KB) per block for 12 MFMAs and the LDS, not the matrix pipe, bounds the loop; the gate has
  // always run two.  DSM_BX3_NT2=0: one.
  const bool nt2 = NT == 1 && e->dot_mode == 1 && sizeof(WT) == 2 && e->bx3_nt2 && a.chunk_loop > 1 && MT == 4 &&
                   (EPI == EPI_STORE || EPI == EPI_QKV) && a.N % 128 == 0 && (long)(a.N / 128) * grid.z >= e->bx3_nt2_min;
  if (nt2) {
    grid.x = a.N / 128;
    a.wg_cols = 128;
    a.nt_stride = 64;
  }
  a.ts = e->timeline ? e->dev_ts_slot(e->tag_gemm[e->sid(st)], e->sid(st), 1, 2) : nullptr;
  const int ph = e->prof_begin(e->tag_gemm[e->sid(st)], st);
  const bool bx3 = e->dot_mode == 1 && sizeof(WT) == 2;  // dot_mode 1: every bf16-weight GEMM on the bf16 matrix pipe
  const bool roll = (a.chunk_loop > 1 && e->roll_prefetch) || smallk;  // whole K in the workgroup with a rolling load window
  constexpr int DMAX = LoopDepth<WT, NT>::MAX;
  const bool deep = DMAX == 4 && e->loop_depth == 4 && !smallk;
  // extra dynamic LDS per GEMM workgroup (never touched): caps how many of them a CU takes, so that another stream's
  // attention workgroups keep register file and wave slots beside them (DSM_GEMM_LDS_PAD, large launches only)
  const size_t pad = ((long)grid.x * grid.y * grid.z >= 1024) ? e->gemm_lds_pad : 0;
#define DSM_LAUNCH_TILED(MTv)                                                                                   \
  if (roll && NT == 2 && MTv == 4 && e->gate_occ3)                                                              \
    hipLaunchKernelGGL((gemm_loop_kernel<WT, KVT, MTv, NT, EPI, 2, 3>), grid, dim3(256), pad, st, a);          \
  else if (roll && deep) hipLaunchKernelGGL((gemm_loop_kernel<WT, KVT, MTv, NT, EPI, DMAX>), grid, dim3(256), pad, st, a); \
  else if (roll) hipLaunchKernelGGL((gemm_loop_kernel<WT, KVT, MTv, NT, EPI, 2>), grid, dim3(256), pad, st, a);     \
  else hipLaunchKernelGGL((gemm_tile_kernel<WT, KVT, MTv, NT, EPI>), grid, dim3(256), pad, st, a);
  // r04: the split-K form at M <= 32 issues every load of its chunk up front (gemm_bx3u_kernel, 48 KB of LDS at MT = 2)
#define DSM_LAUNCH_BX3(MTv)                                                                                     \
  if (a.chunk_loop > 1) hipLaunchKernelGGL((gemm_bx3_kernel<KVT, MTv, NT, EPI, true>), grid, dim3(256), pad, st, a); \
  else if (MTv == 2 && e->bx3u && e->bx3u_late && pad == 0) hipLaunchKernelGGL((gemm_bx3u_kernel<KVT, 2, NT, EPI, 8, 0, 4>), grid, dim3(256), 8 * 3 * 32 * 32 * 2, st, a); \
  else if (MTv <= 2 && e->bx3u && pad == 0) hipLaunchKernelGGL((gemm_bx3u_kernel<KVT, (MTv <= 2 ? MTv : 2), NT, EPI, 8>), grid, dim3(256), 8 * 3 * 16 * (MTv <= 2 ? MTv : 2) * 32 * 2, st, a); \
  else hipLaunchKernelGGL((gemm_bx3_kernel<KVT, MTv, NT, EPI, false>), grid, dim3(256), pad, st, a);
  if (nt2) {
    hipLaunchKernelGGL((gemm_bx3_kernel<KVT, 4, (NT == 1 ? 2 : NT), (EPI == EPI_GATE ? EPI_STORE : EPI), true>), grid, dim3(256), pad, st, a);
  } else if (bx3 && EPI != EPI_RVQ) {
    if (MT == 4) { DSM_LAUNCH_BX3(4) } else if (MT == 2) { DSM_LAUNCH_BX3(2) } else { DSM_LAUNCH_BX3(1) }
  } else if (MT == 4) { DSM_LAUNCH_TILED(4) } else if (MT == 2) { DSM_LAUNCH_TILED(2) } else { DSM_LAUNCH_TILED(1) }
#undef DSM_LAUNCH_BX3
#undef DSM_LAUNCH_TILED
  const bool rows_ok = (EPI == EPI_STORE) && a.norm_out && a.vec && !a.Y2 && a.Y && a.N <= 4096 && a.ymap.bstride == 0;
  if (chunks > 1 && !((EPI == EPI_QKV || EPI == EPI_STORE) && a.defer_reduce)) {  // deferred: the consumer sums the slabs (AttnFused, LogitSrc)
    if (rows_ok) {
      if (a.N <= 1024) hipLaunchKernelGGL(gemm_reduce_rows_kernel<1>, dim3(a.M), dim3(256), 0, st, a, chunks);
      else if (a.N <= 2048) hipLaunchKernelGGL(gemm_reduce_rows_kernel<2>, dim3(a.M), dim3(512), 0, st, a, chunks);
      else hipLaunchKernelGGL(gemm_reduce_rows_kernel<4>, dim3(a.M), dim3(1024), 0, st, a, chunks);
    } else {
      const int out_tiles = mtiles * ((a.N + 15) / 16);
      hipLaunchKernelGGL((gemm_reduce_kernel<KVT, EPI>), dim3((out_tiles + 3) / 4), dim3(256), 0, st, a, chunks);
    }
  }
  e->prof_end(ph, st);
  HIPCHK(hipGetLastError());
  if (a.norm_out && !(chunks > 1 && rows_ok)) {  // the norm could not be fused: run it on the stored rows
    hipLaunchKernelGGL(row_norm_kernel, dim3(a.M), dim3(256), 0, st, a.norm_out, a.Y, a.norm_w, a.norm_b, a.M,
                       a.N, a.norm_eps, a.norm_rms);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

template <typename WT, typename KVT, int EPI, int NT>
int launch_gemm_t(dsm_engine* e, hipStream_t st, GemmArgs& a, bool aligned) {
  if (aligned && a.K % 32 == 0) return launch_gemm_tiled<WT, KVT, EPI, NT>(e, st, a);
  const int chunks = (a.Kpad + DSM_KC - 1) / DSM_KC;
  const int rounds = (chunks + 15) / 16;          // chunks per wave when there are more than 16
  const int S = (chunks + rounds - 1) / rounds;   // waves per workgroup
  const int tiles16 = (a.N + 15) / 16;  // for the gate a.N is the hidden width: one (gate, up) tile pair per block
  const int nx = (EPI == EPI_GATE) ? tiles16 : (tiles16 + NT - 1) / NT;
  // M-tiles per wave: as many as possible (weights are re-read once per m-group) while the grid still covers
  // the 256 CUs at least twice and the chunk partials fit in LDS
  int MT = (NT == 2) ? 2 : 4;  // NT=2 x MT=4 would need > 128 VGPRs (spills under the 1024-thread cap)
  while (MT > 1 && (a.M <= 16 * (MT / 2) || (long)nx * ((a.M + 16 * MT - 1) / (16 * MT)) * S < 2048 ||
                    (chunks > 1 && (size_t)chunks * NT * MT * 1024 > 64 * 1024)))
    MT /= 2;
  if (chunks > 1 && (size_t)chunks * NT * MT * 1024 > 160 * 1024) {
    e->set_error("GEMM K=%d: chunk partials do not fit in LDS", a.K);
    return DSM_ERR_INVALID;
  }
  // 16-byte epilogue accesses need every row offset to be a multiple of 4 floats
  auto ok4 = [](const RowMap& m) { return m.ld % 4 == 0 && m.bstride % 4 == 0; };
  a.vec = (a.N % 4 == 0) && (!a.Y || ok4(a.ymap)) && (!a.Y2 || ok4(a.y2map)) && (!a.res || ok4(a.rmap));
  aligned = aligned && (a.K % 32 == 0);  // the fast kernel has no K-tail handling
  dim3 grid(nx, (a.M + 16 * MT - 1) / (16 * MT));
  dim3 block(64 * S);
  size_t lds = chunks > 1 ? (size_t)chunks * NT * MT * 1024 : 0;
  const bool bx3 = e->dot_mode == 1 && sizeof(WT) == 2 && EPI != EPI_RVQ;
#define DSM_LAUNCH(MTv, AL)                                                                                   \
  do {                                                                                                        \
    if (bx3) hipLaunchKernelGGL((gemm_mfma_kernel<WT, KVT, MTv, NT, EPI, AL, true>), grid, block, lds, st, a); \
    else hipLaunchKernelGGL((gemm_mfma_kernel<WT, KVT, MTv, NT, EPI, AL>), grid, block, lds, st, a);          \
  } while (0)
  const int ph = e->prof_begin(e->tag_gemm[e->sid(st)], st);
  if (MT == 1) {
    if (aligned) DSM_LAUNCH(1, true); else DSM_LAUNCH(1, false);
  } else if (MT == 2) {
    if (aligned) DSM_LAUNCH(2, true); else DSM_LAUNCH(2, false);
  } else {
    if (aligned) DSM_LAUNCH(4, true); else DSM_LAUNCH(4, false);
  }
  e->prof_end(ph, st);
#undef DSM_LAUNCH
  HIPCHK(hipGetLastError());
  if (a.norm_out) {
    hipLaunchKernelGGL(row_norm_kernel, dim3(a.M), dim3(256), 0, st, a.norm_out, a.Y, a.norm_w, a.norm_b, a.M,
                       a.N, a.norm_eps, a.norm_rms);
    HIPCHK(hipGetLastError());
  }
  return 0;
}

// Run `body` (a sequence of launches on `st`) eagerly, or — once it has run twice with the same key — capture it into a
// hipGraph and replay that from then on.  key: everything the body's launch arguments depend on that may change between
// calls (caller-supplied pointers, branch selectors).
template <typename F>
int run_captured(dsm_engine* e, dsm_engine::GraphSlot& gs, hipStream_t st, uint64_t key, F&& body) {
  if (!e->use_graphs || e->prof_mask != 0 || gs.disabled) { e->eager_bodies += 1; return body(); }
  if (gs.exec && gs.key == key && gs.ws_gen == e->ws_gen) {
    HIPCHK(hipGraphLaunch(gs.exec, st));
    e->graph_launches += 1;
    return 0;
  }
  if (gs.key != key || gs.ws_gen != e->ws_gen) {  // new arguments: settle again before capturing
    gs.key = key;
    gs.ws_gen = e->ws_gen;
    gs.warm = 0;
    if (gs.exec) { (void)hipGraphExecDestroy(gs.exec); gs.exec = nullptr; }
  }
  if (gs.warm < 2) {
    gs.warm += 1;
    e->eager_bodies += 1;
    const int rc = body();
    if (gs.ws_gen != e->ws_gen) { gs.ws_gen = e->ws_gen; gs.warm = 0; }  // a workspace moved during this run
    return rc;
  }
  // Relaxed mode: the body only launches kernels and async copies on `st`; other threads are kept out by the API lock.
  ApiExclusive alone(e);
  hipError_t hb = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
  if (hb != hipSuccess) {
    (void)hipGetLastError();
    e->note_capture_failure("hipStreamBeginCapture", hb);
    if (++gs.failures >= dsm_engine::kMaxCaptureTries) gs.disabled = true;
    gs.warm = 0;
    e->eager_bodies += 1;
    return body();
  }
  e->capturing = true;
  e->capture_failed = false;
  const int rc = body();
  e->capturing = false;
  hipGraph_t g = nullptr;
  const hipError_t he = hipStreamEndCapture(st, &g);
  hipError_t hi = hipSuccess;
  if (!rc && he == hipSuccess && g) hi = hipGraphInstantiate(&gs.exec, g, nullptr, nullptr, 0);
  if (g) (void)hipGraphDestroy(g);
  if (rc || he != hipSuccess || !g || hi != hipSuccess) {
    // Never silent (r02 swallowed this and stayed eager for good): counted in dsm_metrics.capture_failures with the first
    // reason kept, the body re-run launch by launch so that the step still happens, and the capture tried again after two
    // more settled runs — up to kMaxCaptureTries times, then this sequence stays eager.
    gs.exec = nullptr;
    (void)hipGetLastError();
    e->note_capture_failure(rc ? (e->capture_failed ? "workspace grew during capture" : "launch error during capture")
                               : he != hipSuccess ? "hipStreamEndCapture" : !g ? "hipStreamEndCapture returned no graph" : "hipGraphInstantiate",
                            rc ? hipSuccess : he != hipSuccess ? he : hi);
    if (++gs.failures >= dsm_engine::kMaxCaptureTries) gs.disabled = true;
    gs.warm = 0;
    e->eager_bodies += 1;
    return body();
  }
  HIPCHK(hipGraphLaunch(gs.exec, st));
  e->graph_launches += 1;
  return 0;
}

RowMap plain_map(int M, int ld) {
  RowMap r;
  r.bstride = 0;
  r.rpb = M > 0 ? M : 1;
  r.ld = ld;
  r.toff = 0;
  return r;
}
RowMap batch_map(long bstride, int rpb, int ld, int toff) {
  RowMap r;
  r.bstride = bstride;
  r.rpb = rpb;
  r.ld = ld;
  r.toff = toff;
  return r;
}

GemmArgs base_args(const Linear& L, const float* X, RowMap xmap, int M) {
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.X = X;
  a.xmap = xmap;
  a.W = L.w;
  a.wpacked = L.packed ? 1 : 0;
  a.Kpad = L.Kpad;
  a.K = L.K;
  a.N = L.N;
  a.M = M;
  a.nt_stride = 16;
  a.bias = L.bias;
  return a;
}

template <typename WT>
int gemm_store(dsm_engine* e, hipStream_t st, GemmArgs& a, bool aligned = true) {
  return launch_gemm_t<WT, float, EPI_STORE, 1>(e, st, a, aligned);
}

// conv as a GEMM over the consumer's concat buffer; output goes to Y (raw) and/or Y2 (ELU copy, usually the
// next conv's concat buffer at time offset S_next)
int run_conv(dsm_engine* e, hipStream_t st, const ConvGeom& c, const float* cat, int B, float* y, RowMap ymap,
             float* y2, RowMap y2map, const float* res, RowMap rmap) {
  const long bstride = (long)(c.S + c.T_in) * c.in_c;
  GemmArgs a = base_args(c.lin, cat, batch_map(bstride, c.T_out, c.stride * c.in_c, 0), B * c.T_out);
  a.Y = y;
  a.ymap = ymap;
  a.Y2 = y2;
  a.y2map = y2map;
  a.res = res;
  a.rmap = rmap;
  bool aligned = ((c.stride * c.in_c) % 4 == 0) && (bstride % 4 == 0);
  return gemm_store<float>(e, st, a, aligned);
}

}  // namespace

#include "dsm_engine_api.inc"
#include "dsm_tts.inc"
#include "dsm_audio.inc"
#include "dsm_worker.inc"
