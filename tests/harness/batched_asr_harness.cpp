// batched_asr_harness.cpp — the calls srv/batched_asr.rs makes, from C++ through the C ABI only.
//
// Reference structure (server/rust/moshi/moshi-server/src/batched_asr.rs:243-524): an encoder_loop thread calls
// Mimi::encode_step and pushes PipelineMsg{audio_tokens, mask, resets} into a sync_channel(100); a model_loop thread
// applies the resets, calls asr::State::step_tokens and forwards the AsrMsgs.  This harness runs the same schedule
// (a) sequentially on one thread and (b) with the two threads + a bounded queue on a second engine, and checks that
// every code / token / message is identical — i.e. that the two sides of the ABI can be driven concurrently.
//
// usage: harness <lm.safetensors> <mimi.safetensors> <batch> <frames>   (tiny configuration, see dsm_amd.config_tiny)
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/dsm.h"

static void tiny_config(dsm_asr_config* c) {  // mirrors dsm_amd.config_tiny()
  memset(c, 0, sizeof *c);
  c->lm = {128, 4, 2, 512, 12, 100000, 1, 1, 1, 0, 0};
  c->text_in_vocab_size = 65; c->text_out_vocab_size = 64; c->audio_vocab_size = 33; c->audio_codebooks = 4;
  c->extra_heads_num = 2; c->extra_heads_dim = 6; c->asr_delay_in_tokens = 2; c->temperature = 0.f; c->kv_bf16 = 1;
  dsm_mimi_config* m = &c->mimi;
  m->channels = 1; m->dimension = 64; m->n_filters = 4; m->n_residual_layers = 1; m->n_ratios = 4;
  m->ratios[0] = 8; m->ratios[1] = 6; m->ratios[2] = 5; m->ratios[3] = 4;
  m->kernel_size = 7; m->residual_kernel_size = 3; m->last_kernel_size = 3; m->dilation_base = 2; m->compress = 2;
  m->transformer = {64, 2, 2, 128, 10, 10000, 0, 0, 1, 1, 1};
  m->quantizer_n_q = 4; m->quantizer_bins = 32; m->quantizer_dim = 16; m->downsample_stride = 2;
}

struct Frame {
  std::vector<uint32_t> codes, text;
  std::vector<float> prs;
  std::vector<dsm_asr_msg> msgs;
  std::vector<uint32_t> msg_tokens;
};

struct PipelineMsg {  // batched_asr.rs:281-288
  std::vector<uint32_t> audio_tokens;
  std::vector<uint8_t> mask;
  std::vector<int> resets;
  int step;
};

#define CHECK(x) do { int rc_ = (x); if (rc_ < 0) { fprintf(stderr, "%s failed: %d %s\n", #x, rc_, dsm_last_error(e)); exit(2); } } while (0)

static void make_inputs(int B, int frames, std::vector<std::vector<float>>& pcm, std::vector<std::vector<uint8_t>>& masks,
                        std::vector<std::vector<int>>& resets) {
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  pcm.assign(frames, std::vector<float>((size_t)B * DSM_FRAME_SIZE));
  masks.assign(frames, std::vector<uint8_t>(B, 1));
  resets.assign(frames, {});
  for (int f = 0; f < frames; ++f) {
    for (int b = 0; b < B; ++b) {
      for (int i = 0; i < DSM_FRAME_SIZE; ++i)
        pcm[f][(size_t)b * DSM_FRAME_SIZE + i] =
            0.1f * (float)sin(2 * M_PI * (110 + 7 * b) * (f * DSM_FRAME_SIZE + i) / 24000.0) + 0.02f * (float)(2 * rnd() - 1);
      masks[f][b] = (b == 0) ? 1 : (rnd() < 0.75);
    }
    if (f % 7 == 5) resets[f].push_back((f / 7) % B);  // a client leaves, the slot is recycled
  }
}

static Frame model_step(dsm_engine* e, int B, int n_q, int nh, const uint32_t* codes, const uint8_t* mask) {
  Frame fr;
  fr.codes.assign(codes, codes + (size_t)B * n_q);
  fr.text.resize(B);
  fr.prs.resize((size_t)nh * B);
  CHECK(dsm_asr_step_tokens(e, codes, mask, fr.text.data(), fr.prs.data()));
  fr.msgs.resize(256);
  fr.msg_tokens.resize(1024);
  int n = dsm_asr_poll_msgs(e, fr.msgs.data(), 256, fr.msg_tokens.data(), 1024);
  fr.msgs.resize(n < 0 ? 0 : n);
  return fr;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s lm mimi batch frames\n", argv[0]); return 1; }
  const int B = atoi(argv[3]), frames = atoi(argv[4]);
  dsm_asr_config cfg;
  tiny_config(&cfg);
  const int n_q = cfg.mimi.quantizer_n_q, nh = cfg.extra_heads_num;
  std::vector<std::vector<float>> pcm;
  std::vector<std::vector<uint8_t>> masks;
  std::vector<std::vector<int>> resets;
  make_inputs(B, frames, pcm, masks, resets);

  // ---- (a) sequential reference run ----
  std::vector<Frame> ref;
  {
    dsm_engine* e = nullptr;
    if (dsm_asr_create(&cfg, 0, B, argv[1], argv[2], &e)) { fprintf(stderr, "create: %s\n", dsm_last_error(nullptr)); return 2; }
    for (int f = 0; f < frames; ++f) {
      for (int slot : resets[f]) { CHECK(dsm_asr_reset_slot(e, slot)); CHECK(dsm_mimi_reset_slot(e, slot)); }
      std::vector<uint32_t> codes((size_t)B * n_q);
      int produced = 0;
      CHECK(dsm_mimi_encode_step(e, pcm[f].data(), masks[f].data(), codes.data(), &produced));
      for (int b = 0; b < B; ++b)
        if (!masks[f][b]) for (int i = 0; i < n_q; ++i) codes[(size_t)b * n_q + i] = 0;  // unspecified for inactive slots
      ref.push_back(model_step(e, B, n_q, nh, codes.data(), masks[f].data()));
    }
    dsm_destroy(e);
  }

  // ---- (b) encoder thread -> bounded queue -> model thread ----
  std::vector<Frame> got(frames);
  {
    dsm_engine* e = nullptr;
    if (dsm_asr_create(&cfg, 0, B, argv[1], argv[2], &e)) { fprintf(stderr, "create: %s\n", dsm_last_error(nullptr)); return 2; }
    std::deque<PipelineMsg> q;
    std::mutex mu;
    std::condition_variable cv;
    const size_t cap = 4;  // small on purpose: both threads are in flight together most of the time
    bool done = false;
    std::thread encoder([&]() {
      for (int f = 0; f < frames; ++f) {
        // the encoder-side Mimi is reset by the encoder thread itself (the reference never resets it at all)
        for (int slot : resets[f]) CHECK(dsm_mimi_reset_slot(e, slot));
        PipelineMsg m;
        m.audio_tokens.resize((size_t)B * n_q);
        m.mask = masks[f];
        m.resets = resets[f];
        m.step = f;
        int produced = 0;
        CHECK(dsm_mimi_encode_step(e, pcm[f].data(), masks[f].data(), m.audio_tokens.data(), &produced));
        for (int b = 0; b < B; ++b)
          if (!masks[f][b]) for (int i = 0; i < n_q; ++i) m.audio_tokens[(size_t)b * n_q + i] = 0;
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return q.size() < cap; });
        q.push_back(std::move(m));
        cv.notify_all();
      }
      std::unique_lock<std::mutex> lk(mu);
      done = true;
      cv.notify_all();
    });
    std::thread model([&]() {
      for (;;) {
        PipelineMsg m;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return !q.empty() || done; });
          if (q.empty()) return;
          m = std::move(q.front());
          q.pop_front();
          cv.notify_all();
        }
        for (int slot : m.resets) CHECK(dsm_asr_reset_slot(e, slot));  // batched_asr.rs:467-471
        got[m.step] = model_step(e, B, n_q, nh, m.audio_tokens.data(), m.mask.data());
      }
    });
    encoder.join();
    model.join();
    // the launch-bound sequences must have been captured and replayed as hipGraphs with both threads inside the library, and no
    // capture may have failed on the way (the old silent fallback to eager launches would have passed unnoticed)
    dsm_metrics mt;
    CHECK(dsm_get_metrics(e, &mt));
    printf("graphs: %llu replays, %llu eager bodies, %llu capture failures%s%s\n", (unsigned long long)mt.graph_launches,
           (unsigned long long)mt.eager_bodies, (unsigned long long)mt.capture_failures, mt.capture_failures ? ": " : "", mt.capture_error);
    const char* genv = getenv("DSM_GRAPHS");
    const bool graphs_on = !(genv && atoi(genv) == 0);
    if (mt.capture_failures != 0 || (graphs_on && frames >= 8 && mt.graph_launches == 0)) {
      fprintf(stderr, "graph capture did not hold under two threads\n");
      return 4;
    }
    dsm_destroy(e);
  }

  // ---- compare ----
  int bad = 0;
  for (int f = 0; f < frames; ++f) {
    const Frame &a = ref[f], &b = got[f];
    bool same = a.codes == b.codes && a.msgs.size() == b.msgs.size();
    for (int s = 0; s < B && same; ++s)
      if (masks[f][s]) {
        same = a.text[s] == b.text[s];
        for (int h = 0; h < nh && same; ++h) same = memcmp(&a.prs[(size_t)h * B + s], &b.prs[(size_t)h * B + s], 4) == 0;
      }
    for (size_t i = 0; i < a.msgs.size() && same; ++i)
      same = a.msgs[i].kind == b.msgs[i].kind && a.msgs[i].batch_idx == b.msgs[i].batch_idx && a.msgs[i].time == b.msgs[i].time &&
             a.msgs[i].n_tokens == b.msgs[i].n_tokens;
    if (!same) { fprintf(stderr, "frame %d differs between the sequential and the threaded run\n", f); ++bad; }
  }
  if (bad) return 3;
  printf("harness ok: %d frames x %d slots identical (sequential vs encoder/model threads)\n", frames, B);
  return 0;
}
