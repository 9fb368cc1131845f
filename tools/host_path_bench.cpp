// host_path_bench.cpp — the end-to-end frame loop of the reference worker (srv/batched_asr.rs:314-522,856-987) through the
// C ABI with HOST audio: what bench.py's device-resident capacity legs leave out.
//
// Per 80 ms frame, as the server does it:
//   socket tasks (here F feeder threads)   one msgpack InMsg::Audio of 1920 samples per open channel -> dsm_worker_send
//                                           (msgpack decode in the caller's thread, then the channel queue)
//   encoder thread                          dsm_worker_step_encode: pre_process (cut a frame per channel), pinned staging,
//                                           H2D of B x 7.7 KB, Mimi encode started; run-ahead up to three frames
//   model thread                            dsm_worker_step_model: LM step of the oldest frame (synchronises), post_process
//                                           (one OutMsg::Step per channel + words), then dsm_worker_recv drains every channel
// Free-running (no 80 ms pacing): the achieved frames per second against 12.5 is the real-time factor of the WHOLE path;
// the per-thread busy times say which side is the limit.  Ring caches are jumped to the steady state first
// (dsm_debug_set_positions), like the device-resident legs.
//
// usage: host_path_bench <lm.safetensors> <mimi.safetensors> <batch> <frames> [feeder threads = 8] [dot_mode = 1]   (stt-1b-en_fr)
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/dsm.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s lm mimi batch frames [feeders]\n", argv[0]); return 1; }
  const int B = atoi(argv[3]), frames = atoi(argv[4]), F = argc > 5 ? atoi(argv[5]) : 8;
  dsm_asr_config cfg;
  dsm_asr_config_stt_1b_en_fr(&cfg);
  cfg.dot_mode = argc > 6 ? atoi(argv[6]) : 1;
  dsm_engine* e = nullptr;
  if (dsm_asr_create(&cfg, 0, B, argv[1], argv[2], &e)) { fprintf(stderr, "create: %s\n", dsm_last_error(nullptr)); return 2; }
  dsm_worker* w = nullptr;
  if (dsm_worker_create(e, &w)) { fprintf(stderr, "worker create failed\n"); return 2; }
  std::vector<int> slots;
  for (int b = 0; b < B; ++b) {
    uint64_t cid;
    const int s = dsm_worker_open(w, &cid);
    if (s < 0) { fprintf(stderr, "open failed\n"); return 2; }
    slots.push_back(s);
  }
  // four distinct frames of audio, encoded once as the wire message a client sends (srv/asr.rs:15-22)
  std::vector<std::vector<uint8_t>> wire(4);
  for (int k = 0; k < 4; ++k) {
    std::vector<float> pcm(DSM_FRAME_SIZE);
    for (int i = 0; i < DSM_FRAME_SIZE; ++i) pcm[i] = 0.1f * (float)sin(2 * M_PI * (110 + 13 * k) * (k * DSM_FRAME_SIZE + i) / 24000.0);
    dsm_in_msg m;
    memset(&m, 0, sizeof m);
    m.kind = DSM_IN_AUDIO; m.pcm = pcm.data(); m.n_pcm = pcm.size();
    const int n = dsm_inmsg_encode(&m, nullptr, 0);
    wire[k].resize((size_t)n);
    dsm_inmsg_encode(&m, wire[k].data(), wire[k].size());
  }
  // Init -> Ready + slot resets, then the steady state of a long-running server
  if (dsm_worker_step(w) < 0) { fprintf(stderr, "first step: %s\n", dsm_worker_last_error(w)); return 2; }
  if (dsm_debug_set_positions(e, 3000, 1000)) { fprintf(stderr, "set_positions failed\n"); return 2; }

  std::atomic<int> fed{0}, encoded{0}, stepped{0};
  std::atomic<bool> failed{false};
  double t_feed = 0, t_enc = 0, t_model = 0, t_recv = 0;
  const int warm = 3, total = frames + warm;
  double t_start = 0;
  std::thread feeder([&] {
    // F socket tasks per frame; the frame's messages arrive while the previous frames are in flight (at most 3 ahead of the model)
    for (int f = 0; f < total && !failed; ++f) {
      while (f - stepped.load() > 3 && !failed) std::this_thread::yield();
      const double t0 = now_ms();
      std::vector<std::thread> th;
      for (int k = 0; k < F; ++k)
        th.emplace_back([&, k] {
          for (int b = k; b < B; b += F)
            if (dsm_worker_send(w, slots[b], wire[(f + b) & 3].data(), wire[(f + b) & 3].size()) != 0) failed = true;
        });
      for (auto& t : th) t.join();
      if (f >= warm) t_feed += now_ms() - t0;
      fed = f + 1;
    }
  });
  std::thread encoder([&] {
    for (int f = 0; f < total && !failed; ++f) {
      while (fed.load() <= f && !failed) std::this_thread::yield();
      double busy = 0;
      for (;;) {
        const double t0 = now_ms();
        const int rc = dsm_worker_step_encode(w);
        if (rc < 0) { fprintf(stderr, "step_encode: %s\n", dsm_worker_last_error(w)); failed = true; break; }
        if (rc == 1) { busy = now_ms() - t0; break; }  // the call that cut and started the frame; "queue full" polls are waiting, not work
        std::this_thread::yield();
        if (failed) break;
      }
      if (f >= warm) t_enc += busy;
      encoded = f + 1;
    }
  });
  std::thread model([&] {
    std::vector<uint8_t> buf(1 << 16);
    for (int f = 0; f < total && !failed; ++f) {
      if (f == warm) t_start = now_ms();
      while (encoded.load() <= f && !failed) std::this_thread::yield();
      const double t0 = now_ms();
      const int rc = dsm_worker_step_model(w);
      if (rc < 0) { fprintf(stderr, "step_model: %s\n", dsm_worker_last_error(w)); failed = true; break; }
      const double t1 = now_ms();
      size_t len;
      for (int b = 0; b < B; ++b)  // send_loop of every socket
        while (dsm_worker_recv(w, slots[b], buf.data(), buf.size(), &len) == 1) {}
      const double t2 = now_ms();
      if (f >= warm) { t_model += t1 - t0; t_recv += t2 - t1; }
      stepped = f + 1;
    }
  });
  feeder.join(); encoder.join(); model.join();
  const double wall = now_ms() - t_start;
  if (failed) return 3;
  dsm_metrics mt;
  dsm_get_metrics(e, &mt);
  const double ms_frame = wall / frames;
  printf("{\"batch\": %d, \"dot_mode\": %d, \"frames\": %d, \"feeder_threads\": %d, \"ms_per_frame\": %.3f, \"rtf\": %.3f, "
         "\"feed_ms_per_frame\": %.3f, \"encoder_thread_ms_per_frame\": %.3f, \"model_thread_ms_per_frame\": %.3f, "
         "\"recv_ms_per_frame\": %.3f, \"wire_bytes_per_frame\": %zu, \"graph_launches\": %llu, \"capture_failures\": %llu}\n",
         B, cfg.dot_mode, frames, F, ms_frame, 80.0 / ms_frame, t_feed / frames, t_enc / frames, t_model / frames, t_recv / frames,
         wire[0].size() * (size_t)B, (unsigned long long)mt.graph_launches, (unsigned long long)mt.capture_failures);
  dsm_worker_destroy(w);
  dsm_destroy(e);
  return 0;
}
