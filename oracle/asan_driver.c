/* asan_driver.c — runs the CPU oracle (oracle/dsm_oracle.c, TEST INFRASTRUCTURE) under AddressSanitizer + UBSan:
 * `make -C oracle asan` builds oracle/_asan/asan_driver from the same sources with -fsanitize=address,undefined.
 * The config structs arrive as raw bytes (written by tests/test_sanitizers_cpu.py from the ctypes mirrors), the weights
 * as the usual synthetic safetensors.  Streams a few frames through encode / step_tokens / decode with mixed masks, a
 * slot reset and a ring wrap, then a few TTS steps; any sanitizer report aborts the process.
 * usage: asan_driver asr_cfg.bin lm.safetensors mimi.safetensors tts_cfg.bin tts_lm.safetensors steps */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dsm_oracle.h"

static int read_all(const char* path, void* dst, size_t n) {
  FILE* f = fopen(path, "rb");
  if (!f) return -1;
  size_t got = fread(dst, 1, n, f);
  fclose(f);
  return got == n ? 0 : -1;
}

int main(int argc, char** argv) {
  if (argc < 7) { fprintf(stderr, "usage: %s asr_cfg.bin lm mimi tts_cfg.bin tts_lm steps\n", argv[0]); return 2; }
  dsm_asr_config cfg;
  dsm_tts_config tcfg;
  if (read_all(argv[1], &cfg, sizeof cfg) || read_all(argv[4], &tcfg, sizeof tcfg)) { fprintf(stderr, "cannot read the config blobs\n"); return 2; }
  const int steps = atoi(argv[6]), B = 3, nq = cfg.mimi.quantizer_n_q;
  char err[512];
  orc_asr* a = orc_asr_create(&cfg, B, argv[2], argv[3], err, sizeof err);
  if (!a) { fprintf(stderr, "create: %s\n", err); return 1; }
  float* pcm = (float*)malloc(sizeof(float) * B * DSM_FRAME_SIZE);
  float* out = (float*)malloc(sizeof(float) * B * DSM_FRAME_SIZE);
  uint32_t* codes = (uint32_t*)calloc((size_t)B * nq, 4);
  uint32_t text[3];
  float prs[DSM_MAX_EXTRA_HEADS * 3];
  dsm_asr_msg msgs[16];
  uint32_t toks[64];
  unsigned long sum = 0;
  for (int s = 0; s < steps; ++s) {
    uint8_t mask[3] = {1, (uint8_t)(s % 3 != 1), (uint8_t)(s % 4 != 2)};
    for (int i = 0; i < B * DSM_FRAME_SIZE; ++i) pcm[i] = 0.1f * sinf(0.01f * (float)(i + 7 * s)) + 0.001f * (float)((i * 31 + s) % 17);
    if (s == steps / 2) { orc_asr_reset_slot(a, 1); orc_mimi_reset_slot(a, 0, 1); }
    if (orc_mimi_encode_step(a, 0, pcm, mask, codes) != 1) { fprintf(stderr, "encode produced no frame\n"); return 1; }
    if (orc_asr_step_tokens(a, codes, mask, text, cfg.extra_heads_num ? prs : NULL)) { fprintf(stderr, "step_tokens failed\n"); return 1; }
    for (int b = 0; b < B; ++b) if (!mask[b]) for (int k = 0; k < nq; ++k) codes[b * nq + k] = 0;
    if (orc_mimi_decode_step(a, 0, codes, mask, out) <= 0) { fprintf(stderr, "decode produced nothing\n"); return 1; }
    int n = orc_asr_poll_msgs(a, msgs, 16, toks, 64);
    for (int b = 0; b < B; ++b) sum = sum * 31 + (mask[b] ? text[b] + codes[b * nq] : 0) + (unsigned long)n;
  }
  orc_asr_destroy(a);
  orc_tts* t = orc_tts_create(&tcfg, 2, argv[5], err, sizeof err);
  if (!t) { fprintf(stderr, "tts create: %s\n", err); return 1; }
  const int S = tcfg.dep_num_slices;
  uint32_t* audio = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)S);
  for (int s = 0; s < steps; ++s) {
    uint32_t prev[2] = {(uint32_t)(5 + s), 7}, tt[2];
    int32_t allowed[2] = {9 + s, DSM_TTS_ALLOW_PAD_OR_EPAD};
    uint8_t mask[2] = {1, (uint8_t)(s != 2)};
    if (orc_tts_step(t, prev, allowed, mask, tt, audio)) { fprintf(stderr, "tts step failed\n"); return 1; }
    sum = sum * 31 + tt[0] + audio[0];
  }
  orc_tts_destroy(t);
  free(pcm); free(out); free(codes); free(audio);
  printf("oracle asan ok: %d frames, checksum %lu\n", steps, sum);
  return 0;
}
