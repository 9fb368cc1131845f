"""Parity at the BENCHMARKED batch shapes: real stt-1b-en_fr dimensions with the batch sizes whose
launch configurations the small-batch tests never reach —

  B = 64   (bench.py's default line): two LM stream groups of 32 slots, bf16 `gemm_tile_kernel<.., MT = 2, ..>`, the
           attention kernel's fused QKV-reduce prologue over 8 split-K slabs;
  B = 130, one stream group (DSM_LM_GROUPS=1): bf16 MT = 4 m-tiles, the in-workgroup chunk loop at its natural
           threshold (gate GEMM: 176 n-tiles x 3 m-blocks >= 384), the LDS-padded attention launch (130 x 16 heads >= 2048
           workgroups), ragged last m-tile (130 = 8 x 16 + 2);
  B = 130, default two groups (80 + 50 slots): MT = 4 with split-K slabs.

Codes are fed directly (random, valid) so that the CPU oracle only runs the LM: one oracle pass at B = 130 is the
reference for all three engines — rows never interact, so slots [0, 64) of the B = 130 oracle are the B = 64 answer
(tests/test_sharding_cpu.py checks that independence on the CPU).  Hidden states, logits, VAD probabilities bit for bit."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WEIGHTS_DIR = os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights")


def _lm_steps(obj, cfg, B, codes, masks):
    out = []
    for s in range(len(codes)):
        t, p = obj.step_tokens(codes[s][:B], masks[s][:B])
        hid = obj.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1).copy()
        lg = obj.debug_read("lm.logits", B * cfg.text_out_vocab_size).reshape(B, -1).copy()
        out.append((t.copy(), p.copy(), hid, lg))
    return out


def _compare(got, want, masks, B, label):
    for s, (g, w) in enumerate(zip(got, want)):
        act = masks[s][:B].astype(bool)
        assert np.array_equal(g[2][act].view(np.uint32), w[2][:B][act].view(np.uint32)), f"{label}: lm.hidden differs at step {s}"
        assert np.array_equal(g[3][act].view(np.uint32), w[3][:B][act].view(np.uint32)), f"{label}: logits differ at step {s}"
        assert np.array_equal(g[0][act], w[0][:B][act]), f"{label}: text tokens differ at step {s}"
        assert np.array_equal(g[1][:, act].view(np.uint32), w[1][:, :B][:, act].view(np.uint32)), f"{label}: VAD differs at step {s}"


def test_stt_1b_batch_64_and_130(gpu, dsm, lib, orc, monkeypatch):
    from dsm_amd import synth
    cfg = dsm.config_stt_1b_en_fr()
    cfg.dot_mode = 0  # the preset says 1; this file runs mode 0 (mode 1 at these shapes: test_bx3_gpu.py)
    lm, mimi = synth.make_synth_weights(cfg, WEIGHTS_DIR, tag="stt-1b-en_fr")
    BO, steps = 130, 2
    rng = np.random.default_rng(5)
    codes = rng.integers(0, cfg.mimi.quantizer_bins, (steps, BO, cfg.audio_codebooks)).astype(np.uint32)
    masks = np.ones((steps, BO), dtype=np.uint8)
    masks[1, [3, 40, 77, 129]] = 0  # second step: a paused slot in every 16-row tile class (first group, second group, ragged tile)
    ora = orc.OracleAsr(cfg, BO, lm, mimi)
    want = _lm_steps(ora, cfg, BO, codes, masks)
    ora.close()
    for B, groups_env, want_groups in ((64, None, [(0, 32), (32, 32)]), (130, "1", [(0, 130)]), (130, None, [(0, 80), (80, 50)])):
        if groups_env:
            monkeypatch.setenv("DSM_LM_GROUPS", groups_env)
        else:
            monkeypatch.delenv("DSM_LM_GROUPS", raising=False)
        eng = dsm.AsrEngine(cfg, B, lm, mimi)
        assert eng.stream_groups() == want_groups
        got = _lm_steps(eng, cfg, B, codes, masks)
        eng.close()
        _compare(got, want, masks, B, f"B={B} groups={len(want_groups)}")
