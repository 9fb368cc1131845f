"""hipGraph replay of the launch-bound inner loops: the captured sequences (Mimi encode, Mimi
decode, each LM stream group's transformer + heads, the TTS step) must give the bits the eager launches give — over
masks, slot resets and ring wrap, i.e. with every per-step variable living in device buffers — and must actually be in
use (dsm_metrics.graph_launches)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_asr(dsm, cfg, B, lm, mimi, steps, masks, resets, pcm):
    eng = dsm.AsrEngine(cfg, B, lm, mimi)
    rng = np.random.default_rng(1)
    log = []
    for s in range(steps):
        for slot in resets.get(s, []):
            eng.reset_batch_idx(slot)
            eng.mimi_reset_batch_idx(slot)
        codes = eng.encode_step(pcm[s], masks[s])
        text, prs = eng.step_tokens(None, masks[s])  # device-resident codes of the encode above
        hid = eng.debug_read("lm.hidden", B * cfg.lm.d_model)
        dec = eng.decode_step(rng.integers(0, cfg.mimi.quantizer_bins, (B, cfg.mimi.quantizer_n_q)).astype(np.uint32), masks[s])
        log.append((codes.copy(), text.copy(), prs.copy(), hid.copy(), dec.copy(), eng.poll_msgs()))
    m = eng.metrics()
    eng.close()
    assert m.capture_failures == 0, m.capture_error  # a capture that does not hold is never silent (dsm_metrics, r03)
    return log, (m.graph_launches, m.eager_bodies)


def test_graph_replay_equals_eager_asr(gpu, dsm, lib, tiny_weights, monkeypatch):
    from dsm_amd import synth
    cfg = dsm.config_tiny()
    B, steps = 35, 30  # 35 slots: two LM stream groups (32 + 3), each with its own graph
    rng = np.random.default_rng(3)
    masks = (rng.random((steps, B)) < 0.75).astype(np.uint8)
    masks[:, 0] = 1
    resets = {7: [1], 15: [0, 33], 16: [33]}
    pcm = synth.synth_pcm(B, steps)
    monkeypatch.setenv("DSM_GRAPHS", "0")
    eager, (g0, e0) = _run_asr(dsm, cfg, B, *tiny_weights, steps, masks, resets, pcm)
    monkeypatch.delenv("DSM_GRAPHS")
    graph, (g1, e1) = _run_asr(dsm, cfg, B, *tiny_weights, steps, masks, resets, pcm)
    groups = int(os.environ.get("DSM_LM_GROUPS", "2"))  # tools/knob_parity_sweep.sh runs this file with one group too
    assert g0 == 0 and e0 == (2 + groups) * steps  # encode, decode and the LM groups of a step, all eager
    # two settling runs per sequence, plus a re-settle whenever a split-K workspace still grew (first decode call): then replay
    n = 2 + groups
    assert g1 + e1 == n * steps and 2 * n <= e1 <= 4 * n and g1 >= n * (steps - 4), (g1, e1)
    for s, (a, b) in enumerate(zip(eager, graph)):
        act = masks[s].astype(bool)
        assert np.array_equal(a[0][act], b[0][act]), f"codes differ at step {s}"
        assert np.array_equal(a[1][act], b[1][act]), f"text tokens differ at step {s}"
        assert np.array_equal(a[2][:, act].view(np.uint32), b[2][:, act].view(np.uint32)), f"VAD differs at step {s}"
        assert np.array_equal(a[3].reshape(B, -1)[act].view(np.uint32), b[3].reshape(B, -1)[act].view(np.uint32)), f"hidden differs at step {s}"
        assert np.array_equal(a[4][act].view(np.uint32), b[4][act].view(np.uint32)), f"decoded PCM differs at step {s}"
        assert a[5] == b[5], f"AsrMsg lists differ at step {s}"


def test_graph_replay_equals_eager_tts(gpu, dsm, lib, monkeypatch):
    from dsm_amd import synth
    from tts_schedule import schedule
    cfg = dsm.config_tts_tiny()
    path = synth.make_synth_tts_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tts_tiny")
    B, steps = 4, 14

    def run():
        eng = dsm.TtsEngine(cfg, B, path)
        out = [tuple(x.copy() for x in eng.step(prev, allowed, mask)) for prev, allowed, mask in schedule(cfg, B, steps)]
        m = eng.metrics()
        eng.close()
        assert m.capture_failures == 0, m.capture_error
        return out, m.graph_launches, m.eager_bodies

    monkeypatch.setenv("DSM_TTS_GROUPS", "1")
    monkeypatch.setenv("DSM_GRAPHS", "0")
    eager, g0, e0 = run()
    monkeypatch.delenv("DSM_GRAPHS")
    graph, g1, e1 = run()
    assert g0 == 0 and e0 == steps
    # two variants (with / without the depformer), two settling runs each, plus a re-settle when a split-K workspace still grew
    assert g1 >= steps // 2 and g1 + e1 == steps and e1 <= 7, (g1, e1)
    # two stream groups of two slots (r03): each group has its own stream, split-K workspace and pair of graphs
    monkeypatch.setenv("DSM_TTS_GROUPS", "2")
    grouped, g2, e2 = run()
    assert g2 + e2 == 2 * steps and g2 >= steps, (g2, e2)
    sched = list(schedule(cfg, B, steps))
    for s, (a, b, c) in enumerate(zip(eager, graph, grouped)):
        act = np.asarray(sched[s][2]).astype(bool)
        assert np.array_equal(a[0][act], b[0][act]) and np.array_equal(a[1][act], b[1][act]), f"TTS tokens differ at step {s}"
        assert np.array_equal(a[0][act], c[0][act]) and np.array_equal(a[1][act], c[1][act]), f"TTS tokens of the two-group engine differ at step {s}"
