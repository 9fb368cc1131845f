"""Two engines, same inputs, many steps: every output must agree bit for bit (a race in a split-K workspace, a deferred
reduce or a graph replay would show as a rare difference).  python tools/soak_determinism.py [asr_steps] [tts_steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsm_amd
from dsm_amd import synth

W = os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights")
asr_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
tts_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
for mode in (1, 0):
    cfg = dsm_amd.config_stt_1b_en_fr()
    cfg.dot_mode = mode
    lm, mimi = synth.make_synth_weights(cfg, W, tag="stt-1b-en_fr")
    B = 64
    a = dsm_amd.AsrEngine(cfg, B, lm, mimi)
    b = dsm_amd.AsrEngine(cfg, B, lm, mimi, arena=a.weight_arena())
    pcm = synth.synth_pcm(B, 16, seed=5)
    rng = np.random.default_rng(1)
    for s in range(asr_steps):
        mask = (rng.random(B) < 0.95).astype(np.uint8)
        ca, ta, pa = a.step_pcm(pcm[s % 16], mask)
        cb, tb, pb = b.step_pcm(pcm[s % 16], mask)
        act = mask.astype(bool)
        assert np.array_equal(ca[act], cb[act]) and np.array_equal(ta[act], tb[act]) and np.array_equal(pa[:, act].view(np.uint32), pb[:, act].view(np.uint32)), f"ASR mode {mode} step {s}"
    ma, mb = a.metrics(), b.metrics()
    print(f"ASR dot_mode {mode}: {asr_steps} steps x {B} slots identical; graph launches {ma.graph_launches}/{mb.graph_launches}, capture failures {ma.capture_failures}/{mb.capture_failures}")
    assert ma.capture_failures == 0 and mb.capture_failures == 0
    b.close(); a.close()
for mode in (1, 0):
    cfg = dsm_amd.config_tts_v202501()
    cfg.dot_mode = mode
    path = synth.make_synth_tts_weights(cfg, W, tag="tts-v202501")
    B = 32
    a, b = dsm_amd.TtsEngine(cfg, B, path), dsm_amd.TtsEngine(cfg, B, path)
    for x in (a, b):
        x.set_sampling(3, 40, 0.7, 11)
    rng = np.random.default_rng(2)
    mask = np.ones(B, np.uint8)
    for s in range(tts_steps):
        prev = rng.integers(4, cfg.text_in_vocab_size - 1, B).astype(np.uint32)
        allowed = rng.integers(4, cfg.text_in_vocab_size - 1, B).astype(np.int32)
        ta, aa = a.step(prev, allowed, mask)
        tb, ab = b.step(prev, allowed, mask)
        assert np.array_equal(ta, tb) and np.array_equal(aa, ab), f"TTS mode {mode} step {s}"
    print(f"TTS dot_mode {mode}: {tts_steps} steps x {B} slots identical; capture failures {a.metrics().capture_failures}/{b.metrics().capture_failures}")
    a.close(); b.close()
print("soak ok")
