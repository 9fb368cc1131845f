import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, dsm_amd as dsm, oracle as orc
from dsm_amd import synth
cfg = dsm.config_tts_v202501(); cfg.text_audio_delay_in_tokens, cfg.max_steps = 3, 64
path = synth.make_synth_tts_weights(cfg, "/tmp/dsm_weights", tag="tts-v202501")
B = 2
eng = dsm.TtsEngine(cfg, B, path); ora = orc.OracleTts(cfg, B, path)
rng = np.random.default_rng(2)
for s in range(8):
    prev = rng.integers(0, cfg.text_in_vocab_size, B).astype(np.uint32)
    allowed = np.array([int(rng.integers(4, 8000)), dsm.TTS_ALLOW_PAD_OR_EPAD], dtype=np.int32)
    mask = np.array([1, 0 if s == 4 else 1], dtype=np.uint8)
    te, ae = eng.step(prev, allowed, mask); to, ao = ora.step(prev, allowed, mask)
    d = (ae != ao)
    print(s, "text", te, to, "diff slots/slices:", [(b, np.nonzero(d[b])[0][:6].tolist()) for b in range(B) if mask[b] and d[b].any()], flush=True)
    if d[mask.astype(bool)].any():
        b = [b for b in range(B) if mask[b] and d[b].any()][0]
        k = int(np.nonzero(d[b])[0][0])
        print(" first diff slot", b, "slice", k, "eng", ae[b][k-1:k+3], "ora", ao[b][k-1:k+3])
