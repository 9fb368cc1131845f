"""Shared driver for the TTS tests: a deterministic schedule of (prev_text_token, allowed, mask, resets) that walks
every branch of tts_streaming::State::step — Text / Pad / PadOrEpad, the forced end-of-pad after
max_consecutive_pads, both delay windows, inactive slots and a mid-run slot reset."""
import numpy as np

from dsm_amd import TTS_ALLOW_PAD, TTS_ALLOW_PAD_OR_EPAD


def schedule(cfg, B, steps, seed=11):
    rng = np.random.default_rng(seed)
    out = []
    for s in range(steps):
        prev = rng.integers(0, cfg.text_in_vocab_size, B).astype(np.uint32)
        allowed = np.empty(B, dtype=np.int32)
        for b in range(B):
            kind = (b + (s // 3)) % 3 if b else 0
            if b == 1:  # long pad run, then PadOrEpad: exercises consecutive_pads > max_consecutive_pads
                kind = 1 if s < cfg.max_consecutive_pads + 3 else 2
            allowed[b] = int(rng.integers(0, cfg.text_in_vocab_size)) if kind == 0 else (
                TTS_ALLOW_PAD if kind == 1 else TTS_ALLOW_PAD_OR_EPAD)
        mask = (rng.random(B) < 0.8).astype(np.uint8)
        mask[0] = 1
        if B > 1:
            mask[1] = 1
        out.append((prev, allowed, mask))
    return out


def run(engine, cfg, B, steps, resets=None, seed=11):
    """Returns per-step (text_tokens, audio) plus the final audio_tokens table of every slot."""
    resets = resets or {}
    trace = []
    for s, (prev, allowed, mask) in enumerate(schedule(cfg, B, steps, seed)):
        for slot in resets.get(s, []):
            engine.reset_batch_idx(slot)
        text, audio = engine.step(prev, allowed, mask)
        act = mask.astype(bool)
        trace.append((np.where(act, text, 0).copy(), np.where(act[:, None], audio, 0).copy()))
    tables = [[engine.audio_tokens(b, i).copy() for i in range(engine.step_idx(b))] for b in range(B)]
    return trace, tables
