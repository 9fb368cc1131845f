#!/usr/bin/env python3
"""Writes tests/golden/tiny_asr.json: outputs of the CPU oracle on the tiny configuration with seeded
synthetic weights / PCM, a mixed mask schedule and slot resets.  The reference (Rust + Candle) cannot be
run offline and ships no golden model outputs, so these vectors pin the ORACLE (against drift) and the
HIP engine (against the oracle); they do not pin either against Candle ("parity unpinned")."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

STEPS, B = 24, 4
RESETS = {7: [1], 15: [0, 2]}


def schedule():
    rng = np.random.default_rng(11)
    m = (rng.random((STEPS, B)) < 0.75).astype(np.uint8)
    m[:, 3] = 1
    return m


def run(make_engine, mimi_reset, dot_mode=0):
    import dsm_amd
    from dsm_amd import synth
    cfg = dsm_amd.config_tiny()
    cfg.dot_mode = dot_mode  # 1: tests/golden/tiny_asr_dot_mode1.json (the bf16-matrix-instruction order)
    lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tiny")
    eng = make_engine(cfg, B, lm, mimi)
    pcm = synth.synth_pcm(B, STEPS)
    masks = schedule()
    out = {"steps": STEPS, "batch": B, "resets": {str(k): v for k, v in RESETS.items()}, "masks": masks.tolist(),
           "codes": [], "text": [], "prs_bits": [], "msgs": []}
    for s in range(STEPS):
        for slot in RESETS.get(s, []):
            eng.reset_batch_idx(slot)
            mimi_reset(eng, slot)
        codes = eng.encode_step(pcm[s], masks[s])
        act = masks[s].astype(bool)
        codes = np.where(act[:, None], codes, 0)  # inactive slots: unspecified -> zeroed before they are fed back
        text, prs = eng.step_tokens(codes, masks[s])
        out["codes"].append(codes.tolist())
        out["text"].append(np.where(act, text, 0).tolist())
        out["prs_bits"].append(np.where(act[None, :], prs.view(np.uint32), 0).tolist())
        out["msgs"].append([list(m) for m in eng.poll_msgs()])
    return out


def run_tts():
    """tests/golden/tiny_tts.json: oracle trace of tests/tts_schedule.run on config_tts_tiny."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dsm_amd
    import oracle
    from dsm_amd import synth
    from tts_schedule import run as tts_run
    cfg = dsm_amd.config_tts_tiny()
    path = synth.make_synth_tts_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tts_tiny")
    b, steps, resets = 4, 20, {8: [1], 13: [0, 3]}
    o = oracle.OracleTts(cfg, b, path)
    trace, tables = tts_run(o, cfg, b, steps, resets=resets)
    gold = {"B": b, "steps": steps, "resets": {str(k): v for k, v in resets.items()},
            "text": [t.tolist() for t, _ in trace], "audio": [a.tolist() for _, a in trace],
            "tables": [[r.tolist() for r in tab] for tab in tables]}
    out = os.path.join(ROOT, "tests", "golden", "tiny_tts.json")
    with open(out, "w") as f:
        json.dump(gold, f, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    import oracle
    if "--tts" in sys.argv:
        run_tts()
        sys.exit(0)
    mode = 1 if "--dot-mode-1" in sys.argv else 0
    res = run(oracle.OracleAsr, lambda e, slot: e.mimi_reset_batch_idx(slot, side=0), dot_mode=mode)
    path = os.path.join(ROOT, "tests", "golden", "tiny_asr_dot_mode1.json" if mode else "tiny_asr.json")
    with open(path, "w") as f:
        json.dump(res, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")
