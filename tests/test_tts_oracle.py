"""CPU tests of the TTS oracle (oracle/dsm_oracle_tts.inc): the delayed-streams bookkeeping of
tts_streaming::State::step (core/tts_streaming.rs:117-242) checked against hand-derived expectations, slot
independence, and the committed trace the GPU parity test also compares with.  The Candle path itself cannot be
run here (no Rust, no checkpoint), so the float side of this oracle is "parity unpinned"."""
import json
import os

import numpy as np
import pytest

from tts_schedule import run

HERE = os.path.dirname(os.path.abspath(__file__))
UNG = 0xFFFFFFFF


@pytest.fixture(scope="module")
def tts(dsm, orc):
    from dsm_amd import synth
    cfg = dsm.config_tts_tiny()
    return cfg, synth.make_synth_tts_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tts_tiny")


def test_delays_and_write_back(dsm, orc, tts):
    cfg, path = tts
    pad = cfg.audio_vocab_size - 1
    o = orc.OracleTts(cfg, 1, path)
    steps = 12
    lat = []
    for s in range(steps):
        text, audio = o.step([5], [7 + s], [1])
        assert text[0] == 7 + s  # AllowedTokens::Text(v) wins over the logits (:178-180)
        if s < cfg.text_audio_delay_in_tokens:
            assert np.all(audio == UNG)  # depformer not sampled yet (:203-205)
        else:
            assert np.all(audio < cfg.audio_vocab_size - 1)  # the depformer cannot emit the pad token
        lat.append(audio[0].copy())
    assert o.step_idx(0) == steps
    for s in range(steps):
        row = o.audio_tokens(0, s)
        # codebook 0 is written at its own step, codebooks > 0 acoustic_delay steps back (:220-236)
        exp0 = pad if s < cfg.text_audio_delay_in_tokens else lat[s][0]
        assert row[0] == exp0
        for k in range(1, cfg.dep_num_slices):
            src = s + cfg.acoustic_delay
            if src >= steps:
                assert row[k] == UNG or src < steps
                continue
            # the slot at index 0 is hit by steps 0..acoustic_delay (saturating_sub); the first write wins
            first_writer = 0 if s == 0 else src
            exp = pad if first_writer < cfg.text_audio_delay_in_tokens else lat[first_writer][k]
            assert row[k] == exp, (s, k)
    o.close()


def test_forced_end_of_pad(dsm, orc, tts):
    cfg, path = tts
    o = orc.OracleTts(cfg, 1, path)
    for s in range(cfg.max_consecutive_pads + 1):
        text, _ = o.step([3], [dsm.TTS_ALLOW_PAD], [1])
        assert text[0] == cfg.text_pad_token
    # consecutive_pads == max + 1 > max: PadOrEpad must now return end-of-pad whatever the logits say (:183-186)
    text, _ = o.step([3], [dsm.TTS_ALLOW_PAD_OR_EPAD], [1])
    assert text[0] == cfg.text_eop_token
    o.close()


def test_slots_are_independent(dsm, orc, tts):
    """A slot's tokens do not depend on what its batch neighbours do (each slot is its own generation)."""
    cfg, path = tts
    steps = 10
    o3 = orc.OracleTts(cfg, 3, path)
    tr3, tab3 = run(o3, cfg, 3, steps)
    o3.close()
    from tts_schedule import schedule
    sched = schedule(cfg, 3, steps)
    o1 = orc.OracleTts(cfg, 1, path)
    for s, (prev, allowed, mask) in enumerate(sched):
        if not mask[0]:
            continue
        text, audio = o1.step(prev[:1], allowed[:1], [1])
        assert text[0] == tr3[s][0][0]
        assert np.array_equal(audio[0], tr3[s][1][0])
    o1.close()


def test_max_step_idx_is_an_error(dsm, orc, tts):
    cfg, path = tts
    small = dsm.TtsConfig.from_buffer_copy(cfg)
    small.max_steps = 4
    o = orc.OracleTts(small, 1, path)
    n = small.max_steps + small.acoustic_delay
    for s in range(n - 1):
        o.step([1], [9], [1])
    with pytest.raises(RuntimeError):  # "max step-idx reached" (:238-240)
        o.step([1], [9], [1])
    for _ in range(2):  # a slot at the limit is refused, not stepped (the reference would panic on text_tokens[n])
        with pytest.raises(RuntimeError):
            o.step([1], [9], [1])
        assert o.step_idx(0) == n
    o.reset_batch_idx(0)
    o.step([1], [9], [1])
    assert o.step_idx(0) == 1
    o.close()


def test_committed_trace(dsm, orc, tts):
    cfg, path = tts
    with open(os.path.join(HERE, "golden", "tiny_tts.json")) as f:
        gold = json.load(f)
    o = orc.OracleTts(cfg, gold["B"], path)
    trace, tables = run(o, cfg, gold["B"], gold["steps"], resets={int(k): v for k, v in gold["resets"].items()})
    o.close()
    assert [t.tolist() for t, _ in trace] == gold["text"]
    assert [a.tolist() for _, a in trace] == gold["audio"]
    assert [[r.tolist() for r in tab] for tab in tables] == gold["tables"]
