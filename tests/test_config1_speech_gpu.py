"""BASELINE.json configs[0] / SURVEY.md §8(d) config 1: B = 1 plumbing on the reference's own speech sample
(`audio/tts-input-sample-01.wav`, here as the 24 kHz mono fixture made by tools/make_audio_fixture.py) at the real
stt-1b-en_fr shapes: Mimi codes, text tokens, VAD heads and LM logits of the HIP engine against the oracle, frame by
frame.  Weights are synthetic (none exist offline), so the transcript is meaningless; what is checked is that real
speech statistics (silence, onsets, large dynamic range) go through both paths identically."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_b1_real_speech_frames(gpu, dsm, lib, orc):
    from dsm_amd import synth
    cfg = dsm.config_stt_1b_en_fr()
    lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="stt-1b-en_fr")
    s16 = np.fromfile(os.path.join(HERE, "golden", "speech_24k_mono.s16"), dtype="<i2")
    pcm = s16.astype(np.float32) / np.float32(32768)
    first, count = 0, 24  # 0.64 s of near-silence (|x| < 1e-3), the onset at frame 8, then speech: 1.92 s in all
    frames = pcm[: (pcm.size // 1920) * 1920].reshape(-1, 1920)
    assert frames.shape[0] == 79 and np.abs(frames[:8]).max() < 1e-3 and np.abs(frames[8:count]).max() > 0.05
    eng = dsm.AsrEngine(cfg, 1, lm, mimi)
    ora = orc.OracleAsr(cfg, 1, lm, mimi)
    mask = np.ones(1, dtype=np.uint8)
    seen = set()
    for i in range(first, first + count):
        ec, et, ep = eng.step_pcm(frames[i][None, :], mask)
        oc, ot, op = ora.step_pcm(frames[i][None, :], mask)
        assert np.array_equal(ec, oc), f"codes differ at frame {i}"
        assert np.array_equal(et, ot), f"text token differs at frame {i}"
        assert np.array_equal(ep.view(np.uint32), op.view(np.uint32)), f"VAD heads differ at frame {i}"
        lg_e = eng.debug_read("lm.logits", cfg.text_out_vocab_size)
        lg_o = ora.debug_read("lm.logits", cfg.text_out_vocab_size)
        assert np.array_equal(lg_e.view(np.uint32), lg_o.view(np.uint32)), f"logits differ at frame {i}"
        assert eng.poll_msgs() == ora.poll_msgs()
        seen.update(int(c) for c in ec[0])
    assert len(seen) > 100  # real audio walks the codebooks, it does not sit on one code
    eng.close()
    ora.close()


def test_b1_bria_mp3_the_named_input_of_config_0(gpu, dsm, lib, orc):
    """BASELINE.json configs[0] names audio/bria.mp3: the request path of srv/batched_asr.rs:834-842 on (the first 10 s of) that
    clip — pcm_decode of the mp3 body (csrc/dsm_mp3.inc), resample 44.1 -> 24 kHz (kaudio::resample's place), 1920-sample frames
    into step_pcm at B = 1 — HIP engine against the oracle on every frame's codes, tokens, VAD heads and logits, in the
    presets' dot_mode.  The decoder and the resampler are parity-unpinned against symphonia / kaudio (tests/test_mp3.py pins
    their properties); what this test adds is that the decoded clip drives both paths identically."""
    from dsm_amd import synth
    cfg = dsm.config_stt_1b_en_fr()
    lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="stt-1b-en_fr")
    body = open(os.path.join(HERE, "golden", "audio", "bria_head.mp3"), "rb").read()
    pcm44, rate = dsm.pcm_decode(body)
    assert rate == 44100 and len(pcm44) == 384 * 1152
    pcm = dsm.resample(pcm44, rate, 24000)
    assert abs(len(pcm) - len(pcm44) * 80 / 147) < 1
    frames = pcm[: (pcm.size // 1920) * 1920].reshape(-1, 1920)
    assert frames.shape[0] == 125 and np.abs(frames).max() > 0.1
    eng = dsm.AsrEngine(cfg, 1, lm, mimi)
    ora = orc.OracleAsr(cfg, 1, lm, mimi)
    mask = np.ones(1, dtype=np.uint8)
    seen = set()
    for i in range(20, 36):  # 1.28 s from inside the clip
        ec, et, ep = eng.step_pcm(frames[i][None, :], mask)
        oc, ot, op = ora.step_pcm(frames[i][None, :], mask)
        assert np.array_equal(ec, oc) and np.array_equal(et, ot), f"codes / token differ at frame {i}"
        assert np.array_equal(ep.view(np.uint32), op.view(np.uint32)), f"VAD heads differ at frame {i}"
        lg_e = eng.debug_read("lm.logits", cfg.text_out_vocab_size)
        lg_o = ora.debug_read("lm.logits", cfg.text_out_vocab_size)
        assert np.array_equal(lg_e.view(np.uint32), lg_o.view(np.uint32)), f"logits differ at frame {i}"
        seen.update(int(c) for c in ec[0])
    assert len(seen) > 100
    eng.close()
    ora.close()
