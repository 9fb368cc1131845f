"""GPU parity of the TTS step (tts_streaming::State::step + LmModel::forward_cond + DepFormer::sample, greedy) through
the C ABI against the oracle: every emitted token and every audio_tokens table entry must be identical."""
import json
import os

import numpy as np
import pytest

from tts_schedule import run, schedule

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def tts(dsm):
    from dsm_amd import synth
    cfg = dsm.config_tts_tiny()
    return cfg, synth.make_synth_tts_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tts_tiny")


def _compare(dsm, orc, cfg, path, B, steps, resets=None):
    eng = dsm.TtsEngine(cfg, B, path)
    ora = orc.OracleTts(cfg, B, path)
    resets = resets or {}
    for s, (prev, allowed, mask) in enumerate(schedule(cfg, B, steps)):
        for slot in resets.get(s, []):
            eng.reset_batch_idx(slot)
            ora.reset_batch_idx(slot)
        te, ae = eng.step(prev, allowed, mask)
        to, ao = ora.step(prev, allowed, mask)
        act = mask.astype(bool)
        n = B * cfg.lm.d_model
        he, ho = eng.debug_read("lm.hidden", n).reshape(B, -1), ora.debug_read("lm.hidden", n).reshape(B, -1)
        assert np.array_equal(he[act].view(np.uint32), ho[act].view(np.uint32)), f"LM hidden bits differ at step {s}"
        assert np.array_equal(te[act], to[act]), f"text tokens differ at step {s}"
        assert np.array_equal(ae[act], ao[act]), f"depformer tokens differ at step {s}: {ae[act]} vs {ao[act]}"
    for b in range(B):
        assert eng.step_idx(b) == ora.step_idx(b)
        for i in range(eng.step_idx(b)):
            assert np.array_equal(eng.audio_tokens(b, i), ora.audio_tokens(b, i))
    eng.close()
    ora.close()


@pytest.mark.parametrize("kv_bf16", [1, 0])
def test_tts_tiny_parity(gpu, dsm, lib, orc, tts, kv_bf16):
    cfg, path = tts
    c = dsm.TtsConfig.from_buffer_copy(cfg)
    c.kv_bf16 = kv_bf16
    _compare(dsm, orc, c, path, 4, 24, resets={9: [2], 15: [0, 3]})


def test_tts_ring_wrap(gpu, dsm, lib, orc, tts):
    """Main-LM context is 16: 40 steps wrap the ring cache more than twice."""
    cfg, path = tts
    _compare(dsm, orc, cfg, path, 2, 40)


def test_tts_committed_trace(gpu, dsm, lib, tts):
    cfg, path = tts
    with open(os.path.join(HERE, "golden", "tiny_tts.json")) as f:
        gold = json.load(f)
    eng = dsm.TtsEngine(cfg, gold["B"], path)
    trace, tables = run(eng, cfg, gold["B"], gold["steps"], resets={int(k): v for k, v in gold["resets"].items()})
    eng.close()
    assert [t.tolist() for t, _ in trace] == gold["text"]
    assert [a.tolist() for _, a in trace] == gold["audio"]
    assert [[r.tolist() for r in tab] for tab in tables] == gold["tables"]


def test_tts_errors(gpu, dsm, lib, tts):
    cfg, path = tts
    small = dsm.TtsConfig.from_buffer_copy(cfg)
    small.max_steps = 4
    eng = dsm.TtsEngine(small, 1, path)
    for s in range(small.max_steps + small.acoustic_delay - 1):
        eng.step([1], [9], [1])
    with pytest.raises(dsm.DsmError, match="max step-idx"):
        eng.step([1], [9], [1])
    eng.close()
    with pytest.raises(dsm.DsmError):
        dsm.TtsEngine(cfg, 1, path + ".missing")


def test_tts_stepping_past_the_limit_is_refused_without_side_effects(gpu, dsm, lib, orc, tts):
    """A slot that reached n_steps (the step that returned "max step-idx reached") must not be stepped again: the
    reference would index text_tokens[n_steps] (a panic); the engine refuses the step and changes nothing, for any number
    of further calls, while the other slot of the batch keeps generating once the full slot is masked out or reset."""
    cfg, path = tts
    small = dsm.TtsConfig.from_buffer_copy(cfg)
    small.max_steps = 4
    n = small.max_steps + small.acoustic_delay
    eng, ora = dsm.TtsEngine(small, 2, path), orc.OracleTts(small, 2, path)
    for s in range(n - 1):  # slot 1 starts one step later than slot 0
        args = ([1, 2], [9, 11], [1, 1 if s else 0])
        te, ae = eng.step(*args)
        to, ao = ora.step(*args)
        assert np.array_equal(te[:1], to[:1]) and np.array_equal(ae[:1], ao[:1])
    with pytest.raises(dsm.DsmError, match="max step-idx"):
        eng.step([1, 2], [9, 11], [1, 1])  # slot 0 reaches the limit; slot 1 was advanced too
    with pytest.raises(RuntimeError):
        ora.step([1, 2], [9, 11], [1, 1])
    assert eng.step_idx(0) == n == ora.step_idx(0) and eng.step_idx(1) == n - 1 == ora.step_idx(1)
    tables = [eng.audio_tokens(b, i).copy() for b in range(2) for i in range(n)]
    for _ in range(2):  # twice past the limit: refused, no state change (this used to write past the vectors)
        with pytest.raises(dsm.DsmError, match="max step-idx"):
            eng.step([1, 2], [9, 11], [1, 1])
        with pytest.raises(RuntimeError):
            ora.step([1, 2], [9, 11], [1, 1])
        assert eng.step_idx(0) == n and eng.step_idx(1) == n - 1 and ora.step_idx(1) == n - 1
        assert all(np.array_equal(a, b) for a, b in zip(tables, [eng.audio_tokens(b, i) for b in range(2) for i in range(n)]))
    eng.reset_batch_idx(0); ora.reset_batch_idx(0)  # the full slot is recycled, the batch goes on
    te, ae = eng.step([1, 2], [9, 11], [1, 0])
    to, ao = ora.step([1, 2], [9, 11], [1, 0])
    assert np.array_equal(te[:1], to[:1]) and np.array_equal(ae[:1], ao[:1]) and eng.step_idx(0) == 1
    eng.close(); ora.close()


def test_tts_engine_creation_is_race_free(gpu, dsm, lib, orc, tts):
    """Regression for a load-time race: the low-rank depformer tables are folded by a GEMM on the engine's
    (non-blocking) stream right after their operands are uploaded on the null stream; without an explicit wait the fold
    could read a half-landed upload and one (slice, token) embedding row came out wrong, once in many runs.  Recreate
    the engine several times over deliberately dirtied device memory and compare the first depformer steps."""
    import torch
    cfg, path = tts
    B = 4
    ora = orc.OracleTts(cfg, B, path)
    want = [ora.step(prev, allowed, mask) for prev, allowed, mask in schedule(cfg, B, 6)]
    ora.close()
    for trial in range(8):
        junk = [torch.full(((3 + trial) << 20,), float("nan"), device="cuda") for _ in range(3)]
        del junk
        torch.cuda.empty_cache()
        eng = dsm.TtsEngine(cfg, B, path)
        for s, (prev, allowed, mask) in enumerate(schedule(cfg, B, 6)):
            te, ae = eng.step(prev, allowed, mask)
            act = mask.astype(bool)
            assert np.array_equal(te[act], want[s][0][act]) and np.array_equal(ae[act], want[s][1][act]), (trial, s)
        eng.close()


def test_tts_seeded_topk_sampling_matches_the_oracle(gpu, dsm, lib, orc, tts):
    """DepFormer::sample / text_lp.sample with Sampling::TopK (core/lm.rs:674, core/tts_streaming.rs:188, srv/tts.rs:401-415):
    per-slot k, temperature and seed; the device sampler (softmax at the temperature, bitonic top-k, ChaCha12 draw) must
    pick the oracle's token at every slice of every step — slot 0 stays ArgMax, slot 1 samples from 5, slot 2 from the whole
    vocabulary (k >= V: candle's multinomial over all tokens in id order), slot 3 from 17 with a mid-run reset (back to
    ArgMax) and re-configuration.  Hundreds of draws per slot; parity vs Candle itself is unpinned (no reference vector)."""
    cfg, path = tts
    B, steps = 4, 16
    eng, ora = dsm.TtsEngine(cfg, B, path), orc.OracleTts(cfg, B, path)
    conf = {1: (5, 0.8, 1234), 2: (250, 1.3, 99), 3: (17, 1.0, 2**40 + 3)}
    for e in (eng, ora):
        for slot, (k, temp, seed) in conf.items():
            e.set_sampling(slot, k, temp, seed)
    greedy = orc.OracleTts(cfg, B, path)
    differs = False
    for s, (prev, allowed, mask) in enumerate(schedule(cfg, B, steps)):
        if s == 9:
            for e in (eng, ora, greedy):
                e.reset_batch_idx(3)
        if s == 11:
            eng.set_sampling(3, 9, 0.7, 5)
            ora.set_sampling(3, 9, 0.7, 5)
        mask = np.ones(B, dtype=np.uint8) if s < 4 else mask
        act = mask.astype(bool)
        te, ae = eng.step(prev, allowed, mask)
        to, ao = ora.step(prev, allowed, mask)
        tg, ag = greedy.step(prev, allowed, mask)
        assert np.array_equal(te[act], to[act]), f"text tokens differ at step {s}"
        assert np.array_equal(ae[act], ao[act]), f"sampled depformer tokens differ at step {s}"
        assert np.array_equal(ae[0], ag[0]) or not act[0]
        differs = differs or (act[1] and not np.array_equal(ao[1], ag[1]))
    assert differs, "sampling never changed a token: the test would prove nothing"
    for b in range(B):
        for i in range(eng.step_idx(b)):
            assert np.array_equal(eng.audio_tokens(b, i), ora.audio_tokens(b, i))
    eng.close(); ora.close(); greedy.close()


def test_tts_logits_rows_that_are_no_multiple_of_four_wide(gpu, dsm, lib, orc):
    """The sampler stages a slot's logits row in LDS (r03): 16-byte accesses for the shipped vocabulary (2048), element by
    element for a width like 34."""
    from dsm_amd import synth
    cfg = dsm.config_tts_tiny()
    cfg.audio_vocab_size = 35  # 34 logits per slice
    path = synth.make_synth_tts_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tts_tiny_vocab35")
    _compare(dsm, orc, cfg, path, 3, 14, resets={6: [1]})
