"""GPU parity of the TTS branch the reference server runs (srv/tts.rs:426-441) — cross-attention to a per-slot source
(LmModel::forward_ca) and classifier-free guidance (two batch rows per slot, l0 * a - l1 * (a - 1) mixes of the text logits and
of every depformer slice, DepFormer::sample_cfg) — through the C ABI against the oracle: every token, every audio_tokens
table entry and the main LM's hidden state of every batch row, bit for bit.  The oracle's own pins for this branch are in
tests/test_tts_ca_oracle.py (numpy float64 evaluation, alpha = 1, source-less slots); vs Candle: parity unpinned."""
import os

import numpy as np
import pytest

from tts_schedule import schedule

pytestmark = pytest.mark.gpu
WDIR = os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights")


def _cfg(dsm, hd64=False, **kw):
    from dsm_amd import synth
    cfg = dsm.config_tts_tiny(cross_attention=True, **kw)
    if hd64:  # head_dim 64 with rings / sources of at most 32 rows: attn_small_kernel (self-attention from stored q, cross-attention)
        cfg.lm.num_heads, cfg.depformer.num_heads = 2, 1
    tag = "tts_tiny_ca" + ("_kvd%d" % cfg.ca_dim if cfg.ca_dim else "") + ("_rms" if cfg.ca_norm else "") + ("_hd64" if hd64 else "")
    return cfg, synth.make_synth_tts_weights(cfg, WDIR, tag=tag)


def _drive(dsm, orc, cfg, path, B, steps, setup, events=None, sampling=None):
    """setup(engine): the per-slot sources / guidance; events {step: fn(engine)} run on both sides before that step."""
    eng, ora = dsm.TtsEngine(cfg, B, path), orc.OracleTts(cfg, B, path)
    for x in (eng, ora):
        setup(x)
        for slot, (k, temp, seed) in (sampling or {}).items():
            x.set_sampling(slot, k, temp, seed)
    rps = 2 if cfg.cfg_rows else 1
    R, d = B * rps, cfg.lm.d_model
    for s, (prev, allowed, mask) in enumerate(schedule(cfg, B, steps)):
        if events and s in events:
            for x in (eng, ora):
                events[s](x)
        te, ae = eng.step(prev, allowed, mask)
        to, ao = ora.step(prev, allowed, mask)
        act = mask.astype(bool)
        he, ho = eng.debug_read("lm.hidden", R * d).reshape(B, rps, d), ora.debug_read("lm.hidden", R * d).reshape(B, rps, d)
        # row 0 of every active slot; row 1 where both sides ran it (a guided slot): the oracle leaves idle rows unspecified too
        assert np.array_equal(he[act, 0].view(np.uint32), ho[act, 0].view(np.uint32)), f"LM hidden bits differ at step {s}"
        assert np.array_equal(te[act], to[act]), f"text tokens differ at step {s}: {te[act]} vs {to[act]}"
        assert np.array_equal(ae[act], ao[act]), f"depformer tokens differ at step {s}"
    for b in range(B):
        assert eng.step_idx(b) == ora.step_idx(b)
        for i in range(eng.step_idx(b)):
            assert np.array_equal(eng.audio_tokens(b, i), ora.audio_tokens(b, i))
    m = eng.metrics()
    assert m.capture_failures == 0, m.capture_error
    eng.close(); ora.close()


@pytest.mark.parametrize("kv_bf16", [1, 0])
@pytest.mark.parametrize("kw", [dict(), dict(ca_dim=40), dict(ca_norm=1), dict(hd64=True)])
def test_cross_attention_tiny(gpu, dsm, lib, orc, kw, kv_bf16):
    from dsm_amd import synth
    cfg, path = _cfg(dsm, **kw)
    cfg.kv_bf16 = kv_bf16
    B = 4
    lens = [24, 1, 0, 13]  # the longest allowed, a single row, none (ca_src = None), a ragged one

    def setup(x):
        for b, n in enumerate(lens):
            if n:
                x.set_ca_src(b, synth.synth_ca_src(cfg, n, 50 + b))

    def swap(x):  # a new request on slot 1: reset (source cleared), then a longer source; slot 3 loses its source
        x.reset_batch_idx(1)
        x.set_ca_src(1, synth.synth_ca_src(cfg, 17, 77))
        x.set_ca_src(3, None)

    _drive(dsm, orc, cfg, path, B, 26, setup, events={11: swap})


def test_guidance_tiny(gpu, dsm, lib, orc):
    """cfg_rows: slots 0 and 2 guided (alpha 2.5 / 1.0), slot 1 with a source and no guidance, slot 3 without a source;
    slot 2 also samples (seeded top-k from the MIXED logits); a reset in the middle and guidance switched on for slot 1."""
    from dsm_amd import synth
    cfg, path = _cfg(dsm, cfg_rows=True)
    B = 4
    empty = synth.synth_ca_src(cfg, 9, 99)

    def setup(x):
        x.set_ca_src(0, synth.synth_ca_src(cfg, 20, 1), empty, 2.5)
        x.set_ca_src(1, synth.synth_ca_src(cfg, 8, 2))
        x.set_ca_src(2, synth.synth_ca_src(cfg, 9, 3), empty, 1.0)

    def swap(x):
        x.reset_batch_idx(1)
        x.set_ca_src(1, synth.synth_ca_src(cfg, 12, 4), synth.synth_ca_src(cfg, 12, 5), 3.0)
        x.set_sampling(1, 6, 0.9, 1234)

    _drive(dsm, orc, cfg, path, B, 28, setup, events={10: swap}, sampling={2: (5, 0.8, 42)})


def test_cross_attention_and_guidance_at_v202501_shapes(gpu, dsm, lib, orc):
    """The real dimensions (2048-d x 16 main LM with norm_cross + cross attention per layer, depformer 1024-d x 4 x 32 slices):
    three slots = six batch rows, sources of 5 x 25 rows (speaker_cond_n_speakers x 2 s at 12.5 Hz), guidance on two slots."""
    from dsm_amd import synth
    cfg = dsm.config_tts_v202501()
    cfg.dot_mode = 0
    cfg.text_audio_delay_in_tokens, cfg.max_steps = 2, 64
    cfg.cross_attention, cfg.ca_norm, cfg.ca_dim, cfg.ca_max_len, cfg.cfg_rows = 1, 0, 0, 128, 1
    path = synth.make_synth_tts_weights(cfg, WDIR, tag="tts-v202501-ca")
    B = 3
    empty = synth.synth_ca_src(cfg, 125, 9)

    def setup(x):
        x.set_ca_src(0, synth.synth_ca_src(cfg, 125, 1), empty, 2.0)
        x.set_ca_src(1, synth.synth_ca_src(cfg, 50, 2))
        x.set_ca_src(2, synth.synth_ca_src(cfg, 128, 3), empty, 1.5)

    try:
        _drive(dsm, orc, cfg, path, B, 5, setup, sampling={2: (50, 0.6, 7)})
        cfg.dot_mode = 1  # the same model on the bf16 matrix instruction (r03): K = 1024 / 2048 / 5632 bx3 GEMMs at six rows
        _drive(dsm, orc, cfg, path, B, 3, setup, sampling={2: (50, 0.6, 7)})
    finally:
        os.remove(path)  # 1.1 GB


def test_source_only_on_a_fresh_slot_and_ca_checkpoint_needs_ca_config(gpu, dsm, lib):
    """(1) a slot that has generated steps refuses a new source / guidance until it is reset (the reference fixes
    ca_src and the batch size at State::new, core/tts_streaming.rs:100-115), clearing stays allowed; (2) a checkpoint that
    carries cross_attention.* tensors is refused by an engine configured without cross attention instead of skipping them."""
    from dsm_amd import synth
    from tts_schedule import schedule
    cfg, path = _cfg(dsm, cfg_rows=True)
    B = 2
    eng = dsm.TtsEngine(cfg, B, path)
    eng.set_ca_src(0, synth.synth_ca_src(cfg, 8, 1))
    for s, (prev, allowed, mask) in enumerate(schedule(cfg, B, 3)):
        eng.step(prev, allowed, mask)
    with pytest.raises(dsm.DsmError, match="fresh slot"):
        eng.set_ca_src(0, synth.synth_ca_src(cfg, 8, 1), synth.synth_ca_src(cfg, 8, 2), 2.0)
    with pytest.raises(dsm.DsmError, match="fresh slot"):
        eng.set_ca_src(1, synth.synth_ca_src(cfg, 5, 3))
    eng.set_ca_src(0, None)  # clearing is allowed at any step
    eng.reset_batch_idx(1)
    eng.set_ca_src(1, synth.synth_ca_src(cfg, 5, 3))
    eng.close()
    plain = dsm.config_tts_tiny()
    with pytest.raises(dsm.DsmError, match="cross_attention"):
        dsm.TtsEngine(plain, B, path)
