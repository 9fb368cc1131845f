"""GPU parity of Mimi::decode_step (core/mimi.rs:217-225) through the C ABI: decoded PCM against the oracle.
north_star asks for 1e-4 RMS on PCM; the kernels follow the oracle's canonical orders, so the test demands
bit equality and states the RMS too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(dsm, orc, cfg, B, lm, mimi, steps, mask_fn, resets=None, seed=5):
    eng = dsm.AsrEngine(cfg, B, lm, mimi)
    ora = orc.OracleAsr(cfg, B, lm, mimi)
    rng = np.random.default_rng(seed)
    resets = resets or {}
    for s in range(steps):
        for slot in resets.get(s, []):
            eng.mimi_reset_batch_idx(slot)
            ora.mimi_reset_batch_idx(slot, side=0)
        codes = rng.integers(0, cfg.mimi.quantizer_bins, (B, cfg.mimi.quantizer_n_q)).astype(np.uint32)
        mask = np.asarray(mask_fn(s), dtype=np.uint8)
        act = mask.astype(bool)
        pe = eng.decode_step(codes, mask)
        po = ora.decode_step(codes, mask, side=0)
        assert pe is not None and po is not None and pe.shape == (B, 1920)
        rms = float(np.sqrt(np.mean((pe[act].astype(np.float64) - po[act]) ** 2))) if act.any() else 0.0
        assert rms <= 1e-4, f"decoded PCM RMS error {rms} at step {s}"
        assert np.array_equal(pe[act].view(np.uint32), po[act].view(np.uint32)), f"PCM bits differ at step {s} (rms {rms})"
        assert np.all(np.isfinite(pe[act]))
    eng.close()
    ora.close()


def test_decode_all_active(gpu, dsm, lib, orc, tiny_weights):
    _run(dsm, orc, dsm.config_tiny(), 3, *tiny_weights, steps=12, mask_fn=lambda s: [1, 1, 1])


def test_decode_masks_resets_and_ring_wrap(gpu, dsm, lib, orc, tiny_weights):
    """Mimi transformer ctx = 10 wraps after 5 frames; slot resets zero the conv/convtr carries only."""
    rng = np.random.default_rng(3)
    masks = (rng.random((20, 4)) < 0.7).astype(np.uint8)
    masks[:, 0] = 1
    _run(dsm, orc, dsm.config_tiny(), 4, *tiny_weights, steps=20, mask_fn=lambda s: masks[s], resets={6: [1], 11: [0, 2]})


def test_decode_after_encode_roundtrip_runs(gpu, dsm, lib, orc, tiny_weights):
    """encode -> decode of the engine's own codes stays finite and matches the oracle fed the same codes."""
    from dsm_amd import synth
    cfg = dsm.config_tiny()
    B = 2
    eng = dsm.AsrEngine(cfg, B, *tiny_weights)
    ora = orc.OracleAsr(cfg, B, *tiny_weights)
    pcm = synth.synth_pcm(B, 6)
    mask = np.ones(B, dtype=np.uint8)
    for s in range(6):
        codes = eng.encode_step(pcm[s], mask)
        assert np.array_equal(codes, ora.encode_step(pcm[s], mask))
        out_e = eng.decode_step(codes, mask)
        out_o = ora.decode_step(codes, mask, side=0)
        assert np.array_equal(out_e.view(np.uint32), out_o.view(np.uint32))
    eng.close()
    ora.close()


def test_weight_g_weight_v_checkpoint(gpu, dsm, lib, orc):
    """The un-folded checkpoint layout (`weight_g` + `weight_v` for every SEANet conv and transposed conv, as the
    published Mimi file stores them): the engine's load-time fold (core/conv.rs:27-45 per output channel, :130-141 per
    input channel) against the oracle's — encode codes + latent and decoded PCM, bit for bit, with a mask and a reset."""
    import os
    from dsm_amd import synth
    cfg = dsm.config_tiny()
    lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tiny", weight_norm=True)
    assert mimi.endswith("_wn.safetensors")
    keys = synth.read_safetensors(mimi).keys()
    assert not any(k.endswith("conv.conv.weight") and k.startswith(("encoder.", "decoder.")) for k in keys)
    assert sum(k.endswith("weight_g") for k in keys) == sum(k.endswith("weight_v") for k in keys) >= 20
    B = 3
    eng = dsm.AsrEngine(cfg, B, lm, mimi)
    ora = orc.OracleAsr(cfg, B, lm, mimi)
    pcm = synth.synth_pcm(B, 10)
    for s in range(10):
        if s == 6:
            eng.mimi_reset_batch_idx(2)
            ora.mimi_reset_batch_idx(2, side=0)
        mask = np.array([1, s % 3 != 1, 1], dtype=np.uint8)
        act = mask.astype(bool)
        ce, co = eng.encode_step(pcm[s], mask), ora.encode_step(pcm[s], mask)
        le = eng.debug_read("mimi.latent", B * cfg.mimi.dimension).reshape(B, -1)
        lo = ora.debug_read("mimi.latent", B * cfg.mimi.dimension).reshape(B, -1)
        assert np.array_equal(le[act].view(np.uint32), lo[act].view(np.uint32)), f"latent differs at step {s}"
        assert np.array_equal(ce[act], co[act]), f"codes differ at step {s}"
        dc = np.where(act[:, None], co, 0).astype(np.uint32)
        pe, po = eng.decode_step(dc, mask), ora.decode_step(dc, mask, side=0)
        assert np.array_equal(pe[act].view(np.uint32), po[act].view(np.uint32)), f"decoded PCM differs at step {s}"
    eng.close()
    ora.close()
