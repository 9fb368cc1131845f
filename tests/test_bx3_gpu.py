"""dot_mode 1 ("bx3") on the GPU: the bf16-weight GEMMs on v_mfma_f32_16x16x32_bf16 (gemm_bx3_kernel / the generic kernel's bx3
branch) against the oracle's integer restatement of that instruction (orc_linear_bx3, csrc/dsm_bf16_mfma_model.h) — codes,
tokens, VAD, AsrMsgs and every float tap bit for bit, like the mode-0 tests:
  * tiny model: masks, resets, ring wrap, f32 / bf16 rings, B = 1 and a ragged B = 17 (the generic kernel: K = 128 < 256 ... one chunk)
  * medium models with the real head dims: d_model 512 = two K-chunks (split-K slabs + the attention prologue's ordered reduce)
  * stt-1b-en_fr at the real dimensions, B = 3, against the oracle; B = 64 (two stream groups, MT = 2 tiles, 8-slab QKV) and
    B = 1024 (the whole-K loop form) by the slot-independence property against a B = 4 engine
  * TTS (cross-attention, guidance, depformer with its K = 40 ... 2048 GEMMs and the load-time table fold), tiny
  * stt-2.6b-en and the v202501 TTS shapes in mode 1 ride on the tests that already build those checkpoints
    (test_parity_full_gpu.py::test_stt_2_6b_en_shapes, test_tts_ca_gpu.py::test_cross_attention_and_guidance_at_v202501_shapes)"""
import os

import numpy as np
import pytest

from test_parity_gpu import run_pair

pytestmark = pytest.mark.gpu
WEIGHTS_DIR = os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights")


def test_tiny_masks_resets_bx3(gpu, dsm, lib, orc, tiny_weights):
    cfg = dsm.config_tiny()
    cfg.dot_mode = 1
    rng = np.random.default_rng(7)
    masks = (rng.random((40, 5)) < 0.7).astype(np.uint8)
    masks[:, 0] = 1
    run_pair(dsm, orc, cfg, 5, *tiny_weights, steps=40, mask_fn=lambda s: masks[s], resets={9: [1], 17: [0, 3], 18: [3], 30: [2]})
    cfg0 = dsm.config_tiny(kv_bf16=0)
    cfg0.dot_mode = 1
    run_pair(dsm, orc, cfg0, 2, *tiny_weights, steps=12, mask_fn=lambda s: [1, 1])
    for B in (1, 17):
        run_pair(dsm, orc, cfg, B, *tiny_weights, steps=4, mask_fn=lambda s: [1] * B)


def test_the_two_modes_differ_in_bits_not_in_value(gpu, dsm, lib, tiny_weights):
    from dsm_amd import synth
    cfg0, cfg1 = dsm.config_tiny(), dsm.config_tiny()
    cfg1.dot_mode = 1
    B = 3
    e0, e1 = dsm.AsrEngine(cfg0, B, *tiny_weights), dsm.AsrEngine(cfg1, B, *tiny_weights)
    pcm = synth.synth_pcm(B, 6)
    differs = False
    for s in range(6):
        m = np.ones(B, np.uint8)
        c0, t0, _ = e0.step_pcm(pcm[s], m)
        c1, t1, _ = e1.step_pcm(pcm[s], m)
        assert np.array_equal(c0, c1)  # Mimi (f32 weights) is untouched by the mode
        h0, h1 = e0.debug_read("lm.hidden", B * cfg0.lm.d_model), e1.debug_read("lm.hidden", B * cfg0.lm.d_model)
        assert np.allclose(h0, h1, rtol=2e-4, atol=2e-5)
        differs = differs or not np.array_equal(h0, h1)
    assert differs
    e0.close(); e1.close()


@pytest.mark.parametrize("name,kw,frames", [
    ("bf16_hd128_ctx300", dict(lm_heads=4, lm_head_dim=128, lm_context=300, kv_bf16=1), 40),
    ("f32_hd64_ctx300", dict(lm_heads=8, lm_head_dim=64, lm_context=300, kv_bf16=0, mimi_head_dim=32, mimi_context=600), 24),
    # rings of at most 32 positions with head_dim 64: attn_small_kernel (one wave per (slot, head)) with the fused QKV prologue,
    # RoPE and ring wrap after 24 frames, on the bf16 and on the f32 ring
    ("bf16_hd64_ctx24", dict(lm_heads=8, lm_head_dim=64, lm_context=24, kv_bf16=1), 40),
    ("f32_hd64_ctx32", dict(lm_heads=8, lm_head_dim=64, lm_context=32, kv_bf16=0), 40),
])
@pytest.mark.parametrize("wk_norm", ["0", "1"])
def test_medium_two_chunk_models_bx3(gpu, dsm, lib, orc, name, kw, frames, wk_norm, monkeypatch):
    """d_model 512 = two K-chunks: split-K slabs through gemm_bx3u_kernel + the attention prologue's ordered reduce, and the
    whole-K forms (r04): the gate through gemm_wk_kernel; with DSM_WK_NORM=1 also out_proj whole-K with the residual in its epilogue
    and norm2 in the gate's prologue (gemm_wkn_kernel: waves 2 and 3 own no chunk and contribute +0 totals)."""
    from dsm_amd import synth
    monkeypatch.setenv("DSM_WK_NORM", wk_norm)
    cfg = dsm.config_medium(**kw)
    cfg.dot_mode = 1
    lm, mimi = synth.make_synth_weights(cfg, WEIGHTS_DIR, tag="medium_" + name)
    B = 5
    rng = np.random.default_rng(11)
    masks = (rng.random((frames, B)) < 0.8).astype(np.uint8)
    masks[:, 0] = 1
    run_pair(dsm, orc, cfg, B, lm, mimi, steps=frames, mask_fn=lambda s: masks[s], resets={7: [1], 15: [2, 0]})


def test_stt_1b_real_dimensions_bx3(gpu, dsm, lib, orc):
    from dsm_amd import synth
    cfg = dsm.config_stt_1b_en_fr()
    cfg.dot_mode = 1
    lm, mimi = synth.make_synth_weights(cfg, WEIGHTS_DIR, tag="stt-1b-en_fr")
    run_pair(dsm, orc, cfg, 3, lm, mimi, steps=3, mask_fn=lambda s: [1, 1, s != 1])


@pytest.mark.parametrize("B", [64, 128, 1024])
def test_real_dimensions_large_batches_by_slot_independence_bx3(gpu, dsm, lib, B):
    """B = 64: two stream groups of 32 (MT = 2, split-K slabs, fused QKV prologue); B = 128: groups of 64 rows — one 64-row tile on
    gemm_bx3_kernel for QKV / gate / ff_out, two 32-row z-tiles on gemm_bx3u_kernel for out_proj (DSM_BX3U_M64); B = 1024: gemm_bx3_kernel's whole-K loop
    form over 8 m-tiles per group.  Streams never interact: every slot must equal, bit for bit, the same stream stepped by the
    B = 4 engine that test_stt_1b_real_dimensions_bx3 ties to the oracle."""
    from dsm_amd import synth
    cfg = dsm.config_stt_1b_en_fr()
    cfg.dot_mode = 1
    lm, mimi = synth.make_synth_weights(cfg, WEIGHTS_DIR, tag="stt-1b-en_fr")
    NS, steps = 4, 3
    rng = np.random.default_rng(21)
    src_pcm = synth.synth_pcm(NS, steps, seed=77)
    src_mask = (rng.random((steps, NS)) < 0.85).astype(np.uint8)
    src_mask[0] = 1
    owner = rng.integers(0, NS, B)
    owner[[0, 1, 2, 3, B // 2 - 1, B // 2, B - 2, B - 1]] = [0, 1, 2, 3, 0, 1, 2, 3]
    small = dsm.AsrEngine(cfg, NS, lm, mimi)
    big = dsm.AsrEngine(cfg, B, lm, mimi, arena=small.weight_arena())
    for rnd in range(2):
        if rnd == 1:  # again on wrapped, full-length rings
            for e in (small, big):
                e.debug_set_positions(4 * cfg.lm.context + 5, 4 * cfg.mimi.transformer.context + 3)
        for s in range(steps):
            cs, ts, ps = small.step_pcm(src_pcm[s], src_mask[s])
            hs = small.debug_read("lm.hidden", NS * cfg.lm.d_model).reshape(NS, -1)
            cb, tb, pb = big.step_pcm(np.ascontiguousarray(src_pcm[s][owner]), np.ascontiguousarray(src_mask[s][owner]))
            hb = big.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
            act = src_mask[s][owner].astype(bool)
            assert np.array_equal(cb[act], cs[owner][act]) and np.array_equal(tb[act], ts[owner][act])
            assert np.array_equal(pb[:, act].view(np.uint32), ps[:, owner][:, act].view(np.uint32))
            assert np.array_equal(hb[act].view(np.uint32), hs[owner][act].view(np.uint32)), f"hidden state differs (round {rnd}, step {s})"
    m = big.metrics()
    assert m.capture_failures == 0, m.capture_error
    big.close(); small.close()


def test_tts_bx3(gpu, dsm, lib, orc):
    """TTS in mode 1: main LM with cross-attention (source width 40: the generic kernel's bx3 branch with a K tail) and guidance,
    the depformer (its per-slice GEMMs, the low-rank table folded at load with the same kernel), seeded top-k on one slot."""
    from dsm_amd import synth
    from tts_schedule import schedule
    cfg = dsm.config_tts_tiny(cross_attention=True, cfg_rows=True, ca_dim=40)
    cfg.dot_mode = 1
    path = synth.make_synth_tts_weights(cfg, WEIGHTS_DIR, tag="tts_tiny_ca_kvd40")
    B = 3
    eng, ora = dsm.TtsEngine(cfg, B, path), orc.OracleTts(cfg, B, path)
    for x in (eng, ora):
        x.set_ca_src(0, synth.synth_ca_src(cfg, 20, 1), synth.synth_ca_src(cfg, 9, 99), 2.5)
        x.set_ca_src(1, synth.synth_ca_src(cfg, 8, 2))
        x.set_sampling(1, 5, 0.8, 42)
    R, d = 2 * B, cfg.lm.d_model
    for s, (prev, allowed, mask) in enumerate(schedule(cfg, B, 20)):
        te, ae = eng.step(prev, allowed, mask)
        to, ao = ora.step(prev, allowed, mask)
        act = mask.astype(bool)
        he, ho = eng.debug_read("lm.hidden", R * d).reshape(B, 2, d), ora.debug_read("lm.hidden", R * d).reshape(B, 2, d)
        assert np.array_equal(he[act, 0].view(np.uint32), ho[act, 0].view(np.uint32)), f"LM hidden bits differ at step {s}"
        assert np.array_equal(te[act], to[act]) and np.array_equal(ae[act], ao[act]), f"tokens differ at step {s}"
    eng.close(); ora.close()

