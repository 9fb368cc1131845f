"""Parity at BASELINE.json's real shapes (stt-1b-en_fr: d 2048, 16 heads x 128, 16 layers, ctx 750, hidden 5632,
32 codebooks x 2048 x 256, real Mimi v0_1) on a small batch so that the CPU oracle finishes in seconds per step.
Exercises the code paths the tiny model cannot: 22-chunk split-K (K = 5632), hd = 128 attention, K = 8192 convs,
the 1-channel K = 7 input conv, 32-stage RVQ."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_weights(dsm):
    from dsm_amd import synth
    import os
    cfg = dsm.config_stt_1b_en_fr()
    cfg.dot_mode = 0  # the preset says 1 (r04); the tests of this file name the mode they run (dot_mode 1: test_bx3_gpu.py and the parametrised ones)
    return cfg, synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="stt-1b-en_fr")


@pytest.mark.parametrize("chunk_loop_min", [None, "1"])
def test_full_size_encode_lm_and_decode(gpu, dsm, lib, orc, full_weights, monkeypatch, chunk_loop_min):
    """chunk_loop_min = "1": every split-K GEMM (LM: 8 and 22 chunks; Mimi convs) walks its chunks inside the workgroup
    and sums them in registers — the path large batches take (GemmArgs::chunk_loop) — instead of writing slabs for a
    reduce launch; the QKV prologue fusion falls back to the direct epilogue.  Same bits either way."""
    from dsm_amd import synth
    cfg, (lm, mimi) = full_weights
    if chunk_loop_min:
        monkeypatch.setenv("DSM_CHUNK_LOOP_MIN", chunk_loop_min)
    B, steps = 3, (4 if chunk_loop_min is None else 2)
    eng = dsm.AsrEngine(cfg, B, lm, mimi)
    ora = orc.OracleAsr(cfg, B, lm, mimi)
    pcm = synth.synth_pcm(B, steps)
    masks = [[1, 1, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]]
    for s in range(steps):
        if s == 2:
            eng.reset_batch_idx(1); ora.reset_batch_idx(1)
            eng.mimi_reset_batch_idx(1); ora.mimi_reset_batch_idx(1, side=0)
        mask = np.array(masks[s], dtype=np.uint8)
        act = mask.astype(bool)
        ec = eng.encode_step(pcm[s], mask)
        oc = ora.encode_step(pcm[s], mask)
        assert np.array_equal(ec[act], oc[act]), f"codes differ at step {s}"
        et, ep = eng.step_tokens(oc, mask)
        ot, op = ora.step_tokens(oc, mask)
        lg_e = eng.debug_read("lm.logits", B * cfg.text_out_vocab_size).reshape(B, -1)
        lg_o = ora.debug_read("lm.logits", B * cfg.text_out_vocab_size).reshape(B, -1)
        assert np.array_equal(lg_e[act].view(np.uint32), lg_o[act].view(np.uint32)), f"logits differ at step {s}"
        assert np.array_equal(et[act], ot[act]) and np.array_equal(ep[:, act].view(np.uint32), op[:, act].view(np.uint32))
        # decode the oracle's codes back to PCM on both sides (inactive slots: valid but unused codes)
        dc = np.where(act[:, None], oc, 0).astype(np.uint32)
        pe = eng.decode_step(dc, mask)
        po = ora.decode_step(dc, mask, side=0)
        assert np.array_equal(pe[act].view(np.uint32), po[act].view(np.uint32)), f"decoded PCM differs at step {s}"
    eng.close()
    ora.close()


@pytest.mark.parametrize("fuse_front", ["1", "0"])
def test_seanet_front_fused_and_as_three_launches(gpu, dsm, lib, orc, full_weights, monkeypatch, fuse_front):
    """The real Mimi front end — init conv (1 -> 64, k 7), residual block (64 -> 32 k 3 -> 64 k 1, skip) — runs as one kernel
    with its intermediates in LDS (`seanet_front_kernel`, default) or as three GEMM launches (DSM_FUSE_FRONT=0): both must give
    the oracle's latents and codes bit for bit over several frames (carried conv frames cross every frame boundary: the fused
    kernel reads them from the k-3 conv's buffer prefix and leaves the newest two frames in its tail), with a paused slot whose
    carried frames must not move, a mid-stream reset, and B = 5 so that tiles of different slots interleave."""
    from dsm_amd import synth
    cfg, (lm, mimi) = full_weights
    monkeypatch.setenv("DSM_FUSE_FRONT", fuse_front)
    B, steps = 5, 5
    eng = dsm.AsrEngine(cfg, B, lm, mimi)
    ora = orc.OracleAsr(cfg, B, lm, mimi)
    pcm = synth.synth_pcm(B, steps, seed=31)
    masks = np.ones((steps, B), dtype=np.uint8)
    masks[1, 2] = 0
    masks[2, 2] = 0
    masks[3, 0] = 0
    for s in range(steps):
        if s == 3:
            eng.mimi_reset_batch_idx(4); ora.mimi_reset_batch_idx(4, side=0)
        act = masks[s].astype(bool)
        ec = eng.encode_step(pcm[s], masks[s])
        oc = ora.encode_step(pcm[s], masks[s])
        lat_e = eng.debug_read("mimi.latent", B * cfg.mimi.dimension).reshape(B, -1)
        lat_o = ora.debug_read("mimi.latent", B * cfg.mimi.dimension).reshape(B, -1)
        assert np.array_equal(lat_e[act].view(np.uint32), lat_o[act].view(np.uint32)), f"latent differs at step {s}"
        assert np.array_equal(ec[act], oc[act]), f"codes differ at step {s}"
    eng.close()
    ora.close()


@pytest.mark.parametrize("dot_mode", [0, 1])
def test_full_size_steady_state_wrapped_ring(gpu, dsm, lib, orc, full_weights, dot_mode):
    """Both canonical dot products (dot_mode 1 is what bench.py times: gemm_bx3u_kernel's split-K slabs + the attention prologue).
    The benchmarked regime itself, at the real dimensions: every ring is jumped past its context (`debug_set_positions` on
    both sides: 3 x 750 + 11 LM frames, 3 x 250 + 5 Mimi frames — wrapped, every one of the 750 / 250 slots visible, write
    index mid-ring) and the engine is compared with the oracle on full-length attention — hd 128 bf16 with six pipelined
    iterations per phase in the LM, hd 64 f32 T = 2 in Mimi — through frames that overwrite ring slots, with a paused slot and
    the fused QKV prologue writing the new K/V row into a full ring.  Bits: latents, codes, LM hidden state, logits, tokens, VAD."""
    from dsm_amd import synth
    cfg, (lm, mimi) = full_weights
    cfg = type(cfg).from_buffer_copy(cfg)
    cfg.dot_mode = dot_mode
    B, steps = 3, 3
    eng = dsm.AsrEngine(cfg, B, lm, mimi)
    ora = orc.OracleAsr(cfg, B, lm, mimi)
    pcm = synth.synth_pcm(B, steps + 1, seed=41)
    ones = np.ones(B, dtype=np.uint8)
    # one ordinary frame first, so that ring rows 0.. hold real keys on both sides; the rest of the rings are zeros on both
    ec, et, ep = eng.step_pcm(pcm[0], ones)
    oc, ot, op = ora.step_pcm(pcm[0], ones)
    assert np.array_equal(ec, oc) and np.array_equal(et, ot)
    lm_pos, mimi_pos = 3 * cfg.lm.context + 11, 3 * cfg.mimi.transformer.context + 5
    eng.debug_set_positions(lm_pos, mimi_pos)
    ora.debug_set_positions(lm_pos, mimi_pos)
    masks = np.ones((steps, B), dtype=np.uint8)
    masks[1, 1] = 0
    for s in range(steps):
        act = masks[s].astype(bool)
        ec, et, ep = eng.step_pcm(pcm[1 + s], masks[s])
        oc, ot, op = ora.step_pcm(pcm[1 + s], masks[s])
        for tap, n in (("mimi1.latent", cfg.mimi.dimension), ("lm.hidden", cfg.lm.d_model), ("lm.logits", cfg.text_out_vocab_size)):
            ge = eng.debug_read(tap, B * n).reshape(B, -1)
            go = ora.debug_read(tap, B * n).reshape(B, -1)
            assert np.array_equal(ge[act].view(np.uint32), go[act].view(np.uint32)), f"{tap} differs at steady-state step {s}"
        assert np.array_equal(ec[act], oc[act]) and np.array_equal(et[act], ot[act]), f"codes / tokens differ at step {s}"
        assert np.array_equal(ep[:, act].view(np.uint32), op[:, act].view(np.uint32)), f"VAD differs at step {s}"
    eng.close()
    ora.close()


def test_stt_2_6b_en_shapes(gpu, dsm, lib, orc):
    """BASELINE.json configs[2]: stt-2.6b-en (48 layers, 32 heads x 64, ctx 375, vocab 4000, no extra heads)."""
    import os
    from dsm_amd import synth
    cfg = dsm.config_stt_2_6b_en()
    cfg.dot_mode = 0  # then the same step in dot_mode 1 below
    lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="stt-2.6b-en")
    B = 2
    eng = dsm.AsrEngine(cfg, B, lm, mimi)
    ora = orc.OracleAsr(cfg, B, lm, mimi)
    pcm = synth.synth_pcm(B, 2)
    mask = np.ones(B, dtype=np.uint8)
    for s in range(2):
        ec, et, _ = eng.step_pcm(pcm[s], mask)
        oc, ot, _ = ora.step_pcm(pcm[s], mask)
        assert np.array_equal(ec, oc) and np.array_equal(et, ot)
        hid_e = eng.debug_read("lm.hidden", B * cfg.lm.d_model)
        hid_o = ora.debug_read("lm.hidden", B * cfg.lm.d_model)
        assert np.array_equal(hid_e.view(np.uint32), hid_o.view(np.uint32))
    eng.close()
    ora.close()
    # BASELINE.json configs[2] is quoted at batch 128: one LM-only step at B = 34 (two stream groups of 32 + 2 slots;
    # the first runs bf16 MT = 2 GEMMs and 1024 attention workgroups of hd 64) — codes fed directly, the oracle skips Mimi
    B = 34
    rng = np.random.default_rng(9)
    codes = rng.integers(0, cfg.mimi.quantizer_bins, (B, cfg.audio_codebooks)).astype(np.uint32)
    mask = np.ones(B, dtype=np.uint8)
    mask[5] = 0
    act = mask.astype(bool)
    eng = dsm.AsrEngine(cfg, B, lm, mimi)
    assert eng.stream_groups() == [(0, 32), (32, 2)]
    et, _ = eng.step_tokens(codes, mask)
    hid_e = eng.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
    lg_e = eng.debug_read("lm.logits", B * cfg.text_out_vocab_size).reshape(B, -1)
    eng.close()
    ora = orc.OracleAsr(cfg, B, lm, mimi)
    ot, _ = ora.step_tokens(codes, mask)
    hid_o = ora.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
    lg_o = ora.debug_read("lm.logits", B * cfg.text_out_vocab_size).reshape(B, -1)
    ora.close()
    assert np.array_equal(hid_e[act].view(np.uint32), hid_o[act].view(np.uint32)), "B=34: lm.hidden differs"
    assert np.array_equal(lg_e[act].view(np.uint32), lg_o[act].view(np.uint32)), "B=34: logits differ"
    assert np.array_equal(et[act], ot[act])
    # the same step in dot_mode 1 (r03): MT = 2 and MT = 1 bx3 tiles, hd-64 bf16 attention with the fused prologue
    cfg1 = type(cfg).from_buffer_copy(cfg)
    cfg1.dot_mode = 1
    eng = dsm.AsrEngine(cfg1, B, lm, mimi)
    et, _ = eng.step_tokens(codes, mask)
    hid_e = eng.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
    eng.close()
    ora = orc.OracleAsr(cfg1, B, lm, mimi)
    ot, _ = ora.step_tokens(codes, mask)
    hid_o1 = ora.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
    ora.close()
    assert np.array_equal(hid_e[act].view(np.uint32), hid_o1[act].view(np.uint32)), "B=34, dot_mode 1: lm.hidden differs"
    assert np.array_equal(et[act], ot[act])
    assert not np.array_equal(hid_o1[act], hid_o[act])  # the two modes are different roundings of the same sums
    for p in (lm,):  # 5 GB: do not leave it in /tmp for the next test session
        os.remove(p)


def test_tts_v202501_shapes(gpu, dsm, lib, orc):
    """BASELINE.json configs[4] shapes: 2048-d x 16-layer main LM (32 input codebooks), depformer 1024-d x 4 layers over
    32 slices with 11 weight groups and rank-128 embeddings.  text_audio_delay_in_tokens is shortened (25 -> 3) so that
    ten steps reach the phase where every codebook feeds back generated tokens."""
    import os
    from dsm_amd import synth
    cfg = dsm.config_tts_v202501()
    cfg.dot_mode = 0  # dot_mode 1 at these shapes: test_tts_ca_gpu.py::test_cross_attention_and_guidance_at_v202501_shapes
    cfg.text_audio_delay_in_tokens, cfg.max_steps = 3, 64
    path = synth.make_synth_tts_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tts-v202501")
    B = 2
    eng = dsm.TtsEngine(cfg, B, path)
    ora = orc.OracleTts(cfg, B, path)
    rng = np.random.default_rng(2)
    for s in range(10):
        prev = rng.integers(0, cfg.text_in_vocab_size, B).astype(np.uint32)
        allowed = np.array([int(rng.integers(4, 8000)), dsm.TTS_ALLOW_PAD_OR_EPAD], dtype=np.int32)
        mask = np.array([1, 0 if s == 4 else 1], dtype=np.uint8)
        act = mask.astype(bool)
        te, ae = eng.step(prev, allowed, mask)
        to, ao = ora.step(prev, allowed, mask)
        he = eng.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
        ho = ora.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
        assert np.array_equal(he[act].view(np.uint32), ho[act].view(np.uint32)), f"LM hidden differs at step {s}"
        assert np.array_equal(te[act], to[act]), f"text tokens differ at step {s}"
        assert np.array_equal(ae[act], ao[act]), f"depformer tokens differ at step {s}"
    for b in range(B):
        for i in range(eng.step_idx(b)):
            assert np.array_equal(eng.audio_tokens(b, i), ora.audio_tokens(b, i))
    eng.close()
    ora.close()
    # BASELINE.json configs[4] is quoted at batch 32: two steps at B = 32 with the text-audio delay removed, so that the
    # batched depformer (32 slices x 32 slots: MT = 2 GEMMs, 16 heads x 32 slots of attention over the slice axis) runs
    # from the first step
    cfg.text_audio_delay_in_tokens = 0
    B = 32
    eng = dsm.TtsEngine(cfg, B, path)
    ora = orc.OracleTts(cfg, B, path)
    for s in range(2):
        prev = rng.integers(0, cfg.text_in_vocab_size, B).astype(np.uint32)
        allowed = np.where(np.arange(B) % 3 == 0, dsm.TTS_ALLOW_PAD_OR_EPAD, rng.integers(4, 8000, B)).astype(np.int32)
        mask = np.ones(B, dtype=np.uint8)
        mask[7] = 0 if s == 1 else 1
        act = mask.astype(bool)
        te, ae = eng.step(prev, allowed, mask)
        to, ao = ora.step(prev, allowed, mask)
        he = eng.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
        ho = ora.debug_read("lm.hidden", B * cfg.lm.d_model).reshape(B, -1)
        assert np.array_equal(he[act].view(np.uint32), ho[act].view(np.uint32)), f"B=32: LM hidden differs at step {s}"
        assert np.array_equal(te[act], to[act]), f"B=32: text tokens differ at step {s}"
        assert np.array_equal(ae[act], ao[act]), f"B=32: depformer tokens differ at step {s}"
    eng.close()
    ora.close()
    os.remove(path)
