"""CPU tests of the TTS branch the reference server runs (srv/tts.rs:426-441): cross-attention to a speaker source
(core/transformer.rs:205-363,747-763; LmModel::forward_ca core/lm.rs:1016-1067) and classifier-free guidance
(core/tts_streaming.rs:164-173,207-214; DepFormer::sample_cfg core/lm.rs:686-733), as restated in oracle/dsm_oracle.c and
oracle/dsm_oracle_tts.inc.  No Candle output exists offline (parity unpinned): the float side is pinned here against an
independent float64 numpy evaluation of one whole main-LM step, the control flow against properties the reference's code
implies (a source-less slot takes the `None` path; cfg_alpha = 1 is the unguided model; the mix formula)."""
import os

import numpy as np
import pytest

from tts_schedule import schedule

WDIR = os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights")


def _cfgs(dsm, **kw):
    from dsm_amd import synth
    cfg = dsm.config_tts_tiny(cross_attention=True, **kw)
    tag = "tts_tiny_ca" + ("_kvd%d" % cfg.ca_dim if cfg.ca_dim else "") + ("_rms" if cfg.ca_norm else "")
    return cfg, synth.make_synth_tts_weights(cfg, WDIR, tag=tag)


def test_a_slot_without_a_source_takes_the_none_path(dsm, orc):
    """(Some(cross_attn), None) => xs (core/transformer.rs:753-760): the cross-attention model with no ca_src set emits what
    the plain model emits (the synthetic tensors are generated per name, so both files hold the same shared weights)."""
    from dsm_amd import synth
    cfg_ca, path_ca = _cfgs(dsm)
    cfg = dsm.config_tts_tiny()
    path = synth.make_synth_tts_weights(cfg, WDIR, tag="tts_tiny")
    B, steps = 3, 14
    a, b = orc.OracleTts(cfg_ca, B, path_ca), orc.OracleTts(cfg, B, path)
    for prev, allowed, mask in schedule(cfg, B, steps):
        ta, aa = a.step(prev, allowed, mask)
        tb, ab = b.step(prev, allowed, mask)
        act = mask.astype(bool)
        assert np.array_equal(ta[act], tb[act]) and np.array_equal(aa[act], ab[act])
        ha, hb = a.debug_read("lm.hidden", B * cfg.lm.d_model), b.debug_read("lm.hidden", B * cfg.lm.d_model)
        assert np.array_equal(ha.reshape(B, -1)[act], hb.reshape(B, -1)[act])
    a.close(); b.close()


def _np_main_lm_step0(cfg, W, text_tok, src):
    """float64 evaluation of LmModel::forward_ca for ONE row at step 0 (one self-attention key: softmax = 1; RoPE at
    position 0 is the identity), written from the reference's formulas, not from the oracle."""
    t = cfg.lm
    d, H = t.d_model, t.num_heads
    hd = d // H
    pad = cfg.audio_vocab_size - 1
    x = W["text_emb.weight"][text_tok].astype(np.float64)
    for cb in range(cfg.audio_codebooks):  # step 0: every codebook feeds the audio pad token (core/tts_streaming.rs:126-147)
        x = x + W[f"emb.{cb}.weight"][pad]
    rms = lambda v, a: v / np.sqrt(np.mean(v * v) + 1e-8) * a.reshape(-1)
    hid = (11 * d // 4) if t.dim_feedforward == 4 * d else 2 * t.dim_feedforward // 3
    for l in range(t.num_layers):
        p = f"transformer.layers.{l}"
        n1 = rms(x, W[f"{p}.norm1.alpha"])
        qkv = W[f"{p}.self_attn.in_proj_weight"].astype(np.float64) @ n1
        v = qkv[2 * d:]
        x = x + W[f"{p}.self_attn.out_proj.weight"].astype(np.float64) @ v
        if src is not None:
            if cfg.ca_norm == 1:
                xc = rms(x, W[f"{p}.norm_cross.alpha"])
            else:
                mu = x.mean()
                xc = (x - mu) / np.sqrt(((x - mu) ** 2).mean() + 1e-5) * W[f"{p}.norm_cross.weight"] + W[f"{p}.norm_cross.bias"]
            q = (W[f"{p}.cross_attention.in_proj_weight_q"].astype(np.float64) @ xc).reshape(H, hd)
            kv = (src.astype(np.float64) @ W[f"{p}.cross_attention.in_proj_weight_kv"].astype(np.float64).T).reshape(len(src), 2, H, hd)
            k, vv = kv[:, 0], kv[:, 1]  # (t, H, hd)
            out = np.zeros((H, hd))
            for h in range(H):
                s = (k[:, h] @ q[h]) * hd ** -0.5
                w = np.exp(s - s.max())
                w /= w.sum()
                out[h] = w @ vv[:, h]
            x = x + W[f"{p}.cross_attention.out_proj.weight"].astype(np.float64) @ out.reshape(d)
        n2 = rms(x, W[f"{p}.norm2.alpha"])
        g = W[f"{p}.gating.linear_in.weight"].astype(np.float64) @ n2
        act = g[:hid] / (1 + np.exp(-g[:hid])) * g[hid:]
        x = x + W[f"{p}.gating.linear_out.weight"].astype(np.float64) @ act
    return rms(x, W["out_norm.alpha"])


@pytest.mark.parametrize("kw", [dict(), dict(ca_dim=40), dict(ca_norm=1)])
def test_cross_attention_against_numpy(dsm, orc, kw):
    from dsm_amd import synth
    cfg, path = _cfgs(dsm, **kw)
    cfg.kv_bf16 = 0  # f32 caches: the only rounding left is f32 arithmetic
    W = synth.read_safetensors(path)
    o = orc.OracleTts(cfg, 2, path)
    src0, src1 = synth.synth_ca_src(cfg, 9, 1), synth.synth_ca_src(cfg, 24, 2)
    o.set_ca_src(0, src0)
    o.set_ca_src(1, src1)
    o.step([5, 17], [3, 3], [1, 1])
    got = o.debug_read("lm.hidden", 2 * cfg.lm.d_model).reshape(2, -1)
    for row, (tok, src) in enumerate([(5, src0), (17, src1)]):
        want = _np_main_lm_step0(cfg, W, tok, src)
        assert np.max(np.abs(got[row] - want)) < 2e-4 * max(1.0, np.max(np.abs(want))), row
        # and the source matters: without it the row is far away
        assert np.max(np.abs(_np_main_lm_step0(cfg, W, tok, None) - want)) > 1e-2
    o.close()


def test_legacy_single_in_proj_weight_layout(dsm, orc):
    """"Case 1" checkpoints (one cross_attention.in_proj_weight, rows q | k | v: core/transformer.rs:232-252) load to the
    same model as the split layout holding the same rows."""
    from dsm_amd import synth
    cfg, _ = _cfgs(dsm)
    legacy = synth.make_synth_tts_weights(cfg, WDIR, tag="tts_tiny_ca_legacy", legacy_ca=True)
    W = synth.read_safetensors(legacy)
    d = cfg.lm.d_model
    # rewrite the legacy file's rows in the split layout
    split = os.path.join(WDIR, "tts_tiny_ca_from_legacy.lm.safetensors")
    if not os.path.exists(split):
        import json, struct
        items = {}
        for k, v in W.items():
            if k.endswith("cross_attention.in_proj_weight"):
                items[k + "_q"], items[k + "_kv"] = v[:d], v[d:]
            else:
                items[k] = v
        header, off, blobs = {}, 0, []
        for k, v in items.items():
            raw = synth.f32_to_bf16_bits(v).tobytes()
            header[k] = {"dtype": "BF16", "shape": list(v.shape), "data_offsets": [off, off + len(raw)]}
            off += len(raw)
            blobs.append(raw)
        hj = json.dumps(header, separators=(",", ":")).encode()
        hj += b" " * ((8 - len(hj) % 8) % 8)
        with open(split, "wb") as f:
            f.write(struct.pack("<Q", len(hj)) + hj + b"".join(blobs))
    a, b = orc.OracleTts(cfg, 1, legacy), orc.OracleTts(cfg, 1, split)
    src = synth.synth_ca_src(cfg, 11, 3)
    a.set_ca_src(0, src); b.set_ca_src(0, src)
    for s in range(8):
        ta, aa = a.step([4 + s], [dsm.TTS_ALLOW_PAD_OR_EPAD], [1])
        tb, ab = b.step([4 + s], [dsm.TTS_ALLOW_PAD_OR_EPAD], [1])
        assert np.array_equal(ta, tb) and np.array_equal(aa, ab)
        assert np.array_equal(a.debug_read("lm.hidden", d), b.debug_read("lm.hidden", d))
    a.close(); b.close()


def test_cfg_alpha_one_is_the_unguided_model_and_the_mix_formula(dsm, orc):
    from dsm_amd import synth
    cfg2, path = _cfgs(dsm, cfg_rows=True)
    cfg1, _ = _cfgs(dsm)
    B, steps = 2, 16
    guided1, plain, guided3 = orc.OracleTts(cfg2, B, path), orc.OracleTts(cfg1, B, path), orc.OracleTts(cfg2, B, path)
    srcs = [synth.synth_ca_src(cfg1, 7 + 5 * b, 20 + b) for b in range(B)]
    empty = synth.synth_ca_src(cfg1, 7, 99)  # SpeakerEncoder::empty(): the unconditional row's source
    for b in range(B):
        guided1.set_ca_src(b, srcs[b], empty, 1.0)  # l0 * 1 - l1 * 0 = l0
        plain.set_ca_src(b, srcs[b])
        guided3.set_ca_src(b, srcs[b], empty, 3.0)
    V = cfg1.text_out_vocab_size
    differs = False
    for s, (prev, allowed, mask) in enumerate(schedule(cfg1, B, steps)):
        allowed[:] = dsm.TTS_ALLOW_PAD_OR_EPAD if s % 2 else allowed
        t1, a1 = guided1.step(prev, allowed, mask)
        tp, ap = plain.step(prev, allowed, mask)
        t3, a3 = guided3.step(prev, allowed, mask)
        act = mask.astype(bool)
        assert np.array_equal(t1[act], tp[act]) and np.array_equal(a1[act], ap[act]), f"alpha = 1 differs from the unguided model at step {s}"
        differs = differs or not np.array_equal(a3[act], ap[act])
        # rows 2b / 2b + 1 of the guided engine: the conditional row is the unguided model's row while the histories agree
        if s == 0:
            rows = guided3.debug_read("lm.logits", 2 * B * V).reshape(B, 2, V)
            base = plain.debug_read("lm.logits", B * V).reshape(B, V)
            assert np.array_equal(rows[:, 0][act], base[act])
            if s % 2:
                continue
        if s % 2:  # PadOrEpad: the text token is decided by the argmax of l0 * a - l1 * (a - 1) in f32 (affine(mul, 0) each)
            rows = guided3.debug_read("lm.logits", 2 * B * V).reshape(B, 2, V)
            fa, fb = np.float32(3.0), np.float32(3.0 - 1.0)
            mix = (rows[:, 0] * fa + np.float32(0)) - (rows[:, 1] * fb + np.float32(0))
            want = np.where(mix.argmax(1) == cfg1.text_pad_token, cfg1.text_pad_token, cfg1.text_eop_token)
            forced = None
            for b in range(B):
                if act[b] and t3[b] != want[b]:
                    # the only other outcome is the forced end-of-pad after max_consecutive_pads
                    assert t3[b] == cfg1.text_eop_token
    assert differs, "cfg_alpha = 3 never changed a token: the guidance is not wired"
    guided1.close(); plain.close(); guided3.close()


def test_reset_clears_source_and_guidance(dsm, orc):
    from dsm_amd import synth
    cfg2, path = _cfgs(dsm, cfg_rows=True)
    a, b = orc.OracleTts(cfg2, 1, path), orc.OracleTts(cfg2, 1, path)
    src, empty = synth.synth_ca_src(cfg2, 12, 5), synth.synth_ca_src(cfg2, 12, 6)
    a.set_ca_src(0, src, empty, 2.0)
    for s in range(6):
        a.step([7], [dsm.TTS_ALLOW_PAD_OR_EPAD], [1])
    a.reset_batch_idx(0)  # a fresh State: no source, no guidance until set again
    for s in range(8):
        ta, aa = a.step([9 + s], [11], [1])
        tb, ab = b.step([9 + s], [11], [1])
        assert np.array_equal(ta, tb) and np.array_equal(aa, ab)
    a.close(); b.close()
