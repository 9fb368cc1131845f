"""Independent cross-check of the oracle's delayed-streams LM step against Hugging Face's
`KyutaiSpeechToTextModel` (a separate PyTorch implementation of the stt checkpoints' decoder; defaults = stt-2.6b-en).
A small random model is exported under the reference's key map (bf16-representable values, q/k rows re-interleaved for
`rope_i`); the oracle then steps through a token stream one frame at a time — its own previous argmax as text input,
the previous frame's codes as audio input — and the logits of every step are compared
with HF's single causal pass over the same token matrix.  Ties the embedding sum, RMSNorm (eps 1e-8), interleaved RoPE,
sliding-window attention, SiLU gating order, output norm and text head to an implementation nobody here wrote."""
import os

import numpy as np
import pytest

from test_oracle_vs_hf_mimi import interleave_rows, write_f32_safetensors

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")


def to_bf16_values(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    u = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def test_lm_steps_match_hf_causal_pass(dsm, orc, tmp_path):
    from transformers import KyutaiSpeechToTextConfig
    from transformers.models.kyutai_speech_to_text.modeling_kyutai_speech_to_text import KyutaiSpeechToTextModel
    from dsm_amd import synth
    cfg = dsm.config_tiny()
    cfg.extra_heads_num = 0
    cfg.kv_bf16 = 0  # f32 ring cache: HF keeps K/V in the compute dtype
    cfg.lm.max_period = 10000
    cfg.lm.context = 64  # HF's single pass is plain causal attention (its sliding window lives in the generation cache)
    cfg.text_out_vocab_size = cfg.text_in_vocab_size  # HF's head spans the input vocabulary
    t, nc, V, cv = cfg.lm, cfg.audio_codebooks, cfg.text_in_vocab_size, cfg.audio_vocab_size
    d, H, hid = t.d_model, t.num_heads, synth.gating_hidden(cfg.lm)
    hf_cfg = KyutaiSpeechToTextConfig(
        vocab_size=V, codebook_vocab_size=cv, hidden_size=d, num_hidden_layers=t.num_layers, num_attention_heads=H,
        num_key_value_heads=H, head_dim=d // H, ffn_dim=2 * hid, num_codebooks=nc, sliding_window=t.context,
        max_position_embeddings=256, audio_pad_token_id=V + nc * cv, audio_bos_token_id=cv - 1, bos_token_id=V - 1,
        pad_token_id=3)  # KyutaiSpeechToTextModel has no codec inside; the default codec_config is never instantiated
    torch.manual_seed(5)
    hf = KyutaiSpeechToTextModel(hf_cfg).eval()
    g = torch.Generator().manual_seed(11)
    sd = hf.state_dict()
    for k in sd:  # bigger-than-init weights so that every block matters, rounded to bf16-representable values
        w = torch.randn(sd[k].shape, generator=g) * (0.3 if "embed" in k else 1.0 / np.sqrt(sd[k].shape[-1]))
        if k.endswith("layernorm.weight") or k == "norm.weight":
            w = 1.0 + 0.1 * torch.randn(sd[k].shape, generator=g)
        sd[k] = torch.from_numpy(to_bf16_values(w.numpy()))
    hf.load_state_dict(sd)
    ref = {}
    table = sd["embed_tokens.embed_tokens.weight"].numpy()
    ref["text_emb.weight"] = table[:V]
    for i in range(nc):
        ref[f"emb.{i}.weight"] = table[V + i * cv: V + (i + 1) * cv]
    for l in range(t.num_layers):
        p, q = f"layers.{l}", f"transformer.layers.{l}"
        w = {n: sd[f"{p}.self_attn.{n}_proj.linear.weight"].numpy() for n in "qkvo"}
        ref[f"{q}.self_attn.in_proj_weight"] = np.concatenate([interleave_rows(w["q"], H), interleave_rows(w["k"], H), w["v"]])
        ref[f"{q}.self_attn.out_proj.weight"] = w["o"]
        ref[f"{q}.norm1.alpha"] = sd[f"{p}.input_layernorm.weight"].numpy().reshape(1, 1, d)
        ref[f"{q}.norm2.alpha"] = sd[f"{p}.post_attention_layernorm.weight"].numpy().reshape(1, 1, d)
        ref[f"{q}.gating.linear_in.weight"] = sd[f"{p}.mlp.fc1.weight"].numpy()
        ref[f"{q}.gating.linear_out.weight"] = sd[f"{p}.mlp.fc2.weight"].numpy()
    ref["out_norm.alpha"] = sd["norm.weight"].numpy().reshape(1, 1, d)
    head = to_bf16_values((torch.randn((V, d), generator=g) / np.sqrt(d)).numpy())
    ref["text_linear.weight"] = head
    lm_path = os.path.join(tmp_path, "hf.lm.safetensors")
    write_f32_safetensors(lm_path, ref)
    _, mimi_path = synth.make_synth_weights(dsm.config_tiny(), os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="tiny")

    steps, B = 30, 2
    rng = np.random.default_rng(3)
    codes = rng.integers(0, cv - 1, (steps, B, nc)).astype(np.uint32)
    ora = orc.OracleAsr(cfg, B, lm_path, mimi_path)
    mask = np.ones(B, dtype=np.uint8)
    text_in = np.full((steps, B), V - 1, dtype=np.int64)       # text_start_token on the first step (core/asr.rs:139-143)
    audio_in = np.full((steps, B, nc), cv - 1, dtype=np.int64)  # audio pad token on the first step (core/asr.rs:165-176)
    got = []
    for s in range(steps):
        tok, _ = ora.step_tokens(codes[s], mask)
        got.append(ora.debug_read("lm.logits", B * V).reshape(B, V).copy())
        if s + 1 < steps:
            text_in[s + 1] = tok
            audio_in[s + 1] = codes[s]
    ora.close()
    ids = torch.from_numpy(np.concatenate([text_in[:, :, None], audio_in], axis=2).transpose(1, 0, 2))  # [B, T, 1+nc]
    with torch.no_grad():
        hidden = hf(input_ids=ids).last_hidden_state  # [B, T, d]
        want = (hidden @ torch.from_numpy(head).T).numpy()
    worst = 0.0
    for s in range(steps):
        err = np.abs(got[s] - want[:, s, :]).max() / max(1.0, np.abs(want[:, s, :]).max())
        worst = max(worst, float(err))
        assert err <= 1e-5, f"logits of step {s} differ from HF by {err}"
        assert np.array_equal(got[s].argmax(-1), want[:, s, :].argmax(-1))
    print(f"worst relative logit error vs HF over {steps} steps: {worst:.2e}")
