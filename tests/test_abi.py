"""The C-ABI library: loads without a GPU, exports every symbol include/dsm.h declares, carries the shipped
presets, and refuses to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(dsm, lib):
    hdr = open(os.path.join(ROOT, "include", "dsm.h")).read()
    declared = set(re.findall(r"\b(dsm_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dsm_asr_msg", "dsm_metrics"}
    assert declared == set(dsm.ABI_SYMBOLS), declared ^ set(dsm.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), f"libdsm_mi355x.so does not export {s}"


def test_struct_sizes_match_the_header(dsm, lib):
    # int-only structs: sizes follow from the field counts in include/dsm.h
    assert C.sizeof(dsm.TransformerConfig) == 11 * 4
    assert C.sizeof(dsm.MimiConfig) == (5 + 8 + 5) * 4 + 11 * 4 + 4 * 4
    assert C.sizeof(dsm.AsrConfig) == 11 * 4 + 8 * 4 + C.sizeof(dsm.MimiConfig) + 4 + 4  # ... kv_bf16, dot_mode (r03)


def test_presets_match_the_shipped_tomls(dsm, lib):
    """configs/stt/config-stt-en_fr-hf.toml:18-56 and config-stt-en-hf.toml:18-49; Mimi v0_1 core/mimi.rs:32-93."""
    c = dsm.config_stt_1b_en_fr()
    assert (c.lm.d_model, c.lm.num_heads, c.lm.num_layers, c.lm.dim_feedforward) == (2048, 16, 16, 8192)
    assert (c.lm.context, c.lm.max_period, c.lm.gating, c.lm.norm, c.lm.positional_embedding) == (750, 100000, 1, 1, 1)
    assert (c.text_in_vocab_size, c.text_out_vocab_size, c.audio_vocab_size, c.audio_codebooks) == (8001, 8000, 2049, 32)
    assert (c.extra_heads_num, c.extra_heads_dim, c.asr_delay_in_tokens) == (4, 6, 6)
    c2 = dsm.config_stt_2_6b_en()
    assert (c2.lm.num_heads, c2.lm.num_layers, c2.lm.context) == (32, 48, 375)
    assert (c2.text_in_vocab_size, c2.text_out_vocab_size, c2.extra_heads_num, c2.asr_delay_in_tokens) == (4001, 4000, 0, 32)
    m = c.mimi
    assert (m.dimension, m.n_filters, m.n_ratios, list(m.ratios)[:4]) == (512, 64, 4, [8, 6, 5, 4])
    assert (m.kernel_size, m.residual_kernel_size, m.last_kernel_size, m.compress) == (7, 3, 3, 2)
    t = m.transformer
    assert (t.d_model, t.num_heads, t.num_layers, t.dim_feedforward, t.context, t.max_period) == (512, 8, 8, 2048, 250, 10000)
    assert (t.gating, t.norm, t.layer_scale, t.conv_layout) == (0, 0, 1, 1)
    assert (m.quantizer_n_q, m.quantizer_bins, m.quantizer_dim, m.downsample_stride) == (32, 2048, 256, 2)


def test_create_fails_loudly_without_a_device_or_files(dsm, lib, tiny_weights):
    import torch
    cfg = dsm.config_tiny()
    if not torch.cuda.is_available():
        with pytest.raises(dsm.DsmError, match="no usable HIP device|no CPU fallback"):
            dsm.AsrEngine(cfg, 2, *tiny_weights)
    # (temperature > 0 is accepted since r04: seeded Gumbel sampling, tests/test_parity_gpu.py)
    with pytest.raises(dsm.DsmError):
        dsm.AsrEngine(cfg, 2, "/nonexistent/lm.safetensors", tiny_weights[1])


def test_safetensors_key_map(dsm, tiny_weights):
    """The synthetic checkpoints carry the reference's tensor names."""
    from dsm_amd import synth
    lm = synth.read_safetensors(tiny_weights[0])
    mimi = synth.read_safetensors(tiny_weights[1])
    for k in ("text_emb.weight", "emb.0.weight", "transformer.layers.0.self_attn.in_proj_weight",
              "transformer.layers.0.self_attn.out_proj.weight", "transformer.layers.1.norm1.alpha",
              "transformer.layers.0.gating.linear_in.weight", "transformer.layers.0.gating.linear_out.weight",
              "out_norm.alpha", "text_linear.weight", "extra_heads.1.weight"):
        assert k in lm, k
    for k in ("encoder.model.0.conv.conv.weight", "encoder.model.1.block.1.conv.conv.weight",
              "encoder.model.3.conv.conv.bias", "encoder.model.14.conv.conv.weight",
              "decoder.model.2.convtr.convtr.weight", "decoder.model.14.conv.conv.weight",
              "encoder_transformer.transformer.layers.0.linear1.weight",
              "encoder_transformer.transformer.layers.0.layer_scale_1.scale",
              "downsample.conv.conv.conv.weight", "upsample.convtr.convtr.convtr.weight",
              "quantizer.rvq_first.input_proj.weight", "quantizer.rvq_rest.vq.layers.2._codebook.embedding_sum",
              "quantizer.rvq_first.vq.layers.0._codebook.cluster_usage"):
        assert k in mimi, k
    assert lm["transformer.layers.0.norm1.alpha"].shape == (1, 1, 128)


def test_presets_select_dot_mode_1_and_the_reference_tts_branch(dsm, lib):
    """r04: the shipped presets are what bench.py times (dot_mode 1) and, for TTS, the branch the reference server runs
    (tts_202501: cross attention in every layer, core/lm.rs:392-396)."""
    assert dsm.config_stt_1b_en_fr().dot_mode == 1 and dsm.config_stt_2_6b_en().dot_mode == 1
    t = dsm.config_tts_v202501()
    assert (t.dot_mode, t.cross_attention, t.ca_norm, t.ca_dim, t.ca_max_len, t.cfg_rows) == (1, 1, 0, 0, 128, 0)
