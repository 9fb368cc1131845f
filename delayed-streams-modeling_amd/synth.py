"""Seeded synthetic checkpoints with the reference's exact safetensors key map.

No real weights exist offline (the tomls point at hf:// downloads), so tests and bench use
Philox-seeded random weights of the real shapes: linear/conv N(0, 1/fan_in), norm alphas
1 + N(0, 0.02), layer scales 0.01, codebooks embedding_sum ~ N(0, s_i^2) with cluster_usage 1.
Weight-norm is stored pre-folded under the `weight` key, which the reference accepts
(core/conv.rs:35-36).  The LM file is BF16 (the reference's checkpoint dtype), Mimi is F32.
"""
import json
import os
import struct
import zlib

import numpy as np

SEED = 0xD5A1


def f32_to_bf16_bits(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    u = u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))
    return (u >> np.uint32(16)).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32)


def _rng(name, seed):
    return np.random.Generator(np.random.Philox(key=(seed << 32) ^ zlib.crc32(name.encode())))


class Spec:
    weight_norm = False

    def __init__(self):
        self.items = []  # (name, shape, kind, param)

    def add(self, name, shape, kind, param=None):
        self.items.append((name, tuple(int(s) for s in shape), kind, param))


def _gen(name, shape, kind, param, seed):
    r = _rng(name, seed)
    n = int(np.prod(shape)) if len(shape) else 1
    if kind == "normal":  # param = std
        return (r.standard_normal(n, dtype=np.float32) * np.float32(param)).reshape(shape)
    if kind == "alpha":
        return (1.0 + 0.02 * r.standard_normal(n, dtype=np.float32)).astype(np.float32).reshape(shape)
    if kind == "const":
        return np.full(shape, param, dtype=np.float32)
    if kind == "gain":  # weight-norm gains: row norms of ~1 (unit-variance layer outputs), spread by `param`
        return (1.0 + param * (2.0 * r.random(n, dtype=np.float32) - 1.0)).astype(np.float32).reshape(shape)
    raise ValueError(kind)


def write_safetensors(path, spec, dtype, seed=SEED):
    """Streams tensors to disk one at a time (the 1b LM file is ~2 GB)."""
    esize = 2 if dtype == "BF16" else 4
    header, off = {}, 0
    for name, shape, _, _ in spec.items:
        n = int(np.prod(shape)) if len(shape) else 1
        header[name] = {"dtype": dtype, "shape": list(shape), "data_offsets": [off, off + n * esize]}
        off += n * esize
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(struct.pack("<Q", len(hj)))
        f.write(hj)
        for name, shape, kind, param in spec.items:
            a = _gen(name, shape, kind, param, seed)
            f.write(f32_to_bf16_bits(a).tobytes() if dtype == "BF16" else a.tobytes())
    os.replace(tmp, path)


def read_safetensors(path):
    """{name: float32 ndarray}; BF16 is upcast exactly.  Pure data parsing, executes nothing."""
    with open(path, "rb") as f:
        (hlen,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(hlen))
    base = 8 + hlen
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    out = {}
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        a, b = meta["data_offsets"]
        raw = mm[base + a:base + b]
        if meta["dtype"] == "F32":
            arr = np.frombuffer(raw, dtype=np.float32)
        elif meta["dtype"] == "BF16":
            arr = bf16_bits_to_f32(np.frombuffer(raw, dtype=np.uint16))
        else:
            raise ValueError(f"unsupported dtype {meta['dtype']} for {name}")
        out[name] = arr.reshape(meta["shape"])
    return out


def _transformer_spec(s, prefix, t, hidden):
    d = t.d_model
    for l in range(t.num_layers):
        p = f"{prefix}.layers.{l}"
        s.add(f"{p}.self_attn.in_proj_weight", (3 * d, d), "normal", d ** -0.5)
        s.add(f"{p}.self_attn.out_proj.weight", (d, d), "normal", d ** -0.5)
        for n in ("norm1", "norm2"):
            if t.norm == 1:
                s.add(f"{p}.{n}.alpha", (1, 1, d), "alpha")
            else:
                s.add(f"{p}.{n}.weight", (d,), "alpha")
                s.add(f"{p}.{n}.bias", (d,), "normal", 0.02)
        if t.gating:
            s.add(f"{p}.gating.linear_in.weight", (2 * hidden, d), "normal", d ** -0.5)
            s.add(f"{p}.gating.linear_out.weight", (d, hidden), "normal", hidden ** -0.5)
        else:
            s.add(f"{p}.linear1.weight", (hidden, d), "normal", d ** -0.5)
            s.add(f"{p}.linear2.weight", (d, hidden), "normal", hidden ** -0.5)
        if t.layer_scale:
            s.add(f"{p}.layer_scale_1.scale", (d,), "const", 0.01)
            s.add(f"{p}.layer_scale_2.scale", (d,), "const", 0.01)


def gating_hidden(t):
    return 11 * t.d_model // 4 if t.dim_feedforward == 4 * t.d_model else 2 * t.dim_feedforward // 3


def lm_spec(cfg):
    s = Spec()
    t = cfg.lm
    d = t.d_model
    s.add("text_emb.weight", (cfg.text_in_vocab_size, d), "normal", 0.3)
    for i in range(cfg.audio_codebooks):
        s.add(f"emb.{i}.weight", (cfg.audio_vocab_size, d), "normal", 0.3)
    _transformer_spec(s, "transformer", t, gating_hidden(t) if t.gating else t.dim_feedforward)
    s.add("out_norm.alpha", (1, 1, d), "alpha")
    s.add("text_linear.weight", (cfg.text_out_vocab_size, d), "normal", d ** -0.5)
    for i in range(cfg.extra_heads_num):
        s.add(f"extra_heads.{i}.weight", (cfg.extra_heads_dim, d), "normal", d ** -0.5)
    return s


def tts_spec(cfg, legacy_ca=False):
    """TTS checkpoint keys (core/lm.rs:501-590): main LM + shared depformer with per-group gating/linear_in.
    cfg.cross_attention adds every layer's norm_cross + cross_attention tensors (legacy_ca: the single in_proj_weight layout)."""
    s = Spec()
    t, dp = cfg.lm, cfg.depformer
    d, D, S, G, lr = t.d_model, dp.d_model, cfg.dep_num_slices, cfg.dep_weight_groups, cfg.dep_low_rank
    s.add("text_emb.weight", (cfg.text_in_vocab_size, d), "normal", 0.3)
    for i in range(cfg.audio_codebooks):
        s.add(f"emb.{i}.weight", (cfg.audio_vocab_size, d), "normal", 0.3)
    _transformer_spec(s, "transformer", t, gating_hidden(t))
    if cfg.cross_attention:  # norm_cross + cross_attention per layer (core/transformer.rs:747-763, :205-290 "Case 2" keys)
        kvd = cfg.ca_dim or d
        for l in range(t.num_layers):
            p = f"transformer.layers.{l}"
            if cfg.ca_norm == 1:
                s.add(f"{p}.norm_cross.alpha", (1, 1, d), "alpha")
            else:
                s.add(f"{p}.norm_cross.weight", (d,), "alpha")
                s.add(f"{p}.norm_cross.bias", (d,), "normal", 0.02)
            if legacy_ca and kvd == d:  # "Case 1": one in_proj_weight [d + 2d, d], rows q | k | v
                s.add(f"{p}.cross_attention.in_proj_weight", (3 * d, d), "normal", d ** -0.5)
            else:
                s.add(f"{p}.cross_attention.in_proj_weight_q", (d, d), "normal", d ** -0.5)
                s.add(f"{p}.cross_attention.in_proj_weight_kv", (2 * d, kvd), "normal", kvd ** -0.5)
            s.add(f"{p}.cross_attention.out_proj.weight", (d, d), "normal", d ** -0.5)
    s.add("out_norm.alpha", (1, 1, d), "alpha")
    s.add("text_linear.weight", (cfg.text_out_vocab_size, d), "normal", d ** -0.5)
    for g in range(G):
        s.add(f"depformer_in.{g}.weight", (D, d), "normal", d ** -0.5)
    w = lr if lr > 0 else D
    s.add("depformer_text_emb.weight", (cfg.text_in_vocab_size, w), "normal", 0.5)
    if lr > 0:
        s.add("depformer_text_emb.low_rank.weight", (D, lr), "normal", lr ** -0.5)
    for k in range(S - 1):
        s.add(f"depformer_emb.{k}.weight", (cfg.audio_vocab_size, w), "normal", 0.5)
        if lr > 0:
            s.add(f"depformer_emb.{k}.low_rank.weight", (D, lr), "normal", lr ** -0.5)
    for k in range(S):
        s.add(f"linears.{k}.weight", (cfg.audio_vocab_size - 1, D), "normal", D ** -0.5)
    hid = gating_hidden(dp)
    for l in range(dp.num_layers):
        p = f"depformer.layers.{l}"
        s.add(f"{p}.self_attn.in_proj_weight", (3 * D, D), "normal", D ** -0.5)
        s.add(f"{p}.self_attn.out_proj.weight", (D, D), "normal", D ** -0.5)
        s.add(f"{p}.norm1.alpha", (1, 1, D), "alpha")
        s.add(f"{p}.norm2.alpha", (1, 1, D), "alpha")
        for g in range(G):
            s.add(f"{p}.gating.{g}.linear_in.weight", (2 * hid, D), "normal", D ** -0.5)
            s.add(f"{p}.gating.{g}.linear_out.weight", (D, hid), "normal", hid ** -0.5)
    return s


def make_synth_tts_weights(cfg, out_dir, seed=SEED, tag="tts", legacy_ca=False):
    """The tag must tell configurations apart (files are cached by name): use e.g. "tts_tiny_ca" with cfg.cross_attention."""
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{tag}.lm.safetensors")
    if not os.path.exists(path):
        write_safetensors(path, tts_spec(cfg, legacy_ca), "BF16", seed)
    return path


def synth_ca_src(cfg, n, seed):
    """A cross-attention source [n][ca_dim] f32 (the voice's `ca_src` tensor of srv/tts.rs:340-347: speaker-encoder output)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, cfg.ca_dim or cfg.lm.d_model)).astype(np.float32)


def _conv(s, prefix, out_c, in_c, k, bias=True):
    if s.weight_norm:  # un-folded: weight = weight_v * weight_g / ||weight_v||_(1,2)  (core/conv.rs:35-43)
        s.add(f"{prefix}.conv.conv.weight_g", (out_c, 1, 1), "gain", 0.25)
        s.add(f"{prefix}.conv.conv.weight_v", (out_c, in_c, k), "normal", 0.7)
    else:
        s.add(f"{prefix}.conv.conv.weight", (out_c, in_c, k), "normal", (in_c * k) ** -0.5)
    if bias:
        s.add(f"{prefix}.conv.conv.bias", (out_c,), "normal", 0.02)


def _convtr(s, prefix, in_c, out_c, k):
    if s.weight_norm:  # norm over dims (1, 2) of [in_c, out_c, k]: one gain per INPUT channel (core/conv.rs:133-139)
        s.add(f"{prefix}.convtr.convtr.weight_g", (in_c, 1, 1), "gain", 0.25)
        s.add(f"{prefix}.convtr.convtr.weight_v", (in_c, out_c, k), "normal", 0.7)
    else:
        s.add(f"{prefix}.convtr.convtr.weight", (in_c, out_c, k), "normal", (in_c * k) ** -0.5)
    s.add(f"{prefix}.convtr.convtr.bias", (out_c,), "normal", 0.02)


def mimi_spec(m, weight_norm=False):
    """weight_norm: store the SEANet convs / transposed convs as `weight_g` + `weight_v` like the published Mimi
    checkpoint does (the loader folds them: core/conv.rs:27-45,130-141) instead of the pre-folded `weight`."""
    s = Spec()
    s.weight_norm = weight_norm
    nf, ratios = m.n_filters, [m.ratios[i] for i in range(m.n_ratios)]
    # encoder — core/seanet.rs:169-252
    idx, mult = 0, 1
    _conv(s, f"encoder.model.{idx}", mult * nf, m.channels, m.kernel_size)
    idx += 1
    for ratio in reversed(ratios):
        dim = mult * nf
        for _ in range(m.n_residual_layers):
            _conv(s, f"encoder.model.{idx}.block.1", dim // m.compress, dim, m.residual_kernel_size)
            _conv(s, f"encoder.model.{idx}.block.3", dim, dim // m.compress, 1)
            idx += 1
        _conv(s, f"encoder.model.{idx + 1}", dim * 2, dim, 2 * ratio)
        idx += 2
        mult *= 2
    _conv(s, f"encoder.model.{idx + 1}", m.dimension, mult * nf, m.last_kernel_size)
    # decoder — core/seanet.rs:322-408
    idx, mult = 0, 1 << len(ratios)
    _conv(s, f"decoder.model.{idx}", mult * nf, m.dimension, m.kernel_size)
    idx += 1
    for ratio in ratios:
        _convtr(s, f"decoder.model.{idx + 1}", mult * nf, mult * nf // 2, 2 * ratio)
        idx += 2
        dim = mult * nf // 2
        for _ in range(m.n_residual_layers):
            _conv(s, f"decoder.model.{idx}.block.1", dim // m.compress, dim, m.residual_kernel_size)
            _conv(s, f"decoder.model.{idx}.block.3", dim, dim // m.compress, 1)
            idx += 1
        mult //= 2
    _conv(s, f"decoder.model.{idx + 1}", m.channels, nf, m.last_kernel_size)
    t = m.transformer
    _transformer_spec(s, "encoder_transformer.transformer", t, t.dim_feedforward)
    _transformer_spec(s, "decoder_transformer.transformer", t, t.dim_feedforward)
    st = m.downsample_stride
    s.add("downsample.conv.conv.conv.weight", (m.dimension, m.dimension, 2 * st), "normal",
          (m.dimension * 2 * st) ** -0.5)
    s.add("upsample.convtr.convtr.convtr.weight", (m.dimension, 1, 2 * st), "normal", 0.5)
    for name, n in (("rvq_first", 1), ("rvq_rest", m.quantizer_n_q - 1)):
        if n <= 0:
            continue
        p = f"quantizer.{name}"
        s.add(f"{p}.input_proj.weight", (m.quantizer_dim, m.dimension, 1), "normal", m.dimension ** -0.5)
        s.add(f"{p}.output_proj.weight", (m.dimension, m.quantizer_dim, 1), "normal", m.quantizer_dim ** -0.5)
        for i in range(n):
            q = f"{p}.vq.layers.{i}._codebook"
            s.add(f"{q}._initialized", (1,), "const", 1.0)
            s.add(f"{q}.cluster_usage", (m.quantizer_bins,), "const", 1.0)
            s.add(f"{q}.embedding_sum", (m.quantizer_bins, m.quantizer_dim), "normal", 0.07 * (0.85 ** i))
    return s


def make_synth_weights(cfg, out_dir, seed=SEED, tag="model", weight_norm=False):
    """Writes <out_dir>/<tag>.lm.safetensors (BF16) and <tag>.mimi.safetensors (F32); returns the two paths.
    Skips files that already exist (generation is deterministic in (cfg, seed)).  weight_norm: see mimi_spec."""
    os.makedirs(out_dir, exist_ok=True)
    lm_path = os.path.join(out_dir, f"{tag}.lm.safetensors")
    mimi_path = os.path.join(out_dir, f"{tag}.mimi{'_wn' if weight_norm else ''}.safetensors")
    if not os.path.exists(lm_path):
        write_safetensors(lm_path, lm_spec(cfg), "BF16", seed)
    if not os.path.exists(mimi_path):
        write_safetensors(mimi_path, mimi_spec(cfg.mimi, weight_norm), "F32", seed)
    return lm_path, mimi_path


def synth_pcm(batch, steps, seed=1000):
    """Per stream s: 0.1 sin(2 pi (110 + 7 s) t) + 0.02 U(-1, 1) at 24 kHz.
    Returns [steps, B, 1920] f32."""
    n = steps * 1920
    t = np.arange(n, dtype=np.float64) / 24000.0
    out = np.empty((batch, n), dtype=np.float32)
    for s in range(batch):
        r = np.random.Generator(np.random.Philox(key=seed + s))
        out[s] = (0.1 * np.sin(2 * np.pi * (110 + 7 * s) * t) + 0.02 * (2 * r.random(n) - 1)).astype(np.float32)
    return np.ascontiguousarray(out.reshape(batch, steps, 1920).transpose(1, 0, 2))
