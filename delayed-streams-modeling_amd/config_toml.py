"""moshi-server config-toml front end: `[modules.<name>]` tables of type "BatchedAsr" -> dsm_asr_config.

The field names are the reference's serde structs (srv/main.rs:110-128,179-184: ModuleConfig::BatchedAsr / AsrConfig;
core/lm.rs:36-46: lm::Config; core/transformer.rs:21-53: transformer::Config), so an existing
`configs/stt/config-stt-*.toml` drives this engine unchanged.  Keys that belong to the Rust host
(static_dir, log_dir, authorized_ids, path, text_tokenizer_file, ...) are returned untouched in `host`."""
import tomli

from . import AsrConfig, load_library
import ctypes as C

_GATING = {None: 0, "silu": 1}
_NORM = {"LayerNorm": 0, "RmsNorm": 1}
_POS = {"None": 0, "Rope": 1}


class ConfigError(ValueError):
    pass


def _transformer(t, dst):
    for k in ("d_model", "num_heads", "num_layers", "dim_feedforward", "context", "max_period"):
        if k not in t:
            raise ConfigError(f"transformer.{k} is required")
        setattr(dst, k, int(t[k]))
    g = t.get("gating")
    g = g.lower() if isinstance(g, str) else g
    if g not in _GATING:
        raise ConfigError(f"gating {g!r} is not supported (silu or absent)")
    dst.gating = _GATING[g]
    if t.get("norm", "RmsNorm") not in _NORM:
        raise ConfigError(f"norm {t.get('norm')!r} is not supported")
    dst.norm = _NORM[t.get("norm", "RmsNorm")]
    pe = t.get("positional_embedding", "Rope")
    if pe not in _POS:
        raise ConfigError(f"positional_embedding {pe!r} is not supported on the batched path")
    dst.positional_embedding = _POS[pe]
    dst.layer_scale = 1 if t.get("layer_scale") is not None else 0
    dst.conv_layout = 1 if t.get("conv_layout", False) else 0
    # what the batched path would bail on (core/batched_transformer.rs:71-73,284-286,344-346,435-437)
    if t.get("kv_repeat", 1) != 1:
        raise ConfigError("only kv-repeat = 1 is supported")
    if t.get("use_conv_block", False):
        raise ConfigError("conv-block is not supported")
    if not t.get("norm_first", True):
        raise ConfigError("only norm_first = true is supported")
    if not t.get("causal", True):
        raise ConfigError("only causal mode is supported")
    if t.get("bias_ff", False) or t.get("bias_attn", False):
        raise ConfigError("bias_ff / bias_attn are not supported on the accelerated path")
    if t.get("cross_attention") is not None:
        raise ConfigError("cross_attention (TTS speaker conditioning) is outside the STT path")


def load_batched_asr(path, module=None):
    """Returns (AsrConfig, host) for the BatchedAsr module of a moshi-server toml.
    host = {batch_size, lm_model_file, audio_tokenizer_file, text_tokenizer_file, path, instance_name, ...}."""
    with open(path, "rb") as f:
        doc = tomli.load(f)
    mods = doc.get("modules", {})
    cands = {k: v for k, v in mods.items() if v.get("type") == "BatchedAsr"}
    if module is not None:
        cands = {module: cands[module]} if module in cands else {}
    if len(cands) != 1:
        raise ConfigError(f"expected exactly one BatchedAsr module, found {sorted(cands)}")
    name, m = next(iter(cands.items()))
    model = m.get("model")
    if model is None:
        raise ConfigError("modules.%s.model is required" % name)
    cfg = AsrConfig()
    _transformer(model["transformer"], cfg.lm)
    for k in ("text_in_vocab_size", "text_out_vocab_size", "audio_vocab_size", "audio_codebooks"):
        setattr(cfg, k, int(model[k]))
    eh = model.get("extra_heads")
    cfg.extra_heads_num = int(eh["num_heads"]) if eh else 0
    cfg.extra_heads_dim = int(eh["dim"]) if eh else 0
    if model.get("depformer") is not None:
        raise ConfigError("depformer is a TTS/dialogue component, not part of BatchedAsr")
    if model.get("conditioners") is not None:
        raise ConfigError("conditioners are outside the accelerated path (the shipped STT tomls define none)")
    cfg.asr_delay_in_tokens = int(m["asr_delay_in_tokens"])
    cfg.temperature = float(m.get("temperature", 0.0) or 0.0)
    dt = m.get("dtype_override")
    if dt not in (None, "bf16", "f32"):
        raise ConfigError(f"dtype_override {dt!r}: only bf16 and f32 exist on this engine")
    cfg.kv_bf16 = 0 if dt == "f32" else 1
    # extension key of this engine (absent in the reference's tomls): which canonical dot product the bf16-weight GEMMs use
    # (include/dsm.h, dsm_asr_config.dot_mode); 1 — the presets' and bench.py's mode — unless the toml says 0
    dm = m.get("dot_mode", 1)
    if dm not in (0, 1):
        raise ConfigError(f"dot_mode {dm!r}: 0 (f32 fma chain) or 1 (bf16 matrix instruction over the exact three-way split)")
    cfg.dot_mode = int(dm)
    # Mimi: Config::v0_1(Some(audio_codebooks)) — srv/batched_asr.rs:754
    load_library().dsm_mimi_config_v0_1(C.byref(cfg.mimi), cfg.audio_codebooks)
    host = {k: m.get(k) for k in ("path", "lm_model_file", "text_tokenizer_file", "audio_tokenizer_file", "batch_size",
                                  "conditioning_delay", "conditioning_learnt_padding", "log_frequency_s")}
    host["module"] = name
    for k in ("static_dir", "log_dir", "instance_name", "authorized_ids", "warmup"):
        host[k] = doc.get(k)
    return cfg, host
