"""dsm_amd — Python binding of libdsm_mi355x.so (the C ABI in include/dsm.h).

This is plumbing for tests / bench / smoke: ctypes mirrors of the config structs and thin
wrappers that hand numpy (host) or torch (device) buffers to the C entry points.  The compute
lives in the HIP library; there is NO Python or CPU fallback — if the library or the GPU is
missing, calls raise.

The directory name carries a hyphen (it mirrors the upstream repo name), so import it through
the `dsm_amd` shim at the repo root.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdsm_mi355x.so")

FRAME_SIZE = 1920
MAX_RATIOS = 8
MAX_EXTRA_HEADS = 8


class TransformerConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "d_model", "num_heads", "num_layers", "dim_feedforward", "context", "max_period",
        "gating", "norm", "positional_embedding", "layer_scale", "conv_layout")]


class MimiConfig(C.Structure):
    _fields_ = [
        ("channels", C.c_int), ("dimension", C.c_int), ("n_filters", C.c_int),
        ("n_residual_layers", C.c_int), ("n_ratios", C.c_int), ("ratios", C.c_int * MAX_RATIOS),
        ("kernel_size", C.c_int), ("residual_kernel_size", C.c_int), ("last_kernel_size", C.c_int),
        ("dilation_base", C.c_int), ("compress", C.c_int), ("transformer", TransformerConfig),
        ("quantizer_n_q", C.c_int), ("quantizer_bins", C.c_int), ("quantizer_dim", C.c_int),
        ("downsample_stride", C.c_int)]


class AsrConfig(C.Structure):
    _fields_ = [
        ("lm", TransformerConfig), ("text_in_vocab_size", C.c_int), ("text_out_vocab_size", C.c_int),
        ("audio_vocab_size", C.c_int), ("audio_codebooks", C.c_int), ("extra_heads_num", C.c_int),
        ("extra_heads_dim", C.c_int), ("asr_delay_in_tokens", C.c_int), ("temperature", C.c_float),
        ("mimi", MimiConfig), ("kv_bf16", C.c_int), ("dot_mode", C.c_int)]

    def copy(self):
        out = AsrConfig()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(AsrConfig))
        return out


class TtsConfig(C.Structure):
    """dsm_tts_config (include/dsm.h): lm::Config + DepFormerConfig + tts_streaming::Config."""
    _fields_ = [
        ("lm", TransformerConfig), ("text_in_vocab_size", C.c_int), ("text_out_vocab_size", C.c_int),
        ("audio_vocab_size", C.c_int), ("audio_codebooks", C.c_int), ("depformer", TransformerConfig),
        ("dep_num_slices", C.c_int), ("dep_low_rank", C.c_int), ("dep_weight_groups", C.c_int),
        ("acoustic_delay", C.c_int), ("text_pad_token", C.c_int), ("text_bos_token", C.c_int),
        ("text_eos_token", C.c_int), ("text_eop_token", C.c_int), ("text_start_token", C.c_int),
        ("text_audio_delay_in_tokens", C.c_int), ("max_consecutive_pads", C.c_int), ("max_steps", C.c_int),
        ("kv_bf16", C.c_int), ("cross_attention", C.c_int), ("ca_norm", C.c_int), ("ca_dim", C.c_int),
        ("ca_max_len", C.c_int), ("cfg_rows", C.c_int), ("dot_mode", C.c_int)]


TTS_UNGENERATED = 0xFFFFFFFF
TTS_ALLOW_PAD, TTS_ALLOW_PAD_OR_EPAD = -1, -2


class InMsg(C.Structure):
    _fields_ = [("kind", C.c_int), ("id", C.c_int64), ("pcm", C.c_void_p), ("n_pcm", C.c_size_t),
                ("data", C.c_void_p), ("n_data", C.c_size_t)]


class OutMsg(C.Structure):
    _fields_ = [("kind", C.c_int), ("text", C.c_char_p), ("time", C.c_double), ("id", C.c_int64),
                ("step_idx", C.c_uint64), ("prs", C.c_void_p), ("n_prs", C.c_size_t), ("buffered_pcm", C.c_uint64)]


IN_KINDS = ["Init", "Marker", "Audio", "OggOpus", "Ping"]
OUT_KINDS = ["Word", "EndWord", "Marker", "Step", "Error", "Ready"]
DETOK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_void_p, C.c_size_t)


class AsrMsg(C.Structure):
    _fields_ = [("kind", C.c_int), ("batch_idx", C.c_int), ("step_idx", C.c_int), ("time", C.c_double),
                ("tokens_offset", C.c_int), ("n_tokens", C.c_int), ("prs", C.c_float * MAX_EXTRA_HEADS)]


BE_ENCODE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8))
BE_RESET = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
BE_STEP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_float))
BE_POLL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(AsrMsg), C.c_int, C.POINTER(C.c_uint32), C.c_int)
BE_ERROR = C.CFUNCTYPE(C.c_char_p, C.c_void_p)
BE_ENCODE_ASYNC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int))
BE_STEP_TICKET = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_float))


class WorkerBackend(C.Structure):
    """dsm_worker_backend (include/dsm.h)."""
    _fields_ = [("self", C.c_void_p), ("batch_size", C.c_int), ("asr_delay_in_tokens", C.c_int),
                ("extra_heads_num", C.c_int), ("encode_step", BE_ENCODE), ("reset_slot", BE_RESET),
                ("step_tokens", BE_STEP), ("poll_msgs", BE_POLL), ("last_error", BE_ERROR),
                ("encode_async", BE_ENCODE_ASYNC), ("step_ticket", BE_STEP_TICKET)]


class Metrics(C.Structure):
    _fields_ = [("last_encode_us", C.c_double), ("last_lm_us", C.c_double),
                ("algorithmic_bytes_encode", C.c_double), ("algorithmic_bytes_lm", C.c_double),
                ("steps_encode", C.c_uint64), ("steps_lm", C.c_uint64),
                ("graph_launches", C.c_uint64), ("eager_bodies", C.c_uint64),
                ("capture_failures", C.c_uint64), ("capture_error", C.c_char * 96)]


MSG_STEP, MSG_WORD, MSG_END_WORD = 0, 1, 2

# Every symbol include/dsm.h declares (the "not gpu" tests check the library exports all of them).
ABI_SYMBOLS = [
    "dsm_mimi_config_v0_1", "dsm_asr_config_stt_1b_en_fr", "dsm_asr_config_stt_2_6b_en",
    "dsm_asr_create", "dsm_asr_create_from_arena", "dsm_asr_weight_arena", "dsm_destroy", "dsm_last_error", "dsm_mimi_encode_step", "dsm_asr_step_tokens",
    "dsm_asr_step_pcm", "dsm_asr_poll_msgs", "dsm_asr_reset_slot", "dsm_mimi_reset_slot", "dsm_sync",
    "dsm_get_metrics", "dsm_batch_size", "dsm_n_q", "dsm_mimi_encode_step_dev", "dsm_asr_step_tokens_dev",
    "dsm_streams_join", "dsm_debug_read", "dsm_asr_step_pcm_dev", "dsm_prof_enable", "dsm_prof_read",
    "dsm_asr_create_replica", "dsm_asr_set_seed", "dsm_debug_set_positions", "dsm_debug_set_text_tokens", "dsm_mimi_decode_step", "dsm_mimi_decode_step_dev",
    "dsm_lm_stream_groups", "dsm_debug_serialize_groups", "dsm_prof_read_device", "dsm_prof_timeline", "dsm_prof_timeline_read",
    "dsm_wav_decode", "dsm_mp3_decode", "dsm_mp3_decode_info", "dsm_mp3_probe", "dsm_resample", "dsm_pcm_decode",
    "dsm_mp3_test_synth", "dsm_mp3_test_tables", "dsm_mp3_test_imdct", "dsm_free", "dsm_linear_resampler_new", "dsm_linear_resampler_process",
    "dsm_linear_resampler_free", "dsm_ogg_demux_new", "dsm_ogg_demux_free", "dsm_ogg_demux_push", "dsm_ogg_demux_next",
    "dsm_ogg_demux_info", "dsm_worker_set_opus_decoder", "dsm_ogg_mux_new", "dsm_ogg_mux_free", "dsm_ogg_mux_header", "dsm_ogg_mux_page",
    "dsm_inmsg_encode", "dsm_outmsg_encode", "dsm_inmsg_decode", "dsm_outmsg_decode", "dsm_worker_create",
    "dsm_worker_create_with_backend", "dsm_worker_destroy", "dsm_worker_last_error", "dsm_worker_set_detokenizer", "dsm_worker_open", "dsm_worker_close",
    "dsm_worker_send", "dsm_worker_send_body", "dsm_worker_step", "dsm_worker_step_encode", "dsm_worker_step_model",
    "dsm_mimi_encode_step_async", "dsm_asr_step_tokens_ticket", "dsm_worker_recv", "dsm_worker_buffered",
    "dsm_tts_config_v202501", "dsm_tts_create", "dsm_tts_destroy", "dsm_tts_last_error", "dsm_tts_step",
    "dsm_tts_audio_tokens", "dsm_tts_step_idx", "dsm_tts_reset_slot", "dsm_tts_debug_read", "dsm_tts_get_metrics", "dsm_tts_set_sampling",
    "dsm_tts_set_ca_src",
]
PROF_TAGS = ["attn_lm", "gemm_lm", "attn_mimi", "gemm_mimi", "rvq", "other"]

_lib = None


def load_library(path=None):
    """dlopen the HIP library (no GPU needed for this) and declare signatures."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback for the MI355X engine.")
    lib = C.CDLL(p)
    vp, ip, u8p, u32p, fp = C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p
    lib.dsm_mimi_config_v0_1.argtypes = [C.POINTER(MimiConfig), C.c_int]
    lib.dsm_asr_config_stt_1b_en_fr.argtypes = [C.POINTER(AsrConfig)]
    lib.dsm_asr_config_stt_2_6b_en.argtypes = [C.POINTER(AsrConfig)]
    lib.dsm_asr_create.argtypes = [C.POINTER(AsrConfig), C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(vp)]
    lib.dsm_asr_create.restype = C.c_int
    lib.dsm_asr_create_from_arena.argtypes = [C.POINTER(AsrConfig), C.c_int, C.c_int, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(vp)]
    lib.dsm_asr_weight_arena.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.dsm_destroy.argtypes = [vp]
    lib.dsm_destroy.restype = None
    lib.dsm_last_error.argtypes = [vp]
    lib.dsm_last_error.restype = C.c_char_p
    lib.dsm_mimi_encode_step.argtypes = [vp, fp, u8p, u32p, C.POINTER(C.c_int)]
    lib.dsm_asr_step_tokens.argtypes = [vp, u32p, u8p, u32p, fp]
    lib.dsm_asr_step_pcm.argtypes = [vp, fp, u8p, u32p, u32p, fp]
    lib.dsm_asr_poll_msgs.argtypes = [vp, C.POINTER(AsrMsg), C.c_int, u32p, C.c_int]
    lib.dsm_asr_reset_slot.argtypes = [vp, C.c_int]
    lib.dsm_mimi_reset_slot.argtypes = [vp, C.c_int]
    lib.dsm_sync.argtypes = [vp]
    lib.dsm_get_metrics.argtypes = [vp, C.POINTER(Metrics)]
    lib.dsm_batch_size.argtypes = [vp]
    lib.dsm_n_q.argtypes = [vp]
    lib.dsm_mimi_encode_step_dev.argtypes = [vp, vp, vp, vp]
    lib.dsm_asr_step_tokens_dev.argtypes = [vp, vp, vp, vp, vp]
    lib.dsm_streams_join.argtypes = [vp]
    lib.dsm_debug_read.argtypes = [vp, C.c_char_p, fp, C.c_size_t]
    lib.dsm_asr_step_pcm_dev.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.dsm_mimi_decode_step.argtypes = [vp, u32p, u8p, fp, C.POINTER(C.c_int)]
    lib.dsm_mimi_decode_step.restype = C.c_int
    lib.dsm_mimi_decode_step_dev.argtypes = [vp, vp, vp, vp]
    lib.dsm_mimi_decode_step_dev.restype = C.c_int
    lib.dsm_debug_set_positions.argtypes = [vp, C.c_uint32, C.c_uint32]
    lib.dsm_debug_set_positions.restype = C.c_int
    lib.dsm_asr_create_replica.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.dsm_asr_create_replica.restype = C.c_int
    lib.dsm_asr_set_seed.argtypes = [vp, C.c_int, C.c_uint64]
    lib.dsm_asr_set_seed.restype = C.c_int
    lib.dsm_debug_set_text_tokens.argtypes = [vp, vp]
    lib.dsm_debug_set_text_tokens.restype = C.c_int
    lib.dsm_prof_enable.argtypes = [vp, C.c_uint]
    lib.dsm_prof_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    for name in ("dsm_mimi_encode_step", "dsm_asr_step_tokens", "dsm_asr_step_pcm", "dsm_asr_poll_msgs",
                 "dsm_asr_reset_slot", "dsm_mimi_reset_slot", "dsm_sync", "dsm_get_metrics", "dsm_batch_size",
                 "dsm_n_q", "dsm_mimi_encode_step_dev", "dsm_asr_step_tokens_dev", "dsm_streams_join",
                 "dsm_debug_read", "dsm_asr_step_pcm_dev", "dsm_prof_enable", "dsm_prof_read"):
        getattr(lib, name).restype = C.c_int
    lib.dsm_prof_read_device.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.dsm_prof_read_device.restype = C.c_int
    lib.dsm_lm_stream_groups.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    lib.dsm_lm_stream_groups.restype = C.c_int
    lib.dsm_debug_serialize_groups.argtypes = [vp, C.c_int]
    lib.dsm_debug_serialize_groups.restype = C.c_int
    lib.dsm_wav_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_int)]
    lib.dsm_wav_decode.restype = C.c_int
    for name in ("dsm_mp3_decode", "dsm_pcm_decode"):
        getattr(lib, name).argtypes = lib.dsm_wav_decode.argtypes
        getattr(lib, name).restype = C.c_int
    lib.dsm_mp3_decode_info.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t), C.POINTER(Mp3Info)]
    lib.dsm_mp3_decode_info.restype = C.c_int
    lib.dsm_mp3_probe.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Mp3Info)]
    lib.dsm_mp3_probe.restype = C.c_int
    lib.dsm_resample.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_size_t)]
    lib.dsm_resample.restype = C.c_int
    lib.dsm_mp3_test_synth.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.dsm_mp3_test_tables.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_int]
    lib.dsm_mp3_test_imdct.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.dsm_free.argtypes = [vp]
    lib.dsm_free.restype = None
    lib.dsm_linear_resampler_new.argtypes = [C.c_uint32, C.c_uint32]
    lib.dsm_linear_resampler_new.restype = vp
    lib.dsm_linear_resampler_process.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t]
    lib.dsm_linear_resampler_process.restype = C.c_size_t
    lib.dsm_linear_resampler_free.argtypes = [vp]
    lib.dsm_linear_resampler_free.restype = None
    lib.dsm_inmsg_encode.argtypes = [C.POINTER(InMsg), vp, C.c_size_t]
    lib.dsm_outmsg_encode.argtypes = [C.POINTER(OutMsg), vp, C.c_size_t]
    lib.dsm_inmsg_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(InMsg), vp, C.c_size_t, vp, C.c_size_t]
    lib.dsm_outmsg_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(OutMsg), vp, C.c_size_t, vp, C.c_size_t]
    lib.dsm_worker_create.argtypes = [vp, C.POINTER(vp)]
    lib.dsm_worker_create_with_backend.argtypes = [C.POINTER(WorkerBackend), C.POINTER(vp)]
    lib.dsm_worker_create_with_backend.restype = C.c_int
    lib.dsm_worker_destroy.argtypes = [vp]
    lib.dsm_worker_destroy.restype = None
    lib.dsm_worker_last_error.argtypes = [vp]
    lib.dsm_worker_last_error.restype = C.c_char_p
    lib.dsm_worker_set_detokenizer.argtypes = [vp, DETOK_FN, vp]
    lib.dsm_worker_set_detokenizer.restype = None
    lib.dsm_worker_open.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.dsm_worker_close.argtypes = [vp, C.c_int]
    lib.dsm_worker_send.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t]
    lib.dsm_worker_step.argtypes = [vp]
    lib.dsm_worker_step_encode.argtypes = [vp]
    lib.dsm_worker_step_model.argtypes = [vp]
    lib.dsm_worker_recv.argtypes = [vp, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.dsm_worker_buffered.argtypes = [vp, C.c_int]
    for name in ("dsm_inmsg_encode", "dsm_outmsg_encode", "dsm_inmsg_decode", "dsm_outmsg_decode", "dsm_worker_create",
                 "dsm_worker_open", "dsm_worker_close", "dsm_worker_send", "dsm_worker_step", "dsm_worker_step_encode", "dsm_worker_step_model",
    "dsm_mimi_encode_step_async", "dsm_asr_step_tokens_ticket", "dsm_worker_recv",
                 "dsm_worker_buffered"):
        getattr(lib, name).restype = C.c_int
    lib.dsm_tts_config_v202501.argtypes = [C.POINTER(TtsConfig)]
    lib.dsm_tts_config_v202501.restype = None
    lib.dsm_tts_create.argtypes = [C.POINTER(TtsConfig), C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]
    lib.dsm_tts_destroy.argtypes = [vp]
    lib.dsm_tts_destroy.restype = None
    lib.dsm_tts_last_error.argtypes = [vp]
    lib.dsm_tts_last_error.restype = C.c_char_p
    lib.dsm_tts_step.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.dsm_tts_audio_tokens.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.dsm_tts_step_idx.argtypes = [vp, C.c_int]
    lib.dsm_tts_reset_slot.argtypes = [vp, C.c_int]
    lib.dsm_tts_debug_read.argtypes = [vp, C.c_char_p, fp, C.c_size_t]
    lib.dsm_tts_get_metrics.argtypes = [vp, C.POINTER(Metrics)]
    lib.dsm_tts_set_sampling.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_uint64]
    lib.dsm_tts_set_ca_src.argtypes = [vp, C.c_int, fp, C.c_int, fp, C.c_int, C.c_double]
    for name in ("dsm_tts_create", "dsm_tts_step", "dsm_tts_audio_tokens", "dsm_tts_step_idx", "dsm_tts_reset_slot",
                 "dsm_tts_debug_read"):
        getattr(lib, name).restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def config_stt_1b_en_fr():
    cfg = AsrConfig()
    load_library().dsm_asr_config_stt_1b_en_fr(C.byref(cfg))
    return cfg


def config_stt_2_6b_en():
    cfg = AsrConfig()
    load_library().dsm_asr_config_stt_2_6b_en(C.byref(cfg))
    return cfg


def config_tiny(kv_bf16=1):
    """A small configuration with every structural feature of stt-1b-en_fr (same ratios, T shapes,
    ring-cache wrap within a few steps) that the CPU oracle steps in milliseconds."""
    cfg = AsrConfig()
    t = cfg.lm
    t.d_model, t.num_heads, t.num_layers, t.dim_feedforward = 128, 4, 2, 512
    t.context, t.max_period, t.gating, t.norm = 12, 100000, 1, 1
    t.positional_embedding, t.layer_scale, t.conv_layout = 1, 0, 0
    cfg.text_in_vocab_size, cfg.text_out_vocab_size = 65, 64
    cfg.audio_vocab_size, cfg.audio_codebooks = 33, 4
    cfg.extra_heads_num, cfg.extra_heads_dim = 2, 6
    cfg.asr_delay_in_tokens, cfg.temperature, cfg.kv_bf16 = 2, 0.0, kv_bf16
    m = cfg.mimi
    m.channels, m.dimension, m.n_filters, m.n_residual_layers = 1, 64, 4, 1
    m.n_ratios = 4
    for i, r in enumerate((8, 6, 5, 4)):
        m.ratios[i] = r
    m.kernel_size, m.residual_kernel_size, m.last_kernel_size = 7, 3, 3
    m.dilation_base, m.compress = 2, 2
    mt = m.transformer
    mt.d_model, mt.num_heads, mt.num_layers, mt.dim_feedforward = 64, 2, 2, 128
    mt.context, mt.max_period, mt.gating, mt.norm = 10, 10000, 0, 0
    mt.positional_embedding, mt.layer_scale, mt.conv_layout = 1, 1, 1
    m.quantizer_n_q, m.quantizer_bins, m.quantizer_dim, m.downsample_stride = 4, 32, 16, 2
    return cfg


def config_medium(lm_heads=4, lm_head_dim=128, lm_context=300, kv_bf16=1, lm_layers=2, mimi_head_dim=64, mimi_context=250):
    """Between tiny and real: the LM and Mimi transformers get the REAL head dims (128 / 64) and ring lengths of a few
    hundred frames, so that the attention kernel's software-pipelined loop runs several iterations per phase (both
    register sets, the tail clamp after the first iteration, ring wrap) and d_model spans more than one 256-wide
    K-chunk (split-K slabs, the fused QKV prologue) — while SEANet, the RVQ and the vocabularies stay small enough for
    the CPU oracle to step a batch in ~10 ms."""
    cfg = config_tiny(kv_bf16)
    t = cfg.lm
    t.d_model, t.num_heads, t.num_layers = lm_heads * lm_head_dim, lm_heads, lm_layers
    t.dim_feedforward, t.context = 4 * t.d_model, lm_context  # gating hidden = 11 d / 4 (a multiple of 32 for d % 128 == 0)
    cfg.text_in_vocab_size, cfg.text_out_vocab_size = 257, 256
    cfg.audio_vocab_size, cfg.audio_codebooks = 65, 8
    cfg.asr_delay_in_tokens = 3
    m = cfg.mimi
    m.dimension, m.n_filters = 2 * mimi_head_dim, 4
    mt = m.transformer
    mt.d_model, mt.num_heads, mt.num_layers, mt.dim_feedforward, mt.context = 2 * mimi_head_dim, 2, 2, 256, mimi_context
    m.quantizer_n_q, m.quantizer_bins, m.quantizer_dim = 8, 64, 32
    return cfg


def config_tts_v202501():
    cfg = TtsConfig()
    load_library().dsm_tts_config_v202501(C.byref(cfg))
    return cfg


def config_tts_tiny(kv_bf16=1, cross_attention=False, cfg_rows=False, ca_dim=0, ca_norm=0):
    """Small TTS configuration with every structural feature of the v202501 model: shared depformer with
    fewer weight groups than slices, low-rank embeddings, acoustic and text-audio delays.  cross_attention / cfg_rows:
    the branch the reference server runs (norm_cross + cross-attention in every main-LM layer, two batch rows per slot)."""
    cfg = TtsConfig()
    t = cfg.lm
    t.d_model, t.num_heads, t.num_layers, t.dim_feedforward = 128, 4, 2, 512
    t.context, t.max_period, t.gating, t.norm = 16, 10000, 1, 1
    t.positional_embedding, t.layer_scale, t.conv_layout = 1, 0, 0
    cfg.text_in_vocab_size, cfg.text_out_vocab_size = 41, 40
    cfg.audio_vocab_size, cfg.audio_codebooks = 33, 6
    dp = cfg.depformer
    dp.d_model, dp.num_heads, dp.num_layers, dp.dim_feedforward = 64, 2, 2, 192
    dp.context, dp.max_period, dp.gating, dp.norm = 6, 10000, 1, 1
    dp.positional_embedding, dp.layer_scale, dp.conv_layout = 0, 0, 0
    cfg.dep_num_slices, cfg.dep_low_rank, cfg.dep_weight_groups = 6, 32, 3
    cfg.acoustic_delay, cfg.text_pad_token, cfg.text_bos_token = 2, 3, 1
    cfg.text_eos_token, cfg.text_eop_token, cfg.text_start_token = 2, 0, 40
    cfg.text_audio_delay_in_tokens, cfg.max_consecutive_pads, cfg.max_steps = 3, 4, 64
    cfg.kv_bf16 = kv_bf16
    if cross_attention:
        cfg.cross_attention, cfg.ca_norm, cfg.ca_dim, cfg.ca_max_len = 1, ca_norm, ca_dim, 24
        cfg.cfg_rows = 1 if cfg_rows else 0
    return cfg


def _ptr(a):
    """Host pointer of a C-contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class DsmError(RuntimeError):
    pass


def encode_in_msg(kind, id=0, pcm=None, data=None):
    """InMsg -> msgpack bytes (client/rust/kyutai-client/src/stt/protocol.rs:47-58)."""
    lib = load_library()
    m = InMsg()
    m.kind, m.id = IN_KINDS.index(kind), id
    keep = []
    if pcm is not None:
        a = np.ascontiguousarray(pcm, dtype=np.float32)
        keep.append(a)
        m.pcm, m.n_pcm = a.ctypes.data, a.size
    if data is not None:
        b = np.frombuffer(bytes(data), dtype=np.uint8).copy()
        keep.append(b)
        m.data, m.n_data = b.ctypes.data, b.size
    n = lib.dsm_inmsg_encode(C.byref(m), None, 0)
    buf = np.zeros(max(n, 1), dtype=np.uint8)
    assert lib.dsm_inmsg_encode(C.byref(m), _ptr(buf), buf.size) == n
    return buf[:n].tobytes()


def decode_in_msg(raw):
    """msgpack bytes -> dict, as the server's rmp_serde::from_slice::<InMsg> (srv/batched_asr.rs:927); None if rejected."""
    lib = load_library()
    m = InMsg()
    pcm, data = np.zeros(len(raw) + 1, dtype=np.float32), np.zeros(len(raw) + 1, dtype=np.uint8)
    if lib.dsm_inmsg_decode(raw, len(raw), C.byref(m), _ptr(pcm), pcm.size, _ptr(data), data.size) != 0:
        return None
    out = {"type": IN_KINDS[m.kind]}
    if out["type"] == "Marker":
        out["id"] = m.id
    elif out["type"] == "Audio":
        out["pcm"] = pcm[:m.n_pcm].copy()
    elif out["type"] == "OggOpus":
        out["data"] = data[:m.n_data].tobytes()
    return out


def encode_out_msg(kind, text="", time=0.0, id=0, step_idx=0, prs=(), buffered_pcm=0):
    """OutMsg -> msgpack bytes, as send_loop serialises it (srv/batched_asr.rs:969-975)."""
    lib = load_library()
    m = OutMsg()
    p = np.ascontiguousarray(prs, dtype=np.float32)
    m.kind, m.text, m.time, m.id, m.step_idx = OUT_KINDS.index(kind), text.encode(), time, id, step_idx
    m.prs, m.n_prs, m.buffered_pcm = p.ctypes.data, p.size, buffered_pcm
    n = lib.dsm_outmsg_encode(C.byref(m), None, 0)
    buf = np.zeros(max(n, 1), dtype=np.uint8)
    assert lib.dsm_outmsg_encode(C.byref(m), _ptr(buf), buf.size) == n
    return buf[:n].tobytes()


def decode_out_msg(raw):
    """msgpack bytes -> dict (client: protocol.rs:60-62 decode_out_msg); None if rejected."""
    lib = load_library()
    m = OutMsg()
    text, prs = C.create_string_buffer(len(raw) + 1), np.zeros(len(raw) + 1, dtype=np.float32)
    if lib.dsm_outmsg_decode(raw, len(raw), C.byref(m), text, len(raw) + 1, _ptr(prs), prs.size) != 0:
        return None
    k = OUT_KINDS[m.kind]
    out = {"type": k}
    if k == "Word":
        out.update(text=text.value.decode(), start_time=m.time)
    elif k == "EndWord":
        out.update(stop_time=m.time)
    elif k == "Marker":
        out.update(id=m.id)
    elif k == "Step":
        out.update(step_idx=m.step_idx, prs=prs[:m.n_prs].tolist(), buffered_pcm=m.buffered_pcm)
    elif k == "Error":
        out.update(message=text.value.decode())
    return out


class Worker:
    """moshi-server's BatchedAsr module above an AsrEngine, minus the sockets: slot table, per-channel PCM queue,
    markers, message fan-out (srv/batched_asr.rs).  Messages cross this interface as the msgpack bytes of the wire."""

    def __init__(self, engine=None, detokenizer=None, backend=None):
        """engine: an AsrEngine (the product path).  backend: a filled WorkerBackend instead (the tests drive the
        worker logic with the CPU oracle through it); the caller keeps its callbacks alive."""
        self.lib, self.engine, self.backend = load_library(), engine, backend
        h = C.c_void_p()
        if backend is not None:
            rc = self.lib.dsm_worker_create_with_backend(C.byref(backend), C.byref(h))
        else:
            rc = self.lib.dsm_worker_create(engine.h, C.byref(h))
        if rc != 0:
            raise DsmError("dsm_worker_create failed")
        self.h = h
        self._cb = None
        if detokenizer is not None:
            def cb(_user, toks, n, out, cap):
                s = detokenizer([toks[i] for i in range(n)]).encode()
                if len(s) > cap:
                    return -1
                C.memmove(out, s, len(s))
                return len(s)
            self._cb = DETOK_FN(cb)
            self.lib.dsm_worker_set_detokenizer(self.h, self._cb, None)

    def _err(self):
        return self.lib.dsm_worker_last_error(self.h).decode()

    def open(self):
        cid = C.c_uint64(0)
        slot = self.lib.dsm_worker_open(self.h, C.byref(cid))
        if slot < 0:
            raise DsmError(self._err())
        return slot

    def close_channel(self, slot):
        self.lib.dsm_worker_close(self.h, slot)

    def send(self, slot, raw):
        rc = self.lib.dsm_worker_send(self.h, slot, raw, len(raw))
        if rc < 0:
            raise DsmError(self._err())
        return rc == 0

    def send_body(self, slot, body):
        """BatchedAsr::handle_query: a whole audio file as the request (Init, the clip at 24 kHz, Marker 0, 10 s of silence)."""
        rc = self.lib.dsm_worker_send_body(self.h, slot, body, len(body))
        if rc < 0:
            raise DsmError(self._err())

    def step(self):
        rc = self.lib.dsm_worker_step(self.h)
        if rc < 0:
            raise DsmError(self._err())
        return rc == 1

    def step_encode(self):
        """encoder_loop iteration (dsm_worker_step_encode): True if a frame was cut and queued for the model side."""
        rc = self.lib.dsm_worker_step_encode(self.h)
        if rc < 0:
            raise DsmError(self._err())
        return rc == 1

    def step_model(self):
        """model_loop + post_process for the oldest queued frame (dsm_worker_step_model)."""
        rc = self.lib.dsm_worker_step_model(self.h)
        if rc < 0:
            raise DsmError(self._err())
        return rc == 1

    def recv(self, slot):
        """All pending OutMsg of the slot, decoded."""
        out, buf, n = [], np.zeros(1 << 16, dtype=np.uint8), C.c_size_t(0)
        while self.lib.dsm_worker_recv(self.h, slot, _ptr(buf), buf.size, C.byref(n)) == 1:
            out.append(decode_out_msg(buf[:n.value].tobytes()))
        return out

    def buffered(self, slot):
        return self.lib.dsm_worker_buffered(self.h, slot)

    def close(self):
        if getattr(self, "h", None):
            self.lib.dsm_worker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mp3Info(C.Structure):
    _fields_ = [("sample_rate", C.c_int), ("channels", C.c_int), ("bitrate_kbps", C.c_int), ("vbr", C.c_int), ("frames", C.c_int),
                ("info_frames", C.c_int), ("resyncs", C.c_int), ("huffman_overruns", C.c_int), ("frames_without_reservoir", C.c_int),
                ("id3v2_bytes", C.c_uint64), ("junk_bytes", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def _take_pcm(lib, ptr, n):
    try:
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32)
    finally:
        lib.dsm_free(ptr)


def mp3_decode(data):
    """(pcm f32 of channel 0, sample_rate, info dict) of an MPEG-1 Layer III body — srv/utils.rs:263-305 pcm_decode (symphonia)."""
    lib = load_library()
    ptr, n, info = C.POINTER(C.c_float)(), C.c_size_t(0), Mp3Info()
    rc = lib.dsm_mp3_decode_info(data, len(data), C.byref(ptr), C.byref(n), C.byref(info))
    if rc != 0:
        msg = lib.dsm_last_error(None)
        raise DsmError(f"dsm_mp3_decode failed ({rc}): {msg.decode() if msg else '?'}")
    return _take_pcm(lib, ptr, n), info.sample_rate, info.as_dict()


def mp3_probe(data):
    lib = load_library()
    info = Mp3Info()
    rc = lib.dsm_mp3_probe(data, len(data), C.byref(info))
    if rc != 0:
        msg = lib.dsm_last_error(None)
        raise DsmError(f"dsm_mp3_probe failed ({rc}): {msg.decode() if msg else '?'}")
    return info.as_dict()


def pcm_decode(data):
    """RIFF/WAVE or mp3 by magic: (pcm of channel 0, sample_rate) — what the worker front end does with a request body."""
    lib = load_library()
    ptr, n, rate = C.POINTER(C.c_float)(), C.c_size_t(0), C.c_int(0)
    rc = lib.dsm_pcm_decode(data, len(data), C.byref(ptr), C.byref(n), C.byref(rate))
    if rc != 0:
        msg = lib.dsm_last_error(None)
        raise DsmError(f"dsm_pcm_decode failed ({rc}): {msg.decode() if msg else '?'}")
    return _take_pcm(lib, ptr, n), rate.value


def resample(pcm, rate_in, rate_out):
    """Whole-buffer windowed-sinc resampler in the place of kaudio::resample (srv/batched_asr.rs:838)."""
    lib = load_library()
    x = np.ascontiguousarray(pcm, dtype=np.float32)
    ptr, n = C.POINTER(C.c_float)(), C.c_size_t(0)
    rc = lib.dsm_resample(x.ctypes.data, x.size, int(rate_in), int(rate_out), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise DsmError(f"dsm_resample failed ({rc})")
    return _take_pcm(lib, ptr, n)


def wav_decode(data):
    """(pcm f32 of channel 0, sample_rate) of a RIFF/WAVE body — srv/utils.rs:263-305 pcm_decode."""
    lib = load_library()
    ptr, n, rate = C.POINTER(C.c_float)(), C.c_size_t(0), C.c_int(0)
    rc = lib.dsm_wav_decode(data, len(data), C.byref(ptr), C.byref(n), C.byref(rate))
    if rc != 0:
        msg = lib.dsm_last_error(None)
        raise DsmError(f"dsm_wav_decode failed ({rc}): {msg.decode() if msg else '?'}")
    try:
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32), rate.value
    finally:
        lib.dsm_free(ptr)


class LinearResampler:
    """kyutai-client-core `LinearResampler` (audio.rs:133-183) behind the C ABI; process() is streaming."""

    def __init__(self, in_rate_hz, out_rate_hz):
        self.lib = load_library()
        self.h = self.lib.dsm_linear_resampler_new(in_rate_hz, out_rate_hz)
        if not self.h:
            raise DsmError("bad sample rates")
        self.ratio = out_rate_hz / in_rate_hz

    def process(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        out = np.zeros(int(pcm.size * self.ratio) + 4, dtype=np.float32)
        n = self.lib.dsm_linear_resampler_process(self.h, _ptr(pcm), pcm.size, _ptr(out), out.size)
        assert n <= out.size
        return out[:n]

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dsm_linear_resampler_free(self.h)
            self.h = None


class AsrEngine:
    """Mirror of the reference's (Mimi::batched, LmModel::batched, asr::State) triple behind the C ABI.

    Method names follow the reference: encode_step (core/mimi.rs:195), step_tokens (core/asr.rs:147),
    step_pcm (core/asr.rs:115), reset_batch_idx (core/asr.rs:257)."""

    def __init__(self, cfg, batch_size, lm_path=None, mimi_path=None, device_id=0, arena=None):
        """From the two safetensors files, or — `arena=(device_ptr, nbytes, manifest_bytes[, keepalive])` — attached to a
        packed weight arena another rank broadcast (dsm_asr_create_from_arena); the caller's buffer must outlive the engine
        (pass it as the optional 4th element and it is kept referenced here)."""
        self.lib = load_library()
        self.cfg = cfg
        self.B = batch_size
        h = C.c_void_p()
        if arena is not None:
            ptr, nbytes, manifest = arena[0], arena[1], bytes(arena[2])
            self._arena_keepalive = arena[3] if len(arena) > 3 else None
            mbuf = (C.c_uint8 * max(len(manifest), 1)).from_buffer_copy(manifest or b"\0")
            rc = self.lib.dsm_asr_create_from_arena(C.byref(cfg), device_id, batch_size, C.c_void_p(ptr), nbytes, mbuf,
                                                    len(manifest), C.byref(h))
        else:
            rc = self.lib.dsm_asr_create(C.byref(cfg), device_id, batch_size, lm_path.encode(), mimi_path.encode(),
                                         C.byref(h))
        if rc != 0:
            msg = self.lib.dsm_last_error(None)
            raise DsmError(f"dsm_asr_create failed ({rc}): {msg.decode() if msg else '?'}")
        self.h = h
        self.n_q = self.lib.dsm_n_q(h)

    @classmethod
    def replica(cls, src, device_id, batch_size=None, cfg=None):
        """A second engine in THIS process on `device_id` whose weights are a device-to-device copy of `src`'s arena
        (dsm_asr_create_replica: hipMemcpyPeerAsync over xGMI; a device copy when device_id is src's own device)."""
        self = cls.__new__(cls)
        self.lib = src.lib
        self.cfg = cfg if cfg is not None else src.cfg
        self.B = batch_size if batch_size is not None else src.B
        h = C.c_void_p()
        rc = self.lib.dsm_asr_create_replica(src.h, C.byref(self.cfg) if cfg is not None else None, device_id, self.B, C.byref(h))
        if rc != 0:
            msg = self.lib.dsm_last_error(None)
            raise DsmError(f"dsm_asr_create_replica failed ({rc}): {msg.decode() if msg else '?'}")
        self.h = h
        self.n_q = self.lib.dsm_n_q(h)
        return self

    def weight_arena(self):
        """(device pointer, nbytes, manifest bytes) of this engine's packed weight arena (dsm_asr_weight_arena)."""
        ptr, n, mp, mn = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        self._check(self.lib.dsm_asr_weight_arena(self.h, C.byref(ptr), C.byref(n), C.byref(mp), C.byref(mn)))
        manifest = C.string_at(mp, mn.value) if mn.value else b""
        return ptr.value, n.value, manifest

    def _check(self, rc):
        if rc < 0:
            msg = self.lib.dsm_last_error(self.h)
            raise DsmError(f"dsm error {rc}: {msg.decode() if msg else '?'}")
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.lib.dsm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_step(self, pcm, mask):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32).reshape(self.B, FRAME_SIZE)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        codes = np.zeros((self.B, self.n_q), dtype=np.uint32)
        produced = C.c_int(0)
        self._check(self.lib.dsm_mimi_encode_step(self.h, _ptr(pcm), _ptr(mask), _ptr(codes), C.byref(produced)))
        return codes if produced.value else None

    def decode_step(self, codes, mask):
        codes = np.ascontiguousarray(codes, dtype=np.uint32).reshape(self.B, self.n_q)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        pcm = np.zeros((self.B, FRAME_SIZE), dtype=np.float32)
        produced = C.c_int(0)
        self._check(self.lib.dsm_mimi_decode_step(self.h, _ptr(codes), _ptr(mask), _ptr(pcm), C.byref(produced)))
        return pcm if produced.value else None

    def decode_step_dev(self, d_codes, d_mask, d_pcm):
        self._check(self.lib.dsm_mimi_decode_step_dev(self.h, d_codes, d_mask, d_pcm))

    def step_tokens(self, codes, mask):
        """codes None: use the device-resident codes of the last encode_step (srv/batched_asr.rs hands them over on-device too)."""
        if codes is not None:
            codes = np.ascontiguousarray(codes, dtype=np.uint32).reshape(self.B, self.n_q)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        text = np.zeros(self.B, dtype=np.uint32)
        nh = self.cfg.extra_heads_num
        prs = np.zeros((max(nh, 1), self.B), dtype=np.float32)
        self._check(self.lib.dsm_asr_step_tokens(self.h, _ptr(codes), _ptr(mask), _ptr(text), _ptr(prs) if nh else None))
        return text, prs[:nh]

    def step_pcm(self, pcm, mask):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32).reshape(self.B, FRAME_SIZE)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        codes = np.zeros((self.B, self.n_q), dtype=np.uint32)
        text = np.zeros(self.B, dtype=np.uint32)
        nh = self.cfg.extra_heads_num
        prs = np.zeros((max(nh, 1), self.B), dtype=np.float32)
        self._check(self.lib.dsm_asr_step_pcm(self.h, _ptr(pcm), _ptr(mask), _ptr(codes), _ptr(text),
                                              _ptr(prs) if nh else None))
        return codes, text, prs[:nh]

    def poll_msgs(self, cap=4096):
        msgs = (AsrMsg * cap)()
        toks = np.zeros(cap * 8, dtype=np.uint32)
        n = self._check(self.lib.dsm_asr_poll_msgs(self.h, msgs, cap, _ptr(toks), toks.size))
        out = []
        for i in range(n):
            m = msgs[i]
            if m.kind == MSG_WORD:
                out.append(("Word", m.batch_idx, m.time, toks[m.tokens_offset:m.tokens_offset + m.n_tokens].tolist()))
            elif m.kind == MSG_END_WORD:
                out.append(("EndWord", m.batch_idx, m.time))
            else:
                out.append(("Step", m.step_idx))
        return out

    def reset_batch_idx(self, slot):
        self._check(self.lib.dsm_asr_reset_slot(self.h, slot))

    def mimi_reset_batch_idx(self, slot):
        self._check(self.lib.dsm_mimi_reset_slot(self.h, slot))

    def sync(self):
        self._check(self.lib.dsm_sync(self.h))

    def metrics(self):
        m = Metrics()
        self._check(self.lib.dsm_get_metrics(self.h, C.byref(m)))
        return m

    def debug_read(self, name, n):
        out = np.zeros(n, dtype=np.float32)
        got = self._check(self.lib.dsm_debug_read(self.h, name.encode(), _ptr(out), n))
        return out[:got]

    # device-pointer variants (ints from torch.Tensor.data_ptr())
    def encode_step_dev(self, d_pcm, d_mask, d_codes):
        self._check(self.lib.dsm_mimi_encode_step_dev(self.h, d_pcm, d_mask, d_codes))

    def step_tokens_dev(self, d_codes, d_mask, d_text, d_prs):
        self._check(self.lib.dsm_asr_step_tokens_dev(self.h, d_codes, d_mask, d_text, d_prs))

    def streams_join(self):
        self._check(self.lib.dsm_streams_join(self.h))

    def step_pcm_dev(self, d_pcm, d_mask, d_codes=None, d_text=None, d_prs=None):
        self._check(self.lib.dsm_asr_step_pcm_dev(self.h, d_pcm, d_mask, d_codes, d_text, d_prs))

    def debug_set_positions(self, lm_pos, mimi_pos):
        self._check(self.lib.dsm_debug_set_positions(self.h, lm_pos, mimi_pos))

    def set_seed(self, slot, seed):
        """temperature > 0: restart the slot's Gumbel-noise stream (ChaCha12 keyed by seed_from_u64(seed))."""
        self._check(self.lib.dsm_asr_set_seed(self.h, slot, seed))

    def debug_set_text_tokens(self, tokens):
        """Teacher forcing: the text token every slot feeds back into its next step."""
        t = np.ascontiguousarray(tokens, dtype=np.uint32)
        assert t.shape == (self.B,)
        self._check(self.lib.dsm_debug_set_text_tokens(self.h, t.ctypes.data))

    def prof_read_device(self):
        """Like prof_read, from the in-kernel device-clock brackets (attention classes only)."""
        tot = (C.c_double * len(PROF_TAGS))()
        cnt = (C.c_uint64 * len(PROF_TAGS))()
        self._check(self.lib.dsm_prof_read_device(self.h, tot, cnt))
        return {t: (tot[i], cnt[i]) for i, t in enumerate(PROF_TAGS)}

    def prof_timeline(self, on):
        self._check(self.lib.dsm_prof_timeline(self.h, 1 if on else 0))

    def prof_timeline_read(self, cap=1 << 16):
        """[(stream id, kind, tag name, start_us, end_us), ...] of every bracketed launch since the last read."""
        sid, kind, tag = (C.c_int * cap)(), (C.c_int * cap)(), (C.c_int * cap)()
        t0, t1 = (C.c_double * cap)(), (C.c_double * cap)()
        n = self._check(self.lib.dsm_prof_timeline_read(self.h, sid, kind, tag, t0, t1, cap))
        return [(sid[i], kind[i], PROF_TAGS[tag[i]], t0[i], t1[i]) for i in range(n)]

    def stream_groups(self):
        """[(first_slot, n_slots), ...] of the LM stream groups."""
        a, b = (C.c_int * 8)(), (C.c_int * 8)()
        g = self._check(self.lib.dsm_lm_stream_groups(self.h, a, b, 8))
        return [(a[i], b[i]) for i in range(g)]

    def debug_serialize_groups(self, on):
        self._check(self.lib.dsm_debug_serialize_groups(self.h, 1 if on else 0))

    def prof_enable(self, tags):
        mask = 0
        for t in tags:
            mask |= 1 << PROF_TAGS.index(t)
        self._check(self.lib.dsm_prof_enable(self.h, mask))

    def prof_read(self):
        tot = (C.c_double * len(PROF_TAGS))()
        cnt = (C.c_uint64 * len(PROF_TAGS))()
        self._check(self.lib.dsm_prof_read(self.h, tot, cnt))
        return {t: (tot[i], cnt[i]) for i, t in enumerate(PROF_TAGS)}


class TtsEngine:
    """B independent tts_streaming::State generations behind the C ABI (core/tts_streaming.rs:117-242).
    step() mirrors State::step(prev_text_token, allowed_tokens, conditions=None) with greedy sampling."""

    def __init__(self, cfg, batch_size, lm_path, device_id=0):
        self.lib = load_library()
        self.cfg, self.B, self.S = cfg, batch_size, cfg.dep_num_slices
        h = C.c_void_p()
        rc = self.lib.dsm_tts_create(C.byref(cfg), device_id, batch_size, lm_path.encode(), C.byref(h))
        if rc != 0:
            msg = self.lib.dsm_tts_last_error(None)
            raise DsmError(f"dsm_tts_create failed ({rc}): {msg.decode() if msg else '?'}")
        self.h = h

    def _check(self, rc):
        if rc < 0:
            msg = self.lib.dsm_tts_last_error(self.h)
            raise DsmError(f"dsm error {rc}: {msg.decode() if msg else '?'}")
        return rc

    def close(self):
        if getattr(self, "h", None):
            self.lib.dsm_tts_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, prev_text_token, allowed, mask):
        prev = np.ascontiguousarray(prev_text_token, dtype=np.uint32).reshape(self.B)
        allowed = np.ascontiguousarray(allowed, dtype=np.int32).reshape(self.B)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        text = np.zeros(self.B, dtype=np.uint32)
        audio = np.zeros((self.B, self.S), dtype=np.uint32)
        self._check(self.lib.dsm_tts_step(self.h, _ptr(prev), _ptr(allowed), _ptr(mask), _ptr(text), _ptr(audio)))
        return text, audio

    def audio_tokens(self, slot, step):
        out = np.zeros(self.S, dtype=np.uint32)
        self._check(self.lib.dsm_tts_audio_tokens(self.h, slot, step, _ptr(out)))
        return out

    def step_idx(self, slot):
        return self._check(self.lib.dsm_tts_step_idx(self.h, slot))

    def reset_batch_idx(self, slot):
        self._check(self.lib.dsm_tts_reset_slot(self.h, slot))

    def debug_read(self, name, n):
        out = np.zeros(n, dtype=np.float32)
        got = self._check(self.lib.dsm_tts_debug_read(self.h, name.encode(), _ptr(out), n))
        return out[:got]

    def metrics(self):
        m = Metrics()
        self._check(self.lib.dsm_tts_get_metrics(self.h, C.byref(m)))
        return m

    def set_sampling(self, slot, top_k, temperature, seed):
        """Sampling::TopK{k, temperature} seeded with `seed` for the slot's text and audio processors (srv/tts.rs:401-415)."""
        self._check(self.lib.dsm_tts_set_sampling(self.h, slot, top_k, temperature, seed))

    def set_ca_src(self, slot, ca_src, ca_src_uncond=None, cfg_alpha=0.0):
        """State::new's ca_src / cfg_alpha for the slot (srv/tts.rs:426-441): ca_src [n][ca_dim] f32 or None;
        ca_src_uncond: the second batch row of a classifier-free-guidance request."""
        a = None if ca_src is None else np.ascontiguousarray(ca_src, dtype=np.float32)
        u = None if ca_src_uncond is None else np.ascontiguousarray(ca_src_uncond, dtype=np.float32)
        self._check(self.lib.dsm_tts_set_ca_src(self.h, slot, _ptr(a), 0 if a is None else a.shape[0], _ptr(u),
                                                0 if u is None else u.shape[0], float(cfg_alpha)))
