"""ctypes wrapper of oracle/libdsm_oracle.so — TEST INFRASTRUCTURE (see oracle/dsm_oracle.h).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

import dsm_amd  # noqa: F401  (registers the package)
from dsm_amd import AsrConfig, AsrMsg, TtsConfig, FRAME_SIZE, MSG_WORD, MSG_END_WORD

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libdsm_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        # a GPU box shows every host core but grants a 16-core share: oversubscribed OpenMP is 10x slower
        os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        vp = C.c_void_p
        L.orc_asr_create.argtypes = [C.POINTER(AsrConfig), C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_asr_create.restype = vp
        L.orc_asr_destroy.argtypes = [vp]
        L.orc_asr_destroy.restype = None
        L.orc_mimi_encode_step.argtypes = [vp, C.c_int, vp, vp, vp]
        L.orc_asr_step_tokens.argtypes = [vp, vp, vp, vp, vp]
        L.orc_mimi_decode_step.argtypes = [vp, C.c_int, vp, vp, vp]
        L.orc_asr_reset_slot.argtypes = [vp, C.c_int]
        L.orc_asr_set_seed.argtypes = [vp, C.c_int, C.c_uint64]
        L.orc_logf.argtypes = [C.c_float]
        L.orc_logf.restype = C.c_float
        L.orc_mimi_reset_slot.argtypes = [vp, C.c_int, C.c_int]
        L.orc_asr_poll_msgs.argtypes = [vp, C.POINTER(AsrMsg), C.c_int, vp, C.c_int]
        L.orc_debug_read.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
        L.orc_debug_set_positions.argtypes = [vp, C.c_uint32, C.c_uint32]
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads.restype = None
        L.orc_dot.argtypes = [vp, vp, C.c_int]
        L.orc_dot.restype = C.c_float
        L.orc_linear.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
        L.orc_linear.restype = None
        L.orc_rmsnorm.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float]
        L.orc_rmsnorm.restype = None
        L.orc_layernorm.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_float]
        L.orc_layernorm.restype = None
        L.orc_attention_head.argtypes = [vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp]
        L.orc_attention_head.restype = None
        L.orc_rope_table.argtypes = [C.c_int, C.c_int, vp]
        L.orc_rope_table.restype = None
        L.orc_rope_apply.argtypes = [vp, C.c_int, vp, C.c_uint32]
        L.orc_rope_apply.restype = None
        L.orc_kvb_new.argtypes = [C.c_int, C.c_int]
        L.orc_kvb_new.restype = vp
        L.orc_kvb_free.argtypes = [vp]
        L.orc_kvb_free.restype = None
        L.orc_kvb_reset_batch_index.argtypes = [vp, C.c_int]
        L.orc_kvb_reset_batch_index.restype = None
        L.orc_kvb_indices_and_mask.argtypes = [vp, C.c_int, vp, vp, vp]
        L.orc_kvb_indices_and_mask.restype = None
        L.orc_kvb_get.argtypes = [vp, vp, vp]
        L.orc_kvb_get.restype = None
        L.orc_conv1d_new.argtypes = [C.c_int] * 7 + [vp, vp]
        L.orc_conv1d_new.restype = vp
        L.orc_conv1d_free.argtypes = [vp]
        L.orc_conv1d_free.restype = None
        L.orc_conv1d_step.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int]
        L.orc_conv1d_forward.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.orc_conv1d_reset_state.argtypes = [vp]
        L.orc_conv1d_reset_state.restype = None
        L.orc_conv1d_reset_batch_idx.argtypes = [vp, C.c_int]
        L.orc_conv1d_reset_batch_idx.restype = None
        L.orc_convtr1d_new.argtypes = [C.c_int] * 6 + [vp, vp]
        L.orc_convtr1d_new.restype = vp
        L.orc_convtr1d_free.argtypes = [vp]
        L.orc_convtr1d_free.restype = None
        L.orc_convtr1d_step.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int]
        L.orc_convtr1d_forward.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.orc_convtr1d_reset_batch_idx.argtypes = [vp, C.c_int]
        L.orc_convtr1d_reset_batch_idx.restype = None
        L.orc_tts_create.argtypes = [C.POINTER(TtsConfig), C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_tts_create.restype = vp
        L.orc_tts_destroy.argtypes = [vp]
        L.orc_tts_destroy.restype = None
        L.orc_tts_step.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_tts_audio_tokens.argtypes = [vp, C.c_int, C.c_int, vp]
        L.orc_tts_step_idx.argtypes = [vp, C.c_int]
        L.orc_tts_reset_slot.argtypes = [vp, C.c_int]
        L.orc_tts_debug_read.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
        L.orc_tts_set_sampling.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_uint64]
        L.orc_tts_set_ca_src.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_double]
        L.orc_sample_topk.argtypes = [vp, C.c_int, C.c_int, C.c_float, vp, vp]
        L.orc_sample_topk.restype = C.c_uint32
        L.orc_chacha_word.argtypes = [vp, C.c_uint64, C.c_int]
        L.orc_chacha_word.restype = C.c_uint32
        L.orc_seed_from_u64.argtypes = [C.c_uint64, vp]
        L.orc_seed_from_u64.restype = None
        L.orc_uniform_f32.argtypes = [C.c_uint32, C.c_float]
        L.orc_uniform_f32.restype = C.c_float
        _lib = L
    return _lib


def set_num_threads(n):
    """OpenMP team size of the oracle from now on (small models step faster with few threads on a shared box)."""
    lib().orc_set_num_threads(int(n))


def default_num_threads():
    return int(os.environ.get("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1))))


def p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle wants C-contiguous arrays"
    return a.ctypes.data_as(C.c_void_p)


class OracleAsr:
    """Same surface as dsm_amd.AsrEngine, computed by the CPU restatement."""

    def __init__(self, cfg, batch_size, lm_path, mimi_path):
        self.L = lib()
        self.cfg, self.B = cfg, batch_size
        # small models: a big OpenMP team only spins at barriers (on a GPU box with a small CPU share, 25x slower)
        set_num_threads(4 if cfg.lm.d_model <= 512 else default_num_threads())
        err = C.create_string_buffer(512)
        self.h = self.L.orc_asr_create(C.byref(cfg), batch_size, lm_path.encode(), mimi_path.encode(), err, 512)
        if not self.h:
            raise RuntimeError("oracle create failed: " + err.value.decode())
        self.n_q = cfg.mimi.quantizer_n_q

    def close(self):
        if getattr(self, "h", None):
            self.L.orc_asr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def encode_step(self, pcm, mask, side=0):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32).reshape(self.B, FRAME_SIZE)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        codes = np.zeros((self.B, self.n_q), dtype=np.uint32)
        n = self.L.orc_mimi_encode_step(self.h, side, p(pcm), p(mask), p(codes))
        return codes if n else None

    def decode_step(self, codes, mask, side=0):
        codes = np.ascontiguousarray(codes, dtype=np.uint32).reshape(self.B, self.n_q)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        pcm = np.zeros((self.B, FRAME_SIZE), dtype=np.float32)
        n = self.L.orc_mimi_decode_step(self.h, side, p(codes), p(mask), p(pcm))
        return pcm if n > 0 else None

    def step_tokens(self, codes, mask):
        codes = np.ascontiguousarray(codes, dtype=np.uint32).reshape(self.B, self.n_q)
        mask = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.B)
        text = np.zeros(self.B, dtype=np.uint32)
        nh = self.cfg.extra_heads_num
        prs = np.zeros((max(nh, 1), self.B), dtype=np.float32)
        self.L.orc_asr_step_tokens(self.h, p(codes), p(mask), p(text), p(prs) if nh else None)
        return text, prs[:nh]

    def step_pcm(self, pcm, mask):
        codes = self.encode_step(pcm, mask, side=1)
        text, prs = self.step_tokens(codes, mask)
        return codes, text, prs

    def poll_msgs(self, cap=4096):
        msgs = (AsrMsg * cap)()
        toks = np.zeros(cap * 8, dtype=np.uint32)
        n = self.L.orc_asr_poll_msgs(self.h, msgs, cap, p(toks), toks.size)
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
        self.L.orc_asr_reset_slot(self.h, slot)

    def set_seed(self, slot, seed):
        self.L.orc_asr_set_seed(self.h, slot, seed)

    def mimi_reset_batch_idx(self, slot, side=0):
        self.L.orc_mimi_reset_slot(self.h, side, slot)

    def debug_set_positions(self, lm_pos, mimi_pos):
        """Mirror of AsrEngine.debug_set_positions: every ring as if that many frames had been appended."""
        self.L.orc_debug_set_positions(self.h, lm_pos, mimi_pos)

    def debug_read(self, name, n):
        out = np.zeros(n, dtype=np.float32)
        got = self.L.orc_debug_read(self.h, name.encode(), p(out), n)
        if got < 0:
            raise KeyError(name)
        return out[:got]


class OracleTts:
    """Same surface as dsm_amd.TtsEngine, computed by the CPU restatement (oracle/dsm_oracle_tts.inc)."""

    def __init__(self, cfg, batch_size, lm_path):
        self.L = lib()
        self.cfg, self.B, self.S = cfg, batch_size, cfg.dep_num_slices
        set_num_threads(4 if cfg.lm.d_model <= 512 else default_num_threads())
        err = C.create_string_buffer(512)
        self.h = self.L.orc_tts_create(C.byref(cfg), batch_size, lm_path.encode(), err, 512)
        if not self.h:
            raise RuntimeError("oracle create failed: " + err.value.decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.orc_tts_destroy(self.h)
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
        rc = self.L.orc_tts_step(self.h, p(prev), p(allowed), p(mask), p(text), p(audio))
        if rc < 0:
            raise RuntimeError(f"oracle tts step failed ({rc})")
        return text, audio

    def audio_tokens(self, slot, step):
        out = np.zeros(self.S, dtype=np.uint32)
        self.L.orc_tts_audio_tokens(self.h, slot, step, p(out))
        return out

    def step_idx(self, slot):
        return self.L.orc_tts_step_idx(self.h, slot)

    def reset_batch_idx(self, slot):
        self.L.orc_tts_reset_slot(self.h, slot)

    def set_sampling(self, slot, top_k, temperature, seed):
        self.L.orc_tts_set_sampling(self.h, slot, top_k, temperature, seed)

    def set_ca_src(self, slot, ca_src, ca_src_uncond=None, cfg_alpha=0.0):
        a = None if ca_src is None else np.ascontiguousarray(ca_src, dtype=np.float32)
        u = None if ca_src_uncond is None else np.ascontiguousarray(ca_src_uncond, dtype=np.float32)
        rc = self.L.orc_tts_set_ca_src(self.h, slot, None if a is None else p(a), 0 if a is None else a.shape[0],
                                       None if u is None else p(u), 0 if u is None else u.shape[0], float(cfg_alpha))
        if rc < 0:
            raise RuntimeError(f"oracle set_ca_src failed ({rc})")

    def debug_read(self, name, n):
        out = np.zeros(n, dtype=np.float32)
        got = self.L.orc_tts_debug_read(self.h, name.encode(), p(out), n)
        if got < 0:
            raise KeyError(name)
        return out[:got]
