"""How far apart are the two canonical dot products of the bf16-weight GEMMs (dsm_asr_config.dot_mode 0 / 1, include/dsm.h)?

Both modes are restated bit for bit by the oracle, so HIP == oracle says nothing about mode 1 against mode 0.  This module runs the
two modes side by side on the same engine build, the same weights (one arena) and the same audio, on full wrapped rings, with the
mode-1 engine TEACHER-FORCED along the mode-0 engine's text tokens (dsm_debug_set_text_tokens: one flipped argmax must not
cascade through the feedback of core/asr.rs:147-160), and reports the largest relative logit difference and the fraction of equal
text tokens.  Used by tests/test_dot_mode_agreement_gpu.py (asserts) and bench.py (`dot_mode_agreement` in the JSON line)."""
import numpy as np


def asr_agreement(dsm, cfg, B, lm_path, mimi_path, arena=None, fill=None, steps=60, seed=4242, device_id=0):
    from . import synth
    cfg0 = type(cfg).from_buffer_copy(cfg)
    cfg1 = type(cfg).from_buffer_copy(cfg)
    cfg0.dot_mode, cfg1.dot_mode = 0, 1
    e0 = dsm.AsrEngine(cfg0, B, lm_path, mimi_path, device_id=device_id, arena=arena)
    e1 = dsm.AsrEngine(cfg1, B, lm_path, mimi_path, device_id=device_id, arena=arena if arena is not None else e0.weight_arena())
    fill = cfg.lm.context if fill is None else fill
    n_pcm = 32
    pcm = synth.synth_pcm(B, n_pcm, seed=seed)
    mask = np.ones(B, dtype=np.uint8)
    V = cfg.text_out_vocab_size
    codes_equal, tok_eq, tok_n, max_rel, max_abs, flips = True, 0, 0, 0.0, 0.0, []
    prev0 = None
    for s in range(fill + steps):
        c0 = e0.encode_step(pcm[s % n_pcm], mask)
        c1 = e1.encode_step(pcm[s % n_pcm], mask)
        codes_equal = codes_equal and bool(np.array_equal(c0, c1))  # Mimi's weights are f32: the mode does not touch it
        if prev0 is not None:
            e1.debug_set_text_tokens(prev0)
        t0, _ = e0.step_tokens(c0, mask)
        t1, _ = e1.step_tokens(c0, mask)
        prev0 = t0.copy()
        if s < fill:
            continue
        l0 = e0.debug_read("lm.logits", B * V).reshape(B, V).astype(np.float64)
        l1 = e1.debug_read("lm.logits", B * V).reshape(B, V).astype(np.float64)
        scale = np.abs(l0).max(axis=1)
        diff = np.abs(l1 - l0).max(axis=1)
        max_abs = max(max_abs, float(diff.max()))
        max_rel = max(max_rel, float((diff / scale).max()))
        tok_eq += int((t0 == t1).sum())
        tok_n += B
        for b in np.nonzero(t0 != t1)[0]:
            top2 = np.sort(l0[b])[-2:]
            flips.append({"step": s - fill, "slot": int(b), "mode0_margin": float(top2[1] - top2[0])})
    e0.close(); e1.close()
    return {"steps": steps, "batch": B, "ring_fill_frames": fill, "mimi_codes_identical": codes_equal,
            "max_rel_logit_err": max_rel, "max_abs_logit_err": max_abs, "text_token_agreement": tok_eq / max(tok_n, 1),
            "tokens_compared": tok_n, "flips": flips[:16], "teacher_forced": True}


def tts_agreement(dsm, cfg, B, path, steps=40):
    """TTS State::step in the two modes, greedy, the caller's text token teacher-forced from mode 0 (the API takes it as an
    argument); the audio tokens feed back inside the engine (depformer slice k reads slice k - 1's token, the LM reads the
    delayed audio history), so an audio flip does cascade: agreement is reported up to and after the first divergence."""
    from . import synth
    cfg0 = type(cfg).from_buffer_copy(cfg)
    cfg1 = type(cfg).from_buffer_copy(cfg)
    cfg0.dot_mode, cfg1.dot_mode = 0, 1
    e0, e1 = dsm.TtsEngine(cfg0, B, path), dsm.TtsEngine(cfg1, B, path)
    if cfg.cross_attention:
        for b in range(B):
            src = synth.synth_ca_src(cfg, 24 + 3 * b, 100 + b)
            e0.set_ca_src(b, src); e1.set_ca_src(b, src)
    mask = np.ones(B, dtype=np.uint8)
    prev = np.full(B, cfg.text_start_token, dtype=np.uint32)
    allowed = np.full(B, dsm.TTS_ALLOW_PAD_OR_EPAD, dtype=np.int32)
    text_eq = audio_eq = text_n = audio_n = 0
    first_div = None
    d = cfg.lm.d_model
    rows = B * (2 if cfg.cfg_rows else 1)
    max_rel_hidden = 0.0
    for s in range(steps):
        t0, a0 = e0.step(prev, allowed, mask)
        t1, a1 = e1.step(prev, allowed, mask)
        if first_div is None:
            h0 = e0.debug_read("lm.hidden", rows * d).astype(np.float64)
            h1 = e1.debug_read("lm.hidden", rows * d).astype(np.float64)
            max_rel_hidden = max(max_rel_hidden, float(np.abs(h1 - h0).max() / np.abs(h0).max()))
        text_eq += int((t0 == t1).sum()); text_n += B
        gen = a0 != dsm.TTS_UNGENERATED
        audio_eq += int((a0[gen] == a1[gen]).sum()); audio_n += int(gen.sum())
        if first_div is None and (not np.array_equal(t0, t1) or not np.array_equal(a0, a1)):
            first_div = s
        prev = t0.astype(np.uint32)
    e0.close(); e1.close()
    return {"steps": steps, "batch": B, "text_token_agreement": text_eq / max(text_n, 1),
            "audio_token_agreement": audio_eq / max(audio_n, 1), "audio_tokens_compared": audio_n,
            "first_divergent_step": first_div, "max_rel_lm_hidden_err_before_divergence": max_rel_hidden,
            "teacher_forced": "text token only (audio tokens feed back inside the engine)"}
