"""Steady-state parity of the attention kernel and the batched ring cache.

`attn_kernel` streams K and V in a two-register-set software pipeline that advances STEP = UNR * 4 waves * G keys per
iteration (bf16 hd 128: 128 keys, bf16 hd 64: 256, f32 hd 128: 64, f32 hd 64: 128, f32 hd 32: 256).  The tiny model's
rings (12 / 10 frames) and the few-frame real-shape tests never leave the first iteration.  The medium configurations
here keep the REAL head dims with rings of 150..600 frames and run for ctx + 2*STEP frames and more, so every test step
past the fill runs both register sets (`ra` / `rb`), the second and later prefetches, the tail clamp in iteration >= 2
and — after `ctx` frames — the wrapped ring (core/batched_transformer.rs:97-113, core/kv_cache.rs:130-237), with mixed
masks and mid-stream slot resets, for the LM (T = 1, fused QKV-reduce prologue because d_model spans two K-chunks) and
for the Mimi transformer (T = 2).  Every float tap is compared bit for bit with the oracle at every step."""
import os

import numpy as np
import pytest

from test_parity_gpu import run_pair

pytestmark = pytest.mark.gpu

WEIGHTS_DIR = os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights")

# (name, config_medium kwargs, attention STEP of the LM kernel, frames to run)
CASES = [
    # bf16 ring, hd 128 (stt-1b's kernel): STEP 128, ring 300 = 2 full iterations + a 44-key tail; Mimi f32 hd 64 T=2 (STEP 128, ring 250: real Mimi)
    ("bf16_hd128_ctx300", dict(lm_heads=4, lm_head_dim=128, lm_context=300, kv_bf16=1), 128, 300 + 2 * 128 + 24),
    # bf16 ring, hd 64 (stt-2.6b's kernel): STEP 256, ring 600 = 2 full iterations + tail
    ("bf16_hd64_ctx600", dict(lm_heads=8, lm_head_dim=64, lm_context=600, kv_bf16=1), 256, 600 + 2 * 256 + 24),
    # f32 ring (the Candle-CPU dtype), hd 128: STEP 64; Mimi f32 hd 128 T=2 (STEP 64) on a 150-frame ring
    ("f32_hd128_ctx150", dict(lm_heads=4, lm_head_dim=128, lm_context=150, kv_bf16=0, mimi_head_dim=128, mimi_context=150), 64, 150 + 2 * 64 + 24),
    # f32 ring, hd 64: STEP 128; Mimi f32 hd 32 T=2 (STEP 256) on a 600-frame ring (fills after 300 steps)
    ("f32_hd64_ctx300", dict(lm_heads=8, lm_head_dim=64, lm_context=300, kv_bf16=0, mimi_head_dim=32, mimi_context=600), 128, 300 + 2 * 128 + 24),
]


@pytest.mark.parametrize("name,kw,step_keys,frames", CASES, ids=[c[0] for c in CASES])
def test_attention_steady_state_ring_wrap_masks_resets(gpu, dsm, lib, orc, name, kw, step_keys, frames):
    from dsm_amd import synth
    cfg = dsm.config_medium(**kw)
    assert cfg.lm.d_model > 256, "two K-chunks: the QKV GEMM leaves split-K slabs to the attention prologue"
    ctx = cfg.lm.context
    assert ctx > 2 * step_keys and frames >= ctx + 2 * step_keys, "the pipelined loop must reach its third iteration, then wrap"
    lm, mimi = synth.make_synth_weights(cfg, WEIGHTS_DIR, tag="medium_" + name)
    B = 5
    rng = np.random.default_rng(11)
    masks = (rng.random((frames, B)) < 0.8).astype(np.uint8)
    masks[:, 0] = 1                       # slot 0 never pauses: first to wrap
    masks[: ctx // 2, 3] = 0              # slot 3 joins late: its ring is at a different fill than the others'
    resets = {ctx // 3: [1], ctx + 7: [2], ctx + step_keys + 3: [0]}  # before the fill, just after the wrap, deep into it
    run_pair(dsm, orc, cfg, B, lm, mimi, steps=frames, mask_fn=lambda s: masks[s], resets=resets)
