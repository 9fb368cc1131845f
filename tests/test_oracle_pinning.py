"""Pins the CPU oracle against everything the reference's own tests hold for the hot path
(SURVEY.md §8c):

  * core/kv_cache.rs:339-405  test_scattered_kv_cache — exact indices + masks (fixture
    tests/golden/kv_cache_kat.json, data only)
  * core/conv.rs:698-723      conv1d / conv_tr1d — streaming step() concatenated == batch forward(),
    |diff| <= 1e-5, on the reference's own (k, stride, dilation, step_size, len, bias) grid

Model outputs have no golden vectors anywhere in the reference (parity vs Candle itself is unpinned)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_kv_cache_known_answers(orc):
    L = orc.lib()
    kat = json.load(open(os.path.join(HERE, "golden", "kv_cache_kat.json")))
    B, ctx = kat["batch"], kat["context"]
    kb = L.orc_kvb_new(B, ctx)
    try:
        for call in kat["calls"]:
            T = call["seq_len"]
            mask = np.array(call["mask"], dtype=np.uint8)
            idx = np.zeros(B * T, dtype=np.uint32)
            att = np.zeros(B * T * ctx, dtype=np.float32)
            L.orc_kvb_indices_and_mask(kb, T, orc.p(mask), orc.p(idx), orc.p(att))
            assert idx.reshape(B, T).tolist() == call["indices"]
            want = np.array(call["attn"], dtype=np.float64)
            want = np.where(want < -1e29, -np.inf, want).astype(np.float32)
            assert np.array_equal(att.reshape(B, T, ctx), want), (att.reshape(B, T, ctx), want)
    finally:
        L.orc_kvb_free(kb)


def test_kv_builder_reset_and_long_run(orc):
    """reset_batch_index zeroes one slot only (core/kv_cache.rs:111-117); idx == pos mod ctx always."""
    L = orc.lib()
    B, ctx = 3, 7
    kb = L.orc_kvb_new(B, ctx)
    rng = np.random.default_rng(0)
    pos_ref = np.zeros(B, dtype=np.int64)
    for step in range(60):
        T = int(rng.integers(1, 3))
        mask = (rng.random(B) < 0.7).astype(np.uint8)
        idx = np.zeros(B * T, dtype=np.uint32)
        att = np.zeros(B * T * ctx, dtype=np.float32)
        L.orc_kvb_indices_and_mask(kb, T, orc.p(mask), orc.p(idx), orc.p(att))
        att = att.reshape(B, T, ctx)
        for b in range(B):
            if mask[b]:
                for t in range(T):
                    assert idx[b * T + t] == (pos_ref[b] + t) % ctx
                    # closed form used by the HIP kernel: visible iff j <= e1 and (e1 - j) % ctx >= T-1-t
                    e1 = pos_ref[b] + T - 1
                    vis = np.array([(j <= e1) and ((e1 - j) % ctx >= T - 1 - t) for j in range(ctx)])
                    assert np.array_equal(att[b, t] == 0, vis), (step, b, t)
                pos_ref[b] += T
            else:
                assert np.all(att[b] == 0)
        if step % 17 == 9:
            L.orc_kvb_reset_batch_index(kb, 1)
            pos_ref[1] = 0
        pos = np.zeros(B, dtype=np.uint32)
        ix = np.zeros(B, dtype=np.uint32)
        L.orc_kvb_get(kb, orc.p(pos), orc.p(ix))
        assert np.array_equal(pos, pos_ref.astype(np.uint32))
        assert np.array_equal(ix, (pos_ref % ctx).astype(np.uint32))
    L.orc_kvb_free(kb)


def _run_conv1d(orc, k, stride, dilation, step_size, length, bias, rng):
    L = orc.lib()
    in_c, out_c = 2, 3
    w = rng.standard_normal((out_c, in_c, k)).astype(np.float32)
    b = rng.standard_normal(out_c).astype(np.float32) if bias else None
    x = rng.standard_normal((1, step_size * length, in_c)).astype(np.float32)  # channels-last [B][T][C]
    conv = L.orc_conv1d_new(1, in_c, out_c, k, stride, dilation, 0, orc.p(w), orc.p(b))
    cap = step_size * length + 8
    y = np.zeros((1, cap, out_c), dtype=np.float32)
    n = L.orc_conv1d_forward(conv, orc.p(x), step_size * length, orc.p(y), cap)
    assert n >= 0
    L.orc_conv1d_reset_state(conv)
    outs = []
    for i in range(length):
        xs = np.ascontiguousarray(x[:, step_size * i:step_size * (i + 1)])
        ys = np.zeros((1, cap, out_c), dtype=np.float32)
        m = L.orc_conv1d_step(conv, orc.p(xs), step_size, None, orc.p(ys), cap)
        assert m >= 0
        outs.append(ys[:, :m])
    L.orc_conv1d_free(conv)
    ys = np.concatenate(outs, axis=1)
    assert ys.shape[1] == n, (ys.shape, n)
    assert np.abs(ys - y[:, :n]).max() <= 1e-5


def _run_convtr1d(orc, k, stride, step_size, length, bias, rng):
    L = orc.lib()
    in_c, out_c = 2, 3
    w = rng.standard_normal((in_c, out_c, k)).astype(np.float32)
    b = rng.standard_normal(out_c).astype(np.float32) if bias else None
    x = rng.standard_normal((1, step_size * length, in_c)).astype(np.float32)
    conv = L.orc_convtr1d_new(1, in_c, out_c, k, stride, 0, orc.p(w), orc.p(b))
    cap = step_size * length * stride + k + 8
    y = np.zeros((1, cap, out_c), dtype=np.float32)
    n = L.orc_convtr1d_forward(conv, orc.p(x), step_size * length, orc.p(y), cap)
    outs = []
    for i in range(length):
        xs = np.ascontiguousarray(x[:, step_size * i:step_size * (i + 1)])
        ys = np.zeros((1, cap, out_c), dtype=np.float32)
        m = L.orc_convtr1d_step(conv, orc.p(xs), step_size, None, orc.p(ys), cap)
        outs.append(ys[:, :m])
    L.orc_convtr1d_free(conv)
    ys = np.concatenate(outs, axis=1)
    assert ys.shape[1] == n
    assert np.abs(ys - y[:, :n]).max() <= 1e-5


def test_conv1d_streaming_equals_batch(orc):
    """The grid of core/conv.rs:698-710."""
    rng = np.random.default_rng(1)
    for step_size in (1, 2, 3):
        for bias in (False, True):
            _run_conv1d(orc, 1, 1, 1, step_size, 5, bias, rng)
            _run_conv1d(orc, 2, 1, 1, step_size, 5, bias, rng)
            _run_conv1d(orc, 2, 2, 1, step_size, 6, bias, rng)
            _run_conv1d(orc, 3, 2, 1, step_size, 8, bias, rng)
            _run_conv1d(orc, 3, 2, 2, step_size, 8, bias, rng)


def test_conv_tr1d_streaming_equals_batch(orc):
    """The grid of core/conv.rs:712-723."""
    rng = np.random.default_rng(2)
    for step_size in (1, 2, 3):
        for bias in (False, True):
            _run_convtr1d(orc, 1, 1, step_size, 5, bias, rng)
            _run_convtr1d(orc, 2, 1, step_size, 5, bias, rng)
            _run_convtr1d(orc, 3, 1, step_size, 5, bias, rng)
            _run_convtr1d(orc, 3, 2, step_size, 5, bias, rng)


def test_conv1d_mask_freezes_state(orc):
    """core/conv.rs:347-367: an inactive slot keeps its carried state; active neighbours advance;
    reset_batch_idx zeroes one slot only (core/conv.rs:274-281)."""
    L = orc.lib()
    rng = np.random.default_rng(3)
    B, in_c, out_c, k, s = 2, 2, 3, 4, 2
    w = rng.standard_normal((out_c, in_c, k)).astype(np.float32)
    mk = lambda: L.orc_conv1d_new(B, in_c, out_c, k, s, 1, 0, orc.p(w), None)
    a, ref = mk(), mk()
    T, cap = 4, 8
    xs = [rng.standard_normal((B, T, in_c)).astype(np.float32) for _ in range(6)]
    masks = [[1, 1], [1, 0], [1, 0], [1, 1], [1, 1], [1, 1]]
    y = np.zeros((B, cap, out_c), dtype=np.float32)
    outs_a = []
    for x, m in zip(xs, masks):
        L.orc_conv1d_step(a, orc.p(x), T, orc.p(np.array(m, dtype=np.uint8)), orc.p(y), cap)
        outs_a.append(y[:, :T // s].copy())
    # slot 1 of `a` saw frames 0,3,4,5 (1,2 masked out): same as a stream fed only those frames
    outs_r = []
    for i in (0, 3, 4, 5):
        L.orc_conv1d_step(ref, orc.p(xs[i]), T, orc.p(np.ones(B, dtype=np.uint8)), orc.p(y), cap)
        outs_r.append(y[:, :T // s].copy())
    for got, want in zip([outs_a[0], outs_a[3], outs_a[4], outs_a[5]], outs_r):
        assert np.array_equal(got[1], want[1])
    L.orc_conv1d_free(a)
    L.orc_conv1d_free(ref)
