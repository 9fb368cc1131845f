"""MPEG-1 Layer III decoder + the windowed-sinc resampler (csrc/dsm_mp3.inc; SURVEY.md §8(f) rank 3: `pcm_decode` of
srv/utils.rs:263-305 and `kaudio::resample` of srv/batched_asr.rs:838 for the mp3 bodies the reference's own samples use).

symphonia / kaudio are un-vendored crates and nothing importable here decodes mp3, so the PCM is PARITY-UNPINNED against the
reference's decoder.  What these tests pin instead:
  * the Huffman tables are complete prefix codes over every (x, y) of their alphabet (Kraft sum exactly 1, no code a prefix of
    another) — a transcription error cannot survive that;
  * the synthesis filter bank reconstructs what the standard's ANALYSIS bank (written here in numpy from its formulas) split:
    > 80 dB; the IMDCT equals its defining sum and cancels time-domain aliasing against a numpy MDCT;
  * a numpy mini ENCODER (analysis bank, MDCT, alias butterflies, quantiser, Huffman writer with these tables: long blocks,
    linbits, count1 region) round-trips three tones through the whole decoder: > 55 dB;
  * on the reference's own files (two as committed fixtures, all five where /root/reference is present): the frame walk finds
    every frame the size implies, loses sync nowhere, every granule's Huffman data ends inside its part2_3_length, the PCM is
    frames x 1152 samples of band-limited speech (energy above 19 kHz < 1e-8 of the total, no clicks at granule boundaries);
  * the resampler: unit DC gain, a tone keeps frequency and level, a tone above the new Nyquist is gone (> 80 dB)."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUD = os.path.join(ROOT, "tests", "golden", "audio")
import ctypes as C


def _tables(lib, t):
    xlen, lb = C.c_int(), C.c_int()
    codes = (C.c_uint16 * 256)()
    lens = (C.c_uint8 * 256)()
    n = lib.dsm_mp3_test_tables(t, C.byref(xlen), C.byref(lb), codes, lens, 256)
    return xlen.value, lb.value, list(codes[:n]), list(lens[:n])


def test_huffman_tables_are_complete_prefix_codes(dsm, lib):
    from fractions import Fraction
    linbits = {16: 1, 17: 2, 18: 3, 19: 4, 20: 6, 21: 8, 22: 10, 23: 13, 24: 4, 25: 5, 26: 6, 27: 7, 28: 8, 29: 9, 30: 11, 31: 13}
    sizes = {1: 2, 2: 3, 3: 3, 5: 4, 6: 4, 7: 6, 8: 6, 9: 6, 10: 8, 11: 8, 12: 8, 13: 16, 15: 16, 32: 4, 33: 4}
    for t in range(34):
        xlen, lb, codes, lens = _tables(lib, t)
        if t in (0, 4, 14):
            assert not codes
            continue
        assert xlen == sizes.get(t, 16) and lb == linbits.get(t, 0) and len(codes) == xlen * xlen
        assert sum(Fraction(1, 2 ** l) for l in lens) == 1, f"table {t}: Kraft sum"
        words = sorted(format(c, "0%db" % l) for c, l in zip(codes, lens))
        assert all(c < 2 ** l for c, l in zip(codes, lens)) and len(set(words)) == len(words)
        assert not any(b.startswith(a) for a, b in zip(words, words[1:])), f"table {t}: not prefix-free"
    # tables 16-23 share one code set, 24-31 another; count1 table B is the inverted 4-bit pattern
    assert _tables(lib, 16)[2:] == _tables(lib, 23)[2:] and _tables(lib, 24)[2:] == _tables(lib, 31)[2:] and _tables(lib, 16)[2] != _tables(lib, 24)[2]
    assert _tables(lib, 33)[2] == [15 - i for i in range(16)]


# ---- the standard's encoder-side transforms in numpy (ISO/IEC 11172-3 Annex C) ------------------------------------------------
def _window(lib):
    """The synthesis window D[i] as the decoder applies it, recovered from unit impulses through dsm_mp3_test_synth (slot 0 of
    subband 0: V[i] = cos((16 + i) pi / 64), out[j] of slot s = V[64 s' ...]) is awkward; the prototype is a fixture instead."""
    base = json.load(open(os.path.join(AUD, "mp3_window_base.json")))
    d = np.zeros(512)
    for i in range(512):
        b = base[i] if i <= 256 else base[512 - i]
        d[i] = (-b if (i >> 6) & 1 else b) / 65536.0
    return d


def _analysis(x, D):
    """C.1.3: 32 new samples per slot -> 32 subband samples.  C[i] = D[i] / 32."""
    Cw = D / 32.0
    n = len(x) // 32
    M = np.cos((2 * np.arange(32)[:, None] + 1) * (np.arange(64)[None, :] - 16) * np.pi / 64)
    fifo = np.zeros(512)
    out = np.zeros((n, 32))
    for s in range(n):
        fifo[32:] = fifo[:-32]
        fifo[:32] = x[32 * s:32 * s + 32][::-1]
        z = Cw * fifo
        y = z.reshape(8, 64).sum(axis=0)
        out[s] = M @ y
    return out


def _synth(lib, sub):
    sub = np.ascontiguousarray(sub, dtype=np.float32)
    pcm = np.zeros(sub.shape[0] * 32, dtype=np.float32)
    assert lib.dsm_mp3_test_synth(sub.ctypes.data, sub.shape[0], pcm.ctypes.data) == 0
    return pcm.astype(np.float64)


def _snr_db(ref, got):
    return 10 * np.log10(np.sum(ref ** 2) / max(np.sum((ref - got) ** 2), 1e-300))


def test_synthesis_bank_inverts_the_standards_analysis_bank(dsm, lib):
    rng = np.random.default_rng(5)
    n = 32 * 400
    x = rng.standard_normal(n) * 0.2 + 0.3 * np.sin(2 * np.pi * 997 / 44100 * np.arange(n))
    y = _synth(lib, _analysis(x, _window(lib)))
    delay = 481  # 512 - 31
    snr = _snr_db(x[1000:n - delay - 1000], y[1000 + delay:n - 1000])
    assert snr > 80, snr


def _imdct(lib, spec, bt):
    spec = np.ascontiguousarray(spec, dtype=np.float32)
    out = np.zeros(36, dtype=np.float32)
    assert lib.dsm_mp3_test_imdct(spec.ctypes.data, bt, out.ctypes.data) == 0
    return out.astype(np.float64)


def _win(bt):
    i = np.arange(36)
    w = np.sin(np.pi / 36 * (i + 0.5))
    if bt == 1:
        w = np.where(i < 18, w, np.where(i < 24, 1.0, np.where(i < 30, np.sin(np.pi / 12 * (i - 18 + 0.5)), 0.0)))
    if bt == 3:
        w = np.where(i < 6, 0.0, np.where(i < 12, np.sin(np.pi / 12 * (i - 6 + 0.5)), np.where(i < 18, 1.0, w)))
    return w


def test_imdct_is_its_defining_sum_and_cancels_aliasing(dsm, lib):
    rng = np.random.default_rng(9)
    C36 = np.cos(np.pi / 72 * (2 * np.arange(36)[:, None] + 1 + 18) * (2 * np.arange(18)[None, :] + 1))
    C12 = np.cos(np.pi / 24 * (2 * np.arange(12)[:, None] + 1 + 6) * (2 * np.arange(6)[None, :] + 1))
    for bt in (0, 1, 3):
        X = rng.standard_normal(18)
        assert np.allclose(_imdct(lib, X, bt), (C36 @ X) * _win(bt), atol=2e-5)
    X = rng.standard_normal(18)
    ref = np.zeros(36)
    ws = np.sin(np.pi / 12 * (np.arange(12) + 0.5))
    for w in range(3):
        ref[6 * w + 6:6 * w + 18] += (C12 @ X[w::3]) * ws
    assert np.allclose(_imdct(lib, X, 2), ref, atol=2e-5)
    # TDAC, long blocks: MDCT (numpy, sine window) -> IMDCT (library) -> overlap-add gives the signal back times 18 / 2
    x = rng.standard_normal(18 * 12)
    out = np.zeros_like(x)
    for b in range(11):
        blk = x[18 * b:18 * b + 36] * _win(0)
        y = _imdct(lib, C36.T @ blk, 0)
        out[18 * b:18 * b + 36] += y
    assert _snr_db(x[18:-18] * 9.0, out[18:-18]) > 100


# ---- a minimal Layer III encoder (mono, long blocks, no reservoir, 320 kbps frames) to drive the WHOLE decoder ------------------
SFB_LONG_44 = [0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 52, 62, 74, 90, 110, 134, 162, 196, 238, 288, 342, 418, 576]


class _BitWriter:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))

    def bytes(self, nbytes):
        b = self.bits + [0] * (8 * nbytes - len(self.bits))
        assert len(b) == 8 * nbytes, "frame budget exceeded"
        return bytes(int("".join(map(str, b[8 * i:8 * i + 8])), 2) for i in range(nbytes))


def _spectrum(lib, x):
    """Analysis bank, MDCT and the encoder's alias butterflies: the 576 lines of every granule."""
    D = _window(lib)
    sub = _analysis(x, D)                       # [slots][32]
    ngr = sub.shape[0] // 18
    ci = np.array([-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037])
    cs, ca = 1 / np.sqrt(1 + ci * ci), ci / np.sqrt(1 + ci * ci)
    C36 = np.cos(np.pi / 72 * (2 * np.arange(36)[:, None] + 1 + 18) * (2 * np.arange(18)[None, :] + 1))
    w = _win(0)
    prev = np.zeros((32, 18))
    grs = []
    for g in range(ngr):
        cur = sub[18 * g:18 * g + 18].T.copy()  # [32][18]
        cur[1::2, 1::2] *= -1.0                  # frequency inversion compensation
        xr = np.zeros(576)
        for sb in range(32):
            xr[18 * sb:18 * sb + 18] = (C36.T @ (np.concatenate([prev[sb], cur[sb]]) * w)) / 9.0
        prev = cur
        for sb in range(31):                     # the encoder's alias butterflies (inverse rotation of the decoder's)
            for k in range(8):
                lo, hi = 18 * sb + 17 - k, 18 * (sb + 1) + k
                a, b = xr[lo], xr[hi]
                xr[lo], xr[hi] = a * cs[k] + b * ca[k], b * cs[k] - a * ca[k]
        grs.append(xr)
    return grs


def _quant(xr, gg):
    return np.rint(np.abs(xr / 2.0 ** ((gg - 210) / 4.0)) ** 0.75).astype(int)


def _pack(lib, grs, gg):
    """Quantise with one global gain (scale factors 0) and write 320 kbps mono frames: region 0 = lines [0, 162) with table 31
    (linbits 13), region 1 = the rest of big_values with table 26 (linbits 6), count1 region with table A."""
    t31, t26, tA = _tables(lib, 31), _tables(lib, 26), _tables(lib, 32)
    frames = []
    stats = {"count1_quads": 0, "linbits_used": 0}
    for f in range(len(grs) // 2):
        bw = _BitWriter()
        bw.put(0xFFFB, 16)                       # sync, MPEG-1, Layer III, no CRC
        bw.put(14, 4); bw.put(0, 2); bw.put(0, 1); bw.put(0, 1)   # 320 kbps, 44.1 kHz, no padding, private
        bw.put(3, 2); bw.put(0, 2); bw.put(0, 4)                   # mono
        side = _BitWriter()
        side.put(0, 9); side.put(0, 5); side.put(0, 4)             # main_data_begin 0, private, scfsi
        main = _BitWriter()
        for g in range(2):
            xr = grs[2 * f + g]
            ix = _quant(xr, gg)
            sg = xr < 0
            nz = np.nonzero(ix)[0]
            last = int(nz[-1]) + 1 if len(nz) else 0
            big = np.nonzero(ix > 1)[0]
            bv = (int(big[-1]) + 2) // 2 * 2 if len(big) else 0   # big_values region: up to the last |ix| > 1, even
            c1_end = bv + ((max(last - bv, 0) + 3) // 4) * 4
            assert c1_end <= 576
            start = len(main.bits)
            r1, r2 = SFB_LONG_44[16], 576                          # region0_count 15, region1_count 7
            for i in range(0, bv, 2):
                xlen, lb, codes, lens = t31 if i < r1 else t26
                xy = []
                for v in ix[i:i + 2]:
                    assert v <= 15 + (2 ** lb - 1 if lb else 0), (v, i)
                    xy.append(min(int(v), 15))
                idx = xy[0] * xlen + xy[1]
                main.put(codes[idx], lens[idx])
                for j, v in enumerate(ix[i:i + 2]):
                    if lb and xy[j] == 15:
                        main.put(int(v) - 15, lb)
                        stats["linbits_used"] += 1
                    if v:
                        main.put(int(sg[i + j]), 1)
            for i in range(bv, c1_end, 4):
                q = [int(v) for v in ix[i:i + 4]]
                assert max(q) <= 1
                idx = q[0] * 8 + q[1] * 4 + q[2] * 2 + q[3]
                main.put(tA[2][idx], tA[3][idx])
                stats["count1_quads"] += 1
                for j in range(4):
                    if q[j]:
                        main.put(int(sg[i + j]), 1)
            p23 = len(main.bits) - start
            assert p23 < 4096
            side.put(p23, 12); side.put(bv // 2, 9); side.put(gg, 8); side.put(0, 4); side.put(0, 1)
            side.put(31, 5); side.put(26, 5); side.put(26, 5)      # table_select: region 0 (linbits 13), 1 (linbits 6), 2 (empty)
            side.put(15, 4); side.put(7, 3)                         # region0_count, region1_count
            side.put(0, 1); side.put(0, 1); side.put(0, 1)          # preflag, scalefac_scale, count1 table A
        frame = bw.bytes(4) + side.bytes(17) + main.bytes(1044 - 21)
        frames.append(frame)
    return b"".join(frames), stats


def test_a_numpy_encoder_round_trips_through_the_decoder(dsm, lib):
    n = 1152 * 24
    t = np.arange(n) / 44100.0
    # 1 kHz (region 0: values far beyond 15, linbits 13), 8 kHz (region 1: linbits 6), and a 15 kHz component whose amplitude is
    # set to the quantiser's step so that its lines come out as +-1 above the last big value (the count1 region)
    x12 = 0.25 * np.sin(2 * np.pi * 1000 * t) + 0.02 * np.sin(2 * np.pi * 8000 * t + 0.3)
    g12 = _spectrum(lib, x12)
    gg = next(g for g in range(100, 220) if all(_quant(xr, g)[:162].max() <= 8000 and _quant(xr, g)[162:].max() <= 78 for xr in g12))
    tone3 = np.sin(2 * np.pi * 15000 * t + 1.0)
    peak3 = max(np.abs(xr).max() for xr in _spectrum(lib, tone3)[4:-4])
    a3 = 1.3 * 2.0 ** ((gg - 210) / 4.0) / peak3
    x = x12 + a3 * tone3
    data, stats = _pack(lib, _spectrum(lib, x), gg)
    assert stats["count1_quads"] > 500 and stats["linbits_used"] > 200, stats
    info = dsm.mp3_probe(data)
    assert (info["frames"], info["bitrate_kbps"], info["channels"], info["resyncs"]) == (24, 320, 1, 0)
    pcm, rate, di = dsm.mp3_decode(data)
    assert rate == 44100 and len(pcm) == n and di["huffman_overruns"] == 0 and di["frames_without_reservoir"] == 0
    delay = 481 + 576  # analysis + synthesis bank, one granule of MDCT overlap
    ref, got = x[4000:n - delay - 4000], pcm.astype(np.float64)[4000 + delay:n - 4000]
    assert _snr_db(ref, got) > 50, _snr_db(ref, got)
    # the two strong tones alone (the faint one sits at the quantiser's floor): project on them
    for f, a in ((1000, 0.25), (8000, 0.02)):
        tt = (np.arange(len(got)) + 4000) / 44100.0
        amp = 2 * np.abs(np.mean(got * np.exp(-2j * np.pi * f * tt)))
        assert abs(amp / a - 1) < 0.01, (f, amp)


# ---- the reference's own files -------------------------------------------------------------------------------------------------
def _band_energy(x, rate):
    seg = 4096
    w = np.hanning(seg)
    P = np.zeros(seg // 2 + 1)
    for i in range(0, len(x) - seg, seg // 2):
        P += np.abs(np.fft.rfft(x[i:i + seg] * w)) ** 2
    return np.fft.rfftfreq(seg, 1.0 / rate), P


def _check_decode(dsm, data, frames, info_frames=None):
    pi = dsm.mp3_probe(data)
    pcm, rate, di = dsm.mp3_decode(data)
    assert pi["frames"] == di["frames"] == frames and rate == 44100 and pi["channels"] == 1
    if info_frames is not None:
        assert pi["info_frames"] == info_frames
    assert pi["resyncs"] == 0 and pi["junk_bytes"] == 0
    assert len(pcm) == frames * 1152
    assert di["huffman_overruns"] == 0
    x = pcm.astype(np.float64)
    assert np.all(np.isfinite(x)) and np.abs(x).max() < 1.0
    rms = np.sqrt(np.mean(x ** 2))
    assert 0.01 < rms < 0.3, rms
    fr, P = _band_energy(x, rate)
    assert P[fr < 4000].sum() / P.sum() > 0.8            # speech
    assert P[fr >= 19000].sum() / P.sum() < 1e-8         # the encoder's low-pass survives: no imaging from the filter bank
    d = np.abs(np.diff(x))
    edges = np.arange(576, len(x) - 1, 576)
    assert d[edges - 1].mean() < 1.5 * d.mean()          # no clicks where granules meet
    return pcm, di


def test_loona_mp3_id3v2_info_frame_vbr(dsm, lib):
    data = open(os.path.join(AUD, "loona.mp3"), "rb").read()
    pcm, di = _check_decode(dsm, data, 42, info_frames=1)
    assert di["id3v2_bytes"] == 44 and di["vbr"] == 1 and di["frames_without_reservoir"] == 0
    pcm2, rate = dsm.pcm_decode(data)
    assert rate == 44100 and np.array_equal(pcm, pcm2)


def test_bria_head_mp3_the_clip_of_baseline_config_0(dsm, lib):
    data = open(os.path.join(AUD, "bria_head.mp3"), "rb").read()
    pcm, di = _check_decode(dsm, data, 384, info_frames=0)
    assert di["bitrate_kbps"] == 128 and di["vbr"] == 0
    # a cut in the middle of the stream: the frames whose main data starts before the cut decode to silence, nothing else breaks
    o = 0
    for _ in range(100):
        o += 417 + ((data[o + 2] >> 1) & 1)
    tail, rate, dt = dsm.mp3_decode(data[o:])
    assert dt["frames"] == 284 and dt["huffman_overruns"] == 0 and 0 < dt["frames_without_reservoir"] <= 3
    k = 1152 * (100 + dt["frames_without_reservoir"] + 1)  # one more frame for the filter-bank history
    assert np.allclose(tail[k - 115200:], pcm[k:], atol=1e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/audio"), reason="the reference tree is only present in the build container")
def test_every_mp3_the_reference_ships(dsm, lib):
    facts = json.load(open(os.path.join(AUD, "mp3_facts.json")))
    import hashlib
    for name, f in facts.items():
        data = open(os.path.join("/root/reference/audio", name), "rb").read()
        assert len(data) == f["bytes"] and hashlib.sha256(data).hexdigest() == f["sha256"]
        frames = f.get("frames_from_size", f["probe"]["frames"])
        _check_decode(dsm, data, frames)


def test_not_mp3_is_an_error_not_a_crash(dsm, lib):
    for junk in (b"", b"\x00" * 5000, b"RIFX" + b"\x01" * 100, b"\xff\xfb" * 300, b"\xff\xf3\x90\x00" * 400):  # last: MPEG-2 Layer III
        with pytest.raises(dsm.DsmError):
            dsm.mp3_decode(junk)


# ---- resampler -------------------------------------------------------------------------------------------------------------------
def test_resampler_44100_to_24000(dsm, lib):
    n = 44100
    t = np.arange(n) / 44100.0
    y = dsm.resample(np.ones(n, np.float32), 44100, 24000)
    assert len(y) == -(-n * 80 // 147) and np.allclose(y[200:-200], 1.0, atol=1e-5)           # unit DC gain
    for f in (440.0, 5000.0, 10500.0):                                                          # pass band: frequency and level kept
        y = dsm.resample(np.sin(2 * np.pi * f * t).astype(np.float32), 44100, 24000).astype(np.float64)
        tt = np.arange(len(y)) / 24000.0
        amp = 2 * np.abs(np.mean(y[500:-500] * np.exp(-2j * np.pi * f * tt[500:-500])))
        assert abs(amp - 1) < 2e-3, (f, amp)
    for f in (13000.0, 15000.0, 20000.0):                                                       # above the new Nyquist: gone
        y = dsm.resample(np.sin(2 * np.pi * f * t).astype(np.float32), 44100, 24000).astype(np.float64)
        assert np.sqrt(np.mean(y[500:-500] ** 2)) < 1e-4 * np.sqrt(0.5), f
    assert np.array_equal(dsm.resample(np.arange(10, dtype=np.float32), 24000, 24000), np.arange(10, dtype=np.float32))
    up = dsm.resample(np.sin(2 * np.pi * 1000 * np.arange(16000) / 16000.0).astype(np.float32), 16000, 24000)
    assert len(up) == 24000 and abs(2 * np.abs(np.mean(up[300:-300] * np.exp(-2j * np.pi * 1000 * np.arange(24000)[300:-300] / 24000.0))) - 1) < 2e-3
