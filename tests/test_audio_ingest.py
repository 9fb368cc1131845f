"""Host-side audio ingest behind the C ABI (no GPU): dsm_wav_decode against Python's `wave`, and the streaming
linear resampler against a numpy restatement of the clients' `LinearResampler`
(client/rust/kyutai-client-core/src/audio.rs:133-183; the reference holds no test vectors for it)."""
import os
import struct
import wave

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HEAD = os.path.join(HERE, "golden", "speech_48k_stereo_head.wav")


def ref_resample(x, in_rate, out_rate):
    step, pos, out = in_rate / out_rate, 0.0, []
    while pos + 1.0 < x.size:
        i = int(np.floor(pos))
        a, b = x[i], x[i + 1]
        out.append(np.float32(a + np.float32(b - a) * np.float32(pos - i)))
        pos += step
    return np.asarray(out, dtype=np.float32)


def test_wav_decode_takes_channel_0(dsm, lib):
    data = open(HEAD, "rb").read()
    pcm, rate = dsm.wav_decode(data)
    w = wave.open(HEAD)
    raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, w.getnchannels())
    assert rate == w.getframerate() == 48000
    assert np.array_equal(pcm, raw[:, 0].astype(np.float32) / np.float32(32768))  # symphonia's s16 -> f32 scaling


@pytest.mark.parametrize("fmt,bits,dtype", [(1, 8, "u1"), (1, 16, "<i2"), (1, 32, "<i4"), (3, 32, "<f4")])
def test_wav_decode_sample_formats(dsm, lib, fmt, bits, dtype):
    rng = np.random.default_rng(bits + fmt)
    n, ch = 100, 3
    if dtype == "<f4":
        x = rng.standard_normal((n, ch)).astype(dtype)
        want = x[:, 0]
    else:
        info = np.iinfo(np.dtype(dtype))
        x = rng.integers(info.min, info.max, (n, ch), endpoint=True).astype(dtype)
        want = ((x[:, 0].astype(np.float64) - (128 if bits == 8 else 0)) / float(1 << (bits - 1))).astype(np.float32)
    body = x.tobytes()
    hdr = (b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"LIST" + struct.pack("<I", 4) + b"abcd" + b"fmt " +
           struct.pack("<IHHIIHH", 16, fmt, ch, 16000, 16000 * ch * bits // 8, ch * bits // 8, bits) + b"data" +
           struct.pack("<I", len(body)))
    pcm, rate = dsm.wav_decode(hdr + body)
    assert rate == 16000 and np.array_equal(pcm, want)


def test_wav_decode_rejects_garbage(dsm, lib):
    with pytest.raises(dsm.DsmError):
        dsm.wav_decode(b"OggS" + b"\0" * 64)
    with pytest.raises(dsm.DsmError):
        dsm.wav_decode(b"RIFF\0\0\0\0WAVE")  # no fmt / data chunk


@pytest.mark.parametrize("rates", [(48000, 24000), (44100, 24000), (16000, 24000)])
def test_linear_resampler_matches_restatement(dsm, lib, rates):
    pcm, _ = dsm.wav_decode(open(HEAD, "rb").read())
    pcm = pcm[:6000]
    got = dsm.LinearResampler(*rates).process(pcm)
    want = ref_resample(pcm, *rates)
    assert got.size == want.size and np.array_equal(got.view(np.uint32), want.view(np.uint32))


class RefStreaming:
    """LinearResampler with its carried state (audio.rs:133-183), restated call for call."""

    def __init__(self, in_rate, out_rate):
        self.step, self.pos, self.buf = in_rate / out_rate, 0.0, np.zeros(0, np.float32)

    def process(self, x):
        if x.size == 0:
            return np.zeros(0, np.float32)
        self.buf = np.concatenate([self.buf, x])
        out = []
        while self.pos + 1.0 < self.buf.size:
            i = int(np.floor(self.pos))
            a, b = self.buf[i], self.buf[i + 1]
            out.append(np.float32(a + np.float32(b - a) * np.float32(self.pos - i)))
            self.pos += self.step
        drain = int(np.floor(self.pos))
        if drain > 0:
            self.buf = self.buf[drain:]
            self.pos -= drain
        return np.asarray(out, dtype=np.float32)


@pytest.mark.parametrize("rates", [(48000, 24000), (44100, 24000)])
def test_linear_resampler_is_streaming(dsm, lib, rates):
    """Ragged chunks (down to single samples and empty buffers): the fractional position and the undrained tail are
    carried across calls (audio.rs:158-181).  Chunk for chunk identical to the restatement; identical to the one-shot
    result when the step is exact in binary (48 -> 24 kHz), within rounding of the carried f64 position otherwise."""
    rng = np.random.default_rng(1)
    x = rng.standard_normal(20000).astype(np.float32)
    one = dsm.LinearResampler(*rates).process(x)
    r, ref, parts, i = dsm.LinearResampler(*rates), RefStreaming(*rates), [], 0
    while i < x.size:
        n = int(rng.integers(0, 700))
        got = r.process(x[i:i + n])
        assert np.array_equal(got.view(np.uint32), ref.process(x[i:i + n]).view(np.uint32))
        parts.append(got)
        i += n
    cat = np.concatenate(parts)
    assert abs(one.size - x.size * rates[1] / rates[0]) <= 2 and abs(cat.size - one.size) <= 1
    m = min(cat.size, one.size)
    if rates == (48000, 24000):
        assert np.array_equal(cat, one)
    else:
        assert np.allclose(cat[:m], one[:m], atol=1e-4)
