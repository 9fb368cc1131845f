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


# ---- Ogg container (RFC 3533 / RFC 7845): the demultiplexer in front of the Opus decoder the host supplies ----
def _ogg_crc(data):
    crc = 0
    for b in data:
        crc ^= b << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if crc & 0x80000000 else (crc << 1) & 0xFFFFFFFF
    return crc


def _ogg_page(serial, seq, segments_payload, htype=0, granule=0):
    """segments_payload: list of (lacing_values, bytes) already laced by the caller"""
    lacing, body = segments_payload
    hdr = b"OggS" + bytes([0, htype]) + granule.to_bytes(8, "little") + serial.to_bytes(4, "little") + seq.to_bytes(4, "little") + b"\0\0\0\0" + bytes([len(lacing)]) + bytes(lacing)
    page = bytearray(hdr + body)
    page[22:26] = _ogg_crc(page).to_bytes(4, "little")
    return bytes(page)


def _lace(packets, continued_tail=False):
    """lacing values + body for whole packets; continued_tail: the last packet goes on in the next page (ends on a 255)"""
    lacing, body = [], b""
    for i, p in enumerate(packets):
        n = len(p)
        last_open = continued_tail and i == len(packets) - 1
        while n >= 255:
            lacing.append(255)
            n -= 255
        if not last_open:
            lacing.append(n)
        else:
            assert n == 0, "an open packet must end on a full segment"
        body += p
    return lacing, body


def _demux_all(lib, chunks):
    import ctypes as C
    lib.dsm_ogg_demux_new.restype = C.c_void_p
    lib.dsm_ogg_demux_free.argtypes = [C.c_void_p]
    lib.dsm_ogg_demux_push.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.dsm_ogg_demux_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    lib.dsm_ogg_demux_info.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    d = lib.dsm_ogg_demux_new()
    out = []
    for ch in chunks:
        assert lib.dsm_ogg_demux_push(d, ch, len(ch)) >= 0
        while True:
            p, n, h = C.c_void_p(), C.c_size_t(), C.c_int()
            if lib.dsm_ogg_demux_next(d, C.byref(p), C.byref(n), C.byref(h)) != 1:
                break
            out.append((C.string_at(p, n.value), bool(h.value)))
    ch_, pre, rate, ok, bad = C.c_int(), C.c_int(), C.c_uint32(), C.c_uint64(), C.c_uint64()
    lib.dsm_ogg_demux_info(d, C.byref(ch_), C.byref(pre), C.byref(rate), C.byref(ok), C.byref(bad))
    lib.dsm_ogg_demux_free(d)
    return out, (ch_.value, pre.value, rate.value, ok.value, bad.value)


def _opus_stream():
    head = b"OpusHead" + bytes([1, 1]) + (312).to_bytes(2, "little") + (48000).to_bytes(4, "little") + b"\0\0" + b"\0"
    tags = b"OpusTags" + (4).to_bytes(4, "little") + b"test" + (0).to_bytes(4, "little")
    rng = np.random.default_rng(4)
    pkts = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (17, 255, 254, 256, 700, 1, 510, 90)]
    big = bytes(rng.integers(0, 256, 255 * 3, dtype=np.uint8))  # 765 bytes: spans two pages below
    pages = [_ogg_page(7, 0, _lace([head]), htype=2), _ogg_page(7, 1, _lace([tags])),
             _ogg_page(7, 2, _lace(pkts[:3])), _ogg_page(7, 3, _lace(pkts[3:5] + [big[:510]], continued_tail=True)),
             _ogg_page(7, 4, _lace([big[510:]] + pkts[5:]), htype=1)]
    return head, tags, pkts, big, pages


def test_ogg_demux_packets_any_split(lib):
    head, tags, pkts, big, pages = _opus_stream()
    stream = b"".join(pages)
    want = [(head, True), (tags, True)] + [(p, False) for p in pkts[:5]] + [(big, False)] + [(p, False) for p in pkts[5:]]
    got, info = _demux_all(lib, [stream])
    assert got == want and info[:3] == (1, 312, 48000) and info[3] == 5 and info[4] == 0
    rng = np.random.default_rng(0)
    for _ in range(20):  # websocket messages cut the byte stream anywhere
        cuts = sorted(rng.integers(0, len(stream), 6).tolist())
        chunks = [stream[a:b] for a, b in zip([0] + cuts, cuts + [len(stream)])]
        assert _demux_all(lib, chunks)[0] == want
    assert _demux_all(lib, [stream[i:i + 1] for i in range(len(stream))])[0] == want  # byte by byte


def test_ogg_demux_damage_and_foreign_streams(lib):
    head, tags, pkts, big, pages = _opus_stream()
    # a flipped payload byte fails the CRC: that page is dropped, and with it the packet it began (the next page's
    # continuation has nothing to continue); everything after resynchronises
    bad3 = bytearray(pages[3]); bad3[40] ^= 0x55
    got, info = _demux_all(lib, [b"".join(pages[:3]) + bytes(bad3) + pages[4]])
    assert [g[0] for g in got] == [head, tags] + pkts[:3] + pkts[5:] and info[4] >= 1
    # garbage (with a stray capture pattern) between pages, and a page of another logical stream
    other = _ogg_page(99, 0, _lace([b"not ours"]))
    noisy = pages[0] + b"\x00\x01OggS\x02junk" + pages[1] + other + b"".join(pages[2:])
    assert [g[0] for g in _demux_all(lib, [noisy])[0]] == [head, tags] + pkts[:5] + [big] + pkts[5:]
    # a chained stream (new beginning-of-stream page): headers are recognised again
    second = [_ogg_page(8, 0, _lace([head]), htype=2), _ogg_page(8, 1, _lace([tags])), _ogg_page(8, 2, _lace([b"abc"]))]
    got, _ = _demux_all(lib, [b"".join(pages) + b"".join(second)])
    assert got[-3:] == [(head, True), (tags, True), (b"abc", False)]
    assert _demux_all(lib, [b"OggS" * 50000])[0] == []  # unbounded garbage: nothing out, nothing hoarded


def test_worker_oggopus_with_a_host_decoder(dsm, lib):
    """InMsg::OggOpus bodies (srv/batched_asr.rs:941-949): demultiplexed per channel, every audio packet handed to the host's
    Opus decoder (here a stand-in that maps packet bytes to samples), the PCM queued like InMsg::Audio.  Without a decoder
    the message is refused."""
    import ctypes as C
    head, tags, pkts, big, pages = _opus_stream()
    calls = []

    @C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_float), C.c_size_t)
    def fake_opus(user, slot, pkt, n, out, cap):
        data = bytes(pkt[:n])
        calls.append((slot, data))
        if data == pkts[1]:
            return -1  # a corrupt packet: reported, skipped, the stream goes on
        k = min(n, cap)
        for i in range(k):
            out[i] = data[i] / 256.0
        return k

    class Eng:  # minimal backend: records what the worker feeds the model
        pass

    be = dsm.WorkerBackend()
    fed = []
    enc = dsm.BE_ENCODE(lambda self, pcm, mask: fed.append((np.ctypeslib.as_array(pcm, (2 * 1920,)).copy(), bytes(mask[:2]))) or 0)
    rst = dsm.BE_RESET(lambda self, slot: 0)
    stp = dsm.BE_STEP(lambda self, mask, text, prs: 0)
    pol = dsm.BE_POLL(lambda self, msgs, cap, toks, tcap: 0)
    be.batch_size, be.asr_delay_in_tokens, be.extra_heads_num = 2, 2, 0
    be.encode_step, be.reset_slot, be.step_tokens, be.poll_msgs = enc, rst, stp, pol
    w = C.c_void_p()
    assert lib.dsm_worker_create_with_backend(C.byref(be), C.byref(w)) == 0
    cid = C.c_uint64()
    slot = lib.dsm_worker_open(w, C.byref(cid))
    body = b"".join(pages)
    msg1, msg2 = dsm.encode_in_msg("OggOpus", data=body[:300]), dsm.encode_in_msg("OggOpus", data=body[300:])
    assert lib.dsm_worker_send(w, slot, msg1, len(msg1)) == -4  # DSM_ERR_STATE: no decoder configured
    lib.dsm_worker_set_opus_decoder.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dsm_worker_set_opus_decoder(w, C.cast(fake_opus, C.c_void_p), None)
    assert lib.dsm_worker_send(w, slot, msg1, len(msg1)) in (0, 1)
    assert lib.dsm_worker_send(w, slot, msg2, len(msg2)) == 1  # one packet failed to decode: reported, rest queued
    audio = [p for p in pkts[:5] + [big] + pkts[5:]]
    assert [c[1] for c in calls] == audio and all(c[0] == slot for c in calls)  # headers never reach the decoder
    n_samples = sum(len(p) for p in audio if p != pkts[1])
    assert lib.dsm_worker_buffered(w, slot) == 0  # still in the channel queue until a step drains it
    assert lib.dsm_worker_step(w) == 1
    assert lib.dsm_worker_buffered(w, slot) == n_samples - 1920  # one frame went to the model, the rest waits
    want = np.concatenate([np.frombuffer(p, np.uint8) / 256.0 for p in audio if p != pkts[1]]).astype(np.float32)
    assert np.array_equal(fed[0][0][slot * 1920:(slot + 1) * 1920], want[:1920]) and fed[0][1][slot] == 1
    lib.dsm_worker_destroy(w)


def test_ogg_mux_pages_are_valid_and_round_trip(lib):
    """TTS output container (srv/tts.rs:188-260): header pages + one page per frame; checked against the Python CRC /
    page layout above and by feeding the bytes back through the demultiplexer."""
    import ctypes as C
    lib.dsm_ogg_mux_new.restype = C.c_void_p
    lib.dsm_ogg_mux_new.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_int]
    lib.dsm_ogg_mux_free.argtypes = [C.c_void_p]
    lib.dsm_ogg_mux_header.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.dsm_ogg_mux_page.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t]
    m = lib.dsm_ogg_mux_new(0xABCD1234, 1, 24000, 312)
    buf = C.create_string_buffer(1 << 17)
    n = lib.dsm_ogg_mux_header(m, buf, len(buf))
    stream = buf.raw[:n]
    assert lib.dsm_ogg_mux_header(m, buf, len(buf)) < 0  # only once
    rng = np.random.default_rng(8)
    frames = []
    for f in range(5):
        pkts = [bytes(rng.integers(0, 256, int(k), dtype=np.uint8)) for k in rng.integers(1, 600, 4)]  # four 20 ms packets per 80 ms frame
        frames.append(pkts)
        arr = (C.c_char_p * 4)(*pkts)
        lens = (C.c_size_t * 4)(*[len(p) for p in pkts])
        assert lib.dsm_ogg_mux_page(m, arr, lens, 4, 3840, 1 if f == 4 else 0, buf, 10) > 10  # too small: size returned, nothing consumed
        n = lib.dsm_ogg_mux_page(m, arr, lens, 4, 3840, 1 if f == 4 else 0, buf, len(buf))
        assert n > 0
        stream += buf.raw[:n]
    assert lib.dsm_ogg_mux_page(m, None, None, 0, 0, 0, buf, len(buf)) < 0  # after the end-of-stream page
    lib.dsm_ogg_mux_free(m)
    # every page: capture pattern, version 0, running sequence numbers, correct CRC, granule = 48 kHz samples so far
    o, seq, granules, flags = 0, 0, [], []
    while o < len(stream):
        assert stream[o:o + 4] == b"OggS" and stream[o + 4] == 0
        nseg = stream[o + 26]
        total = 27 + nseg + sum(stream[o + 27:o + 27 + nseg])
        page = bytearray(stream[o:o + total])
        crc = int.from_bytes(page[22:26], "little")
        page[22:26] = b"\0\0\0\0"
        assert _ogg_crc(page) == crc
        assert int.from_bytes(page[18:22], "little") == seq and int.from_bytes(page[14:18], "little") == 0xABCD1234
        granules.append(int.from_bytes(page[6:14], "little"))
        flags.append(page[5])
        seq += 1
        o += total
    assert granules == [0, 0, 3840, 7680, 11520, 15360, 19200] and flags == [2, 0, 0, 0, 0, 0, 4]
    got, info = _demux_all(lib, [stream[:100], stream[100:]])
    assert [g[1] for g in got[:2]] == [True, True] and got[0][0][:8] == b"OpusHead" and got[1][0][:8] == b"OpusTags"
    assert [g[0] for g in got[2:]] == [p for fr in frames for p in fr] and info[:3] == (1, 312, 24000)
