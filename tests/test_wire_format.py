"""The msgpack wire format of InMsg / OutMsg (srv/asr.rs:15-34) behind the C ABI, on the CPU:
  * the reference's own known-answer vector for OutMsg::Word (client/rust/kyutai-client/src/stt/protocol.rs:82-97),
  * byte-exact agreement with an independent encoder (the `msgpack` module packing the same struct map; rmp_serde
    writes usize / i64 in their smallest form, f64 as float64, f32 as float32),
  * decoding of everything that encoder emits, field order and unknown fields included (serde semantics)."""
import struct

import msgpack
import numpy as np
import pytest

WORD_KAT = bytes([
    0x83, 0xa4, *b"type", 0xa4, *b"Word", 0xa4, *b"text", 0xa5, *b"hello", 0xaa, *b"start_time",
    0xcb, 0x3f, 0xf8, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00])






def pack_f32_list(xs):
    return (bytes([0x90 | len(xs)]) if len(xs) < 16 else b"\xdc" + struct.pack(">H", len(xs))) + b"".join(
        b"\xca" + struct.pack(">f", x) for x in xs)


def test_reference_known_answer_word(dsm, lib):
    assert dsm.decode_out_msg(WORD_KAT) == {"type": "Word", "text": "hello", "start_time": 1.5}
    assert dsm.encode_out_msg("Word", text="hello", time=1.5) == WORD_KAT


@pytest.mark.parametrize("kind,kw,obj", [
    ("Word", dict(text="héllo wörld", time=12.34), {"type": "Word", "text": "héllo wörld", "start_time": 12.34}),
    ("EndWord", dict(time=0.08), {"type": "EndWord", "stop_time": 0.08}),
    ("Marker", dict(id=7), {"type": "Marker", "id": 7}),
    ("Marker", dict(id=-1234567890123), {"type": "Marker", "id": -1234567890123}),
    ("Marker", dict(id=300), {"type": "Marker", "id": 300}),
    ("Error", dict(text="Server at capacity - no free channels available"),
     {"type": "Error", "message": "Server at capacity - no free channels available"}),
    ("Ready", dict(), {"type": "Ready"}),
])
def test_out_msg_bytes_match_independent_encoder(dsm, lib, kind, kw, obj):
    raw = dsm.encode_out_msg(kind, **kw)
    assert raw == msgpack.packb(obj, use_single_float=False)
    assert dsm.decode_out_msg(raw) == obj


def test_step_message(dsm, lib):
    prs = [0.25, 0.5, 0.125, 0.0625]
    raw = dsm.encode_out_msg("Step", step_idx=70000, prs=prs, buffered_pcm=1920 * 3)
    want = (b"\x84" + msgpack.packb("type") + msgpack.packb("Step") + msgpack.packb("step_idx") + msgpack.packb(70000) +
            msgpack.packb("prs") + pack_f32_list(prs) + msgpack.packb("buffered_pcm") + msgpack.packb(5760))
    assert raw == want
    assert dsm.decode_out_msg(raw) == {"type": "Step", "step_idx": 70000, "prs": prs, "buffered_pcm": 5760}


def test_in_msg_roundtrip_and_bytes(dsm, lib):
    pcm = np.array([0.0, -0.25, 0.5, 1.0], dtype=np.float32)  # protocol.rs:68-78 roundtrip_audio_with_vec_f32
    raw = dsm.encode_in_msg("Audio", pcm=pcm)
    assert raw == b"\x82" + msgpack.packb("type") + msgpack.packb("Audio") + msgpack.packb("pcm") + pack_f32_list(pcm.tolist())
    got = dsm.decode_in_msg(raw)
    assert got["type"] == "Audio" and np.array_equal(got["pcm"], pcm)
    assert dsm.encode_in_msg("Init") == msgpack.packb({"type": "Init"})
    assert dsm.encode_in_msg("Ping") == msgpack.packb({"type": "Ping"})
    assert dsm.encode_in_msg("Marker", id=42) == msgpack.packb({"type": "Marker", "id": 42})
    assert dsm.decode_in_msg(dsm.encode_in_msg("Marker", id=-5)) == {"type": "Marker", "id": -5}
    ogg = dsm.encode_in_msg("OggOpus", data=b"OggS\x00\xff")
    assert ogg == msgpack.packb({"type": "OggOpus", "data": [79, 103, 103, 83, 0, 255]})  # Vec<u8> is a sequence
    assert dsm.decode_in_msg(ogg) == {"type": "OggOpus", "data": b"OggS\x00\xff"}


def test_decoder_follows_serde_semantics(dsm, lib):
    big = np.random.default_rng(0).standard_normal(4000).astype(np.float32)
    # what other clients send: float64 samples (msgpack-python), fields before the tag, unknown fields, bin payloads
    got = dsm.decode_in_msg(msgpack.packb({"pcm": big.tolist(), "type": "Audio", "extra": {"a": [1, 2, None]}}))
    assert got["type"] == "Audio" and np.array_equal(got["pcm"], big)
    assert dsm.decode_in_msg(msgpack.packb({"type": "OggOpus", "data": b"\x01\x02"}))["data"] == b"\x01\x02"
    assert dsm.decode_in_msg(msgpack.packb({"type": "Marker", "id": 2 ** 40})) == {"type": "Marker", "id": 2 ** 40}
    # rejected, like rmp_serde::from_slice would: unknown variant, missing field, wrong field type, truncated, not a map
    for bad in (msgpack.packb({"type": "Nope"}), msgpack.packb({"type": "Marker"}), msgpack.packb({"type": "Marker", "id": "x"}),
                msgpack.packb({"type": "Audio", "pcm": [0.5, 0.25]})[:-3], msgpack.packb(["Audio"]), b""):
        assert dsm.decode_in_msg(bad) is None
    assert dsm.decode_out_msg(msgpack.packb({"type": "Word", "text": "x"})) is None
