"""CPU sanitizer recipe (AddressSanitizer + UBSan; the GPU pool has none):
  * tests/sanitize: host-only build of the parsers of untrusted bytes — msgpack + worker (csrc/dsm_worker.inc), RIFF/WAVE
    (csrc/dsm_audio.inc), MPEG-1 Layer III + resampler (csrc/dsm_mp3.inc), safetensors headers (csrc/dsm_safetensors.h) — driven with truncated, oversized, deeply nested
    and bit-flipped inputs (host_fuzz.cpp), including the 3 M-level nesting that used to overflow the stack;
  * `make -C oracle asan`: the CPU oracle itself, streamed through encode / LM / decode / TTS on the tiny configs."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")


def test_untrusted_input_parsers_under_asan_ubsan(tmp_path):
    d = os.path.join(ROOT, "tests", "sanitize")
    subprocess.check_call(["make", "-s", "-C", d, "host_fuzz"])
    r = subprocess.run([os.path.join(d, "host_fuzz"), str(tmp_path), os.path.join(ROOT, "tests", "golden", "audio")], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "host_fuzz ok" in r.stdout


def test_oracle_under_asan_ubsan(dsm, tiny_weights, tmp_path):
    from dsm_amd import synth
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    cfg, tcfg = dsm.config_tiny(), dsm.config_tts_tiny()
    tpath = synth.make_synth_tts_weights(tcfg, os.path.dirname(tiny_weights[0]), tag="tts_tiny")
    a, t = tmp_path / "asr_cfg.bin", tmp_path / "tts_cfg.bin"
    a.write_bytes(bytes(cfg))
    t.write_bytes(bytes(tcfg))
    r = subprocess.run([os.path.join(ROOT, "oracle", "_asan", "asan_driver"), str(a), tiny_weights[0], tiny_weights[1], str(t),
                        tpath, "16"], capture_output=True, text=True, timeout=600, env=ENV)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "oracle asan ok" in r.stdout


def test_deeply_nested_msgpack_is_refused_by_the_product_library(dsm, lib):
    """`decode_in` skipped unknown fields recursively; 3 M nested arrays overflowed the stack and took
    the worker down.  rmp_serde stops at depth 1024 and recv_loop logs and carries on (srv/batched_asr.rs:927-951)."""
    for nest in (b"\x91", b"\x81\x00"):
        assert dsm.decode_in_msg(b"\x82\xa4type\xa4Ping\xa1x" + nest * 3_000_000 + b"\x00") is None
    # an unknown field nested within the limit is ignored, like serde does
    assert dsm.decode_in_msg(b"\x82\xa4type\xa4Ping\xa1x" + b"\x91" * 600 + b"\xc0") == {"type": "Ping"}
    assert dsm.decode_in_msg(b"\x81" + b"\xc0" * ((64 << 20) + 8)) is None  # beyond the websocket message limit
