#!/usr/bin/env python3
"""Builds the speech fixtures of tests/golden/ from the reference's own sample `audio/tts-input-sample-01.wav`
(48 kHz stereo s16, 6.33 s; BASELINE.json configs[0] / SURVEY.md §8(d) config 1).  Data only:
  speech_48k_stereo_head.wav  first 0.25 s, re-wrapped in a minimal RIFF header (wav-decode / resampler tests)
  speech_24k_mono.s16         the whole clip, channel 0, 48 kHz -> 24 kHz with the clients' two-point linear resampler
                              (numpy restatement below), rounded to s16: the B = 1 engine-vs-oracle input.
Run in the build container (the GPU box has no /root/reference)."""
import os
import struct
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/audio/tts-input-sample-01.wav"


def linear_resample(x, in_rate, out_rate):
    """kyutai-client-core/src/audio.rs:141-183 LinearResampler::process_into on one buffer."""
    step, pos, out = in_rate / out_rate, 0.0, []
    x = np.asarray(x, dtype=np.float32)
    while pos + 1.0 < x.size:
        i = int(np.floor(pos))
        a, b = x[i], x[i + 1]
        out.append(np.float32(a + np.float32(b - a) * np.float32(pos - i)))
        pos += step
    return np.asarray(out, dtype=np.float32)


if __name__ == "__main__":
    w = wave.open(SRC)
    ch, rate, n = w.getnchannels(), w.getframerate(), w.getnframes()
    raw = w.readframes(n)
    head = raw[: 12000 * ch * 2]
    hdr = (b"RIFF" + struct.pack("<I", 36 + len(head)) + b"WAVEfmt " +
           struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(head)))
    gold = os.path.join(ROOT, "tests", "golden")
    open(os.path.join(gold, "speech_48k_stereo_head.wav"), "wb").write(hdr + head)
    pcm = np.frombuffer(raw, dtype="<i2").reshape(-1, ch)[:, 0].astype(np.float32) / np.float32(32768)
    out = linear_resample(pcm, rate, 24000)
    s16 = np.clip(np.rint(out * 32768.0), -32768, 32767).astype("<i2")
    s16.tofile(os.path.join(gold, "speech_24k_mono.s16"))
    print(len(head) + 44, "bytes head;", s16.size, "samples at 24 kHz =", s16.size / 24000, "s")
