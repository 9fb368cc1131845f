#!/usr/bin/env python3
"""Makes the mp3 fixtures of tests/test_mp3.py from the reference's own audio samples (run in the build container, where
/root/reference exists; the GPU box only sees the committed files):
  tests/golden/audio/loona.mp3          the whole 9 KB sample (ID3v2.4 tag + Info frame + 42 audio frames, 56 kbps)
  tests/golden/audio/bria_head.mp3      the first 384 frames (10 s) of audio/bria.mp3 — BASELINE.json configs[0] names this clip —
                                        cut at a frame boundary
  tests/golden/audio/mp3_facts.json     per FULL reference file: byte size, sha256, and what follows from the container alone (frame
                                        count from the frame walk of this repo's parser AND from size / frame length where the
                                        stream is CBR), so that the full-file test can run wherever the files are present
Data files, not source: inputs for the decoder."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsm_amd
SRC = "/root/reference/audio"
OUT = os.path.join(ROOT, "tests", "golden", "audio")
os.makedirs(OUT, exist_ok=True)
facts = {}
for name in sorted(os.listdir(SRC)):
    if not name.endswith(".mp3"):
        continue
    data = open(os.path.join(SRC, name), "rb").read()
    info = dsm_amd.mp3_probe(data)
    f = {"bytes": len(data), "sha256": hashlib.sha256(data).hexdigest(), "probe": info}
    if not info["vbr"]:  # CBR: frames follow from the size alone (417 / 418 bytes at 128 kbps, 44.1 kHz)
        f["frames_from_size"] = round((len(data) - info["id3v2_bytes"]) / (144000 * info["bitrate_kbps"] / info["sample_rate"]))
    facts[name] = f
open(os.path.join(OUT, "loona.mp3"), "wb").write(open(os.path.join(SRC, "loona.mp3"), "rb").read())
b = open(os.path.join(SRC, "bria.mp3"), "rb").read()
o, n = 0, 0
while n < 384:  # CBR 128 kbps: 144000 * 128 / 44100 = 417 (+ padding bit)
    assert b[o] == 0xFF and (b[o + 1] & 0xFE) == 0xFA, o
    o += 417 + ((b[o + 2] >> 1) & 1)
    n += 1
open(os.path.join(OUT, "bria_head.mp3"), "wb").write(b[:o])
json.dump(facts, open(os.path.join(OUT, "mp3_facts.json"), "w"), indent=1)
print(json.dumps(facts, indent=1))
