#!/usr/bin/env python3
"""Device-side timeline of one step of the two-stream pipeline (encoder stream + LM group streams), from in-kernel
wall-clock brackets (dsm_prof_timeline): which launches really run beside which.  rocprofv3's tracing serialises the
queues, so it cannot show this.   python tools/timeline.py [B] [steps]  (DSM_STAGGER / DSM_LM_GROUPS as usual)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, dsm_amd
from dsm_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = dsm_amd.config_stt_1b_en_fr()
lm, mimi = synth.make_synth_weights(cfg, os.environ.get("DSM_WEIGHTS_DIR", "/tmp/dsm_weights"), tag="stt-1b-en_fr")
eng = dsm_amd.AsrEngine(cfg, B, lm, mimi)
dev = torch.device("cuda", 0)
pcm = torch.from_numpy(synth.synth_pcm(min(B, 64), 4)).to(dev).repeat(1, (B + 63) // 64, 1)[:, :B].contiguous()
mask = torch.ones(B, dtype=torch.uint8, device=dev)
text = torch.zeros(B, dtype=torch.int32, device=dev); prs = torch.zeros(4 * B, device=dev); codes = torch.zeros(B * 32, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
eng.debug_set_positions(3000, 1000)
def step(i):
    eng.encode_step_dev(pcm[i % 4].data_ptr(), mask.data_ptr(), codes.data_ptr())
    eng.step_tokens_dev(None, mask.data_ptr(), text.data_ptr(), prs.data_ptr())
for i in range(6): step(i)
torch.cuda.synchronize()
eng.prof_enable(dsm_amd.PROF_TAGS)
eng.prof_timeline(True)
eng.prof_read(); eng.prof_timeline_read()
for i in range(3): step(i)          # settle in eager mode (profiling forces eager launches)
torch.cuda.synchronize()
eng.prof_timeline_read()
for i in range(steps): step(i)
torch.cuda.synchronize()
recs = sorted(eng.prof_timeline_read(), key=lambda r: r[3])
eng.prof_enable([])
kinds = {0: "attn", 1: "gemm", 2: "reduce"}
span = max(r[4] for r in recs) - min(r[3] for r in recs)
print(f"B={B} steps={steps} groups={eng.stream_groups()} stagger={os.environ.get('DSM_STAGGER', '1')}: {len(recs)} bracketed launches over {span:.0f} us "
      f"({span / steps:.0f} us/step; profiling brackets force eager launches)")
sids = sorted({r[0] for r in recs})
busy = {s: sum(r[4] - r[3] for r in recs if r[0] == s) for s in sids}
print("stream busy us/step:", {s: round(busy[s] / steps, 1) for s in sids}, "(0 = encoder, 1 + g = LM group g)")
# pairwise concurrency: for every LM attention launch, what ran beside it on the other streams
def overlap(a, b):
    return max(0.0, min(a[4], b[4]) - max(a[3], b[3]))
tot = {}
for r in recs:
    if r[1] != 0 or r[2] != "attn_lm":
        continue
    d = r[4] - r[3]
    for o in recs:
        if o[0] == r[0]:
            continue
        ov = overlap(r, o)
        if ov > 0:
            key = (kinds[o[1]] + ":" + o[2])
            tot[key] = tot.get(key, 0.0) + ov
    tot["_attn_total"] = tot.get("_attn_total", 0.0) + d
print("while an LM attention launch ran, the other streams ran (us/step):", {k: round(v / steps, 1) for k, v in sorted(tot.items())})
n_attn = sum(1 for r in recs if r[1] == 0 and r[2] == "attn_lm")
print("LM attention avg launch us:", round(tot.get("_attn_total", 0) / max(n_attn, 1), 1))
# the first 120 launches as a table
print("%-4s %-7s %-10s %10s %10s %8s" % ("sid", "kind", "class", "start_us", "end_us", "dur_us"))
for r in recs[:160]:
    print("%-4d %-7s %-10s %10.1f %10.1f %8.1f" % (r[0], kinds[r[1]], r[2], r[3], r[4], r[4] - r[3]))
eng.close()
