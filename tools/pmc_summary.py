#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per run, no tracing
domains).  Usage: pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <heads> > pmc_hbm_traffic.json
Units are the counters' own (KB).  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts 64 B per 128-B request on
gfx950 for wide coalesced reads: bench.py doubles it before comparing with a byte count."""
import collections, csv, glob, json, os, sys


def load(d, counter):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        key = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]), int(r["Workgroup_Size"]))
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
    return agg


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --fast-fill --no-overlap --steps 3. Units: KB as "
       "reported. FETCH_SIZE counts 64 B per 128-B request on gfx950 for wide coalesced reads (MI355X_MICROARCH.md): double "
       "it before comparing with a byte count; WRITE_SIZE is exact for 16-B/lane stores.", "kernels": []}
for key, (n, tot) in sorted(fetch.items(), key=lambda kv: -kv[1][1])[:24]:
    name, grid, wg = key
    e = {"kernel": f"{name} grid={grid} wg={wg}", "dispatches": n, "FETCH_SIZE_KB_per_dispatch": round(tot / n, 1),
         "WRITE_SIZE_KB_per_dispatch": round(write[key][1] / max(write[key][0], 1), 1) if key in write else None}
    if name.startswith("void attn_kernel<unsigned short, 128, 1"):  # the LM attention (a fourth template argument since r04: keys per batch)
        e["slots_per_dispatch"] = grid // wg // heads  # one workgroup per (slot, head)
    out["kernels"].append(e)
json.dump(out, sys.stdout, indent=1)
