#!/usr/bin/env python3
"""Per-(kernel, grid) summary of a rocprofv3 kernel-trace CSV; also prints the inter-kernel idle time."""
import collections, csv, glob, os, sys
# gpurun_out/ accumulates across visits to the GPU box (results are merged back, never deleted): take the NEWEST visit's files
# (written within ten minutes of the newest one) and, among those, the bench process (the largest: child processes are traced too)
files = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")
newest = max(os.path.getmtime(x) for x in files)
f = max((x for x in files if newest - os.path.getmtime(x) < 600), key=os.path.getsize)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("(GemmArgs)", "").replace("unsigned short", "bf16").split("(")[0][:48]
    wg = int(r["Workgroup_Size_X"])
    key = (n, int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), wg)
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{k[0]:50s} grid={k[1]:5d}x{k[2]:4d}x{k[3]:4d} wg={k[4]:4d} n={len(v):5d} avg={sum(v)/len(v):7.1f}us share={sum(v)/tot*100:5.1f}%")
# idle gaps inside the last 10 steps: one conv_state_shift launch per Mimi encode = per step (lm_input runs once per stream
# GROUP, i.e. twice per step by default — r02's summary counted half steps)
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("conv_state_shift_kernel")]
if len(idx) <= 12:
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("lm_input_kernel")]
if len(idx) > 12:
    a, b = idx[-11], idx[-1]
    span = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b])
    print(f"last 10 steps: span {span/1e4:.1f} us/step, kernel busy {busy/1e4:.1f} us/step, idle {100*(1-busy/span):.1f}% ({b-a} launches = {(b-a)/10:.0f}/step)")
