#!/usr/bin/env python3
"""Launch sequence of ONE step from a rocprofv3 kernel-trace CSV: every kernel between two launches of the marker kernel,
in start order, with its duration and the gap to the previous kernel's end.  usage: trace_sequence.py DIR MARKER [step]"""
import csv, glob, os, sys
files = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")
newest = max(os.path.getmtime(x) for x in files)  # the newest visit's bench process (gpurun_out/ keeps earlier visits' files)
f = max((x for x in files if newest - os.path.getmtime(x) < 600), key=os.path.getsize)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2]
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else -3
a, b = idx[k], idx[k + 1]
prev_end = None
tot = 0
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("void ", "").replace("(GemmArgs)", "").replace("unsigned short", "bf16").split("(")[0][:56]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Workgroup_Size_X"])
    gap = (s - prev_end) / 1000 if prev_end else 0.0
    print(f"{n:58s} grid={int(r['Grid_Size_X'])//wg:6d}x{int(r['Grid_Size_Y']):3d}x{int(r['Grid_Size_Z']):5d} wg={wg:4d} dur={(e-s)/1000:7.1f}us gap={gap:6.1f}us")
    prev_end = max(prev_end or 0, e)
    tot += e - s
print(f"{b-a} launches, {tot/1000:.1f} us of kernel time, span {(int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/1000:.1f} us")
