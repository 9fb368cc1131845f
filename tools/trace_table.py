"""Per (kernel, grid) average durations from a rocprofv3 --kernel-trace CSV: python tools/trace_table.py <dir> [substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
rows = collections.OrderedDict()
import os
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
newest = max(os.path.getmtime(x) for x in files)  # gpurun_out/ keeps earlier visits' files: only the newest visit's (ten minutes)
for f in (x for x in files if newest - os.path.getmtime(x) < 600):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if sub and sub not in n:
            continue
        key = (n[:90], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
        rows.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for (n, gx, gy, wg), v in rows.items():
    v2 = sorted(v)[len(v) // 10: len(v) - len(v) // 10] or v
    print("%8.2f us (min %7.2f, n=%4d)  wgs %5d x %-3d  %s" % (sum(v2) / len(v2), min(v), len(v), int(gx) // int(wg), int(gy), n))
