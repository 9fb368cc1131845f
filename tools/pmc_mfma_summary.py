#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from a rocprofv3 --pmc pass (counter_collection CSV):
   util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)   (busy cycles are summed over the SIMDs,
   GRBM_GUI_ACTIVE over the 8 XCDs: MI355X_MICROARCH.md, DVFS give-back).  Also wave-cycle shares where collected."""
import collections, csv, glob, json, os, sys
f = max(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("void ", "").replace("(GemmArgs)", "").replace("unsigned short", "bf16").split("(")[0][:60]
    key = (name, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[key] += 1
rows = []
for k, c in agg.items():
    n = max(cnt[k], 1)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    row = {"kernel": k[0], "grid": k[1], "dispatches": n, "gui_active_per_dispatch_xcd_cycles": gui / n / 8,
           "mfma_util": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8 * 1024)}
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if nm in c:
                row[nm + "_share"] = c[nm] / wc
    if "SQ_LDS_BANK_CONFLICT" in c:
        row["lds_bank_conflict_cycles_per_dispatch"] = c["SQ_LDS_BANK_CONFLICT"] / n
    rows.append(row)
rows.sort(key=lambda r: -r["gui_active_per_dispatch_xcd_cycles"] * r["dispatches"])
print(json.dumps({"source": os.path.relpath(f), "kernels": rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]}, indent=1))
