mkdir -p gpurun_out/r03
out=gpurun_out/r03/mode1_groups.txt; : > $out
for g in 1 2 4; do
DSM_LM_GROUPS=$g python bench.py --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs 400,2048 --dot-mode 1 --steps 60 > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - >> $out <<PY
import json
d=json.load(open("/tmp/b.json"))
print("groups $g B=64 ms_per_step %.3f" % d["ms_per_step"], d["step_breakdown_us_single_stream"], {k:round(v.get("ms_per_step",0),2) for k,v in d["capacity"]["legs"].items()})
PY
DSM_LM_GROUPS=$g python bench.py --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs "" --dot-mode 1 --steps 60 --part lm > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - >> $out <<PY
import json
d=json.load(open("/tmp/b.json"))
print("groups $g B=64 LM only ms_per_step %.3f" % d["ms_per_step"])
PY
done
cat $out
