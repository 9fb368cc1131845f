mkdir -p gpurun_out/r03
out=gpurun_out/r03/groups_threshold.txt; : > $out
for g in 1 2; do
DSM_LM_GROUPS=$g timeout -k 10 300 python bench.py --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs "32,48,96,128,192,256" --steps 100 > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - >> $out <<PY
import json
d=json.load(open("/tmp/b.json"))
print("groups $g: B=64 %.3f" % d["ms_per_step"], {k:round(v.get("ms_per_step",0),3) for k,v in d["capacity"]["legs"].items()})
PY
DSM_LM_GROUPS=$g timeout -k 10 300 python bench.py --config stt-2.6b-en --batch 128 --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs "" --steps 50 > /tmp/b.json 2> /tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - >> $out <<PY
import json
d=json.load(open("/tmp/b.json"))
print("groups $g: stt-2.6b B=128 %.3f" % d["ms_per_step"])
PY
done
cat $out
