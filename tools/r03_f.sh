mkdir -p gpurun_out/r03
for m in 0 1; do
python bench.py --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs 400,1024,2048,2304,2560 --dot-mode $m > gpurun_out/r03/bench_mode$m.json 2> gpurun_out/r03/bench_mode$m.err || exit 1
python - <<PY
import json
d=json.load(open("gpurun_out/r03/bench_mode$m.json"))
print("mode $m ms_per_step", d["ms_per_step"], d["step_breakdown_us_single_stream"])
print({k:(round(v.get("ms_per_step",0),2), round(v.get("rtf",0),3)) for k,v in d["capacity"]["legs"].items()})
PY
done
