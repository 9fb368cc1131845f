mkdir -p gpurun_out/r03
for m in 0 1; do
python bench.py --batch 2048 --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs "" --dot-mode $m --steps 8 --warmup 3 > gpurun_out/r03/bench_b2048_mode$m.json 2> gpurun_out/r03/bench_b2048_mode$m.err || { tail -5 gpurun_out/r03/bench_b2048_mode$m.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r03/bench_b2048_mode$m.json"))
print("mode $m B=2048 ms_per_step %.2f" % d["ms_per_step"], d["step_breakdown_us_single_stream"], "attn frac", round(d["roofline"]["frac"],3), round(d["roofline"]["isolated_single_stream"]["frac"],3))
PY
done
