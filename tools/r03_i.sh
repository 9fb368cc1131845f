mkdir -p gpurun_out/r03
python -m pytest tests/test_bx3_gpu.py tests/test_tts_ca_gpu.py -x -q -m gpu > gpurun_out/r03/bx3_tests2.txt 2>&1; rc=$?; tail -15 gpurun_out/r03/bx3_tests2.txt; [ $rc = 0 ] || exit 1
for g in 1 2; do
DSM_LM_GROUPS=$g python bench.py --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs 400,2048,2304 --dot-mode 1 --steps 60 > gpurun_out/r03/bench_mode1_pre_g$g.json 2> gpurun_out/r03/bench_mode1_pre_g$g.err || { tail -5 gpurun_out/r03/bench_mode1_pre_g$g.err; exit 1; }
python - <<PY
import json
d=json.load(open("gpurun_out/r03/bench_mode1_pre_g$g.json"))
print("groups $g B=64 ms_per_step %.3f" % d["ms_per_step"], d["step_breakdown_us_single_stream"], {k:round(v.get("ms_per_step",0),2) for k,v in d["capacity"]["legs"].items()})
PY
done
