mkdir -p gpurun_out/r03
python -m pytest tests -x -q -m gpu > gpurun_out/r03/gpu_tests_1.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03/gpu_tests_1.txt
timeout -k 10 400 python bench.py > gpurun_out/r03/bench_1.json 2> gpurun_out/r03/bench_1.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03/bench_1.json'))
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
print(json.dumps(d.get('step_breakdown_us_single_stream'))[:600])
print(json.dumps(d.get('capacity'))[:900])
PY
