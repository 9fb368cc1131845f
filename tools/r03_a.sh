mkdir -p gpurun_out/r03
for g in 1 2 3 4; do
  DSM_TTS_GROUPS=$g timeout -k 10 200 python bench.py --workload tts --batch 32 --steps 40 --warmup 5 > gpurun_out/r03/tts_b32_g$g.json 2> gpurun_out/r03/tts_b32_g$g.err && python -c "
import json,sys
d=json.load(open('gpurun_out/r03/tts_b32_g$g.json'))
print('groups', $g, 'ms_per_step', round(d['ms_per_step'],3))"
done
python -m pytest tests/test_tts_gpu.py tests/test_graphs_gpu.py -x -q -m gpu 2>&1 | tail -3
DSM_TTS_GROUPS=4 python -m pytest tests/test_tts_gpu.py tests/test_graphs_gpu.py -x -q -m gpu 2>&1 | tail -3
