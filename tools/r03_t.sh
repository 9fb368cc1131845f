mkdir -p gpurun_out/r03
python -m pytest tests/test_tts_gpu.py tests/test_tts_ca_gpu.py tests/test_graphs_gpu.py tests/test_bx3_gpu.py "tests/test_parity_full_gpu.py::test_tts_v202501_shapes" tests/test_large_batch_property_gpu.py -x -q -m gpu -k "tts or TTS" > gpurun_out/r03/tts_tests5.txt 2>&1; rc=$?; tail -6 gpurun_out/r03/tts_tests5.txt; [ $rc = 0 ] || exit 1
for m in 1 0; do
python bench.py --workload tts --batch 32 --steps 50 --warmup 5 --dot-mode $m > gpurun_out/r03/tts_head_m$m.json 2> gpurun_out/r03/tts_head_m$m.err || { tail -3 gpurun_out/r03/tts_head_m$m.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r03/tts_head_m$m.json')); print('TTS B=32 mode $m: %.3f ms/step' % d['ms_per_step'], d['roofline']['frac'])"
done
