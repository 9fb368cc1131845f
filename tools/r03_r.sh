mkdir -p gpurun_out/r03
out=gpurun_out/r03/tts_groups.txt; : > $out
for B in 64 128; do for g in 1 2; do
DSM_TTS_GROUPS=$g timeout -k 10 200 python bench.py --workload tts --batch $B --steps 30 --warmup 3 > /tmp/t.json 2> /tmp/t.err || { tail -3 /tmp/t.err; exit 1; }
python - >> $out <<PY
import json
d=json.load(open("/tmp/t.json")); print("TTS B=$B groups $g: %.2f ms/step, %.0f x realtime" % (d["ms_per_step"], d["value"]))
PY
done; done
cat $out
