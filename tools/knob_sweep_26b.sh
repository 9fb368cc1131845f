#!/bin/bash
# stt-2.6b-en at B = 128 (BASELINE.json configs[2]) over the number of LM stream groups, one box, back to back; failures keep
# their exit code and stderr tail.
B="python bench.py --config stt-2.6b-en --batch 128 --fast-fill --steps 30 --warmup 5 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
for g in 2 1 3 4; do
  err=$(mktemp)
  out=$(DSM_LM_GROUPS=$g timeout -k 10 400 bash -c "$B" 2>"$err"); rc=$?
  if [ $rc -ne 0 ]; then echo "groups $g: FAILED rc=$rc: $(grep -v amdgpu.ids "$err" | tail -n 3 | tr '\n' ' ' | cut -c1-300)"
  else echo "groups $g: $(echo "$out" | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4))")"; fi
  rm -f "$err"
done
