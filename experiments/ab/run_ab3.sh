#!/bin/bash
# libA against libB on one box: STT step, LM alone (B = 64), step at B = 2048, stt-2.6b B = 128
L=delayed-streams-modeling_amd/libdsm_mi355x.so
C64="python bench.py --fast-fill --steps 100 --warmup 10 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4), round(j['roofline'].get('isolated_single_stream',{}).get('avg_launch_us',0),2))"; }
for round in 1 2 3; do
  for v in A B; do
    cp experiments/ab/lib$v.so $L
    echo "$v  step (ms, attention alone us) $(bash -c "$C64" 2>/dev/null | ms)  lm $(bash -c "$C64 --part lm" 2>/dev/null | ms)  B=2048 $(bash -c "$C64 --batch 2048 --steps 12 --warmup 3" 2>/dev/null | ms)  2.6b $(python bench.py --config stt-2.6b-en --batch 128 --fast-fill --steps 50 --warmup 5 --no-cpu-baseline --capacity-legs '' --host-path-legs '' --other-configs '' --no-agreement 2>/dev/null | ms)"
  done
done
