#!/bin/bash
# libA against libB on one box: STT step, LM alone, encode alone (B = 64), TTS B = 32
L=delayed-streams-modeling_amd/libdsm_mi355x.so
C64="python bench.py --fast-fill --steps 100 --warmup 10 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4))"; }
for round in 1 2 3; do
  for v in A B; do
    cp experiments/ab/lib$v.so $L
    echo "$v  step $(bash -c "$C64" 2>/dev/null | ms)  lm $(bash -c "$C64 --part lm" 2>/dev/null | ms)  enc $(bash -c "$C64 --part enc" 2>/dev/null | ms)  tts $(python bench.py --workload tts --batch 32 --steps 50 --warmup 5 --tts-guided-leg 0 2>/dev/null | ms)"
  done
done
cp experiments/ab/libB.so $L
