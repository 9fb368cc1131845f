#!/bin/bash
# static 48 KB LDS / 136 VGPRs allocated (libA) against dynamic LDS (libB) with and without the late half of the weight requests
L=delayed-streams-modeling_amd/libdsm_mi355x.so
C64="python bench.py --fast-fill --steps 100 --warmup 10 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4))"; }
for round in 1 2 3; do
  cp experiments/ab/libA.so $L
  echo "A (static LDS)            step $(bash -c "$C64" 2>/dev/null | ms)  lm $(bash -c "$C64 --part lm" 2>/dev/null | ms)  tts $(python bench.py --workload tts --batch 32 --steps 50 --warmup 5 --tts-guided-leg 0 2>/dev/null | ms)"
  cp experiments/ab/libB.so $L
  echo "B (dynamic LDS, late 4)   step $(bash -c "$C64" 2>/dev/null | ms)  lm $(bash -c "$C64 --part lm" 2>/dev/null | ms)  tts $(python bench.py --workload tts --batch 32 --steps 50 --warmup 5 --tts-guided-leg 0 2>/dev/null | ms)"
  echo "B (dynamic LDS, late 0)   step $(DSM_BX3U_LATE=0 bash -c "$C64" 2>/dev/null | ms)  lm $(DSM_BX3U_LATE=0 bash -c "$C64 --part lm" 2>/dev/null | ms)  tts $(DSM_BX3U_LATE=0 python bench.py --workload tts --batch 32 --steps 50 --warmup 5 --tts-guided-leg 0 2>/dev/null | ms)"
done
