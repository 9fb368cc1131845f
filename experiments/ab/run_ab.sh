#!/bin/bash
# A/B two builds of the library on ONE box: experiments/ab/libA.so, libB.so (each copied over the package's .so in turn)
L=delayed-streams-modeling_amd/libdsm_mi355x.so
B="python bench.py --fast-fill --steps 100 --warmup 10 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
T="python bench.py --workload tts --batch 32 --steps 50 --warmup 5 --tts-guided-leg 0"
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4))"; }
for round in 1 2; do
  for v in A B; do
    cp experiments/ab/lib$v.so $L
    echo "$v stt $(bash -c "$B" 2>/dev/null | ms)  tts $(bash -c "$T" 2>/dev/null | ms)"
  done
done
