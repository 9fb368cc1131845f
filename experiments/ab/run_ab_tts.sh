#!/bin/bash
# A/B two builds of the library on ONE box, TTS plain and guided legs: experiments/ab/libA.so, libB.so
L=delayed-streams-modeling_amd/libdsm_mi355x.so
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4), round(j.get('guided_leg',{}).get('ms_per_step',0),4))"; }
for round in 1 2; do
  for v in A B; do
    cp experiments/ab/lib$v.so $L
    echo "$v tts (plain, guided) $(python bench.py --workload tts --batch 32 --steps 50 --warmup 5 2>/dev/null | ms)"
  done
done
