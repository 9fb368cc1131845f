#!/bin/bash
# DSM_ATTN_SMALL on / off on one box: TTS B = 32 (plain and guided legs)
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4), round(j.get('guided_leg',{}).get('ms_per_step',0),4))"; }
for round in 1 2; do
  for v in 1 0; do
    echo "DSM_ATTN_SMALL=$v tts (plain, guided) $(DSM_ATTN_SMALL=$v python bench.py --workload tts --batch 32 --steps 50 --warmup 5 2>/dev/null | ms)"
  done
done
