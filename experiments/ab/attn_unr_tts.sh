#!/bin/bash
# TTS B = 32 with the bf16-ring attention at 4 (default by shape) / 8 keys per batch, same box
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4), round(j.get('guided_leg',{}).get('ms_per_step',0),4))"; }
for round in 1 2 3; do
  for v in 0 8; do
    echo "DSM_ATTN_UNR=$v tts (plain, guided) $(DSM_ATTN_UNR=$v python bench.py --workload tts --batch 32 --steps 50 --warmup 5 2>/dev/null | ms)"
  done
done
