#!/bin/bash
# DSM_BX3U_M64 on / off on one box: stt-2.6b-en B = 128 (two groups of 64 rows) and the guided TTS leg (64 rows)
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4), round(j.get('guided_leg',{}).get('ms_per_step',0),4))"; }
for round in 1 2; do
  for v in 1 0; do
    echo "DSM_BX3U_M64=$v stt-2.6b $(DSM_BX3U_M64=$v python bench.py --config stt-2.6b-en --batch 128 --fast-fill --steps 50 --warmup 5 --no-cpu-baseline --capacity-legs '' --host-path-legs '' --other-configs '' --no-agreement 2>/dev/null | ms)  tts (plain, guided) $(DSM_BX3U_M64=$v python bench.py --workload tts --batch 32 --steps 50 --warmup 5 2>/dev/null | ms)"
  done
done
