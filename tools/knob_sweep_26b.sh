#!/bin/bash
B="python bench.py --config stt-2.6b-en --batch 128 --fast-fill --steps 30 --warmup 5 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4))"; }
for g in 2 1 3 4; do echo "groups $g: $(DSM_LM_GROUPS=$g bash -c "$B" 2>/dev/null | ms)"; done
