#!/bin/bash
C64="python bench.py --fast-fill --steps 100 --warmup 10 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
ms() { python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4))"; }
for round in 1 2 3 4; do
  echo "default $(bash -c "$C64" 2>/dev/null | ms)   DSM_STREAM_PRIO=1 $(DSM_STREAM_PRIO=1 bash -c "$C64" 2>/dev/null | ms)   DSM_STAGGER=0 $(DSM_STAGGER=0 bash -c "$C64" 2>/dev/null | ms)   B=2048: default $(bash -c "$C64 --batch 2048 --steps 12 --warmup 3" 2>/dev/null | ms) prio1 $(DSM_STREAM_PRIO=1 bash -c "$C64 --batch 2048 --steps 12 --warmup 3" 2>/dev/null | ms)"
done
