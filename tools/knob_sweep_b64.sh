#!/bin/bash
B="python bench.py --fast-fill --steps 100 --warmup 10 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
run() { name=$1; shift; r=$(env "$@" bash -c "$B" 2>/dev/null | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4))"); echo "$name $r"; }
run default X=1
run default2 X=1
run bx3u0 DSM_BX3U=0
run stagger2 DSM_STAGGER=2
run stagger0 DSM_STAGGER=0
run groups1 DSM_LM_GROUPS=1
run groups3 DSM_LM_GROUPS=3
run groups4 DSM_LM_GROUPS=4
run prio1 DSM_STREAM_PRIO=1
run attn_nt0 DSM_ATTN_NT=0
run fuseqkv0 DSM_FUSE_QKV=0
