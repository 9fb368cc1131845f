#!/bin/bash
# Re-sweeps the stream / group knobs of the B = 64 step on the current build (one box, back to back).  A run that fails keeps its
# exit code and the tail of its stderr in the table (VERDICT r03 housekeeping: a sweep that records a bare FAILED cannot tell a
# refused knob from a hang); each run has its own timeout.
B="python bench.py --fast-fill --steps 100 --warmup 10 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
run() {
  name=$1; shift
  err=$(mktemp)
  out=$(env "$@" timeout -k 10 300 bash -c "$B" 2>"$err"); rc=$?
  if [ $rc -ne 0 ]; then
    echo "$name FAILED rc=$rc: $(grep -v amdgpu.ids "$err" | tail -n 3 | tr '\n' ' ' | cut -c1-300)"
  else
    echo "$name $(echo "$out" | python3 -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(j['ms_per_step'],4))")"
  fi
  rm -f "$err"
}
run default X=1
run default2 X=1
run bx3u0 DSM_BX3U=0
run wpack0 DSM_WPACK=0
run stagger2 DSM_STAGGER=2
run stagger0 DSM_STAGGER=0
run groups1 DSM_LM_GROUPS=1
run groups3 DSM_LM_GROUPS=3
run groups4 DSM_LM_GROUPS=4
run prio1 DSM_STREAM_PRIO=1
run attn_nt0 DSM_ATTN_NT=0
run fuseqkv0 DSM_FUSE_QKV=0
