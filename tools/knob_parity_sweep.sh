#!/bin/bash
# The parity tests under every fallback knob: each alternative path must stay bit-exact against the oracle too.
# usage (on the GPU box): tools/knob_parity_sweep.sh > gpurun_out/knob_parity.txt
T="tests/test_parity_gpu.py tests/test_bx3_gpu.py tests/test_tts_gpu.py tests/test_tts_ca_gpu.py tests/test_graphs_gpu.py"
ONLY=${ONLY:-}
run() {
  name=$1; shift
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
  log=$(mktemp)
  env "$@" timeout -k 10 900 python -m pytest $T -x -q -m gpu > "$log" 2>&1; rc=$?
  echo "$name rc=$rc: $(tail -n 1 "$log")"
  if [ $rc -ne 0 ]; then tail -n 30 "$log"; fi
  rm -f "$log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
run graphs0 DSM_GRAPHS=0
run groups1 DSM_LM_GROUPS=1
run fuseqkv0 DSM_FUSE_QKV=0
run wpack0 DSM_WPACK=0
run bx3u0 DSM_BX3U=0
run small0_m64_0 DSM_ATTN_SMALL=0 DSM_BX3U_M64=0
run wkgate0 DSM_WK_GATE_CHUNKS=0
run late0 DSM_BX3U_LATE=0
run attn_unr4 DSM_ATTN_UNR=4
run attn_unr8 DSM_ATTN_UNR=8
