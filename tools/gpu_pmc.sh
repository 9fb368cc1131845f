#!/bin/bash
# PMC passes over a short steady-state bench (each counter group in its own run, no tracing domains).
set -o pipefail
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
R=$PWD
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc/$name -- python3 bench.py --fast-fill --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/$name.json 2> gpurun_out/pmc/$name.err; echo "$name rc=$?"
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
