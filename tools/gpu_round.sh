#!/bin/bash
# One GPU-box visit: default bench (as the driver runs it), rocprofv3 kernel trace + PMC passes.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
# kernel trace of the same command line shape (steady-state ring positions, no CPU leg)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -- python3 bench.py --fast-fill --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/prof/trace_bench.json 2> gpurun_out/prof/trace.err; echo "trace rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace_1stream -- python3 bench.py --fast-fill --no-overlap --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/prof/trace_1stream_bench.json 2> gpurun_out/prof/trace_1stream.err; echo "trace1 rc=$?"
# HBM traffic counters, each in its own pass (no tracing domains)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/prof/pmc_$c -- python3 bench.py --fast-fill --no-overlap --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/pmc_$c.json 2> gpurun_out/prof/pmc_$c.err; echo "pmc $c rc=$?"
done
python3 tools/pmc_summary.py gpurun_out/prof/pmc_FETCH_SIZE gpurun_out/prof/pmc_WRITE_SIZE 16 > gpurun_out/prof/pmc_hbm_traffic.json
python3 tools/trace_summary.py gpurun_out/prof/trace 12
