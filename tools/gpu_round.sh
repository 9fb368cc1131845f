#!/bin/bash
# One GPU-box visit: tests, smoke, rocprofv3 kernel trace + PMC passes of the bench.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
timeout -k 10 200 python -u __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
# kernel trace (steady-state positions, no CPU baseline): per-kernel average durations
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/trace -- python3 bench.py --fast-fill --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/prof/trace_bench.json 2> gpurun_out/prof/trace.err; echo "trace rc=$?"
# HBM traffic counters, each in its own pass
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -- python3 bench.py --fast-fill --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/pmc_fetch_bench.json 2> gpurun_out/prof/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -- python3 bench.py --fast-fill --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof/pmc_write_bench.json 2> gpurun_out/prof/pmc_write.err; echo "pmc write rc=$?"
ls -R gpurun_out/prof | head -40
du -sh gpurun_out/prof
