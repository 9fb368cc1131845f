#!/bin/bash
# One GPU-box visit that produces everything profiles/rNN/ keeps: the default bench (as the driver runs it), rocprofv3 kernel
# traces (serialized single-stream run and the default overlapped run), the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, one
# counter per pass, no tracing domains) and the MFMA-utilisation PMC pass (B = 64 and B = 1024).  Profiling runs use eager
# launches (DSM_GRAPHS=0: one dispatch record per kernel).  Outputs under gpurun_out/prof/.
set -o pipefail
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
P=$R/gpurun_out/prof
step() { name=$1; secs=$2; shift 2; echo "=== $name"; timeout -k 10 $secs "$@"; rc=$?; echo "=== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
step bench 500 bash -c "python bench.py > $P/bench_default_run.json 2> $P/bench_default.err"
step enq0 120 bash -c "DSM_GRAPHS=0 python experiments/host_enqueue_rate.py > $P/host_enqueue_eager.txt 2>/dev/null"
step enq1 120 bash -c "python experiments/host_enqueue_rate.py > $P/host_enqueue_graphs.txt 2>/dev/null"
export DSM_GRAPHS=0
step trace1 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_1stream -- python3 bench.py --fast-fill --no-overlap --steps 30 --warmup 3 --no-cpu-baseline --capacity-legs "" > $P/trace_1stream_bench.json 2> $P/trace_1stream.err
step trace 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --fast-fill --steps 30 --warmup 3 --no-cpu-baseline --capacity-legs "" > $P/trace_bench.json 2> $P/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  step pmc_$c 400 rocprofv3 --pmc $c --output-format csv -d $P/pmc_$c -- python3 bench.py --fast-fill --no-overlap --steps 3 --warmup 1 --no-cpu-baseline --capacity-legs "" > $P/pmc_$c.json 2> $P/pmc_$c.err
done
python3 tools/pmc_summary.py $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE 16 > $P/pmc_hbm_traffic.json
MF="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
step pmc_mfma64 300 rocprofv3 --pmc $MF --output-format csv -d $P/pmc_mfma_b64 -- python3 bench.py --fast-fill --no-overlap --steps 3 --warmup 1 --no-cpu-baseline --capacity-legs "" > $P/pmc_mfma_b64_bench.json 2> $P/pmc_mfma_b64.err
python3 tools/pmc_mfma_summary.py $P/pmc_mfma_b64 30 > $P/pmc_mfma_b64.json
step pmc_mfma1024 300 rocprofv3 --pmc $MF --output-format csv -d $P/pmc_mfma_b1024 -- python3 bench.py --batch 1024 --fast-fill --no-overlap --steps 2 --warmup 1 --no-cpu-baseline --capacity-legs "" > $P/pmc_mfma_b1024_bench.json 2> $P/pmc_mfma_b1024.err
python3 tools/pmc_mfma_summary.py $P/pmc_mfma_b1024 30 > $P/pmc_mfma_b1024.json
python3 tools/trace_summary.py $P/trace_1stream 14
ls $P
