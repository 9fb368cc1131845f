#!/bin/bash
# One GPU-box visit that produces everything profiles/rNN/ keeps: the default bench (as the driver runs it), rocprofv3 kernel
# traces (serialized single-stream run and the default overlapped run), the HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, one
# counter per pass, no tracing domains) and the MFMA-utilisation PMC pass (B = 64 and B = 1024).  Profiling runs use eager
# launches (DSM_GRAPHS=0: one dispatch record per kernel).  Outputs under gpurun_out/prof/.
# usage: tools/gpu_round.sh [1|2|3|all]   (a gpurun call is limited to 1200 s: part 1 = default bench + traces + PMC passes of the
# headline config, part 2 = parts of the step, the capacity batch, the other BASELINE configs, TTS, probes,
# part 3 = stt-2.6b-en kernel trace + HBM PMC passes, B = 2048 timeline)
set -o pipefail
PART=${1:-all}
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
R=$PWD
P=$R/gpurun_out/prof
step() { name=$1; secs=$2; shift 2; echo "=== $name"; timeout -k 10 $secs "$@"; rc=$?; echo "=== $name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi; }
QUIET='--no-cpu-baseline --capacity-legs "" --host-path-legs "" --other-configs "" --no-agreement'
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
rm -rf $P/trace_1stream $P/trace $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE $P/pmc_mfma_b64 $P/pmc_mfma_b1024  # gpurun_out/ accumulates across visits: never publish an earlier visit's tables
step bench 500 bash -c "python bench.py > $P/bench_default_run.json 2> $P/bench_default.err"
step enq0 120 bash -c "DSM_GRAPHS=0 python experiments/host_enqueue_rate.py > $P/host_enqueue_eager.txt 2>/dev/null"
step enq1 120 bash -c "python experiments/host_enqueue_rate.py > $P/host_enqueue_graphs.txt 2>/dev/null"
export DSM_GRAPHS=0
step trace1 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_1stream -- python3 bench.py --fast-fill --no-overlap --steps 30 --warmup 3 --no-cpu-baseline --capacity-legs "" --host-path-legs "" --other-configs "" --no-agreement > $P/trace_1stream_bench.json 2> $P/trace_1stream.err
step trace 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 bench.py --fast-fill --steps 30 --warmup 3 --no-cpu-baseline --capacity-legs "" --host-path-legs "" --other-configs "" --no-agreement > $P/trace_bench.json 2> $P/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  step pmc_$c 400 rocprofv3 --pmc $c --output-format csv -d $P/pmc_$c -- python3 bench.py --fast-fill --no-overlap --steps 3 --warmup 1 --no-cpu-baseline --capacity-legs "" --host-path-legs "" --other-configs "" --no-agreement > $P/pmc_$c.json 2> $P/pmc_$c.err
done
python3 tools/pmc_summary.py $P/pmc_FETCH_SIZE $P/pmc_WRITE_SIZE 16 > $P/pmc_hbm_traffic.json
MF="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
step pmc_mfma64 300 rocprofv3 --pmc $MF --output-format csv -d $P/pmc_mfma_b64 -- python3 bench.py --fast-fill --no-overlap --steps 3 --warmup 1 --no-cpu-baseline --capacity-legs "" --host-path-legs "" --other-configs "" --no-agreement > $P/pmc_mfma_b64_bench.json 2> $P/pmc_mfma_b64.err
python3 tools/pmc_mfma_summary.py $P/pmc_mfma_b64 30 > $P/pmc_mfma_b64.json
step pmc_mfma1024 300 rocprofv3 --pmc $MF --output-format csv -d $P/pmc_mfma_b1024 -- python3 bench.py --batch 1024 --fast-fill --no-overlap --steps 2 --warmup 1 --no-cpu-baseline --capacity-legs "" --host-path-legs "" --other-configs "" --no-agreement > $P/pmc_mfma_b1024_bench.json 2> $P/pmc_mfma_b1024.err
python3 tools/pmc_mfma_summary.py $P/pmc_mfma_b1024 30 > $P/pmc_mfma_b1024.json
unset DSM_GRAPHS
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
rm -rf $P/trace_tts $P/m64probe
# where the time goes at the capacity batch, the other two BASELINE configurations, and the co-run probes
X=$P/extra
mkdir -p $X
C="python bench.py --batch 2048 --fast-fill --steps 12 --warmup 3 --no-cpu-baseline --capacity-legs '' --host-path-legs '' --other-configs '' --no-agreement"
step b2048_all 200 bash -c "$C > $X/b2048_all.json 2> $X/b2048_all.err"
step b2048_lm 200 bash -c "$C --part lm > $X/b2048_lm.json 2> $X/b2048_lm.err"
step b2048_enc 200 bash -c "$C --part enc > $X/b2048_enc.json 2> $X/b2048_enc.err"
step b2048_lm_g1 200 bash -c "DSM_LM_GROUPS=1 $C --part lm > $X/b2048_lm_one_group.json 2> $X/b2048_lm_g1.err"
step b2048_1s 200 bash -c "$C --no-overlap > $X/b2048_single_stream.json 2> $X/b2048_1s.err"
step b26 300 bash -c "python bench.py --config stt-2.6b-en --batch 128 --fast-fill --steps 50 --warmup 5 --no-cpu-baseline --capacity-legs '' --host-path-legs '' > $X/bench_stt_2.6b_b128.json 2> $X/b26.err"
step tts 200 bash -c "python bench.py --workload tts --batch 32 --steps 50 --warmup 5 --no-cpu-baseline --capacity-legs '' > $X/bench_tts_b32.json 2> $X/tts.err"
step tts_m0 200 bash -c "python bench.py --workload tts --batch 32 --steps 50 --warmup 5 --dot-mode 0 > $X/bench_tts_b32_dot_mode0.json 2> $X/tts_m0.err"
step tts_trace 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_tts -- python3 bench.py --workload tts --batch 32 --steps 30 --warmup 3 > $P/trace_tts_bench.json 2> $P/trace_tts.err
C64="python bench.py --fast-fill --steps 100 --warmup 10 --no-cpu-baseline --host-path-legs '' --capacity-legs '' --other-configs '' --no-agreement"
step b64_lm 200 bash -c "$C64 --part lm > $X/b64_lm.json 2> $X/b64_lm.err"
step b64_enc 200 bash -c "$C64 --part enc > $X/b64_enc.json 2> $X/b64_enc.err"
step b64_lm_g1 200 bash -c "DSM_LM_GROUPS=1 $C64 --part lm > $X/b64_lm_one_group.json 2> $X/b64_lm_g1.err"
step b64_g1 200 bash -c "DSM_LM_GROUPS=1 $C64 > $X/b64_one_group.json 2> $X/b64_g1.err"
step b64_m0 300 bash -c "python bench.py --fast-fill --dot-mode 0 --no-cpu-baseline --host-path-legs '' --capacity-legs 400,2048 --other-configs '' --no-agreement > $X/bench_b64_dot_mode0.json 2> $X/b64_m0.err"
if [ -x experiments/gemm_m64_probe ]; then
  step m64probe 200 rocprofv3 --kernel-trace --output-format csv -d $P/m64probe -- ./experiments/gemm_m64_probe > $X/gemm_m64_probe.out 2>&1
  python3 tools/trace_table.py $P/m64probe probe > $X/gemm_m64_probe.txt
fi
if [ -x experiments/gemm_wk_probe ]; then
  step wkprobe 200 bash -c "cd experiments && ./gemm_wk_probe 1 > $X/gemm_wk_probe.txt 2>&1 && ./gemm_wk_probe 2 > $X/gemm_wk_probe_stamps.txt 2>&1"
fi
fi
if [ "$PART" = 3 ] || [ "$PART" = all ]; then
rm -rf $P/trace_26b $P/pmc26_FETCH_SIZE $P/pmc26_WRITE_SIZE
X=$P/extra
mkdir -p $X
export DSM_GRAPHS=0
A26="python3 bench.py --config stt-2.6b-en --batch 128 --fast-fill --no-overlap --no-cpu-baseline --capacity-legs '' --host-path-legs ''"
step trace26 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace_26b -- python3 bench.py --config stt-2.6b-en --batch 128 --fast-fill --no-overlap --steps 20 --warmup 3 --no-cpu-baseline --capacity-legs "" --host-path-legs "" > $P/trace_26b_bench.json 2> $P/trace_26b.err
for c in FETCH_SIZE WRITE_SIZE; do
  step pmc26_$c 400 rocprofv3 --pmc $c --output-format csv -d $P/pmc26_$c -- python3 bench.py --config stt-2.6b-en --batch 128 --fast-fill --no-overlap --steps 2 --warmup 1 --no-cpu-baseline --capacity-legs "" --host-path-legs "" > $P/pmc26_$c.json 2> $P/pmc26_$c.err
done
python3 tools/pmc_summary.py $P/pmc26_FETCH_SIZE $P/pmc26_WRITE_SIZE 32 > $P/pmc_hbm_traffic_stt_2.6b.json
python3 tools/trace_summary.py $P/trace_26b 14 > $P/kernel_trace_stt_2.6b_summary.txt
unset DSM_GRAPHS
step tl2048 300 bash -c "python tools/timeline.py 2048 2 > $X/timeline_b2048.txt 2>&1"
step tl64 200 bash -c "python tools/timeline.py 64 2 > $X/timeline_b64.txt 2>&1"
step sweep 400 bash -c "tools/knob_sweep_b64.sh > $X/knob_sweep_b64.txt 2>&1"
fi
python3 tools/trace_summary.py $P/trace_1stream 14 2>/dev/null
ls $P $P/extra
