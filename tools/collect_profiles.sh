#!/bin/bash
# Copies what tools/gpu_round.sh left under gpurun_out/prof/ (scratch) into profiles/<round>/ (tracked), under stable names.
# usage: tools/collect_profiles.sh r02
set -e
R=${1:?round name}
P=gpurun_out/prof
D=profiles/$R
mkdir -p $D
cp $P/bench_default_run.json $D/bench_default_run.json
cp $P/host_enqueue_eager.txt $P/host_enqueue_graphs.txt $D/
cp $P/pmc_hbm_traffic.json $P/pmc_mfma_b64.json $P/pmc_mfma_b1024.json $D/
# rocprofv3 writes one directory per traced process: keep the kernel-stats table of the one that ran the bench (the largest)
for t in trace_1stream:kernel_stats_single_stream trace:kernel_stats_overlapped; do
  src=${t%%:*}; dst=${t##*:}
  f=$(ls -t $P/$src/*/*_kernel_stats.csv | head -1)  # tools/gpu_round.sh wipes the trace directories at the start of a visit and traces no child legs: the newest table is this visit's
  cp "$f" $D/$dst.csv
  grep '^{' $P/${src}_bench.json > $D/$dst.bench.json || true
done
python3 tools/trace_summary.py $P/trace_1stream 14 > $D/kernel_trace_single_stream_summary.txt
python3 tools/trace_summary.py $P/trace 14 > $D/kernel_trace_overlapped_summary.txt
if [ -d $P/trace_tts ]; then
  f=$(ls -t $P/trace_tts/*/*_kernel_stats.csv | head -1)
  cp "$f" $D/tts_kernel_stats.csv
  grep '^{' $P/trace_tts_bench.json > $D/tts_kernel_stats.bench.json || true
  python3 tools/trace_table.py $P/trace_tts > $D/tts_kernel_trace_summary.txt
fi
if [ -d $P/trace_26b ]; then
  f=$(ls -t $P/trace_26b/*/*_kernel_stats.csv | head -1)
  cp "$f" $D/kernel_stats_stt_2.6b_single_stream.csv
  grep '^{' $P/trace_26b_bench.json > $D/kernel_stats_stt_2.6b_single_stream.bench.json || true
  cp $P/kernel_trace_stt_2.6b_summary.txt $P/pmc_hbm_traffic_stt_2.6b.json $D/
fi
if [ -d $P/extra ]; then
  mkdir -p $D/extra
  for f in $P/extra/*.json; do grep '^{' "$f" > $D/extra/$(basename "$f") || true; done
  cp $P/extra/*.txt $D/extra/ 2>/dev/null || true
fi
ls -la $D $D/extra
