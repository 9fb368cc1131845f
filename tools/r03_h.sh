mkdir -p gpurun_out/r03/prof_m1g1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
DSM_LM_GROUPS=1 DSM_GRAPHS=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r03/prof_m1g1 -o m1g1 -- python3 bench.py --fast-fill --no-cpu-baseline --host-path-legs "" --capacity-legs "" --dot-mode 1 --steps 30 --warmup 5 --no-overlap > gpurun_out/r03/prof_m1g1/bench.json 2> gpurun_out/r03/prof_m1g1/bench.err || { tail -5 gpurun_out/r03/prof_m1g1/bench.err; exit 1; }
find gpurun_out/r03/prof_m1g1 -name "*kernel_stats*" | head
f=$(find gpurun_out/r03/prof_m1g1 -name "*kernel_stats.csv" | head -1)
head -40 $f | cut -c1-260
find gpurun_out/r03/prof_m1g1 -name "*kernel_trace.csv" -size +60M -delete
