mkdir -p gpurun_out/r03/m64probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r03/m64probe -- ./experiments/gemm_m64_probe > gpurun_out/r03/m64probe/out.txt 2>&1 || { tail -5 gpurun_out/r03/m64probe/out.txt; exit 1; }
python tools/trace_table.py gpurun_out/r03/m64probe probe2 > gpurun_out/r03/m64probe/table.txt
cat gpurun_out/r03/m64probe/table.txt
find gpurun_out/r03/m64probe -name "*.csv" -size +20M -delete
