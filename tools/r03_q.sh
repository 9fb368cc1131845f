mkdir -p gpurun_out/r03
python -m pytest tests -x -q -m gpu > gpurun_out/r03/gpu_tests_2.txt 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03/gpu_tests_2.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/smoke_2.txt 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03/smoke_2.txt
