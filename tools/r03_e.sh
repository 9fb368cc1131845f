mkdir -p gpurun_out/r03
python -m pytest tests/test_bx3_gpu.py -x -q -m gpu > gpurun_out/r03/bx3_tests.txt 2>&1; echo "rc=$?"; tail -30 gpurun_out/r03/bx3_tests.txt
