mkdir -p gpurun_out/r03
python -m pytest tests/test_bx3_gpu.py tests/test_golden.py tests/test_tts_gpu.py -x -q -m gpu > gpurun_out/r03/bx3_tests4.txt 2>&1; echo "rc=$?"; tail -8 gpurun_out/r03/bx3_tests4.txt
