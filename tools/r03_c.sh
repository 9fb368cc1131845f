mkdir -p gpurun_out/r03
python -m pytest tests/test_tts_ca_gpu.py tests/test_tts_gpu.py tests/test_graphs_gpu.py -x -q -m gpu > gpurun_out/r03/tts_ca_tests.txt 2>&1; echo "rc=$?"; tail -30 gpurun_out/r03/tts_ca_tests.txt
