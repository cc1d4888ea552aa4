mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_h256p_gpu.py -x -q 2>&1 | tail -30 > gpurun_out/h256p_test.log
echo "--- timing" >> gpurun_out/h256p_test.log
timeout 300 python tools/time_literal256.py gridworld 4096 256 256 >> gpurun_out/h256p_test.log 2>&1
timeout 300 python tools/time_literal256.py basic 8 1024 256 >> gpurun_out/h256p_test.log 2>&1
TMA_NO_PERSIST=1 timeout 300 python tools/time_literal256.py gridworld 4096 256 256 >> gpurun_out/h256p_test.log 2>&1
timeout 300 python tools/h256p_ticks.py gridworld 1024 256 >> gpurun_out/h256p_test.log 2>&1
cat gpurun_out/h256p_test.log
