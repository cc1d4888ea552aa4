mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r06_gputest.log
echo "--- timing" >> gpurun_out/r06_gputest.log
timeout 300 python tools/time_literal256.py gridworld 4096 256 256 >> gpurun_out/r06_gputest.log 2>&1
timeout 300 python tools/time_literal256.py basic 8 1024 256 >> gpurun_out/r06_gputest.log 2>&1
timeout 300 python tools/h256p_ticks.py gridworld 1024 256 >> gpurun_out/r06_gputest.log 2>&1
cat gpurun_out/r06_gputest.log
