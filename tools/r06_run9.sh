mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_h256p_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/r06_t9.log
cat gpurun_out/r06_t9.log
bash tools/r06_ab2.sh 2>&1 | grep -v "grad call"
timeout 300 python tools/h256p_ticks.py gridworld 1024 256 2>&1 | grep -v amdgpu.ids
