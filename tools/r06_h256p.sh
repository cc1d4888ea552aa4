# the 256-wide persistent epoch kernel: its tests, the literal-batch timings and the phase ticks (GPU box)
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_h256p_gpu.py -x -q 2>&1 | tail -8 ) > gpurun_out/h256p.log
for i in 1 2; do timeout 300 python tools/time_literal256.py gridworld 4096 256 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/h256p.log; done
timeout 300 python tools/time_literal256.py basic 8 1024 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/h256p.log
timeout 300 python tools/h256p_ticks.py gridworld 1024 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/h256p.log
cat gpurun_out/h256p.log
