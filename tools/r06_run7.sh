mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_h256p_gpu.py "tests/test_ppo_gpu.py::test_persistent_epoch_kernel_falls_back_to_launches_when_it_cannot_run" "tests/test_ppo_gpu.py::test_persistent_epoch_kernel_long_epoch_stays_with_the_launch_path" -x -q 2>&1 | grep -v amdgpu.ids | tail -25 ) > gpurun_out/r06_t7.log
for i in 1; do
timeout 300 python tools/time_literal256.py basic 8 1024 256 10 2>&1 | grep "optimizer steps" >> gpurun_out/r06_t7.log
TMA_EPOCH_PER_CALL=1 timeout 300 python tools/time_literal256.py basic 8 1024 256 10 2>&1 | grep "optimizer steps" >> gpurun_out/r06_t7.log
done
cat gpurun_out/r06_t7.log
