mkdir -p gpurun_out
: > gpurun_out/roll8.log
timeout 1200 python -m pytest tests/test_ppo_gpu.py -x -q -k "test_native_rollout_equals_stepwise_composition and 256 and f32" 2>&1 | tail -5 >> gpurun_out/roll8.log
CFG="basic 8 2048 256 f32 gridworld 8 2048 256 f32 gridworld 2048 256 256 f32 basic 1024 256 256 f32"
for i in 1 2; do
echo "== 16-env tiles" >> gpurun_out/roll8.log
TMA_WIDE_F32_ROWS=16 timeout 300 python tools/time_rollout.py $CFG 2>&1 | grep "rollout" >> gpurun_out/roll8.log
echo "== 8-env tiles" >> gpurun_out/roll8.log
timeout 300 python tools/time_rollout.py $CFG 2>&1 | grep "rollout" >> gpurun_out/roll8.log
done
cat gpurun_out/roll8.log
