# bf16 gradient kernel A/B on one box: previous library against this one -- digest of the parameters after two train() calls, time per optimizer step, kernel alone
mkdir -p gpurun_out
: > gpurun_out/bf_ab.log
CFG=${CFG:-"ball3d 256 bf16 131072 push 256 bf16 131072 gridworld 256 bf16 131072"}
for i in 1 2; do
for lib in tools/bin/libtma_hip_prev.so three-mlagents_amd/csrc/libtma_hip.so; do
echo "== $lib" >> gpurun_out/bf_ab.log
TMA_LIB_PATH=$lib timeout 600 python tools/train_digest.py $CFG 2>&1 | grep "optimizer step" >> gpurun_out/bf_ab.log
TMA_LIB_PATH=$lib timeout 300 python tools/time_grad.py $CFG 2>&1 | grep "grad call" >> gpurun_out/bf_ab.log
done
done
cat gpurun_out/bf_ab.log
