# A/B of two libraries on ONE box: the literal-batch epoch (h256p) and the wide gradient launch groups
mkdir -p gpurun_out
OLD=${1:-tools/bin/libtma_hip_prev.so}
NEW=three-mlagents_amd/csrc/libtma_hip.so
: > gpurun_out/ab2.log
for i in 1 2; do
for lib in $OLD $NEW; do
echo "== $lib" >> gpurun_out/ab2.log
TMA_LIB_PATH=$lib timeout 300 python tools/time_literal256.py gridworld 4096 256 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab2.log
TMA_LIB_PATH=$lib timeout 300 python tools/time_grad.py gridworld 256 f32 131072 ball3d 256 bf16 131072 gridworld 256 bf16x3 131072 2>&1 | grep "grad call" >> gpurun_out/ab2.log
done
done
cat gpurun_out/ab2.log
