# A/B of two builds of the library on one box: alternating runs of the literal-batch timing (GPU box)
mkdir -p gpurun_out
A=three-mlagents_amd/csrc/libtma_hip.so
B=three-mlagents_amd/csrc/libtma_hip_qab.so
: > gpurun_out/ab.log
for i in 1 2 3; do
  echo "A:" >> gpurun_out/ab.log; TMA_LIB_PATH=$A timeout 300 python tools/time_literal256.py gridworld 4096 256 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log
  echo "B:" >> gpurun_out/ab.log; TMA_LIB_PATH=$B timeout 300 python tools/time_literal256.py gridworld 4096 256 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log
done
echo "A basic:" >> gpurun_out/ab.log; TMA_LIB_PATH=$A timeout 300 python tools/time_literal256.py basic 8 1024 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log
echo "B basic:" >> gpurun_out/ab.log; TMA_LIB_PATH=$B timeout 300 python tools/time_literal256.py basic 8 1024 256 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab.log
cat gpurun_out/ab.log
