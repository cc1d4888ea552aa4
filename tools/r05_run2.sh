#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/grad_bits_ab.py tools/bin/libtma_hip_prev.so three-mlagents_amd/csrc/libtma_hip.so ball3d bf16 16384 crawler bf16 16384 gridworld bf16 16384 2>&1 | grep -v amdgpu.ids
ARGS="ball3d 256 bf16 131072 push 256 bf16 131072 crawler 256 bf16 131072"
for i in 1 2; do
echo "--- prev"; TMA_LIB_PATH=$PWD/tools/bin/libtma_hip_prev.so python tools/time_grad.py $ARGS 2>&1 | grep -v amdgpu.ids
echo "--- new"; python tools/time_grad.py $ARGS 2>&1 | grep -v amdgpu.ids
done
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
