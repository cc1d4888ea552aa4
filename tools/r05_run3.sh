#!/bin/bash
cd $GRAFT_REPO_ROOT
ARGS="ball3d 256 bf16 131072 push 256 bf16 131072"
for i in 1 2 3; do
echo "--- new (row-major A images)"; python tools/time_grad.py $ARGS 2>&1 | grep -v amdgpu.ids
echo "--- tr4 (transposed reads at 64-row groups)"; TMA_LIB_PATH=$PWD/tools/bin/libtma_hip_tr4.so python tools/time_grad.py $ARGS 2>&1 | grep -v amdgpu.ids
done
TMA_LIB_PATH=$PWD/tools/bin/libtma_hip_tr4.so python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -3
