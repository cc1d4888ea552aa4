#!/bin/bash
# A/B of two libraries on ONE box: bf16 gradient launch group (Ball3D / Push / Crawler shapes), alternating
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
OLD=${1:-tools/bin/libtma_hip_prev.so}
NEW=three-mlagents_amd/csrc/libtma_hip.so
for i in 1 2 3; do
for lib in $OLD $NEW; do
echo "== $lib"
TMA_LIB_PATH=$lib python tools/time_grad.py ball3d 256 bf16 131072 push 256 bf16 131072 2>&1 | grep -v "^$" | tail -4
done
done
python -m pytest tests/test_bf16_gpu.py tests/test_wave_layouts_gpu.py -x -q -m gpu 2>&1 | tail -2
