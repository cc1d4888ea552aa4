#!/bin/bash
# A/B of two libraries on ONE box: f32 256-wide gradient launch group (GridWorld / Push / Ball3D shapes at 131072 samples; 256 / 1024 samples)
cd $GRAFT_REPO_ROOT
OLD=${1:-tools/bin/libtma_hip_prev.so}
NEW=three-mlagents_amd/csrc/libtma_hip.so
python -m pytest tests/test_ppo_gpu.py tests/test_wave_layouts_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
for lib in $OLD $NEW; do
echo "== $lib"
TMA_LIB_PATH=$lib python tools/time_grad.py gridworld 256 f32 131072 push 256 f32 131072 gridworld 256 f32 256 gridworld 256 f32 1024 2>&1 | grep "grad call"
done
done
