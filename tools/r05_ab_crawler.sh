#!/bin/bash
# A/B of two libraries on ONE box: bf16 gradient launch group at the Crawler width (two launches: chain pass + dW1 pass)
cd $GRAFT_REPO_ROOT
OLD=${1:-tools/bin/libtma_hip_prev.so}
NEW=three-mlagents_amd/csrc/libtma_hip.so
python -m pytest tests/test_bf16_gpu.py tests/test_wave_layouts_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do
for lib in $OLD $NEW; do
echo "== $lib"
TMA_LIB_PATH=$lib python tools/time_grad.py crawler 256 bf16 131072 2>&1 | grep "grad call"
done
done
