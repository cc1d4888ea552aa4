#!/bin/bash
# small-minibatch f32 256x256 gradient call: eight-wave half groups (ring 4) against four-wave half groups (ring TMA_HALF_RING)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== default (eight waves)"; python tools/time_grad.py gridworld 256 f32 256 gridworld 256 f32 1024 2>&1 | grep "grad call"
echo "== TMA_WIDE_NW4=1"; TMA_WIDE_NW4=1 python tools/time_grad.py gridworld 256 f32 256 gridworld 256 f32 1024 2>&1 | grep "grad call"
done
