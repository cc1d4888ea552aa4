#!/bin/bash
# small-minibatch f32 256x256 gradient call with and without the deferred dW2 (TMA_NO_DEFER_W2=1: the slab path), one box
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ppo_gpu.py -x -q -m gpu -k "gradient_matches_autograd or adam_step_local or wide_policy or epoch_loop or ppo_learns" 2>&1 | tail -2
for i in 1 2; do
echo "== deferred"; python tools/time_grad.py gridworld 256 f32 256 gridworld 256 f32 1024 basic 256 f32 256 2>&1 | grep "grad call"
echo "== TMA_NO_DEFER_W2=1"; TMA_NO_DEFER_W2=1 python tools/time_grad.py gridworld 256 f32 256 gridworld 256 f32 1024 basic 256 f32 256 2>&1 | grep "grad call"
done
