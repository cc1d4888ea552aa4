#!/bin/bash
# The A/B switches of the library select the kernels of earlier rounds; each must keep passing the tests of the paths it touches.
# Run on the GPU box: bash tools/test_switches.sh
run() { echo "== $1"; env $1 python -m pytest $2 -m gpu -q -x ${3:+-k "$3"} 2>&1 | tail -1; }  # $3: a -k expression (tests that assert WHICH path ran are deselected under the switch that disables it)
run "TMA_SYNC_EVAL=1" "tests/test_dropin_gpu.py"
run "TMA_SYNC_LOGGING=1" "tests/test_dropin_gpu.py"
run "TMA_BF_NW4=1" "tests/test_bf16_gpu.py tests/test_rollout_oracle_gpu.py"
run "TMA_WIDE_NW4=1" "tests/test_ppo_gpu.py" "not deferred_dw2"
run "TMA_NO_HALF_GROUPS=1" "tests/test_ppo_gpu.py" "not deferred_dw2"
run "TMA_CONT_TWO_NET=1" "tests/test_bf16_gpu.py tests/test_rollout_oracle_gpu.py"
run "TMA_CONT_SERIAL=1" "tests/test_bf16_gpu.py tests/test_rollout_oracle_gpu.py"
run "TMA_WIDE_ROWS=32" "tests/test_bf16_gpu.py tests/test_rollout_oracle_gpu.py"
run "TMA_CONT_ROWS=32" "tests/test_bf16_gpu.py tests/test_rollout_oracle_gpu.py"
run "TMA_NO_DZ1_CACHE=1" "tests/test_bf16_gpu.py"
run "TMA_NO_NATIVE_RCCL=1" "tests/test_dist_gpu.py"
# round 5
run "TMA_DP_NO_FOLD=1" "tests/test_dist_gpu.py tests/test_ppo_gpu.py"
run "TMA_STEP_THREADS=64" "tests/test_env_gpu.py"
run "TMA_BF_NPI=128" "tests/test_bf16_gpu.py"
# (single-launch timing of the plain VecEnv.step path, round-3 library against this build on this box: python tools/step_ab.py --lib A.so --lib B.so)
run "TMA_P2P_NO_FUSE=1" "tests/test_dist_gpu.py"   # the peer exchange as push / pull launches of its own
run "TMA_NO_SPLIT3=1" "tests/test_dist_gpu.py"      # mfma_dtype 2 on the exact-f32 kernels
run "TMA_NO_DEFER_W2=1" "tests/test_ppo_gpu.py"            # small f32 256-wide minibatches on the slab path
# round 6
run "TMA_NO_PERSIST256=1" "tests/test_h256p_gpu.py tests/test_ppo_gpu.py" "not falls_back and not all_epochs_in_one_persistent_launch"   # the 256-wide literal-batch epoch as per-minibatch launches
run "TMA_EPOCH_PER_CALL=1" "tests/test_h256p_gpu.py"                          # one persistent launch per epoch instead of per train()
run "TMA_WIDE_F32_ROWS=16" "tests/test_ppo_gpu.py tests/test_rollout_oracle_gpu.py"   # the f32 256-wide fused rollout in 16-env tiles at every env count
run "TMA_WIDE_F32_ROWS=8" "tests/test_ppo_gpu.py"                             # ... and in 8-env tiles beyond 2048 envs
run "TMA_NO_CONT_F32_FUSED=1" "tests/test_ppo_gpu.py" "crawler or ant"   # the f32 Box-action rollouts step by step
run "TMA_ROLL2=1" "tests/test_ppo_gpu.py tests/test_rollout_oracle_gpu.py" "rollout"   # the headline rollout on two waves per tile (round 5) instead of four
