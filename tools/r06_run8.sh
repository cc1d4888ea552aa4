mkdir -p gpurun_out
: > gpurun_out/r06_t8.log
for i in 1 2; do
timeout 300 python tools/time_literal256.py basic 8 1024 256 10 2>&1 | grep "optimizer steps" >> gpurun_out/r06_t8.log
done
cat > /tmp/prof_train.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env
env = make_vector_env("basic", n_envs=8, seed=1)
m = PPO("MlpPolicy", env, n_steps=1024, batch_size=256, n_epochs=10, seed=1, policy_kwargs={"net_arch": [256, 256]})
for _ in range(3):
    m.collect_rollouts(); m.train()
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d gpurun_out/r06_basic_train -- python /tmp/prof_train.py > gpurun_out/r06_basic_train.log 2>&1
cp $(ls -t gpurun_out/r06_basic_train/*/*kernel_stats.csv | head -1) gpurun_out/r06_basic_train_kernel_stats.csv
cp $(ls -t gpurun_out/r06_basic_train/*/*hip_api_stats.csv | head -1) gpurun_out/r06_basic_train_hip_stats.csv
cat gpurun_out/r06_t8.log; head -12 gpurun_out/r06_basic_train_kernel_stats.csv | cut -c1-150; head -14 gpurun_out/r06_basic_train_hip_stats.csv
