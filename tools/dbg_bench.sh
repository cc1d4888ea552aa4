cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
timeout 120 python - <<'PY'
import time, torch, sys
sys.path.insert(0,'.')
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.training import make_vector_env
for (N,T,B) in [(4096,32,4096),(4096,128,16384),(4096,1024,131072)]:
    env = make_vector_env('gridworld', n_envs=N, seed=1)
    model = PPO("MlpPolicy", env, n_steps=T, batch_size=B, n_epochs=2, seed=1, policy_kwargs={"net_arch":[64,64]})
    torch.cuda.synchronize(); t0=time.perf_counter()
    model.collect_rollouts(); torch.cuda.synchronize(); t1=time.perf_counter()
    model.train(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(N,T,B,'rollout %.1f ms'%((t1-t0)*1e3),'train(2 epochs) %.1f ms'%((t2-t1)*1e3), flush=True)
    model.collect_rollouts(); torch.cuda.synchronize(); t3=time.perf_counter()
    model.train(); torch.cuda.synchronize(); t4=time.perf_counter()
    print('  2nd: rollout %.1f ms'%((t3-t2)*1e3),'train %.1f ms'%((t4-t3)*1e3), model.pop_train_stats(), flush=True)
    env.close()
PY
echo "rc=$?"
