#!/usr/bin/env python3
"""One train() of a wide policy, timed, with a digest of the parameters behind it: an A/B of two libraries (TMA_LIB_PATH) or of a switch
that must not change a bit.  usage: train_digest.py [task hidden dtype batch]..."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env

args = sys.argv[1:] or ["ball3d", "256", "bf16", "131072"]
for i in range(0, len(args), 4):
    task, H, dt, B = args[i], int(args[i + 1]), args[i + 2], int(args[i + 3])
    n_envs = 4096 if task not in ("crawler", "push") else 2048
    env = make_vector_env(task, n_envs=n_envs, seed=1)
    m = PPO("MlpPolicy", env, n_steps=max(32, 4 * B // n_envs), batch_size=B, n_epochs=2, seed=1, policy_kwargs={"net_arch": [H, H], "mfma_dtype": dt})
    m.collect_rollouts(); m.train(); torch.cuda.synchronize()
    best = None
    for _ in range(3):
        m.collect_rollouts(); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.train(); torch.cuda.synchronize(); dt_s = time.perf_counter() - t0
        best = dt_s if best is None else min(best, dt_s)
    n_mb = 2 * max(32, 4 * B // n_envs) * n_envs // B
    digest = hashlib.sha256(b"".join(v.detach().cpu().contiguous().numpy().tobytes() for _, v in sorted(m.policy.state_dict().items()))).hexdigest()[:16]
    st = m.pop_train_stats()
    print(f"{task} H={H} {dt} B={B}: {best / n_mb * 1e6:.1f} us per optimizer step ({n_mb} per train()); grad_norm {st.get('train/grad_norm', float('nan')):.9f}; parameters sha256 {digest}", flush=True)
