#!/usr/bin/env python3
"""Time one tma_ppo_minibatch_grad launch group and one tma_policy_act launch (HIP events) for a task / width / MFMA dtype.
usage: time_grad.py [task hidden dtype batch]..."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from three_mlagents_amd import _lib
from three_mlagents_amd.ppo import PPO
from three_mlagents_amd.harness import make_vector_env


def med_us(fn, reps=30, group=4):
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(group):
            fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 / group for a, b in evs)
    return ts[len(ts) // 2]


args = sys.argv[1:] or ["ball3d", "256", "bf16", "131072"]
for i in range(0, len(args), 4):
    task, H, dt, B = args[i], int(args[i + 1]), args[i + 2], int(args[i + 3])
    env = make_vector_env(task, n_envs=4096, seed=1)
    m = PPO("MlpPolicy", env, n_steps=max(32, B // 4096), batch_size=B, n_epochs=1, seed=1, policy_kwargs={"net_arch": [H, H], "mfma_dtype": dt})
    m.collect_rollouts()
    mb = _lib.Minibatch(None, 1, 0, 0, B)
    L = _lib.lib()

    def grad():
        _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(m.policy.params), C.byref(m.policy.dims), C.byref(m._rollout_view), C.byref(mb), C.byref(m._hp),
                                            _lib.ptr(m.grad), _lib.ptr(m.workspace), m._stream()))

    obs = m.buf["obs"][0]

    def act():
        m.policy.act(obs, rng_seed=1, rng_step=0)

    for _ in range(3):
        grad(), act()
    ks = []
    if L.tma_debug_time_grad_kernel(1) == 0:  # the persistent kernel alone (events recorded by the library around that launch)
        us = C.c_float(0.0)
        for _ in range(20):
            grad()
            if L.tma_debug_last_grad_kernel_us(C.byref(us)) == 0:
                ks.append(us.value)
        L.tma_debug_time_grad_kernel(0)
    kern = sorted(ks)[len(ks) // 2] if ks else float("nan")
    print(f"{task} H={H} {dt} B={B}: grad call {med_us(grad):.1f} us (kernel alone {kern:.1f} us)   act(4096) {med_us(act):.1f} us", flush=True)
    if os.environ.get("TMA_PHASE_DUMP") and dt == "bf16":
        grad(); torch.cuda.synchronize()
        ws = m.workspace
        end_offs = ws.numel() - (((1 << 22) // 1024 + (1 << 22) // 256) * 16 + 8192 * 8)  # offsets cache ends before the epoch / norm partials
        tail = ws[end_offs - 64 * 4:end_offs].cpu().numpy().view("int64")  # last 64 int32 of the offsets cache
        names = ["tail: P6 (MT4: dh1+dz1+dW1) of prev group", "P0 commit", "P1 L1", "P2 L2fwd", "P3a", "P3 head+loss+Z3", "P4 dW3+dz2", "P5 (MT4: dW2 only)", "after loop"]
        for role, o in (("pi", 0), ("vf", 12)):
            v = tail[o:o + 9]
            print("   ", role, "cycles:", {n: int(x) for n, x in zip(names, v)}, "sum", int(v.sum()))
