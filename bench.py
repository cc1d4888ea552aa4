#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of full PPO iterations on GridWorld, 4096 envs per MI355X (BASELINE.json configs[1]).

One "step" = one complete PPO iteration of the hot path on one GPU:
    rollout of n_steps vector steps (policy forward + sampling, batched env step with auto-reset / terminal obs /
    Monitor sums, timeout bootstrap, MT19937 reset-ring refills) -> GAE -> n_epochs x minibatches of
    (advantage stats, forward+backward of the clipped-surrogate loss, [RCCL all-reduce of the flat gradient], clip + Adam).
Nothing is skipped inside the timed region.  `value` = env-steps of all ranks / max-over-ranks wall time.

Run:  python bench.py [--gpus N --steps K --warmup W]      (N > 1: launched by torch.distributed.run, one rank per GPU)
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # CPU-baseline leg: OpenMP (oracle) and torch pools must not spin against each other

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic HBM bytes per env-step of the step kernel as this build lays it out: state words in+out,
# Monitor sum in+out, episode index in+out, action in, obs + reward + 2 flags out.  SURVEY.md §8d's formula (i32 fields)
# is reported next to it.
LAYOUT_BYTES = {"gridworld": 4 + 4 + 4 + 8 + 8 + 4 + 4 + 16 + 4 + 2, "push": 58, "basic": 4 + 8 + 16 + 8 + 84 + 6, "ball3d": 4 + 72 + 16 + 8 + 24 + 6}
SURVEY_BYTES = {"gridworld": 90, "push": 74, "basic": 110, "ball3d": 114}
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense peak (~2.5 PF)


def mlp_flops_per_sample(D, H, A):
    fwd = 2 * ((D * H + H * H) * 2 + H * A + H)
    return fwd, 3 * fwd


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--task", default="gridworld")
    ap.add_argument("--n-envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--n-steps", type=int, default=1024)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--n-epochs", type=int, default=10)
    ap.add_argument("--batch-size", type=int, default=0, help="0 -> 32 minibatches per epoch (the reference's default schedule: 8 envs x 1024 / 256)")
    ap.add_argument("--mfma-dtype", default="f32", choices=["f32", "bf16"],
                    help="MFMA operand type of the hidden-layer GEMMs (bf16: hidden 128/192/256 only; BASELINE.json configs[2])")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--sweep", action="store_true", help="also run the env-count sweep of the step kernel (extra JSON field)")
    return ap.parse_args()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def timed_kernel_us(fn, reps, stream_sync, group=1):
    """Duration of one launch from HIP events recorded on the launch stream.  group > 1 brackets `group` back-to-back
    launches with one event pair (a per-launch event pair costs more than a ~5 us kernel); returns (mean, median) per launch."""
    import torch

    n_groups = max(1, reps // group)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_groups)]
    for a, b in evs:
        a.record()
        for _ in range(group):
            fn()
        b.record()
    stream_sync()
    ts = sorted(a.elapsed_time(b) * 1e3 / group for a, b in evs)
    return sum(ts) / len(ts), ts[len(ts) // 2]


def cpu_baseline(args, seconds):
    """Oracle (C port, OpenMP over envs) + torch-CPU restatement of the SB3 policy/update, on a bounded sample of the workload."""
    import numpy as np
    import torch

    from oracle import oracle as orc
    from oracle import sb3_ref

    cores = min(os.cpu_count() or 1, 32)  # threads actually used by both the OpenMP oracle and torch
    torch.set_num_threads(cores)
    log(f"cpu_baseline: {cores} threads, budget {seconds:.0f} s")
    N, D, A, H = args.n_envs, orc.obs_dim(args.task), orc.num_actions(args.task), args.hidden
    T = 16
    env = orc.OracleVecEnv(args.task, N, seed=args.seed, threads=cores)
    sd = sb3_ref.init_policy(D, H, A, False, seed=args.seed)
    obs = torch.from_numpy(env.reset())
    t0 = time.perf_counter()
    iters = 0
    env_only_t = 0.0
    while True:
        b_obs, b_act, b_lp, b_val, b_rew, b_done = [], [], [], [], [], []
        for _ in range(T):
            with torch.no_grad():
                logits, values = sb3_ref.forward(sd, obs)
                dist = torch.distributions.Categorical(logits=logits)
                act = dist.sample()
                lp = dist.log_prob(act)
            e0 = time.perf_counter()
            o = env.step(act.numpy().astype(np.int32))
            env_only_t += time.perf_counter() - e0
            b_obs.append(obs), b_act.append(act), b_lp.append(lp), b_val.append(values), b_rew.append(torch.from_numpy(o["rew32"]))
            b_done.append(torch.from_numpy((o["term"] | o["trunc"]).astype(np.float32)))
            obs = torch.from_numpy(o["obs"])
        with torch.no_grad():
            _, last_v = sb3_ref.forward(sd, obs)
        rew, val, done = torch.stack(b_rew).numpy(), torch.stack(b_val).numpy(), torch.stack(b_done).numpy()
        es = np.concatenate([np.zeros((1, N), np.float32), done[:-1]])
        adv, ret = orc.gae(rew, val, es, last_v.numpy(), done[-1].astype(np.uint8))
        tr = sb3_ref.RefTrainer(sd)
        fo, fa, fl = torch.cat(b_obs), torch.cat(b_act), torch.cat(b_lp)
        fadv, fret = torch.from_numpy(adv).reshape(-1), torch.from_numpy(ret).reshape(-1)
        total = T * N
        bs = args.batch_size if args.batch_size > 0 else max(256, total // 32)
        for _ in range(args.n_epochs):
            perm = torch.randperm(total)
            for s in range(0, total, bs):
                idx = perm[s:s + bs]
                tr.step(fo[idx], fa[idx], fl[idx], fadv[idx], fret[idx], clip_range=0.2, ent_coef=0.01, vf_coef=0.5)
            if iters > 0 and time.perf_counter() - t0 > 3 * seconds:
                break  # hard bound: never let the baseline leg run away
        sd = {k: v.detach() for k, v in tr.sd.items()}
        iters += 1
        el = time.perf_counter() - t0
        if el > seconds:
            break
    return {
        "value": iters * T * N / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": f"{iters} reduced PPO iterations of {T} vector steps x {N} envs ({args.task}, MLP {H}x{H}, {args.n_epochs} epochs): C oracle env "
                  f"(OpenMP, {cores} threads) + torch-CPU restatement of SB3 policy/GAE/update ({cores} threads)",
        "env_only_steps_per_s": iters * T * N / max(env_only_t, 1e-9),
    }


def main():
    args = parse()
    import torch

    from three_mlagents_amd import dist

    rank, local_rank, world = dist.init_from_env()
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from three_mlagents_amd import _lib
    from three_mlagents_amd.ppo import PPO
    from three_mlagents_amd.training import make_vector_env

    N, T = args.n_envs, args.n_steps
    total = N * T
    batch = args.batch_size if args.batch_size > 0 else max(256, total // 32)
    env = make_vector_env(args.task, n_envs=N, seed=args.seed, device=dev, env_offset=rank * N)
    model = PPO("MlpPolicy", env, learning_rate=3e-4, n_steps=T, batch_size=batch, n_epochs=args.n_epochs, gamma=0.99, gae_lambda=0.95,
                clip_range=0.2, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, seed=args.seed,
                policy_kwargs={"net_arch": {"pi": [args.hidden] * 2, "vf": [args.hidden] * 2}, "mfma_dtype": args.mfma_dtype})
    D, A = model.policy.obs_dim, model.policy.act_dim

    def iteration():
        model.collect_rollouts()
        model.train()

    log(f"rank {rank}/{world}: engine ready, N={N} T={T} batch={batch}")
    for _ in range(args.warmup):
        iteration()
    log("warmup done")
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    t_roll = 0.0
    for _ in range(args.steps):
        torch.cuda.synchronize()  # attribute the asynchronous tail of the previous update to the update, not to this rollout
        r0 = time.perf_counter()
        model.collect_rollouts()
        torch.cuda.synchronize()
        t_roll += time.perf_counter() - r0
        model.train()
    torch.cuda.synchronize()
    dist.barrier()
    el_local = time.perf_counter() - t0
    el = dist.allreduce_max_float(el_local, device=dev)
    t_roll = dist.allreduce_max_float(t_roll, device=dev)
    log(f"timed region done: {el:.3f} s for {args.steps} iterations")
    env_steps = world * total * args.steps
    updates = args.steps * args.n_epochs * ((total + batch - 1) // batch)
    train_stats = model.pop_train_stats()

    out = {
        "metric": "env_steps_per_sec", "value": env_steps / el, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mfma_dtype, "data": "synthetic",
        "config": {
            "workload": f"{args.task}, {N} envs/GPU, full PPO iterations: n_steps={T}, MLP pi/vf {args.hidden}x{args.hidden} tanh, "
                        f"n_epochs={args.n_epochs}, batch_size={batch} ({(total + batch - 1) // batch} minibatches/epoch), lr=3e-4, gamma=0.99, "
                        f"gae_lambda=0.95, clip=0.2, ent=0.01, vf=0.5, max_grad_norm=0.5",
            "envs_per_gpu": N, "n_steps": T, "batch_size": batch, "n_epochs": args.n_epochs, "hidden": args.hidden,
            "parallelism": f"dp{world} (envs sharded, all-reduce of the flat f32 gradient per minibatch)" if world > 1 else "single GPU",
        },
        "ppo_updates_per_sec": updates / max(el - t_roll, 1e-9), "ppo_updates_per_iteration": updates // args.steps,
        "rollout_env_steps_per_sec": env_steps / max(t_roll, 1e-9), "rollout_ms": t_roll / args.steps * 1e3,
        "update_ms": (el - t_roll) / args.steps * 1e3, "train_stats": {k: round(v, 6) for k, v in train_stats.items()},
    }

    if rank == 0:
        # ---- per-kernel durations with HIP events on the launch stream (rank 0, after the timed region) ----
        import ctypes as C

        eng = env.engine
        sync = lambda: torch.cuda.current_stream(dev).synchronize()  # noqa: E731
        b = model.buf
        L = _lib.lib()
        lay, sv = LAYOUT_BYTES.get(args.task, 0), SURVEY_BYTES.get(args.task, 0)
        flops_fwd, flops_fb = mlp_flops_per_sample(D, args.hidden, A)
        # ---- dominant kernel of the timed region: the PPO minibatch forward+backward (MFMA-bound) ----
        mb = _lib.Minibatch(None, 1, 0, 0, min(batch, total))

        def grad_once():
            _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(model.policy.params), C.byref(model.policy.dims), C.byref(model._rollout_view), C.byref(mb),
                                                C.byref(model._hp), _lib.ptr(model.grad), _lib.ptr(model.workspace), model._stream()))

        for _ in range(3):
            grad_once()
        g_avg, g_grp = timed_kernel_us(grad_once, 40, sync, group=4)  # whole launch group of the call (advantage pass + kernel + slab reduction)
        # the dominant kernel alone: HIP events recorded by the library around that launch, on the stream it is launched on
        ks = []
        if L.tma_debug_time_grad_kernel(1) == 0:
            us = C.c_float(0.0)
            for _ in range(24):
                grad_once()
                if L.tma_debug_last_grad_kernel_us(C.byref(us)) == 0:
                    ks.append(us.value)
            L.tma_debug_time_grad_kernel(0)
        g_med = sorted(ks)[len(ks) // 2] if ks else g_grp
        model.grad.zero_()
        log(f"minibatch gradient kernel: median {g_med:.1f} us (launch group of the call: {g_grp:.1f} us)")
        tf = mb.count * flops_fb / (g_med * 1e-6) / 1e12
        fast = args.hidden == 64 and D <= 16 and not model.policy.continuous and mb.count >= 16384
        bf = args.mfma_dtype == "bf16"
        wide = args.hidden in (128, 192, 256)
        peak = MFMA_BF16_PEAK_TFLOPS if bf else MFMA_F32_PEAK_TFLOPS
        kname = ("tma::ppo_grad_wide_bf_kernel" if bf else "tma::ppo_grad_h64_kernel" if fast else "tma::ppo_grad_wide_kernel" if wide else "tma::ppo_grad_kernel")
        out["roofline"] = {
            "kernel": kname, "launch_group_us": g_grp,
            "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
            "launch_us": g_med, "launch_us_source": "HIP events recorded by the library around the kernel launch on its stream (median of 24)" if ks else
                         "HIP events around the whole tma_ppo_minibatch_grad call",
            "samples_per_launch": int(mb.count), "flops_per_sample_fwd_bwd": flops_fb,
            "note": ("bf16-operand MFMA (v_mfma_f32_16x16x32_bf16, f32 accumulate; ~2.5 PFLOP/s dense peak)" if bf else
                     "exact-f32 MFMA (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s dense peak)") + "; flops = SURVEY.md 8d formula, fwd + bwd = 3 x fwd",
        }
        pmc_bf = os.path.join(ROOT, "profiles", "r01_gradbf_kernel_pmc.json")
        if bf and args.hidden == 256 and args.task == "ball3d" and os.path.exists(pmc_bf):
            pmc = json.load(open(pmc_bf))
            out["roofline"]["traffic"] = pmc.get("traffic_bytes_per_launch")
            out["roofline"]["traffic_source"] = "profiles/r01_gradbf_kernel_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2)"
        pmc_path = os.path.join(ROOT, "profiles", "r01_grad_kernel_pmc.json")
        if fast and args.task == "gridworld" and os.path.exists(pmc_path):
            pmc = json.load(open(pmc_path))
            out["roofline"]["traffic"] = pmc.get("traffic_bytes_per_launch")
            out["roofline"]["traffic_source"] = "profiles/r01_grad_kernel_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2)"
        # ---- batched env step kernel (the kernel north_star names), launched exactly as VecEnv.step does ----
        acts = b["actions"][0].contiguous()
        outs = dict(obs=torch.empty((1, N, D), device=dev), rew=torch.empty((1, N), device=dev), term=torch.empty((1, N), dtype=torch.uint8, device=dev),
                    trunc=torch.empty((1, N), dtype=torch.uint8, device=dev), term_obs=torch.empty((1, N, D), device=dev))

        def step_once():
            eng.step(acts, outputs=outs, want_episode=False)

        while eng.steps_until_refill() < eng.ring_depth:  # (tasks without an MT19937 reset never need a refill)
            step_once()
        reps = max(1, min(eng.ring_depth, 64) - 1)  # stays inside one refill window: only step kernels between the two events
        adt = _lib.ACT_F32 if model.policy.continuous else _lib.ACT_I32

        def step_burst():
            _lib.check(L.tma_env_step_repeat(eng._h, _lib.ptr(acts), adt, reps, _lib.ptr(outs["obs"]), _lib.ptr(outs["rew"]), _lib.ptr(outs["term"]),
                                             _lib.ptr(outs["trunc"]), _lib.ptr(outs["term_obs"]), model._stream()))
            step_once()  # closes the refill window (this launch and the refill are outside the timed burst)

        bursts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(L.tma_env_step_repeat(eng._h, _lib.ptr(acts), adt, reps, _lib.ptr(outs["obs"]), _lib.ptr(outs["rew"]), _lib.ptr(outs["term"]),
                                             _lib.ptr(outs["trunc"]), _lib.ptr(outs["term_obs"]), model._stream()))
            e1.record()
            step_once()
            sync()
            bursts.append(e0.elapsed_time(e1) * 1e3 / reps)
        bursts.sort()
        med_us = bursts[len(bursts) // 2]
        log(f"step kernel at {N} envs: median {med_us:.2f} us per launch ({reps} native back-to-back launches per burst)")
        step_gbps = N * lay / (med_us * 1e-6) / 1e9
        out["roofline_step_kernel"] = {
            "kernel": f"tma::step_kernel<{args.task}> (1 vector step, {N} envs, back-to-back launches)", "bound": "hbm", "achieved": step_gbps,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": step_gbps / HBM_PEAK_GBPS, "traffic": None, "bytes_per_env_step": lay, "launch_us": med_us,
            "survey_formula_bytes_per_env_step": sv, "survey_formula_GBps": N * sv / (med_us * 1e-6) / 1e9,
            "note": f"{N} envs move {N * lay / 1e6:.2f} MB per launch: launch-latency-bound, not HBM-bound (SURVEY.md 7.3-4); the training rollout uses the "
                    "fused multi-step kernel instead; see roofline_step_kernel_saturated for the HBM-bound regime",
        }
        # the same kernel where it is HBM-bound: 4M envs, 1 step per launch, device-generated action tape
        try:
            if world > 1:
                raise RuntimeError("skipped in multi-GPU runs (measured at N=1)")
            from three_mlagents_amd.vec_env import HipEnvEngine

            Nb = (1 << 22) if D <= 32 else (1 << 18)  # wide observations: fewer envs, still far beyond the L2/MALL capacity
            big = HipEnvEngine(args.task, Nb, seed=args.seed, ring_depth=8)
            big.reset()
            bo = {k: v for k, v in big._out(1).items() if k in ("obs", "rew", "term", "trunc")}
            tt = [0]

            def big_step():
                big.step(None, n_steps=1, tape_seed=1, tape_t0=tt[0], outputs=bo, want_terminal_obs=False, want_episode=False)
                tt[0] += 1

            for _ in range(8):
                big_step()
            _, med_big = timed_kernel_us(big_step, 40, sync)
            log(f"saturated step kernel median {med_big:.1f} us")
            bytes_big = lay - 4  # tape: no action read
            gb = Nb * bytes_big / (med_big * 1e-6) / 1e9
            sat = {"kernel": f"tma::step_kernel<{args.task}> (1 vector step, {Nb} envs, on-device action tape)", "bound": "hbm", "achieved": gb,
                   "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBPS, "launch_us": med_big, "bytes_per_env_step": bytes_big,
                   "env_steps_per_s_kernel_only": Nb / (med_big * 1e-6), "survey_formula_GBps": Nb * (sv - 4) / (med_big * 1e-6) / 1e9, "traffic": None}
            pmc_path = os.path.join(ROOT, "profiles", "r01_step_kernel_pmc.json")
            if args.task == "gridworld" and os.path.exists(pmc_path):
                pmc = json.load(open(pmc_path))
                sat["traffic"] = pmc.get("traffic_bytes_per_launch")
                sat["traffic_source"] = "profiles/r01_step_kernel_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2)"
            out["roofline_step_kernel_saturated"] = sat
            big.close()
            del big, bo
        except Exception as exc:  # noqa: BLE001
            out["roofline_step_kernel_saturated"] = {"error": str(exc)}
        if not args.no_cpu_baseline and world == 1:  # contract: CPU baseline on rank 0 at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": str(exc)}
        if args.sweep:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import env_sweep

            out["env_sweep"] = [env_sweep.run(args.task, n, 32, 3, pl) for n in (4096, 65536, 1 << 20, 1 << 22) for pl in (1, 32)]
        print(json.dumps(out), flush=True)
    env.close()
    dist.barrier()


if __name__ == "__main__":
    main()
