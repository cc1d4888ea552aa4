#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of full PPO iterations on GridWorld, 4096 envs per MI355X (BASELINE.json configs[1]).

One "step" = one complete PPO iteration of the hot path on one GPU:
    rollout of n_steps vector steps (policy forward + sampling, batched env step with auto-reset / terminal obs /
    Monitor sums, timeout bootstrap, MT19937 reset-ring refills) -> GAE -> n_epochs x minibatches of
    (advantage stats, forward+backward of the clipped-surrogate loss, [RCCL all-reduce of the flat gradient], clip + Adam).
Nothing is skipped inside the timed region.  `value` = env-steps of all ranks / max-over-ranks wall time.

Run:  python bench.py [--gpus N --steps K --warmup W]
  N > 1 started as a plain process: this file starts N ranks itself (a child `python -m torch.distributed.run`, one rank per GPU,
  before this process makes any GPU call) and exits with the child's status; started under torch.distributed.run it is one rank.
Prints ONE compact JSON line (< 4 KB) on rank 0: the contract keys, `roofline` (dominant kernel), `cpu_baseline` and one-number summaries of
the other legs.  The full objects -- (N = 1) `extra_configs`, the other single-GPU BASELINE.json configs, time-boxed; `literal_batch_256`, the
reference's literal batch_size on the headline shape; the harness leg; the per-config CPU table; the step / GAE / rollout rooflines -- go
to `bench_extras.json` beside this file (and to gpurun_out/ when present); the line's `extras_path` names it.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument("--mfma-dtype", default="f32", choices=["f32", "bf16", "bf16x3"],
                    help="MFMA operand type of the hidden-layer GEMMs (bf16: hidden 128/192/256 only; BASELINE.json configs[2])")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=36.0, help="CPU-baseline budget over all five configs x {1 thread, all cores}")
    ap.add_argument("--no-extras", action="store_true", help="skip extra_configs / literal_batch_256 (N = 1 default runs include them)")
    ap.add_argument("--sweep", action="store_true", help="also run the env-count sweep of the step kernel (extra JSON field)")
    return ap.parse_args()


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` as a plain process: start N fresh rank processes (torch.distributed.run, rendezvous on 127.0.0.1) and
    return the launcher's exit status.  Nothing here initialises the GPU: the ranks are children, never an exec of a process that
    holds a HIP context (torch.cuda.device_count() only counts devices)."""
    import torch

    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} needs {n} visible GPUs, this machine shows {have}; nothing was measured", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    log("starting ranks: " + " ".join(cmd))
    return subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))


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


# ------------------------------------------------------------------------------------------------------------------------
# GPU legs
# ------------------------------------------------------------------------------------------------------------------------
def build_model(task, n_envs, n_steps, hidden, mfma, batch, n_epochs, seed, dev, rank=0):
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    env = make_vector_env(task, n_envs=n_envs, seed=seed, device=dev, env_offset=rank * n_envs)
    model = PPO("MlpPolicy", env, learning_rate=3e-4, n_steps=n_steps, batch_size=batch, n_epochs=n_epochs, gamma=0.99, gae_lambda=0.95,
                clip_range=0.2, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, seed=seed,
                policy_kwargs={"net_arch": {"pi": [hidden] * 2, "vf": [hidden] * 2}, "mfma_dtype": mfma})
    return env, model


def time_iterations(model, steps, warmup):
    """`warmup` untimed + `steps` timed PPO iterations, bracketed by barrier + synchronize; (seconds, rollout seconds), max over ranks."""
    import torch

    from three_mlagents_amd import dist

    for _ in range(warmup):
        model.collect_rollouts()
        model.train()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    t_roll = 0.0
    for _ in range(steps):
        torch.cuda.synchronize()  # attribute the asynchronous tail of the previous update to the update, not to this rollout
        r0 = time.perf_counter()
        model.collect_rollouts()
        torch.cuda.synchronize()
        t_roll += time.perf_counter() - r0
        model.train()
    torch.cuda.synchronize()
    dist.barrier()
    el = time.perf_counter() - t0
    model._bench_local = (el, t_roll)  # this rank's own clock (dp_timing: per-rank rollout / update split)
    return dist.allreduce_max_float(el, device=model.device), dist.allreduce_max_float(t_roll, device=model.device)


def grad_kernel_roofline(model, task, hidden, mfma, batch, reps=24):
    """The dominant kernel of the update: persistent forward+backward of one minibatch.  Duration from HIP events the library records
    around that launch on the stream it launches on (tma_debug_time_grad_kernel); flops = SURVEY.md 8d formula x samples."""
    import ctypes as C

    import torch

    from three_mlagents_amd import _lib

    L = _lib.lib()
    dev = model.device
    total = model.n_steps * model.n_envs
    D, A = model.policy.obs_dim, model.policy.act_dim
    mb = _lib.Minibatch(None, 1, 0, 0, min(batch, total))

    def grad_once():
        _lib.check(L.tma_ppo_minibatch_grad(_lib.ptr(model.policy.params), C.byref(model.policy.dims), C.byref(model._rollout_view), C.byref(mb),
                                            C.byref(model._hp), _lib.ptr(model.grad), _lib.ptr(model.workspace), model._stream()))

    for _ in range(3):
        grad_once()
    _, g_grp = timed_kernel_us(grad_once, max(8, reps), lambda: torch.cuda.current_stream(dev).synchronize(), group=4)
    ks = []
    if L.tma_debug_time_grad_kernel(1) == 0:
        us = C.c_float(0.0)
        for _ in range(reps):
            grad_once()
            if L.tma_debug_last_grad_kernel_us(C.byref(us)) == 0:
                ks.append(us.value)
        L.tma_debug_time_grad_kernel(0)
    g_med = sorted(ks)[len(ks) // 2] if ks else g_grp
    model.grad.zero_()
    _, flops_fb = mlp_flops_per_sample(D, hidden, A)
    tf = mb.count * flops_fb / (g_med * 1e-6) / 1e12
    bf = mfma == "bf16"
    fast = hidden == 64 and D <= 16 and not model.policy.continuous and mb.count >= 256
    wide = hidden in (128, 192, 256)
    peak = MFMA_BF16_PEAK_TFLOPS if bf else MFMA_F32_PEAK_TFLOPS
    split = mfma == "bf16x3" and wide and mb.count >= 4096
    kname = ("tma::ppo_grad_split3_kernel" if split else "tma::ppo_grad_wide_bf_kernel" if bf else "tma::ppo_grad_h64_kernel" if fast else
             "tma::ppo_grad_wide_kernel" if wide else "tma::ppo_grad_kernel")
    f32_equiv = None
    if split:
        # the three-term split ISSUES bf16 MFMAs: six products on every chain GEMM (forward and input-gradient), three on the weight gradients
        # -- (6 + 6 + 3) / 3 = 5 x the f32-equivalent flops (csrc/tma_split3.h; the SQ counters of profiles/r06_gradsplit3_sq_pmc.json give the
        # measured count) -- and its roofline is the bf16 pipe's.  The f32-equivalent rate against the f32 peak stays as a second pair.
        f32_equiv = {"achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "frac": tf / MFMA_F32_PEAK_TFLOPS, "note": "f32-EQUIVALENT flops (SURVEY.md 8d formula) against the f32 MFMA peak: "
                     "what the update is worth, not a roofline of the pipe it runs on"}
        tf, peak = 5.0 * tf, MFMA_BF16_PEAK_TFLOPS
    return {
        "kernel": kname, "launch_group_us": g_grp, "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
        "f32_equivalent": f32_equiv,
        "launch_us": g_med, "launch_us_source": "HIP events recorded by the library around the kernel launch on its stream (median)" if ks else
                     "HIP events around the whole tma_ppo_minibatch_grad call",
        "samples_per_launch": int(mb.count), "flops_per_sample_fwd_bwd": flops_fb,
        # what follows the gradient kernel inside one tma_ppo_minibatch_grad call: the slab reduction launch and the boundary in front of it
        "slab_reduce_us": (g_grp - g_med) if ks else None,
        "note": ("three-term bf16 split of the f32 update (csrc/tma_split3.h): ISSUED bf16 MFMA flops = 5 x the f32-equivalent flops (six bf16 products per chain "
                 "GEMM, three per weight gradient) against the bf16 MFMA peak; `f32_equivalent` carries the rate the update is worth" if split else
                 "bf16-operand MFMA (v_mfma_f32_16x16x32_bf16, f32 accumulate; ~2.5 PFLOP/s dense peak)" if bf else
                 "exact-f32 MFMA (v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s dense peak)") + "; flops = SURVEY.md 8d formula, fwd + bwd = 3 x fwd",
    }


def kernel_spills(family):
    """Spilled VGPRs / scratch bytes of every instantiation of a kernel family, read from the AMDGPU metadata notes of the built objects
    (tools/kernel_regs.py: csrc/*.o travel with the tree).  None when the objects or llvm-readelf are not there."""
    try:
        import subprocess
        import sys as _sys

        _sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_regs as kr

        rows = []
        for obj in sorted(__import__("glob").glob(os.path.join(ROOT, "three-mlagents_amd", "csrc", "*.o"))):
            for elf in kr.device_elfs(obj):
                rows += list(kr.kernels(elf))
        names = subprocess.run([kr.CXXFILT], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
        out = {}
        for r, n in zip(rows, names):
            short = __import__("re").sub(r"\(.*", "", n).replace("void ", "").replace("tma::", "")
            if family.replace("tma::", "") in short and r["vgpr_spill"].isdigit():
                out[short] = {"spilled_vgprs": int(r["vgpr_spill"]), "scratch_bytes": int(r["scratch"]) if r["scratch"].isdigit() else None, "vgprs": int(r["vgpr"]) if r["vgpr"].isdigit() else None}
        return out or None
    except Exception:  # noqa: BLE001
        return None


def spill_summary(family, pick=None):
    """{family, instantiation (when `pick` names one), spilled_vgprs, max over the family}"""
    ks = kernel_spills(family)
    if not ks:
        return None
    mx = max(v["spilled_vgprs"] for v in ks.values())
    one = next((k for k in ks if pick and pick in k), None)
    return {"family": family, "instantiation": one, "spilled_vgprs": ks[one]["spilled_vgprs"] if one else None, "scratch_bytes": ks[one]["scratch_bytes"] if one else None,
            "family_max_spilled_vgprs": mx, "family_instantiations": len(ks)}


def attach_pmc_traffic(roof, name):
    """HBM bytes per launch from the committed rocprofv3 --pmc summaries (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2
    corrected as MI355X_MICROARCH.md prescribes); newest round first."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{name}_pmc.json")
        if os.path.exists(path):
            with open(path) as f:
                roof["traffic"] = json.load(f).get("traffic_bytes_per_launch")
            roof["traffic_source"] = f"profiles/{rnd}_{name}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2)"
            return


def step_kernel_rooflines(out, args, env, model, world):
    import ctypes as C

    import torch

    from three_mlagents_amd import _lib

    L = _lib.lib()
    eng, dev, b = env.engine, model.device, model.buf
    N, D = args.n_envs, model.policy.obs_dim
    sync = lambda: torch.cuda.current_stream(dev).synchronize()  # noqa: E731
    lay, sv = LAYOUT_BYTES.get(args.task, 0), SURVEY_BYTES.get(args.task, 0)
    acts = b["actions"][0].contiguous()
    outs = dict(obs=torch.empty((1, N, D), device=dev), rew=torch.empty((1, N), device=dev), term=torch.empty((1, N), dtype=torch.uint8, device=dev),
                trunc=torch.empty((1, N), dtype=torch.uint8, device=dev), term_obs=torch.empty((1, N, D), device=dev))

    def step_once():
        eng.step(acts, outputs=outs, want_episode=False)

    while eng.steps_until_refill() < eng.ring_depth:  # (tasks without an MT19937 reset never need a refill)
        step_once()
    reps = max(1, min(eng.ring_depth, 64) - 1)  # stays inside one refill window: only step kernels between the two events
    adt = _lib.ACT_F32 if model.policy.continuous else _lib.ACT_I32
    bursts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.tma_env_step_repeat(eng._h, _lib.ptr(acts), adt, reps, _lib.ptr(outs["obs"]), _lib.ptr(outs["rew"]), _lib.ptr(outs["term"]),
                                         _lib.ptr(outs["trunc"]), _lib.ptr(outs["term_obs"]), model._stream()))
        e1.record()
        step_once()  # closes the refill window (this launch and the refill are outside the timed burst)
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
    # the fused rollout chunk: policy + value forward, sampling, env step, buffer writes of one vector step (10 % of the headline iteration)
    try:
        T = model.n_steps
        b_rb = model._rb

        def collect_once():
            _lib.check(L.tma_rollout_collect(eng._h, _lib.ptr(model.policy.params), C.byref(model.policy.dims), C.byref(b_rb), 0, T, T, args.seed & 0xFFFFFFFF,
                                             (model._rollout_counter * T) & 0xFFFFFFFF, eng.env_offset & 0xFFFFFFFF, 0.99, 1, 0, model._stream()))
            model._rollout_counter += 1

        collect_once()
        _, roll_us = timed_kernel_us(collect_once, 6, sync)
        A_ = model.policy.act_dim
        fwd_flops, _ = mlp_flops_per_sample(D, args.hidden, A_)
        per_step_us = roll_us / T
        tf = N * fwd_flops / (per_step_us * 1e-6) / 1e12
        wbytes = N * (4 * D + 4 * (A_ if model.policy.continuous else 1) + 16)
        peak = MFMA_BF16_PEAK_TFLOPS if args.mfma_dtype == "bf16" else MFMA_F32_PEAK_TFLOPS
        fused = {64: "tma::rollout_chunk2_h64_kernel" if os.environ.get("TMA_ROLL2") else "tma::rollout_chunk4_h64_kernel", 256: "tma::rollout_chunk_wide_bf_kernel" if args.mfma_dtype == "bf16" else "tma::rollout_chunk_wide_f32_kernel"}
        out["roofline_rollout_kernel"] = {
            "kernel": f"{fused.get(args.hidden, 'per-step policy_fwd + step_kernel launches')}<{args.task}> (policy + value forward, sampling, env step with auto-reset, "
                      f"buffer writes; up to one reset-ring window = {eng.ring_depth} vector steps per launch)",
            "bound": "mfma", "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
            "us_per_vector_step": per_step_us, "collect_call_us": roll_us, "vector_steps_per_call": T, "flops_per_env_step_fwd": fwd_flops,
            "buffer_bytes_per_vector_step": wbytes, "buffer_write_GBps": wbytes / (per_step_us * 1e-6) / 1e9,
            "note": "HIP events around tma_rollout_collect over the whole rollout (chunk launches + MT19937 ring refills) / n_steps; flops = SURVEY.md 8d forward "
                    "formula (both nets) x envs, bytes = 8d's rollout-buffer write (4D + 4A' + 16) x envs.  A vector step is ONE 16-env tile per CU: each net on two "
                    "waves (round 6), a dependent chain of 4 + 32 + 16 MFMAs with a tanh between layers, then the action and the env step on one wave: bound by "
                    "that chain's latency (DESIGN.md section 5.1), not by the pipe or by HBM; the rocprofv3 mean of the chunk kernel is in "
                    "profiles/r06_bench_n1_kernel_stats.csv"}
        if args.hidden == 64 and args.task == "gridworld" and N == 4096:  # the committed counter summary is of this shape (tools/r06_rollout_pmc.sh)
            pmc = os.path.join(ROOT, "profiles", "r06_rollout_kernel_pmc.json")
            if os.path.exists(pmc):
                with open(pmc) as f:
                    out["roofline_rollout_kernel"]["traffic"] = json.load(f).get("traffic_bytes_per_vector_step")
                out["roofline_rollout_kernel"]["traffic_source"] = ("profiles/r06_rollout_kernel_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                                                                    "FETCH_SIZE x2; per vector step like buffer_bytes_per_vector_step)")
    except Exception as exc:  # noqa: BLE001
        out["roofline_rollout_kernel"] = {"error": repr(exc)}
    # GAE over the rollout that was just collected (20 B per (t, env) in SB3's layout; the engine's flag bytes make it 18)
    try:
        T = model.n_steps
        adv_o, ret_o = torch.empty_like(b["rewards"]), torch.empty_like(b["rewards"])

        def gae_once():
            _lib.check(L.tma_gae_flags(_lib.ptr(b["rewards"]), _lib.ptr(b["values"]), _lib.ptr(b["terminated"]), _lib.ptr(b["truncated"]),
                                       _lib.ptr(b["last_values"]), 0.99, 0.95, T, N, _lib.ptr(adv_o), _lib.ptr(ret_o), model._stream()))

        for _ in range(3):
            gae_once()
        _, gae_us = timed_kernel_us(gae_once, 20, sync)
        gae_bytes = T * N * 18
        gae_gbps = gae_bytes / (gae_us * 1e-6) / 1e9
        out["roofline_gae_kernel"] = {
            "kernel": "tma::gae_pc_kernel (producer / consumer chunks through LDS)" if N <= 65536 else "tma::gae_kernel", "bound": "hbm", "achieved": gae_gbps,
            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gae_gbps / HBM_PEAK_GBPS, "traffic": None, "launch_us": gae_us, "bytes_per_t_env": 18,
            "note": f"T = {T}, N = {N}: {gae_bytes / 1e6:.0f} MB per call; the recurrence is one rounded f32 chain per env (bit-identical to SB3's loop), so "
                    "at rollout sizes the bound is that chain (about 25 cycles per step on the chain wave), not HBM"}
        del adv_o, ret_o
    except Exception as exc:  # noqa: BLE001
        out["roofline_gae_kernel"] = {"error": str(exc)}
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
        if args.task == "gridworld":
            attach_pmc_traffic(sat, "step_kernel")
        out["roofline_step_kernel_saturated"] = sat
        big.close()
        del big, bo
    except Exception as exc:  # noqa: BLE001
        out["roofline_step_kernel_saturated"] = {"error": str(exc)}


# the other single-GPU BASELINE.json configs, at the per-GPU shard size their config names (configs[3]/[4] shard 8192/4 and 16384/8 envs
# = 2048 per GPU; T = 2048 for the benchmark tier, training.py:362), same PPO schedule as the headline (32 minibatches x 10 epochs)
EXTRA_CONFIGS = [
    dict(name="configs[2] Ball3D 4096 envs, MLP(256,256) bf16", task="ball3d", n_envs=4096, n_steps=1024, hidden=256, mfma="bf16"),
    dict(name="configs[3] Push 2048 envs/GPU (the per-GPU shard of 8192 over 4), MLP(256,256) bf16 (engine option: BASELINE.json does not name bf16 here)", task="push", n_envs=2048, n_steps=2048, hidden=256,
         mfma="bf16"),
    dict(name="configs[4] Crawler-shape 172/20, 2048 envs/GPU (the per-GPU shard of 16384 over 8), MLP(256,256) bf16 (engine option: BASELINE.json does not name bf16 here)", task="crawler", n_envs=2048,
         n_steps=2048, hidden=256, mfma="bf16"),
    dict(name="configs[0] Basic 8 envs, MLP(256,256) f32, the reference's literal batch 256", task="basic", n_envs=8, n_steps=1024, hidden=256, mfma="f32",
         batch=256),
    # SURVEY.md 8d config (2), second half: the headline env with the reference's DEFAULT net and dtype (training.py:363-365)
    dict(name="configs[1] GridWorld 4096 envs, MLP(256,256) f32 (reference default net)", task="gridworld", n_envs=4096, n_steps=1024, hidden=256, mfma="f32",
         steps=6, warmup=1),
    # BASELINE.json names bf16 for configs[2] only: the like-for-like f32 figures of the configs[3] / [4] shards (exact-f32 MFMA wide kernel)
    # ... and the same fp32 policy with the UPDATE on the bf16 MFMA as a three-term split (mfma_dtype "bf16x3", opt-in, f32-class accuracy: round 5)
    dict(name="configs[1] GridWorld 4096 envs, MLP(256,256) f32 weights, bf16x3 update (opt-in)", task="gridworld", n_envs=4096, n_steps=1024, hidden=256, mfma="bf16x3",
         steps=6, warmup=1),
    dict(name="configs[3] Push 2048 envs/GPU, MLP(256,256) f32 (the reference's dtype)", task="push", n_envs=2048, n_steps=2048, hidden=256, mfma="f32",
         steps=6, warmup=1),
    # the task the reference's registry actually binds to "crawler": gym Ant-v5, 105 observations / 8 torques (backend/mlagents/envs.py:274-277) --
    # build-defined chain dynamics at that shape (parity unpinned, as configs[4]); its literal schedule (8 envs, batch 256) and a 2048-env shard
    dict(name="reference 'ant' shape 105/8 (Ant-v5), 8 envs, MLP(256,256) f32, the reference's literal batch 256", task="ant", n_envs=8, n_steps=1024, hidden=256, mfma="f32",
         batch=256, steps=6, warmup=1),
    dict(name="reference 'ant' shape 105/8, 2048 envs/GPU, MLP(256,256) f32", task="ant", n_envs=2048, n_steps=2048, hidden=256, mfma="f32", steps=4, warmup=1),
    dict(name="configs[4] Crawler-shape 172/20, 2048 envs/GPU, MLP(256,256) f32 (the reference's dtype)", task="crawler", n_envs=2048, n_steps=2048, hidden=256,
         mfma="f32", steps=6, warmup=1),
    dict(name="configs[3] Push 2048 envs/GPU, MLP(256,256) f32 weights, bf16x3 update (opt-in)", task="push", n_envs=2048, n_steps=2048, hidden=256, mfma="bf16x3",
         steps=6, warmup=1),
]


def run_extra(cfg, args, dev):
    import torch

    total = cfg["n_envs"] * cfg["n_steps"]
    batch = cfg.get("batch") or max(256, total // 32)
    env, model = build_model(cfg["task"], cfg["n_envs"], cfg["n_steps"], cfg["hidden"], cfg["mfma"], batch, args.n_epochs, args.seed, dev)
    try:
        steps, warm = cfg.get("steps", 10), cfg.get("warmup", 3)  # (SURVEY.md 8d asks for warm-up + >= 10 timed iterations; each is 50-200 ms)
        el, t_roll = time_iterations(model, steps, warm)
        updates = steps * args.n_epochs * ((total + batch - 1) // batch)
        roof = grad_kernel_roofline(model, cfg["task"], cfg["hidden"], cfg["mfma"], batch, reps=12)
        if cfg["mfma"] == "bf16":
            attach_pmc_traffic(roof, {"ball3d": "gradbf_kernel", "push": "gradbf_push_kernel", "crawler": "gradbf_crawler_kernel"}.get(cfg["task"], "none"))
        elif cfg["task"] == "basic":
            attach_pmc_traffic(roof, "gradwide_basic_kernel")
        elif cfg["task"] == "gridworld" and cfg["hidden"] == 256:
            attach_pmc_traffic(roof, "gradwide_gridworld_kernel")
        res = {"config": cfg["name"], "task": cfg["task"], "envs_per_gpu": cfg["n_envs"], "n_steps": cfg["n_steps"], "hidden": cfg["hidden"],
               "dtype": cfg["mfma"], "batch_size": batch, "n_epochs": args.n_epochs, "steps": steps, "warmup": warm,
               "env_steps_per_sec": steps * total / el, "ms_per_step": el / steps * 1e3, "rollout_ms": t_roll / steps * 1e3,
               "update_ms": (el - t_roll) / steps * 1e3, "ppo_updates_per_sec": updates / max(el - t_roll, 1e-9),
               "roofline": {k: roof[k] for k in ("kernel", "launch_us", "achieved", "peak", "unit", "frac", "samples_per_launch", "traffic", "f32_equivalent")},
               "slab_reduce_us": roof.get("slab_reduce_us"), "grad_kernel_spilled_vgprs": spill_summary(roof["kernel"])}
        log(f"extra {cfg['name']}: {res['env_steps_per_sec'] / 1e6:.2f} M env-steps/s, {res['ms_per_step']:.1f} ms/iter, grad {roof['launch_us']:.0f} us")
        return res
    finally:
        env.close()
        del model, env
        torch.cuda.empty_cache()


def literal_batch_256(args, dev, hidden=None):
    """The reference's literal PPO schedule on the headline shape (training.py:379 batch_size=256 at 4096 envs x 1024 steps = 16 384
    optimizer steps per epoch): one full rollout + ONE epoch timed; the 10-epoch iteration is composed from them."""
    import torch

    if hidden is not None:  # the same schedule on another net width (the reference's default 256 x 256: per-minibatch launches)
        args = argparse.Namespace(**{**vars(args), "hidden": hidden})
    env, model = build_model(args.task, args.n_envs, args.n_steps, args.hidden, args.mfma_dtype, 256, 1, args.seed, dev)
    try:
        model.collect_rollouts()
        model.train()
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        model.collect_rollouts()
        torch.cuda.synchronize()
        t_roll = time.perf_counter() - r0
        model.train()
        torch.cuda.synchronize()
        t_epoch = time.perf_counter() - r0 - t_roll
        total = args.n_envs * args.n_steps
        n_mb = (total + 255) // 256
        st = model.pop_train_stats()
        return {"batch_size": 256, "optimizer_steps_per_epoch": n_mb, "epoch_ms": t_epoch * 1e3, "rollout_ms": t_roll * 1e3,
                "ppo_updates_per_sec": n_mb / t_epoch, "env_steps_per_sec_10_epochs_composed": total / (t_roll + args.n_epochs * t_epoch),
                "approx_kl": round(st["train/approx_kl"], 6), "us_per_optimizer_step": t_epoch / n_mb * 1e6,
                "update_path": ("per-minibatch launches (TMA_NO_PERSIST set)" if os.environ.get("TMA_NO_PERSIST") else
                                "persistent epoch kernel ppo_epoch_h64p_kernel (csrc/tma_h64p.hip): one launch per epoch")
                if args.hidden == 64 and args.mfma_dtype == "f32" else
                ("persistent epoch kernel ppo_epoch_h256p_kernel (csrc/tma_h256p.hip): one launch per epoch, 2 x 32 workgroups on two XCDs"
                 if args.hidden == 256 and args.mfma_dtype == "f32" and not os.environ.get("TMA_NO_PERSIST") and not model.policy.continuous
                 and model.policy.obs_dim <= 32 and model.policy.act_dim <= 16 else "per-minibatch launches"),
                # the step as a fraction of the f32 MFMA peak (SURVEY.md 8d flops; the whole step, exchanges included, not one kernel's launch)
                "mfma_f32_frac_of_step": 256 * mlp_flops_per_sample(model.policy.obs_dim, args.hidden, model.policy.act_dim)[1] / (t_epoch / n_mb) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                "workload": f"{args.task}, {args.n_envs} envs x {args.n_steps} steps, MLP {args.hidden}x{args.hidden}, batch_size 256 "
                            "(reference training.py:379), one epoch timed"}
    finally:
        env.close()
        del model, env
        torch.cuda.empty_cache()


def harness_train_task(args, dev):
    """The drop-in path end to end: wall time of harness.train_task (vector env + eval env construction, EvalCallback at the reference's
    `eval_freq // n_envs` cadence, Monitor rows, TensorBoard / progress files, policy zip, final evaluation, metadata.json) for two PPO
    iterations of the headline shape, next to the same two iterations through PPO directly (construction included on both sides)."""
    import shutil
    import tempfile

    import torch

    from three_mlagents_amd import harness

    batch = max(256, args.n_envs * args.n_steps // 32)
    pk = {"net_arch": {"pi": [args.hidden] * 2, "vf": [args.hidden] * 2}, "mfma_dtype": args.mfma_dtype}
    res = {}
    # pass 1: two iterations (warms every kernel module); pass 2: two iterations -- the figure the fixed costs of a run dominate (env + eval-env
    # construction, final evaluation, policy zip, metadata: ~20 ms next to 72 ms of training); pass 3: ten iterations, where they amortise
    for rep, iters in enumerate((2, 2, 10)):
        total = iters * args.n_envs * args.n_steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env, model = build_model(args.task, args.n_envs, args.n_steps, args.hidden, args.mfma_dtype, batch, args.n_epochs, args.seed, dev)
        model.learn(total)
        torch.cuda.synchronize()
        t_direct = time.perf_counter() - t0
        env.close()
        del model, env
        tmp = tempfile.mkdtemp(prefix="tma_bench_harness_")
        cwd = os.getcwd()
        try:
            os.chdir(tmp)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cfg = harness.TrainConfig(args.task, total_timesteps=total, n_envs=args.n_envs, seed=args.seed, run_name="bench", verbose=0)
            out = harness.train_task(cfg, model_kwargs={"batch_size": batch, "n_epochs": args.n_epochs, "n_steps": args.n_steps, "policy_kwargs": pk})
            torch.cuda.synchronize()
            t_harness = time.perf_counter() - t0
            rows = 0
            mon = os.path.join(tmp, "runs", harness.tasks.resolve(args.task).id, "bench", "monitor", "0.monitor.csv")
            if os.path.exists(mon):
                with open(mon) as f:
                    rows = sum(1 for ln in f if ln[:1] not in "#r")
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp, ignore_errors=True)
        leg = {"iterations": iters, "total_timesteps": total, "train_task_seconds": t_harness, "direct_ppo_seconds": t_direct,
               "overhead_frac": t_harness / t_direct - 1.0, "env_steps_per_sec_train_task": total / t_harness,
               "env_steps_per_sec_direct": total / t_direct, "monitor_rows_written": rows, "mean_reward": out.mean_reward,
               "eval_episodes": out.eval_episodes}
        if rep == 1:
            res = {"workload": f"train_task({args.task}, n_envs={args.n_envs}, total_timesteps={total}, batch_size={batch}, MLP {args.hidden}x{args.hidden}): PPO "
                               "iterations + EvalCallback (eval_freq 10000 // n_envs vector steps, 100 episodes on a 128-env device eval vector, queued on a parameter snapshot and collected under the next update) + Monitor rows + tb / "
                               "progress files + policy zip + final evaluation + metadata.json, next to the same iterations through PPO directly (construction "
                               "included on both sides)", **leg}
        elif rep == 2:
            res["ten_iterations"] = leg
            res["steady_state_overhead_frac"] = ((leg["train_task_seconds"] - res["train_task_seconds"]) /
                                                 max(leg["direct_ppo_seconds"] - res["direct_ppo_seconds"], 1e-9)) - 1.0
    return res


def threshold_leg():
    """The reference's declared reward thresholds (registry.py:64,80,96,112,128) through harness.train_task, on the reference's own schedule
    (`literal`: its n_envs, total_timesteps, batch 256) and at 4096 envs (`scaled`): final deterministic evaluation, timesteps and seconds to the
    first evaluation at the threshold, wall time of the whole train_task call (tools/threshold_runs.py; tests/test_thresholds_gpu.py asserts them)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import threshold_runs

    res = {}
    for task in ("basic", "gridworld", "ball3d", "push", "walljump"):
        for sched in ("literal", "scaled"):
            r = threshold_runs.run(task, sched, seed=1)
            r.pop("eval_curve", None)
            if (task, sched) == ("gridworld", "literal"):
                # the one case whose FINAL evaluation sits at its threshold (PPO with one env on a task whose reference default is DQN; 100
                # episodes with a standard deviation of 0.6): reported -- and asserted in tests/test_thresholds_gpu.py -- as the median of five seeds
                finals = {1: r["final_eval_mean"]}
                for seed in (2, 3, 4, 5):
                    finals[seed] = threshold_runs.run(task, sched, seed=seed)["final_eval_mean"]
                med = sorted(finals.values())[2]
                r.update({"final_eval_mean_by_seed": finals, "final_eval_mean_seed_1": r["final_eval_mean"], "final_eval_mean": med, "reached": bool(med >= r["threshold"]),
                          "final_eval_note": "median of seeds 1..5 (the other fields are seed 1's run)"})
            res.setdefault(task, {})[sched] = r
            f = r.get("first_eval_at_threshold") or {}
            log(f"threshold {task} {sched}: final {r['final_eval_mean']:.3f}{' (median of five seeds: ' + ', '.join(f'{v:.3f}' for v in r['final_eval_mean_by_seed'].values()) + ')' if 'final_eval_mean_by_seed' in r else ''}"
                f" (threshold {r['threshold']}), first at {f.get('timesteps')} steps / {f.get('device_seconds')} s, train_task {r['train_task_wall_seconds']:.2f} s")
    return res


# ------------------------------------------------------------------------------------------------------------------------
# CPU leg (the oracle as the stated baseline; never the thing shipped)
# ------------------------------------------------------------------------------------------------------------------------
# the five BASELINE.json configs as the CPU legs see them (per-GPU shard sizes as in EXTRA_CONFIGS; the reference's default net 256x256 for
# everything but the headline, which BASELINE.json names at 64x64); batch 0 -> the GPU leg's schedule (32 minibatches per epoch)
CPU_CONFIGS = [
    dict(name="configs[0] Basic 8 envs, MLP(256,256), batch 256 (the reference's literal config)", task="basic", n_envs=8, n_steps=1024, hidden=256, batch=256),
    dict(name="configs[1] GridWorld 4096 envs, MLP(64,64) (headline)", task="gridworld", n_envs=4096, n_steps=1024, hidden=64, batch=0, headline=True),
    dict(name="configs[2] Ball3D 4096 envs, MLP(256,256)", task="ball3d", n_envs=4096, n_steps=1024, hidden=256, batch=0),
    dict(name="configs[3] Push 2048 envs (one GPU's shard of 8192), T=2048, MLP(256,256)", task="push", n_envs=2048, n_steps=2048, hidden=256, batch=0),
    dict(name="configs[4] Crawler-shape 172/20, 2048 envs (one GPU's shard of 16384), T=2048, MLP(256,256)", task="crawler", n_envs=2048, n_steps=2048,
         hidden=256, batch=0),
]


def _cpu_legs(cfg, threads, seconds, n_epochs, seed):
    """One config on `threads` host threads (C oracle env: OpenMP over envs; torch-CPU restatement of SB3's policy / loss / Adam:
    torch.set_num_threads), each leg time-boxed to about a quarter of `seconds`:
      env_only  vector steps on a fixed action tape (auto-reset, Monitor sums, terminal observations included);
      env_gae   the same steps + GAE over them (no policy);
      rollout   policy forward + sampling + env step, then GAE -- what collect_rollouts does;
      update    optimizer steps (forward, loss, backward, clip, Adam) on minibatches drawn from that rollout, scaled to the config's batch size.
    `env_steps_per_s` composes rollout + n_epochs x minibatches x update into full PPO iterations (the metric of the GPU leg)."""
    import numpy as np
    import torch

    from oracle import oracle as orc
    from oracle import sb3_ref

    torch.set_num_threads(threads)
    task, N, T, H = cfg["task"], cfg["n_envs"], cfg["n_steps"], cfg["hidden"]
    D, A = orc.obs_dim(task), orc.num_actions(task)
    cont = A == 0
    Ad = orc.act_dim(task) if cont else A
    total = N * T
    batch = cfg["batch"] or max(256, total // 32)
    n_mb = (total + batch - 1) // batch
    leg = max(0.3, seconds / 4.0)
    env = orc.OracleVecEnv(task, N, seed=seed, threads=threads)
    env.reset()
    rng = np.random.default_rng(seed)
    # ---- env only (+ GAE) ----
    Tmax = T
    tape = (rng.uniform(-1.0, 1.0, size=(min(Tmax, 64), N, Ad)).astype(np.float32) if cont else orc.action_tape(seed + 2, N, Tmax, A))
    obs = np.zeros((N, D), np.float32)
    rew_all, term_all, trunc_all = np.zeros((Tmax, N), np.float32), np.zeros((Tmax, N), np.uint8), np.zeros((Tmax, N), np.uint8)
    t0 = time.perf_counter()
    n_env = 0
    while n_env < Tmax and (time.perf_counter() - t0 < leg or n_env < 2):
        env.step_fast(tape[n_env % len(tape)], obs, rew_all[n_env], term_all[n_env], trunc_all[n_env])  # (rows are contiguous views: no copies)
        n_env += 1
    t_env = time.perf_counter() - t0
    rew, done = rew_all[:n_env], (term_all[:n_env] | trunc_all[:n_env]).astype(np.float32)
    val = rng.normal(size=rew.shape).astype(np.float32)
    es = np.concatenate([np.zeros((1, N), np.float32), done[:-1]])
    g0 = time.perf_counter()
    orc.gae(rew, val, es, val[-1], done[-1].astype(np.uint8), threads=threads)
    t_gae = time.perf_counter() - g0
    # ---- rollout: policy forward + sampling + env step, then GAE ----
    sd = sb3_ref.init_policy(D, H, Ad, cont, seed=seed)
    ob = torch.from_numpy(env.reset())
    b_obs, b_act, b_lp, b_val, b_rew, b_done = [], [], [], [], [], []
    Tneed = max(2, min(T, -(-min(batch, 16384) // N)))  # enough rows for the update leg's sample
    t0 = time.perf_counter()
    n_roll = 0
    while n_roll < T and (time.perf_counter() - t0 < leg or n_roll < Tneed):
        with torch.no_grad():
            out, values = sb3_ref.forward(sd, ob)
            dist = (torch.distributions.Normal(out, torch.ones_like(out) * sd["log_std"].exp()) if cont else torch.distributions.Categorical(logits=out))
            act = dist.sample()
            lp = dist.log_prob(act).sum(dim=1) if cont else dist.log_prob(act)
        o = env.step(np.clip(act.numpy(), -1.0, 1.0).astype(np.float32) if cont else act.numpy().astype(np.int32))
        b_obs.append(ob), b_act.append(act), b_lp.append(lp), b_val.append(values), b_rew.append(torch.from_numpy(o["rew32"]))
        b_done.append(torch.from_numpy((o["term"] | o["trunc"]).astype(np.float32)))
        ob = torch.from_numpy(o["obs"])
        n_roll += 1
    with torch.no_grad():
        _, last_v = sb3_ref.forward(sd, ob)
    rew, val, done = torch.stack(b_rew).numpy(), torch.stack(b_val).numpy(), torch.stack(b_done).numpy()
    es = np.concatenate([np.zeros((1, N), np.float32), done[:-1]])
    adv, ret = orc.gae(rew, val, es, last_v.numpy(), done[-1].astype(np.uint8), threads=threads)
    t_roll = time.perf_counter() - t0
    roll_per_env_step = t_roll / (n_roll * N)
    # ---- update: optimizer steps on minibatches of the rollout sample, scaled linearly to the config's batch size ----
    tr = sb3_ref.RefTrainer(sd)
    fo, fa, fl = torch.cat(b_obs), torch.cat(b_act), torch.cat(b_lp)
    fadv, fret = torch.from_numpy(adv).reshape(-1), torch.from_numpy(ret).reshape(-1)
    have = fo.shape[0]
    bs = min(batch, have, 16384 if threads > 1 else 4096)
    n_upd, t1 = 0, time.perf_counter()
    while True:
        idx = torch.randperm(have)[:bs]
        tr.step(fo[idx], fa[idx], fl[idx], fadv[idx], fret[idx], clip_range=0.2, ent_coef=0.01, vf_coef=0.5)
        n_upd += 1
        if time.perf_counter() - t1 > leg or n_upd >= n_epochs * n_mb:
            break
    t_upd = (time.perf_counter() - t1) / n_upd * (batch / bs)
    iter_s = total * roll_per_env_step + n_epochs * n_mb * t_upd
    del env
    return {"threads": threads, "env_only_steps_per_s": n_env * N / t_env, "env_gae_steps_per_s": n_env * N / (t_env + t_gae),
            "rollout_env_steps_per_s": 1.0 / roll_per_env_step, "ppo_updates_per_sec": 1.0 / t_upd, "update_ms_per_minibatch": t_upd * 1e3,
            "env_steps_per_s": total / iter_s, "batch_size": batch, "minibatches_per_epoch": n_mb,
            "update_minibatch_samples_timed": bs, "update_scale_factor": batch / bs,  # (t_update is measured at bs samples and scaled linearly to `batch`)
            "sample": f"env-only {n_env} x {N} steps ({t_env:.2f} s) + GAE ({t_gae * 1e3:.1f} ms); rollout {n_roll} x {N} steps incl. policy + GAE ({t_roll:.2f} s); "
                      f"{n_upd} optimizer steps on {bs}-sample minibatches, scaled x{batch / bs:.0f} to batch {batch}"}


def cpu_baseline(args, seconds, batch):
    """The same PPO iterations on the host cores of this box: the build's C restatement of the reference's envs
    (oracle/tma_oracle.c, bit-exact against fixtures generated from the reference) + the torch-CPU restatement of SB3's policy / GAE / update
    (oracle/sb3_ref.py), for each of the five BASELINE.json configs, (a) on ONE thread, (b) on 32 threads and (c) on every logical CPU of the box.  A full
    iteration would take minutes, so every leg is a bounded sample of the same workload (see `sample` in each entry) and full iterations are
    composed from the legs with the GPU leg's schedule.  `value` is the headline config on all cores."""
    cores_all = os.cpu_count() or 1
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    mid = min(32, cores_all)  # a pool the size of one CCD group: what the small legs usually want on a many-core host
    log(f"cpu_baseline: 1, {mid} and {cores_all} threads, budget {seconds:.0f} s over {len(CPU_CONFIGS)} configs")
    weights = [0.3 if c.get("headline") else 0.7 / (len(CPU_CONFIGS) - 1) for c in CPU_CONFIGS]
    configs, head = [], None
    for cfg, w in zip(CPU_CONFIGS, weights):
        if cfg.get("headline"):  # the headline is whatever this run measured on the GPU (defaults = BASELINE configs[1])
            cfg = dict(cfg, task=args.task, n_envs=args.n_envs, n_steps=args.n_steps, hidden=args.hidden, batch=batch)
        entry = {"config": cfg["name"], "task": cfg["task"], "envs": cfg["n_envs"], "n_steps": cfg["n_steps"], "hidden": cfg["hidden"]}
        try:
            entry["single_thread"] = _cpu_legs(cfg, 1, seconds * w / 3, args.n_epochs, args.seed)
            if mid not in (1, cores_all):
                entry[f"threads_{mid}"] = _cpu_legs(cfg, mid, seconds * w / 3, args.n_epochs, args.seed)
            entry["all_cores"] = _cpu_legs(cfg, cores_all, seconds * w / 3, args.n_epochs, args.seed)
            legs = [v for k, v in entry.items() if isinstance(v, dict) and "env_steps_per_s" in v]
            entry["best"] = max(legs, key=lambda v: v["env_steps_per_s"])["threads"]
            log(f"cpu {cfg['task']}: " + ", ".join(f"{v['threads']} thr {v['env_steps_per_s']:.0f}" for v in legs) + " env-steps/s; env-only " +
                ", ".join(f"{v['env_only_steps_per_s']:.3g}" for v in legs))
        except Exception as exc:  # noqa: BLE001
            entry["error"] = repr(exc)
        configs.append(entry)
        if cfg.get("headline"):
            head = entry
    # `value`: the headline config at the thread count that served it best (on a 256-thread host the small legs LOSE to one thread when every
    # OpenMP / torch pool is 256 wide: the all-cores figures are reported as measured, not as the baseline's best)
    legs = [v for v in (head or {}).values() if isinstance(v, dict) and "env_steps_per_s" in v]
    a = max(legs, key=lambda v: v["env_steps_per_s"]) if legs else {}
    one = (head or {}).get("single_thread") or {}
    return {
        "value": a.get("env_steps_per_s"), "unit": "env-steps/s", "cores": a.get("threads"), "kind": "port", "cpu_model": cpu_model, "host_logical_cpus": cores_all,
        "single_thread_value": one.get("env_steps_per_s"), "all_cores_value": ((head or {}).get("all_cores") or {}).get("env_steps_per_s"),
        "reference_python_calibration": "BASELINE.md section 3 (measured in the survey container, 2.6 GHz Xeon, one core): the reference's own Python envs run "
                                        "62-64 k raw env.step()/s/core for GridWorld (no policy, no VecEnv); through SB3's DummyVecEnv + PPO the reference "
                                        "trains at about 1-2 k env-steps/s.  The C port timed here is the build's restatement, not the reference's Python: "
                                        "its single-thread env-only rate (configs[*].single_thread.env_only_steps_per_s) is what relates to the 62-64 k figure",
        "extrapolated": True, "update_scale_factor": a.get("update_scale_factor"), "update_minibatch_samples_timed": a.get("update_minibatch_samples_timed"),
        "sample": f"extrapolated: every leg is a time-boxed sample and the update leg is timed on {a.get('update_minibatch_samples_timed')}-sample minibatches and scaled x{a.get('update_scale_factor') or 1:.0f} "
                  f"to the schedule's batch; headline: {args.task}, {args.n_envs} envs x {args.n_steps} steps, MLP {args.hidden}x{args.hidden}, {args.n_epochs} epochs x minibatches of {batch}; "
                  f"{a.get('threads')} of {cores_all} logical CPUs (the best of 1 / {mid} / {cores_all} threads); " + str(a.get("sample")) + "; value = n_envs*n_steps / (n_envs*n_steps*t_rollout_per_env_step + n_epochs*n_minibatches*t_update)",
        "rollout_env_steps_per_s": a.get("rollout_env_steps_per_s"), "env_only_steps_per_s": a.get("env_only_steps_per_s"),
        "env_gae_steps_per_s": a.get("env_gae_steps_per_s"), "update_ms_per_minibatch": a.get("update_ms_per_minibatch"),
        "ppo_updates_per_sec": a.get("ppo_updates_per_sec"), "configs": configs,
    }

# ------------------------------------------------------------------------------------------------------------------------
# output: ONE compact JSON line (the driver keeps only the last ~8 KB of stdout) + everything else in a side file
# ------------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096


def _r(v, sig=5):
    """Floats to `sig` significant digits (the line is a record, not a checksum); containers recursively."""
    if isinstance(v, float):
        return float(f"{v:.{sig}g}") if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _roof_short(r, extra=()):
    if not isinstance(r, dict):
        return None
    if "error" in r:
        return {"error": str(r["error"])[:120]}
    o = _pick(r, ("kernel", "bound", "launch_us", "achieved", "peak", "unit", "frac", "traffic", *extra))
    if "kernel" in o:
        o["kernel"] = o["kernel"].split(" (")[0][:64]
    if r.get("traffic_source"):
        o["traffic_source"] = r["traffic_source"].split(" (")[0][:64]
    return o


def write_extras(out):
    """Everything the line leaves out (extra configs, literal-batch legs, harness timing, the CPU table, the per-kernel notes), as
    indented JSON beside bench.py -- and under gpurun_out/ when that scratch directory exists, so a gpurun call brings it home."""
    name = os.environ.get("TMA_BENCH_EXTRAS", "bench_extras.json" if out.get("n_gpus", 1) == 1 else f"bench_extras_n{out.get('n_gpus')}.json")
    written = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, os.path.basename(name)), "w") as f:
                json.dump(out, f, indent=1)
            written = written or os.path.relpath(os.path.join(d, os.path.basename(name)), ROOT)
        except OSError as exc:
            log(f"could not write {d}/{name}: {exc}")
    return written


def compact_line(out, extras_path=None, limit=LINE_LIMIT):
    """The contract's one JSON line, bounded to `limit` bytes: the contract keys, `config`, `roofline` (dominant kernel), `cpu_baseline`,
    and one-number summaries of the other legs; the full objects live in `extras_path`.  Optional blocks are dropped last-first if a
    run's strings ever push the line over the limit (the contract keys, roofline and cpu_baseline are never dropped)."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = out.get("config", {})
    line["config"] = {**_pick(cfg, ("envs_per_gpu", "n_steps", "batch_size", "n_epochs", "hidden")),
                      "workload": str(cfg.get("workload", ""))[:260], "parallelism": str(cfg.get("parallelism", "")).split(":")[0][:40]}
    line.update(_pick(out, ("ppo_updates_per_sec", "rollout_ms", "update_ms")))
    line["roofline"] = _roof_short(out.get("roofline"), ("samples_per_launch", "flops_per_sample_fwd_bwd"))
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = ({"error": str(cb["error"])[:160]} if "error" in cb else
                                {**_pick(cb, ("value", "unit", "cores", "kind", "cpu_model", "host_logical_cpus", "single_thread_value", "all_cores_value",
                                              "env_only_steps_per_s")), "sample": str(cb.get("sample", ""))[:200]})
    optional = []  # (key, value) in the order they are dropped if the line is too long: last first
    if isinstance(out.get("dp_timing"), dict):
        dp = out["dp_timing"]

        def med(x):
            return _pick(x, ("calls_timed", "median_us", "max_us", "bytes")) if isinstance(x, dict) else x

        optional.append(("dp_timing", {"backend": dp.get("backend"), "allreduce_path": str(dp.get("allreduce_path", "")).split(" (")[0], "peer_exchange": (str(dp["peer_exchange"])[:160] if dp.get("peer_exchange") else None),
                                       "grad_allreduce": med(dp.get("grad_allreduce")), "adv_sums_allreduce": med(dp.get("adv_sums_allreduce")),
                                       "grad_allreduces_per_iteration": dp.get("grad_allreduces_per_iteration"),
                                       "adv_sums_allreduces_per_iteration": dp.get("adv_sums_allreduces_per_iteration"),
                                       "per_rank_update_ms": dp.get("per_rank_update_ms"), "per_rank_rollout_ms": dp.get("per_rank_rollout_ms")}))
    roofs = {}
    for key, short in (("roofline_step_kernel_saturated", "step_saturated"), ("roofline_step_kernel", "step_4096"), ("roofline_gae_kernel", "gae"),
                       ("roofline_rollout_kernel", "rollout")):
        r = out.get(key)
        if isinstance(r, dict):
            roofs[short] = ({"error": str(r["error"])[:80]} if "error" in r else
                            {k: v for k, v in _pick(r, ("bound", "launch_us", "us_per_vector_step", "achieved", "unit", "frac", "traffic")).items() if v is not None})
    if roofs:
        optional.append(("other_rooflines", roofs))
    if isinstance(out.get("extra_configs"), list):
        rows = {"_": "[task, envs/GPU, hidden, dtype, env_steps_per_sec, ms_per_step, grad_kernel_us, grad_roofline_frac]"}
        for e in out["extra_configs"]:
            key = f"{str(e.get('config', '?'))[:10]} {e.get('task', '')} {e.get('hidden', '')} {e.get('dtype', '')}".strip()
            rows[key] = ("error: " + str(e["error"])[:60]) if "error" in e else [
                e.get("task"), e.get("envs_per_gpu"), e.get("hidden"), e.get("dtype"), e.get("env_steps_per_sec"), e.get("ms_per_step"),
                (e.get("roofline") or {}).get("launch_us"), (e.get("roofline") or {}).get("frac")]
        optional.append(("extra_configs", rows))
    lit = {}
    for key, short in (("literal_batch_256", "h64"), ("literal_batch_256_h256", "h256")):
        r = out.get(key)
        if isinstance(r, dict):
            lit[short] = {"error": str(r["error"])[:80]} if "error" in r else _pick(r, ("ppo_updates_per_sec", "us_per_optimizer_step", "env_steps_per_sec_10_epochs_composed"))
    if lit:
        optional.append(("literal_batch_256", lit))
    h = out.get("harness_train_task")
    if isinstance(h, dict):
        optional.append(("harness_train_task", {"error": str(h["error"])[:80]} if "error" in h else
                         _pick(h, ("train_task_seconds", "direct_ppo_seconds", "overhead_frac", "steady_state_overhead_frac"))))
    th = out.get("thresholds")
    if isinstance(th, dict):  # per task: [threshold, final eval literal, final eval scaled, seconds to threshold literal, ... scaled]
        optional.append(("thresholds", {"error": str(th["error"])[:80]} if "error" in th else {
            "_": "task: [threshold, final_literal, final_scaled, s_to_threshold_literal, s_to_threshold_scaled]",
            **{t: [(v.get("literal") or {}).get("threshold"), (v.get("literal") or {}).get("final_eval_mean"), (v.get("scaled") or {}).get("final_eval_mean"),
                   ((v.get("literal") or {}).get("first_eval_at_threshold") or {}).get("device_seconds"),
                   ((v.get("scaled") or {}).get("first_eval_at_threshold") or {}).get("device_seconds")] for t, v in th.items() if isinstance(v, dict)}}))
    if isinstance(out.get("train_stats"), dict):
        optional.append(("train_stats", _pick(out["train_stats"], ("train/approx_kl", "train/clip_fraction", "train/explained_variance"))))
    line["extras_path"] = extras_path
    for k, v in optional:
        line[k] = _r(v, 4)  # (summaries: four digits; the side file has the full values)
    exact = {k: line[k] for k in ("value", "ms_per_step") if k in line}  # (a reader checks value against ms_per_step: these two are not rounded)

    def dump():
        return json.dumps({**_r(line), **exact}, separators=(",", ":"))

    text = dump()
    while len(text) >= limit and optional:
        k, _ = optional.pop()
        line.pop(k, None)
        line["dropped_to_fit"] = line.get("dropped_to_fit", []) + [k]
        text = dump()
    return text


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))  # before anything touches the GPU in this process
    import torch

    from three_mlagents_amd import dist

    rank, local_rank, world = dist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start it as `python bench.py --gpus N` or under "
                         f"`torch.distributed.run --nproc-per-node N ... bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    if os.environ.get("TMA_BENCH_ONE_DEVICE"):  # test hook (with TMA_DIST_BACKEND=gloo): every rank on device 0 of a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    N, T = args.n_envs, args.n_steps
    total = N * T
    batch = args.batch_size if args.batch_size > 0 else max(256, total // 32)
    env, model = build_model(args.task, N, T, args.hidden, args.mfma_dtype, batch, args.n_epochs, args.seed, dev, rank)
    log(f"rank {rank}/{world}: engine ready, N={N} T={T} batch={batch}")
    if world > 1:
        model.dp_timing = {}  # HIP events around the collectives of the first minibatches / epochs of every train() call
        if model._native_comm is not None:  # (native RCCL path: the library brackets its own ncclAllReduce calls)
            model._native_comm.timing(model.dp_timing_samples)
    el, t_roll = time_iterations(model, args.steps, args.warmup)
    log(f"timed region done: {el:.3f} s for {args.steps} iterations")
    env_steps = world * total * args.steps
    updates = args.steps * args.n_epochs * ((total + batch - 1) // batch)
    train_stats = model.pop_train_stats()

    out = {
        "metric": "env_steps_per_sec", "value": env_steps / el, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mfma_dtype, "data": "synthetic",
        "config": {
            "workload": f"{args.task}, {N} envs/GPU, full PPO iterations: n_steps={T}, MLP pi/vf {args.hidden}x{args.hidden} tanh, "
                        f"n_epochs={args.n_epochs}, batch_size={batch}/GPU ({(total + batch - 1) // batch} minibatches/epoch), lr=3e-4, gamma=0.99, "
                        f"gae_lambda=0.95, clip=0.2, ent=0.01, vf=0.5, max_grad_norm=0.5",
            "envs_per_gpu": N, "n_steps": T, "batch_size": batch, "n_epochs": args.n_epochs, "hidden": args.hidden,
            "parallelism": (f"dp{world}: envs sharded by contiguous blocks, per minibatch one all-reduce of the flat f32 gradient (RCCL, or the peer "
                            f"exchange over xGMI when it is the faster one: dp_timing.allreduce_path), per epoch one all-reduce of the minibatches' "
                            f"advantage (sum, sumsq) pairs") if world > 1 else "single GPU",
        },
        "ppo_updates_per_sec": updates / max(el - t_roll, 1e-9), "ppo_updates_per_iteration": updates // args.steps,
        "rollout_env_steps_per_sec": env_steps / max(t_roll, 1e-9), "rollout_ms": t_roll / args.steps * 1e3,
        "update_ms": (el - t_roll) / args.steps * 1e3, "train_stats": {k: round(v, 6) for k, v in train_stats.items()},
    }

    if world > 1:
        # what a multi-GPU number is made of: per-collective HIP-event durations on rank 0 and every rank's own rollout / update split
        import torch.distributed as td

        mine = torch.tensor([model._bench_local[1] / args.steps * 1e3, (model._bench_local[0] - model._bench_local[1]) / args.steps * 1e3],
                            dtype=torch.float64, device=dev)
        per_rank = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(per_rank, mine)
        timing = model.dp_timing_collect()
        model.dp_timing = None
        out["dp_timing"] = {
            "backend": td.get_backend(),
            "allreduce_path": ("native peer exchange (libtma_hip.so: slab_reduce_kernel stores the reduced gradient straight into every rank's inbox over xGMI, the "
                               "sum-of-squares pass adds the ranks' words in rank order -- no collective launch in the minibatch chain)"
                               if model._native_comm is not None and model._native_comm.p2p_enabled else
                               "native (libtma_hip.so: ncclAllReduce on the compute stream from inside tma_ppo_train_epoch_dp)"
                               if model._native_comm is not None else "callback (ctypes -> Python -> torch.distributed.all_reduce)"),
            "peer_exchange": getattr(model._native_comm, "p2p_note", None) if model._native_comm is not None else None,
            "grad_allreduce": timing["grad_allreduce_us"], "adv_sums_allreduce": timing["adv_allreduce_us"],
            "grad_allreduces_per_iteration": updates // args.steps, "adv_sums_allreduces_per_iteration": args.n_epochs,
            "per_rank_rollout_ms": [float(t[0]) for t in per_rank], "per_rank_update_ms": [float(t[1]) for t in per_rank],
            "note": "durations are HIP events on rank 0's compute stream: from the end of the kernel that produced the tensor to the point where the "
                    "reduced tensor is usable (collective + both stream hand-offs); update_ms - n * median is what the GPU spent in kernels",
        }
    if rank == 0:
        roof = grad_kernel_roofline(model, args.task, args.hidden, args.mfma_dtype, batch)
        log(f"minibatch gradient kernel: median {roof['launch_us']:.1f} us (launch group of the call: {roof['launch_group_us']:.1f} us)")
        if args.mfma_dtype == "bf16" and args.hidden == 256 and args.task == "ball3d":
            attach_pmc_traffic(roof, "gradbf_kernel")
        if roof["kernel"] == "tma::ppo_grad_h64_kernel" and args.task == "gridworld":
            attach_pmc_traffic(roof, "grad_kernel")
        roof["spills"] = spill_summary(roof["kernel"], pick=f"ppo_grad_h64_kernel<{model.policy.obs_dim if model.policy.obs_dim in (4, 6) else 0}, 2>" if roof["kernel"].endswith("h64_kernel") else None)
        out["roofline"] = roof
        step_kernel_rooflines(out, args, env, model, world)
    env.close()
    del model
    torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not args.no_extras:
            try:
                out["literal_batch_256"] = literal_batch_256(args, dev)
                log(f"literal batch 256: {out['literal_batch_256']['ppo_updates_per_sec']:.0f} optimizer steps/s")
            except Exception as exc:  # noqa: BLE001
                out["literal_batch_256"] = {"error": str(exc)}
            try:  # ... and with the reference's default net (training.py:363-365): literal in both the width and the batch size
                out["literal_batch_256_h256"] = literal_batch_256(args, dev, hidden=256)
                log(f"literal batch 256, 256x256: {out['literal_batch_256_h256']['ppo_updates_per_sec']:.0f} optimizer steps/s")
            except Exception as exc:  # noqa: BLE001
                out["literal_batch_256_h256"] = {"error": str(exc)}
            extras = []
            for cfg in EXTRA_CONFIGS:
                try:
                    extras.append(run_extra(cfg, args, dev))
                except Exception as exc:  # noqa: BLE001
                    extras.append({"config": cfg["name"], "error": str(exc)})
            out["extra_configs"] = extras
            try:
                out["harness_train_task"] = harness_train_task(args, dev)
                log(f"harness train_task: {out['harness_train_task']['train_task_seconds']:.3f} s vs direct {out['harness_train_task']['direct_ppo_seconds']:.3f} s")
            except Exception as exc:  # noqa: BLE001
                out["harness_train_task"] = {"error": repr(exc)}
            try:
                out["thresholds"] = threshold_leg()
            except Exception as exc:  # noqa: BLE001
                out["thresholds"] = {"error": repr(exc)}
        if not args.no_cpu_baseline and world == 1:  # contract: CPU baseline on rank 0 at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds, batch)
            except Exception as exc:  # noqa: BLE001
                out["cpu_baseline"] = {"error": str(exc)}
        if args.sweep:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import env_sweep

            out["env_sweep"] = [env_sweep.run(args.task, n, 32, 3, pl) for n in (4096, 65536, 1 << 20, 1 << 22) for pl in (1, 32)]
        extras_path = write_extras(out)
    # The JSON line must be the LAST line of the job's stdout.  Native libraries write there too through C stdio (RCCL announces itself with a
    # "Librccl path : ..." line at communicator creation), fully buffered when stdout is a pipe and flushed only at process exit -- i.e. BEHIND
    # the line, from every rank.  So: every rank empties its C buffers, all ranks meet, and only then rank 0 prints.
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    sys.stdout.flush()
    dist.barrier()
    if rank == 0:
        print(compact_line(out, extras_path), flush=True)
    dist.barrier()


if __name__ == "__main__":
    main()
