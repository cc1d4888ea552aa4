"""GPU parity tests of the HIP vector-env engine (through the C ABI) against the golden vectors captured from
the reference and against the CPU oracle.  Integer tasks: bit-exact.  Ball3D: <= 1e-5 (north_star tolerance),
with the bit-exact mismatch count reported."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

TASKS = ["basic", "gridworld", "push", "ball3d", "walljump"]
FLOAT_TOL = 1e-5  # north_star: float tasks within 1e-5


def _engine(task, n, **kw):
    from three_mlagents_amd.vec_env import HipEnvEngine

    return HipEnvEngine(task, n, **kw)


def _cmp(task, name, got, ref, ctx):
    got, ref = np.asarray(got), np.asarray(ref)
    if task in ("ball3d", "crawler", "ant") and got.dtype.kind == "f":
        assert np.allclose(got, ref, rtol=0, atol=FLOAT_TOL), (task, name, ctx, np.abs(got - ref).max())
        return int((got != ref).sum())
    assert np.array_equal(got, ref), (task, name, ctx)
    return 0


def _run_golden(task, g, prefix, ring_depth):
    n, T, base, tape, n_act, D = [int(x) for x in g[prefix + "meta"]]
    eng = _engine(task, n, seed=base, ring_depth=ring_depth)
    inexact = _cmp(task, "reset_obs", eng.reset().cpu().numpy(), g[prefix + "reset_obs"], 0)
    actions = torch.from_numpy(g[prefix + "actions"]).cuda()
    for t in range(T):
        o = eng.step(actions[t])
        inexact += _cmp(task, "obs", o["obs"][0].cpu(), g[prefix + "obs"][t], t)
        inexact += _cmp(task, "rew", o["rew"][0].cpu(), g[prefix + "rewards_f32"][t], t)
        assert np.array_equal(o["term"][0].cpu().numpy().astype(bool), g[prefix + "terminated"][t]), (task, t)
        assert np.array_equal(o["trunc"][0].cpu().numpy().astype(bool), g[prefix + "truncated"][t]), (task, t)
        done = g[prefix + "terminated"][t] | g[prefix + "truncated"][t]
        inexact += _cmp(task, "term_obs", o["term_obs"][0].cpu().numpy()[done], g[prefix + "terminal_obs"][t][done], t)
        inexact += _cmp(task, "ep_ret", o["ep_ret"][0].cpu(), g[prefix + "ep_ret"][t], t)
        assert np.array_equal(o["ep_len"][0].cpu().numpy(), g[prefix + "ep_len"][t]), (task, t)
    assert np.array_equal(eng.episode_index().cpu().numpy(), g[prefix + "episodes_per_env"])
    eng.close()
    return inexact


@pytest.mark.parametrize("task", TASKS)
def test_golden_rollouts_from_reference(golden, task):
    g = golden(task)
    inexact = _run_golden(task, g, "", ring_depth=8)
    inexact += _run_golden(task, g, "b_", ring_depth=2)
    print(f"[{task}] elements not bit-identical to the reference: {inexact}")
    # Ball3D: the GPU's sin(double) may differ from the generating host's libm in the last ulp, which the 1e-5 bound inside _cmp
    # absorbs; observed so far: 0 elements.  Bounded so that a regression to "thousands of 1-ulp differences" cannot pass silently
    # (the fixtures hold ~2e5 compared elements: 0.1 % of them).
    assert inexact == 0 if task != "ball3d" else inexact <= 200, inexact


@pytest.mark.parametrize("task", TASKS)
def test_golden_seeded_resets(golden, task):
    g = golden(task)
    seeds = g["reset_seeds"]
    for s, obs_ref, st_ref in list(zip(seeds, g["reset_seed_obs"], g["reset_seed_state"]))[::7]:
        eng = _engine(task, 1, seed=int(s), ring_depth=2)
        obs = eng.reset().cpu().numpy()[0]
        st = eng.get_state().cpu().numpy()[0]
        assert np.array_equal(obs, obs_ref), (task, s)
        assert np.array_equal(st[: len(st_ref)], st_ref), (task, s)
        eng.close()


@pytest.mark.parametrize("task", TASKS)
def test_golden_transitions_state_injection(golden, task):
    g = golden(task)
    tin, tout, tobs = g["tr_in"], g["tr_out"], g["tr_obs"]
    n = len(tin)
    sdim = {"basic": 2, "gridworld": 8, "push": 6, "ball3d": 8, "walljump": 4}[task]
    eng = _engine(task, n, seed=1, ring_depth=2)
    eng.reset()
    eng.set_state(tin[:, :sdim].astype(np.float64))
    act = torch.from_numpy(tin[:, sdim].astype(np.int32)).cuda()
    o = eng.step(act, want_terminal_obs=True)
    done = (o["term"][0] | o["trunc"][0]).cpu().numpy().astype(bool)
    # the observation the legacy step returns is the terminal obs where the episode ended, else the new obs
    obs = np.where(done[:, None], o["term_obs"][0].cpu().numpy(), o["obs"][0].cpu().numpy())
    if task == "ball3d":
        assert np.allclose(obs, tobs, rtol=0, atol=FLOAT_TOL)
        assert np.allclose(o["rew"][0].cpu().numpy(), tout[:, 7].astype(np.float32), rtol=0, atol=FLOAT_TOL)
        legacy_done = tout[:, 8].astype(bool)
    else:
        assert np.array_equal(obs, tobs)
        rcol = {"basic": 2, "gridworld": 3, "push": 5, "walljump": 4}[task]
        assert np.array_equal(o["rew"][0].cpu().numpy(), tout[:, rcol].astype(np.float32))
        legacy_done = (tout[:, rcol + 1] + (tout[:, rcol + 2] if task == "basic" else 0)).astype(bool)
    assert np.array_equal(done, legacy_done)
    eng.close()


@pytest.mark.parametrize("task", TASKS + ["crawler", "ant"])
@pytest.mark.parametrize("mode", ["actions", "tape_multi"])
def test_against_oracle_many_envs(task, mode):
    n, T, base, tape_seed, offset, depth = 1000, 192, 7, 99, 5000, 16
    eng = _engine(task, n, seed=base, env_offset=offset, ring_depth=depth)
    ref = orc.OracleVecEnv(task, n, seed=base, env_offset=offset)
    inexact = _cmp(task, "reset", eng.reset().cpu().numpy(), ref.reset(), 0)
    n_act = orc.num_actions(task)
    if task in ("crawler", "ant"):
        rng = np.random.default_rng(3)
        actions = rng.uniform(-1.3, 1.3, size=(T, n, orc.act_dim(task))).astype(np.float32)
    else:
        actions = orc.action_tape(tape_seed, n, T, n_act, env_offset=offset)
    outs = []
    if mode == "actions" or task in ("crawler", "ant"):
        dev_actions = torch.from_numpy(actions).cuda()
        for t in range(T):
            o = eng.step(dev_actions[t])
            outs.append({k: v[0].cpu().numpy() for k, v in o.items()})
    else:  # device-generated tape, `depth` steps per launch
        for t0 in range(0, T, depth):
            o = eng.step(None, n_steps=depth, tape_seed=tape_seed, tape_t0=t0)
            for s in range(depth):
                outs.append({k: v[s].cpu().numpy() for k, v in o.items()})
    for t in range(T):
        r = ref.step(actions[t])
        o = outs[t]
        done = (r["term"] | r["trunc"]).astype(bool)
        inexact += _cmp(task, "obs", o["obs"], r["obs"], t)
        inexact += _cmp(task, "rew", o["rew"], r["rew32"], t)
        assert np.array_equal(o["term"], r["term"]) and np.array_equal(o["trunc"], r["trunc"]), (task, t)
        inexact += _cmp(task, "term_obs", o["term_obs"][done], r["term_obs"][done], t)
        inexact += _cmp(task, "ep_ret", o["ep_ret"], r["ep_ret"], t)
        assert np.array_equal(o["ep_len"], r["ep_len"]), (task, t)
    assert np.array_equal(eng.episode_index().cpu().numpy().astype(np.uint32), ref.episode_index())
    st = eng.get_state().cpu().numpy()
    inexact += _cmp(task, "state", st, ref.get_state(), "final")
    print(f"[{task}/{mode}] not bit-identical to oracle: {inexact}")
    if task in ("basic", "gridworld", "push", "walljump"):
        assert inexact == 0
    eng.close()


def _random_shapes():
    rng = np.random.default_rng(20260)
    cases = []
    for task in TASKS + ["bicycle", "brickbreak", "glider"]:
        for _ in range(3):
            depth = int(rng.choice([2, 3, 5, 16, 64, 512]))  # (the engine takes ring depths in [2, 4096])
            cases.append((task, int(rng.choice([1, 2, 3, 63, 64, 65, 127, 257, 1000])), depth, int(rng.integers(1, 5)) * depth + int(rng.integers(0, depth)),
                          int(rng.integers(0, 2**31 - 1)), int(rng.choice([0, 1, 4096, 2**20 - 3]))))
    return cases


@pytest.mark.parametrize("task,n,depth,T,seed,offset", _random_shapes())
def test_against_oracle_random_shapes(task, n, depth, T, seed, offset):
    """Seeded random draws of (env count, reset-ring depth, steps -- not a multiple of the ring depth, so launches of uneven length and a refill
    in between --, base seed, env offset of the shard) per task, device-generated action tape: every plane against the C oracle.  Edge shapes
    in the draw set: one env, two steps per launch, a 64-lane wave plus or minus one env, an offset that carries into the episode-seed bits."""
    eng = _engine(task, n, seed=seed, env_offset=offset, ring_depth=depth)
    ref = orc.OracleVecEnv(task, n, seed=seed, env_offset=offset)
    inexact = _cmp(task, "reset", eng.reset().cpu().numpy(), ref.reset(), 0)
    tape_seed = seed ^ 0x5A5A
    actions = orc.action_tape(tape_seed, n, T, orc.num_actions(task), env_offset=offset)
    float_task = task in ("ball3d", "bicycle", "brickbreak", "glider")
    cmp_task = "ball3d" if float_task else task  # (_cmp's tolerance branch)
    t = 0
    while t < T:
        k = min(depth, T - t, eng.steps_until_refill())
        o = eng.step(None, n_steps=k, tape_seed=tape_seed, tape_t0=t)
        for s in range(k):
            r = ref.step(actions[t + s])
            done = (r["term"] | r["trunc"]).astype(bool)
            inexact += _cmp(cmp_task, "obs", o["obs"][s].cpu().numpy(), r["obs"], t + s)
            inexact += _cmp(cmp_task, "rew", o["rew"][s].cpu().numpy(), r["rew32"], t + s)
            assert np.array_equal(o["term"][s].cpu().numpy(), r["term"]) and np.array_equal(o["trunc"][s].cpu().numpy(), r["trunc"]), (task, t + s)
            inexact += _cmp(cmp_task, "term_obs", o["term_obs"][s].cpu().numpy()[done], r["term_obs"][done], t + s)
            assert np.array_equal(o["ep_len"][s].cpu().numpy(), r["ep_len"]), (task, t + s)
        t += k
    assert np.array_equal(eng.episode_index().cpu().numpy().astype(np.uint32), ref.episode_index())
    if not float_task:
        assert inexact == 0
    eng.close()


@pytest.mark.parametrize("task", ["gridworld", "push"])
def test_refill_fallback_generator_is_exact(task):
    """With the register-resident MT19937 window shortened, most seeds need the general in-memory generator: the
    trajectories must still equal the oracle's bit for bit (exercises the rejection-sampling overflow path)."""
    n, T, depth = 600, 96, 8
    eng = _engine(task, n, seed=5, ring_depth=depth)
    eng.set_option("refill_small_window", 1)
    ref = orc.OracleVecEnv(task, n, seed=5)
    assert np.array_equal(eng.reset().cpu().numpy(), ref.reset())
    tape = orc.action_tape(17, n, T, orc.num_actions(task))
    for t0 in range(0, T, depth):
        o = eng.step(None, n_steps=depth, tape_seed=17, tape_t0=t0)
        for s in range(depth):
            r = ref.step(tape[t0 + s])
            assert np.array_equal(o["obs"][s].cpu().numpy(), r["obs"]) and np.array_equal(o["rew"][s].cpu().numpy(), r["rew32"]), (task, t0 + s)
    assert np.array_equal(eng.get_state().cpu().numpy(), ref.get_state())
    eng.close()


def test_sharding_equals_one_big_env():
    """env_offset sharding: two shards of 512 reproduce envs [0,512) and [512,1024) of one 1024-env engine."""
    T, tape = 64, 5
    big = _engine("gridworld", 1024, seed=11, ring_depth=16)
    a = _engine("gridworld", 512, seed=11, env_offset=0, ring_depth=16)
    b = _engine("gridworld", 512, seed=11, env_offset=512, ring_depth=16)
    ob, oa, obb = big.reset(), a.reset(), b.reset()
    assert torch.equal(ob[:512], oa) and torch.equal(ob[512:], obb)
    for t0 in range(0, T, 16):
        rb = big.step(None, n_steps=16, tape_seed=tape, tape_t0=t0)
        ra = a.step(None, n_steps=16, tape_seed=tape, tape_t0=t0)
        rbb = b.step(None, n_steps=16, tape_seed=tape, tape_t0=t0)
        for k in ("obs", "rew", "term", "trunc"):
            assert torch.equal(rb[k][:, :512], ra[k]) and torch.equal(rb[k][:, 512:], rbb[k]), k


def test_full_size_properties_gridworld():
    """BASELINE config 2 size (4096 envs x 1024 steps): structural invariants + determinism + Monitor aggregate."""
    N, T, depth = 4096, 1024, 32
    runs = []
    for _ in range(2):
        eng = _engine("gridworld", N, seed=1, ring_depth=depth)
        eng.reset()
        n_done = 0
        sum_len = 0
        rew_sum = 0.0
        chk = 0
        for t0 in range(0, T, depth):
            o = eng.step(None, n_steps=depth, tape_seed=1, tape_t0=t0)
            done = (o["term"] | o["trunc"]).bool()
            assert not (o["term"] & o["trunc"]).any()
            n_done += int(done.sum())
            sum_len += int(o["ep_len"].sum())
            assert int(o["ep_len"].max()) <= 100
            rew_sum += float(o["rew"].double().sum())
            obs = o["obs"]
            assert bool(((obs[..., 2] + obs[..., 3]) == 1.0).all()) and float(obs[..., :2].abs().max()) <= 1.0
            chk = (chk * 1000003 + int(o["obs"].view(torch.int32).sum().item()) + int(o["rew"].view(torch.int32).sum().item())) % (2**61 - 1)
        s_ret, s_len, cnt = eng.pop_episode_stats()
        assert cnt == n_done and int(s_len) == sum_len
        assert int(eng.episode_index().sum()) == n_done
        runs.append((n_done, sum_len, chk))
        eng.close()
    assert runs[0] == runs[1]
    assert 20 < N * T / runs[0][0] < 45  # random-policy episodes are ~31 steps


def test_gae_matches_oracle_bit_exact():
    from three_mlagents_amd import _lib

    rng = np.random.default_rng(0)
    for T, N in [(1, 1), (7, 3), (128, 257), (1024, 64)]:
        r = rng.normal(size=(T, N)).astype(np.float32)
        v = rng.normal(size=(T, N)).astype(np.float32)
        es = (rng.random((T, N)) < 0.1).astype(np.float32)
        lv = rng.normal(size=N).astype(np.float32)
        d = (rng.random(N) < 0.3).astype(np.uint8)
        adv_ref, ret_ref = orc.gae(r, v, es, lv, d, 0.99, 0.95)
        tr, tv, tes, tlv, td = (torch.from_numpy(x).cuda() for x in (r, v, es, lv, d))
        adv = torch.empty_like(tr)
        ret = torch.empty_like(tr)
        _lib.check(_lib.lib().tma_gae(_lib.ptr(tr), _lib.ptr(tv), _lib.ptr(tes), _lib.ptr(tlv), _lib.ptr(td), 0.99, 0.95, T, N,
                                      _lib.ptr(adv), _lib.ptr(ret), _lib.stream_ptr()))
        assert np.array_equal(adv.cpu().numpy(), adv_ref) and np.array_equal(ret.cpu().numpy(), ret_ref), (T, N)


def test_error_mapping_on_gpu():
    from three_mlagents_amd.vec_env import HipEnvEngine

    with pytest.raises(KeyError):
        HipEnvEngine("no-such-task", 4)
    with pytest.raises(ValueError):
        HipEnvEngine("gridworld", 0)
    eng = HipEnvEngine("gridworld", 4, ring_depth=4)
    with pytest.raises(ValueError):
        eng.step(torch.zeros(4, dtype=torch.int32, device="cuda"))  # step before reset
    eng.reset()
    with pytest.raises(ValueError):
        eng.step(torch.zeros(4, dtype=torch.float32, device="cuda"))  # wrong dtype for Discrete
    with pytest.raises(ValueError):
        eng.step(None, n_steps=5)  # exceeds steps until refill


@pytest.mark.parametrize("task", ["gridworld", "basic", "walljump"])
def test_episode_log_records_every_finished_episode(task):
    """tma_env_episode_log / tma_env_pop_episode_log (the per-episode Monitor rows, reference training.py:85-86): the records the step
    kernel appends equal the (return, length) pairs the same steps report through `infos[i]["episode"]`, env by env and in order; an
    overflowing log keeps counting.  Then the same check against the fused rollout kernels through PPO.collect_rollouts."""
    from three_mlagents_amd.vec_env import HipVecEnv

    env = HipVecEnv(task, 96, seed=3)
    env.engine.episode_log(4096)
    env.reset()
    rng = np.random.default_rng(0)
    want = {i: [] for i in range(96)}
    n_act = env.action_space.n
    for _ in range(160):
        _, _, dones, infos = env.step(rng.integers(0, n_act, 96))
        for i in np.nonzero(dones)[0]:
            want[int(i)].append((infos[i]["episode"]["r"], infos[i]["episode"]["l"]))
    r, l, e, seen = env.engine.pop_episode_log()
    assert seen == len(r) == sum(len(v) for v in want.values()) > 0
    got = {i: [] for i in range(96)}
    for rr, ll, ee in zip(r, l, e):
        got[int(ee)].append((round(float(rr), 6), int(ll)))
    for i in range(96):
        assert len(got[i]) == len(want[i])
        for (gr, gl), (wr, wl) in zip(got[i], want[i]):
            assert gl == wl and abs(gr - wr) <= 1e-5 * max(1.0, abs(wr))
    assert env.engine.pop_episode_log()[3] == 0  # emptied
    env.engine.episode_log(8)  # overflow: 8 records kept, every episode counted
    for _ in range(120):
        env.step(rng.integers(0, n_act, 96))
    r, l, e, seen = env.engine.pop_episode_log()
    assert len(r) == 8 and seen > 8
    env.close()


def test_episode_log_from_the_fused_rollout():
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    for hidden, mfma in ((64, "f32"), (256, "bf16"), (256, "f32")):
        env = make_vector_env("gridworld", n_envs=256, seed=2)
        model = PPO("MlpPolicy", env, n_steps=128, batch_size=2048, n_epochs=1, seed=2, policy_kwargs={"net_arch": [hidden, hidden], "mfma_dtype": mfma})
        env.engine.episode_log(1 << 16)
        model.collect_rollouts()
        s_ret, s_len, cnt = env.engine.pop_episode_stats()
        r, l, e, seen = env.engine.pop_episode_log()
        done = (model.buf["terminated"] | model.buf["truncated"]).sum().item()
        assert seen == len(r) == cnt == done > 0
        assert abs(float(r.astype(np.float64).sum()) - s_ret) <= 1e-4 * max(1.0, abs(s_ret)) and int(l.sum()) == int(s_len)
        assert 0 <= e.min() and e.max() < 256 and 1 <= l.min() and l.max() <= 100
        env.close()


def test_detached_episode_log_is_read_on_a_side_stream_while_the_next_rollout_runs():
    """tma_env_detach_episode_log / tma_env_pop_detached_episode_log (round 4): a host-side swap of the Monitor aggregate / episode-log buffers --
    what the kernels launched BEFORE the detach wrote is read back on a side stream behind an event while kernels launched after it fill the
    other set; nothing is lost, nothing is counted twice, and the two halves equal one undivided run."""
    import ctypes as C

    import torch

    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    def rollouts(split):
        env = make_vector_env("gridworld", n_envs=512, seed=7)
        model = PPO("MlpPolicy", env, n_steps=128, batch_size=4096, n_epochs=1, seed=7, policy_kwargs={"net_arch": [64, 64]})
        eng = env.engine
        eng.episode_log(1 << 16)
        model.collect_rollouts()
        first = None
        if split:
            eng.detach_episode_log()
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(eng.device))
            with pytest.raises(ValueError):
                eng.detach_episode_log()  # one detached set at a time
        model.collect_rollouts()  # (queued before the first half is read)
        if split:
            side = torch.cuda.Stream(eng.device)
            side.wait_event(ev)
            first = eng.pop_detached_episode_log(C.c_void_p(side.cuda_stream))
            with pytest.raises(ValueError):
                eng.pop_detached_episode_log()  # nothing detached any more
        s_ret, s_len, cnt = eng.pop_episode_stats()
        r, l, e, seen = eng.pop_episode_log()
        env.close()
        return first, (s_ret, s_len, cnt), r, l, e, seen

    first, st2, r2, l2, e2, seen2 = rollouts(True)
    _, st, r, l, e, seen = rollouts(False)
    (s_ret1, s_len1, cnt1), r1, l1, e1, seen1 = first
    assert cnt1 == seen1 == len(r1) > 0 and st2[2] == seen2 == len(r2) > 0
    assert cnt1 + st2[2] == st[2] == seen and s_len1 + st2[1] == st[1]
    assert abs((s_ret1 + st2[0]) - st[0]) <= 1e-9 * max(1.0, abs(st[0]))
    # per env the episodes of the two halves, in order, are the episodes of the undivided run (the log order across envs is execution order)
    for env_i in range(0, 512, 37):
        halves = [(float(a), int(b)) for a, b, c in zip(r1, l1, e1) if c == env_i] + [(float(a), int(b)) for a, b, c in zip(r2, l2, e2) if c == env_i]
        whole = [(float(a), int(b)) for a, b, c in zip(r, l, e) if c == env_i]
        assert halves == whole and len(whole) > 0


def test_env_handles_reuse_their_device_blocks():
    """tma_env_create / tma_env_destroy (round 4): the ~25 device blocks of a handle come from a size-keyed cache instead of hipMalloc / hipFree
    (the callers this library replaces build and close a vector env per run and per evaluation).  A recycled handle must behave like a fresh
    one -- no kernel may depend on what a block held before -- and building / closing handles in a loop must not grow the process."""
    from three_mlagents_amd.harness import make_vector_env

    def rollout(seed):
        env = make_vector_env("gridworld", n_envs=1024, seed=seed)
        eng = env.engine
        eng.episode_log(1 << 14)
        obs0 = eng.reset().clone()
        out = eng.step(None, n_steps=64, tape_seed=5, tape_t0=0)
        res = (obs0.cpu(), out["obs"][-1].cpu().clone(), out["rew"].cpu().clone(), eng.pop_episode_log(), eng.pop_episode_stats())
        env.close()
        return res

    first = rollout(3)
    rollout(4)  # leaves other contents in the blocks the next handle gets back
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    again = rollout(3)
    for _ in range(20):
        rollout(9)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert torch.equal(first[0], again[0]) and torch.equal(first[1], again[1]) and torch.equal(first[2], again[2])
    rows = lambda log: sorted(zip(log[2].tolist(), log[1].tolist(), log[0].tolist()))  # (records of one vector step land in atomic order)
    assert rows(first[3]) == rows(again[3]) and len(first[3][0]) > 0 and first[3][3] == again[3][3]
    assert first[4][2] == again[4][2] and abs(first[4][0] - again[4][0]) <= 1e-9 * max(1.0, abs(first[4][0])) and first[4][1] == again[4][1]
    assert free0 - free1 < (8 << 20), (free0, free1)  # (nothing accumulates: every close hands its blocks to the next create)
