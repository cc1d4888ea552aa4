#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by driving the REFERENCE's own env code.

Runs only in the build container (needs /root/reference, read-only).  Nothing from the
reference is copied: the fixtures are data (inputs + the outputs the reference produced).

What is imported, unmodified, from the reference:
  backend/mlagents/envs.py        (BasicMoveToGoalEnv, LegacySingleAgentGymAdapter, make_*_env)
  backend/examples/gridworld.py   (GridWorldEnv)
  backend/examples/ball3d.py      (Ball3DEnv)
  backend/examples/push.py        (PushEnv)
  backend/examples/walljump.py    (WallJumpEnv)
  backend/examples/bicycle.py, brick_break.py, glider.py   (BicycleEnv, BrickBreakEnv, GliderEnv: SURVEY.md 8f rank N3)
`gymnasium` is not installed here, so a ~20-line stand-in (Env with a no-op reset, Box/Discrete
holders) is placed in sys.modules first; it carries no arithmetic.

The driver below restates what SB3's DummyVecEnv + Monitor do around those envs
 with the build's per-(env, episode) seeding contract:
    episode k of env i is reset with  seed s(i,k) = (base + i + k * 2**20) mod 2**32
(k = 0 coincides with the reference's own `seed + rank`, backend/mlagents/training.py:80).
Actions come from a counter-based tape  a(i,t) = mix32(tape_seed, i, t) % n_actions.

The three float tasks (bicycle, brickbreak, glider) are generated in a child process that runs with
NPY_DISABLE_CPU_FEATURES set to the AVX-512 groups: numpy then takes np.tan / np.arctan2 of a float64 from libm instead of its
AVX-512 SVML kernels (which differ from libm in the last bit for ~0.5 % of the arguments), i.e. the fixtures are the reference as it
runs on a host without AVX-512 -- the configuration a plain-C restatement can match bit for bit.  The five earlier fixtures are
generated with numpy's default dispatch, as before.

Usage:  python tools/gen_golden.py [task ...]   (writes tests/golden/<task>.npz; no arguments = every task + the RNG fixture)
"""
from __future__ import annotations

import importlib.util
import os
import subprocess
import sys
import types

import numpy as np

REF = "/root/reference/backend"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

EP_STRIDE = 1 << 20


def episode_seed(base: int, i: int, k: int) -> int:
    return (base + i + k * EP_STRIDE) & 0xFFFFFFFF


def mix32(seed, i, t):
    """murmur3-finalizer style counter hash, all arithmetic mod 2**32 (vectorised)."""
    with np.errstate(over="ignore"):
        x = (np.uint32(seed) * np.uint32(0x9E3779B1)) ^ (np.asarray(i, np.uint32) * np.uint32(0x85EBCA77)) ^ (
            np.asarray(t, np.uint32) * np.uint32(0xC2B2AE3D)
        )
        x = np.asarray(x, np.uint32)
        x ^= x >> np.uint32(16)
        x *= np.uint32(0x85EBCA6B)
        x ^= x >> np.uint32(13)
        x *= np.uint32(0xC2B2AE35)
        x ^= x >> np.uint32(16)
    return x


def action_tape(tape_seed: int, n_envs: int, T: int, n_actions: int) -> np.ndarray:
    t = np.arange(T, dtype=np.uint32)[:, None]
    i = np.arange(n_envs, dtype=np.uint32)[None, :]
    return (mix32(tape_seed, i, t) % np.uint32(n_actions)).astype(np.int32)


def install_standin_gymnasium():
    gym = types.ModuleType("gymnasium")

    class Env:
        metadata = {}

        def reset(self, *, seed=None, options=None):
            return None

        def close(self):
            pass

    class Space:
        pass

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Discrete(Space):
        def __init__(self, n):
            self.n = int(n)

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Space, spaces.Box, spaces.Discrete = Space, Box, Discrete
    gym.Env, gym.spaces = Env, spaces
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces


def load_reference_envs():
    install_standin_gymnasium()
    sys.path.insert(0, REF)
    spec = importlib.util.spec_from_file_location("ref_mlagents_envs", os.path.join(REF, "mlagents", "envs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------
# internal-state capture / injection for the legacy envs (attribute names are the reference's)
# ------------------------------------------------------------------------------------------
def get_state(task: str, env) -> np.ndarray:
    """Flat float64 state vector of the underlying env object (layout documented per task)."""
    if task == "basic":
        return np.array([env.position, env.steps], dtype=np.float64)
    e = env.env
    if task == "gridworld":
        return np.array(
            [*e.agent_pos, *e.green_goals[0], *e.red_goals[0], int(e.current_goal_type), e.steps], dtype=np.float64
        )
    if task == "push":
        return np.array([*e.agent_pos, *e.box_pos, e.goal_pos[0], e.steps], dtype=np.float64)
    if task == "walljump":
        return np.array([e.agent_x, e.in_air, e.wall_height, e.steps], dtype=np.float64)
    if task == "ball3d":
        first = 1.0 if e.rot.dtype == np.float32 else 0.0
        return np.array([*e.rot, *e.pos, *e.vel, e.steps, first], dtype=np.float64)
    if task == "bicycle":
        return np.array([e.x, e.z, e.theta, e.phi, e.phi_dot, e.delta, *e.goal_pos, e.dist_to_goal, e.steps], dtype=np.float64)
    if task == "brickbreak":
        return np.array([e.paddle_x, *e.ball_pos, *e.ball_vel, e.steps, *e.bricks.flatten()], dtype=np.float64)
    if task == "glider":
        return np.array([*e.pos, *e.vel, *e.rot, *e.ang_vel, e.current_waypoint_index, e.steps], dtype=np.float64)
    raise KeyError(task)


FACTORY = {"brickbreak": "make_brick_break_env"}  # registry id -> factory name where they differ (backend/mlagents/registry.py:133-146)


def factory(mod, task):
    return getattr(mod, FACTORY.get(task, f"make_{task}_env"))


def vec_rollout(mod, task: str, n_envs: int, T: int, base_seed: int, tape_seed: int):
    make = factory(mod, task)
    envs = [make() for _ in range(n_envs)]
    n_act = envs[0].action_space.n
    D = envs[0].observation_space.shape[0]
    actions = action_tape(tape_seed, n_envs, T, n_act)
    ep_idx = [0] * n_envs
    reset_obs = np.zeros((n_envs, D), np.float32)
    for i, env in enumerate(envs):
        o, _ = env.reset(seed=episode_seed(base_seed, i, 0))
        reset_obs[i] = o
    obs = np.zeros((T, n_envs, D), np.float32)
    term_obs = np.zeros((T, n_envs, D), np.float32)
    rew64 = np.zeros((T, n_envs), np.float64)
    rew32 = np.zeros((T, n_envs), np.float32)
    term = np.zeros((T, n_envs), np.bool_)
    trunc = np.zeros((T, n_envs), np.bool_)
    ep_ret = np.zeros((T, n_envs), np.float64)
    ep_ret_round = np.zeros((T, n_envs), np.float64)
    ep_len = np.zeros((T, n_envs), np.int32)
    steps_info = np.zeros((T, n_envs), np.int32)
    mon_rewards = [[] for _ in range(n_envs)]
    for t in range(T):
        for i, env in enumerate(envs):
            o, r, te, tr, info = env.step(int(actions[t, i]))
            assert isinstance(te, bool) and isinstance(tr, bool)
            mon_rewards[i].append(float(r))  # Monitor.step
            rew64[t, i] = float(r)
            rew32[t, i] = np.float32(r)  # DummyVecEnv buf_rews is float32
            term[t, i], trunc[t, i] = te, tr
            steps_info[t, i] = info["steps"]
            if te or tr:
                s = sum(mon_rewards[i])
                ep_ret[t, i] = s
                ep_ret_round[t, i] = round(s, 6)
                ep_len[t, i] = len(mon_rewards[i])
                mon_rewards[i] = []
                term_obs[t, i] = o
                ep_idx[i] += 1
                o, _ = env.reset(seed=episode_seed(base_seed, i, ep_idx[i]))
            obs[t, i] = o
    return dict(
        actions=actions,
        reset_obs=reset_obs,
        obs=obs,
        terminal_obs=term_obs,
        rewards_f64=rew64,
        rewards_f32=rew32,
        terminated=term,
        truncated=trunc,
        ep_ret=ep_ret,
        ep_ret_round6=ep_ret_round,
        ep_len=ep_len,
        info_steps=steps_info,
        episodes_per_env=np.array(ep_idx, np.int32),
        meta=np.array([n_envs, T, base_seed, tape_seed, n_act, D], np.int64),
    )


def seeded_resets(mod, task: str, seeds: np.ndarray):
    make = factory(mod, task)
    env = make()
    D = env.observation_space.shape[0]
    obs = np.zeros((len(seeds), D), np.float32)
    states = []
    for j, s in enumerate(seeds):
        o, _ = env.reset(seed=int(s))
        obs[j] = o
        states.append(get_state(task, env))
    return dict(reset_seeds=seeds.astype(np.uint32), reset_seed_obs=obs, reset_seed_state=np.stack(states))


# ------------------------------------------------------------------------------------------
# single transitions under state injection (legacy 3-tuple, below the adapter)
# ------------------------------------------------------------------------------------------
def grid_transitions(mod, rng):
    from examples.gridworld import GridWorldEnv

    e = GridWorldEnv()
    rows_in, rows_out, obs_out = [], [], []
    cells = [(x, y) for x in range(5) for y in range(5)]
    for _ in range(6000):
        idx = rng.choice(25, size=3, replace=False)
        a, g, r = cells[idx[0]], cells[idx[1]], cells[idx[2]]
        gt = int(rng.integers(0, 2))
        steps = int(rng.choice([0, 1, 50, 98, 99]))
        act = int(rng.integers(0, 5))
        e.agent_pos, e.green_goals, e.red_goals, e.current_goal_type, e.steps = a, [g], [r], gt, steps
        o, rew, done = e.step(act)
        rows_in.append([*a, *g, *r, gt, steps, act])
        rows_out.append([*e.agent_pos, e.steps, float(rew), float(done)])
        obs_out.append(o)
    return dict(tr_in=np.array(rows_in, np.int32), tr_out=np.array(rows_out, np.float64), tr_obs=np.stack(obs_out))


def push_transitions(mod, rng):
    from examples.push import PushEnv

    e = PushEnv()
    rows_in, rows_out, obs_out = [], [], []
    cells = [(x, y) for x in range(6) for y in range(6)]
    for _ in range(8000):
        idx = rng.choice(36, size=2, replace=False)
        a, b = cells[idx[0]], cells[idx[1]]
        gx = int(rng.integers(0, 6))
        steps = int(rng.choice([0, 1, 60, 118, 119]))
        act = int(rng.integers(0, 5))
        e.agent_pos, e.box_pos, e.goal_pos, e.steps = a, b, (gx, 5), steps
        o, rew, done = e.step(act)
        rows_in.append([*a, *b, gx, steps, act])
        rows_out.append([*e.agent_pos, *e.box_pos, e.steps, float(rew), float(done)])
        obs_out.append(o)
    return dict(tr_in=np.array(rows_in, np.int32), tr_out=np.array(rows_out, np.float64), tr_obs=np.stack(obs_out))


def ball3d_transitions(mod, rng):
    from examples.ball3d import Ball3DEnv

    e = Ball3DEnv()
    rows_in, rows_out, obs_out = [], [], []
    for n in range(6000):
        first = bool(n % 3 == 0)
        if first:  # state right after a reset: rot is a float32 array
            rot = rng.uniform(-0.25, 0.25, 2).astype(np.float32)
        else:  # later steps: rot is float64 (np.clip with np.float64 bounds)
            rot = np.clip(rng.uniform(-0.5, 0.5, 2), -np.deg2rad(25.0), np.deg2rad(25.0))
        pos = rng.uniform(-3.2, 3.2, 2).astype(np.float32)
        vel = rng.uniform(-4.0, 4.0, 2).astype(np.float32)
        steps = int(rng.choice([0, 1, 100, 198, 199]))
        act = int(rng.integers(0, 5))
        e.rot, e.pos, e.vel, e.steps = rot.copy(), pos.copy(), vel.copy(), steps
        rows_in.append([*rot.astype(np.float64), *pos.astype(np.float64), *vel.astype(np.float64), steps, float(first), act])
        o, rew, done = e.step(act)
        assert e.rot.dtype == np.float64 and isinstance(rew, np.float32)
        rows_out.append([*e.rot, *e.pos.astype(np.float64), *e.vel.astype(np.float64), e.steps, float(rew), float(done)])
        obs_out.append(o)
    return dict(tr_in=np.array(rows_in, np.float64), tr_out=np.array(rows_out, np.float64), tr_obs=np.stack(obs_out))


def walljump_transitions(mod, rng):
    from examples.walljump import WallJumpEnv

    e = WallJumpEnv()
    rows_in, rows_out, obs_out = [], [], []
    for x in range(20):
        for in_air in range(4):
            for wall in (0, 1):
                for act in range(4):
                    steps = int(rng.choice([0, 1, 148, 149]))
                    e.agent_x, e.in_air, e.wall_height, e.steps = x, in_air, wall, steps
                    o, rew, done = e.step(act)
                    rows_in.append([x, in_air, wall, steps, act])
                    rows_out.append([e.agent_x, e.in_air, e.wall_height, e.steps, float(rew), float(done)])
                    obs_out.append(o)
    return dict(tr_in=np.array(rows_in, np.int32), tr_out=np.array(rows_out, np.float64), tr_obs=np.stack(obs_out))


def float_task_transitions(mod, task, rng):
    """Single legacy steps (below the adapter) from states visited by seeded random-action episodes: state in, action -> state out,
    reward, done, observation.  The injected state is the flat vector of get_state()."""
    env = factory(mod, task)()
    n_act = env.action_space.n
    rows_in, rows_out, obs_out = [], [], []
    for ep in range(40):
        env.reset(seed=1000 + ep)
        e = env.env
        for t in range(100):
            act = int(rng.integers(0, n_act))
            rows_in.append([*get_state(task, env), act])
            o, rew, done = e.step(act)
            rows_out.append([*get_state(task, env), float(rew), float(done)])
            obs_out.append(np.asarray(o, np.float64))
            if done:
                break
    return dict(tr_in=np.array(rows_in, np.float64), tr_out=np.array(rows_out, np.float64), tr_obs=np.stack(obs_out))


def basic_transitions(mod):
    env = mod.make_basic_env()
    rows_in, rows_out, obs_out = [], [], []
    for pos in range(21):
        for steps in (0, 48, 49):
            for act in range(3):
                env.reset(seed=1, options={"position": pos})
                env.steps = steps
                o, r, te, tr, info = env.step(act)
                rows_in.append([pos, steps, act])
                rows_out.append([info["position"], info["steps"], float(r), float(te), float(tr)])
                obs_out.append(o)
    return dict(tr_in=np.array(rows_in, np.int32), tr_out=np.array(rows_out, np.float64), tr_obs=np.stack(obs_out))


def rng_fixture():
    """numpy legacy global-RNG primitives (the third-party algorithm the legacy envs consume)."""
    seeds = np.array([0, 1, 5, 7, 123, 321, 10_001, 2**20 + 3, 2**31 + 17, 2**32 - 1], np.uint64)
    raw, shuf25, shuf36, choice2, randint6, unif = [], [], [], [], [], []
    for s in seeds:
        np.random.seed(int(s))
        raw.append(np.random.randint(0, 2**32, size=1300, dtype=np.uint64))  # crosses two twists
        np.random.seed(int(s))
        cells = list(range(25))
        np.random.shuffle(cells)
        shuf25.append(cells)
        choice2.append(int(np.random.choice([0, 1])))
        np.random.seed(int(s))
        cells = list(range(36))
        np.random.shuffle(cells)
        shuf36.append(cells)
        randint6.append(int(np.random.randint(0, 6)))
        np.random.seed(int(s))
        unif.append(np.random.uniform(-1.5, 1.5, size=6))
    return dict(
        seeds=seeds,
        raw_u32=np.array(raw, np.uint32),
        shuffle25=np.array(shuf25, np.int32),
        choice2_after25=np.array(choice2, np.int32),
        shuffle36=np.array(shuf36, np.int32),
        randint6_after36=np.array(randint6, np.int32),
        uniform_pm1p5=np.array(unif, np.float64),
    )


FLOAT_TASKS = ("bicycle", "brickbreak", "glider")
NO_AVX512 = "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL AVX512_SPR"


def main():
    os.makedirs(OUT, exist_ok=True)
    want = sys.argv[1:]
    cfg = {
        # task: (n_envs, T, base_seed, tape_seed)
        "basic": (8, 400, 1, 11),
        "gridworld": (16, 700, 1, 12),
        "push": (16, 900, 1, 13),
        "ball3d": (16, 900, 1, 14),
        "walljump": (16, 700, 1, 15),
        "bicycle": (16, 500, 1, 16),
        "brickbreak": (16, 900, 1, 17),
        "glider": (16, 2600, 1, 18),
    }
    floats = [t for t in FLOAT_TASKS if not want or t in want]
    if floats and os.environ.get("NPY_DISABLE_CPU_FEATURES") != NO_AVX512:
        # the float tasks: a child with the AVX-512 dispatch of numpy disabled (see the module docstring)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), *floats], env={**os.environ, "NPY_DISABLE_CPU_FEATURES": NO_AVX512})
        want = [t for t in (want or [*cfg, "rng"]) if t not in FLOAT_TASKS]
        if not want:
            return
    mod = load_reference_envs()
    rng = np.random.default_rng(20261002)
    for task, (n, T, base, tape) in cfg.items():
        if task in FLOAT_TASKS and os.environ.get("NPY_DISABLE_CPU_FEATURES") != NO_AVX512:
            continue
        # (the shared generator advances per task in table order whether or not the task is written, so a partial run reproduces the files)
        if task in FLOAT_TASKS:
            rng_t = np.random.default_rng(20261002 + tape)
        d = vec_rollout(mod, task, n, T, base, tape) if (not want or task in want) else None
        if d is not None:
            # a second rollout at the reference test-suite's seed (tests/test_mlagents.py:86 uses 321)
            d2 = vec_rollout(mod, task, 4, 300, 321, tape + 100)
            d.update({f"b_{k}": v for k, v in d2.items()})
            d.update(seeded_resets(mod, task, np.concatenate([np.arange(0, 200), [321, 10_001, 2**20 + 1, 2**32 - 1]])))
        if task == "gridworld":
            tr = grid_transitions(mod, rng)
        elif task == "push":
            tr = push_transitions(mod, rng)
        elif task == "ball3d":
            tr = ball3d_transitions(mod, rng)
        elif task == "walljump":
            tr = walljump_transitions(mod, rng)
        elif task in FLOAT_TASKS:
            tr = float_task_transitions(mod, task, rng_t)
        else:
            tr = basic_transitions(mod)
        if d is None:
            continue
        d.update(tr)
        path = os.path.join(OUT, f"{task}.npz")
        np.savez_compressed(path, **d)
        print(task, "episodes/env:", d["episodes_per_env"].tolist(), "->", path, os.path.getsize(path), "B")
    if not want or "rng" in want:
        path = os.path.join(OUT, "numpy_legacy_rng.npz")
        np.savez_compressed(path, **rng_fixture())
        print("rng ->", path, os.path.getsize(path), "B")


if __name__ == "__main__":
    main()
