"""Pins the CPU oracle (oracle/tma_oracle.c) against golden vectors captured from the reference.

Fixtures: tests/golden/*.npz, produced by tools/gen_golden.py importing
/root/reference/backend/{mlagents/envs.py, examples/gridworld.py, examples/ball3d.py, examples/push.py}.
Integer tasks must be bit-exact; Ball3D is checked bit-exact too (observed) with a 1e-5 fallback bound
documented in DESIGN.md.
"""
import numpy as np
import pytest

from oracle import oracle as orc

FLOAT_TASKS = ["bicycle", "brickbreak", "glider"]  # f64 physics, fixtures generated with numpy's libm paths (tools/gen_golden.py)
TASKS = ["basic", "gridworld", "push", "ball3d", "walljump"] + FLOAT_TASKS


def test_known_answer_basic_reference_unit_test():
    # /root/reference/backend/tests/test_mlagents.py:32-45 -- reset(seed=1) -> position 10; step(2) -> 11
    st, obs = orc.reset_from_seed("basic", 1)
    assert st[0] == 10 and obs[10] == 1.0 and obs.sum() == 1.0
    st, obs, r, done = orc.legacy_step("basic", st, 2)
    assert st[0] == 11 and st[1] == 1 and r == -0.01 and not done and obs[11] == 1.0


def test_numpy_legacy_rng(golden):
    g = golden("numpy_legacy_rng")
    for k, s in enumerate(g["seeds"]):
        raw = orc.mt_raw(int(s), 1300)
        assert np.array_equal(raw, g["raw_u32"][k]), f"raw stream differs for seed {s}"
        perm, nxt = orc.mt_shuffle(int(s), 25, next_max=1)
        assert np.array_equal(perm, g["shuffle25"][k]) and nxt == g["choice2_after25"][k]
        perm, nxt = orc.mt_shuffle(int(s), 36, next_max=5)
        assert np.array_equal(perm, g["shuffle36"][k]) and nxt == g["randint6_after36"][k]
        u = orc.mt_uniform(int(s), -1.5, 1.5, 6)
        assert np.array_equal(u, g["uniform_pm1p5"][k])


@pytest.mark.parametrize("task", TASKS)
def test_seeded_resets(golden, task):
    g = golden(task)
    for s, obs_ref, st_ref in zip(g["reset_seeds"], g["reset_seed_obs"], g["reset_seed_state"]):
        st, obs = orc.reset_from_seed(task, int(s))
        assert np.array_equal(obs, obs_ref), (task, s)
        assert np.array_equal(st[: len(st_ref)], st_ref), (task, s, st, st_ref)


@pytest.mark.parametrize("task", TASKS)
def test_single_transitions(golden, task):
    g = golden(task)
    tin, tout, tobs = g["tr_in"], g["tr_out"], g["tr_obs"]
    for row_in, row_out, o_ref in zip(tin, tout, tobs):
        if task == "basic":
            pos, steps, act = row_in
            st, obs, r, term = orc.legacy_step(task, [pos, steps], int(act))
            trunc = (st[1] >= 50) and not term
            assert [st[0], st[1], r, float(term), float(trunc)] == list(row_out)
        elif task == "gridworld":
            st, obs, r, done = orc.legacy_step(task, row_in[:8].astype(np.float64), int(row_in[8]))
            assert [st[0], st[1], st[7], r, float(done)] == list(row_out)
        elif task == "push":
            st, obs, r, done = orc.legacy_step(task, row_in[:6].astype(np.float64), int(row_in[6]))
            assert [st[0], st[1], st[2], st[3], st[5], r, float(done)] == list(row_out)
        elif task == "walljump":
            st, obs, r, done = orc.legacy_step(task, row_in[:4].astype(np.float64), int(row_in[4]))
            assert [st[0], st[1], st[2], st[3], r, float(done)] == list(row_out)
        elif task in FLOAT_TASKS:
            st, obs, r, done = orc.legacy_step(task, row_in[:-1], int(row_in[-1]))
            got = [*st[: len(row_out) - 2], r, float(done)]
            assert got == list(row_out), (row_in, got, row_out)
            o_ref = o_ref.astype(np.float32)  # the legacy env returns float64; the adapter casts (backend/mlagents/envs.py:147)
        else:
            st, obs, r, done = orc.legacy_step(task, row_in[:8], int(row_in[8]))
            got = [*st[:7], r, float(done)]
            assert got == list(row_out), (row_in, got, row_out)
        assert np.array_equal(obs, o_ref)


def _check_rollout(task, g, prefix=""):
    n, T, base, tape, n_act, D = [int(x) for x in g[prefix + "meta"]]
    actions = orc.action_tape(tape, n, T, n_act)
    assert np.array_equal(actions, g[prefix + "actions"])
    env = orc.OracleVecEnv(task, n, seed=base)
    assert np.array_equal(env.reset(), g[prefix + "reset_obs"])
    for t in range(T):
        o = env.step(actions[t])
        for key, ref in [("obs", "obs"), ("rew32", "rewards_f32"), ("rew64", "rewards_f64"), ("term_obs", "terminal_obs"),
                         ("ep_ret", "ep_ret"), ("ep_len", "ep_len")]:
            assert np.array_equal(o[key], g[prefix + ref][t]), (task, t, key)
        assert np.array_equal(o["term"].astype(bool), g[prefix + "terminated"][t]), (task, t)
        assert np.array_equal(o["trunc"].astype(bool), g[prefix + "truncated"][t]), (task, t)
    assert np.array_equal(env.episode_index(), g[prefix + "episodes_per_env"])


@pytest.mark.parametrize("task", TASKS)
def test_vec_rollout_matches_reference(golden, task):
    _check_rollout(task, golden(task))
    _check_rollout(task, golden(task), prefix="b_")
