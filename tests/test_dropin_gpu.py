"""GPU mirror of the reference's env-contract / construct-and-predict tests and of the CLI train/evaluate path
(/root/reference/backend/tests/test_mlagents.py:32-45,51-101; cli.py:70-95)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_basic_env_reset_and_step():  # test_mlagents.py:32-45
    from three_mlagents_amd.tasks import make_env

    env = make_env("basic")
    try:
        obs, info = env.reset(seed=1)
        assert obs.shape == env.observation_space.shape and info["position"] == 10
        next_obs, reward, terminated, truncated, info = env.step(2)
        assert next_obs.shape == env.observation_space.shape and isinstance(reward, float) and reward == -0.01
        assert terminated is False and truncated is False and info["position"] == 11
        obs, info = env.reset(options={"position": 16})
        _, reward, terminated, truncated, info = env.step(2)
        assert terminated and not truncated and info["position"] == 17 and abs(reward - 0.99) < 1e-6
    finally:
        env.close()


def test_trainable_env_contracts_match_declared_spaces():  # test_mlagents.py:51-72
    from three_mlagents_amd.tasks import ENGINE_TASKS, make_env

    for task in ENGINE_TASKS.values():
        env = make_env(task.id)
        try:
            obs, _ = env.reset(seed=123)
            assert env.observation_space.contains(obs), task.id
            next_obs, reward, terminated, truncated, _ = env.step(env.action_space.sample())
            assert env.observation_space.contains(next_obs), task.id
            assert isinstance(float(reward), float) and isinstance(terminated, bool) and isinstance(truncated, bool)
        finally:
            env.close()


@pytest.mark.parametrize("task", ["basic", "gridworld", "push", "walljump"])
def test_single_env_returns_the_reference_python_float_reward(golden, task):
    """Seam S1: `env.step()` of the reference returns a Python float computed in float64 (Basic's goal step: -0.01 + 0.1 =
    0.09000000000000001, backend/mlagents/envs.py:65-84).  HipSingleEnv hands back exactly that float -- from the per-task table of the
    finite float64 reward set (envs.reward_table), not a decimal rounding of the kernel's float32 -- asserted `==` against `rewards_f64` of
    the reference-generated fixture over whole multi-episode trajectories of two envs of the vector."""
    from three_mlagents_amd.tasks import make_env

    g = golden(task)
    n_envs, T, base_seed = (int(x) for x in g["meta"][:3])
    env = make_env(task)
    try:
        for i in (0, n_envs - 1):
            obs, _ = env.reset(seed=base_seed + i)
            assert np.array_equal(obs, g["reset_obs"][i])
            seen = set()
            for t in range(min(T, 400)):
                obs, r, te, tr, _ = env.step(int(g["actions"][t, i]))
                assert isinstance(r, float) and r == float(g["rewards_f64"][t, i]), (task, i, t, r, float(g["rewards_f64"][t, i]))
                assert te == bool(g["terminated"][t, i]) and tr == bool(g["truncated"][t, i])
                seen.add(r)
                if te or tr:
                    assert np.array_equal(obs, g["terminal_obs"][t, i])
                    obs, _ = env.reset()
                assert np.array_equal(obs, g["obs"][t, i])
            assert len(seen) >= 2
    finally:
        env.close()


def test_single_env_matches_reference_seeded_reset(golden):
    from three_mlagents_amd.tasks import make_env

    g = golden("gridworld")
    env = make_env("gridworld")
    for s, obs_ref in list(zip(g["reset_seeds"], g["reset_seed_obs"]))[:5]:
        obs, info = env.reset(seed=int(s))
        assert np.array_equal(obs, obs_ref) and info == {"steps": 0}
    # /root/reference probe quoted in SURVEY.md §8c: reset(seed=7) -> [0.25,-0.75,1,0]; 100 no-ops -> truncated, not terminated
    obs, _ = env.reset(seed=7)
    assert obs.tolist() == [0.25, -0.75, 1.0, 0.0]
    for _ in range(100):
        obs, r, te, tr, info = env.step(0)
    assert tr and not te and info["steps"] == 100
    env.close()


def test_registered_algorithms_construct_and_predict():  # test_mlagents.py:74-101
    from three_mlagents_amd.harness import ALGORITHMS, make_vector_env, ppo_defaults
    from three_mlagents_amd.tasks import ENGINE_TASKS

    for task in ENGINE_TASKS.values():
        vec_env = make_vector_env(task.id, n_envs=2, seed=321)
        try:
            kwargs = {**ppo_defaults(task), "n_steps": 16}
            model = ALGORITHMS["ppo"]("MlpPolicy", vec_env, seed=321, **kwargs)
            action, _ = model.predict(vec_env.reset(), deterministic=True)
            assert action is not None and len(action) == 2
            obs, rew, dones, infos = vec_env.step(action)
            assert obs.shape == (2,) + vec_env.observation_space.shape and rew.dtype == np.float32 and dones.dtype == bool and len(infos) == 2
        finally:
            vec_env.close()


def test_vec_env_infos_carry_sb3_keys():
    from three_mlagents_amd.vec_env import HipVecEnv, HipVectorEnv

    env = HipVecEnv("gridworld", 64, seed=3)
    env.reset()
    rng = np.random.default_rng(0)
    seen = False
    for _ in range(150):
        obs, rew, dones, infos = env.step(rng.integers(0, 5, 64))
        for i in np.nonzero(dones)[0]:
            info = infos[i]
            assert set(info) >= {"terminal_observation", "TimeLimit.truncated", "episode"} and set(info["episode"]) == {"r", "l", "t"}
            assert info["terminal_observation"].shape == (4,) and info["episode"]["l"] <= 100
            seen = True
    assert seen
    env.close()
    genv = HipVectorEnv("push", 8, seed=1)
    obs, infos = genv.reset(seed=5)
    obs, rew, term, trunc, infos = genv.step(np.zeros(8, np.int64))
    assert obs.shape == (8, 4) and term.dtype == bool and trunc.dtype == bool
    genv.close()


def test_cli_train_and_evaluate(tmp_path, monkeypatch, capsys):  # cli.py:70-95 ; artefacts training.py:172-207
    monkeypatch.chdir(tmp_path)
    from three_mlagents_amd import __main__ as cli

    cli.main(["train", "basic", "--algorithm", "ppo", "--n-envs", "8", "-t", "4096", "--eval-episodes", "4", "--eval-freq", "2048", "--run-name", "t1", "--quiet"])
    out = json.loads(capsys.readouterr().out)
    assert out["task_id"] == "basic" and out["algorithm"] == "ppo" and out["model_filename"] == "basic_policy_t1.zip"
    assert os.path.exists(tmp_path / "policies" / "basic_policy_t1.zip")
    meta = json.loads((tmp_path / "runs" / "basic" / "t1" / "metadata.json").read_text())
    assert meta["run_id"] == "t1" and len(meta["episode_rewards"]) == 4 and meta["task"]["id"] == "basic"
    assert meta["schedule"]["batch_size"] == 256 and meta["schedule"]["batch_size_from"] == "256 * max(1, n_envs // 8)"  # the reference's literal value at its env count
    assert os.path.exists(tmp_path / "runs" / "basic" / "t1" / "eval" / "evaluations.npz")
    prog = (tmp_path / "runs" / "basic" / "t1" / "tb" / "progress.csv").read_text().splitlines()
    assert "rollout/ep_rew_mean" in prog[0] and "train/approx_kl" in prog[0] and len(prog) >= 2
    # SB3 Monitor layout, one file per env of the vector as the reference writes them (training.py:84-86: monitor_dir / f"{rank}"), one row per episode
    mdir = tmp_path / "runs" / "basic" / "t1" / "monitor"
    assert sorted(p.name for p in mdir.iterdir()) == [f"{k}.monitor.csv" for k in range(8)]
    n_rows = 0
    for k in range(8):
        mon = (mdir / f"{k}.monitor.csv").read_text().splitlines()
        assert mon[0].startswith("#{") and "t_start" in json.loads(mon[0][1:]) and mon[1] == "r,l,t"
        rows = [ln.split(",") for ln in mon[2:] if not ln.startswith("#")]
        assert all(len(r) == 3 and 1 <= int(r[1]) <= 50 for r in rows)  # Basic: at most 50 steps per episode
        assert sum(int(r[1]) for r in rows) <= 1024 and len(rows) >= 1024 // 50 - 1  # the env's own steps (one 1024-step rollout), cut into its episodes
        assert [float(r[2]) for r in rows] == sorted(float(r[2]) for r in rows)
        n_rows += len(rows)
    assert n_rows >= 8192 // 50 - 8
    from three_mlagents_amd.tb_events import read_scalars

    tb_dir = tmp_path / "runs" / "basic" / "t1" / "tb" / "PPO_1"
    events = read_scalars(str(next(tb_dir.iterdir())))  # TensorBoard event file with the scalars SB3's logger writes
    assert len(events) >= 1 and {"rollout/ep_rew_mean", "train/approx_kl", "time/fps"} <= set(events[0][1]) and events[-1][0] >= 4096
    cli.main(["evaluate", "basic", "basic_policy_t1.zip", "--episodes", "3"])
    ev = json.loads(capsys.readouterr().out)
    assert ev["episodes"] == 3 and len(ev["episode_lengths"]) == 3
    from three_mlagents_amd.harness import predict_action

    assert predict_action("basic", np.eye(21, dtype=np.float32)[10], "basic_policy_t1.zip") in (0, 1, 2)


def test_load_zip_with_sb3_schema(tmp_path):
    """A zip laid out the way stable-baselines3 writes it: SB3's own `data` keys, policy.pth with SB3's state_dict
    names -- the policy shape is recovered from the tensors and `predict` works."""
    import io
    import zipfile

    import torch

    from oracle import sb3_ref
    from three_mlagents_amd.ppo import PPO

    sd = sb3_ref.init_policy(4, 64, 5, False, seed=7)
    path = tmp_path / "gridworld_policy_sb3like.zip"
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("data", json.dumps({"policy_class": {":type:": "<class 'abc.ABCMeta'>", ":serialized:": "gAWV..."}, "n_envs": 8,
                                        "learning_rate": {":type:": "<class 'function'>", ":serialized:": "gAWV..."}, "gamma": 0.99, "n_steps": 1024}))
        bio = io.BytesIO()
        torch.save(sd, bio)
        z.writestr("policy.pth", bio.getvalue())
        z.writestr("policy.optimizer.pth", b"not-a-pickle")
        z.writestr("_stable_baselines3_version", "2.9.0")
    model = PPO.load(str(path))
    obs = torch.randn(9, 4, generator=torch.Generator().manual_seed(0))
    act, _ = model.predict(obs.numpy(), deterministic=True)
    logits, _ = sb3_ref.forward(sd, obs)
    assert np.array_equal(act, logits.argmax(dim=1).numpy()) and model.n_steps == 1024 and model.learning_rate == 3e-4


def test_engine_runs_from_a_worker_thread():
    """The reference runs train_task under asyncio.to_thread (/root/reference/backend/main.py:152, websocket_training.py:98): env
    stepping, a full PPO iteration (tma_rollout_collect, tma_gae_flags, tma_ppo_minibatch_grad, tma_ppo_adam_step_local) and predict
    must work off the main thread -- every tma_policy_* / tma_ppo_* entry makes tma_policy_dims.device current on the calling thread --
    and give the same bits as on the main thread."""
    import asyncio
    import threading

    import torch

    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    def work():
        env = make_vector_env("gridworld", n_envs=64, seed=5)
        model = PPO("MlpPolicy", env, n_steps=32, batch_size=512, n_epochs=2, seed=5, policy_kwargs={"net_arch": [64, 64]})
        model.learn(2 * 64 * 32)
        obs, rew, dones, infos = env.step(np.zeros(64, np.int64))
        act, _ = model.predict(obs, deterministic=True)
        params = model.policy.params[: model.policy.n_trainable].cpu()
        env.close()
        return threading.current_thread() is threading.main_thread(), params, act

    async def off_thread():
        return await asyncio.to_thread(work)

    was_main, p_thread, a_thread = asyncio.run(off_thread())
    is_main, p_main, a_main = work()
    assert not was_main and is_main
    assert torch.equal(p_thread, p_main) and np.array_equal(a_thread, a_main) and torch.isfinite(p_main).all()


def test_bridge_streams_run_steps_with_device_state(tmp_path, monkeypatch):
    """bridge.run_for_websocket (reference websocket_training.py:141-185): a saved policy drives one device env and every `run_step` frame
    carries that env's state read back from the GPU; the Crawler-shape readback has the reference wrapper's three keys."""
    import asyncio

    monkeypatch.chdir(tmp_path)
    from three_mlagents_amd import bridge, harness

    harness.train_task(harness.TrainConfig("gridworld", 2048, "ppo", 1, 8, 2, 100_000, run_name="b1", verbose=0))
    frames = []

    class Sock:
        async def send_json(self, payload):
            frames.append(payload)

    episodes = asyncio.run(bridge.run_for_websocket(Sock(), "gridworld", model_filename="gridworld_policy_b1.zip", sleep_seconds=0.0, max_steps=130))
    assert len(frames) == 130 and all(f["type"] == "run_step" and set(f["state"]) == set(bridge.STATE_FIELDS["gridworld"]) for f in frames)
    assert episodes >= 1 and frames[-1]["episode"] >= episodes and all(0 <= f["state"]["agentX"] < 5 and 1 <= f["state"]["steps"] <= 100 or f["state"]["steps"] == 0 for f in frames)
    from three_mlagents_amd.vec_env import HipVecEnv

    env = HipVecEnv("crawler", 4, seed=1)
    env.reset()
    st = bridge.state_for_viz(env, 2)
    assert set(st) == {"basePos", "baseOri", "jointAngles", "steps"} and len(st["basePos"]) == 3 and len(st["baseOri"]) == 4 and len(st["jointAngles"]) == 8
    env.close()


def test_saved_zip_follows_the_sb3_layout(tmp_path):
    """PPO.save (training.py:172-175 `model.save`): the zip members SB3's load_from_zip_file reads; `policy.pth` loads strictly into a torch
    module laid out like SB3's ActorCriticPolicy and reproduces the engine's values; `policy.optimizer.pth` is a torch.optim.Adam
    state_dict for that module; PPO.load(env=...) restores parameters AND moments bit for bit."""
    import io
    import zipfile

    import torch

    from test_harness_cpu import _sb3_like_policy
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    for task, cont in (("gridworld", False), ("crawler", True)):
        env = make_vector_env(task, n_envs=64, seed=1)
        model = PPO("MlpPolicy", env, n_steps=32, batch_size=512, n_epochs=2, seed=1, policy_kwargs={"net_arch": [64, 64]})
        model.learn(2 * 64 * 32)
        path = tmp_path / f"{task}_policy.zip"
        model.save(str(path))
        with zipfile.ZipFile(path) as z:
            assert {"data", "policy.pth", "policy.optimizer.pth", "pytorch_variables.pth", "_stable_baselines3_version", "system_info.txt"} <= set(z.namelist())
            data = json.loads(z.read("data"))
            sd = torch.load(io.BytesIO(z.read("policy.pth")), weights_only=True)
            opt_sd = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), weights_only=True)
        for key in ("policy_class", "observation_space", "action_space"):
            assert set(data[key]) >= {":type:", ":serialized:"}
        for key in ("learning_rate", "n_steps", "gamma", "gae_lambda", "n_envs", "clip_range", "use_sde", "policy_kwargs", "seed", "num_timesteps"):
            assert key in data and not isinstance(data[key], dict) or key == "policy_kwargs"
        assert data["policy_kwargs"] == {"net_arch": [64, 64]} and data["n_envs"] == 64 and data["num_timesteps"] == 2 * 64 * 32
        D, A = model.policy.obs_dim, model.policy.act_dim
        net = _sb3_like_policy(D, 64, A, cont)
        assert list(sd) == [n for n, _ in net.named_parameters()]
        net.load_state_dict(sd, strict=True)
        obs = torch.randn(17, D)
        v_torch = net.value_net(net.mlp_extractor.value_net(obs)).squeeze(1)
        assert torch.allclose(model.policy.predict_values(obs.cuda()).cpu(), v_torch, rtol=1e-5, atol=1e-5)
        opt = torch.optim.Adam(net.parameters(), lr=1.0, eps=1e-5)
        opt.load_state_dict(opt_sd)
        assert float(opt.state[net.action_net.weight]["step"]) == model._adam_step > 0
        assert torch.equal(opt.state[net.value_net.bias]["exp_avg"], model.policy.named_from_flat(model.exp_avg)["value_net.bias"])
        env2 = make_vector_env(task, n_envs=64, seed=1)
        again = PPO.load(str(path), env=env2)
        assert torch.equal(again.policy.params, model.policy.params) and torch.equal(again.exp_avg, model.exp_avg) and torch.equal(again.exp_avg_sq, model.exp_avg_sq)
        assert again._adam_step == model._adam_step and again.num_timesteps == model.num_timesteps and again.n_steps == 32
        env.close(), env2.close()


@pytest.mark.parametrize("task,hidden,mfma,n_envs,episodes", [("gridworld", 64, "f32", 64, 100), ("ball3d", 256, "bf16", 48, 30), ("basic", 256, "f32", 8, 50),
                                                             ("ant", 256, "bf16", 40, 40), ("push", 64, "f32", 7, 20)])
def test_device_side_evaluation_equals_the_per_step_loop(task, hidden, mfma, n_envs, episodes):
    """evaluate_policy (native deterministic rollout chunks over all envs of the evaluation vector, episodes read from the device episode
    log: reference training.py:177-184,240-247) reports exactly the episodes the per-step host loop reports -- returns (f64 Monitor sums),
    lengths and order -- for the fused H = 64 / bf16-wide / Box-action kernels and the per-step path, and for sampled actions too."""
    from three_mlagents_amd.evaluation import evaluate_policy, evaluate_policy_stepwise
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    train_env = make_vector_env(task, n_envs=64, seed=1)
    model = PPO("MlpPolicy", train_env, n_steps=32, batch_size=512, n_epochs=2, seed=1, ent_coef=0.01, policy_kwargs={"net_arch": [hidden, hidden], "mfma_dtype": mfma})
    model.learn(2 * 64 * 32)  # (a policy that is not the orthogonal init: the argmax is not degenerate)
    env_a, env_b = make_vector_env(task, n_envs=n_envs, seed=10_001), make_vector_env(task, n_envs=n_envs, seed=10_001)
    for det in (True, False):
        ra, la = evaluate_policy(model, env_a, n_eval_episodes=episodes, deterministic=det, return_episode_rewards=True)
        rb, lb = evaluate_policy_stepwise(model, env_b, n_eval_episodes=episodes, deterministic=det, return_episode_rewards=True)
        assert len(ra) == len(rb) == episodes and la == lb, (task, det)
        assert ra == rb, (task, det, max(abs(x - y) for x, y in zip(ra, rb)))
    ra2, la2 = evaluate_policy(model, env_a, n_eval_episodes=episodes, deterministic=True, return_episode_rewards=True, chunk_steps=7)
    rb2, lb2 = evaluate_policy_stepwise(model, env_b, n_eval_episodes=episodes, deterministic=True, return_episode_rewards=True)
    assert ra2 == rb2 and la2 == lb2  # a chunk length that divides nothing
    m, s = evaluate_policy(model, env_a, n_eval_episodes=episodes)
    assert abs(m - float(np.mean(rb2))) < 1e-12 and abs(s - float(np.std(rb2))) < 1e-12
    for e in (train_env, env_a, env_b):
        e.close()


def test_eval_callback_repeats_rows_while_the_policy_has_not_moved(tmp_path, monkeypatch):
    """train_task at 4096 envs: `eval_freq // n_envs` is 2 vector steps (reference training.py:156), i.e. 512 evaluation rows per
    1024-step rollout on a policy that cannot change inside a rollout.  Every row is kept (the reference's cadence); the episodes are
    run once per optimizer state."""
    monkeypatch.chdir(tmp_path)
    from three_mlagents_amd import harness
    from three_mlagents_amd.callbacks import EvalCallback

    seen = {}
    orig = EvalCallback._on_training_end

    def spy(self):
        seen["cb"] = self
        orig(self)

    monkeypatch.setattr(EvalCallback, "_on_training_end", spy)
    cfg = harness.TrainConfig("gridworld", total_timesteps=2 * 4096 * 1024, n_envs=4096, eval_episodes=100, run_name="big", verbose=0)
    res = harness.train_task(cfg, model_kwargs={"batch_size": 131072, "policy_kwargs": {"net_arch": [64, 64]}})
    cb = seen["cb"]
    assert cb.eval_freq == 2 and len(cb.evaluations_timesteps) == 2 * 1024 // 2
    assert cb.n_fresh_evaluations == 2  # once per rollout (the optimizer stepped in between), not 1024 times
    ev = np.load(tmp_path / "runs" / "gridworld" / "big" / "eval" / "evaluations.npz")
    assert ev["timesteps"].shape == (1024,) and ev["results"].shape == (1024, 100) and ev["ep_lengths"].shape == (1024, 100)
    assert np.array_equal(ev["results"][0], ev["results"][511]) and not np.array_equal(ev["results"][0], ev["results"][512])
    assert ev["timesteps"][0] == 2 * 4096 and ev["timesteps"][-1] == 2 * 4096 * 1024
    assert res.eval_episodes == 100 and res.total_timesteps == 2 * 4096 * 1024
    meta = json.loads((tmp_path / "runs" / "gridworld" / "big" / "metadata.json").read_text())
    assert meta["substituted_for"] == "dqn" and len(meta["episode_rewards"]) == 100
    # the schedule that was trained with is on record (deviation 10: batch_size follows the env count unless given / TMA_LITERAL_BATCH)
    assert meta["schedule"] == {"batch_size": 131072, "n_steps": 1024, "n_epochs": 10, "n_envs": 4096, "minibatches_per_epoch": 32,
                                "reference_batch_size": 256, "literal_batch_env": False, "batch_size_from": "model_kwargs"}


def test_pipelined_logging_leaves_the_files_of_the_synchronous_order(tmp_path, monkeypatch):
    """PPO.learn queues rollout k + 1 before it has seen the statistics of update k (detached episode log read on a side stream, two-phase train
    statistics, evaluation beside the next rollout, callbacks in bulk: DESIGN.md deviation 11).  Everything it writes must be what the
    synchronous order (TMA_SYNC_LOGGING=1: pop, log, evaluate, then the next rollout) writes: every progress row except the clock, every Monitor
    row except its timestamp, every evaluation, the parameters."""
    import csv

    import torch

    from three_mlagents_amd import harness

    def run(name, sync):
        monkeypatch.chdir(tmp_path)
        for var in ("TMA_SYNC_LOGGING", "TMA_SYNC_EVAL"):  # (TMA_SYNC_EVAL=1: EvalCallback waits for every evaluation where it starts it)
            if sync:
                monkeypatch.setenv(var, "1")
            else:
                monkeypatch.delenv(var, raising=False)
        cfg = harness.TrainConfig("gridworld", total_timesteps=6 * 256 * 64, n_envs=256, eval_episodes=20, eval_freq=4096, run_name=name, verbose=0, seed=3)
        res = harness.train_task(cfg, model_kwargs={"n_steps": 64, "batch_size": 2048, "n_epochs": 2, "policy_kwargs": {"net_arch": [64, 64]}})
        root = tmp_path / "runs" / "gridworld" / name
        with open(root / "tb" / "progress.csv") as f:
            prog = list(csv.DictReader(f))
        mon = [ln.split(",")[:2] for ln in (root / "monitor" / "0.monitor.csv").read_text().splitlines()[2:] if not ln.startswith("#")]
        ev = np.load(root / "eval" / "evaluations.npz")
        model = harness.load_model("gridworld", res.model_filename)
        import io
        import zipfile

        with zipfile.ZipFile(root / "best_model" / "best_model.zip") as z:  # written from the snapshot the deferred evaluation ran on
            best = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
            best_opt = torch.load(io.BytesIO(z.read("policy.optimizer.pth")), map_location="cpu", weights_only=True)
            best_steps = __import__("json").loads(z.read("data").decode())["num_timesteps"]
        return prog, mon, {k: ev[k].copy() for k in ev.files}, model.policy.params.cpu(), res, (best, best_opt, best_steps)

    prog_p, mon_p, ev_p, par_p, res_p, best_p = run("pipelined", False)
    prog_s, mon_s, ev_s, par_s, res_s, best_s = run("sync", True)
    assert len(prog_p) == len(prog_s) == 6
    for a, b in zip(prog_p, prog_s):
        for k in a:
            if k != "time/fps":  # (rollout/ep_rew_mean is a sum of per-block double atomics: its last bit depends on their order in either mode)
                x, y = float(a[k]), float(b[k])
                assert (x != x and y != y) or abs(x - y) <= 1e-12 * max(1.0, abs(y)), (k, a[k], b[k])
    # (one file for the vector: rows are in the order the kernels' atomics logged them, which varies from run to run within a vector step)
    assert sorted(mon_p) == sorted(mon_s) and len(mon_p) > 100
    assert all(np.array_equal(ev_p[k], ev_s[k]) for k in ev_s) and ev_p["results"].shape[1] == 20
    assert torch.equal(par_p, par_s) and res_p.mean_reward == res_s.mean_reward
    # the best model: the same evaluation won in both orders, and the zip holds the parameters / optimizer moments / step counters of THAT policy
    assert best_p[2] == best_s[2] and best_p[0].keys() == best_s[0].keys() and all(torch.equal(best_p[0][k], best_s[0][k]) for k in best_s[0])
    st_p, st_s = best_p[1].get("state", {}), best_s[1].get("state", {})
    assert st_p.keys() == st_s.keys() and all(torch.equal(torch.as_tensor(st_p[i][k]), torch.as_tensor(st_s[i][k])) for i in st_s for k in st_s[i])


def test_stats_window_size_selects_sb3s_last_100_episode_window(tmp_path):
    """SB3 logs rollout/ep_rew_mean over its ep_info_buffer (deque of the last `stats_window_size` = 100 finished episodes, carried across
    iterations).  Opt-in here (default: the interval mean, reproducible): the window mean must be the mean of the last 100 Monitor records."""
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    env = make_vector_env("gridworld", n_envs=16, seed=3, monitor_dir=str(tmp_path / "monitor"))
    model = PPO("MlpPolicy", env, n_steps=256, batch_size=512, n_epochs=1, seed=3, policy_kwargs={"net_arch": [64, 64]}, stats_window_size=100)
    model.learn(16 * 256 * 3)
    rows = []
    for rank in range(16):
        for ln in (tmp_path / "monitor" / f"{rank}.monitor.csv").read_text().splitlines()[2:]:
            if not ln.startswith("#"):
                rows.append(float(ln.split(",")[0]))
    assert len(rows) > 300 and len(model._ep_info_r) == 100
    win = float(np.mean(model._ep_info_r))
    assert abs(model.logger_values["rollout/ep_rew_mean"] - win) < 1e-12
    assert min(rows) - 1e-9 <= win <= max(rows) + 1e-9 and all(any(abs(x - r) < 1e-6 for r in set(round(v, 6) for v in rows)) for x in set(model._ep_info_r))
    env.close()


def test_per_env_monitor_files_above_64_envs_are_opt_in(tmp_path):
    """The reference wraps EVERY env of the vector in its own Monitor (`<rank>.monitor.csv`, training.py:84-86).  Up to 64 envs the engine
    writes those files by default; above that it writes one file for the vector unless the limit is raised (PPO.monitor_per_env_limit /
    TMA_MONITOR_PER_ENV_LIMIT): then every env gets its file, its rows are that env's episodes in the order they finished, and all files
    together hold what the single file would -- the same multiset of (return, length) rows."""
    from three_mlagents_amd.harness import make_vector_env
    from three_mlagents_amd.ppo import PPO

    def run(sub, limit):
        env = make_vector_env("gridworld", n_envs=160, seed=5, monitor_dir=str(tmp_path / sub))
        model = PPO("MlpPolicy", env, n_steps=64, batch_size=1024, n_epochs=1, seed=5, policy_kwargs={"net_arch": [64, 64]})
        if limit:
            model.monitor_per_env_limit = limit
        model.learn(160 * 64 * 3)
        env.close()
        files = sorted((tmp_path / sub).iterdir(), key=lambda p: int(p.name.split(".")[0]))
        rows = {}
        for p in files:
            lines = p.read_text().splitlines()
            assert lines[0].startswith("#{") and lines[1] == "r,l,t"
            rows[int(p.name.split(".")[0])] = [tuple(ln.split(",")[:2]) for ln in lines[2:] if not ln.startswith("#")]
        return rows

    one = run("single", None)
    per = run("per_env", 4096)
    assert list(one) == [0] and list(per) == list(range(160))
    assert sorted(x for v in per.values() for x in v) == sorted(one[0]) and len(one[0]) > 300
    assert sum(1 for v in per.values() if v) > 100  # (most envs finished at least one episode in 192 steps)
    ts = [float(ln.split(",")[2]) for ln in (tmp_path / "per_env" / "7.monitor.csv").read_text().splitlines()[2:] if not ln.startswith("#")]
    assert ts == sorted(ts)
