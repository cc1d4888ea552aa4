"""Host-side contract of the drop-in surface that needs no GPU: task resolution and its error types, the request/result records,
the PPO default table, policy-file lookup, the runner's flag grammar, declared spaces.  The behaviours asserted are the ones the
reference's own tests pin (/root/reference/backend/tests/test_mlagents.py:24-30,47-49,105-122)."""
import dataclasses

import numpy as np
import pytest

from three_mlagents_amd import harness, tasks


def test_engine_task_table():
    assert set(tasks.ENGINE_TASKS) == {"basic", "gridworld", "ball3d", "push", "ant", "walljump"}
    for t in tasks.ENGINE_TASKS.values():
        card = t.card()
        assert card["trainable"] is True and card["id"] == t.id and card["policy_prefix"].endswith("_policy")
    assert tasks.resolve("gridworld").ppo_n_steps == 1024 and tasks.resolve("push").ppo_n_steps == 2048  # foundation vs benchmark tier
    assert tasks.resolve("ant").pinned is False and tasks.resolve("gridworld").pinned  # crawler dynamics are build-defined
    # library facts ride on the card when the .so is built (it is, in this repo)
    assert tasks.resolve("basic").card()["obs_dim"] == 21 and tasks.resolve("ant").card()["act_dim"] == 20


def test_name_resolution_and_error_types():  # test_mlagents.py:47-49 + registry.py:359-369
    assert tasks.resolve("Crawler").id == "ant" and tasks.resolve("GRIDWORLD").kernel == "gridworld"
    with pytest.raises(KeyError):
        tasks.resolve("not-a-task")
    for ref_only in ("bicycle", "brick-break", "self_driving_car", "fish"):
        with pytest.raises(ValueError):
            tasks.resolve(ref_only)
    with pytest.raises(ValueError):
        tasks.make_env("bicycle")


def test_predict_requires_model_file(tmp_path, monkeypatch):  # test_mlagents.py:105-108
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        harness.predict_action("basic", np.zeros(21, dtype=np.float32), "missing.zip")
    with pytest.raises(FileNotFoundError):
        harness.find_policy(tasks.resolve("basic"))  # no zip at all
    assert not (tmp_path / "policies").exists()  # lookups never create directories


def test_policy_lookup_accepts_path_or_name(tmp_path, monkeypatch):  # test_mlagents.py:110-122
    monkeypatch.chdir(tmp_path)
    d = tmp_path / "policies"
    d.mkdir()
    (d / "basic_policy_a.zip").write_bytes(b"x")
    (d / "basic_policy_b.zip").write_bytes(b"x")
    t = tasks.resolve("basic")
    assert harness.find_policy(t, str(d / "basic_policy_a.zip")) == d / "basic_policy_a.zip"
    assert harness.find_policy(t, "basic_policy_a.zip").name == "basic_policy_a.zip"
    assert harness.find_policy(t).name == "basic_policy_b.zip"  # newest by name (run ids start with a timestamp)


def test_ppo_default_table():  # value table training.py:361-391
    kw = harness.ppo_defaults(tasks.resolve("gridworld"))
    assert (kw["n_steps"], kw["batch_size"], kw["n_epochs"], kw["learning_rate"]) == (1024, 256, 10, 3e-4)
    assert (kw["gamma"], kw["gae_lambda"], kw["clip_range"], kw["ent_coef"], kw["vf_coef"], kw["max_grad_norm"]) == (0.99, 0.95, 0.2, 0.01, 0.5, 0.5)
    assert kw["policy_kwargs"] == {"net_arch": {"pi": [256, 256], "vf": [256, 256]}}
    assert harness.ppo_defaults(tasks.resolve("push"))["n_steps"] == 2048 and "ppo" in harness.ALGORITHMS


def test_records_and_train_errors():  # training.py:40-68,105-114
    cfg = harness.TrainConfig("basic")
    assert (cfg.seed, cfg.eval_freq, cfg.deterministic_eval, cfg.save_policy, cfg.verbose, cfg.algorithm) == (1, 10_000, True, True, 1, None)
    assert [f.name for f in dataclasses.fields(harness.TrainResult)][:4] == ["task_id", "algorithm", "run_id", "model_filename"]
    with pytest.raises(dataclasses.FrozenInstanceError):
        cfg.seed = 2
    with pytest.raises(ValueError):
        harness.train_task(harness.TrainConfig("fish"))
    with pytest.raises(ValueError):
        harness.train_task(harness.TrainConfig("basic", algorithm="sarsa"))
    with pytest.raises(ValueError):
        harness.train_task(harness.TrainConfig("basic", algorithm="dqn"))  # known to SB3, not on the engine
    with pytest.raises(KeyError):
        harness.train_task(harness.TrainConfig("nope"))


def test_runner_grammar():  # cli.py:14-41
    from three_mlagents_amd.__main__ import parser

    a = parser().parse_args(["train", "basic", "--algorithm", "ppo", "--n-envs", "8", "-t", "1000", "--quiet"])
    assert (a.command, a.task, a.algorithm, a.n_envs, a.timesteps, a.seed, a.eval_freq, a.quiet) == ("train", "basic", "ppo", 8, 1000, 1, 10_000, True)
    a = parser().parse_args(["evaluate", "gridworld", "m.zip", "--stochastic"])
    assert (a.seed, a.stochastic, a.episodes) == (10_001, True, None)


def test_spaces_match_reference_declarations():  # envs.py:38-44,166-199
    from three_mlagents_amd.spaces import task_spaces

    for name, (d, n) in {"basic": (21, 3), "gridworld": (4, 5), "ball3d": (6, 5), "push": (4, 5), "walljump": (4, 4)}.items():
        obs_space, act_space = task_spaces(name)
        assert obs_space.shape == (d,) and obs_space.dtype == np.float32 and act_space.n == n
        assert act_space.contains(act_space.sample()) and not act_space.contains(n)
    obs_space, _ = task_spaces("gridworld")
    assert obs_space.contains(np.array([0.25, -0.75, 1, 0], np.float32)) and not obs_space.contains(np.array([2, 0, 0, 0], np.float32))
