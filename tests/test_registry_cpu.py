"""Mirror of the reference's registry / path tests (/root/reference/backend/tests/test_mlagents.py:24-30,47-49,105-122)
against the drop-in surface; no GPU needed."""
import numpy as np
import pytest

from three_mlagents_amd.registry import get_task, list_task_cards, list_tasks, make_env
from three_mlagents_amd.training import ALGORITHMS, POLICIES_DIR, TrainConfig, _default_model_kwargs, _default_policy, _resolve_model_path, predict_action, train_task


def test_trainable_tasks_have_factories():  # test_mlagents.py:25-30
    trainable = list_tasks(include_roadmap=False)
    assert len(trainable) >= 5
    for task in trainable:
        assert task.trainable and task.card()["trainable"] is True
    assert {t.id for t in trainable} == {"basic", "gridworld", "ball3d", "push", "ant", "walljump"}
    assert len(list_task_cards()) == 19 and "env_factory" not in list_task_cards()[0]


def test_alias_resolution():  # test_mlagents.py:47-49
    assert get_task("brick-break").id == "brickbreak"
    assert get_task("self_driving_car").id == "self-driving-car"
    with pytest.raises(KeyError):
        get_task("not-a-task")  # registry.py:359-362
    with pytest.raises(ValueError):
        make_env("bicycle")  # registry.py:368-369: registered but not trainable here


def test_predict_requires_model_file():  # test_mlagents.py:105-108
    with pytest.raises(FileNotFoundError):
        predict_action("basic", np.zeros(21, dtype=np.float32), "missing.zip")


def test_model_path_resolver_accepts_policy_relative_path():  # test_mlagents.py:110-122
    task = get_task("basic")
    model_path = POLICIES_DIR / "resolver_regression_test.zip"
    try:
        model_path.parent.mkdir(parents=True, exist_ok=True)
        model_path.write_bytes(b"placeholder")
        assert _resolve_model_path(task, str(model_path)) == model_path
        assert _resolve_model_path(task, model_path.name) == model_path
    finally:
        model_path.unlink(missing_ok=True)


def test_default_ppo_kwargs_match_reference():  # training.py:361-391
    kw = _default_model_kwargs("ppo", train_env=None, task=get_task("gridworld"), total_timesteps=64, tensorboard_log="/tmp/tb", verbose=0)
    assert (kw["n_steps"], kw["batch_size"], kw["n_epochs"], kw["learning_rate"]) == (1024, 256, 10, 3e-4)
    assert (kw["gamma"], kw["gae_lambda"], kw["clip_range"], kw["ent_coef"], kw["vf_coef"], kw["max_grad_norm"]) == (0.99, 0.95, 0.2, 0.01, 0.5, 0.5)
    assert kw["policy_kwargs"] == {"net_arch": {"pi": [256, 256], "vf": [256, 256]}}
    assert _default_model_kwargs("ppo", train_env=None, task=get_task("push"), total_timesteps=64, tensorboard_log="/tmp/tb", verbose=0)["n_steps"] == 2048
    assert _default_policy(get_task("basic")) == "MlpPolicy" and "ppo" in ALGORITHMS


def test_train_task_error_behaviour():  # training.py:105-114
    with pytest.raises(ValueError):
        train_task(TrainConfig(task_id="fish"))
    with pytest.raises(ValueError):
        train_task(TrainConfig(task_id="basic", algorithm="sarsa"))
    with pytest.raises(KeyError):
        train_task(TrainConfig(task_id="nope"))


def test_cli_grammar():  # cli.py:14-41
    from three_mlagents_amd.cli import build_parser

    a = build_parser().parse_args(["train", "basic", "--algorithm", "ppo", "--n-envs", "8", "-t", "1000", "--quiet"])
    assert (a.command, a.task, a.algorithm, a.n_envs, a.timesteps, a.seed, a.eval_freq, a.quiet) == ("train", "basic", "ppo", 8, 1000, 1, 10_000, True)
    a = build_parser().parse_args(["evaluate", "gridworld", "m.zip", "--stochastic"])
    assert (a.seed, a.stochastic, a.episodes) == (10_001, True, None)


def test_spaces_match_reference_declarations():  # envs.py:38-44,166-199
    from three_mlagents_amd.spaces import task_spaces

    for name, (d, n) in {"basic": (21, 3), "gridworld": (4, 5), "ball3d": (6, 5), "push": (4, 5), "walljump": (4, 4)}.items():
        obs_space, act_space = task_spaces(name)
        assert obs_space.shape == (d,) and obs_space.dtype == np.float32 and act_space.n == n
        assert act_space.contains(act_space.sample()) and not act_space.contains(n)
    obs_space, _ = task_spaces("gridworld")
    assert obs_space.contains(np.array([0.25, -0.75, 1, 0], np.float32)) and not obs_space.contains(np.array([2, 0, 0, 0], np.float32))
