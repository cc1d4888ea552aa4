"""Train / evaluate harness over the HIP engine: the callers either side of `model.learn()`.

Seams: S2 `make_vector_env(task_id, *, n_envs, seed, monitor_dir)` and S3 `ALGORITHMS["ppo"]`, plus a short
`train_task` / `evaluate_model` / `predict_action` harness with the reference's call signatures, artefact layout
(`policies/<prefix>_<run_id>.zip`, `runs/<task>/<run_id>/{monitor,eval,tb}/`, `metadata.json`) and error types, so that callers
written against /root/reference/backend/mlagents/training.py (`cli.py:70-95`, `main.py:149-184`) keep working.  It is written from
that interface description, not from the reference's source; the reference's own orchestration module can equally stay in place
with three one-line substitutions.
"""
from __future__ import annotations

import contextlib
import dataclasses
import json
import os
import secrets
import statistics
import time
from pathlib import Path
from typing import Any

import numpy as np

from . import tasks
from .callbacks import CallbackList, EvalCallback
from .evaluation import evaluate_policy
from .ppo import PPO
from .vec_env import HipVecEnv

POLICIES_DIR, RUNS_DIR = (Path(name) for name in ("policies", "runs"))
ALGORITHMS = dict(ppo=PPO)  # the one algorithm north_star puts on the GPU
EVAL_ENVS = int(os.environ.get("TMA_EVAL_ENVS", "128"))  # width of the evaluation vector (train_task / evaluate_model): up to one env per episode
_SB3_NAMES = {"a2c", "dqn", "ppo", "sac", "td3"}  # what the reference's table accepts (training.py:31-37)

# (name, type, default) -- the reference's request / result records (training.py:40-68); `device` is an engine-only extra
_REQUEST = [("task_id", str), ("total_timesteps", "int | None", None), ("algorithm", "str | None", None), ("seed", int, 1),
            ("n_envs", "int | None", None), ("eval_episodes", "int | None", None), ("eval_freq", int, 10_000), ("deterministic_eval", bool, True),
            ("policy", "str | None", None), ("run_name", "str | None", None), ("save_policy", bool, True), ("verbose", int, 1),
            ("device", "str | None", None)]
_RESULT = [("task_id", str), ("algorithm", str), ("run_id", str), ("model_filename", str), ("model_path", str), ("run_dir", str),
           ("mean_reward", float), ("std_reward", float), ("eval_episodes", int), ("total_timesteps", int), ("metadata_path", str)]
TrainConfig = dataclasses.make_dataclass("TrainConfig", [f if len(f) == 2 else (f[0], f[1], dataclasses.field(default=f[2])) for f in _REQUEST], frozen=True)
TrainResult = dataclasses.make_dataclass("TrainResult", _RESULT, frozen=True)


def make_vector_env(task_id, *, n_envs, seed, monitor_dir=None, device=None, env_offset=0):
    """Seam S2.  Env `rank` of the vector starts from seed + env_offset + rank; Monitor sums are kept by the step kernel."""
    venv = HipVecEnv(tasks.resolve(task_id).kernel, int(n_envs), seed=seed, device=device, env_offset=env_offset)
    venv.monitor_dir = monitor_dir
    return venv


def ppo_defaults(task: tasks.EngineTask, n_envs: int | None = None) -> dict[str, Any]:
    """Hyper-parameters the reference hands to SB3's PPO (value table: training.py:361-391).

    `batch_size`: the reference's 256 (training.py:379) is 32 samples per env of the 8 envs it trains with.  With `n_envs` given, the table
    scales the literal number by the env count over the reference's 8 -- `256 * max(1, n_envs // 8)`: the same minibatches-per-epoch as the
    reference whatever the task's n_steps (32 at n_steps 1024, 64 at 2048), never below the literal 256 -- because at 4096 envs a batch of 256
    would be 163 840 optimizer steps per iteration on minibatches that fill 3 % of the GPU.  Exactly the reference's 256 at up to 15 envs for
    EVERY task; TMA_LITERAL_BATCH=1 (or an explicit model_kwargs["batch_size"]) selects 256 at any size.  `train_task` records the value that
    was used and the switch in metadata.json (`schedule`)."""
    width = [256, 256]
    knob = os.environ.get("TMA_MFMA_DTYPE", "").lower()  # engine knob: bf16 (BASELINE configs[2]) or bf16x3 (the f32 update as a three-term bf16 split)
    extra = {"mfma_dtype": knob} if knob in ("bf16", "bf16x3") else {}
    batch = 256
    if n_envs is not None and not os.environ.get("TMA_LITERAL_BATCH"):
        batch = 256 * max(1, int(n_envs) // 8)
    return dict(learning_rate=3e-4, n_steps=task.ppo_n_steps, batch_size=batch, n_epochs=10, gamma=0.99, gae_lambda=0.95, clip_range=0.2,
                ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, policy_kwargs={"net_arch": {"pi": width, "vf": list(width)}, **extra})


def _algorithm_for(requested: str | None, task: tasks.EngineTask | None = None) -> tuple[str, str | None]:
    """-> (engine algorithm, name it stands in for).  An explicit non-PPO request is an error; a task whose catalogue default is
    another SB3 algorithm (the reference defaults basic / gridworld / push / walljump to DQN with one env) runs the engine's PPO and
    the substitution is returned -- train_task writes it into metadata.json (`substituted_for`) and warns when verbose."""
    if requested is None:
        default = getattr(task, "default_algorithm", "ppo")
        return "ppo", (default if default != "ppo" else None)
    want = requested.lower()
    if want in ALGORITHMS:
        return want, None
    known = "an SB3 algorithm without an MI355X implementation" if want in _SB3_NAMES else "not a known algorithm"
    raise ValueError(f"Unsupported algorithm '{requested}' ({known}). Use one of {sorted(ALGORITHMS)}.")


class _Run:
    """Directory layout of one training run."""

    def __init__(self, task: tasks.EngineTask, run_id: str):
        self.id, self.root = run_id, RUNS_DIR / task.id / run_id
        self.monitor, self.eval, self.tb, self.best = (self.root / d for d in ("monitor", "eval", "tb", "best_model"))
        for d in (POLICIES_DIR, self.monitor, self.eval, self.tb):
            os.makedirs(d, exist_ok=True)
        self.zip_name = f"{task.policy_prefix}_{run_id}.zip"
        self.zip_path = POLICIES_DIR / self.zip_name
        self.metadata = self.root / "metadata.json"


def train_task(config, *, callback=None, model_kwargs=None):
    task = tasks.resolve(config.task_id)
    algo, stands_in_for = _algorithm_for(config.algorithm, task)
    if stands_in_for and config.verbose:
        import warnings

        warnings.warn(f"task '{task.id}': the reference's default algorithm is {stands_in_for.upper()}; this engine trains PPO instead "
                      f"(recorded as substituted_for in metadata.json)", stacklevel=2)
    budget = int(config.total_timesteps or task.total_timesteps)
    n_envs = int(config.n_envs or task.n_envs)
    episodes = int(config.eval_episodes or task.eval_episodes)
    run = _Run(task, config.run_name or f"{task.id}_{algo}_{time.strftime('%Y%m%d_%H%M%S')}_{secrets.token_hex(4)}")
    with contextlib.ExitStack() as stack:
        def opened(n, seed_shift, mon=None):
            e = make_vector_env(task.id, n_envs=n, seed=config.seed + seed_shift, monitor_dir=mon, device=config.device)
            stack.callback(e.close)
            return e

        # the eval vector is seeded 10 000 past the training seed (the reference: ONE env, episodes one after the other on the host; here the
        # episodes are spread over up to EVAL_ENVS device envs stepped together -- evaluation.py, SB3's even split of episodes over envs)
        venv, eval_env = opened(n_envs, 0, run.monitor), opened(max(1, min(EVAL_ENVS, episodes)), 10_000)
        hp = {**ppo_defaults(task, n_envs), "tensorboard_log": str(run.tb), "verbose": config.verbose, **(model_kwargs or {})}
        model = ALGORITHMS[algo](config.policy or "MlpPolicy", venv, seed=config.seed, **hp)
        ev = dict(n_eval_episodes=episodes, deterministic=config.deterministic_eval)
        hooks = [EvalCallback(eval_env, log_path=str(run.eval), best_model_save_path=str(run.best), verbose=config.verbose,
                              eval_freq=max(config.eval_freq // n_envs, 1), **ev)]  # eval_freq counts vector steps
        hooks += [callback] if callback is not None else []
        model.learn(total_timesteps=budget, callback=CallbackList(hooks), progress_bar=False)
        if config.save_policy:
            model.save(run.zip_path)
        returns, lengths = evaluate_policy(model, eval_env, return_episode_rewards=True, **ev)
        mean, std = statistics.fmean(returns), statistics.pstdev(returns)
        from . import __version__

        schedule = dict(batch_size=model.batch_size, n_steps=model.n_steps, n_epochs=model.n_epochs, n_envs=n_envs,
                        minibatches_per_epoch=-(-n_envs * model.n_steps // model.batch_size), reference_batch_size=256,
                        literal_batch_env=bool(os.environ.get("TMA_LITERAL_BATCH")), batch_size_from="model_kwargs" if "batch_size" in (model_kwargs or {})
                        else ("TMA_LITERAL_BATCH" if os.environ.get("TMA_LITERAL_BATCH") else "256 * max(1, n_envs // 8)"))
        record = dict(task=task.card(), config=dataclasses.asdict(config), algorithm=algo, substituted_for=stands_in_for, schedule=schedule, run_id=run.id, model_filename=run.zip_name,
                      model_path=str(run.zip_path), mean_reward=mean, std_reward=std, episode_rewards=[float(r) for r in returns],
                      episode_lengths=[int(n) for n in lengths], train_log=model.logger_values,
                      data_parallel=dict(world_size=getattr(model, "world_size", 1), allreduce_path=getattr(model, "allreduce_path", "none")),
                      software=dict(three_mlagents_amd=__version__, engine="libtma_hip.so (gfx950)"), created_at=time.strftime("%Y-%m-%dT%H:%M:%S%z"))
        with open(run.metadata, "w", encoding="utf-8") as fh:
            json.dump(record, fh, indent=2, default=str)
    return TrainResult(task.id, algo, run.id, run.zip_name, str(run.zip_path), str(run.root), mean, std, episodes, budget, str(run.metadata))


def find_policy(task: tasks.EngineTask, name_or_path=None) -> Path:
    """A path, a file name under policies/, or None for the newest zip of the task; FileNotFoundError otherwise."""
    if name_or_path is None:
        newest = max(POLICIES_DIR.glob(f"{task.policy_prefix}_*.zip"), default=None, key=lambda p: p.name)
        if newest is None:
            raise FileNotFoundError(f"No policy zip for task '{task.id}' under {POLICIES_DIR}/.")
        return newest
    for cand in (Path(name_or_path), POLICIES_DIR / str(name_or_path)):
        if cand.is_file():
            return cand
    raise FileNotFoundError(f"Model not found: {name_or_path}")


def load_model(task, name_or_path=None):
    task = tasks.resolve(task) if isinstance(task, str) else task
    return PPO.load(find_policy(task, name_or_path))


def evaluate_model(task_id, name_or_path, *, episodes=None, deterministic=True, seed=10_001):
    """-> dict(task_id, model, episodes, mean_reward, std_reward, episode_rewards, episode_lengths)"""
    task = tasks.resolve(task_id)
    path = find_policy(task, name_or_path)
    model = PPO.load(path)
    n = int(episodes or task.eval_episodes)
    with contextlib.closing(make_vector_env(task.id, n_envs=max(1, min(EVAL_ENVS, n)), seed=seed)) as env:
        returns, lengths = evaluate_policy(model, env, n_eval_episodes=n, deterministic=deterministic, return_episode_rewards=True)
    return dict(task_id=task.id, model=str(path), episodes=n, mean_reward=statistics.fmean(returns), std_reward=statistics.pstdev(returns),
                episode_rewards=[float(r) for r in returns], episode_lengths=[int(k) for k in lengths])


def predict_action(task_id, obs, model_filename=None):
    """Deterministic action for one observation: int for Discrete tasks, list of floats for Box tasks."""
    action, _ = load_model(task_id, model_filename).predict(np.float32(obs), deterministic=True)
    return action.tolist() if getattr(action, "ndim", 0) else int(action)
