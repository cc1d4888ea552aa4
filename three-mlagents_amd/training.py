"""Training / evaluation orchestration with the reference's API (/root/reference/backend/mlagents/training.py):
`TrainConfig`, `TrainResult`, `make_vector_env`, `make_eval_env`, `train_task`, `evaluate_model`, `load_model`,
`predict_action`, `latest_model_filename`, `_resolve_model_path`, `_infer_algorithm_from_metadata`, `_default_policy`,
`_default_model_kwargs`, `_make_run_id`, `ALGORITHMS`, `POLICIES_DIR`, `RUNS_DIR` -- same names, argument meaning,
artefact layout (`policies/<prefix>_<run_id>.zip`, `runs/<task>/<run_id>/{monitor,eval,tb}/`, `metadata.json`) and error
behaviour.  What sits underneath is the HIP engine instead of DummyVecEnv + SB3.
"""
from __future__ import annotations

import json
import os
import platform
import uuid
from dataclasses import asdict, dataclass
from datetime import datetime, timezone
from pathlib import Path
from typing import Any

import numpy as np

from . import __version__ as _engine_version
from .callbacks import BaseCallback, CallbackList, EvalCallback
from .evaluation import evaluate_policy
from .ppo import PPO
from .registry import TaskSpec, get_task
from .vec_env import HipVecEnv

POLICIES_DIR = Path("policies")
RUNS_DIR = Path("runs")

# the engine accelerates the PPO path named by north_star; the reference's other SB3 algorithms
# (training.py:31-37: a2c, dqn, sac, td3) have no MI355X implementation here.
ALGORITHMS: dict[str, type] = {"ppo": PPO}
_REFERENCE_ALGORITHMS = ("a2c", "dqn", "ppo", "sac", "td3")


@dataclass(frozen=True)
class TrainConfig:  # training.py:40-53 (+ engine-only optional knobs at the end)
    task_id: str
    total_timesteps: int | None = None
    algorithm: str | None = None
    seed: int = 1
    n_envs: int | None = None
    eval_episodes: int | None = None
    eval_freq: int = 10_000
    deterministic_eval: bool = True
    policy: str | None = None
    run_name: str | None = None
    save_policy: bool = True
    verbose: int = 1
    device: str | None = None


@dataclass(frozen=True)
class TrainResult:  # training.py:56-68
    task_id: str
    algorithm: str
    run_id: str
    model_filename: str
    model_path: str
    run_dir: str
    mean_reward: float
    std_reward: float
    eval_episodes: int
    total_timesteps: int
    metadata_path: str


def _engine_task(task: TaskSpec) -> str:
    return "crawler" if task.id == "ant" else task.id


def make_vector_env(task_id: str, *, n_envs: int, seed: int, monitor_dir: Path | None = None, device=None, env_offset: int = 0) -> HipVecEnv:
    """training.py:71-89: n_envs envs, env `rank` seeded `seed + rank`, Monitor bookkeeping built in."""
    task = get_task(task_id)
    if not task.trainable:
        raise ValueError(f"Task '{task_id}' is not a Gymnasium/SB3 trainable task yet.")
    env = HipVecEnv(_engine_task(task), n_envs, seed=seed, device=device, env_offset=env_offset)
    env.monitor_dir = monitor_dir
    return env


def make_eval_env(task_id: str, *, seed: int, device=None) -> HipVecEnv:
    """training.py:92-95: one Monitor-wrapped env reset with `seed`."""
    task = get_task(task_id)
    if not task.trainable:
        raise ValueError(f"Task '{task_id}' is not a Gymnasium/SB3 trainable task yet.")
    return HipVecEnv(_engine_task(task), 1, seed=seed, device=device)


def train_task(config: TrainConfig, *, callback: BaseCallback | None = None, model_kwargs: dict[str, Any] | None = None) -> TrainResult:
    task = get_task(config.task_id)
    if not task.trainable:
        raise ValueError(f"Task '{task.id}' is not trainable through Gymnasium/SB3 yet.")
    requested = (config.algorithm or task.default_algorithm).lower()
    algorithm_name = requested
    substituted = None
    if algorithm_name not in ALGORITHMS:
        if config.algorithm is None and requested in _REFERENCE_ALGORITHMS:
            substituted, algorithm_name = requested, "ppo"  # registry default is an off-path SB3 algorithm: use the engine's PPO
        else:
            raise ValueError(f"Unsupported algorithm '{algorithm_name}'. Use one of {sorted(ALGORITHMS)}.")

    total_timesteps = config.total_timesteps or task.total_timesteps
    n_envs = config.n_envs or task.n_envs
    eval_episodes = config.eval_episodes or task.eval_episodes
    run_id = config.run_name or _make_run_id(task.id, algorithm_name)
    run_dir = RUNS_DIR / task.id / run_id
    monitor_dir, eval_dir, tb_dir = run_dir / "monitor", run_dir / "eval", run_dir / "tb"
    for path in (POLICIES_DIR, run_dir, monitor_dir, eval_dir, tb_dir):
        path.mkdir(parents=True, exist_ok=True)

    train_env = make_vector_env(task.id, n_envs=n_envs, seed=config.seed, monitor_dir=monitor_dir, device=config.device)
    eval_env = make_eval_env(task.id, seed=config.seed + 10_000, device=config.device)
    try:
        policy = config.policy or _default_policy(task)
        algo_cls = ALGORITHMS[algorithm_name]
        kwargs = _default_model_kwargs(algorithm_name, train_env=train_env, task=task, total_timesteps=total_timesteps,
                                       tensorboard_log=str(tb_dir), verbose=config.verbose)
        if model_kwargs:
            kwargs.update(model_kwargs)
        model = algo_cls(policy, train_env, seed=config.seed, **kwargs)
        callbacks: list[Any] = [
            EvalCallback(eval_env, best_model_save_path=str(run_dir / "best_model"), log_path=str(eval_dir),
                         eval_freq=max(1, config.eval_freq // max(1, n_envs)), n_eval_episodes=eval_episodes,
                         deterministic=config.deterministic_eval, verbose=config.verbose, warn=True)
        ]
        if callback is not None:
            callbacks.append(callback)
        model.learn(total_timesteps=total_timesteps, callback=CallbackList(callbacks), progress_bar=False)

        model_filename = f"{task.policy_prefix}_{run_id}.zip"
        model_path = POLICIES_DIR / model_filename
        if config.save_policy:
            model.save(model_path)

        episode_rewards, episode_lengths = evaluate_policy(model, eval_env, n_eval_episodes=eval_episodes, deterministic=config.deterministic_eval,
                                                           return_episode_rewards=True, warn=True)
        mean_reward = float(np.mean(episode_rewards))
        std_reward = float(np.std(episode_rewards))
        metadata = {
            "task": task.card(), "config": asdict(config), "algorithm": algorithm_name, "run_id": run_id, "model_filename": model_filename,
            "model_path": str(model_path), "mean_reward": mean_reward, "std_reward": std_reward,
            "episode_rewards": [float(r) for r in episode_rewards], "episode_lengths": [int(length) for length in episode_lengths],
            "software": {"python": platform.python_version(), "three_mlagents_amd": _engine_version, "engine": "libtma_hip.so (gfx950)"},
            "created_at": datetime.now(timezone.utc).isoformat(),
        }
        if substituted:
            metadata["algorithm_substituted_for"] = substituted
        metadata["train_log"] = model.logger_values
        metadata_path = run_dir / "metadata.json"
        metadata_path.write_text(json.dumps(metadata, indent=2, default=str), encoding="utf-8")
        return TrainResult(task_id=task.id, algorithm=algorithm_name, run_id=run_id, model_filename=model_filename, model_path=str(model_path),
                           run_dir=str(run_dir), mean_reward=mean_reward, std_reward=std_reward, eval_episodes=eval_episodes,
                           total_timesteps=total_timesteps, metadata_path=str(metadata_path))
    finally:
        train_env.close()
        eval_env.close()


def evaluate_model(task_id: str, model_filename_or_path: str, *, episodes: int | None = None, deterministic: bool = True,
                   seed: int = 10_001) -> dict[str, Any]:
    task = get_task(task_id)
    model = load_model(task, model_filename_or_path)
    eval_env = make_eval_env(task.id, seed=seed)
    try:
        n_eval_episodes = episodes or task.eval_episodes
        rewards, lengths = evaluate_policy(model, eval_env, n_eval_episodes=n_eval_episodes, deterministic=deterministic,
                                           return_episode_rewards=True, warn=True)
        return {
            "task_id": task.id, "model": str(_resolve_model_path(task, model_filename_or_path)), "episodes": n_eval_episodes,
            "mean_reward": float(np.mean(rewards)), "std_reward": float(np.std(rewards)),
            "episode_rewards": [float(r) for r in rewards], "episode_lengths": [int(length) for length in lengths],
        }
    finally:
        eval_env.close()


def load_model(task: TaskSpec, model_filename_or_path: str | None = None):
    model_path = _resolve_model_path(task, model_filename_or_path)
    algorithm_name = _infer_algorithm_from_metadata(task, model_path) or task.default_algorithm
    algo_cls = ALGORITHMS.get(algorithm_name, PPO)
    return algo_cls.load(model_path)


def predict_action(task_id: str, obs: np.ndarray, model_filename: str | None = None) -> int | list[float]:
    task = get_task(task_id)
    model = load_model(task, model_filename)
    obs = np.asarray(obs, dtype=np.float32)
    action, _ = model.predict(obs, deterministic=True)
    if isinstance(action, np.ndarray):
        if action.ndim == 0:
            return int(action.item())
        return action.tolist()
    return int(action)


def latest_model_filename(task_id: str) -> str:
    task = get_task(task_id)
    matches = sorted(POLICIES_DIR.glob(f"{task.policy_prefix}_*.zip"), reverse=True)
    if not matches:
        raise FileNotFoundError(f"No SB3 policy zip found for task '{task.id}'.")
    return matches[0].name


def _resolve_model_path(task: TaskSpec, model_filename_or_path: str | None) -> Path:
    if model_filename_or_path is None:
        model_filename_or_path = latest_model_filename(task.id)
    path = Path(model_filename_or_path)
    if path.exists():
        return path
    if not path.is_absolute():
        policy_path = POLICIES_DIR / path
        if policy_path.exists():
            return policy_path
        path = policy_path
    raise FileNotFoundError(f"Model not found: {path}")


def _infer_algorithm_from_metadata(task: TaskSpec, model_path: Path) -> str | None:
    stem = model_path.name.removesuffix(".zip")
    for metadata_path in (RUNS_DIR / task.id).glob("*/metadata.json"):
        try:
            metadata = json.loads(metadata_path.read_text(encoding="utf-8"))
        except json.JSONDecodeError:
            continue
        if metadata.get("model_filename") == model_path.name or metadata.get("run_id") in stem:
            algorithm = metadata.get("algorithm") or metadata.get("config", {}).get("algorithm")
            return (algorithm or task.default_algorithm).lower()
    return None


def _default_policy(task: TaskSpec) -> str:
    return "CnnPolicy" if task.observation == "image" else "MlpPolicy"


def _default_model_kwargs(algorithm_name: str, *, train_env, task: TaskSpec, total_timesteps: int, tensorboard_log: str, verbose: int) -> dict[str, Any]:
    """PPO branch of training.py:361-391 (the other branches configure SB3 algorithms this engine does not provide)."""
    common: dict[str, Any] = {"tensorboard_log": tensorboard_log, "verbose": verbose}
    if algorithm_name == "ppo":
        n_steps = 1024 if task.research_tier == "foundation" else 2048
        if task.observation == "image":
            raise ValueError(f"Task '{task.id}' needs task-specific CNN policy settings.")
        policy_kwargs: dict[str, Any] = {"net_arch": {"pi": [256, 256], "vf": [256, 256]}}
        # engine extension (no counterpart in the reference): TMA_MFMA_DTYPE=bf16 runs the 256-wide GEMMs on the bf16 MFMA
        # (f32 master weights / accumulation; BASELINE.json configs[2]).  Default f32 = the reference's numerics.
        if os.environ.get("TMA_MFMA_DTYPE", "f32").lower() == "bf16":
            policy_kwargs["mfma_dtype"] = "bf16"
        return {**common, "learning_rate": 3e-4, "n_steps": n_steps, "batch_size": 256, "n_epochs": 10, "gamma": 0.99, "gae_lambda": 0.95,
                "clip_range": 0.2, "ent_coef": 0.01, "vf_coef": 0.5, "max_grad_norm": 0.5, "policy_kwargs": policy_kwargs}
    return common


def _make_run_id(task_id: str, algorithm_name: str) -> str:
    timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    return f"{task_id}_{algorithm_name}_{timestamp}_{uuid.uuid4().hex[:8]}"
