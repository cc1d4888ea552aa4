"""three-mlagents_amd: MI355X-native vectorized-env + PPO engine for the three-mlagents Gymnasium tasks.

The compute is libtma_hip.so (C ABI: include/tma.h).  Host side: `vec_env` (SB3 VecEnv / Gymnasium VectorEnv surfaces), `envs`
(single Gymnasium-shaped env), `ppo` (SB3-shaped PPO), `harness` (make_vector_env / train_task / evaluate_model), `tasks`
(the engine's task table).  Import as `three_mlagents_amd` (alias package at the repo root; the directory name has a hyphen).
"""
__version__ = "0.2.0"

from .tasks import ENGINE_TASKS, EngineTask, make_env, resolve  # noqa: E402

__all__ = ["ENGINE_TASKS", "EngineTask", "make_env", "resolve", "__version__"]
