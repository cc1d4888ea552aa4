"""The tasks the HIP engine implements, and how task names resolve to them.

This is NOT a mirror of the reference's registry module: that file (a catalogue of 19 demos with titles, tags and publication
notes, /root/reference/backend/mlagents/registry.py) stays in the reference.  A maintainer keeps it and only points the
`env_factory` of the rows below at `three_mlagents_amd.envs.make_*_env`.  What the engine itself needs per
task is small: the kernel name, the file-name prefix of policy zips, and the defaults the reference's `train_task` would pick
(timesteps / eval episodes / env count: registry.py:52-116,225-240; PPO n_steps by research tier: training.py:362).  Spaces and
episode limits are not repeated here -- they come from the library (`tma_task_*`, include/tma.h).

Error behaviour kept from the reference's lookup (registry.py:359-369): unknown name -> KeyError; a name the reference knows but
the engine has no kernel for -> ValueError.
"""
from __future__ import annotations

from typing import Any, NamedTuple


class EngineTask(NamedTuple):
    id: str               # the reference's task id
    kernel: str           # task name inside libtma_hip.so (tma_task_id)
    policy_prefix: str    # policies/<prefix>_<run_id>.zip
    research_tier: str    # "foundation" -> PPO n_steps 1024, anything else 2048
    total_timesteps: int
    eval_episodes: int
    n_envs: int
    reward_threshold: float | None = None
    pinned: bool = True   # False: dynamics are build-defined, no reference vectors exist
    default_algorithm: str = "ppo"  # the reference's catalogue default (registry.py `default_algorithm`); the engine runs PPO either way

    @property
    def ppo_n_steps(self) -> int:
        return 1024 if self.research_tier == "foundation" else 2048

    trainable = True  # every row of this table has kernels

    def card(self) -> dict[str, Any]:
        lib_facts = {}
        try:  # dims from the library when it is built; the card is still printable without it
            from . import _lib

            L, t = _lib.lib(), _lib.task_id(self.kernel)
            lib_facts = {"obs_dim": L.tma_task_obs_dim(t), "num_actions": L.tma_task_num_actions(t), "act_dim": L.tma_task_act_dim(t),
                         "max_episode_steps": L.tma_task_max_episode_steps(t)}
        except ImportError:
            pass
        return {**self._asdict(), "ppo_n_steps": self.ppo_n_steps, "trainable": True, **lib_facts}


# id  kernel  tier  timesteps  eval_episodes  n_envs  reward_threshold  pinned  default_algorithm (registry.py:59,75,91,107,123,139,154,169,231)
# ("ant": the reference delegates to gymnasium Ant-v5 / MuJoCo, envs.py:274-277.  The engine's `ant` kernel has Ant-v5's SHAPES -- 105 observations in
#  its order, 8 torques -- so a policy zip the reference trained loads and runs; `crawler` is BASELINE.json's 172-observation / 20-action chain.
#  Both run the build's articulated-chain dynamics: parity unpinned)
_ROWS = """
basic      basic      foundation    25000   50  1  0.85  yes  dqn
gridworld  gridworld  foundation   100000  100  1  0.75  yes  dqn
ball3d     ball3d     foundation   150000   30  8  150   yes  ppo
push       push       benchmark    200000  100  1  0.65  yes  dqn
walljump   walljump   benchmark    150000  100  1  0.7   yes  dqn
brickbreak brickbreak benchmark    500000   50  8  -     yes  ppo
bicycle    bicycle    benchmark    500000   50  8  -     yes  ppo
glider     glider     frontier    1000000   50  8  -     yes  ppo
ant        ant        benchmark   3000000   20  8  -     no  ppo
crawler    crawler    benchmark   3000000   20  8  -     no  ppo
"""


def _parse(rows: str) -> dict[str, EngineTask]:
    table = {}
    for line in rows.split("\n"):
        if line.strip():
            tid, kern, tier, steps, episodes, envs, thr, pinned, algo = line.split()
            table[tid] = EngineTask(tid, kern, tid + "_policy", tier, int(steps), int(episodes), int(envs), None if thr == "-" else float(thr),
                                    pinned == "yes", algo)
    return table


ENGINE_TASKS = _parse(_ROWS)

_SPELLINGS: dict[str, str] = {}
# ids the reference's catalogue also lists; no kernels here (out of the hot-path scope)
_REFERENCE_ONLY = frozenset("labyrinth astrodynamics kraken worm foodcollector intersection minecraft simcity fish "
                            "self-driving-car".split())


def canonical(name: str) -> str:
    key = "-".join(str(name).strip().lower().split("_"))
    if key in ("brick-break", "food-collector"):
        key = key.replace("-", "")
    return _SPELLINGS.get(key, key)


def resolve(name: str) -> EngineTask:
    key = canonical(name)
    hit = ENGINE_TASKS.get(key)
    if hit is not None:
        return hit
    if key in _REFERENCE_ONLY:
        raise ValueError(f"Task '{name}' has no MI355X kernels: it is not a Gymnasium/SB3 trainable task on this engine.")
    raise KeyError(f"Unknown task '{name}'. Engine tasks: {', '.join(sorted(ENGINE_TASKS))}")


def make_env(name: str):
    """Single Gymnasium-shaped env (seam S1: what `TaskSpec.env_factory()` returns)."""
    from .envs import HipSingleEnv

    return HipSingleEnv(resolve(name).kernel)
