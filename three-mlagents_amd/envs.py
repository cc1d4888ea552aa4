"""Single-environment Gymnasium surface over the HIP engine (a 1-env shard of the vector engine).

Mirrors what the reference's factories return (/root/reference/backend/mlagents/envs.py:162-199,274-277): an object with
`observation_space`, `action_space`, `reset(*, seed=None, options=None) -> (obs, info)` and
`step(action) -> (obs, reward: float, terminated: bool, truncated: bool, info)`.
`reset(seed=s)` reproduces `LegacySingleAgentGymAdapter.reset(seed=s)` (np.random.seed(s), constructor reset, explicit
reset; envs.py:110-123) bit for bit; an unseeded `reset()` starts the next episode of the per-episode seed contract.
"""
from __future__ import annotations

from typing import Any

import numpy as np
import torch

from . import _lib


_REWARD_TABLES: dict[str, dict[float, float]] = {}


def reward_table(task_id: str) -> dict[float, float]:
    """float32 reward (as a Python float) -> the float64 the reference's step() returns for it, for the tasks whose reward takes finitely
    many values.  Each value is formed here by the same float64 operations in the same order as the reference forms it -- Basic
    (backend/mlagents/envs.py:65-72): -0.01, then += 0.1 or += 1.0; GridWorld (examples/gridworld.py:75-90): -0.01 or +-1.0 assigned;
    Push (examples/push.py:77,112-122): -0.01 += 0.05 * d(agent, box) += 0.3 * d(box, goal) with the distance changes in {-1, 0, 1} (a push
    moves the box with the agent, so only one of the two is non-zero), -= 0.05 on a cancelled push (both changes 0), 1.0 assigned on the goal
    row; WallJump (examples/walljump.py:59,77,81,91): -0.01, -= 0.02 (wall) or -= 0.03 (needless jump), 1.0 assigned."""
    vals: list[float] = []
    if task_id == "basic":
        for bonus in (None, 0.1, 1.0):
            r = -0.01
            if bonus is not None:
                r += bonus
            vals.append(r)
    elif task_id == "gridworld":
        vals = [-0.01, 1.0, -1.0]
    elif task_id == "push":
        for d_ab, d_bg in ((0, 0), (-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (-1, 1), (1, -1), (1, 1)):
            r = -0.01
            r += 0.05 * d_ab
            r += 0.3 * d_bg
            vals.append(r)
        r = -0.01
        r += 0.05 * 0
        r += 0.3 * 0
        r -= 0.05
        vals += [r, 1.0]
    elif task_id == "walljump":
        vals = [-0.01, -0.01 - 0.02, -0.01 - 0.03, 1.0]
    table: dict[float, float] = {}
    for v in vals:
        key = float(np.float32(v))
        assert table.setdefault(key, v) == v, (task_id, key)  # two different float64 values behind one float32 would make the table ambiguous
    return table


class HipSingleEnv:
    metadata = {"render_modes": []}
    render_mode = None
    spec = None

    def __init__(self, task: str, *, seed: int = 0, device=None):
        from .vec_env import HipEnvEngine

        self.engine = HipEnvEngine(task, 1, seed=seed, device=device, ring_depth=8)
        self.task_id = self.engine.task_name
        self.observation_space = self.engine.observation_space
        self.action_space = self.engine.action_space
        self.max_episode_steps = self.engine.max_episode_steps
        self.steps = 0
        self._started = False
        self._pending_obs = None  # observation of the auto-started next episode, handed out by the next reset()
        # float64 reward of the last step, written by the step kernel beside its float32 rounding (include/tma.h tma_env_set_reward64): what
        # the reference's env.step hands back for the float64-physics tasks (Bicycle / BrickBreak / Glider), whose rewards are no finite set
        self._rew64 = torch.zeros((1, 1), dtype=torch.float64, device=self.engine.device)
        _lib.check(_lib.lib().tma_env_set_reward64(self.engine._h, _lib.ptr(self._rew64), 1))

    def _info(self, steps: int) -> dict[str, Any]:
        info = {"steps": int(steps)}
        if self.task_id == "basic":
            info["position"] = int(self.engine.get_state()[0, 0].item())
        return info

    def reset(self, *, seed: int | None = None, options: dict[str, Any] | None = None):
        if seed is not None:
            self.engine.seed(int(seed))
            self.action_space.seed(int(seed))
        if seed is not None or not self._started:
            obs = self.engine.reset().cpu().numpy()[0]
            self._started = True
        elif self._pending_obs is not None:
            obs = self._pending_obs
        else:  # reset in the middle of an episode: abandon it and start episode 0 of the current seed again
            obs = self.engine.reset().cpu().numpy()[0]
        self._pending_obs = None
        self.steps = 0
        if self.task_id == "basic" and options and "position" in options:  # envs.py:54-57
            pos = int(np.clip(int(options["position"]), 0, 20))
            self.engine.set_state(np.array([[float(pos), 0.0]]))
            obs = np.zeros(21, np.float32)
            obs[pos] = 1.0
        return obs.astype(np.float32), self._info(0)

    def step(self, action):
        if not self._started:
            raise RuntimeError("step() called before reset()")
        if self.engine.num_actions > 0:
            a = torch.tensor([int(action)], dtype=torch.int64)
        else:
            a = torch.as_tensor(np.asarray(action, np.float32).reshape(1, -1))
        out = self.engine.step(a.to(self.engine.device))
        term = bool(out["term"][0, 0].item())
        trunc = bool(out["trunc"][0, 0].item())
        reward = float(out["rew"][0, 0].item()) if self.task_id == "ball3d" else self._reward64(out)
        self.steps += 1
        if term or trunc:
            obs = out["term_obs"][0, 0].cpu().numpy()
            self._pending_obs = out["obs"][0, 0].cpu().numpy()
            info = {"steps": int(out["ep_len"][0, 0].item())}
            if self.task_id == "basic":
                info["position"] = int(np.argmax(obs))
        else:
            obs = out["obs"][0, 0].cpu().numpy()
            info = self._info(self.steps)
        return obs.astype(np.float32), reward, term, trunc, info

    def _reward64(self, out) -> float:
        """The reference returns Python floats computed in float64 (-0.01, 0.09000000000000001, ...); the kernel's float32 reward is the
        float32 rounding of one of a small finite set of such values per task (`reward_table`): hand back exactly that float64."""
        r32 = float(out["rew"][0, 0].item())
        table = _REWARD_TABLES.get(self.task_id)
        if table is None:
            table = _REWARD_TABLES[self.task_id] = reward_table(self.task_id)
        exact = table.get(r32)
        if exact is not None:
            return exact
        return float(self._rew64[0, 0].item())  # tasks without a finite reward set (float64 physics): the kernel's own float64

    def close(self) -> None:
        if getattr(self, "_rew64", None) is not None:  # the engine must not keep a pointer into a tensor this wrapper is about to drop
            try:
                _lib.check(_lib.lib().tma_env_set_reward64(self.engine._h, None, 0))
            except Exception:  # noqa: BLE001  (an engine that is already closed)
                pass
            self._rew64 = None
        self.engine.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def render(self):
        return None

    @property
    def unwrapped(self):
        return self


def make_basic_env() -> HipSingleEnv:
    return HipSingleEnv("basic")


def make_ball3d_env() -> HipSingleEnv:
    return HipSingleEnv("ball3d")


def make_gridworld_env() -> HipSingleEnv:
    return HipSingleEnv("gridworld")


def make_push_env() -> HipSingleEnv:
    return HipSingleEnv("push")


def make_walljump_env() -> HipSingleEnv:
    return HipSingleEnv("walljump")


def make_brick_break_env() -> HipSingleEnv:
    return HipSingleEnv("brickbreak")


def make_bicycle_env() -> HipSingleEnv:
    return HipSingleEnv("bicycle")


def make_glider_env() -> HipSingleEnv:
    return HipSingleEnv("glider")


def make_ant_env() -> HipSingleEnv:
    """The reference's `ant` factory (envs.py:274-277) by shape: 105 observations in Ant-v5's order, 8 torques; dynamics build-defined."""
    return HipSingleEnv("ant")


def make_crawler_env() -> HipSingleEnv:
    """BASELINE.json configs[4]: the 172-observation / 20-action articulated chain (build-defined, no counterpart in the reference)."""
    return HipSingleEnv("crawler")
