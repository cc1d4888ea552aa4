"""Observation/action space holders with the Gymnasium surface the reference's callers touch.

The reference declares its spaces with `gymnasium.spaces.Box/Discrete` (backend/mlagents/envs.py:38-44,166-199).
When gymnasium is importable those classes are used as-is; this container does not ship it, so the same
attributes/methods (`shape`, `dtype`, `n`, `low`, `high`, `contains`, `sample`, `seed`, `__repr__`) are provided here.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    from gymnasium.spaces import Box, Discrete, Space  # type: ignore
except Exception:  # noqa: BLE001

    class Space:
        def __init__(self, shape=None, dtype=None, seed=None):
            self._shape = None if shape is None else tuple(shape)
            self.dtype = None if dtype is None else np.dtype(dtype)
            self._np_random = np.random.default_rng(seed)

        @property
        def shape(self):
            return self._shape

        def seed(self, seed=None):
            self._np_random = np.random.default_rng(seed)
            return [seed]

        def __contains__(self, x):
            return self.contains(x)

    class Box(Space):
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            if shape is None:
                shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
            super().__init__(shape, dtype, seed)
            self.low = np.full(self._shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, self.dtype)
            self.high = np.full(self._shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, self.dtype)

        def contains(self, x) -> bool:
            x = np.asarray(x)
            if not np.can_cast(x.dtype, self.dtype):
                return False
            return bool(x.shape == self._shape and np.all(x >= self.low) and np.all(x <= self.high))

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._np_random.uniform(lo, hi).astype(self.dtype)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self._shape}, {self.dtype})"

        def __eq__(self, other):
            return isinstance(other, Box) and self._shape == other._shape and np.array_equal(self.low, other.low) and np.array_equal(self.high, other.high)

    class Discrete(Space):
        def __init__(self, n, seed=None, start=0):
            super().__init__((), np.int64, seed)
            self.n = int(n)
            self.start = int(start)

        def contains(self, x) -> bool:
            if isinstance(x, (int, np.integer)):
                v = int(x)
            elif isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
                v = int(x)
            else:
                return False
            return self.start <= v < self.start + self.n

        def sample(self):
            return np.int64(self.start + self._np_random.integers(self.n))

        def __repr__(self):
            return f"Discrete({self.n})"

        def __eq__(self, other):
            return isinstance(other, Discrete) and self.n == other.n and self.start == other.start


def task_spaces(task_name: str):
    """(observation_space, action_space) exactly as the reference's factories declare them
    (backend/mlagents/envs.py:38-44 basic, :166-175 ball3d, :178-187 gridworld, :190-199 push; crawler is the
    BASELINE 172/20 synthetic shape; ant carries the shapes of envs.py:274-277 on the same build-defined dynamics)."""
    if task_name == "basic":
        return Box(0.0, 1.0, shape=(21,), dtype=np.float32), Discrete(3)
    if task_name == "ball3d":
        return Box(-np.inf, np.inf, shape=(6,), dtype=np.float32), Discrete(5)
    if task_name == "gridworld":
        return Box(-1.0, 1.0, shape=(4,), dtype=np.float32), Discrete(5)
    if task_name == "push":
        return Box(-1.0, 1.0, shape=(4,), dtype=np.float32), Discrete(5)
    if task_name == "walljump":  # envs.py:202-211
        return Box(-1.0, 1.0, shape=(4,), dtype=np.float32), Discrete(4)
    if task_name == "bicycle":  # envs.py:228-239
        return Box(-np.inf, np.inf, shape=(7,), dtype=np.float32), Discrete(3)
    if task_name == "brickbreak":  # envs.py:214-225
        return Box(-np.inf, np.inf, shape=(45,), dtype=np.float32), Discrete(3)
    if task_name == "glider":  # envs.py:242-253
        return Box(-np.inf, np.inf, shape=(16,), dtype=np.float32), Discrete(5)
    if task_name == "crawler":  # BASELINE.json configs[4]: 172 observations, 20 actions
        return Box(-np.inf, np.inf, shape=(172,), dtype=np.float32), Box(-1.0, 1.0, shape=(20,), dtype=np.float32)
    if task_name == "ant":  # envs.py:274-277: the spaces gymnasium's Ant-v5 declares (exclude_current_positions_from_observation=True)
        return Box(-np.inf, np.inf, shape=(105,), dtype=np.float32), Box(-1.0, 1.0, shape=(8,), dtype=np.float32)
    raise KeyError(task_name)
