"""Progress / run streaming for the browser demos, on top of the engine.

The reference streams three kinds of JSON messages over an accepted FastAPI WebSocket
(/root/reference/backend/mlagents/websocket_training.py): `progress` while `model.learn` runs (its callback, lines 19-51), `trained`
with the artefact names when it returns (lines 98-112), and `run_step` frames (an optional `state` from `env.get_state_for_viz()`)
while a saved policy drives one visualisation env (lines 141-185).  This module emits the same message shapes from the engine's
callback protocol and device envs.  It never imports a web framework: `websocket` is any object with an awaitable `send_json(dict)`;
a `closed()` predicate (or an `application_state` whose name is not exactly CONNECTED, as starlette's has) stops the run loop.
"""
from __future__ import annotations

import asyncio
import dataclasses
from typing import Any, Callable

import numpy as np

from . import harness, tasks
from .callbacks import BaseCallback

# names of the flat state words the engine reads back per task (csrc/tma_tasks.h to_flat; oracle/tma_oracle.c layout)
STATE_FIELDS = {
    "basic": ("position", "steps"),
    "gridworld": ("agentX", "agentY", "greenX", "greenY", "redX", "redY", "goalType", "steps"),
    "push": ("agentX", "agentY", "boxX", "boxY", "goalX", "steps"),
    "ball3d": ("rotX", "rotZ", "ballX", "ballZ", "velX", "velZ", "steps", "first"),
    "walljump": ("x", "inAir", "wall", "steps"),
    "bicycle": ("x", "z", "theta", "phi", "phiDot", "delta", "goalX", "goalZ", "distToGoal", "steps"),
    "brickbreak": ("paddleX", "ballX", "ballY", "velX", "velY", "steps") + tuple(f"brick{k}" for k in range(40)),
    "glider": ("posX", "posY", "posZ", "velX", "velY", "velZ", "roll", "pitch", "yaw", "rollRate", "pitchRate", "yawRate", "waypoint", "steps"),
}


def state_for_viz(env, index: int = 0) -> dict[str, Any]:
    """Host readback of ONE env of a device vector: named state words (and, for the Crawler-shape task, the three fields the reference's
    MuJoCo wrapper reports, backend/examples/crawler.py:56-85: base position, orientation quaternion, the first eight joint angles)."""
    eng = getattr(env, "engine", env)
    row = eng.get_state()[index].cpu().numpy()
    name = eng.task_name
    if name in STATE_FIELDS:
        return {k: (int(v) if float(v).is_integer() else float(v)) for k, v in zip(STATE_FIELDS[name], row)}
    nj = (len(row) - 9) // 3  # joints, velocities, previous actions, 8 root words, step count
    root = row[3 * nj:3 * nj + 8]
    return {"basePos": [float(root[0]), float(root[1]), float(root[2])], "baseOri": [1.0, 0.0, 0.0, 0.0],
            "jointAngles": [float(x) for x in row[:8]], "steps": int(row[-1])}


class ProgressCallback(BaseCallback):
    """`progress` frames every `progress_freq` timesteps.  `emit(dict)` must be safe to call from the training thread, e.g.
    `lambda p: asyncio.run_coroutine_threadsafe(ws.send_json(p), loop)`."""

    def __init__(self, emit: Callable[[dict], Any], total_timesteps: int, progress_freq: int = 2_000):
        super().__init__()
        self.emit, self.total, self.every, self._last = emit, max(1, int(total_timesteps)), max(1, int(progress_freq)), 0

    def frame(self) -> dict:
        return progress_frame(int(self.num_timesteps), self.total, type(self.model).__name__ if self.model is not None else None)

    def _on_step(self) -> bool:
        if self.num_timesteps - self._last >= self.every:
            self._last = self.num_timesteps
            self.emit(self.frame())
        return True


_COPIED_FROM_RESULT = "model_filename algorithm mean_reward std_reward eval_episodes run_dir metadata_path".split()


def progress_frame(timesteps: int, total: int, algorithm, **extra) -> dict:
    """One `progress` message; the reference fills `episode` with the timestep count and leaves reward / loss empty."""
    frame = dict.fromkeys(("reward", "loss"))
    frame.update(type="progress", timesteps=timesteps, episode=timesteps, progress=min(1.0, timesteps / total), algorithm=algorithm, **extra)
    return frame


def trained_frame(result) -> dict:
    """The `trained` message for a harness.TrainResult: artefact names, evaluation summary, and the run id under the two names the
    browser reads it by."""
    r = dataclasses.asdict(result) if dataclasses.is_dataclass(result) else dict(result)
    run_id = str(r["run_id"])
    frame = {k: r[k] for k in _COPIED_FROM_RESULT}
    frame.update(type="trained", file_url="/policies/" + r["model_filename"], timestamp=run_id, session_uuid=run_id.rsplit("_", 1)[-1])
    return frame


async def train_for_websocket(websocket, task_id: str, *, total_timesteps=None, algorithm=None, seed: int = 1, n_envs=None, eval_episodes=None,
                              eval_freq: int = 10_000, progress_freq: int = 2_000, run_name=None, train=harness.train_task) -> dict:
    """Train `task_id` in a worker thread, streaming `progress` frames, then send `trained`.  Returns the result as a dict."""
    task = tasks.resolve(task_id)
    cfg = harness.TrainConfig(task_id, total_timesteps, algorithm, seed, n_envs, eval_episodes, eval_freq, run_name=run_name, verbose=0)
    loop = asyncio.get_running_loop()
    cb = ProgressCallback(lambda p: asyncio.run_coroutine_threadsafe(websocket.send_json(p), loop), total_timesteps or task.total_timesteps, progress_freq)
    await websocket.send_json(progress_frame(0, cb.total, algorithm or "default", task_id=task.id))
    result = await asyncio.to_thread(train, cfg, callback=cb)
    await websocket.send_json(trained_frame(result))
    return dataclasses.asdict(result) if dataclasses.is_dataclass(result) else dict(result)


def _connected(websocket) -> bool:
    closed = getattr(websocket, "closed", None)
    if callable(closed):
        return not closed()
    state = getattr(websocket, "application_state", None)
    if state is None:
        return True
    # the reference loops on `application_state == WebSocketState.CONNECTED` (websocket_training.py:159): an exact comparison --
    # "DISCONNECTED" also ENDS in "CONNECTED"
    return str(getattr(state, "name", state)).rsplit(".", 1)[-1] == "CONNECTED"


async def run_for_websocket(websocket, task_id: str, *, model_filename=None, seed: int = 10_001, sleep_seconds: float = 0.03, max_steps=None,
                            action_transform=None) -> int:
    """Drive ONE device env with a saved policy and stream a `run_step` frame per step (with the env's state) until the socket closes
    or `max_steps` frames were sent.  Returns the number of episodes finished."""
    task = tasks.resolve(task_id)
    model = harness.load_model(task, model_filename)
    transform = action_transform or (lambda a: a)
    env = harness.make_vector_env(task.id, n_envs=1, seed=seed)
    episodes, sent = 0, 0
    try:
        obs = env.reset()
        while _connected(websocket) and (max_steps is None or sent < max_steps):
            action, _ = model.predict(np.asarray(obs, np.float32), deterministic=True)
            obs, _, dones, _ = env.step(np.asarray(transform(action)))
            await websocket.send_json({"type": "run_step", "episode": episodes + 1, "state": state_for_viz(env, 0)})
            sent += 1
            episodes += int(bool(dones[0]))
            await asyncio.sleep(sleep_seconds)
    finally:
        env.close()
    return episodes
