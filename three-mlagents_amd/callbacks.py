"""Callback protocol the reference relies on (SB3 BaseCallback / CallbackList / EvalCallback).

The reference builds `CallbackList([EvalCallback(...), user_callback])` (/root/reference/backend/mlagents/training.py:151-170)
and its WebSocket bridge subclasses BaseCallback reading `self.num_timesteps` / `self.model`
(/root/reference/backend/mlagents/websocket_training.py:19-51).  Any object with `init_callback(model)` and
`on_step() -> bool` works; returning False stops training.
"""
from __future__ import annotations

import os
from typing import Any

import numpy as np


class BaseCallback:
    def __init__(self, verbose: int = 0):
        self.model = None
        self.training_env = None
        self.n_calls = 0
        self.num_timesteps = 0
        self.verbose = verbose
        self.locals: dict[str, Any] = {}
        self.globals: dict[str, Any] = {}
        self.parent = None

    def init_callback(self, model) -> None:
        self.model = model
        self.training_env = model.get_env()
        self._init_callback()

    def _init_callback(self) -> None:
        pass

    def on_training_start(self, locals_=None, globals_=None) -> None:
        self.locals, self.globals = locals_ or {}, globals_ or {}
        self.num_timesteps = self.model.num_timesteps
        self._on_training_start()

    def _on_training_start(self) -> None:
        pass

    def on_rollout_start(self) -> None:
        self._on_rollout_start()

    def _on_rollout_start(self) -> None:
        pass

    def _on_step(self) -> bool:
        return True

    def on_step(self) -> bool:
        self.n_calls += 1
        self.num_timesteps = self.model.num_timesteps
        return self._on_step()

    # Engine extension: the rollout runs natively in chunks of many vector steps, so the model reports them in bulk.  The default is SB3's
    # contract replayed step by step (num_timesteps advances by `per_step`, then on_step(); False stops the rollout at that step); a callback
    # that sets `takes_bulk_steps` implements bulk_steps() and is spared the per-step Python round trips (EvalCallback: 512 identical
    # bookkeeping calls per rollout at 4096 envs).
    takes_bulk_steps = False

    def on_steps(self, n: int, per_step: int) -> tuple[int, bool]:
        for k in range(int(n)):
            self.model.num_timesteps += per_step
            if not self.on_step():
                return k + 1, False
        return int(n), True

    def on_rollout_end(self) -> None:
        self._on_rollout_end()

    def _on_rollout_end(self) -> None:
        pass

    def on_training_end(self) -> None:
        self._on_training_end()

    def _on_training_end(self) -> None:
        pass

    def flush(self) -> None:
        """Write out whatever the callback buffers in memory; PPO.learn calls it when training unwinds on an exception (on_training_end does
        not run then).  Engine extension of the SB3 protocol: SB3's own callbacks write as they go."""

    def on_update_queued(self) -> None:
        """Engine extension: PPO.learn calls it right after train() has QUEUED an update -- the GPU is busy for the length of the update, host work
        done here costs no wall time (EvalCallback collects its deferred evaluation)."""


class CallbackList(BaseCallback):
    def __init__(self, callbacks):
        super().__init__()
        self.callbacks = [as_callback(c) for c in callbacks]

    def _init_callback(self) -> None:
        for c in self.callbacks:
            c.init_callback(self.model)

    def _on_training_start(self) -> None:
        for c in self.callbacks:
            c.on_training_start(self.locals, self.globals)

    def _on_rollout_start(self) -> None:
        for c in self.callbacks:
            c.on_rollout_start()

    def _on_step(self) -> bool:
        ok = True
        for c in self.callbacks:
            ok = c.on_step() and ok
        return ok

    def on_steps(self, n: int, per_step: int) -> tuple[int, bool]:
        if not all(getattr(c, "takes_bulk_steps", False) for c in self.callbacks):
            return super().on_steps(n, per_step)
        start = self.model.num_timesteps
        self.n_calls += int(n)
        for c in self.callbacks:
            c.bulk_steps(int(n), int(per_step), start)
        self.model.num_timesteps = start + int(n) * int(per_step)
        self.num_timesteps = self.model.num_timesteps
        return int(n), True

    def _on_rollout_end(self) -> None:
        for c in self.callbacks:
            c.on_rollout_end()

    def _on_training_end(self) -> None:
        for c in self.callbacks:
            c.on_training_end()

    def flush(self) -> None:
        for c in self.callbacks:
            c.flush()

    def on_update_queued(self) -> None:
        for c in self.callbacks:
            c.on_update_queued()


class _Duck(BaseCallback):
    """Adapts any object exposing a subset of the protocol (e.g. an SB3-style callback written against the reference)."""

    def __init__(self, obj):
        super().__init__()
        self.obj = obj

    def _call(self, name, *a):
        fn = getattr(self.obj, name, None)
        return fn(*a) if callable(fn) else None

    def init_callback(self, model) -> None:
        self.model = model
        if self._call("init_callback", model) is None and hasattr(self.obj, "model"):
            self.obj.model = model

    def on_training_start(self, l=None, g=None) -> None:
        self._call("on_training_start", l or {}, g or {})

    def on_rollout_start(self) -> None:
        self._call("on_rollout_start")

    def on_step(self) -> bool:
        if hasattr(self.obj, "num_timesteps"):
            try:
                self.obj.num_timesteps = self.model.num_timesteps
            except AttributeError:
                pass
        r = self._call("on_step")
        return True if r is None else bool(r)

    def on_rollout_end(self) -> None:
        self._call("on_rollout_end")

    def on_training_end(self) -> None:
        self._call("on_training_end")

    def flush(self) -> None:
        self._call("flush")


def as_callback(cb) -> BaseCallback:
    if cb is None:
        return BaseCallback()
    if isinstance(cb, BaseCallback):
        return cb
    if isinstance(cb, (list, tuple)):
        return CallbackList(cb)
    return _Duck(cb)


class EvalCallback(BaseCallback):
    """Periodic deterministic evaluation (SB3 EvalCallback as configured at training.py:152-161): every `eval_freq` calls, run
    `n_eval_episodes`, append a row to `<log_path>/evaluations.npz`, keep the best model.

    The row cadence is SB3's (one row per `eval_freq` calls).  Two things differ in how a row is produced: (1) the evaluation itself is
    device-side (evaluation.py: native rollout chunks over every env of `eval_env`); (2) every evaluation starts from `reset()`, whose
    episode seeds are fixed (seed + env + k * 2^20), so with `deterministic=True` its result is a function of the parameters alone --
    while the optimizer has not stepped since the last evaluation (`model._n_updates` / Adam step unchanged: `on_step` calls inside
    one rollout; with thousands of envs `eval_freq // n_envs` is a handful of vector steps) the previous result is repeated instead
    of re-running identical episodes.  `evaluations.npz` is rewritten after a fresh evaluation (at most every `flush_interval_s`
    seconds: the file grows by hundreds of rows per rollout at thousands of envs) and at training end, not once per repeated row."""

    def __init__(self, eval_env, best_model_save_path=None, log_path=None, eval_freq=10000, n_eval_episodes=5, deterministic=True, verbose=0, warn=True):
        super().__init__(verbose)
        self.eval_env, self.best_model_save_path, self.log_path = eval_env, best_model_save_path, log_path
        self.eval_freq, self.n_eval_episodes, self.deterministic = int(eval_freq), int(n_eval_episodes), deterministic
        self.best_mean_reward = -np.inf
        self.last_mean_reward = -np.inf
        self.evaluations_timesteps, self.evaluations_results, self.evaluations_length = [], [], []
        self.n_fresh_evaluations = 0
        self._cached_key, self._cached = None, None
        self._dirty = False
        self._pending = None  # a deferred evaluation in flight on the model's side stream (_evaluate_fresh)

    def _policy_key(self):
        m = self.model
        return (getattr(m, "_n_updates", None), getattr(m, "_adam_step", None), id(getattr(m, "policy", None)))

    flush_interval_s = 2.0  # evaluations.npz is rewritten whole (SB3 does the same): at most this often while training, and once at the end

    def _flush(self, force: bool = True) -> None:
        import time as _time

        if self._pending is not None:  # rows of the evaluation in flight have no results yet
            if not force:
                return
            self._complete_pending()
        now = _time.monotonic()
        if not force and now - getattr(self, "_last_flush", -1e9) < self.flush_interval_s:
            return
        self._last_flush = now
        if self._dirty and self.log_path is not None:
            os.makedirs(self.log_path, exist_ok=True)
            np.savez(os.path.join(self.log_path, "evaluations"), timesteps=self.evaluations_timesteps, results=self.evaluations_results,
                     ep_lengths=self.evaluations_length)
        self._dirty = False

    takes_bulk_steps = True

    def _on_step(self) -> bool:
        if self.eval_freq > 0 and self.n_calls % self.eval_freq == 0:
            self._tick()
        return True

    def bulk_steps(self, n: int, per_step: int, start_timesteps: int) -> None:
        """The n on_step() calls of a native rollout chunk at once: the calls that fall on the evaluation cadence do exactly what _on_step does
        (same rows with the same timesteps, a fresh evaluation where the optimizer has stepped), the others do nothing but count."""
        first = self.n_calls + 1
        self.n_calls += n
        if self.eval_freq <= 0:
            return
        c = -(-first // self.eval_freq) * self.eval_freq
        while c < first + n:
            self.num_timesteps = self.model.num_timesteps = start_timesteps + (c - first + 1) * per_step
            self._tick()
            c += self.eval_freq
        self.num_timesteps = start_timesteps + n * per_step

    def _evaluate_fresh(self):
        """The deterministic evaluation (and the best-model zip) on the model's SIDE stream, behind the event the last update left: the compute
        stream is already running the next rollout (queued before the callbacks are called), and nothing here writes what it reads."""
        import contextlib

        import torch

        from .evaluation import evaluate_policy

        model = self.model
        side = model.side_stream() if hasattr(model, "side_stream") and getattr(self.eval_env, "device", None) == getattr(model, "device", None) else None
        main = torch.cuda.current_stream(model.device) if side is not None else None
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if side is not None:
                ev = getattr(model, "_update_done", None)
                if ev is not None:
                    side.wait_event(ev)
                else:  # no update yet: the parameters were written by whatever the compute stream did before
                    side.wait_stream(main)
            if side is not None and hasattr(model, "freeze_for_save") and not os.environ.get("TMA_SYNC_EVAL"):
                # DEFERRED: queue the evaluation (reset + the first chunk: one episode per env at the reference's 100 episodes) on a snapshot of the
                # parameters and return; the result is collected when the next fresh evaluation starts, at a flush or at training end -- the host
                # goes straight on to queue train() instead of waiting here for the previous update AND the episodes (TMA_SYNC_EVAL=1: wait).
                from .evaluation import evaluate_policy_begin

                frozen = model.freeze_for_save() if self.best_model_save_path is not None else {"params": model.policy.params.clone()}
                snap = torch.cuda.Event(enable_timing=False)
                snap.record(side)
                main.wait_event(snap)  # train() may overwrite the live buffers once the clones exist
                state = evaluate_policy_begin(model, self.eval_env, n_eval_episodes=self.n_eval_episodes, deterministic=self.deterministic,
                                              params=frozen["params"], assume_clean_log=True)
                self._pending = {"state": state, "frozen": frozen, "side": side, "rows": [], "timesteps": self.num_timesteps}
                return
            rew, length = evaluate_policy(model, self.eval_env, n_eval_episodes=self.n_eval_episodes, deterministic=self.deterministic,
                                          return_episode_rewards=True)
            self._cached = (np.asarray(rew, np.float64), np.asarray(length, np.int64))
            mean = float(np.mean(self._cached[0]))
            if mean > self.best_mean_reward and self.best_model_save_path is not None:
                model.save(os.path.join(self.best_model_save_path, "best_model"))  # (reads the parameters: same stream, same reasoning)

    def _complete_pending(self) -> None:
        """Collect the deferred evaluation: its rows get their results, best_model.zip is written from the snapshot the evaluation ran on."""
        p = self._pending
        if p is None:
            return
        self._pending = None
        import torch

        from .evaluation import evaluate_policy_finish

        with torch.cuda.stream(p["side"]):  # (the stream the chunk was queued on: the pop synchronises only that one)
            rew, length = evaluate_policy_finish(p["state"], return_episode_rewards=True)
            self._cached = (np.asarray(rew, np.float64), np.asarray(length, np.int64))
            for i in p["rows"]:
                self.evaluations_results[i], self.evaluations_length[i] = self._cached
            self.last_mean_reward = float(np.mean(self._cached[0]))
            if self.last_mean_reward > self.best_mean_reward:
                if self.best_model_save_path is not None:
                    self.model.save(os.path.join(self.best_model_save_path, "best_model"), _frozen=p["frozen"])
                self.best_mean_reward = self.last_mean_reward
        if self.verbose >= 1:
            print(f"Eval num_timesteps={p['timesteps']}, episode_reward={self.last_mean_reward:.2f} +/- {float(np.std(self._cached[0])):.2f}")
        self._flush(force=self.n_fresh_evaluations <= 1)

    def _tick(self) -> None:
        key = self._policy_key()
        fresh = not (self.deterministic and key == self._cached_key and None not in key[:2])
        if fresh:
            self._complete_pending()
            self._evaluate_fresh()
            self._cached_key = key
            self.n_fresh_evaluations += 1
        if self._pending is not None:  # the row keeps SB3's cadence and timestep; its results arrive with _complete_pending
            self.evaluations_timesteps.append(self.num_timesteps)
            self.evaluations_results.append(None)
            self.evaluations_length.append(None)
            self._pending["rows"].append(len(self.evaluations_timesteps) - 1)
            self._dirty = True
            return
        rewards, lengths = self._cached
        self.evaluations_timesteps.append(self.num_timesteps)
        self.evaluations_results.append(rewards)
        self.evaluations_length.append(lengths)
        self._dirty = True
        if fresh:
            self.last_mean_reward = float(np.mean(rewards))
            self._flush(force=self.n_fresh_evaluations <= 1)
            if self.verbose >= 1:
                print(f"Eval num_timesteps={self.num_timesteps}, episode_reward={self.last_mean_reward:.2f} +/- {float(np.std(rewards)):.2f}")
            if self.last_mean_reward > self.best_mean_reward:
                self.best_mean_reward = self.last_mean_reward

    def _on_training_end(self) -> None:
        self._flush()

    def flush(self) -> None:
        self._flush()

    def on_update_queued(self) -> None:
        self._complete_pending()
