"""`python -m three_mlagents_amd <verb> ...` -- a small runner for the harness, accepting the verbs and flags of the reference's
`three-mlagents` command (grammar: /root/reference/backend/mlagents/cli.py:14-41) so existing invocations keep working.
The reference's own CLI module is the one a maintainer keeps; this is a convenience for driving the engine stand-alone."""
from __future__ import annotations

import argparse
import dataclasses
import json
import sys

# verb -> [(flags, argparse keywords)]
VERBS = {
    "list": [(("--trainable-only",), {"action": "store_true"})],
    "inspect": [(("task",), {})],
    "train": [(("task",), {}), (("--algorithm", "-a"), {}), (("--timesteps", "-t"), {"type": int}), (("--seed",), {"type": int, "default": 1}),
              (("--n-envs",), {"type": int}), (("--eval-episodes",), {"type": int}), (("--eval-freq",), {"type": int, "default": 10_000}),
              (("--run-name",), {}), (("--quiet",), {"action": "store_true"})],
    "evaluate": [(("task",), {}), (("model",), {}), (("--episodes",), {"type": int}), (("--seed",), {"type": int, "default": 10_001}),
                 (("--stochastic",), {"action": "store_true"})],
}


def parser() -> argparse.ArgumentParser:
    top = argparse.ArgumentParser(prog="three-mlagents")
    verbs = top.add_subparsers(dest="command", required=True)
    for verb, options in VERBS.items():
        p = verbs.add_parser(verb)
        for flags, kw in options:
            p.add_argument(*flags, **kw)
    return top


def run(argv=None) -> dict | list:
    from . import harness, tasks

    a = parser().parse_args(argv)
    if a.command == "list":
        return [t.card() for t in tasks.ENGINE_TASKS.values()]
    if a.command == "inspect":
        env = tasks.make_env(a.task)
        spaces = {"task": a.task, **{k: repr(getattr(env, k)) for k in ("observation_space", "action_space")}}
        env.close()
        return spaces
    if a.command == "train":
        cfg = harness.TrainConfig(a.task, a.timesteps, a.algorithm, a.seed, a.n_envs, a.eval_episodes, a.eval_freq, run_name=a.run_name,
                                  verbose=int(not a.quiet))
        return dataclasses.asdict(harness.train_task(cfg))
    return harness.evaluate_model(a.task, a.model, episodes=a.episodes, deterministic=not a.stochastic, seed=a.seed)


def main(argv=None) -> None:
    print(json.dumps(run(argv), indent=2))


if __name__ == "__main__":
    main(sys.argv[1:])
