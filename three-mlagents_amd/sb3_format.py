"""The pieces of a stable-baselines3 model zip that only SB3 can normally write (the reference saves with
`model.save(...)` at backend/mlagents/training.py:172-175 and reads back with `PPO.load(...)`): the `data` JSON whose non-JSON members
are base64 cloudpickle blobs, and the torch.optim.Adam state_dict of `policy.optimizer.pth`.

stable-baselines3 and gymnasium are not installed here, so the three blobs `BaseAlgorithm.load` cannot do without are produced from
their public pickle forms instead of from live objects:
  * `policy_class`     -- a class pickles BY REFERENCE: the GLOBAL opcode naming `stable_baselines3.common.policies.ActorCriticPolicy`.
  * `observation_space`, `action_space` -- gymnasium spaces pickle as `copyreg.__newobj__(cls)` + the instance `__dict__`, and
    `Space.__setstate__` restores them with `self.__dict__.update(state)`; the blob is written by pickling a stand-in object whose class
    carries gymnasium's module / qualified name (`gymnasium.spaces.box.Box`, `gymnasium.spaces.discrete.Discrete`) and whose `__dict__`
    holds the attributes gymnasium's own instances have.
Everything else SB3's `_setup_model` reads after `model.__dict__.update(data)` is a JSON scalar.  VERIFIED HERE ONLY against torch itself
(optimizer / policy state_dicts load into a torch module laid out like SB3's ActorCriticPolicy) and against a re-statement of SB3's
`json_to_data` + gymnasium's `__setstate__` (tests/test_harness_cpu.py); never against SB3, which this image does not have.
"""
from __future__ import annotations

import base64
import contextlib
import pickle
import sys
import types
from typing import Any

import numpy as np

POLICY_CLASS = ("stable_baselines3.common.policies", "ActorCriticPolicy")


def class_reference_pickle(module: str, qualname: str) -> bytes:
    """pickle.dumps(cls) for an importable class: PROTO 2, GLOBAL module / name, STOP."""
    return b"\x80\x02c" + module.encode() + b"\n" + qualname.encode() + b"\n."


@contextlib.contextmanager
def _stand_in(module: str, qualname: str):
    """A class that pickles as `module.qualname`, registered under stub modules for the duration of one dumps() call."""
    made, cls = [], type(qualname, (), {"__module__": module, "__qualname__": qualname})
    parts = module.split(".")
    for depth in range(1, len(parts) + 1):
        name = ".".join(parts[:depth])
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            made.append(name)
    had = getattr(sys.modules[module], qualname, None)
    setattr(sys.modules[module], qualname, cls)
    try:
        yield cls
    finally:
        if had is None:
            delattr(sys.modules[module], qualname)
        else:
            setattr(sys.modules[module], qualname, had)
        for name in made:
            sys.modules.pop(name, None)


def space_state(space) -> tuple[str, str, dict[str, Any]]:
    """(module, class name, instance __dict__) of the gymnasium space equal to `space` (a spaces.Box / spaces.Discrete of this package
    or gymnasium's own)."""
    if hasattr(space, "n"):
        return "gymnasium.spaces.discrete", "Discrete", {"n": np.int64(space.n), "start": np.int64(getattr(space, "start", 0)), "_shape": (),
                                                         "dtype": np.dtype(np.int64), "_np_random": None}
    low, high = np.asarray(space.low, np.float32), np.asarray(space.high, np.float32)

    def short(a):
        return str(a.flat[0]) if a.size and np.all(a == a.flat[0]) else str(a)

    return "gymnasium.spaces.box", "Box", {"dtype": np.dtype(np.float32), "_shape": tuple(int(x) for x in space.shape), "low": low, "high": high,
                                           "low_repr": short(low), "high_repr": short(high), "bounded_below": np.isfinite(low),
                                           "bounded_above": np.isfinite(high), "_np_random": None}


def space_pickle(space) -> bytes:
    module, name, state = space_state(space)
    with _stand_in(module, name) as cls:
        obj = cls()
        obj.__dict__.update(state)
        return pickle.dumps(obj, protocol=2)


def serialized(blob: bytes, type_name: str, **readable) -> dict[str, Any]:
    """One non-JSON member of `data`, as SB3's data_to_json writes it."""
    return {":type:": type_name, ":serialized:": base64.b64encode(blob).decode(), **readable}


def data_members(observation_space, action_space) -> dict[str, Any]:
    """The three members of `data` BaseAlgorithm.load needs as live objects."""
    om, on, _ = space_state(observation_space)
    am, an, _ = space_state(action_space)
    return {
        "policy_class": serialized(class_reference_pickle(*POLICY_CLASS), "<class 'abc.ABCMeta'>", __module__=POLICY_CLASS[0]),
        "observation_space": serialized(space_pickle(observation_space), f"<class '{om}.{on}'>", _shape=list(observation_space.shape)),
        "action_space": serialized(space_pickle(action_space), f"<class '{am}.{an}'>", _shape=list(action_space.shape)),
    }


# torch registers an ActorCriticPolicy's parameters in this order: the module's own (log_std) first, then the sub-modules in the order
# _build creates them -- the order torch.optim.Adam's state_dict indexes them by
def parameter_order(continuous: bool) -> list[str]:
    names = ["log_std"] if continuous else []
    for net in ("policy_net", "value_net"):
        for layer in (0, 2):
            names += [f"mlp_extractor.{net}.{layer}.weight", f"mlp_extractor.{net}.{layer}.bias"]
    return names + ["action_net.weight", "action_net.bias", "value_net.weight", "value_net.bias"]


def adam_state_dict(order, exp_avg: dict, exp_avg_sq: dict, step: int, lr: float, eps: float = 1e-5) -> dict[str, Any]:
    """torch.optim.Adam.state_dict() for parameters `order` with the given moments (per-parameter tensors in SB3's [out][in] layout)."""
    import torch

    state = {}
    if step > 0:
        for i, name in enumerate(order):
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": exp_avg[name].clone(), "exp_avg_sq": exp_avg_sq[name].clone()}
    group = {"lr": lr, "betas": (0.9, 0.999), "eps": eps, "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
             "differentiable": False, "fused": None, "params": list(range(len(order)))}
    return {"state": state, "param_groups": [group]}
