"""Host-side contract of the drop-in surface that needs no GPU: task resolution and its error types, the request/result records,
the PPO default table, policy-file lookup, the runner's flag grammar, declared spaces.  The behaviours asserted are the ones the
reference's own tests pin (/root/reference/backend/tests/test_mlagents.py:24-30,47-49,105-122)."""
import dataclasses

import numpy as np
import pytest

from three_mlagents_amd import harness, tasks


def test_engine_task_table():
    assert set(tasks.ENGINE_TASKS) == {"basic", "gridworld", "ball3d", "push", "ant", "crawler", "walljump", "brickbreak", "bicycle", "glider"}
    for t in tasks.ENGINE_TASKS.values():
        card = t.card()
        assert card["trainable"] is True and card["id"] == t.id and card["policy_prefix"].endswith("_policy")
    assert tasks.resolve("gridworld").ppo_n_steps == 1024 and tasks.resolve("push").ppo_n_steps == 2048  # foundation vs benchmark tier
    assert tasks.resolve("ant").pinned is False and tasks.resolve("crawler").pinned is False and tasks.resolve("gridworld").pinned  # chain dynamics are build-defined
    # library facts ride on the card when the .so is built (it is, in this repo)
    assert tasks.resolve("basic").card()["obs_dim"] == 21 and tasks.resolve("ant").card()["act_dim"] == 8 and tasks.resolve("ant").card()["obs_dim"] == 105
    assert tasks.resolve("crawler").card()["act_dim"] == 20 and tasks.resolve("crawler").card()["obs_dim"] == 172


def test_name_resolution_and_error_types():  # test_mlagents.py:47-49 + registry.py:359-369
    assert tasks.resolve("Crawler").id == "crawler" and tasks.resolve("ANT").kernel == "ant" and tasks.resolve("GRIDWORLD").kernel == "gridworld"
    with pytest.raises(KeyError):
        tasks.resolve("not-a-task")
    assert tasks.resolve("brick-break").kernel == "brickbreak" and tasks.resolve("Bicycle").id == "bicycle"  # registry.py:354 spelling
    for ref_only in ("labyrinth", "self_driving_car", "fish"):
        with pytest.raises(ValueError):
            tasks.resolve(ref_only)
    with pytest.raises(ValueError):
        tasks.make_env("kraken")


def test_predict_requires_model_file(tmp_path, monkeypatch):  # test_mlagents.py:105-108
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        harness.predict_action("basic", np.zeros(21, dtype=np.float32), "missing.zip")
    with pytest.raises(FileNotFoundError):
        harness.find_policy(tasks.resolve("basic"))  # no zip at all
    assert not (tmp_path / "policies").exists()  # lookups never create directories


def test_policy_lookup_accepts_path_or_name(tmp_path, monkeypatch):  # test_mlagents.py:110-122
    monkeypatch.chdir(tmp_path)
    d = tmp_path / "policies"
    d.mkdir()
    (d / "basic_policy_a.zip").write_bytes(b"x")
    (d / "basic_policy_b.zip").write_bytes(b"x")
    t = tasks.resolve("basic")
    assert harness.find_policy(t, str(d / "basic_policy_a.zip")) == d / "basic_policy_a.zip"
    assert harness.find_policy(t, "basic_policy_a.zip").name == "basic_policy_a.zip"
    assert harness.find_policy(t).name == "basic_policy_b.zip"  # newest by name (run ids start with a timestamp)


def test_ppo_default_table():  # value table training.py:361-391
    kw = harness.ppo_defaults(tasks.resolve("gridworld"))
    assert (kw["n_steps"], kw["batch_size"], kw["n_epochs"], kw["learning_rate"]) == (1024, 256, 10, 3e-4)
    assert (kw["gamma"], kw["gae_lambda"], kw["clip_range"], kw["ent_coef"], kw["vf_coef"], kw["max_grad_norm"]) == (0.99, 0.95, 0.2, 0.01, 0.5, 0.5)
    assert kw["policy_kwargs"] == {"net_arch": {"pi": [256, 256], "vf": [256, 256]}}
    assert harness.ppo_defaults(tasks.resolve("push"))["n_steps"] == 2048 and "ppo" in harness.ALGORITHMS
    # with the env count known the table scales the literal 256 by n_envs / 8 (the reference's env count): the reference's minibatches per
    # epoch for every task, never below its literal 256
    g = tasks.resolve("gridworld")
    assert [harness.ppo_defaults(g, n)["batch_size"] for n in (1, 8, 15, 16, 4096)] == [256, 256, 256, 512, 131072]
    assert harness.ppo_defaults(tasks.resolve("push"), 2048)["batch_size"] == 2048 * 2048 // 64  # n_steps 2048: 64 minibatches, as 8 x 2048 / 256
    for name in ("brickbreak", "bicycle", "glider", "ant", "crawler", "ball3d", "push"):  # benchmark-tier tasks at their default 8 envs
        t = tasks.resolve(name)
        assert harness.ppo_defaults(t, t.n_envs)["batch_size"] == 256 or t.n_envs >= 16, name
        assert harness.ppo_defaults(t, 8)["batch_size"] == 256, name


def test_records_and_train_errors():  # training.py:40-68,105-114
    cfg = harness.TrainConfig("basic")
    assert (cfg.seed, cfg.eval_freq, cfg.deterministic_eval, cfg.save_policy, cfg.verbose, cfg.algorithm) == (1, 10_000, True, True, 1, None)
    assert [f.name for f in dataclasses.fields(harness.TrainResult)][:4] == ["task_id", "algorithm", "run_id", "model_filename"]
    with pytest.raises(dataclasses.FrozenInstanceError):
        cfg.seed = 2
    with pytest.raises(ValueError):
        harness.train_task(harness.TrainConfig("fish"))
    with pytest.raises(ValueError):
        harness.train_task(harness.TrainConfig("basic", algorithm="sarsa"))
    with pytest.raises(ValueError):
        harness.train_task(harness.TrainConfig("basic", algorithm="dqn"))  # known to SB3, not on the engine
    with pytest.raises(KeyError):
        harness.train_task(harness.TrainConfig("nope"))
    # the reference's catalogue default for basic / gridworld / push / walljump is DQN (registry.py:59,91,107,123): the engine's PPO stands in,
    # and the harness reports which algorithm it replaced (written to metadata.json as `substituted_for`)
    assert harness._algorithm_for(None, tasks.resolve("gridworld")) == ("ppo", "dqn")
    assert harness._algorithm_for(None, tasks.resolve("ball3d")) == ("ppo", None)
    assert harness._algorithm_for("PPO", tasks.resolve("push")) == ("ppo", None)
    assert {t.id for t in tasks.ENGINE_TASKS.values() if t.default_algorithm == "dqn"} == {"basic", "gridworld", "push", "walljump"}


def test_runner_grammar():  # cli.py:14-41
    from three_mlagents_amd.__main__ import parser

    a = parser().parse_args(["train", "basic", "--algorithm", "ppo", "--n-envs", "8", "-t", "1000", "--quiet"])
    assert (a.command, a.task, a.algorithm, a.n_envs, a.timesteps, a.seed, a.eval_freq, a.quiet) == ("train", "basic", "ppo", 8, 1000, 1, 10_000, True)
    a = parser().parse_args(["evaluate", "gridworld", "m.zip", "--stochastic"])
    assert (a.seed, a.stochastic, a.episodes) == (10_001, True, None)


def test_spaces_match_reference_declarations():  # envs.py:38-44,166-199
    from three_mlagents_amd.spaces import task_spaces

    for name, (d, n) in {"basic": (21, 3), "gridworld": (4, 5), "ball3d": (6, 5), "push": (4, 5), "walljump": (4, 4), "bicycle": (7, 3), "brickbreak": (45, 3),
                         "glider": (16, 5)}.items():
        obs_space, act_space = task_spaces(name)
        assert obs_space.shape == (d,) and obs_space.dtype == np.float32 and act_space.n == n
        assert act_space.contains(act_space.sample()) and not act_space.contains(n)
    obs_space, _ = task_spaces("gridworld")
    assert obs_space.contains(np.array([0.25, -0.75, 1, 0], np.float32)) and not obs_space.contains(np.array([2, 0, 0, 0], np.float32))


def test_tensorboard_event_writer_round_trip_and_protobuf_schema(tmp_path):
    """tb_events.py: CRC-32C known answer, writer -> reader round trip, and every record parses under the public Event / Summary schema
    built with the protobuf runtime (an independent decoder of the bytes)."""
    import struct

    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

    from three_mlagents_amd import tb_events as tb

    assert tb.crc32c(b"123456789") == 0xE3069283
    w = tb.EventWriter(str(tmp_path))
    w.add_scalars({"train/approx_kl": 0.0125, "time/fps": 1.5e6, "skipped": float("nan")}, step=4096, wall_time=12.5)
    w.add_scalars({"rollout/ep_rew_mean": -0.75}, step=8192)
    got = tb.read_scalars(w.path)
    assert got[0] == (4096, {"train/approx_kl": struct.unpack("<f", struct.pack("<f", 0.0125))[0], "time/fps": 1.5e6}) and got[1] == (8192, {"rollout/ep_rew_mean": -0.75})
    fd = descriptor_pb2.FileDescriptorProto(name="ev.proto", package="t", syntax="proto3")
    val = fd.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=9, label=1)
    val.field.add(name="simple_value", number=2, type=2, label=1)
    summ = fd.message_type.add(name="Summary")
    summ.field.add(name="value", number=1, type=11, label=3, type_name=".t.Value")
    ev = fd.message_type.add(name="Event")
    ev.field.add(name="wall_time", number=1, type=1, label=1)
    ev.field.add(name="step", number=2, type=3, label=1)
    ev.field.add(name="file_version", number=3, type=9, label=1)
    ev.field.add(name="summary", number=5, type=11, label=1, type_name=".t.Summary")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Event = message_factory.GetMessageClass(pool.FindMessageTypeByName("t.Event"))
    data, pos, events = open(w.path, "rb").read(), 0, []
    while pos < len(data):
        (n,) = struct.unpack("<Q", data[pos:pos + 8])
        events.append(Event.FromString(data[pos + 12:pos + 12 + n]))
        pos += 16 + n
    assert events[0].file_version == "brain.Event:2" and events[1].step == 4096 and events[1].wall_time == 12.5
    assert {v.tag: v.simple_value for v in events[1].summary.value}["time/fps"] == 1.5e6 and events[2].summary.value[0].tag == "rollout/ep_rew_mean"


def test_bridge_message_shapes_without_a_gpu():
    """bridge.py: the `progress` / `trained` frames of the reference's WebSocket protocol (websocket_training.py:19-51,98-112) from the
    engine's callback protocol, with a stand-in trainer (no GPU) and a stand-in socket."""
    import asyncio

    from three_mlagents_amd import bridge, harness

    sent = []

    class Sock:
        async def send_json(self, payload):
            sent.append(payload)

    class Model:
        num_timesteps = 0

        def get_env(self):
            return None

    def fake_train(cfg, *, callback):
        m = Model()
        callback.init_callback(m)
        callback.on_training_start()
        for _ in range(5):
            m.num_timesteps += 1500
            assert callback.on_step()
        return harness.TrainResult(cfg.task_id, "ppo", "gridworld_ppo_20260101_000000_ab12cd34", "gridworld_policy_x.zip", "policies/gridworld_policy_x.zip",
                                   "runs/gridworld/x", 0.5, 0.1, 7, 7500, "runs/gridworld/x/metadata.json")

    out = asyncio.run(bridge.train_for_websocket(Sock(), "gridworld", total_timesteps=7500, progress_freq=2000, train=fake_train))
    kinds = [p["type"] for p in sent]
    assert kinds[0] == "progress" and sent[0]["timesteps"] == 0 and sent[0]["task_id"] == "gridworld" and kinds[-1] == "trained"
    prog = [p for p in sent[1:] if p["type"] == "progress"]
    assert [p["timesteps"] for p in prog] == [3000, 6000] and prog[-1]["progress"] == 0.8 and prog[0]["algorithm"] == "Model"
    assert set(prog[0]) == {"type", "episode", "reward", "loss", "timesteps", "progress", "algorithm"}
    t = sent[-1]
    assert t["file_url"] == "/policies/gridworld_policy_x.zip" and t["session_uuid"] == "ab12cd34" and t["eval_episodes"] == 7 and out["mean_reward"] == 0.5
    assert set(bridge.STATE_FIELDS) == {"basic", "gridworld", "push", "ball3d", "walljump", "bicycle", "brickbreak", "glider"}


def test_bridge_connected_predicate_is_exact():
    """run_for_websocket stops on `application_state != CONNECTED` as the reference does (websocket_training.py:159): the state named
    DISCONNECTED -- which also ENDS in "CONNECTED" -- is a closed socket."""
    import enum

    from three_mlagents_amd import bridge

    class WebSocketState(enum.Enum):  # starlette's shape
        CONNECTING = 0
        CONNECTED = 1
        DISCONNECTED = 2
        RESPONSE = 3

    class Sock:
        application_state = WebSocketState.CONNECTED

    s = Sock()
    assert bridge._connected(s)
    for st in (WebSocketState.DISCONNECTED, WebSocketState.CONNECTING, WebSocketState.RESPONSE):
        s.application_state = st
        assert not bridge._connected(s), st
    s.application_state = "WebSocketState.CONNECTED"
    assert bridge._connected(s)
    s.application_state = "WebSocketState.DISCONNECTED"
    assert not bridge._connected(s)

    class Plain:  # no state attribute at all: only max_steps bounds the loop
        pass

    assert bridge._connected(Plain())

    class Closable:
        def __init__(self):
            self.flag = False

        def closed(self):
            return self.flag

    c = Closable()
    assert bridge._connected(c)
    c.flag = True
    assert not bridge._connected(c)


def _sb3_like_policy(D, H, A, continuous):
    """A torch module with the parameter names and registration order of SB3's ActorCriticPolicy (MlpExtractor with separate 2-layer
    pi / vf nets; `log_std` a direct parameter of the policy)."""
    import torch
    from torch import nn

    class Extractor(nn.Module):
        def __init__(self):
            super().__init__()
            self.policy_net = nn.Sequential(nn.Linear(D, H), nn.Tanh(), nn.Linear(H, H), nn.Tanh())
            self.value_net = nn.Sequential(nn.Linear(D, H), nn.Tanh(), nn.Linear(H, H), nn.Tanh())

    class Policy(nn.Module):
        def __init__(self):
            super().__init__()
            self.mlp_extractor = Extractor()
            self.action_net = nn.Linear(H, A)
            if continuous:
                self.log_std = nn.Parameter(torch.zeros(A))
            self.value_net = nn.Linear(H, 1)

    return Policy()


def test_sb3_zip_members_without_sb3(monkeypatch):
    """sb3_format.py: the three pickled members of `data` restore through a re-statement of SB3's json_to_data (base64 -> cloudpickle.loads)
    into classes that restore state the way gymnasium's spaces do (`__dict__.update`), the parameter order is torch's registration order of
    an ActorCriticPolicy, and the Adam state_dict loads into a real torch.optim.Adam over such a module and steps."""
    import base64
    import sys
    import types

    import cloudpickle
    import numpy as np
    import torch

    from three_mlagents_amd import sb3_format as sf
    from three_mlagents_amd.spaces import Box, Discrete

    assert "gymnasium" not in sys.modules and "stable_baselines3" not in sys.modules  # (the image has neither; stand-ins below)
    members = sf.data_members(Box(-1.0, 1.0, (4,), np.float32), Discrete(5))
    assert "gymnasium" not in sys.modules  # the writer leaves no stub module behind

    class Space:  # gymnasium.spaces.Space.__setstate__: legacy key renames, then __dict__.update
        def __setstate__(self, state):
            state = dict(state)
            for old, new in (("shape", "_shape"), ("np_random", "_np_random")):
                if old in state:
                    state[new] = state.pop(old)
            self.__dict__.update(state)

        shape = property(lambda self: self._shape)

    for name, attrs in (("gymnasium", {}), ("gymnasium.spaces", {}), ("gymnasium.spaces.box", {"Box": type("Box", (Space,), {})}),
                        ("gymnasium.spaces.discrete", {"Discrete": type("Discrete", (Space,), {})}), ("stable_baselines3", {}),
                        ("stable_baselines3.common", {}), ("stable_baselines3.common.policies", {"ActorCriticPolicy": type("ActorCriticPolicy", (), {})})):
        mod = types.ModuleType(name)
        for k, v in attrs.items():
            v.__module__, v.__qualname__ = name, k
            setattr(mod, k, v)
        monkeypatch.setitem(sys.modules, name, mod)
    got = {k: cloudpickle.loads(base64.b64decode(v[":serialized:"].encode())) for k, v in members.items()}  # SB3's json_to_data, per member
    assert got["policy_class"] is sys.modules["stable_baselines3.common.policies"].ActorCriticPolicy
    ob, ac = got["observation_space"], got["action_space"]
    assert type(ob).__name__ == "Box" and ob.shape == (4,) and ob.dtype == np.float32 and np.array_equal(ob.low, -np.ones(4, np.float32))
    assert np.array_equal(ob.high, np.ones(4, np.float32)) and ob.bounded_below.all() and ob.low_repr == "-1.0" and ob._np_random is None
    assert type(ac).__name__ == "Discrete" and int(ac.n) == 5 and int(ac.start) == 0 and ac.shape == () and ac.dtype == np.int64

    for cont in (False, True):
        D, H, A = 4, 64, 5
        net = _sb3_like_policy(D, H, A, cont)
        order = sf.parameter_order(cont)
        assert [n for n, _ in net.named_parameters()] == order
        moments = {n: torch.rand_like(p) for n, p in net.named_parameters()}
        sd = sf.adam_state_dict(order, moments, {k: v * v for k, v in moments.items()}, step=7, lr=3e-4)
        opt = torch.optim.Adam(net.parameters(), lr=1.0, eps=1e-5)
        opt.load_state_dict(sd)
        assert opt.param_groups[0]["lr"] == 3e-4 and float(opt.state[next(iter(net.parameters()))]["step"]) == 7.0
        for p in net.parameters():
            p.grad = torch.ones_like(p)
        opt.step()
        assert float(opt.state[next(iter(net.parameters()))]["step"]) == 8.0
        fresh = sf.adam_state_dict(order, {}, {}, step=0, lr=3e-4)  # an untrained model: empty state, still loadable
        torch.optim.Adam(net.parameters()).load_state_dict(fresh)


def test_eval_callback_runs_episodes_once_per_optimizer_state(tmp_path, monkeypatch):
    """EvalCallback (training.py:152-161 configures SB3's): the reference's row cadence -- one row of `evaluations.npz` per `eval_freq` calls --
    with the episodes re-run only after the optimizer stepped (a deterministic evaluation from fixed reset seeds is a function of the
    parameters alone); a stochastic evaluation is never repeated from the cache.  No GPU: the evaluation itself is a stand-in."""
    from three_mlagents_amd import callbacks, evaluation

    calls = []

    def fake_evaluate(model, env, n_eval_episodes, deterministic, return_episode_rewards):
        calls.append(model._n_updates)
        return [float(model._n_updates)] * n_eval_episodes, [7] * n_eval_episodes

    monkeypatch.setattr(evaluation, "evaluate_policy", fake_evaluate)

    class Model:
        num_timesteps, _n_updates, _adam_step, policy = 0, 0, 0, object()

        def get_env(self):
            return None

        def save(self, path):
            saved.append(path)

    saved = []
    for deterministic, want_fresh in ((True, 3), (False, 12)):
        calls.clear()
        m = Model()
        cb = callbacks.EvalCallback(object(), log_path=str(tmp_path / f"e{int(deterministic)}"), best_model_save_path=str(tmp_path / "best"), eval_freq=2,
                                    n_eval_episodes=4, deterministic=deterministic)
        cb.flush_interval_s = 0.0
        cb.init_callback(m)
        cb.on_training_start()
        for rollout in range(3):
            for _ in range(8):  # eight vector steps of one rollout: the policy cannot move in between
                m.num_timesteps += 4096
                assert cb.on_step()
            m._n_updates += 10  # train(): ten epochs
            m._adam_step += 320
        cb.on_training_end()
        assert len(calls) == want_fresh == cb.n_fresh_evaluations
        ev = np.load(tmp_path / f"e{int(deterministic)}" / "evaluations.npz")
        assert ev["timesteps"].tolist() == [4096 * 2 * k for k in range(1, 13)] and ev["results"].shape == (12, 4) and ev["ep_lengths"].shape == (12, 4)
        assert ev["results"][:, 0].tolist() == [0.0] * 4 + [10.0] * 4 + [20.0] * 4  # rows of a rollout repeat that rollout's evaluation
    assert len(saved) == 6  # a new best after every optimizer state, in both runs


@pytest.mark.parametrize("task", ["basic", "gridworld", "push", "walljump"])
def test_reward_table_is_the_reference_float64_reward_set(task):
    """envs.reward_table maps every float32 reward of the reference-generated fixtures to exactly the float64 the reference returned
    (`rewards_f64`: Basic 0.09000000000000001, Push -0.060000000000000005, ...), with no two float64 values behind one float32."""
    import os

    from three_mlagents_amd.envs import reward_table

    tab = reward_table(task)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{task}.npz"))
    for pre in ("", "b_"):
        r32, r64 = g[pre + "rewards_f32"].reshape(-1), g[pre + "rewards_f64"].reshape(-1)
        assert np.array_equal(np.array([tab[float(x)] for x in r32]), r64)
    assert all(float(np.float32(v)) == k for k, v in tab.items())
    assert reward_table("ball3d") == {}


def test_eval_callback_bulk_steps_equal_the_per_step_loop():
    """callbacks.BaseCallback.on_steps (round 4): a native rollout chunk reports its vector steps in bulk.  EvalCallback.bulk_steps must leave
    exactly what SB3's per-step protocol leaves -- the same rows at the same timesteps, a fresh evaluation exactly where the optimizer has stepped,
    the same counters -- and a CallbackList with a callback that does not take bulk steps must fall back to the per-step loop (num_timesteps
    advancing by n_envs per call, a False return stopping the rollout at that step)."""
    from three_mlagents_amd.callbacks import BaseCallback, CallbackList, EvalCallback

    class Model:
        def __init__(self):
            self.num_timesteps, self._n_updates, self._adam_step, self.policy = 0, 0, 0, object()

        def get_env(self):
            return None

    def run(bulk, chunks=((7, 3), (12, 3), (5, 3)), eval_freq=4):
        m = Model()
        ev = EvalCallback(eval_env=None, eval_freq=eval_freq, n_eval_episodes=2, deterministic=True)
        fresh_at = []

        def fake_eval():
            fresh_at.append((m.num_timesteps, m._n_updates))
            ev._cached = (np.array([float(m._n_updates), 1.0]), np.array([3, 4]))

        ev._evaluate_fresh = fake_eval
        cb = CallbackList([ev])
        cb.init_callback(m)
        for n, per_step in chunks:
            if bulk:
                done, go = cb.on_steps(n, per_step)
            else:
                done, go = BaseCallback.on_steps(cb, n, per_step)
            assert (done, go) == (n, True)
            m._n_updates += 1  # the optimizer steps between rollouts
            m._adam_step += 10
        return m.num_timesteps, ev.n_calls, ev.num_timesteps, list(ev.evaluations_timesteps), [r.tolist() for r in ev.evaluations_results], fresh_at, ev.n_fresh_evaluations

    assert run(True) == run(False)
    ts, calls, _, rows, results, fresh, n_fresh = run(True)
    assert ts == 24 * 3 and calls == 24 and rows == [12, 24, 36, 48, 60, 72] and n_fresh == 3  # one fresh evaluation per optimizer state
    assert [r[0] for r in results] == [0.0, 1.0, 1.0, 1.0, 2.0, 2.0]

    class Stopper(BaseCallback):  # an SB3-style user callback: no bulk_steps -> the whole list is driven per step
        def _on_step(self):
            return self.n_calls < 5

    m = Model()
    ev = EvalCallback(eval_env=None, eval_freq=2, n_eval_episodes=1)
    ev._evaluate_fresh = lambda: setattr(ev, "_cached", (np.array([0.0]), np.array([1])))
    cb = CallbackList([ev, Stopper()])
    cb.init_callback(m)
    assert cb.on_steps(10, 8) == (5, False) and m.num_timesteps == 40 and ev.n_calls == 5 and ev.evaluations_timesteps == [16, 32]
