"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/tma.h
declares; status codes map to the reference's exception types; no compute is attempted without a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from three_mlagents_amd import _lib

    header = open(os.path.join(ROOT, "include", "tma.h")).read()
    declared = set(re.findall(r"\b(tma_[a-z0-9_]+)\s*\(", header))
    declared -= {"tma_last_error"} - {"tma_last_error"}
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libtma_hip.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert L.tma_version() == int(re.search(r"#define TMA_VERSION (\d+)", header).group(1)) >= 200


def test_task_metadata_matches_reference_spaces():
    from three_mlagents_amd import _lib

    L = _lib.lib()
    # backend/mlagents/envs.py:38-44,166-199 ; max_episode_steps envs.py:35 + examples MAX_STEPS_PER_EP
    expect = {"basic": (21, 3, 50), "gridworld": (4, 5, 100), "ball3d": (6, 5, 200), "push": (4, 5, 120), "crawler": (172, 0, 1000), "walljump": (4, 4, 150),
              "bicycle": (7, 3, 2000), "brickbreak": (45, 3, 2000), "glider": (16, 5, 4000), "ant": (105, 0, 1000)}
    for name, (d, a, m) in expect.items():
        t = _lib.task_id(name)
        assert (L.tma_task_obs_dim(t), L.tma_task_num_actions(t), L.tma_task_max_episode_steps(t)) == (d, a, m)
    # `ant`: the shapes of the reference's Ant-v5 task (envs.py:274-277); `crawler`: BASELINE.json's 172 / 20 shape -- two tasks, one chain
    assert _lib.task_id("ant") != _lib.task_id("crawler") and (L.tma_task_act_dim(_lib.task_id("ant")), L.tma_task_act_dim(_lib.task_id("crawler"))) == (8, 20)


def test_status_codes_map_to_reference_exception_types():
    import ctypes as C

    from three_mlagents_amd import _lib

    with pytest.raises(KeyError):
        _lib.task_id("definitely-not-a-task")
    h = C.c_void_p()
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().tma_env_create(1, 0, 0, 1, 0, 8, C.byref(h)))
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().tma_gae(None, None, None, None, None, 0.99, 0.95, 4, 4, None, None, None))


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from three_mlagents_amd.vec_env import HipEnvEngine

    with pytest.raises(RuntimeError):
        HipEnvEngine("gridworld", 8)


def test_monitor_rows_are_printed_like_python_prints_them(tmp_path):
    """tma_monitor_append_rows (host code, no GPU): SB3's Monitor writes `round(r, 6),l,round(t, 6)` through csv.DictWriter
    (reference training.py:85-86) -- the native writer must produce the same numbers, and the same text in positional notation."""
    import ctypes as C

    import numpy as np

    from three_mlagents_amd import _lib

    rng = np.random.default_rng(0)
    r = np.concatenate([rng.normal(size=500) * 10.0 ** rng.integers(-4, 4, 500), [0.0, 1.0, -1.0, 0.95, -0.03, 150.0, 1e-7, 0.1 + 0.2, 123456.789]])
    length = rng.integers(1, 4000, len(r)).astype(np.int32)
    t = np.cumsum(rng.random(len(r)))
    path = tmp_path / "0.monitor.csv"
    path.write_text("#{}\nr,l,t\n")
    _lib.check(_lib.lib().tma_monitor_append_rows(str(path).encode(), r.ctypes.data_as(C.c_void_p), length.ctypes.data_as(C.c_void_p),
                                                  t.ctypes.data_as(C.c_void_p), len(r)))
    lines = path.read_text().splitlines()
    assert lines[:2] == ["#{}", "r,l,t"] and len(lines) == 2 + len(r)
    for line, rv, lv, tv in zip(lines[2:], r, length, t):
        a, b, c = line.split(",")
        assert float(a) == round(float(rv), 6) and int(b) == int(lv) and float(c) == round(float(tv), 6), line
        if 1e-4 <= abs(round(float(rv), 6)) < 1e15:  # (Python switches to exponent form below 1e-4: "3.5e-05" vs "0.000035", the same number)
            assert a == repr(round(float(rv), 6)), (a, rv)
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().tma_monitor_append_rows(str(tmp_path / "no" / "such" / "dir.csv").encode(), r.ctypes.data_as(C.c_void_p),
                                                      length.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), 1))


def test_comm_entry_points_validate_their_arguments():
    """tma_comm_* (the library's RCCL communicator): argument errors are reported before RCCL or a GPU is touched."""
    import ctypes as C

    from three_mlagents_amd import _lib

    L = _lib.lib()
    assert L.tma_comm_available() in (0, 1)
    h = C.c_void_p()
    ident = (C.c_ubyte * 128)()
    for world, rank in ((0, 0), (2, 2), (2, -1)):
        with pytest.raises(ValueError):
            _lib.check(L.tma_comm_create(ident, world, rank, -1, C.byref(h)))
    with pytest.raises(ValueError):
        _lib.check(L.tma_comm_create(None, 1, 0, -1, C.byref(h)))
    with pytest.raises(ValueError):
        _lib.check(L.tma_comm_unique_id(None))
    with pytest.raises(ValueError):
        _lib.check(L.tma_comm_allreduce(None, None, 4, 0, None))
    with pytest.raises(ValueError):
        _lib.check(L.tma_comm_bind_stream(None, None))
    assert L.tma_comm_allreduce_cb(None, None, 4) == 1 and L.tma_comm_destroy(None) == 0


def test_policy_dims_validation_and_the_split_layout():
    """tma_policy_param_count validates the shape without a GPU.  mfma_dtype 2 (three-term bf16 split of the f32 update, round 5): the f32 mode's
    buffer + three planes of the fragment-major images per net; refused outside Discrete / hidden 256 / <= 32 observations."""
    import ctypes as C

    from three_mlagents_amd import _lib

    L = _lib.lib()

    def count(D, H, A, cont, dtype):
        d = _lib.PolicyDims(D, H, A, cont, dtype, -1)
        nt, n = C.c_int64(0), C.c_int64(0)
        rc = L.tma_policy_param_count(C.byref(d), C.byref(nt), C.byref(n))
        return rc, nt.value, n.value

    rc0, nt0, n0 = count(6, 256, 5, 0, 0)
    rc2, nt2, n2 = count(6, 256, 5, 0, 2)
    assert rc0 == 0 and rc2 == 0 and nt0 == nt2 == (6 * 256 + 256 + 256 * 256 + 256) * 2 + 256 * 5 + 5 + 256 + 1
    plane = lambda n_out: 256 * 32 + 2 * 256 * 256 + 16 * 256 + 256 * 32  # noqa: E731  (bf16 elements per plane: W1 | W2 fwd | W2 bwd | W3 fwd | W3 bwd)
    assert n2 - n0 == 3 * (plane(5) + plane(1)) // 2
    for bad in ((6, 128, 5, 0), (6, 256, 5, 1), (40, 256, 5, 0)):  # hidden 128, Box actions, 40 observations
        rc, _, _ = count(*bad, 2)
        assert rc == _lib.TMA_ERR_INVALID and b"mfma_dtype 2" in L.tma_last_error()
    rc, _, _ = count(6, 256, 5, 0, 3)
    assert rc == _lib.TMA_ERR_INVALID


def test_peer_exchange_entry_points_refuse_bad_use_without_touching_a_gpu():
    """include/tma.h tma_comm_p2p_* (ABI 207): argument and ordering errors come back as status codes before any HIP call -- a communicator
    without an RCCL side, never attached to peers, must refuse to enable its exchange and must refuse an all-reduce (nothing to carry it)."""
    import ctypes as C

    from three_mlagents_amd import _lib

    L = _lib.lib()
    h = C.c_void_p()
    for world, rank in ((0, 0), (9, 0), (2, 2), (2, -1)):  # the exchange serves 1..8 ranks of one node
        with pytest.raises(ValueError):
            _lib.check(L.tma_comm_create_p2p(world, rank, -1, C.byref(h)))
    _lib.check(L.tma_comm_create_p2p(2, 1, -1, C.byref(h)))  # (device -1: the calling thread's current device -- no HIP call yet)
    try:
        en, calls, bad, words = C.c_int(7), C.c_int64(7), C.c_int(7), C.c_int64(7)
        _lib.check(L.tma_comm_p2p_status(h, C.byref(en), C.byref(calls), C.byref(bad), C.byref(words)))
        assert (en.value, calls.value, bad.value, words.value) == (0, 0, 0, 0)
        with pytest.raises(ValueError):
            _lib.check(L.tma_comm_p2p_enable(h, 1))  # attach first
        with pytest.raises(ValueError):
            _lib.check(L.tma_comm_p2p_attach(h, None))  # prepare first
        with pytest.raises(ValueError):
            _lib.check(L.tma_comm_p2p_attach_local(h, None))  # (round 6: the in-process form has the same precondition)
        with pytest.raises(ValueError):
            _lib.check(L.tma_comm_p2p_attach_local(h, (C.c_void_p * 2)(h, h)))  # still not prepared
        with pytest.raises(ValueError):
            _lib.check(L.tma_comm_p2p_prepare(h, 0, (C.c_ubyte * 128)()))  # empty slots
        with pytest.raises(ValueError):
            _lib.check(L.tma_comm_p2p_set_timeout(h, 0.0))
        buf = (C.c_float * 16)()
        with pytest.raises(ValueError):
            _lib.check(L.tma_comm_allreduce(h, buf, 16, 0, None))  # no RCCL side and no exchange: nothing can carry it
        assert "peer exchange" in _lib.last_error()
        assert L.tma_comm_allreduce_cb(h, buf, 16) != 0  # the callback form reports failure instead of raising through C frames
    finally:
        _lib.check(L.tma_comm_destroy(h))


def test_round_6_entry_points_refuse_bad_arguments_without_touching_a_gpu():
    """include/tma.h, ABI 208: tma_ppo_train_epochs_local (all epochs of an update in one call) and the capacity argument of
    tma_env_set_reward64 validate before any HIP work."""
    import ctypes as C

    from three_mlagents_amd import _lib

    L = _lib.lib()
    assert L.tma_version() >= 208
    dims = _lib.PolicyDims(4, 256, 5, 0, 0, -1)
    hp = _lib.PPOHParams(0.2, 0.01, 0.5, 1)
    rv = _lib.Rollout(None, None, None, None, None, 16, 64)
    args = lambda n_epochs, params: (params, C.byref(dims), C.byref(rv), 1, 0, n_epochs, 256, C.byref(hp), None, None, None, 1, 3e-4, 0.9, 0.999, 1e-5, 0.5, None, None)  # noqa: E731
    with pytest.raises(ValueError):
        _lib.check(L.tma_ppo_train_epochs_local(*args(3, None)))  # null buffers
    assert "null argument" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(L.tma_env_set_reward64(None, None, 0))  # null handle
