"""Two ranks on ONE GPU (gloo backend carrying the CUDA gradient tensor): exercises the complete data-parallel PPO path --
env sharding by env_offset, per-minibatch gradient all-reduce, 1/world scaling inside the Adam kernel -- and checks the
replicas stay bit-identical.  (RCCL needs one device per rank; on the 8-GPU node the same code runs with backend nccl.)"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from three_mlagents_amd import dist

    dist.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    from three_mlagents_amd.ppo import PPO
    from three_mlagents_amd.training import make_vector_env

    N = 256
    env = make_vector_env("gridworld", n_envs=N, seed=1, env_offset=rank * N)
    model = PPO("MlpPolicy", env, n_steps=64, batch_size=16384 // 2, n_epochs=2, seed=1, policy_kwargs={"net_arch": [64, 64]})
    assert model.world_size == world and model.rank == rank
    first_obs = env.engine.reset().cpu()
    model._last_obs_valid = False
    model.learn(2 * world * N * 64)
    q.put((rank, model.policy.params[: model.policy.n_trainable].cpu().numpy(), first_obs.numpy(), model.num_timesteps))  # numpy: pickled by value
    env.close()
    dist.barrier()
    import torch.distributed as td

    td.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_one_gpu_replicas_stay_identical():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, p0, o0, t0), (r1, p1, o1, t1) = res
    p0, p1, o0, o1 = (torch.from_numpy(x) for x in (p0, p1, o0, o1))
    assert torch.equal(p0, p1)  # same all-reduced gradient, same Adam state -> bit-identical replicas
    assert not torch.equal(o0, o1)  # the shards really are different envs (global indices 0..255 vs 256..511)
    assert t0 == t1 == 2 * world * 256 * 64  # num_timesteps counts the whole job
    # and the shards are the two halves of one 512-env engine
    from three_mlagents_amd.vec_env import HipEnvEngine

    big = HipEnvEngine("gridworld", 512, seed=1)
    ob = big.reset().cpu()
    assert torch.equal(ob[:256], o0) and torch.equal(ob[256:], o1)
    # a single-rank run on the same data takes different steps (it sees half of each global batch)
    from three_mlagents_amd.ppo import PPO
    from three_mlagents_amd.training import make_vector_env

    env = make_vector_env("gridworld", n_envs=256, seed=1)
    solo = PPO("MlpPolicy", env, n_steps=64, batch_size=16384 // 2, n_epochs=2, seed=1, policy_kwargs={"net_arch": [64, 64]})
    solo.learn(2 * 256 * 64)
    assert not torch.equal(solo.policy.params[: solo.policy.n_trainable].cpu(), p0)
