"""world_size-2 gloo test of the data-parallel plumbing (three-mlagents_amd/dist.py): env sharding and the gradient
all-reduce that PPO.train issues once per minibatch.  Runs on CPU."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from three_mlagents_amd import dist

    rk, lr, ws = dist.init_from_env(backend="gloo")
    assert (rk, ws) == (rank, world) and dist.world_size() == world and dist.rank() == rank
    off, n = dist.shard_envs(4096)
    # identical replicas + different local gradients -> identical averaged update on every rank
    torch.manual_seed(0)
    params = torch.randn(1000)
    grad = torch.full((1000,), float(rank + 1))
    dist.allreduce_sum_(grad)
    params -= 0.1 * grad / ws
    mx = dist.allreduce_max_float(10.0 + rank, device="cpu")
    dist.barrier()
    q.put((rank, off, n, float(grad[0]), float(params.sum()), mx))


@pytest.mark.timeout(120)
def test_two_rank_gloo_sharding_and_gradient_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, off0, n0, g0, s0, m0), (r1, off1, n1, g1, s1, m1) = res
    assert (off0, n0, off1, n1) == (0, 4096, 4096, 4096)  # rank r owns global envs [r*n, (r+1)*n)
    assert g0 == g1 == 3.0 and s0 == s1  # sum over ranks; replicas stay bit-identical
    assert m0 == m1 == 11.0


@pytest.mark.timeout(120)
def test_bench_launcher_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` (N > 1) starts N ranks itself, before any GPU call; with fewer than N GPUs visible (none in this
    container) it must say so and exit non-zero instead of silently measuring one GPU -- and print no JSON line."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.device_count() >= 2:
        pytest.skip("multi-GPU machine")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=100)
    assert r.returncode == 2 and "needs 2 visible GPUs" in r.stderr and r.stdout.strip() == ""
    # a rank count that disagrees with --gpus is an error too, not a warning
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=100, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""
