"""World-size-2 gloo tests of the N>1 logic (CPU): gradient all-reduce and the bench's reductions."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from instance_nerf_amd.nerf.utils import allreduce_gradients
    torch.manual_seed(0)
    big = torch.nn.Parameter(torch.zeros(5_000_000))          # goes as its own message (>= 16 MB)
    smalls = [torch.nn.Parameter(torch.zeros(64, 32)), torch.nn.Parameter(torch.zeros(3, 64))]
    for i, p in enumerate([big] + smalls):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    allreduce_gradients([big] + smalls, world)
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate([big] + smalls))
    # bench-style reductions: max of times, sum of samples
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    n = torch.tensor([100.0 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    dist.barrier()
    q.put((rank, ok, float(t), float(n)))
    dist.destroy_process_group()


def test_allreduce_gradients_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res)
    assert all(r[2] == 2.0 and r[3] == 300.0 for r in res)


def test_views_are_sharded_without_overlap():
    """bench.py's view assignment (i*world+rank) % n covers distinct views per rank per step."""
    world, n = 4, 8
    for i in range(6):
        views = [(i * world + r) % n for r in range(world)]
        assert len(set(views)) == world
