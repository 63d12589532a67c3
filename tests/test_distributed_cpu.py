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
    assert allreduce_gradients([big] + smalls, world) == 1.0
    ok = all(torch.allclose(p.grad, torch.full_like(p, 1.5 * (i + 1))) for i, p in enumerate([big] + smalls))
    # average=False: the SUM stays in .grad and the caller gets the factor to fold into the optimiser
    for i, p in enumerate([big] + smalls):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    ok = ok and allreduce_gradients([big] + smalls, world, average=False) == 0.5
    ok = ok and all(torch.allclose(p.grad, torch.full_like(p, 3.0 * (i + 1))) for i, p in enumerate([big] + smalls))
    # bf16 payload for the table gradient (INR_GRAD_DTYPE=bf16): same sums to bf16 accuracy, small tensors stay fp32
    from instance_nerf_amd.nerf import utils
    utils.grad_sync.payload = "bf16"
    gen = torch.Generator().manual_seed(rank)
    mine = torch.randn(5_000_000, generator=gen)
    other = torch.randn(5_000_000, generator=torch.Generator().manual_seed(1 - rank))
    big.grad = mine.clone()
    smalls[0].grad = torch.full_like(smalls[0], 1.0 + 2 ** -12)        # not representable in bf16
    smalls[1].grad = torch.ones_like(smalls[1])
    allreduce_gradients([big] + smalls, world, average=False)
    want = mine + other
    ok = ok and float((big.grad - want).abs().max()) < 2 ** -6 * float(want.abs().max())
    ok = ok and float((big.grad - want).abs().max()) > 0                # it really went through bf16
    ok = ok and torch.equal(smalls[0].grad, torch.full_like(smalls[0], 2.0 + 2 ** -11))
    utils.grad_sync.payload = "fp32"
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


def _scatter_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from instance_nerf_amd.nerf import utils
    gs = utils.grad_sync
    gs.world_size, gs.schedule, gs.enabled = world, "reduce_scatter", True
    ok = True
    try:
        for payload in ("fp32", "bf16"):
            gs.payload = payload
            # a "table" of 1001 x 2 (rows the world size does not divide) reduced in two row ranges, as the backward does
            table = torch.nn.Parameter(torch.zeros(1001, 2))
            small = torch.nn.Parameter(torch.zeros(7, 3))
            table._inr_split_row = 400
            full = [torch.randn(1001, 2, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
            table.grad = full[rank].clone()
            small.grad = torch.full_like(small, float(rank + 1))
            want = sum(full)
            scale = utils.allreduce_gradients([table, small], world, average=False, sharded=True)
            ok = ok and scale == 1.0 / world
            pieces = table._inr_grad_shards
            ok = ok and pieces is not None and len(pieces) == 2 and table.grad is None
            # this rank's pieces are the matching rows of the sum; all ranks' pieces tile the table exactly once
            got = torch.full((2002,), float("nan"))
            cover = torch.zeros(2002)
            for pc in pieces:
                a, n = pc["own"], pc["grad"].numel()
                ok = ok and n == pc["count"]
                got[a:a + n] = pc["grad"]
                cover[a:a + n] += 1
            tol = 1e-6 if payload == "fp32" else 2 ** -6 * float(want.abs().max())
            mine = cover > 0
            ok = ok and float((got[mine] - want.reshape(-1)[mine]).abs().max()) <= tol
            dist.all_reduce(cover)
            ok = ok and bool((cover == 1).all())
            ok = ok and torch.equal(small.grad, torch.full_like(small, float(sum(range(1, world + 1)))))
            # "optimiser": every rank writes its rows, then the rows travel to all (allgather_params)
            flat = table.data.view(-1)
            for pc in pieces:
                flat[pc["own"]:pc["own"] + pc["count"]] = pc["grad"]
            gs.allgather_params([table])
            ok = ok and table._inr_grad_shards is None
            ok = ok and float((table.data - want).abs().max()) <= tol
    except Exception as e:                                     # noqa: BLE001
        ok = f"{type(e).__name__}: {e}"
    dist.barrier()
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_scatter_schedule_with_rows_the_world_does_not_divide(world):
    """grad_sync.schedule = "reduce_scatter": every row range of a table gradient is reduce-scattered - padded on the wire
    when the world size does not divide it (round-3 advisor: with 3 or 6 ranks one of the two level ranges of the
    BASELINE table fell back to an all-reduce whose rows the optimiser then never saw).  The pieces of all ranks tile
    the table exactly once, carry the sum, and the updated rows come back whole."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scatter_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert all(ok is True for _, ok in res), res


def test_views_are_sharded_without_overlap():
    """bench.py's view assignment (i + rank) % n: distinct views per rank at every step, and over n steps every rank
    renders every view once (the views' sample counts differ by +-25 %: a rank pinned to one view would make the
    max-over-ranks time of the scaling bench depend on which view it drew)."""
    n = 8
    for world in (1, 2, 4, 8):
        for i in range(6):
            views = [(i + r) % n for r in range(world)]
            assert len(set(views)) == world
        for r in range(world):
            assert sorted((i + r) % n for i in range(n)) == list(range(n))


def test_shard_range_covers_rays_in_whole_groups():
    from instance_nerf_amd.nerf.utils import shard_range
    for n in (640000, 4096, 1000, 17, 5):
        for world in (1, 2, 3, 8):
            b = [shard_range(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert all(lo % 16 == 0 for lo, _ in b)


def test_shard_indices_partition_the_rays_in_whole_groups():
    from instance_nerf_amd.nerf.utils import shard_indices
    for n in (640000, 4096, 5000, 2049):
        for world in (1, 2, 3, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            if n < 1024 * world:
                continue
            allidx = torch.cat(parts)
            assert allidx.numel() == n and torch.equal(torch.sort(allidx).values, torch.arange(n))
            for p in parts:                       # runs of whole 16-ray groups; sizes within one chunk of each other
                assert int(p[0]) % 16 == 0 and abs(p.numel() - n / world) <= 1024
                starts = p[::16]
                assert bool((starts % 16 == 0).all())


class _FakeModel:
    """render() stand-in with per-ray results that depend on the ray only (the sharding logic is host-side)."""
    def render(self, rays_o, rays_d, **kw):
        s = rays_o.sum(-1) + 2 * rays_d.sum(-1)
        return {"image": torch.stack([s, 2 * s, 3 * s], -1), "depth": s + 1, "weights_sum": s * 0 + 0.5,
                "num_samples": torch.zeros(2)}


def _render_worker(rank, world, port, q, n):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from instance_nerf_amd.nerf.utils import render_sharded
    g = torch.Generator().manual_seed(1)
    ro, rd = torch.randn(1, n, 3, generator=g), torch.randn(1, n, 3, generator=g)
    full = _FakeModel().render(ro, rd)
    got = render_sharded(_FakeModel(), ro, rd, rank, world)
    ok = all(torch.equal(got[k], full[k]) for k in ("image", "depth", "weights_sum")) and "num_samples" not in got
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [4096, 5000])        # even shards (all_gather) and uneven ones (broadcasts)
def test_render_sharded_world2(n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_render_worker, args=(r, 2, port, q, n)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res)


def test_bench_diagnostics_never_raise_without_a_gpu(tmp_path):
    """bench.py's opt-in run-to-run diagnostics (--diagnostics, tools/bench_diagnostics.py; `clocks`: shader clock sampled
    through sysfs, the XCD map of the process) are decoration around the timed region: with no GPU, no sysfs node or no probe library they report nothing instead of
    failing the run; the sysfs parser takes the starred line of a pp_dpm_sclk file."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "tools", "bench_diagnostics.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    s = bench.SclkSampler("cpu")
    assert s.path is None
    s.start()
    assert s.stop() is None
    got = bench.xcd_map_probe()
    assert got is None or "error" in got or "workgroups_per_xcd" in got
    f = tmp_path / "pp_dpm_sclk"
    f.write_text("0: 500Mhz\n1: 2151Mhz *\n2: 2400Mhz\n")
    s.path = str(f)
    assert s._read() == 2151
    s.start()
    import time
    time.sleep(0.05)
    out = s.stop()
    assert out["sclk_mhz_median"] == 2151 and out["samples"] >= 1


def _legs_worker(rank, world, port, q):
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from bench_legs import Legs
    line = {} if rank == 0 else None
    legs = Legs(rank, world, torch.device("cpu"), line)

    def collective_ok():
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        return {"sum": float(t), "extra": "dropped by keep"}

    def fails_on_rank_1_before_its_collective():
        if rank == 1:
            raise RuntimeError("out of memory on rank 1")
        return {"value": 7}                       # (rank 0 got through its purely local part)

    calls = []

    def never_runs():
        calls.append(1)
        t = torch.zeros(1)
        dist.all_reduce(t)                        # would hang if only some ranks got here
        return {"value": 1}

    a = legs.run("first", collective_ok, keep=("sum",))
    b = legs.run("second", fails_on_rank_1_before_its_collective)
    c = legs.run("third", never_runs)
    d = legs.run("fourth", never_runs)
    dist.barrier()
    q.put((rank, a, b, c, d, len(calls), legs.healthy, line))
    dist.destroy_process_group()


def test_bench_legs_agree_on_a_one_rank_failure_and_skip_the_rest():
    """tools/bench_legs.py (bench.py's N > 1 legs): a leg that raises on ONE rank is recorded as failed on every rank - rank
    0's record says so even though its own call succeeded - and the collective legs after it are skipped on ALL ranks
    instead of leaving the healthy ranks blocked in a collective the failed one never joins."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_legs_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, a, b, c, d, n_calls, healthy, line in res:
        assert a == {"sum": 3.0, "extra": "dropped by keep"}
        assert "error" in b and ("rank 1" in b["error"] if rank == 1 else b["error"] == "failed on another rank")
        assert c == d == {"error": "skipped: an earlier collective leg failed on some rank"}
        assert n_calls == 0 and healthy is False
    line = res[0][7]
    assert line["first"] == {"sum": 3.0} and line["second"] == {"error": "failed on another rank"} and "skipped" in line["third"]["error"]
    assert res[1][7] is None
