"""Secondary measurements of bench.py (everything in its JSON line beyond the contract's headline, `roofline` and
`cpu_baseline` objects): the training-step probes (eager and captured two-stream pipeline), the multi-GPU collective
record, BASELINE configs[4] (extraction + RoIAlign-3D), rendering with the instance head, the -O variants and the trained
scene.  Split out of bench.py in round 4 (round-3 verdict: the measuring script had grown to 58 KB); bench.py imports it
unless --no-train-probe is given.  Nothing here is imported by the product."""
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_SAMPLE = 1024         # 16 levels x 8 corners x 2 features x 4 B


SCATTER_JSON = os.path.join("profiles", "r06_scatter_requests.json")


def scatter_requests(stage):
    """Memory-side atomic requests per sample of the table-gradient scatter and the unit's measured rate, from the
    committed PMC profile (profiles/r06_scatter_requests.json, tools/pmc_train.sh + tools/scatter_requests_json.py);
    quoted only for the kernel sources it was measured on.  -> (requests per sample, unit rate in requests/s) or None."""
    from instance_nerf_amd import build
    path = os.path.join(ROOT, SCATTER_JSON)
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    if t.get("source_sha") != build.source_sha("scatter"):
        return None
    from instance_nerf_amd.nerf import network as _network
    form = {0: "fp32 atomics", 32: "int32 (fixed point)", 64: "int64 sums"}[_network.fx_bits()]
    if t.get("scatter_form", "fp32 atomics") != form:
        return None                           # measured on the other form of the scatter
    return float(t["instance_stage" if stage == "instance" else "nerf_stage"]["requests_per_sample"]), float(t["unit_rate_requests_per_s"])


def build_network(dev, seed=0):
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.scene import RoomScene
    torch.manual_seed(seed)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10).to(dev)   # upstream init U(-1e-4,1e-4)
    room = RoomScene()
    net.density_bitfield.copy_(torch.from_numpy(room.density_bitfield(128, 1.0)).to(dev))
    return net.eval(), room


def collective_record(dev, rank, world, backend, red_dev, table_bytes=6119864 * 2 * 4):
    """world > 1, EVERY rank calls it: what a reader needs to trust that the collectives of this run crossed N distinct
    GPUs (round-3 verdict: RCCL has never executed under this repository; the first record must verify itself) - the
    backend and its version, every rank's device (PCI address + name, all-gathered: N distinct ones, or the run is a
    dry run and says so), and a 10-iteration all-reduce of one table gradient (48.96 MB fp32) with its bus bandwidth
    2 (N-1)/N x bytes / time, the figure DESIGN.md section 4 budgets 0.25 ms per step for."""
    import torch.distributed as dist
    pr = torch.cuda.get_device_properties(dev)
    ident = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x} {pr.name}"
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:                                 # noqa: BLE001
            version = f"unavailable ({type(e).__name__})"
    n_it = 10 if backend == "nccl" else 2                      # a gloo dry run moves the 49 MB through host memory
    buf = torch.ones(table_bytes // 4, dtype=torch.float32, device=red_dev)
    for _ in range(2):
        dist.all_reduce(buf)
    dist.barrier()
    if red_dev.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_it):
        dist.all_reduce(buf)
    if red_dev.type == "cuda":
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_it
    tm = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dt = float(tm.item())
    distinct = len(set(idents))
    return {"backend": backend + (" (RCCL)" if backend == "nccl" else " (dry run: host transport, ranks may share a GPU)"),
            "rccl_version": version, "ranks": world, "devices": idents, "distinct_devices": distinct,
            "all_ranks_on_distinct_gpus": distinct == world,
            "allreduce_table_gradient": {"bytes": table_bytes, "iterations": n_it, "ms": round(dt * 1e3, 4),
                                         "bus_gb_per_s": round(2 * (world - 1) / world * table_bytes / dt / 1e9, 1),
                                         "what": "dist.all_reduce of one fp32 table gradient, max over ranks; bus bandwidth = "
                                                 "2 (N-1)/N x bytes / time (the per-link figure a ring is bound by)"}}


def train_probe(dev, rank=0, world=1, red_dev=None, steps=20, warmup=5, stage="instance", mode="eager", schedule=None,
                fp16=False, bound=1, dt_gamma=0.0):
    """Secondary measurement (not the headline value): instance-field training step, BASELINE configs[2]
    (K=64 logits, 4096 rays/batch per GPU, NeRF frozen): march -> frozen NeRF (fused) -> instance grid encode ->
    MLP -> K-channel compositing -> CE -> backward (atomic scatter) -> [gradient all-reduce] -> fused Adam.
    With world > 1 this is configs[3]: every rank draws its own rays, parameters are replicated and the
    gradients (49 MB hash table + MLP) are all-reduced over RCCL each step; all ranks must call it.
    stage="nerf": the same loop for the NeRF itself (MSE on rgb; table + sigma/colour nets trained).
    bound > 1 (round 5): the room enlarged `bound` times in a volume of 1 + log2(bound) occupancy cascades, level table up
    to 2048 * bound, steps growing with dt_gamma - the configuration a real 3D-FRONT scene trains at under torch-ngp."""
    import argparse
    import torch.distributed as dist
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    from instance_nerf_amd.nerf.utils import Trainer, grad_sync as _grad_sync
    torch.manual_seed(0)                   # replicated initial parameters
    net = NeRFNetwork(cuda_ray=True, bound=bound, min_near=0.05, density_thresh=10,
                      num_instances=64 if stage == "instance" else 0).to(dev)
    ds = SyntheticRoomDataset(dev, num_rays=4096, num_instances=64, rank=rank, scale=float(bound))
    net.density_bitfield.copy_(torch.from_numpy(ds.room.density_bitfield(128, float(bound))).to(dev))
    # mode "pipelined" (one process): Trainer(use_graph=True, look_ahead=True) - every step is ONE hipGraph replay that
    # also holds, forked off before the scatter, the parameter-independent head of the next batch on a second stream
    # (march; in the instance stage also the frozen NeRF's forward and the weight compositing)
    piped = mode == "pipelined" and world == 1
    if schedule is not None:               # "all_reduce" | "reduce_scatter": how the table gradient crosses the links
        _grad_sync.schedule = schedule
    tr = Trainer("bench", None, net, stage=stage, device=dev, iters=1000, update_extra_interval=16,
                 local_rank=rank, world_size=world, ema_decay=0.95,     # upstream's main scripts train with the EMA on
                 use_graph=piped, look_ahead=piped, shade_ahead=piped,
                 fp16=fp16,             # upstream's -O: the frozen NeRF of the instance stage with -O's numerics
                 workspace=None, use_checkpoint="scratch", mute=True)   # a measurement: no checkpoint look-up, nothing on stdout
    if dt_gamma:
        tr.opt = argparse.Namespace(dt_gamma=float(dt_gamma), max_steps=1024, T_thresh=1e-4)
    # Upstream's loop, occupancy update included: every 16 steps update_extra_state() queries the density of 128^3
    # (later 128^3 / 2) cells, refreshes the grid / bitfield and sets mean_count, which sizes the sample buffers of
    # the next 16 steps (no host sync inside a step).  The field is untrained here, so the grid it produces says
    # nothing about the scene: the update runs - and is timed - in full, then the analytic bitfield of the synthetic
    # room is put back (a 256 KB device copy).
    analytic = net.density_bitfield.clone()
    real_update = net.update_extra_state
    n_updates = [0]

    def update_and_restore(*a, **kw):
        real_update(*a, **kw)
        net.density_bitfield.copy_(analytic)
        n_updates[0] += 1
    net.update_extra_state = update_and_restore
    tr.global_step = 1                     # the first update comes after 15 steps, like every later one
    batches = [ds.batch() for _ in range(4)]
    # the step's dominant kernel is the table-gradient scatter (k_grid_bwd): events around its launch, on its stream
    from instance_nerf_amd.nerf import network as _network_mod
    scatter_events = []
    real_table_backward = _network_mod._table_backward

    def timed_table_backward(*a, **kw):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        out = real_table_backward(*a, **kw)
        e1.record(st)
        scatter_events.append((e0, e1))
        return out
    if not piped:                          # (a captured step cannot hold timing events)
        _network_mod._table_backward = timed_table_backward

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first = last = 0.0
    # upstream's loop.  INR_BENCH_LOOK_AHEAD=1: with ONE batch of look-ahead (Trainer(look_ahead=True)): the next batch's
    # ray/box test and march are queued on a side stream beside this step's scatter - off by default: in THIS loop the
    # extra host work per step cancels the ~35 us it hides (profiles/r03_NOTES.txt 16)
    look_ahead = os.environ.get("INR_BENCH_LOOK_AHEAD", "0") == "1" or piped
    nxt = (lambda j: batches[j % 4]) if look_ahead else (lambda j: None)
    for i in range(max(warmup, 4) + 16):   # >= one occupancy update: mean_count is set, the steady state begins
        l = float(tr.train_one_step(batches[i % 4], nxt(i + 1)))
        first = l if i == 0 else first
    k = max(warmup, 4) + 16
    import gc
    gc.collect()                           # the render network of the headline measurement dies here, not mid-loop -
    gc.disable()                           # and BEFORE the last warm-up steps: the first step after a full collection
    #                                        costs the host ~0.8 ms more (measured), which would land in the timed region
    spin = 0
    while tr.global_step % 16 != 1 or spin < 2:     # start the timed region right after an update: K timed steps then
        tr.train_one_step(batches[k % 4], nxt(k + 1))      # contain floor(K / 16) updates (1 for the default K = 20)
        k += 1
        spin += 1
    assert net.mean_count > 0
    per_step = torch.zeros(steps, dtype=torch.int32, device=dev)      # samples of each timed step: ONE tiny launch per step
    for i in range(2):                     # loads the code object of the counting op below, untimed
        torch.clamp(net.last_counter[0], max=net.mean_count, out=per_step[0])
    per_step.zero_()

    def timed_region():
        nonlocal k, last
        per_step.zero_()
        n_updates[0] = 0
        scatter_events.clear()
        barrier()
        t0 = time.perf_counter()
        stamps = [t0]
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        marks[0].record()
        for i in range(steps):
            last = tr.train_one_step(batches[(k + i) % 4], nxt(k + i + 1))
            torch.clamp(net.last_counter[0], max=net.mean_count, out=per_step[i])     # the march THIS step consumed
            marks[i + 1].record()
            stamps.append(time.perf_counter())
        # the time the host needs to QUEUE a step: the median over the steps (the step with the occupancy update waits
        # for the device inside its read-back); == ms_per_step when the host is the limit
        host = float(np.median(np.diff(stamps))) * steps
        barrier()
        el = time.perf_counter() - t0
        k += steps
        sm = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]       # device-side time of every step
        sc = sum(a.elapsed_time(b) for a, b in scatter_events) / max(len(scatter_events), 1)
        cnt = int(per_step.sum().item())
        if world > 1:
            t = torch.tensor([el, float(cnt)], dtype=torch.float64, device=red_dev)
            tm, ts = t[:1].clone(), t[1:].clone()
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dist.all_reduce(ts, op=dist.ReduceOp.SUM)
            return float(tm.item()), host, sm, sc, cnt, float(ts.item()), n_updates[0], len(scatter_events)
        return el, host, sm, sc, cnt, float(cnt), n_updates[0], len(scatter_events)

    # TWO timed regions of `steps` steps, each bracketed by barriers and aligned the same way to the occupancy updates;
    # the line reports the faster one and lists both: a box that is still tearing down an earlier process (the GPU test
    # suite) now and then stalls ONE step for 20-30 ms (profiles/r03_NOTES.txt 17), which says nothing about the path
    last = 0.0
    regions = [timed_region()]
    while tr.global_step % 16 != 1:
        tr.train_one_step(batches[k % 4], nxt(k + 1))
        k += 1
    regions.append(timed_region())
    gc.enable()
    _network_mod._table_backward = real_table_backward
    # ms_per_step = the MEAN of the two regions (round-3 verdict: not the faster one); the steps' device times, the host
    # time and the scatter's launches are those of the first region
    elapsed, host_s, step_ms, scatter_ms, n, n_all, n_upd, n_scatter = regions[0]
    if abs(regions[1][5] - regions[0][5]) <= 0.02 * regions[0][5]:
        elapsed = 0.5 * (regions[0][0] + regions[1][0])
    dt = elapsed / steps
    reduced = sum(p.numel() for g in tr.optimizer.param_groups for p in g["params"]) * 4
    # per live sample 1024 B per grid forward + 2048 B per TRAINED grid backward; per step the optimiser
    # sweep of the trained grid (7 x 49 MB).  Instance stage: frozen NeRF fwd + instance fwd + instance bwd = 4096 B;
    # NeRF stage: fwd + bwd of the one grid = 3072 B.  Rank 0's own samples over its own step time.
    per_sample = 4096 if stage == "instance" else 3072
    table_bytes = int(net.encoder.embeddings.numel()) * 4
    adam_bytes = 7 * table_bytes
    own_dt = dt
    step_bytes = per_sample * (n / steps) + adam_bytes
    roofline = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "kernel": "k_grid_bwd (table-gradient scatter, fp32 atomics; opt-in int32 sums with INR_FX_GRAD=1; "
                          "memory-side atomic request rate, profiles/r06_NOTES.txt)",
                "algorithmic_bytes_per_sample": 2048, "launches": n_scatter,
                "avg_launch_ms": round(scatter_ms, 4),
                "achieved": round(2048 * (n / steps) / (scatter_ms / 1e3) / 1e9, 1) if scatter_ms > 0 else None,
                "frac": round(2048 * (n / steps) / (scatter_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if scatter_ms > 0 else None,
                "traffic": None,
                "atomic_unit": None,
                "step": {"algorithmic_bytes_per_sample": per_sample, "optimizer_bytes_per_step": adam_bytes,
                         "achieved": round(step_bytes / own_dt / 1e9, 1),
                         "frac": round(step_bytes / own_dt / 1e9 / HBM_PEAK_GBS, 4)}}
    sr = scatter_requests(stage)
    if sr is not None and scatter_ms > 0:
        # what really bounds the scatter: every fp32 atomic is forwarded to the memory side, which takes ~21 G 64-byte
        # requests per second whatever their width, the table size or the occupancy (profiles/r03_NOTES.txt 1)
        req = sr[0] * (n / steps)
        roofline["atomic_unit"] = {"requests_per_sample": sr[0], "requests_per_launch": int(req),
                                   "achieved_g_requests_per_s": round(req / (scatter_ms / 1e3) / 1e9, 2),
                                   "peak_g_requests_per_s": round(sr[1] / 1e9, 1),
                                   "frac": round(req / (scatter_ms / 1e3) / sr[1], 4),
                                   "source": SCATTER_JSON + " (rocprofv3 PMC on these kernel sources; peak from "
                                             "tools/micro/atomic_type_bench.hip)"}
    what = ("instance-field training step, K=64, 4096 rays/batch per GPU, NeRF frozen, parameter EMA 0.95 "
            f"(BASELINE configs[{2 if world == 1 else 3}])") if stage == "instance" else \
        "NeRF training step (MSE on rgb: hash table + sigma/colour nets, parameter EMA 0.95), 4096 rays/batch per GPU"
    if bound != 1:
        what += (f"; bound {bound}: room enlarged {bound}x, {net.cascade} occupancy cascades, finest level {2048 * bound}, "
                 f"dt_gamma {dt_gamma:g}")
    return {"workload": what,
            "n_gpus": world, "timed_steps": steps, "ms_per_step": round(dt * 1e3, 3),
            # median / max of the steps' device-side times: one stalled step (a host hiccup, an allocator slow path, the
            # occupancy update) moves the mean above but not the median
            "ms_per_step_median": round(float(np.median(step_ms)), 3), "ms_per_step_max": round(max(step_ms), 3),
            # the loop as it runs between two synchronisations: the first timed step starts on a device the barrier has
            # just idled (host-paced, ~2x a step) - the mean device time of the others, occupancy update included
            "ms_per_step_running": round(float(np.mean(step_ms[1:])), 3) if steps > 1 else None,
            "ms_of_each_step": [round(v, 3) for v in step_ms],
            "host_enqueue_ms_per_step": round(host_s / steps * 1e3, 3),     # a busy host shows here first
            "samples_per_step": int(n_all) // steps,
            "msamples_per_s": round(n_all / steps / dt / 1e6, 2),
            # instance stage: rays whose mask label is -1 (10 % of the synthetic loader's, as unmatched segments are in
            # match_seg.py's output) carry no loss; Trainer(prune_ignored=True) never marches them
            "ignored_rays": ({"fraction_of_the_batch": round(float(np.mean([float((b["masks"] < 0).float().mean()) for b in batches])), 4),
                              "marched": not tr.prune_ignored} if stage == "instance" else None),
            "allreduce_mb_per_step": round(reduced / 1e6, 1) if world > 1 else 0.0,
            "gradient_schedule": (f"{_grad_sync.schedule}, payload {_grad_sync.payload}, "
                                  f"{'started inside the backward' if _grad_sync.enabled else 'after the backward'}") if world > 1 else None,
            "occupancy_updates_in_timed_steps": n_upd, "roofline": roofline,
            "ms_per_step_of_both_timed_regions": [round(r[0] / steps * 1e3, 3) for r in regions],
            "mode": ("pipelined: one hipGraph per step, next batch's head on a second stream inside it" if piped else
                     "eager: upstream's loop, one stream"),
            "graphs_captured": (sorted("own head" * k[1] + "prefetched head" * (not k[1]) + " + look-ahead" * k[2]
                                       for k in tr._pipe["graphs"]) if piped and tr._pipe else None),
            "loss_first": round(first, 4), "loss_last": round(float(last), 4)}


def config5_probe(dev):
    """Secondary measurement, BASELINE configs[4]: rgb-sigma lattice extraction at 160^3 (one fused launch per chunk:
    gather + sigma net once per voxel, colour net for 4 fixed view directions) + 3-D RoIAlign of 256 boxes to 10^3 bins
    on a [1,256,40,40,40] feature volume.  Replicas only: one scene per GPU, no collective."""
    from instance_nerf_amd.extract import extract_rgbsigma
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    net, _ = build_network(dev)
    extract_rgbsigma(net, max_side=160)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        _, res = extract_rgbsigma(net, max_side=160)
    torch.cuda.synchronize()
    t_ext = (time.perf_counter() - t0) / 5
    feat = torch.randn(1, 256, 40, 40, 40, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    lo = torch.rand(256, 3, device=dev, generator=gen) * 100
    rois = torch.cat([lo, lo + 10 + torch.rand(256, 3, device=dev, generator=gen) * 50], 1)
    inds = torch.zeros(256, dtype=torch.int32, device=dev)

    def event_ms(fn, n):
        fn()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            fn()
            ev[i + 1].record()
        torch.cuda.synchronize()
        return sum(ev[i].elapsed_time(ev[i + 1]) for i in range(n)) / n

    t_roi = event_ms(lambda: roi_align_3d(feat, rois, inds, 10, 10, 10, 0.25), 20) * 1e-3
    x = feat.clone().requires_grad_(True)
    out = roi_align_3d(x, rois, inds, 10, 10, 10, 0.25)
    g = torch.randn_like(out)

    def bwd():
        x.grad = None
        out.backward(g, retain_graph=True)
    t_bwd = event_ms(bwd, 5) * 1e-3
    n = int(res.prod())
    # byte floors (the roofline that bounds both: HBM, 8 TB/s).  RoIAlign: every output element written once and the
    # feature volume read once = 262.1 + 65.5 MB.  Extraction: 1024 B of table rows per voxel (the gather of the
    # fused field kernel; the sigma net once and the colour net 4 times per voxel ride on it) + the 16 B row written.
    roi_bytes = out.numel() * 4 + feat.numel() * 4
    ext_bytes = n * (1024 + 16)
    return {"workload": "rgb-sigma extraction 160^3 (4 view directions) + RoIAlign-3D 256 boxes -> 10^3 x 256 ch on "
                        "[1,256,40,40,40] (BASELINE configs[4]), per GPU",
            "extract_ms": round(t_ext * 1e3, 3), "extract_mvoxels_per_s": round(n / t_ext / 1e6, 1),
            "extract_roofline": {"bound": "hbm", "achieved": round(ext_bytes / t_ext / 1e9, 1), "peak": 8000.0,
                                 "unit": "GB/s", "frac": round(ext_bytes / t_ext / 8e12, 4),
                                 "bytes_per_voxel": 1040, "note": "wall clock over 5 extractions, lattice generation included"},
            "roi_align_forward_ms": round(t_roi * 1e3, 4),
            "roi_align_backward_ms": round(t_bwd * 1e3, 4),
            "roi_align_roofline": {"bound": "hbm", "achieved": round(roi_bytes / t_roi / 1e9, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(roi_bytes / t_roi / 8e12, 4),
                                   "compulsory_mb": round(roi_bytes / 1e6, 1),
                                   "kernel": "k_roi_align3d_sep_fwd<4, false> (separable; events on the launch stream, 20 launches)",
                                   "backward_note": "the autograd backward: inr_roi_align_3d_backward_ws = zero fill of the 65.5 MB channels-fastest "
                                                    "scratch + k_roi_align3d_sep_bwd_cl + transposing copy into grad_input "
                                                    "(in place, round 4: k_roi_align3d_sep_bwd + zero fill, 0.58-0.60 ms)"}}


def instance_render_probe(dev, frames=8):
    """Secondary measurement: the same 800x800 views rendered WITH the instance head (K = 64 logits composited per
    pixel; SURVEY a13 at render time): march -> NeRF field (table feed) -> compositing with weights -> instance field
    with `w * logits` accumulated on chip (k_instance_render).  2048 B of algorithmic table traffic per sample."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.utils import get_rays
    from instance_nerf_amd.scene import RoomScene
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=64).to(dev).eval()
    room = RoomScene()
    net.density_bitfield.copy_(torch.from_numpy(room.density_bitfield(128, 1.0)).to(dev))
    poses, intr, H, W = room.cameras()
    pd = torch.from_numpy(poses).to(dev)

    def frame(v):
        r = get_rays(pd[v:v + 1], intr, H, W, patch=4)
        with torch.no_grad():
            return net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
    frame(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    counts = [frame(v % pd.shape[0])["num_samples"] for v in range(frames)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(int(c[0]) for c in counts)
    out = {"workload": "render 800x800 with the instance head, K=64 logits per pixel (both fields evaluated per sample)",
           "ms_per_frame": round(dt / frames * 1e3, 3), "value": round(n / dt / 1e6, 1), "unit": "Msamples/s",
           "algorithmic_bytes_per_sample": 2 * BYTES_PER_SAMPLE,
           "frac_of_hbm_peak": round(n * 2 * BYTES_PER_SAMPLE / dt / 1e9 / HBM_PEAK_GBS, 4)}
    # the product's view loop for this network (Trainer.test of the instance stage): row-major rays from the loader, views
    # alternating on the two streams of the trainer's FramePipeline, the instance render behind the field gate too
    try:
        from instance_nerf_amd.nerf.utils import Trainer
        tr = Trainer("bench_inst_views", None, net, stage="instance", device=dev, workspace=None, use_checkpoint="scratch",
                     mute=True)
        for q in net.parameters():
            q.requires_grad_(False)
        net.eval()

        def loader(k):
            for v in range(k):
                r = get_rays(pd[v % pd.shape[0]:v % pd.shape[0] + 1], intr, H, W)
                yield {"rays_o": r["rays_o"], "rays_d": r["rays_d"], "H": H, "W": W}
        for _ in tr.render_sequence(loader(3), infer_mode="fused"):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cs = [o["num_samples"] for _, o in tr.render_sequence(loader(frames), infer_mode="fused")]
        torch.cuda.synchronize()
        dt_p = time.perf_counter() - t0
        n_p = sum(int(c[0]) for c in cs)
        out["pipelined"] = {"ms_per_frame": round(dt_p / frames * 1e3, 3), "value": round(n_p / dt_p / 1e6, 1),
                            "frac_of_hbm_peak": round(n_p * 2 * BYTES_PER_SAMPLE / dt_p / 1e9 / HBM_PEAK_GBS, 4),
                            "what": "Trainer.render_sequence (FramePipeline), the loop Trainer.test runs"}
    except Exception as e:                                    # noqa: BLE001
        out["pipelined"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    # the same frames with upstream's -O numerics on BOTH fields (opt-in: fp16 table copies, single-pass fp16 MLPs)
    ref = frame(0)
    net.half_table = net.mlp_fp16 = True
    fast = frame(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for v in range(frames):
        frame(v % pd.shape[0])
    torch.cuda.synchronize()
    dt_o = time.perf_counter() - t0
    out["O_numerics"] = {"ms_per_frame": round(dt_o / frames * 1e3, 3), "value": round(n / dt_o / 1e6, 1),
                         "max_abs_diff_image": float((fast["image"] - ref["image"]).abs().max()),
                         "max_abs_diff_logits": float((fast["instance"] - ref["instance"]).abs().max()),
                         "max_abs_logit": float(ref["instance"].abs().max())}
    return out


BOUND_TRAFFIC_JSON = os.path.join("profiles", "r06_bound_traffic.json")


def bound_traffic(bound, dt_gamma, frame_path="fused"):
    """Fabric read requests per sample of the frame path taken (fused kernel | pre-pass + kernel) at this configuration, from the committed PMC profile
    (profiles/r06_bound_traffic.json: tools/pmc_bound.sh + tools/bound_traffic_json.py); quoted only for the kernel sources
    it was measured on.  -> (record of the configuration, random-line rate of the fabric in requests/s) or None."""
    from instance_nerf_amd import build
    path = os.path.join(ROOT, BOUND_TRAFFIC_JSON)
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    if t.get("source_sha") != build.source_sha("field"):
        return None
    key = f"{bound}:{bound}:{int(round(1 / dt_gamma)) if dt_gamma else 0}:0:0" + ("-sliced" if frame_path == "sliced" else "")
    rec = t["configs"].get(key)
    return None if rec is None else (rec, float(t["random_line_rate_of_the_fabric_g_per_s"]) * 1e9)


def bound_render_probe(dev, bound=4, dt_gamma=1.0 / 128, frames=8):
    """Secondary measurement (round-4 verdict item 1): the headline render off its tuned configuration - the synthetic
    room enlarged `bound` times inside a bound-`bound` volume: 1 + log2(bound) occupancy cascades, level table up to
    2048 * bound (13 of 16 levels hashed at bound 8), steps growing with the distance (dt_gamma) or constant (0).
    Same loop as the headline's one-stream leg: 800x800 views through net.render (fused), events around the field kernel
    on its stream.  Upstream init U(-1e-4, 1e-4) (transparent field: every marched sample is evaluated)."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.utils import get_rays
    from instance_nerf_amd.scene import RoomScene
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True, bound=bound, min_near=0.05, density_thresh=10).to(dev).eval()
    room = RoomScene(scale=float(bound))
    net.density_bitfield.copy_(torch.from_numpy(room.density_bitfield(128, float(bound))).to(dev))
    poses, intr, H, W = room.cameras()
    pd = torch.from_numpy(poses).to(dev)
    ev = []
    inner = net.forward_table

    def timed(*a, **kw):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        out = inner(*a, **kw)
        e1.record(st)
        ev.append((e0, e1))
        return out
    net.forward_table = timed

    def frame(v):
        r = get_rays(pd[v:v + 1], intr, H, W, patch=4)
        with torch.no_grad():
            return net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused", dt_gamma=dt_gamma)
    for v in range(8):                     # (frame_slices = "auto": its four probing frames and the decision land here)
        frame(v % pd.shape[0])
        torch.cuda.synchronize()
    ev.clear()
    t0 = time.perf_counter()
    counts = [frame(v % pd.shape[0])["num_samples"] for v in range(frames)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(int(c[0]) for c in counts)
    kms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
    tb = net.encoder.table
    probe = getattr(net, "_slice_probe", None) or {}
    path = "sliced" if probe.get("choice") else "fused"
    # what binds the kernel off the tuned configuration is not the algorithmic byte count but the number of 128-byte
    # fabric requests its L2 misses cause (profiles/r05_NOTES.txt 1-2): both fractions side by side
    fabric = None
    bt = bound_traffic(bound, dt_gamma, path)
    if bt is not None and kms > 0:
        rec, ceiling = bt
        req_s = rec["fabric_read_requests_per_sample"] * (n / frames) / (kms / 1e3)
        fabric = {"frame_path": path, "requests_per_sample": rec["fabric_read_requests_per_sample"], "bytes_per_sample": rec["fabric_read_bytes_per_sample"],
                  "l2_hit_rate": rec["l2_hit_rate"], "achieved_g_requests_per_s": round(req_s / 1e9, 1),
                  "traffic_gb_per_s": round(req_s * 128 / 1e9, 1), "ceiling_g_requests_per_s": round(ceiling / 1e9, 1),
                  "frac": round(req_s / ceiling, 4),
                  "note": ("behind the level-major pre-pass the fabric is no longer the limit (most fine-level lines come from "
                           "the L2 the level fits in); the gather is then bound by L1 fills (profiles/r05_NOTES.txt 3)")
                  if path == "sliced" else "the fused kernel off the tuned configuration is bound by this request rate",
                  "source": BOUND_TRAFFIC_JSON + " (rocprofv3 PMC on these kernel sources, view 0; ceiling: random 128-byte lines "
                                                 "over eight 4 MiB levels, tools/micro/level_xcd_bench.hip)"}
    return {"fabric_requests": fabric,
            "frame_path": {"taken": path, "mode": str(net.frame_slices),
                           "probed_ms_per_msample": {k: [round(v * 1e6, 4) for v in vs] for k, vs in
                                                     (("fused", probe.get("ms", {}).get(False, [])),
                                                      ("sliced", probe.get("ms", {}).get(True, [])))},
                           "what": "fused = one kernel; sliced = the three finest levels by a level-major pre-pass, then the "
                                   "fused kernel on the other thirteen (inr_nerf_forward_table_sliced); auto keeps the faster"},
            "workload": f"render 800x800, sigma+rgb, room enlarged {bound}x in a bound-{bound} volume: {net.cascade} occupancy "
                        f"cascades, levels 16 .. {int(tb['resolutions'][-1])} ({int(tb['hashed'].sum())} of 16 hashed, "
                        f"T = {tb['total_rows']}), dt_gamma {dt_gamma:g}",
            "ms_per_frame": round(dt / frames * 1e3, 3), "value": round(n / dt / 1e6, 1), "unit": "Msamples/s",
            "samples_per_frame": n // frames, "field_kernel_ms": round(kms, 4),
            "algorithmic_bytes_per_sample": BYTES_PER_SAMPLE,
            "field_frac_of_hbm_peak": round(n / frames * BYTES_PER_SAMPLE / (kms / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
            "end_to_end_frac_of_hbm_peak": round(n * BYTES_PER_SAMPLE / dt / 1e9 / HBM_PEAK_GBS, 4)}


def render_sharded_probe(dev, rank, world, red_dev, frames=8, res=800):
    """world > 1, EVERY rank calls it.  STRONG scaling of the render path (round-4 verdict item 7; the headline is weak
    scaling: one whole view per rank): ONE 800x800 view at a time, split over the ranks the way the product splits it
    (`nerf/utils.py::shard_indices`: 1024-ray chunks of the 4x4-patch-ordered ray list dealt round robin - contiguous
    ranges are as unbalanced as the frame's rows), parameters and bitfield replicated, no collective on the critical
    path.  Time per frame = max over ranks (barrier before and after), samples = sum over ranks; then the
    product's `render_sharded` itself, which also all-gathers image / depth / opacity onto every rank."""
    import torch.distributed as dist
    from instance_nerf_amd.nerf.utils import get_rays, render_sharded, shard_indices
    net, room = build_network(dev)
    poses, intr, H, W = room.cameras(H=res, W=res, focal=res / 2.0)
    pd = torch.from_numpy(poses).to(dev)
    mine_idx = shard_indices(H * W, rank, world, dev)

    def rays(v):
        r = get_rays(pd[v:v + 1], intr, H, W, patch=4)
        return r["rays_o"], r["rays_d"]

    def shard(v):
        ro, rd = rays(v)
        with torch.no_grad():
            return net.render(ro[:, mine_idx].contiguous(), rd[:, mine_idx].contiguous(), bg_color=1, infer_mode="fused")

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
    shard(0)
    shard(1)
    barrier()
    t0 = time.perf_counter()
    counts = [shard(v % pd.shape[0])["num_samples"] for v in range(frames)]
    barrier()
    el = time.perf_counter() - t0
    mine = float(sum(int(c[0]) for c in counts))
    t = torch.tensor([el, mine], dtype=torch.float64, device=red_dev)
    tm, ts = t[:1].clone(), t[1:].clone()
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)
    dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    el, total = float(tm[0]), float(ts[0])
    out = {"workload": f"render {res}x{res}, one view at a time, the view's rays dealt to {world} ranks in 1024-ray chunks "
                       "(nerf/utils.py::shard_indices), replicated parameters",
           "scaling": "strong", "n_gpus": world, "frames": frames, "rays_of_rank_0": int(mine_idx.numel()),
           "ms_per_frame": round(el / frames * 1e3, 3), "value": round(total / el / 1e6, 1), "unit": "Msamples/s"}
    # the product's entry point: the same shards + the all-gather of every per-ray result
    try:
        ro, rd = rays(0)
        render_sharded(net, ro, rd, rank, world, bg_color=1, infer_mode="fused")
        barrier()
        t0 = time.perf_counter()
        for v in range(frames):
            ro, rd = rays(v % pd.shape[0])
            full = render_sharded(net, ro, rd, rank, world, bg_color=1, infer_mode="fused")
        barrier()
        el_g = time.perf_counter() - t0
        tg = torch.tensor([el_g], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        el_g = float(tg[0])
        out["with_gather"] = {"ms_per_frame": round(el_g / frames * 1e3, 3), "value": round(total / el_g / 1e6, 1),
                              "bytes_gathered_per_frame": H * W * 20,
                              "full_frame_on_every_rank": list(full["image"].shape) == [1, H * W, 3],
                              "what": "nerf/utils.py::render_sharded: image, depth and opacity all-gathered onto every rank"}
        ok = 1.0
    except Exception as e:                                    # noqa: BLE001 - the same code on every rank: all fail together
        out["with_gather"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def half_table_probe(dev, frames=8, mlp_fp16=False):
    """Secondary measurement: the headline frames with the OPT-IN half-precision table copy (NeRFNetwork.half_table;
    upstream's -O / fp16 storage): 512 B of algorithmic table traffic per sample.  Not the headline: its outputs differ
    from the fp32 table's by ~1e-3 relative.  mlp_fp16: additionally the single-pass fp16 MLP (NeRFNetwork.mlp_fp16) -
    both halves of upstream's -O; the object then carries the largest difference to the default path on view 0."""
    from instance_nerf_amd.nerf.utils import get_rays
    net, room = build_network(dev)
    net.half_table = True
    net.mlp_fp16 = bool(mlp_fp16)
    poses, intr, H, W = room.cameras()
    pd = torch.from_numpy(poses).to(dev)
    ev = []
    inner = net.forward_table

    def timed(*a, **kw):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        out = inner(*a, **kw)
        e1.record(st)
        ev.append((e0, e1))
        return out
    net.forward_table = timed

    def frame(v):
        r = get_rays(pd[v:v + 1], intr, H, W, patch=4)
        with torch.no_grad():
            return net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
    frame(0)
    torch.cuda.synchronize()
    ev.clear()
    t0 = time.perf_counter()
    counts = [frame(v % pd.shape[0])["num_samples"] for v in range(frames)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = sum(int(c[0]) for c in counts)
    kms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
    out = {"workload": "render 800x800, sigma+rgb, fp16 copy of the hash table (opt-in, NeRFNetwork.half_table)"
                       + (" + single-pass fp16 MLP (opt-in, NeRFNetwork.mlp_fp16): upstream's -O numerics class" if mlp_fp16 else ""),
           "ms_per_frame": round(dt / frames * 1e3, 3), "value": round(n / dt / 1e6, 1), "unit": "Msamples/s",
           "field_kernel_ms": round(kms, 4), "algorithmic_bytes_per_sample": BYTES_PER_SAMPLE // 2,
           "field_frac_of_hbm_peak": round(n / frames * (BYTES_PER_SAMPLE // 2) / (kms / 1e3) / 1e9 / HBM_PEAK_GBS, 4)}
    if mlp_fp16:
        fast = frame(0)["image"]
        net.half_table = net.mlp_fp16 = False
        ref = frame(0)["image"]
        mse = float(((fast - ref) ** 2).mean())
        out["vs_default_path"] = {"max_abs_diff": float((fast - ref).abs().max()),
                                  "psnr_db": round(-10 * math.log10(max(mse, 1e-20)), 1)}
    return out


def timed_trained_steps(tr, net, ds, stage, n=128, ab=True):
    """Steady-state training steps of a Trainer whose scene is already TRAINED (opaque surfaces, learned occupancy grid:
    2.4x the samples per step of the untrained bench scene), eager loop over pre-made batches of the on-disk loader
    (GPU-resident images and masks): ms per step from device events, samples per step from the march's device counter,
    and the table-gradient scatter's share with its request-rate roofline (round-4 verdict item 2c; since round 6 THE
    training figure of the line: this is where a real run spends its time).  -> the "train_step" object of the trained
    scene for this stage."""
    from instance_nerf_amd.nerf import network as _network_mod
    ev = []
    real = _network_mod._table_backward

    def timed(*a, **kw):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        out = real(*a, **kw)
        e1.record(st)
        ev.append((e0, e1))
        return out
    batches = [ds[i % len(ds)] for i in range(8)]
    for i in range(8):
        tr.train_one_step(batches[i])
    _network_mod._table_backward = timed
    try:
        per = torch.zeros(n, dtype=torch.int32, device=tr.device)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        torch.cuda.synchronize()
        marks[0].record()
        for i in range(n):
            tr.train_one_step(batches[i % 8])
            torch.clamp(net.last_counter[0], max=max(int(net.mean_count), 1), out=per[i])
            marks[i + 1].record()
        torch.cuda.synchronize()
    finally:
        _network_mod._table_backward = real
    ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(n)]
    sc = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
    samples = float(per.sum().item()) / n
    out = {"steps": n, "ms_per_step": round(float(np.mean(ms)), 3), "ms_per_step_median": round(float(np.median(ms)), 3),
           "samples_per_step": int(samples), "msamples_per_s": round(samples / (float(np.median(ms)) * 1e-3) / 1e6, 1),
           "scatter_ms": round(sc, 4), "scatter_share_of_step": round(sc / float(np.median(ms)), 3),
           "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                        "kernel": "k_grid_bwd (table-gradient scatter)", "algorithmic_bytes_per_sample": 2048,
                        "achieved": round(2048 * samples / (sc / 1e3) / 1e9, 1) if sc > 0 else None,
                        "frac": round(2048 * samples / (sc / 1e3) / 1e9 / HBM_PEAK_GBS, 4) if sc > 0 else None,
                        "atomic_unit": None}}
    sr = scatter_requests(stage)
    if sr is not None and sc > 0:
        req = sr[0] * samples
        out["roofline"]["atomic_unit"] = {"requests_per_sample": sr[0], "achieved_g_requests_per_s": round(req / (sc / 1e3) / 1e9, 2),
                                          "peak_g_requests_per_s": round(sr[1] / 1e9, 1), "frac": round(req / (sc / 1e3) / sr[1], 4),
                                          "source": SCATTER_JSON + " (requests per sample measured on the UNTRAINED bench scene)"}
    # opt-in A/B (round 6, nerf/network.py::FX_GRAD): the same steps with the table gradient summed as int32 fixed point - the
    # memory-side atomic unit takes integer adds 28 % faster and steps become bit-reproducible; how often did a level come
    # within 8x of the int32 range ("near miss") and how much of the range did the largest row sum ever use
    table = net.instance_encoder.embeddings if stage == "instance" else net.encoder.embeddings
    out["scatter_form"] = f"int{_network_mod.fx_bits()} sums" if _network_mod.FX_GRAD else "fp32 atomics"
    if ab and not _network_mod.FX_GRAD:
        for key, form in (("fixed_point", 32), ("fixed_point64", 64)):
            _network_mod.FX_GRAD = form
            try:
                other = timed_trained_steps(tr, net, ds, stage, n=96, ab=False)
                st = getattr(table, "_fx_state", None)
                h = st[:96].cpu().numpy() if st is not None else np.zeros(96, np.float32)
                out[key] = {**{k: other[k] for k in ("steps", "ms_per_step", "ms_per_step_median", "scatter_ms", "scatter_share_of_step")},
                            "near_misses_so_far": int(h[49]), "peak_use_of_the_integer_range": round(float(h[80:96].max()), 4),
                            "what": f"opt-in (INR_FX_GRAD={form} / Trainer(fixed_point_grad={form})): int{form} sums of the table gradient"
                                    + ("; near miss = a level-step whose largest row sum used more than 1/8 of the range, peak use 1.0 "
                                       "would be a wrap (tools/fx_dynamics_probe.py)" if form == 32 else
                                       " in a separate accumulator: quantum 2e-16 of a level's maximum, invisible to Adam - the faithful "
                                       "order-independent form")}
            except Exception as e:                                # noqa: BLE001
                out[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                _network_mod.FX_GRAD = 0
    return out


def train_loop_probe(dev, scene, stage, state, K=16, min_steps=240):
    """Round-5 verdict item 2: the PRODUCT's training loop, loader included - `Trainer.train_one_epoch(NeRFDataset(path,
    device=dev, preload=True[, mask_dir]).dataloader())`, the loop a user of the reference's README.md:58-66 runs - on the
    trained synthetic room read back from disk (images + transforms_train.json + matched-mask .npy files), eager and as
    the captured two-stream pipeline (`use_graph + look_ahead`), parameter EMA 0.95 as upstream's main scripts.  Wall
    clock over >= `min_steps` steps (whole epochs; one synchronisation before, one after), next to the SAME trainer
    stepping through pre-made batches of the same loader (`train_one_step(batch, next_batch)`, no loader work inside the
    loop): `vs_premade_batches` = loader-loop step rate / pre-made step rate (the verdict's bar: >= 0.9 pipelined).
    `state`: the trained network's state_dict + occupancy scalars; every mode starts from it."""
    import math as _m
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import NeRFDataset
    from instance_nerf_amd.nerf.utils import Trainer
    out = {"what": "Trainer.train_one_epoch over NeRFDataset(preload=True).dataloader(): per step the loader draws 4096 "
                   "pixels of one image, generates their rays and gathers rgb"
                   + (" and the matched-mask labels" if stage == "instance" else "") + " on the GPU; EMA 0.95",
           "stage": stage, "views": None}
    for mode in ("eager", "pipelined"):
        piped = mode == "pipelined"
        try:
            torch.manual_seed(0)
            net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10,
                              num_instances=K if stage == "instance" else 0).to(dev)
            net.load_state_dict(state["sd"], strict=False)
            net.mean_density, net.iter_density, net.mean_count = state["mean_density"], state["iter_density"], state["mean_count"]
            ds = NeRFDataset(scene["path"], type="train", device=dev, scale=1.0, num_rays=4096, preload=True,
                             mask_dir=scene["mask_dir"] if stage == "instance" else None,
                             num_instances=K if stage == "instance" else 0)
            loader = ds.dataloader()
            out["views"] = len(ds)
            tr = Trainer(f"loop_{stage}_{mode}", None, net, stage=stage, device=dev, lr=1e-3, iters=10 ** 6, workspace=None,
                         ema_decay=0.95, use_graph=piped, look_ahead=piped, mute=True,
                         update_extra_interval=16 if stage == "nerf" else 10 ** 9)
            tr.global_step = 1 if stage == "instance" else 0
            epochs = max(1, _m.ceil(min_steps / len(ds)))
            for _ in range(max(3, _m.ceil(64 / len(ds)))):        # warm-up: buffer sizes settle, graphs are captured
                tr.train_one_epoch(loader)
            # (a) pre-made batches of this loader, stepped through directly
            batches = [ds[i % len(ds)] for i in range(8)]
            n_pre = epochs * len(ds)
            for i in range(8):
                tr.train_one_step(batches[i % 8], batches[(i + 1) % 8] if piped else None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_pre):
                tr.train_one_step(batches[i % 8], batches[(i + 1) % 8] if piped else None)
            torch.cuda.synchronize()
            dt_pre = (time.perf_counter() - t0) / n_pre
            # (b) the product's loop
            tr.train_one_epoch(loader)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(epochs):
                tr.train_one_epoch(loader)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (epochs * len(ds))
            # (c) samples per step of the loop: one more epoch with a one-launch counter read per step (untimed)
            per = torch.zeros(len(ds), dtype=torch.int32, device=dev)
            real_step, i_box = tr.train_one_step, [0]

            def counted(data, next_data=None):
                r = real_step(data, next_data)
                torch.clamp(net.last_counter[0], max=max(int(net.mean_count), 1), out=per[i_box[0] % len(ds)])
                i_box[0] += 1
                return r
            tr.train_one_step = counted
            tr.train_one_epoch(loader)
            tr.train_one_step = real_step
            samples = float(per.sum().item()) / len(ds)
            out[mode] = {"steps": epochs * len(ds), "epochs": epochs, "ms_per_step": round(dt * 1e3, 4),
                         "steps_per_s": round(1.0 / dt, 1), "samples_per_step": int(samples),
                         "msamples_per_s": round(samples / dt / 1e6, 1),
                         "premade_batches": {"steps": n_pre, "ms_per_step": round(dt_pre * 1e3, 4),
                                             "steps_per_s": round(1.0 / dt_pre, 1)},
                         "vs_premade_batches": round(dt_pre / dt, 4),
                         "loss_last_epoch": round(float(tr.stats["loss"][-1]), 5)}
            if piped:
                out[mode]["graphs_captured"] = len(tr._pipe["graphs"]) if tr._pipe else 0
        except Exception as e:                                    # noqa: BLE001
            out[mode] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def trained_scene_probe(dev, steps=1500, with_oracle=True, K=16):
    """Secondary measurement: a TRAINED scene.  The headline scene is an untrained (transparent) field, so no
    ray of it ever terminates; a trained 3D-FRONT room is opaque.  The synthetic room is written to disk the way the
    reference's stages read a scene (24 views of 400x400: images + transforms_train.json + matched-mask .npy files,
    `RoomScene.write_dataset`), the NeRF is trained from those files for `steps` steps by the product's own loop
    (`Trainer.train(NeRFDataset(...).dataloader())`: occupancy grid learned by update_extra_state), then: the
    steady-state training step on the trained scene (the training figure of record), the product's training loop with
    its loader against pre-made batches (`train_loop`), and the eight 800x800 bench views in the three inference modes:
    ms per frame, marched vs evaluated samples, time and roofline fraction of the field kernel, and - on 4096 random
    pixels of view 0 - the difference to the C oracle run with the trained weights and the learned bitfield."""
    import shutil
    import tempfile
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import NeRFDataset
    from instance_nerf_amd.nerf.utils import Trainer, get_rays
    from instance_nerf_amd.scene import RoomScene
    torch.manual_seed(0)
    room = RoomScene()
    scene_dir = tempfile.mkdtemp(prefix="inr_bench_scene_")
    t0 = time.perf_counter()
    scene = room.write_dataset(scene_dir, n_views=24, H=400, W=400, num_instances=K, ignore_frac=0.1)
    write_s = time.perf_counter() - t0
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10).to(dev)
    ds = NeRFDataset(scene_dir, type="train", device=dev, scale=1.0, num_rays=4096, preload=True)
    ds.room = room
    tr = Trainer("trained", None, net, stage="nerf", device=dev, lr=1e-2, iters=steps, workspace=None, mute=True)
    epochs = -(-steps // len(ds))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train(ds.dataloader(), max_epochs=epochs)
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    try:
        step_nerf = timed_trained_steps(tr, net, ds, "nerf")
    except Exception as e:                                    # noqa: BLE001
        step_nerf = {"error": f"{type(e).__name__}: {e}"[:300]}
    state = {"sd": {k: v.clone() for k, v in net.state_dict().items()}, "mean_density": net.mean_density,
             "iter_density": net.iter_density, "mean_count": net.mean_count}
    loops = {}
    try:
        loops["nerf"] = train_loop_probe(dev, scene, "nerf", state, K=K)
    except Exception as e:                                        # noqa: BLE001
        loops["nerf"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    net.eval()
    poses, intr, H, W = ds.room.cameras()
    pd = torch.from_numpy(poses).to(dev)
    ev = []

    def timed(fn):
        def wrapper(*a, **kw):
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            out = fn(*a, **kw)
            e1.record(st)
            ev.append((e0, e1))
            return out
        return wrapper
    net.forward_table = timed(net.forward_table)
    net.nerf_render = timed(net.nerf_render)
    out = {"workload": f"NeRF of the synthetic room trained {epochs * len(ds)} steps from disk ({len(ds)} views of 400x400 through "
                       "NeRFDataset, 4096 rays per step, learned occupancy grid), then the 8 bench views at 800x800",
           "train_seconds": round(train_s, 2), "train_steps_per_s": round(epochs * len(ds) / train_s, 1),
           "dataset_write_seconds": round(write_s, 1), "train_loop": loops,
           "occupied_cells": round(float((net.density_grid > min(net.mean_density, net.density_thresh)).float().mean()), 4),
           "train_step": {"what": "steady-state eager training steps ON the trained scene (128 steps after the training run): "
                                  "4096 rays per batch of the on-disk loader, learned occupancy grid", "nerf_stage": step_nerf}}
    frame0 = frame0_fast = None
    # "fused_O": the two-kernel path with both opt-in halves of upstream's -O (fp16 table copy + single-pass fp16 MLP)
    for mode in ("fused", "fused_terminate", "auto", "fused_O"):
        net.half_table = net.mlp_fp16 = mode == "fused_O"

        def frame(v, mode=mode):
            r = get_rays(pd[v:v + 1], intr, H, W, patch=4)
            with torch.no_grad():
                return r, net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused" if mode == "fused_O" else mode)
        frame(0)
        frame(1)                                       # "auto": the second call knows the first one's skippable fraction
        torch.cuda.synchronize()
        ev.clear()
        t0 = time.perf_counter()
        res = [frame(v) for v in range(8)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        marched = sum(int(o["num_samples"][0]) for _, o in res)
        evaluated = sum(int(o["num_evaluated"][0]) if "num_evaluated" in o else int(o["num_samples"][0]) for _, o in res)
        kms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        out[mode] = {"ms_per_frame": round(dt * 1e3, 3), "marched_msamples": round(marched / 8 / 1e6, 2),
                     "evaluated_msamples": round(evaluated / 8 / 1e6, 2), "field_kernel_ms": round(kms, 3),
                     "field_frac_of_hbm_peak": round(evaluated / 8 * BYTES_PER_SAMPLE / (kms / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                     "path_taken": "terminate" if "num_evaluated" in res[-1][1] else "two-kernel"}
        if mode == "fused_O":
            out[mode]["algorithmic_bytes_per_sample"] = BYTES_PER_SAMPLE // 2
            out[mode]["field_frac_of_hbm_peak"] = round(out[mode]["field_frac_of_hbm_peak"] / 2, 4)
        if mode in ("auto", "fused_O"):
            r0, o0 = res[0]
            f = torch.empty(H * W, 3, device=dev)
            f[r0["inds"][0]] = o0["image"][0]
            if mode == "auto":
                frame0 = f
                out["mean_opacity"] = round(float(o0["weights_sum"].mean()), 3)
            else:
                frame0_fast = f
    net.half_table = net.mlp_fp16 = False
    if frame0 is not None:
        # the metric's "PSNR" and "instance mIoU" against the scene's analytic ground truth (traced on the host):
        #   view 0 of the bench views = TRAINING view 0 at twice the training resolution (every fourth row and column);
        #   a HELD-OUT pose at 400x400, for the default path and for the -O numerics;
        #   instance stage on the frozen trained NeRF (K = 16 head, 400 steps of 4096 rays, masks with 10 % ignore
        #   labels): mIoU of the argmax of the rendered logits on the held-out pose
        psnr = lambda a, b: round(-10 * math.log10(max(float(((a - b) ** 2).mean()), 1e-20)), 2)
        pix = (np.arange(0, H, 4)[:, None] * W + np.arange(0, W, 4)[None, :]).reshape(-1)
        r = get_rays(pd[:1], intr, H, W, inds=torch.from_numpy(pix).to(dev))
        gt, _, _ = ds.room.trace(r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy())
        gt = torch.from_numpy(gt).to(dev)
        q = {"training_view_0_at_800": {"default": psnr(frame0[pix], gt), "pixels": int(pix.shape[0])}}
        if frame0_fast is not None:
            q["training_view_0_at_800"]["O_numerics"] = psnr(frame0_fast[pix], gt)
        held = torch.from_numpy(ds.room.look_at([0.3, -0.2, 0.1])[None]).to(dev)
        rh = get_rays(held, ds.intrinsics, ds.H, ds.W, patch=4)
        gt_h, ids_h, _ = ds.room.trace(rh["rays_o"][0].cpu().numpy(), rh["rays_d"][0].cpu().numpy())
        gt_h = torch.from_numpy(gt_h).to(dev)
        q["held_out_pose_at_400"] = {"pixels": int(gt_h.shape[0])}
        for name, flag in (("default", False), ("O_numerics", True)):
            net.half_table = net.mlp_fp16 = flag
            with torch.no_grad():
                q["held_out_pose_at_400"][name] = psnr(net.render(rh["rays_o"], rh["rays_d"], bg_color=1)["image"][0], gt_h)
        net.half_table = net.mlp_fp16 = False
        out["psnr_db_vs_ground_truth"] = q
        try:
            from instance_nerf_amd.nerf.utils import MIoUMeter
            net2 = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=K).to(dev)
            net2.load_state_dict(net.state_dict(), strict=False)          # the trained NeRF + its occupancy grid
            net2.mean_density, net2.iter_density, net2.mean_count = net.mean_density, net.iter_density, net.mean_count
            # the instance stage from the matched-mask files (matched/<img>.npy, 10 % of the pixels -1), product's loop
            ds2 = NeRFDataset(scene_dir, type="train", device=dev, scale=1.0, num_rays=4096, preload=True,
                              mask_dir=scene["mask_dir"], num_instances=K)
            n_inst = 1500                       # to convergence (round 3 stopped at 400 steps with the CE still at 0.3)
            ep2 = -(-n_inst // len(ds2))
            n_inst = ep2 * len(ds2)
            tr2 = Trainer("trained_inst", None, net2, stage="instance", device=dev, lr=1e-2, iters=n_inst,
                          update_extra_interval=10 ** 9, workspace=None, mute=True)
            tr2.global_step = 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(ep2):
                tr2.train_one_epoch(ds2.dataloader())
            torch.cuda.synchronize()
            inst_s = time.perf_counter() - t0
            ce = [tr2.stats["loss"][0], tr2.stats["loss"][-1]]            # mean CE of the first / the last epoch
            try:
                out["train_step"]["instance_stage"] = timed_trained_steps(tr2, net2, ds2, "instance")
            except Exception as e:                            # noqa: BLE001
                out["train_step"]["instance_stage"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            try:
                st2 = {"sd": {k: v.clone() for k, v in net2.state_dict().items()}, "mean_density": net2.mean_density,
                       "iter_density": net2.iter_density, "mean_count": net2.mean_count}
                loops["instance"] = train_loop_probe(dev, scene, "instance", st2, K=K)
            except Exception as e:                            # noqa: BLE001
                loops["instance"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            net2.eval()

            def score(pose):
                rr = get_rays(pose, ds.intrinsics, ds.H, ds.W, patch=4)
                _, ids, _ = ds.room.trace(rr["rays_o"][0].cpu().numpy(), rr["rays_d"][0].cpu().numpy())
                with torch.no_grad():
                    pred = net2.render(rr["rays_o"], rr["rays_d"], bg_color=1)["instance"][0].argmax(-1).cpu()
                truth = torch.from_numpy(ids % K)
                m = MIoUMeter(K)
                m.update(pred, truth)
                both = m.measure_both()
                # two definitions under distinct names (round-4 advisor): over the ids present in the ground truth, and
                # over every id with a non-empty union - the figure of rounds 1-3, where ids that only the prediction
                # contains (a few stray pixels) count as classes with IoU 0 (tools/miou_probe.py, r04_NOTES 4)
                return {"miou_gt_ids": round(both["miou_gt_ids"], 3), "miou_all_ids": round(both["miou_all_ids"], 3),
                        "pixel_accuracy": round(float((pred == truth).float().mean()), 4),
                        "ids_in_view": int((m.truth > 0).sum())}
            out["instance_miou_vs_ground_truth"] = {"training_view_0_at_400": score(ds.poses[:1]),
                                                    "held_out_pose_at_400": score(held), "classes": K, "steps": n_inst,
                                                    "ce_first_epoch": round(ce[0], 4), "ce_last_epoch": round(ce[-1], 4),
                                                    "train_seconds": round(inst_s, 2),
                                                    "train_steps_per_s": round(n_inst / inst_s, 1)}
        except Exception as e:                                # noqa: BLE001
            out["instance_miou_vs_ground_truth"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if frame0 is not None and frame0_fast is not None:      # the -O numerics against the default path, all pixels of view 0
        d = (frame0_fast - frame0).double()
        out["fused_O"]["vs_default_path"] = {"max_abs_diff": float(d.abs().max()),
                                             "psnr_db": round(-10 * math.log10(max(float((d ** 2).mean()), 1e-20)), 1)}
    if with_oracle:
        from oracle import c_port, hashgrid, rays as orays
        sd = net.state_dict()
        p = {"embeddings": sd["encoder.embeddings"], "sigma_w0": sd["sigma_net.0.weight"], "sigma_w1": sd["sigma_net.1.weight"],
             "color_w0": sd["color_net.0.weight"], "color_w1": sd["color_net.1.weight"], "color_w2": sd["color_net.2.weight"]}
        p = {k: v.detach().float().cpu() for k, v in p.items()}
        inds = np.sort(np.random.default_rng(11).permutation(H * W)[:4096])
        r = orays.get_rays(poses[:1], intr, H, W, inds=inds)
        ref = c_port.render(r["rays_o"][0], r["rays_d"][0], p, hashgrid.level_table(), net.density_bitfield.cpu().numpy(),
                            min_near=0.05)
        got = frame0.cpu().numpy()[inds].astype(np.float64)
        mse = float(np.mean((got - ref["image"].astype(np.float64)) ** 2))
        out["parity"] = {"against": "C oracle, trained weights + learned bitfield, 4096 random pixels of view 0 (auto mode)",
                         "max_abs_diff": float(np.abs(got - ref["image"]).max()),
                         "psnr_db": round(10.0 * np.log10(1.0 / mse), 1) if mse > 0 else None}
        if frame0_fast is not None:
            got = frame0_fast.cpu().numpy()[inds].astype(np.float64)
            mse = float(np.mean((got - ref["image"].astype(np.float64)) ** 2))
            out["fused_O"]["parity"] = {"against": "the same oracle pixels (fp32 oracle; -O numerics on the GPU)",
                                        "max_abs_diff": float(np.abs(got - ref["image"]).max()),
                                        "psnr_db": round(10.0 * np.log10(1.0 / mse), 1) if mse > 0 else None}
    shutil.rmtree(scene_dir, ignore_errors=True)
    return out
