#!/usr/bin/env python3
"""bench.py - Msamples/s of the render hot path on MI355X (contract: see DESIGN.md "Measurement").

One step = one full 800x800 render of the synthetic 3D-FRONT-like room (BASELINE.json
configs[1]: hash-grid NeRF L=16,F=2, sigma+colour, SURVEY.md section 8d scene and cameras):
ray generation -> ray/AABB -> occupancy march (count, scan, write) -> fused hash-gather + SH +
MLP (bf16x3-split MFMA, fp32 accumulate) -> alpha compositing, everything on the GPU with parameters, bitfield and
poses resident in HBM before the timed region.  Samples = live (occupied) samples the field
evaluated.  N > 1 ranks (torchrun, one process per GPU): every rank renders its own views -
no data-path collective - and value = all samples / max-over-ranks time ("weak" scaling).

Rank 0 prints TWO JSON lines: the full record (every probe's object; also written to
gpurun_out/bench_full_n<N>.json) and, LAST, the compact contract line (tools/bench_line.py: the contract's keys, the
`roofline` and `cpu_baseline` objects and one number per secondary leg; never more than 6 KB - round 5's single 20 KB
line could not be parsed by the driver).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0           # what a float4 copy kernel reaches on this part (same guide: 79 % of the spec peak)
BYTES_PER_SAMPLE = 1024         # 16 levels x 8 corners x 2 features x 4 B
MFMA_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak (same guide; AMD's 5 PF headline includes 2:1 sparsity)
# useful MLP flops per sample: sigma net 32 -> 64 -> 16, colour net 31 -> 64 -> 64 -> 3 (2 flops per multiply-add); the
# kernel executes each product three times (bf16 hi/lo split: hi.hi + hi.lo + lo.hi) on v_mfma_f32_16x16x32_bf16
MLP_FLOPS_PER_SAMPLE = 2 * (32 * 64 + 64 * 16 + 31 * 64 + 64 * 64 + 64 * 3)
MFMA_PASSES = 3


TRAFFIC_JSON = os.path.join("profiles", "r06_traffic.json")
MFMA_JSON = os.path.join("profiles", "r06_mfma.json")


def measured_mfma_busy():
    """MFMA pipe utilisation of the fused field kernel (and of the two head-backward kernels of the training steps) from
    the committed PMC profile (profiles/r06_mfma.json, tools/mfma_json.py over the `mfma` pass of tools/pmc_bench.sh /
    tools/pmc_train.sh): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 256 CUs x 4 SIMDs) - the counter ticks in
    cycles per SIMD (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units").  Quoted only for the kernel sources it was
    measured on.  -> dict or None."""
    from instance_nerf_amd import build
    path = os.path.join(ROOT, MFMA_JSON)
    if not os.path.exists(path):
        return None
    t = json.load(open(path))
    return t if t.get("source_sha") == build.source_sha("field") else None


def measured_traffic_bytes_per_sample(res):
    """HBM-side bytes per sample of the fused field kernel from the committed PMC profile
    (profiles/r06_traffic.json: FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, gfx950 x2
    read correction; written by tools/traffic_json.py from a tools/pmc_bench.sh run).  PMC counters cannot be
    collected from inside this process, so the figure is only quoted for the kernel sources it was measured on (the
    file carries their sha): None if the profile is absent, belongs to other sources, or the workload differs."""
    from instance_nerf_amd import build
    path = os.path.join(ROOT, TRAFFIC_JSON)
    if res != 800 or not os.path.exists(path):
        return None
    t = json.load(open(path))
    if t.get("source_sha") != build.source_sha("field"):
        return None
    return float(t["read_bytes_per_sample"]) + float(t["write_bytes_per_sample"])


sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_legs import Legs  # noqa: E402
from bench_line import emit  # noqa: E402
from bench_secondary import (bound_render_probe, build_network, collective_record, config5_probe,  # noqa: E402
                             half_table_probe, instance_render_probe, render_sharded_probe, train_probe,
                             trained_scene_probe)


def cpu_baseline(room, net, frame0=None, chunk=16384, budget_s=12.0, max_chunks=40, render_view0=None):
    """The C restatement of the whole path (oracle/c/inr_oracle.c through oracle/c_port.py, kind 'port': slab test ->
    occupancy march -> hash-grid gather + SH + MLPs -> compositing, one ray at a time, OpenMP over rays on every
    host core) on a bounded sample of the same workload: chunks of random pixels of view 0 until ~budget_s seconds
    of wall time have been used.  Same table, weights (copied from the GPU network), bitfield and camera as the GPU
    measurement, so the pixels it renders double as the parity check of the line: `frame0` (the GPU's view 0, row-major
    [H, W, 3]) is compared with the oracle's colours at the sampled pixels -> second return value."""
    from oracle import c_port, hashgrid, rays as orays
    table = hashgrid.level_table()
    sd = net.state_dict()
    p = {"embeddings": sd["encoder.embeddings"], "sigma_w0": sd["sigma_net.0.weight"], "sigma_w1": sd["sigma_net.1.weight"],
         "color_w0": sd["color_net.0.weight"], "color_w1": sd["color_net.1.weight"], "color_w2": sd["color_net.2.weight"]}
    p = {k: v.detach().float().cpu() for k, v in p.items()}
    bits = room.density_bitfield(128, 1.0)
    poses, intr, H, W = room.cameras()
    perm = np.random.default_rng(7).permutation(H * W)
    threads = c_port.num_threads()
    c_port.render(*[orays.get_rays(poses[:1], intr, H, W, inds=perm[:256])[k][0] for k in ("rays_o", "rays_d")],
                  p, table, bits, min_near=0.05)                               # page the table in, untimed
    total, rays_done, t_used, n = 0, 0, 0.0, 0
    seen, colours = [], []
    while t_used < budget_s and n < max_chunks:
        inds = np.sort(perm[n * chunk:(n + 1) * chunk])
        r = orays.get_rays(poses[:1], intr, H, W, inds=inds)
        t0 = time.perf_counter()
        out = c_port.render(r["rays_o"][0], r["rays_d"][0], p, table, bits, min_near=0.05)
        t_used += time.perf_counter() - t0
        total += out["total"]
        rays_done += len(inds)
        seen.append(inds)
        colours.append(out["image"])
        n += 1
    base = {"value": round(total / t_used / 1e6, 5), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{rays_done} random rays of view 0 (800x800 camera) in {n} chunks of {chunk}, {total} samples, "
                      f"C oracle (gcc -O2, OpenMP x{threads}) march+field+composite, {t_used:.1f} s"}
    parity = None
    if frame0 is not None:
        torch.cuda.synchronize()
        ref = np.concatenate(colours).astype(np.float64)
        got = frame0.reshape(-1, 3).cpu().numpy()[np.concatenate(seen)].astype(np.float64)
        mse = float(np.mean((got - ref) ** 2))
        parity = {"against": "C oracle, same weights / bitfield / camera, view 0", "pixels": int(ref.shape[0]),
                  "max_abs_diff": float(np.abs(got - ref).max()),
                  "psnr_db": round(10.0 * np.log10(1.0 / mse), 1) if mse > 0 else None,
                  "tolerance": "tests: 1e-4 per channel (fp32 path, bf16x3-split MLP GEMMs)"}
        if render_view0 is not None:
            # The throughput network is upstream's initialisation (table U(-1e-4, 1e-4)): sigma ~ 1 and rgb ~ 0.5
            # everywhere, so the comparison above says little.  Once more with an O(1) table - U(-1, 1), every level
            # matters, rays become semi-transparent - outside any timed region: 16384 random pixels of view 0.
            with torch.no_grad():
                net.encoder.embeddings.uniform_(-1.0, 1.0, generator=torch.Generator(device=net.encoder.embeddings.device).manual_seed(5))
            got = render_view0().reshape(-1, 3).cpu().numpy()
            p["embeddings"] = net.encoder.embeddings.detach().float().cpu()
            inds = np.sort(perm[:16384])
            r = orays.get_rays(poses[:1], intr, H, W, inds=inds)
            ref = c_port.render(r["rays_o"][0], r["rays_d"][0], p, table, bits, min_near=0.05)
            d = got[inds].astype(np.float64) - ref["image"].astype(np.float64)
            parity["o1_table"] = {"what": "the same network with the table redrawn from U(-1,1) (O(1) densities and colours, "
                                          "semi-transparent rays), 16384 random pixels of view 0, untimed",
                                  "max_abs_diff": float(np.abs(d).max()),
                                  "psnr_db": round(10.0 * np.log10(1.0 / float(np.mean(d ** 2))), 1) if np.any(d) else None,
                                  "mean_opacity": round(float(ref["weights_sum"].mean()), 3)}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-probe", action="store_true")
    ap.add_argument("--train-steps", type=int, default=100,
                    help="timed steps of the secondary training probes (round-5 verdict item 4: >= 100; two regions each)")
    ap.add_argument("--no-trained-scene", action="store_true", help="skip the trained-scene leg (~25 s)")
    ap.add_argument("--pipeline-probe", action="store_true",
                    help="(default since late round 3; kept for old command lines) measure the same frames through "
                         "the other view loop AFTER the headline's timed region")
    ap.add_argument("--no-pipeline-probe", action="store_true", help="skip the other view loop")
    ap.add_argument("--diagnostics", action="store_true",
                    help="record the GPU's clocks during the timed region and the workgroup-to-XCD map (\"clocks\")")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="1 (default since round 4): the headline loop is Trainer.render_sequence as Trainer.test runs it "
                         "- views through FramePipeline, two alternating streams; 0: one view at a time on one stream")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) as CHILD processes -
        # nothing in this process has touched the GPU yet, and it is never replaced by exec - and pass their
        # exit code on.  The driver's own `python -m torch.distributed.run ... bench.py --gpus N` skips this.
        import subprocess
        port = os.environ.get("MASTER_PORT", str(29500 + os.getpid() % 2000))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    t_start = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(json.dumps({"error": f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                                   f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus})"}))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU: bench.py measures the HIP path only (no CPU fallback)"}))
        sys.exit(1)
    # one process per GPU; the modulo only matters for the single-GPU dry run of the N>1 logic
    # (INR_DIST_BACKEND=gloo, several ranks sharing cuda:0)
    local_dev = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    backend = os.environ.get("INR_DIST_BACKEND", "nccl")          # "nccl" == RCCL on ROCm
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a collective that a dead rank never joins raises after 5 minutes instead of blocking for the default 10-30
        tmo = datetime.timedelta(seconds=300)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    from instance_nerf_amd.nerf.utils import get_rays
    net, room = build_network(dev)
    poses, intr, H, W = room.cameras(H=args.res, W=args.res, focal=args.res / 2.0)
    poses_d = torch.from_numpy(poses).to(dev)

    # The wrapper carries the timing events around the dominant kernel, on the stream it is launched on.
    ev_pairs = []

    def timed(fn):
        def wrapper(x, *rest, **kw):
            st = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            out = fn(x, *rest, **kw)
            e1.record(st)
            ev_pairs.append((e0, e1, x.shape[0]))
            return out
        return wrapper
    net.forward = timed(net.forward)
    net.forward_table = timed(net.forward_table)      # the entry the fused frame path uses

    # The headline loop is the PRODUCT's view loop: Trainer.render_sequence, what Trainer.test / evaluate_one_epoch
    # iterate over (round-3 verdict: the fastest multi-view path must be the product path and the headline).  The
    # loader hands over the H*W rays of a view in row-major pixel order, as upstream's loaders do; the trainer renders
    # them 4x4-patch by patch and returns every per-ray output in the caller's order.  With --pipeline 1 the views
    # alternate on the two streams of a FramePipeline (march of view i+1 and compositing of view i-1 under the field
    # kernel of view i); --pipeline 0 is upstream's one-view-at-a-time loop, reported beside it as "one_stream".
    from instance_nerf_amd.nerf.utils import Trainer
    viewer = Trainer("bench_views", None, net, stage="nerf", device=dev, workspace=None, use_checkpoint="scratch", mute=True)
    viewer.opt = argparse.Namespace(dt_gamma=0, max_steps=1024, T_thresh=1e-4)
    net.eval()

    def views(first, n):
        for i in range(first, first + n):
            view = (i + rank) % poses_d.shape[0]   # every rank cycles through all views (their sample counts differ by +-25 %)
            r = get_rays(poses_d[view:view + 1], intr, H, W)
            yield {"rays_o": r["rays_o"], "rays_d": r["rays_d"], "H": H, "W": W}

    def render_views(first, n, pipeline):
        """-> (sample counters of the n views, the last view's frame [H,W,3])"""
        cs, frame = [], None
        for _, out in viewer.render_sequence(views(first, n), pipeline=pipeline):
            # samples the field actually evaluated (the early-terminating mode may evaluate fewer than were marched)
            cs.append(out["num_evaluated"] if "num_evaluated" in out else out["num_samples"])
            frame = out["image"]
        return cs, frame.view(H, W, 3)

    def step(i):                                   # one view, one stream (CPU baseline / parity leg)
        cs, frame = render_views(i, 1, False)
        return {"frame": frame, "num_samples": cs[0]}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import gc
    # a box that has just run other GPU processes (the driver's test suite with its 8-rank children) is still tearing
    # them down for a second or two; the pipelined loop hands a frame to the device every 5 ms and shows a busy host
    # first (one of three runs right behind the suite read 5.87 instead of 6.3 Gsamples/s, the one-stream leg later in
    # the same process was normal).  Untimed, before the W warm-up steps.
    time.sleep(float(os.environ.get("INR_BENCH_SETTLE_S", "2")))
    render_views(0, args.warmup, bool(args.pipeline))
    gc.collect()
    gc.disable()               # no collector pause inside the timed region (a frame is 6 ms, a gen-2 pass ~10 ms)
    sclk = None
    if args.diagnostics and rank == 0:         # opt-in: a sampler thread beside the timed region (tools/bench_diagnostics.py)
        from bench_diagnostics import SclkSampler, xcd_map_probe
        sclk = SclkSampler(dev)
    barrier()
    ev_pairs.clear()
    counters = []
    if sclk is not None:
        sclk.start()
    t0 = time.perf_counter()
    counters, _ = render_views(args.warmup, args.steps, bool(args.pipeline))
    barrier()
    elapsed = time.perf_counter() - t0
    clocks = sclk.stop() if sclk is not None else None
    gc.enable()
    if clocks is not None:
        clocks["xcd_map"] = xcd_map_probe()

    n_samples = int(sum(int(c[0]) for c in counters))
    kernel_ms = sum(a.elapsed_time(b) for a, b, _ in ev_pairs)
    n_launch = len(ev_pairs)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([float(n_samples)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed_all, samples_all = float(tmax.item()), float(tot.item())

    line = None
    if rank == 0:
        avg_kernel_s = kernel_ms / 1e3 / max(n_launch, 1)
        per_launch = n_samples / max(n_launch, 1)
        achieved = per_launch * BYTES_PER_SAMPLE / avg_kernel_s / 1e9
        tbs = measured_traffic_bytes_per_sample(args.res)
        traffic = None if tbs is None else round(per_launch * tbs / avg_kernel_s / 1e9, 1)
        line = {
            "metric": "Msamples/sec (train+infer) 3D-FRONT 800x800 at 1/2/4/8 MI355X; PSNR parity",
            "value": round(samples_all / elapsed_all / 1e6, 3),
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_all / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # gathers, interpolation, SH, marching and compositing are IEEE fp32; the MLP GEMMs run on
            # v_mfma_f32_16x16x32_bf16 with a 3-term bf16 split of both operands (~2^-16 relative, fp32 accumulate)
            "dtype_note": "MLP GEMMs: bf16x3-split MFMA, fp32 accumulate; everything else IEEE fp32",
            "config": {"workload": f"render {args.res}x{args.res} synthetic 3D-FRONT-like room, hash-grid NeRF "
                                   "L=16 F=2 T=6119864 sigma+rgb (BASELINE configs[1]), one view per step per GPU",
                       "samples_per_step": n_samples // args.steps, "rays_per_step": H * W,
                       "parallelism": f"views sharded over {world} GPU(s), no data-path collective"},
            # ONE definition since round 6 (round-5 advisor): `frac` / `achieved` / `avg_launch_ms` are the dominant
            # kernel's launches INSIDE THE TIMED REGION - the loop `value` and `ms_per_step` come from (algorithmic bytes
            # per launch over the mean launch time, events on the launch stream; the rocprofv3 kernel trace of the same
            # command under profiles/ must agree).  With --pipeline 1 the kernel shares the CUs there with the next view's
            # marchers; the same kernel with the chip to itself (the one-stream loop of this process) is
            # `frac_kernel_alone`.  Rounds 1-3 and 5 printed the alone figure as `frac`, round 4 this one.
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "in_timed_region_frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         # FETCH_SIZE / WRITE_SIZE count the L2's memory-side (fabric) requests: reads the 256 MB Infinity
                         # Cache serves are included, so this is fabric traffic, an upper bound on DRAM traffic
                         "traffic_kind": "fabric (L2 misses; Infinity-Cache hits included)",
                         "traffic_bytes_per_sample": tbs,
                         "traffic_source": (TRAFFIC_JSON + " (rocprofv3 PMC on these kernel sources, GB/s at this run's "
                                            "launch time)") if traffic is not None else
                         "no PMC profile of these kernel sources under profiles/ (tools/pmc_bench.sh + tools/traffic_json.py)",
                         # second denominator: what a streaming copy reaches on this part (the algorithmic figure can
                         # exceed it only because part of the rows come from the L2 / Infinity Cache)
                         "achievable_copy_peak": HBM_COPY_GBS,
                         "frac_of_achievable_copy": round(achieved / HBM_COPY_GBS, 4),
                         "traffic_frac_of_achievable_copy": None if traffic is None else round(traffic / HBM_COPY_GBS, 4),
                         "kernel": "k_nerf_fwd<true,true> (fused hash gather + SH table + MLP)",
                         "avg_launch_ms": round(avg_kernel_s * 1e3, 4), "launches": n_launch,
                         "algorithmic_bytes_per_sample": BYTES_PER_SAMPLE},
            "clocks": clocks,
        }
        # north_star's second evidence item: MFMA utilisation.  Effective rate from this run's launches, pipe busy
        # fraction from the committed PMC pass over these kernel sources.
        tfl = per_launch * MLP_FLOPS_PER_SAMPLE * MFMA_PASSES / avg_kernel_s / 1e12
        mb = measured_mfma_busy()
        line["roofline"]["mfma"] = {
            "tflops": round(tfl, 1), "peak": MFMA_PEAK_TFLOPS, "frac": round(tfl / MFMA_PEAK_TFLOPS, 4),
            "flops_per_sample": MLP_FLOPS_PER_SAMPLE, "passes": MFMA_PASSES,
            "busy": None if mb is None else mb["k_nerf_fwd"]["mfma_busy"],
            "head_bwd_busy": None if mb is None else {k: v["mfma_busy"] for k, v in mb.items()
                                                      if isinstance(v, dict) and "head_bwd" in k},
            "what": "tflops = samples x 18 688 useful MLP flops x 3 bf16-split passes / launch time, against the dense bf16 "
                    "MFMA peak; busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 256 CUs x 4 SIMDs)",
            "source": (MFMA_JSON if mb is not None else "no PMC profile of these kernel sources (tools/mfma_json.py)")}
        # whole-frame fraction of the roofline: every launch of the loop, not only the field kernel
        line["end_to_end"] = {"achieved": round(samples_all / world / elapsed_all * BYTES_PER_SAMPLE / 1e9, 1),
                              "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(samples_all / world / elapsed_all * BYTES_PER_SAMPLE / 1e9 / HBM_PEAK_GBS, 4),
                              "what": "samples x 1024 B over the wall clock of the timed loop, per GPU"}
        line["loop"] = ("Trainer.render_sequence, FramePipeline (two alternating streams; field kernels serialised)"
                        if args.pipeline else "Trainer.render_sequence, one view at a time on one stream")
        if not args.no_pipeline_probe:
            # the same frames through the OTHER loop, after the headline's timed region
            try:
                other = not args.pipeline
                n_o = min(args.steps, 60)
                render_views(0, 4, other)
                torch.cuda.synchronize()
                ev_pairs.clear()
                t0 = time.perf_counter()
                cs, _ = render_views(4, n_o, other)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                ns = int(sum(int(c[0]) for c in cs))
                kms = sum(a.elapsed_time(b) for a, b, _ in ev_pairs) / max(len(ev_pairs), 1)
                line["pipelined" if other else "one_stream"] = {
                    "value": round(ns / el / 1e6, 3), "unit": "Msamples/s", "steps": n_o,
                    "ms_per_step": round(el / n_o * 1e3, 3), "field_kernel_ms": round(kms, 4),
                    "field_frac_of_hbm_peak": round(ns / n_o * BYTES_PER_SAMPLE / (kms / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                    "what": ("Trainer.render_sequence(pipeline=True): views alternate on two streams, field kernels "
                             "serialised; march of view i+1 and compositing of view i-1 run under field kernel i") if other
                    else "Trainer.render_sequence(pipeline=False): upstream's loop, one view at a time on one stream"}
                if not other:
                    alone = ns / n_o * BYTES_PER_SAMPLE / (kms / 1e3) / 1e9
                    line["roofline"].update({"frac_kernel_alone": round(alone / HBM_PEAK_GBS, 4),
                                             "achieved_alone": round(alone, 1), "avg_launch_ms_alone": round(kms, 4)})
            except Exception as e:                            # noqa: BLE001
                line["pipelined" if not args.pipeline else "one_stream"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"], line["parity"] = cpu_baseline(room, net, step(0)["frame"],
                                                                    render_view0=lambda: step(0)["frame"])
            except Exception as e:                            # noqa: BLE001
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    del net
    if world > 1:
        secondary_multi(args, line, dev, rank, world, backend, red_dev, t_start)
    elif not args.no_train_probe:
        secondary_single(args, line, dev, red_dev)
    if rank == 0:
        line["bench_wall_s"] = round(time.perf_counter() - t_start, 1)
        emit(line, world, ROOT)
    if world > 1:
        dist.barrier()             # the ranks leave together (rank 0 alone ran the single-GPU legs after the probes)
        dist.destroy_process_group()


def _try(line, key, fn):
    """One secondary leg: a failure costs its own key, never the line."""
    try:
        line[key] = fn()
    except Exception as e:                                        # noqa: BLE001
        line[key] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return line[key]


TRAIN_KEEP = ("ms_per_step", "ms_per_step_median", "ms_per_step_running", "ms_per_step_of_both_timed_regions",
              "host_enqueue_ms_per_step", "samples_per_step", "msamples_per_s", "occupancy_updates_in_timed_steps",
              "mode", "graphs_captured", "loss_first", "loss_last", "timed_steps")


def secondary_single(args, line, dev, red_dev):
    """world == 1: every other measurement of the record, each in its own try-block."""
    def train_legs(stage, key):
        ts = train_probe(dev, 0, 1, red_dev, steps=args.train_steps, stage=stage)
        # the same loop as a captured two-stream pipeline (Trainer(use_graph=True, look_ahead=True)): the figure the compact
        # line quotes (round-5 verdict item 4) - the eager loop's MEAN moves with the host's enqueue time
        try:
            ov = train_probe(dev, 0, 1, red_dev, steps=args.train_steps, stage=stage, mode="pipelined")
            ts["overlapped"] = {k: ov[k] for k in TRAIN_KEEP if k in ov}
            ts["overlapped"]["step_frac_of_hbm_peak"] = ov["roofline"]["step"]["frac"]
        except Exception as e:                                    # noqa: BLE001
            ts["overlapped"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if stage == "instance":
            # upstream's -O on the instance stage (Trainer(fp16=True)): trained parameters stay fp32, the frozen
            # NeRF's forward gathers from the fp16 table copy and runs the one-pass fp16 MLP
            try:
                ov = train_probe(dev, 0, 1, red_dev, steps=args.train_steps, mode="pipelined", fp16=True)
                ts["overlapped_O"] = {k: ov[k] for k in TRAIN_KEEP if k in ov}
            except Exception as e:                                # noqa: BLE001
                ts["overlapped_O"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        return ts
    _try(line, "train_step", lambda: train_legs("instance", "train_step"))
    _try(line, "train_step_nerf", lambda: train_legs("nerf", "train_step_nerf"))
    single_gpu_legs(args, line, dev, red_dev, world=1)


def single_gpu_legs(args, line, dev, red_dev, world):
    """Rank 0 only, no collectives: configs[4] (world 1), the instance / -O renders, the configurations off the tuned one,
    the trained scene."""
    if world == 1:
        _try(line, "extract_roialign", lambda: config5_probe(dev))
    _try(line, "render_instance", lambda: instance_render_probe(dev))
    _try(line, "render_half_table", lambda: half_table_probe(dev))
    _try(line, "render_fast", lambda: half_table_probe(dev, mlp_fp16=True))
    # off the tuned configuration (round-4 verdict item 1): bound 2 and 4 (cascades, finer level tables), the
    # render with growing and with constant steps, and both training stages at bound 4
    for key, kw in (("render_bound2", dict(bound=2, dt_gamma=1.0 / 128)), ("render_bound4", dict(bound=4, dt_gamma=1.0 / 128)),
                    ("render_bound4_constant_steps", dict(bound=4, dt_gamma=0.0))):
        _try(line, key, lambda kw=kw: bound_render_probe(dev, **kw))
    if world == 1:
        drop = ("ms_of_each_step", "graphs_captured", "allreduce_mb_per_step", "gradient_schedule", "n_gpus")
        for key, st in (("train_step_bound4", "instance"), ("train_step_nerf_bound4", "nerf")):
            _try(line, key, lambda st=st: {k: v for k, v in train_probe(
                dev, 0, 1, red_dev, steps=args.train_steps, stage=st, bound=4, dt_gamma=1.0 / 128).items() if k not in drop})
        if not args.no_trained_scene:
            _try(line, "trained_scene", lambda: trained_scene_probe(dev, with_oracle=not args.no_cpu_baseline))


def secondary_multi(args, line, dev, rank, world, backend, red_dev, t_start):
    """world > 1.  Everything here contains collectives: a rank that dies or hangs in one leaves the others waiting.  Guards:
    (1) the headline is put in a side file first; (2) a watchdog THREAD on rank 0 (a blocked collective holds the main
    thread inside C++, where no signal handler runs) prints the record as far as it got - full line, then the compact
    line - and exits non-zero if the legs have not finished after 7 minutes; (3) every leg runs in its own try-block
    and the ranks agree on its outcome BEFORE the next leg (all_reduce MIN of an ok flag): a leg that failed on any rank
    poisons only itself, and if a rank is left in a state where it cannot take part in collectives any more the remaining
    collective legs are skipped on ALL ranks (round-5 advisor: a one-rank exception used to leave the others blocked
    until the RCCL timeout).  Order (round-5 verdict item 8): collective record, training steps, sharded render,
    configs[4]; then rank 0's single-GPU legs."""
    watchdog = None
    if rank == 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"bench_headline_n{world}.json"), "w") as f:
            f.write(json.dumps(line) + "\n")
        import threading

        def bail():
            line.setdefault("train_step", {"error": "collectives / training probes did not finish in 420 s"})
            line["error"] = "watchdog: a collective leg did not finish in 420 s"
            line["bench_wall_s"] = round(time.perf_counter() - t_start, 1)
            emit(line, world, ROOT)
            # a hung collective is a FAILED run: non-zero, so that the launcher tears the other ranks down and the
            # caller sees it (the headline is also in gpurun_out/bench_headline_n<N>.json)
            os._exit(3)
        watchdog = threading.Timer(420.0, bail)
        watchdog.daemon = True
        watchdog.start()

    legs = Legs(rank, world, red_dev, line)
    leg = legs.run

    try:
        rec = leg("collective", lambda: collective_record(dev, rank, world, backend, red_dev))
        if backend == "nccl" and rec.get("all_ranks_on_distinct_gpus") is False:
            # an RCCL run whose ranks share GPUs measures nothing about xGMI: fail loudly instead of reporting a curve
            if rank == 0:
                line["error"] = f"{rec['distinct_devices']} distinct GPUs for {world} ranks"
                emit(line, world, ROOT)
            if watchdog is not None:
                watchdog.cancel()
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(4)
        if not args.no_train_probe:
            from instance_nerf_amd.nerf.utils import grad_sync as _gs
            default_schedule = _gs.schedule
            leg("train_step", lambda: train_probe(dev, rank, world, red_dev, steps=args.train_steps))
            leg("train_step_nerf", lambda: train_probe(dev, rank, world, red_dev, steps=args.train_steps, stage="nerf"))
            # the other gradient schedule, back to back (same bytes on the links, 7/8 of the
            # optimiser sweep saved per rank): the xGMI box decides which one becomes the default
            other = "reduce_scatter" if default_schedule != "reduce_scatter" else "all_reduce"
            leg("train_step_other_schedule",
                lambda: train_probe(dev, rank, world, red_dev, steps=args.train_steps, schedule=other),
                keep=("ms_per_step", "ms_per_step_median", "ms_per_step_of_both_timed_regions", "samples_per_step",
                      "msamples_per_s", "gradient_schedule", "allreduce_mb_per_step", "loss_first", "loss_last"))
            _gs.schedule = default_schedule
            # strong scaling beside the headline's weak scaling: one frame over all ranks
            leg("render_sharded", lambda: render_sharded_probe(dev, rank, world, red_dev, res=args.res))
            # BASELINE configs[4] as the metric states it: N scenes, one per GPU, no collective on the data path -
            # every rank extracts and pools its own scene; aggregate extraction rate, RoIAlign time max over ranks

            def config5_all():
                try:
                    mine = config5_probe(dev)
                except Exception as e:                            # noqa: BLE001
                    mine = {"error": f"{type(e).__name__}: {e}"[:300]}
                every = [None] * world
                dist.all_gather_object(every, mine)
                ok = [o for o in every if "error" not in o]
                if len(ok) < world:
                    return {"error": f"{world - len(ok)} of {world} ranks failed: " + str([o["error"] for o in every if "error" in o][:1])}
                return {"workload": ok[0]["workload"] + f"; {world} scenes, one per GPU (replicas, no collective)",
                        "n_gpus": world,
                        "extract_mvoxels_per_s": round(sum(o["extract_mvoxels_per_s"] for o in ok), 1),
                        "extract_ms_max_over_ranks": max(o["extract_ms"] for o in ok),
                        "roi_align_forward_ms_max_over_ranks": max(o["roi_align_forward_ms"] for o in ok),
                        "roi_align_backward_ms_max_over_ranks": max(o["roi_align_backward_ms"] for o in ok),
                        "per_rank": [{k: o[k] for k in ("extract_ms", "extract_mvoxels_per_s", "roi_align_forward_ms",
                                                        "roi_align_backward_ms")} for o in ok]}
            leg("extract_roialign", config5_all)
    finally:
        if watchdog is not None:
            watchdog.cancel()
    if rank == 0:
        line["collective_legs_wall_s"] = round(time.perf_counter() - t_start, 1)
        if not args.no_train_probe:
            single_gpu_legs(args, line, dev, red_dev, world)


if __name__ == "__main__":
    main()
