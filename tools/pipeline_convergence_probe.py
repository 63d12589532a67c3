"""Does the captured two-stream training pipeline (Trainer(use_graph=True, look_ahead=True), the opt-in fast mode) train
to the same place as the eager loop?  The parity tests compare a few dozen steps bit for bit; this probe runs both
stages of the synthetic room TO CONVERGENCE in both modes - occupancy updates every 16 steps inside the run, learning-rate
decay, parameter EMA - and scores each against the scene's analytic ground truth on a held-out pose.

  NeRF stage      1500 steps, 4096 rays of 24 views at 400x400      -> PSNR (dB) on the held-out pose
  instance stage  1500 steps on that frozen NeRF, K = 16 head,        -> mIoU on the held-out pose (classes in the truth)
                  10 % ignore labels; pipeline with the shaded head

Same seeds and batches in both modes (the batch list is drawn once).  python tools/pipeline_convergence_probe.py [steps [out.json]]"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instance_nerf_amd.nerf import NeRFNetwork                                 # noqa: E402
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset              # noqa: E402
from instance_nerf_amd.nerf.utils import MIoUMeter, Trainer, get_rays         # noqa: E402

dev = torch.device("cuda", 0)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
OUT = sys.argv[2] if len(sys.argv) > 2 else None           # the json goes here (stdout also carries the Trainer's log)
K = 16


def run(stage, pipelined, batches, init):
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10,
                      num_instances=K if stage == "instance" else 0).to(dev)
    if init is not None:
        net.load_state_dict(init["state"], strict=False)
        net.mean_density, net.iter_density, net.mean_count = init["mean_density"], init["iter_density"], init["mean_count"]
    kw = dict(update_extra_interval=10 ** 9) if stage == "instance" else {}
    tr = Trainer(f"conv_{stage}_{int(pipelined)}", None, net, stage=stage, device=dev, lr=1e-2, iters=STEPS,
                 use_graph=pipelined, look_ahead=pipelined, **kw)
    if stage == "instance":
        tr.global_step = 1
    captures = [0]
    if pipelined:
        inner = tr._pipe_capture

        def counted(*a, **k):
            captures[0] += 1
            return inner(*a, **k)
        tr._pipe_capture = counted
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    first = last = None
    tail = STEPS - STEPS // 3                          # the last third alone: buffer sizes have settled, no capture left
    n_dev = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(STEPS):
        if i == tail:
            torch.cuda.synchronize()
            t1, cap1 = time.perf_counter(), captures[0]
        nxt = batches[i + 1] if pipelined and i + 1 < STEPS else None
        loss = tr.train_one_step(batches[i], nxt)
        if i == 0:
            first = float(loss)
        if i >= tail:
            n_dev += net.last_counter[0]
    last = float(loss)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    net.eval()
    return net, tr, {"seconds": round(t2 - t0, 2), "ms_per_step": round((t2 - t0) / STEPS * 1e3, 3),
                     "ms_per_step_last_third": round((t2 - t1) / (STEPS - tail) * 1e3, 3),
                     "samples_per_step_last_third": int(n_dev) // (STEPS - tail),
                     "graph_captures": captures[0], "graph_captures_last_third": captures[0] - cap1,
                     "loss_first": round(first, 4), "loss_last": round(last, 5),
                     "pipeline_used": bool(getattr(tr, "_pipe", None))}


def main():
    out = {"steps_per_stage": STEPS}
    ds = SyntheticRoomDataset(dev, H=400, W=400, n_views=24, num_rays=4096)
    nerf_batches = [ds.batch() for _ in range(STEPS)]
    held = torch.from_numpy(ds.room.look_at([0.3, -0.2, 0.1])[None]).to(dev)
    rh = get_rays(held, ds.intrinsics, ds.H, ds.W, patch=4)
    gt, ids, _ = ds.room.trace(rh["rays_o"][0].cpu().numpy(), rh["rays_d"][0].cpu().numpy())
    gt = torch.from_numpy(gt).to(dev)
    truth = torch.from_numpy(ids % K)

    nets = {}
    for pipelined in (False, True):
        net, tr, rec = run("nerf", pipelined, nerf_batches, None)
        with torch.no_grad():
            img = net.render(rh["rays_o"], rh["rays_d"], bg_color=1)["image"][0]
        rec["psnr_db_held_out"] = round(-10 * math.log10(max(float(((img - gt) ** 2).mean()), 1e-20)), 3)
        rec["occupied_cells"] = round(float((net.density_grid > min(net.mean_density, net.density_thresh)).float().mean()), 4)
        out["nerf_" + ("pipelined" if pipelined else "eager")] = rec
        nets[pipelined] = net
        del tr

    # instance stage: BOTH modes start from the EAGER run's NeRF, so the difference is the instance loop's alone
    base = nets[False]
    init = {"state": {k: v.clone() for k, v in base.state_dict().items()}, "mean_density": base.mean_density,
            "iter_density": base.iter_density, "mean_count": base.mean_count}
    ds2 = SyntheticRoomDataset(dev, H=400, W=400, n_views=24, num_rays=4096, num_instances=K, ignore_frac=0.1)
    inst_batches = [ds2.batch() for _ in range(STEPS)]
    for pipelined in (False, True):
        net, tr, rec = run("instance", pipelined, inst_batches, init)
        with torch.no_grad():
            pred = net.render(rh["rays_o"], rh["rays_d"], bg_color=1)["instance"][0].argmax(-1).cpu()
        m = MIoUMeter(K)
        m.update(pred, truth)
        rec["miou_held_out"] = round(float(m.measure()), 4)
        rec["pixel_accuracy"] = round(float((pred == truth).float().mean()), 4)
        out["instance_" + ("pipelined" if pipelined else "eager")] = rec
        del tr
    out["psnr_gap_db"] = round(out["nerf_pipelined"]["psnr_db_held_out"] - out["nerf_eager"]["psnr_db_held_out"], 3)
    out["miou_gap"] = round(out["instance_pipelined"]["miou_held_out"] - out["instance_eager"]["miou_held_out"], 4)
    print(json.dumps(out, indent=1))
    if OUT:
        with open(OUT, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
