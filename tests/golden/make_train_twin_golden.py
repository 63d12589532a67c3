"""Converged-quality parity artefact (round-3 verdict, item 5): tests/golden/train_twin.npz.

The torch ORACLE trains both stages at BASELINE configs[0] size - 64x64 views of the synthetic room (8 training poses
+ 1 held-out pose, analytic ground truth from RoomScene.trace), 1024 rays per step, fixed batches, the analytic
occupancy bitfield, no jitter - first the NeRF (MSE on rgb, N_NERF steps), then the K = 16 instance field on the frozen
NeRF (cross entropy vs the ground-truth ids mod K, every 10th ray ignored, N_INST steps), with torch.optim.Adam and the
Trainer's learning-rate rule.  It then renders the held-out view and records its PSNR against the ground truth and the
mIoU of its arg-max ids.  tests/test_train_twin.py trains the HIP path (Trainer) on the SAME batches from the SAME
initial parameters and must land within 0.1 dB / 0.01 mIoU of these numbers: the metric string's "PSNR parity" and
"instance mIoU within tolerance" with a test behind them.  PARITY UNPINNED in the sense of DESIGN.md section 0: the
twin is this repository's oracle, not the reference.

Run from the repo root (~6 minutes on 8 cores):  python tests/golden/make_train_twin_golden.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from instance_nerf_amd.scene import RoomScene           # noqa: E402
from oracle import field, hashgrid, rays, render        # noqa: E402

H = W = 64
K = 16
N_RAYS = 1024
N_NERF = 120
N_INST = 80
MIN_NEAR = 0.05
LR, ITERS = 1e-2, 1000


def workload():
    """Cameras, fixed batches and analytic ground truth; deterministic (numpy default_rng seeds)."""
    room = RoomScene()
    poses, intr, _, _ = room.cameras(n=9, seed=1, H=H, W=W, focal=W / 2.0)       # 0..7 train, 8 held out
    rng = np.random.default_rng(77)
    steps = [(int(rng.integers(0, 8)), rng.integers(0, H * W, size=N_RAYS)) for _ in range(N_NERF + N_INST)]
    gt_rgb, gt_ids = [], []
    for v in range(9):
        r = rays.get_rays(poses[v:v + 1], intr, H, W)
        rgb, ids, _ = room.trace(r["rays_o"][0], r["rays_d"][0])
        gt_rgb.append(rgb)
        gt_ids.append(ids)
    return {"room": room, "poses": poses.astype(np.float32), "intrinsics": intr, "steps": steps,
            "gt_rgb": np.stack(gt_rgb).astype(np.float32), "gt_ids": np.stack(gt_ids).astype(np.int64)}


def initial_params(table):
    return field.init_params(seed=21, table=table, table_std=1e-4, K=K)


def labels_of(ids, step):
    """CE targets of a batch: ground-truth id mod K, every 10th ray of the batch ignored (-1)."""
    lab = ids % K
    return np.where((np.arange(lab.shape[0]) + step) % 10 == 0, -1, lab)


def psnr(img, gt):
    return float(10.0 * np.log10(1.0 / np.mean((np.asarray(img, np.float64) - np.asarray(gt, np.float64)) ** 2)))


def miou(pred, truth, n_classes=K):
    """Mean IoU over the classes present in the truth (the product's MIoUMeter rule)."""
    ious = []
    for c in range(n_classes):
        p, t = pred == c, truth == c
        if t.sum():
            ious.append(np.logical_and(p, t).sum() / np.logical_or(p, t).sum())
    return float(np.mean(ious))


def lr_at(step):
    return LR * 0.1 ** min((step + 2) / ITERS, 1)          # the Trainer's rule with global_step starting at 1


def run_oracle(cfg, bits, table, p0):
    nerf_keys = ("embeddings", "sigma_w0", "sigma_w1", "color_w0", "color_w1", "color_w2")
    inst_keys = ("inst_embeddings", "inst_w0", "inst_w1", "inst_w2")
    p = {k: v.clone() for k, v in p0.items()}
    losses, totals = [], []
    for stage, keys, n0, n1 in (("nerf", nerf_keys, 0, N_NERF), ("instance", inst_keys, N_NERF, N_NERF + N_INST)):
        for k in p:
            p[k] = p[k].detach().requires_grad_(k in keys)
        opt = torch.optim.Adam([p[k] for k in keys], lr=LR, betas=(0.9, 0.99), eps=1e-15)
        for s in range(n0, n1):
            view, inds = cfg["steps"][s]
            r = rays.get_rays(cfg["poses"][view:view + 1], cfg["intrinsics"], H, W, inds=inds)
            for g in opt.param_groups:
                g["lr"] = lr_at(s - n0)
            o = render.render_train(r["rays_o"][0], r["rays_d"][0], p, table, bits, min_near=MIN_NEAR,
                                    with_instance=stage == "instance")
            if stage == "nerf":
                loss = ((o["image"] - torch.from_numpy(cfg["gt_rgb"][view][inds])) ** 2).mean()
            else:
                loss = render.instance_ce_loss(o["instance"], labels_of(cfg["gt_ids"][view][inds], s))
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
            totals.append(int(o["total"]))
    p = {k: v.detach() for k, v in p.items()}
    r = rays.get_rays(cfg["poses"][8:9], cfg["intrinsics"], H, W)
    with torch.no_grad():
        held = render.render_train(r["rays_o"][0], r["rays_d"][0], p, table, bits, min_near=MIN_NEAR, with_instance=True)
    image = held["image"].numpy()
    ids = held["instance"].argmax(-1).numpy()
    return np.asarray(losses), np.asarray(totals), image, ids


def main():
    cfg = workload()
    bits = cfg["room"].density_bitfield(128, 1.0)
    table = hashgrid.level_table()
    t0 = time.time()
    losses, totals, image, ids = run_oracle(cfg, bits, table, initial_params(table))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_twin.npz")
    res = {"psnr_db": psnr(image, cfg["gt_rgb"][8]), "miou": miou(ids, cfg["gt_ids"][8] % K)}
    np.savez_compressed(out, losses=losses, totals=totals, held_out_image=image.astype(np.float16),
                        held_out_ids=ids.astype(np.int8), psnr_db=res["psnr_db"], miou=res["miou"],
                        step_views=np.asarray([v for v, _ in cfg["steps"]]),
                        step_inds=np.stack([i for _, i in cfg["steps"]]).astype(np.int32),
                        n_nerf=N_NERF, n_inst=N_INST)
    print(f"train_twin.npz: {time.time() - t0:.0f} s, NeRF loss {losses[0]:.4f} -> {losses[N_NERF - 1]:.4f}, "
          f"CE {losses[N_NERF]:.4f} -> {losses[-1]:.4f}, held-out PSNR {res['psnr_db']:.2f} dB, mIoU {res['miou']:.3f}")


if __name__ == "__main__":
    main()
