"""Golden vectors of BASELINE configs[0] (64x64 Blender-style views, 1024 rays per step) from the CPU oracle:
tests/golden/config0.npz.  PARITY UNPINNED (see make_golden.py): these pin this repository's oracle.
Run from the repo root:  python tests/golden/make_config0_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import config0_workload                   # noqa: E402
from instance_nerf_amd.scene import RoomScene           # noqa: E402
from oracle import field, hashgrid, rays, render        # noqa: E402


def config0_params(table):
    """(parity parameters for the render check: table U(-1,1) so outputs are O(1); upstream-style initial parameters
    for the training steps: table U(-1e-4, 1e-4))."""
    return (field.init_params(seed=11, table=table, table_std=1.0, K=0),
            field.init_params(seed=12, table=table, table_std=1e-4, K=0))


def run_oracle(cfg, bits, table, p0, p_train, lr=1e-2, iters=100):
    """-> (image of view 0 [H*W,3], depth, sample total, per-step (loss, samples)) with the torch oracle."""
    H, W, intr = cfg["H"], cfg["W"], cfg["intrinsics"]
    r = rays.get_rays(cfg["poses"][cfg["view"]:cfg["view"] + 1], intr, H, W)
    out = render.render_train(r["rays_o"][0], r["rays_d"][0], p0, table, bits, min_near=cfg["min_near"])
    image, depth, total = out["image"].detach().numpy(), out["depth"].detach().numpy(), out["total"]
    trained = ("embeddings", "sigma_w0", "sigma_w1", "color_w0", "color_w1", "color_w2")
    p = {k: v.clone().requires_grad_(k in trained) for k, v in p_train.items()}
    opt = torch.optim.Adam([p[k] for k in trained], lr=lr, betas=(0.9, 0.99), eps=1e-15)
    losses, totals = [], []
    for s, (view, inds) in enumerate(cfg["steps"]):
        r = rays.get_rays(cfg["poses"][view:view + 1], intr, H, W, inds=inds)
        ro, rd = r["rays_o"][0], r["rays_d"][0]
        for g in opt.param_groups:
            g["lr"] = lr * 0.1 ** min((s + 2) / iters, 1)      # the Trainer's rule (global_step starts at 1)
        o = render.render_train(ro, rd, p, table, bits, min_near=cfg["min_near"])
        loss = ((o["image"] - torch.from_numpy(np.abs(rd))) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
        totals.append(o["total"])
    return image, depth, total, np.asarray(losses), np.asarray(totals)


def main():
    cfg = config0_workload()
    bits = RoomScene().density_bitfield(128, 1.0)
    table = hashgrid.level_table()
    p0, p_train = config0_params(table)
    image, depth, total, losses, totals = run_oracle(cfg, bits, table, p0, p_train)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config0.npz"),
                        poses=cfg["poses"], image=image.astype(np.float32), depth=depth.astype(np.float32),
                        total=total, losses=losses, totals=totals,
                        step_views=np.asarray([v for v, _ in cfg["steps"]]),
                        step_inds=np.stack([i for _, i in cfg["steps"]]))
    print("config0.npz: view-0 samples", total, "losses", losses, "step samples", totals)


if __name__ == "__main__":
    main()
