"""Magnitudes of dL/d(encoder output) - the per-sample, per-level gradients the table scatter multiplies by the trilinear
weights - in steady-state training steps on the trained room, both stages (round 6: is there a FAITHFUL skip threshold
for the fp32 scatter?  Adam with eps 1e-15 damps a row gradient g to lr g / (|g| + 1e-15): contributions far below 1e-15
cannot move a row).  Prints the cumulative distribution of |denc| over decades.
usage: python tools/denc_histogram_probe.py [steps=1500]"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd.nerf import NeRFNetwork, network                     # noqa: E402
from instance_nerf_amd.nerf.provider import NeRFDataset                     # noqa: E402
from instance_nerf_amd.nerf.utils import Trainer                            # noqa: E402
from instance_nerf_amd.scene import RoomScene                               # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda", 0)
room = RoomScene()
d = tempfile.mkdtemp(prefix="inr_denc_")
scene = room.write_dataset(d, n_views=24, H=400, W=400, num_instances=16, ignore_frac=0.1)
edges = np.array([0.0] + [10.0 ** e for e in range(-40, 1, 2)])
hist = {}
real = network._table_backward


def spy(lib, x, denc, desc, M, bound, g_emb, emb):
    if spy.on and M:
        a = denc[:M].abs().flatten()
        h = torch.histc(torch.log10(a.clamp(min=1e-45)), bins=23, min=-46, max=0)
        key = spy.stage
        hist[key] = hist.get(key, 0) + h.cpu().numpy()
        hist[key + "_zero"] = hist.get(key + "_zero", 0) + int((a == 0).sum())
        hist[key + "_n"] = hist.get(key + "_n", 0) + a.numel()
    return real(lib, x, denc, desc, M, bound, g_emb, emb)


spy.on = False
network._table_backward = spy
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=16).to(dev)
for stage in ("nerf", "instance"):
    ds = NeRFDataset(d, type="train", device=dev, scale=1.0, num_rays=4096, preload=True, seed=0,
                     mask_dir=scene["mask_dir"] if stage == "instance" else None, num_instances=16 if stage == "instance" else 0)
    tr = Trainer("denc", None, net, stage=stage, device=dev, lr=1e-2, iters=steps, workspace=None, mute=True,
                 update_extra_interval=16 if stage == "nerf" else 10 ** 9)
    tr.global_step = 0 if stage == "nerf" else 1
    spy.stage = stage
    it = iter(())
    for s in range(steps + 32):
        try:
            b = next(it)
        except StopIteration:
            it = iter(ds)
            b = next(it)
        spy.on = s >= steps
        tr.train_one_step(b)
    spy.on = False
    h, z, n = hist[stage], hist[stage + "_zero"], hist[stage + "_n"]
    print(f"{stage} stage, 32 steady-state steps after {steps}: {n / 32:.0f} entries of denc per step, exactly zero {z / n:.3f}")
    cum = np.cumsum(h) / n
    for i in range(23):
        hi = -46 + 2 * (i + 1)
        if h[i] > 0:
            print(f"   |denc| < 1e{hi:+d}: {cum[i]:.4f}")
