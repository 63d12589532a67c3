"""How the per-level maxima of the table gradient move from step to step in both training stages (round 6: constants of
the fixed-point scatter).  Trains the synthetic room from disk (NeRF stage, then the instance stage on the frozen result),
reads the fixed-point state after every step (a host sync per step: a probe, not a benchmark), and replays the scale rule
offline for several (decay, headroom) pairs: how many level-steps would have come within 4x of the int32 range (near
miss), how many would have WRAPPED, and what the quantum is relative to the step's own maximum.
usage: python tools/fx_dynamics_probe.py [steps=1500]"""
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd.nerf import NeRFNetwork                              # noqa: E402
from instance_nerf_amd.nerf.provider import NeRFDataset                     # noqa: E402
from instance_nerf_amd.nerf.utils import Trainer                            # noqa: E402
from instance_nerf_amd.scene import RoomScene                               # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda", 0)
room = RoomScene()
d = tempfile.mkdtemp(prefix="inr_fxdyn_")
scene = room.write_dataset(d, n_views=24, H=400, W=400, num_instances=16, ignore_frac=0.1)


def train(stage, net):
    ds = NeRFDataset(d, type="train", device=dev, scale=1.0, num_rays=4096, preload=True, seed=0,
                     mask_dir=scene["mask_dir"] if stage == "instance" else None, num_instances=16 if stage == "instance" else 0)
    tr = Trainer("fxdyn", None, net, stage=stage, device=dev, lr=1e-2, iters=steps, workspace=None, mute=True,
                 update_extra_interval=16 if stage == "nerf" else 10 ** 9)
    tr.global_step = 0 if stage == "nerf" else 1
    table = net.instance_encoder.embeddings if stage == "instance" else net.encoder.embeddings
    maxima = []
    it = iter(())
    for s in range(steps):
        try:
            b = next(it)
        except StopIteration:
            it = iter(ds)
            b = next(it)
        tr.train_one_step(b)
        maxima.append(table._fx_state[32:48].cpu().numpy().copy())
    return np.stack(maxima)                    # [steps, 16]


def replay(m, decay, headroom):
    ref = m[0].copy()
    near = wrap = 0
    worst = 0.0
    coarse = []
    for t in range(1, len(m)):
        with np.errstate(divide="ignore"):
            scale = np.where(ref > 0, 2.0 ** np.floor(np.log2(2.0 ** 30 / (headroom * ref))), 0.0)
        used = m[t] * scale / 2.0 ** 31                       # fraction of the int32 range the largest row sum takes
        near += int((used > 0.125).sum())
        wrap += int((used > 1.0).sum())
        worst = max(worst, float(used.max()))
        ok = (m[t] > 0) & (scale > 0)
        coarse.append(np.median(1.0 / (scale[ok] * m[t][ok])))  # quantum / this step's maximum
        ref = np.maximum(m[t], decay * ref)
    return near, wrap, worst, float(np.median(coarse))


torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=16).to(dev)
for stage in ("nerf", "instance"):
    m = train(stage, net)
    r = m[1:] / np.maximum(m[:-1], 1e-30)
    print(f"{stage} stage, {steps} steps: step-to-step growth of a level's maximum: median {np.median(r):.2f}, 99th pct {np.percentile(r, 99):.1f}, "
          f"99.9th {np.percentile(r, 99.9):.1f}, max {r.max():.1f}  (level of the max: {int(np.unravel_index(r.argmax(), r.shape)[1])})")
    print(f"  {'decay':>6} {'headroom':>8}  near misses  wraps  worst use of the range  quantum / step maximum (median)")
    for decay in (0.5, 0.75, 0.9, 0.97, 0.99):
        for headroom in (64.0, 256.0, 1024.0):
            near, wrap, worst, q = replay(m, decay, headroom)
            print(f"  {decay:6.2f} {headroom:8.0f}  {near:11d}  {wrap:5d}  {worst:22.3f}  {q:.1e}")
