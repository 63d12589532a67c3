"""Times the 3-D mask projector (masks.soft_project) on one 800x800 view of the bench scene with 30 voxel
masks at 160^3: the whole call and the projection launch alone.  python tools/project_probe.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd.masks import pack_mask_words, soft_project          # noqa: E402
from instance_nerf_amd.nerf import NeRFNetwork                               # noqa: E402
from instance_nerf_amd.nerf.utils import get_rays                            # noqa: E402
from instance_nerf_amd.scene import RoomScene                                # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
room = RoomScene()
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05).to(dev).eval()
net.density_bitfield.copy_(torch.from_numpy(room.density_bitfield(128, 1.0)).to(dev))
poses, intr, H, W = room.cameras(H=800, W=800, focal=400.0)
r = get_rays(torch.from_numpy(poses[:1]).to(dev), intr, H, W, patch=4)
masks = np.random.default_rng(0).random((30, 160, 160, 160)) > 0.7
packed = (30, pack_mask_words(masks, dev))
for _ in range(2):
    soft, ws = soft_project(net, None, [-1, -1, -1], [1, 1, 1], r["rays_o"][0], r["rays_d"][0], packed=packed)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    soft, ws = soft_project(net, None, [-1, -1, -1], [1, 1, 1], r["rays_o"][0], r["rays_d"][0], packed=packed)
torch.cuda.synchronize()
print(f"soft_project 800x800, 30 masks of 160^3: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per view "
      f"(soft max {float(soft.max()):.3f}, opacity {float(ws.mean()):.3f})")
