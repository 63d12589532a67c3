"""When does each workgroup of the fused field kernel run dry?  Needs the profiling build `tools/build_probe.py 3`
(INR_LIB_PATH=tools/_probe/libinr_probe3.so): k_nerf_fwd then leaves (start, end) of every workgroup in the geo buffer.
One 800x800 frame's samples through the plain feed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from bench import build_network
from instance_nerf_amd import _lib, raymarching
from instance_nerf_amd._lib import check, ptr
from instance_nerf_amd.nerf.utils import get_rays

dev = torch.device("cuda", 0)
net, room = build_network(dev)
poses, intr, H, W = room.cameras()
lib = _lib.load()
for view in (0, 3):
    r = get_rays(torch.from_numpy(poses[view:view + 1]).to(dev), intr, H, W, patch=4)
    ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_infer, net.min_near)
    xyzs, dirs, deltas, rays = raymarching.march_rays_patch(ro, rd, 1, net.density_bitfield, 1, 128, nears, fars)
    M = xyzs.shape[0]
    sigma = torch.empty(M, device=dev); rgb = torch.empty(M, 3, device=dev)
    for rep in range(3):
        dbg = torch.zeros(4096, dtype=torch.int64, device=dev)
        check(lib.inr_nerf_forward(ptr(xyzs), ptr(dirs), M, None, 1.0, ptr(net.encoder.embeddings.data), net.encoder.desc,
                                   ptr(net._packed_weights("nerf")), 1.0, ptr(sigma), ptr(rgb), ptr(dbg.view(torch.float32)),
                                   _lib.stream_ptr()), "fwd")
        torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(-1, 2)
    t = t[t[:, 1] > 0]
    t0 = t[:, 0].min()
    end = (t[:, 1] - t0) / 100.0          # 100 MHz constant clock -> us
    xcd = np.arange(t.shape[0]) % 8
    print(f"view {view}: {M} samples, {t.shape[0]} workgroups; kernel {end.max():.0f} us; workgroups run dry at "
          f"min {end.min():.0f} / median {np.median(end):.0f} / max {end.max():.0f} us")
    print("   last workgroup of each XCD:", [int(end[xcd == x].max()) for x in range(8)],
          " first dry:", [int(end[xcd == x].min()) for x in range(8)])
