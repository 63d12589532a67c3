"""Does the ORDER of the lattice points matter to the extraction kernel (k_nerf_fwd_dirs)?  160^3 voxel centres in
(w, l, h) order with h fastest (the writer's order), with w fastest (x is the fastest index of the hash table's rows),
in 4x4x1 / 2x2x4 bricks, and with 1 direction instead of 4 (how much is the colour net?).
python tools/extract_order_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd.extract import VIEW_DIRS, lattice          # noqa: E402
from instance_nerf_amd.nerf import NeRFNetwork                     # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05).to(dev).eval()
res = [160, 160, 160]
pts = lattice([-1, -1, -1], [1, 1, 1], res, dev).view(160, 160, 160, 3)
dirs = torch.from_numpy(VIEW_DIRS).to(dev)


def timed(x, d, n=10):
    x = x.reshape(-1, 3).contiguous()
    net.forward_dirs(x, d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        net.forward_dirs(x, d)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def bricks(p, bw, bl, bh):
    W, L, H = p.shape[:3]
    q = p.view(W // bw, bw, L // bl, bl, H // bh, bh, 3).permute(0, 2, 4, 5, 3, 1, 6)    # brick-major, w fastest inside
    return q


print(f"h fastest (writer's order), 4 dirs: {timed(pts, dirs):.3f} ms")
print(f"w fastest, 4 dirs:                  {timed(pts.permute(2, 1, 0, 3), dirs):.3f} ms")
print(f"16x1x1 runs along w inside (l,h):   {timed(pts.permute(1, 2, 0, 3), dirs):.3f} ms")
print(f"4x4x1 bricks:                       {timed(bricks(pts, 4, 4, 1), dirs):.3f} ms")
print(f"4x2x2 bricks:                       {timed(bricks(pts, 4, 2, 2), dirs):.3f} ms")
print(f"8x8x8 bricks of 4x4x1:              {timed(bricks(pts, 8, 8, 8), dirs):.3f} ms")
print(f"h fastest, 1 dir:                   {timed(pts, dirs[:1]):.3f} ms")
print(f"4x4x1 bricks, 1 dir:                {timed(bricks(pts, 4, 4, 1), dirs[:1]):.3f} ms")
from instance_nerf_amd.extract import lattice_axes               # noqa: E402
axes = lattice_axes([-1, -1, -1], [1, 1, 1], res, dev)
for nd in (4, 1):
    net.forward_lattice(axes, dirs[:nd])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        net.forward_lattice(axes, dirs[:nd])
    e1.record()
    torch.cuda.synchronize()
    print(f"lattice launch (runs along w, from the axes), {nd} dir(s): {e0.elapsed_time(e1) / 10:.3f} ms")
with torch.no_grad():
    x = pts.reshape(-1, 3).contiguous()
    d = dirs[:1].expand(x.shape[0], 3).contiguous()
    net(x, d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        net(x, d)
    e1.record()
    torch.cuda.synchronize()
    print(f"plain field kernel (sigma + rgb, one dir per point), h fastest: {e0.elapsed_time(e1) / 10:.3f} ms")
