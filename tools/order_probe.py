"""Field-kernel time vs ray processing order (row-major, 2D tiles of various sizes, Morton)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_network
from instance_nerf_amd import raymarching
from instance_nerf_amd.nerf.utils import get_rays

dev = torch.device("cuda", 0)
net, room = build_network(dev)
poses, intr, H, W = room.cameras()
view = int(sys.argv[1]) if len(sys.argv) > 1 else 0

def morton2(i, j):
    def ex(v):
        v = (v | (v << 8)) & 0x00FF00FF; v = (v | (v << 4)) & 0x0F0F0F0F
        v = (v | (v << 2)) & 0x33333333; v = (v | (v << 1)) & 0x55555555
        return v
    return ex(i) | (ex(j) << 1)

ii, jj = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy")
ii, jj = ii.reshape(-1).long(), jj.reshape(-1).long()
orders = {"row-major": torch.arange(H * W)}
for t in (4, 8, 16, 32):
    key = ((jj // t) * (W // t) + (ii // t)) * (t * t) + (jj % t) * t + (ii % t)
    orders[f"tile{t}x{t}"] = torch.argsort(key)
orders["morton"] = torch.argsort(morton2(ii, jj))
for name, inds in orders.items():
    r = get_rays(torch.from_numpy(poses[view:view + 1]).to(dev), intr, H, W, inds=inds.to(dev))
    ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_infer, net.min_near)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1, net.density_bitfield, 1, 128, nears, fars, force_all_rays=True)
    M = xyzs.shape[0]
    with torch.no_grad():
        net(xyzs, dirs); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            net(xyzs, dirs)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:10s} M={M} field {dt*1e3:.3f} ms {M/dt/1e9:.3f} Gsamples/s", flush=True)
