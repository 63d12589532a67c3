"""Field-kernel time vs SAMPLE layout: ray-major vs patch-depth-major (samples of a PxP pixel patch
interleaved by step index).  Emulated by permuting xyzs/dirs with torch; validates the layout idea."""
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
ii, jj = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="xy")
ii, jj = ii.reshape(-1).long(), jj.reshape(-1).long()

def run(name, xyzs, dirs):
    M = xyzs.shape[0]
    with torch.no_grad():
        net(xyzs, dirs); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            net(xyzs, dirs)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:28s} M={M} field {dt*1e3:.3f} ms {M/dt/1e9:.3f} Gsamples/s", flush=True)

for P in (2, 4, 8):
    key = ((jj // P) * (W // P) + (ii // P)) * (P * P) + (jj % P) * P + (ii % P)
    inds = torch.argsort(key).to(dev)
    r = get_rays(torch.from_numpy(poses[view:view + 1]).to(dev), intr, H, W, inds=inds)
    ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_infer, net.min_near)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1, net.density_bitfield, 1, 128, nears, fars, force_all_rays=True)
    M = xyzs.shape[0]
    run(f"tile{P}x{P} ray-major", xyzs, dirs)
    cnt = rays[:, 2].long()
    ray_of = torch.repeat_interleave(torch.arange(rays.shape[0], device=dev), cnt)
    k_of = torch.arange(M, device=dev) - rays[:, 1].long()[ray_of]
    patch = ray_of // (P * P)
    key2 = (patch * 1024 + k_of) * (P * P) + (ray_of % (P * P))
    perm = torch.argsort(key2)
    run(f"tile{P}x{P} patch-depth-major", xyzs[perm].contiguous(), dirs[perm].contiguous())

# upper-bound experiment: full 3D Morton sort of all samples (global) and within 16x16 patches
from instance_nerf_amd import raymarching as rm
P = 16
key = ((jj // P) * (W // P) + (ii // P)) * (P * P) + (jj % P) * P + (ii % P)
inds = torch.argsort(key).to(dev)
r = get_rays(torch.from_numpy(poses[view:view + 1]).to(dev), intr, H, W, inds=inds)
ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
nears, fars = rm.near_far_from_aabb(ro, rd, net.aabb_infer, net.min_near)
xyzs, dirs, deltas, rays = rm.march_rays_train(ro, rd, 1, net.density_bitfield, 1, 128, nears, fars, force_all_rays=True)
M = xyzs.shape[0]
q = ((xyzs + 1) * 0.5 * 1023).clamp(0, 1023).int()
code = rm.morton3D(q).long()
run("global 3D-morton sort", xyzs[torch.argsort(code)].contiguous(), dirs[torch.argsort(code)].contiguous())
cnt = rays[:, 2].long()
ray_of = torch.repeat_interleave(torch.arange(rays.shape[0], device=dev), cnt)
patch = ray_of // (P * P)
perm = torch.argsort(patch * (1 << 30) + code)
run("16x16-patch 3D-morton sort", xyzs[perm].contiguous(), dirs[perm].contiguous())
