"""Full-frame render WITH the instance head (K=64 logits composited per pixel)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.nerf.utils import get_rays
from instance_nerf_amd.scene import RoomScene

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=64).to(dev).eval()
room = RoomScene()
net.density_bitfield.copy_(torch.from_numpy(room.density_bitfield(128, 1.0)).to(dev))
poses, intr, H, W = room.cameras()
r = get_rays(torch.from_numpy(poses[:1]).to(dev), intr, H, W, patch=4)
with torch.no_grad():
    out = net.render(r["rays_o"], r["rays_d"], bg_color=1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = net.render(r["rays_o"], r["rays_d"], bg_color=1)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
M = int(out["num_samples"][0])
print(f"render+instance: {dt*1e3:.2f} ms per frame, {M} samples, {M/dt/1e6:.0f} Msamples/s, instance {tuple(out['instance'].shape)}")

# column efficiency of the group-owned traversal: a step of a 16-ray group is one 16-column MFMA tile whatever the
# number of rays that still have a sample there
from instance_nerf_amd import raymarching
ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_infer, net.min_near)
_, _, _, rays = raymarching.march_rays_patch(ro, rd, net.bound, net.density_bitfield, net.cascade, net.grid_size, nears, fars, 0, 1024)
cnt = rays[:, 2].view(-1, 16)
steps = int(cnt.max(dim=1).values.sum())
print(f"group steps {steps} x 16 = {steps*16} columns for {int(cnt.sum())} samples: column efficiency {int(cnt.sum())/(steps*16):.3f}")
