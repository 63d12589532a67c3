"""Rendering a TRAINED scene (the realistic regime: learned occupancy grid, opaque surfaces): train the NeRF of the
synthetic room for a few seconds, then time full 800x800 frames in the two inference modes and report how many samples
are marched and how many the field evaluates."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
from instance_nerf_amd.nerf.utils import Trainer, get_rays

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10).to(dev)
ds = SyntheticRoomDataset(dev, H=400, W=400, n_views=24, num_rays=4096)
tr = Trainer("p", None, net, stage="nerf", device=dev, lr=1e-2, iters=steps)
for s in range(steps):
    tr.train_one_step(ds.batch())
net.eval()
occ = float((net.density_grid > min(net.mean_density, net.density_thresh)).float().mean())
print(f"trained {steps} steps: occupied cells {occ:.3f}, mean_count {net.mean_count}")
poses, intr, H, W = ds.room.cameras()
pd = torch.from_numpy(poses).to(dev)
for mode in ("fused", "fused_terminate"):
    def frame(v):
        r = get_rays(pd[v:v + 1], intr, H, W, patch=4)
        with torch.no_grad():
            return net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode=mode)
    frame(0); torch.cuda.synchronize()
    t0 = time.perf_counter(); tot = ev = 0
    for v in range(8):
        o = frame(v); tot += int(o["num_samples"][0]); ev += int(o["num_evaluated"][0]) if "num_evaluated" in o else int(o["num_samples"][0])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print(f"{mode:16s} {dt*1e3:7.2f} ms/frame  marched {tot/8/1e6:.2f} M  evaluated {ev/8/1e6:.2f} M  mean opacity {float(o['weights_sum'].mean()):.3f}")
