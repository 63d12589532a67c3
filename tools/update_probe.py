"""Times NeRFRenderer.update_extra_state (every 16 training steps) - the full 128^3 sweep of the first 16
updates and the half-random / half-occupied sweep afterwards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.scene import RoomScene

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10).to(dev).train()
for phase, it in (("full sweep (iter_density < 16)", 0), ("steady state (iter_density >= 16)", 16)):
    net.iter_density = it
    if it:
        net.density_grid.copy_(torch.rand_like(net.density_grid) * (torch.rand_like(net.density_grid) < 0.1))
    for _ in range(2):
        net.update_extra_state()
    net.iter_density = it
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        net.update_extra_state()
        net.iter_density = it
    torch.cuda.synchronize()
    print(f"{phase}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per update")
