"""Four 800x800 views through the two-kernel path and through the early-terminating kernel on the transparent bench scene
(the same samples are evaluated): the workload of the PMC comparison `bash tools/pmc_passes.sh <tag> tools/terminate_pmc_probe.py`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_network
from instance_nerf_amd.nerf.utils import get_rays

dev = torch.device("cuda", 0)
net, room = build_network(dev)
poses, intr, H, W = room.cameras()
pd = torch.from_numpy(poses).to(dev)
for mode in ("fused", "fused_terminate"):
    for v in range(4):
        r = get_rays(pd[v:v + 1], intr, H, W, patch=4)
        with torch.no_grad():
            o = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode=mode)
    torch.cuda.synchronize()
print("M=", int(o["num_samples"][0]))
