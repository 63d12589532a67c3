"""End-to-end demo on the synthetic room: stage 1 NeRF (MSE), stage 2 instance field (CE, NeRF frozen).
Reports PSNR on a held-out view and instance mIoU.  usage: python tools/train_demo.py [nerf_steps] [inst_steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
from instance_nerf_amd.nerf.utils import MIoUMeter, PSNRMeter, Trainer, get_rays

nerf_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
inst_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda", 0)
torch.manual_seed(int(os.environ.get("DEMO_SEED", "0")))
K = 16
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=K).to(dev)
ds = SyntheticRoomDataset(dev, H=400, W=400, n_views=24, num_rays=4096, num_instances=K, ignore_frac=0.1)
room = ds.room

def eval_view(view_pose, stage):
    r = get_rays(view_pose, ds.intrinsics, ds.H, ds.W, patch=4)
    rgb, ids, _ = room.trace(r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy())
    net.eval()
    with torch.no_grad():
        out = net.render(r["rays_o"], r["rays_d"], bg_color=1)
    if stage == "nerf":
        m = PSNRMeter(); m.update(out["image"][0], torch.from_numpy(rgb).to(dev)); return m.measure()
    m = MIoUMeter(K); m.update(out["instance"][0].argmax(-1), torch.from_numpy(ids % K)); return m.measure()

held_out = torch.from_numpy(room.look_at([0.3, -0.2, 0.1])[None]).to(dev)
# ---- stage 1: NeRF from scratch, occupancy grid learned with update_extra_state every 16 steps
for p in list(net.instance_encoder.parameters()) + list(net.instance_net.parameters()):
    p.requires_grad_(False)
tr = Trainer("demo", None, net, stage="nerf", device=dev, lr=1e-2, iters=nerf_steps)
t0 = time.perf_counter()
for s in range(nerf_steps):
    loss = tr.train_one_step(ds.batch())
    if s % 250 == 0 or s == nerf_steps - 1:
        print(f"[nerf] step {s:5d} loss {float(loss):.5f} mean_density {net.mean_density:.3f} occupied {float((net.density_grid > min(net.mean_density, 10)).float().mean()):.3f}", flush=True)
torch.cuda.synchronize()
print(f"[nerf] {nerf_steps} steps in {time.perf_counter()-t0:.1f} s (incl. analytic ground-truth tracing on the CPU)")
print(f"[nerf] held-out view PSNR {eval_view(held_out, 'nerf'):.2f} dB")
# ---- stage 2: instance field, NeRF frozen
for p in list(net.instance_encoder.parameters()) + list(net.instance_net.parameters()):
    p.requires_grad_(True)
tr2 = Trainer("demo_inst", None, net, stage="instance", device=dev, lr=1e-2, iters=inst_steps, update_extra_interval=10 ** 9)
tr2.global_step = 1
for s in range(inst_steps):
    loss = tr2.train_one_step(ds.batch())
    if s % 150 == 0 or s == inst_steps - 1:
        print(f"[inst] step {s:5d} CE {float(loss):.4f}", flush=True)
print(f"[inst] held-out view instance mIoU {eval_view(held_out, 'instance'):.3f}")
