"""BASELINE configs[4]: density-grid extraction (160^3) + 3-D RoIAlign of 256 boxes -> 10^3 on [1,256,40,40,40]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_network
from instance_nerf_amd.extract import extract_rgbsigma
from instance_nerf_amd.roi_align.roi_align import roi_align_3d

dev = torch.device("cuda", 0)
net, room = build_network(dev)
extract_rgbsigma(net, max_side=160); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    g, res = extract_rgbsigma(net, max_side=160)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
n = int(res.prod())
print(f"extract_rgbsigma {tuple(res)}: {dt*1e3:.2f} ms ({n} points x (density + 4 colour queries), {n/dt/1e6:.0f} Mpoints/s)")
feat = torch.randn(1, 256, 40, 40, 40, device=dev, requires_grad=True)
gen = torch.Generator(device=dev).manual_seed(0)
lo = torch.rand(256, 3, device=dev, generator=gen) * 100
rois = torch.cat([lo, lo + 10 + torch.rand(256, 3, device=dev, generator=gen) * 50], 1)
inds = torch.zeros(256, dtype=torch.int32, device=dev)
out = roi_align_3d(feat, rois, inds, 10, 10, 10, 0.25); out.sum().backward(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = roi_align_3d(feat, rois, inds, 10, 10, 10, 0.25)
torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / 5
t0 = time.perf_counter()
for _ in range(5):
    feat.grad = None
    out = roi_align_3d(feat, rois, inds, 10, 10, 10, 0.25); out.sum().backward()
torch.cuda.synchronize(); tfb = (time.perf_counter() - t0) / 5
print(f"roi_align_3d 256 rois -> 10^3 x 256 ch: forward {tf*1e3:.3f} ms, forward+backward {tfb*1e3:.3f} ms, out {tuple(out.shape)}")
