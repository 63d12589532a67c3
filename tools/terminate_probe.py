"""Full 800x800 frames: two-kernel path vs field+compositing with early termination, transparent and opaque scene."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import build_network
from instance_nerf_amd.nerf.utils import get_rays

dev = torch.device("cuda", 0)
net, room = build_network(dev)
poses, intr, H, W = room.cameras()
pd = torch.from_numpy(poses).to(dev)
for scale, name in ((1.0, "transparent (random init, no ray terminates)"), (3000.0, "opaque (density x3000)")):
    net.density_scale = scale
    for mode in ("fused", "fused_terminate", "auto"):
        def frame(v):
            r = get_rays(pd[v:v + 1], intr, H, W, patch=4)
            with torch.no_grad():
                return net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode=mode)
        frame(0); torch.cuda.synchronize()
        import gc
        gc.collect(); gc.disable()      # a gen-2 pass inside the loop (~50 ms here) once read as "12 ms per frame"
        t0 = time.perf_counter(); tot = 0; ev = 0
        for v in range(8):
            o = frame(v); tot += int(o["num_samples"][0]); ev += int(o["num_evaluated"][0]) if "num_evaluated" in o else int(o["num_samples"][0])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
        gc.enable()
        print(f"{name:45s} {mode:16s} {dt*1e3:7.2f} ms/frame  marched {tot/8/1e6:.1f} M  evaluated {ev/8/1e6:.1f} M  mean opacity {float(o['weights_sum'].mean()):.3f}")
