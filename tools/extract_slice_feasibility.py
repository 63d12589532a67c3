"""Would the level-major pre-pass of the sliced frame path (k_grid_fine_slices + k_nerf_fwd<.., kPre>) pay for the lattice
extraction?  The 160^3 lattice's points, in the extraction kernel's own traversal order (runs of 16 along W), are fed
through the FRAME kernels as one 'ray' (one colour pass, like extract_dirs_probe's D = 1): fused vs sliced.
python tools/extract_slice_feasibility.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_network                                    # noqa: E402
from instance_nerf_amd import extract                              # noqa: E402

dev = torch.device("cuda", 0)
net, _ = build_network(dev)
res = extract.grid_resolution([-1, -1, -1], [1, 1, 1], 160)
ax = extract.lattice_axes(np.asarray([-1, -1, -1], np.float32), np.asarray([1, 1, 1], np.float32), res, dev)
aw, al, ah = [a.float() for a in ax]
W, L, H = [int(v) for v in res]
b = float(net.bound)
# traversal: (il, ih) rows, iw fastest
x = aw.view(1, 1, W).expand(L, H, W)
y = al.view(L, 1, 1).expand(L, H, W)
z = ah.view(1, H, 1).expand(L, H, W)
x01 = ((torch.stack([x, y, z], -1).reshape(-1, 3).clamp(-b, b) + b) / (2 * b)).contiguous()
M = x01.shape[0]
ray_ids = torch.zeros(M, dtype=torch.int32, device=dev)
rd = torch.tensor([[0.0, 0.0, 1.0]], device=dev)
outs = {}
with torch.no_grad():
    shq = net.sh_table(rd)
    for mode, flag in (("fused", False), ("sliced", True)):
        net.frame_slices = flag
        net._slice_probe = None
        for _ in range(3):
            outs[mode] = net.forward_table(x01, ray_ids, rd, shq=shq)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            net.forward_table(x01, ray_ids, rd, shq=shq)
        e1.record()
        torch.cuda.synchronize()
        print(f"{mode:7s} {e0.elapsed_time(e1) / 10:.3f} ms per {M} lattice points (frame kernels, one colour pass)")
print("bit-identical:", bool(torch.equal(outs["fused"][0], outs["sliced"][0]) and torch.equal(outs["fused"][1], outs["sliced"][1])))
