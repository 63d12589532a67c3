"""Lattice extraction kernel (k_nerf_fwd_dirs<lattice>) at 160^3 with 1 / 2 / 4 view directions: is it gather- or MLP-bound?
python tools/extract_dirs_probe.py"""
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
axes = extract.lattice_axes(np.asarray([-1, -1, -1], np.float32), np.asarray([1, 1, 1], np.float32), res, dev)
dirs = torch.from_numpy(extract.VIEW_DIRS).to(dev)
for D in ([int(os.environ["PROBE_DIRS"])] if os.environ.get("PROBE_DIRS") else (1, 2, 4)):
    d = dirs[:D].contiguous()
    net.forward_lattice(axes, d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        net.forward_lattice(axes, d)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    n = int(res.prod())
    print(f"D={D}: {ms:.3f} ms per 160^3 lattice, {n / ms / 1e3:.0f} Mvoxels/s, frac of 1040 B/voxel x 8 TB/s {n * 1040 / (ms * 1e-3) / 8e12:.3f}")
