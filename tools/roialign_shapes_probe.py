"""Separable vs lane-per-output RoIAlign-3D over the shapes a feature pyramid produces (levels of 80^3 .. 5^3, 5 / 5 / 7 / 10 / 14
bins, 64 .. 512 boxes): no shape may be slower on the default (separable) kernels.  python tools/roialign_shapes_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd import _lib                                        # noqa: E402
from instance_nerf_amd.roi_align.roi_align import roi_align_3d           # noqa: E402

dev = "cuda"
lib = _lib.load()


def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


gen = torch.Generator(device=dev).manual_seed(0)
print(f"{'volume':>18} {'boxes':>5} {'out':>4}  separable  lane/output   (ms, forward)")
for C, S in ((256, 80), (256, 40), (256, 20), (256, 10), (256, 5), (64, 40), (16, 40)):
    feat = torch.randn(1, C, S, S, S, device=dev)
    for K in (64, 512):
        lo = torch.rand(K, 3, device=dev, generator=gen) * 0.6 * S
        rois = torch.cat([lo, lo + 1 + torch.rand(K, 3, device=dev, generator=gen) * 0.4 * S], 1)
        inds = torch.zeros(K, dtype=torch.int32, device=dev)
        for o in (5, 7, 10, 14):
            ts = []
            for mode in (2, 1):
                lib.inr_roi_align_3d_set_mode(mode)
                try:
                    ts.append(timed(lambda: roi_align_3d(feat, rois, inds, o, o, o, 1.0)))
                except RuntimeError:
                    ts.append(float("nan"))
            lib.inr_roi_align_3d_set_mode(0)
            flag = "" if not ts[0] > ts[1] else "   <-- separable slower"
            print(f"[1,{C},{S},{S},{S}]".rjust(18), f"{K:5d} {o:4d}   {ts[0]:8.4f}   {ts[1]:8.4f}{flag}")

# The reference's third call shape (round-4 verdict): the ground-truth mask crop of the mask loss,
# roi_align_3d(gt_masks[:, None], rois, (M, M, M), 1.0) - C = 1 on [G,1,160,160,160], M = 20, one box per positive
# proposal (/root/reference/nerf_rcnn/model/nerf_rcnn.py:819-831, 846-849).  `auto` = what the product runs.
print("\nGT-mask crop [G,1,160,160,160] -> 20^3, scale 1.0 (forward only: the masks carry no gradient)")
G = 30
masks = (torch.rand(G, 1, 160, 160, 160, device=dev, generator=gen) > 0.5).float()
for K in (64, 512):
    lo = torch.rand(K, 3, device=dev, generator=gen) * 100
    rois = torch.cat([lo, lo + 8 + torch.rand(K, 3, device=dev, generator=gen) * 50], 1)
    inds = torch.randint(0, G, (K,), device=dev, generator=gen).to(torch.int32)
    ts = {}
    for name, mode in (("auto", 0), ("lane/output", 1)):
        lib.inr_roi_align_3d_set_mode(mode)
        ts[name] = timed(lambda: roi_align_3d(masks, rois, inds, 20, 20, 20, 1.0))
    lib.inr_roi_align_3d_set_mode(0)
    cells = float(((rois[:, 3:] - rois[:, :3]).clamp(max=160).prod(1)).sum())
    print(f"  {K:4d} boxes   auto {ts['auto']:8.4f} ms   lane/output {ts['lane/output']:8.4f} ms   "
          f"(voxels inside the boxes: {cells / 1e6:.1f} M = {cells * 4 / 1e6:.0f} MB read once)")
