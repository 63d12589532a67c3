"""Times the two RoIAlign-3D implementations on BASELINE configs[4] (256 boxes -> 10^3 x 256 channels on
[1,256,40,40,40]) with events on the launch stream: forward and backward, ms per launch and the fraction of the
compulsory-bytes floor (output + input once = 328 MB at 8 TB/s).  python tools/roialign_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd import _lib                                        # noqa: E402
from instance_nerf_amd.roi_align.roi_align import roi_align_3d           # noqa: E402


def timed(fn, n=20):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[len(ts) // 2]


def main():
    dev = "cuda"
    lib = _lib.load()
    feat = torch.randn(1, 256, 40, 40, 40, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    lo = torch.rand(256, 3, device=dev, generator=gen) * 100
    rois = torch.cat([lo, lo + 10 + torch.rand(256, 3, device=dev, generator=gen) * 50], 1)
    inds = torch.zeros(256, dtype=torch.int32, device=dev)
    g = torch.randn(256, 256, 10, 10, 10, device=dev)
    out_bytes, in_bytes = g.numel() * 4, feat.numel() * 4
    res = {}
    for name, mode in (("separable", 2), ("lane_per_output", 1)):
        lib.inr_roi_align_3d_set_mode(mode)
        x = feat.clone().requires_grad_(True)
        fwd = timed(lambda: roi_align_3d(feat, rois, inds, 10, 10, 10, 0.25))
        out = roi_align_3d(x, rois, inds, 10, 10, 10, 0.25)

        def bwd():
            x.grad = None
            out.backward(g, retain_graph=True)
        t_b = timed(bwd, 10)
        res[name] = {"forward_ms": round(fwd, 4), "forward_frac_of_byte_floor": round((out_bytes + in_bytes) / (fwd * 1e-3) / 8e12, 3),
                     "backward_ms_incl_zero_fill": round(t_b, 4)}
    lib.inr_roi_align_3d_set_mode(0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
