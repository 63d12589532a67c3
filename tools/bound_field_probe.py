"""The fused field kernel on one frame's worth of marched samples, OFF the tuned configuration (round-4 verdict item 1):
which of bound / level table / scene scale / step growth costs what.  One line per configuration.

usage: python tools/bound_field_probe.py [reps] [config ...]
  config = bound:scene_scale:dt_gamma_inverse:desired_resolution:log2_T   (0 = default for that field)
  e.g.   1:1:0:0:0   the headline     4:4:128:0:0   bound 4, room x4, dt_gamma 1/128, levels up to 8192
Under rocprofv3 give ONE config (the PMC summary then belongs to it).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from instance_nerf_amd import raymarching  # noqa: E402
from instance_nerf_amd.nerf import NeRFNetwork  # noqa: E402
from instance_nerf_amd.nerf.utils import get_rays  # noqa: E402
from instance_nerf_amd.scene import RoomScene  # noqa: E402

args = sys.argv[1:]
reps = int(args[0]) if args else 5
configs = args[1:] or ["1:1:0:0:0", "2:2:128:0:0", "4:4:128:0:0", "4:4:0:0:0"]
dev = torch.device("cuda", 0)
VIEW = int(os.environ.get("PROBE_VIEW", "0"))
# frame paths to run: fused | sliced | auto (NeRFNetwork.frame_slices)
MODES = os.environ.get("PROBE_MODES", "fused").split(",")


def one(cfg):
    b, s, g, res, lt = (float(v) for v in cfg.split(":"))
    bound = int(b)
    gamma = 1.0 / g if g else 0.0
    kw = {}
    if res:
        kw["desired_resolution"] = int(res)
    if lt:
        kw["log2_hashmap_size"] = int(lt)
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True, bound=bound, min_near=0.05, density_thresh=10, encoder_kwargs=kw).to(dev).eval()
    room = RoomScene(scale=s)
    C, H = net.cascade, net.grid_size
    net.density_bitfield.copy_(torch.from_numpy(room.density_bitfield(H, float(bound))).to(dev))
    poses, intr, Hi, Wi = room.cameras()
    r = get_rays(torch.from_numpy(poses[VIEW:VIEW + 1]).to(dev), intr, Hi, Wi, patch=4)
    ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_infer, net.min_near)
    xyzs, dirs, deltas, rays = raymarching.march_rays_patch(ro, rd, bound, net.density_bitfield, C, H, nears, fars, gamma,
                                                            1024, table=True)
    M = xyzs.shape[0]
    tb = net.encoder.table
    ref = None
    for mode in MODES:
        net.frame_slices = {"fused": False, "sliced": True}.get(mode, "auto")
        net._slice_probe = None
        with torch.no_grad():
            shq = net.sh_table(rd)
            for _ in range(10 if mode == "auto" else 1):      # auto: four timed calls + the decision before the clock
                out = net.forward_table(xyzs, dirs, rd, shq=shq)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                net.forward_table(xyzs, dirs, rd, shq=shq)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if os.environ.get("PROBE_EACH"):
            # every launch on its own clock, back to back: does the time of ONE kernel depend on how long the device has
            # been busy (clock ramp after the host-paced start)?
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
            with torch.no_grad():
                evs[0].record()
                for i in range(40):
                    net.forward_table(xyzs, dirs, rd, shq=shq)
                    evs[i + 1].record()
            torch.cuda.synchronize()
            print("   each of 40 back-to-back launches (ms):", " ".join(f"{evs[i].elapsed_time(evs[i + 1]):.2f}" for i in range(40)))
        same = ""
        if ref is None:
            ref = out
        else:
            same = "  bit-identical to the first mode: " + str(bool(torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])))
        if mode == "auto":
            same += f"  auto chose {'sliced' if (net._slice_probe or {}).get('choice') else 'fused'}"
        print(f"{cfg:22s} {mode:10s} bound {bound} scene x{s:g} dt_gamma {gamma:.5f} levels..{int(tb['resolutions'][-1])} "
              f"hashed {int(tb['hashed'].sum())} T {tb['total_rows']}  M={M}  field {ms:.3f} ms  "
              f"{M / ms / 1e6:.3f} Gsamples/s  frac {M * 1024 / (ms * 1e-3) / 8e12:.4f}{same}", flush=True)
    del net, xyzs, dirs, deltas, rays


for c in configs:
    one(c)
