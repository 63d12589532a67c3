"""Runs only the fused field kernel on one frame's worth of marched samples (profiling target).
usage: python tools/field_probe.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import build_network  # noqa: E402
from instance_nerf_amd import raymarching  # noqa: E402
from instance_nerf_amd.nerf.utils import get_rays  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
net, room = build_network(dev)
if os.environ.get("PROBE_LOG2_T"):        # smaller hash tables: where does the time go when every level fits the L2?
    from instance_nerf_amd.encoding import get_encoder
    net.encoder, _ = get_encoder("hashgrid", desired_resolution=2048, log2_hashmap_size=int(os.environ["PROBE_LOG2_T"]))
    net.encoder.to(dev)
    net._packed = {}
    print("table rows", net.encoder.table["total_rows"], "MB", net.encoder.table["total_rows"] * 8 / 1e6)
poses, intr, H, W = room.cameras()
PATCH = int(os.environ.get("PROBE_PATCH", "4"))
r = get_rays(torch.from_numpy(poses[:1]).to(dev), intr, H, W, patch=PATCH)
ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_infer, net.min_near)
if os.environ.get("PROBE_RAYMAJOR"):
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1, net.density_bitfield, 1, 128, nears, fars,
                                                            force_all_rays=True)
else:
    xyzs, dirs, deltas, rays = raymarching.march_rays_patch(ro, rd, 1, net.density_bitfield, 1, 128, nears, fars,
                                                            table=bool(os.environ.get("PROBE_TABLE")))
M = xyzs.shape[0]
run = (lambda: net.forward_table(xyzs, dirs, rd)) if os.environ.get("PROBE_TABLE") else (lambda: net(xyzs, dirs))
with torch.no_grad():
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"M={M} field {dt*1e3:.3f} ms  {M/dt/1e9:.3f} Gsamples/s  {M*1024/dt/1e9:.1f} GB/s algorithmic")
