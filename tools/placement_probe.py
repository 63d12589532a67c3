"""Is the headline's bimodality (0.80 vs 0.88 of the roofline between calls) a property of WHERE the buffers of a
process land, or of the box at that moment?  One process: the fused field kernel on one frame's samples, timed with
the hash table (and then the sample buffers) moved to fresh allocations behind paddings of different sizes; the
clocks rocm-smi reports are sampled while the kernel loops.  usage: python tools/placement_probe.py [reps]"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import build_network  # noqa: E402
from instance_nerf_amd import raymarching  # noqa: E402
from instance_nerf_amd.nerf.utils import get_rays  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
net, room = build_network(dev)
poses, intr, H, W = room.cameras()
r = get_rays(torch.from_numpy(poses[:1]).to(dev), intr, H, W, patch=4)
ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_infer, net.min_near)
xyzs, dirs, deltas, rays = raymarching.march_rays_patch(ro, rd, 1, net.density_bitfield, 1, 128, nears, fars, table=True)
M = xyzs.shape[0]


def clocks():
    out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout
    got = {}
    for line in out.splitlines():
        for k in ("sclk", "mclk", "fclk", "socclk"):
            if k + " clock" in line and "(" in line:
                got[k] = line.split("(")[1].split(")")[0]
    return got


def timed(label):
    seen = []
    stop = threading.Event()

    def sample():
        while not stop.is_set():
            seen.append(clocks())
            time.sleep(0.05)
    with torch.no_grad():
        net.forward_table(xyzs, dirs, rd)
        torch.cuda.synchronize()
        th = threading.Thread(target=sample)
        th.start()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps * 8):
            net.forward_table(xyzs, dirs, rd)
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
    ms = e0.elapsed_time(e1) / (reps * 8)
    mid = seen[len(seen) // 2] if seen else {}
    print(f"{label:44s} field {ms:.3f} ms = {M * 1024 / ms / 1e6 / 8000:.4f} of 8 TB/s   table @ {net.encoder.embeddings.data_ptr():#x}"
          f"  clocks under load {mid}", flush=True)


timed("as built")
timed("as built, again")
pads = []
for mb in (64, 1, 300, 7, 1500, 33):
    pads.append(torch.empty(mb << 20, dtype=torch.uint8, device=dev))
    net.encoder.embeddings.data = net.encoder.embeddings.data.clone()
    net._packed = {}
    timed(f"table moved behind a {mb} MB padding")
for mb in (5, 900):
    pads.append(torch.empty(mb << 20, dtype=torch.uint8, device=dev))
    xyzs, dirs, rd = xyzs.clone(), dirs.clone(), rd.clone()
    timed(f"sample buffers moved behind a {mb} MB padding")
torch.cuda.empty_cache()
timed("after empty_cache (paddings kept)")
