"""Times the table-gradient scatter per level on the samples of a real training batch (4096 rays of the room)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instance_nerf_amd import _lib, raymarching
from instance_nerf_amd._lib import check, ptr, stream_ptr
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
lib = _lib.load()
dev = torch.device("cuda")
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, num_instances=64).to(dev)
ds = SyntheticRoomDataset(dev, num_rays=4096, num_instances=64)
bits = torch.from_numpy(ds.room.density_bitfield(128, 1.0)).to(dev)
b = ds.batch()
ro, rd = b["rays_o"][0], b["rays_d"][0]
aabb = torch.tensor([-1, -1, -1, 1, 1, 1.0], device=dev)
nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.05)
xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars)
M = xyzs.shape[0]
g = torch.randn(M, 32, device=dev)
enc = net.instance_encoder
gemb = torch.zeros_like(enc.embeddings.data)
levels = [0, 16]
def run():
    check(lib.inr_grid_encode_backward_levels(ptr(xyzs), ptr(g), None, enc.desc, M, 1.0, ptr(gemb), levels[0], levels[1],
                                              stream_ptr()), "bwd")
def timed(n=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("M", M, "all levels", round(timed(), 1), "us")
res = enc.table["resolutions"]
for l in range(16):
    levels[:] = [l, l + 1]
    print(f"level {l:2d} res {int(res[l]):5d}: {timed():7.1f} us")
