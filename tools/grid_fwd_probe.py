"""Times the stand-alone hash-grid forward (inr_grid_encode_forward) and the fused instance forward on the samples
of a real training batch (4096 random rays of the room)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instance_nerf_amd import raymarching
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
dev = torch.device("cuda")
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, num_instances=64).to(dev).eval()
ds = SyntheticRoomDataset(dev, num_rays=4096, num_instances=64)
bits = torch.from_numpy(ds.room.density_bitfield(128, 1.0)).to(dev)
b = ds.batch()
ro, rd = b["rays_o"][0], b["rays_d"][0]
nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_train, 0.05)
xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 1.0, bits, 1, 128, nears, fars, force_all_rays=True)
M = xyzs.shape[0]
flush = torch.empty(96 << 20, dtype=torch.float32, device=dev)       # 384 MB: evicts the table from L2 and most of the MALL
def timed(fn, n=20):
    ts = []
    for _ in range(n):
        flush.add_(1.0)                       # something else runs between two uses of the table, as in a training step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
with torch.no_grad():
    print("M", M)
    print(f"grid_encode_forward (stand-alone, [M,32] out): {timed(lambda: net.instance_encoder(xyzs, bound=1)):7.1f} us")
    print(f"fused instance forward (gather + MLP)        : {timed(lambda: net.instance(xyzs)):7.1f} us")
    print(f"fused NeRF forward                           : {timed(lambda: net(xyzs, dirs)):7.1f} us")
