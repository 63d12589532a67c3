"""Does the table-gradient scatter (bound by the memory-side atomic rate, CUs mostly waiting) tolerate other kernels
beside it?  The next training step's march + frozen-NeRF forward are independent of the current step's backward in
the instance stage: scatter on stream A, march + field forward on stream B, serial vs concurrent."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instance_nerf_amd import _lib, raymarching
from instance_nerf_amd._lib import check, ptr
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset

dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=64).to(dev).eval()
ds = SyntheticRoomDataset(dev, num_rays=4096, num_instances=64)
net.density_bitfield.copy_(torch.from_numpy(ds.room.density_bitfield(128, 1.0)).to(dev))
b = ds.batch()
ro, rd = b["rays_o"].view(-1, 3).contiguous(), b["rays_d"].view(-1, 3).contiguous()
nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_train, net.min_near)
lib = _lib.load()
emb = net.instance_encoder.embeddings.data
g_emb = torch.zeros_like(emb)


def march():
    return raymarching.march_rays_train(ro, rd, net.bound, net.density_bitfield, net.cascade, net.grid_size, nears, fars,
                                        force_all_rays=True)


xyzs, dirs, deltas, rays = march()
M = xyzs.shape[0]
denc = torch.randn(M, 32, device=dev) * 1e-3
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def scatter():
    check(lib.inr_grid_encode_backward(ptr(xyzs), ptr(denc), net.instance_encoder.desc, M, float(net.bound), ptr(g_emb),
                                       torch.cuda.current_stream().cuda_stream), "bwd")


def other():
    with torch.no_grad():
        march()
        net(xyzs, dirs)


def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def concurrent():
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sa):
        scatter()
    with torch.cuda.stream(sb):
        other()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)


print(f"M = {M}")
print(f"scatter alone            {timeit(scatter):7.1f} us")
print(f"march + nerf fwd alone   {timeit(other):7.1f} us")
print(f"serial                   {timeit(lambda: (scatter(), other())):7.1f} us")
print(f"two streams              {timeit(concurrent):7.1f} us")
