"""Would the NEXT step's march (near/far + noise + count + scan + write: independent of the parameters) hide under this
step's backward (scatter: atomic-bound, CUs mostly idle)?  Baseline steps vs the same steps with an EXTRA march of the
next batch queued on a side stream right before the backward: if the extra work is hidden the step time does not move
and a prefetching Trainer would save the inline march."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from instance_nerf_amd import raymarching
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
from instance_nerf_amd.nerf.utils import Trainer, allreduce_gradients

stage = sys.argv[1] if len(sys.argv) > 1 else "instance"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=64 if stage == "instance" else 0).to(dev)
ds = SyntheticRoomDataset(dev, num_rays=4096, num_instances=64)
net.density_bitfield.copy_(torch.from_numpy(ds.room.density_bitfield(128, 1.0)).to(dev))
tr = Trainer("probe", None, net, stage=stage, device=dev, iters=1000, update_extra_interval=10 ** 9, ema_decay=0.95)
tr.global_step = 1
batches = [ds.batch() for _ in range(8)]
peak = 0
for i in range(8):
    tr.train_one_step(batches[i % 8])
    peak = max(peak, int(net.step_counter[(net.local_step - 1) % 16, 0]))
net.mean_count = (int(peak * 1.02) + 127) // 128 * 128
for i in range(8):
    tr.train_one_step(batches[i % 8])
side = torch.cuda.Stream()
params = [p for g in tr.optimizer.param_groups for p in g["params"]]
one = raymarching.unit_gradient(dev)
scratch_counter = torch.zeros(2, dtype=torch.int32, device=dev)


def march(data):
    ro, rd = data["rays_o"].view(-1, 3), data["rays_d"].view(-1, 3)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, net.aabb_train, net.min_near)
    return raymarching.march_rays_train(ro, rd, net.bound, net.density_bitfield, net.cascade, net.grid_size, nears, fars,
                                        scratch_counter, net.mean_count, True, 128, False, 0, 1024)


def run(extra, steps=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        data = batches[i % 8]
        net.train(); tr.global_step += 1
        tr.optimizer.zero_grad()
        _, _, loss = tr.train_step(data)
        if extra == "side":
            with torch.cuda.stream(side):
                keep = march(batches[(i + 1) % 8])
        elif extra == "inline":
            keep = march(batches[(i + 1) % 8])
        loss.backward(gradient=one)
        scale = allreduce_gradients(params, 1, average=False); tr._lr_step()
        tr.optimizer.step_impl(scale)
        tr.ema.update()
        if extra == "side":
            torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for rep in range(2):
    base, inline, sidet = run(None), run("inline"), run("side")
    print(f"{stage}: step {base:.3f} ms | + a second march inline {inline:.3f} (+{(inline - base) * 1e3:.0f} us) | + the same march on a "
          f"side stream before the backward {sidet:.3f} (+{(sidet - base) * 1e3:.0f} us)")
