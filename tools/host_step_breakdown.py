"""Where the HOST time of an eager training step goes (instance stage, 4096 rays): perf_counter around the phases of
Trainer.train_one_step, no device synchronisation inside the loop (the GPU lags behind; only queueing cost is seen)."""
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
if os.environ.get("O_FLAGS", "0") == "1":      # upstream's -O numerics where a field is only evaluated (frozen NeRF)
    net.half_table = net.mlp_fp16 = True
batches = [ds.batch() for _ in range(8)]
peak = 0
for i in range(8):
    tr.train_one_step(batches[i % 8])
    peak = max(peak, int(net.step_counter[(net.local_step - 1) % 16, 0]))
net.mean_count = (int(peak * 1.02) + 127) // 128 * 128
for i in range(8):
    tr.train_one_step(batches[i % 8])
torch.cuda.synchronize()
acc = {}
def lap(name, t):
    now = time.perf_counter(); acc[name] = acc.get(name, 0.0) + now - t; return now
steps = 300
params = [p for g in tr.optimizer.param_groups for p in g["params"]]
one = raymarching.unit_gradient(dev)
t_all = time.perf_counter()
for i in range(steps):
    data = batches[i % 8]
    t = time.perf_counter()
    net.train(); tr.global_step += 1
    tr.optimizer.zero_grad(); t = lap("zero_grad", t)
    _, _, loss = tr.train_step(data); t = lap("forward (render + loss)", t)
    loss.backward(gradient=one); t = lap("backward", t)
    scale = allreduce_gradients(params, 1, average=False); tr._lr_step(); t = lap("allreduce(noop) + lr", t)
    tr.optimizer.step_impl(scale); t = lap("optimizer.step_impl", t)
    tr.ema.update(); t = lap("ema.update", t)
host = time.perf_counter() - t_all
torch.cuda.synchronize()
total = time.perf_counter() - t_all
print(f"{stage}: host {host / steps * 1e3:.3f} ms per step, wall {total / steps * 1e3:.3f} ms per step")
for k, v in acc.items():
    print(f"  {k:28s} {v / steps * 1e6:7.1f} us")
