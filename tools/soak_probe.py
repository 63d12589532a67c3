"""Long-run stability of the product's training loops (round 6): the synthetic room from disk, NeRF stage for `n_nerf` steps
through Trainer.train (eager: epochs, occupancy updates, no checkpoints), then the instance stage for `n_inst` steps as the
captured two-stream pipeline (use_graph + look_ahead).  Reports steps/s, NaN steps, graph captures, peak memory, held-out
PSNR / mIoU and - with INR_FX_GRAD=1 - the fixed-point counters.  usage: python tools/soak_probe.py [n_nerf=20000] [n_inst=10000]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd.nerf import NeRFNetwork, network                     # noqa: E402
from instance_nerf_amd.nerf.provider import NeRFDataset                     # noqa: E402
from instance_nerf_amd.nerf.utils import MIoUMeter, Trainer, get_rays       # noqa: E402
from instance_nerf_amd.scene import RoomScene                               # noqa: E402

n_nerf = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_inst = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda", 0)
K = 16
room = RoomScene()
d = tempfile.mkdtemp(prefix="inr_soak_")
scene = room.write_dataset(d, n_views=24, H=400, W=400, num_instances=K, ignore_frac=0.1)
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=K).to(dev)
held = torch.from_numpy(room.look_at([0.3, -0.2, 0.1])[None]).to(dev)


def held_out():
    net.eval()
    r = get_rays(held, (200.0, 200.0, 200.0, 200.0), 400, 400, patch=4)
    rgb, ids, _ = room.trace(r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy())
    with torch.no_grad():
        out = net.render(r["rays_o"], r["rays_d"], bg_color=1)
    mse = float(((out["image"][0] - torch.from_numpy(rgb).to(dev)) ** 2).mean())
    m = MIoUMeter(K)
    m.update(out["instance"][0].argmax(-1).cpu(), torch.from_numpy(ids % K))
    return -10 * np.log10(mse), m.measure_both()["miou_gt_ids"]


def counters(table):
    st = getattr(table, "_fx_state", None)
    if st is None or not network.FX_GRAD:
        return ""
    h = st[:96].cpu().numpy()
    return f", int{network.fx_bits()} sums: {int(h[48])} steps, {int(h[49])} near misses, peak use of the integer range {h[80:96].max():.3f}"


ds = NeRFDataset(d, type="train", device=dev, scale=1.0, num_rays=4096, preload=True)
tr = Trainer("soak_nerf", None, net, stage="nerf", device=dev, lr=1e-2, iters=n_nerf, workspace=None, mute=True, ema_decay=0.95)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.train(ds.dataloader(), max_epochs=-(-n_nerf // len(ds)))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
psnr, _ = held_out()
print(f"NeRF stage: {tr.global_step} steps in {dt:.1f} s = {tr.global_step / dt:.0f} steps/s (eager, EMA 0.95, {tr.epoch} epochs), "
      f"loss {tr.stats['loss'][0]:.5f} -> {tr.stats['loss'][-1]:.6f}, NaN steps {sum(tr.stats['nan_steps'])}, mean_count {net.mean_count}, "
      f"held-out PSNR {psnr:.2f} dB, parameters finite {all(bool(torch.isfinite(p).all()) for p in net.parameters())}"
      + counters(net.encoder.embeddings))
ds2 = NeRFDataset(d, type="train", device=dev, scale=1.0, num_rays=4096, preload=True, mask_dir=scene["mask_dir"], num_instances=K)
net.mean_density = net.mean_density
ti = Trainer("soak_inst", None, net, stage="instance", device=dev, lr=1e-2, iters=n_inst, workspace=None, mute=True, ema_decay=0.95,
             update_extra_interval=10 ** 9, use_graph=True, look_ahead=True)
ti.global_step = 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(-(-n_inst // len(ds2))):
    ti.train_one_epoch(ds2.dataloader())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
_, miou = held_out()
print(f"instance stage: {ti.global_step - 1} steps in {dt:.1f} s = {(ti.global_step - 1) / dt:.0f} steps/s (captured two-stream pipeline, "
      f"{len(ti._pipe['graphs']) if ti._pipe else 0} graphs), CE {ti.stats['loss'][0]:.4f} -> {ti.stats['loss'][-1]:.5f}, "
      f"NaN steps {sum(ti.stats['nan_steps'])}, held-out mIoU {miou:.3f}, parameters finite "
      f"{all(bool(torch.isfinite(p).all()) for p in net.parameters())}, peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.2f} GiB"
      + counters(net.instance_encoder.embeddings))
