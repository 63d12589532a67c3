"""Steady-state training step: eager launches vs one captured hipGraph (Trainer(use_graph=True))."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
from instance_nerf_amd.nerf.utils import Trainer

dev = torch.device("cuda", 0)
for stage in ("instance", "nerf"):
    for use_graph in (False, True):
        torch.manual_seed(0)
        net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10,
                          num_instances=64 if stage == "instance" else 0).to(dev)
        ds = SyntheticRoomDataset(dev, num_rays=4096, num_instances=64)
        net.density_bitfield.copy_(torch.from_numpy(ds.room.density_bitfield(128, 1.0)).to(dev))
        tr = Trainer("g", None, net, stage=stage, device=dev, iters=1000, update_extra_interval=10 ** 9, use_graph=use_graph)
        tr.global_step = 1
        batches = [ds.batch() for _ in range(4)]
        tot = []
        for i in range(5):
            tr.train_one_step(batches[i % 4]); tot.append(int(net.step_counter[(net.local_step - 1) % 16, 0]))
        net.mean_count = (int(sum(tot) / len(tot)) + 16383) // 16384 * 16384
        losses = [float(tr.train_one_step(batches[i % 4])) for i in range(4)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(40):
            l = tr.train_one_step(batches[i % 4])
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("   first losses", [round(x, 5) for x in losses])
        print(f"{stage:8s} graph={use_graph}: {(t2 - t0) / 40 * 1e3:.3f} ms/step (host issue {(t1 - t0) / 40 * 1e3:.3f}), "
              f"loss {losses[0]:.4f} -> {float(l):.4f}, adam steps {tr.optimizer.step_count}")
