"""Two-rank data parallelism simulated in ONE process (two backward passes, gradients averaged, one optimiser step)
against one step on the union batch: how far apart do the parameters end up after three Adam steps, with the fused
NeRF head and with the round-2 chain?  (Adam with eps 1e-15 turns a gradient that is rounding noise into a full step.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_ddp import _batch, _make

for fused in (True, False):
    res = {}
    for mode in ("split", "union"):
        room, net, tr = _make("nerf", 1, 0)
        net.fused_nerf_head = fused
        for s in range(3):
            parts = [_batch(room, "nerf", r, s) for r in range(2)]
            if mode == "union":
                tr.train_one_step({k: torch.cat([p[k] for p in parts], 1) for k in parts[0]})
            else:
                tr.optimizer.zero_grad()
                tr.global_step += 1
                acc = None
                for p in parts:
                    _, _, loss = tr.train_step(p)
                    loss.backward()
                tr._lr_step()
                tr.optimizer.step(grad_scale=0.5)
        res[mode] = {k: v.detach().cpu().numpy().copy() for k, v in net.named_parameters() if v.requires_grad}
    for k in res["split"]:
        d = np.abs(res["split"][k] - res["union"][k]).ravel()
        print(f"fused_head={fused} {k:24s} max {d.max():.2e}  >2e-3: {int((d > 2e-3).sum())}  >5e-4: {int((d > 5e-4).sum())}  of {d.size}")
