"""Which quantity of an instance-stage training step differs first between two identical runs?  (round 6: with the table
gradient summed as int32 the NeRF stage is bit-reproducible, the instance stage drifts apart by 1e-7 after ~17 steps.)
Per step: bit checksums of the loss, of every gradient as the optimiser receives it, and of every parameter after it.
usage: python tools/repro_probe.py [stage=instance] [steps=24]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd.nerf import NeRFNetwork                              # noqa: E402
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset            # noqa: E402
from instance_nerf_amd.nerf.utils import Trainer                            # noqa: E402
from instance_nerf_amd.scene import RoomScene                               # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "instance"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
DEV = torch.device("cuda", 0)
room = RoomScene()


def bits(t):
    return int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum())


def run():
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=16 if stage == "instance" else 0).to(DEV)
    net.density_bitfield.copy_(torch.from_numpy(room.density_bitfield(128, 1.0)).to(DEV))
    ds = SyntheticRoomDataset(DEV, H=200, W=200, n_views=8, num_rays=2048, num_instances=16, seed=4)
    tr = Trainer("repro", None, net, stage=stage, device=DEV, lr=1e-2, iters=200, workspace=None, mute=True,
                 ema_decay=0.95, update_extra_interval=16 if stage == "nerf" else 10 ** 9)
    tr.global_step = 0 if stage == "nerf" else 1
    rec = []
    real = tr.optimizer.step_impl
    names = {id(p): n for n, p in net.named_parameters()}

    def step_impl(scale=1.0):
        cur = {"grad " + names[id(p)]: bits(p.grad) for g in tr.optimizer.param_groups for p in g["params"] if p.grad is not None}
        cur["counter"] = int(net.last_counter[0]) if getattr(net, "last_counter", None) is not None else -1
        rec.append(cur)
        return real(scale)
    tr.optimizer.step_impl = step_impl
    for s in range(steps):
        b = ds.batch()
        loss = tr.train_one_step(b)
        rec[-1]["loss"] = bits(loss.reshape(1))
        tab = net.instance_encoder.embeddings if stage == "instance" else net.encoder.embeddings
        st = getattr(tab, "_fx_state", None)
        if st is not None:
            h = st[:64].cpu().numpy()
            rec[-1]["_fx"] = (int(h[48]), int(h[49]), int((h[:16] == 0).sum()), [round(float(v), 1) for v in (h[32:48] * 0 + h[32:48])[:0]],
                              [f"{h[32 + l] / max(h[16 + l], 1e-30):.2f}" for l in (0, 3, 8, 15)], [f"{h[32 + l]:.1e}" for l in (0, 3, 8, 15)])
        rec[-1]["batch"] = bits(b["rays_d"]) ^ bits(b["masks"].to(torch.int32)) if "masks" in b else bits(b["rays_d"])
        for n, p in net.named_parameters():
            if p.requires_grad:
                rec[-1]["param " + n] = bits(p)
    return rec


a, b = run(), run()
for s, (x, y) in enumerate(zip(a, b)):
    bad = [k for k in x if k != "_fx" and x[k] != y.get(k)]
    fx = x.get("_fx")
    print(f"step {s:2d}: counter {x['counter']}: " + ("identical" if not bad else "DIFFERS in " + ", ".join(bad))
          + (f"   fixed steps {fx[0]}, near misses {fx[1]}, levels on fp32 next step {fx[2]}, max/ref of levels 0,3,8,15: {fx[4]}, max: {fx[5]}" if fx else ""))
