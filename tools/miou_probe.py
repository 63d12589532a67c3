"""Why is the held-out instance mIoU of bench.py's trained scene low (0.355 after 400 steps in round 3)?
Trains the room's NeRF (1500 steps, learned occupancy grid) with the product's Trainer, then the K = 16 instance field on
the frozen NeRF for INST_STEPS steps over all training views, and reports for a TRAINING view and the HELD-OUT pose:
pixel accuracy, mIoU (the product's MIoUMeter: classes present in truth or prediction), per-class IoU with the
ground-truth pixel counts, and the confusion of the worst classes.  python tools/miou_probe.py [inst_steps]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd.nerf import NeRFNetwork                              # noqa: E402
from instance_nerf_amd.nerf.provider import SyntheticRoomDataset           # noqa: E402
from instance_nerf_amd.nerf.utils import MIoUMeter, Trainer, get_rays       # noqa: E402

dev = torch.device("cuda", 0)
K = 16
inst_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
torch.manual_seed(0)
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=K).to(dev)
ds = SyntheticRoomDataset(dev, H=400, W=400, n_views=24, num_rays=4096, num_instances=K, ignore_frac=0.1)
tr = Trainer("nerf", None, net, stage="nerf", device=dev, lr=1e-2, iters=1500, workspace=None, mute=True)
for _ in range(1500):
    tr.train_one_step(ds.batch())
tr2 = Trainer("inst", None, net, stage="instance", device=dev, lr=1e-2, iters=inst_steps, update_extra_interval=10 ** 9,
              workspace=None, mute=True)
tr2.global_step = 1
ce = []
for i in range(inst_steps):
    l = tr2.train_one_step(ds.batch())
    if i % max(inst_steps // 10, 1) == 0 or i == inst_steps - 1:
        ce.append(round(float(l), 4))
net.eval()


def report(pose, name):
    r = get_rays(pose, ds.intrinsics, ds.H, ds.W, patch=4)
    gt, ids, _ = ds.room.trace(r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy())
    with torch.no_grad():
        out = net.render(r["rays_o"], r["rays_d"], bg_color=1)
    pred = out["instance"][0].argmax(-1).cpu().numpy()
    truth = ids % K
    m = MIoUMeter(K)
    m.update(torch.from_numpy(pred), torch.from_numpy(truth))
    per = {}
    for c in range(K):
        p, t = pred == c, truth == c
        u = int((p | t).sum())
        if u:
            per[c] = {"iou": round(float((p & t).sum() / u), 3), "truth_pixels": int(t.sum()), "pred_pixels": int(p.sum())}
    psnr = -10 * np.log10(float(((out["image"][0].cpu().numpy() - gt) ** 2).mean()))
    opac = float(out["weights_sum"].mean())
    worst = sorted(per, key=lambda c: per[c]["iou"])[:4]
    conf = {int(c): {int(k): int(((truth == c) & (pred == k)).sum()) for k in np.unique(pred[truth == c])} for c in worst
            if per[c]["truth_pixels"]}
    return {"view": name, "pixel_accuracy": round(float((pred == truth).mean()), 4), "miou": round(m.measure(), 4),
            "miou_over_truth_classes_only": round(float(np.mean([v["iou"] for v in per.values() if v["truth_pixels"]])), 4),
            "psnr_db": round(psnr, 2), "mean_opacity": round(opac, 3), "per_class": per,
            "where_the_worst_classes_go (truth -> {pred: pixels})": conf}


res = {"instance_steps": inst_steps, "ce_curve": ce,
       "training_view_0": report(ds.poses[:1] if hasattr(ds, "poses") else None, "training view 0"),
       "held_out": report(torch.from_numpy(ds.room.look_at([0.3, -0.2, 0.1])[None]).to(dev), "held-out pose")}
print(json.dumps(res))
