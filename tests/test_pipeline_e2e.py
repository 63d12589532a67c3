"""The product claim, whole (round-4 verdict item 4): one scene through EVERY file boundary of the reference's pipeline,
in sequence, each file written in the reference's format and read back from disk by the next stage:

  transforms.json + images  --NeRFDataset-->  NeRF stage (Trainer)
      --extract_rgbsigma / write_features_npz-->  features/<scene>.npz     (read back through the reference's consumer
                                                                              contract, oracle/consumers.load_feature)
      --[NeRF-RCNN, out of scope: analytic boxes stand in]-->  masks/<scene>.npz   (run_rcnn.py:652-666 layout)
      --load_3d_masks / project_3d_masks-->  proj/<img>_<inst>.png
      --[Mask2Former + match_seg.py: the analytic 2-D segments and the oracle's restatement of the matching rule,
         pinned to the reference's own run by tests/test_match_seg_oracle.py]-->  matched/<img>.npy
      --NeRFDataset(mask_dir)-->  instance stage (Trainer)  -->  rendered instance ids on a held-out pose.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_one_scene_through_every_file_boundary(tmp_path, room):
    from PIL import Image
    from instance_nerf_amd import extract, masks as pmasks
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import NeRFDataset
    from instance_nerf_amd.nerf.utils import MIoUMeter, Trainer, get_rays
    from oracle import consumers, rays as orays
    rng = np.random.default_rng(0)
    H = W = 200
    n_views, K = 32, 16
    scene = tmp_path / "scene"
    os.makedirs(scene / "images")
    poses, intr, _, _ = room.cameras(n=n_views, H=H, W=W, focal=W / 2.0)
    names, frames, truth_ids = [], [], {}
    for i, P in enumerate(poses):
        r = orays.get_rays(P[None], intr, H, W)
        rgb, ids, _ = room.trace(r["rays_o"][0], r["rays_d"][0])
        name = f"{i:04d}"
        names.append(name)
        truth_ids[name] = ids.reshape(H, W)
        Image.fromarray((rgb.reshape(H, W, 3) * 255).astype(np.uint8)).save(scene / "images" / f"{name}.png")
        T = np.eye(4, dtype=np.float32)          # the file stores the Blender-convention matrix (inverse of nerf_matrix_to_ngp)
        T[[1, 2, 0], 0], T[[1, 2, 0], 1], T[[1, 2, 0], 2], T[[1, 2, 0], 3] = P[:3, 0], -P[:3, 1], -P[:3, 2], P[:3, 3]
        frames.append({"file_path": f"images/{name}.png", "transform_matrix": T.tolist()})
    with open(scene / "transforms_train.json", "w") as f:
        json.dump({"fl_x": W / 2.0, "fl_y": W / 2.0, "cx": W / 2.0, "cy": H / 2.0, "w": W, "h": H, "frames": frames}, f)

    # ---- stage 1: the NeRF, from the files
    ds = NeRFDataset(str(scene), type="train", device=DEV, scale=1.0, num_rays=4096)
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=K).to(DEV)
    tr = Trainer("e2e_nerf", None, net, stage="nerf", device=torch.device(DEV), lr=1e-2, iters=1500)
    it = iter(())
    losses = []
    for step in range(2000):      # (1500 + 1500 steps left the held-out mIoU at 0.75-0.86 from run to run; with 500 / 1500 more
        #                           steps at the schedule's final learning rate - lr x 0.1 from step 1500 on - 0.90-0.94)
        try:
            batch = next(it)
        except StopIteration:
            it = iter(ds)
            batch = next(it)
        losses.append(tr.train_one_step(batch).detach().float().reshape(()))
    first, last = float(torch.stack(losses[:20]).mean()), float(torch.stack(losses[-100:]).mean())
    assert last < 0.25 * first, (first, last)

    # ---- boundary 1: features/<scene>.npz, read back the way the reference's loader reads it
    os.makedirs(tmp_path / "features")
    grid, res = extract.extract_rgbsigma(net, max_side=64)
    fpath = extract.write_features_npz(str(tmp_path / "features" / "scene.npz"), grid, [-1, -1, -1], [1, 1, 1])
    feat = consumers.load_feature(np.load(fpath))                    # datasets.py:766-792 restated, pinned by its own fixture
    assert feat.shape == (4, 64, 64, 64) and 0.0 <= feat[3].min() and feat[3].max() <= 1.0
    inside = np.zeros((64, 64, 64), bool)
    centres = (np.arange(64) + 0.5) / 64 * 2 - 1
    gx, gy, gz = np.meshgrid(centres, centres, centres, indexing="ij")
    box_masks = []
    for lo, hi in zip(room.lo, room.hi):
        m = (gx >= lo[0]) & (gx <= hi[0]) & (gy >= lo[1]) & (gy <= hi[1]) & (gz >= lo[2]) & (gz <= hi[2])
        box_masks.append(m)
        inside |= m
    free = (np.abs(gx) < 0.85) & (np.abs(gy) < 0.85) & (np.abs(gz) < 0.85) & ~inside
    shell = inside & ~(np.roll(inside, 1, 0) & np.roll(inside, -1, 0) & np.roll(inside, 1, 1) & np.roll(inside, -1, 1)
                       & np.roll(inside, 1, 2) & np.roll(inside, -1, 2))
    assert feat[3][shell].mean() > 2 * feat[3][free].mean()          # the boxes' surfaces are where the trained density is

    # ---- boundary 2: masks/<scene>.npz in run_rcnn.py:652-666's layout (the detector is out of scope: analytic boxes)
    os.makedirs(tmp_path / "masks")
    k = len(box_masks)
    boxes = np.stack([np.concatenate([(lo + 1) / 2 * 64, (hi + 1) / 2 * 64]) for lo, hi in zip(room.lo, room.hi)]).astype(np.float32)
    np.savez(tmp_path / "masks" / "scene.npz", masks=np.stack(box_masks), scores=np.linspace(0.99, 0.6, k).astype(np.float32),
             labels=np.ones(k, np.int64), boxes=boxes)
    m3 = pmasks.load_3d_masks(str(tmp_path / "masks" / "scene.npz"))
    assert m3["masks"].shape == (k, 64, 64, 64)

    # ---- boundary 3: proj/<img>_<inst>.png
    proj_dir = tmp_path / "proj"
    pmasks.project_3d_masks(net, m3["masks"], [-1, -1, -1], [1, 1, 1], poses, intr, H, W, proj_dir=str(proj_dir), img_names=names)
    proj_files = sorted(os.listdir(proj_dir))
    assert len(proj_files) > 3 * n_views and not any(f.endswith("_0.png") for f in proj_files)

    # ---- boundary 4: matched/<img>.npy - 2-D segments with arbitrary panoptic ids (what Mask2Former would hand over) are
    # renamed to the 3-D instances by the reference's matching rule
    matched = tmp_path / "matched"
    os.makedirs(matched)
    agree = []
    for name in names:
        ids = truth_ids[name]
        pan = np.where(ids > 0, (ids * 7 + 3) % 97 + 1, 200).astype(np.int32)        # walls: one stuff segment, id 200
        pan[rng.random(pan.shape) < 0.03] = 0                                         # a few unlabeled pixels
        info = [{"id": int(s), "isthing": s != 200, "name": "wall-other-merged" if s == 200 else "chair"}
                for s in np.unique(pan) if s > 0]
        files, inst = consumers.projections_of(proj_files, name)
        pm = [pmasks.read_png_gray(str(proj_dir / f)) > 0 for f in files]
        out = consumers.match_seg(consumers.convert_seg(pan, info), pm, inst)
        np.save(matched / f"{name}.npy", out)
        lab = out >= 0
        if lab.any():                       # (a camera inside a box sees one segment that no projection explains: all -1)
            agree.append((out[lab] == ids[lab]).mean())
    # the chain so far reproduces the scene's instance ids (mask i = box i = id i + 1) on nearly every view
    assert len(agree) >= n_views - 3 and np.mean(agree) > 0.97

    # ---- stage 2: the instance field, supervised from the matched files
    ds2 = NeRFDataset(str(scene), type="train", device=DEV, scale=1.0, num_rays=4096, mask_dir=str(matched), num_instances=K)
    net.mean_density = net.mean_density      # (forces the pending device value to the host before the stage changes)
    ti = Trainer("e2e_inst", None, net, stage="instance", device=torch.device(DEV), lr=1e-2, iters=1500,
                 update_extra_interval=10 ** 9)
    ti.global_step = 1
    it = iter(())
    ce, kept = [], []
    for step in range(3000):
        try:
            batch = next(it)
        except StopIteration:
            it = iter(ds2)
            batch = next(it)
        ce.append(ti.train_one_step(batch).detach().float().reshape(()))
        kept.append((batch["masks"] >= 0).sum().reshape(()))
    ce, kept = torch.stack(ce), torch.stack(kept)
    # a view whose every pixel is unmatched (the camera inside a box, see above) gives batches without a single labelled
    # ray: their cross entropy is the mean over an empty set - NaN, as torch's - and, all rays being pruned, they touch no
    # parameter.  Exactly those steps are NaN; the statistics below are over the others.
    assert bool((torch.isnan(ce) == (kept == 0)).all()) and float((kept == 0).float().mean()) < 0.15
    assert all(bool(torch.isfinite(q).all()) for q in net.parameters())
    fin = torch.isfinite(ce)
    first, last = float(ce[:40][fin[:40]].mean()), float(ce[-200:][fin[-200:]].mean())
    assert last < 0.5 * first, (first, last)

    # ---- the claim: rendered instance ids on a pose that no stage has seen
    net.eval()
    held = torch.from_numpy(room.look_at([0.3, -0.2, 0.1])[None]).to(DEV)
    rh = get_rays(held, intr, H, W, patch=4)
    _, gt, _ = room.trace(rh["rays_o"][0].cpu().numpy(), rh["rays_d"][0].cpu().numpy())
    with torch.no_grad():
        pred = net.render(rh["rays_o"], rh["rays_d"], bg_color=1)["instance"][0].argmax(-1).cpu()
    meter = MIoUMeter(K)
    meter.update(pred, torch.from_numpy(gt))
    both = meter.measure_both()
    acc = float((pred == torch.from_numpy(gt)).float().mean())
    assert both["miou_gt_ids"] >= 0.8 and acc >= 0.9, (both, acc)
