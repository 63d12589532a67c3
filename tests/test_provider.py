"""NeRFDataset: the on-disk format of the reference's NeRF-stage data (transforms*.json + images [+ matched masks])."""
import json
import os

import numpy as np
import pytest
import torch


def _write_scene(root, n=4, H=12, W=16, with_alpha=True):
    from PIL import Image
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    os.makedirs(os.path.join(root, "masks"), exist_ok=True)
    rng = np.random.default_rng(0)
    frames, imgs, masks = [], [], []
    for i in range(n):
        a = rng.integers(0, 256, size=(H, W, 4 if with_alpha else 3), dtype=np.uint8)
        Image.fromarray(a).save(os.path.join(root, "images", f"{i:04d}.png"))
        m = rng.integers(-1, 7, size=(H, W)).astype(np.int32)
        np.save(os.path.join(root, "masks", f"{i:04d}.npy"), m)
        T = np.eye(4)
        T[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        T[:3, 3] = rng.normal(size=3)
        frames.append({"file_path": f"images/{i:04d}", "transform_matrix": T.tolist()})
        imgs.append(a); masks.append(m)
    meta = {"camera_angle_x": 0.6911, "frames": frames[::-1]}          # unsorted on purpose
    with open(os.path.join(root, "transforms_train.json"), "w") as f:
        json.dump(meta, f)
    return imgs, masks, frames


def test_nerf_dataset_reads_transforms_images_and_masks(tmp_path):
    from instance_nerf_amd.nerf.provider import NeRFDataset, nerf_matrix_to_ngp
    from instance_nerf_amd.nerf.utils import get_rays
    imgs, masks, frames = _write_scene(str(tmp_path))
    ds = NeRFDataset(str(tmp_path), type="train", num_rays=50, mask_dir=str(tmp_path / "masks"), num_instances=5)
    assert len(ds) == 4 and ds.names == ["0000", "0001", "0002", "0003"] and (ds.H, ds.W) == (12, 16)
    fx, fy, cx, cy = ds.intrinsics
    assert abs(fx - 16 / (2 * np.tan(0.6911 / 2))) < 1e-4 and fx == fy and (cx, cy) == (8.0, 6.0)
    # pose convention: axes (y, z, x), camera y/z flipped, translation * 0.33
    T = np.asarray(frames[2]["transform_matrix"], np.float32)
    P = nerf_matrix_to_ngp(T)
    assert np.allclose(ds.poses[2].numpy(), P)
    assert np.allclose(P[:3, 3], T[[1, 2, 0], 3] * 0.33) and np.allclose(P[:3, 0], T[[1, 2, 0], 0])
    assert np.allclose(P[:3, 1], -T[[1, 2, 0], 1]) and abs(np.linalg.det(P[:3, :3]) - 1) < 1e-5
    b = ds[1]
    assert b["rays_o"].shape == (1, 50, 3) and b["images"].shape == (1, 50, 3) and b["masks"].shape == (1, 50)
    # alpha composited on white, labels gathered at the same pixels, ids without a logit become ignore
    a = imgs[1].astype(np.float32) / 255
    want = (a[..., :3] * a[..., 3:] + 1 - a[..., 3:]).reshape(-1, 3)
    torch.manual_seed(3)
    b = ds[1]
    torch.manual_seed(3)
    inds = get_rays(ds.poses[1:2], ds.intrinsics, 12, 16, 50)["inds"][0]
    assert np.allclose(b["images"][0].numpy(), want[inds.numpy()], atol=1e-6)
    lab = masks[1].reshape(-1)[inds.numpy()]
    assert (b["masks"][0].numpy() == np.where(lab >= 5, -1, lab)).all()
    # evaluation split: whole images, row-major
    ev = NeRFDataset(str(tmp_path), type="train", num_rays=50)
    ev.training, ev.num_rays = False, -1
    full = ev[0]
    assert full["rays_d"].shape == (1, 12 * 16, 3) and full["images"].shape == (1, 12, 16, 3)
    assert len(list(iter(ds))) == 4


def test_nerf_dataset_upstream_call_convention(tmp_path):
    """NeRFDataset(opt, device=device, type='train').dataloader() - upstream's main script - reads path / scale /
    num_rays from the namespace; the loader carries the dataset as ``_data`` (poses, intrinsics for the Trainer)."""
    from argparse import Namespace
    from instance_nerf_amd.nerf.provider import NeRFDataset
    _write_scene(str(tmp_path))
    opt = Namespace(path=str(tmp_path), scale=0.5, offset=[0, 0, 0], bound=1, num_rays=33, preload=False, fp16=False)
    a = NeRFDataset(opt, device=torch.device("cpu"), type="train")
    b = NeRFDataset(opt, torch.device("cpu"), "train")                        # positional, upstream's order
    c = NeRFDataset(opt, "cpu")
    for ds in (a, b, c):
        assert ds.type == "train" and ds.device == torch.device("cpu") and ds.num_rays == 33 and len(ds) == 4
    ref = NeRFDataset(str(tmp_path), type="train", scale=0.5, num_rays=33)
    assert torch.equal(a.poses, ref.poses)
    loader = a.dataloader()
    assert loader._data is a and loader.has_gt and loader._data.intrinsics == ref.intrinsics
    assert next(iter(loader))["rays_o"].shape == (1, 33, 3)


def test_nerf_dataset_errors(tmp_path):
    from instance_nerf_amd.nerf.provider import NeRFDataset
    with pytest.raises(FileNotFoundError):
        NeRFDataset(str(tmp_path), type="train")
    _write_scene(str(tmp_path), n=2)
    os.remove(tmp_path / "masks" / "0001.npy")
    with pytest.raises(FileNotFoundError):
        NeRFDataset(str(tmp_path), type="train", mask_dir=str(tmp_path / "masks"))


def test_room_written_to_disk_reads_back_through_the_loader(tmp_path):
    """RoomScene.write_dataset (what bench.py and the GPU tests train from) -> NeRFDataset: the poses survive the
    Blender-convention round trip exactly, pixels are the analytic colours to 8 bits, labels are the instance ids modulo
    the class count with the requested fraction set to -1 (the matched-mask layout of match_seg.py:131-140)."""
    from instance_nerf_amd.nerf.provider import NeRFDataset
    from instance_nerf_amd.scene import RoomScene
    room = RoomScene()
    sc = room.write_dataset(str(tmp_path / "room"), n_views=3, H=24, W=32, num_instances=8, ignore_frac=0.25)
    assert sorted(os.listdir(tmp_path / "room")) == ["images", "matched", "transforms_train.json"]
    ds = NeRFDataset(sc["path"], type="train", device="cpu", scale=1.0, num_rays=256, mask_dir=sc["mask_dir"], num_instances=8)
    assert len(ds) == 3 and (ds.H, ds.W) == (24, 32) and ds.intrinsics == (16.0, 16.0, 16.0, 12.0)
    assert np.abs(ds.poses.numpy() - sc["poses"]).max() == 0.0
    b = ds[2]
    rgb, ids, _ = room.trace(b["rays_o"][0].numpy(), b["rays_d"][0].numpy())
    assert np.abs(rgb - b["images"][0].numpy()).max() <= 0.5 / 255 + 1e-6
    lab = b["masks"][0].numpy()
    keep = lab >= 0
    assert (lab[keep] == (ids % 8)[keep]).all() and 0.1 < 1 - keep.mean() < 0.4
    m = np.load(os.path.join(sc["mask_dir"], "0001.npy"))
    assert m.dtype == np.int32 and m.shape == (24, 32) and m.min() == -1 and m.max() < 8
    assert ((m == -1) | (m == sc["ids"][1] % 8)).all()


def test_every_rank_draws_its_own_views(tmp_path):
    """One process per GPU: NeRFDataset(rank=r) permutes the views with its own stream (the pixel draw's seed moves with it)."""
    from instance_nerf_amd.nerf.provider import NeRFDataset
    _write_scene(str(tmp_path), n=8)
    orders = []
    for r in range(3):
        ds = NeRFDataset(str(tmp_path), type="train", num_rays=16, rank=r, seed=5)
        orders.append([b["index"][0] for b in ds])
        assert sorted(orders[-1]) == list(range(8)) and ds.seed == 5 + 1000 * r
    assert orders[0] != orders[1] and orders[1] != orders[2]
    again = NeRFDataset(str(tmp_path), type="train", num_rays=16, rank=1, seed=5)
    assert [b["index"][0] for b in again] == orders[1]
