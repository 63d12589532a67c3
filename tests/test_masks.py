"""f3 / f4: mask supervision loader and 3-D mask projector."""
import os

import numpy as np
import pytest
import torch


def test_load_matched_masks_and_labels(tmp_path):
    from instance_nerf_amd.masks import labels_for_rays, load_matched_masks
    m = np.asarray([[0, 1, -1], [2, 70, 3]], dtype=np.int64)
    np.save(tmp_path / "0001.npy", m)
    np.save(tmp_path / "0002.npy", m.T.copy())
    d = load_matched_masks(str(tmp_path))
    assert sorted(d) == ["0001", "0002"] and d["0001"].dtype == np.int32
    lab = labels_for_rays(d["0001"], torch.tensor([0, 1, 2, 3, 4, 5]), num_instances=64)
    assert lab.tolist() == [0, 1, -1, 2, -1, 3]            # id 70 has no logit -> ignored


def test_png_roundtrip(tmp_path):
    from instance_nerf_amd.masks import read_png_gray, save_png_gray
    img = (np.random.default_rng(0).random((13, 21)) > 0.5).astype(np.uint8) * 255
    save_png_gray(str(tmp_path / "a_1.png"), img)
    assert (read_png_gray(str(tmp_path / "a_1.png")) == img).all()


@pytest.mark.gpu
def test_projector_matches_oracle(tmp_path, room, room_bitfield, level_table, params_k16):
    """Soft projections = oracle compositing of the voxel-mask values with the NeRF weights."""
    from instance_nerf_amd.masks import project_3d_masks, read_png_gray, soft_project
    from oracle import composite, field, march, rays as orays
    from test_gpu_parity import _network, _t
    from conftest import scene_rays
    net = _network(params_k16, K=0).eval()
    net.density_bitfield.copy_(_t(room_bitfield))
    res = 20
    occ = room.occupancy_grid(res, 1.0)
    masks = np.zeros((3, res, res, res), np.float32)
    masks[0, :10], masks[1, :, :10], masks[2, :, :, 10:] = occ[:10], occ[:, :10], occ[:, :, 10:]
    ro, rd = scene_rays(room, 200, seed=71)
    soft, ws = soft_project(net, masks, [-1, -1, -1], [1, 1, 1], _t(ro), _t(rd))
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = orays.near_far_from_aabb(ro, rd, aabb, 0.05)
    m = march.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars)
    with torch.no_grad():
        sig, rgb = field.nerf_forward(torch.from_numpy(m["xyzs"]), torch.from_numpy(m["dirs"]), params_k16, 1.0, level_table)
    cell = np.clip(np.floor((m["xyzs"] + 1) / 2 * res).astype(int), 0, res - 1)
    vals = masks[:, cell[:, 0], cell[:, 1], cell[:, 2]].T
    ref = composite.composite_rays_train(sig, rgb, m["deltas"], m["rays"], 1e-4, extra=vals)
    assert np.abs(soft.cpu().numpy() - ref["extra"].numpy()).max() < 1e-4
    # the launch that reads one 32-bit word per voxel (round 4) against the tensor-op version it replaced - a float
    # [M, k] matrix of mask values composited as k extra channels: the same sums in the same order, bit for bit; with
    # 40 masks (two words per voxel), a bounding box inside the volume and rays that leave it
    from instance_nerf_amd import raymarching
    rng = np.random.default_rng(5)
    many = rng.random((40, 7, 9, 11)) > 0.6
    lo, hi = [-0.8, -0.9, -0.7], [0.9, 0.6, 0.8]
    ro_t, rd_t = _t(ro), _t(rd)
    soft40, ws40 = soft_project(net, many, lo, hi, ro_t, rd_t)
    with torch.no_grad():
        nears_t, fars_t = raymarching.near_far_from_aabb(ro_t, rd_t, net.aabb_infer, net.min_near)
        xyzs, dirs, deltas, rays_t = raymarching.march_rays_patch(ro_t, rd_t, net.bound, net.density_bitfield, net.cascade,
                                                                 net.grid_size, nears_t, fars_t, 0, 1024)
        sigmas, rgbs = net(xyzs, dirs)
        mt = torch.from_numpy(many).to(xyzs.device).float()
        rs = torch.tensor(many.shape[1:], device=xyzs.device, dtype=torch.float32)
        lo_t, hi_t = torch.tensor(lo, device=xyzs.device), torch.tensor(hi, device=xyzs.device)
        cell = ((xyzs - lo_t) / (hi_t - lo_t) * rs).floor().long()
        inside = ((cell >= 0) & (cell < rs.long())).all(-1)
        cell = torch.minimum(cell.clamp(min=0), rs.long() - 1)
        chunks = []
        for b in range(0, 40, 20):           # the extra-channel compositing takes at most 64 channels
            vals = mt[b:b + 20, cell[:, 0], cell[:, 1], cell[:, 2]].t().contiguous() * inside[:, None]
            chunks.append(raymarching.composite_rays_patch(sigmas, rgbs, deltas, rays_t, 1e-4, extra=vals)[3])
    assert torch.equal(soft40, torch.cat(chunks, 1)) and float(soft40.max()) > 0.05
    # file output contract: <img>_<inst>.png, ids from 1, channel 0 > 0 = foreground
    poses, intr, H, W = room.cameras(n=1, H=32, W=32, focal=16.0)
    out = project_3d_masks(net, masks, [-1, -1, -1], [1, 1, 1], poses, intr, 32, 32, proj_dir=str(tmp_path), thresh=0.02)
    files = sorted(os.listdir(tmp_path))
    assert files and all(f.startswith("0000_") and not f.endswith("_0.png") for f in files)
    i = int(files[0].split("_")[1].split(".")[0]) - 1
    assert ((read_png_gray(str(tmp_path / files[0])) > 0) == out[0, i]).all()
