"""Mask plumbing either side of the instance-field trainer.

f3  ``load_matched_masks``: reads what /root/reference/Mask2Former_sample/match_seg.py:140 writes -
    one ``<img>.npy`` per view, int32 [H, W], -1 = ignore / unmatched, 0 = background, > 0 = instance id
    (match_seg.py:65-91,131-138) - and turns it into per-ray labels for the CE loss (ignore_index -1).
f4  ``project_3d_masks``: counterpart of the reference's ``scripts/project_3d_masks.py`` (in the
    un-vendored submodule, /root/reference/README.md:63).  Input: NeRF-RCNN's discrete 3-D masks
    (/root/reference/nerf_rcnn/run_rcnn.py:652-666: ``masks [k, W, L, H]``); output:
    ``<proj_dir>/<img>_<inst>.png`` with foreground = channel 0 > 0 and instance ids starting at 1
    (match_seg.py:55-62,99-102: files ``*_0.png`` are skipped by the consumer).  A pixel belongs to
    instance i when the NeRF-weighted mask value along its ray, sum_s w_s * mask_i(x_s), exceeds
    ``thresh``: the K-channel compositing of the render path with mask_i as the extra channel.
"""
import os
import struct
import zlib

import numpy as np
import torch

from . import raymarching
from .nerf.utils import get_rays


# ---------------------------------------------------------------------------------------- f3
HDF5_DATASET = "cp_instance_id_segmaps"       # match_seg.py:142-143


def load_matched_masks(seg_dir, names=None):
    """-> dict {image name: int32 [H, W]} from ``<seg_dir>/<name>.npy`` (match_seg.py:140) and, for images that only
    have it, from the ``<name>.hdf5`` mirror the reference writes beside it (match_seg.py:142-143: dataset
    ``cp_instance_id_segmaps``; read without h5py by ``hdf5_lite``).  When both exist the ``.npy`` file is read."""
    from . import hdf5_lite
    listing = sorted(os.listdir(seg_dir))
    stems = {f[:-4]: f for f in listing if f.endswith(".npy")}
    for f in listing:
        if f.endswith(".hdf5") and f[:-5] not in stems:
            stems[f[:-5]] = f
    if names is not None:
        stems = {k: v for k, v in stems.items() if k in set(names)}
    out = {}
    for stem in sorted(stems):
        f = stems[stem]
        path = os.path.join(seg_dir, f)
        m = np.load(path) if f.endswith(".npy") else hdf5_lite.read_dataset(path, HDF5_DATASET)
        if m.ndim != 2:
            raise ValueError(f"{f}: expected an [H, W] instance-id map, got shape {m.shape}")
        out[stem] = m.astype(np.int32)
    return out


def labels_for_rays(mask, inds, num_instances):
    """mask int32 [H, W], inds int64 [N] flat pixel indices -> int64 [N] CE targets (-1 stays ignore;
    ids >= num_instances are ignored too, they have no logit)."""
    m = torch.as_tensor(mask).reshape(-1)
    inds = torch.as_tensor(inds)
    # the gather runs where the mask lives (a loader keeps its masks on the GPU: no host round trip per batch -
    # round-3 verdict: `inds.cpu()` here put a synchronisation into every training step)
    lab = m[inds.to(m.device)].long()
    return torch.where(lab >= num_instances, torch.full_like(lab, -1), lab)


# ---------------------------------------------------------------------------------------- png
def save_png_gray(path, img):
    """Minimal 8-bit grayscale PNG writer (no imaging library in the image); ``cv2.imread`` decodes it to
    three equal channels, so ``img[:, :, 0] > 0`` (match_seg.py:59) sees the mask."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0))
           + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    with open(path, "wb") as f:
        f.write(png)


def read_png_gray(path):
    """Decoder for files written by ``save_png_gray`` (tests)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    return np.frombuffer(raw, np.uint8).reshape(h, w + 1)[:, 1:].copy()


# ---------------------------------------------------------------------------------------- f4
def load_3d_masks(path):
    """Reads ``masks/<scene>.npz`` as /root/reference/nerf_rcnn/run_rcnn.py:652-666 writes it: ``masks`` bool
    [k, W, L, H] (the output of the reference's ``paste_masks_in_image``, model/utils.py:704-782, thresholded at 0.5),
    ``scores`` float [k] sorted DESCENDING (top ``save_top_k`` detections, run_rcnn.py:658-664), ``labels`` int [k],
    ``boxes`` float [k, 6] (x1, y1, z1, x2, y2, z2 in grid units).  Instance id of mask i is i + 1
    (match_seg.py:99-102 skips ``*_0.png``).  -> dict of numpy arrays with those four keys."""
    z = np.load(path)
    missing = {"masks", "scores", "labels", "boxes"} - set(z.files)
    if missing:
        raise ValueError(f"{path}: missing keys {sorted(missing)} (expected the layout of run_rcnn.py:665-666)")
    masks, scores, labels, boxes = z["masks"], z["scores"], z["labels"], z["boxes"]
    k = masks.shape[0]
    if masks.ndim != 4 or scores.shape != (k,) or labels.shape != (k,) or boxes.shape != (k, 6):
        raise ValueError(f"{path}: masks {masks.shape}, scores {scores.shape}, labels {labels.shape}, boxes {boxes.shape} "
                         "are not [k,W,L,H], [k], [k], [k,6]")
    return {"masks": masks.astype(bool), "scores": scores.astype(np.float32), "labels": labels.astype(np.int64),
            "boxes": boxes.astype(np.float32)}


def pack_mask_words(masks, device):
    """bool [k, W, L, H] -> list of int32 tensors [W, L, H], one per 32 masks: bit i of word j = mask 32 j + i contains
    the voxel (the layout ``inr_project_masks_patch`` reads; int32 is the storage type, the kernel reads uint32)."""
    m = torch.as_tensor(masks).to(device).bool()
    out = []
    for base in range(0, m.shape[0], 32):
        chunk = m[base:base + 32].to(torch.int64)
        shifts = torch.arange(chunk.shape[0], device=device, dtype=torch.int64).view(-1, 1, 1, 1)
        w = (chunk << shifts).sum(0)
        w = torch.where(w >= 2 ** 31, w - 2 ** 32, w)                 # the same 32 bits as a signed value
        out.append(w.to(torch.int32).contiguous())
    return out


@torch.no_grad()
def soft_project(model, masks, bbox_min, bbox_max, rays_o, rays_d, T_thresh=1e-4, dt_gamma=0, max_steps=1024, packed=None):
    """NeRF-weighted projection of k voxel masks along rays.  masks [k, W, L, H] (boolean: non-zero = inside, as
    ``run_rcnn.py:652-666`` writes them); rays [N, 3].  -> (soft [N, k] = sum_s w_s * mask(x_s), weights_sum [N]).
    ``packed``: ``(k, pack_mask_words(masks, device))`` from an earlier call (a caller projecting many views packs once).

    March (patch-interleaved frame layout) -> field -> compositing with the per-sample weights kept -> ONE launch per
    32 masks that walks every ray's samples, looks the sample's voxel up in a 32-bit word per voxel and adds the weight
    to the accumulators of the masks whose bit is set (``inr_project_masks_patch``; round 4 - until then the mask values
    of all samples were gathered into a float [M, k] matrix, 3.8 GB for an 800x800 frame and 30 masks, and composited as
    k extra channels: same sums in the same order, bit-identical)."""
    from . import _lib
    from ._lib import check, ptr, stream_ptr
    dev = rays_o.device
    k, words = packed if packed is not None else (len(masks), pack_mask_words(masks, dev))
    if k == 0:
        raise ValueError("soft_project: no masks")
    W, L, H = (int(v) for v in words[0].shape)
    bbox = torch.tensor([float(v) for v in np.asarray(bbox_min, dtype=np.float32)] +
                        [float(v) for v in np.asarray(bbox_max, dtype=np.float32)], dtype=torch.float32)
    nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, model.aabb_infer, model.min_near)
    xyzs, dirs, deltas, rays = raymarching.march_rays_patch(rays_o, rays_d, model.bound, model.density_bitfield,
                                                           model.cascade, model.grid_size, nears, fars, dt_gamma, max_steps)
    sigmas, rgbs = model(xyzs, dirs)
    sigmas = sigmas * model.density_scale
    ws, _, _, wbuf = raymarching.composite_rays_patch(sigmas, rgbs, deltas, rays, T_thresh, return_weights=True)
    N, M = rays.shape[0], xyzs.shape[0]
    soft = torch.empty(N, k, dtype=torch.float32, device=dev)
    lib = _lib.load()
    for j, wj in enumerate(words):
        base = 32 * j
        check(lib.inr_project_masks_patch(ptr(xyzs, torch.float32, "xyzs", allow_none=M == 0),
                                          ptr(wbuf, torch.float32, "weights", allow_none=M == 0),
                                          ptr(rays, torch.int32, "rays", allow_none=N == 0), N, M,
                                          ptr(wj, torch.int32, "mask_words"), W, L, H,
                                          _lib.host_ptr(bbox, torch.float32, "bbox"), k, base, min(32, k - base),
                                          ptr(soft, allow_none=N == 0), stream_ptr()), "project_masks_patch")
    return soft, ws


@torch.no_grad()
def project_3d_masks(model, masks, bbox_min, bbox_max, poses, intrinsics, H, W, proj_dir=None, img_names=None,
                     thresh=0.5):
    """-> bool array [n_views, k, H, W]; when ``proj_dir`` is given also writes ``<img>_<inst>.png``
    (inst = mask index + 1) for every non-empty projection."""
    dev = next(model.parameters()).device
    poses = torch.as_tensor(poses).to(dev).float()
    k = len(masks)
    out = np.zeros((poses.shape[0], k, H, W), dtype=bool)
    was_training = model.training
    model.eval()
    packed = (k, pack_mask_words(masks, dev))          # one 32-bit word per voxel and 32 masks, built once for all views
    for v in range(poses.shape[0]):
        r = get_rays(poses[v:v + 1], intrinsics, H, W, patch=4 if (H % 4 == 0 and W % 4 == 0) else 0)
        soft, _ = soft_project(model, None, bbox_min, bbox_max, r["rays_o"][0], r["rays_d"][0], packed=packed)
        flat = torch.zeros(H * W, k, device=dev)
        flat[r["inds"][0]] = soft
        out[v] = (flat > thresh).t().reshape(k, H, W).cpu().numpy()
        if proj_dir is not None:
            os.makedirs(proj_dir, exist_ok=True)
            name = img_names[v] if img_names is not None else f"{v:04d}"
            for i in range(k):
                if out[v, i].any():
                    save_png_gray(os.path.join(proj_dir, f"{name}_{i + 1}.png"), out[v, i].astype(np.uint8) * 255)
    model.train(was_training)
    return out
