"""Synthetic dataset with the shape of upstream's ``NeRFDataset`` batches.

No dataset exists offline, so the loader renders ground truth analytically from
``instance_nerf_amd.scene.RoomScene``: rgb by first-hit ray tracing, instance ids with a fraction
set to -1 (ignore), matching the supervision format of the reference's matched masks
(/root/reference/Mask2Former_sample/match_seg.py:131-140: int32 [H,W], -1 ignore, 0 background, >0 id).
"""
import numpy as np
import torch

from ..scene import RoomScene
from .utils import get_rays


class SyntheticRoomDataset:
    def __init__(self, device, H=800, W=800, n_views=8, num_rays=4096, training=True, ignore_frac=0.1, seed=2,
                 num_instances=64, rank=0, sort_pixels=False, scale=1.0):
        self.device, self.H, self.W, self.num_rays, self.training = device, H, W, num_rays, training
        self.sort_pixels = sort_pixels      # the batch's random pixels in image order (same set of rays per batch)
        self.room = RoomScene(scale=scale)      # scale s: the room enlarged s times (train it with bound = s)
        poses, self.intrinsics, _, _ = self.room.cameras(n=n_views, H=H, W=W, focal=W / 2.0)
        self.poses = torch.from_numpy(poses).to(device)
        self.rng = np.random.default_rng(seed + 1000 * rank)
        self.ignore_frac, self.num_instances = ignore_frac, num_instances
        self.gen = torch.Generator(device="cpu").manual_seed(seed + 1000 * rank)

    def __len__(self):
        return self.poses.shape[0]

    def batch(self, view=None):
        """One training batch: num_rays random pixels of a random view."""
        view = int(self.rng.integers(0, len(self))) if view is None else view
        N = self.num_rays if self.training else -1
        inds = torch.randint(0, self.H * self.W, (self.num_rays,), generator=self.gen) if N > 0 else None
        if inds is not None:
            inds = (torch.sort(inds).values if self.sort_pixels else inds).to(self.device)
        r = get_rays(self.poses[view:view + 1], self.intrinsics, self.H, self.W, inds=inds)
        ro, rd = r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy()
        rgb, ids, _ = self.room.trace(ro, rd)
        ids = ids % self.num_instances
        drop = self.rng.random(ids.shape[0]) < self.ignore_frac
        masks = np.where(drop, -1, ids)
        return {"rays_o": r["rays_o"], "rays_d": r["rays_d"], "H": self.H, "W": self.W,
                "images": torch.from_numpy(rgb)[None].to(self.device),
                "masks": torch.from_numpy(masks)[None].to(self.device), "index": [view]}

    def __iter__(self):
        for v in range(len(self)):
            yield self.batch(v if not self.training else None)


# ------------------------------------------------------------------------------------------------------------
# Loader for the on-disk format the reference's NeRF stage trains on (README.md:58-66 -> upstream's
# ``nerf/provider.py::NeRFDataset`` [U]: the instant-ngp / Blender ``transforms*.json`` convention the 3D-FRONT NeRF
# data ships in).  Host-side logic only; rays come from ``get_rays`` (HIP on a GPU device).
def nerf_matrix_to_ngp(pose, scale=0.33, offset=(0.0, 0.0, 0.0)):
    """Blender/NeRF camera-to-world [4,4] -> the renderer's frame: axes (x, y, z) <- (y, z, x), the camera's y and z
    columns flipped (OpenGL -> OpenCV), translation scaled and shifted (upstream ``nerf_matrix_to_ngp`` [U])."""
    p = np.asarray(pose, dtype=np.float32)
    o = np.asarray(offset, dtype=np.float32)
    return np.array([[p[1, 0], -p[1, 1], -p[1, 2], p[1, 3] * scale + o[0]],
                     [p[2, 0], -p[2, 1], -p[2, 2], p[2, 3] * scale + o[1]],
                     [p[0, 0], -p[0, 1], -p[0, 2], p[0, 3] * scale + o[2]],
                     [0, 0, 0, 1]], dtype=np.float32)


class NeRFDataset:
    """``transforms.json`` (or ``transforms_{train,val,test}.json``) + images -> ray batches of upstream's shape.

    JSON keys read: ``frames[].file_path`` (extension optional, ``.png`` assumed), ``frames[].transform_matrix``;
    intrinsics from ``fl_x``/``fl_y`` (pixels) or ``camera_angle_x``/``camera_angle_y`` (radians), principal point
    ``cx``/``cy`` (default: image centre), size ``w``/``h`` (default: the first image).  ``downscale`` divides the
    image size and the intrinsics.  RGBA images are composited on white (``bg_color = 1``, the Trainer's default).
    ``mask_dir`` (instance stage): per-image ``<name>.npy`` int32 [H, W] as written by the reference's
    ``Mask2Former_sample/match_seg.py:131-140`` (-1 ignore, 0 background, > 0 instance id).
    Batches (``__getitem__`` / iteration): training -> ``num_rays`` random pixels of ONE image:
    ``rays_o, rays_d [1,N,3]``, ``images [1,N,3]``, ``masks [1,N]`` (if ``mask_dir``), ``H, W, index``;
    otherwise the full image in row-major order.  ``seed`` / ``rank``: the view order and the pixel draws are functions of
    ``seed + 1000 * rank`` (one process per GPU: every rank its own batches).
    """

    def __init__(self, path, type="train", device="cpu", downscale=1, scale=0.33, offset=(0, 0, 0), num_rays=4096,
                 mask_dir=None, num_instances=0, preload=True, seed=0, n_test=10, rank=0):
        import json
        import os
        if hasattr(path, "path"):
            # upstream's call: NeRFDataset(opt, device=device, type='train', downscale=1, n_test=10) - the second
            # positional argument is the device there; path / scale / offset / num_rays / preload come from ``opt``
            opt = path

            def is_device(v):
                return isinstance(v, torch.device) or (isinstance(v, str) and v.split(":")[0] in ("cpu", "cuda"))
            if is_device(type) and not is_device(device):
                type, device = device, type                    # upstream's positional order: (opt, device, type)
            elif is_device(type):
                type, device = "train", type                   # NeRFDataset(opt, device)
            path = opt.path
            scale, offset = getattr(opt, "scale", scale), getattr(opt, "offset", offset)
            num_rays, preload = getattr(opt, "num_rays", num_rays), getattr(opt, "preload", preload)
            mask_dir = getattr(opt, "mask_dir", mask_dir)
            num_instances = getattr(opt, "num_instances", num_instances)
            seed, rank = getattr(opt, "seed", seed), getattr(opt, "local_rank", rank)
            self.opt = opt
        self.root, self.type, self.device = path, type, torch.device(device)
        self.training = type in ("train", "all", "trainval")
        self.num_rays = num_rays if self.training else -1
        self.num_instances = num_instances
        cand = [os.path.join(path, "transforms.json"), os.path.join(path, f"transforms_{type}.json")]
        files = [c for c in cand if os.path.exists(c)]
        if not files:
            raise FileNotFoundError(f"no transforms.json / transforms_{type}.json under {path}")
        with open(files[0]) as f:
            meta = json.load(f)
        frames = sorted(meta["frames"], key=lambda fr: fr["file_path"])
        if os.path.basename(files[0]) == "transforms.json" and type in ("train", "val"):
            # upstream's split of a single transforms.json: the first frame is held out for validation
            frames = frames[1:] if type == "train" else frames[:1]
        from PIL import Image
        names, poses, images = [], [], []
        for fr in frames:
            fp = os.path.join(path, fr["file_path"])
            if not os.path.splitext(fp)[1]:
                fp += ".png"
            if not os.path.exists(fp):
                continue
            names.append(os.path.splitext(os.path.basename(fp))[0])
            poses.append(nerf_matrix_to_ngp(np.asarray(fr["transform_matrix"], dtype=np.float32), scale, offset))
            img = Image.open(fp)
            if downscale != 1:
                img = img.resize((img.width // downscale, img.height // downscale), Image.BILINEAR)
            a = np.asarray(img, dtype=np.float32) / 255.0
            if a.ndim == 2:
                a = np.repeat(a[..., None], 3, -1)
            if a.shape[-1] == 4:
                a = a[..., :3] * a[..., 3:] + (1.0 - a[..., 3:])
            images.append(a[..., :3])
        if not names:
            raise FileNotFoundError(f"none of the {len(frames)} frames of {files[0]} has an image on disk")
        self.names = names
        self.H, self.W = images[0].shape[:2]
        H0 = float(meta.get("h", self.H * downscale)) / downscale
        W0 = float(meta.get("w", self.W * downscale)) / downscale
        if (int(round(H0)), int(round(W0))) != (self.H, self.W):
            raise ValueError(f"transforms say {H0:g}x{W0:g}, images are {self.H}x{self.W}")
        if "fl_x" in meta or "fl_y" in meta:
            fx = float(meta.get("fl_x", meta.get("fl_y"))) / downscale
            fy = float(meta.get("fl_y", meta.get("fl_x"))) / downscale
        elif "camera_angle_x" in meta or "camera_angle_y" in meta:
            fx = self.W / (2 * np.tan(meta["camera_angle_x"] / 2)) if "camera_angle_x" in meta else None
            fy = self.H / (2 * np.tan(meta["camera_angle_y"] / 2)) if "camera_angle_y" in meta else None
            fx, fy = fx or fy, fy or fx
        else:
            raise ValueError("transforms have neither fl_x/fl_y nor camera_angle_x/camera_angle_y")
        cx = float(meta["cx"]) / downscale if "cx" in meta else self.W / 2
        cy = float(meta["cy"]) / downscale if "cy" in meta else self.H / 2
        self.intrinsics = (float(fx), float(fy), float(cx), float(cy))
        dev = self.device if preload else torch.device("cpu")
        self.poses = torch.from_numpy(np.stack(poses)).to(self.device)
        self.images = torch.from_numpy(np.stack(images)).to(dev)                  # [B, H, W, 3]
        self.masks = None
        if mask_dir is not None:
            from ..masks import load_matched_masks
            found = load_matched_masks(mask_dir, names)
            missing = [n for n in names if n not in found]
            if missing:
                raise FileNotFoundError(f"no matched mask for {missing[:3]}{'...' if len(missing) > 3 else ''} in {mask_dir}")
            m = np.stack([found[n] for n in names])
            if m.shape[1:] != (self.H * downscale, self.W * downscale) and m.shape[1:] != (self.H, self.W):
                raise ValueError(f"masks are {m.shape[1:]}, images {self.H}x{self.W}")
            if m.shape[1:] != (self.H, self.W):
                m = m[:, ::downscale, ::downscale][:, :self.H, :self.W]            # labels: nearest, never blended
            self.masks = torch.from_numpy(np.ascontiguousarray(m).astype(np.int32, copy=False)).to(dev)
        # one process per GPU: every rank draws its OWN views and pixels (``rank`` enters both streams of randomness; the
        # ranks' gradients are averaged, so identical draws on all ranks would be one batch computed N times)
        self.rng = np.random.default_rng(int(seed) + 1000 * int(rank))
        # training batches of a device-resident dataset come from ONE launch (inr_sample_training_batch): pixel draw, rays,
        # rgb gather, label gather.  The draw is counter-based: batch number `_draws` of this loader under `seed`.
        self.seed, self._draws = int(seed) + 1000 * int(rank), 0
        self.fused_batches = True

    def __len__(self):
        return self.poses.shape[0]

    def _fused_batch(self, index):
        """One training batch in one launch (``inr_sample_training_batch``, include/inr.h): what the tensor-op path below
        does in eight launches - ``torch.randint`` draw, ``get_rays``, image gather, ``labels_for_rays`` - with the pixel
        draw made reproducible from (seed, batch number).  Same dict, same shapes and dtypes."""
        from .. import _lib
        lib = _lib.load()
        n, dev = int(self.num_rays), self.device
        C = int(self.images.shape[-1])
        inds = torch.empty(n, dtype=torch.int64, device=dev)
        rays_o = torch.empty(1, n, 3, dtype=torch.float32, device=dev)
        rays_d = torch.empty(1, n, 3, dtype=torch.float32, device=dev)
        rgb = torch.empty(1, n, C, dtype=torch.float32, device=dev)
        labels = torch.empty(1, n, dtype=torch.int64, device=dev) if self.masks is not None else None
        fx, fy, cx, cy = self.intrinsics
        _lib.check(lib.inr_sample_training_batch(
            _lib.ptr(self.poses[index], torch.float32, "pose"), float(fx), float(fy), float(cx), float(cy), int(self.H),
            int(self.W), _lib.ptr(self.images[index], torch.float32, "image"), C,
            _lib.ptr(self.masks[index], torch.int32, "mask") if self.masks is not None else None,
            int(self.num_instances or (1 << 30)), self.seed, self._draws & 0x7FFFFFFF, n, _lib.ptr(inds), _lib.ptr(rays_o),
            _lib.ptr(rays_d), _lib.ptr(rgb), _lib.ptr(labels) if labels is not None else None, _lib.stream_ptr()),
            "sample_training_batch")
        self._draws += 1
        out = {"H": self.H, "W": self.W, "rays_o": rays_o, "rays_d": rays_d, "index": [index], "images": rgb}
        if labels is not None:
            out["masks"] = labels
        return out

    def __getitem__(self, index):
        from ..masks import labels_for_rays
        index = int(index)
        if (self.training and self.fused_batches and self.num_rays > 0 and self.device.type == "cuda" and self.images.is_cuda
                and self.images.dtype == torch.float32 and (self.masks is None or self.masks.is_cuda)):
            return self._fused_batch(index)
        r = get_rays(self.poses[index:index + 1], self.intrinsics, self.H, self.W, self.num_rays)
        inds = r["inds"][0]
        out = {"H": self.H, "W": self.W, "rays_o": r["rays_o"], "rays_d": r["rays_d"], "index": [index]}
        img = self.images[index].reshape(-1, 3)
        out["images"] = img[inds.to(img.device)].to(self.device)[None] if self.training else \
            self.images[index].to(self.device)[None]
        if self.masks is not None:
            lab = labels_for_rays(self.masks[index], inds, self.num_instances or (1 << 30))    # on the masks' device
            out["masks"] = (lab if self.training else lab.view(self.H, self.W)).to(self.device)[None]
        return out

    def __iter__(self):
        order = self.rng.permutation(len(self)) if self.training else range(len(self))
        for i in order:
            yield self[i]

    def dataloader(self):
        """Upstream returns a ``DataLoader(list(range(n)), batch_size=1, collate_fn=self.collate)`` that carries the
        dataset as ``_data`` (the Trainer reads ``_data.poses`` / ``_data.intrinsics`` for ``mark_untrained_grid``) and
        a ``has_gt`` flag; iterating this object yields the same batches in the same way."""
        self._data = self
        self.has_gt = self.images is not None
        return self
