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
                 num_instances=64, rank=0):
        self.device, self.H, self.W, self.num_rays, self.training = device, H, W, num_rays, training
        self.room = RoomScene()
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
        inds = torch.randint(0, self.H * self.W, (self.num_rays,), generator=self.gen).to(self.device) if N > 0 else None
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
