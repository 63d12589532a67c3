"""Ray generation and ray/AABB intersection (oracle; test infrastructure only).

Follows SURVEY.md Appendix A.1 "get_rays" and "near_far_from_aabb"
(rows a1/a2 of section 8a; upstream symbols ``nerf/utils.py::get_rays`` and
``raymarching.near_far_from_aabb`` of the un-vendored submodule pinned at
/root/reference/README.md:27,59).  Parity unpinned - see ``oracle/__init__``.
"""
import numpy as np

F32 = np.float32
FLT_MAX = np.finfo(np.float32).max


def get_rays(poses, intrinsics, H, W, N=-1, rng=None, inds=None):
    """poses f32[B,4,4] (camera-to-world, NGP convention), intrinsics (fx,fy,cx,cy).

    Returns dict(rays_o f32[B,n,3], rays_d f32[B,n,3], inds i64[B,n]).
    Pixel (i=column, j=row) centre at +0.5; dir = ((i-cx)/fx, (j-cy)/fy, 1),
    normalised in fp32, rotated by R (rays_d = dir @ R^T); rays_o = t.
    N>0 draws N flat pixel indices per pose (shared across the batch, as
    upstream does) from ``rng`` unless ``inds`` is given.
    """
    poses = np.asarray(poses, dtype=F32)
    B = poses.shape[0]
    fx, fy, cx, cy = [F32(v) for v in intrinsics]
    if inds is None:
        if N > 0:
            rng = rng or np.random.default_rng(0)
            inds = rng.integers(0, H * W, size=(N,), dtype=np.int64)
        else:
            inds = np.arange(H * W, dtype=np.int64)
    inds = np.asarray(inds, dtype=np.int64)
    i = (inds % W).astype(F32) + F32(0.5)
    j = (inds // W).astype(F32) + F32(0.5)
    xs = (i - cx) / fx
    ys = (j - cy) / fy
    zs = np.ones_like(xs)
    d = np.stack([xs, ys, zs], -1)                       # [n,3]
    nrm = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
    d = d / nrm[:, None]
    R = poses[:, :3, :3]                                 # [B,3,3]
    # rays_d[b,n,r] = sum_c d[n,c] * R[b,r,c], accumulated c = 0,1,2 in fp32
    rays_d = (d[None, :, None, 0] * R[:, None, :, 0]
              + d[None, :, None, 1] * R[:, None, :, 1]
              + d[None, :, None, 2] * R[:, None, :, 2]).astype(F32)
    rays_o = np.broadcast_to(poses[:, None, :3, 3], rays_d.shape).astype(F32).copy()
    return dict(rays_o=rays_o, rays_d=rays_d,
                inds=np.broadcast_to(inds[None], (B, inds.shape[0])).copy())


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """Slab test. rays_* f32[N,3], aabb f32[6]=(xmin,ymin,zmin,xmax,ymax,zmax).

    Returns nears, fars f32[N]; a miss gives near = far = FLT_MAX; otherwise
    near = max(near, min_near).  rd = 1/d is a correctly rounded fp32 divide;
    a zero direction component gives +-inf and the IEEE results follow.
    """
    o = np.asarray(rays_o, dtype=F32)
    d = np.asarray(rays_d, dtype=F32)
    aabb = np.asarray(aabb, dtype=F32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        rd = F32(1.0) / d
        near = np.full(o.shape[0], -np.inf, dtype=F32)
        far = np.full(o.shape[0], np.inf, dtype=F32)
        miss = np.zeros(o.shape[0], dtype=bool)
        for a in range(3):
            t0 = (aabb[a] - o[:, a]) * rd[:, a]
            t1 = (aabb[a + 3] - o[:, a]) * rd[:, a]
            lo = np.where(t0 > t1, t1, t0)       # swap so lo <= hi (NaN keeps t0,t1 order)
            hi = np.where(t0 > t1, t0, t1)
            if a > 0:
                miss |= (near > hi) | (lo > far)
            near = np.where(lo > near, lo, near)
            far = np.where(hi < far, hi, far)
        near = np.where(near < F32(min_near), F32(min_near), near)
    near = np.where(miss, FLT_MAX, near).astype(F32)
    far = np.where(miss, FLT_MAX, far).astype(F32)
    return near, far


MASK64 = (1 << 64) - 1


def _mix64(z):
    """SplitMix64 finaliser on Python ints (exact 64-bit wrap-around)."""
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def sample_pixels(seed, step, n, H, W):
    """The counter-based pixel draw of the fused loader launch (include/inr.h ``inr_sample_training_batch``; no upstream
    counterpart - upstream draws with ``torch.randint``; any i.i.d. uniform draw with replacement is the same loader):
    inds[k] = ((mix64(seed + GOLDEN * (step * 2^32 + k)) >> 32) * (H*W)) >> 32.  -> int64 [n]."""
    hw = H * W
    out = np.empty(n, dtype=np.int64)
    for k in range(n):
        h = _mix64((seed + 0x9E3779B97F4A7C15 * ((step << 32) + k)) & MASK64)
        out[k] = ((h >> 32) * hw) >> 32
    return out


def sample_training_batch(pose, intrinsics, H, W, image, mask, num_instances, seed, step, n):
    """Restatement of ``inr_sample_training_batch``: pixels by ``sample_pixels``, rays by ``get_rays`` (upstream
    ``nerf/utils.py::get_rays`` [U]), rgb = image[inds], labels = mask[inds] with ids >= num_instances -> -1 (the
    loader's view of /root/reference/Mask2Former_sample/match_seg.py:131-140 masks).  image f32 [H,W,C] or None,
    mask int32 [H,W] or None."""
    inds = sample_pixels(seed, step, n, H, W)
    r = get_rays(np.asarray(pose, dtype=F32)[None], intrinsics, H, W, inds=inds)
    out = {"inds": inds, "rays_o": r["rays_o"][0], "rays_d": r["rays_d"][0]}
    if image is not None:
        img = np.asarray(image, dtype=F32)
        out["rgb"] = img.reshape(H * W, -1)[inds]
    if mask is not None:
        lab = np.asarray(mask).reshape(-1)[inds].astype(np.int64)
        out["labels"] = np.where(lab >= num_instances, -1, lab)
    return out
