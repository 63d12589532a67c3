"""Synthetic 3D-FRONT-like room used for measurement.

No dataset is available offline, so throughput and parity runs use this
analytic scene: a wall/floor/ceiling shell at |coord| in [0.90, 0.94] plus 12
axis-aligned boxes (centres U(-0.7,0.7), half extents U(0.05,0.25),
``numpy.random.default_rng(0)``); instance id = box index + 1, walls = 0.
Cameras: 800x800, fx = fy = 400, cx = cy = 400; 8 poses (seed 1) with the eye
in U(-0.5,0.5)^3 looking at the origin, up = +z.

``RoomScene(scale=s)`` is the same room enlarged s times about the origin (walls at
s * [0.90, 0.94], cameras in s * U(-0.5,0.5)^3): with ``bound = s`` it fills a torch-ngp volume of
1 + ceil(log2 s) occupancy cascades the way a real 3D-FRONT room trained at bound 2-8 does, and
``density_bitfield(H, bound)`` then returns all cascades (cascade c covers [-min(2^c, bound), ..]^3).

This is data generation, not part of the render algorithm; both the product
code (bench) and the tests use it.  numpy only.
"""
import numpy as np

F32 = np.float32
WALL_IN, WALL_OUT = 0.90, 0.94


class RoomScene:
    def __init__(self, n_boxes=12, seed=0, scale=1.0):
        rng = np.random.default_rng(seed)
        self.scale = float(scale)
        self.wall_in, self.wall_out = WALL_IN * self.scale, WALL_OUT * self.scale
        self.centres = rng.uniform(-0.7, 0.7, size=(n_boxes, 3)) * self.scale
        self.halves = rng.uniform(0.05, 0.25, size=(n_boxes, 3)) * self.scale
        self.lo = np.maximum(self.centres - self.halves, -self.wall_in)
        self.hi = np.minimum(self.centres + self.halves, self.wall_in)
        pal = np.random.default_rng(seed + 100).uniform(0.15, 0.95, size=(n_boxes + 1, 3))
        pal[0] = (0.8, 0.8, 0.75)
        self.palette = pal.astype(F32)

    # ------------------------------------------------------------------ occupancy
    def occupancy_grid(self, H=128, bound=1.0):
        """bool[H,H,H] indexed [x,y,z]: cell of the grid over [-bound, bound]^3 overlaps geometry (one cascade)."""
        edges = -bound + 2.0 * bound * np.arange(H + 1) / H
        lo_e, hi_e = edges[:-1], edges[1:]

        def overlap(a, b):            # cells whose [lo,hi] interval overlaps [a,b]
            return (hi_e > a) & (lo_e < b)

        occ = np.zeros((H, H, H), dtype=bool)
        inside = overlap(-self.wall_out, self.wall_out)
        wall = overlap(self.wall_in, self.wall_out) | overlap(-self.wall_out, -self.wall_in)
        ix, iy, iz = inside[:, None, None], inside[None, :, None], inside[None, None, :]
        occ |= wall[:, None, None] & iy & iz
        occ |= wall[None, :, None] & ix & iz
        occ |= wall[None, None, :] & ix & iy
        for lo, hi in zip(self.lo, self.hi):
            m = [overlap(lo[a], hi[a]) for a in range(3)]
            occ |= m[0][:, None, None] & m[1][None, :, None] & m[2][None, None, :]
        return occ

    def density_bitfield(self, H=128, bound=1.0):
        """u8[C*H^3/8], C = 1 + ceil(log2 bound) cascades one after the other, each in Morton order (bit i of byte k =
        cell with morton code 8k+i); cascade c is the grid over [-min(2^c, bound), min(2^c, bound)]^3."""
        C = 1 + int(np.ceil(np.log2(max(bound, 1.0))))
        if C > 1:
            return np.concatenate([self._bitfield_of(self.occupancy_grid(H, min(2.0 ** c, bound))) for c in range(C)])
        return self._bitfield_of(self.occupancy_grid(H, bound))

    @staticmethod
    def _bitfield_of(occ):
        H = occ.shape[0]
        r = np.arange(H, dtype=np.uint32)

        def ex(v):
            v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
            v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
            v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
            v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
            return v
        e = ex(r)
        code = (e[:, None, None] | (e[None, :, None] << np.uint32(1))
                | (e[None, None, :] << np.uint32(2))).astype(np.int64)
        flat = np.zeros(H ** 3, dtype=np.uint8)
        flat[code.ravel()] = occ.ravel().astype(np.uint8)
        return np.packbits(flat.reshape(-1, 8), axis=1, bitorder="little").ravel()

    # ------------------------------------------------------------------ cameras
    @staticmethod
    def look_at(eye, target=(0, 0, 0), up=(0, 0, 1)):
        eye = np.asarray(eye, dtype=np.float64)
        fwd = np.asarray(target, dtype=np.float64) - eye
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.asarray(up, dtype=np.float64))
        if np.linalg.norm(right) < 1e-8:
            right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        pose = np.eye(4)
        pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, eye
        return pose.astype(F32)

    def cameras(self, n=8, seed=1, H=800, W=800, focal=400.0):
        rng = np.random.default_rng(seed)
        eyes = rng.uniform(-0.5, 0.5, size=(n, 3)) * self.scale
        poses = np.stack([self.look_at(e) for e in eyes])
        return poses, (focal, focal, W / 2.0, H / 2.0), H, W

    # ------------------------------------------------------------------ ground truth
    def trace(self, rays_o, rays_d):
        """Analytic first hit.  Returns (rgb f32[N,3], instance i64[N], t f32[N])."""
        o = np.asarray(rays_o, dtype=np.float64)
        d = np.asarray(rays_d, dtype=np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            rd = 1.0 / d
            # walls: exit point of the inner room box [-WALL_IN, WALL_IN]^3
            wi = self.wall_in
            t_exit = np.min(np.where(d > 0, (wi - o) * rd, (-wi - o) * rd), axis=1)
            best_t = t_exit.copy()
            best_id = np.zeros(o.shape[0], dtype=np.int64)
            axis_hit = np.argmin(np.where(d > 0, (wi - o) * rd, (-wi - o) * rd), axis=1)
            for b, (lo, hi) in enumerate(zip(self.lo, self.hi)):
                t0 = (lo - o) * rd
                t1 = (hi - o) * rd
                tn = np.max(np.minimum(t0, t1), axis=1)
                tf = np.min(np.maximum(t0, t1), axis=1)
                hit = (tn <= tf) & (tf > 0)
                tb = np.where(tn > 0, tn, 0.0)
                upd = hit & (tb < best_t)
                ax = np.argmax(np.minimum(t0, t1), axis=1)
                best_t = np.where(upd, tb, best_t)
                best_id = np.where(upd, b + 1, best_id)
                axis_hit = np.where(upd, ax, axis_hit)
        shade = np.asarray([1.0, 0.85, 0.7])[axis_hit]
        rgb = self.palette[best_id] * shade[:, None]
        return rgb.astype(F32), best_id, best_t.astype(F32)

    # ------------------------------------------------------------------ files
    def write_dataset(self, path, n_views=24, H=400, W=400, seed=1, num_instances=0, ignore_frac=0.1, mask_seed=2):
        """The room as the files the reference's two training stages read (README.md:58-66): ``images/NNNN.png`` (8-bit
        rgb of the analytic first hit), ``transforms_train.json`` (instant-ngp convention: Blender camera matrices,
        ``fl_x/fl_y/cx/cy/w/h``; load with ``NeRFDataset(path, scale=1.0)``) and, with ``num_instances`` > 0,
        ``matched/NNNN.npy`` - int32 [H, W] labels in the layout of ``Mask2Former_sample/match_seg.py:131-140`` (-1 ignore,
        0 background, > 0 instance id; ``ignore_frac`` of the pixels carry -1 as unmatched segments do there).
        -> dict(path, mask_dir, names, poses, intrinsics, ids [n, H, W]).  Data generation (numpy + PIL), used by bench.py
        and the tests to run the product's loader and training loop from disk."""
        import json
        import os
        from PIL import Image
        os.makedirs(os.path.join(path, "images"), exist_ok=True)
        poses, intr, _, _ = self.cameras(n=n_views, seed=seed, H=H, W=W, focal=W / 2.0)
        fx, fy, cx, cy = intr
        jj, ii = np.meshgrid(np.arange(H, dtype=np.float64) + 0.5, np.arange(W, dtype=np.float64) + 0.5, indexing="ij")
        d_cam = np.stack([(ii - cx) / fx, (jj - cy) / fy, np.ones_like(ii)], -1).reshape(-1, 3)
        d_cam /= np.linalg.norm(d_cam, axis=1, keepdims=True)
        rng = np.random.default_rng(mask_seed)
        mask_dir = os.path.join(path, "matched") if num_instances else None
        if mask_dir:
            os.makedirs(mask_dir, exist_ok=True)
        names, frames, all_ids = [], [], []
        for i, P in enumerate(poses):
            P64 = P.astype(np.float64)
            rd = d_cam @ P64[:3, :3].T
            ro = np.broadcast_to(P64[:3, 3], rd.shape)
            rgb, ids, _ = self.trace(ro, rd)
            name = f"{i:04d}"
            names.append(name)
            all_ids.append(ids.reshape(H, W))
            Image.fromarray((rgb.reshape(H, W, 3) * 255.0 + 0.5).astype(np.uint8)).save(os.path.join(path, "images", name + ".png"))
            T = np.eye(4, dtype=np.float32)     # the file stores the Blender-convention matrix (inverse of nerf_matrix_to_ngp)
            T[[1, 2, 0], 0], T[[1, 2, 0], 1], T[[1, 2, 0], 2], T[[1, 2, 0], 3] = P[:3, 0], -P[:3, 1], -P[:3, 2], P[:3, 3]
            frames.append({"file_path": f"images/{name}.png", "transform_matrix": T.tolist()})
            if mask_dir:
                lab = (ids % num_instances).astype(np.int32)
                lab[rng.random(lab.shape[0]) < ignore_frac] = -1
                np.save(os.path.join(mask_dir, name + ".npy"), lab.reshape(H, W))
        with open(os.path.join(path, "transforms_train.json"), "w") as f:
            json.dump({"fl_x": fx, "fl_y": fy, "cx": cx, "cy": cy, "w": W, "h": H, "frames": frames}, f)
        return {"path": path, "mask_dir": mask_dir, "names": names, "poses": poses, "intrinsics": intr,
                "ids": np.stack(all_ids)}

    def instance_of_points(self, x):
        """Instance id of points (0 = wall/empty, b+1 inside box b; first box wins)."""
        x = np.asarray(x, dtype=np.float64)
        ids = np.zeros(x.shape[0], dtype=np.int64)
        for b in reversed(range(len(self.lo))):
            inside = np.all((x >= self.lo[b]) & (x <= self.hi[b]), axis=1)
            ids = np.where(inside, b + 1, ids)
        return ids


def blender_poses(n=100, radius=4.0311, seed=0):
    """Config #1 style poses: cameras on a sphere looking at the origin."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    v[:, 2] = np.abs(v[:, 2])
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    return np.stack([RoomScene.look_at(radius * e) for e in v])
