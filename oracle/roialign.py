"""3-D RoIAlign (oracle; test infrastructure only) - SURVEY.md 8f row f2.

The reference calls ``roi_align.roi_align.roi_align_3d(input, rois, roi_inds, ow, ol, oh, spatial_scale)``
(/root/reference/nerf_rcnn/model/utils.py:604-609); the extension that implements it is an un-vendored
submodule (/root/reference/.gitmodules:1-3) -> PARITY UNPINNED.  The wrapper documents the semantics as
"modified from torchvision.ops.roi_align ... average pooling ... adaptive ceil(roi_size / output_size)
sampling grid" (utils.py:556-592), with ``sampling_ratio`` never forwarded (utils.py:597,608).  This is
therefore torchvision's roi_align (aligned=False, sampling_ratio<=0) extended to three axes:
x <-> W (dim 2), y <-> L (dim 3), z <-> H (dim 4).
"""
import math

import numpy as np


def _interp(vol, x, y, z):
    """vol [C, W, L, H]; torchvision's bilinear_interpolate boundary rules, per axis."""
    C, W, L, H = vol.shape
    if x < -1.0 or x > W or y < -1.0 or y > L or z < -1.0 or z > H:
        return np.zeros(C, dtype=np.float64)
    lo, hi, fr = [], [], []
    for v, n in ((x, W), (y, L), (z, H)):
        v = max(v, 0.0)
        l = int(v)
        if l >= n - 1:
            l = h = n - 1
            v = float(l)
        else:
            h = l + 1
        lo.append(l), hi.append(h), fr.append(v - l)
    out = np.zeros(C, dtype=np.float64)
    for cx, wx in ((lo[0], 1 - fr[0]), (hi[0], fr[0])):
        for cy, wy in ((lo[1], 1 - fr[1]), (hi[1], fr[1])):
            for cz, wz in ((lo[2], 1 - fr[2]), (hi[2], fr[2])):
                out += wx * wy * wz * vol[:, cx, cy, cz]
    return out


def _geometry(roi, osz, spatial_scale):
    """RoI geometry in float32, operation by operation as torchvision's kernels (T = float) and the HIP kernels do
    it: the sampling grid ceil(size / out) is an integer DECISION - a box whose scaled size is a whole multiple of
    the output size within float32 rounding gets one more sample per bin in float64 than in float32 (found by
    replaying the reference's recorded calls, tests/test_reference_calls.py: 0.03 absolute on such a box)."""
    f32 = np.float32
    start = [f32(roi[a]) * f32(spatial_scale) for a in range(3)]
    end = [f32(roi[a + 3]) * f32(spatial_scale) for a in range(3)]
    size = [np.maximum(end[a] - start[a], f32(1.0)) for a in range(3)]
    binsz = [size[a] / f32(osz[a]) for a in range(3)]
    grid = [int(np.ceil(size[a] / f32(osz[a]))) for a in range(3)]
    count = max(grid[0] * grid[1] * grid[2], 1)

    def coord(a, p, i):        # start + p * bin + (i + 0.5) * bin / grid, float32, left to right
        return float(start[a] + f32(p) * binsz[a] + (f32(i) + f32(0.5)) * binsz[a] / f32(grid[a]))
    return grid, count, coord


def roi_align_3d(inp, rois, roi_inds, ow, ol, oh, spatial_scale):
    """inp f32[N,C,W,L,H], rois f32[K,6] (x1,y1,z1,x2,y2,z2), roi_inds i32[K] -> f32[K,C,ow,ol,oh]."""
    inp = np.asarray(inp, dtype=np.float64)
    K = len(rois)
    C = inp.shape[1]
    out = np.zeros((K, C, ow, ol, oh), dtype=np.float64)
    for k in range(K):
        vol = inp[int(roi_inds[k])]
        grid, count, coord = _geometry(rois[k], (ow, ol, oh), spatial_scale)
        for pw in range(ow):
            for pl in range(ol):
                for ph in range(oh):
                    acc = np.zeros(C)
                    for ix in range(grid[0]):
                        x = coord(0, pw, ix)
                        for iy in range(grid[1]):
                            y = coord(1, pl, iy)
                            for iz in range(grid[2]):
                                z = coord(2, ph, iz)
                                acc += _interp(vol, x, y, z)
                    out[k, :, pw, pl, ph] = acc / count
    return out.astype(np.float32)


def roi_align_3d_at(inp, rois, roi_inds, ow, ol, oh, spatial_scale, points):
    """The same op at chosen output elements only (full-size checks: BASELINE configs[4] has 65.5 M of them).
    points int[P,5] = (k, c, pw, pl, ph) -> f32[P]."""
    inp = np.asarray(inp)
    out = np.zeros(len(points), dtype=np.float64)
    for n, (k, c, pw, pl, ph) in enumerate(np.asarray(points).tolist()):
        vol = inp[int(roi_inds[k]), c:c + 1].astype(np.float64)
        grid, count, coord = _geometry(rois[k], (ow, ol, oh), spatial_scale)
        acc = 0.0
        for ix in range(grid[0]):
            x = coord(0, pw, ix)
            for iy in range(grid[1]):
                y = coord(1, pl, iy)
                for iz in range(grid[2]):
                    acc += _interp(vol, x, y, coord(2, ph, iz))[0]
        out[n] = acc / count
    return out.astype(np.float32)
