"""Morton codes, bit packing and the occupancy-grid update (oracle; test only).

Follows SURVEY.md Appendix A.1 "Morton", "packbits", "Occupancy update"
(rows a3 of section 8a; upstream ``raymarching.{morton3D,morton3D_invert,
packbits}`` and ``NeRFRenderer.update_extra_state`` of the un-vendored
submodule pinned at /root/reference/README.md:27,59).  Parity unpinned.
"""
import numpy as np

U32 = np.uint32
F32 = np.float32


def _expand_bits(v):
    v = v.astype(U32)
    v = (v * U32(0x00010001)) & U32(0xFF0000FF)
    v = (v * U32(0x00000101)) & U32(0x0F00F00F)
    v = (v * U32(0x00000011)) & U32(0xC30C30C3)
    v = (v * U32(0x00000005)) & U32(0x49249249)
    return v


def morton3D(coords):
    """coords int[N,3] (each < 1024) -> uint32[N]; x in bit 0, y bit 1, z bit 2."""
    c = np.asarray(coords)
    with np.errstate(over="ignore"):
        return (_expand_bits(c[:, 0]) | (_expand_bits(c[:, 1]) << U32(1))
                | (_expand_bits(c[:, 2]) << U32(2))).astype(U32)


def _compact_bits(x):
    x = x.astype(U32) & U32(0x49249249)
    x = (x | (x >> U32(2))) & U32(0xC30C30C3)
    x = (x | (x >> U32(4))) & U32(0x0F00F00F)
    x = (x | (x >> U32(8))) & U32(0xFF0000FF)
    x = (x | (x >> U32(16))) & U32(0x0000FFFF)
    return x


def morton3D_invert(indices):
    """uint32[N] -> int32[N,3]."""
    m = np.asarray(indices).astype(U32)
    return np.stack([_compact_bits(m), _compact_bits(m >> U32(1)),
                     _compact_bits(m >> U32(2))], -1).astype(np.int32)


def packbits(grid, thresh):
    """grid f32[n] (n % 8 == 0) -> u8[n/8]; byte k bit i = grid[8k+i] > thresh."""
    g = np.asarray(grid, dtype=F32).reshape(-1, 8)
    bits = (g > F32(thresh)).astype(np.uint8)
    w = (np.uint8(1) << np.arange(8, dtype=np.uint8))
    return (bits * w).sum(-1).astype(np.uint8)


def cell_centers(coords, H, cascade_level, bound, jitter=None):
    """World position of grid cells.  coords int[n,3] -> f32[n,3].

    x = 2*c/(H-1) - 1;  b = min(2^cas, bound);  half = b/H
    cas_x = x*(b-half) + jitter*half,  jitter in [-1,1) (0 when None).
    """
    c = np.asarray(coords).astype(F32)
    x = F32(2.0) * c / F32(H - 1) - F32(1.0)
    b = F32(min(2.0 ** cascade_level, bound))
    half = b / F32(H)
    out = x * (b - half)
    if jitter is not None:
        out = out + np.asarray(jitter, dtype=F32) * half
    return out.astype(F32)


def update_density_grid(density_grid, sigma_fn, H, cascade, bound, decay=0.95,
                        density_scale=1.0, density_thresh=10.0, coords=None,
                        jitter=None):
    """One full occupancy update.  density_grid f32[C,H^3] in Morton order.

    ``sigma_fn(xyz f32[n,3]) -> f32[n]``.  coords int[n,3] defaults to all H^3
    cells (the first-16-updates regime); jitter f32[C,n,3] in [-1,1) or None.
    grid = max(grid*decay, sigma*density_scale) where grid >= 0 (cells marked
    -1 by mark_untrained_grid stay untouched);  mean over clamp(grid, 0);
    thresh = min(mean, density_thresh);  bitfield = packbits(grid, thresh).
    Returns (new_grid, bitfield u8[C*H^3/8], mean_density).
    """
    grid = np.array(density_grid, dtype=F32, copy=True)
    if coords is None:
        r = np.arange(H, dtype=np.int32)
        xx, yy, zz = np.meshgrid(r, r, r, indexing="ij")
        coords = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], -1)
    idx = morton3D(coords).astype(np.int64)
    tmp = np.full_like(grid, -1.0)
    for cas in range(cascade):
        j = None if jitter is None else jitter[cas]
        xyz = cell_centers(coords, H, cas, bound, j)
        tmp[cas, idx] = (np.asarray(sigma_fn(xyz), dtype=F32) * F32(density_scale))
    valid = (grid >= 0) & (tmp >= 0)
    grid[valid] = np.maximum(grid[valid] * F32(decay), tmp[valid])
    mean_density = float(np.clip(grid, 0, None).mean(dtype=np.float64))
    thresh = min(mean_density, density_thresh)
    return grid, packbits(grid.reshape(-1), thresh), mean_density


def mark_untrained_cells(poses, intrinsic, H, cascade, bound):
    """Cells no camera sees (upstream ``NeRFRenderer.mark_untrained_grid``): for every cascade the
    cell centre x = (2c/(H-1) - 1) * (b - b/H) is moved into each camera frame, cam = R^T (x - t); the cell is
    seen if z > 0, |x| < cx/fx * z + 2*b/H and |y| < cy/fy * z + 2*b/H.  Returns bool [C, H^3] in Morton order,
    True = unseen (the renderer stores -1 there)."""
    fx, fy, cx, cy = intrinsic
    r = np.arange(H, dtype=np.int32)
    xx, yy, zz = np.meshgrid(r, r, r, indexing="ij")
    coords = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], -1)
    idx = morton3D(coords).astype(np.int64)
    poses = np.asarray(poses, dtype=np.float32)
    unseen = np.ones((cascade, H ** 3), dtype=bool)
    for cas in range(cascade):
        b = min(2.0 ** cas, bound)
        half = b / H
        world = (F32(2.0) * coords.astype(F32) / F32(H - 1) - F32(1.0)) * F32(b - half)
        seen = np.zeros(coords.shape[0], dtype=bool)
        for P in poses:
            cam = (world - P[:3, 3]) @ P[:3, :3]
            seen |= (cam[:, 2] > 0) & (np.abs(cam[:, 0]) < cx / fx * cam[:, 2] + half * 2) \
                & (np.abs(cam[:, 1]) < cy / fy * cam[:, 2] + half * 2)
        unseen[cas, idx] = ~seen
    return unseen


def sample_cells(grid, u, n, n_slices=4096):
    """The steady-state cell choice of the occupancy update: n uniformly random cells + n picks, with replacement,
    among the occupied cells (grid > 0) of one cascade - upstream's ``randint`` coordinates / ``nonzero(grid > 0)
    [randint]`` (SURVEY Appendix A.1 "Occupancy update"), drawn so that the picks come out grouped by slices of the
    Morton range without a sort: an i.i.d. uniform sample = multinomial counts per slice (the histogram of one set of
    uniform draws) + uniform positions inside the slices (a second set).  Restates csrc/raymarch.hip::k_occ_hist /
    k_occ_pick exactly (float32 slice draw, float64 position arithmetic).

    grid float32 [n_cells] (Morton order); u float32 [4n]: slice draws [2n] (uniform half, occupied half), in-slice
    draws [2n].  -> int32 [2n] Morton indices, uniform half first."""
    grid = np.asarray(grid, F32)
    u = np.asarray(u, F32)
    n_cells = grid.shape[0]
    occ = np.nonzero(grid > 0)[0]
    out = np.empty(2 * n, np.int64)
    for half in range(2):
        draws = u[half * n:(half + 1) * n]
        b = np.minimum((draws * F32(n_slices)).astype(np.int64), n_slices - 1)
        b_sorted = np.sort(b)                             # slot j lies in the slice whose scanned count range holds j
        v = (b_sorted.astype(np.float64) + u[2 * n + half * n:2 * n + (half + 1) * n].astype(np.float64)) / n_slices
        if half == 0:
            out[:n] = np.minimum((v * n_cells).astype(np.int64), n_cells - 1)
        elif occ.size == 0:
            out[n:] = 0
        else:
            out[n:] = occ[np.minimum((v * occ.size).astype(np.int64), occ.size - 1)]
    return out.astype(np.int32)
