"""Oracle self-consistency: integer / ray-marching parts (CPU)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import march, occupancy, rays
from conftest import scene_rays


@settings(max_examples=50, deadline=None)
@given(st.lists(st.tuples(*[st.integers(0, 1023)] * 3), min_size=1, max_size=64))
def test_morton_roundtrip(coords):
    c = np.asarray(coords, dtype=np.int32)
    m = occupancy.morton3D(c)
    assert (occupancy.morton3D_invert(m) == c).all()
    # bit-interleave definition, checked bit by bit
    for (x, y, z), code in zip(coords, m):
        ref = 0
        for b in range(10):
            ref |= ((x >> b) & 1) << (3 * b) | ((y >> b) & 1) << (3 * b + 1) | ((z >> b) & 1) << (3 * b + 2)
        assert ref == int(code)


def test_packbits_matches_numpy():
    g = np.random.default_rng(0).normal(size=4096).astype(np.float32)
    ours = occupancy.packbits(g, 0.1)
    ref = np.packbits((g > 0.1).reshape(-1, 8), axis=1, bitorder="little").ravel()
    assert (ours == ref).all()


def test_sample_cells_is_a_uniform_sample_grouped_by_slices():
    """oracle/occupancy.py::sample_cells (restated by inr_occ_sample_cells): the uniform half is uniform over the cells
    (chi-square over 512 coarse bins), the occupied half is uniform over the occupied cells and hits nothing else, both
    come out grouped by slice, and an empty occupied set falls back to cell 0."""
    rng = np.random.default_rng(0)
    n_cells = 64 ** 3
    n = n_cells // 4
    grid = np.where(rng.random(n_cells) < 0.1, rng.random(n_cells) + 0.1, 0.0).astype(np.float32)
    grid[::97] = -1.0
    idx = occupancy.sample_cells(grid, rng.random(4 * n, dtype=np.float32), n)
    assert idx.dtype == np.int32 and idx.shape == (2 * n,) and idx.min() >= 0 and idx.max() < n_cells
    counts = np.bincount(idx[:n] // (n_cells // 512), minlength=512)
    chi2 = float(((counts - n / 512) ** 2 / (n / 512)).sum())
    assert 380 < chi2 < 660                                   # 511 degrees of freedom: mean 511, sigma 32
    occ = np.nonzero(grid > 0)[0]
    assert np.isin(idx[n:], occ).all()
    rank = np.searchsorted(occ, idx[n:])
    counts = np.bincount(rank * 64 // occ.size, minlength=64)
    chi2 = float(((counts - n / 64) ** 2 / (n / 64)).sum())
    assert 25 < chi2 < 115                                    # 63 degrees of freedom
    assert (np.diff(idx[:n] // (n_cells // 4096)) >= 0).all()
    assert (np.diff(rank) >= -(occ.size // 4096 + 1)).all()     # never further back than one slice of the ranks
    assert (occupancy.sample_cells(np.zeros(n_cells, np.float32), rng.random(4 * n, dtype=np.float32), n)[n:] == 0).all()


def test_scene_bitfield_is_morton_packbits(room, room_bitfield):
    occ = room.occupancy_grid(128, 1.0)
    r = np.arange(128)
    xx, yy, zz = np.meshgrid(r, r, r, indexing="ij")
    co = np.stack([xx.ravel(), yy.ravel(), zz.ravel()], -1)
    g = np.zeros(128 ** 3, np.float32)
    g[occupancy.morton3D(co)] = occ.ravel()
    assert (occupancy.packbits(g, 0.5) == room_bitfield).all()
    assert 0.05 < occ.mean() < 0.5


def test_near_far_basic():
    o = np.asarray([[0, 0, 0], [0, 0, -3], [0, 5, 0], [0.5, 0.5, 0.5]], np.float32)
    d = np.asarray([[0, 0, 1], [0, 0, 1], [1, 0, 0], [-1, 0, 0]], np.float32)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = rays.near_far_from_aabb(o, d, aabb, min_near=0.05)
    assert np.allclose(n[[0, 1, 3]], [0.05, 2.0, 0.05]) and np.allclose(f[[0, 1, 3]], [1.0, 4.0, 1.5])
    assert n[2] == np.finfo(np.float32).max and f[2] == n[2]


def test_get_rays_unit_and_center(room):
    poses, intr, H, W = room.cameras()
    r = rays.get_rays(poses[:2], intr, H, W, inds=np.asarray([0, H * W - 1, (H // 2) * W + W // 2]))
    assert r["rays_d"].shape == (2, 3, 3)
    assert np.allclose(np.linalg.norm(r["rays_d"], axis=-1), 1, atol=1e-6)
    # the central pixel looks (almost) along the camera forward axis = pose[:3,2]
    assert np.allclose(r["rays_d"][:, 2], poses[:2, :3, 2], atol=2e-3)


def _scalar_march(o, d, bits, near, far, noise, bound=1.0, C=1, H=128, dt_gamma=0.0, max_steps=1024):
    """Independent per-ray scalar restatement (pure Python loop, small N only)."""
    f = np.float32
    dt_min, dt_max = march.dt_limits(max_steps, C, H)
    clamp = lambda v, lo, hi: min(max(v, lo), hi)
    rd = [f(1.0) / x if x != 0 else f(np.inf) * np.copysign(f(1), x) for x in d]
    sg = [np.copysign(f(1), x) for x in d]
    t = f(near + clamp(f(near * f(dt_gamma)), dt_min, dt_max) * f(noise))
    last = t
    out = []
    while t < far and len(out) < max_steps:
        p = [f(clamp(f(o[a] + f(t * d[a])), f(-bound), f(bound))) for a in range(3)]
        dt = f(clamp(f(t * f(dt_gamma)), dt_min, dt_max))
        lp = min(C - 1, max(0, int(np.frexp(max(abs(v) for v in p))[1])))
        ld = min(C - 1, max(0, int(np.frexp(f(f(dt * f(H)) * f(0.5)))[1])))
        lvl = max(lp, ld)
        mb = f(min(2.0 ** lvl, bound))
        rmb = f(1.0) / mb
        n = [int(clamp(int(f(f(f(f(p[a] * rmb) + f(1)) * f(0.5)) * f(H))), 0, H - 1)) for a in range(3)]
        code = int(occupancy.morton3D(np.asarray([n]))[0]) + lvl * H ** 3
        if (bits[code >> 3] >> (code & 7)) & 1:
            tn = f(t + dt)
            out.append((p, dt, f(tn - last)))
            t = tn
            last = tn
        else:
            tc = []
            for a in range(3):
                aa = f(f(f(n[a]) + f(0.5)) + f(f(0.5) * sg[a]))
                with np.errstate(all="ignore"):
                    tc.append(f(f(f(f(f(f(aa * f(1.0 / H)) * f(2)) - f(1)) * mb) - p[a]) * rd[a]))
            tm = np.fmin(tc[0], np.fmin(tc[1], tc[2]))
            tt = f(t + np.fmax(f(0), tm))
            while True:
                t = f(t + f(clamp(f(t * f(dt_gamma)), dt_min, dt_max)))
                if not (t < tt):
                    break
    return out


@pytest.mark.parametrize("dt_gamma", [0.0, 1.0 / 128])
def test_march_train_matches_scalar_loop(room, room_bitfield, dt_gamma):
    ro, rd = scene_rays(room, n=24, seed=5)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    noises = np.random.default_rng(1).random(24).astype(np.float32)
    m = march.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars, noises, dt_gamma, 1024)
    assert m["rays"][:, 1].tolist() == np.concatenate([[0], np.cumsum(m["rays"][:-1, 2])]).tolist()
    for i in range(24):
        ref = _scalar_march(ro[i], rd[i], room_bitfield, nears[i], fars[i], noises[i], dt_gamma=dt_gamma)
        off, cnt = m["rays"][i, 1], m["rays"][i, 2]
        assert cnt == len(ref)
        for k, (p, dt, dl) in enumerate(ref):
            assert (m["xyzs"][off + k] == np.asarray(p, np.float32)).all()
            assert m["deltas"][off + k, 0] == dt and m["deltas"][off + k, 1] == dl


def test_march_train_overflow_drops_trailing_rays(room, room_bitfield):
    ro, rd = scene_rays(room, n=32, seed=6)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    full = march.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars)
    M = int(full["total"] * 0.6)
    cut = march.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars, M=M)
    assert (cut["rays"] == full["rays"]).all()
    kept = (full["rays"][:, 1] + full["rays"][:, 2]) <= M
    last = int((full["rays"][kept, 1] + full["rays"][kept, 2]).max())
    assert (cut["xyzs"][:last] == full["xyzs"][:last]).all()
    assert (cut["xyzs"][last:] == 0).all()


def test_march_infer_concatenation_equals_train(room, room_bitfield):
    """Stepping a ray n_step samples at a time visits the same samples as the train march."""
    from oracle import composite
    ro, rd = scene_rays(room, n=16, seed=7)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    full = march.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars)
    N = 16
    alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    got = [[] for _ in range(N)]
    ws, dp, im = np.zeros(N, np.float32), np.zeros(N, np.float32), np.zeros((N, 3), np.float32)
    n_alive = N
    while n_alive:
        x, d, dl = march.march_rays(n_alive, 5, alive, rays_t, ro, rd, room_bitfield, 1.0, 1, 128, nears, fars)
        for n in range(n_alive):
            for s in range(5):
                if dl[n * 5 + s, 0] > 0:
                    got[alive[n]].append(x[n * 5 + s])
        # zero density: nothing terminates early, rays_t advances by sum(deltas[:,1])
        composite.composite_rays(n_alive, 5, alive, rays_t, np.zeros(len(x), np.float32),
                                 np.zeros((len(x), 3), np.float32), dl, ws, dp, im)
        alive = alive[alive >= 0]
        n_alive = len(alive)
    for i in range(N):
        off, cnt = full["rays"][i, 1], full["rays"][i, 2]
        assert len(got[i]) == cnt
        if cnt:
            assert (np.stack(got[i]) == full["xyzs"][off:off + cnt]).all()


# ---------------------------------------------------------------------------------------- fused loader draw (round 6)
def test_loader_pixel_draw_is_a_pure_function_of_seed_step_and_index():
    """oracle/rays.py::sample_pixels restates the counter-based draw of inr_sample_training_batch: reproducible from
    (seed, step), in range, different for other steps and seeds, and pinned by known answers (a change of the hash or of
    the range reduction changes which pixels every seeded training run sees)."""
    from oracle import rays
    a = rays.sample_pixels(7, 3, 4096, 400, 400)
    assert a.dtype == np.int64 and a.min() >= 0 and a.max() < 160000
    assert (a == rays.sample_pixels(7, 3, 4096, 400, 400)).all()
    assert (a != rays.sample_pixels(7, 4, 4096, 400, 400)).mean() > 0.99
    assert (a != rays.sample_pixels(8, 3, 4096, 400, 400)).mean() > 0.99
    assert a[:8].tolist() == [159129, 25489, 140597, 83549, 48411, 138873, 78629, 74064]
    # prefix property: sample k does not depend on n
    assert (rays.sample_pixels(7, 3, 100, 400, 400) == a[:100]).all()
    # uniform over the image: chi-square of 64 equal bins over 65536 draws (63 dof: mean 63, sd 11.2)
    b = rays.sample_pixels(1, 0, 65536, 800, 800)
    counts = np.bincount(b // 10000, minlength=64).astype(np.float64)
    chi2 = ((counts - 1024.0) ** 2 / 1024.0).sum()
    assert chi2 < 63 + 5 * 11.2, chi2
    assert rays.sample_pixels(0, 0, 16, 1, 1).tolist() == [0] * 16


def test_loader_batch_restatement_matches_the_tensor_op_loader():
    """sample_training_batch = the draw + get_rays + two gathers, label rule of masks.labels_for_rays."""
    from oracle import rays
    rng = np.random.default_rng(3)
    H, W, K = 12, 16, 5
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    pose[:3, 3] = rng.normal(size=3)
    img = rng.random((H, W, 3)).astype(np.float32)
    mask = rng.integers(-1, 9, size=(H, W)).astype(np.int32)
    out = rays.sample_training_batch(pose, (20.0, 21.0, 8.0, 6.0), H, W, img, mask, K, seed=5, step=2, n=64)
    inds = out["inds"]
    ref = rays.get_rays(pose[None], (20.0, 21.0, 8.0, 6.0), H, W, inds=inds)
    assert (out["rays_d"] == ref["rays_d"][0]).all() and (out["rays_o"] == ref["rays_o"][0]).all()
    assert (out["rgb"] == img.reshape(-1, 3)[inds]).all()
    lab = mask.reshape(-1)[inds]
    assert (out["labels"] == np.where(lab >= K, -1, lab)).all() and out["labels"].dtype == np.int64
    assert (out["labels"] == -1).any() and (out["labels"] >= 0).any()
