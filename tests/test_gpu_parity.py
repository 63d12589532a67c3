"""HIP path (through the C ABI) vs the CPU oracle and the golden vectors.  Needs an MI355X.

Tolerances: integer / index / sample-position work is BIT-EXACT; floating-point
field outputs are within 1e-3 of the fp32 oracle as BASELINE.json's north_star
states (the assertions below use tighter bounds where the arithmetic allows).
"""
import os

import numpy as np
import pytest
import torch

from conftest import scene_rays

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
def _seeds(n):
    """Seeds of a fuzz test: 0..n-1, or the range INR_FUZZ_SEEDS=lo:hi names (bug hunts beyond the committed set)."""
    spec = os.environ.get("INR_FUZZ_SEEDS")
    if spec:
        lo, hi = (int(v) for v in spec.split(":"))
        return range(lo, hi)
    return range(n)


DEV = "cuda:0"


def _t(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


@pytest.fixture(scope="module")
def rm():
    from instance_nerf_amd import raymarching
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return raymarching


@pytest.fixture(scope="module")
def bits_dev(room_bitfield):
    return _t(room_bitfield)


def test_device_is_gfx950():
    import ctypes
    from instance_nerf_amd import _lib
    props = (ctypes.c_int64 * 4)()
    _lib.check(_lib.load().inr_device_info(0, props))
    assert props[1] == 64 and props[3] == 950, list(props)


# ---------------------------------------------------------------------------- integer work
def test_near_far_bit_exact(rm, room):
    from oracle import rays
    ro, rd = scene_rays(room, n=4096, seed=3)
    ro[:8] = [[0, 0, 0], [0, 0, -3], [0, 5, 0], [.5, .5, .5], [2, 2, 2], [0, 0, 0.99], [-3, 0, 0], [0, 0, 0]]
    rd[:8] = [[0, 0, 1], [0, 0, 1], [1, 0, 0], [-1, 0, 0], [-.6, -.6, -.52915], [0, 1, 0], [1, 0, 0], [0, 1, 0]]
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    n, f = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    gn, gf = rm.near_far_from_aabb(_t(ro), _t(rd), _t(aabb), 0.05)
    assert (gn.cpu().numpy() == n).all() and (gf.cpu().numpy() == f).all()


def test_morton_packbits_bit_exact(rm):
    from oracle import occupancy
    rng = np.random.default_rng(0)
    c = rng.integers(0, 1024, size=(10000, 3)).astype(np.int32)
    m = rm.morton3D(_t(c))
    assert (m.cpu().numpy().astype(np.uint32) == occupancy.morton3D(c)).all()
    assert (rm.morton3D_invert(m).cpu().numpy() == c).all()
    g = rng.normal(size=(2, 4096)).astype(np.float32)
    assert (rm.packbits(_t(g), 0.1).cpu().numpy() == occupancy.packbits(g.ravel(), 0.1)).all()
    assert rm.morton3D(_t(c[:0])).numel() == 0          # empty input


@pytest.fixture(params=["thread_per_ray", "wave_per_ray"])
def marcher(request, rm):
    """Both training marchers (csrc/raymarch.hip: march_ray / march_ray_coop) must produce the same bits."""
    rm.set_march_mode("wave_per_ray" if request.param == "wave_per_ray" else "lane_per_ray")
    yield request.param
    rm.set_march_mode(None)


@pytest.mark.parametrize("tag,gamma", [("g0", 0.0), ("g1", 1.0 / 128)])
def test_march_train_golden_bit_exact(rm, bits_dev, tag, gamma, marcher):
    g = np.load(os.path.join(G, "march.npz"))
    xyzs, dirs, deltas, rays = rm.march_rays_train(_t(g["rays_o"]), _t(g["rays_d"]), 1.0, bits_dev, 1, 128,
                                                   _t(g["nears"]), _t(g["fars"]), dt_gamma=gamma, max_steps=1024,
                                                   noises=_t(g["noises"]))
    assert (rays.cpu().numpy() == g[f"{tag}_rays"]).all()
    M = g[f"{tag}_xyzs"].shape[0]
    assert (xyzs.cpu().numpy()[:M] == g[f"{tag}_xyzs"]).all()
    assert (deltas.cpu().numpy()[:M] == g[f"{tag}_deltas"]).all()
    d = dirs.cpu().numpy()
    for n, off, cnt in g[f"{tag}_rays"]:
        assert (d[off:off + cnt] == g["rays_d"][n]).all()


def test_march_train_large_vs_oracle(rm, room, room_bitfield, bits_dev, marcher):
    """2048 rays across two cameras: counts, offsets and every sample position bit-exact."""
    from oracle import march, rays
    ro = np.concatenate([scene_rays(room, 1024, cam=c, seed=30 + c)[0] for c in (0, 5)])
    rd = np.concatenate([scene_rays(room, 1024, cam=c, seed=30 + c)[1] for c in (0, 5)])
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    noises = np.random.default_rng(9).random(2048).astype(np.float32)
    ref = march.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars, noises, 0.0, 1024)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    xyzs, dirs, deltas, rr = rm.march_rays_train(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars),
                                                 counter, noises=_t(noises))
    assert counter.cpu().tolist() == [ref["total"], 2048]
    assert (rr.cpu().numpy() == ref["rays"]).all()
    assert (xyzs.cpu().numpy()[:ref["total"]] == ref["xyzs"]).all()
    assert (deltas.cpu().numpy()[:ref["total"]] == ref["deltas"]).all()
    # overflow: M below the total drops exactly the trailing rays
    M = int(ref["total"] * 0.5)
    cut = march.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars, noises, 0.0, 1024, M=M)
    x2, _, _, _ = rm.march_rays_train(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars), None, M,
                                      noises=_t(noises))
    assert (x2.cpu().numpy() == cut["xyzs"]).all()


def test_march_train_three_wave_per_ray_regimes(rm, room, room_bitfield, bits_dev):
    """The wave-per-ray marcher has two write passes: up to 8192 rays the count pass parks the samples of its one walk
    in the workspace and the write pass copies them (staged); above that (up to 32768 rays) the write pass walks
    again.  Both, and the lane-per-ray marcher, give the C oracle's bits - also when max_steps exceeds the stage's
    row reservation (1024), which falls back to walking twice."""
    from oracle import c_port
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    for n, max_steps in ((9000, 1024), (3000, 1500), (3000, 1024)):
        ro, rd = scene_rays(room, n, cam=n % 8, seed=n)
        nears, fars = c_port.near_far_from_aabb(ro, rd, aabb, 0.05)
        ref = c_port.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars, None, 0.0, max_steps)
        for mode in ("wave_per_ray", "lane_per_ray"):
            rm.set_march_mode(mode)
            try:
                x, d, dl, rr = rm.march_rays_train(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars),
                                                   max_steps=max_steps, force_all_rays=True)
            finally:
                rm.set_march_mode(None)
            M = ref["total"]
            assert x.shape[0] == M and (rr.cpu().numpy() == ref["rays"]).all(), (n, max_steps, mode)
            assert (x.cpu().numpy() == ref["xyzs"]).all() and (dl.cpu().numpy() == ref["deltas"]).all()
            assert (d.cpu().numpy() == ref["dirs"]).all()


def test_march_train_cascades_and_max_steps(rm, marcher):
    """Two cascades with a growing step (dt_gamma > 0), a random sparse bitfield (long skips that leave a
    64-candidate window, isolated hits) and a max_steps small enough to cut rays short."""
    from oracle import march, rays
    rng = np.random.default_rng(12)
    bits = (rng.random(2 * 64 ** 3 // 8) < 0.3).astype(np.uint8) * rng.integers(1, 256, 2 * 64 ** 3 // 8).astype(np.uint8)
    n = 777
    ro = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd[:5] = np.asarray([[1, 0, 0], [0, -1, 0], [0, 0, 1], [0.6, 0.8, 0], [0, 0.6, -0.8]], np.float32)   # axis-parallel
    aabb = np.asarray([-2, -2, -2, 2, 2, 2], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.2)
    noises = rng.random(n).astype(np.float32)
    for gamma, max_steps in ((1.0 / 128, 1024), (0.0, 1024), (1.0 / 128, 23), (0.0, 64)):
        ref = march.march_rays_train(ro, rd, bits, 2.0, 2, 64, nears, fars, noises, gamma, max_steps)
        x, d, dl, rr = rm.march_rays_train(_t(ro), _t(rd), 2.0, _t(bits), 2, 64, _t(nears), _t(fars),
                                           dt_gamma=gamma, max_steps=max_steps, noises=_t(noises))
        assert (rr.cpu().numpy() == ref["rays"]).all(), (gamma, max_steps)
        assert ref["total"] > 2000
        assert (x.cpu().numpy()[:ref["total"]] == ref["xyzs"]).all()
        assert (dl.cpu().numpy()[:ref["total"]] == ref["deltas"]).all()
        assert (d.cpu().numpy()[:ref["total"]] == ref["dirs"]).all()


@pytest.mark.parametrize("seed", _seeds(12))
def test_march_fuzz_against_the_c_oracle(rm, seed):
    """Random configurations - cascades 1..3 (bound 1, 2, 4), grid 32..128, occupancy 0.5 %..60 %, constant and growing
    steps, max_steps 16..1024, rays from inside and outside the volume, axis-aligned directions (infinite
    reciprocals), jittered starts - all three marchers (lane-per-ray, wave-per-ray, patch writer) against the scalar C
    restatement: counts, offsets, positions and deltas bit for bit."""
    from oracle import c_port
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.integers(1, 4))
    bound = float(2 ** (C - 1))
    H = int(rng.choice([32, 64, 128]))
    fill = float(rng.choice([0.005, 0.05, 0.3, 0.6]))
    bits = (rng.random(C * H ** 3 // 8) < fill).astype(np.uint8) * rng.integers(1, 256, C * H ** 3 // 8).astype(np.uint8)
    dt_gamma = float(rng.choice([0.0, 1.0 / 256, 1.0 / 128, 1.0 / 32]))
    max_steps = int(rng.choice([16, 100, 512, 1024]))
    n = int(rng.choice([1, 15, 16, 17, 333, 1500]))
    ro = rng.uniform(-1.3 * bound, 1.3 * bound, size=(n, 3)).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd[: min(n, 3)] = np.eye(3, dtype=np.float32)[: min(n, 3)] * np.float32(rng.choice([-1.0, 1.0]))
    aabb = np.asarray([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = c_port.near_far_from_aabb(ro, rd, aabb, 0.2)
    noises = rng.random(n).astype(np.float32)
    ref = c_port.march_rays_train(ro, rd, bits, bound, C, H, nears, fars, noises, dt_gamma, max_steps)
    gn, gf = rm.near_far_from_aabb(_t(ro), _t(rd), _t(aabb), 0.2)
    assert (gn.cpu().numpy() == nears).all() and (gf.cpu().numpy() == fars).all()
    for coop in ("lane_per_ray", "wave_per_ray"):
        rm.set_march_mode(coop)
        try:
            xyzs, dirs, deltas, rays = rm.march_rays_train(_t(ro), _t(rd), bound, _t(bits), C, H, gn, gf, dt_gamma=dt_gamma,
                                                           max_steps=max_steps, noises=_t(noises), force_all_rays=True)
        finally:
            rm.set_march_mode(None)
        M = ref["total"]
        assert (rays.cpu().numpy() == ref["rays"]).all(), (coop, C, H, fill, dt_gamma, max_steps, n)
        assert (xyzs.cpu().numpy()[:M] == ref["xyzs"]).all() and (deltas.cpu().numpy()[:M] == ref["deltas"]).all()
        assert (dirs.cpu().numpy()[:M] == ref["dirs"]).all()
    # the frame writer: same samples in the patch-interleaved order (a permutation inside every 16-ray group)
    ref0 = c_port.march_rays_train(ro, rd, bits, bound, C, H, nears, fars, None, dt_gamma, max_steps)
    xp, dp, dlp, rp = rm.march_rays_patch(_t(ro), _t(rd), bound, _t(bits), C, H, gn, gf, dt_gamma, max_steps)
    assert (rp[:, 2].cpu().numpy() == ref0["rays"][:, 2]).all()
    got = xp.cpu().numpy()[: ref0["total"]]
    assert got.shape[0] == ref0["total"]
    key = lambda a: a[np.lexsort(a.T[::-1])]
    assert (key(got) == key(ref0["xyzs"])).all()


def test_march_infer_step_bit_exact(rm, room, room_bitfield, bits_dev):
    from oracle import march, rays
    ro, rd = scene_rays(room, 300, seed=41)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    alive = np.random.default_rng(0).permutation(300)[:200].astype(np.int32)
    rays_t = (nears + np.random.default_rng(1).random(300).astype(np.float32) * 0.4).astype(np.float32)
    x, d, dl = march.march_rays(200, 6, alive, rays_t, ro, rd, room_bitfield, 1.0, 1, 128, nears, fars)
    gx, gd, gdl = rm.march_rays(200, 6, _t(alive), _t(rays_t), _t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears),
                                _t(fars))
    assert (gx.cpu().numpy()[:1200] == x).all() and (gdl.cpu().numpy()[:1200] == dl).all()
    assert (gd.cpu().numpy()[:1200] == d).all()


def test_compact_alive(rm):
    a = np.random.default_rng(0).integers(-1, 50, size=5000).astype(np.int32)
    a[a < 25] = -1
    out, n = rm.compact_alive(_t(a), 5000)
    assert n == int((a >= 0).sum()) and (out.cpu().numpy() == a[a >= 0]).all()
    out, n = rm.compact_alive(_t(np.full(100, -1, np.int32)), 100)
    assert n == 0


# ---------------------------------------------------------------------------- encoders
def _encoder(level_table):
    from instance_nerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(desired_resolution=2048).to(DEV)
    for k in ("offsets", "scales", "resolutions", "hashed"):
        assert (enc.table[k] == level_table[k]).all(), k       # host table == oracle table
    return enc


def test_grid_indices_bit_exact(level_table):
    """Table column 0 = row number (exact in fp32, T < 2^24): at lattice points the encoder
    returns the row the oracle indexes, for dense and hashed levels alike."""
    from oracle import hashgrid
    enc = _encoder(level_table)
    T = level_table["total_rows"]
    emb = torch.zeros(T, 2)
    emb[:, 0] = torch.arange(T, dtype=torch.float32)
    enc.embeddings.data.copy_(emb)
    rng = np.random.default_rng(0)
    for l in (0, 3, 4, 5, 9, 15):
        res = int(level_table["resolutions"][l])
        scale = np.float64(level_table["scales"][l])
        gp = rng.integers(1, res - 1, size=(256, 3))
        x = torch.tensor(((gp - 0.5 + 1e-3) / scale * 2 - 1), dtype=torch.float32).clamp(-1, 1)
        idx, w = hashgrid.corner_indices_weights(x, 1.0, level_table)
        best = w[:, l].argmax(-1)
        assert (w[:, l].max(-1).values > 0.97).all()
        want = idx[torch.arange(256), l, best].float() - float(level_table["offsets"][l]) * 0
        with torch.no_grad():
            got = enc(x.to(DEV))[:, 2 * l].cpu()
        ref = hashgrid.encode(x, emb, 1.0, level_table)[:, 2 * l]
        assert torch.allclose(got, ref, rtol=1e-6, atol=0.5)
        # nearest-row check: blending weight of the other corners is < 3%, rows differ by >= 1
        assert ((got - want).abs() / want.clamp(min=1) < 0.2).float().mean() > 0.9


@pytest.mark.parametrize("seed", _seeds(8))
def test_grid_encoder_fuzz_against_the_c_oracle(seed):
    """Random grid shapes - 2..16 levels, base resolution 4..32, 2^8..2^19 rows per hashed level, finest resolution
    64..4096, bound 1..4 (dense-only, mixed and hashed-only tables) - stand-alone encoder forward and table gradient
    against the scalar C / numpy restatements, points on the faces and outside the volume included."""
    from instance_nerf_amd.gridencoder import GridEncoder
    from oracle import c_port, hashgrid
    rng = np.random.default_rng(500 + seed)
    L = int(rng.choice([2, 5, 8, 13, 16]))
    base = int(rng.choice([4, 16, 32]))
    log2_t = int(rng.choice([8, 12, 15, 19]))
    res = int(rng.choice([64, 512, 2048, 4096]))
    bound = float(rng.choice([1.0, 2.0, 4.0]))
    enc = GridEncoder(num_levels=L, base_resolution=base, log2_hashmap_size=log2_t, desired_resolution=res).to(DEV)
    tb = hashgrid.level_table(num_levels=L, base_resolution=base, log2_hashmap_size=log2_t, desired_resolution=res)
    for k in ("offsets", "scales", "resolutions", "hashed"):
        assert (enc.table[k] == tb[k]).all(), k
    gen = torch.Generator().manual_seed(seed)
    emb = torch.rand(tb["total_rows"], 2, generator=gen) * 2 - 1
    enc.embeddings.data.copy_(emb)
    x = (torch.rand(3000, 3, generator=gen) * 2.2 - 1.1) * bound
    x[:4] = torch.tensor([[1.0, 1, 1], [-1.0, -1, -1], [1.0, -1, 0.3], [0.0, 0, 0]]) * bound
    out = enc(x.to(DEV), bound=bound)
    ref = c_port.grid_encode(x.numpy(), emb.numpy(), bound, tb)
    assert out.shape == (3000, 2 * L)
    assert np.abs(out.detach().cpu().numpy() - ref).max() < 2e-6, (L, base, log2_t, res, bound)
    go = torch.randn(3000, 2 * L, generator=gen)
    out.backward(go.to(DEV))
    g_ref = hashgrid.encode_backward_table(x, go, bound, tb)
    # fp32 atomics land in any order: a row of a coarse level collects hundreds of O(1) terms of both signs, so the
    # rounding of its sum is bounded by (terms x eps x sum |term|), not by the (cancelled) sum itself - compare against
    # the row's absolute mass (seed 2: 5 levels from resolution 4, ~190 terms per coarse row, failed 5e-5 absolute once
    # in round 5 with unchanged kernels)
    mass = hashgrid.encode_backward_table(x, go.abs(), bound, tb)
    err = (enc.embeddings.grad.cpu() - g_ref).abs()
    assert bool((err <= 2e-5 + 2e-6 * mass).all()), float((err / (2e-5 + 2e-6 * mass)).max())


def test_grid_encode_input_gradient(level_table, params_k16):
    """Positions that require grad (upstream's dy_dx path): d encode / dx against the oracle's autograd through the
    interpolation weights; out-of-range points get zero; the table gradient of the same backward is unchanged."""
    from oracle import hashgrid
    enc = _encoder(level_table)
    enc.embeddings.data.copy_(params_k16["embeddings"])
    gen = torch.Generator().manual_seed(8)
    x = torch.rand(2000, 3, generator=gen) * 2.1 - 1.05
    go = torch.randn(2000, 32, generator=gen)
    xd = x.to(DEV).requires_grad_(True)
    enc(xd).backward(go.to(DEV))
    ref = hashgrid.encode_input_grad(x, go, params_k16["embeddings"], 1.0, level_table)
    inside = ((x >= -1) & (x <= 1)).all(-1)
    assert (xd.grad.cpu()[~inside] == 0).all() and inside.float().mean() > 0.7
    scale = ref.abs().max()
    assert (xd.grad.cpu() - ref).abs().max() < 1e-4 * scale, ((xd.grad.cpu() - ref).abs().max(), scale)
    assert torch.allclose(enc.embeddings.grad.cpu(), hashgrid.encode_backward_table(x, go, 1.0, level_table), atol=2e-5, rtol=1e-4)


def test_grid_encode_forward_golden(level_table, params_k16):
    g = np.load(os.path.join(G, "field.npz"))
    enc = _encoder(level_table)
    enc.embeddings.data.copy_(params_k16["embeddings"])
    with torch.no_grad():
        out = enc(_t(g["x"])).cpu().numpy()
    assert np.abs(out - g["enc"]).max() < 1e-5
    with torch.no_grad():
        assert enc(_t(g["x"][:0])).shape == (0, 32)            # empty input


def test_grid_encode_backward_vs_oracle(level_table, params_k16):
    from oracle import hashgrid
    enc = _encoder(level_table)
    enc.embeddings.data.copy_(params_k16["embeddings"])
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(3000, 3, generator=gen) * 2 - 1
    go = torch.randn(3000, 32, generator=gen)
    out = enc(x.to(DEV))
    out.backward(go.to(DEV))
    ref = hashgrid.encode_backward_table(x, go, 1.0, level_table)
    got = enc.embeddings.grad.cpu()
    assert (got != 0).sum() > 100000
    assert torch.allclose(got, ref, atol=2e-5, rtol=1e-4)      # float atomics: order-dependent rounding


def test_points_outside_the_grid(level_table, params_k16):
    """Regression (round-1 advisor): coordinates outside [-bound, bound] (and NaN) used to index foreign rows on
    dense levels and scatter gradients into them.  Now, as upstream's flag_oob: zero features, no gradient - in the
    stand-alone encoder, the fused no-grad field kernels and both fused training paths."""
    from oracle import field, hashgrid
    enc = _encoder(level_table)
    enc.embeddings.data.copy_(params_k16["embeddings"])
    gen = torch.Generator().manual_seed(11)
    x = torch.rand(4000, 3, generator=gen) * 3 - 1.5                       # ~70 % of the points are outside
    x[:6] = torch.tensor([[1.0, -1.0, 1.0], [1.000001, 0, 0], [0, -40.0, 0], [float("nan"), 0, 0],
                          [3e38, 0, 0], [-1.0, -1.0, -1.0]])
    inside = ((x >= -1) & (x <= 1)).all(-1)
    assert 0.1 < inside.float().mean() < 0.6
    go = torch.randn(4000, 32, generator=gen)
    out = enc(x.to(DEV))
    out.backward(go.to(DEV))
    ref = hashgrid.encode(x, params_k16["embeddings"], 1.0, level_table)
    got = out.detach().cpu()
    assert (got[~inside] == 0).all()
    assert torch.allclose(got, ref, atol=1e-5)
    ref_g = hashgrid.encode_backward_table(x, go, 1.0, level_table)
    assert torch.allclose(enc.embeddings.grad.cpu(), ref_g, atol=2e-5, rtol=1e-4)
    # fused field kernels, no grad
    net = _network(params_k16, K=16).eval()
    d = torch.nn.functional.normalize(torch.randn(4000, 3, generator=gen), dim=-1)
    with torch.no_grad():
        sigma, rgb = net(x.to(DEV), d.to(DEV))
        den = net.density(x.to(DEV))
        logits = net.instance(x.to(DEV))
        rs, rc = field.nerf_forward(x, d, params_k16, 1.0, level_table)
        rl = field.instance_logits(x, params_k16, 1.0, level_table)
    assert torch.allclose(sigma.cpu(), rs, rtol=1e-4, atol=1e-6) and (sigma.cpu()[~inside] == 1).all()
    assert (rgb.cpu() - rc).abs().max() < 1e-5 and (den["geo_feat"].cpu()[~inside] == 0).all()
    assert (logits.cpu() - rl).abs().max() < 1e-4 and (logits.cpu()[~inside] == 0).all()
    # fused training paths: table gradients come from the inside points only
    net.train()
    sigma, rgb = net(x.to(DEV), d.to(DEV))
    (sigma.clamp(max=10).sum() + rgb.sum()).backward()
    logits = net.instance(x.to(DEV))
    logits.square().sum().backward()
    p = {k: v.clone().requires_grad_(True) for k, v in params_k16.items()}
    rs, rc = field.nerf_forward(x, d, p, 1.0, level_table)
    (rs.clamp(max=10).sum() + rc.sum()).backward()
    field.instance_logits(x, p, 1.0, level_table).square().sum().backward()
    for name, ref_t in (("encoder.embeddings", p["embeddings"]), ("instance_encoder.embeddings", p["inst_embeddings"])):
        got_g = dict(net.named_parameters())[name].grad.cpu()
        # (a ReLU pre-activation within rounding of zero may fall on either side: norm-wise comparison)
        assert torch.linalg.norm(got_g - ref_t.grad) < 5e-3 * torch.linalg.norm(ref_t.grad), name
        assert (got_g - ref_t.grad).abs().max() < 2e-2 * ref_t.grad.abs().max(), name
        assert ((got_g != 0) == (ref_t.grad != 0)).float().mean() > 0.9999, name


def test_sh_forward_backward(level_table):
    from instance_nerf_amd.shencoder import SHEncoder
    from oracle import sh
    g = np.load(os.path.join(G, "field.npz"))
    d = torch.tensor(g["d"], requires_grad=True)
    ref = sh.sh_encode(d)
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(0))
    ref.backward(go)
    dd = _t(g["d"]).requires_grad_(True)
    out = SHEncoder()(dd)
    out.backward(go.to(DEV))
    assert np.abs(out.detach().cpu().numpy() - g["sh"]).max() < 1e-6
    assert torch.allclose(dd.grad.cpu(), d.grad, atol=1e-5)


# ---------------------------------------------------------------------------- fused field (MFMA)
def _network(params, K=16, **kw):
    from instance_nerf_amd.nerf import NeRFNetwork
    net = NeRFNetwork(cuda_ray=True, num_instances=K, min_near=0.05, **kw).to(DEV)
    sd = {"encoder.embeddings": params["embeddings"], "sigma_net.0.weight": params["sigma_w0"],
          "sigma_net.1.weight": params["sigma_w1"], "color_net.0.weight": params["color_w0"],
          "color_net.1.weight": params["color_w1"], "color_net.2.weight": params["color_w2"]}
    if K:
        sd.update({"instance_encoder.embeddings": params["inst_embeddings"], "instance_net.0.weight": params["inst_w0"],
                   "instance_net.1.weight": params["inst_w1"], "instance_net.2.weight": params["inst_w2"]})
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected
    return net


def test_fused_field_golden(params_k16):
    g = np.load(os.path.join(G, "field.npz"))
    net = _network(params_k16).eval()
    with torch.no_grad():
        sigma, rgb = net(_t(g["x"]), _t(g["d"]))
        den = net.density(_t(g["x"]))
        logits = net.instance(_t(g["x"]))
    assert np.allclose(sigma.cpu().numpy(), g["sigma"], rtol=1e-4, atol=1e-6)
    assert np.abs(rgb.cpu().numpy() - g["rgb"]).max() < 1e-5
    assert np.allclose(den["sigma"].cpu().numpy(), g["sigma"], rtol=1e-4, atol=1e-6)
    assert np.abs(den["geo_feat"].cpu().numpy() - g["geo"]).max() < 1e-4
    assert np.abs(logits.cpu().numpy() - g["logits"]).max() < 1e-4


def test_half_precision_table_is_the_fp32_path_on_rounded_values(params_k16, room, room_bitfield):
    """NeRFNetwork.half_table (opt-in; upstream's -O / fp16 storage; Trainer(fp16=True) switches it on for evaluation):
    the eval kernel gathers from a half-precision copy of the table.  Everything else is unchanged, so the frame is
    BIT-IDENTICAL to the fp32 path run on a table whose values were rounded to fp16 first - and within 3e-3 of the
    unrounded one (11 significant bits per table value)."""
    from instance_nerf_amd.nerf.utils import Trainer, get_rays
    poses, intr, H, W = room.cameras(n=1, H=96, W=96, focal=48.0)
    r = get_rays(_t(poses[:1]), intr, 96, 96, patch=4)

    def frame(net):
        with torch.no_grad():
            return net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
    net = _network({k: v.clone() for k, v in params_k16.items()}, K=0).eval()
    net.density_bitfield.copy_(_t(room_bitfield))
    full = frame(net)
    net.half_table = True
    half = frame(net)
    assert int(half["num_samples"][0]) == int(full["num_samples"][0])
    d = (half["image"] - full["image"]).abs().max()
    assert 0 < float(d) < 3e-3, float(d)
    rounded = _network({k: v.clone() for k, v in params_k16.items()}, K=0).eval()
    rounded.density_bitfield.copy_(_t(room_bitfield))
    with torch.no_grad():
        rounded.encoder.embeddings.copy_(rounded.encoder.embeddings.half().float())
    ref = frame(rounded)
    assert torch.equal(half["image"], ref["image"]) and torch.equal(half["depth"], ref["depth"])
    # the copy follows the master table
    with torch.no_grad():
        net.encoder.embeddings.mul_(0.5)
    again = frame(net)
    assert not torch.equal(again["image"], half["image"])
    # upstream's flag
    tr = Trainer("h", None, _network(params_k16, K=0), stage="nerf", device=torch.device(DEV), fp16=True, workspace=None)
    assert tr.model.half_table and tr.model.mlp_fp16 and tr.fp16


def test_single_pass_fp16_mlp_is_the_opt_in_fast_path(level_table, room, room_bitfield):
    """NeRFNetwork.mlp_fp16 (opt-in, inference; with half_table the two halves of upstream's -O): the MLP GEMMs take ONE
    fp16 MFMA pass with fp32 accumulation.  Same samples, an image within 3e-3 (6e-3 on the fp16 table) of the default
    path's and NOT equal to it (fp16 operands: 2^-12 relative), at least 50 dB on O(1) densities and colours;
    training renders and the default path are untouched; on both table formats."""
    from instance_nerf_amd.nerf.utils import get_rays
    poses, intr, H, W = room.cameras(n=1, H=96, W=96, focal=48.0)
    r = get_rays(_t(poses[:1]), intr, 96, 96, patch=4)
    from oracle import field
    net = _network(field.init_params(seed=41, table=level_table, table_std=1.0, K=0), K=0).eval()   # O(1) outputs
    net.density_bitfield.copy_(_t(room_bitfield))

    def frame():
        with torch.no_grad():
            return net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
    full = frame()
    assert 0.05 < float(full["weights_sum"].mean()) < 0.999
    for half in (False, True):
        net.half_table, net.mlp_fp16 = half, True
        fast = frame()
        assert int(fast["num_samples"][0]) == int(full["num_samples"][0])
        d = (fast["image"] - full["image"]).abs()
        psnr = -10 * np.log10(float((d ** 2).mean()))
        print(f"fp16 MLP, half table {half}: max abs {float(d.max()):.2e}, {psnr:.1f} dB")
        assert 0 < float(d.max()) < (6e-3 if half else 3e-3) and psnr > 50, (half, float(d.max()), psnr)
    # the early-terminating kernel takes the same numerics when both options are on (density x300: rays do terminate)
    net.half_table = net.mlp_fp16 = False
    net.density_scale = 300.0
    with torch.no_grad():
        t_full = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused_terminate")
        net.half_table = net.mlp_fp16 = True
        t_fast = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused_terminate")
        two = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
    assert "num_evaluated" in t_fast and int(t_fast["num_evaluated"][0]) < int(t_fast["num_samples"][0])
    d = float((t_fast["image"] - t_full["image"]).abs().max())
    assert 0 < d < 3e-3, d
    assert float((t_fast["image"] - two["image"]).abs().max()) < 2e-4       # both -O paths agree up to T_thresh effects
    net.density_scale = 1.0
    net.half_table = net.mlp_fp16 = False
    assert torch.equal(frame()["image"], full["image"])
    net.mlp_fp16 = True
    net.train()
    with torch.no_grad():
        a = net.render(r["rays_o"][:, :256], r["rays_d"][:, :256], bg_color=1, perturb=False, force_all_rays=True)["image"]
    net.mlp_fp16 = False
    with torch.no_grad():
        b = net.render(r["rays_o"][:, :256], r["rays_d"][:, :256], bg_color=1, perturb=False, force_all_rays=True)["image"]
    assert torch.equal(a, b)                           # the training path never takes the fast kernel


def test_instance_render_with_O_numerics(level_table, room, room_bitfield):
    """half_table + mlp_fp16 on a network with an instance head: the rendered logits come from k_instance_render<K, true>
    (fp16 copy of the instance table, one fp16 MFMA pass per layer) - same samples, logits within 2e-3 of the default
    path's relative to their size and not identical; K = 64 and a padded K = 20."""
    from instance_nerf_amd.nerf.utils import get_rays
    from oracle import field
    poses, intr, H, W = room.cameras(n=1, H=64, W=64, focal=32.0)
    r = get_rays(_t(poses[:1]), intr, 64, 64, patch=4)
    for K in (64, 20):
        net = _network(field.init_params(seed=43, table=level_table, table_std=1.0, K=K), K=K).eval()
        net.density_bitfield.copy_(_t(room_bitfield))

        def frame():
            with torch.no_grad():
                return net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
        full = frame()
        net.half_table = net.mlp_fp16 = True
        fast = frame()
        assert fast["instance"].shape == full["instance"].shape == (1, 64 * 64, K)
        assert int(fast["num_samples"][0]) == int(full["num_samples"][0])
        d = float((fast["instance"] - full["instance"]).abs().max())
        scale = float(full["instance"].abs().max())
        assert 0 < d < 2e-3 * scale, (K, d, scale)
        net.half_table = net.mlp_fp16 = False
        assert torch.equal(frame()["instance"], full["instance"])


def test_frozen_nerf_of_the_instance_stage_takes_O_numerics_when_asked(level_table, room, room_bitfield):
    """half_table + mlp_fp16 also cover a NeRF that is only EVALUATED during training - the frozen NeRF of the instance
    stage (inr_nerf_forward_fast) - and nothing that is trained: instance-stage renders move by a little (and the
    instance gradients with them), NeRF-stage training renders and gradients keep their bits."""
    from oracle import field
    p = field.init_params(seed=47, table=level_table, table_std=1.0, K=16)
    ro, rd = scene_rays(room, 600, cam=2, seed=61)
    labels = _t(np.random.default_rng(3).integers(-1, 16, size=(1, 600))).long()

    def run(stage, flags):
        net = _network({k: v.clone() for k, v in p.items()}, K=16).train()
        net.density_bitfield.copy_(_t(room_bitfield))
        net.half_table = net.mlp_fp16 = flags
        if stage == "instance":
            net.freeze_nerf()
            out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=True, ce_labels=labels)
            out["instance_ce"].backward()
        else:
            out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=True)
            (out["image"] ** 2).mean().backward()
        grads = {k: q.grad.clone() for k, q in net.named_parameters() if q.grad is not None}
        return out["image"].detach().clone(), grads
    img_a, ga = run("instance", False)
    img_b, gb = run("instance", True)
    d = float((img_a - img_b).abs().max())
    assert 0 < d < 3e-3, d
    for k in ga:
        rel = float(torch.linalg.norm(ga[k] - gb[k]) / torch.linalg.norm(ga[k]))
        assert rel < 2e-2, (k, rel)
    img_c, gc = run("nerf", False)
    img_d, gd = run("nerf", True)
    assert torch.equal(img_c, img_d)
    for k in gc:
        rel = float(torch.linalg.norm(gc[k] - gd[k]) / torch.linalg.norm(gc[k]))
        assert rel < 2e-5, (k, rel)              # the scatter's atomics round in launch order; nothing else differs


def test_device_packers_equal_the_host_packers(level_table):
    """NeRFNetwork._packed_weights packs device-resident weights ON the device (no device->host copies: the occupancy
    update of the NeRF stage re-packs every 16 steps): the forward image has the bits of the host packers', for both
    fields and a padded K."""
    from instance_nerf_amd import _lib
    from oracle import field
    lib = _lib.load()
    for K in (64, 20):
        net = _network(field.init_params(seed=53, table=level_table, table_std=1.0, K=K), K=K).eval()
        for which in ("nerf", "instance"):
            net._packed.clear()
            net._host_pack_only = False
            dev_img = net._packed_weights(which).clone()
            net._packed.clear()
            net._host_pack_only = True
            host_img = net._packed_weights(which)
            assert dev_img.shape == host_img.shape and torch.equal(dev_img.view(torch.int32), host_img.view(torch.int32)), (K, which)
        net._host_pack_only = False


def test_exact_fp32_mlp_build(params_k16):
    """The -DINR_MLP_FP32=1 build (MLP GEMMs on v_mfma_f32_16x16x4_f32, exact fp32 products) stays alive: the same
    golden field vectors through libinr_hip_fp32.so in a child process (a process binds one library).  Both builds
    meet the golden tolerances; the exact build is at fp32 rounding of the oracle, the default bf16x3 split within
    2^-16-class error of it."""
    import subprocess
    import sys
    from instance_nerf_amd import build
    lib = build.LIB_FP32
    assert os.path.exists(lib), "libinr_hip_fp32.so is built by __graft_entry__.build()"
    code = f"""
import os, sys, numpy as np, torch
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
from test_gpu_parity import _network, _t, G
from oracle.field import init_params
from oracle.hashgrid import level_table
g = np.load(os.path.join(G, "field.npz"))
net = _network(init_params(seed=0, table=level_table(), table_std=1.0, K=16)).eval()
with torch.no_grad():
    sigma, rgb = net(_t(g["x"]), _t(g["d"]))
    logits = net.instance(_t(g["x"]))
print("ERR", float(np.abs(sigma.cpu().numpy() / g["sigma"] - 1).max()), float(np.abs(rgb.cpu().numpy() - g["rgb"]).max()),
      float(np.abs(logits.cpu().numpy() - g["logits"]).max()))
"""
    errs = {}
    for name, path in (("fp32", lib), ("bf16x3", build.LIB)):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, INR_LIB_PATH=path), capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        errs[name] = [float(v) for v in [l for l in r.stdout.splitlines() if l.startswith("ERR")][-1].split()[1:]]
    print("sigma rel / rgb abs / logits abs:", errs)
    assert errs["fp32"][0] < 2e-5 and errs["fp32"][1] < 2e-6 and errs["fp32"][2] < 2e-5, errs
    assert errs["bf16x3"][0] < 1e-4 and errs["bf16x3"][1] < 1e-5 and errs["bf16x3"][2] < 1e-4, errs


def test_training_gradients_under_the_exact_fp32_build():
    """Round-4 verdict item 5: the -DINR_MLP_FP32=1 build now covers the TRAINING entry points (device weight packers,
    k_nerf_head_bwd, k_instance_head_bwd on v_mfma_f32_16x16x4_f32), so the gradient fuzz can say which part of its
    2e-2 norm-wise tolerance belongs to the split-bf16 MLP: the same eight set-ups (both stages, K = 64 / 31 / 16 / 5)
    (plus the eight off the tuned configuration: bound 1 / 2 / 4, table sizes, 12 / 16 levels) through both libraries,
    each in a child process (a process binds one library).  Exact fp32: every gradient tensor within 1e-4 norm-wise of
    torch autograd through the oracle (measured: 4e-6 worst) - what is left is fp32 summation order, the
    trunc-exp / sigmoid hardware approximations (1e-7 class) and the rare ReLU pre-activation that fp32 rounding itself
    puts on the other side of zero.  The default build on the SAME inputs is reported next to it."""
    import json
    import subprocess
    import sys
    from instance_nerf_amd import build
    assert os.path.exists(build.LIB_FP32), "libinr_hip_fp32.so is built by __graft_entry__.build()"
    code = f"""
import json, os, sys
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})
from test_gpu_parity import _gradient_fuzz_case
out = {{}}
for varied in (False, True):
    for seed in range(8):
        got = {{}}
        _gradient_fuzz_case(seed, varied=varied, collect=got)
        out.update({{f"{{'v' if varied else 't'}}{{seed}}:{{st}}:{{k}}": v for (st, k), v in got.items()}})
print("ERRS", json.dumps(out))
"""
    errs = {}
    for name, path in (("fp32", build.LIB_FP32), ("bf16x3", build.LIB)):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, INR_LIB_PATH=path), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        errs[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith("ERRS")][-1][5:])
    worst = {n: max(v[0] for v in e.values()) for n, e in errs.items()}
    print("worst norm-wise gradient error per build:", worst)
    print({k: (errs["fp32"][k][0], errs["bf16x3"][k][0], errs["fp32"][k][1]) for k in sorted(errs["fp32"])})
    assert len(errs["fp32"]) == 2 * (4 * 6 + 4 * 4)                # tuned + varied: four NeRF-stage set-ups x 6 tensors, four instance x 4
    assert worst["fp32"] < 1e-4, {k: v for k, v in errs["fp32"].items() if v[0] >= 1e-4}
    assert worst["bf16x3"] < 2e-2


def test_fused_equals_unfused_and_ragged_sizes(params_k16):
    """The MFMA path and the encoder+rocBLAS path agree; sizes that are not tile multiples work."""
    net = _network(params_k16)
    gen = torch.Generator().manual_seed(11)
    for M in (1, 15, 16, 17, 1000, 4097):
        x = (torch.rand(M, 3, generator=gen) * 2 - 1).to(DEV)
        d = torch.nn.functional.normalize(torch.randn(M, 3, generator=gen), dim=1).to(DEV)
        with torch.no_grad():
            s0, c0 = net(x, d)
            l0 = net.instance(x)
        s1, c1 = net(x, d)                      # grad enabled -> unfused path
        l1 = net.instance(x)
        assert s1.requires_grad and not s0.requires_grad
        assert torch.allclose(s0, s1.detach(), rtol=1e-4, atol=1e-6)
        assert (c0 - c1.detach()).abs().max() < 1e-5
        assert (l0 - l1.detach()).abs().max() < 1e-4
    with torch.no_grad():
        s, c = net(torch.zeros(0, 3, device=DEV), torch.zeros(0, 3, device=DEV))
    assert s.shape == (0,) and c.shape == (0, 3)


# ---------------------------------------------------------------------------- compositing
def test_composite_train_golden(rm):
    g = np.load(os.path.join(G, "composite.npz"))
    s = _t(g["sigmas"]).requires_grad_(True)
    c = _t(g["rgbs"]).requires_grad_(True)
    e = _t(g["extra"]).requires_grad_(True)
    ws, depth, img, ex = rm.composite_rays_train(s, c, _t(g["deltas"]), _t(g["rays"]), 1e-4, extra=e)
    assert np.abs(ws.detach().cpu().numpy() - g["weights_sum"]).max() < 1e-5
    assert np.abs(img.detach().cpu().numpy() - g["image"]).max() < 1e-5
    assert np.abs(depth.cpu().numpy() - g["depth"]).max() < 1e-5
    assert np.abs(ex.detach().cpu().numpy() - g["extra_out"]).max() < 1e-4
    ((ws * _t(g["g_ws"])).sum() + (img * _t(g["g_img"])).sum() + (ex * _t(g["g_extra"])).sum()).backward()
    assert np.allclose(s.grad.cpu().numpy(), g["grad_sigmas"], atol=3e-5, rtol=1e-3)
    assert np.abs(c.grad.cpu().numpy() - g["grad_rgbs"]).max() < 1e-5
    assert np.abs(e.grad.cpu().numpy() - g["grad_extra"]).max() < 1e-5
    # no extra channels / zero-count rays only
    ws2, _, img2 = rm.composite_rays_train(s.detach(), c.detach(), _t(g["deltas"]), _t(g["rays"]), 1e-4)
    assert torch.equal(ws2, ws.detach()) and torch.equal(img2, img.detach())


# ---------------------------------------------------------------------------- end to end
@pytest.mark.parametrize("mode", ["fused", "wavefront"])
def test_render_infer_golden(params_k16, room_bitfield, mode):
    g = np.load(os.path.join(G, "render.npz"))
    net = _network(params_k16).eval()
    net.density_bitfield.copy_(_t(room_bitfield))
    with torch.no_grad():
        out = net.render(_t(g["rays_o"])[None], _t(g["rays_d"])[None], staged=False, bg_color=1, perturb=False,
                         infer_mode=mode)
    assert np.abs(out["image"][0].cpu().numpy() - g["infer_image"]).max() < 1e-4
    assert np.abs(out["weights_sum"][0].cpu().numpy() - g["infer_ws"]).max() < 1e-4
    assert np.abs(out["instance"][0].cpu().numpy() - g["infer_instance"]).max() < 1e-3
    # depth over the ABSOLUTE ray parameter, as upstream's inference compositing (and the oracle's restatement of its
    # loop) accumulates it - the one-pass modes add the start parameter back (round 2: they used to return the
    # training-style value)
    assert np.abs(out["depth"][0].cpu().numpy() - g["infer_depth"]).max() < 1e-4
    if mode == "fused":
        assert int(out["num_samples"][0]) == int(g["train_total"])


def test_render_train_golden_and_gradients(params_k16, room, room_bitfield, level_table):
    """Instance-field training step: rendered logits and table/MLP gradients match the oracle."""
    from oracle import render
    g = np.load(os.path.join(G, "render.npz"))
    net = _network(params_k16).train()
    net.density_bitfield.copy_(_t(room_bitfield))
    net.freeze_nerf()
    out = net.render(_t(g["rays_o"])[None], _t(g["rays_d"])[None], bg_color=1, perturb=False, force_all_rays=True)
    assert np.abs(out["image"][0].detach().cpu().numpy() - g["train_image"]).max() < 1e-4
    assert np.abs(out["instance"][0].detach().cpu().numpy() - g["train_instance"]).max() < 1e-3
    _, labels, _ = room.trace(g["rays_o"], g["rays_d"])
    labels = np.where(np.arange(len(labels)) % 7 == 0, -1, labels % 16)
    loss = torch.nn.functional.cross_entropy(out["instance"][0], _t(labels).long(), ignore_index=-1)
    loss.backward()
    p = {k: v.clone() for k, v in params_k16.items()}
    for k in ("inst_embeddings", "inst_w0", "inst_w1", "inst_w2"):
        p[k].requires_grad_(True)
    ref = render.render_train(g["rays_o"], g["rays_d"], p, level_table, room_bitfield, min_near=0.05,
                              with_instance=True)
    rl = render.instance_ce_loss(ref["instance"], labels)
    rl.backward()
    assert abs(loss.item() - rl.item()) < 1e-4
    assert torch.allclose(net.instance_net[2].weight.grad.cpu(), p["inst_w2"].grad, atol=1e-4, rtol=1e-3)
    assert torch.allclose(net.instance_net[0].weight.grad.cpu(), p["inst_w0"].grad, atol=1e-4, rtol=1e-3)
    ge, re_ = net.instance_encoder.embeddings.grad.cpu(), p["inst_embeddings"].grad
    assert (re_ != 0).sum() > 1000
    assert torch.allclose(ge, re_, atol=1e-5, rtol=1e-3)
    assert net.encoder.embeddings.grad is None            # NeRF stayed frozen


@pytest.mark.parametrize("K,N", [(64, 4096), (16, 777), (37, 5), (64, 0)])
def test_cross_entropy_matches_torch(K, N):
    """The instance stage's loss (mean CE, ignore_index -1) - value and gradient - against F.cross_entropy: ragged
    sizes, K not a power of two, every row ignored (NaN, as torch), an upstream gradient other than 1."""
    from instance_nerf_amd import raymarching
    gen = torch.Generator().manual_seed(K + N)
    logits = (torch.randn(N, K, generator=gen) * 4).to(DEV)
    labels = torch.randint(0, K, (N,), generator=gen)
    labels[torch.rand(N, generator=gen) < 0.2] = -1
    labels = labels.to(DEV)
    a = logits.clone().requires_grad_(True)
    b = logits.clone().requires_grad_(True)
    la = raymarching.cross_entropy(a, labels, ignore_index=-1)
    lb = torch.nn.functional.cross_entropy(b, labels, ignore_index=-1)
    if N == 0 or (labels >= 0).sum() == 0:
        assert torch.isnan(la) and torch.isnan(lb)
        return
    assert abs(float(la) - float(lb)) < 1e-5 * max(1.0, abs(float(lb)))
    (la * 2.5).backward()
    (lb * 2.5).backward()
    assert (a.grad - b.grad).abs().max() < 1e-6
    assert (a.grad[labels < 0] == 0).all()
    la2 = raymarching.cross_entropy(logits, torch.full_like(labels, -1))
    assert torch.isnan(la2)


def test_adam_matches_torch():
    import ctypes
    from instance_nerf_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(0)
    for n in (4096, 1003):
        p0 = torch.randn(n, generator=gen)
        ref = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for step in range(1, 4):
            g = torch.randn(n, generator=gen)
            ref.grad = g.clone()
            opt.step()
            _lib.check(lib.inr_adam_step(_lib.ptr(p), _lib.ptr(g.to(DEV)), _lib.ptr(m), _lib.ptr(v), n, 1e-2, 0.9,
                                         0.99, 1e-15, step, 1.0, _lib.stream_ptr()))
        assert torch.allclose(p.cpu(), ref.detach(), atol=1e-6, rtol=1e-5)


def test_occupancy_update_matches_oracle(params_k16, level_table):
    """update_extra_state with the jitter disabled equals the oracle's grid/bitfield."""
    from oracle import field, occupancy
    net = _network(params_k16, K=0).eval()
    torch.manual_seed(0)
    import instance_nerf_amd.nerf.renderer as R
    orig = torch.rand_like
    try:
        torch.rand_like = lambda t: torch.full_like(t, 0.5)       # jitter (2*0.5-1) = 0
        net.update_extra_state()
    finally:
        torch.rand_like = orig
    with torch.no_grad():
        sig = lambda xyz: field.density(torch.from_numpy(xyz), params_k16, 1.0, level_table)["sigma"].numpy()
        grid, bits, mean = occupancy.update_density_grid(np.zeros((1, 128 ** 3), np.float32), sig, 128, 1, 1.0,
                                                         density_thresh=net.density_thresh)
    got = net.density_grid.cpu().numpy()
    assert np.allclose(got, grid, rtol=2e-4, atol=1e-6)
    assert abs(net.mean_density - mean) / mean < 1e-4
    # cells within float noise of the threshold may flip; everything else must agree bit for bit
    thr = min(mean, net.density_thresh)
    near = np.abs(grid.ravel() - thr) < 1e-3 * thr
    diff = np.unpackbits(net.density_bitfield.cpu().numpy(), bitorder="little") != np.unpackbits(bits, bitorder="little")
    assert not (diff & ~near).any()


def test_occupancy_update_reads_back_the_sample_count_first_and_the_mean_on_demand(params_k16):
    """update_extra_state sets ``mean_count`` from the counters of the steps before it (upstream: the mean of
    step_counter[:total_step, 0]) and leaves the mean density on the device until ``mean_density`` is read; reading it
    gives the number the bit field's threshold was formed from, and setting it (checkpoint load) overrides a pending one."""
    net = _network(params_k16, K=0).train()
    totals = [1000, 3000, 2001]
    for i, t in enumerate(totals):
        net.step_counter[i, 0] = t
    net.local_step = len(totals)
    net.update_extra_state()
    assert net.mean_count == int(sum(totals) / len(totals)) and net.local_step == 0
    pending = net.__dict__["_mean_density_dev"]
    assert pending is not None and pending.is_cuda
    expect = float(net.density_grid.clamp(min=0).double().mean())
    assert abs(net.mean_density - expect) <= 1e-5 * expect
    assert net.__dict__["_mean_density_dev"] is None and isinstance(net.mean_density, float)
    assert float(pending[1]) == sum(totals)                # the kernel's own total agrees with the early read-back
    net.update_extra_state()                               # no step in between: the count stays, the mean is pending
    assert net.mean_count == int(sum(totals) / len(totals))
    net.mean_density = 0.25
    assert net.mean_density == 0.25 and net.__dict__["_mean_density_dev"] is None


def test_occupancy_update_two_cascades_with_unseen_cells():
    """bound = 2 (two cascades, 64^3 grid, desired resolution 4096), cells marked -1 beforehand, density_scale != 1:
    the full sweep of update_extra_state (jitter disabled) against the oracle's update - grid, mean, bitfield."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from oracle import field, hashgrid, occupancy
    tb = hashgrid.level_table(desired_resolution=4096)
    p = field.init_params(seed=3, table=tb, table_std=1.0)
    net = NeRFNetwork(cuda_ray=True, bound=2, min_near=0.2, grid_size=64, density_scale=0.7, density_thresh=2.0).to(DEV).eval()
    net.load_state_dict({"encoder.embeddings": p["embeddings"], "sigma_net.0.weight": p["sigma_w0"],
                         "sigma_net.1.weight": p["sigma_w1"], "color_net.0.weight": p["color_w0"],
                         "color_net.1.weight": p["color_w1"], "color_net.2.weight": p["color_w2"]}, strict=False)
    gen = torch.Generator().manual_seed(1)
    start = torch.rand(2, 64 ** 3, generator=gen) * 3.0
    start[torch.rand(2, 64 ** 3, generator=gen) < 0.1] = -1.0
    net.density_grid.copy_(start.to(DEV))
    orig = torch.rand_like
    try:
        torch.rand_like = lambda t: torch.full_like(t, 0.5)
        net.update_extra_state(decay=0.8)
    finally:
        torch.rand_like = orig
    with torch.no_grad():
        sig = lambda xyz: field.density(torch.from_numpy(xyz), p, 2.0, tb)["sigma"].numpy()
        grid, bits, mean = occupancy.update_density_grid(start.numpy(), sig, 64, 2, 2.0, decay=0.8, density_scale=0.7,
                                                         density_thresh=2.0)
    got = net.density_grid.cpu().numpy()
    assert (got[start.numpy() < 0] == -1).all()
    assert np.allclose(got, grid, rtol=2e-4, atol=1e-6)
    assert abs(net.mean_density - mean) < 1e-4 * mean
    thr = min(mean, 2.0)
    near = np.abs(grid.ravel() - thr) < 1e-3 * thr
    diff = np.unpackbits(net.density_bitfield.cpu().numpy(), bitorder="little") != np.unpackbits(bits, bitorder="little")
    assert not (diff & ~near).any() and diff.mean() < 1e-3


@pytest.mark.parametrize("H,density", [(128, 0.05), (64, 0.5), (32, 0.0), (128, 1.0)])
def test_occupancy_cell_sampling_equals_the_oracle(H, density):
    """inr_occ_sample_cells (three launches: ballot masks + block counts, slice histogram, picks) against
    oracle/occupancy.py::sample_cells on the same uniform draws: every pick bit for bit - uniform half and occupied
    half, also with no cell occupied (all picks on cell 0) and with every cell occupied - and the statistics the
    update relies on: the occupied half only hits occupied cells, both halves are spread over the whole range."""
    from instance_nerf_amd import _lib
    from oracle import occupancy
    lib = _lib.load()
    gen = torch.Generator().manual_seed(H)
    n_cells = H ** 3
    grid = torch.rand(n_cells, generator=gen) * 10.0
    grid[torch.rand(n_cells, generator=gen) >= density] = 0.0
    grid[::53] = -1.0                                                        # unseen cells are not occupied either
    n = n_cells // 4
    u = torch.rand(4 * n, generator=gen)
    g_dev, u_dev = grid.to(DEV), u.to(DEV)
    idx = torch.full((2 * n,), -7, dtype=torch.int32, device=DEV)
    work = torch.empty(lib.inr_occ_sample_workspace_bytes(n_cells) // 8 + 1, dtype=torch.int64, device=DEV)
    _lib.check(lib.inr_occ_sample_cells(_lib.ptr(g_dev), n_cells, _lib.ptr(u_dev), n, _lib.ptr(idx), _lib.ptr(work),
                                        _lib.stream_ptr()), "occ_sample_cells")
    got = idx.cpu().numpy()
    want = occupancy.sample_cells(grid.numpy(), u.numpy(), n)
    assert np.array_equal(got, want)
    assert got.min() >= 0 and got.max() < n_cells
    occupied = grid.numpy() > 0
    if occupied.any():
        assert occupied[got[n:]].all()
        assert len(np.unique(got[n:])) > 0.3 * min(n, occupied.sum())
    else:
        assert (got[n:] == 0).all()
    assert len(np.unique(got[:n])) > 0.2 * n_cells                          # n = n_cells / 4 draws: ~22 % distinct cells
    assert (np.diff(got[:n] // (n_cells // 4096)) >= 0).all()               # grouped by slice, never sorted inside


def test_occupancy_update_steady_state_sweep(params_k16, level_table):
    """After the first 16 updates only H^3/4 random cells + H^3/4 random OCCUPIED cells are refreshed per call
    (device-side compaction of the occupied set, no host round trip).  Every cell either keeps its value or becomes
    max(old * decay, sigma(cell centre)); cells marked -1 stay out; most occupied cells are among the refreshed;
    the bitfield is packbits(grid, min(mean, density_thresh)) of the resulting grid, threshold formed on the device."""
    from oracle import occupancy
    net = _network(params_k16, K=0).eval()
    H = net.grid_size
    gen = torch.Generator().manual_seed(3)
    old = torch.rand(1, H ** 3, generator=gen) * 40.0 * (torch.rand(1, H ** 3, generator=gen) < 0.05)
    old[0, ::97] = -1.0                                                      # unseen cells
    net.density_grid.copy_(old.to(DEV))
    net.iter_density = 16
    net.density_thresh = 5.0
    orig = torch.rand_like
    try:
        torch.rand_like = lambda t: torch.full_like(t, 0.5)                  # no jitter: sigma at the cell centres
        net.update_extra_state(decay=0.9)
    finally:
        torch.rand_like = orig
    new = net.density_grid.cpu()
    assert net.iter_density == 17
    cells = torch.arange(H ** 3, dtype=torch.int32, device=DEV)
    from instance_nerf_amd import raymarching
    coords = raymarching.morton3D_invert(cells).float()
    centres = (2 * coords / (H - 1) - 1) * (1.0 - 1.0 / H)
    with torch.no_grad():
        sig = net.density(centres)["sigma"].cpu()
    cand = torch.maximum(old[0] * 0.9, sig)
    changed = new[0] != old[0]
    assert (new[0][old[0] < 0] == -1).all()
    rel = ((new[0] - cand).abs() / cand.abs().clamp(min=1e-6))[changed]
    assert rel.max() < 1e-4, (rel.max(), (rel > 1e-5).sum(), changed.sum())
    frac_changed = changed.float().mean().item()
    assert 0.2 < frac_changed < 0.5                                          # <= H^3/2 distinct cells refreshed
    occupied = old[0] > 0
    assert changed[occupied].float().mean() > 0.9                            # the occupied half really targets occupied cells
    mean = new.clamp(min=0).double().mean().item()
    assert abs(net.mean_density - mean) < 1e-5 * mean
    thr = min(np.float32(net.mean_density), np.float32(5.0))
    assert (net.density_bitfield.cpu().numpy() == occupancy.packbits(new.numpy().ravel(), thr)).all()


# ---------------------------------------------------------------------------- patch-interleaved layout
def test_patch_layout_is_a_permutation_of_ray_major(rm, room, room_bitfield, bits_dev):
    """The fused-frame writer places sample (ray, k) at slot base + sum_i min(c_i,k) + #{i<r: c_i>k}:
    un-permuting with the host-side reference map gives the oracle's ray-major samples bit for bit."""
    from oracle import march, rays
    ro, rd = scene_rays(room, 1000, seed=61)            # 1000: last group is partial (1000 % 16 = 8)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    ref = march.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars)
    xyzs, dirs, deltas, rr = rm.march_rays_patch(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars))
    assert (rr.cpu().numpy() == ref["rays"]).all()
    slots = rm.patch_slots(rr)
    assert slots.shape[0] == ref["total"] and len(set(slots.tolist())) == ref["total"]
    assert (xyzs.cpu().numpy()[slots] == ref["xyzs"]).all()
    assert (deltas.cpu().numpy()[slots] == ref["deltas"]).all()
    assert (dirs.cpu().numpy()[slots] == ref["dirs"]).all()
    # every group's k-th samples are adjacent: slots of (ray r, step 0) within a group are consecutive
    first = slots[torch.as_tensor(ref["rays"][:16, 1].astype(np.int64))[ref["rays"][:16, 2] > 0]]
    assert (first.sort().values == torch.arange(first.min(), first.min() + len(first))).all()


def test_table_feed_equals_plain_feed_bit_for_bit(rm, room, room_bitfield, bits_dev, level_table):
    """march_rays_patch(table=True) -> forward_table(x01, ray ids, per-ray SH table) is the same arithmetic as
    march_rays_patch() -> forward(x, d): normalised coordinates and outputs must be bit-identical."""
    from oracle import field, rays
    ro, rd = scene_rays(room, 1000, seed=63)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    a = (_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars))
    xyzs, dirs, deltas, rr = rm.march_rays_patch(*a)
    x01, ids, deltas_t, rr_t = rm.march_rays_patch(*a, table=True)
    assert (rr == rr_t).all() and (deltas == deltas_t).all()
    assert ((xyzs + 1.0) / 2.0 == x01).all()
    assert (_t(rd)[ids.long()] == dirs).all()
    net = _network(field.init_params(seed=9, table=level_table, table_std=0.3), K=0)
    with torch.no_grad():
        s0, c0 = net(xyzs, dirs)
        s1, c1 = net.forward_table(x01, ids, _t(rd))
    assert (s0 == s1).all() and (c0 == c1).all()
    # empty batch
    e = net.forward_table(x01[:0], ids[:0], _t(rd))
    assert e[0].shape == (0,) and e[1].shape == (0, 3)


def test_composite_patch_equals_ray_major(rm, room, room_bitfield, bits_dev):
    from oracle import rays
    ro, rd = scene_rays(room, 700, seed=62)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    xp, dp, dlp, rr = rm.march_rays_patch(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars))
    xr, dr, dlr, rr2 = rm.march_rays_train(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars))
    M = xp.shape[0]
    gen = torch.Generator().manual_seed(0)
    sig_r = (torch.rand(M, generator=gen) * 80).to(DEV)
    rgb_r = torch.rand(M, 3, generator=gen).to(DEV)
    ext_r = torch.randn(M, 16, generator=gen).to(DEV)
    slots = rm.patch_slots(rr).to(DEV)
    sig_p, rgb_p, ext_p = torch.empty_like(sig_r), torch.empty_like(rgb_r), torch.empty_like(ext_r)
    sig_p[slots], rgb_p[slots], ext_p[slots] = sig_r, rgb_r, ext_r
    a = rm.composite_rays_train(sig_r, rgb_r, dlr[:M], rr2, 1e-4, extra=ext_r)
    b = rm.composite_rays_patch(sig_p, rgb_p, dlp, rr, 1e-4, extra=ext_p)
    for u, v in zip(a, b):
        assert torch.allclose(u, v, atol=1e-5, rtol=1e-5)


def test_get_rays_patch_order_covers_image():
    from instance_nerf_amd.nerf.utils import get_rays, patch_order
    inds = patch_order(40, 24, 4, "cpu")
    assert sorted(inds.tolist()) == list(range(40 * 24))
    first = inds[:16]
    assert set((first // 24).tolist()) == {0, 1, 2, 3} and set((first % 24).tolist()) == {0, 1, 2, 3}


# ---------------------------------------------------------------------------- training parity
@pytest.fixture
def fx_grad(request):
    """The forms of the table-gradient scatter (round 6): False = fp32 atomics (the default), True = int32 sums, 64 = int64
    sums (both opt-in).  Yields 0 / 32 / 64."""
    from instance_nerf_amd.nerf import network
    old = network.FX_GRAD
    network.FX_GRAD = 64 if request.param == 64 else (32 if request.param else 0)
    yield network.FX_GRAD
    network.FX_GRAD = old


@pytest.mark.parametrize("fx_grad", [False, True, 64], indirect=True)
def test_trainer_matches_oracle_training(room, room_bitfield, level_table, fx_grad):
    """NeRF training (MSE on rgb): the HIP Trainer and the CPU oracle, started from the same parameters
    and fed the same ray batches (no jitter), follow the same loss curve - with fp32 atomics
    (default) and with the table gradient summed as int32 (opt-in)."""
    from instance_nerf_amd.nerf.utils import Trainer
    from oracle import field, render
    p0 = field.init_params(seed=3, table=level_table, table_std=1e-4)
    net = _network({k: v.clone() for k, v in p0.items()}, K=0)
    net.density_bitfield.copy_(_t(room_bitfield))
    tr = Trainer("t", None, net, stage="nerf", device=torch.device(DEV), lr=1e-2, iters=100, update_extra_interval=10 ** 9)
    tr.global_step = 1
    p = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    steps, iters = 8, 100
    batches = []
    for s in range(steps):
        ro, rd = scene_rays(room, 256, cam=s % 8, seed=100 + s)
        rgb, _, _ = room.trace(ro, rd)
        batches.append((ro, rd, rgb))
    hip_losses, ref_losses = [], []
    orig_render = net.render
    net.render = lambda *a, **kw: orig_render(*a, **{**kw, "perturb": False, "force_all_rays": True})
    for s, (ro, rd, rgb) in enumerate(batches):
        data = {"rays_o": _t(ro)[None], "rays_d": _t(rd)[None], "images": _t(rgb)[None]}
        hip_losses.append(float(tr.train_one_step(data)))
        for g in opt.param_groups:
            g["lr"] = 1e-2 * 0.1 ** min((s + 2) / iters, 1)          # Trainer: global_step starts at 1, +1 before the step
        out = render.render_train(ro, rd, p, level_table, room_bitfield, min_near=0.05, bg_color=1.0)
        loss = ((out["image"] - torch.from_numpy(rgb)) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    assert ref_losses[-1] < ref_losses[0]
    for a, b in zip(hip_losses, ref_losses):
        assert abs(a - b) < 2e-3 * max(abs(b), 1e-3), (hip_losses, ref_losses)
    psnr = lambda l: -10 * np.log10(l)
    assert abs(psnr(hip_losses[-1]) - psnr(ref_losses[-1])) < 0.05
    w = net.sigma_net[0].weight.detach().cpu()
    # Eight Adam steps of 1e-2 on a table initialised at 1e-4: a row whose gradient is below the fixed-point quantum (6e-8
    # of its level's largest: ~5e-5 of the touched entries, corners with a trilinear weight of ~1e-7) stays put where the
    # oracle's fp32 sum moves it by a full +-lr (Adam, eps 1e-15, is blind to a gradient's size).  The losses agree to
    # 1e-6 relative either way; a first-layer weight whose gradient is near a sign change then lands up to ~half an
    # lr step away (6.2e-3 measured; 1e-3 with fp32 atomics, whose own perturbation is rounding noise): one step's size
    # is the band for the int32 form.
    dw = float((w - p["sigma_w0"].detach()).abs().max())
    print(f"fixed point {fx_grad}: max |sigma_w0 - oracle| after {steps} steps = {dw:.2e}; losses {hip_losses[-1]:.6f} / {ref_losses[-1]:.6f}")
    assert dw < (1.0e-2 if fx_grad == 32 else 2e-3), dw          # int64 sums (quantum 2e-16 of the level's maximum): fp32's band


def test_linear_wgrad_matches_torch():
    from instance_nerf_amd.nerf.network import HipLinear
    gen = torch.Generator().manual_seed(0)
    for M, n_in, n_out in ((1000, 32, 64), (5003, 64, 64), (777, 31, 64), (300, 64, 3), (4, 64, 16)):
        lin = HipLinear(n_in, n_out, bias=False).to(DEV)
        x = torch.randn(M, n_in, generator=gen).to(DEV).requires_grad_(True)
        gy = torch.randn(M, n_out, generator=gen).to(DEV)
        lin(x).backward(gy)
        ref_w = gy.t().double() @ x.detach().double()
        assert torch.allclose(lin.weight.grad.double(), ref_w, atol=1e-3, rtol=1e-4)
        assert torch.allclose(x.grad, gy @ lin.weight.detach(), atol=1e-5)


# ---------------------------------------------------------------------------- other configurations / full size
def test_bound2_two_cascades_bit_exact(rm):
    """bound = 2 -> 2 cascades, mip level chosen per sample from position and step size; dt_gamma > 0."""
    from oracle import hashgrid, march, rays
    rng = np.random.default_rng(5)
    H, C, bound = 64, 2, 2.0
    bits = (rng.random(C * H ** 3 // 8) < 0.03).astype(np.uint8) * rng.integers(1, 256, C * H ** 3 // 8).astype(np.uint8)
    n = 512
    ro = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    aabb = np.asarray([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.2)
    noises = rng.random(n).astype(np.float32)
    for gamma in (0.0, 1.0 / 256):
        ref = march.march_rays_train(ro, rd, bits, bound, C, H, nears, fars, noises, gamma, 512)
        x, d, dl, rr = rm.march_rays_train(_t(ro), _t(rd), bound, _t(bits), C, H, _t(nears), _t(fars), dt_gamma=gamma,
                                           max_steps=512, noises=_t(noises))
        assert ref["total"] > 1000
        assert (rr.cpu().numpy() == ref["rays"]).all()
        assert (x.cpu().numpy()[:ref["total"]] == ref["xyzs"]).all()
        assert (dl.cpu().numpy()[:ref["total"]] == ref["deltas"]).all()
    # encoder at bound 2 (desired resolution 4096: different level table)
    from instance_nerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(desired_resolution=4096, num_levels=12, log2_hashmap_size=17).to(DEV)
    tb = hashgrid.level_table(num_levels=12, log2_hashmap_size=17, desired_resolution=4096)
    emb = (torch.rand(tb["total_rows"], 2, generator=torch.Generator().manual_seed(0)) * 2 - 1)
    enc.embeddings.data.copy_(emb)
    xx = torch.tensor(ref["xyzs"][:4000])
    with torch.no_grad():
        got = enc(xx.to(DEV), bound=bound).cpu()
    want = hashgrid.encode(xx, emb, bound, tb)
    assert (got - want).abs().max() < 1e-5


def test_full_frame_properties(params_k16, room, room_bitfield, level_table):
    """800x800 (640 000 rays): the patch-interleaved path and the ray-major path render the same image;
    offsets are the exclusive scan of the counts; every ray's opacity is in [0, 1]."""
    from instance_nerf_amd.nerf.utils import get_rays
    net = _network(params_k16, K=0).eval()
    net.density_bitfield.copy_(_t(room_bitfield))
    poses, intr, H, W = room.cameras()
    r = get_rays(_t(poses[2:3]), intr, H, W, patch=4)
    with torch.no_grad():
        a = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
        b = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused_raymajor")
    assert int(a["num_samples"][0]) == int(b["num_samples"][0]) > 20_000_000
    assert (a["image"] - b["image"]).abs().max() < 2e-5
    assert (a["weights_sum"] - b["weights_sum"]).abs().max() < 2e-5
    assert a["weights_sum"].min() >= 0 and a["weights_sum"].max() <= 1 + 1e-5
    assert torch.isfinite(a["image"]).all()
    # scan property on the full ray set
    from instance_nerf_amd import raymarching as _rm
    nears, fars = _rm.near_far_from_aabb(r["rays_o"][0], r["rays_d"][0], net.aabb_infer, 0.05)
    _, _, _, rays = _rm.march_rays_patch(r["rays_o"][0], r["rays_d"][0], 1.0, net.density_bitfield,
                                                                                 1, 128, nears, fars)
    cnt = rays[:, 2].long()
    assert torch.equal(rays[:, 1].long(), torch.cumsum(cnt, 0) - cnt)
    assert torch.equal(rays[:, 0].long(), torch.arange(H * W, device=DEV))
    # a random subset of the full frame against the oracle run on exactly those rays (pixels of the full-size job
    # are independent, so this ties the 640 000-ray launch to the oracle at a size it finishes in seconds)
    from oracle import render
    pick = torch.from_numpy(np.random.default_rng(17).choice(H * W, size=1200, replace=False)).to(DEV)
    ro, rd = r["rays_o"][0][pick].cpu().numpy(), r["rays_d"][0][pick].cpu().numpy()
    with torch.no_grad():
        ref = render.render_train(ro, rd, params_k16, level_table, room_bitfield, min_near=0.05)
    assert np.abs(a["image"][0][pick].cpu().numpy() - ref["image"].numpy()).max() < 1e-4
    assert (cnt[pick].cpu().numpy() == ref["rays"][:, 2]).all()


def test_full_size_frame_against_the_c_oracle(rm, params_k16, room, room_bitfield, bits_dev, level_table):
    """BASELINE-size job (800x800 = 640 000 rays) against the scalar C restatement (oracle/c): EVERY ray's sample
    count and offset bit-exact, every sample position of the first 20 000 rays bit-exact, and the WHOLE rendered
    frame - all 640 000 pixels, ~26 M samples, table U(-1,1) so outputs are O(1) - within 1e-4 (the C oracle does it
    in ~10 s on the box's 16 usable cores)."""
    from instance_nerf_amd.nerf.utils import get_rays
    from oracle import c_port
    poses, intr, H, W = room.cameras()
    r = get_rays(_t(poses[2:3]), intr, H, W)
    ro, rd = r["rays_o"][0], r["rays_d"][0]
    nears, fars = rm.near_far_from_aabb(ro, rd, _t(np.asarray([-1, -1, -1, 1, 1, 1], np.float32)), 0.05)
    xyzs, dirs, deltas, rays = rm.march_rays_train(ro, rd, 1.0, bits_dev, 1, 128, nears, fars, force_all_rays=True)
    cn, cf = c_port.near_far_from_aabb(ro.cpu().numpy(), rd.cpu().numpy(), [-1, -1, -1, 1, 1, 1], 0.05)
    assert (cn == nears.cpu().numpy()).all() and (cf == fars.cpu().numpy()).all()
    ref = c_port.march_rays_train(ro.cpu().numpy(), rd.cpu().numpy(), room_bitfield, 1.0, 1, 128, cn, cf)
    assert ref["total"] > 20_000_000
    assert (rays.cpu().numpy() == ref["rays"]).all()
    m = int(ref["rays"][20000, 1])
    assert (xyzs[:m].cpu().numpy() == ref["xyzs"][:m]).all() and (deltas[:m].cpu().numpy() == ref["deltas"][:m]).all()
    del xyzs, dirs, deltas, ref
    net = _network(params_k16, K=0).eval()
    net.density_bitfield.copy_(bits_dev)
    r = get_rays(_t(poses[2:3]), intr, H, W, patch=4)
    with torch.no_grad():
        out = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
    c = c_port.render(r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy(), params_k16, level_table, room_bitfield,
                      min_near=0.05, absolute_depth=True)                      # inference: depth over the absolute t
    assert int(out["num_samples"][0]) == c["total"] > 20_000_000
    assert np.abs(out["image"][0].cpu().numpy() - c["image"]).max() < 1e-4
    assert np.abs(out["weights_sum"][0].cpu().numpy() - c["weights_sum"]).max() < 1e-4
    assert np.abs(out["depth"][0].cpu().numpy() - c["depth"]).max() < 1e-4


def test_full_size_frame_at_bound4_against_the_c_oracle(rm):
    """The BASELINE-size job off the tuned configuration (round-4 verdict item 1b): the synthetic room enlarged 4x in a
    bound-4 volume - THREE occupancy cascades, finest level 8192 (12 of the 16 levels hashed), steps growing with the
    distance (dt_gamma = 1/128, torch-ngp's setting for bound > 1) - 800x800 = 640 000 rays against the scalar C
    restatement: every ray's sample count and offset bit-exact, the sample positions of the first 20 000 rays bit-exact,
    and the whole frame (table U(-1,1), O(1) outputs) within 1e-4.  A second view with constant steps (dt_gamma = 0:
    ~190 samples per ray, rays that cross all three cascades) on 30 000 random pixels."""
    from instance_nerf_amd.nerf.utils import get_rays
    from instance_nerf_amd.scene import RoomScene
    from oracle import c_port, field, hashgrid
    bound, C, H = 4.0, 3, 128
    big = RoomScene(scale=bound)
    bits = big.density_bitfield(H, bound)
    assert bits.shape[0] == C * H ** 3 // 8
    table = hashgrid.level_table(desired_resolution=int(2048 * bound))
    assert int(table["hashed"].sum()) == 12
    p = field.init_params(seed=3, table=table, table_std=1.0, K=0)
    net = _network(p, K=0, bound=int(bound)).eval()
    assert net.cascade == C and (net.encoder.table["offsets"] == table["offsets"]).all()
    net.density_bitfield.copy_(_t(bits))
    net.density_scale = 1.0             # table U(-1,1), steps growing with t: mean opacity ~0.4 (semi-transparent rays)
    poses, intr, Hi, Wi = big.cameras()
    aabb = [-bound] * 3 + [bound] * 3
    r = get_rays(_t(poses[2:3]), intr, Hi, Wi)
    ro, rd = r["rays_o"][0], r["rays_d"][0]
    nears, fars = rm.near_far_from_aabb(ro, rd, _t(np.asarray(aabb, np.float32)), 0.05)
    xyzs, dirs, deltas, rays = rm.march_rays_train(ro, rd, bound, _t(bits), C, H, nears, fars, force_all_rays=True,
                                                   dt_gamma=1 / 128)
    cn, cf = c_port.near_far_from_aabb(ro.cpu().numpy(), rd.cpu().numpy(), aabb, 0.05)
    assert (cn == nears.cpu().numpy()).all() and (cf == fars.cpu().numpy()).all()
    ref = c_port.march_rays_train(ro.cpu().numpy(), rd.cpu().numpy(), bits, bound, C, H, cn, cf, dt_gamma=1 / 128)
    assert ref["total"] > 10_000_000
    assert (rays.cpu().numpy() == ref["rays"]).all()
    m = int(ref["rays"][20000, 1])
    assert (xyzs[:m].cpu().numpy() == ref["xyzs"][:m]).all() and (deltas[:m].cpu().numpy() == ref["deltas"][:m]).all()
    # every cascade is really used: samples outside [-1,1]^3 and outside [-2,2]^3
    far_out = np.abs(ref["xyzs"]).max(1)
    assert (far_out > 2).mean() > 0.05 and ((far_out > 1) & (far_out <= 2)).mean() > 0.05 and (far_out <= 1).mean() > 0.05
    del xyzs, dirs, deltas, ref
    r = get_rays(_t(poses[2:3]), intr, Hi, Wi, patch=4)
    with torch.no_grad():
        out = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused", dt_gamma=1 / 128)
    c = c_port.render(r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy(), p, table, bits, bound=bound, cascade=C,
                      H=H, min_near=0.05, dt_gamma=1 / 128, density_scale=1.0, absolute_depth=True)
    assert int(out["num_samples"][0]) == c["total"] > 10_000_000
    assert 0.15 < float(c["weights_sum"].mean()) < 0.98
    assert np.abs(out["image"][0].cpu().numpy() - c["image"]).max() < 1e-4
    assert np.abs(out["weights_sum"][0].cpu().numpy() - c["weights_sum"]).max() < 1e-4
    assert np.abs(out["depth"][0].cpu().numpy() - c["depth"]).max() < 1e-4
    # constant steps, another view, the early-terminating kernel as well
    pick = torch.from_numpy(np.sort(np.random.default_rng(23).choice(Hi * Wi, size=30000, replace=False))).to(DEV)
    r = get_rays(_t(poses[5:6]), intr, Hi, Wi, inds=pick)
    c = c_port.render(r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy(), p, table, bits, bound=bound, cascade=C,
                      H=H, min_near=0.05, dt_gamma=0.0, density_scale=1.0, absolute_depth=True)
    for mode in ("fused", "fused_terminate"):
        with torch.no_grad():
            out = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode=mode, dt_gamma=0)
        assert int(out["num_samples"][0]) == c["total"] > 3_000_000, mode
        assert np.abs(out["image"][0].cpu().numpy() - c["image"]).max() < 1e-4, mode
        assert np.abs(out["depth"][0].cpu().numpy() - c["depth"]).max() < 1e-4, mode


def test_xcd_sliced_frame_path_is_bit_identical():
    """Round 5: the sliced frame path (``NeRFNetwork.frame_slices``: the three finest levels evaluated level by level by a
    pre-pass, the fused kernel on the other thirteen) returns the fused kernel's numbers bit for bit - table U(-1,1), a
    bound-4 frame of 800x800 rays and ragged small batches - and through ``render`` the same image; "auto" never probes
    below a finest level of 8192."""
    from instance_nerf_amd import raymarching as rmod
    from instance_nerf_amd.nerf.utils import get_rays
    from instance_nerf_amd.scene import RoomScene
    from oracle import field, hashgrid
    bound = 4
    big = RoomScene(scale=float(bound))
    table = hashgrid.level_table(desired_resolution=2048 * bound)
    p = field.init_params(seed=11, table=table, table_std=1.0, K=0)
    net = _network(p, K=0, bound=bound).eval()
    net.density_bitfield.copy_(_t(big.density_bitfield(128, float(bound))))
    poses, intr, H, W = big.cameras()
    r = get_rays(_t(poses[1:2]), intr, H, W, patch=4)
    ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
    nears, fars = rmod.near_far_from_aabb(ro, rd, net.aabb_infer, net.min_near)
    xyzs, ids, deltas, rays = rmod.march_rays_patch(ro, rd, bound, net.density_bitfield, net.cascade, 128, nears, fars,
                                                    1 / 128, 1024, table=True)
    assert xyzs.shape[0] > 10_000_000
    with torch.no_grad():
        shq = net.sh_table(rd)
        for M in (xyzs.shape[0], 1, 17, 1000, 70001):
            net.frame_slices = False
            ref = net.forward_table(xyzs[:M].contiguous(), ids[:M].contiguous(), rd, shq=shq)
            net.frame_slices = True
            got = net.forward_table(xyzs[:M].contiguous(), ids[:M].contiguous(), rd, shq=shq)
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), M
        net.frame_slices = False
        a = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused", dt_gamma=1 / 128)
        net.frame_slices = True
        b = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused", dt_gamma=1 / 128)
        assert torch.equal(a["image"], b["image"]) and torch.equal(a["depth"], b["depth"])
        # auto: decides by measurement at bound 4 (either answer is legal), never leaves the fused kernel at bound 1
        net.frame_slices = "auto"
        net._slice_probe = None
        for _ in range(8):
            c = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused", dt_gamma=1 / 128)
            torch.cuda.synchronize()
        assert torch.equal(a["image"], c["image"]) and net._slice_probe["choice"] in (True, False)
        # round 6: three timings per path (AB BA AB), decided on the median; the render says which path it took; new
        # parameters reopen the decision
        assert all(len(v) >= 3 for v in net._slice_probe["ms"].values())
        assert c["frame_path"] == ("sliced" if net._slice_probe["choice"] else "fused")
        net.load_state_dict(net.state_dict())
        assert net._slice_probe is None
    small = _network(field.init_params(seed=1, table=hashgrid.level_table(), table_std=1.0, K=0), K=0).eval()
    assert small.frame_slices == "auto" and small._use_slices(1 << 24) is False and small._slice_probe is None


def test_generic_sampler_without_cuda_ray(params_k16, room, level_table):
    """NeRFNetwork(cuda_ray=False).render(): upstream's default sampler (128 uniform + 128 importance samples, no
    occupancy grid) as tensor-op glue around the HIP ray/box test and field kernels, against the oracle's ray-by-ray
    restatement; staged chunks equal the single batch; training mode back-propagates into table and weights."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from oracle import render
    net = NeRFNetwork(cuda_ray=False, num_instances=0, min_near=0.05).to(DEV).eval()
    net.load_state_dict({"encoder.embeddings": params_k16["embeddings"], "sigma_net.0.weight": params_k16["sigma_w0"],
                         "sigma_net.1.weight": params_k16["sigma_w1"], "color_net.0.weight": params_k16["color_w0"],
                         "color_net.1.weight": params_k16["color_w1"], "color_net.2.weight": params_k16["color_w2"]},
                        strict=False)
    net.density_scale = 0.3                                        # table U(-1,1): keep the rays semi-transparent
    ro, rd = scene_rays(room, 40, seed=5)
    ro[0], rd[0] = [5, 5, 5], [1, 0, 0]                            # misses the box: background only
    with torch.no_grad():
        a = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, num_steps=64, upsample_steps=64)
        b = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, num_steps=64, upsample_steps=64, staged=True, max_ray_batch=16)
    ref = render.render_generic(ro, rd, params_k16, level_table, min_near=0.05, num_steps=64, upsample_steps=64,
                                density_scale=0.3)
    assert a["image"].shape == (1, 40, 3) and (a["image"][0, 0] == 1).all() and a["weights_sum"][0, 0] == 0
    assert 0.05 < ref["weights_sum"][1:].mean() < 0.999
    assert np.abs(a["image"][0].cpu().numpy() - ref["image"]).max() < 2e-3
    assert np.abs(a["weights_sum"][0].cpu().numpy() - ref["weights_sum"]).max() < 2e-3
    assert np.abs(a["depth"][0].cpu().numpy() - ref["depth"]).max() < 2e-3
    assert (a["image"] - b["image"]).abs().max() < 1e-6
    net.train()
    out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, num_steps=32, upsample_steps=32, perturb=True)
    out["image"].sum().backward()
    assert net.encoder.embeddings.grad.abs().sum() > 0 and net.color_net[2].weight.grad.abs().sum() > 0


@pytest.mark.parametrize("seed", _seeds(6))
def test_composite_fuzz_against_the_c_oracle(rm, seed):
    """Random ray sets - empty rays, single-sample rays, rays that terminate early (large sigma), K = 0 / 16 / 64 extra
    channels, sample buffers smaller than the total (dropped rays) - training compositing forward against the scalar
    C restatement, and the patch-interleaved inference compositing against the same numbers."""
    from oracle import c_port
    rng = np.random.default_rng(700 + seed)
    N = int(rng.choice([1, 16, 33, 500]))
    cnt = rng.integers(0, 90, size=N).astype(np.int32)
    cnt[rng.random(N) < 0.15] = 0
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.int32)
    total = int(cnt.sum())
    rays = np.stack([rng.permutation(N).astype(np.int32), off, cnt], -1)
    M = total if seed % 2 == 0 else max(total - int(cnt[-1]) - 1, 0)          # odd seeds: the last ray does not fit
    sig = (rng.random(max(total, 1)) ** 3 * float(rng.choice([1.0, 40.0, 2000.0]))).astype(np.float32)
    rgb = rng.random((max(total, 1), 3)).astype(np.float32)
    dl = np.stack([np.full(max(total, 1), 3.4e-3, np.float32), rng.random(max(total, 1)).astype(np.float32) * 0.05], -1)
    K = int(rng.choice([0, 16, 64]))
    ex = rng.normal(size=(max(total, 1), K)).astype(np.float32) if K else None
    ref = c_port.composite_rays_train(sig[:M], rgb[:M], dl[:M], rays, 1e-4, extra=None if ex is None else ex[:M])
    out = rm.composite_rays_train(_t(sig[:M]), _t(rgb[:M]), _t(dl[:M]), _t(rays), 1e-4, extra=None if ex is None else _t(ex[:M]))
    assert np.abs(out[0].cpu().numpy() - ref["weights_sum"]).max() < 2e-6
    assert np.abs(out[1].cpu().numpy() - ref["depth"]).max() < 1e-5
    assert np.abs(out[2].cpu().numpy() - ref["image"]).max() < 2e-6
    if K:
        assert np.abs(out[3].cpu().numpy() - ref["extra"]).max() < 2e-5
    if seed % 2 == 1 and cnt[-1] > 0:
        assert float(out[0][int(rays[-1, 0])]) == 0.0                          # the dropped ray composites to nothing


def _fuzz_volume(rng, varied):
    """The volume a fuzz case renders in.  varied=False: the configuration every kernel was tuned on (bound 1, one
    128^3 cascade, the default level table).  varied=True (round-4 verdict item 1): bound 1 / 2 / 4 (1-3 occupancy
    cascades of 64^3 or 128^3 cells), hash tables of 2^15 .. 2^19 rows per level, finest resolutions from 512 to
    2048 * bound (so between 2 and 13 of the 16 levels are hashed) and, one case in four, 12 levels (the composable
    path: HIP encoder + BLAS layers).  -> (bound, cascades, grid H, oracle level table, encoder_kwargs)."""
    from oracle import hashgrid
    if not varied:
        return 1.0, 1, 128, hashgrid.level_table(), {}
    bound = float(rng.choice([1.0, 2.0, 2.0, 4.0, 4.0]))
    C = 1 + int(np.ceil(np.log2(bound)))
    H = int(rng.choice([64, 128]))
    kw = {"num_levels": int(rng.choice([16, 16, 16, 12])), "log2_hashmap_size": int(rng.choice([15, 17, 19])),
          "desired_resolution": int(rng.choice([512, 2048, 2048 * bound, 2048 * bound]))}
    return bound, C, H, hashgrid.level_table(**kw), kw


def _render_fuzz_case(seed, varied):
    from oracle import c_port, field
    rng = np.random.default_rng((14000 if varied else 4000) + seed)
    bound, C, H, table, enc_kw = _fuzz_volume(rng, varied)
    K = int(rng.choice([0, 5, 16, 31, 48, 64]))              # 31 = the reference's 30 detections + background
    p = field.init_params(seed=seed, table=table, table_std=1.0, K=K)
    fill = float(rng.choice([0.0, 0.002, 0.05, 0.5, 1.0]))
    nb = C * H ** 3 // 8
    bits = (rng.random(nb) < fill).astype(np.uint8) * rng.integers(1, 256, nb).astype(np.uint8)
    if fill == 1.0:
        bits[:] = 255
    n = int(rng.choice([1, 7, 16, 100, 700]))
    ro = (rng.uniform(-0.9, 0.9, size=(n, 3)) * bound).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    if n > 2:
        ro[0], rd[0] = [4 * bound, 4 * bound, 4 * bound], [1, 0, 0]          # misses the box
    scale = float(rng.choice([1.0, 30.0, 1000.0])) if fill < 0.5 else float(rng.choice([0.05, 1.0]))
    gamma = float(rng.choice([0.0, 1.0 / 128]))
    steps = int(rng.choice([64, 1024])) if fill < 0.5 else 64
    net = _network(p, K=K, bound=int(bound), grid_size=H, encoder_kwargs=enc_kw).eval()
    assert net.cascade == C and (net.encoder.table["offsets"] == table["offsets"]).all()
    net.density_bitfield.copy_(_t(bits))
    net.density_scale = scale
    vol = dict(bound=bound, cascade=C, H=H)
    ref = c_port.render(ro, rd, p, table, bits, min_near=0.05, dt_gamma=gamma, max_steps=steps,
                        with_instance=K > 0, density_scale=scale, absolute_depth=True, **vol)       # inference semantics
    ref_train = c_port.render(ro, rd, p, table, bits, min_near=0.05, dt_gamma=gamma, max_steps=steps,
                              with_instance=K > 0, density_scale=scale, **vol)
    modes = ["fused", "fused_terminate", "fused_raymajor", "auto"] + (["wavefront"] if n <= 100 and steps == 64 else [])
    for mode in modes:
        with torch.no_grad():
            out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, dt_gamma=gamma, max_steps=steps, infer_mode=mode)
        tag = (mode, K, fill, n, scale, gamma, steps, bound, H, tuple(enc_kw.items()))
        if "num_samples" in out:
            assert int(out["num_samples"][0]) == ref["total"], tag
        # upstream's alive-ray loop advances `step` by n_step = clamp(N // n_alive, 1, 8) per iteration, so a ray that
        # is still alive when the budget runs out gets up to 7 samples MORE than max_steps, how many depends on the
        # other rays; the one-pass modes stop at exactly max_steps like the training marcher.
        # Rays that hit the cap are therefore only compared in the one-pass modes.
        ok = np.ones(n, bool) if mode != "wavefront" else ref["counts"] < steps
        assert np.abs(out["image"][0].cpu().numpy() - ref["image"])[ok].max(initial=0) < 2e-4, tag
        assert np.abs(out["weights_sum"][0].cpu().numpy() - ref["weights_sum"])[ok].max(initial=0) < 2e-4, tag
        hit = np.isfinite(ref["depth"]) & ok                # rays that miss the box: 0 / 0 on both sides
        assert np.abs(out["depth"][0].cpu().numpy() - ref["depth"])[hit].max(initial=0) < 2e-4, tag
        if K:
            lim = 2e-3 * max(1.0, float(np.abs(ref["instance"]).max()))
            assert np.abs(out["instance"][0].cpu().numpy() - ref["instance"])[ok].max(initial=0) < lim, tag
    net.train()
    with torch.no_grad():
        out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, dt_gamma=gamma, max_steps=steps, perturb=False,
                         force_all_rays=True)
    assert np.abs(out["image"][0].cpu().numpy() - ref["image"]).max() < 2e-4
    hit = np.isfinite(ref_train["depth"])
    assert np.abs(out["depth"][0].cpu().numpy() - ref_train["depth"])[hit].max(initial=0) < 2e-4   # training: t from the first step
    if K:
        assert np.abs(out["instance"][0].cpu().numpy() - ref["instance"]).max() < 2e-3 * max(1.0, float(np.abs(ref["instance"]).max()))


@pytest.mark.parametrize("seed", _seeds(10))
def test_render_fuzz_against_the_c_oracle(seed):
    """Whole renders on random set-ups - occupancy from empty to full, 1..700 rays (ragged 16-ray groups), K = 0 / 16 /
    48 / 64 instance logits, opaque and transparent densities, constant and growing steps, every inference mode and
    the training-mode path - against the scalar C restatement: sample totals exact, image / opacity / depth / rendered
    logits within 1e-4 / 1e-3."""
    _render_fuzz_case(seed, varied=False)


@pytest.mark.parametrize("seed", _seeds(12))
def test_render_fuzz_over_bounds_and_level_tables_against_the_c_oracle(seed):
    """The same whole-render fuzz off the tuned configuration (``_fuzz_volume(varied=True)``): bound 1 / 2 / 4 with their
    occupancy cascades, 64^3 and 128^3 grids, table sizes 2^15 .. 2^19, finest resolutions 512 .. 8192, 12 and 16
    levels."""
    _render_fuzz_case(seed, varied=True)


def _gradient_fuzz_case(seed, varied, collect=None):
    """collect: a dict that receives {(stage, tensor): (norm-wise relative error, samples)} INSTEAD of the tolerance
    assertion (the exact-fp32 A/B test reads the errors of both builds)."""
    from oracle import field, render
    rng = np.random.default_rng((19000 if varied else 9000) + seed)
    bound, C, H, level_table, enc_kw = _fuzz_volume(rng, varied)
    # (stage, K) by seed: the two are independent of each other (round 2 took K from seed % 5 and the stage from
    # seed % 2, so K = 64 only ever met the NeRF stage); 31 = the reference's 30 detections + background
    stage = "nerf" if seed % 2 == 0 else "instance"
    K = [64, 31, 16, 5][(seed // 2) % 4]
    p = field.init_params(seed=seed, table=level_table, table_std=1.0, K=K)
    fill = float(rng.choice([0.02, 0.2]))
    nb = C * H ** 3 // 8
    bits = (rng.random(nb) < fill).astype(np.uint8) * rng.integers(1, 256, nb).astype(np.uint8)
    n = int(rng.choice([33, 90, 150]))
    ro = (rng.uniform(-0.8, 0.8, size=(n, 3)) * bound).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    scale = float(rng.choice([0.3, 3.0]))
    gamma = float(rng.choice([0.0, 1.0 / 128]))
    net = _network({k: v.clone() for k, v in p.items()}, K=K, bound=int(bound), grid_size=H, encoder_kwargs=enc_kw).train()
    assert net.cascade == C
    net.density_bitfield.copy_(_t(bits))
    net.density_scale = scale
    trained = ("embeddings", "sigma_w0", "sigma_w1", "color_w0", "color_w1", "color_w2") if stage == "nerf" else \
        ("inst_embeddings", "inst_w0", "inst_w1", "inst_w2")
    names = {"embeddings": "encoder.embeddings", "sigma_w0": "sigma_net.0.weight", "sigma_w1": "sigma_net.1.weight",
             "color_w0": "color_net.0.weight", "color_w1": "color_net.1.weight", "color_w2": "color_net.2.weight",
             "inst_embeddings": "instance_encoder.embeddings", "inst_w0": "instance_net.0.weight",
             "inst_w1": "instance_net.1.weight", "inst_w2": "instance_net.2.weight"}
    if stage == "instance":
        net.freeze_nerf()
    else:
        for q in list(net.instance_encoder.parameters()) + list(net.instance_net.parameters()):
            q.requires_grad_(False)
    q = {k: v.clone().requires_grad_(k in trained) for k, v in p.items()}
    ref = render.render_train(ro, rd, q, level_table, bits, bound=bound, cascade=C, H=H, min_near=0.05, dt_gamma=gamma,
                              max_steps=256, with_instance=stage == "instance", density_scale=scale)
    out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=True, dt_gamma=gamma,
                     max_steps=256)
    assert int(out["num_samples"][0]) == ref["total"] > 0
    if stage == "nerf":
        target = rng.random((n, 3)).astype(np.float32)
        loss = ((out["image"][0] - _t(target)) ** 2).mean()
        rl = ((ref["image"] - torch.from_numpy(target)) ** 2).mean()
    else:
        labels = rng.integers(-1, K, size=n)
        loss = torch.nn.functional.cross_entropy(out["instance"][0], _t(labels).long(), ignore_index=-1)
        rl = render.instance_ce_loss(ref["instance"], labels)
    loss.backward()
    rl.backward()
    assert abs(float(loss) - float(rl)) < 1e-4 * max(1.0, abs(float(rl))), (stage, float(loss), float(rl))
    params = dict(net.named_parameters())
    for k in trained:
        got, want = params[names[k]].grad.cpu(), q[k].grad
        assert want.abs().sum() > 0, k
        # norm-wise, at north_star's 1e-3 (round 5; 2e-2 / 1e-1 until then).  Measured on these sixteen set-ups
        # (test_training_gradients_under_the_exact_fp32_build): 2.7e-5 worst for the default split-bf16 build, 4e-6 for
        # the exact-fp32 one.  What the old tolerance allowed for is a RARE event, not a level: a ReLU pre-activation
        # within rounding of zero falling on the other side in the split-bf16 forward - one flipped unit of one sample is
        # ~1 % of the whole gradient when a set-up has only ~100 samples (seed 31 of an INR_FUZZ_SEEDS sweep in round 3:
        # 128 samples, 85 of 13746 touched rows differ, composable path 3e-6).  A sweep that hits one shows it as a
        # single failing seed that passes under INR_LIB_PATH=libinr_hip_fp32.so.
        tol = 1e-3
        if collect is not None:
            collect[(stage, k)] = (float(torch.linalg.norm(got - want) / torch.linalg.norm(want)), int(ref["total"]))
            continue
        assert torch.linalg.norm(got - want) < tol * torch.linalg.norm(want), (stage, k, ref["total"])
    for k in set(names) - set(trained):
        if names[k] in params:
            assert params[names[k]].grad is None, k


@pytest.mark.parametrize("seed", _seeds(8))
def test_training_gradients_fuzz_against_the_oracle(seed):
    """Both training stages on random set-ups (occupancy, density scale, ray count, K, growing / constant steps, a
    sample buffer that drops the last rays): loss and ALL gradients - table, sigma / colour nets or instance nets -
    of the fused training kernels against torch autograd through the numpy/torch oracle."""
    _gradient_fuzz_case(seed, varied=False)


@pytest.mark.parametrize("seed", _seeds(8))
def test_training_gradients_fuzz_over_bounds_and_level_tables(seed):
    """The gradient fuzz off the tuned configuration (``_fuzz_volume(varied=True)``: bound 1 / 2 / 4, cascades, table
    sizes, finest resolutions, 12 or 16 levels)."""
    _gradient_fuzz_case(seed, varied=True)


def test_training_batch_that_misses_the_volume(params_k16, room_bitfield):
    """Every ray of a training batch misses the box (no samples at all): the render is the background, the loss is
    finite, backward and the optimiser step run (zero table gradient) - found by the compositing fuzz test: the C ABI
    refused the empty sample arrays."""
    from instance_nerf_amd.nerf.utils import Trainer
    net = _network({k: v.clone() for k, v in params_k16.items()}, K=16)
    net.density_bitfield.copy_(_t(room_bitfield))
    ro = torch.full((1, 64, 3), 5.0, device=DEV)
    rd = torch.tensor([1.0, 0.0, 0.0], device=DEV).expand(1, 64, 3).contiguous()
    for stage, extra in (("nerf", {"images": torch.rand(1, 64, 3, device=DEV)}),
                         ("instance", {"masks": torch.randint(0, 16, (1, 64), device=DEV)})):
        tr = Trainer("m", None, net, stage=stage, device=torch.device(DEV), update_extra_interval=10 ** 9)
        tr.global_step = 1
        before = net.encoder.embeddings.detach().clone()
        loss = tr.train_one_step({"rays_o": ro, "rays_d": rd, **extra})
        assert torch.isfinite(loss)
        if stage == "nerf":
            assert torch.equal(net.encoder.embeddings.detach(), before)        # nothing was sampled: nothing moves
    net.eval()
    with torch.no_grad():
        out = net.render(ro, rd, bg_color=1)
    assert (out["image"] == 1).all() and (out["weights_sum"] == 0).all() and (out["instance"] == 0).all()


def test_rays_missing_the_volume(rm, bits_dev):
    ro = np.asarray([[5, 5, 5], [0, 0, 3], [0, 0, 0]], np.float32)
    rd = np.asarray([[1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    aabb = _t(np.asarray([-1, -1, -1, 1, 1, 1], np.float32))
    nears, fars = rm.near_far_from_aabb(_t(ro), _t(rd), aabb, 0.05)
    x, d, dl, rays = rm.march_rays_patch(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, nears, fars)
    assert rays[:2, 2].tolist() == [0, 0] and rays[2, 2] > 0
    ws, depth, img = rm.composite_rays_patch(torch.zeros(x.shape[0], device=DEV), torch.zeros(x.shape[0], 3, device=DEV), dl, rays)
    assert ws.tolist() == [0.0, 0.0, 0.0]


def test_ray_box_test_with_ignored_labels(rm, room, bits_dev):
    """inr_near_far_from_aabb_skip: rays whose label is the ignored one come back exactly as rays that miss the volume do
    (near = far = FLT_MAX: the marchers give them no samples), every other ray as from the plain test - bit for bit; any
    ignore value, not only -1."""
    poses, intr, H, W = room.cameras(H=64, W=64, focal=32.0)
    r = __import__("instance_nerf_amd.nerf.utils", fromlist=["get_rays"]).get_rays(_t(poses[:1]), intr, H, W)
    ro, rd = r["rays_o"].view(-1, 3), r["rays_d"].view(-1, 3)
    aabb = _t(np.asarray([-1, -1, -1, 1, 1, 1], np.float32))
    labels = torch.randint(-1, 5, (ro.shape[0],), device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    n0, f0 = rm.near_far_from_aabb(ro, rd, aabb, 0.05)
    for ignore in (-1, 3):
        n1, f1 = rm.near_far_from_aabb(ro, rd, aabb, 0.05, skip_labels=labels, ignore_index=ignore)
        keep = labels != ignore
        assert 0 < int(keep.sum()) < keep.numel()
        assert torch.equal(n1[keep], n0[keep]) and torch.equal(f1[keep], f0[keep])
        big = torch.finfo(torch.float32).max
        assert bool((n1[~keep] == big).all()) and bool((f1[~keep] == big).all())
        x, d, dl, rays = rm.march_rays_patch(ro, rd, 1.0, bits_dev, 1, 128, n1, f1)
        assert int(rays[~keep, 2].sum()) == 0 and int(rays[keep, 2].sum()) > 0
    n2, f2 = rm.near_far_from_aabb(ro, rd, aabb, 0.05, skip_labels=labels.to(torch.int32), ignore_index=3)     # any integer dtype
    assert torch.equal(n2, n1) and torch.equal(f2, f1)
    with pytest.raises(RuntimeError):
        rm.near_far_from_aabb(ro, rd, aabb, 0.05, skip_labels=labels[:-1])


@pytest.mark.parametrize("cap", [0, 40, 256, 1024])
def test_capture_replay_is_bit_identical(rm, room, room_bitfield, bits_dev, cap, monkeypatch, request):
    """The write pass replaying the recorded candidate bit mask, re-marching rays that outrun the mask, or
    marching twice all produce the oracle's samples bit for bit (both writers, thread-per-ray marcher)."""
    rm.set_march_mode("lane_per_ray")
    request.addfinalizer(lambda: rm.set_march_mode(None))
    from oracle import march, rays
    monkeypatch.setattr(rm, "SAMPLE_CAP", cap)
    monkeypatch.setattr(rm, "SAMPLE_CAP_TRAIN", cap)
    ro, rd = scene_rays(room, 777, seed=81)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    noises = np.random.default_rng(3).random(777).astype(np.float32)
    ref = march.march_rays_train(ro, rd, room_bitfield, 1.0, 1, 128, nears, fars, noises, 1.0 / 256, 1024)
    x, d, dl, rr = rm.march_rays_train(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars), dt_gamma=1.0 / 256,
                                       noises=_t(noises))
    assert (rr.cpu().numpy() == ref["rays"]).all()
    assert (x.cpu().numpy()[:ref["total"]] == ref["xyzs"]).all() and (dl.cpu().numpy()[:ref["total"]] == ref["deltas"]).all()
    xp, dp, dlp, rp = rm.march_rays_patch(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars), dt_gamma=1.0 / 256,
                                          noises=_t(noises))
    slots = rm.patch_slots(rp)
    assert (xp.cpu().numpy()[slots] == ref["xyzs"]).all() and (dlp.cpu().numpy()[slots] == ref["deltas"]).all()
    assert (dp.cpu().numpy()[slots] == ref["dirs"]).all()


def test_get_rays_bit_exact(room):
    from instance_nerf_amd.nerf.utils import get_rays
    from oracle import rays
    poses, intr, H, W = room.cameras()
    inds = np.random.default_rng(0).integers(0, H * W, size=5000)
    ref = rays.get_rays(poses[:3], intr, H, W, inds=inds)
    got = get_rays(_t(poses[:3]), intr, H, W, inds=_t(inds))
    assert (got["rays_d"].cpu().numpy() == ref["rays_d"]).all()
    assert (got["rays_o"].cpu().numpy() == ref["rays_o"]).all()
    full = get_rays(_t(poses[:1]), intr, 16, 12)
    assert full["rays_d"].shape == (1, 192, 3) and torch.allclose(full["rays_d"].norm(dim=-1), torch.ones(1, 192, device=DEV), atol=1e-6)
    host = get_rays(torch.from_numpy(poses[:3]), intr, H, W, inds=torch.from_numpy(inds))      # CPU branch
    assert np.allclose(host["rays_d"].numpy(), ref["rays_d"], atol=1e-7)


def test_instance_training_matches_oracle(room, room_bitfield, level_table):
    """Instance-field stage (NeRF frozen, CE on rendered logits, ignore -1): HIP Trainer vs oracle from the same
    parameters and batches - same loss curve, same predicted instance ids (mIoU of the two label maps ~ 1)."""
    from instance_nerf_amd.nerf.utils import MIoUMeter, Trainer
    from oracle import field, render
    K = 16
    p0 = field.init_params(seed=4, table=level_table, table_std=1e-4, K=K)
    p0["embeddings"] = (torch.rand(p0["embeddings"].shape, generator=torch.Generator().manual_seed(6)) * 2 - 1)
    net = _network({k: v.clone() for k, v in p0.items()}, K=K)
    net.density_bitfield.copy_(_t(room_bitfield))
    tr = Trainer("i", None, net, stage="instance", device=torch.device(DEV), lr=1e-2, iters=100, update_extra_interval=10 ** 9)
    tr.global_step = 1
    orig_render = net.render
    net.render = lambda *a, **kw: orig_render(*a, **{**kw, "perturb": False, "force_all_rays": True})
    p = {k: v.clone() for k, v in p0.items()}
    train_keys = ("inst_embeddings", "inst_w0", "inst_w1", "inst_w2")
    for k in train_keys:
        p[k].requires_grad_(True)
    opt = torch.optim.Adam([p[k] for k in train_keys], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ro, rd = scene_rays(room, 192, seed=130)
    _, ids, _ = room.trace(ro, rd)
    labels = np.where(np.arange(192) % 9 == 0, -1, ids % K)
    hip, ref = [], []
    for s in range(6):
        data = {"rays_o": _t(ro)[None], "rays_d": _t(rd)[None], "masks": _t(labels)[None]}
        hip.append(float(tr.train_one_step(data)))
        for g in opt.param_groups:
            g["lr"] = 1e-2 * 0.1 ** min((s + 2) / 100, 1)
        out = render.render_train(ro, rd, p, level_table, room_bitfield, min_near=0.05, with_instance=True)
        loss = render.instance_ce_loss(out["instance"], labels)
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref.append(float(loss))
    assert ref[-1] < ref[0]
    for a, b in zip(hip, ref):
        assert abs(a - b) < 2e-3 * max(abs(b), 1e-3), (hip, ref)
    net.eval()
    with torch.no_grad():
        pred_hip = orig_render(_t(ro)[None], _t(rd)[None], bg_color=1)["instance"][0].argmax(-1).cpu()
        pred_ref = render.render_train(ro, rd, p, level_table, room_bitfield, min_near=0.05, with_instance=True)["instance"].argmax(-1)
    meter = MIoUMeter(K)
    meter.update(pred_hip, pred_ref)
    assert meter.measure() > 0.98


def test_config2_instance_step_at_full_size_against_the_oracle(room, room_bitfield, level_table):
    """BASELINE configs[2] at its stated size, end to end: 4096 rays of the room, K = 64 logits, NeRF frozen, mask
    labels with 10 % ignored (-1) - the loss and all four instance-gradient tensors (hash table, three layers) of the
    HIP training path against torch autograd through the oracle, norm-wise within 1e-3 (~240 k samples; the oracle
    takes ~8 s on the CPU)."""
    from oracle import field, render
    K, N = 64, 4096
    p = field.init_params(seed=3, table=level_table, table_std=1.0, K=K)
    ro, rd = scene_rays(room, N, cam=1, seed=77)
    _, ids, _ = room.trace(ro, rd)
    labels = np.where(np.random.default_rng(3).random(N) < 0.1, -1, ids % K)
    net = _network({k: v.clone() for k, v in p.items()}, K=K).train()
    net.density_bitfield.copy_(_t(room_bitfield))
    net.freeze_nerf()
    trained = {"inst_embeddings": "instance_encoder.embeddings", "inst_w0": "instance_net.0.weight",
               "inst_w1": "instance_net.1.weight", "inst_w2": "instance_net.2.weight"}
    q = {k: v.clone().requires_grad_(k in trained) for k, v in p.items()}
    ref = render.render_train(ro, rd, q, level_table, room_bitfield, min_near=0.05, with_instance=True)
    rl = render.instance_ce_loss(ref["instance"], labels)
    rl.backward()
    from instance_nerf_amd import raymarching
    out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=True)
    assert int(out["num_samples"][0]) == ref["total"] > 150_000
    loss = raymarching.cross_entropy(out["instance"][0], _t(labels).long(), ignore_index=-1)      # the Trainer's loss
    loss.backward()
    assert abs(float(loss) - float(rl)) < 1e-4 * abs(float(rl)), (float(loss), float(rl))
    lim = 2e-3 * max(1.0, float(ref["instance"].detach().abs().max()))
    assert (out["instance"][0].detach().cpu() - ref["instance"].detach()).abs().max() < lim
    params = dict(net.named_parameters())
    for k, name in trained.items():
        got, want = params[name].grad.cpu(), q[k].grad
        assert want.abs().sum() > 0, k
        rel = float(torch.linalg.norm(got - want) / torch.linalg.norm(want))
        assert rel < 1e-3, (k, rel)
    for name, prm in params.items():
        if name not in trained.values():
            assert prm.grad is None, name


@pytest.mark.parametrize("fx_grad", [False, True, 64], indirect=True)
def test_trainer_checkpoint_roundtrip(tmp_path, room, room_bitfield, level_table, fx_grad):
    """save_checkpoint / load_checkpoint restore model, occupancy state and optimiser moments: two trainers
    continue identically (upstream keys: epoch, global_step, stats, model, optimizer, mean_count, mean_density)."""
    from instance_nerf_amd.nerf.utils import Trainer
    from oracle import field
    p0 = field.init_params(seed=8, table=level_table, table_std=1e-4)

    def make():
        net = _network({k: v.clone() for k, v in p0.items()}, K=0)
        net.density_bitfield.copy_(_t(room_bitfield))
        tr = Trainer("ck", None, net, stage="nerf", device=torch.device(DEV), iters=50, update_extra_interval=10 ** 9,
                     workspace=str(tmp_path), ema_decay=0.95)
        tr.global_step = 1
        orig = net.render
        net.render = lambda *a, **kw: orig(*a, **{**kw, "perturb": False, "force_all_rays": True})
        return tr

    def batch(s):
        ro, rd = scene_rays(room, 128, cam=s % 8, seed=300 + s)
        rgb, _, _ = room.trace(ro, rd)
        return {"rays_o": _t(ro)[None], "rays_d": _t(rd)[None], "images": _t(rgb)[None]}

    a = make()
    for s in range(3):
        a.train_one_step(batch(s))
    path = a.save_checkpoint(full=True)          # upstream: full = with optimizer / scheduler / EMA state
    state = torch.load(path, map_location="cpu", weights_only=False)
    assert {"epoch", "global_step", "stats", "model", "optimizer", "mean_count", "mean_density"} <= set(state)
    b = make()
    b.load_checkpoint(path)
    assert b.global_step == a.global_step
    la = float(a.train_one_step(batch(3)))
    lb = float(b.train_one_step(batch(3)))
    assert abs(la - lb) < 1e-6 * max(1.0, abs(la))
    assert torch.allclose(a.model.encoder.embeddings, b.model.encoder.embeddings, atol=1e-7)
    if fx_grad:
        # int32 table-gradient sums (opt-in): the checkpoint carries the levels' scales, so the resumed trainer rounds its
        # row sums to the same quanta and continues with the SAME BITS in every parameter
        assert "fx_state" in state and list(state["fx_state"]) == ["encoder.embeddings"] and state["fx_bits"] == fx_grad
        assert la == lb
        for (n, p), (_, q) in zip(a.model.named_parameters(), b.model.named_parameters()):
            assert torch.equal(p, q), n
    else:
        assert "fx_state" not in state
    # parameter EMA (upstream ema_decay=0.95): follows torch_ema's recurrence, travels with the checkpoint, and
    # evaluation runs on the averaged parameters and puts the live ones back
    assert "ema" in state and a.ema.num_updates == 4 == b.ema.num_updates
    for sa, sb in zip(a.ema.shadow, b.ema.shadow):
        assert torch.allclose(sa, sb, atol=1e-7)
    w = a.model.sigma_net[0].weight
    live = w.detach().clone()
    shadow = a.ema.shadow[[id(q) for q in a.ema.params].index(id(w))]
    assert not torch.equal(live, shadow)
    seen = {}
    orig_eval = a.eval_step
    a.eval_step = lambda data: (seen.setdefault("w", w.detach().clone()), orig_eval(data))[1]
    a.evaluate([batch(5)])
    assert torch.equal(seen["w"], shadow) and torch.equal(w.detach(), live)


@pytest.mark.parametrize("density_scale", [1.0, 300.0])
def test_fused_terminate_equals_two_kernel_path(params_k16, room, room_bitfield, density_scale):
    """Field + compositing in one launch with per-group early termination renders the same image and instance
    logits as the two-kernel path; on an opaque scene it evaluates only a fraction of the marched samples."""
    from instance_nerf_amd.nerf.utils import get_rays
    net = _network(params_k16, K=16).eval()
    net.density_bitfield.copy_(_t(room_bitfield))
    net.density_scale = density_scale
    poses, intr, H, W = room.cameras(n=2, H=64, W=64, focal=32.0)
    r = get_rays(_t(poses[1:2]), intr, 64, 64, patch=4)
    with torch.no_grad():
        a = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
        b = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused_terminate")
    assert (a["image"] - b["image"]).abs().max() < 1e-4
    assert (a["weights_sum"] - b["weights_sum"]).abs().max() < 1e-4
    assert (a["instance"] - b["instance"]).abs().max() < 1e-3
    total, ev = int(b["num_samples"][0]), int(b["num_evaluated"][0])
    assert 0 < ev <= total
    if density_scale > 1:
        assert ev < 0.6 * total          # opaque: most samples behind the first surface are never evaluated
    else:
        assert ev > 0.9 * total


@pytest.mark.parametrize("mode,K", [("fused", 0), ("fused", 16), ("fused_terminate", 16)])
def test_frame_pipeline_is_bit_identical(params_k16, room, room_bitfield, mode, K):
    """FramePipeline (views alternate on two streams, field kernels serialised, overlap placement on): same bits as
    net.render for view after view, buffers of finished views being reused while the next view is in flight."""
    from instance_nerf_amd.nerf.renderer import FramePipeline
    from instance_nerf_amd.nerf.utils import get_rays
    net = _network(params_k16, K=K).eval()
    net.density_bitfield.copy_(_t(room_bitfield))
    poses, intr, H, W = room.cameras(n=3, H=96, W=96, focal=48.0)
    rays = [get_rays(_t(poses[v:v + 1]), intr, 96, 96, patch=4) for v in range(3)]
    keys = ("image", "depth", "weights_sum") + (("instance",) if K else ())
    with torch.no_grad():
        ref = [net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode=mode) for r in rays]
    torch.cuda.synchronize()
    pipe = FramePipeline(net)
    with torch.no_grad():
        outs = [pipe.render(rays[k % 3]["rays_o"], rays[k % 3]["rays_d"], bg_color=1, infer_mode=mode) for k in range(9)]
    assert [o["stream"] for o in outs[:4]] == [pipe.streams[0], pipe.streams[1]] * 2
    pipe.close()
    for k, o in enumerate(outs):
        for key in keys:
            assert torch.equal(o[key], ref[k % 3][key]), (k, key)
    with pytest.raises(RuntimeError):
        pipe.render(rays[0]["rays_o"], rays[0]["rays_d"], staged=True)


def test_sliced_frames_through_the_frame_pipeline():
    """The sliced frame path under FramePipeline: views alternate on two streams and share the network's ONE workspace of
    precomputed fine-level features - both launches of a view sit inside the field gate, so view i+1's pre-pass cannot
    overwrite what view i's fused kernel still reads.  Nine views of three sizes (the workspace is reused and regrown),
    bit-identical to the fused kernel view by view."""
    from instance_nerf_amd.nerf.renderer import FramePipeline
    from instance_nerf_amd.nerf.utils import get_rays
    from instance_nerf_amd.scene import RoomScene
    from oracle import field, hashgrid
    bound = 4
    big = RoomScene(scale=float(bound))
    p = field.init_params(seed=2, table=hashgrid.level_table(desired_resolution=2048 * bound), table_std=1.0, K=0)
    net = _network(p, K=0, bound=bound).eval()
    net.density_bitfield.copy_(_t(big.density_bitfield(128, float(bound))))
    sizes = (160, 96, 240)
    rays = []
    for v, S in enumerate(sizes):
        poses, intr, H, W = big.cameras(n=3, H=S, W=S, focal=S / 2.0)
        rays.append(get_rays(_t(poses[v:v + 1]), intr, S, S, patch=4))
    net.frame_slices = False
    with torch.no_grad():
        ref = [net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused", dt_gamma=1 / 128) for r in rays]
    torch.cuda.synchronize()
    net.frame_slices = True
    pipe = FramePipeline(net)
    with torch.no_grad():
        outs = [pipe.render(rays[k % 3]["rays_o"], rays[k % 3]["rays_d"], bg_color=1, infer_mode="fused", dt_gamma=1 / 128)
                for k in range(9)]
    pipe.close()
    assert net.__dict__.get("_slice_ws") is not None
    for k, o in enumerate(outs):
        for key in ("image", "depth", "weights_sum"):
            assert torch.equal(o[key], ref[k % 3][key]), (k, key)


@pytest.mark.parametrize("K", [0, 16])
def test_trainer_view_loops_go_through_the_frame_pipeline_bit_identically(params_k16, room, room_bitfield, K, tmp_path):
    """Trainer.evaluate_one_epoch / test / render_sequence render a loader's views through FramePipeline (two alternating
    streams): every view's image, depth and instance logits carry the bits of eval_step / test_step on that view alone,
    the metric is the one-stream metric, and the PNG files are the same bytes."""
    from instance_nerf_amd.nerf.utils import Trainer, get_rays
    net = _network(params_k16, K=K)
    net.density_bitfield.copy_(_t(room_bitfield))
    poses, intr, H, W = room.cameras(n=5, H=64, W=64, focal=32.0)
    views = []
    for v in range(5):
        r = get_rays(_t(poses[v:v + 1]), intr, H, W)              # row-major rays, as upstream's loaders hand them over
        d = {"rays_o": r["rays_o"], "rays_d": r["rays_d"], "H": H, "W": W}
        if K:
            d["masks"] = torch.randint(-1, K, (1, H * W), device=DEV)
        else:
            d["images"] = torch.rand(1, H, W, 3, device=DEV)
        views.append(d)
    tr = Trainer("seq", None, net, stage="instance" if K else "nerf", device=torch.device(DEV), workspace=str(tmp_path),
                 use_checkpoint="scratch", mute=True)
    net.eval()
    ref = [tr.test_step(d) for d in views]
    seq = [tr._test_outputs(d, o) for d, o in tr.render_sequence(views)]
    one = [tr._test_outputs(d, o) for d, o in tr.render_sequence(views, pipeline=False)]
    for a, b, c in zip(ref, seq, one):
        for x, y, z in zip(a, b, c):
            assert (x is None and y is None) or (torch.equal(x, y) and torch.equal(x, z))
    tr.pipeline_views = True
    r1 = tr.evaluate_one_epoch(views)
    files1 = tr.test(views, save_path=str(tmp_path / "a"), name="v")
    tr.pipeline_views = False
    r0 = tr.evaluate_one_epoch(views)
    files0 = tr.test(views, save_path=str(tmp_path / "b"), name="v")
    assert r1 == r0 and len(files1) == len(files0) == 5
    for f1, f0 in zip(files1, files0):
        assert open(f1, "rb").read() == open(f0, "rb").read()


def test_instance_head_with_31_classes_runs_on_the_fused_kernels(level_table, room, room_bitfield):
    """K = 31 (the reference's 30 detections + background, run_rcnn.py:75 / match_seg.py:69) is not a multiple of the
    16-channel MFMA tile: the fused kernels run it with a zero-padded output layer.  Nothing padded leaks: logits,
    rendered logits and the weight gradient have 31 channels, values equal the composable (unfused) path, and a
    training step moves the loss."""
    from oracle import field
    from instance_nerf_amd.nerf.utils import get_rays
    p = field.init_params(seed=3, table=level_table, table_std=1.0, K=31)
    net = _network(p, K=31)
    assert net._fusable_inst and net._k_pad == 32
    net.density_bitfield.copy_(_t(room_bitfield))
    x = torch.rand(1000, 3, device=DEV) * 1.8 - 0.9
    net.eval()
    with torch.no_grad():
        fused = net.instance(x)
        net._fusable_inst = False
        plain = net.instance(x)
        net._fusable_inst = True
    assert fused.shape == plain.shape == (1000, 31) and fused.is_contiguous()
    assert (fused - plain).abs().max() < 1e-3 * max(1.0, float(plain.abs().max()))
    poses, intr, H, W = room.cameras(n=1, H=64, W=64, focal=32.0)
    r = get_rays(_t(poses[:1]), intr, 64, 64, patch=4)
    with torch.no_grad():
        a = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")          # k_instance_render, 32 channels
        b = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused_raymajor")  # logits + K-channel compositing
    assert a["instance"].shape == b["instance"].shape == (1, 4096, 31)
    assert (a["instance"] - b["instance"]).abs().max() < 1e-3 * max(1.0, float(b["instance"].abs().max()))
    net.train()
    net.freeze_nerf()
    out = net.render(r["rays_o"][:, :1024], r["rays_d"][:, :1024], bg_color=1, perturb=False, force_all_rays=True)
    labels = torch.randint(0, 31, (1024,), device=DEV)
    loss = torch.nn.functional.cross_entropy(out["instance"][0], labels)
    loss.backward()
    g = net.instance_net[2].weight.grad
    assert g.shape == (31, 64) and torch.isfinite(g).all() and g.abs().max() > 0
    assert net.instance_encoder.embeddings.grad.abs().max() > 0


def test_parameter_ema_inside_the_optimiser_launch():
    """FusedAdam.attach_ema: shadow += (1 - d) * (param - shadow) applied in the Adam kernel equals torch_ema's update
    applied after the step (ParamEMA.update), including for a tensor that has no gradient in one of the steps, an odd
    length (scalar tail of the float4 loop) and the warm-up of the decay (d = min(decay, (1 + n) / (10 + n)))."""
    from instance_nerf_amd.nerf.utils import FusedAdam, ParamEMA
    torch.manual_seed(0)
    shapes = [(1003,), (64, 32), (7,)]

    def setup():
        ps = [torch.nn.Parameter(torch.randn(*sh, device=DEV)) for sh in shapes]
        return ps, FusedAdam([{"params": ps, "lr": 1e-2}]), ParamEMA(ps, 0.95)
    torch.manual_seed(1); pa, oa, ea = setup()
    torch.manual_seed(1); pb, ob, eb = setup()
    oa.attach_ema(ea)
    for step in range(12):
        gs = [torch.randn_like(p) for p in pa]
        for k, (x, y) in enumerate(zip(pa, pb)):
            x.grad = y.grad = None
            if not (step == 3 and k == 1):             # tensor 1 has no gradient in step 3
                x.grad, y.grad = gs[k].clone(), gs[k].clone()
        oa.step(); ea.update()                         # update(): nothing left to do
        ob.step(); eb.update()                         # the reference order of upstream's loop
        assert ea.num_updates == eb.num_updates == step + 1
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
    for sa, sb in zip(ea.shadow, eb.shadow):
        assert (sa - sb).abs().max() < 1e-6


@pytest.mark.parametrize("M", [524288 - 16, 524288, 524288 + 1, 524288 + 16 * 7 + 3, 2 * 524288 + 16 * 1024 * 3 + 5,
                               8 * 524288 + 16 * 8191 + 15])
def test_interleaved_xcd_schedule_covers_every_tile_once(params_k16, M):
    """From 32768 tiles on, the fused field kernels deal 1024-tile chunks of the tile stream round robin to the XCDs and
    split what is left after the last complete round into eighths (field_fused.hip::make_sched); the second half of the
    rounds and the remainder are drawn from per-XCD cursors, with stealing (TileWalk).  Sizes around the switch-over,
    with empty / tiny / ragged remainders: every sample is evaluated - the outputs equal the evaluation in pieces that
    are too small for either schedule, bit for bit."""
    net = _network(params_k16, K=0).eval()
    g = torch.Generator(device=DEV).manual_seed(M)
    x = torch.rand(M, 3, device=DEV, generator=g) * 1.9 - 0.95
    d = torch.nn.functional.normalize(torch.randn(M, 3, device=DEV, generator=g), dim=-1)
    with torch.no_grad():
        s_all, c_all = net(x, d)
        pieces = [net(x[a:a + 400000], d[a:a + 400000]) for a in range(0, M, 400000)]
    assert torch.equal(s_all, torch.cat([p[0] for p in pieces]))
    assert torch.equal(c_all, torch.cat([p[1] for p in pieces]))
    assert torch.isfinite(s_all).all() and torch.isfinite(c_all).all()


def test_auto_mode_follows_the_skippable_fraction(params_k16, room, room_bitfield):
    """infer_mode="auto": the fraction of marched samples that lie behind the point where their whole 16-ray group
    has terminated - counted by the compositing kernel of the two-kernel path, reported by the terminating kernel
    itself - selects the early-terminating kernel only where it skips enough (read through a pinned buffer once its
    copy has landed: no call waits for a previous frame).  Opacity alone is not the criterion: a scene can be opaque
    and still need nearly every sample."""
    from instance_nerf_amd.nerf.utils import get_rays
    poses, intr, H, W = room.cameras(n=2, H=64, W=64, focal=32.0)
    r = get_rays(_t(poses[1:2]), intr, 64, 64, patch=4)
    for density_scale, opaque in ((300.0, True), (1e-3, False)):
        net = _network(params_k16, K=16).eval()
        net.density_bitfield.copy_(_t(room_bitfield))
        net.density_scale = density_scale
        with torch.no_grad():
            first = net.render(r["rays_o"], r["rays_d"], bg_color=1)            # nothing known yet: two-kernel path
            assert "num_evaluated" not in first
            torch.cuda.synchronize()
            assert (net._recent_skippable() > net.terminate_above) == opaque
            second = net.render(r["rays_o"], r["rays_d"], bg_color=1)
        assert ("num_evaluated" in second) == opaque
        assert (first["image"] - second["image"]).abs().max() < 1e-4
        for _ in range(8):                                                         # slots are recycled, never exhausted
            with torch.no_grad():
                net.render(r["rays_o"], r["rays_d"], bg_color=1)
        torch.cuda.synchronize()
        net._recent_skippable()
        assert len(net._skippable_free) == 4 and not net._skippable_pending
        if opaque:        # the counter of the two-kernel path predicts what the terminating kernel really skips
            with torch.no_grad():
                t = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused_terminate")
            real = 1.0 - int(t["num_evaluated"][0]) / int(t["num_samples"][0])
            net.__dict__.pop("_skippable_value", None)
            with torch.no_grad():
                net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
            torch.cuda.synchronize()
            assert abs(net._recent_skippable() - real) < 0.02, (net._recent_skippable(), real)
    # a HALF-transparent scene (rays of a 16-ray group terminate at different depths): both paths must report the SAME
    # fraction - samples of the steps at which the whole group is dead (round-2 advisor: the terminating kernel used to
    # count live rays only, so its fraction was larger and "auto" stuck to it once entered)
    net = _network(params_k16, K=16).eval()
    net.density_bitfield.copy_(_t(room_bitfield))
    seen = []
    for density_scale in (3.0, 20.0, 100.0):
        net.density_scale = density_scale
        net.__dict__.pop("_skippable_value", None)
        with torch.no_grad():
            net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
            torch.cuda.synchronize()
            predicted = net._recent_skippable()
            t = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused_terminate")
        real = 1.0 - int(t["num_evaluated"][0]) / int(t["num_samples"][0])
        assert abs(predicted - real) < 1e-3, (density_scale, predicted, real)
        seen.append(real)
    assert any(0.02 < v < 0.9 for v in seen), seen          # at least one of them is genuinely in between


@pytest.mark.parametrize("mode", ["auto", "fused_terminate", "fused"])
def test_staged_render_on_an_opaque_scene_two_frames(params_k16, room, room_bitfield, mode):
    """Regression (round-1 advisor): staged=True with the default infer_mode on an opaque scene.  The early-terminating
    branch adds a [1]-shaped counter to the results; staged rendering must concatenate per-ray results only, use ONE
    mode for all chunks of a frame, and survive the switch of modes between the first and second frame.  This is the
    call Trainer.eval_step / test_step make."""
    from instance_nerf_amd.nerf.utils import get_rays, Trainer
    net = _network(params_k16, K=16).eval()
    net.density_bitfield.copy_(_t(room_bitfield))
    net.density_scale = 300.0
    net.min_staged_batch = 0
    poses, intr, H, W = room.cameras(n=2, H=64, W=64, focal=32.0)
    r = get_rays(_t(poses[0:2]), intr, 64, 64, patch=4)                       # B = 2 views of 4096 rays
    with torch.no_grad():
        whole = net.render(r["rays_o"][:1], r["rays_d"][:1], bg_color=1, infer_mode="fused")
        frames = []
        for _ in range(2):
            out = net.render(r["rays_o"], r["rays_d"], staged=True, max_ray_batch=1008, bg_color=1, infer_mode=mode)
            torch.cuda.synchronize()
            frames.append(out)
    for out in frames:
        assert set(out) >= {"image", "depth", "weights_sum", "instance"}
        assert "num_samples" not in out and "num_evaluated" not in out
        assert out["image"].shape == (2, 4096, 3) and out["instance"].shape == (2, 4096, 16)
        assert (out["image"][:1] - whole["image"]).abs().max() < 1e-4
        assert (out["weights_sum"][:1] - whole["weights_sum"]).abs().max() < 1e-4
        assert (out["instance"][:1] - whole["instance"]).abs().max() < 1e-3
    if mode == "auto":
        assert net._recent_skippable() > net.terminate_above                    # the second frame took the other branch
    # the trainer's evaluation calls (default mode, staged) on the same opaque scene, twice
    tr = Trainer("t", None, net, stage="instance", device=torch.device(DEV))
    data = {"rays_o": r["rays_o"], "rays_d": r["rays_d"], "masks": torch.zeros(2, 4096, dtype=torch.int64, device=DEV)}
    for _ in range(2):
        pred, depth, truth, loss = tr.eval_step(data)
        img, dep, inst = tr.test_step(data)
        assert pred.shape == (2, 4096) and img.shape == (2, 4096, 3) and inst.shape == (2, 4096, 16)
        assert torch.isfinite(loss)


def test_trainer_renders_row_major_views_in_patch_order(params_k16, room, room_bitfield):
    """Upstream's evaluation loaders pass the H*W rays of a view in row-major order with H and W beside them; the
    Trainer renders them patch by patch (compact 16-ray groups) and returns every per-ray output in the caller's order:
    the same bits as rendering the row-major rays directly."""
    from instance_nerf_amd.nerf.utils import get_rays, Trainer
    net = _network(params_k16, K=16).eval()
    net.density_bitfield.copy_(_t(room_bitfield))
    poses, intr, H, W = room.cameras(n=1, H=64, W=96, focal=40.0)
    r = get_rays(_t(poses[:1]), intr, 64, 96)                                  # row-major, as upstream's loader
    with torch.no_grad():
        direct = net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")
    tr = Trainer("t", None, net, stage="instance", device=torch.device(DEV))
    seen = {}
    inner = net.render
    net.render = lambda ro, rd, **kw: (seen.setdefault("first_dirs", rd[0, :16].clone()), inner(ro, rd, **kw))[1]
    data = {"rays_o": r["rays_o"], "rays_d": r["rays_d"], "H": 64, "W": 96}
    img, dep, inst = tr.test_step(data)
    net.render = inner
    assert img.shape == (1, 64, 96, 3) and dep.shape == (1, 64, 96) and inst.shape == (1, 64, 96, 16)
    assert torch.equal(img.reshape(1, -1, 3), direct["image"])
    assert torch.equal(dep.reshape(1, -1), direct["depth"])
    assert torch.equal(inst.reshape(1, -1, 16), direct["instance"])
    # the renderer saw a 4x4 patch first - pixels (0..3, 0..3) - not the first 16 pixels of row 0
    assert torch.equal(seen["first_dirs"], r["rays_d"][0].view(64, 96, 3)[:4, :4].reshape(16, 3))


@pytest.mark.parametrize("K,drop", [(64, False), (31, False), (16, True), (40, False)])
def test_instance_head_node_equals_the_composable_chain(level_table, room, room_bitfield, K, drop):
    """The one-node instance head (k_instance_fwd<.., enc only> + K-channel compositing forward; k_instance_head_bwd =
    compositing backward + recomputed hidden layers + input-gradient chain + the three weight gradients in ONE launch)
    against the round-2 chain of separate kernels on the same network and rays: rendered logits bit for bit (same
    forward arithmetic), every instance gradient to fp32 summation-order accuracy.  drop: the sample buffer is sized
    from a mean_count below the batch's total, so the last rays are dropped and rows nobody owns exist."""
    from oracle import field
    p = field.init_params(seed=21, table=level_table, table_std=1.0, K=K)
    ro, rd = scene_rays(room, 700, cam=3, seed=55)
    labels = np.random.default_rng(5).integers(-1, K, size=700)

    def run(fused, fused_ce=False):
        net = _network({k: v.clone() for k, v in p.items()}, K=K).train()
        net.density_bitfield.copy_(_t(room_bitfield))
        net.freeze_nerf()
        net.fused_instance_head = fused
        if drop:
            with torch.no_grad():
                full = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=True)
            net.local_step = 0
            net.mean_count = (int(full["num_samples"][0]) * 2 // 3 // 128) * 128
        extra = {"ce_labels": _t(labels).long()[None]} if fused_ce else {}
        out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=not drop, **extra)
        loss = torch.nn.functional.cross_entropy(out["instance"][0], _t(labels).long(), ignore_index=-1)
        if fused_ce:       # the loss formed inside the compositing launch: same value, and IT is what gets differentiated
            assert abs(float(out["instance_ce"]) - float(loss)) < 1e-5 * max(1.0, abs(float(loss)))
            loss = out["instance_ce"]
        else:
            assert "instance_ce" not in out
        loss.backward()
        grads = {n: q.grad.clone() for n, q in net.named_parameters() if q.grad is not None}
        return out["instance"][0].detach().clone(), grads
    inst_a, ga = run(True)
    inst_b, gb = run(False)
    inst_c, gc = run(True, fused_ce=True)
    assert inst_a.shape == (700, K) and torch.equal(inst_a, inst_b) and torch.equal(inst_a, inst_c)
    for n in ga:
        rel = float(torch.linalg.norm(gc[n] - ga[n]) / torch.linalg.norm(ga[n]))
        assert rel < 2e-5, ("fused ce", n, rel)
    if drop:
        assert (inst_a[-20:] == 0).all() and (inst_a[:20] != 0).any()      # the tail of the ray list was dropped
    assert sorted(ga) == sorted(gb) == ["instance_encoder.embeddings", "instance_net.0.weight", "instance_net.1.weight",
                                        "instance_net.2.weight"]
    for n in ga:
        assert ga[n].shape == gb[n].shape and gb[n].abs().sum() > 0, n
        rel = float(torch.linalg.norm(ga[n] - gb[n]) / torch.linalg.norm(gb[n]))
        assert rel < 2e-5, (n, rel)


@pytest.mark.parametrize("n", [900, 1])
def test_nerf_head_node_equals_the_unfused_chain(level_table, room, room_bitfield, n):
    """NeRF stage: the one-launch backward (k_nerf_head_bwd: forward recomputed from the saved encoder output, input-
    gradient chain, five weight gradients on the fp32 matrix cores) against the round-2 chain (saved activations,
    k_nerf_bwd, five split-K weight-gradient launches) on the same network and rays: image bit for bit, every
    gradient to summation-order accuracy."""
    from oracle import field
    p = field.init_params(seed=23, table=level_table, table_std=1.0, K=0)
    ro, rd = scene_rays(room, n, cam=2, seed=56)
    target = np.random.default_rng(6).random((n, 3)).astype(np.float32)

    def run(fused):
        net = _network({k: v.clone() for k, v in p.items()}, K=0).train()
        net.density_bitfield.copy_(_t(room_bitfield))
        net.fused_nerf_head = fused
        out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=True)
        ((out["image"][0] - _t(target)) ** 2).mean().backward()
        return out["image"][0].detach().clone(), {k: q.grad.clone() for k, q in net.named_parameters() if q.grad is not None}
    img_a, ga = run(True)
    img_b, gb = run(False)
    assert torch.equal(img_a, img_b)
    assert sorted(ga) == sorted(gb) and len(ga) == 6
    for k in ga:
        assert ga[k].shape == gb[k].shape, k
        if gb[k].abs().sum() == 0:
            assert ga[k].abs().sum() == 0, k
            continue
        rel = float(torch.linalg.norm(ga[k] - gb[k]) / torch.linalg.norm(gb[k]))
        assert rel < 2e-5, (k, rel)


@pytest.mark.parametrize("bg", ["white", "triple", "per_ray"])
@pytest.mark.parametrize("n", [700, 1])
def test_fused_image_loss_equals_the_torch_tail(level_table, room, room_bitfield, n, bg):
    """NeRF stage: render(mse_target=gt) - background blend, depth normalisation, mean squared error and its gradients
    in one launch each way (inr_finish_rays_mse) - against the same render followed by torch's element-wise tail and
    ``MSELoss(reduction='none')(pred, gt).mean()`` (upstream's train_step): image, depth and loss to fp32 rounding,
    every parameter gradient to summation-order accuracy; also with the loss scaled (fp16 grad scaler) and with the
    shaded image used a second time outside the loss."""
    from oracle import field
    p = field.init_params(seed=29, table=level_table, table_std=1.0, K=0)
    ro, rd = scene_rays(room, n, cam=1, seed=57)
    rng = np.random.default_rng(8)
    target = _t(rng.random((1, n, 3)).astype(np.float32))
    bg_color = {"white": 1, "triple": (0.2, 0.5, 0.9), "per_ray": _t(rng.random((1, n, 3)).astype(np.float32))}[bg]
    probe = _t(rng.standard_normal((1, n, 3)).astype(np.float32))

    def run(fused, scale, reuse):
        net = _network({k: v.clone() for k, v in p.items()}, K=0).train()
        net.density_bitfield.copy_(_t(room_bitfield))
        kw = dict(mse_target=target) if fused else {}
        out = net.render(_t(ro)[None], _t(rd)[None], bg_color=bg_color, perturb=False, force_all_rays=True, **kw)
        assert ("image_mse" in out) == fused
        loss = out["image_mse"] if fused else torch.nn.MSELoss(reduction="none")(out["image"], target).mean()
        total = loss * scale
        if reuse:
            total = total + (out["image"] * probe).sum() * 1e-3
        total.backward()
        grads = {k: q.grad.clone() for k, q in net.named_parameters() if q.grad is not None}
        return out["image"].detach(), out["depth"].detach(), loss.detach(), grads
    for scale, reuse in ((1.0, False), (1024.0, False), (1.0, True)):
        img_a, dep_a, loss_a, ga = run(True, scale, reuse)
        img_b, dep_b, loss_b, gb = run(False, scale, reuse)
        assert float((img_a - img_b).abs().max()) < 5e-7 and float((dep_a - dep_b).abs().max()) < 5e-7
        assert abs(float(loss_a) - float(loss_b)) <= 2e-6 * abs(float(loss_b))
        assert sorted(ga) == sorted(gb) and len(ga) == 6
        for k in ga:
            if gb[k].abs().sum() == 0:
                assert ga[k].abs().sum() == 0, k
                continue
            rel = float(torch.linalg.norm(ga[k] - gb[k]) / torch.linalg.norm(gb[k]))
            assert rel < 2e-5, (k, scale, reuse, rel)


def test_trainer_nerf_stage_uses_the_fused_image_loss(level_table, room, room_bitfield):
    """Trainer.train_step (stage 'nerf', default criterion) takes the loss from the renderer's fused tail; a custom
    criterion keeps upstream's ``criterion(pred, gt).mean()``; both return the same (pred, gt, loss) for an RGBA batch
    once the random background is pinned."""
    from instance_nerf_amd.nerf.utils import Trainer
    from oracle import field
    p = field.init_params(seed=31, table=level_table, table_std=1.0, K=0)
    n = 512
    ro, rd = scene_rays(room, n, cam=0, seed=58)
    rgba = _t(np.random.default_rng(9).random((1, n, 4)).astype(np.float32))
    data = {"rays_o": _t(ro)[None], "rays_d": _t(rd)[None], "images": rgba}
    out = []
    for criterion in (None, torch.nn.MSELoss(reduction="none")):
        net = _network({k: v.clone() for k, v in p.items()}, K=0)
        net.density_bitfield.copy_(_t(room_bitfield))
        tr = Trainer("t", None, net, criterion=criterion, stage="nerf", device=torch.device(DEV), workspace=None,
                     mute=True, update_extra_interval=10 ** 9)
        net.train()
        torch.manual_seed(5)                         # the per-ray random background and the march jitter
        seen = {}
        render = net.render

        def spy(*a, **k):
            seen["fused"] = "mse_target" in k
            r = render(*a, **k)
            seen["key"] = "image_mse" in r
            return r
        net.render = spy
        pred, gt, loss = tr.train_step(data)
        # a whole step (backward seeded with the cached unit gradient: no ones_like fill, no multiplication by one)
        torch.manual_seed(5)
        tr.global_step = 1
        step_loss = float(tr.train_one_step(data))
        grads = {k: q.grad.clone() for k, q in net.named_parameters() if q.grad is not None}
        out.append((pred.detach(), gt, float(loss), dict(seen), step_loss, grads))
    assert out[0][3] == {"fused": True, "key": True} and out[1][3] == {"fused": False, "key": False}
    assert torch.equal(out[0][1], out[1][1])
    assert float((out[0][0] - out[1][0]).abs().max()) < 5e-7
    assert abs(out[0][2] - out[1][2]) <= 2e-6 * abs(out[1][2])
    assert abs(out[0][4] - out[1][4]) <= 2e-6 * abs(out[1][4]) and abs(out[0][4] - out[0][2]) <= 2e-6 * abs(out[0][2])
    ga, gb = out[0][5], out[1][5]
    assert sorted(ga) == sorted(gb) and len(ga) == 6
    for k in ga:
        rel = float(torch.linalg.norm(ga[k] - gb[k]) / torch.linalg.norm(gb[k]))
        assert rel < 2e-5, (k, rel)


def test_label_outside_the_classes_poisons_the_loss(level_table, room, room_bitfield):
    """torch's cross_entropy asserts on the device for a label that is neither ignore_index nor a class; both HIP
    losses - inr_cross_entropy and the epilogue of the K-channel compositing - return NaN instead of dropping the row
    silently (round-2 advisor: a detection-count mismatch would otherwise train on fewer rows unnoticed)."""
    from instance_nerf_amd import raymarching
    from oracle import field
    K = 16
    logits = torch.randn(50, K, device=DEV)
    labels = torch.randint(0, K, (50,), device=DEV)
    assert torch.isfinite(raymarching.cross_entropy(logits, labels))
    labels[7] = K
    assert torch.isnan(raymarching.cross_entropy(logits, labels))
    p = field.init_params(seed=2, table=level_table, table_std=1.0, K=K)
    net = _network(p, K=K).train()
    net.density_bitfield.copy_(_t(room_bitfield))
    net.freeze_nerf()
    ro, rd = scene_rays(room, 64, seed=3)
    lab = torch.randint(-1, K, (1, 64), device=DEV)
    out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=True, ce_labels=lab)
    assert torch.isfinite(out["instance_ce"])
    lab[0, 5] = K + 3
    out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=True, ce_labels=lab)
    assert torch.isnan(out["instance_ce"])
    # every label ignored: the mean over an empty set, NaN as in torch - and a backward that does not crash
    lab = torch.full((1, 64), -1, device=DEV)
    out = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, perturb=False, force_all_rays=True, ce_labels=lab)
    assert torch.isnan(out["instance_ce"])
    assert torch.isnan(torch.nn.functional.cross_entropy(out["instance"][0].detach(), lab[0], ignore_index=-1))
    out["instance_ce"].backward()
    assert net.instance_encoder.embeddings.grad is not None


def test_composite_train_with_dropped_rays(rm, room, room_bitfield, bits_dev):
    """Sample buffers sized from mean_count: rays that overflow M are dropped by the writer and must composite to
    zero (and get zero gradients) - never be read past the end of the buffers (regression: GPU memory fault)."""
    from oracle import rays
    ro, rd = scene_rays(room, 600, seed=91)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    _, _, _, rr_full = rm.march_rays_train(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars))
    total = int(rr_full[:, 2].sum())
    M = (total // 2 // 128) * 128
    x, d, dl, rr = rm.march_rays_train(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars), None, M, False, 128)
    assert x.shape[0] == M
    kept = ((rr[:, 1] + rr[:, 2]) <= M)
    gen = torch.Generator().manual_seed(0)
    sig = (torch.rand(M, generator=gen) * 30).to(DEV).requires_grad_(True)
    rgb = torch.rand(M, 3, generator=gen).to(DEV).requires_grad_(True)
    ext = torch.randn(M, 16, generator=gen).to(DEV).requires_grad_(True)
    ws, depth, img, eo = rm.composite_rays_train(sig, rgb, dl, rr, 1e-4, extra=ext)
    assert (ws[~kept] == 0).all() and (img[~kept] == 0).all() and (eo[~kept] == 0).all()
    assert (ws[kept & (rr[:, 2] > 0)] > 0).all()
    (ws.sum() + img.sum() + eo.sum()).backward()
    last = int((rr[kept, 1] + rr[kept, 2]).max())
    assert (sig.grad[last:] == 0).all() and (ext.grad[last:] == 0).all() and torch.isfinite(sig.grad).all()


@pytest.mark.parametrize("regime", ["dropped", "padded", "exact"])
def test_composite_backward_writes_every_row_when_given_the_total(rm, room, bits_dev, regime):
    """inr_composite_rays_train_backward with total_dev (the marcher's counter): the gradient buffers arrive
    uninitialised and the launch itself zeroes what no ray owns - the rows of the dropped ray that starts inside the
    buffer, the padding behind the total, the samples behind a ray's termination point.  Bit for bit the gradients of
    the zero-initialised call, with the allocator's free blocks poisoned beforehand."""
    from oracle import rays
    ro, rd = scene_rays(room, 500, seed=92)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    _, _, _, rr_full = rm.march_rays_train(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars))
    total = int(rr_full[:, 2].sum())
    M = {"dropped": (total * 2 // 3 // 128) * 128, "padded": total + 1000, "exact": -1}[regime]
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    x, d, dl, rr = rm.march_rays_train(_t(ro), _t(rd), 1.0, bits_dev, 1, 128, _t(nears), _t(fars), counter, M, False, -1,
                                       M < 0)
    M = x.shape[0]
    assert int(counter[0]) == total and (M == total) == (regime == "exact")
    gen = torch.Generator().manual_seed(1)
    sig0 = torch.rand(M, generator=gen) * 60            # opaque enough for early termination inside rays
    rgb0 = torch.rand(M, 3, generator=gen)
    g_img = torch.randn(500, 3, generator=gen).to(DEV)
    g_ws = torch.randn(500, generator=gen).to(DEV)

    def run(total_dev):
        sig, rgb = sig0.to(DEV).requires_grad_(True), rgb0.to(DEV).requires_grad_(True)
        ws, depth, img = rm.composite_rays_train(sig, rgb, dl, rr, 1e-4, total_dev=total_dev)
        poison = [torch.full((M,), float("nan"), device=DEV), torch.full((M, 3), float("nan"), device=DEV)]
        del poison                                       # the next empty_like of these sizes gets NaN-filled blocks
        torch.autograd.backward([img, ws], [g_img, g_ws])
        return sig.grad, rgb.grad
    gs_a, gc_a = run(None)
    gs_b, gc_b = run(counter)
    assert torch.isfinite(gs_b).all() and torch.isfinite(gc_b).all()
    assert torch.equal(gs_a, gs_b) and torch.equal(gc_a, gc_b)
    owned = torch.zeros(M, dtype=torch.bool, device=DEV)
    for off, cnt in rr[:, 1:].tolist():
        if off + cnt <= M:
            owned[off:off + cnt] = True
    assert (regime == "exact") == bool(owned.all())
    assert (gs_b[~owned] == 0).all() and (gc_b[~owned] == 0).all() and gs_b[owned].abs().sum() > 0


def test_dropping_a_prefetched_march_never_steps_over_a_later_slot(room):
    """Round-4 advisor (renderer.py: drop_ahead): a prefetched march that is dropped gives its step_counter slot back
    only while it is still the most recent one; with another training render in between the counter position stays,
    so the next render cannot overwrite that real step's sample total."""
    from instance_nerf_amd.nerf import NeRFNetwork
    net = NeRFNetwork(cuda_ray=True, num_instances=0, min_near=0.05).to(DEV).train()
    net.density_bitfield.copy_(_t(room.density_bitfield(128, 1.0)))
    net.mean_count = 60000
    ro, rd = (_t(a)[None] for a in scene_rays(room, 512, seed=3))
    ro2, rd2 = (_t(a)[None] for a in scene_rays(room, 512, seed=4))
    a = net.march_ahead(ro, rd)
    assert a is not None and a["slot_taken"] and a["slot_index"] == 0 and net.local_step == 1
    net.drop_ahead(a)                                   # still the newest slot: handed back
    assert net.local_step == 0
    a = net.march_ahead(ro, rd)                         # slot 0 again
    with torch.no_grad():
        net.render(ro2, rd2, bg_color=1, perturb=False)  # an unrelated training render takes slot 1
    assert net.local_step == 2
    total_1 = int(net.step_counter[1, 0])
    net.drop_ahead(a)                                   # NOT the newest slot any more: nothing moves
    assert net.local_step == 2 and not a["slot_taken"]
    with torch.no_grad():
        net.render(ro, rd, bg_color=1, perturb=False)    # goes to slot 2
    assert net.local_step == 3 and int(net.step_counter[1, 0]) == total_1 > 0


def test_look_ahead_march_changes_nothing_but_the_schedule(room):
    """Trainer.train_one_step(data, next_data): the next batch's ray/box test and march are queued on a side stream
    under this step's backward.  Same seeds, same batches, occupancy updates every 4 steps (no look-ahead across an
    update), jittered marching: per step the consumed march has the same sample total, the loss the same value up to
    the scatter's rounding order, and after 14 steps the parameters agree as two plain runs do."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    from instance_nerf_amd.nerf.utils import Trainer

    def run(ahead):
        torch.manual_seed(3)
        net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10).to(DEV)
        ds = SyntheticRoomDataset(torch.device(DEV), num_rays=1024)
        net.density_bitfield.copy_(torch.from_numpy(ds.room.density_bitfield(128, 1.0)).to(DEV))
        analytic = net.density_bitfield.clone()
        real = net.update_extra_state

        def update(*a, **kw):                      # the update runs (mean_count!) but the analytic grid is kept
            real(*a, **kw)
            net.density_bitfield.copy_(analytic)
        net.update_extra_state = update
        tr = Trainer("la", None, net, stage="nerf", device=torch.device(DEV), iters=100, update_extra_interval=4,
                     workspace=None, mute=True)
        tr.global_step = 1
        batches = [ds.batch() for _ in range(15)]
        torch.manual_seed(11)
        used, losses, totals = 0, [], []
        for i in range(14):
            losses.append(float(tr.train_one_step(batches[i], batches[i + 1] if ahead else None)))
            totals.append(int(net.last_counter[0]))
            used += int(tr._ahead is not None)
        return losses, totals, used, {k: v.detach().clone() for k, v in net.named_parameters()}
    la, ta, used_a, pa = run(True)
    lb, tb, used_b, pb = run(False)
    assert used_b == 0 and 6 <= used_a <= 11        # steady-state steps that are not followed by an update
    assert ta == tb
    for x, y in zip(la, lb):
        assert abs(x - y) <= 2e-4 * abs(y), (la, lb)
    for k in pa:
        d = (pa[k] - pb[k]).abs()
        assert float((d > 2e-3).float().mean()) < 1e-5, k


def test_training_from_scratch_with_mean_count_buffers(room):
    """40 NeRF steps from a zero occupancy grid: update_extra_state every 16 steps switches march_rays_train to
    buffers sized from mean_count (no host sync, overflowing rays dropped).  Loss stays finite and decreases."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    from instance_nerf_amd.nerf.utils import Trainer
    torch.manual_seed(0)
    dev = torch.device(DEV)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10).to(dev)
    ds = SyntheticRoomDataset(dev, H=200, W=200, n_views=8, num_rays=2048)
    tr = Trainer("scratch", None, net, stage="nerf", device=dev, lr=1e-2, iters=200)
    losses = [float(tr.train_one_step(ds.batch())) for _ in range(40)]
    assert net.mean_count > 0 and net.iter_density >= 2
    assert all(np.isfinite(losses))
    assert np.mean(losses[-8:]) < np.mean(losses[:8])
    assert 0.0 < float((net.density_bitfield != 0).float().mean()) <= 1.0


def test_mark_untrained_grid_matches_oracle(room):
    from instance_nerf_amd.nerf import NeRFNetwork
    from oracle import occupancy
    net = NeRFNetwork(cuda_ray=True, bound=2, min_near=0.05, grid_size=32).to(DEV)     # 2 cascades, small grid
    poses = np.stack([room.look_at([0.8, 0.1, 0.0], target=(2, 0.2, 0.1)), room.look_at([-0.5, 0.5, 0.3], target=(0, 2, 0))])
    intr = (40.0, 40.0, 32.0, 32.0)
    net.mark_untrained_grid(torch.from_numpy(poses), intr)
    ref = occupancy.mark_untrained_cells(poses, intr, 32, 2, 2.0)
    got = (net.density_grid.cpu().numpy() == -1)
    assert 0.05 < ref.mean() < 0.95
    assert (got != ref).mean() < 1e-3            # float rounding at frustum boundaries only
    # marked cells never become occupied by an update
    net.update_extra_state()
    assert (net.density_grid.cpu().numpy()[got] == -1).all()


def test_render_bound2_and_staged_chunks(room):
    """bound = 2 (two cascades, desired resolution 4096) end to end vs the oracle, and staged rendering in
    max_ray_batch chunks equals the single-batch result."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from oracle import field, hashgrid, render
    bound = 2.0
    tb = hashgrid.level_table(desired_resolution=4096)
    p = field.init_params(seed=5, table=tb, table_std=1.0)
    net = NeRFNetwork(cuda_ray=True, bound=2, min_near=0.2, grid_size=64).to(DEV).eval()
    assert net.cascade == 2 and (net.encoder.table["scales"] == tb["scales"]).all()
    net.load_state_dict({"encoder.embeddings": p["embeddings"], "sigma_net.0.weight": p["sigma_w0"],
                         "sigma_net.1.weight": p["sigma_w1"], "color_net.0.weight": p["color_w0"],
                         "color_net.1.weight": p["color_w1"], "color_net.2.weight": p["color_w2"]}, strict=False)
    rng = np.random.default_rng(2)
    bits = (rng.random(2 * 64 ** 3 // 8) < 0.02).astype(np.uint8) * rng.integers(1, 256, 2 * 64 ** 3 // 8).astype(np.uint8)
    net.density_bitfield.copy_(_t(bits))
    n = 300
    ro = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
    rd = rng.normal(size=(n, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    ref = render.render_train(ro, rd, p, tb, bits, bound=bound, cascade=2, H=64, min_near=0.2, dt_gamma=1 / 128,
                              max_steps=512)
    net.min_staged_batch = 0                       # upstream's exact chunking (the default floor is 2^20 rays)
    with torch.no_grad():
        a = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, dt_gamma=1 / 128, max_steps=512, infer_mode="fused")
        b = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, dt_gamma=1 / 128, max_steps=512, staged=True,
                       max_ray_batch=64, infer_mode="fused")
        c = net.render(_t(ro)[None], _t(rd)[None], bg_color=1, dt_gamma=1 / 128, max_steps=512, infer_mode="fused_terminate")
    assert int(a["num_samples"][0]) == ref["total"] > 500
    assert np.abs(a["image"][0].cpu().numpy() - ref["image"].detach().numpy()).max() < 1e-4
    assert (a["image"] - b["image"]).abs().max() < 1e-6 and (a["weights_sum"] - b["weights_sum"]).abs().max() < 1e-6
    assert (a["image"] - c["image"]).abs().max() < 1e-4


@pytest.mark.parametrize("K", [16, 48, 64])
def test_instance_field_fused_training_kernels(level_table, K):
    """C ABI of the fused instance-field training path: device-packed weights, forward with saved activations and
    the one-launch input-gradient chain, each against fp32 torch on the same inputs (ReLU masks taken from the
    kernel's own activations: a pre-activation within 1e-5 of zero may legitimately fall on either side)."""
    from instance_nerf_amd import _lib
    from instance_nerf_amd._lib import check, ptr, stream_ptr
    from oracle import field, hashgrid
    lib = _lib.load()
    p = field.init_params(seed=21, table=level_table, table_std=0.5, K=K)
    net = _network(p, K=K)
    gen = torch.Generator().manual_seed(4)
    M = 5000 + 7                                        # last tile is partial
    x = (torch.rand(M, 3, generator=gen) * 2 - 1).to(DEV)
    g = torch.randn(M, K, generator=gen).to(DEV)
    w0, w1, w2 = [l.weight.detach() for l in net.instance_net]
    emb = net.instance_encoder.embeddings.data
    pf = torch.empty(lib.inr_instance_packed_floats(K), device=DEV)
    pb = torch.empty(lib.inr_instance_bwd_packed_floats(), device=DEV)
    check(lib.inr_instance_pack_weights_device(ptr(w0), ptr(w1), ptr(w2), K, ptr(pf), ptr(pb), stream_ptr()), "pack")
    assert (pf.cpu() == net._packed_weights("instance").cpu()).all()      # same bits as the host packer
    logits, enc = torch.empty(M, K, device=DEV), torch.empty(M, 32, device=DEV)
    h1, h2 = torch.empty(M, 64, device=DEV), torch.empty(M, 64, device=DEV)
    check(lib.inr_instance_forward_train(ptr(x), M, 1.0, ptr(emb), net.instance_encoder.desc, ptr(pf), K, ptr(logits),
                                         ptr(enc), ptr(h1), ptr(h2), stream_ptr()), "forward_train")
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    with torch.no_grad():
        enc_r = hashgrid.encode(x.cpu(), p["inst_embeddings"], 1.0, level_table).to(DEV)
        h1_r = torch.relu(enc_r @ w0.t())
        h2_r = torch.relu(h1_r @ w1.t())
        lg_r = h2_r @ w2.t()
    assert rel(enc, enc_r) < 1e-5 and rel(h1, h1_r) < 1e-4 and rel(h2, h2_r) < 1e-4 and rel(logits, lg_r) < 1e-4
    with torch.no_grad():
        assert (net.eval().instance(x) == logits).all()                   # inference kernel: identical logits
    dz2, dz1, denc = torch.empty(M, 64, device=DEV), torch.empty(M, 64, device=DEV), torch.empty(M, 32, device=DEV)
    check(lib.inr_instance_backward(ptr(g), K, ptr(h1), ptr(h2), M, ptr(pb), ptr(dz2), ptr(dz1), ptr(denc),
                                    stream_ptr()), "backward")
    dz2_r = (g @ w2) * (h2 > 0)
    dz1_r = (dz2_r @ w1) * (h1 > 0)
    denc_r = dz1_r @ w0
    assert rel(dz2, dz2_r) < 1e-4 and rel(dz1, dz1_r) < 1e-4 and rel(denc, denc_r) < 1e-4
    # empty batch
    assert lib.inr_instance_backward(None, K, None, None, 0, None, None, None, None, stream_ptr()) == 0


def test_instance_field_fused_training_autograd(level_table):
    """End to end through autograd: the fused path and the composable path (HIP encoder + rocBLAS layers) give the
    same logits and gradients, up to the handful of samples whose ReLU pre-activation is within rounding of zero
    (those contribute to one path and not the other)."""
    from oracle import field
    K = 64
    p = field.init_params(seed=22, table=level_table, table_std=0.5, K=K)
    net = _network(p, K=K).train()
    gen = torch.Generator().manual_seed(5)
    M = 6000
    x = (torch.rand(M, 3, generator=gen) * 2 - 1).to(DEV)
    gy = torch.randn(M, K, generator=gen).to(DEV)
    params = [net.instance_encoder.embeddings] + [l.weight for l in net.instance_net]
    res = {}
    for fused in (True, False):
        net.fused_instance_train = fused
        for q in params:
            q.grad = None
        out = net.instance(x)
        assert out.requires_grad
        out.backward(gy)
        res[fused] = (out.detach().clone(), [q.grad.detach().clone() for q in params])
    assert (res[True][0] - res[False][0]).abs().max() < 1e-4 * res[False][0].abs().max()
    for a, b in zip(res[True][1], res[False][1]):
        assert a.shape == b.shape
        assert torch.linalg.norm(a - b) < 2e-2 * torch.linalg.norm(b)
        if a.shape[0] > 64:          # the table: a flipped sample only touches its own 16 x 8 rows
            bad = ((a - b).abs() > 1e-3 * b.abs().max()).sum().item()
            assert bad <= 0.002 * int((b != 0).sum()), bad


def test_nerf_field_fused_training_kernels(level_table):
    """C ABI of the fused NeRF-field training path against fp32 torch: device-packed weights (same bits as the host
    packer), forward with saved activations, and the one-launch input-gradient chain (ReLU masks from the kernel's own
    activations)."""
    from instance_nerf_amd import _lib
    from instance_nerf_amd._lib import check, ptr, stream_ptr
    from oracle import field, hashgrid, sh
    lib = _lib.load()
    p = field.init_params(seed=31, table=level_table, table_std=0.5)
    net = _network(p, K=0)
    gen = torch.Generator().manual_seed(6)
    M = 4000 + 5
    x = (torch.rand(M, 3, generator=gen) * 2 - 1).to(DEV)
    d = torch.randn(M, 3, generator=gen)
    d = (d / d.norm(dim=1, keepdim=True)).to(DEV)
    ws0, ws1 = [l.weight.detach() for l in net.sigma_net]
    wc0, wc1, wc2 = [l.weight.detach() for l in net.color_net]
    emb = net.encoder.embeddings.data
    pf = torch.empty(lib.inr_nerf_packed_floats(), device=DEV)
    pb = torch.empty(lib.inr_nerf_bwd_packed_floats(), device=DEV)
    check(lib.inr_nerf_pack_weights_device(ptr(ws0), ptr(ws1), ptr(wc0), ptr(wc1), ptr(wc2), ptr(pf), ptr(pb),
                                           stream_ptr()), "pack")
    assert (pf.cpu() == net._packed_weights("nerf").cpu()).all()
    E = lambda w: torch.empty(M, w, device=DEV)
    sigma, rgb = torch.empty(M, device=DEV), E(3)
    enc, h1, so, cin, c1, c2 = E(32), E(64), E(16), E(32), E(64), E(64)
    check(lib.inr_nerf_forward_train(ptr(x), ptr(d), M, 1.0, ptr(emb), net.encoder.desc, ptr(pf), ptr(sigma), ptr(rgb),
                                     ptr(enc), ptr(h1), ptr(so), ptr(cin), ptr(c1), ptr(c2), stream_ptr()), "fwd")
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    with torch.no_grad():
        enc_r = hashgrid.encode(x.cpu(), p["embeddings"], 1.0, level_table).to(DEV)
        h1_r = torch.relu(enc_r @ ws0.t())
        so_r = h1_r @ ws1.t()
        sh_r = sh.sh_encode(d.cpu()).to(DEV) if hasattr(sh, "sh_encode") else net.encoder_dir(d)
        cin_r = torch.cat([sh_r, so_r[:, 1:]], -1)
        c1_r = torch.relu(cin_r @ wc0.t())
        c2_r = torch.relu(c1_r @ wc1.t())
        rgb_r = torch.sigmoid(c2_r @ wc2.t())
        s0, r0 = net(x, d)                              # inference kernel
    assert rel(enc, enc_r) < 1e-5 and rel(h1, h1_r) < 1e-4 and rel(so, so_r) < 1e-4
    assert rel(cin[:, :31], cin_r) < 1e-4 and (cin[:, 31] == 0).all()
    assert rel(c1, c1_r) < 1e-4 and rel(c2, c2_r) < 1e-4 and rel(rgb, rgb_r) < 1e-4
    assert rel(sigma, torch.exp(so_r[:, 0])) < 1e-4
    assert (sigma == s0).all() and (rgb == r0).all()
    g_sigma = torch.randn(M, generator=gen).to(DEV)
    g_rgb = torch.randn(M, 3, generator=gen).to(DEV)
    d_o, dz_c2, dz_c1, d_so, dz_h1, d_enc = E(4), E(64), E(64), E(16), E(64), E(32)
    check(lib.inr_nerf_backward(ptr(g_sigma), ptr(g_rgb), ptr(rgb), ptr(so), ptr(h1), ptr(c1), ptr(c2), M, 1.0, ptr(pb),
                                ptr(d_o), ptr(dz_c2), ptr(dz_c1), ptr(d_so), ptr(dz_h1), ptr(d_enc), stream_ptr()), "bwd")
    do_r = g_rgb * rgb * (1 - rgb)
    dzc2_r = (do_r @ wc2) * (c2 > 0)
    dzc1_r = (dzc2_r @ wc1) * (c1 > 0)
    dcin_r = dzc1_r @ wc0
    dso_r = torch.cat([(g_sigma * torch.exp(so[:, 0].clamp(-15, 15)))[:, None], dcin_r[:, 16:]], -1)
    dzh1_r = (dso_r @ ws1) * (h1 > 0)
    denc_r = dzh1_r @ ws0
    assert rel(d_o[:, :3], do_r) < 1e-5 and (d_o[:, 3] == 0).all()
    assert rel(dz_c2, dzc2_r) < 1e-4 and rel(dz_c1, dzc1_r) < 1e-4 and rel(d_so, dso_r) < 1e-4
    assert rel(dz_h1, dzh1_r) < 1e-4 and rel(d_enc, denc_r) < 1e-4


def test_nerf_field_fused_training_autograd(level_table):
    """End to end through autograd: fused vs composable NeRF-field training path (see the instance-field twin)."""
    from oracle import field
    p = field.init_params(seed=32, table=level_table, table_std=0.5)
    net = _network(p, K=0).train()
    gen = torch.Generator().manual_seed(7)
    M = 6000
    x = (torch.rand(M, 3, generator=gen) * 2 - 1).to(DEV)
    d = torch.randn(M, 3, generator=gen)
    d = (d / d.norm(dim=1, keepdim=True)).to(DEV)
    gs, gc = torch.randn(M, generator=gen).to(DEV), torch.randn(M, 3, generator=gen).to(DEV)
    params = net._nerf_params()
    res = {}
    for fused in (True, False):
        net.fused_nerf_train = fused
        for q in params:
            q.grad = None
        sigma, rgb = net(x, d)
        assert sigma.requires_grad and rgb.requires_grad
        ((sigma * gs).sum() + (rgb * gc).sum()).backward()
        res[fused] = (sigma.detach().clone(), rgb.detach().clone(), [q.grad.detach().clone() for q in params])
    assert (res[True][0] - res[False][0]).abs().max() < 1e-4 * res[False][0].abs().max()
    assert (res[True][1] - res[False][1]).abs().max() < 1e-5
    for a, b in zip(res[True][2], res[False][2]):
        assert a.shape == b.shape
        assert torch.linalg.norm(a - b) < 2e-2 * torch.linalg.norm(b)


@pytest.mark.parametrize("opt_kind", ["torch_adam", "fused_adam"])
def test_trainer_built_the_way_upstreams_main_script_builds_it(tmp_path, room, opt_kind):
    """The construction of upstream's main_nerf.py, verbatim in shape: optimizer = lambda model: Adam(model.get_params(lr)),
    lr_scheduler = lambda optimizer: LambdaLR(...), metrics=[PSNRMeter()], fp16=True (accepted, fp32 is computed),
    use_checkpoint='latest', then train(train_loader, valid_loader, max_epochs) and test(loader): epochs are
    checkpointed with rotation, the validation loss is the result, the best checkpoint is kept, a second Trainer on
    the same workspace resumes from the newest checkpoint, test() writes the frames."""
    from argparse import Namespace
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.utils import FusedAdam, PSNRMeter, Trainer, get_rays
    opt = Namespace(lr=1e-2, iters=64, update_extra_interval=4, dt_gamma=0, max_steps=256, cuda_ray=True, fp16=True,
                    workspace=str(tmp_path), seed=0, num_rays=256, some_cli_flag_the_renderer_ignores=True)
    poses, intr, H, W = room.cameras(n=4, H=32, W=32, focal=16.0)

    def view(i, n=None):
        r = get_rays(_t(poses[i:i + 1]), intr, H, W, N=n if n else -1)
        rgb, _, _ = room.trace(r["rays_o"][0].cpu().numpy(), r["rays_d"][0].cpu().numpy())
        d = {"rays_o": r["rays_o"], "rays_d": r["rays_d"], "H": H, "W": W}
        d["images"] = _t(rgb)[None] if n else _t(rgb).view(1, H, W, 3)
        return d
    train_loader = [view(i % 3, 256) for i in range(6)]
    valid_loader = [view(3)]

    def build():
        torch.manual_seed(0)
        model = NeRFNetwork(encoding="hashgrid", bound=1, cuda_ray=True, density_scale=1, min_near=0.05, density_thresh=10,
                            bg_radius=-1)
        criterion = torch.nn.MSELoss(reduction="none")
        if opt_kind == "torch_adam":
            optimizer = lambda model: torch.optim.Adam(model.get_params(opt.lr), betas=(0.9, 0.99), eps=1e-15)
        else:
            optimizer = lambda model: FusedAdam(model.get_params(opt.lr), betas=(0.9, 0.99), eps=1e-15)
        scheduler = lambda optimizer: torch.optim.lr_scheduler.LambdaLR(optimizer, lambda it: 0.1 ** min(it / opt.iters, 1))
        return Trainer("ngp", opt, model, device=torch.device(DEV), workspace=opt.workspace, optimizer=optimizer,
                       criterion=criterion, ema_decay=0.95, fp16=opt.fp16, lr_scheduler=scheduler,
                       scheduler_update_every_step=True, metrics=[PSNRMeter()], use_checkpoint="latest", eval_interval=1,
                       mute=True)
    tr = build()
    assert tr.update_extra_interval == 4 and tr.epoch == 0
    tr.train(train_loader, valid_loader, 3)
    assert tr.epoch == 3 and tr.global_step == 18 and len(tr.stats["loss"]) == 3 and len(tr.stats["valid_loss"]) == 3
    assert tr.stats["results"] == tr.stats["valid_loss"]                       # use_loss_as_metric (upstream's default)
    assert abs(tr.optimizer.param_groups[0]["lr"] - 1e-2 * 0.1 ** (18 / 64)) < 1e-9
    ck = sorted(os.listdir(tmp_path / "checkpoints"))
    assert ck == ["ngp.pth", "ngp_ep0002.pth", "ngp_ep0003.pth"]              # two rotating files + the best one
    assert tr.stats["best_result"] == min(tr.stats["results"])
    assert tr.model.mean_count > 0 and tr.stats["loss"][-1] < tr.stats["loss"][0]
    again = build()                                                            # 'latest': resumes where the first stopped
    assert again.epoch == 3 and again.global_step == 18
    assert torch.equal(again.model.encoder.embeddings, tr.model.encoder.embeddings)
    assert abs(again.optimizer.param_groups[0]["lr"] - tr.optimizer.param_groups[0]["lr"]) < 1e-12
    torch.manual_seed(5)                                                       # the step jitters its ray starts
    la = float(tr.train_one_step(train_loader[0]))
    torch.manual_seed(5)
    lb = float(again.train_one_step(train_loader[0]))
    assert abs(la - lb) < 1e-6 * max(1.0, abs(la))
    files = again.test([view(3)], save_path=str(tmp_path / "results"))
    assert len(files) == 1 and os.path.exists(files[0]) and os.path.exists(files[0].replace("_rgb", "_depth"))


def test_trainer_runs_on_a_transforms_json_scene(tmp_path, room):
    """Data on disk in the reference's NeRF-stage format -> NeRFDataset -> both Trainer stages (a few steps each):
    the loader, the HIP ray generator and the fused training kernels fit together; losses are finite and fall."""
    import json
    from PIL import Image
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import NeRFDataset
    from instance_nerf_amd.nerf.utils import Trainer
    from oracle import rays as orays
    H = W = 48
    poses, intr, _, _ = room.cameras(n=4, H=H, W=W, focal=W / 2.0)
    os.makedirs(tmp_path / "images"), os.makedirs(tmp_path / "masks")
    frames = []
    for i, P in enumerate(poses):
        r = orays.get_rays(P[None], intr, H, W)
        rgb, ids, _ = room.trace(r["rays_o"][0], r["rays_d"][0])
        Image.fromarray((rgb.reshape(H, W, 3) * 255).astype(np.uint8)).save(tmp_path / "images" / f"v{i}.png")
        np.save(tmp_path / "masks" / f"v{i}.npy", (ids % 8).reshape(H, W).astype(np.int32))
        # invert nerf_matrix_to_ngp (scale 1, no offset): the file stores the Blender-convention matrix
        T = np.eye(4, dtype=np.float32)
        T[[1, 2, 0], 0], T[[1, 2, 0], 1], T[[1, 2, 0], 2], T[[1, 2, 0], 3] = P[:3, 0], -P[:3, 1], -P[:3, 2], P[:3, 3]
        frames.append({"file_path": f"images/v{i}.png", "transform_matrix": T.tolist()})
    with open(tmp_path / "transforms_train.json", "w") as f:
        json.dump({"fl_x": W / 2.0, "fl_y": W / 2.0, "cx": W / 2.0, "cy": H / 2.0, "w": W, "h": H, "frames": frames}, f)
    ds = NeRFDataset(str(tmp_path), type="train", device=DEV, scale=1.0, num_rays=1024, mask_dir=str(tmp_path / "masks"),
                     num_instances=16)
    assert torch.allclose(ds.poses.cpu(), torch.from_numpy(poses), atol=1e-6)
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, num_instances=16).to(DEV)
    net.density_bitfield.copy_(_t(room.density_bitfield(128, 1.0)))
    tr = Trainer("json", None, net, stage="nerf", device=torch.device(DEV), iters=200, update_extra_interval=10 ** 9)
    tr.global_step = 1
    losses = [float(tr.train_one_step(ds[i % len(ds)])) for i in range(30)]
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5])
    ti = Trainer("json", None, net, stage="instance", device=torch.device(DEV), iters=200, update_extra_interval=10 ** 9)
    ti.global_step = 1
    ce = [float(ti.train_one_step(ds[i % len(ds)])) for i in range(30)]
    assert np.isfinite(ce).all() and np.mean(ce[-5:]) < np.mean(ce[:5])


@pytest.mark.parametrize("fx_grad", [False, True, 64], indirect=True)
@pytest.mark.parametrize("stage", ["nerf", "instance"])
def test_captured_training_step_equals_eager(stage, fx_grad):
    """Trainer(use_graph=True): the steady-state step captured once as a hipGraph (march, fields, compositing, loss,
    backward, Adam with its step-dependent scalars in device memory) and replayed follows the eager trainer step for
    step - same ray jitter, same learning-rate schedule, same bias correction - and survives an occupancy update."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    from instance_nerf_amd.nerf.utils import Trainer
    dev = torch.device(DEV)
    runs = {}
    for use_graph in (False, True):
        torch.manual_seed(0)
        net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, num_instances=16 if stage == "instance" else 0).to(dev)
        ds = SyntheticRoomDataset(dev, num_rays=1024, num_instances=16, seed=5)
        net.density_bitfield.copy_(_t(ds.room.density_bitfield(128, 1.0)))
        tr = Trainer("g", None, net, stage=stage, device=dev, iters=200, update_extra_interval=10 ** 9, use_graph=use_graph,
                     ema_decay=0.95)
        tr.global_step = 1
        batches = [ds.batch() for _ in range(3)]
        losses = [float(tr.train_one_step(batches[i % 3])) for i in range(3)]          # eager: exact sample totals
        net.mean_count = 65536                                                        # steady state from here on
        losses += [float(tr.train_one_step(batches[i % 3])) for i in range(6)]
        assert (tr._graph is not None) == use_graph
        net.mean_count = 81920                                                        # buffer size changes: re-capture
        losses += [float(tr.train_one_step(batches[i % 3])) for i in range(4)]
        runs[use_graph] = (losses, [p.detach().clone() for g in tr.optimizer.param_groups for p in g["params"]],
                           tr.optimizer.step_count, [s.clone() for s in tr.ema.shadow], net.local_step)
    a, b = runs[False], runs[True]
    assert a[2] == b[2] == 13 and a[4] == b[4]
    assert np.allclose(a[0], b[0], rtol=2e-3), (a[0], b[0])
    assert a[0][-1] < a[0][0]
    if fx_grad:                 # int32 table-gradient sums: the replayed graph and the eager step are the same bits
        assert a[0] == b[0], (a[0], b[0])
        for p, q in zip(a[1] + a[3], b[1] + b[3]):
            assert torch.equal(p, q)
        return
    for p, q in zip(a[1], b[1]):
        assert torch.linalg.norm(p - q) < 1e-2 * torch.linalg.norm(p)
    for p, q in zip(a[3], b[3]):
        assert torch.linalg.norm(p - q) < 1e-2 * torch.linalg.norm(p)


@pytest.mark.parametrize("fx_grad", [False, True, 64], indirect=True)
@pytest.mark.parametrize("stage", ["instance", "instance+shade", "nerf"])
def test_pipelined_captured_step_equals_eager(stage, fx_grad):
    """Trainer(use_graph=True, look_ahead=True): ONE hipGraph per step holds the step and, forked off before the
    table-gradient scatter, the parameter-independent head of the NEXT batch on a second stream - ray/box test and
    march, and in the instance stage the frozen NeRF's forward and the weight compositing.  Against the eager trainer
    on the same batches, with occupancy updates every 4 steps (the first step after one computes its own head, the last
    one before it does not look ahead): the same sample total at every step - the same rays, the same jitter - losses
    that agree as two eager runs do, parameters and their EMA as close as the scatter's summation order allows."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    from instance_nerf_amd.nerf.utils import Trainer
    dev = torch.device(DEV)
    runs = {}
    stage, shade = stage.split("+")[0], stage.endswith("+shade")
    for piped in (False, True):
        torch.manual_seed(0)
        net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, num_instances=16 if stage == "instance" else 0).to(dev)
        ds = SyntheticRoomDataset(dev, num_rays=1024, num_instances=16, seed=5)
        net.density_bitfield.copy_(_t(ds.room.density_bitfield(128, 1.0)))
        analytic = net.density_bitfield.clone()
        real = net.update_extra_state

        def update(*a, real=real, net=net, analytic=analytic, **kw):
            real(*a, **kw)                           # the update runs (mean_count!) but the analytic grid is kept,
            net.density_bitfield.copy_(analytic)     # and both runs size their buffers like the captured one does
            if net.mean_count > 0:
                net.mean_count = (net.mean_count + 16383) // 16384 * 16384
        net.update_extra_state = update
        tr = Trainer("p", None, net, stage=stage, device=dev, iters=200, update_extra_interval=4, use_graph=piped,
                     look_ahead=piped, shade_ahead=shade, ema_decay=0.95, workspace=None, mute=True)
        tr.global_step = 1
        batches = [ds.batch() for _ in range(19)]
        torch.manual_seed(11)
        losses, totals = [], []
        for i in range(18):
            losses.append(float(tr.train_one_step(batches[i], batches[i + 1])) if piped
                          else float(tr.train_one_step(batches[i])))
            totals.append(int(net.last_counter[0]))
        kinds = sorted(k[1:] for k in tr._pipe["graphs"]) if piped else None
        runs[piped] = (losses, totals, [p.detach().clone() for g in tr.optimizer.param_groups for p in g["params"]],
                       [s.clone() for s in tr.ema.shadow], tr.optimizer.step_count, net.local_step, kinds)
    a, b = runs[False], runs[True]
    # every kind of step was captured and replayed: own head + look-ahead, prefetched head + look-ahead, prefetched head only
    assert (True, True) in b[6] and (False, True) in b[6] and (False, False) in b[6], b[6]
    assert a[1] == b[1], (a[1], b[1])
    assert a[4] == b[4] == 18 and a[5] == b[5]
    assert np.allclose(a[0], b[0], rtol=2e-3), (a[0], b[0])
    if fx_grad:
        # with the table gradient summed as int32 (opt-in, round 6) nothing in a step depends on the order of arrival any
        # more: the captured two-stream pipeline and the eager loop are the SAME computation - every loss, every parameter
        # and every EMA shadow bit for bit
        assert a[0] == b[0], (a[0], b[0])
        for p, q in zip(a[2] + a[3], b[2] + b[3]):
            assert torch.equal(p, q)
        return
    for p, q in zip(a[2], b[2]):
        assert torch.linalg.norm(p - q) < 1e-2 * torch.linalg.norm(p)
    for p, q in zip(a[3], b[3]):
        assert torch.linalg.norm(p - q) < 1e-2 * torch.linalg.norm(p)


@pytest.mark.parametrize("mode", ["eager", "look_ahead", "pipelined"])
def test_ignored_rays_are_never_marched_in_the_instance_stage(mode):
    """Trainer(stage="instance", prune_ignored=True) (round-4 verdict item 2a): rays whose matched-mask label is -1
    carry no loss and no gradient, so the ray/box test reports them as misses (inr_near_far_from_aabb_skip) and the
    march gives them no samples.  30 % of the rays ignored: the first step's loss is the SAME number with and without
    pruning (same parameters; the per-ray cross-entropy sum runs over the same rays in the same order), its gradients
    agree to summation order (the surviving samples sit in other 16-sample tiles), 30 % fewer samples are marched, and
    over a run - eager, with the look-ahead march, as the captured two-stream pipeline - the losses stay together as
    two plain runs do."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    from instance_nerf_amd.nerf.utils import Trainer
    dev = torch.device(DEV)
    piped, ahead = mode == "pipelined", mode != "eager"
    runs = {}
    for prune in (False, True):
        torch.manual_seed(0)
        net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, num_instances=16).to(dev)
        ds = SyntheticRoomDataset(dev, num_rays=2048, num_instances=16, seed=5, ignore_frac=0.3)
        net.density_bitfield.copy_(_t(ds.room.density_bitfield(128, 1.0)))
        tr = Trainer("pr", None, net, stage="instance", device=dev, iters=200, update_extra_interval=1000,
                     use_graph=piped, look_ahead=ahead, prune_ignored=prune, workspace=None, mute=True)
        batches = [ds.batch() for _ in range(14)]
        first = None
        if mode == "eager":
            torch.manual_seed(7)
            pred, _, loss = tr.train_step(batches[0])
            tr.optimizer.zero_grad()
            loss.backward()
            first = (float(loss), int(net.last_counter[0]),
                     [p.grad.detach().clone() for g in tr.optimizer.param_groups for p in g["params"]], pred.detach().clone())
            tr.optimizer.zero_grad()
        # a steady state for the look-ahead / captured steps: buffers sized for the unpruned batch in both runs
        net.mean_count = 2048 * 64
        tr.global_step = 1
        torch.manual_seed(11)
        losses, totals = [], []
        for i in range(12):
            losses.append(float(tr.train_one_step(batches[i], batches[i + 1] if ahead else None)))
            totals.append(int(net.last_counter[0]))
        runs[prune] = (first, losses, totals, [b["masks"] for b in batches])
    a, b = runs[False], runs[True]
    for x, y, m in zip(a[2], b[2], a[3]):
        kept = float((m >= 0).float().mean())
        assert 0.6 < kept < 0.8
        assert abs(y / x - kept) < 0.03, (x, y, kept)           # samples marched fall with the ignored fraction
    assert np.allclose(a[1], b[1], rtol=2e-3), (a[1], b[1])
    if mode == "eager":
        (la, ta, ga, pa), (lb, tb, gb, pb) = a[0], b[0]
        assert la == lb and tb < 0.8 * ta
        for x, y in zip(ga, gb):
            assert float(torch.linalg.norm(x - y)) <= 1e-5 * float(torch.linalg.norm(x)), (x.shape,)
        keep = (a[3][0] >= 0).reshape(-1)
        assert torch.equal(pa.reshape(-1, pa.shape[-1])[keep], pb.reshape(-1, pb.shape[-1])[keep])
        assert float(pb.reshape(-1, pb.shape[-1])[~keep].abs().max()) == 0.0


def test_shade_ahead_is_bit_identical_to_the_inline_head(room):
    """march_ahead(shade=True): ray/box test, march, frozen NeRF forward and compositing forward queued on a side stream
    into persistent buffers give the render the same bits as computing them in the step (eager, no graph): loss and every
    gradient of the instance stage identical."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    dev = torch.device(DEV)
    torch.manual_seed(1)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, num_instances=16).to(dev).train()
    net.freeze_nerf()
    ds = SyntheticRoomDataset(dev, num_rays=1024, num_instances=16, seed=5)
    net.density_bitfield.copy_(_t(ds.room.density_bitfield(128, 1.0)))
    b = ds.batch()
    net.mean_count = 65536
    assert net.shade_ahead_applies()
    out = {}
    for ahead in (False, True):
        for q in net.parameters():
            q.grad = None
        torch.manual_seed(4)
        kw = dict(staged=False, bg_color=1, perturb=True, force_all_rays=False, ce_labels=b["masks"])
        if ahead:
            side = torch.cuda.Stream()
            marched = net.march_ahead(b["rays_o"], b["rays_d"], perturb=True, stream=side, shade=True)
            assert marched is not None and marched["shaded"] is not None
            kw["marched"] = marched
        r = net.render(b["rays_o"], b["rays_d"], **kw)
        r["instance_ce"].backward()
        torch.cuda.synchronize()
        out[ahead] = (float(r["instance_ce"]), r["instance"].detach().clone(), r["image"].detach().clone(),
                      int(net.last_counter[0]),
                      {k: v.grad.clone() for k, v in net.named_parameters() if v.grad is not None})
    x, y = out[False], out[True]
    assert x[3] == y[3] and x[0] == y[0]
    assert torch.equal(x[1], y[1]) and torch.equal(x[2], y[2])
    for k in x[4]:
        if "embeddings" in k:        # the scatter's atomics sum in arrival order
            assert torch.linalg.norm(x[4][k] - y[4][k]) <= 1e-5 * torch.linalg.norm(x[4][k]), k
        else:
            assert torch.equal(x[4][k], y[4][k]), k


def test_train_time_O_keeps_the_instance_stage_within_one_percent(room):
    """Trainer(fp16=True, stage="instance") - upstream's -O for the stage this repository trains with a frozen NeRF: the
    frozen field's forward runs with -O's numerics (fp16 copy of its table: half the bytes every XCD pulls through its
    fabric port; single-pass fp16 MLP: inr_nerf_forward_fast) while everything that is TRAINED - instance table, instance
    MLP, their gradients, Adam's moments - stays fp32 (no GradScaler: nothing trained is ever stored in half precision).
    50 steps from the same parameters on the same batches: the loss stays within 1 % of the fp32 run at every step."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    from instance_nerf_amd.nerf.utils import Trainer
    dev = torch.device(DEV)
    runs = {}
    for fp16 in (False, True):
        torch.manual_seed(0)
        net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, num_instances=16).to(dev)
        with torch.no_grad():                      # an O(1) NeRF (semi-transparent rays): its rounding matters
            net.encoder.embeddings.uniform_(-1, 1, generator=torch.Generator(device=DEV).manual_seed(9))
        ds = SyntheticRoomDataset(dev, num_rays=1024, num_instances=16, seed=5)
        net.density_bitfield.copy_(_t(ds.room.density_bitfield(128, 1.0)))
        tr = Trainer("o", None, net, stage="instance", device=dev, iters=200, update_extra_interval=10 ** 9, fp16=fp16,
                     workspace=None, mute=True)
        tr.global_step = 1
        assert net.half_table == fp16 and net.mlp_fp16 == fp16
        calls = []
        lib_fast = net._fused_nerf

        def spy(*a, _f=lib_fast, **kw):
            calls.append(1)
            return _f(*a, **kw)
        net._fused_nerf = spy
        orig = net.render
        net.render = lambda *a, **kw: orig(*a, **{**kw, "perturb": False, "force_all_rays": True})
        batches = [ds.batch() for _ in range(5)]
        losses = [float(tr.train_one_step(batches[i % 5])) for i in range(50)]
        assert len(calls) == 50                                     # the frozen field ran on the fused no-grad kernel
        assert all(p.dtype == torch.float32 for p in net.parameters())
        st = tr.optimizer.state[net.instance_encoder.embeddings]
        assert st["exp_avg"].dtype == torch.float32 and st["exp_avg_sq"].dtype == torch.float32
        runs[fp16] = losses
    a, b = np.asarray(runs[False]), np.asarray(runs[True])
    assert a[-5:].mean() < 0.5 * a[:5].mean()
    assert np.abs(a - b).max() <= 0.01 * np.abs(a).max() and (np.abs(a - b) / np.abs(a)).max() < 0.01, (a[-5:], b[-5:])
    assert np.abs(a - b).max() > 0                                  # it really ran other numerics


def test_render_through_the_registered_custom_ops(params_k16, room, room_bitfield, level_table):
    """torch.ops.inr.*: a training-mode render assembled from the registered ops - ray/box test, march, hash-grid
    encode + tiny MLPs (torch), compositing - equals the module path (NeRFNetwork.render) on the same rays: sample
    counts exactly, image within fp32 rounding of the unfused arithmetic, and gradients flow to the table through the
    ops' registered autograd formulas; the fused no-grad op gives the module's sigma / rgb bit for bit."""
    from instance_nerf_amd import ops  # noqa: F401  (registers torch.ops.inr)
    net = _network({k: v.clone() for k, v in params_k16.items()}, K=0).train()
    net.density_bitfield.copy_(_t(room_bitfield))
    ro, rd = scene_rays(room, 300, seed=71)
    ro_t, rd_t = _t(ro), _t(rd)
    targs = ops.table_args(net.encoder.table)
    nears, fars = torch.ops.inr.near_far_from_aabb(ro_t, rd_t, net.aabb_train, 0.05)
    xyzs, dirs, deltas, rays, counter = torch.ops.inr.march_rays_train(ro_t, rd_t, 1.0, net.density_bitfield, 1, 128, nears,
                                                                        fars, None, 0.0, 1024, -1)
    emb = net.encoder.embeddings
    enc = torch.ops.inr.grid_encode(xyzs, emb, 1.0, *targs)
    h = torch.relu(enc @ net.sigma_net[0].weight.t()) @ net.sigma_net[1].weight.t()
    sigma = torch.exp(h[:, 0])
    cin = torch.cat([net.encoder_dir(dirs), h[:, 1:]], -1)
    c = torch.relu(torch.relu(cin @ net.color_net[0].weight.t()) @ net.color_net[1].weight.t()) @ net.color_net[2].weight.t()
    rgb = torch.sigmoid(c)
    ws, depth, image = torch.ops.inr.composite_rays_train(sigma, rgb, deltas, rays, 1e-4)
    image = image + (1 - ws)[:, None]
    ref = net.render(ro_t[None], rd_t[None], bg_color=1, perturb=False, force_all_rays=True)
    assert int(counter[0]) == int(ref["num_samples"][0]) == xyzs.shape[0] > 5000
    assert (image - ref["image"][0]).abs().max() < 2e-4
    image.square().mean().backward()
    g_ops = emb.grad.clone()
    emb.grad = None
    ref["image"].square().mean().backward()
    assert g_ops.abs().sum() > 0
    assert float(torch.linalg.norm(g_ops - emb.grad) / torch.linalg.norm(emb.grad)) < 2e-3
    with torch.no_grad():
        s_mod, c_mod = net(xyzs, dirs)
        s_op, c_op = torch.ops.inr.nerf_forward(xyzs, dirs, emb, net.sigma_net[0].weight, net.sigma_net[1].weight,
                                                net.color_net[0].weight, net.color_net[1].weight, net.color_net[2].weight,
                                                1.0, *targs)
    assert torch.equal(s_mod, s_op) and torch.equal(c_mod, c_op)


def test_pipelined_step_survives_many_occupancy_updates_and_late_captures():
    """The captured pipeline over 70 steps with an occupancy update every 5: graph kinds that are first needed long after
    the buffer sets' prefetch records were made (a late capture must not be refused by the renderer's staleness check),
    a caller's own update between two steps (the prefetched head is then recomputed), and sample totals that stay equal
    to the eager trainer's throughout."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    from instance_nerf_amd.nerf.utils import Trainer
    dev = torch.device(DEV)
    runs = {}
    for piped in (False, True):
        torch.manual_seed(0)
        net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, num_instances=16).to(dev)
        ds = SyntheticRoomDataset(dev, num_rays=512, num_instances=16, seed=7)
        net.density_bitfield.copy_(_t(ds.room.density_bitfield(128, 1.0)))
        analytic = net.density_bitfield.clone()
        real = net.update_extra_state

        def update(*a, real=real, net=net, analytic=analytic, **kw):
            real(*a, **kw)
            net.density_bitfield.copy_(analytic)
            if net.mean_count > 0:
                net.mean_count = (net.mean_count + 16383) // 16384 * 16384
        net.update_extra_state = update
        tr = Trainer("p2", None, net, stage="instance", device=dev, iters=500, update_extra_interval=5, use_graph=piped,
                     look_ahead=piped, ema_decay=0.95, workspace=None, mute=True)
        tr.global_step = 1
        batches = [ds.batch() for _ in range(71)]
        torch.manual_seed(3)
        totals, losses = [], []
        for i in range(70):
            if i == 33:
                net.update_extra_state()             # a caller's own update, off the trainer's schedule
            losses.append(float(tr.train_one_step(batches[i], batches[i + 1] if piped else None)))
            totals.append(int(net.last_counter[0]))
        runs[piped] = (totals, losses)
    # up to the caller's update the two runs draw the same jitter: identical sample totals; the update makes the pipeline
    # throw away a head it had already drawn jitter for (one draw more than the eager run), so from there on the rays are
    # jittered differently - the same training, not the same numbers
    assert runs[False][0][:33] == runs[True][0][:33]
    assert np.allclose(runs[False][1][:33], runs[True][1][:33], rtol=5e-3)
    a, b = np.asarray(runs[False][0][33:], float), np.asarray(runs[True][0][33:], float)
    assert np.abs(a - b).max() < 0.02 * a.max()
    assert np.isfinite(runs[True][1]).all() and np.allclose(runs[False][1][33:], runs[True][1][33:], rtol=0.1, atol=0.02)


# ---------------------------------------------------------------------------------------- fused loader launch (round 6)
@pytest.mark.parametrize("channels,with_mask", [(3, True), (4, False), (3, False)])
def test_fused_loader_launch_is_bit_exact_against_the_oracle(channels, with_mask):
    """inr_sample_training_batch (one launch: pixel draw, rays, rgb gather, label gather) against
    oracle/rays.py::sample_training_batch: indices, rays, colours and labels bit for bit; and its rays are exactly
    inr_get_rays' for the pixels it drew."""
    from instance_nerf_amd import _lib
    from instance_nerf_amd.nerf.utils import get_rays
    from oracle import rays as orays
    lib = _lib.load()
    rng = np.random.default_rng(9)
    H, W, K, n = 37, 53, 6, 5000
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
    pose[:3, 3] = rng.normal(size=3).astype(np.float32)
    intr = (41.5, 40.25, 26.5, 18.5)
    img = rng.random((H, W, channels)).astype(np.float32)
    mask = rng.integers(-1, 10, size=(H, W)).astype(np.int32) if with_mask else None
    for seed, step in ((0, 0), (123456789, 77), (2 ** 40 + 5, 2 ** 31 - 1)):
        ref = orays.sample_training_batch(pose, intr, H, W, img, mask, K, seed, step, n)
        P, I = _t(pose), _t(img)
        Mk = _t(mask) if with_mask else None
        inds = torch.empty(n, dtype=torch.int64, device=DEV)
        ro, rd = torch.empty(n, 3, device=DEV), torch.empty(n, 3, device=DEV)
        rgb = torch.empty(n, channels, device=DEV)
        lab = torch.empty(n, dtype=torch.int64, device=DEV) if with_mask else None
        _lib.check(lib.inr_sample_training_batch(_lib.ptr(P), *intr, H, W, _lib.ptr(I), channels,
                                                 _lib.ptr(Mk) if with_mask else None, K, seed, step, n, _lib.ptr(inds),
                                                 _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(rgb),
                                                 _lib.ptr(lab) if with_mask else None, _lib.stream_ptr()))
        assert (inds.cpu().numpy() == ref["inds"]).all()
        assert (rd.cpu().numpy() == ref["rays_d"]).all() and (ro.cpu().numpy() == ref["rays_o"]).all()
        assert (rgb.cpu().numpy() == ref["rgb"]).all()
        if with_mask:
            assert (lab.cpu().numpy() == ref["labels"]).all()
        g = get_rays(P[None], intr, H, W, inds=inds)
        assert torch.equal(g["rays_d"][0], rd) and torch.equal(g["rays_o"][0], ro)
    # empty batch: nothing launched, nothing touched; bad arguments come back as codes
    assert lib.inr_sample_training_batch(None, *intr, H, W, None, 3, None, 0, 0, 0, 0, None, None, None, None, None, None) == 0
    assert lib.inr_sample_training_batch(_lib.ptr(P), *intr, H, W, _lib.ptr(I), 5, None, 0, 0, 0, n, _lib.ptr(inds), _lib.ptr(ro),
                                         _lib.ptr(rd), _lib.ptr(rgb), None, None) == -1


def test_nerf_dataset_batches_come_from_the_fused_launch_and_match_the_tensor_op_loader(tmp_path, room):
    """NeRFDataset on the GPU hands out training batches from the fused launch: same keys, shapes and dtypes as the
    tensor-op loader, the colours and labels of the pixels it drew (checked through the batch's own rays: the ray of a
    pixel identifies it), reproducible from the seed, and different from batch to batch."""
    from instance_nerf_amd.nerf.provider import NeRFDataset
    from instance_nerf_amd.nerf.utils import get_rays
    from oracle import rays as orays
    scene = room.write_dataset(str(tmp_path / "s"), n_views=3, H=40, W=48, num_instances=8, ignore_frac=0.2)
    kw = dict(type="train", device=DEV, scale=1.0, num_rays=512, mask_dir=scene["mask_dir"], num_instances=8, seed=11)
    ds = NeRFDataset(scene["path"], **kw)
    b0, b1 = ds[1], ds[1]
    assert set(b0) == {"H", "W", "rays_o", "rays_d", "index", "images", "masks"}
    assert b0["rays_o"].shape == (1, 512, 3) and b0["images"].shape == (1, 512, 3) and b0["masks"].shape == (1, 512)
    assert b0["masks"].dtype == torch.int64 and b0["images"].dtype == torch.float32 and b0["index"] == [1]
    assert not torch.equal(b0["rays_d"], b1["rays_d"])                     # the batch counter advances
    again = NeRFDataset(scene["path"], **kw)[1]
    assert torch.equal(again["rays_d"], b0["rays_d"]) and torch.equal(again["masks"], b0["masks"])
    # against the restatement, from the files' content
    ref = orays.sample_training_batch(ds.poses[1].cpu().numpy(), ds.intrinsics, 40, 48, ds.images[1].cpu().numpy(),
                                      ds.masks[1].cpu().numpy(), 8, 11, 0, 512)
    assert (b0["rays_d"][0].cpu().numpy() == ref["rays_d"]).all() and (b0["images"][0].cpu().numpy() == ref["rgb"]).all()
    assert (b0["masks"][0].cpu().numpy() == ref["labels"]).all() and (ref["labels"] == -1).any()
    # the tensor-op loader (fused_batches = False) keeps working and has the same layout
    ds.fused_batches = False
    t = ds[1]
    assert {k: (tuple(v.shape), v.dtype) for k, v in t.items() if torch.is_tensor(v)} == \
        {k: (tuple(v.shape), v.dtype) for k, v in b0.items() if torch.is_tensor(v)}
    g = get_rays(ds.poses[1:2], ds.intrinsics, 40, 48, inds=torch.from_numpy(ref["inds"]).to(DEV))
    assert torch.equal(g["rays_d"], b0["rays_d"])


def test_launch_failure_is_a_return_code():
    """"returns a negative code, never exit()" (the habit avoided:
    /root/reference/nerf_rcnn/model/rotated_iou/cuda_op/cuda_utils.h:26-35 exit()s on a CUDA error).  A launch the
    runtime refuses - the field kernel asked for 1 MiB of LDS per workgroup through inr_set_overlap_placement, a CU has
    160 KB - comes back from THAT call as INR_ELAUNCH with the runtime's message; the process lives, the switch can be
    reset, and the next frame renders the same pixels as before.  In a child process: a broken library must fail this
    test, not end the session."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from instance_nerf_amd import _lib
from instance_nerf_amd.nerf import NeRFNetwork
from instance_nerf_amd.nerf.utils import get_rays
from instance_nerf_amd.scene import RoomScene
lib = _lib.load()
dev = "cuda:0"
torch.manual_seed(0)
room = RoomScene()
net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10).to(dev).eval()
net.density_bitfield.copy_(torch.from_numpy(room.density_bitfield(128, 1.0)).to(dev))
with torch.no_grad():
    net.encoder.embeddings.uniform_(-1.0, 1.0)
poses, intr, H, W = room.cameras(H=64, W=64, focal=32.0)
r = get_rays(torch.from_numpy(poses[:1]).to(dev), intr, H, W, patch=4)
render = lambda: net.render(r["rays_o"], r["rays_d"], bg_color=1, infer_mode="fused")["image"]
with torch.no_grad():
    ref = render().clone()
    assert lib.inr_set_overlap_placement(-5) == -1 and lib.inr_set_overlap_placement(7) == -1
    assert lib.inr_set_overlap_placement(1 << 20) == 0
    try:
        render()
        raise SystemExit("the over-sized launch was accepted")
    except RuntimeError as e:
        msg = str(e)
    assert "code -2" in msg and "nerf_forward_table" in msg, msg            # INR_ELAUNCH + the runtime's text
    assert lib.inr_set_overlap_placement(0) == 0
    torch.cuda.synchronize()
    again = render()
    assert torch.equal(again, ref)
    assert lib.inr_set_overlap_placement(64 * 1024) == 0                    # a legal explicit size: same pixels
    assert torch.equal(render(), ref)
    lib.inr_set_overlap_placement(0)
print("alive")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "alive" in res.stdout, (res.returncode, res.stdout[-500:], res.stderr[-1500:])


# ---------------------------------------------------------------------------------------- fixed-point scatter (round 6)
def _fx_step(lib, x, go, desc, L, bound, T, fx, ranges=None):
    """One training step's worth of calls: scatter (per level range), finishing pass, scale update -> fp32 gradient."""
    from instance_nerf_amd import _lib
    g = torch.zeros(T, 2, device=DEV)
    for lo, hi in (ranges or ((0, L),)):
        _lib.check(lib.inr_grid_encode_backward_levels_fx(_lib.ptr(x), _lib.ptr(go), None, desc, x.shape[0], float(bound),
                                                          _lib.ptr(g), lo, hi, _lib.ptr(fx), _lib.stream_ptr()))
        _lib.check(lib.inr_grid_grad_finish_fx(_lib.ptr(g), desc, lo, hi, _lib.ptr(fx), _lib.stream_ptr()))
    _lib.check(lib.inr_grid_fx_update(_lib.ptr(fx), L, 128.0, 32, _lib.stream_ptr()))
    return g


@pytest.mark.parametrize("bound", [1.0, 4.0])
def test_fixed_point_table_gradient(level_table, bound):
    """The int32 form of the table-gradient scatter (include/inr.h, round 6) against the oracle's scatter-add and
    against the fp32 atomics it replaces: the first step has no scales (fp32 atomics, bit pattern of rounds 1-5), the
    second runs on int32 sums - same gradient within the fp32 path's own rounding, every value a multiple of its
    level's quantum, and the SAME BITS when the step is repeated (integer sums do not depend on the order of arrival;
    fp32 atomics do).  A 100x jump of the gradient is a near miss (more than 1/8 of the range used): still exact,
    counted, peak use recorded; a non-finite gradient poisons and resets its level."""
    from instance_nerf_amd import _lib
    from instance_nerf_amd.gridencoder import GridEncoder
    from oracle import hashgrid
    lib = _lib.load()
    tb = level_table if bound == 1.0 else hashgrid.level_table(desired_resolution=8192)
    enc = GridEncoder(desired_resolution=int(tb["resolutions"][-1])).to(DEV)
    L, T, desc = 16, int(tb["total_rows"]), enc.desc
    gen = torch.Generator().manual_seed(21)
    M = 60000
    # ray-like samples (runs of neighbouring points: the in-wave run merge and the coarse levels' long chains are in play)
    o = (torch.rand(M // 50, 1, 3, generator=gen) * 1.6 - 0.8) * bound
    d = torch.nn.functional.normalize(torch.randn(M // 50, 1, 3, generator=gen), dim=-1)
    x = (o + d * torch.linspace(0, 0.5 * bound, 50).view(1, 50, 1)).reshape(-1, 3).clamp(-bound, bound).contiguous()
    go = (torch.randn(M, 32, generator=gen) * torch.logspace(-6, -2, M).view(-1, 1)[torch.randperm(M, generator=gen)]).contiguous()
    ref = hashgrid.encode_backward_table(x, go, bound, tb)
    xd, god = x.to(DEV), go.to(DEV)
    fx = torch.zeros(_lib.GRID_FX_STATE_FLOATS, device=DEV)
    g1 = _fx_step(lib, xd, god, desc, L, bound, T, fx)                        # no scales yet: fp32 atomics
    plain = torch.zeros(T, 2, device=DEV)
    _lib.check(lib.inr_grid_encode_backward_levels(_lib.ptr(xd), _lib.ptr(god), None, desc, M, float(bound), _lib.ptr(plain), 0, L,
                                                   _lib.stream_ptr()))
    nrm = float(ref.norm())
    assert float((g1.cpu() - ref).norm()) < 2e-6 * nrm and float((plain.cpu() - ref).norm()) < 2e-6 * nrm
    st = fx.cpu().numpy()
    scales = st[:16].copy()
    assert (scales > 0).all() and (np.log2(scales) == np.round(np.log2(scales))).all()            # powers of two
    offs = tb["offsets"]
    for l in range(16):                                       # 128 x the level's maximum fits: 2^22 < max * scale <= 2^23
        mx = float(ref[offs[l]:offs[l + 1]].abs().max())
        assert abs(st[32 + l] - mx) <= 1e-5 * mx and 2.0 ** 22 < mx * scales[l] <= 2.0 ** 23 * 1.001, (l, mx, scales[l])
    assert st[48] == 0 and st[49] == 0
    saved = fx.clone()
    g2 = _fx_step(lib, xd, god, desc, L, bound, T, fx)                        # int32 sums
    assert float((g2.cpu() - ref).norm()) < 1e-5 * nrm                        # north_star tolerance: 1e-3; measured 2.2e-6
    for l in (0, 5, 15):
        q = (g2[offs[l]:offs[l + 1]].double() * float(scales[l])).cpu()
        assert bool((q == q.round()).all())                                    # multiples of the level's quantum
    assert fx.cpu().numpy()[48] == 1
    fx.copy_(saved)
    g3 = _fx_step(lib, xd, god, desc, L, bound, T, fx, ranges=((8, L), (0, 8)))    # the two ranges of the N > 1 schedule
    assert torch.equal(g2, g3)                                                 # the SAME BITS
    # fp32 atomics, repeated: the same values up to rounding, rarely the same bits
    plain2 = torch.zeros(T, 2, device=DEV)
    _lib.check(lib.inr_grid_encode_backward_levels(_lib.ptr(xd), _lib.ptr(god), None, desc, M, float(bound), _lib.ptr(plain2), 0, L,
                                                   _lib.stream_ptr()))
    assert float((plain2 - plain).norm()) < 1e-6 * nrm
    # near miss: 100 x the gradient on the scales of the previous step - no wrap (128 x headroom), counted (more than 1/8
    # of the range used), the peak use recorded, and the next step's scales follow the new maximum (still int32 sums: no
    # step ever depends on the order of arrival)
    g4 = _fx_step(lib, xd, (god * 100).contiguous(), desc, L, bound, T, fx)
    assert float((g4.cpu() - 100 * ref).norm()) < 1e-5 * 100 * nrm
    st = fx.cpu().numpy()
    assert st[49] == 16 and (st[:16] > 0).all() and (st[:16] < scales / 32).all()
    assert (st[80:96] > 100 * 2.0 ** 22 / 2.0 ** 31 * 0.999).all() and (st[80:96] <= 100 * 2.0 ** 23 / 2.0 ** 31 * 1.001).all()
    g5 = _fx_step(lib, xd, (god * 100).contiguous(), desc, L, bound, T, fx)
    assert float((g5.cpu() - 100 * ref).norm()) < 1e-5 * 100 * nrm and fx.cpu().numpy()[49] == 16
    # a gradient that falls away: the reference decays by 3 % per step, the scales follow it up slowly
    for _ in range(24):
        _fx_step(lib, xd, god, desc, L, bound, T, fx)
    st2 = fx.cpu().numpy()
    assert (st2[16:32] < st[16:32] * 0.97 ** 24 * 1.001).all() and (st2[16:32] > st[16:32] * 0.97 ** 24 * 0.999).all()
    assert (st2[:16] >= st[:16] * 2).all()
    # a non-finite contribution has no int32 image: the level's whole gradient becomes NaN (loud), the level is reset
    bad = (god * 100).contiguous()
    bad[7, 3] = float("inf")                                                              # feature 3 = level 1
    gb = _fx_step(lib, xd, bad, desc, L, bound, T, fx)
    st = fx.cpu().numpy()
    assert st[1] == 0 and st[17] == 0 and (st[[0] + list(range(2, 16))] > 0).all() and st[64 + 1] == 0
    assert bool(torch.isnan(gb[offs[1]:offs[2]]).all()) and bool(torch.isfinite(gb[:offs[1]]).all()) and bool(torch.isfinite(gb[offs[2]:]).all())
    bad[7, 3] = float("nan")
    gb = _fx_step(lib, xd, bad, desc, L, bound, T, fx)                                   # level 1 on fp32 atomics now: NaN rows
    assert bool(torch.isnan(gb[offs[1]:offs[2]]).any()) and fx.cpu().numpy()[1] == 0
    # fx_state = NULL is the fp32 entry point
    g6 = torch.zeros(T, 2, device=DEV)
    _lib.check(lib.inr_grid_encode_backward_levels_fx(_lib.ptr(xd), _lib.ptr(god), None, desc, M, float(bound), _lib.ptr(g6), 0, L,
                                                      None, _lib.stream_ptr()))
    assert float((g6.cpu() - ref).norm()) < 2e-6 * nrm


@pytest.mark.parametrize("stage", ["nerf", "instance"])
def test_training_steps_are_bit_reproducible_with_the_fixed_point_scatter(stage, room):
    """Two trainers from the same seed, 40 steps each (occupancy updates, EMA, a seeded SyntheticRoomDataset): with the
    table gradient summed as int32 (opt-in: network.FX_GRAD / Trainer(fixed_point_grad=True)) every parameter ends with
    the SAME BITS.  With fp32 atomics (network.FX_GRAD = False) the two runs drift apart in the last bits of the table within
    a few steps - the order in which waves reach the memory-side atomic unit differs from launch to launch."""
    from instance_nerf_amd.nerf import NeRFNetwork, network
    from instance_nerf_amd.nerf.provider import SyntheticRoomDataset
    from instance_nerf_amd.nerf.utils import Trainer

    def run(fx):
        old_fx, network.FX_GRAD = network.FX_GRAD, fx
        try:
            torch.manual_seed(0)
            net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=16 if stage == "instance" else 0).to(DEV)
            net.density_bitfield.copy_(_t(room.density_bitfield(128, 1.0)))
            ds = SyntheticRoomDataset(torch.device(DEV), H=200, W=200, n_views=8, num_rays=2048, num_instances=16, seed=4)
            tr = Trainer("repro", None, net, stage=stage, device=torch.device(DEV), lr=1e-2, iters=200, workspace=None, mute=True,
                         ema_decay=0.95, update_extra_interval=16 if stage == "nerf" else 10 ** 9)
            tr.global_step = 0 if stage == "nerf" else 1
            losses = [float(tr.train_one_step(ds.batch())) for _ in range(40)]
            name = "encoder.embeddings" if stage == "nerf" else "instance_encoder.embeddings"
            return losses, {k: v.clone() for k, v in net.state_dict().items()}, name
        finally:
            network.FX_GRAD = old_fx
    for form in (True, 64):
        la, sa, name = run(form)
        lb, sb, _ = run(form)
        assert la == lb and la[-1] < la[0], form
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (form, k)
    lc, sc, _ = run(False)
    # the fp32 path trains the same way (losses within a few 1e-3 relative of the fixed-point run's) ...
    assert abs(lc[-1] - la[-1]) < 0.05 * abs(la[-1]) + 1e-6
    # ... and the two kinds of runs end close in the table (not the same bits: different rounding of every row sum)
    d = float((sa[name] - sc[name]).norm() / sc[name].norm())
    assert d < 0.05, d


def test_fixed_point_scale_update_matches_the_oracle_rule():
    """k_grid_fx_update against oracle/hashgrid.py::fx_next_scale on random references and step maxima (the block maxima of
    the finishing pass are written by hand): scales and references bit for bit."""
    from instance_nerf_amd import _lib
    from oracle import hashgrid
    lib = _lib.load()
    rng = np.random.default_rng(77)
    for trial in range(6):
        ref = (10.0 ** rng.uniform(-12, 2, 16)).astype(np.float32)
        ref[rng.integers(0, 16, 3)] = 0.0
        mx = (ref * 10.0 ** rng.uniform(-3, 1.5, 16)).astype(np.float32)
        mx[ref == 0] = (10.0 ** rng.uniform(-9, 0, int((ref == 0).sum()))).astype(np.float32)
        if trial == 5:
            mx[2], mx[9], mx[11] = np.inf, np.inf, 0.0            # (the finishing pass reports a NaN gradient as +inf)
        st = np.zeros(_lib.GRID_FX_STATE_FLOATS, np.float32)
        st[16:32] = ref
        st[:16] = 1.0                                                 # (an old scale: not an input of the rule)
        for l in range(16):
            slots = (mx[l] * rng.uniform(0, 1, 256)).astype(np.float32) if np.isfinite(mx[l]) else np.zeros(256, np.float32)
            slots[rng.integers(0, 256)] = mx[l]
            st[96 + 256 * l:96 + 256 * (l + 1)] = slots
        fx = _t(st)
        _lib.check(lib.inr_grid_fx_update(_lib.ptr(fx), 16, 128.0, 32, _lib.stream_ptr()))
        got = fx.cpu().numpy()
        s_ref, r_ref = hashgrid.fx_next_scale(ref, mx)
        assert (got[:16] == s_ref).all(), (trial, got[:16], s_ref)
        assert (got[16:32] == r_ref).all() and (got[64:80] == 0).all()


@pytest.mark.parametrize("fx_grad", [False, True, 64], indirect=True)
def test_epoch_with_an_unmatched_image_keeps_training_and_its_statistics(tmp_path, room, fx_grad):
    """An image in which the 2-D matching explained nothing (every label -1: /root/reference/Mask2Former_sample/
    match_seg.py:111-138 writes such masks for a camera inside an object) gives batches without a single labelled ray.
    Their cross entropy is the mean over an empty set - NaN, as torch's - but all rays are pruned, so the step touches no
    parameter; `train_one_epoch` counts it (`stats['nan_steps']`) and leaves it out of the epoch's mean loss, which it
    reads from the device once per epoch."""
    from instance_nerf_amd.nerf import NeRFNetwork
    from instance_nerf_amd.nerf.provider import NeRFDataset
    from instance_nerf_amd.nerf.utils import Trainer
    sc = room.write_dataset(str(tmp_path / "s"), n_views=6, H=48, W=64, num_instances=8, ignore_frac=0.1)
    np.save(os.path.join(sc["mask_dir"], "0002.npy"), np.full((48, 64), -1, np.int32))
    ds = NeRFDataset(sc["path"], type="train", device=DEV, scale=1.0, num_rays=1024, mask_dir=sc["mask_dir"], num_instances=8)
    torch.manual_seed(0)
    net = NeRFNetwork(cuda_ray=True, bound=1, min_near=0.05, density_thresh=10, num_instances=8).to(DEV)
    net.density_bitfield.copy_(_t(room.density_bitfield(128, 1.0)))
    tr = Trainer("nan", None, net, stage="instance", device=torch.device(DEV), lr=1e-2, iters=100, workspace=None, mute=True,
                 update_extra_interval=10 ** 9)
    tr.global_step = 1
    for _ in range(3):
        tr.train_one_epoch(ds.dataloader())
    assert tr.stats["nan_steps"] == [1, 1, 1] and len(tr.stats["loss"]) == 3
    assert all(np.isfinite(v) for v in tr.stats["loss"]) and tr.stats["loss"][-1] < tr.stats["loss"][0]
    assert all(bool(torch.isfinite(p).all()) for p in net.parameters())
    assert tr.global_step == 1 + 18


def test_fused_loader_at_full_size():
    """BASELINE sizes: an 800x800 image, 2^20 draws in one launch - every index against the oracle's draw, all in range,
    uniform over the image (chi-square of 256 equal bins), rays of unit length, colours and labels those of the drawn
    pixels, and the draw of one step disjoint in sequence from the next step's."""
    from instance_nerf_amd import _lib
    from oracle import rays as orays
    lib = _lib.load()
    H = W = 800
    n = 1 << 20
    rng = np.random.default_rng(3)
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0].astype(np.float32)
    img = torch.rand(H, W, 3, device=DEV)
    mask = torch.randint(-1, 70, (H, W), device=DEV, dtype=torch.int32)
    P = _t(pose)
    outs = []
    for step in (5, 6):
        inds = torch.empty(n, dtype=torch.int64, device=DEV)
        ro, rd = torch.empty(n, 3, device=DEV), torch.empty(n, 3, device=DEV)
        rgb = torch.empty(n, 3, device=DEV)
        lab = torch.empty(n, dtype=torch.int64, device=DEV)
        _lib.check(lib.inr_sample_training_batch(_lib.ptr(P), 400.0, 400.0, 400.0, 400.0, H, W, _lib.ptr(img), 3, _lib.ptr(mask), 64,
                                                 99, step, n, _lib.ptr(inds), _lib.ptr(ro), _lib.ptr(rd), _lib.ptr(rgb), _lib.ptr(lab),
                                                 _lib.stream_ptr()))
        outs.append(inds.clone())
        if step == 5:
            ref = orays.sample_pixels(99, 5, n, H, W)
            assert (inds.cpu().numpy() == ref).all()
            assert int(inds.min()) >= 0 and int(inds.max()) < H * W
            counts = torch.bincount(inds // 2500, minlength=256).double()
            chi2 = float(((counts - n / 256) ** 2 / (n / 256)).sum())
            assert chi2 < 255 + 5 * (2 * 255) ** 0.5, chi2
            assert float((rd.norm(dim=-1) - 1).abs().max()) < 2e-6 and torch.equal(ro, P[:3, 3].expand(n, 3))
            assert torch.equal(rgb, img.view(-1, 3)[inds])
            m = mask.view(-1)[inds].long()
            assert torch.equal(lab, torch.where(m >= 64, torch.full_like(m, -1), m))
    assert float((outs[0] == outs[1]).double().mean()) < 1e-4


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_fixed_point_table_gradient_fuzz_over_level_tables(seed):
    """The int32 form of the scatter over the level tables of the encoder fuzz (2..16 levels, base 4..32, 2^8..2^19 rows,
    finest level 64..4096, bound 1/2/4, points partly outside the volume): second step on int32 sums against the oracle's
    scatter-add (1e-5 norm-wise; north_star: 1e-3), multiples of each level's quantum, the same bits when repeated."""
    from instance_nerf_amd import _lib
    from instance_nerf_amd.gridencoder import GridEncoder
    from oracle import hashgrid
    lib = _lib.load()
    rng = np.random.default_rng(900 + seed)
    L = int(rng.choice([2, 5, 8, 13, 16]))
    base = int(rng.choice([4, 16, 32]))
    log2_t = int(rng.choice([8, 12, 15, 19]))
    res = int(rng.choice([64, 512, 2048, 4096]))
    bound = float(rng.choice([1.0, 2.0, 4.0]))
    enc = GridEncoder(num_levels=L, base_resolution=base, log2_hashmap_size=log2_t, desired_resolution=res).to(DEV)
    tb = hashgrid.level_table(num_levels=L, base_resolution=base, log2_hashmap_size=log2_t, desired_resolution=res)
    T, desc, offs = int(tb["total_rows"]), enc.desc, tb["offsets"]
    gen = torch.Generator().manual_seed(seed)
    M = 20000
    x = ((torch.rand(M, 3, generator=gen) * 2.2 - 1.1) * bound).contiguous()
    go = (torch.randn(M, 2 * L, generator=gen) * 10.0 ** (torch.rand(M, 1, generator=gen) * 4 - 6)).contiguous()
    ref = hashgrid.encode_backward_table(x, go, bound, tb)
    xd, god = x.to(DEV), go.to(DEV)
    fx = torch.zeros(_lib.GRID_FX_STATE_FLOATS, device=DEV)

    def step():
        g = torch.zeros(T, 2, device=DEV)
        _lib.check(lib.inr_grid_encode_backward_levels_fx(_lib.ptr(xd), _lib.ptr(god), None, desc, M, bound, _lib.ptr(g), 0, L,
                                                          _lib.ptr(fx), _lib.stream_ptr()))
        _lib.check(lib.inr_grid_grad_finish_fx(_lib.ptr(g), desc, 0, L, _lib.ptr(fx), _lib.stream_ptr()))
        _lib.check(lib.inr_grid_fx_update(_lib.ptr(fx), L, 128.0, 32, _lib.stream_ptr()))
        return g
    step()                                                       # fp32 atomics: sets the scales
    scales = fx[:16].cpu().numpy().copy()
    assert (scales[:L] > 0).all() and (scales[L:] == 0).all()
    saved = fx.clone()
    g2 = step()
    nrm = float(ref.norm())
    assert float((g2.cpu() - ref).norm()) < 1e-5 * nrm, (L, base, log2_t, res, bound)
    for l in range(L):
        q = (g2[offs[l]:offs[l + 1]].double() * float(scales[l])).cpu()
        assert bool((q == q.round()).all()), l
    fx.copy_(saved)
    assert torch.equal(step(), g2)


@pytest.mark.parametrize("bound", [1.0, 4.0])
def test_int64_table_gradient(level_table, bound):
    """The 64-bit form of the order-independent scatter (include/inr.h inr_grid_encode_backward_levels_fx64): int64 sums in a
    separate accumulator with a quantum of ~2e-16 of each level's maximum.  First step without scales: fp32 atomics straight
    into the gradient; second step: the accumulator - the oracle's scatter-add to fp32 rounding (tighter than the fp32
    atomics themselves), the SAME BITS when repeated or split into the two level ranges of the N > 1 schedule, the
    accumulator zero again afterwards; a 100x jump of the gradient uses 0.05 of the int64 range."""
    from instance_nerf_amd import _lib
    from instance_nerf_amd.gridencoder import GridEncoder
    from oracle import hashgrid
    lib = _lib.load()
    tb = level_table if bound == 1.0 else hashgrid.level_table(desired_resolution=8192)
    enc = GridEncoder(desired_resolution=int(tb["resolutions"][-1])).to(DEV)
    L, T, desc = 16, int(tb["total_rows"]), enc.desc
    gen = torch.Generator().manual_seed(31)
    M = 60000
    o = (torch.rand(M // 50, 1, 3, generator=gen) * 1.6 - 0.8) * bound
    d = torch.nn.functional.normalize(torch.randn(M // 50, 1, 3, generator=gen), dim=-1)
    x = (o + d * torch.linspace(0, 0.5 * bound, 50).view(1, 50, 1)).reshape(-1, 3).clamp(-bound, bound).contiguous()
    go = (torch.randn(M, 32, generator=gen) * torch.logspace(-12, -2, M).view(-1, 1)[torch.randperm(M, generator=gen)]).contiguous()
    ref64 = hashgrid.encode_backward_table(x, go, bound, tb)
    xd, god = x.to(DEV), go.to(DEV)
    fx = torch.zeros(_lib.GRID_FX_STATE_FLOATS, device=DEV)
    acc = torch.zeros(T, 2, dtype=torch.int64, device=DEV)

    def step(g_out, ranges=((0, L),)):
        g = torch.zeros(T, 2, device=DEV)
        for lo, hi in ranges:
            _lib.check(lib.inr_grid_encode_backward_levels_fx64(_lib.ptr(xd), _lib.ptr(g_out), None, desc, M, float(bound), _lib.ptr(g),
                                                                _lib.ptr(acc), lo, hi, _lib.ptr(fx), _lib.stream_ptr()))
            _lib.check(lib.inr_grid_grad_finish_fx64(_lib.ptr(acc), _lib.ptr(g), desc, lo, hi, _lib.ptr(fx), _lib.stream_ptr()))
        _lib.check(lib.inr_grid_fx_update(_lib.ptr(fx), L, 1024.0, 64, _lib.stream_ptr()))
        return g
    nrm = float(ref64.norm())
    g1 = step(god)                                            # no scales: fp32 atomics into g
    assert float((g1.cpu() - ref64).norm()) < 2e-6 * nrm and int(acc.abs().max()) == 0
    scales = fx[:16].cpu().numpy().astype(np.float64)
    offs = tb["offsets"]
    for l in range(16):                                       # 1024 x the level's maximum fits: 2^51 < max * scale <= 2^52
        mx = float(ref64[offs[l]:offs[l + 1]].abs().max())
        assert 2.0 ** 51 * 0.999 < mx * scales[l] <= 2.0 ** 52 * 1.001, (l, mx, scales[l])
    saved = fx.clone()
    g2 = step(god)
    assert float((g2.cpu() - ref64).norm()) < 1e-6 * nrm and int(acc.abs().max()) == 0
    # rows the int32 form would freeze are alive here: every non-zero row of the oracle is non-zero
    assert float(((ref64 != 0) & (g2.cpu() == 0)).double().mean()) < 1e-3
    fx.copy_(saved)
    assert torch.equal(step(god, ranges=((8, L), (0, 8))), g2)
    g3 = step((god * 100).contiguous())                       # 100 x against a headroom of 1024: 0.05 of the range, no near miss
    st = fx.cpu().numpy()
    assert float((g3.cpu() - 100 * ref64).norm()) < 1e-6 * 100 * nrm and st[49] == 0 and 0.02 < st[80:96].max() < 0.06
    with pytest.raises(RuntimeError):
        _lib.check(lib.inr_grid_fx_update(_lib.ptr(fx), L, 1024.0, 48, _lib.stream_ptr()))
