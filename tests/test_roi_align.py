"""f2: 3-D RoIAlign.  Parity UNPINNED (the reference's extension is an un-vendored submodule): the HIP op
is compared with this repository's torchvision-semantics oracle; the call signature is the reference's."""
import numpy as np
import pytest
import torch

from oracle import roialign


def test_oracle_identity_and_average():
    """A RoI covering whole voxels with out == size reproduces the volume; a 2x pooled RoI averages."""
    vol = np.arange(2 * 4 * 4 * 4, dtype=np.float32).reshape(1, 2, 4, 4, 4)
    out = roialign.roi_align_3d(vol, np.asarray([[0, 0, 0, 4, 4, 4]], np.float32), [0], 4, 4, 4, 1.0)
    # torchvision aligned=False samples bin centres at +0.5 -> trilinear between voxel i and i+1 (clamped at the end)
    x = np.clip(np.arange(4) + 0.5, 0, 3)
    lo, fr = np.floor(x).astype(int), x - np.floor(x)
    hi = np.minimum(lo + 1, 3)
    ref = vol[0]
    for ax in (1, 2, 3):
        a = np.take(ref, lo, axis=ax)
        b = np.take(ref, hi, axis=ax)
        shape = [1, 1, 1, 1]
        shape[ax] = 4
        ref = a * (1 - fr.reshape(shape)) + b * fr.reshape(shape)
    assert np.allclose(out[0], ref, atol=1e-5)


def test_signature_matches_reference_call():
    """The reference calls roi_align.roi_align.roi_align_3d(input, rois, roi_inds, ow, ol, oh, scale)."""
    import inspect
    import instance_nerf_amd.roi_align as ra
    sig = inspect.signature(ra.roi_align.roi_align_3d)
    assert list(sig.parameters) == ["input", "rois", "roi_inds", "out_w", "out_l", "out_h", "spatial_scale"]


def test_oracle_at_points_equals_the_full_oracle():
    rng = np.random.default_rng(3)
    vol = rng.normal(size=(2, 3, 6, 5, 7)).astype(np.float32)
    rois = np.asarray([[0.4, 0.2, 1.0, 9.0, 8.0, 11.0], [-2, 1, 3, 30, 6, 9]], np.float32)
    inds = np.asarray([1, 0], np.int32)
    full = roialign.roi_align_3d(vol, rois, inds, 3, 2, 4, 0.5)
    pts = np.stack([rng.integers(0, n, 64) for n in full.shape], 1)
    got = roialign.roi_align_3d_at(vol, rois, inds, 3, 2, 4, 0.5, pts)
    assert np.array_equal(got, full[tuple(pts.T)])


def _set_mode(mode):
    from instance_nerf_amd import _lib
    _lib.check(_lib.load().inr_roi_align_3d_set_mode(mode), "roi_align_3d_set_mode")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [2, 1, 0])       # separable, one lane per output element, auto
def test_hip_roi_align_matches_oracle_and_autograd(mode):
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    rng = np.random.default_rng(0)
    vol = rng.normal(size=(2, 3, 9, 8, 7)).astype(np.float32)
    rois = np.asarray([[0, 0, 0, 9, 8, 7], [1.3, 0.7, 2.2, 6.1, 7.5, 6.9], [2, 2, 2, 2.4, 2.5, 2.2],
                       [-3, -2, -1, 4, 5, 3], [5, 4, 3, 20, 20, 20], [0.5, 0.5, 0.5, 8.5, 7.5, 6.5],
                       [40, 40, 40, 60, 60, 60], [-30, -30, -30, -20, -20, -20], [-50, 1, 1, 90, 7, 6]], np.float32) * 2
    inds = np.asarray([0, 1, 1, 0, 1, 0, 1, 0, 1], np.int32)
    _set_mode(mode)
    try:
        for osz, scale in (((3, 3, 3), 0.5), ((5, 4, 2), 0.5), ((2, 2, 2), 0.25), ((7, 7, 7), 1.0), ((11, 13, 12), 0.5)):
            ref = roialign.roi_align_3d(vol, rois, inds, *osz, scale)
            x = torch.tensor(vol, device="cuda", requires_grad=True)
            out = roi_align_3d(x, torch.tensor(rois, device="cuda"), torch.tensor(inds, device="cuda"), *osz, scale)
            assert np.abs(out.detach().cpu().numpy() - ref).max() < 1e-5, (mode, osz)
            # backward = transpose of the (linear) forward: <out, g> == <x, grad>
            g = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
            out.backward(g)
            lhs = (out.detach() * g).sum().item()
            rhs = (x.detach() * x.grad).sum().item()
            assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs)), (mode, osz)
        empty = roi_align_3d(torch.tensor(vol, device="cuda"), torch.zeros(0, 6, device="cuda"),
                             torch.zeros(0, dtype=torch.int32, device="cuda"), 3, 3, 3, 1.0)
        assert empty.shape == (0, 3, 3, 3, 3)
    finally:
        _set_mode(0)


def config4_boxes(gen_seed=0, K=256, dev="cuda"):
    """bench.py's BASELINE configs[4] boxes (input-scale units, spatial_scale 0.25 onto a 40^3 level) plus the shapes
    a detector really emits: one box larger than the volume, one thinner than a voxel, one outside."""
    gen = torch.Generator(device=dev).manual_seed(gen_seed)
    lo = torch.rand(K, 3, device=dev, generator=gen) * 100
    rois = torch.cat([lo, lo + 10 + torch.rand(K, 3, device=dev, generator=gen) * 50], 1)
    rois[0] = torch.tensor([-20.0, -8, -4, 190, 170, 200], device=dev)
    rois[1] = torch.tensor([50.0, 60, 70, 50.5, 61, 70.2], device=dev)
    rois[2] = torch.tensor([400.0, 400, 400, 440, 450, 460], device=dev)
    rois[3] = torch.tensor([0.0, 0, 0, 160, 160, 160], device=dev)
    return rois


@pytest.mark.gpu
def test_roi_align_at_baseline_config4_size():
    """BASELINE configs[4] at its stated size: [1,256,40,40,40], 256 boxes -> 10^3.  2000 sampled output elements
    against the oracle; the two HIP implementations against each other on all 65.5 M; the backward as the
    transpose of the forward and separable against lane-per-output."""
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    dev = "cuda"
    feat = torch.randn(1, 256, 40, 40, 40, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    rois = config4_boxes()
    inds = torch.zeros(256, dtype=torch.int32, device=dev)
    outs, grads = {}, {}
    g = torch.randn(256, 256, 10, 10, 10, device=dev, generator=torch.Generator(device=dev).manual_seed(6))
    try:
        for mode in (2, 3, 1):       # separable (backward: workspace form), separable accumulating in place, lane per output
            _set_mode(mode)
            x = feat.clone().requires_grad_(True)
            out = roi_align_3d(x, rois, inds, 10, 10, 10, 0.25)
            out.backward(g)
            outs[mode], grads[mode] = out.detach(), x.grad
            lhs, rhs = (out.detach().double() * g.double()).sum().item(), (feat.double() * x.grad.double()).sum().item()
            assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs)), (mode, lhs, rhs)
    finally:
        _set_mode(0)
    assert (outs[2] - outs[1]).abs().max().item() < 2e-5
    assert (grads[2] - grads[1]).abs().max().item() < 1e-3 * grads[1].abs().max().item()
    assert (grads[3] - grads[1]).abs().max().item() < 1e-3 * grads[1].abs().max().item()
    assert outs[2][2].abs().max().item() == 0.0          # the box outside the volume pools nothing
    rng = np.random.default_rng(7)
    pts = np.stack([rng.integers(0, n, 2000) for n in (256, 256, 10, 10, 10)], 1)
    pts[:200, 0] = rng.integers(0, 4, 200)                # the special boxes get their share
    ref = roialign.roi_align_3d_at(feat.cpu().numpy(), rois.cpu().numpy(), inds.cpu().numpy(), 10, 10, 10, 0.25, pts)
    got = outs[2].cpu().numpy()[tuple(pts.T)]
    assert np.abs(got - ref).max() < 2e-5


@pytest.mark.gpu
def test_roi_align_channel_tails_and_batches():
    """Channel counts that are not multiples of the 4 a lane carries, several volumes, non-cubic levels."""
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    rng = np.random.default_rng(11)
    try:
        for C, shape in ((1, (5, 6, 4)), (5, (8, 5, 9)), (18, (6, 6, 6))):
            vol = rng.normal(size=(3, C) + shape).astype(np.float32)
            K = 7
            lo = rng.uniform(-2, 6, size=(K, 3))
            rois = np.concatenate([lo, lo + rng.uniform(0.2, 12, size=(K, 3))], 1).astype(np.float32)
            inds = rng.integers(0, 3, K).astype(np.int32)
            ref = roialign.roi_align_3d(vol, rois, inds, 4, 3, 5, 0.7)
            for mode in (2, 1):
                _set_mode(mode)
                out = roi_align_3d(torch.tensor(vol, device="cuda"), torch.tensor(rois, device="cuda"),
                                   torch.tensor(inds, device="cuda"), 4, 3, 5, 0.7)
                assert np.abs(out.cpu().numpy() - ref).max() < 1e-5, (C, shape, mode)
    finally:
        _set_mode(0)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(8)))
def test_roi_align_fuzz_against_the_oracle(seed):
    """Random volumes (4..11 cells a side, 1..9 channels, 1..3 volumes), output sizes 1..14 a side (all three
    accumulator widths of the separable forward), scales, and boxes of every kind - inside, straddling a face, larger
    than the volume (sampling grids > 2: the table path behind the four register taps), thinner than a voxel, outside:
    both HIP implementations against the oracle, and the backward as the forward's transpose."""
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    rng = np.random.default_rng(1000 + seed)
    N, C = int(rng.integers(1, 4)), int(rng.integers(1, 10))
    shape = tuple(int(v) for v in rng.integers(4, 12, 3))
    osz = tuple(int(v) for v in rng.integers(1, 15, 3)) if seed % 2 else tuple(int(v) for v in rng.integers(1, 6, 3))
    scale = float(rng.choice([1.0, 0.5, 0.25, 0.7]))
    K = 6
    lo = rng.uniform(-3, max(shape), size=(K, 3)) / scale
    ext = np.stack([rng.uniform(0.05, 3.0 * max(shape), 3) if k % 3 == 0 else rng.uniform(0.3, 6, 3) for k in range(K)]) / scale
    rois = np.concatenate([lo, lo + ext], 1).astype(np.float32)
    rois[-1] = np.asarray([500, 500, 500, 520, 510, 530], np.float32)               # outside
    inds = rng.integers(0, N, K).astype(np.int32)
    vol = rng.normal(size=(N, C) + shape).astype(np.float32)
    ref = roialign.roi_align_3d(vol, rois, inds, *osz, scale)
    g = torch.randn(ref.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed))
    try:
        for mode in (2, 1):
            _set_mode(mode)
            x = torch.tensor(vol, device="cuda", requires_grad=True)
            out = roi_align_3d(x, torch.tensor(rois, device="cuda"), torch.tensor(inds, device="cuda"), *osz, scale)
            assert np.abs(out.detach().cpu().numpy() - ref).max() < 2e-5, (mode, N, C, shape, osz, scale)
            out.backward(g)
            lhs = (out.detach().double() * g.double()).sum().item()
            rhs = (x.detach().double() * x.grad.double()).sum().item()
            assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs)), (mode, lhs, rhs)
    finally:
        _set_mode(0)


@pytest.mark.gpu
def test_backward_of_regions_wider_than_the_fixed_roles():
    """The separable backward gives every thread a fixed role per pass (a cell and an output column) when
    sampled-rows x out_h and the region's depth fit the workgroup; RoIs beyond that (here regions up to 30 cells at 12
    bins: 360 > 256, next to small ones in the same launch) take its generic loops.  Both against the lane-per-output
    kernel, and as the forward's transpose."""
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(3)
    feat = torch.randn(2, 6, 30, 30, 30, device=dev, generator=gen)
    rois = torch.tensor([[0, 0, 0, 30, 30, 30], [2, 3, 1, 29.5, 28, 30], [10, 10, 10, 14, 13, 12], [-4, 5, 8, 40, 9, 26],
                         [5, 5, 5, 6, 30, 7], [1, 1, 1, 27, 5, 5]], dtype=torch.float32, device=dev)
    inds = torch.tensor([0, 1, 0, 1, 1, 0], dtype=torch.int32, device=dev)
    g = torch.randn(6, 6, 12, 12, 12, device=dev, generator=gen)
    grads = {}
    try:
        for mode in (2, 1):
            _set_mode(mode)
            x = feat.clone().requires_grad_(True)
            out = roi_align_3d(x, rois, inds, 12, 12, 12, 1.0)
            out.backward(g)
            grads[mode] = x.grad
            lhs, rhs = (out.detach().double() * g.double()).sum().item(), (feat.double() * x.grad.double()).sum().item()
            assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs)), (mode, lhs, rhs)
    finally:
        _set_mode(0)
    assert (grads[2] - grads[1]).abs().max().item() < 1e-4 * grads[1].abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_backward_workspace_form(case):
    """inr_roi_align_3d_backward_ws (channel counts that are multiples of 16): accumulation into a channels-fastest scratch
    volume + transposing copy.  Several volumes, non-cubic levels, boxes inside / straddling a face / larger than the volume /
    thinner than a voxel / outside, regions wider than one role per thread (case 1), a cell no box reaches (grad exactly 0,
    although the call gets an uninitialised grad_input): against the lane-per-output kernel, the in-place separable
    kernel, and as the forward's transpose; the library refuses the form where it does not apply."""
    from instance_nerf_amd import _lib
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    N, C, shape, osz, scale = [(2, 16, (12, 9, 11), (5, 4, 6), 0.5), (1, 32, (30, 30, 30), (12, 12, 12), 1.0),
                               (3, 48, (6, 6, 6), (1, 2, 3), 0.7), (1, 64, (16, 20, 12), (7, 7, 7), 0.25)][case]
    rng = np.random.default_rng(50 + case)
    K = 9
    lo = rng.uniform(-3, max(shape), size=(K, 3)) / scale
    ext = np.stack([rng.uniform(0.05, 3.0 * max(shape), 3) if k % 3 == 0 else rng.uniform(0.3, 6, 3) for k in range(K)]) / scale
    rois = np.concatenate([lo, lo + ext], 1).astype(np.float32)
    rois[-1] = np.asarray([500, 500, 500, 520, 510, 530], np.float32)               # outside
    rois[0] = np.asarray([0, 0, 0] + [s / scale for s in shape], np.float32)         # the whole volume (widest region)
    inds = rng.integers(0, N, K).astype(np.int32)
    inds[0] = 0
    dev = "cuda"
    lib = _lib.load()
    assert lib.inr_roi_align_3d_backward_workspace_bytes(N, C, *shape, K, *osz) == N * C * int(np.prod(shape)) * 4
    assert lib.inr_roi_align_3d_backward_workspace_bytes(N, C + 3, *shape, K, *osz) == 0
    assert lib.inr_roi_align_3d_backward_workspace_bytes(N, C, *shape, K, 4, 20, 20) == 0
    vol = torch.tensor(rng.normal(size=(N, C) + shape).astype(np.float32), device=dev)
    g = torch.randn((K, C) + osz, device=dev, generator=torch.Generator(device=dev).manual_seed(case))
    grads = {}
    try:
        for mode in (2, 3, 1):
            _set_mode(mode)
            x = vol.clone().requires_grad_(True)
            torch.empty(N * C * int(np.prod(shape)), device=dev).fill_(float("nan"))   # what a recycled block may hold
            out = roi_align_3d(x, torch.tensor(rois, device=dev), torch.tensor(inds, device=dev), *osz, scale)
            out.backward(g)
            grads[mode] = x.grad
            lhs, rhs = (out.detach().double() * g.double()).sum().item(), (vol.double() * x.grad.double()).sum().item()
            assert abs(lhs - rhs) < 1e-5 * max(1.0, abs(lhs)), (mode, lhs, rhs)
            assert lib.inr_roi_align_3d_backward_workspace_bytes(N, C, *shape, K, *osz) == (N * C * int(np.prod(shape)) * 4 if mode == 2 else 0)
    finally:
        _set_mode(0)
    tol = 1e-4 * grads[1].abs().max().item()
    assert (grads[2] - grads[1]).abs().max().item() < tol and (grads[3] - grads[1]).abs().max().item() < tol
    assert bool(torch.isfinite(grads[2]).all())
    assert torch.equal(grads[2] == 0, grads[1] == 0) or ((grads[2] == 0) ^ (grads[1] == 0)).float().mean().item() < 1e-3
    if N > 1:                                            # volumes no box points at: exactly zero
        unused = [n for n in range(N) if n not in set(inds[:-1].tolist())]
        for n in unused:
            assert float(grads[2][n].abs().max()) == 0.0


@pytest.mark.gpu
def test_gt_mask_crop_call_shape():
    """The reference's THIRD call shape (round-4 verdict): the ground-truth mask crop of the mask loss,
    roi_align_3d(gt_masks[:, None], rois, (M, M, M), 1.0) with C = 1 on full-resolution [G,1,160,160,160] volumes and
    M = 20 (/root/reference/nerf_rcnn/model/nerf_rcnn.py:819-831, 846-849: M = the mask head's output side).  Boxes of
    object size in voxels (sampling grids up to 8 per bin), one per proposal, matched to one of G masks: 600 sampled
    output elements against the oracle, and what `auto` picks against the lane-per-output kernel on everything."""
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    dev = "cuda"
    G, K, M = 6, 40, 20
    gen = torch.Generator(device=dev).manual_seed(3)
    masks = (torch.rand(G, 160, 160, 160, device=dev, generator=gen) > 0.5).float()
    lo = torch.rand(K, 3, device=dev, generator=gen) * 100 + 4
    rois = torch.cat([lo, lo + 6 + torch.rand(K, 3, device=dev, generator=gen) * 50], 1)
    rois[0] = torch.tensor([0.0, 0, 0, 160, 160, 160], device=dev)             # the whole volume: grid 8 per bin
    rois[1] = torch.tensor([150.0, 150, 150, 170, 175, 158], device=dev)        # straddles the far faces
    inds = torch.randint(0, G, (K,), device=dev, generator=gen).to(torch.int32)
    vol = masks[:, None].contiguous()
    outs = {}
    try:
        for mode in (0, 1):
            _set_mode(mode)
            outs[mode] = roi_align_3d(vol, rois, inds, M, M, M, 1.0)
    finally:
        _set_mode(0)
    assert outs[0].shape == (K, 1, M, M, M)
    assert (outs[0] - outs[1]).abs().max().item() < 2e-5
    rng = np.random.default_rng(4)
    pts = np.stack([rng.integers(0, n, 600) for n in (K, 1, M, M, M)], 1)
    pts[:60, 0] = rng.integers(0, 2, 60)
    ref = roialign.roi_align_3d_at(vol.cpu().numpy(), rois.cpu().numpy(), inds.cpu().numpy(), M, M, M, 1.0, pts)
    assert np.abs(outs[0].cpu().numpy()[tuple(pts.T)] - ref).max() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [2, 1])
def test_non_finite_voxels_outside_a_box_do_not_leak_into_it(mode):
    """Round-4 advisor: the separable forward multiplied the cells of its four-cell z window that no sample touches by a
    zero weight - 0 x Inf = NaN where the lane-per-output kernel, the oracle and torchvision never read the voxel.  Every
    voxel further than the trilinear support from the box is set to Inf / NaN: the output must not change, in either
    kernel, for boxes in the interior and boxes against the far faces (window pulled back from the end of the row)."""
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(9)
    vol = torch.randn(1, 8, 24, 24, 24, device=dev, generator=gen)
    boxes = torch.tensor([[5.2, 6.1, 7.3, 11.8, 12.2, 13.9], [17.0, 16.5, 18.2, 23.6, 23.9, 23.2],
                          [0.0, 0.3, 0.1, 4.5, 3.9, 5.2], [9.0, 9.0, 9.0, 10.1, 10.4, 9.9]], device=dev)
    inds = torch.zeros(1, dtype=torch.int32, device=dev)
    _set_mode(mode)
    try:
        for b in range(boxes.shape[0]):
            box = boxes[b:b + 1]
            for osz in ((4, 4, 4), (7, 5, 3)):
                clean = roi_align_3d(vol, box, inds, *osz, 1.0)
                lo = torch.clamp(torch.floor(box[0, :3]).long(), min=0)                 # cells any sample can read:
                hi = torch.clamp(torch.floor(box[0, 3:]).long() + 1, max=23)            # floor(v) and floor(v) + 1
                keep = torch.zeros(24, 24, 24, dtype=torch.bool, device=dev)
                keep[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] = True
                bad = vol.clone()
                bad[0, :, ~keep] = float("inf")
                bad[0, 3, ~keep] = float("nan")
                got = roi_align_3d(bad, box, inds, *osz, 1.0)
                assert torch.isfinite(got).all(), (mode, b, osz)
                assert torch.equal(got, clean), (mode, b, osz)
    finally:
        _set_mode(0)


@pytest.mark.gpu
def test_registered_custom_op_equals_the_module_function():
    """torch.ops.inr.roi_align_3d (schema + fake implementation + autograd formula over the same C ABI) gives the module
    function's output and gradient bit for bit, and torch.library's opcheck accepts its registration."""
    from instance_nerf_amd import ops  # noqa: F401
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    rng = np.random.default_rng(2)
    vol = torch.tensor(rng.normal(size=(2, 5, 8, 7, 9)).astype(np.float32), device="cuda")
    rois = torch.tensor([[0.5, 1, 2, 12, 11, 14], [3, 2, 1, 9, 9, 9], [-4, 0, 0, 30, 5, 5]], dtype=torch.float32, device="cuda")
    inds = torch.tensor([0, 1, 1], dtype=torch.int32, device="cuda")
    a = vol.clone().requires_grad_(True)
    b = vol.clone().requires_grad_(True)
    ya = roi_align_3d(a, rois, inds, 4, 3, 5, 0.5)
    yb = torch.ops.inr.roi_align_3d(b, rois, inds, 4, 3, 5, 0.5)
    assert torch.equal(ya, yb)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-6)          # atomics: summation order
    torch.library.opcheck(torch.ops.inr.roi_align_3d.default, (vol, rois, inds, 4, 3, 5, 0.5),
                          test_utils=("test_schema", "test_faketensor"))
