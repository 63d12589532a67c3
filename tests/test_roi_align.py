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


@pytest.mark.gpu
def test_hip_roi_align_matches_oracle_and_autograd():
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    rng = np.random.default_rng(0)
    vol = rng.normal(size=(2, 3, 9, 8, 7)).astype(np.float32)
    rois = np.asarray([[0, 0, 0, 9, 8, 7], [1.3, 0.7, 2.2, 6.1, 7.5, 6.9], [2, 2, 2, 2.4, 2.5, 2.2],
                       [-3, -2, -1, 4, 5, 3], [5, 4, 3, 20, 20, 20], [0.5, 0.5, 0.5, 8.5, 7.5, 6.5]], np.float32) * 2
    inds = np.asarray([0, 1, 1, 0, 1, 0], np.int32)
    for osz, scale in (((3, 3, 3), 0.5), ((5, 4, 2), 0.5), ((2, 2, 2), 0.25)):
        ref = roialign.roi_align_3d(vol, rois, inds, *osz, scale)
        x = torch.tensor(vol, device="cuda", requires_grad=True)
        out = roi_align_3d(x, torch.tensor(rois, device="cuda"), torch.tensor(inds, device="cuda"), *osz, scale)
        assert np.abs(out.detach().cpu().numpy() - ref).max() < 1e-5
        # backward = transpose of the (linear) forward: <out, g> == <x, grad>
        g = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        out.backward(g)
        lhs = (out.detach() * g).sum().item()
        rhs = (x.detach() * x.grad).sum().item()
        assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
    empty = roi_align_3d(torch.tensor(vol, device="cuda"), torch.zeros(0, 6, device="cuda"),
                         torch.zeros(0, dtype=torch.int32, device="cuda"), 3, 3, 3, 1.0)
    assert empty.shape == (0, 3, 3, 3, 3)
