"""Consumer-side contracts of the reference for the "next" rows (oracle; test infrastructure only).

These ARE pinned to the reference: /root/reference/nerf_rcnn/datasets.py is importable in the build
container, and tests/golden/make_reference_golden.py runs the reference's own functions to produce
tests/golden/features_consumer.npz, against which the restatements below are checked.
"""
import numpy as np


def ngp_density_to_alpha(density):
    """/root/reference/nerf_rcnn/datasets.py:865-866 - channel 3 of rgbsigma is the RAW (pre-activation)
    density; alpha = clip(1 - exp(-exp(d) / 100), 0, 1)."""
    return np.clip(1.0 - np.exp(-np.exp(density) / 100.0), 0.0, 1.0)


def load_feature(npz, normalize_density=True, transpose_yz=False):
    """/root/reference/nerf_rcnn/datasets.py:766-792 (SegmentationDataset.load_feature), numpy only.

    npz: mapping with 'rgbsigma' ([W,L,H,4] or flat [H*L*W,4]) and 'resolution' [3].
    Returns float array [C, W, L, H] (the 4-D case is returned as (3,0,1,2) transpose).
    """
    rgbsigma = np.array(npz["rgbsigma"])
    if normalize_density:
        rgbsigma[..., -1] = ngp_density_to_alpha(rgbsigma[..., -1])
    res = npz["resolution"]
    if rgbsigma.ndim == 2:
        rgbsigma = rgbsigma.reshape(res[2], res[1], res[0], -1)
        rgbsigma = np.transpose(rgbsigma, (3, 0, 2, 1) if transpose_yz else (3, 2, 1, 0))
    else:
        rgbsigma = np.transpose(rgbsigma, (3, 0, 1, 2))
    if rgbsigma.dtype == np.uint8:
        rgbsigma = rgbsigma.astype(np.float32) / 255.0
    return rgbsigma
