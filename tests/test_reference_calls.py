"""What the reference's OWN Python does at the boundaries of the "next" rows, recorded in the build container by
tests/golden/make_reference_calls_golden.py (imports /root/reference/nerf_rcnn with a recording stand-in for the
un-vendored RoIAlign extension) and replayed here: the call convention of roi_align_3d (non-contiguous rois, int32
indices), the multi-scale pooler's level assignment, the 3-D mask file layout and the feature-file metadata.
The extension's ARITHMETIC stays unpinned (un-vendored submodule): values are compared with this repository's
torchvision-semantics oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import consumers, roialign

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(G, "reference_calls.npz"))


def _call(ref, i):
    key = str(ref[f"call{i}_input_key"])
    meta = ref[f"call{i}_meta"]
    return dict(input=ref[key], rois=ref[f"call{i}_rois"], roi_inds=ref[f"call{i}_roi_inds"], meta=meta,
                sizes=tuple(int(v) for v in meta[6:9]), scale=float(ref[f"call{i}_scale"]),
                types=[str(t) for t in ref[f"call{i}_types"]])


def test_how_the_reference_calls_the_extension(ref):
    """/root/reference/nerf_rcnn/model/utils.py:596-609: ``rois = rois[:, 1:]`` is a NON-contiguous view (row stride
    7, storage offset 1) of the [K,7] tensor, ``roi_inds`` is ``.contiguous().to(torch.int)`` = int32, the output size
    arrives as three ints and the scale as a Python float; sampling_ratio is never forwarded."""
    n = int(ref["n_calls"])
    assert n == 5
    for i in range(n):
        c = _call(ref, i)
        input_contig, rois_contig, s0, s1, off, inds_contig = (int(v) for v in c["meta"][:6])
        assert input_contig == 1 and inds_contig == 1
        assert (s0, s1, off) == (7, 1, 1)
        assert rois_contig == (1 if c["rois"].shape[0] == 1 else 0)        # a one-row slice counts as contiguous
        assert c["types"] == ["torch.int32", "float", "torch.float32", "torch.float32"]
        assert c["rois"].shape[1] == 6 and c["roi_inds"].dtype == np.int32
        assert c["roi_inds"].min() >= 0 and c["roi_inds"].max() < c["input"].shape[0]


def test_multiscale_level_assignment(ref):
    """poolers.py:115-188: one extension call per pyramid level that has boxes, scale 2^-k of the level, RoIs of all
    images concatenated with their image index; the level restatement (oracle/consumers.py::level_mapper) reproduces
    the reference's LevelMapper on the recorded boxes and on the power-of-two edge cases."""
    boxes = np.concatenate([ref["ms_boxes0"], ref["ms_boxes1"]])
    k_min, k_max = (int(v) for v in ref["ms_kmin_kmax"])
    assert (k_min, k_max) == (2, 4) and list(ref["ms_scales"]) == [0.25, 0.125, 0.0625]
    assert (consumers.level_mapper(boxes, k_min, k_max) == ref["ms_levels"]).all()
    eb = np.zeros((len(ref["lm_sides"]), 6), np.float32)
    eb[:, 3:] = ref["lm_sides"][:, None]
    assert (consumers.level_mapper(eb, 2, 4) == ref["lm_levels"]).all()
    # calls 2, 3, 4 are the pooler's: level l receives exactly the RoIs mapped to l, in order, image index in front
    levels = ref["ms_levels"]
    img = np.concatenate([np.zeros(len(ref["ms_boxes0"])), np.ones(len(ref["ms_boxes1"]))])
    assert int(ref["ms_n_calls"]) == 3
    tags = np.concatenate([ref["ms_result_tags0"], ref["ms_result_tags1"]])
    for l in range(3):
        c = _call(ref, 2 + l)
        sel = levels == l
        assert np.array_equal(c["rois"], boxes[sel]) and np.array_equal(c["roi_inds"], img[sel].astype(np.int32))
        assert c["scale"] == float(ref["ms_scales"][l]) and c["sizes"] == (4, 4, 4)
        assert c["input"].shape == ref[f"ms_feat{l}"].shape
        assert (tags[sel] == 3 + l).all()                                  # the pooler scattered that call's rows back


def test_oracle_accepts_the_recorded_calls(ref):
    c = _call(ref, 0)
    out = roialign.roi_align_3d(c["input"], c["rois"], c["roi_inds"], *c["sizes"], c["scale"])
    assert out.shape == (5, 3, 3, 2, 4) and np.isfinite(out).all()


def test_3d_mask_file_layout(ref, tmp_path):
    """run_rcnn.py:652-666: np.savez(masks=, scores=, labels=, boxes=) of the top-k detections by score, masks as the
    reference's paste_masks_in_image returns them (bool [k, W, L, H])."""
    from instance_nerf_amd.masks import load_3d_masks
    path = str(tmp_path / "scene.npz")
    np.savez(path, masks=ref["masks"], scores=ref["scores"], labels=ref["labels"], boxes=ref["boxes"])
    m = load_3d_masks(path)
    assert m["masks"].dtype == bool and m["masks"].shape == (4,) + tuple(int(v) for v in ref["mask_grid_shape"])
    assert ref["masks"].dtype == bool                                       # what the reference's paste produced
    assert (np.diff(m["scores"]) <= 0).all() and np.allclose(m["scores"], [0.9, 0.8, 0.7, 0.5])
    order = np.argsort(-ref["scores"], kind="stable")
    assert np.array_equal(m["masks"], ref["mask_pasted_all"][[1, 5, 3, 2]]) and (order == np.arange(4)).all()
    # every pasted mask lies inside its (clipped) box, as paste_masks_in_image builds it
    for mask, box in zip(m["masks"], m["boxes"]):
        idx = np.argwhere(mask)
        assert len(idx) and (idx.min(0) >= np.floor(np.maximum(box[:3], 0)) - 1).all() and (idx.max(0) <= np.ceil(box[3:]) + 1).all()
    np.savez(path, masks=ref["masks"], scores=ref["scores"])
    with pytest.raises(ValueError, match="missing keys"):
        load_3d_masks(path)


def test_feature_file_metadata_feeds_the_reference_consumer(ref, tmp_path):
    """scripts/proposals2ngp.py:16-61 reads resolution / bbox_min / bbox_max / scale / offset / from_mitsuba from the
    feature file: write_features_npz writes exactly those keys, and the restated consumer maps proposals to the boxes
    the reference's own function returned."""
    from instance_nerf_amd.extract import write_features_npz
    W, L, H = (int(v) for v in ref["meta_resolution"])
    path = write_features_npz(str(tmp_path / "scene.npz"), np.zeros((W, L, H, 4), np.float32), ref["meta_bbox_min"],
                              ref["meta_bbox_max"], scale=float(ref["meta_scale"]), offset=ref["meta_offset"],
                              from_mitsuba=bool(ref["meta_from_mitsuba"]))
    feats = np.load(path)
    assert {"rgbsigma", "resolution", "bbox_min", "bbox_max", "scale", "offset", "from_mitsuba"} <= set(feats.files)
    ori, pos, ext = consumers.proposals_to_ngp_boxes(ref["meta_proposals"], feats)
    assert np.allclose(ori, ref["meta_orientation"], atol=1e-6)
    assert np.allclose(pos, ref["meta_position"], atol=1e-5)
    assert np.allclose(ext, ref["meta_extents"], atol=1e-5)


@pytest.mark.gpu
def test_hip_roi_align_replays_the_reference_calls(ref):
    """Every recorded call through instance_nerf_amd.roi_align.roi_align_3d with tensors shaped the way the reference
    passes them - rois as the non-contiguous [:, 1:] view of a [K,7] tensor, int32 indices - against the oracle."""
    from instance_nerf_amd.roi_align.roi_align import roi_align_3d
    for i in range(int(ref["n_calls"])):
        c = _call(ref, i)
        full = torch.cat([torch.from_numpy(c["roi_inds"]).float()[:, None], torch.from_numpy(c["rois"])], 1).cuda()
        rois = full[:, 1:]
        assert rois.stride() == (7, 1) and rois.storage_offset() == 1
        inds = full[:, 0].contiguous().to(torch.int)
        x = torch.from_numpy(c["input"]).cuda().requires_grad_(True)
        out = roi_align_3d(x, rois, inds, *c["sizes"], c["scale"])
        want = roialign.roi_align_3d(c["input"], c["rois"], c["roi_inds"], *c["sizes"], c["scale"])
        # fp32 sums of up to 9^3 samples per bin (the 130-voxel box on the finest level) against the float64 oracle
        err = np.abs(out.detach().cpu().numpy() - want).max()
        assert out.shape == want.shape and err < 5e-5, (i, err)
        out.sum().backward()                                                # the pooler's output feeds a trained head
        assert x.grad is not None and torch.isfinite(x.grad).all()
