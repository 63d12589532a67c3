"""Golden vectors recorded FROM THE REFERENCE'S OWN PYTHON at the boundaries of the "next" rows:

* how the reference really calls the RoIAlign extension (`/root/reference/nerf_rcnn/model/utils.py:556-609`,
  `model/poolers.py:24-61,115-188`): a RECORDING stand-in for the un-vendored `roi_align` extension module is installed,
  the reference's `roi_align_3d` / `_multiscale_roi_align_3d` / `LevelMapper` are driven on small inputs, and every call
  is written down as it arrives - tensor values, dtypes, strides (the rois are a NON-contiguous column slice), the int
  index type, sizes and scale;
* the 3-D mask file the projector reads (`run_rcnn.py:652-666`): masks produced by the reference's
  `paste_masks_in_image` (`model/utils.py:646-782`) stored with the writer's key names, top-k ordering and dtypes;
* the feature-file metadata keys (`scripts/proposals2ngp.py:16-61`): the reference's `proposals_to_ngp_boxes` run on a
  metadata dict with exactly the keys `instance_nerf_amd.extract.write_features_npz` writes.

Runs only in the build container (needs /root/reference).  Output: tests/golden/reference_calls.npz (data only).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/nerf_rcnn"
OUT = os.path.dirname(os.path.abspath(__file__))

CALLS = []


def _recording_roi_align_3d(input, rois, roi_inds, out_w, out_l, out_h, spatial_scale):
    CALLS.append(dict(input=input.detach().clone(), input_contig=input.is_contiguous(), rois=rois.detach().clone(),
                      rois_contig=rois.is_contiguous(), rois_stride=tuple(rois.stride()),
                      rois_storage_offset=int(rois.storage_offset()), roi_inds=roi_inds.detach().clone(),
                      inds_dtype=str(roi_inds.dtype), inds_contig=roi_inds.is_contiguous(),
                      sizes=(int(out_w), int(out_l), int(out_h)), scale=float(spatial_scale),
                      scale_type=type(spatial_scale).__name__))
    # a stand-in result of the right shape; tagged by call so that the reference's scatter of per-level results
    # (poolers.py:173-182) can be replayed from the recorded calls alone
    return torch.full((rois.shape[0], input.shape[1], out_w, out_l, out_h), float(len(CALLS)), dtype=input.dtype)


def main():
    ext = types.ModuleType("roi_align")
    ext.roi_align = types.ModuleType("roi_align.roi_align")
    ext.roi_align.roi_align_3d = _recording_roi_align_3d
    sys.modules["roi_align"] = ext
    sys.modules["roi_align.roi_align"] = ext.roi_align
    for m in ("sort_vertices", "wandb", "cv2", "h5py"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path.insert(0, REF)
    from model import poolers, utils as mu

    out = {}
    g = torch.Generator().manual_seed(0)

    # ---- 1. single-level call through model/utils.py::roi_align_3d, Tensor[K,7] and List[Tensor[L,6]] inputs ----
    def rnd(*shape):                      # few distinct values: the fixture compresses to ~1 byte per value
        return torch.randint(-16, 17, shape, generator=g).float() / 8
    feat = rnd(2, 3, 9, 8, 7)
    lo = torch.rand(5, 3, generator=g) * 20
    boxes7 = torch.cat([torch.tensor([0., 1, 1, 0, 1])[:, None], lo, lo + 3 + torch.rand(5, 3, generator=g) * 12], 1)
    mu.roi_align_3d(feat, boxes7, output_size=(3, 2, 4), spatial_scale=0.25, sampling_ratio=2)
    lst = [boxes7[boxes7[:, 0] == b][:, 1:].contiguous() for b in range(2)]
    mu.roi_align_3d(feat, lst, output_size=2, spatial_scale=0.25)

    # ---- 2. multi-scale pooling (poolers.py::MultiScaleRoIAlign3D -> _multiscale_roi_align_3d) ----
    feats = [rnd(2, 2, 40 // s, 32 // s, 24 // s) for s in (1, 2, 4)]   # strides 4, 8, 16 of a 160-grid
    image_shapes = [(160, 128, 96), (150, 120, 90)]
    sizes = torch.tensor([[6., 6, 6], [30, 28, 25], [70, 64, 50], [130, 100, 80], [12, 40, 90], [200, 180, 170]])
    b0 = torch.cat([torch.rand(6, 3, generator=g) * 20, torch.zeros(6, 3)], 1)
    b0[:, 3:] = b0[:, :3] + sizes
    b1 = b0[[2, 0, 5, 4]].clone() + 1.5
    pool = poolers.MultiScaleRoIAlign3D(output_size=4, sampling_ratio=2, canonical_scale=160, canonical_level=4)
    n_before = len(CALLS)
    res = pool(feats, [b0, b1], image_shapes)
    out["ms_scales"] = np.asarray(pool.scales, np.float64)
    out["ms_kmin_kmax"] = np.asarray([pool.map_levels.k_min, pool.map_levels.k_max])
    out["ms_levels"] = pool.map_levels([b0, b1]).numpy()
    out["ms_boxes0"], out["ms_boxes1"] = b0.numpy(), b1.numpy()
    out["ms_image_shapes"] = np.asarray(image_shapes)
    out["ms_n_calls"] = np.asarray(len(CALLS) - n_before)
    out["ms_result_tags0"] = res[0][:, 0, 0, 0, 0].numpy()       # which recorded call produced each RoI's rows
    out["ms_result_tags1"] = res[1][:, 0, 0, 0, 0].numpy()
    for i, f in enumerate(feats):
        out[f"ms_feat{i}"] = f.numpy()

    # ---- 3. LevelMapper alone (poolers.py:24-61) on edge sizes: exact powers of two around the canonical scale ----
    lm = poolers.LevelMapper(2, 4, canonical_scale=160, canonical_level=4)
    edge = torch.tensor([10., 39.99, 40, 40.01, 79.99, 80, 80.01, 160, 320])
    eb = torch.zeros(len(edge), 6)
    eb[:, 3:] = edge[:, None]
    out["lm_sides"], out["lm_levels"] = edge.numpy(), lm([eb]).numpy()

    out["n_calls"] = np.asarray(len(CALLS))
    for i, c in enumerate(CALLS):
        # the feature tensor of a call is stored once: by the key of an identical array already in the file
        same = [k for k, v in out.items() if isinstance(v, np.ndarray) and v.shape == tuple(c["input"].shape)
                and v.dtype == np.float32 and (v == c["input"].numpy()).all()]
        if same:
            out[f"call{i}_input_key"] = np.asarray(same[0])
        else:
            out[f"call{i}_input"] = c["input"].numpy()
            out[f"call{i}_input_key"] = np.asarray(f"call{i}_input")
        out[f"call{i}_rois"] = c["rois"].numpy()
        out[f"call{i}_roi_inds"] = c["roi_inds"].numpy()
        out[f"call{i}_meta"] = np.asarray([int(c["input_contig"]), int(c["rois_contig"]), c["rois_stride"][0],
                                           c["rois_stride"][1], c["rois_storage_offset"], int(c["inds_contig"]),
                                           *c["sizes"]], np.int64)
        out[f"call{i}_scale"] = np.asarray(c["scale"], np.float64)
        out[f"call{i}_types"] = np.asarray([c["inds_dtype"], c["scale_type"], str(c["input"].dtype), str(c["rois"].dtype)])

    # ---- 4. 3-D masks as run_rcnn.py:652-666 writes them ----
    K, side = 6, 8
    mprob = torch.rand(K, side, side, side, generator=g)
    boxes = torch.tensor([[2., 3, 1, 14, 12, 10], [0, 0, 0, 6, 6, 6], [10, 2, 5, 23, 9, 17], [-3, 4, 2, 8, 19, 9],
                          [5, 5, 5, 9, 9, 9], [1, 8, 3, 22, 17, 15]])
    grid_shape = (24, 20, 18)
    pasted = mu.paste_masks_in_image(mprob, boxes, grid_shape)
    pasted = pasted if torch.is_tensor(pasted) else torch.as_tensor(pasted)
    scores = torch.tensor([0.3, 0.9, 0.5, 0.7, 0.1, 0.8])
    labels = torch.ones(K, dtype=torch.int64)
    top_k = 4                                                       # run_rcnn.py:658-664 (save_top_k)
    inds = np.argsort(scores.numpy())[::-1][:top_k]
    out["mask_in_prob"], out["mask_in_boxes"] = mprob.numpy(), boxes.numpy()
    out["mask_grid_shape"] = np.asarray(grid_shape)
    out["mask_pasted_all"] = pasted.numpy()
    out["masks"], out["scores"] = pasted.numpy()[inds], scores.numpy()[inds]
    out["labels"], out["boxes"] = labels.numpy()[inds], boxes.numpy()[inds]

    # ---- 5. feature-file metadata through scripts/proposals2ngp.py::proposals_to_ngp_boxes ----
    sys.path.insert(0, os.path.join(REF, "scripts"))
    sys.modules.setdefault("tqdm", types.ModuleType("tqdm")).tqdm = lambda x, **k: x
    import proposals2ngp as p2n
    meta = dict(resolution=np.asarray([24, 20, 18]), bbox_min=np.asarray([-1.0, -0.8, -0.6]),
                bbox_max=np.asarray([1.0, 0.9, 0.75]), scale=np.asarray(0.33), offset=np.asarray([0.5, 0.45, 0.55]),
                from_mitsuba=np.asarray(False))
    props = boxes.numpy()[:4]
    got = p2n.proposals_to_ngp_boxes(props, meta)
    for k, v in meta.items():
        out[f"meta_{k}"] = v
    out["meta_proposals"] = props
    out["meta_orientation"] = np.asarray([b["orientation"] for b in got])
    out["meta_position"] = np.asarray([b["position"] for b in got])
    out["meta_extents"] = np.asarray([b["extents"] for b in got])

    np.savez_compressed(os.path.join(OUT, "reference_calls.npz"), **out)
    print("calls recorded:", len(CALLS))
    for i, c in enumerate(CALLS):
        print(i, tuple(c["input"].shape), tuple(c["rois"].shape), "contig", c["rois_contig"], "stride", c["rois_stride"],
              c["inds_dtype"], c["sizes"], c["scale"], c["scale_type"])
    print("mask dtype", pasted.dtype, tuple(pasted.shape), "levels", out["ms_levels"], "lm", out["lm_levels"])


if __name__ == "__main__":
    main()
