"""The 2-D mask matching rule (the step between ``project_3d_masks`` and the instance-field trainer): the oracle's
restatement (oracle/consumers.py) against what the REFERENCE's own ``match_seg()`` produced on the same inputs
(/root/reference/Mask2Former_sample/match_seg.py:94-150, run by tests/golden/make_match_seg_golden.py; the projected-mask
PNGs of that run were written by this repository's PNG writer and read by the reference)."""
import os

import numpy as np

from oracle import consumers

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "match_seg.npz")


def _case(z, img):
    info = [{"id": int(i), "isthing": bool(t), "name": str(n)} for (i, t, _), n in zip(z[f"info_{img}"], z[f"names_{img}"])]
    files, ids = consumers.projections_of([str(f) for f in z["proj_files"]], img)
    masks = [z["proj_" + f[:-4]] for f in files]
    return z[f"seg_{img}"], info, masks, ids


def test_matching_rule_equals_the_reference_run_bit_for_bit():
    z = np.load(G)
    for img in (str(i) for i in z["images"]):
        seg, info, masks, ids = _case(z, img)
        out = consumers.match_seg(consumers.convert_seg(seg, info), masks, ids)
        assert out.dtype == z[f"out_{img}"].dtype == np.int32
        assert np.array_equal(out, z[f"out_{img}"]), img
        assert np.array_equal(z[f"hdf5_{img}"], z[f"out_{img}"])             # the .hdf5 mirror holds the same map


def test_the_fixture_exercises_every_branch():
    z = np.load(G)
    files = [str(f) for f in z["proj_files"]]
    assert "0000_0.png" in files and consumers.projections_of(files, "0000")[1] == [12, 3, 5, 7]      # sorted by NAME; _0 dropped
    assert consumers.projections_of(files, "0001")[0][0] == "00010_9.png"     # prefix match: another image's file is taken
    assert consumers.projections_of(files, "0010") == ([], [])
    out = z["out_0000"]
    seg = z["seg_0000"]
    assert set(np.unique(out)) == {-1, 0, 3, 7}
    assert (out[seg == 0] == -1).all()                  # unlabeled
    assert (out[seg == 3] == 0).all()                   # wall-brick -> background
    assert (out[seg == 1] == 3).all()                   # chair: matched to instance 3
    assert (out[seg == 5] == -1).all()                  # a thing that no projection reaches: ignore
    assert (out[seg == 4] == 7).all()                   # shelf: instance 7 beats the sliver of instance 12
    assert (z["out_0002"][z["seg_0002"] == 2] == 21).all()       # two candidates: the larger IoU wins
    assert set(np.unique(z["out_0010"])) == {-1, 0}              # no projections: every segment ignored


def test_matched_maps_load_through_the_product_reader(tmp_path):
    """What the reference wrote (.npy as np.save, .hdf5 through h5py - here re-encoded as .npy only, the .hdf5 reader has
    its own fixtures) comes back from ``load_matched_masks`` as the int32 label maps the trainer samples."""
    from instance_nerf_amd.masks import labels_for_rays, load_matched_masks
    z = np.load(G)
    for img in (str(i) for i in z["images"]):
        np.save(tmp_path / f"{img}.npy", z[f"out_{img}"])
    got = load_matched_masks(str(tmp_path))
    assert sorted(got) == sorted(str(i) for i in z["images"])
    m = got["0002"]
    assert m.dtype == np.int32 and np.array_equal(m, z["out_0002"])
    lab = labels_for_rays(m, np.arange(m.size), num_instances=16)
    assert lab.shape == (m.size,) and set(lab.tolist()) == {-1, 0, 3, 7}       # id 21 >= 16 classes: ignored
