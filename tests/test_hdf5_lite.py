"""f3: the .hdf5 mirror of the matched instance-id maps (/root/reference/Mask2Former_sample/match_seg.py:142-143) read
without h5py.  The fixtures under tests/golden/hdf5/ were written by the REAL h5py 3.3.0 / HDF5 1.10.6 (this image's
conda interpreter) exactly as the reference writes them - tests/golden/make_hdf5_golden.py - plus the layouts a
BlenderProc container uses for the same dataset; expected.npz holds the arrays."""
import os
import shutil

import numpy as np
import pytest

from instance_nerf_amd import hdf5_lite

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5")


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(G, "expected.npz"))


def test_every_fixture_reads_back_bit_for_bit(expected):
    assert len(expected.files) >= 9
    for k in expected.files:
        got = hdf5_lite.read_dataset(os.path.join(G, k + ".hdf5"), "cp_instance_id_segmaps")
        want = expected[k]
        assert got.shape == want.shape and got.dtype == want.dtype.newbyteorder("="), k
        assert np.array_equal(got, want), k
    # the reference's own call wrote the first one: int64, contiguous layout, library defaults
    ref = hdf5_lite.read_dataset(os.path.join(G, "match_seg_int64.hdf5"), "cp_instance_id_segmaps")
    assert ref.dtype == np.int64 and ref.min() == -1 and ref.max() <= 30


def test_groups_other_datasets_and_missing_names():
    f = os.path.join(G, "two_datasets.hdf5")
    assert hdf5_lite.list_objects(f) == ["colors", "cp_instance_id_segmaps", "extras"]
    assert hdf5_lite.list_objects(f, "extras") == ["depth"]
    assert hdf5_lite.read_dataset(f, "colors").shape == (48, 64, 3)
    assert hdf5_lite.read_dataset(f, "extras/depth").dtype == np.float64
    with pytest.raises(KeyError):
        hdf5_lite.read_dataset(f, "nope")
    with pytest.raises(KeyError):
        hdf5_lite.read_dataset(f, "extras")                 # a group is not a dataset


def test_unsupported_files_are_refused_by_name(tmp_path):
    """A file in the newer format (libver='latest': superblock 3, version-2 object headers) and a file that is no HDF5
    at all raise - nothing is ever misread silently."""
    with pytest.raises(NotImplementedError, match="superblock"):
        hdf5_lite.read_dataset(os.path.join(G, "libver_latest.hdf5"), "cp_instance_id_segmaps")
    p = tmp_path / "x.hdf5"
    p.write_bytes(b"\x93NUMPY" + b"\0" * 600)
    with pytest.raises(ValueError, match="not an HDF5"):
        hdf5_lite.read_dataset(str(p), "cp_instance_id_segmaps")


def test_load_matched_masks_reads_the_hdf5_mirror(tmp_path, expected):
    """match_seg.py writes <img>.npy AND <img>.hdf5: a directory that kept only the mirror loads the same maps; when
    both are there the .npy wins; ``names`` filters both kinds."""
    from instance_nerf_amd.masks import load_matched_masks
    ids = expected["match_seg_int64"]
    np.save(tmp_path / "0001.npy", ids)
    shutil.copy(os.path.join(G, "match_seg_int64.hdf5"), tmp_path / "0001.hdf5")
    shutil.copy(os.path.join(G, "match_seg_int32.hdf5"), tmp_path / "0002.hdf5")
    shutil.copy(os.path.join(G, "chunked_gzip.hdf5"), tmp_path / "0003.hdf5")
    d = load_matched_masks(str(tmp_path))
    assert sorted(d) == ["0001", "0002", "0003"] and all(v.dtype == np.int32 for v in d.values())
    assert np.array_equal(d["0001"], ids) and np.array_equal(d["0002"], ids) and np.array_equal(d["0003"], ids)
    assert sorted(load_matched_masks(str(tmp_path), names=["0002"])) == ["0002"]
