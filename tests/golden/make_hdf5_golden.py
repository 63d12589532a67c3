"""Golden fixtures for the .hdf5 mirror of the matched instance-id maps (/root/reference/Mask2Former_sample/match_seg.py:
142-143: `h5py.File(<img>.hdf5, 'w').create_dataset('cp_instance_id_segmaps', data=output)`), written here by the REAL
h5py / libhdf5 exactly as the reference writes them - the image's main interpreter has no h5py, its conda python has:

    /opt/conda/bin/python3.9 tests/golden/make_hdf5_golden.py          (h5py 3.3.0, HDF5 1.10.6)

tests/golden/hdf5/*.hdf5 + expected.npz (the arrays, for the reader test of instance_nerf_amd/masks.py::read_hdf5_dataset).
Besides the reference's own call (contiguous layout, library defaults) two variants a BlenderProc container uses for the
same dataset name: chunked + gzip, chunked + shuffle + gzip."""
import os

import h5py
import numpy as np

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdf5")
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(0)
ids = rng.integers(-1, 31, size=(48, 64))                      # what match_seg.py builds: int64, -1 ignore, 0 background
ids[:8] = -1
expected = {}


def write(name, arr, **kw):
    with h5py.File(os.path.join(out, name), "w") as f:
        f.create_dataset("cp_instance_id_segmaps", data=arr, **kw)
    expected[name] = arr


write("match_seg_int64.hdf5", ids)                              # the reference's call, verbatim
write("match_seg_int32.hdf5", ids.astype(np.int32))
write("match_seg_uint8.hdf5", np.clip(ids, 0, 255).astype(np.uint8))
write("chunked_gzip.hdf5", ids.astype(np.int32), chunks=(16, 32), compression="gzip", compression_opts=4)
write("chunked_shuffle_gzip.hdf5", ids, chunks=(48, 64), compression="gzip", shuffle=True)
# 130 chunks: more than one leaf of the chunk B-tree holds (64 entries) -> an internal node above two leaves; blocky ids
# (an instance map is piecewise constant) so that the fixture stays small
big = np.repeat(np.repeat(rng.integers(-1, 31, size=(20, 26)), 8, 0), 8, 1).astype(np.int32)       # [160, 208]
write("chunked_many.hdf5", big, chunks=(16, 16), compression="gzip")
with h5py.File(os.path.join(out, "two_datasets.hdf5"), "w") as f:          # the wanted dataset is not the only object
    f.create_dataset("colors", data=rng.random((48, 64, 3)).astype(np.float32))
    f.create_dataset("cp_instance_id_segmaps", data=ids.astype(np.int16))
    f.create_group("extras").create_dataset("depth", data=rng.random((4, 4)))
expected["two_datasets.hdf5"] = ids.astype(np.int16)
write("big_endian_float.hdf5", rng.random((5, 7)).astype(">f4"))
# compact layout (the data inside the object header) through the low-level API
small = rng.integers(-1, 31, size=(6, 5)).astype(np.int32)
with h5py.File(os.path.join(out, "compact.hdf5"), "w") as f:
    dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
    dcpl.set_layout(h5py.h5d.COMPACT)
    space = h5py.h5s.create_simple(small.shape)
    dsid = h5py.h5d.create(f.id, b"cp_instance_id_segmaps", h5py.h5t.NATIVE_INT32, space, dcpl)
    dsid.write(h5py.h5s.ALL, h5py.h5s.ALL, small)
expected["compact.hdf5"] = small
# what the reader must REFUSE by name instead of misreading: the newer file format (version-2 object headers)
with h5py.File(os.path.join(out, "libver_latest.hdf5"), "w", libver="latest") as f:
    f.create_dataset("cp_instance_id_segmaps", data=ids)
np.savez_compressed(os.path.join(out, "expected.npz"), **{k.replace(".hdf5", ""): v for k, v in expected.items()})
print({k: (v.shape, str(v.dtype), os.path.getsize(os.path.join(out, k))) for k, v in expected.items()})
