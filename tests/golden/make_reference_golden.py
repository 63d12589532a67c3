"""Golden vectors produced BY THE REFERENCE ITSELF for the consumer contracts of the "next" rows.

Runs only in the build container (needs /root/reference); imports the reference's
nerf_rcnn/datasets.py (with import stubs for packages the image lacks) and records what
SegmentationDataset.load_feature (datasets.py:766-792) and ngp_density_to_alpha (datasets.py:865-866)
return for small inputs.  Output: tests/golden/features_consumer.npz (data only).
"""
import io
import os
import sys
import types

import numpy as np

REF = "/root/reference/nerf_rcnn"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    for m in ("roi_align", "roi_align.roi_align", "sort_vertices", "wandb", "cv2", "h5py"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path.insert(0, REF)
    import datasets as ref

    rng = np.random.default_rng(0)
    W, L, H = 5, 6, 7
    grid = rng.normal(size=(W, L, H, 4)).astype(np.float32)
    grid[..., 3] = rng.uniform(-4, 9, size=(W, L, H)).astype(np.float32)           # raw densities
    flat = rng.normal(size=(H * L * W, 4)).astype(np.float32)
    flat[:, 3] = rng.uniform(-4, 9, size=H * L * W).astype(np.float32)
    u8 = rng.integers(0, 256, size=(W, L, H, 4)).astype(np.uint8)
    res = np.asarray([W, L, H], dtype=np.int64)
    out = dict(grid=grid, flat=flat, u8=u8, resolution=res,
               dens_in=np.linspace(-12, 12, 49).astype(np.float32))
    out["dens_alpha"] = ref.ngp_density_to_alpha(out["dens_in"])

    def run(arr, transpose_yz, normalize):
        buf = io.BytesIO()
        np.savez(buf, rgbsigma=arr, resolution=res)
        path = os.path.join(OUT, "_tmp_feature.npz")
        open(path, "wb").write(buf.getvalue())
        fake = types.SimpleNamespace(normalize_density=normalize, normalize_fn=ref.ngp_density_to_alpha,
                                     transpose_yz=transpose_yz)
        t = ref.SegmentationDataset.load_feature(fake, path)
        os.remove(path)
        return t.numpy()

    out["exp_grid"] = run(grid, False, True)
    out["exp_flat_noT"] = run(flat, False, True)
    out["exp_flat_T"] = run(flat, True, True)
    out["exp_u8"] = run(u8, False, False)
    np.savez_compressed(os.path.join(OUT, "features_consumer.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
