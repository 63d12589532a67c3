"""Golden fixture for the 2-D mask matching step (/root/reference/Mask2Former_sample/match_seg.py:94-150), produced by
running the REFERENCE's own `match_seg()` on small synthetic inputs.

    python tests/golden/make_match_seg_golden.py

Two interpreters: the image's main python (this repository's PNG writer needs torch-free code only, but lives in a
module that imports torch) writes the inputs; the image's conda python (/opt/conda/bin/python3.9: real h5py 3.3.0,
matplotlib, tqdm - the main interpreter has no h5py) imports and runs the reference.  cv2 is in neither: the three calls
match_seg.py makes (imread, cvtColor(RGB2BGR), imwrite) are given by a PIL-backed stand-in module of a dozen lines.
Output: tests/golden/match_seg.npz (inputs + the reference's .npy outputs + what its .hdf5 mirrors hold).
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Mask2Former_sample"
CONDA = "/opt/conda/bin/python3.9"

STAGE2 = r'''
import json, os, sys, types
import numpy as np
from PIL import Image
cv2 = types.ModuleType("cv2")                      # absent from the image: the three calls match_seg.py makes
cv2.COLOR_RGB2BGR = 4
cv2.imread = lambda p: np.array(Image.open(p).convert("RGB"))[:, :, ::-1].copy()
cv2.cvtColor = lambda img, code: img[:, :, ::-1].copy()
cv2.imwrite = lambda p, img: Image.fromarray(img[:, :, ::-1].copy()).save(p)
sys.modules["cv2"] = cv2
os.chdir(sys.argv[1])                              # match_seg.py opens coco_id_to_name.json relative to the cwd
sys.path.insert(0, sys.argv[1])
import match_seg as ref
work = sys.argv[2]
ref.match_seg(os.path.join(work, "proj"), os.path.join(work, "seg"), os.path.join(work, "out"))
import h5py
names = {"things": ref.coco_things_id_to_name, "stuff": ref.coco_stuff_id_to_name}
json.dump({k: {str(i): v for i, v in d.items()} for k, d in names.items()}, open(os.path.join(work, "names.json"), "w"))
for f in sorted(os.listdir(os.path.join(work, "out"))):
    if f.endswith(".hdf5"):
        with h5py.File(os.path.join(work, "out", f), "r") as h:
            np.save(os.path.join(work, "out", f + ".npy"), h["cp_instance_id_segmaps"][...])
'''


def main():
    sys.path.insert(0, ROOT)
    from instance_nerf_amd.masks import save_png_gray
    rng = np.random.default_rng(0)
    H, W = 48, 64
    work = tempfile.mkdtemp()
    for d in ("proj", "seg", "out"):
        os.makedirs(os.path.join(work, d))
    coco = json.load(open(os.path.join(REF, "coco_id_to_name.json")))
    things, stuff = coco["thing_classes"], coco["stuff_classes"]
    fixture = {}
    imgs = ["0000", "0001", "0002", "0010"]
    for n, img in enumerate(imgs):
        # panoptic map: blocks of segment ids 1..6 on an unlabeled (0) canvas
        seg = np.zeros((H, W), np.int32)
        boxes = [(2, 2, 20, 18), (4, 24, 22, 44), (26, 4, 44, 30), (24, 36, 46, 60), (0, 48, 16, 63), (30, 30, 40, 36)]
        info = []
        cats = [("chair", True), ("couch", True), ("wall-brick", False), ("shelf", False), ("person", True), ("rug-merged", False)]
        for sid, ((y0, x0, y1, x1), (name, isthing)) in enumerate(zip(boxes, cats), start=1):
            seg[y0 + n:y1 + n, x0:x1] = sid
            lst = things if isthing else stuff
            info.append({"id": sid, "isthing": isthing, "category_id": lst.index(name)})
        np.save(os.path.join(work, "seg", f"{img}.npy"), seg)
        json.dump(info, open(os.path.join(work, "seg", f"{img}.json"), "w"))
        fixture[f"seg_{img}"] = seg
        fixture[f"info_{img}"] = np.asarray([[s["id"], int(s["isthing"]), s["category_id"]] for s in info], np.int64)
        fixture[f"names_{img}"] = np.asarray([(things if s["isthing"] else stuff)[s["category_id"]] for s in info])
        if img == "0010":
            continue                                # an image without projections: every segment becomes -1
        # projected 3-D masks: instance 3 covers most of segment 1, instance 7 straddles segments 2 and 4, instance 12 a
        # sliver of segment 4 (IoU below the threshold), instance 5 only background, `_0.png` must be skipped
        projs = {3: (3, 3, 19, 19), 7: (10, 30, 40, 50), 12: (44, 58, 47, 62), 5: (30, 6, 40, 20), 0: (0, 0, 48, 64)}
        if n == 2:
            projs[21] = (4 + n, 24, 22 + n, 44)     # two candidates for segment 2: the larger IoU wins
        for inst, (y0, x0, y1, x1) in projs.items():
            m = np.zeros((H, W), np.uint8)
            m[y0:y1, x0:x1] = 255
            m &= (rng.random((H, W)) > 0.05).astype(np.uint8) * 255          # ragged edges
            save_png_gray(os.path.join(work, "proj", f"{img}_{inst}.png"), m)
            fixture[f"proj_{img}_{inst}"] = m > 0
    # a projection of ANOTHER image whose name starts with an image's name (the reference matches by prefix)
    m = np.zeros((H, W), np.uint8)
    m[26:44, 4:30] = 255
    save_png_gray(os.path.join(work, "proj", "00010_9.png"), m)
    fixture["proj_00010_9"] = m > 0
    fixture["proj_files"] = np.asarray(sorted(os.listdir(os.path.join(work, "proj"))))
    fixture["images"] = np.asarray(imgs)
    subprocess.check_call([CONDA, "-c", STAGE2, REF, work])
    for img in imgs:
        fixture[f"out_{img}"] = np.load(os.path.join(work, "out", f"{img}.npy"))
        fixture[f"hdf5_{img}"] = np.load(os.path.join(work, "out", f"{img}.hdf5.npy"))
    np.savez_compressed(os.path.join(HERE, "match_seg.npz"), **fixture)
    print({k: (v.shape, str(v.dtype)) for k, v in fixture.items() if k.startswith("out_")})
    for img in imgs:
        print(img, np.unique(fixture[f"out_{img}"], return_counts=True))


if __name__ == "__main__":
    main()
