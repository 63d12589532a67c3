"""Regenerates the golden vectors under tests/golden/ from the CPU oracle.

PARITY UNPINNED: the reference renderer is an un-vendored submodule
(/root/reference/.gitmodules:4-6, README.md:27,59) and ships no vectors, so
these fixtures pin *this repository's oracle* (regression + GPU comparison
data), not the reference.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from instance_nerf_amd.scene import RoomScene          # noqa: E402
from oracle import composite, field, hashgrid, march, rays, render, sh  # noqa: E402
from conftest import scene_rays                         # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    room = RoomScene()
    bits = room.density_bitfield(128, 1.0)
    table = hashgrid.level_table()
    p = field.init_params(seed=0, table=table, table_std=1.0, K=16)
    aabb = np.asarray([-1, -1, -1, 1, 1, 1], np.float32)

    # ---- marching ---------------------------------------------------------
    ro, rd = scene_rays(room, n=64, seed=21)
    nears, fars = rays.near_far_from_aabb(ro, rd, aabb, 0.05)
    noises = np.random.default_rng(22).random(64).astype(np.float32)
    out = dict(rays_o=ro, rays_d=rd, nears=nears, fars=fars, noises=noises,
               bitfield_sha256=digest(bits))
    for tag, g in (("g0", 0.0), ("g1", 1.0 / 128)):
        m = march.march_rays_train(ro, rd, bits, 1.0, 1, 128, nears, fars, noises, g, 1024)
        out.update({f"{tag}_rays": m["rays"], f"{tag}_xyzs": m["xyzs"], f"{tag}_deltas": m["deltas"]})
    np.savez_compressed(os.path.join(OUT, "march.npz"), **out)

    # ---- field ------------------------------------------------------------
    g = torch.Generator().manual_seed(31)
    x = (torch.rand(256, 3, generator=g) * 2 - 1)
    x[:4] = torch.tensor([[-1., -1, -1], [1, 1, 1], [0, 0, 0], [1, -1, 0.25]])
    d = torch.randn(256, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    with torch.no_grad():
        enc = hashgrid.encode(x, p["embeddings"], 1.0, table)
        den = field.density(x, p, 1.0, table)
        rgb = field.color(d, den["geo_feat"], p)
        logits = field.instance_logits(x, p, 1.0, table)
        shv = sh.sh_encode(d)
    np.savez_compressed(os.path.join(OUT, "field.npz"), x=x.numpy(), d=d.numpy(), enc=enc.numpy(),
                        sigma=den["sigma"].numpy(), geo=den["geo_feat"].numpy(), rgb=rgb.numpy(),
                        logits=logits.numpy(), sh=shv.numpy(),
                        emb_sha256=digest(p["embeddings"].numpy()),
                        sigma_w0_sha256=digest(p["sigma_w0"].numpy()))

    # ---- compositing ------------------------------------------------------
    rng = np.random.default_rng(41)
    N = 40
    cnt = rng.integers(0, 48, size=N)
    cnt[5] = 0
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    M = int(cnt.sum())
    rr = np.stack([rng.permutation(N), off, cnt], -1).astype(np.int32)
    sig = (rng.random(M) * 40).astype(np.float32)
    col = rng.random((M, 3)).astype(np.float32)
    ext = rng.normal(size=(M, 16)).astype(np.float32)
    dl = np.stack([np.full(M, 0.0034), rng.random(M) * 0.02 + 0.0034], -1).astype(np.float32)
    s = torch.tensor(sig, requires_grad=True)
    c = torch.tensor(col, requires_grad=True)
    e = torch.tensor(ext, requires_grad=True)
    o = composite.composite_rays_train(s, c, dl, rr, 1e-4, extra=e)
    gws = rng.normal(size=N).astype(np.float32)
    gim = rng.normal(size=(N, 3)).astype(np.float32)
    gex = rng.normal(size=(N, 16)).astype(np.float32)
    ((o["weights_sum"] * torch.tensor(gws)).sum() + (o["image"] * torch.tensor(gim)).sum()
     + (o["extra"] * torch.tensor(gex)).sum()).backward()
    np.savez_compressed(os.path.join(OUT, "composite.npz"), rays=rr, sigmas=sig, rgbs=col, extra=ext, deltas=dl,
                        weights_sum=o["weights_sum"].detach().numpy(), depth=o["depth"].detach().numpy(),
                        image=o["image"].detach().numpy(), extra_out=o["extra"].detach().numpy(),
                        g_ws=gws, g_img=gim, g_extra=gex, grad_sigmas=s.grad.numpy(), grad_rgbs=c.grad.numpy(),
                        grad_extra=e.grad.numpy())

    # ---- full render ------------------------------------------------------
    ro, rd = scene_rays(room, n=64, seed=51)
    a = render.render_train(ro, rd, p, table, bits, min_near=0.05, with_instance=True)
    b = render.render_infer(ro, rd, p, table, bits, min_near=0.05, with_instance=True)
    np.savez_compressed(os.path.join(OUT, "render.npz"), rays_o=ro, rays_d=rd,
                        train_image=a["image"].detach().numpy(), train_ws=a["weights_sum"].detach().numpy(),
                        train_depth=a["depth"].detach().numpy(), train_instance=a["instance"].detach().numpy(),
                        train_total=a["total"], infer_image=b["image"], infer_ws=b["weights_sum"],
                        infer_depth=b["depth"], infer_instance=b["instance"], infer_evaluated=b["evaluated"])
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
