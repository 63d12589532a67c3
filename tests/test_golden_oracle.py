"""The oracle reproduces the committed golden vectors (CPU).

Parity unpinned (no reference vectors exist): these pin the
oracle itself so that GPU results and future oracle edits are compared with a
fixed set of numbers.
"""
import hashlib
import os

import numpy as np
import torch

from oracle import composite, field, hashgrid, march, render, sh

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_golden_march(room_bitfield):
    g = np.load(os.path.join(G, "march.npz"))
    assert str(g["bitfield_sha256"]) == _sha(room_bitfield)
    for tag, gam in (("g0", 0.0), ("g1", 1.0 / 128)):
        m = march.march_rays_train(g["rays_o"], g["rays_d"], room_bitfield, 1.0, 1, 128, g["nears"], g["fars"],
                                   g["noises"], gam, 1024)
        assert (m["rays"] == g[f"{tag}_rays"]).all()
        assert (m["xyzs"] == g[f"{tag}_xyzs"]).all()          # bit exact
        assert (m["deltas"] == g[f"{tag}_deltas"]).all()


def test_golden_field(level_table, params_k16):
    g = np.load(os.path.join(G, "field.npz"))
    p = params_k16
    assert str(g["emb_sha256"]) == _sha(p["embeddings"].numpy())
    x, d = torch.from_numpy(g["x"]), torch.from_numpy(g["d"])
    with torch.no_grad():
        assert np.allclose(hashgrid.encode(x, p["embeddings"], 1.0, level_table).numpy(), g["enc"], atol=1e-6)
        den = field.density(x, p, 1.0, level_table)
        assert np.allclose(den["sigma"].numpy(), g["sigma"], rtol=1e-5, atol=1e-6)
        assert np.allclose(field.color(d, den["geo_feat"], p).numpy(), g["rgb"], atol=1e-6)
        assert np.allclose(field.instance_logits(x, p, 1.0, level_table).numpy(), g["logits"], atol=1e-5)
        assert np.allclose(sh.sh_encode(d).numpy(), g["sh"], atol=1e-6)


def test_golden_composite():
    g = np.load(os.path.join(G, "composite.npz"))
    o = composite.composite_rays_train(g["sigmas"], g["rgbs"], g["deltas"], g["rays"], 1e-4, extra=g["extra"])
    assert np.allclose(o["weights_sum"].numpy(), g["weights_sum"], atol=1e-6)
    assert np.allclose(o["image"].numpy(), g["image"], atol=1e-6)
    assert np.allclose(o["depth"].numpy(), g["depth"], atol=1e-6)
    assert np.allclose(o["extra"].numpy(), g["extra_out"], atol=1e-5)
    gs, gc = composite.composite_backward_analytic(g["g_ws"], g["g_img"], g["sigmas"], g["rgbs"], g["deltas"],
                                                   g["rays"], g["weights_sum"], g["image"])
    assert np.allclose(gs, g["grad_sigmas"], atol=2e-5, rtol=1e-4)
    assert np.allclose(gc, g["grad_rgbs"], atol=1e-6)


def test_golden_render(room_bitfield, level_table, params_k16):
    g = np.load(os.path.join(G, "render.npz"))
    a = render.render_train(g["rays_o"], g["rays_d"], params_k16, level_table, room_bitfield, min_near=0.05,
                            with_instance=True)
    assert a["total"] == int(g["train_total"])
    assert np.allclose(a["image"].detach().numpy(), g["train_image"], atol=1e-5)
    assert np.allclose(a["instance"].detach().numpy(), g["train_instance"], atol=1e-4)
