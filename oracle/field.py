"""NeRF field and instance field (oracle; test infrastructure only).

Follows SURVEY.md section 8a rows a9, a11, a13 and Appendix A.1 "Network"
(upstream ``nerf/network.py::NeRFNetwork`` and ``activation.trunc_exp`` of the
un-vendored submodule pinned at /root/reference/README.md:27,59; the instance
head is the fork's addition, [U-fork] in the survey).  Parity unpinned.

Weights follow torch's ``nn.Linear`` layout ``[out, in]``; no biases.
    sigma_net : 32 -> 64 -> 16      (ReLU between)   sigma = trunc_exp(h[0]), geo = h[1:16]
    color_net : 31 -> 64 -> 64 -> 3 (ReLU between)   rgb = sigmoid(.) on cat(sh16(d), geo15)
    inst_net  : 32 -> 64 -> 64 -> K (ReLU between)   raw logits, position only
"""
import math

import numpy as np
import torch

from .hashgrid import encode, level_table
from .sh import sh_encode


class _TruncExp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def _kaiming_uniform(gen, out_f, in_f):
    bound = 1.0 / math.sqrt(in_f)           # nn.Linear default: U(-1/sqrt(in), 1/sqrt(in))
    return (torch.rand((out_f, in_f), generator=gen, dtype=torch.float32) * 2 - 1) * bound


def init_params(seed=0, table=None, table_std=1e-4, K=0, hidden=64, geo_feat_dim=15):
    """Random parameters.  table_std=1e-4 is upstream's U(-1e-4,1e-4) init;
    parity fixtures use table_std=1.0 so that outputs are O(1)."""
    table = table or level_table()
    gen = torch.Generator().manual_seed(seed)
    T, F = table["total_rows"], table["level_dim"]
    in_dim = table["num_levels"] * F
    p = dict(
        embeddings=(torch.rand((T, F), generator=gen) * 2 - 1) * table_std,
        sigma_w0=_kaiming_uniform(gen, hidden, in_dim),
        sigma_w1=_kaiming_uniform(gen, 1 + geo_feat_dim, hidden),
        color_w0=_kaiming_uniform(gen, hidden, 16 + geo_feat_dim),
        color_w1=_kaiming_uniform(gen, hidden, hidden),
        color_w2=_kaiming_uniform(gen, 3, hidden),
    )
    if K:
        p.update(
            inst_embeddings=(torch.rand((T, F), generator=gen) * 2 - 1) * table_std,
            inst_w0=_kaiming_uniform(gen, hidden, in_dim),
            inst_w1=_kaiming_uniform(gen, hidden, hidden),
            inst_w2=_kaiming_uniform(gen, K, hidden),
        )
    return p


def density(x, p, bound, table):
    """x f32[M,3] -> dict(sigma f32[M], geo_feat f32[M,15])."""
    enc = encode(x, p["embeddings"], bound, table)
    h = torch.relu(enc @ p["sigma_w0"].t()) @ p["sigma_w1"].t()
    return dict(sigma=trunc_exp(h[:, 0]), geo_feat=h[:, 1:], sigma_raw=h[:, 0])


def color(d, geo_feat, p):
    """d f32[M,3] unit, geo_feat f32[M,15] -> rgb f32[M,3]."""
    h = torch.cat([sh_encode(d), geo_feat], -1)
    h = torch.relu(h @ p["color_w0"].t())
    h = torch.relu(h @ p["color_w1"].t())
    return torch.sigmoid(h @ p["color_w2"].t())


def nerf_forward(x, d, p, bound, table):
    """-> sigma f32[M], rgb f32[M,3]."""
    den = density(x, p, bound, table)
    return den["sigma"], color(d, den["geo_feat"], p)


def instance_logits(x, p, bound, table):
    """x f32[M,3] -> f32[M,K] raw logits of the position-only instance field."""
    enc = encode(x, p["inst_embeddings"], bound, table)
    h = torch.relu(enc @ p["inst_w0"].t())
    h = torch.relu(h @ p["inst_w1"].t())
    return h @ p["inst_w2"].t()


def to_numpy(p):
    return {k: np.ascontiguousarray(v.detach().numpy()) for k, v in p.items()}
