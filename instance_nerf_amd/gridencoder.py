"""Host-side mirror of torch-ngp's ``gridencoder`` module over the C ABI.

``GridEncoder`` keeps upstream's constructor and ``forward(inputs, bound=1)``
signature and its parameter name (``embeddings`` [T, F], init U(-1e-4, 1e-4)) so
checkpoints of the reference's submodule (/root/reference/.gitmodules:4-6) map
one-to-one.  The level table is computed once on the host and handed to the
kernels (DESIGN.md "level table").  No CPU fallback.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, raymarching
from ._lib import check, ptr, stream_ptr

# Morton-ordering the samples before the scatter is available but OFF by default: measured on the 209k-sample
# instance-training step it costs more (argsort + gathers: 3.38 ms/step) than it saves in atomics (3.04 ms/step).
SORT_MIN_SAMPLES = int(__import__("os").environ.get("INR_BWD_SORT_MIN", str(1 << 62)))


def level_table(num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                desired_resolution=2048, input_dim=3, per_level_scale=None):
    """Per-level (offset, scale, resolution, hashed) table.

    per_level_scale = 2^(log2(desired/base)/(L-1)); resolution_l = ceil(base * pls^l);
    rows_l = min(2^log2_hashmap_size, (resolution_l + 1)^3) rounded up to a multiple of 8.
    scale_l = float32(base * pls^l - 1); a level is hashed iff (ceil(scale_l) + 2)^3 > rows_l.
    """
    if per_level_scale is None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / max(num_levels - 1, 1))
    pls = float(per_level_scale)
    max_params = 2 ** log2_hashmap_size
    offsets, scales, ress, hashed = [0], [], [], []
    for l in range(num_levels):
        res = int(np.ceil(base_resolution * pls ** l))
        rows = min(max_params, (res + 1) ** input_dim)
        rows = int(np.ceil(rows / 8) * 8)
        offsets.append(offsets[-1] + rows)
        scale = np.float32(np.exp2(l * np.log2(pls)) * base_resolution - 1.0)
        gres = int(np.ceil(scale)) + 1
        scales.append(scale)
        ress.append(gres)
        hashed.append(1 if (gres + 1) ** input_dim > rows else 0)
    return dict(num_levels=num_levels, level_dim=level_dim, per_level_scale=pls,
                offsets=np.asarray(offsets, dtype=np.uint32), scales=np.asarray(scales, dtype=np.float32),
                resolutions=np.asarray(ress, dtype=np.uint32), hashed=np.asarray(hashed, dtype=np.uint32),
                total_rows=int(offsets[-1]))


class _GridEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, desc, bound, out_dim):
        lib = _lib.load()
        inputs = inputs.contiguous().float()
        M = inputs.shape[0]
        out = torch.empty(M, out_dim, dtype=torch.float32, device=inputs.device)
        check(lib.inr_grid_encode_forward(ptr(inputs, torch.float32, "inputs"),
                                          ptr(embeddings, torch.float32, "embeddings"), desc, M, float(bound),
                                          ptr(out), stream_ptr()), "grid_encode_forward")
        ctx.save_for_backward(inputs, embeddings)
        ctx.desc, ctx.bound = desc, bound
        return out

    @staticmethod
    def backward(ctx, grad):
        lib = _lib.load()
        inputs, embeddings = ctx.saved_tensors
        grad = grad.contiguous().float()
        M = inputs.shape[0]
        g_in = None
        if ctx.needs_input_grad[0]:              # upstream's dy_dx path: positions that require grad
            g_in = torch.empty_like(inputs)
            check(lib.inr_grid_encode_backward_input(ptr(inputs), ptr(grad, torch.float32, "grad"),
                                                     ptr(embeddings.detach(), torch.float32, "embeddings"), ctx.desc, M,
                                                     float(ctx.bound), ptr(g_in), stream_ptr()), "grid_encode_backward_input")
        if not ctx.needs_input_grad[1]:
            return g_in, None, None, None, None
        g_emb = torch.zeros_like(embeddings)
        order = None
        if M >= SORT_MIN_SAMPLES:
            # Morton order of the sample positions (10 bits per axis): neighbouring lanes become neighbours
            # in space, so equal table rows merge in-wave and the remaining atomics are address-adjacent
            q = ((inputs + ctx.bound) * (1023.0 / (2.0 * ctx.bound))).clamp_(0, 1023).to(torch.int32)
            order = torch.argsort(raymarching.morton3D(q)).to(torch.int32)
        check(lib.inr_grid_encode_backward_ordered(ptr(inputs), ptr(grad, torch.float32, "grad"),
                                                   ptr(order, torch.int32, "order", allow_none=True), ctx.desc, M,
                                                   float(ctx.bound), ptr(g_emb), stream_ptr()), "grid_encode_backward")
        return g_in, g_emb, None, None, None


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False):
        super().__init__()
        if input_dim != 3 or level_dim != 2 or gridtype != "hash" or align_corners:
            raise RuntimeError("GridEncoder (HIP): only input_dim=3, level_dim=2, gridtype='hash', "
                               "align_corners=False are implemented")
        pls = None if desired_resolution is not None else per_level_scale
        self.table = level_table(num_levels, level_dim, base_resolution, log2_hashmap_size,
                                 desired_resolution or base_resolution, input_dim, per_level_scale=pls)
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale = self.table["per_level_scale"]
        self.log2_hashmap_size, self.base_resolution = log2_hashmap_size, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype, self.align_corners = gridtype, align_corners
        self.register_buffer("offsets", torch.from_numpy(self.table["offsets"].astype(np.int32)))
        self.n_params = self.table["total_rows"] * level_dim
        self.embeddings = nn.Parameter(torch.empty(self.table["total_rows"], level_dim))
        self.embeddings._is_hash_table = True        # nerf/network.py::fx_state hangs the fixed-point gradient state here
        self.desc = _lib.make_grid_desc(self.table)
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {int(self.table['resolutions'][-1])} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)}")

    def forward(self, inputs, bound=1):
        """inputs [..., 3] in [-bound, bound] -> [..., L*F]."""
        prefix = list(inputs.shape[:-1])
        out = _GridEncode.apply(inputs.reshape(-1, self.input_dim), self.embeddings, self.desc, bound,
                                self.output_dim)
        return out.view(prefix + [self.output_dim])
