"""Multiresolution hash-grid encoding (oracle; test infrastructure only).

Follows SURVEY.md section 8a rows a6-a8 and Appendix A.1 "Hash grid"
(upstream ``gridencoder.GridEncoder`` / ``kernel_grid`` / ``kernel_grid_backward``
of the un-vendored submodule pinned at /root/reference/README.md:27,59; the
hash is the Instant-NGP spatial hash, Mueller et al. 2022, eq. 4).
Parity unpinned - see ``oracle/__init__``.

Canonical choices (documented in DESIGN.md):
* the per-level (scale, resolution, offset, hashed) table is computed on the
  host (``level_table``) - float64 maths rounded once to float32 - and handed
  to both the oracle and the HIP kernels;
* corner order c = 0..7 with bit d of c selecting the +1 neighbour on axis d;
  the corner weight is ((wx * wy) * wz); features accumulate in corner order.
"""
import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


def level_table(num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                desired_resolution=2048, input_dim=3):
    """Host-side level table.

    per_level_scale = 2^(log2(desired/base)/(L-1));  resolution_l =
    ceil(base * pls^l);  rows_l = min(2^log2_hashmap_size, (resolution_l+1)^3)
    rounded up to a multiple of 8;  offsets = exclusive scan of rows.
    scale_l = float32(base * pls^l - 1) and grid_res_l = ceil(scale_l) + 1 are
    what the interpolation uses (== resolution_l).  A level is hashed iff
    (grid_res_l + 1)^3 > rows_l.
    """
    pls = np.exp2(np.log2(desired_resolution / base_resolution) / max(num_levels - 1, 1))
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
    return dict(num_levels=num_levels, level_dim=level_dim,
                per_level_scale=float(pls),
                offsets=np.asarray(offsets, dtype=np.uint32),
                scales=np.asarray(scales, dtype=np.float32),
                resolutions=np.asarray(ress, dtype=np.uint32),
                hashed=np.asarray(hashed, dtype=np.uint32),
                total_rows=int(offsets[-1]))


def corner_indices_weights(x, bound, table):
    """x f32[M,3] in [-bound,bound] -> (idx i64[M,L,8] absolute row, w f32[M,L,8]).

    torch implementation (fp32 arithmetic, int64 index maths masked to uint32
    where the hash needs wraparound).

    Out of range (upstream's ``flag_oob``, SURVEY Appendix A.1 "Later upstream versions zero the output
    for inputs outside [0,1]"): a sample whose normalised coordinate leaves [0,1] on any axis (NaN
    included) gets weight 0 on every corner of every level - zero features, no table gradient; its
    indices are those of x01 = 0 (any valid row would do).
    """
    x = torch.as_tensor(x, dtype=torch.float32)
    M = x.shape[0]
    L = table["num_levels"]
    b = torch.tensor(float(bound), dtype=torch.float32)
    x01 = (x + b) / (2 * b)
    oob = ~((x01 >= 0) & (x01 <= 1)).all(-1)
    x01 = torch.where(oob[:, None], torch.zeros_like(x01), x01)
    idx_all = torch.empty((M, L, 8), dtype=torch.int64)
    w_all = torch.empty((M, L, 8), dtype=torch.float32)
    for l in range(L):
        scale = torch.tensor(float(table["scales"][l]), dtype=torch.float32)
        res = int(table["resolutions"][l])
        off = int(table["offsets"][l])
        rows = int(table["offsets"][l + 1]) - off
        pos = x01 * scale + 0.5
        pg = torch.floor(pos)
        fr = pos - pg
        pg = pg.to(torch.int64)
        for c in range(8):
            cx = pg[:, 0] + ((c >> 0) & 1)
            cy = pg[:, 1] + ((c >> 1) & 1)
            cz = pg[:, 2] + ((c >> 2) & 1)
            wx = fr[:, 0] if (c >> 0) & 1 else 1 - fr[:, 0]
            wy = fr[:, 1] if (c >> 1) & 1 else 1 - fr[:, 1]
            wz = fr[:, 2] if (c >> 2) & 1 else 1 - fr[:, 2]
            if table["hashed"][l]:
                h = ((cx * PRIMES[0]) & 0xFFFFFFFF) ^ ((cy * PRIMES[1]) & 0xFFFFFFFF) \
                    ^ ((cz * PRIMES[2]) & 0xFFFFFFFF)
                i = h % rows
            else:
                s = res + 1
                i = (cx + cy * s + cz * s * s) % rows
            idx_all[:, l, c] = i + off
            w_all[:, l, c] = (wx * wy) * wz
    w_all[oob] = 0.0
    return idx_all, w_all


def encode(x, embeddings, bound, table):
    """x f32[M,3], embeddings f32[T,F] -> f32[M, L*F] (level-major features).

    Differentiable w.r.t. ``embeddings`` through torch autograd (the backward
    is the scatter-add of row a8).
    """
    idx, w = corner_indices_weights(x.detach() if torch.is_tensor(x) else x, bound, table)
    M, L, _ = idx.shape
    F = embeddings.shape[1]
    out = torch.zeros((M, L, F), dtype=torch.float32)
    for c in range(8):                      # accumulate in corner order
        out = out + w[:, :, c, None] * embeddings[idx[:, :, c]]
    return out.reshape(M, L * F)


def encode_backward_table(x, grad_out, bound, table):
    """Analytic table gradient: grad f32[T,F] = scatter_add(w * grad_out)."""
    idx, w = corner_indices_weights(x, bound, table)
    M, L, _ = idx.shape
    F = table["level_dim"]
    g = torch.as_tensor(grad_out, dtype=torch.float32).reshape(M, L, 1, F)
    contrib = (w[..., None] * g).reshape(-1, F)
    grad = torch.zeros((table["total_rows"], F), dtype=torch.float32)
    grad.index_add_(0, idx.reshape(-1), contrib)
    return grad


def encode_input_grad(x, grad_out, embeddings, bound, table):
    """d(sum(encode(x) * grad_out)) / dx, f32[M,3]: the a.e. derivative of the trilinear interpolation (upstream's
    ``dy_dx`` path of gridencoder, taken when the positions require grad).  Indices come from the detached positions,
    the weights are rebuilt from the fractional parts with autograd on."""
    x = torch.as_tensor(x, dtype=torch.float32).clone().requires_grad_(True)
    idx, _ = corner_indices_weights(x.detach(), bound, table)
    b = torch.tensor(float(bound), dtype=torch.float32)
    x01 = (x + b) / (2 * b)
    oob = ~((x01 >= 0) & (x01 <= 1)).all(-1)
    M, L = x.shape[0], table["num_levels"]
    F = embeddings.shape[1]
    out = torch.zeros((M, L, F), dtype=torch.float32)
    for l in range(L):
        pos = x01 * float(table["scales"][l]) + 0.5
        fr = pos - torch.floor(pos).detach()
        for c in range(8):
            wx = fr[:, 0] if c & 1 else 1 - fr[:, 0]
            wy = fr[:, 1] if c & 2 else 1 - fr[:, 1]
            wz = fr[:, 2] if c & 4 else 1 - fr[:, 2]
            w = torch.where(oob, torch.zeros(()), (wx * wy) * wz)
            out[:, l] = out[:, l] + w[:, None] * embeddings.detach()[idx[:, l, c]]
    (out.reshape(M, L * F) * torch.as_tensor(grad_out, dtype=torch.float32)).sum().backward()
    return x.grad


def fx_next_scale(old_ref, step_max, headroom=128.0):
    """Scale rule of the fixed-point (int32) table-gradient scatter (include/inr.h ``inr_grid_fx_update``; no upstream
    counterpart - upstream's ``grid_encode_backward`` sums with fp32 ``atomicAdd``): per level, from the largest |row
    gradient| of the step that has just finished and the running reference,
        ref'  = max(step_max, 0.97 * ref)         (0 when step_max is not finite)
        scale = 2 ** floor(log2(2**30 / (headroom * ref')))  clamped to 2**+-100      (0 when ref' == 0)
    all in float32.  -> (scale, ref') as float32 arrays."""
    old_ref = np.asarray(old_ref, dtype=np.float32)
    m = np.asarray(step_max, dtype=np.float32)
    finite = np.isfinite(m)
    ref = np.where(finite, np.maximum(m, np.float32(0.97) * old_ref), np.float32(0)).astype(np.float32)
    with np.errstate(divide="ignore", over="ignore", invalid="ignore"):
        e = np.floor(np.log2(np.float32(1073741824.0) / (np.float32(headroom) * ref).astype(np.float32)).astype(np.float32))
    e = np.clip(e, -100.0, 100.0)
    scale = np.where((ref > 0) & finite, np.exp2(e.astype(np.float64)), 0.0).astype(np.float32)
    return scale, ref


def fx_quantise(grad, scale):
    """What a level's gradient looks like after the int32 sum when EVERY contribution is rounded once at the end (the
    kernel rounds each wave's run sums; the difference is below one quantum per contributing wave): round(g * scale) /
    scale.  Used by tests as the reference point between the fp32 sum and the kernel's result."""
    g = np.asarray(grad, dtype=np.float64)
    return (np.rint(g * float(scale)) / float(scale)).astype(np.float32)
