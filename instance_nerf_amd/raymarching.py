"""Host-side mirror of torch-ngp's ``raymarching`` extension module over the C ABI.

Same function names, argument meaning and return shapes as the upstream module
the reference's (un-vendored) ``instance_nerf`` submodule ships
(/root/reference/.gitmodules:4-6, README.md:27,59), so
``NeRFRenderer`` code written against it runs unchanged.  Every function is a
thin wrapper: validate -> allocate outputs -> one or a few launches on the
current torch stream.  No CPU fallback.

Differences from upstream, all deliberate:
* ``march_rays_train`` assigns sample slots by an exclusive scan in ray order
  (deterministic) instead of racing atomics; ``rays[:,0]`` is therefore always
  ``arange(N)``;
* ``composite_rays_train`` optionally composites K extra channels (the fork's
  instance logits) with the weights detached.
"""
import math

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

F32, I32, U8 = torch.float32, torch.int32, torch.uint8

# Step candidates per ray whose hit/miss pattern the count pass records as a bit mask (0 = march twice): the
# write pass then regenerates the candidate sequence t_{i+1} = t_i + dt(t_i) and emits at the set bits instead
# of walking the occupancy grid again; rays needing more candidates are re-marched.  1024 covers every ray of a
# bound-1 scene (far - near <= 2 sqrt(3), dt >= 2 sqrt(3) / 1024); 4 bytes of scratch per ray and 32 candidates.
# Training batches of <= 32768 rays use the wave-per-ray marcher and never capture.
SAMPLE_CAP = int(__import__("os").environ.get("INR_SAMPLE_CAP", "1024"))
SAMPLE_CAP_TRAIN = int(__import__("os").environ.get("INR_SAMPLE_CAP_TRAIN", "0"))


def _f(t):
    return t.contiguous().float() if (t.dtype != F32 or not t.is_contiguous()) else t


def set_overlap_placement(on):
    """Placement of the eval field kernel and the march kernels when they run on two streams at once (FramePipeline):
    see inr_set_overlap_placement in include/inr.h.  Process-wide; results do not change."""
    check(_lib.load().inr_set_overlap_placement(1 if on else 0), "set_overlap_placement")


def set_march_mode(mode):
    """Which training marcher runs: None / "auto" (by batch size), "lane_per_ray" or "wave_per_ray" - see
    inr_set_march_mode in include/inr.h.  Both produce the same bits; process-wide."""
    code = {None: -1, "auto": -1, "lane_per_ray": 0, "wave_per_ray": 1}[mode]
    check(_lib.load().inr_set_march_mode(code), "set_march_mode")


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2, out=None, skip_labels=None, ignore_index=-1):
    """rays_o, rays_d [N,3]; aabb [6] -> nears, fars [N] (written into ``out = (nears, fars)`` when given).
    skip_labels (int64 [N]): rays labelled ``ignore_index`` come back as misses - the march gives them no samples."""
    lib = _lib.load()
    rays_o, rays_d, aabb = _f(rays_o).view(-1, 3), _f(rays_d).view(-1, 3), _f(aabb)
    N = rays_o.shape[0]
    if out is not None:
        nears, fars = out
    else:
        nears = torch.empty(N, dtype=F32, device=rays_o.device)
        fars = torch.empty(N, dtype=F32, device=rays_o.device)
    if skip_labels is not None:
        labels = skip_labels.reshape(-1).contiguous().long()      # (int32 label maps, e.g. match_seg's .npy files, are accepted)
        if labels.shape[0] != N:
            raise RuntimeError(f"near_far_from_aabb: {labels.shape[0]} labels for {N} rays")
        check(lib.inr_near_far_from_aabb_skip(ptr(rays_o, F32, "rays_o"), ptr(rays_d, F32, "rays_d"), ptr(aabb, F32, "aabb"),
                                              N, float(min_near), ptr(labels, torch.int64, "skip_labels"),
                                              int(ignore_index), ptr(nears), ptr(fars), stream_ptr()),
              "near_far_from_aabb_skip")
        return nears, fars
    check(lib.inr_near_far_from_aabb(ptr(rays_o, F32, "rays_o"), ptr(rays_d, F32, "rays_d"), ptr(aabb, F32, "aabb"),
                                     N, float(min_near), ptr(nears), ptr(fars), stream_ptr()), "near_far_from_aabb")
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    """rays_o, rays_d [N,3], radius -> coords [N,2] in [-1,1]: where each ray leaves the sphere of that radius, as
    (2 theta / pi - 1, phi / pi) with y the up axis - upstream's ``raymarching.sph_from_ray`` [U], the input of its
    background model (``bg_radius > 0``).  The background model itself is outside the hot path, so this
    helper is plain tensor ops; it exists because callers written against upstream import the name."""
    o, d = rays_o.float(), rays_d.float()
    A = (d * d).sum(-1)
    B = (o * d).sum(-1)                                   # in fact B / 2
    C = (o * o).sum(-1) - float(radius) ** 2
    t = (-B + torch.sqrt(B * B - A * C)) / A              # the larger root: the exit point
    p = o + t.unsqueeze(-1) * d
    theta = torch.atan2(torch.sqrt(p[..., 0] ** 2 + p[..., 2] ** 2), p[..., 1])
    phi = torch.atan2(p[..., 2], p[..., 0])
    return torch.stack([2 * theta / math.pi - 1, phi / math.pi], -1)


def morton3D(coords):
    """coords int32 [N,3] (each < 1024) -> int32 [N]."""
    lib = _lib.load()
    coords = coords.int().contiguous()
    N = coords.shape[0]
    out = torch.empty(N, dtype=I32, device=coords.device)
    check(lib.inr_morton3D(ptr(coords, I32, "coords"), N, ptr(out), stream_ptr()), "morton3D")
    return out


def morton3D_invert(indices):
    """int32 [N] -> int32 [N,3]."""
    lib = _lib.load()
    indices = indices.int().contiguous()
    N = indices.shape[0]
    out = torch.empty(N, 3, dtype=I32, device=indices.device)
    check(lib.inr_morton3D_invert(ptr(indices, I32, "indices"), N, ptr(out), stream_ptr()), "morton3D_invert")
    return out


def packbits(grid, thresh, bitfield=None):
    """grid f32 [C, H^3] (or flat) -> uint8 [C*H^3/8]; bit i of byte k = grid[8k+i] > thresh."""
    lib = _lib.load()
    grid = _f(grid)
    n = grid.numel()
    if n % 8:
        raise RuntimeError("packbits: number of cells must be a multiple of 8")
    if bitfield is None:
        bitfield = torch.empty(n // 8, dtype=U8, device=grid.device)
    check(lib.inr_packbits(ptr(grid, F32, "grid"), n // 8, float(thresh), ptr(bitfield, U8, "bitfield"),
                           stream_ptr()), "packbits")
    return bitfield


def march_train_buffers(n_rays, n_samples, device, shade=False):
    """Persistent buffers for ``march_rays_train(..., out=)``: a caller that marches on a side stream every step
    (``NeRFRenderer.march_ahead``) must not allocate there - the caching allocator keeps one pool per stream, and blocks
    handed across streams cost ~0.2 ms of host time per step in deferred frees.  ``shade``: also the outputs of the
    frozen field and of the compositing forward (``march_ahead(shade=True)``)."""
    lib = _lib.load()
    b = {"n_rays": n_rays, "n_samples": n_samples,
         "nears": torch.empty(n_rays, dtype=F32, device=device), "fars": torch.empty(n_rays, dtype=F32, device=device),
         "noises": torch.empty(n_rays, dtype=F32, device=device), "rays": torch.empty(n_rays, 3, dtype=I32, device=device),
         "ws": torch.empty(lib.inr_march_workspace_bytes(n_rays, SAMPLE_CAP_TRAIN) // 8 + 1, dtype=torch.int64, device=device),
         "buf": torch.empty(n_samples * 8, dtype=F32, device=device)}
    if shade:
        b.update({"sigmas": torch.empty(n_samples, dtype=F32, device=device),
                  "rgbs": torch.empty(n_samples, 3, dtype=F32, device=device),
                  "weights_sum": torch.empty(n_rays, dtype=F32, device=device),
                  "depth": torch.empty(n_rays, dtype=F32, device=device),
                  "image": torch.empty(n_rays, 3, dtype=F32, device=device),
                  "weights": torch.empty(n_samples, dtype=F32, device=device),
                  "sample_ray": torch.empty(n_samples, dtype=I32, device=device)})
    return b


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, step_counter=None,
                     mean_count=-1, perturb=False, align=-1, force_all_rays=False, dt_gamma=0, max_steps=1024,
                     noises=None, separate_buffers=False, out=None):
    """-> xyzs [M,3], dirs [M,3], deltas [M,2], rays int32 [N,3] = (ray, offset, count).

    With ``mean_count <= 0`` or ``force_all_rays`` the exact sample count is read
    back (one 4-byte device->host copy, as upstream) and M is exact.  Otherwise M
    = mean_count (aligned) and rays overflowing M are dropped; no host sync.
    ``step_counter`` (int32 [2]) receives (total samples, N).
    """
    lib = _lib.load()
    rays_o, rays_d = _f(rays_o).view(-1, 3), _f(rays_d).view(-1, 3)
    dev = rays_o.device
    N = rays_o.shape[0]
    nears, fars = _f(nears), _f(fars)
    if noises is None:
        if not perturb:
            noises = None
        elif out is not None:
            noises = torch.rand(N, dtype=F32, device=dev, out=out["noises"])
        else:
            noises = torch.rand(N, dtype=F32, device=dev)
    else:
        noises = _f(noises)
    if step_counter is None:
        step_counter = torch.zeros(2, dtype=I32, device=dev)
    cap = SAMPLE_CAP_TRAIN
    if out is not None:                       # ``march_train_buffers``: nothing is allocated (steady state only)
        if force_all_rays or mean_count <= 0 or out["n_rays"] != N or out["n_samples"] < int(mean_count):
            raise RuntimeError("march_rays_train(out=) needs the steady state and buffers of the batch's size")
        rays, ws = out["rays"], out["ws"]
    else:
        rays = torch.empty(N, 3, dtype=I32, device=dev)
        ws = torch.empty(lib.inr_march_workspace_bytes(N, cap) // 8 + 1, dtype=torch.int64, device=dev)
    args = (ptr(rays_o, F32, "rays_o"), ptr(rays_d, F32, "rays_d"), ptr(density_bitfield, U8, "density_bitfield"),
            float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H))
    check(lib.inr_march_rays_train_count(*args, ptr(nears, F32, "nears"), ptr(fars, F32, "fars"),
                                         ptr(noises, F32, "noises", allow_none=True), ptr(rays),
                                         ptr(step_counter, I32, "step_counter"), ptr(ws), cap, stream_ptr()),
          "march_rays_train_count")
    if force_all_rays or mean_count <= 0:
        M = int(step_counter[0].item())
    else:
        M = int(mean_count)
    if align > 0:
        M += align - M % align if M % align else 0
    # rows no ray owns (padding, dropped rays) must be zero: the staged marcher's write pass clears them itself
    alloc = torch.empty if lib.inr_march_write_fills_unowned_rows(N, cap, int(max_steps)) else torch.zeros
    if out is not None:
        if alloc is not torch.empty or M > out["n_samples"]:
            raise RuntimeError("march_rays_train(out=) needs the staged marcher (it clears unowned rows itself)")
        buf = out["buf"]
        xyzs, dirs, deltas = buf[:M * 3].view(M, 3), buf[M * 3:M * 6].view(M, 3), buf[M * 6:M * 8].view(M, 2)
    elif separate_buffers:     # three allocations (the registered custom op may not return views of one buffer)
        xyzs, dirs, deltas = (alloc(M, 3, dtype=F32, device=dev), alloc(M, 3, dtype=F32, device=dev),
                              alloc(M, 2, dtype=F32, device=dev))
    else:
        buf = alloc(M * 8, dtype=F32, device=dev)
        xyzs, dirs, deltas = buf[:M * 3].view(M, 3), buf[M * 3:M * 6].view(M, 3), buf[M * 6:].view(M, 2)
    check(lib.inr_march_rays_train_write(*args, M, ptr(nears), ptr(fars), ptr(noises, F32, "noises", allow_none=True),
                                         ptr(rays), ptr(xyzs), ptr(dirs), ptr(deltas), ptr(ws), cap, stream_ptr()),
          "march_rays_train_write")
    return xyzs, dirs, deltas, rays


GROUP = 16   # rays per group of the patch-interleaved layout (csrc/raymarch.hip::kGroup)


_PINNED_COUNT = {}


def _read_count(counter, while_waiting=None):
    """counter[0] on the host.  The copy goes to a pinned buffer without blocking; ``while_waiting()`` may queue work
    that does not depend on the count behind it (it runs on the GPU while the host waits for the copy and then
    prepares the next launches - the one bubble of a frame), then the host waits for the copy alone."""
    key = counter.device.index
    host = _PINNED_COUNT.get(key)
    if host is None:
        host = _PINNED_COUNT[key] = torch.empty(2, dtype=I32, pin_memory=True)
    host.copy_(counter[:2], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    if while_waiting is not None:
        while_waiting()
    ev.synchronize()
    return int(host[0])


def march_rays_patch(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, dt_gamma=0, max_steps=1024,
                     noises=None, counter=None, table=False, while_waiting=None):
    """Full-frame inference march in the PATCH-INTERLEAVED layout (no upstream counterpart; it
    replaces the alive-ray loop).  Rays are grouped 16 at a time in the order given; inside a group
    all k-th samples are adjacent:  slot(r,k) = rays[g0,1] + sum_i min(c_i,k) + #{i<r: c_i>k}.

    -> xyzs [M,3], dirs [M,3], deltas [M,2], rays int32 [N,3] = (ray, offset-of-ray-in-ray-order, count);
    M is the exact sample count (one 4-byte device->host read).

    ``table=True`` is the feed of ``NeRFNetwork.forward_table``: xyzs come back NORMALISED,
    (p + bound) / (2 bound), and the second result is the int32 ray id of every sample instead of its direction.
    ``while_waiting``: called once the count pass and the read-back of M are queued (see ``_read_count``).
    """
    lib = _lib.load()
    rays_o, rays_d = _f(rays_o).view(-1, 3), _f(rays_d).view(-1, 3)
    dev = rays_o.device
    N = rays_o.shape[0]
    nears, fars = _f(nears), _f(fars)
    noises = _f(noises) if noises is not None else None
    if counter is None:
        counter = torch.zeros(2, dtype=I32, device=dev)
    rays = torch.empty(N, 3, dtype=I32, device=dev)
    cap = SAMPLE_CAP
    ws = torch.empty(lib.inr_march_workspace_bytes(N, cap) // 8 + 1, dtype=torch.int64, device=dev)
    args = (ptr(rays_o, F32, "rays_o"), ptr(rays_d, F32, "rays_d"), ptr(density_bitfield, U8, "density_bitfield"),
            float(bound), float(dt_gamma), int(max_steps), N, int(C), int(H))
    check(lib.inr_march_rays_train_count(*args, ptr(nears, F32, "nears"), ptr(fars, F32, "fars"),
                                         ptr(noises, F32, "noises", allow_none=True), ptr(rays),
                                         ptr(counter, I32, "counter"), ptr(ws), cap, stream_ptr()),
          "march_rays_train_count")
    M = _read_count(counter, while_waiting)
    xyzs = torch.empty(M, 3, dtype=F32, device=dev)       # every row is written: no memset needed
    dirs = None if table else torch.empty(M, 3, dtype=F32, device=dev)
    ray_ids = torch.empty(M, dtype=I32, device=dev) if table else None
    deltas = torch.empty(M, 2, dtype=F32, device=dev)
    check(lib.inr_march_rays_patch_write(*args, M, ptr(nears), ptr(fars), ptr(noises, F32, "noises", allow_none=True),
                                         ptr(rays), ptr(xyzs, allow_none=M == 0), ptr(dirs, allow_none=True),
                                         ptr(deltas, allow_none=M == 0), ptr(ws), cap,
                                         ptr(ray_ids, allow_none=True), 1 if table else 0, stream_ptr()),
          "march_rays_patch_write")
    return xyzs, (ray_ids if table else dirs), deltas, rays


def composite_rays_patch(sigmas, rgbs, deltas, rays, T_thresh=1e-4, extra=None, return_weights=False, skippable=None):
    """Compositing of the patch-interleaved layout (inference only, no autograd).
    -> weights_sum [N], depth [N], image [N,3] (, extra_out [N,K]) (, weights [M] when return_weights).
    ``skippable`` (int64 [1] on the device, optional) is incremented by the number of samples that lie behind the
    point where their whole 16-ray group has terminated - what the early-terminating kernel would not evaluate."""
    lib = _lib.load()
    sigmas, rgbs, deltas = _f(sigmas), _f(rgbs), _f(deltas)
    dev = rays.device
    N, M = rays.shape[0], sigmas.shape[0]
    K = 0 if extra is None else extra.shape[1]
    ws = torch.empty(N, dtype=F32, device=dev)
    depth = torch.empty(N, dtype=F32, device=dev)
    image = torch.empty(N, 3, dtype=F32, device=dev)
    extra_out = torch.empty(N, K, dtype=F32, device=dev) if K else None
    wbuf = torch.empty(max(M, 1), dtype=F32, device=dev) if (K or return_weights) else None
    none_ok = M == 0
    check(lib.inr_composite_rays_patch_forward(
        ptr(sigmas, F32, "sigmas", allow_none=none_ok), ptr(rgbs, F32, "rgbs", allow_none=none_ok),
        ptr(deltas, F32, "deltas", allow_none=none_ok), ptr(rays, I32, "rays"), N, M, float(T_thresh),
        ptr(_f(extra) if extra is not None else None, allow_none=True), K, ptr(ws), ptr(depth), ptr(image),
        ptr(extra_out, allow_none=True), ptr(wbuf, allow_none=True), ptr(skippable, torch.int64, "skippable", allow_none=True),
        stream_ptr()), "composite_rays_patch_forward")
    out = (ws, depth, image) + ((extra_out,) if K else ())
    return out + ((wbuf,) if return_weights else ())


def patch_slots(rays):
    """Reference (torch, host-side index maths) of the patch-interleaved slot map: for every ray n and
    step k < count_n the sample row.  Returns int64 [total] rows in RAY-MAJOR order, i.e.
    ``xyzs_patch[patch_slots(rays)]`` is the ray-major sample array.  Used by tests and tools."""
    r = rays.cpu().long()
    N = r.shape[0]
    cnt, off = r[:, 2], r[:, 1]
    pad = (-N) % GROUP
    c = torch.cat([cnt, torch.zeros(pad, dtype=torch.long)]).view(-1, GROUP)          # [G,16]
    base = torch.cat([off, torch.zeros(pad, dtype=torch.long)]).view(-1, GROUP)[:, 0]  # offset of the group's first ray
    out = []
    for g in range(c.shape[0]):
        cg = c[g]
        kmax = int(cg.max())
        if kmax == 0:
            continue
        k = torch.arange(kmax)[:, None]                                               # [k,1]
        act = cg[None, :] > k                                                          # [k,16]
        nact = act.sum(1)
        S = base[g] + torch.cat([torch.zeros(1, dtype=torch.long), nact.cumsum(0)[:-1]])
        rank = act.long().cumsum(1) - act.long()
        slot = S[:, None] + rank                                                       # [k,16]
        for i in range(GROUP):
            out.append(slot[: int(cg[i]), i])
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.long)


class _CompositeRaysTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas, rgbs, extra, deltas, rays, T_thresh, want_weights=False, total_dev=None):
        lib = _lib.load()
        sigmas, rgbs, deltas = _f(sigmas), _f(rgbs), _f(deltas)
        dev = sigmas.device
        N = rays.shape[0]
        K = 0 if extra is None else extra.shape[1]
        if extra is not None:
            extra = _f(extra)
        # every ray's outputs are written by the kernels (zeros for dropped rays); weight rows no ray owns are
        # never read - so nothing needs a fill
        ws = torch.empty(N, dtype=F32, device=dev)
        depth = torch.empty(N, dtype=F32, device=dev)
        image = torch.empty(N, 3, dtype=F32, device=dev)
        extra_out = torch.empty(N, K, dtype=F32, device=dev) if K else None
        wbuf = torch.empty(sigmas.shape[0], dtype=F32, device=dev) if (K or want_weights) else None
        sample_ray = torch.empty(sigmas.shape[0], dtype=I32, device=dev) if want_weights else None
        check(lib.inr_composite_rays_train_forward(
            ptr(sigmas, F32, "sigmas"), ptr(rgbs, F32, "rgbs"), ptr(deltas, F32, "deltas"), ptr(rays, I32, "rays"),
            N, sigmas.shape[0], float(T_thresh), ptr(extra, F32, "extra", allow_none=True), K, ptr(ws), ptr(depth),
            ptr(image),
            ptr(extra_out, allow_none=True), ptr(wbuf, allow_none=True), ptr(sample_ray, allow_none=True), stream_ptr()),
            "composite_rays_train_forward")
        ctx.save_for_backward(sigmas, rgbs, extra, deltas, rays, ws, image, wbuf, total_dev)
        ctx.T_thresh = T_thresh
        ctx.K = K
        ctx.set_materialize_grads(False)          # unused outputs reach backward as None, not as zero fills
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            ctx.mark_non_differentiable(depth)
        else:                                      # frozen density/colour field: only the K channels carry gradient
            ctx.mark_non_differentiable(depth, ws, image)
        out = (ws, depth, image) + ((extra_out,) if K else ())
        if want_weights:                           # per-sample weights (detached) + owning ray row, for the fused instance head
            ctx.mark_non_differentiable(wbuf, sample_ray)
            out = out + (wbuf, sample_ray)
        ctx.n_out = len(out)
        return out

    @staticmethod
    def backward(ctx, g_ws, g_depth, g_image, *rest):
        g_extra = rest[0] if ctx.K else None
        lib = _lib.load()
        sigmas, rgbs, extra, deltas, rays, ws, image, wbuf, total_dev = ctx.saved_tensors
        N = rays.shape[0]
        K = ctx.K
        dev = sigmas.device
        need_field = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        gs = gc = None
        if need_field:
            g_ws = _f(g_ws) if g_ws is not None else None
            g_image = _f(g_image) if g_image is not None else torch.zeros(N, 3, dtype=F32, device=dev)
            # with the marcher's total the launch writes every row itself (zeros where no ray owns one)
            alloc = torch.empty_like if total_dev is not None else torch.zeros_like
            gs = alloc(sigmas)
            gc = alloc(rgbs)
        ge = None
        if K and g_extra is not None and ctx.needs_input_grad[2]:
            g_extra = _f(g_extra)
            ge = torch.zeros_like(extra)        # rows no ray owns must stay zero: they flow into the field backward
        else:
            g_extra = None
        if need_field or ge is not None:
            check(lib.inr_composite_rays_train_backward(
                ptr(g_ws, allow_none=True) if need_field else None, ptr(g_image) if need_field else None,
                ptr(g_extra, allow_none=True), ptr(sigmas), ptr(rgbs),
                ptr(extra, allow_none=True), ptr(deltas), ptr(rays), ptr(ws), ptr(image), ptr(wbuf, allow_none=True), N,
                sigmas.shape[0], float(ctx.T_thresh), K, ptr(gs, allow_none=True), ptr(gc, allow_none=True),
                ptr(ge, allow_none=True), ptr(total_dev, I32, "total_dev", allow_none=True), stream_ptr()),
                "composite_rays_train_backward")
        return gs, gc, ge, None, None, None, None, None


@torch.no_grad()
def composite_rays_train_into(sigmas, rgbs, deltas, rays, T_thresh, out):
    """The forward of ``composite_rays_train(return_weights=True)`` into the buffers of ``march_train_buffers(shade=True)``
    - the same launch with the same arguments, nothing allocated, no autograd node (a frozen field: every output is
    non-differentiable anyway).  -> weights_sum [N], depth [N], image [N,3], weights [M], sample_ray int32 [M]."""
    lib = _lib.load()
    N, M = rays.shape[0], sigmas.shape[0]
    ws, depth, image = out["weights_sum"], out["depth"], out["image"]
    wbuf, sample_ray = out["weights"][:M], out["sample_ray"][:M]
    check(lib.inr_composite_rays_train_forward(
        ptr(sigmas, F32, "sigmas"), ptr(rgbs, F32, "rgbs"), ptr(deltas, F32, "deltas"), ptr(rays, I32, "rays"),
        N, M, float(T_thresh), None, 0, ptr(ws), ptr(depth), ptr(image), None, ptr(wbuf), ptr(sample_ray), stream_ptr()),
        "composite_rays_train_forward")
    return ws, depth, image, wbuf, sample_ray


def composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh=1e-4, extra=None, return_weights=False, total_dev=None):
    """-> weights_sum [N], depth [N], image [N,3] (, extra_out [N,K] when ``extra`` [M,K] is given)
    (, weights [M], sample_ray int32 [M] when ``return_weights``: the detached per-sample compositing weights and the
    output row of the ray that owns each sample - the inputs of the fused instance head).

    Differentiable w.r.t. sigmas, rgbs and extra.  The K extra channels are
    composited with the weights detached (instance field vs. a frozen NeRF).

    ``total_dev`` (int32 device tensor whose first element is the marcher's sample total - ``step_counter`` of
    ``march_rays_train``; the rays' rows must tile [0, total), as the marcher leaves them): the backward then writes
    every gradient row itself and skips the two zero fills.
    """
    return _CompositeRaysTrain.apply(sigmas, rgbs, extra, deltas, rays, T_thresh, bool(return_weights), total_dev)


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
               align=-1, perturb=False, dt_gamma=0, max_steps=1024):
    """-> xyzs, dirs [n_alive*n_step,3], deltas [n_alive*n_step,2] (rows a ray did not fill are zero)."""
    lib = _lib.load()
    rays_o, rays_d = _f(rays_o).view(-1, 3), _f(rays_d).view(-1, 3)
    dev = rays_o.device
    M = n_alive * n_step
    if align > 0 and M % align:
        M += align - M % align
    xyzs = torch.zeros(M, 3, dtype=F32, device=dev)
    dirs = torch.zeros(M, 3, dtype=F32, device=dev)
    deltas = torch.zeros(M, 2, dtype=F32, device=dev)
    check(lib.inr_march_rays(n_alive, n_step, ptr(rays_alive, I32, "rays_alive"), ptr(rays_t, F32, "rays_t"),
                             ptr(rays_o, F32, "rays_o"), ptr(rays_d, F32, "rays_d"), float(bound), float(dt_gamma),
                             int(max_steps), int(C), int(H), ptr(density_bitfield, U8, "density_bitfield"),
                             ptr(_f(near)), ptr(_f(far)), ptr(xyzs), ptr(dirs), ptr(deltas), stream_ptr()),
          "march_rays")
    return xyzs, dirs, deltas


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas, weights_sum, depth, image,
                   T_thresh=1e-4, extra=None, extra_acc=None):
    """In-place accumulate of one inference step; marks finished rays with -1 in rays_alive."""
    lib = _lib.load()
    sigmas, rgbs, deltas = _f(sigmas), _f(rgbs), _f(deltas)
    K = 0 if extra is None else extra.shape[1]
    check(lib.inr_composite_rays(n_alive, n_step, ptr(rays_alive, I32, "rays_alive"), ptr(rays_t, F32, "rays_t"),
                                 ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(weights_sum, F32, "weights_sum"),
                                 ptr(depth, F32, "depth"), ptr(image, F32, "image"), float(T_thresh),
                                 ptr(_f(extra) if extra is not None else None, allow_none=True),
                                 ptr(extra_acc, F32, "extra_acc", allow_none=True), K, stream_ptr()),
          "composite_rays")


def compact_alive(rays_alive, n_alive):
    """Order-preserving removal of dead (-1) entries.  Returns (compacted int32 [n_alive], n_out int)."""
    lib = _lib.load()
    dev = rays_alive.device
    scratch = lib.inr_march_workspace_bytes(n_alive, 0) // 4
    out = torch.empty(n_alive + scratch, dtype=I32, device=dev)
    n_out = torch.zeros(2, dtype=I32, device=dev)
    check(lib.inr_compact_alive(ptr(rays_alive, I32, "rays_alive"), n_alive, ptr(out), ptr(n_out), stream_ptr()),
          "compact_alive")
    n = int(n_out[0].item())
    return out[:n], n


FINISH_MSE_MAX_RAYS = 65536          # include/inr.h INR_FINISH_MSE_MAX_RAYS

_UNIT = {}


def unit_gradient(device):
    """A cached 0-dim float32 one on ``device``: ``loss.backward(gradient=unit_gradient(dev))`` starts the backward
    without the fill launch of the implicit ``ones_like(loss)``, and a loss node of this module that is handed this very
    tensor skips its multiplication by one (each a ~5 us launch per training step).  Never written to."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0))
    t = _UNIT.get(key)
    if t is None:
        t = _UNIT[key] = torch.ones((), dtype=F32, device=device)
    return t


def _is_unit(g):
    u = _UNIT.get((g.device.type, g.device.index if g.device.index is not None else 0))
    return u is not None and g.dim() == 0 and g.dtype == F32 and g.data_ptr() == u.data_ptr()


class _FinishRaysMSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, weights_sum, depth, nears, fars, bg3, bg_rays, target):
        lib = _lib.load()
        image, weights_sum, depth, target = _f(image), _f(weights_sum), _f(depth), _f(target).view(-1, 3)
        N = image.shape[0]
        dev = image.device
        image_out = torch.empty_like(image)
        depth_out = torch.empty_like(depth)
        grad = torch.empty(4 * N, dtype=F32, device=dev)
        loss = torch.empty((), dtype=F32, device=dev)
        if bg_rays is not None:
            bg_rays = _f(bg_rays).view(-1, 3)
            bg3 = (0.0, 0.0, 0.0)
        check(lib.inr_finish_rays_mse(ptr(image, F32, "image"), ptr(depth, F32, "depth"),
                                      ptr(weights_sum, F32, "weights_sum"), ptr(_f(nears), F32, "nears"),
                                      ptr(_f(fars), F32, "fars"), bg3[0], bg3[1], bg3[2],
                                      ptr(bg_rays, F32, "bg_rays", allow_none=True), ptr(target, F32, "target"), N,
                                      ptr(image_out), ptr(depth_out), ptr(grad), ptr(loss), stream_ptr()),
              "finish_rays_mse")
        ctx.save_for_backward(grad, bg_rays)
        ctx.bg3 = bg3
        ctx.N = N
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(depth_out)
        return image_out, depth_out, loss

    @staticmethod
    def backward(ctx, g_image, g_depth, g_loss):
        grad, bg_rays = ctx.saved_tensors
        N = ctx.N
        gi = gw = None
        if g_loss is not None:
            # one launch for both gradients ([0, 3N) image, [3N, 4N) weights_sum); none for the unit seed
            g = grad if _is_unit(g_loss) else grad * g_loss
            gi, gw = g[:3 * N].view(N, 3), g[3 * N:]
        if g_image is not None:                    # the shaded image is also used outside the loss
            bg = bg_rays if bg_rays is not None else g_image.new_tensor(ctx.bg3)
            gw_img = -(g_image.reshape(N, 3) * bg).sum(-1)
            gi = g_image.reshape(N, 3) if gi is None else gi + g_image.reshape(N, 3)
            gw = gw_img if gw is None else gw + gw_img
        return gi, gw, None, None, None, None, None, None


def finish_rays_mse(image, weights_sum, depth, nears, fars, bg_color, target):
    """The training tail of the NeRF stage as ONE autograd node and one launch each way (inr_finish_rays_mse):
    -> (image + (1 - weights_sum) * bg, clamp(depth - nears, 0) / (fars - nears), mean((shaded image - target)^2)).
    ``bg_color``: a number, a host-side triple, or a per-ray tensor [N,3] (upstream's random background for RGBA
    batches).  The loss is upstream's ``MSELoss(reduction='none')(pred, gt).mean()``; gradients flow to ``image`` and
    ``weights_sum`` (from the loss and, if the caller uses it elsewhere, from the shaded image).  N <= 65536 rays."""
    bg_rays = None
    if torch.is_tensor(bg_color):
        if bg_color.numel() == 3 * image.shape[0]:
            bg_rays, bg3 = bg_color, (0.0, 0.0, 0.0)
        else:
            bg3 = tuple(float(c) for c in bg_color.reshape(-1).expand(3).tolist())
    elif isinstance(bg_color, (int, float)):
        bg3 = (float(bg_color),) * 3
    else:
        bg3 = tuple(float(c) for c in bg_color)
    return _FinishRaysMSE.apply(image, weights_sum, depth, nears, fars, bg3, bg_rays, target)


class _CrossEntropy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        lib = _lib.load()
        logits = _f(logits)
        labels = labels.contiguous().long()
        N, K = logits.shape
        grad = torch.empty_like(logits)
        acc = torch.empty(128, dtype=F32, device=logits.device)
        loss = torch.empty((), dtype=F32, device=logits.device)
        check(lib.inr_cross_entropy(ptr(logits, F32, "logits", allow_none=N == 0),
                                    ptr(labels, torch.int64, "labels", allow_none=N == 0), N, K, int(ignore_index),
                                    ptr(grad, allow_none=N == 0), ptr(acc), ptr(loss), stream_ptr()), "cross_entropy")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def cross_entropy(logits, labels, ignore_index=-1):
    """Mean cross entropy of logits [N,K] (K <= 64) against int64 labels [N], rows with ``ignore_index`` skipped - the
    mask-supervised loss of the instance stage - value and gradient in two HIP launches instead of
    torch's log_softmax / nll_loss pairs.  Same values as ``F.cross_entropy(logits, labels, ignore_index=...)``; a label
    that is neither ``ignore_index`` nor a class - torch asserts on the device for it - makes the loss NaN here (such a
    row is never dropped silently: a detection-count mismatch must not train on fewer rows unnoticed)."""
    return _CrossEntropy.apply(logits, labels, ignore_index)
