"""Registered PyTorch custom ops over the C ABI: ``torch.ops.inr.*``.

BASELINE.json's north star asks for "PyTorch-ROCm custom ops over a thin C-ABI".  The modules of this package
(``raymarching``, ``gridencoder``, ``nerf.network`` ...) call the ABI through ``ctypes`` from ``autograd.Function``s;
this module ALSO registers the extension-level entry points with the dispatcher (``torch.library.custom_op``: schema,
fake/meta implementation, autograd formula), so that they are visible as ``torch.ops.inr.<name>``, appear by name in
profiler traces, and can sit inside a ``torch.compile``d region without a graph break.  ctypes stays the transport:
every op body is one or two calls into libinr_hip.so on the current stream; there is no second implementation.

Ops (names follow upstream's extension modules):
  inr::near_far_from_aabb       raymarching.near_far_from_aabb                          (a2)
  inr::march_rays_train         raymarching.march_rays_train (count + scan + write)     (a4)
  inr::composite_rays_train     raymarching.composite_rays_train, differentiable        (a12)
  inr::grid_encode              gridencoder forward, differentiable w.r.t. the table    (a7 / a8)
  inr::nerf_forward             NeRFNetwork.forward without autograd (fused kernel)     (a9, a10, a11)
  inr::roi_align_3d             roi_align.roi_align.roi_align_3d, differentiable w.r.t. input  (f2; the reference's one
                                in-tree FFI call, /root/reference/nerf_rcnn/model/utils.py:608)
The level table of a hash grid travels as plain integer / float lists (``GridEncoder.table``): a custom-op schema
knows tensors, numbers and lists of numbers, not ctypes structures.
"""
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib, raymarching
from ._lib import check, ptr, stream_ptr

F32, I32 = torch.float32, torch.int32


def table_args(table):
    """``GridEncoder.table`` -> the (offsets, scales, resolutions, hashed) lists the ops take."""
    return ([int(v) for v in table["offsets"]], [float(v) for v in table["scales"]],
            [int(v) for v in table["resolutions"]], [int(v) for v in table["hashed"]])


def _desc(offsets, scales, resolutions, hashed):
    return _lib.make_grid_desc(dict(num_levels=len(scales), level_dim=2, offsets=np.asarray(offsets, np.uint32),
                                    scales=np.asarray(scales, np.float32), resolutions=np.asarray(resolutions, np.uint32),
                                    hashed=np.asarray(hashed, np.uint32)))


# ------------------------------------------------------------------------------------------------ rays
@torch.library.custom_op("inr::near_far_from_aabb", mutates_args=())
def near_far_from_aabb(rays_o: Tensor, rays_d: Tensor, aabb: Tensor, min_near: float) -> Tuple[Tensor, Tensor]:
    return raymarching.near_far_from_aabb(rays_o, rays_d, aabb, min_near)


@near_far_from_aabb.register_fake
def _(rays_o, rays_d, aabb, min_near):
    n = rays_o.reshape(-1, 3).shape[0]
    return rays_o.new_empty(n, dtype=F32), rays_o.new_empty(n, dtype=F32)


@torch.library.custom_op("inr::march_rays_train", mutates_args=())
def march_rays_train(rays_o: Tensor, rays_d: Tensor, bound: float, density_bitfield: Tensor, cascade: int, grid_size: int,
                     nears: Tensor, fars: Tensor, noises: Optional[Tensor], dt_gamma: float, max_steps: int,
                     num_samples: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> xyzs [M,3], dirs [M,3], deltas [M,2], rays int32 [N,3], counter int32 [2] = (total samples, N).
    num_samples > 0: M = num_samples (upstream's mean_count sizing: rays that overflow are dropped, no host sync);
    num_samples <= 0: M = the exact total (one 4-byte read-back)."""
    counter = torch.zeros(2, dtype=I32, device=rays_o.device)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(
        rays_o, rays_d, bound, density_bitfield, cascade, grid_size, nears, fars, counter, num_samples, False, -1,
        num_samples <= 0, dt_gamma, max_steps, noises=noises, separate_buffers=True)
    return xyzs, dirs, deltas, rays, counter


@march_rays_train.register_fake
def _(rays_o, rays_d, bound, density_bitfield, cascade, grid_size, nears, fars, noises, dt_gamma, max_steps, num_samples):
    n = rays_o.reshape(-1, 3).shape[0]
    m = num_samples if num_samples > 0 else torch.library.get_ctx().new_dynamic_size()
    f = lambda *s: rays_o.new_empty(*s, dtype=F32)
    return f(m, 3), f(m, 3), f(m, 2), rays_o.new_empty(n, 3, dtype=I32), rays_o.new_empty(2, dtype=I32)


# ------------------------------------------------------------------------------------------------ compositing
@torch.library.custom_op("inr::composite_rays_train", mutates_args=())
def composite_rays_train(sigmas: Tensor, rgbs: Tensor, deltas: Tensor, rays: Tensor,
                         T_thresh: float) -> Tuple[Tensor, Tensor, Tensor]:
    """-> weights_sum [N], depth [N], image [N,3] (depth carries no gradient, as upstream)."""
    lib = _lib.load()
    sigmas, rgbs, deltas = (t.contiguous().float() for t in (sigmas, rgbs, deltas))
    N, M, dev = rays.shape[0], sigmas.shape[0], sigmas.device
    ws, depth, image = (torch.empty(N, dtype=F32, device=dev), torch.empty(N, dtype=F32, device=dev),
                        torch.empty(N, 3, dtype=F32, device=dev))
    none_ok = M == 0
    check(lib.inr_composite_rays_train_forward(ptr(sigmas, allow_none=none_ok), ptr(rgbs, allow_none=none_ok),
                                               ptr(deltas, allow_none=none_ok), ptr(rays, I32, "rays"), N, M, float(T_thresh),
                                               None, 0, ptr(ws), ptr(depth), ptr(image), None, None, None, stream_ptr()),
          "composite_rays_train_forward")
    return ws, depth, image


@composite_rays_train.register_fake
def _(sigmas, rgbs, deltas, rays, T_thresh):
    n = rays.shape[0]
    return sigmas.new_empty(n), sigmas.new_empty(n), sigmas.new_empty(n, 3)


@torch.library.custom_op("inr::composite_rays_train_backward", mutates_args=())
def composite_rays_train_backward(grad_ws: Tensor, grad_image: Tensor, sigmas: Tensor, rgbs: Tensor, deltas: Tensor,
                                  rays: Tensor, weights_sum: Tensor, image: Tensor, T_thresh: float) -> Tuple[Tensor, Tensor]:
    lib = _lib.load()
    N, M = rays.shape[0], sigmas.shape[0]
    gs, gc = torch.zeros_like(sigmas), torch.zeros_like(rgbs)
    if M:
        check(lib.inr_composite_rays_train_backward(ptr(grad_ws.contiguous().float()), ptr(grad_image.contiguous().float()),
                                                    None, ptr(sigmas), ptr(rgbs), None, ptr(deltas), ptr(rays, I32, "rays"),
                                                    ptr(weights_sum), ptr(image), None, N, M, float(T_thresh), 0, ptr(gs),
                                                    ptr(gc), None, None, stream_ptr()), "composite_rays_train_backward")
    return gs, gc


@composite_rays_train_backward.register_fake
def _(grad_ws, grad_image, sigmas, rgbs, deltas, rays, weights_sum, image, T_thresh):
    return torch.empty_like(sigmas), torch.empty_like(rgbs)


def _composite_setup(ctx, inputs, output):
    sigmas, rgbs, deltas, rays, T_thresh = inputs
    ws, _, image = output
    ctx.save_for_backward(sigmas.contiguous().float(), rgbs.contiguous().float(), deltas.contiguous().float(), rays, ws, image)
    ctx.T_thresh = T_thresh


def _composite_backward(ctx, g_ws, g_depth, g_image):
    sigmas, rgbs, deltas, rays, ws, image = ctx.saved_tensors
    g_ws = torch.zeros_like(ws) if g_ws is None else g_ws
    g_image = torch.zeros_like(image) if g_image is None else g_image
    gs, gc = torch.ops.inr.composite_rays_train_backward(g_ws, g_image, sigmas, rgbs, deltas, rays, ws, image, ctx.T_thresh)
    return gs, gc, None, None, None


composite_rays_train.register_autograd(_composite_backward, setup_context=_composite_setup)


# ------------------------------------------------------------------------------------------------ hash grid
@torch.library.custom_op("inr::grid_encode", mutates_args=())
def grid_encode(x: Tensor, embeddings: Tensor, bound: float, offsets: List[int], scales: List[float],
                resolutions: List[int], hashed: List[int]) -> Tensor:
    """x [M,3] in [-bound, bound], embeddings [T,2] -> [M, 2 L] (level-major)."""
    lib = _lib.load()
    x = x.contiguous().float()
    M, L = x.shape[0], len(scales)
    out = torch.empty(M, 2 * L, dtype=F32, device=x.device)
    check(lib.inr_grid_encode_forward(ptr(x, F32, "x", allow_none=M == 0), ptr(embeddings, F32, "embeddings"),
                                      _desc(offsets, scales, resolutions, hashed), M, float(bound),
                                      ptr(out, allow_none=M == 0), stream_ptr()), "grid_encode_forward")
    return out


@grid_encode.register_fake
def _(x, embeddings, bound, offsets, scales, resolutions, hashed):
    return x.new_empty(x.shape[0], 2 * len(scales), dtype=F32)


@torch.library.custom_op("inr::grid_encode_backward", mutates_args=())
def grid_encode_backward(x: Tensor, grad_out: Tensor, rows: int, bound: float, offsets: List[int], scales: List[float],
                         resolutions: List[int], hashed: List[int]) -> Tensor:
    """dL/d(embeddings) [rows,2] of grid_encode: the atomic table scatter."""
    lib = _lib.load()
    x, grad_out = x.contiguous().float(), grad_out.contiguous().float()
    g = torch.zeros(rows, 2, dtype=F32, device=x.device)
    if x.shape[0]:
        check(lib.inr_grid_encode_backward(ptr(x), ptr(grad_out), _desc(offsets, scales, resolutions, hashed), x.shape[0],
                                           float(bound), ptr(g), stream_ptr()), "grid_encode_backward")
    return g


@grid_encode_backward.register_fake
def _(x, grad_out, rows, bound, offsets, scales, resolutions, hashed):
    return x.new_empty(rows, 2, dtype=F32)


def _grid_setup(ctx, inputs, output):
    x, embeddings, bound, offsets, scales, resolutions, hashed = inputs
    ctx.save_for_backward(x)
    ctx.args = (embeddings.shape[0], bound, offsets, scales, resolutions, hashed)


def _grid_backward(ctx, grad):
    (x,) = ctx.saved_tensors
    rows, bound, offsets, scales, resolutions, hashed = ctx.args
    g = torch.ops.inr.grid_encode_backward(x, grad, rows, bound, offsets, scales, resolutions, hashed)
    return None, g, None, None, None, None, None


grid_encode.register_autograd(_grid_backward, setup_context=_grid_setup)


# ------------------------------------------------------------------------------------------------ fused field
@torch.library.custom_op("inr::nerf_forward", mutates_args=())
def nerf_forward(x: Tensor, d: Tensor, embeddings: Tensor, sigma_w0: Tensor, sigma_w1: Tensor, color_w0: Tensor,
                 color_w1: Tensor, color_w2: Tensor, bound: float, offsets: List[int], scales: List[float],
                 resolutions: List[int], hashed: List[int]) -> Tuple[Tensor, Tensor]:
    """(x [M,3], d [M,3] unit) -> (sigma [M], rgb [M,3]) of the standard architecture (hash grid L x 2 -> 64 -> 16,
    SH-4 + 15 geo features -> 64 -> 64 -> 3) in ONE launch, no autograd (weights are packed on the host per call: use
    NeRFNetwork for a render loop, which caches the packed image)."""
    lib = _lib.load()
    x, d = x.contiguous().float(), d.contiguous().float()
    M = x.shape[0]
    host = [w.detach().float().cpu().contiguous() for w in (sigma_w0, sigma_w1, color_w0, color_w1, color_w2)]
    buf = torch.empty(lib.inr_nerf_packed_floats(), dtype=F32)
    check(lib.inr_nerf_pack_weights(*[_lib.host_ptr(h, F32) for h in host], _lib.host_ptr(buf, F32)), "nerf_pack_weights")
    packed = buf.to(x.device)
    sigma, rgb = torch.empty(M, dtype=F32, device=x.device), torch.empty(M, 3, dtype=F32, device=x.device)
    if M:
        check(lib.inr_nerf_forward(ptr(x), ptr(d), M, None, float(bound), ptr(embeddings, F32, "embeddings"),
                                   _desc(offsets, scales, resolutions, hashed), ptr(packed), 1.0, ptr(sigma), ptr(rgb), None,
                                   stream_ptr()), "nerf_forward")
    return sigma, rgb


@nerf_forward.register_fake
def _(x, d, embeddings, sigma_w0, sigma_w1, color_w0, color_w1, color_w2, bound, offsets, scales, resolutions, hashed):
    return x.new_empty(x.shape[0], dtype=F32), x.new_empty(x.shape[0], 3, dtype=F32)


OPS = ("near_far_from_aabb", "march_rays_train", "composite_rays_train", "composite_rays_train_backward", "grid_encode",
       "grid_encode_backward", "nerf_forward")


# ------------------------------------------------------------------------------------------------ 3-D RoIAlign
@torch.library.custom_op("inr::roi_align_3d", mutates_args=())
def roi_align_3d(input: Tensor, rois: Tensor, roi_inds: Tensor, out_w: int, out_l: int, out_h: int,
                 spatial_scale: float) -> Tensor:
    """input [N,C,W,L,H], rois [K,6], roi_inds int [K] -> [K,C,out_w,out_l,out_h] (torchvision roi_align semantics on
    three axes; the separable HIP kernels of csrc/roialign.hip)."""
    lib = _lib.load()
    input, rois = input.contiguous().float(), rois.contiguous().float()
    roi_inds = roi_inds.contiguous().to(I32)
    N, C, W, L, H = input.shape
    K = rois.shape[0]
    out = torch.empty(K, C, out_w, out_l, out_h, dtype=F32, device=input.device)
    check(lib.inr_roi_align_3d_forward(ptr(input, F32, "input", allow_none=input.numel() == 0),
                                       ptr(rois, F32, "rois", allow_none=K == 0), ptr(roi_inds, I32, "roi_inds", allow_none=K == 0),
                                       N, C, W, L, H, K, out_w, out_l, out_h, float(spatial_scale),
                                       ptr(out, allow_none=K == 0), stream_ptr()), "roi_align_3d_forward")
    return out


@roi_align_3d.register_fake
def _(input, rois, roi_inds, out_w, out_l, out_h, spatial_scale):
    return input.new_empty(rois.shape[0], input.shape[1], out_w, out_l, out_h, dtype=F32)


@torch.library.custom_op("inr::roi_align_3d_backward", mutates_args=())
def roi_align_3d_backward(grad_out: Tensor, rois: Tensor, roi_inds: Tensor, N: int, W: int, L: int, H: int,
                          spatial_scale: float) -> Tensor:
    """dL/d(input) [N,C,W,L,H] of roi_align_3d: the transposed separable passes + atomics (through the channels-fastest
    workspace form where it applies)."""
    from .roi_align.roi_align import roi_align_3d_grad_input
    grad_out, rois = grad_out.contiguous().float(), rois.contiguous().float()
    roi_inds = roi_inds.contiguous().to(I32)
    K, C, ow, ol, oh = grad_out.shape
    return roi_align_3d_grad_input(grad_out, rois, roi_inds, (N, C, W, L, H), (ow, ol, oh, float(spatial_scale)))


@roi_align_3d_backward.register_fake
def _(grad_out, rois, roi_inds, N, W, L, H, spatial_scale):
    return grad_out.new_empty(N, grad_out.shape[1], W, L, H, dtype=F32)


def _roi_setup(ctx, inputs, output):
    input, rois, roi_inds, out_w, out_l, out_h, spatial_scale = inputs
    ctx.save_for_backward(rois, roi_inds)
    ctx.args = (input.shape[0], input.shape[2], input.shape[3], input.shape[4], spatial_scale)


def _roi_backward(ctx, grad):
    rois, roi_inds = ctx.saved_tensors
    N, W, L, H, scale = ctx.args
    return torch.ops.inr.roi_align_3d_backward(grad, rois, roi_inds, N, W, L, H, scale), None, None, None, None, None, None


roi_align_3d.register_autograd(_roi_backward, setup_context=_roi_setup)
