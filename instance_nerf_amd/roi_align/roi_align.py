"""``roi_align_3d`` with the signature the reference calls
(/root/reference/nerf_rcnn/model/utils.py:608): differentiable w.r.t. ``input``."""
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


_WS = {}
_COVER = {}


def _workspace(device, nbytes):
    """Scratch of the backward's workspace form, one per device and stream (the call zeroes what it uses; a buffer shared
    between streams would race).  Grows on demand and is given back when a call needs less than a quarter of it (round-5
    advisor: a grow-only cache kept N*C*V*4 bytes alive per stream for the life of the process); ``release_workspace()``
    drops everything."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes or buf.numel() > 4 * max(nbytes, 1 << 20):
        buf = _WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return buf


def release_workspace():
    """Frees the cached scratch volumes of the workspace backward (and the cached RoI statistics)."""
    _WS.clear()
    _COVER.clear()


def _covered_voxels(rois, shape, cfg):
    """Voxels inside the RoIs' regions, summed over the RoIs - what decides between the two backward forms
    (``inr_roi_align_3d_backward_prefers_workspace``).  The RoIs live on the device and a read-back per call would
    synchronise, so the figure is measured ONCE per call shape: the first call queues the reduction and an asynchronous
    copy to pinned host memory and answers -1 (unknown: the library then assumes the lower bound and keeps the in-place
    form); later calls of the same shape use the value once the copy has landed.  RoI statistics of a training run are
    stable from step to step; a shape whose RoIs change character can be re-measured with ``release_workspace()``."""
    N, C, W, L, H = shape
    ow, ol, oh, scale = cfg
    key = (rois.device, N, C, W, L, H, rois.shape[0], ow, ol, oh, float(scale))
    st = _COVER.get(key)
    if st is None:
        ext = ((rois[:, 3:] - rois[:, :3]).float() * float(scale) + 1.0)
        lim = torch.tensor([W, L, H], dtype=torch.float32, device=rois.device)
        total = torch.minimum(ext.clamp(min=1.0), lim).prod(1).sum(dtype=torch.float64).reshape(1)
        host = torch.empty(1, dtype=torch.float64).pin_memory()
        host.copy_(total, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st = _COVER[key] = {"host": host, "event": ev, "value": None}
    if st["value"] is None and st["event"].query():
        st["value"] = int(st["host"][0])
    return -1 if st["value"] is None else st["value"]


class _RoIAlign3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, rois, roi_inds, out_w, out_l, out_h, spatial_scale):
        lib = _lib.load()
        input = input.contiguous().float()
        rois = rois.contiguous().float()
        roi_inds = roi_inds.contiguous().to(torch.int32)
        if input.dim() != 5 or rois.dim() != 2 or rois.shape[1] != 6 or roi_inds.shape[0] != rois.shape[0]:
            raise RuntimeError("roi_align_3d: input must be [N,C,W,L,H], rois [K,6], roi_inds [K]")
        N, C, W, L, H = input.shape
        K = rois.shape[0]
        out = torch.empty(K, C, out_w, out_l, out_h, dtype=torch.float32, device=input.device)
        check(lib.inr_roi_align_3d_forward(ptr(input, torch.float32, "input", allow_none=input.numel() == 0),
                                           ptr(rois, torch.float32, "rois", allow_none=K == 0),
                                           ptr(roi_inds, torch.int32, "roi_inds", allow_none=K == 0), N, C, W, L, H, K,
                                           out_w, out_l, out_h, float(spatial_scale), ptr(out, allow_none=K == 0),
                                           stream_ptr()), "roi_align_3d_forward")
        ctx.save_for_backward(rois, roi_inds)
        ctx.shape = (N, C, W, L, H)
        ctx.cfg = (out_w, out_l, out_h, float(spatial_scale))
        return out

    @staticmethod
    def backward(ctx, grad):
        rois, roi_inds = ctx.saved_tensors
        return (roi_align_3d_grad_input(grad, rois, roi_inds, ctx.shape, ctx.cfg), None, None, None, None, None, None)


def roi_align_3d_grad_input(grad, rois, roi_inds, shape, cfg):
    """dL/d(input) f32[N,C,W,L,H] for grad f32[K,C,ow,ol,oh]; shape = (N,C,W,L,H), cfg = (ow,ol,oh,spatial_scale).  Shared
    by the autograd function above and the registered custom op (ops.py)."""
    lib = _lib.load()
    N, C, W, L, H = shape
    ow, ol, oh, scale = cfg
    K = rois.shape[0]
    grad = grad.contiguous().float()
    # the workspace form (channels-fastest accumulation + transposing copy: fewer and fuller atomic requests) where the
    # library offers it for these extents; it OVERWRITES grad_input, so no zero fill here
    # ... AND expects it to be faster (cost model in the library, fed with the measured RoI coverage of this call shape:
    # the form adds three passes over the N*C*V volume, which only a large covered fraction pays for)
    need = int(lib.inr_roi_align_3d_backward_workspace_bytes(N, C, W, L, H, K, ow, ol, oh)) if K > 0 and N * C * W * L * H else 0
    if need > 0 and not lib.inr_roi_align_3d_backward_prefers_workspace(N, C, W, L, H, K, ow, ol, oh,
                                                                         _covered_voxels(rois, shape, cfg)):
        need = 0
    if need > 0:
        gin = torch.empty(N, C, W, L, H, dtype=torch.float32, device=grad.device)
        ws = _workspace(grad.device, need)
        check(lib.inr_roi_align_3d_backward_ws(ptr(grad), ptr(rois), ptr(roi_inds), N, C, W, L, H, K, ow, ol, oh, float(scale),
                                               ptr(gin), ws.data_ptr(), need, stream_ptr()), "roi_align_3d_backward_ws")
        return gin
    gin = torch.zeros(N, C, W, L, H, dtype=torch.float32, device=grad.device)
    check(lib.inr_roi_align_3d_backward(ptr(grad, allow_none=K == 0), ptr(rois, allow_none=K == 0),
                                        ptr(roi_inds, allow_none=K == 0), N, C, W, L, H, K, ow, ol, oh, float(scale),
                                        ptr(gin, allow_none=gin.numel() == 0), stream_ptr()), "roi_align_3d_backward")
    return gin


def roi_align_3d(input, rois, roi_inds, out_w, out_l, out_h, spatial_scale):
    """input f32[N,C,W,L,H], rois f32[K,6] (x1,y1,z1,x2,y2,z2), roi_inds int[K] -> f32[K,C,out_w,out_l,out_h]."""
    return _RoIAlign3D.apply(input, rois, roi_inds, int(out_w), int(out_l), int(out_h), spatial_scale)
