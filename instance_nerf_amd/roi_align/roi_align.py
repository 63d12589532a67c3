"""``roi_align_3d`` with the signature the reference calls
(/root/reference/nerf_rcnn/model/utils.py:608): differentiable w.r.t. ``input``."""
import torch

from .. import _lib
from .._lib import check, ptr, stream_ptr


_WS = {}


def _workspace(device, nbytes):
    """Grow-only scratch of the backward's workspace form, one per device and stream (the call zeroes what it uses; a
    buffer shared between streams would race)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return buf


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
    need = int(lib.inr_roi_align_3d_backward_workspace_bytes(N, C, W, L, H, K, ow, ol, oh)) if K > 0 and N * C * W * L * H else 0
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
