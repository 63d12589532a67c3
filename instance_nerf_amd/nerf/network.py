"""``NeRFNetwork`` - hash-grid NeRF + instance field (8b).

Constructor, parameter names (``encoder.embeddings``, ``sigma_net.N.weight``,
``color_net.N.weight``) and the ``forward / density / color / get_params`` methods
follow upstream ``nerf/network.py`` of the reference's un-vendored submodule
(/root/reference/.gitmodules:4-6, README.md:27,59) so its checkpoints load.
The instance head (``instance_encoder``, ``instance_net``, ``num_instances``) is
this repository's reading of the fork's addition [U-fork]: a position-only
hash-grid + MLP producing K raw logits.

Execution paths, all HIP:
* no-grad (render / occupancy update / frozen NeRF under instance training):
  ONE fused kernel per call - gather + SH + MLPs on the matrix cores
  (csrc/field_fused.hip);
* grad, standard architecture (``_NerfFieldFn`` / ``_InstanceFieldFn``): a fused
  forward that keeps the activations, ONE fused backward for the whole
  input-gradient chain, the two-pass MFMA weight-gradient kernel and the atomic
  table scatter;
* grad, any other layer shape (or ``fused_*_train = False``): hand-written HIP
  encoders with the tiny MLP GEMMs left to rocBLAS through ``nn.Linear`` -
  upstream's default configuration (its ``--ff`` fused MLP is optional there too).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from .._lib import check, host_ptr, ptr, stream_ptr
from ..activation import trunc_exp
from ..encoding import get_encoder
from .renderer import NeRFRenderer


class _LinearFn(torch.autograd.Function):
    """y = x W^T with the weight gradient on the split-K MFMA kernel (inr_linear_wgrad): the reduction runs
    over ~2e5 samples with a <= 64x64 result, a shape the BLAS library serialises on a few workgroups."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return F.linear(x, w)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = gy @ w
        if ctx.needs_input_grad[1]:
            lib = _lib.load()
            gyc, xc = gy.contiguous(), x.contiguous()
            gw = torch.zeros_like(w)
            ws = torch.empty(lib.inr_linear_wgrad_workspace_bytes() // 4, dtype=torch.float32, device=w.device)
            check(lib.inr_linear_wgrad(ptr(xc, torch.float32, "x"), ptr(gyc, torch.float32, "grad_y"), xc.shape[0],
                                       w.shape[1], w.shape[0], ptr(gw), ptr(ws), stream_ptr()), "linear_wgrad")
        return gx, gw


_before_scatter = {}      # {"hook": callable}: called once, right before the next table-gradient scatter is queued


# Fixed-point table-gradient scatter (round 6; include/inr.h "Fixed-point form of the table-gradient scatter"): OPT-IN,
# INR_FX_GRAD=32 | 64 (1 = 32) / Trainer(fixed_point_grad=...) / network.FX_GRAD = 32 | 64.  The gradient becomes independent
# of the order in which the waves' requests arrive (training steps are bit-reproducible).
#   32: int32 sums - the memory-side atomic unit takes them 28 % faster than fp32 adds - but rows whose gradient is below the
#       level's quantum (1.2e-7 of its recent maximum) get none, where Adam (eps 1e-15) moves them by a full lr step under
#       fp32 atomics: same converged quality, slower early convergence of weakly supervised rows (profiles/r06_NOTES.txt 5);
#   64: int64 sums in a separate accumulator - quantum 2e-16 of the level's maximum, invisible to Adam: the faithful
#       reproducible form, at about the fp32 scatter's speed.
# The default is upstream's arithmetic: fp32 atomics.
def _fx_env():
    v = os.environ.get("INR_FX_GRAD", "0")
    return {"0": 0, "": 0, "1": 32, "32": 32, "64": 64}.get(v, 0)


FX_GRAD = _fx_env()                    # 0 | 32 | 64 (True counts as 32)
FX_HEADROOM = float(os.environ.get("INR_FX_HEADROOM", "128"))
FX_HEADROOM64 = float(os.environ.get("INR_FX_HEADROOM64", "1024"))


def fx_bits():
    return 64 if FX_GRAD == 64 else (32 if FX_GRAD else 0)


def fx_acc64(emb, create=True):
    """The table's int64 accumulator of the 64-bit form ([T,2], zero between steps), kept on the Parameter object."""
    acc = getattr(emb, "_fx_acc64", None)
    if (acc is None or acc.device != emb.device or acc.shape != emb.shape) and create and emb.is_cuda:
        acc = torch.zeros(emb.shape, dtype=torch.int64, device=emb.device)
        emb._fx_acc64 = acc
    return acc


def fx_state(emb, create=True):
    """The table's fixed-point state (device floats, zero = "no scale yet: fp32 atomics"), kept on the Parameter object;
    created on first use - ``Trainer`` creates it before it captures a step (nothing may be allocated inside a capture)."""
    st = getattr(emb, "_fx_state", None)
    if (st is None or st.device != emb.device) and create and emb.is_cuda:
        st = torch.zeros(_lib.GRID_FX_STATE_FLOATS, dtype=torch.float32, device=emb.device)
        emb._fx_state = st
        emb._fx_primed = 0               # 0 | 32 | 64: the form the scales were primed for
    return st


def _table_backward(lib, x, denc, desc, M, bound, g_emb, emb):
    """Table-gradient scatter; returns the gradient to hand back to autograd (None when it was installed directly).

    One process: one launch over all levels.  Several ranks: two level ranges, fine levels first, each handed to the
    gradient all-reduce as soon as its launch is queued (nerf/utils.py::grad_sync), so the collective of the first
    range runs under the scatter of the second and under the weight-gradient kernels.  In that case the buffer
    becomes ``emb.grad`` right here and autograd gets None: autograd would otherwise COPY the returned tensor into
    ``.grad`` (it adopts it only while no one else references it - the in-flight collective does) at a moment when
    the buffer is half reduced (tests/test_gpu_ddp.py caught exactly that: one run in three diverged)."""
    from .utils import grad_sync
    L = int(desc.num_levels)
    hook = _before_scatter.pop("hook", None)
    if hook is not None:
        hook()      # Trainer's look-ahead: the next batch's march goes on a side stream right beside THIS launch (the
        #             scatter is bound by the memory-side atomic unit and leaves the CUs idle; the MFMA-bound kernels
        #             before it do not)
    if grad_sync.active() and emb.data_ptr() in grad_sync.early:
        # a second backward before allreduce_gradients(): the first gradient is already being summed over the ranks,
        # adding an unreduced one to it cannot be reduced correctly afterwards
        raise RuntimeError("gradient accumulation over several backward passes is not supported together with the "
                           "overlapped table-gradient all-reduce; set INR_GRAD_OVERLAP=0")
    overlap = grad_sync.active() and L > 8 and emb.grad is None
    bits = fx_bits() if (g_emb.is_cuda and g_emb.data_ptr() % 16 == 0) else 0
    fx = fx_state(emb) if bits else None
    acc = fx_acc64(emb) if bits == 64 else None
    headroom = FX_HEADROOM64 if bits == 64 else FX_HEADROOM

    def scatter(dst, lo, hi):
        if bits == 64:
            check(lib.inr_grid_encode_backward_levels_fx64(ptr(x), ptr(denc), None, desc, M, float(bound), ptr(dst), ptr(acc), lo, hi,
                                                           ptr(fx), stream_ptr()), "grid_encode_backward (int64 sums)")
        else:
            check(lib.inr_grid_encode_backward_levels_fx(ptr(x), ptr(denc), None, desc, M, float(bound), ptr(dst), lo, hi,
                                                         ptr(fx, allow_none=True), stream_ptr()), "grid_encode_backward")

    def finish(dst, lo, hi):
        if bits == 64:
            check(lib.inr_grid_grad_finish_fx64(ptr(acc), ptr(dst), desc, lo, hi, ptr(fx), stream_ptr()), "grid_grad_finish_fx64")
        else:
            check(lib.inr_grid_grad_finish_fx(ptr(dst), desc, lo, hi, ptr(fx), stream_ptr()), "grid_grad_finish_fx")

    if fx is not None and M and getattr(emb, "_fx_primed", 0) != bits and not torch.cuda.is_current_stream_capturing():
        # The table's very first backward (in this form) has no scales yet and would run on fp32 atomics - the one step
        # whose result depends on the order of arrival.  Prime instead: scatter once into a scratch buffer only to learn the
        # levels' magnitudes (finish + update set the scales), then take the step itself on integer sums like every later one.
        fx.zero_()
        scratch = torch.zeros_like(g_emb)
        scatter(scratch, 0, L)
        finish(scratch, 0, L)
        check(lib.inr_grid_fx_update(ptr(fx), L, headroom, bits, stream_ptr()), "grid_fx_update (priming)")
        emb._fx_primed = bits
        del scratch
    for lo, hi in (((8, L), (0, 8)) if overlap else ((0, L),)):
        if M:          # a batch without a single sample still takes part in the collectives below (zeros): every rank
            #            must issue the same sequence of all-reduces or the job hangs
            scatter(g_emb, lo, hi)
        if fx is not None:
            # integer sums -> fp32 gradients (+ the levels' maxima): from here on g_emb is an ordinary gradient
            finish(g_emb, lo, hi)
        if overlap:
            a, b = int(desc.offsets[lo]), int(desc.offsets[hi])
            grad_sync.reduce_async(g_emb[a:b], emb, a * g_emb.shape[1])
    if fx is not None:
        check(lib.inr_grid_fx_update(ptr(fx), L, headroom, bits, stream_ptr()), "grid_fx_update")      # next step's scales
    if overlap:
        emb.grad = g_emb
        grad_sync.mark(emb, g_emb)
        return None
    return g_emb


class _InstanceFieldFn(torch.autograd.Function):
    """Instance field x -> logits for TRAINING in two fused launches (csrc/field_fused.hip): the forward gathers,
    runs the three layers on the matrix cores and keeps enc / h1 / h2 for the backward; the backward runs the
    whole input-gradient chain (W^T as the MFMA A operand, ReLU masks from the saved activations) in one kernel.
    Weight gradients stay on the split-K kernel, the table gradient on the atomic scatter."""

    @staticmethod
    def forward(ctx, x, emb, w0, w1, w2, desc, bound):
        lib = _lib.load()
        f32 = torch.float32
        K, M, dev = w2.shape[0], x.shape[0], x.device
        pf = torch.empty(lib.inr_instance_packed_floats(K), dtype=f32, device=dev)
        pb = torch.empty(lib.inr_instance_bwd_packed_floats(), dtype=f32, device=dev)
        check(lib.inr_instance_pack_weights_device(ptr(w0.detach().contiguous(), f32, "w0"),
                                                   ptr(w1.detach().contiguous(), f32, "w1"),
                                                   ptr(w2.detach().contiguous(), f32, "w2"), K, ptr(pf), ptr(pb),
                                                   stream_ptr()), "instance_pack_weights_device")
        logits = torch.empty(M, K, dtype=f32, device=dev)
        act = torch.empty(M, 32 + 64 + 64, dtype=f32, device=dev) if M else torch.empty(0, 160, dtype=f32, device=dev)
        enc, h1, h2 = act.view(-1)[:M * 32].view(M, 32), act.view(-1)[M * 32:M * 96].view(M, 64), \
            act.view(-1)[M * 96:].view(M, 64)
        check(lib.inr_instance_forward_train(ptr(x, f32, "x", allow_none=M == 0), M, float(bound),
                                             ptr(emb.detach(), f32, "embeddings"), desc, ptr(pf), K,
                                             ptr(logits, allow_none=M == 0), ptr(enc, allow_none=M == 0),
                                             ptr(h1, allow_none=M == 0), ptr(h2, allow_none=M == 0), stream_ptr()),
              "instance_forward_train")
        ctx.save_for_backward(x, enc, h1, h2, pb, emb)
        ctx.desc, ctx.bound, ctx.K = desc, bound, K
        return logits

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        f32 = torch.float32
        x, enc, h1, h2, pb, emb = ctx.saved_tensors
        K, M, dev = ctx.K, x.shape[0], x.device
        g = g.contiguous().float()
        grads = torch.empty(M, 64 + 64 + 32, dtype=f32, device=dev)
        dz2, dz1, denc = grads.view(-1)[:M * 64].view(M, 64), grads.view(-1)[M * 64:M * 128].view(M, 64), \
            grads.view(-1)[M * 128:].view(M, 32)
        gw = torch.zeros(64 * 32 + 64 * 64 + K * 64, dtype=f32, device=dev)      # one fill for the three weights
        gw0, gw1, gw2 = gw[:2048].view(64, 32), gw[2048:6144].view(64, 64), gw[6144:].view(K, 64)
        g_emb = torch.zeros_like(emb)
        if M:
            check(lib.inr_instance_backward(ptr(g, f32, "grad_logits"), K, ptr(h1), ptr(h2), M, ptr(pb), ptr(dz2),
                                            ptr(dz1), ptr(denc), stream_ptr()), "instance_backward")
        g_emb = _table_backward(lib, x, denc, ctx.desc, M, ctx.bound, g_emb, emb)          # first: its all-reduce can start
        if M:
            ws = torch.empty(lib.inr_linear_wgrad_workspace_bytes() // 4, dtype=f32, device=dev)   # reused in stream order
            for xin, gy, n_in, n_out, out in ((h2, g, 64, K, gw2), (h1, dz2, 64, 64, gw1), (enc, dz1, 32, 64, gw0)):
                check(lib.inr_linear_wgrad(ptr(xin), ptr(gy), M, n_in, n_out, ptr(out), ptr(ws), stream_ptr()),
                      "linear_wgrad")
        return None, g_emb, gw0, gw1, gw2, None, None


class _InstanceHeadFn(torch.autograd.Function):
    """The instance head of a TRAINING render as one autograd node: samples x [M,3] + their (detached) compositing
    weights and owning rays -> rendered logits [N,K] (+ the mask loss, see below).  Forward: fused instance field (only
    the encoder output is kept) + K-channel compositing.  Backward: ONE launch (csrc/field_fused.hip::
    k_instance_head_bwd) from dL/d(rendered logits) to dL/denc and the three weight gradients - the [M,K] logit
    gradient, the saved hidden activations, their gradients and the three split-K weight-gradient launches of
    ``_InstanceFieldFn`` never exist - then the table scatter.  Same arithmetic as the composable chain
    ``_InstanceFieldFn`` -> ``composite_rays_train(extra=...)`` except that the weight gradients are summed in a
    different order (fp32 MFMA over 16-sample tiles).

    ``labels`` (int64 [N], optional): the cross entropy of the rendered logits against them (``ignore_index`` rows
    skipped, classes 0 .. n_classes-1) is formed by the compositing kernel itself while a ray's logits are in
    registers, and the node returns it as a second output; its backward hands the unnormalised softmax - onehot rows to
    the backward launch together with two device scalars (1 / kept rows, dL/dloss) - no [N,K] pass for the loss at all.
    """

    @staticmethod
    def forward(ctx, x, weights, sample_ray, rays, n_dev, labels, n_classes, ignore_index, emb, w0, w1, w2, desc, bound):
        lib = _lib.load()
        f32 = torch.float32
        K, M, N, dev = w2.shape[0], x.shape[0], rays.shape[0], x.device
        pf = torch.empty(lib.inr_instance_packed_floats(K), dtype=f32, device=dev)
        pb = torch.empty(lib.inr_instance_bwd_packed_floats(), dtype=f32, device=dev)
        check(lib.inr_instance_pack_weights_device(ptr(w0.detach().contiguous(), f32, "w0"),
                                                   ptr(w1.detach().contiguous(), f32, "w1"),
                                                   ptr(w2.detach().contiguous(), f32, "w2"), K, ptr(pf), ptr(pb),
                                                   stream_ptr()), "instance_pack_weights_device")
        logits = torch.empty(M, K, dtype=f32, device=dev)
        enc = torch.empty(M, 32, dtype=f32, device=dev)
        none_ok = M == 0
        check(lib.inr_instance_forward_enc(ptr(x, f32, "x", allow_none=none_ok), M, ptr(n_dev, torch.int32, "n_dev", allow_none=True),
                                           float(bound), ptr(emb.detach(), f32, "embeddings"), desc, ptr(pf), K,
                                           ptr(logits, allow_none=none_ok), ptr(enc, allow_none=none_ok), stream_ptr()),
              "instance_forward_enc")
        out = torch.empty(N, K, dtype=f32, device=dev)
        dpix = ce_ws = ce_out = None
        if labels is not None:
            labels = labels.contiguous().long().reshape(-1)
            if labels.shape[0] != N:
                raise RuntimeError(f"ce_labels: {labels.shape[0]} labels for {N} rays")
            dpix = torch.empty(N, K, dtype=f32, device=dev)
            ce_ws = torch.empty(max(N, 1) * 4, dtype=f32, device=dev)
            ce_out = torch.empty(4, dtype=f32, device=dev)
        check(lib.inr_composite_rays_extra_forward(ptr(weights, f32, "weights", allow_none=none_ok),
                                                   ptr(logits, allow_none=none_ok), ptr(rays, torch.int32, "rays"), N, M, K,
                                                   ptr(out, allow_none=N == 0), ptr(labels, torch.int64, "labels", allow_none=True),
                                                   int(n_classes), int(ignore_index), ptr(dpix, allow_none=True),
                                                   ptr(ce_ws, allow_none=True), ptr(ce_out, allow_none=True), stream_ptr()),
              "composite_rays_extra_forward")
        ctx.save_for_backward(x, enc, weights, sample_ray, pf, pb, emb, dpix, ce_out)
        ctx.n_dev, ctx.desc, ctx.bound, ctx.K, ctx.N = n_dev, desc, bound, K, N
        ctx.set_materialize_grads(False)
        loss = ce_out[0] if labels is not None else out.new_zeros(())
        return out, loss

    @staticmethod
    def backward(ctx, g, g_loss):
        lib = _lib.load()
        f32 = torch.float32
        x, enc, weights, sample_ray, pf, pb, emb, dpix, ce_out = ctx.saved_tensors
        K, M, N, dev = ctx.K, x.shape[0], ctx.N, x.device
        scale_a = scale_b = None
        if dpix is not None and g_loss is not None and g is None:
            g_pix = dpix                                  # unnormalised: the launch scales by 1 / kept and dL/dloss
            scale_a, scale_b = ce_out[1:2], g_loss.detach().float().reshape(1).contiguous()
        elif dpix is not None and g_loss is not None:     # the caller also differentiates through the rendered logits
            g_pix = (g.float() + dpix * (ce_out[1] * g_loss.float())).contiguous()
        elif g is not None:
            g_pix = g.contiguous().float()
        else:
            g_pix = torch.zeros(N, K, dtype=f32, device=dev)
        denc = torch.empty(M, 32, dtype=f32, device=dev)
        gw = torch.empty(64 * 32 + 64 * 64 + K * 64, dtype=f32, device=dev)          # written by the reduce kernel
        gw0, gw1, gw2 = gw[:2048].view(64, 32), gw[2048:6144].view(64, 64), gw[6144:].view(K, 64)
        ws = torch.empty(lib.inr_instance_head_workspace_bytes() // 4, dtype=f32, device=dev)
        g_emb = torch.empty_like(emb)              # zero-filled by the backward launch, on the side
        none_ok = M == 0
        check(lib.inr_instance_head_backward(ptr(enc, allow_none=none_ok), ptr(weights, allow_none=none_ok),
                                             ptr(sample_ray, torch.int32, "sample_ray", allow_none=none_ok),
                                             ptr(g_pix, f32, "grad_pix", allow_none=N == 0), K, N, M,
                                             ptr(ctx.n_dev, torch.int32, "n_dev", allow_none=True),
                                             ptr(scale_a, f32, "scale_a", allow_none=True),
                                             ptr(scale_b, f32, "scale_b", allow_none=True), ptr(pf), ptr(pb),
                                             ptr(denc, allow_none=none_ok), ptr(ws), ptr(gw0), ptr(gw1), ptr(gw2),
                                             ptr(g_emb), g_emb.numel(), stream_ptr()), "instance_head_backward")
        g_emb = _table_backward(lib, x, denc, ctx.desc, M, ctx.bound, g_emb, emb)
        return None, None, None, None, None, None, None, None, g_emb, gw0, gw1, gw2, None, None


class _NerfFieldFn(torch.autograd.Function):
    """(x, d) -> (sigma, rgb) of the NeRF field for TRAINING: a fused forward that keeps only the encoder output and
    ONE backward launch (csrc/field_fused.hip::k_nerf_head_bwd) that recomputes the forward from it, runs the whole
    input-gradient chain (colour net -> geo features / density logit -> sigma net -> encoder) and accumulates the five
    weight gradients on the fp32 matrix cores; then the atomic table scatter.  (Rounds 1-2: six saved activation
    arrays - 1088 B per sample -, k_nerf_bwd and five split-K weight-gradient launches; still available as
    ``_NerfFieldFnUnfused`` / ``NeRFNetwork.fused_nerf_head = False``.)  sigma is returned WITHOUT density_scale
    (the renderer applies it)."""

    @staticmethod
    def forward(ctx, x, d, emb, ws0, ws1, wc0, wc1, wc2, desc, bound):
        lib = _lib.load()
        f32 = torch.float32
        M, dev = x.shape[0], x.device
        pf = torch.empty(lib.inr_nerf_packed_floats(), dtype=f32, device=dev)
        pb = torch.empty(lib.inr_nerf_bwd_packed_floats(), dtype=f32, device=dev)
        ws = [w.detach().contiguous() for w in (ws0, ws1, wc0, wc1, wc2)]
        check(lib.inr_nerf_pack_weights_device(*[ptr(w, f32, "weight") for w in ws], ptr(pf), ptr(pb), stream_ptr()),
              "nerf_pack_weights_device")
        sigma = torch.empty(M, dtype=f32, device=dev)
        rgb = torch.empty(M, 3, dtype=f32, device=dev)
        enc = torch.empty(M, 32, dtype=f32, device=dev)
        if M:
            check(lib.inr_nerf_forward_enc(ptr(x, f32, "x"), ptr(d, f32, "d"), M, float(bound),
                                           ptr(emb.detach(), f32, "embeddings"), desc, ptr(pf), ptr(sigma), ptr(rgb),
                                           ptr(enc), stream_ptr()), "nerf_forward_enc")
        ctx.save_for_backward(x, d, enc, pf, pb, emb)
        ctx.desc, ctx.bound = desc, bound
        ctx.set_materialize_grads(False)
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        lib = _lib.load()
        f32 = torch.float32
        x, d, enc, pf, pb, emb = ctx.saved_tensors
        M, dev = x.shape[0], x.device
        g_sigma = torch.zeros(M, dtype=f32, device=dev) if g_sigma is None else g_sigma.contiguous().float()
        g_rgb = torch.zeros(M, 3, dtype=f32, device=dev) if g_rgb is None else g_rgb.contiguous().float()
        d_enc = torch.empty(M, 32, dtype=f32, device=dev)
        sizes = (16 * 64, 64 * 64, 64 * 31, 16 * 64, 64 * 32)   # wc2 (16 rows, 3 live), wc1, wc0, ws1, ws0
        gw = torch.empty(sum(sizes), dtype=f32, device=dev)     # written by the reduce kernel
        gwc2, gwc1, gwc0, gws1, gws0 = [g.view(*shape) for g, shape in zip(
            gw.split(sizes), ((16, 64), (64, 64), (64, 31), (16, 64), (64, 32)))]
        wsp = torch.empty(lib.inr_instance_head_workspace_bytes() // 4, dtype=f32, device=dev)
        g_emb = torch.empty_like(emb)              # zero-filled by the backward launch, on the side
        none_ok = M == 0
        check(lib.inr_nerf_head_backward(ptr(enc, allow_none=none_ok), ptr(d, allow_none=none_ok),
                                         ptr(g_sigma, allow_none=none_ok), ptr(g_rgb, allow_none=none_ok), M, 1.0, ptr(pf),
                                         ptr(pb), ptr(d_enc, allow_none=none_ok), ptr(wsp), ptr(gws0), ptr(gws1), ptr(gwc0),
                                         ptr(gwc1), ptr(gwc2), ptr(g_emb), g_emb.numel(), stream_ptr()), "nerf_head_backward")
        g_emb = _table_backward(lib, x, d_enc, ctx.desc, M, ctx.bound, g_emb, emb)
        return None, None, g_emb, gws0, gws1, gwc0, gwc1, gwc2[:3], None, None


class _NerfFieldFnUnfused(torch.autograd.Function):
    """(x, d) -> (sigma, rgb) of the NeRF field for TRAINING: one fused forward that keeps the activations and one
    fused backward for the whole input-gradient chain (colour net -> geo features / density logit -> sigma net ->
    encoder), csrc/field_fused.hip::k_nerf_fwd<.., kSave> / k_nerf_bwd.  Weight gradients: the two-pass MFMA kernel;
    table gradient: the atomic scatter.  sigma is returned WITHOUT density_scale (the renderer applies it)."""

    @staticmethod
    def forward(ctx, x, d, emb, ws0, ws1, wc0, wc1, wc2, desc, bound):
        lib = _lib.load()
        f32 = torch.float32
        M, dev = x.shape[0], x.device
        pf = torch.empty(lib.inr_nerf_packed_floats(), dtype=f32, device=dev)
        pb = torch.empty(lib.inr_nerf_bwd_packed_floats(), dtype=f32, device=dev)
        ws = [w.detach().contiguous() for w in (ws0, ws1, wc0, wc1, wc2)]
        check(lib.inr_nerf_pack_weights_device(*[ptr(w, f32, "weight") for w in ws], ptr(pf), ptr(pb), stream_ptr()),
              "nerf_pack_weights_device")
        sigma = torch.empty(M, dtype=f32, device=dev)
        rgb = torch.empty(M, 3, dtype=f32, device=dev)
        widths = (32, 64, 16, 32, 64, 64)                       # enc, h1, so, cin, c1, c2
        act = torch.empty(M * sum(widths), dtype=f32, device=dev)
        parts, off = [], 0
        for wdt in widths:
            parts.append(act[off:off + M * wdt].view(M, wdt))
            off += M * wdt
        enc, h1, so, cin, c1, c2 = parts
        if M:
            check(lib.inr_nerf_forward_train(ptr(x, f32, "x"), ptr(d, f32, "d"), M, float(bound),
                                             ptr(emb.detach(), f32, "embeddings"), desc, ptr(pf), ptr(sigma), ptr(rgb),
                                             ptr(enc), ptr(h1), ptr(so), ptr(cin), ptr(c1), ptr(c2), stream_ptr()),
                  "nerf_forward_train")
        ctx.save_for_backward(x, rgb, enc, h1, so, cin, c1, c2, pb, emb)
        ctx.desc, ctx.bound = desc, bound
        ctx.set_materialize_grads(False)
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        lib = _lib.load()
        f32 = torch.float32
        x, rgb, enc, h1, so, cin, c1, c2, pb, emb = ctx.saved_tensors
        M, dev = x.shape[0], x.device
        g_sigma = torch.zeros(M, dtype=f32, device=dev) if g_sigma is None else g_sigma.contiguous().float()
        g_rgb = torch.zeros(M, 3, dtype=f32, device=dev) if g_rgb is None else g_rgb.contiguous().float()
        widths = (4, 64, 64, 16, 64, 32)                        # d_o, dz_c2, dz_c1, d_so, dz_h1, d_enc
        buf = torch.empty(M * sum(widths), dtype=f32, device=dev)
        parts, off = [], 0
        for wdt in widths:
            parts.append(buf[off:off + M * wdt].view(M, wdt))
            off += M * wdt
        d_o, dz_c2, dz_c1, d_so, dz_h1, d_enc = parts
        sizes = (4 * 64, 64 * 64, 64 * 32, 16 * 64, 64 * 32)    # wc2 (4 rows, 3 live), wc1, wc0 (32 cols, 31 live), ws1, ws0
        gw = torch.zeros(sum(sizes), dtype=f32, device=dev)
        gwc2, gwc1, gwc0, gws1, gws0 = [g.view(*shape) for g, shape in zip(
            gw.split(sizes), ((4, 64), (64, 64), (64, 32), (16, 64), (64, 32)))]
        g_emb = torch.zeros_like(emb)
        if M:
            check(lib.inr_nerf_backward(ptr(g_sigma), ptr(g_rgb), ptr(rgb), ptr(so), ptr(h1), ptr(c1), ptr(c2), M, 1.0,
                                        ptr(pb), ptr(d_o), ptr(dz_c2), ptr(dz_c1), ptr(d_so), ptr(dz_h1), ptr(d_enc),
                                        stream_ptr()), "nerf_backward")
        g_emb = _table_backward(lib, x, d_enc, ctx.desc, M, ctx.bound, g_emb, emb)
        if M:
            wsp = torch.empty(lib.inr_linear_wgrad_workspace_bytes() // 4, dtype=f32, device=dev)
            for xin, gy, n_in, n_out, out in ((c2, d_o, 64, 4, gwc2), (c1, dz_c2, 64, 64, gwc1), (cin, dz_c1, 32, 64, gwc0),
                                              (h1, d_so, 64, 16, gws1), (enc, dz_h1, 32, 64, gws0)):
                check(lib.inr_linear_wgrad(ptr(xin), ptr(gy), M, n_in, n_out, ptr(out), ptr(wsp), stream_ptr()),
                      "linear_wgrad")
        return None, None, g_emb, gws0, gws1, gwc0[:, :31], gwc1, gwc2[:3], None, None


class HipLinear(nn.Linear):
    """nn.Linear(bias=False) whose backward uses the HIP weight-gradient kernel (same parameters/state dict)."""

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.in_features <= 64
                and self.out_features <= 64 and torch.is_grad_enabled() and self.weight.requires_grad):
            return _LinearFn.apply(x, self.weight)
        return F.linear(x, self.weight)


def _mlp(in_dim, hidden, out_dim, num_layers):
    layers = []
    for l in range(num_layers):
        layers.append(HipLinear(in_dim if l == 0 else hidden, out_dim if l == num_layers - 1 else hidden, bias=False))
    return nn.ModuleList(layers)


def _run_mlp(net, h):
    for l, layer in enumerate(net):
        h = layer(h)
        if l != len(net) - 1:
            h = F.relu(h, inplace=True)
    return h


class NeRFNetwork(NeRFRenderer):
    _warned_unfused = False

    def __init__(self, encoding="hashgrid", encoding_dir="sphere_harmonics", encoding_bg="hashgrid", num_layers=2,
                 hidden_dim=64, geo_feat_dim=15, num_layers_color=3, hidden_dim_color=64, num_layers_bg=2,
                 hidden_dim_bg=64, bound=1, num_instances=0, num_layers_instance=3, hidden_dim_instance=64,
                 encoder_kwargs=None, **kwargs):
        """``encoder_kwargs`` (not upstream's): overrides for both hash grids' ``get_encoder`` arguments (``num_levels``,
        ``log2_hashmap_size``, ``desired_resolution``, ``base_resolution``); upstream fixes them at 16 levels, 2^19 rows
        and ``desired_resolution = 2048 * bound``, which stay the defaults."""
        super().__init__(bound, **kwargs)
        self.num_layers, self.hidden_dim, self.geo_feat_dim = num_layers, hidden_dim, geo_feat_dim
        enc_kw = dict(desired_resolution=2048 * bound)
        enc_kw.update(encoder_kwargs or {})
        self.encoder, self.in_dim = get_encoder(encoding, **enc_kw)
        self.sigma_net = _mlp(self.in_dim, hidden_dim, 1 + geo_feat_dim, num_layers)
        self.num_layers_color, self.hidden_dim_color = num_layers_color, hidden_dim_color
        self.encoder_dir, self.in_dim_dir = get_encoder(encoding_dir)
        self.color_net = _mlp(self.in_dim_dir + geo_feat_dim, hidden_dim_color, 3, num_layers_color)
        self.num_instances = int(num_instances)
        in_dim_inst = 0
        if self.num_instances:
            self.instance_encoder, in_dim_inst = get_encoder(encoding, **enc_kw)
            self.instance_net = _mlp(in_dim_inst, hidden_dim_instance, self.num_instances, num_layers_instance)
        # row at which the overlapped gradient all-reduce splits a table (levels 8.. first, then 0..7: _table_backward);
        # allreduce_gradients uses the same split for a rank whose backward never ran (a batch without samples)
        for enc in (self.encoder, getattr(self, "instance_encoder", None)):
            desc = getattr(enc, "desc", None)
            if desc is not None and int(desc.num_levels) > 8:
                enc.embeddings._inr_split_row = int(desc.offsets[8])
        self._fusable = (encoding == "hashgrid" and encoding_dir == "sphere_harmonics" and num_layers == 2
                         and hidden_dim == 64 and geo_feat_dim == 15 and num_layers_color == 3
                         and hidden_dim_color == 64 and self.in_dim == 32)
        # the fused instance kernels work on 16-channel MFMA tiles: any K <= 64 runs on them with the output layer
        # zero-padded to the next multiple of 16 (the reference's K is 30 detections + background = 31); the padded
        # channels are cut off again before anything outside this class sees them
        self._k_pad = 16 * ((self.num_instances + 15) // 16)
        self._fusable_inst = (0 < self.num_instances <= 64 and num_layers_instance == 3
                              and hidden_dim_instance == 64 and encoding == "hashgrid" and in_dim_inst == 32)
        # A shape the fused kernels were not written for still runs - HIP encoders + BLAS layers, upstream's default
        # structure - but it is another, slower and untimed product: say so once, with the reason (round-4 verdict 10).
        why = []
        if not self._fusable:
            why.append("NeRF field: " + ", ".join(
                f"{n}={v!r} (fused: {w!r})" for n, v, w in (
                    ("encoding", encoding, "hashgrid"), ("encoding_dir", encoding_dir, "sphere_harmonics"),
                    ("num_layers", num_layers, 2), ("hidden_dim", hidden_dim, 64), ("geo_feat_dim", geo_feat_dim, 15),
                    ("num_layers_color", num_layers_color, 3), ("hidden_dim_color", hidden_dim_color, 64),
                    ("encoder output (levels x features)", self.in_dim, 32)) if v != w))
        if self.num_instances and not self._fusable_inst:
            why.append("instance field: " + ", ".join(
                f"{n}={v!r} (fused: {w})" for n, v, w in (
                    ("num_instances", self.num_instances, "<= 64"), ("num_layers_instance", num_layers_instance, 3),
                    ("hidden_dim_instance", hidden_dim_instance, 64), ("encoder output", in_dim_inst, 32))
                if not (v <= 64 if w == "<= 64" else v == w)))
        self.unfused_reason = "; ".join(why) or None
        if self.unfused_reason and not NeRFNetwork._warned_unfused:
            NeRFNetwork._warned_unfused = True
            import warnings
            warnings.warn("NeRFNetwork: this shape leaves the fused HIP field kernels and runs on the composable path "
                          "(HIP hash-grid / SH encoders + BLAS MLP layers; slower, not the benchmarked path) - "
                          + self.unfused_reason, RuntimeWarning, stacklevel=2)
        self._packed = {}
        self.fused_instance_train = True     # False: HIP encoder + rocBLAS layers (the composable path)
        self.fused_nerf_train = True
        self.fused_nerf_head = True          # False: saved activations + k_nerf_bwd + five weight-gradient launches
        # Opt-in (upstream's `-O` stores and computes in fp16): full-frame inference gathers from a half-precision COPY
        # of the hash table (512 instead of 1024 bytes of table traffic per sample); parameters, training, index
        # arithmetic, blending and the MLPs stay fp32.  Outputs differ from the fp32 table's by ~1e-3 relative.
        self.half_table = False
        self._half_cache = None
        self._half_cache_inst = None
        # Opt-in, the other half of `-O`: full-frame inference runs the MLP GEMMs as ONE fp16 MFMA pass (weights and
        # activations rounded to fp16, fp32 accumulation) instead of the three-pass bf16 split that keeps the default
        # fp32-class.  Outputs within a few 1e-3 of the default's; training and every other path are unaffected.
        self.mlp_fp16 = False
        # Frame path of the fused field: False = one kernel; True = sliced (inr_nerf_forward_table_sliced: the three finest
        # levels by a level-major pre-pass); "auto" (the default) = measure both on the first frames of this network and
        # keep the faster - the sliced path pays where the finest levels have no locality at all (a scene filling a
        # bound >= 4 volume), the fused kernel elsewhere
        self.frame_slices = __import__("os").environ.get("INR_FRAME_SLICES", "auto")
        self._slice_probe = None
        self.last_frame_path = "fused"         # what the last large frame ran on ("fused" | "sliced" [+ " (probing)"])

    # ---- packed MFMA weights (cached until a weight tensor changes) ------------------------------
    def _packed_weights(self, which):
        lib = _lib.load()
        if which in ("nerf", "nerf_f16"):
            ws = [self.sigma_net[0].weight, self.sigma_net[1].weight, self.color_net[0].weight,
                  self.color_net[1].weight, self.color_net[2].weight]
        else:                                  # "instance", "instance_f16"
            ws = [l.weight for l in self.instance_net]
        key = tuple((w.data_ptr(), w._version) for w in ws)
        hit = self._packed.get(which)
        if hit is not None and hit[0] == key:
            return hit[1]
        if which in ("nerf", "instance") and ws[0].is_cuda and not getattr(self, "_host_pack_only", False):
            # weights that live on the device are packed there (same layout and rounding as the host packers): the
            # host route costs five device->host copies, each a synchronisation - ~1 ms per occupancy update of the
            # NeRF stage, whose weights change every step
            f32 = torch.float32
            dev_ws = [w.detach().float().contiguous() for w in ws]
            try:
                if which == "nerf":
                    pf = torch.empty(lib.inr_nerf_packed_floats(), dtype=f32, device=ws[0].device)
                    pb = torch.empty(lib.inr_nerf_bwd_packed_floats(), dtype=f32, device=ws[0].device)
                    check(lib.inr_nerf_pack_weights_device(*[ptr(w, f32, "weight") for w in dev_ws], ptr(pf), ptr(pb),
                                                           stream_ptr()), "nerf_pack_weights_device")
                else:
                    if self._k_pad != self.num_instances:
                        dev_ws[2] = torch.nn.functional.pad(dev_ws[2], (0, 0, 0, self._k_pad - self.num_instances)).contiguous()
                    pf = torch.empty(lib.inr_instance_packed_floats(self._k_pad), dtype=f32, device=ws[0].device)
                    pb = torch.empty(lib.inr_instance_bwd_packed_floats(), dtype=f32, device=ws[0].device)
                    check(lib.inr_instance_pack_weights_device(*[ptr(w, f32, "weight") for w in dev_ws], self._k_pad,
                                                               ptr(pf), ptr(pb), stream_ptr()), "instance_pack_weights_device")
                self._packed[which] = (key, pf)
                return pf
            except RuntimeError:
                self._host_pack_only = True          # the exact-fp32 build has no device packer
        host = [w.detach().float().cpu().contiguous() for w in ws]
        if which.startswith("instance") and self._k_pad != self.num_instances:
            host[2] = torch.nn.functional.pad(host[2], (0, 0, 0, self._k_pad - self.num_instances)).contiguous()
        if which in ("nerf", "nerf_f16"):
            buf = torch.empty(lib.inr_nerf_packed_floats(), dtype=torch.float32)
            pack = lib.inr_nerf_pack_weights if which == "nerf" else lib.inr_nerf_pack_weights_f16
            check(pack(*[host_ptr(h, torch.float32) for h in host], host_ptr(buf, torch.float32)), "nerf_pack_weights")
        else:
            buf = torch.empty(lib.inr_instance_packed_floats(self._k_pad), dtype=torch.float32)
            pack = lib.inr_instance_pack_weights if which == "instance" else lib.inr_instance_pack_weights_f16
            check(pack(*[host_ptr(h, torch.float32) for h in host], self._k_pad, host_ptr(buf, torch.float32)),
                  "instance_pack_weights")
        dev = buf.to(ws[0].device)
        self._packed[which] = (key, dev)
        return dev

    def _half_table(self):
        """fp16 copy of the NeRF table (the opt-in -O paths), refreshed whenever the fp32 master changes."""
        emb = self.encoder.embeddings
        key = (emb.data_ptr(), emb._version)
        if self._half_cache is None or self._half_cache[0] != key:
            self._half_cache = (key, emb.detach().to(torch.float16).contiguous())
        return self._half_cache[1]

    def _needs_grad(self, params):
        return torch.is_grad_enabled() and any(p.requires_grad for p in params)

    def _nerf_params(self):
        return [self.encoder.embeddings] + [l.weight for l in self.sigma_net] + [l.weight for l in self.color_net]

    def _fused_nerf(self, x, d, want_rgb, want_geo, out=None):
        """``out`` = (sigma [M], rgb [M,3]): caller-owned result buffers (``march_ahead(shade=True)`` runs this on a
        side stream, where nothing may be allocated)."""
        lib = _lib.load()
        x = x.contiguous().float()
        M = x.shape[0]
        dev = x.device
        if out is not None:
            sigma, rgb = out
        else:
            sigma = torch.empty(M, dtype=torch.float32, device=dev)
            rgb = torch.empty(M, 3, dtype=torch.float32, device=dev) if want_rgb else None
        geo = torch.empty(M, self.geo_feat_dim, dtype=torch.float32, device=dev) if want_geo else None
        if want_rgb:
            d = d.contiguous().float()
        if (want_rgb and not want_geo and self.half_table and self.mlp_fp16
                and (not self.training or not self.encoder.embeddings.requires_grad)):
            # opt-in -O numerics for a NeRF that is only evaluated: inference, and the FROZEN NeRF of the instance stage
            # (its forward is bound by every XCD pulling the whole table through its fabric port: half the bytes)
            check(lib.inr_nerf_forward_fast(ptr(x, torch.float32, "x"), ptr(d, torch.float32, "d"), M, None, float(self.bound),
                                            ptr(self._half_table(), torch.float16), self.encoder.desc,
                                            ptr(self._packed_weights("nerf_f16")), 1.0, ptr(sigma), ptr(rgb), stream_ptr()),
                  "nerf_forward_fast")
            return sigma, rgb, geo
        check(lib.inr_nerf_forward(ptr(x, torch.float32, "x"), ptr(d, torch.float32, "d", allow_none=not want_rgb),
                                   M, None, float(self.bound), ptr(self.encoder.embeddings.data, torch.float32),
                                   self.encoder.desc, ptr(self._packed_weights("nerf")), 1.0, ptr(sigma),
                                   ptr(rgb, allow_none=True), ptr(geo, allow_none=True), stream_ptr()),
              "nerf_forward")
        return sigma, rgb, geo

    # order of the six probing frames: each path three times, in pairs AB / BA / AB, so that under a FramePipeline
    # (views alternating between two streams) neither path is always timed on the same stream or view parity
    _SLICE_PROBE_ORDER = (False, True, True, False, False, True)

    def _use_slices(self, M):
        """Which frame path this call takes (see ``frame_slices``).  "auto": frames of >= 2^21 samples on the 16-level
        table with hashed fine levels are timed with events on their stream, around the field launches only (the sliced
        path's workspace is in place before the first event) - three calls on each path in the order AB BA AB, nothing
        waits for them - and once all six timings have landed the path with the lower MEDIAN time per sample is kept
        (3 % margin in favour of the fused kernel).  The decision is reopened when the parameters are replaced
        (``load_state_dict``) and when an occupancy update has moved the mean sample count of a training step by more
        than 10 % since it was taken; ``last_frame_path`` says what the last frame ran on and ``render()`` reports it."""
        mode = self.frame_slices
        self.last_frame_path = "fused"
        if mode in (False, 0, "0", "off", "False"):
            return False
        tb = self.encoder.table
        if not (M > 0 and int(tb["num_levels"]) == 16 and bool(tb["hashed"][8:].all())):
            return False
        if mode in (True, 1, "1", "on", "True"):
            self.last_frame_path = "sliced"
            return True
        if M < (1 << 21):                         # "auto": frames only (a batch has nothing to pipeline)
            return False
        # "auto" only ever probes where the sliced path has a chance: a finest level of 8192+ (bound >= 4 under upstream's
        # 2048 x bound rule).  Below that the fused kernel won every measurement (bound 1: 5.1 vs 6.4 ms, bound 2: 3.2 vs
        # 3.8 ms, profiles/r05_NOTES.txt 6) and a probe would put slower frames into somebody's loop for nothing.
        if int(tb["resolutions"][-1]) < 8192:
            return False
        p = self._slice_probe
        if p is not None and p["choice"] is not None:
            # the scene the decision was taken on: a later occupancy update that changes the samples per step by > 10 %
            # means other levels of locality - measure again
            mc = int(getattr(self, "mean_count", 0) or 0)
            if (getattr(self, "iter_density", 0) != p["iter_density"] and p["mean_count"] > 0 and mc > 0
                    and abs(mc - p["mean_count"]) > 0.1 * p["mean_count"]):
                p = self._slice_probe = None
        if p is None:
            p = self._slice_probe = {"calls": 0, "pending": [], "ms": {False: [], True: []}, "choice": None,
                                     "iter_density": 0, "mean_count": 0, "decisions": getattr(self, "_slice_decisions", 0)}
        if p["choice"] is not None:
            self.last_frame_path = "sliced" if p["choice"] else "fused"
            return p["choice"]
        for rec in list(p["pending"]):
            if rec[1].query():
                p["ms"][rec[2]].append(rec[0].elapsed_time(rec[1]) / max(rec[3], 1))
                p["pending"].remove(rec)
        if len(p["ms"][False]) >= 3 and len(p["ms"][True]) >= 3:
            med = {k: sorted(v)[len(v) // 2] for k, v in p["ms"].items()}
            p["choice"] = med[True] < 0.97 * med[False]              # (3 % margin: the fused kernel on a tie)
            p["iter_density"] = getattr(self, "iter_density", 0)
            p["mean_count"] = int(getattr(self, "mean_count", 0) or 0)
            self._slice_decisions = p["decisions"] + 1
            self.last_frame_path = "sliced" if p["choice"] else "fused"
            return p["choice"]
        if p["calls"] >= len(self._SLICE_PROBE_ORDER) + 6:   # timings never landed (a stream nobody synchronises): stay fused
            return False
        use = self._SLICE_PROBE_ORDER[p["calls"] % len(self._SLICE_PROBE_ORDER)]
        p["calls"] += 1
        p["armed"] = (use, M)
        self.last_frame_path = ("sliced" if use else "fused") + " (probing)"
        return use

    def _open_slice_probe(self):
        """Called right before the field launch(es) of a probing frame: the first event of its timing."""
        p = self._slice_probe
        if p and p.get("armed") is not None:
            use, M = p.pop("armed")
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            p["open"] = (e0, use, M)

    def _close_slice_probe(self):
        p = self._slice_probe
        if p and p.get("open") is not None:
            e0, use, M = p.pop("open")
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            p["pending"].append((e0, e1, use, M))

    def load_state_dict(self, state_dict, *args, **kwargs):
        """``nn.Module.load_state_dict`` + the frame-path decision of the OLD parameters is dropped (``frame_slices =
        "auto"`` measures again on the next large frames: another scene has other levels of locality)."""
        out = super().load_state_dict(state_dict, *args, **kwargs)
        self._slice_probe = None
        return out

    @torch.no_grad()
    def sh_table(self, rays_d):
        """Per-ray direction table [N,16] of ``forward_table`` (degree-4 SH in the lane order of the fused kernel)."""
        lib = _lib.load()
        rays_d = rays_d.contiguous().float().view(-1, 3)
        shq = torch.empty(rays_d.shape[0], 16, dtype=torch.float32, device=rays_d.device)
        check(lib.inr_sh_table_q(ptr(rays_d, torch.float32, "rays_d"), rays_d.shape[0], ptr(shq), stream_ptr()),
              "sh_table_q")
        return shq

    def forward_table(self, x01, ray_ids, rays_d, shq=None):
        """forward() for the fused frame path: x01 [M,3] normalised samples and ray_ids int32 [M] from
        ``march_rays_patch(table=True)``, rays_d [N,3].  The direction encoding is evaluated once per RAY into a
        table (``shq``: that table when the caller has already built it); results equal
        ``forward(x, rays_d[ray_ids])``.  None when the fused kernel does not apply."""
        if not self._fusable:
            return None
        lib = _lib.load()
        M = x01.shape[0]
        dev = x01.device
        if shq is None:
            shq = self.sh_table(rays_d)
        sigma = torch.empty(M, dtype=torch.float32, device=dev)
        rgb = torch.empty(M, 3, dtype=torch.float32, device=dev)
        half = self.half_table and not self.training
        if half:
            # opt-in (upstream's -O / fp16 storage): the eval kernel gathers from a half-precision copy of the table,
            # refreshed whenever the fp32 master changes
            self._half_table()
        if self.mlp_fp16 and not self.training:
            # opt-in (the other half of upstream's -O): one fp16 MFMA pass per MLP GEMM instead of the fp32-class split
            table = self._half_cache[1] if half else self.encoder.embeddings.data
            check(lib.inr_nerf_forward_table_fast(ptr(x01, torch.float32, "x01", allow_none=M == 0),
                                                  ptr(ray_ids, torch.int32, "ray_ids", allow_none=M == 0), ptr(shq), M,
                                                  float(self.bound), ptr(table), 1 if half else 0,
                                                  self.encoder.desc, ptr(self._packed_weights("nerf_f16")), 1.0,
                                                  ptr(sigma, allow_none=M == 0), ptr(rgb, allow_none=M == 0), stream_ptr()),
                  "nerf_forward_table_fast")
            return sigma, rgb
        if half:
            check(lib.inr_nerf_forward_table_half(ptr(x01, torch.float32, "x01", allow_none=M == 0),
                                                  ptr(ray_ids, torch.int32, "ray_ids", allow_none=M == 0), ptr(shq), M,
                                                  float(self.bound), ptr(self._half_cache[1], torch.float16),
                                                  self.encoder.desc, ptr(self._packed_weights("nerf")), 1.0,
                                                  ptr(sigma, allow_none=M == 0), ptr(rgb, allow_none=M == 0), stream_ptr()),
                  "nerf_forward_table_half")
            return sigma, rgb
        if self._use_slices(M):
            # sliced frame path (round 5): the three finest levels by a level-major pre-pass (every XCD's L2 then holds
            # the one level the chip is working on), the fused kernel on the other thirteen; same numbers bit for bit
            need = lib.inr_nerf_forward_table_sliced_workspace_bytes(M) // 4
            ws = self.__dict__.get("_slice_ws")
            if ws is None or ws.numel() < need or ws.device != dev:
                # kept between frames (24 bytes per sample: 3.4 GB for a 141 M-sample frame - a fresh block of that size
                # per call cost tens of milliseconds in the allocator on the first frames); consumers run in stream order
                ws = self.__dict__["_slice_ws"] = torch.empty(int(need * 1.25), dtype=torch.float32, device=dev)
            ws.record_stream(torch.cuda.current_stream())      # FramePipeline: views on two streams share it (inside the gate)
            self._open_slice_probe()                            # (after the workspace is in place: no allocation inside a timing)
            check(lib.inr_nerf_forward_table_sliced(ptr(x01, torch.float32, "x01"), ptr(ray_ids, torch.int32, "ray_ids"),
                                                    ptr(shq), M, float(self.bound),
                                                    ptr(self.encoder.embeddings.data, torch.float32), self.encoder.desc,
                                                    ptr(self._packed_weights("nerf")), 1.0, ptr(sigma), ptr(rgb), ptr(ws),
                                                    stream_ptr()), "nerf_forward_table_sliced")
            self._close_slice_probe()
            return sigma, rgb
        self._open_slice_probe()
        check(lib.inr_nerf_forward_table(ptr(x01, torch.float32, "x01", allow_none=M == 0),
                                         ptr(ray_ids, torch.int32, "ray_ids", allow_none=M == 0), ptr(shq), M,
                                         float(self.bound), ptr(self.encoder.embeddings.data, torch.float32),
                                         self.encoder.desc, ptr(self._packed_weights("nerf")), 1.0,
                                         ptr(sigma, allow_none=M == 0), ptr(rgb, allow_none=M == 0), stream_ptr()),
              "nerf_forward_table")
        self._close_slice_probe()
        return sigma, rgb

    @torch.no_grad()
    def forward_dirs(self, x, dirs):
        """x [M,3], dirs [D,3] unit (D <= 8) -> float [M,4] = (rgb averaged over the D view directions, raw density
        logit = log sigma): one gather and one sigma-net pass per point, the colour net once per direction, in one
        launch (the rgb-sigma lattice extraction, ``instance_nerf_amd/extract.py``).  None when the fused kernel does
        not apply."""
        if not self._fusable:
            return None
        lib = _lib.load()
        x = x.contiguous().float()
        M, D = x.shape[0], dirs.shape[0]
        sh = self.encoder_dir(dirs.to(x.device).contiguous().float()).contiguous()          # [D,16] (HIP SH kernel)
        out = torch.empty(M, 4, dtype=torch.float32, device=x.device)
        check(lib.inr_nerf_forward_dirs(ptr(x, torch.float32, "x", allow_none=M == 0), M, float(self.bound),
                                        ptr(self.encoder.embeddings.data, torch.float32), self.encoder.desc,
                                        ptr(self._packed_weights("nerf")), ptr(sh, torch.float32, "sh_dirs"), D,
                                        ptr(out, allow_none=M == 0), stream_ptr()), "nerf_forward_dirs")
        return out

    @torch.no_grad()
    def forward_lattice(self, axes, dirs, logit_min=float("-inf"), sh=None):
        """``forward_dirs`` for the lattice ``axes = (ax_w [W], ax_l [L], ax_h [H])`` (float32, on the device): -> float
        [W, L, H, 4] without a point tensor, the kernel walking the lattice in runs along W (the table's fastest row
        index; 2.4x faster than the h-fastest point list, ``inr_nerf_forward_lattice``).  Coordinates are clamped to
        [-bound, bound]; ``logit_min`` clamps the density logit from below inside the launch; ``sh``: the directions' SH
        rows [D,16] when the caller has them already.  None when the fused kernel does not apply."""
        if not self._fusable:
            return None
        lib = _lib.load()
        ax = [a.contiguous().float() for a in axes]
        W, L, H = (int(a.shape[0]) for a in ax)
        D = dirs.shape[0]
        if sh is None:
            sh = self.encoder_dir(dirs.to(ax[0].device).contiguous().float()).contiguous()  # [D,16] (HIP SH kernel)
        out = torch.empty(W, L, H, 4, dtype=torch.float32, device=ax[0].device)
        if W * L * H:
            check(lib.inr_nerf_forward_lattice(ptr(ax[0], torch.float32, "ax_w"), ptr(ax[1], torch.float32, "ax_l"),
                                               ptr(ax[2], torch.float32, "ax_h"), W, L, H, float(self.bound),
                                               ptr(self.encoder.embeddings.data, torch.float32), self.encoder.desc,
                                               ptr(self._packed_weights("nerf")), ptr(sh, torch.float32, "sh_dirs"), D,
                                               float(logit_min), ptr(out), stream_ptr()), "nerf_forward_lattice")
        return out

    # ---- upstream API -----------------------------------------------------------------------------
    def forward(self, x, d):
        """x [M,3] in [-bound,bound], d [M,3] unit -> sigma [M], color [M,3]."""
        if self._fusable and not self._needs_grad(self._nerf_params()):
            sigma, rgb, _ = self._fused_nerf(x, d, True, False)
            return sigma, rgb
        if (self._fusable and self.fused_nerf_train and x.is_cuda and torch.is_grad_enabled()
                and all(p.requires_grad for p in self._nerf_params()) and not (x.requires_grad or d.requires_grad)):
            fn = _NerfFieldFn if self.fused_nerf_head else _NerfFieldFnUnfused
            return fn.apply(x.contiguous().float(), d.contiguous().float(), self.encoder.embeddings,
                                      self.sigma_net[0].weight, self.sigma_net[1].weight, self.color_net[0].weight,
                                      self.color_net[1].weight, self.color_net[2].weight, self.encoder.desc, self.bound)
        h = _run_mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        sigma = trunc_exp(h[..., 0])
        geo_feat = h[..., 1:]
        h = torch.cat([self.encoder_dir(d), geo_feat], dim=-1)
        color = torch.sigmoid(_run_mlp(self.color_net, h))
        return sigma, color

    def density(self, x):
        """x [M,3] -> {'sigma': [M], 'geo_feat': [M,15]}."""
        params = [self.encoder.embeddings] + [l.weight for l in self.sigma_net]
        if self._fusable and not self._needs_grad(params):
            sigma, _, geo = self._fused_nerf(x, None, False, True)
            return {"sigma": sigma, "geo_feat": geo}
        h = _run_mlp(self.sigma_net, self.encoder(x, bound=self.bound))
        return {"sigma": trunc_exp(h[..., 0]), "geo_feat": h[..., 1:]}

    def density_sigma(self, x):
        params = [self.encoder.embeddings] + [l.weight for l in self.sigma_net]
        if self._fusable and not self._needs_grad(params):
            return self._fused_nerf(x, None, False, False)[0]
        return self.density(x)["sigma"]

    def color(self, x, d, mask=None, geo_feat=None, **kwargs):
        """rgb [M,3]; rows where ``mask`` is False are zero (upstream semantics)."""
        if geo_feat is None:
            geo_feat = self.density(x)["geo_feat"]
        if mask is not None:
            rgbs = torch.zeros(mask.shape[0], 3, dtype=x.dtype, device=x.device)
            if not mask.any():
                return rgbs
            x, d, geo_feat = x[mask], d[mask], geo_feat[mask]
        h = torch.cat([self.encoder_dir(d), geo_feat], dim=-1)
        h = torch.sigmoid(_run_mlp(self.color_net, h))
        if mask is not None:
            rgbs[mask] = h.to(rgbs.dtype)
            return rgbs
        return h

    def instance(self, x):
        """x [M,3] -> raw instance logits [M,K] (None when the network has no instance head)."""
        if not self.num_instances:
            return None
        out = self._instance_for_compositing(x)
        return out if out.shape[1] == self.num_instances else out[:, :self.num_instances].contiguous()

    def _instance_for_compositing(self, x):
        """Logits as the fused kernels produce them: [M, K rounded up to 16] (the extra channels are exactly zero and
        the renderer drops them after compositing), or [M, K] from the composable path."""
        params = [self.instance_encoder.embeddings] + [l.weight for l in self.instance_net]
        K, Kp = self.num_instances, self._k_pad
        if self._fusable_inst and not self._needs_grad(params):
            lib = _lib.load()
            x = x.contiguous().float()
            M = x.shape[0]
            out = torch.empty(M, Kp, dtype=torch.float32, device=x.device)
            check(lib.inr_instance_forward(ptr(x, torch.float32, "x"), M, None, float(self.bound),
                                           ptr(self.instance_encoder.embeddings.data, torch.float32),
                                           self.instance_encoder.desc, ptr(self._packed_weights("instance")),
                                           Kp, ptr(out), stream_ptr()), "instance_forward")
            return out
        if (self._fusable_inst and self.fused_instance_train and x.is_cuda and all(p.requires_grad for p in params)
                and torch.is_grad_enabled() and not x.requires_grad):
            w2 = self.instance_net[2].weight
            if Kp != K:
                w2 = torch.nn.functional.pad(w2, (0, 0, 0, Kp - K))       # autograd cuts its gradient back to [K, 64]
            return _InstanceFieldFn.apply(x.contiguous().float(), self.instance_encoder.embeddings,
                                          self.instance_net[0].weight, self.instance_net[1].weight, w2,
                                          self.instance_encoder.desc, self.bound)
        return _run_mlp(self.instance_net, self.instance_encoder(x, bound=self.bound))

    fused_instance_head = True        # False: _InstanceFieldFn + composite_rays_train(extra=...) (round-2 path)

    def instance_head_available(self, x):
        """The one-node instance head applies: fused kernels, every instance parameter trained, x without gradient."""
        if not (self.num_instances and self._fusable_inst and self.fused_instance_train and self.fused_instance_head):
            return False
        params = [self.instance_encoder.embeddings] + [l.weight for l in self.instance_net]
        return (x.is_cuda and torch.is_grad_enabled() and all(p.requires_grad for p in params) and not x.requires_grad)

    def instance_head_train(self, x, weights, sample_ray, rays, n_dev=None, ce_labels=None, ce_ignore_index=-1):
        """Rendered instance logits [N, K_pad] of a training batch from the ray-major samples x [M,3], their detached
        compositing weights [M] and owning-ray rows int32 [M] (``composite_rays_train(return_weights=True)``) and the
        march's rays [N,3]; ``n_dev``: the march's device-side sample counter (rows beyond it are not evaluated).
        With ``ce_labels`` (int64 [N]) -> (logits, mean cross entropy over the rows whose label is not
        ``ce_ignore_index``), the loss formed inside the compositing launch (see ``_InstanceHeadFn``)."""
        K, Kp = self.num_instances, self._k_pad
        w2 = self.instance_net[2].weight
        if Kp != K:
            w2 = torch.nn.functional.pad(w2, (0, 0, 0, Kp - K))       # autograd cuts its gradient back to [K, 64]
        out, loss = _InstanceHeadFn.apply(x.contiguous().float(), weights, sample_ray, rays, n_dev, ce_labels, K,
                                          ce_ignore_index, self.instance_encoder.embeddings, self.instance_net[0].weight,
                                          self.instance_net[1].weight, w2, self.instance_encoder.desc, self.bound)
        return out if ce_labels is None else (out, loss)

    @torch.no_grad()
    def nerf_render(self, xyzs, deltas, rays, rays_d, T_thresh=1e-4, want_weights=False, normalised=False):
        """Field + compositing with early termination in one launch (inference, patch-interleaved layout).
        -> (weights_sum [N], depth [N], image [N,3], weights [M] | None, evaluated int64[1]); None if not fusable."""
        if not self._fusable:
            return None
        lib = _lib.load()
        N, M = rays.shape[0], xyzs.shape[0]
        dev = rays.device
        ws = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        wbuf = torch.empty(max(M, 1), dtype=torch.float32, device=dev) if want_weights else None
        evaluated = torch.zeros(33, dtype=torch.int64, device=dev)     # [0] evaluated samples, [1..32] the launch's group cursors
        fn, table, packed = lib.inr_nerf_render, self.encoder.embeddings.data, "nerf"
        if self.half_table and self.mlp_fp16 and not self.training:        # opt-in: upstream's -O numerics
            fn, table, packed = lib.inr_nerf_render_fast, self._half_table(), "nerf_f16"
        check(fn(ptr(xyzs, torch.float32, "xyzs", allow_none=M == 0), ptr(deltas, torch.float32, "deltas", allow_none=M == 0),
                 ptr(rays, torch.int32, "rays"), ptr(rays_d.contiguous(), torch.float32, "rays_d"), N, M, float(self.bound),
                 ptr(table), self.encoder.desc, ptr(self._packed_weights(packed)), float(self.density_scale),
                 float(T_thresh), ptr(ws), ptr(depth), ptr(image), ptr(wbuf, allow_none=True), ptr(evaluated),
                 1 if normalised else 0, stream_ptr()), "nerf_render")
        return ws, depth, image, wbuf, evaluated[:1]

    @torch.no_grad()
    def instance_render(self, xyzs, rays, weights, normalised=False):
        """Rendered instance logits [N, K] from the patch-interleaved samples and their compositing weights, with
        the per-sample logits kept on chip (inference only).  None when the fused kernel does not apply.
        normalised: ``xyzs`` are the (x + bound) / (2 bound) coordinates of the table feed."""
        if not (self.num_instances and self._fusable_inst):
            return None
        lib = _lib.load()
        N, M = rays.shape[0], xyzs.shape[0]
        out = torch.empty(N, self._k_pad, dtype=torch.float32, device=rays.device)
        cursors = torch.zeros(32, dtype=torch.int64, device=rays.device)        # the launch's dynamic group schedule
        if self.half_table and self.mlp_fp16 and not self.training:
            # opt-in, upstream's -O numerics for the instance field too: fp16 copy of its table, one-pass fp16 MLP
            emb = self.instance_encoder.embeddings
            key = (emb.data_ptr(), emb._version)
            if self._half_cache_inst is None or self._half_cache_inst[0] != key:
                self._half_cache_inst = (key, emb.detach().to(torch.float16).contiguous())
            check(lib.inr_instance_render_fast(ptr(xyzs, torch.float32, "xyzs", allow_none=M == 0), ptr(rays, torch.int32, "rays"),
                                               ptr(weights, torch.float32, "weights", allow_none=M == 0), N, M,
                                               float(self.bound), ptr(self._half_cache_inst[1], torch.float16),
                                               self.instance_encoder.desc, ptr(self._packed_weights("instance_f16")),
                                               self._k_pad, ptr(out), 1 if normalised else 0, ptr(cursors), stream_ptr()),
                  "instance_render_fast")
            return out if self._k_pad == self.num_instances else out[:, :self.num_instances].contiguous()
        check(lib.inr_instance_render(ptr(xyzs, torch.float32, "xyzs", allow_none=M == 0), ptr(rays, torch.int32, "rays"),
                                      ptr(weights, torch.float32, "weights", allow_none=M == 0), N, M, float(self.bound),
                                      ptr(self.instance_encoder.embeddings.data, torch.float32),
                                      self.instance_encoder.desc, ptr(self._packed_weights("instance")),
                                      self._k_pad, ptr(out), 1 if normalised else 0, ptr(cursors), stream_ptr()),
              "instance_render")
        return out if self._k_pad == self.num_instances else out[:, :self.num_instances].contiguous()

    def get_params(self, lr):
        params = [{"params": self.encoder.parameters(), "lr": lr},
                  {"params": self.sigma_net.parameters(), "lr": lr},
                  {"params": self.encoder_dir.parameters(), "lr": lr},
                  {"params": self.color_net.parameters(), "lr": lr}]
        if self.num_instances:
            params += [{"params": self.instance_encoder.parameters(), "lr": lr},
                       {"params": self.instance_net.parameters(), "lr": lr}]
        return params

    def freeze_nerf(self):
        """Instance-field stage: the NeRF is loaded and frozen, only the instance head trains."""
        for p in self._nerf_params():
            p.requires_grad_(False)
