"""Ray generation, metrics and the Trainer.

``get_rays`` and the ``Trainer`` method names follow upstream ``nerf/utils.py`` of
the reference's un-vendored submodule (/root/reference/.gitmodules:4-6,
README.md:27,59).  The multi-GPU idiom (one process per GPU, NCCL==RCCL process
group, barrier after construction) mirrors the only in-tree example,
/root/reference/nerf_rcnn/run_rcnn.py:755-760,780,823-826.
"""
import math
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import _lib


_PATCH_INDS = {}


SUPER_TILE = int(os.environ.get("INR_SUPER_TILE", "0"))   # pixels; 0 = patches in row-major order


def patch_order(H, W, patch, device, super_tile=None):
    """Flat pixel indices of an HxW image enumerated patch by patch (patch x patch pixels, row-major
    inside a patch).  16 consecutive entries = one 4x4 patch = one group of the patch-interleaved
    sample layout.  With ``super_tile`` (a multiple of ``patch``) the patches themselves are visited
    super-tile by super-tile, which keeps vertically adjacent patches close in the stream.
    Cached per (H, W, patch, super_tile, device)."""
    st = SUPER_TILE if super_tile is None else super_tile
    key = (H, W, patch, st, str(device))
    if key not in _PATCH_INDS:
        jj, ii = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        npx = (W + patch - 1) // patch
        k = ((jj // patch) * npx + (ii // patch))
        if st:
            nsx = (W + st - 1) // st
            pps = st // patch
            k = ((jj // st) * nsx + (ii // st)) * (pps * pps) + ((jj % st) // patch) * pps + ((ii % st) // patch)
        k = k * (patch * patch) + (jj % patch) * patch + (ii % patch)
        _PATCH_INDS[key] = torch.argsort(k).to(device)
    return _PATCH_INDS[key]


@torch.no_grad()
def get_rays(poses, intrinsics, H, W, N=-1, error_map=None, inds=None, patch=0):
    """poses [B,4,4] (camera-to-world), intrinsics (fx,fy,cx,cy) -> dict(rays_o, rays_d [B,N,3], inds [B,N]).

    Pixel centres at +0.5; dir = ((i-cx)/fx, (j-cy)/fy, 1) normalised, rotated by R; N>0 draws N
    random pixels (shared across the batch; with ``error_map`` [B, 128*128] they are drawn per batch element
    in proportion to the map, as upstream).  ``patch=4`` (full images only) enumerates the pixels
    4x4-patch by patch instead of row-major - same rays, an order the renderer's patch-interleaved
    layout turns into compact tiles; scatter results back with ``image.view(-1,3)[inds] = pred``.
    """
    device = poses.device
    B = poses.shape[0]
    fx, fy, cx, cy = intrinsics
    if inds is None and N > 0 and error_map is not None:
        # upstream's importance sampling: draw coarse cells of the [B, 128*128] error map without replacement,
        # then a uniform pixel inside each cell; every batch element gets its own pixels
        inds_coarse = torch.multinomial(error_map.to(device), N, replacement=False)          # [B, N]
        cx_, cy_ = torch.div(inds_coarse, 128, rounding_mode="floor"), inds_coarse % 128
        sx, sy = H / 128, W / 128
        px = (cx_ * sx + torch.rand(B, N, device=device) * sx).long().clamp(max=H - 1)
        py = (cy_ * sy + torch.rand(B, N, device=device) * sy).long().clamp(max=W - 1)
        per_batch = px * W + py
        parts = [get_rays(poses[b:b + 1], intrinsics, H, W, inds=per_batch[b]) for b in range(B)]
        return {"rays_o": torch.cat([p["rays_o"] for p in parts]), "rays_d": torch.cat([p["rays_d"] for p in parts]),
                "inds": per_batch, "inds_coarse": inds_coarse}
    if inds is None:
        if N > 0:
            inds = torch.randint(0, H * W, size=[N], device=device)
        elif patch:
            inds = patch_order(H, W, patch, device)
        else:
            inds = torch.arange(H * W, device=device)
    inds = inds.contiguous().long()
    n = inds.shape[0]
    if poses.is_cuda:
        lib = _lib.load()
        poses = poses.contiguous().float()
        rays_o = torch.empty(B, n, 3, dtype=torch.float32, device=device)
        rays_d = torch.empty(B, n, 3, dtype=torch.float32, device=device)
        _lib.check(lib.inr_get_rays(_lib.ptr(poses, torch.float32, "poses"), B, float(fx), float(fy), float(cx), float(cy),
                                    int(W), _lib.ptr(inds, torch.int64, "inds", allow_none=n == 0), n,
                                    _lib.ptr(rays_o, allow_none=B * n == 0), _lib.ptr(rays_d, allow_none=B * n == 0),
                                    _lib.stream_ptr()), "get_rays")
        return {"rays_o": rays_o, "rays_d": rays_d, "inds": inds[None].expand(B, -1)}
    # host tensors (dataset preparation on the CPU): same arithmetic with torch ops
    i = (inds % W).float() + 0.5
    j = torch.div(inds, W, rounding_mode="floor").float() + 0.5
    xs = (i - cx) / fx
    ys = (j - cy) / fy
    zs = torch.ones_like(xs)
    d = torch.stack([xs, ys, zs], -1)
    d = d / torch.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).unsqueeze(-1)
    R = poses[:, :3, :3].float()
    rays_d = (d[None, :, None, 0] * R[:, None, :, 0] + d[None, :, None, 1] * R[:, None, :, 1]
              + d[None, :, None, 2] * R[:, None, :, 2])
    rays_o = poses[:, None, :3, 3].float().expand_as(rays_d).contiguous()
    return {"rays_o": rays_o, "rays_d": rays_d.contiguous(), "inds": inds[None].expand(B, -1)}


def seed_everything(seed):
    """upstream ``nerf/utils.py::seed_everything``: python, numpy and torch (host + device) generators."""
    import random
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def linear_to_srgb(x):
    """upstream ``nerf/utils.py::linear_to_srgb``."""
    return torch.where(x < 0.0031308, 12.92 * x, 1.055 * x.clamp(min=0.0031308) ** 0.41666 - 0.055)


def srgb_to_linear(x):
    """upstream ``nerf/utils.py::srgb_to_linear``."""
    return torch.where(x < 0.04045, x / 12.92, ((x.clamp(min=0.04045) + 0.055) / 1.055) ** 2.4)


class PSNRMeter:
    def __init__(self):
        self.V, self.N = 0.0, 0

    def clear(self):
        self.V, self.N = 0.0, 0

    def update(self, preds, truths):
        mse = torch.mean((preds.detach().float() - truths.detach().float()) ** 2).item()
        self.V += -10 * math.log10(max(mse, 1e-12))
        self.N += 1

    def measure(self):
        return self.V / max(self.N, 1)

    def report(self):
        return f"PSNR = {self.measure():.6f}"


class MIoUMeter:
    """Mean IoU over the instance ids present in the ground truth (ignore label -1).  Until round 4 the mean ran over
    every id with a non-empty union, i.e. also over ids that only the PREDICTION contains: a handful of stray pixels of
    an id that is not in the view at all counted as a class with IoU 0 - the held-out pose of bench.py's trained scene
    read 0.51 with 97 % of its pixels right and 0.87 over the ids that are actually there (tools/miou_probe.py,
    profiles/r04_NOTES.txt 4).  ``measure(all_predicted=True)`` gives the old figure."""

    def __init__(self, num_classes):
        self.K = num_classes
        self.clear()

    def clear(self):
        self.inter = torch.zeros(self.K, dtype=torch.float64)
        self.union = torch.zeros(self.K, dtype=torch.float64)
        self.truth = torch.zeros(self.K, dtype=torch.float64)

    def update(self, pred_ids, true_ids):
        p, t = pred_ids.reshape(-1).cpu(), true_ids.reshape(-1).cpu()
        keep = t >= 0
        p, t = p[keep], t[keep]
        for k in range(self.K):
            pk, tk = p == k, t == k
            self.inter[k] += (pk & tk).sum()
            self.union[k] += (pk | tk).sum()
            self.truth[k] += tk.sum()

    def measure(self, all_predicted=False):
        present = self.union > 0 if all_predicted else self.truth > 0
        return float((self.inter[present] / self.union[present]).mean()) if present.any() else 0.0

    def measure_both(self):
        """Both definitions under distinct names (round-4 advisor: the default changed in round 4 and every consumer
        silently switched to the more lenient figure): ``miou_gt_ids`` - mean over the ids present in the ground truth;
        ``miou_all_ids`` - mean over every id with a non-empty union, so stray predicted ids count as classes with IoU 0
        (the figure of rounds 1-3: a regression that hallucinates ids shows here)."""
        return {"miou_gt_ids": self.measure(False), "miou_all_ids": self.measure(True)}

    def report(self):
        b = self.measure_both()
        return f"mIoU = {b['miou_gt_ids']:.6f} (over ground-truth ids; {b['miou_all_ids']:.6f} counting ids only predicted)"


class FusedAdam(torch.optim.Optimizer):
    """Adam(betas, eps) with one fused HIP sweep for all tensors of a step (reads p,g,m,v; writes p,m,v once).

    A ``torch.optim.Optimizer``: ``param_groups``, ``zero_grad``, ``state_dict`` / ``load_state_dict`` are the base
    class's, the per-parameter state is torch.optim.Adam's (``step``, ``exp_avg``, ``exp_avg_sq``) - so upstream's
    checkpoints (a torch Adam state dict under 'optimizer') load, what this class saves loads into torch's Adam, and
    upstream's ``lr_scheduler(optimizer)`` (a ``LambdaLR``) accepts it.  Parameters without a gradient (frozen NeRF in
    the instance stage) are skipped.
    """

    def __init__(self, params, lr=1e-2, betas=(0.9, 0.99), eps=1e-15):
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self.step_count = 0
        self._ema = None

    # betas / eps: one value for the whole optimiser (the kernel takes them once per launch)
    @property
    def betas(self):
        return tuple(self.param_groups[0]["betas"])

    @property
    def eps(self):
        return self.param_groups[0]["eps"]

    def _moments(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["step"] = torch.tensor(float(self.step_count))
            st["exp_avg"] = torch.zeros_like(p.data)
            st["exp_avg_sq"] = torch.zeros_like(p.data)
        return st

    def state_dict(self):
        # torch keeps one 'step' tensor per parameter; here all trained tensors step together, so the per-parameter
        # entries are only materialised when a state dict is asked for (not 4-9 host tensors per training step)
        for st in self.state.values():
            if "exp_avg" in st:
                st["step"] = torch.tensor(float(self.step_count))
        return super().state_dict()

    def load_state_dict(self, sd):
        """torch.optim layout (own checkpoints and upstream's Adam checkpoints); round 1's private layout
        ({'step', 'lrs', 'state': {i: (m, v)}}) is converted first."""
        if "param_groups" not in sd:
            flat = [p for g in self.param_groups for p in g["params"]]
            for g, lr in zip(self.param_groups, sd["lrs"]):
                g["lr"] = lr
            for i, (m, v) in sd["state"].items():
                p = flat[int(i)]
                self.state[p] = {"step": torch.tensor(float(sd["step"])), "exp_avg": m.to(p.device), "exp_avg_sq": v.to(p.device)}
            self.step_count = int(sd["step"])
            return
        super().load_state_dict(sd)
        steps = [int(float(st["step"])) for st in self.state.values() if "step" in st]
        self.step_count = max(steps) if steps else 0
        for p, st in self.state.items():                     # the kernel wants contiguous fp32 moments
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k] = st[k].to(p.device, torch.float32).contiguous()

    def _jobs(self):
        """(tensor to update, gradient, exp_avg, exp_avg_sq, group, owner parameter, (first flat element, count) or None).
        A parameter whose gradient arrived as reduce-scattered pieces (``grad_sync`` schedule "reduce_scatter":
        ``_inr_grad_shards``) is updated on this rank's pieces only - flat views of the parameter and its moments."""
        jobs = []
        for g in self.param_groups:
            for p in g["params"]:
                if not p.requires_grad:
                    continue
                pieces = getattr(p, "_inr_grad_shards", None)
                if pieces:
                    st = self._moments(p)
                    flat = lambda t: t.view(-1)
                    for piece in pieces:
                        a, n = piece["own"], piece["grad"].numel()
                        if n == 0:              # a rank beyond the end of a padded slice owns nothing of it
                            continue
                        jobs.append((flat(p.data)[a:a + n], piece["grad"], flat(st["exp_avg"])[a:a + n],
                                     flat(st["exp_avg_sq"])[a:a + n], g, p, (a, n)))
                    continue
                if p.grad is None:
                    continue
                st = self._moments(p)
                jobs.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], g, p, None))
        return jobs

    def sync_shards(self):
        """Reduce-scatter schedule: every rank advances the moments of its own rows only; before the state is saved (or
        the schedule is switched off) the pieces are gathered so that every rank holds the complete moments.  A
        collective: all ranks call it."""
        for g in self.param_groups:
            for p in g["params"]:
                layout = getattr(p, "_inr_shard_layout", None)
                if layout and p in self.state and "exp_avg" in self.state[p]:
                    for k in ("exp_avg", "exp_avg_sq"):
                        grad_sync.allgather_pieces(self.state[p][k], layout)

    @torch.no_grad()
    def attach_ema(self, ema):
        """The parameter EMA (``ParamEMA``) is then advanced inside the optimiser's launch, while the new parameter is
        in registers (7 + 2 instead of 7 + 3 memory streams per element); the Trainer's ``ema.update()`` that follows
        the step finds nothing left to do.  ``None`` detaches."""
        self._ema = ema

    def step(self, closure=None, grad_scale=1.0):
        """One launch for all tensors of the step (inr_adam_step_multi, 16 tensors per call)."""
        loss = closure() if closure is not None else None
        self.step_impl(grad_scale)
        return loss

    def zero_grad(self, set_to_none=True):
        """``torch.optim.Optimizer.zero_grad`` without its per-call bookkeeping (40 us -> 2 us per training step; the
        step is ~15 launches of ~5 us each, the host must not be the slower side)."""
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none or p.grad is None:
                    p.grad = None
                else:
                    p.grad.detach_().zero_()

    @torch.no_grad()
    def step_impl(self, grad_scale=1.0):
        """The body of ``step`` - callable directly (``Trainer.train_one_step`` does): torch.optim wraps ``step`` in a
        profiler range and hook dispatch that cost more host time than this function."""
        import ctypes
        lib = _lib.load()
        self.step_count += 1
        jobs = []
        owners = []
        for p, grad, m, v, g, owner, piece in self._jobs():
            grad = grad if grad.is_contiguous() else grad.contiguous()
            for t, name in ((p, "param"), (grad, "grad")):        # the moments are this class's own fp32 buffers
                if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                    _lib.ptr(t, torch.float32, name)               # raises with the reason
            jobs.append((p, grad, m, v, float(g["lr"]), owner, piece))
            if not any(o is owner for o in owners):
                owners.append(owner)
        ema = self._ema
        shadow_of = ema.begin_fused_update() if ema is not None else None       # {id(param): shadow}, weight set

        def shadow_ptr(j):                  # the shadow rows that go with the job's tensor (a piece of a sharded table)
            sh = shadow_of.get(id(j[5]))
            if sh is None:
                return None
            if j[6] is not None:
                sh = sh.view(-1)[j[6][0]:j[6][0] + j[6][1]]
            return sh.data_ptr()
        for i in range(0, len(jobs), 16):
            chunk = jobs[i:i + 16]
            n = len(chunk)
            arr = lambda k: (ctypes.c_void_p * n)(*[j[k].data_ptr() for j in chunk])
            numels = (ctypes.c_int64 * n)(*[j[0].numel() for j in chunk])
            lrs = (ctypes.c_float * n)(*[j[4] for j in chunk])
            if shadow_of is None:
                _lib.check(lib.inr_adam_step_multi(n, arr(0), arr(1), arr(2), arr(3), numels, lrs, self.betas[0],
                                                   self.betas[1], self.eps, self.step_count, float(grad_scale),
                                                   _lib.stream_ptr()), "adam_step_multi")
            else:
                shadows = (ctypes.c_void_p * n)(*[shadow_ptr(j) for j in chunk])
                _lib.check(lib.inr_adam_ema_step_multi(n, arr(0), arr(1), arr(2), arr(3), numels, lrs, self.betas[0],
                                                       self.betas[1], self.eps, self.step_count, float(grad_scale),
                                                       shadows, float(ema.fused_weight), _lib.stream_ptr()),
                           "adam_ema_step_multi")
        if ema is not None:
            ema.end_fused_update({id(o) for o in owners})
        for p in owners:
            # the C ABI wrote p in place behind autograd's back: bump the version counter so
            # cached MFMA-packed weights (NeRFNetwork._packed_weights) are refreshed
            torch.autograd.graph.increment_version(p)

    # -- hipGraph support: the step-dependent scalars live in a device tensor --------------------------------
    def refresh_hyper(self):
        """Advance the step counter and write [eps_t, lr_t..., ema weight] for the NEXT launch of ``step_captured`` (a
        tiny kernel on the stream whose arguments carry the values: call it right before the launch / graph replay)."""
        import ctypes
        lib = _lib.load()
        self.step_count += 1
        lrs = [float(j[4]["lr"]) for j in self._hyper_jobs]
        arr = (ctypes.c_float * len(lrs))(*lrs)
        if self._ema is not None:
            self._ema.begin_fused_update()                # counts the update, sets fused_weight = 1 - decay_t
            _lib.check(lib.inr_adam_set_hyper_ema(arr, len(lrs), self.betas[0], self.betas[1], self.eps, self.step_count,
                                                  float(self._ema.fused_weight), _lib.ptr(self._hyper_dev),
                                                  _lib.stream_ptr()), "adam_set_hyper_ema")
        else:
            _lib.check(lib.inr_adam_set_hyper(arr, len(lrs), self.betas[0], self.betas[1], self.eps, self.step_count,
                                              _lib.ptr(self._hyper_dev), _lib.stream_ptr()), "adam_set_hyper")

    def hyper_tensor(self, device):
        if getattr(self, "_hyper_dev", None) is None:
            self._hyper_dev = torch.zeros(18, dtype=torch.float32, device=device)
        return self._hyper_dev

    @torch.no_grad()
    def step_captured(self):
        """``step()`` for use inside a captured graph: identical arithmetic, learning rate, bias correction and - with an
        attached ``ParamEMA`` - the average's weight read from device memory (``refresh_hyper``).  At most 16 tensors."""
        import ctypes
        lib = _lib.load()
        jobs = self._jobs()
        if len(jobs) > 16:
            raise RuntimeError("step_captured handles at most 16 parameter tensors")
        if any(j[6] is not None for j in jobs):
            raise RuntimeError("step_captured does not take reduce-scattered gradient pieces")
        self.hyper_tensor(jobs[0][0].device)
        self._hyper_jobs = jobs
        n = len(jobs)
        for j in jobs:
            for t, name in ((j[0].data, "param"), (j[1], "grad"), (j[2], "exp_avg"), (j[3], "exp_avg_sq")):
                _lib.ptr(t, torch.float32, name)
        arr = lambda k: (ctypes.c_void_p * n)(*[j[k].data_ptr() for j in jobs])
        numels = (ctypes.c_int64 * n)(*[j[0].numel() for j in jobs])
        if self._ema is not None:
            shadow_of = {id(p): sh for p, sh in zip(self._ema.params, self._ema.shadow)}
            shadows = (ctypes.c_void_p * n)(*[(shadow_of[id(j[5])].data_ptr() if id(j[5]) in shadow_of else None)
                                              for j in jobs])
            _lib.check(lib.inr_adam_ema_step_multi_dev(n, arr(0), arr(1), arr(2), arr(3), numels,
                                                       _lib.ptr(self._hyper_dev), self.betas[0], self.betas[1], 1.0,
                                                       shadows, _lib.stream_ptr()), "adam_ema_step_multi_dev")
        else:
            _lib.check(lib.inr_adam_step_multi_dev(n, arr(0), arr(1), arr(2), arr(3), numels, _lib.ptr(self._hyper_dev),
                                                   self.betas[0], self.betas[1], 1.0, _lib.stream_ptr()),
                       "adam_step_multi_dev")

    def bump_versions(self):
        """After a replay of the captured step: the parameters changed behind autograd's back (packed-weight caches key
        on the version), and the attached average has been advanced by the sweep."""
        owners = [j[5] for j in getattr(self, "_hyper_jobs", [])]
        for p in owners:
            torch.autograd.graph.increment_version(p)
        if self._ema is not None:
            self._ema.end_fused_update({id(p) for p in owners})


class _GradSync:
    """Early, overlapped all-reduce of the hash-table gradient (training row).  The table gradient is the
    LAST thing a backward produces and by far the largest message (49 MB per trained grid), so waiting for the whole
    backward before starting the collective leaves the links idle for the entire step.  With more than one rank the
    fused backward functions (nerf/network.py::_table_backward) run the scatter in two level ranges - fine levels
    first - and hand each finished row range to ``reduce_async``: RCCL moves levels 8..15 over xGMI while the CUs
    scatter levels 0..7 and reduce the weight gradients.  ``allreduce_gradients`` then only has to wait."""

    def __init__(self):
        self.world_size = 1
        self.enabled = os.environ.get("INR_GRAD_OVERLAP", "1") != "0"
        # INR_GRAD_DTYPE=bf16: the TABLE gradient travels over the links as bf16 (24.5 instead of 49 MB per trained
        # grid and step; converted on the way out and back, summed by the collective in bf16 - 8 mantissa bits; Adam
        # normalises the gradient's scale, so what changes is the direction of a step by ~0.4 %).  Off by default:
        # the fp32 all-reduce is exact up to summation order.  MLP gradients (a few KB) always go as fp32.
        self.payload = os.environ.get("INR_GRAD_DTYPE", "fp32").lower()
        # INR_GRAD_SCHEDULE=reduce_scatter: every row range of a table gradient is reduce-SCATTERED instead of
        # all-reduced - each rank receives the summed gradient of 1 / world of the rows, runs Adam (and the parameter
        # EMA) on those rows only and the updated rows are all-gathered: the same bytes on the links (an all-reduce IS
        # a reduce-scatter + an all-gather), but the optimiser sweep over a 49 MB table (7-9 memory streams, 340-440 MB)
        # shrinks to 1 / world per rank.  Moments and EMA rows of the other ranks go stale locally and are gathered
        # when somebody needs them (checkpoint, evaluation: ``Trainer._sync_shards``).  FusedAdam only; off by default.
        self.schedule = os.environ.get("INR_GRAD_SCHEDULE", "all_reduce").lower()
        self.sharded_ok = False         # set by allreduce_gradients(sharded=True) callers: the optimiser takes pieces
        self.handles = []
        self.early = {}                 # parameter data_ptr -> (data_ptr, numel) of the gradient whose slices are in flight
        self.pieces = {}                # parameter data_ptr -> [piece]: reduce-scattered row ranges of its gradient

    def active(self):
        return self.enabled and self.world_size > 1 and dist.is_available() and dist.is_initialized()

    def scatter_mode(self, numel=0):
        """The reduce-scatter schedule applies - to EVERY slice of a table gradient, whatever its size: a slice that the
        world size does not divide is padded on the wire and the last ranks' pieces are shorter (round-3 advisor: the
        mode used to be decided per slice from numel % world_size, so two level ranges of one table could differ - with
        3 or 6 ranks and T = 6 119 864 rows - and the all-reduced rows were then never updated)."""
        return self.schedule == "reduce_scatter" and self.sharded_ok and self.world_size > 1

    def reduce_async(self, view, param=None, flat_lo=0):
        """Sum ``view`` (a contiguous slice of a table gradient) over the ranks, asynchronously; ``finish()`` waits.
        ``param`` / ``flat_lo`` (the parameter the slice belongs to and the slice's first flat element): with the
        reduce-scatter schedule the rank receives only its 1 / world of the slice, kept as a piece for the optimiser."""
        if param is not None and self.scatter_mode():
            W, numel = self.world_size, view.numel()
            n = (numel + W - 1) // W                          # elements per rank on the wire
            dt = torch.bfloat16 if self.payload == "bf16" else torch.float32
            if numel == n * W:
                wire = view.reshape(-1).to(dt)                # fp32: a view, nothing is copied
            else:                                             # pad: the tail belongs to nobody
                wire = torch.zeros(n * W, dtype=dt, device=view.device)
                wire[:numel] = view.reshape(-1)
            out = torch.empty(n, dtype=dt, device=view.device)
            h = dist.reduce_scatter_tensor(out, wire, async_op=True)
            rank = dist.get_rank()
            piece = {"lo": int(flat_lo), "n": numel, "own": int(flat_lo) + rank * n, "grad": out, "wire": wire,
                     "count": max(0, min(n, numel - rank * n)), "stride": n}
            self.pieces.setdefault(param.data_ptr(), []).append(piece)
            self.handles.append((h, None, None))
        elif self.payload == "bf16":
            wire = view.to(torch.bfloat16)
            self.handles.append((dist.all_reduce(wire, async_op=True), view, wire))
        else:
            self.handles.append((dist.all_reduce(view, async_op=True), None, None))

    def mark(self, param, grad):
        self.early[param.data_ptr()] = (grad.data_ptr(), grad.numel())

    def finish(self):
        for h, view, wire in self.handles:
            h.wait()
            if wire is not None:
                view.copy_(wire)
        self.handles = []
        for pieces in self.pieces.values():
            for piece in pieces:
                piece.pop("wire", None)
                if piece["grad"].dtype != torch.float32:
                    piece["grad"] = piece["grad"].float()
                if piece["grad"].numel() != piece["count"]:
                    piece["grad"] = piece["grad"][:piece["count"]]      # the wire's padding is not a gradient

    def reset(self):
        self.finish()
        self.early = {}

    def hand_over_pieces(self, params):
        """After ``finish()``: the reduce-scattered pieces move onto their parameters (``_inr_grad_shards`` for
        ``FusedAdam``; ``_inr_shard_layout`` remembers which rows this rank keeps current)."""
        for p in params:
            pieces = self.pieces.pop(p.data_ptr(), None)
            if pieces:
                p._inr_grad_shards = pieces
                p._inr_shard_layout = [(q["lo"], q["n"], q["own"], q["count"], q["stride"]) for q in pieces]
                # the full-size buffer holds this rank's UNREDUCED gradient: nothing may mistake it for the sum
                # (clip_grad_norm_, logging); the optimiser takes the pieces
                p.grad = None
        self.pieces = {}

    @staticmethod
    def allgather_pieces(tensor, layout):
        """Every rank's piece (own, count) of each range (lo, n) of the flat ``tensor`` -> all ranks, in place."""
        flat = tensor.view(-1)
        W = dist.get_world_size()
        for lo, n, own, cnt, stride in layout:
            if stride * W == n:
                dist.all_gather_into_tensor(flat[lo:lo + n], flat[own:own + cnt].clone())
            else:                                             # padded on the wire: the last pieces are shorter
                mine = torch.zeros(stride, dtype=flat.dtype, device=flat.device)
                mine[:cnt] = flat[own:own + cnt]
                full = torch.empty(stride * W, dtype=flat.dtype, device=flat.device)
                dist.all_gather_into_tensor(full, mine)
                flat[lo:lo + n] = full[:n]

    def allgather_params(self, params):
        """After the optimiser step of the reduce-scatter schedule: the updated rows of every rank travel to all."""
        for p in params:
            if getattr(p, "_inr_grad_shards", None):
                self.allgather_pieces(p.data, p._inr_shard_layout)
                p._inr_grad_shards = None
                torch.autograd.graph.increment_version(p)


grad_sync = _GradSync()


def allreduce_gradients(params, world_size, bucket_bytes=64 << 20, average=True, sharded=False):
    """Gradient all-reduce for ray-batch data parallelism: dense fp32 buckets over
    RCCL (backend 'nccl' on ROCm) or gloo.  The hash-table gradient (49 MB) goes as ONE message - or is already in
    flight, started from inside the backward in two level ranges (``grad_sync``); small MLP gradients are flattened
    into one bucket.  ``average=False`` leaves the SUM in the ``.grad`` tensors: the caller folds 1 / world_size into
    the optimiser (``FusedAdam.step(grad_scale=1 / world_size)``) instead of paying a read-modify-write sweep over
    every gradient (98 MB of traffic per trained table).  -> the factor the caller still has to apply (1 or
    1 / world_size).  ``sharded=True`` (the caller's optimiser is ``FusedAdam`` and ``grad_sync.allgather_params`` follows
    its step): with ``grad_sync.schedule == "reduce_scatter"`` the table gradients come back as this rank's pieces
    (``param._inr_grad_shards``) instead of full sums in ``param.grad``."""
    if world_size <= 1:
        return 1.0
    if sharded != grad_sync.sharded_ok and grad_sync.early:
        raise RuntimeError("grad_sync.sharded_ok must be set before the backward (Trainer does it at construction)")
    grad_sync.sharded_ok = bool(sharded)
    # Every rank must issue the SAME sequence of collectives whatever its batch looked like.  A rank whose rays all
    # missed the volume has no gradient at all (or an autograd function that never ran): it contributes zeros, and a
    # table it did not hand to the collective from inside its backward goes in the same two level ranges, in the same
    # order, that the other ranks used there.
    params = [p for p in params if p.requires_grad]
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    grads = [p.grad for p in params]
    small = []
    handles = []
    for p in params:
        g = p.grad
        e = grad_sync.early.get(p.data_ptr())
        if e is not None:
            if e != (g.data_ptr(), g.numel()):
                raise RuntimeError("the table gradient was replaced while its all-reduce was in flight "
                                   "(INR_GRAD_OVERLAP=0 turns the early all-reduce off)")
            continue                    # its row ranges were handed to the collective during the backward
        split = getattr(p, "_inr_split_row", 0)
        if grad_sync.active() and split:
            g = g if g.is_contiguous() else g.contiguous()
            p.grad = g
            w = g.shape[1] if g.dim() > 1 else 1
            grad_sync.reduce_async(g[split:], p, split * w)       # fine levels first, as in _table_backward
            grad_sync.reduce_async(g[:split], p, 0)
        elif g.numel() * 4 >= bucket_bytes // 4:
            if g.is_contiguous():
                grad_sync.reduce_async(g, p, 0)     # a table gradient: with the configured payload type
            else:
                handles.append(dist.all_reduce(g, async_op=True))
        else:
            small.append(g)
    grads = [p.grad for p in params]
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        dist.all_reduce(flat)
        off = 0
        for g in small:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    for h in handles:
        h.wait()
    grad_sync.reset()
    grad_sync.hand_over_pieces(params)
    if not average:
        return 1.0 / world_size
    for p in params:
        for piece in getattr(p, "_inr_grad_shards", None) or ():
            piece["grad"].div_(world_size)
        if not getattr(p, "_inr_grad_shards", None):
            p.grad.div_(world_size)
    return 1.0


def shard_range(n_rays, rank, world_size, align=16):
    """Contiguous ray range [lo, hi) of this rank; boundaries are multiples of ``align`` (the 16-ray groups of the
    patch-interleaved sample layout must not straddle two GPUs)."""
    groups = (n_rays + align - 1) // align
    lo = min(n_rays, (groups * rank // world_size) * align)
    hi = min(n_rays, (groups * (rank + 1) // world_size) * align)
    return lo, hi


def shard_indices(n_rays, rank, world_size, device=None, chunk_groups=64, align=16):
    """Ray indices of this rank when a frame is split over the GPUs of a node: chunks of ``chunk_groups`` whole 16-ray
    groups (1024 rays) dealt round robin.  Contiguous ranges (``shard_range``) are NOT balanced - the top rows of a
    frame see the ceiling, the middle rows the whole room, and the frame is as slow as its slowest rank (the same
    effect the XCD schedule of the group-owned kernels had, profiles/r02_NOTES.txt 20); interleaved chunks are, and a
    chunk is still a compact run of patches."""
    chunk = chunk_groups * align
    n_chunks = (n_rays + chunk - 1) // chunk
    if rank >= n_chunks:
        return torch.empty(0, dtype=torch.int64, device=device)
    mine = torch.arange(rank, n_chunks, world_size, device=device)
    idx = (mine[:, None] * chunk + torch.arange(chunk, device=device)[None, :]).reshape(-1)
    return idx[idx < n_rays]


@torch.no_grad()
def render_sharded(model, rays_o, rays_d, rank=0, world_size=1, keys=("image", "depth", "weights_sum", "instance"),
                   **kwargs):
    """One frame split over the GPUs of a node (render row): rays are independent, parameters and the
    occupancy bitfield are replicated, every rank renders its chunks of the ray list (``shard_indices``: 1024-ray
    chunks dealt round robin, so that every rank gets the same mix of cheap and expensive image regions) and the
    results are all-gathered (RCCL; 16 + 4K bytes per ray, nothing on the critical path of the kernels) and put back in
    the caller's ray order.  rays_o, rays_d [B,N,3] -> dict of full [B,N,...] tensors on every rank."""
    B, N = rays_o.shape[:2]
    if world_size == 1:
        return model.render(rays_o, rays_d, **kwargs)
    index = [shard_indices(N, r, world_size, rays_o.device) for r in range(world_size)]
    if any(i.numel() == 0 for i in index):
        raise ValueError(f"{N} rays are fewer than one 1024-ray chunk per rank for {world_size} ranks")
    out = model.render(rays_o[:, index[rank]].contiguous(), rays_d[:, index[rank]].contiguous(), **kwargs)
    result = {}
    for k in keys:
        if k not in out:
            continue
        tail = out[k].shape[2:]
        parts = [torch.empty(B, i.numel(), *tail, dtype=out[k].dtype, device=rays_o.device) for i in index]
        mine = out[k].contiguous()
        if len({i.numel() for i in index}) == 1:
            dist.all_gather(parts, mine)
        else:
            _all_gather_uneven(parts, mine, rank)
        full = torch.empty(B, N, *tail, dtype=out[k].dtype, device=rays_o.device)
        for i, part in zip(index, parts):
            full[:, i] = part
        result[k] = full
    return result


def _all_gather_uneven(parts, mine, rank):
    """all_gather for shards of different length: one broadcast per rank (world_size <= 8 on a node)."""
    for r, buf in enumerate(parts):
        if r == rank:
            buf.copy_(mine)
        dist.broadcast(buf, src=r)


class ParamEMA:
    """Exponential moving average of the trained parameters, the subset of ``torch_ema.ExponentialMovingAverage``
    upstream's Trainer uses (``update`` after every optimiser step, ``store``/``copy_to``/``restore`` around
    evaluation, ``state_dict``): shadow += (1 - d) * (param - shadow), d = min(decay, (1 + n) / (10 + n))."""

    def __init__(self, params, decay):
        self.params = [p for p in params if p.requires_grad]
        self.decay, self.num_updates = float(decay), 0
        self.shadow = [p.detach().clone() for p in self.params]
        self.backup = None
        self._fused_done = False
        self.fused_weight = 0.0

    @torch.no_grad()
    def update(self):
        if self._fused_done:                 # FusedAdam.attach_ema: the optimiser's launch has already advanced it
            self._fused_done = False
            return
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        torch._foreach_lerp_(self.shadow, [p.detach() for p in self.params], 1.0 - d)

    # -- update inside the optimiser's launch (FusedAdam.attach_ema) --
    def begin_fused_update(self):
        """Counts the update, sets ``fused_weight`` = 1 - d and returns {id(param): shadow}."""
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        self.fused_weight = 1.0 - d
        return {id(p): s for p, s in zip(self.params, self.shadow)}

    @torch.no_grad()
    def end_fused_update(self, done_ids):
        """Parameters the optimiser did not touch this step (no gradient) still move their average."""
        rest = [(s, p.detach()) for p, s in zip(self.params, self.shadow) if id(p) not in done_ids]
        if rest:
            torch._foreach_lerp_([s for s, _ in rest], [p for _, p in rest], self.fused_weight)
        self._fused_done = True

    @torch.no_grad()
    def sync_shards(self):
        """Reduce-scatter schedule (``grad_sync``): the fused update advanced only this rank's rows of a sharded table's
        average; gather the other ranks' rows.  A collective: all ranks call it (``Trainer._sync_shards``)."""
        for p, s in zip(self.params, self.shadow):
            layout = getattr(p, "_inr_shard_layout", None)
            if layout:
                grad_sync.allgather_pieces(s, layout)

    @torch.no_grad()
    def store(self):
        self.backup = [p.detach().clone() for p in self.params]

    @torch.no_grad()
    def copy_to(self):
        for p, s in zip(self.params, self.shadow):
            p.copy_(s)                       # through torch: bumps the version counter (packed-weight caches refresh)

    @torch.no_grad()
    def restore(self):
        for p, b in zip(self.params, self.backup):
            p.copy_(b)
        self.backup = None

    def state_dict(self):
        return {"decay": self.decay, "num_updates": self.num_updates, "shadow_params": self.shadow}

    def load_state_dict(self, sd):
        self.decay, self.num_updates = sd["decay"], sd["num_updates"]
        for s, v in zip(self.shadow, sd["shadow_params"]):
            s.copy_(v.to(s.device))


def copy_tensors(pairs):
    """[(dst, src)] device tensors of equal size: ONE launch for up to 8 of them (``inr_copy_multi``) when they are
    contiguous GPU tensors of the same dtype with 4-byte-multiple sizes, ``dst.copy_(src)`` otherwise."""
    import ctypes
    fast = [(d, s) for d, s in pairs if d.is_cuda and s.is_cuda and d.dtype == s.dtype and d.is_contiguous()
            and s.is_contiguous() and d.numel() == s.numel() and (d.numel() * d.element_size()) % 4 == 0
            and d.device == s.device and d.data_ptr() % 4 == 0 and s.data_ptr() % 4 == 0]
    for d, s in pairs:
        if not any(d is f[0] for f in fast):
            d.copy_(s, non_blocking=True)
    lib = _lib.load()
    for i in range(0, len(fast), 8):
        chunk = fast[i:i + 8]
        n = len(chunk)
        _lib.check(lib.inr_copy_multi(n, (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in chunk]),
                                      (ctypes.c_void_p * n)(*[s.data_ptr() for _, s in chunk]),
                                      (ctypes.c_int64 * n)(*[d.numel() * d.element_size() for d, _ in chunk]),
                                      _lib.stream_ptr()), "copy_multi")


class Trainer:
    """Counterpart of upstream's ``Trainer`` (``nerf/utils.py`` of the un-vendored submodule) for the two stages the
    reference runs: NeRF training (MSE on rgb) and instance-field training (NeRF frozen,
    cross entropy of rendered logits vs. matched-mask ids, ignore -1).

    The constructor takes upstream's arguments in upstream's order and meaning - ``optimizer`` is called with the model,
    ``lr_scheduler`` with the optimizer, ``metrics`` are meters with ``update / measure / report / clear``,
    ``use_checkpoint`` in {"latest", "latest_model", "best", "scratch", <path>} - so the construction in upstream's
    ``main_nerf.py`` works unchanged; ``fp16=True`` (upstream's ``-O``) keeps everything TRAINED in fp32 (no autocast,
    no GradScaler) and runs every field that is only evaluated with -O's numerics (``NeRFNetwork.half_table`` +
    ``NeRFNetwork.mlp_fp16``: fp16 table copy, single-pass fp16 MLP): the evaluation / test renders and, at train time,
    the frozen NeRF of the instance stage;
    ``use_tensorboardX`` is accepted and ignored.  The DEFAULTS are upstream's too [U: recalled, the submodule is not
    vendored] - ``use_checkpoint="latest"`` (a workspace that holds ``<name>_ep*.pth`` files is resumed from),
    ``scheduler_update_every_step=False`` (a caller-supplied scheduler is stepped once per epoch unless the caller
    says otherwise, as upstream's main script does) - rounds 1-2 had "scratch" / True here, which silently changed
    what a script relying on the defaults did (round-2 advisor).  Same methods and step semantics (``train_step`` / ``eval_step`` /
    ``test_step`` / ``train`` / ``evaluate`` / ``test`` / ``save_checkpoint`` / ``load_checkpoint``).  Defaults when the
    caller passes neither optimizer nor scheduler are upstream's main-script values: Adam(betas .9/.99, eps 1e-15) as
    ``FusedAdam``, lr * 0.1^(step/iters), occupancy update every ``opt.update_extra_interval`` (16) steps.

    Keyword-only extensions: ``stage`` ("nerf" | "instance"), ``lr``, ``iters``, ``fused_adam``,
    ``update_extra_interval``, ``use_graph``, ``look_ahead``, ``prune_ignored``, ``fixed_point_grad``.  One process per GPU; with world_size > 1 each rank draws its own rays and
    gradients are all-reduced (RCCL), the table gradient from inside the backward (``grad_sync``).
    """

    def __init__(self, name, opt, model, criterion=None, optimizer=None, ema_decay=None, lr_scheduler=None, metrics=None,
                 local_rank=0, world_size=1, device=None, mute=False, fp16=False, eval_interval=1, max_keep_ckpt=2,
                 workspace="workspace", best_mode="min", use_loss_as_metric=True, report_metric_at_train=False,
                 use_checkpoint="latest", use_tensorboardX=True, scheduler_update_every_step=False, *,
                 lr=1e-2, iters=30000, fused_adam=True, stage="nerf", update_extra_interval=None, use_graph=False,
                 look_ahead=False, shade_ahead=None, prune_ignored=True, fixed_point_grad=None):
        self.name, self.opt, self.model = name, opt, model
        self.world_size, self.local_rank = world_size, local_rank
        grad_sync.world_size = world_size
        self.device = device or (torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu"))
        self.stage = stage
        self.workspace = workspace
        # upstream's fp16 / -O flag.  Everything that is TRAINED stays fp32 here - parameters, gradients, Adam's moments,
        # the EMA: no autocast, no GradScaler (nothing trained is ever stored or accumulated in half precision, so there
        # is nothing to scale).  What -O switches on are -O's numerics for every field that is only EVALUATED: the
        # evaluation / test renders, and at train time the FROZEN NeRF of the instance stage (NeRFNetwork.half_table +
        # NeRFNetwork.mlp_fp16 -> inr_nerf_forward_fast: the fp16 copy of its table halves the bytes every XCD pulls
        # through its fabric port for a random-ray batch, the MLP runs as one fp16 MFMA pass; instance step 0.872 ->
        # 0.848 ms, loss within 1 % of the fp32 step over 50 steps:
        # tests/test_gpu_parity.py::test_train_time_O_keeps_the_instance_stage_within_one_percent)
        self.mute, self.fp16 = mute, bool(fp16)
        if self.fp16 and hasattr(model, "half_table"):
            model.half_table = True
            model.mlp_fp16 = True
        self.metrics = list(metrics) if metrics else []
        self.eval_interval, self.max_keep_ckpt = eval_interval, max_keep_ckpt
        self.best_mode, self.use_loss_as_metric = best_mode, use_loss_as_metric
        self.report_metric_at_train = report_metric_at_train
        self.scheduler_update_every_step = scheduler_update_every_step
        if update_extra_interval is None:
            update_extra_interval = getattr(opt, "update_extra_interval", 16) if opt is not None else 16
        self.update_extra_interval = update_extra_interval
        self._default_criterion = criterion is None      # upstream's MSELoss(reduction='none') + .mean(): fusable
        self.criterion = criterion or (torch.nn.MSELoss(reduction="none") if stage == "nerf" else None)
        model.to(self.device)
        if stage == "instance":
            model.freeze_nerf()
        if optimizer is not None:
            self.optimizer = optimizer(self.model)             # upstream: lambda model: Adam(model.get_params(lr), ...)
        else:
            groups = [{"params": [p for p in g["params"] if p.requires_grad], "lr": g["lr"]} for g in model.get_params(lr)]
            groups = [g for g in groups if g["params"]]
            if fused_adam and self.device.type == "cuda":
                self.optimizer = FusedAdam(groups, lr=lr, betas=(0.9, 0.99), eps=1e-15)
            else:
                self.optimizer = torch.optim.Adam(groups, lr=lr, betas=(0.9, 0.99), eps=1e-15)
        # the reduce-scatter gradient schedule (grad_sync.schedule) needs an optimiser that takes gradient pieces and a
        # captured graph cannot hold its collectives: FusedAdam, eager steps
        grad_sync.sharded_ok = isinstance(self.optimizer, FusedAdam) and world_size > 1 and not use_graph
        self._ahead, self._side_stream = None, None      # the next batch's march, queued under this step's backward
        # opt-in: train_one_epoch hands every step the NEXT batch too (train_one_step(data, next_data)).  -35 us per
        # step (4 %) in tools/train_probe.py; in bench.py's loop (EMA, occupancy updates, per-step events) the extra
        # ~0.2 ms of host work per step made the host the limit on most boxes: off by default (r03 notes 16)
        self.look_ahead = bool(look_ahead)
        # opt-in on top of look_ahead, instance stage: the look-ahead also runs the frozen NeRF's forward and the weight
        # compositing of the next batch (NeRFRenderer.march_ahead(shade=True)).  Bit-identical.  What it buys is small:
        # beside the table-gradient scatter the gather-bound forward takes 350-415 us instead of 100, the scatter
        # 370 -> 456 and the optimiser sweep 51 -> 117 - all three queue on the L2's fabric ports
        # (profiles/r04_NOTES.txt 2); the march (VALU-bound, bitfield in the L2) is what really hides.
        # (default: on exactly when the step is the captured pipeline, where it measured -4..-6 % on the step in spite of
        # that; in the eager look-ahead it only adds host work)
        self.shade_ahead = bool(use_graph and look_ahead) if shade_ahead is None else bool(shade_ahead)
        self.pipe_fork = os.environ.get("INR_PIPE_FORK", "scatter")     # probe switch: where the captured step forks
        # instance stage: rays whose matched-mask label is -1 carry neither loss nor gradient (cross entropy with
        # ignore_index -1), so they are reported to the marcher as misses and cost nothing (render(ce_prune=True)):
        # same loss, same gradients; the step's `pred` rows of those rays are zeros instead of rendered logits
        self.prune_ignored = bool(prune_ignored)
        # opt-in: the table gradient summed as int32 fixed point (nerf/network.py::FX_GRAD - faster scatter, bit-reproducible
        # steps, rows below the level's quantum get no gradient).  A PROCESS-WIDE switch of the backward functions: None
        # leaves it as INR_FX_GRAD set it (default off).
        if fixed_point_grad is not None:
            from . import network as _network
            _network.FX_GRAD = 64 if fixed_point_grad == 64 else (32 if fixed_point_grad else 0)
        self.iters = iters
        # upstream: lr_scheduler = lambda optimizer: LambdaLR(optimizer, lambda it: 0.1 ** min(it / opt.iters, 1)),
        # stepped after every optimiser step; without one, exactly that rule is applied to the param groups
        self.lr_scheduler = lr_scheduler(self.optimizer) if lr_scheduler is not None else None
        # upstream: ema_decay=0.95 from its main scripts; evaluation then runs on the averaged parameters
        trained = [p for g in self.optimizer.param_groups for p in g["params"] if p.requires_grad]
        self.ema = ParamEMA(trained, ema_decay) if ema_decay is not None else None
        self.base_lrs = [g["lr"] for g in self.optimizer.param_groups]
        # use_graph: in the steady state (sample buffers sized by mean_count, no host read-back) the whole step -
        # march, fields, compositing, loss, backward, Adam - is captured once per buffer size as a hipGraph and
        # replayed; one process, FusedAdam only, built-in learning-rate rule.  Falls back to the eager step otherwise.
        self.use_graph = (bool(use_graph) and world_size == 1 and isinstance(self.optimizer, FusedAdam)
                          and self.lr_scheduler is None)
        if self.ema is not None and isinstance(self.optimizer, FusedAdam):
            self.optimizer.attach_ema(self.ema)        # the average advances inside the optimiser's launch (eager and
            #                                            captured steps: inr_adam_ema_step_multi / _dev)
        self._graph = None
        self._pipe = None                              # captured two-stream pipeline (use_graph and look_ahead)
        self.global_step = 0
        self.local_step = 0
        self.epoch = 0
        self.stats = {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None}
        self.bg_color = 1
        self.log_ptr = None
        self.ckpt_path = self.best_path = None
        if self.workspace is not None:
            self.ckpt_path = os.path.join(self.workspace, "checkpoints")
            self.best_path = os.path.join(self.ckpt_path, f"{self.name}.pth")
        if self.workspace is not None and use_checkpoint != "scratch":
            if use_checkpoint in ("latest", "latest_model"):
                self.load_checkpoint(model_only=use_checkpoint == "latest_model")
            elif use_checkpoint == "best":
                if self.best_path and os.path.exists(self.best_path):
                    self.load_checkpoint(self.best_path)
                else:
                    self.load_checkpoint()
            else:
                self.load_checkpoint(use_checkpoint)

    def log(self, *args, **kwargs):
        if self.local_rank == 0 and not self.mute:
            print(*args, **kwargs)

    # -- steps -------------------------------------------------------------------------------
    def _lr_step(self):
        if self.lr_scheduler is not None:      # upstream's scheduler object: stepped AFTER the optimiser (see below)
            return
        f = 0.1 ** min(self.global_step / self.iters, 1)
        for g, b in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = b * f

    def _render_kwargs(self):
        return vars(self.opt) if self.opt is not None else {}

    def train_step(self, data):
        """data: rays_o, rays_d [B,N,3] and images [B,N,3|4] (stage 'nerf') or masks int64 [B,N] (stage 'instance').
        RGBA images are blended over a random per-ray background that the render receives too (upstream).
        -> (pred, truth, loss).  Instance stage with ``prune_ignored`` (the default): rays labelled -1 are never
        marched, so their rows of ``pred`` are ZEROS, not rendered logits - loss and gradients are those of the unpruned
        step (cross entropy ignores them either way), but anything that reads ``pred`` at train time (metrics, logging)
        must mask ``truth < 0`` out; ``Trainer(prune_ignored=False)`` renders them."""
        bg_color = self.bg_color
        gt = None
        if self.stage == "nerf":
            images = data["images"]
            gt = images
            if images.shape[-1] == 4:
                bg_color = torch.rand_like(images[..., :3])
                gt = images[..., :3] * images[..., 3:] + bg_color * (1 - images[..., 3:])
        extra = {}
        if self.stage == "instance" and getattr(self.model, "cuda_ray", False) and data["rays_o"].is_cuda:
            extra["ce_labels"] = data["masks"]      # the renderer may form the mask loss inside its compositing launch
            extra["ce_prune"] = self.prune_ignored  # rays labelled -1 carry no loss: never marched
        if (self.stage == "nerf" and self._default_criterion and getattr(self.model, "cuda_ray", False)
                and data["rays_o"].is_cuda):
            extra["mse_target"] = gt                # the renderer may fold blend + loss + gradients into one launch
        ahead = self._ahead
        if ahead is not None:
            self._ahead = None
            # the prefetch belongs to THESE tensors (identity, not addresses: a freshly allocated batch may reuse the
            # address of a freed one - round-3 advisor)
            # ... and, in the instance stage, was pruned with THESE labels (round-5 advisor: a march that left out the
            # rays another mask tensor ignores would silently drop labelled rays of this batch)
            if (ahead["rays_o"] is data["rays_o"] and ahead["rays_d"] is data["rays_d"]
                    and ahead.get("skip_labels") is self._skip_labels(data)):
                extra["marched"] = ahead            # this batch's march was queued under the previous step's backward
            elif hasattr(self.model, "drop_ahead"):
                self.model.drop_ahead(ahead)        # never consumed: its step_counter slot is given back
        outputs = self.model.render(data["rays_o"], data["rays_d"], staged=False, bg_color=bg_color, perturb=True,
                                    force_all_rays=False, **extra, **self._render_kwargs())
        if self.stage == "nerf":
            pred = outputs["image"]
            loss = outputs["image_mse"] if "image_mse" in outputs else self.criterion(pred, gt).mean()
            return pred, gt, loss
        logits = outputs["instance"]
        K = logits.shape[-1]
        if "instance_ce" in outputs:
            loss = outputs["instance_ce"]
        elif logits.is_cuda and K <= 64 and logits.dtype == torch.float32:
            from .. import raymarching
            loss = raymarching.cross_entropy(logits.reshape(-1, K), data["masks"].reshape(-1), ignore_index=-1)
        else:
            loss = torch.nn.functional.cross_entropy(logits.reshape(-1, K), data["masks"].reshape(-1).long(), ignore_index=-1)
        return logits, data["masks"], loss

    @staticmethod
    def _as_image(t, data, channels=None):
        """[B, H*W(, C)] -> [B, H, W(, C)] when the batch carries the image size (upstream's evaluation loaders)."""
        if "H" in data and "W" in data and t.dim() >= 2 and t.shape[1] == int(data["H"]) * int(data["W"]):
            tail = t.shape[2:]
            return t.reshape(t.shape[0], int(data["H"]), int(data["W"]), *tail)
        return t

    def _render_view(self, data, **kwargs):
        """``model.render`` for a whole view.  Upstream's evaluation loaders hand over the H*W rays of an image in
        row-major pixel order; 16 consecutive rays are then a 1x16 strip, and the renderer's 16-ray groups are
        compact in space only for 4x4 patches (field kernel +7 % time on strips, measured).  Rays are independent,
        so they are rendered in patch order and every per-ray output is put back in the caller's order."""
        rays_o, rays_d = data["rays_o"], data["rays_d"]
        H, W = int(data.get("H", 0) or 0), int(data.get("W", 0) or 0)
        if not (getattr(self.model, "cuda_ray", False) and rays_o.is_cuda and rays_o.dim() == 3 and H * W == rays_o.shape[1]
                and H % 4 == 0 and W % 4 == 0):
            return self.model.render(rays_o, rays_d, **kwargs)
        inds = patch_order(H, W, 4, rays_o.device)
        out = self.model.render(rays_o[:, inds].contiguous(), rays_d[:, inds].contiguous(), **kwargs)
        for k, v in list(out.items()):
            if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == H * W:
                back = torch.empty_like(v)
                back[:, inds] = v
                out[k] = back
        return out

    @torch.no_grad()
    def eval_step(self, data):
        """-> (prediction, depth, truth, loss); images may come as [B,N,C] or [B,H,W,C] (upstream), C = 3 or 4."""
        outputs = self._render_view(data, staged=True, bg_color=self.bg_color, perturb=False, **self._render_kwargs())
        return self._eval_outputs(data, outputs)

    def _eval_outputs(self, data, outputs):
        if self.stage == "nerf":
            images = data["images"]
            flat = images.reshape(images.shape[0], -1, images.shape[-1])
            gt = flat[..., :3] * flat[..., 3:] + self.bg_color * (1 - flat[..., 3:]) if flat.shape[-1] == 4 else flat
            loss = self.criterion(outputs["image"], gt).mean()
            return (self._as_image(outputs["image"], data), self._as_image(outputs["depth"], data),
                    self._as_image(gt, data), loss)
        logits = outputs["instance"]
        K = logits.shape[-1]
        masks = data["masks"].reshape(logits.shape[0], -1)
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, K), masks.reshape(-1).long(), ignore_index=-1)
        return (self._as_image(logits.argmax(-1), data), self._as_image(outputs["depth"], data),
                self._as_image(masks, data), loss)

    @torch.no_grad()
    def test_step(self, data, bg_color=None, perturb=False):
        outputs = self._render_view(data, staged=True, bg_color=self.bg_color if bg_color is None else bg_color,
                                    perturb=perturb, **self._render_kwargs())
        return self._test_outputs(data, outputs)

    def _test_outputs(self, data, outputs):
        inst = outputs.get("instance")
        return (self._as_image(outputs["image"], data), self._as_image(outputs["depth"], data),
                None if inst is None else self._as_image(inst, data))

    pipeline_views = True      # evaluate / test render view sequences through FramePipeline (two alternating streams)

    @torch.no_grad()
    def render_sequence(self, loader, pipeline=None, **kwargs):
        """Yields ``(data, outputs)`` for every view of ``loader`` - what ``evaluate_one_epoch`` and ``test`` (and
        bench.py's headline loop) iterate over.  With ``cuda_ray`` on a GPU the views alternate on the two streams of a
        ``FramePipeline``: the ray/box test and march of view i+1 and the compositing of view i-1 run under the field
        kernel of view i (the field kernels themselves stay one after the other).  A view's outputs are yielded once the
        NEXT view has been queued and its own stream has finished; frames are bit-identical to ``eval_step`` /
        ``test_step`` on each view by itself (rays are independent; tests/test_gpu_parity.py).  Upstream renders one view
        at a time on the default stream; ``pipeline=False`` (or ``Trainer.pipeline_views = False``) does that."""
        kw = dict(staged=True, bg_color=self.bg_color, perturb=False)
        kw.update(self._render_kwargs())
        kw.update(kwargs)
        use = self.pipeline_views if pipeline is None else bool(pipeline)
        use = use and getattr(self.model, "cuda_ray", False) and self.device.type == "cuda"
        if not use:
            for data in loader:
                yield data, self._render_view(data, **kw)
            return
        from .renderer import FramePipeline
        # ONE pipeline (two streams) per trainer, reused by every sequence: HIP maps streams onto a handful of hardware
        # queues, and a fresh pair per call landed on ONE queue - no overlap at all, 5.9 ms per frame instead of 5.0
        # (kernel trace, profiles/r04_NOTES.txt 3)
        pipe = self.__dict__.get("_frame_pipe")
        if pipe is None or pipe.net is not self.model:
            pipe = self.__dict__["_frame_pipe"] = FramePipeline(self.model, self.device)
        pipe.open()
        try:
            pending = None
            for data in loader:
                cur = torch.cuda.current_stream()
                st = pipe.next_stream()
                st.wait_stream(cur)
                for k in ("rays_o", "rays_d"):
                    if torch.is_tensor(data.get(k)) and data[k].is_cuda:
                        data[k].record_stream(st)
                with torch.cuda.stream(st):
                    out = self._render_view(data, field_gate=pipe, **kw)
                    done = torch.cuda.Event()
                    done.record(st)
                if pending is not None:
                    yield self._finish_view(*pending)
                pending = (data, out, done)
            if pending is not None:
                yield self._finish_view(*pending)
        finally:
            pipe.close()

    @staticmethod
    def _finish_view(data, out, done):
        done.synchronize()
        cur = torch.cuda.current_stream()
        for v in out.values():                 # allocated on the pipeline's stream, consumed on the caller's
            if torch.is_tensor(v) and v.is_cuda:
                v.record_stream(cur)
        return data, out

    # -- loops -------------------------------------------------------------------------------
    # -- captured step ----------------------------------------------------------------------
    GRAPH_ALIGN = 16384        # buffer sizes are rounded up to this many samples so that a graph survives the
                               # small changes of mean_count at every occupancy update

    def _graph_capacity(self, mean_count):
        """Sample-buffer size of the captured step for a measured ``mean_count``: rounded up to GRAPH_ALIGN, and the size
        already captured for is kept while it still holds the batch and is at most max(2 units, 1/8) too large - the
        16-step mean of a converged scene wanders by a few thousand samples, and every change of the size is a new set
        of graphs (tools/pipeline_convergence_probe.py: without the hold, 84 captures in 31 updates)."""
        a = self.GRAPH_ALIGN
        need = (mean_count + a - 1) // a * a
        held = getattr(self, "_held_capacity", 0)
        if need <= held <= need + max(2 * a, need // 8):
            need = held
        self._held_capacity = need
        return need

    def _graph_key(self, data):
        return (self.model.mean_count, self.stage) + tuple((k, tuple(v.shape)) for k, v in sorted(data.items())
                                                            if torch.is_tensor(v))

    def _capture(self, data):
        m = self.model
        static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()}
        slot = m.local_step % 16
        opt = self.optimizer
        for g in opt.param_groups:              # moments and the hyper-parameter tensor must exist BEFORE the capture
            for p in g["params"]:               # (an allocation + zero fill inside it would be replayed every step)
                if p.requires_grad:
                    opt._moments(p)
                    if getattr(p, "_is_hash_table", False):
                        from . import network as _network
                        _network.fx_state(p)        # the fixed-point gradient state of a table: not inside a capture either
                        if _network.fx_bits() == 64:
                            _network.fx_acc64(p)
        opt.hyper_tensor(self.device)
        self.optimizer.zero_grad()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _, _, loss = self.train_step(static)
            loss.backward()
            self.optimizer.step_captured()
        m.local_step -= 1                       # the capture pass launched nothing; undo its bookkeeping
        self._graph = {"graph": g, "static": static, "loss": loss, "slot": slot, "key": self._graph_key(data)}

    def _replay(self, data):
        G, m = self._graph, self.model
        copy_tensors([(G["static"][k], v) for k, v in data.items() if torch.is_tensor(v)])
        self._lr_step()
        self.optimizer.refresh_hyper()
        G["graph"].replay()
        cur = m.local_step % 16
        if cur != G["slot"]:                    # the graph always writes the slot it was captured with
            m.step_counter[cur].copy_(m.step_counter[G["slot"]])
        m.local_step += 1
        self.optimizer.bump_versions()
        if self.ema is not None:
            self.ema.update()
        return G["loss"].detach()

    # -- captured two-stream pipeline (use_graph and look_ahead) -------------------------------------------------
    # The step is bound by the table-gradient scatter (memory-side atomic unit; the CUs idle for most of it) and by the
    # host (0.6-0.85 ms to enqueue 0.83 ms of kernels; a second queue costs the runtime ~0.3 ms more per step,
    # profiles/r03_NOTES.txt 16).  Both at once: ONE hipGraph per step holds the step's own launches on the capture
    # stream and, forked off right before the scatter, the parameter-independent head of the NEXT batch on a second
    # stream - ray/box test, march and, in the instance stage, the frozen NeRF's forward and the weight compositing
    # (``NeRFRenderer.march_ahead(shade=True)``) - joined again before the graph ends.  Two persistent buffer sets
    # alternate (step i reads set A and fills B, step i+1 reads B and fills A), so there is one graph per (set, does the
    # step compute its own head, does it look ahead): the first step after an occupancy update computes its own head
    # (the grid changed), the last one before an update does not look ahead.
    def _pipe_applies(self, data):
        m = self.model
        if not (self.use_graph and self.look_ahead and m.cuda_ray and m.mean_count > 0 and data["rays_o"].is_cuda
                and hasattr(m, "march_ahead")):
            return False
        # the persistent buffer sets need the staged wave-per-ray marcher (batches up to 8192 rays, max_steps <= 1024,
        # march mode not forced to lane-per-ray): other batches take the one-stream captured step
        from .. import raymarching
        kw = self._render_kwargs()
        return bool(_lib.load().inr_march_write_fills_unowned_rows(data["rays_o"].numel() // 3, raymarching.SAMPLE_CAP_TRAIN,
                                                                    int(kw.get("max_steps", 1024))))

    def _pipe_args(self):
        kw = self._render_kwargs()
        return dict(dt_gamma=kw.get("dt_gamma", 0), perturb=True, max_steps=kw.get("max_steps", 1024),
                    shade=self.stage == "instance" and self.shade_ahead, T_thresh=kw.get("T_thresh", 1e-4))

    def _skip_labels(self, data):
        """The labels whose ignored rays the march leaves out (instance stage, ``prune_ignored``), or None."""
        return data["masks"] if (self.stage == "instance" and self.prune_ignored and "masks" in data) else None

    def _pipe_init(self, data):
        from .. import raymarching
        m, dev = self.model, data["rays_o"].device
        N = data["rays_o"].numel() // 3
        M_al = (int(m.mean_count) + 127) // 128 * 128
        shade = self.stage == "instance" and self.shade_ahead and m.shade_ahead_applies()
        sets = []
        for _ in range(2):
            sets.append({"bufs": raymarching.march_train_buffers(N, M_al, dev, shade=shade),
                         "static": {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data.items()},
                         "counter": torch.zeros(2, dtype=torch.int32, device=dev), "marched": None})
        opt = self.optimizer
        for g in opt.param_groups:              # moments and the hyper-parameter tensor must exist BEFORE a capture
            for p in g["params"]:
                if p.requires_grad:
                    opt._moments(p)
                    if getattr(p, "_is_hash_table", False):
                        from . import network as _network
                        _network.fx_state(p)        # the fixed-point gradient state of a table: not inside a capture either
                        if _network.fx_bits() == 64:
                            _network.fx_acc64(p)
        opt.hyper_tensor(dev)
        self._pipe = {"key": self._graph_key(data), "sets": sets, "graphs": {}, "turn": 0, "primed": False,
                      "expect": None, "side": torch.cuda.Stream(device=dev)}

    def _pipe_capture(self, turn, prime, ahead):
        from . import network as _network
        from .. import raymarching
        P, m = self._pipe, self.model
        S, Nx = P["sets"][turn], P["sets"][turn ^ 1]
        args = self._pipe_args()
        if not prime and S["marched"] is None:
            raise RuntimeError("pipeline: a step that consumes a prefetched head was asked for before any prefetch")
        self.optimizer.zero_grad()
        torch.cuda.synchronize()
        step0 = m.local_step
        side = P["side"]
        g = torch.cuda.CUDAGraph()
        try:
            self._pipe_capture_body(g, S, Nx, side, prime, ahead, args)
        finally:
            _network._before_scatter.pop("hook", None)       # a failed capture must not leave its fork for an eager backward
            self._ahead = None
            m.local_step = step0                # the capture pass launched nothing; undo its bookkeeping
        G = {"graph": g, "loss": self._pipe_loss}
        P["graphs"][(turn, prime, ahead)] = G
        return G

    def _pipe_capture_body(self, g, S, Nx, side, prime, ahead, args):
        from . import network as _network
        from .. import raymarching
        m = self.model
        with torch.cuda.graph(g):
            if prime:
                S["marched"] = m.march_ahead(S["static"]["rays_o"], S["static"]["rays_d"], stream=None, bufs=S["bufs"],
                                             counter=S["counter"], skip_labels=self._skip_labels(S["static"]), **args)
                if S["marched"] is None:
                    raise RuntimeError("pipeline: march_ahead refused the batch (not the steady state / staged marcher)")
            marched = dict(S["marched"])
            marched["consume"] = lambda: None        # same graph or an earlier replay on the same stream: ordered already
            # (the dict may come from an earlier capture; whether the buffers' CONTENT is current is _pipe_step's business
            # at every replay, so the renderer's own staleness check must not refuse it while the graph is captured)
            marched["grid_state"] = getattr(m, "iter_density", 0)
            self._ahead = marched

            def hook():
                Nx["marched"] = m.march_ahead(Nx["static"]["rays_o"], Nx["static"]["rays_d"], stream=side, bufs=Nx["bufs"],
                                              counter=Nx["counter"], skip_labels=self._skip_labels(Nx["static"]), **args)
            if ahead and self.pipe_fork == "start":
                hook()                               # fork right away: beside the whole step
            elif ahead:
                _network._before_scatter["hook"] = hook      # fork from inside the backward, right before the scatter
            _, _, loss = self.train_step(S["static"])
            if self._ahead is not None:
                raise RuntimeError("pipeline: the render did not take the prefetched head")
            loss.backward(gradient=raymarching.unit_gradient(loss.device))
            if ahead and _network._before_scatter.pop("hook", None) is not None:
                hook()                               # no fused table backward ran (composable path): fork here
            self.optimizer.step_captured()
            if ahead:
                torch.cuda.current_stream().wait_stream(side)        # join: the graph ends with both branches done
        self._pipe_loss = loss

    def _pipe_step(self, data, next_data):
        m = self.model
        key = self._graph_key(data)
        if self._pipe is None or self._pipe["key"] != key:
            self._pipe_init(data)
        P = self._pipe
        t = P["turn"]
        S, Nx = P["sets"][t], P["sets"][t ^ 1]
        # the prefetched head is this batch's only if it was marched for these tensors through the occupancy grid as it is
        # NOW (an update between the two steps - the trainer's own never falls there, a caller's may - bumps iter_density)
        prime = not (P["primed"] and P["expect"] is data["rays_o"] and P.get("grid_state") == getattr(m, "iter_density", 0))
        pairs = []
        if prime:
            pairs += [(S["static"][k], v) for k, v in data.items() if torch.is_tensor(v)]
        # no look-ahead across an occupancy update (global_step is already advanced: the next call starts with one
        # when it is a multiple of the interval), nor into a batch of another shape
        ahead = (next_data is not None and self.global_step % self.update_extra_interval != 0
                 and all(torch.is_tensor(next_data.get(k)) and next_data[k].shape == v.shape and next_data[k].dtype == v.dtype
                         for k, v in Nx["static"].items() if torch.is_tensor(v)))
        if ahead:
            pairs += [(Nx["static"][k], v) for k, v in next_data.items() if torch.is_tensor(v)]
        if not prime:
            # the march of THIS batch ran in the previous replay, into the set's own counter: it goes to the renderer's
            # slot together with the inputs (one launch); a step that marches itself copies after its replay
            pairs.append((m.step_counter[m.local_step % 16], S["counter"]))
        G = P["graphs"].get((t, prime, ahead)) or self._pipe_capture(t, prime, ahead)
        copy_tensors(pairs)
        self._lr_step()
        self.optimizer.refresh_hyper()
        G["graph"].replay()
        if prime:
            m.step_counter[m.local_step % 16].copy_(S["counter"])
        m.local_step += 1
        m.last_counter = S["counter"]
        self.optimizer.bump_versions()
        if self.ema is not None:
            self.ema.update()
        P["primed"], P["expect"] = ahead, (next_data["rays_o"] if ahead else None)
        P["grid_state"] = getattr(m, "iter_density", 0)
        P["turn"] = t ^ 1
        return G["loss"].detach()

    def _march_ahead(self, next_data):
        """Queues the march of the NEXT batch on a side stream (``NeRFRenderer.march_ahead``): it does not depend on the
        parameters, and right now the device is about to run this step's backward, whose scatter leaves the CUs idle.
        Not before an occupancy update (the grid would change under the prefetched samples), not for captured steps."""
        m = self.model
        if (next_data is None or self.use_graph or not getattr(m, "cuda_ray", False) or not next_data["rays_o"].is_cuda
                or m.mean_count <= 0 or self.global_step % self.update_extra_interval == 0
                or not hasattr(m, "march_ahead")):
            return
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=next_data["rays_o"].device)
        kw = self._render_kwargs()
        ro, rd = next_data["rays_o"], next_data["rays_d"]
        if not (ro.is_contiguous() and rd.is_contiguous() and ro.dtype == torch.float32 and rd.dtype == torch.float32):
            return                                   # the buffers marched from must be the ones render() will see
        self._ahead = m.march_ahead(ro, rd, dt_gamma=kw.get("dt_gamma", 0), perturb=True,
                                    max_steps=kw.get("max_steps", 1024), stream=self._side_stream,
                                    shade=self.stage == "instance" and self.shade_ahead, T_thresh=kw.get("T_thresh", 1e-4),
                                    skip_labels=self._skip_labels(next_data))

    def train_one_step(self, data, next_data=None):
        """One optimisation step on ``data``.  ``next_data`` (optional): the batch the NEXT call will get - its ray/box
        test and march are then queued on a side stream under this step's backward (``train_one_epoch`` looks one batch
        ahead; same results, ~45 us per step less)."""
        self.model.train()
        if self.model.cuda_ray and self.global_step % self.update_extra_interval == 0:
            self.model.update_extra_state()
            if self.use_graph and self.model.mean_count > 0:
                self.model.mean_count = self._graph_capacity(self.model.mean_count)
        self.global_step += 1
        if self._pipe_applies(data):
            return self._pipe_step(data, next_data)
        if self.use_graph and self.model.cuda_ray and self.model.mean_count > 0:
            if self._graph is None or self._graph["key"] != self._graph_key(data):
                self._capture(data)
            return self._replay(data)
        self.optimizer.zero_grad()
        _, _, loss = self.train_step(data)
        from . import network as _network
        if next_data is not None:
            # queued from inside the backward, right before the table-gradient scatter (nerf/network.py::_table_backward)
            _network._before_scatter["hook"] = lambda: self._march_ahead(next_data)
        if loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32:
            from .. import raymarching
            loss.backward(gradient=raymarching.unit_gradient(loss.device))     # no ones_like fill launch
        else:
            loss.backward()
        if _network._before_scatter.pop("hook", None) is not None:
            self._march_ahead(next_data)            # no fused table backward ran (composable path): queue it now
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        fused = isinstance(self.optimizer, FusedAdam)
        # FusedAdam takes the 1 / world_size of the gradient average as a factor inside its sweep
        scale = allreduce_gradients(params, self.world_size, average=not fused, sharded=grad_sync.sharded_ok)
        self._lr_step()
        if fused:
            if self.lr_scheduler is None:
                self.optimizer.step_impl(scale)
            else:                                   # torch's schedulers count calls of the wrapped optimizer.step
                self.optimizer.step(grad_scale=scale)
            if self.world_size > 1:
                grad_sync.allgather_params(params)      # reduce-scatter schedule: the rows each rank updated -> all
        else:
            self.optimizer.step()
        if self.lr_scheduler is not None and self.scheduler_update_every_step:
            self.lr_scheduler.step()
        if self.ema is not None:
            self.ema.update()
        return loss.detach()

    def _sync_shards(self):
        """Reduce-scatter gradient schedule: Adam moments and EMA rows are kept current on their owner rank only; gather
        them before anything reads them whole (checkpoint, evaluation on the averaged parameters).  A collective - every
        rank calls it at the same point; a no-op for the default all-reduce schedule."""
        if self.world_size > 1 and dist.is_available() and dist.is_initialized():
            if isinstance(self.optimizer, FusedAdam):
                self.optimizer.sync_shards()
            if self.ema is not None:
                self.ema.sync_shards()

    def train(self, train_loader, valid_loader=None, max_epochs=1):
        """Upstream's epoch loop: cells no training camera sees are marked once (loaders that expose ``_data.poses`` /
        ``_data.intrinsics``), every epoch is checkpointed (``max_keep_ckpt`` rotating files), every ``eval_interval``
        epochs the validation loader is evaluated and the best result kept."""
        if self.model.cuda_ray and self.epoch == 0:
            d = getattr(train_loader, "_data", None)
            if d is not None and hasattr(d, "poses") and hasattr(d, "intrinsics"):
                self.model.mark_untrained_grid(d.poses, d.intrinsics)
        for _ in range(self.epoch + 1, self.epoch + max_epochs + 1):
            self.epoch += 1
            self.train_one_epoch(train_loader)
            if self.workspace is not None:
                self._sync_shards()                     # all ranks: rank 0 then writes complete moments / averages
            if self.workspace is not None and self.local_rank == 0:
                self.save_checkpoint(full=True, best=False)
            if valid_loader is not None and self.epoch % max(self.eval_interval, 1) == 0:
                self.evaluate_one_epoch(valid_loader)
                if self.workspace is not None and self.local_rank == 0:
                    self.save_checkpoint(full=False, best=True)

    def train_one_epoch(self, loader):
        total, n = 0.0, 0
        for m in self.metrics:
            m.clear()
        it = iter(loader)
        data = next(it, None)
        # The steps' losses are kept ON THE DEVICE and read once per epoch: upstream's `loss.item()` per step is a host
        # synchronisation per step, which leaves the device idle while the host queues the next step (round 6: 0.75-1.3 ms
        # of kernels per step against 0.4-0.5 ms of enqueue time).  One slot per step, so that the epoch's mean can leave
        # out the steps whose loss is not a number - a batch without a single labelled ray (an image in which match_seg
        # matched nothing: every label -1) has the cross entropy of an empty set, NaN as torch's, and touches no parameter.
        slots = None
        try:
            slots = torch.empty(max(len(loader), 1), dtype=torch.float32, device=self.device)
        except TypeError:
            pass                                     # a loader without a length: summed on the device instead
        acc, n_acc, kept_host = None, 0, []
        while data is not None:
            nxt = next(it, None)                     # one batch of look-ahead: its march may run under this step's backward
            loss = self.train_one_step(data, nxt if self.look_ahead else None)
            if torch.is_tensor(loss) and loss.is_cuda:
                if slots is not None and n < slots.numel():
                    slots[n].copy_(loss.detach().reshape(()))
                else:
                    acc = loss.detach().float().clone() if acc is None else acc.add_(loss.detach())
                    n_acc += 1
            else:
                kept_host.append(float(loss))
            n += 1
            data = nxt
        vals = list(kept_host)
        if slots is not None and n:
            vals += slots[:min(n, slots.numel())].tolist()
        finite = [v for v in vals if v == v and abs(v) != float("inf")]
        self.stats.setdefault("nan_steps", []).append(len(vals) - len(finite))
        total, n = sum(finite), len(finite)
        if acc is not None:                          # the length-less loader's sum (not NaN-aware: one value for all its steps)
            total, n = total + float(acc), n + n_acc
        if self.lr_scheduler is not None and not self.scheduler_update_every_step:
            self.lr_scheduler.step()
        self.stats["loss"].append(total / max(n, 1))
        self.log(f"==> epoch {self.epoch}: loss {self.stats['loss'][-1]:.6f}, lr {self.optimizer.param_groups[0]['lr']:.6f}")

    def evaluate(self, loader, name=None):
        return self.evaluate_one_epoch(loader, name)

    @torch.no_grad()
    def evaluate_one_epoch(self, loader, name=None):
        """Runs the metrics (the caller's, else PSNR for the NeRF stage / mIoU for the instance stage) on the averaged
        parameters when an EMA is kept; appends to ``stats['results']`` (first metric, or the loss when
        ``use_loss_as_metric`` and no metric is given) and ``stats['valid_loss']``."""
        self.model.eval()
        meters = self.metrics or [PSNRMeter() if self.stage == "nerf" else MIoUMeter(self.model.num_instances)]
        for m in meters:
            m.clear()
        self._sync_shards()
        if self.ema is not None:
            self.ema.store()
            self.ema.copy_to()
        total, n = 0.0, 0
        # a subclass (or a caller) that replaced eval_step - upstream's extension point - gets its method called, one
        # view at a time; the stock step goes through the pipelined view loop
        custom = "eval_step" in self.__dict__ or type(self).eval_step is not Trainer.eval_step
        steps = ((d, None) for d in loader) if custom else self.render_sequence(loader)
        for data, outputs in steps:
            pred, _, truth, loss = self.eval_step(data) if custom else self._eval_outputs(data, outputs)
            for m in meters:
                m.update(pred, truth)
            total += float(loss)
            n += 1
        if self.ema is not None:
            self.ema.restore()
        self.model.train()
        # upstream: the validation loss is the result unless use_loss_as_metric=False (then the first metric).  Without
        # caller-supplied metrics the stage's own meter (PSNR / mIoU, higher is better) is the result.
        as_loss = bool(self.metrics) and self.use_loss_as_metric
        self._result_mode = self.best_mode if self.metrics else "max"
        result = total / max(n, 1) if as_loss else meters[0].measure()
        if self.world_size > 1:
            t = torch.tensor([result], dtype=torch.float64, device=self.device)
            dist.all_reduce(t)
            result = float(t.item()) / self.world_size
        self.stats["valid_loss"].append(total / max(n, 1))
        self.stats["results"].append(result)
        for m in meters:
            self.log(m.report())
        return result

    @torch.no_grad()
    def test(self, loader, save_path=None, name=None, write_video=False):
        """Renders every view of ``loader`` and writes ``<name>_<i>_rgb.png`` / ``_depth.png`` (instance stage: also
        ``_instance.png`` with the arg-max ids) into ``save_path`` (default ``<workspace>/results``).  Upstream also
        assembles a video with imageio, which this image does not have: ``write_video`` is accepted and ignored."""
        from PIL import Image
        save_path = save_path or os.path.join(self.workspace or ".", "results")
        name = name or f"{self.name}_ep{self.epoch:04d}"
        os.makedirs(save_path, exist_ok=True)
        self.model.eval()
        written = []
        custom = "test_step" in self.__dict__ or type(self).test_step is not Trainer.test_step
        steps = ((d, None) for d in loader) if custom else self.render_sequence(loader)
        for i, (data, outputs) in enumerate(steps):
            rgb, depth, inst = self.test_step(data) if custom else self._test_outputs(data, outputs)
            if rgb.dim() != 4:
                raise ValueError("test() needs batches that carry the image size ('H', 'W')")
            img = (rgb[0].clamp(0, 1) * 255).byte().cpu().numpy()
            dep = (depth[0].clamp(0, 1) * 255).byte().cpu().numpy()
            Image.fromarray(img).save(os.path.join(save_path, f"{name}_{i:04d}_rgb.png"))
            Image.fromarray(dep).save(os.path.join(save_path, f"{name}_{i:04d}_depth.png"))
            written.append(os.path.join(save_path, f"{name}_{i:04d}_rgb.png"))
            if inst is not None:
                ids = inst[0].argmax(-1).clamp(0, 255).byte().cpu().numpy()
                Image.fromarray(ids).save(os.path.join(save_path, f"{name}_{i:04d}_instance.png"))
        self.model.train()
        return written

    # -- checkpoint (upstream keys: epoch, global_step, stats, model, optimizer, lr_scheduler, ema, mean_count, mean_density)
    def save_checkpoint(self, name=None, full=False, best=False, remove_old=True, path=None):
        """``full``: with optimizer / scheduler / EMA state (to resume training); ``best``: written to
        ``<workspace>/checkpoints/<name>.pth`` only when the last result improves on ``stats['best_result']``
        (``best_mode``); otherwise a rotating ``<name>_ep####.pth`` (``max_keep_ckpt`` files kept)."""
        state = {"epoch": self.epoch, "global_step": self.global_step, "stats": self.stats}
        if self.model.cuda_ray:
            state["mean_count"] = self.model.mean_count
            state["mean_density"] = self.model.mean_density
        if full or path is not None:
            state["optimizer"] = self.optimizer.state_dict()
            if self.lr_scheduler is not None:
                state["lr_scheduler"] = self.lr_scheduler.state_dict()
            if self.ema is not None:
                state["ema"] = self.ema.state_dict()
            # fixed-point gradient scales of the trained tables (nerf/network.py::fx_state): a resumed run then rounds the
            # next step's row sums to the same quanta the uninterrupted run does (an extra key; upstream's loader ignores it)
            fx = {n: p._fx_state[:96].detach().cpu().clone() for n, p in self.model.named_parameters()
                  if getattr(p, "_fx_state", None) is not None and getattr(p, "_fx_primed", 0)}
            if fx:
                from . import network as _network
                state["fx_state"], state["fx_bits"] = fx, _network.fx_bits()
        if path is None and best:
            results = self.stats["results"]
            if not results:
                return None
            cur, prev = results[-1], self.stats["best_result"]
            mode = getattr(self, "_result_mode", self.best_mode)
            better = prev is None or (cur < prev if mode == "min" else cur > prev)
            if not better:
                return None
            self.stats["best_result"] = cur
            if self.ema is not None:                       # the best checkpoint holds the averaged parameters
                self.ema.store()
                self.ema.copy_to()
            state["model"] = self.model.state_dict()
            if self.ema is not None:
                self.ema.restore()
            path = self.best_path
        else:
            state["model"] = self.model.state_dict()
            if path is None:
                name = name or f"{self.name}_ep{self.epoch:04d}"
                path = os.path.join(self.ckpt_path or os.path.join(self.workspace or ".", "checkpoints"), f"{name}.pth")
                if remove_old:
                    self.stats["checkpoints"].append(path)
                    while len(self.stats["checkpoints"]) > self.max_keep_ckpt:
                        old = self.stats["checkpoints"].pop(0)
                        if os.path.exists(old) and self.local_rank == 0:
                            os.remove(old)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        if self.local_rank == 0:
            torch.save(state, path)
        return path

    def load_checkpoint(self, checkpoint=None, model_only=False):
        """``checkpoint`` None: the newest ``<name>_ep*.pth`` of the workspace (nothing happens when there is none).
        Upstream semantics: a file without a 'model' key is a bare model state dict; missing / unexpected keys are
        reported, not fatal; a failing optimizer / scheduler / EMA restore only warns (e.g. an Adam state saved for a
        different set of trained parameters)."""
        self._pipe = self._graph = None          # captured steps hold buffers sized and primed for the old state
        import glob
        import warnings
        if checkpoint is None:
            found = sorted(glob.glob(os.path.join(self.ckpt_path or "", f"{self.name}_ep*.pth")))
            if not found:
                self.log("[INFO] no checkpoint found, model randomly initialized")
                return
            checkpoint = found[-1]
        state = torch.load(checkpoint, map_location=self.device, weights_only=False)
        if "model" not in state:
            self.model.load_state_dict(state, strict=False)
            return
        missing, unexpected = self.model.load_state_dict(state["model"], strict=False)
        if missing or unexpected:
            warnings.warn(f"load_checkpoint: missing keys {list(missing)}, unexpected keys {list(unexpected)}")
        if self.model.cuda_ray:
            self.model.mean_count = state.get("mean_count", 0)
            self.model.mean_density = state.get("mean_density", 0)
        if model_only:
            return
        self.epoch = state.get("epoch", self.epoch)
        self.global_step = state.get("global_step", self.global_step)
        self.stats = {**self.stats, **state.get("stats", {})}
        if "fx_state" in state:
            from . import network as _network
            named = dict(self.model.named_parameters())
            for n, saved in state["fx_state"].items():
                p = named.get(n)
                if p is not None and p.is_cuda:
                    st = _network.fx_state(p)
                    st.zero_()
                    st[:saved.numel()].copy_(saved.to(st.device))
                    p._fx_primed = int(state.get("fx_bits", 32))
        for key, obj in (("optimizer", self.optimizer), ("lr_scheduler", self.lr_scheduler), ("ema", self.ema)):
            if obj is not None and key in state:
                try:
                    obj.load_state_dict(state[key])
                except Exception as e:                                # noqa: BLE001 - upstream warns and goes on
                    warnings.warn(f"load_checkpoint: {key} state not restored ({type(e).__name__}: {e})")
