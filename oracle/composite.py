"""Alpha compositing along rays (oracle; test infrastructure only).

Follows SURVEY.md section 8a rows a12/a13/a5 and Appendix A.1 "Composite"
(upstream ``raymarching.composite_rays_train`` forward/backward and
``composite_rays`` of the un-vendored submodule pinned at
/root/reference/README.md:27,59; the K-channel accumulation is the fork's
addition).  Parity unpinned.

Per ray, in sample order, starting from T = 1, t = 0:
    alpha = 1 - exp(-sigma * deltas[:,0]);   w = alpha * T
    weights_sum += w;  t += deltas[:,1];  depth += w*t;  image += w*rgb;
    extra += w*feat (K channels, optional);  T *= (1 - alpha);  stop if T < T_thresh
"""
import numpy as np
import torch

F32 = np.float32


def _padded(rays, device=None):
    rays = torch.as_tensor(np.asarray(rays), dtype=torch.int64)
    off, cnt = rays[:, 1], rays[:, 2]
    Lmax = int(cnt.max().item()) if rays.shape[0] else 0
    Lmax = max(Lmax, 1)
    ar = torch.arange(Lmax)[None, :]
    valid = ar < cnt[:, None]
    idx = torch.where(valid, off[:, None] + ar, torch.zeros_like(ar))
    return idx, valid


def composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh=1e-4, extra=None,
                         detach_weights_for_extra=True):
    """Differentiable (torch) forward.

    sigmas f32[M], rgbs f32[M,3], deltas f32[M,2], rays i32[N,3]=(id,offset,count),
    extra f32[M,K] or None.  Returns dict(weights_sum[N], depth[N], image[N,3],
    extra[N,K] or None, weights f32[M] (per-sample w, zero where unused)).
    Outputs are indexed by ray id rays[:,0].  ``extra`` is composited with the
    weights detached (the instance field is trained against a frozen NeRF,
    SURVEY a13).
    """
    sigmas = torch.as_tensor(sigmas, dtype=torch.float32)
    rgbs = torch.as_tensor(rgbs, dtype=torch.float32)
    deltas = torch.as_tensor(deltas, dtype=torch.float32)
    N = len(rays)
    rid = torch.as_tensor(np.asarray(rays)[:, 0], dtype=torch.int64)
    idx, valid = _padded(rays)
    sg = sigmas[idx]
    d0 = deltas[idx, 0]
    d1 = torch.where(valid, deltas[idx, 1], torch.zeros(()))
    alpha = torch.where(valid, 1 - torch.exp(-sg * d0), torch.zeros(()))
    one_m = 1 - alpha
    Tincl = torch.cumprod(one_m, dim=1)
    T = torch.cat([torch.ones((N, 1)), Tincl[:, :-1]], dim=1)      # T before sample i
    with torch.no_grad():
        used = valid & (T >= T_thresh)
    w = torch.where(used, alpha * T, torch.zeros(()))
    t = torch.cumsum(d1, dim=1)
    ws = w.sum(1)
    depth = (w.detach() * t).sum(1)                                 # depth gets no gradient
    image = (w[..., None] * rgbs[idx]).sum(1)
    out_ws = torch.zeros(N).index_add(0, rid, ws)
    out_depth = torch.zeros(N).index_add(0, rid, depth)
    out_img = torch.zeros(N, 3).index_add(0, rid, image)
    out = dict(weights_sum=out_ws, depth=out_depth, image=out_img, extra=None)
    if extra is not None:
        extra = torch.as_tensor(extra, dtype=torch.float32)
        we = w.detach() if detach_weights_for_extra else w
        e = (we[..., None] * extra[idx]).sum(1)
        out["extra"] = torch.zeros(N, extra.shape[1]).index_add(0, rid, e)
    wflat = torch.zeros(sigmas.shape[0])
    wflat[idx[used]] = w.detach()[used]
    out["weights"] = wflat
    return out


def composite_backward_analytic(grad_ws, grad_image, sigmas, rgbs, deltas, rays,
                                weights_sum, image, T_thresh=1e-4):
    """The closed form the HIP backward kernel implements (numpy, sequential).

    d/d rgb_i   = g_img * w_i
    d/d sigma_i = deltas[i,0] * ( g_img . (T_{i+1}*rgb_i - (C_final - C_upto_i))
                                  + g_ws * (1 - ws_final) )
    """
    sigmas = np.asarray(sigmas, F32)
    rgbs = np.asarray(rgbs, F32)
    deltas = np.asarray(deltas, F32)
    g_s = np.zeros_like(sigmas)
    g_c = np.zeros_like(rgbs)
    for rid, off, cnt in np.asarray(rays):
        T = F32(1.0)
        acc = np.zeros(3, F32)
        for i in range(off, off + cnt):
            alpha = F32(1.0) - np.exp(-sigmas[i] * deltas[i, 0])
            w = alpha * T
            acc = acc + w * rgbs[i]
            T = T * (F32(1.0) - alpha)
            g_c[i] = grad_image[rid] * w
            g_s[i] = deltas[i, 0] * (np.dot(grad_image[rid], T * rgbs[i] - (image[rid] - acc))
                                     + grad_ws[rid] * (F32(1.0) - weights_sum[rid]))
            if T < T_thresh:
                break
    return g_s, g_c


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, deltas,
                   weights_sum, depth, image, T_thresh=1e-4, extra_in=None, extra_acc=None):
    """Inference accumulate-in-place step (numpy).  Mutates rays_alive, rays_t,
    weights_sum, depth, image (and extra_acc f32[N,K] when given).

    For live ray slot n (ray index r = rays_alive[n]) and its n_step rows:
        stop at the first row with deltas[.,0] == 0 (ray ended in march_rays);
        alpha = 1 - exp(-sigma*delta0); T = 1 - weights_sum[r]; w = alpha*T;
        weights_sum += w; t += delta1; depth += w*t; image += w*rgb;
        stop if T*(1-alpha) < T_thresh.
    A ray that stopped early is marked dead (rays_alive[n] = -1); otherwise
    rays_t[r] = t.
    """
    for n in range(n_alive):
        r = int(rays_alive[n])
        if r < 0:
            continue
        t = F32(rays_t[r])
        ws = F32(weights_sum[r])
        d = F32(depth[r])
        c = image[r].astype(F32).copy()
        e = None if extra_acc is None else extra_acc[r].astype(F32).copy()
        step = 0
        while step < n_step:
            i = n * n_step + step
            if deltas[i, 0] == 0:
                break
            alpha = F32(1.0) - np.exp(-F32(sigmas[i]) * F32(deltas[i, 0]))
            T = F32(1.0) - ws
            w = alpha * T
            ws = ws + w
            t = t + F32(deltas[i, 1])
            d = d + w * t
            c = c + w * rgbs[i].astype(F32)
            if e is not None:
                e = e + w * extra_in[i].astype(F32)
            step += 1
            if T * (F32(1.0) - alpha) < T_thresh:
                # terminated by opacity: consume no more rows
                step = -1
                break
        if step < n_step:            # ended (no more samples) or terminated
            rays_alive[n] = -1
        else:
            rays_t[r] = t
        weights_sum[r] = ws
        depth[r] = d
        image[r] = c
        if e is not None:
            extra_acc[r] = e
