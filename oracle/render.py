"""End-to-end render loops (oracle; test infrastructure only).

Follows SURVEY.md section 3.1/3.2 call stacks and Appendix A.1 "Inference
loop" (upstream ``NeRFRenderer.run_cuda`` of the un-vendored submodule pinned
at /root/reference/README.md:27,59).  Parity unpinned.
"""
import numpy as np
import torch

from . import composite as comp
from . import field, march
from .rays import near_far_from_aabb

F32 = np.float32


def aabb_of(bound):
    return np.asarray([-bound, -bound, -bound, bound, bound, bound], dtype=F32)


def render_train(rays_o, rays_d, p, table, bitfield, bound=1.0, cascade=1, H=128,
                 min_near=0.2, dt_gamma=0.0, max_steps=1024, T_thresh=1e-4,
                 noises=None, bg_color=1.0, with_instance=False, density_scale=1.0):
    """Training-mode render of N rays (differentiable through torch).

    Returns dict(image[N,3], depth[N], weights_sum[N], instance[N,K]|None,
    rays, xyzs, dirs, deltas, sigmas, rgbs, total).
    """
    nears, fars = near_far_from_aabb(rays_o, rays_d, aabb_of(bound), min_near)
    m = march.march_rays_train(rays_o, rays_d, bitfield, bound, cascade, H, nears, fars,
                               noises=noises, dt_gamma=dt_gamma, max_steps=max_steps)
    xyzs = torch.from_numpy(m["xyzs"])
    dirs = torch.from_numpy(m["dirs"])
    sigmas, rgbs = field.nerf_forward(xyzs, dirs, p, bound, table)
    sigmas = sigmas * density_scale
    extra = field.instance_logits(xyzs, p, bound, table) if with_instance else None
    c = comp.composite_rays_train(sigmas, rgbs, m["deltas"], m["rays"], T_thresh, extra=extra)
    image = c["image"] + (1 - c["weights_sum"])[:, None] * bg_color
    nf = torch.from_numpy(np.stack([nears, fars], -1))
    depth = torch.clamp(c["depth"] - nf[:, 0], min=0) / (nf[:, 1] - nf[:, 0])
    return dict(image=image, depth=depth, weights_sum=c["weights_sum"], instance=c["extra"],
                rays=m["rays"], xyzs=m["xyzs"], dirs=m["dirs"], deltas=m["deltas"],
                sigmas=sigmas, rgbs=rgbs, total=m["total"], nears=nears, fars=fars,
                raw_depth=c["depth"], weights=c["weights"])


@torch.no_grad()
def render_infer(rays_o, rays_d, p, table, bitfield, bound=1.0, cascade=1, H=128,
                 min_near=0.2, dt_gamma=0.0, max_steps=1024, T_thresh=1e-4,
                 bg_color=1.0, with_instance=False, density_scale=1.0, max_n_step=8):
    """Inference wavefront loop: n_step = clamp(N // n_alive, 1, max_n_step)."""
    N = rays_o.shape[0]
    nears, fars = near_far_from_aabb(rays_o, rays_d, aabb_of(bound), min_near)
    ws = np.zeros(N, F32)
    depth = np.zeros(N, F32)
    image = np.zeros((N, 3), F32)
    K = p["inst_w2"].shape[0] if with_instance else 0
    inst = np.zeros((N, K), F32) if with_instance else None
    rays_alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    n_alive = N
    evaluated = 0
    step = 0
    while step < max_steps and n_alive > 0:
        n_step = max(min(N // n_alive, max_n_step), 1)
        xyzs, dirs, deltas = march.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d,
                                              bitfield, bound, cascade, H, nears, fars,
                                              dt_gamma, max_steps)
        live = deltas[:, 0] > 0
        evaluated += int(live.sum())
        sig, rgb = field.nerf_forward(torch.from_numpy(xyzs), torch.from_numpy(dirs), p, bound, table)
        sig = (sig * density_scale).numpy()
        ex = field.instance_logits(torch.from_numpy(xyzs), p, bound, table).numpy() if with_instance else None
        comp.composite_rays(n_alive, n_step, rays_alive, rays_t, sig, rgb.numpy(), deltas,
                            ws, depth, image, T_thresh, extra_in=ex, extra_acc=inst)
        rays_alive = rays_alive[rays_alive >= 0]
        n_alive = rays_alive.shape[0]
        step += n_step
    image = image + (1 - ws)[:, None] * F32(bg_color)
    dnorm = np.clip(depth - nears, 0, None) / (fars - nears)
    return dict(image=image, depth=dnorm, weights_sum=ws, instance=inst, evaluated=evaluated,
                raw_depth=depth)


@torch.no_grad()
def render_generic(rays_o, rays_d, p, table, bound=1.0, min_near=0.2, num_steps=128, upsample_steps=128,
                   density_scale=1.0, bg_color=1.0):
    """Upstream's sampler WITHOUT an occupancy grid (``NeRFRenderer.run``; Appendix A does not spell
    it out - this is NeRF's coarse + importance sampling as torch-ngp runs it in evaluation mode: no jitter,
    deterministic inverse-CDF samples).  Written ray by ray with numpy searchsorted, independently of the
    product's vectorised version.  -> dict(image [N,3], depth [N], weights_sum [N])."""
    ro, rd = np.asarray(rays_o, F32), np.asarray(rays_d, F32)
    N = ro.shape[0]
    nears, fars = near_far_from_aabb(ro, rd, aabb_of(bound), min_near)
    image, depth, wsum = np.zeros((N, 3), F32), np.zeros(N, F32), np.zeros(N, F32)
    lo, hi = -F32(bound), F32(bound)

    def field_at(z, o, d):
        x = np.clip(o[None, :] + d[None, :] * z[:, None], lo, hi).astype(F32)
        den = field.density(torch.from_numpy(x), p, bound, table)
        return x, den["sigma"].numpy() * F32(density_scale), den["geo_feat"]

    def weights_of(z, sigma, dist):
        deltas = np.append(z[1:] - z[:-1], dist).astype(F32)
        alphas = 1 - np.exp(-deltas * sigma)
        T = np.cumprod(np.append(1.0, 1 - alphas + 1e-15))[:-1]
        return (alphas * T).astype(F32), deltas

    for n in range(N):
        if not (fars[n] > nears[n] and fars[n] < 3.0e38):
            image[n] = bg_color
            continue
        near, far = nears[n], fars[n]
        z = (near + (far - near) * np.linspace(0.0, 1.0, num_steps, dtype=F32)).astype(F32)
        dist = F32((far - near) / num_steps)
        x, sigma, geo = field_at(z, ro[n], rd[n])
        if upsample_steps > 0:
            w, deltas = weights_of(z, sigma, dist)
            mid = z[:-1] + 0.5 * deltas[:-1]                       # bin edges: T - 1 of them, T - 2 bins
            pdf = w[1:-1] + 1e-5
            pdf = pdf / pdf.sum()
            cdf = np.append(0.0, np.cumsum(pdf)).astype(F32)
            u = np.linspace(0.5 / upsample_steps, 1 - 0.5 / upsample_steps, upsample_steps, dtype=F32)
            idx = np.searchsorted(cdf, u, side="right")
            below, above = np.maximum(idx - 1, 0), np.minimum(idx, len(cdf) - 1)
            den = cdf[above] - cdf[below]
            den = np.where(den < 1e-5, 1.0, den)
            t = (u - cdf[below]) / den
            z_new = (mid[below] + t * (mid[above] - mid[below])).astype(F32)
            x2, s2, g2 = field_at(z_new, ro[n], rd[n])
            order = np.argsort(np.append(z, z_new), kind="stable")
            z = np.append(z, z_new)[order]
            x = np.concatenate([x, x2])[order]
            sigma = np.append(sigma, s2)[order]
            geo = torch.cat([geo, g2])[torch.from_numpy(order)]
        w, _ = weights_of(z, sigma, dist)
        d = torch.from_numpy(np.repeat(rd[n][None], len(z), 0))
        rgb = field.color(d, geo, p).numpy()
        rgb = np.where((w > 1e-4)[:, None], rgb, 0.0)
        wsum[n] = w.sum()
        depth[n] = (w * np.clip((z - near) / (far - near), 0, 1)).sum()
        image[n] = (w[:, None] * rgb).sum(0) + (1 - wsum[n]) * bg_color
    return dict(image=image, depth=depth, weights_sum=wsum)


def instance_ce_loss(logits, labels):
    """Cross entropy over rendered logits, ignore_index = -1, mean over kept rays."""
    labels = torch.as_tensor(labels, dtype=torch.int64)
    return torch.nn.functional.cross_entropy(logits, labels, ignore_index=-1)
