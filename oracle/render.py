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


def instance_ce_loss(logits, labels):
    """Cross entropy over rendered logits, ignore_index = -1, mean over kept rays."""
    labels = torch.as_tensor(labels, dtype=torch.int64)
    return torch.nn.functional.cross_entropy(logits, labels, ignore_index=-1)
