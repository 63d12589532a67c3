"""Occupancy-grid ray marching (oracle; test infrastructure only).

Follows SURVEY.md Appendix A.1 "march_rays_train" and "Inference loop"
(rows a4/a5 of section 8a; upstream ``raymarching.march_rays_train`` /
``march_rays`` of the un-vendored submodule pinned at
/root/reference/README.md:27,59).  Parity unpinned - see ``oracle/__init__``.

Arithmetic contract (what "bit-exact" means for the HIP kernels): every
operation below is a single IEEE-754 binary32 operation, evaluated in exactly
the written order, with no fused multiply-add.

    p      = clamp(o + t*d, -bound, bound)                       (mul, add)
    dt     = clamp(t*dt_gamma, dt_min, dt_max)
    level  = max(mip_from_pos(p), mip_from_dt(dt))
    mb     = min(2^level, bound);  rmb = 1/mb
    n      = clamp((int)(((p*rmb + 1)*0.5)*H), 0, H-1)            (trunc to zero)
    bit    = level*H^3 + morton(n)
    hit :  emit (p, d, dt, t+dt - t_prev);  t_prev = t = t + dt
    miss:  tc = (((n + 0.5 + 0.5*sgn(d))*(1/H)*2 - 1)*mb - p) * (1/d)
           tt = t + max(0, min(tc.x, tc.y, tc.z))                 (minNum: NaN ignored)
           do  t = t + clamp(t*dt_gamma, dt_min, dt_max)  while t < tt
    dt_min = 2*sqrt(3)/max_steps,  dt_max = 2*sqrt(3)*2^(C-1)/H   (fp32 constants
    formed as float32(2*sqrt(3)) / max_steps etc., see ``dt_limits``)
"""
import numpy as np

from .occupancy import morton3D

F32 = np.float32
SQRT3_2 = F32(3.4641016151377544)   # float32(2*sqrt(3))


def dt_limits(max_steps, cascade, H):
    dt_min = SQRT3_2 / F32(max_steps)
    dt_max = SQRT3_2 * F32(2 ** (cascade - 1)) / F32(H)
    return F32(dt_min), F32(dt_max)


def _clamp(x, lo, hi):
    return np.minimum(np.maximum(x, lo), hi)


def _mip_from_pos(p, cascade):
    mx = np.max(np.abs(p), axis=-1)
    _, e = np.frexp(mx)
    return np.clip(e, 0, cascade - 1).astype(np.int32)


def _mip_from_dt(dt, H, cascade):
    _, e = np.frexp(dt * F32(H) * F32(0.5))
    return np.clip(e, 0, cascade - 1).astype(np.int32)


class _Marcher:
    """Vectorised state machine shared by the train and inference marchers."""

    def __init__(self, rays_o, rays_d, bitfield, bound, cascade, H, dt_gamma, max_steps):
        self.o = np.asarray(rays_o, dtype=F32)
        self.d = np.asarray(rays_d, dtype=F32)
        self.bits = np.asarray(bitfield, dtype=np.uint8)
        self.bound = F32(bound)
        self.C = int(cascade)
        self.H = int(H)
        self.dt_gamma = F32(dt_gamma)
        self.dt_min, self.dt_max = dt_limits(max_steps, cascade, H)
        with np.errstate(divide="ignore"):
            self.rd = F32(1.0) / self.d
        self.sgn = np.copysign(F32(1.0), self.d).astype(F32)

    def dt_of(self, t):
        return _clamp(t * self.dt_gamma, self.dt_min, self.dt_max).astype(F32)

    def probe(self, idx, t):
        """For rays idx at parameter t: (p, dt, occupied, tt)."""
        o, d = self.o[idx], self.d[idx]
        p = _clamp(o + t[:, None] * d, -self.bound, self.bound).astype(F32)
        dt = self.dt_of(t)
        level = np.maximum(_mip_from_pos(p, self.C), _mip_from_dt(dt, self.H, self.C))
        mb = np.minimum(np.ldexp(F32(1.0), level).astype(F32), self.bound)
        rmb = F32(1.0) / mb
        f = ((p * rmb[:, None] + F32(1.0)) * F32(0.5)) * F32(self.H)
        n = np.clip(f.astype(np.int32), 0, self.H - 1)
        bit = level.astype(np.int64) * (self.H ** 3) + morton3D(n).astype(np.int64)
        occ = (self.bits[bit >> 3] >> (bit & 7).astype(np.uint8)) & 1
        with np.errstate(invalid="ignore", over="ignore"):
            a = n.astype(F32) + F32(0.5) + F32(0.5) * self.sgn[idx]
            c = ((a * F32(1.0 / self.H) * F32(2.0) - F32(1.0)) * mb[:, None] - p) * self.rd[idx]
            tm = np.fmin(c[:, 0], np.fmin(c[:, 1], c[:, 2]))
            tt = t + np.fmax(F32(0.0), tm)
        return p, dt, occ.astype(bool), tt.astype(F32)


def march_rays_train(rays_o, rays_d, bitfield, bound, cascade, H, nears, fars,
                     noises=None, dt_gamma=0.0, max_steps=1024, M=None):
    """Deterministic two-pass march.

    Returns dict(xyzs f32[M,3], dirs f32[M,3], deltas f32[M,2], rays i32[N,3],
    total int).  rays[n] = (n, offset, count) with offset = exclusive scan of
    the counts in ray order (canonical choice, see oracle/__init__).  Rays with
    count 0 keep a row.  If ``M`` is given and offset+count > M the ray is
    dropped exactly as upstream drops overflowing rays: its ``rays`` row is
    written, no samples are.  Unused sample rows are zero.
    """
    mk = _Marcher(rays_o, rays_d, bitfield, bound, cascade, H, dt_gamma, max_steps)
    N = mk.o.shape[0]
    nears = np.asarray(nears, dtype=F32)
    fars = np.asarray(fars, dtype=F32)
    noises = np.zeros(N, F32) if noises is None else np.asarray(noises, dtype=F32)
    t0 = (nears + mk.dt_of(nears) * noises).astype(F32)

    t = t0.copy()
    tt = np.full(N, -np.inf, dtype=F32)          # skip target; t < tt means "skipping"
    last_t = t0.copy()
    count = np.zeros(N, np.int32)
    s_ray, s_p, s_dt, s_del = [], [], [], []
    active = np.nonzero((t < fars) & (count < max_steps))[0]
    while active.size:
        ta = t[active]
        skipping = ta < tt[active]
        # rays mid-skip take one more whole-dt step
        sk = active[skipping]
        if sk.size:
            t[sk] = t[sk] + mk.dt_of(t[sk])
        pr = active[~skipping]
        if pr.size:
            p, dt, occ, tnew = mk.probe(pr, t[pr])
            hit = pr[occ]
            if hit.size:
                tn = (t[hit] + dt[occ]).astype(F32)
                s_ray.append(hit)
                s_p.append(p[occ])
                s_dt.append(dt[occ])
                s_del.append((tn - last_t[hit]).astype(F32))
                t[hit] = tn
                last_t[hit] = tn
                count[hit] += 1
            ms = pr[~occ]
            if ms.size:
                tt[ms] = tnew[~occ]
                t[ms] = t[ms] + mk.dt_of(t[ms])        # the "do" of do-while
        active = active[(t[active] < fars[active]) & (count[active] < max_steps)]

    offsets = np.zeros(N, np.int64)
    np.cumsum(count[:-1], out=offsets[1:])
    total = int(count.sum())
    if M is None:
        M = total
    rays = np.stack([np.arange(N), offsets, count], -1).astype(np.int32)
    xyzs = np.zeros((M, 3), F32)
    dirs = np.zeros((M, 3), F32)
    deltas = np.zeros((M, 2), F32)
    if s_ray:
        r = np.concatenate(s_ray)
        P = np.concatenate(s_p)
        DT = np.concatenate(s_dt)
        DL = np.concatenate(s_del)
        # k-th emission of ray r (emission order is time order, stable sort keeps it)
        order = np.argsort(r, kind="stable")
        r, P, DT, DL = r[order], P[order], DT[order], DL[order]
        k = np.arange(r.size) - np.repeat(np.cumsum(count) - count, count)
        slot = offsets[r] + k
        keep = (offsets[r] + count[r]) <= M
        slot, r, P, DT, DL = slot[keep], r[keep], P[keep], DT[keep], DL[keep]
        xyzs[slot] = P
        dirs[slot] = mk.d[r]
        deltas[slot, 0] = DT
        deltas[slot, 1] = DL
    return dict(xyzs=xyzs, dirs=dirs, deltas=deltas, rays=rays, total=total)


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bitfield, bound,
               cascade, H, nears, fars, dt_gamma=0.0, max_steps=1024):
    """Inference step: up to n_step samples for each live ray, no jitter.

    rays_alive i32[n_alive] (indices into the N rays), rays_t f32[N] (not
    modified here - composite_rays advances it).  Returns xyzs, dirs
    f32[n_alive*n_step,3], deltas f32[n_alive*n_step,2]; rows a ray did not
    fill stay zero (deltas[.,0] == 0 marks the end of a ray for composite_rays).
    """
    mk = _Marcher(rays_o, rays_d, bitfield, bound, cascade, H, dt_gamma, max_steps)
    alive = np.asarray(rays_alive[:n_alive], dtype=np.int64)
    fars_a = np.asarray(fars, dtype=F32)[alive]
    t = np.asarray(rays_t, dtype=F32)[alive].copy()
    last_t = t.copy()
    tt = np.full(n_alive, -np.inf, dtype=F32)
    step = np.zeros(n_alive, np.int32)
    xyzs = np.zeros((n_alive * n_step, 3), F32)
    dirs = np.zeros((n_alive * n_step, 3), F32)
    deltas = np.zeros((n_alive * n_step, 2), F32)
    act = np.nonzero((t < fars_a) & (step < n_step))[0]
    while act.size:
        skipping = t[act] < tt[act]
        sk = act[skipping]
        if sk.size:
            t[sk] = t[sk] + mk.dt_of(t[sk])
        pr = act[~skipping]
        if pr.size:
            p, dt, occ, tnew = mk.probe(alive[pr], t[pr])
            hit = pr[occ]
            if hit.size:
                tn = (t[hit] + dt[occ]).astype(F32)
                slot = hit * n_step + step[hit]
                xyzs[slot] = p[occ]
                dirs[slot] = mk.d[alive[hit]]
                deltas[slot, 0] = dt[occ]
                deltas[slot, 1] = tn - last_t[hit]
                t[hit] = tn
                last_t[hit] = tn
                step[hit] += 1
            ms = pr[~occ]
            if ms.size:
                tt[ms] = tnew[~occ]
                t[ms] = t[ms] + mk.dt_of(t[ms])
        act = act[(t[act] < fars_a[act]) & (step[act] < n_step)]
    return xyzs, dirs, deltas
