"""ctypes binding of the C oracle (oracle/c/inr_oracle.c).  TEST INFRASTRUCTURE ONLY - see oracle/__init__.

The C file is a second, independent restatement of SURVEY.md Appendix A (scalar, one ray at a time);
``tests/test_oracle_c.py`` checks it against the numpy/torch oracle and the golden vectors, and
``bench.py``'s ``cpu_baseline`` leg times ``render`` (OpenMP over rays) on the host cores.
Parity unpinned (oracle/__init__).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "c", "inr_oracle.c")
# INR_ORACLE_LIB: another build of the same source (tests/test_oracle_c.py::test_ubsan_build points it at the
# -fsanitize=undefined library in a child process)
LIB_PATH = os.environ.get("INR_ORACLE_LIB") or os.path.join(_HERE, "c", "liborc.so")
_lib = None

F32, I32, U8 = np.float32, np.int32, np.uint8


class _Grid(ctypes.Structure):
    _fields_ = [("num_levels", ctypes.c_int32), ("offsets", ctypes.c_uint32 * 17), ("scales", ctypes.c_float * 16),
                ("resolutions", ctypes.c_uint32 * 16), ("hashed", ctypes.c_uint32 * 16)]


class _Nerf(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("embeddings", "sigma_w0", "sigma_w1", "color_w0", "color_w1", "color_w2")]


class _Inst(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("embeddings", "w0", "w1", "w2")] + [("K", ctypes.c_int32)]


def build(force=False):
    if os.environ.get("INR_ORACLE_LIB"):
        return LIB_PATH                                  # an explicitly chosen build: never rebuilt here
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(_SRC):
        subprocess.check_call(["make", "-s", "-B", "-C", os.path.join(_HERE, "c"), "liborc.so"])
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.orc_march_count.restype = ctypes.c_int64
        _lib.orc_render.restype = ctypes.c_int64
        _lib.orc_num_threads.restype = ctypes.c_int
        if "OMP_NUM_THREADS" not in os.environ:
            _lib.orc_set_num_threads(usable_cores())
    return _lib


def usable_cores():
    """Cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota (a container with
    256 visible cores and a 16-core quota runs 128 OpenMP threads slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def num_threads():
    return int(load().orc_num_threads())


def set_num_threads(n):
    load().orc_set_num_threads(int(n))


def _a(x, dt):
    return np.ascontiguousarray(np.asarray(x.detach().numpy() if hasattr(x, "detach") else x), dtype=dt)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _grid(table):
    g = _Grid()
    L = int(table["num_levels"])
    g.num_levels = L
    for i in range(L + 1):
        g.offsets[i] = int(table["offsets"][i])
    for i in range(L):
        g.scales[i] = float(table["scales"][i])
        g.resolutions[i] = int(table["resolutions"][i])
        g.hashed[i] = int(table["hashed"][i])
    return g


def _nerf(p):
    keep = [_a(p[k], F32) for k in ("embeddings", "sigma_w0", "sigma_w1", "color_w0", "color_w1", "color_w2")]
    return _Nerf(*[k.ctypes.data for k in keep]), keep


def _inst(p):
    keep = [_a(p[k], F32) for k in ("inst_embeddings", "inst_w0", "inst_w1", "inst_w2")]
    return _Inst(*[k.ctypes.data for k in keep], int(keep[3].shape[0])), keep


def morton3D(coords):
    c = _a(coords, I32)
    out = np.empty(c.shape[0], np.uint32)
    load().orc_morton3D(_p(c), ctypes.c_int64(c.shape[0]), _p(out))
    return out


def morton3D_invert(idx):
    i = _a(idx, np.uint32)
    out = np.empty((i.shape[0], 3), I32)
    load().orc_morton3D_invert(_p(i), ctypes.c_int64(i.shape[0]), _p(out))
    return out


def packbits(grid, thresh):
    g = _a(grid, F32).reshape(-1)
    out = np.empty(g.shape[0] // 8, U8)
    load().orc_packbits(_p(g), ctypes.c_int64(g.shape[0]), ctypes.c_float(thresh), _p(out))
    return out


def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    o, d, bb = _a(rays_o, F32), _a(rays_d, F32), _a(aabb, F32)
    n, f = np.empty(o.shape[0], F32), np.empty(o.shape[0], F32)
    load().orc_near_far(_p(o), _p(d), _p(bb), ctypes.c_float(min_near), ctypes.c_int64(o.shape[0]), _p(n), _p(f))
    return n, f


def march_rays_train(rays_o, rays_d, bitfield, bound, cascade, H, nears, fars, noises=None, dt_gamma=0.0,
                     max_steps=1024, M=None):
    """Same contract as oracle.march.march_rays_train (deterministic scan, overflowing rays dropped)."""
    o, d, bits = _a(rays_o, F32), _a(rays_d, F32), _a(bitfield, U8)
    n, f = _a(nears, F32), _a(fars, F32)
    nz = None if noises is None else _a(noises, F32)
    N = o.shape[0]
    rays = np.zeros((N, 3), I32)
    args = (_p(o), _p(d), _p(bits), ctypes.c_float(bound), ctypes.c_int32(cascade), ctypes.c_int32(H), _p(n), _p(f),
            _p(nz), ctypes.c_float(dt_gamma), ctypes.c_int32(max_steps), ctypes.c_int64(N))
    total = int(load().orc_march_count(*args, _p(rays)))
    M = total if M is None else int(M)
    xyzs, dirs, deltas = np.zeros((M, 3), F32), np.zeros((M, 3), F32), np.zeros((M, 2), F32)
    load().orc_march_write(*args, _p(rays), ctypes.c_int64(M), _p(xyzs), _p(dirs), _p(deltas))
    return dict(xyzs=xyzs, dirs=dirs, deltas=deltas, rays=rays, total=total)


def grid_encode(x, embeddings, bound, table):
    xx, emb, g = _a(x, F32), _a(embeddings, F32), _grid(table)
    out = np.empty((xx.shape[0], 2 * g.num_levels), F32)
    load().orc_grid_encode(_p(xx), _p(emb), ctypes.byref(g), ctypes.c_float(bound), ctypes.c_int64(xx.shape[0]), _p(out))
    return out


def sh_encode(d):
    dd = _a(d, F32)
    out = np.empty((dd.shape[0], 16), F32)
    load().orc_sh4(_p(dd), ctypes.c_int64(dd.shape[0]), _p(out))
    return out


def nerf_forward(x, d, p, bound, table, want_geo=False):
    xx, dd, g = _a(x, F32), _a(d, F32), _grid(table)
    P, keep = _nerf(p)
    M = xx.shape[0]
    sigma, rgb = np.empty(M, F32), np.empty((M, 3), F32)
    geo = np.empty((M, 15), F32) if want_geo else None
    load().orc_nerf_forward(_p(xx), _p(dd), ctypes.c_int64(M), ctypes.c_float(bound), ctypes.byref(g), ctypes.byref(P),
                            _p(sigma), _p(rgb), _p(geo))
    return (sigma, rgb, geo) if want_geo else (sigma, rgb)


def instance_logits(x, p, bound, table):
    xx, g = _a(x, F32), _grid(table)
    I, keep = _inst(p)
    out = np.empty((xx.shape[0], I.K), F32)
    load().orc_instance_forward(_p(xx), ctypes.c_int64(xx.shape[0]), ctypes.c_float(bound), ctypes.byref(g), ctypes.byref(I),
                                _p(out))
    return out


def composite_rays_train(sigmas, rgbs, deltas, rays, T_thresh=1e-4, extra=None):
    s, c, dl, rr = _a(sigmas, F32), _a(rgbs, F32), _a(deltas, F32), _a(rays, I32)
    N, M = rr.shape[0], s.shape[0]
    ex = None if extra is None else _a(extra, F32)
    K = 0 if ex is None else ex.shape[1]
    ws, dp, img = np.zeros(N, F32), np.zeros(N, F32), np.zeros((N, 3), F32)
    eo = None if ex is None else np.zeros((N, K), F32)
    w = np.zeros(M, F32)
    load().orc_composite_train(_p(s), _p(c), _p(dl), _p(rr), ctypes.c_int64(N), ctypes.c_int64(M), ctypes.c_float(T_thresh),
                               _p(ex), ctypes.c_int32(K), _p(ws), _p(dp), _p(img), _p(eo), _p(w))
    return dict(weights_sum=ws, depth=dp, image=img, extra=eo, weights=w)


def render(rays_o, rays_d, p, table, bitfield, bound=1.0, cascade=1, H=128, min_near=0.2, dt_gamma=0.0,
           max_steps=1024, T_thresh=1e-4, bg_color=1.0, with_instance=False, density_scale=1.0, absolute_depth=False):
    """The whole path, one ray at a time (OpenMP over rays): what oracle.render.render_train computes with
    perturb off; ``absolute_depth=True`` accumulates the depth over the absolute ray parameter as upstream's
    INFERENCE compositing does (oracle.render.render_infer), False over t counted from the first step as its
    training compositing does.  -> dict(image, depth, weights_sum, instance | None, counts, total)."""
    o, d, bits, g = _a(rays_o, F32), _a(rays_d, F32), _a(bitfield, U8), _grid(table)
    P, keep = _nerf(p)
    I, keep2 = _inst(p) if with_instance else (None, None)
    N = o.shape[0]
    img, dp, ws = np.empty((N, 3), F32), np.empty(N, F32), np.empty(N, F32)
    inst = np.empty((N, I.K), F32) if with_instance else None
    counts = np.empty(N, I32)
    total = load().orc_render(_p(o), _p(d), ctypes.c_int64(N), _p(bits), ctypes.c_float(bound), ctypes.c_int32(cascade),
                              ctypes.c_int32(H), ctypes.c_float(min_near), ctypes.c_float(dt_gamma),
                              ctypes.c_int32(max_steps), ctypes.c_float(T_thresh), ctypes.c_float(bg_color),
                              ctypes.c_float(density_scale), ctypes.c_int32(1 if absolute_depth else 0), ctypes.byref(g),
                              ctypes.byref(P),
                              ctypes.byref(I) if with_instance else None, _p(img), _p(dp), _p(ws), _p(inst), _p(counts))
    return dict(image=img, depth=dp, weights_sum=ws, instance=inst, counts=counts, total=int(total))
