"""Does it pay to keep the next view's MARCH off most of the chip?  In the pipelined frame loop the march of view i+1
runs beside the field kernel of view i on all 256 CUs and slows it by about its own duration.  Here the march kernels
go to a stream created with a CU mask (hipExtStreamCreateWithCUMask) - a fraction of the CUs - while the field kernel
keeps the whole chip; its hybrid tile schedule moves work away from slower workgroups by itself.
python tools/cu_mask_probe.py [mask ...]     mask = none | lowN (the N lowest bits) | strideN (every N-th bit)"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import build_network  # noqa: E402
from instance_nerf_amd import raymarching  # noqa: E402
from instance_nerf_amd.nerf.renderer import FramePipeline  # noqa: E402
from instance_nerf_amd.nerf.utils import get_rays  # noqa: E402

dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so")
net, room = build_network(dev)
poses, intr, H, W = room.cameras()
pd = torch.from_numpy(poses).to(dev)
rays = []
for v in range(pd.shape[0]):
    r = get_rays(pd[v:v + 1], intr, H, W, patch=4)
    rays.append((r["rays_o"], r["rays_d"]))


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(st.value, device=dev)


def mask_bits(name):
    if name.startswith("low"):
        return (1 << int(name[3:])) - 1
    if name.startswith("stride"):
        n = int(name[6:])
        return sum(1 << i for i in range(0, 256, n))
    raise ValueError(name)


real_march = raymarching.march_rays_patch
real_near_far = raymarching.near_far_from_aabb


def run(name):
    side = {}
    if name != "none":
        bits = mask_bits(name)

        def on_side(fn):
            def wrapper(*a, **kw):
                main = torch.cuda.current_stream()
                key = main.cuda_stream
                if key not in side:
                    side[key] = masked_stream(bits)
                mk = side[key]
                mk.wait_stream(main)
                with torch.cuda.stream(mk):
                    out = fn(*a, **kw)
                main.wait_stream(mk)
                for t in out:
                    if torch.is_tensor(t):
                        t.record_stream(main)
                return out
            return wrapper
        raymarching.march_rays_patch = on_side(real_march)
    else:
        raymarching.march_rays_patch = real_march
    ev = []
    inner = net.forward_table.__func__ if hasattr(net.forward_table, "__func__") else None
    orig_ft = type(net).forward_table

    def timed(self, *a, **kw):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        out = orig_ft(self, *a, **kw)
        e1.record(st)
        ev.append((e0, e1))
        return out
    net.forward_table = timed.__get__(net)
    with FramePipeline(net) as pipe, torch.no_grad():
        for v in range(6):
            pipe.render(*rays[v % len(rays)], bg_color=1, infer_mode="fused")
        pipe.synchronize()
        ev.clear()
        t0 = time.perf_counter()
        n = 0
        outs = []
        for v in range(24):
            outs.append(pipe.render(*rays[v % len(rays)], bg_color=1, infer_mode="fused")["num_samples"])
        pipe.synchronize()
        dt = time.perf_counter() - t0
        n = sum(int(c[0]) for c in outs)
    kms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    del net.forward_table
    print(f"march on {name:10s}: {dt / 24 * 1e3:.3f} ms per frame, {n / dt / 1e9:.3f} Gsamples/s, field kernel {kms:.3f} ms", flush=True)


for m in sys.argv[1:] or ["none", "low32", "stride8", "low64", "stride4", "low128", "none"]:
    try:
        run(m)
    except Exception as e:                              # noqa: BLE001
        print(f"march on {m}: {type(e).__name__}: {e}")
raymarching.march_rays_patch = real_march
