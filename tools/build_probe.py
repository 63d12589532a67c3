"""Builds ablation variants of the library for profiling (never shipped) into tools/_probe/: probe modes 2 (no MLP) and 3
(per-workgroup time stamps) of the fused field kernel, or NAME=-Dflag[,-Dflag] for other compile-time variants
(-DINR_PROBE_STATIC=1, -DINR_PROBE_SLOW_XCD=1, ...).  Select one with INR_LIB_PATH=tools/_probe/libinr_<name>.so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instance_nerf_amd import build as b  # noqa: E402

out = os.path.join(ROOT, "tools", "_probe")
os.makedirs(out, exist_ok=True)
# arguments: probe modes ("1", "2") or NAME=-Dflag[,-Dflag] for other compile-time variants
for mode in sys.argv[1:] or ["2", "3"]:
    if "=" in mode:
        name, flags = mode.split("=", 1)
        defs = flags.split(",")
    else:
        name, defs = f"probe{mode}", [f"-DINR_PROBE_MODE={mode}"]
    if any(d.startswith("-DINR_PROBE") for d in defs) and "-DINR_PROBE_BUILD" not in defs:
        # the ablation hooks live in csrc/probe/field_probe.h, which the product build never includes (round 6): a probe
        # switch without this define is a compile error in field_fused.hip
        defs = ["-DINR_PROBE_BUILD"] + defs
    objs = []
    for src in b.SOURCES:
        obj = os.path.join(out, f"{src[:-4]}.{name}.o")
        subprocess.check_call(["hipcc", "-x", "hip", "-c", os.path.join(b.CSRC, src), "-o", obj] + defs + b.FLAGS)
        objs.append(obj)
    lib = os.path.join(out, f"libinr_{name}.so")
    subprocess.check_call(["hipcc", "-shared", "-o", lib] + objs + ["--offload-arch=gfx950"])
    print(lib)
