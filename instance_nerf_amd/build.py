"""Builds libinr_hip.so (the C-ABI library declared in include/inr.h) for gfx950.

hipcc cross-compiles without a GPU.  The library is built IN-TREE
(instance_nerf_amd/csrc/libinr_hip.so) so that it travels with the source
snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libinr_hip.so")
SOURCES = ["raymarch.hip", "encoders.hip", "field_fused.hip", "roialign.hip"]
HEADERS = ["common.h", "grid_common.h", os.path.join("..", "..", "include", "inr.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


# which files define which measured kernels: a committed profile belongs to these files only (an edit of the RoIAlign or the
# marchers, or of a comment in include/inr.h, does not make a measurement of the field kernel stale)
KERNEL_FILES = {
    "field": ["field_fused.hip", "common.h", "grid_common.h"],        # k_nerf_fwd, k_instance_fwd, k_*_head_bwd, k_nerf_fwd_dirs
    "scatter": ["encoders.hip", "common.h", "grid_common.h"],         # k_grid_bwd, k_adam_*
}


def source_sha(kind=None):
    """sha256 over the kernel sources and headers the library is built from - `kind` None: all of them; "field" /
    "scatter": the files that define that group of kernels (KERNEL_FILES).  Profiles that describe a particular build of
    a kernel (profiles/r0x_traffic.json, _mfma, _bound_traffic: "field"; _scatter_requests: "scatter") carry the sha of
    their group, and bench.py only quotes them while it matches."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(KERNEL_FILES[kind] if kind else SOURCES + HEADERS):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


LIB_FP32 = os.path.join(CSRC, "libinr_hip_fp32.so")


def build_fp32(force=False, verbose=True):
    """The exact-fp32 A/B library: the same sources with -DINR_MLP_FP32=1 (MLP GEMMs on v_mfma_f32_16x16x4_f32 instead
    of the 3-term bf16 split).  Inference AND training entry points since round 5 (the device-side weight packers write
    the fp32 fragment layout there); only the opt-in -O variants (`*_fast`) are absent.  Loaded in child processes by
    tests/test_gpu_parity.py::test_exact_fp32_mlp_build / ::test_training_gradients_under_the_exact_fp32_build."""
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(LIB_FP32) and all(os.path.getmtime(d) <= os.path.getmtime(LIB_FP32) for d in deps):
        return LIB_FP32
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".fp32.o"))
        cmd = [hipcc, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj, "-DINR_MLP_FP32=1"] + FLAGS
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, "-shared", "-o", LIB_FP32] + objs + ["--offload-arch=gfx950"])
    return LIB_FP32


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libinr_hip.so")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj] + FLAGS
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_fp32(force="--force" in sys.argv)
    print(LIB)
