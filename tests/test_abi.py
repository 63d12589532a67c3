"""The C-ABI library loads on a CPU-only box and exports every symbol include/inr.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "inr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(inr_[a-zA-Z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from instance_nerf_amd import _lib
    assert sorted(_lib.EXPORTS) == _declared()


def _prototypes():
    """name -> list of parameter type strings, parsed from the header."""
    src = open(os.path.join(ROOT, "include", "inr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(inr_[a-zA-Z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = " ".join(m.group(2).split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        out[m.group(1)] = params
    return out


def test_ctypes_signatures_match_the_header():
    """Every argtypes list of the binding has the arity of its prototype, and pointer / float / 64-bit / 32-bit
    parameters sit where the header puts them (a wrong ctypes width corrupts arguments silently)."""
    import ctypes
    from instance_nerf_amd import _lib
    protos = _prototypes()
    assert sorted(protos) == _declared()
    ptr_like = (ctypes.c_void_p, ctypes.c_char_p)
    for name, params in protos.items():
        restype, argtypes = _lib._SIGS[name]
        assert len(argtypes) == len(params), (name, len(argtypes), params)
        for t, decl in zip(argtypes, params):
            is_ptr = "*" in decl or decl.split()[0] == "inr_stream_t"
            if is_ptr:
                assert t in ptr_like or hasattr(t, "contents") or t is _lib.P, (name, decl, t)
            elif decl.startswith("float"):
                assert t is ctypes.c_float, (name, decl, t)
            elif decl.startswith("int64_t"):
                assert t is ctypes.c_int64, (name, decl, t)
            elif decl.startswith(("int32_t", "int ")):
                assert t is ctypes.c_int32, (name, decl, t)
            else:
                raise AssertionError(f"{name}: unhandled parameter type {decl!r}")


def test_library_loads_and_exports_all_symbols():
    from instance_nerf_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from instance_nerf_amd import build
        build.build(verbose=False)
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "inr.h")).read()
    assert lib.inr_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define INR_ABI_VERSION (\d+)", header).group(1))
    for name in _declared():
        assert hasattr(lib, name), name


def test_no_cpu_fallback():
    """CPU tensors are rejected loudly: the product path never silently runs elsewhere."""
    import torch
    from instance_nerf_amd import raymarching
    with pytest.raises(RuntimeError, match="GPU tensor"):
        raymarching.near_far_from_aabb(torch.zeros(4, 3), torch.ones(4, 3), torch.tensor([-1., -1, -1, 1, 1, 1]))


def test_host_side_packing_roundtrip():
    """Weight packing is host code.  Default build: bf16 head + remainder per weight
    ([mt][step][hi|lo][lane][8]); every weight appears exactly once and hi + lo reproduces it to 2^-16."""
    import ctypes
    import numpy as np
    from instance_nerf_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    ws = [rng.normal(size=s).astype(np.float32) for s in [(64, 32), (16, 64), (64, 31), (64, 64), (3, 64)]]
    out = np.zeros(lib.inr_nerf_packed_floats(), np.float32)
    rc = lib.inr_nerf_pack_weights(*[w.ctypes.data_as(ctypes.c_void_p) for w in ws], out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    h = out.view(np.uint16).astype(np.uint32)
    vals = (h << 16).view(np.float32).reshape(-1, 2, 64, 8)          # [mt*step, hi|lo, lane, e]
    rec = (vals[:, 0] + vals[:, 1]).ravel()
    nz = np.sort(rec[rec != 0])
    ref = np.sort(np.concatenate([w.ravel() for w in ws]))
    assert nz.shape == ref.shape
    assert np.allclose(nz, ref, rtol=2.0 ** -15, atol=1e-30)


def test_product_never_imports_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "instance_nerf_amd")):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M):
                    bad.append(f)
    assert not bad


def test_missing_library_fails_loudly(monkeypatch):
    """No silent fallback: without libinr_hip.so every op raises (checked on a fresh loader state)."""
    import pytest
    from instance_nerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libinr_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_roi_align_backward_cost_model_is_host_arithmetic():
    """inr_roi_align_3d_backward_prefers_workspace (round 6): the workspace backward only where the voxels inside the
    RoIs' regions exceed about half the N*V volume; with the coverage unknown the lower bound K * min(bins, V) decides."""
    from instance_nerf_amd import _lib
    lib = _lib.load()
    f = lib.inr_roi_align_3d_backward_prefers_workspace
    assert f(1, 256, 40, 40, 40, 256, 10, 10, 10, -1) == 1               # BASELINE configs[4]: 0.34 ms against 0.58 in place
    assert f(1, 256, 40, 40, 40, 256, 10, 10, 10, 200_000) == 1
    assert f(1, 256, 80, 80, 80, 64, 7, 7, 7, -1) == 0                   # a fine pyramid level, few small boxes: 0.17 vs 0.38 ms
    assert f(1, 256, 80, 80, 80, 64, 7, 7, 7, 15_000) == 0
    assert f(1, 256, 80, 80, 80, 512, 7, 7, 7, 2_990_000) == 1           # the same level under 512 large boxes: 4.8 vs 2.8 ms
    assert f(1, 256, 20, 20, 20, 512, 7, 7, 7, 107_000) == 1             # a coarse level: the volume passes cost nothing
    assert f(1, 64, 40, 40, 40, 64, 7, 7, 7, 4_500) == 0                 # few channels, few small boxes: 0.023 vs 0.032 ms
    assert f(1, 250, 40, 40, 40, 256, 10, 10, 10, -1) == 0               # C % 16 != 0: the form does not exist
    assert f(1, 256, 40, 40, 40, 0, 10, 10, 10, -1) == 0 and f(-1, 256, 40, 40, 40, 256, 10, 10, 10, -1) == 0
    assert lib.inr_roi_align_3d_backward_workspace_bytes(1, 256, 80, 80, 80, 64, 7, 7, 7) > 0     # available, not preferred


def test_probe_switches_cannot_reach_the_product_build():
    """Round-5 verdict item 7a: ablation variants that produce wrong results (no MLP, plain stores instead of atomics, ...)
    live outside the sources build.py compiles - csrc/probe/, which only tools/build_probe.py adds by defining
    INR_PROBE_BUILD - and a stray -DINR_PROBE_* on the product build is a compile error, not a silently wrong library."""
    import shutil
    import subprocess
    from instance_nerf_amd import build
    for src in build.SOURCES:
        text = open(os.path.join(build.CSRC, src)).read()
        assert "SEP_BWD_PROBE" not in text and "SEP_NO_ZMASK" not in text, src
        assert not re.search(r"#\s*if\s+INR_PROBE_(MODE|STATIC|SLOW_XCD)", text), src      # only the hook macros remain
    assert os.path.exists(os.path.join(build.CSRC, "probe", "field_probe.h"))
    assert "probe" not in " ".join(build.SOURCES + build.HEADERS)
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    cmd = [hipcc, "-x", "hip", "-E", os.path.join(build.CSRC, "field_fused.hip"), "-o", os.devnull, "--offload-arch=gfx950",
           "--cuda-device-only", "-std=c++17"]
    bad = subprocess.run(cmd + ["-DINR_PROBE_MODE=2"], capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "INR_PROBE_BUILD" in bad.stderr
    ok = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-500:]
