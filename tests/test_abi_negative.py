"""Negative paths of the C ABI ("returns a negative code, never exit()"; the habit avoided:
/root/reference/nerf_rcnn/model/rotated_iou/cuda_op/cuda_utils.h:26-35 `exit()`s on a CUDA error, utils.h:6-31 only
asserts).  Every export of include/inr.h is called with null pointers, with negative sizes, and - where the header
names a limit - with K > 64, 17 levels, misaligned buffers, too many tensors: each call must come back with
INR_EINVAL and a message from inr_last_error(), and the process must still be alive afterwards.  The library loads on
a CPU-only box and validation precedes every launch, so this is a CPU test; the calls run in a child process so that a
crash is a test failure, not the end of the test session."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EINVAL, ELAUNCH, ENODEV = -1, -2, -3

# exports whose arguments cannot be wrong (no pointers, every integer meaningful) or that are pure size queries
SIZE_QUERIES = {"inr_occ_sample_workspace_bytes", "inr_march_workspace_bytes", "inr_nerf_bwd_packed_floats",
                "inr_instance_bwd_packed_floats", "inr_instance_head_workspace_bytes",
                "inr_nerf_forward_table_sliced_workspace_bytes", "inr_nerf_packed_floats", "inr_instance_packed_floats",
                "inr_roi_align_3d_backward_workspace_bytes", "inr_linear_wgrad_workspace_bytes",
                "inr_march_write_fills_unowned_rows", "inr_roi_align_3d_backward_prefers_workspace"}
NO_BAD_VALUE = set()


@pytest.fixture(scope="module")
def results():
    from instance_nerf_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_negative_child.py")], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", "")))
    assert r.returncode == 0, f"the child died (rc {r.returncode}): an export crashed on bad arguments\n{r.stderr[-2000:]}"
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-1000:]
    out = json.loads(lines[-1])
    assert out["alive"] == [0, "reached the end"]
    return out


def test_every_export_is_covered(results):
    from instance_nerf_amd import _lib
    for name in _lib.EXPORTS:
        if name in ("inr_abi_version", "inr_last_error"):
            continue
        assert f"{name}:null" in results and f"{name}:negative" in results, name


def test_null_pointers_are_rejected_with_einval_and_a_message(results):
    from instance_nerf_amd import _lib
    import ctypes
    for name, (restype, argtypes) in _lib._SIGS.items():
        has_ptr = any(t is _lib.P or (isinstance(t, type) and issubclass(t, ctypes._Pointer)) for t in argtypes)
        if not has_ptr or name in SIZE_QUERIES:
            continue
        rc, msg = results[f"{name}:null"]
        assert rc == EINVAL and msg, (name, rc, msg)


def test_negative_sizes_are_rejected_before_any_launch(results):
    """Valid (host) pointers, every size -1: INR_EINVAL - except the entry points that take no size at all, whose launch
    then fails on this GPU-less box with INR_ELAUNCH / INR_ENODEV (still a return code, still alive), and the host-side
    packers, which succeed on valid host buffers."""
    from instance_nerf_amd import _lib
    import ctypes
    no_sizes_host = {"inr_nerf_pack_weights", "inr_nerf_pack_weights_f16"}
    for name, (restype, argtypes) in _lib._SIGS.items():
        if name in SIZE_QUERIES or name in NO_BAD_VALUE or name in ("inr_abi_version", "inr_last_error"):
            continue
        rc, msg = results[f"{name}:negative"]
        has_int = any(t in (ctypes.c_int32, ctypes.c_int64) for t in argtypes)
        if name in no_sizes_host:
            assert rc == 0, (name, rc, msg)
        elif not has_int:
            assert rc in (EINVAL, ELAUNCH, ENODEV) and msg, (name, rc, msg)
        elif name == "inr_set_march_mode":
            assert rc == 0                              # -1 IS a mode (automatic)
        elif name == "inr_device_info":
            assert rc in (EINVAL, ENODEV) and msg, (name, rc, msg)
        else:
            assert rc == EINVAL and msg, (name, rc, msg)


def test_size_queries_never_report_a_size_for_bad_arguments(results):
    for name in ("inr_instance_packed_floats", "inr_roi_align_3d_backward_workspace_bytes",
                 "inr_nerf_forward_table_sliced_workspace_bytes", "inr_occ_sample_workspace_bytes"):
        rc, _ = results[f"{name}:negative"]
        assert rc <= 0, (name, rc)
    assert results["inr_roi_align_3d_backward_prefers_workspace:negative"][0] == 0        # "no": never a preference for bad sizes
    assert results["inr_instance_packed_floats:K_65"][0] < 0 and "K" in results["inr_instance_packed_floats:K_65"][1]


@pytest.mark.parametrize("key,needle", [
    ("inr_grid_encode_forward:num_levels_17", "num_levels"), ("inr_grid_encode_backward:num_levels_17", "num_levels"),
    ("inr_nerf_forward:num_levels_17", "num_levels"), ("inr_instance_forward:num_levels_17", "num_levels"),
    ("inr_grid_encode_forward:level_dim_4", "level_dim"), ("inr_grid_encode_backward:level_dim_4", "level_dim"),
    ("inr_instance_forward:K_65", "K"), ("inr_instance_forward:K_80", "K"), ("inr_instance_forward_enc:K_80", "K"),
    ("inr_instance_pack_weights:K_65", "K"), ("inr_instance_pack_weights_device:K_80", "K"),
    ("inr_instance_head_backward:K_80", "K"), ("inr_instance_render:K_80", "K"), ("inr_cross_entropy:K_65", "K"),
    ("inr_composite_rays_extra_forward:K_65", ""),
    ("inr_cross_entropy:acc_misaligned", "misaligned"), ("inr_sh_table_q:out_misaligned", "misaligned"),
    ("inr_nerf_forward_dirs:out_misaligned", "misaligned"),
    ("inr_copy_multi:n_9", "8"), ("inr_adam_step_multi:n_17", "16"), ("inr_finish_rays_mse:N_too_large", "N"),
    ("inr_sh_encode_forward:degree_5", "degree"), ("inr_linear_wgrad:n_in_65", "64"),
    ("inr_sample_training_batch:channels_5", "channels"), ("inr_sample_training_batch:negative_step", "step"),
    ("inr_sample_training_batch:image_too_large", "size"),
    ("inr_set_march_mode:mode_7", "mode"), ("inr_roi_align_3d_set_mode:mode_9", "mode"),
    ("inr_roi_align_3d_forward:zero_bins", "size"), ("inr_roi_align_3d_backward_ws:workspace_too_small", "workspace"),
    ("inr_nerf_forward_table_sliced:12_levels", "16-level"),
])
def test_named_limits_of_the_header(results, key, needle):
    rc, msg = results[key]
    assert rc == EINVAL and msg and needle in msg, (key, rc, msg)


def test_a_launch_that_cannot_succeed_returns_a_code():
    """On a box without a GPU every launch fails: the library reports INR_ELAUNCH with the runtime's message instead of
    aborting (cf. cuda_utils.h:26-35 of the reference's extension, which exit()s).  On a GPU box the launch of this entry
    point with host pointers is asynchronous and may well be accepted; tests/test_gpu_parity.py covers the GPU side."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the launch-failure case of a GPU box is test_gpu_parity.py::test_launch_failure_is_a_return_code")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_negative_child.py")], capture_output=True, text=True, timeout=600)
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    rc, msg = out["inr_nerf_pack_weights_device:launch"]
    assert rc == ELAUNCH and msg, (rc, msg)
