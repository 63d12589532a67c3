"""Child process of tests/test_abi_negative.py: calls EVERY export of include/inr.h with bad arguments and prints one
JSON object {"<name>:<variant>": [return code, message]}.  Runs on a CPU-only box: the library loads without a GPU, and
a call that validates its arguments never reaches a launch.  A crash (abort, segfault, exit()) ends this process without
the final line, which is what the parent asserts on."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instance_nerf_amd import _lib  # noqa: E402

lib = _lib.load()
HOST = ctypes.create_string_buffer(1 << 16)              # 64 KB of zeroed, readable, writable host memory
ADDR = (ctypes.addressof(HOST) + 255) // 256 * 256


def is_pointer(t):
    return t is _lib.P or t is ctypes.c_char_p or (isinstance(t, type) and issubclass(t, ctypes._Pointer))


def pointer(t, addr):
    return ctypes.c_void_p(addr) if t is _lib.P else ctypes.cast(addr, t)


def call(name, args):
    rc = int(getattr(lib, name)(*args))
    msg = lib.inr_last_error()
    return [rc, msg.decode() if msg else ""]


def good_desc(levels=16, level_dim=2):
    d = _lib.GridDesc()
    d.num_levels, d.level_dim = levels, level_dim
    off = 0
    for l in range(min(levels, _lib.MAX_LEVELS)):
        d.offsets[l] = off
        d.scales[l] = float(16 * 2 ** l - 1)
        d.resolutions[l] = 16 * 2 ** l
        d.hashed[l] = 1
        off += 4096
    d.offsets[min(levels, _lib.MAX_LEVELS)] = off
    return d


out = {}
# ---- every export, two generic variants: null pointers with plausible sizes; valid (host) pointers with negative sizes
for name, (restype, argtypes) in _lib._SIGS.items():
    if name in ("inr_abi_version", "inr_last_error"):
        continue
    for variant in ("null", "negative"):
        args = []
        for t in argtypes:
            if is_pointer(t):
                args.append(None if variant == "null" else pointer(t, ADDR))
            elif t is ctypes.c_float:
                args.append(1.0)
            else:
                args.append(16 if variant == "null" else -1)
        out[f"{name}:{variant}"] = call(name, args)


# ---- targeted cases: everything valid except the one thing named
def args_for(name, **over):
    """All pointers -> the host buffer, sizes 16, floats 1; `over` replaces arguments by position."""
    _, argtypes = _lib._SIGS[name]
    args = [pointer(t, ADDR) if is_pointer(t) else (1.0 if t is ctypes.c_float else 16) for t in argtypes]
    for k, v in over.items():
        args[int(k[1:])] = v
    return args


desc17 = good_desc(levels=17)
desc_f4 = good_desc(level_dim=4)
ok_desc = good_desc()
for name in ("inr_grid_encode_forward", "inr_grid_encode_backward"):
    out[f"{name}:num_levels_17"] = call(name, args_for(name, a2=ctypes.byref(desc17)))
    out[f"{name}:level_dim_4"] = call(name, args_for(name, a2=ctypes.byref(desc_f4)))
out["inr_nerf_forward:num_levels_17"] = call("inr_nerf_forward", args_for("inr_nerf_forward", a6=ctypes.byref(desc17)))
out["inr_instance_forward:num_levels_17"] = call("inr_instance_forward", args_for("inr_instance_forward", a5=ctypes.byref(desc17)))
# K > 64 / K not a multiple of 16 wherever an entry point takes the instance head's K
out["inr_instance_forward:K_65"] = call("inr_instance_forward", args_for("inr_instance_forward", a5=ctypes.byref(ok_desc), a7=65))
out["inr_instance_forward:K_80"] = call("inr_instance_forward", args_for("inr_instance_forward", a5=ctypes.byref(ok_desc), a7=80))
out["inr_instance_forward_enc:K_80"] = call("inr_instance_forward_enc", args_for("inr_instance_forward_enc", a5=ctypes.byref(ok_desc), a7=80))
out["inr_instance_pack_weights:K_65"] = call("inr_instance_pack_weights", args_for("inr_instance_pack_weights", a3=65))
out["inr_instance_pack_weights_device:K_80"] = call("inr_instance_pack_weights_device", args_for("inr_instance_pack_weights_device", a3=80))
out["inr_instance_head_backward:K_80"] = call("inr_instance_head_backward", args_for("inr_instance_head_backward", a4=80))
out["inr_instance_render:K_80"] = call("inr_instance_render", args_for("inr_instance_render", a7=ctypes.byref(ok_desc), a9=80))
out["inr_cross_entropy:K_65"] = call("inr_cross_entropy", args_for("inr_cross_entropy", a3=65))
out["inr_composite_rays_extra_forward:K_65"] = call("inr_composite_rays_extra_forward", args_for("inr_composite_rays_extra_forward", a5=65))
out["inr_instance_packed_floats:K_65"] = [int(lib.inr_instance_packed_floats(65)), (lib.inr_last_error() or b"").decode()]
# misaligned buffers where the header asks for an alignment
out["inr_cross_entropy:acc_misaligned"] = call("inr_cross_entropy", args_for("inr_cross_entropy", a6=ctypes.c_void_p(ADDR + 4)))
out["inr_sh_table_q:out_misaligned"] = call("inr_sh_table_q", args_for("inr_sh_table_q", a2=ctypes.c_void_p(ADDR + 4)))
out["inr_nerf_forward_dirs:out_misaligned"] = call("inr_nerf_forward_dirs", args_for(
    "inr_nerf_forward_dirs", a4=ctypes.byref(ok_desc), a7=4, a8=ctypes.c_void_p(ADDR + 4)))
out["inr_copy_multi:n_9"] = call("inr_copy_multi", args_for("inr_copy_multi", a0=9))
out["inr_adam_step_multi:n_17"] = call("inr_adam_step_multi", args_for("inr_adam_step_multi", a0=17))
out["inr_finish_rays_mse:N_too_large"] = call("inr_finish_rays_mse", args_for("inr_finish_rays_mse", a10=65537))
out["inr_sh_encode_forward:degree_5"] = call("inr_sh_encode_forward", args_for("inr_sh_encode_forward", a2=5))
out["inr_linear_wgrad:n_in_65"] = call("inr_linear_wgrad", args_for("inr_linear_wgrad", a3=65))
out["inr_sample_training_batch:channels_5"] = call("inr_sample_training_batch", args_for("inr_sample_training_batch", a8=5))
out["inr_sample_training_batch:negative_step"] = call("inr_sample_training_batch", args_for("inr_sample_training_batch", a12=-1))
out["inr_sample_training_batch:image_too_large"] = call("inr_sample_training_batch", args_for(
    "inr_sample_training_batch", a5=65536, a6=65536))
out["inr_set_march_mode:mode_7"] = call("inr_set_march_mode", [7])
out["inr_roi_align_3d_set_mode:mode_9"] = call("inr_roi_align_3d_set_mode", [9])
out["inr_roi_align_3d_forward:zero_bins"] = call("inr_roi_align_3d_forward", args_for("inr_roi_align_3d_forward", a9=0))
out["inr_roi_align_3d_backward_ws:workspace_too_small"] = call("inr_roi_align_3d_backward_ws", args_for("inr_roi_align_3d_backward_ws", a15=64))
out["inr_nerf_forward_table_sliced:12_levels"] = call("inr_nerf_forward_table_sliced", args_for(
    "inr_nerf_forward_table_sliced", a6=ctypes.byref(good_desc(levels=12))))
# ---- a launch that cannot succeed (this box has no GPU; on a GPU box the parent skips this key): INR_ELAUNCH, not abort
out["inr_nerf_pack_weights_device:launch"] = call("inr_nerf_pack_weights_device", args_for("inr_nerf_pack_weights_device"))
out["alive"] = [0, "reached the end"]
sys.stdout.write(json.dumps(out) + "\n")
