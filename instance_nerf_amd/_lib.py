"""ctypes binding of libinr_hip.so (the C ABI in include/inr.h).

There is NO fallback: if the library is missing or a call is made without a
GPU tensor the caller gets a RuntimeError.  torch is imported first so that the
library binds to the HIP runtime already loaded by PyTorch-ROCm (same
``libamdhip64.so.7`` soname), which makes torch's streams and allocations
directly usable by the kernels.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_uint32, c_void_p

import torch  # noqa: F401  (must precede CDLL: shares the HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("INR_LIB_PATH") or os.path.join(_HERE, "csrc", "libinr_hip.so")   # env override: profiling builds only
MAX_LEVELS = 16
GRID_FX_STATE_FLOATS = 4192      # include/inr.h INR_GRID_FX_STATE_FLOATS


class GridDesc(Structure):
    _fields_ = [("num_levels", c_int32), ("level_dim", c_int32),
                ("offsets", c_uint32 * (MAX_LEVELS + 1)), ("scales", c_float * MAX_LEVELS),
                ("resolutions", c_uint32 * MAX_LEVELS), ("hashed", c_uint32 * MAX_LEVELS)]


P = c_void_p
_SIGS = {
    "inr_abi_version": (c_int32, []),
    "inr_last_error": (c_char_p, []),
    "inr_set_overlap_placement": (c_int32, [c_int32]),
    "inr_set_march_mode": (c_int32, [c_int32]),
    "inr_device_info": (c_int32, [c_int32, POINTER(c_int64)]),
    "inr_get_rays": (c_int32, [P, c_int64, c_float, c_float, c_float, c_float, c_int32, P, c_int64, P, P, P]),
    "inr_sample_training_batch": (c_int32, [P, c_float, c_float, c_float, c_float, c_int32, c_int32, P, c_int32, P, c_int32,
                                            c_int64, c_int64, c_int64, P, P, P, P, P, P]),
    "inr_near_far_from_aabb": (c_int32, [P, P, P, c_int64, c_float, P, P, P]),
    "inr_near_far_from_aabb_skip": (c_int32, [P, P, P, c_int64, c_float, P, c_int64, P, P, P]),
    "inr_morton3D": (c_int32, [P, c_int64, P, P]),
    "inr_morton3D_invert": (c_int32, [P, c_int64, P, P]),
    "inr_packbits": (c_int32, [P, c_int64, c_float, P, P]),
    "inr_mark_untrained_grid": (c_int32, [P, c_int32, c_float, c_float, c_float, c_float, c_int32, c_int32, c_float, P, P]),
    "inr_occ_cell_positions": (c_int32, [P, P, c_int64, c_int32, c_float, P, P]),
    "inr_occ_update": (c_int32, [P, P, P, c_int64, c_int64, c_float, c_float, P, P, P]),
    "inr_packbits_mean": (c_int32, [P, c_int64, P, c_float, P, P, P, c_int32, c_int32, P, P]),
    "inr_occ_sample_workspace_bytes": (c_int64, [c_int64]),
    "inr_occ_sample_cells": (c_int32, [P, c_int64, P, c_int64, P, P, P]),
    "inr_march_write_fills_unowned_rows": (c_int32, [c_int64, c_int32, c_int32]),
    "inr_march_workspace_bytes": (c_int64, [c_int64, c_int32]),
    "inr_march_rays_train_count": (c_int32, [P, P, P, c_float, c_float, c_int32, c_int64, c_int32, c_int32,
                                             P, P, P, P, P, P, c_int32, P]),
    "inr_march_rays_train_write": (c_int32, [P, P, P, c_float, c_float, c_int32, c_int64, c_int32, c_int32,
                                             c_int64, P, P, P, P, P, P, P, P, c_int32, P]),
    "inr_march_rays_patch_write": (c_int32, [P, P, P, c_float, c_float, c_int32, c_int64, c_int32, c_int32,
                                             c_int64, P, P, P, P, P, P, P, P, c_int32, P, c_int32, P]),
    "inr_nerf_bwd_packed_floats": (c_int64, []),
    "inr_nerf_pack_weights_device": (c_int32, [P, P, P, P, P, P, P, P]),
    "inr_nerf_forward_train": (c_int32, [P, P, c_int64, c_float, P, POINTER(GridDesc), P, P, P, P, P, P, P, P, P, P]),
    "inr_nerf_backward": (c_int32, [P, P, P, P, P, P, P, c_int64, c_float, P, P, P, P, P, P, P, P]),
    "inr_instance_bwd_packed_floats": (c_int64, []),
    "inr_instance_pack_weights_device": (c_int32, [P, P, P, c_int32, P, P, P]),
    "inr_instance_forward_train": (c_int32, [P, c_int64, c_float, P, POINTER(GridDesc), P, c_int32, P, P, P, P, P]),
    "inr_instance_backward": (c_int32, [P, c_int32, P, P, c_int64, P, P, P, P, P]),
    "inr_nerf_forward_enc": (c_int32, [P, P, c_int64, c_float, P, POINTER(GridDesc), P, P, P, P, P]),
    "inr_nerf_head_backward": (c_int32, [P, P, P, P, c_int64, c_float, P, P, P, P, P, P, P, P, P, P, c_int64, P]),
    "inr_instance_forward_enc": (c_int32, [P, c_int64, P, c_float, P, POINTER(GridDesc), P, c_int32, P, P, P]),
    "inr_instance_head_workspace_bytes": (c_int64, []),
    "inr_instance_head_backward": (c_int32, [P, P, P, P, c_int32, c_int64, c_int64, P, P, P, P, P, P, P, P, P, P, P, c_int64, P]),
    "inr_adam_step_multi": (c_int32, [c_int32, P, P, P, P, P, P, c_float, c_float, c_float, c_int32, c_float, P]),
    "inr_adam_ema_step_multi": (c_int32, [c_int32, P, P, P, P, P, P, c_float, c_float, c_float, c_int32, c_float, P,
                                          c_float, P]),
    "inr_adam_set_hyper": (c_int32, [P, c_int32, c_float, c_float, c_float, c_int32, P, P]),
    "inr_adam_step_multi_dev": (c_int32, [c_int32, P, P, P, P, P, P, c_float, c_float, c_float, P]),
    "inr_copy_multi": (c_int32, [c_int32, P, P, P, P]),
    "inr_adam_set_hyper_ema": (c_int32, [P, c_int32, c_float, c_float, c_float, c_int32, c_float, P, P]),
    "inr_adam_ema_step_multi_dev": (c_int32, [c_int32, P, P, P, P, P, P, c_float, c_float, c_float, P, P]),
    "inr_cross_entropy": (c_int32, [P, P, c_int64, c_int32, c_int64, P, P, P, P]),
    "inr_finish_rays": (c_int32, [P, P, P, P, P, P, c_float, c_float, c_float, c_int64, P, P, P]),
    "inr_finish_rays_mse": (c_int32, [P, P, P, P, P, c_float, c_float, c_float, P, P, c_int64, P, P, P, P, P]),
    "inr_sh_table_q": (c_int32, [P, c_int64, P, P]),
    "inr_nerf_forward_table": (c_int32, [P, P, P, c_int64, c_float, P, POINTER(GridDesc), P, c_float, P, P, P]),
    "inr_nerf_forward_table_half": (c_int32, [P, P, P, c_int64, c_float, P, POINTER(GridDesc), P, c_float, P, P, P]),
    "inr_nerf_forward_table_fast": (c_int32, [P, P, P, c_int64, c_float, P, c_int32, POINTER(GridDesc), P, c_float, P, P, P]),
    "inr_nerf_forward_table_sliced_workspace_bytes": (c_int64, [c_int64]),
    "inr_nerf_forward_table_sliced": (c_int32, [P, P, P, c_int64, c_float, P, POINTER(GridDesc), P, c_float, P, P, P, P]),
    "inr_nerf_pack_weights_f16": (c_int32, [P, P, P, P, P, P]),
    "inr_composite_rays_patch_forward": (c_int32, [P, P, P, P, c_int64, c_int64, c_float, P, c_int32, P, P, P, P, P, P, P]),
    "inr_project_masks_patch": (c_int32, [P, P, P, c_int64, c_int64, P, c_int32, c_int32, c_int32, P, c_int32, c_int32, c_int32,
                                          P, P]),
    "inr_march_rays": (c_int32, [c_int64, c_int32, P, P, P, P, c_float, c_float, c_int32, c_int32, c_int32,
                                 P, P, P, P, P, P, P]),
    "inr_composite_rays": (c_int32, [c_int64, c_int32, P, P, P, P, P, P, P, P, c_float, P, P, c_int32, P]),
    "inr_compact_alive": (c_int32, [P, c_int64, P, P, P]),
    "inr_composite_rays_extra_forward": (c_int32, [P, P, P, c_int64, c_int64, c_int32, P, P, c_int32, c_int64, P, P, P, P]),
    "inr_composite_rays_train_forward": (c_int32, [P, P, P, P, c_int64, c_int64, c_float, P, c_int32, P, P, P, P, P, P, P]),
    "inr_composite_rays_train_backward": (c_int32, [P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int64, c_float, c_int32,
                                                    P, P, P, P, P]),
    "inr_grid_encode_forward": (c_int32, [P, P, POINTER(GridDesc), c_int64, c_float, P, P]),
    "inr_grid_encode_backward": (c_int32, [P, P, POINTER(GridDesc), c_int64, c_float, P, P]),
    "inr_grid_encode_backward_ordered": (c_int32, [P, P, P, POINTER(GridDesc), c_int64, c_float, P, P]),
    "inr_grid_encode_backward_input": (c_int32, [P, P, P, POINTER(GridDesc), c_int64, c_float, P, P]),
    "inr_grid_encode_backward_levels": (c_int32, [P, P, P, POINTER(GridDesc), c_int64, c_float, P, c_int32, c_int32, P]),
    "inr_grid_encode_backward_levels_fx": (c_int32, [P, P, P, POINTER(GridDesc), c_int64, c_float, P, c_int32, c_int32, P, P]),
    "inr_grid_grad_finish_fx": (c_int32, [P, POINTER(GridDesc), c_int32, c_int32, P, P]),
    "inr_grid_fx_update": (c_int32, [P, c_int32, c_float, c_int32, P]),
    "inr_grid_encode_backward_levels_fx64": (c_int32, [P, P, P, POINTER(GridDesc), c_int64, c_float, P, P, c_int32, c_int32, P, P]),
    "inr_grid_grad_finish_fx64": (c_int32, [P, P, POINTER(GridDesc), c_int32, c_int32, P, P]),
    "inr_sh_encode_forward": (c_int32, [P, c_int64, c_int32, P, P]),
    "inr_sh_encode_backward": (c_int32, [P, P, c_int64, c_int32, P, P]),
    "inr_nerf_packed_floats": (c_int64, []),
    "inr_nerf_pack_weights": (c_int32, [P, P, P, P, P, P]),
    "inr_instance_packed_floats": (c_int64, [c_int32]),
    "inr_instance_pack_weights": (c_int32, [P, P, P, c_int32, P]),
    "inr_nerf_forward": (c_int32, [P, P, c_int64, P, c_float, P, POINTER(GridDesc), P, c_float, P, P, P, P]),
    "inr_nerf_forward_fast": (c_int32, [P, P, c_int64, P, c_float, P, POINTER(GridDesc), P, c_float, P, P, P]),
    "inr_nerf_forward_dirs": (c_int32, [P, c_int64, c_float, P, POINTER(GridDesc), P, P, c_int32, P, P]),
    "inr_nerf_forward_lattice": (c_int32, [P, P, P, c_int32, c_int32, c_int32, c_float, P, POINTER(GridDesc), P, P, c_int32,
                                           c_float, P, P]),
    "inr_instance_forward": (c_int32, [P, c_int64, P, c_float, P, POINTER(GridDesc), P, c_int32, P, P]),
    "inr_roi_align_3d_set_mode": (c_int32, [c_int32]),
    "inr_roi_align_3d_forward": (c_int32, [P, P, P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32,
                                           c_int32, c_int32, c_float, P, P]),
    "inr_roi_align_3d_backward": (c_int32, [P, P, P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32,
                                            c_int32, c_int32, c_float, P, P]),
    "inr_roi_align_3d_backward_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32,
                                                            c_int32, c_int32]),
    "inr_roi_align_3d_backward_prefers_workspace": (c_int32, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32,
                                                              c_int32, c_int32, c_int64]),
    "inr_roi_align_3d_backward_ws": (c_int32, [P, P, P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int64, c_int32,
                                               c_int32, c_int32, c_float, P, P, c_int64, P]),
    "inr_nerf_render": (c_int32, [P, P, P, P, c_int64, c_int64, c_float, P, POINTER(GridDesc), P, c_float, c_float,
                                  P, P, P, P, P, c_int32, P]),
    "inr_nerf_render_fast": (c_int32, [P, P, P, P, c_int64, c_int64, c_float, P, POINTER(GridDesc), P, c_float, c_float,
                                  P, P, P, P, P, c_int32, P]),
    "inr_instance_render": (c_int32, [P, P, P, c_int64, c_int64, c_float, P, POINTER(GridDesc), P, c_int32, P, c_int32, P, P]),
    "inr_instance_render_fast": (c_int32, [P, P, P, c_int64, c_int64, c_float, P, POINTER(GridDesc), P, c_int32, P, c_int32, P, P]),
    "inr_instance_pack_weights_f16": (c_int32, [P, P, P, c_int32, P]),
    "inr_linear_wgrad_workspace_bytes": (c_int64, []),
    "inr_linear_wgrad": (c_int32, [P, P, c_int64, c_int32, c_int32, P, P, P]),
    "inr_adam_step": (c_int32, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_int32, c_float, P]),
}
EXPORTS = tuple(_SIGS)

ABI_VERSION = 9          # include/inr.h INR_ABI_VERSION this binding was written against
_lib = None


def load():
    """Loads the library (once).  Raises RuntimeError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m instance_nerf_amd.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)         # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.inr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libinr_hip.so ABI version {lib.inr_abi_version()} != {ABI_VERSION} (include/inr.h "
                           "INR_ABI_VERSION): rebuild with `python -m instance_nerf_amd.build --force`")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().inr_last_error()
        raise RuntimeError(f"libinr_hip {what} failed (code {rc}): {msg.decode() if msg else ''}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr():
    """The current HIP stream of the current device (every launch of the library goes on it).  The raw accessors cost
    ~0.3 us; ``torch.cuda.current_stream().cuda_stream`` ~10 us - a dozen calls per training step."""
    if _raw_stream is not None and _cur_device is not None:
        return c_void_p(_raw_stream(_cur_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None, name="tensor", allow_none=False):
    """Device pointer of a contiguous GPU tensor, with the TORCH_CHECK-style
    argument validation the reference's extension does on the C++ side
    (/root/reference/nerf_rcnn/model/rotated_iou/cuda_op/utils.h:6-31)."""
    if t is None:
        if allow_none:
            return None
        raise RuntimeError(f"{name} must not be None")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return c_void_p(t.data_ptr())


def host_ptr(t, dtype, name="tensor"):
    if t.is_cuda or not t.is_contiguous() or t.dtype != dtype:
        raise RuntimeError(f"{name} must be a contiguous CPU {dtype} tensor")
    return c_void_p(t.data_ptr())


def make_grid_desc(table):
    """table: dict from gridencoder.level_table()."""
    d = GridDesc()
    L = int(table["num_levels"])
    d.num_levels = L
    d.level_dim = int(table["level_dim"])
    for i in range(L + 1):
        d.offsets[i] = int(table["offsets"][i])
    for i in range(L):
        d.scales[i] = float(table["scales"][i])
        d.resolutions[i] = int(table["resolutions"][i])
        d.hashed[i] = int(table["hashed"][i])
    return d
