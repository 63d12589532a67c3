"""MI355X-native instance-field NeRF render/train hot path.

The package mirrors the extension/module names of torch-ngp, the reference's
un-vendored hot-path submodule, over a C ABI
(include/inr.h) implemented with hand-written HIP kernels for gfx950.
"""
__all__ = ["raymarching", "gridencoder", "shencoder", "activation", "encoding", "nerf", "scene"]
