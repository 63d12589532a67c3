"""Host-side mirror of torch-ngp's ``shencoder`` module over the C ABI."""
import torch
import torch.nn as nn

from . import _lib
from ._lib import check, ptr, stream_ptr


class _SHEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, degree):
        lib = _lib.load()
        inputs = inputs.contiguous().float()
        M = inputs.shape[0]
        out = torch.empty(M, degree * degree, dtype=torch.float32, device=inputs.device)
        check(lib.inr_sh_encode_forward(ptr(inputs, torch.float32, "inputs"), M, degree, ptr(out), stream_ptr()),
              "sh_encode_forward")
        ctx.save_for_backward(inputs)
        ctx.degree = degree
        return out

    @staticmethod
    def backward(ctx, grad):
        lib = _lib.load()
        (inputs,) = ctx.saved_tensors
        grad = grad.contiguous().float()
        g = torch.empty_like(inputs)
        check(lib.inr_sh_encode_backward(ptr(grad), ptr(inputs), inputs.shape[0], ctx.degree, ptr(g), stream_ptr()),
              "sh_encode_backward")
        return g, None


class SHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        if input_dim != 3 or not 1 <= degree <= 4:
            raise RuntimeError("SHEncoder (HIP): input_dim must be 3 and degree in 1..4")
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2

    def __repr__(self):
        return f"SHEncoder: input_dim={self.input_dim} degree={self.degree}"

    def forward(self, inputs, size=1):
        prefix = list(inputs.shape[:-1])
        out = _SHEncode.apply((inputs / size).reshape(-1, 3), self.degree)
        return out.view(prefix + [self.output_dim])
