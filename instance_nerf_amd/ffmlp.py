"""``ffmlp.FFMLP`` with upstream's constructor (SURVEY.md Appendix A.2, row 8 of section 2): torch-ngp's optional
"fully fused" bias-free ReLU MLP (``--ff``), one flat ``weights`` parameter holding the layers back to back
(first layer [hidden, input], hidden layers [hidden, hidden], last layer [output, hidden], row-major).

The render / train hot path does not go through this module - ``NeRFNetwork`` runs its MLPs inside the fused field
kernels (csrc/field_fused.hip) - it exists so that code written against upstream's ``FFMLP`` keeps working; layers run
as ``HipLinear`` (rocBLAS forward, the split-K MFMA kernel for weight gradients)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class FFMLP(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim, num_layers, activation="relu"):
        super().__init__()
        if activation != "relu":
            raise NotImplementedError("FFMLP (HIP): relu only")
        if num_layers < 2:
            raise ValueError("FFMLP needs at least two layers")
        self.input_dim, self.output_dim, self.hidden_dim, self.num_layers = input_dim, output_dim, hidden_dim, num_layers
        self.shapes = [(hidden_dim, input_dim)] + [(hidden_dim, hidden_dim)] * (num_layers - 2) + [(output_dim, hidden_dim)]
        self.num_parameters = sum(o * i for o, i in self.shapes)
        self.weights = nn.Parameter(torch.zeros(self.num_parameters))
        self.reset_parameters()

    def reset_parameters(self):
        std = math.sqrt(3 / self.hidden_dim)            # upstream's uniform(-std, std)
        self.weights.data.uniform_(-std, std)

    def layer_weights(self):
        off, out = 0, []
        for o, i in self.shapes:
            out.append(self.weights[off:off + o * i].view(o, i))
            off += o * i
        return out

    def forward(self, inputs):
        from .nerf.network import _LinearFn
        prefix = inputs.shape[:-1]
        h = inputs.reshape(-1, self.input_dim).float()
        ws = self.layer_weights()
        for l, w in enumerate(ws):
            use_hip = h.is_cuda and w.shape[0] <= 64 and w.shape[1] <= 64 and torch.is_grad_enabled() and w.requires_grad
            h = _LinearFn.apply(h, w) if use_hip else F.linear(h, w)
            if l != len(ws) - 1:
                h = F.relu(h, inplace=True)
        return h.view(*prefix, self.output_dim)

    def __repr__(self):
        return (f"FFMLP: input_dim={self.input_dim} output_dim={self.output_dim} hidden_dim={self.hidden_dim} "
                f"num_layers={self.num_layers} activation=relu")
