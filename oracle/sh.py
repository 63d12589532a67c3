"""Degree-4 real spherical harmonics (oracle; test infrastructure only).

Follows SURVEY.md Appendix A.1 "SH degree 4" (row a10; upstream
``shencoder.SHEncoder`` of the un-vendored submodule pinned at
/root/reference/README.md:27,59).  Parity unpinned.
"""
import torch


def sh_encode(d, degree=4):
    """d f32[M,3] (unit) -> f32[M, degree^2]; torch, differentiable."""
    d = torch.as_tensor(d, dtype=torch.float32)
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    out = [torch.full_like(x, 0.28209479177387814)]
    if degree > 1:
        out += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
    if degree > 2:
        out += [1.0925484305920792 * xy, -1.0925484305920792 * yz,
                0.94617469575755997 * z2 - 0.31539156525251999,
                -1.0925484305920792 * xz, 0.54627421529603959 * (x2 - y2)]
    if degree > 3:
        out += [0.59004358992664352 * y * (-3.0 * x2 + y2),
                2.8906114426405538 * xy * z,
                0.45704579946446572 * y * (1.0 - 5.0 * z2),
                0.3731763325901154 * z * (5.0 * z2 - 3.0),
                0.45704579946446572 * x * (1.0 - 5.0 * z2),
                1.4453057213202769 * z * (x2 - y2),
                0.59004358992664352 * x * (-x2 + 3.0 * y2)]
    return torch.stack(out, -1)
