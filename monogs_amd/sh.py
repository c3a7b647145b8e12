"""Real spherical-harmonics helpers for render()'s `convert_SHs_python` branch.

Same functions as /root/reference gaussian_splatting/utils/sh_utils.py:55-126 (eval_sh,
RGB2SH, SH2RGB), written basis-first: the degree-0..3 basis Y_k(d) is evaluated once per
direction and contracted with the coefficients, colour = sum_k sh[..., k] * Y_k(d).
Pinned against the reference's outputs in tests/test_cpu_oracle.py.
"""
from __future__ import annotations

import torch

# band normalisation constants of the real SH basis (l = 0, 1, 2, 3)
_L0 = 0.28209479177387814
_L1 = 0.4886025119029199
_L2 = (1.0925484305920792, 0.31539156525252005, 0.5462742152960396)
_L3 = (0.5900435899266435, 2.890611442640554, 0.4570457994644658, 0.3731763325901154,
       1.445305721320277)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """[..., 3] unit directions -> [..., (deg+1)^2] basis values in the 3DGS ordering."""
    if not 0 <= deg <= 3:
        raise ValueError("SH degree must be 0..3")
    x, y, z = dirs.unbind(dim=-1)
    cols = [torch.full_like(x, _L0)]
    if deg >= 1:
        cols += [-_L1 * y, _L1 * z, -_L1 * x]
    if deg >= 2:
        x2, y2, z2 = x * x, y * y, z * z
        cols += [_L2[0] * x * y, -_L2[0] * y * z, _L2[1] * (2 * z2 - x2 - y2), -_L2[0] * x * z,
                 _L2[2] * (x2 - y2)]
        if deg >= 3:
            cols += [-_L3[0] * y * (3 * x2 - y2), _L3[1] * x * y * z, -_L3[2] * y * (4 * z2 - x2 - y2),
                     _L3[3] * z * (2 * z2 - 3 * x2 - 3 * y2), -_L3[2] * x * (4 * z2 - x2 - y2),
                     _L3[4] * z * (x2 - y2), -_L3[0] * x * (x2 - 3 * y2)]
    return torch.stack(cols, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh [..., C, K], dirs [..., 3] (unit) -> [..., C]."""
    Y = sh_basis(deg, dirs)
    k = Y.shape[-1]
    if sh.shape[-1] < k:
        raise ValueError("not enough SH coefficients for the requested degree")
    return (sh[..., :k] * Y.unsqueeze(-2)).sum(dim=-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / _L0


def SH2RGB(sh):
    return sh * _L0 + 0.5
