"""Camera-pose helpers on the host side of the hot path (row a11 of SURVEY §8a).

Contract of /root/reference utils/pose_utils.py:26-98: tau = [rho(3); theta(3)], left
perturbation T_new = Exp(tau) @ T_w2c, deltas zeroed after the update.  The reference calls
lietorch.SE3.exp (pose_utils.py:92); lietorch is not installed here, so the closed form is
written out: with W = [theta]x and phi = |theta|,
    R = I + a W + b W^2,   V = I + b W + c W^2,   t = V rho,
    a = sin(phi)/phi, b = (1 - cos(phi))/phi^2, c = (phi - sin(phi))/phi^3
(series values for phi < 1e-5, the reference's threshold).  Pinned against
torch.linalg.matrix_exp in tests/test_cpu_oracle.py.
"""
from __future__ import annotations

import torch


def skew_sym_mat(x: torch.Tensor) -> torch.Tensor:
    z = torch.zeros((), device=x.device, dtype=x.dtype)
    return torch.stack([torch.stack([z, -x[2], x[1]]), torch.stack([x[2], z, -x[0]]),
                        torch.stack([-x[1], x[0], z])])


def _rodrigues_coefficients(phi: torch.Tensor):
    if float(phi) < 1e-5:
        one = torch.ones((), device=phi.device, dtype=phi.dtype)
        return one, 0.5 * one, one / 6.0
    p2 = phi * phi
    return torch.sin(phi) / phi, (1.0 - torch.cos(phi)) / p2, (phi - torch.sin(phi)) / (p2 * phi)


def _so3_pair(theta: torch.Tensor):
    """(R, V) of the SE(3) exponential for rotation vector theta."""
    W = skew_sym_mat(theta)
    W2 = W @ W
    a, b, c = _rodrigues_coefficients(torch.linalg.norm(theta))
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    return I + a * W + b * W2, I + b * W + c * W2


def SO3_exp(theta: torch.Tensor) -> torch.Tensor:
    return _so3_pair(theta)[0]


def V(theta: torch.Tensor) -> torch.Tensor:
    return _so3_pair(theta)[1]


def SE3_exp(tau: torch.Tensor) -> torch.Tensor:
    R, Vm = _so3_pair(tau[3:])
    T = torch.eye(4, device=tau.device, dtype=tau.dtype)
    T[:3, :3] = R
    T[:3, 3] = Vm @ tau[:3]
    return T


def update_pose(camera, converged_threshold: float = 1e-4) -> bool:
    """camera.T <- Exp([cam_trans_delta; cam_rot_delta]) @ camera.T, zero the deltas, report
    whether |tau| fell below the threshold (pose_utils.py:88-98)."""
    with torch.no_grad():
        tau = torch.cat([camera.cam_trans_delta, camera.cam_rot_delta])
        camera.T.copy_(SE3_exp(tau) @ camera.T)     # in place: native code may hold this buffer's address
        small = bool(torch.dot(tau, tau) < converged_threshold * converged_threshold)
        camera.cam_trans_delta.zero_()
        camera.cam_rot_delta.zero_()
    return small
