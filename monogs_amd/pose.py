"""Camera-pose helpers on the host side of the hot path (row a11 of SURVEY §8a).

Mirrors /root/reference utils/pose_utils.py:13-98: tau = [rho(3); theta(3)],
left perturbation T_new = Exp(tau) @ T_w2c, deltas zeroed after the update.  The
reference calls lietorch.SE3.exp (pose_utils.py:92); lietorch is not installed here, so
the closed form of its own pure-torch SE3_exp (pose_utils.py:26-74) is restated.
"""
from __future__ import annotations

import torch


def skew_sym_mat(x: torch.Tensor) -> torch.Tensor:
    ssm = torch.zeros(3, 3, device=x.device, dtype=x.dtype)
    ssm[0, 1], ssm[0, 2] = -x[2], x[1]
    ssm[1, 0], ssm[1, 2] = x[2], -x[0]
    ssm[2, 0], ssm[2, 1] = -x[1], x[0]
    return ssm


def SO3_exp(theta: torch.Tensor) -> torch.Tensor:
    W = skew_sym_mat(theta)
    W2 = W @ W
    angle = torch.linalg.norm(theta)
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    if angle < 1e-5:
        return I + W + 0.5 * W2
    return I + (torch.sin(angle) / angle) * W + ((1 - torch.cos(angle)) / angle ** 2) * W2


def V(theta: torch.Tensor) -> torch.Tensor:
    I = torch.eye(3, device=theta.device, dtype=theta.dtype)
    W = skew_sym_mat(theta)
    W2 = W @ W
    angle = torch.linalg.norm(theta)
    if angle < 1e-5:
        return I + 0.5 * W + (1.0 / 6.0) * W2
    return (I + W * ((1.0 - torch.cos(angle)) / angle ** 2)
            + W2 * ((angle - torch.sin(angle)) / angle ** 3))


def SE3_exp(tau: torch.Tensor) -> torch.Tensor:
    rho, theta = tau[:3], tau[3:]
    T = torch.eye(4, device=tau.device, dtype=tau.dtype)
    T[:3, :3] = SO3_exp(theta)
    T[:3, 3] = V(theta) @ rho
    return T


def update_pose(camera, converged_threshold: float = 1e-4) -> bool:
    """camera.T <- Exp([cam_trans_delta; cam_rot_delta]) @ camera.T, zero the deltas,
    report convergence (pose_utils.py:88-98)."""
    with torch.no_grad():
        tau = torch.cat([camera.cam_trans_delta, camera.cam_rot_delta], dim=0)
        camera.T = SE3_exp(tau) @ camera.T
        converged = bool((tau ** 2).sum() < converged_threshold ** 2)
        camera.cam_rot_delta.data.fill_(0)
        camera.cam_trans_delta.data.fill_(0)
    return converged
