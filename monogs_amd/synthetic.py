"""Deterministic synthetic scenes for parity tests and bench.py (BASELINE.md §4,
SURVEY.md §8d).  Generated on the CPU with a seeded torch.Generator, then moved
to the device by the caller.

Camera: TUM fr3_office intrinsics (configs/mono/tum/fr3_office.yaml:6-16 in the
reference tree), scaled with the image size; T_w2c = I; znear 0.01, zfar 100.
Matrices follow utils/camera_utils.py:94-104; the projection is the closed pinhole form that
graphics_utils.py:56-77 evaluates to (checked against reference-generated goldens).
"""
from __future__ import annotations

import math
from typing import NamedTuple

import torch

SH_C0 = 0.28209479177387814


def projection_matrix2(znear, zfar, cx, cy, fx, fy, W, H):
    """Pinhole projection used with `projmatrix_raw` (what getProjectionMatrix2,
    graphics_utils.py:56-77, evaluates to once its frustum bounds are substituted): NDC x = (2 fx X/Z
    + (2 cx - W)) / W, likewise y; depth row maps [znear, zfar] to [0, 1]; w = Z."""
    f, n = float(zfar), float(znear)
    return torch.tensor([[2.0 * fx / W, 0.0, (2.0 * cx - W) / W, 0.0],
                         [0.0, 2.0 * fy / H, (2.0 * cy - H) / H, 0.0],
                         [0.0, 0.0, f / (f - n), -f * n / (f - n)],
                         [0.0, 0.0, 1.0, 0.0]], dtype=torch.float32)


class Camera(NamedTuple):
    H: int
    W: int
    fx: float
    fy: float
    cx: float
    cy: float
    tanfovx: float
    tanfovy: float
    viewmatrix: torch.Tensor       # T^T           (camera_utils.py:94-96)
    projmatrix: torch.Tensor       # V @ P^T       (camera_utils.py:98-104)
    projmatrix_raw: torch.Tensor   # P^T           (slam_frontend.py:1815-1825)


# Replica calibration (configs/rgbd/replica/base_config.yaml:17-30): 1200x680
REPLICA_INTRINSICS = (600.0, 600.0, 599.5, 339.5)


def make_camera(W=640, H=480, T_w2c: torch.Tensor | None = None, intrinsics=None) -> Camera:
    """`intrinsics` = (fx, fy, cx, cy) in pixels of a W x H image; default: fr3_office's,
    scaled with the image width."""
    if intrinsics is None:
        s = W / 640.0
        fx, fy, cx, cy = 535.4 * s, 539.2 * s, 320.1 * s, 247.6 * s
    else:
        fx, fy, cx, cy = (float(v) for v in intrinsics)
    fovx = 2 * math.atan(W / (2 * fx))
    fovy = 2 * math.atan(H / (2 * fy))
    P = projection_matrix2(0.01, 100.0, cx, cy, fx, fy, W, H).t().contiguous()
    T = torch.eye(4) if T_w2c is None else T_w2c.float()
    V = T.t().contiguous()
    return Camera(H, W, fx, fy, cx, cy, math.tan(fovx / 2), math.tan(fovy / 2),
                  V, (V @ P).contiguous(), P)


class Scene(NamedTuple):
    cam: Camera
    means3D: torch.Tensor    # [N,3]
    log_scales: torch.Tensor  # [N,3]  (activation exp,     gaussian_model.py:54)
    rot: torch.Tensor        # [N,4]  (activation normalise, gaussian_model.py:62)
    opacity_logit: torch.Tensor  # [N,1] (activation sigmoid, gaussian_model.py:59)
    features_dc: torch.Tensor    # [N,1,3] SH degree-0 coefficients
    gt_image: torch.Tensor   # [3,H,W]
    gt_depth: torch.Tensor   # [1,H,W]
    bg: torch.Tensor         # [3]


def make_scene(N: int, W: int = 640, H: int = 480, seed: int = 0, intrinsics=None) -> Scene:
    g = torch.Generator().manual_seed(seed)
    cam = make_camera(W, H, intrinsics=intrinsics)

    def U(*shape):
        return torch.rand(*shape, generator=g)

    def Nrm(*shape):
        return torch.randn(*shape, generator=g)

    u = (U(N) * 1.10 - 0.05) * W
    v = (U(N) * 1.10 - 0.05) * H
    z = 0.5 + 5.5 * U(N)
    xyz = torch.stack([(u - cam.cx) * z / cam.fx, (v - cam.cy) * z / cam.fy, z], dim=1)
    sigma_px = torch.exp(math.log(1.5) + 0.6 * Nrm(N))
    s_world = (sigma_px * z / cam.fx)[:, None] * torch.exp(0.3 * Nrm(N, 3))
    rot = Nrm(N, 4)
    rot = rot / rot.norm(dim=1, keepdim=True)
    opacity_logit = 1.5 * Nrm(N, 1)
    rgb = U(N, 3)
    f_dc = ((rgb - 0.5) / SH_C0)[:, None, :]
    gt_image = U(3, H, W)
    gt_depth = 0.5 + 5.5 * U(1, H, W)
    return Scene(cam, xyz.contiguous(), torch.log(s_world).contiguous(), rot.contiguous(),
                 opacity_logit.contiguous(), f_dc.contiguous(), gt_image, gt_depth,
                 torch.zeros(3))


def activated(scene: Scene):
    """(means3D, scales, rotations, opacities, shs) as GaussianModel's getters give
    them to render() (gaussian_model.py:77-102)."""
    return (scene.means3D, torch.exp(scene.log_scales),
            torch.nn.functional.normalize(scene.rot), torch.sigmoid(scene.opacity_logit),
            scene.features_dc)


def synthetic_loss(image, depth, scene: Scene):
    """L = mean|image-G| + 0.05*mean|depth-Gd| (mirrors the shape of
    slam_utils.py:243-253)."""
    return ((image - scene.gt_image.to(image)).abs().mean()
            + 0.05 * (depth - scene.gt_depth.to(depth)).abs().mean())
