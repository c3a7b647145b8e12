"""Keyframe insertion on the device.

Contract of /root/reference gaussian_splatting/scene/gaussian_model.py:108-205
(`create_pcd_from_image` / `create_pcd_from_image_and_depth`): back-project the valid depth
pixels of a keyframe to world space, keep a random 1/downsample of them, and initialise new
Gaussians there - SH-0 colour from the image, isotropic log-scale from the HIP knn
(`distCUDA2`), identity rotation, opacity 0.5.

The reference does the back-projection and the sub-sampling with open3d on the CPU (two
host round trips per keyframe); here everything stays in HBM.  The sub-sample is drawn
with torch.randperm, so WHICH pixels are kept differs from open3d's generator (same
distribution; parity of the selection is unpinned), everything else is deterministic given
the selection.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .knn import distCUDA2
from .sh import RGB2SH


def monocular_depth_prior(height: int, width: int, scale: float = 2.0, device="cuda",
                          generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """Depth guess used when no depth sensor exists (gaussian_model.py:124-129):
    (1 + (N(0,1) - 0.5) * 0.05) * scale per pixel."""
    n = torch.randn(height, width, device=device, generator=generator)
    return (1.0 + (n - 0.5) * 0.05) * scale


def create_pcd_from_image_and_depth(cam, image: torch.Tensor, depth: torch.Tensor, *,
                                    downsample_factor: float, point_size: float = 0.01,
                                    adaptive_pointsize: bool = True, isotropic: bool = True,
                                    max_sh_degree: int = 0, depth_trunc: float = 100.0,
                                    generator: Optional[torch.Generator] = None):
    """cam: fx, fy, cx, cy, T (4x4 world-to-camera on the device), exposure_a/b/eps.
    image [3,H,W] in [0,1]; depth [H,W] metres (<= 0 or > depth_trunc = invalid).
    Returns (xyz[P,3], features[P,3,K], log_scales[P,1|3], rots[P,4], opacity_logit[P,1])."""
    dev = depth.device
    H, W = depth.shape
    with torch.no_grad():
        img = (torch.abs(cam.exposure_a) + cam.exposure_eps) * image + cam.exposure_b
        rgb = torch.floor(torch.clamp(img, 0.0, 1.0) * 255.0) / 255.0      # uint8 round trip
        valid = (depth > 0) & (depth <= depth_trunc)
        if adaptive_pointsize:
            point_size = min(0.05, point_size * float(torch.median(depth)))
        idx = torch.nonzero(valid.reshape(-1)).reshape(-1)
        keep = int(idx.numel() / downsample_factor)
        sel = idx[torch.randperm(idx.numel(), device=dev, generator=generator)[:keep]]
        v, u = torch.div(sel, W, rounding_mode="floor").float(), (sel % W).float()
        z = depth.reshape(-1)[sel].float()
        p_cam = torch.stack([(u - cam.cx) * z / cam.fx, (v - cam.cy) * z / cam.fy, z], dim=1)
        R, t = cam.T[:3, :3].float(), cam.T[:3, 3].float()
        xyz = (p_cam - t) @ R                 # R^T (p - t), row-vector form
        col = rgb.reshape(3, -1)[:, sel].t().contiguous()
        K = (max_sh_degree + 1) ** 2
        feats = torch.zeros(keep, 3, K, device=dev)
        feats[:, :, 0] = RGB2SH(col)
        dist2 = torch.clamp_min(distCUDA2(xyz.contiguous()), 1e-7) * point_size
        scales = torch.log(torch.sqrt(dist2))[:, None]
        if not isotropic:
            scales = scales.repeat(1, 3)
        rots = torch.zeros(keep, 4, device=dev)
        rots[:, 0] = 1.0
        opac = torch.full((keep, 1), math.log(0.5 / (1 - 0.5)), device=dev)   # inverse_sigmoid(0.5)
    return xyz, feats, scales, rots, opac
