"""Mirror of MonoGS's render() (row a1 of SURVEY §8a; reference:
/root/reference gaussian_splatting/gaussian_renderer/__init__.py:25-180).

Same call signature and the same result dict, so the SLAM loops can be driven without the
reference tree; the rasteriser underneath is monogs_amd.rasterizer (HIP).  With
<repo>/dropin on PYTHONPATH the reference's own file also works unchanged.

Duck-typed inputs
  pc:        get_xyz, get_opacity, get_scaling, get_rotation, get_features,
             active_sh_degree, max_sh_degree (+ get_covariance() for compute_cov3D_python)
  viewpoint: FoVx, FoVy, image_height, image_width, world_view_transform,
             full_proj_transform, projection_matrix, camera_center, cam_rot_delta,
             cam_trans_delta
"""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

_SKETCH_KEYS = ("sketch_mode", "sketch_dim", "stack_dim", "sketch_dtau", "sketch_indices")
_SKETCH_OFF = (0, 0, 0, None, None)


def _settings_for(view, pc, bg_color, scaling_modifier):
    return GaussianRasterizationSettings(
        int(view.image_height), int(view.image_width), math.tan(0.5 * view.FoVx),
        math.tan(0.5 * view.FoVy), bg_color, scaling_modifier, view.world_view_transform,
        view.full_proj_transform, view.projection_matrix, pc.active_sh_degree,
        view.camera_center, False, False)


def _shape_inputs(pc, pipe, scaling_modifier):
    """(scales, rotations, cov3D_precomp): exactly one of the two forms (:85-96)."""
    if getattr(pipe, "compute_cov3D_python", False):
        return None, None, pc.get_covariance(scaling_modifier)
    scales = pc.get_scaling
    if scales.shape[-1] == 1:                       # isotropic model: broadcast to 3 axes
        scales = scales.repeat(1, 3)
    return scales, pc.get_rotation, None


def _colour_inputs(pc, pipe, view, override_color):
    """(shs, colors_precomp): SH evaluated natively unless asked otherwise (:100-116)."""
    if override_color is not None:
        return None, override_color
    if not getattr(pipe, "convert_SHs_python", False):
        return pc.get_features, None
    from .sh import eval_sh
    feats = pc.get_features
    coeffs = feats.transpose(1, 2).reshape(-1, 3, (pc.max_sh_degree + 1) ** 2)
    dirs = pc.get_xyz - view.camera_center.repeat(feats.shape[0], 1)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    return None, torch.clamp_min(eval_sh(pc.active_sh_degree, coeffs, dirs) + 0.5, 0.0)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0,
           override_color=None, mask=None, num_backward_gaussians=-1, forward_sketch_args=None):
    xyz = pc.get_xyz
    if xyz.shape[0] == 0:
        return None
    # leaf whose gradient is dL/d(ndc) of the projected means (densification statistic)
    screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0
    if screenspace_points.requires_grad:
        screenspace_points.retain_grad()

    scales, rotations, cov3D = _shape_inputs(pc, pipe, scaling_modifier)
    shs, colors = _colour_inputs(pc, pipe, viewpoint_camera, override_color)
    per_gaussian = dict(means3D=xyz, means2D=screenspace_points, opacities=pc.get_opacity, shs=shs,
                        colors_precomp=colors, scales=scales, rotations=rotations,
                        cov3D_precomp=cov3D)
    if mask is not None:
        per_gaussian = {k: (None if t is None else t[mask]) for k, t in per_gaussian.items()}
    sketch = dict(zip(_SKETCH_KEYS, _SKETCH_OFF))
    if forward_sketch_args is not None:
        sketch = {k: forward_sketch_args[k] for k in _SKETCH_KEYS}

    rasterizer = GaussianRasterizer(_settings_for(viewpoint_camera, pc, bg_color, scaling_modifier))
    image, radii, depth, opacity, n_touched = rasterizer(
        theta=viewpoint_camera.cam_rot_delta, rho=viewpoint_camera.cam_trans_delta,
        num_backward_gaussians=num_backward_gaussians, **per_gaussian, **sketch)
    return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii, "depth": depth, "opacity": opacity, "n_touched": n_touched}
