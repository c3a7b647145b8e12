"""Mirror of MonoGS's render() (row a1 of SURVEY §8a):
/root/reference gaussian_splatting/gaussian_renderer/__init__.py:25-180.

Same signature, same argument marshalling and the same result dict; the rasteriser
underneath is monogs_amd.rasterizer (HIP).  With <repo>/dropin on PYTHONPATH the
reference's own file works unchanged; this copy exists so the hot path can be driven
(tests, bench, tracking/mapping loops) without the reference tree.

`pc` needs: get_xyz, get_opacity, get_scaling, get_rotation, get_features,
active_sh_degree, max_sh_degree and, when pipe.compute_cov3D_python, get_covariance().
`viewpoint_camera` needs: FoVx, FoVy, image_height, image_width, world_view_transform,
full_proj_transform, projection_matrix, camera_center, cam_rot_delta, cam_trans_delta.
"""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

SH_C0 = 0.28209479177387814


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0,
           override_color=None, mask=None, num_backward_gaussians=-1, forward_sketch_args=None):
    if pc.get_xyz.shape[0] == 0:
        return None
    xyz = pc.get_xyz
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        projmatrix_raw=viewpoint_camera.projection_matrix,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=False)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D, means2D, opacity = xyz, screenspace_points, pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        sc = pc.get_scaling
        scales = sc.repeat(1, 3) if sc.shape[-1] == 1 else sc
        rotations = pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False):
            from .sh import eval_sh
            feats = pc.get_features
            shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    sk = forward_sketch_args or {}
    sel = (lambda t: t) if mask is None else (lambda t: None if t is None else t[mask])
    rendered_image, radii, depth, opacity_img, n_touched = rasterizer(
        means3D=sel(means3D), means2D=sel(means2D), shs=sel(shs), colors_precomp=sel(colors_precomp),
        opacities=sel(opacity), scales=sel(scales), rotations=sel(rotations),
        cov3D_precomp=sel(cov3D_precomp), theta=viewpoint_camera.cam_rot_delta,
        rho=viewpoint_camera.cam_trans_delta, num_backward_gaussians=num_backward_gaussians,
        sketch_mode=sk.get("sketch_mode", 0), sketch_dim=sk.get("sketch_dim", 0),
        stack_dim=sk.get("stack_dim", 0), sketch_dtau=sk.get("sketch_dtau"),
        sketch_indices=sk.get("sketch_indices"))
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii, "depth": depth,
            "opacity": opacity_img, "n_touched": n_touched}
