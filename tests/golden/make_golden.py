"""Generate golden vectors from the REFERENCE's own in-tree Python helpers.

Run in the build container only (the reference tree does not exist on the GPU box):
    python tests/golden/make_golden.py
It imports, on the CPU, the pieces of /root/reference that ARE present (the CUDA
rasteriser itself is an absent submodule - see oracle/torch_raster.py) and stores their
outputs on seeded inputs in tests/golden/reference_helpers.npz:

  * camera matrices          utils/camera_utils.py:94-104, graphics_utils.py:56-77,84
  * 3-D covariance           gaussian_splatting/utils/general_utils.py:98-149
  * SH evaluation, RGB2SH    gaussian_splatting/utils/sh_utils.py:55-126
  * losses, Huber            utils/slam_utils.py:58-75,188-253

utils/pose_utils.py cannot be imported here (it imports lietorch, which is not installed:
ordinary ModuleNotFoundError), so SE3_exp is pinned in tests/test_cpu_oracle.py against
torch.linalg.matrix_exp instead.

Only arrays (inputs and expected outputs) are stored; no reference source is copied.
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, REF)
import utils.configs as _cfg  # noqa: E402

_cfg.cuda_device = "cpu"
from gaussian_splatting.utils.general_utils import (build_rotation, build_scaling_rotation,  # noqa: E402
                                                    strip_symmetric)
from gaussian_splatting.utils.graphics_utils import focal2fov, getProjectionMatrix2  # noqa: E402
from gaussian_splatting.utils.sh_utils import RGB2SH, eval_sh  # noqa: E402
from utils.camera_utils import Camera  # noqa: E402
from utils.slam_utils import (HuberLoss, get_loss_mapping, get_loss_tracking_per_pixel)  # noqa: E402

g = torch.Generator().manual_seed(1234)
out = {}

# --- camera ---------------------------------------------------------------------------
fx, fy, cx, cy, W, H = 535.4, 539.2, 320.1, 247.6, 640, 480
P = getProjectionMatrix2(znear=0.01, zfar=100.0, fx=fx, fy=fy, cx=cx, cy=cy, W=W, H=H).transpose(0, 1)
tw = torch.zeros(4, 4)
tw[:3, 3] = torch.tensor([0.05, -0.03, 0.1])
tw[0, 1], tw[0, 2], tw[1, 0], tw[1, 2], tw[2, 0], tw[2, 1] = -0.03, -0.04, 0.03, -0.02, 0.04, 0.02
T = torch.linalg.matrix_exp(tw)
cam = Camera(0, torch.rand(3, H, W, generator=g), None, torch.eye(4), P, fx, fy, cx, cy,
             focal2fov(fx, W), focal2fov(fy, H), H, W, device="cpu")
cam.T = T.clone()
out["cam_intr"] = np.array([fx, fy, cx, cy, W, H], dtype=np.float64)
out["cam_T"] = T.numpy()
out["cam_projection_matrix"] = P.numpy()
out["cam_world_view"] = cam.world_view_transform.numpy()
out["cam_full_proj"] = cam.full_proj_transform.numpy()
out["cam_fov"] = np.array([focal2fov(fx, W), focal2fov(fy, H)], dtype=np.float64)

# --- covariance ------------------------------------------------------------------------
s = torch.exp(torch.randn(32, 3, generator=g) * 0.5 - 2.0)
q = torch.randn(32, 4, generator=g)
L = build_scaling_rotation(1.3 * s, q)
out["cov_scale"], out["cov_quat"] = s.numpy(), q.numpy()
out["cov_modifier"] = np.array(1.3)
out["cov_R"] = build_rotation(q).numpy()
out["cov_packed"] = strip_symmetric(L @ L.transpose(1, 2)).numpy()

# --- SH --------------------------------------------------------------------------------
sh = torch.randn(24, 3, 16, generator=g)          # eval_sh layout [..., C, K]
d = torch.nn.functional.normalize(torch.randn(24, 3, generator=g), dim=1)
out["sh_coeffs"], out["sh_dirs"] = sh.numpy(), d.numpy()
for deg in range(4):
    out[f"sh_rgb_deg{deg}"] = eval_sh(deg, sh, d).numpy()
rgb = torch.rand(10, 3, generator=g)
out["rgb2sh_in"], out["rgb2sh_out"] = rgb.numpy(), RGB2SH(rgb).numpy()

# --- losses ----------------------------------------------------------------------------
h, w = 24, 32
img = torch.rand(3, h, w, generator=g)
dep = torch.rand(1, h, w, generator=g) * 4
opa = torch.rand(1, h, w, generator=g)
gt = torch.rand(3, h, w, generator=g)
gt_depth = torch.rand(1, h, w, generator=g) * 4
gt_depth[gt_depth < 0.4] = 0.0


class VP:
    pass


vp = VP()
vp.original_image = gt
vp.exposure_a = torch.tensor([0.9])
vp.exposure_b = torch.tensor([0.05])
vp.exposure_eps = 1e-8
vp.rgb_pixel_mask_mapping = (gt.sum(0) > 0.3).view(1, h, w)
vp.gt_depth = gt_depth
out.update(loss_img=img.numpy(), loss_depth=dep.numpy(), loss_opacity=opa.numpy(), loss_gt=gt.numpy(),
           loss_gt_depth=gt_depth.numpy(), loss_mask=vp.rgb_pixel_mask_mapping.numpy(),
           loss_exposure=np.array([0.9, 0.05, 1e-8]))
cfg_mono = {"Training": {"monocular": True, "rgb_boundary_threshold": 0.3}}
cfg_rgbd = {"Training": {"monocular": False, "rgb_boundary_threshold": 0.3, "alpha": 0.9}}
out["loss_mapping_mono"] = get_loss_mapping(cfg_mono, img, dep, vp, opa).numpy()
out["loss_mapping_mono_init"] = get_loss_mapping(cfg_mono, img, dep, vp, opa, initialization=True).numpy()
out["loss_mapping_rgbd"] = get_loss_mapping(cfg_rgbd, img, dep, vp, opa).numpy()
out["loss_tracking_pp"] = get_loss_tracking_per_pixel(cfg_mono, img, dep, opa, vp).numpy()
x = torch.linspace(-1, 1, 41, requires_grad=True)
y = HuberLoss.apply(x, 0.1)
y.sum().backward()
out["huber_x"], out["huber_y"], out["huber_dx"] = x.detach().numpy(), y.detach().numpy(), x.grad.numpy()

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_helpers.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, os.path.getsize(dst), "bytes;", len(out), "arrays")
