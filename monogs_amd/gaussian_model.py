"""The Gaussian map as the mapping loop sees it: raw parameters, their optimiser, the
densification statistics and the per-Gaussian keyframe bookkeeping, all resident in HBM.

Host-side mirror of the parts of /root/reference gaussian_splatting/scene/gaussian_model.py
that the hot path touches (same attribute and method names, so render(), the loop bodies and
map_update.py work on either):
  activations / getters            :54-102
  create_pcd_from_image            :108-205   -> keyframe_init.py (device back-projection + knn)
  extend_from_pcd[_seq]            :210-245   -> one mgs_map_append launch
  training_setup / learning rate   :247-312
  reset_opacity[_nonvisible]       :364-377   -> mgs_map_finish_iteration (reset_mode)
  prune / densify                  :485-691   -> map_update.py (plan + gather kernels)
  add_densification_stats          :693-697
PLY I/O is not part of the path.  Differences by design: `unique_kfIDs` / `n_obs` are device
int32 tensors (the reference keeps them on the CPU, forcing a sync per prune), and the
optimiser is FusedGaussianAdam (one launch per step).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _cabi
from .map_update import FusedGaussianAdam, _ATTR, _stream
from .map_update import densify_and_prune as _densify_and_prune
from .map_update import prune_points as _prune_points


def inverse_sigmoid(x: float) -> float:
    return math.log(x / (1.0 - x))


def expon_lr(step, lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear decay with an optional eased start (general_utils.py:80-95, `helper`)."""
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    delay = 1.0
    if lr_delay_steps > 0:
        delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
    t = min(max(step / max_steps, 0.0), 1.0)
    return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class OptimizationParams:
    """opt_params of configs/mono/tum/base_config.yaml:279-293 (defaults)."""
    position_lr_init = 0.0016
    position_lr_final = 0.0000016
    position_lr_delay_mult = 0.01
    position_lr_max_steps = 30000
    feature_lr = 0.0025
    opacity_lr = 0.05
    scaling_lr = 0.001
    rotation_lr = 0.001
    percent_dense = 0.01
    lambda_dssim = 0.2
    densification_interval = 100
    opacity_reset_interval = 3000
    densify_from_iter = 500
    densify_until_iter = 15000
    densify_grad_threshold = 0.0002

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


class GaussianModel:
    def __init__(self, sh_degree: int = 0, config: Optional[dict] = None, device="cuda", isotropic: bool = False):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self.device = torch.device(device)
        self.config = config
        self.isotropic = isotropic
        K = (sh_degree + 1) ** 2
        dev = self.device
        e = lambda *shape: nn.Parameter(torch.empty(*shape, device=dev))
        self._xyz, self._features_dc, self._features_rest = e(0, 3), e(0, 1, 3), e(0, K - 1, 3)
        self._scaling, self._rotation, self._opacity = e(0, 1 if isotropic else 3), e(0, 4), e(0, 1)
        self.max_radii2D = torch.zeros(0, device=dev)
        self.xyz_gradient_accum = torch.zeros(0, 1, device=dev)
        self.denom = torch.zeros(0, 1, device=dev)
        self.unique_kfIDs = torch.zeros(0, dtype=torch.int32, device=dev)
        self.n_obs = torch.zeros(0, dtype=torch.int32, device=dev)
        self.optimizer = None
        self.percent_dense = 0.01
        self.spatial_lr_scale = 0.0

    # ---- activations (:54-102) ------------------------------------------------------------
    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def __len__(self):
        return int(self._xyz.shape[0])

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    def init_lr(self, spatial_lr_scale):
        self.spatial_lr_scale = spatial_lr_scale

    # ---- optimiser (:247-312) ---------------------------------------------------------------
    def training_setup(self, opt=None):
        opt = opt or OptimizationParams()
        self.percent_dense = opt.percent_dense
        n = len(self)
        self.xyz_gradient_accum = torch.zeros(n, 1, device=self.device)
        self.denom = torch.zeros(n, 1, device=self.device)
        s = self.spatial_lr_scale
        groups = [
            {"params": [self._xyz], "lr": opt.position_lr_init * s, "name": "xyz"},
            {"params": [self._features_dc], "lr": opt.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": opt.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": opt.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": opt.scaling_lr * s, "name": "scaling"},
            {"params": [self._rotation], "lr": opt.rotation_lr, "name": "rotation"},
        ]
        self.optimizer = FusedGaussianAdam(groups, lr=0.0, eps=1e-15)
        self.lr_init, self.lr_final = opt.position_lr_init * s, opt.position_lr_final * s
        self.lr_delay_mult, self.max_steps = opt.position_lr_delay_mult, opt.position_lr_max_steps

    def update_learning_rate(self, iteration):
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g["lr"] = expon_lr(iteration, self.lr_init, self.lr_final, lr_delay_mult=self.lr_delay_mult,
                                   max_steps=self.max_steps)
                return g["lr"]

    # ---- keyframe insertion (:108-245) ------------------------------------------------------
    def create_pcd_from_image(self, cam, init=False, scale=2.0, depthmap=None, generator=None):
        """Back-projection of a keyframe into new Gaussians on the device.  `depthmap` [H,W]
        (tensor, metres) as the frontend passes it; monocular without one: the noisy constant-depth
        prior of :124-129."""
        from .keyframe_init import create_pcd_from_image_and_depth, monocular_depth_prior
        ds = (self.config or {}).get("Dataset", {})
        H, W = int(cam.image_height), int(cam.image_width)
        if depthmap is None:
            depth = getattr(cam, "gt_depth", None)
            if depth is None or ds.get("sensor_type", "monocular") == "monocular":
                depth = monocular_depth_prior(H, W, scale, self.device, generator)
        else:
            depth = depthmap
        depth = torch.as_tensor(depth, dtype=torch.float32, device=self.device).reshape(H, W)
        return create_pcd_from_image_and_depth(
            cam, cam.original_image, depth,
            downsample_factor=ds.get("pcd_downsample_init" if init else "pcd_downsample", 32 if init else 64),
            point_size=ds.get("point_size", 0.01), adaptive_pointsize=ds.get("adaptive_pointsize", True),
            isotropic=self.isotropic, max_sh_degree=self.max_sh_degree, generator=generator)

    def extend_from_pcd(self, fused_point_cloud, features, scales, rots, opacities, kf_id):
        """Append rows to every per-Gaussian tensor - parameters, Adam moments (zeros), keyframe
        ids, observation counts - in one launch; statistics restart from zero for ALL Gaussians
        (densification_postfix :591-594).  `features` is [P,3,K] as create_pcd returns it."""
        dev = self.device
        new = {
            "xyz": fused_point_cloud.detach().float().contiguous(),
            "f_dc": features[:, :, 0:1].transpose(1, 2).detach().float().contiguous(),
            "f_rest": features[:, :, 1:].transpose(1, 2).detach().float().contiguous(),
            "opacity": opacities.detach().float().reshape(-1, 1).contiguous(),
            "scaling": scales.detach().float().contiguous(),
            "rotation": rots.detach().float().contiguous(),
        }
        n_old, n_new = len(self), int(new["xyz"].shape[0])
        if n_new == 0:
            return
        rows = n_old + n_new
        args = _cabi.MapAppendArgs()
        keep, k = [], 0

        def add(old, extra):
            nonlocal k
            old = old.detach().contiguous()
            width = 1
            for d in old.shape[1:]:
                width *= int(d)
            dst = torch.empty((rows,) + tuple(old.shape[1:]), dtype=old.dtype, device=dev)
            if width > 0:
                assert old.element_size() == 4 and (extra is None or (extra.element_size() == 4 and tuple(extra.shape[1:]) == tuple(old.shape[1:])))
                t = args.tensors[k]
                t.src, t.dst, t.width, t.mode = old.data_ptr(), dst.data_ptr(), width, 0
                args.new_rows[k] = None if extra is None else extra.data_ptr()
                k += 1
            keep.append((old, extra, dst))
            return dst

        new_params, new_states = {}, {}
        have_opt = self.optimizer is not None
        names = [g["name"] for g in self.optimizer.param_groups] if have_opt else list(new)
        for name in names:
            p = getattr(self, _ATTR[name])
            new_params[name] = add(p, new[name])
            st = self.optimizer.state.get(p) if have_opt else None
            if st is not None and "exp_avg" in st:
                new_states[name] = (add(st["exp_avg"], None), add(st["exp_avg_sq"], None), st)
        kf = torch.full((n_new,), int(kf_id), dtype=torch.int32, device=dev)
        new_kf = add(self.unique_kfIDs.to(dev, torch.int32), kf)
        new_obs = add(self.n_obs.to(dev, torch.int32), None)
        args.num_tensors, args.rows_old, args.rows_new = k, n_old, n_new
        _cabi.check(_cabi.lib().mgs_map_append(C.byref(args), _stream(dev)), "mgs_map_append")
        for name in names:
            old = getattr(self, _ATTR[name])
            par = nn.Parameter(new_params[name].requires_grad_(True))
            if have_opt:
                for g in self.optimizer.param_groups:
                    if g["name"] == name:
                        g["params"][0] = par
                if name in new_states:
                    m, v, st = new_states[name]
                    st["exp_avg"], st["exp_avg_sq"] = m, v
                    del self.optimizer.state[old]
                    self.optimizer.state[par] = st
            setattr(self, _ATTR[name], par)
        self.unique_kfIDs, self.n_obs = new_kf, new_obs
        self.xyz_gradient_accum = torch.zeros(rows, 1, device=dev)
        self.denom = torch.zeros(rows, 1, device=dev)
        self.max_radii2D = torch.zeros(rows, device=dev)

    def extend_from_pcd_seq(self, cam_info, kf_id=-1, init=False, scale=2.0, depthmap=None, generator=None):
        pcd = self.create_pcd_from_image(cam_info, init, scale=scale, depthmap=depthmap, generator=generator)
        self.extend_from_pcd(*pcd, kf_id)

    # ---- maintenance (:364-377, :485-697) ---------------------------------------------------------
    def _reset(self, mode: int, value: float, denom_inc: Optional[torch.Tensor]):
        n = len(self)
        if n == 0:
            return
        a = _cabi.MapFinishArgs()
        a.num_gaussians, a.reset_mode, a.reset_value = n, mode, value
        a.opacity_logits = self._opacity.data_ptr()
        if denom_inc is not None:
            denom_inc = denom_inc.to(self.device, torch.float32).reshape(-1).contiguous()
            a.denom_inc = denom_inc.data_ptr()
        st = self.optimizer.state.get(self._opacity) if self.optimizer is not None else None
        if st is not None and "exp_avg" in st:
            a.opacity_exp_avg, a.opacity_exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        _cabi.check(_cabi.lib().mgs_map_finish_iteration(C.byref(a), _stream(self.device)), "mgs_map_finish_iteration")

    def reset_opacity(self):
        """opacity <- 0.01 for every Gaussian, the opacity group's Adam moments zeroed (:364-367)."""
        self._reset(1, 0.01, None)

    def reset_opacity_nonvisible(self, visibility_filters, keep_visible_logits=False):
        """opacity <- 0.4 for the Gaussians outside every filter (:369-377).  As in the reference, a Gaussian
        INSIDE a filter gets its activated opacity as the new raw parameter (:375: logit <- sigmoid(logit));
        `keep_visible_logits=True` leaves those logits alone instead."""
        seen = torch.zeros(len(self), device=self.device)
        for f in visibility_filters:
            seen += f.to(self.device, torch.float32)
        self._reset(3 if keep_visible_logits else 2, 0.4, seen)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, noise=None, generator=None):
        _densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, noise=noise, generator=generator)

    def prune_points(self, mask):
        _prune_points(self, mask)

    @torch.no_grad()
    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        g = torch.linalg.norm(viewspace_point_tensor.grad[:, :2], dim=-1, keepdim=True)
        self.xyz_gradient_accum += torch.where(update_filter[:, None], g, torch.zeros_like(g))
        self.denom += update_filter[:, None].float()
