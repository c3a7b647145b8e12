"""Inner-loop bodies of MonoGS tracking and mapping, driving the HIP rasteriser.

Rows a12 / a13 of SURVEY §8a.  Mirrors, for the loop bodies only (keyframe management,
densification policy, queues and GUI are out of scope):
  * tracking, first order   /root/reference utils/slam_frontend.py:455-630
  * tracking, second order  utils/slam_frontend.py:269-338 (sketch args), :632-710 (LM step)
  * mapping                 utils/slam_backend.py:171-332
Hyper-parameters default to configs/mono/tum/base_config.yaml:245-290.
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn

from .gaussian_renderer import render
from .losses import HuberLoss, get_loss_mapping, get_loss_tracking_per_pixel
from .pose import SE3_exp, update_pose


class ViewCamera(nn.Module):
    """The subset of utils/camera_utils.py:Camera (:10-108) the loop bodies touch."""

    def __init__(self, uid, image, T_w2c, projection_matrix, fovx, fovy, H, W, device,
                 gt_depth=None, intrinsics=None):
        super().__init__()
        self.uid, self.device = uid, device
        if intrinsics is None:      # pinhole with a centred principal point, from the field of view
            intrinsics = (W / (2 * math.tan(0.5 * fovx)), H / (2 * math.tan(0.5 * fovy)), 0.5 * W, 0.5 * H)
        self.fx, self.fy, self.cx, self.cy = (float(v) for v in intrinsics)
        self.T = T_w2c.to(device=device, dtype=torch.float32).clone()
        self.original_image = image.to(device)
        self.FoVx, self.FoVy, self.image_height, self.image_width = fovx, fovy, H, W
        self.cam_rot_delta = nn.Parameter(torch.zeros(3, device=device))
        self.cam_trans_delta = nn.Parameter(torch.zeros(3, device=device))
        self.exposure_eps = 1e-8
        self.exposure_a = nn.Parameter(torch.tensor([1.0], device=device))
        self.exposure_b = nn.Parameter(torch.tensor([0.0], device=device))
        self.projection_matrix = projection_matrix.to(device)
        self.rgb_pixel_mask_mapping = (self.original_image.sum(dim=0) > 0.01).view(1, H, W)
        self.gt_depth = None if gt_depth is None else gt_depth.to(device)

    @property
    def world_view_transform(self):
        return self.T.transpose(0, 1)

    @property
    def full_proj_transform(self):
        return self.world_view_transform @ self.projection_matrix

    @property
    def camera_center(self):
        return self.world_view_transform     # as in the reference (camera_utils.py:106-108)


class GaussianParams(nn.Module):
    """Leaf parameters + activations as GaussianModel exposes them to render()
    (gaussian_splatting/scene/gaussian_model.py:54-102)."""

    def __init__(self, xyz, log_scales, rot, opacity_logit, features_dc, features_rest=None):
        super().__init__()
        self._xyz = nn.Parameter(xyz.clone())
        self._scaling = nn.Parameter(log_scales.clone())
        self._rotation = nn.Parameter(rot.clone())
        self._opacity = nn.Parameter(opacity_logit.clone())
        self._features_dc = nn.Parameter(features_dc.clone())
        rest = features_rest if features_rest is not None else features_dc.new_zeros(
            features_dc.shape[0], 0, 3)
        self._features_rest = nn.Parameter(rest.clone())
        self.active_sh_degree = 0
        self.max_sh_degree = int(math.isqrt(1 + rest.shape[1])) - 1

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))


class Pipe:
    compute_cov3D_python = False
    convert_SHs_python = False


DEFAULT_CONFIG = {"Training": {"monocular": True, "rgb_boundary_threshold": 0.01,
                               "lr": {"cam_rot_delta": 0.003, "cam_trans_delta": 0.001,
                                      "exposure_a": 0.02, "exposure_b": 0.02},
                               "RGN": {"use_huber": True, "huber_delta": 0.01, "pnorm": 1}}}


def tracking_norm(config=DEFAULT_CONFIG):
    """(huber_delta, p) of the first-order tracking objective as the reference chooses them
    (slam_frontend.py:596-600): Huber + L2 when RGN.use_huber, else no Huber (delta 0) and the
    RGN.pnorm-norm (configs/mono/tum/base_config.yaml:247-249 ships use_huber True, pnorm 1)."""
    rgn = config["Training"]["RGN"]
    if rgn["use_huber"]:
        return float(rgn["huber_delta"]), 2.0
    return 0.0, float(rgn["pnorm"])


def make_pose_optimizer(viewpoint: ViewCamera, config=DEFAULT_CONFIG):
    lr = config["Training"]["lr"]
    return torch.optim.Adam([
        {"params": [viewpoint.cam_rot_delta], "lr": lr["cam_rot_delta"]},
        {"params": [viewpoint.cam_trans_delta], "lr": lr["cam_trans_delta"]},
        {"params": [viewpoint.exposure_a], "lr": lr["exposure_a"]},
        {"params": [viewpoint.exposure_b], "lr": lr["exposure_b"]}])


def tracking_step_first_order(viewpoint, gaussians, pose_optimizer, background, pipe=Pipe,
                              config=DEFAULT_CONFIG):
    """One first-order tracking iteration (slam_frontend.py:493-630): render, per-pixel
    residual, Huber + L2 norm (or the RGN.pnorm-norm without Huber, :596-600), backward, Adam on (rot, trans, exposure), update_pose."""
    render_pkg = render(viewpoint, gaussians, pipe, background)
    res = get_loss_tracking_per_pixel(config, render_pkg["render"], render_pkg["depth"],
                                      render_pkg["opacity"], viewpoint)
    # the reference's best-iterate criterion: ||residual||_1 before Huber (slam_frontend.py:510)
    render_pkg["tracking_l1"] = res.detach().abs().sum()
    delta, p = tracking_norm(config)
    if delta > 0:
        res = HuberLoss.apply(res, delta)
    loss = torch.norm(res.flatten(), p=p)
    pose_optimizer.zero_grad()
    loss.backward()
    with torch.no_grad():
        pose_optimizer.step()
        render_pkg["tracking_step_norm"] = torch.cat([viewpoint.cam_trans_delta, viewpoint.cam_rot_delta]).norm()
        converged = update_pose(viewpoint)
    return loss.detach(), converged, render_pkg


def tracking_step_first_order_fused(viewpoint, gaussians, fused_optimizer, background, pipe=Pipe,
                                    config=DEFAULT_CONFIG):
    """Same iteration as tracking_step_first_order with the loss and the optimiser step +
    update_pose as fused HIP launches (monogs_amd/tracking_fused.py).  Returns the
    convergence flag as a device tensor (no host sync inside)."""
    from .tracking_fused import tracking_loss
    render_pkg = render(viewpoint, gaussians, pipe, background)
    delta, p = tracking_norm(config)
    with torch.no_grad():
        render_pkg["tracking_l1"] = get_loss_tracking_per_pixel(
            config, render_pkg["render"], render_pkg["depth"], render_pkg["opacity"], viewpoint).abs().sum()
    loss = tracking_loss(render_pkg["render"], render_pkg["opacity"], viewpoint, delta, p)
    fused_optimizer.zero_grad()
    loss.backward()
    converged = fused_optimizer.step()
    return loss.detach(), converged, render_pkg


def gen_forward_sketch_args(height, width, repeat_dim, stack_dim, sketch_dim, device,
                            generator: Optional[torch.Generator] = None):
    """CountSketch bookkeeping of slam_frontend.py:269-338: every repeat draws a random
    permutation of the pixels, cuts its first chunk*stack*sketch entries into disjoint
    buckets of `chunk` pixels, and records the bucket id of every pixel per stack."""
    m, d = height * width, stack_dim * sketch_dim
    chunk = m // d
    idx = torch.full((repeat_dim, stack_dim * m), -1, dtype=torch.int32, device=device)
    rows = torch.empty(repeat_dim, stack_dim, sketch_dim, chunk, dtype=torch.int32, device=device)
    cols = torch.empty_like(rows)
    vals = torch.arange(sketch_dim, dtype=torch.int32, device=device).view(1, -1, 1).expand(
        stack_dim, sketch_dim, chunk).reshape(-1)
    offs = (torch.arange(stack_dim, device=device) * m).view(-1, 1, 1).expand(
        stack_dim, sketch_dim, chunk).reshape(-1)
    for i in range(repeat_dim):
        perm = torch.randperm(m, device=device, generator=generator)[:chunk * d]
        idx[i, perm + offs] = vals
        rows[i] = (perm // width).view(stack_dim, sketch_dim, chunk).to(torch.int32)
        cols[i] = (perm % width).view(stack_dim, sketch_dim, chunk).to(torch.int32)
    weights = torch.randint(0, 2, (repeat_dim, height, width), device=device,
                            generator=generator).float() * 2 - 1
    return {"sketch_mode": 1, "repeat_dim": repeat_dim, "stack_dim": stack_dim,
            "sketch_dim": sketch_dim,
            "sketch_indices": idx.reshape(repeat_dim, stack_dim, height, width),
            "rand_indices": (rows, cols), "rand_indices_row": rows, "rand_indices_col": cols,
            "sketch_dtau": torch.empty(stack_dim, sketch_dim, 6, device=device, requires_grad=True),
            "sketch_dexposure": torch.empty(stack_dim, sketch_dim, 2, device=device,
                                            requires_grad=True),
            "chunk_size": chunk, "rand_weights": weights}


def sketch_args_from_buckets(bucket: torch.Tensor, weights: torch.Tensor, height, width, stack_dim,
                             sketch_dim):
    """forward_sketch_args for given bucket partitions: `bucket` [H*W] or [repeat, H*W] =
    stack * sketch_dim + k or -1 and `weights` (same shape) = +-1, as mgs_sketch_assign produces them."""
    dev = bucket.device
    m, d = height * width, stack_dim * sketch_dim
    chunk = m // d
    bucket, weights = bucket.reshape(-1, m), weights.reshape(-1, m)
    R = bucket.shape[0]
    idx = torch.full((R, stack_dim, m), -1, dtype=torch.int32, device=dev)
    rows = torch.empty(R, stack_dim, sketch_dim, chunk, dtype=torch.int32, device=dev)
    cols = torch.empty_like(rows)
    for r in range(R):
        b = bucket[r].long()
        p = torch.nonzero(b >= 0).squeeze(1)
        idx[r, b[p] // sketch_dim, p] = (b[p] % sketch_dim).to(torch.int32)
        order = p[torch.argsort(b[p], stable=True)]                   # pixels grouped by bucket
        rows[r] = (order // width).view(stack_dim, sketch_dim, chunk).to(torch.int32)
        cols[r] = (order % width).view(stack_dim, sketch_dim, chunk).to(torch.int32)
    return {"sketch_mode": 1, "repeat_dim": R, "stack_dim": stack_dim, "sketch_dim": sketch_dim,
            "sketch_indices": idx.reshape(R, stack_dim, height, width),
            "rand_indices": (rows, cols), "rand_indices_row": rows, "rand_indices_col": cols,
            "sketch_dtau": torch.empty(stack_dim, sketch_dim, 6, device=dev, requires_grad=True),
            "sketch_dexposure": torch.empty(stack_dim, sketch_dim, 2, device=dev, requires_grad=True),
            "chunk_size": chunk, "rand_weights": weights.reshape(R, height, width).float()}


def tracking_step_second_order(viewpoint, gaussians, background, lambda_, repeat_dim=1,
                               stack_dim=16, sketch_dim=64, pipe=Pipe, config=DEFAULT_CONFIG,
                               generator=None, fused_solve=False, fsa=None, return_pkg=False):
    """One sketched Levenberg-Marquardt iteration (slam_frontend.py:484-710): sketched
    render, bucket-summed residual Sf, `repeat_dim` backward passes harvesting the sketched
    Jacobian SJ[(repeat*stack*sketch), 8], damped least squares, left-multiplicative pose
    step and exposure step.  `lambda_` is the damping, or a callable that maps this render's L1
    residual (the reference's loss_tracking_scalar) to it - the trust-region rule of :536-545 needs
    the current loss before the solve.  Returns (l1, x, SJ, Sf[, render_pkg])."""
    H, W = viewpoint.image_height, viewpoint.image_width
    m, dper = H * W, stack_dim * sketch_dim
    if fsa is None:
        fsa = gen_forward_sketch_args(H, W, repeat_dim, stack_dim, sketch_dim, viewpoint.device,
                                      generator)
    render_pkg = render(viewpoint, gaussians, pipe, background, forward_sketch_args=fsa)
    res = get_loss_tracking_per_pixel(config, render_pkg["render"], render_pkg["depth"],
                                      render_pkg["opacity"], viewpoint, forward_sketch_args=fsa)
    l1 = res.detach().abs().sum()      # loss_tracking_scalar (slam_frontend.py:510): before Huber
    if callable(lambda_):
        lambda_ = lambda_(l1)
    rgn = config["Training"]["RGN"]
    if rgn["use_huber"]:
        res = HuberLoss.apply(res, rgn["huber_delta"])
    res = res.sum(dim=0) / (m / dper)
    weighted = res * fsa["rand_weights"]
    rows, cols = fsa["rand_indices_row"].long(), fsa["rand_indices_col"].long()
    bi = torch.arange(repeat_dim, device=res.device).view(-1, 1, 1, 1)
    Sf = weighted[bi, rows, cols].sum(dim=-1)
    n = 8
    SJ = torch.empty(repeat_dim, stack_dim, sketch_dim, n, device=res.device)
    for i in range(repeat_dim):
        for p in (viewpoint.cam_rot_delta, viewpoint.cam_trans_delta, viewpoint.exposure_a,
                  viewpoint.exposure_b, fsa["sketch_dtau"], fsa["sketch_dexposure"]):
            p.grad = None
        weighted[i].backward(gradient=torch.ones_like(weighted[i]), retain_graph=True)
        SJ[i] = torch.cat((fsa["sketch_dtau"].grad, fsa["sketch_dexposure"].grad), dim=2)
    if fused_solve:    # one HIP launch: normal equations + Cholesky + pose / exposure step
        from .tracking_fused import lm_solve_step
        with torch.no_grad():
            SJ = SJ.reshape(-1, n)
            Sf = Sf.flatten()
            x = lm_solve_step(SJ, Sf, lambda_, viewpoint)
        return (l1, x, SJ, Sf, render_pkg) if return_pkg else (l1, x, SJ, Sf)
    with torch.no_grad():
        SJ = SJ.reshape(-1, n)
        Sf = Sf.flatten()
        A = torch.cat((SJ, torch.eye(n, device=SJ.device) * math.sqrt(lambda_)), dim=0)
        b = torch.cat((Sf, torch.zeros(n, device=SJ.device)), dim=0)
        x = torch.linalg.lstsq(A, -b).solution
        # TempCamera.step (slam_frontend.py:49-53): tau = x[:6] = [trans; rot], exposure x[6:8]
        viewpoint.T.copy_(SE3_exp(x[:6]) @ viewpoint.T)
        viewpoint.exposure_a += x[6]
        viewpoint.exposure_b += x[7]
    return (l1, x, SJ, Sf, render_pkg) if return_pkg else (l1, x, SJ, Sf)


class TempCamera:
    """Copy of the per-frame state the tracking loop snapshots and restores
    (slam_frontend.py:28-53)."""

    def __init__(self, viewpoint):
        self.T = viewpoint.T.detach().clone()
        self.exposure_a = viewpoint.exposure_a.detach().clone()
        self.exposure_b = viewpoint.exposure_b.detach().clone()

    def assign(self, viewpoint):
        with torch.no_grad():
            viewpoint.T.copy_(self.T)
            viewpoint.exposure_a.copy_(self.exposure_a)
            viewpoint.exposure_b.copy_(self.exposure_b)
            viewpoint.cam_rot_delta.zero_()
            viewpoint.cam_trans_delta.zero_()


def track_frame(viewpoint, gaussians, background, first_order_iters=40, second_order_iters=10,
                use_first_order_best=True, use_best_loss=True, pipe=Pipe, config=DEFAULT_CONFIG,
                stack_dim=16, sketch_dim=64, initial_lambda=1e-3, min_lambda=1e-6, max_lambda=1e7,
                increase_factor=5.0, decrease_factor=5.0, second_order_converged_threshold=1e-5,
                generator=None, fused=False, fsa_fn=None, repeat_dim=1, trace=None):
    """The reference's tracking loop for one frame, reference-shaped Python on the HIP rasteriser
    (slam_frontend.py:455-822 with override_mode "none"): first-order iterations (Adam on the pose
    deltas and the exposure; a converged one leaves the whole loop, :623-626), then sketched LM
    iterations with the lambda rule of :536-545.  Every iteration's L1 residual (before Huber, :510)
    is compared with the best so far and the rendered state snapshotted (:523-528); the second-order
    phase starts from the best first-order state (`use_first_order_best`, :465-470); the frame ends
    at the best state and returns ITS render_pkg (`use_best_loss`, :819-822).
    Returns (render_pkg, best_l1, best_iteration, iterations).  `fsa_fn(i)` may supply the sketch
    arguments of second-order iteration i (tests: the native tracker's partitions).  `trace` (a list)
    receives per iteration (L1 of its render, |step| it took, converged)."""
    if fused:
        from .tracking_fused import FusedPoseOptimizer
        lr = config["Training"]["lr"]
        opt = FusedPoseOptimizer(viewpoint, lr["cam_rot_delta"], lr["cam_trans_delta"], lr["exposure_a"],
                                 lr["exposure_b"])
        step = tracking_step_first_order_fused
    else:
        opt = make_pose_optimizer(viewpoint, config)
        step = tracking_step_first_order
    best_l1, best_state, best_pkg, best_it = float("inf"), None, None, -1
    lam, old_l1 = [initial_lambda], [None]

    def lambda_rule(l1):                       # slam_frontend.py:536-545, evaluated before the solve
        l1 = float(l1)
        if old_l1[0] is not None:
            lam[0] = (max(lam[0] / decrease_factor, min_lambda) if l1 < old_l1[0]
                      else min(lam[0] * increase_factor, max_lambda))
        old_l1[0] = l1
        return lam[0]

    pkg, it = None, 0
    for itr in range(first_order_iters + second_order_iters):
        second = itr >= first_order_iters
        if itr == first_order_iters and best_state is not None and use_first_order_best:
            best_state.assign(viewpoint)
        state = TempCamera(viewpoint)            # the state this iteration renders
        if not second:
            _, converged, pkg = step(viewpoint, gaussians, opt, background, pipe, config)
            converged = bool(converged)
            l1 = float(pkg["tracking_l1"])
            step_norm = pkg.get("tracking_step_norm")
        else:
            fsa = None if fsa_fn is None else fsa_fn(itr - first_order_iters)
            l1_t, x, _, _, pkg = tracking_step_second_order(
                viewpoint, gaussians, background, lambda_rule, repeat_dim, stack_dim, sketch_dim, pipe, config,
                generator, fused_solve=True, fsa=fsa, return_pkg=True)
            l1 = float(l1_t)
            step_norm = x.norm()
            converged = bool(step_norm < second_order_converged_threshold)
            if converged:                        # the converged step is never assigned (:699-706)
                state.assign(viewpoint)
        it += 1
        if trace is not None:
            trace.append((l1, None if step_norm is None else float(step_norm), converged))
        if l1 < best_l1:
            best_l1, best_state, best_pkg, best_it = l1, state, pkg, itr
        if converged:
            break
    if use_best_loss and best_state is not None:
        best_state.assign(viewpoint)
        pkg = best_pkg
    return pkg, best_l1, best_it, it


def mapping_step(window: List[ViewCamera], gaussians, gaussian_optimizer, keyframe_optimizer,
                 background, pipe=Pipe, config=DEFAULT_CONFIG, pose_window=3, bucket=None,
                 fused_loss=False, window_indices: Optional[List[int]] = None, render_fn=None):
    """One mapping iteration over the keyframe window (slam_backend.py:171-332): render
    every view, sum the mapping losses (+ isotropic-scale regulariser), ONE backward, the
    densification statistics, optimiser steps and update_pose of the first pose_window
    keyframes.

    Keyframe-parallel form: with `bucket` (monogs_amd.parallel.FlatGradBucket)
    `window` is the LOCAL shard of the views, `window_indices` their positions in the global
    window (default: rank, rank + world, ...), and gradients / statistics are all-reduced before
    the optimiser step.  The view-independent regulariser (slam_backend.py:244-246) is added on
    rank 0 only - the all-reduce SUMS the ranks' gradients, so adding it everywhere would count
    it world-size times - and the update_pose gate (:328-332) uses the GLOBAL window position."""
    rank, world = 0, 1
    if bucket is not None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
    if window_indices is None:
        window_indices = [rank + world * i for i in range(len(window))] if bucket is not None else list(range(len(window)))
    render_fn = render if render_fn is None else render_fn
    loss = 0.0
    pkgs = []
    for vp in window:
        pkg = render_fn(vp, gaussians, pipe, background)
        if fused_loss:
            from .tracking_fused import mapping_loss
            loss = loss + mapping_loss(config, pkg["render"], pkg["depth"], vp)
        else:
            loss = loss + get_loss_mapping(config, pkg["render"], pkg["depth"], vp, pkg["opacity"])
        pkgs.append(pkg)
    if rank == 0:
        scaling = gaussians.get_scaling
        loss = loss + 10 * torch.abs(scaling - scaling.mean(dim=1, keepdim=True)).mean()
    gaussian_optimizer.zero_grad(set_to_none=True)
    if keyframe_optimizer is not None:
        keyframe_optimizer.zero_grad(set_to_none=True)
    if torch.is_tensor(loss):
        loss.backward()
    with torch.no_grad():
        N = gaussians.get_xyz.shape[0]
        grad_norm = torch.zeros(N, device=gaussians.get_xyz.device)
        denom = torch.zeros_like(grad_norm)
        max_radii = torch.zeros(N, dtype=torch.int32, device=grad_norm.device)
        for pkg in pkgs:   # add_densification_stats (gaussian_model.py:693-697)
            vis = pkg["visibility_filter"]
            g2 = torch.linalg.norm(pkg["viewspace_points"].grad[:, :2], dim=-1)
            grad_norm += torch.where(vis, g2, torch.zeros_like(g2))
            denom += vis.float()
            max_radii = torch.maximum(max_radii, pkg["radii"].to(torch.int32))
        if bucket is not None:
            for p in bucket.params:            # a rank without views still takes part in the sum
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            grad_norm, denom, max_radii = bucket.all_reduce_stats(grad_norm, denom, max_radii)
        gaussian_optimizer.step()
        if keyframe_optimizer is not None:
            keyframe_optimizer.step()
        for vp, gi in zip(window, window_indices):
            if gi < pose_window and vp.uid != 0:
                update_pose(vp)
    loss = loss.detach() if torch.is_tensor(loss) else torch.zeros(())
    return loss, grad_norm, denom, max_radii
