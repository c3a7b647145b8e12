"""Fused tracking-loop glue (HIP): the monocular tracking objective and the pose
optimiser step + update_pose, each a couple of launches instead of ~100 PyTorch kernels.

Semantics are those of the reference's PyTorch code (checked in tests against
monogs_amd/losses.py and torch.optim.Adam + monogs_amd/pose.update_pose):
  * loss     utils/slam_utils.py:188-205 (+ Huber :58-75, p-norm slam_frontend.py:596-600)
  * optimise utils/slam_frontend.py:364-392,606-615 and utils/pose_utils.py:88-98
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _cabi


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class _TrackingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, opacity, gt, mask, exposure_a, exposure_b, exposure_eps, huber_delta, pnorm=2.0):
        dev = image.device
        if dev.type != "cuda":
            raise RuntimeError("fused tracking loss runs on the GPU only; use monogs_amd.losses on CPU")
        lib = _cabi.lib()
        image_c = image.detach().float().contiguous()
        opa_c = opacity.detach().float().contiguous()
        gt_c = gt.detach().float().contiguous()
        mask_c = None if mask is None else mask.detach().float().contiguous()
        HW = int(image_c.shape[-1] * image_c.shape[-2])
        partial = torch.empty(int(lib.mgs_tracking_loss_partial_count(HW)), dtype=torch.float32, device=dev)
        scalars = torch.empty(2, dtype=torch.float32, device=dev)
        a = _cabi.TrackingLossArgs()
        a.image, a.opacity, a.gt = image_c.data_ptr(), opa_c.data_ptr(), gt_c.data_ptr()
        a.mask = None if mask_c is None else mask_c.data_ptr()
        a.exposure_a, a.exposure_b = exposure_a.data_ptr(), exposure_b.data_ptr()
        a.exposure_eps, a.huber_delta, a.num_pixels = float(exposure_eps), float(huber_delta), HW
        a.pnorm = float(pnorm)
        a.partial, a.scalars = partial.data_ptr(), scalars.data_ptr()
        _cabi.check(lib.mgs_tracking_loss_forward(C.byref(a), _stream(dev)), "mgs_tracking_loss_forward")
        ctx.save_for_backward(image_c, opa_c, gt_c, mask_c, exposure_a, exposure_b, partial, scalars)
        ctx.consts = (float(exposure_eps), float(huber_delta), HW, float(pnorm))
        return scalars[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        if grad_out is None:
            return (None,) * 9
        image_c, opa_c, gt_c, mask_c, exposure_a, exposure_b, partial, scalars = ctx.saved_tensors
        dev = image_c.device
        lib = _cabi.lib()
        eps, delta, HW, pnorm = ctx.consts
        go = grad_out.detach().float().reshape(1).contiguous()
        g_img = torch.empty_like(image_c)
        g_a = torch.empty(1, dtype=torch.float32, device=dev)
        g_b = torch.empty(1, dtype=torch.float32, device=dev)
        a = _cabi.TrackingLossArgs()
        a.image, a.opacity, a.gt = image_c.data_ptr(), opa_c.data_ptr(), gt_c.data_ptr()
        a.mask = None if mask_c is None else mask_c.data_ptr()
        a.exposure_a, a.exposure_b = exposure_a.data_ptr(), exposure_b.data_ptr()
        a.exposure_eps, a.huber_delta, a.num_pixels = eps, delta, HW
        a.pnorm = pnorm
        a.partial, a.scalars = partial.data_ptr(), scalars.data_ptr()
        a.grad_out, a.grad_image = go.data_ptr(), g_img.data_ptr()
        a.grad_a, a.grad_b = g_a.data_ptr(), g_b.data_ptr()
        _cabi.check(lib.mgs_tracking_loss_backward(C.byref(a), _stream(dev)), "mgs_tracking_loss_backward")
        return (g_img, None, None, None, g_a.reshape(exposure_a.shape), g_b.reshape(exposure_b.shape),
                None, None, None)


def tracking_loss(image, opacity, viewpoint, huber_delta=0.01, pnorm=2.0):
    """|| Huber( opacity * mask * ((|a|+eps) image + b - gt) ) ||_p for `viewpoint`
    (attributes original_image, rgb_pixel_mask_mapping, exposure_a/b/eps).  The reference uses p = 2 with
    Huber and RGN.pnorm without (slam_frontend.py:596-600): see slam_loops.tracking_norm."""
    mask = viewpoint.rgb_pixel_mask_mapping
    return _TrackingLoss.apply(image, opacity, viewpoint.original_image, mask, viewpoint.exposure_a,
                               viewpoint.exposure_b, viewpoint.exposure_eps, huber_delta, pnorm)


class FusedPoseOptimizer:
    """Adam on (cam_rot_delta, cam_trans_delta, exposure_a, exposure_b) + update_pose in ONE
    launch.  `step()` returns a device int32 flag tensor (1 = converged): reading it is the
    caller's (optional) host sync."""

    def __init__(self, viewpoint, lr_rot=0.003, lr_trans=0.001, lr_a=0.02, lr_b=0.02,
                 betas=(0.9, 0.999), eps=1e-8, converged_threshold=1e-4):
        self.vp = viewpoint
        dev = viewpoint.cam_rot_delta.device
        self.exp_avg = torch.zeros(8, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(8, dtype=torch.float32, device=dev)
        self.converged = torch.zeros(1, dtype=torch.int32, device=dev)
        self.t = 0
        self.hp = (lr_rot, lr_trans, lr_a, lr_b, betas[0], betas[1], eps, converged_threshold)

    def zero_grad(self):
        for p in (self.vp.cam_rot_delta, self.vp.cam_trans_delta, self.vp.exposure_a, self.vp.exposure_b):
            p.grad = None

    def step(self, update_pose=True):
        vp = self.vp
        dev = vp.cam_rot_delta.device
        self.t += 1
        a = _cabi.PoseAdamArgs()
        a.cam_rot_delta, a.cam_trans_delta = vp.cam_rot_delta.data_ptr(), vp.cam_trans_delta.data_ptr()
        a.exposure_a, a.exposure_b = vp.exposure_a.data_ptr(), vp.exposure_b.data_ptr()
        g = [p.grad for p in (vp.cam_rot_delta, vp.cam_trans_delta, vp.exposure_a, vp.exposure_b)]
        g = [None if x is None else x.contiguous() for x in g]
        a.grad_rot, a.grad_trans, a.grad_a, a.grad_b = [None if x is None else x.data_ptr() for x in g]
        a.exp_avg, a.exp_avg_sq = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        if update_pose:
            assert vp.T.is_contiguous() and vp.T.dtype == torch.float32
            a.T = vp.T.data_ptr()
        a.converged = self.converged.data_ptr()
        a.step = self.t
        (a.lr_rot, a.lr_trans, a.lr_a, a.lr_b, a.beta1, a.beta2, a.eps, a.converged_threshold) = self.hp
        _cabi.check(_cabi.lib().mgs_pose_adam_step(C.byref(a), _stream(dev)), "mgs_pose_adam_step")
        return self.converged


def lm_solve_step(SJ: torch.Tensor, Sf: torch.Tensor, lambda_: float, viewpoint=None) -> torch.Tensor:
    """x = argmin ||[SJ; sqrt(lambda) I] x + [Sf; 0]|| (8 unknowns) in one launch; with
    `viewpoint`, also T <- Exp(x[:6]) T and exposure += x[6:8] (slam_frontend.py:672-697)."""
    dev = SJ.device
    SJc = SJ.detach().float().reshape(-1, 8).contiguous()
    Sfc = Sf.detach().float().reshape(-1).contiguous()
    x = torch.empty(8, dtype=torch.float32, device=dev)
    a = _cabi.LMStepArgs()
    a.SJ, a.Sf, a.rows, a.lam = SJc.data_ptr(), Sfc.data_ptr(), int(SJc.shape[0]), float(lambda_)
    if viewpoint is not None:
        assert viewpoint.T.is_contiguous() and viewpoint.T.dtype == torch.float32
        a.T = viewpoint.T.data_ptr()
        a.exposure_a, a.exposure_b = viewpoint.exposure_a.data_ptr(), viewpoint.exposure_b.data_ptr()
    a.x_out = x.data_ptr()
    _cabi.check(_cabi.lib().mgs_lm_solve_step(C.byref(a), _stream(dev)), "mgs_lm_solve_step")
    return x


_loss_scratch: dict = {}


def _zeroed_partials(dev, count):
    """Per-(device, stream) scratch for the loss kernels' block partials, zeroed ONCE: the forward's
    last-workgroup ticket lives behind the partials and every call restores it to zero, so calls
    enqueued in order on one stream can share the buffer."""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, count)
    buf = _loss_scratch.get(key)
    if buf is None:
        if len(_loss_scratch) > 64:
            _loss_scratch.clear()
        buf = _loss_scratch[key] = torch.zeros(count, dtype=torch.float32, device=dev)
    return buf


class _MappingLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, depth, gt, gt_depth, mask, exposure_a, exposure_b, exposure_eps, w_rgb,
                w_depth, depth_mask_threshold, apply_exposure):
        dev = image.device
        if dev.type != "cuda":
            raise RuntimeError("fused mapping loss runs on the GPU only; use monogs_amd.losses on CPU")
        lib = _cabi.lib()
        f = lambda t: None if t is None else t.detach().float().contiguous()
        image_c, depth_c, gt_c, gtd_c, mask_c = f(image), f(depth), f(gt), f(gt_depth), f(mask)
        HW = int(image_c.shape[-1] * image_c.shape[-2])
        partial = _zeroed_partials(dev, int(lib.mgs_tracking_loss_partial_count(HW)))
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        a = _cabi.MappingLossArgs()
        ptr = lambda t: None if t is None else t.data_ptr()
        a.image, a.gt, a.mask, a.depth, a.gt_depth = ptr(image_c), ptr(gt_c), ptr(mask_c), ptr(depth_c), ptr(gtd_c)
        a.exposure_a, a.exposure_b = ptr(exposure_a), ptr(exposure_b)
        a.exposure_eps, a.w_rgb, a.w_depth = float(exposure_eps), float(w_rgb), float(w_depth)
        a.depth_mask_threshold, a.apply_exposure, a.num_pixels = float(depth_mask_threshold), int(apply_exposure), HW
        a.partial, a.loss = partial.data_ptr(), loss.data_ptr()
        a.partial_ticket_ready = 1          # one launch: the last workgroup finishes the sum
        _cabi.check(lib.mgs_mapping_loss_forward(C.byref(a), _stream(dev)), "mgs_mapping_loss_forward")
        ctx.save_for_backward(image_c, depth_c, gt_c, gtd_c, mask_c, exposure_a, exposure_b, partial)
        ctx.consts = (float(exposure_eps), float(w_rgb), float(w_depth), float(depth_mask_threshold),
                      int(apply_exposure), HW, depth is not None)
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        if grad_out is None:
            return (None,) * 12
        image_c, depth_c, gt_c, gtd_c, mask_c, exposure_a, exposure_b, partial = ctx.saved_tensors
        eps, w_rgb, w_depth, thr, apply_exposure, HW, has_depth = ctx.consts
        dev = image_c.device
        lib = _cabi.lib()
        go = grad_out.detach().float().reshape(1).contiguous()
        g_img = torch.empty_like(image_c)
        g_dep = torch.empty_like(depth_c) if has_depth else None
        g_a = torch.empty(1, dtype=torch.float32, device=dev) if apply_exposure else None
        g_b = torch.empty(1, dtype=torch.float32, device=dev) if apply_exposure else None
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        a = _cabi.MappingLossArgs()
        ptr = lambda t: None if t is None else t.data_ptr()
        a.image, a.gt, a.mask, a.depth, a.gt_depth = ptr(image_c), ptr(gt_c), ptr(mask_c), ptr(depth_c), ptr(gtd_c)
        a.exposure_a, a.exposure_b = ptr(exposure_a), ptr(exposure_b)
        a.exposure_eps, a.w_rgb, a.w_depth = eps, w_rgb, w_depth
        a.depth_mask_threshold, a.apply_exposure, a.num_pixels = thr, apply_exposure, HW
        a.partial, a.loss, a.grad_out = partial.data_ptr(), loss.data_ptr(), go.data_ptr()
        a.grad_image, a.grad_depth, a.grad_a, a.grad_b = ptr(g_img), ptr(g_dep), ptr(g_a), ptr(g_b)
        _cabi.check(lib.mgs_mapping_loss_backward(C.byref(a), _stream(dev)), "mgs_mapping_loss_backward")
        ga = None if g_a is None else g_a.reshape(exposure_a.shape)
        gb = None if g_b is None else g_b.reshape(exposure_b.shape)
        return g_img, g_dep, None, None, None, ga, gb, None, None, None, None, None


def mapping_loss(config, image, depth, viewpoint, initialization=False):
    """Fused get_loss_mapping (utils/slam_utils.py:224-253) for `viewpoint`."""
    mono = config["Training"]["monocular"]
    alpha = 1.0 if mono else config["Training"].get("alpha", 0.95)
    return _MappingLoss.apply(image, None if mono else depth, viewpoint.original_image,
                              None if mono else viewpoint.gt_depth, viewpoint.rgb_pixel_mask_mapping,
                              viewpoint.exposure_a, viewpoint.exposure_b, viewpoint.exposure_eps,
                              alpha, 0.0 if mono else 1.0 - alpha, 0.01, 0 if initialization else 1)


def l1_image_depth_loss(image, depth, gt_image, gt_depth, w_depth=0.05):
    """mean|image - gt| + w_depth * mean|depth - gt_depth| (the synthetic benchmark loss of
    BASELINE.md §4) through the same fused kernels."""
    return _MappingLoss.apply(image, depth, gt_image, gt_depth, None, None, None, 0.0, 1.0, w_depth,
                              -1.0, 0)


def l1_image_depth_loss_backward(image, depth, gt_image, gt_depth, w_depth=0.05, *, mask=None, viewpoint=None,
                                 w_rgb=1.0, depth_mask_threshold=-1.0):
    """loss = w_rgb * mean|m (image' - gt)| + w_depth * mean|dm (depth - gt_depth)| AND its backward in
    ONE kernel launch: value and gradients of an L1 objective do not depend on each other, so
    `mgs_mapping_loss_fused` writes dL/dimage, dL/ddepth next to the block sums in a single pass, and
    the gradients are handed to autograd directly (`torch.autograd.backward((image, depth), grads)` -
    exactly what `loss.backward()` propagates for a scalar loss).  With `viewpoint` the exposure
    (|a| + eps) image + b of utils/slam_utils.py:224-253 is applied and viewpoint.exposure_a/b receive
    their gradients.  Returns the detached loss (a device scalar written by the same launch)."""
    dev = image.device
    if dev.type != "cuda":
        raise RuntimeError("fused loss runs on the GPU only")
    lib = _cabi.lib()
    f = lambda t: None if t is None else t.detach().float().contiguous()
    image_c, depth_c, gt_c, gtd_c, mask_c = f(image), f(depth), f(gt_image), f(gt_depth), f(mask)
    HW = int(image_c.shape[-1] * image_c.shape[-2])
    use_depth = w_depth != 0.0 and depth is not None
    partial = _zeroed_partials(dev, int(lib.mgs_mapping_loss_partial_count(HW)))
    g_img = torch.empty_like(image_c)
    g_dep = torch.empty_like(depth_c) if use_depth else None
    out = torch.empty(3, dtype=torch.float32, device=dev)        # loss, d/da, d/db
    a = _cabi.MappingLossArgs()
    ptr = lambda t: None if t is None else t.data_ptr()
    a.image, a.gt, a.mask = ptr(image_c), ptr(gt_c), ptr(mask_c)
    a.depth, a.gt_depth = (ptr(depth_c), ptr(gtd_c)) if use_depth else (None, None)
    if viewpoint is not None:
        a.exposure_a, a.exposure_b = viewpoint.exposure_a.data_ptr(), viewpoint.exposure_b.data_ptr()
        a.exposure_eps, a.apply_exposure = float(viewpoint.exposure_eps), 1
        a.grad_a, a.grad_b = out[1:].data_ptr(), out[2:].data_ptr()
    a.w_rgb, a.w_depth = float(w_rgb), float(w_depth) if use_depth else 0.0
    a.depth_mask_threshold, a.num_pixels = float(depth_mask_threshold), HW
    a.partial, a.grad_image, a.grad_depth = partial.data_ptr(), g_img.data_ptr(), ptr(g_dep)
    a.loss, a.partial_ticket_ready = out.data_ptr(), 1            # the last workgroup finishes the sums
    _cabi.check(lib.mgs_mapping_loss_fused(C.byref(a), None, _stream(dev)), "mgs_mapping_loss_fused")
    tensors, grads = [image], [g_img.view_as(image)]
    if use_depth and depth.requires_grad:
        tensors.append(depth)
        grads.append(g_dep.view_as(depth))
    torch.autograd.backward(tensors, grads)
    if viewpoint is not None:
        for p, g in ((viewpoint.exposure_a, out[1]), (viewpoint.exposure_b, out[2])):
            if p.requires_grad:
                p.grad = g.reshape(p.shape).clone() if p.grad is None else p.grad + g.reshape(p.shape)
    return out[0]
