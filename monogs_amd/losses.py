"""Loss glue of the tracking / mapping inner loops (host side, PyTorch elementwise ops).

Mirrors /root/reference utils/slam_utils.py: HuberLoss (:58-75), ApplyExposure with the
sketched exposure Jacobian (:115-185), get_loss_tracking_per_pixel (:188-205) and
get_loss_mapping (:224-253).  These define which rasteriser outputs receive gradient; they
are not part of the native hot path.  `viewpoint` only needs the attributes used below.
"""
from __future__ import annotations

import torch


class HuberLoss(torch.autograd.Function):
    """Signed pseudo-Huber residual: x inside |x| < delta, sign(x) sqrt(2 delta |x| - delta^2)
    outside (slam_utils.py:58-75)."""

    @staticmethod
    def forward(ctx, x, delta=0.1):
        ctx.delta = delta
        ctx.save_for_backward(x)
        return torch.where(x.abs() < delta, x,
                           torch.sqrt(2 * delta * x.abs() - delta ** 2) * torch.sign(x))

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        delta = ctx.delta
        return torch.where(x.abs() < delta, grad_output,
                           grad_output * delta / torch.sqrt(2 * delta * x.abs() - delta ** 2)), None


class ApplyExposure(torch.autograd.Function):
    """(|a| + eps) * image + b; in sketch mode the backward also bucket-sums the per-pixel
    exposure Jacobian into sketch_dexposure.grad[stack, sketch, 2] for repeat #k of the same
    forward (slam_utils.py:115-185)."""

    @staticmethod
    def forward(ctx, image, exposure_a, exposure_b, exposure_eps, sketch_mode=0, sketch_dim=0,
                stack_dim=0, rand_indices=None, sketch_dexposure=None):
        ctx.sketch_mode, ctx.sketch_dim, ctx.stack_dim = sketch_mode, sketch_dim, stack_dim
        rows = cols = None
        if sketch_mode != 0:
            rows, cols = rand_indices
            ctx.repeat_iter = 0
        ctx.save_for_backward(image, exposure_a, exposure_b, sketch_dexposure, rows, cols)
        return (torch.abs(exposure_a) + exposure_eps) * image + exposure_b

    @staticmethod
    def backward(ctx, grad_output):
        image, a, b, sk, rows, cols = ctx.saved_tensors
        goi = grad_output * image
        grad_image = torch.abs(a) * grad_output
        grad_a = goi.sum().reshape(a.shape)
        grad_b = grad_output.sum().reshape(b.shape)
        grad_sk = None
        if ctx.sketch_mode != 0:
            r, c = rows[ctx.repeat_iter].long(), cols[ctx.repeat_iter].long()
            grad_sk = torch.empty(sk.shape, device=grad_output.device)
            grad_sk[:, :, 0] = goi.sum(0)[r, c].sum(-1)
            grad_sk[:, :, 1] = grad_output.sum(0)[r, c].sum(-1)
            ctx.repeat_iter += 1
        return grad_image, grad_a, grad_b, None, None, None, None, None, grad_sk


def apply_exposure(image, viewpoint):
    return (torch.abs(viewpoint.exposure_a) + viewpoint.exposure_eps) * image + viewpoint.exposure_b


def get_loss_tracking_per_pixel(config, image, depth, opacity, viewpoint, forward_sketch_args=None):
    """Monocular per-pixel tracking residual [3,H,W] (slam_utils.py:188-205); the RGB-D
    variant raises NotImplementedError in the reference (:220) and so does this."""
    if forward_sketch_args is None:
        image_ab = apply_exposure(image, viewpoint)
    else:
        f = forward_sketch_args
        image_ab = ApplyExposure.apply(image, viewpoint.exposure_a, viewpoint.exposure_b,
                                       viewpoint.exposure_eps, f["sketch_mode"], f["sketch_dim"],
                                       f["stack_dim"], f["rand_indices"], f["sketch_dexposure"])
    if not config["Training"]["monocular"]:
        raise NotImplementedError("RGB-D per-pixel tracking loss is not implemented in the reference")
    gt = viewpoint.original_image.to(image.device)
    m = viewpoint.rgb_pixel_mask_mapping
    return opacity * (image_ab * m - gt * m)


def get_loss_mapping(config, image, depth, viewpoint, opacity=None, initialization=False):
    """slam_utils.py:224-253."""
    image_ab = image if initialization else apply_exposure(image, viewpoint)
    gt = viewpoint.original_image.to(image.device)
    m = viewpoint.rgb_pixel_mask_mapping
    l1_rgb = torch.abs(image_ab * m - gt * m)
    if config["Training"]["monocular"]:
        return l1_rgb.mean()
    alpha = config["Training"].get("alpha", 0.95)
    dm = (viewpoint.gt_depth > 0.01).view(*depth.shape)
    l1_depth = torch.abs(depth * dm - viewpoint.gt_depth * dm)
    return alpha * l1_rgb.mean() + (1 - alpha) * l1_depth.mean()
