"""Keyframe-parallel mapping over one process per GPU.

The reference has no distributed code (one GPU, `slam.py:109-117`); its mapping iteration
is a sum of independent per-view losses over the keyframe window followed by one
backward (`utils/slam_backend.py:183-247`).  That sum shards by view: Gaussians are
replicated, view i of the window goes to rank i, and the only exchange step is ONE
all-reduce(sum) over RCCL/xGMI of a flat fp32 buffer holding the Gaussian parameter
gradients (xyz 3 + f_dc 3K + opacity 1 + scale 3 + rot 4) plus the two densification
statistics of `gaussian_model.py:693-697` (||means2D.grad[:, :2]|| and the visibility
count), followed by an all-reduce(max) of the radii (`max_radii2D`, slam_backend.py:292-296).
Per-view parameters (pose deltas, exposure) stay on the owning rank.
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.distributed as dist

from .pose import SE3_exp


def view_pose(i: int) -> torch.Tensor:
    """Deterministic T_w2c of synthetic keyframe i: a small orbit around the first view."""
    if i == 0:
        return torch.eye(4)
    a = 2.0 * math.pi * i / 8.0
    tau = torch.tensor([0.06 * math.cos(a), 0.04 * math.sin(a), 0.02 * (i % 3 - 1),
                        0.010 * math.sin(a), 0.012 * math.cos(a), 0.004 * i])
    return SE3_exp(tau)


class FlatGradBucket:
    """One contiguous fp32 buffer = [param grads..., grad-norm stat, visibility stat]; a
    single collective per iteration (xGMI is point-to-point: one large message per step
    beats several small ones)."""

    def __init__(self, params: Sequence[torch.Tensor]):
        self.params = list(params)
        self.N = int(self.params[0].shape[0])
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params] + [self.N, self.N]
        self.flat = torch.empty(sum(self.sizes), dtype=torch.float32, device=dev)
        self.radii = torch.empty(self.N, dtype=torch.int32, device=dev)

    def pack(self, means2D_grad: torch.Tensor, radii: torch.Tensor) -> None:
        """Single local view: statistics derived from its means2D.grad and radii.  On the GPU
        one HIP launch (mgs_pack_mapping_grads) instead of ~12 PyTorch kernels."""
        if self.flat.is_cuda:
            import ctypes as C
            from . import _cabi
            grads = [p.grad.contiguous() for p in self.params]
            m2d = means2D_grad.contiguous()
            rad = radii.to(torch.int32).contiguous()
            assert all(g.dtype == torch.float32 for g in grads) and m2d.dtype == torch.float32
            ptrs = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
            nums = (C.c_int64 * len(grads))(*[g.numel() for g in grads])
            stream = C.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream)
            _cabi.check(_cabi.lib().mgs_pack_mapping_grads(ptrs, nums, len(grads), m2d.data_ptr(), rad.data_ptr(),
                                                           self.N, self.flat.data_ptr(), self.radii.data_ptr(),
                                                           stream), "mgs_pack_mapping_grads")
            return
        vis = radii > 0
        g2 = torch.linalg.norm(means2D_grad[:, :2], dim=-1)
        self.pack_stats(torch.where(vis, g2, torch.zeros_like(g2)), vis.to(torch.float32), radii)

    def pack_stats(self, grad_norm: torch.Tensor, denom: torch.Tensor, radii: torch.Tensor) -> None:
        parts = [p.grad.reshape(-1) for p in self.params] + [grad_norm, denom]
        torch.cat(parts, out=self.flat)
        self.radii.copy_(radii)

    def unpack(self):
        """Reduced gradients as views into the flat buffer + (grad_norm_sum, denom, max_radii)."""
        out, off = [], 0
        for p, n in zip(self.params, self.sizes):
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        stat = self.flat[off:off + self.N]
        denom = self.flat[off + self.N:off + 2 * self.N]
        return stat, denom, self.radii

    def all_reduce_stats(self, grad_norm, denom, radii, group=None):
        """Several local views already folded into (grad_norm, denom, max radii)."""
        self.pack_stats(grad_norm, denom, radii)
        return self._reduce(group)

    def all_reduce(self, means2D_grad: torch.Tensor, radii: torch.Tensor, group=None):
        self.pack(means2D_grad, radii)
        return self._reduce(group)

    force_collective = False      # run the collectives even in a 1-rank group (backend smoke tests)

    def _reduce(self, group=None):
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self.force_collective):
            if dist.get_backend(group) == "gloo" and self.flat.is_cuda:
                # CPU rehearsal of the exchange (tests / single-GPU dry runs): stage through
                # host memory; the production path is RCCL on device buffers below
                f, r = self.flat.cpu(), self.radii.cpu()
                dist.all_reduce(f, op=dist.ReduceOp.SUM, group=group)
                dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
                self.flat.copy_(f)
                self.radii.copy_(r)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                dist.all_reduce(self.radii, op=dist.ReduceOp.MAX, group=group)
        return self.unpack()


def all_gather_visibility(n_touched: torch.Tensor, group=None) -> torch.Tensor:
    """Occlusion-aware visibility of every view of the window on every rank
    (utils/slam_backend.py:251-255: `occ_aware_visibility[kf] = n_touched > 0`), needed on prune
    iterations for `n_obs` (:262-265).  Each rank contributes its own view's mask, bit-packed
    (N / 8 bytes per view over the wire); returns bool [world, N] in rank order."""
    vis = n_touched > 0
    N = int(vis.shape[0])
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return vis[None]
    world = dist.get_world_size(group)
    pad = (-N) % 8
    bits = torch.cat([vis, vis.new_zeros(pad)]) if pad else vis
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=vis.device)
    packed = (bits.view(-1, 8).to(torch.uint8) * weights).sum(dim=1).to(torch.uint8)
    staged = dist.get_backend(group) == "gloo" and packed.is_cuda      # CPU rehearsal of the exchange
    send = packed.cpu() if staged else packed
    out = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(out, send, group=group)
    allp = torch.stack(out).to(vis.device)
    unpacked = ((allp[:, :, None] >> torch.arange(8, device=vis.device, dtype=torch.uint8)) & 1).bool()
    return unpacked.reshape(world, -1)[:, :N]


def observation_counts(visibility: torch.Tensor) -> torch.Tensor:
    """n_obs of slam_backend.py:262-265: in how many views of the window each Gaussian is seen."""
    return visibility.to(torch.int32).sum(dim=0)


def broadcast_split_noise(num_selected: int, device, generator=None, src: int = 0, group=None) -> torch.Tensor:
    """The random offsets of densify_and_split (gaussian_model.py:608-609) must be identical on
    every rank or the replicated maps diverge: rank `src` draws the [2 * n_selected, 3] unit
    normals and broadcasts them (pass the result as `noise=` to map_update.densify_and_prune)."""
    noise = torch.empty(2 * num_selected, 3, device=device)
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    if not distributed or dist.get_rank(group) == src:
        noise = torch.randn(2 * num_selected, 3, device=device, generator=generator)
    if distributed:
        if dist.get_backend(group) == "gloo" and noise.is_cuda:
            tmp = noise.cpu()
            dist.broadcast(tmp, src=src, group=group)
            noise = tmp.to(device)
        else:
            dist.broadcast(noise, src=src, group=group)
    return noise
