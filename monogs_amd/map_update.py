"""Map maintenance on the device: fused multi-tensor Adam for the
Gaussian parameters and densify / prune as ~5 launches with ONE host sync.

Mirrors, for a GaussianModel-shaped object (attributes `_xyz, _features_dc, _features_rest,
_opacity, _scaling, _rotation, optimizer, xyz_gradient_accum, denom, max_radii2D,
unique_kfIDs, n_obs, percent_dense`):
  * GaussianModel.training_setup's optimiser      gaussian_model.py:247-285
  * densify_and_prune / prune_points              gaussian_model.py:485-691
of /root/reference/gaussian_splatting/scene/.  The reference rebuilds every tensor with
boolean indexing + torch.cat (dozens of kernels, several implicit host syncs, and CPU-resident
`unique_kfIDs` / `n_obs`); here a plan (stable positions from wave-ballot prefix sums) is
computed once and every tensor is rebuilt by one gather launch.  `unique_kfIDs` and `n_obs`
are kept as device int32 tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn as nn

from . import _cabi

_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
_ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
         "scaling": "_scaling", "rotation": "_rotation"}


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class FusedGaussianAdam:
    """Drop-in for `torch.optim.Adam(l, lr=0.0, eps=1e-15)` as GaussianModel uses it:
    `param_groups` (dicts with "params": [one tensor], "lr", "name"), `state[param]` with
    "step" / "exp_avg" / "exp_avg_sq", `step()`, `zero_grad()`.  One HIP launch per step."""

    def __init__(self, param_groups, lr=0.0, betas=(0.9, 0.999), eps=1e-15):
        self.param_groups = [dict(g) for g in param_groups]
        for g in self.param_groups:
            g.setdefault("lr", lr)
            assert len(g["params"]) == 1, "one tensor per group (gaussian_model.py:252-283)"
        if len(self.param_groups) > _cabi.ADAM_MAX_GROUPS:
            raise ValueError("too many parameter groups")
        self.betas, self.eps = betas, eps
        self.state = {}

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            p = g["params"][0]
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        live = [(g, g["params"][0]) for g in self.param_groups if g["params"][0].grad is not None]
        if not live:
            return
        dev = live[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("FusedGaussianAdam runs on the GPU only (HIP kernel, gfx950)")
        arr = (_cabi.AdamGroup * len(live))()
        keep = []
        for i, (g, p) in enumerate(live):
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
            st["step"] = int(st["step"]) + 1      # per tensor, as torch.optim.Adam counts
            grad = p.grad.contiguous()
            assert p.is_contiguous() and p.dtype == torch.float32 and grad.dtype == torch.float32
            keep.append(grad)
            arr[i].param, arr[i].grad = p.data_ptr(), grad.data_ptr()
            arr[i].exp_avg, arr[i].exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            arr[i].numel, arr[i].lr, arr[i].step = p.numel(), float(g["lr"]), st["step"]
        _cabi.check(_cabi.lib().mgs_adam_step_multi(arr, len(live), self.betas[0], self.betas[1], self.eps,
                                                    _stream(dev)), "mgs_adam_step_multi")


def _rebuild(model, plan: _cabi.MapPlanArgs, totals, noise: Optional[torch.Tensor], generator=None):
    """Apply a counted plan: emit the row map, gather every per-Gaussian tensor, swap the
    new tensors into the model and its optimiser."""
    dev = model._xyz.device
    lib = _cabi.lib()
    n_orig, n_clone, n_par, n_sel = (int(t) for t in totals)
    rows = n_orig + n_clone + 2 * n_par
    src_index = torch.empty(max(rows, 1), dtype=torch.int32, device=dev)
    noise_row = torch.empty(max(2 * n_par, 1), dtype=torch.int32, device=dev)
    plan.src_index, plan.noise_row = src_index.data_ptr(), noise_row.data_ptr()
    _cabi.check(lib.mgs_map_plan_emit(C.byref(plan), _stream(dev)), "mgs_map_plan_emit")
    if n_par > 0:
        if noise is None:
            noise = torch.randn(2 * n_sel, 3, device=dev, generator=generator)   # one draw per selected Gaussian (:608-609)
        noise = noise.to(dev, torch.float32).contiguous()
        assert noise.shape == (2 * n_sel, 3)

    ga = _cabi.MapGatherArgs()
    k = 0
    keep = []

    def add(src, mode):
        nonlocal k
        src = src.detach().contiguous()
        width = 1
        for d in src.shape[1:]:
            width *= int(d)
        dst = torch.empty((rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
        if src.numel() > 0 and rows > 0:
            assert src.element_size() == 4
            t = ga.tensors[k]
            t.src, t.dst, t.width, t.mode = src.data_ptr(), dst.data_ptr(), width, mode
            k += 1
        keep.append((src, dst))
        return dst

    opt = model.optimizer
    new_params, new_states = {}, {}
    modes = {"xyz": _cabi.GATHER_SPLIT_XYZ, "scaling": _cabi.GATHER_SPLIT_SCALING}
    for g in opt.param_groups:
        name = g["name"]
        p = g["params"][0]
        new_params[name] = add(p, modes.get(name, _cabi.GATHER_COPY))
        st = opt.state.get(p)
        if st is not None and "exp_avg" in st:
            new_states[name] = (add(st["exp_avg"], _cabi.GATHER_ZERO_NEW),
                                add(st["exp_avg_sq"], _cabi.GATHER_ZERO_NEW), st)
    kf = model.unique_kfIDs.to(dev, torch.int32)
    nobs = model.n_obs.to(dev, torch.int32)
    new_kf, new_nobs = add(kf, _cabi.GATHER_COPY), add(nobs, _cabi.GATHER_COPY)
    ga.num_tensors, ga.rows, ga.num_children = k, rows, n_par
    ga.src_index = src_index.data_ptr()
    ga.rotations, ga.log_scales = model._rotation.data_ptr(), model._scaling.data_ptr()
    ga.noise = None if n_par == 0 else noise.data_ptr()
    ga.noise_row = noise_row.data_ptr()
    if k > 0:
        _cabi.check(lib.mgs_map_gather(C.byref(ga), _stream(dev)), "mgs_map_gather")

    for g in opt.param_groups:
        name = g["name"]
        old = g["params"][0]
        new = nn.Parameter(new_params[name].requires_grad_(True))
        if name in new_states:
            m, v, st = new_states[name]
            st["exp_avg"], st["exp_avg_sq"] = m, v
            del opt.state[old]
            opt.state[new] = st
        g["params"][0] = new
        setattr(model, _ATTR[name], new)
    model.unique_kfIDs, model.n_obs = new_kf, new_nobs
    return rows, src_index


def _plan(model, dev, n):
    lib = _cabi.lib()
    plan = _cabi.MapPlanArgs()
    plan.n = n
    flags = torch.empty(n, dtype=torch.uint8, device=dev)
    counts = torch.empty(4 * int(lib.mgs_map_plan_blocks(n)), dtype=torch.int32, device=dev)
    totals = torch.empty(4, dtype=torch.int32, device=dev)
    plan.flags, plan.block_counts, plan.totals = flags.data_ptr(), counts.data_ptr(), totals.data_ptr()
    return plan, (flags, counts, totals)


@torch.no_grad()
def densify_and_prune(model, max_grad, min_opacity, extent, max_screen_size, noise=None, generator=None):
    """GaussianModel.densify_and_prune (gaussian_model.py:674-691).  `noise` ([2*n_selected, 3]
    unit normals, one row per split child in the reference's repeat order) replaces the
    reference's torch.normal draw, for reproducible tests; `generator` seeds the draw instead."""
    dev = model._xyz.device
    n = int(model._xyz.shape[0])
    if n == 0:
        return
    plan, keep = _plan(model, dev, n)
    ga = model.xyz_gradient_accum.reshape(-1).contiguous()
    dn = model.denom.reshape(-1).contiguous()
    ls, ol = model._scaling.detach().contiguous(), model._opacity.detach().reshape(-1).contiguous()
    plan.grad_accum, plan.denom = ga.data_ptr(), dn.data_ptr()
    plan.log_scales, plan.opacity_logit = ls.data_ptr(), ol.data_ptr()
    plan.grad_threshold, plan.dense_extent = float(max_grad), float(model.percent_dense * extent)
    plan.min_opacity = float(min_opacity)
    plan.big_extent = float(0.1 * extent) if max_screen_size else -1.0
    _cabi.check(_cabi.lib().mgs_map_plan_count(C.byref(plan), _stream(dev)), "mgs_map_plan_count")
    totals = keep[2].tolist()                      # the one host sync of the rebuild
    rows, _ = _rebuild(model, plan, totals, noise, generator)
    # densification_postfix (:591-594): statistics restart from zero for every Gaussian
    model.xyz_gradient_accum = torch.zeros(rows, 1, device=dev)
    model.denom = torch.zeros(rows, 1, device=dev)
    model.max_radii2D = torch.zeros(rows, device=dev)


@torch.no_grad()
def prune_points(model, mask):
    """GaussianModel.prune_points (gaussian_model.py:540-556): remove rows where mask is True."""
    dev = model._xyz.device
    n = int(model._xyz.shape[0])
    if n == 0:
        return
    plan, keep = _plan(model, dev, n)
    m8 = mask.to(dev).to(torch.uint8).contiguous()
    plan.prune_mask = m8.data_ptr()
    _cabi.check(_cabi.lib().mgs_map_plan_count(C.byref(plan), _stream(dev)), "mgs_map_plan_count")
    totals = keep[2].tolist()
    rows, src_index = _rebuild(model, plan, totals, None)
    idx = (src_index[:rows] & 0x3FFFFFFF).long()
    model.xyz_gradient_accum = model.xyz_gradient_accum[idx]
    model.denom = model.denom[idx]
    model.max_radii2D = model.max_radii2D[idx]
