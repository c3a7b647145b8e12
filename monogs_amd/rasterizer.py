"""Host-side mirror of the `diff_gaussian_rasterization` Python package that MonoGS
imports (`gaussian_splatting/gaussian_renderer/__init__.py:15-18` in the reference
tree): `GaussianRasterizationSettings`, `GaussianRasterizer`, and the autograd
Function underneath.  Same names, argument meaning and error behaviour; all compute
goes through the C ABI in include/monogs_raster.h (HIP kernels for gfx950).

PyTorch is used for device memory (caching allocator), the current HIP stream and
autograd bookkeeping only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _cabi


class GaussianRasterizationSettings(NamedTuple):
    """Field order as constructed at gaussian_renderer/__init__.py:61-75."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    projmatrix_raw: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# pair-capacity high-water mark per device: the blend stage is launched optimistically
# with this capacity while the host waits (concurrently) for the true pair count.
# Backward treatment of the EWA field-of-view clamp (include/monogs_raster.h: clamp_gradient_mode):
# "upstream" (default: stop-gradient through the clamp, as the absent CUDA extension is believed to
# behave - the gradients the north star's tolerance is stated against) or "exact" (the derivative of
# the forward as autograd would give it).  Only splats centred outside 1.3x the field of view differ.
CLAMP_GRADIENT_MODES = {"upstream": 0, "exact": 1}
_clamp_gradient_mode = CLAMP_GRADIENT_MODES["upstream"]


def set_clamp_gradient_mode(mode: str) -> None:
    global _clamp_gradient_mode
    _clamp_gradient_mode = CLAMP_GRADIENT_MODES[mode]


_capacity_hint: dict = {}
_tile_max_hint: dict = {}
_bwd_scratch: dict = {}
_sketch_scratch: dict = {}
last_stats: dict = {}


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32c(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _round_cap(n: int) -> int:
    return max(1024, (int(n * 1.25) + 1023) // 1024 * 1024)


_sizes_cache: dict = {}


def _sizes(shape: "_cabi.RasterShape"):
    """mgs_raster_workspace_query, memoised (pure function of the integer shape)."""
    key = (shape.num_gaussians, shape.width, shape.height, shape.sh_coeffs, shape.pair_capacity)
    sz = _sizes_cache.get(key)
    if sz is None:
        if len(_sizes_cache) > 256:
            _sizes_cache.clear()
        sz = _sizes_cache[key] = _cabi.workspace_sizes(shape)
    return sz


_host_slots: dict = {}


def _host_counter(dev_index):
    """Round-robin pinned int32 slots + events for the asynchronous read of the pair count
    (allocating pinned memory or an event per call costs more than the kernels it guards)."""
    ring = _host_slots.get(dev_index)
    if ring is None:
        ring = _host_slots[dev_index] = {
            "buf": torch.zeros(16, dtype=torch.int32, pin_memory=True),
            "ev": [torch.cuda.Event() for _ in range(8)], "i": 0}
    i = ring["i"]
    ring["i"] = (i + 1) % 8
    return ring["buf"][2 * i:2 * i + 2], ring["ev"][i]      # [0] = D, [1] = pairs in the fullest tile


_SPIN_QUERIES = int(os.environ.get("MGS_EVENT_SPIN", "400"))


def _wait_event(ev):
    """Wait for the pair-count event with a bounded busy-poll first: the host learns D a few microseconds after the
    GPU wrote it instead of after a blocking wait's wake-up (tens of microseconds on a virtualised host), and what is
    left of the step's host work - loss and backward enqueue - has to fit under the ~100 us the GPU still needs for
    the blend stage.  Falls back to the blocking wait after ~0.5 ms of polling (MGS_EVENT_SPIN=0: always block)."""
    q = ev.query
    for _ in range(_SPIN_QUERIES):
        if q():
            return
    ev.synchronize()


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, theta, rho, raster_settings, sketch_mode, sketch_dim, stack_dim,
                sketch_dtau, sketch_indices):
        st = raster_settings
        dev = means3D.device
        if dev.type != "cuda":
            raise RuntimeError(
                "monogs_amd rasteriser runs on the GPU only (HIP kernels, gfx950); "
                f"got tensors on {dev}. There is no CPU fallback.")
        lib = _cabi.lib()
        N = int(means3D.shape[0])
        H, W = int(st.image_height), int(st.image_width)
        means3D_c = _f32c(means3D)
        sh_c = _f32c(sh) if sh is not None and sh.numel() else None
        col_c = _f32c(colors_precomp) if colors_precomp is not None and colors_precomp.numel() else None
        op_c = _f32c(opacities).reshape(-1)
        cov_c = _f32c(cov3Ds_precomp) if cov3Ds_precomp is not None and cov3Ds_precomp.numel() else None
        sc_c = _f32c(scales) if scales is not None and scales.numel() else None
        rot_c = _f32c(rotations) if rotations is not None and rotations.numel() else None
        view = _f32c(st.viewmatrix.to(dev))
        proj = _f32c(st.projmatrix.to(dev))
        praw = _f32c(st.projmatrix_raw.to(dev))
        campos = _f32c(st.campos.to(dev)).reshape(-1)
        bg = _f32c(st.bg.to(dev)).reshape(-1)
        K = int(sh_c.shape[1]) if sh_c is not None else 0

        hint = _capacity_hint.get(dev.index)
        shape = _cabi.RasterShape(N, W, H, int(st.sh_degree), K, int(hint or 0),
                                  float(st.tanfovx), float(st.tanfovy), float(st.scale_modifier))
        sizes = _sizes(shape)
        geom = torch.empty(int(sizes.geom_bytes), dtype=torch.uint8, device=dev)
        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        opacity = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(N, dtype=torch.int32, device=dev)
        n_touched = torch.empty(N, dtype=torch.int32, device=dev)

        a = _cabi.ForwardArgs()
        a.shape = shape
        a.means3D, a.scales, a.rotations = _ptr(means3D_c), _ptr(sc_c), _ptr(rot_c)
        a.cov3D_precomp, a.opacities = _ptr(cov_c), _ptr(op_c)
        a.shs, a.colors_precomp = _ptr(sh_c), _ptr(col_c)
        a.viewmatrix, a.projmatrix, a.projmatrix_raw = _ptr(view), _ptr(proj), _ptr(praw)
        a.campos, a.bg = _ptr(campos), _ptr(bg)
        a.geom, a.bins = _ptr(geom), None
        a.out_color, a.out_depth, a.out_opacity = _ptr(color), _ptr(depth), _ptr(opacity)
        a.radii, a.n_touched = _ptr(radii), _ptr(n_touched)
        stream = _stream_ptr(dev)

        # the pair count D lands directly in a pinned host slot (no copy kernel in the stream)
        host_cnt, ev = _host_counter(dev.index)
        a.pair_count_out = host_cnt.data_ptr()
        # the second sort launch (tiles of 1025..4096 pairs) is skipped while the previous forward on
        # this device had no tile near that size (a surprise is still sorted correctly, in HBM)
        a.big_tile_pass = -1 if _tile_max_hint.get(dev.index, 1 << 30) <= 900 else 0
        _cabi.check(lib.mgs_raster_forward_project(C.byref(a), stream), "mgs_raster_forward_project")
        ev.record(torch.cuda.current_stream(dev))

        def run_blend(cap):
            shape.pair_capacity = cap
            a.shape = shape
            sz = _sizes(shape)
            bins_ = torch.empty(int(sz.bins_bytes), dtype=torch.uint8, device=dev)
            a.bins = _ptr(bins_)
            _cabi.check(lib.mgs_raster_forward_blend(C.byref(a), stream), "mgs_raster_forward_blend")
            return bins_

        bins = None
        retried = False
        prepared = None
        needs_grad = any(ctx.needs_input_grad[:10])
        saved_small = (means3D_c, sh_c, col_c, op_c, sc_c, rot_c, cov_c, view, proj, praw, campos, bg, geom)
        # Everything of the bookkeeping that does not depend on the pair count happens BEFORE the wait for it: the GPU
        # is busy with the projection stage anyway, and what runs after the wait - the caller's loss and the backward
        # enqueue - has to fit under the ~100 us the GPU still needs for the blend stage.
        ctx.raster_settings = st
        ctx.sketch = (int(sketch_mode), int(sketch_dim), int(stack_dim))
        ctx.sketch_indices = sketch_indices
        ctx.repeat_iter = 0
        ctx.op_shape = tuple(opacities.shape)
        ctx.has = (sh_c is not None, col_c is not None, sc_c is not None, rot_c is not None,
                   cov_c is not None)
        ctx.mark_non_differentiable(radii, n_touched)
        # undefined output gradients stay None (otherwise autograd fills three zero tensors -
        # two of them N ints - before every backward)
        ctx.set_materialize_grads(False)
        saved = False
        if hint:
            bins = run_blend(int(hint))      # optimistic: overlaps the host wait below
            if needs_grad:                   # ... and so does the preparation of the backward
                prepared = _RasterizeGaussians._prepare_backward(
                    saved_small + (bins,), st, (N, W, H, int(st.sh_degree), K, int(hint)),
                    (int(sketch_mode), int(sketch_dim), int(stack_dim)))
            ctx.save_for_backward(*saved_small, bins)
            saved = True
        _wait_event(ev)
        D, tile_max = host_cnt.tolist()
        _tile_max_hint[dev.index] = tile_max
        if bins is None or D > shape.pair_capacity:
            retried = bins is not None
            prepared = None                  # sized for the old capacity
            bins = run_blend(_round_cap(D))
            saved = False
        # high-water mark with slow decay, so one unusually heavy view does not pin memory forever
        cap = int(shape.pair_capacity)
        _capacity_hint[dev.index] = max(_round_cap(D), int(cap * 0.97) // 1024 * 1024)
        last_stats.update(pairs=D, capacity=cap, retried=retried, N=N)
        ctx.pairs = D
        ctx.prepared = prepared
        ctx.shape_tuple = (N, W, H, int(st.sh_degree), K, cap)
        if not saved:
            ctx.save_for_backward(*saved_small, bins)
        return color, radii, depth, opacity, n_touched

    @staticmethod
    def _prepare_backward(saved, st, shape_tuple, sketch):
        """Everything of the backward that does not depend on the incoming gradients: output /
        scratch allocations and the argument block.  Called from forward() while the host is
        waiting for the GPU anyway, so that backward() itself is a handful of pointer stores and
        one library call (the host work between the pair-count wait and the backward launch is
        what decides whether the GPU runs dry on a slow host)."""
        (means3D, sh, col, op, sc, rot, cov, view, proj, praw, campos, bg, geom, bins) = saved
        dev = means3D.device
        N, W, H, deg, K, cap = shape_tuple
        shape = _cabi.RasterShape(N, W, H, deg, K, cap, float(st.tanfovx), float(st.tanfovy),
                                  float(st.scale_modifier))
        sizes = _sizes(shape)
        # one allocation for all gradient outputs (views are handed to autograd, which adopts
        # them as .grad without copying) and a per-device scratch reused by every backward on
        # the stream (it is dead once the launch sequence has run)
        ncol = 3 * K if sh is not None else 3
        widths = [3, 3, ncol, 1, 3 if sc is not None else 0, 4 if rot is not None else 0,
                  6 if cov is not None else 0]
        offs, off = [], 0
        for w in widths:                      # every part starts on a 256-B boundary (float4 stores)
            offs.append(off)
            off += (N * w + 63) // 64 * 64
        flat = torch.empty(off + 64, dtype=torch.float32, device=dev)
        parts = [flat[o:o + N * w].view(N, w) if w else None for o, w in zip(offs, widths)]
        skey = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = _bwd_scratch.get(skey)
        if ws is None or ws.numel() < int(sizes.bwd_bytes):
            ws = _bwd_scratch[skey] = torch.empty(int(sizes.bwd_bytes), dtype=torch.uint8, device=dev)
        out = {
            "bwd_ws": ws,
            "g_means3D": parts[0], "g_means2D": parts[1],
            "g_colors": parts[2].view(N, K, 3) if sh is not None else parts[2],
            "g_op": parts[3].view(N), "g_sc": parts[4], "g_rot": parts[5], "g_cov": parts[6],
            "g_tau": flat[off:off + 6],
            "sizes": sizes,
        }
        b = _cabi.BackwardArgs()
        f = b.fwd
        f.shape = shape
        f.means3D, f.scales, f.rotations = _ptr(means3D), _ptr(sc), _ptr(rot)
        f.cov3D_precomp, f.opacities = _ptr(cov), _ptr(op)
        f.shs, f.colors_precomp = _ptr(sh), _ptr(col)
        f.viewmatrix, f.projmatrix, f.projmatrix_raw = _ptr(view), _ptr(proj), _ptr(praw)
        f.campos, f.bg = _ptr(campos), _ptr(bg)
        f.geom, f.bins = _ptr(geom), _ptr(bins)
        b.bwd = _ptr(out["bwd_ws"])
        b.grad_means3D, b.grad_means2D = _ptr(out["g_means3D"]), _ptr(out["g_means2D"])
        b.grad_colors, b.grad_opacities = _ptr(out["g_colors"]), _ptr(out["g_op"])
        b.grad_scales, b.grad_rotations, b.grad_cov3D = _ptr(out["g_sc"]), _ptr(out["g_rot"]), _ptr(out["g_cov"])
        b.grad_tau = _ptr(out["g_tau"])
        b.clamp_gradient_mode = _clamp_gradient_mode
        out["args"] = b
        return out

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_opacity, grad_n_touched):
        saved = ctx.saved_tensors
        means3D = saved[0]
        st = ctx.raster_settings
        dev = means3D.device
        lib = _cabi.lib()
        N, W, H, deg, K, cap = ctx.shape_tuple
        # buffers prepared by forward() serve the first backward; a repeated backward
        # (retain_graph, sketch repeats) must not overwrite gradients already handed out
        pre, ctx.prepared = getattr(ctx, "prepared", None), None
        if pre is None:
            pre = _RasterizeGaussians._prepare_backward(saved, st, ctx.shape_tuple, ctx.sketch)
        b = pre["args"]
        gc = _f32c(grad_color) if grad_color is not None else torch.zeros(3, H, W, device=dev)
        gd = _f32c(grad_depth) if grad_depth is not None else None
        b.grad_color, b.grad_depth = _ptr(gc), _ptr(gd)
        b.pair_count_bound = int(ctx.pairs)       # D of this forward (read from the pinned slot): sizes the blend grid
        sketch_mode, sketch_dim, stack_dim = ctx.sketch
        g_sketch = None
        keep = []
        if sketch_mode != 0:
            idx_all = ctx.sketch_indices
            idx = idx_all[ctx.repeat_iter].contiguous()   # [stack,H,W] int32
            ctx.repeat_iter += 1
            g_sketch = torch.empty(stack_dim, sketch_dim, 6, dtype=torch.float32, device=dev)
            # per-(device, stream) scratch like the backward's own: it is dead once the launch sequence has run,
            # needs no initial state, and at ~100 B per pair of capacity is too large to allocate per repeat
            skey = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
            sk_ws = _sketch_scratch.get(skey)
            if sk_ws is None or sk_ws.numel() < int(pre["sizes"].sketch_bytes):
                sk_ws = _sketch_scratch[skey] = torch.empty(int(pre["sizes"].sketch_bytes), dtype=torch.uint8, device=dev)
            keep += [idx, sk_ws]
            b.sketch_mode, b.sketch_dim, b.stack_dim = sketch_mode, sketch_dim, stack_dim
            b.sketch_indices, b.grad_sketch_dtau, b.sketch_ws = _ptr(idx), _ptr(g_sketch), _ptr(sk_ws)
        _cabi.check(lib.mgs_raster_backward(C.byref(b), _stream_ptr(dev)), "mgs_raster_backward")
        if st.debug:      # as upstream: device-side checks only in debug mode (they cost a host sync)
            off = int(pre["sizes"].off_counters)
            short = int(saved[12][off + 8:off + 12].view(torch.int32).item())
            if short:
                raise RuntimeError(f"mgs_raster_backward: pair_count_bound {int(ctx.pairs)} is below the forward's pair "
                                   f"count - the blend grid left some of the {short} work items unwalked")

        has_sh, has_col, _, _, _ = ctx.has
        g_colors, g_tau = pre["g_colors"], pre["g_tau"]
        return (pre["g_means3D"], pre["g_means2D"], g_colors if has_sh else None,
                g_colors if has_col else None, pre["g_op"].reshape(ctx.op_shape), pre["g_sc"], pre["g_rot"],
                pre["g_cov"], g_tau[3:], g_tau[:3], None, None, None, None, g_sketch, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, theta, rho, raster_settings, sketch_mode=0, sketch_dim=0,
                        stack_dim=0, sketch_dtau=None, sketch_indices=None):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales,
                                     rotations, cov3Ds_precomp, theta, rho, raster_settings,
                                     sketch_mode, sketch_dim, stack_dim, sketch_dtau,
                                     sketch_indices)


class GaussianRasterizer(nn.Module):
    """Callable with the 16 keyword arguments used at
    gaussian_renderer/__init__.py:151-168; returns
    (rendered_image, radii, depth, opacity, n_touched)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test of the extension's `markVisible` (unused by MonoGS): the near-plane
        test the rasteriser itself applies, p_view.z > 0.2."""
        with torch.no_grad():
            V = self.raster_settings.viewmatrix.to(positions.device, torch.float32)
            z = positions.float() @ V[:3, 2] + V[3, 2]
            return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, theta=None, rho=None,
                num_backward_gaussians=-1, sketch_mode=0, sketch_dim=0, stack_dim=0,
                sketch_dtau=None, sketch_indices=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if num_backward_gaussians is not None and int(num_backward_gaussians) >= 0:
            # The fork's extension takes this keyword (gaussian_renderer/__init__.py:143,162; RGN.*.num_backward_gaussians,
            # -1 in every shipped configuration, and slam_frontend.py:493-495 never forwards it).  What a non-negative
            # value does lives in the absent CUDA source: refuse it loudly rather than ignore it silently.
            raise NotImplementedError(
                f"num_backward_gaussians={num_backward_gaussians}: only -1 (all Gaussians take part in the backward) is "
                "implemented; the semantics of a limit are defined in the absent diff-gaussian-rasterization-w-pose source")
        dev = means3D.device
        if theta is None:
            theta = torch.zeros(3, dtype=torch.float32, device=dev)
        if rho is None:
            rho = torch.zeros(3, dtype=torch.float32, device=dev)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales,
                                   rotations, cov3D_precomp, theta, rho, self.raster_settings,
                                   sketch_mode, sketch_dim, stack_dim, sketch_dtau, sketch_indices)
