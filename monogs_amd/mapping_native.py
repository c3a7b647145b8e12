"""Mapping iterations enqueued natively: one C-ABI call per (view, iteration).

Mirror of the loop bodies of /root/reference utils/slam_backend.py - `BackEnd.map` (:157-333),
`initialize_map` (:91-155) and the per-keyframe optimiser set-up of the "keyframe" message
(:437-489) - for a `GaussianModel` (monogs_amd/gaussian_model.py) and a dict of keyframe
cameras (slam_loops.ViewCamera-like).  Queues, the frontend hand-shake, keyframe policy and the
GUI stay out of scope.

Per iteration (cf. the Python body in slam_loops.mapping_step, which costs ~1.5 ms of host time
per VIEW in autograd / allocator / ctypes work):
  mgs_map_activate                       exp / sigmoid / normalize of the raw parameters, once
  mgs_mapping_view_iteration x views     forward, L1 objective + gradients in one pass, backward
                                         chained through the activations and ACCUMULATED into one
                                         flat buffer (which is also the all-reduce buffer), the
                                         densification statistics, occ-aware visibility, and the
                                         view's own pose / exposure Adam + update_pose
  [all_reduce(sum) flat, all_reduce(max) radii]   keyframe-parallel only
  mgs_map_finish_iteration               statistics fold (+ reset_opacity_nonvisible when due)
  densify_and_prune (every gaussian_update_every) / mgs_adam_step_multi (FusedGaussianAdam)
The window's views are sharded round-robin over the ranks; the two random old keyframes of
:215-242 continue the round-robin (with 8 window views on 8 ranks they are a second view on ranks
0 and 1), drawn from a generator seeded identically on every rank.

Deliberate differences from the reference, none of which changes a result it defines:
  * the `prune=True` pass (:259-290) runs forward-only (the reference also runs a backward whose
    gradients it then drops for the Gaussians and leaves, stale, on the camera parameters);
  * the per-view pose / exposure Adam step is taken right after the view's backward instead of
    after the Gaussian step (the keyframe optimiser's groups are per view: the steps commute).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import _cabi
from .gaussian_model import GaussianModel

DEFAULT_MAP_CONFIG = {
    "Training": {
        "monocular": True, "alpha": 0.95, "rgb_boundary_threshold": 0.01,
        "init_itr_num": 1050, "init_gaussian_update": 100, "init_gaussian_reset": 500,
        "init_gaussian_th": 0.005, "init_gaussian_extent": 30, "mapping_itr_num": 150,
        "gaussian_update_every": 150, "gaussian_update_offset": 50, "gaussian_th": 0.7,
        "gaussian_extent": 1.0, "gaussian_reset": 2001, "size_threshold": 20, "window_size": 8,
        "pose_window": 3, "prune_mode": "slam",
        "lr": {"cam_rot_delta": 0.003, "cam_trans_delta": 0.001, "exposure_a": 0.02, "exposure_b": 0.02},
    },
    "opt_params": {"densify_grad_threshold": 0.0002, "densify_from_iter": 500},
}


def _align(n, q=64):
    return (n + q - 1) // q * q


class _ViewState:
    """Device-side state of one keyframe inside the mapper: its optimiser moments, camera
    matrices and the float mask the loss kernel reads."""

    def __init__(self, vp, dev):
        self.exp_avg = torch.zeros(8, device=dev)
        self.exp_avg_sq = torch.zeros(8, device=dev)
        self.step = 0
        self.view = torch.empty(4, 4, device=dev)
        self.full = torch.empty(4, 4, device=dev)
        self.matrices_fresh = False
        self.T_ptr = 0
        self.loss = torch.zeros(1, device=dev)
        self.refresh(vp, dev)

    def refresh(self, vp, dev):
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        self.gt = f32(vp.original_image)
        m = getattr(vp, "rgb_pixel_mask_mapping", None)
        self.mask = None if m is None else f32(m)
        d = getattr(vp, "gt_depth", None)
        self.gt_depth = None if d is None else f32(torch.as_tensor(d)).reshape(1, *self.gt.shape[-2:])
        self.proj = f32(vp.projection_matrix)


class _Lane:
    """Everything one in-flight view needs for itself: a HIP stream, the rasteriser workspaces and
    outputs, its accumulation buffer and its pair-count slots.  The views of a mapping iteration
    are independent until the optimiser step, so NativeMapper keeps `concurrent_views` of them in
    flight on different streams: the latency-bound front-end kernels and the low-occupancy tails of
    the blend kernels of one view overlap with the other view's work."""

    def __init__(self, dev, stream):
        self.dev, self.stream = dev, stream            # stream None: torch's current stream
        self.host_D = torch.zeros(2, dtype=torch.int32).pin_memory()      # [0] = D, [1] = fullest tile
        self.d_max = torch.zeros(1, dtype=torch.int32, device=dev)
        self.loss_accum = torch.zeros(1, device=dev)
        self.shape_key, self.cap_alloc = None, 0
        self.flat = self.radii_max = self.radii = self.n_touched = None
        self.event = torch.cuda.Event()

    def torch_stream(self):
        return self.stream if self.stream is not None else torch.cuda.current_stream(self.dev)

    def stream_ptr(self):
        return C.c_void_p(self.torch_stream().cuda_stream)


class NativeMapper:
    def __init__(self, gaussians: GaussianModel, background, config: Optional[dict] = None,
                 cameras_extent: float = 6.0, group=None, capacity_margin: float = 1.5, seed: int = 0,
                 concurrent_views: int = 2):
        cfg = {k: dict(v) for k, v in DEFAULT_MAP_CONFIG.items()}
        for k, v in (config or {}).items():
            if isinstance(v, dict):
                cfg.setdefault(k, {}).update(v)
            else:
                cfg[k] = v
        self.cfg = cfg
        tr = cfg["Training"]
        self.gaussians = gaussians
        self.dev = gaussians.device
        if self.dev.type != "cuda":
            raise RuntimeError("NativeMapper runs on the GPU only (HIP kernels, gfx950)")
        self.bg = background.detach().to(self.dev, torch.float32).reshape(-1).contiguous()
        self.monocular = bool(tr["monocular"])
        self.cameras_extent = cameras_extent
        self.group = group
        self.capacity_margin = capacity_margin
        self.rng = torch.Generator().manual_seed(seed)      # same draw on every rank
        self.viewpoints: Dict[int, object] = {}
        self.states: Dict[int, _ViewState] = {}
        self.current_window: List[int] = []
        self.occ_aware_visibility: Dict[int, torch.Tensor] = {}
        self.iteration_count = 0
        self.initialized = not self.monocular
        self.frames_to_optimize = tr["pose_window"]
        self.capacity = 0
        self._N = -1
        self._shape_key = None
        self.lanes = [_Lane(self.dev, None)] + [_Lane(self.dev, torch.cuda.Stream(device=self.dev))
                                                for _ in range(max(1, int(concurrent_views)) - 1)]
        self.loss_accum = self.lanes[0].loss_accum
        self.last_loss = None
        self.overflow_regrows = 0
        # optional phase timing (bench.py): a list collects, per iteration, HIP events on the main stream
        # around [activate + this rank's views] [exchange] [statistics fold + optimiser step]
        self.timing: Optional[list] = None

    # ---- distributed helpers ---------------------------------------------------------------------
    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(self.group), dist.get_world_size(self.group)
        return 0, 1

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    # ---- buffers sized by the number of Gaussians ------------------------------------------------
    def _ensure_model_buffers(self):
        g = self.gaussians
        N = len(g)
        if N == self._N and self._param_ptrs == tuple(getattr(g, a).data_ptr() for a in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")):
            return
        dev = self.dev
        K = int(g._features_dc.shape[1] + g._features_rest.shape[1])
        sd = int(g._scaling.shape[1])
        self.K, self.sd = K, sd
        widths = [("xyz", 3), ("f_dc", 3), ("f_rest", 3 * (K - 1)), ("opacity", 1), ("scaling", sd), ("rotation", 4),
                  ("gradnorm", 1), ("denom", 1)]
        self.off, off = {}, 0
        for name, w in widths:
            self.off[name] = (off, N * w)
            off += _align(N * w)
        if N != self._N:
            if len(self.lanes) > 1:
                torch.cuda.synchronize(self.dev)        # buffers of other streams are about to be replaced
            for ln in self.lanes:
                with torch.cuda.stream(ln.torch_stream()):   # the caching allocator ties a block to its stream
                    ln.flat = torch.zeros(off, device=dev)
                    ln.radii_max = torch.zeros(N, dtype=torch.int32, device=dev)
                    ln.radii = torch.empty(N, dtype=torch.int32, device=dev)
                    ln.n_touched = torch.empty(N, dtype=torch.int32, device=dev)
                ln.shape_key = None             # geom depends on N
            self.flat, self.radii_max = self.lanes[0].flat, self.lanes[0].radii_max
            self.scales = torch.empty(N, 3, device=dev)
            self.rots = torch.empty(N, 4, device=dev)
            self.opac = torch.empty(N, device=dev)
            self.shs = torch.empty(N, K, 3, device=dev) if K > 1 else None
            self._need_probe = True
        self._N = N
        self._param_ptrs = tuple(getattr(g, a).data_ptr() for a in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest"))
        # the optimiser reads the accumulated gradients in place: .grad = views into the flat buffer
        self._grad_views = {}
        for name, attr in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"),
                           ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation")):
            o, n = self.off[name]
            self._grad_views[attr] = self.flat[o:o + n].view_as(getattr(g, attr))

    def _section(self, name, lane=None):
        o, n = self.off[name]
        return (self.lanes[0] if lane is None else lane).flat[o:o + n]

    def _ensure_workspaces(self, ln, W, H, deg):
        with torch.cuda.stream(ln.torch_stream()):
            self._alloc_workspaces(ln, W, H, deg)
        if ln is self.lanes[0]:      # the last render of lane 0 is what callers look at
            self.color, self.depth, self.opacity = ln.color, ln.depth, ln.opacity
            self.radii, self.n_touched = ln.radii, ln.n_touched

    def _alloc_workspaces(self, ln, W, H, deg):
        key = (self._N, W, H, self.K)
        if key != ln.shape_key:
            shape = _cabi.RasterShape(self._N, W, H, deg, self.K, max(self.capacity, 1024), 1.0, 1.0, 1.0)
            sizes = _cabi.workspace_sizes(shape)
            ln.geom = torch.empty(int(sizes.geom_bytes), dtype=torch.uint8, device=self.dev)
            ln.color = torch.empty(3, H, W, device=self.dev)
            ln.depth = torch.empty(1, H, W, device=self.dev)
            ln.opacity = torch.empty(1, H, W, device=self.dev)
            ln.grad_image = torch.empty(3, H, W, device=self.dev)
            ln.grad_depth = torch.empty(1, H, W, device=self.dev)
            ln.grad_tau = torch.zeros(6, device=self.dev)
            ln.loss_partial = torch.empty(int(_cabi.lib().mgs_mapping_loss_partial_count(H * W)), device=self.dev)
            ln.shape_key = key
            ln.cap_alloc = 0
        if ln.cap_alloc != self.capacity:
            shape = _cabi.RasterShape(self._N, W, H, deg, self.K, self.capacity, 1.0, 1.0, 1.0)
            sizes = _cabi.workspace_sizes(shape)
            ln.bins = torch.empty(int(sizes.bins_bytes), dtype=torch.uint8, device=self.dev)
            ln.bwd = torch.empty(int(sizes.bwd_bytes), dtype=torch.uint8, device=self.dev)
            ln.cap_alloc = self.capacity

    # ---- window / optimiser set-up (the "keyframe" message, :427-489) ------------------------------------
    def add_keyframe(self, kf_idx: int, viewpoint):
        self.viewpoints[kf_idx] = viewpoint
        self.states[kf_idx] = _ViewState(viewpoint, self.dev)

    def set_window(self, current_window: List[int], frames_to_optimize: Optional[int] = None):
        """New keyframe optimiser: Adam state of every window view restarts (:437-489)."""
        tr = self.cfg["Training"]
        self.current_window = list(current_window)
        self.frames_to_optimize = tr["pose_window"] if frames_to_optimize is None else frames_to_optimize
        for kf in self.current_window:
            st = self.states[kf]
            st.exp_avg.zero_()
            st.exp_avg_sq.zero_()
            st.step = 0

    # ---- one view ----------------------------------------------------------------------------------
    def _view_args(self, kf_idx, cam_idx, *, accumulate, add_reg, initialization=False, forward_only=False,
                   in_window=True, stats=True, lane=None):
        vp, st, g = self.viewpoints[kf_idx], self.states[kf_idx], self.gaussians
        tr = self.cfg["Training"]
        ln = self.lanes[0] if lane is None else lane
        W, H = int(vp.image_width), int(vp.image_height)
        self._ensure_workspaces(ln, W, H, int(g.active_sh_degree))
        a = _cabi.MappingViewArgs()
        f = a.fwd
        f.shape = _cabi.RasterShape(self._N, W, H, int(g.active_sh_degree), self.K, self.capacity,
                                    math.tan(0.5 * vp.FoVx), math.tan(0.5 * vp.FoVy), 1.0)
        f.means3D, f.scales, f.rotations = g._xyz.data_ptr(), self.scales.data_ptr(), self.rots.data_ptr()
        f.opacities = self.opac.data_ptr()
        f.shs = g._features_dc.data_ptr() if self.K == 1 else self.shs.data_ptr()
        f.viewmatrix, f.projmatrix, f.projmatrix_raw = st.view.data_ptr(), st.full.data_ptr(), st.proj.data_ptr()
        f.campos, f.bg = st.view.data_ptr(), self.bg.data_ptr()
        f.geom, f.bins = ln.geom.data_ptr(), ln.bins.data_ptr()
        f.out_color, f.out_depth, f.out_opacity = ln.color.data_ptr(), ln.depth.data_ptr(), ln.opacity.data_ptr()
        f.radii, f.n_touched = ln.radii.data_ptr(), ln.n_touched.data_ptr()
        f.pair_count_out, f.pair_count_max = ln.host_D.data_ptr(), ln.d_max.data_ptr()
        f.big_tile_pass = -1 if 0 < int(ln.host_D[1]) <= 900 else 0
        a.bwd, a.grad_image, a.grad_tau = ln.bwd.data_ptr(), ln.grad_image.data_ptr(), ln.grad_tau.data_ptr()
        a.grad_depth = ln.grad_depth.data_ptr()
        # objective (utils/slam_utils.py:224-253)
        L = a.loss
        L.gt = st.gt.data_ptr()
        L.mask = None if st.mask is None else st.mask.data_ptr()
        L.exposure_a, L.exposure_b = vp.exposure_a.data_ptr(), vp.exposure_b.data_ptr()
        L.exposure_eps = float(getattr(vp, "exposure_eps", 1e-8))
        if self.monocular:
            L.w_rgb, L.w_depth = 1.0, 0.0
        else:
            alpha = float(tr.get("alpha", 0.95))
            L.w_rgb, L.w_depth = alpha, 1.0 - alpha
            if st.gt_depth is None:
                raise RuntimeError(f"keyframe {kf_idx}: RGB-D mapping needs viewpoint.gt_depth")
            L.gt_depth = st.gt_depth.data_ptr()
        L.depth_mask_threshold = 0.01
        L.apply_exposure = 0 if initialization else 1
        L.num_pixels = H * W
        L.partial = ln.loss_partial.data_ptr()
        # this view's optimiser (:452-489): pose deltas for the first frames_to_optimize window views,
        # exposure for every window view; keyframe 0 is the fixed reference; extra views: none
        A = a.adam
        optimise = in_window and not initialization and kf_idx != 0
        pose_opt = optimise and cam_idx < self.frames_to_optimize
        assert vp.T.is_contiguous() and vp.T.dtype == torch.float32 and vp.T.device == self.dev
        if vp.T.data_ptr() != st.T_ptr:
            st.matrices_fresh, st.T_ptr = False, vp.T.data_ptr()
        A.T = vp.T.data_ptr()
        if pose_opt:
            A.cam_rot_delta, A.cam_trans_delta = vp.cam_rot_delta.data_ptr(), vp.cam_trans_delta.data_ptr()
        if optimise:
            A.exposure_a, A.exposure_b = vp.exposure_a.data_ptr(), vp.exposure_b.data_ptr()
        A.exp_avg, A.exp_avg_sq = st.exp_avg.data_ptr(), st.exp_avg_sq.data_ptr()
        lr = tr["lr"]
        A.lr_rot, A.lr_trans = 0.5 * lr["cam_rot_delta"], 0.5 * lr["cam_trans_delta"]
        A.lr_a, A.lr_b = lr["exposure_a"], lr["exposure_b"]
        A.beta1, A.beta2, A.eps, A.converged_threshold = 0.9, 0.999, 1e-8, 1e-4
        A.no_pose_update = 0 if (pose_opt and cam_idx < tr["pose_window"]) else 1
        if not forward_only:
            st.step += 1
        A.step = max(1, st.step)
        a.loss_view, a.loss_accum = st.loss.data_ptr(), ln.loss_accum.data_ptr()
        a.camera_matrices_valid = 1 if st.matrices_fresh else 0
        a.forward_only = 1 if forward_only else 0
        # accumulation target: the flat gradient buffer (+ statistics of this view)
        M = a.accum
        M.scale_dims, M.accumulate, M.add_regulariser, M.regulariser_weight = self.sd, int(accumulate), int(add_reg), 10.0
        M.raw_rotations = g._rotation.data_ptr()
        ptr = lambda name: self._section(name, ln).data_ptr()
        M.grad_xyz, M.grad_features_dc, M.grad_opacity = ptr("xyz"), ptr("f_dc"), ptr("opacity")
        M.grad_features_rest = ptr("f_rest") if self.K > 1 else None
        M.grad_scaling, M.grad_rotation = ptr("scaling"), ptr("rotation")
        if stats:
            M.gradnorm_inc, M.denom_inc, M.radii_max = ptr("gradnorm"), ptr("denom"), ln.radii_max.data_ptr()
        if in_window:
            vis = self.occ_aware_visibility.get(kf_idx)
            if vis is None or vis.shape[0] != self._N:
                vis = self.occ_aware_visibility[kf_idx] = torch.zeros(self._N, dtype=torch.uint8, device=self.dev)
            M.visibility = vis.data_ptr()
        return a, st

    def _probe_capacity(self, kf_idx):
        """Synchronous stage-1 forward of one view: the pair count sizes the workspaces."""
        self.capacity = max(self.capacity, 1024)
        a, st = self._view_args(kf_idx, 0, accumulate=False, add_reg=False, forward_only=True)
        st.step = max(0, st.step)
        lib = _cabi.lib()
        vp = self.viewpoints[kf_idx]
        _cabi.check(lib.mgs_camera_from_pose(vp.T.data_ptr(), st.proj.data_ptr(), st.view.data_ptr(),
                                             st.full.data_ptr(), self._stream()), "mgs_camera_from_pose")
        _cabi.check(lib.mgs_raster_forward_project(C.byref(a.fwd), self._stream()), "mgs_raster_forward_project")
        torch.cuda.current_stream(self.dev).synchronize()
        D = int(self.lanes[0].host_D[0].item())
        self.lanes[0].d_max.zero_()
        self.capacity = max(self.capacity, 1024, (int(D * self.capacity_margin) + 1023) // 1024 * 1024)
        self._need_probe = False

    def _run_view(self, kf_idx, cam_idx, lane=None, **kw):
        if self._need_probe:
            self._probe_capacity(kf_idx)
        ln = self.lanes[0] if lane is None else lane
        # D of an earlier view lands in pinned host memory without a sync: grow BEFORE it overflows
        seen = max(int(l.host_D[0].item()) for l in self.lanes)
        if seen > 0.9 * self.capacity:
            if len(self.lanes) > 1:
                torch.cuda.synchronize(self.dev)        # workspaces of other streams are about to be replaced
            self.capacity = (int(seen * self.capacity_margin) + 1023) // 1024 * 1024
            self.overflow_regrows += 1
        a, st = self._view_args(kf_idx, cam_idx, lane=ln, **kw)
        _cabi.check(_cabi.lib().mgs_mapping_view_iteration(C.byref(a), ln.stream_ptr()), "mgs_mapping_view_iteration")
        st.matrices_fresh = not kw.get("forward_only", False)   # the Adam kernel wrote the updated matrices
        ln.keep = a

    def check_capacity(self) -> bool:
        """True iff every forward since the last check was rendered completely (host sync)."""
        worst = max(int(l.d_max.item()) for l in self.lanes)
        for l in self.lanes:
            l.d_max.zero_()
        return worst <= self.capacity

    def _fan_out(self, jobs, rank, prune):
        """Enqueue this rank's views of one iteration round-robin over the lanes and fold the lanes'
        accumulators into lane 0 (the buffer the optimiser and the all-reduce read)."""
        L = self.lanes
        used = L[:max(1, min(len(L), len(jobs)))]
        main = torch.cuda.current_stream(self.dev)
        if len(used) > 1:                      # the other streams start behind the activation kernel
            L[0].event.record(main)
            for ln in used[1:]:
                ln.stream.wait_event(L[0].event)
        for n, (gi, (kf, ci, inw)) in enumerate(jobs):
            ln = used[n % len(used)]
            if n < len(used) and ln is not L[0]:
                with torch.cuda.stream(ln.stream):
                    ln.loss_accum.zero_()
            self._run_view(kf, ci, lane=ln, accumulate=n >= len(used), add_reg=(n == 0 and rank == 0), in_window=inw,
                           forward_only=prune, stats=not prune)
        for ln in used[1:]:                    # join, then fold
            ln.event.record(ln.stream)
            main.wait_event(ln.event)
            if not prune:
                torch.add(L[0].flat, ln.flat, out=L[0].flat)
                torch.maximum(L[0].radii_max, ln.radii_max, out=L[0].radii_max)
                L[0].loss_accum += ln.loss_accum

    # ---- map() (:157-333) -------------------------------------------------------------------------------------
    def _shard(self, items):
        rank, world = self._world()
        return [(i, it) for i, it in enumerate(items) if i % world == rank], rank, world

    def _activate(self):
        g = self.gaussians
        a = _cabi.MapActivateArgs()
        a.num_gaussians, a.scale_dims, a.sh_coeffs = self._N, self.sd, self.K
        a.log_scales, a.raw_rotations, a.opacity_logits = g._scaling.data_ptr(), g._rotation.data_ptr(), g._opacity.data_ptr()
        a.features_dc = g._features_dc.data_ptr()
        a.features_rest = g._features_rest.data_ptr() if self.K > 1 else None
        a.scales, a.rotations, a.opacities = self.scales.data_ptr(), self.rots.data_ptr(), self.opac.data_ptr()
        a.shs = self.shs.data_ptr() if self.K > 1 else None
        _cabi.check(_cabi.lib().mgs_map_activate(C.byref(a), self._stream()), "mgs_map_activate")

    def _exchange(self, world):
        if world > 1:
            if dist.get_backend(self.group) == "gloo":      # CPU rehearsal (tests / 1-GPU dry runs)
                f, r = self.flat.cpu(), self.radii_max.cpu()
                dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(r, op=dist.ReduceOp.MAX, group=self.group)
                self.flat.copy_(f)
                self.radii_max.copy_(r)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(self.radii_max, op=dist.ReduceOp.MAX, group=self.group)

    def _finish(self, reset_mode=0, reset_value=0.4, stats=True):
        g = self.gaussians
        a = _cabi.MapFinishArgs()
        a.num_gaussians = self._N
        a.denom_inc = self._section("denom").data_ptr()
        if stats:
            a.gradnorm_inc, a.radii_max = self._section("gradnorm").data_ptr(), self.radii_max.data_ptr()
            a.xyz_gradient_accum, a.denom, a.max_radii2D = g.xyz_gradient_accum.data_ptr(), g.denom.data_ptr(), g.max_radii2D.data_ptr()
        a.reset_mode, a.reset_value = reset_mode, reset_value
        if reset_mode:
            a.opacity_logits = g._opacity.data_ptr()
            st = g.optimizer.state.get(g._opacity)
            if st is not None and "exp_avg" in st:
                a.opacity_exp_avg, a.opacity_exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        _cabi.check(_cabi.lib().mgs_map_finish_iteration(C.byref(a), self._stream()), "mgs_map_finish_iteration")

    def _step_gaussians(self, skip=()):
        g = self.gaussians
        for attr, view in self._grad_views.items():
            getattr(g, attr).grad = None if attr in skip else view
        g.optimizer.step()
        for attr in self._grad_views:
            getattr(g, attr).grad = None

    def map(self, current_window: Optional[List[int]] = None, prune: bool = False, iters: int = 1) -> bool:
        tr = self.cfg["Training"]
        window = list(self.current_window if current_window is None else current_window)
        if not window:
            return False
        g = self.gaussians
        in_window = set(window)
        random_stack = [k for k in self.viewpoints if k not in in_window]
        gaussian_split = False
        for st in self.states.values():       # T may have been moved from outside (tracking) since the last call
            st.matrices_fresh = False
        for _ in range(iters):
            self.iteration_count += 1
            self._ensure_model_buffers()
            extras = [random_stack[i] for i in torch.randperm(len(random_stack), generator=self.rng)[:2].tolist()]
            if prune:
                extras = []
            jobs, rank, world = self._shard([(kf, ci, True) for ci, kf in enumerate(window)] + [(kf, -1, False) for kf in extras])
            ev = None
            if self.timing is not None and not prune:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record()
            self._activate()
            self.loss_accum.zero_()
            if not jobs:
                self.flat.zero_()
                self.radii_max.zero_()
            self._fan_out(jobs, rank, prune)
            if prune:
                self._prune_pass(window, world)
                return False
            if ev:
                ev[1].record()
            self._exchange(world)
            if ev:
                ev[2].record()
            update_gaussian = self.iteration_count % tr["gaussian_update_every"] == tr["gaussian_update_offset"]
            reset = (self.iteration_count % tr["gaussian_reset"]) == 0 and not update_gaussian
            self._finish(reset_mode=2 if reset else 0, reset_value=0.4)
            if update_gaussian:
                g.densify_and_prune(self.cfg["opt_params"]["densify_grad_threshold"], tr["gaussian_th"],
                                    self.cameras_extent * tr["gaussian_extent"], tr["size_threshold"],
                                    generator=self._split_generator())
                gaussian_split = True
                # the rebuilt parameters carry no gradient: the reference's optimizer.step() moves nothing
            else:
                self._step_gaussians(skip=("_opacity",) if reset else ())
                gaussian_split = gaussian_split or reset
            g.update_learning_rate(self.iteration_count)
            self.last_loss = self.loss_accum.clone()
            if ev:
                ev[3].record()
                self.timing.append((ev, len(jobs)))
        self._sync_views(window)
        return gaussian_split

    def timing_summary(self):
        """Mean milliseconds per iteration of the phases recorded while `self.timing` was a list
        (host sync): this rank's views (activation + forward / loss / backward / per-view Adam of its
        shard), the exchange (all-reduce(sum) of the flat buffer + all-reduce(max) of the radii), and
        the statistics fold + Gaussian optimiser step; plus the views this rank rendered per iteration."""
        torch.cuda.synchronize(self.dev)
        t = self.timing or []
        if not t:
            return None
        n = len(t)
        return {"compute_ms": sum(e[0].elapsed_time(e[1]) for e, _ in t) / n,
                "exchange_ms": sum(e[1].elapsed_time(e[2]) for e, _ in t) / n,
                "update_ms": sum(e[2].elapsed_time(e[3]) for e, _ in t) / n,
                "views_per_iteration": sum(j for _, j in t) / n, "iterations": n}

    def _sync_views(self, window):
        """Keyframe-parallel: a view's pose and exposure are stepped on the rank that owns it; at
        the end of a map() call the owners publish them (one small all-reduce: non-owners add zeros),
        so that every rank holds the whole window's cameras whatever the next window's sharding is."""
        rank, world = self._world()
        if world == 1:
            return
        buf = torch.zeros(len(window), 18, device=self.dev)
        for i, kf in enumerate(window):
            if i % world == rank:
                vp = self.viewpoints[kf]
                buf[i, :16] = vp.T.reshape(-1)
                buf[i, 16], buf[i, 17] = vp.exposure_a.detach().reshape(()), vp.exposure_b.detach().reshape(())
        if dist.get_backend(self.group) == "gloo":
            tmp = buf.cpu()
            dist.all_reduce(tmp, group=self.group)
            buf = tmp.to(self.dev)
        else:
            dist.all_reduce(buf, group=self.group)
        with torch.no_grad():
            for i, kf in enumerate(window):
                vp = self.viewpoints[kf]
                vp.T.copy_(buf[i, :16].view(4, 4))
                vp.exposure_a.fill_(float(buf[i, 16]))
                vp.exposure_b.fill_(float(buf[i, 17]))
                self.states[kf].matrices_fresh = False

    def _split_generator(self):
        """densify_and_split's random offsets (gaussian_model.py:608-609) come from a generator
        keyed by the iteration: replicated maps (keyframe-parallel) draw identical offsets on every
        rank without a broadcast, and a run is reproducible."""
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(0x5EED + 7919 * self.iteration_count)
        return gen

    def _prune_pass(self, window, world):
        """:259-290 - observation counts over the window's occ-aware visibility, prune the
        recently inserted Gaussians seen by <= 3 views ("slam"), or by < 3 ("odometry")."""
        tr = self.cfg["Training"]
        g = self.gaussians
        if world > 1:
            # every rank learns every window view's visibility: round k gathers the k-th local view of
            # each rank (window position k * world + rank); ranks without one contribute zeros
            from .parallel import all_gather_visibility
            rank = self._world()[0]
            for k in range((len(window) + world - 1) // world):
                i = k * world + rank
                mine = self.occ_aware_visibility[window[i]] if i < len(window) else torch.zeros(self._N, dtype=torch.uint8, device=self.dev)
                rows = all_gather_visibility(mine.to(torch.int32), self.group)        # bool [world, N]
                for r in range(world):
                    if k * world + r < len(window):
                        self.occ_aware_visibility[window[k * world + r]] = rows[r].to(torch.uint8)
        if len(window) == tr["window_size"]:
            n_obs = torch.zeros(self._N, dtype=torch.int32, device=self.dev)
            for kf in window:
                n_obs += self.occ_aware_visibility[kf].to(torch.int32)
            g.n_obs = n_obs
            to_prune = None
            if tr["prune_mode"] == "odometry":
                to_prune = n_obs < 3
            elif tr["prune_mode"] == "slam":
                newest = sorted(window, reverse=True)
                mask = g.unique_kfIDs >= (newest[2] if self.initialized else 0)
                to_prune = (n_obs <= 3) & mask
            if to_prune is not None and self.monocular:
                g.prune_points(to_prune)
                keep = ~to_prune
                for kf in window:
                    self.occ_aware_visibility[kf] = self.occ_aware_visibility[kf][keep]
            self.initialized = True

    # ---- initialize_map (:91-155) -------------------------------------------------------------------------------
    def initialize_map(self, kf_idx: int, iters: Optional[int] = None):
        tr, op = self.cfg["Training"], self.cfg["opt_params"]
        g = self.gaussians
        iters = tr["init_itr_num"] if iters is None else iters
        for it in range(iters):
            self.iteration_count += 1
            self._ensure_model_buffers()
            self._activate()
            self.loss_accum.zero_()
            self._run_view(kf_idx, 0, accumulate=False, add_reg=False, initialization=True)
            self._finish()
            densify = it % tr["init_gaussian_update"] == 0
            if densify:
                g.densify_and_prune(op["densify_grad_threshold"], tr["init_gaussian_th"],
                                    self.cameras_extent * tr["init_gaussian_extent"], None)
            reset = self.iteration_count == tr["init_gaussian_reset"] or self.iteration_count == op["densify_from_iter"]
            if reset:
                g.reset_opacity()
            if not densify:
                self._step_gaussians(skip=("_opacity",) if reset else ())
        self.last_loss = self.loss_accum.clone()
        return self.occ_aware_visibility.get(kf_idx)
