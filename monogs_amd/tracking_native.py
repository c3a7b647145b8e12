"""First-order tracking iterations enqueued natively (one C-ABI call per iteration).

`tracking_step_first_order[_fused]` in slam_loops.py mirror the reference's Python loop
body (utils/slam_frontend.py:493-630) and spend ~1 ms of HOST time per iteration in
autograd, tensor allocation and ctypes marshalling - more than the GPU needs for the whole
iteration.  During tracking the map is frozen, so everything that does not change between
iterations is prepared once here (activated Gaussian attributes, workspaces at a fixed pair
capacity, the argument block) and an iteration is a single `mgs_tracking_iteration` call:
camera matrices from T, rasteriser forward, monocular tracking objective, pose-only
rasteriser backward, Adam + update_pose, all on the current stream with no host sync.

Semantics per iteration are those of `tracking_step_first_order_fused`
(tests/test_raster_gpu.py::test_native_tracking_matches_python_loop).
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _cabi
from . import rasterizer as R


class NativeTracker:
    """Tracks `viewpoint` (slam_loops.ViewCamera-like: T, projection_matrix, FoVx/FoVy,
    image size, original_image, rgb_pixel_mask_mapping, exposure_a/b/eps, cam_*_delta)
    against the frozen `gaussians` (get_xyz/get_scaling/get_rotation/get_opacity/
    get_features, active_sh_degree).  `viewpoint.T`, the exposure parameters and the
    convergence flag live on the device and are updated in place by `step()`."""

    def __init__(self, viewpoint, gaussians, background, huber_delta=0.01, lr_rot=0.003,
                 lr_trans=0.001, lr_a=0.02, lr_b=0.02, betas=(0.9, 0.999), eps=1e-8,
                 converged_threshold=1e-4, capacity_margin=1.5, pnorm=2.0):
        # (huber_delta, pnorm) as slam_loops.tracking_norm(config) returns them: the reference's first-order
        # objective is Huber + L2 when RGN.use_huber, else the RGN.pnorm-norm without Huber (huber_delta = 0),
        # slam_frontend.py:596-600.  p = 1 and p = 2 ride in the forward blend's epilogue, any other p >= 1
        # costs one more launch per iteration.
        vp = viewpoint
        dev = vp.T.device
        if dev.type != "cuda":
            raise RuntimeError("NativeTracker runs on the GPU only (HIP kernels, gfx950)")
        self.vp, self.dev = vp, dev
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        with torch.no_grad():
            self.means = f32(gaussians.get_xyz)
            sc = gaussians.get_scaling
            self.scales = f32(sc.repeat(1, 3) if sc.shape[-1] == 1 else sc)
            self.rots = f32(gaussians.get_rotation)
            self.opac = f32(gaussians.get_opacity).reshape(-1)
            self.shs = f32(gaussians.get_features)
        N, K = int(self.means.shape[0]), int(self.shs.shape[1])
        H, W = int(vp.image_height), int(vp.image_width)
        self.N, self.H, self.W = N, H, W
        assert vp.T.is_contiguous() and vp.T.dtype == torch.float32
        self.proj = f32(vp.projection_matrix)
        self.view = torch.empty(4, 4, device=dev)
        self.full = torch.empty(4, 4, device=dev)
        self.bg = f32(background).reshape(-1)
        self.gt = f32(vp.original_image)
        m = getattr(vp, "rgb_pixel_mask_mapping", None)
        self.mask = None if m is None else f32(m)
        lib = _cabi.lib()
        stream = self._stream()
        _cabi.check(lib.mgs_camera_from_pose(vp.T.data_ptr(), self.proj.data_ptr(), self.view.data_ptr(),
                                             self.full.data_ptr(), stream), "mgs_camera_from_pose")

        # one probing forward (stage 1 only) sizes the pair capacity for the whole run
        shape = _cabi.RasterShape(N, W, H, int(gaussians.active_sh_degree), K, 0,
                                  math.tan(0.5 * vp.FoVx), math.tan(0.5 * vp.FoVy), 1.0)
        sizes = _cabi.workspace_sizes(shape)
        self.geom = torch.empty(int(sizes.geom_bytes), dtype=torch.uint8, device=dev)
        self.color = torch.empty(3, H, W, device=dev)
        self.depth = torch.empty(1, H, W, device=dev)
        self.opacity = torch.empty(1, H, W, device=dev)
        self.radii = torch.empty(N, dtype=torch.int32, device=dev)
        self.n_touched = torch.empty(N, dtype=torch.int32, device=dev)
        a = _cabi.TrackingIterArgs()
        f = a.fwd
        f.shape = shape
        f.means3D, f.scales, f.rotations = self.means.data_ptr(), self.scales.data_ptr(), self.rots.data_ptr()
        f.opacities, f.shs = self.opac.data_ptr(), self.shs.data_ptr()
        f.viewmatrix, f.projmatrix, f.projmatrix_raw = self.view.data_ptr(), self.full.data_ptr(), self.proj.data_ptr()
        f.campos, f.bg, f.geom = self.view.data_ptr(), self.bg.data_ptr(), self.geom.data_ptr()
        f.out_color, f.out_depth, f.out_opacity = self.color.data_ptr(), self.depth.data_ptr(), self.opacity.data_ptr()
        f.radii, f.n_touched = self.radii.data_ptr(), self.n_touched.data_ptr()
        # sticky high-water mark of the pair count over ALL iterations since the last check: an
        # overflow in the middle of a run is seen even when the last iteration fits
        self._d_max = torch.zeros(1, dtype=torch.int32, device=dev)
        f.pair_count_max = self._d_max.data_ptr()
        self._host_D = torch.zeros(2, dtype=torch.int32).pin_memory()       # [0] = D, [1] = fullest tile
        f.pair_count_out = self._host_D.data_ptr()
        _cabi.check(lib.mgs_raster_forward_project(C.byref(f), stream), "mgs_raster_forward_project")
        off = int(sizes.off_counters)
        self._counter = self.geom[off:off + 4].view(torch.int32)
        D = int(self._counter.item())
        self.capacity_margin = capacity_margin
        self._alloc_bins(a, max(1024, int(D * capacity_margin)))

        self.grad_image = torch.empty(3, H, W, device=dev)
        self.grad_tau = torch.zeros(6, device=dev)
        self.grad_exposure = torch.zeros(2, device=dev)
        self.one = torch.ones(1, device=dev)
        a.grad_image, a.grad_tau = self.grad_image.data_ptr(), self.grad_tau.data_ptr()
        a.grad_exposure, a.one = self.grad_exposure.data_ptr(), self.one.data_ptr()
        L = a.loss
        self.partial = torch.empty(int(lib.mgs_tracking_loss_partial_count(H * W)), device=dev)
        self.scalars = torch.zeros(2, device=dev)
        L.gt = self.gt.data_ptr()
        L.mask = None if self.mask is None else self.mask.data_ptr()
        L.exposure_a, L.exposure_b = vp.exposure_a.data_ptr(), vp.exposure_b.data_ptr()
        L.exposure_eps, L.huber_delta, L.num_pixels = float(vp.exposure_eps), float(huber_delta), H * W
        if not (float(pnorm) >= 1.0):
            raise ValueError(f"pnorm must be >= 1, got {pnorm}")
        L.pnorm = float(pnorm)
        L.partial, L.scalars = self.partial.data_ptr(), self.scalars.data_ptr()
        A = a.adam
        self.exp_avg = torch.zeros(8, device=dev)
        self.exp_avg_sq = torch.zeros(8, device=dev)
        self.converged = torch.zeros(1, dtype=torch.int32, device=dev)
        A.cam_rot_delta, A.cam_trans_delta = vp.cam_rot_delta.data_ptr(), vp.cam_trans_delta.data_ptr()
        A.exposure_a, A.exposure_b = vp.exposure_a.data_ptr(), vp.exposure_b.data_ptr()
        A.exp_avg, A.exp_avg_sq = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        A.T, A.converged = vp.T.data_ptr(), self.converged.data_ptr()
        (A.lr_rot, A.lr_trans, A.lr_a, A.lr_b, A.beta1, A.beta2, A.eps, A.converged_threshold) = (
            lr_rot, lr_trans, lr_a, lr_b, betas[0], betas[1], eps, converged_threshold)
        # best-iterate block shared by the first- and second-order iterations (slam_frontend.py:423-425,
        # 523-528): {best L1, T[16], a, b, index of the best iteration, iteration counter}
        self.best = torch.zeros(_cabi.TRACK_BEST_FLOATS, device=dev)
        self.best[0] = float("inf")
        a.best = self.best.data_ptr()
        A.sticky_converged = 1
        self.args = a
        self.t = 0
        self._matrices_fresh = False
        self._T_ptr = vp.T.data_ptr()
        self._d_max.zero_()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _alloc_bins(self, a, cap):
        a.fwd.shape.pair_capacity = cap
        sizes = _cabi.workspace_sizes(a.fwd.shape)
        self.bins = torch.empty(int(sizes.bins_bytes), dtype=torch.uint8, device=self.dev)
        self.bwd = torch.empty(int(sizes.bwd_bytes), dtype=torch.uint8, device=self.dev)
        a.fwd.bins, a.bwd = self.bins.data_ptr(), self.bwd.data_ptr()
        self.capacity = cap

    # ---- second order (sketched Levenberg-Marquardt, slam_frontend.py:455-710) ----------
    def enable_second_order(self, stack_dim=16, sketch_dim=64, initial_lambda=1e-3, max_lambda=1e7,
                            min_lambda=1e-6, increase_factor=5.0, decrease_factor=5.0,
                            converged_threshold=1e-5, seed=0, keep_sketch=False, repeat_dim=1):
        """Allocate the sketch scratch; defaults are configs/mono/tum/base_config.yaml:255-268.
        `keep_sketch`: leave Sf / SJ of the last iteration readable (`self.sketch`) - the accumulators are
        then cleared by memset launches at the start of an iteration instead of by their consumer at its end
        (four launches more per iteration).  `repeat_dim` (base_config.yaml:258): that many sketched backward
        passes over the one render of an iteration, each with its own partition, rows stacked
        (slam_frontend.py:654-669)."""
        dev, HW = self.dev, self.H * self.W
        d = stack_dim * sketch_dim
        R = int(repeat_dim)
        if R < 1:
            raise ValueError("repeat_dim must be >= 1")
        so = _cabi.TrackingSOArgs()
        C.memmove(C.byref(so.base), C.byref(self.args), C.sizeof(_cabi.TrackingIterArgs))
        sizes = _cabi.workspace_sizes(so.base.fwd.shape)
        self.so_bucket = torch.empty(R, HW, dtype=torch.int32, device=dev)
        self.so_weights = torch.empty(R, HW, device=dev)
        self.so_accum = torch.zeros(9 * R * d + 4, device=dev)
        # (so_accum is zero-filled once: the iteration's kernels keep the accumulators zero between calls -
        # scratch_kept_zero - so no memset launches are needed per iteration; sketch_ws needs no initial state)
        self.so_sketch_ws = torch.empty(int(sizes.sketch_bytes), dtype=torch.uint8, device=dev)
        self.lm_state = torch.tensor([initial_lambda, 0.0, 0.0, 0.0], device=dev)
        self._lm_initial = self.lm_state.clone()
        self.so_x = torch.zeros(8, device=dev)
        so.stack_dim, so.sketch_dim, so.repeat_dim = stack_dim, sketch_dim, R
        so.bucket, so.weights = self.so_bucket.data_ptr(), self.so_weights.data_ptr()
        so.accum, so.sketch_ws = self.so_accum.data_ptr(), self.so_sketch_ws.data_ptr()
        so.lm.lm_state, so.lm.x_out = self.lm_state.data_ptr(), self.so_x.data_ptr()
        so.lm.increase_factor, so.lm.decrease_factor = increase_factor, decrease_factor
        so.lm.min_lambda, so.lm.max_lambda = min_lambda, max_lambda
        so.lm.converged_threshold = converged_threshold
        so.scratch_kept_zero = 0 if keep_sketch else 1
        self.so_args, self.so_d, self.so_seed, self.so_t, self.so_repeat = so, R * d, int(seed), 0, R

    def step_second_order(self):
        """Enqueue one sketched LM iteration: fresh random bucket partition, forward, sketched
        residual, sketch-mode backward, damped solve and pose / exposure step, all on the
        device.  Returns lm_state = [lambda, ||residual||_1, 1, converged] (device tensor)."""
        so = self.so_args
        self._sync_pose_pointer()
        # pointers that _alloc_bins may have replaced since enable_second_order
        so.base.fwd.bins, so.base.bwd = self.args.fwd.bins, self.args.bwd
        if so.base.fwd.shape.pair_capacity != self.args.fwd.shape.pair_capacity:
            # the backward's sketch scratch holds one slab per potential backward item: it grows with the capacity
            so.base.fwd.shape.pair_capacity = self.args.fwd.shape.pair_capacity
            need = int(_cabi.workspace_sizes(so.base.fwd.shape).sketch_bytes)
            if self.so_sketch_ws.numel() < need:
                self.so_sketch_ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
                so.sketch_ws = self.so_sketch_ws.data_ptr()
        so.base.adam.T = self.args.adam.T
        so.base.best = self.args.best
        so.base.camera_matrices_valid = 1 if self._matrices_fresh else 0
        self.so_t += 1
        so.key = (self.so_seed * 0x9E3779B97F4A7C15 + self.so_t) & 0xFFFFFFFFFFFFFFFF
        _cabi.check(_cabi.lib().mgs_tracking_iteration_second_order(C.byref(so), self._stream()),
                    "mgs_tracking_iteration_second_order")
        self._matrices_fresh = True      # the LM kernel wrote the matrices of the stepped pose
        return self.lm_state

    @property
    def sketch(self):
        """(Sf [R d], SJ [R d, 8]) of the last second-order iteration (views / a small cat); needs
        enable_second_order(keep_sketch=True) - otherwise the LM kernel has zeroed them again."""
        if self.so_args.scratch_kept_zero:
            raise RuntimeError("enable_second_order(keep_sketch=True) is needed to read the sketch back")
        d = self.so_d
        a = self.so_accum
        return a[:d], torch.cat((a[3 * d:9 * d].view(d, 6), a[d:3 * d].view(d, 2)), dim=1)

    def _sync_pose_pointer(self):
        """`viewpoint.T` may have been REBOUND to a new tensor by Python code between two native
        calls (camera.T = ...): follow it instead of updating a stale buffer, and rebuild the camera
        matrices.  (In-place writes to T by others are not detectable: call invalidate_matrices().)"""
        vp = self.vp
        if vp.T.data_ptr() != self._T_ptr:
            if not (vp.T.is_contiguous() and vp.T.dtype == torch.float32 and vp.T.device == self.dev):
                raise RuntimeError("viewpoint.T must stay a contiguous float32 4x4 tensor on the tracker's device")
            self._T_ptr = vp.T.data_ptr()
            self.args.adam.T = self._T_ptr
            self._matrices_fresh = False

    def invalidate_matrices(self):
        self._matrices_fresh = False

    def step(self):
        """Enqueue one iteration; returns the device convergence flag (int32[1])."""
        self._sync_pose_pointer()
        self.t += 1
        self.args.adam.step = self.t
        # pinned, written by every forward: no sync needed to read the previous iteration's value
        self.args.fwd.big_tile_pass = -1 if 0 < int(self._host_D[1]) <= 900 else 0
        self.args.camera_matrices_valid = 1 if self._matrices_fresh else 0
        _cabi.check(_cabi.lib().mgs_tracking_iteration(C.byref(self.args), self._stream()),
                    "mgs_tracking_iteration")
        self._matrices_fresh = True      # the Adam kernel wrote the matrices of the updated pose
        return self.converged

    @property
    def loss(self):
        return self.scalars[0]

    def pairs(self) -> int:
        """Pair count D of the last forward (host sync)."""
        return int(self._counter.item())

    def check_capacity(self):
        """True if EVERY iteration since the last check rendered completely (the forward keeps a
        sticky maximum of the pair count on the device); otherwise grows the workspaces for the
        worst count seen and returns False - the caller re-runs the affected iterations (`run`
        does, from a snapshot of the state)."""
        worst = int(self._d_max.item())
        self._d_max.zero_()
        if worst <= self.capacity:
            return True
        self._alloc_bins(self.args, int(worst * self.capacity_margin))
        return False

    def reset_frame(self):
        """The per-frame state of the reference's loop (slam_frontend.py:423-429): best iterate = none,
        lambda_ = initial_lambda, no previous second-order loss, and both sticky convergence flags cleared
        (once set they turn every later step() / step_second_order() into a no-op: a caller driving its own
        step loop over several frames calls this between frames; run() does).  The Adam moments are NOT
        reset here but by `reset_optimizer()`, which run() also calls by default: the reference builds a new
        optimiser per frame."""
        self.reset_best()
        self.converged.zero_()
        if hasattr(self, "lm_state"):
            self.lm_state.copy_(self._lm_initial)

    def reset_optimizer(self):
        """torch.optim.Adam(opt_params) of a new frame (slam_frontend.py:455): zero moments, step count 0."""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.t = 0

    # ---- best iterate (slam_frontend.py:423-425, 465-470, 523-528, 819-822) -----------------
    def reset_best(self):
        self.best.zero_()
        self.best[0] = float("inf")

    @property
    def best_loss(self):
        """||residual||_1 of the best iterate so far (device scalar; inf before the first iteration)."""
        return self.best[0]

    @property
    def last_l1(self):
        """||residual||_1 (before Huber) of the render the LAST iteration started from - the reference's
        loss_tracking_scalar (slam_frontend.py:510); device scalar."""
        return self.best[21]

    @property
    def last_step_norm(self):
        """|tau| applied by the last first-order iteration's update_pose (pose_utils.py:88-98), or |x| of the
        last LM solve (slam_frontend.py:693); device scalar."""
        return self.best[22]

    def best_iteration(self) -> int:
        """0-based index (first- and second-order iterations counted together) of the best iterate
        (host sync); -1 before the first iteration."""
        b = self.best.cpu()
        return int(b[19].item()) if math.isfinite(b[0].item()) else -1

    def assign_best(self):
        """TempCamera.assign of the best iterate (slam_frontend.py:37-42): pose and exposure go back to
        the state whose render had the smallest L1 residual; stays on the device (no sync).  A no-op
        before the first iteration (best_viewpoint_params is None)."""
        vp = self.vp
        self._sync_pose_pointer()
        with torch.no_grad():
            have = torch.isfinite(self.best[0])
            vp.T.copy_(torch.where(have, self.best[1:17].view(4, 4), vp.T))
            vp.exposure_a.copy_(torch.where(have, self.best[17:18], vp.exposure_a))
            vp.exposure_b.copy_(torch.where(have, self.best[18:19], vp.exposure_b))
            vp.cam_rot_delta.zero_()
            vp.cam_trans_delta.zero_()
        self._matrices_fresh = False

    def render_current(self):
        """Forward only at the current pose (the tracker's output buffers: color / depth / opacity /
        radii / n_touched): after assign_best() these are the best iterate's render_pkg, which the
        reference hands to the keyframe test (slam_frontend.py:819-822, 1918-1924)."""
        self._sync_pose_pointer()
        lib, stream, f = _cabi.lib(), self._stream(), self.args.fwd
        _cabi.check(lib.mgs_camera_from_pose(self.vp.T.data_ptr(), self.proj.data_ptr(), self.view.data_ptr(),
                                             self.full.data_ptr(), stream), "mgs_camera_from_pose")
        _cabi.check(lib.mgs_raster_forward_project(C.byref(f), stream), "mgs_raster_forward_project")
        _cabi.check(lib.mgs_raster_forward_blend(C.byref(f), stream), "mgs_raster_forward_blend")
        self._matrices_fresh = True
        return {"render": self.color, "depth": self.depth, "opacity": self.opacity, "radii": self.radii,
                "n_touched": self.n_touched, "visibility_filter": self.radii > 0}

    def _snapshot(self):
        vp = self.vp
        keep = dict(T=vp.T.detach().clone(), a=vp.exposure_a.detach().clone(), b=vp.exposure_b.detach().clone(),
                    rot=vp.cam_rot_delta.detach().clone(), trans=vp.cam_trans_delta.detach().clone(),
                    m=self.exp_avg.clone(), v=self.exp_avg_sq.clone(), t=self.t, best=self.best.clone())
        if hasattr(self, "lm_state"):
            keep["lm"], keep["so_t"] = self.lm_state.clone(), self.so_t
        return keep

    def _restore(self, keep):
        vp = self.vp
        with torch.no_grad():
            vp.T.copy_(keep["T"])
            vp.exposure_a.copy_(keep["a"]); vp.exposure_b.copy_(keep["b"])
            vp.cam_rot_delta.copy_(keep["rot"]); vp.cam_trans_delta.copy_(keep["trans"])
            self.exp_avg.copy_(keep["m"]); self.exp_avg_sq.copy_(keep["v"])
            self.best.copy_(keep["best"])
            if "lm" in keep:
                self.lm_state.copy_(keep["lm"])
                self.so_t = keep["so_t"]
        self.t = keep["t"]
        self.converged.zero_()
        self._matrices_fresh = False

    def run(self, max_iters=100, check_every=10, second_order_iters=0, use_first_order_best=True,
            use_best_loss=True, render_best=True, reset_optimizer=True):
        """The reference's loop for one frame (slam_frontend.py:455-822): first-order iterations until
        converged or max_iters, then `second_order_iters` sketched LM iterations (enable_second_order
        first) unless the first order converged (its `break` leaves the whole loop, :623-626).  Every
        iteration compares the L1 norm of its un-Hubered residual with the best so far on the device
        (:510, :523-528); with `use_first_order_best` the second-order phase starts from the best
        first-order state (:465-470), with `use_best_loss` the frame ends at the best state (:819-822)
        and - `render_best` - the output buffers are re-rendered there, so that n_touched / depth /
        opacity are the best iterate's (what :1918-1924 consume).  Both default to True as in
        configs/mono/tum/base_config.yaml:268-273.  The convergence flags are read back every
        `check_every` iterations only; they are sticky on the device (an iteration enqueued after
        convergence changes nothing), so the result does not depend on `check_every`.
        If ANY iteration overflowed the fixed pair capacity, pose, exposure, optimiser and best-iterate
        state are restored from the snapshot taken on entry, the workspaces grow and the run is
        repeated: no truncated render ever reaches the result.  Returns the iterations enqueued.
        `reset_optimizer` (default True): the reference builds a new torch.optim.Adam for every frame
        (slam_frontend.py:453-455), so a tracker that is run() again starts from zero moments and step count 0,
        exactly like a freshly built one; pass False to carry the moments over on purpose."""
        self.reset_frame()
        if reset_optimizer:
            self.reset_optimizer()
        keep = self._snapshot()
        for attempt in range(4):
            it = 0
            first_converged = False
            while it < max_iters:
                for _ in range(min(check_every, max_iters - it)):
                    self.step()
                    it += 1
                if int(self.converged.item()):
                    first_converged = True
                    break
            if second_order_iters > 0 and not first_converged:
                if use_first_order_best:
                    self.assign_best()
                done = 0
                last = None
                while done < second_order_iters:
                    for _ in range(min(check_every, second_order_iters - done)):
                        if done == second_order_iters - 1 and not use_best_loss:
                            # The reference computes a step in its last iteration but only ever APPLIES a step at
                            # the top of the next one (slam_frontend.py:474-479): the frame ends at the state the
                            # last iteration rendered.  (With use_best_loss the best rendered state wins anyway.)
                            vp = self.vp
                            last = (vp.T.detach().clone(), vp.exposure_a.detach().clone(), vp.exposure_b.detach().clone())
                        self.step_second_order()
                        done += 1
                        it += 1
                    if float(self.lm_state[3].item()) != 0.0:
                        break
                if last is not None:
                    with torch.no_grad():
                        self.vp.T.copy_(last[0]); self.vp.exposure_a.copy_(last[1]); self.vp.exposure_b.copy_(last[2])
                    self._matrices_fresh = False
            if use_best_loss:
                self.assign_best()
                if render_best:
                    self.render_current()
            if self.check_capacity():
                return it
            self._restore(keep)
        raise RuntimeError("pair capacity still exceeded after growing the workspaces three times")
