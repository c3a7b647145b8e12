"""Config-4-shaped run without the dataset: a synthetic 640x480 sequence with fr3_office
intrinsics pushed through the hot-path pieces in the order MonoGS chains them, with the
hyper-parameters of configs/mono/tum/base_config.yaml (40 first-order + 10 second-order tracking
iterations per frame :250-268, window 8 / pose window 3 :31-32, 150 mapping iterations per
keyframe :26, densify every 150 at offset 50 :27-28, opacity reset :31, 1050 initialisation
iterations :20-24; `single_thread: True`, i.e. tracking and mapping alternate in one process).

TUM fr3_office itself is not available offline (BASELINE.json config 4); if a copy is mounted,
point MONOGS_TUM_DIR at it and `load_sequence` reads it through eval_metrics.TUMSequence instead
of rendering the synthetic world.  Nothing is downloaded.

What is mirrored (/root/reference utils/slam_frontend.py, slam_backend.py):
  initialisation       frontend.initialize :236-267 + add_new_keyframe(init) :183-230,
                       backend "init" message :413-425 -> initialize_map
  per frame            tracking :340-902 from the previous pose (NativeTracker: first order until
                       converged, then the sketched LM iterations), median depth :900
  keyframe insertion   add_new_keyframe :183-230 (monocular: rendered depth, outliers replaced by the
                       median, noise), backend "keyframe" message :427-493 (extend_from_pcd_seq, new
                       keyframe optimiser, map(iters), map(prune=True))
What is NOT the reference's: the keyframe POLICY (is_keyframe / add_to_window :1692-1783 are pure
policy, out of scope): a keyframe every `kf_interval` frames, window = the newest `window_size`
keyframes.
"""
from __future__ import annotations

import math
import os
import time
from typing import List, Optional

import torch

from . import synthetic as S
from .gaussian_model import GaussianModel
from .mapping_native import NativeMapper
from .pose import SE3_exp
from .slam_loops import GaussianParams, Pipe, ViewCamera
from .tracking_native import NativeTracker


class Frame:
    """One input frame: image [3,H,W], optional sensor depth [H,W], ground-truth pose."""

    def __init__(self, uid, image, depth, T_gt):
        self.uid, self.image, self.depth, self.T_gt = uid, image, depth, T_gt


def make_world(n_gaussians: int, W: int, H: int, poses: List[torch.Tensor], seed: int = 0, sigma_px: float = 2.5):
    """A static world of Gaussians that fills the view frusta along the trajectory: a third of
    them are placed (as synthetic.make_scene places them) in front of the first, the middle and the
    last pose each.  Returns slam_loops.GaussianParams on the CPU."""
    anchors = [poses[0], poses[len(poses) // 2], poses[-1]]
    parts = []
    for k, T in enumerate(anchors):
        sc = S.make_scene(n_gaussians // 3, W, H, seed=seed + 17 * k)
        R, t = T[:3, :3], T[:3, 3]
        xyz_w = (sc.means3D - t) @ R                         # camera -> world: R^T (p - t)
        # (orientations are random, so they are left as drawn)
        scale = sc.log_scales + math.log(sigma_px / 1.5)
        parts.append((xyz_w, scale, sc.rot, sc.opacity_logit + 1.0, sc.features_dc))
    cat = [torch.cat([p[i] for p in parts]) for i in range(5)]
    return GaussianParams(*cat)


def trajectory(n_frames: int, step=(0.012, -0.006, 0.008, 0.004, -0.003, 0.002)) -> List[torch.Tensor]:
    """World-to-camera poses of a smooth hand-held-like motion: ~1.6 cm and ~0.3 deg per frame."""
    tau = torch.tensor(step)
    return [SE3_exp(k * tau) for k in range(n_frames)]


def load_sequence(n_frames: int, W: int = 640, H: int = 480, dev="cuda", world_gaussians: int = 150_000, seed: int = 0,
                  sigma_px: float = 2.5):
    """(frames, camera, source): the mounted TUM sequence if MONOGS_TUM_DIR is set, else frames
    rendered from a synthetic world along `trajectory`."""
    cam = S.make_camera(W, H)
    tum = os.environ.get("MONOGS_TUM_DIR")
    if tum and os.path.isdir(tum):
        from .eval_metrics import TUMSequence
        seq = TUMSequence(tum)
        frames = []
        for k in range(min(n_frames, len(seq))):
            img, depth, T = seq[k]
            frames.append(Frame(k, img.to(dev), None if depth is None else depth.to(dev), T))
        return frames, cam, f"TUM sequence at {tum}"
    from .gaussian_renderer import render
    poses = trajectory(n_frames)
    world = make_world(world_gaussians, W, H, poses, seed=seed, sigma_px=sigma_px)
    world = GaussianParams(*(t.to(dev) for t in (world._xyz.data, world._scaling.data, world._rotation.data,
                                                 world._opacity.data, world._features_dc.data)))
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    bg = torch.zeros(3, device=dev)
    frames = []
    with torch.no_grad():
        for k, T in enumerate(poses):
            v = ViewCamera(k, torch.zeros(3, H, W), T, cam.projmatrix_raw, fovx, fovy, H, W, dev)
            pkg = render(v, world, Pipe, bg)
            frames.append(Frame(k, pkg["render"].clamp(0, 1).clone(), pkg["depth"][0].clone(), T))
    return frames, cam, f"synthetic world, {world_gaussians} Gaussians, {n_frames} frames @ {W}x{H}"


def _median_depth(depth, opacity, mask=None):
    """get_median_depth (utils/slam_utils.py:286-297) with the std."""
    valid = (depth > 0) & (opacity > 0.95)
    if mask is not None:
        valid = valid & mask
    d = depth[valid]
    if d.numel() < 2:
        return None, None, valid
    return d.median(), d.std(), valid


def keyframe_depth(frame_image, depth, opacity, sensor_depth=None, generator=None, rgb_boundary_threshold=0.01):
    """add_new_keyframe (utils/slam_frontend.py:183-234): the depth map a new keyframe's Gaussians
    are back-projected with.  Monocular: the depth rendered at the tracked pose, outliers (beyond one
    std of the median, or not opaque) replaced by the median, plus noise; first keyframe: 2 m +- 0.3.
    With a depth sensor: the observed depth.  Pixels without image content are dropped (depth 0)."""
    valid_rgb = (frame_image.sum(dim=0) > rgb_boundary_threshold)[None]
    dev = frame_image.device
    if sensor_depth is not None:
        d = sensor_depth.reshape(1, *frame_image.shape[1:]).clone()
    elif depth is None:
        d = 2 * torch.ones(1, *frame_image.shape[1:], device=dev)
        d += torch.randn(d.shape, device=dev, generator=generator) * 0.3
    else:
        d = depth.detach().clone()
        med, std, valid = _median_depth(d, opacity.detach(), valid_rgb)
        if med is None:
            d = 2 * torch.ones_like(d)
        else:
            bad = (d > med + std) | (d < med - std) | ~valid
            d[bad] = med
            d = d + torch.randn(d.shape, device=dev, generator=generator) * torch.where(bad, std * 0.5, std * 0.2)
    d[~valid_rgb] = 0
    return d[0]


def run_sequence(frames, cam, dev, *, sensor_depth: bool = False, kf_interval: int = 5, window_size: int = 8,
                 init_iters: int = 1050, mapping_iters: int = 150, first_order_iters: int = 40,
                 second_order_iters: int = 10, seed: int = 0, config: Optional[dict] = None, log=None,
                 use_first_order_best: bool = True, use_best_loss: bool = True):
    """Tracking + mapping over `frames`; returns a dict with the estimated poses, timings and the
    final map.  `sensor_depth`: insert keyframes from the frames' depth (RGB-D initialisation) instead
    of the monocular prior / rendered depth."""
    H, W = cam.H, cam.W
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    gen = torch.Generator(device=dev).manual_seed(seed)
    bg = torch.zeros(3, device=dev)
    cfg = {"Training": {"window_size": window_size, "monocular": True},
           "Dataset": {"sensor_type": "depth" if sensor_depth else "monocular", "pcd_downsample": 64,
                       "pcd_downsample_init": 32, "point_size": 0.01, "adaptive_pointsize": True}}
    for k, v in (config or {}).items():
        cfg.setdefault(k, {}).update(v)
    gm = GaussianModel(0, config=cfg, device=dev)
    gm.init_lr(6.0)
    gm.training_setup()
    mapper = NativeMapper(gm, bg, config=cfg, cameras_extent=6.0, seed=seed)

    def camera(fr: Frame, T):
        v = ViewCamera(fr.uid, fr.image, T, cam.projmatrix_raw, fovx, fovy, H, W, dev,
                       intrinsics=(cam.fx, cam.fy, cam.cx, cam.cy))
        v.T_gt = fr.T_gt
        return v

    t_track = t_map = 0.0
    n_track_iters = n_map_iters = n_map_views = 0
    cams = {}
    # ---- initialisation (frame 0 fixes the world frame at its ground-truth pose) ----
    f0 = frames[0]
    cams[0] = camera(f0, f0.T_gt.to(dev).float().clone())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d0 = keyframe_depth(f0.image, None, None, f0.depth if sensor_depth else None, gen)
    gm.extend_from_pcd_seq(cams[0], kf_id=0, init=True, depthmap=d0, generator=gen)
    mapper.add_keyframe(0, cams[0])
    mapper.set_window([0])
    mapper.initialize_map(0, iters=init_iters)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    window = [0]
    kf_ids = [0]
    last_kf = 0
    for k in range(1, len(frames)):
        fr = frames[k]
        vp = camera(fr, cams[k - 1].T.detach().clone())           # previous pose (:358-362)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trk = NativeTracker(vp, gm, bg)
        if second_order_iters > 0:
            trk.enable_second_order(stack_dim=16, sketch_dim=64, initial_lambda=1e-3, seed=seed + k)
        # one frame of the reference's loop incl. its best-iterate bookkeeping (slam_frontend.py:455-822;
        # use_first_order_best / use_best_loss as in configs/mono/tum/base_config.yaml:268-273); the
        # tracker's depth / opacity / n_touched buffers end up rendered at the best iterate
        it = trk.run(max_iters=first_order_iters, check_every=10, second_order_iters=second_order_iters,
                     use_first_order_best=use_first_order_best, use_best_loss=use_best_loss)
        torch.cuda.synchronize()
        t_track += time.perf_counter() - t0
        n_track_iters += it
        cams[k] = vp
        if k - last_kf >= kf_interval:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            depth_map = keyframe_depth(fr.image, trk.depth, trk.opacity, fr.depth if sensor_depth else None, gen)
            gm.extend_from_pcd_seq(vp, kf_id=k, init=False, depthmap=depth_map, generator=gen)
            window = ([k] + window)[:window_size]
            kf_ids.append(k)
            last_kf = k
            mapper.add_keyframe(k, vp)
            full = len(window) == window_size
            ba = full and not mapper.initialized            # initial BA: all but one keyframe's pose, 300 its (:440-451)
            mapper.set_window(window, frames_to_optimize=window_size - 1 if ba else None)
            iters = mapping_iters
            mapper.map(window, iters=iters)
            mapper.map(window, prune=True)
            torch.cuda.synchronize()
            t_map += time.perf_counter() - t0
            n_map_iters += iters
            n_map_views += iters * (len(window) + min(2, len(kf_ids) - len(window)))
            if log:
                log(f"keyframe {k}: window {window}, {len(gm)} Gaussians, loss {float(mapper.last_loss):.4f}")
    ok = mapper.check_capacity()
    return {"cameras": cams, "kf_ids": kf_ids, "gaussians": gm, "mapper": mapper, "t_init": t_init,
            "t_track": t_track, "t_map": t_map, "n_track_iters": n_track_iters, "n_map_iters": n_map_iters,
            "n_map_views": n_map_views, "capacity_ok": ok, "frames_tracked": len(frames) - 1}


def evaluate(result, frames, dev, every: int = 1, monocular: bool = True):
    """ATE RMSE over all tracked frames and over the keyframes (Sim(3)-aligned when `monocular`:
    the scale of a monocular map is free; SE(3)-aligned otherwise, eval_utils.py:26-44),
    PSNR of the final map rendered at the estimated poses of the non-keyframes."""
    from . import eval_metrics as E
    from .gaussian_renderer import render
    cams, gm = result["cameras"], result["gaussians"]
    ids = sorted(cams)
    ate_all = E.eval_ate(cams, ids, monocular=monocular)
    ate_kf = E.eval_ate(cams, result["kf_ids"], monocular=monocular) if len(result["kf_ids"]) >= 3 else float("nan")
    gt_c = torch.stack([torch.linalg.inv(frames[i].T_gt.double())[:3, 3] for i in ids])
    path = float((gt_c[1:] - gt_c[:-1]).norm(dim=1).sum())
    bg = torch.zeros(3, device=dev)
    ps = []
    with torch.no_grad():
        for i in ids[::every]:
            if i in result["kf_ids"]:
                continue
            v = cams[i]
            img = render(v, gm, Pipe, bg)["render"]
            img = ((torch.abs(v.exposure_a) + v.exposure_eps) * img + v.exposure_b).clamp(0, 1)
            ps.append(float(E.psnr(img.unsqueeze(0), frames[i].image.unsqueeze(0))))
    return {"ate_rmse_m": ate_all, "ate_rmse_keyframes_m": ate_kf, "path_length_m": path,
            "psnr_db": sum(ps) / max(1, len(ps)), "psnr_frames": len(ps), "gaussians": len(gm)}
