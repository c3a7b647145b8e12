"""Extra legs of bench.py beyond the headline rasteriser number: mapping iterations/s (row a13)
and the config-4-shaped SLAM surrogate (tracking ms/frame, mapping its/s, ATE, PSNR)."""
from __future__ import annotations

import math
import time

import torch


def bench_mapping(sc, dev, iters: int = 10):
    """Mapping iterations/s for an 8-view window + 2 old keyframes (slam_backend.py:183-242) on the
    frozen-size SYN-C map (300k Gaussians @ 640x480): the reference-shaped Python body
    (slam_loops.mapping_step: autograd binding, fused loss, FusedGaussianAdam) against the native
    one (mapping_native.NativeMapper: one C-ABI call per view) with 1, 2 (default) and 3 views in
    flight on separate HIP streams."""
    import torch.nn as nn
    from .gaussian_model import GaussianModel
    from .mapping_native import NativeMapper
    from .parallel import view_pose
    from .slam_loops import Pipe, ViewCamera, mapping_step
    cam = sc.cam
    H, W, N = cam.H, cam.W, sc.means3D.shape[0]
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    bg = torch.zeros(3, device=dev)
    out = {"views_per_iteration": 10, "map": f"SYN-C map, {N} Gaussians @ {W}x{H}, window 8 + 2 old keyframes"}
    for mode in ("python", "native_1_stream", "native", "native_3_streams"):
        gm = GaussianModel(0, device=dev)
        gm._xyz = nn.Parameter(sc.means3D.to(dev).contiguous())
        gm._features_dc = nn.Parameter(sc.features_dc.to(dev).contiguous())
        gm._features_rest = nn.Parameter(torch.zeros(N, 0, 3, device=dev))
        gm._scaling = nn.Parameter(sc.log_scales.to(dev).contiguous())
        gm._rotation = nn.Parameter(sc.rot.to(dev).contiguous())
        gm._opacity = nn.Parameter(sc.opacity_logit.to(dev).contiguous())
        gm.max_radii2D = torch.zeros(N, device=dev)
        gm.unique_kfIDs = torch.zeros(N, dtype=torch.int32, device=dev)
        gm.n_obs = torch.zeros(N, dtype=torch.int32, device=dev)
        gm.init_lr(6.0)
        gm.training_setup()
        views = [ViewCamera(i, sc.gt_image, view_pose(i), cam.projmatrix_raw, fovx, fovy, H, W, dev) for i in range(10)]
        if mode == "python":
            groups = []
            for v in views[1:8]:
                groups += [{"params": [v.cam_rot_delta], "lr": 0.0015}, {"params": [v.cam_trans_delta], "lr": 0.0005},
                           {"params": [v.exposure_a], "lr": 0.02}, {"params": [v.exposure_b], "lr": 0.02}]
            kopt = torch.optim.Adam(groups)
            cfg = {"Training": {"monocular": True, "rgb_boundary_threshold": 0.01}}

            def it():
                o = mapping_step(views, gm, gm.optimizer, kopt, bg, Pipe, cfg, pose_window=3, fused_loss=True)
                gm.xyz_gradient_accum += o[1][:, None]
                gm.denom += o[2][:, None]
                gm.max_radii2D = torch.maximum(gm.max_radii2D, o[3].float())
        else:
            lanes = {"native_1_stream": 1, "native": 2, "native_3_streams": 3}[mode]
            mp = NativeMapper(gm, bg, config={"Training": {"gaussian_update_every": 10 ** 9, "gaussian_reset": 10 ** 9}},
                              concurrent_views=lanes)
            for i, v in enumerate(views):
                mp.add_keyframe(i, v)
            mp.set_window(list(range(7, -1, -1)))

            def it(n=1):
                mp.map(iters=n)       # ONE map() call runs n iterations, as the backend does per keyframe
        for _ in range(2):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "python":
            for _ in range(iters):
                it()
        else:
            it(iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"{mode}_iters_per_s"] = round(iters / dt, 2)
        out[f"{mode}_ms_per_view"] = round(dt / iters / 10 * 1e3, 4)
        if mode.startswith("native") and not mp.check_capacity():
            raise RuntimeError("native mapping bench overflowed its pair capacity")
    return out


def bench_slam_surrogate(dev, n_frames: int = 41, **kw):
    """BASELINE config 4's shape on a synthetic sequence (slam_surrogate.py): fr3_office intrinsics,
    640x480, the reference's tracking / mapping budgets."""
    from . import slam_surrogate as SS
    frames, cam, source = SS.load_sequence(n_frames, 640, 480, dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = SS.run_sequence(frames, cam, dev, **kw)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ev = SS.evaluate(res, frames, dev)
    nt = max(1, res["frames_tracked"])
    return {"source": source, "frames": len(frames), "keyframes": len(res["kf_ids"]),
            "tracking_ms_per_frame": round(res["t_track"] / nt * 1e3, 2),
            "tracking_iters_per_s": round(res["n_track_iters"] / max(res["t_track"], 1e-9), 1),
            "mapping_iters_per_s": round(res["n_map_iters"] / max(res["t_map"], 1e-9), 1),
            "mapping_views_per_s": round(res["n_map_views"] / max(res["t_map"], 1e-9), 1),
            "init_s": round(res["t_init"], 3), "wall_s": round(wall, 2),
            "fps_total": round(len(frames) / wall, 2),
            "ate_rmse_m": round(ev["ate_rmse_m"], 5), "ate_rmse_keyframes_m": round(ev["ate_rmse_keyframes_m"], 5),
            "path_length_m": round(ev["path_length_m"], 4), "psnr_db": round(ev["psnr_db"], 2),
            "gaussians": ev["gaussians"], "capacity_ok": bool(res["capacity_ok"]),
            "budgets": "tracking 40 first-order + 10 second-order its/frame, window 8, 150 mapping its/keyframe, "
                       "1050 init its (configs/mono/tum/base_config.yaml)"}
