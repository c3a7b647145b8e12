"""Extra legs of bench.py beyond the headline rasteriser number: mapping iterations/s (row a13)
and the config-4-shaped SLAM surrogate (tracking ms/frame, mapping its/s, ATE, PSNR)."""
from __future__ import annotations

import math
import time

import torch


def _model_from_scene(sc, dev):
    import torch.nn as nn
    from .gaussian_model import GaussianModel
    N = sc.means3D.shape[0]
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
    # Stationary workload: with real learning rates the map drifts away from the SYN-C statistics within
    # a few hundred iterations against random targets (splats fade, D falls, iterations get 2x faster).
    # All learning rates are 0: every kernel runs exactly the same work (Adam included), nothing moves.
    for g in gm.optimizer.param_groups:
        g["lr"] = 0.0
    gm.lr_init = gm.lr_final = 0.0
    return gm


FROZEN_POSE_LR = {"cam_rot_delta": 0.0, "cam_trans_delta": 0.0, "exposure_a": 0.0, "exposure_b": 0.0}
# (densification is switched off as well - "gaussian_update_every" 1e9 with an offset that is never hit; the
# default offset of 50 would densify + prune at iteration 50 and leave a different, much smaller map)


def _spread(rates):
    r = sorted(rates)
    return {"median": round(r[len(r) // 2], 2), "min": round(r[0], 2), "max": round(r[-1], 2), "repeats": len(r)}


def bench_mapping(sc, dev, iters: int = 100, repeats: int = 3):
    """Mapping iterations/s for an 8-view window + 2 old keyframes (slam_backend.py:183-242) on the
    frozen-size SYN-C map (300k Gaussians @ 640x480): the reference-shaped Python body
    (slam_loops.mapping_step: autograd binding, fused loss, FusedGaussianAdam) against the native
    one (mapping_native.NativeMapper: one C-ABI call per view) with 1, 2 (default) and 3 views in
    flight on separate HIP streams.  `iters` iterations per timed run, `repeats` runs per mode:
    median rate + spread."""
    from .mapping_native import NativeMapper
    from .parallel import view_pose
    from .slam_loops import Pipe, ViewCamera, mapping_step
    cam = sc.cam
    H, W, N = cam.H, cam.W, sc.means3D.shape[0]
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    bg = torch.zeros(3, device=dev)
    out = {"views_per_iteration": 10, "iterations_per_run": iters,
           "map": f"SYN-C map, {N} Gaussians @ {W}x{H}, window 8 + 2 old keyframes, all learning rates 0 (stationary)"}
    for mode in ("python", "native_1_stream", "native", "native_3_streams"):
        gm = _model_from_scene(sc, dev)
        views = [ViewCamera(i, sc.gt_image, view_pose(i), cam.projmatrix_raw, fovx, fovy, H, W, dev) for i in range(10)]
        if mode == "python":
            groups = []
            for v in views[1:8]:
                groups += [{"params": [v.cam_rot_delta], "lr": 0.0}, {"params": [v.cam_trans_delta], "lr": 0.0},
                           {"params": [v.exposure_a], "lr": 0.0}, {"params": [v.exposure_b], "lr": 0.0}]
            kopt = torch.optim.Adam(groups)
            cfg = {"Training": {"monocular": True, "rgb_boundary_threshold": 0.01}}

            def it(n=1):
                for _ in range(n):
                    o = mapping_step(views, gm, gm.optimizer, kopt, bg, Pipe, cfg, pose_window=3, fused_loss=True)
                    gm.xyz_gradient_accum += o[1][:, None]
                    gm.denom += o[2][:, None]
                    gm.max_radii2D = torch.maximum(gm.max_radii2D, o[3].float())
        else:
            lanes = {"native_1_stream": 1, "native": 2, "native_3_streams": 3}[mode]
            mp = NativeMapper(gm, bg, config={"Training": {"gaussian_update_every": 10 ** 9, "gaussian_update_offset": 10 ** 9 - 1, "gaussian_reset": 10 ** 9,
                                                           "lr": FROZEN_POSE_LR}}, concurrent_views=lanes)
            for i, v in enumerate(views):
                mp.add_keyframe(i, v)
            mp.set_window(list(range(7, -1, -1)))

            def it(n=1):
                mp.map(iters=n)       # ONE map() call runs n iterations, as the backend does per keyframe
        it(2)
        rates = []
        n_run = iters if mode != "python" else max(10, iters // 4)      # the Python body is ~4x slower
        for _ in range(repeats):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            it(n_run)
            torch.cuda.synchronize()
            rates.append(n_run / (time.perf_counter() - t0))
        sp = _spread(rates)
        out[f"{mode}_iters_per_s"] = sp["median"]
        out[f"{mode}_iters_per_s_spread"] = sp
        out[f"{mode}_ms_per_view"] = round(1e3 / sp["median"] / 10, 4)
        if mode.startswith("native") and not mp.check_capacity():
            raise RuntimeError("native mapping bench overflowed its pair capacity")
        if len(gm) != N:
            raise RuntimeError(f"mapping bench: the map changed size ({N} -> {len(gm)}): not the stated workload")
    return out


def bench_mapping_sharded(dev, rank: int, world: int, backend, n_gaussians: int = 300_000, iters: int = 100,
                          repeats: int = 3, profile_iters: int = 20):
    """BASELINE config 5's iteration through the PRODUCT path: NativeMapper.map on a Replica-sized
    RGB-D window (office0 calibration 1200x680, configs/rgbd/replica/base_config.yaml:12,27-28; window of
    8 keyframes + 2 random old ones, utils/slam_backend.py:183-247), the 10 views sharded round-robin
    over the ranks (view i -> rank i mod world: with 8 ranks, ranks 0 and 1 render a second view),
    gradients accumulated on the device straight into the flat buffer that RCCL all-reduces, the same
    fused Adam step on every replica.  Every rank calls this; the timed region is bracketed by
    barrier + synchronize and the max over ranks is taken.  world == 1 gives the single-GPU figure
    the scaling is judged against.  Returns a dict on rank 0 (None elsewhere)."""
    import torch.distributed as dist
    from . import synthetic as S
    from .mapping_native import NativeMapper
    from .parallel import view_pose
    from .slam_loops import ViewCamera
    W, H = 1200, 680
    sc = S.make_scene(n_gaussians, W, H, seed=0, intrinsics=S.REPLICA_INTRINSICS)
    cam = sc.cam
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    bg = torch.zeros(3, device=dev)
    gm = _model_from_scene(sc, dev)
    views = [ViewCamera(i, sc.gt_image, view_pose(i), cam.projmatrix_raw, fovx, fovy, H, W, dev, gt_depth=sc.gt_depth,
                        intrinsics=(cam.fx, cam.fy, cam.cx, cam.cy)) for i in range(10)]
    mp = NativeMapper(gm, bg, config={"Training": {"monocular": False, "window_size": 8, "gaussian_update_every": 10 ** 9, "gaussian_update_offset": 10 ** 9 - 1,
                                                   "gaussian_reset": 10 ** 9, "lr": FROZEN_POSE_LR}})
    for i, v in enumerate(views):
        mp.add_keyframe(i, v)
    window = list(range(9, 1, -1))
    mp.set_window(window)
    multi = world > 1 or (dist.is_available() and dist.is_initialized())

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not multi:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    mp.map(window, iters=5)
    rates = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        mp.map(window, iters=iters)
        barrier()
        rates.append(iters / max_over_ranks(time.perf_counter() - t0))
    ok = mp.check_capacity()
    if len(gm) != n_gaussians:
        raise RuntimeError(f"sharded mapping bench: the map changed size ({n_gaussians} -> {len(gm)})")
    mp.timing = []
    mp.map(window, iters=profile_iters)
    ts = mp.timing_summary()
    mp.timing = None
    mine = {"rank": rank, "compute_ms": round(ts["compute_ms"], 4), "exchange_ms": round(ts["exchange_ms"], 4),
            "update_ms": round(ts["update_ms"], 4), "views": ts["views_per_iteration"], "capacity_ok": bool(ok),
            "pairs_capacity": int(mp.capacity)}
    gathered = [mine]
    if multi:
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
    if rank != 0:
        return None
    if not all(g["capacity_ok"] for g in gathered):
        raise RuntimeError("sharded mapping bench overflowed its pair capacity")
    sp = _spread(rates)
    flat_bytes = mp.flat.numel() * 4 + mp.radii_max.numel() * 4
    return {"workload": f"Replica-sized RGB-D mapping window: {n_gaussians} Gaussians @ {W}x{H} (office0 calibration), "
                        "8 keyframes + 2 random old keyframes per iteration, NativeMapper.map (2 views in flight per rank), "
                        "all learning rates 0 (every kernel runs, the workload stays stationary)",
            "mapping_iters_per_s": sp["median"], "mapping_iters_per_s_spread": sp,
            "views_per_s": round(sp["median"] * 10, 1), "iterations_per_run": iters, "views_per_iteration": 10,
            "views_per_rank": [g["views"] for g in gathered],
            "ranks_with_most_views": [g["rank"] for g in gathered if g["views"] == max(x["views"] for x in gathered)],
            "compute_ms_per_rank": [g["compute_ms"] for g in gathered],
            "exchange_ms_per_rank": [g["exchange_ms"] for g in gathered],
            "exchange_ms": round(max(g["exchange_ms"] for g in gathered), 4),
            "update_ms_per_rank": [g["update_ms"] for g in gathered],
            "exchange_bytes": flat_bytes, "world": world,
            "exchange": "all_reduce(sum) of the flat fp32 buffer the backward accumulates into (no pack pass) + "
                        "all_reduce(max) of int32 radii; one 18-float row per view published at the end of map()"}


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
