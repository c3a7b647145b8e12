#!/usr/bin/env python3
"""Benchmark of the hot path: rasteriser forward+backward (incl. pose Jacobian) on the
SYN-C workload of BASELINE.md §4 (640x480, 300k Gaussians, SH degree 0).

    python bench.py --gpus N --steps K --warmup W

N = 1: one view per step through the autograd binding (what MonoGS's render() +
loss.backward() exercises).  N > 1 (launched by torch.distributed.run, one rank per
GPU): keyframe-parallel mapping - Gaussians replicated, one view per rank,
one RCCL all-reduce(sum) of the flat Gaussian-gradient buffer per step; value = views/s
over all ranks ("weak" scaling: per-GPU work fixed).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`
(dominant kernel, HIP-event timed inside the library on the launch stream) and
`cpu_baseline` (C++ host emulation on the host cores, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gaussians", type=int, default=300_000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--no-tracking", action="store_true")
    ap.add_argument("--tracking-iters", type=int, default=100)
    return ap.parse_args()


def bench_tracking(sc, dev, iters):
    """Tracking iterations/s (row a12): the loop body of utils/slam_frontend.py:455-751 on
    the frozen SYN-C map - first order (render, Huber/L2, backward, Adam, update_pose; once
    with the reference's PyTorch glue, once with the fused HIP loss / optimiser+update_pose) and
    second order (sketched LM, repeat 1 / stack 16 / sketch 64 as in
    configs/mono/tum/base_config.yaml:256-260).  fr3_office itself is not available offline;
    intrinsics and image size are fr3_office's."""
    import math
    from monogs_amd.gaussian_renderer import render
    from monogs_amd.pose import SE3_exp
    from monogs_amd.slam_loops import (GaussianParams, Pipe, ViewCamera, make_pose_optimizer,
                                       tracking_step_first_order, tracking_step_first_order_fused,
                                       tracking_step_second_order)
    from monogs_amd.tracking_fused import FusedPoseOptimizer
    cam = sc.cam
    H, W = cam.H, cam.W
    gauss = GaussianParams(sc.means3D.to(dev), sc.log_scales.to(dev), sc.rot.to(dev),
                           sc.opacity_logit.to(dev), sc.features_dc.to(dev))
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    bg = torch.zeros(3, device=dev)

    def view(T):
        return ViewCamera(1, torch.zeros(3, H, W), T, cam.projmatrix_raw, fovx, fovy, H, W, dev)

    with torch.no_grad():
        target = render(view(torch.eye(4)), gauss, Pipe, bg)["render"].clone()
    out = {}
    from monogs_amd.tracking_native import NativeTracker
    for mode in ("first_order", "first_order_fused", "first_order_native", "second_order",
                 "second_order_fused", "second_order_native"):
        vp = view(SE3_exp(torch.tensor([0.01, -0.008, 0.006, 0.002, -0.003, 0.002])))
        vp.original_image = target
        vp.rgb_pixel_mask_mapping = (target.sum(0) > 0.01).view(1, H, W)
        opt = make_pose_optimizer(vp)
        fopt = FusedPoseOptimizer(vp)
        gen = torch.Generator(device=dev).manual_seed(0)
        n = iters if mode.startswith("first") else max(10, iters // 4)
        if mode == "first_order_native":
            n = 4 * iters
            trk = NativeTracker(vp, gauss, bg)
        if mode == "second_order_native":
            n = 2 * iters
            trk = NativeTracker(vp, gauss, bg)
            trk.enable_second_order(stack_dim=16, sketch_dim=64, initial_lambda=1e-3)

        def it():
            if mode == "first_order":
                tracking_step_first_order(vp, gauss, opt, bg)
            elif mode == "first_order_fused":
                tracking_step_first_order_fused(vp, gauss, fopt, bg)
            elif mode == "first_order_native":
                trk.step()
            elif mode == "second_order_native":
                trk.step_second_order()
            else:
                tracking_step_second_order(vp, gauss, bg, lambda_=1e-3, repeat_dim=1, stack_dim=16,
                                           sketch_dim=64, generator=gen,
                                           fused_solve=mode.endswith("fused"))
        for _ in range(5):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            it()
        torch.cuda.synchronize()
        out[mode + "_iters_per_s"] = round(n / (time.perf_counter() - t0), 2)
        if mode.endswith("native") and not trk.check_capacity():
            raise RuntimeError("native tracking bench overflowed its pair capacity")
    out["map"] = f"frozen SYN-C map, {sc.means3D.shape[0]} Gaussians @ {W}x{H}"
    return out


def bench_map_update(sc, dev):
    """Map maintenance on the SYN-C map: the Gaussian optimiser step
    (fused HIP launch vs torch.optim.Adam, gaussian_model.py:285) and one densify_and_prune
    (gaussian_model.py:674-691) through the plan + gather kernels."""
    import torch.nn as nn
    from monogs_amd.map_update import FusedGaussianAdam, densify_and_prune
    N = sc.means3D.shape[0]
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    attr = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    src = (sc.means3D, sc.features_dc, sc.features_dc.new_zeros(N, 0, 3), sc.opacity_logit.reshape(N, 1),
           sc.log_scales, sc.rot)
    out = {}
    for kind in ("torch", "fused"):
        class M:
            percent_dense = 0.01
        m = M()
        groups = []
        for n_, a, t in zip(names, attr, src):
            p = nn.Parameter(t.clone().float().to(dev).contiguous())
            setattr(m, a, p)
            groups.append({"params": [p], "lr": 1e-3, "name": n_})
        opt = (torch.optim.Adam(groups, lr=0.0, eps=1e-15) if kind == "torch"
               else FusedGaussianAdam(groups, lr=0.0, eps=1e-15))
        m.optimizer = opt
        grads = [torch.randn_like(g["params"][0]) * 1e-3 for g in groups]

        def step():
            for g, gr in zip(groups, grads):
                g["params"][0].grad = gr
            opt.step()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        out[f"adam_{kind}_us"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
    # densify_and_prune on the fused-optimiser model (statistics chosen so ~10 % clone/split)
    g = torch.Generator(device=dev).manual_seed(0)
    m.xyz_gradient_accum = torch.rand(N, 1, device=dev, generator=g) * 2.2e-4
    m.denom = torch.ones(N, 1, device=dev)
    m.max_radii2D = torch.zeros(N, device=dev)
    m.unique_kfIDs = torch.zeros(N, dtype=torch.int32, device=dev)
    m.n_obs = torch.zeros(N, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    densify_and_prune(m, 2e-4, 0.005, 6.0, 20)
    torch.cuda.synchronize()
    out["densify_and_prune_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
    out["gaussians_before_after"] = [N, int(m._xyz.shape[0])]
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (HIP kernels only; no CPU fallback)")
    # Run autograd's backward on the calling thread.  By default the engine hands GPU nodes to
    # a per-device worker thread; at ~0.37 ms of GPU work per step that hand-off is visible and
    # noisy (measured on one box: 0.38-0.53 ms/step with it, 0.367-0.370 without).  A PyTorch
    # runtime switch, not a change to what is computed (INTEGRATION.md recommends it for MonoGS).
    torch.autograd.set_multithreading_enabled(False)
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MGS_DIST_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:   # functional rehearsal only (e.g. 2 ranks sharing one GPU with gloo)
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as entry
    if not os.path.exists(entry.LIB):
        entry.build()
    from monogs_amd import _cabi, rasterizer as R, synthetic as S
    from monogs_amd.parallel import FlatGradBucket, view_pose
    from monogs_amd.tracking_fused import l1_image_depth_loss

    N, W, H = args.gaussians, args.width, args.height
    sc = S.make_scene(N, W, H, seed=0)
    # every rank renders its own view of the same (replicated) map
    cam = S.make_camera(W, H, view_pose(rank)) if distributed else sc.cam
    m, s, r, o, sh = S.activated(sc)
    params = [t.to(dev).requires_grad_() for t in (m, s, r, o, sh)]
    theta = torch.zeros(3, device=dev, requires_grad=True)
    rho = torch.zeros(3, device=dev, requires_grad=True)
    bg = sc.bg.to(dev)
    st = R.GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, bg, 1.0,
                                         cam.viewmatrix.to(dev), cam.projmatrix.to(dev),
                                         cam.projmatrix_raw.to(dev), 0, cam.viewmatrix.to(dev),
                                         False, False)
    ras = R.GaussianRasterizer(st)
    gt_img, gt_dep = sc.gt_image.to(dev), sc.gt_depth.to(dev)
    bucket = FlatGradBucket(params) if distributed else None

    def step(exchange=True):
        for p in params:
            p.grad = None
        theta.grad = None
        rho.grad = None
        m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
        img, radii, dep, opa, nt = ras(means3D=params[0], means2D=m2d, shs=params[4],
                                       opacities=params[3], scales=params[1],
                                       rotations=params[2], theta=theta, rho=rho)
        # L = mean|image - G| + 0.05 mean|depth - Gd|, fused HIP loss kernels
        loss = l1_image_depth_loss(img, dep, gt_img, gt_dep, 0.05)
        loss.backward()
        if bucket is not None and exchange:
            bucket.all_reduce(m2d.grad, radii)
        return loss

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], dtype=torch.float64,
                            device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * args.steps / dt          # views (frames) per second, whole job
    D = int(R.last_stats["pairs"])

    # ---- per-kernel timing (separate pass; events slow the stream down slightly) ----
    roofline = None
    kernels = {}
    if rank == 0 and args.profile_steps > 0:
        torch.cuda.synchronize()
        _cabi.profile_enable(True)
        for _ in range(args.profile_steps):
            step(exchange=False)   # rank-0-only pass: no collectives here
        torch.cuda.synchronize()
        prof = _cabi.profile_read()
        _cabi.profile_enable(False)
        kernels = {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items()}  # us / launch
        # algorithmic bytes per launch of each kernel (SURVEY §8d split)
        HW = W * H
        alg = {
            "preprocess": 56 * N + 48 * N + 8 * N,
            "bin_count": 32 * N + 4 * N,
            "bin_emit": 32 * N + 12 * D,
            "tile_sort": 24 * D,
            "blend_fwd": 52 * D + 28 * HW,
            "blend_bwd": 52 * D + 40 * D + 24 * HW,
            "preprocess_bwd": 56 * N + 40 * D + 48 * N + 68 * N,
        }
        dom = max((k for k in kernels if k in alg), key=lambda k: kernels[k])
        achieved = alg[dom] / (kernels[dom] * 1e-6) / 1e9
        # HBM traffic per launch from the committed PMC pass (rocprofv3 cannot run inside
        # this process); only valid for the default workload it was collected on
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if (N, W, H) == (300_000, 640, 480) and os.path.exists(tpath):
            traffic = json.load(open(tpath))["bytes_per_launch"].get(dom)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                    "algorithmic_bytes": alg[dom], "avg_us": kernels[dom]}

    # ---- CPU baseline: the C++ host emulation on the host cores (rank 0, N = 1) ----
    cpu_baseline = None
    if rank == 0 and not distributed and not args.no_cpu_baseline:
        from oracle import torch_raster as O
        from oracle.host_emul import HostEmul
        st_c = O.RasterSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg, 1.0, cam.viewmatrix,
                                cam.projmatrix, cam.projmatrix_raw, 0, cam.viewmatrix, False, False)
        em = HostEmul()
        gi = torch.sign(torch.randn(3, H, W)) / (3 * H * W)
        gd = 0.05 * torch.sign(torch.randn(1, H, W)) / (H * W)
        reps, t_cpu = 0, 0.0
        while t_cpu < 10.0 and reps < 50:
            t1 = time.perf_counter()
            em.forward(st_c, m, sh, None, o, s, r, None, exact_cull=False)
            em.backward(gi, gd)
            t_cpu += time.perf_counter() - t1
            reps += 1
        cpu_baseline = {"value": round(reps / t_cpu, 3), "unit": "frames/s",
                        "cores": em.num_threads(), "kind": "port",
                        "sample": f"{reps} fwd+bwd of the same SYN-C workload "
                                  f"({N} Gaussians @ {W}x{H}), OpenMP C++ host emulation "
                                  "(oracle/host_emul.cpp)"}

    # ---- tracking iterations/s on a frozen synthetic map (second BASELINE metric) ----
    tracking = None
    if rank == 0 and not distributed and not args.no_tracking:
        tracking = bench_tracking(sc, dev, args.tracking_iters)

    map_update = None
    if rank == 0 and not distributed and not args.no_tracking:
        map_update = bench_map_update(sc, dev)

    if rank == 0:
        out = {
            "metric": "rasteriser fwd+bwd fps @640x480/300k Gaussians",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"SYN-C: {N} Gaussians @ {W}x{H}, SH degree 0, fwd+bwd incl. "
                                   "pose Jacobian through the autograd binding",
                       "pairs_D": D, "views_per_step": world, "autograd_multithreading": False,
                       "parallelism": f"keyframe-parallel x{world}" if distributed else "single view"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels_us": kernels,
            "tracking": tracking, "map_update": map_update,
        }
        print(json.dumps(out))
    if distributed:
        dist.barrier()      # rank 0 has finished its (collective-free) profiling pass
        dist.destroy_process_group()


if __name__ == "__main__":
    import faulthandler
    import traceback
    faulthandler.enable()
    try:
        main()
    except BaseException:
        sys.stderr.write(f"[rank {os.environ.get('RANK', '0')}] bench.py failed:\n{traceback.format_exc()}\n")
        sys.stderr.flush()
        raise
