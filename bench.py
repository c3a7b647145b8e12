#!/usr/bin/env python3
"""Benchmark of the hot path: rasteriser forward+backward (incl. pose Jacobian) on the
SYN-C workload of BASELINE.md §4 (640x480, 300k Gaussians, SH degree 0).

    python bench.py --gpus N --steps K --warmup W [--workload synC|replica]

N = 1: one view per step through the autograd binding (what MonoGS's render() +
loss.backward() exercises).

N > 1: keyframe-parallel mapping - Gaussians replicated, one view per rank,
one all-reduce(sum) of the flat Gaussian-gradient buffer + one all-reduce(max) of the radii
per step; value = views/s over all ranks ("weak" scaling: per-GPU work fixed) through the same
autograd binding as N = 1, so that the driver's efficiency compares like with like.  A second leg,
`mapping_sharded`, times BASELINE config 5 through the PRODUCT path: NativeMapper.map on a
Replica-sized RGB-D window (1200x680, 8 keyframes + 2 old ones) with its 10 views sharded over the
ranks (bench_legs.bench_mapping_sharded: mapping_iters_per_s, views_per_s, exchange_ms, per-rank
compute, which ranks held the most views); at N = 1 the same leg runs unsharded (`mapping_replica`).  Launched
either by the driver (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N
...`: RANK / LOCAL_RANK / WORLD_SIZE come from the environment) or directly
(`python bench.py --gpus N`): with WORLD_SIZE unset this process starts the N ranks itself
as a CHILD `torch.distributed.run` BEFORE touching the GPU and exits with the child's code.
Backend: "nccl" (= RCCL) when the node has >= N GPUs, otherwise a gloo rehearsal in which the
ranks share the GPU(s) and the flat buffer is staged through host memory (functional check of
the exchange on a 1-GPU box; not a performance number - the JSON says which one ran).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant
kernel, HIP-event timed inside the library on the launch stream) and `cpu_baseline`
(N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (width, height, intrinsics or None = fr3_office scaled, description)
    "synC": (640, 480, None, "SYN-C"),
    "replica": (1200, 680, (600.0, 600.0, 599.5, 339.5), "Replica-sized (office0 calibration)"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="synC")
    ap.add_argument("--gaussians", type=int, default=300_000)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=20)
    ap.add_argument("--no-tracking", action="store_true")
    ap.add_argument("--tracking-iters", type=int, default=100)
    ap.add_argument("--no-mapping", action="store_true")
    ap.add_argument("--mapping-gaussians", type=int, default=None,
                    help="map size of the sharded Replica-sized mapping leg (default: --gaussians)")
    ap.add_argument("--mapping-iters", type=int, default=100)
    ap.add_argument("--no-slam", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=2.0)
    ap.add_argument("--lean", action="store_true",
                    help="headline + roofline only (for rocprofv3 runs): no cpu baseline, tracking, "
                         "mapping, SLAM or sustained legs")
    args = ap.parse_args(argv)
    if args.lean:
        args.no_cpu_baseline = args.no_tracking = args.no_mapping = args.no_slam = True
        args.sustain_seconds = 0.0
    return args


def csrc_tree_hash():
    """sha256 over the kernel sources (monogs_amd/csrc/*, include/monogs_raster.h; names and contents without
    comments and white space, sorted):
    what profiles/pmc_traffic.json was collected on must be what this run executes, or `roofline.traffic`
    is null (profiles/collect.sh stores the hash at collection)."""
    import hashlib
    import re
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "monogs_amd", "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(ROOT, "include", "monogs_raster.h"))
    for f in files:
        src = open(f, encoding="utf-8", errors="replace").read()
        # comments and white space do not change the kernels: a reworded comment must not void the traffic figure
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        src = re.sub(r"//[^\n]*", " ", src)
        src = " ".join(src.split())
        h.update(os.path.basename(f).encode())
        h.update(src.encode())
    return h.hexdigest()[:16]


def spawn_ranks(args, argv):
    """`bench.py --gpus N` without a launcher: start N ranks as a child torch.distributed.run.
    Nothing in THIS process has initialised the GPU (no torch.cuda call so far), and the program
    is not replaced (no exec): the child is waited for and its exit code returned."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def bench_tracking(sc, dev, iters):
    """Tracking iterations/s (row a12): the loop body of utils/slam_frontend.py:455-751 on
    the frozen SYN-C map - first order (render, Huber/L2, backward, Adam, update_pose; once
    with the reference's PyTorch glue, once with the fused HIP loss / optimiser+update_pose) and
    second order (sketched LM, repeat 1 / stack 16 / sketch 64 as in
    configs/mono/tum/base_config.yaml:256-260).  fr3_office itself is not available offline;
    intrinsics and image size are fr3_office's."""
    import math
    import torch
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
    import torch
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


def bench_knn(dev):
    """simple-knn's distCUDA2 (row a10; gaussian_model.py:185-191) at the sizes MonoGS calls it with: 640x480 / 64 =
    4 800 points per keyframe (pcd_downsample 64), / 32 = 9 600 at initialisation, Replica 1200x680 / 64 = 12 750 and
    / 32 = 25 500.  Exact tiled brute force: P (P - 1) distance evaluations per call.  HIP-event timed
    around 20 calls (both launches of a call), scratch allocated outside the timed region."""
    import ctypes as C
    import torch
    from monogs_amd import _cabi
    lib = _cabi.lib()
    g = torch.Generator().manual_seed(0)
    out = {}
    for P in (4800, 9600, 12750, 25500):
        pts = (torch.rand(P, 3, generator=g) * torch.tensor([4.0, 3.0, 5.0])).to(dev)
        res = torch.empty(P, device=dev)
        scratch = torch.empty(int(lib.mgs_knn_scratch_bytes(P)), dtype=torch.uint8, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        def call():
            _cabi.check(lib.mgs_knn_dist2(pts.data_ptr(), P, res.data_ptr(), scratch.data_ptr(), stream), "mgs_knn_dist2")
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out[str(P)] = {"knn_us": round(us, 1), "distance_evals_per_s": round(P * (P - 1) / (us * 1e-6), -6)}
    return out


def exchange_model(world, nbytes):
    """DESIGN.md section 6's model of the per-iteration exchange (all-reduce(sum) of the flat fp32 buffer + all-reduce(max)
    of the radii, `nbytes` in all) on the fully connected xGMI mesh (7 links x ~153 GB/s per GPU): a direct
    reduce-scatter + all-gather moves nbytes / N per link and phase, a ring is bound by one link; both plus ~0.04 ms
    of launch + synchronisation of the two collectives.  Printed beside the measured exchange_ms so that a SCALE record
    judges itself."""
    if world < 2:
        return None
    link = 153e9
    direct = 2 * nbytes / world / link * 1e3
    ring = 2 * (world - 1) / world * nbytes / link * 1e3
    return {"direct_ms": round(direct + 0.04, 4), "ring_ms": round(ring + 0.04, 4), "bytes": int(nbytes),
            "link_GBps": 153.0, "fixed_ms": 0.04,
            "reading": "exchange_ms near direct_ms: RCCL used the mesh; near ring_ms: a ring - the remedy is a hand-rolled "
                       "reduce-scatter / all-gather over the 7 links (DESIGN.md section 6)"}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        # no launcher: become the launcher (before any GPU call in this process)
        raise SystemExit(spawn_ranks(args, argv))
    world = int(env_world or "1")
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"--nproc-per-node {args.gpus} (or drop the launcher: bench.py starts the ranks itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MGS_BENCH_FORCE_DIST=1: run the N-rank code path (process group, exchange, gathers) with whatever
    # world the launcher gave, even 1 - the way to exercise the RCCL backend on a 1-GPU box
    distributed = world > 1 or os.environ.get("MGS_BENCH_FORCE_DIST") == "1"

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (HIP kernels only; no CPU fallback)")
    # Run autograd's backward on the calling thread.  By default the engine hands GPU nodes to
    # a per-device worker thread; at ~0.37 ms of GPU work per step that hand-off is visible and
    # noisy (measured on one box: 0.38-0.53 ms/step with it, 0.367-0.370 without).  A PyTorch
    # runtime switch, not a change to what is computed (INTEGRATION.md recommends it for MonoGS);
    # the mode is calibrated below and both figures are reported (`autograd_engine_calibration_fps`).
    torch.autograd.set_multithreading_enabled(False)
    ndev = max(1, torch.cuda.device_count())
    shared_gpu = world > 1 and ndev < world
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" IS RCCL on ROCm; it needs one GPU per rank
        backend = os.environ.get("MGS_DIST_BACKEND", "gloo" if shared_gpu else "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:   # functional rehearsal only (ranks sharing a GPU, buffers staged through the host)
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as entry
    if not os.path.exists(entry.LIB):
        entry.build_product()
    from monogs_amd import _cabi, rasterizer as R, synthetic as S
    from monogs_amd.parallel import FlatGradBucket, view_pose
    from monogs_amd.tracking_fused import l1_image_depth_loss_backward

    wW, wH, intr, wname = WORKLOADS[args.workload]
    N, W, H = args.gaussians, args.width or wW, args.height or wH
    if (W, H) != (wW, wH):
        intr = None
    sc = S.make_scene(N, W, H, seed=0, intrinsics=intr)
    # every rank renders its own view of the same (replicated) map
    cam = S.make_camera(W, H, view_pose(rank), intrinsics=intr) if distributed else sc.cam
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
    if bucket is not None and world == 1:
        bucket.force_collective = True
    ex_events = []

    def step(exchange=True, timed_exchange=False):
        for p in params:
            p.grad = None
        theta.grad = None
        rho.grad = None
        m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
        img, radii, dep, opa, nt = ras(means3D=params[0], means2D=m2d, shs=params[4],
                                       opacities=params[3], scales=params[1],
                                       rotations=params[2], theta=theta, rho=rho)
        # L = mean|image - G| + 0.05 mean|depth - Gd|: value and gradients in ONE fused
        # HIP launch, gradients handed to autograd (what loss.backward() would propagate)
        loss = l1_image_depth_loss_backward(img, dep, gt_img, gt_dep, 0.05)
        if bucket is not None and exchange:
            if timed_exchange:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            bucket.all_reduce(m2d.grad, radii)     # pack (1 launch) + all-reduce(sum) + all-reduce(max)
            if timed_exchange:
                e1.record()
                ex_events.append((e0, e1))
        return loss

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(k, **kw):
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step(**kw)
        barrier()
        return time.perf_counter() - t0

    def max_over_ranks(x):
        if not distributed:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    # Which autograd engine mode is faster depends on the host (round 1: backward on the calling
    # thread was steadier; round 2: on some boxes the default worker-thread engine wins by 10 %).
    # Both are PyTorch runtime switches, not changes to what is computed: calibrate untimed, time
    # the K steps in the faster mode, report which one and both calibration figures.
    calib = {}
    if not distributed:
        for mode in (False, True):
            torch.autograd.set_multithreading_enabled(mode)
            for _ in range(5):
                step()
            calib[mode] = 40 / timed(40)
        engine_mt = calib[True] > 1.03 * calib[False]
        torch.autograd.set_multithreading_enabled(engine_mt)
    else:
        engine_mt = False
    dt = max_over_ranks(timed(args.steps))
    ms_per_step = dt / args.steps * 1e3
    value = world * args.steps / dt          # views (frames) per second, whole job
    # what a MonoGS user gets without touching the autograd engine setting: the same K steps with
    # PyTorch's default (worker-thread) engine, first-class beside the calibrated figure
    value_default_engine = None
    if not distributed:
        if engine_mt:
            value_default_engine = value
        else:
            torch.autograd.set_multithreading_enabled(True)
            for _ in range(5):
                step()
            value_default_engine = world * args.steps / timed(args.steps)
            torch.autograd.set_multithreading_enabled(engine_mt)
    D = int(R.last_stats["pairs"])

    # ---- multi-rank extras: exchange cost and per-rank compute, measured separately ----
    multi = None
    if distributed:
        k = min(args.steps, 50)
        t_compute = timed(k, exchange=False) / k * 1e3           # this rank's fwd+loss+bwd alone
        timed(k, timed_exchange=True)
        ex_ms = sum(a.elapsed_time(b) for a, b in ex_events) / max(1, len(ex_events))
        gathered = [None] * world
        props = torch.cuda.get_device_properties(dev)
        bus = "%04x:%02x:%02x" % tuple(getattr(props, k, 0) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        dist.all_gather_object(gathered, {"rank": rank, "compute_ms": round(t_compute, 4),
                                          "exchange_ms": round(ex_ms, 4), "pairs_D": D,
                                          "device": torch.cuda.get_device_name(dev), "pci_bus": bus,
                                          "uuid": str(getattr(props, "uuid", "")), "local_rank": local_rank,
                                          "host": socket.gethostname()})
        flat_bytes = bucket.flat.numel() * 4 + bucket.radii.numel() * 4
        backend_name = "rccl" if backend == "nccl" else backend
        if shared_gpu:
            backend_name += " (host-staged rehearsal, ranks share a GPU: not a performance number)"
        multi = {"backend": backend_name,
                 # proof that N ranks ran on N distinct GPUs: what the process group reports and each rank's device
                 "ranks_seen": dist.get_world_size(),
                 "devices_per_rank": [f"{g['device']} @ {g['pci_bus']} (cuda:{g['local_rank']}, {g['host']})" for g in gathered],
                 "distinct_gpus": len({(g["host"], g["pci_bus"], g["uuid"]) for g in gathered}),
                 "exchange_ms": round(max(g["exchange_ms"] for g in gathered), 4),
                 "compute_ms_per_rank": [g["compute_ms"] for g in gathered],
                 "exchange_ms_per_rank": [g["exchange_ms"] for g in gathered],
                 "pairs_D_per_rank": [g["pairs_D"] for g in gathered],
                 "exchange_bytes": flat_bytes,
                 "exchange": "1 pack launch + all_reduce(sum) of the flat fp32 gradient+statistics buffer "
                             "+ all_reduce(max) of int32 radii"}

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
        # HBM traffic per launch from the committed PMC pass of THIS round's kernels (rocprofv3
        # cannot run inside this process; profiles/collect.sh regenerates the file); only valid
        # for the default workload it was collected on
        traffic, traffic_source, valu = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        here = csrc_tree_hash()
        if (N, W, H) != (300_000, 640, 480):
            traffic_source = "none: the committed PMC pass covers the default workload only"
        elif not os.path.exists(tpath):
            traffic_source = "none: profiles/pmc_traffic.json absent"
        else:
            tj = json.load(open(tpath))
            if tj.get("csrc_tree_hash") == here:
                traffic = tj["bytes_per_launch"].get(dom)
                valu = (tj.get("valu") or {}).get(dom)
                traffic_source = (f"profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                  f"profiles/collect.sh on these kernel sources (csrc hash {here})")
            else:
                traffic_source = (f"none: profiles/pmc_traffic.json was collected on kernel sources "
                                  f"{tj.get('csrc_tree_hash', '(unrecorded)')}, this run executes {here} - stale, "
                                  "re-run profiles/collect.sh")
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(achieved / 8000.0, 4), "traffic": traffic,
                    "traffic_source": traffic_source,
                    "algorithmic_bytes": alg[dom], "avg_us": kernels[dom],
                    # what actually bounds the kernel (BASELINE defines `bound` / `frac` against HBM; the kernel is bound
                    # by its VALU instruction stream): from the same PMC collection, same source hash, or null
                    "valu_busy": None if valu is None else valu["valu_busy"],
                    "valu_instructions": None if valu is None else valu["SQ_INSTS_VALU"],
                    "valu_note": None if valu is None else
                    "SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * kernel cycles): VALU issue cycles summed over the waves per SIMD "
                    "cycle; ~1 = a SIMD always has a wave issuing VALU (a lone wave issues one per ~4.4 cycles)"}

    # ---- single-GPU extras: sustained rate and the default autograd engine ----
    extras = {}
    if rank == 0 and not distributed and args.sustain_seconds > 0:
        k, t_run = 0, 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while t_run < args.sustain_seconds:
            for _ in range(100):
                step()
            torch.cuda.synchronize()
            k += 100
            t_run = time.perf_counter() - t0
        extras["sustained_fps"] = {"value": round(k / t_run, 1), "seconds": round(t_run, 2), "steps": k}
    if calib:
        extras["autograd_engine_calibration_fps"] = {"calling_thread": round(calib[False], 1),
                                                     "default_worker_thread": round(calib[True], 1)}

    # ---- CPU baseline (rank 0, N = 1): (a) the C++ host emulation on the same workload on the
    # host cores, (b) BASELINE.md §3's row: the fp32 PyTorch oracle on SYN-A (5k @ 160x120) ----
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
                        "sample": f"{reps} fwd+bwd of the same workload "
                                  f"({N} Gaussians @ {W}x{H}), OpenMP C++ host emulation "
                                  "(oracle/host_emul.cpp)"}
        sa = S.make_scene(5000, 160, 120, seed=0)
        ma, sa_s, ra, oa, sha = S.activated(sa)
        ca = sa.cam
        st_a = O.RasterSettings(ca.H, ca.W, ca.tanfovx, ca.tanfovy, sa.bg, 1.0, ca.viewmatrix, ca.projmatrix,
                                ca.projmatrix_raw, 0, ca.viewmatrix, False, False)
        times = []
        for i in range(6):      # 1 warm-up + 5 timed
            L = [t.clone().requires_grad_() for t in (ma, sa_s, ra, oa, sha)]
            th_c, rh_c = torch.zeros(3, requires_grad=True), torch.zeros(3, requires_grad=True)
            t1 = time.perf_counter()
            oi, _, od, _, _, _ = O.rasterize(L[0], None, L[4], None, L[3], L[1], L[2], None, st_a, th_c, rh_c)
            S.synthetic_loss(oi, od, sa).backward()
            times.append(time.perf_counter() - t1)
        med = sorted(times[1:])[2]
        cpu_baseline["torch_oracle_syn_a"] = {
            "value": round(1.0 / med, 3), "unit": "frames/s", "cores": torch.get_num_threads(),
            "host_cpus": os.cpu_count(), "kind": "port",
            "sample": "median of 5 fwd+bwd (autograd) of the fp32 PyTorch oracle on SYN-A "
                      "(5000 Gaussians @ 160x120, BASELINE config 1), 1 warm-up"}

    # ---- tracking iterations/s on a frozen synthetic map (second BASELINE metric) ----
    tracking = map_update = mapping = slam = None
    single = rank == 0 and not distributed and (N, W, H) == (300_000, 640, 480)
    if single and not args.no_tracking:
        tracking = bench_tracking(sc, dev, args.tracking_iters)
        map_update = bench_map_update(sc, dev)
    mapping_replica = mapping_sharded = None
    if single and not args.no_mapping:
        from monogs_amd.bench_legs import bench_mapping, bench_mapping_sharded
        mapping = bench_mapping(sc, dev)
        mapping_replica = bench_mapping_sharded(dev, 0, 1, None, args.mapping_gaussians or N, args.mapping_iters)
    if distributed and not args.no_mapping:
        # config 5 through the product path; EVERY rank takes part (collectives inside)
        from monogs_amd.bench_legs import bench_mapping_sharded
        mapping_sharded = bench_mapping_sharded(dev, rank, world, backend, args.mapping_gaussians or N, args.mapping_iters)
        if mapping_sharded is not None:
            mapping_sharded["backend"] = "rccl" if backend == "nccl" else f"{backend} (host-staged rehearsal)"
    if single and not args.no_slam:
        from monogs_amd.bench_legs import bench_slam_surrogate
        slam = bench_slam_surrogate(dev)
    knn = bench_knn(dev) if (rank == 0 and not distributed and not args.lean) else None

    if rank == 0:
        out = {
            "metric": "rasteriser fwd+bwd fps @640x480/300k Gaussians",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{wname}: {N} Gaussians @ {W}x{H}, SH degree 0, fwd+bwd incl. "
                                   "pose Jacobian through the autograd binding",
                       "pairs_D": D, "views_per_step": world, "autograd_multithreading": bool(engine_mt),
                       # the autograd engine mode `value` was timed in (the faster of an untimed calibration;
                       # `value_default_engine` = PyTorch's default, what an untouched MonoGS process gets)
                       "value_engine_mode": "default_worker_thread" if engine_mt else "calling_thread",
                       "parallelism": f"keyframe-parallel x{world}" if distributed else "single view"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "kernels_us": kernels,
        }
        if multi is not None:
            out["multi_gpu"] = multi
            out["exchange_ms"] = multi["exchange_ms"]
            out["exchange_model"] = exchange_model(world, multi["exchange_bytes"])
            if mapping_sharded is not None and "exchange_bytes" in mapping_sharded:
                mapping_sharded["exchange_model"] = exchange_model(world, mapping_sharded["exchange_bytes"])
        if value_default_engine is not None:
            out["value_default_engine"] = round(value_default_engine, 2)
        out.update(extras)
        for k, v in (("tracking", tracking), ("map_update", map_update), ("mapping", mapping),
                     ("mapping_replica", mapping_replica), ("mapping_sharded", mapping_sharded), ("slam", slam),
                     ("knn", knn)):
            if v is not None:
                out[k] = v
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()      # rank 0 has finished its (collective-free) profiling pass
        dist.destroy_process_group()


if __name__ == "__main__":
    import faulthandler
    import traceback
    faulthandler.enable()
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        sys.stderr.write(f"[rank {os.environ.get('RANK', '0')}] bench.py failed:\n{traceback.format_exc()}\n")
        sys.stderr.flush()
        raise
