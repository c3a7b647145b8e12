"""2-rank rehearsal of the keyframe-parallel exchange on the GPU box.

The box has ONE GPU, so both ranks share it and the collective runs over gloo with the flat
buffer staged through the host (`FlatGradBucket._reduce`); the device-side work - rasteriser
forward / backward per rank and the pack kernel - is the production path.  Checked: the reduced
gradients and densification statistics equal the single-process sum over the same two views,
and `bench.py --gpus 2` really starts two ranks and prints a 2-rank line with `exchange_ms`.
"""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

N, W, H = 20000, 320, 240


def _view_step(sc, params, rank_view, dev):
    """forward + synthetic loss + backward of one view; returns (means2D grad, radii)."""
    from monogs_amd import rasterizer as R, synthetic as S
    from monogs_amd.parallel import view_pose
    from monogs_amd.tracking_fused import l1_image_depth_loss
    cam = S.make_camera(W, H, view_pose(rank_view))
    st = R.GaussianRasterizationSettings(H, W, cam.tanfovx, cam.tanfovy, sc.bg.to(dev), 1.0,
                                         cam.viewmatrix.to(dev), cam.projmatrix.to(dev),
                                         cam.projmatrix_raw.to(dev), 0, cam.viewmatrix.to(dev), False, False)
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    img, radii, dep, opa, nt = R.GaussianRasterizer(st)(
        means3D=params[0], means2D=m2d, shs=params[4], opacities=params[3], scales=params[1],
        rotations=params[2])
    l1_image_depth_loss(img, dep, sc.gt_image.to(dev), sc.gt_depth.to(dev), 0.05).backward()
    return m2d.grad, radii


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from monogs_amd import synthetic as S
    from monogs_amd.parallel import FlatGradBucket
    sc = S.make_scene(N, W, H, seed=2)
    params = [t.to(dev).requires_grad_() for t in S.activated(sc)]
    g2d, radii = _view_step(sc, params, rank, dev)
    bucket = FlatGradBucket(params)
    stat, denom, max_radii = bucket.all_reduce(g2d, radii)
    torch.cuda.synchronize()
    if rank == 0:
        reduced = [p.grad.clone() for p in params]
        red_stat, red_den, red_rad = stat.clone(), denom.clone(), max_radii.clone()
        # the same two views, one process: autograd accumulates the parameter gradients
        for p in params:
            p.grad = None
        want_stat = torch.zeros(N, device=dev)
        want_den = torch.zeros(N, device=dev)
        want_rad = torch.zeros(N, dtype=torch.int32, device=dev)
        for v in range(world):
            g, r = _view_step(sc, params, v, dev)
            vis = r > 0
            want_stat += torch.where(vis, torch.linalg.norm(g[:, :2], dim=-1), torch.zeros_like(want_stat))
            want_den += vis.float()
            want_rad = torch.maximum(want_rad, r)
        errs = [float((a - p.grad).abs().max() / (p.grad.abs().max() + 1e-30)) for a, p in zip(reduced, params)]
        ret["grad_err"] = max(errs)
        ret["grad_norms"] = [float(p.grad.norm()) for p in params]
        ret["stat_err"] = float((red_stat - want_stat).abs().max() / (want_stat.abs().max() + 1e-30))
        ret["den_ok"] = bool(torch.equal(red_den, want_den))
        ret["rad_ok"] = bool(torch.equal(red_rad, want_rad))
        ret["visible_both"] = int((want_den == 2).sum())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_equals_the_single_process_sum(built):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + ((os.getpid() + 251) % 500)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["visible_both"] > 1000          # the two views overlap: the sum is a real sum
    assert all(n > 0 for n in ret["grad_norms"])
    assert ret["grad_err"] < 1e-6, dict(ret)    # same kernels, fixed summation order: a + b exactly
    assert ret["stat_err"] < 1e-6 and ret["den_ok"] and ret["rad_ok"], dict(ret)


def test_bench_gpus_2_starts_two_ranks(built):
    """`python bench.py --gpus 2` with no launcher around it (the form VERDICT r1 found dead)."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--gaussians", "20000", "--width", "320", "--height", "240", "--profile-steps", "0",
                        "--no-cpu-baseline", "--no-tracking", "--no-slam", "--sustain-seconds", "0",
                        "--mapping-gaussians", "30000", "--mapping-iters", "6"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["views_per_step"] == 2
    # the config-5 leg: NativeMapper.map, Replica-sized RGB-D window of 8 + 2 views sharded over the 2 ranks
    ms = out["mapping_sharded"]
    assert ms["world"] == 2 and ms["views_per_rank"] == [5.0, 5.0] and ms["ranks_with_most_views"] == [0, 1]
    assert ms["mapping_iters_per_s"] > 0 and abs(ms["views_per_s"] - 10 * ms["mapping_iters_per_s"]) < 0.1
    assert ms["exchange_ms"] > 0 and len(ms["compute_ms_per_rank"]) == 2 and all(c > 0 for c in ms["compute_ms_per_rank"])
    exact = (30000 * (3 + 3 + 1 + 3 + 4) + 2 * 30000) * 4 + 30000 * 4         # sections are padded to 64 floats
    assert exact <= ms["exchange_bytes"] <= exact + 8 * 64 * 4 and "1200x680" in ms["workload"]
    mg = out["multi_gpu"]
    assert len(mg["compute_ms_per_rank"]) == 2 and mg["exchange_ms"] > 0 and out["exchange_ms"] == mg["exchange_ms"]
    assert mg["exchange_bytes"] == (20000 * (3 + 3 + 4 + 1 + 3) + 2 * 20000) * 4 + 20000 * 4
    # a launcher whose world disagrees with --gpus is refused instead of silently running 1 rank
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lean"], env=env2, cwd=ROOT,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE" in (r2.stderr + r2.stdout)


# ---------------------------------------------------------------------------------------------
# Keyframe-parallel NATIVE mapping (BASELINE config 5's shape, rehearsed on one GPU): the window's
# views are sharded over 2 ranks, gradients + statistics accumulate on the device straight into the
# flat buffer that is all-reduced, every rank applies the identical optimiser step, densification
# (seeded split noise) and prune decision (all-gathered visibility).  Result must equal the
# single-process NativeMapper on the same window.
def _native_mapping_run(group_world, dev, iters=4, replica=False):
    from monogs_amd.mapping_native import NativeMapper
    from test_gpu_mapping import _window_fixture, replica_window_fixture
    if replica:     # BASELINE config 5's shape: RGB-D, 1200x680, window 8 + 2 old keyframes
        _, gm, views = replica_window_fixture(dev, N=50000, n_kf=10, seed=43)
        cfg = {"Training": {"monocular": False, "window_size": 8, "gaussian_update_every": 3, "gaussian_update_offset": 0}}
        window = [9, 8, 7, 6, 5, 4, 3, 2]
    else:
        _, gm, views = _window_fixture(N=3000, n_views=6, dev=dev, seed=31)
        gm.unique_kfIDs = (torch.arange(3000, device=dev) % 6).to(torch.int32)
        cfg = {"Training": {"window_size": 4, "gaussian_update_every": 3, "gaussian_update_offset": 0}}
        window = [5, 4, 3, 2]
    mp_ = NativeMapper(gm, torch.zeros(3, device=dev), config=cfg)
    for i, v in enumerate(views):
        mp_.add_keyframe(i, v)
    mp_.set_window(window)
    mp_.map(window, iters=iters)          # iteration 3 densifies; views 0, 1 are the random extras
    n_mid = len(gm)
    mp_.map(window, prune=True)
    torch.cuda.synchronize()
    assert mp_.check_capacity()
    out = {"n_mid": n_mid, "n": len(gm), "xyz": gm._xyz.detach().cpu(), "opacity": gm._opacity.detach().cpu(),
           "scaling": gm._scaling.detach().cpu(), "ids": gm.unique_kfIDs.cpu(), "n_obs": gm.n_obs.cpu(),
           "loss": float(mp_.last_loss)}
    for v in views:
        out[f"T{v.uid}"] = v.T.detach().cpu()
        out[f"a{v.uid}"] = float(v.exposure_a)
    return out


def _native_mapping_worker(rank, world, port, ret, replica=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _native_mapping_run(world, dev, replica=replica)
    ret[rank] = {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()}
    dist.barrier()
    dist.destroy_process_group()


def test_native_mapper_sharded_over_two_ranks_matches_single_process(built):
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    single = _native_mapping_run(1, torch.device("cuda", 0))
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + ((os.getpid() + 377) % 500)
    mp.spawn(_native_mapping_worker, args=(2, port, ret), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    # the replicas stay identical ...
    for k in ("n_mid", "n"):
        assert r0[k] == r1[k]
    for k in ("xyz", "opacity", "scaling", "ids", "n_obs"):
        assert np.array_equal(r0[k], r1[k]), k
    # ... and equal to the single-process result: same rows survive, parameters agree up to the
    # summation order of the views (Adam normalises the gradient: compare the bulk)
    assert single["n_mid"] == r0["n_mid"] and single["n"] == r0["n"] and single["n_mid"] != 3000
    assert np.array_equal(single["ids"].numpy(), r0["ids"]) and np.array_equal(single["n_obs"].numpy(), r0["n_obs"])
    for k, tol in (("xyz", 2e-4), ("opacity", 5e-3), ("scaling", 2e-4)):
        close = (np.abs(single[k].numpy() - r0[k]) <= tol).mean()
        assert close > 0.995, (k, close)
    # every view's pose / exposure is stepped by the rank that owns it and published to the others at
    # the end of map(): both ranks hold the whole window
    for uid in range(6):
        assert np.array_equal(r0[f"T{uid}"], r1[f"T{uid}"]) and r0[f"a{uid}"] == r1[f"a{uid}"], uid
        assert np.allclose(r0[f"T{uid}"], single[f"T{uid}"].numpy(), atol=5e-5), uid
        assert abs(r0[f"a{uid}"] - single[f"a{uid}"]) < 2e-4
        moved = not np.allclose(r0[f"T{uid}"], _pose0(uid))
        assert moved == (uid in (5, 4, 3))          # window positions 0..2: pose optimised + update_pose


def test_native_mapper_sharded_at_the_config5_shape(built):
    """The same check at BASELINE config 5's shape (RGB-D, 1200x680 Replica calibration, 50 000
    Gaussians, window of 8 + 2 old keyframes = 5 views per rank on 2 ranks): the sharded replicas stay
    bit-identical, incl. a densification, and equal the single-process NativeMapper."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    single = _native_mapping_run(1, torch.device("cuda", 0), replica=True)
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + ((os.getpid() + 419) % 500)
    mp.spawn(_native_mapping_worker, args=(2, port, ret, True), nprocs=2, join=True)
    r0, r1 = ret[0], ret[1]
    assert r0["n_mid"] == r1["n_mid"] and r0["n"] == r1["n"]
    for k in ("xyz", "opacity", "scaling", "ids", "n_obs"):
        assert np.array_equal(r0[k], r1[k]), k
    assert single["n_mid"] == r0["n_mid"] and single["n"] == r0["n"] and single["n_mid"] != 50000
    assert np.array_equal(single["ids"].numpy(), r0["ids"])
    # n_obs counts views with n_touched > 0, and n_touched counts contributions with T (1 - alpha) > 0.5: a
    # threshold in fp32.  The single process adds the views' gradients in another order than the two
    # ranks do (a + b + c ... vs (a + c ...) + (b + d ...)), so after four Adam steps a borderline
    # Gaussian among the 8 x 19 k (view, Gaussian) verdicts may fall on the other side
    assert (single["n_obs"].numpy() == r0["n_obs"]).mean() > 0.999
    for k, tol in (("xyz", 2e-4), ("opacity", 5e-3), ("scaling", 2e-4)):
        close = (np.abs(single[k].numpy() - r0[k]) <= tol).mean()
        assert close > 0.995, (k, close)
    # the loss scalar is rank-local (this rank's views): the two shards add up to the window's objective
    assert abs(single["loss"] - (r0["loss"] + r1["loss"])) <= 1e-3 * abs(single["loss"])
    for uid in range(10):
        assert np.array_equal(r0[f"T{uid}"], r1[f"T{uid}"]) and r0[f"a{uid}"] == r1[f"a{uid}"], uid
        assert np.allclose(r0[f"T{uid}"], single[f"T{uid}"].numpy(), atol=5e-5), uid


def _pose0(uid):
    from monogs_amd.parallel import view_pose
    return view_pose(uid).numpy()


def test_bench_rccl_backend_with_a_one_rank_group(built):
    """The box has one GPU, so RCCL cannot be run across ranks here; a 1-rank group still takes the
    whole N-rank code path of bench.py on the `nccl` (= RCCL) backend: process group on the device,
    pack + all_reduce(sum) + all_reduce(max), the gathers and the barriers."""
    env = dict(os.environ, MGS_BENCH_FORCE_DIST="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MGS_DIST_BACKEND"):
        env.pop(k, None)
    port = 29500 + ((os.getpid() + 433) % 500)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "3", "--warmup", "1", "--gaussians", "20000", "--width", "320",
                        "--height", "240", "--lean", "--profile-steps", "0"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["multi_gpu"]["backend"] == "rccl" and out["multi_gpu"]["exchange_ms"] > 0
