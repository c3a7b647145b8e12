"""Per-kernel times of the native tracking iterations (row a12) and of the native mapping view
iteration (row a13) on the GPU box:

    python profiles/tracking_profile.py [gaussians ...]      # default: 300000 8000

For every map size: wall time per iteration of NativeTracker.step() (first order) and
step_second_order() (sketched LM), and the per-kernel averages from the library's own HIP-event
log (mgs_profile_enable / mgs_profile_read: events on the launch stream).  SYN-C-shaped frozen map
at 640x480 (fr3_office intrinsics), target = a render of the same map from the identity pose."""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monogs_amd import _cabi, synthetic as S  # noqa: E402
from monogs_amd.gaussian_renderer import render  # noqa: E402
from monogs_amd.pose import SE3_exp  # noqa: E402
from monogs_amd.slam_loops import GaussianParams, Pipe, ViewCamera  # noqa: E402
from monogs_amd.tracking_native import NativeTracker  # noqa: E402


def profile(step, warm, timed, prof):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(timed):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / timed * 1e6
    _cabi.profile_enable(True)
    for _ in range(prof):
        step()
    torch.cuda.synchronize()
    p = _cabi.profile_read()
    _cabi.profile_enable(False)
    k = {n: v[0] / v[1] * 1e3 for n, v in p.items()}
    return wall, k


def main():
    dev = torch.device("cuda:0")
    spatial = "--tile-order" in sys.argv       # experiment: Gaussians re-ordered by the tile of their projected centre
    sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [300000, 8000]
    for N in sizes:
        sc = S.make_scene(N, 640, 480, seed=0)
        cam = sc.cam
        if spatial:
            p = sc.means3D
            u = p[:, 0] / p[:, 2] * cam.fx + cam.cx
            v = p[:, 1] / p[:, 2] * cam.fy + cam.cy
            key = (v.clamp(0, cam.H - 1) // 16).long() * 64 + (u.clamp(0, cam.W - 1) // 16).long()
            order = torch.argsort(key, stable=True)
            sc = sc._replace(means3D=sc.means3D[order].contiguous(), log_scales=sc.log_scales[order].contiguous(),
                             rot=sc.rot[order].contiguous(), opacity_logit=sc.opacity_logit[order].contiguous(),
                             features_dc=sc.features_dc[order].contiguous())
            print("Gaussians in tile order of their projected centres (experiment)")
        H, W = cam.H, cam.W
        gauss = GaussianParams(sc.means3D.to(dev), sc.log_scales.to(dev), sc.rot.to(dev), sc.opacity_logit.to(dev),
                               sc.features_dc.to(dev))
        fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
        bg = torch.zeros(3, device=dev)

        def view(T):
            return ViewCamera(1, torch.zeros(3, H, W), T, cam.projmatrix_raw, fovx, fovy, H, W, dev)
        with torch.no_grad():
            target = render(view(torch.eye(4)), gauss, Pipe, bg)["render"].clone()
        vp = view(SE3_exp(torch.tensor([0.01, -0.008, 0.006, 0.002, -0.003, 0.002])))
        vp.original_image = target
        vp.rgb_pixel_mask_mapping = (target.sum(0) > 0.01).view(1, H, W)
        trk = NativeTracker(vp, gauss, bg)
        trk.args.adam.sticky_converged = 0        # keep stepping after convergence: this is a timing run
        wall, k = profile(trk.step, 20, 400, 50)
        print(f"{N} Gaussians @ {W}x{H}, D = {trk.pairs()}: first order {wall:.1f} us / iteration "
              f"({1e6 / wall:.0f} its/s), kernels {sum(k.values()):.1f} us:",
              {n: round(v, 1) for n, v in sorted(k.items(), key=lambda kv: -kv[1])})
        trk.enable_second_order(stack_dim=16, sketch_dim=64, initial_lambda=1e-3)
        wall, k = profile(trk.step_second_order, 10, 200, 30)
        print(f"{N} Gaussians @ {W}x{H}: second order (stack 16, sketch 64) {wall:.1f} us / iteration "
              f"({1e6 / wall:.0f} its/s), kernels {sum(k.values()):.1f} us:",
              {n: round(v, 1) for n, v in sorted(k.items(), key=lambda kv: -kv[1])})
        assert trk.check_capacity()
        mapping_profile(sc, dev)


def mapping_profile(sc, dev):
    """One view in flight (so that the events bracket one kernel at a time), window 8 + 2 old keyframes,
    all learning rates 0 and densification off: the stationary workload of bench.py's `mapping` leg."""
    from monogs_amd.bench_legs import FROZEN_POSE_LR, _model_from_scene
    from monogs_amd.mapping_native import NativeMapper
    from monogs_amd.parallel import view_pose
    cam = sc.cam
    H, W, N = cam.H, cam.W, sc.means3D.shape[0]
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    bg = torch.zeros(3, device=dev)
    gm = _model_from_scene(sc, dev)
    views = [ViewCamera(i, sc.gt_image, view_pose(i), cam.projmatrix_raw, fovx, fovy, H, W, dev) for i in range(10)]
    mp = NativeMapper(gm, bg, config={"Training": {"gaussian_update_every": 10 ** 9, "gaussian_update_offset": 10 ** 9 - 1,
                                                   "gaussian_reset": 10 ** 9, "lr": FROZEN_POSE_LR}}, concurrent_views=1)
    for i, v in enumerate(views):
        mp.add_keyframe(i, v)
    mp.set_window(list(range(7, -1, -1)))
    mp.map(iters=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mp.map(iters=20)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 20 * 1e6
    _cabi.profile_enable(True)
    mp.map(iters=5)
    torch.cuda.synchronize()
    p = _cabi.profile_read()
    _cabi.profile_enable(False)
    k = {n: v[0] / v[1] * 1e3 for n, v in p.items()}
    calls = {n: v[1] / 5 for n, v in p.items()}
    per_iter = sum(k[n] * calls[n] for n in k)
    print(f"{N} Gaussians @ {W}x{H}: mapping, one view in flight, {wall:.0f} us / iteration of 10 views "
          f"({wall / 10:.1f} us / view), kernels {per_iter:.0f} us / iteration; us per launch (launches per iteration):",
          {n: (round(v, 1), calls[n]) for n, v in sorted(k.items(), key=lambda kv: -kv[1] * calls[kv[0]])})
    assert mp.check_capacity() and len(gm) == N


if __name__ == "__main__":
    main()
