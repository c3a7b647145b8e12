"""End-to-end integration of the hot-path pieces on a synthetic sequence (GPU): keyframe
insertion from depth, mapping iterations (fused loss + fused Adam), densify / prune, native
first- and second-order tracking, ATE / PSNR evaluation.  Not a SLAM system (keyframe policy,
queues and windows are out of scope): the smallest loop that makes every component
consume another's output, the way utils/slam_frontend.py / slam_backend.py chain them."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


class _Frame:
    pass


def test_mini_slam_on_a_synthetic_sequence(built):
    from monogs_amd import synthetic as S
    from monogs_amd import eval_metrics as E
    from monogs_amd import map_update as MU
    from monogs_amd.gaussian_renderer import render
    from monogs_amd.keyframe_init import create_pcd_from_image_and_depth
    from monogs_amd.pose import SE3_exp
    from monogs_amd.slam_loops import DEFAULT_CONFIG, GaussianParams, Pipe, ViewCamera, mapping_step
    from monogs_amd.tracking_native import NativeTracker

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    W, H = 160, 120
    sc = S.make_scene(6000, W, H, seed=33)
    cam = sc.cam
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    bg = torch.zeros(3, device=dev)
    world = GaussianParams(sc.means3D.to(dev), sc.log_scales.to(dev), sc.rot.to(dev),
                           sc.opacity_logit.to(dev), sc.features_dc.to(dev))

    def view(uid, T):
        return ViewCamera(uid, torch.zeros(3, H, W), T, cam.projmatrix_raw, fovx, fovy, H, W, dev)

    # ground-truth sequence rendered from the "world" map
    step = torch.tensor([0.04, -0.02, 0.025, 0.012, -0.009, 0.006])
    poses_gt = [SE3_exp(k * step) for k in range(6)]
    frames_gt = []
    with torch.no_grad():
        for k, T in enumerate(poses_gt):
            pkg = render(view(k, T), world, Pipe, bg)
            frames_gt.append((pkg["render"].clone(), pkg["depth"][0].clone()))

    # ---- map initialisation from keyframe 0 (gaussian_model.py:108-205) ----
    kf0 = view(0, poses_gt[0])
    kf0.fx, kf0.fy = W / (2 * cam.tanfovx), H / (2 * cam.tanfovy)
    kf0.cx, kf0.cy = (W - 1) * 0.5, (H - 1) * 0.5
    img0, dep0 = frames_gt[0]
    xyz, feats, scales, rots, opac = create_pcd_from_image_and_depth(
        kf0, img0, torch.where(dep0 > 0.05, dep0, torch.zeros_like(dep0)), downsample_factor=2.0,
        generator=torch.Generator(device=dev).manual_seed(1))
    model = GaussianParams(xyz, scales.repeat(1, 3), rots, opac, feats.transpose(1, 2).contiguous())
    model.percent_dense = 0.01
    groups = [{"params": [model._xyz], "lr": 1.6e-4, "name": "xyz"},
              {"params": [model._features_dc], "lr": 2.5e-3, "name": "f_dc"},
              {"params": [model._features_rest], "lr": 1.25e-4, "name": "f_rest"},
              {"params": [model._opacity], "lr": 0.05, "name": "opacity"},
              {"params": [model._scaling], "lr": 1e-3, "name": "scaling"},
              {"params": [model._rotation], "lr": 1e-3, "name": "rotation"}]
    model.optimizer = MU.FusedGaussianAdam(groups, lr=0.0, eps=1e-15)
    n0 = int(model._xyz.shape[0])
    model.xyz_gradient_accum = torch.zeros(n0, 1, device=dev)
    model.denom = torch.zeros(n0, 1, device=dev)
    model.max_radii2D = torch.zeros(n0, device=dev)
    model.unique_kfIDs = torch.zeros(n0, dtype=torch.int32, device=dev)
    model.n_obs = torch.zeros(n0, dtype=torch.int32, device=dev)

    def keyframe(uid, T, k):
        v = view(uid, T)
        v.original_image = frames_gt[k][0]
        v.rgb_pixel_mask_mapping = (frames_gt[k][0].sum(0) > 0.01).view(1, H, W)
        return v

    def psnr_at(T, k):
        with torch.no_grad():
            img = render(view(99, T), model, Pipe, bg)["render"].clamp(0, 1)
        return E.psnr(img.unsqueeze(0), frames_gt[k][0].unsqueeze(0)).item()

    def map_iterations(window, n):
        loss = None
        for _ in range(n):
            loss, grad_norm, denom, radii = mapping_step(window, model, model.optimizer, None, bg,
                                                         config=DEFAULT_CONFIG, fused_loss=True)
            model.xyz_gradient_accum += grad_norm[:, None]
            model.denom += denom[:, None]
            model.max_radii2D = torch.maximum(model.max_radii2D, radii.float())
        return loss.item()

    window = [keyframe(0, poses_gt[0], 0)]
    p_before = psnr_at(poses_gt[0], 0)
    l_first = map_iterations(window, 5)
    l_last = map_iterations(window, 60)
    p_after = psnr_at(poses_gt[0], 0)
    assert l_last < l_first and p_after > p_before + 1.0 and p_after > 18.0

    # ---- densify / prune on the accumulated statistics (gaussian_model.py:674-691) ----
    n_before = int(model._xyz.shape[0])
    MU.densify_and_prune(model, 2e-4, 0.05, 6.0, 20)
    n_after = int(model._xyz.shape[0])
    assert n_after != n_before and model.optimizer.param_groups[0]["params"][0] is model._xyz

    # ---- mapping over the whole keyframe window (pose-supervised: keyframe policy, joint pose
    # refinement and per-keyframe insertion are out of scope), one more densification ----
    window = [keyframe(k, poses_gt[k].clone(), k) for k in range(6)]
    map_iterations(window, 40)
    MU.densify_and_prune(model, 2e-4, 0.05, 6.0, 20)
    map_iterations(window, 40)
    n_final = int(model._xyz.shape[0])

    # ---- track every frame natively from the previous frame's pose ----
    est = [poses_gt[0].clone()]
    frames_eval, track_errors = [], []
    for k in range(1, 6):
        vp = keyframe(k, poses_gt[k - 1].clone(), k)  # constant-position motion model
        trk = NativeTracker(vp, model, bg)
        trk.run(max_iters=80, check_every=10)
        trk.enable_second_order(stack_dim=4, sketch_dim=16, initial_lambda=1e-3, seed=k)
        for _ in range(5):
            trk.step_second_order()
        assert trk.check_capacity()
        e_init = (torch.linalg.inv(poses_gt[k - 1])[:3, 3] - torch.linalg.inv(poses_gt[k])[:3, 3]).norm().item()
        e_trk = (torch.linalg.inv(vp.T.cpu())[:3, 3] - torch.linalg.inv(poses_gt[k])[:3, 3]).norm().item()
        print(f"frame {k}: camera-centre error {e_init * 1e3:.1f} mm (motion model) -> {e_trk * 1e3:.1f} mm (tracked), "
              f"loss {trk.loss.item():.3f}, lambda {trk.lm_state[0].item():.2e}")
        track_errors.append((e_init, e_trk))
        est.append(vp.T.detach().clone())
    for k in range(6):
        f = _Frame()
        f.uid, f.T, f.T_gt = k, est[k].cpu(), poses_gt[k]
        frames_eval.append(f)
    ate = E.eval_ate(frames_eval, list(range(6)), monocular=False)
    path = sum((poses_gt[k + 1][:3, 3] - poses_gt[k][:3, 3]).norm().item() for k in range(5))
    print(f"mini-SLAM: ATE RMSE {ate:.5f} m over a {path:.3f} m path; PSNR kf0 {p_before:.1f} -> {p_after:.1f} dB; "
          f"Gaussians {n0} -> {n_after} -> {n_final}")
    ratio = sum(t / i for i, t in track_errors) / len(track_errors)
    assert ratio < 0.6, track_errors            # tracking removes most of the motion-model error
    assert ate < 0.5 * path / 5                 # well below one frame-to-frame step
    assert psnr_at(poses_gt[3], 3) > 12.0        # the mapped views are reproduced (coarse 160x120 toy map)
