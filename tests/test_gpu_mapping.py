"""Native mapping iteration (row a13, utils/slam_backend.py:171-332) on the GPU.

Parity chain: window-summed gradients of the native path (one C-ABI call per view, gradients
chained through the activations and accumulated on the device) against the torch ORACLE - the sum
of per-view autograd through the same activations and the reference's mapping objective
(utils/slam_utils.py:224-253) plus the isotropic regulariser (slam_backend.py:244-246) - and the
native loop against the reference-shaped Python loop on the same kernels.
"""
import math

import pytest
import torch

from conftest import oracle_settings, rel_err

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


def _window_fixture(N=4000, W=160, H=120, n_views=3, seed=11, dev=None, rgbd=False, intrinsics=None, sh_degree=0,
                    active_degree=None):
    from monogs_amd import synthetic as S
    from monogs_amd.gaussian_model import GaussianModel
    from monogs_amd.parallel import view_pose
    from monogs_amd.slam_loops import ViewCamera
    sc = S.make_scene(N, W, H, seed=seed, intrinsics=intrinsics)
    cam = sc.cam
    fovx, fovy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
    g = torch.Generator().manual_seed(seed + 1)
    gm = GaussianModel(sh_degree, device=dev)
    import torch.nn as nn
    gm._xyz = nn.Parameter(sc.means3D.to(dev).contiguous())
    gm._features_dc = nn.Parameter(sc.features_dc.to(dev).contiguous())
    K = (sh_degree + 1) ** 2
    # stored SH bands above the constant one (K > 1): cat(features_dc, features_rest) in mgs_map_activate, the
    # general-degree instantiation of the preprocess backward and grad_features_rest
    gm._features_rest = nn.Parameter((0.3 * torch.randn(N, K - 1, 3, generator=g)).to(dev).contiguous())
    gm.active_sh_degree = sh_degree if active_degree is None else active_degree
    gm._scaling = nn.Parameter(sc.log_scales.to(dev).contiguous())
    # un-normalised quaternions: the chain through normalize() is part of what is tested
    gm._rotation = nn.Parameter((sc.rot * (0.5 + torch.rand(N, 1, generator=g))).to(dev).contiguous())
    gm._opacity = nn.Parameter(sc.opacity_logit.to(dev).contiguous())
    gm.max_radii2D = torch.zeros(N, device=dev)
    gm.unique_kfIDs = torch.zeros(N, dtype=torch.int32, device=dev)
    gm.n_obs = torch.zeros(N, dtype=torch.int32, device=dev)
    gm.init_lr(6.0)
    gm.training_setup()
    views = []
    for i in range(n_views):
        img = torch.rand(3, H, W, generator=g)
        img[:, : H // 8] = 0.0            # a masked border: rgb_pixel_mask_mapping is exercised
        depth = 0.5 + 5.5 * torch.rand(1, H, W, generator=g) if rgbd else None
        v = ViewCamera(i, img, view_pose(i), cam.projmatrix_raw, fovx, fovy, H, W, dev, gt_depth=depth,
                       intrinsics=(cam.fx, cam.fy, cam.cx, cam.cy))
        with torch.no_grad():
            v.exposure_a.fill_(1.0 + 0.07 * i)
            v.exposure_b.fill_(0.02 * i - 0.01)
        views.append(v)
    return sc, gm, views


def _oracle_window_gradients(sc, gm, views, rgbd=False, alpha=0.95):
    """Sum over the views of the reference's mapping objective through the torch oracle, with
    the raw parameters as leaves; returns their gradients, the statistics and the per-view pose /
    exposure gradients."""
    from monogs_amd import synthetic as S
    from oracle import torch_raster as O
    leaves = {k: getattr(gm, k).detach().cpu().clone().requires_grad_() for k in
              ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")}
    deg = int(gm.active_sh_degree)
    N = leaves["_xyz"].shape[0]
    total = 0.0
    per_view, stats = [], dict(gradnorm=torch.zeros(N), denom=torch.zeros(N), radii=torch.zeros(N, dtype=torch.int32),
                               vis=[])
    for v in views:
        cam = S.make_camera(v.image_width, v.image_height, v.T.detach().cpu(), intrinsics=(v.fx, v.fy, v.cx, v.cy))
        # campos: the reference's camera_center IS world_view_transform (camera_utils.py:106-108), i.e. the
        # extension reads the first three floats of the view matrix; the native mapper passes the same pointer
        # and oracle_settings defaults to it
        st = oracle_settings(cam, torch.zeros(3), deg=deg)
        theta = torch.zeros(3, requires_grad=True)
        rho = torch.zeros(3, requires_grad=True)
        a = v.exposure_a.detach().cpu().clone().requires_grad_()
        b = v.exposure_b.detach().cpu().clone().requires_grad_()
        m2d = torch.zeros(N, 3, requires_grad=True)
        img, radii, dep, opa, nt, _ = O.rasterize(
            leaves["_xyz"], m2d, torch.cat((leaves["_features_dc"], leaves["_features_rest"]), dim=1), None,
            torch.sigmoid(leaves["_opacity"]),
            torch.exp(leaves["_scaling"]), torch.nn.functional.normalize(leaves["_rotation"]), None, st, theta, rho)
        gt = v.original_image.cpu()
        mask = v.rgb_pixel_mask_mapping.cpu().float()
        image_ab = (torch.abs(a) + v.exposure_eps) * img + b
        l_rgb = torch.abs(image_ab * mask - gt * mask).mean()
        if rgbd:
            gd = v.gt_depth.cpu()
            dm = (gd > 0.01).float()
            loss = alpha * l_rgb + (1 - alpha) * torch.abs(dep * dm - gd * dm).mean()
        else:
            loss = l_rgb
        total = total + loss
        per_view.append(dict(theta=theta, rho=rho, a=a, b=b, m2d=m2d, radii=radii, nt=nt, loss=loss))
    s = torch.exp(leaves["_scaling"])
    total = total + 10 * torch.abs(s - s.mean(dim=1, keepdim=True)).mean()
    total.backward()
    for pv in per_view:
        vis = pv["radii"] > 0
        g2 = torch.linalg.norm(pv["m2d"].grad[:, :2], dim=-1)
        stats["gradnorm"] += torch.where(vis, g2, torch.zeros_like(g2))
        stats["denom"] += vis.float()
        stats["radii"] = torch.maximum(stats["radii"], pv["radii"])
        stats["vis"].append(pv["nt"] > 0)
    return {k: t.grad for k, t in leaves.items()}, stats, per_view, float(total)


@pytest.mark.parametrize("rgbd", [False, True])
def test_native_window_gradients_match_the_oracle(built, rgbd):
    _check_window_against_oracle(rgbd)


@pytest.mark.parametrize("sh_degree,active", [(1, 1), (3, 3), (2, 1), (1, 0)])
def test_native_window_gradients_match_the_oracle_with_sh_bands(built, sh_degree, active):
    """Mapping mode with K > 1 stored SH coefficients (sh_degree 1: K = 4; 3: K = 16): mgs_map_activate's
    cat(features_dc, features_rest), k_preprocess_bwd<MAP, general degree> (the view-direction term of
    dL/dxyz included) and mgs_map_accum_args.grad_features_rest, accumulated over a 3-view window, against
    the oracle.  (2, 1) / (1, 0): stored degree above the active one (oneupSHdegree has not caught up,
    gaussian_model.py:104-106) - the higher bands must get exactly zero."""
    _check_window_against_oracle(False, sh_degree=sh_degree, active_degree=active)


def test_native_window_gradients_match_the_oracle_at_the_replica_shape(built):
    """BASELINE config 5's view shape - Replica office0 calibration, 1200x680, RGB-D objective
    (configs/rgbd/replica/base_config.yaml:12,27-28; utils/slam_utils.py:243-253) - through the native
    mapping iteration against the torch oracle, on a 3-view window at a reduced map size (the oracle
    walks 3225 tiles per view on the CPU)."""
    from monogs_amd import synthetic as S
    _check_window_against_oracle(True, N=12000, W=1200, H=680, intrinsics=S.REPLICA_INTRINSICS, seed=17)


def _check_window_against_oracle(rgbd, **fixture_kw):
    from monogs_amd.mapping_native import NativeMapper
    from monogs_amd.pose import SE3_exp
    dev = _dev()
    sc, gm, views = _window_fixture(dev=dev, rgbd=rgbd, **fixture_kw)
    want, stats, per_view, total = _oracle_window_gradients(sc, gm, views, rgbd=rgbd)
    T0 = [v.T.clone() for v in views]
    ab0 = [(float(v.exposure_a), float(v.exposure_b)) for v in views]
    mp = NativeMapper(gm, torch.zeros(3, device=dev), config={"Training": {"monocular": not rgbd}})
    for i, v in enumerate(views):
        mp.add_keyframe(i, v)
    mp.set_window([0, 1, 2])
    mp._ensure_model_buffers()
    mp._activate()
    mp.loss_accum.zero_()
    for n, kf in enumerate([0, 1, 2]):
        mp._run_view(kf, n, accumulate=n > 0, add_reg=n == 0, in_window=True)
    torch.cuda.synchronize()
    assert mp.check_capacity()
    names = [("xyz", "_xyz"), ("f_dc", "_features_dc"), ("opacity", "_opacity"), ("scaling", "_scaling"),
             ("rotation", "_rotation")]
    if gm._features_rest.shape[1] > 0:
        names.append(("f_rest", "_features_rest"))
    for name, attr in names:
        got = mp._section(name).view_as(getattr(gm, attr)).cpu()
        if attr == "_features_rest":
            act = (int(gm.active_sh_degree) + 1) ** 2 - 1          # bands above the active degree: exactly zero
            assert float(got[:, act:].abs().max()) == 0.0 if act < got.shape[1] else True
            assert float(want[attr][:, act:].abs().max()) == 0.0 if act < got.shape[1] else True
            if act == 0:
                continue
        err = rel_err(got, want[attr])
        assert err < 2e-3, f"{attr}: window-summed gradient rel err {err}"
    assert rel_err(mp._section("gradnorm").cpu(), stats["gradnorm"]) < 2e-3
    assert torch.equal(mp._section("denom").cpu(), stats["denom"])
    assert torch.equal(mp.radii_max.cpu(), stats["radii"])
    for i in range(3):
        agree = (mp.occ_aware_visibility[i].cpu().bool() == stats["vis"][i]).float().mean()
        assert agree > 0.999          # n_touched counts T > 0.5 contributions: a threshold, fp32 on both sides
    # objective value: the regulariser is not part of the native loss scalar
    s = torch.exp(gm._scaling.detach().cpu())
    reg = float(10 * torch.abs(s - s.mean(dim=1, keepdim=True)).mean())
    assert abs(float(mp.loss_accum) - (total - reg)) < 1e-4 * abs(total)
    # per-view optimiser: Adam on (rot, trans) at half the tracking rates + exposure, update_pose for
    # the first pose_window views; keyframe 0 is never moved (slam_backend.py:452-489, :328-332)
    assert torch.equal(views[0].T, T0[0]) and (float(views[0].exposure_a), float(views[0].exposure_b)) == ab0[0]
    for i in (1, 2):
        pv = per_view[i]
        params = [torch.zeros(3, requires_grad=True), torch.zeros(3, requires_grad=True),
                  torch.tensor([ab0[i][0]], requires_grad=True), torch.tensor([ab0[i][1]], requires_grad=True)]
        opt = torch.optim.Adam([{"params": [params[0]], "lr": 0.0015}, {"params": [params[1]], "lr": 0.0005},
                                {"params": [params[2]], "lr": 0.02}, {"params": [params[3]], "lr": 0.02}])
        for p, gr in zip(params, (pv["theta"].grad, pv["rho"].grad, pv["a"].grad, pv["b"].grad)):
            p.grad = gr.reshape(p.shape).clone()
        opt.step()
        T_want = SE3_exp(torch.cat([params[1].detach(), params[0].detach()])) @ T0[i].cpu()
        assert torch.allclose(views[i].T.cpu(), T_want, atol=2e-6), i
        assert abs(float(views[i].exposure_a) - float(params[2])) < 1e-6
        assert abs(float(views[i].exposure_b) - float(params[3])) < 1e-6
        assert float(views[i].cam_rot_delta.abs().max()) == 0.0


def test_native_map_loop_matches_the_python_loop(built):
    """NativeMapper.map against slam_loops.mapping_step (reference-shaped body, autograd, the same
    HIP rasteriser, FusedGaussianAdam) from the same state: 3 iterations over a 3-view window."""
    from monogs_amd.mapping_native import NativeMapper
    from monogs_amd.slam_loops import Pipe, mapping_step
    dev = _dev()
    iters = 3
    _, gm_a, views_a = _window_fixture(dev=dev, seed=12)
    _, gm_b, views_b = _window_fixture(dev=dev, seed=12)
    # python loop
    groups = []
    for i, v in enumerate(views_a):
        if i == 0:
            continue
        groups += [{"params": [v.cam_rot_delta], "lr": 0.0015}, {"params": [v.cam_trans_delta], "lr": 0.0005},
                   {"params": [v.exposure_a], "lr": 0.02}, {"params": [v.exposure_b], "lr": 0.02}]
    kopt = torch.optim.Adam(groups)
    bg = torch.zeros(3, device=dev)
    cfg = {"Training": {"monocular": True, "rgb_boundary_threshold": 0.01}}
    losses_a = []
    for it in range(iters):
        sa = torch.exp(gm_a._scaling.detach())
        reg = float(10 * torch.abs(sa - sa.mean(dim=1, keepdim=True)).mean())
        out = mapping_step(views_a, gm_a, gm_a.optimizer, kopt, bg, Pipe, cfg, pose_window=3)
        gm_a.xyz_gradient_accum += out[1][:, None]
        gm_a.denom += out[2][:, None]
        gm_a.max_radii2D = torch.maximum(gm_a.max_radii2D, out[3].float())
        gm_a.update_learning_rate(it + 1)
        losses_a.append(float(out[0]) - reg)      # the native loss scalar carries no regulariser
    # native loop
    mp = NativeMapper(gm_b, bg)
    for i, v in enumerate(views_b):
        mp.add_keyframe(i, v)
    mp.set_window([0, 1, 2])
    losses_b = []
    for it in range(iters):
        mp.map(iters=1)
        losses_b.append(float(mp.last_loss))
    torch.cuda.synchronize()
    assert mp.check_capacity()
    for la, lb in zip(losses_a, losses_b):
        assert abs(la - lb) < 2e-3 * abs(la), (losses_a, losses_b)
    assert losses_b[-1] < losses_b[0]
    # Adam normalises the gradient, so a Gaussian whose gradient is ~0 may step either way: compare
    # the bulk, not every element
    for attr, lr in (("_xyz", 0.0016 * 6), ("_features_dc", 0.0025), ("_opacity", 0.05), ("_scaling", 0.006), ("_rotation", 0.001)):
        a, b = getattr(gm_a, attr).detach(), getattr(gm_b, attr).detach()
        close = ((a - b).abs() <= 0.05 * lr * iters + 1e-6).float().mean()
        assert close > 0.995, (attr, float(close))
    assert rel_err(gm_b.xyz_gradient_accum, gm_a.xyz_gradient_accum) < 5e-3
    assert torch.equal(gm_b.denom, gm_a.denom) or rel_err(gm_b.denom, gm_a.denom) < 1e-3
    for va, vb in zip(views_a, views_b):
        assert torch.allclose(va.T, vb.T, atol=5e-5)
        assert abs(float(va.exposure_a) - float(vb.exposure_a)) < 2e-4


def test_extend_from_pcd_appends_rows_and_optimizer_state(built):
    """GaussianModel.extend_from_pcd (gaussian_model.py:210-245, :525-593) as one launch."""
    dev = _dev()
    _, gm, views = _window_fixture(N=1000, dev=dev, seed=3)
    N = len(gm)
    g = torch.Generator(device=dev).manual_seed(0)
    for grp in gm.optimizer.param_groups:            # one optimiser step so that moments exist
        p = grp["params"][0]
        p.grad = torch.randn(p.shape, device=dev, generator=g) * 1e-3
    gm.optimizer.step()
    before = {n: getattr(gm, a).detach().clone() for n, a in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("opacity", "_opacity"),
                                                               ("scaling", "_scaling"), ("rotation", "_rotation"))}
    m_before = {grp["name"]: gm.optimizer.state[grp["params"][0]]["exp_avg"].clone() for grp in gm.optimizer.param_groups
                if grp["params"][0] in gm.optimizer.state}
    gm.xyz_gradient_accum += 1.0
    P = 137
    xyz = torch.randn(P, 3, device=dev, generator=g)
    feats = torch.randn(P, 3, 1, device=dev, generator=g)
    scales = torch.randn(P, 3, device=dev, generator=g)
    rots = torch.randn(P, 4, device=dev, generator=g)
    opac = torch.randn(P, 1, device=dev, generator=g)
    gm.extend_from_pcd(xyz, feats, scales, rots, opac, kf_id=7)
    assert len(gm) == N + P
    assert torch.equal(gm._xyz[:N], before["xyz"]) and torch.equal(gm._xyz[N:], xyz)
    assert torch.equal(gm._features_dc[:N], before["f_dc"]) and torch.equal(gm._features_dc[N:, 0], feats[:, :, 0])
    assert torch.equal(gm._scaling[N:], scales) and torch.equal(gm._rotation[N:], rots) and torch.equal(gm._opacity[N:], opac)
    assert gm._features_rest.shape == (N + P, 0, 3)
    for grp in gm.optimizer.param_groups:
        p = grp["params"][0]
        assert p is getattr(gm, {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                                 "scaling": "_scaling", "rotation": "_rotation"}[grp["name"]])
        if grp["name"] in m_before and p.numel():
            st = gm.optimizer.state[p]
            assert torch.equal(st["exp_avg"][:N], m_before[grp["name"]]) and float(st["exp_avg"][N:].abs().max()) == 0.0
            assert st["exp_avg_sq"].shape == p.shape and st["step"] == 1
    assert torch.equal(gm.unique_kfIDs[N:], torch.full((P,), 7, dtype=torch.int32, device=dev))
    assert int(gm.unique_kfIDs[:N].abs().sum()) == 0 and int(gm.n_obs.sum()) == 0
    assert gm.xyz_gradient_accum.shape == (N + P, 1) and float(gm.xyz_gradient_accum.abs().max()) == 0.0
    assert gm.max_radii2D.shape == (N + P,) and gm.denom.shape == (N + P, 1)
    # the extended model still steps
    for grp in gm.optimizer.param_groups:
        p = grp["params"][0]
        p.grad = torch.ones_like(p)
    gm.optimizer.step()
    assert float((gm._xyz[N:] - xyz).abs().min()) > 0


def test_reset_opacity_and_nonvisible(built):
    """gaussian_model.py:364-377 + replace_tensor_to_optimizer :470-483."""
    dev = _dev()
    _, gm, _ = _window_fixture(N=500, dev=dev, seed=4)
    p = gm._opacity
    p.grad = torch.ones_like(p)
    gm.optimizer.step()
    keep = gm._opacity.detach().clone()
    vis1 = torch.zeros(500, dtype=torch.bool, device=dev); vis1[:100] = True
    vis2 = torch.zeros(500, dtype=torch.bool, device=dev); vis2[50:200] = True
    gm.reset_opacity_nonvisible([vis1, vis2], keep_visible_logits=True)
    want = math.log(0.4 / 0.6)
    assert torch.equal(gm._opacity[:200], keep[:200])
    gm._opacity.data.copy_(keep)
    gm.reset_opacity_nonvisible([vis1, vis2])         # the reference to the letter (gaussian_model.py:375)
    assert torch.allclose(gm._opacity[:200], torch.sigmoid(keep[:200]), rtol=1e-6, atol=1e-7)
    assert torch.allclose(gm._opacity[200:], torch.full_like(gm._opacity[200:], want), atol=1e-6)
    st = gm.optimizer.state[gm._opacity]
    assert float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0
    gm.reset_opacity()
    assert torch.allclose(gm._opacity, torch.full_like(gm._opacity, math.log(0.01 / 0.99)), atol=1e-6)


def test_prune_pass_and_extra_views(built):
    """map(prune=True) (:259-290): n_obs from the window's occ-aware visibility, Gaussians of the
    recent keyframes seen by <= 3 views removed; map() with old keyframes outside the window renders
    two of them per iteration (:215-242) into the statistics but not into occ_aware_visibility."""
    from monogs_amd.mapping_native import NativeMapper
    dev = _dev()
    _, gm, views = _window_fixture(N=3000, n_views=7, dev=dev, seed=5)
    gm.unique_kfIDs = (torch.arange(3000, device=dev) % 7).to(torch.int32)
    mp = NativeMapper(gm, torch.zeros(3, device=dev), config={"Training": {"window_size": 5}})
    for i, v in enumerate(views):
        mp.add_keyframe(i, v)
    window = [6, 5, 4, 3, 2]
    mp.set_window(window)
    mp.map(iters=2)
    torch.cuda.synchronize()
    assert set(mp.occ_aware_visibility) == set(window)
    # 5 window views + the 2 old keyframes: a Gaussian seen everywhere was counted 7 times per iteration
    assert float(gm.denom.max()) == 14.0
    N = len(gm)
    ids = gm.unique_kfIDs.clone()
    mp.map(prune=True)
    torch.cuda.synchronize()
    # the decision is a pure function of the visibilities the pass itself left behind (filtered by the
    # survivors) and of the ids: reconstruct it from the pre-prune ids and the recorded n_obs
    n_obs = gm.n_obs            # rebuilt with the survivors' rows
    assert mp.initialized and len(gm) < N
    assert int((n_obs <= 3).sum()) == 0          # not yet initialised: every Gaussian with n_obs <= 3 went
    for k in window:
        assert mp.occ_aware_visibility[k].shape[0] == len(gm)
    assert len(gm.unique_kfIDs) == len(gm) and gm.optimizer.state[gm._xyz]["exp_avg"].shape[0] == len(gm)
    assert gm.xyz_gradient_accum.shape[0] == len(gm) and gm.max_radii2D.shape[0] == len(gm)
    # initialised: only Gaussians inserted by the three newest keyframes of the window may be pruned
    with torch.no_grad():                         # hide a block of Gaussians from every view
        gm._opacity[:600] = -20.0
    ids2 = gm.unique_kfIDs.clone()
    mp.map(prune=True)
    torch.cuda.synchronize()
    hidden_old = int((ids2[:600] < 4).sum())
    assert int((gm.unique_kfIDs < 4).sum()) == int((ids2 < 4).sum())          # ids < window-sorted[2] = 4 untouched
    assert len(gm) <= len(ids2) - (600 - hidden_old)                          # the hidden recent ones are gone


def test_initialize_map_runs_the_reference_schedule(built):
    """initialize_map (:91-146): single view, no exposure, densify every init_gaussian_update, opacity reset."""
    from monogs_amd.mapping_native import NativeMapper
    from monogs_amd.gaussian_renderer import render
    from monogs_amd.slam_loops import Pipe
    dev = _dev()
    _, gm, views = _window_fixture(N=3000, n_views=1, dev=dev, seed=6)
    # target: the scene itself rendered with different colours -> the colours must be learned
    with torch.no_grad():
        target = render(views[0], gm, Pipe, torch.zeros(3, device=dev))["render"].clone()
        gm._features_dc.mul_(0.3)
    views[0].original_image = target
    views[0].rgb_pixel_mask_mapping = (target.sum(0) > 0.01).view(1, *target.shape[1:])
    mp = NativeMapper(gm, torch.zeros(3, device=dev),
                      config={"Training": {"init_gaussian_update": 20, "init_gaussian_reset": 45}})
    mp.add_keyframe(0, views[0])
    mp.initialize_map(0, iters=1)
    l0 = float(mp.last_loss)
    n0 = len(gm)
    mp.initialize_map(0, iters=41)              # densify_and_prune at its iterations 0, 20, 40
    torch.cuda.synchronize()
    assert mp.check_capacity()
    assert float(mp.last_loss) < 0.7 * l0
    assert len(gm) != n0
    assert float(gm.get_opacity.max()) > 0.5
    mp.initialize_map(0, iters=3)               # iteration_count 43..45: reset_opacity at 45
    torch.cuda.synchronize()
    assert float(gm.get_opacity.max()) < 0.011
    assert mp.occ_aware_visibility[0].shape[0] == len(gm)


def replica_window_fixture(dev, N=60000, n_kf=10, seed=41):
    """BASELINE config 5's shape on one GPU: Replica office0 calibration, 1200x680, RGB-D, `n_kf`
    keyframes (a window of 8 + the old keyframes the 2 random extra views are drawn from); random
    target images / depth maps with a masked border, as in the small fixtures."""
    from monogs_amd import synthetic as S
    sc, gm, views = _window_fixture(N=N, W=1200, H=680, n_views=n_kf, seed=seed, dev=dev, rgbd=True,
                                    intrinsics=S.REPLICA_INTRINSICS)
    gm.unique_kfIDs = (torch.arange(N, device=dev) % n_kf).to(torch.int32)
    return sc, gm, views


def test_native_mapper_at_the_config5_shape(built):
    """NativeMapper over an 8-view window + 2 random old keyframes, RGB-D, 1200x680 (Replica office0
    calibration), 60 000 Gaussians - the per-iteration work of BASELINE config 5 on one GPU
    (utils/slam_backend.py:183-247): finite, complete renders, the summed objective decreases, every
    one of the 10 views enters the statistics, the prune pass builds n_obs from 8 visibilities."""
    from monogs_amd.mapping_native import NativeMapper
    dev = _dev()
    N = 60000
    _, gm, views = replica_window_fixture(dev, N=N)
    cfg = {"Training": {"monocular": False, "window_size": 8, "gaussian_update_every": 1000}}
    mp = NativeMapper(gm, torch.zeros(3, device=dev), config=cfg)
    for i, v in enumerate(views):
        mp.add_keyframe(i, v)
    window = [9, 8, 7, 6, 5, 4, 3, 2]
    mp.set_window(window)
    T0 = {v.uid: v.T.clone() for v in views}
    mp.map(iters=1)
    l0 = float(mp.last_loss)
    assert float(gm.denom.max()) == 10.0                   # 8 window views + 2 extras in the statistics
    mp.map(iters=14)
    torch.cuda.synchronize()
    l1 = float(mp.last_loss)
    assert mp.check_capacity()
    assert math.isfinite(l0) and math.isfinite(l1) and l1 < l0, (l0, l1)
    for attr in ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"):
        assert bool(torch.isfinite(getattr(gm, attr)).all()), attr
    assert float(gm.denom.max()) == 150.0 and float(gm.xyz_gradient_accum.max()) > 0
    assert set(mp.occ_aware_visibility) == set(window)
    assert mp.color.shape == (3, 680, 1200) and bool(torch.isfinite(mp.color).all())
    # window positions 0..2 (pose_window) are pose-optimised, old keyframes (extras) never move
    for uid in (9, 8, 7):
        assert not torch.equal(views[uid].T, T0[uid])
    for uid in (0, 1, 6, 2):
        assert torch.equal(views[uid].T, T0[uid]), uid
    mp.map(prune=True)
    torch.cuda.synchronize()
    assert gm.n_obs.shape[0] == len(gm) == N and int(gm.n_obs.max()) == 8     # RGB-D: counted, not pruned (:286)
