"""The HIP path against the committed oracle vectors of tests/golden/ (generated in the build
container by tests/golden/make_mid_golden.py): full-path evidence at sizes that cross the
segment / sort-class boundaries which does NOT run through csrc/raster_math.h on both sides
(VERDICT r1, weak item 2), an SH-3 case, the sketch KAT, knn on a keyframe-sized point set, and
both backward treatments of the EWA clamp.  Tolerances: forward image L1 <= 1e-4, gradients
<= 1e-3 relative (norm-wise), pose gradient <= 2e-3 (north star)."""
import os

import numpy as np
import pytest
import torch

import scenes
from conftest import gpu_settings, rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FWD_L1, BWD_REL = 1e-4, 1e-3


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _hip(sc, shs=None, deg=0, campos=None):
    from monogs_amd import synthetic as S
    from monogs_amd.rasterizer import GaussianRasterizer
    dev = _dev()
    m, s, r, o, sh0 = S.activated(sc)
    sh = sh0 if shs is None else shs
    L = [t.clone().to(dev).requires_grad_() for t in (m, s, r, o, sh)]
    theta = torch.zeros(3, device=dev, requires_grad=True)
    rho = torch.zeros(3, device=dev, requires_grad=True)
    m2d = torch.zeros(m.shape[0], 3, device=dev, requires_grad=True)
    st = gpu_settings(sc.cam, sc.bg, dev, deg=deg, campos=campos)
    img, radii, dep, opa, nt = GaussianRasterizer(st)(means3D=L[0], means2D=m2d, shs=L[4], opacities=L[3],
                                                      scales=L[1], rotations=L[2], theta=theta, rho=rho)
    S.synthetic_loss(img, dep, sc).backward()
    torch.cuda.synchronize()
    g = {"grad_means3D": L[0].grad, "grad_scales": L[1].grad, "grad_rot": L[2].grad, "grad_opacity": L[3].grad,
         "grad_sh": L[4].grad, "grad_means2D": m2d.grad, "grad_tau": torch.cat([rho.grad, theta.grad])}
    return img, dep, opa, radii, nt, g


def _check(want, img, dep, opa, radii, nt, g, tau_tol=2e-3):
    t = lambda k: torch.from_numpy(np.asarray(want[k]))
    step = int(want["thin_step"]) if "thin_step" in want.files else 1
    assert (img.cpu() - t("image")).abs().mean().item() <= FWD_L1
    assert (dep.cpu() - t("depth")).abs().mean().item() <= 1e-3
    assert (opa.cpu() - t("opacity")).abs().mean().item() <= FWD_L1
    assert (radii.cpu() != t("radii")).float().mean().item() <= 1e-3
    ntw = t("n_touched")
    assert (nt.cpu() - ntw).abs().sum().item() <= 0.002 * ntw.sum().item() + 5
    for k, v in g.items():
        tol = tau_tol if k == "grad_tau" else BWD_REL
        if k in want.files:
            assert rel_err(v, t(k)) <= tol, (k, rel_err(v, t(k)))
        else:
            e = rel_err(v[::step], t(k + "_thin"))
            assert e <= tol, (k, e)
            n = float(want[k + "_norm"])
            assert abs(float(v.double().norm()) - n) <= tol * n, k


def test_hip_matches_the_crowded_mid_size_vectors(built):
    """33 000 Gaussians @ 320x240 with tiles of > 1024 and > 4096 splats."""
    from monogs_amd import rasterizer as R
    want = np.load(os.path.join(GOLD, "mid_crowded.npz"))
    out = _hip(scenes.crowded_scene())
    # exact culling removes pairs the reference binning would emit and never blend
    assert R.last_stats["pairs"] <= int(want["pairs"])
    _check(want, *out)


def _check_full(want, img, dep, opa, radii, nt, g=None):
    """Against the thinned full-size vectors of tests/golden/make_full_golden.py: images at every
    `pix_step`-th row / column plus whole-image mean and L1 mass, per-Gaussian arrays at every
    `g_step`-th Gaussian plus whole-array norms / sums."""
    ps, gs = int(want["pix_step"]), int(want["g_step"])
    for name, a, tol in (("image", img, FWD_L1), ("depth", dep, 1e-3), ("opacity", opa, FWD_L1)):
        a = a.detach().cpu()
        sub = torch.from_numpy(want[name + "_sub"])
        assert (a[:, ::ps, ::ps] - sub).abs().mean().item() <= tol, name
        assert abs(a.double().mean().item() - float(want[name + "_mean"])) <= tol, name
        assert abs(a.double().abs().sum().item() - float(want[name + "_l1"])) <= tol * a.numel(), name
    r = radii.cpu()
    assert (r[::gs] != torch.from_numpy(want["radii_thin"])).float().mean().item() <= 1e-3
    assert abs(int(r.long().sum()) - int(want["radii_sum"])) <= 1e-4 * int(want["radii_sum"])
    ntw = torch.from_numpy(want["n_touched_thin"])
    assert (nt.cpu()[::gs] - ntw).abs().sum().item() <= 0.002 * ntw.sum().item() + 5
    assert abs(int(nt.long().sum()) - int(want["n_touched_sum"])) <= 0.002 * int(want["n_touched_sum"])
    if g is None:
        return
    for k, v in g.items():
        tol = 2e-3 if k == "grad_tau" else BWD_REL
        if k == "grad_tau":
            e = rel_err(v, torch.from_numpy(want[k]))
        else:
            e = rel_err(v[::gs], torch.from_numpy(want[k + "_thin"]))
            n = float(want[k + "_norm"])
            assert abs(float(v.double().norm()) - n) <= tol * n, k
        assert e <= tol, (k, e)


def test_hip_matches_the_full_size_oracle_vectors_syn_b(built):
    """BASELINE config 2: 100 000 Gaussians @ 640x480, forward, against the torch oracle's committed
    (thinned) outputs - evidence at the configured size that does not share csrc/raster_math.h."""
    from monogs_amd import rasterizer as R
    from monogs_amd import synthetic as S
    want = np.load(os.path.join(GOLD, "syn_b_oracle.npz"))
    img, dep, opa, radii, nt, _ = _hip(S.make_scene(100000, 640, 480, seed=0))
    assert R.last_stats["pairs"] <= int(want["pairs"])
    assert int((radii > 0).sum()) == int(want["n_visible"])
    _check_full(want, img, dep, opa, radii, nt)


def test_hip_matches_the_full_size_oracle_vectors_syn_c(built):
    """BASELINE config 3 (the workload bench.py times): 300 000 Gaussians @ 640x480, forward and
    every gradient sink incl. dL/dtau, against the torch oracle's committed (thinned) outputs."""
    from monogs_amd import rasterizer as R
    from monogs_amd import synthetic as S
    want = np.load(os.path.join(GOLD, "syn_c_oracle.npz"))
    img, dep, opa, radii, nt, g = _hip(S.make_scene(300000, 640, 480, seed=0))
    assert R.last_stats["pairs"] <= int(want["pairs"])
    assert int((radii > 0).sum()) == int(want["n_visible"])
    _check_full(want, img, dep, opa, radii, nt, g)


def test_hip_matches_the_sh3_vectors(built):
    want = np.load(os.path.join(GOLD, "sh3.npz"))
    sc, shs, campos = scenes.sh3_inputs()
    _check(want, *_hip(sc, shs=shs, deg=3, campos=campos))


def test_clamp_gradient_modes_match_their_oracle_variants(built):
    """mgs_backward_args.clamp_gradient_mode: 0 = the treatment the absent CUDA extension is believed
    to use (default), 1 = exact derivative; each against the matching oracle variant on a scene
    with hundreds of visible splats beyond 1.3x the field of view."""
    from monogs_amd import rasterizer as R
    want = np.load(os.path.join(GOLD, "wide_clamp.npz"))
    sc = scenes.wide_scene()
    assert R._clamp_gradient_mode == R.CLAMP_GRADIENT_MODES["upstream"]      # the product default
    try:
        for mode in ("exact", "upstream"):
            R.set_clamp_gradient_mode(mode)
            img, dep, opa, radii, nt, g = _hip(sc)
            assert (img.cpu() - torch.from_numpy(want["image"])).abs().mean().item() <= FWD_L1
            for k in ("grad_means3D", "grad_scales", "grad_rot", "grad_opacity", "grad_sh"):
                e = rel_err(g[k], torch.from_numpy(want[f"{k}_{mode}"]))
                assert e <= BWD_REL, (mode, k, e)
            assert rel_err(g["grad_tau"], torch.from_numpy(want[f"grad_tau_{mode}"])) <= 2e-3
            other = "upstream" if mode == "exact" else "exact"
            # ... and NOT the other one: the two differ by ~1.6 % here
            assert rel_err(g["grad_means3D"], torch.from_numpy(want[f"grad_means3D_{other}"])) > 5e-3
    finally:
        R.set_clamp_gradient_mode("upstream")


def test_sketch_kat(built):
    """Sketched pose Jacobian, stack 4 / sketch 8, against the committed oracle rows
    (construction of utils/slam_frontend.py:1031-1127)."""
    from monogs_amd import synthetic as S
    from monogs_amd.rasterizer import GaussianRasterizer
    kat = np.load(os.path.join(GOLD, "sketch_kat.npz"))
    dev = _dev()
    sc, A, B, fsa = scenes.sketch_kat_setup()
    m, s, r, o, sh = S.activated(sc)
    N = m.shape[0]
    idx = fsa["sketch_indices"]
    stack, sketch = idx.shape[1], int(fsa["sketch_dim"])
    sk = torch.empty(stack, sketch, 6, device=dev, requires_grad=True)
    theta = torch.zeros(3, device=dev, requires_grad=True)
    rho = torch.zeros(3, device=dev, requires_grad=True)
    img, radii, dep, opa, nt = GaussianRasterizer(gpu_settings(sc.cam, sc.bg, dev))(
        means3D=m.to(dev), means2D=torch.zeros(N, 3, device=dev), shs=sh.to(dev), opacities=o.to(dev),
        scales=s.to(dev), rotations=r.to(dev), theta=theta, rho=rho, sketch_mode=1, sketch_dim=sketch,
        stack_dim=stack, sketch_dtau=sk, sketch_indices=idx.to(dev))
    assert (img.cpu() - torch.from_numpy(kat["image"])).abs().mean().item() <= FWD_L1
    res = (img * A.to(dev)).sum(0) + (dep * B.to(dev))[0]
    w = res * fsa["rand_weights"][0].to(dev)
    w.backward(gradient=torch.ones_like(w))
    SJ = torch.from_numpy(kat["SJ"])
    scale = SJ.abs().max().item()
    assert (sk.grad.cpu() - SJ).abs().max().item() <= 2e-3 * scale


def test_knn_4800(built):
    from monogs_amd.knn import distCUDA2
    want = torch.from_numpy(np.load(os.path.join(GOLD, "knn_4800.npz"))["dist2"])
    got = distCUDA2(scenes.knn_points().to(_dev())).cpu()
    assert torch.allclose(got, want, rtol=2e-5, atol=1e-12)
