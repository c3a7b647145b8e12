"""The CPU oracle against the committed vectors of tests/golden/ (they freeze the oracle: a change
of its arithmetic shows up here, on the CPU, before it can silently move the GPU parity target),
plus the oracle's own derivative checks (SURVEY §8c i): fp64 gradcheck and the two backward
treatments of the EWA clamp."""
import os

import numpy as np
import pytest
import torch

import scenes
from conftest import oracle_settings, rel_err
from monogs_amd import synthetic as S
from oracle import torch_raster as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _oracle(sc, shs=None, deg=0, campos=None, clamp_grad="exact"):
    m, s, r, o, sh0 = S.activated(sc)
    sh = sh0 if shs is None else shs
    L = [t.clone().requires_grad_() for t in (m, s, r, o, sh)]
    theta = torch.zeros(3, requires_grad=True)
    rho = torch.zeros(3, requires_grad=True)
    m2d = torch.zeros(m.shape[0], 3, requires_grad=True)
    img, radii, dep, opa, nt, info = O.rasterize(L[0], m2d, L[4], None, L[3], L[1], L[2], None,
                                                 oracle_settings(sc.cam, sc.bg, deg=deg, campos=campos), theta, rho,
                                                 clamp_grad=clamp_grad)
    S.synthetic_loss(img, dep, sc).backward()
    g = {"grad_means3D": L[0].grad, "grad_scales": L[1].grad, "grad_rot": L[2].grad, "grad_opacity": L[3].grad,
         "grad_sh": L[4].grad, "grad_means2D": m2d.grad, "grad_tau": torch.cat([rho.grad, theta.grad])}
    return img.detach(), dep.detach(), opa.detach(), radii, nt, g, info


def _check_against(want, img, dep, opa, radii, nt, g):
    t = lambda k: torch.from_numpy(np.asarray(want[k]))
    step = int(want["thin_step"]) if "thin_step" in want else 1
    assert torch.allclose(img, t("image"), atol=1e-6) and torch.allclose(dep, t("depth"), atol=1e-5)
    assert torch.allclose(opa, t("opacity"), atol=1e-6)
    assert torch.equal(radii.to(torch.int32), t("radii")) and torch.equal(nt.to(torch.int32), t("n_touched"))
    for k, v in g.items():
        if k in want.files:
            assert rel_err(v, t(k)) < 1e-5, k
        else:
            assert rel_err(v[::step], t(k + "_thin")) < 1e-5, k
            assert abs(float(v.double().norm()) - float(want[k + "_norm"])) < 1e-5 * float(want[k + "_norm"]), k


def test_oracle_reproduces_the_sh3_vectors():
    want = np.load(os.path.join(GOLD, "sh3.npz"))
    sc, shs, campos = scenes.sh3_inputs()
    img, dep, opa, radii, nt, g, _ = _oracle(sc, shs=shs, deg=3, campos=campos)
    _check_against(want, img, dep, opa, radii, nt, g)


def test_oracle_reproduces_the_crowded_mid_size_vectors():
    """33 000 Gaussians @ 320x240 (the slowest CPU test, ~30 s): tiles of > 1024 and > 4096 splats."""
    want = np.load(os.path.join(GOLD, "mid_crowded.npz"))
    assert int(want["tiles_over_1024"]) >= 3 and int(want["tiles_over_4096"]) >= 1
    img, dep, opa, radii, nt, g, info = _oracle(scenes.crowded_scene())
    assert info["pairs"] == int(want["pairs"])
    _check_against(want, img, dep, opa, radii, nt, g)


def test_clamp_gradient_variants_of_the_oracle():
    """Forward identical; on SYN-A (BASELINE config 1) no visible splat is clamped, so the two
    treatments give the same gradients; on the wide scene they differ by the recorded amount
    (~1.6 % of dL/dmeans3D, ~0.7 % of dL/dtau)."""
    want = np.load(os.path.join(GOLD, "wide_clamp.npz"))
    assert int(want["clamped_visible"]) > 300
    sc = scenes.wide_scene()
    ex = _oracle(sc, clamp_grad="exact")
    up = _oracle(sc, clamp_grad="upstream")
    assert torch.equal(ex[0], up[0]) and torch.equal(ex[1], up[1])
    for k in ("grad_means3D", "grad_scales", "grad_rot", "grad_opacity", "grad_sh", "grad_tau"):
        assert rel_err(ex[5][k], torch.from_numpy(want[k + "_exact"])) < 1e-5, k
        assert rel_err(up[5][k], torch.from_numpy(want[k + "_upstream"])) < 1e-5, k
    d_mean = rel_err(up[5]["grad_means3D"], ex[5]["grad_means3D"])
    d_tau = rel_err(up[5]["grad_tau"], ex[5]["grad_tau"])
    assert 5e-3 < d_mean < 5e-2 and 1e-3 < d_tau < 5e-2, (d_mean, d_tau)
    assert rel_err(up[5]["grad_scales"], ex[5]["grad_scales"]) < 1e-7       # the covariance path is untouched
    sa = S.make_scene(5000, 160, 120, seed=0)
    a, b = _oracle(sa, clamp_grad="exact"), _oracle(sa, clamp_grad="upstream")
    for k in a[5]:
        assert rel_err(b[5][k], a[5][k]) < 1e-6, k


def test_sketch_kat_and_knn_vectors():
    want = np.load(os.path.join(GOLD, "knn_4800.npz"))
    pts = scenes.knn_points()
    d2 = O.dist2_knn3(pts)
    assert torch.allclose(d2, torch.from_numpy(want["dist2"]), rtol=1e-6, atol=0)
    from scipy.spatial import cKDTree
    dd, _ = cKDTree(pts.double().numpy()).query(pts.double().numpy(), k=4)
    assert np.allclose((dd[:, 1:] ** 2).mean(1), d2.numpy(), rtol=2e-5)
    # the sketched Jacobian: bucket rows add up to the full pose gradient of the functional
    kat = np.load(os.path.join(GOLD, "sketch_kat.npz"))
    sc, A, B, fsa = scenes.sketch_kat_setup()
    m, s, r, o, sh = S.activated(sc)
    th = torch.zeros(3, requires_grad=True)
    rh = torch.zeros(3, requires_grad=True)
    oimg, _, odep, _, _, _ = O.rasterize(m, None, sh, None, o, s, r, None, oracle_settings(sc.cam, sc.bg), th, rh)
    assert torch.allclose(oimg.detach(), torch.from_numpy(kat["image"]), atol=1e-6)
    w = ((oimg * A).sum(0) + (odep * B)[0]) * fsa["rand_weights"][0]
    idx = fsa["sketch_indices"][0]
    SJ = torch.from_numpy(kat["SJ"])
    for st_ in range(idx.shape[0]):
        th.grad = None
        rh.grad = None
        w[idx[st_] >= 0].sum().backward(retain_graph=True)
        assert rel_err(SJ[st_].sum(0), torch.cat([rh.grad, th.grad])) < 1e-4


def test_oracle_fp64_gradcheck():
    """torch.autograd.gradcheck of the oracle in double precision on 12 Gaussians @ 32x32
    (SURVEY §8c i): every differentiable input incl. the pose perturbation (rho, theta)."""
    torch.manual_seed(0)
    W = H = 32
    N = 12
    sc = S.make_scene(N, W, H, seed=4)
    cam = sc.cam
    dt = torch.float64
    st = oracle_settings(cam, torch.tensor([0.1, 0.2, 0.3]), dtype=dt)
    m = sc.means3D.to(dt)
    s = (torch.exp(sc.log_scales) * 6.0).to(dt)            # several pixels wide: every splat covers pixels
    r = torch.nn.functional.normalize(sc.rot).to(dt)
    o = torch.sigmoid(sc.opacity_logit).clamp(0.2, 0.85).to(dt)   # below the 0.99 cap (straight-through there)
    sh = sc.features_dc.to(dt)
    leaves = [t.clone().requires_grad_() for t in (m, s, r, o, sh)]

    def f(m_, s_, r_, o_, sh_):
        img, _, dep, _, _, _ = O.rasterize(m_, None, sh_, None, o_, s_, r_, None, st, clamp_grad="exact")
        return img, dep

    img, _, dep, _, nt, info = O.rasterize(leaves[0], None, leaves[4], None, leaves[3], leaves[1], leaves[2], None, st, clamp_grad="exact")
    assert int((nt > 0).sum()) >= N // 2 and float(img.detach().std()) > 0.01
    assert torch.autograd.gradcheck(f, tuple(leaves), eps=1e-6, atol=1e-6, rtol=1e-4, nondet_tol=0.0)

    # Pose: the extension's contract evaluates the Jacobian at tau = 0 whatever values theta / rho
    # hold (they are autograd leaves only, pose_utils.py:88-98), so the forward does not depend on
    # them and gradcheck cannot see them.  Check d/dtau of a random functional through the (rho, theta)
    # leaves against central differences of the SAME functional rendered with T = Exp(tau) T0.
    g = torch.Generator().manual_seed(1)
    A = torch.randn(3, H, W, generator=g, dtype=dt)
    B = torch.randn(1, H, W, generator=g, dtype=dt)
    theta = torch.zeros(3, dtype=dt, requires_grad=True)
    rho = torch.zeros(3, dtype=dt, requires_grad=True)
    img, _, dep, _, _, _ = O.rasterize(m, None, sh, None, o, s, r, None, st, theta, rho, clamp_grad="exact")
    ((img * A).sum() + (dep * B).sum()).backward()
    analytic = torch.cat([rho.grad, theta.grad])

    def functional(tau):
        T = O.se3_exp(tau) @ st.viewmatrix.t()
        V = T.t().contiguous()
        st2 = st._replace(viewmatrix=V, projmatrix=(V @ st.projmatrix_raw).contiguous(), campos=V)
        i2, _, d2, _, _, _ = O.rasterize(m, None, sh, None, o, s, r, None, st2, clamp_grad="exact")
        return float((i2 * A).sum() + (d2 * B).sum())

    h = 1e-6
    fd = torch.zeros(6, dtype=dt)
    for k in range(6):
        e = torch.zeros(6, dtype=dt)
        e[k] = h
        fd[k] = (functional(e) - functional(-e)) / (2 * h)
    assert rel_err(analytic, fd) < 1e-6, (analytic, fd)


def test_full_size_forward_vectors_freeze_the_oracle():
    """tests/golden/syn_b_oracle.npz (BASELINE config 2: 100 000 @ 640x480, forward) re-derived here;
    syn_c_oracle.npz (300 000, forward + backward: ~100 s of container time) is regenerated by
    tests/golden/make_full_golden.py only and checked against the HIP path on the GPU box."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_full_golden", os.path.join(GOLD, "make_full_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    got = mod.run(100000, False)
    want = np.load(os.path.join(GOLD, "syn_b_oracle.npz"))
    assert int(got["pairs"]) == int(want["pairs"]) and int(got["n_visible"]) == int(want["n_visible"])
    for k in ("image_sub", "depth_sub", "opacity_sub"):
        assert np.abs(got[k] - want[k]).max() <= 1e-6, k
    assert np.array_equal(got["radii_thin"], want["radii_thin"])
    assert np.array_equal(got["n_touched_thin"], want["n_touched_thin"])
    c = np.load(os.path.join(GOLD, "syn_c_oracle.npz"))
    assert int(c["n_visible"]) > 250000 and c["grad_tau"].shape == (6,) and int(c["g_step"]) == 16
    # the forward half of syn_c is re-derived as well (about ten seconds); its gradients are not
    gotc = mod.run(300000, False)
    assert int(gotc["pairs"]) == int(c["pairs"]) and int(gotc["n_visible"]) == int(c["n_visible"])
    for k in ("image_sub", "depth_sub", "opacity_sub"):
        assert np.abs(gotc[k] - c[k]).max() <= 1e-6, k
    assert np.array_equal(gotc["radii_thin"], c["radii_thin"]) and np.array_equal(gotc["n_touched_thin"], c["n_touched_thin"])


def test_sketch_problem_restatement_matches_the_reference():
    """oracle/sketch_problem.py against what the REFERENCE's tests/sketch_utils.py returned on the same
    seeds (tests/golden/sketch_bound.npz, made by make_sketch_bound.py): the problem generator, the
    CountSketch matrix, the distortion and the two bounds run_test asserts (sketch_utils.py:58-124)."""
    from scipy.linalg import lstsq
    from oracle import sketch_problem as SP
    G = np.load(os.path.join(GOLD, "sketch_bound.npz"))
    n, noise, lam, x_norm, smax, smin = G["params"]
    assert (int(n), lam) == (SP.REFERENCE_TEST["n"], SP.REFERENCE_TEST["lambda_"])
    for tag in ("small", "large"):
        m, seed = int(G[f"{tag}_m"]), int(G[f"{tag}_seed"])
        A, b, x = SP.gen_problem(m, int(n), smax, smin, lam, noise, x_norm, seed=seed)
        if tag == "small":
            assert np.allclose(A, G["small_A"], rtol=1e-5, atol=1e-9) and np.allclose(b, G["small_b"], rtol=1e-5, atol=1e-9)
        else:
            assert np.allclose(A[::64], G["large_A_rows64"], rtol=1e-5, atol=1e-9)
            assert np.allclose(b[::64], G["large_b_rows64"], rtol=1e-5, atol=1e-9)
        assert np.allclose(A.T @ A, G[f"{tag}_AtA"], rtol=1e-9, atol=1e-12)
        assert np.allclose(A.T @ b, G[f"{tag}_Atb"], rtol=1e-9, atol=1e-12)
        assert np.allclose(x, G[f"{tag}_x"], rtol=1e-12)
        # the reference's own (unsigned, with-replacement) sketch: same solution, distortion and bounds
        S = SP.count_sketch(G[f"{tag}_ref_bucket"].astype(np.int64), 32)
        SA, Sb = S @ A, S @ b
        At, bt = SP.damped(SA, Sb, lam)
        x_sketch = lstsq(At, bt)[0]
        assert np.allclose(x_sketch, G[f"{tag}_ref_x_sketch"], rtol=1e-7, atol=1e-13)
        x_opt, ub, ub_hat, st = SP.bounds(A, b, lam, SA, Sb, x_sketch, 32)
        assert np.allclose(x_opt, G[f"{tag}_x_opt"], rtol=1e-8, atol=1e-14)
        assert abs(st["res"] - float(G[f"{tag}_res"])) < 1e-10 and abs(st["sigma_min"] - float(G[f"{tag}_sigma_min"])) < 1e-9
        assert abs(st["distortion"] - float(G[f"{tag}_ref_distortion"])) < 1e-9
        assert np.allclose([ub, ub_hat], G[f"{tag}_ref_bounds"], rtol=1e-6)
        assert np.linalg.norm(x_opt - x_sketch) < min(ub, ub_hat)
