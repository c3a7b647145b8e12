"""oracle/map_update_ref.py and monogs_amd.losses.ApplyExposure against what the REFERENCE's own code returned
(tests/golden/map_update_ref.npz, produced by tests/golden/make_map_update_golden.py from the imported
gaussian_splatting/scene/gaussian_model.py:364-377,485-697 and utils/slam_utils.py:115-185).  CPU only; the GPU
suite checks the HIP path against the same arrays (tests/test_raster_gpu.py, tests/test_gpu_tracking_best.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import map_update_ref as MR

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "map_update_ref.npz")
STATE_KEYS = ([n for n in MR.PARAMS] + ["exp_avg_" + n for n in MR.PARAMS] + ["exp_avg_sq_" + n for n in MR.PARAMS]
              + ["kf", "n_obs", "grad_accum", "denom", "max_radii"])


@pytest.fixture(scope="module")
def G():
    return np.load(GOLD)


def load_state(G, prefix):
    return {k: torch.from_numpy(G[f"{prefix}_{k}"]) for k in STATE_KEYS}


def assert_state(got, want, exact_keys=(), rtol=1e-6, atol=1e-7):
    for k in STATE_KEYS:
        g, w = got[k], want[k]
        assert tuple(g.shape) == tuple(w.shape), (k, g.shape, w.shape)
        if w.dtype.is_floating_point and k not in exact_keys:
            assert torch.allclose(g, w, rtol=rtol, atol=atol), (k, float((g - w).abs().max()))
        else:
            assert torch.equal(g.to(w.dtype), w), k


@pytest.mark.parametrize("tag", ["dp_screen", "dp_plain"])
def test_densify_and_prune_restatement_matches_the_reference(G, tag):
    before, want = load_state(G, tag + "_in"), load_state(G, tag + "_out")
    max_grad, min_opacity, extent, max_screen, percent_dense = (float(x) for x in G[tag + "_args"])
    noise = torch.from_numpy(G[tag + "_unit_noise"])
    got = MR.densify_and_prune({k: v.clone() for k, v in before.items()}, max_grad, min_opacity, extent,
                               max_screen if max_screen > 0 else None, percent_dense, noise)
    assert got["xyz"].shape[0] != before["xyz"].shape[0]
    # rows, order, copies, Adam moments, ids and the zeroed statistics are exact; the split children's
    # position / scale go through exp / log / a matrix product whose summation order may differ by an ulp
    assert_state(got, want, exact_keys=("f_dc", "f_rest", "opacity", "rotation", "exp_avg_xyz", "exp_avg_sq_xyz"))


def test_prune_points_restatement_matches_the_reference(G):
    got = MR.prune_points(load_state(G, "dp_screen_out"), torch.from_numpy(G["pp_mask"]))
    want = load_state(G, "pp_out")
    assert_state(got, want, exact_keys=STATE_KEYS)


def test_add_densification_stats_restatement_matches_the_reference(G):
    ga, dn = MR.add_densification_stats(torch.from_numpy(G["ads_in_grad_accum"]), torch.from_numpy(G["ads_in_denom"]),
                                        torch.from_numpy(G["ads_viewspace_grad"]), torch.from_numpy(G["ads_filter"]))
    assert torch.allclose(ga, torch.from_numpy(G["ads_out_grad_accum"]), rtol=1e-6, atol=0)
    assert torch.equal(dn, torch.from_numpy(G["ads_out_denom"]))


def test_opacity_resets_restatement_matches_the_reference(G):
    st = load_state(G, "pp_out")
    filters = [torch.from_numpy(f) for f in G["ron_filters"]]
    a = MR.reset_opacity_nonvisible(st, filters)
    assert torch.allclose(a["opacity"], torch.from_numpy(G["ron_out_opacity"]), rtol=1e-6, atol=1e-7)
    assert torch.equal(a["exp_avg_opacity"], torch.from_numpy(G["ron_out_exp_avg"]))
    assert torch.equal(a["exp_avg_sq_opacity"], torch.from_numpy(G["ron_out_exp_avg_sq"]))
    # the quirk of gaussian_model.py:375 is real: a visible Gaussian's new logit is sigmoid(old logit)
    seen = torch.stack(filters).any(0)
    assert bool(seen.any()) and bool((~seen).any())
    assert torch.allclose(a["opacity"][seen], torch.sigmoid(st["opacity"][seen]), rtol=1e-6)
    assert not torch.allclose(a["opacity"][seen], st["opacity"][seen], atol=1e-2)
    b = MR.reset_opacity(a)
    assert torch.allclose(b["opacity"], torch.from_numpy(G["ro_out_opacity"]), rtol=1e-6)
    assert torch.equal(b["exp_avg_opacity"], torch.from_numpy(G["ro_out_exp_avg"]))
    assert torch.equal(b["exp_avg_sq_opacity"], torch.from_numpy(G["ro_out_exp_avg_sq"]))
    assert torch.equal(b["exp_avg_xyz"], torch.from_numpy(G["ro_out_exp_avg_xyz"]))      # other groups untouched


@pytest.mark.parametrize("tag", ["pos", "neg"])
def test_apply_exposure_matches_the_reference_in_sketch_mode(G, tag):
    """losses.ApplyExposure (the host-side mirror the Python second-order body uses) forward and both repeats of the
    backward, exposure_a > 0 and < 0: grad_image = |a| grad (no eps), grad_a = sum(grad image) (no sign(a)),
    slam_utils.py:145-149."""
    from monogs_amd.losses import ApplyExposure
    a0, b0, eps = (float(x) for x in G[f"ae_{tag}_a_b_eps"])
    rows, cols = torch.from_numpy(G["ae_rows"]), torch.from_numpy(G["ae_cols"])
    repeat, stack, sketch, _ = rows.shape
    im = torch.from_numpy(G["ae_image"]).clone().requires_grad_()
    a = torch.tensor([a0], requires_grad=True)
    b = torch.tensor([b0], requires_grad=True)
    sk = torch.zeros(stack, sketch, 2, requires_grad=True)
    y = ApplyExposure.apply(im, a, b, eps, 1, sketch, stack, (rows, cols), sk)
    assert torch.equal(y.detach(), torch.from_numpy(G[f"ae_{tag}_forward"]))
    for r in range(repeat):
        for p in (im, a, b, sk):
            p.grad = None
        y.backward(gradient=torch.from_numpy(G["ae_grad_out"][r]), retain_graph=True)
        for name, t in (("grad_image", im), ("grad_a", a), ("grad_b", b), ("grad_sketch", sk)):
            want = torch.from_numpy(G[f"ae_{tag}_r{r}_{name}"])
            assert torch.allclose(t.grad, want, rtol=1e-6, atol=1e-7), (tag, r, name)
    if tag == "neg":      # not the exact derivative: d/da of (|a| + eps) image would carry sign(a) = -1
        exact = -(torch.from_numpy(G["ae_grad_out"][repeat - 1]) * im.detach()).sum()
        assert float(a.grad) * float(exact) < 0


def test_learning_rate_schedule_matches_the_reference(G):
    """GaussianModel.update_learning_rate (gaussian_model.py:298-312 -> general_utils.helper :80-95) at the iterations the
    fixture holds, through the product's host-side mirror (monogs_amd.gaussian_model.expon_lr; pure Python, no GPU)."""
    from monogs_amd.gaussian_model import expon_lr
    init, final, delay_mult, max_steps, scale = (float(x) for x in G["lr_setup"])
    for it, want in zip(G["lr_iterations"], G["lr_values"]):
        got = expon_lr(int(it), init * scale, final * scale, lr_delay_mult=delay_mult, max_steps=int(max_steps))
        assert abs(got - float(want)) <= 1e-12 * abs(float(want)), (int(it), got, float(want))
    assert abs(float(G["lr_values"][0]) - init * scale) < 1e-15 and abs(float(G["lr_values"][-1]) - final * scale) < 1e-15


def test_extend_from_pcd_restatement_matches_the_reference(G):
    got = MR.extend_from_pcd(load_state(G, "ext_in"), *(torch.from_numpy(G[f"ext_new_{k}"]) for k in
                             ("xyz", "features", "scales", "rots", "opacities")), int(G["ext_kf_id"]))
    assert_state(got, load_state(G, "ext_out"), exact_keys=STATE_KEYS)


def test_se3_exponential_matches_the_reference(G):
    """monogs_amd.pose.SE3_exp / SO3_exp / V (what update_pose applies, host side) against the reference's own in-tree
    closed form (utils/pose_utils.py:13-74) incl. the identity, an angle below its small-angle threshold and one near
    pi.  (The reference's update_pose calls lietorch.SE3.exp, pose_utils.py:92: lietorch is absent, so this closed form
    - also checked against torch.linalg.matrix_exp in tests/test_cpu_oracle.py - is as far as the pin goes.)"""
    from monogs_amd.pose import SE3_exp, SO3_exp, V
    taus = torch.from_numpy(G["pose_taus"])
    for i, t in enumerate(taus):
        assert torch.allclose(SE3_exp(t), torch.from_numpy(G["pose_SE3_exp_f64"][i]), rtol=0, atol=1e-14), i
        assert torch.allclose(SO3_exp(t[3:]), torch.from_numpy(G["pose_SO3_exp_f64"][i]), rtol=0, atol=1e-14), i
        assert torch.allclose(V(t[3:]), torch.from_numpy(G["pose_V_f64"][i]), rtol=0, atol=1e-14), i
        assert torch.allclose(SE3_exp(t.float()), torch.from_numpy(G["pose_SE3_exp_f32"][i]), rtol=0, atol=2e-6), i


def test_ssim_matches_the_reference(G):
    """eval_metrics.ssim (what the offline evaluation harness scores renders with, row f4) against the reference's own
    gaussian_splatting/utils/loss_utils.py:61-101 (11x11 Gaussian window, sigma 1.5, zero padding, C1 / C2), mean and
    per-image forms."""
    from monogs_amd import eval_metrics as E
    a, b = torch.from_numpy(G["ssim_a"]), torch.from_numpy(G["ssim_b"])
    assert abs(float(E.ssim(a, b)) - float(G["ssim_mean"])) < 1e-6
    assert torch.allclose(E.ssim(a, b, size_average=False), torch.from_numpy(G["ssim_per_image"]), atol=1e-6)
    assert abs(float(E.ssim(a[:1], b[:1])) - float(G["ssim_single"])) < 1e-6
    assert abs(float((a - b).abs().mean()) - float(G["l1_loss"])) < 1e-7
