"""Golden vectors for the map-maintenance glue and the sketched exposure Jacobian, produced by RUNNING the
reference's own Python on the CPU (build container only; the reference tree does not exist on the GPU box):

    python tests/golden/make_map_update_golden.py      ->  tests/golden/map_update_ref.npz

What is run (nothing of it is copied; only arrays - inputs and what the reference returned - are stored):
  * gaussian_splatting/scene/gaussian_model.py: GaussianModel.training_setup (:247-285) with torch.optim.Adam
    steps to populate the moments, then densify_and_prune (:674-691; both max_screen_size cases, with and
    without f_rest) -> densify_and_clone (:636-672), densify_and_split (:598-634), densification_postfix /
    cat_tensors_to_optimizer (:525-596), prune_points / _prune_optimizer (:485-556); reset_opacity and
    reset_opacity_nonvisible (:364-377, replace_tensor_to_optimizer :470-483); add_densification_stats (:693-697);
    update_learning_rate (:298-312, general_utils.helper :80-95); extend_from_pcd (:210-236)
  * utils/slam_utils.py: ApplyExposure forward / backward in sketch mode (:115-185), two repeats over one
    forward, exposure_a positive AND negative (the backward is not the exact derivative: no sign(a), no eps)
  * utils/pose_utils.py: the in-tree SE(3) exponential SE3_exp / SO3_exp / V (:13-74) in fp64 and fp32
  * gaussian_splatting/utils/loss_utils.py: ssim (:61-101) and l1_loss (:21-22), what eval_rendering scores with

Import recipe = tests/test_cpu_reference_binding.py: utils.configs.cuda_device = "cpu" before anything else,
EMPTY `open3d` / `plyfile` / `lietorch` / `cv2` modules (not installed; nothing of any of them runs on these paths), `simple_knn` from
dropin/.  `torch.cuda.synchronize` (called by ApplyExposure.backward for its timers, :134,181) is replaced by a
no-op in THIS script.  The split's random draw (`torch.normal(mean=0, std=stds)`, :608-609) is recovered as unit
normals by replaying the same generator state with std = 1 (asserted to reproduce the reference's samples bit
for bit when multiplied by stds).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path[:0] = [ROOT + "/dropin", ROOT, REF]
for name in ("open3d", "plyfile", "lietorch", "cv2"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
import utils.configs as _cfg  # noqa: E402

_cfg.cuda_device = "cpu"
from gaussian_splatting.scene.gaussian_model import GaussianModel  # noqa: E402
import utils.slam_utils as SU  # noqa: E402

torch.cuda.synchronize = lambda *a, **k: None      # ApplyExposure.backward's timers; CPU tensors need none

out = {}
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
        "scaling": "_scaling", "rotation": "_rotation"}


class Opt:      # the fields training_setup reads (configs/mono/tum/base_config.yaml: opt_params)
    percent_dense = 0.01
    position_lr_init, position_lr_final = 0.0016, 0.0000016
    position_lr_delay_mult, position_lr_max_steps = 0.01, 30000
    feature_lr, opacity_lr, scaling_lr, rotation_lr = 0.0025, 0.05, 0.001, 0.001


def make_model(n, rest, seed):
    g = torch.Generator().manual_seed(seed)
    m = GaussianModel(sh_degree=1 if rest else 0)
    m.init_lr(6.0)
    m._xyz = torch.nn.Parameter(torch.randn(n, 3, generator=g) * 2)
    m._features_dc = torch.nn.Parameter(torch.randn(n, 1, 3, generator=g))
    m._features_rest = torch.nn.Parameter(torch.randn(n, rest, 3, generator=g))
    m._opacity = torch.nn.Parameter(torch.randn(n, 1, generator=g) * 2)
    m._scaling = torch.nn.Parameter(torch.randn(n, 3, generator=g) * 0.8 - 3.0)
    m._rotation = torch.nn.Parameter(torch.randn(n, 4, generator=g))
    m.training_setup(Opt)
    for it in range(2):       # populate the Adam moments through the reference's own optimiser
        for a in ATTR.values():
            p = getattr(m, a)
            p.grad = torch.randn(p.shape, generator=g) * 1e-2
        m.optimizer.step()
    m.xyz_gradient_accum = torch.rand(n, 1, generator=g) * 4e-4
    m.denom = torch.randint(0, 3, (n, 1), generator=g).float()        # zeros -> NaN grads (:676)
    m.max_radii2D = torch.rand(n, generator=g) * 30
    m.unique_kfIDs = torch.randint(0, 9, (n,), generator=g).int()
    m.n_obs = torch.randint(0, 5, (n,), generator=g).int()
    return m, g


def state(m):
    st = {}
    for name, a in ATTR.items():
        p = getattr(m, a)
        st[name] = p.detach().numpy().copy()
        s = m.optimizer.state[p]
        st["exp_avg_" + name] = s["exp_avg"].numpy().copy()
        st["exp_avg_sq_" + name] = s["exp_avg_sq"].numpy().copy()
    st["kf"], st["n_obs"] = m.unique_kfIDs.numpy().copy(), m.n_obs.numpy().copy()
    st["grad_accum"], st["denom"] = m.xyz_gradient_accum.numpy().copy(), m.denom.numpy().copy()
    st["max_radii"] = m.max_radii2D.numpy().copy()
    return st


def put(prefix, st):
    for k, v in st.items():
        out[f"{prefix}_{k}"] = v


# ---- densify_and_prune, two cases -----------------------------------------------------------------------
for tag, n, rest, max_screen, seed in (("dp_screen", 900, 3, 20, 11), ("dp_plain", 700, 0, None, 12)):
    m, g = make_model(n, rest, seed)
    before = state(m)
    put(tag + "_in", before)
    extent, max_grad, min_opacity = 6.0, 2e-4, 0.1
    # the split's draw (:608-609), recorded: torch.normal is wrapped IN THIS SCRIPT for the duration of the call -
    # the reference's samples are returned unchanged, and the same generator state is replayed with std = 1
    captured = {}
    _normal = torch.normal

    def recording_normal(mean, std):
        rng = torch.get_rng_state()
        samples = _normal(mean=mean, std=std)
        torch.set_rng_state(rng)
        unit = _normal(mean=torch.zeros_like(mean), std=torch.ones_like(std))
        assert torch.equal(unit * std + mean, samples)       # bit for bit what the reference drew
        captured["unit"], captured["n"] = unit, std.shape[0] // 2
        return samples

    torch.manual_seed(1000 + seed)
    torch.normal = recording_normal
    try:
        m.densify_and_prune(max_grad, min_opacity, extent, max_screen)
    finally:
        torch.normal = _normal
    unit, n_split = captured["unit"], captured["n"]
    assert n_split > 10
    after = state(m)
    assert after["xyz"].shape[0] != n
    # the recovered unit normals reproduce the reference's draw: a split child's position is
    # R(q) (unit * scale) + parent (:609-613); checked in tests against these stored outputs
    out[tag + "_unit_noise"] = unit.numpy()
    out[tag + "_args"] = np.array([max_grad, min_opacity, extent, max_screen or 0, Opt.percent_dense], dtype=np.float64)
    out[tag + "_n_split"] = np.array(n_split)
    put(tag + "_out", after)
    # leaves registered in the optimiser (the product must leave the same structure)
    for grp in m.optimizer.param_groups:
        assert grp["params"][0] is getattr(m, ATTR[grp["name"]]) and grp["params"][0].requires_grad

    if tag == "dp_screen":
        # ---- prune_points(mask) alone on the densified model (slam_backend.py:86,280) -------------------
        mask = torch.rand(after["xyz"].shape[0], generator=g) < 0.3
        m.prune_points(mask)
        out["pp_mask"] = mask.numpy()
        put("pp_out", state(m))
        # ---- add_densification_stats (:693-697) ----------------------------------------------------------
        n2 = m.get_xyz.shape[0]
        m.xyz_gradient_accum = torch.rand(n2, 1, generator=g) * 1e-3
        m.denom = torch.randint(0, 4, (n2, 1), generator=g).float()
        vsp = torch.zeros(n2, 3, requires_grad=True)
        vsp.grad = torch.randn(n2, 3, generator=g) * 1e-3
        filt = torch.rand(n2, generator=g) < 0.6
        out["ads_in_grad_accum"], out["ads_in_denom"] = m.xyz_gradient_accum.numpy().copy(), m.denom.numpy().copy()
        out["ads_viewspace_grad"], out["ads_filter"] = vsp.grad.numpy().copy(), filt.numpy()
        m.add_densification_stats(vsp, filt)
        out["ads_out_grad_accum"], out["ads_out_denom"] = m.xyz_gradient_accum.numpy().copy(), m.denom.numpy().copy()
        # ---- reset_opacity_nonvisible (:370-377), then reset_opacity (:364-368) --------------------------
        filters = [torch.rand(n2, generator=g) < 0.3 for _ in range(3)]
        out["ron_filters"] = torch.stack(filters).numpy()
        m.reset_opacity_nonvisible(filters)
        s = m.optimizer.state[m._opacity]
        out["ron_out_opacity"] = m._opacity.detach().numpy().copy()
        out["ron_out_exp_avg"], out["ron_out_exp_avg_sq"] = s["exp_avg"].numpy().copy(), s["exp_avg_sq"].numpy().copy()
        assert m.optimizer.param_groups[3]["params"][0] is m._opacity
        m.reset_opacity()
        s = m.optimizer.state[m._opacity]
        out["ro_out_opacity"] = m._opacity.detach().numpy().copy()
        out["ro_out_exp_avg"], out["ro_out_exp_avg_sq"] = s["exp_avg"].numpy().copy(), s["exp_avg_sq"].numpy().copy()
        # the other groups' moments are untouched by either reset
        out["ro_out_exp_avg_xyz"] = m.optimizer.state[m._xyz]["exp_avg"].numpy().copy()

# ---- update_learning_rate (:298-312 -> general_utils.helper :80-95) and extend_from_pcd (:210-236) -----------
m, g = make_model(400, 3, 21)
its = np.array([0, 1, 10, 150, 1500, 15000, 29999, 30000, 45000], dtype=np.int64)
out["lr_iterations"] = its
out["lr_setup"] = np.array([Opt.position_lr_init, Opt.position_lr_final, Opt.position_lr_delay_mult,
                            Opt.position_lr_max_steps, 6.0], dtype=np.float64)      # ..., spatial_lr_scale
out["lr_values"] = np.array([m.update_learning_rate(int(i)) for i in its], dtype=np.float64)
assert m.optimizer.param_groups[0]["lr"] == out["lr_values"][-1]
put("ext_in", state(m))
P, K = 57, 4                  # a keyframe's new points; features arrive as [P, 3, K] (create_pcd_from_image :199-205)
new_xyz = torch.randn(P, 3, generator=g)
new_feat = torch.randn(P, 3, K, generator=g)
new_scales, new_rots, new_opac = torch.randn(P, 3, generator=g), torch.randn(P, 4, generator=g), torch.randn(P, 1, generator=g)
out["ext_new_xyz"], out["ext_new_features"] = new_xyz.numpy().copy(), new_feat.numpy().copy()
out["ext_new_scales"], out["ext_new_rots"], out["ext_new_opacities"] = new_scales.numpy().copy(), new_rots.numpy().copy(), new_opac.numpy().copy()
out["ext_kf_id"] = np.array(13)
m.extend_from_pcd(new_xyz, new_feat, new_scales, new_rots, new_opac, 13)
put("ext_out", state(m))

# ---- ApplyExposure in sketch mode (slam_utils.py:115-185) -------------------------------------------------
g = torch.Generator().manual_seed(77)
H, W, stack, sketch, repeat = 24, 32, 4, 8, 2
chunk = H * W // (stack * sketch)
image = torch.rand(3, H, W, generator=g)
out["ae_image"] = image.numpy()
rows = torch.empty(repeat, stack, sketch, chunk, dtype=torch.long)
cols = torch.empty_like(rows)
for r in range(repeat):       # a random partition per repeat, as slam_frontend.py:297-314 builds it
    perm = torch.randperm(H * W, generator=g)[: stack * sketch * chunk].view(stack, sketch, chunk)
    rows[r], cols[r] = perm // W, perm % W
out["ae_rows"], out["ae_cols"] = rows.numpy(), cols.numpy()
grad_outs = [torch.randn(3, H, W, generator=g) for _ in range(repeat)]
out["ae_grad_out"] = torch.stack(grad_outs).numpy()
for tag, a0 in (("pos", 0.9), ("neg", -0.7)):
    im = image.clone().requires_grad_()
    a = torch.tensor([a0], requires_grad=True)
    b = torch.tensor([0.05], requires_grad=True)
    sk = torch.zeros(stack, sketch, 2, requires_grad=True)
    y = SU.ApplyExposure.apply(im, a, b, 1e-8, 1, sketch, stack, (rows, cols), sk)
    out[f"ae_{tag}_a_b_eps"] = np.array([a0, 0.05, 1e-8])
    out[f"ae_{tag}_forward"] = y.detach().numpy().copy()
    for r in range(repeat):
        for p in (im, a, b, sk):
            p.grad = None
        y.backward(gradient=grad_outs[r], retain_graph=True)
        out[f"ae_{tag}_r{r}_grad_image"] = im.grad.numpy().copy()
        out[f"ae_{tag}_r{r}_grad_a"], out[f"ae_{tag}_r{r}_grad_b"] = a.grad.numpy().copy(), b.grad.numpy().copy()
        out[f"ae_{tag}_r{r}_grad_sketch"] = sk.grad.numpy().copy()

# ---- the in-tree SE(3) exponential (utils/pose_utils.py:13-74: skew_sym_mat, SO3_exp, V, SE3_exp) --------------------
# pose_utils imports lietorch at its top (not installed: an EMPTY module stands in, as for open3d / plyfile); the four
# functions run here are pure torch.  update_pose itself (:88-98) calls lietorch.SE3.exp and cannot run: it stays
# pinned only through this closed form.
import utils.pose_utils as PU  # noqa: E402
g = torch.Generator().manual_seed(5)
taus = torch.randn(12, 6, generator=g, dtype=torch.float64) * torch.tensor([0.3, 0.3, 0.3, 0.8, 0.8, 0.8], dtype=torch.float64)
taus[0] = 0.0                                     # identity
taus[1, 3:] = taus[1, 3:] * 1e-7                  # below the reference's small-angle threshold (1e-5)
taus[2, 3:] = taus[2, 3:] / taus[2, 3:].norm() * 3.1      # close to pi
out["pose_taus"] = taus.numpy()
out["pose_SE3_exp_f64"] = torch.stack([PU.SE3_exp(t) for t in taus]).numpy()
out["pose_SE3_exp_f32"] = torch.stack([PU.SE3_exp(t.float()) for t in taus]).numpy()
out["pose_SO3_exp_f64"] = torch.stack([PU.SO3_exp(t[3:]) for t in taus]).numpy()
out["pose_V_f64"] = torch.stack([PU.V(t[3:]) for t in taus]).numpy()

# ---- SSIM / L1 of the evaluation harness (gaussian_splatting/utils/loss_utils.py:21-22,61-101; eval_utils.py:147-150) -
# loss_utils imports cv2 at its top for l1_loss_weight (not installed: an EMPTY module stands in); ssim / l1_loss are torch
import gaussian_splatting.utils.loss_utils as LU  # noqa: E402
g = torch.Generator().manual_seed(9)
a = torch.rand(2, 3, 40, 52, generator=g)
b = (a + 0.1 * torch.randn(2, 3, 40, 52, generator=g)).clamp(0, 1)
out["ssim_a"], out["ssim_b"] = a.numpy(), b.numpy()
out["ssim_mean"] = LU.ssim(a, b).numpy()
out["ssim_per_image"] = LU.ssim(a, b, size_average=False).numpy()
out["ssim_single"] = LU.ssim(a[:1], b[:1]).numpy()
out["l1_loss"] = LU.l1_loss(a, b).numpy()

path = os.path.join(HERE, "map_update_ref.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
