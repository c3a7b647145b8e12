"""PyTorch restatement of the reference's map maintenance (TEST INFRASTRUCTURE ONLY), used
to check monogs_amd/map_update.py:
  * densify_and_prune / densify_and_clone / densify_and_split / prune_points /
    densification_postfix / cat_tensors_to_optimizer / _prune_optimizer of
    /root/reference/gaussian_splatting/scene/gaussian_model.py:485-691
  * reset_opacity / reset_opacity_nonvisible / replace_tensor_to_optimizer (:364-377, :470-483),
    add_densification_stats (:693-697)
written against a plain dict-of-tensors state instead of nn.Parameters + torch.optim state,
so results can be compared tensor by tensor.  The split's random draw (:609) is an argument.

PINNED since round 5: tests/golden/map_update_ref.npz holds what the reference's own GaussianModel returned
on seeded inputs (tests/golden/make_map_update_golden.py runs it on the CPU in the build container);
tests/test_cpu_map_update_golden.py holds every function below to those arrays, and the GPU tests check the
HIP path against the SAME arrays.
"""
from __future__ import annotations

import torch

PARAMS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def rotation_matrix(q):
    """general_utils.py:114-137 (normalises the quaternion first)."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def _select(state, keep):
    """prune_points / _prune_optimizer (:506-556) with keep = ~mask."""
    out = {k: v[keep] for k, v in state.items()}
    return out


def _append(state, new):
    """densification_postfix / cat_tensors_to_optimizer (:558-596): parameters are
    concatenated, Adam moments extended with zeros, statistics restart from zero."""
    out = {}
    for name in PARAMS:
        out[name] = torch.cat((state[name], new[name]), 0)
        for mom in ("exp_avg_", "exp_avg_sq_"):
            out[mom + name] = torch.cat((state[mom + name], torch.zeros_like(new[name])), 0)
    out["kf"] = torch.cat((state["kf"], new["kf"]))
    out["n_obs"] = torch.cat((state["n_obs"], new["n_obs"]))
    n = out["xyz"].shape[0]
    out["grad_accum"] = torch.zeros(n, 1)
    out["denom"] = torch.zeros(n, 1)
    out["max_radii"] = torch.zeros(n)
    return out


def densify_and_prune(state, max_grad, min_opacity, extent, max_screen_size, percent_dense, unit_noise):
    """:674-691.  `unit_noise` [2*n_split, 3]: standard normals that the split scales by the
    parents' activated scales (torch.normal(mean=0, std=stds) at :608-609)."""
    grads = state["grad_accum"] / state["denom"]
    grads[grads.isnan()] = 0.0
    scale = torch.exp(state["scaling"])
    # densify_and_clone (:636-672)
    sel = (grads.norm(dim=-1) >= max_grad) & (scale.max(dim=1).values <= percent_dense * extent)
    new = {k: state[k][sel] for k in PARAMS}
    new["kf"], new["n_obs"] = state["kf"][sel], state["n_obs"][sel]
    state = _append(state, new)
    # densify_and_split (:598-634) on the extended set; clones carry gradient 0
    n_now = state["xyz"].shape[0]
    padded = torch.zeros(n_now)
    padded[:grads.shape[0]] = grads.squeeze(-1)
    scale = torch.exp(state["scaling"])
    sel = (padded >= max_grad) & (scale.max(dim=1).values > percent_dense * extent)
    N = 2
    stds = scale[sel].repeat(N, 1)
    samples = unit_noise * stds
    rots = rotation_matrix(state["rotation"][sel]).repeat(N, 1, 1)
    new = {
        "xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + state["xyz"][sel].repeat(N, 1),
        "scaling": torch.log(scale[sel].repeat(N, 1) / (0.8 * N)),
        "rotation": state["rotation"][sel].repeat(N, 1),
        "f_dc": state["f_dc"][sel].repeat(N, 1, 1),
        "f_rest": state["f_rest"][sel].repeat(N, 1, 1),
        "opacity": state["opacity"][sel].repeat(N, 1),
        "kf": state["kf"][sel].repeat(N), "n_obs": state["n_obs"][sel].repeat(N),
    }
    n_sel = int(sel.sum())
    state = _append(state, new)
    drop = torch.cat((sel, torch.zeros(N * n_sel, dtype=torch.bool)))
    state = _select(state, ~drop)
    # final prune (:680-691)
    prune = (torch.sigmoid(state["opacity"]) < min_opacity).squeeze(-1)
    if max_screen_size:
        big_vs = state["max_radii"] > max_screen_size
        big_ws = torch.exp(state["scaling"]).max(dim=1).values > 0.1 * extent
        prune = prune | big_vs | big_ws
    return _select(state, ~prune)


def prune_points(state, mask):
    return _select(state, ~mask)


def inverse_sigmoid(x):
    """general_utils.py:18-19."""
    return torch.log(x / (1 - x))


def reset_opacity(state):
    """:364-367 + replace_tensor_to_optimizer (:470-483): every logit <- inverse_sigmoid(0.01), the opacity
    group's Adam moments zeroed; nothing else is touched."""
    out = dict(state)
    out["opacity"] = inverse_sigmoid(torch.ones_like(state["opacity"]) * 0.01)
    out["exp_avg_opacity"] = torch.zeros_like(state["opacity"])
    out["exp_avg_sq_opacity"] = torch.zeros_like(state["opacity"])
    return out


def reset_opacity_nonvisible(state, visibility_filters):
    """:369-377.  Gaussians outside every filter get the logit of 0.4.  A reference quirk that this
    restatement keeps (the fixture pins it): for a Gaussian INSIDE a filter the reference stores
    `self.get_opacity[filter]` - the ACTIVATED opacity sigmoid(logit) - as the new raw parameter (:375), so a
    visible Gaussian's logit l becomes sigmoid(l).  The opacity group's Adam moments are zeroed for all."""
    out = dict(state)
    new = inverse_sigmoid(torch.ones_like(state["opacity"]) * 0.4)
    act = torch.sigmoid(state["opacity"])
    for f in visibility_filters:
        new[f] = act[f]
    out["opacity"] = new
    out["exp_avg_opacity"] = torch.zeros_like(new)
    out["exp_avg_sq_opacity"] = torch.zeros_like(new)
    return out


def add_densification_stats(grad_accum, denom, viewspace_grad, update_filter):
    """:693-697: the norm of the first two columns of the screen-space gradient is added, and the counter
    incremented, where the filter holds."""
    ga, dn = grad_accum.clone(), denom.clone()
    ga[update_filter] += torch.norm(viewspace_grad[update_filter, :2], dim=-1, keepdim=True)
    dn[update_filter] += 1
    return ga, dn


def extend_from_pcd(state, xyz, features, scales, rots, opacities, kf_id):
    """:210-236: a keyframe's new Gaussians appended through densification_postfix - `features` arrives as
    [P, 3, K] (create_pcd_from_image, :199-205) and is split into f_dc [P, 1, 3] / f_rest [P, K - 1, 3]; the new rows
    carry keyframe id `kf_id`, zero observation counts and zero Adam moments; the statistics restart for ALL rows."""
    new = {"xyz": xyz, "f_dc": features[:, :, 0:1].transpose(1, 2).contiguous(),
           "f_rest": features[:, :, 1:].transpose(1, 2).contiguous(), "opacity": opacities, "scaling": scales,
           "rotation": rots, "kf": torch.ones(xyz.shape[0]).int() * kf_id, "n_obs": torch.zeros(xyz.shape[0]).int()}
    return _append(state, new)
