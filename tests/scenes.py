"""Deterministic scenes shared by the golden-vector generators (tests/golden/make_*.py) and the
tests that check against the committed vectors.  Inputs are re-derived from seeds; only oracle
OUTPUTS are stored under tests/golden/."""
import math

import torch

from monogs_amd import synthetic as S


def crowded_scene(W=320, H=240, n_base=24000, n_cluster=9000, seed=7):
    """SYN-like scene plus a dense, semi-transparent cluster inside one 16x16 tile and its
    neighbours: tile lists of > 1024 and > 4096 splats (both sort size classes and the in-HBM
    sort), dozens of 64-splat segments per tile (checkpoints), saturating and never-saturating
    quadrants side by side."""
    sc = S.make_scene(n_base, W, H, seed=seed)
    cam = sc.cam
    g = torch.Generator().manual_seed(seed + 1)
    u = 100.0 + 14.0 * torch.rand(n_cluster, generator=g)            # inside tile (6, 5) mostly
    v = 84.0 + 10.0 * torch.rand(n_cluster, generator=g)
    z = 1.0 + 4.0 * torch.rand(n_cluster, generator=g)
    xyz = torch.stack([(u - cam.cx) * z / cam.fx, (v - cam.cy) * z / cam.fy, z], dim=1)
    sig = torch.exp(math.log(1.2) + 0.4 * torch.randn(n_cluster, generator=g))
    ls = torch.log((sig * z / cam.fx)[:, None] * torch.exp(0.3 * torch.randn(n_cluster, 3, generator=g)))
    rot = torch.nn.functional.normalize(torch.randn(n_cluster, 4, generator=g))
    op = -3.2 + 0.8 * torch.randn(n_cluster, 1, generator=g)        # mostly 2-8 % opaque: long live lists
    fdc = ((torch.rand(n_cluster, 3, generator=g) - 0.5) / S.SH_C0)[:, None, :]
    return sc._replace(means3D=torch.cat([sc.means3D, xyz]).contiguous(),
                       log_scales=torch.cat([sc.log_scales, ls]).contiguous(),
                       rot=torch.cat([sc.rot, rot]).contiguous(),
                       opacity_logit=torch.cat([sc.opacity_logit, op]).contiguous(),
                       features_dc=torch.cat([sc.features_dc, fdc]).contiguous())


def wide_scene(N=2000, W=160, H=120, seed=5):
    """Large splats centred up to 70 % of the image size OUTSIDE the image: several hundred visible
    splats beyond 1.3x the field of view, where the EWA clamp is active and the two backward
    treatments (oracle `clamp_grad`, mgs_backward_args.clamp_gradient_mode) differ."""
    g = torch.Generator().manual_seed(seed)
    sc = S.make_scene(N, W, H, seed=seed)
    cam = sc.cam
    u = (torch.rand(N, generator=g) * 2.4 - 0.7) * W
    v = (torch.rand(N, generator=g) * 2.4 - 0.7) * H
    z = 0.5 + 5.5 * torch.rand(N, generator=g)
    xyz = torch.stack([(u - cam.cx) * z / cam.fx, (v - cam.cy) * z / cam.fy, z], dim=1)
    sig = torch.exp(math.log(14.0) + 0.5 * torch.randn(N, generator=g))
    ls = torch.log((sig * z / cam.fx)[:, None] * torch.exp(0.3 * torch.randn(N, 3, generator=g)))
    return sc._replace(means3D=xyz.contiguous(), log_scales=ls.contiguous())


def sh3_inputs(N=3000, W=160, H=120, seed=9):
    """Degree-3 spherical harmonics (K = 16), a moved camera and a true camera centre."""
    from monogs_amd.pose import SE3_exp
    sc = S.make_scene(N, W, H, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    T = SE3_exp(torch.tensor([0.05, -0.03, 0.04, 0.02, -0.015, 0.01]))
    cam = S.make_camera(W, H, T)
    rest = 0.15 * torch.randn(N, 15, 3, generator=g)
    shs = torch.cat([sc.features_dc, rest], dim=1).contiguous()
    campos = torch.linalg.inv(T)[:3, 3].contiguous()
    return sc._replace(cam=cam), shs, campos


def knn_points(P=4800, seed=11):
    """Keyframe-sized point set (640*480/64 points): a noisy depth sheet."""
    g = torch.Generator().manual_seed(seed)
    uv = torch.rand(P, 2, generator=g)
    z = 1.5 + 0.8 * torch.sin(4 * uv[:, 0]) * torch.cos(3 * uv[:, 1]) + 0.02 * torch.randn(P, generator=g)
    return torch.stack([(uv[:, 0] - 0.5) * z, (uv[:, 1] - 0.5) * 0.75 * z, z], dim=1).contiguous()


def sketch_kat_setup(N=2000, W=160, H=120, stack=4, sketch=8, seed=13):
    """Inputs of the sketch known-answer test (SURVEY §8c v): SYN-A-shaped scene, linear pixel
    functional (A, B), one random bucket partition per stack."""
    from monogs_amd.slam_loops import gen_forward_sketch_args
    sc = S.make_scene(N, W, H, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    A = torch.randn(3, H, W, generator=g)
    B = torch.randn(1, H, W, generator=g)
    fsa = gen_forward_sketch_args(H, W, 1, stack, sketch, "cpu", generator=g)
    return sc, A, B, fsa
