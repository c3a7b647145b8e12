"""Thinned oracle vectors at the FULL BASELINE sizes (SURVEY §8c item iii) from THIS repository's
CPU oracle (oracle/torch_raster.py, fp32, per-tile autograd):

    python tests/golden/make_full_golden.py [synb] [sync]   # -> tests/golden/{syn_b,syn_c}_oracle.npz

  syn_b   BASELINE config 2: 100 000 Gaussians @ 640x480, forward only
  syn_c   BASELINE config 3: 300 000 Gaussians @ 640x480, forward + every gradient sink incl. dL/dtau

Inputs are `monogs_amd.synthetic.make_scene(N, 640, 480, seed=0)` (the scene bench.py times);
only oracle outputs are stored, thinned so that a fixture stays at a few MB:
images at every 2nd row / column (+ the full-image mean and L1 mass), per-Gaussian arrays for every
`G_STEP`-th Gaussian (+ full norms / sums).  They are NOT reference outputs (the reference's
rasteriser source is absent); they make the evidence for configs 2 and 3 independent of
csrc/raster_math.h, which the host-emulation test shares with the kernels.
Container time: a few minutes and < 30 GB for syn_c.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import oracle_settings  # noqa: E402
from monogs_amd import synthetic as S  # noqa: E402
from oracle import torch_raster as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PIX_STEP, G_STEP = 2, 16


def thin_image(name, a, out):
    out[name + "_sub"] = a[:, ::PIX_STEP, ::PIX_STEP].copy()
    out[name + "_mean"] = np.float64(a.astype(np.float64).mean())
    out[name + "_l1"] = np.float64(np.abs(a.astype(np.float64)).sum())


def thin_gauss(name, a, out):
    out[name + "_thin"] = a[::G_STEP].copy()
    if a.dtype.kind == "f":
        out[name + "_norm"] = np.float64(np.linalg.norm(a.astype(np.float64)))
    else:
        out[name + "_sum"] = np.int64(a.astype(np.int64).sum())


def run(N, backward):
    sc = S.make_scene(N, 640, 480, seed=0)
    m, s, r, o, sh = S.activated(sc)
    L = [t.clone().requires_grad_(backward) for t in (m, s, r, o, sh)]
    theta = torch.zeros(3, requires_grad=backward)
    rho = torch.zeros(3, requires_grad=backward)
    m2d = torch.zeros(N, 3, requires_grad=backward)
    st = oracle_settings(sc.cam, sc.bg)
    out = {"pix_step": np.int64(PIX_STEP), "g_step": np.int64(G_STEP)}
    with torch.set_grad_enabled(backward):
        img, radii, dep, opa, nt, info = O.rasterize(L[0], m2d, L[4], None, L[3], L[1], L[2], None, st, theta, rho)
    thin_image("image", img.detach().numpy(), out)
    thin_image("depth", dep.detach().numpy(), out)
    thin_image("opacity", opa.detach().numpy(), out)
    thin_gauss("radii", radii.numpy().astype(np.int32), out)
    thin_gauss("n_touched", nt.numpy().astype(np.int32), out)
    out["pairs"] = np.int64(info["pairs"])
    out["n_visible"] = np.int64(info["n_visible"])
    if backward:
        loss = S.synthetic_loss(img, dep, sc)
        loss.backward()
        out["loss"] = np.float64(loss.item())
        for k, v in (("grad_means3D", L[0].grad), ("grad_scales", L[1].grad), ("grad_rot", L[2].grad),
                     ("grad_opacity", L[3].grad), ("grad_sh", L[4].grad), ("grad_means2D", m2d.grad)):
            thin_gauss(k, v.numpy(), out)
        out["grad_tau"] = torch.cat([rho.grad, theta.grad]).numpy()
    return out


if __name__ == "__main__":
    which = sys.argv[1:] or ["synb", "sync"]
    for name, N, bwd, f in (("synb", 100000, False, "syn_b_oracle.npz"), ("sync", 300000, True, "syn_c_oracle.npz")):
        if name not in which:
            continue
        t0 = time.time()
        out = run(N, bwd)
        path = os.path.join(OUT, f)
        np.savez_compressed(path, **out)
        print(f, int(out["pairs"]), int(out["n_visible"]), os.path.getsize(path), "bytes", f"{time.time() - t0:.0f}s",
              flush=True)
