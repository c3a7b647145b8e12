"""Golden vectors beyond SYN-A from THIS repository's CPU oracle (oracle/torch_raster.py, fp32),
SURVEY §8c items iii, v, vi, at sizes that cross the kernels' structural boundaries:

    python tests/golden/make_mid_golden.py      # -> tests/golden/{mid_crowded,sh3,wide_clamp,sketch_kat,knn_4800}.npz

  mid_crowded  33 000 Gaussians @ 320x240 with a tile of > 4096 splats and neighbours of > 1024
               (both LDS sort classes + the in-HBM sort, tens of checkpoint segments per tile)
  sh3          SH degree 3, moved camera, true camera centre
  wide_clamp   splats beyond 1.3x the field of view: gradients of BOTH clamp treatments
  sketch_kat   sketched pose Jacobian, stack 4 / sketch 8 (the reference's check_grad construction,
               utils/slam_frontend.py:1031-1127, with the oracle as the right-hand side)
  knn_4800     distCUDA2 on a keyframe-sized point set

Inputs are re-derived from seeds by tests/scenes.py; only oracle outputs are stored.  They are
NOT reference outputs: the reference's rasteriser source is absent; they make the
full-path GPU evidence independent of csrc/raster_math.h.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from conftest import oracle_settings  # noqa: E402
from monogs_amd import synthetic as S  # noqa: E402
from oracle import torch_raster as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def run(sc, shs=None, deg=0, campos=None, clamp_grad="exact"):
    m, s, r, o, sh0 = S.activated(sc)
    sh = sh0 if shs is None else shs
    L = [t.clone().requires_grad_() for t in (m, s, r, o, sh)]
    theta = torch.zeros(3, requires_grad=True)
    rho = torch.zeros(3, requires_grad=True)
    m2d = torch.zeros(m.shape[0], 3, requires_grad=True)
    st = oracle_settings(sc.cam, sc.bg, deg=deg, campos=campos)
    img, radii, dep, opa, nt, info = O.rasterize(L[0], m2d, L[4], None, L[3], L[1], L[2], None, st, theta, rho,
                                                 clamp_grad=clamp_grad)
    S.synthetic_loss(img, dep, sc).backward()
    return {"image": img.detach().numpy(), "depth": dep.detach().numpy(), "opacity": opa.detach().numpy(),
            "radii": radii.numpy().astype(np.int32), "n_touched": nt.numpy().astype(np.int32),
            "grad_means3D": L[0].grad.numpy(), "grad_scales": L[1].grad.numpy(), "grad_rot": L[2].grad.numpy(),
            "grad_opacity": L[3].grad.numpy(), "grad_sh": L[4].grad.numpy(), "grad_means2D": m2d.grad.numpy(),
            "grad_tau": torch.cat([rho.grad, theta.grad]).numpy(), "pairs": np.int64(info["pairs"])}, info


def thin(d, keep_full=("image", "depth", "opacity", "radii", "n_touched", "grad_tau", "pairs", "grad_means3D",
                       "grad_opacity"), step=4):
    """Per-Gaussian gradient arrays other than means3D / opacity are stored for every `step`-th
    Gaussian (plus their full norms), which keeps the fixture at a few MB."""
    out = {}
    for k, v in d.items():
        if k in keep_full or np.ndim(v) == 0:
            out[k] = v
        else:
            out[k + "_thin"] = v[::step].copy()
            out[k + "_norm"] = np.float64(np.linalg.norm(v.astype(np.float64)))
    out["thin_step"] = np.int64(step)
    return out


def main():
    t0 = time.time()
    sc = scenes.crowded_scene()
    d, info = run(sc)
    proj = info["proj"]
    # tile occupancy of the bounding-square binning, to document which size classes are hit
    vis = proj.radii > 0
    gx = (sc.cam.W + 15) // 16
    counts = {}
    rmin, rmax = proj.rect_min[vis], proj.rect_max[vis]
    occ = torch.zeros((sc.cam.H + 15) // 16, gx, dtype=torch.int64)
    for a, b in zip(rmin.tolist(), rmax.tolist()):
        occ[a[1]:b[1], a[0]:b[0]] += 1
    d["tile_occupancy_max"] = np.int64(int(occ.max()))
    d["tiles_over_1024"] = np.int64(int((occ > 1024).sum()))
    d["tiles_over_4096"] = np.int64(int((occ > 4096).sum()))
    np.savez_compressed(os.path.join(OUT, "mid_crowded.npz"), **thin(d))
    print("mid_crowded", int(occ.max()), int((occ > 1024).sum()), int((occ > 4096).sum()), f"{time.time() - t0:.1f}s")

    sc3, shs, campos = scenes.sh3_inputs()
    d, _ = run(sc3, shs=shs, deg=3, campos=campos)
    np.savez_compressed(os.path.join(OUT, "sh3.npz"), **thin(d, step=2))

    scw = scenes.wide_scene()
    ex, info = run(scw, clamp_grad="exact")
    up, _ = run(scw, clamp_grad="upstream")
    V = scw.cam.viewmatrix
    pv = (torch.cat([scw.means3D, torch.ones(len(scw.means3D), 1)], 1) @ V)[:, :3]
    cl = ((pv[:, 0] / pv[:, 2]).abs() > 1.3 * scw.cam.tanfovx) | ((pv[:, 1] / pv[:, 2]).abs() > 1.3 * scw.cam.tanfovy)
    out = {"image": ex["image"], "depth": ex["depth"], "radii": ex["radii"],
           "clamped_visible": np.int64(int((cl & (torch.from_numpy(ex["radii"]) > 0)).sum()))}
    for k in ("grad_means3D", "grad_scales", "grad_rot", "grad_opacity", "grad_sh", "grad_tau"):
        out[k + "_exact"] = ex[k]
        out[k + "_upstream"] = up[k]
    np.savez_compressed(os.path.join(OUT, "wide_clamp.npz"), **out)

    sck, A, B, fsa = scenes.sketch_kat_setup()
    m, s, r, o, sh = S.activated(sck)
    L = [t.clone().requires_grad_() for t in (m, s, r, o, sh)]
    th = torch.zeros(3, requires_grad=True)
    rh = torch.zeros(3, requires_grad=True)
    oimg, _, odep, _, _, _ = O.rasterize(L[0], None, L[4], None, L[3], L[1], L[2], None,
                                         oracle_settings(sck.cam, sck.bg), th, rh)
    res = (oimg * A).sum(0) + (odep * B)[0]
    ow = res * fsa["rand_weights"][0]
    idx = fsa["sketch_indices"][0]
    stack, sketch = idx.shape[0], int(fsa["sketch_dim"])
    SJ = torch.zeros(stack, sketch, 6)
    for st_ in range(stack):
        for k in range(sketch):
            th.grad = None
            rh.grad = None
            ow[idx[st_] == k].sum().backward(retain_graph=True)
            SJ[st_, k] = torch.cat([rh.grad, th.grad])
    np.savez_compressed(os.path.join(OUT, "sketch_kat.npz"), SJ=SJ.numpy(), image=oimg.detach().numpy())

    pts = scenes.knn_points()
    np.savez_compressed(os.path.join(OUT, "knn_4800.npz"), dist2=O.dist2_knn3(pts).numpy())
    for f in ("mid_crowded", "sh3", "wide_clamp", "sketch_kat", "knn_4800"):
        print(f, os.path.getsize(os.path.join(OUT, f + ".npz")), "bytes")
    print(f"total {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()
