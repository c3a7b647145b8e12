"""Golden vectors of the SYN-A workload (BASELINE config 1 shape: 5 000 Gaussians @ 160x120,
SURVEY §8c item iii / iv) from THIS repository's CPU oracle (oracle/torch_raster.py, fp32):

    python tests/golden/make_syn_golden.py        # -> tests/golden/syn_a_oracle.npz

They freeze the oracle (tests/test_cpu_oracle.py re-derives them on the CPU) and give the HIP
path a committed fixture to be checked against on the GPU box (tests/test_raster_gpu.py).
They are NOT reference outputs: the reference's rasteriser source is absent.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import oracle_settings  # noqa: E402
from monogs_amd import synthetic as S  # noqa: E402
from oracle import torch_raster as O  # noqa: E402


def compute():
    sc = S.make_scene(5000, 160, 120, seed=0)
    m, s, r, o, sh = S.activated(sc)
    L = [t.clone().requires_grad_() for t in (m, s, r, o, sh)]
    theta = torch.zeros(3, requires_grad=True)
    rho = torch.zeros(3, requires_grad=True)
    st = oracle_settings(sc.cam, sc.bg)
    img, radii, dep, opa, nt, _ = O.rasterize(L[0], None, L[4], None, L[3], L[1], L[2], None, st, theta, rho)
    S.synthetic_loss(img, dep, sc).backward()
    return {
        "image": img.detach().numpy(), "depth": dep.detach().numpy(), "opacity": opa.detach().numpy(),
        "radii": radii.numpy().astype(np.int32), "n_touched": nt.numpy().astype(np.int32),
        "grad_means3D": L[0].grad.numpy(), "grad_scales": L[1].grad.numpy(), "grad_rot": L[2].grad.numpy(),
        "grad_opacity": L[3].grad.numpy(), "grad_sh": L[4].grad.numpy(),
        "grad_tau": torch.cat([rho.grad, theta.grad]).numpy(),
    }


if __name__ == "__main__":
    out = compute()
    path = os.path.join(ROOT, "tests", "golden", "syn_a_oracle.npz")
    np.savez_compressed(path, **out)
    print(path, {k: v.shape for k, v in out.items()}, os.path.getsize(path), "bytes")
