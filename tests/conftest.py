import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    import __graft_entry__ as g
    g.build()
    return True


def oracle_settings(cam, bg, deg=0, campos=None, dtype=torch.float32, scale_modifier=1.0):
    from oracle import torch_raster as O
    cp = cam.viewmatrix if campos is None else campos
    return O.RasterSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, bg.to(dtype), scale_modifier,
                            cam.viewmatrix.to(dtype), cam.projmatrix.to(dtype),
                            cam.projmatrix_raw.to(dtype), deg, cp.to(dtype), False, False)


def gpu_settings(cam, bg, dev, deg=0, campos=None, scale_modifier=1.0):
    from monogs_amd.rasterizer import GaussianRasterizationSettings
    cp = cam.viewmatrix if campos is None else campos
    return GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, bg.to(dev),
                                         scale_modifier, cam.viewmatrix.to(dev),
                                         cam.projmatrix.to(dev), cam.projmatrix_raw.to(dev), deg,
                                         cp.to(dev), False, False)


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
