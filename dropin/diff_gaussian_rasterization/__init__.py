"""Drop-in for the `diff_gaussian_rasterization` package MonoGS imports at
gaussian_splatting/gaussian_renderer/__init__.py:15-18 (reference tree).  Put
<repo>/dropin and <repo> on PYTHONPATH and MonoGS's files import unchanged; every
call lands in the HIP kernels behind include/monogs_raster.h."""
from monogs_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,
                                   rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
