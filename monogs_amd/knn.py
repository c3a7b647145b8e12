"""Host-side mirror of `simple_knn._C.distCUDA2` (reference call site:
gaussian_splatting/scene/gaussian_model.py:185-191)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _cabi


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 runs on the GPU only (HIP kernel); there is no CPU fallback")
    lib = _cabi.lib()
    pts = points.detach()
    if pts.dtype != torch.float32:
        pts = pts.float()
    pts = pts.contiguous()
    P = int(pts.shape[0])
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    scratch = torch.empty(int(lib.mgs_knn_scratch_bytes(P)), dtype=torch.uint8, device=pts.device)
    stream = C.c_void_p(torch.cuda.current_stream(pts.device).cuda_stream)
    _cabi.check(lib.mgs_knn_dist2(pts.data_ptr(), P, out.data_ptr(), scratch.data_ptr(), stream),
                "mgs_knn_dist2")
    return out
