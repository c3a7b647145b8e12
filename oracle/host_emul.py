"""ctypes binding of oracle/host_emul.cpp (TEST INFRASTRUCTURE ONLY; see its header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhost_emul.so")
_SRC = os.path.join(_HERE, "host_emul.cpp")
_HDR = os.path.join(_HERE, "..", "monogs_amd", "csrc", "raster_math.h")


def build(force: bool = False) -> str:
    stale = (not os.path.exists(_SO)
             or os.path.getmtime(_SO) < max(os.path.getmtime(_SRC), os.path.getmtime(_HDR)))
    if force or stale:
        subprocess.check_call(["g++", "-O2", "-fopenmp", "-shared", "-fPIC", "-std=c++17",
                               _SRC, "-o", _SO])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.emul_create.restype = C.c_void_p
        _lib.emul_num_threads.restype = C.c_int
        _lib.emul_destroy.argtypes = [C.c_void_p]
        _lib.emul_forward.restype = C.c_int64
    return _lib


def _p(t, typ=C.c_float):
    if t is None:
        return C.POINTER(typ)()
    assert t.is_contiguous()
    return C.cast(t.data_ptr(), C.POINTER(typ))


class HostEmul:
    """Forward/backward through the CPU emulation.  Tensors are float32 CPU."""

    def __init__(self):
        self.h = C.c_void_p(lib().emul_create())

    @staticmethod
    def num_threads() -> int:
        """OpenMP threads the emulation runs on (omp_get_max_threads)."""
        return int(lib().emul_num_threads())

    def __del__(self):
        try:
            lib().emul_destroy(self.h)
        except Exception:
            pass

    def forward(self, st, means3D, shs, colors_precomp, opacities, scales, rotations,
                cov3D_precomp, exact_cull=True):
        f = lambda t: None if t is None else t.detach().float().contiguous()
        self.N = means3D.shape[0]
        self.st = st
        self.inp = dict(means3D=f(means3D), shs=f(shs), col=f(colors_precomp),
                        op=f(opacities).reshape(-1).contiguous(), scales=f(scales),
                        rot=f(rotations), cov=f(cov3D_precomp))
        H, W = int(st.image_height), int(st.image_width)
        K = 0 if shs is None else shs.shape[1]
        self.K = K
        img = torch.empty(3, H, W)
        dep = torch.empty(1, H, W)
        opa = torch.empty(1, H, W)
        radii = torch.empty(self.N, dtype=torch.int32)
        nt = torch.empty(self.N, dtype=torch.int32)
        self.cam = [f(st.viewmatrix), f(st.projmatrix), f(st.projmatrix_raw),
                    f(st.campos).reshape(-1)[:3].contiguous(), f(st.bg)]
        i = self.inp
        pairs = lib().emul_forward(
            self.h, C.c_int(self.N), C.c_int(W), C.c_int(H), C.c_float(st.tanfovx),
            C.c_float(st.tanfovy), C.c_float(st.scale_modifier), C.c_int(int(st.sh_degree)),
            C.c_int(K), _p(self.cam[0]), _p(self.cam[1]), _p(self.cam[2]), _p(self.cam[3]),
            _p(self.cam[4]), _p(i["means3D"]), _p(i["scales"]), _p(i["rot"]), _p(i["cov"]),
            _p(i["op"]), _p(i["shs"]), _p(i["col"]), C.c_int(1 if exact_cull else 0),
            _p(img), _p(dep), _p(opa), _p(radii, C.c_int32), _p(nt, C.c_int32))
        self.pairs = int(pairs)
        return img, radii, dep, opa, nt

    def backward(self, grad_color, grad_depth):
        N, K, i = self.N, self.K, self.inp
        gc = grad_color.detach().float().contiguous()
        gd = None if grad_depth is None else grad_depth.detach().float().contiguous()
        out = dict(
            means3D=torch.empty(N, 3), means2D=torch.empty(N, 3),
            colors=torch.empty(N, K, 3) if i["shs"] is not None else torch.empty(N, 3),
            opacities=torch.empty(N), scales=torch.empty(N, 3), rotations=torch.empty(N, 4),
            cov3D=torch.empty(N, 6), tau=torch.empty(6))
        lib().emul_backward(
            self.h, _p(self.cam[4]), _p(i["means3D"]), _p(i["scales"]), _p(i["rot"]), _p(i["cov"]),
            _p(i["shs"]), _p(gc), _p(gd), _p(out["means3D"]), _p(out["means2D"]),
            _p(out["colors"]), _p(out["opacities"]),
            _p(out["scales"]) if i["scales"] is not None else _p(None),
            _p(out["rotations"]) if i["rot"] is not None else _p(None),
            _p(out["cov3D"]) if i["cov"] is not None else _p(None), _p(out["tau"]))
        return out
