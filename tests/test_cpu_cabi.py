"""CPU tests of the C-ABI boundary: the library builds, loads and exports every symbol
include/monogs_raster.h declares; host-only entry points work without a GPU; the Python
mirror refuses to run without a GPU instead of falling back."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    hdr = open(os.path.join(ROOT, "include", "monogs_raster.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mgs_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(built):
    from monogs_amd import _cabi
    lib = C.CDLL(_cabi.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/monogs_raster.h but not exported"
    assert set(_cabi.EXPORTS) == set(names)
    assert _cabi.lib().mgs_abi_version() == _cabi.ABI_VERSION


def test_struct_layout_matches_header(built):
    """ctypes mirrors must have the C sizes (LP64: 9*4 bytes shape padded to 40)."""
    from monogs_amd import _cabi
    assert C.sizeof(_cabi.RasterShape) == 36
    assert C.sizeof(_cabi.ForwardArgs) == 40 + 21 * 8 + 8      # 21 pointers + big_tile_pass + reserved
    assert C.sizeof(_cabi.WorkspaceSizes) == 12 * 8
    # every ctypes mirror against the compiler's sizeof (mgs_struct_size)
    lib = _cabi.lib()
    for i, cls in enumerate(_cabi.struct_mirrors()):
        assert C.sizeof(cls) == lib.mgs_struct_size(i), cls.__name__
    assert lib.mgs_struct_size(len(_cabi.struct_mirrors())) == -1


def test_workspace_query_and_status_strings(built):
    from monogs_amd import _cabi
    lib = _cabi.lib()
    sh = _cabi.RasterShape(300000, 640, 480, 0, 1, 1_000_000, 0.6, 0.45, 1.0)
    sz = _cabi.workspace_sizes(sh)
    assert sz.geom_bytes >= 300000 * 48 + 640 * 480 * 8
    assert sz.bins_bytes >= 1_000_000 * 12
    assert sz.bwd_bytes >= 1_000_000 * 40          # ten raw pixel sums per pair, 40-B records
    assert sz.off_records % 256 == 0 and sz.off_keys % 256 == 0
    bad = _cabi.RasterShape(0, 640, 480, 0, 1, 0, 0.6, 0.45, 1.0)
    out = _cabi.WorkspaceSizes()
    assert lib.mgs_raster_workspace_query(C.byref(bad), C.byref(out)) == -1
    assert b"bad argument" in lib.mgs_status_string(-1)
    assert lib.mgs_status_string(0) == b"ok"
    with pytest.raises(RuntimeError):
        _cabi.check(-1, "x")


def test_null_arguments_are_rejected_without_touching_the_gpu(built):
    from monogs_amd import _cabi
    lib = _cabi.lib()
    a = _cabi.ForwardArgs()
    a.shape = _cabi.RasterShape(10, 32, 32, 0, 1, 0, 0.6, 0.45, 1.0)
    assert lib.mgs_raster_forward_project(C.byref(a), None) == -1
    assert lib.mgs_raster_forward_project(None, None) == -1
    b = _cabi.BackwardArgs()
    assert lib.mgs_raster_backward(C.byref(b), None) == -1
    assert lib.mgs_knn_dist2(None, 5, None, None, None) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_python_mirror_has_no_cpu_fallback(built):
    from conftest import gpu_settings
    from monogs_amd import synthetic as S
    from monogs_amd.knn import distCUDA2
    from monogs_amd.rasterizer import GaussianRasterizer
    sc = S.make_scene(8, 32, 32)
    m, s, r, o, sh = S.activated(sc)
    ras = GaussianRasterizer(gpu_settings(sc.cam, sc.bg, "cpu"))
    with pytest.raises(RuntimeError, match="GPU only"):
        ras(means3D=m, means2D=torch.zeros(8, 3), opacities=o, shs=sh, scales=s, rotations=r)
    with pytest.raises(RuntimeError):
        distCUDA2(torch.rand(10, 3))


def test_dropin_module_names(built):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        import diff_gaussian_rasterization as d
        from simple_knn._C import distCUDA2  # noqa: F401
        st = d.GaussianRasterizationSettings._fields
        assert st == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier",
                      "viewmatrix", "projmatrix", "projmatrix_raw", "sh_degree", "campos",
                      "prefiltered", "debug")
        assert callable(d.GaussianRasterizer)
    finally:
        sys.path.pop(0)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under monogs_amd/ or dropin/ may import it."""
    for base in ("monogs_amd", "dropin"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), (dp, f)
