"""ctypes binding of the C ABI in include/monogs_raster.h.

The shared library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950) at
monogs_amd/lib/libmonogs_raster.so.  There is NO fallback: if the library is missing
or an entry point is absent, importing `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MGS_LIB_PATH: load another build of the SAME library (kernel experiments, the -DMGS_STAMP build)
LIB_PATH = os.environ.get("MGS_LIB_PATH") or os.path.join(_HERE, "lib", "libmonogs_raster.so")

ABI_VERSION = 8

EXPORTS = (
    "mgs_abi_version", "mgs_struct_size", "mgs_status_string", "mgs_raster_workspace_query",
    "mgs_raster_forward_project", "mgs_raster_forward_blend", "mgs_raster_backward",
    "mgs_knn_scratch_bytes", "mgs_knn_dist2", "mgs_profile_enable", "mgs_profile_read",
    "mgs_pose_adam_step", "mgs_tracking_loss_partial_count", "mgs_tracking_loss_forward",
    "mgs_tracking_loss_backward", "mgs_tracking_loss_fused", "mgs_tracking_loss_onepass", "mgs_lm_solve_step", "mgs_mapping_loss_forward",
    "mgs_mapping_loss_backward", "mgs_camera_from_pose", "mgs_tracking_iteration",
    "mgs_adam_step_multi", "mgs_map_plan_blocks", "mgs_map_plan_count", "mgs_map_plan_emit",
    "mgs_map_gather", "mgs_pack_mapping_grads", "mgs_sketch_assign", "mgs_sketch_residual",
    "mgs_tracking_iteration_second_order", "mgs_map_activate", "mgs_mapping_loss_partial_count",
    "mgs_mapping_loss_fused", "mgs_mapping_view_iteration", "mgs_map_finish_iteration", "mgs_map_append",
)

_fp = C.c_void_p  # device pointers travel as plain addresses


class RasterShape(C.Structure):
    _fields_ = [("num_gaussians", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
                ("pair_capacity", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float),
                ("scale_modifier", C.c_float)]


class WorkspaceSizes(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "geom_bytes", "bins_bytes", "bwd_bytes", "sketch_bytes", "off_records",
        "off_pair_count", "off_tile_offset", "off_final_T", "off_n_contrib", "off_counters",
        "off_keys", "off_payload")]


class ForwardArgs(C.Structure):
    _fields_ = [("shape", RasterShape)] + [(n, _fp) for n in (
        "means3D", "scales", "rotations", "cov3D_precomp", "opacities", "shs", "colors_precomp",
        "viewmatrix", "projmatrix", "projmatrix_raw", "campos", "bg", "geom", "bins",
        "out_color", "out_depth", "out_opacity", "radii", "n_touched", "pair_count_out",
        "pair_count_max")] + [("big_tile_pass", C.c_int32), ("reserved0", C.c_int32)]


class MapAccumArgs(C.Structure):
    _fields_ = ([("scale_dims", C.c_int32), ("accumulate", C.c_int32), ("add_regulariser", C.c_int32),
                 ("regulariser_weight", C.c_float)]
                + [(n, _fp) for n in ("raw_rotations", "grad_xyz", "grad_features_dc", "grad_features_rest",
                                      "grad_opacity", "grad_scaling", "grad_rotation", "gradnorm_inc",
                                      "denom_inc", "radii_max", "visibility")])


class BackwardArgs(C.Structure):
    _fields_ = ([("fwd", ForwardArgs)] + [(n, _fp) for n in (
        "grad_color", "grad_depth", "bwd", "grad_means3D", "grad_means2D", "grad_colors",
        "grad_opacities", "grad_scales", "grad_rotations", "grad_cov3D", "grad_tau")]
        + [("sketch_mode", C.c_int32), ("sketch_dim", C.c_int32), ("stack_dim", C.c_int32),
           ("sketch_indices", _fp), ("grad_sketch_dtau", _fp), ("sketch_ws", _fp),
           ("sketch_bucket_flat", _fp), ("map_accum", C.POINTER(MapAccumArgs)),
           ("clamp_gradient_mode", C.c_int32), ("pair_count_bound", C.c_int32)])


class PoseAdamArgs(C.Structure):
    _fields_ = ([(n, _fp) for n in (
        "cam_rot_delta", "cam_trans_delta", "exposure_a", "exposure_b", "grad_rot", "grad_trans",
        "grad_a", "grad_b", "exp_avg", "exp_avg_sq", "T", "converged")]
        + [("step", C.c_int32)] + [(n, C.c_float) for n in (
            "lr_rot", "lr_trans", "lr_a", "lr_b", "beta1", "beta2", "eps", "converged_threshold")]
        + [("tau_partials", _fp), ("num_tau_partials", C.c_int32), ("exposure_partials", _fp),
           ("num_exposure_partials", C.c_int32), ("projection", _fp), ("viewmatrix_out", _fp),
           ("projmatrix_out", _fp), ("no_pose_update", C.c_int32), ("loss_partials", _fp),
           ("num_loss_partials", C.c_int32), ("loss_w_rgb", C.c_float), ("loss_w_depth", C.c_float),
           ("loss_view", _fp), ("loss_accum", _fp), ("loss_grad_out", _fp),
           ("loss_norm_mode", C.c_int32), ("sticky_converged", C.c_int32), ("l1_partials", _fp),
           ("best", _fp), ("num_l1_partials", C.c_int32), ("loss_pnorm", C.c_float)])


TRACK_BEST_FLOATS = 24     # MGS_TRACK_BEST_FLOATS: {best L1, T[16], a, b, index of the best iteration, counter,
                           #  L1 of the last iteration's render, |step| of the last iteration, spare}


class MappingLossArgs(C.Structure):
    _fields_ = ([(n, _fp) for n in ("image", "gt", "mask", "depth", "gt_depth", "exposure_a", "exposure_b")]
                + [("exposure_eps", C.c_float), ("w_rgb", C.c_float), ("w_depth", C.c_float),
                   ("depth_mask_threshold", C.c_float), ("apply_exposure", C.c_int32),
                   ("num_pixels", C.c_int64)]
                + [(n, _fp) for n in ("partial", "loss", "grad_out", "grad_image", "grad_depth",
                                      "grad_a", "grad_b")]
                + [("partial_ticket_ready", C.c_int32), ("reserved0", C.c_int32)])


class LMStepArgs(C.Structure):
    _fields_ = [("SJ", _fp), ("Sf", _fp), ("rows", C.c_int32), ("lam", C.c_float), ("T", _fp),
                ("exposure_a", _fp), ("exposure_b", _fp), ("x_out", _fp),
                ("sj_tau", _fp), ("sj_exposure", _fp), ("lm_state", _fp), ("loss", _fp),
                ("increase_factor", C.c_float), ("decrease_factor", C.c_float),
                ("min_lambda", C.c_float), ("max_lambda", C.c_float), ("converged_threshold", C.c_float),
                ("reserved0", C.c_int32), ("best", _fp), ("projection", _fp), ("viewmatrix_out", _fp),
                ("projmatrix_out", _fp), ("zero_after", _fp), ("zero_count", C.c_int32), ("reserved1", C.c_int32)]


class TrackingLossArgs(C.Structure):
    _fields_ = ([(n, _fp) for n in ("image", "opacity", "gt", "mask", "exposure_a", "exposure_b")]
                + [("exposure_eps", C.c_float), ("huber_delta", C.c_float), ("num_pixels", C.c_int64)]
                + [(n, _fp) for n in ("partial", "scalars", "grad_out", "grad_image", "grad_a", "grad_b")]
                + [("pnorm", C.c_float), ("reserved0", C.c_int32)])


class TrackingIterArgs(C.Structure):
    _fields_ = [("fwd", ForwardArgs), ("bwd", _fp), ("grad_image", _fp), ("grad_tau", _fp),
                ("grad_exposure", _fp), ("one", _fp), ("loss", TrackingLossArgs),
                ("adam", PoseAdamArgs), ("camera_matrices_valid", C.c_int32), ("reserved0", C.c_int32),
                ("best", _fp)]


class SketchResidualArgs(C.Structure):
    _fields_ = ([(n, _fp) for n in ("image", "opacity", "gt", "mask", "exposure_a", "exposure_b")]
                + [("exposure_eps", C.c_float), ("huber_delta", C.c_float), ("num_pixels", C.c_int64),
                   ("stack_dim", C.c_int32), ("sketch_dim", C.c_int32)]
                + [(n, _fp) for n in ("bucket", "weights", "grad_image", "Sf", "sj_exposure", "l1")]
                + [("assign", C.c_int32), ("reserved0", C.c_int32), ("assign_key", C.c_uint64)])


class TrackingSOArgs(C.Structure):
    _fields_ = [("base", TrackingIterArgs), ("stack_dim", C.c_int32), ("sketch_dim", C.c_int32),
                ("key", C.c_uint64), ("bucket", _fp), ("weights", _fp), ("accum", _fp),
                ("sketch_ws", _fp), ("lm", LMStepArgs), ("scratch_kept_zero", C.c_int32), ("repeat_dim", C.c_int32)]


ADAM_MAX_GROUPS = 8
GATHER_MAX_TENSORS = 24
GATHER_COPY, GATHER_ZERO_NEW, GATHER_SPLIT_SCALING, GATHER_SPLIT_XYZ = 0, 1, 2, 3


class AdamGroup(C.Structure):
    _fields_ = [("param", _fp), ("grad", _fp), ("exp_avg", _fp), ("exp_avg_sq", _fp),
                ("numel", C.c_int64), ("lr", C.c_float), ("step", C.c_int32)]


class MapPlanArgs(C.Structure):
    _fields_ = ([("n", C.c_int32)] + [(n, _fp) for n in ("grad_accum", "denom", "log_scales", "opacity_logit")]
                + [(n, C.c_float) for n in ("grad_threshold", "dense_extent", "min_opacity", "big_extent")]
                + [(n, _fp) for n in ("prune_mask", "flags", "block_counts", "totals", "src_index", "noise_row")])


class GatherTensor(C.Structure):
    _fields_ = [("src", _fp), ("dst", _fp), ("width", C.c_int32), ("mode", C.c_int32)]


class MapGatherArgs(C.Structure):
    _fields_ = [("tensors", GatherTensor * GATHER_MAX_TENSORS), ("num_tensors", C.c_int32),
                ("rows", C.c_int64), ("num_children", C.c_int32), ("src_index", _fp),
                ("rotations", _fp), ("log_scales", _fp), ("noise", _fp), ("noise_row", _fp)]


class MapActivateArgs(C.Structure):
    _fields_ = ([("num_gaussians", C.c_int32), ("scale_dims", C.c_int32), ("sh_coeffs", C.c_int32)]
                + [(n, _fp) for n in ("log_scales", "raw_rotations", "opacity_logits", "features_dc",
                                      "features_rest", "scales", "rotations", "opacities", "shs")])


class MappingViewArgs(C.Structure):
    _fields_ = [("fwd", ForwardArgs), ("bwd", _fp), ("grad_image", _fp), ("grad_depth", _fp), ("grad_tau", _fp),
                ("loss", MappingLossArgs), ("adam", PoseAdamArgs), ("accum", MapAccumArgs),
                ("loss_view", _fp), ("loss_accum", _fp), ("camera_matrices_valid", C.c_int32),
                ("forward_only", C.c_int32)]


class MapFinishArgs(C.Structure):
    _fields_ = ([("num_gaussians", C.c_int32)]
                + [(n, _fp) for n in ("gradnorm_inc", "denom_inc", "radii_max", "xyz_gradient_accum", "denom",
                                      "max_radii2D")]
                + [("reset_mode", C.c_int32), ("reset_value", C.c_float)]
                + [(n, _fp) for n in ("opacity_logits", "opacity_exp_avg", "opacity_exp_avg_sq")])


class MapAppendArgs(C.Structure):
    _fields_ = [("tensors", GatherTensor * GATHER_MAX_TENSORS), ("new_rows", _fp * GATHER_MAX_TENSORS),
                ("num_tensors", C.c_int32), ("rows_old", C.c_int64), ("rows_new", C.c_int64)]


_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the native library; raise loudly if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback.")
    # PyTorch first: its wheel carries its own copy of the HIP runtime.  If this library were loaded before
    # torch, its DT_NEEDED libamdhip64 would bind to the system copy, torch would bring in (and initialise)
    # a second runtime instance, and launches through this library fail with "no ROCm-capable device is
    # detected" (seen with `python __graft_entry__.py smoke`, which builds - and used to load - before
    # importing torch).  With torch's runtime already in the process the loader resolves ours to it.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(L, name):
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}")
    L.mgs_camera_from_pose.restype = C.c_int32
    L.mgs_camera_from_pose.argtypes = [C.c_void_p] * 5
    L.mgs_tracking_iteration.restype = C.c_int32
    L.mgs_tracking_iteration.argtypes = [C.POINTER(TrackingIterArgs), C.c_void_p]
    L.mgs_adam_step_multi.restype = C.c_int32
    L.mgs_adam_step_multi.argtypes = [C.POINTER(AdamGroup), C.c_int32, C.c_double, C.c_double, C.c_double,
                                      C.c_void_p]
    L.mgs_tracking_loss_fused.restype = C.c_int32
    L.mgs_tracking_loss_fused.argtypes = [C.POINTER(TrackingLossArgs), C.POINTER(C.c_int32), C.c_void_p]
    L.mgs_tracking_loss_onepass.restype = C.c_int32
    L.mgs_tracking_loss_onepass.argtypes = [C.POINTER(TrackingLossArgs), C.POINTER(C.c_int32), C.c_void_p]
    L.mgs_sketch_assign.restype = C.c_int32
    L.mgs_sketch_assign.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p,
                                    C.c_void_p]
    L.mgs_sketch_residual.restype = C.c_int32
    L.mgs_sketch_residual.argtypes = [C.POINTER(SketchResidualArgs), C.c_void_p]
    L.mgs_tracking_iteration_second_order.restype = C.c_int32
    L.mgs_tracking_iteration_second_order.argtypes = [C.POINTER(TrackingSOArgs), C.c_void_p]
    L.mgs_pack_mapping_grads.restype = C.c_int32
    L.mgs_pack_mapping_grads.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mgs_map_plan_blocks.restype = C.c_int32
    L.mgs_map_plan_blocks.argtypes = [C.c_int32]
    for fn in (L.mgs_map_plan_count, L.mgs_map_plan_emit):
        fn.restype = C.c_int32
        fn.argtypes = [C.POINTER(MapPlanArgs), C.c_void_p]
    L.mgs_map_gather.restype = C.c_int32
    L.mgs_map_gather.argtypes = [C.POINTER(MapGatherArgs), C.c_void_p]
    L.mgs_abi_version.restype = C.c_int32
    L.mgs_status_string.restype = C.c_char_p
    L.mgs_status_string.argtypes = [C.c_int32]
    L.mgs_raster_workspace_query.restype = C.c_int32
    L.mgs_raster_workspace_query.argtypes = [C.POINTER(RasterShape), C.POINTER(WorkspaceSizes)]
    for fn in (L.mgs_raster_forward_project, L.mgs_raster_forward_blend):
        fn.restype = C.c_int32
        fn.argtypes = [C.POINTER(ForwardArgs), C.c_void_p]
    L.mgs_raster_backward.restype = C.c_int32
    L.mgs_raster_backward.argtypes = [C.POINTER(BackwardArgs), C.c_void_p]
    L.mgs_knn_scratch_bytes.restype = C.c_uint64
    L.mgs_knn_scratch_bytes.argtypes = [C.c_int32]
    L.mgs_knn_dist2.restype = C.c_int32
    L.mgs_knn_dist2.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.mgs_profile_enable.restype = C.c_int32
    L.mgs_profile_enable.argtypes = [C.c_int32]
    L.mgs_profile_read.restype = C.c_int32
    L.mgs_profile_read.argtypes = [C.c_int32, C.c_char_p, C.POINTER(C.c_float),
                                   C.POINTER(C.c_int32)]
    L.mgs_pose_adam_step.restype = C.c_int32
    L.mgs_pose_adam_step.argtypes = [C.POINTER(PoseAdamArgs), C.c_void_p]
    for fn in (L.mgs_mapping_loss_forward, L.mgs_mapping_loss_backward):
        fn.restype = C.c_int32
        fn.argtypes = [C.POINTER(MappingLossArgs), C.c_void_p]
    L.mgs_lm_solve_step.restype = C.c_int32
    L.mgs_lm_solve_step.argtypes = [C.POINTER(LMStepArgs), C.c_void_p]
    L.mgs_tracking_loss_partial_count.restype = C.c_int32
    L.mgs_tracking_loss_partial_count.argtypes = [C.c_int64]
    for fn in (L.mgs_tracking_loss_forward, L.mgs_tracking_loss_backward):
        fn.restype = C.c_int32
        fn.argtypes = [C.POINTER(TrackingLossArgs), C.c_void_p]
    L.mgs_struct_size.restype = C.c_int32
    L.mgs_struct_size.argtypes = [C.c_int32]
    L.mgs_map_activate.restype = C.c_int32
    L.mgs_map_activate.argtypes = [C.POINTER(MapActivateArgs), C.c_void_p]
    L.mgs_mapping_loss_partial_count.restype = C.c_int32
    L.mgs_mapping_loss_partial_count.argtypes = [C.c_int64]
    L.mgs_mapping_loss_fused.restype = C.c_int32
    L.mgs_mapping_loss_fused.argtypes = [C.POINTER(MappingLossArgs), C.c_void_p, C.c_void_p]
    L.mgs_mapping_view_iteration.restype = C.c_int32
    L.mgs_mapping_view_iteration.argtypes = [C.POINTER(MappingViewArgs), C.c_void_p]
    L.mgs_map_finish_iteration.restype = C.c_int32
    L.mgs_map_finish_iteration.argtypes = [C.POINTER(MapFinishArgs), C.c_void_p]
    L.mgs_map_append.restype = C.c_int32
    L.mgs_map_append.argtypes = [C.POINTER(MapAppendArgs), C.c_void_p]
    if L.mgs_abi_version() != ABI_VERSION:
        raise NativeLibraryError(
            f"ABI mismatch: library {L.mgs_abi_version()} vs binding {ABI_VERSION}")
    _lib = L
    return L


def struct_mirrors():
    """ctypes mirror of every argument struct, in mgs_struct_size() order."""
    return [RasterShape, WorkspaceSizes, ForwardArgs, BackwardArgs, PoseAdamArgs, MappingLossArgs,
            LMStepArgs, TrackingLossArgs, TrackingIterArgs, SketchResidualArgs, TrackingSOArgs,
            AdamGroup, MapPlanArgs, GatherTensor, MapGatherArgs, MapAccumArgs, MapActivateArgs,
            MappingViewArgs, MapFinishArgs, MapAppendArgs]


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().mgs_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (status {status})")


def workspace_sizes(shape: RasterShape) -> WorkspaceSizes:
    out = WorkspaceSizes()
    check(lib().mgs_raster_workspace_query(C.byref(shape), C.byref(out)),
          "mgs_raster_workspace_query")
    return out


def profile_enable(on: bool) -> None:
    check(lib().mgs_profile_enable(1 if on else 0), "mgs_profile_enable")


def profile_read(max_entries: int = 64) -> dict:
    """{kernel name: (total_ms, launches)} since the last read (synchronises)."""
    names = C.create_string_buffer(32 * max_entries)
    ms = (C.c_float * max_entries)()
    cnt = (C.c_int32 * max_entries)()
    n = lib().mgs_profile_read(max_entries, names, ms, cnt)
    if n < 0:
        check(n, "mgs_profile_read")
    out = {}
    for k in range(n):
        nm = names.raw[32 * k:32 * k + 32].split(b"\0", 1)[0].decode()
        out[nm] = (float(ms[k]), int(cnt[k]))
    return out
