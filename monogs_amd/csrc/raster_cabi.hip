// extern "C" entry points declared in include/monogs_raster.h.
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "launch.h"
#include "raster_kernels.h"
#include "sketch_kernels.h"

namespace mgs {
namespace {
struct ProfRec { const char* name; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;
}  // namespace
bool profile_on() { return g_prof_on; }
hipError_t& launch_error_slot() {
  static thread_local hipError_t e = hipSuccess;
  return e;
}
void profile_push(const char* name, hipEvent_t a, hipEvent_t b) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back({name, a, b});
}
}  // namespace mgs

namespace mgs {
int launch_forward_project(const KP& P, hipStream_t st);
int launch_forward_blend(const KP& P, hipStream_t st);
int launch_backward(const KP& P, const KB& B, hipStream_t st, bool skip_tau_reduce, const SketchFuse* fuse);
int launch_knn(const float* pts, int n, float* out, void* scratch, hipStream_t st);
int launch_visibility(const int* n_touched, unsigned char* vis, int n, hipStream_t st);
uint64_t knn_scratch_bytes(int n);
}  // namespace mgs

using namespace mgs;

namespace {

bool shape_ok(const mgs_raster_shape& s) {
  return s.num_gaussians >= 1 && s.width >= 1 && s.height >= 1 && s.sh_degree >= 0 &&
         s.sh_degree <= 3 && s.tanfovx > 0.f && s.tanfovy > 0.f && s.pair_capacity >= 0;
}

int fill_kp(const mgs_forward_args& a, bool need_bins, bool need_outputs, KP& P) {
  const mgs_raster_shape& s = a.shape;
  if (!shape_ok(s)) return MGS_ERR_BAD_ARGUMENT;
  if (!a.means3D || !a.opacities || !a.viewmatrix || !a.projmatrix || !a.projmatrix_raw ||
      !a.campos || !a.bg || !a.geom)
    return MGS_ERR_BAD_ARGUMENT;
  if (!a.cov3D_precomp && (!a.scales || !a.rotations)) return MGS_ERR_BAD_ARGUMENT;
  if (!a.shs && !a.colors_precomp) return MGS_ERR_BAD_ARGUMENT;
  if (a.shs && s.sh_coeffs < (s.sh_degree + 1) * (s.sh_degree + 1)) return MGS_ERR_BAD_ARGUMENT;
  if (need_bins && s.pair_capacity > 0 && !a.bins) return MGS_ERR_BAD_ARGUMENT;
  if (need_outputs &&
      (!a.out_color || !a.out_depth || !a.out_opacity || !a.radii || !a.n_touched))
    return MGS_ERR_BAD_ARGUMENT;
  const Layout L = make_layout(s);
  P.obj = KObj{};
  P.N = s.num_gaussians; P.W = s.width; P.H = s.height;
  P.grid_x = (s.width + kTile - 1) / kTile; P.grid_y = (s.height + kTile - 1) / kTile;
  P.T = P.grid_x * P.grid_y;
  P.pack = (s.num_gaussians <= kPackMaxN && P.T <= (1 << kPackBits)) ? 1 : 0;
  P.deg = s.sh_degree; P.K = a.shs ? s.sh_coeffs : 0; P.cap = s.pair_capacity;
  P.tanfovx = s.tanfovx; P.tanfovy = s.tanfovy;
  P.focal_x = s.width / (2.0f * s.tanfovx); P.focal_y = s.height / (2.0f * s.tanfovy);
  P.mod = s.scale_modifier;
  P.means = a.means3D; P.scales = a.cov3D_precomp ? nullptr : a.scales;
  P.rots = a.cov3D_precomp ? nullptr : a.rotations; P.covp = a.cov3D_precomp;
  P.opac = a.opacities; P.shs = a.colors_precomp ? nullptr : a.shs; P.precol = a.colors_precomp;
  P.V = a.viewmatrix; P.PM = a.projmatrix; P.Praw = a.projmatrix_raw; P.campos = a.campos;
  P.bg = a.bg;
  char* g = (char*)a.geom;
  P.rec = (SplatRec*)(g + L.rec);
  P.pair_count = (int*)(g + L.pair_count);
  P.pair_off = (int*)(g + L.pair_off);
  P.block_prefix = (int*)(g + L.block_prefix);
  P.scan_tmp = (int*)(g + L.scan_tmp);
  P.hit_mask = (unsigned int*)(g + L.hit_mask);
  P.tile_count = (int*)(g + L.tile_count);
  P.tile_offset = (int*)(g + L.tile_offset);
  P.tile_cursor = (int*)(g + L.tile_cursor);
  P.bin_table = (int*)(g + L.bin_table);
  P.final_TC = (float4*)(g + L.final_TC);
  P.final_DL = (int2*)(g + L.final_DL);
  P.seg_offset = (int*)(g + L.seg_offset);
  P.quad_last = (int*)(g + L.quad_last);
  P.counters = (int*)(g + L.counters);
  char* b = (char*)a.bins;
  P.keys = b ? (unsigned long long*)(b + L.keys) : nullptr;
  P.payload = b ? (unsigned int*)(b + L.payload) : nullptr;
  P.seg_rec = b ? (int4*)(b + L.seg_rec) : nullptr;
  P.ckpt = b ? (float*)(b + L.ckpt) : nullptr;
  P.reach = b ? (uint4*)(b + L.reach) : nullptr;
  P.max_segs = (int)L.max_segs;
  P.out_color = a.out_color; P.out_depth = a.out_depth; P.out_opacity = a.out_opacity;
  P.radii = a.radii; P.n_touched = a.n_touched; P.d_out = a.pair_count_out; P.d_max = a.pair_count_max;
  if (P.T <= kBinMaxTilesLds) { int nblk; bin_grid(P.N, nblk, P.per_block); }
  else P.per_block = 0x7fffffff;
  P.big_pass = a.big_tile_pass < 0 ? 0 : 1;
  P.clamp_up = 1;
  return MGS_OK;
}

}  // namespace

extern "C" {

int32_t mgs_abi_version(void) { return MGS_ABI_VERSION; }

int32_t mgs_struct_size(int32_t which) {
  switch (which) {
    case 0: return (int32_t)sizeof(mgs_raster_shape);
    case 1: return (int32_t)sizeof(mgs_workspace_sizes);
    case 2: return (int32_t)sizeof(mgs_forward_args);
    case 3: return (int32_t)sizeof(mgs_backward_args);
    case 4: return (int32_t)sizeof(mgs_pose_adam_args);
    case 5: return (int32_t)sizeof(mgs_mapping_loss_args);
    case 6: return (int32_t)sizeof(mgs_lm_step_args);
    case 7: return (int32_t)sizeof(mgs_tracking_loss_args);
    case 8: return (int32_t)sizeof(mgs_tracking_iter_args);
    case 9: return (int32_t)sizeof(mgs_sketch_residual_args);
    case 10: return (int32_t)sizeof(mgs_tracking_so_args);
    case 11: return (int32_t)sizeof(mgs_adam_group);
    case 12: return (int32_t)sizeof(mgs_map_plan_args);
    case 13: return (int32_t)sizeof(mgs_gather_tensor);
    case 14: return (int32_t)sizeof(mgs_map_gather_args);
    case 15: return (int32_t)sizeof(mgs_map_accum_args);
    case 16: return (int32_t)sizeof(mgs_map_activate_args);
    case 17: return (int32_t)sizeof(mgs_mapping_view_args);
    case 18: return (int32_t)sizeof(mgs_map_finish_args);
    case 19: return (int32_t)sizeof(mgs_map_append_args);
    default: return -1;
  }
}

const char* mgs_status_string(int32_t status) {
  switch (status) {
    case MGS_OK: return "ok";
    case MGS_ERR_BAD_ARGUMENT: return "bad argument (null pointer, non-positive size or unsupported degree)";
    case MGS_ERR_LAUNCH: return "kernel launch failed (the kernel and HIP's message were written to stderr)";
    case MGS_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}

int32_t mgs_raster_workspace_query(const mgs_raster_shape* shape, mgs_workspace_sizes* out) {
  if (!shape || !out || !shape_ok(*shape)) return MGS_ERR_BAD_ARGUMENT;
  const Layout L = make_layout(*shape);
  out->geom_bytes = L.geom_bytes; out->bins_bytes = L.bins_bytes; out->bwd_bytes = L.bwd_bytes;
  out->sketch_bytes = L.sketch_bytes;
  out->off_records = L.rec; out->off_pair_count = L.pair_count;
  out->off_tile_offset = L.tile_offset; out->off_final_T = L.final_TC;
  out->off_n_contrib = L.final_DL; out->off_counters = L.counters;
  out->off_keys = L.keys; out->off_payload = L.payload;
  return MGS_OK;
}

int32_t mgs_raster_forward_project(const mgs_forward_args* args, void* stream) {
  if (!args) return MGS_ERR_BAD_ARGUMENT;
  KP P;
  const int rc = fill_kp(*args, false, true, P);
  if (rc != MGS_OK) return rc;
  return launch_forward_project(P, (hipStream_t)stream);
}

int32_t mgs_raster_forward_blend(const mgs_forward_args* args, void* stream) {
  if (!args) return MGS_ERR_BAD_ARGUMENT;
  KP P;
  const int rc = fill_kp(*args, true, true, P);
  if (rc != MGS_OK) return rc;
  return launch_forward_blend(P, (hipStream_t)stream);
}

// skip_tau_reduce: the caller sums tau_partial itself (*tau_partials / *num_partials are set)
static int32_t raster_backward_impl(const mgs_backward_args* args, void* stream, bool skip_tau_reduce,
                                    const float** tau_partials, int32_t* num_partials,
                                    bool sketch_only = false, bool scratch_kept_zero = false,
                                    const SketchFuse* fuse = nullptr) {
  if (!args) return MGS_ERR_BAD_ARGUMENT;
  KP P;
  const int rc = fill_kp(args->fwd, true, false, P);
  if (rc != MGS_OK) return rc;
  if (!args->grad_color || !args->bwd || !args->grad_tau) return MGS_ERR_BAD_ARGUMENT;
  if (args->clamp_gradient_mode != 0 && args->clamp_gradient_mode != 1) return MGS_ERR_BAD_ARGUMENT;
  P.clamp_up = args->clamp_gradient_mode == MGS_CLAMP_GRAD_UPSTREAM;
  {   // per-Gaussian gradients: all of the mandatory four, or none at all (pose-only / mapping mode)
    const int have = (args->grad_means3D != nullptr) + (args->grad_means2D != nullptr) +
                     (args->grad_colors != nullptr) + (args->grad_opacities != nullptr);
    if (have != 0 && have != 4) return MGS_ERR_BAD_ARGUMENT;
    if (have == 0 && (args->grad_scales || args->grad_rotations || args->grad_cov3D))
      return MGS_ERR_BAD_ARGUMENT;
    if (args->map_accum) {
      const mgs_map_accum_args& m = *args->map_accum;
      if (have != 0 || args->sketch_mode != 0 || !args->fwd.scales || !args->fwd.rotations || !args->fwd.shs ||
          args->fwd.cov3D_precomp || args->fwd.colors_precomp || (m.scale_dims != 1 && m.scale_dims != 3) ||
          !m.raw_rotations || !m.grad_xyz || !m.grad_features_dc || !m.grad_opacity || !m.grad_scaling ||
          !m.grad_rotation || (args->fwd.shape.sh_coeffs > 1 && !m.grad_features_rest) ||
          ((m.gradnorm_inc != nullptr) != (m.denom_inc != nullptr)) || (m.visibility && !args->fwd.n_touched))
        return MGS_ERR_BAD_ARGUMENT;
    }
  }
  if (args->sketch_mode != 0 &&
      ((!args->sketch_indices && !args->sketch_bucket_flat) || !args->grad_sketch_dtau || !args->sketch_ws ||
       args->sketch_dim < 1 || args->stack_dim < 1))
    return MGS_ERR_BAD_ARGUMENT;
  const Layout L = make_layout(args->fwd.shape);
  KB B;
  B.grad_color = args->grad_color; B.grad_depth = args->grad_depth;
  char* w = (char*)args->bwd;
  B.pair_grad = (float*)(w + L.pair_grad);
  B.tau_partial = (float*)(w + L.tau_partial);
  B.g_means3D = args->grad_means3D; B.g_means2D = args->grad_means2D;
  B.g_colors = args->grad_colors; B.g_opac = args->grad_opacities;
  B.g_scales = args->grad_scales; B.g_rots = args->grad_rotations; B.g_cov = args->grad_cov3D;
  B.g_tau = args->grad_tau;
  B.pair_bound = args->pair_count_bound > 0 ? args->pair_count_bound : 0;
  B.sketch_mode = args->sketch_mode; B.sketch_dim = args->sketch_dim; B.stack_dim = args->stack_dim;
  B.sketch_only = (sketch_only && args->sketch_mode != 0) ? 1 : 0;
  B.scratch_kept_zero = (scratch_kept_zero && args->sketch_mode != 0) ? 1 : 0;
  B.sketch_idx = args->sketch_indices; B.g_sketch = args->grad_sketch_dtau;
  B.sketch_flat = args->sketch_indices ? nullptr : args->sketch_bucket_flat;
  char* sw = (char*)args->sketch_ws;
  B.slabs = sw ? (float*)(sw + L.slabs) : nullptr;
  B.slab_mask = sw ? (unsigned int*)(sw + L.slab_mask) : nullptr;
  B.splat_jac = sw ? (float*)(sw + L.splat_jac) : nullptr;
  memset(&B.map, 0, sizeof(B.map));
  if (args->map_accum) {
    const mgs_map_accum_args& m = *args->map_accum;
    KM& M = B.map;
    M.on = 1; M.scale_dims = m.scale_dims; M.accumulate = m.accumulate; M.add_reg = m.add_regulariser;
    M.reg_scale = m.regulariser_weight / (3.0f * (float)P.N);
    M.raw_rot = m.raw_rotations;
    M.g_xyz = m.grad_xyz; M.g_fdc = m.grad_features_dc; M.g_frest = m.grad_features_rest;
    M.g_opacity = m.grad_opacity; M.g_scaling = m.grad_scaling; M.g_rotation = m.grad_rotation;
    M.gradnorm_inc = m.gradnorm_inc; M.denom_inc = m.denom_inc; M.radii_max = m.radii_max;
    M.visibility = m.visibility;
    P.n_touched = args->fwd.n_touched;
  }
  if (tau_partials) *tau_partials = B.tau_partial;
  if (num_partials) *num_partials = (P.N + kPreBlock - 1) / kPreBlock;
  return launch_backward(P, B, (hipStream_t)stream, skip_tau_reduce, fuse);
}

int32_t mgs_raster_backward(const mgs_backward_args* args, void* stream) {
  return raster_backward_impl(args, stream, false, nullptr, nullptr);
}

int32_t mgs_tracking_iteration(const mgs_tracking_iter_args* args, void* stream) {
  if (!args || !args->bwd || !args->grad_image || !args->grad_tau || !args->grad_exposure ||
      !args->one || !args->adam.T || !args->fwd.viewmatrix || !args->fwd.projmatrix)
    return MGS_ERR_BAD_ARGUMENT;
  if (args->fwd.shape.pair_capacity < 1) return MGS_ERR_BAD_ARGUMENT;
  // 9 launches: camera matrices (unless the caller says they are valid - the previous
  // iteration's Adam kernel has already written them), 5 forward (the last one also evaluates the objective:
  // un-normalised gradients + partial sums), 2 backward, 1 Adam + update_pose (which also sums the tau / exposure / squared-residual block
  // partials, applies the 1 / loss of the norm and refreshes the camera matrices).
  int32_t rc = MGS_OK;
  if (!args->camera_matrices_valid) {
    rc = mgs_camera_from_pose(args->adam.T, args->fwd.projmatrix_raw, const_cast<float*>(args->fwd.viewmatrix),
                              const_cast<float*>(args->fwd.projmatrix), stream);
    if (rc != MGS_OK) return rc;
  }
  if ((rc = mgs_raster_forward_project(&args->fwd, stream)) != MGS_OK) return rc;
  // The objective rides in the epilogue of the forward blend (raster_kernels.h: KObj): the quadrant wave that
  // finishes a pixel forms its residual, writes d(loss)/d(image) and leaves its sums as one partial per wave
  // in the geom workspace - no loss launch between the forward and the backward.
  mgs_tracking_loss_args L = args->loss;
  if (!L.gt || !L.exposure_a || !L.exposure_b || !L.scalars) return MGS_ERR_BAD_ARGUMENT;
  KP P;
  if ((rc = fill_kp(args->fwd, true, true, P)) != MGS_OK) return rc;
  const Layout lay = make_layout(args->fwd.shape);
  float* obj_partial = reinterpret_cast<float*>(static_cast<char*>(args->fwd.geom) + lay.obj_partial);
  if (L.pnorm > 0.f && L.pnorm < 1.f) return MGS_ERR_BAD_ARGUMENT;
  const float pn = L.pnorm > 0.f ? L.pnorm : 2.f;
  int32_t nblk = 4 * P.T;
  if (pn == 1.f || pn == 2.f) {
    P.obj.on = 1;
    P.obj.p1 = pn == 1.f;
    P.obj.exposure_eps = L.exposure_eps; P.obj.huber_delta = L.huber_delta;
    P.obj.gt = L.gt; P.obj.mask = L.mask; P.obj.exposure_a = L.exposure_a; P.obj.exposure_b = L.exposure_b;
    P.obj.grad_image = args->grad_image; P.obj.partial = obj_partial;
    if ((rc = launch_forward_blend(P, (hipStream_t)stream)) != MGS_OK) return rc;
    L.partial = obj_partial;
  } else {
    // any other p >= 1 (powf per sample): the plain forward blend, then the one-pass loss kernel
    // (needs loss.partial: mgs_tracking_loss_partial_count floats)
    if (!L.partial) return MGS_ERR_BAD_ARGUMENT;
    if ((rc = launch_forward_blend(P, (hipStream_t)stream)) != MGS_OK) return rc;
    L.image = args->fwd.out_color; L.opacity = args->fwd.out_opacity; L.grad_image = args->grad_image;
    L.num_pixels = (int64_t)P.W * P.H;
    if ((rc = mgs_tracking_loss_onepass(&L, &nblk, stream)) != MGS_OK) return rc;
  }
  mgs_backward_args B;
  memset(&B, 0, sizeof(B));
  B.fwd = args->fwd;
  B.grad_color = args->grad_image;
  B.bwd = args->bwd;
  B.grad_tau = args->grad_tau;
  const float* tau_partials = nullptr;
  int32_t npre = 0;
  if ((rc = raster_backward_impl(&B, stream, true, &tau_partials, &npre)) != MGS_OK) return rc;
  mgs_pose_adam_args A = args->adam;
  A.grad_trans = nullptr; A.grad_rot = nullptr; A.grad_a = nullptr; A.grad_b = nullptr;
  A.tau_partials = tau_partials; A.num_tau_partials = npre;
  A.exposure_partials = L.partial + nblk; A.num_exposure_partials = nblk;
  A.loss_partials = L.partial; A.num_loss_partials = nblk;
  A.loss_norm_mode = 1; A.loss_grad_out = args->one; A.loss_pnorm = pn;
  A.loss_view = L.scalars; A.loss_accum = nullptr;
  A.best = args->best; A.l1_partials = L.partial + 3 * (size_t)nblk; A.num_l1_partials = nblk;
  A.projection = args->fwd.projmatrix_raw;
  A.viewmatrix_out = const_cast<float*>(args->fwd.viewmatrix);
  A.projmatrix_out = const_cast<float*>(args->fwd.projmatrix);
  return mgs_pose_adam_step(&A, stream);
}

int32_t mgs_mapping_view_iteration(const mgs_mapping_view_args* args, void* stream) {
  if (!args || !args->adam.T || !args->fwd.viewmatrix || !args->fwd.projmatrix || args->fwd.shape.pair_capacity < 1)
    return MGS_ERR_BAD_ARGUMENT;
  if (!args->forward_only && (!args->bwd || !args->grad_image || !args->grad_tau || !args->loss.partial || !args->loss.gt))
    return MGS_ERR_BAD_ARGUMENT;
  int32_t rc = MGS_OK;
  if (!args->camera_matrices_valid) {
    rc = mgs_camera_from_pose(args->adam.T, args->fwd.projmatrix_raw, const_cast<float*>(args->fwd.viewmatrix),
                              const_cast<float*>(args->fwd.projmatrix), stream);
    if (rc != MGS_OK) return rc;
  }
  if ((rc = mgs_raster_forward_project(&args->fwd, stream)) != MGS_OK) return rc;
  if ((rc = mgs_raster_forward_blend(&args->fwd, stream)) != MGS_OK) return rc;
  if (args->forward_only) {
    if (args->accum.visibility) {
      return launch_visibility((const int*)args->fwd.n_touched, args->accum.visibility,
                               args->fwd.shape.num_gaussians, (hipStream_t)stream);
    }
    return MGS_OK;
  }
  // objective: value + gradients in one pass; the block sums are consumed by the Adam kernel
  mgs_mapping_loss_args L = args->loss;
  L.image = args->fwd.out_color; L.depth = args->fwd.out_depth;
  L.grad_image = args->grad_image; L.grad_depth = L.w_depth != 0.f ? args->grad_depth : nullptr;
  L.grad_out = nullptr;
  int32_t nblk = 0;
  if ((rc = mgs_mapping_loss_fused(&L, &nblk, stream)) != MGS_OK) return rc;
  // full backward in mapping mode
  mgs_backward_args B;
  memset(&B, 0, sizeof(B));
  B.fwd = args->fwd;
  B.grad_color = args->grad_image;
  B.grad_depth = L.grad_depth;
  B.bwd = args->bwd;
  B.grad_tau = args->grad_tau;
  B.map_accum = &args->accum;
  const float* tau_partials = nullptr;
  int32_t npre = 0;
  if ((rc = raster_backward_impl(&B, stream, true, &tau_partials, &npre)) != MGS_OK) return rc;
  // this view's pose / exposure optimiser step (+ update_pose), loss value
  mgs_pose_adam_args A = args->adam;
  A.grad_trans = nullptr; A.grad_rot = nullptr; A.grad_a = nullptr; A.grad_b = nullptr;
  A.tau_partials = tau_partials; A.num_tau_partials = npre;
  A.exposure_partials = L.apply_exposure ? L.partial + 2 * nblk : nullptr; A.num_exposure_partials = nblk;
  if (!L.apply_exposure) { A.exposure_a = nullptr; A.exposure_b = nullptr; }
  A.loss_partials = L.partial; A.num_loss_partials = nblk;
  const float hw = (float)L.num_pixels;
  A.loss_w_rgb = L.w_rgb / (3.f * hw); A.loss_w_depth = L.w_depth / hw;
  A.loss_view = args->loss_view; A.loss_accum = args->loss_accum;
  A.projection = args->fwd.projmatrix_raw;
  A.viewmatrix_out = const_cast<float*>(args->fwd.viewmatrix);
  A.projmatrix_out = const_cast<float*>(args->fwd.projmatrix);
  return mgs_pose_adam_step(&A, stream);
}

int32_t mgs_tracking_iteration_second_order(const mgs_tracking_so_args* args, void* stream) {
  if (!args) return MGS_ERR_BAD_ARGUMENT;
  const mgs_tracking_iter_args& b = args->base;
  if (!b.bwd || !b.grad_image || !b.grad_tau || !b.adam.T || !b.fwd.viewmatrix || !b.fwd.projmatrix ||
      b.fwd.shape.pair_capacity < 1 || !args->bucket || !args->weights || !args->accum || !args->sketch_ws ||
      !args->lm.lm_state || !args->lm.x_out || args->stack_dim < 1 || args->sketch_dim < 1 || args->repeat_dim < 0)
    return MGS_ERR_BAD_ARGUMENT;
  const int64_t HW = (int64_t)b.fwd.shape.width * b.fwd.shape.height;
  const int d = args->stack_dim * args->sketch_dim;
  const int R = args->repeat_dim > 0 ? args->repeat_dim : 1;
  const size_t rows = (size_t)R * d;
  // accum: Sf[R d] | sj_exposure[R d, 2] | sj_tau[R d, 6] | l1, l1 of the later repeats (dropped), pad
  float* Sf = args->accum;
  float* sj_exp = Sf + rows;
  float* sj_tau = sj_exp + 2 * rows;
  float* l1 = sj_tau + 6 * rows;
  hipStream_t st = (hipStream_t)stream;
  int32_t rc = MGS_OK;
  if (!b.camera_matrices_valid) {
    rc = mgs_camera_from_pose(b.adam.T, b.fwd.projmatrix_raw, const_cast<float*>(b.fwd.viewmatrix),
                              const_cast<float*>(b.fwd.projmatrix), stream);
    if (rc != MGS_OK) return rc;
  }
  if ((rc = mgs_raster_forward_project(&b.fwd, stream)) != MGS_OK) return rc;
  if ((rc = mgs_raster_forward_blend(&b.fwd, stream)) != MGS_OK) return rc;
  // Sf, sj_exposure and l1 are accumulated with atomics (sj_tau is zeroed by the backward).  With
  // scratch_kept_zero the caller guarantees zeros on the first call and the LM kernel / the bucket kernel -
  // the consumers - restore them: four hipMemsetAsync launches (~5 us each) less per iteration.
  const bool kept = args->scratch_kept_zero != 0;
  if (!kept && (!hip_ok("memset(sketch accumulators)", hipMemsetAsync(Sf, 0, sizeof(float) * 3 * rows, st)) ||
                !hip_ok("memset(sketch l1)", hipMemsetAsync(l1, 0, sizeof(float) * 4, st)))) {
    launches_ok();
    return MGS_ERR_LAUNCH;
  }
  // From here on the accumulators may hold partial sums.  In kept-zero mode only the LAST kernels of the sequence
  // restore the zeros, so a failure in between (bad argument, unsupported size, a failed launch) must not leave
  // them dirty for the next call: clear them before returning the error (the backward's own scratch - slabs and
  // their masks - is rewritten completely by every launch).
  auto fail = [&](int32_t code) {
    if (kept) (void)hipMemsetAsync(args->accum, 0, sizeof(float) * (9 * rows + 4), st);
    return code;
  };
  // `repeat_dim` backward passes over ONE render (utils/slam_frontend.py:654-669): repeat r draws its own
  // partition and weights, sums its residual buckets into Sf[r] and harvests SJ[r]; the rows are stacked.
  for (int r = 0; r < R; r++) {
    mgs_sketch_residual_args Rr;
    memset(&Rr, 0, sizeof(Rr));
    Rr.image = b.fwd.out_color; Rr.opacity = b.fwd.out_opacity; Rr.gt = b.loss.gt; Rr.mask = b.loss.mask;
    Rr.exposure_a = b.loss.exposure_a; Rr.exposure_b = b.loss.exposure_b;
    Rr.exposure_eps = b.loss.exposure_eps; Rr.huber_delta = b.loss.huber_delta; Rr.num_pixels = HW;
    Rr.stack_dim = args->stack_dim; Rr.sketch_dim = args->sketch_dim;
    Rr.bucket = args->bucket + (size_t)r * HW; Rr.weights = args->weights + (size_t)r * HW; Rr.grad_image = b.grad_image;
    Rr.Sf = Sf + (size_t)r * d; Rr.sj_exposure = sj_exp + 2 * (size_t)r * d;
    Rr.l1 = r == 0 ? l1 : l1 + 1;               // the L1 criterion is the render's: counted once
    Rr.assign = 1;                              // the bucket partition is drawn inside the residual pass
    Rr.assign_key = args->key + 0x9E3779B97F4A7C15ull * (uint64_t)r;
    // the residual pass runs in ONE launch with the per-splat Jacobian preparation of the backward (first repeat;
    // the preparation depends on the camera alone, the later repeats reuse it)
    SketchFuse fuse;
    fuse.residual = &Rr; fuse.skip_prep = r > 0 ? 1 : 0;
    if (!Rr.gt || !Rr.exposure_a || !Rr.exposure_b || !sketch_keys(HW, args->stack_dim, args->sketch_dim, Rr.assign_key, fuse.keys))
      return fail(MGS_ERR_BAD_ARGUMENT);
    fuse.keys.on = 1;
    mgs_backward_args B;
    memset(&B, 0, sizeof(B));
    B.fwd = b.fwd; B.grad_color = b.grad_image; B.bwd = b.bwd; B.grad_tau = b.grad_tau;
    B.sketch_mode = 1; B.sketch_dim = args->sketch_dim; B.stack_dim = args->stack_dim;
    B.sketch_bucket_flat = args->bucket + (size_t)r * HW; B.grad_sketch_dtau = sj_tau + 6 * (size_t)r * d;
    B.sketch_ws = args->sketch_ws;
    // only grad_sketch_dtau is consumed by the LM step: J-only backward (no per-splat sums,
    // no preprocess backward, no grad_tau)
    if ((rc = raster_backward_impl(&B, stream, true, nullptr, nullptr, true, kept, &fuse)) != MGS_OK) return fail(rc);
  }
  mgs_lm_step_args L = args->lm;
  L.SJ = nullptr; L.sj_tau = sj_tau; L.sj_exposure = sj_exp; L.Sf = Sf; L.rows = (int32_t)rows; L.loss = l1;
  L.T = b.adam.T; L.exposure_a = b.adam.exposure_a; L.exposure_b = b.adam.exposure_b;
  L.best = b.best;
  if (kept) { L.zero_after = args->accum; L.zero_count = (int32_t)(9 * rows + 4); }
  L.projection = b.fwd.projmatrix_raw;
  L.viewmatrix_out = const_cast<float*>(b.fwd.viewmatrix);
  L.projmatrix_out = const_cast<float*>(b.fwd.projmatrix);
  if ((rc = mgs_lm_solve_step(&L, stream)) != MGS_OK) return fail(rc);
  return MGS_OK;
}

int32_t mgs_profile_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return MGS_OK;
}

int32_t mgs_profile_read(int32_t max_entries, char* names, float* total_ms, int32_t* launches) {
  if (max_entries < 0 || (max_entries > 0 && (!names || !total_ms || !launches)))
    return MGS_ERR_BAD_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  for (ProfRec& r : g_prof) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
    int k = 0;
    for (; k < n; k++)
      if (strncmp(names + 32 * k, r.name, 31) == 0) break;
    if (k == n) {
      if (n >= max_entries) continue;
      strncpy(names + 32 * k, r.name, 31);
      names[32 * k + 31] = 0;
      total_ms[k] = 0.f;
      launches[k] = 0;
      n++;
    }
    total_ms[k] += ms;
    launches[k] += 1;
  }
  g_prof.clear();
  return n;
}

uint64_t mgs_knn_scratch_bytes(int32_t num_points) { return num_points < 1 ? 256 : knn_scratch_bytes(num_points); }

int32_t mgs_knn_dist2(const float* points, int32_t num_points, float* out, void* scratch,
                      void* stream) {
  if (!points || !out || !scratch || num_points < 1) return MGS_ERR_BAD_ARGUMENT;
  return launch_knn(points, num_points, out, scratch, (hipStream_t)stream);
}

}  // extern "C"
