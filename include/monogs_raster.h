/* C ABI of the MI355X-native differentiable Gaussian rasteriser + simple-knn.
 *
 * This is the drop-in boundary for the two native extensions MonoGS binds
 * (paths relative to the reference tree /root/reference):
 *
 *   diff_gaussian_rasterization._C.rasterize_gaussians / rasterize_gaussians_backward
 *       - constructed/called at gaussian_splatting/gaussian_renderer/__init__.py:61-77,151-168
 *       - upstream binding: submodules/diff-gaussian-rasterization (git submodule,
 *         .gitmodules:4-7, source absent from the tree)
 *   simple_knn._C.distCUDA2
 *       - called at gaussian_splatting/scene/gaussian_model.py:18,185-191
 *       - upstream binding: submodules/simple-knn (.gitmodules:1-3, source absent)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless named host_*; all floats are
 *     fp32, ids/counters int32, sort keys uint64; tensors are dense row-major;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and no
 *     entry point synchronises with the host;
 *   - the library owns no device memory and keeps no state between calls: scratch
 *     lives in caller-owned workspaces whose sizes come from mgs_raster_workspace_query;
 *   - every function returns 0 on success or a negative mgs_status; nothing throws
 *     across this boundary.  mgs_status_string() names a code.
 */
#ifndef MONOGS_RASTER_H
#define MONOGS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGS_ABI_VERSION 8

typedef enum mgs_status {
  MGS_OK = 0,
  MGS_ERR_BAD_ARGUMENT = -1,   /* null pointer / non-positive size / bad degree */
  MGS_ERR_LAUNCH = -2,         /* hipGetLastError() after a launch was not hipSuccess */
  MGS_ERR_UNSUPPORTED = -3,    /* a size the kernels are not built for (e.g. a backward with pair_capacity > 53 687 091:
                                  the pair records are addressed through a 2-GiB buffer descriptor) */
} mgs_status;

/* Problem shape shared by forward and backward. */
typedef struct mgs_raster_shape {
  int32_t num_gaussians;   /* N >= 1 (N == 0 is handled above the boundary,
                              gaussian_renderer/__init__.py:43-44) */
  int32_t width, height;
  int32_t sh_degree;       /* active degree 0..3 */
  int32_t sh_coeffs;       /* K: coefficient triples stored per Gaussian in `shs` [N,K,3] */
  int32_t pair_capacity;   /* capacity (in (tile,Gaussian) pairs) of the `bins` workspace */
  float tanfovx, tanfovy;
  float scale_modifier;
} mgs_raster_shape;

/* Byte sizes of the caller-owned workspaces for a shape. */
typedef struct mgs_workspace_sizes {
  uint64_t geom_bytes;   /* per-Gaussian records, per-tile ranges, per-pixel blend state;
                            written by forward, read by backward (save it on the autograd ctx) */
  uint64_t bins_bytes;   /* pair_capacity sorted (key, payload) pairs + per-item records, quadrant reach words and blend checkpoints (5 KB + 32 B per 32 pairs and per tile); forward -> backward */
  uint64_t bwd_bytes;    /* backward-only scratch (per-pair reduced gradients, scans) */
  uint64_t sketch_bytes; /* extra backward scratch, only when sketch_mode != 0: 144 B per Gaussian + one 6-KB slab per TWO
                            backward items and per tile, i.e. ~96 B per pair of capacity + 6 KB per tile (no initial state needed) */
  /* byte offsets inside `geom` of arrays tests may inspect */
  uint64_t off_records;      /* N x 48 B  mgs SplatRec                       */
  uint64_t off_pair_count;   /* N x int32 pairs emitted per Gaussian         */
  uint64_t off_tile_offset;  /* (T+1) x int32 exclusive scan of tile counts  */
  uint64_t off_final_T;      /* 256 T x float4 (T, C0, C1, C2), quadrant-major */
  uint64_t off_n_contrib;    /* 256 T x (float depth, int32 n_contrib)       */
  uint64_t off_counters;     /* 4 x int32: [0] = total pairs D               */
  /* byte offsets inside `bins` */
  uint64_t off_keys;         /* capacity x uint64: depth_bits<<32 | gaussian id, sorted per tile */
  uint64_t off_payload;      /* capacity x uint32: pair index within its Gaussian */
} mgs_workspace_sizes;

typedef struct mgs_forward_args {
  mgs_raster_shape shape;
  /* inputs */
  const float* means3D;         /* [N,3] */
  const float* scales;          /* [N,3] or NULL when cov3D_precomp is given */
  const float* rotations;       /* [N,4] (r,x,y,z), used as given */
  const float* cov3D_precomp;   /* [N,6] or NULL */
  const float* opacities;       /* [N] */
  const float* shs;             /* [N,K,3] or NULL when colors_precomp is given */
  const float* colors_precomp;  /* [N,3] or NULL */
  const float* viewmatrix;      /* [4,4] = T_w2c^T     (camera_utils.py:94-96)  */
  const float* projmatrix;      /* [4,4] = view @ proj (camera_utils.py:98-104) */
  const float* projmatrix_raw;  /* [4,4] projection    (slam_frontend.py:1815-1825) */
  const float* campos;          /* >= 3 floats */
  const float* bg;              /* [3] */
  /* workspaces */
  void* geom;
  void* bins;
  /* outputs */
  float* out_color;    /* [3,H,W] */
  float* out_depth;    /* [1,H,W] */
  float* out_opacity;  /* [1,H,W] */
  int32_t* radii;      /* [N] */
  int32_t* n_touched;  /* [N] */
  /* optional, TWO ints: stage 1 also stores [0] = the pair count D and [1] = the largest number of
   * pairs in one tile here.  May be pinned HOST memory (device-accessible): the caller then learns
   * both from an event recorded after stage 1 without a device-to-host copy in the stream. */
  int32_t* pair_count_out;
  /* optional DEVICE counter: stage 1 does atomicMax(*pair_count_max, D).  Never reset by the
   * library, so after any number of fixed-capacity forwards (mgs_tracking_iteration,
   * mgs_mapping_view_iteration) *pair_count_max > pair_capacity says that at least one of them
   * was rendered incompletely, whichever it was. */
  int32_t* pair_count_max;
  /* Stage 2 sorts tiles of up to 1024 pairs in one launch and tiles of 1025..4096 pairs in a second
   * one (16-wave workgroups, 32 KB of LDS) that costs ~6 us even when no such tile exists.
   * 0 (default): always launch it.  -1: skip it - a caller that knows from pair_count_out[1] of the
   * previous, nearly identical view that no tile is that crowded; should one appear after all, the
   * first launch sorts it in place in HBM (slower, same order). */
  int32_t big_tile_pass;
  int32_t reserved0;
} mgs_forward_args;

/* Mapping mode of the backward (row a13, utils/slam_backend.py:171-332): instead of storing
 * the gradients w.r.t. the ACTIVATED attributes of one view, the last kernel of the backward
 * chains them through GaussianModel's activations (gaussian_model.py:54-62,77-102: exp, sigmoid,
 * normalize, cat(features_dc, features_rest)) and accumulates them over the views of the window
 * into the flat buffer the optimiser (and, keyframe-parallel, the all-reduce) consumes - what
 * autograd's accumulation over the summed loss of slam_backend.py:183-247 does - together with
 * the densification statistics of the view (gaussian_model.py:693-697, slam_backend.py:292-299).
 * All pointers are device pointers into caller-owned buffers. */
typedef struct mgs_map_accum_args {
  int32_t scale_dims;           /* 3, or 1: isotropic model, _scaling is [N,1] (renderer broadcast :92-95) */
  int32_t accumulate;           /* 0: overwrite (first view of an iteration), 1: add */
  int32_t add_regulariser;      /* != 0: also add d/d_scaling of  weight * mean|s - mean_k(s)|
                                   (slam_backend.py:244-246; once per iteration, every Gaussian) */
  float regulariser_weight;     /* 10 in the reference */
  const float* raw_rotations;   /* _rotation [N,4] before normalisation */
  float* grad_xyz;              /* [N,3] */
  float* grad_features_dc;      /* [N,1,3] */
  float* grad_features_rest;    /* [N,K-1,3] or NULL when K == 1 */
  float* grad_opacity;          /* [N]   w.r.t. the logit */
  float* grad_scaling;          /* [N,scale_dims] w.r.t. the log scale */
  float* grad_rotation;         /* [N,4] w.r.t. the raw quaternion */
  float* gradnorm_inc;          /* [N] += ||dL/d ndc||  where radii > 0, or NULL */
  float* denom_inc;             /* [N] += 1             where radii > 0, or NULL */
  int32_t* radii_max;           /* [N] max over the views of radii, or NULL */
  uint8_t* visibility;          /* [N] = n_touched > 0 of THIS view (occ-aware visibility,
                                   slam_backend.py:251-255), or NULL */
} mgs_map_accum_args;

typedef struct mgs_backward_args {
  mgs_forward_args fwd;        /* same inputs / workspaces as the forward call */
  const float* grad_color;     /* [3,H,W] */
  const float* grad_depth;     /* [1,H,W] or NULL */
  void* bwd;                   /* bwd_bytes of scratch */
  /* outputs (all overwritten).  The per-Gaussian gradients (means3D, means2D, colors,
   * opacities, scales, rotations, cov3D) may ALL be NULL: pose-only backward (tracking
   * optimises the camera alone, utils/slam_frontend.py:364-392), grad_tau is still produced. */
  float* grad_means3D;         /* [N,3] */
  float* grad_means2D;         /* [N,3]: (dL/dndc_x, dL/dndc_y, 0) - densification statistic,
                                  gaussian_model.py:693-697 */
  float* grad_colors;          /* [N,K,3] when shs, else [N,3] */
  float* grad_opacities;       /* [N] */
  float* grad_scales;          /* [N,3] or NULL */
  float* grad_rotations;       /* [N,4] or NULL */
  float* grad_cov3D;           /* [N,6] or NULL */
  float* grad_tau;             /* [6] = [rho(3); theta(3)]: cam_trans_delta, cam_rot_delta
                                  (pose_utils.py:88-92) */
  /* sketched pose Jacobian (slam_frontend.py:269-338,654-669); sketch_mode 0 = off */
  int32_t sketch_mode;
  int32_t sketch_dim;
  int32_t stack_dim;
  const int32_t* sketch_indices;  /* [stack,H,W] slice for this repeat, bucket id or -1 */
  float* grad_sketch_dtau;        /* [stack,sketch,6] */
  void* sketch_ws;                /* sketch_bytes of scratch (sketch_mode != 0) */
  /* compact alternative to sketch_indices (the buckets of one repeat are disjoint, so one
   * int per pixel suffices): [H,W] = stack * sketch_dim + bucket, or -1.  Used when
   * sketch_indices is NULL. */
  const int32_t* sketch_bucket_flat;
  /* mapping mode (HOST pointer, or NULL): the seven per-Gaussian gradient pointers above must
   * then be NULL, scales / rotations / opacities / shs of `fwd` must be the outputs of
   * mgs_map_activate, and the gradients are chained and accumulated as described above. */
  const mgs_map_accum_args* map_accum;
  /* How the backward treats the field-of-view clamp of the EWA Jacobian, t.x = clamp(x/z) * z
   * (only splats whose centre lies outside 1.3x the field of view are affected; the forward is the
   * same): MGS_CLAMP_GRAD_UPSTREAM = 0 (the default of a zero-initialised block) = t.x held constant
   * in z and its x-gradient zeroed when clamped - what the public CUDA lineage of the absent
   * extension is believed to do, i.e. the gradients the north star's tolerance refers to;
   * MGS_CLAMP_GRAD_EXACT = 1 = the exact derivative of the forward (what plain autograd of the
   * forward gives).  DESIGN.md §2 quantifies the difference.  (ABI 5 had the two values swapped.) */
#define MGS_CLAMP_GRAD_UPSTREAM 0
#define MGS_CLAMP_GRAD_EXACT 1
  int32_t clamp_gradient_mode;
  /* An UPPER BOUND of the forward's pair count D when the caller knows one (the autograd binding has read the
   * exact D from the pinned slot before the backward is enqueued), 0 = none.  Sizes the grid of the blend
   * backward: D / 32 + tiles work items instead of pair_capacity / 32 + tiles.  A value BELOW the true count
   * loses work items - the native entry points, which cannot know D of the forward they have just enqueued,
   * leave it 0.  The kernel notices: counters[2] of the geom workspace (int32 at off_counters + 8, zeroed by every
   * forward) is set to the item count by a backward whose grid did not cover the forward's items - its gradients
   * are then incomplete (the Python binding passes the exact D and checks the word when raster_settings.debug). */
  int32_t pair_count_bound;
} mgs_backward_args;

int32_t mgs_abi_version(void);
/* sizeof() of the argument structs, in declaration order (0 = mgs_raster_shape,
 * 1 = mgs_workspace_sizes, 2 = mgs_forward_args, 3 = mgs_backward_args, 4 = mgs_pose_adam_args,
 * 5 = mgs_mapping_loss_args, 6 = mgs_lm_step_args, 7 = mgs_tracking_loss_args,
 * 8 = mgs_tracking_iter_args, 9 = mgs_sketch_residual_args, 10 = mgs_tracking_so_args,
 * 11 = mgs_adam_group, 12 = mgs_map_plan_args, 13 = mgs_gather_tensor, 14 = mgs_map_gather_args,
 * 15 = mgs_map_accum_args, 16 = mgs_map_activate_args, 17 = mgs_mapping_view_args,
 * 18 = mgs_map_finish_args, 19 = mgs_map_append_args);
 * -1 for an unknown index.  Lets a foreign-language binding verify its struct mirrors. */
int32_t mgs_struct_size(int32_t which);
const char* mgs_status_string(int32_t status);

/* Sizes/offsets of the workspaces for `shape` (uses shape->pair_capacity). */
int32_t mgs_raster_workspace_query(const mgs_raster_shape* shape, mgs_workspace_sizes* out);

/* Forward, stage 1: projection + EWA splat + pair counting + tile scan.
 * Needs `geom` only.  On return (stream order) counters[0] in geom holds D, the
 * number of (tile, Gaussian) pairs; radii is final. */
int32_t mgs_raster_forward_project(const mgs_forward_args* args, void* stream);

/* Forward, stage 2: pair emission, per-tile depth sort, front-to-back blend.
 * Safe for any pair_capacity; results are complete iff D <= pair_capacity. */
int32_t mgs_raster_forward_blend(const mgs_forward_args* args, void* stream);

/* Backward of the whole rasteriser (re-entrant: reads geom/bins, never writes them).
 * MGS_ERR_UNSUPPORTED when shape.pair_capacity * 40 B exceeds 2 GiB. */
int32_t mgs_raster_backward(const mgs_backward_args* args, void* stream);

/* simple-knn: out[i] = mean of the 3 smallest squared distances from points[i] to
 * the other points (exact).  `scratch` must hold mgs_knn_scratch_bytes(P) bytes. */
uint64_t mgs_knn_scratch_bytes(int32_t num_points);
int32_t mgs_knn_dist2(const float* points, int32_t num_points, float* out, void* scratch,
                      void* stream);

/* ---- tracking-loop glue --------------------------------------- */

/* torch.optim.Adam step (betas, eps, bias correction as PyTorch; no weight decay) on the
 * four per-camera parameter groups of utils/slam_frontend.py:364-392, then update_pose
 * (utils/pose_utils.py:88-98): T <- Exp([trans_delta; rot_delta]) T, deltas zeroed,
 * *converged = |tau| < threshold.  State order in exp_avg / exp_avg_sq: rot(3), trans(3),
 * a, b.  A NULL grad skips its group; T == NULL skips the pose update. */
typedef struct mgs_pose_adam_args {
  float* cam_rot_delta;        /* [3] */
  float* cam_trans_delta;      /* [3] */
  float* exposure_a;           /* [1] or NULL */
  float* exposure_b;           /* [1] or NULL */
  const float* grad_rot;
  const float* grad_trans;
  const float* grad_a;
  const float* grad_b;
  float* exp_avg;              /* [8] */
  float* exp_avg_sq;           /* [8] */
  float* T;                    /* [4,4] row-major world-to-camera, updated in place, or NULL */
  int32_t* converged;          /* out, or NULL */
  int32_t step;                /* 1-based */
  float lr_rot, lr_trans, lr_a, lr_b;
  float beta1, beta2, eps;
  float converged_threshold;
  /* optional fusions (NULL / 0 = off), used by mgs_tracking_iteration to save launches:
   * the block partials of dL/dtau ([n,6], instead of grad_trans / grad_rot) and of the exposure
   * gradients ([2,n]: d/da then d/db, instead of grad_a / grad_b) are summed here in index
   * order, and the camera matrices of the UPDATED pose are written (viewmatrix = T^T,
   * projmatrix = viewmatrix @ projection). */
  const float* tau_partials;
  int32_t num_tau_partials;
  const float* exposure_partials;
  int32_t num_exposure_partials;
  const float* projection;
  float* viewmatrix_out;
  float* projmatrix_out;
  /* mgs_mapping_view_iteration extras (all 0 / NULL otherwise).  A NULL parameter pointer
   * (cam_rot_delta, cam_trans_delta, exposure_a, exposure_b) skips that group even when its
   * gradient is available.  no_pose_update != 0: step the deltas but leave T alone (the reference
   * optimises the deltas of `frames_to_optimize` keyframes and applies update_pose to the first
   * `pose_window` only, utils/slam_backend.py:452-474 vs :328-332).  loss_partials: [2,n] block sums of
   * the mapping objective (colour, depth); loss = w_rgb * sum_c + w_depth * sum_d is stored to
   * *loss_view and added to *loss_accum. */
  int32_t no_pose_update;
  const float* loss_partials;
  int32_t num_loss_partials;
  float loss_w_rgb, loss_w_depth;
  float* loss_view;
  float* loss_accum;
  /* loss_norm_mode == 1 (mgs_tracking_iteration): loss_partials are [n] block sums of the squared
   * tracking residuals written by mgs_tracking_loss_onepass together with UN-normalised gradients
   * (the objective is the norm sqrt(sum h^2), its gradient carries 1 / loss; everything between
   * the loss and this kernel is linear in the image gradient).  This kernel forms loss = sqrt(sum),
   * scales every gradient it sums by (*loss_grad_out or 1) / loss before the Adam step and stores
   * loss and 1 / loss to loss_view[0], loss_view[1]. */
  const float* loss_grad_out;
  int32_t loss_norm_mode;
  /* != 0: once *converged is set the kernel does nothing at all (no step, no best-iterate update):
   * the reference leaves its loop at the first converged iteration (utils/slam_frontend.py:623-626);
   * a caller that reads the flag only every few iterations gets the same result. */
  int32_t sticky_converged;
  /* Best-iterate bookkeeping of the tracking loop (utils/slam_frontend.py:423-425, 510, 523-528):
   * l1_partials = [n] block sums of |residual| (UN-Hubered, the reference's
   * loss_tracking_scalar = ||loss_tracking_img||_1) of the render this iteration started from;
   * when their sum is below best[0] the state that was rendered (T, exposure_a, exposure_b BEFORE
   * this step - what TempCamera(viewpoint) copies) is stored: best = float[MGS_TRACK_BEST_FLOATS]
   * {best L1 (caller sets +inf), T[16], a, b, index of the best iteration, iteration counter,
   *  [21] the criterion (L1) of the LAST iteration's render, [22] |step| of the last iteration (|tau| applied by
   *  update_pose, or |x| of the LM solve) - a per-iteration trace for tests and logging, [23] spare}. */
  const float* l1_partials;
  float* best;
  int32_t num_l1_partials;
  /* loss_norm_mode == 1: the objective is the p-norm (sum |h|^p)^(1/p) of utils/slam_frontend.py:596-600
   * (p = 2 with Huber, RGN.pnorm without; configs/mono/tum/base_config.yaml:249); loss_partials are block
   * sums of |h|^p, loss = sum^(1/p) and the gradients are scaled by loss^(1-p).  <= 0 means 2. */
  float loss_pnorm;
} mgs_pose_adam_args;
#define MGS_TRACK_BEST_FLOATS 24

int32_t mgs_pose_adam_step(const mgs_pose_adam_args* args, void* stream);

/* Mapping objective (utils/slam_utils.py:224-253):
 *   loss = w_rgb * mean_{3HW} | m ((|a|+eps) image + b - gt) | + w_depth * mean_{HW} | dm (depth - gt_depth) |
 * m = mask (float 0/1, or NULL), dm = gt_depth > depth_mask_threshold (threshold < 0: no
 * mask).  Monocular: w_rgb = 1, w_depth = 0; RGB-D: w_rgb = alpha, w_depth = 1 - alpha.
 * apply_exposure = 0 is `initialization=True`.  `partial` holds
 * mgs_tracking_loss_partial_count(num_pixels) floats. */
typedef struct mgs_mapping_loss_args {
  const float* image;        /* [3,H,W] */
  const float* gt;           /* [3,H,W] */
  const float* mask;         /* [1,H,W] or NULL */
  const float* depth;        /* [1,H,W] or NULL (w_depth == 0) */
  const float* gt_depth;     /* [1,H,W] or NULL */
  const float* exposure_a;   /* [1] (apply_exposure != 0) */
  const float* exposure_b;   /* [1] */
  float exposure_eps;
  float w_rgb, w_depth;
  float depth_mask_threshold;
  int32_t apply_exposure;
  int64_t num_pixels;
  float* partial;
  float* loss;               /* [1] out */
  /* backward only */
  const float* grad_out;     /* [1] */
  float* grad_image;         /* [3,H,W] */
  float* grad_depth;         /* [1,H,W] or NULL */
  float* grad_a;             /* [1] or NULL */
  float* grad_b;             /* [1] or NULL */
  /* != 0: the caller guarantees that the int at partial[2 n] (n = partial count / 3) was zero when
   * the call was enqueued (e.g. a scratch zeroed once and reused: every call restores the zero).
   * mgs_mapping_loss_forward then sums the block partials in the workgroup that finishes last
   * instead of launching a second kernel. */
  int32_t partial_ticket_ready;
  int32_t reserved0;
} mgs_mapping_loss_args;

int32_t mgs_mapping_loss_forward(const mgs_mapping_loss_args* args, void* stream);
int32_t mgs_mapping_loss_backward(const mgs_mapping_loss_args* args, void* stream);

/* Sketched Levenberg-Marquardt step (utils/slam_frontend.py:672-697 + TempCamera.step
 * :49-53): solves (SJ^T SJ + lambda I) x = -SJ^T Sf for the 8 unknowns
 * [trans(3), rot(3), exposure_a, exposure_b] (identical to the reference's damped
 * torch.linalg.lstsq), writes x_out[8] and applies T <- Exp(x[:6]) T, exposure += x[6:8]
 * (T / exposure may be NULL to only solve). */
typedef struct mgs_lm_step_args {
  const float* SJ;      /* [rows, 8] */
  const float* Sf;      /* [rows] */
  int32_t rows;
  float lambda;         /* > 0 */
  float* T;             /* [4,4] or NULL */
  float* exposure_a;    /* [1] or NULL */
  float* exposure_b;    /* [1] or NULL */
  float* x_out;         /* [8] */
  /* optional extensions (all NULL / 0 for the plain solve) */
  const float* sj_tau;       /* [rows, 6]: used with sj_exposure when SJ == NULL */
  const float* sj_exposure;  /* [rows, 2] */
  float* lm_state;           /* [4] device-resident trust-region state {lambda, previous loss,
                                has_previous, converged}; when given, `lambda` is ignored and the
                                rule of utils/slam_frontend.py:536-545 runs on the device first:
                                loss < previous ? lambda = max(lambda / decrease, min_lambda)
                                                : lambda = min(lambda * increase, max_lambda) */
  const float* loss;         /* [1] current ||residual||_1 (lm_state only) */
  float increase_factor, decrease_factor, min_lambda, max_lambda;
  float converged_threshold; /* lm_state[3] = |x| < threshold (slam_frontend.py:699) */
  /* With lm_state: a converged step is NOT applied and every later call is a no-op (the reference
   * assigns new_viewpoint_params at slam_frontend.py:690-691 but only APPLIES it at the top of the next
   * iteration, :474-479, and the `break` of :699-706 comes first).  A non-converged step is applied at once
   * here: equal to the reference for every iteration but the LAST of a frame, whose step the reference
   * never applies - a caller that ends a frame on the last rendered state (use_best_loss off) restores it
   * (NativeTracker.run does).
   * best (or NULL): same float[MGS_TRACK_BEST_FLOATS] block as mgs_pose_adam_args.best; *loss is the
   * criterion, the pose / exposure this iteration rendered (before the step) is what is stored. */
  int32_t reserved0;
  float* best;
  /* optional (all three or none): the camera matrices of the STEPPED pose are written here (viewmatrix = T^T,
   * projmatrix = viewmatrix @ projection) - mgs_tracking_iteration_second_order then needs no
   * mgs_camera_from_pose launch in the next iteration (base.camera_matrices_valid) */
  const float* projection;
  float* viewmatrix_out;
  float* projmatrix_out;
  /* optional: zero_count floats at zero_after are set to 0 once every row has been read (the sketch
   * accumulators of the native second-order iteration: their consumer leaves them ready for the next one) */
  float* zero_after;
  int32_t zero_count;
  int32_t reserved1;
} mgs_lm_step_args;

int32_t mgs_lm_solve_step(const mgs_lm_step_args* args, void* stream);

/* Monocular tracking objective (utils/slam_utils.py:188-205, :58-75; norm at
 * utils/slam_frontend.py:596-598):
 *   loss = || Huber_delta( opacity * mask * ((|a| + eps) * image + b - gt) ) ||_p   (p = pnorm, below)
 * huber_delta <= 0 disables Huber.  `partial` holds mgs_tracking_loss_partial_count floats
 * (four per reduction block: forward sums, the two exposure-gradient sums, and - onepass only -
 * the block sums of |residual| before Huber),
 * `scalars` 2 floats ([0] = loss, [1] = 1/loss) written by forward and read by backward. */
typedef struct mgs_tracking_loss_args {
  const float* image;          /* [3,H,W] */
  const float* opacity;        /* [1,H,W] */
  const float* gt;             /* [3,H,W] */
  const float* mask;           /* [1,H,W] as float 0/1, or NULL */
  const float* exposure_a;     /* [1] */
  const float* exposure_b;     /* [1] */
  float exposure_eps;
  float huber_delta;
  int64_t num_pixels;          /* H*W */
  float* partial;
  float* scalars;
  /* backward only */
  const float* grad_out;       /* [1] dL/dloss */
  float* grad_image;           /* [3,H,W] */
  float* grad_a;               /* [1] */
  float* grad_b;               /* [1] */
  /* p of the norm, utils/slam_frontend.py:596-600: the reference uses p = 2 when use_huber is on and
   * RGN.pnorm (configs/mono/tum/base_config.yaml:249: 1) when it is off.  <= 0 means 2 (a zero-initialised
   * block keeps the Hubered L2 objective).  p = 1 and p = 2 are closed forms; any other p >= 1 goes
   * through powf.  `partial` then holds block sums of |h|^p, scalars[0] = loss = (sum)^(1/p) and
   * scalars[1] = loss^(1-p), the factor of the norm's derivative. */
  float pnorm;
  int32_t reserved0;
} mgs_tracking_loss_args;

int32_t mgs_tracking_loss_partial_count(int64_t num_pixels);
int32_t mgs_tracking_loss_forward(const mgs_tracking_loss_args* args, void* stream);
int32_t mgs_tracking_loss_backward(const mgs_tracking_loss_args* args, void* stream);
/* Forward + backward in two launches (no finish kernels): every workgroup of the backward sums
 * the forward's block sums itself; scalars[0] = loss; the exposure-gradient block partials are
 * left at partial[n .. 3n) ([2, n]: d/da then d/db, n = *num_blocks_out) for a consumer that
 * sums them (mgs_pose_adam_step: exposure_partials).  grad_a / grad_b are not written. */
int32_t mgs_tracking_loss_fused(const mgs_tracking_loss_args* args, int32_t* num_blocks_out, void* stream);
/* ONE launch: block sums of the squared residuals ([n] at partial), UN-normalised image gradient
 * (as if loss were 1) and un-normalised exposure partials ([2,n] at partial + n).  The consumer
 * (mgs_pose_adam_step with loss_norm_mode = 1) applies grad_out / loss.  args->scalars is not
 * written here.  partial + 3n: [n] block sums of |residual| BEFORE Huber (the best-iterate criterion
 * of utils/slam_frontend.py:510).  Replaces the two launches of mgs_tracking_loss_fused inside
 * mgs_tracking_iteration (reference: utils/slam_utils.py:188-217 get_loss_tracking_rgb + autograd). */
int32_t mgs_tracking_loss_onepass(const mgs_tracking_loss_args* args, int32_t* num_blocks_out, void* stream);


/* ---- native tracking iteration (row a12: utils/slam_frontend.py:493-630) ---------------- */

/* Camera matrices of utils/camera_utils.py:94-108 from the world-to-camera pose:
 * viewmatrix = T^T, projmatrix = viewmatrix @ projection (projection as stored by the
 * reference, i.e. already transposed, camera_utils.py:86-89).  campos is the reference's
 * camera_center, which IS world_view_transform (:106-108): pass the viewmatrix buffer. */
int32_t mgs_camera_from_pose(const float* T, const float* projection, float* viewmatrix,
                             float* projmatrix, void* stream);

/* One first-order monocular tracking iteration, enqueued as a fixed launch sequence with
 * no host round trip (the Python loop body costs ~1 ms of host time per iteration):
 *   camera matrices from T -> rasteriser forward (project + blend at the caller's fixed
 *   pair capacity); the blend pass evaluates the tracking objective in its epilogue (the arithmetic of
 *   mgs_tracking_loss_onepass: d loss / d image into grad_image, the four sums as one partial per 8x8-pixel
 *   quadrant in the geom workspace - loss.partial is not used by this entry) -> pose-only rasteriser
 *   backward -> Adam on (rot, trans, exposure a, b) + update_pose (mgs_pose_adam_step), with
 *   the small reduction kernels and the 1 / loss of the norm folded into their consumers (9 launches).
 * fwd.viewmatrix / fwd.projmatrix / fwd.campos must point at caller-owned device buffers
 * (16/16/>=3 floats; campos may alias viewmatrix) that this call REWRITES from T;
 * fwd.projmatrix_raw is the projection.  The forward is complete iff counters[0] (pair
 * count D, in geom) <= shape.pair_capacity: the caller checks that lazily.  Per-iteration
 * results: loss_scalars[0] = loss, *adam.converged, the updated T / exposure. */
typedef struct mgs_tracking_iter_args {
  mgs_forward_args fwd;
  void* bwd;                  /* bwd_bytes of backward scratch */
  float* grad_image;          /* [3,H,W] scratch: dL/d(render) */
  float* grad_tau;            /* [6] scratch: [rho; theta] */
  float* grad_exposure;       /* [2] scratch: d/da, d/db */
  const float* one;           /* [1] = 1.0f (dL/dloss) */
  mgs_tracking_loss_args loss;   /* image/opacity/grad_* fields are filled in by the call */
  mgs_pose_adam_args adam;       /* grad_* fields are filled in by the call; T must be set */
  int32_t camera_matrices_valid; /* != 0: fwd.viewmatrix / projmatrix already match T (every
                                    mgs_tracking_iteration leaves them so): skip that launch */
  int32_t reserved0;
  float* best;                /* float[MGS_TRACK_BEST_FLOATS] best-iterate block (see mgs_pose_adam_args), or
                                 NULL; shared by the first- and second-order iterations of a frame */
} mgs_tracking_iter_args;

int32_t mgs_tracking_iteration(const mgs_tracking_iter_args* args, void* stream);



/* ---- native second-order (sketched Levenberg-Marquardt) tracking iteration ----------------
 * utils/slam_frontend.py:455-710 with in_second_order = True.
 *
 * mgs_sketch_assign: the CountSketch bookkeeping of :269-338 as one kernel.  The reference
 * draws torch.randperm(H*W) and cuts its first chunk*stack*sketch entries (chunk =
 * H*W / (stack*sketch)) into disjoint buckets; here pixel p gets position q = pi_key(p) of a
 * keyed pseudo-random PERMUTATION of [0, H*W) (invertible multiply / xor-shift rounds on
 * ceil(log2 HW) bits with cycle walking), bucket[p] = q / chunk if q < chunk*stack*sketch else
 * -1, and weights[p] = +-1 from a hash bit.  Same structure (disjoint buckets of exactly
 * `chunk` pixels), no sort, no index tensors. */
int32_t mgs_sketch_assign(int64_t num_pixels, int32_t stack_dim, int32_t sketch_dim, uint64_t key,
                          int32_t* bucket, float* weights, void* stream);

/* Sketched monocular tracking residual (utils/slam_utils.py:188-205 + Huber :58-75 + the
 * bucket sums of slam_frontend.py:636-650 and ApplyExposure's sketched Jacobian
 * slam_utils.py:150-185), one pass over the image:
 *   r_c = opacity * mask * ((|a|+eps) image_c + b - gt_c);  l1 = sum |r_c|
 *   weighted_p = weights_p / (HW / (stack*sketch)) * sum_c Huber(r_c)
 *   Sf[b] += weighted_p;  sj_exposure[b] += d weighted_p / d(a, b);  grad_image = d weighted_p / d image
 * with the derivatives through the exposure taken as the reference's hand-written ApplyExposure.backward takes
 * them (slam_utils.py:145-149; NOT the exact derivative): d/d image carries |a| without eps, d/da carries
 * image WITHOUT sign(a) - identical while exposure_a > 0, pinned for a < 0 by tests/golden/map_update_ref.npz.
 * Sf / sj_exposure / l1 must be zero on entry (they are accumulated with atomics). */
typedef struct mgs_sketch_residual_args {
  const float* image;        /* [3,H,W] */
  const float* opacity;      /* [1,H,W] */
  const float* gt;           /* [3,H,W] */
  const float* mask;         /* [1,H,W] float 0/1 or NULL */
  const float* exposure_a;
  const float* exposure_b;
  float exposure_eps, huber_delta;
  int64_t num_pixels;
  int32_t stack_dim, sketch_dim;
  const int32_t* bucket;     /* from mgs_sketch_assign */
  const float* weights;
  float* grad_image;         /* [3,H,W] out */
  float* Sf;                 /* [stack*sketch] accumulated */
  float* sj_exposure;        /* [stack*sketch, 2] accumulated */
  float* l1;                 /* [1] accumulated */
  /* assign != 0: the partition of mgs_sketch_assign(num_pixels, stack_dim, sketch_dim, assign_key) is
   * evaluated inside this pass and WRITTEN to bucket / weights (which then are outputs): one launch less */
  int32_t assign;
  int32_t reserved0;
  uint64_t assign_key;
} mgs_sketch_residual_args;

int32_t mgs_sketch_residual(const mgs_sketch_residual_args* args, void* stream);

/* One second-order iteration as a fixed launch sequence (no host round trip): camera
 * matrices from T (skipped when base.camera_matrices_valid: the previous iteration's LM kernel wrote them) ->
 * forward -> zero accumulators -> mgs_sketch_residual with the partition of mgs_sketch_assign(key)
 * evaluated inside it -> Jacobian-only backward in sketch mode (grad_sketch_dtau via the compact
 * bucket map; the per-splat sums, the per-Gaussian chain and grad_tau are skipped: the LM step
 * consumes the sketched Jacobian alone) -> mgs_lm_solve_step with the device-resident
 * trust-region state.
 * `base` as for mgs_tracking_iteration (its adam block is unused except T / exposure);
 * `accum` holds repeat*stack*sketch*9 + 4 floats: Sf[R d] | sj_exposure[R d,2] | sj_tau[R d,6] | l1, pad.
 * repeat_dim (RGN.second_order.repeat_dim, configs/mono/tum/base_config.yaml:258; utils/slam_frontend.py:654-669):
 * R backward passes over ONE render, each with its own partition (key + r * golden ratio) and weights; the rows of
 * Sf / SJ are stacked ([R d] rows in the solve).  `bucket` / `weights` then hold R planes of H*W.
 * If a step of the sequence fails after the accumulators were touched, they are cleared before the error is
 * returned, so scratch_kept_zero survives a failed call. */
typedef struct mgs_tracking_so_args {
  mgs_tracking_iter_args base;
  int32_t stack_dim, sketch_dim;
  uint64_t key;              /* changes every iteration */
  int32_t* bucket;           /* [repeat, H*W] scratch */
  float* weights;            /* [repeat, H*W] scratch */
  float* accum;
  void* sketch_ws;           /* sketch_bytes of backward scratch */
  mgs_lm_step_args lm;       /* SJ / Sf / sj_* / loss fields are filled in by the call */
  /* != 0: `accum` was zero-filled before the FIRST call and nobody else writes it; the LM kernel - its consumer -
   * restores the zeros, so no hipMemsetAsync is enqueued per iteration.  (`sketch_ws` needs no initial state:
   * every launch rewrites what it reads - the per-run slabs of Jacobian rows and their masks.) */
  int32_t scratch_kept_zero;
  int32_t repeat_dim;        /* >= 1; 0 means 1 */
} mgs_tracking_so_args;

int32_t mgs_tracking_iteration_second_order(const mgs_tracking_so_args* args, void* stream);

/* ---- map maintenance on the device ---------------------------------- */

#define MGS_ADAM_MAX_GROUPS 8
#define MGS_GATHER_MAX_TENSORS 24

/* torch.optim.Adam step (betas / eps / bias correction as PyTorch, no weight decay, no
 * amsgrad) over all parameter groups of GaussianModel.training_setup
 * (gaussian_splatting/scene/gaussian_model.py:252-285; stepped at utils/slam_backend.py:142,
 * 322,365) in one launch. */
typedef struct mgs_adam_group {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
  float lr;
  int32_t step;          /* 1-based step count of THIS tensor (PyTorch keeps it per parameter:
                            a tensor without a gradient does not advance) */
} mgs_adam_group;

/* betas / eps are doubles so that 1 - beta rounds to fp32 exactly as in PyTorch (which
 * evaluates it in Python floats): with a float beta2 = 0.999f, 1 - beta2 is off by 1.3e-5. */
int32_t mgs_adam_step_multi(const mgs_adam_group* groups, int32_t num_groups, double beta1,
                            double beta2, double eps, void* stream);

/* Keyframe-parallel mapping: one launch packs this rank's parameter gradients
 * and the densification statistics of its view (gaussian_model.py:693-697:
 * ||means2D.grad[:, :2]|| where radii > 0, and the visibility count) into the flat buffer that
 * is all-reduced over RCCL: flat = [grad_0 | ... | grad_{k-1} | grad-norm stat [N] | visible [N]].
 * radii is copied to radii_out (all-reduced with max). */
int32_t mgs_pack_mapping_grads(const float* const* grads, const int64_t* numels, int32_t num_grads,
                               const float* means2D_grad, const int32_t* radii,
                               int64_t num_gaussians, float* flat, int32_t* radii_out, void* stream);

/* Rebuild plan of GaussianModel.densify_and_prune (gaussian_model.py:674-691 =
 * densify_and_clone :636-672, densify_and_split :598-634, prune_points :540-556) or, with
 * prune_mask != NULL, of prune_points(mask) alone.  Rows of the rebuilt arrays, in the
 * reference's order: surviving originals, surviving clones, first split children, second split
 * children (each group in index order - stable).
 *   mgs_map_plan_count: per-Gaussian decisions -> flags, totals[0..3] = surviving originals,
 *                       clones, split parents (so rows = t0 + t1 + 2 * t2) and t3 = ALL Gaussians
 *                       selected for splitting (the reference draws 2 * t3 random offsets at :608-609
 *                       before the final prune removes some children);
 *   mgs_map_plan_emit:  src_index[row] = parent | kind << 30 (kind 0 original, 1 clone,
 *                       2 / 3 first / second child), rows entries; noise_row[k] = row of the
 *                       [2 * t3, 3] noise tensor used by the k-th child row, 2 * t2 entries.
 * The caller reads totals between the two calls to allocate (one host sync per rebuild). */
typedef struct mgs_map_plan_args {
  int32_t n;
  /* densify_and_prune inputs (ignored when prune_mask is given) */
  const float* grad_accum;      /* [n] xyz_gradient_accum */
  const float* denom;           /* [n] */
  const float* log_scales;      /* [n,3] _scaling */
  const float* opacity_logit;   /* [n]   _opacity */
  float grad_threshold;         /* max_grad */
  float dense_extent;           /* percent_dense * extent */
  float min_opacity;
  float big_extent;             /* 0.1 * extent when max_screen_size is set, else <= 0 */
  const uint8_t* prune_mask;    /* [n] 1 = remove, or NULL */
  /* scratch / outputs */
  uint8_t* flags;               /* [n] */
  int32_t* block_counts;        /* [4 * mgs_map_plan_blocks(n)] */
  int32_t* totals;              /* [4] */
  uint32_t* src_index;          /* [rows], mgs_map_plan_emit only */
  int32_t* noise_row;           /* [max(1, 2 * totals[2])], mgs_map_plan_emit only */
} mgs_map_plan_args;

int32_t mgs_map_plan_blocks(int32_t n);
int32_t mgs_map_plan_count(const mgs_map_plan_args* args, void* stream);
int32_t mgs_map_plan_emit(const mgs_map_plan_args* args, void* stream);

/* Rebuild per-Gaussian tensors (32-bit elements, `width` per row) from a plan in one launch:
 * dst[row, :] = f(src[parent(row), :]). */
enum {
  MGS_GATHER_COPY = 0,           /* every row copies its parent (features, opacity, rotation, ids) */
  MGS_GATHER_ZERO_NEW = 1,       /* originals copy, clones / children get 0 (Adam moments, :560-577) */
  MGS_GATHER_SPLIT_SCALING = 2,  /* children: log(exp(s) / 1.6)  (:615-617), float */
  MGS_GATHER_SPLIT_XYZ = 3       /* children: xyz + R(q) (noise * exp(s))  (:609-614), float, width 3 */
};

typedef struct mgs_gather_tensor {
  const void* src;
  void* dst;
  int32_t width;
  int32_t mode;
} mgs_gather_tensor;

typedef struct mgs_map_gather_args {
  mgs_gather_tensor tensors[MGS_GATHER_MAX_TENSORS];
  int32_t num_tensors;
  int64_t rows;                 /* rows of the rebuilt arrays */
  int32_t num_children;         /* totals[2]; the last 2 * num_children rows are split children */
  const uint32_t* src_index;
  const float* rotations;       /* parents' _rotation [n,4]   (MGS_GATHER_SPLIT_XYZ) */
  const float* log_scales;      /* parents' _scaling [n,3]    (MGS_GATHER_SPLIT_XYZ) */
  const float* noise;           /* [2 * totals[3], 3] unit normals */
  const int32_t* noise_row;     /* from mgs_map_plan_emit */
} mgs_map_gather_args;

int32_t mgs_map_gather(const mgs_map_gather_args* args, void* stream);

/* ---- native mapping iteration (row a13: utils/slam_backend.py:171-332) -------------------- */

/* GaussianModel's activations for one mapping iteration (gaussian_model.py:54-62,77-102), one
 * launch: scales = exp(_scaling) (an isotropic [N,1] model is broadcast to 3 axes as the
 * renderer does, gaussian_renderer/__init__.py:92-95), rotations = normalize(_rotation),
 * opacities = sigmoid(_opacity), shs = cat(_features_dc, _features_rest).  `shs` may be NULL
 * when K == 1: features_dc is then used in place. */
typedef struct mgs_map_activate_args {
  int32_t num_gaussians;
  int32_t scale_dims;            /* 3 or 1 */
  int32_t sh_coeffs;             /* K */
  const float* log_scales;       /* [N,scale_dims] */
  const float* raw_rotations;    /* [N,4] */
  const float* opacity_logits;   /* [N] */
  const float* features_dc;      /* [N,1,3] */
  const float* features_rest;    /* [N,K-1,3] or NULL */
  float* scales;                 /* out [N,3] */
  float* rotations;              /* out [N,4] */
  float* opacities;              /* out [N] */
  float* shs;                    /* out [N,K,3] or NULL (K == 1) */
} mgs_map_activate_args;

int32_t mgs_map_activate(const mgs_map_activate_args* args, void* stream);

/* One view of one mapping iteration as a fixed launch sequence with no host round trip
 * (the body of the loops at utils/slam_backend.py:183-242 plus this view's share of :247-332):
 *   camera matrices from T -> rasteriser forward at the caller's fixed pair capacity ->
 *   mapping objective (utils/slam_utils.py:224-253) value + gradients in one pass ->
 *   rasteriser backward in mapping mode (mgs_map_accum_args: gradients chained through the
 *   activations and ACCUMULATED over the views, densification statistics, occ-aware visibility)
 *   -> Adam on this view's (cam_rot_delta, cam_trans_delta, exposure_a, exposure_b) + update_pose
 *   (mgs_pose_adam_args; the keyframe optimiser of :452-489 is per view, so its step commutes
 *   with the other views).
 * forward_only != 0 stops after the forward and only writes accum.visibility (the prune pass of
 * :259-290 consumes nothing else).  loss.image / depth / grad_* / partial and adam.grad_* /
 * *_partials are filled in by the call; loss.partial must hold
 * mgs_mapping_loss_partial_count(HW) floats. */
typedef struct mgs_mapping_view_args {
  mgs_forward_args fwd;
  void* bwd;                     /* bwd_bytes of backward scratch */
  float* grad_image;             /* [3,H,W] scratch */
  float* grad_depth;             /* [1,H,W] scratch, or NULL when loss.w_depth == 0 */
  float* grad_tau;               /* [6] scratch */
  mgs_mapping_loss_args loss;
  mgs_pose_adam_args adam;
  mgs_map_accum_args accum;
  float* loss_view;              /* [1] out: this view's loss, or NULL */
  float* loss_accum;             /* [1] += this view's loss, or NULL */
  int32_t camera_matrices_valid;
  int32_t forward_only;
} mgs_mapping_view_args;

int32_t mgs_mapping_loss_partial_count(int64_t num_pixels);
/* Value and gradients of the mapping objective in one pass (upstream gradient 1 unless
 * grad_out is given): grad_image / grad_depth are written, `partial` receives the block sums
 * [4][*num_blocks_out] = colour residual, depth residual, d/da, d/db for a consumer that sums
 * them (mgs_pose_adam_step: loss_partials / exposure_partials).  With partial_ticket_ready != 0
 * (the int at partial[4 n] zero when enqueued; restored by the call) the workgroup that finishes
 * last also writes *loss, *grad_a, *grad_b (each optional). */
int32_t mgs_mapping_loss_fused(const mgs_mapping_loss_args* args, int32_t* num_blocks_out, void* stream);
int32_t mgs_mapping_view_iteration(const mgs_mapping_view_args* args, void* stream);

/* End of a mapping iteration (after the optional all-reduce of the flat buffer), one launch:
 *   xyz_gradient_accum += gradnorm_inc, denom += denom_inc, max_radii2D = max(max_radii2D, radii_max)
 * (slam_backend.py:292-299; max_radii2D is a float tensor in the reference), and optionally
 * GaussianModel.reset_opacity / reset_opacity_nonvisible (gaussian_model.py:364-377 +
 * replace_tensor_to_optimizer :470-483): opacity logit <- inverse_sigmoid(reset_value) for every
 * Gaussian (reset_mode 1) or for those NOT visible in any view of this iteration, i.e.
 * denom_inc == 0 (reset_mode 2, 3), and the opacity group's Adam moments zeroed (all of them,
 * as the reference does).  reset_mode 2 is the reference to the letter: gaussian_model.py:375 stores
 * the ACTIVATED opacity of a visible Gaussian as its new raw parameter, so its logit l becomes
 * sigmoid(l) (pinned by tests/golden/map_update_ref.npz, generated from the reference's own code);
 * reset_mode 3 leaves the visible Gaussians' logits untouched (what the code presumably meant). */
typedef struct mgs_map_finish_args {
  int32_t num_gaussians;
  const float* gradnorm_inc;     /* [N] or NULL (no statistics this iteration) */
  const float* denom_inc;        /* [N] (also the visibility source of reset_mode 2) */
  const int32_t* radii_max;      /* [N] */
  float* xyz_gradient_accum;     /* [N] */
  float* denom;                  /* [N] */
  float* max_radii2D;            /* [N] */
  int32_t reset_mode;            /* 0 none, 1 reset_opacity, 2 reset_opacity_nonvisible (reference), 3 same, visible logits kept */
  float reset_value;             /* 0.01 / 0.4 */
  float* opacity_logits;         /* [N] (reset_mode != 0) */
  float* opacity_exp_avg;        /* [N] or NULL */
  float* opacity_exp_avg_sq;     /* [N] or NULL */
} mgs_map_finish_args;

int32_t mgs_map_finish_iteration(const mgs_map_finish_args* args, void* stream);

/* GaussianModel.extend_from_pcd (gaussian_model.py:210-245 -> cat_tensors_to_optimizer :525-557,
 * densification_postfix :559-593) in one launch: dst[t] = cat(old[t], new[t]) for up to
 * MGS_GATHER_MAX_TENSORS row-major 32-bit tensors; new == NULL appends zeros (Adam moments). */
typedef struct mgs_map_append_args {
  mgs_gather_tensor tensors[MGS_GATHER_MAX_TENSORS];   /* src = old rows, dst = rebuilt rows, mode unused */
  const void* new_rows[MGS_GATHER_MAX_TENSORS];        /* [rows_new, width] or NULL = zeros */
  int32_t num_tensors;
  int64_t rows_old, rows_new;
} mgs_map_append_args;

int32_t mgs_map_append(const mgs_map_append_args* args, void* stream);

/* Per-kernel timing (diagnostics; used by bench.py for the roofline line).  While
 * enabled every kernel launch is bracketed by hipEvents on the launch stream.
 * mgs_profile_read waits for the recorded events, aggregates them by kernel name into
 * names[k*32 .. k*32+31] / total_ms[k] / launches[k], clears the log and returns the
 * number of distinct names (<= max_entries), or a negative status. */
int32_t mgs_profile_enable(int32_t on);
int32_t mgs_profile_read(int32_t max_entries, char* names, float* total_ms, int32_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* MONOGS_RASTER_H */
