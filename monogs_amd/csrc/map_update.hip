// Map maintenance on the device: the Gaussian optimiser step and
// densify / prune as a handful of launches instead of the reference's per-tensor boolean
// indexing, torch.cat and foreach-Adam kernels.
//
//   k_adam_multi     torch.optim.Adam step (no weight decay, no amsgrad) over up to 8
//                    parameter groups in ONE launch        gaussian_model.py:252-285,
//                                                          slam_backend.py:142,322,365
//   k_plan_count /   densify_and_prune's decisions per Gaussian (clone, split, final prune)
//   k_plan_scan  /   and the position of every surviving original / clone / split child in
//   k_plan_emit      the rebuilt arrays: wave ballot + popcount prefix sums, stable, in the
//                    reference's order [originals, clones, children copy 0, children copy 1]
//                                                          gaussian_model.py:598-691
//   k_gather_rows    rebuilds every per-Gaussian tensor (6 parameters, 12 Adam moments,
//                    keyframe ids, observation counts) from the plan in ONE launch, applying
//                    the split transform to xyz / scaling     gaussian_model.py:485-596
//
// All HBM-bound streaming kernels: coalesced along the destination, row-gathers on the source.
#include <hip/hip_runtime.h>

#include "../../include/monogs_raster.h"
#include "launch.h"

namespace mgs {

// ---------------------------------------------------------------------------------
struct AdamPack {
  float* p[MGS_ADAM_MAX_GROUPS];
  const float* g[MGS_ADAM_MAX_GROUPS];
  float* m[MGS_ADAM_MAX_GROUPS];
  float* v[MGS_ADAM_MAX_GROUPS];
  long long chunk_end[MGS_ADAM_MAX_GROUPS];   // inclusive scan of ceil(numel / 4)
  long long numel[MGS_ADAM_MAX_GROUPS];
  float step_size[MGS_ADAM_MAX_GROUPS];       // lr / (1 - beta1^t)
  float inv_bc2_sqrt[MGS_ADAM_MAX_GROUPS];    // 1 / sqrt(1 - beta2^t)
  int n;
  float beta1, beta2, eps;
  float om_beta1, om_beta2;                   // 1 - beta, rounded from double as PyTorch does
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamPack& A,
                                         float step_size, float inv_bc2_sqrt) {
  // torch/optim/adam.py (_single_tensor_adam): exp_avg.lerp_(grad, 1 - beta1);
  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2);
  // denom = (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps); param.addcdiv_(exp_avg, denom, -step_size)
  m = m + A.om_beta1 * (g - m);
  v = A.beta2 * v + A.om_beta2 * g * g;
  const float denom = sqrtf(v) * inv_bc2_sqrt + A.eps;
  p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(256) void k_adam_multi(AdamPack A, long long total_chunks, int vec) {
  for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < total_chunks; c += (long long)gridDim.x * 256) {
    int grp = 0;
    while (grp < A.n - 1 && c >= A.chunk_end[grp]) grp++;
    const long long c0 = grp ? A.chunk_end[grp - 1] : 0;
    const long long e = (c - c0) * 4;
    const long long left = A.numel[grp] - e;
    float* p = A.p[grp] + e;
    const float* g = A.g[grp] + e;
    float* m = A.m[grp] + e;
    float* v = A.v[grp] + e;
    const float ss = A.step_size[grp], ib = A.inv_bc2_sqrt[grp];
    if (vec && left >= 4) {
      float4 P = *reinterpret_cast<float4*>(p), M = *reinterpret_cast<float4*>(m), V = *reinterpret_cast<float4*>(v);
      const float4 G = *reinterpret_cast<const float4*>(g);
      adam_one(P.x, G.x, M.x, V.x, A, ss, ib);
      adam_one(P.y, G.y, M.y, V.y, A, ss, ib);
      adam_one(P.z, G.z, M.z, V.z, A, ss, ib);
      adam_one(P.w, G.w, M.w, V.w, A, ss, ib);
      *reinterpret_cast<float4*>(p) = P; *reinterpret_cast<float4*>(m) = M; *reinterpret_cast<float4*>(v) = V;
    } else {
      const int k = left < 4 ? (int)left : 4;
      for (int i = 0; i < k; i++) {
        float P = p[i], M = m[i], V = v[i];
        adam_one(P, g[i], M, V, A, ss, ib);
        p[i] = P; m[i] = M; v[i] = V;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// Rebuild plan.  flags per Gaussian: bit0 keep the original, bit1 keep its clone, bit2 keep
// its two split children, bit3 selected for splitting (the reference draws its random offsets
// for every selected Gaussian, before the final prune: the ordinal among the selected is the
// child's row in the noise tensor).
constexpr int kPlanBlock = 1024;   // Gaussians per workgroup (one per thread)

__device__ __forceinline__ unsigned int plan_flags(const mgs_map_plan_args& A, int i) {
  if (A.prune_mask) return A.prune_mask[i] ? 0u : 1u;          // prune_points(mask) only
  // densify_and_prune (gaussian_model.py:674-691)
  float grad = A.grad_accum[i] / A.denom[i];
  if (grad != grad) grad = 0.f;                                // grads[grads.isnan()] = 0
  const float s0 = expf(A.log_scales[3 * i]), s1 = expf(A.log_scales[3 * i + 1]), s2 = expf(A.log_scales[3 * i + 2]);
  const float smax = fmaxf(s0, fmaxf(s1, s2));
  const bool hot = grad >= A.grad_threshold;
  const bool clone = hot && smax <= A.dense_extent;            // :640-647
  const bool split = hot && smax > A.dense_extent;             // :600-607
  const float opacity = 1.f / (1.f + expf(-A.opacity_logit[i]));
  const bool low = opacity < A.min_opacity;                    // :680
  // max_radii2D was just zeroed by densification_postfix (:594), so the screen-size test of
  // :682 is always false here; the world-size test (:683) applies when max_screen_size is set
  const bool big = A.big_extent > 0.f && smax > A.big_extent;
  const bool big_child = A.big_extent > 0.f && smax / 1.6f > A.big_extent;
  unsigned int f = 0;
  if (!split && !low && !big) f |= 1u;
  if (clone && !low && !big) f |= 2u;
  if (split && !low && !big_child) f |= 4u;
  if (split) f |= 8u;
  return f;
}

__global__ __launch_bounds__(kPlanBlock) void k_plan_count(mgs_map_plan_args A) {
  __shared__ int s_cnt[4];
  if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * kPlanBlock + threadIdx.x;
  const unsigned int f = i < A.n ? plan_flags(A, i) : 0u;
  if (i < A.n) A.flags[i] = (unsigned char)f;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const unsigned long long m = __ballot((f >> b) & 1u);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_cnt[b], __popcll(m));
  }
  __syncthreads();
  if (threadIdx.x < 4) A.block_counts[blockIdx.x * 4 + threadIdx.x] = s_cnt[threadIdx.x];
}

// exclusive scan of the per-block counts (4 columns) by one workgroup; totals[0..3]
__global__ __launch_bounds__(1024) void k_plan_scan(mgs_map_plan_args A, int nblk) {
  __shared__ int s[1024];
  for (int col = 0; col < 4; col++) {
    int carry = 0;
    for (int base = 0; base < nblk; base += 1024) {
      const int b = base + threadIdx.x;
      const int v = b < nblk ? A.block_counts[b * 4 + col] : 0;
      s[threadIdx.x] = v;
      __syncthreads();
      for (int off = 1; off < 1024; off <<= 1) {
        const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
      }
      if (b < nblk) A.block_counts[b * 4 + col] = carry + s[threadIdx.x] - v;
      carry += s[1023];
      __syncthreads();
    }
    if (threadIdx.x == 0) A.totals[col] = carry;
  }
}

// src_index[j] = parent | kind << 30 for every row j of the rebuilt arrays
// (kind 0 original, 1 clone, 2 / 3 first / second split child)
__global__ __launch_bounds__(kPlanBlock) void k_plan_emit(mgs_map_plan_args A) {
  __shared__ int s_wave[4][kPlanBlock / 64];
  const int i = blockIdx.x * kPlanBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned int f = i < A.n ? A.flags[i] : 0u;
  int pre[4];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const unsigned long long m = __ballot((f >> b) & 1u);
    pre[b] = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[b][wave] = __popcll(m);
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < 4; b++)
    for (int w = 0; w < wave; w++) pre[b] += s_wave[b][w];
  const int n_orig = A.totals[0], n_clone = A.totals[1], n_child = A.totals[2];
  if (f & 1u) A.src_index[A.block_counts[blockIdx.x * 4] + pre[0]] = (unsigned int)i;
  if (f & 2u) A.src_index[n_orig + A.block_counts[blockIdx.x * 4 + 1] + pre[1]] = (unsigned int)i | (1u << 30);
  if (f & 4u) {
    const int c = A.block_counts[blockIdx.x * 4 + 2] + pre[2];
    A.src_index[n_orig + n_clone + c] = (unsigned int)i | (2u << 30);
    A.src_index[n_orig + n_clone + n_child + c] = (unsigned int)i | (3u << 30);
    const int ord = A.block_counts[blockIdx.x * 4 + 3] + pre[3];      // ordinal among the selected
    A.noise_row[c] = ord;
    A.noise_row[n_child + c] = A.totals[3] + ord;
  }
}

// ---------------------------------------------------------------------------------
struct GatherPack {
  const unsigned int* src[MGS_GATHER_MAX_TENSORS];
  unsigned int* dst[MGS_GATHER_MAX_TENSORS];
  int width[MGS_GATHER_MAX_TENSORS];
  int mode[MGS_GATHER_MAX_TENSORS];
  int n;
  const unsigned int* src_index;
  long long rows;
  // split transform (gaussian_model.py:609-620)
  const float* rot;          // parents' raw quaternions [N,4]
  const float* log_scales;   // parents' log scales [N,3]
  const float* noise;        // unit normals [2 * n_selected, 3]
  const int* noise_row;      // [2 * n_child] noise row of every child row
  long long child_base;      // first child row in the rebuilt arrays
};

__global__ __launch_bounds__(256) void k_gather_rows(GatherPack G) {
  const int t = blockIdx.y;
  if (t >= G.n) return;
  const int w = G.width[t], mode = G.mode[t];
  const long long total = G.rows * w;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long row = e / w;
    const int col = (int)(e - row * w);
    const unsigned int si = G.src_index[row];
    const unsigned int parent = si & 0x3fffffffu, kind = si >> 30;
    unsigned int val = G.src[t][(size_t)parent * w + col];
    if (mode == MGS_GATHER_ZERO_NEW) {
      if (kind != 0u) val = 0u;
    } else if (mode == MGS_GATHER_SPLIT_SCALING) {
      if (kind >= 2u) val = __float_as_uint(logf(expf(__uint_as_float(val)) / 1.6f));   // / (0.8 * N), N = 2
    } else if (mode == MGS_GATHER_SPLIT_XYZ) {
      if (kind >= 2u) {
        // new_xyz = R(q / |q|) (noise * scale) + xyz   (:609-614, build_rotation general_utils.py:114-137)
        const float4 qq = reinterpret_cast<const float4*>(G.rot)[parent];
        const float inv = 1.f / sqrtf(qq.x * qq.x + qq.y * qq.y + qq.z * qq.z + qq.w * qq.w);
        const float r = qq.x * inv, x = qq.y * inv, y = qq.z * inv, z = qq.w * inv;
        const float* nz = G.noise + (size_t)G.noise_row[row - G.child_base] * 3;
        const float s0 = nz[0] * expf(G.log_scales[3 * (size_t)parent]);
        const float s1 = nz[1] * expf(G.log_scales[3 * (size_t)parent + 1]);
        const float s2 = nz[2] * expf(G.log_scales[3 * (size_t)parent + 2]);
        float d;
        if (col == 0) d = (1.f - 2.f * (y * y + z * z)) * s0 + 2.f * (x * y - r * z) * s1 + 2.f * (x * z + r * y) * s2;
        else if (col == 1) d = 2.f * (x * y + r * z) * s0 + (1.f - 2.f * (x * x + z * z)) * s1 + 2.f * (y * z - r * x) * s2;
        else d = 2.f * (x * z - r * y) * s0 + 2.f * (y * z + r * x) * s1 + (1.f - 2.f * (x * x + y * y)) * s2;
        val = __float_as_uint(d + __uint_as_float(val));
      }
    }
    G.dst[t][e] = val;
  }
}

// ---------------------------------------------------------------------------------
// Keyframe-parallel mapping: pack this rank's gradients and densification
// statistics into the flat all-reduce buffer in ONE launch:
//   flat = [grad_0 | grad_1 | ... | ||means2D.grad[:, :2]|| masked by visibility | visibility]
// (gaussian_model.py:693-697: the statistics are formed per view BEFORE the reduction).
struct PackPack {
  const float* src[MGS_ADAM_MAX_GROUPS];
  long long end[MGS_ADAM_MAX_GROUPS];     // inclusive scan of numel
  int n;
  const float* means2D_grad;              // [N,3]
  const int* radii;                       // [N]
  int* radii_out;                         // [N]
  long long N;
  float* flat;
};

__global__ __launch_bounds__(256) void k_pack_grads(PackPack A) {
  const long long total = A.end[A.n - 1] + 2 * A.N;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    float v;
    if (e < A.end[A.n - 1]) {
      int grp = 0;
      while (grp < A.n - 1 && e >= A.end[grp]) grp++;
      v = A.src[grp][e - (grp ? A.end[grp - 1] : 0)];
    } else {
      const long long i = e - A.end[A.n - 1];
      if (i < A.N) {
        const bool vis = A.radii[i] > 0;
        const float gx = A.means2D_grad[3 * i], gy = A.means2D_grad[3 * i + 1];
        v = vis ? sqrtf(gx * gx + gy * gy) : 0.f;
        A.radii_out[i] = A.radii[i];
      } else {
        v = A.radii[i - A.N] > 0 ? 1.f : 0.f;
      }
    }
    A.flat[e] = v;
  }
}

// ---------------------------------------------------------------------------------
// Native mapping iteration (row a13): the activations of GaussianModel for one iteration,
// the end-of-iteration statistics fold / opacity reset, and keyframe insertion's row append.
__global__ __launch_bounds__(256) void k_map_activate(mgs_map_activate_args A) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.num_gaussians) return;
  if (A.scale_dims == 3) {
#pragma unroll
    for (int k = 0; k < 3; k++) A.scales[3 * (size_t)i + k] = expf(A.log_scales[3 * (size_t)i + k]);
  } else {
    const float sc = expf(A.log_scales[i]);
    A.scales[3 * (size_t)i] = sc; A.scales[3 * (size_t)i + 1] = sc; A.scales[3 * (size_t)i + 2] = sc;
  }
  const float4 q = reinterpret_cast<const float4*>(A.raw_rotations)[i];
  // torch.nn.functional.normalize: q / max(|q|, 1e-12)
  const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  reinterpret_cast<float4*>(A.rotations)[i] = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
  A.opacities[i] = 1.f / (1.f + expf(-A.opacity_logits[i]));
  if (A.shs) {
    const int K = A.sh_coeffs;
    float* dst = A.shs + (size_t)3 * K * i;
#pragma unroll
    for (int c = 0; c < 3; c++) dst[c] = A.features_dc[3 * (size_t)i + c];
    for (int k = 3; k < 3 * K; k++) dst[k] = A.features_rest[(size_t)3 * (K - 1) * i + (k - 3)];
  }
}

__global__ __launch_bounds__(256) void k_map_finish(mgs_map_finish_args A, float reset_logit) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.num_gaussians) return;
  if (A.gradnorm_inc) {
    A.xyz_gradient_accum[i] += A.gradnorm_inc[i];
    A.denom[i] += A.denom_inc[i];
    A.max_radii2D[i] = fmaxf(A.max_radii2D[i], (float)A.radii_max[i]);
  }
  if (A.reset_mode != 0) {
    if (A.reset_mode == 1 || !(A.denom_inc[i] > 0.f)) {
      A.opacity_logits[i] = reset_logit;
    } else if (A.reset_mode == 2) {
      // gaussian_model.py:375 stores `self.get_opacity[filter]` - the ACTIVATED opacity - as the new raw parameter of
      // a visible Gaussian: its logit l becomes sigmoid(l) (pinned by tests/golden/map_update_ref.npz: ron_*)
      A.opacity_logits[i] = 1.f / (1.f + expf(-A.opacity_logits[i]));
    }
    if (A.opacity_exp_avg) A.opacity_exp_avg[i] = 0.f;
    if (A.opacity_exp_avg_sq) A.opacity_exp_avg_sq[i] = 0.f;
  }
}

struct AppendPack {
  const unsigned int* old_rows[MGS_GATHER_MAX_TENSORS];
  const unsigned int* new_rows[MGS_GATHER_MAX_TENSORS];
  unsigned int* dst[MGS_GATHER_MAX_TENSORS];
  int width[MGS_GATHER_MAX_TENSORS];
  int n;
  long long rows_old, rows_new;
};

__global__ __launch_bounds__(256) void k_map_append(AppendPack G) {
  const int t = blockIdx.y;
  if (t >= G.n) return;
  const long long w = G.width[t], n_old = G.rows_old * w, total = (G.rows_old + G.rows_new) * w;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256)
    G.dst[t][e] = e < n_old ? G.old_rows[t][e] : (G.new_rows[t] ? G.new_rows[t][e - n_old] : 0u);
}

}  // namespace mgs

using namespace mgs;

extern "C" {

int32_t mgs_adam_step_multi(const mgs_adam_group* groups, int32_t num_groups, double beta1, double beta2,
                            double eps, void* stream) {
  if (!groups || num_groups < 1 || num_groups > MGS_ADAM_MAX_GROUPS) return MGS_ERR_BAD_ARGUMENT;
  AdamPack A;
  long long chunks = 0;
  int vec = 1, n = 0;
  for (int i = 0; i < num_groups; i++) {
    const mgs_adam_group& g = groups[i];
    if (g.numel < 0) return MGS_ERR_BAD_ARGUMENT;
    if (g.numel == 0) continue;
    if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq || g.step < 1) return MGS_ERR_BAD_ARGUMENT;
    const double bc1 = 1.0 - pow(beta1, (double)g.step), bc2 = 1.0 - pow(beta2, (double)g.step);
    if (((uintptr_t)g.param | (uintptr_t)g.grad | (uintptr_t)g.exp_avg | (uintptr_t)g.exp_avg_sq) & 15u) vec = 0;
    A.p[n] = g.param; A.g[n] = g.grad; A.m[n] = g.exp_avg; A.v[n] = g.exp_avg_sq;
    A.numel[n] = g.numel;
    chunks += (g.numel + 3) / 4;
    A.chunk_end[n] = chunks;
    A.step_size[n] = (float)((double)g.lr / bc1);
    A.inv_bc2_sqrt[n] = (float)(1.0 / sqrt(bc2));
    n++;
  }
  if (n == 0) return MGS_OK;
  A.n = n; A.beta1 = (float)beta1; A.beta2 = (float)beta2; A.eps = (float)eps;
  A.om_beta1 = (float)(1.0 - beta1); A.om_beta2 = (float)(1.0 - beta2);
  const long long want = (chunks + 255) / 256;
  const int grid = (int)(want < 1 ? 1 : (want > 65535 * 16 ? 65535 * 16 : want));
  launch("adam_multi", k_adam_multi, dim3(grid), dim3(256), (hipStream_t)stream, A, chunks, vec);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_pack_mapping_grads(const float* const* grads, const int64_t* numels, int32_t num_grads,
                               const float* means2D_grad, const int32_t* radii, int64_t num_gaussians,
                               float* flat, int32_t* radii_out, void* stream) {
  if (!grads || !numels || num_grads < 1 || num_grads > MGS_ADAM_MAX_GROUPS || !means2D_grad || !radii ||
      num_gaussians < 1 || !flat || !radii_out)
    return MGS_ERR_BAD_ARGUMENT;
  PackPack A;
  long long run = 0;
  for (int i = 0; i < num_grads; i++) {
    if (!grads[i] || numels[i] < 1) return MGS_ERR_BAD_ARGUMENT;
    A.src[i] = grads[i];
    run += numels[i];
    A.end[i] = run;
  }
  A.n = num_grads; A.means2D_grad = means2D_grad; A.radii = radii; A.radii_out = radii_out;
  A.N = num_gaussians; A.flat = flat;
  const long long total = run + 2 * num_gaussians;
  long long want = (total + 255) / 256;
  if (want > 8192) want = 8192;
  launch("pack_grads", k_pack_grads, dim3((unsigned)want), dim3(256), (hipStream_t)stream, A);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_map_plan_blocks(int32_t n) { return n < 1 ? 0 : (n + kPlanBlock - 1) / kPlanBlock; }

static bool plan_ok(const mgs_map_plan_args* a) {
  if (!a || a->n < 1 || !a->flags || !a->block_counts || !a->totals) return false;
  if (!a->prune_mask && (!a->grad_accum || !a->denom || !a->log_scales || !a->opacity_logit)) return false;
  return true;
}

int32_t mgs_map_plan_count(const mgs_map_plan_args* a, void* stream) {
  if (!plan_ok(a)) return MGS_ERR_BAD_ARGUMENT;
  const int nb = mgs_map_plan_blocks(a->n);
  launch("plan_count", k_plan_count, dim3(nb), dim3(kPlanBlock), (hipStream_t)stream, *a);
  launch("plan_scan", k_plan_scan, dim3(1), dim3(1024), (hipStream_t)stream, *a, nb);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_map_plan_emit(const mgs_map_plan_args* a, void* stream) {
  if (!plan_ok(a) || !a->src_index || !a->noise_row) return MGS_ERR_BAD_ARGUMENT;
  launch("plan_emit", k_plan_emit, dim3(mgs_map_plan_blocks(a->n)), dim3(kPlanBlock), (hipStream_t)stream, *a);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_map_gather(const mgs_map_gather_args* a, void* stream) {
  if (!a || a->num_tensors < 1 || a->num_tensors > MGS_GATHER_MAX_TENSORS || !a->src_index || a->rows < 0)
    return MGS_ERR_BAD_ARGUMENT;
  if (a->rows == 0) return MGS_OK;
  GatherPack G;
  long long maxw = 1;
  for (int i = 0; i < a->num_tensors; i++) {
    const mgs_gather_tensor& t = a->tensors[i];
    if (!t.src || !t.dst || t.width < 1 || t.mode < 0 || t.mode > MGS_GATHER_SPLIT_XYZ) return MGS_ERR_BAD_ARGUMENT;
    if (t.mode == MGS_GATHER_SPLIT_XYZ && (t.width != 3 || !a->rotations || !a->log_scales || ((!a->noise || !a->noise_row) && a->num_children > 0)))
      return MGS_ERR_BAD_ARGUMENT;
    G.src[i] = (const unsigned int*)t.src; G.dst[i] = (unsigned int*)t.dst; G.width[i] = t.width; G.mode[i] = t.mode;
    if (t.width > maxw) maxw = t.width;
  }
  G.n = a->num_tensors; G.src_index = a->src_index; G.rows = a->rows;
  G.rot = a->rotations; G.log_scales = a->log_scales; G.noise = a->noise; G.noise_row = a->noise_row;
  G.child_base = a->rows - 2 * (long long)a->num_children;
  long long want = (a->rows * maxw + 255) / 256;
  if (want > 4096) want = 4096;
  launch("gather_rows", k_gather_rows, dim3((unsigned)want, (unsigned)a->num_tensors), dim3(256), (hipStream_t)stream, G);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_map_activate(const mgs_map_activate_args* a, void* stream) {
  if (!a || a->num_gaussians < 1 || (a->scale_dims != 1 && a->scale_dims != 3) || a->sh_coeffs < 1 ||
      !a->log_scales || !a->raw_rotations || !a->opacity_logits || !a->scales || !a->rotations || !a->opacities)
    return MGS_ERR_BAD_ARGUMENT;
  if (a->shs && (!a->features_dc || (a->sh_coeffs > 1 && !a->features_rest))) return MGS_ERR_BAD_ARGUMENT;
  if (!a->shs && a->sh_coeffs != 1) return MGS_ERR_BAD_ARGUMENT;
  launch("map_activate", k_map_activate, dim3((a->num_gaussians + 255) / 256), dim3(256), (hipStream_t)stream, *a);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_map_finish_iteration(const mgs_map_finish_args* a, void* stream) {
  if (!a || a->num_gaussians < 1 || a->reset_mode < 0 || a->reset_mode > 3) return MGS_ERR_BAD_ARGUMENT;
  if (a->gradnorm_inc && (!a->denom_inc || !a->radii_max || !a->xyz_gradient_accum || !a->denom || !a->max_radii2D))
    return MGS_ERR_BAD_ARGUMENT;
  if (a->reset_mode != 0 && (!a->opacity_logits || !(a->reset_value > 0.f) || !(a->reset_value < 1.f))) return MGS_ERR_BAD_ARGUMENT;
  if (a->reset_mode >= 2 && !a->denom_inc) return MGS_ERR_BAD_ARGUMENT;
  if (!a->gradnorm_inc && a->reset_mode == 0) return MGS_OK;
  // inverse_sigmoid(x) = log(x / (1 - x))   (gaussian_splatting/utils/general_utils.py)
  const float logit = a->reset_mode ? (float)log((double)a->reset_value / (1.0 - (double)a->reset_value)) : 0.f;
  launch("map_finish", k_map_finish, dim3((a->num_gaussians + 255) / 256), dim3(256), (hipStream_t)stream, *a, logit);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_map_append(const mgs_map_append_args* a, void* stream) {
  if (!a || a->num_tensors < 1 || a->num_tensors > MGS_GATHER_MAX_TENSORS || a->rows_old < 0 || a->rows_new < 0)
    return MGS_ERR_BAD_ARGUMENT;
  if (a->rows_old + a->rows_new == 0) return MGS_OK;
  AppendPack G;
  long long maxw = 1;
  for (int i = 0; i < a->num_tensors; i++) {
    const mgs_gather_tensor& t = a->tensors[i];
    if ((!t.src && a->rows_old > 0) || !t.dst || t.width < 1) return MGS_ERR_BAD_ARGUMENT;
    G.old_rows[i] = (const unsigned int*)t.src; G.new_rows[i] = (const unsigned int*)a->new_rows[i];
    G.dst[i] = (unsigned int*)t.dst; G.width[i] = t.width;
    if (t.width > maxw) maxw = t.width;
  }
  G.n = a->num_tensors; G.rows_old = a->rows_old; G.rows_new = a->rows_new;
  long long want = ((a->rows_old + a->rows_new) * maxw + 255) / 256;
  if (want > 4096) want = 4096;
  launch("map_append", k_map_append, dim3((unsigned)want, (unsigned)a->num_tensors), dim3(256), (hipStream_t)stream, G);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

}  // extern "C"
