// Tracking-loop glue fused into a handful of launches (after the
// rasteriser the tracking iteration was dominated by ~100 tiny PyTorch kernels).
//
//  * k_pose_adam_update: torch.optim.Adam step for the four per-camera parameter groups
//    (cam_rot_delta, cam_trans_delta, exposure_a, exposure_b; utils/slam_frontend.py:364-392)
//    followed by update_pose (utils/pose_utils.py:88-98): T <- Exp([rho; theta]) T, deltas
//    zeroed, convergence flag.  One thread; replaces ~60 launches and two host syncs.
//  * k_track_loss_fwd / k_track_loss_bwd: monocular tracking objective
//    || Huber( opacity * mask * ((|a|+eps) * image + b - gt) ) ||_2
//    (utils/slam_utils.py:188-205 + :58-75 + slam_frontend.py:596-598), forward and
//    backward (d/d image, d/d a, d/d b), HBM-bound streaming kernels.
#include <hip/hip_runtime.h>

#include "../../include/monogs_raster.h"
#include "launch.h"
#include "objective_math.h"
#include "sketch_kernels.h"

namespace mgs {

__device__ inline void so3_exp_V(const float th[3], float R[9], float V[9]) {
  const float x = th[0], y = th[1], z = th[2];
  const float W[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
  float W2[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      W2[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
  const float angle = sqrtf(x * x + y * y + z * z);
  float a, b, c, d;   // R = I + a W + b W2 ; V = I + c W + d W2
  if (angle < 1e-5f) {
    a = 1.f; b = 0.5f; c = 0.5f; d = 1.f / 6.f;
  } else {
    const float s = sinf(angle), co = cosf(angle), a2 = angle * angle;
    a = s / angle; b = (1.f - co) / a2; c = b; d = (angle - s) / (a2 * angle);
  }
  for (int i = 0; i < 9; i++) {
    const float I = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
    R[i] = I + a * W[i] + b * W2[i];
    V[i] = I + c * W[i] + d * W2[i];
  }
}

__global__ __launch_bounds__(512) void k_pose_adam_update(mgs_pose_adam_args A) {
  // optional fused reductions: wave w < 6 sums component w of the tau partials, waves 6 / 7 the
  // two exposure sums, each in a fixed order (independent strided loads, then a wave sum)
  __shared__ float s_g[8];     // rot(3), trans(3), a, b
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (A.tau_partials && wave < 6) {
    float v = 0.f;
#pragma unroll 8
    for (int i = lane; i < A.num_tau_partials; i += 64) v += A.tau_partials[i * 6 + wave];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) s_g[wave < 3 ? 3 + wave : wave - 3] = v;      // tau = [rho (trans); theta (rot)]
  }
  // Many partials (the objective evaluated in the forward's epilogue leaves one per quadrant wave, 4T of them):
  // all 512 threads sum the four arrays together - every load of a thread independent, a wave sum, the eight
  // wave totals added in a fixed order by thread 0 below - instead of one wave walking each array.
  const bool coop = A.loss_partials && A.exposure_partials && A.loss_norm_mode == 1 && A.num_loss_partials > 1024 &&
                    A.num_exposure_partials == A.num_loss_partials &&
                    (!A.l1_partials || A.num_l1_partials == A.num_loss_partials);
  __shared__ float s_coop[4][8];
  if (coop) {
    const int n = A.num_loss_partials;
    const float* l1p = (A.l1_partials && A.best) ? A.l1_partials : nullptr;
    // (an absent array is read from a present one and its sum dropped: the loads stay unconditional, so all of
    // an unrolled trip's loads are in flight together)
    const float* r3 = l1p ? l1p : A.loss_partials;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll 5
    for (int i = threadIdx.x; i < n; i += 512) {
      v0 += A.loss_partials[i];
      v1 += A.exposure_partials[i];
      v2 += A.exposure_partials[n + i];
      v3 += r3[i];
    }
    if (!l1p) v3 = 0.f;
    for (int off = 32; off > 0; off >>= 1) {
      v0 += __shfl_down(v0, off); v1 += __shfl_down(v1, off); v2 += __shfl_down(v2, off); v3 += __shfl_down(v3, off);
    }
    if (lane == 0) { s_coop[0][wave] = v0; s_coop[1][wave] = v1; s_coop[2][wave] = v2; s_coop[3][wave] = v3; }
  }
  if (!coop && A.exposure_partials && wave >= 6) {
    const int c = wave - 6;
    float v = 0.f;
#pragma unroll 8
    for (int i = lane; i < A.num_exposure_partials; i += 64) v += A.exposure_partials[c * A.num_exposure_partials + i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) s_g[6 + c] = v;
  }
  __shared__ float s_loss[2];
  if (!coop && A.loss_partials && wave >= 6 && (A.loss_norm_mode == 0 || wave == 6)) {   // colour / depth block sums, or sum h^2
    const int c = wave - 6;
    float v = 0.f;
#pragma unroll 8
    for (int i = lane; i < A.num_loss_partials; i += 64) v += A.loss_partials[c * A.num_loss_partials + i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) s_loss[c] = v;
  }
  __shared__ float s_l1;
  if (!coop && A.l1_partials && A.best && wave == 5) {       // (wave 5 has also summed a tau component: both are short)
    float v = 0.f;
#pragma unroll 8
    for (int i = lane; i < A.num_l1_partials; i += 64) v += A.l1_partials[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) s_l1 = v;
  }
  __syncthreads();
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (coop) {
    float t[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      t[c] = 0.f;
#pragma unroll
      for (int w = 0; w < 8; w++) t[c] += s_coop[c][w];
    }
    s_loss[0] = t[0]; s_g[6] = t[1]; s_g[7] = t[2]; s_l1 = t[3];
  }
  if (A.sticky_converged && A.converged && *A.converged != 0) return;    // the reference has left its loop
  float grad_scale = 1.f;
  if (A.loss_partials && A.loss_norm_mode == 1) {
    // tracking objective = (sum |h|^p)^(1/p): the gradients summed above lack its loss^(1-p) (1 / loss for p = 2)
    float l, inv;
    norm_finish(s_loss[0], norm_p(A.loss_pnorm), l, inv);
    grad_scale = inv * (A.loss_grad_out ? A.loss_grad_out[0] : 1.f);
    if (A.loss_view) { A.loss_view[0] = l; A.loss_view[1] = inv; }
  } else if (A.loss_partials) {
    const float l = A.loss_w_rgb * s_loss[0] + A.loss_w_depth * s_loss[1];
    if (A.loss_view) A.loss_view[0] = l;
    if (A.loss_accum) A.loss_accum[0] += l;
  }
  // One thread, ~60 scalars of state: every load is issued BEFORE the first store (a store in
  // between makes the compiler re-load what may alias, and each dependent global round trip costs
  // ~1 us here: this kernel went 14.7 -> 10.7 us with the loads batched, profiles/r02_tracking_profile.txt).
  float* P[4] = {A.cam_rot_delta, A.cam_trans_delta, A.exposure_a, A.exposure_b};
  const float* G[4] = {A.tau_partials ? &s_g[0] : A.grad_rot, A.tau_partials ? &s_g[3] : A.grad_trans,
                       A.exposure_partials ? (A.exposure_a ? &s_g[6] : nullptr) : A.grad_a,
                       A.exposure_partials ? (A.exposure_b ? &s_g[7] : nullptr) : A.grad_b};
  const float lr[4] = {A.lr_rot, A.lr_trans, A.lr_a, A.lr_b};
  constexpr int first[4] = {0, 3, 6, 7}, len[4] = {3, 3, 1, 1};
  float p[8], gr[8], m[8], v[8], Tm[16], proj[16];
  bool on[8];
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int i = 0; i < len[g]; i++) {
      const int o = first[g] + i;
      on[o] = G[g] != nullptr && P[g] != nullptr;
      p[o] = P[g] ? P[g][i] : 0.f;
      gr[o] = on[o] ? G[g][i] * grad_scale : 0.f;
      m[o] = A.exp_avg[o];
      v[o] = A.exp_avg_sq[o];
    }
  const bool have_T = A.T != nullptr;
  const bool want_mats = have_T && A.viewmatrix_out && A.projmatrix_out && A.projection;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    Tm[i] = have_T ? A.T[i] : 0.f;
    proj[i] = want_mats ? A.projection[i] : 0.f;
  }
  // best-iterate snapshot: the state this iteration RENDERED (before the step below)
  float best_prev = 0.f, best_count = 0.f;
  if (A.best) { best_prev = A.best[0]; best_count = A.best[20]; }
  const bool improved = A.best && A.l1_partials && have_T && s_l1 < best_prev;
  float snap[18];
#pragma unroll
  for (int i = 0; i < 16; i++) snap[i] = Tm[i];
  snap[16] = p[6]; snap[17] = p[7];
  const float bc1 = 1.f - powf(A.beta1, (float)A.step);
  const float bc2s = sqrtf(1.f - powf(A.beta2, (float)A.step));
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int i = 0; i < len[g]; i++) {
      const int o = first[g] + i;
      if (!on[o]) continue;
      m[o] = A.beta1 * m[o] + (1.f - A.beta1) * gr[o];
      v[o] = A.beta2 * v[o] + (1.f - A.beta2) * gr[o] * gr[o];
      const float denom = sqrtf(v[o]) / bc2s + A.eps;
      p[o] -= (lr[g] / bc1) * (m[o] / denom);
    }
  // pose update from the stepped deltas (registers)
  const bool move = have_T && !A.no_pose_update && A.cam_rot_delta && A.cam_trans_delta;
  const float th[3] = {move ? p[0] : 0.f, move ? p[1] : 0.f, move ? p[2] : 0.f};
  const float rho[3] = {move ? p[3] : 0.f, move ? p[4] : 0.f, move ? p[5] : 0.f};
  if (move) {
    float R[9], V[9];
    so3_exp_V(th, R, V);
    float t[3];
    for (int i = 0; i < 3; i++) t[i] = V[3 * i] * rho[0] + V[3 * i + 1] * rho[1] + V[3 * i + 2] * rho[2];
    float Tn[12];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++)
        Tn[4 * i + j] = R[3 * i] * Tm[j] + R[3 * i + 1] * Tm[4 + j] + R[3 * i + 2] * Tm[8 + j] +
                        (j == 3 ? t[i] : 0.f) * Tm[15];
    for (int i = 0; i < 12; i++) Tm[i] = Tn[i];
    for (int i = 0; i < 6; i++) p[i] = 0.f;           // deltas are consumed by update_pose
  }
  // ---- stores ----
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int i = 0; i < len[g]; i++) {
      const int o = first[g] + i;
      if (on[o]) { A.exp_avg[o] = m[o]; A.exp_avg_sq[o] = v[o]; }
      if (on[o] || (move && o < 6)) P[g][i] = p[o];     // applied deltas are zeroed even without a step
    }
  if (A.best) {
    if (improved) {
      A.best[0] = s_l1;
#pragma unroll
      for (int i = 0; i < 18; i++) A.best[1 + i] = snap[i];
      A.best[19] = best_count;
    }
    A.best[20] = best_count + 1.f;
    if (A.l1_partials) A.best[21] = s_l1;      // criterion of THIS iteration's render (a trace for tests / logging)
  }
  if (have_T) {
    if (move) for (int i = 0; i < 12; i++) A.T[i] = Tm[i];
    const float n2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2] + rho[0] * rho[0] + rho[1] * rho[1] +
                     rho[2] * rho[2];
    if (A.best) A.best[22] = sqrtf(n2);        // |tau| of the step just applied (what the convergence test sees)
    if (A.converged) *A.converged = n2 < A.converged_threshold * A.converged_threshold ? 1 : 0;
    if (want_mats) {   // matrices of the updated pose: view = T^T, full = view @ projection
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
          float acc = 0.f;
          for (int k = 0; k < 4; k++) acc += Tm[4 * k + i] * proj[4 * k + j];
          A.projmatrix_out[4 * i + j] = acc;
          A.viewmatrix_out[4 * i + j] = Tm[4 * j + i];
        }
    }
  }
}

// viewmatrix = T^T, projmatrix = viewmatrix @ projection (row-major 4x4, camera_utils.py:94-104)
__global__ void k_camera_from_pose(const float* T, const float* proj, float* view, float* full) {
  const int i = threadIdx.x >> 2, j = threadIdx.x & 3;      // 16 threads
  if (threadIdx.x >= 16) return;
  float acc = 0.f;
  for (int k = 0; k < 4; k++) acc += T[4 * k + i] * proj[4 * k + j];   // view[i][k] = T[k][i]
  const float v = T[4 * j + i];
  full[4 * i + j] = acc;
  view[4 * i + j] = v;
}

// ---------------------------------------------------------------------------------
constexpr int kLossBlock = 256;
constexpr int kLossBlocks = 512;

__device__ __forceinline__ float block_sum(float v, float* s_red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < kLossBlock / 64; w++) t += s_red[w];
  return t;
}

// partial[b] = sum over the block's pixels of |h|^p (3 channels)
__global__ __launch_bounds__(kLossBlock) void k_track_loss_fwd(mgs_tracking_loss_args A) {
  __shared__ float s_red[kLossBlock / 64];
  const float gain = fabsf(A.exposure_a[0]) + A.exposure_eps, bias = A.exposure_b[0];
  const float pn = norm_p(A.pnorm);
  const size_t HW = (size_t)A.num_pixels;
  float acc = 0.f;
  for (size_t p = (size_t)blockIdx.x * kLossBlock + threadIdx.x; p < HW; p += (size_t)gridDim.x * kLossBlock) {
    const float om = A.opacity[p] * (A.mask ? A.mask[p] : 1.f);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float dh, phi, gam;
      const float h = huber(om * (gain * A.image[c * HW + p] + bias - A.gt[c * HW + p]), A.huber_delta, dh);
      norm_terms(h, pn, phi, gam);
      acc += phi;
    }
  }
  const float t = block_sum(acc, s_red);
  if (threadIdx.x == 0) A.partial[blockIdx.x] = t;
}

// loss = (sum partial)^(1/p); scalars[0] = loss, scalars[1] = loss^(1-p) (1/loss for p = 2; 0 if loss == 0)
__global__ __launch_bounds__(kLossBlock) void k_track_loss_finish(mgs_tracking_loss_args A, int nblk) {
  __shared__ float s_red[kLossBlock / 64];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblk; i += kLossBlock) acc += A.partial[i];
  const float t = block_sum(acc, s_red);
  if (threadIdx.x == 0) {
    float l, sc;
    norm_finish(t, norm_p(A.pnorm), l, sc);
    A.scalars[0] = l;
    A.scalars[1] = sc;
  }
}

// grad_image = gout/loss * h * h' * om * gain ; partial sums for d/da, d/db
// nfwd > 0 (fused form, mgs_tracking_iteration): the forward's nfwd block sums are added up
// here by every workgroup (no k_track_loss_finish launch) and the exposure partials go
// BEHIND them in `partial` (summed by k_pose_adam_update, no k_track_loss_bwd_finish launch).
__global__ __launch_bounds__(kLossBlock) void k_track_loss_bwd(mgs_tracking_loss_args A, int nfwd) {
  __shared__ float s_red[kLossBlock / 64];
  const float a = A.exposure_a[0];
  const float gain = fabsf(a) + A.exposure_eps, bias = A.exposure_b[0];
  const float sgn = a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f);
  const float pn = norm_p(A.pnorm);
  float inv_loss = 0.f;      // loss^(1-p), the factor of the norm's derivative
  if (nfwd > 0) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nfwd; i += kLossBlock) acc += A.partial[i];
    float l;
    norm_finish(block_sum(acc, s_red), pn, l, inv_loss);
    if (blockIdx.x == 0 && threadIdx.x == 0) { A.scalars[0] = l; A.scalars[1] = inv_loss; }
    __syncthreads();
  } else {
    inv_loss = A.scalars[1];
  }
  const float k = A.grad_out[0] * inv_loss;
  const size_t HW = (size_t)A.num_pixels;
  float ga = 0.f, gb = 0.f;
  for (size_t p = (size_t)blockIdx.x * kLossBlock + threadIdx.x; p < HW; p += (size_t)gridDim.x * kLossBlock) {
    const float om = A.opacity[p] * (A.mask ? A.mask[p] : 1.f);
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float im = A.image[c * HW + p];
      float dh, phi, gam;
      const float h = huber(om * (gain * im + bias - A.gt[c * HW + p]), A.huber_delta, dh);
      norm_terms(h, pn, phi, gam);
      const float gr = k * gam * dh * om;     // dL/d(residual before opacity) * om
      A.grad_image[c * HW + p] = gr * gain;
      ga += gr * im;
      gb += gr;
    }
  }
  const float ta = block_sum(ga, s_red);
  const float tb = block_sum(gb, s_red);
  if (threadIdx.x == 0) {
    A.partial[nfwd + blockIdx.x] = ta * sgn;
    A.partial[nfwd + gridDim.x + blockIdx.x] = tb;
  }
}

__global__ __launch_bounds__(kLossBlock) void k_track_loss_bwd_finish(mgs_tracking_loss_args A, int nblk) {
  __shared__ float s_red[kLossBlock / 64];
  float x = 0.f, y = 0.f;
  for (int i = threadIdx.x; i < nblk; i += kLossBlock) { x += A.partial[i]; y += A.partial[nblk + i]; }
  const float ta = block_sum(x, s_red);
  const float tb = block_sum(y, s_red);
  if (threadIdx.x == 0) { A.grad_a[0] = ta; A.grad_b[0] = tb; }
}

// One pass (mgs_tracking_loss_onepass): block sums of |h|^p, the image gradient and the exposure
// partials WITHOUT the loss^(1-p) factor of the norm's derivative (1 / loss for p = 2); k_pose_adam_update applies it
// (loss_norm_mode).  partial = [n] sum |h|^p | [n] d/da | [n] d/db | [n] sum |r| (before Huber).
// VEC: four consecutive pixels per thread and trip through 16-B loads / stores (as k_map_loss_fused<true>).
template <bool VEC>
__global__ __launch_bounds__(kLossBlock) void k_track_loss_onepass(mgs_tracking_loss_args A) {
  __shared__ float s_red[kLossBlock / 64];
  const float a = A.exposure_a[0];
  const float gain = fabsf(a) + A.exposure_eps, bias = A.exposure_b[0];
  const float sgn = a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f);
  const size_t HW = (size_t)A.num_pixels;
  const float pn = norm_p(A.pnorm);
  float acc = 0.f, ga = 0.f, gb = 0.f, l1 = 0.f;
  auto sample = [&](float om, float im, float gt) {      // one colour sample: sums + d/d image
    float dh, phi, gam;
    const float r = om * (gain * im + bias - gt);
    l1 += fabsf(r);
    const float h = huber(r, A.huber_delta, dh);
    norm_terms(h, pn, phi, gam);
    acc += phi;
    const float gr = gam * dh * om;
    ga += gr * im;
    gb += gr;
    return gr * gain;
  };
  if constexpr (VEC) {
    const size_t Q = HW / 4;
    for (size_t q = (size_t)blockIdx.x * kLossBlock + threadIdx.x; q < Q; q += (size_t)gridDim.x * kLossBlock) {
      float4 om = reinterpret_cast<const float4*>(A.opacity)[q];
      if (A.mask) {
        const float4 m = reinterpret_cast<const float4*>(A.mask)[q];
        om.x *= m.x; om.y *= m.y; om.z *= m.z; om.w *= m.w;
      }
      float4 im[3], gt[3], g[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        im[c] = reinterpret_cast<const float4*>(A.image + c * HW)[q];
        gt[c] = reinterpret_cast<const float4*>(A.gt + c * HW)[q];
      }
      // pixel by pixel, channel by channel: the per-pixel order of additions of the scalar form
#pragma unroll
      for (int c = 0; c < 3; c++) g[c].x = sample(om.x, im[c].x, gt[c].x);
#pragma unroll
      for (int c = 0; c < 3; c++) g[c].y = sample(om.y, im[c].y, gt[c].y);
#pragma unroll
      for (int c = 0; c < 3; c++) g[c].z = sample(om.z, im[c].z, gt[c].z);
#pragma unroll
      for (int c = 0; c < 3; c++) g[c].w = sample(om.w, im[c].w, gt[c].w);
#pragma unroll
      for (int c = 0; c < 3; c++) reinterpret_cast<float4*>(A.grad_image + c * HW)[q] = g[c];
    }
  } else {
    for (size_t p = (size_t)blockIdx.x * kLossBlock + threadIdx.x; p < HW; p += (size_t)gridDim.x * kLossBlock) {
      const float om = A.opacity[p] * (A.mask ? A.mask[p] : 1.f);
#pragma unroll
      for (int c = 0; c < 3; c++) A.grad_image[c * HW + p] = sample(om, A.image[c * HW + p], A.gt[c * HW + p]);
    }
  }
  const float t = block_sum(acc, s_red);
  const float ta = block_sum(ga, s_red);
  const float tb = block_sum(gb, s_red);
  const float tl = block_sum(l1, s_red);
  if (threadIdx.x == 0) {
    A.partial[blockIdx.x] = t;
    A.partial[gridDim.x + blockIdx.x] = ta * sgn;
    A.partial[2 * gridDim.x + blockIdx.x] = tb;
    A.partial[3 * gridDim.x + blockIdx.x] = tl;
  }
}

// ---------------------------------------------------------------------------------
// Damped least squares of the sketched LM step (utils/slam_frontend.py:672-697):
//   x = argmin || [SJ; sqrt(lambda) I] x + [Sf; 0] ||   <=>   (SJ^T SJ + lambda I) x = -SJ^T Sf
// with 8 unknowns [trans(3), rot(3), exposure_a, exposure_b], followed by TempCamera.step
// (:49-53): T <- Exp(x[:6]) T, exposure += x[6:8].  One workgroup: fp64 row reduction,
// 8x8 Cholesky by thread 0.  (torch.linalg.lstsq on this 1032x8 problem costs ~3.7 ms of
// host-side solver setup per call.)
__global__ __launch_bounds__(256) void k_lm_solve_step(mgs_lm_step_args A) {
  __shared__ double s_red[4][44];
  const float loss_in = A.loss ? A.loss[0] : 0.f;     // read before the accumulators are zeroed below
  double acc[44];
#pragma unroll
  for (int i = 0; i < 44; i++) acc[i] = 0.0;
  for (int r = threadIdx.x; r < A.rows; r += 256) {
    float row[8];
    if (A.SJ) {
#pragma unroll
      for (int i = 0; i < 8; i++) row[i] = A.SJ[(size_t)r * 8 + i];
    } else {
#pragma unroll
      for (int i = 0; i < 6; i++) row[i] = A.sj_tau[(size_t)r * 6 + i];
      row[6] = A.sj_exposure[(size_t)r * 2]; row[7] = A.sj_exposure[(size_t)r * 2 + 1];
    }
    const float f = A.Sf[r];
    int k = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = i; j < 8; j++) acc[k++] += (double)row[i] * (double)row[j];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[36 + i] += (double)row[i] * (double)f;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 44; i++) {
    double v = acc[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) s_red[wave][i] = v;
  }
  __syncthreads();
  // every thread has read its rows: leave the accumulators zero for the next iteration (zero_after)
  if (A.zero_after)
    for (int i = threadIdx.x; i < A.zero_count; i += 256) A.zero_after[i] = 0.f;
  if (threadIdx.x != 0) return;
  if (A.lm_state && A.lm_state[3] != 0.f) return;      // converged earlier: the reference has left its loop
  double H[8][8], g[8];
  int k = 0;
  for (int i = 0; i < 8; i++)
    for (int j = i; j < 8; j++) {
      const double v = s_red[0][k] + s_red[1][k] + s_red[2][k] + s_red[3][k];
      H[i][j] = v; H[j][i] = v;
      k++;
    }
  float lambda = A.lambda;
  if (A.lm_state) {      // trust-region rule on the device (slam_frontend.py:536-545)
    lambda = A.lm_state[0];
    const float loss = loss_in;
    if (A.lm_state[2] != 0.f)
      lambda = loss < A.lm_state[1] ? fmaxf(lambda / A.decrease_factor, A.min_lambda)
                                    : fminf(lambda * A.increase_factor, A.max_lambda);
    A.lm_state[0] = lambda; A.lm_state[1] = loss; A.lm_state[2] = 1.f;
  }
  for (int i = 0; i < 8; i++) {
    g[i] = -(s_red[0][36 + i] + s_red[1][36 + i] + s_red[2][36 + i] + s_red[3][36 + i]);
    H[i][i] += (double)lambda;
  }
  // Cholesky H = L L^T (lambda > 0 makes H positive definite)
  double L[8][8];
  for (int i = 0; i < 8; i++)
    for (int j = 0; j <= i; j++) {
      double sum = H[i][j];
      for (int q = 0; q < j; q++) sum -= L[i][q] * L[j][q];
      L[i][j] = (i == j) ? sqrt(sum > 1e-300 ? sum : 1e-300) : sum / L[j][j];
    }
  double y[8], x[8];
  for (int i = 0; i < 8; i++) {
    double sum = g[i];
    for (int q = 0; q < i; q++) sum -= L[i][q] * y[q];
    y[i] = sum / L[i][i];
  }
  for (int i = 7; i >= 0; i--) {
    double sum = y[i];
    for (int q = i + 1; q < 8; q++) sum -= L[q][i] * x[q];
    x[i] = sum / L[i][i];
  }
  for (int i = 0; i < 8; i++) A.x_out[i] = (float)x[i];
  bool converged = false;
  double xn2 = 0.0;
  for (int i = 0; i < 8; i++) xn2 += x[i] * x[i];
  if (A.lm_state) {
    converged = sqrt(xn2) < (double)A.converged_threshold;
    A.lm_state[3] = converged ? 1.f : 0.f;
  }
  if (A.best && A.loss && A.T) {     // best iterate = the state this iteration rendered (before the step)
    const float count = A.best[20];
    if (loss_in < A.best[0]) {
      A.best[0] = loss_in;
      for (int i = 0; i < 16; i++) A.best[1 + i] = A.T[i];
      A.best[17] = A.exposure_a ? A.exposure_a[0] : 0.f;
      A.best[18] = A.exposure_b ? A.exposure_b[0] : 0.f;
      A.best[19] = count;
    }
    A.best[20] = count + 1.f;
    A.best[21] = loss_in;                      // criterion of this iteration's render
    A.best[22] = (float)sqrt(xn2);             // |x| of this iteration's step
  }
  if (converged) return;             // slam_frontend.py:699-706: the converged step is never assigned
  if (A.T) {
    const float rho[3] = {(float)x[0], (float)x[1], (float)x[2]};
    const float th[3] = {(float)x[3], (float)x[4], (float)x[5]};
    float R[9], V[9];
    so3_exp_V(th, R, V);
    float t[3];
    for (int i = 0; i < 3; i++) t[i] = V[3 * i] * rho[0] + V[3 * i + 1] * rho[1] + V[3 * i + 2] * rho[2];
    float Tn[12];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 4; j++)
        Tn[4 * i + j] = R[3 * i] * A.T[j] + R[3 * i + 1] * A.T[4 + j] + R[3 * i + 2] * A.T[8 + j] +
                        (j == 3 ? t[i] : 0.f) * A.T[15];
    for (int i = 0; i < 12; i++) A.T[i] = Tn[i];
    if (A.projection && A.viewmatrix_out && A.projmatrix_out) {
      // camera matrices of the stepped pose (viewmatrix = T^T, projmatrix = viewmatrix @ projection), so
      // that the next iteration needs no mgs_camera_from_pose launch
      float Tm[16];
      for (int i = 0; i < 12; i++) Tm[i] = Tn[i];
      for (int i = 12; i < 16; i++) Tm[i] = A.T[i];
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
          float acc = 0.f;
          for (int k = 0; k < 4; k++) acc += Tm[4 * k + i] * A.projection[4 * k + j];
          A.projmatrix_out[4 * i + j] = acc;
          A.viewmatrix_out[4 * i + j] = Tm[4 * j + i];
        }
    }
  }
  if (A.exposure_a) A.exposure_a[0] += (float)x[6];
  if (A.exposure_b) A.exposure_b[0] += (float)x[7];
}

// ---------------------------------------------------------------------------------
// Mapping objective (utils/slam_utils.py:224-253):
//   loss = w_rgb * mean_{3HW} | m * ((|a|+eps) img + b - gt) |  +  w_depth * mean_{HW} | dm * (depth - gt_depth) |
// with m = rgb_pixel_mask_mapping and dm = gt_depth > 0.01 (RGB-D only; w_depth = 0 for
// monocular).  apply_exposure = 0 reproduces `initialization=True` (image used as is).
// One streaming pass for the value, one for the gradients.
__global__ __launch_bounds__(kLossBlock) void k_map_loss_fwd(mgs_mapping_loss_args A, int fused_finish) {
  __shared__ float s_red[kLossBlock / 64];
  const float gain = A.apply_exposure ? fabsf(A.exposure_a[0]) + A.exposure_eps : 1.f;
  const float bias = A.apply_exposure ? A.exposure_b[0] : 0.f;
  const size_t HW = (size_t)A.num_pixels;
  float sc = 0.f, sd = 0.f;
  for (size_t p = (size_t)blockIdx.x * kLossBlock + threadIdx.x; p < HW; p += (size_t)gridDim.x * kLossBlock) {
    const float m = A.mask ? A.mask[p] : 1.f;
#pragma unroll
    for (int c = 0; c < 3; c++) sc += fabsf(m * (gain * A.image[c * HW + p] + bias - A.gt[c * HW + p]));
    if (A.w_depth != 0.f) {
      const float gdp = A.gt_depth[p];
      const float dm = (A.depth_mask_threshold < 0.f || gdp > A.depth_mask_threshold) ? 1.f : 0.f;
      sd += fabsf(dm * (A.depth[p] - gdp));
    }
  }
  const float tc = block_sum(sc, s_red);
  const float td = block_sum(sd, s_red);
  if (threadIdx.x == 0) {   // write-through (sc1) stores: visible to the last workgroup's sc1 loads without a fence
    __hip_atomic_store(&A.partial[blockIdx.x], tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&A.partial[gridDim.x + blockIdx.x], td, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!fused_finish) return;
  // The workgroup that finishes last sums the block partials (saves the dependent finish launch).
  // Fence-free hand-off as in k_bin_colsum (MI355X_MICROARCH.md, "Hand-offs measured with sc1
  // loads"): the two partials were stored write-through, the storing lane drains them
  // (s_waitcnt) and takes an agent-scope ticket; the last arriver reads with sc1 loads.  A full
  // __threadfence() here costs an L2 write-back per workgroup: 7.7 -> 17 us for this kernel.
  // The ticket (the int behind the 2 n partials) is zero on entry and is reset here, so the same
  // scratch serves the next call.
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int* ticket = reinterpret_cast<int*>(A.partial + 2 * gridDim.x);
    s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
    if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  const int nblk = gridDim.x;
  float x = 0.f, y = 0.f;
  for (int i = threadIdx.x; i < nblk; i += kLossBlock) {
    x += __hip_atomic_load(&A.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    y += __hip_atomic_load(&A.partial[nblk + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const float sumc = block_sum(x, s_red);
  const float sumd = block_sum(y, s_red);
  if (threadIdx.x == 0) {
    const float hw = (float)A.num_pixels;
    A.loss[0] = A.w_rgb * sumc / (3.f * hw) + A.w_depth * sumd / hw;
  }
}

__global__ __launch_bounds__(kLossBlock) void k_map_loss_finish(mgs_mapping_loss_args A, int nblk) {
  __shared__ float s_red[kLossBlock / 64];
  float x = 0.f, y = 0.f;
  for (int i = threadIdx.x; i < nblk; i += kLossBlock) { x += A.partial[i]; y += A.partial[nblk + i]; }
  const float tc = block_sum(x, s_red);
  const float td = block_sum(y, s_red);
  if (threadIdx.x == 0) {
    const float hw = (float)A.num_pixels;
    A.loss[0] = A.w_rgb * tc / (3.f * hw) + A.w_depth * td / hw;
  }
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ __launch_bounds__(kLossBlock) void k_map_loss_bwd(mgs_mapping_loss_args A) {
  __shared__ float s_red[kLossBlock / 64];
  const float a = A.apply_exposure ? A.exposure_a[0] : 1.f;
  const float gain = A.apply_exposure ? fabsf(a) + A.exposure_eps : 1.f;
  const float bias = A.apply_exposure ? A.exposure_b[0] : 0.f;
  const size_t HW = (size_t)A.num_pixels;
  const float hw = (float)A.num_pixels;
  const float kc = A.grad_out[0] * A.w_rgb / (3.f * hw), kd = A.grad_out[0] * A.w_depth / hw;
  float ga = 0.f, gb = 0.f;
  for (size_t p = (size_t)blockIdx.x * kLossBlock + threadIdx.x; p < HW; p += (size_t)gridDim.x * kLossBlock) {
    const float m = A.mask ? A.mask[p] : 1.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float im = A.image[c * HW + p];
      const float g = kc * m * sgn(m * (gain * im + bias - A.gt[c * HW + p]));
      A.grad_image[c * HW + p] = g * gain;
      ga += g * im;
      gb += g;
    }
    if (A.grad_depth) {
      float g = 0.f;
      if (A.w_depth != 0.f) {
        const float gdp = A.gt_depth[p];
        const float dm = (A.depth_mask_threshold < 0.f || gdp > A.depth_mask_threshold) ? 1.f : 0.f;
        g = kd * dm * sgn(dm * (A.depth[p] - gdp));
      }
      A.grad_depth[p] = g;
    }
  }
  const float ta = block_sum(ga, s_red);
  const float tb = block_sum(gb, s_red);
  if (threadIdx.x == 0) {
    A.partial[blockIdx.x] = ta * sgn(a);
    A.partial[gridDim.x + blockIdx.x] = tb;
  }
}

__global__ __launch_bounds__(kLossBlock) void k_map_loss_bwd_finish(mgs_mapping_loss_args A, int nblk) {
  __shared__ float s_red[kLossBlock / 64];
  float x = 0.f, y = 0.f;
  for (int i = threadIdx.x; i < nblk; i += kLossBlock) { x += A.partial[i]; y += A.partial[nblk + i]; }
  const float ta = block_sum(x, s_red);
  const float tb = block_sum(y, s_red);
  if (threadIdx.x == 0) {
    if (A.grad_a) A.grad_a[0] = A.apply_exposure ? ta : 0.f;
    if (A.grad_b) A.grad_b[0] = A.apply_exposure ? tb : 0.f;
  }
}

// Value AND gradients of the mapping objective in ONE pass (mgs_mapping_view_iteration): the L1
// gradient does not depend on the loss value, so nothing has to wait for a reduction.  Block
// partials [4][n]: sum |colour residual|, sum |depth residual|, d/da, d/db; their consumer
// (k_pose_adam_update) sums them in a fixed order.  The upstream gradient is 1 unless
// A.grad_out is given.
// VEC: four consecutive pixels per thread and trip through 16-B loads / stores (num_pixels % 4 == 0 and
// 16-B aligned planes, checked on the host): at 640x480 every thread makes ONE trip with ten independent
// 16-B loads in flight instead of 2-3 trips of dword loads, and 300 workgroups leave 300 partials.
template <bool VEC>
__global__ __launch_bounds__(kLossBlock) void k_map_loss_fused(mgs_mapping_loss_args A) {
  __shared__ float s_red[kLossBlock / 64];
  const float a = A.apply_exposure ? A.exposure_a[0] : 1.f;
  const float gain = A.apply_exposure ? fabsf(a) + A.exposure_eps : 1.f;
  const float bias = A.apply_exposure ? A.exposure_b[0] : 0.f;
  const size_t HW = (size_t)A.num_pixels;
  const float hw = (float)A.num_pixels;
  const float go = A.grad_out ? A.grad_out[0] : 1.f;
  const float kc = go * A.w_rgb / (3.f * hw), kd = go * A.w_depth / hw;
  float sc = 0.f, sd = 0.f, ga = 0.f, gb = 0.f;
  // one colour sample / one depth sample: sums + gradient
  auto colour = [&](float m, float im, float gt) {
    const float r = m * (gain * im + bias - gt);
    sc += fabsf(r);
    const float g = kc * m * sgn(r);
    ga += g * im;
    gb += g;
    return g * gain;
  };
  auto depth = [&](float d, float gdp) {
    const float dm = (A.depth_mask_threshold < 0.f || gdp > A.depth_mask_threshold) ? 1.f : 0.f;
    const float r = dm * (d - gdp);
    sd += fabsf(r);
    return kd * dm * sgn(r);
  };
  if constexpr (VEC) {
    const size_t Q = HW / 4;
    const float4 one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    for (size_t q = (size_t)blockIdx.x * kLossBlock + threadIdx.x; q < Q; q += (size_t)gridDim.x * kLossBlock) {
      const float4 m = A.mask ? reinterpret_cast<const float4*>(A.mask)[q] : one4;
      float4 im[3], gt[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        im[c] = reinterpret_cast<const float4*>(A.image + c * HW)[q];
        gt[c] = reinterpret_cast<const float4*>(A.gt + c * HW)[q];
      }
      float4 dp = one4, gd = one4;
      const bool with_depth = A.w_depth != 0.f;
      if (with_depth) {
        dp = reinterpret_cast<const float4*>(A.depth)[q];
        gd = reinterpret_cast<const float4*>(A.gt_depth)[q];
      }
      // same order of additions per pixel as the scalar form: pixel by pixel, channel by channel, depth last
      float4 g[3], gdo = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < 3; c++) g[c].x = colour(m.x, im[c].x, gt[c].x);
      if (with_depth) gdo.x = depth(dp.x, gd.x);
#pragma unroll
      for (int c = 0; c < 3; c++) g[c].y = colour(m.y, im[c].y, gt[c].y);
      if (with_depth) gdo.y = depth(dp.y, gd.y);
#pragma unroll
      for (int c = 0; c < 3; c++) g[c].z = colour(m.z, im[c].z, gt[c].z);
      if (with_depth) gdo.z = depth(dp.z, gd.z);
#pragma unroll
      for (int c = 0; c < 3; c++) g[c].w = colour(m.w, im[c].w, gt[c].w);
      if (with_depth) gdo.w = depth(dp.w, gd.w);
#pragma unroll
      for (int c = 0; c < 3; c++) reinterpret_cast<float4*>(A.grad_image + c * HW)[q] = g[c];
      if (A.grad_depth) reinterpret_cast<float4*>(A.grad_depth)[q] = gdo;
    }
  } else {
    for (size_t p = (size_t)blockIdx.x * kLossBlock + threadIdx.x; p < HW; p += (size_t)gridDim.x * kLossBlock) {
      const float m = A.mask ? A.mask[p] : 1.f;
#pragma unroll
      for (int c = 0; c < 3; c++) A.grad_image[c * HW + p] = colour(m, A.image[c * HW + p], A.gt[c * HW + p]);
      if (A.w_depth != 0.f) {
        const float g = depth(A.depth[p], A.gt_depth[p]);
        if (A.grad_depth) A.grad_depth[p] = g;
      } else if (A.grad_depth) {
        A.grad_depth[p] = 0.f;
      }
    }
  }
  const float tc = block_sum(sc, s_red);
  const float td = block_sum(sd, s_red);
  const float ta = block_sum(ga, s_red);
  const float tb = block_sum(gb, s_red);
  const int n = gridDim.x;
  if (threadIdx.x == 0) {   // write-through stores: see the hand-off below
    __hip_atomic_store(&A.partial[blockIdx.x], tc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&A.partial[n + blockIdx.x], td, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&A.partial[2 * n + blockIdx.x], A.apply_exposure ? ta * sgn(a) : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&A.partial[3 * n + blockIdx.x], A.apply_exposure ? tb : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!A.partial_ticket_ready) return;
  // optional finish by the workgroup that arrives last (fence-free hand-off as in k_map_loss_fwd; the
  // ticket is the int behind the 4 n partials, zero on entry, reset here): loss and exposure gradients
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int* ticket = reinterpret_cast<int*>(A.partial + 4 * n);
    s_last = atomicAdd(ticket, 1) == n - 1;
    if (s_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < n; i += kLossBlock)
#pragma unroll
    for (int c = 0; c < 4; c++) acc[c] += __hip_atomic_load(&A.partial[c * n + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float tot[4];
#pragma unroll
  for (int c = 0; c < 4; c++) tot[c] = block_sum(acc[c], s_red);
  if (threadIdx.x == 0) {
    if (A.loss) A.loss[0] = A.w_rgb * tot[0] / (3.f * hw) + A.w_depth * tot[1] / hw;
    if (A.grad_a) A.grad_a[0] = tot[2];
    if (A.grad_b) A.grad_b[0] = tot[3];
  }
}


__global__ __launch_bounds__(256) void k_sketch_assign(long long m, int chunk, int d, int bits,
                                                       unsigned int k0, unsigned int k1, unsigned int k2,
                                                       int* bucket, float* weights) {
  const unsigned int mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < m; p += (long long)gridDim.x * 256) {
    unsigned int x = (unsigned int)p;
    do { x = perm_round(x, mask, bits, k0, k1); } while ((long long)x >= m);
    const long long used = (long long)chunk * d;
    bucket[p] = (long long)x < used ? (int)(x / (unsigned int)chunk) : -1;
    weights[p] = (hash32((unsigned int)p ^ k2) & 0x10000u) ? 1.f : -1.f;
  }
}

__global__ __launch_bounds__(kSketchThreads) void k_sketch_residual(mgs_sketch_residual_args A, SketchKeys K) {
  extern __shared__ float s_acc[];   // [d][3]: Sf, d/da, d/db
  __shared__ float s_red[kSketchThreads / 64];
  sketch_residual_block(A, K, s_acc, s_red, blockIdx.x, gridDim.x);
}

static int loss_blocks(int64_t hw) {
  const int64_t b = (hw + kLossBlock - 1) / kLossBlock;
  return (int)(b < kLossBlocks ? (b < 1 ? 1 : b) : kLossBlocks);
}

}  // namespace mgs

using namespace mgs;

extern "C" {

int32_t mgs_camera_from_pose(const float* T, const float* projection, float* viewmatrix,
                             float* projmatrix, void* stream) {
  if (!T || !projection || !viewmatrix || !projmatrix) return MGS_ERR_BAD_ARGUMENT;
  launch("camera_from_pose", k_camera_from_pose, dim3(1), dim3(64), (hipStream_t)stream, T, projection,
         viewmatrix, projmatrix);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_pose_adam_step(const mgs_pose_adam_args* a, void* stream) {
  if (!a || !a->exp_avg || !a->exp_avg_sq || a->step < 1) return MGS_ERR_BAD_ARGUMENT;
  // a NULL delta pointer skips its group (mapping: keyframes outside the pose window)
  if ((!a->cam_rot_delta || !a->cam_trans_delta) && !a->loss_partials && !a->exposure_partials && !a->grad_a && !a->grad_b)
    return MGS_ERR_BAD_ARGUMENT;
  if ((a->grad_a && !a->exposure_a) || (a->grad_b && !a->exposure_b)) return MGS_ERR_BAD_ARGUMENT;
  launch("pose_adam_update", k_pose_adam_update, dim3(1), dim3(512), (hipStream_t)stream, *a);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

static bool map_args_ok(const mgs_mapping_loss_args* a) {
  if (!a || !a->image || !a->gt || !a->partial || !a->loss || a->num_pixels < 1) return false;
  if (a->apply_exposure && (!a->exposure_a || !a->exposure_b)) return false;
  if (a->w_depth != 0.f && (!a->depth || !a->gt_depth)) return false;
  return true;
}

int32_t mgs_mapping_loss_forward(const mgs_mapping_loss_args* a, void* stream) {
  if (!map_args_ok(a)) return MGS_ERR_BAD_ARGUMENT;
  const int nb = loss_blocks(a->num_pixels);
  launch("map_loss_fwd", k_map_loss_fwd, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a, a->partial_ticket_ready ? 1 : 0);
  if (!a->partial_ticket_ready)
    launch("map_loss_finish", k_map_loss_finish, dim3(1), dim3(kLossBlock), (hipStream_t)stream, *a, nb);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_mapping_loss_backward(const mgs_mapping_loss_args* a, void* stream) {
  if (!map_args_ok(a) || !a->grad_out || !a->grad_image) return MGS_ERR_BAD_ARGUMENT;
  const int nb = loss_blocks(a->num_pixels);
  launch("map_loss_bwd", k_map_loss_bwd, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a);
  if (a->grad_a || a->grad_b)   // exposure gradients are the only consumers of the partial sums
    launch("map_loss_bwd_fin", k_map_loss_bwd_finish, dim3(1), dim3(kLossBlock), (hipStream_t)stream, *a, nb);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_mapping_loss_partial_count(int64_t num_pixels) { return 4 * loss_blocks(num_pixels) + 1; }

// Single-pass form used by mgs_mapping_view_iteration; partial = [4][*nblk_out].
int32_t mgs_mapping_loss_fused(const mgs_mapping_loss_args* a, int32_t* nblk_out, void* stream) {
  if (!a || !a->image || !a->gt || !a->partial || !a->grad_image || a->num_pixels < 1) return MGS_ERR_BAD_ARGUMENT;
  if (a->apply_exposure && (!a->exposure_a || !a->exposure_b)) return MGS_ERR_BAD_ARGUMENT;
  if (a->w_depth != 0.f && (!a->depth || !a->gt_depth || !a->grad_depth)) return MGS_ERR_BAD_ARGUMENT;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = a->num_pixels % 4 == 0 && al16(a->image) && al16(a->gt) && al16(a->mask) && al16(a->depth) &&
                   al16(a->gt_depth) && al16(a->grad_image) && al16(a->grad_depth);
  int nb = loss_blocks(a->num_pixels);
  if (vec) {
    nb = loss_blocks(a->num_pixels / 4);
    launch("map_loss_fused", k_map_loss_fused<true>, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a);
  } else {
    launch("map_loss_fused", k_map_loss_fused<false>, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a);
  }
  if (nblk_out) *nblk_out = nb;
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_lm_solve_step(const mgs_lm_step_args* a, void* stream) {
  if (!a || !a->Sf || !a->x_out || a->rows < 1) return MGS_ERR_BAD_ARGUMENT;
  if (!a->SJ && (!a->sj_tau || !a->sj_exposure)) return MGS_ERR_BAD_ARGUMENT;
  if (a->lm_state ? (!a->loss || !(a->increase_factor > 0.f) || !(a->decrease_factor > 0.f)) : !(a->lambda > 0.f))
    return MGS_ERR_BAD_ARGUMENT;
  launch("lm_solve_step", k_lm_solve_step, dim3(1), dim3(256), (hipStream_t)stream, *a);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_sketch_assign(int64_t num_pixels, int32_t stack_dim, int32_t sketch_dim, uint64_t key,
                          int32_t* bucket, float* weights, void* stream) {
  SketchKeys K;
  if (!bucket || !weights || !sketch_keys(num_pixels, stack_dim, sketch_dim, key, K)) return MGS_ERR_BAD_ARGUMENT;
  const int nb = loss_blocks(num_pixels);
  launch("sketch_assign", k_sketch_assign, dim3(nb), dim3(256), (hipStream_t)stream, (long long)num_pixels, K.chunk,
         stack_dim * sketch_dim, K.bits, K.k0, K.k1, K.k2, bucket, weights);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_sketch_residual(const mgs_sketch_residual_args* a, void* stream) {
  if (!a || !a->image || !a->opacity || !a->gt || !a->exposure_a || !a->exposure_b || !a->bucket ||
      !a->weights || !a->grad_image || !a->Sf || !a->sj_exposure || !a->l1 || a->num_pixels < 1 ||
      a->stack_dim < 1 || a->sketch_dim < 1)
    return MGS_ERR_BAD_ARGUMENT;
  const size_t smem = sizeof(float) * 3 * (size_t)a->stack_dim * a->sketch_dim;
  if (smem > 48 * 1024) return MGS_ERR_UNSUPPORTED;
  const int64_t want = (a->num_pixels + kSketchThreads - 1) / kSketchThreads;
  const int nb = (int)(want < kSketchBlocks ? want : kSketchBlocks);
  SketchKeys K = {0, 0, 0, 0u, 0u, 0u};
  if (a->assign && !sketch_keys(a->num_pixels, a->stack_dim, a->sketch_dim, a->assign_key, K)) return MGS_ERR_BAD_ARGUMENT;
  launch_smem("sketch_residual", k_sketch_residual, dim3(nb), dim3(kSketchThreads), smem, (hipStream_t)stream, *a, K);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_tracking_loss_partial_count(int64_t num_pixels) { return 4 * loss_blocks(num_pixels); }

// p of the norm: <= 0 selects 2; 0 < p < 1 is not a norm (and its derivative is unbounded at 0)
static bool pnorm_ok(float p) { return !(p > 0.f && p < 1.f) && p == p; }

// Fused form used by mgs_tracking_iteration: forward sums + backward in two launches; the
// exposure partials ([2, nblk]) start at partial + nblk.
int32_t mgs_tracking_loss_fused(const mgs_tracking_loss_args* a, int32_t* nblk_out, void* stream) {
  if (!a || !a->image || !a->opacity || !a->gt || !a->exposure_a || !a->exposure_b || !a->partial ||
      !a->scalars || !a->grad_out || !a->grad_image || a->num_pixels < 1)
    return MGS_ERR_BAD_ARGUMENT;
  if (!pnorm_ok(a->pnorm)) return MGS_ERR_BAD_ARGUMENT;
  const int nb = loss_blocks(a->num_pixels);
  launch("track_loss_fwd", k_track_loss_fwd, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a);
  launch("track_loss_bwd", k_track_loss_bwd, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a, nb);
  if (nblk_out) *nblk_out = nb;
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_tracking_loss_onepass(const mgs_tracking_loss_args* a, int32_t* nblk_out, void* stream) {
  if (!a || !a->image || !a->opacity || !a->gt || !a->exposure_a || !a->exposure_b || !a->partial ||
      !a->grad_image || a->num_pixels < 1)
    return MGS_ERR_BAD_ARGUMENT;
  if (!pnorm_ok(a->pnorm)) return MGS_ERR_BAD_ARGUMENT;
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool vec = a->num_pixels % 4 == 0 && al16(a->image) && al16(a->opacity) && al16(a->gt) && al16(a->mask) &&
                   al16(a->grad_image);
  int nb = loss_blocks(a->num_pixels);
  if (vec) {
    nb = loss_blocks(a->num_pixels / 4);
    launch("track_loss", k_track_loss_onepass<true>, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a);
  } else {
    launch("track_loss", k_track_loss_onepass<false>, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a);
  }
  if (nblk_out) *nblk_out = nb;
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_tracking_loss_forward(const mgs_tracking_loss_args* a, void* stream) {
  if (!a || !a->image || !a->opacity || !a->gt || !a->exposure_a || !a->exposure_b || !a->partial ||
      !a->scalars || a->num_pixels < 1)
    return MGS_ERR_BAD_ARGUMENT;
  if (!pnorm_ok(a->pnorm)) return MGS_ERR_BAD_ARGUMENT;
  const int nb = loss_blocks(a->num_pixels);
  launch("track_loss_fwd", k_track_loss_fwd, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a);
  launch("track_loss_finish", k_track_loss_finish, dim3(1), dim3(kLossBlock), (hipStream_t)stream, *a, nb);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

int32_t mgs_tracking_loss_backward(const mgs_tracking_loss_args* a, void* stream) {
  if (!a || !a->image || !a->opacity || !a->gt || !a->exposure_a || !a->exposure_b || !a->partial ||
      !a->scalars || !a->grad_out || !a->grad_image || !a->grad_a || !a->grad_b || a->num_pixels < 1)
    return MGS_ERR_BAD_ARGUMENT;
  if (!pnorm_ok(a->pnorm)) return MGS_ERR_BAD_ARGUMENT;
  const int nb = loss_blocks(a->num_pixels);
  launch("track_loss_bwd", k_track_loss_bwd, dim3(nb), dim3(kLossBlock), (hipStream_t)stream, *a, 0);
  launch("track_loss_bwd_fin", k_track_loss_bwd_finish, dim3(1), dim3(kLossBlock), (hipStream_t)stream, *a, nb);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

}  // extern "C"
