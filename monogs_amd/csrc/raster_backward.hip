// Backward pass of the rasteriser for gfx950 (CDNA4, wave64).  Atomic-free on the
// gradient path:
//   k_scan_*        exclusive scan of per-Gaussian pair counts -> pair_base (slot bases)
//   k_blend_bwd     1 workgroup / tile: back-to-front replay; per splat the 10 screen-space
//                   gradients are reduced over the tile's 256 pixels (DPP wave reduction +
//                   LDS) and stored ONCE, with plain stores, at the pair's slot
//   k_preprocess_bwd 1 thread / Gaussian: streams its contiguous slots, chains to
//                   means3D / scale / rot / SH / opacity / means2D and the per-Gaussian
//                   pose gradient, block-reduced to one partial per workgroup
//   k_tau_reduce    fixed-order sum of the partials -> grad_tau[6]
//
// Replaces rasterize_gaussians_backward of the reference's CUDA extension (its
// autograd.Function is invoked through gaussian_renderer/__init__.py:151-168; gradient
// sinks: gaussian_model.py:252-285,693-697, slam_frontend.py:365-378,606-611).
#include "launch.h"
#include "raster_kernels.h"

namespace mgs {

__device__ __forceinline__ void load_camera_b(Camera& c, const KP& P) {
#pragma unroll
  for (int i = 0; i < 16; i++) { c.V[i] = P.V[i]; c.PM[i] = P.PM[i]; c.Praw[i] = P.Praw[i]; }
  c.campos[0] = P.campos[0]; c.campos[1] = P.campos[1]; c.campos[2] = P.campos[2];
  c.W = P.W; c.H = P.H; c.tanfovx = P.tanfovx; c.tanfovy = P.tanfovy;
  c.focal_x = P.focal_x; c.focal_y = P.focal_y; c.scale_modifier = P.mod;
  c.sh_degree = P.deg; c.sh_coeffs = P.K; c.grid_x = P.grid_x; c.grid_y = P.grid_y;
}

// ---------------------------------------------------------------------------------
// wave64 sum via DPP (row = 16 lanes): quad_perm, row_ror, row_bcast15/31.  The total
// lands in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return v + __int_as_float(moved);
}

__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x124, 0xF>(v);   // row_ror:4
  v = dpp_add<0x128, 0xF>(v);   // row_ror:8
  v = dpp_add<0x142, 0xA>(v);   // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xC>(v);   // row_bcast:31 -> rows 2,3
  return v;
}

// ---------------------------------------------------------------------------------
// pair_base = exclusive scan of pair_count (N elements), three small launches.
__global__ __launch_bounds__(256) void k_scan_reduce(KP P, KB B) {
  __shared__ int s[256];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * kScanBlock + tid * 8;
  int sum = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) sum += (base + i < P.N) ? P.pair_count[base + i] : 0;
  s[tid] = sum;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) s[tid] += s[tid + off];
    __syncthreads();
  }
  if (tid == 0) B.block_sums[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(1024) void k_scan_sums(KB B, int nblk) {
  __shared__ int s[1024];
  const int tid = threadIdx.x;
  int carry = 0;
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + tid;
    const int v = (i < nblk) ? B.block_sums[i] : 0;
    s[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = (tid >= off) ? s[tid - off] : 0;
      __syncthreads();
      s[tid] += t;
      __syncthreads();
    }
    if (i < nblk) B.block_sums[i] = carry + s[tid] - v;
    carry += s[1023];
    __syncthreads();
  }
  if (tid == 0) B.block_sums[nblk] = carry;
}

__global__ __launch_bounds__(256) void k_scan_write(KP P, KB B) {
  __shared__ int s[256];
  const int tid = threadIdx.x;
  const int base = blockIdx.x * kScanBlock + tid * 8;
  int v[8], sum = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) { v[i] = (base + i < P.N) ? P.pair_count[base + i] : 0; sum += v[i]; }
  s[tid] = sum;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int t = (tid >= off) ? s[tid - off] : 0;
    __syncthreads();
    s[tid] += t;
    __syncthreads();
  }
  int run = B.block_sums[blockIdx.x] + s[tid] - sum;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (base + i < P.N) B.pair_base[base + i] = run;
    run += v[i];
  }
  if (base <= P.N - 1 && P.N - 1 < base + 8) B.pair_base[P.N] = run;
}

// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blend_bwd(KP P, KB B) {
  __shared__ float4 s_r0[256], s_r1[256], s_r2[256];
  __shared__ int s_slot[256];
  __shared__ float s_acc[256 * 10];
  __shared__ int s_maxlast;
  const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int tx = tile % P.grid_x, ty = tile / P.grid_x;
  const int px = tx * kTile + (tid & 15), py = ty * kTile + (tid >> 4);
  const bool inside = px < P.W && py < P.H;
  int start = P.tile_offset[tile], end = P.tile_offset[tile + 1];
  start = min(start, P.cap); end = min(end, P.cap);
  const int n = end - start;
  if (n <= 0) return;
  const size_t pix = (size_t)py * P.W + px, HW = (size_t)P.W * P.H;
  int last = 0;
  PixBwd st;
  {
    float gp[3] = {0.f, 0.f, 0.f}, gd = 0.f, Tf = 1.f;
    if (inside) {
      last = P.n_contrib[pix];
      Tf = P.final_T[pix];
      gp[0] = B.grad_color[pix]; gp[1] = B.grad_color[HW + pix]; gp[2] = B.grad_color[2 * HW + pix];
      if (B.grad_depth) gd = B.grad_depth[pix];
    }
    const float bg[3] = {P.bg[0], P.bg[1], P.bg[2]};
    pixbwd_init(st, Tf, gp, gd, bg);
  }
  if (tid == 0) s_maxlast = 0;
  __syncthreads();
  atomicMax(&s_maxlast, last);
  __syncthreads();
  const int maxlast = s_maxlast;
  const float fpx = (float)px, fpy = (float)py;
  const int nbatches = (n + 255) / 256;
  for (int b = nbatches - 1; b >= 0; b--) {
    const int base = b * 256;
    const int nb = min(256, n - base);
    __syncthreads();   // previous batch fully flushed
    int slot = -1;
    if (tid < nb) {
      const int k = start + base + tid;
      const unsigned int id = (unsigned int)P.keys[k];
      slot = B.pair_base[id] + (int)P.payload[k];
      s_slot[tid] = slot;
      if (base < maxlast) {
        const float4* src = reinterpret_cast<const float4*>(P.rec + id);
        s_r0[tid] = src[0]; s_r1[tid] = src[1]; s_r2[tid] = src[2];
      }
    }
    if (base >= maxlast) {   // no pixel of this tile ever reached these splats
      if (tid < nb) {
        float4* dst = B.pair_grad + (size_t)slot * 3;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        dst[0] = z; dst[1] = z; dst[2] = z;
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < 10; i++) s_acc[tid * 10 + i] = 0.f;
    __syncthreads();
    for (int j = nb - 1; j >= 0; j--) {
      const int idx = base + j;
      bool c = false;
      SplatGrad g;
      if (idx < last) {
        const float4 a = s_r0[j], bb = s_r1[j], cc = s_r2[j];
        SplatLite s;
        s.x = a.x; s.y = a.y; s.depth = a.z; s.o = a.w;
        s.A = bb.x; s.B = bb.y; s.C = bb.z;
        s.r = cc.x; s.g = cc.y; s.b = cc.z;
        c = blend_backward_step(fpx, fpy, s, st, g);
      }
      if (__ballot(c) == 0ull) continue;
      float v[10];
      v[0] = c ? g.gx : 0.f; v[1] = c ? g.gy : 0.f; v[2] = c ? g.gA : 0.f; v[3] = c ? g.gB : 0.f;
      v[4] = c ? g.gC : 0.f; v[5] = c ? g.gop : 0.f; v[6] = c ? g.gr : 0.f; v[7] = c ? g.gg : 0.f;
      v[8] = c ? g.gb : 0.f; v[9] = c ? g.gdepth : 0.f;
#pragma unroll
      for (int i = 0; i < 10; i++) v[i] = wave_sum_to_lane63(v[i]);
      if (lane == 63) {
#pragma unroll
        for (int i = 0; i < 10; i++) atomicAdd(&s_acc[j * 10 + i], v[i]);
      }
    }
    __syncthreads();
    if (tid < nb) {
      const float* a = &s_acc[tid * 10];
      float4* dst = B.pair_grad + (size_t)slot * 3;
      dst[0] = make_float4(a[0], a[1], a[2], a[3]);
      dst[1] = make_float4(a[4], a[5], a[6], a[7]);
      dst[2] = make_float4(a[8], a[9], 0.f, 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kPreBlock) void k_preprocess_bwd(KP P, KB B) {
  __shared__ float s_tau[kPreBlock / 64][6];
  const int idx = blockIdx.x * kPreBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (idx < P.N) {
    const float4 r1 = reinterpret_cast<const float4*>(P.rec + idx)[1];
    const float4 r2 = reinterpret_cast<const float4*>(P.rec + idx)[2];
    const int radius = __float_as_int(r1.w);
    const unsigned int flags = __float_as_uint(r2.w);
    float dmean[3] = {0.f, 0.f, 0.f}, dndc[2] = {0.f, 0.f}, dop = 0.f;
    float dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float grgb[3] = {0.f, 0.f, 0.f};
    const float p[3] = {P.means[3 * idx], P.means[3 * idx + 1], P.means[3 * idx + 2]};
    if (radius > 0) {
      float a[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int s0 = B.pair_base[idx], s1 = B.pair_base[idx + 1];
      for (int s = s0; s < s1; s++) {
        const float4* src = B.pair_grad + (size_t)s * 3;
        const float4 x = src[0], y = src[1], z = src[2];
        a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w;
        a[4] += y.x; a[5] += y.y; a[6] += y.z; a[7] += y.w;
        a[8] += z.x; a[9] += z.y;
      }
      Camera cam;
      load_camera_b(cam, P);
      float sc[3], q[4], c6[6];
      const float *psc = nullptr, *pq = nullptr, *pc6 = nullptr;
      if (P.covp) {
#pragma unroll
        for (int i = 0; i < 6; i++) c6[i] = P.covp[6 * (size_t)idx + i];
        pc6 = c6;
      } else {
        sc[0] = P.scales[3 * idx]; sc[1] = P.scales[3 * idx + 1]; sc[2] = P.scales[3 * idx + 2];
        const float4 qq = reinterpret_cast<const float4*>(P.rots)[idx];
        q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
        psc = sc; pq = q;
      }
      const float g_xy[2] = {a[0], a[1]};
      const float g_con[3] = {a[2], a[3], a[4]};
      GaussGrad gg;
      project_gaussian_backward(cam, p, psc, pq, pc6, g_xy, g_con, a[5], a[9], gg);
#pragma unroll
      for (int i = 0; i < 3; i++) { dmean[i] = gg.dmean[i]; dscale[i] = gg.dscale[i]; grgb[i] = a[6 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++) drot[i] = gg.drot[i];
#pragma unroll
      for (int i = 0; i < 6; i++) { dcov[i] = gg.dcov6[i]; tau[i] = gg.dtau[i]; }
      dndc[0] = gg.dndc[0]; dndc[1] = gg.dndc[1];
      dop = gg.dop;
    }
    // colours
    if (P.shs) {
      float* dsh = B.g_colors + (size_t)3 * P.K * idx;
      if (radius > 0) {
        if (P.deg == 0) {
#pragma unroll
          for (int c = 0; c < 3; c++) dsh[c] = (flags & (1u << c)) ? 0.f : SH_C0 * grgb[c];
          for (int k = 3; k < 3 * P.K; k++) dsh[k] = 0.f;
        } else {
          sh_backward(P.deg, P.K, P.shs + (size_t)3 * P.K * idx, p, P.campos, flags, grgb, dsh, dmean);
        }
      } else {
        for (int k = 0; k < 3 * P.K; k++) dsh[k] = 0.f;
      }
    } else {
      B.g_colors[3 * idx] = grgb[0]; B.g_colors[3 * idx + 1] = grgb[1]; B.g_colors[3 * idx + 2] = grgb[2];
    }
    B.g_means3D[3 * idx] = dmean[0]; B.g_means3D[3 * idx + 1] = dmean[1]; B.g_means3D[3 * idx + 2] = dmean[2];
    B.g_means2D[3 * idx] = dndc[0]; B.g_means2D[3 * idx + 1] = dndc[1]; B.g_means2D[3 * idx + 2] = 0.f;
    B.g_opac[idx] = dop;
    if (B.g_scales) { B.g_scales[3 * idx] = dscale[0]; B.g_scales[3 * idx + 1] = dscale[1]; B.g_scales[3 * idx + 2] = dscale[2]; }
    if (B.g_rots) reinterpret_cast<float4*>(B.g_rots)[idx] = make_float4(drot[0], drot[1], drot[2], drot[3]);
    if (B.g_cov) {
#pragma unroll
      for (int i = 0; i < 6; i++) B.g_cov[6 * (size_t)idx + i] = dcov[i];
    }
  }
  // block reduction of the pose gradient (fixed order -> deterministic)
#pragma unroll
  for (int i = 0; i < 6; i++) tau[i] = wave_sum_to_lane63(tau[i]);
  if (lane == 63) {
#pragma unroll
    for (int i = 0; i < 6; i++) s_tau[wave][i] = tau[i];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float s = 0.f;
    for (int w = 0; w < kPreBlock / 64; w++) s += s_tau[w][threadIdx.x];
    B.tau_partial[blockIdx.x * 6 + threadIdx.x] = s;
  }
}

__global__ __launch_bounds__(384) void k_tau_reduce(KB B, int nblk) {
  // 6 components x 64 lanes; each lane strides the partials, then a wave sum.
  const int comp = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < nblk; i += 64) s += B.tau_partial[i * 6 + comp];
  s = wave_sum_to_lane63(s);
  if (lane == 63) B.g_tau[comp] = s;
}

// ---------------------------------------------------------------------------------
int launch_backward(const KP& P, const KB& B, hipStream_t st) {
  const int nscan = (P.N + kScanBlock - 1) / kScanBlock;
  launch("scan_reduce", k_scan_reduce, dim3(nscan), dim3(256), st, P, B);
  launch("scan_sums", k_scan_sums, dim3(1), dim3(1024), st, B, nscan);
  launch("scan_write", k_scan_write, dim3(nscan), dim3(256), st, P, B);
  launch("blend_bwd", k_blend_bwd, dim3(P.T), dim3(256), st, P, B);
  const int npre = (P.N + kPreBlock - 1) / kPreBlock;
  launch("preprocess_bwd", k_preprocess_bwd, dim3(npre), dim3(kPreBlock), st, P, B);
  launch("tau_reduce", k_tau_reduce, dim3(1), dim3(384), st, B, npre);
  return hipGetLastError() == hipSuccess ? MGS_OK : MGS_ERR_LAUNCH;
}

}  // namespace mgs
