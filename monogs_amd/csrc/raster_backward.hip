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

// Ten wave64 sums at once, written directly as v_add_f32_dpp (the compiler otherwise
// SLP-packs the adds into v_pk_add_f32, which cannot carry a DPP modifier, and emits
// v_mov_dpp + v_mov + v_pk_add: 150 instructions instead of 60).  Each DPP step runs over
// the ten registers in turn, so a register is re-read nine instructions after it was
// written (>= the 2 wait states a DPP read needs); the leading s_nop covers the first.
// Totals land in lane 63.
__device__ __forceinline__ void wave_sum10_to_lane63(float (&r)[10]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %5, %5, %5 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %6, %6, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %7, %7, %7 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %9, %9, %9 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]));
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
// Back-to-front replay, one workgroup (2 waves) per 16x16 tile, TWO pixels per lane
// (rows y and y+8 of the same column, so dx is shared and the fp32 work of the pair
// packs into v_pk_* instructions).
//
// Per splat and pixel the replay produces a weight W = dL/dG * G and a blend weight
// w = alpha * T; every screen-space gradient of the splat is a pixel sum of W or w times
// a monomial of (dx, dy) or the pixel's upstream gradient:
//   S1 = sum W, Sx = sum W dx, Sy = sum W dy, Sxx, Sxy, Syy      (mean2D, conic, opacity)
//   Rr, Rg, Rb = sum w * dL/dC_ch,  Rd = sum w * dL/dD            (colour, depth)
// The lane first adds its two pixels, the 10 sums are reduced over the wave with DPP,
// written by lane 63 to the wave's own LDS partial (no atomics, fixed order =>
// deterministic) and combined / converted by the thread that staged the splat, which
// stores the pair's 40-B record once at its slot.
//
// A splat that does not contribute to a pixel is replayed as a transparent layer
// (alpha = 0): the recurrences stay branch-free and the "colour of the previously
// visited splat" is wave-uniform.
constexpr int kBwdBatch = 128;
constexpr int kBwdThreads = 128;
constexpr float kLog2eB = 1.4426950408889634f;
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kBwdThreads) void k_blend_bwd(KP P, KB B) {
  __shared__ float4 s_r0[kBwdBatch], s_r1[kBwdBatch], s_r2[kBwdBatch];
  __shared__ float4 s_part[2][kBwdBatch][3];
  __shared__ int s_maxlast;
  const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = tile % P.grid_x, ty = tile / P.grid_x;
  const int px = tx * kTile + (tid & 15);
  const int py0 = ty * kTile + (tid >> 4), py1 = py0 + 8;
  int start = P.tile_offset[tile], end = P.tile_offset[tile + 1];
  start = min(start, P.cap); end = min(end, P.cap);
  const int n = end - start;
  if (n <= 0) return;
  const size_t HW = (size_t)P.W * P.H;
  int last[2] = {0, 0};
  v2f g0 = {0.f, 0.f}, g1 = {0.f, 0.f}, g2 = {0.f, 0.f}, gd = {0.f, 0.f};
  v2f T = {1.f, 1.f}, Tfbg = {0.f, 0.f};
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int py = q ? py1 : py0;
    if (px < P.W && py < P.H) {
      const size_t pix = (size_t)py * P.W + px;
      last[q] = P.n_contrib[pix];
      T[q] = P.final_T[pix];
      g0[q] = B.grad_color[pix]; g1[q] = B.grad_color[HW + pix]; g2[q] = B.grad_color[2 * HW + pix];
      if (B.grad_depth) gd[q] = B.grad_depth[pix];
      Tfbg[q] = -T[q] * (P.bg[0] * g0[q] + P.bg[1] * g1[q] + P.bg[2] * g2[q]);
    }
  }
  if (tid == 0) s_maxlast = 0;
  __syncthreads();
  atomicMax(&s_maxlast, max(last[0], last[1]));
  __syncthreads();
  const int maxlast = s_maxlast;
  const float fpx = (float)px;
  const v2f fpy = {(float)py0, (float)py1};
  v2f a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, ad = {0.f, 0.f};
  v2f la = {0.f, 0.f};
  float lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, lcd = 0.f;
  const int nbatches = (n + kBwdBatch - 1) / kBwdBatch;
  for (int b = nbatches - 1; b >= 0; b--) {
    const int base = b * kBwdBatch;
    const int nb = min(kBwdBatch, n - base);
    __syncthreads();   // previous batch fully flushed
    int slot = -1;
    float4 q0, q1;
    if (tid < nb) {
      const int k = start + base + tid;
      const unsigned int id = (unsigned int)P.keys[k];
      slot = B.pair_base[id] + (int)P.payload[k];
      if (base < maxlast) {
        const float4* src = reinterpret_cast<const float4*>(P.rec + id);
        q0 = src[0]; q1 = src[1];
        const float4 q2 = src[2];
        s_r0[tid] = make_float4(q0.x, q0.y, -0.5f * kLog2eB * q1.x, -kLog2eB * q1.y);
        s_r1[tid] = make_float4(-0.5f * kLog2eB * q1.z, q0.w, q0.z, q2.x);
        s_r2[tid] = make_float4(q2.y, q2.z, 0.f, 0.f);
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
    {   // zero this wave's partials: 128 x 3 float4 per wave, 6 per lane
      float4* pp = &s_part[wave][0][0];
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < (kBwdBatch * 3) / 64; i++) pp[lane + 64 * i] = z;
    }
    __syncthreads();
    float4 u = s_r0[nb - 1], v = s_r1[nb - 1];
    float2 cgb = *reinterpret_cast<const float2*>(&s_r2[nb - 1]);
    for (int j = nb - 1; j >= 0; j--) {
      // software prefetch of the next (j-1) record: LDS latency overlaps this iteration
      const int jn = max(j - 1, 0);
      const float4 un = s_r0[jn], vn = s_r1[jn];
      const float2 cn = *reinterpret_cast<const float2*>(&s_r2[jn]);
      const float dx = u.x - fpx;
      const v2f dy = v2f{u.y, u.y} - fpy;
      const v2f pw = dx * (u.z * dx + u.w * dy) + v.x * dy * dy;
      const v2f G = v2f{__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
      const v2f araw = v.y * G;
      const v2f alpha = v2f{fminf(kAlphaMax, araw.x), fminf(kAlphaMax, araw.y)};
      const int idx = base + j;
      const bool c0 = idx < last[0] && pw.x <= 0.f && alpha.x >= kAlphaMin;
      const bool c1 = idx < last[1] && pw.y <= 0.f && alpha.y >= kAlphaMin;
      if (__ballot(c0 || c1) != 0ull) {
        const v2f ae = v2f{c0 ? alpha.x : 0.f, c1 ? alpha.y : 0.f};
        // fold the pending layer into the accumulation behind us
        a0 += la * (lc0 - a0); a1 += la * (lc1 - a1); a2 += la * (lc2 - a2); ad += la * (lcd - ad);
        const v2f om = 1.f - ae;
        const v2f rom = v2f{__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
        T *= rom;
        v2f dLda = (v.w - a0) * g0 + (cgb.x - a1) * g1 + (cgb.y - a2) * g2 + (v.z - ad) * gd;
        dLda = dLda * T + Tfbg * rom;
        v2f Wt = araw * dLda;
        Wt = v2f{c0 ? Wt.x : 0.f, c1 ? Wt.y : 0.f};
        const v2f w = ae * T;
        la = ae;
        lc0 = v.w; lc1 = cgb.x; lc2 = cgb.y; lcd = v.z;
        const v2f Wy = Wt * dy, Wyy = Wy * dy;
        const v2f wr = w * g0, wg = w * g1, wb = w * g2, wd = w * gd;
        float r[10];
        const float Ws = Wt.x + Wt.y, Sy = Wy.x + Wy.y;
        r[0] = Ws; r[1] = Ws * dx; r[2] = Sy; r[3] = r[1] * dx; r[4] = Sy * dx;
        r[5] = Wyy.x + Wyy.y;
        r[6] = wr.x + wr.y; r[7] = wg.x + wg.y; r[8] = wb.x + wb.y; r[9] = wd.x + wd.y;
        wave_sum10_to_lane63(r);
        if (lane == 63) {
          s_part[wave][j][0] = make_float4(r[0], r[1], r[2], r[3]);
          s_part[wave][j][1] = make_float4(r[4], r[5], r[6], r[7]);
          *reinterpret_cast<float2*>(&s_part[wave][j][2]) = make_float2(r[8], r[9]);
        }
      }
      u = un; v = vn; cgb = cn;
    }
    __syncthreads();
    if (tid < nb) {
      const float4 p0a = s_part[0][tid][0], p1a = s_part[0][tid][1], p2a = s_part[0][tid][2];
      const float4 p0b = s_part[1][tid][0], p1b = s_part[1][tid][1], p2b = s_part[1][tid][2];
      const float S1 = p0a.x + p0b.x, Sx = p0a.y + p0b.y, Sy = p0a.z + p0b.z;
      const float Sxx = p0a.w + p0b.w, Sxy = p1a.x + p1b.x, Syy = p1a.y + p1b.y;
      const float A = q1.x, Bc = q1.y, Cc = q1.z, o = q0.w;
      float4* dst = B.pair_grad + (size_t)slot * 3;
      dst[0] = make_float4(-(A * Sx + Bc * Sy), -(Cc * Sy + Bc * Sx), -0.5f * Sxx, -Sxy);
      dst[1] = make_float4(-0.5f * Syy, S1 / o, p1a.z + p1b.z, p1a.w + p1b.w);
      dst[2] = make_float4(p2a.x + p2b.x, p2a.y + p2b.y, 0.f, 0.f);
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
  launch("blend_bwd", k_blend_bwd, dim3(P.T), dim3(kBwdThreads), st, P, B);
  const int npre = (P.N + kPreBlock - 1) / kPreBlock;
  launch("preprocess_bwd", k_preprocess_bwd, dim3(npre), dim3(kPreBlock), st, P, B);
  launch("tau_reduce", k_tau_reduce, dim3(1), dim3(384), st, B, npre);
  return hipGetLastError() == hipSuccess ? MGS_OK : MGS_ERR_LAUNCH;
}

}  // namespace mgs
