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
#include "wave_reduce.h"

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

// Slot of a Gaussian's first pair: pairs of one Gaussian are contiguous and Gaussians are
// laid out in index order, so the gather pass streams.  The offsets come out of the forward
// binning (block-local scan + scan of the block totals), no scan is launched here.
__device__ __forceinline__ int pair_slot_base(const KP& P, int idx) {
  return P.block_prefix[idx / P.counters[2]] + P.pair_off[idx];
}

// ---------------------------------------------------------------------------------
// Blend backward, segment-parallel.  One wave per (tile, kSeg-splat segment) work item,
// FOUR pixels per lane (rows y, y+4, y+8, y+12 of one column: dx is shared, the fp32 work
// packs into v_pk_*).  The forward checkpointed the per-pixel blend state (T, prefix
// colour F) in front of every segment, so items are independent: no serial chain over a
// tile's whole list, ~D/kSeg equal-sized items instead of T ragged ones.
//
// Front-to-back replay inside the segment.  With S = (C_final + T_final bg) - F_i (colour
// still to come behind splat i, background included) the derivative of the pixel w.r.t. the
// splat's alpha is
//   dL/dalpha_i = sum_ch dL/dC_ch * (T_i c_ch - S_ch / (1 - alpha_i))
// and every screen-space gradient of the splat is a pixel sum of W = dL/dG * G or of the
// blend weight w = alpha T times a monomial of (dx, dy) / the pixel's upstream gradient:
//   S1 = sum W, Sx = sum W dx, Sy = sum W dy, Sxx, Sxy, Syy      (mean2D, conic, opacity)
//   Rr, Rg, Rb = sum w * dL/dC_ch,  Rd = sum w * dL/dD            (colour, depth)
// The lane adds its four pixels, the 10 sums are reduced over the wave with DPP, lane 63
// parks them in LDS, and after the loop each lane converts two splats' sums into the
// 40-B pair record and stores it ONCE at the pair's slot (plain stores, no atomics,
// fixed summation order => deterministic).
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr float kLog2eB = 1.4426950408889634f;

template <bool SKETCH>
__global__ __launch_bounds__(64) void k_blend_bwd(KP P, KB B) {
  __shared__ float4 s_r0[kSeg], s_r1[kSeg];
  __shared__ float2 s_r2[kSeg];
  __shared__ float4 s_out[kSeg][3];
  __shared__ float4 s_coef[SKETCH ? kSeg : 1][9];   // per splat: 6 tau components x 6 coefficients
  const int item = xcd_remap<kBwdChunk>(blockIdx.x), lane = threadIdx.x;
  if (item >= min(P.seg_offset[P.T], P.max_segs)) return;
  const int tile = P.seg_tile[item];
  const int seg = item - P.seg_offset[tile];
  int start = P.tile_offset[tile], end = P.tile_offset[tile + 1];
  start = min(start, P.cap); end = min(end, P.cap);
  const int base = seg * kSeg;
  const int nb = min(kSeg, end - start - base);
  if (nb <= 0) return;
  const int tx = tile % P.grid_x, ty = tile / P.grid_x;
  const int px = tx * kTile + (lane & 15);
  const int pyb = ty * kTile + (lane >> 4);
  const size_t HW = (size_t)P.W * P.H;

  // ---- stage the segment's records (2 per lane), remember slot + raw conic ---------
  constexpr int kStage = kSeg / 64;   // records staged per lane
  int slot[kStage];
  float4 qa[kStage], qb[kStage];
#pragma unroll
  for (int h = 0; h < kStage; h++) {
    slot[h] = -1;
    const int jj = lane + 64 * h;
    if (jj < nb) {
      const int k = start + base + jj;
      const unsigned int id = (unsigned int)P.keys[k];
      slot[h] = pair_slot_base(P, (int)id) + (int)P.payload[k];
      const float4* src = reinterpret_cast<const float4*>(P.rec + id);
      qa[h] = src[0]; qb[h] = src[1];
      const float4 q2 = src[2];
      s_r0[jj] = make_float4(qa[h].x, qa[h].y, -0.5f * kLog2eB * qb[h].x, -kLog2eB * qb[h].y);
      s_r1[jj] = make_float4(-0.5f * kLog2eB * qb[h].z, qa[h].w, qa[h].z, q2.x);
      s_r2[jj] = make_float2(q2.y, q2.z);
      if constexpr (SKETCH) {
        const float4* cj = reinterpret_cast<const float4*>(B.splat_jac + (size_t)id * 36);
#pragma unroll
        for (int i = 0; i < 9; i++) s_coef[jj][i] = cj[i];
      }
    }
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    s_out[jj][0] = z; s_out[jj][1] = z; s_out[jj][2] = z;
  }

  // ---- per-pixel data: pixel q of this lane is tile pixel lane + 64 q --------------------
  // State per pixel: T (transmittance in front of the next splat) and the scalar
  //   gS = sum_ch dL/dC_ch * S_ch,   S = (C_final + T_final * bg) - F
  // (S = colour/depth still to come BEHIND the splats visited so far, background included,
  // F = prefix colour from the checkpoint).  dL/dalpha only ever needs S through g.S, and
  // g.S updates with one FMA per splat (gS -= w * g.c), so the four S channels never live
  // in registers.
  int last[4];
  v2f gA0, gA1, gA2, gAd, gB0, gB1, gB2, gBd;        // dL/dC, dL/dD   (A: q=0,1  B: q=2,3)
  v2f gSA, gSB;                                       // g . S
  v2f TA = {1.f, 1.f}, TB = {1.f, 1.f};
  const float bg0 = P.bg[0], bg1 = P.bg[1], bg2 = P.bg[2];
  const float* ck = (seg > 0) ? P.ckpt + (size_t)item * (5 * 256) : nullptr;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int py = pyb + 4 * q;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, gd = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f, cd = 0.f;
    int l = 0;
    if (px < P.W && py < P.H) {
      const size_t pix = (size_t)py * P.W + px;
      l = P.n_contrib[pix];
      g0 = B.grad_color[pix]; g1 = B.grad_color[HW + pix]; g2 = B.grad_color[2 * HW + pix];
      if (B.grad_depth) gd = B.grad_depth[pix];
      const float tf = P.final_T[pix];
      c0 = P.final_C[pix] + tf * bg0; c1 = P.final_C[HW + pix] + tf * bg1;
      c2 = P.final_C[2 * HW + pix] + tf * bg2;
      cd = P.final_C[3 * HW + pix];
    }
    last[q] = l;
    float t = 1.f;
    if (ck) {
      const int p = lane + 64 * q;
      t = ck[p]; c0 -= ck[256 + p]; c1 -= ck[512 + p]; c2 -= ck[768 + p]; cd -= ck[1024 + p];
    }
    const int e = q & 1;
    if (q < 2) {
      gA0[e] = g0; gA1[e] = g1; gA2[e] = g2; gAd[e] = gd;
      gSA[e] = g0 * c0 + g1 * c1 + g2 * c2 + gd * cd; TA[e] = t;
    } else {
      gB0[e] = g0; gB1[e] = g1; gB2[e] = g2; gBd[e] = gd;
      gSB[e] = g0 * c0 + g1 * c1 + g2 * c2 + gd * cd; TB[e] = t;
    }
  }
  // does any pixel of the tile reach this segment?
  int ml = max(max(last[0], last[1]), max(last[2], last[3]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ml = max(ml, __shfl_xor(ml, off));
  if (base >= ml) {
#pragma unroll
    for (int h = 0; h < kStage; h++)
      if (slot[h] >= 0) {
        float4* dst = B.pair_grad + (size_t)slot[h] * 3;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        dst[0] = z; dst[1] = z; dst[2] = z;
      }
    return;
  }
  __syncthreads();

  v2f JA[6], JB[6];   // sketch mode: per-pixel pose-Jacobian rows of this segment
#pragma unroll
  for (int t = 0; t < 6; t++) { JA[t] = v2f{0.f, 0.f}; JB[t] = v2f{0.f, 0.f}; }
  // reduce-scatter bookkeeping (wave_reduce.h): which of the ten sums this lane ends up with
  const unsigned long long b3mask = __ballot((lane & 8) != 0);
  const bool wextra = lane == 31 || lane == 63;
  const int wofs = wextra ? (lane == 31 ? 8 : 9)
                          : ((lane & 7) == 0 ? ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1) + 4 * ((lane >> 3) & 1) : -1);
  const float fpx = (float)px;
  const v2f fpyA = {(float)pyb, (float)(pyb + 4)}, fpyB = {(float)(pyb + 8), (float)(pyb + 12)};
  float4 u = s_r0[0], v = s_r1[0];
  float2 cgb = s_r2[0];
  for (int j = 0; j < nb; j++) {
    const int jn = min(j + 1, nb - 1);
    const float4 un = s_r0[jn], vn = s_r1[jn];
    const float2 cn = s_r2[jn];
    const int idx = base + j;
    const float dx = u.x - fpx;
    const float tA = u.z * dx;
    const v2f dyA = v2f{u.y, u.y} - fpyA, dyB = v2f{u.y, u.y} - fpyB;
    const v2f pwA = dx * (tA + u.w * dyA) + v.x * dyA * dyA;
    const v2f pwB = dx * (tA + u.w * dyB) + v.x * dyB * dyB;
    const v2f arA = v.y * v2f{__builtin_amdgcn_exp2f(pwA.x), __builtin_amdgcn_exp2f(pwA.y)};
    const v2f arB = v.y * v2f{__builtin_amdgcn_exp2f(pwB.x), __builtin_amdgcn_exp2f(pwB.y)};
    const v2f alA = v2f{fminf(kAlphaMax, arA.x), fminf(kAlphaMax, arA.y)};
    const v2f alB = v2f{fminf(kAlphaMax, arB.x), fminf(kAlphaMax, arB.y)};
    const bool k0 = idx < last[0] && pwA.x <= 0.f && alA.x >= kAlphaMin;
    const bool k1 = idx < last[1] && pwA.y <= 0.f && alA.y >= kAlphaMin;
    const bool k2 = idx < last[2] && pwB.x <= 0.f && alB.x >= kAlphaMin;
    const bool k3 = idx < last[3] && pwB.y <= 0.f && alB.y >= kAlphaMin;
    if (__ballot(k0 || k1 || k2 || k3) != 0ull) {
      const v2f aeA = v2f{k0 ? alA.x : 0.f, k1 ? alA.y : 0.f};
      const v2f aeB = v2f{k2 ? alB.x : 0.f, k3 ? alB.y : 0.f};
      const v2f wA = aeA * TA, wB = aeB * TB;
      // g . c (c is the splat's colour/depth, uniform over the wave)
      const v2f gcA = gA0 * v.w + gA1 * cgb.x + gA2 * cgb.y + gAd * v.z;
      const v2f gcB = gB0 * v.w + gB1 * cgb.x + gB2 * cgb.y + gBd * v.z;
      gSA -= wA * gcA; gSB -= wB * gcB;
      const v2f omA = 1.f - aeA, omB = 1.f - aeB;
      const v2f roA = v2f{__builtin_amdgcn_rcpf(omA.x), __builtin_amdgcn_rcpf(omA.y)};
      const v2f roB = v2f{__builtin_amdgcn_rcpf(omB.x), __builtin_amdgcn_rcpf(omB.y)};
      const v2f dA = TA * gcA - roA * gSA, dB = TB * gcB - roB * gSB;
      TA *= omA; TB *= omB;
      v2f WA = arA * dA, WB = arB * dB;
      WA = v2f{k0 ? WA.x : 0.f, k1 ? WA.y : 0.f};
      WB = v2f{k2 ? WB.x : 0.f, k3 ? WB.y : 0.f};
      const v2f WyA = WA * dyA, WyB = WB * dyB;
      const v2f WyyA = WyA * dyA + WyB * dyB;
      const v2f Ws2 = WA + WB, Sy2 = WyA + WyB;
      const v2f r6 = wA * gA0 + wB * gB0, r7 = wA * gA1 + wB * gB1;
      const v2f r8 = wA * gA2 + wB * gB2, r9 = wA * gAd + wB * gBd;
      if constexpr (SKETCH) {
        // J_t += W (c0 dx + c1 dy + c2 dx^2 + c3 dx dy + c4 dy^2) + (w dL/dD) c5
        const v2f XA1 = WA * dx, XB1 = WB * dx;
        const v2f XA3 = XA1 * dx, XB3 = XB1 * dx, XA4 = XA1 * dyA, XB4 = XB1 * dyB;
        const v2f XA5 = WyA * dyA, XB5 = WyB * dyB;
        const v2f XA6 = wA * gAd, XB6 = wB * gBd;
        const float* cf = reinterpret_cast<const float*>(&s_coef[j][0]);
#pragma unroll
        for (int t = 0; t < 6; t++) {
          const float c0 = cf[6 * t], c1 = cf[6 * t + 1], c2 = cf[6 * t + 2], c3 = cf[6 * t + 3],
                      c4 = cf[6 * t + 4], c5 = cf[6 * t + 5];
          JA[t] += c0 * XA1 + c1 * WyA + c2 * XA3 + c3 * XA4 + c4 * XA5 + c5 * XA6;
          JB[t] += c0 * XB1 + c1 * WyB + c2 * XB3 + c3 * XB4 + c4 * XB5 + c5 * XB6;
        }
      }
      float r[10];
      const float Ws = Ws2.x + Ws2.y, Sy = Sy2.x + Sy2.y;
      r[0] = Ws; r[1] = Ws * dx; r[2] = Sy; r[3] = r[1] * dx; r[4] = Sy * dx;
      r[5] = WyyA.x + WyyA.y;
      r[6] = r6.x + r6.y; r[7] = r7.x + r7.y; r[8] = r8.x + r8.y; r[9] = r9.x + r9.y;
      float mres, eres;
      wave_sum10_scatter(r, b3mask, mres, eres);
      if (wofs >= 0) reinterpret_cast<float*>(&s_out[j][0])[wofs] = wextra ? eres : mres;
    }
    u = un; v = vn; cgb = cn;
  }
  if constexpr (SKETCH) {
    // pixel rows of different segments of a tile meet in pix_jac: float atomics, planar
    // [6][H*W] so a wave instruction covers 16-pixel runs of contiguous addresses
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int py = pyb + 4 * q;
      if (px < P.W && py < P.H) {
        const size_t pix = (size_t)py * P.W + px;
#pragma unroll
        for (int t = 0; t < 6; t++) {
          const float val = (q < 2) ? JA[t][q & 1] : JB[t][q & 1];
          atomicAdd(&B.pix_jac[(size_t)t * HW + pix], val);
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < kStage; h++) {
    if (slot[h] >= 0) {
      const int jj = lane + 64 * h;
      const float4 p0 = s_out[jj][0], p1 = s_out[jj][1], p2 = s_out[jj][2];
      const float S1 = p0.x, Sx = p0.y, Sy = p0.z, Sxx = p0.w, Sxy = p1.x, Syy = p1.y;
      const float A = qb[h].x, Bc = qb[h].y, Cc = qb[h].z, o = qa[h].w;
      float4* dst = B.pair_grad + (size_t)slot[h] * 3;
      dst[0] = make_float4(-(A * Sx + Bc * Sy), -(Cc * Sy + Bc * Sx), -0.5f * Sxx, -Sxy);
      dst[1] = make_float4(-0.5f * Syy, S1 / o, p1.z, p1.w);
      dst[2] = make_float4(p2.x, p2.y, 0.f, 0.f);
    }
  }
}

// ---------------------------------------------------------------------------------
// Sketched pose Jacobian (rogerhh fork: slam_frontend.py:269-338 producer, :654-669
// consumer; contract row a9).  For backward call #r the extension must return
//   grad_sketch_dtau[s, k, :] = sum over pixels p with sketch_indices[r, s, p] == k of
//                               dL/dpixel_p . d pixel_p / d tau          (tau = [rho; theta])
// i.e. bucket sums of per-PIXEL Jacobian rows, which a per-Gaussian reduction cannot
// give.  Three steps:
//   k_sketch_prep    per Gaussian: d(x, y, A, B, C, depth)/d tau (6 x 6) by six unit-gradient
//                    calls of the same chain used for the ordinary backward, folded with the
//                    conic into 36 polynomial coefficients
//   k_blend_bwd<1>   per (pixel, splat): J_p += W * poly(dx, dy) + w dL/dD * c  (6 components)
//   k_sketch_bucket  per pixel: J_p -> LDS-privatised bucket table -> grad_sketch_dtau
__global__ __launch_bounds__(kPreBlock) void k_sketch_prep(KP P, KB B) {
  const int idx = blockIdx.x * kPreBlock + threadIdx.x;
  if (idx >= P.N) return;
  const float4 r1 = reinterpret_cast<const float4*>(P.rec + idx)[1];
  float* out = B.splat_jac + (size_t)idx * 36;
  if (__float_as_int(r1.w) <= 0) return;    // never staged by a blend item
  Camera cam;
  load_camera_b(cam, P);
  const float p[3] = {P.means[3 * idx], P.means[3 * idx + 1], P.means[3 * idx + 2]};
  float sc[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool has_cov = P.covp != nullptr;
  if (has_cov) {
#pragma unroll
    for (int i = 0; i < 6; i++) c6[i] = P.covp[6 * (size_t)idx + i];
  } else {
    sc[0] = P.scales[3 * idx]; sc[1] = P.scales[3 * idx + 1]; sc[2] = P.scales[3 * idx + 2];
    const float4 qq = reinterpret_cast<const float4*>(P.rots)[idx];
    q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
  }
  float M[6][6];   // rows: x, y, A, B, C, depth ; columns: tau
#pragma unroll
  for (int row = 0; row < 6; row++) {
    const float g_xy[2] = {row == 0 ? 1.f : 0.f, row == 1 ? 1.f : 0.f};
    const float g_con[3] = {row == 2 ? 1.f : 0.f, row == 3 ? 1.f : 0.f, row == 4 ? 1.f : 0.f};
    GaussGrad gg;
    if (has_cov) project_gaussian_backward(cam, p, nullptr, nullptr, c6, g_xy, g_con, 0.f, row == 5 ? 1.f : 0.f, gg);
    else project_gaussian_backward(cam, p, sc, q, nullptr, g_xy, g_con, 0.f, row == 5 ? 1.f : 0.f, gg);
#pragma unroll
    for (int t = 0; t < 6; t++) M[row][t] = gg.dtau[t];
  }
  const float A = r1.x, Bc = r1.y, Cc = r1.z;
#pragma unroll
  for (int t = 0; t < 6; t++) {
    out[6 * t + 0] = -(A * M[0][t] + Bc * M[1][t]);
    out[6 * t + 1] = -(Bc * M[0][t] + Cc * M[1][t]);
    out[6 * t + 2] = -0.5f * M[2][t];
    out[6 * t + 3] = -M[3][t];
    out[6 * t + 4] = -0.5f * M[4][t];
    out[6 * t + 5] = M[5][t];
  }
}

constexpr int kBucketBlocks = 128;

__global__ __launch_bounds__(256) void k_sketch_bucket(KP P, KB B) {
  extern __shared__ float s_acc[];    // stack * sketch * 6
  const int nacc = B.stack_dim * B.sketch_dim * 6;
  const size_t HW = (size_t)P.W * P.H;
  for (int i = threadIdx.x; i < nacc; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < HW; p += (size_t)gridDim.x * 256) {
    float J[6];
#pragma unroll
    for (int t = 0; t < 6; t++) J[t] = B.pix_jac[(size_t)t * HW + p];
    for (int s = 0; s < B.stack_dim; s++) {
      const int k = B.sketch_idx[(size_t)s * HW + p];
      if (k >= 0 && k < B.sketch_dim) {
        float* a = &s_acc[(s * B.sketch_dim + k) * 6];
#pragma unroll
        for (int t = 0; t < 6; t++) atomicAdd(&a[t], J[t]);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nacc; i += 256) {
    const float v = s_acc[i];
    if (v != 0.f) atomicAdd(&B.g_sketch[i], v);
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
      const int s0 = pair_slot_base(P, idx), s1 = s0 + P.pair_count[idx];
      for (int s = s0; s < s1; s++) {
        const float4* src = B.pair_grad + (size_t)s * 3;
        const float4 x = src[0], y = src[1], z = src[2];
        a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w;
        a[4] += y.x; a[5] += y.y; a[6] += y.z; a[7] += y.w;
        a[8] += z.x; a[9] += z.y;
      }
      Camera cam;
      load_camera_b(cam, P);
      const float g_xy[2] = {a[0], a[1]};
      const float g_con[3] = {a[2], a[3], a[4]};
      GaussGrad gg;
      if (P.covp) {
        float c6[6];
#pragma unroll
        for (int i = 0; i < 6; i++) c6[i] = P.covp[6 * (size_t)idx + i];
        project_gaussian_backward(cam, p, nullptr, nullptr, c6, g_xy, g_con, a[5], a[9], gg);
      } else {
        const float sc[3] = {P.scales[3 * idx], P.scales[3 * idx + 1], P.scales[3 * idx + 2]};
        const float4 qq = reinterpret_cast<const float4*>(P.rots)[idx];
        const float q[4] = {qq.x, qq.y, qq.z, qq.w};
        project_gaussian_backward(cam, p, sc, q, nullptr, g_xy, g_con, a[5], a[9], gg);
      }
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
  if (B.sketch_mode != 0) {
    const size_t HW = (size_t)P.W * P.H;
    const size_t nacc = (size_t)B.stack_dim * B.sketch_dim * 6;
    if (nacc * sizeof(float) > 64 * 1024) return MGS_ERR_UNSUPPORTED;
    if (hipMemsetAsync(B.pix_jac, 0, HW * 6 * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(B.g_sketch, 0, nacc * sizeof(float), st) != hipSuccess)
      return MGS_ERR_LAUNCH;
    launch("sketch_prep", k_sketch_prep, dim3((P.N + kPreBlock - 1) / kPreBlock), dim3(kPreBlock), st, P, B);
    launch("blend_bwd_sketch", k_blend_bwd<true>, dim3(grid_pad(P.max_segs, kBwdChunk)), dim3(64), st, P, B);
    launch_smem("sketch_bucket", k_sketch_bucket, dim3(kBucketBlocks), dim3(256), nacc * sizeof(float), st, P, B);
  } else {
    launch("blend_bwd", k_blend_bwd<false>, dim3(grid_pad(P.max_segs, kBwdChunk)), dim3(64), st, P, B);
  }
  const int npre = (P.N + kPreBlock - 1) / kPreBlock;
  launch("preprocess_bwd", k_preprocess_bwd, dim3(npre), dim3(kPreBlock), st, P, B);
  launch("tau_reduce", k_tau_reduce, dim3(1), dim3(384), st, B, npre);
  return hipGetLastError() == hipSuccess ? MGS_OK : MGS_ERR_LAUNCH;
}

}  // namespace mgs
