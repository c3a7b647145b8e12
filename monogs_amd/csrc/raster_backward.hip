// Backward pass of the rasteriser for gfx950 (CDNA4, wave64).  Atomic-free on the gradient path:
//   k_blend_bwd      1 wave / (tile, 32-splat item), 4 pixels per lane (one per 8x8 quadrant):
//                    front-to-back replay from the forward's per-item checkpoint; per splat the
//                    ten screen-space sums are reduced over the wave in registers (permlane swap +
//                    DPP reduce-scatter) and stored ONCE, with plain stores, at the pair's slot
//   k_preprocess_bwd 1 thread / Gaussian (the wave stages its contiguous run of pair records in LDS), chains to
//                    means3D / scale / rot / SH / opacity / means2D and the per-Gaussian pose
//                    gradient, block-reduced to one partial per workgroup; in mapping mode it also
//                    chains through the model's activations and accumulates over the views
//   k_tau_reduce     fixed-order sum of the partials -> grad_tau[6]
// (slot offsets come out of the forward's binning pass; no scan is launched here.)
//
// Replaces rasterize_gaussians_backward of the reference's CUDA extension (its
// autograd.Function is invoked through gaussian_renderer/__init__.py:151-168; gradient
// sinks: gaussian_model.py:252-285,693-697, slam_frontend.py:365-378,606-611).
#include <cstdlib>
#include "launch.h"
#include "raster_kernels.h"
#include "sketch_kernels.h"
#include "wave_reduce.h"
#define MGS_DIAG_BACKWARD
#include "diag_stamp.h"

namespace mgs {

__device__ __forceinline__ void load_camera_b(Camera& c, const KP& P) {
#pragma unroll
  for (int i = 0; i < 16; i++) { c.V[i] = P.V[i]; c.PM[i] = P.PM[i]; c.Praw[i] = P.Praw[i]; }
  c.campos[0] = P.campos[0]; c.campos[1] = P.campos[1]; c.campos[2] = P.campos[2];
  c.W = P.W; c.H = P.H; c.tanfovx = P.tanfovx; c.tanfovy = P.tanfovy;
  c.focal_x = P.focal_x; c.focal_y = P.focal_y; c.scale_modifier = P.mod;
  c.sh_degree = P.deg; c.sh_coeffs = P.K; c.grid_x = P.grid_x; c.grid_y = P.grid_y;
  c.clamp_grad_upstream = P.clamp_up;
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

// Slot of a Gaussian's first pair: pairs of one Gaussian are contiguous and Gaussians are
// laid out in index order, so the gather pass streams.  The offsets come out of the forward
// binning (block-local scan + scan of the block totals), no scan is launched here.
__device__ __forceinline__ int pair_slot_base(const KP& P, int idx) {
  return P.block_prefix[idx / P.per_block] + P.pair_off[idx];
}

// ---------------------------------------------------------------------------------
// Blend backward, item-parallel.  One wave per work item = kItem consecutive splats of one tile's
// list, FOUR pixels per lane, one in each 8x8 quadrant of the tile.  The lane that stages a splat
// also evaluates the culling bound on the four quadrant boxes; the wave then visits, per
// splat, only the quadrants that can be reached (wave-uniform scalar bit tests on ballot masks).
// The quadrant body is written on float2 operands so that it maps onto v_pk_{add,mul,fma}_f32
// (measured on gfx950: a packed FMA issues in about 1.25x the time of a scalar one).  The forward
// checkpointed the per-pixel blend state (T, prefix colour F) in front of every segment, so items
// are independent: no serial chain over a tile's whole list, ~D/kItem similar-sized items instead
// of T ragged ones.
//
// Front-to-back replay inside the segment.  With S = (C_final + T_final bg) - F_i (colour
// still to come behind splat i, background included) the derivative of the pixel w.r.t. the
// splat's alpha is
//   dL/dalpha_i = sum_ch dL/dC_ch * (T_i c_ch - S_ch / (1 - alpha_i))
// and every screen-space gradient of the splat is a pixel sum of W = dL/dG * G or of the
// blend weight w = alpha T times a monomial of (dx, dy) / the pixel's upstream gradient:
//   S1 = sum W, Sx = sum W dx, Sy = sum W dy, Sxx, Sxy, Syy      (mean2D, conic, opacity)
//   Rr, Rg, Rb = sum w * dL/dC_ch,  Rd = sum w * dL/dD            (colour, depth)
// The lane adds its four pixels, the 10 sums are reduced over the wave in registers
// (wave_reduce.h: the totals land in ten different lanes) and those lanes store their dword of
// the pair's 40-B record at the pair's slot in ONE store instruction (plain stores, no atomics,
// fixed summation order => deterministic).  The per-Gaussian conic / opacity map the raw sums to
// screen-space gradients once, in k_preprocess_bwd.
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr float kLog2eB = 1.4426950408889634f;

// JONLY (sketch mode only): nothing but the per-pixel pose-Jacobian rows is wanted - the
// second-order tracking iteration uses grad_sketch_dtau alone (slam_frontend.py:654-669), so
// the per-splat sums, their reduction and the pair records are skipped altogether.
// POSE (plain mode only): pose-only backward (tracking: every per-Gaussian gradient pointer is
// NULL).  dL/dtau needs the mean / conic / depth sums only, so the colour and opacity sums are
// not formed and six values instead of ten are reduced per splat.

// Orders this wave's LDS writes before its own later LDS reads (the staged records belong to one
// wave: no workgroup barrier is wanted).  LDS operations of one wave execute in order; the waitcnt +
// memory clobber keep the compiler from moving accesses across.
// v_min_f32 without the canonicalising v_max the compiler puts in front of fminf
// (a: a wave-uniform bound, kept in a scalar register)
__device__ __forceinline__ float min_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "s"(a), "v"(b)); return r; }

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <bool SKETCH, bool JONLY = false, bool POSE = false>
__global__ __launch_bounds__(64, SKETCH ? (JONLY ? 4 : 3) : 6) void k_blend_bwd(KP P, KB B) {
  MGS_BSTAMP;
  static_assert(SKETCH || !JONLY, "JONLY is a sketch-mode variant");
  static_assert(!(SKETCH && POSE), "POSE is a plain-mode variant");
  static_assert(kSeg == 64, "one staged record per lane");
  __shared__ float4 s_r0[kSeg], s_r1[kSeg];
  __shared__ float2 s_r2[kSeg];             // 2560 B of LDS in all
  // per splat: 6 features x 6 tau components.  An item holds at most kItem (32) splats - half a 64-lane segment:
  // sized for that (4.6 KB instead of 9.2 KB per wave), the staging no longer caps a CU at 13 waves (round 4).
  __shared__ float4 s_coef[SKETCH ? kItem : 1][9];
  // (Round 4 tried the scalar path for them - the 36 coefficients of a splat are wave-uniform: s_load_dwordx16 from
  // the constant address space into SGPRs, v_pk_fma_f32 with scalar operands, no LDS staging: 124 -> 115 VGPRs, 106
  // SGPRs with 4 spilled, and 176.6 -> 185.8 us: the loads cannot be issued a splat ahead (72 SGPRs) and their
  // latency sits in the walk.  profiles/r04_backward_blend_tuning.txt.)
  // Sketch mode: a workgroup takes kSketchReps consecutive items (mostly of one tile) and leaves its per-pixel
  // Jacobian rows as ONE slab per run of items of a tile (plain coalesced stores, flush_jacobian below) instead of
  // one per item; k_sketch_bucket adds up a tile's slabs.
  constexpr int kReps = SKETCH ? kSketchReps : 1;
  const int n_items = min(P.seg_offset[P.T], P.max_segs);
  int item_first = xcd_remap<kBwdChunk>(blockIdx.x) * kReps;
  const int lane = threadIdx.x;
  // A grid sized from a pair_count_bound BELOW the forward's pair count leaves the trailing items unwalked (their pair
  // records unwritten): say so in counters[2] (zeroed by every forward; include/monogs_raster.h: mgs_backward_args).
  if (blockIdx.x == 0 && lane == 0 && (long long)gridDim.x * kReps < (long long)n_items) P.counters[2] = n_items;
  MGS_BORDER(item_first, SKETCH);
  if (item_first < 0) return;
  if (item_first >= n_items) return;
  // (The grid covers every work item: capacity / kItem + T of them, or pair_count_bound / kItem + T when the caller
  // knows the forward's pair count.  A grid-stride loop over the items - which would make ANY grid safe, e.g. one
  // sized from the previous view's count - cost this kernel 8 VGPRs and its seventh wave per SIMD: not shipped.)
  // sketch mode: per-pixel pose-Jacobian rows, as pairs (tau 0,1) (2,3) (4,5), of the tile in hand
  v2f J2[SKETCH ? 4 : 1][3];
#pragma unroll
  for (int q = 0; q < (SKETCH ? 4 : 1); q++)
#pragma unroll
    for (int t = 0; t < 3; t++) J2[q][t] = v2f{0.f, 0.f};
  unsigned int jq_mask = 0u;       // quadrants in which this lane's rows received a contribution
  int jtile = -1, jitem = 0;       // the tile in hand and the slab of this wave's run in it (slab_index of the run's first item)
  auto flush_jacobian = [&]() {
    if constexpr (SKETCH) {
      if (jtile < 0) return;
      // This run's rows as one slab, float[6][256] in quadrant-major order (entry t * 256 + 64 q + lane): plain
      // 256-B stores, only the quadrants that received something; the mask word says which.  EVERY run start of
      // the forward's item structure gets its word from this launch (also an all-zero one), so the reader never
      // sees a stale word and nothing has to be cleared between launches.
      // (jq_mask is per lane - a lane sets a bit only where it blended something: the wave's union decides what
      // is stored, and every lane stores its row, zeros included)
      unsigned int wmask = 0u;
#pragma unroll
      for (int q = 0; q < 4; q++) wmask |= (__ballot((jq_mask >> q) & 1u) != 0ull ? 1u : 0u) << q;
      float* slab = B.slabs + (size_t)jitem * (6 * 256) + lane;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if ((wmask >> q) & 1u) {                    // wave-uniform
#pragma unroll
          for (int t = 0; t < 6; t++) slab[t * 256 + 64 * q] = (t & 1) ? J2[q][t >> 1].y : J2[q][t >> 1].x;
        }
#pragma unroll
        for (int t = 0; t < 3; t++) J2[q][t] = v2f{0.f, 0.f};
      }
      jq_mask = wmask;
      if (lane == 0) B.slab_mask[jitem] = jq_mask;
      jq_mask = 0u;
    }
  };
  for (int rep = 0; rep < kReps; rep++) {
  const int item = item_first + rep;
  if (item >= n_items) break;
  if (rep > 0) wave_lds_fence();     // orders the reuse of the staged records
  // One 16-B record per item (written by the tile sort) instead of a chain of dependent loads:
  // tile, index of the item's first key, number of splats (<= kItem), position in the tile's list.
  const int4 sr = P.seg_rec[item];
  const int tile = sr.x, k0 = sr.y, nb = sr.z, base = sr.w;
  // ... and, beside it, which of the item's splats can reach each quadrant of the tile: the 32-bit halves of the
  // ballots the forward's quadrant waves formed over their 64-splat segments (exact box test, raster_forward.hip).
  // Until round 5 every staged lane repeated the four box tests here: ~12 of the kernel's 73 VALU instructions per pair.
  const uint4 rw = P.reach[item];
  MGS_BFINE(0, "s_waitcnt lgkmcnt(0)");
  MGS_BITEM(item, base);
  if constexpr (SKETCH) {
    if (tile != jtile) { flush_jacobian(); jtile = tile; jitem = slab_index(item, tile); }
  }
  if (nb <= 0) continue;
  __builtin_assume(nb <= kItem);
  const int tx = tile % P.grid_x, ty = tile / P.grid_x;
  // Pixel q of this lane lies in QUADRANT q of the tile: (qx + 8 (q & 1), qy + 8 (q >> 1)).
  const int qx = tx * kTile + (lane & 7), qy = ty * kTile + (lane >> 3);
  const size_t HW = (size_t)P.W * P.H;
  // the key of this lane's first splat is requested first; the ~60 per-pixel loads below are
  // issued while it is in flight, and the record gather that depends on it comes after them
  unsigned int lo_next = lane < nb ? (unsigned int)P.keys[k0 + lane] : 0u;

  // ---- per-pixel state (loaded once per item, carried across its segments in registers) ------
  // T (transmittance in front of the next splat) and the scalar
  //   gS = sum_ch dL/dC_ch * S_ch,   S = (C_final + T_final * bg) - F
  // (S = colour/depth still to come BEHIND the splats visited so far, background included,
  // F = prefix colour from the checkpoint).  dL/dalpha only ever needs S through g.S, and
  // g.S updates with one FMA per splat (gS -= w * g.c).
  // per quadrant: how many of this item's positions lie in front of the pixel's last contribution
  // (n_contrib - base, clamped to [0, 255]), one byte each: ONE register instead of four
  static_assert(kItem <= 255, "packed per-quadrant list ends");
  unsigned int lastp = 0u;
  float g0[4], g1[4], g2[4], gd[4], T[4], gS[4];
  const float bg0 = P.bg[0], bg1 = P.bg[1], bg2 = P.bg[2];
  const float* ck = (MGS_ABL_CKPT && base > 0) ? P.ckpt + (size_t)item * (5 * 256) : nullptr;
  // Last list position that still contributes anywhere in each quadrant (stored by the forward's
  // quadrant waves; a quadrant outside the image was never rendered).  A splat behind it cannot
  // contribute in that quadrant - the forward had stopped visiting the saturated quadrant - so its
  // reach bit is dropped at staging time, and a quadrant that is saturated in front of this whole
  // item loads nothing at all.
  int qlast[4];
  {
    const int4 ql = *reinterpret_cast<const int4*>(P.quad_last + 4 * tile);    // wave-uniform: one scalar load
    qlast[0] = ql.x; qlast[1] = ql.y; qlast[2] = ql.z; qlast[3] = ql.w;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (tx * kTile + 8 * (q & 1) >= P.W || ty * kTile + 8 * (q >> 1) >= P.H) qlast[q] = 0;
  }
  const int tile_last = max(max(qlast[0], qlast[1]), max(qlast[2], qlast[3]));
  MGS_BFINE(1, "s_waitcnt lgkmcnt(0)");
#pragma unroll
  for (int q = 0; q < 4; q++) {
    g0[q] = g1[q] = g2[q] = gd[q] = 0.f;
    T[q] = 0.f;
    gS[q] = 0.f;
  }
  if (tile_last > base) {      // wave-uniform; an item behind the tile's last contribution loads nothing
    // All loads of the four quadrants are issued back to back, branch-free (addresses of pixels
    // outside the image are clamped, quadrants that are already saturated are loaded anyway and
    // masked below), and only then consumed: one memory round trip for the item's per-pixel state.
    // (With a branch per quadrant the compiler drained the loads quadrant by quadrant: eight
    // round trips in a row at the head of every item.)
    int2 dl[4];
    float4 tc[4];
    bool in_img[4];
    // A quadrant that is saturated in front of this item is masked below anyway: it loads a LIVE quadrant's lines
    // in its place (wave-uniform select of the quadrant index), so that the dead quadrant's checkpoint, per-pixel
    // state and gradients are not fetched from HBM (round 4: FETCH_SIZE 116.4 -> 110.6 MB per launch, same time).
    int qsel[4];
    {
      int qlive = 0;
#pragma unroll
      for (int q = 3; q >= 0; q--) if (qlast[q] > base) qlive = q;
#pragma unroll
      for (int q = 0; q < 4; q++) qsel[q] = qlast[q] > base ? q : qlive;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int px = qx + 8 * (q & 1), py = qy + 8 * (q >> 1);
      in_img[q] = px < P.W && py < P.H;
      const int pxl = qx + 8 * (qsel[q] & 1), pyl = qy + 8 * (qsel[q] >> 1);
      const size_t pix = (size_t)min(pyl, P.H - 1) * P.W + min(pxl, P.W - 1);
      const size_t qi = (size_t)tile * 256 + 64 * qsel[q] + lane;     // quadrant-major: coalesced
      dl[q] = P.final_DL[qi];
      g0[q] = B.grad_color[pix]; g1[q] = B.grad_color[HW + pix]; g2[q] = B.grad_color[2 * HW + pix];
      gd[q] = B.grad_depth ? B.grad_depth[pix] : 0.f;
      tc[q] = P.final_TC[qi];
    }
    float4 k4[4];
    float k3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; q++) k4[q] = make_float4(1.f, 0.f, 0.f, 0.f);
    if (ck) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int p = 64 * qsel[q] + lane;
        k4[q] = reinterpret_cast<const float4*>(ck)[p];
        k3[q] = ck[1024 + p];
      }
    }

    MGS_BFINE(2, "s_waitcnt vmcnt(0)");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (!in_img[q]) { g0[q] = 0.f; g1[q] = 0.f; g2[q] = 0.f; gd[q] = 0.f; }
      const float tf = tc[q].x;
      const int ln_q = dl[q].y;
      const float c0 = tc[q].y + tf * bg0 - k4[q].y, c1 = tc[q].z + tf * bg1 - k4[q].z;
      const float c2 = tc[q].w + tf * bg2 - k4[q].w, cd = __int_as_float(dl[q].x) - k3[q];
      const float gs = g0[q] * c0 + g1[q] * c1 + g2[q] * c2 + gd[q] * cd;
      // a pixel whose list ended in front of this item contributes nothing here (and the forward
      // stops checkpointing a quadrant once all its pixels are saturated: never let that
      // unwritten state into the arithmetic); the same for a quadrant that is done as a whole
      const bool on = qlast[q] > base && ln_q > base;
      lastp |= (unsigned int)min(max((qlast[q] > base ? ln_q : 0) - base, 0), 255) << (8 * q);
      T[q] = on ? k4[q].x : 0.f;
      gS[q] = on ? gs : 0.f;
      if (qlast[q] <= base) { g0[q] = 0.f; g1[q] = 0.f; g2[q] = 0.f; gd[q] = 0.f; }
    }
  }

  MGS_BMARK(0);
  MGS_BFINE(3, "");
  // reduce-scatter bookkeeping (wave_reduce.h): which of the ten sums this lane ends up with
  const unsigned long long b3mask = __ballot((lane & 8) != 0);
  const bool wextra = lane == 31 || lane == 63;
  // dword of the pair record this lane stores after the reduction (-1: none).  Record layout:
  // S1 | Sx Sy | Sxx Sxy | Syy | Rr Rg Rb | Rd.  POSE reduces (Sx, Sy, Sxx, Sxy | Syy, Rd) only.
  const int wofs_ = POSE ? (wextra ? (lane == 31 ? 5 : 9)
                                  : ((lane & 15) == 0 ? 1 + ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1) : -1))
                        : (wextra ? (lane == 31 ? 8 : 9)
                                  : ((lane & 7) == 0 ? ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1) + 4 * ((lane >> 3) & 1) : -1));
  // byte offset of that dword as an unsigned register (0x80000000...: none): ONE VGPR for offset and
  // predicate - a 64-bit per-lane address and a sign-extended index cost four and spilled at 7 waves / SIMD
  const unsigned int wofs4 = wofs_ >= 0 ? (unsigned int)wofs_ * 4u : kPairBufferExtent;
  // pair_grad as a raw buffer of kPairBufferExtent bytes (the launch refuses a larger one): offsets from
  // kPairBufferExtent on are out of range (the range check covers soffset + voffset; checked on gfx950)
  const __amdgpu_buffer_rsrc_t pair_rsrc = __builtin_amdgcn_make_buffer_rsrc(B.pair_grad, 0, kPairBufferExtent, 0x00020000);
  // Packed operands: the per-quadrant body is written on float2 values so that it maps onto
  // v_pk_{add,mul,fma}_f32 without register shuffles (measured on gfx950: a packed FMA issues
  // in about the time of a scalar one, so pairs of independent FMAs halve their issue cost).
  v2f Pq[4], G01[4], G2d[4];
  const v2f P0 = {(float)qx, (float)qy};       // this lane's pixel in quadrant 0; quadrant q adds (8 (q & 1), 8 (q >> 1))
#pragma unroll
  for (int q = 0; q < 4; q++) {
    Pq[q] = v2f{(float)(8 * (q & 1)), (float)(8 * (q >> 1))};       // compile-time constants: no registers
    G01[q] = v2f{g0[q], g1[q]};
    G2d[q] = v2f{g2[q], gd[q]};
  }

  // per-segment state of the walk
  int slot = -1, sub_base = base;
  unsigned long long mq[4] = {0ull, 0ull, 0ull, 0ull};
  unsigned long long written = 0ull;
  // one splat: (u, v, bd2) = its staged record; mq = the quadrants the segment's splats reach
  auto visit = [&](int j, const float4 u, const float4 v, const float2 bd2) {
    const v2f mu = v2f{u.x, u.y} - P0, RG = {v.z, v.w}, BD = {bd2.x, bd2.y};
    const int idx = sub_base + j;
    const v2f* cf2 = reinterpret_cast<const v2f*>(&s_coef[SKETCH ? j : 0][0]);   // [feature][tau pair]
    // pixel sums of this splat: S1 | (Sx, Sy) | (Sxx, Sxy) | Syy | (Rr, Rg) | (Rb, Rd)
    float r0 = 0.f, r5 = 0.f;
    v2f R12 = {0.f, 0.f}, R34 = {0.f, 0.f}, R67 = {0.f, 0.f}, R89 = {0.f, 0.f};
    bool any = false;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (!((mq[q] >> j) & 1ull)) continue;          // wave-uniform (scalar bit test)
      const v2f d = mu - Pq[q];
      // clamped at 0 like the forward's form of the exponent (the quadratic form is <= 0; only
      // rounding can make it positive)
      const float pw = d.x * (u.z * d.x + u.w * d.y) + v.x * d.y * d.y;
      const float ar = v.y * exp2_sat(pw);
      const bool k = (unsigned int)(idx - base) < ((lastp >> (8 * q)) & 0xffu) && ar >= kAlphaMin;
      MGS_BLANES(k);
      if (__ballot(k) == 0ull) continue;              // wave-uniform
      any = true;
      // a: opacity * G of the lanes that blend the splat.  The blended alpha is its clamp at 0.99 and the
      // gradient flows through the UNclamped product (as upstream: dL/dG = opacity * dL/dalpha).
      auto body = [&](const float a) {
        const float ae = min_f32(kAlphaMax, a);
        const float w = ae * T[q];
        const v2f cc = __builtin_elementwise_fma(G2d[q], BD, G01[q] * RG);
        const float gc = cc.x + cc.y;                   // g . c
        gS[q] -= w * gc;
        const float om = 1.f - ae;
        const float ro = __builtin_amdgcn_rcpf(om);
        const float dA = T[q] * gc - ro * gS[q];
        T[q] *= om;
        const float Wt = a * dA;
        const v2f Wxy = v2f{Wt, Wt} * d;                // (W dx, W dy)
        if constexpr (POSE) {
          R12 += Wxy;
          R34 = __builtin_elementwise_fma(v2f{Wxy.x, Wxy.x}, d, R34);
          r5 = __builtin_fmaf(Wxy.y, d.y, r5);
          r0 = __builtin_fmaf(w, G2d[q].y, r0);          // Rd (r0 is free in this variant)
        } else if constexpr (!JONLY) {
          r0 += Wt;
          R12 += Wxy;
          R34 = __builtin_elementwise_fma(v2f{Wxy.x, Wxy.x}, d, R34);
          r5 = __builtin_fmaf(Wxy.y, d.y, r5);
          const v2f ww = {w, w};
          R67 = __builtin_elementwise_fma(ww, G01[q], R67);
          R89 = __builtin_elementwise_fma(ww, G2d[q], R89);
        }
        if constexpr (SKETCH) {
          jq_mask |= 1u << q;      // this lane's row of the quadrant is non-zero
          // J_t += W (c0 dx + c1 dy + c2 dx^2 + c3 dx dy + c4 dy^2) + (w dL/dD) c5 for the six tau
          // components: the splat's 36 coefficients are staged in LDS in feature-major order, so two
          // tau components share one packed FMA (18 instead of 36 per quadrant)
          const float X[6] = {Wxy.x, Wxy.y, Wxy.x * d.x, Wxy.x * d.y, Wxy.y * d.y, w * G2d[q].y};
#pragma unroll
          for (int i = 0; i < 6; i++) {
            const v2f xi = {X[i], X[i]};
#pragma unroll
            for (int t = 0; t < 3; t++)
              J2[q][t] = __builtin_elementwise_fma(cf2[3 * i + t], xi, J2[q][t]);
          }
        }
      };
      // Quadrant 0 finds the pixel sums at zero and SETS them: written with a select, every lane computes
      // and a lane that does not blend the splat yields exact zeros (no zeroing per splat on this path).
      // The other quadrants run under the EXEC mask of the blending lanes - no select; the other lanes' T
      // and g.S simply stay and the sums receive nothing from them.
      if (q == 0) body(k ? ar : 0.f);
      else if (k) body(ar);
    }
    if (!JONLY && any) {
      // the ten wave totals land in ten different lanes; each stores its own dword of the
      // pair's record (slot of splat j broadcast from lane j): one store instruction per splat
      float mres, eres;
      if constexpr (POSE) {
        float r[6] = {R12.x, R12.y, R34.x, R34.y, r5, r0};
        wave_sum6_scatter(r, mres, eres);
      } else {
        float r[10] = {r0, R12.x, R12.y, R34.x, R34.y, r5, R67.x, R67.y, R89.x, R89.y};
        wave_sum10_scatter(r, b3mask, mres, eres);
      }
      const int sj = __builtin_amdgcn_readlane(slot, j);
      // wave-uniform; buffer store: the record's offset is a scalar (soffset), the lane adds its 32-bit byte
      // offset, and a lane without a dword has an offset beyond the buffer's extent, which drops its store
      // (no per-lane 64-bit address, no exec masking)
      if (MGS_ABL_PAIR && sj >= 0)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(wextra ? eres : mres), pair_rsrc, wofs4, sj * (kPairStride * 4), 0);
      written |= 1ull << j;
    }
  };

  // ---- the item's segments, 64 splats (one staged record per lane) at a time ------------------
  for (int sub = 0; sub < nb; sub += kSeg) {
    const int nsub = min(kSeg, nb - sub);
    sub_base = base + sub;
    const unsigned int lo = lo_next;
    // key of the NEXT segment: in flight while this one is staged and walked
    lo_next = (sub + kSeg + lane < nb) ? (unsigned int)P.keys[k0 + sub + kSeg + lane] : 0u;
    // does any pixel of the tile reach this segment?  (if not, none of the later ones either,
    // but their pairs still need their zero records)
    const bool dead = tile_last <= sub_base;
    // stage the segment's records (one per lane): slot, raw conic
    slot = -1;
    if (sub > 0) wave_lds_fence();    // the staged records belong to this wave alone: no workgroup barrier
    if (lane < nsub) {
      const unsigned int id = P.pack ? lo >> kPackBits : lo;
      if constexpr (!JONLY) {
        slot = pair_slot_base(P, (int)id) + (int)(P.pack ? (lo & ((1u << kPackBits) - 1u)) : P.payload[k0 + sub + lane]);
        // slots are Gaussian-major positions among ALL pairs: with an undersized capacity some
        // lie beyond the pair_grad buffer (the forward was incomplete anyway) - never touch them
        if (slot >= P.cap) slot = -1;
      }
      if (!dead) {
        const float4* src = reinterpret_cast<const float4*>(P.rec + MGS_ABL_REC(id));
        const float4 qa = src[0];     // 16 + 12 + 12 B: no dead components (see k_blend_fwd)
        const float3 qb = *reinterpret_cast<const float3*>(src + 1), q2 = *reinterpret_cast<const float3*>(src + 2);
        s_r0[lane] = make_float4(qa.x, qa.y, -0.5f * kLog2eB * qb.x, -kLog2eB * qb.y);
        s_r1[lane] = make_float4(-0.5f * kLog2eB * qb.z, qa.w, q2.x, q2.y);   // (C', opacity, r, g)
        s_r2[lane] = make_float2(q2.z, qa.z);                                  // (b, depth)
        if constexpr (SKETCH) {
          const float4* cj = reinterpret_cast<const float4*>(B.splat_jac + (size_t)id * 36);
#pragma unroll
          for (int i = 0; i < 9; i++) s_coef[lane][i] = cj[i];
        }
      }
    }
    written = 0ull;
    if (!dead) {
      // quadrant reach masks of the whole segment as four wave-uniform 64-bit words (bit j = splat
      // j reaches quadrant q): they live in SGPRs, so the walk below skips unreachable splats and
      // quadrants with scalar bit tests - no LDS round trip in front of every splat
      // (a splat behind the quadrant's last contribution cannot contribute - the forward had stopped visiting the
      // saturated quadrant, and did not write reach words any more: only the first qlast - base bits are kept)
      static_assert(kItem <= 32, "a reach word per (item, quadrant) covers 32 splats");
      const unsigned int rwq[4] = {rw.x, rw.y, rw.z, rw.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int keep = min(max(qlast[q] - sub_base, 0), nsub);
        mq[q] = (unsigned long long)(rwq[q] & (unsigned int)((1ull << keep) - 1ull));
      }
      wave_lds_fence();
      // Walk over the splats that reach any quadrant (set bits of the union mask), two-way
      // unrolled with the NEXT splat's record prefetched from LDS above the arithmetic of the
      // current one, so that no LDS latency sits between two splats and no registers rotate.
      unsigned long long todo = mq[0] | mq[1] | mq[2] | mq[3];
      MGS_BMARK(1);
      MGS_BCOUNT(__popcll(mq[0]) + __popcll(mq[1]) + __popcll(mq[2]) + __popcll(mq[3]), __popcll(todo));
      if (todo != 0ull) {
        int j0 = __builtin_ctzll(todo);
        float4 u0 = s_r0[j0], v0 = s_r1[j0];
        float2 w0 = s_r2[j0];
        while (true) {
          todo &= todo - 1ull;
          const int j1 = __builtin_ctzll(todo) & 63;       // todo == 0: harmless read of slot 63
          const float4 u1 = s_r0[j1], v1 = s_r1[j1];
          const float2 w1 = s_r2[j1];
          __builtin_amdgcn_sched_barrier(0);               // keep the prefetch above the arithmetic
          visit(j0, u0, v0, w0);
          if (todo == 0ull) break;
          todo &= todo - 1ull;
          j0 = __builtin_ctzll(todo) & 63;
          u0 = s_r0[j0]; v0 = s_r1[j0]; w0 = s_r2[j0];
          __builtin_amdgcn_sched_barrier(0);
          visit(j1, u1, v1, w1);
          if (todo == 0ull) break;
        }
      }
    }
    MGS_BMARK(2);
    // splats of the segment that no pixel reached: zero record
    if (!JONLY && slot >= 0 && !((written >> lane) & 1ull))
    {
      float2* dst = reinterpret_cast<float2*>(B.pair_grad + (size_t)slot * kPairStride);     // 40-B records: 8-B aligned
      const float2 z = make_float2(0.f, 0.f);
#pragma unroll
      for (int i = 0; i < kPairStride / 2; i++) dst[i] = z;
    }
  }
  }   // rep
  flush_jacobian();
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
//   k_blend_bwd<1>   per (pixel, splat): J_p += W * poly(dx, dy) + w dL/dD * c  (6 components) in registers; the rows of
//                    every run of consecutive items of one tile a wave walked leave as one SLAB (float[6][256],
//                    quadrant-major, plain stores) + a mask word of the quadrants written
//   k_sketch_bucket  per tile: adds up its slabs; per pixel: row -> LDS-privatised bucket table -> grad_sketch_dtau
__device__ __forceinline__ void sketch_prep_gaussian(const KP& P, const KB& B, int idx) {
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
  for (int t = 0; t < 6; t++) {      // feature-major: out[6 * feature + tau component]
    out[6 * 0 + t] = -(A * M[0][t] + Bc * M[1][t]);
    out[6 * 1 + t] = -(Bc * M[0][t] + Cc * M[1][t]);
    out[6 * 2 + t] = -0.5f * M[2][t];
    out[6 * 3 + t] = -M[3][t];
    out[6 * 4 + t] = -0.5f * M[4][t];
    out[6 * 5 + t] = M[5][t];
  }
}

__global__ __launch_bounds__(kPreBlock) void k_sketch_prep(KP P, KB B) {
  sketch_prep_gaussian(P, B, blockIdx.x * kPreBlock + threadIdx.x);
}

// The residual pass of the second-order tracking iteration (sketch_kernels.h: streaming + bucket atomics, latency-
// bound, 26 us alone at 640x480) and the per-splat preparation above (VALU-bound, 24 us alone at 300 k Gaussians) do not
// depend on each other: ONE launch, the first `res_blocks` workgroups take the pixels, the others the Gaussians, and
// the two populations share the CUs (round 4: 50 -> ~28 us for the pair).
static_assert(kPreBlock == kSketchThreads, "one block size for both roles");
__global__ __launch_bounds__(kPreBlock) void k_sketch_prep_residual(KP P, KB B, mgs_sketch_residual_args A, SketchKeys K,
                                                                    int res_blocks) {
  extern __shared__ float s_acc[];   // residual role: [d][3]
  __shared__ float s_red[kSketchThreads / 64];
  if ((int)blockIdx.x < res_blocks) sketch_residual_block(A, K, s_acc, s_red, blockIdx.x, res_blocks);
  else sketch_prep_gaussian(P, B, ((int)blockIdx.x - res_blocks) * kPreBlock + threadIdx.x);
}

constexpr int kBucketBlocks = 256;      // persistent workgroups (one per CU), four tiles in hand each (64 / 128 / 256 / 512
                                        // measured: 56.7 / 38.5 / 32.7 / 39.9 us)
constexpr int kBucketThreads = 1024;

// Per tile: per-pixel rows = sum of the tile's slabs (k_blend_bwd<SKETCH> left one per run of kSketchReps items:
// at the tile's first item and at every multiple of kSketchReps inside it), then the pixel's bucket(s) in an
// LDS-privatised table, flushed with one float atomic per entry and workgroup.  A 256-thread quarter of the
// workgroup holds a tile in the quadrant-major order of the slabs (thread = 64 q + lane): coalesced 1-KB loads.
__global__ __launch_bounds__(kBucketThreads) void k_sketch_bucket(KP P, KB B) {
  extern __shared__ float s_acc[];    // stack * sketch * 6
  const int nacc = B.stack_dim * B.sketch_dim * 6;
  const size_t HW = (size_t)P.W * P.H;
  for (int i = threadIdx.x; i < nacc; i += kBucketThreads) s_acc[i] = 0.f;
  __syncthreads();
  const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255, q = tid >> 6, lane = tid & 63;
  for (int tile = blockIdx.x * 4 + sub; tile < P.T; tile += gridDim.x * 4) {
    const int a = P.seg_offset[tile], b = min(P.seg_offset[tile + 1], P.max_segs);
    float J[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // The loads are unconditional and the quadrant bit selects the VALUE (a quadrant the run never reached holds
    // whatever the buffer held): no branch between the slabs, so several slabs' loads are in flight together
    // (with a branch per slab the ~10 slabs of a tile were ten dependent round trips).
#pragma unroll 4
    for (int i = a; i < b; i = (i / kSketchReps + 1) * kSketchReps) {       // wave-uniform
      const int sid = slab_index(i, tile);
      const bool on = (B.slab_mask[sid] >> q) & 1u;
      const float* slab = B.slabs + (size_t)sid * (6 * 256) + tid;
#pragma unroll
      for (int t = 0; t < 6; t++) {
        const float v = slab[t * 256];
        J[t] += on ? v : 0.f;
      }
    }
    const int px = (tile % P.grid_x) * kTile + (lane & 7) + 8 * (q & 1);
    const int py = (tile / P.grid_x) * kTile + (lane >> 3) + 8 * (q >> 1);
    if (px >= P.W || py >= P.H) continue;
    const size_t p = (size_t)py * P.W + px;
    if (B.sketch_flat) {      // one bucket per pixel
      const int bk = B.sketch_flat[p];
      if (bk >= 0 && bk < B.stack_dim * B.sketch_dim) {
        float* acc = &s_acc[bk * 6];
#pragma unroll
        for (int t = 0; t < 6; t++) atomicAdd(&acc[t], J[t]);
      }
    } else {
      for (int st = 0; st < B.stack_dim; st++) {
        const int k = B.sketch_idx[(size_t)st * HW + p];
        if (k >= 0 && k < B.sketch_dim) {
          float* acc = &s_acc[(st * B.sketch_dim + k) * 6];
#pragma unroll
          for (int t = 0; t < 6; t++) atomicAdd(&acc[t], J[t]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nacc; i += kBucketThreads) {
    const float v = s_acc[i];
    if (v != 0.f) atomicAdd(&B.g_sketch[i], v);
  }
}

// ---------------------------------------------------------------------------------
// MAP (mapping mode, mgs_map_accum_args): instead of storing the gradients w.r.t. the activated
// attributes of this view, chain them through GaussianModel's activations
// (gaussian_model.py:54-62: scaling = exp, opacity = sigmoid, rotation = normalize) and
// accumulate into the per-iteration gradient buffer of the raw parameters, add the isotropic
// regulariser's gradient (slam_backend.py:244-246) and this view's densification statistics
// (gaussian_model.py:693-697, slam_backend.py:292-299) and occ-aware visibility (:251-255).
// (no waves-per-SIMD bound: with an explicit 4 the same kernel ran 26 -> 35 us, with 5 or 6 it spills)
// SH0 (mapping mode): active SH degree 0 - the colour gradient is three scalars; the general form
// keeps a 48-entry coefficient array that lives in scratch memory.
template <bool MAP, bool SH0 = false>
__global__ __launch_bounds__(kPreBlock) void k_preprocess_bwd(KP P, KB B) {
  __shared__ float s_tau[kPreBlock / 64][6];
  __shared__ float2 s_stage[kPreBlock / 64][kPreChunk * (kPairStride / 2)];     // 5 KB per wave
  __shared__ float4 s_park_q[MAP ? kPreBlock : 1];     // mapping mode: raw rotation and opacity of this thread's
  __shared__ float s_park_o[MAP ? kPreBlock : 1];      // Gaussian, parked until the activation chain
  const int idx = blockIdx.x * kPreBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float tau[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // Every load that depends on idx alone is issued here, ahead of any use and outside the
  // "visible" branch: one memory round trip, then one more for the pair records (it was five in a
  // row: record -> slot offsets -> pair records -> scales / rotation -> model parameters).
  const bool valid = idx < P.N;
  const int idc = valid ? idx : 0;            // clamped: lanes behind the last Gaussian load entry 0 and drop it
  const float4* recp = reinterpret_cast<const float4*>(P.rec + idc);
  const float4 r0 = recp[0], r1 = recp[1], r2 = recp[2];
  const float p[3] = {P.means[3 * idc], P.means[3 * idc + 1], P.means[3 * idc + 2]};
  const int pair_cnt = valid ? P.pair_count[idc] : 0, pair_o = P.pair_off[idc];
  // per_block is a multiple of kPreBlock: the binning block of this whole workgroup (scalar load)
  const int pair_b = P.block_prefix[(blockIdx.x * kPreBlock) / P.per_block];
  float sc[3] = {1.f, 1.f, 1.f}, c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float4 qq = make_float4(1.f, 0.f, 0.f, 0.f);
  if (P.covp) {
#pragma unroll
    for (int i = 0; i < 6; i++) c6[i] = P.covp[6 * (size_t)idc + i];
  } else {
    sc[0] = P.scales[3 * idc]; sc[1] = P.scales[3 * idc + 1]; sc[2] = P.scales[3 * idc + 2];
    qq = reinterpret_cast<const float4*>(P.rots)[idc];
  }
  if constexpr (MAP) {
    // needed only by the activation chain at the very end: requested with the other loads (same round
    // trip) but parked in LDS, not in five registers across the whole kernel (97 -> <= 96 VGPRs keeps
    // five waves per SIMD; at four this variant ran 29.4 -> 34.8 us)
    s_park_q[threadIdx.x] = reinterpret_cast<const float4*>(B.map.raw_rot)[idc];
    s_park_o[threadIdx.x] = P.opac[idc];
  }
  // ---- the pair records of this wave's 64 Gaussians: ONE contiguous run of 40-B records (slots are
  // Gaussian-major in index order), streamed into LDS with coalesced 8-B loads, kPreChunk records per
  // trip; every lane then adds up its own Gaussian's records in slot order.  (One thread chasing its
  // own records read 16 B per lane at a 40..48-B stride: 3.4 TB/s.)
  float acc[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  {
    const bool has = valid && __float_as_int(r1.w) > 0 && pair_cnt > 0;
    // clamped to the capacity of the pair_grad buffer (see k_blend_bwd)
    const int s0 = has ? min(pair_b + pair_o, P.cap) : 0x7fffffff;
    const int s1 = has ? min(min(pair_b + pair_o, P.cap) + pair_cnt, P.cap) : 0;
    int lo = s0, hi = s1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
    const int S0 = __builtin_amdgcn_readfirstlane(lo), S1 = __builtin_amdgcn_readfirstlane(hi);
    float2* stage = s_stage[wave];
    const float2* src = reinterpret_cast<const float2*>(B.pair_grad);
    for (int c = S0; c < S1; c += kPreChunk) {                   // wave-uniform
      const int n = min(kPreChunk, S1 - c) * (kPairStride / 2);  // 8-B words of this trip
      const size_t w0 = (size_t)c * (kPairStride / 2);
      // every load of the trip is in flight before the first one is stored (a rolled loop waited for each
      // pair of loads in turn: five memory round trips per trip instead of one)
      constexpr int kWords = kPreChunk * (kPairStride / 2) / 64;   // 8-B words per lane and trip
      float2 w[kWords];
#pragma unroll
      for (int u = 0; u < kWords; u++) {
        const int i = lane + 64 * u;
        w[u] = src[w0 + min(i, n - 1)];        // clamped: the last trip is ragged, the load stays unconditional
      }
#pragma unroll
      for (int u = 0; u < kWords; u++) {
        const int i = lane + 64 * u;
        if (i < n) stage[i] = w[u];
      }
      wave_lds_fence();
      const int a0 = max(s0, c), a1 = min(s1, c + kPreChunk);
      for (int sl = a0; sl < a1; sl++) {
        const float2* rp = stage + (sl - c) * (kPairStride / 2);
        const float2 v0 = rp[0], v1 = rp[1], v2 = rp[2], v3 = rp[3], v4 = rp[4];
        acc[0] += v0.x; acc[1] += v0.y; acc[2] += v1.x; acc[3] += v1.y; acc[4] += v2.x;
        acc[5] += v2.y; acc[6] += v3.x; acc[7] += v3.y; acc[8] += v4.x; acc[9] += v4.y;
      }
      wave_lds_fence();                                          // before the next trip overwrites the stage
    }
  }
  if (valid) {
    __builtin_amdgcn_sched_barrier(0);
    const int radius = __float_as_int(r1.w);
    const unsigned int flags = __float_as_uint(r2.w);
    float dmean[3] = {0.f, 0.f, 0.f}, dndc[2] = {0.f, 0.f}, dop = 0.f;
    float dscale[3] = {0.f, 0.f, 0.f}, drot[4] = {0.f, 0.f, 0.f, 0.f};
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float grgb[3] = {0.f, 0.f, 0.f};
    if (radius > 0) {
      float a[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // the wave staged its whole run of records in LDS above; this Gaussian's share, in slot order
      // (the summation order is fixed: bit-reproducible, also across ranks)
#pragma unroll
      for (int k = 0; k < 10; k++) a[k] = acc[k];
      if (!MAP && !B.g_means3D) { a[0] = 0.f; a[6] = 0.f; a[7] = 0.f; a[8] = 0.f; }   // pose-only: not produced
      // a[] = raw pixel sums (S1, Sx, Sy, Sxx, Sxy, Syy, Rr, Rg, Rb, Rd) over all tiles of
      // the Gaussian; the conic / opacity are per-Gaussian, so the linear map to screen-space
      // gradients is applied once here instead of once per pair
      const float cA = r1.x, cB = r1.y, cC = r1.z;
      const float g_xy[2] = {-(cA * a[1] + cB * a[2]), -(cC * a[2] + cB * a[1])};
      const float g_con[3] = {-0.5f * a[3], -a[4], -0.5f * a[5]};
      const float g_op = a[0] != 0.f ? a[0] / r0.w : 0.f;   // opacity 0 never reaches a pixel
      Camera cam;
      load_camera_b(cam, P);
      GaussGrad gg;
      if (P.covp) {
        project_gaussian_backward(cam, p, nullptr, nullptr, c6, g_xy, g_con, g_op, a[9], gg);
      } else {
        const float q[4] = {qq.x, qq.y, qq.z, qq.w};
        project_gaussian_backward(cam, p, sc, q, nullptr, g_xy, g_con, g_op, a[9], gg);
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
    if constexpr (MAP) {
      const KM& M = B.map;
      const bool add = M.accumulate != 0;
      auto put = [&](float* dst, float v) { *dst = add ? *dst + v : v; };
      // Accumulating a view in which this Gaussian is not visible changes nothing: all its gradients and both
      // statistics are zero - no read-modify-write of its 14 + 2 floats.  (Gaussians of old keyframes sit in
      // contiguous index ranges: in a real window whole waves take this path.)  The regulariser, when this launch
      // adds it, goes to every Gaussian.
      const bool touch = !(add && !M.add_reg && radius <= 0);
      if (touch) {
      // colour coefficients first: for degree > 0 sh_backward adds the view-direction term to dmean
      if constexpr (SH0) {
        float d3[3] = {0.f, 0.f, 0.f};
        if (radius > 0 && P.shs) {
#pragma unroll
          for (int c = 0; c < 3; c++) d3[c] = (flags & (1u << c)) ? 0.f : SH_C0 * grgb[c];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) put(&M.g_fdc[3 * (size_t)idx + c], d3[c]);
        if (M.g_frest && !add)      // stored degree above the active one: the higher bands get no gradient
          for (int k = 3; k < 3 * P.K; k++) M.g_frest[(size_t)3 * (P.K - 1) * idx + (k - 3)] = 0.f;
      } else {
        // the coefficient gradients go to memory as they are produced (no 48-entry array: it lived in scratch
        // memory, 208 B per lane, and held this instantiation at 4 waves per SIMD)
        float* frest = M.g_frest ? M.g_frest + (size_t)3 * (P.K - 1) * idx : nullptr;
        auto sink = [&](int k, float v) {
          if (k < 3) put(&M.g_fdc[3 * (size_t)idx + k], v);
          else if (frest) put(&frest[k - 3], v);
        };
        int active = 0;       // coefficient floats that received a gradient
        if (radius > 0 && P.shs) {
          if (P.deg == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) sink(c, (flags & (1u << c)) ? 0.f : SH_C0 * grgb[c]);
            active = 3;
          } else {
            sh_backward_emit(P.deg, P.shs + (size_t)3 * P.K * idx, p, P.campos, flags, grgb, dmean, sink);
            active = 3 * (P.deg + 1) * (P.deg + 1);
          }
        }
        if (!add)             // overwrite mode: everything that got no gradient is zero (accumulating: unchanged)
          for (int k = active; k < 3 * P.K; k++) sink(k, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 3; i++) put(&M.g_xyz[3 * (size_t)idx + i], dmean[i]);
      // opacity = sigmoid(logit)
      const float o = s_park_o[threadIdx.x];
      put(&M.g_opacity[idx], dop * o * (1.f - o));
      // scaling = exp(log scale) (+ regulariser: weight * mean_{N x 3} |s_k - mean_k s|)
      {
        const float s0 = sc[0], s1 = sc[1], s2 = sc[2];
        float g0 = dscale[0] * s0, g1 = dscale[1] * s1, g2 = dscale[2] * s2;
        if (M.scale_dims == 3) {
          if (M.add_reg) {
            const float mean = (s0 + s1 + s2) * (1.f / 3.f);
            auto sg = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
            const float e0 = sg(s0 - mean), e1 = sg(s1 - mean), e2 = sg(s2 - mean);
            const float em = (e0 + e1 + e2) * (1.f / 3.f);
            g0 += M.reg_scale * (e0 - em) * s0;
            g1 += M.reg_scale * (e1 - em) * s1;
            g2 += M.reg_scale * (e2 - em) * s2;
          }
          put(&M.g_scaling[3 * (size_t)idx], g0);
          put(&M.g_scaling[3 * (size_t)idx + 1], g1);
          put(&M.g_scaling[3 * (size_t)idx + 2], g2);
        } else {   // isotropic [N,1] broadcast: the three axes share one parameter (regulariser = 0)
          put(&M.g_scaling[idx], g0 + g1 + g2);
        }
      }
      // rotation = q / |q|:  dL/dq = (g - qn (qn . g)) / |q|
      {
        const float4 qr = s_park_q[threadIdx.x];
        const float n2 = qr.x * qr.x + qr.y * qr.y + qr.z * qr.z + qr.w * qr.w;
        const float inv = 1.f / fmaxf(sqrtf(n2), 1e-12f);
        const float qn[4] = {qr.x * inv, qr.y * inv, qr.z * inv, qr.w * inv};
        const float dot = qn[0] * drot[0] + qn[1] * drot[1] + qn[2] * drot[2] + qn[3] * drot[3];
        float4* dst = reinterpret_cast<float4*>(M.g_rotation) + idx;
        float4 gq = make_float4((drot[0] - qn[0] * dot) * inv, (drot[1] - qn[1] * dot) * inv,
                                (drot[2] - qn[2] * dot) * inv, (drot[3] - qn[3] * dot) * inv);
        if (add) { const float4 old = *dst; gq.x += old.x; gq.y += old.y; gq.z += old.z; gq.w += old.w; }
        *dst = gq;
      }
      }   // touch
      // statistics of this view
      const bool vis = radius > 0;
      if (vis || !add) {
        if (M.gradnorm_inc) put(&M.gradnorm_inc[idx], vis ? sqrtf(dndc[0] * dndc[0] + dndc[1] * dndc[1]) : 0.f);
        if (M.denom_inc) put(&M.denom_inc[idx], vis ? 1.f : 0.f);
        if (M.radii_max) M.radii_max[idx] = add ? max(M.radii_max[idx], radius) : radius;
      }
      if (M.visibility) M.visibility[idx] = P.n_touched[idx] > 0 ? 1 : 0;
    } else if (B.g_means3D) {   // NULL: pose-only backward (tracking), nothing per Gaussian is stored
    // colours
    if (P.shs) {
      float* dsh = B.g_colors + (size_t)3 * P.K * idx;
      if (radius > 0) {
        if (SH0 || P.deg == 0) {
#pragma unroll
          for (int c = 0; c < 3; c++) dsh[c] = (flags & (1u << c)) ? 0.f : SH_C0 * grgb[c];
          for (int k = 3; k < 3 * P.K; k++) dsh[k] = 0.f;
        } else {
          if constexpr (!SH0) sh_backward(P.deg, P.K, P.shs + (size_t)3 * P.K * idx, p, P.campos, flags, grgb, dsh, dmean);
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

// occ-aware visibility of a forward-only (prune) pass: slam_backend.py:251-255
__global__ void k_visibility_only(const int* n_touched, unsigned char* vis, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) vis[i] = n_touched[i] > 0 ? 1 : 0;
}

// Fixed-order sum of the block partials -> grad_tau[6].  (Folding this into k_preprocess_bwd
// with a last-workgroup ticket was measured slower: the hand-off makes every workgroup drain
// its 20 MB of gradient stores before it may retire.)
__global__ __launch_bounds__(768) void k_tau_reduce(KB B, int nblk) {
  // 6 components x 2 halves of the partials x 64 lanes; every lane's loads are independent (fully
  // unrolled in flight), then a wave sum, then the two halves are added in a fixed order.
  __shared__ float s_half[12];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int comp = wave % 6, half = wave / 6;
  const int mid = (nblk + 1) / 2;
  const int i0 = half ? mid : 0, i1 = half ? nblk : mid;
  float s = 0.f;
#pragma unroll 16
  for (int i = i0 + lane; i < i1; i += 64) s += B.tau_partial[i * 6 + comp];
  s = wave_sum_to_lane63(s);
  if (lane == 63) s_half[wave] = s;
  __syncthreads();
  if (threadIdx.x < 6) B.g_tau[threadIdx.x] = s_half[threadIdx.x] + s_half[6 + threadIdx.x];
}

int launch_visibility(const int* n_touched, unsigned char* vis, int n, hipStream_t st) {
  launch("visibility", k_visibility_only, dim3((n + 255) / 256), dim3(256), st, n_touched, vis, n);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------
int launch_backward(const KP& P, const KB& B, hipStream_t st, bool skip_tau_reduce, const SketchFuse* fuse) {
  // k_blend_bwd stores the pair records through a raw buffer descriptor of kPairBufferExtent bytes (53 M pairs)
  if ((unsigned long long)max(P.cap, 0) * (kPairStride * 4) > kPairBufferExtent) return MGS_ERR_UNSUPPORTED;
  // work items to launch for: a tile holds at most one partly filled item more than its pairs / kItem
  const int items = B.pair_bound > 0 ? (int)min((long long)P.max_segs, (long long)B.pair_bound / kItem + P.T) : P.max_segs;
  if (B.sketch_mode != 0) {
    const size_t HW = (size_t)P.W * P.H;
    const size_t nacc = (size_t)B.stack_dim * B.sketch_dim * 6;
    if (nacc * sizeof(float) > 64 * 1024) return MGS_ERR_UNSUPPORTED;
    if (!B.scratch_kept_zero && !hip_ok("memset(sketched Jacobian)", hipMemsetAsync(B.g_sketch, 0, nacc * sizeof(float), st))) {
      launches_ok();      // (reported above; the per-thread slot is cleared for the next entry point)
      return MGS_ERR_LAUNCH;
    }
    const int prep_blocks = (P.N + kPreBlock - 1) / kPreBlock;
    if (fuse && fuse->residual) {      // the residual pass rides beside the preparation (or alone: later repeats)
      const mgs_sketch_residual_args& A = *fuse->residual;
      const size_t smem = sizeof(float) * 3 * (size_t)A.stack_dim * A.sketch_dim;
      if (smem > 48 * 1024) return MGS_ERR_UNSUPPORTED;
      const long long want = (A.num_pixels + kSketchThreads - 1) / kSketchThreads;
      const int res_blocks = (int)(want < kSketchBlocks ? want : kSketchBlocks);
      launch_smem("sketch_prep_residual", k_sketch_prep_residual, dim3(res_blocks + (fuse->skip_prep ? 0 : prep_blocks)),
                  dim3(kPreBlock), smem, st, P, B, A, fuse->keys, res_blocks);
    } else if (!(fuse && fuse->skip_prep)) {
      launch("sketch_prep", k_sketch_prep, dim3(prep_blocks), dim3(kPreBlock), st, P, B);
    }
    if (B.sketch_only)
      launch("blend_bwd_sketch", k_blend_bwd<true, true>, dim3(grid_pad((items + kSketchReps - 1) / kSketchReps, kBwdChunk)), dim3(64), st, P, B);
    else
      launch("blend_bwd_sketch", k_blend_bwd<true, false>, dim3(grid_pad((items + kSketchReps - 1) / kSketchReps, kBwdChunk)), dim3(64), st, P, B);
    launch_smem("sketch_bucket", k_sketch_bucket, dim3(kBucketBlocks), dim3(kBucketThreads), nacc * sizeof(float), st, P, B);
    if (B.sketch_only) return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
  } else {
    if (B.g_means3D || B.map.on)
      launch("blend_bwd", k_blend_bwd<false>, dim3(grid_pad(items, kBwdChunk)), dim3(64), st, P, B);
    else   // pose-only (tracking)
      launch("blend_bwd", k_blend_bwd<false, false, true>, dim3(grid_pad(items, kBwdChunk)), dim3(64), st, P, B);
  }
  const int npre = (P.N + kPreBlock - 1) / kPreBlock;
  if (B.map.on && P.deg == 0) launch("preprocess_bwd_map", k_preprocess_bwd<true, true>, dim3(npre), dim3(kPreBlock), st, P, B);
  else if (B.map.on) launch("preprocess_bwd_map", k_preprocess_bwd<true, false>, dim3(npre), dim3(kPreBlock), st, P, B);
  else if (!P.shs || P.deg == 0) launch("preprocess_bwd", k_preprocess_bwd<false, true>, dim3(npre), dim3(kPreBlock), st, P, B);
  else launch("preprocess_bwd", k_preprocess_bwd<false, false>, dim3(npre), dim3(kPreBlock), st, P, B);
  if (!skip_tau_reduce) launch("tau_reduce", k_tau_reduce, dim3(1), dim3(768), st, B, npre);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

}  // namespace mgs
