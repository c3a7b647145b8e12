// Kernel-side parameter blocks and workspace layout shared by the .hip translation
// units.  Host-only callers use include/monogs_raster.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/monogs_raster.h"
#include "raster_math.h"

// Traffic ablations (variant builds of profiles/tools/build_variant.sh only; the product is built without them):
//   -DMGS_ABL_REC0    every splat-record gather of the blend kernels reads record 0  -> what the gathers fetch
//   -DMGS_ABL_NOCKPT  the forward writes no blend checkpoints, the backward reads none -> what the checkpoints move
//   -DMGS_ABL_NOPAIR  the backward stores no pair records                              -> what the pair records write
// (results are wrong by construction; only durations and FETCH_SIZE / WRITE_SIZE of such a build mean anything)
#ifdef MGS_ABL_REC0
#define MGS_ABL_REC(id) 0u
#else
#define MGS_ABL_REC(id) (id)
#endif
#ifdef MGS_ABL_NOCKPT
#define MGS_ABL_CKPT false
#else
#define MGS_ABL_CKPT true
#endif
#ifdef MGS_ABL_NOPAIR
#define MGS_ABL_PAIR false
#else
#define MGS_ABL_PAIR true
#endif

namespace mgs {

// One struct carries everything a forward or backward kernel may need; it is passed
// by value (kernarg segment -> SGPRs).
// Tracking objective evaluated in the forward blend's epilogue (native first-order tracking iteration only:
// mgs_tracking_iteration).  The quadrant wave that has just finished a pixel holds its colour and opacity in
// registers; it forms the residual of tracking.hip's k_track_loss_onepass there, writes d(loss)/d(image) for the
// backward and leaves its four sums as ONE partial per quadrant wave - the loss launch between the forward and
// the backward is gone.
struct KObj {
  int on;
  int p1;                                             // 1: the norm is p = 1 (else p = 2; any other p takes the loss kernel)
  float exposure_eps, huber_delta;
  const float *gt, *mask, *exposure_a, *exposure_b;   // gt [3][H*W]; mask [H*W] or null
  float* grad_image;                                  // [3][H*W]
  float* partial;                                     // [4][4T]: sum |h|^p | d/da | d/db | sum |r|, one entry per quadrant wave
};

struct KP {
  int N, W, H, grid_x, grid_y, T, deg, K, cap;
  int pack;                // 1: sort key low word = id << kPackBits | pair index (no payload array)
  float tanfovx, tanfovy, focal_x, focal_y, mod;
  const float *means, *scales, *rots, *covp, *opac, *shs, *precol;
  const float *V, *PM, *Praw, *campos, *bg;
  // geom workspace
  SplatRec* rec;
  int* pair_count;
  int* pair_off;           // N: slot offset of the Gaussian's first pair inside its binning block
  int* block_prefix;       // kBinBlocks+1: exclusive scan of the blocks' pair totals
  int* scan_tmp;           // scratch of the fallback N-scan
  unsigned int* hit_mask;  // N: which tiles of a small rectangle (<= kBinSmallRect, row-major bit i) the count pass
                           // found reachable - the emit pass replays the bits instead of the exact tests
  int* tile_count;
  int* tile_offset;
  int* tile_cursor;
  int* bin_table;          // kBinBlocks x T per-block tile histograms / bases
  // per-pixel results of the forward that only the backward reads, in QUADRANT-MAJOR order:
  // entry (tile * 4 + quadrant) * 64 + lane, lane = (y & 7) * 8 + (x & 7) - the order in which a
  // quadrant wave of either blend kernel holds its pixels, so every access is one coalesced 256-B run
  float4* final_TC;        // [256 T] (T, C0, C1, C2) at the end of the list (colour without background)
  int2* final_DL;          // [256 T] (depth as float bits, n_contrib = position behind the last contribution)
  int* quad_last;          // [4 T] max of n_contrib over the quadrant: the list position behind which the
                           // quadrant is saturated (the backward skips it from there on)
  int* seg_offset;         // T+1: exclusive scan of ceil(n_t / kItem): first backward item of a tile
  int* counters;
  // bins workspace
  unsigned long long* keys;
  unsigned int* payload;
  int4* seg_rec;           // backward item -> (tile, first key index, splats in the item, position in the list)
  float* ckpt;             // per item: float4[256] (T, C0, C1, C2) then float[256] D: blend state before its first splat
  uint4* reach;            // per item: for each quadrant of its tile, bit j = splat j of the item can reach the quadrant
                           // (the forward's quadrant waves form these ballots anyway; the backward reads them with the
                           // item record instead of repeating four box tests per staged splat: round 5)
  int max_segs;
  // forward outputs
  float *out_color, *out_depth, *out_opacity;
  int *radii, *n_touched;
  int* d_out;              // optional extra destination of D (may be pinned host memory)
  int* d_max;              // optional sticky high-water mark of D (atomicMax)
  int per_block;           // Gaussians per binning workgroup (a multiple of kPreBlock; 0x7fffffff on the
                           // global-atomics fallback): block_prefix[idx / per_block] + pair_off[idx] = first slot
  int big_pass;            // 1: tiles of more than 1024 pairs are left to the second sort launch
  int clamp_up;            // backward: mgs_backward_args.clamp_gradient_mode
  KObj obj;                // forward blend: objective in the epilogue (off unless the tracking iteration sets it)
};

struct KM {   // mapping mode of the preprocess backward (mgs_map_accum_args)
  int on, scale_dims, accumulate, add_reg;
  float reg_scale;          // weight / (3 N)
  const float* raw_rot;
  float *g_xyz, *g_fdc, *g_frest, *g_opacity, *g_scaling, *g_rotation;
  float *gradnorm_inc, *denom_inc;
  int* radii_max;
  unsigned char* visibility;
};

struct KB {   // backward extras
  const float *grad_color, *grad_depth;
  float* pair_grad;        // cap x kPairStride floats: the ten raw pixel sums of a pair at its slot
  float* tau_partial;      // nblocks x 6
  float *g_means3D, *g_means2D, *g_colors, *g_opac, *g_scales, *g_rots, *g_cov, *g_tau;
  int pair_bound;           // upper bound of the forward's pair count if the caller knows one (sizes the blend grid), else 0
  int sketch_mode, sketch_dim, stack_dim;
  int sketch_only;          // sketch mode: produce grad_sketch_dtau only (no per-splat sums, no grad_tau)
  int scratch_kept_zero;    // sketch mode: g_sketch is zero on entry and its consumer restores the zeros (native
                            // second-order iteration: no hipMemsetAsync launches per iteration)
  const int* sketch_idx;
  const int* sketch_flat;   // [H*W] stack * sketch_dim + bucket or -1 (compact alternative)
  float* g_sketch;
  // Sketch mode: the per-pixel pose-Jacobian rows of a tile are the SUM over its backward items.  Every wave of
  // k_blend_bwd<SKETCH> leaves the rows of each run of consecutive items of one tile it walked as ONE slab -
  // float[6][256] in quadrant-major pixel order, plain coalesced stores - at the index of the run's first item, and
  // k_sketch_bucket adds up a tile's slabs (until round 4 the rows met in a per-pixel array through float atomics:
  // 75 MB of 32-B-segment atomics per launch bounded the kernel).  slab_mask[i] = quadrants written (every run
  // start is written by every launch, so nothing is ever stale).
  float* slabs;            // (max_segs / kSketchReps + T + 1) x 6 x 256, indexed by slab_index()
  unsigned int* slab_mask; // one word per slab
  float* splat_jac;        // N x 36 per-splat d(xy,conic,depth)/dtau (sketch mode)
  KM map;
};

// XCD-aware work-item order (cdna_hip_programming.md T1): workgroups are dealt round-robin
// over the 8 XCDs, so with the identity mapping neighbouring tiles / the segments of one
// tile - which share splat records, per-pixel gradients and checkpoints - land in 8
// different L2s.  Here every XCD instead receives CHUNK consecutive logical items at a
// time, chunks dealt round-robin (a contiguous eighth of the grid per XCD would unbalance
// the XCDs: work per tile varies smoothly over the image and the padded tail is empty).
// Bijective on a grid padded to a multiple of 8*CHUNK.  Speed only: results do not depend
// on placement.
// exp2(min(x, 0)): the quadratic form of a splat is <= 0 and only rounding makes it positive; clamping the
// RESULT to [0, 1] is the same function and folds into v_exp_f32's clamp output modifier.
__device__ __forceinline__ float exp2_sat(float x) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x), 0.f, 1.f); }

// Launch-order index -> row-major tile id when the tiles are walked in 4x4 BLOCKS (bands of four tile rows, inside a
// band column blocks of four tiles, inside a block row-major; the last band / column block may be narrower).  A
// bijection of [0, gx * gy).  Used by the forward-blend placement experiment (-DMGS_FWD_BLOCKS).
__host__ __device__ inline int tile_from_block_order(int i, int gx, int gy) {
  const int band = i / (4 * gx), r = i - band * 4 * gx;
  const int h = min(4, gy - 4 * band);                 // rows of this band (1..4)
  const int c = r / (4 * h), r2 = r - c * 4 * h;       // column block, position inside it
  const int wc = min(4, gx - 4 * c);                   // its width (1..4)
  const int ty = 4 * band + r2 / wc, tx = 4 * c + r2 % wc;
  return ty * gx + tx;
}

template <int CHUNK>
__device__ __forceinline__ int xcd_remap(int bid) {
  const int xcd = bid & 7, slot = bid >> 3;
  return ((slot / CHUNK) * 8 + xcd) * CHUNK + (slot % CHUNK);
}
inline int grid_pad(int n, int chunk) { const int q = 8 * chunk; return (n + q - 1) / q * q; }
constexpr int kBwdChunk = 16;        // (8 / 32 / 64 re-measured with the 32-splat items: within 1 %)
constexpr int kSketchReps = 2;     // consecutive items per workgroup of the sketch-mode blend backward (r3: 1 / 2 / 4: 213 / 189 / 218 us;
                                   // r4, with slabs: 2 / 3 / 4: 178 / 192 / 214 us)

// Slab of the run of backward items that starts at item `item` of tile `tile` (sketch mode): runs start at a tile's
// first item and at every multiple of kSketchReps, so item / kSketchReps + tile is strictly increasing along the run
// starts - unique - and stays below max_segs / kSketchReps + T + 1: half the slabs one per item would take.
__host__ __device__ inline int slab_index(int item, int tile) { return item / kSketchReps + tile; }

constexpr uint64_t kAlign = 256;
inline uint64_t align_up(uint64_t v) { return (v + kAlign - 1) / kAlign * kAlign; }

struct Layout {
  uint64_t rec, pair_count, pair_off, hit_mask, block_prefix, scan_tmp, tile_count, tile_offset, tile_cursor, bin_table, final_TC, final_DL,
      quad_last, seg_offset, obj_partial, counters, geom_bytes;
  uint64_t keys, payload, seg_rec, reach, ckpt, max_segs, bins_bytes;
  uint64_t pair_grad, tau_partial, bwd_bytes;
  uint64_t slab_mask, slabs, splat_jac, sketch_bytes;
};

constexpr int kScanBlock = 2048;   // elements per block in the pair_base scan
constexpr int kPreBlock = 256;
// Packed sort keys: when N <= 2^20 and T <= 2^12 the pair's index within its Gaussian fits
// beside the Gaussian id in the low key word, so the per-tile sort moves 8 B per pair instead of
// 12 B.  The order is unchanged: a Gaussian occurs once per tile, so (depth, id) is already unique.
constexpr int kPackBits = 12;
constexpr int kPackMaxN = 1 << 20;
constexpr int kSeg = 64;            // splats staged at a time by the blend kernels (one record per lane)
// A backward work item is kItem consecutive splats of one tile's list (the per-pixel state is loaded
// once per item and carried across its segments in registers; the forward checkpoints the blend
// state in front of every item).  Measured on SYN-C (profiles/r02_item_size_experiment.txt):
// 1 segment per item 136 us, 2: 160 us, 4: 192 us - larger items cut the per-item loads but leave
// fewer items (3.7k at 4) than the chip has wave slots (~5.6k), and the kernel then runs at the
// latency of one long item instead of at the chip's throughput.  So: 1.
// HALF a segment per item is faster still (126 -> ~112 us): the kernel's duration is set by its last
// round of items, which run two or three to a SIMD at the latency of a lone wave; shorter items make
// that round shorter, at the price of a second per-pixel state load and checkpoint per segment.
constexpr int kItem = 32;
// Record of one (tile, Gaussian) pair in pair_grad: (S1, Sx, Sy, Sxx, Sxy, Syy, Rr, Rg, Rb, Rd), 40 B,
// no padding: k_preprocess_bwd streams a wave's contiguous run of records with coalesced loads.
constexpr int kPairStride = 10;
constexpr unsigned int kPairBufferExtent = 0x80000000u;   // k_blend_bwd addresses pair_grad as a raw buffer of this many bytes
constexpr int kPreChunk = 128;      // records staged per wave and trip in k_preprocess_bwd (5 KB of LDS per wave)
constexpr int kBinBlocks = 512;    // workgroups of the LDS-privatised binning passes
constexpr int kBinSmallMap = 65536;  // up to here the binning passes run 256-thread workgroups of 256 Gaussians
constexpr int kBinMaxTilesLds = 12288;   // T above this falls back to global atomics

// Grid of the LDS-privatised binning passes (defined in raster_forward.hip): number of workgroups
// and Gaussians per workgroup for N Gaussians.
void bin_grid(int N, int& nblk, int& per);

inline Layout make_layout(const mgs_raster_shape& s) {
  Layout L;
  const uint64_t N = (uint64_t)s.num_gaussians;
  const uint64_t gx = (s.width + kTile - 1) / kTile, gy = (s.height + kTile - 1) / kTile;
  const uint64_t T = gx * gy, HW = (uint64_t)s.width * s.height;
  const uint64_t cap = (uint64_t)(s.pair_capacity > 0 ? s.pair_capacity : 0);
  uint64_t o = 0;
  L.rec = o; o = align_up(o + N * sizeof(SplatRec));
  L.pair_count = o; o = align_up(o + N * 4);
  L.pair_off = o; o = align_up(o + N * 4);
  L.hit_mask = o; o = align_up(o + N * 4);
  L.block_prefix = o; o = align_up(o + (uint64_t)(kBinBlocks + 1) * 4);
  {
    const uint64_t nscan = (N + kScanBlock - 1) / kScanBlock + 1;
    L.scan_tmp = o; o = align_up(o + (nscan > (uint64_t)kBinBlocks + 1 ? nscan : (uint64_t)kBinBlocks + 1) * 4);
  }
  L.tile_count = o; o = align_up(o + T * 4);
  L.tile_offset = o; o = align_up(o + (T + 1) * 4);
  L.tile_cursor = o; o = align_up(o + T * 4);
  L.bin_table = o; o = align_up(o + (uint64_t)kBinBlocks * T * 4);
  L.final_TC = o; o = align_up(o + T * 256 * 16);
  L.final_DL = o; o = align_up(o + T * 256 * 8);
  L.quad_last = o; o = align_up(o + T * 4 * 4);
  L.seg_offset = o; o = align_up(o + (T + 1) * 4);
  L.obj_partial = o; o = align_up(o + T * 16 * 4);      // KObj::partial: 4 sums x 4T quadrant waves
  L.counters = o; o = align_up(o + 16);
  L.geom_bytes = o;
  o = 0;
  L.keys = o; o = align_up(o + cap * 8);
  L.payload = o; o = align_up(o + cap * 4);
  L.max_segs = cap / kItem + T;
  L.seg_rec = o; o = align_up(o + L.max_segs * 16);
  L.reach = o; o = align_up(o + L.max_segs * 16);
  L.ckpt = o; o = align_up(o + L.max_segs * 5 * 256 * 4);
  L.bins_bytes = o;
  o = 0;
  L.pair_grad = o; o = align_up(o + cap * kPairStride * 4);
  const uint64_t npre = (N + kPreBlock - 1) / kPreBlock;
  L.tau_partial = o; o = align_up(o + npre * 6 * 4);
  L.bwd_bytes = o;
  o = 0;
  const uint64_t nslabs = L.max_segs / kSketchReps + T + 1;      // slab_index() of the last possible run start + 1
  L.slab_mask = o; o = align_up(o + nslabs * 4);
  L.splat_jac = o; o = align_up(o + N * 36 * 4);
  L.slabs = o; o = align_up(o + nslabs * 6 * 256 * 4);
  L.sketch_bytes = o;
  return L;
}

}  // namespace mgs
