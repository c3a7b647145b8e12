// Forward pass of the rasteriser for gfx950 (CDNA4, wave64).
//
// Pipeline (one launch each, all on the caller's stream, no host sync):
//   k_bin_lds<count> 1024-thread workgroup per contiguous chunk of Gaussians: projection + EWA -> 48-B
//                    SplatRec, exact tile culling, per-chunk tile histogram in LDS (no global atomics),
//                    per-Gaussian slot offsets
//   k_bin_colsum     column prefix of the chunk histograms -> per-chunk bases and tile totals; the
//                    workgroup that finishes last also scans the tile totals (-> tile_offset, D)
//   k_bin_lds<emit>  same walk: (depth | id | pair index) keys into the tiles' ranges
//   k_tile_sort_reg  1 workgroup / tile  : register-resident bitonic sort of the tile's range by (depth, id)
//                    (DPP / ds_bpermute exchanges; k_tile_sort: LDS / HBM network for tiles beyond 1024 pairs);
//                    two size classes (<= 1024, <= 4096 keys), larger tiles sort in place in HBM
//   k_blend_fwd      1 wave / 8x8 quadrant (its own 64-thread workgroup): front-to-back blend over
//                    the tile's sorted list, 64 splats staged in LDS at a time, per-pixel state
//                    checkpointed every kItem = 32 splats for the item-parallel backward
// (k_preprocess / k_bin / k_scan_* / k_tile_scan: fallback for images with more tiles than fit an LDS table.)
//
// Replaces rasterize_gaussians (forward) of the reference's CUDA extension, called at
// /root/reference gaussian_splatting/gaussian_renderer/__init__.py:151-168.
#include <type_traits>

#include "launch.h"
#include "raster_kernels.h"
#include "objective_math.h"
#define MGS_DIAG_FORWARD
#include "diag_stamp.h"

namespace mgs {

__device__ __forceinline__ void load_camera(Camera& c, const KP& P) {
#pragma unroll
  for (int i = 0; i < 16; i++) { c.V[i] = P.V[i]; c.PM[i] = P.PM[i]; c.Praw[i] = P.Praw[i]; }
  c.campos[0] = P.campos[0]; c.campos[1] = P.campos[1]; c.campos[2] = P.campos[2];
  c.W = P.W; c.H = P.H; c.tanfovx = P.tanfovx; c.tanfovy = P.tanfovy;
  c.focal_x = P.focal_x; c.focal_y = P.focal_y; c.scale_modifier = P.mod;
  c.sh_degree = P.deg; c.sh_coeffs = P.K; c.grid_x = P.grid_x; c.grid_y = P.grid_y;
  c.clamp_grad_upstream = P.clamp_up;
}

// ---------------------------------------------------------------------------------
// Projection + EWA of Gaussian idx -> its 48-B record (stored) and radius; returns the two
// record quads the binning needs.
__device__ __forceinline__ void project_and_store(const KP& P, int idx, float4& r0, float4& r1) {
  Camera cam;
  load_camera(cam, P);
  const float p[3] = {P.means[3 * idx], P.means[3 * idx + 1], P.means[3 * idx + 2]};
  const float* pcol = P.precol ? P.precol + (size_t)3 * idx : nullptr;   // read in place
  const float* psh = P.shs ? P.shs + (size_t)3 * P.K * idx : nullptr;
  SplatRec rec;
  // two call sites instead of selecting between pointers to local arrays: a pointer select
  // forces the arrays into scratch memory
  if (P.covp) {
    float c6[6];
#pragma unroll
    for (int i = 0; i < 6; i++) c6[i] = P.covp[6 * (size_t)idx + i];
    project_gaussian(cam, p, nullptr, nullptr, c6, psh, pcol, P.opac[idx], rec);
  } else {
    const float sc[3] = {P.scales[3 * idx], P.scales[3 * idx + 1], P.scales[3 * idx + 2]};
    const float4 qq = reinterpret_cast<const float4*>(P.rots)[idx];
    const float q[4] = {qq.x, qq.y, qq.z, qq.w};
    project_gaussian(cam, p, sc, q, nullptr, psh, pcol, P.opac[idx], rec);
  }
  r0 = make_float4(rec.x, rec.y, rec.depth, rec.opacity);
  r1 = make_float4(rec.ca, rec.cb, rec.cc, __int_as_float(rec.radius));
  float4* dst = reinterpret_cast<float4*>(P.rec + idx);
  dst[0] = r0;
  dst[1] = r1;
  dst[2] = make_float4(rec.r, rec.g, rec.b, __uint_as_float(rec.flags));
  P.radii[idx] = rec.radius;
}

__global__ __launch_bounds__(kPreBlock) void k_preprocess(KP P) {
  const int idx = blockIdx.x * kPreBlock + threadIdx.x;
  for (int i = idx; i < P.T; i += gridDim.x * kPreBlock) P.tile_count[i] = 0;
  if (idx < 4) P.counters[idx] = 0;
  if (idx >= P.N) return;
  float4 r0, r1;
  project_and_store(P, idx, r0, r1);
}


// ---------------------------------------------------------------------------------
// Binning.  Count (emit = 0) or emit (emit = 1) the (tile, Gaussian) pairs.  The culling
// decision is taken by the same instruction stream in both modes (runtime flag, one
// kernel), so the two passes always agree.
//
// LDS-privatised form (default): kBinBlocks workgroups each own a contiguous chunk of
// Gaussians and a T-entry table in LDS.  Count mode builds the chunk's tile histogram
// with LDS atomics and stores it (coalesced) as row b of bin_table; k_bin_colsum turns
// the rows into exclusive per-block bases and the tile totals; emit mode reloads its row
// as LDS cursors and hands out positions with returning LDS atomics.  No global atomics:
// 2*D LDS atomics replace 2*D contended HBM-side atomics (185 + 251 us -> see DESIGN.md).
//
// Load balance: a lane walks its own Gaussian's tile rectangle only when that has at most
// kBinSmallRect tiles; larger rectangles (rare, but they would stall the other 63 lanes)
// are handed to the whole wave, 64 candidate tiles per step, with the pair index taken
// from a ballot prefix count.
constexpr int kBinSmallRect = 12;

__device__ __forceinline__ void bin_one_pair(const KP& P, int* s_tile, int emit, int t,
                                             unsigned long long key, unsigned int j) {
  const int pos = atomicAdd(&s_tile[t], 1);
  if (emit && pos < P.cap) {
    if (P.pack) {
      P.keys[pos] = (key & 0xFFFFFFFF00000000ull) | ((key & 0xFFFFFull) << kPackBits) | j;
    } else {
      P.keys[pos] = key;
      P.payload[pos] = j;
    }
  }
}

// project != 0 (count pass only): the projection itself runs here too - the count pass is the
// first consumer of the records, so k_preprocess's launch and the re-read of its output go away.
// THREADS: 1024 for large maps (a workgroup per CU), 256 for small ones (N <= kBinSmallMap: the cost
// of these passes is then their serial floor - table init, barriers, row write - and smaller, more
// numerous workgroups cut it: 8 k Gaussians 17.5 / 12.8 us -> see profiles/r02_forward_blend_tuning.txt).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_bin_lds(KP P, int emit, int per_block, int project) {
  constexpr int kBinThreads = THREADS;
  extern __shared__ int s_tile[];
  __shared__ int s_wave[kBinThreads / 64];
  int carry = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  int* row = P.bin_table + (size_t)blockIdx.x * P.T;
  for (int t = tid; t < P.T; t += kBinThreads) s_tile[t] = emit ? (P.tile_offset[t] + row[t]) : 0;
  if (project && blockIdx.x == 0 && tid < 4) P.counters[tid] = 0;   // consumed by later launches only
  if (emit)   // the blend pass accumulates n_touched; every (re)run of stage 2 starts from zero
    for (int i = blockIdx.x * kBinThreads + tid; i < P.N; i += gridDim.x * kBinThreads) P.n_touched[i] = 0;
  __syncthreads();
  const int g0 = blockIdx.x * per_block, g1 = min(P.N, g0 + per_block);
  for (int ibase = g0; ibase < g1; ibase += kBinThreads) {
    const int idx = ibase + tid;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    int radius = 0;
    unsigned int hit = 0u;        // emit: the count pass's verdicts on this Gaussian's (small) rectangle
    if (idx < g1) {
      if (project) {
        project_and_store(P, idx, r0, r1);
      } else {
        r0 = reinterpret_cast<const float4*>(P.rec + idx)[0];
        r1 = reinterpret_cast<const float4*>(P.rec + idx)[1];
        if (emit) hit = P.hit_mask[idx];
      }
      radius = __float_as_int(r1.w);
    }
    int cnt = 0;
    int rmin[2] = {0, 0}, rmax[2] = {0, 0};
    float qmax = -1.f;
    if (radius > 0) {
      tile_rect(r0.x, r0.y, radius, P.grid_x, P.grid_y, rmin, rmax);
      qmax = splat_qmax(r0.w);
    }
    const int rw = rmax[0] - rmin[0], area = rw * (rmax[1] - rmin[1]);
    const unsigned long long key = (((unsigned long long)__float_as_uint(r0.z)) << 32) | (unsigned int)idx;
    if (area > 0 && area <= kBinSmallRect) {
      // The exact tile tests (closed-form minimum of the conic form over the tile box, up to twelve per
      // lane) run in the count pass only; it leaves one bit per tile of the rectangle and the emit pass
      // replays them - same decisions by construction, 6-7 us of VALU work less in the second pass.
      unsigned int bits = emit ? hit : 0u, bit = 1u;
      for (int ty = rmin[1]; ty < rmax[1]; ty++)
        for (int tx = rmin[0]; tx < rmax[0]; tx++, bit <<= 1) {
          bool ok;
          if (emit) {
            ok = (bits & bit) != 0u;
          } else {
            ok = tile_reachable(r0.x, r0.y, r1.x, r1.y, r1.z, qmax, tx, ty, P.W, P.H);
            if (ok) bits |= bit;
          }
          if (!ok) continue;
          bin_one_pair(P, s_tile, emit, ty * P.grid_x + tx, key, (unsigned int)cnt);
          cnt++;
        }
      if (!emit) hit = bits;
    }
    // wave-cooperative pass over the large rectangles of this wave
    unsigned long long big = __ballot(area > kBinSmallRect);
    while (big) {
      const int src = __ffsll((long long)big) - 1;
      big &= big - 1;
      const float bx = __shfl(r0.x, src), by = __shfl(r0.y, src);
      const float bA = __shfl(r1.x, src), bB = __shfl(r1.y, src), bC = __shfl(r1.z, src);
      const float bq = __shfl(qmax, src);
      const int bx0 = __shfl(rmin[0], src), by0 = __shfl(rmin[1], src);
      const int brw = __shfl(rw, src), barea = __shfl(area, src);
      const unsigned int klo = (unsigned int)__shfl((int)(unsigned int)key, src);
      const unsigned int khi = (unsigned int)__shfl((int)(key >> 32), src);
      const unsigned long long bkey = ((unsigned long long)khi << 32) | klo;
      int total = 0;
      for (int cb = 0; cb < barea; cb += 64) {
        const int i = cb + lane;
        bool ok = false;
        int t = 0;
        if (i < barea) {
          const int ry = i / brw, rx = i - ry * brw;
          ok = tile_reachable(bx, by, bA, bB, bC, bq, bx0 + rx, by0 + ry, P.W, P.H);
          t = (by0 + ry) * P.grid_x + bx0 + rx;
        }
        const unsigned long long m = __ballot(ok);
        if (ok) {
          const int j = total + __popcll(m & ((1ull << lane) - 1ull));
          bin_one_pair(P, s_tile, emit, t, bkey, (unsigned int)j);
        }
        total += __popcll(m);
      }
      if (lane == src) cnt = total;
    }
    if (!emit && idx < g1) {
      P.pair_count[idx] = cnt;
      P.hit_mask[idx] = hit;
    }
  }
  if (!emit) {
    // Block-local exclusive scan of the pair counts -> slot offset of every Gaussian inside the block, in a
    // second, light walk over the block's Gaussians (each thread re-reads the counts it has just stored).
    // Inside the main loop the scan's two workgroup barriers made every wave wait for the slowest one of
    // each trip (9 us of a wave's 29); here the work between the barriers is a load and six shuffles.
    for (int ibase = g0; ibase < g1; ibase += kBinThreads) {
      const int idx = ibase + tid;
      const int cnt = idx < g1 ? P.pair_count[idx] : 0;
      int incl = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      __syncthreads();
      if (lane == 63) s_wave[tid >> 6] = incl;
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < kBinThreads / 64; w++) {
        const int x = s_wave[w];
        if (w < (tid >> 6)) before += x;
        total += x;
      }
      if (idx < g1) P.pair_off[idx] = carry + before + incl - cnt;
      carry += total;
    }
    __syncthreads();
    for (int t = tid; t < P.T; t += kBinThreads) row[t] = s_tile[t];
    if (tid == 0) {
      P.scan_tmp[blockIdx.x] = carry;          // block total, scanned by k_tile_scan
    }
  }
}

// Fallback for images with more tiles than fit an LDS table: global atomics.
__global__ __launch_bounds__(256) void k_bin(KP P, int emit) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P.N) return;
  const float4 r0 = reinterpret_cast<const float4*>(P.rec + idx)[0];
  const float4 r1 = reinterpret_cast<const float4*>(P.rec + idx)[1];
  const int radius = __float_as_int(r1.w);
  int cnt = 0;
  if (radius > 0) {
    int rmin[2], rmax[2];
    tile_rect(r0.x, r0.y, radius, P.grid_x, P.grid_y, rmin, rmax);
    const float qmax = splat_qmax(r0.w);
    const unsigned long long keyhi = ((unsigned long long)__float_as_uint(r0.z)) << 32;
    for (int ty = rmin[1]; ty < rmax[1]; ty++) {
      for (int tx = rmin[0]; tx < rmax[0]; tx++) {
        if (!tile_reachable(r0.x, r0.y, r1.x, r1.y, r1.z, qmax, tx, ty, P.W, P.H)) continue;
        const int t = ty * P.grid_x + tx;
        if (emit) {
          const int pos = P.tile_offset[t] + atomicAdd(&P.tile_cursor[t], 1);
          if (pos < P.cap) {
            P.keys[pos] = keyhi | (unsigned int)idx;
            P.payload[pos] = (unsigned int)cnt;
          }
        } else {
          atomicAdd(&P.tile_count[t], 1);
        }
        cnt++;
      }
    }
  }
  if (!emit) P.pair_count[idx] = cnt;
}

// ---------------------------------------------------------------------------------
// Fallback path only (more tiles than fit the LDS table): pair_off = exclusive scan of
// pair_count over all N Gaussians, three small launches.
__global__ __launch_bounds__(256) void k_scan_reduce(KP P) {
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
  if (tid == 0) P.scan_tmp[blockIdx.x] = s[0];
}

__global__ __launch_bounds__(1024) void k_scan_sums(KP P, int nblk) {
  __shared__ int s[1024];
  const int tid = threadIdx.x;
  int carry = 0;
  for (int base = 0; base < nblk; base += 1024) {
    const int i = base + tid;
    const int v = (i < nblk) ? P.scan_tmp[i] : 0;
    s[tid] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = (tid >= off) ? s[tid - off] : 0;
      __syncthreads();
      s[tid] += t;
      __syncthreads();
    }
    if (i < nblk) P.scan_tmp[i] = carry + s[tid] - v;
    carry += s[1023];
    __syncthreads();
  }
  if (tid == 0) P.scan_tmp[nblk] = carry;
}

__global__ __launch_bounds__(256) void k_scan_write(KP P) {
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
  int run = P.scan_tmp[blockIdx.x] + s[tid] - sum;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    if (base + i < P.N) P.pair_off[base + i] = run;
    run += v[i];
  }
}

// ---------------------------------------------------------------------------------
// Exclusive scans of the T tile counts (-> tile_offset, D) and of the per-tile segment
// backward item counts ceil(n_t / kItem) (-> seg_offset) by one 1024-thread workgroup.
__device__ __forceinline__ void tile_scan_body(const KP& P, int nbin, int* s_sum, int* s_seg) {
  // Three exclusive scans over the 1024 threads at once (binning-block totals, tile pair counts,
  // tile item counts) + the largest tile: inclusive scans inside each wave with shuffles, the sixteen
  // wave totals through LDS - ONE workgroup barrier (the Hillis-Steele form took forty).
  int* s_w = s_sum;            // [4][16]: wave totals of the three scans, wave maxima
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int vb = (nbin > 0 && tid < nbin) ? P.scan_tmp[tid] : 0;     // nbin <= 1024
  const int per = (P.T + 1023) / 1024;
  const int lo = tid * per, hi = min(lo + per, P.T);
  int local = 0, lseg = 0, lmax = 0;
  for (int i = lo; i < hi; i++) {
    const int c = __hip_atomic_load(&P.tile_count[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    local += c;
    lseg += (c + kItem - 1) / kItem;
    lmax = max(lmax, c);
  }
  int ib = vb, ip = local, is = lseg;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int tb = __shfl_up(ib, off), tp = __shfl_up(ip, off), ts = __shfl_up(is, off);
    if (lane >= off) { ib += tb; ip += tp; is += ts; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) lmax = max(lmax, __shfl_xor(lmax, off));
  __syncthreads();             // the caller may still be reading the LDS it lends us
  if (lane == 63) { s_w[wave] = ib; s_w[16 + wave] = ip; s_w[32 + wave] = is; s_w[48 + wave] = lmax; }
  __syncthreads();
  int ob = 0, op = 0, os = 0, tot_p = 0, tot_s = 0, tmax = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const int xb = s_w[w], xp = s_w[16 + w], xs = s_w[32 + w];
    if (w < wave) { ob += xb; op += xp; os += xs; }
    tot_p += xp; tot_s += xs;
    tmax = max(tmax, s_w[48 + w]);
  }
  if (nbin > 0) {   // exclusive scan of the binning blocks' pair totals
    if (tid < nbin) P.block_prefix[tid] = ob + ib - vb;
  } else if (tid == 0) {   // fallback path: pair_off already holds the global scan
    P.block_prefix[0] = 0;
  }
  int run = op + ip - local, rseg = os + is - lseg;
  for (int i = lo; i < hi; i++) {
    const int c = __hip_atomic_load(&P.tile_count[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    P.tile_offset[i] = run;
    P.seg_offset[i] = rseg;
    run += c;
    rseg += (c + kItem - 1) / kItem;
  }
  if (tid == 1023) {
    P.tile_offset[P.T] = tot_p;
    P.seg_offset[P.T] = tot_s;
    P.counters[0] = tot_p;
    if (P.d_out) {   // D and the largest tile, for the caller's capacity check / choice of sort launches
      __hip_atomic_store(P.d_out + 1, tmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(P.d_out, tot_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    P.counters[1] = tot_s;
    if (P.d_max) atomicMax(P.d_max, tot_p);
  }
  (void)s_seg;
}

__global__ __launch_bounds__(1024) void k_tile_scan(KP P, int nbin) {
  __shared__ int s_sum[1024], s_seg[1024];
  tile_scan_body(P, nbin, s_sum, s_seg);
}

// ---------------------------------------------------------------------------------
// Exclusive prefix down the rows of bin_table, total -> tile_count.  64 tiles x 16 row
// groups per workgroup; each thread owns kBinBlocks/8 rows in registers (independent
// loads), the groups are stitched through LDS.
constexpr int kColGroups = 16;
constexpr int kColRows = kBinBlocks / kColGroups;

__global__ __launch_bounds__(64 * kColGroups) void k_bin_colsum(KP P, int nblk) {
  __shared__ int s_sum2[2048];
  int (*s_sum)[64] = reinterpret_cast<int (*)[64]>(s_sum2);
  const int tl = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tl;
  int v[kColRows];
  int sum = 0;
#pragma unroll
  for (int i = 0; i < kColRows; i++) {
    const int b = grp * kColRows + i;
    v[i] = (t < P.T && b < nblk) ? P.bin_table[(size_t)b * P.T + t] : 0;
    sum += v[i];
  }
  s_sum[grp][tl] = sum;
  __syncthreads();
  int run = 0, tot = 0;
#pragma unroll
  for (int g = 0; g < kColGroups; g++) {
    const int x = s_sum[g][tl];
    if (g < grp) run += x;
    tot += x;
  }
  if (t < P.T) {
#pragma unroll
    for (int i = 0; i < kColRows; i++) {
      const int b = grp * kColRows + i;
      if (b < nblk) P.bin_table[(size_t)b * P.T + t] = run;
      run += v[i];
    }
    if (grp == 0) __hip_atomic_store(&P.tile_count[t], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // The workgroup that finishes last also runs the tile scan (saves a dependent launch).
  // Fence-free hand-off (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads"): the tile
  // totals are stored write-through (sc1) by wave 0 (see above), that wave drains them and one
  // of its lanes takes an agent-scope ticket; the last arriver reads tile_count with sc1 loads
  // behind a workgroup barrier.  counters[3] was zeroed by k_preprocess of this forward.
  __shared__ int s_last;
  if (threadIdx.x < 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      const int ticket = atomicAdd(&P.counters[3], 1);
      s_last = ticket == (int)gridDim.x - 1;
    }
  }
  __syncthreads();
  if (!s_last) return;
  tile_scan_body(P, nblk, s_sum2, s_sum2 + 1024);
}


// ---------------------------------------------------------------------------------
// Bitonic network for arbitrary n with virtual +inf padding: every comparator puts the
// smaller key at the lower index, comparators whose upper index is >= n are no-ops.
// Keys are unique (depth bits | Gaussian id), so the result is a deterministic total
// order whatever order the emit pass filled the segment in.
//
// Synchronisation: the m/2 comparators of a sub-stage are dealt to the waves in contiguous
// runs, so wave w only touches elements [w*chunk, (w+1)*chunk) whenever the sub-stage's
// block size is <= chunk.  Those sub-stages (all but 3 for m = 1024 on 4 waves) need no
// workgroup barrier - LDS operations of one wave execute in order - only the few with
// larger blocks are bracketed by __syncthreads().  WAVE_LOCAL = false (HBM fallback for
// oversized tiles) keeps a barrier after every sub-stage.
template <bool WAVE_LOCAL, bool HAS_VAL, int NWAVES, typename KP_, typename VP_>
__device__ __forceinline__ void bitonic_sort(KP_ key, VP_ val, int n, int tid) {
  int m = 2;
  while (m < n) m <<= 1;
  const int half_m = m >> 1;
  const int nw = WAVE_LOCAL ? max(1, min(NWAVES, m >> 7)) : NWAVES;     // active waves
  const int chunk = m / nw;                                   // elements owned by a wave
  const int cpw = half_m / nw;                                // comparators per wave
  const int wave = tid >> 6, lane = tid & 63;
  const bool active = wave < nw;
  bool prev_global = true;                                    // data was just loaded by all
  auto sync = [&](bool global) {
    if (!WAVE_LOCAL || global || prev_global) {
      __syncthreads();
    } else {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    prev_global = global;
  };
  // One LDS round trip per sub-stage: all keys and payloads of the thread's comparators
  // are read first (independent loads), swaps are written afterwards.
  constexpr int kMaxCmp = 2;   // comparators in flight per thread (m = 1024 on 4 waves: exactly 2)
  auto run = [&](auto&& pair_of) {
    for (int c0 = 0; c0 < cpw; c0 += 64 * kMaxCmp) {
      int lo[kMaxCmp], hi[kMaxCmp];
      unsigned long long a[kMaxCmp], b[kMaxCmp];
      unsigned int va[kMaxCmp], vb[kMaxCmp];
#pragma unroll
      for (int u = 0; u < kMaxCmp; u++) {
        const int c = c0 + lane + 64 * u;
        lo[u] = -1;
        if (c < cpw) {
          pair_of(wave * cpw + c, lo[u], hi[u]);
          if (hi[u] >= n) lo[u] = -1;
        }
        if (lo[u] >= 0) {
          a[u] = key[lo[u]]; b[u] = key[hi[u]];
          if constexpr (HAS_VAL) { va[u] = val[lo[u]]; vb[u] = val[hi[u]]; }
        }
      }
#pragma unroll
      for (int u = 0; u < kMaxCmp; u++) {
        if (lo[u] >= 0 && a[u] > b[u]) {
          key[lo[u]] = b[u]; key[hi[u]] = a[u];
          if constexpr (HAS_VAL) { val[lo[u]] = vb[u]; val[hi[u]] = va[u]; }
        }
      }
    }
  };
  // Register-resident sub-stages: a thread loads four consecutive elements (32 B, conflict-free
  // wide LDS reads) and runs the compare-exchanges at distance 2 and 1 - or the whole k = 2 and
  // k = 4 rounds - in registers: one LDS round trip instead of two or three, and none of the
  // 2-way bank conflicts the short-distance stages have.  Elements beyond n are virtual +inf.
  auto reg_pass = [&](bool first_rounds) {
    constexpr unsigned long long kInf = ~0ull;
    for (int e0 = 4 * lane; e0 < chunk; e0 += 256) {
      const int b0 = wave * chunk + e0;
      if (b0 >= n) continue;
      unsigned long long k4[4];
      unsigned int v4[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        k4[i] = b0 + i < n ? key[b0 + i] : kInf;
        if constexpr (HAS_VAL) v4[i] = b0 + i < n ? val[b0 + i] : 0u;
      }
      auto ce = [&](int a, int b) {
        if (k4[a] > k4[b]) {
          const unsigned long long t = k4[a]; k4[a] = k4[b]; k4[b] = t;
          if constexpr (HAS_VAL) { const unsigned int u = v4[a]; v4[a] = v4[b]; v4[b] = u; }
        }
      };
      if (first_rounds) { ce(0, 1); ce(2, 3); ce(0, 3); ce(1, 2); ce(0, 1); ce(2, 3); }
      else { ce(0, 2); ce(1, 3); ce(0, 1); ce(2, 3); }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (b0 + i < n) {
          key[b0 + i] = k4[i];
          if constexpr (HAS_VAL) val[b0 + i] = v4[i];
        }
      }
    }
  };
  int k = 2;
  if (m >= 4) {        // rounds k = 2 and k = 4
    sync(4 > chunk);
    if (active) reg_pass(true);
    k = 8;
  }
  for (; k <= m; k <<= 1) {
    {
      sync(k > chunk);
      const int half = k >> 1;
      if (active)
        run([&](int i, int& lo, int& hi) {
          const int blk = i / half, off = i - blk * half;
          lo = blk * k + off; hi = blk * k + k - 1 - off;
        });
    }
    for (int j = k >> 2; j >= 4; j >>= 1) {
      sync(2 * j > chunk);
      if (active)
        run([&](int i, int& lo, int& hi) {
          lo = 2 * i - (i & (j - 1)); hi = lo + j;
        });
    }
    if (k >= 8) {      // distances 2 and 1
      sync(4 > chunk);
      if (active) reg_pass(false);
    }
  }
  __syncthreads();
}

// Two launches by tile size class, so that the common small tiles run at 12 KB of LDS per
// workgroup (high occupancy) and only crowded tiles pay for 48 KB; tiles beyond 4096 pairs
// sort in place in HBM with the same network.
template <int CAP, int MIN_N, bool PACKED, int THREADS>
__global__ __launch_bounds__(THREADS) void k_tile_sort(KP P) {
  __shared__ unsigned long long s_key[CAP];
  __shared__ unsigned int s_val[PACKED ? 1 : CAP];
  const int tid = threadIdx.x;
  for (int tile = blockIdx.x; tile < P.T; tile += gridDim.x) {
  int start = P.tile_offset[tile], end = P.tile_offset[tile + 1];
  const int n_all = end - start;                    // as counted by the scans (seg_offset)
  start = min(start, P.cap); end = min(end, P.cap);
  const int n = end - start;
  if (MIN_N == 0) {   // item -> tile map for the item-parallel backward
    // EVERY item the scan counted gets a record - also those cut off by an undersized pair
    // capacity (0 splats), so that the backward never reads an unwritten record
    const int s0 = P.seg_offset[tile], ns = (n_all + kItem - 1) / kItem;
    for (int i = tid; i < ns; i += THREADS)
      if (s0 + i < P.max_segs)
        P.seg_rec[s0 + i] = make_int4(tile, min(start + i * kItem, P.cap), max(0, min(kItem, n - i * kItem)), i * kItem);
  }
  if (n <= 1 || n <= MIN_N || (MIN_N == 0 && n > CAP && P.big_pass)) continue;   // workgroup-uniform
  unsigned long long* gk = P.keys + start;
  unsigned int* gv = PACKED ? nullptr : P.payload + start;
  if (n <= CAP) {
    __syncthreads();      // the previous tile of this workgroup is done with the LDS arrays
    for (int i = tid; i < n; i += THREADS) {
      s_key[i] = gk[i];
      if constexpr (!PACKED) s_val[i] = gv[i];
    }
    bitonic_sort<true, !PACKED, THREADS / 64>(s_key, s_val, n, tid);
    for (int i = tid; i < n; i += THREADS) {
      gk[i] = s_key[i];
      if constexpr (!PACKED) gv[i] = s_val[i];
    }
  } else {
    bitonic_sort<false, !PACKED, THREADS / 64>(gk, gv, n, tid);
  }
  }
}

// ---------------------------------------------------------------------------------
// Register-resident sort of the common size class (n <= 1024): 256 threads, four consecutive
// elements per thread.  The network is the one above (per merge size k: partner e ^ (k - 1), then
// e ^ j for j = k/4 ... 1; the lower index keeps the smaller key), but an exchange only leaves the
// registers when it has to:
//   partner in the same thread (distances 1, 2, 3)         -> compare-exchange in registers
//   partner in the same wave (lane xor 1 ... 63)           -> ds_bpermute (LDS crossbar, no memory,
//                                                             no barrier)
//   partner in another wave (k = 512, 1024: three stages)  -> through LDS with a barrier
// 19 + 33 + 3 stages for 1024 keys instead of 55 LDS round trips.  Elements beyond n are
// virtual +inf (no real key has all bits set: the depth word is a positive float).
template <int CTRL, int BANK>
__device__ __forceinline__ int dpp_move(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK, false); }
template <int X>
__device__ __forceinline__ int lane_xor_dpp(int v) {     // value of lane (lane ^ X); checked on gfx950 for every X
  if constexpr (X == 1) return dpp_move<0xB1, 0xF>(v, v);          // quad_perm [1,0,3,2]
  else if constexpr (X == 2) return dpp_move<0x4E, 0xF>(v, v);     // quad_perm [2,3,0,1]
  else if constexpr (X == 3) return dpp_move<0x1B, 0xF>(v, v);     // quad_perm [3,2,1,0]
  else if constexpr (X == 7) return dpp_move<0x141, 0xF>(v, v);    // row_half_mirror
  else if constexpr (X == 15) return dpp_move<0x140, 0xF>(v, v);   // row_mirror
  else if constexpr (X == 8) return dpp_move<0x128, 0xF>(v, v);    // row_ror:8
  else { static_assert(X == 4, "no DPP form"); const int t = dpp_move<0x12C, 0x5>(v, v); return dpp_move<0x124, 0xA>(t, v); }
}

template <bool PACKED>
__global__ __launch_bounds__(256) void k_tile_sort_reg(KP P) {
  __shared__ unsigned long long s_xk[1024];
  __shared__ unsigned int s_xv[PACKED ? 1 : 1024];
  const int tid = threadIdx.x, lane = tid & 63, tile = blockIdx.x;
  int start = P.tile_offset[tile], end = P.tile_offset[tile + 1];
  const int n_all = end - start;                    // as counted by the scans (seg_offset)
  start = min(start, P.cap); end = min(end, P.cap);
  const int n = end - start;
  {   // item -> tile map for the item-parallel backward
    // EVERY item the scan counted gets a record - also those cut off by an undersized pair
    // capacity (0 splats), so that the backward never reads an unwritten record
    const int s0 = P.seg_offset[tile], ns = (n_all + kItem - 1) / kItem;
    for (int i = tid; i < ns; i += 256)
      if (s0 + i < P.max_segs)
        P.seg_rec[s0 + i] = make_int4(tile, min(start + i * kItem, P.cap), max(0, min(kItem, n - i * kItem)), i * kItem);
  }
  if (n <= 1) return;
  unsigned long long* gk = P.keys + start;
  unsigned int* gv = PACKED ? nullptr : P.payload + start;
  if (n > 1024) {          // workgroup-uniform
    if (!P.big_pass) bitonic_sort<false, !PACKED, 4>(gk, gv, n, tid);   // in place in HBM (no second launch)
    return;
  }
  constexpr unsigned long long kInf = ~0ull;
  unsigned long long k[4];
  unsigned int v[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int e = 4 * tid + r;
    k[r] = e < n ? gk[e] : kInf;
    if constexpr (!PACKED) v[r] = e < n ? gv[e] : 0u;
  }
  int m = 4;
  while (m < n) m <<= 1;
  // compare-exchange inside the thread: the lower index keeps the smaller key
  auto ce = [&](int a, int b) {
    if (k[a] > k[b]) {
      const unsigned long long t = k[a]; k[a] = k[b]; k[b] = t;
      if constexpr (!PACKED) { const unsigned int u = v[a]; v[a] = v[b]; v[b] = u; }
    }
  };
  // keep the smaller (lower thread) or the larger (upper thread) of own / partner element
  auto keep = [&](int r, unsigned long long pk, unsigned int pv, bool lower) {
    const bool take = (pk < k[r]) == lower;          // keys are unique
    k[r] = take ? pk : k[r];
    if constexpr (!PACKED) v[r] = take ? pv : v[r];
  };
  // exchange with thread tid ^ X (X > 0); FLIP: the partner of element r is the other thread's 3 - r
  auto exchange = [&](auto flip_tag, int X) {
    constexpr bool FLIP = decltype(flip_tag)::value;
    // the thread with the lower index holds the lower elements: decided by the highest bit of X
    const bool lower = (tid & (1 << (31 - __builtin_clz(X)))) == 0;
    unsigned long long pk[4];
    unsigned int pv[4] = {0u, 0u, 0u, 0u};
    if (X < 64) {
      // lane xor 1, 2, 3, 7, 8, 15 is one DPP move (quad_perm / row_half_mirror / row_ror:8 /
      // row_mirror), xor 4 is two (row_ror:12 on banks 0 and 2, row_ror:4 on banks 1 and 3): VALU
      // work of this SIMD.  The rest goes through ds_bpermute, whose throughput is shared by the
      // whole CU and bounded the first version of this kernel (36 exchange stages x 8 dwords).
      const int addr = (lane ^ X) << 2;
      auto fetch_all = [&](auto xc) {           // one switch per stage, the eight moves below it
        constexpr int XC = decltype(xc)::value;
        auto from_partner = [&](int w) -> int {
          if constexpr (XC == 0) return __builtin_amdgcn_ds_bpermute(addr, w);
          else return lane_xor_dpp<XC>(w);
        };
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const unsigned long long mine = k[FLIP ? 3 - r : r];
          const unsigned int lo = (unsigned int)from_partner((int)(unsigned int)mine);
          const unsigned int hi = (unsigned int)from_partner((int)(unsigned int)(mine >> 32));
          pk[r] = ((unsigned long long)hi << 32) | lo;
          if constexpr (!PACKED) pv[r] = (unsigned int)from_partner((int)v[FLIP ? 3 - r : r]);
        }
      };
      switch (X) {
        case 1: fetch_all(std::integral_constant<int, 1>{}); break;
        case 2: fetch_all(std::integral_constant<int, 2>{}); break;
        case 3: fetch_all(std::integral_constant<int, 3>{}); break;
        case 4: fetch_all(std::integral_constant<int, 4>{}); break;
        case 7: fetch_all(std::integral_constant<int, 7>{}); break;
        case 8: fetch_all(std::integral_constant<int, 8>{}); break;
        case 15: fetch_all(std::integral_constant<int, 15>{}); break;
        default: fetch_all(std::integral_constant<int, 0>{}); break;
      }
    } else {
      __syncthreads();                 // the previous exchange has been read
#pragma unroll
      for (int r = 0; r < 4; r++) {
        s_xk[4 * tid + r] = k[r];
        if constexpr (!PACKED) s_xv[4 * tid + r] = v[r];
      }
      __syncthreads();
      const int pt = tid ^ X;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        pk[r] = s_xk[4 * pt + (FLIP ? 3 - r : r)];
        if constexpr (!PACKED) pv[r] = s_xv[4 * pt + (FLIP ? 3 - r : r)];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) keep(r, pk[r], pv[r], lower);
  };
  // merge sizes 2 and 4 stay inside the thread
  ce(0, 1); ce(2, 3);
  ce(0, 3); ce(1, 2); ce(0, 1); ce(2, 3);
  for (int kk = 8; kk <= m; kk <<= 1) {
    exchange(std::true_type{}, (kk - 1) >> 2);                          // partner e ^ (kk - 1)
    for (int j = kk >> 2; j >= 4; j >>= 1) exchange(std::false_type{}, j >> 2);   // partner e ^ j
    ce(0, 2); ce(1, 3); ce(0, 1); ce(2, 3);                             // j = 2, 1
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int e = 4 * tid + r;
    if (e < n) {
      gk[e] = k[r];
      if constexpr (!PACKED) gv[e] = v[r];
    }
  }
}

// ---------------------------------------------------------------------------------
// Front-to-back blend.  One WAVE per 8x8 quadrant of a tile (one pixel per lane), launched
// as its own 64-thread workgroup: the four quadrant waves of a tile are fully independent,
// so there is no workgroup barrier anywhere, a saturated quadrant retires at once, and the
// dispatcher balances 4T equal-shaped items instead of T ragged ones.  (A CU admits
// workgroups only up to 64 KB of LDS in total - measured with in-kernel stamps: the former
// 256-thread / 13.5 KB form ran 4 deep per CU and the last 176 tiles started 50 us late -
// this form needs 2.5 KB.)
//
// The wave walks the tile's sorted list 64 splats (one segment) at a time: lane l loads
// splat l's record and tests whether the splat can reach THIS quadrant at all (closed-form
// bound of the tile culling on the 8x8 pixel box - exact: a skipped splat has alpha < 1/255
// on all 64 pixels); reachable splats are parked in LDS with the conic pre-multiplied by
// -0.5*log2(e) (a pixel evaluation is a few FMAs + one v_exp_f32) and the wave visits them
// in list order by scanning the ballot mask.  Global loads are software-pipelined: ids two
// segments ahead, records one segment ahead, so the id -> record dependent latency hides
// under the previous segment's arithmetic.  In front of every backward item (kItem splats) the
// per-pixel blend state (T, C, D) is checkpointed for the item-parallel backward; a quadrant
// whose pixels are all saturated stops writing checkpoints (the backward never reads the state
// of a pixel whose n_contrib lies in front of the item).
constexpr float kLog2e = 1.4426950408889634f;
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kFwdPrioStep = 128;  // splats of remaining list per s_setprio level
constexpr int kFwdChunk = 4;       // the four quadrants of a tile share an XCD (same records)
static_assert(kSeg == 64, "one staged record per lane");

__device__ __forceinline__ unsigned int fwd_load_id(const KP& P, int start, int n, int k) {
  // low half of the 64-bit key = Gaussian id (<< kPackBits | pair index when packed)
  const unsigned int lo = k < n ? reinterpret_cast<const unsigned int*>(P.keys + start)[2 * (size_t)k] : 0u;
  return P.pack ? lo >> kPackBits : lo;
}

// m &= ~(1 << j) on a wave-uniform 64-bit mask in ONE scalar instruction (the C form
// m &= m - 1 costs s_add_u32 + s_addc_u32 + s_and_b64).
__device__ __forceinline__ void mask_clear_bit(unsigned long long& m, int j) {
  asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(j));
}


// OBJ: the tracking objective in the epilogue (raster_kernels.h: KObj).
template <bool OBJ>
__global__ __launch_bounds__(64, 6) void k_blend_fwd(KP P) {
  MGS_STAMP_SCOPE;
  // per staged splat 48 B: (a0, a1, a2, A) (B, C, opacity, -) (r, g, b, depth)
  __shared__ float4 s_rec[kSeg * 3];
#ifdef MGS_FWD_BLOCKS
  // experiment: an XCD takes 4x4 blocks of tiles (launch-order index -> tile through tile_from_block_order)
  const int item0 = xcd_remap<64>(blockIdx.x);
  if (item0 >= 4 * P.T) return;
  const int tile = tile_from_block_order(item0 >> 2, P.grid_x, P.grid_y), quad = item0 & 3, lane = threadIdx.x;
  const int item = 4 * tile + quad;
#else
  const int item = xcd_remap<kFwdChunk>(blockIdx.x);
  if (item >= 4 * P.T) return;
  const int tile = item >> 2, quad = item & 3, lane = threadIdx.x;
#endif
  const int tx = tile % P.grid_x, ty = tile / P.grid_x;
  const int qx0 = tx * kTile + 8 * (quad & 1), qy0 = ty * kTile + 8 * (quad >> 1);
  if (qx0 >= P.W || qy0 >= P.H) {                        // quadrant outside the image
    if constexpr (OBJ) {                                 // ... still owns an entry of the objective's partial sums
      if (lane < 4) P.obj.partial[(size_t)lane * 4 * P.T + item] = 0.f;
    }
    return;
  }
  const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
  const int ptile = quad * 64 + lane;                    // pixel index in the tile, quadrant-major
  const bool inside = px < P.W && py < P.H;
  int start = P.tile_offset[tile], end = P.tile_offset[tile + 1];
  start = min(start, P.cap); end = min(end, P.cap);
  const int n = end - start;
  const int seg0 = P.seg_offset[tile];
  // this quadrant's pixel-centre box, clipped to the image
  const float bx0 = (float)qx0, by0 = (float)qy0;
  const float bx1 = fminf(bx0 + 7.f, (float)(P.W - 1)), by1 = fminf(by0 + 7.f, (float)(P.H - 1));
  // T > 0: transmittance of a live pixel (T >= kTStop is invariant).  A pixel that has saturated -
  // or lies outside the image - keeps its transmittance with the SIGN flipped: T (1 - alpha) is
  // then negative, "test_T < kTStop" fires again at every later splat and nothing is blended, so
  // no separate live flag is carried through the visit.
  float T = inside ? 1.f : -1.f;
  // pixel offset from the quadrant centre: the exponent is evaluated as a polynomial in it
  const float xh = (float)(lane & 7) - 3.5f, yh = (float)(lane >> 3) - 3.5f;
  const float qcx = (float)qx0 + 3.5f, qcy = (float)qy0 + 3.5f;
  v2f C01 = {0.f, 0.f}, C2D = {0.f, 0.f};     // (C0, C1), (C2, depth): packed-FMA operands
  int last = 0;

  // pipeline prologue: ids of segments 0 and 1, records of segment 0
  // (ids default to 0, a valid record, so the record loads need no branch)
  unsigned int id_n1 = fwd_load_id(P, start, n, kSeg + lane);
  unsigned int cid = fwd_load_id(P, start, n, lane);
  const float4* src0 = reinterpret_cast<const float4*>(P.rec + MGS_ABL_REC(cid));
  // 16 + 12 + 12 bytes: the fourth dwords of the conic and colour rows are not used here, and a
  // dead component of a wide load is a free register to the allocator - anything it parks there
  // has to wait for the load to land (it cost one memory round trip per segment).
  float4 ca = src0[0];
  float3 cb = *reinterpret_cast<const float3*>(src0 + 1), cc = *reinterpret_cast<const float3*>(src0 + 2);

  int touched_prev = 0;
  unsigned int cid_prev = 0u;
  // kItem < kSeg: a backward item starts in the MIDDLE of a segment, too.  The state in front of
  // it is parked in registers and stored with the next iteration's memory traffic (a store issued
  // in the middle of the walk would be the youngest operation at the loop's s_waitcnt).
  constexpr int kParts = kItem < kSeg ? kSeg / kItem : 1;
  static_assert(kParts <= 2 && (kItem >= kSeg || kSeg % kItem == 0), "at most one checkpoint inside a segment");
  int sg_mid = -1;
  float mid_T = 0.f;
  v2f mid_01 = {0.f, 0.f}, mid_2D = {0.f, 0.f};
  auto flush_mid = [&]() {
    if (MGS_ABL_CKPT && sg_mid >= 0 && sg_mid < P.max_segs) {
      float* ck = P.ckpt + (size_t)sg_mid * (5 * 256);
      reinterpret_cast<float4*>(ck)[ptile] = make_float4(mid_T, mid_01.x, mid_01.y, mid_2D.x);
      ck[1024 + ptile] = mid_2D.y;
    }
    sg_mid = -1;
  };
  for (int base = 0; base < n; base += kSeg) {
    {   // Longest-remaining-first among the waves that share a SIMD: with equal priorities the short
        // lists finish early and the long ones are left to run alone at one instruction per ~4.4
        // cycles; with the long lists served first the SIMD stays shared until the end
        // (88 -> 77 us on SYN-C; steps of 64 ... 512 splats measured, profiles/r02_forward_blend_tuning.txt).
      const int rem = n - base;
      if (rem > 3 * kFwdPrioStep) __builtin_amdgcn_s_setprio(3);
      else if (rem > 2 * kFwdPrioStep) __builtin_amdgcn_s_setprio(2);
      else if (rem > kFwdPrioStep) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    MGS_MARK(3);
    // everything in flight was issued a whole walk ago: drain it here, on every path, so that the
    // compiler has no reason to wait (with a count sized for its shortest path) further down
    asm volatile("" :: "v"(ca.x), "v"(cb.x), "v"(cc.x), "v"(id_n1));
    // The reach test comes FIRST and all memory traffic of the iteration is issued behind it: the
    // wait in front of the test then only covers operations issued a whole walk earlier.  (s_waitcnt
    // counts in order and the compiler sizes it for the path with the fewest later operations; with
    // the next segment's loads, the checkpoint stores or the n_touched atomic in front of the test it
    // waited for those, one memory round trip per segment.)
    const bool reach = base + lane < n &&
                       box_reachable(ca.x, ca.y, cb.x, cb.y, cb.z, splat_qmax(ca.w), bx0, by0, bx1, by1);
    unsigned long long m = __builtin_amdgcn_ballot_w64(reach);
    MGS_STAMP_SEG(m);
    MGS_MARK(0);
    __builtin_amdgcn_sched_barrier(0);
    if (touched_prev > 0) atomicAdd(&P.n_touched[cid_prev], touched_prev);   // previous segment's counts
    if constexpr (kParts > 1) flush_mid();
    {   // this quadrant's reach bits of the segment's backward items: lane l stores the word of item base / kItem + l
      static_assert(kItem == 32 && kSeg == 64, "one 32-bit reach word per (item, quadrant)");
      const int sgr = seg0 + base / kItem + lane;
      if (lane < 2 && base + lane * kItem < n && sgr < P.max_segs)
        reinterpret_cast<unsigned int*>(P.reach)[4 * (size_t)sgr + quad] = lane ? (unsigned int)(m >> 32) : (unsigned int)m;
    }
    if (base > 0 && base % kItem == 0) {   // checkpoint: state in front of this backward item
      const int sg = seg0 + base / kItem;
      if (MGS_ABL_CKPT && sg < P.max_segs) {
        float* ck = P.ckpt + (size_t)sg * (5 * 256);
        reinterpret_cast<float4*>(ck)[ptile] = make_float4(fabsf(T), C01.x, C01.y, C2D.x);
        ck[1024 + ptile] = C2D.y;
      }
    }
    // loads of the NEXT segment (records) and of the one after it (ids)
    const unsigned int nid = id_n1;
    const float4* src1 = reinterpret_cast<const float4*>(P.rec + MGS_ABL_REC(nid));
    const float4 na = src1[0];
    const float3 nb4 = *reinterpret_cast<const float3*>(src1 + 1), nc = *reinterpret_cast<const float3*>(src1 + 2);
    id_n1 = fwd_load_id(P, start, n, base + 2 * kSeg + lane);
    __builtin_amdgcn_sched_barrier(0);
    int touched = 0;
    bool count_touch = false;
    if (m != 0ull) {
      __syncthreads();   // single-wave workgroup: orders the LDS traffic, no hardware barrier
      if (reach) {
        // log2 of the Gaussian falloff at pixel offset (x, y) from the quadrant centre, with
        // d0 = mean - centre and (A, B, C) = -log2(e) (conic.x / 2, conic.y, conic.z / 2):
        //   A (d0x - x)^2 + B (d0x - x)(d0y - y) + C (d0y - y)^2
        //     = a0 + x (a1 + A x + B y) + y (a2 + C y)
        // five FMAs per pixel on per-lane constants (x, y); |x|, |y| <= 3.5 keeps the cancellation
        // error of the expanded form at a few 1e-6 of the exponent wherever alpha matters.
        const float A = -0.5f * kLog2e * cb.x, B = -kLog2e * cb.y, Cc = -0.5f * kLog2e * cb.z;
        const float d0x = ca.x - qcx, d0y = ca.y - qcy;
        const float a0 = d0x * (A * d0x + B * d0y) + Cc * d0y * d0y;
        const float a1 = -(2.f * A * d0x + B * d0y), a2 = -(B * d0x + 2.f * Cc * d0y);
        s_rec[3 * lane] = make_float4(a0, a1, a2, A);
        s_rec[3 * lane + 1] = make_float4(B, Cc, ca.w, 0.f);           // (B, C, opacity, -)
        s_rec[3 * lane + 2] = make_float4(cc.x, cc.y, cc.z, ca.z);     // (r, g), (b, depth): packed-FMA pairs
      }
      __syncthreads();
      // n_touched only counts contributions made while T(1-alpha) > 0.5: once no pixel of the
      // quadrant is that transparent any more the counting code is skipped (wave-uniform)
      MGS_MARK(1);
      count_touch = __builtin_amdgcn_ballot_w64(T > kTouchT) != 0ull;
    }
    {
      // the walk is instantiated twice (with / without the counting code) so that the choice
      // costs one branch per segment instead of instructions in every visit
      auto walk = [&](auto touch_tag, unsigned long long m) {
        constexpr bool kTouch = decltype(touch_tag)::value;
        auto visit = [&](int j, const float4 u, const float4 v, const float4 c) {
          const float t1 = __builtin_fmaf(v.x, yh, __builtin_fmaf(u.w, xh, u.y));
          const float t2 = __builtin_fmaf(v.y, yh, u.z);
          // the quadratic form is <= 0; the clamp only removes rounding excursions of the expanded
          // form (the reference's "power > 0: skip" can fire on rounding alone, too)
          // (exp2 clamped to [0, 1] = exp2 of the exponent clamped at 0, and the clamp is an output modifier
          // of v_exp_f32: no instruction)
          const float pw = __builtin_fmaf(yh, t2, __builtin_fmaf(xh, t1, u.x));
          float a = fminf(kAlphaMax, v.z * exp2_sat(pw));
          a = a >= kAlphaMin ? a : 0.f;
          const float test_T = __builtin_fmaf(-a, T, T);
          // T >= kTStop is invariant on live pixels, so test_T < kTStop implies a > 0: the pixel
          // saturates here, this splat is NOT blended and nothing after it is (dead pixels: T < 0).
          const bool stop = test_T < kTStop;
          const float w = stop ? 0.f : a * T;
          const v2f ww = {w, w};
          const v2f rg = {c.x, c.y}, bdv = {c.z, c.w};
          C01 = __builtin_elementwise_fma(rg, ww, C01);
          C2D = __builtin_elementwise_fma(bdv, ww, C2D);
          T = stop ? -fabsf(T) : test_T;
          const bool contrib = w > 0.f;
          last = contrib ? (base + j + 1) : last;
          if constexpr (kTouch) {
            // n_touched: contributions made while T (1 - alpha) > 0.5; lane j owns splat j's count
            const unsigned long long tm = __builtin_amdgcn_ballot_w64(contrib) & __builtin_amdgcn_fcmpf(test_T, kTouchT, 2 /* OGT */);
            const int cnt = __popcll(tm);
            // v_writelane with two SGPR sources needs the lane select in M0 (one constant-bus read);
            // M0 is a reserved register, so it is handed back as it was found
            int tv = touched, m0_saved;
            asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                : "+v"(tv), "=&s"(m0_saved) : "s"(cnt), "s"(j));
            touched = tv;
          }
        };
        // two-way unrolled walk over the set bits of m: splat j in (u0, v0, w0), the next one
        // is prefetched into (u1, v1, w1) and vice versa, so no registers are rotated
        int j0 = __builtin_ctzll(m);
        float4 u0 = s_rec[3 * j0], v0 = s_rec[3 * j0 + 1], w0 = s_rec[3 * j0 + 2];
        while (true) {
          mask_clear_bit(m, j0);
          const int j1 = __builtin_ctzll(m) & 63;       // m == 0: harmless read of slot 63
          const float4 u1 = s_rec[3 * j1], v1 = s_rec[3 * j1 + 1], w1 = s_rec[3 * j1 + 2];
          __builtin_amdgcn_sched_barrier(0);           // keep the prefetch above the arithmetic
          visit(j0, u0, v0, w0);
          if (m == 0ull) break;
          mask_clear_bit(m, j1);
          j0 = __builtin_ctzll(m) & 63;
          u0 = s_rec[3 * j0]; v0 = s_rec[3 * j0 + 1]; w0 = s_rec[3 * j0 + 2];
          __builtin_amdgcn_sched_barrier(0);
          visit(j1, u1, v1, w1);
          if (m == 0ull) break;
        }
      };
      auto walk_part = [&](unsigned long long mm) {      // mm != 0 implies m != 0: the records are staged
        if (mm == 0ull) return;
        if (count_touch) walk(std::true_type{}, mm);
        else walk(std::false_type{}, mm);
      };
      if constexpr (kParts > 1) {
        walk_part(m & ((1ull << kItem) - 1ull));
        if (base + kItem < n) {
          mid_T = fabsf(T); mid_01 = C01; mid_2D = C2D;
          sg_mid = seg0 + (base + kItem) / kItem;
        }
        walk_part(m & ~((1ull << kItem) - 1ull));
      } else {
        walk_part(m);
      }
      MGS_MARK(2);
    }
    touched_prev = touched; cid_prev = cid;
    if (__builtin_amdgcn_ballot_w64(T > 0.f) == 0ull) break;   // quadrant saturated
    cid = nid; ca = na; cb = nb4; cc = nc;
  }
  if (touched_prev > 0) atomicAdd(&P.n_touched[cid_prev], touched_prev);
  if constexpr (kParts > 1) flush_mid();
  {   // state for the backward, quadrant-major (coalesced); lanes outside the image hold last = 0
    int lane_q = lane;
    asm volatile("" : "+v"(lane_q));
    const size_t qi = (size_t)tile * 256 + quad * 64 + lane_q;
    T = fabsf(T);
    P.final_TC[qi] = make_float4(T, C01.x, C01.y, C2D.x);
    P.final_DL[qi] = make_int2(__float_as_int(C2D.y), last);
    int m = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off));
    if (lane == 0) P.quad_last[4 * tile + quad] = m;
  }
  // the pixel position is derived again from the lane id (an opaque copy: otherwise px, py and
  // the inside flag stay live across the whole walk and the kernel needs a scratch slot)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int px_e = qx0 + (lane_e & 7), py_e = qy0 + (lane_e >> 3);
  if (px_e < P.W && py_e < P.H) {
    const size_t HW = (size_t)P.W * P.H;
    const size_t pix = (size_t)py_e * P.W + px_e;
    P.out_color[pix] = C01.x + T * P.bg[0];
    P.out_color[HW + pix] = C01.y + T * P.bg[1];
    P.out_color[2 * HW + pix] = C2D.x + T * P.bg[2];
    P.out_depth[pix] = C2D.y;
    P.out_opacity[pix] = 1.f - T;
  }
  if constexpr (OBJ) {
    // Tracking objective || Huber(opacity * mask * ((|a| + eps) * image + b - gt)) ||_p (p = 2 or 1), one pixel per lane:
    // the per-sample arithmetic of k_track_loss_onepass (tracking.hip), channel by channel; d(loss)/d(image)
    // WITHOUT the loss^(1-p) of the norm (k_pose_adam_update applies it to the pose gradient, which is linear in
    // it).  The wave's four sums become partial entry `item` (= 4 tile + quadrant).
    float acc = 0.f, ga = 0.f, gb = 0.f, l1 = 0.f;
    const float a = P.obj.exposure_a[0];
    if (px_e < P.W && py_e < P.H) {
      const size_t HW = (size_t)P.W * P.H;
      const size_t pix = (size_t)py_e * P.W + px_e;
      const float gain = fabsf(a) + P.obj.exposure_eps, bias = P.obj.exposure_b[0];
      const float om = (1.f - T) * (P.obj.mask ? P.obj.mask[pix] : 1.f);
      const float im[3] = {C01.x + T * P.bg[0], C01.y + T * P.bg[1], C2D.x + T * P.bg[2]};
      float gt[3];
#pragma unroll
      for (int c = 0; c < 3; c++) gt[c] = P.obj.gt[c * HW + pix];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float dh;
        const float r = om * (gain * im[c] + bias - gt[c]);
        l1 += fabsf(r);
        const float h = huber(r, P.obj.huber_delta, dh);
        float phi, gam;
        norm_terms(h, P.obj.p1 ? 1.f : 2.f, phi, gam);
        acc += phi;
        const float gr = gam * dh * om;
        ga += gr * im[c];
        gb += gr;
        P.obj.grad_image[c * HW + pix] = gr * gain;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      acc += __shfl_down(acc, off); ga += __shfl_down(ga, off); gb += __shfl_down(gb, off); l1 += __shfl_down(l1, off);
    }
    if (lane == 0) {
      const size_t n = 4 * (size_t)P.T;
      const float sgn = a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f);
      P.obj.partial[item] = acc;
      P.obj.partial[n + item] = ga * sgn;
      P.obj.partial[2 * n + item] = gb;
      P.obj.partial[3 * n + item] = l1;
    }
  }
}

// ---------------------------------------------------------------------------------
static inline int check_launch() {
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

static inline int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return cus;
}

// A 1024-thread binning workgroup fills a CU (its SGPR count admits one per CU), so a grid
// just above the CU count runs a second, nearly empty round: round down to a whole number of
// rounds (300k Gaussians: 293 -> 256 workgroups of 1172).
static inline int bin_blocks(int N) {
  if (N <= kBinSmallMap) return max(1, (N + 255) / 256);      // 256-thread workgroups, <= 256 of them
  int b = max(1, min(kBinBlocks, (N + 1023) / 1024));
  const int cus = num_cus();
  if (b > cus) b = b / cus * cus;
  return b;
}

// Gaussians per workgroup are rounded up to a multiple of kPreBlock, so that idx / per is uniform
// over a workgroup of the preprocess backward (its block_prefix entry is one scalar load).
void bin_grid(int N, int& nblk, int& per) {
  const int b = bin_blocks(N);
  per = ((N + b - 1) / b + kPreBlock - 1) / kPreBlock * kPreBlock;
  nblk = (N + per - 1) / per;
}

int launch_forward_project(const KP& P, hipStream_t st) {
  if (P.T <= kBinMaxTilesLds) {
    int nblk, per;
    bin_grid(P.N, nblk, per);
    if (P.N <= kBinSmallMap)
      launch_smem("project_bin_count", k_bin_lds<256>, dim3(nblk), dim3(256), sizeof(int) * (size_t)P.T, st, P, 0, per, 1);
    else
      launch_smem("project_bin_count", k_bin_lds<1024>, dim3(nblk), dim3(1024), sizeof(int) * (size_t)P.T, st, P, 0, per, 1);
    launch("bin_colsum", k_bin_colsum, dim3((P.T + 63) / 64), dim3(64 * kColGroups), st, P, nblk);   // + tile scan
  } else {
    launch("preprocess", k_preprocess, dim3((P.N + kPreBlock - 1) / kPreBlock), dim3(kPreBlock), st, P);
    launch("bin_count", k_bin, dim3((P.N + 255) / 256), dim3(256), st, P, 0);
    const int nscan = (P.N + kScanBlock - 1) / kScanBlock;
    launch("scan_reduce", k_scan_reduce, dim3(nscan), dim3(256), st, P);
    launch("scan_sums", k_scan_sums, dim3(1), dim3(1024), st, P, nscan);
    launch("scan_write", k_scan_write, dim3(nscan), dim3(256), st, P);
    launch("tile_scan", k_tile_scan, dim3(1), dim3(1024), st, P, 0);
  }
  return check_launch();
}

int launch_forward_blend(const KP& P, hipStream_t st) {
  // cursors restart at 0 on every call so a retry with a larger capacity is valid
  // (n_touched is zeroed by the emit pass of the LDS path)
  if (P.T <= kBinMaxTilesLds) {
    int nblk, per;
    bin_grid(P.N, nblk, per);
    if (P.N <= kBinSmallMap)
      launch_smem("bin_emit", k_bin_lds<256>, dim3(nblk), dim3(256), sizeof(int) * (size_t)P.T, st, P, 1, per, 0);
    else
      launch_smem("bin_emit", k_bin_lds<1024>, dim3(nblk), dim3(1024), sizeof(int) * (size_t)P.T, st, P, 1, per, 0);
  } else {
    if (!hip_ok("memset(tile cursors)", hipMemsetAsync(P.tile_cursor, 0, sizeof(int) * (size_t)P.T, st)) ||
        !hip_ok("memset(n_touched)", hipMemsetAsync(P.n_touched, 0, sizeof(int) * (size_t)P.N, st))) {
      launches_ok();
      return MGS_ERR_LAUNCH;
    }
    launch("bin_emit", k_bin, dim3((P.N + 255) / 256), dim3(256), st, P, 1);
  }
  // crowded tiles are rare: for them a small grid walks the tile list instead of T mostly idle workgroups
  // (crowded tiles get 1024 threads: 16 waves on the 4096-key network - 170 -> ~35 us when every
  // tile of a 320x240 view holds ~2500 splats)
  if (P.pack) {
    launch("tile_sort", k_tile_sort_reg<true>, dim3(P.T), dim3(256), st, P);
    if (P.big_pass) launch("tile_sort_big", k_tile_sort<4096, 1024, true, 1024>, dim3(min(P.T, 512)), dim3(1024), st, P);
  } else {
    launch("tile_sort", k_tile_sort_reg<false>, dim3(P.T), dim3(256), st, P);
    if (P.big_pass) launch("tile_sort_big", k_tile_sort<4096, 1024, false, 1024>, dim3(min(P.T, 512)), dim3(1024), st, P);
  }
#ifdef MGS_FWD_BLOCKS
  constexpr int kFwdGridChunk = 64;
#else
  constexpr int kFwdGridChunk = kFwdChunk;
#endif
  if (P.obj.on) launch("blend_fwd", k_blend_fwd<true>, dim3(grid_pad(4 * P.T, kFwdGridChunk)), dim3(64), st, P);
  else launch("blend_fwd", k_blend_fwd<false>, dim3(grid_pad(4 * P.T, kFwdGridChunk)), dim3(64), st, P);
  return check_launch();
}

}  // namespace mgs
