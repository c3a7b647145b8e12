// Forward pass of the rasteriser for gfx950 (CDNA4, wave64).
//
// Pipeline (one launch each, all on the caller's stream, no host sync):
//   k_preprocess   1 thread / Gaussian : projection + EWA -> 48-B SplatRec      (HBM bound)
//   k_bin<count>   1 thread / Gaussian : exact tile culling, per-tile counts    (HBM + int atomics)
//   k_tile_scan    1 workgroup         : exclusive scan of T tile counts -> D
//   k_bin<emit>    1 thread / Gaussian : (depth|id, j) pairs into per-tile segments
//   k_tile_sort    1 workgroup / tile  : LDS bitonic sort of the tile's segment by (depth,id)
//   k_blend_fwd    1 workgroup / tile  : LDS-staged front-to-back alpha blend
//
// Replaces rasterize_gaussians (forward) of the reference's CUDA extension, called at
// /root/reference gaussian_splatting/gaussian_renderer/__init__.py:151-168.
#include "launch.h"
#include "raster_kernels.h"

namespace mgs {

__device__ __forceinline__ void load_camera(Camera& c, const KP& P) {
#pragma unroll
  for (int i = 0; i < 16; i++) { c.V[i] = P.V[i]; c.PM[i] = P.PM[i]; c.Praw[i] = P.Praw[i]; }
  c.campos[0] = P.campos[0]; c.campos[1] = P.campos[1]; c.campos[2] = P.campos[2];
  c.W = P.W; c.H = P.H; c.tanfovx = P.tanfovx; c.tanfovy = P.tanfovy;
  c.focal_x = P.focal_x; c.focal_y = P.focal_y; c.scale_modifier = P.mod;
  c.sh_degree = P.deg; c.sh_coeffs = P.K; c.grid_x = P.grid_x; c.grid_y = P.grid_y;
}

// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kPreBlock) void k_preprocess(KP P) {
  const int idx = blockIdx.x * kPreBlock + threadIdx.x;
  for (int i = idx; i < P.T; i += gridDim.x * kPreBlock) P.tile_count[i] = 0;
  if (idx < 4) P.counters[idx] = 0;
  if (idx >= P.N) return;
  Camera cam;
  load_camera(cam, P);
  const float p[3] = {P.means[3 * idx], P.means[3 * idx + 1], P.means[3 * idx + 2]};
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
  float col[3];
  const float* pcol = nullptr;
  if (P.precol) {
    col[0] = P.precol[3 * idx]; col[1] = P.precol[3 * idx + 1]; col[2] = P.precol[3 * idx + 2];
    pcol = col;
  }
  const float* psh = P.shs ? P.shs + (size_t)3 * P.K * idx : nullptr;
  SplatRec rec;
  project_gaussian(cam, p, psc, pq, pc6, psh, pcol, P.opac[idx], rec);
  float4* dst = reinterpret_cast<float4*>(P.rec + idx);
  dst[0] = make_float4(rec.x, rec.y, rec.depth, rec.opacity);
  dst[1] = make_float4(rec.ca, rec.cb, rec.cc, __int_as_float(rec.radius));
  dst[2] = make_float4(rec.r, rec.g, rec.b, __uint_as_float(rec.flags));
  P.radii[idx] = rec.radius;
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
__global__ __launch_bounds__(kBinThreads) void k_bin_lds(KP P, int emit, int per_block) {
  extern __shared__ int s_tile[];
  const int tid = threadIdx.x;
  int* row = P.bin_table + (size_t)blockIdx.x * P.T;
  for (int t = tid; t < P.T; t += kBinThreads) s_tile[t] = emit ? (P.tile_offset[t] + row[t]) : 0;
  __syncthreads();
  const int g0 = blockIdx.x * per_block, g1 = min(P.N, g0 + per_block);
  for (int idx = g0 + tid; idx < g1; idx += kBinThreads) {
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
          const int pos = atomicAdd(&s_tile[t], 1);
          if (emit && pos < P.cap) {
            P.keys[pos] = keyhi | (unsigned int)idx;
            P.payload[pos] = (unsigned int)cnt;
          }
          cnt++;
        }
      }
    }
    if (!emit) P.pair_count[idx] = cnt;
  }
  if (!emit) {
    __syncthreads();
    for (int t = tid; t < P.T; t += kBinThreads) row[t] = s_tile[t];
  }
}

// One thread per tile: exclusive prefix down the kBinBlocks rows, total -> tile_count.
__global__ __launch_bounds__(64) void k_bin_colsum(KP P, int nblk) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= P.T) return;
  int run = 0;
  for (int b = 0; b < nblk; b++) {
    const int v = P.bin_table[(size_t)b * P.T + t];
    P.bin_table[(size_t)b * P.T + t] = run;
    run += v;
  }
  P.tile_count[t] = run;
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
// Exclusive scan of the T tile counts by one 1024-thread workgroup.
__global__ __launch_bounds__(1024) void k_tile_scan(KP P) {
  __shared__ int s_sum[1024];
  const int tid = threadIdx.x;
  const int per = (P.T + 1023) / 1024;
  const int lo = tid * per, hi = min(lo + per, P.T);
  int local = 0;
  for (int i = lo; i < hi; i++) local += P.tile_count[i];
  s_sum[tid] = local;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (tid >= off) ? s_sum[tid - off] : 0;
    __syncthreads();
    s_sum[tid] += v;
    __syncthreads();
  }
  int run = s_sum[tid] - local;
  for (int i = lo; i < hi; i++) {
    P.tile_offset[i] = run;
    run += P.tile_count[i];
  }
  if (tid == 1023) {
    P.tile_offset[P.T] = s_sum[1023];
    P.counters[0] = s_sum[1023];
  }
}

// ---------------------------------------------------------------------------------
// Bitonic network for arbitrary n with virtual +inf padding: every comparator puts the
// smaller key at the lower index, comparators whose upper index is >= n are no-ops.
// Keys are unique (depth bits | Gaussian id), so the result is a deterministic total
// order whatever order the emit pass filled the segment in.
template <typename KP_, typename VP_>
__device__ __forceinline__ void bitonic_sort(KP_ key, VP_ val, int n, int tid, int nthr) {
  int m = 1;
  while (m < n) m <<= 1;
  const int half_m = m >> 1;
  for (int k = 2; k <= m; k <<= 1) {
    {
      const int half = k >> 1;
      for (int i = tid; i < half_m; i += nthr) {
        const int blk = i / half, off = i - blk * half;
        const int lo = blk * k + off, hi = blk * k + k - 1 - off;
        if (hi < n) {
          const unsigned long long a = key[lo], b = key[hi];
          if (a > b) {
            key[lo] = b; key[hi] = a;
            const unsigned int va = val[lo]; val[lo] = val[hi]; val[hi] = va;
          }
        }
      }
      __syncthreads();
    }
    for (int j = k >> 2; j > 0; j >>= 1) {
      for (int i = tid; i < half_m; i += nthr) {
        const int lo = 2 * i - (i & (j - 1)), hi = lo + j;
        if (hi < n) {
          const unsigned long long a = key[lo], b = key[hi];
          if (a > b) {
            key[lo] = b; key[hi] = a;
            const unsigned int va = val[lo]; val[lo] = val[hi]; val[hi] = va;
          }
        }
      }
      __syncthreads();
    }
  }
}

constexpr int kSortLds = 4096;   // pairs sorted in LDS (48 KB); larger tiles sort in HBM

__global__ __launch_bounds__(256) void k_tile_sort(KP P) {
  __shared__ unsigned long long s_key[kSortLds];
  __shared__ unsigned int s_val[kSortLds];
  const int tile = blockIdx.x, tid = threadIdx.x;
  int start = P.tile_offset[tile], end = P.tile_offset[tile + 1];
  start = min(start, P.cap); end = min(end, P.cap);
  const int n = end - start;
  if (n <= 1) return;
  unsigned long long* gk = P.keys + start;
  unsigned int* gv = P.payload + start;
  if (n <= kSortLds) {
    for (int i = tid; i < n; i += 256) { s_key[i] = gk[i]; s_val[i] = gv[i]; }
    __syncthreads();
    bitonic_sort(s_key, s_val, n, tid, 256);
    for (int i = tid; i < n; i += 256) { gk[i] = s_key[i]; gv[i] = s_val[i]; }
  } else {
    __syncthreads();
    bitonic_sort(gk, gv, n, tid, 256);
  }
}

// ---------------------------------------------------------------------------------
// Front-to-back blend: 256 threads = one 16x16 tile, one pixel per lane; the tile's
// sorted splat list is staged through LDS 256 records at a time (all lanes of a wave
// read the same record -> LDS broadcast).  The staging thread pre-multiplies the conic
// by -0.5*log2(e) so the per-pixel evaluation is 5 FMAs + one v_exp_f32.
constexpr float kLog2e = 1.4426950408889634f;

__global__ __launch_bounds__(256) void k_blend_fwd(KP P) {
  __shared__ float4 s_r0[256], s_r1[256], s_r2[256];
  __shared__ unsigned int s_id[256];
  __shared__ int s_cnt[256];
  const int tile = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int tx = tile % P.grid_x, ty = tile / P.grid_x;
  const int px = tx * kTile + (tid & 15), py = ty * kTile + (tid >> 4);
  const bool inside = px < P.W && py < P.H;
  int start = P.tile_offset[tile], end = P.tile_offset[tile + 1];
  start = min(start, P.cap); end = min(end, P.cap);
  bool done = !inside;
  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, D = 0.f;
  int last = 0;
  const float fpx = (float)px, fpy = (float)py;
  for (int base = start; base < end; base += 256) {
    if (__syncthreads_count(done) == 256) break;
    const int k = base + tid;
    if (k < end) {
      const unsigned int id = (unsigned int)P.keys[k];
      const float4* src = reinterpret_cast<const float4*>(P.rec + id);
      const float4 a = src[0], b = src[1], c = src[2];
      // (x, y, A', B') (C', opacity, depth, r) (g, b, -, -)
      s_r0[tid] = make_float4(a.x, a.y, -0.5f * kLog2e * b.x, -kLog2e * b.y);
      s_r1[tid] = make_float4(-0.5f * kLog2e * b.z, a.w, a.z, c.x);
      s_r2[tid] = make_float4(c.y, c.z, 0.f, 0.f);
      s_id[tid] = id;
    }
    s_cnt[tid] = 0;
    __syncthreads();
    const int nb = min(256, end - base);
    const int cbase = base - start;
    for (int j = 0; j < nb; j++) {
      if (__ballot(!done) == 0ull) break;
      const float4 a = s_r0[j], b = s_r1[j];
      const float2 c = *reinterpret_cast<const float2*>(&s_r2[j]);
      const float dx = a.x - fpx, dy = a.y - fpy;
      const float pw = dx * (a.z * dx + a.w * dy) + b.x * dy * dy;
      const float alpha = fminf(kAlphaMax, b.y * __builtin_amdgcn_exp2f(pw));
      const bool valid = !done && pw <= 0.f && alpha >= kAlphaMin;
      const float test_T = T * (1.f - alpha);
      const bool stop = valid && test_T < kTStop;
      done = done || stop;
      const bool contrib = valid && !stop;
      const float w = contrib ? alpha * T : 0.f;
      C0 += b.w * w; C1 += c.x * w; C2 += c.y * w;
      D += b.z * w;
      T = contrib ? test_T : T;
      last = contrib ? (cbase + j + 1) : last;
      const unsigned long long m = __ballot(contrib && test_T > kTouchT);
      if (m) {
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) atomicAdd(&s_cnt[j], __popcll(m));
      }
    }
    __syncthreads();
    if (tid < nb && s_cnt[tid] > 0) atomicAdd(&P.n_touched[s_id[tid]], s_cnt[tid]);
  }
  if (inside) {
    const size_t pix = (size_t)py * P.W + px, HW = (size_t)P.W * P.H;
    P.final_T[pix] = T;
    P.n_contrib[pix] = last;
    P.out_color[pix] = C0 + T * P.bg[0];
    P.out_color[HW + pix] = C1 + T * P.bg[1];
    P.out_color[2 * HW + pix] = C2 + T * P.bg[2];
    P.out_depth[pix] = D;
    P.out_opacity[pix] = 1.f - T;
  }
}

// ---------------------------------------------------------------------------------
static inline int check_launch() {
  return hipGetLastError() == hipSuccess ? MGS_OK : MGS_ERR_LAUNCH;
}

static inline int bin_blocks(int N) {
  return max(1, min(kBinBlocks, (N + kBinThreads - 1) / kBinThreads));
}

int launch_forward_project(const KP& P, hipStream_t st) {
  const int nblk = (P.N + kPreBlock - 1) / kPreBlock;
  launch("preprocess", k_preprocess, dim3(nblk), dim3(kPreBlock), st, P);
  if (P.T <= kBinMaxTilesLds) {
    const int nblk = bin_blocks(P.N), per = (P.N + nblk - 1) / nblk;
    launch_smem("bin_count", k_bin_lds, dim3(nblk), dim3(kBinThreads), sizeof(int) * (size_t)P.T, st, P, 0, per);
    launch("bin_colsum", k_bin_colsum, dim3((P.T + 63) / 64), dim3(64), st, P, nblk);
  } else {
    launch("bin_count", k_bin, dim3((P.N + 255) / 256), dim3(256), st, P, 0);
  }
  launch("tile_scan", k_tile_scan, dim3(1), dim3(1024), st, P);
  return check_launch();
}

int launch_forward_blend(const KP& P, hipStream_t st) {
  // cursors restart at 0 on every call so a retry with a larger capacity is valid
  if (hipMemsetAsync(P.tile_cursor, 0, sizeof(int) * (size_t)P.T, st) != hipSuccess ||
      hipMemsetAsync(P.n_touched, 0, sizeof(int) * (size_t)P.N, st) != hipSuccess)
    return MGS_ERR_LAUNCH;
  if (P.T <= kBinMaxTilesLds) {
    const int nblk = bin_blocks(P.N), per = (P.N + nblk - 1) / nblk;
    launch_smem("bin_emit", k_bin_lds, dim3(nblk), dim3(kBinThreads), sizeof(int) * (size_t)P.T, st, P, 1, per);
  } else {
    launch("bin_emit", k_bin, dim3((P.N + 255) / 256), dim3(256), st, P, 1);
  }
  launch("tile_sort", k_tile_sort, dim3(P.T), dim3(256), st, P);
  launch("blend_fwd", k_blend_fwd, dim3(P.T), dim3(256), st, P);
  return check_launch();
}

}  // namespace mgs
