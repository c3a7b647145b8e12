// simple-knn replacement: mean squared distance to the 3 nearest other points (exact).
// Replaces simple_knn._C.distCUDA2, called at /root/reference
// gaussian_splatting/scene/gaussian_model.py:185-191 with P = 2.4k..25k points.
//
// P is small, so the exact answer comes from a tiled brute force, parallelised over
// (query block) x (candidate slice) so that even P = 2400 fills the chip.  Round 5 rewrite (the first form - one
// query per lane, candidates staged through LDS with two barriers per 256, a branchy insert - ran 111 / 540 us at
// P = 4 800 / 25 500: one LDS round trip per candidate at 1.4 waves per SIMD, and a 19-workgroup merge):
//   * a lane owns Q queries (1, 2 or 4 by P: more evaluations per candidate fetch at large P, more workgroups at
//     small P) and keeps their three best squared distances in registers;
//   * the candidate index is wave-uniform, so candidates come through the SCALAR path (s_load from the constant
//     address space, eight candidates = 24 dwords per batch) - no LDS, no barrier, and the loads of the next batch are
//     in flight under the arithmetic of this one;
//   * two queries share packed arithmetic (v_pk_add / v_pk_mul / v_pk_fma_f32), every distance goes through a
//     branch-free sorted insert (five v_min / v_max), and the self test (candidate == query) is only compiled into
//     the batches that overlap the workgroup's own query range;
//   * partial results are stored component-major ([slice][3][P]: coalesced), merged by a second small kernel.
// No sort, no tree, no atomics; deterministic.  Fewer than four points: a missing neighbour counts as FLT_MAX
// (what the absent extension's FLT_MAX-initialised best[3] yields [UPSTREAM-KNOWLEDGE]; oracle: dist2_knn3).
#include <hip/hip_runtime.h>

#include "../../include/monogs_raster.h"
#include "launch.h"

namespace mgs {

constexpr int kKnnBlock = 256;
constexpr int kKnnMaxSlices = 64;
constexpr int kKnnBatch = 8;         // candidates per scalar-load batch
constexpr float kKnnInf = 3.4028235e38f;

// Branch-free sorted insert.  (fminf / fmaxf make the compiler canonicalise an operand with a v_max_f32 v, v, v first;
// raw v_min_f32 / v_max_f32 through inline asm - five instructions per insert instead of up to eight - measured 1-5 %
// SLOWER back to back on one box, 22.1 / 48.9 / 63.8 / 205 us against 21.8 / 48.0 / 61.2 / 195 us: not taken.)
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }

__device__ __forceinline__ void best3_insert(float d, float& b0, float& b1, float& b2) {
  const float t0 = vmax(b0, d);
  b0 = vmin(b0, d);
  const float t1 = vmax(b1, t0);
  b1 = vmin(b1, t0);
  b2 = vmin(b2, t1);
}

struct KnnPlan { int q, qblocks, slices, slice_len; };

static KnnPlan knn_plan(int n) {
  KnnPlan p;
  p.q = n >= 20000 ? 4 : (n >= 11000 ? 2 : 1);
  p.qblocks = (n + kKnnBlock * p.q - 1) / (kKnnBlock * p.q);
  // As many workgroups as are resident AT ONCE (256 CUs x waves per SIMD: 78 VGPRs -> 6 at Q = 4, else 8), not more: the
  // workgroups all take the same time, so a second, nearly empty round would double the kernel's duration (it did:
  // 2 432 workgroups at P = 9 600 and 1 600 at 25 500 against 2 048 / 1 536 slots).  A slice is a whole number of batches.
  const int resident = 256 * (p.q == 4 ? 6 : 8);
  int s = resident / p.qblocks;
  if (s > kKnnMaxSlices) s = kKnnMaxSlices;
  const int max_s = (n + 4 * kKnnBatch - 1) / (4 * kKnnBatch);      // at least four batches per slice
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  p.slice_len = ((n + s - 1) / s + kKnnBatch - 1) / kKnnBatch * kKnnBatch;
  p.slices = (n + p.slice_len - 1) / p.slice_len;
  return p;
}

typedef const __attribute__((address_space(4))) float* cfloat_p;    // constant address space: uniform loads go scalar
typedef float v2f __attribute__((ext_vector_type(2)));

// Q (even, or 1) queries per lane.  The squared distance of a PAIR of queries to one candidate is six packed
// instructions (v_pk_add / v_pk_mul / v_pk_fma_f32 with the candidate's coordinates as scalar operands: two
// evaluations for ~1.1x the issue time of one); every evaluation goes straight into the branch-free sorted insert.
// (A first version kept a batch's distances and skipped the inserts when no lane improved: with 64 Q queries per wave
// some lane nearly always does, and the 32 kept distances cost the kernel its occupancy - 168 VGPRs, 227 us at 25 500.)
template <int Q>
__global__ __launch_bounds__(kKnnBlock) void k_knn_partial(const float* __restrict__ pts, int n, int slice_len,
                                                           float* __restrict__ part) {
  constexpr int QP = (Q + 1) / 2;                                    // query pairs (Q = 1: the second half idles)
  const int tid = threadIdx.x;
  const int q_first = blockIdx.x * (kKnnBlock * Q);                  // this workgroup's queries: [q_first, q_first + 256 Q)
  const int c0 = blockIdx.y * slice_len, c1 = min(n, c0 + slice_len);
  v2f qx[QP], qy[QP], qz[QP];
  float b0[2 * QP], b1[2 * QP], b2[2 * QP];
  int qi[2 * QP];
#pragma unroll
  for (int k = 0; k < 2 * QP; k++) {
    qi[k] = k < Q ? q_first + k * kKnnBlock + tid : 0x7fffffff;
    const int qc = min(qi[k], n - 1);
    const float x = pts[3 * qc], y = pts[3 * qc + 1], z = pts[3 * qc + 2];
    if (k & 1) { qx[k >> 1].y = x; qy[k >> 1].y = y; qz[k >> 1].y = z; }
    else { qx[k >> 1].x = x; qy[k >> 1].x = y; qz[k >> 1].x = z; }
    b0[k] = b1[k] = b2[k] = kKnnInf;
  }
  cfloat_p cp = (cfloat_p)pts;
  // one candidate (index jc, coordinates wave-uniform); SELF: it may be one of this workgroup's queries
  auto candidate = [&](int jc, float cx, float cy, float cz, auto self_tag) {
    constexpr bool SELF = decltype(self_tag)::value;
#pragma unroll
    for (int p = 0; p < QP; p++) {
      const v2f dx = v2f{cx, cx} - qx[p], dy = v2f{cy, cy} - qy[p], dz = v2f{cz, cz} - qz[p];
      v2f d = dx * dx;
      d = __builtin_elementwise_fma(dy, dy, d);
      d = __builtin_elementwise_fma(dz, dz, d);
      float d0 = d.x, d1 = d.y;
      if (SELF) {
        if (jc == qi[2 * p]) d0 = kKnnInf;
        if (jc == qi[2 * p + 1]) d1 = kKnnInf;
      }
      best3_insert(d0, b0[2 * p], b1[2 * p], b2[2 * p]);
      if (2 * p + 1 < Q) best3_insert(d1, b0[2 * p + 1], b1[2 * p + 1], b2[2 * p + 1]);
    }
  };
  auto batch = [&](int j, auto self_tag) {
    float c[3 * kKnnBatch];
#pragma unroll
    for (int i = 0; i < 3 * kKnnBatch; i++) c[i] = cp[3 * (size_t)j + i];        // wave-uniform: s_load_dwordx8 x 3
#pragma unroll
    for (int i = 0; i < kKnnBatch; i++) candidate(j + i, c[3 * i], c[3 * i + 1], c[3 * i + 2], self_tag);
  };
  const int full_end = c0 + max(c1 - c0, 0) / kKnnBatch * kKnnBatch;
  for (int j = c0; j < full_end; j += kKnnBatch) {
    if (j + kKnnBatch > q_first && j < q_first + kKnnBlock * Q) batch(j, std::true_type{});     // wave-uniform
    else batch(j, std::false_type{});
  }
  for (int j = full_end; j < c1; j++)         // tail of the last slice: fewer than a batch
    candidate(j, cp[3 * (size_t)j], cp[3 * (size_t)j + 1], cp[3 * (size_t)j + 2], std::true_type{});
#pragma unroll
  for (int k = 0; k < Q; k++)
    if (qi[k] < n) {
      float* o = part + (size_t)blockIdx.y * 3 * n + qi[k];       // [slice][3][n]
      o[0] = b0[k]; o[n] = b1[k]; o[2 * (size_t)n] = b2[k];
    }
}

__global__ __launch_bounds__(kKnnBlock) void k_knn_merge(const float* __restrict__ part, int n,
                                                         int slices, float* __restrict__ out) {
  const int q = blockIdx.x * kKnnBlock + threadIdx.x;
  if (q >= n) return;
  float b0 = kKnnInf, b1 = kKnnInf, b2 = kKnnInf;
  // The loads of four slices are in flight together (the loop MUST be unrolled: rolled, it is one memory round trip
  // per slice - 60 of them at P = 4 800 - and the merge, not the distance pass, sets the call's time: 34 instead
  // of 22 us; a tail loop takes the remainder)
  constexpr int U = 4;
  int s = 0;
  for (; s + U <= slices; s += U) {
    float v[3 * U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float* p = part + (size_t)(s + u) * 3 * n + q;
      v[3 * u] = p[0]; v[3 * u + 1] = p[n]; v[3 * u + 2] = p[2 * (size_t)n];
    }
#pragma unroll
    for (int i = 0; i < 3 * U; i++) {
      const float t0 = fmaxf(b0, v[i]);
      b0 = fminf(b0, v[i]);
      const float t1 = fmaxf(b1, t0);
      b1 = fminf(b1, t0);
      b2 = fminf(b2, t1);
    }
  }
  for (; s < slices; s++) {
    const float* p = part + (size_t)s * 3 * n + q;
    const float x = p[0], y = p[n], z = p[2 * (size_t)n];
    const float vv[3] = {x, y, z};
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const float t0 = fmaxf(b0, vv[i]);
      b0 = fminf(b0, vv[i]);
      const float t1 = fmaxf(b1, t0);
      b1 = fminf(b1, t0);
      b2 = fminf(b2, t1);
    }
  }
  out[q] = (b0 + b1 + b2) / 3.0f;
}

uint64_t knn_scratch_bytes(int n) { return (uint64_t)kKnnMaxSlices * (uint64_t)n * 3 * sizeof(float) + 256; }

int launch_knn(const float* pts, int n, float* out, void* scratch, hipStream_t st) {
  const KnnPlan p = knn_plan(n);
  float* part = (float*)scratch;
  const dim3 grid(p.qblocks, p.slices), block(kKnnBlock);
  if (p.q == 4) launch("knn_partial", k_knn_partial<4>, grid, block, st, pts, n, p.slice_len, part);
  else if (p.q == 2) launch("knn_partial", k_knn_partial<2>, grid, block, st, pts, n, p.slice_len, part);
  else launch("knn_partial", k_knn_partial<1>, grid, block, st, pts, n, p.slice_len, part);
  launch("knn_merge", k_knn_merge, dim3((n + kKnnBlock - 1) / kKnnBlock), block, st, (const float*)part, n, p.slices, out);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

}  // namespace mgs
