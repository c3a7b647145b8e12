// simple-knn replacement: mean squared distance to the 3 nearest other points (exact).
// Replaces simple_knn._C.distCUDA2, called at /root/reference
// gaussian_splatting/scene/gaussian_model.py:185-191 with P = 2.4k..25k points.
//
// P is small, so the exact answer comes from a tiled brute force, parallelised over
// (query block) x (candidate slice) so that even P = 2400 fills the chip: each workgroup
// owns 256 query points (one per lane) and one slice of the candidates, candidates are
// staged through LDS (coalesced loads, LDS broadcast reads), every lane keeps its three best
// squared distances in registers (the insert is skipped wave-uniformly when no lane
// improves), and a second small kernel merges the slices.  No sort, no tree, no atomics;
// deterministic.
#include <hip/hip_runtime.h>

#include "../../include/monogs_raster.h"
#include "launch.h"

namespace mgs {

constexpr int kKnnBlock = 256;
constexpr int kKnnMaxSlices = 64;

__device__ __forceinline__ void best3_insert(float d, float& b0, float& b1, float& b2) {
  if (d < b2) {
    if (d < b1) {
      b2 = b1;
      if (d < b0) { b1 = b0; b0 = d; } else { b1 = d; }
    } else {
      b2 = d;
    }
  }
}

static int knn_slices(int n) {
  const int qb = (n + kKnnBlock - 1) / kKnnBlock;
  int s = 1024 / qb;
  if (s < 1) s = 1;
  if (s > kKnnMaxSlices) s = kKnnMaxSlices;
  const int cb = qb;                       // candidate blocks of 256
  if (s > cb) s = cb;
  return s;
}

__global__ __launch_bounds__(kKnnBlock) void k_knn_partial(const float* __restrict__ pts, int n,
                                                           int slice_len, float* __restrict__ part) {
  __shared__ float s_x[kKnnBlock], s_y[kKnnBlock], s_z[kKnnBlock];
  const int tid = threadIdx.x;
  const int q = blockIdx.x * kKnnBlock + tid;
  const int c0 = blockIdx.y * slice_len, c1 = min(n, c0 + slice_len);
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (q < n) { qx = pts[3 * q]; qy = pts[3 * q + 1]; qz = pts[3 * q + 2]; }
  const float inf = 3.4028235e38f;
  float b0 = inf, b1 = inf, b2 = inf;
  for (int base = c0; base < c1; base += kKnnBlock) {
    __syncthreads();
    const int c = base + tid;
    if (c < c1) { s_x[tid] = pts[3 * c]; s_y[tid] = pts[3 * c + 1]; s_z[tid] = pts[3 * c + 2]; }
    __syncthreads();
    const int nb = min(kKnnBlock, c1 - base);
    for (int j = 0; j < nb; j++) {
      const float dx = s_x[j] - qx, dy = s_y[j] - qy, dz = s_z[j] - qz;
      const float d = dx * dx + dy * dy + dz * dz;
      const bool better = d < b2 && (base + j) != q;
      if (__builtin_amdgcn_ballot_w64(better) != 0ull) {
        if (better) best3_insert(d, b0, b1, b2);
      }
    }
  }
  if (q < n) {
    float* o = part + ((size_t)blockIdx.y * n + q) * 3;
    o[0] = b0; o[1] = b1; o[2] = b2;
  }
}

__global__ __launch_bounds__(kKnnBlock) void k_knn_merge(const float* __restrict__ part, int n,
                                                         int slices, float* __restrict__ out) {
  const int q = blockIdx.x * kKnnBlock + threadIdx.x;
  if (q >= n) return;
  const float inf = 3.4028235e38f;
  float b0 = inf, b1 = inf, b2 = inf;
  for (int s = 0; s < slices; s++) {
    const float* p = part + ((size_t)s * n + q) * 3;
    best3_insert(p[0], b0, b1, b2);
    best3_insert(p[1], b0, b1, b2);
    best3_insert(p[2], b0, b1, b2);
  }
  out[q] = (b0 + b1 + b2) / 3.0f;
}

uint64_t knn_scratch_bytes(int n) { return (uint64_t)knn_slices(n) * (uint64_t)n * 3 * sizeof(float) + 256; }

int launch_knn(const float* pts, int n, float* out, void* scratch, hipStream_t st) {
  const int slices = knn_slices(n);
  const int qb = (n + kKnnBlock - 1) / kKnnBlock;
  const int slice_len = ((n + slices - 1) / slices + kKnnBlock - 1) / kKnnBlock * kKnnBlock;
  float* part = (float*)scratch;
  launch("knn_partial", k_knn_partial, dim3(qb, slices), dim3(kKnnBlock), st, pts, n, slice_len, part);
  launch("knn_merge", k_knn_merge, dim3(qb), dim3(kKnnBlock), st, (const float*)part, n, slices, out);
  return launches_ok() ? MGS_OK : MGS_ERR_LAUNCH;
}

}  // namespace mgs
