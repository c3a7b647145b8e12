// simple-knn replacement: mean squared distance to the 3 nearest other points (exact).
// Replaces simple_knn._C.distCUDA2, called at /root/reference
// gaussian_splatting/scene/gaussian_model.py:185-191 with P = 2.4k..25k points.
//
// P is small, so the exact answer comes from a tiled brute force: each workgroup owns
// 256 query points (one per lane) and a slice of the candidate range, candidates are
// staged through LDS (coalesced float loads, LDS broadcast reads), every lane keeps its
// three best squared distances in registers, and the slices are merged by a second tiny
// kernel.  No sort, no tree, no atomics; deterministic.
#include <hip/hip_runtime.h>

#include "../../include/monogs_raster.h"
#include "launch.h"

namespace mgs {

constexpr int kKnnBlock = 256;

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

__global__ __launch_bounds__(kKnnBlock) void k_knn(const float* __restrict__ pts, int n,
                                                   float* __restrict__ out) {
  __shared__ float s_x[kKnnBlock], s_y[kKnnBlock], s_z[kKnnBlock];
  const int tid = threadIdx.x;
  const int q = blockIdx.x * kKnnBlock + tid;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (q < n) { qx = pts[3 * q]; qy = pts[3 * q + 1]; qz = pts[3 * q + 2]; }
  float b0 = 3.4028235e38f, b1 = 3.4028235e38f, b2 = 3.4028235e38f;
  for (int base = 0; base < n; base += kKnnBlock) {
    __syncthreads();
    const int c = base + tid;
    if (c < n) { s_x[tid] = pts[3 * c]; s_y[tid] = pts[3 * c + 1]; s_z[tid] = pts[3 * c + 2]; }
    __syncthreads();
    const int nb = min(kKnnBlock, n - base);
    for (int j = 0; j < nb; j++) {
      const float dx = s_x[j] - qx, dy = s_y[j] - qy, dz = s_z[j] - qz;
      const float d = dx * dx + dy * dy + dz * dz;
      if (base + j != q) best3_insert(d, b0, b1, b2);
    }
  }
  if (q < n) out[q] = (b0 + b1 + b2) / 3.0f;
}

int launch_knn(const float* pts, int n, float* out, hipStream_t st) {
  launch("knn", k_knn, dim3((n + kKnnBlock - 1) / kKnnBlock), dim3(kKnnBlock), st, pts, n, out);
  return hipGetLastError() == hipSuccess ? MGS_OK : MGS_ERR_LAUNCH;
}

}  // namespace mgs
