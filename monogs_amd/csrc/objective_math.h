// Per-sample arithmetic of the tracking objective, shared by tracking.hip's loss kernels and the objective
// epilogue of k_blend_fwd (raster_forward.hip): both must form the same residual bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace mgs {

// Pseudo-Huber of the reference's tracking objective (utils/slam_utils.py:58-75): h(x) = x for |x| < delta,
// else sign(x) sqrt(2 delta |x| - delta^2); dh = dh/dx.  delta <= 0 switches it off.
__device__ __forceinline__ float huber(float x, float delta, float& dh) {
  const float ax = fabsf(x);
  if (delta <= 0.f || ax < delta) { dh = 1.f; return x; }
  const float s = sqrtf(2.f * delta * ax - delta * delta);
  dh = delta / s;
  return copysignf(s, x);
}

}  // namespace mgs
