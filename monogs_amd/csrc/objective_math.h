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

// The p-norm objective (sum |h|^p)^(1/p) of utils/slam_frontend.py:596-600 (p = 2 with Huber, RGN.pnorm
// without).  Per sample: phi = |h|^p (what is summed) and gam = |h|^(p-1) sign(h); with S = sum phi the value is
// loss = S^(1/p) and d loss / d h = loss^(1-p) gam - the scalar factor is applied by the consumer of the sums
// (norm_finish), everything in between being linear in it.  p = 2: (h^2, h), the form of rounds 1-3.
// p = 1: (|h|, sign h), sign(0) = 0 as torch.norm's derivative has it.
__device__ __forceinline__ float norm_p(float pnorm) { return pnorm > 0.f ? pnorm : 2.f; }

__device__ __forceinline__ void norm_terms(float h, float p, float& phi, float& gam) {
  if (p == 2.f) { phi = h * h; gam = h; return; }
  const float ah = fabsf(h);
  if (p == 1.f) { phi = ah; gam = h > 0.f ? 1.f : (h < 0.f ? -1.f : 0.f); return; }
  const float q = ah > 0.f ? powf(ah, p - 1.f) : 0.f;
  phi = q * ah; gam = copysignf(q, h);
}

// loss = S^(1/p) and scale = loss^(1-p) (0 when the loss is 0: torch.norm's derivative at the origin)
__device__ __forceinline__ void norm_finish(float S, float p, float& loss, float& scale) {
  if (p == 2.f) { loss = sqrtf(S); scale = loss > 0.f ? 1.f / loss : 0.f; return; }
  if (p == 1.f) { loss = S; scale = 1.f; return; }
  loss = S > 0.f ? powf(S, 1.f / p) : 0.f;
  scale = loss > 0.f ? powf(loss, 1.f - p) : 0.f;
}

}  // namespace mgs
