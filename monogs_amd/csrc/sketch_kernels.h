// Pieces of the sketched second-order tracking iteration shared by tracking.hip (the stand-alone entry points) and
// raster_backward.hip (the launch that runs the residual pass BESIDE the per-splat Jacobian preparation): the keyed
// partition of the pixels into buckets (slam_frontend.py:269-338) and the residual pass itself
// (utils/slam_utils.py:188-205 + :58-75, slam_frontend.py:636-650).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/monogs_raster.h"
#include "objective_math.h"

namespace mgs {

constexpr int kSketchThreads = 256;

// ---------------------------------------------------------------------------------
// Keyed pseudo-random permutation of [0, m): invertible rounds (add, odd multiply, xor-shift)
// on `bits` = ceil(log2 m) bits, cycle-walked into range.  Every round is a bijection of
// [0, 2^bits), so the composition is one, and walking a point of [0, m) along its cycle until
// it lands in [0, m) again yields a bijection of [0, m).
__device__ __forceinline__ unsigned int perm_round(unsigned int x, unsigned int mask, int bits,
                                                   unsigned int k0, unsigned int k1) {
  const int h = bits > 2 ? bits / 2 : 1;
  x = (x + k0) & mask;
  x = (x * 0x9E3779B1u) & mask;
  x ^= x >> h;
  x = (x * 0x85EBCA6Bu) & mask;
  x = (x + k1) & mask;
  x ^= x >> (h > 1 ? h - 1 : 1);
  x = (x * 0xC2B2AE35u) & mask;
  x ^= x >> h;
  return x;
}

__device__ __forceinline__ unsigned int hash32(unsigned int x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

constexpr int kSketchBlocks = 256;   // one workgroup per CU (12 KB of LDS bucket sums each)

struct SketchKeys { int on, chunk, bits; unsigned int k0, k1, k2; };   // on != 0: assign bucket / weight here

// One workgroup's share of the sketched residual pass (block `bid` of `nblocks`, kSketchThreads threads); s_acc =
// 3 * stack * sketch floats of LDS ([d][3]: Sf, d/da, d/db), s_red = kSketchThreads / 64 floats.
__device__ __forceinline__ void sketch_residual_block(const mgs_sketch_residual_args& A, const SketchKeys& K, float* s_acc,
                                                      float* s_red, int bid, int nblocks) {
  const int d = A.stack_dim * A.sketch_dim;
  for (int i = threadIdx.x; i < 3 * d; i += kSketchThreads) s_acc[i] = 0.f;
  __syncthreads();
  // The sketched path goes through the reference's hand-written ApplyExposure.backward (utils/slam_utils.py:145-149),
  // which is NOT the exact derivative of (|a| + eps) image + b: grad_image = |a| grad (no eps) and
  // grad_a = sum(grad image) (no sign(a)).  Followed to the letter here (pinned by tests/golden/map_update_ref.npz:
  // ae_*); the first-order path differentiates the same expression by autograd and keeps sign(a) and eps
  // (k_track_loss_bwd / _onepass, the forward blend's objective epilogue).
  const float gain_bwd = fabsf(A.exposure_a[0]);
  const float gain = gain_bwd + A.exposure_eps, bias = A.exposure_b[0];
  const size_t HW = (size_t)A.num_pixels;
  const float scale = (float)d / (float)A.num_pixels;      // 1 / (m / (stack * sketch))
  float l1 = 0.f;
  for (size_t p = (size_t)bid * kSketchThreads + threadIdx.x; p < HW; p += (size_t)nblocks * kSketchThreads) {
    const float om = A.opacity[p] * (A.mask ? A.mask[p] : 1.f);
    int b;
    float wsign;
    if (K.on) {    // the partition of mgs_sketch_assign, evaluated (and left behind for the backward) in this pass
      const unsigned int pmask = K.bits >= 32 ? 0xFFFFFFFFu : ((1u << K.bits) - 1u);
      unsigned int x = (unsigned int)p;
      do { x = perm_round(x, pmask, K.bits, K.k0, K.k1); } while ((size_t)x >= HW);
      b = (long long)x < (long long)K.chunk * d ? (int)(x / (unsigned int)K.chunk) : -1;
      wsign = (hash32((unsigned int)p ^ K.k2) & 0x10000u) ? 1.f : -1.f;
      const_cast<int32_t*>(A.bucket)[p] = b;
      const_cast<float*>(A.weights)[p] = wsign;
    } else {
      b = A.bucket[p];
      wsign = A.weights[p];
    }
    const float w = wsign * scale;
    float hs = 0.f, da = 0.f, db = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float im = A.image[c * HW + p];
      const float r = om * (gain * im + bias - A.gt[c * HW + p]);
      l1 += fabsf(r);
      float dh;
      hs += huber(r, A.huber_delta, dh);
      const float g = w * dh * om;              // d weighted / d (gain * image + bias)
      A.grad_image[c * HW + p] = g * gain_bwd;
      da += g * im;
      db += g;
    }
    if (b >= 0 && b < d) {
      atomicAdd(&s_acc[3 * b], w * hs);
      atomicAdd(&s_acc[3 * b + 1], da);
      atomicAdd(&s_acc[3 * b + 2], db);
    }
  }
  // block sum of the L1 criterion (kSketchThreads / 64 waves)
  for (int off = 32; off > 0; off >>= 1) l1 += __shfl_down(l1, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = l1;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < kSketchThreads / 64; w++) t += s_red[w];
    atomicAdd(A.l1, t);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < d; i += kSketchThreads) {
    const float f = s_acc[3 * i], x = s_acc[3 * i + 1], y = s_acc[3 * i + 2];
    if (f != 0.f) atomicAdd(&A.Sf[i], f);
    if (x != 0.f) atomicAdd(&A.sj_exposure[2 * i], x);
    if (y != 0.f) atomicAdd(&A.sj_exposure[2 * i + 1], y);
  }
}

// chunk size, index bits and the three 32-bit round keys (splitmix64 of the 64-bit key) of the partition
// launch_backward option (second-order tracking iteration): run this residual pass in the same launch as the
// per-splat Jacobian preparation; skip_prep: the preparation of this forward has already run (later repeats)
struct SketchFuse {
  const mgs_sketch_residual_args* residual;
  SketchKeys keys;
  int skip_prep;
};

inline bool sketch_keys(int64_t num_pixels, int32_t stack_dim, int32_t sketch_dim, uint64_t key, SketchKeys& K) {
  if (num_pixels < 1 || num_pixels > 0x7fffffffLL || stack_dim < 1 || sketch_dim < 1) return false;
  const int d = stack_dim * sketch_dim;
  K.chunk = (int)(num_pixels / d);
  if (K.chunk < 1) return false;
  K.bits = 1;
  while ((1LL << K.bits) < num_pixels) K.bits++;
  uint64_t z = key + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
  K.k0 = (unsigned int)z; K.k1 = (unsigned int)(z >> 32);
  z = (z + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
  K.k2 = (unsigned int)(z >> 16);
  K.on = 1;
  return true;
}


}  // namespace mgs
