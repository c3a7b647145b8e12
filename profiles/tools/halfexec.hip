// Does a wave64 VALU instruction with an empty EXEC half issue faster on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  const int lane = threadIdx.x & 63;
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  bool on = true;
  if (MODE == 1) on = lane < 32;            // upper half empty
  if (MODE == 2) on = (lane & 1) == 0;      // every other lane
  if (MODE == 3) on = lane < 16;            // one row of 16
  if (MODE == 4) on = lane >= 32;           // lower half empty
  if (on) {
    for (int i = 0; i < iters; i++) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        a0 = __builtin_fmaf(a0, s, 1.0f); a1 = __builtin_fmaf(a1, s, 1.0f); a2 = __builtin_fmaf(a2, s, 1.0f); a3 = __builtin_fmaf(a3, s, 1.0f);
        a4 = __builtin_fmaf(a4, s, 1.0f); a5 = __builtin_fmaf(a5, s, 1.0f); a6 = __builtin_fmaf(a6, s, 1.0f); a7 = __builtin_fmaf(a7, s, 1.0f);
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE>
float run(float* d, int blocks, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, iters, 0.999f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 0.999f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  const int blocks = 256 * 8, iters = 2000;   // 8 workgroups of 4 waves per CU: 8 waves / SIMD
  float* d; hipMalloc(&d, sizeof(float) * blocks * 256);
  const double inst = (double)blocks * 4 * iters * 64;    // wave-instructions
  const char* names[] = {"all 64 lanes", "lanes 0-31", "even lanes", "lanes 0-15", "lanes 32-63"};
  float ms[5] = {run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters), run<3>(d, blocks, iters), run<4>(d, blocks, iters)};
  for (int m = 0; m < 5; m++)
    printf("%-14s %.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[m], ms[m], ms[m] * 1e-3 * 2.4e9 / (inst / 1024));
  return 0;
}
