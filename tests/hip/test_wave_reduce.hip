// Unit test of csrc/wave_reduce.h on the GPU: every lane contributes distinct integers
// (exact in fp32), so any wrong lane mapping shows up as a wrong total or a wrong slot.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../monogs_amd/csrc/wave_reduce.h"

__global__ void k(float* main_out, float* extra_out) {
  const int lane = threadIdx.x;
  float r[10];
  for (int i = 0; i < 10; i++) r[i] = (float)((lane + 1) * (i + 1) + (lane % 7) * i);
  const unsigned long long m = __ballot((lane & 8) != 0);
  float a, b;
  mgs::wave_sum10_scatter(r, m, a, b);
  main_out[lane] = a;
  extra_out[lane] = b;
}

__global__ void k6(float* main_out, float* extra_out) {
  const int lane = threadIdx.x;
  float r[6];
  for (int i = 0; i < 6; i++) r[i] = (float)((lane + 1) * (i + 1) + (lane % 7) * i);
  float a, b;
  mgs::wave_sum6_scatter(r, a, b);
  main_out[lane] = a;
  extra_out[lane] = b;
}

static int check6() {
  float *dm, *de;
  hipMalloc(&dm, 64 * 4); hipMalloc(&de, 64 * 4);
  hipLaunchKernelGGL(k6, dim3(1), dim3(64), 0, 0, dm, de);
  std::vector<float> hm(64), he(64);
  hipMemcpy(hm.data(), dm, 256, hipMemcpyDeviceToHost);
  hipMemcpy(he.data(), de, 256, hipMemcpyDeviceToHost);
  double want[6];
  for (int i = 0; i < 6; i++) {
    want[i] = 0;
    for (int l = 0; l < 64; l++) want[i] += (l + 1) * (i + 1) + (l % 7) * i;
  }
  int bad = 0;
  for (int l = 0; l < 64; l++) {      // every lane of a row holds the row's value
    const int idx = ((l >> 5) & 1) + 2 * ((l >> 4) & 1);
    if (hm[l] != (float)want[idx]) { printf("sum6 main lane %d idx %d got %f want %f\n", l, idx, hm[l], want[idx]); bad++; }
  }
  if (he[31] != (float)want[4]) { printf("sum6 extra lane 31 got %f want %f\n", he[31], want[4]); bad++; }
  if (he[63] != (float)want[5]) { printf("sum6 extra lane 63 got %f want %f\n", he[63], want[5]); bad++; }
  return bad;
}

int main() {
  float *dm, *de;
  hipMalloc(&dm, 64 * 4); hipMalloc(&de, 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dm, de);
  std::vector<float> hm(64), he(64);
  hipMemcpy(hm.data(), dm, 256, hipMemcpyDeviceToHost);
  hipMemcpy(he.data(), de, 256, hipMemcpyDeviceToHost);
  double want[10];
  for (int i = 0; i < 10; i++) {
    want[i] = 0;
    for (int l = 0; l < 64; l++) want[i] += (l + 1) * (i + 1) + (l % 7) * i;
  }
  int bad = 0;
  for (int l = 0; l < 64; l += 8) {
    const int idx = ((l >> 5) & 1) + 2 * ((l >> 4) & 1) + 4 * ((l >> 3) & 1);
    if (hm[l] != (float)want[idx]) { printf("main lane %d idx %d got %f want %f\n", l, idx, hm[l], want[idx]); bad++; }
  }
  if (he[31] != (float)want[8]) { printf("extra lane 31 got %f want %f\n", he[31], want[8]); bad++; }
  if (he[63] != (float)want[9]) { printf("extra lane 63 got %f want %f\n", he[63], want[9]); bad++; }
  if (bad) {
    printf("main :"); for (int l = 0; l < 64; l++) printf(" %g", hm[l]); printf("\n");
    printf("extra:"); for (int l = 0; l < 64; l++) printf(" %g", he[l]); printf("\n");
    printf("want :"); for (int i = 0; i < 10; i++) printf(" %g", want[i]); printf("\n");
  }
  bad += check6();
  printf(bad ? "FAIL\n" : "PASS\n");
  return bad ? 1 : 0;
}
