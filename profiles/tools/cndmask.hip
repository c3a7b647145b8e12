// cost of compare + select sequences on gfx950: VCC (VOP2 forms) against SGPR pairs (VOP3 forms)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
template <int KIND>
__global__ void k(long long* out, float* sink, int iters, float seed) {
  float a = seed + threadIdx.x, b = seed * 2.f, c = 0.5f, d = 1.5f, e = 2.5f, f = 3.5f;
  asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(a), "v"(b) : "vcc");
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) asm volatile(REP16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %1, vcc\n v_cndmask_b32_e32 %3, %3, %1, vcc\n v_cndmask_b32_e32 %4, %4, %1, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "vcc");
    if (KIND == 1) asm volatile(REP16("v_cmp_le_f32_e32 vcc, %1, %0\n s_nop 1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cmp_le_f32_e32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32_e32 %2, %2, %1, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "vcc");
    if (KIND == 2) asm volatile(REP16("v_cmp_le_f32_e64 s[20:21], %1, %0\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cmp_le_f32_e64 s[22:23], %1, %2\n s_nop 1\n v_cndmask_b32_e64 %2, %2, %1, s[22:23]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s20", "s21", "s22", "s23");
    if (KIND == 3) asm volatile(REP16("v_cmp_le_f32_e64 s[20:21], %1, %0\n v_cmp_le_f32_e64 s[22:23], %1, %2\n s_nop 0\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %1, s[22:23]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "s20", "s21", "s22", "s23");
    if (KIND == 4) asm volatile(REP16("v_cmp_le_f32_e32 vcc, %1, %0\n v_cmp_le_f32_e32 vcc, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "vcc");
    if (KIND == 5) asm volatile(REP16("v_cmp_le_f32_e32 vcc, %1, %0\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) :: "vcc");
    if (KIND == 6) asm volatile(REP16("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %1\n v_max_f32 %3, %3, %1\n v_max_f32 %4, %4, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (a + b + c + d + e + f == 12345.f) sink[0] = a;
}
template <int KIND>
void run(const char* name, int per_iter) {
  long long* out; float* sink;
  (void)hipMalloc(&out, sizeof(long long) * 65536); (void)hipMalloc(&sink, 4);
  const int iters = 2000;
  printf("%-58s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, sink, 10, 1.0f);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, sink, iters, 1.0f);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), out, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v; s /= h.size();
    printf("  wps%d: %6.2f /wave %5.2f /SIMD |", wps, s / ((double)iters * per_iter), s / ((double)iters * per_iter) / wps);
  }
  printf("\n");
}
int main() {
  printf("cycles per UNIT (unit named per row), per wave and per SIMD\n");
  run<0>("v_cndmask_e32 (vcc set once); unit = 1 instr", 64);
  run<1>("cmp_e32 vcc; s_nop 1; cndmask_e32 vcc; unit = 1 pair", 32);
  run<2>("cmp_e64 sgpr; s_nop 1; cndmask_e64 sgpr; unit = 1 pair", 32);
  run<3>("2x cmp_e64 sgpr; s_nop 0; 2x cndmask_e64; unit = 1 pair", 32);
  run<4>("v_cmp_e32 vcc only; unit = 1 instr", 32);
  run<5>("cmp_e32 vcc; 2 v_mul; cndmask_e32 vcc; unit = 4 instrs", 16);
  run<6>("v_max_f32 VOP2; unit = 1 instr", 64);
  return 0;
}
