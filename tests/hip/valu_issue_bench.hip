// VALU issue-cost microbenchmark (diagnostic, not part of the product or the test suite):
// cycles per instruction per SIMD for several instruction kinds at 1, 2, 4, 8 waves per SIMD.
// Build: hipcc -O2 --offload-arch=gfx950 tests/hip/valu_issue_bench.hip -o valu_issue_bench
// Results on MI355X: profiles/r01_valu_issue_microbench.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND>
__global__ void k(long long* out, float* sink, int iters, float seed) {
  float a = seed + threadIdx.x, b = seed * 2.f, c = 0.5f, d = 1.5f, e = 2.5f, f = 3.5f, g = 4.5f, h = 5.5f;
  unsigned long long m = __ballot(threadIdx.x & 1);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 1) asm volatile(REP16("v_cndmask_b32_e64 %0, %0, %1, %6\n v_cndmask_b32_e64 %3, %3, %1, %6\n v_cndmask_b32_e64 %4, %4, %1, %6\n v_cndmask_b32_e64 %5, %5, %1, %6\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(m));
    if (KIND == 2) asm volatile(REP16("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %3, %1\n v_cmp_lt_f32_e64 s[24:25], %4, %1\n v_cmp_lt_f32_e64 s[26:27], %5, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "s20","s21","s22","s23","s24","s25","s26","s27");
    if (KIND == 3) asm volatile(REP16("v_exp_f32 %0, %0\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 4) asm volatile(REP16("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %3, %3, %1, %2\n") : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e), "+v"(*(double*)&g));
    if (KIND == 6) asm volatile(REP16("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 7) asm volatile(REP16("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %3, %4\n v_permlane16_swap_b32 %5, %2\n v_permlane16_swap_b32 %6, %7\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    if (KIND == 8) asm volatile(REP16("v_rcp_f32 %0, %0\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 9) asm volatile(REP16("v_mul_f32 %0, %0, %1\n v_mul_f32 %3, %3, %1\n v_sub_f32 %4, %4, %1\n v_min_f32 %5, %5, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 10) asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %0, %1, %2\n v_fma_f32 %4, %3, %1, %2\n v_fma_f32 %5, %4, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 11) asm volatile(REP16("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %3, %1, %2\n v_fmac_f32 %4, %1, %2\n v_fmac_f32 %5, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 12) asm volatile(REP16("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %3, %3, %1\n") : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e), "+v"(*(double*)&g));
    if (KIND == 13) asm volatile(REP16("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %3, %3, %1\n") : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e), "+v"(*(double*)&g));
    if (KIND == 14) asm volatile(REP16("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %3, %3, %1, vcc\n v_cndmask_b32_e32 %4, %4, %1, vcc\n v_cndmask_b32_e32 %5, %5, %1, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");
    if (KIND == 15) asm volatile(REP16("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cmp_lt_f32_e32 vcc, %3, %1\n v_cmp_lt_f32_e32 vcc, %4, %1\n v_cmp_lt_f32_e32 vcc, %5, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");
    if (KIND == 16) asm volatile(REP16("v_mov_b32 %0, %1\n v_mov_b32 %3, %1\n v_mov_b32 %4, %1\n v_mov_b32 %5, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    // partial EXEC: does the SIMD skip 16-lane passes whose lanes are all off?
    if (KIND == 17 || KIND == 18 || KIND == 19) {
      const unsigned long long em = KIND == 17 ? 0xFFFFull : (KIND == 18 ? 0xFFFFFFFFull : 0x000F000F000F000Full);
      asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %6\n"
                   REP16("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2\n")
                   "s_mov_b64 exec, s[20:21]\n" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "s"(em) : "s20", "s21");
    }
    if (KIND == 20) asm volatile(REP16("v_fma_f32 %0, %0, %1, %2\n s_add_u32 s20, s20, 1\n v_fma_f32 %3, %3, %1, %2\n s_and_b32 s21, s20, 3\n v_fma_f32 %4, %4, %1, %2\n s_lshl_b32 s22, s21, 1\n v_fma_f32 %5, %5, %1, %2\n s_cmp_eq_u32 s22, 0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "s20", "s21", "s22", "scc");
    if (KIND == 21) asm volatile(REP16("v_mul_f32 %0, %0, %1\n v_fmac_f32 %3, %1, %2\n v_mul_f32 %4, %4, %1\n v_fmac_f32 %5, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 22) asm volatile(REP16("v_rcp_f32 %0, %0\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 23) asm volatile(REP16("v_exp_f32 %0, %0\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1\n v_mul_f32 %5, %5, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 24) asm volatile(REP16("v_mul_f32 %0, %0, %1\n v_mul_f32 %3, %3, %1\n v_mul_f32 %4, %4, %1\n v_mul_f32 %5, %5, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 25) asm volatile(REP16("v_add_f32 %0, %0, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1\n v_add_f32 %5, %5, %1\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 26) asm volatile(REP16("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %3, %3, %1, vcc\n v_cndmask_b32_e32 %4, %4, %1, vcc\n v_cndmask_b32_e32 %5, %5, %1, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");
    if (KIND == 27) asm volatile(REP16("v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %3, %3, %1, vcc\n v_cndmask_b32_e64 %4, %4, %1, vcc\n v_cndmask_b32_e64 %5, %5, %1, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");
    if (KIND == 28) asm volatile(REP16("v_cndmask_b32_e32 %0, %1, %2, vcc\n v_cndmask_b32_e32 %3, %1, %2, vcc\n v_cndmask_b32_e32 %4, %1, %2, vcc\n v_cndmask_b32_e32 %5, %1, %2, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");
    if (KIND == 29) asm volatile(REP16("v_cmp_lt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %3, 0, %1, vcc\n v_mul_f32 %4, %3, %1\n v_fma_f32 %5, %4, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "vcc");
    if (KIND == 30) asm volatile(REP16("v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %3, 0, %1, s[20:21]\n v_mul_f32 %4, %3, %1\n v_fma_f32 %5, %4, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "s20", "s21");
    if (KIND == 31) asm volatile(REP16("v_mul_f32 %0, %0, %1\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 32) asm volatile(REP16("v_mul_f32 %0, %0, %1\n v_mul_f32 %3, %3, %1\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 33) asm volatile(REP16("v_pk_fma_f32 %0, %0, %1, %2\n v_mul_f32 %4, %4, %5\n v_pk_fma_f32 %3, %3, %1, %2\n v_mul_f32 %5, %5, %4\n") : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e), "+v"(*(double*)&g), "+v"(d), "+v"(h));
    if (KIND == 34) asm volatile(REP16("v_cmp_lt_u64_e64 s[20:21], %0, %1\n v_cmp_lt_u64_e64 s[22:23], %1, %2\n v_cmp_lt_u64_e64 s[24:25], %2, %3\n v_cmp_lt_u64_e64 s[26:27], %3, %0\n") : "+v"(*(double*)&a), "+v"(*(double*)&c), "+v"(*(double*)&e), "+v"(*(double*)&g) :: "s20","s21","s22","s23","s24","s25","s26","s27");
    if (KIND == 35) asm volatile(REP16("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %0 row_ror:4 row_mask:0xf bank_mask:0x5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
    if (KIND == 36) asm volatile(REP16("ds_bpermute_b32 %0, %1, %0\n ds_bpermute_b32 %3, %1, %3\n ds_bpermute_b32 %4, %1, %4\n ds_bpermute_b32 %5, %1, %5\n s_waitcnt lgkmcnt(0)\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
  if (a + b + c + d + e + f + g + h == 12345.f) sink[0] = a;
}

template <int KIND>
void run(const char* name, int per_iter) {
  long long* out; float* sink;
  hipMalloc(&out, sizeof(long long) * 65536); hipMalloc(&sink, 4);
  const int iters = 2000;
  printf("%-28s", name);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;   // 256-thread blocks: 1 wave per SIMD each
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, sink, 10, 1.0f);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, sink, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), out, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v; s /= h.size();
    // cycles per instruction per SIMD (throughput) = wave cycles / (instrs) / waves-per-SIMD
    printf("  wps%d: %.2f cyc/instr/wave, %.2f cyc/instr/SIMD |", wps, s / ((double)iters * per_iter), s / ((double)iters * per_iter) / wps);
  }
  printf("\n");
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int which = argc > 1 ? atoi(argv[1]) : -1;
#define RUN(K, name, n) if (which < 0 || which == K) run<K>(name, n)
  RUN(0, "v_fma_f32 (indep x4)", 64);
  RUN(10, "v_fma_f32 (dependent chain)", 64);
  RUN(9, "v_mul/sub/min VOP2", 64);
  RUN(1, "v_cndmask_e64 (sgpr mask)", 64);
  RUN(2, "v_cmp_e64 -> sgpr", 64);
  RUN(3, "v_exp_f32", 64);
  RUN(8, "1 rcp + 3 fma", 64);
  RUN(4, "v_pk_fma_f32", 32);
  RUN(6, "v_add_f32_dpp", 64);
  RUN(7, "v_permlane32/16_swap", 64);
  RUN(11, "v_fmac_f32 (VOP2)", 64);
  RUN(24, "v_mul_f32 (VOP2)", 64);
  RUN(25, "v_add_f32 (VOP2)", 64);
  RUN(21, "v_mul + v_fmac mix", 64);
  RUN(12, "v_pk_mul_f32", 32);
  RUN(13, "v_pk_add_f32", 32);
  RUN(14, "v_cndmask_e32 (vcc)", 64);
  RUN(15, "v_cmp_e32 -> vcc", 64);
  RUN(16, "v_mov_b32", 64);
  RUN(22, "v_rcp_f32", 64);
  RUN(23, "1 exp + 3 mul", 64);
  RUN(17, "v_fma exec=16 lanes", 64);
  RUN(18, "v_fma exec=32 lanes", 64);
  RUN(19, "v_fma exec=4 lanes/row", 64);
  RUN(20, "v_fma + salu interleaved", 64);
  RUN(26, "1 cmp_e32 + 3 cndmask_e32", 64);
  RUN(27, "v_cndmask_e64 (vcc)", 64);
  RUN(28, "v_cndmask_e32 indep", 64);
  RUN(29, "cmp32/cndmask32/mul/fma", 64);
  RUN(30, "cmp64/cndmask64/mul/fma", 64);
  RUN(31, "1 mul + 3 fma", 64);
  RUN(32, "2 mul + 2 fma", 64);
  RUN(33, "pk_fma + mul alternating", 64);
  RUN(34, "v_cmp_lt_u64 -> sgpr", 64);
  RUN(35, "v_mov_b32_dpp", 64);
  RUN(36, "ds_bpermute_b32 (4 + wait)", 64);
  return 0;
}
