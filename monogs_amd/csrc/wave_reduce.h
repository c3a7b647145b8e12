// Ten wave64 float sums with a reduce-scatter tree (gfx950).
//
// A plain DPP butterfly needs 6 adds per value (60 for the ten per-splat sums of the
// blend backward).  Here the first two levels use v_permlane32_swap / v_permlane16_swap,
// which exchange half-waves / 16-lane rows BETWEEN two registers: one swap + one add halves
// the number of live registers instead of leaving every lane with a copy.  28 instructions
// instead of 60.
//
// On return
//   main : lanes with (lane & 7) == 0 hold the wave total of r[idx],
//          idx = ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1) + 4 * ((lane >> 3) & 1)   (r[0..7])
//   extra: lane 31 holds the total of r[8], lane 63 the total of r[9]
// `b3mask` must be __ballot((lane & 8) != 0).  All 64 lanes must be active.
//
// Written as one asm block: the compiler otherwise SLP-packs neighbouring adds into
// v_pk_add_f32 (no DPP modifier) and the DPP/permlane read-after-write wait states are
// placed by hand (s_nop) because hipcc pads nothing inside an asm statement.
// (The hazard itself - as hipcc's recogniser places it for the builtin forms - is two wait states between a VALU write
// and a DPP / v_permlane*_swap read of the register, none behind a swap; the s_nop 1 behind every group below is more
// than that.  Round 5 trimmed the ten-sum block to the minimum, five s_nop instead of ten, results identical: k_blend_bwd
// 101.0 -> 103.0 us on one box, back to back, twice.  The padding stays: the idle slots go to the SIMD's other waves.)
#pragma once
#include <hip/hip_runtime.h>

namespace mgs {

__device__ __forceinline__ void wave_sum10_scatter(float (&r)[10], unsigned long long b3mask,
                                                   float& main, float& extra) {
  float keep, send;
  asm volatile(
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1\n\t"
      "v_permlane32_swap_b32 %2, %3\n\t"
      "v_permlane32_swap_b32 %4, %5\n\t"
      "v_permlane32_swap_b32 %6, %7\n\t"
      "v_permlane32_swap_b32 %8, %9\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %1\n\t"
      "v_add_f32 %2, %2, %3\n\t"
      "v_add_f32 %4, %4, %5\n\t"
      "v_add_f32 %6, %6, %7\n\t"
      "v_add_f32 %8, %8, %9\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %2\n\t"
      "v_permlane16_swap_b32 %4, %6\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %2\n\t"
      "v_add_f32 %4, %4, %6\n\t"
      "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %10, %0, %4, %12\n\t"
      "v_cndmask_b32_e64 %11, %4, %0, %12\n\t"
      "v_add_f32_dpp %8, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %10, %11, %10 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %10, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %8, %8, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %10, %10, %10 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %10, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]),
        "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "=&v"(keep), "=&v"(send)
      : "s"(b3mask));
  main = keep;
  extra = r[8];
}

// Six wave64 sums (the pose-only backward needs Sx, Sy, Sxx, Sxy, Syy, Rd only): r[0..3] go
// through one permlane32 and one permlane16 level and a 4-step DPP row reduction, r[4..5]
// through one permlane32 level and a 5-step DPP reduction.  17 instructions.  On return
//   main : every lane of 16-lane row R holds the wave total of r[idx],
//          idx = ((lane >> 5) & 1) + 2 * ((lane >> 4) & 1)            (r[0..3])
//   extra: lane 31 holds the total of r[4], lane 63 the total of r[5]
// All 64 lanes must be active.
__device__ __forceinline__ void wave_sum6_scatter(float (&r)[6], float& main, float& extra) {
  asm volatile(
      "s_nop 1\n\t"
      "v_permlane32_swap_b32 %0, %1\n\t"
      "v_permlane32_swap_b32 %2, %3\n\t"
      "v_permlane32_swap_b32 %4, %5\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %1\n\t"
      "v_add_f32 %2, %2, %3\n\t"
      "v_add_f32 %4, %4, %5\n\t"
      "s_nop 1\n\t"
      "v_permlane16_swap_b32 %0, %2\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32 %0, %0, %2\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]));
  main = r[0];
  extra = r[4];
}

}  // namespace mgs
