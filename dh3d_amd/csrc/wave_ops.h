// Wave64 reductions on the DPP crossbar (no LDS round trip, ONE VALU op per step) for gfx950.
//   quad_perm [1,0,3,2] -> quad_perm [2,3,0,1] -> row_half_mirror -> row_mirror   : every lane of a
//   16-lane row holds the row result; row_bcast:15 (rows 1,3) -> row_bcast:31 (rows 2,3): lane 63 holds
//   the wave result, which v_readlane broadcasts through an SGPR.  Valid for idempotent ops (max / min).
// Written as inline asm so that each step is a single v_max_f32_dpp / v_min_i32_dpp (hipcc emits
// v_mov_b32_dpp + op + a canonicalising v_max per step); the `s_nop 1` in front of every step is the
// gfx9 "VALU write -> DPP read of the same VGPR" hazard (2 wait states), which the compiler does not
// pad inside an asm statement.  Rows disabled by row_mask keep their value (dst == src).
#pragma once
#include <hip/hip_runtime.h>

#define DH3D_DPP_ROW16(OP)                                                      \
  "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
  "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
  "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
  "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
#define DH3D_DPP_ROWS(OP)                                                       \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"    \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"    \
  "s_nop 1\n\t"

// The same chains on a NAMED asm operand (for hand-written loops): wave result in lane 63 of %[R].
#define DH3D_DPP_WAVE_N(OP, R)                                                                   \
  "s_nop 1\n\t" OP " %[" R "], %[" R "], %[" R "] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
  "s_nop 1\n\t" OP " %[" R "], %[" R "], %[" R "] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
  "s_nop 1\n\t" OP " %[" R "], %[" R "], %[" R "] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
  "s_nop 1\n\t" OP " %[" R "], %[" R "], %[" R "] row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
  "s_nop 1\n\t" OP " %[" R "], %[" R "], %[" R "] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
  "s_nop 1\n\t" OP " %[" R "], %[" R "], %[" R "] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"

// max / min over each 16-lane row, result in every lane of the row
__device__ __forceinline__ float row16_max_f32(float v) {
  asm volatile(DH3D_DPP_ROW16("v_max_f32_dpp") "s_nop 1\n\t" : "+v"(v));
  return v;
}
__device__ __forceinline__ int row16_min_i32(int v) {
  asm volatile(DH3D_DPP_ROW16("v_min_i32_dpp") "s_nop 1\n\t" : "+v"(v));
  return v;
}
// wave-wide, result uniform (SGPR)
__device__ __forceinline__ float wave_max_f32(float v) {
  asm volatile(DH3D_DPP_ROW16("v_max_f32_dpp") DH3D_DPP_ROWS("v_max_f32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_i32(int v) {
  asm volatile(DH3D_DPP_ROW16("v_min_i32_dpp") DH3D_DPP_ROWS("v_min_i32_dpp") : "+v"(v));
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  asm volatile(DH3D_DPP_ROW16("v_min_u32_dpp") DH3D_DPP_ROWS("v_min_u32_dpp") : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ float wave_min_f32(float v) {
  asm volatile(DH3D_DPP_ROW16("v_min_f32_dpp") DH3D_DPP_ROWS("v_min_f32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float row16_min_f32(float v) {
  asm volatile(DH3D_DPP_ROW16("v_min_f32_dpp") "s_nop 1\n\t" : "+v"(v));
  return v;
}

// Sums (not idempotent, but every step pairs DISJOINT partial groups, so nothing is counted twice):
// after the four row steps every lane of a 16-lane row holds the row's sum; row_bcast:15 adds row 0's (2's) sum into
// row 1 (3).  Result: lanes 16..31 hold the sum of lanes 0..31, lanes 48..63 the sum of lanes 32..63.
__device__ __forceinline__ float row16_sum_f32(float v) {  // every lane: the sum over its 16-lane row
  asm volatile(DH3D_DPP_ROW16("v_add_f32_dpp") "s_nop 1\n\t" : "+v"(v));
  return v;
}
__device__ __forceinline__ float half32_sum_f32(float v) {
  asm volatile(DH3D_DPP_ROW16("v_add_f32_dpp")
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
               : "+v"(v));
  return v;
}
// Two per-lane partial sums a, b (one value per lane each) -> lanes 16..31: sum of a over the wave, lanes 48..63: sum
// of b over the wave.  v_permlane32_swap (gfx950) exchanges a's upper half with b's lower half in one instruction.
__device__ __forceinline__ float pair_wave_sum_f32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return half32_sum_f32(__uint_as_float(r[0]) + __uint_as_float(r[1]));
}
