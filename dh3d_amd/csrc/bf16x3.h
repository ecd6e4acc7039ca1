// Exact three-way split of an f32 into bf16 chunks, a = c1 + c2 + c3 (8 + 8 + 8 significand bits, truncation;
// every remainder is exactly representable in f32).  Feeding the six products c_i * d_j with i + j <= 4 to the
// bf16 matrix pipe (f32 accumulate) reproduces the f32 product to ~2^-23 relative: an f32-accurate GEMM at
// 6 bf16 MFMAs per K=16 instead of 8 f32 MFMAs per K=16 (gfx950: 2.7x less matrix-pipe time).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float a, unsigned &c1, unsigned &c2, unsigned &c3) {
  const unsigned u1 = __float_as_uint(a) & 0xFFFF0000u;
  const float r1 = a - __uint_as_float(u1);                 // exact
  const unsigned u2 = __float_as_uint(r1) & 0xFFFF0000u;
  const float r2 = r1 - __uint_as_float(u2);                // exact, <= 8 significant bits left
  c1 = u1 >> 16; c2 = u2 >> 16; c3 = __float_as_uint(r2) >> 16;
}

// (hi16(b) << 16) | hi16(a): two truncated bf16 in one dword, one v_perm_b32
__device__ __forceinline__ unsigned pack_hi16(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}

// four f32 -> three planes of four bf16 (8 bytes each)
__device__ __forceinline__ void split3x4(const float4 v, uint2 &c1, uint2 &c2, uint2 &c3) {
  const float ax = __uint_as_float(__float_as_uint(v.x) & 0xFFFF0000u), ay = __uint_as_float(__float_as_uint(v.y) & 0xFFFF0000u);
  const float az = __uint_as_float(__float_as_uint(v.z) & 0xFFFF0000u), aw = __uint_as_float(__float_as_uint(v.w) & 0xFFFF0000u);
  const float rx = v.x - ax, ry = v.y - ay, rz = v.z - az, rw = v.w - aw;
  const float bx = __uint_as_float(__float_as_uint(rx) & 0xFFFF0000u), by = __uint_as_float(__float_as_uint(ry) & 0xFFFF0000u);
  const float bz = __uint_as_float(__float_as_uint(rz) & 0xFFFF0000u), bw = __uint_as_float(__float_as_uint(rw) & 0xFFFF0000u);
  const float tx = rx - bx, ty = ry - by, tz = rz - bz, tw = rw - bw;
  c1 = make_uint2(pack_hi16(v.x, v.y), pack_hi16(v.z, v.w));
  c2 = make_uint2(pack_hi16(rx, ry), pack_hi16(rz, rw));
  c3 = make_uint2(pack_hi16(tx, ty), pack_hi16(tz, tw));
}
