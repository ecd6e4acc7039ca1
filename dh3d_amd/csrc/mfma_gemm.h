// Exact-f32 MFMA tile GEMM shared by the fused point-major kernels.
//
// out[TM x Dout] = A[TM x Kd] (row-major in LDS) * W[Kd x Dout] (fragment-packed in global memory).
//
// v_mfma_f32_32x32x2_f32 takes ONE f32 of A and of B per lane per instruction:
//   A operand: lane l holds A[i = l&31][k = l>>5]      B operand: lane l holds B[k = l>>5][j = l&31]
//   D (16 regs): row = (reg&3) + 8*(reg>>2) + 4*(l>>5), col = l&31
// A dot product may visit k in any order, so k is consumed in blocks of 8 with the order
// {t, 4+t | t=0..3}: lane half h = l>>5 fetches the four consecutive values k = 8*kb + 4*h + t with a
// single 16-byte access (ds_read_b128 for A, global_load_dwordx4 for the packed W) and feeds them to
// four back-to-back MFMAs.  The result is a plain f32 fma chain per output (no reduced precision).
//
// Packed W layout (dh3d_pack_weight): packed[((nb*KB + kb)*64 + lane)*4 + t]
//      = W[8*kb + 4*(lane>>5) + t][32*nb + (lane&31)],  KB = Kd/8, nb = column block of 32.
// LDS leading dimension must be Kd+4 floats: 16-byte aligned rows and (ld/4) odd, which makes the
// ds_read_b128 A-fragment reads bank-conflict free (rows differ mod 16 inside each 16-lane group).
#pragma once
#include "common.h"

// NT column blocks (cb0, cb0+cbstride, ...) of one 32-row block starting at LDS row `row0`.
template <int NT>
__device__ __forceinline__ void wave_gemm_f32(const float *s_A, int ldA, int row0,
                                              const float *__restrict__ wpacked, int KB, int cb0,
                                              int cbstride, f32x16 (&acc)[NT]) {
  const int lane = threadIdx.x & 63;
  const float *aptr = s_A + (size_t)(row0 + (lane & 31)) * ldA + 4 * (lane >> 5);
  const f32x4 *wp = reinterpret_cast<const f32x4 *>(wpacked) + lane;
  // Main loop: k-blocks in groups of 4 with the NEXT group's B fragments already in flight.  The packed weight comes
  // from L2 (~700 cycles under load); one group is 16*NT MFMAs = 1024*NT cycles of matrix pipe, which covers it.
  // Two buffers that swap roles inside ONE loop body (no copy between them), requests unconditional (the last one a
  // harmless repeat) and fenced off from the products with a scheduling barrier.  (Rounds 1-3 had `cur = nxt` copies
  // and a conditional request: with KB a compile-time constant the loop was unrolled and the scheduler sank every
  // request next to its use -- one fragment in flight, s_waitcnt vmcnt(1) in front of every fourth MFMA; rolled, the
  // copies made the products wait for the requests just issued.  Found in the ISA, round 4.)
  const int NG = KB >> 2;
  int kb = NG * 4;
  if (NG > 0) {
    auto request = [&](f32x4 (&b)[4][NT], int g) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < NT; ++j) b[u][j] = wp[(size_t)((cb0 + j * cbstride) * KB + g * 4 + u) * 64];
    };
    auto multiply = [&](const f32x4 (&b)[4][NT], int g) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(aptr + (g * 4 + u) * 8);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b[u][j][0], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b[u][j][1], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b[u][j][2], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b[u][j][3], acc[j], 0, 0, 0);
        }
      }
    };
    f32x4 b0[4][NT], b1[4][NT];
    request(b0, 0);
    int g = 0;
#pragma unroll 1
    for (; g + 1 < NG; g += 2) {
      request(b1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(b0, g);
      request(b0, g + 2 < NG ? g + 2 : g + 1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(b1, g + 1);
    }
    if (g < NG) multiply(b0, g);  // an odd number of groups: the last one is already in b0
  }
  if (kb >= KB) return;
  // tail (KB not a multiple of 4): one k-block at a time
  f32x4 bnext[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bnext[j] = wp[(size_t)((cb0 + j * cbstride) * KB + kb) * 64];
  for (; kb < KB; ++kb) {
    const f32x4 a4 = *reinterpret_cast<const f32x4 *>(aptr + kb * 8);
    f32x4 bcur[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      bcur[j] = bnext[j];
      if (kb + 1 < KB) bnext[j] = wp[(size_t)((cb0 + j * cbstride) * KB + kb + 1) * 64];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], bcur[j][0], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], bcur[j][1], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], bcur[j][2], acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], bcur[j][3], acc[j], 0, 0, 0);
    }
  }
}

// Both 32-row blocks of a 64-row LDS tile against NTC column blocks (cb0, cb0+cbstride, ...): every packed
// weight fragment is fetched by exactly ONE wave of the workgroup and used for two MFMA row blocks.  (With
// the (row block, column half) split above two waves fetch each fragment; at 2 workgroups per CU that put
// ~20 TB/s of weight traffic on the L2s and capped the wide layers at ~60 % of the matrix pipe.)
template <int NTC>
__device__ __forceinline__ void wave_gemm_f32_rows2(const float *s_A, int ldA, const float *__restrict__ wpacked,
                                                    int KB, int cb0, int cbstride, f32x16 (&acc)[2][NTC]) {
  const int lane = threadIdx.x & 63;
  const float *aptr0 = s_A + (size_t)(lane & 31) * ldA + 4 * (lane >> 5);
  const float *aptr1 = aptr0 + (size_t)32 * ldA;
  const f32x4 *wp = reinterpret_cast<const f32x4 *>(wpacked) + lane;
  f32x4 nxt[2][NTC];  // two k-blocks in flight
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < NTC; ++j) nxt[u][j] = wp[(size_t)((cb0 + j * cbstride) * KB + (u < KB ? u : 0)) * 64];
  for (int kb = 0; kb < KB; kb += 2) {
    f32x4 cur[2][NTC];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        cur[u][j] = nxt[u][j];
        const int kn = kb + 2 + u;
        if (kn < KB) nxt[u][j] = wp[(size_t)((cb0 + j * cbstride) * KB + kn) * 64];
      }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (kb + u < KB) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(aptr0 + (kb + u) * 8);
        const f32x4 a1 = *reinterpret_cast<const f32x4 *>(aptr1 + (kb + u) * 8);
#pragma unroll
        for (int j = 0; j < NTC; ++j) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], cur[u][j][t], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], cur[u][j][t], acc[1][j], 0, 0, 0);
          }
        }
      }
    }
  }
}

__device__ __forceinline__ int mfma_row(int reg, int lane) {
  return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NT]) {
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
}

// Epilogue + store of one wave's tiles to a row-major [R, Dout] output; optional residual added
// after the activation.  grow0 = global row of LDS row 0 of this workgroup's tile.
template <int NT>
__device__ __forceinline__ void wave_store_f32(const f32x16 (&acc)[NT], long long grow0, int row0,
                                               int cb0, int cbstride, long long R, int Dout,
                                               const EpilogueArgs &ep, const float *__restrict__ residual,
                                               float *__restrict__ out) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = (cb0 + j * cbstride) * 32 + (lane & 31);
    float pb = 0.f, sc = 1.f, sh = 0.f;
    if (ep.pre_bias) pb = ep.pre_bias[col];
    if (ep.scale) sc = ep.scale[col];
    if (ep.shift) sh = ep.shift[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long grow = grow0 + row0 + mfma_row(r, lane);
      if (grow < R) {
        float v = dh3d_act((acc[j][r] + pb) * sc + sh, ep.act);
        if (residual) v += residual[grow * Dout + col];
        out[grow * Dout + col] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Wide epilogue.  Storing an accumulator tile straight from registers is 16 global_store_dword per tile per
// lane (256 B per instruction); measured on the flex_conv tile that store phase took as long as the GEMM
// itself (store-issue bound, tools/flex_probe.py).  Instead: epilogue in registers -> tile into LDS (the A
// tile is dead by then) -> the whole workgroup writes full rows with one 16-byte store per lane.
struct EpilogueRegs {
  float pb, sc, sh;
};
// fetch the per-column parameters early (before the GEMM) so their latency is off the tail
__device__ __forceinline__ EpilogueRegs epilogue_prefetch(const EpilogueArgs &ep, int col) {
  EpilogueRegs e{0.f, 1.f, 0.f};
  if (ep.pre_bias) e.pb = ep.pre_bias[col];
  if (ep.scale) e.sc = ep.scale[col];
  if (ep.shift) e.sh = ep.shift[col];
  return e;
}
template <int NT>
__device__ __forceinline__ void wave_tiles_to_lds(const f32x16 (&acc)[NT], const EpilogueRegs (&er)[NT], int act,
                                                  float *s_out, int ldo, int row0, int cb0, int cbstride) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int col = (cb0 + j * cbstride) * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r)
      s_out[(size_t)(row0 + mfma_row(r, lane)) * ldo + col] = dh3d_act((acc[j][r] + er[j].pb) * er[j].sc + er[j].sh, act);
  }
}
// all 256 threads: rows [0, TM) x Dout floats from LDS (+ residual) -> out, float4 per lane
__device__ __forceinline__ void block_store_rows(const float *s_out, int ldo, int TM, long long grow0, long long R,
                                                 int Dout, const float *__restrict__ residual,
                                                 float *__restrict__ out) {
  const int cv = Dout / 4;
  for (int e = threadIdx.x; e < TM * cv; e += 256) {
    const int p = e / cv, c4 = (e - p * cv) * 4;
    const long long g = grow0 + p;
    if (g < R) {
      float4 v = *reinterpret_cast<const float4 *>(s_out + (size_t)p * ldo + c4);
      if (residual) {
        const float4 q = *reinterpret_cast<const float4 *>(residual + g * Dout + c4);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      *reinterpret_cast<float4 *>(out + g * Dout + c4) = v;
    }
  }
}
