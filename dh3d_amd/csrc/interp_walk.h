// The walk over the fine points that the training kernels of interp_train.hip and netvlad_train.hip share (gfx950).
//
// A workgroup of 256 threads owns 128 consecutive points of one cloud in Morton order (records of dh3d_spatial_sort).
// Every point mixes three coarse rows (three_nn indices, inverse-distance weights, core/backbones.py:89-100); a block of
// spatially coherent points touches few distinct coarse rows (46 on average at N/8 samples, 62 at most), so the block
// builds a SLOT TABLE: a bitmap of the coarse rows it touches -> prefix sums -> slot = rank of the row among them.  Rows
// with a slot below CAP are staged in LDS by the caller, the rest are addressed in memory (slot = -1 - row: the
// overflow path).  Per point the table holds the three slots, 1 + the point's original index (0: padding point of the
// last block) and the three weights plus one per-point scalar of the caller's choice.
#pragma once
#include <hip/hip_runtime.h>

namespace dh3d_walk {

constexpr int kP = 128;  // fine points per workgroup

// must round like three_interp_fwd_kernel<IDW> / interp_head_lds_kernel (no contraction)
#pragma clang fp contract(off)
__device__ __forceinline__ void idw3(float d1, float d2, float d3, float &w1, float &w2, float &w3) {
  const float r1 = 1.0f / fmaxf(d1, 1e-10f), r2 = 1.0f / fmaxf(d2, 1e-10f), r3 = 1.0f / fmaxf(d3, 1e-10f);
  const float norm = (r1 + r2) + r3;
  w1 = r1 / norm; w2 = r2 / norm; w3 = r3 / norm;
}
__device__ __forceinline__ float4 mix3(const float4 a, const float4 b, const float4 c, float w1, float w2, float w3) {
  float4 r;
  r.x = (a.x * w1 + b.x * w2) + c.x * w3;
  r.y = (a.y * w1 + b.y * w2) + c.y * w3;
  r.z = (a.z * w1 + b.z * w2) + c.z * w3;
  r.w = (a.w * w1 + b.w * w2) + c.w * w3;
  return r;
}
#pragma clang fp contract(fast)

struct SlotTable {
  int *slot;       // [kP][4]: slots (or -1 - coarse row) of the three neighbours, .w = 1 + original index (0: none)
  float *w;        // [kP][4]: the three weights, .w = the caller's per-point scalar (0 on padding points)
  unsigned *bits;  // [32]: bitmap of the coarse rows of the cloud (m <= 1024) this block touches
  int *pre;        // [33]: exclusive prefix sums of the bitmap words' population counts, [32] = the total
  int *row;        // [CAP]: slot -> coarse row
};

// Builds the table for block `blk` of cloud `bi` (all 256 threads; contains barriers).  idx [B,n,3]; weights from
// `weight` [B,n,3] if given, else inverse-distance weights of `dist` [B,n,3]; order [B,n] spatial_sort records or null
// (index order); scalar(r, tid) -> the per-point scalar of the point with global row r (called by thread tid < kP for
// live points only).  Returns the number of distinct rows (nd = min(that, CAP) are staged; more: overflow).
template <int CAP, typename Scalar>
__device__ __forceinline__ int build_slot_table(const SlotTable &T, const int32_t *__restrict__ idx,
                                                const float *__restrict__ dist, const float *__restrict__ weight,
                                                const float4 *__restrict__ order, int bi, int blk, int n, int m,
                                                Scalar &&scalar) {
  const int tid = threadIdx.x;
  if (tid < 32) T.bits[tid] = 0u;
  __syncthreads();
  int my_i[3] = {0, 0, 0}, my_orig = 0;
  bool have = false;
  if (tid < kP) {
    const int q = blk * kP + tid;
    if (q < n) {
      const int orig = order ? __float_as_int(order[(size_t)bi * n + q].w) : q;
      const long long r = (long long)bi * n + orig;
      float w1, w2, w3;
      if (weight) { w1 = weight[r * 3]; w2 = weight[r * 3 + 1]; w3 = weight[r * 3 + 2]; }
      else idw3(dist[r * 3], dist[r * 3 + 1], dist[r * 3 + 2], w1, w2, w3);
      *reinterpret_cast<float4 *>(T.w + tid * 4) = make_float4(w1, w2, w3, scalar(r, tid));
      my_orig = orig;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        my_i[t] = idx[r * 3 + t];
        atomicOr(&T.bits[my_i[t] >> 5], 1u << (my_i[t] & 31));
      }
      have = true;
    } else {  // padding point of the last block: slot 0 with zero weights, flagged invalid
      *reinterpret_cast<float4 *>(T.w + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<int4 *>(T.slot + tid * 4) = make_int4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  if (tid < 64) {
    int c = tid < 32 ? __popc(T.bits[tid]) : 0, v = c;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const int o = __shfl_up(v, off, 64);
      if ((tid & 63) >= off) v += o;
    }
    if (tid < 32) T.pre[tid] = v - c;
    if (tid == 31) T.pre[32] = v;
  }
  __syncthreads();
  if (have) {
    int sl3[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int j = my_i[t];
      const int slot = T.pre[j >> 5] + __popc(T.bits[j >> 5] & ((1u << (j & 31)) - 1u));
      sl3[t] = slot < CAP ? slot : -1 - j;
    }
    *reinterpret_cast<int4 *>(T.slot + tid * 4) = make_int4(sl3[0], sl3[1], sl3[2], 1 + my_orig);
  }
  for (int j = tid; j < m; j += 256) {
    if ((T.bits[j >> 5] >> (j & 31)) & 1u) {
      const int slot = T.pre[j >> 5] + __popc(T.bits[j >> 5] & ((1u << (j & 31)) - 1u));
      if (slot < CAP) T.row[slot] = j;
    }
  }
  __syncthreads();
  return T.pre[32];
}

}  // namespace dh3d_walk
