// PointNet++ grouping / interpolation operators for gfx950.
//   group_point(+grad)        replaces tf_ops/grouping/tf_grouping_g.cu:94-132 (one CUDA block per
//                             cloud, serial over channels) with a flat, channel-coalesced gather.
//   three_nn                  replaces the host loop tf_ops/interpolation/tf_interpolate.cpp:60-103
//                             (the reference has no GPU kernel: every call crossed PCIe twice).
//   three_interpolate(+grad)  replaces tf_interpolate.cpp:107-153.
//   three_interpolate_idw     fuses the weight computation of core/backbones.py:92-95.
// Arithmetic that decides integer outputs (three_nn) is written with explicit, unfused roundings.
#include <limits.h>

#include "common.h"
#include "wave_ops.h"

#pragma clang fp contract(off)

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------ group_point
__global__ __launch_bounds__(kBlock) void group_point_fwd_kernel(
    long long rows, int n, int c, int rows_per_cloud, const float *__restrict__ points,
    const int32_t *__restrict__ idx, float *__restrict__ out) {
  const long long total = rows * c;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const long long row = e / c;
    const int l = (int)(e - row * c);
    const long long bi = row / rows_per_cloud;
    const int ii = idx[row];
    out[e] = points[(bi * n + ii) * c + l];
  }
}

// c % 4 == 0: one 16-byte element per lane (the model's gathers: 64 / 128 channels)
__global__ __launch_bounds__(kBlock) void group_point_fwd4_kernel(
    long long rows, int n, int c4, int rows_per_cloud, const float4 *__restrict__ points,
    const int32_t *__restrict__ idx, float4 *__restrict__ out) {
  const long long total = rows * c4;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const long long row = e / c4;
    const int l = (int)(e - row * c4);
    const long long bi = row / rows_per_cloud;
    out[e] = points[(bi * n + idx[row]) * c4 + l];
  }
}

__global__ __launch_bounds__(kBlock) void group_point_bwd_kernel(
    long long rows, int n, int c, int rows_per_cloud, const float *__restrict__ grad_out,
    const int32_t *__restrict__ idx, float *__restrict__ grad_points) {
  const long long total = rows * c;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const long long row = e / c;
    const int l = (int)(e - row * c);
    const long long bi = row / rows_per_cloud;
    const int ii = idx[row];
    atomicAdd(&grad_points[(bi * n + ii) * c + l], grad_out[e]);
  }
}

// ------------------------------------------------------------------ three_nn
constexpr int kNNChunk = 1024;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// LDS image of 4 consecutive candidates (12 floats): [x0 x1 y0 y1] [z0 z1 x2 x3] [y2 y3 z2 z3]
__device__ __forceinline__ int nn_slot(int c, int comp) {
  const int g = c >> 2, r = c & 3;
  return g * 12 + ((r < 2) ? (2 * comp + r) : (4 + 2 * comp + r));
}

// lane = query (64 per workgroup); the four waves split the candidates (32-candidate steps, interleaved) so a
// query group keeps 4 waves per SIMD busy instead of one long scan, and their four 3-deep lists are merged
// through LDS on (distance, index) -- the reference's strict '<' in index order = smallest index among equals.
// 8 candidates per packed-f32 sub-step (unfused mul/add, the reference's roundings); the 3-deep insertion only
// runs when one of them beats the lane's current third-best.
__global__ __launch_bounds__(kBlock) void three_nn_kernel(int n, int m, const float *__restrict__ xyz1,
                                                         const float *__restrict__ xyz2,
                                                         float *__restrict__ dist,
                                                         int32_t *__restrict__ idx) {
  __shared__ __attribute__((aligned(16))) float s_c[kNNChunk * 3];
  __shared__ float s_md[3][3][64];  // partial lists of waves 1..3: [wave-1][rank][query]
  __shared__ int s_mi[3][3][64];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const float *q = xyz1 + ((size_t)b * n + j) * 3;
  const float *cand = xyz2 + (size_t)b * m * 3;
  float x1 = 0.f, y1 = 0.f, z1 = 0.f;
  if (j < n) { x1 = q[0]; y1 = q[1]; z1 = q[2]; }
  const f32x2 qx = {x1, x1}, qy = {y1, y1}, qz = {z1, z1};
  // double 1e40 of the reference: any finite float compares below it, +inf does not.
  float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
  int bi1 = 0, bi2 = 0, bi3 = 0;
  for (int base = 0; base < m; base += kNNChunk) {
    const int len = min(kNNChunk, m - base);
    const int len8 = (len + 31) & ~31;  // padded (+inf) to the 32-candidate step
    const int len32 = len8;
    __syncthreads();
    for (int e = threadIdx.x; e < len8 * 3; e += kBlock) {
      const int c = e / 3, comp = e - c * 3;
      s_c[nn_slot(c, comp)] = c < len ? cand[(size_t)base * 3 + e] : INFINITY;  // padding: d = inf, never < best
    }
    __syncthreads();
    // 32 candidates per iteration with one wave-uniform test, so the LDS reads and packed ops pipeline
    for (int k = 32 * wave; k < len32; k += 32 * (kBlock / 64)) {
      f32x2 d[4][4];
      float mn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float *g0 = s_c + ((k >> 2) + 2 * u) * 12;
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(g0), a1 = *reinterpret_cast<const f32x4 *>(g0 + 4),
                    a2 = *reinterpret_cast<const f32x4 *>(g0 + 8), b0 = *reinterpret_cast<const f32x4 *>(g0 + 12),
                    b1 = *reinterpret_cast<const f32x4 *>(g0 + 16), b2 = *reinterpret_cast<const f32x4 *>(g0 + 20);
        {
          const f32x2 dx = f32x2{a0[0], a0[1]} - qx, dy = f32x2{a0[2], a0[3]} - qy, dz = f32x2{a1[0], a1[1]} - qz;
          d[u][0] = (dx * dx + dy * dy) + dz * dz;
        }
        {
          const f32x2 dx = f32x2{a1[2], a1[3]} - qx, dy = f32x2{a2[0], a2[1]} - qy, dz = f32x2{a2[2], a2[3]} - qz;
          d[u][1] = (dx * dx + dy * dy) + dz * dz;
        }
        {
          const f32x2 dx = f32x2{b0[0], b0[1]} - qx, dy = f32x2{b0[2], b0[3]} - qy, dz = f32x2{b1[0], b1[1]} - qz;
          d[u][2] = (dx * dx + dy * dy) + dz * dz;
        }
        {
          const f32x2 dx = f32x2{b1[2], b1[3]} - qx, dy = f32x2{b2[0], b2[1]} - qy, dz = f32x2{b2[2], b2[3]} - qz;
          d[u][3] = (dx * dx + dy * dy) + dz * dz;
        }
        mn[u] = fminf(fminf(fminf(d[u][0][0], d[u][0][1]), fminf(d[u][1][0], d[u][1][1])),
                      fminf(fminf(d[u][2][0], d[u][2][1]), fminf(d[u][3][0], d[u][3][1])));
      }
      const float m = fminf(fminf(mn[0], mn[1]), fminf(mn[2], mn[3]));
      if (__any(j < n && m < best3)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j < n && mn[u] < best3) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {  // in index order: strict '<' keeps the first of equal distances
              const float dv = d[u][t >> 1][t & 1];
              const int kk = base + k + 8 * u + t;
              if (dv < best1) {
                best3 = best2; bi3 = bi2; best2 = best1; bi2 = bi1; best1 = dv; bi1 = kk;
              } else if (dv < best2) {
                best3 = best2; bi3 = bi2; best2 = dv; bi2 = kk;
              } else if (dv < best3) {
                best3 = dv; bi3 = kk;
              }
            }
          }
        }
      }
    }
  }
  if (wave > 0) {
    s_md[wave - 1][0][lane] = best1; s_md[wave - 1][1][lane] = best2; s_md[wave - 1][2][lane] = best3;
    s_mi[wave - 1][0][lane] = bi1; s_mi[wave - 1][1][lane] = bi2; s_mi[wave - 1][2][lane] = bi3;
  }
  __syncthreads();
  if (wave == 0 && j < n) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const float dv = s_md[w][t][lane];
        const int kk = s_mi[w][t][lane];
        if (dv < best1 || (dv == best1 && kk < bi1)) {
          best3 = best2; bi3 = bi2; best2 = best1; bi2 = bi1; best1 = dv; bi1 = kk;
        } else if (dv < best2 || (dv == best2 && kk < bi2)) {
          best3 = best2; bi3 = bi2; best2 = dv; bi2 = kk;
        } else if (dv < best3 || (dv == best3 && kk < bi3)) {
          best3 = dv; bi3 = kk;
        }
      }
    const size_t o = ((size_t)b * n + j) * 3;
    dist[o] = best1; dist[o + 1] = best2; dist[o + 2] = best3;
    idx[o] = bi1; idx[o + 1] = bi2; idx[o + 2] = bi3;
  }
}

// ------------------------------------------------------------------ three_nn on ordered clouds
// Same outputs as three_nn_kernel from the Morton-ordered records of both sets (spatial.hip: float4 (x, y, z,
// bits(original index))).  What the brute-force kernel spends its time on is not the 8 flop per pair but the 3-deep
// insertion: with queries and candidates in arbitrary order SOME lane of the wave improves its list in nearly every
// 8-candidate sub-step (P ~ 64 * 8 * 3/k until k > 1536), so the branch is taken all the way through.  Here a wave
// holds 64 CONSECUTIVE queries of the Morton order (a compact region) and starts its candidate scan at the position
// of the sampled set's own Morton order that corresponds to that region (farthest-point samples spread evenly over
// the cloud, so rank-proportional is close), wrapping around: the lists are nearly final after the first 128
// candidates and the rest of the scan is the bare distance test.  Every candidate is still visited -- no pruning
// structure, nothing to get wrong -- and equal distances resolve to the smallest candidate index explicitly, so ids
// and distances are bit-identical to three_nn_kernel whatever the visiting order (tf_interpolate.cpp:66-96).
__global__ __launch_bounds__(kBlock) void three_nn_sorted_kernel(int n, int m, const float4 *__restrict__ qs,
                                                                const float4 *__restrict__ cs,
                                                                float *__restrict__ dist, int32_t *__restrict__ idx) {
  __shared__ __attribute__((aligned(16))) float s_c[kNNChunk * 3];
  __shared__ int s_k[kNNChunk];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const bool valid = j < n;
  const float4 q = qs[(size_t)b * n + (valid ? j : blockIdx.x * 64)];
  const float4 *cand = cs + (size_t)b * m;
  const f32x2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
  // The 3-deep list as 64-bit keys (bits(d) << 32 | index): d >= 0, so unsigned order is (distance, index) -- the
  // smallest index among equal distances, i.e. the reference's strict '<' in index order -- and an insertion is three
  // compares + selects with no branch (the branchy form was ~5x the cost of the distance arithmetic per visit).
  // Start: (+inf, 0), the reference's 1e40 / index 0.
  const unsigned long long kInf = (unsigned long long)__float_as_uint(INFINITY) << 32;
  unsigned long long k1 = kInf, k2 = kInf, k3 = kInf;
  // rank-proportional start, 64 candidates before the centre of this query group's image, aligned to the 32-step
  const int nchunks = (m + kNNChunk - 1) / kNNChunk;
  long long c0 = ((long long)blockIdx.x * 64 + 32) * m / n - 64;
  c0 = c0 < 0 ? 0 : c0;
  const int chunk0 = (int)(c0 / kNNChunk);
  for (int ci = 0; ci < nchunks; ++ci) {
    const int base = ((chunk0 + ci) % nchunks) * kNNChunk;
    const int len = min(kNNChunk, m - base);
    const int len32 = (len + 31) & ~31;  // padded (+inf) to the 32-candidate step
    int k0 = ci == 0 ? ((int)(c0 - base) & ~31) : 0;
    k0 = k0 < len32 ? k0 : 0;
    __syncthreads();
    for (int e = threadIdx.x; e < len32; e += kBlock) {
      float4 r = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(INT_MAX));  // padding: d = inf, never taken
      if (e < len) r = cand[base + e];
      s_c[nn_slot(e, 0)] = r.x; s_c[nn_slot(e, 1)] = r.y; s_c[nn_slot(e, 2)] = r.z;
      s_k[e] = __float_as_int(r.w);
    }
    __syncthreads();
    for (int kr = 32 * wave; kr < len32; kr += 32 * (kBlock / 64)) {
      int k = k0 + kr;
      k = k >= len32 ? k - len32 : k;  // wrap around inside the chunk
      f32x2 d[4][4];
      float mn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float *g0 = s_c + ((k >> 2) + 2 * u) * 12;
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(g0), a1 = *reinterpret_cast<const f32x4 *>(g0 + 4),
                    a2 = *reinterpret_cast<const f32x4 *>(g0 + 8), b0 = *reinterpret_cast<const f32x4 *>(g0 + 12),
                    b1 = *reinterpret_cast<const f32x4 *>(g0 + 16), b2 = *reinterpret_cast<const f32x4 *>(g0 + 20);
        {
          const f32x2 dx = f32x2{a0[0], a0[1]} - qx, dy = f32x2{a0[2], a0[3]} - qy, dz = f32x2{a1[0], a1[1]} - qz;
          d[u][0] = (dx * dx + dy * dy) + dz * dz;
        }
        {
          const f32x2 dx = f32x2{a1[2], a1[3]} - qx, dy = f32x2{a2[0], a2[1]} - qy, dz = f32x2{a2[2], a2[3]} - qz;
          d[u][1] = (dx * dx + dy * dy) + dz * dz;
        }
        {
          const f32x2 dx = f32x2{b0[0], b0[1]} - qx, dy = f32x2{b0[2], b0[3]} - qy, dz = f32x2{b1[0], b1[1]} - qz;
          d[u][2] = (dx * dx + dy * dy) + dz * dz;
        }
        {
          const f32x2 dx = f32x2{b1[2], b1[3]} - qx, dy = f32x2{b2[0], b2[1]} - qy, dz = f32x2{b2[2], b2[3]} - qz;
          d[u][3] = (dx * dx + dy * dy) + dz * dz;
        }
        mn[u] = fminf(fminf(fminf(d[u][0][0], d[u][0][1]), fminf(d[u][1][0], d[u][1][1])),
                      fminf(fminf(d[u][2][0], d[u][2][1]), fminf(d[u][3][0], d[u][3][1])));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // '<=': an equal distance with a smaller index must still be seen
        if (__any(valid && mn[u] <= __uint_as_float((unsigned)(k3 >> 32)))) {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const unsigned long long x = ((unsigned long long)__float_as_uint(d[u][t >> 1][t & 1]) << 32) |
                                         (unsigned)s_k[k + 8 * u + t];
            const bool l1 = x < k1, l2 = x < k2, l3 = x < k3;
            k3 = l2 ? k2 : (l3 ? x : k3);
            k2 = l1 ? k1 : (l2 ? x : k2);
            k1 = l1 ? x : k1;
          }
        }
      }
    }
  }
  __shared__ unsigned long long s_mk[3][3][64];  // partial lists of waves 1..3
  if (wave > 0) { s_mk[wave - 1][0][lane] = k1; s_mk[wave - 1][1][lane] = k2; s_mk[wave - 1][2][lane] = k3; }
  __syncthreads();
  if (wave == 0 && valid) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned long long x = s_mk[w][t][lane];
        const bool l1 = x < k1, l2 = x < k2, l3 = x < k3;
        k3 = l2 ? k2 : (l3 ? x : k3);
        k2 = l1 ? k1 : (l2 ? x : k2);
        k1 = l1 ? x : k1;
      }
    const size_t o = ((size_t)b * n + __float_as_int(q.w)) * 3;
    dist[o] = __uint_as_float((unsigned)(k1 >> 32)); dist[o + 1] = __uint_as_float((unsigned)(k2 >> 32));
    dist[o + 2] = __uint_as_float((unsigned)(k3 >> 32));
    idx[o] = (int)(unsigned)k1; idx[o + 1] = (int)(unsigned)k2; idx[o + 2] = (int)(unsigned)k3;
  }
}

// The same search with the candidates pruned by their group boxes (m <= kNNChunk: the whole sampled set in LDS -- every
// DH3D level).  three_nn_sorted_kernel evaluates every candidate for every query: 2.5 k VALU instructions per wave, the
// kernel is VALU-issue bound (profiles/r03_d_pmc_three_nn.txt: 37 us for 8 x 8192 against 8 x 1024).  Here
//   A. every wave scans ONE 32-candidate step around the rank-proportional start (128 candidates per query group):
//      every lane gets a 3-deep list; the waves share, per query, the smallest of their third distances -- each is an
//      upper bound of the query's true third distance;
//   B. lane = candidate group (64 Morton-consecutive samples, box from dh3d_spatial_sort): the groups whose box lies
//      within the loosest bound of the query group's box are dealt round-robin to the waves; before a group is scanned
//      the same test per QUERY (point to box, own bound); steps of phase A are not scanned again.
// Skipping is exact: the box distance carries a (1 - 1e-5) factor against roundings of a few 1e-7, and '<=' keeps a
// candidate at exactly the third distance (a smaller index must still be seen).  Same lists, same merge as above.
__global__ __launch_bounds__(kBlock) void three_nn_pruned_kernel(int n, int m, const float4 *__restrict__ qs,
                                                                const float *__restrict__ qbox,
                                                                const float4 *__restrict__ cs,
                                                                const float *__restrict__ cbox,
                                                                float *__restrict__ dist, int32_t *__restrict__ idx) {
  __shared__ __attribute__((aligned(16))) float s_c[kNNChunk * 3];
  __shared__ int s_k[kNNChunk];
  __shared__ float s_b[4][64];
  __shared__ unsigned long long s_mk[3][3][64];  // partial lists of waves 1..3
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // (the query group is rotated by the cloud: workgroups x, x + 256, ... share a CU -- the same group of every second cloud
  //  -- and clouds of one kind have the same kind of region at the same place of their order: round 6, csrc/knn.hip)
  const int bx = (int)((blockIdx.x + 37u * (unsigned)b) % gridDim.x);
  const int j = bx * 64 + lane;
  const bool valid = j < n;
  const float4 q = qs[(size_t)b * n + (valid ? j : bx * 64)];
  const float4 *cand = cs + (size_t)b * m;
  const f32x2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
  const int len32 = (m + 31) & ~31, nsteps = len32 >> 5;
  for (int e = threadIdx.x; e < len32; e += kBlock) {
    float4 r = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(INT_MAX));  // padding: d = inf, never taken
    if (e < m) r = cand[e];
    s_c[nn_slot(e, 0)] = r.x; s_c[nn_slot(e, 1)] = r.y; s_c[nn_slot(e, 2)] = r.z;
    s_k[e] = __float_as_int(r.w);
  }
  // boxes: the query group's, and (lane = candidate group) the candidates'
  const int NGq = (n + 63) / 64, MG = (m + 63) / 64;
  const float *qb = qbox + ((size_t)b * NGq + bx) * 8;
  const float qlx = qb[0], qly = qb[1], qlz = qb[2], qhx = qb[4], qhy = qb[5], qhz = qb[6];
  float clx = INFINITY, cly = INFINITY, clz = INFINITY, chx = -INFINITY, chy = -INFINITY, chz = -INFINITY, bd = INFINITY;
  if (lane < MG) {
    const float *cb = cbox + ((size_t)b * MG + lane) * 8;
    clx = cb[0]; cly = cb[1]; clz = cb[2]; chx = cb[4]; chy = cb[5]; chz = cb[6];
    const float ex = fmaxf(fmaxf(clx - qhx, qlx - chx), 0.f), ey = fmaxf(fmaxf(cly - qhy, qly - chy), 0.f),
                ez = fmaxf(fmaxf(clz - qhz, qlz - chz), 0.f);
    bd = (ex * ex + ey * ey + ez * ez) * 0.99999f;
  }
  const unsigned long long kInf = (unsigned long long)__float_as_uint(INFINITY) << 32;
  unsigned long long k1 = kInf, k2 = kInf, k3 = kInf;
  float kb = INFINITY;  // the screen: min(own third distance, the shared bound)
  __syncthreads();

  auto scan_step = [&](int step) __attribute__((always_inline)) {
    const int k = step << 5;
    f32x2 d[4][4];
    float mn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float *g0 = s_c + ((k >> 2) + 2 * u) * 12;
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(g0), a1 = *reinterpret_cast<const f32x4 *>(g0 + 4),
                  a2 = *reinterpret_cast<const f32x4 *>(g0 + 8), b0 = *reinterpret_cast<const f32x4 *>(g0 + 12),
                  b1 = *reinterpret_cast<const f32x4 *>(g0 + 16), b2 = *reinterpret_cast<const f32x4 *>(g0 + 20);
      {
        const f32x2 dx = f32x2{a0[0], a0[1]} - qx, dy = f32x2{a0[2], a0[3]} - qy, dz = f32x2{a1[0], a1[1]} - qz;
        d[u][0] = (dx * dx + dy * dy) + dz * dz;
      }
      {
        const f32x2 dx = f32x2{a1[2], a1[3]} - qx, dy = f32x2{a2[0], a2[1]} - qy, dz = f32x2{a2[2], a2[3]} - qz;
        d[u][1] = (dx * dx + dy * dy) + dz * dz;
      }
      {
        const f32x2 dx = f32x2{b0[0], b0[1]} - qx, dy = f32x2{b0[2], b0[3]} - qy, dz = f32x2{b1[0], b1[1]} - qz;
        d[u][2] = (dx * dx + dy * dy) + dz * dz;
      }
      {
        const f32x2 dx = f32x2{b1[2], b1[3]} - qx, dy = f32x2{b2[0], b2[1]} - qy, dz = f32x2{b2[2], b2[3]} - qz;
        d[u][3] = (dx * dx + dy * dy) + dz * dz;
      }
      mn[u] = fminf(fminf(fminf(d[u][0][0], d[u][0][1]), fminf(d[u][1][0], d[u][1][1])),
                    fminf(fminf(d[u][2][0], d[u][2][1]), fminf(d[u][3][0], d[u][3][1])));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // '<=': an equal distance with a smaller index must still be seen
      if (__any(valid && mn[u] <= kb)) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const unsigned long long x = ((unsigned long long)__float_as_uint(d[u][t >> 1][t & 1]) << 32) |
                                       (unsigned)s_k[k + 8 * u + t];
          const bool l1 = x < k1, l2 = x < k2, l3 = x < k3;
          k3 = l2 ? k2 : (l3 ? x : k3);
          k2 = l1 ? k1 : (l2 ? x : k2);
          k1 = l1 ? x : k1;
        }
        kb = fminf(kb, __uint_as_float((unsigned)(k3 >> 32)));
      }
    }
  };

  // A. one step per wave around the rank-proportional start
  long long c0 = ((long long)bx * 64 + 32) * m / n - 64;
  c0 = c0 < 0 ? 0 : c0;
  const int s0 = (int)(c0 >> 5);
  unsigned long long done = 0ull;  // steps of phase A (all four waves')
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int st = (s0 + w) % nsteps;
    if (!((done >> st) & 1ull)) {
      done |= 1ull << st;
      if (w == wave) scan_step(st);
    }
  }
  s_b[wave][lane] = __uint_as_float((unsigned)(k3 >> 32));
  __syncthreads();
  kb = fminf(fminf(s_b[0][lane], s_b[1][lane]), fminf(s_b[2][lane], s_b[3][lane]));
  const float wb = wave_max_f32(valid ? kb : 0.f);

  // B. the candidate groups that can still matter, dealt to the waves
  unsigned long long mask = __ballot(lane < MG && bd <= wb);
  int mine = 0;
  while (mask != 0ull) {
    const int g = __builtin_ctzll(mask);
    mask &= mask - 1;
    // the group's two 32-candidate steps are dealt to the waves ONE BY ONE (round 6: whole groups left a wave up to two steps
    // behind its neighbours, and the workgroup ends with its slowest wave)
    const bool own0 = (mine & 3) == wave, own1 = ((mine + 1) & 3) == wave;
    mine += 2;
    if (!own0 && !own1) continue;
    const float lx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(clx), g)),
                ly = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cly), g)),
                lz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(clz), g)),
                hx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(chx), g)),
                hy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(chy), g)),
                hz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(chz), g));
    const float px = fmaxf(fmaxf(lx - q.x, q.x - hx), 0.f), py = fmaxf(fmaxf(ly - q.y, q.y - hy), 0.f),
                pz = fmaxf(fmaxf(lz - q.z, q.z - hz), 0.f);
    if (!__any(valid && (px * px + py * py + pz * pz) * 0.99999f <= kb)) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int st = 2 * g + h;
      if ((h == 0 ? own0 : own1) && st < nsteps && !((done >> st) & 1ull)) scan_step(st);
    }
  }

  if (wave > 0) { s_mk[wave - 1][0][lane] = k1; s_mk[wave - 1][1][lane] = k2; s_mk[wave - 1][2][lane] = k3; }
  __syncthreads();
  if (wave == 0 && valid) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned long long x = s_mk[w][t][lane];
        const bool l1 = x < k1, l2 = x < k2, l3 = x < k3;
        k3 = l2 ? k2 : (l3 ? x : k3);
        k2 = l1 ? k1 : (l2 ? x : k2);
        k1 = l1 ? x : k1;
      }
    const size_t o = ((size_t)b * n + __float_as_int(q.w)) * 3;
    dist[o] = __uint_as_float((unsigned)(k1 >> 32)); dist[o + 1] = __uint_as_float((unsigned)(k2 >> 32));
    dist[o + 2] = __uint_as_float((unsigned)(k3 >> 32));
    idx[o] = (int)(unsigned)k1; idx[o + 1] = (int)(unsigned)k2; idx[o + 2] = (int)(unsigned)k3;
  }
}

// ------------------------------------------------------------------ three_interpolate
// IDW: weights derived from squared distances (core/backbones.py:92-95); else read from `weight`.
template <bool IDW, int VEC>
__global__ __launch_bounds__(kBlock) void three_interp_fwd_kernel(
    long long rows, int n, int m, int c, const float *__restrict__ points,
    const int32_t *__restrict__ idx, const float *__restrict__ wd, float *__restrict__ out) {
  const int cv = c / VEC;
  const long long total = rows * cv;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const long long row = e / cv;
    const int l = (int)(e - row * cv) * VEC;
    const long long bi = row / n;
    const int i1 = idx[row * 3], i2 = idx[row * 3 + 1], i3 = idx[row * 3 + 2];
    float w1 = wd[row * 3], w2 = wd[row * 3 + 1], w3 = wd[row * 3 + 2];
    if (IDW) {
      const float r1 = 1.0f / fmaxf(w1, 1e-10f), r2 = 1.0f / fmaxf(w2, 1e-10f),
                  r3 = 1.0f / fmaxf(w3, 1e-10f);
      const float norm = (r1 + r2) + r3;
      w1 = r1 / norm; w2 = r2 / norm; w3 = r3 / norm;
    }
    const float *p1 = points + (bi * m + i1) * c + l;
    const float *p2 = points + (bi * m + i2) * c + l;
    const float *p3 = points + (bi * m + i3) * c + l;
    float *o = out + row * c + l;
    if (VEC == 4) {
      const float4 a = *reinterpret_cast<const float4 *>(p1);
      const float4 bq = *reinterpret_cast<const float4 *>(p2);
      const float4 cq = *reinterpret_cast<const float4 *>(p3);
      float4 r;
      r.x = (a.x * w1 + bq.x * w2) + cq.x * w3;
      r.y = (a.y * w1 + bq.y * w2) + cq.y * w3;
      r.z = (a.z * w1 + bq.z * w2) + cq.z * w3;
      r.w = (a.w * w1 + bq.w * w2) + cq.w * w3;
      *reinterpret_cast<float4 *>(o) = r;
    } else {
      o[0] = (p1[0] * w1 + p2[0] * w2) + p3[0] * w3;
    }
  }
}

__global__ __launch_bounds__(kBlock) void three_interp_bwd_kernel(
    long long rows, int n, int m, int c, const float *__restrict__ grad_out,
    const int32_t *__restrict__ idx, const float *__restrict__ weight,
    float *__restrict__ grad_points) {
  const long long total = rows * c;
  for (long long e = (long long)blockIdx.x * kBlock + threadIdx.x; e < total;
       e += (long long)gridDim.x * kBlock) {
    const long long row = e / c;
    const int l = (int)(e - row * c);
    const long long bi = row / n;
    const float g = grad_out[e];
#pragma unroll
    for (int t = 0; t < 3; ++t)
      atomicAdd(&grad_points[(bi * m + idx[row * 3 + t]) * c + l], g * weight[row * 3 + t]);
  }
}

inline int flat_grid(long long total) {
  long long g = (total + kBlock - 1) / kBlock;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

DH3D_API int dh3d_group_point_fwd(int b, int n, int c, int m, int nsample, const float *points,
                                  const int32_t *idx, float *out, void *stream) {
  DH3D_REQUIRE(points && idx && out && b > 0 && n > 0 && c > 0 && m > 0 && nsample > 0);
  const long long rows = (long long)b * m * nsample;
  if (c % 4 == 0)
    hipLaunchKernelGGL(group_point_fwd4_kernel, dim3(flat_grid(rows * (c / 4))), dim3(kBlock), 0, (hipStream_t)stream,
                       rows, n, c / 4, m * nsample, reinterpret_cast<const float4 *>(points), idx,
                       reinterpret_cast<float4 *>(out));
  else
    hipLaunchKernelGGL(group_point_fwd_kernel, dim3(flat_grid(rows * c)), dim3(kBlock), 0,
                       (hipStream_t)stream, rows, n, c, m * nsample, points, idx, out);
  return dh3d_launch_status();
}

DH3D_API int dh3d_group_point_bwd(int b, int n, int c, int m, int nsample, const float *grad_out,
                                  const int32_t *idx, float *grad_points, void *stream) {
  DH3D_REQUIRE(grad_out && idx && grad_points && b > 0 && n > 0 && c > 0 && m > 0 && nsample > 0);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * n * c, s) != hipSuccess)
    return DH3D_ERR_LAUNCH;
  const long long rows = (long long)b * m * nsample;
  hipLaunchKernelGGL(group_point_bwd_kernel, dim3(flat_grid(rows * c)), dim3(kBlock), 0, s, rows, n, c,
                     m * nsample, grad_out, idx, grad_points);
  return dh3d_launch_status();
}

DH3D_API int dh3d_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist,
                           int32_t *idx, void *stream) {
  DH3D_REQUIRE(xyz1 && xyz2 && dist && idx && b > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(b <= 65535);
  hipLaunchKernelGGL(three_nn_kernel, dim3(dh3d_cdiv(n, 64), b), dim3(kBlock), 0,
                     (hipStream_t)stream, n, m, xyz1, xyz2, dist, idx);
  return dh3d_launch_status();
}

DH3D_API int dh3d_three_nn_sorted(int b, int n, int m, const float *sorted1, const float *gbox1,
                                  const float *sorted2, const float *gbox2, float *dist, int32_t *idx,
                                  void *stream) {
  DH3D_REQUIRE(sorted1 && sorted2 && dist && idx && b > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(b <= 65535);
  if (m <= kNNChunk && m >= 128 && gbox1 && gbox2) {  // the sampled set fits LDS: candidate groups pruned by their boxes
    hipLaunchKernelGGL(three_nn_pruned_kernel, dim3(dh3d_cdiv(n, 64), b), dim3(kBlock), 0, (hipStream_t)stream, n, m,
                       reinterpret_cast<const float4 *>(sorted1), gbox1, reinterpret_cast<const float4 *>(sorted2), gbox2,
                       dist, idx);
    return dh3d_launch_status();
  }
  hipLaunchKernelGGL(three_nn_sorted_kernel, dim3(dh3d_cdiv(n, 64), b), dim3(kBlock), 0, (hipStream_t)stream, n, m,
                     reinterpret_cast<const float4 *>(sorted1), reinterpret_cast<const float4 *>(sorted2), dist, idx);
  return dh3d_launch_status();
}

DH3D_API int dh3d_three_interpolate_fwd(int b, int m, int c, int n, const float *points,
                                        const int32_t *idx, const float *weight, float *out,
                                        void *stream) {
  DH3D_REQUIRE(points && idx && weight && out && b > 0 && m > 0 && c > 0 && n > 0);
  const long long rows = (long long)b * n;
  hipStream_t s = (hipStream_t)stream;
  if (c % 4 == 0)
    hipLaunchKernelGGL((three_interp_fwd_kernel<false, 4>), dim3(flat_grid(rows * (c / 4))), dim3(kBlock),
                       0, s, rows, n, m, c, points, idx, weight, out);
  else
    hipLaunchKernelGGL((three_interp_fwd_kernel<false, 1>), dim3(flat_grid(rows * c)), dim3(kBlock), 0, s,
                       rows, n, m, c, points, idx, weight, out);
  return dh3d_launch_status();
}

DH3D_API int dh3d_three_interpolate_idw_fwd(int b, int m, int c, int n, const float *points,
                                            const int32_t *idx, const float *dist, float *out,
                                            void *stream) {
  DH3D_REQUIRE(points && idx && dist && out && b > 0 && m > 0 && c > 0 && n > 0);
  DH3D_SUPPORTED(c % 4 == 0);
  const long long rows = (long long)b * n;
  hipLaunchKernelGGL((three_interp_fwd_kernel<true, 4>), dim3(flat_grid(rows * (c / 4))), dim3(kBlock), 0,
                     (hipStream_t)stream, rows, n, m, c, points, idx, dist, out);
  return dh3d_launch_status();
}

DH3D_API int dh3d_three_interpolate_bwd(int b, int n, int c, int m, const float *grad_out,
                                        const int32_t *idx, const float *weight, float *grad_points,
                                        void *stream) {
  DH3D_REQUIRE(grad_out && idx && weight && grad_points && b > 0 && m > 0 && c > 0 && n > 0);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * m * c, s) != hipSuccess)
    return DH3D_ERR_LAUNCH;
  const long long rows = (long long)b * n;
  hipLaunchKernelGGL(three_interp_bwd_kernel, dim3(flat_grid(rows * c)), dim3(kBlock), 0, s, rows, n, m, c,
                     grad_out, idx, weight, grad_points);
  return dh3d_launch_status();
}
