// Brute-force exact kNN for gfx950 -- replaces KnnBruteforceFunctor<GPUDevice,float,int>
// (user_ops/kernels/knn_bruteforce_kernel_gpu.cu.cc:36-134,163-228).
//
// The reference sorts all N (distance, id) pairs per query with cub::BlockRadixSort and keeps K.
// Here one lane owns one query and streams every candidate of its cloud from LDS (all 64 lanes
// read the same float4 -> one broadcast ds_read_b128 per candidate per wave); the running top-K
// lives in registers as a sorted list.  A candidate is first screened on the squared distance
// against a conservative bound derived from the current K-th entry, so the IEEE sqrt and the
// register insertion run only for the few candidates that can enter the list.
//
// Bit-exactness (compiled with -ffp-contract=off; every rounding below is explicit):
//   distance  d = sqrt( fma(dz,dz, fma(dy,dy, dx*dx)) ), dx = c.x - q.x   (gpu.cu.cc:102-107 with
//             nvcc's default fma contraction of `sum += val*val`)
//   order     ascending (d, tb), tb(x) = (x % C_THREADS)*C_VPT + x / C_THREADS  -- the rank of
//             point x in CUB's blocked arrangement, which the stable radix sort preserves among
//             equal keys (gpu.cu.cc:98-123); (C_THREADS, C_VPT) from the N ladder (:181-216).
#include <float.h>
#include <limits.h>

#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kQueriesPerBlock = 256;
constexpr int kChunk = 1024;  // candidates staged per LDS round (16 KiB as float4)

struct KnnLadder {
  int log2ct;  // log2(C_THREADS)
  int ctmask;  // C_THREADS-1
  int cv;      // C_VPT
};

static KnnLadder knn_ladder(int N) {
  int t, v;
  if (N <= 32) { t = 32; v = 1; }
  else if (N <= 64) { t = 64; v = 1; }
  else if (N <= 128) { t = 128; v = 1; }
  else if (N <= 256) { t = 128; v = 2; }
  else if (N <= 512) { t = 128; v = 4; }
  else if (N <= 1024) { t = 256; v = 4; }
  else if (N <= 2048) { t = 256; v = 8; }
  else if (N <= 4096) { t = 512; v = 8; }
  else if (N <= 8192) { t = 1024; v = 8; }
  else { t = 1024; v = (N + 1023) / 1024; }  // superset: upstream stops at 8192
  KnnLadder l;
  l.ctmask = t - 1;
  l.cv = v;
  l.log2ct = 0;
  while ((1 << l.log2ct) < t) ++l.log2ct;
  return l;
}

__device__ __forceinline__ bool knn_less(float d0, int t0, float d1, int t1) {
  return d0 < d1 || (d0 == d1 && t0 < t1);
}

// XYZ_LAYOUT: false = positions [B,3,N] (op layout), true = xyz [B,N,3].
template <int KMAX, bool XYZ_LAYOUT>
__global__ __launch_bounds__(kQueriesPerBlock) void knn_kernel(const float *__restrict__ pos, int N,
                                                              int K, KnnLadder lad,
                                                              int32_t *__restrict__ nn,
                                                              float *__restrict__ dist) {
  __shared__ float4 s_c[kChunk];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int y = blockIdx.x * kQueriesPerBlock + tid;
  const float *pc = pos + (size_t)b * 3 * N;

  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (y < N) {
    if (XYZ_LAYOUT) { qx = pc[(size_t)y * 3]; qy = pc[(size_t)y * 3 + 1]; qz = pc[(size_t)y * 3 + 2]; }
    else { qx = pc[y]; qy = pc[(size_t)N + y]; qz = pc[(size_t)2 * N + y]; }
  }

  float bd[KMAX];
  int bt[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) { bd[i] = INFINITY; bt[i] = INT_MAX; }
  float bound = INFINITY;  // s > bound  =>  sqrt(s) > bd[KMAX-1]

  for (int base = 0; base < N; base += kChunk) {
    const int len = min(kChunk, N - base);
    __syncthreads();
    if (XYZ_LAYOUT) {
      for (int e = tid; e < len * 3; e += kQueriesPerBlock) {
        float v = pc[(size_t)base * 3 + e];
        reinterpret_cast<float *>(s_c)[(e / 3) * 4 + (e % 3)] = v;
      }
    } else {
      for (int e = tid; e < len; e += kQueriesPerBlock) {
        s_c[e] = make_float4(pc[base + e], pc[(size_t)N + base + e], pc[(size_t)2 * N + base + e], 0.f);
      }
    }
    __syncthreads();
    if (y < N) {
#pragma unroll 8
      for (int j = 0; j < len; ++j) {
        const float4 c = s_c[j];
        const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
        const float s = __fmaf_rn(dz, dz, __fmaf_rn(dy, dy, __fmul_rn(dx, dx)));
        if (s <= bound) {
          const float d = sqrtf(s);  // IEEE-rounded (llvm.sqrt.f32; -fhip-fp32-correctly-rounded-divide-sqrt default)
          const int x = base + j;
          const int tb = (x & lad.ctmask) * lad.cv + (x >> lad.log2ct);
          if (knn_less(d, tb, bd[KMAX - 1], bt[KMAX - 1])) {
            bd[KMAX - 1] = d;
            bt[KMAX - 1] = tb;
#pragma unroll
            for (int i = KMAX - 1; i > 0; --i) {
              const bool lt = knn_less(bd[i], bt[i], bd[i - 1], bt[i - 1]);
              const float d_hi = lt ? bd[i - 1] : bd[i];
              const float d_lo = lt ? bd[i] : bd[i - 1];
              const int t_hi = lt ? bt[i - 1] : bt[i];
              const int t_lo = lt ? bt[i] : bt[i - 1];
              bd[i] = d_hi; bd[i - 1] = d_lo;
              bt[i] = t_hi; bt[i - 1] = t_lo;
            }
            // (1+2^-20)-inflated square of the K-th distance: any s above it has sqrt(s) > bd[K-1].
            bound = __fmul_rn(__fmul_rn(bd[KMAX - 1], bd[KMAX - 1]), 1.000001f);
          }
        }
      }
    }
  }

  if (y < N) {
    int32_t *o_nn = nn + ((size_t)b * N + y) * K;
    float *o_d = dist + ((size_t)b * N + y) * K;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      if (i < K) {
        const int tb = bt[i];
        if (tb == INT_MAX) {  // fewer than K points: reference pads with id -1 / FLT_MAX (:110-111)
          o_nn[i] = -1;
          o_d[i] = FLT_MAX;
        } else {
          o_nn[i] = ((tb % lad.cv) << lad.log2ct) + tb / lad.cv;
          o_d[i] = bd[i];
        }
      }
    }
  }
}

template <bool XYZ>
int knn_launch(const float *pos, int B, int N, int K, int32_t *nn, float *dist, hipStream_t s) {
  const KnnLadder lad = knn_ladder(N);
  dim3 grid(dh3d_cdiv(N, kQueriesPerBlock), B), block(kQueriesPerBlock);
  if (K <= 4) hipLaunchKernelGGL((knn_kernel<4, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  else if (K <= 8) hipLaunchKernelGGL((knn_kernel<8, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  else if (K <= 16) hipLaunchKernelGGL((knn_kernel<16, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  else if (K <= 32) hipLaunchKernelGGL((knn_kernel<32, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  else hipLaunchKernelGGL((knn_kernel<64, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  return dh3d_launch_status();
}

}  // namespace

DH3D_API int dh3d_knn_bruteforce(const float *positions, int B, int Dp, int N, int K, int32_t *nn,
                                 float *dist, void *stream) {
  DH3D_REQUIRE(positions && nn && dist && B > 0 && N > 0 && K > 0 && Dp > 0);
  DH3D_SUPPORTED(Dp == 3 && K <= 64 && B <= 65535);
  return knn_launch<false>(positions, B, N, K, nn, dist, (hipStream_t)stream);
}

DH3D_API int dh3d_knn_bruteforce_xyz(const float *xyz, int B, int N, int K, int32_t *nn, float *dist,
                                     void *stream) {
  DH3D_REQUIRE(xyz && nn && dist && B > 0 && N > 0 && K > 0);
  DH3D_SUPPORTED(K <= 64 && B <= 65535);
  return knn_launch<true>(xyz, B, N, K, nn, dist, (hipStream_t)stream);
}
