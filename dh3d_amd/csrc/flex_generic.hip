// Shape-generic flex operators in the reference's channels-first layouts -- the drop-in entry points
// for the TF custom ops FlexConv / FlexPool / ConvPointset and their gradients.
// These keep the reference's per-(k,din,dout) formulation and summation order
// (user_ops/kernels/flex_conv_kernel_gpu.cu.cc:103-126 etc.), accept any Din/Dout/K, and are what
// the op-level parity tests exercise.  The model path uses the fused point-major kernels of
// flex_pm.hip / dense.hip instead.  Templates over the scalar type: the reference registers float AND double kernels
// (flex_conv_op.cc:97-106) and its gradient tests run in double (test_flex_convolution.py:93-115) -- the *_f64 entry
// points below are these kernels for double.
#include <float.h>

#include <limits>

#include "common.h"

namespace {

constexpr int kPts = 128;  // points per block (one per lane)
constexpr int kDT = 16;    // output channels held in registers per thread
constexpr int kMaxDp = 4;

#define AT3(p, b, c, n, C, N) (p)[((size_t)(b) * (C) + (c)) * (size_t)(N) + (n)]

// ------------------------------------------------------------------ flex_conv forward
template <typename T>
__global__ __launch_bounds__(kPts) void flex_conv_fwd_generic(
    const T *__restrict__ feat, const T *__restrict__ theta, const T *__restrict__ bias,
    const int32_t *__restrict__ nbr, const T *__restrict__ pos, int N, int K, int Dp, int Din,
    int Dout, T *__restrict__ out) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * kPts + threadIdx.x;
  const int o0 = blockIdx.y * kDT;
  if (n >= N) return;
  T res[kDT];
#pragma unroll
  for (int o = 0; o < kDT; ++o) res[o] = T(0);
  T p0[kMaxDp];
  for (int dp = 0; dp < Dp; ++dp) p0[dp] = AT3(pos, b, dp, n, Dp, N);  // centre = point n (gpu.cu.cc:77-79)
  for (int k = 0; k < K; ++k) {
    const int nk = AT3(nbr, b, k, n, K, N);
    T q[kMaxDp];
    for (int dp = 0; dp < Dp; ++dp) q[dp] = AT3(pos, b, dp, nk, Dp, N) - p0[dp];
    for (int i = 0; i < Din; ++i) {
      const T fk = AT3(feat, b, i, nk, Din, N);
#pragma unroll
      for (int o = 0; o < kDT; ++o) {
        if (o0 + o < Dout) {
          T w = T(0);
          for (int dp = 0; dp < Dp; ++dp) w += q[dp] * theta[((size_t)dp * Din + i) * Dout + o0 + o];
          w += bias[(size_t)i * Dout + o0 + o];
          res[o] += w * fk;
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < kDT; ++o)
    if (o0 + o < Dout) AT3(out, b, o0 + o, n, Dout, N) = res[o];
}

// ------------------------------------------------------------------ flex_conv backward
// d features: one thread per (point, input channel tile); scatter with atomics as the reference does
// (gpu.cu.cc:362-364).  Centre = rank-0 neighbour (gpu.cu.cc:314).
template <typename T>
__global__ __launch_bounds__(kPts) void flex_conv_bwd_feat_generic(
    const T *__restrict__ theta, const T *__restrict__ bias, const int32_t *__restrict__ nbr,
    const T *__restrict__ pos, const T *__restrict__ top, int N, int K, int Dp, int Din,
    int Dout, T *__restrict__ gfeat) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * kPts + threadIdx.x;
  const int j = blockIdx.y;
  if (n >= N) return;
  const int c0 = AT3(nbr, b, 0, n, K, N);
  T p0[kMaxDp];
  for (int dp = 0; dp < Dp; ++dp) p0[dp] = AT3(pos, b, dp, c0, Dp, N);
  for (int k = 0; k < K; ++k) {
    const int nk = AT3(nbr, b, k, n, K, N);
    T q[kMaxDp];
    for (int dp = 0; dp < Dp; ++dp) q[dp] = AT3(pos, b, dp, nk, Dp, N) - p0[dp];
    T acc = T(0);
    for (int l = 0; l < Dout; ++l) {
      T w = bias[(size_t)j * Dout + l];
      for (int dp = 0; dp < Dp; ++dp) w += theta[((size_t)dp * Din + j) * Dout + l] * q[dp];
      acc += w * AT3(top, b, l, n, Dout, N);
    }
    atomicAdd(&AT3(gfeat, b, j, nk, Din, N), acc);
  }
}

template <typename T>
__device__ __forceinline__ T block_sum_256(T v, T *s_red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// d theta / d bias: one block per (din j, dout l) pair as gpu.cu.cc:168-248, reduced over (b,n,k).
template <typename T>
__global__ __launch_bounds__(256) void flex_conv_bwd_theta_generic(
    const T *__restrict__ feat, const int32_t *__restrict__ nbr, const T *__restrict__ pos,
    const T *__restrict__ top, int B, int N, int K, int Dp, int Din, int Dout,
    T *__restrict__ gtheta, T *__restrict__ gbias) {
  __shared__ T s_red[4];
  const int l = blockIdx.x, j = blockIdx.y;
  T sb = T(0), st[kMaxDp];
  for (int dp = 0; dp < kMaxDp; ++dp) st[dp] = T(0);
  for (long long e = threadIdx.x; e < (long long)B * N; e += 256) {
    const int b = (int)(e / N), n = (int)(e % N);
    const T t = AT3(top, b, l, n, Dout, N);
    const int c0 = AT3(nbr, b, 0, n, K, N);
    for (int k = 0; k < K; ++k) {
      const int nk = AT3(nbr, b, k, n, K, N);
      const T f = AT3(feat, b, j, nk, Din, N);
      sb += f * t;
      for (int dp = 0; dp < Dp; ++dp)
        st[dp] += f * (AT3(pos, b, dp, nk, Dp, N) - AT3(pos, b, dp, c0, Dp, N)) * t;
    }
  }
  const T rb = block_sum_256(sb, s_red);
  if (threadIdx.x == 0) gbias[(size_t)j * Dout + l] = rb;
  for (int dp = 0; dp < Dp; ++dp) {
    const T r = block_sum_256(st[dp], s_red);
    if (threadIdx.x == 0) gtheta[((size_t)dp * Din + j) * Dout + l] = r;
  }
}

// ------------------------------------------------------------------ flex_pool
template <typename T>
__global__ __launch_bounds__(kPts) void flex_pool_fwd_generic(const T *__restrict__ feat,
                                                             const int32_t *__restrict__ nbr, int N,
                                                             int K, int D, T *__restrict__ out,
                                                             int32_t *__restrict__ argmax) {
  const int b = blockIdx.z, d = blockIdx.y;
  const int n = blockIdx.x * kPts + threadIdx.x;
  if (n >= N) return;
  T best = std::numeric_limits<T>::lowest();  // numeric_limits<T>::lowest(), flex_pool_kernel_gpu.cu.cc:44
  int besti = 0;
  for (int k = 0; k < K; ++k) {
    const int g = AT3(nbr, b, k, n, K, N);
    const T v = AT3(feat, b, d, g, D, N);
    if (best < v) { besti = g; best = v; }
  }
  AT3(out, b, d, n, D, N) = best;
  AT3(argmax, b, d, n, D, N) = besti;
}

template <typename T>
__global__ __launch_bounds__(kPts) void flex_pool_bwd_generic(const T *__restrict__ top,
                                                             const int32_t *__restrict__ argmax, int N,
                                                             int D, T *__restrict__ gfeat) {
  const int b = blockIdx.z, d = blockIdx.y;
  const int n = blockIdx.x * kPts + threadIdx.x;
  if (n >= N) return;
  atomicAdd(&AT3(gfeat, b, d, AT3(argmax, b, d, n, D, N), D, N), AT3(top, b, d, n, D, N));
}

// ------------------------------------------------------------------ conv_pointset
template <typename T>
__global__ __launch_bounds__(kPts) void conv_pointset_fwd_generic(
    const T *__restrict__ feat, const T *__restrict__ theta, const T *__restrict__ bias,
    const int32_t *__restrict__ nbr, int N, int K, int Din, int Dout, T *__restrict__ out) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * kPts + threadIdx.x;
  const int o0 = blockIdx.y * kDT;
  if (n >= N) return;
  T res[kDT];
#pragma unroll
  for (int o = 0; o < kDT; ++o) res[o] = T(0);
  const int n0 = AT3(nbr, b, 0, n, K, N);
  for (int k = 0; k < K; ++k) {
    const int nk = AT3(nbr, b, k, n, K, N);
    for (int i = 0; i < Din; ++i) {
      const T dv = AT3(feat, b, i, nk, Din, N) - AT3(feat, b, i, n0, Din, N);
#pragma unroll
      for (int o = 0; o < kDT; ++o)
        if (o0 + o < Dout) res[o] += theta[(size_t)i * Dout + o0 + o] * dv;
    }
  }
#pragma unroll
  for (int o = 0; o < kDT; ++o)  // bias exactly once (conv_pointset_kernel.cc:60)
    if (o0 + o < Dout) AT3(out, b, o0 + o, n, Dout, N) = res[o] + bias[o0 + o];
}

template <typename T>
__global__ __launch_bounds__(kPts) void conv_pointset_bwd_feat_generic(
    const T *__restrict__ theta, const int32_t *__restrict__ nbr, const T *__restrict__ top,
    int N, int K, int Din, int Dout, T *__restrict__ gfeat) {
  const int b = blockIdx.z;
  const int n = blockIdx.x * kPts + threadIdx.x;
  const int j = blockIdx.y;
  if (n >= N) return;
  T acc = T(0);
  for (int l = 0; l < Dout; ++l) acc += theta[(size_t)j * Dout + l] * AT3(top, b, l, n, Dout, N);
  const int n0 = AT3(nbr, b, 0, n, K, N);
  for (int k = 0; k < K; ++k) {
    const int nk = AT3(nbr, b, k, n, K, N);
    atomicAdd(&AT3(gfeat, b, j, nk, Din, N), acc);
    atomicAdd(&AT3(gfeat, b, j, n0, Din, N), -acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_pointset_bwd_theta_generic(
    const T *__restrict__ feat, const int32_t *__restrict__ nbr, const T *__restrict__ top,
    int B, int N, int K, int Din, int Dout, T *__restrict__ gtheta, T *__restrict__ gbias) {
  __shared__ T s_red[4];
  const int l = blockIdx.x, j = blockIdx.y;
  T st = T(0), sb = T(0);
  for (long long e = threadIdx.x; e < (long long)B * N; e += 256) {
    const int b = (int)(e / N), n = (int)(e % N);
    const T t = AT3(top, b, l, n, Dout, N);
    sb += t;
    const int n0 = AT3(nbr, b, 0, n, K, N);
    const T f0 = AT3(feat, b, j, n0, Din, N);
    for (int k = 0; k < K; ++k) st += (AT3(feat, b, j, AT3(nbr, b, k, n, K, N), Din, N) - f0) * t;
  }
  const T rt = block_sum_256(st, s_red);
  if (threadIdx.x == 0) gtheta[(size_t)j * Dout + l] = rt;
  if (j == 0) {
    const T rb = block_sum_256(sb, s_red);
    if (threadIdx.x == 0) gbias[l] = rb;
  }
}


template <typename T>
int flex_conv_fwd_launch(const T *features, const T *theta, const T *bias, const int32_t *neighborhood,
                         const T *positions, int B, int N, int K, int Dp, int Din, int Dout, T *output, void *stream) {
  DH3D_REQUIRE(features && theta && bias && neighborhood && positions && output);
  DH3D_REQUIRE(B > 0 && N > 0 && K > 0 && Dp > 0 && Din > 0 && Dout > 0);
  DH3D_SUPPORTED(Dp <= kMaxDp && B <= 65535 && dh3d_cdiv(Dout, kDT) <= 65535);
  hipLaunchKernelGGL(flex_conv_fwd_generic<T>, dim3(dh3d_cdiv(N, kPts), dh3d_cdiv(Dout, kDT), B), dim3(kPts), 0,
                     (hipStream_t)stream, features, theta, bias, neighborhood, positions, N, K, Dp, Din, Dout, output);
  return dh3d_launch_status();
}

template <typename T>
int flex_conv_bwd_launch(const T *features, const T *theta, const T *bias, const int32_t *neighborhood,
                         const T *positions, const T *topdiff, int B, int N, int K, int Dp, int Din, int Dout,
                         T *grad_features, T *grad_theta, T *grad_bias, void *stream) {
  DH3D_REQUIRE(features && theta && bias && neighborhood && positions && topdiff && grad_features && grad_theta &&
               grad_bias);
  DH3D_REQUIRE(B > 0 && N > 0 && K > 0 && Dp > 0 && Din > 0 && Dout > 0);
  DH3D_SUPPORTED(Dp <= kMaxDp && B <= 65535 && Din <= 65535 && Dout <= 65535);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_features, 0, sizeof(T) * (size_t)B * Din * N, s) != hipSuccess) return DH3D_ERR_LAUNCH;
  hipLaunchKernelGGL(flex_conv_bwd_feat_generic<T>, dim3(dh3d_cdiv(N, kPts), Din, B), dim3(kPts), 0, s, theta, bias,
                     neighborhood, positions, topdiff, N, K, Dp, Din, Dout, grad_features);
  hipLaunchKernelGGL(flex_conv_bwd_theta_generic<T>, dim3(Dout, Din), dim3(256), 0, s, features, neighborhood, positions,
                     topdiff, B, N, K, Dp, Din, Dout, grad_theta, grad_bias);
  return dh3d_launch_status();
}

template <typename T>
int flex_pool_fwd_launch(const T *features, const int32_t *neighborhood, int B, int N, int K, int D, T *output,
                         int32_t *argmax, void *stream) {
  DH3D_REQUIRE(features && neighborhood && output && argmax && B > 0 && N > 0 && K > 0 && D > 0);
  DH3D_SUPPORTED(B <= 65535 && D <= 65535);
  hipLaunchKernelGGL(flex_pool_fwd_generic<T>, dim3(dh3d_cdiv(N, kPts), D, B), dim3(kPts), 0, (hipStream_t)stream,
                     features, neighborhood, N, K, D, output, argmax);
  return dh3d_launch_status();
}

template <typename T>
int flex_pool_bwd_launch(const T *topdiff, const int32_t *argmax, int B, int N, int D, T *grad_features, void *stream) {
  DH3D_REQUIRE(topdiff && argmax && grad_features && B > 0 && N > 0 && D > 0);
  DH3D_SUPPORTED(B <= 65535 && D <= 65535);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_features, 0, sizeof(T) * (size_t)B * D * N, s) != hipSuccess) return DH3D_ERR_LAUNCH;
  hipLaunchKernelGGL(flex_pool_bwd_generic<T>, dim3(dh3d_cdiv(N, kPts), D, B), dim3(kPts), 0, s, topdiff, argmax, N, D,
                     grad_features);
  return dh3d_launch_status();
}

template <typename T>
int conv_pointset_fwd_launch(const T *features, const T *theta, const T *bias, const int32_t *neighborhood, int B, int N,
                             int K, int Din, int Dout, T *output, void *stream) {
  DH3D_REQUIRE(features && theta && bias && neighborhood && output);
  DH3D_REQUIRE(B > 0 && N > 0 && K > 0 && Din > 0 && Dout > 0);
  DH3D_SUPPORTED(B <= 65535 && dh3d_cdiv(Dout, kDT) <= 65535);
  hipLaunchKernelGGL(conv_pointset_fwd_generic<T>, dim3(dh3d_cdiv(N, kPts), dh3d_cdiv(Dout, kDT), B), dim3(kPts), 0,
                     (hipStream_t)stream, features, theta, bias, neighborhood, N, K, Din, Dout, output);
  return dh3d_launch_status();
}

template <typename T>
int conv_pointset_bwd_launch(const T *features, const T *theta, const int32_t *neighborhood, const T *topdiff, int B,
                             int N, int K, int Din, int Dout, T *grad_features, T *grad_theta, T *grad_bias,
                             void *stream) {
  DH3D_REQUIRE(features && theta && neighborhood && topdiff && grad_features && grad_theta && grad_bias);
  DH3D_REQUIRE(B > 0 && N > 0 && K > 0 && Din > 0 && Dout > 0);
  DH3D_SUPPORTED(B <= 65535 && Din <= 65535 && Dout <= 65535);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_features, 0, sizeof(T) * (size_t)B * Din * N, s) != hipSuccess) return DH3D_ERR_LAUNCH;
  hipLaunchKernelGGL(conv_pointset_bwd_feat_generic<T>, dim3(dh3d_cdiv(N, kPts), Din, B), dim3(kPts), 0, s, theta,
                     neighborhood, topdiff, N, K, Din, Dout, grad_features);
  hipLaunchKernelGGL(conv_pointset_bwd_theta_generic<T>, dim3(Dout, Din), dim3(256), 0, s, features, neighborhood,
                     topdiff, B, N, K, Din, Dout, grad_theta, grad_bias);
  return dh3d_launch_status();
}

}  // namespace

// float32 (the drop-in entry points) and float64 (the reference also registers double kernels; its gradient tests use them)
#define DH3D_FLEX_GENERIC_API(SUFFIX, T)                                                                               \
  DH3D_API int dh3d_flex_conv_fwd##SUFFIX(const T *features, const T *theta, const T *bias, const int32_t *neighborhood, \
                                          const T *positions, int B, int N, int K, int Dp, int Din, int Dout, T *output, \
                                          void *stream) {                                                              \
    return flex_conv_fwd_launch<T>(features, theta, bias, neighborhood, positions, B, N, K, Dp, Din, Dout, output,     \
                                   stream);                                                                            \
  }                                                                                                                    \
  DH3D_API int dh3d_flex_conv_bwd##SUFFIX(const T *features, const T *theta, const T *bias, const int32_t *neighborhood, \
                                          const T *positions, const T *topdiff, int B, int N, int K, int Dp, int Din,  \
                                          int Dout, T *grad_features, T *grad_theta, T *grad_bias, void *stream) {     \
    return flex_conv_bwd_launch<T>(features, theta, bias, neighborhood, positions, topdiff, B, N, K, Dp, Din, Dout,    \
                                   grad_features, grad_theta, grad_bias, stream);                                      \
  }                                                                                                                    \
  DH3D_API int dh3d_flex_pool_fwd##SUFFIX(const T *features, const int32_t *neighborhood, int B, int N, int K, int D,  \
                                          T *output, int32_t *argmax, void *stream) {                                  \
    return flex_pool_fwd_launch<T>(features, neighborhood, B, N, K, D, output, argmax, stream);                        \
  }                                                                                                                    \
  DH3D_API int dh3d_flex_pool_bwd##SUFFIX(const T *topdiff, const int32_t *argmax, int B, int N, int D,                \
                                          T *grad_features, void *stream) {                                            \
    return flex_pool_bwd_launch<T>(topdiff, argmax, B, N, D, grad_features, stream);                                   \
  }                                                                                                                    \
  DH3D_API int dh3d_conv_pointset_fwd##SUFFIX(const T *features, const T *theta, const T *bias,                        \
                                              const int32_t *neighborhood, int B, int N, int K, int Din, int Dout,     \
                                              T *output, void *stream) {                                               \
    return conv_pointset_fwd_launch<T>(features, theta, bias, neighborhood, B, N, K, Din, Dout, output, stream);       \
  }                                                                                                                    \
  DH3D_API int dh3d_conv_pointset_bwd##SUFFIX(const T *features, const T *theta, const int32_t *neighborhood,          \
                                              const T *topdiff, int B, int N, int K, int Din, int Dout,                \
                                              T *grad_features, T *grad_theta, T *grad_bias, void *stream) {           \
    return conv_pointset_bwd_launch<T>(features, theta, neighborhood, topdiff, B, N, K, Din, Dout, grad_features,      \
                                       grad_theta, grad_bias, stream);                                                 \
  }
DH3D_FLEX_GENERIC_API(, float)
DH3D_FLEX_GENERIC_API(_f64, double)
#undef DH3D_FLEX_GENERIC_API
