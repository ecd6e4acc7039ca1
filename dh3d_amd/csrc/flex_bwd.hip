// flex_conv backward in its FACTORISED form (gfx950), and the reference-layout (channels-first) fast paths built on
// the fused point-major kernels.
//
// Forward:  out = S @ Wcat,  S = [S0|Sx|Sy|Sz],  S0[n,i] = sum_k f[nk,i],  Sd[n,i] = sum_k (p[nk,d]-p[c(n),d]) f[nk,i],
//           Wcat = [bias; theta_x; theta_y; theta_z]  ([4*Din, Dout]).
// Backward: dWcat = S^T dOut            -> rows [0,Din) = grad_bias, rows [Din,4Din) = grad_theta   (MFMA, gemm.hip)
//           dS    = dOut Wcat^T         ([R, 4*Din], MFMA)
//           df[nk,i] += dS0[n,i] + sum_d (p[nk,d]-p[c(n),d]) dSd[n,i]                               (f32 atomics)
// -- 2*R*4Din*Dout*2 flop on the matrix pipe instead of the reference formulation's three 9*K*Din*Dout-flop VALU
// kernels (flex_conv_kernel_gpu.cu.cc:168-385), whose feature gradient is the same atomics scatter.
// Centre c(n): the point itself (GPU forward rule, :77-79) or the rank-0 neighbour (both reference backward paths,
// :196-202,314); identical under exact kNN.
#include "internal.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

// S [R, 4*Din] (component-major columns).  One lane = (point, 4 channels).
__global__ __launch_bounds__(256) void flex_S_kernel(const float *__restrict__ feat, const float *__restrict__ xyz,
                                                    const int32_t *__restrict__ nbr, long long R, int N, int K,
                                                    int Din, int rank0, float *__restrict__ S) {
  const int lpr = Din / 4;
  const long long total = R * lpr;
  for (long long e = (long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const long long n = e / lpr;
    const int r4 = (int)(e - n * lpr) * 4;
    const long long cl0 = (n / N) * N;
    const int32_t *nb = nbr + n * K;
    const long long c = rank0 ? cl0 + nb[0] : n;
    const float px = xyz[c * 3], py = xyz[c * 3 + 1], pz = xyz[c * 3 + 2];
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), sx = s0, sy = s0, sz = s0;
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
      const long long g = cl0 + nb[k];
      const float4 f = *reinterpret_cast<const float4 *>(feat + g * Din + r4);
      const float dx = xyz[g * 3] - px, dy = xyz[g * 3 + 1] - py, dz = xyz[g * 3 + 2] - pz;
      s0.x += f.x; s0.y += f.y; s0.z += f.z; s0.w += f.w;
      sx.x = fmaf(dx, f.x, sx.x); sx.y = fmaf(dx, f.y, sx.y); sx.z = fmaf(dx, f.z, sx.z); sx.w = fmaf(dx, f.w, sx.w);
      sy.x = fmaf(dy, f.x, sy.x); sy.y = fmaf(dy, f.y, sy.y); sy.z = fmaf(dy, f.z, sy.z); sy.w = fmaf(dy, f.w, sy.w);
      sz.x = fmaf(dz, f.x, sz.x); sz.y = fmaf(dz, f.y, sz.y); sz.z = fmaf(dz, f.z, sz.z); sz.w = fmaf(dz, f.w, sz.w);
    }
    float *row = S + n * 4 * Din + r4;
    *reinterpret_cast<float4 *>(row) = s0;
    *reinterpret_cast<float4 *>(row + Din) = sx;
    *reinterpret_cast<float4 *>(row + 2 * Din) = sy;
    *reinterpret_cast<float4 *>(row + 3 * Din) = sz;
  }
}

// df[nk, :] += dS0[n, :] + dx dSx[n, :] + dy dSy[n, :] + dz dSz[n, :]
__global__ __launch_bounds__(256) void flex_scatter_kernel(const float *__restrict__ dS, const float *__restrict__ xyz,
                                                          const int32_t *__restrict__ nbr, long long R, int N, int K,
                                                          int Din, int rank0, float *__restrict__ dfeat) {
  const int lpr = Din / 4;
  const long long total = R * lpr;
  for (long long e = (long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const long long n = e / lpr;
    const int r4 = (int)(e - n * lpr) * 4;
    const long long cl0 = (n / N) * N;
    const int32_t *nb = nbr + n * K;
    const long long c = rank0 ? cl0 + nb[0] : n;
    const float px = xyz[c * 3], py = xyz[c * 3 + 1], pz = xyz[c * 3 + 2];
    const float *row = dS + n * 4 * Din + r4;
    const float4 d0 = *reinterpret_cast<const float4 *>(row), d1 = *reinterpret_cast<const float4 *>(row + Din),
                 d2 = *reinterpret_cast<const float4 *>(row + 2 * Din), d3 = *reinterpret_cast<const float4 *>(row + 3 * Din);
    for (int k = 0; k < K; ++k) {
      const long long g = cl0 + nb[k];
      const float dx = xyz[g * 3] - px, dy = xyz[g * 3 + 1] - py, dz = xyz[g * 3 + 2] - pz;
      float *dst = dfeat + g * Din + r4;
      unsafeAtomicAdd(dst, fmaf(dz, d3.x, fmaf(dy, d2.x, fmaf(dx, d1.x, d0.x))));
      unsafeAtomicAdd(dst + 1, fmaf(dz, d3.y, fmaf(dy, d2.y, fmaf(dx, d1.y, d0.y))));
      unsafeAtomicAdd(dst + 2, fmaf(dz, d3.z, fmaf(dy, d2.z, fmaf(dx, d1.z, d0.z))));
      unsafeAtomicAdd(dst + 3, fmaf(dz, d3.w, fmaf(dy, d2.w, fmaf(dx, d1.w, d0.w))));
    }
  }
}

inline int flat_grid256(long long work) {
  long long g = (work + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

DH3D_API size_t dh3d_flex_conv_pm_bwd_workspace_bytes(int B, int N, int Din, int Dout) {
  if (B <= 0 || N <= 0 || Din <= 0 || Dout <= 0 || Din % 4 || Dout % 4) return 0;
  const size_t R = (size_t)B * N;
  return 2 * al256(sizeof(float) * R * 4 * Din) + al256(sizeof(float) * (size_t)Dout * 4 * Din);
}

DH3D_API int dh3d_flex_conv_pm_bwd(const float *features, const float *xyz, const int32_t *nbr, const float *theta,
                                   const float *bias, const float *grad_out, int B, int N, int K, int Din, int Dout,
                                   int center_rank0, void *workspace, size_t workspace_bytes, float *grad_features,
                                   float *grad_theta, float *grad_bias, void *stream) {
  DH3D_REQUIRE(features && xyz && nbr && theta && bias && grad_out && workspace && grad_theta && grad_bias);
  DH3D_REQUIRE(B > 0 && N > 0 && K > 0 && Din > 0 && Dout > 0);
  DH3D_SUPPORTED(Din % 4 == 0 && Dout % 4 == 0);
  DH3D_REQUIRE(workspace_bytes >= dh3d_flex_conv_pm_bwd_workspace_bytes(B, N, Din, Dout));
  hipStream_t s = (hipStream_t)stream;
  const long long R = (long long)B * N;
  const int KD = 4 * Din;
  char *w = static_cast<char *>(workspace);
  float *S = reinterpret_cast<float *>(w);
  float *dS = reinterpret_cast<float *>(w + al256(sizeof(float) * R * KD));
  float *WT = reinterpret_cast<float *>(w + 2 * al256(sizeof(float) * R * KD));  // Wcat^T [Dout, 4*Din]
  hipLaunchKernelGGL(flex_S_kernel, dim3(flat_grid256(R * (Din / 4))), dim3(256), 0, s, features, xyz, nbr, R, N, K, Din,
                     center_rank0, S);
  // dWcat = S^T dOut: rows [0, Din) -> grad_bias, the rest -> grad_theta ([3, Din, Dout] is rows Din.. of Wcat)
  int st = dh3d_internal_gemm(true, S, KD, grad_out, Dout, grad_bias, Dout, KD, Dout, (int)R, grad_theta, Din, false, s);
  if (st != DH3D_OK) return st;
  if (!grad_features) return DH3D_OK;
  st = dh3d_internal_transpose32(bias, WT, 1, Din, Dout, KD, 0, s);                       // columns [0, Din)
  if (st != DH3D_OK) return st;
  st = dh3d_internal_transpose32(theta, WT + Din, 3, Din, Dout, KD, Din, s);              // columns [(1+d)*Din, ...)
  if (st != DH3D_OK) return st;
  st = dh3d_internal_gemm(false, grad_out, Dout, WT, KD, dS, KD, (int)R, KD, Dout, nullptr, 0, false, s);
  if (st != DH3D_OK) return st;
  if (hipMemsetAsync(grad_features, 0, sizeof(float) * R * Din, s) != hipSuccess) return DH3D_ERR_LAUNCH;
  hipLaunchKernelGGL(flex_scatter_kernel, dim3(flat_grid256(R * (Din / 4))), dim3(256), 0, s, dS, xyz, nbr, R, N, K, Din,
                     center_rank0, grad_features);
  return dh3d_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Reference-layout fast paths: user_ops tensors are channels-first ([B,C,N] features, [B,K,N] neighbourhoods,
// [B,3,N] positions).  A neighbour row in that layout is C scattered 4-byte reads (N*4 bytes apart), so the gather
// kernels want point-major rows: the inputs are transposed through LDS tiles into the caller's workspace (HBM-bound,
// ~2*size/5 TB/s each), the fused point-major kernel runs, and the result is transposed back.
// 1 = fused kernel (x6: the persistent bf16x6 one), 2 = any other channel counts that are multiples of four: the
// factorisation in two launches -- S = [S0|Sx|Sy|Sz] materialised by flex_S_kernel, then out = S @ [bias; theta] on the
// GEMM kernels (gemm.hip) -- instead of the reference formulation's 9*K*Din*Dout flop per point on the vector unit
// (flex_generic.hip: 3.4 ms at 64 -> 64, 8 x 8192, where this form takes ~0.1 ms)
static int fast_fwd_shape(int K, int Dp, int Din, int Dout, bool *x6) {
  *x6 = false;
  if (Dp != 3) return 0;
  *x6 = K == 8 && Dout == 64 && (Din == 32 || Din == 64);
  if (*x6) return 1;
  static const int ok[][2] = {{32, 64}, {32, 128}, {64, 64}, {64, 128}, {64, 256}, {128, 128}, {128, 256}};
  for (auto &p : ok)
    if (p[0] == Din && p[1] == Dout) return 1;
  return (Din % 4 == 0 && Dout % 4 == 0) ? 2 : 0;
}

DH3D_API size_t dh3d_flex_conv_fwd_workspace_bytes(int B, int N, int K, int Dp, int Din, int Dout) {
  bool x6;
  const int kind = (B <= 0 || N <= 1 || K <= 0) ? 0 : fast_fwd_shape(K, Dp, Din, Dout, &x6);
  if (kind == 0) return 0;
  const size_t R = (size_t)B * N;
  if (x6 && R * Din * 4 >= (1ull << 32)) return 0;
  if (kind == 2) {  // + S [R, 4*Din] and the concatenated weight
    if (R * 4 * Din >= (1ull << 31)) return 0;
    return al256(4 * R * Din) + al256(4 * R * K) + al256(4 * R * 3) + al256((size_t)4 * 4 * Din * Dout) +
           al256(4 * R * Dout) + al256(4 * R * 4 * Din);
  }
  return al256(4 * R * Din) + al256(4 * R * K) + al256(4 * R * 3) + al256((size_t)6 * 4 * Din * Dout) + al256(4 * R * Dout);
}

DH3D_API int dh3d_flex_conv_fwd_ws(const float *features, const float *theta, const float *bias,
                                   const int32_t *neighborhood, const float *positions, int B, int N, int K, int Dp,
                                   int Din, int Dout, float *output, void *workspace, size_t workspace_bytes,
                                   void *stream) {
  DH3D_REQUIRE(features && theta && bias && neighborhood && positions && output && workspace);
  const size_t need = dh3d_flex_conv_fwd_workspace_bytes(B, N, K, Dp, Din, Dout);
  DH3D_SUPPORTED(need != 0);
  DH3D_REQUIRE(workspace_bytes >= need);
  bool x6;
  const int kind = fast_fwd_shape(K, Dp, Din, Dout, &x6);
  hipStream_t s = (hipStream_t)stream;
  const size_t R = (size_t)B * N;
  char *w = static_cast<char *>(workspace);
  float *f_pm = reinterpret_cast<float *>(w); w += al256(4 * R * Din);
  int32_t *nbr_pm = reinterpret_cast<int32_t *>(w); w += al256(4 * R * K);
  float *xyz_pm = reinterpret_cast<float *>(w); w += al256(4 * R * 3);
  void *wp = w; w += al256((size_t)(kind == 2 ? 4 : 6) * 4 * Din * Dout);
  float *out_pm = reinterpret_cast<float *>(w);
  int st;
  if ((st = dh3d_internal_transpose32(features, f_pm, B, Din, N, 0, 0, s)) != DH3D_OK) return st;
  if ((st = dh3d_internal_transpose32(neighborhood, nbr_pm, B, K, N, 0, 0, s)) != DH3D_OK) return st;
  if ((st = dh3d_internal_transpose32(positions, xyz_pm, B, 3, N, 0, 0, s)) != DH3D_OK) return st;
  if (kind == 2) {
    w += al256(4 * R * Dout);
    float *S = reinterpret_cast<float *>(w);
    float *Wcat = static_cast<float *>(wp);  // [bias; theta_x; theta_y; theta_z]: [4*Din, Dout]
    if (hipMemcpyAsync(Wcat, bias, sizeof(float) * Din * Dout, hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(Wcat + (size_t)Din * Dout, theta, sizeof(float) * 3 * Din * Dout, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return DH3D_ERR_LAUNCH;
    // centre = the point itself (the GPU forward's rule, flex_conv_kernel_gpu.cu.cc:77-79)
    hipLaunchKernelGGL(flex_S_kernel, dim3(flat_grid256((long long)R * (Din / 4))), dim3(256), 0, s, f_pm, xyz_pm, nbr_pm,
                       (long long)R, N, K, Din, 0, S);
    if ((st = dh3d_launch_status()) != DH3D_OK) return st;
    st = dh3d_internal_gemm(false, S, 4 * Din, Wcat, Dout, out_pm, Dout, (int)R, Dout, 4 * Din, nullptr, 0, false, s);
  } else if (x6) {
    if ((st = dh3d_pack_flex_weight_x3(theta, bias, Din, Dout, wp, stream)) != DH3D_OK) return st;
    st = dh3d_flex_conv_pm_x6_fwd(f_pm, xyz_pm, nbr_pm, wp, B, N, K, Din, Dout, nullptr, out_pm, stream);
  } else {
    if ((st = dh3d_pack_flex_weight(theta, bias, Din, Dout, static_cast<float *>(wp), stream)) != DH3D_OK) return st;
    st = dh3d_flex_conv_pm_fwd(f_pm, xyz_pm, nbr_pm, static_cast<const float *>(wp), B, N, K, Din, Dout, nullptr, out_pm,
                               stream);
  }
  if (st != DH3D_OK) return st;
  return dh3d_internal_transpose32(out_pm, output, B, N, Dout, 0, 0, s);
}

DH3D_API size_t dh3d_flex_conv_bwd_workspace_bytes(int B, int N, int K, int Dp, int Din, int Dout) {
  if (B <= 0 || N <= 0 || K <= 0 || Dp != 3 || Din % 4 || Dout % 4) return 0;
  const size_t R = (size_t)B * N;
  return al256(4 * R * Din) * 2 + al256(4 * R * K) + al256(4 * R * 3) + al256(4 * R * Dout) +
         dh3d_flex_conv_pm_bwd_workspace_bytes(B, N, Din, Dout);
}

DH3D_API int dh3d_flex_conv_bwd_ws(const float *features, const float *theta, const float *bias,
                                   const int32_t *neighborhood, const float *positions, const float *topdiff, int B,
                                   int N, int K, int Dp, int Din, int Dout, float *grad_features, float *grad_theta,
                                   float *grad_bias, void *workspace, size_t workspace_bytes, void *stream) {
  DH3D_REQUIRE(features && theta && bias && neighborhood && positions && topdiff && grad_features && grad_theta &&
               grad_bias && workspace);
  const size_t need = dh3d_flex_conv_bwd_workspace_bytes(B, N, K, Dp, Din, Dout);
  DH3D_SUPPORTED(need != 0);
  DH3D_REQUIRE(workspace_bytes >= need);
  hipStream_t s = (hipStream_t)stream;
  const size_t R = (size_t)B * N;
  char *w = static_cast<char *>(workspace);
  float *f_pm = reinterpret_cast<float *>(w); w += al256(4 * R * Din);
  float *df_pm = reinterpret_cast<float *>(w); w += al256(4 * R * Din);
  int32_t *nbr_pm = reinterpret_cast<int32_t *>(w); w += al256(4 * R * K);
  float *xyz_pm = reinterpret_cast<float *>(w); w += al256(4 * R * 3);
  float *g_pm = reinterpret_cast<float *>(w); w += al256(4 * R * Dout);
  int st;
  if ((st = dh3d_internal_transpose32(features, f_pm, B, Din, N, 0, 0, s)) != DH3D_OK) return st;
  if ((st = dh3d_internal_transpose32(neighborhood, nbr_pm, B, K, N, 0, 0, s)) != DH3D_OK) return st;
  if ((st = dh3d_internal_transpose32(positions, xyz_pm, B, 3, N, 0, 0, s)) != DH3D_OK) return st;
  if ((st = dh3d_internal_transpose32(topdiff, g_pm, B, Dout, N, 0, 0, s)) != DH3D_OK) return st;
  // both reference backward paths centre on the rank-0 neighbour (flex_conv_kernel_gpu.cu.cc:196-202,314)
  st = dh3d_flex_conv_pm_bwd(f_pm, xyz_pm, nbr_pm, theta, bias, g_pm, B, N, K, Din, Dout, 1, w,
                             dh3d_flex_conv_pm_bwd_workspace_bytes(B, N, Din, Dout), df_pm, grad_theta, grad_bias, stream);
  if (st != DH3D_OK) return st;
  return dh3d_internal_transpose32(df_pm, grad_features, B, N, Din, 0, 0, s);
}

DH3D_API size_t dh3d_flex_pool_fwd_workspace_bytes(int B, int N, int K, int D) {
  if (B <= 0 || N <= 0 || K <= 0 || D <= 0 || D % 4) return 0;
  const size_t R = (size_t)B * N;
  return al256(4 * R * D) * 3 + al256(4 * R * K);
}

DH3D_API int dh3d_flex_pool_fwd_ws(const float *features, const int32_t *neighborhood, int B, int N, int K, int D,
                                   float *output, int32_t *argmax, void *workspace, size_t workspace_bytes,
                                   void *stream) {
  DH3D_REQUIRE(features && neighborhood && output && argmax && workspace);
  const size_t need = dh3d_flex_pool_fwd_workspace_bytes(B, N, K, D);
  DH3D_SUPPORTED(need != 0);
  DH3D_REQUIRE(workspace_bytes >= need);
  hipStream_t s = (hipStream_t)stream;
  const size_t R = (size_t)B * N;
  char *w = static_cast<char *>(workspace);
  float *f_pm = reinterpret_cast<float *>(w); w += al256(4 * R * D);
  float *o_pm = reinterpret_cast<float *>(w); w += al256(4 * R * D);
  int32_t *a_pm = reinterpret_cast<int32_t *>(w); w += al256(4 * R * D);
  int32_t *nbr_pm = reinterpret_cast<int32_t *>(w);
  int st;
  if ((st = dh3d_internal_transpose32(features, f_pm, B, D, N, 0, 0, s)) != DH3D_OK) return st;
  if ((st = dh3d_internal_transpose32(neighborhood, nbr_pm, B, K, N, 0, 0, s)) != DH3D_OK) return st;
  if ((st = dh3d_flex_pool_pm_fwd(f_pm, nbr_pm, B, N, K, D, o_pm, a_pm, stream)) != DH3D_OK) return st;
  if ((st = dh3d_internal_transpose32(o_pm, output, B, N, D, 0, 0, s)) != DH3D_OK) return st;
  return dh3d_internal_transpose32(a_pm, argmax, B, N, D, 0, 0, s);
}
