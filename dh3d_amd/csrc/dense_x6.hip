// f32-accurate GEMM on the bf16 matrix pipe ("bf16x6") for the widest dense layer of the path: the
// per-point attention / detector head 256 -> 1024 -> 1 (core/backbones.py:132-173), 68.7 GFLOP at the
// global-descriptor bench shape -- 0.44 ms at 100 % of the f32 MFMA peak, and the largest single kernel.
//
// gfx950 has no TF32-like mode; the f32 MFMA runs at 1/16 of the bf16 rate.  An f32 value splits EXACTLY
// into three bf16 chunks by truncation,  a = a1 + a2 + a3  (8 + 8 + 8 significand bits; every remainder is
// exact in f32), so  a*b = sum_ij a_i*b_j  with every bf16 x bf16 product exact in the f32 accumulator.
// Keeping the six terms with i + j <= 4 drops only a2*b3 + a3*b2 + a3*b3 <= 2^-23 |a*b| -- below f32
// rounding -- and costs 6 bf16 MFMAs per K=16 instead of 8 f32 MFMAs per K=16: 2.7x less matrix-pipe time
// for a result that matches the f32 chain to ~1e-7 relative (tests compare against float64).  This is not a
// reduced-precision mode: no input bit is discarded before the products are formed.
//
// The k -> (lane group, element) slot assignment inside one v_mfma_f32_32x32x16_bf16 is the same for the A
// and the B operand, and a dot product is invariant under a common permutation of k, so A and B fragments
// are both filled with k = 16*kb + 8*(lane>>5) + j  (j = 0..7) and no hardware slot table is needed.
#include "bf16x3.h"

namespace {

constexpr int kTM = 64;

// W [Kd, Dout] f32 -> packed[((nb*KB + kb)*3 + plane)*64 + lane][8 bf16],
//   element j = chunk_plane( W[16*kb + 8*(lane>>5) + j][32*nb + (lane&31)] )
__global__ __launch_bounds__(256) void pack_weight_x3_kernel(const float *__restrict__ W, int Kd, int Dout,
                                                            unsigned short *__restrict__ packed) {
  const int KB = Kd / 16;
  const long long total = (long long)Kd * Dout;  // one thread per weight
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    const long long blk = e >> 9;  // nb*KB + kb
    const int kb = (int)(blk % KB), nb = (int)(blk / KB);
    const float w = W[(size_t)(kb * 16 + 8 * (lane >> 5) + j) * Dout + nb * 32 + (lane & 31)];
    unsigned c1, c2, c3;
    split3(w, c1, c2, c3);
    const size_t base = ((size_t)blk * 3) * 512 + (size_t)lane * 8 + j;
    packed[base] = (unsigned short)c1;
    packed[base + 512] = (unsigned short)c2;
    packed[base + 1024] = (unsigned short)c3;
  }
}

// att[r] = sigmoid( sum_j act(bn(h[r,:] @ W[:,j] + b[j])) * w_fc[j] + b_fc )
__global__ __launch_bounds__(256) void mlp_head_x6_kernel(const float *__restrict__ h, int C,
                                                         const uint4 *__restrict__ wp, int H, EpilogueArgs ep,
                                                         const float *__restrict__ w_fc, float b_fc,
                                                         long long R, float *__restrict__ att) {
  extern __shared__ __attribute__((aligned(16))) unsigned short s_A[];  // [3][kTM][C+8] bf16 chunks
  __shared__ float s_part[2][kTM];
  const int LD = C + 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long grow0 = (long long)blockIdx.x * kTM;
  // ---- stage 64 rows, splitting every f32 into its three bf16 chunks (once per element)
  const int cv = C / 4;
  for (int e = tid; e < kTM * cv; e += 256) {
    const int p = e / cv, c4 = (e - p * cv) * 4;
    const long long g = grow0 + p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < R) v = *reinterpret_cast<const float4 *>(h + g * C + c4);
    unsigned a1[4], a2[4], a3[4];
    split3(v.x, a1[0], a2[0], a3[0]); split3(v.y, a1[1], a2[1], a3[1]);
    split3(v.z, a1[2], a2[2], a3[2]); split3(v.w, a1[3], a2[3], a3[3]);
    unsigned short *dst = s_A + (size_t)p * LD + c4;
    *reinterpret_cast<uint2 *>(dst) = make_uint2(a1[0] | (a1[1] << 16), a1[2] | (a1[3] << 16));
    *reinterpret_cast<uint2 *>(dst + (size_t)kTM * LD) = make_uint2(a2[0] | (a2[1] << 16), a2[2] | (a2[3] << 16));
    *reinterpret_cast<uint2 *>(dst + (size_t)2 * kTM * LD) = make_uint2(a3[0] | (a3[1] << 16), a3[2] | (a3[3] << 16));
  }
  __syncthreads();

  const int row0 = (wave & 1) * 32;
  const int KB = C / 16, NB = H / 32;
  const unsigned short *arow = s_A + (size_t)(row0 + (lane & 31)) * LD + 8 * (lane >> 5);
  float part[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) part[r] = 0.f;

  for (int cb = (wave >> 1); cb < NB; cb += 4) {  // this wave: column blocks cb and cb+2
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    uint4 bn[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) bn[j][p] = wp[((size_t)((cb + 2 * j) * KB) * 3 + p) * 64 + lane];
    for (int kb = 0; kb < KB; ++kb) {
      uint4 bc[2][3];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          bc[j][p] = bn[j][p];
          if (kb + 1 < KB) bn[j][p] = wp[((size_t)((cb + 2 * j) * KB + kb + 1) * 3 + p) * 64 + lane];
        }
      const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(arow + kb * 16);
      const bf16x8 a2 = *reinterpret_cast<const bf16x8 *>(arow + (size_t)kTM * LD + kb * 16);
      const bf16x8 a3 = *reinterpret_cast<const bf16x8 *>(arow + (size_t)2 * kTM * LD + kb * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, bc[j][0]);
        const bf16x8 b2 = __builtin_bit_cast(bf16x8, bc[j][1]);
        const bf16x8 b3 = __builtin_bit_cast(bf16x8, bc[j][2]);
        // smallest terms first
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = (cb + 2 * j) * 32 + (lane & 31);
      float pb = 0.f, sc = 1.f, sh = 0.f;
      if (ep.pre_bias) pb = ep.pre_bias[col];
      if (ep.scale) sc = ep.scale[col];
      if (ep.shift) sh = ep.shift[col];
      const float wf = w_fc[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) part[r] = fmaf(dh3d_act((acc[j][r] + pb) * sc + sh, ep.act), wf, part[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) part[r] += __shfl_xor(part[r], off, 64);
  }
  if ((lane & 31) == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_part[wave >> 1][row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = part[r];
  }
  __syncthreads();
  if (tid < kTM) {
    const long long g = grow0 + tid;
    if (g < R) att[g] = 1.f / (1.f + expf(-(s_part[0][tid] + s_part[1][tid] + b_fc)));
  }
}

}  // namespace

DH3D_API int dh3d_pack_weight_x3(const float *W, int Kd, int Dout, void *packed, void *stream) {
  DH3D_REQUIRE(W && packed && Kd > 0 && Dout > 0);
  DH3D_SUPPORTED(Kd % 16 == 0 && Dout % 32 == 0);
  hipLaunchKernelGGL(pack_weight_x3_kernel, dim3(dh3d_cdiv((long long)Kd * Dout, 256)), dim3(256), 0,
                     (hipStream_t)stream, W, Kd, Dout, static_cast<unsigned short *>(packed));
  return dh3d_launch_status();
}

DH3D_API int dh3d_mlp_head_pm_x6_fwd(const float *h, int R, int C, const void *wpacked_x3, int H,
                                     const dh3d_epilogue *ep, const float *w_fc, float b_fc, float *att,
                                     void *stream) {
  DH3D_REQUIRE(h && wpacked_x3 && w_fc && att && R > 0 && C > 0 && H > 0);
  DH3D_SUPPORTED(C % 16 == 0 && C <= 384 && H % 128 == 0);
  const size_t lds = sizeof(unsigned short) * 3 * kTM * (C + 8);
  auto kern = mlp_head_x6_kernel;
  DH3D_ALLOW_BIG_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(dh3d_cdiv(R, kTM)), dim3(256), lds, (hipStream_t)stream, h, C,
                     static_cast<const uint4 *>(wpacked_x3), H, dh3d_ep(ep), w_fc, b_fc, (long long)R, att);
  return dh3d_launch_status();
}
