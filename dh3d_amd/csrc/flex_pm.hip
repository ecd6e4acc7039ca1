// Fused point-major flex operators for gfx950 (the model path).
//
// flex_conv (reference: user_ops/kernels/flex_conv_kernel_gpu.cu.cc:46-158) computes
//     out[n,o] = sum_k sum_i ( bias[i,o] + sum_d theta[d,i,o] * (p[nk,d]-p[n,d]) ) * f[nk,i].
// The reference synthesises a weight per (k,i,o): 9 flop per MAC, f32-VALU bound by 17-80x over its
// memory time.  Pulling the k-sum inside gives the exact algebraic factorisation
//     out[n,:] = [S0 | Sx | Sy | Sz][n,:] @ [bias; theta_x; theta_y; theta_z]
//     S0[n,i] = sum_k f[nk,i]      Sd[n,i] = sum_k (p[nk,d]-p[n,d]) * f[nk,i]
// i.e. a K-neighbour gather-reduce (memory shaped: each neighbour row f[nk,:] is one contiguous
// 4*Din-byte read in the point-major layout) followed by a [TM, 4*Din] x [4*Din, Dout] GEMM on the
// exact-f32 MFMA pipe.  One kernel: phase A fills the S tile in LDS, phase B runs the GEMM from LDS
// and applies feature_bias + BatchNorm + activation in the store.  Workgroups are tiles of
// consecutive points of one cloud, so an XCD's L2 sees one cloud's features at a time.
#include <float.h>

#include "mfma_gemm.h"

namespace {

// ------------------------------------------------------------------ flex_conv
template <int DIN, int DOUT, int TMSEL = 0>
struct FlexCfg {
  // points per workgroup: 64 for the narrow inputs, 32 for Din = 128; TMSEL = 32 forces the small tile (launches with
  // fewer 64-point tiles than CUs: the N/8 levels)
  static constexpr int TM = TMSEL ? TMSEL : (DIN <= 64 ? 64 : 32);
  static constexpr int KD = 4 * DIN;                // GEMM depth
  static constexpr int LD = KD + 4;                 // LDS leading dimension (floats)
  static constexpr int MB = TM / 32;                // 32-row blocks
  static constexpr int NB = DOUT / 32;              // 32-col blocks
  static constexpr int NT = NB * MB / 4;            // tiles per wave
  static constexpr int LPR = DIN / 4;               // lanes per gathered row (float4 each)
  static constexpr int PPR = 256 / LPR;             // points per gather round
  static constexpr int ROUNDS = TM / PPR;
  static_assert(NT >= 1 && NB * MB % 4 == 0, "tile must split over 4 waves");
};

#ifdef DH3D_FLEX_PROBE  // dev instrumentation (tools/flex_probe.py): cycle stamps of a few workgroups
__device__ long long g_fprobe[64 * 8];
#define FPROBE(i) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_fprobe[blockIdx.x * 8 + (i)] = clock64(); } while (0)
#else
#define FPROBE(i) do { } while (0)
#endif

// KT = compile-time neighbourhood size (8 on the DH3D path) or 0 for a run-time K.
// POST: the finished output tile goes through one more linear layer [DOUT -> 64] (wpost = dh3d_pack_weight of [DOUT, 64], no
// bias / activation) before it leaves the chip, both results stored: the cluster logits `coarse @ cluster_weights` of
// NetVLAD behind the global flex_conv (core/backbones.py:213-216 on the commuted form) -- a 17 us launch + its gap on
// the global step's critical chain become ~12 % more MFMA work in this kernel.
template <int DIN, int DOUT, int KT, int TMSEL = 0, bool POST = false>
__global__ __launch_bounds__(256) void flex_conv_pm_kernel(
    const float *__restrict__ feat, const float *__restrict__ xyz, const int32_t *__restrict__ nbr,
    const float *__restrict__ wpacked, long long R, int N, int K, EpilogueArgs ep,
    float *__restrict__ out, const int32_t *__restrict__ remap, int Nsrc, const float *__restrict__ wpost = nullptr,
    float *__restrict__ out2 = nullptr) {
  // remap (may be NULL): the features are rows of a LARGER per-cloud map [B, Nsrc, DIN] and point j of this level is
  // row remap[b*N + j] of it -- group_point (the sampled level's feature gather, core/tf_utils.py:92-95) fused into the
  // neighbour gather: one more dependent index load instead of a kernel + its dependency gap
  using C = FlexCfg<DIN, DOUT, TMSEL>;
  extern __shared__ __attribute__((aligned(16))) float s_S[];  // [TM][LD]
  const int tid = threadIdx.x;
  const long long grow0 = (long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * C::TM;

  // ---- phase A: gather-reduce S = [S0|Sx|Sy|Sz] for TM points
  FPROBE(0);
  const int r4 = (tid % C::LPR) * 4;
  if (KT > 0) {
    // Two dependent memory round trips for the whole tile instead of ~5 per round: (1) every round's
    // neighbour ids, (2) every neighbour row + coordinate of two rounds at a time, all in flight together.
    // (The round-by-round form spent ~16k cycles per tile waiting on vmcnt(0): profiles/r01_e.)
    constexpr int KK = KT > 0 ? KT : 1;
    constexpr int HF = KK <= 8 ? 2 : 1;  // rounds whose neighbour rows are in flight together (registers: 4 * KK per round)
    int nid[C::ROUNDS][KK];
    float pxyz[C::ROUNDS][3];
    long long cloud0[C::ROUNDS];
    bool ok[C::ROUNDS];
#pragma unroll
    for (int rd = 0; rd < C::ROUNDS; ++rd) {
      const long long n = grow0 + rd * C::PPR + tid / C::LPR;
      ok[rd] = n < R;
      const long long nn = ok[rd] ? n : 0;
      cloud0[rd] = (nn / N) * N;
      const int4 *ip = reinterpret_cast<const int4 *>(nbr + nn * KK);
#pragma unroll
      for (int q = 0; q < KK / 4; ++q) {
        const int4 v = ip[q];
        nid[rd][4 * q] = v.x; nid[rd][4 * q + 1] = v.y; nid[rd][4 * q + 2] = v.z; nid[rd][4 * q + 3] = v.w;
      }
      pxyz[rd][0] = xyz[nn * 3]; pxyz[rd][1] = xyz[nn * 3 + 1]; pxyz[rd][2] = xyz[nn * 3 + 2];
    }
#pragma unroll
    for (int rp = 0; rp < C::ROUNDS; rp += HF) {
      float4 fv[HF][KK];
      float qv[HF][KK][3];
      long long gf[HF][KK];
#pragma unroll
      for (int h = 0; h < HF; ++h)
#pragma unroll
        for (int k = 0; k < KK; ++k) {
          const long long g = cloud0[rp + h] + nid[rp + h][k];
          gf[h][k] = remap ? (cloud0[rp + h] / N) * Nsrc + remap[g] : g;
        }
#pragma unroll
      for (int h = 0; h < HF; ++h)
#pragma unroll
        for (int k = 0; k < KK; ++k) {
          const long long g = cloud0[rp + h] + nid[rp + h][k];
          fv[h][k] = *reinterpret_cast<const float4 *>(feat + gf[h][k] * DIN + r4);
          qv[h][k][0] = xyz[g * 3]; qv[h][k][1] = xyz[g * 3 + 1]; qv[h][k][2] = xyz[g * 3 + 2];
        }
#pragma unroll
      for (int h = 0; h < HF; ++h) {
        const int rd = rp + h;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), sx = s0, sy = s0, sz = s0;
#pragma unroll
        for (int k = 0; k < KK; ++k) {
          const float4 f = fv[h][k];
          const float dx = qv[h][k][0] - pxyz[rd][0], dy = qv[h][k][1] - pxyz[rd][1], dz = qv[h][k][2] - pxyz[rd][2];
          s0.x += f.x; s0.y += f.y; s0.z += f.z; s0.w += f.w;
          sx.x = fmaf(dx, f.x, sx.x); sx.y = fmaf(dx, f.y, sx.y); sx.z = fmaf(dx, f.z, sx.z); sx.w = fmaf(dx, f.w, sx.w);
          sy.x = fmaf(dy, f.x, sy.x); sy.y = fmaf(dy, f.y, sy.y); sy.z = fmaf(dy, f.z, sy.z); sy.w = fmaf(dy, f.w, sy.w);
          sz.x = fmaf(dz, f.x, sz.x); sz.y = fmaf(dz, f.y, sz.y); sz.z = fmaf(dz, f.z, sz.z); sz.w = fmaf(dz, f.w, sz.w);
        }
        if (!ok[rd]) { s0 = make_float4(0.f, 0.f, 0.f, 0.f); sx = s0; sy = s0; sz = s0; }
        float *row = s_S + (size_t)(rd * C::PPR + tid / C::LPR) * C::LD + r4;
        *reinterpret_cast<float4 *>(row) = s0;
        *reinterpret_cast<float4 *>(row + DIN) = sx;
        *reinterpret_cast<float4 *>(row + 2 * DIN) = sy;
        *reinterpret_cast<float4 *>(row + 3 * DIN) = sz;
      }
    }
  } else {
#pragma unroll
    for (int rd = 0; rd < C::ROUNDS; ++rd) {
      const int p = rd * C::PPR + tid / C::LPR;
      const long long n = grow0 + p;
      float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), sx = s0, sy = s0, sz = s0;
      if (n < R) {
        const long long cl0 = (n / N) * N;
        const float px = xyz[n * 3], py = xyz[n * 3 + 1], pz = xyz[n * 3 + 2];
        const int32_t *nb = nbr + n * K;
#pragma unroll 4
        for (int k = 0; k < K; ++k) {
          const long long g = cl0 + nb[k];
          const long long gfk = remap ? (cl0 / N) * Nsrc + remap[g] : g;
          const float4 f = *reinterpret_cast<const float4 *>(feat + gfk * DIN + r4);
          const float dx = xyz[g * 3] - px, dy = xyz[g * 3 + 1] - py, dz = xyz[g * 3 + 2] - pz;
          s0.x += f.x; s0.y += f.y; s0.z += f.z; s0.w += f.w;
          sx.x = fmaf(dx, f.x, sx.x); sx.y = fmaf(dx, f.y, sx.y); sx.z = fmaf(dx, f.z, sx.z); sx.w = fmaf(dx, f.w, sx.w);
          sy.x = fmaf(dy, f.x, sy.x); sy.y = fmaf(dy, f.y, sy.y); sy.z = fmaf(dy, f.z, sy.z); sy.w = fmaf(dy, f.w, sy.w);
          sz.x = fmaf(dz, f.x, sz.x); sz.y = fmaf(dz, f.y, sz.y); sz.z = fmaf(dz, f.z, sz.z); sz.w = fmaf(dz, f.w, sz.w);
        }
      }
      float *row = s_S + (size_t)p * C::LD + r4;
      *reinterpret_cast<float4 *>(row) = s0;
      *reinterpret_cast<float4 *>(row + DIN) = sx;
      *reinterpret_cast<float4 *>(row + 2 * DIN) = sy;
      *reinterpret_cast<float4 *>(row + 3 * DIN) = sz;
    }
  }
  FPROBE(1);
  __syncthreads();
  FPROBE(2);

  // ---- phase B: S @ Wcat on the f32 MFMA pipe, epilogue in the store
  const int wave = tid >> 6;
  const int row0 = (C::MB == 2) ? (wave & 1) * 32 : 0;
  const int cb0 = (C::MB == 2) ? (wave >> 1) : wave;
  constexpr int cbstride = (C::MB == 2) ? 2 : 4;
  f32x16 acc[C::NT];
  zero_acc<C::NT>(acc);
  EpilogueRegs er[C::NT];
#pragma unroll
  for (int j = 0; j < C::NT; ++j) er[j] = epilogue_prefetch(ep, (cb0 + j * cbstride) * 32 + (tid & 31));
  wave_gemm_f32<C::NT>(s_S, C::LD, row0, wpacked, C::KD / 8, cb0, cbstride, acc);
#ifdef DH3D_FLEX_PROBE
  asm volatile("" :: "v"(acc[0][0]));
#endif
  FPROBE(3);
  // wide epilogue through the (now dead) S tile: LD = 4*Din + 4 >= Dout + 4 for every supported shape
  static_assert(C::LD >= DOUT + 4, "output tile must fit the S tile");
  __syncthreads();
  wave_tiles_to_lds<C::NT>(acc, er, ep.act, s_S, C::LD, row0, cb0, cbstride);
  __syncthreads();
  block_store_rows(s_S, C::LD, C::TM, grow0, R, DOUT, nullptr, out);
  if (POST) {
    static_assert(!POST || (C::TM == 32 && C::LD >= DOUT + 64 + 4), "POST: 32-point tiles, room for 64 more columns");
    if (wave < 2) {  // [32, DOUT] x [DOUT, 64]: one 32-column block per wave, the tile in LDS is the A operand
      f32x16 pacc[1];
      zero_acc<1>(pacc);
      wave_gemm_f32<1>(s_S, C::LD, 0, wpost, DOUT / 8, wave, 1, pacc);
      const EpilogueRegs none[1] = {EpilogueRegs{0.f, 1.f, 0.f}};
      wave_tiles_to_lds<1>(pacc, none, DH3D_ACT_NONE, s_S + DOUT, C::LD, 0, wave, 1);  // columns [DOUT, DOUT + 64) of the rows
    }
    __syncthreads();
    block_store_rows(s_S + DOUT, C::LD, C::TM, grow0, R, 64, nullptr, out2);
  }
  FPROBE(4);
}

template <int DIN, int DOUT>
int flex_conv_pm_launch(const float *feat, const float *xyz, const int32_t *nbr, const float *wpacked,
                        int B, int N, int K, const EpilogueArgs &ep, float *out, hipStream_t s,
                        const int32_t *remap, int Nsrc) {
  using C = FlexCfg<DIN, DOUT>;
  const long long R = (long long)B * N;
  if constexpr (DIN == 64 && DOUT >= 128) {
    // fewer 64-point tiles than CUs (the N/8 levels): 32-point tiles keep the whole chip busy
    if (K == 8 && R <= 64 * 256) {
      using C32 = FlexCfg<DIN, DOUT, 32>;
      auto kern = flex_conv_pm_kernel<DIN, DOUT, 8, 32>;
      DH3D_ALLOW_BIG_LDS(kern);
      hipLaunchKernelGGL(kern, dim3(dh3d_cdiv(R, 32)), dim3(256), sizeof(float) * 32 * C32::LD, s, feat, xyz, nbr,
                         wpacked, R, N, K, ep, out, remap, Nsrc, (const float *)nullptr, (float *)nullptr);
      return dh3d_launch_status();
    }
  }
  const size_t lds = sizeof(float) * C::TM * C::LD;
  const dim3 grid(dh3d_cdiv(R, C::TM)), block(256);
  if (K == 8) {
    auto kern = flex_conv_pm_kernel<DIN, DOUT, 8>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, grid, block, lds, s, feat, xyz, nbr, wpacked, R, N, K, ep, out, remap, Nsrc, (const float *)nullptr, (float *)nullptr);
  } else if (K == 12 && DIN == 128 && DOUT == 128) {
    // BASELINE config 5's stress kernel (localdesc_extract.py:146,166: K = 12 on 128-d features): the two-round-trip
    // gather with a compile-time K instead of the run-time-K loop (a dependent id -> row load chain per neighbour)
    auto kern = flex_conv_pm_kernel<DIN, DOUT, 12>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, grid, block, lds, s, feat, xyz, nbr, wpacked, R, N, K, ep, out, remap, Nsrc, (const float *)nullptr, (float *)nullptr);
  } else {
    auto kern = flex_conv_pm_kernel<DIN, DOUT, 0>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, grid, block, lds, s, feat, xyz, nbr, wpacked, R, N, K, ep, out, remap, Nsrc, (const float *)nullptr, (float *)nullptr);
  }
  return dh3d_launch_status();
}

// ------------------------------------------------------------------ flex_pool
__global__ __launch_bounds__(256) void flex_pool_pm_kernel(const float *__restrict__ feat,
                                                          const int32_t *__restrict__ nbr, long long R,
                                                          int N, int K, int C, float *__restrict__ out,
                                                          int32_t *__restrict__ argmax) {
  const int cv = C / 4;
  const long long total = R * cv;
  for (long long e = (long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const long long n = e / cv;
    const int c4 = (int)(e - n * cv) * 4;
    const long long cloud0 = (n / N) * N;
    const int32_t *nb = nbr + n * K;
    float4 best = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    int4 bi = make_int4(0, 0, 0, 0);
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
      const int g = nb[k];
      const float4 v = *reinterpret_cast<const float4 *>(feat + (cloud0 + g) * C + c4);
      if (best.x < v.x) { best.x = v.x; bi.x = g; }
      if (best.y < v.y) { best.y = v.y; bi.y = g; }
      if (best.z < v.z) { best.z = v.z; bi.z = g; }
      if (best.w < v.w) { best.w = v.w; bi.w = g; }
    }
    *reinterpret_cast<float4 *>(out + n * C + c4) = best;
    if (argmax) *reinterpret_cast<int4 *>(argmax + n * C + c4) = bi;
  }
}

// ------------------------------------------------------------------ flex_avg (neighbour sum)
// Flex_Avg (core/layers.py:342-436) is flex_conv with theta = 0 (non-trainable) and bias = eye(Dout): every other
// term of FlexConv's sum is an exact 0*f, what is left is out[n,c] = sum_k f[nbr[n,k],c] in neighbour order; the
// caller's factor (backbones.py:82: 1/knn) is applied to the finished sum, as upstream.
__global__ __launch_bounds__(256) void flex_avg_pm_kernel(const float *__restrict__ feat,
                                                         const int32_t *__restrict__ nbr, long long R, int N,
                                                         int K, int C, float scale, float *__restrict__ out) {
  const int cv = C / 4;
  const long long total = R * cv;
  for (long long e = (long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const long long n = e / cv;
    const int c4 = (int)(e - n * cv) * 4;
    const long long cloud0 = (n / N) * N;
    const int32_t *nb = nbr + n * K;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
      const float4 v = *reinterpret_cast<const float4 *>(feat + (cloud0 + nb[k]) * C + c4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
    *reinterpret_cast<float4 *>(out + n * C + c4) = acc;
  }
}

// ------------------------------------------------------------------ conv_pointset on coordinates
// out[n,o] = sum_k sum_i theta[i,o]*(x[nk,i]-x[n0,i]) + bias[o]  (conv_pointset_kernel.cc:46-64), Din=3.
// One lane per point: the K neighbour offsets are gathered ONCE (the first version used Dout/4 lanes per point,
// each re-gathering them: 280 loads per point, load-issue bound); the 32 outputs are produced four at a time with
// the same per-k fma chain as before, and leave through LDS so that the stores are full 128-byte rows.
template <int KT>
__global__ __launch_bounds__(256) void conv_pointset_pm_kernel(
    const float *__restrict__ xyz, const int32_t *__restrict__ nbr, const float *__restrict__ theta,
    const float *__restrict__ bias, long long R, int N, int K, int Dout, EpilogueArgs ep,
    float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_o[];  // [256][Dout + 4]
  const int LDO = Dout + 4;
  const int tid = threadIdx.x;
  const long long n0 = (long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * 256;
  const long long n = n0 + tid;
  constexpr int KK = KT > 0 ? KT : 1;
  float dx[KK], dy[KK], dz[KK];
  if (n < R) {
    const long long cloud0 = (n / N) * N;
    const int32_t *nb = nbr + n * KK;
    const long long g0 = cloud0 + nb[0];
    const float x0 = xyz[g0 * 3], y0 = xyz[g0 * 3 + 1], z0 = xyz[g0 * 3 + 2];
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      const long long g = cloud0 + nb[k];
      dx[k] = xyz[g * 3] - x0; dy[k] = xyz[g * 3 + 1] - y0; dz[k] = xyz[g * 3 + 2] - z0;
    }
  } else {
#pragma unroll
    for (int k = 0; k < KK; ++k) dx[k] = dy[k] = dz[k] = 0.f;
  }
  for (int o4 = 0; o4 < Dout; o4 += 4) {  // theta / bias: uniform addresses -> scalar loads
    const float4 tx = *reinterpret_cast<const float4 *>(theta + o4);
    const float4 ty = *reinterpret_cast<const float4 *>(theta + Dout + o4);
    const float4 tz = *reinterpret_cast<const float4 *>(theta + 2 * Dout + o4);
    const Ep4 q = ep4_prefetch(ep, o4);  // requested with theta, not one dependent load per channel behind the sum
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      acc.x = fmaf(tz.x, dz[k], fmaf(ty.x, dy[k], fmaf(tx.x, dx[k], acc.x)));
      acc.y = fmaf(tz.y, dz[k], fmaf(ty.y, dy[k], fmaf(tx.y, dx[k], acc.y)));
      acc.z = fmaf(tz.z, dz[k], fmaf(ty.z, dy[k], fmaf(tx.z, dx[k], acc.z)));
      acc.w = fmaf(tz.w, dz[k], fmaf(ty.w, dy[k], fmaf(tx.w, dx[k], acc.w)));
    }
    const float4 bq = *reinterpret_cast<const float4 *>(bias + o4);
    float4 r;
    r.x = dh3d_act(((acc.x + bq.x) + q.pb.x) * q.sc.x + q.sh.x, ep.act);
    r.y = dh3d_act(((acc.y + bq.y) + q.pb.y) * q.sc.y + q.sh.y, ep.act);
    r.z = dh3d_act(((acc.z + bq.z) + q.pb.z) * q.sc.z + q.sh.z, ep.act);
    r.w = dh3d_act(((acc.w + bq.w) + q.pb.w) * q.sc.w + q.sh.w, ep.act);
    *reinterpret_cast<float4 *>(s_o + (size_t)tid * LDO + o4) = r;
  }
  __syncthreads();
  const int cv = Dout / 4;
  for (int e = tid; e < 256 * cv; e += 256) {
    const int p = e / cv, c4 = (e - p * cv) * 4;
    if (n0 + p < R)
      *reinterpret_cast<float4 *>(out + (n0 + p) * Dout + c4) = *reinterpret_cast<const float4 *>(s_o + (size_t)p * LDO + c4);
  }
}

// run-time K (not the model's path): Dout/4 lanes per point
__global__ __launch_bounds__(256) void conv_pointset_pm_anyk_kernel(
    const float *__restrict__ xyz, const int32_t *__restrict__ nbr, const float *__restrict__ theta,
    const float *__restrict__ bias, long long R, int N, int K, int Dout, EpilogueArgs ep,
    float *__restrict__ out) {
  const int cv = Dout / 4;
  const long long total = R * cv;
  for (long long e = (long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const long long n = e / cv;
    const int o4 = (int)(e - n * cv) * 4;
    const long long cloud0 = (n / N) * N;
    const int32_t *nb = nbr + n * K;
    const long long g0 = cloud0 + nb[0];
    const float x0 = xyz[g0 * 3], y0 = xyz[g0 * 3 + 1], z0 = xyz[g0 * 3 + 2];
    const float4 tx = *reinterpret_cast<const float4 *>(theta + o4);
    const float4 ty = *reinterpret_cast<const float4 *>(theta + Dout + o4);
    const float4 tz = *reinterpret_cast<const float4 *>(theta + 2 * Dout + o4);
    const Ep4 q = ep4_prefetch(ep, o4);  // requested with theta, not one dependent load per channel behind the sum
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < K; ++k) {
      const long long g = cloud0 + nb[k];
      const float dx = xyz[g * 3] - x0, dy = xyz[g * 3 + 1] - y0, dz = xyz[g * 3 + 2] - z0;
      acc.x = fmaf(tz.x, dz, fmaf(ty.x, dy, fmaf(tx.x, dx, acc.x)));
      acc.y = fmaf(tz.y, dz, fmaf(ty.y, dy, fmaf(tx.y, dx, acc.y)));
      acc.z = fmaf(tz.z, dz, fmaf(ty.z, dy, fmaf(tx.z, dx, acc.z)));
      acc.w = fmaf(tz.w, dz, fmaf(ty.w, dy, fmaf(tx.w, dx, acc.w)));
    }
    const float4 bq = *reinterpret_cast<const float4 *>(bias + o4);
    float4 r;
    r.x = dh3d_act(((acc.x + bq.x) + q.pb.x) * q.sc.x + q.sh.x, ep.act);
    r.y = dh3d_act(((acc.y + bq.y) + q.pb.y) * q.sc.y + q.sh.y, ep.act);
    r.z = dh3d_act(((acc.z + bq.z) + q.pb.z) * q.sc.z + q.sh.z, ep.act);
    r.w = dh3d_act(((acc.w + bq.w) + q.pb.w) * q.sc.w + q.sh.w, ep.act);
    *reinterpret_cast<float4 *>(out + n * Dout + o4) = r;
  }
}

// ------------------------------------------------------------------ conv_pointset on coordinates + flex_pool, fused
// The first two operators of the backbone (core/backbones.py:107-110: conv_pointset 3 -> 32, BNReLU, flex_pool) without
// the [B, N, Dout] map in between.  conv_pointset on coordinates is linear in ONE 3-vector per point,
//   S[j] = sum_k (p[nbr[j,k]] - p[nbr[j,0]]),      conv[j, o] = theta[:, o] . S[j] + bias[o]
// (conv_pointset_kernel.cc:46-64 with Din = 3), so the pooled value  max_k act(bn(conv[nbr[n,k], o]))  needs the eight
// neighbours' S vectors (16 bytes each), not their Dout-wide rows:
//   pass 1 (pointset_sum_kernel):  S[j] for every point, one lane per point, [R, 4] floats (1 MB at 8 x 8192);
//   pass 2 (pointset_pool_kernel): Dout/4 lanes per point: the K ids once, K float4 gathers of S, 3 fma + epilogue per
//           (neighbour, channel), running max, one 16-byte store per lane (a row = one 128-byte line at Dout = 32).
// Against the two separate kernels: the 8.4 MB map is neither written nor gathered (8 x 128 bytes per point through
// the L2), and the conv is evaluated with Dout/4 lanes per point (the separate kernel: one lane per point, one wave
// per SIMD on the whole chip).  Values: theta . (sum of differences) instead of the per-neighbour fma chain of
// conv_pointset_pm_kernel -- the same sum associated differently (~1e-7 relative), both within the reference's 1e-4.
__global__ __launch_bounds__(256) void pointset_sum_kernel(const float *__restrict__ xyz,
                                                          const int32_t *__restrict__ nbr, long long R, int N,
                                                          float4 *__restrict__ S) {
  const long long n = (long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (n >= R) return;
  const long long cloud0 = (n / N) * N;
  const int4 a = *reinterpret_cast<const int4 *>(nbr + n * 8), b = *reinterpret_cast<const int4 *>(nbr + n * 8 + 4);
  const int id[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  float px[8], py[8], pz[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float *q = xyz + (cloud0 + id[k]) * 3;
    px[k] = q[0]; py[k] = q[1]; pz[k] = q[2];
  }
  float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) { sx += px[k] - px[0]; sy += py[k] - py[0]; sz += pz[k] - pz[0]; }
  S[n] = make_float4(sx, sy, sz, 0.f);
}

// CV = Dout / 4 lanes per point (compile time: the point / channel split is a shift); 32-bit indexing (R * CV < 2^31)
template <int CV>
__global__ __launch_bounds__(256) void pointset_pool_kernel(const float4 *__restrict__ S, const int32_t *__restrict__ nbr,
                                                           const float *__restrict__ theta, const float *__restrict__ bias,
                                                           int R, int N, EpilogueArgs ep, float *__restrict__ out) {
  constexpr int Dout = CV * 4;
  const unsigned e = (unsigned)dh3d_xcd_remap(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
  const unsigned n = e / CV;
  if (n >= (unsigned)R) return;
  const int c4 = (int)(e % CV) * 4;
  const unsigned cloud0 = (n / (unsigned)N) * (unsigned)N;
  const int4 a = *reinterpret_cast<const int4 *>(nbr + (size_t)n * 8), b = *reinterpret_cast<const int4 *>(nbr + (size_t)n * 8 + 4);
  // (everything below that does not depend on the ids is requested before the ids are waited for)
  const float4 tx = *reinterpret_cast<const float4 *>(theta + c4);
  const float4 ty = *reinterpret_cast<const float4 *>(theta + Dout + c4);
  const float4 tz = *reinterpret_cast<const float4 *>(theta + 2 * Dout + c4);
  const float4 bq = *reinterpret_cast<const float4 *>(bias + c4);
  const Ep4 q = ep4_prefetch(ep, c4);
  const int act = ep.act;
  const int id[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  float4 s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = S[cloud0 + (unsigned)id[k]];
  // max_k act(sc * ((theta . S_k + bias) + pb) + sh): every step after theta . S_k is a monotone function of it (rounded
  // additions and multiplications by a constant are monotone; relu / sigmoid / identity too) -- non-decreasing for
  // sc >= 0, non-increasing for sc < 0 -- so the maximum over the neighbours is that function of the largest (smallest)
  // theta . S_k: three products and one max per (neighbour, channel), the epilogue ONCE per channel.  Bit-identical to
  // evaluating it per neighbour (the same operations meet the same operand).
  const float4 sg = make_float4(q.sc.x < 0.f ? -1.f : 1.f, q.sc.y < 0.f ? -1.f : 1.f, q.sc.z < 0.f ? -1.f : 1.f,
                                q.sc.w < 0.f ? -1.f : 1.f);
  const float4 ux = make_float4(tx.x * sg.x, tx.y * sg.y, tx.z * sg.z, tx.w * sg.w);  // (exact sign flips)
  const float4 uy = make_float4(ty.x * sg.x, ty.y * sg.y, ty.z * sg.z, ty.w * sg.w);
  const float4 uz = make_float4(tz.x * sg.x, tz.y * sg.y, tz.z * sg.z, tz.w * sg.w);
  float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    m.x = fmaxf(m.x, fmaf(uz.x, s[k].z, fmaf(uy.x, s[k].y, ux.x * s[k].x)));
    m.y = fmaxf(m.y, fmaf(uz.y, s[k].z, fmaf(uy.y, s[k].y, ux.y * s[k].x)));
    m.z = fmaxf(m.z, fmaf(uz.z, s[k].z, fmaf(uy.z, s[k].y, ux.z * s[k].x)));
    m.w = fmaxf(m.w, fmaf(uz.w, s[k].z, fmaf(uy.w, s[k].y, ux.w * s[k].x)));
  }
  float4 best;
  best.x = dh3d_act(((m.x * sg.x + bq.x) + q.pb.x) * q.sc.x + q.sh.x, act);
  best.y = dh3d_act(((m.y * sg.y + bq.y) + q.pb.y) * q.sc.y + q.sh.y, act);
  best.z = dh3d_act(((m.z * sg.z + bq.z) + q.pb.z) * q.sc.z + q.sh.z, act);
  best.w = dh3d_act(((m.w * sg.w + bq.w) + q.pb.w) * q.sc.w + q.sh.w, act);
  *reinterpret_cast<float4 *>(out + (size_t)n * Dout + c4) = best;
}

inline int flat_grid(long long total) {
  long long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

DH3D_API int dh3d_flex_conv_pm_gather_fwd(const float *features, const int32_t *remap, int Nsrc, const float *xyz,
                                          const int32_t *nbr, const float *wpacked, int B, int N, int K, int Din,
                                          int Dout, const dh3d_epilogue *ep, float *out, void *stream) {
  DH3D_REQUIRE(features && xyz && nbr && wpacked && out && B > 0 && N > 0 && K > 0 && (!remap || Nsrc > 0));
  const EpilogueArgs e = dh3d_ep(ep);
  hipStream_t s = (hipStream_t)stream;
#define DH3D_FLEX_CASE(DI, DO) \
  if (Din == DI && Dout == DO)  \
  return flex_conv_pm_launch<DI, DO>(features, xyz, nbr, wpacked, B, N, K, e, out, s, remap, Nsrc)
  DH3D_FLEX_CASE(32, 64);
  DH3D_FLEX_CASE(32, 128);
  DH3D_FLEX_CASE(64, 64);
  DH3D_FLEX_CASE(64, 128);
  DH3D_FLEX_CASE(64, 256);
  DH3D_FLEX_CASE(128, 128);
  DH3D_FLEX_CASE(128, 256);
#undef DH3D_FLEX_CASE
  return DH3D_ERR_UNSUPPORTED;
}

DH3D_API int dh3d_flex_conv_pm_post_fwd(const float *features, const float *xyz, const int32_t *nbr, const float *wpacked,
                                        int B, int N, int K, int Din, int Dout, const dh3d_epilogue *ep, float *out,
                                        const float *wpost_packed, int Dpost, float *out2, void *stream) {
  DH3D_REQUIRE(features && xyz && nbr && wpacked && out && wpost_packed && out2 && B > 0 && N > 0);
  DH3D_SUPPORTED(Din == 128 && Dout == 256 && K == 8 && Dpost == 64);
  using C = FlexCfg<128, 256>;
  const long long R = (long long)B * N;
  auto kern = flex_conv_pm_kernel<128, 256, 8, 0, true>;
  DH3D_ALLOW_BIG_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(dh3d_cdiv(R, C::TM)), dim3(256), sizeof(float) * C::TM * C::LD, (hipStream_t)stream, features,
                     xyz, nbr, wpacked, R, N, K, dh3d_ep(ep), out, (const int32_t *)nullptr, 0, wpost_packed, out2);
  return dh3d_launch_status();
}

DH3D_API int dh3d_flex_conv_pm_fwd(const float *features, const float *xyz, const int32_t *nbr,
                                   const float *wpacked, int B, int N, int K, int Din, int Dout,
                                   const dh3d_epilogue *ep, float *out, void *stream) {
  return dh3d_flex_conv_pm_gather_fwd(features, nullptr, 0, xyz, nbr, wpacked, B, N, K, Din, Dout, ep, out, stream);
}

DH3D_API int dh3d_flex_pool_pm_fwd(const float *features, const int32_t *nbr, int B, int N, int K, int C,
                                   float *out, int32_t *argmax, void *stream) {
  DH3D_REQUIRE(features && nbr && out && B > 0 && N > 0 && K > 0 && C > 0);
  DH3D_SUPPORTED(C % 4 == 0);
  const long long R = (long long)B * N;
  hipLaunchKernelGGL(flex_pool_pm_kernel, dim3(flat_grid(R * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     features, nbr, R, N, K, C, out, argmax);
  return dh3d_launch_status();
}

DH3D_API int dh3d_flex_avg_pm_fwd(const float *features, const int32_t *nbr, int B, int N, int K, int C,
                                  float scale, float *out, void *stream) {
  DH3D_REQUIRE(features && nbr && out && B > 0 && N > 0 && K > 0 && C > 0);
  DH3D_SUPPORTED(C % 4 == 0);
  const long long R = (long long)B * N;
  hipLaunchKernelGGL(flex_avg_pm_kernel, dim3(flat_grid(R * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     features, nbr, R, N, K, C, scale, out);
  return dh3d_launch_status();
}

DH3D_API int dh3d_conv_pointset_pm_fwd(const float *xyz, const int32_t *nbr, const float *theta,
                                       const float *bias, int B, int N, int K, int Dout,
                                       const dh3d_epilogue *ep, float *out, void *stream) {
  DH3D_REQUIRE(xyz && nbr && theta && bias && out && B > 0 && N > 0 && K > 0 && Dout > 0);
  DH3D_SUPPORTED(Dout % 4 == 0);
  const long long R = (long long)B * N;
  if (K == 8 && Dout <= 128) {
    const size_t lds = sizeof(float) * 256 * (Dout + 4);
    auto kern = conv_pointset_pm_kernel<8>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, dim3(dh3d_cdiv(R, 256)), dim3(256), lds, (hipStream_t)stream, xyz, nbr, theta, bias, R, N,
                       K, Dout, dh3d_ep(ep), out);
  } else {
    hipLaunchKernelGGL(conv_pointset_pm_anyk_kernel, dim3(flat_grid(R * (Dout / 4))), dim3(256), 0,
                       (hipStream_t)stream, xyz, nbr, theta, bias, R, N, K, Dout, dh3d_ep(ep), out);
  }
  return dh3d_launch_status();
}

DH3D_API int dh3d_pointset_sum_pm(const float *xyz, const int32_t *nbr, int B, int N, int K, float *S, void *stream) {
  DH3D_REQUIRE(xyz && nbr && S && B > 0 && N > 0);
  DH3D_SUPPORTED(K == 8);
  const long long R = (long long)B * N;
  hipLaunchKernelGGL(pointset_sum_kernel, dim3(dh3d_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, xyz, nbr, R, N,
                     reinterpret_cast<float4 *>(S));
  return dh3d_launch_status();
}

DH3D_API int dh3d_conv_pointset_pool_pm_fwd(const float *xyz, const int32_t *nbr, const float *theta, const float *bias,
                                            int B, int N, int K, int Dout, const dh3d_epilogue *ep, float *scratch,
                                            float *out, void *stream) {
  DH3D_REQUIRE(xyz && nbr && theta && bias && scratch && out && B > 0 && N > 0 && Dout > 0);
  DH3D_SUPPORTED(K == 8 && (Dout == 32 || Dout == 64 || Dout == 128) && (long long)B * N * (Dout / 4) < (1ll << 31));
  const long long R = (long long)B * N;
  hipStream_t s = (hipStream_t)stream;
  float4 *S = reinterpret_cast<float4 *>(scratch);
  hipLaunchKernelGGL(pointset_sum_kernel, dim3(dh3d_cdiv(R, 256)), dim3(256), 0, s, xyz, nbr, R, N, S);
  const dim3 grid(dh3d_cdiv(R * (Dout / 4), 256)), block(256);
  const EpilogueArgs e = dh3d_ep(ep);
  if (Dout == 32)
    hipLaunchKernelGGL(pointset_pool_kernel<8>, grid, block, 0, s, S, nbr, theta, bias, (int)R, N, e, out);
  else if (Dout == 64)
    hipLaunchKernelGGL(pointset_pool_kernel<16>, grid, block, 0, s, S, nbr, theta, bias, (int)R, N, e, out);
  else
    hipLaunchKernelGGL(pointset_pool_kernel<32>, grid, block, 0, s, S, nbr, theta, bias, (int)R, N, e, out);
  return dh3d_launch_status();
}

#ifdef DH3D_FLEX_PROBE
DH3D_API int dh3d_flex_probe_read(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_fprobe), sizeof(long long) * n) == hipSuccess ? 0 : 3;
}
#endif
