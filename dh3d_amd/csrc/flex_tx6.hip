// flex_conv on 32-point tiles with the tile GEMM on the bf16 matrix pipe at f32 accuracy (bf16x6, bf16x3.h) -- the
// sampled levels' shapes (Din 64 / 128 -> Dout 128 / 256, K = 8) and BASELINE config 5's 128 -> 128, K = 12.
//
// flex_conv_pm_kernel (flex_pm.hip) spends its time like this on 8 x 1024 points, 128 -> 128 (clock stamps of 64
// workgroups, tools/flex_probe_n8.py): gather 9.3 k cycles, GEMM 20.4 k, store 2.9 k -- and the GEMM phase IS the f32 matrix
// pipe's floor for one 32-point tile per CU (256 v_mfma_f32_32x32x2_f32 of 64 cycles per SIMD = 16.4 k).  The same
// product as six v_mfma_f32_32x32x16_bf16 per 16 k (32 cycles each) is 2.7x less pipe time.  Same factorisation, same
// gather (two dependent round trips per tile, no load under a branch); the S tile goes to LDS as three bf16 planes
// (split in registers by the thread that reduced it), the weight comes pre-split in fragment order
// (dh3d_pack_flex_weight_x3) from L2, four K-steps ahead.  Reference: FlexConvCuda::ForwardKernel
// (user_ops/flex_conv_kernel_gpu.cu.cc:46-158); f32-accurate (max deviation from the fp64 result ~2e-6 of the largest
// output, the same as flex_conv_x6_kernel).
#include "bf16x3.h"
#include "mfma_gemm.h"

namespace {

#ifdef DH3D_FLEX_PROBE  // dev instrumentation (tools/flex_probe_n8.py): cycle stamps of a few workgroups
__device__ long long g_tprobe[64 * 8];
#define TPROBE(i) do { if (threadIdx.x == 0 && blockIdx.x < 64) g_tprobe[blockIdx.x * 8 + (i)] = clock64(); } while (0)
#else
#define TPROBE(i) do { } while (0)
#endif

template <int DIN, int DOUT>
struct TileCfg {
  static constexpr int TM = 32;
  static constexpr int KD = 4 * DIN;    // GEMM depth
  static constexpr int LDB = KD + 8;    // bf16 per row of a plane: 16-byte aligned rows, an odd number of 16-byte units
  static constexpr int KS = KD / 16;    // K-steps of one bf16 MFMA
  static constexpr int NT = DOUT / 128; // 32-column blocks per wave (four waves)
  static constexpr int LPR = DIN / 4;   // lanes per gathered row (float4 each)
  static constexpr int PPR = 256 / LPR; // points per gather round
  static constexpr int ROUNDS = TM / PPR;
  static_assert(DOUT % 128 == 0 && (LDB * 2 / 16) % 2 == 1, "tile layout");
};

// POST: the finished tile goes through one more linear layer [DOUT -> 64] on the f32 pipe (wpost = dh3d_pack_weight of
// [DOUT, 64]) before it leaves the chip, both results stored (see flex_conv_pm_kernel).
// HALF: the planes hold one half of K at a time -- [S0|Sx], then [Sy|Sz], which wait in the registers of the threads that
// reduced them: 51 KB of LDS instead of 100 at Din = 128, so that TWO workgroups share a CU and one's gather runs under the
// other's products.  For launches with more tiles than CUs (cfg 5's 512 tiles, the global step's sampled level).
template <int DIN, int DOUT, int KT, bool POST, bool HALF>
__global__ __launch_bounds__(256) void flex_conv_tx6_kernel(const float *__restrict__ feat, const float *__restrict__ xyz,
                                                           const int32_t *__restrict__ nbr,
                                                           const uint4 *__restrict__ wp3, long long R, int N,
                                                           EpilogueArgs ep, float *__restrict__ out,
                                                           const float *__restrict__ wpost, float *__restrict__ out2) {
  using C = TileCfg<DIN, DOUT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  unsigned short *s_P = reinterpret_cast<unsigned short *>(s_raw);  // [3][TM][LDP] bf16 planes of S = [S0|Sx|Sy|Sz] (or a half)
  float *s_out = reinterpret_cast<float *>(s_raw);                  // [TM][LDO] after the GEMM
  constexpr int LDO = DOUT + (POST ? 64 : 0) + 4;
  constexpr int LDP = HALF ? C::KD / 2 + 8 : C::LDB;  // bf16 per plane row
  constexpr int PLANE = C::TM * LDP;
  static_assert((LDP * 2 / 16) % 2 == 1, "plane rows: an odd number of 16-byte units");
  static_assert((size_t)C::TM * LDO * 4 <= (size_t)3 * PLANE * 2, "the output tile reuses the planes");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long grow0 = (long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * C::TM;

  // ---- phase A: gather-reduce S for TM points (two dependent round trips for the whole tile), split, planes
  TPROBE(0);
  const int r4 = (tid % C::LPR) * 4;
  float4 keep[HALF ? C::ROUNDS : 1][2];  // HALF: [Sy|Sz] of this thread's rows until the planes are free again
  auto to_planes = [&](const float4 v, unsigned short *dst) __attribute__((always_inline)) {
    uint2 c1, c2, c3;
    split3x4(v, c1, c2, c3);
    *reinterpret_cast<uint2 *>(dst) = c1;
    *reinterpret_cast<uint2 *>(dst + PLANE) = c2;
    *reinterpret_cast<uint2 *>(dst + 2 * PLANE) = c3;
  };
  {
    constexpr int HF = KT <= 8 ? 2 : 1;  // rounds whose neighbour rows are in flight together
    int nid[C::ROUNDS][KT];
    float pxyz[C::ROUNDS][3];
    long long cloud0[C::ROUNDS];
    bool ok[C::ROUNDS];
    const unsigned bq0 = (unsigned)(grow0 / N);  // (uniform: one division per workgroup, then at most a step per point)
#pragma unroll
    for (int rd = 0; rd < C::ROUNDS; ++rd) {
      const long long n = grow0 + rd * C::PPR + tid / C::LPR;
      ok[rd] = n < R;
      const long long nn = ok[rd] ? n : grow0;
      unsigned b = bq0;
      long long off = nn - (long long)bq0 * N;
      while (off >= N) { off -= N; ++b; }
      cloud0[rd] = (long long)b * N;
      const int4 *ip = reinterpret_cast<const int4 *>(nbr + nn * KT);
#pragma unroll
      for (int q = 0; q < KT / 4; ++q) {
        const int4 v = ip[q];
        nid[rd][4 * q] = v.x; nid[rd][4 * q + 1] = v.y; nid[rd][4 * q + 2] = v.z; nid[rd][4 * q + 3] = v.w;
      }
      pxyz[rd][0] = xyz[nn * 3]; pxyz[rd][1] = xyz[nn * 3 + 1]; pxyz[rd][2] = xyz[nn * 3 + 2];
    }
#pragma unroll
    for (int rp = 0; rp < C::ROUNDS; rp += HF) {
      float4 fv[HF][KT];
      float qv[HF][KT][3];
#pragma unroll
      for (int h = 0; h < HF; ++h)
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const long long g = cloud0[rp + h] + nid[rp + h][k];
          fv[h][k] = *reinterpret_cast<const float4 *>(feat + g * DIN + r4);
          qv[h][k][0] = xyz[g * 3]; qv[h][k][1] = xyz[g * 3 + 1]; qv[h][k][2] = xyz[g * 3 + 2];
        }
#pragma unroll
      for (int h = 0; h < HF; ++h) {
        const int rd = rp + h;
        float4 s[4];
        s[0] = make_float4(0.f, 0.f, 0.f, 0.f); s[1] = s[0]; s[2] = s[0]; s[3] = s[0];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          const float4 f = fv[h][k];
          const float dx = qv[h][k][0] - pxyz[rd][0], dy = qv[h][k][1] - pxyz[rd][1], dz = qv[h][k][2] - pxyz[rd][2];
          s[0].x += f.x; s[0].y += f.y; s[0].z += f.z; s[0].w += f.w;
          s[1].x = fmaf(dx, f.x, s[1].x); s[1].y = fmaf(dx, f.y, s[1].y); s[1].z = fmaf(dx, f.z, s[1].z); s[1].w = fmaf(dx, f.w, s[1].w);
          s[2].x = fmaf(dy, f.x, s[2].x); s[2].y = fmaf(dy, f.y, s[2].y); s[2].z = fmaf(dy, f.z, s[2].z); s[2].w = fmaf(dy, f.w, s[2].w);
          s[3].x = fmaf(dz, f.x, s[3].x); s[3].y = fmaf(dz, f.y, s[3].y); s[3].z = fmaf(dz, f.z, s[3].z); s[3].w = fmaf(dz, f.w, s[3].w);
        }
        unsigned short *row = s_P + (size_t)(rd * C::PPR + tid / C::LPR) * LDP + r4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (!ok[rd]) s[c] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (HALF && c >= 2) keep[HALF ? rd : 0][c - 2] = s[c];
          else to_planes(s[c], row + c * DIN);
        }
      }
    }
  }
  TPROBE(1);
  __syncthreads();
  TPROBE(2);

  // ---- phase B: S @ Wcat, six bf16 products per f32 product, weight fragments four K-steps ahead
  f32x16 acc[C::NT];
  zero_acc<C::NT>(acc);
  EpilogueRegs er[C::NT];
#pragma unroll
  for (int j = 0; j < C::NT; ++j) er[j] = epilogue_prefetch(ep, (wave + j * 4) * 32 + (lane & 31));
  {
    const unsigned short *abase = s_P + (size_t)(lane & 31) * LDP + 8 * (lane >> 5);
    const uint4 *wl = wp3 + lane;
    constexpr int G = C::NT == 1 ? 4 : 2;  // K-steps per group: 24 products = 768 cycles of matrix pipe either way
    static_assert(C::KS % G == 0, "whole groups");
    static_assert((C::KS / G) % 2 == 0, "an even number of groups: two buffers swap roles inside one loop body");
    auto request = [&](uint4 (&b)[G][C::NT][3], int ks) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < G; ++u)
#pragma unroll
        for (int j = 0; j < C::NT; ++j)
#pragma unroll
          for (int p = 0; p < 3; ++p) b[u][j][p] = wl[((size_t)((wave + j * 4) * C::KS + ks + u) * 3 + p) * 64];
    };
    // the group's A fragments are all requested before its first product (one LDS latency per group, not three reads
    // waited for in front of every K-step: with ONE wave per SIMD nothing else hides them)
    auto multiply = [&](const uint4 (&b)[G][C::NT][3], int ks) __attribute__((always_inline)) {  // ks: within the planes
      bf16x8 a[G][3];
#pragma unroll
      for (int u = 0; u < G; ++u)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[u][p] = *reinterpret_cast<const bf16x8 *>(abase + (size_t)p * PLANE + (ks + u) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < G; ++u) {
#define DH3D_TX6_PRODUCT(PA, PB)                                                                    \
  _Pragma("unroll") for (int j = 0; j < C::NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16( \
      a[u][PA], __builtin_bit_cast(bf16x8, b[u][j][PB]), acc[j], 0, 0, 0);
        DH3D_TX6_PRODUCT(2, 0) DH3D_TX6_PRODUCT(0, 2) DH3D_TX6_PRODUCT(1, 1)
        DH3D_TX6_PRODUCT(1, 0) DH3D_TX6_PRODUCT(0, 1) DH3D_TX6_PRODUCT(0, 0)
#undef DH3D_TX6_PRODUCT
      }
    };
    // two buffers that swap roles inside one loop body (no copies), requests unconditional (the last one a harmless
    // repeat) and fenced off from the products: see wave_gemm_f32 (mfma_gemm.h) for what the compiler does otherwise
    uint4 b0[G][C::NT][3], b1[G][C::NT][3];
    constexpr int KSP = HALF ? C::KS / 2 : C::KS;  // K-steps per fill of the planes
    static_assert((KSP / G) % 2 == 0, "an even number of groups per fill");
    request(b0, 0);
#pragma unroll 1
    for (int ks = 0; ks < KSP; ks += 2 * G) {
      request(b1, ks + G);
      __builtin_amdgcn_sched_barrier(0);
      multiply(b0, ks);
      request(b0, ks + 2 * G < C::KS ? ks + 2 * G : ks + G);  // (HALF: the last one is the second fill's first group)
      __builtin_amdgcn_sched_barrier(0);
      multiply(b1, ks + G);
    }
    if (HALF) {
      __syncthreads();  // every wave is done with [S0|Sx]
#pragma unroll
      for (int rd = 0; rd < C::ROUNDS; ++rd) {
        unsigned short *row = s_P + (size_t)(rd * C::PPR + tid / C::LPR) * LDP + r4;
        to_planes(keep[HALF ? rd : 0][0], row);
        to_planes(keep[HALF ? rd : 0][1], row + DIN);
      }
      __syncthreads();
#pragma unroll 1
      for (int ks = KSP; ks < C::KS; ks += 2 * G) {
        request(b1, ks + G);
        __builtin_amdgcn_sched_barrier(0);
        multiply(b0, ks - KSP);
        request(b0, ks + 2 * G < C::KS ? ks + 2 * G : ks + G);
        __builtin_amdgcn_sched_barrier(0);
        multiply(b1, ks + G - KSP);
      }
    }
  }
#ifdef DH3D_FLEX_PROBE
  asm volatile("" :: "v"(acc[0][0]));
#endif
  TPROBE(3);
  // ---- epilogue through the (now dead) planes
  __syncthreads();
  wave_tiles_to_lds<C::NT>(acc, er, ep.act, s_out, LDO, 0, wave, 4);
  __syncthreads();
  block_store_rows(s_out, LDO, C::TM, grow0, R, DOUT, nullptr, out);
  if (POST) {
    if (wave < 2) {  // [32, DOUT] x [DOUT, 64] on the f32 pipe: one 32-column block per wave, the tile in LDS is the A operand
      f32x16 pacc[1];
      zero_acc<1>(pacc);
      wave_gemm_f32<1>(s_out, LDO, 0, wpost, DOUT / 8, wave, 1, pacc);
      const EpilogueRegs none[1] = {EpilogueRegs{0.f, 1.f, 0.f}};
      wave_tiles_to_lds<1>(pacc, none, DH3D_ACT_NONE, s_out + DOUT, LDO, 0, wave, 1);
    }
    __syncthreads();
    block_store_rows(s_out + DOUT, LDO, C::TM, grow0, R, 64, nullptr, out2);
  }
  TPROBE(4);
}

template <int DIN, int DOUT, int KT, bool POST, bool HALF>
int tx6_launch_h(const float *feat, const float *xyz, const int32_t *nbr, const void *wp3, int B, int N,
               const EpilogueArgs &ep, float *out, const float *wpost, float *out2, hipStream_t s) {
  using C = TileCfg<DIN, DOUT>;
  const long long R = (long long)B * N;
  auto kern = flex_conv_tx6_kernel<DIN, DOUT, KT, POST, HALF>;
  DH3D_ALLOW_BIG_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(dh3d_cdiv(R, C::TM)), dim3(256), (size_t)3 * C::TM * (HALF ? C::KD / 2 + 8 : C::LDB) * 2,
                     s, feat, xyz, nbr, static_cast<const uint4 *>(wp3), R, N, ep, out, wpost, out2);
  return dh3d_launch_status();
}

template <int DIN, int DOUT, int KT, bool POST>
int tx6_launch(const float *feat, const float *xyz, const int32_t *nbr, const void *wp3, int B, int N,
               const EpilogueArgs &ep, float *out, const float *wpost, float *out2, hipStream_t s) {
  // more tiles than CUs: half-K planes, two workgroups per CU (Din = 128: 51 KB each instead of 100)
#ifndef DH3D_TX6_HALF_TILES
#define DH3D_TX6_HALF_TILES 256
#endif
  if (DIN == 128 && (long long)B * N > 32ll * DH3D_TX6_HALF_TILES)
    return tx6_launch_h<DIN, DOUT, KT, POST, true>(feat, xyz, nbr, wp3, B, N, ep, out, wpost, out2, s);
  return tx6_launch_h<DIN, DOUT, KT, POST, false>(feat, xyz, nbr, wp3, B, N, ep, out, wpost, out2, s);
}

}  // namespace

DH3D_API int dh3d_flex_conv_pm_tile_x6_fwd(const float *features, const float *xyz, const int32_t *nbr,
                                           const void *wpacked_x3, int B, int N, int K, int Din, int Dout,
                                           const dh3d_epilogue *ep, float *out, const float *wpost_packed, int Dpost,
                                           float *out2, void *stream) {
  DH3D_REQUIRE(features && xyz && nbr && wpacked_x3 && out && B > 0 && N > 0 && K > 0);
  DH3D_REQUIRE(!wpost_packed || out2);
  const EpilogueArgs e = dh3d_ep(ep);
  hipStream_t s = (hipStream_t)stream;
  if (wpost_packed) {
    DH3D_SUPPORTED(K == 8 && Din == 128 && Dout == 256 && Dpost == 64);
    return tx6_launch<128, 256, 8, true>(features, xyz, nbr, wpacked_x3, B, N, e, out, wpost_packed, out2, s);
  }
#define DH3D_TX6_CASE(DI, DO, KK) \
  if (Din == DI && Dout == DO && K == KK) return tx6_launch<DI, DO, KK, false>(features, xyz, nbr, wpacked_x3, B, N, e, out, nullptr, nullptr, s);
  DH3D_TX6_CASE(64, 128, 8) DH3D_TX6_CASE(128, 128, 8) DH3D_TX6_CASE(128, 256, 8) DH3D_TX6_CASE(128, 128, 12)
#undef DH3D_TX6_CASE
  return DH3D_ERR_UNSUPPORTED;
}

#ifdef DH3D_FLEX_PROBE
DH3D_API int dh3d_flex_tprobe_read(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tprobe), sizeof(long long) * n) == hipSuccess ? 0 : 3;
}
#endif
