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
#include <type_traits>

#include "bf16x3.h"
#include "wave_ops.h"
#include "internal.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {


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
//
// A tiled GEMM: one workgroup owns 128 rows and walks the 1024 columns in tiles of 256; the K loop moves in
// chunks of 32 through a double-buffered LDS stage holding the A chunk (split into its three bf16 planes as
// it is staged) and the matching slice of the pre-split weight.  Each of the eight waves (2 x 4, two per
// SIMD so one covers the other's waits) keeps a 64 x 64 block of f32 accumulators (four independent MFMA
// chains); a weight fragment read from LDS feeds two row blocks and an A fragment two column blocks, and the
// packed weight crosses L2 -> CU once per 128 rows (1.5 GB per launch at R = 131072 instead of
// the 6.4 GB of a 64-row, weight-streaming layout -- which is what made the first version no faster than f32).
constexpr int HTM = 128, HTN = 256, HKC = 32;
constexpr int LDA = HKC + 8;                       // bf16 per row of a staged A plane (80 B: conflict-free b128)
constexpr int A_STAGE = 3 * HTM * LDA;             // bf16 elements per A buffer
constexpr int B_STAGE = (HTN / 32) * 2 * 3 * 64;   // uint4 per B buffer  [cb 8][ks 2][plane 3][lane 64]

// WC = waves along the columns (2 rows x WC): WC = 4 -> eight waves of 64 x 64, two per SIMD; WC = 2 -> four waves
// of 64 x 128, one per SIMD (more register reuse per LDS byte, no partner wave to cover its waits).
template <int WC>
__global__ __launch_bounds__(128 * WC) void mlp_head_x6_kernel(const float *__restrict__ h, int C,
                                                              const uint4 *__restrict__ wp, int H, EpilogueArgs ep,
                                                              const float *__restrict__ w_fc, float b_fc,
                                                              long long R, float *__restrict__ att) {
  constexpr int T = 128 * WC;         // threads
  constexpr int NCW = 8 / WC;         // 32-column blocks per wave
  constexpr int AQ = T / HTM;         // threads per staged A row
  constexpr int NF4 = 8 / AQ;         // float4 per thread per chunk
  constexpr int DMA = 48 / (2 * WC);  // LDS-DMA instructions per wave per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  unsigned short *s_A = reinterpret_cast<unsigned short *>(s_raw);                      // [2][3][HTM][LDA]
  uint4 *s_B = reinterpret_cast<uint4 *>(s_raw + (size_t)2 * A_STAGE * 2);             // [2][B_STAGE]
  float *s_part = reinterpret_cast<float *>(s_raw + (size_t)2 * A_STAGE * 2 + (size_t)2 * B_STAGE * 16);  // [WC][HTM]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const long long grow0 = (long long)blockIdx.x * HTM;
  const int KB = C / 16, NCH = C / HKC, NT = H / HTN, IT = NT * NCH;
  // staging roles: A -- AQ threads per row, 4*NF4 consecutive k each; B -- LDS-DMA
  const int ar = tid / AQ, ah = tid % AQ;
  long long arow = grow0 + ar;
  if (arow >= R) arow = R - 1;  // rows past R repeat the last row; their results are not stored
  const float *asrc = h + arow * C + ah * (4 * NF4);
  float4 pa[NF4];
  // A chunk of iteration `it` -> registers (split and written to LDS by stage_a)
  auto prefetch_a = [&](int it) __attribute__((always_inline)) {
    const int nt = it / NCH, ch = it - nt * NCH;
    const float4 *ap = reinterpret_cast<const float4 *>(asrc + ch * HKC);
#pragma unroll
    for (int j = 0; j < NF4; ++j) pa[j] = ap[j];
  };
  // weight slice of iteration `it` -> LDS buffer `buf` by LDS-DMA (global_load_lds_dwordx4: no registers; each
  // wave-instruction lands 64 x 16 B contiguously).  48 KB per chunk.
  auto dma_b = [&](int it, int buf) __attribute__((always_inline)) {
    const int nt = it / NCH, ch = it - nt * NCH;
#pragma unroll
    for (int j = 0; j < DMA; ++j) {
      const int e0 = (wave * DMA + j) * 64, cb = e0 / 384, rem = e0 - cb * 384 + lane;
      const uint4 *src = wp + ((size_t)(nt * (HTN / 32) + cb) * KB + ch * 2) * 192 + rem;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(s_B + (size_t)buf * B_STAGE + e0),
                                       16, 0, 0);
    }
  };
  auto stage_a = [&](int buf) __attribute__((always_inline)) {
    unsigned short *dst = s_A + (size_t)buf * A_STAGE + (size_t)ar * LDA + ah * (4 * NF4);
#pragma unroll
    for (int j = 0; j < NF4; j += 2) {
      uint2 c1[2], c2[2], c3[2];
      split3x4(pa[j], c1[0], c2[0], c3[0]);
      split3x4(pa[j + 1], c1[1], c2[1], c3[1]);
      *reinterpret_cast<uint4 *>(dst + 4 * j) = make_uint4(c1[0].x, c1[0].y, c1[1].x, c1[1].y);
      *reinterpret_cast<uint4 *>(dst + 4 * j + HTM * LDA) = make_uint4(c2[0].x, c2[0].y, c2[1].x, c2[1].y);
      *reinterpret_cast<uint4 *>(dst + 4 * j + 2 * HTM * LDA) = make_uint4(c3[0].x, c3[0].y, c3[1].x, c3[1].y);
    }
  };
  auto stage_sync = [&]() __attribute__((always_inline)) {  // LDS-DMA counts on vmcnt
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  float part[2][16];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[rb][r] = 0.f;

  prefetch_a(0);
  dma_b(0, 0);
  stage_a(0);
  stage_sync();
  int it = 0;
  for (int nt = 0; nt < NT; ++nt) {
    f32x16 acc[2][NCW];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < NCW; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
    for (int ch = 0; ch < NCH; ++ch, ++it) {
      const int nxt = it + 1 < IT ? it + 1 : it;  // unconditional prefetch: the last one is a harmless repeat
      prefetch_a(nxt);
      const int buf = it & 1;
      dma_b(nxt, buf ^ 1);                   // lands under this chunk's MFMAs (waited for in stage_sync)
      __builtin_amdgcn_sched_barrier(0);     // keep both prefetches up here: the scheduler sinks loads to their use
      const unsigned short *abase = s_A + (size_t)buf * A_STAGE + (size_t)(wr * 64 + (lane & 31)) * LDA + 8 * (lane >> 5);
      const uint4 *bbase = s_B + (size_t)buf * B_STAGE + (size_t)(wc * NCW) * (2 * 3 * 64) + lane;
      // the fragments of BOTH k-steps are requested before the first MFMA: one exposed LDS latency per chunk
      bf16x8 a[2][2][3], b[2][NCW][3];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            a[ks][rb][p] = *reinterpret_cast<const bf16x8 *>(abase + (size_t)p * HTM * LDA + (size_t)rb * 32 * LDA + ks * 16);
#pragma unroll
        for (int cb = 0; cb < NCW; ++cb)
#pragma unroll
          for (int p = 0; p < 3; ++p) b[ks][cb][p] = __builtin_bit_cast(bf16x8, bbase[((cb * 2 + ks) * 3 + p) * 64]);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // six products, smallest first; 2*NCW independent accumulators between two uses of the same one
#define DH3D_X6_PRODUCT(PA, PB)                                                                           \
  _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) _Pragma("unroll") for (int cb = 0; cb < NCW; ++cb)     \
      acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][rb][PA], b[ks][cb][PB], acc[rb][cb], 0, 0, 0);
        DH3D_X6_PRODUCT(2, 0) DH3D_X6_PRODUCT(0, 2) DH3D_X6_PRODUCT(1, 1)
        DH3D_X6_PRODUCT(1, 0) DH3D_X6_PRODUCT(0, 1) DH3D_X6_PRODUCT(0, 0)
#undef DH3D_X6_PRODUCT
      }
      stage_a(buf ^ 1);
      stage_sync();
    }
    // epilogue of this column tile: BN + activation, dot with w_fc, kept per lane (one column per lane)
#pragma unroll
    for (int cb = 0; cb < NCW; ++cb) {
      const int col = nt * HTN + (wc * NCW + cb) * 32 + (lane & 31);
      float pb = 0.f, sc = 1.f, sh = 0.f;
      if (ep.pre_bias) pb = ep.pre_bias[col];
      if (ep.scale) sc = ep.scale[col];
      if (ep.shift) sh = ep.shift[col];
      const float wf = w_fc[col];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          part[rb][r] = fmaf(dh3d_act((acc[rb][cb][r] + pb) * sc + sh, ep.act), wf, part[rb][r]);
    }
  }
  // row sums: across the 32 column lanes, then across the column waves
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) part[rb][r] += __shfl_xor(part[rb][r], off, 64);
    }
  if ((lane & 31) == 0) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        s_part[wc * HTM + wr * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = part[rb][r];
  }
  __syncthreads();
  if (tid < HTM) {
    const long long g = grow0 + tid;
    float z = 0.f;
#pragma unroll
    for (int w = 0; w < WC; ++w) z += s_part[w * HTM + tid];
    if (g < R) att[g] = 1.f / (1.f + expf(-(z + b_fc)));
  }
}

// ------------------------------------------------------------------------------------------------
// out[R, Dout] = act(bn([x1 | x2] @ W + b)) (+ residual) for the wide 1x1 convs (Dout = 128 or 256), same tiling
// and staging as the head above; NC = 32-column blocks per wave (1: Dout = 128, 2: Dout = 256).  The finished
// 128 x Dout tile goes through LDS (the stage buffers are dead by then) so that every lane stores 16 bytes of
// a full output row, residual added on the way.
// The interpolation arithmetic must round exactly like three_interp_fwd_kernel<IDW> (pointnet2.hip is compiled
// without contraction): no fused multiply-adds in these two helpers.
#pragma clang fp contract(off)
__device__ __forceinline__ void idw_weights(float d1, float d2, float d3, float &w1, float &w2, float &w3) {
  const float r1 = 1.0f / fmaxf(d1, 1e-10f), r2 = 1.0f / fmaxf(d2, 1e-10f), r3 = 1.0f / fmaxf(d3, 1e-10f);
  const float norm = (r1 + r2) + r3;
  w1 = r1 / norm; w2 = r2 / norm; w3 = r3 / norm;
}
__device__ __forceinline__ float4 idw_mix(const float4 a, const float4 b, const float4 c, float w1, float w2, float w3) {
  float4 r;
  r.x = (a.x * w1 + b.x * w2) + c.x * w3;
  r.y = (a.y * w1 + b.y * w2) + c.y * w3;
  r.z = (a.z * w1 + b.z * w2) + c.z * w3;
  r.w = (a.w * w1 + b.w * w2) + c.w * w3;
  return r;
}
#pragma clang fp contract(fast)
// the same mix with fused multiply-adds (3 instructions per channel instead of 5): for the COMMUTED walks, whose
// arithmetic is not the reference's association anyway (the interpolation there runs on rows a linear layer has
// already been applied to)
__device__ __forceinline__ float4 idw_mix_fma(const float4 a, const float4 b, const float4 c, float w1, float w2, float w3) {
  float4 r;
  r.x = fmaf(c.x, w3, fmaf(b.x, w2, a.x * w1));
  r.y = fmaf(c.y, w3, fmaf(b.y, w2, a.y * w1));
  r.z = fmaf(c.z, w3, fmaf(b.z, w2, a.z * w1));
  r.w = fmaf(c.w, w3, fmaf(b.w, w2, a.w * w1));
  return r;
}

// Up-sampling source for the x1 half (three_interpolate with inverse-distance weights, core/backbones.py:91-95,
// fused into the A staging): x1[r, :] = sum_t w[r,t] * points[cloud(r), idx[r,t], :], exactly the arithmetic of
// three_interp_fwd_kernel<IDW> (unfused, same association), so the fused and the two-kernel paths agree bit for bit.
struct UpsampleSrc {
  const float *points;   // [B, m, C1]   (null: x1 is read directly)
  const int32_t *idx;    // [R, 3]
  const float *dist;     // [R, 3] squared distances
  int n, m;              // rows per cloud of the fine / coarse level
};

// Optional last step of the local path (core/model.py:177-181): out_cat[r] = [prefix[r, 0:3] | l2_normalize(y[r])]
// written straight from the output tile (Dout == 128), instead of y: saves a pass over the [R,128] tensor.
struct L2CatOut {
  float *out;           // [R, 3 + 128]   (null: plain output)
  const float *prefix;  // [R, 3]
  float eps;            // tf.nn.l2_normalize: y * rsqrt(max(sum(y^2), eps))
};

// Optional second 1x1 conv summed into the output: out = act(BN([x1|x2] W)) + act_sc(BN_sc(x3 W_sc)) -- the local
// backbone's shortcut branch (core/backbones.py:123), whose [R,128] result would otherwise be written by one kernel and
// read back by this one (134 MB at 131 k rows).  x3's K-chunks follow the main ones in the same pipeline (the packed
// weight is the two matrices stacked) and feed a second accumulator set.
struct ShortcutSrc {
  const float *x3;  // [R, C3]   (null: none)
  int C3;
  EpilogueArgs ep;
};

template <int NC, bool SC>
__global__ __launch_bounds__(512) void linear_x6_kernel(const float *__restrict__ x1, int C1,
                                                       const float *__restrict__ x2, int C2,
                                                       const uint4 *__restrict__ wp, EpilogueArgs ep,
                                                       const float *__restrict__ residual, long long R,
                                                       float *__restrict__ out, UpsampleSrc up, L2CatOut l2,
                                                       ShortcutSrc sc, long long wslice, long long oslice) {
  // blockIdx.y = column slice of a wider layer: its own packed weight and output plane (dh3d_linear_slices_pm_x6_fwd)
  wp += (size_t)blockIdx.y * wslice;
  out += (size_t)blockIdx.y * oslice;
  constexpr int TN = NC * 128;                      // columns of the tile = Dout
  constexpr int BST = (TN / 32) * 2 * 3 * 64;       // uint4 per B buffer
  constexpr int DMA = BST / 64 / 8;                 // LDS-DMA instructions per wave per chunk
  constexpr int LDO = TN + 4;                       // leading dimension of the output tile in LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  unsigned short *s_A = reinterpret_cast<unsigned short *>(s_raw);          // [2][3][HTM][LDA]
  uint4 *s_B = reinterpret_cast<uint4 *>(s_raw + (size_t)2 * A_STAGE * 2);  // [2][BST]
  float *s_out = reinterpret_cast<float *>(s_raw);                          // [HTM][LDO] after the K loop
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const long long grow0 = (long long)blockIdx.x * HTM;
  const int C3 = SC ? sc.C3 : 0;
  const int C = C1 + C2 + C3, KB = C / 16, NCH = C / HKC, NCH_MAIN = (C1 + C2) / HKC;
  const int ar = tid >> 2, ah = tid & 3;
  long long arow = grow0 + ar;
  if (arow >= R) arow = R - 1;  // rows past R repeat the last row; they are not stored
  float4 pa[2];
  // up-sampling: this thread's row gathers from three coarse rows with fixed weights
  const float *ip1 = nullptr, *ip2 = nullptr, *ip3 = nullptr;
  float w1 = 0.f, w2 = 0.f, w3 = 0.f;
  if (up.points) {
    const long long bi = arow / up.n;
    const int i1 = up.idx[arow * 3], i2 = up.idx[arow * 3 + 1], i3 = up.idx[arow * 3 + 2];
    idw_weights(up.dist[arow * 3], up.dist[arow * 3 + 1], up.dist[arow * 3 + 2], w1, w2, w3);
    ip1 = up.points + (bi * up.m + i1) * C1;
    ip2 = up.points + (bi * up.m + i2) * C1;
    ip3 = up.points + (bi * up.m + i3) * C1;
  }
  auto prefetch_a = [&](int ch) __attribute__((always_inline)) {
    const int k0 = ch * HKC + ah * 8;
    if (up.points && k0 < C1) {  // uniform per chunk
      const float4 *a = reinterpret_cast<const float4 *>(ip1 + k0), *b = reinterpret_cast<const float4 *>(ip2 + k0),
                   *c = reinterpret_cast<const float4 *>(ip3 + k0);
      pa[0] = idw_mix(a[0], b[0], c[0], w1, w2, w3);
      pa[1] = idw_mix(a[1], b[1], c[1], w1, w2, w3);
    } else {
      const float *src = k0 < C1 ? x1 + arow * C1 + k0
                         : (!SC || k0 < C1 + C2) ? x2 + arow * C2 + (k0 - C1)
                                                 : sc.x3 + arow * C3 + (k0 - C1 - C2);
      const float4 *ap = reinterpret_cast<const float4 *>(src);
      pa[0] = ap[0]; pa[1] = ap[1];
    }
  };
  auto dma_b = [&](int ch, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < DMA; ++j) {
      const int e0 = (wave * DMA + j) * 64, cb = e0 / 384, rem = e0 - cb * 384 + lane;
      const uint4 *src = wp + ((size_t)cb * KB + ch * 2) * 192 + rem;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(s_B + (size_t)buf * BST + e0), 16, 0, 0);
    }
  };
  auto stage_a = [&](int buf) __attribute__((always_inline)) {
    unsigned short *dst = s_A + (size_t)buf * A_STAGE + (size_t)ar * LDA + ah * 8;
    uint2 c1[2], c2[2], c3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) split3x4(pa[j], c1[j], c2[j], c3[j]);
    *reinterpret_cast<uint4 *>(dst) = make_uint4(c1[0].x, c1[0].y, c1[1].x, c1[1].y);
    *reinterpret_cast<uint4 *>(dst + HTM * LDA) = make_uint4(c2[0].x, c2[0].y, c2[1].x, c2[1].y);
    *reinterpret_cast<uint4 *>(dst + 2 * HTM * LDA) = make_uint4(c3[0].x, c3[0].y, c3[1].x, c3[1].y);
  };
  auto stage_sync = [&]() __attribute__((always_inline)) {  // LDS-DMA counts on vmcnt
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  prefetch_a(0);
  dma_b(0, 0);
  stage_a(0);
  stage_sync();
  f32x16 acc[2][NC], acc2[SC ? 2 : 1][SC ? NC : 1];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int cb = 0; cb < NC; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[rb][cb][r] = 0.f;
        if (SC) acc2[rb][cb][r] = 0.f;
      }
  auto mfma_chunk = [&](int buf, f32x16 (&ac)[2][NC]) __attribute__((always_inline)) {
    const unsigned short *abase = s_A + (size_t)buf * A_STAGE + (size_t)(wr * 64 + (lane & 31)) * LDA + 8 * (lane >> 5);
    const uint4 *bbase = s_B + (size_t)buf * BST + (size_t)(wc * NC) * (2 * 3 * 64) + lane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[2][3], b[NC][3];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[rb][p] = *reinterpret_cast<const bf16x8 *>(abase + (size_t)p * HTM * LDA + (size_t)rb * 32 * LDA + ks * 16);
#pragma unroll
      for (int cb = 0; cb < NC; ++cb)
#pragma unroll
        for (int p = 0; p < 3; ++p) b[cb][p] = __builtin_bit_cast(bf16x8, bbase[((cb * 2 + ks) * 3 + p) * 64]);
#define DH3D_X6_PRODUCT(PA, PB)                                                                          \
  _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) _Pragma("unroll") for (int cb = 0; cb < NC; ++cb)     \
      ac[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rb][PA], b[cb][PB], ac[rb][cb], 0, 0, 0);
      DH3D_X6_PRODUCT(2, 0) DH3D_X6_PRODUCT(0, 2) DH3D_X6_PRODUCT(1, 1)
      DH3D_X6_PRODUCT(1, 0) DH3D_X6_PRODUCT(0, 1) DH3D_X6_PRODUCT(0, 0)
#undef DH3D_X6_PRODUCT
    }
  };
  for (int ch = 0; ch < NCH; ++ch) {
    const int nxt = ch + 1 < NCH ? ch + 1 : ch;  // unconditional prefetch: the last one is a harmless repeat
    prefetch_a(nxt);
    const int buf = ch & 1;
    dma_b(nxt, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SC) {
      if (ch < NCH_MAIN) mfma_chunk(buf, acc);
      else mfma_chunk(buf, acc2);
    } else {
      mfma_chunk(buf, acc);
    }
    stage_a(buf ^ 1);
    stage_sync();
  }
  // epilogue: bias + BN + activation in registers -> tile in LDS -> full-row 16-byte stores (+ residual)
#pragma unroll
  for (int cb = 0; cb < NC; ++cb) {
    const int col = (wc * NC + cb) * 32 + (lane & 31);
    float pb = 0.f, scl = 1.f, sh = 0.f, pb2 = 0.f, scl2 = 1.f, sh2 = 0.f;
    if (ep.pre_bias) pb = ep.pre_bias[col];
    if (ep.scale) scl = ep.scale[col];
    if (ep.shift) sh = ep.shift[col];
    if (SC) {
      if (sc.ep.pre_bias) pb2 = sc.ep.pre_bias[col];
      if (sc.ep.scale) scl2 = sc.ep.scale[col];
      if (sc.ep.shift) sh2 = sc.ep.shift[col];
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float y = dh3d_act((acc[rb][cb][r] + pb) * scl + sh, ep.act);
        if (SC) y += dh3d_act((acc2[rb][cb][r] + pb2) * scl2 + sh2, sc.ep.act);
        s_out[(size_t)(wr * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDO + col] = y;
      }
  }
  __syncthreads();
  if (NC == 1 && l2.out) {
    // residual into the tile, row norms (4 threads per row), then one flat coalesced store of rows x 131 floats
    float *s_inv = s_out + (size_t)HTM * LDO;
    {
      const int p = tid >> 2, q = tid & 3;
      const long long g = grow0 + p;
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < TN / 16; ++i) {
        const int c4 = (i * 4 + q) * 4;
        float4 v = *reinterpret_cast<const float4 *>(s_out + (size_t)p * LDO + c4);
        if (residual && g < R) {
          const float4 rq = *reinterpret_cast<const float4 *>(residual + g * TN + c4);
          v.x += rq.x; v.y += rq.y; v.z += rq.z; v.w += rq.w;
          *reinterpret_cast<float4 *>(s_out + (size_t)p * LDO + c4) = v;
        }
        ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
      }
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      if (q == 0) s_inv[p] = rsqrtf(fmaxf(ss, l2.eps));
    }
    __syncthreads();
    constexpr int W = 3 + TN;
    const long long rows = R - grow0 < HTM ? R - grow0 : HTM;
    float *o = l2.out + grow0 * W;
    const float *pf = l2.prefix + grow0 * 3;
    for (int e = tid; e < (int)rows * W; e += 512) {
      const int p = e / W, c = e - p * W;
      o[e] = c < 3 ? pf[p * 3 + c] : s_out[(size_t)p * LDO + (c - 3)] * s_inv[p];
    }
    return;
  }
  constexpr int CV = TN / 4;
  for (int e = tid; e < HTM * CV; e += 512) {
    const int p = e / CV, c4 = (e - p * CV) * 4;
    const long long g = grow0 + p;
    if (g < R) {
      float4 v = *reinterpret_cast<const float4 *>(s_out + (size_t)p * LDO + c4);
      if (residual) {
        const float4 q = *reinterpret_cast<const float4 *>(residual + g * TN + c4);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      }
      *reinterpret_cast<float4 *>(out + g * TN + c4) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 64 -> 128 (the local step's two full-resolution 1x1 convs: 65536 rows, K = 64) without LDS, barriers or a K pipeline.
// linear_x6_kernel is built for long K: 110 KB of staging per 128-row workgroup = ONE workgroup per CU, and with two
// K-chunks its prologue / epilogue latency is the whole kernel (18.6 us for 50 MB).  Here a wave owns 32 rows x 128
// columns: every lane loads the 8 floats per K-step its MFMA A operand wants straight from the row (32 B pieces, the
// whole 256 B row over the four K-steps), splits them in registers, takes the weight fragments (packed in fragment
// order, 48 KB, L2-resident) one K-step ahead, and stores from the accumulators -- 128 B row segments per column block.
template <bool RES, int NCB, int ACT>  // ACT: DH3D_ACT_NONE / DH3D_ACT_RELU at compile time, -1 = read ep.act
__global__ __launch_bounds__(256) void linear_k64_x6_kernel(const float *__restrict__ x, const uint4 *__restrict__ wp,
                                                            EpilogueArgs ep, const float *__restrict__ residual,
                                                            long long R, float *__restrict__ out) {
  constexpr int C = 64, KB = C / 16, DOUT = 128, WPR = 4 / NCB;  // WPR waves share a 32-row tile (column blocks each)
  const int lane = threadIdx.x & 63, half = lane >> 5, lr = lane & 31;
  const int wave = threadIdx.x >> 6, cb0 = (wave % WPR) * NCB;
  const long long row0 = ((long long)blockIdx.x * NCB + wave / WPR) * 32;
  if (row0 >= R) return;
  long long arow = row0 + lr;
  if (arow >= R) arow = R - 1;  // rows past R repeat the last row; they are not stored
  const float4 *ap = reinterpret_cast<const float4 *>(x + arow * C + 8 * half);
  float4 av[KB][2];
#pragma unroll
  for (int ks = 0; ks < KB; ++ks) {
    av[ks][0] = ap[ks * 4];
    av[ks][1] = ap[ks * 4 + 1];
  }
  uint4 bq[2][NCB][3];  // this K-step's fragments and the next one's (all four at once would be 192 VGPRs: one wave per SIMD)
  auto load_b = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[ks & 1][cb][p] = wp[((size_t)((cb0 + cb) * KB + ks) * 3 + p) * 64 + lane];
  };
  load_b(0);
  float pb[NCB], scl[NCB], sh[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int col = (cb0 + cb) * 32 + lr;
    pb[cb] = ep.pre_bias ? ep.pre_bias[col] : 0.f;
    scl[cb] = ep.scale ? ep.scale[col] : 1.f;
    sh[cb] = ep.shift ? ep.shift[col] : 0.f;
  }
  f32x16 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < KB; ++ks) {
    if (ks + 1 < KB) load_b(ks + 1);
    asm volatile("" ::: "memory");  // (keeps the compiler from issuing every K-step's fragments up front ...
    __builtin_amdgcn_sched_barrier(0);  //  ... and from sinking them between this K-step's products: then the next step waits on the last of them)
    uint2 c1[2], c2[2], c3[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) split3x4(av[ks][j], c1[j], c2[j], c3[j]);
    bf16x8 a[3];
    a[0] = __builtin_bit_cast(bf16x8, make_uint4(c1[0].x, c1[0].y, c1[1].x, c1[1].y));
    a[1] = __builtin_bit_cast(bf16x8, make_uint4(c2[0].x, c2[0].y, c2[1].x, c2[1].y));
    a[2] = __builtin_bit_cast(bf16x8, make_uint4(c3[0].x, c3[0].y, c3[1].x, c3[1].y));
#define DH3D_K64_PRODUCT(PA, PB)                                                                      \
  _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16( \
      a[PA], __builtin_bit_cast(bf16x8, bq[ks & 1][cb][PB]), acc[cb], 0, 0, 0);
    DH3D_K64_PRODUCT(2, 0) DH3D_K64_PRODUCT(0, 2) DH3D_K64_PRODUCT(1, 1)
    DH3D_K64_PRODUCT(1, 0) DH3D_K64_PRODUCT(0, 1) DH3D_K64_PRODUCT(0, 0)
#undef DH3D_K64_PRODUCT
  }
  // epilogue from the accumulators: two rows at a time in packed f32 (a lane's column is fixed per block), one pointer
  // per column block with the row offsets as immediates
  typedef float v2f __attribute__((ext_vector_type(2)));
  const int act = ACT >= 0 ? ACT : ep.act;
  auto store_tile = [&](auto ragged) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const long long e0 = (row0 + 4 * half) * DOUT + (cb0 + cb) * 32 + lr;
      float *op = out + e0;
      const float *rp = RES ? residual + e0 : nullptr;
      const v2f pb2 = {pb[cb], pb[cb]}, sc2 = {scl[cb], scl[cb]}, sh2 = {sh[cb], sh[cb]};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        v2f y = {acc[cb][r], acc[cb][r + 1]};
        y = __builtin_elementwise_fma(y + pb2, sc2, sh2);
        if (ACT == DH3D_ACT_RELU) y = __builtin_elementwise_max(y, v2f{0.f, 0.f});
        else if (ACT < 0) y = v2f{dh3d_act(y[0], act), dh3d_act(y[1], act)};
        const int t0 = (r & 3) + 8 * (r >> 2);  // rows t0 + 4 half and the next one of the tile
        if (!decltype(ragged)::value || row0 + 4 * half + t0 < R) op[t0 * DOUT] = RES ? y[0] + rp[t0 * DOUT] : y[0];
        if (!decltype(ragged)::value || row0 + 4 * half + t0 + 1 < R)
          op[(t0 + 1) * DOUT] = RES ? y[1] + rp[(t0 + 1) * DOUT] : y[1];
      }
    }
  };
  if (row0 + 32 <= R) store_tile(std::false_type{});
  else store_tile(std::true_type{});
}

// ------------------------------------------------------------------------------------------------
// The attention head on an UP-SAMPLED input, with the wide 1x1 conv commuted through the interpolation.
// globalatt_block (core/backbones.py:156-173) is  sigmoid(w_fc . relu(BN(W x + b)) + b_fc)  on x = the 3-NN inverse-
// distance interpolation of the N/8-level features (backbones.py:91-95).  Interpolation and conv are both linear and
// the weights sum to one, so  W interp(x) + b = interp(W x) + b:  the 256 -> 1024 GEMM -- the largest dense op of the
// global path -- runs on the m = n/8 coarse rows (8x fewer flops; linear_x6 slices H[j] = x W[:, 256 j : 256 j + 256])
// and this kernel interpolates the 1024-wide rows instead: a wave per fine point, lane l = channels 4 l .. 4 l + 3 of
// each 256-slice, three gathered rows (12 KB per point from L2 -- a cloud's H is 2 MB), bias + BN + ReLU + the dot with
// w_fc in registers, one wave reduction.  Same value up to the rounding of a different association.
__global__ __launch_bounds__(256) void interp_head_kernel(const float *__restrict__ H, int NS, long long Rc,
                                                         const int32_t *__restrict__ idx,
                                                         const float *__restrict__ dist, int n, int m, EpilogueArgs ep,
                                                         const float *__restrict__ w_fc, float b_fc, long long R,
                                                         float *__restrict__ att) {
  constexpr int MAXS = 4;  // 256-channel slices (hidden width <= 1024)
  const int lane = threadIdx.x & 63;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
  // this lane's channels: slice j, columns 4*lane .. 4*lane+3
  float4 pb[MAXS], sc[MAXS], sh[MAXS], wf[MAXS];
#pragma unroll
  for (int j = 0; j < MAXS; ++j) {
    const int c = j * 256 + lane * 4;
    pb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    sc[j] = make_float4(1.f, 1.f, 1.f, 1.f);
    sh[j] = pb[j];
    wf[j] = pb[j];
    if (j < NS) {
      if (ep.pre_bias) pb[j] = *reinterpret_cast<const float4 *>(ep.pre_bias + c);
      if (ep.scale) sc[j] = *reinterpret_cast<const float4 *>(ep.scale + c);
      if (ep.shift) sh[j] = *reinterpret_cast<const float4 *>(ep.shift + c);
      wf[j] = *reinterpret_cast<const float4 *>(w_fc + c);
    }
  }
  // Workgroups are dispatched round-robin over the 8 XCDs, each with its own 4 MB L2, and a cloud's H is 2 MB: XCD x
  // takes clouds x, x + 8, ... one after the other, all its waves on the same cloud at a time (with waves simply walking
  // the rows in order, every L2 saw eight clouds at once and the gathers went to the far cache: 176 us).
  const int xcd = blockIdx.x & 7;
  const long long wx = (long long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6), nwx = (long long)(gridDim.x >> 3) * 4;
  const int B = (int)(R / n);
  const int per = (int)((n + nwx - 1) / nwx);
  (void)wid; (void)nw;
  for (int bi = xcd; bi < B; bi += 8) {
    const int q0 = (int)(wx * per), q1 = q0 + per < n ? q0 + per : n;
    if (q0 >= n) continue;
    long long r = (long long)bi * n + q0;
    int i1 = idx[r * 3], i2 = idx[r * 3 + 1], i3 = idx[r * 3 + 2];
    float d1 = dist[r * 3], d2 = dist[r * 3 + 1], d3 = dist[r * 3 + 2];
    for (int q = q0; q < q1; ++q, ++r) {
    const float *p1 = H + ((long long)bi * m + i1) * 256 + lane * 4;
    const float *p2 = H + ((long long)bi * m + i2) * 256 + lane * 4;
    const float *p3 = H + ((long long)bi * m + i3) * 256 + lane * 4;
    float w1, w2, w3;
    idw_weights(d1, d2, d3, w1, w2, w3);
    float4 a[MAXS], b[MAXS], c[MAXS];
#pragma unroll
    for (int j = 0; j < MAXS; ++j)
      if (j < NS) {
        a[j] = *reinterpret_cast<const float4 *>(p1 + (size_t)j * Rc * 256);
        b[j] = *reinterpret_cast<const float4 *>(p2 + (size_t)j * Rc * 256);
        c[j] = *reinterpret_cast<const float4 *>(p3 + (size_t)j * Rc * 256);
      }
    if (q + 1 < q1) {  // the next point's indices ride behind this point's rows
      i1 = idx[(r + 1) * 3]; i2 = idx[(r + 1) * 3 + 1]; i3 = idx[(r + 1) * 3 + 2];
      d1 = dist[(r + 1) * 3]; d2 = dist[(r + 1) * 3 + 1]; d3 = dist[(r + 1) * 3 + 2];
    }
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < MAXS; ++j)
      if (j < NS) {
        const float4 v = idw_mix(a[j], b[j], c[j], w1, w2, w3);
        z = fmaf(dh3d_act((v.x + pb[j].x) * sc[j].x + sh[j].x, ep.act), wf[j].x, z);
        z = fmaf(dh3d_act((v.y + pb[j].y) * sc[j].y + sh[j].y, ep.act), wf[j].y, z);
        z = fmaf(dh3d_act((v.z + pb[j].z) * sc[j].z + sh[j].z, ep.act), wf[j].z, z);
        z = fmaf(dh3d_act((v.w + pb[j].w) * sc[j].w + sh[j].w, ep.act), wf[j].w, z);
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
    if (lane == 0) att[r] = 1.f / (1.f + expf(-(z + b_fc)));
    }
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

#ifdef DH3D_DEV  // dev builds only (tools/dense_bench.py): column waves of the head GEMM (4 = eight waves, 2 = four)
static int g_head_wc = 4;
DH3D_API void dh3d_dev_set_head_wc(int wc) { g_head_wc = wc == 2 ? 2 : 4; }
#else
static constexpr int g_head_wc = 4;
#endif

DH3D_API int dh3d_mlp_head_pm_x6_fwd(const float *h, int R, int C, const void *wpacked_x3, int H,
                                     const dh3d_epilogue *ep, const float *w_fc, float b_fc, float *att,
                                     void *stream) {
  DH3D_REQUIRE(h && wpacked_x3 && w_fc && att && R > 0 && C > 0 && H > 0);
  DH3D_SUPPORTED(C % HKC == 0 && H % HTN == 0);
  const size_t lds = (size_t)2 * A_STAGE * 2 + (size_t)2 * B_STAGE * 16 + sizeof(float) * 4 * HTM;
  if (g_head_wc == 2) {
    auto kern = mlp_head_x6_kernel<2>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, dim3(dh3d_cdiv(R, HTM)), dim3(256), lds, (hipStream_t)stream, h, C,
                       static_cast<const uint4 *>(wpacked_x3), H, dh3d_ep(ep), w_fc, b_fc, (long long)R, att);
  } else {
    auto kern = mlp_head_x6_kernel<4>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, dim3(dh3d_cdiv(R, HTM)), dim3(512), lds, (hipStream_t)stream, h, C,
                       static_cast<const uint4 *>(wpacked_x3), H, dh3d_ep(ep), w_fc, b_fc, (long long)R, att);
  }
  return dh3d_launch_status();
}

static int linear_x6_launch(const float *x1, int C1, const float *x2, int C2, const void *wpacked_x3, int R, int Dout,
                            const dh3d_epilogue *ep, const float *residual, float *out, void *stream,
                            const UpsampleSrc &up, const L2CatOut &l2 = L2CatOut{nullptr, nullptr, 0.f},
                            const ShortcutSrc &sc = ShortcutSrc{nullptr, 0, EpilogueArgs{nullptr, nullptr, nullptr, 0}},
                            int slices = 1) {
  DH3D_SUPPORTED(C1 % HKC == 0 && C2 % HKC == 0 && sc.C3 % HKC == 0 && (Dout == 128 || Dout == 256));
  const EpilogueArgs e = dh3d_ep(ep);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(dh3d_cdiv(R, HTM), slices), block(512);
  const uint4 *wp = static_cast<const uint4 *>(wpacked_x3);
  if (C1 == 64 && C2 == 0 && Dout == 128 && slices == 1 && !up.points && !l2.out && !sc.x3) {
#ifndef DH3D_K64_NCB
#define DH3D_K64_NCB 4
#endif
    constexpr int kNcb = DH3D_K64_NCB;  // column blocks per wave
    const dim3 g64(dh3d_cdiv(R, 32 * kNcb));
#define DH3D_K64_LAUNCH(RES, ACT) \
  hipLaunchKernelGGL((linear_k64_x6_kernel<RES, kNcb, ACT>), g64, dim3(256), 0, s, x1, wp, e, residual, (long long)R, out)
#define DH3D_K64_ACT(RES)                                                \
  {                                                                      \
    if (e.act == DH3D_ACT_RELU) DH3D_K64_LAUNCH(RES, DH3D_ACT_RELU);     \
    else if (e.act == DH3D_ACT_NONE) DH3D_K64_LAUNCH(RES, DH3D_ACT_NONE); \
    else DH3D_K64_LAUNCH(RES, -1);                                       \
  }
    if (residual) DH3D_K64_ACT(true) else DH3D_K64_ACT(false)
#undef DH3D_K64_ACT
#undef DH3D_K64_LAUNCH
    return dh3d_launch_status();
  }
  const long long wslice = (long long)3 * (C1 + C2 + sc.C3) * Dout * 2 / 16;  // uint4 per packed slice
  const long long oslice = (long long)R * Dout;
  if (Dout == 128) {
    size_t lds = (size_t)2 * A_STAGE * 2 + (size_t)2 * (128 / 32) * 2 * 3 * 64 * 16;
    const size_t tile = sizeof(float) * HTM * (128 + 4) + sizeof(float) * HTM;  // + row norms (L2CatOut)
    if (tile > lds) lds = tile;
    if (sc.x3) {
      auto kern = linear_x6_kernel<1, true>;
      DH3D_ALLOW_BIG_LDS(kern);
      hipLaunchKernelGGL(kern, grid, block, lds, s, x1, C1, x2, C2, wp, e, residual, (long long)R, out, up, l2, sc, wslice,
                         oslice);
    } else {
      auto kern = linear_x6_kernel<1, false>;
      DH3D_ALLOW_BIG_LDS(kern);
      hipLaunchKernelGGL(kern, grid, block, lds, s, x1, C1, x2, C2, wp, e, residual, (long long)R, out, up, l2, sc, wslice,
                         oslice);
    }
  } else {
    size_t lds = (size_t)2 * A_STAGE * 2 + (size_t)2 * (256 / 32) * 2 * 3 * 64 * 16;
    const size_t tile = sizeof(float) * HTM * (256 + 4);
    if (tile > lds) lds = tile;
    if (l2.out || sc.x3) return DH3D_ERR_UNSUPPORTED;
    auto kern = linear_x6_kernel<2, false>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, grid, block, lds, s, x1, C1, x2, C2, wp, e, residual, (long long)R, out, up, l2, sc, wslice,
                         oslice);
  }
  return dh3d_launch_status();
}

DH3D_API int dh3d_linear_pm_x6_fwd(const float *x1, int C1, const float *x2, int C2, const void *wpacked_x3, int R,
                                   int Dout, const dh3d_epilogue *ep, const float *residual, float *out,
                                   void *stream) {
  DH3D_REQUIRE(x1 && wpacked_x3 && out && R > 0 && C1 > 0 && C2 >= 0 && Dout > 0 && (C2 == 0 || x2));
  const UpsampleSrc none{nullptr, nullptr, nullptr, 1, 1};
  return linear_x6_launch(x1, C1, x2, C2, wpacked_x3, R, Dout, ep, residual, out, stream, none);
}

DH3D_API int dh3d_upsample_linear_pm_x6_fwd(const float *points, const int32_t *idx, const float *dist, int B, int n,
                                            int m, int C1, const float *x2, int C2, const void *wpacked_x3, int Dout,
                                            const dh3d_epilogue *ep, const float *residual, float *out, void *stream) {
  DH3D_REQUIRE(points && idx && dist && wpacked_x3 && out && B > 0 && n > 0 && m > 0 && C1 > 0 && C2 >= 0 &&
               Dout > 0 && (C2 == 0 || x2));
  DH3D_REQUIRE((long long)B * n < (1LL << 31));
  const UpsampleSrc up{points, idx, dist, n, m};
  return linear_x6_launch(points, C1, x2, C2, wpacked_x3, B * n, Dout, ep, residual, out, stream, up);
}

// The same, finishing the local path in its store: out_cat [B*n, 3 + 128] = [prefix | l2_normalize(y, eps)]
// (core/model.py:177-181); Dout == 128.  y itself is not written.
DH3D_API int dh3d_upsample_linear_l2cat_pm_x6_fwd(const float *points, const int32_t *idx, const float *dist, int B,
                                                  int n, int m, int C1, const float *x2, int C2,
                                                  const void *wpacked_x3, int Dout, const dh3d_epilogue *ep,
                                                  const float *residual, const float *prefix, float l2_eps,
                                                  float *out_cat, void *stream) {
  DH3D_REQUIRE(points && idx && dist && wpacked_x3 && out_cat && prefix && B > 0 && n > 0 && m > 0 && C1 > 0 &&
               C2 >= 0 && (C2 == 0 || x2));
  DH3D_SUPPORTED(Dout == 128);
  DH3D_REQUIRE((long long)B * n < (1LL << 31));
  const UpsampleSrc up{points, idx, dist, n, m};
  const L2CatOut l2{out_cat, prefix, l2_eps};
  return linear_x6_launch(points, C1, x2, C2, wpacked_x3, B * n, Dout, ep, residual, out_cat, stream, up, l2);
}

// out = act(BN([upsample(points) | x2] W)) + act_sc(BN_sc(x3 W_sc)): the up-sampling concat conv with the backbone's
// shortcut conv (core/backbones.py:123) computed in the same kernel instead of written and read back.  wpacked_x3 =
// dh3d_pack_weight_x3 of [W; W_sc] stacked ([C1 + C2 + C3, Dout]), Dout == 128.  prefix != NULL: `out` is
// [B*n, 3 + 128] = [prefix | l2_normalize(sum, l2_eps)] as in dh3d_upsample_linear_l2cat_pm_x6_fwd.
DH3D_API int dh3d_upsample_linear_shortcut_pm_x6_fwd(const float *points, const int32_t *idx, const float *dist, int B,
                                                     int n, int m, int C1, const float *x2, int C2, const float *x3,
                                                     int C3, const void *wpacked_x3, int Dout,
                                                     const dh3d_epilogue *ep, const dh3d_epilogue *ep_shortcut,
                                                     const float *prefix, float l2_eps, float *out, void *stream) {
  DH3D_REQUIRE(points && idx && dist && wpacked_x3 && out && x3 && B > 0 && n > 0 && m > 0 && C1 > 0 && C2 >= 0 &&
               C3 > 0 && (C2 == 0 || x2));
  DH3D_SUPPORTED(Dout == 128);
  DH3D_REQUIRE((long long)B * n < (1LL << 31));
  const UpsampleSrc up{points, idx, dist, n, m};
  const L2CatOut l2{prefix ? out : nullptr, prefix, l2_eps};
  const ShortcutSrc sc{x3, C3, dh3d_ep(ep_shortcut)};
  return linear_x6_launch(points, C1, x2, C2, wpacked_x3, B * n, Dout, ep, nullptr, out, stream, up, l2, sc);
}

// A layer wider than 256 as `slices` column slices of 256 in ONE launch: out [slices][R][256] = x1 @ W[:, 256 j : 256 j + 256],
// wpacked_x3 = the dh3d_pack_weight_x3 images of the slices one after the other.  No epilogue, no residual.
DH3D_API int dh3d_linear_slices_pm_x6_fwd(const float *x1, int C1, const void *wpacked_x3, int R, int slices,
                                          float *out, void *stream) {
  DH3D_REQUIRE(x1 && wpacked_x3 && out && R > 0 && C1 > 0 && slices > 0);
  DH3D_SUPPORTED(slices <= 65535);
  const UpsampleSrc none{nullptr, nullptr, nullptr, 1, 1};
  return linear_x6_launch(x1, C1, nullptr, 0, wpacked_x3, R, 256, nullptr, nullptr, out, stream, none,
                          L2CatOut{nullptr, nullptr, 0.f},
                          ShortcutSrc{nullptr, 0, EpilogueArgs{nullptr, nullptr, nullptr, 0}}, slices);
}

// att [B*n] = sigmoid(w_fc . act(BN(interp(H) + pre_bias)) + b_fc), H = the hidden layer WITHOUT bias computed on the
// coarse level: [Hd/256][B*m][256] (one dh3d_linear_pm_x6_fwd per 256-column slice of the weight), idx / dist [B*n, 3]
// from dh3d_three_nn (squared distances), Hd in {256, 512, 768, 1024}.
// ---------------------------------------------------------------------------------------------------------------
// The same head with the fine points walked in MORTON order and the coarse rows staged in LDS (round 2).
// interp_head_kernel moves 12 KB per fine point through the L2 (1.6 GB per cfg-3 step, 11.6 TB/s: L2-bound).  128
// consecutive points of the Morton order are a compact region whose 384 neighbour references hit only ~35-45 DISTINCT
// coarse rows: a workgroup finds them with a bitmap (one bit per coarse row of the cloud, prefix popcounts = slots),
// and for each 256-channel slice stages those rows ONCE (1 KB each, <= 64 slots = 64 KB, so two workgroups share a CU)
// and lets its eight waves read them from LDS -- ~10x less L2 traffic.  The 64 lane partials of a point and slice are
// summed right away (two points per butterfly) into a per-point logit in LDS: a plain loop over the points, small code
// and few registers (keeping 32 per-lane partials in registers across the slices, fully unrolled, cost 296-512 VGPRs
// and, with the generic activation inlined 256 times, twice the instruction cache).  (The first attempt,
// DESIGN.md dead end (j), staged whole 4 KB rows: 32 slots did not hold a block's rows and one workgroup filled a CU.)
// Rows beyond the slot capacity (never seen on uniform clouds) are read from global memory like before.
constexpr int kIHP = 128;    // fine points per workgroup
constexpr int kIHCap = 64;   // staged coarse rows per slice: a 128-point block touches 46 distinct rows on average, 62 at most (uniform clouds); 48 slots overflowed in 1/3-2/3 of the blocks
constexpr int kIHW = 8;             // waves per workgroup (two workgroups per CU: four waves per SIMD)
constexpr int kIHT = kIHW * 64;      // threads
constexpr int kIHPW = kIHP / kIHW;   // points per wave
constexpr int kIHLst = kIHCap + 1 + 3 * kIHP;  // slot-major reference lists: [kIHCap + 1] offsets, [3 * kIHP] entries (point * 4 + t)
constexpr int kIHPlan = (kIHP * 4 * 2 + kIHP + 32 + 33 + kIHCap + kIHLst + 3) / 4 * 4;  // dwords of the plan image (1732)
constexpr int kIHTab = kIHPlan + (kIHP /*s_z*/ + kIHP /*s_inv*/ + 64 + 3) / 4 * 4;  // floats of tables

// VLAD (the global descriptor path): the same walk continues into NetVLAD's soft assignment
// (core/backbones.py:207-255) -- the up-sampled feature map is never built.  With x[n] = sum_t w_t c[i_t] (c = the
// coarse rows) and xn = x / |x|:
//   logits   s[n,:] = xn[n] Wc = (1/|x[n]|) * sum_t w_t (c Wc)[i_t]        -> interpolate the rows of  cw = c Wc  (64 wide)
//   VLAD     V[k,d] = sum_n a[n,k] xn[n,d] = sum_j A'[j,k] c[j,d],   A'[j,k] = sum_{(n,t): i_t = j} a[n,k] w_t / |x[n]|
// so after the attention logit of a point the kernel (a) mixes the point's three coarse rows once more for |x|,
// (b) mixes three cw rows for the logits, BatchNorm + softmax over the 64 lanes, times the attention just computed,
// (c) adds a * w_t / |x| into A' rows -- in LDS for the staged rows, flushed with 64-lane f32 atomics at the end -- and
// the caller finishes with a [m x 64]^T [m x 256] GEMM per cloud on the COARSE rows.  Replaces three_interpolate (134
// MB written and read back at cfg 3) and the per-point part of netvlad_assign_accumulate; the f32 atomics make the
// global descriptor reproducible to ~1e-7 instead of bit for bit.
// a staged row (slot >= 0) from LDS, or -- only in blocks that exceeded the slot capacity (OVF) -- row -1-slot from
// global memory.  The common case has no branch at all: a taken scalar branch costs ~35 cycles and there would be
// three per point and slice.
template <bool OVF>
__device__ __forceinline__ float4 ih_row4(const float *s_rows, const float *gbase, int slot, int lane, int rs = 256) {
  if (OVF && slot < 0) return *reinterpret_cast<const float4 *>(gbase + (size_t)(-1 - slot) * rs + lane * 4);
  return *reinterpret_cast<const float4 *>(s_rows + (size_t)slot * 256 + lane * 4);
}

// Row requests go through buffer loads: a scalar resource (base of the slice) + ONE 32-bit byte offset per row in a VGPR.
// (As flat 64-bit addresses the eight row offsets took sixteen registers and were spilled around the slice loop.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ih_rsrc(const float *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, -1, 0x00020000);  // raw, no bounds clamp
}

// A wave-uniform slot: through the scalar unit only where the overflow test branches on it (a plain LDS address needs a
// VGPR anyway: three v_readfirstlane per point and slice saved in the common case).
template <bool OVF>
__device__ __forceinline__ int ih_uniform(int v) { return OVF ? __builtin_amdgcn_readfirstlane(v) : v; }

// The same row as two packed-f32 pairs, and the inverse-distance mix on them: v_pk_mul_f32 / v_pk_fma_f32 do two lanes'
// worth of idw_mix_fma per issue slot with the same association (bit-equal results).  Only for phases WITHOUT matrix
// work beside them: packed f32 occupies the matrix pipe on gfx950.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Row4 { f32x2 lo, hi; };
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <bool OVF>
__device__ __forceinline__ Row4 ih_row4p(const float *s_rows, const float *gbase, int slot, int lane, int rs = 256) {
  const float4 v = ih_row4<OVF>(s_rows, gbase, slot, lane, rs);
  return Row4{f32x2{v.x, v.y}, f32x2{v.z, v.w}};
}
__device__ __forceinline__ Row4 idw_mix_pk(const Row4 a, const Row4 b, const Row4 c, float w1, float w2, float w3) {
  const f32x2 W1 = {w1, w1}, W2 = {w2, w2}, W3 = {w3, w3};
  return Row4{pk_fma(c.lo, W3, pk_fma(b.lo, W2, a.lo * W1)), pk_fma(c.hi, W3, pk_fma(b.hi, W2, a.hi * W1))};
}

// The block's SLOT TABLE (first kIHPlan dwords of the workgroup's LDS: s_slot | s_w | s_orig | s_bits | s_pre | s_row | lists), built
// from the three_nn result: bitmap of the coarse rows the block's 128 points touch -> prefix popcounts -> slot = rank of the
// row.  Three dependent global round trips (order -> idx / dist -> ...) and five barriers: 17 us of the walk when built in
// the walk's own launch (round 5, tools/walk_phases.sh) -- round 6: walk_plan_kernel builds it once behind three_nn, off the
// critical chain, and the walk copies the image in (dh3d_walk_plan / dh3d_global_walk_planned_fwd).
// LISTS (the plan kernel): also the references to every staged row, slot-major -- s_loff[slot] .. s_loff[slot + 1] index
// entries `point * 4 + t` of s_lst (within a list in the order of the LDS atomics that filled it: the scatter's sums are
// f32 atomics anyway; sorting each list by one thread cost the plan 68 us) -- for the NetVLAD scatter of the walk.
template <bool LISTS>
__device__ __forceinline__ void ih_build_table(float *s_ih, const int32_t *__restrict__ idx, const float *__restrict__ dist,
                                               const float4 *__restrict__ order, int bi, int blk, int n, int m) {
  int *s_slot = reinterpret_cast<int *>(s_ih);
  float *s_w = reinterpret_cast<float *>(s_slot + kIHP * 4);
  int *s_orig = reinterpret_cast<int *>(s_w + kIHP * 4);
  unsigned *s_bits = reinterpret_cast<unsigned *>(s_orig + kIHP);
  int *s_pre = reinterpret_cast<int *>(s_bits + 32);
  int *s_row = s_pre + 33;
  const int tid = threadIdx.x;
  if (tid < 32) s_bits[tid] = 0u;
  __syncthreads();
  // ---- the block's points, their neighbours and weights; mark the coarse rows they touch
  int my_i[3] = {0, 0, 0};
  if (tid < kIHP) {
    const int q = blk * kIHP + tid;
    int orig = -1;
    if (q < n) {
      orig = order ? __float_as_int(order[(size_t)bi * n + q].w) : q;
      const long long r = (long long)bi * n + orig;
      float w1, w2, w3;
      idw_weights(dist[r * 3], dist[r * 3 + 1], dist[r * 3 + 2], w1, w2, w3);
      *reinterpret_cast<float4 *>(s_w + tid * 4) = make_float4(w1, w2, w3, 0.f);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        my_i[t] = idx[r * 3 + t];
        atomicOr(&s_bits[my_i[t] >> 5], 1u << (my_i[t] & 31));
      }
    }
    s_orig[tid] = orig;
    if (orig < 0) {  // padding point of the last block: harmless reads of slot 0 with zero weights
      *reinterpret_cast<float4 *>(s_w + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<int4 *>(s_slot + tid * 4) = make_int4(0, 0, 0, 0);
    }
  }
  __syncthreads();
  if (tid < 64) {  // exclusive prefix of the 32 word popcounts (one wave, shuffle scan)
    int c = tid < 32 ? __popc(s_bits[tid]) : 0, v = c;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const int o = __shfl_up(v, off, 64);
      if ((tid & 63) >= off) v += o;
    }
    if (tid < 32) s_pre[tid] = v - c;
    if (tid == 31) s_pre[32] = v;
  }
  __syncthreads();
  if (tid < kIHP && s_orig[tid] >= 0) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int j = my_i[t];
      const int slot = s_pre[j >> 5] + __popc(s_bits[j >> 5] & ((1u << (j & 31)) - 1u));
      s_slot[tid * 4 + t] = slot < kIHCap ? slot : -1 - j;
    }
  }
  for (int j = tid; j < m; j += kIHT) {  // slot -> coarse row
    if ((s_bits[j >> 5] >> (j & 31)) & 1u) {
      const int slot = s_pre[j >> 5] + __popc(s_bits[j >> 5] & ((1u << (j & 31)) - 1u));
      if (slot < kIHCap) s_row[slot] = j;
    }
  }
  if (LISTS) {
    int *s_loff = s_row + kIHCap, *s_lst = s_loff + kIHCap + 1;
    __shared__ int s_cnt[kIHCap], s_fill[kIHCap];
    if (tid < kIHCap) { s_cnt[tid] = 0; s_fill[tid] = 0; }
    __syncthreads();
    int sl3[3] = {-1, -1, -1};
    if (tid < kIHP && s_orig[tid] >= 0) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        sl3[t] = s_slot[tid * 4 + t];
        if (sl3[t] >= 0) atomicAdd(&s_cnt[sl3[t]], 1);
      }
    }
    __syncthreads();
    if (tid < 64) {  // exclusive prefix of the kIHCap = 64 counts
      const int c = s_cnt[tid];
      int v = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off, 64);
        if (tid >= off) v += o;
      }
      s_loff[tid] = v - c;
      if (tid == 63) s_loff[64] = v;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 3; ++t)
      if (sl3[t] >= 0) s_lst[s_loff[sl3[t]] + atomicAdd(&s_fill[sl3[t]], 1)] = tid * 4 + t;
    __syncthreads();
  }
  __syncthreads();
}

struct VladTail {
  const float *coarse;    // [B, m, 256]
  const float *cw;        // [B, m, 64] = coarse @ cluster_weights
  const float *cl_scale;  // [64] folded cluster BatchNorm
  const float *cl_shift;
  float *apart;           // [B, m, 64]  A' (zeroed by the launcher)
  float *asum;            // [B, 64]     sum_n a[n,:] (zeroed by the launcher)
  const float *b_dev;     // (either instantiation) the fc bias as a device scalar, added to b_fc; may be NULL
  long long h_ss;         // (either) layout of H: slice stride and row stride in floats; 0 = the slice layout
  int h_rs;               //          [NS][Rc][256] (h_ss = Rc * 256, h_rs = 256); row-major [Rc][Hd]: 256, Hd
  const int *plan;        // (either) slot tables of all blocks, [B][nblk][kIHPlan] dwords (dh3d_walk_plan), or NULL: built here
};

__global__ __launch_bounds__(kIHT) void walk_plan_kernel(const int32_t *__restrict__ idx, const float *__restrict__ dist,
                                                         const float4 *__restrict__ order, int n, int m, int nblk,
                                                         int *__restrict__ plan) {
  __shared__ __attribute__((aligned(16))) float s_tab[kIHPlan];
  const int bi = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  ih_build_table<true>(s_tab, idx, dist, order, bi, blk, n, m);
  int4 *dst = reinterpret_cast<int4 *>(plan) + (size_t)(bi * nblk + blk) * (kIHPlan / 4);
  for (int e = tid; e < kIHPlan / 4; e += kIHT) dst[e] = reinterpret_cast<const int4 *>(s_tab)[e];
}

template <bool VLAD>
__global__ __launch_bounds__(kIHT, 2 * kIHW / 4) void interp_head_lds_kernel(const float *__restrict__ H, int NS, long long Rc,
                                                             const int32_t *__restrict__ idx,
                                                             const float *__restrict__ dist,
                                                             const float4 *__restrict__ order, int B, int n, int m,
                                                             int nblk, EpilogueArgs ep, const float *__restrict__ w_fc,
                                                             float b_fc, float *__restrict__ att, VladTail vt) {
  extern __shared__ __attribute__((aligned(16))) float s_ih[];
  // the small tables FIRST: their addresses fit the 16-bit offset field of the ds instructions (behind 64 KB of rows every
  // table read cost a v_add), the rows behind them
  int *s_slot = reinterpret_cast<int *>(s_ih);                    // [kIHP][4] slot (or -1 - coarse row)
  float *s_w = reinterpret_cast<float *>(s_slot + kIHP * 4);      // [kIHP][4] interpolation weights
  int *s_orig = reinterpret_cast<int *>(s_w + kIHP * 4);          // [kIHP] original index of the fine point (-1: none)
  unsigned *s_bits = reinterpret_cast<unsigned *>(s_orig + kIHP); // [32] bitmap over the cloud's coarse rows
  int *s_pre = reinterpret_cast<int *>(s_bits + 32);              // [33] popcount prefix
  int *s_row = s_pre + 33;                                        // [kIHCap] slot -> coarse row
  int *s_loff = s_row + kIHCap;                                   // [kIHCap + 1] (planned walk only) offsets of the slots' reference lists
  int *s_lst = s_loff + kIHCap + 1;                               // [3 kIHP] entries point * 4 + t, slot-major
  float *s_z = s_ih + kIHPlan;                                    // [kIHP] logits, then s_inv [kIHP] and s_asum [64]
  float *s_rows = s_ih + kIHTab;                                  // [kIHCap][256]   (reused by the NetVLAD part)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD x takes clouds x, x + 8, ...: a cloud's H (2 MB) stays in one L2
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int bi = xcd + 8 * (seq / nblk), blk = seq % nblk;
  if (bi >= B) return;
  if (vt.plan) {  // the table was built behind three_nn (walk_plan_kernel): one coalesced copy, one barrier
    const int4 *src = reinterpret_cast<const int4 *>(vt.plan) + (size_t)(bi * nblk + blk) * (kIHPlan / 4);
    for (int e = tid; e < kIHPlan / 4; e += kIHT) reinterpret_cast<int4 *>(s_ih)[e] = src[e];
    __syncthreads();
  } else {
    ih_build_table<false>(s_ih, idx, dist, order, bi, blk, n, m);
  }
  const int nd = min(s_pre[32], kIHCap);
  const bool overflow = s_pre[32] > kIHCap;  // block-uniform: some rows are not staged
  const long long SS = vt.h_ss ? vt.h_ss : Rc * 256;
  const int RS = vt.h_rs ? vt.h_rs : 256;
  const float lo = ep.act == DH3D_ACT_RELU ? 0.f : -__builtin_inff();  // ReLU as a clamp, or no activation
  if (tid < kIHP) s_z[tid] = 0.f;
  // ---- staging: wave w takes slots w, w + 8, ...; the rows of slice sl + 1 are requested BEFORE the points of slice
  // sl are worked on and parked in registers until the buffer is free (staging and per-point work took about the same
  // time when they ran one after the other).  The parked rows are ONE 32-float vector value: as a float4 array the
  // compiler kept them in scratch, stored behind every load and reloaded -- which is why the first attempt at this
  // double buffer measured slower (139 vs 100 us).  Measured by compiling parts out (tools/walk_phases.sh, 32 x 4096):
  // slot table 17 us, the slices 39 (per-point work 36: LDS reads of 3 KB per point and slice are ~20 of it), |x| 9,
  // soft assignment 7, MFMA scatter + atomics 17.
  typedef float f32park __attribute__((ext_vector_type(4 * (kIHCap / kIHW))));
  static_assert(kIHCap % kIHW == 0 && kIHCap / kIHW <= 16, "parked rows: one vector value");
  f32park rg;
  // Row requests: buffer loads off a scalar resource, the byte offset of a row rebuilt from the slot table each time (eight
  // LDS broadcasts per wave and slice; kept in registers the offsets were spilled around the slice loop, and a scratch
  // reload between two requests waits for the request before it: vmcnt counts in issue order).
#define DH3D_IH_REQUEST(BASE, RSTRIDE)                                                          \
  {                                                                                             \
    const __amdgpu_buffer_rsrc_t rs_ = ih_rsrc(BASE);                                           \
    _Pragma("unroll") for (int u = 0; u < kIHCap / kIHW; ++u) {                                 \
      const int r = wave + kIHW * u;                                                            \
      const unsigned off_ = 4u * (unsigned)((bi * m + s_row[r < nd ? r : 0]) * (RSTRIDE) + lane * 4); \
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_, off_, 0, 0);                   \
      rg[4 * u] = __uint_as_float(v.x); rg[4 * u + 1] = __uint_as_float(v.y);                   \
      rg[4 * u + 2] = __uint_as_float(v.z); rg[4 * u + 3] = __uint_as_float(v.w);               \
    }                                                                                           \
  }
  DH3D_IH_REQUEST(H, RS)
  for (int sl = 0; sl < NS; ++sl) {
    const float *Hs = H + (size_t)sl * SS + (size_t)bi * m * RS;
    // The slice's epilogue parameters FIRST, ahead of the next slice's row requests: vmcnt counts in issue order, so
    // issued behind those requests (rounds 3-5) their first use -- the first point of the slice -- waited for the next
    // slice's rows and the double buffer overlapped nothing.
    const int c = sl * 256 + lane * 4;
    // (four loads back to back, no branch between them -- a branch made each wait for the one before; an absent
    // parameter reads w_fc and is replaced by its neutral value)
    float4 pb = *reinterpret_cast<const float4 *>((ep.pre_bias ? ep.pre_bias : w_fc) + c);
    float4 sc = *reinterpret_cast<const float4 *>((ep.scale ? ep.scale : w_fc) + c);
    float4 sh = *reinterpret_cast<const float4 *>((ep.shift ? ep.shift : w_fc) + c);
    const float4 wf = *reinterpret_cast<const float4 *>(w_fc + c);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < kIHCap / kIHW; ++u) {  // all slots, used or not: no branch between the loads and these stores
      const int r = wave + kIHW * u;
      *reinterpret_cast<float4 *>(s_rows + (size_t)r * 256 + lane * 4) =
          make_float4(rg[4 * u], rg[4 * u + 1], rg[4 * u + 2], rg[4 * u + 3]);
    }
    {  // ONE straight-line request whatever comes next (a branch here joins paths with different numbers of loads in
       // flight, and the counter wait behind the join then covers the requests as well): the next slice, or behind the
       // last one the coarse FEATURE rows of the |x| pass below (same slots), or -- attention only -- the last slice again
      const bool more = sl + 1 < NS;
      const float *nb = more ? H + (size_t)(sl + 1) * SS : (VLAD ? vt.coarse : H + (size_t)sl * SS);
      const int nrs = (more || !VLAD) ? RS : 256;
#if !defined(DH3D_IH_SKIP) || !(DH3D_IH_SKIP & 2)
      DH3D_IH_REQUEST(nb, nrs)
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!ep.pre_bias) pb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!ep.scale) sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (!ep.shift) sh = make_float4(0.f, 0.f, 0.f, 0.f);
    sh.x = fmaf(pb.x, sc.x, sh.x); sh.y = fmaf(pb.y, sc.y, sh.y); sh.z = fmaf(pb.z, sc.z, sh.z); sh.w = fmaf(pb.w, sc.w, sh.w);
    const f32x2 sc_lo = {sc.x, sc.y}, sc_hi = {sc.z, sc.w}, sh_lo = {sh.x, sh.y}, sh_hi = {sh.z, sh.w};
    const f32x2 wf_lo = {wf.x, wf.y}, wf_hi = {wf.z, wf.w}, lo2 = {lo, lo};
    __syncthreads();
    // ---- the wave's 16 points, four at a time (a plain loop: small code, bounded registers).  Slots and weights are
    // wave-uniform; only the overflow variant sends the slots through the scalar unit (its test -- a row beyond the slot
    // capacity: read from global memory -- is a scalar branch).  Per point and slice: three 16-byte LDS reads, six packed
    // instructions for the mix, two packed fma + four max + two packed for BatchNorm (pre-bias folded into the shift) +
    // ReLU + the fc dot: 23 VALU instructions where the scalar form had 37 -- which by itself changed nothing (116 us
    // either way): the loop waited for the next slice's rows, see the parameter loads above.  (Keeping the per-lane
    // partial sums in registers across the slices -- one cross-lane reduction per point instead of one per point and
    // slice -- needs the loop fully unrolled: 280 VGPRs, spills.)
    auto slice_points = [&](auto ovf) __attribute__((always_inline)) {
      constexpr bool OVF = decltype(ovf)::value;
#pragma nounroll
      for (int p = 0; p < kIHPW; p += 4) {
        float part[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int pt = wave * kIHPW + p + h;
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const int s0 = ih_uniform<OVF>(si.x), s1 = ih_uniform<OVF>(si.y), s2 = ih_uniform<OVF>(si.z);
          const Row4 v = idw_mix_pk(ih_row4p<OVF>(s_rows, Hs, s0, lane, RS), ih_row4p<OVF>(s_rows, Hs, s1, lane, RS),
                                    ih_row4p<OVF>(s_rows, Hs, s2, lane, RS), sw.x, sw.y, sw.z);  // padding: weight 0
          const f32x2 t0 = __builtin_elementwise_max(pk_fma(v.lo, sc_lo, sh_lo), lo2);
          const f32x2 t1 = __builtin_elementwise_max(pk_fma(v.hi, sc_hi, sh_hi), lo2);
          const f32x2 d = pk_fma(t1, wf_hi, t0 * wf_lo);
          part[h] = d.x + d.y;
        }
        // row sums of two points per reduction: one half-swap + five DPP adds, no LDS crossbar (wave_ops.h)
        const float t01 = pair_wave_sum_f32(part[0], part[1]), t23 = pair_wave_sum_f32(part[2], part[3]);
        if ((lane & 31) == 16) {
          s_z[wave * kIHPW + p + (lane >> 5)] += t01;
          s_z[wave * kIHPW + p + 2 + (lane >> 5)] += t23;
        }
      }
    };
#if !defined(DH3D_IH_SKIP) || !(DH3D_IH_SKIP & 1)   // dev timing knob (tools/walk_phases.sh): results wrong when set
    if (overflow) slice_points(std::true_type{}); else slice_points(std::false_type{});
#endif
    __syncthreads();  // the rows are overwritten by the next slice
  }
  if (tid < kIHP) {
    const float bias = b_fc + (vt.b_dev ? vt.b_dev[0] : 0.f);
    const float a = s_orig[tid] >= 0 ? 1.f / (1.f + expf(-(s_z[tid] + bias))) : 0.f;  // padding points weigh nothing
    if (s_orig[tid] >= 0 && att) att[(size_t)bi * n + s_orig[tid]] = a;
    if (VLAD) s_z[tid] = a;
  }
  if (!VLAD) return;
  // ================= NetVLAD assignment on the same block =================
  float *s_inv = s_z + kIHP;                 // [kIHP] 1 / |x|
  float *s_asum = s_inv + kIHP;              // [64]
  float *s_a = s_rows + 16 * 256;            // [kIHP][64]  a[n, k] = softmax * attention of the block's points (behind the cw rows)
  // ---- (a) |x|: the coarse feature rows, staged like a slice
  {
    const float *Cs = vt.coarse + (size_t)bi * m * 256;
#pragma unroll
    for (int u = 0; u < kIHCap / kIHW; ++u) {  // requested behind the last slice
      const int r = wave + kIHW * u;
      *reinterpret_cast<float4 *>(s_rows + (size_t)r * 256 + lane * 4) =
          make_float4(rg[4 * u], rg[4 * u + 1], rg[4 * u + 2], rg[4 * u + 3]);
    }
    __syncthreads();
    auto norm_points = [&](auto ovf) __attribute__((always_inline)) {
      constexpr bool OVF = decltype(ovf)::value;
#pragma nounroll
      for (int p = 0; p < kIHPW; p += 4) {
        float part[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int pt = wave * kIHPW + p + h;
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const int s0 = ih_uniform<OVF>(si.x), s1 = ih_uniform<OVF>(si.y), s2 = ih_uniform<OVF>(si.z);
          const Row4 v = idw_mix_pk(ih_row4p<OVF>(s_rows, Cs, s0, lane), ih_row4p<OVF>(s_rows, Cs, s1, lane),
                                    ih_row4p<OVF>(s_rows, Cs, s2, lane), sw.x, sw.y, sw.z);
          const f32x2 q = pk_fma(v.hi, v.hi, v.lo * v.lo);
          part[h] = q.x + q.y;
        }
        const float t01 = pair_wave_sum_f32(part[0], part[1]), t23 = pair_wave_sum_f32(part[2], part[3]);
        if ((lane & 31) == 16) {  // tf.nn.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))
          s_inv[wave * kIHPW + p + (lane >> 5)] = rsqrtf(fmaxf(t01, 1e-12f));
          s_inv[wave * kIHPW + p + 2 + (lane >> 5)] = rsqrtf(fmaxf(t23, 1e-12f));
        }
      }
    };
#if !defined(DH3D_GT_SKIP) || !(DH3D_GT_SKIP & 2)   // dev timing knob (tools/gt_bench.py): results wrong when set
    if (overflow) norm_points(std::true_type{}); else norm_points(std::false_type{});
#endif
    __syncthreads();
  }
  // ---- (b, c): cw rows (64 floats each) at the head of the row buffer, A' behind them
  const float *Ws = vt.cw + (size_t)bi * m * 64;
  for (int e = tid; e < nd * 16; e += kIHT) {
    const int r = e >> 4, q = e & 15;
    *reinterpret_cast<float4 *>(s_rows + r * 64 + q * 4) = *reinterpret_cast<const float4 *>(Ws + (size_t)s_row[r] * 64 + q * 4);
  }
  if (tid < 64) s_asum[tid] = 0.f;
  __syncthreads();
  {
    const float csc = vt.cl_scale[lane], csh = vt.cl_shift[lane];
    float *Ab = vt.apart + (size_t)bi * m * 64;
    float asum_acc = 0.f;
    auto assign_points = [&](auto ovf) __attribute__((always_inline)) {
      constexpr bool OVF = decltype(ovf)::value;
#pragma nounroll
      for (int p = 0; p < kIHPW; p += 2) {
        float e2[2];
        int sl_[2][3];
        float cf[2][3];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pt = wave * kIHPW + p + h;
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const float inv = s_inv[pt];
          sl_[h][0] = __builtin_amdgcn_readfirstlane(si.x); sl_[h][1] = __builtin_amdgcn_readfirstlane(si.y);
          sl_[h][2] = __builtin_amdgcn_readfirstlane(si.z);
          cf[h][0] = sw.x * inv; cf[h][1] = sw.y * inv; cf[h][2] = sw.z * inv;
          float mixv = 0.f;
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int sq = sl_[h][t];
            const float cv = (OVF && sq < 0) ? Ws[(size_t)(-1 - sq) * 64 + lane] : s_rows[sq * 64 + lane];
            mixv = fmaf(cf[h][t], cv, mixv);                     // (1/|x|) * sum_t w_t cw[i_t, lane]
          }
          const float z = fmaf(mixv, csc, csh);                  // cluster BatchNorm (folded)
          e2[h] = __expf(z - wave_max_f32(z));
        }
        const float hs = pair_wave_sum_f32(e2[0], e2[1]);
        const float sum0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hs), 16));
        const float sum1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hs), 48));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pt = wave * kIHPW + p + h;
          const float a = e2[h] * (s_z[pt] * __frcp_rn(h == 0 ? sum0 : sum1));  // softmax * attention (0: padding)
          asum_acc += a;
          s_a[pt * 64 + lane] = a;
          if (OVF) {  // rows that did not fit the staging area: straight to memory
#pragma unroll
            for (int t = 0; t < 3; ++t)
              if (sl_[h][t] < 0) unsafeAtomicAdd(&Ab[(size_t)(-1 - sl_[h][t]) * 64 + lane], a * cf[h][t]);
          }
        }
      }
    };
#if !defined(DH3D_GT_SKIP) || !(DH3D_GT_SKIP & 1)
    if (overflow) assign_points(std::true_type{}); else assign_points(std::false_type{});
#endif
    unsafeAtomicAdd(&s_asum[lane], asum_acc);
    __syncthreads();
#if !defined(DH3D_GT_SKIP) || !(DH3D_GT_SKIP & 4)
    if (vt.plan) {
      // Planned walk (round 6): the plan carries, per staged row, the list of the (point, t) references to it.  Wave w owns
      // slots 8w .. 8w + 7 = ONE contiguous run of the slot-major list (~48 references): lane l fetches entry l and its
      // coefficient w_t / |x| in parallel, then a uniform loop broadcasts them (v_readlane) and every lane = cluster adds
      // coef * a[point, lane] -- one LDS read + one fma per reference -- flushing a row with one 64-lane atomic when the slot
      // changes.  (The MFMA form below builds a [64 x 128] selection matrix by compares for 4 of the 8 waves: 12 us of the walk.)
      float *Abp = vt.apart + (size_t)bi * m * 64;
      const int s_first = wave * (kIHCap / kIHW), s_last = min(nd, s_first + kIHCap / kIHW);
      if (s_first < s_last) {  // wave-uniform
        // lane u <= 8 holds the list offset of slot s_first + u (the run's end for slots past nd)
        const int offv = s_loff[min(s_first + (lane & 15), s_last)];
        const int beg = __builtin_amdgcn_readlane(offv, 0);
        const char *ap = reinterpret_cast<const char *>(s_a + lane);
        int key = 0, loaded = -1;
        float cf = 0.f;
        int r0 = 0;  // position in the wave's run of references
        for (int j = s_first; j < s_last; ++j) {
          const int b1 = __builtin_amdgcn_readlane(offv, j - s_first + 1) - beg;
          float acc = 0.f;
          while (r0 < b1) {  // uniform
            const int chunk = r0 >> 6;
            if (chunk != loaded) {  // (once per wave unless its eight rows have more than 64 references)
              const int p = beg + chunk * 64 + lane;
              const int e = s_lst[min(p, 3 * kIHP - 1)];
              key = (e >> 2) * 256;            // byte offset of the point's row of a[., 64]
              cf = s_w[e] * s_inv[e >> 2];     // w_t / |x|
              loaded = chunk;
            }
            const int stop = min(b1 - chunk * 64, 64);
            int r = r0 & 63;
            float acc2 = 0.f;
            for (; r + 1 < stop; r += 2) {  // two references per trip: their LDS reads overlap (the compiler does not unroll this loop)
              const int o0 = __builtin_amdgcn_readlane(key, r), o1 = __builtin_amdgcn_readlane(key, r + 1);
              const float c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cf), r));
              const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cf), r + 1));
              const float a0 = *reinterpret_cast<const float *>(ap + o0), a1 = *reinterpret_cast<const float *>(ap + o1);
              acc = fmaf(c0, a0, acc);
              acc2 = fmaf(c1, a1, acc2);
            }
            if (r < stop) {
              const int o0 = __builtin_amdgcn_readlane(key, r);
              const float c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cf), r));
              acc = fmaf(c0, *reinterpret_cast<const float *>(ap + o0), acc);
            }
            acc += acc2;
            r0 = chunk * 64 + stop;
          }
          unsafeAtomicAdd(&Abp[(size_t)s_row[j] * 64 + lane], acc);
        }
      }
    } else
    // A'[slot, k] = sum_n S[n, slot] * a[n, k] with S[n, slot_t(n)] = w_t(n) / |x_n|: a [64 x 128] x [128 x 64] product on
    // the matrix cores, S built in registers from the slot table (an LDS float atomic per (point, t, cluster) cost 115 us:
    // ds_add_f32 retires roughly one lane every 2.4 cycles).  Wave = one 32x32 tile: slots 32*(wave>>1).., clusters
    // 32*(wave&1)..
    {
      // (of the eight waves the first four -- one per SIMD -- take the four tiles; splitting the points over two copies of
      // the tiles halved each wave's MFMA chain but doubled the atomics: 233 -> 228 us for the whole tail with ONE copy)
#ifndef DH3D_IH_SCATTER_COPIES
#define DH3D_IH_SCATTER_COPIES 1
#endif
      constexpr int kCopies = DH3D_IH_SCATTER_COPIES;
      const int w4 = wave & 3, half = wave >> 2;
      const int ti = w4 >> 1, tj = w4 & 1;
      if (ti * 32 < nd && half < kCopies) {  // block-uniform per wave pair
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int jrow = ti * 32 + (lane & 31), kk = lane >> 5;
#pragma unroll 4
        for (int st = 0; st < kIHP / 2 / kCopies; ++st) {
          const int pt = half * (kIHP / kCopies) + 2 * st + kk;
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const float inv = s_inv[pt];
          const float sv = (si.x == jrow ? sw.x : 0.f) + (si.y == jrow ? sw.y : 0.f) + (si.z == jrow ? sw.z : 0.f);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv * inv, s_a[pt * 64 + tj * 32 + (lane & 31)], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;  // 32x32 accumulator layout
#if !defined(DH3D_GT_SKIP) || !(DH3D_GT_SKIP & 8)
          if (j < nd) unsafeAtomicAdd(&Ab[(size_t)s_row[j] * 64 + tj * 32 + (lane & 31)], acc[r]);
#else
          if (j < nd && acc[r] == 123.456f) Ab[0] = acc[r];
#endif
        }
      }
    }
#endif
    if (tid < 64) unsafeAtomicAdd(&vt.asum[(size_t)bi * 64 + tid], s_asum[tid]);
  }
}

// `order` (may be NULL): the dh3d_spatial_sort records [B,n,4] of the FINE cloud; with it the points are walked in Morton
// order and the coarse rows are staged in LDS (m <= 1024); results equal dh3d_interp_head_fwd up to the summation order
// of the 1024-term row dot.
static size_t interp_head_lds_bytes() {
  return sizeof(float) * ((size_t)kIHCap * 256 + kIHTab);
}

DH3D_API int dh3d_interp_head_sorted_fwd(const float *H, int Hd, const int32_t *idx, const float *dist,
                                         const float *order, int B, int n, int m, const dh3d_epilogue *ep,
                                         const float *w_fc, float b_fc, float *att, void *stream) {
  DH3D_REQUIRE(H && idx && dist && w_fc && att && B > 0 && n > 0 && m > 0 && Hd > 0);
  DH3D_SUPPORTED(Hd % 256 == 0 && Hd <= 1024 && m <= 1024 && (!ep || ep->act != DH3D_ACT_SIGMOID) &&
                 (long long)B * m * Hd < (1ll << 30) /* 32-bit byte offsets of the row requests */);
  const int nblk = dh3d_cdiv(n, kIHP);
  const int per_xcd = dh3d_cdiv(B, 8) * nblk;
  DH3D_ALLOW_BIG_LDS(interp_head_lds_kernel<false>);
  hipLaunchKernelGGL(interp_head_lds_kernel<false>, dim3(8 * per_xcd), dim3(kIHT), interp_head_lds_bytes(),
                     (hipStream_t)stream, H, Hd / 256, (long long)B * m, idx, dist, reinterpret_cast<const float4 *>(order),
                     B, n, m, nblk, dh3d_ep(ep), w_fc, b_fc, att, VladTail{});
  return dh3d_launch_status();
}

// Same with the fc bias read from device memory (a trainable parameter: no host round trip in the training step).
DH3D_API int dh3d_interp_head_sorted_fwd_dev(const float *H, int Hd, int row_major, const int32_t *idx,
                                             const float *dist, const float *order, int B, int n, int m,
                                             const dh3d_epilogue *ep, const float *w_fc, const float *b_fc_dev,
                                             float *att, void *stream) {
  DH3D_REQUIRE(H && idx && dist && w_fc && b_fc_dev && att && B > 0 && n > 0 && m > 0 && Hd > 0);
  DH3D_SUPPORTED(Hd % 256 == 0 && Hd <= 1024 && m <= 1024 && (!ep || ep->act != DH3D_ACT_SIGMOID) &&
                 (long long)B * m * Hd < (1ll << 30) /* 32-bit byte offsets of the row requests */);
  const int nblk = dh3d_cdiv(n, kIHP);
  const int per_xcd = dh3d_cdiv(B, 8) * nblk;
  DH3D_ALLOW_BIG_LDS(interp_head_lds_kernel<false>);
  VladTail vt{};
  vt.b_dev = b_fc_dev;
  if (row_major) { vt.h_ss = 256; vt.h_rs = Hd; }  // H = [B*m][Hd] (one GEMM's output) instead of 256-column slices
  hipLaunchKernelGGL(interp_head_lds_kernel<false>, dim3(8 * per_xcd), dim3(kIHT), interp_head_lds_bytes(),
                     (hipStream_t)stream, H, Hd / 256, (long long)B * m, idx, dist, reinterpret_cast<const float4 *>(order),
                     B, n, m, nblk, dh3d_ep(ep), w_fc, 0.f, att, vt);
  return dh3d_launch_status();
}

// Attention head + NetVLAD soft assignment in one walk over the fine points (see VladTail above).  Outputs: att
// [B,n] (may be NULL) and `accum` = [ apart B*m*64 | asum B*64 | V B*64*256 ] floats, zeroed here in one fill: apart =
// A', asum its column sums, V[b] = apart[b]^T coarse[b] (exact-f32 MFMA GEMM, accumulating).  The caller finishes
// with dh3d_netvlad_tail_fwd(V, asum, ...).
static int global_tail_launch(const float *H, int Hd, const float *coarse, const float *cw, const int32_t *idx,
                              const float *dist, const float *order, int B, int n, int m, const dh3d_epilogue *ep,
                              const float *w_fc, float b_fc, const float *cl_scale, const float *cl_shift, float *att,
                              float *accum, bool zero_here, hipStream_t s, bool with_gemm = true,
                              const int *plan = nullptr) {
  DH3D_REQUIRE(H && coarse && cw && idx && dist && w_fc && cl_scale && cl_shift && accum);
  DH3D_REQUIRE(B > 0 && n > 0 && m > 0 && Hd > 0);
  DH3D_SUPPORTED(Hd % 256 == 0 && Hd <= 1024 && m <= 1024 && (!ep || ep->act != DH3D_ACT_SIGMOID) &&
                 (long long)B * m * Hd < (1ll << 30) /* 32-bit byte offsets of the row requests */);
  float *apart = accum, *asum = apart + (size_t)B * m * 64, *V = asum + (size_t)B * 64;
  if (zero_here &&
      hipMemsetAsync(accum, 0, sizeof(float) * ((size_t)B * m * 64 + (size_t)B * 64 + (with_gemm ? (size_t)B * 64 * 256 : 0)), s) != hipSuccess)
    return DH3D_ERR_LAUNCH;
  const int nblk = dh3d_cdiv(n, kIHP);
  const int per_xcd = dh3d_cdiv(B, 8) * nblk;
  DH3D_ALLOW_BIG_LDS(interp_head_lds_kernel<true>);
  hipLaunchKernelGGL(interp_head_lds_kernel<true>, dim3(8 * per_xcd), dim3(kIHT), interp_head_lds_bytes(), s, H, Hd / 256,
                     (long long)B * m, idx, dist, reinterpret_cast<const float4 *>(order), B, n, m, nblk, dh3d_ep(ep),
                     w_fc, b_fc, att, VladTail{coarse, cw, cl_scale, cl_shift, apart, asum, nullptr, 0, 0, plan});
  const int st = dh3d_launch_status();
  if (st != DH3D_OK || !with_gemm) return st;
  return dh3d_internal_gemm_tn_batched(apart, coarse, B, m, 64, 256, V, true, s);
}

DH3D_API int dh3d_global_tail_fwd(const float *H, int Hd, const float *coarse, const float *cw, const int32_t *idx,
                                  const float *dist, const float *order, int B, int n, int m, const dh3d_epilogue *ep,
                                  const float *w_fc, float b_fc, const float *cl_scale, const float *cl_shift, float *att,
                                  float *accum, void *stream) {
  return global_tail_launch(H, Hd, coarse, cw, idx, dist, order, B, n, m, ep, w_fc, b_fc, cl_scale, cl_shift, att, accum,
                            true, (hipStream_t)stream);
}

// the walk alone: accum = [ apart B*m*64 | asum B*64 ] floats; zero_accum != 0: cleared here, else ZEROED BY THE CALLER (a
// fill issued off the critical chain).  dh3d_netvlad_tail_assign_fwd(apart, coarse, asum, m, ...) finishes (it forms
// V = apart^T coarse inside its finalize kernel).
DH3D_API int dh3d_global_walk_fwd(const float *H, int Hd, const float *coarse, const float *cw, const int32_t *idx,
                                  const float *dist, const float *order, int B, int n, int m, const dh3d_epilogue *ep,
                                  const float *w_fc, float b_fc, const float *cl_scale, const float *cl_shift, float *att,
                                  float *accum, int zero_accum, void *stream) {
  return global_tail_launch(H, Hd, coarse, cw, idx, dist, order, B, n, m, ep, w_fc, b_fc, cl_scale, cl_shift, att, accum,
                            zero_accum != 0, (hipStream_t)stream, false);
}

// The walk's slot tables built ahead of it (round 6): one image of kIHPlan dwords per 128-point block of the Morton order,
// written behind three_nn (off the global step's critical chain) and copied into the walk's LDS with one coalesced read.
// plan: dh3d_walk_plan_bytes(B, n) bytes; valid for the (idx, dist, order) it was built from.
DH3D_API size_t dh3d_walk_plan_bytes(int B, int n) {
  if (B <= 0 || n <= 0) return 0;
  return sizeof(int) * (size_t)B * dh3d_cdiv(n, kIHP) * kIHPlan;
}

DH3D_API int dh3d_walk_plan(const int32_t *idx, const float *dist, const float *order, int B, int n, int m, void *plan,
                            void *stream) {
  DH3D_REQUIRE(idx && dist && plan && B > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(m <= 1024 && B <= 65535);
  const int nblk = dh3d_cdiv(n, kIHP);
  hipLaunchKernelGGL(walk_plan_kernel, dim3(nblk, B), dim3(kIHT), 0, (hipStream_t)stream, idx, dist,
                     reinterpret_cast<const float4 *>(order), n, m, nblk, static_cast<int *>(plan));
  return dh3d_launch_status();
}

// dh3d_global_walk_fwd with the slot tables of dh3d_walk_plan (plan may be NULL: built inside the walk as before).
DH3D_API int dh3d_global_walk_planned_fwd(const float *H, int Hd, const float *coarse, const float *cw, const int32_t *idx,
                                          const float *dist, const float *order, const void *plan, int B, int n, int m,
                                          const dh3d_epilogue *ep, const float *w_fc, float b_fc, const float *cl_scale,
                                          const float *cl_shift, float *att, float *accum, int zero_accum, void *stream) {
  return global_tail_launch(H, Hd, coarse, cw, idx, dist, order, B, n, m, ep, w_fc, b_fc, cl_scale, cl_shift, att, accum,
                            zero_accum != 0, (hipStream_t)stream, false, static_cast<const int *>(plan));
}

DH3D_API int dh3d_interp_head_fwd(const float *H, int Hd, const int32_t *idx, const float *dist, int B, int n, int m,
                                  const dh3d_epilogue *ep, const float *w_fc, float b_fc, float *att, void *stream) {
  DH3D_REQUIRE(H && idx && dist && w_fc && att && B > 0 && n > 0 && m > 0 && Hd > 0);
  DH3D_SUPPORTED(Hd % 256 == 0 && Hd <= 1024);
  const long long R = (long long)B * n;
  int grid = (int)((R + 3) / 4);
  if (grid > 256 * 16) grid = 256 * 16;
  grid = (grid + 7) & ~7;  // whole rounds of the 8 XCDs
  hipLaunchKernelGGL(interp_head_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, H, Hd / 256,
                     (long long)B * m, idx, dist, n, m, dh3d_ep(ep), w_fc, b_fc, R, att);
  return dh3d_launch_status();
}
