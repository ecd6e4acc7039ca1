// Per-point dense layers of the DH3D backbone for gfx950 (core/tf_utils.py:99-109 feature_conv1d_1,
// core/backbones.py:45-55 se_res_bottleneck, :132-173 detection / attention heads,
// core/model.py:177-181 l2-normalise + concat).  In the reference these are tensorpack Conv2D /
// BatchNorm / TF element-wise ops with a transpose between every block; here every layer is one
// kernel on [R, C] rows (R = B*N points): rows staged through LDS once, exact-f32 MFMA GEMM against a
// fragment-packed weight, and bias + BatchNorm + activation (+ residual) applied in the store.
#include "mfma_gemm.h"
#include "wave_ops.h"
#include "bf16x3.h"

namespace {

#ifdef DH3D_SE_PROBE  // dev instrumentation (tools/se_res_phases.py): cycle stamps of se_res_mfma_kernel's phases, 64 workgroups
__device__ long long g_seprobe[64 * 16];
#define SEPROBE(i) do { if (threadIdx.x == 0 && blockIdx.x % 16 == 0 && blockIdx.x / 16 < 64) g_seprobe[(blockIdx.x / 16) * 16 + (i)] = clock64(); } while (0)
#else
#define SEPROBE(i) do { } while (0)
#endif

// ------------------------------------------------------------------ weight packing
__global__ __launch_bounds__(256) void pack_weight_kernel(const float *__restrict__ W, int Kd, int Dout,
                                                         float *__restrict__ packed) {
  const int KB = Kd / 8;
  const long long total = (long long)Kd * Dout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int t = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    const long long blk = e >> 8;  // nb*KB + kb
    const int kb = (int)(blk % KB);
    const int nb = (int)(blk / KB);
    const int k = kb * 8 + 4 * (lane >> 5) + t;
    const int col = nb * 32 + (lane & 31);
    packed[e] = W[(size_t)k * Dout + col];
  }
}

// Wcat = [bias; theta_x; theta_y; theta_z] ([4*Din, Dout]) packed directly from theta/bias.
__global__ __launch_bounds__(256) void pack_flex_weight_kernel(const float *__restrict__ theta,
                                                              const float *__restrict__ bias, int Din,
                                                              int Dout, float *__restrict__ packed) {
  const int Kd = 4 * Din;
  const int KB = Kd / 8;
  const long long total = (long long)Kd * Dout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total;
       e += (long long)gridDim.x * 256) {
    const int t = (int)(e & 3);
    const int lane = (int)((e >> 2) & 63);
    const long long blk = e >> 8;
    const int kb = (int)(blk % KB);
    const int nb = (int)(blk / KB);
    const int k = kb * 8 + 4 * (lane >> 5) + t;
    const int col = nb * 32 + (lane & 31);
    const int comp = k / Din, i = k % Din;
    packed[e] = comp == 0 ? bias[(size_t)i * Dout + col]
                          : theta[((size_t)(comp - 1) * Din + i) * Dout + col];
  }
}

// ------------------------------------------------------------------ linear
constexpr int kTM = 64;

// Stage rows [grow0, grow0+64) of [x1 | x2] into LDS (ld = Kd+4), zero-filling past R.
__device__ __forceinline__ void stage_rows(const float *__restrict__ x1, int C1,
                                           const float *__restrict__ x2, int C2, long long grow0,
                                           long long R, float *s_A, int ld) {
  const int Kd = C1 + C2;
  const int kv = Kd / 4;
  for (int e = threadIdx.x; e < kTM * kv; e += 256) {
    const int p = e / kv;
    const int c4 = (e - p * kv) * 4;
    const long long g = grow0 + p;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g < R) {
      v = c4 < C1 ? *reinterpret_cast<const float4 *>(x1 + g * C1 + c4)
                  : *reinterpret_cast<const float4 *>(x2 + g * C2 + (c4 - C1));
    }
    *reinterpret_cast<float4 *>(s_A + (size_t)p * ld + c4) = v;
  }
}

template <int NT>
__global__ __launch_bounds__(256) void linear_pm_kernel(const float *__restrict__ x1, int C1,
                                                       const float *__restrict__ x2, int C2,
                                                       const float *__restrict__ wpacked, long long R,
                                                       int Dout, EpilogueArgs ep,
                                                       const float *__restrict__ residual,
                                                       float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_A[];  // max(Kd, Dout) + 4 floats per row
  const int Kd = C1 + C2;
  const int ld = Kd + 4;
  const long long grow0 = (long long)blockIdx.x * kTM;
  stage_rows(x1, C1, x2, C2, grow0, R, s_A, ld);
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int row0 = (wave & 1) * 32;
  const int cb0 = wave >> 1;
  f32x16 acc[NT];
  zero_acc<NT>(acc);
  EpilogueRegs er[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) er[j] = epilogue_prefetch(ep, (cb0 + 2 * j) * 32 + (threadIdx.x & 31));
  wave_gemm_f32<NT>(s_A, ld, row0, wpacked, Kd / 8, cb0, 2, acc);
  // wide epilogue: tile -> LDS (A is dead) -> float4 rows (+ residual)
  const int ldo = Dout + 4;
  __syncthreads();
  wave_tiles_to_lds<NT>(acc, er, ep.act, s_A, ldo, row0, cb0, 2);
  __syncthreads();
  block_store_rows(s_A, ldo, kTM, grow0, R, Dout, residual, out);
}

// ------------------------------------------------------------------ MLP head (wide hidden layer kept on chip)
// att[r] = sigmoid( sum_j relu(bn(h[r,:] @ W[:,j] + b[j])) * w_fc[j] + b_fc ); H = 64*PASSES*... columns.
__global__ __launch_bounds__(256) void mlp_head_pm_kernel(const float *__restrict__ h, int C,
                                                         const float *__restrict__ wpacked, int H,
                                                         EpilogueArgs ep, const float *__restrict__ w_fc,
                                                         float b_fc, long long R,
                                                         float *__restrict__ att) {
  extern __shared__ __attribute__((aligned(16))) float s_A[];
  __shared__ float s_part[4][kTM];  // [wave][row]
  const int ld = C + 4;
  const long long grow0 = (long long)blockIdx.x * kTM;
  stage_rows(h, C, h, 0, grow0, R, s_A, ld);  // (a literal nullptr here crashes hipcc 7.2's inliner)
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int NB = H / 32;
  float part[2][16];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) part[q][r] = 0.f;
  // wave w owns column blocks w, w+4, ... and BOTH row blocks: each weight fragment is fetched once per WG
  for (int cb = wave; cb < NB; cb += 4) {
    f32x16 acc[2][1];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][0][r] = 0.f;
    wave_gemm_f32_rows2<1>(s_A, ld, wpacked, C / 8, cb, 1, acc);
    const int col = cb * 32 + (lane & 31);
    float pb = 0.f, sc = 1.f, sh = 0.f;
    if (ep.pre_bias) pb = ep.pre_bias[col];
    if (ep.scale) sc = ep.scale[col];
    if (ep.shift) sh = ep.shift[col];
    const float wf = w_fc[col];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        part[q][r] = fmaf(dh3d_act((acc[q][0][r] + pb) * sc + sh, ep.act), wf, part[q][r]);
  }
  // reduce over the 32 columns held by the lanes of each half-wave
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) part[q][r] += __shfl_xor(part[q][r], off, 64);
    }
  if ((lane & 31) == 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_part[wave][q * 32 + mfma_row(r, lane)] = part[q][r];
  }
  __syncthreads();
  if (threadIdx.x < kTM) {
    const long long g = grow0 + threadIdx.x;
    if (g < R) {
      const float logit = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]) + b_fc;
      att[g] = 1.f / (1.f + expf(-logit));
    }
  }
}

// ------------------------------------------------------------------ squeeze-excite residual
template <int C>
__global__ __launch_bounds__(256) void se_res_pm_kernel(const float *__restrict__ x,
                                                       const float *__restrict__ pool,
                                                       const float *__restrict__ W1,
                                                       const float *__restrict__ b1,
                                                       const float *__restrict__ W2,
                                                       const float *__restrict__ b2, long long R,
                                                       float *__restrict__ out) {
  constexpr int Cq = C / 4;
  constexpr int LDP = C + 1;
  __shared__ float s_pool[kTM * LDP];
  __shared__ float s_h[kTM * (Cq + 1)];
  __shared__ float s_W1[C * Cq];
  __shared__ __attribute__((aligned(16))) float s_W2[Cq * C];
  const int tid = threadIdx.x;
  const long long grow0 = (long long)blockIdx.x * kTM;
  for (int e = tid; e < C * Cq; e += 256) { s_W1[e] = W1[e]; s_W2[e] = W2[e]; }
  for (int e = tid; e < kTM * C; e += 256) {
    const int p = e / C, c = e % C;
    const long long g = grow0 + p;
    s_pool[p * LDP + c] = g < R ? pool[g * C + c] : 0.f;
  }
  __syncthreads();
  // squeeze: thread (p = tid/4, j-range (tid%4)*Cq/4 ...)
  {
    const int p = tid >> 2;
    constexpr int JW = Cq / 4;
    const int j0 = (tid & 3) * JW;
    float acc[JW];
#pragma unroll
    for (int j = 0; j < JW; ++j) acc[j] = b1[j0 + j];
    for (int c = 0; c < C; ++c) {
      const float v = s_pool[p * LDP + c];
#pragma unroll
      for (int j = 0; j < JW; ++j) acc[j] = fmaf(v, s_W1[c * Cq + j0 + j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < JW; ++j) s_h[p * (Cq + 1) + j0 + j] = acc[j] > 0.f ? acc[j] : 0.f;
  }
  __syncthreads();
  // excite: thread (p = tid/4, channel range (tid%4)*C/4 ...), 4 channels at a time
  {
    const int p = tid >> 2;
    const long long g = grow0 + p;
    constexpr int CW = C / 4;
    const int c0 = (tid & 3) * CW;
    if (g < R) {
      for (int cc = 0; cc < CW; cc += 4) {
        const int c = c0 + cc;
        float4 gate = *reinterpret_cast<const float4 *>(b2 + c);
        for (int j = 0; j < Cq; ++j) {
          const float hv = s_h[p * (Cq + 1) + j];
          const float4 w = *reinterpret_cast<const float4 *>(&s_W2[j * C + c]);
          gate.x = fmaf(hv, w.x, gate.x); gate.y = fmaf(hv, w.y, gate.y);
          gate.z = fmaf(hv, w.z, gate.z); gate.w = fmaf(hv, w.w, gate.w);
        }
        const float4 xv = *reinterpret_cast<const float4 *>(x + g * C + c);
        float4 r;
        r.x = xv.x + xv.x * (1.f / (1.f + expf(-gate.x)));
        r.y = xv.y + xv.y * (1.f / (1.f + expf(-gate.y)));
        r.z = xv.z + xv.z * (1.f / (1.f + expf(-gate.z)));
        r.w = xv.w + xv.w * (1.f / (1.f + expf(-gate.w)));
        r.x = r.x > 0.f ? r.x : 0.f; r.y = r.y > 0.f ? r.y : 0.f;
        r.z = r.z > 0.f ? r.z : 0.f; r.w = r.w > 0.f ? r.w : 0.f;
        *reinterpret_cast<float4 *>(out + g * C + c) = r;
      }
    }
  }
}

// A 64-row f32 tile in LDS (leading dimension LDT, K = 64 columns) through one more 1x1 conv 64 -> 128 on the bf16 pipe at
// f32 accuracy (bf16x6, as linear_k64_x6_kernel) without leaving the workgroup: wave w owns row block w & 1 and the column
// blocks 2 (w >> 1), 2 (w >> 1) + 1; the A fragments are split in registers on their way out of LDS, the weight
// fragments (dh3d_pack_weight_x3, L2-resident) come one K-step ahead, the result leaves from the accumulators (bias /
// BatchNorm / ReLU in packed f32).  `tail`: {weight, epilogue, output [R, 128]}.
struct TileTail {
  const uint4 *wp3;
  EpilogueArgs ep;  // act: DH3D_ACT_NONE or DH3D_ACT_RELU
  float *out;
};
template <int LDT>
__device__ __forceinline__ void tile_tail_k64_x6(const float *s_tile, const TileTail &t, long long grow0, long long R) {
  constexpr int KB = 4, DOUT = 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, lr = lane & 31;
  const int rb = wave & 1, cb0 = (wave >> 1) * 2;
  const float *ap = s_tile + (size_t)(rb * 32 + lr) * LDT + 8 * half;
  const uint4 *wl = t.wp3 + lane;
  uint4 bq[2][2][3];
  auto load_b = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int p = 0; p < 3; ++p) bq[ks & 1][c][p] = wl[((size_t)((cb0 + c) * KB + ks) * 3 + p) * 64];
  };
  load_b(0);
  float pb[2], sc[2], sh[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int col = (cb0 + c) * 32 + lr;
    pb[c] = t.ep.pre_bias ? t.ep.pre_bias[col] : 0.f;
    sc[c] = t.ep.scale ? t.ep.scale[col] : 1.f;
    sh[c] = t.ep.shift ? t.ep.shift[col] : 0.f;
  }
  f32x16 acc[2];
  zero_acc<2>(acc);
#pragma unroll
  for (int ks = 0; ks < KB; ++ks) {
    if (ks + 1 < KB) load_b(ks + 1);
    const float4 a0 = *reinterpret_cast<const float4 *>(ap + ks * 16), a1 = *reinterpret_cast<const float4 *>(ap + ks * 16 + 4);
    __builtin_amdgcn_sched_barrier(0);
    uint2 c1[2], c2[2], c3[2];
    split3x4(a0, c1[0], c2[0], c3[0]);
    split3x4(a1, c1[1], c2[1], c3[1]);
    bf16x8 a[3];
    a[0] = __builtin_bit_cast(bf16x8, make_uint4(c1[0].x, c1[0].y, c1[1].x, c1[1].y));
    a[1] = __builtin_bit_cast(bf16x8, make_uint4(c2[0].x, c2[0].y, c2[1].x, c2[1].y));
    a[2] = __builtin_bit_cast(bf16x8, make_uint4(c3[0].x, c3[0].y, c3[1].x, c3[1].y));
#define DH3D_TT_PRODUCT(PA, PB)                                                                           \
  _Pragma("unroll") for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(        \
      a[PA], __builtin_bit_cast(bf16x8, bq[ks & 1][c][PB]), acc[c], 0, 0, 0);
    DH3D_TT_PRODUCT(2, 0) DH3D_TT_PRODUCT(0, 2) DH3D_TT_PRODUCT(1, 1)
    DH3D_TT_PRODUCT(1, 0) DH3D_TT_PRODUCT(0, 1) DH3D_TT_PRODUCT(0, 0)
#undef DH3D_TT_PRODUCT
  }
  typedef float v2f __attribute__((ext_vector_type(2)));
  const float lo = t.ep.act == DH3D_ACT_RELU ? 0.f : -__builtin_inff();  // ReLU as a clamp, or no activation
  const long long row0 = grow0 + rb * 32 + 4 * half;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    float *op = t.out + row0 * DOUT + (cb0 + c) * 32 + lr;
    const v2f pb2 = {pb[c], pb[c]}, sc2 = {sc[c], sc[c]}, sh2 = {sh[c], sh[c]}, lo2 = {lo, lo};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      v2f y = {acc[c][r], acc[c][r + 1]};
      y = __builtin_elementwise_max(__builtin_elementwise_fma(y + pb2, sc2, sh2), lo2);
      const int t0 = (r & 3) + 8 * (r >> 2);
      if (row0 + t0 < R) op[t0 * DOUT] = y[0];
      if (row0 + t0 + 1 < R) op[(t0 + 1) * DOUT] = y[1];
    }
  }
}

// The same block on the matrix pipe.  The VALU version above issues ~1700 ds_read_b32 per lane (one per weight) and
// needs 25-30 us per 64-row tile; here the two small GEMMs run on v_mfma_f32_32x32x2_f32 against weights packed with
// dh3d_pack_weight (W1 padded to 32 columns, W2 to 32 rows: the padding contributes exact zeros):
//   squeeze  [64, C] x [C, 32]   waves 0/1 = the two row blocks
//   excite   [64, 32] x [32, C]  all four waves, C/32 column blocks
// then out = relu(x + x * sigmoid(.)) row-wise with 16-byte accesses.
// POOL: the pooled rows are not read but formed here, flex_pool's neighbour maximum (flex_pool_kernel_gpu.cu.cc:30-63)
// fused into the staging -- the [R, C] pooled map is never written or read back and one launch (+ its dependency gap on
// the critical tail of the local step) disappears.  Same values as flex_pool_pm_kernel: a maximum is order-independent.
// CONV: the block's output tile goes through one more 1x1 conv (C -> 64, its own bias / BatchNorm / activation)
// before it leaves the chip: `before_stage2_conv1d` behind stage 1 (core/backbones.py:117) -- both results are stored,
// the tile is not read back and a launch on the global path's critical chain disappears.
// TMR: rows per workgroup -- 64, or 32 (C == 128) for launches with fewer 64-row tiles than CUs (the sampled levels:
// 8192 rows = 128 tiles on 256 CUs, every phase of the tile behind a barrier).
// TAILS (C = 64, CONV): two more 64 -> 128 convs ride in the launch -- tail A on the block's output, tail B on the conv's
// output (the local step's shortcut conv and the commuted concat conv's lower block: two launches over the tiles this
// kernel already holds); `out` may then be NULL (nothing else reads the block's output).
template <int C, bool POOL, bool CONV, int TMR = kTM, bool TAILS = false>
__global__ __launch_bounds__(256) void se_res_mfma_kernel(const float *__restrict__ x, const float *__restrict__ pool,
                                                         const int32_t *__restrict__ nbr, int N, int K,
                                                         const float *__restrict__ w1p, const float *__restrict__ b1p,
                                                         const float *__restrict__ w2p, const float *__restrict__ b2,
                                                         long long R, float *__restrict__ out,
                                                         const float *__restrict__ wconv, EpilogueArgs cep,
                                                         float *__restrict__ out2, TileTail ta = TileTail{},
                                                         TileTail tb = TileTail{}) {
  static_assert(!TAILS || (C == 64 && CONV && TMR == 64), "tails: the 64-wide block with its conv, 64-row tiles");
  constexpr int LDP = C + 4;   // pooled rows, later the gate tile
  constexpr int LDH = 32 + 4;  // hidden rows
  static_assert(TMR == 64 || (TMR == 32 && C == 128 && POOL), "32-row tiles: four column blocks for four waves; pooled staging only");
  constexpr int RB = TMR / 32;  // 32-row blocks
  __shared__ __attribute__((aligned(16))) float s_p[TMR * LDP];
  __shared__ __attribute__((aligned(16))) float s_h[TMR * LDH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // POOL gathers neighbour rows of the tile's own cloud: every XCD gets a CONTIGUOUS range of tiles, so a cloud's map is
  // fetched into one L2 instead of all eight (PMC, 8 x 8192 x 64: FETCH_SIZE 57.9 MB raw with the round-robin order)
  const long long grow0 = (long long)(POOL ? dh3d_xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x) * TMR;
  SEPROBE(0);
  if (POOL) {
    // the tile's neighbour ids once into LDS (as row offsets; every id is used by C/4 lanes), then all K row reads of a
    // lane in flight together
    int *s_nb = reinterpret_cast<int *>(s_h);  // [TMR][K <= 9] ints: the hidden tile is not live yet
    const bool k8 = K == 8;
    if (k8) {
      for (int e = tid; e < TMR * 8; e += 256) {
        const long long g = grow0 + (e >> 3);
        s_nb[e] = g < R ? (int)((g / N) * N) + nbr[g * 8 + (e & 7)] : 0;
      }
      __syncthreads();
    }
    SEPROBE(1);
    constexpr int CVP = C / 4;
    if (k8) {
      // two points per lane and pass: their 16 row reads are requested together and none sits under a branch (rows past
      // the end read row 0 through s_nb and are zeroed afterwards) -- the loop was kTM * CVP / 256 dependent round trips
      constexpr int IT = TMR * CVP / 256;
      static_assert(IT % 2 == 0, "pairs of passes");
#pragma unroll
      for (int it = 0; it < IT; it += 2) {
        float4 v[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = tid + (it + u) * 256, p = e / CVP, c4 = (e - p * CVP) * 4;
#pragma unroll
          for (int k = 0; k < 8; ++k) v[u][k] = *reinterpret_cast<const float4 *>(x + (size_t)s_nb[p * 8 + k] * C + c4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = tid + (it + u) * 256, p = e / CVP, c4 = (e - p * CVP) * 4;
          float4 best = make_float4(-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);  // -FLT_MAX
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            best.x = fmaxf(best.x, v[u][k].x); best.y = fmaxf(best.y, v[u][k].y);
            best.z = fmaxf(best.z, v[u][k].z); best.w = fmaxf(best.w, v[u][k].w);
          }
          if (grow0 + p >= R) best = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4 *>(s_p + (size_t)p * LDP + c4) = best;
        }
      }
    } else {
      for (int e = tid; e < TMR * CVP; e += 256) {
        const int p = e / CVP, c4 = (e - p * CVP) * 4;
        const long long g = grow0 + p;
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < R) {
          best = make_float4(-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);  // -FLT_MAX
          const long long cloud0 = (g / N) * N;
          const int32_t *nb = nbr + g * K;
#pragma unroll 4
          for (int k = 0; k < K; ++k) {
            const float4 v = *reinterpret_cast<const float4 *>(x + (cloud0 + nb[k]) * C + c4);
            best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
          }
        }
        *reinterpret_cast<float4 *>(s_p + (size_t)p * LDP + c4) = best;
      }
    }
  } else {
    stage_rows(pool, C, pool, 0, grow0, R, s_p, LDP);
  }
  SEPROBE(2);
  __syncthreads();
  SEPROBE(3);
  if (wave < RB) {  // squeeze: relu(pool @ W1 + b1)
    f32x16 acc[1];
    zero_acc<1>(acc);
    wave_gemm_f32<1>(s_p, LDP, wave * 32, w1p, C / 8, 0, 1, acc);
    const float bb = b1p[lane & 31];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[0][r] + bb;
      s_h[(size_t)(wave * 32 + mfma_row(r, lane)) * LDH + (lane & 31)] = v > 0.f ? v : 0.f;
    }
  }
  SEPROBE(4);
  __syncthreads();
  SEPROBE(5);
  {  // excite: sigmoid(h @ W2 + b2) -> gate tile over the (dead) pooled rows
    // column blocks per wave: (row block = wave & 1, column blocks (wave >> 1) + 2 j), or with one row block: wave + 4 j
    constexpr int NT = C / 32 * RB / 4, CBS = 4 / RB;
    const int rb0 = RB == 2 ? (wave & 1) * 32 : 0, cbf = RB == 2 ? wave >> 1 : wave;
    f32x16 acc[NT];
    zero_acc<NT>(acc);
    wave_gemm_f32<NT>(s_h, LDH, rb0, w2p, 32 / 8, cbf, CBS, acc);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = (cbf + CBS * j) * 32 + (lane & 31);
      const float bb = b2[col];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        s_p[(size_t)(rb0 + mfma_row(r, lane)) * LDP + col] = __builtin_amdgcn_rcpf(1.f + __expf(-(acc[j][r] + bb)));  // (sigmoid to ~1e-7:
                                                                     // v_exp + v_rcp instead of the ~20 instructions of expf and an IEEE division)
    }
  }
  SEPROBE(6);
  __syncthreads();
  SEPROBE(7);
  constexpr int CV = C / 4;
  {
    // the tile's own rows: every pass's read requested before the first is used (rows past the end re-read the last row)
    constexpr int ITG = TMR * CV / 256;
    float4 xr[ITG];
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      const int e = tid + it * 256, p = e / CV, c4 = (e - p * CV) * 4;
      const long long g = grow0 + p < R ? grow0 + p : R - 1;
      xr[it] = *reinterpret_cast<const float4 *>(x + g * C + c4);
    }
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      const int e = tid + it * 256, p = e / CV, c4 = (e - p * CV) * 4;
      const long long g = grow0 + p;
      if (g < R) {
        const float4 xv = xr[it];
        const float4 gt = *reinterpret_cast<const float4 *>(s_p + (size_t)p * LDP + c4);
        float4 r;
        r.x = xv.x + xv.x * gt.x; r.y = xv.y + xv.y * gt.y; r.z = xv.z + xv.z * gt.z; r.w = xv.w + xv.w * gt.w;
        r.x = r.x > 0.f ? r.x : 0.f; r.y = r.y > 0.f ? r.y : 0.f;
        r.z = r.z > 0.f ? r.z : 0.f; r.w = r.w > 0.f ? r.w : 0.f;
        if (!TAILS || out) *reinterpret_cast<float4 *>(out + g * C + c4) = r;
        if (CONV) *reinterpret_cast<float4 *>(s_p + (size_t)p * LDP + c4) = r;  // over its own gate entry
      } else if (CONV) {
        *reinterpret_cast<float4 *>(s_p + (size_t)p * LDP + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  SEPROBE(8);
  if (CONV) {  // out2 = act(bn(tile @ Wconv + b)), C columns: C / 64 accumulators of 32 x 32 per wave
    constexpr int NTC = C / 32 * RB / 4, CBC = 4 / RB;
    __syncthreads();
    SEPROBE(9);
    const int row0 = RB == 2 ? (wave & 1) * 32 : 0, cb0 = RB == 2 ? wave >> 1 : wave;
    f32x16 acc[NTC];
    zero_acc<NTC>(acc);
    EpilogueRegs er[NTC];
#pragma unroll
    for (int j = 0; j < NTC; ++j) er[j] = epilogue_prefetch(cep, (cb0 + CBC * j) * 32 + (lane & 31));
    if constexpr (TAILS) tile_tail_k64_x6<LDP>(s_p, ta, grow0, R);  // on the block's output tile
    wave_gemm_f32<NTC>(s_p, LDP, row0, wconv, C / 8, cb0, CBC, acc);
    SEPROBE(10);
    __syncthreads();
    wave_tiles_to_lds<NTC>(acc, er, cep.act, s_p, LDP, row0, cb0, CBC);
    __syncthreads();
    SEPROBE(11);
    block_store_rows(s_p, LDP, TMR, grow0, R, C, nullptr, out2);
    SEPROBE(12);
    if constexpr (TAILS) tile_tail_k64_x6<LDP>(s_p, tb, grow0, R);  // on the conv's output tile
  }
}

// ------------------------------------------------------------------ l2 normalise (+ prefix concat)
__global__ __launch_bounds__(256) void l2norm_concat_kernel(const float *__restrict__ x, long long R, int C,
                                                           float eps, const float *__restrict__ prefix,
                                                           int P, float *__restrict__ out) {
  const int sub = threadIdx.x & 31;                        // 32 lanes per row
  const long long row = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (row >= R) return;                                    // whole 32-lane group exits together
  float ss = 0.f;
  for (int c = sub; c < C; c += 32) { const float v = x[row * C + c]; ss = fmaf(v, v, ss); }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
  const float inv = rsqrtf(fmaxf(ss, eps));  // tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), eps))
  float *o = out + row * (P + C);
  for (int c = sub; c < P; c += 32) o[c] = prefix[row * P + c];
  for (int c = sub; c < C; c += 32) o[P + c] = x[row * C + c] * inv;
}

// ------------------------------------------------------------------ commuted concat conv: gather + epilogue
// The concat conv behind an up-sampling, y = act(BN([interp(coarse) | x2] W + b)) (core/backbones.py:89-100), is linear
// in its two inputs and the three interpolation weights act on rows:  interp(coarse) W_top = interp(coarse W_top).  So
// the GEMM against W_top runs on the COARSE rows (N/8 of them) and the one against W_bot on x2 as soon as x2 exists
// (beside the farthest-point sampling, off the critical chain); what is left behind the sampled level is this kernel:
//   v = w0*Cw[i0] + w1*Cw[i1] + w2*Cw[i2] + P[n] + b  ->  BN, activation  ->  + residual  ->  out / [prefix | l2norm(v)]
// 32 lanes x float4 per row (C == 128), a wave = two rows; the l2-normalisation's row sum is a 32-lane DPP-free
// shuffle reduction.  HBM: P + residual + out rows (~1.5 KB per point), the coarse rows come from L2.
// A wave takes RPW consecutive rows of one cloud: lane r < RPW loads row r's three indices and distances and turns the
// distances into weights ONCE (the six IEEE divisions cost ~14 instructions each and were issued per two rows);
// the row loop fetches them with ds_bpermute.  No load sits under a branch (ragged tails clamp the row, the store is
// predicated); the cloud comes from the block index (no 64-bit division).
// PART / RES / L2CAT / ACT (-1: read ep.act) are compile-time so that the row loop is one basic block.
template <int RPW, bool PART, bool RES, bool L2CAT, int ACT>
__global__ __launch_bounds__(256) void interp_combine_kernel(const float *__restrict__ cw, const int32_t *__restrict__ idx,
                                                            const float *__restrict__ dist,
                                                            const float *__restrict__ part, int n, int m, int B, int nblk,
                                                            EpilogueArgs ep, const float *__restrict__ residual,
                                                            const float *__restrict__ prefix, float l2_eps,
                                                            float *__restrict__ out) {
  constexpr int C = 128;
  __shared__ float s_cat[L2CAT ? 4 * RPW * (C + 3) : 1];   // L2CAT: the four waves' output rows, staged for contiguous stores
  const int lane = threadIdx.x & 63, half = lane >> 5, sub = lane & 31, c4 = sub * 4;
  // XCD-aware (round 6): as grid (blocks, B) a cloud's blocks went round-robin over the eight XCDs and each L2 fetched every
  // cloud's coarse rows; a linear grid with XCD x serving clouds x, x + 8, ... keeps a cloud's 512 KB in ONE L2
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int blk = seq % nblk;
  const long long cloud = xcd + 8 * (seq / nblk);
  if (cloud >= B) return;
  const int row0 = (blk * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= n) return;
  const Ep4 e = ep4_prefetch(ep, c4);
  const int act = ACT >= 0 ? ACT : ep.act;
  int o1, o2, o3;
  float w1, w2, w3;
  {
    const int r = min(row0 + (lane & (RPW - 1)), n - 1);
    const long long g = (cloud * n + r) * 3;
    o1 = idx[g] * C; o2 = idx[g + 1] * C; o3 = idx[g + 2] * C;
    // the inverse-distance weights of core/backbones.py:92-95, same arithmetic as three_interp_fwd_kernel<true>
    const float r1 = 1.0f / fmaxf(dist[g], 1e-10f), r2 = 1.0f / fmaxf(dist[g + 1], 1e-10f),
                r3 = 1.0f / fmaxf(dist[g + 2], 1e-10f);
    const float norm = (r1 + r2) + r3;
    w1 = r1 / norm; w2 = r2 / norm; w3 = r3 / norm;
  }
  const float *cwb = cw + cloud * m * C + c4;
  const long long base = cloud * n;
#pragma unroll 4
  for (int t = 0; t < RPW / 2; ++t) {
    const int src = 2 * t + half;
    const bool live = row0 + src < n;
    const long long row = base + min(row0 + src, n - 1);
    const int a1 = __shfl(o1, src, 64), a2 = __shfl(o2, src, 64), a3 = __shfl(o3, src, 64);
    const float u1 = __shfl(w1, src, 64), u2 = __shfl(w2, src, 64), u3 = __shfl(w3, src, 64);
    const float4 a = *reinterpret_cast<const float4 *>(cwb + a1);
    const float4 bq = *reinterpret_cast<const float4 *>(cwb + a2);
    const float4 cq = *reinterpret_cast<const float4 *>(cwb + a3);
    float4 v;
    v.x = (a.x * u1 + bq.x * u2) + cq.x * u3; v.y = (a.y * u1 + bq.y * u2) + cq.y * u3;
    v.z = (a.z * u1 + bq.z * u2) + cq.z * u3; v.w = (a.w * u1 + bq.w * u2) + cq.w * u3;
    if (PART) {
      const float4 p = *reinterpret_cast<const float4 *>(part + row * C + c4);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    v.x = dh3d_act((v.x + e.pb.x) * e.sc.x + e.sh.x, act); v.y = dh3d_act((v.y + e.pb.y) * e.sc.y + e.sh.y, act);
    v.z = dh3d_act((v.z + e.pb.z) * e.sc.z + e.sh.z, act); v.w = dh3d_act((v.w + e.pb.w) * e.sc.w + e.sh.w, act);
    if (RES) {
      const float4 q = *reinterpret_cast<const float4 *>(residual + row * C + c4);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    if (L2CAT) {  // [xyz | l2_normalize(v)] (core/model.py:177-181)
      float ss = row16_sum_f32((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
      ss += __shfl_xor(ss, 16, 64);
      const float inv = rsqrtf(fmaxf(ss, l2_eps));
      const float pf = prefix[row * 3 + min(sub, 2)];
      // 131-float rows: a lane's four channels start at byte 524 row + 12 + 16 sub -- stored from here they are four
      // 4-byte stores per lane with a 16-byte stride (a quarter of every line per instruction).  The wave's RPW rows are
      // ONE contiguous run of 524 RPW bytes: staged in LDS (the wave's own region) and written as consecutive dwords below.
      float *so = s_cat + (size_t)(threadIdx.x >> 6) * (RPW * (C + 3)) + (size_t)src * (C + 3);
      if (sub < 3) so[sub] = pf;
      so[3 + c4] = v.x * inv; so[4 + c4] = v.y * inv; so[5 + c4] = v.z * inv; so[6 + c4] = v.w * inv;
    } else if (live) {
      *reinterpret_cast<float4 *>(out + row * C + c4) = v;
    }
  }
  if (L2CAT) {
    const int rows = min(RPW, n - row0);                  // wave-uniform
    const float *sw = s_cat + (size_t)(threadIdx.x >> 6) * (RPW * (C + 3));
    float *ow = out + (base + row0) * (C + 3);
    const int total = rows * (C + 3);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // this wave's LDS writes above, read back by other lanes
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int e = lane; e < total; e += 64) ow[e] = sw[e];
  }
}

}  // namespace

DH3D_API int dh3d_interp_combine_fwd(const float *coarse_w, const int32_t *idx, const float *dist, const float *partial,
                                     int B, int N, int M, int C, const dh3d_epilogue *ep, const float *residual,
                                     const float *prefix, float l2_eps, float *out, void *stream) {
  DH3D_REQUIRE(coarse_w && idx && dist && out && B > 0 && N > 0 && M > 0);
  DH3D_SUPPORTED(C == 128);
  DH3D_SUPPORTED((long long)M * C < (1ll << 31) && B <= 65535);
#ifndef DH3D_IC_ROWS
#define DH3D_IC_ROWS 16
#endif
  constexpr int kRows = DH3D_IC_ROWS;  // per wave
  const EpilogueArgs e = dh3d_ep(ep);
  const int nblk = dh3d_cdiv(N, 4 * kRows);
  const dim3 grid(8 * dh3d_cdiv(B, 8) * nblk);
  hipStream_t s = (hipStream_t)stream;
#define DH3D_IC_LAUNCH(PART, RES, L2, ACT)                                                                          \
  hipLaunchKernelGGL((interp_combine_kernel<kRows, PART, RES, L2, ACT>), grid, dim3(256), 0, s, coarse_w, idx, dist, \
                     partial, N, M, B, nblk, e, residual, prefix, l2_eps, out)
#define DH3D_IC_ACT(PART, RES, L2)                                     \
  {                                                                    \
    if (e.act == DH3D_ACT_RELU) DH3D_IC_LAUNCH(PART, RES, L2, DH3D_ACT_RELU); \
    else if (e.act == DH3D_ACT_NONE) DH3D_IC_LAUNCH(PART, RES, L2, DH3D_ACT_NONE); \
    else DH3D_IC_LAUNCH(PART, RES, L2, -1);                            \
  }
#define DH3D_IC_RES(PART, L2)                   \
  {                                             \
    if (residual) DH3D_IC_ACT(PART, true, L2)   \
    else DH3D_IC_ACT(PART, false, L2)           \
  }
  if (partial) {
    if (prefix) DH3D_IC_RES(true, true) else DH3D_IC_RES(true, false)
  } else {
    if (prefix) DH3D_IC_RES(false, true) else DH3D_IC_RES(false, false)
  }
#undef DH3D_IC_RES
#undef DH3D_IC_ACT
#undef DH3D_IC_LAUNCH
  return dh3d_launch_status();
}

DH3D_API int dh3d_pack_weight(const float *W, int Kd, int Dout, float *packed, void *stream) {
  DH3D_REQUIRE(W && packed && Kd > 0 && Dout > 0);
  DH3D_SUPPORTED(Kd % 8 == 0 && Dout % 32 == 0);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(dh3d_cdiv((long long)Kd * Dout, 256)), dim3(256), 0,
                     (hipStream_t)stream, W, Kd, Dout, packed);
  return dh3d_launch_status();
}

DH3D_API int dh3d_pack_flex_weight(const float *theta, const float *bias, int Din, int Dout,
                                   float *packed, void *stream) {
  DH3D_REQUIRE(theta && bias && packed && Din > 0 && Dout > 0);
  DH3D_SUPPORTED(Din % 2 == 0 && Dout % 32 == 0);
  hipLaunchKernelGGL(pack_flex_weight_kernel, dim3(dh3d_cdiv((long long)4 * Din * Dout, 256)), dim3(256),
                     0, (hipStream_t)stream, theta, bias, Din, Dout, packed);
  return dh3d_launch_status();
}

DH3D_API int dh3d_linear_pm_fwd(const float *x1, int C1, const float *x2, int C2, const float *wpacked,
                                int R, int Dout, const dh3d_epilogue *ep, const float *residual,
                                float *out, void *stream) {
  DH3D_REQUIRE(x1 && wpacked && out && R > 0 && C1 > 0 && C2 >= 0 && Dout > 0 && (C2 == 0 || x2));
  const int Kd = C1 + C2;
  DH3D_SUPPORTED(C1 % 4 == 0 && C2 % 4 == 0 && Kd % 8 == 0 && Dout % 64 == 0 && Dout <= 256 && Kd <= 512);
  const size_t lds = sizeof(float) * kTM * ((Kd > Dout ? Kd : Dout) + 4);
  const EpilogueArgs e = dh3d_ep(ep);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(dh3d_cdiv(R, kTM)), block(256);
#define DH3D_LIN_CASE(NTV)                                                                            \
  {                                                                                                   \
    auto kern = linear_pm_kernel<NTV>;                                                                \
    DH3D_ALLOW_BIG_LDS(kern);                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds, s, x1, C1, x2, C2, wpacked, (long long)R, Dout, e,     \
                       residual, out);                                                                \
  }
  switch (Dout / 64) {
    case 1: DH3D_LIN_CASE(1) break;
    case 2: DH3D_LIN_CASE(2) break;
    case 3: DH3D_LIN_CASE(3) break;
    default: DH3D_LIN_CASE(4) break;
  }
#undef DH3D_LIN_CASE
  return dh3d_launch_status();
}

DH3D_API int dh3d_mlp_head_pm_fwd(const float *h, int R, int C, const float *wpacked, int H,
                                  const dh3d_epilogue *ep, const float *w_fc, float b_fc, float *att,
                                  void *stream) {
  DH3D_REQUIRE(h && wpacked && w_fc && att && R > 0 && C > 0 && H > 0);
  DH3D_SUPPORTED(C % 8 == 0 && C <= 512 && H % 128 == 0);
  const size_t lds = sizeof(float) * kTM * (C + 4);
  auto kern = mlp_head_pm_kernel;
  DH3D_ALLOW_BIG_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(dh3d_cdiv(R, kTM)), dim3(256), lds, (hipStream_t)stream, h, C, wpacked, H,
                     dh3d_ep(ep), w_fc, b_fc, (long long)R, att);
  return dh3d_launch_status();
}

DH3D_API int dh3d_se_res_pm_fwd(const float *x, const float *pool, const float *W1, const float *b1,
                                const float *W2, const float *b2, int R, int C, float *out,
                                void *stream) {
  DH3D_REQUIRE(x && pool && W1 && b1 && W2 && b2 && out && R > 0);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(dh3d_cdiv(R, kTM)), block(256);
  if (C == 64)
    hipLaunchKernelGGL(se_res_pm_kernel<64>, grid, block, 0, s, x, pool, W1, b1, W2, b2, (long long)R, out);
  else if (C == 128)
    hipLaunchKernelGGL(se_res_pm_kernel<128>, grid, block, 0, s, x, pool, W1, b1, W2, b2, (long long)R, out);
  else
    return DH3D_ERR_UNSUPPORTED;
  return dh3d_launch_status();
}

DH3D_API int dh3d_se_res_pm_packed_fwd(const float *x, const float *pool, const float *w1packed, const float *b1pad,
                                       const float *w2packed, const float *b2, int R, int C, float *out,
                                       void *stream) {
  DH3D_REQUIRE(x && pool && w1packed && b1pad && w2packed && b2 && out && R > 0);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(dh3d_cdiv(R, kTM)), block(256);
  if (C == 64)
    hipLaunchKernelGGL((se_res_mfma_kernel<64, false, false>), grid, block, 0, s, x, pool, nullptr, 1, 0, w1packed, b1pad,
                       w2packed, b2, (long long)R, out, nullptr, EpilogueArgs{}, nullptr);
  else if (C == 128)
    hipLaunchKernelGGL((se_res_mfma_kernel<128, false, false>), grid, block, 0, s, x, pool, nullptr, 1, 0, w1packed, b1pad,
                       w2packed, b2, (long long)R, out, nullptr, EpilogueArgs{}, nullptr);
  else
    return DH3D_ERR_UNSUPPORTED;
  return dh3d_launch_status();
}

// se_res_bottleneck on the flex_pool of x (core/backbones.py:76-79: SE on max_pool) in one launch: pooled rows formed
// while staging.  x [B*N, C], nbr [B, N, K].
DH3D_API int dh3d_se_res_pool_pm_packed_fwd(const float *x, const int32_t *nbr, int B, int N, int K,
                                            const float *w1packed, const float *b1pad, const float *w2packed,
                                            const float *b2, int C, float *out, void *stream) {
  DH3D_REQUIRE(x && nbr && w1packed && b1pad && w2packed && b2 && out && B > 0 && N > 0 && K > 0);
  hipStream_t s = (hipStream_t)stream;
  const long long R = (long long)B * N;
  const dim3 grid(dh3d_cdiv(R, kTM)), block(256);
  if (C == 64)
    hipLaunchKernelGGL((se_res_mfma_kernel<64, true, false>), grid, block, 0, s, x, x, nbr, N, K, w1packed, b1pad, w2packed, b2,
                       R, out, nullptr, EpilogueArgs{}, nullptr);
  else if (C == 128 && R <= kTM * 256)  // fewer 64-row tiles than CUs (the sampled levels): 32-row tiles
    hipLaunchKernelGGL((se_res_mfma_kernel<128, true, false, 32>), dim3(dh3d_cdiv(R, 32)), block, 0, s, x, x, nbr, N, K,
                       w1packed, b1pad, w2packed, b2, R, out, nullptr, EpilogueArgs{}, nullptr);
  else if (C == 128)
    hipLaunchKernelGGL((se_res_mfma_kernel<128, true, false>), grid, block, 0, s, x, x, nbr, N, K, w1packed, b1pad, w2packed, b2,
                       R, out, nullptr, EpilogueArgs{}, nullptr);
  else
    return DH3D_ERR_UNSUPPORTED;
  return dh3d_launch_status();
}

// dh3d_se_res_pool_pm_packed_fwd followed by a 1x1 conv 64 -> 64 on the block's output (wconv = dh3d_pack_weight of
// [64, 64], `ep` its bias / BatchNorm / activation) in the same launch: out [B*N, 64] and out2 [B*N, 64] both stored.
DH3D_API int dh3d_se_res_pool_conv_pm_fwd(const float *x, const int32_t *nbr, int B, int N, int K, const float *w1packed,
                                          const float *b1pad, const float *w2packed, const float *b2, int C, float *out,
                                          const float *wconv_packed, const dh3d_epilogue *ep, int Dout, float *out2,
                                          void *stream) {
  DH3D_REQUIRE(x && nbr && w1packed && b1pad && w2packed && b2 && out && wconv_packed && out2 && B > 0 && N > 0 && K > 0);
  DH3D_SUPPORTED(((C == 64 && Dout == 64) || (C == 128 && Dout == 128)) && (!ep || ep->act != DH3D_ACT_SIGMOID));
  const long long R = (long long)B * N;
  if (C == 64)
    hipLaunchKernelGGL((se_res_mfma_kernel<64, true, true>), dim3(dh3d_cdiv(R, kTM)), dim3(256), 0, (hipStream_t)stream, x,
                       x, nbr, N, K, w1packed, b1pad, w2packed, b2, R, out, wconv_packed, dh3d_ep(ep), out2);
  else if (R <= kTM * 256)  // fewer 64-row tiles than CUs (the sampled levels): 32-row tiles
    hipLaunchKernelGGL((se_res_mfma_kernel<128, true, true, 32>), dim3(dh3d_cdiv(R, 32)), dim3(256), 0, (hipStream_t)stream,
                       x, x, nbr, N, K, w1packed, b1pad, w2packed, b2, R, out, wconv_packed, dh3d_ep(ep), out2);
  else
    hipLaunchKernelGGL((se_res_mfma_kernel<128, true, true>), dim3(dh3d_cdiv(R, kTM)), dim3(256), 0, (hipStream_t)stream, x,
                       x, nbr, N, K, w1packed, b1pad, w2packed, b2, R, out, wconv_packed, dh3d_ep(ep), out2);
  return dh3d_launch_status();
}

// dh3d_se_res_pool_conv_pm_fwd (C = 64) with two more 1x1 convs 64 -> 128 in the launch, on the bf16 pipe at f32 accuracy:
// out_a = act_a(bn_a(y @ Wa)) on the block's output y, out_b = act_b(bn_b(z @ Wb)) on z = the 64 -> 64 conv's output
// (Wa / Wb = dh3d_pack_weight_x3 of [64, 128]; act NONE or RELU).  `out` (y) may be NULL: not stored.
DH3D_API int dh3d_se_res_pool_conv_tails_pm_fwd(const float *x, const int32_t *nbr, int B, int N, int K, const float *w1packed,
                                                const float *b1pad, const float *w2packed, const float *b2, float *out,
                                                const float *wconv_packed, const dh3d_epilogue *ep, float *out2,
                                                const void *wa_x3, const dh3d_epilogue *ep_a, float *out_a,
                                                const void *wb_x3, const dh3d_epilogue *ep_b, float *out_b, void *stream) {
  DH3D_REQUIRE(x && nbr && w1packed && b1pad && w2packed && b2 && wconv_packed && out2 && wa_x3 && out_a && wb_x3 && out_b &&
               B > 0 && N > 0 && K > 0);
  const EpilogueArgs ea = dh3d_ep(ep_a), eb = dh3d_ep(ep_b);
  DH3D_SUPPORTED((!ep || ep->act != DH3D_ACT_SIGMOID) && ea.act != DH3D_ACT_SIGMOID && eb.act != DH3D_ACT_SIGMOID);
  const long long R = (long long)B * N;
  const TileTail ta{static_cast<const uint4 *>(wa_x3), ea, out_a}, tb{static_cast<const uint4 *>(wb_x3), eb, out_b};
  hipLaunchKernelGGL((se_res_mfma_kernel<64, true, true, 64, true>), dim3(dh3d_cdiv(R, kTM)), dim3(256), 0,
                     (hipStream_t)stream, x, x, nbr, N, K, w1packed, b1pad, w2packed, b2, R, out, wconv_packed, dh3d_ep(ep),
                     out2, ta, tb);
  return dh3d_launch_status();
}

DH3D_API int dh3d_l2norm_concat_fwd(const float *x, int R, int C, float eps, const float *prefix, int P,
                                    float *out, void *stream) {
  DH3D_REQUIRE(x && out && R > 0 && C > 0 && P >= 0 && (P == 0 || prefix));
  hipLaunchKernelGGL(l2norm_concat_kernel, dim3(dh3d_cdiv(R, 8)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)R, C, eps, prefix, P, out);
  return dh3d_launch_status();
}

#ifdef DH3D_SE_PROBE
DH3D_API int dh3d_se_probe_read(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_seprobe), sizeof(long long) * n) == hipSuccess ? 0 : 3;
}
#endif
