// The tail of the local-descriptor forward behind the sampled level, in ONE launch (core/backbones.py:89-100,117-123,
// core/model.py:177-181):
//   y[n]   = relu(BN_c( interp3(coarse Wtop)[n] + x2[n] Wbot + b_c )) + relu(BN_s( x1[n] Ws + b_s ))
//   out[n] = [ xyz[n] | l2_normalize(y[n]) ]   (local descriptors)   or   y[n]   (the global path's input)
// Before: two 64 -> 128 GEMM launches (linear_k64_x6_kernel: the shortcut conv on x1, the concat conv's lower block on x2,
// each reading 16.7 MB and WRITING 33.5 MB at 8 x 8192 points) and interp_combine_kernel reading both maps back
// (67 MB) -- 200 MB of HBM traffic, 52 us.  Here a wave owns 32 rows: both GEMMs run from registers (bf16x6:
// f32-accurate, the A rows split in registers exactly as linear_k64_x6_kernel does), their two 32 x 128 results stay in
// the accumulators, and the up-sampling, BatchNorm / ReLU, the sum, the row normalisation and the 131-float rows are
// computed in the accumulator layout (lane = output column, 16 rows per lane): 68 MB of traffic (x1, x2 in; rows out).
//
// Workgroup = 8 waves (two per SIMD) = 8 independent 32-row tiles; the packed weights (48 KB each, fragment order) pass
// through LDS one after the other and are read back per K-step (no register double buffer: 128 accumulator registers
// leave no room for one).  Per-row interpolation operands (three row offsets, three inverse-distance weights) are
// computed once per row by lane = row and handed over through LDS.
#include "bf16x3.h"
#include "wave_ops.h"

namespace {

constexpr int kTailWaves = 8;
constexpr int kTailC = 64, kTailKB = kTailC / 16, kTailD = 128, kTailNCB = kTailD / 32;
constexpr int kTailWFrag = kTailNCB * kTailKB * 3 * 64;  // uint4 per packed weight
// ONE weight in LDS at a time (48 KB + 8 KB of row operands): a workgroup then fits on a CU beside a farthest-point-sampling
// workgroup of another step in flight (~100 KB of LDS, held for most of that step) -- with both weights resident (104 KB)
// those CUs took none of this kernel's workgroups and the 256 of them ran as two waves on the rest
constexpr size_t kTailLds = (size_t)kTailWFrag * 16 + (size_t)kTailWaves * 32 * 8 * 4;

struct TailArgs {
  const float *x1, *x2;            // [R, 64] stage-1 output, before_stage2 output
  const uint4 *wp_s, *wp_l;        // dh3d_pack_weight_x3 of the shortcut conv [64,128] and the concat conv's lower block [64,128]
  EpilogueArgs ep_s, ep_c;         // shortcut: bias / BN (ReLU); concat conv: bias / BN (ReLU)
  const float *cw;                 // [B, m, 128] coarse rows x the concat conv's upper block
  const int32_t *idx;              // [B, n, 3]
  const float *dist;               // [B, n, 3]
  const float *prefix;             // [B, n, 3] xyz
  float l2_eps;
  float *out;                      // [B, n, 131]
  int n, m;
  long long R;
};

// L2CAT: [prefix | l2_normalize(y)] rows of 131 floats; else the plain [R, 128] map y (the global path's local features)
template <bool L2CAT>
__global__ __launch_bounds__(kTailWaves * 64) void local_tail_fused_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  uint4 *s_w = reinterpret_cast<uint4 *>(s_raw);                              // [NCB][KB][3][64]: the shortcut weight, then the lower block
  float *s_rw = reinterpret_cast<float *>(s_raw + (size_t)kTailWFrag * 16);   // [waves][32 rows][8]: o1 o2 o3 (int bits) w1 w2 w3 - -
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, lr = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int kPerThread = kTailWFrag / (kTailWaves * 64);
  static_assert(kPerThread * kTailWaves * 64 == kTailWFrag, "weight copy");
  uint4 wl[kPerThread];   // the lower block: requested now, parked in registers until the shortcut GEMM has left LDS
#pragma unroll
  for (int u = 0; u < kPerThread; ++u) {
    s_w[tid + u * kTailWaves * 64] = a.wp_s[tid + u * kTailWaves * 64];
    wl[u] = a.wp_l[tid + u * kTailWaves * 64];
  }
  // (XCD-aware, round 6: contiguous tile ranges per XCD -- round-robin, every L2 fetched every cloud's coarse rows: 176 MB of
  // fetches at 32 x 4096 where the operands are 80)
  const long long row0 = ((long long)dh3d_xcd_remap(blockIdx.x, gridDim.x) * kTailWaves + wave) * 32;
  const bool alive = row0 < a.R;   // wave-uniform; n % 32 == 0: a tile lies inside one cloud and is complete
  const long long rowc = alive ? row0 : 0;
  const long long cloud = rowc / a.n;
  // ---- this lane's A rows (32 B pieces of row lr: the whole 256 B row over the four K-steps) and the row's
  // interpolation operands
  const float4 *ap1 = reinterpret_cast<const float4 *>(a.x1 + (rowc + lr) * kTailC + 8 * half);
  const float4 *ap2 = reinterpret_cast<const float4 *>(a.x2 + (rowc + lr) * kTailC + 8 * half);
  float4 av[kTailKB][2];
#pragma unroll
  for (int ks = 0; ks < kTailKB; ++ks) { av[ks][0] = ap1[ks * 4]; av[ks][1] = ap1[ks * 4 + 1]; }
  {
    const long long g = (rowc + lr) * 3;
    const int i1 = a.idx[g], i2 = a.idx[g + 1], i3 = a.idx[g + 2];
    // the inverse-distance weights of core/backbones.py:92-95, same arithmetic as three_interp_fwd_kernel<true>
    float w1, w2, w3;
    {
#pragma clang fp contract(off)
      const float r1 = 1.0f / fmaxf(a.dist[g], 1e-10f), r2 = 1.0f / fmaxf(a.dist[g + 1], 1e-10f),
                  r3 = 1.0f / fmaxf(a.dist[g + 2], 1e-10f);
      const float norm = (r1 + r2) + r3;
      w1 = r1 / norm; w2 = r2 / norm; w3 = r3 / norm;
    }
    if (half == 0) {
      float *q = s_rw + ((size_t)wave * 32 + lr) * 8;
      *reinterpret_cast<float4 *>(q) = make_float4(__int_as_float(i1 * kTailD), __int_as_float(i2 * kTailD),
                                                   __int_as_float(i3 * kTailD), w1);
      *reinterpret_cast<float2 *>(q + 4) = make_float2(w2, w3);
    }
  }
  __syncthreads();  // the shortcut weight (and this wave's row operands) are in LDS

  f32x16 acc_s[kTailNCB], acc_l[kTailNCB];
  auto gemm = [&](f32x16 (&acc)[kTailNCB], const uint4 *sw) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < kTailNCB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < kTailKB; ++ks) {
      uint4 bq[kTailNCB][3];
#pragma unroll
      for (int cb = 0; cb < kTailNCB; ++cb)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[cb][p] = sw[((size_t)(cb * kTailKB + ks) * 3 + p) * 64 + lane];
      uint2 c1[2], c2[2], c3[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) split3x4(av[ks][j], c1[j], c2[j], c3[j]);
      bf16x8 af[3];
      af[0] = __builtin_bit_cast(bf16x8, make_uint4(c1[0].x, c1[0].y, c1[1].x, c1[1].y));
      af[1] = __builtin_bit_cast(bf16x8, make_uint4(c2[0].x, c2[0].y, c2[1].x, c2[1].y));
      af[2] = __builtin_bit_cast(bf16x8, make_uint4(c3[0].x, c3[0].y, c3[1].x, c3[1].y));
#define DH3D_TAIL_PRODUCT(PA, PB)                                                                     \
  _Pragma("unroll") for (int cb = 0; cb < kTailNCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16( \
      af[PA], __builtin_bit_cast(bf16x8, bq[cb][PB]), acc[cb], 0, 0, 0);
      DH3D_TAIL_PRODUCT(2, 0) DH3D_TAIL_PRODUCT(0, 2) DH3D_TAIL_PRODUCT(1, 1)
      DH3D_TAIL_PRODUCT(1, 0) DH3D_TAIL_PRODUCT(0, 1) DH3D_TAIL_PRODUCT(0, 0)
#undef DH3D_TAIL_PRODUCT
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // ---- shortcut = relu(BN_s(x1 Ws + b_s)), in place  (waves past the last tile take part in the barriers only)
  if (alive) gemm(acc_s, s_w);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kPerThread; ++u) s_w[tid + u * kTailWaves * 64] = wl[u];
  __syncthreads();
  if (!alive) return;
#pragma unroll
  for (int ks = 0; ks < kTailKB; ++ks) { av[ks][0] = ap2[ks * 4]; av[ks][1] = ap2[ks * 4 + 1]; }  // x2 rows: in flight under the epilogue
#pragma unroll
  for (int cb = 0; cb < kTailNCB; ++cb) {
    const int col = cb * 32 + lr;
    const float pb = a.ep_s.pre_bias ? a.ep_s.pre_bias[col] : 0.f, sc = a.ep_s.scale ? a.ep_s.scale[col] : 1.f;
    const float sh = fmaf(pb, sc, a.ep_s.shift ? a.ep_s.shift[col] : 0.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_s[cb][r] = fmaxf(fmaf(acc_s[cb][r], sc, sh), 0.f);
  }
  // ---- lower block of the concat conv: x2 Wbot
  gemm(acc_l, s_w);
  // ---- up-sampling of the coarse rows + sum + BatchNorm / ReLU + shortcut, row normalisation, rows out -- in the
  // accumulator layout: register r of lane (lr, half) = row (r & 3) + 8 (r >> 2) + 4 half, column 32 cb + lr
  float pbc[kTailNCB], scc[kTailNCB], shc[kTailNCB];
#pragma unroll
  for (int cb = 0; cb < kTailNCB; ++cb) {
    const int col = cb * 32 + lr;
    pbc[cb] = a.ep_c.pre_bias ? a.ep_c.pre_bias[col] : 0.f;
    scc[cb] = a.ep_c.scale ? a.ep_c.scale[col] : 1.f;
    shc[cb] = fmaf(pbc[cb], scc[cb], a.ep_c.shift ? a.ep_c.shift[col] : 0.f);
  }
  const float *cwb = a.cw + cloud * (long long)a.m * kTailD + lr;
  constexpr int OW = L2CAT ? kTailD + 3 : kTailD;
  float *orow = a.out + row0 * OW + (L2CAT ? 3 : 0) + lr;
  const float *rw = s_rw + (size_t)wave * 32 * 8;
#pragma unroll
  for (int rg = 0; rg < 16; rg += 2) {   // two rows per lane in flight: 24 gathers
    float g[2][3][kTailNCB], w[2][3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = rg + u, row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float4 q0 = *reinterpret_cast<const float4 *>(rw + row * 8);
      const float2 q1 = *reinterpret_cast<const float2 *>(rw + row * 8 + 4);
      const int o[3] = {__float_as_int(q0.x), __float_as_int(q0.y), __float_as_int(q0.z)};
      w[u][0] = q0.w; w[u][1] = q1.x; w[u][2] = q1.y;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int cb = 0; cb < kTailNCB; ++cb) g[u][t][cb] = cwb[o[t] + cb * 32];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = rg + u, row = (r & 3) + 8 * (r >> 2) + 4 * half;
      float y[kTailNCB], ss = 0.f;
#pragma unroll
      for (int cb = 0; cb < kTailNCB; ++cb) {
        // (the same association as interp_combine_kernel: ((a w1 + b w2) + c w3) + partial, then BN with the bias folded)
        float v = fmaf(g[u][2][cb], w[u][2], fmaf(g[u][1][cb], w[u][1], g[u][0][cb] * w[u][0])) + acc_l[cb][r];
        v = fmaxf(fmaf(v, scc[cb], shc[cb]), 0.f) + acc_s[cb][r];
        y[cb] = v;
        ss = fmaf(v, v, ss);
      }
      float inv = 1.f;
      if (L2CAT) {
        ss = row16_sum_f32(ss);     // over the 32 lanes of this half (the row's 128 columns), in every lane
        ss += __shfl_xor(ss, 16, 64);
        inv = rsqrtf(fmaxf(ss, a.l2_eps));
      }
#pragma unroll
      for (int cb = 0; cb < kTailNCB; ++cb) orow[(size_t)row * OW + cb * 32] = L2CAT ? y[cb] * inv : y[cb];
    }
  }
  // the xyz prefix of the 32 rows: 96 floats
  if (L2CAT) for (int e = lane; e < 96; e += 64) {
    const int p = e / 3, c = e - 3 * p;
    a.out[(row0 + p) * (kTailD + 3) + c] = a.prefix[(row0 + p) * 3 + c];
  }
}

}  // namespace

DH3D_API int dh3d_local_tail_fused_fwd(const float *x1, const float *x2, const void *wpacked_x3_shortcut,
                                       const void *wpacked_x3_lower, const dh3d_epilogue *ep_shortcut,
                                       const dh3d_epilogue *ep_concat, const float *coarse_w, const int32_t *idx,
                                       const float *dist, const float *prefix, float l2_eps, int B, int N, int M,
                                       float *out, void *stream) {
  DH3D_REQUIRE(x1 && x2 && wpacked_x3_shortcut && wpacked_x3_lower && coarse_w && idx && dist && out && B > 0 && N > 0 && M > 0);
  DH3D_SUPPORTED(N % 32 == 0 && (long long)M * kTailD < (1ll << 31));
  DH3D_SUPPORTED((!ep_shortcut || ep_shortcut->act == DH3D_ACT_RELU) && (!ep_concat || ep_concat->act == DH3D_ACT_RELU));
  TailArgs a;
  a.x1 = x1; a.x2 = x2;
  a.wp_s = static_cast<const uint4 *>(wpacked_x3_shortcut); a.wp_l = static_cast<const uint4 *>(wpacked_x3_lower);
  a.ep_s = dh3d_ep(ep_shortcut); a.ep_c = dh3d_ep(ep_concat);
  a.cw = coarse_w; a.idx = idx; a.dist = dist; a.prefix = prefix; a.l2_eps = l2_eps; a.out = out;
  a.n = N; a.m = M; a.R = (long long)B * N;
  const dim3 grid(dh3d_cdiv(a.R, 32 * kTailWaves)), block(kTailWaves * 64);
  if (prefix) {
    DH3D_ALLOW_BIG_LDS(local_tail_fused_kernel<true>);
    hipLaunchKernelGGL(local_tail_fused_kernel<true>, grid, block, kTailLds, (hipStream_t)stream, a);
  } else {   // no prefix: the plain [B, N, 128] map (l2_eps unused)
    DH3D_ALLOW_BIG_LDS(local_tail_fused_kernel<false>);
    hipLaunchKernelGGL(local_tail_fused_kernel<false>, grid, block, kTailLds, (hipStream_t)stream, a);
  }
  return dh3d_launch_status();
}
