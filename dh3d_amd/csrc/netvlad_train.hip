// NetVLAD's assignment in the TRAINING step with its rows commuted through the up-sampling (gfx950).
//
// netvlad (core/backbones.py:202-262) runs on the three_interpolate'd rows x[n] = sum_t w_t c[i_t] (core/backbones.py:
// 89-100; c: the B*m sampled rows, 256 wide):  xn = l2_normalize(x),  s = xn Wc,  z = BN_train(s),  a = softmax(z) att,
// V[b] = a^T xn,  asum[b] = sum_n a.  With r[n] = rsqrt(max(|x[n]|^2, 1e-12)) and cw = c Wc (a GEMM on the sampled rows)
//     s[n]  = r[n] sum_t w_t cw[i_t]                         V[b] = A'^T c,   A'[j,k] = sum_{(n,t): i_t = j} a[n,k] r[n] w_t
// and, given dV and dasum (E = c dV^T, a GEMM on the sampled rows; e = interp(E)),
//     da = r e + dasum;   datt = sum_k da p;   dz = p (da att - sum_k da att p);   ds = BN backward of dz;
//     dcw[j] = sum_{(n,t)} w_t r ds;   dWc = c^T dcw;   q[n] = r^2 sum_k ds s + r^3 sum_k a e;
//     dc = A' dV + dcw Wc^T - interp^T(q x)
// (algebra checked against float64 autograd by tools/netvlad_commute_check.py and tests/test_host_logic.py).  Nothing
// 256 wide is ever written for the B*n fine points: the materialised form moved x, xn, dxn, dx (92 MB each at 22 x
// 4096 points) through nine kernels and five GEMMs on 90 k rows (~490 us); here the per-point state is 64 wide (s, p,
// dz) and the GEMMs run on the 11 k sampled rows.  Four walks over the fine points in Morton order (the walk of
// interp_train.hip: 128 points per workgroup, the <= 64 distinct coarse rows of a block found with a bitmap and staged
// in LDS; a wave owns a point, a lane a cluster, softmax and the row sums are wave reductions on the DPP crossbar):
//   MODE 0  r, s and the column statistics of s (f64, one partial row per cloud)            stages c and cw rows
//   MODE 1  p = softmax(s scale + shift), a = p att, asum, A' (scatter)                      no staging
//   MODE 2  da, datt, dz, t2 = sum_k a e and the BatchNorm backward sums S1, S2             stages E rows
//   MODE 3  ds = k1 dz - k2 - k3 s, q, dcw (scatter)                                        no staging
// The two scatters are the [64 x 128] x [128 x 64] MFMA product on a slot matrix built in registers of
// interp_head_lds_kernel<true> (dense_x6.hip).  The last term of dc is MODE 4 of interp_bn_kernel (interp_train.hip).
// Rows of padding clouds (mask) take no part.
#include "common.h"
#include "interp_walk.h"
#include "wave_ops.h"

#include <type_traits>

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

using dh3d_walk::kP;
using dh3d_walk::mix3;
constexpr int kPW = kP / 4;
constexpr int kCap0 = 56;  // staged rows in MODE 0 (c and cw rows: 71.7 KB; blocks touching more read the excess from L2)
constexpr int kCap = 64;   // slots in the other modes

struct NvArgs {
  const float *c;        // [B*m, 256] sampled rows (MODE 0)
  const float *rows64;   // [B*m, 64]: cw = c Wc (MODE 0), E = c dV^T (MODE 2)
  const int32_t *idx;    // [B, n, 3]
  const float *dist;     // [B, n, 3]
  const float4 *order;   // [B, n] spatial_sort records of the fine cloud (may be null: index order)
  int B, n, m, nblk;
  const unsigned char *mask;  // [B] or null
  float *s;              // [B*n, 64] by original point index: written by MODE 0, read by 1, 2, 3
  float *rinv;           // [B*n]
  const float *att;      // [B*n]
  float *p;              // [B*n, 64]: written by MODE 1, read by MODE 2
  float *dz;             // [B*n, 64]: written by MODE 2, read by MODE 3
  float *datt, *t2, *q;  // [B*n]
  const float *v0, *v1, *v2;  // MODE 1: scale, shift;  MODE 2: mean, rstd;  MODE 3: k1 (= scale), k2, k3
  const float *dasum;    // [B, 64] (MODE 2)
  double *s0, *s1;       // [B][64] per-cloud partials: sum / sumsq (MODE 0), S1 / S2 (MODE 2)
  float *asum;           // [B, 64] (MODE 1, atomics)
  float *scat;           // [B*m, 64]: A' (MODE 1), dcw (MODE 3); atomics
};

__device__ __forceinline__ float bcast(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

template <int MODE>
__global__ __launch_bounds__(256) void nv_walk_kernel(const NvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  constexpr int CAP = MODE == 0 ? kCap0 : kCap;
  constexpr int BUF = MODE == 0 ? kCap0 * 320 : MODE == 2 ? kCap * 64 : kP * 64;  // rows / E rows / a or ds of the block
  float *s_buf = s_mem;
  int *s_slot = reinterpret_cast<int *>(s_buf + BUF);             // [kP][4] slots (or -1 - coarse row), .w = 1 + original index (0: none)
  float *s_w = reinterpret_cast<float *>(s_slot + kP * 4);        // [kP][4] weights
  float *s_f0 = s_w + kP * 4;                                     // [kP] per-point scalars: att
  float *s_f1 = s_f0 + kP;                                        //      rinv
  float *s_f2 = s_f1 + kP;                                        //      t2
  float *s_red = s_f2 + kP;                                       // [2][4][64] partial sums of the four waves / asum
  unsigned *s_bits = reinterpret_cast<unsigned *>(s_red + 512);   // [32]
  int *s_pre = reinterpret_cast<int *>(s_bits + 32);              // [33]
  int *s_row = s_pre + 33;                                        // [kCap]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int bi = xcd + 8 * (seq / a.nblk), blk = seq % a.nblk;
  if (bi >= a.B) return;
  const int n = a.n, m = a.m;
  if (a.mask && !a.mask[bi]) {  // a padding cloud: no statistics, zero gradients
    if (MODE == 2 && tid < kP && blk * kP + tid < n) a.datt[(size_t)bi * n + blk * kP + tid] = 0.f;
    return;
  }
  if (tid < kP) { s_f0[tid] = 0.f; s_f1[tid] = 0.f; s_f2[tid] = 0.f; }  // (padding points keep these)
  const dh3d_walk::SlotTable tab{s_slot, s_w, s_bits, s_pre, s_row};
  const int distinct = dh3d_walk::build_slot_table<CAP>(tab, a.idx, a.dist, nullptr, a.order, bi, blk, n, m,
                                                        [&](long long r, int t) {
                                                          // per-point scalars of the live points: att, rinv, t2
                                                          if (MODE == 1 || MODE == 2) s_f0[t] = a.att[r];
                                                          if (MODE != 0) s_f1[t] = a.rinv[r];
                                                          if (MODE == 3) s_f2[t] = a.t2[r];
                                                          return 0.f;
                                                        });
  const int nd = min(distinct, CAP);
  const bool overflow = distinct > CAP;

  // this wave's points: original row of point p (or -1), every lane the same value
  auto orig_of = [&](int p) { return __builtin_amdgcn_readfirstlane(s_slot[(wave * kPW + p) * 4 + 3]) - 1; };
  const size_t rbase = (size_t)bi * n;

  // ---- the scatter of MODE 1 / 3: out[slot, k] += sum_p S[p, slot] val[p, k],  S[p, slot_t(p)] = w_t(p) r(p).
  // Wave = one 32x32 tile of the [64 slots x 64 clusters] result: slots 32 (wave >> 1).., clusters 32 (wave & 1)..
  auto scatter = [&](const float *s_val) __attribute__((always_inline)) {
    float *out = a.scat + (size_t)bi * m * 64;
    const int ti = wave >> 1, tj = wave & 1;
    if (ti * 32 < nd) {  // uniform per wave pair
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int jrow = ti * 32 + (lane & 31), kk = lane >> 5;
#pragma unroll 4
      for (int st = 0; st < kP / 2; ++st) {
        const int pt = 2 * st + kk;
        const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
        const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
        const float sv = (si.x == jrow ? sw.x : 0.f) + (si.y == jrow ? sw.y : 0.f) + (si.z == jrow ? sw.z : 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sv * s_f1[pt], s_val[pt * 64 + tj * 32 + (lane & 31)], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;  // 32x32 accumulator layout
        if (j < nd) unsafeAtomicAdd(&out[(size_t)s_row[j] * 64 + tj * 32 + (lane & 31)], acc[r]);
      }
    }
  };
  // rows that did not fit the slot table (MODE 1 / 3, CAP = 64: rare): straight to memory
  auto scatter_overflow = [&](const int (&sl)[3], const float (&wt)[3], float rv, float val) __attribute__((always_inline)) {
    float *out = a.scat + (size_t)bi * m * 64;
#pragma unroll
    for (int t = 0; t < 3; ++t)
      if (sl[t] < 0) unsafeAtomicAdd(&out[(size_t)(-1 - sl[t]) * 64 + lane], val * wt[t] * rv);
  };
  // the four waves hold different points of the same clusters: through LDS, then one f64 atomic per cluster and sum
  auto flush_stats = [&](float A1, float A2) __attribute__((always_inline)) {
    s_red[(0 * 4 + wave) * 64 + lane] = A1;
    s_red[(1 * 4 + wave) * 64 + lane] = A2;
    __syncthreads();
    if (tid < 128) {
      const int k = tid >> 6, cl = tid & 63;
      const double v = ((double)s_red[(k * 4 + 0) * 64 + cl] + s_red[(k * 4 + 1) * 64 + cl]) +
                       ((double)s_red[(k * 4 + 2) * 64 + cl] + s_red[(k * 4 + 3) * 64 + cl]);
      unsafeAtomicAdd((k == 0 ? a.s0 : a.s1) + (size_t)bi * 64 + cl, v);
    }
  };

  if constexpr (MODE == 0) {
    // ---- stage the c rows (256 floats) and the cw rows (64 floats) of the block's slots
    float *s_rows = s_buf, *s_cw = s_buf + kCap0 * 256;
    const float *cb = a.c + (size_t)bi * m * 256, *wb = a.rows64 + (size_t)bi * m * 64;
#pragma unroll
    for (int u = 0; u < kCap0 / 4; ++u) {
      const int r = wave + 4 * u;
      *reinterpret_cast<float4 *>(s_rows + (size_t)r * 256 + lane * 4) =
          *reinterpret_cast<const float4 *>(cb + (size_t)s_row[r < nd ? r : 0] * 256 + lane * 4);
    }
    for (int e = tid; e < nd * 16; e += 256) {
      const int r = e >> 4, q4 = e & 15;
      *reinterpret_cast<float4 *>(s_cw + r * 64 + q4 * 4) = *reinterpret_cast<const float4 *>(wb + (size_t)s_row[r] * 64 + q4 * 4);
    }
    __syncthreads();
    float A1 = 0.f, A2 = 0.f;
    auto points = [&](auto ovf) __attribute__((always_inline)) {
      constexpr bool OVF = decltype(ovf)::value;
      for (int p = 0; p < kPW; p += 2) {
        float part[2], u[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pt = wave * kPW + p + h;
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const int sl[3] = {__builtin_amdgcn_readfirstlane(si.x), __builtin_amdgcn_readfirstlane(si.y),
                             __builtin_amdgcn_readfirstlane(si.z)};
          const float wt[3] = {sw.x, sw.y, sw.z};
          float4 rw[3];
          float cv[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            if (OVF && sl[t] < 0) {
              rw[t] = *reinterpret_cast<const float4 *>(cb + (size_t)(-1 - sl[t]) * 256 + lane * 4);
              cv[t] = wb[(size_t)(-1 - sl[t]) * 64 + lane];
            } else {
              rw[t] = *reinterpret_cast<const float4 *>(s_rows + (size_t)sl[t] * 256 + lane * 4);
              cv[t] = s_cw[sl[t] * 64 + lane];
            }
          }
          const float4 x = mix3(rw[0], rw[1], rw[2], wt[0], wt[1], wt[2]);
          part[h] = fmaf(x.w, x.w, fmaf(x.z, x.z, fmaf(x.y, x.y, x.x * x.x)));
          u[h] = fmaf(wt[2], cv[2], fmaf(wt[1], cv[1], wt[0] * cv[0]));
        }
        const float t01 = pair_wave_sum_f32(part[0], part[1]);
        const float tot[2] = {bcast(t01, 16), bcast(t01, 48)};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int o = orig_of(p + h);
          if (o >= 0) {  // wave-uniform
            const float rv = rsqrtf(fmaxf(tot[h], 1e-12f));  // tf.nn.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))
            const float sv = rv * u[h];
            a.s[(rbase + o) * 64 + lane] = sv;
            if (lane == 0) a.rinv[rbase + o] = rv;
            A1 += sv;
            A2 = fmaf(sv, sv, A2);
          }
        }
      }
    };
    if (overflow) points(std::true_type{}); else points(std::false_type{});
    flush_stats(A1, A2);
  } else if constexpr (MODE == 1) {
    float *s_a = s_buf;  // [kP][64]
    float sv[kPW];
#pragma unroll
    for (int p = 0; p < kPW; ++p) {
      const int o = orig_of(p);
      sv[p] = a.s[(rbase + (o >= 0 ? o : 0)) * 64 + lane];
    }
    if (tid < 64) s_red[tid] = 0.f;
    const float csc = a.v0[lane], csh = a.v1[lane];
    float asum_acc = 0.f;
#pragma unroll
    for (int p = 0; p < kPW; p += 2) {
      float e2[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float z = fmaf(sv[p + h], csc, csh);
        e2[h] = __expf(z - wave_max_f32(z));
      }
      const float hs = pair_wave_sum_f32(e2[0], e2[1]);
      const float sum[2] = {bcast(hs, 16), bcast(hs, 48)};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pt = wave * kPW + p + h;
        const int o = orig_of(p + h);
        const float pv = o >= 0 ? e2[h] * __frcp_rn(sum[h]) : 0.f;
        const float av = pv * s_f0[pt];
        if (o >= 0) a.p[(rbase + o) * 64 + lane] = pv;
        s_a[pt * 64 + lane] = av;
        asum_acc += av;
        if (overflow) {
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const int sl[3] = {__builtin_amdgcn_readfirstlane(si.x), __builtin_amdgcn_readfirstlane(si.y),
                             __builtin_amdgcn_readfirstlane(si.z)};
          const float wt[3] = {sw.x, sw.y, sw.z};
          scatter_overflow(sl, wt, s_f1[pt], av);
        }
      }
    }
    __syncthreads();  // s_red zeroed
    unsafeAtomicAdd(&s_red[lane], asum_acc);
    __syncthreads();  // s_a complete, asum complete
    scatter(s_a);
    if (tid < 64) unsafeAtomicAdd(&a.asum[(size_t)bi * 64 + tid], s_red[tid]);
  } else if constexpr (MODE == 2) {
    float *s_e = s_buf;  // [kCap][64] rows of E
    const float *eb = a.rows64 + (size_t)bi * m * 64;
    for (int e = tid; e < nd * 16; e += 256) {
      const int r = e >> 4, q4 = e & 15;
      *reinterpret_cast<float4 *>(s_e + r * 64 + q4 * 4) = *reinterpret_cast<const float4 *>(eb + (size_t)s_row[r] * 64 + q4 * 4);
    }
    float pv[kPW], sv[kPW];
#pragma unroll
    for (int p = 0; p < kPW; ++p) {
      const int o = orig_of(p);
      const float pl = a.p[(rbase + (o >= 0 ? o : 0)) * 64 + lane];
      pv[p] = o >= 0 ? pl : 0.f;
      sv[p] = a.s[(rbase + (o >= 0 ? o : 0)) * 64 + lane];
    }
    const float das = a.dasum[(size_t)bi * 64 + lane], mean = a.v0[lane], rstd = a.v1[lane];
    __syncthreads();
    float A1 = 0.f, A2 = 0.f;
    auto points = [&](auto ovf) __attribute__((always_inline)) {
      constexpr bool OVF = decltype(ovf)::value;
#pragma unroll
      for (int p = 0; p < kPW; p += 2) {
        float ev[2], dp[2], pa[2], pb[2], pc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pt = wave * kPW + p + h;
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const int sl[3] = {__builtin_amdgcn_readfirstlane(si.x), __builtin_amdgcn_readfirstlane(si.y),
                             __builtin_amdgcn_readfirstlane(si.z)};
          const float wt[3] = {sw.x, sw.y, sw.z};
          float e = 0.f;
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const float rowv = (OVF && sl[t] < 0) ? eb[(size_t)(-1 - sl[t]) * 64 + lane] : s_e[sl[t] * 64 + lane];
            e = fmaf(wt[t], rowv, e);
          }
          const float att = s_f0[pt];
          const float da = fmaf(s_f1[pt], e, das);
          ev[h] = e;
          dp[h] = da * att;
          pa[h] = da * pv[p + h];             // -> datt
          pb[h] = dp[h] * pv[p + h];          // -> softmax backward's inner product
          pc[h] = pv[p + h] * att * e;        // -> t2 = sum_k a e
        }
        const float ra = pair_wave_sum_f32(pa[0], pa[1]), rb = pair_wave_sum_f32(pb[0], pb[1]),
                    rc = pair_wave_sum_f32(pc[0], pc[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int o = orig_of(p + h);
          if (o >= 0) {  // wave-uniform
            const int hl = h == 0 ? 16 : 48;
            const float dzv = pv[p + h] * (dp[h] - bcast(rb, hl));
            a.dz[(rbase + o) * 64 + lane] = dzv;
            if (lane == 0) {
              a.datt[rbase + o] = bcast(ra, hl);
              a.t2[rbase + o] = bcast(rc, hl);
            }
            const float sh = (sv[p + h] - mean) * rstd;
            A1 += dzv;
            A2 = fmaf(dzv, sh, A2);
          }
        }
      }
    };
    points(std::true_type{});  // (64 slots: overflow is rare and its test is a scalar branch -- one copy of the loop)
    flush_stats(A1, A2);
  } else {
    float *s_d = s_buf;  // [kP][64] ds of the block's points
    float dzv[kPW], sv[kPW];
#pragma unroll
    for (int p = 0; p < kPW; ++p) {
      const int o = orig_of(p);
      dzv[p] = a.dz[(rbase + (o >= 0 ? o : 0)) * 64 + lane];
      sv[p] = a.s[(rbase + (o >= 0 ? o : 0)) * 64 + lane];
    }
    const float k1 = a.v0[lane], k2 = a.v1[lane], k3 = a.v2[lane];
#pragma unroll
    for (int p = 0; p < kPW; p += 2) {
      float ds[2], g[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int o = orig_of(p + h);
        ds[h] = o >= 0 ? fmaf(k1, dzv[p + h], -k2) - k3 * sv[p + h] : 0.f;
        g[h] = ds[h] * sv[p + h];
      }
      const float rg = pair_wave_sum_f32(g[0], g[1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pt = wave * kPW + p + h;
        const int o = orig_of(p + h);
        s_d[pt * 64 + lane] = ds[h];
        if (o >= 0 && lane == 0) {
          const float rv = s_f1[pt];
          a.q[rbase + o] = rv * rv * (bcast(rg, h == 0 ? 16 : 48) + rv * s_f2[pt]);  // r^2 sum ds s + r^3 sum a e
        }
        if (overflow) {
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const int sl[3] = {__builtin_amdgcn_readfirstlane(si.x), __builtin_amdgcn_readfirstlane(si.y),
                             __builtin_amdgcn_readfirstlane(si.z)};
          const float wt[3] = {sw.x, sw.y, sw.z};
          scatter_overflow(sl, wt, s_f1[pt], ds[h]);
        }
      }
    }
    __syncthreads();
    scatter(s_d);
  }
}

size_t nv_lds(int mode) {
  const size_t buf = mode == 0 ? (size_t)kCap0 * 320 : mode == 2 ? (size_t)kCap * 64 : (size_t)kP * 64;
  return sizeof(float) * (buf + kP * 4 * 2 + 3 * kP + 512 + 32 + 33 + kCap + 3);  // 78.9 KB (MODE 0), 39.9 KB, 23.5 KB
}

template <int MODE>
int launch(const NvArgs &a, hipStream_t s) {
  auto kern = nv_walk_kernel<MODE>;
  DH3D_ALLOW_BIG_LDS(kern);
  const int per_xcd = dh3d_cdiv(a.B, 8) * a.nblk;
  hipLaunchKernelGGL(kern, dim3(8 * per_xcd), dim3(256), nv_lds(MODE), s, a);
  return dh3d_launch_status();
}

NvArgs base_args(const int32_t *idx, const float *dist, const float *order, int B, int n, int m, const unsigned char *mask) {
  NvArgs a{};
  a.idx = idx; a.dist = dist; a.order = reinterpret_cast<const float4 *>(order);
  a.B = B; a.n = n; a.m = m; a.nblk = dh3d_cdiv(n, kP); a.mask = mask;
  return a;
}

}  // namespace

// c [B*m, 256] sampled rows, cw [B*m, 64] = c Wc; idx / dist: three_nn of the fine points [B,n,3]; order: spatial_sort
// records of the fine clouds [B,n,4] (may be NULL); mask [B] bytes (may be NULL).  Writes s [B*n, 64] and rinv [B*n] by
// original point index and the per-cloud partial column sums / sums of squares of s: part [2][B][64] f64 (zeroed by the CALLER).
DH3D_API int dh3d_netvlad_commuted_fwd_stats(const float *c, const float *cw, const int32_t *idx, const float *dist,
                                             const float *order, int B, int n, int m, const unsigned char *mask, float *s,
                                             float *rinv, double *part, void *stream) {
  DH3D_REQUIRE(c && cw && idx && dist && s && rinv && part && B > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(m <= 1024);
  hipStream_t st = (hipStream_t)stream;
  NvArgs a = base_args(idx, dist, order, B, n, m, mask);
  a.c = c; a.rows64 = cw; a.s = s; a.rinv = rinv; a.s0 = part; a.s1 = part + (size_t)B * 64;
  return launch<0>(a, st);
}

// p [B*n, 64] = softmax(s scale + shift);  asum [B, 64] = sum_n p att;  Ap [B*m, 64] = A' (both zeroed by the CALLER, f32 atomics).
DH3D_API int dh3d_netvlad_commuted_fwd_assign(const float *s, const float *rinv, const float *att, const float *scale,
                                              const float *shift, const int32_t *idx, const float *dist, const float *order,
                                              int B, int n, int m, const unsigned char *mask, float *p, float *asum,
                                              float *Ap, void *stream) {
  DH3D_REQUIRE(s && rinv && att && scale && shift && idx && dist && p && asum && Ap && B > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(m <= 1024);
  hipStream_t st = (hipStream_t)stream;
  NvArgs a = base_args(idx, dist, order, B, n, m, mask);
  a.s = const_cast<float *>(s); a.rinv = const_cast<float *>(rinv); a.att = att; a.v0 = scale; a.v1 = shift;
  a.p = p; a.asum = asum; a.scat = Ap;
  return launch<1>(a, st);
}

// E [B*m, 64] = c dV^T;  dasum [B, 64];  mean / rstd [64] of the forward BatchNorm.  Writes dz [B*n, 64], datt [B*n]
// (0 for padding clouds), t2 [B*n] and the per-cloud partials of S1 = sum dz, S2 = sum dz shat: part [2][B][64] f64
// (zeroed by the CALLER).
DH3D_API int dh3d_netvlad_commuted_bwd_sums(const float *E, const float *p, const float *s, const float *att,
                                            const float *rinv, const float *dasum, const float *mean, const float *rstd,
                                            const int32_t *idx, const float *dist, const float *order, int B, int n, int m,
                                            const unsigned char *mask, float *dz, float *datt, float *t2, double *part,
                                            void *stream) {
  DH3D_REQUIRE(E && p && s && att && rinv && dasum && mean && rstd && idx && dist && dz && datt && t2 && part);
  DH3D_REQUIRE(B > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(m <= 1024);
  hipStream_t st = (hipStream_t)stream;
  NvArgs a = base_args(idx, dist, order, B, n, m, mask);
  a.rows64 = E; a.p = const_cast<float *>(p); a.s = const_cast<float *>(s); a.att = att; a.rinv = const_cast<float *>(rinv);
  a.dasum = dasum; a.v0 = mean; a.v1 = rstd; a.dz = dz; a.datt = datt; a.t2 = t2;
  a.s0 = part; a.s1 = part + (size_t)B * 64;
  return launch<2>(a, st);
}

// ds = k1 dz - k2 - k3 s (the coefficients of dh3d_bn_bwd_finalize, k1 = scale);  q [B*n] = r^2 sum_k ds s + r^3 t2;
// dcw [B*m, 64] = interp^T(r ds) (zeroed by the CALLER, f32 atomics).
DH3D_API int dh3d_netvlad_commuted_bwd_apply(const float *dz, const float *s, const float *rinv, const float *t2,
                                             const float *k1, const float *k2, const float *k3, const int32_t *idx,
                                             const float *dist, const float *order, int B, int n, int m,
                                             const unsigned char *mask, float *q, float *dcw, void *stream) {
  DH3D_REQUIRE(dz && s && rinv && t2 && k1 && k2 && k3 && idx && dist && q && dcw && B > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(m <= 1024);
  hipStream_t st = (hipStream_t)stream;
  NvArgs a = base_args(idx, dist, order, B, n, m, mask);
  a.dz = const_cast<float *>(dz); a.s = const_cast<float *>(s); a.rinv = const_cast<float *>(rinv); a.t2 = const_cast<float *>(t2);
  a.v0 = k1; a.v1 = k2; a.v2 = k3; a.q = q; a.scat = dcw;
  return launch<3>(a, st);
}
