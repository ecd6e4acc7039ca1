// Training-mode attention head with its 256 -> 1024 convolution commuted through the up-sampling (gfx950).
//
// globalatt_block (core/backbones.py:156-173) runs  conv 256->1024 -> BatchNorm (batch statistics) -> ReLU -> fc 1024->1
// -> sigmoid  on the three_interpolate'd rows (core/backbones.py:89-100).  The interpolation is linear and acts on rows,
// the convolution is linear and acts on channels, so  conv(interp(c)) = interp(conv(c)):  the three 47-GFLOP exact-f32
// GEMMs of the training step (forward, dX, dW on Bt*N = 90 k rows) become 6-GFLOP ones on the Bt*N/8 sampled rows, and
// the [90 k x 1024] pre-activation h = interp(G), G = c W + b, is never written: every pass that needs it walks the fine
// points in Morton order (records of dh3d_spatial_sort), 128 points per workgroup, finds the <= 64 distinct coarse rows
// the block touches with a bitmap, stages them in LDS one 256-channel slice at a time and mixes three of them per point
// -- the walk of interp_head_lds_kernel (dense_x6.hip), which stays the forward pass proper (its BatchNorm epilogue
// takes the batch statistics).  The three passes here:
//   MODE 0  column statistics of h:          sum_n h[n,c], sum_n h[n,c]^2                      (forward, before the head)
//   MODE 1  BatchNorm backward sums:         S1 = sum dz, S2 = sum dz * xhat, S3 = sum dlogit * relu(y) (= d w_fc)
//   MODE 2  dh = k1 dz - k2 - k3 h  scattered back through the interpolation:  dG[i_t(n), c] += w_t(n) dh[n, c]
//   MODE 3  the same scatter for MATERIALISED gradient rows dY [B*n, 256] and explicit weights: the backward of
//           three_interpolate (tf_interpolate.cpp:131-153) without one global atomic per (point, neighbour, channel)
//   MODE 4  dh = -q[n] h[n] with a per-POINT factor q, ADDED to dG: the l2-normalisation term of NetVLAD's commuted
//           backward (netvlad_train.hip: dc -= interp^T(q x), x = interp(c) rebuilt from the staged rows)
// with dz[n,c] = dlogit[n] * w_fc[c] * [y > 0] (the rank-one gradient of train.hip's attention head).  The scatter of
// MODE 2 is a product on the matrix cores: per 32 points, dG_tile[slot, c] += sum_n S[n, slot] dh[n, c] with S built in
// registers from the slot table (S[n, slot_t(n)] = w_t(n)) -- LDS float atomics retire about one lane per 2.4 cycles
// and three_interp_bwd's global atomics take 218 us for a quarter of the channels; the staged tile is added to memory
// once per block and slice (f32 atomics on <= 64 rows).  Rows of padding clouds (mask) take no part.
#include "common.h"
#include "interp_walk.h"
#include "wave_ops.h"

#include <type_traits>

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

using dh3d_walk::idw3;
using dh3d_walk::kP;
using dh3d_walk::mix3;
constexpr int kCap = 64;   // staged coarse rows per slice (a 128-point block touches 46 on average, 62 at most)
constexpr int kPW = kP / 4;
// MODE 2 keeps a chunk of dh rows next to the staged rows; 56 slots + 16 points x 272 floats leave room for two
// workgroups per CU (79.6 KB each; with 64 + 32 x 288 only one fitted and the pass took 375 instead of ~300 us).
constexpr int kCap2 = 56;  // staged rows in MODE 2 (blocks touching more take the overflow path for the excess)
constexpr int kCH = 16;    // points per dh chunk
constexpr int kLDH = 272;  // row stride of the dh chunk (floats)

template <bool OVF>
__device__ __forceinline__ float4 row4(const float *s_rows, const float *gbase, int slot, int lane, int rs) {
  if (OVF && slot < 0) return *reinterpret_cast<const float4 *>(gbase + (size_t)(-1 - slot) * rs + lane * 4);
  return *reinterpret_cast<const float4 *>(s_rows + (size_t)slot * 256 + lane * 4);
}

// The parked rows live in ONE 64-float vector value (an SSA value: a float4 array of the same size was kept in scratch
// -- stored right behind every load and reloaded -- whatever the control flow around it looked like).
typedef float f32x64 __attribute__((ext_vector_type(64)));
template <int NR>
__device__ __forceinline__ f32x64 request_rows(const float *Gslice, const int (&rowoff)[NR]) {
  f32x64 rg;
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const float4 v = *reinterpret_cast<const float4 *>(Gslice + rowoff[u]);
    rg[4 * u] = v.x; rg[4 * u + 1] = v.y; rg[4 * u + 2] = v.z; rg[4 * u + 3] = v.w;
  }
  return rg;
}

struct InterpBnArgs {
  const float *G;            // coarse @ W + b: 256-column slices [NS][Rc][256] (SS = Rc*256, RS = 256) or row-major
  int NS;                    //   [Rc][Hd] (SS = 256, RS = Hd); dG has the same layout
  long long Rc;              // B * m
  long long SS;              // slice stride (floats)
  int RS;                    // row stride (floats)
  const int32_t *idx;        // [B, n, 3]
  const float *dist;         // [B, n, 3]
  const float4 *order;       // [B, n] spatial_sort records of the fine cloud
  int B, n, m, nblk;
  const unsigned char *mask; // [B] or null
  const float *dlogit;       // [B * n] by original point index (MODE 1, 2)
  const float *wfc;          // [NS * 256]
  const float *v0, *v1, *v2, *v3;  // MODE 1: mean, rstd, gamma, beta;  MODE 2: scale, shift, k2, k3
  double *s0, *s1, *s2;      // MODE 0: sum, sumsq;  MODE 1: S1, S2, S3 -- one partial row [Hd] per cloud
  float *dG;                 // MODE 2: [NS][Rc][256]
  const float *dY;           // MODE 3: [B * n, 256] gradient rows by original point index
  const float *weight;       // MODE 3: [B, n, 3] interpolation weights (instead of dist)
};

template <int MODE>
__global__ __launch_bounds__(256) void interp_bn_kernel(const InterpBnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  constexpr int CAP = MODE >= 2 ? kCap2 : kCap;  // (MODE 4 = MODE 2 with another dh)
  constexpr int ROWS = MODE == 3 ? 0 : CAP;                        // MODE 3 stages nothing
  float *s_rows = s_mem;                                           // [ROWS][256]
  int *s_slot = reinterpret_cast<int *>(s_rows + ROWS * 256);      // [kP][4] slots (or -1 - coarse row), .w = 1 + original index (0: none)
  float *s_w = reinterpret_cast<float *>(s_slot + kP * 4);         // [kP][4] weights, .w = dlogit (0: padding)
  unsigned *s_bits = reinterpret_cast<unsigned *>(s_w + kP * 4);   // [32]
  int *s_pre = reinterpret_cast<int *>(s_bits + 32);               // [33]
  int *s_row = s_pre + 33;                                         // [kCap]
  float *s_x = reinterpret_cast<float *>(s_row + kCap + 3);        // MODE 2: [kCH][kLDH] dh chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int bi = xcd + 8 * (seq / a.nblk), blk = seq % a.nblk;
  if (bi >= a.B) return;
  if (a.mask && !a.mask[bi]) return;  // a padding cloud: no statistics, zero gradients
  const int n = a.n, m = a.m;
  const dh3d_walk::SlotTable tab{s_slot, s_w, s_bits, s_pre, s_row};
  // the per-point scalar: dlogit (MODE 1, 2) / q (MODE 4); MODE 3 brings its own interpolation weights
  const int distinct = dh3d_walk::build_slot_table<CAP>(
      tab, a.idx, a.dist, MODE == 3 ? a.weight : nullptr, a.order, bi, blk, n, m,
      [&](long long r, int) { return (MODE == 0 || MODE == 3) ? 0.f : a.dlogit[r]; });
  const int nd = min(distinct, CAP);
  const bool overflow = distinct > CAP;

  // The rows of slice sl + 1 are requested BEFORE the points of slice sl are worked on and parked in registers (CAP / 4
  // float4 per lane) until the buffer is free: staging and per-point work each took ~32 us of the 97 us statistics
  // pass when they ran one after the other.
  static_assert(CAP / 4 <= 16, "parked rows: one f32x64");
  f32x64 rg = {};
  int rowoff[CAP / 4];  // this lane's float offset into a slice for each of its rows
#pragma unroll
  for (int u = 0; u < CAP / 4; ++u) {
    const int r = wave + 4 * u;
    rowoff[u] = (bi * m + s_row[r < nd ? r : 0]) * a.RS + lane * 4;
  }
  // (not in MODE 2: 64 more registers would leave one wave per SIMD there)
  constexpr bool PREFETCH = MODE != 2 && MODE != 4;
#if !defined(DH3D_IB_EXP) || !(DH3D_IB_EXP & 16)   // exp 16: no staging (results wrong)
  if (MODE != 3 && PREFETCH) rg = request_rows<CAP / 4>(a.G, rowoff);
#endif
  for (int sl = 0; sl < a.NS; ++sl) {
    const float *Gs = MODE == 3 ? nullptr : a.G + (size_t)sl * a.SS + (size_t)bi * m * a.RS;
#if !defined(DH3D_IB_EXP) || !(DH3D_IB_EXP & 16)
    if (MODE != 3) {
      if (!PREFETCH) rg = request_rows<CAP / 4>(a.G + (size_t)sl * a.SS, rowoff);
#pragma unroll
      for (int u = 0; u < CAP / 4; ++u) {  // all CAP slots, used or not: no branch between the loads and these stores
        const int r = wave + 4 * u;
        *reinterpret_cast<float4 *>(s_rows + (size_t)r * 256 + lane * 4) =
            make_float4(rg[4 * u], rg[4 * u + 1], rg[4 * u + 2], rg[4 * u + 3]);
      }
      // (unconditional -- the last slice once more: a load under a branch would be merged through memory)
      if (PREFETCH) rg = request_rows<CAP / 4>(a.G + (size_t)(sl + 1 < a.NS ? sl + 1 : sl) * a.SS, rowoff);
    }
#endif
    const int c = sl * 256 + lane * 4;
    float4 q0 = {}, q1 = {}, q2 = {}, q3 = {}, wf = {};
    if (MODE == 1 || MODE == 2) {
      q0 = *reinterpret_cast<const float4 *>(a.v0 + c); q1 = *reinterpret_cast<const float4 *>(a.v1 + c);
      q2 = *reinterpret_cast<const float4 *>(a.v2 + c); q3 = *reinterpret_cast<const float4 *>(a.v3 + c);
      wf = *reinterpret_cast<const float4 *>(a.wfc + c);
    }
    __syncthreads();

    if (MODE == 0 || MODE == 1) {
      float4 A1 = {}, A2 = {}, A3 = {};
      auto points = [&](auto ovf) __attribute__((always_inline)) {
        constexpr bool OVF = decltype(ovf)::value;
#pragma unroll 4
        for (int p = 0; p < kPW; ++p) {
          const int pt = wave * kPW + p;
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const int s0 = __builtin_amdgcn_readfirstlane(si.x), s1 = __builtin_amdgcn_readfirstlane(si.y),
                    s2 = __builtin_amdgcn_readfirstlane(si.z);
          const float4 h = mix3(row4<OVF>(s_rows, Gs, s0, lane, a.RS), row4<OVF>(s_rows, Gs, s1, lane, a.RS),
                                row4<OVF>(s_rows, Gs, s2, lane, a.RS), sw.x, sw.y, sw.z);
          if (MODE == 0) {
            const float f = si.w ? 1.f : 0.f;  // padding points: h = 0 anyway (zero weights)
            A1.x += h.x; A1.y += h.y; A1.z += h.z; A1.w += h.w;
            A2.x = fmaf(h.x, h.x, A2.x); A2.y = fmaf(h.y, h.y, A2.y); A2.z = fmaf(h.z, h.z, A2.z); A2.w = fmaf(h.w, h.w, A2.w);
            (void)f;
          } else {
            // xhat = (h - mean) rstd; y = xhat gamma + beta; dz = dlogit w_fc [y > 0]   (dlogit = 0 on padding points)
            const float dl = sw.w;
#define DH3D_IB_SUMS(X)                                                                         \
  {                                                                                             \
    const float xh = (h.X - q0.X) * q1.X;                                                       \
    const float y = fmaf(xh, q2.X, q3.X);                                                       \
    const float dz = y > 0.f ? dl * wf.X : 0.f;                                                 \
    A1.X += dz; A2.X = fmaf(dz, xh, A2.X); A3.X = fmaf(dl, fmaxf(y, 0.f), A3.X);               \
  }
            DH3D_IB_SUMS(x) DH3D_IB_SUMS(y) DH3D_IB_SUMS(z) DH3D_IB_SUMS(w)
#undef DH3D_IB_SUMS
          }
        }
      };
#if !defined(DH3D_IB_EXP) || !(DH3D_IB_EXP & 32)   // exp 32: no per-point work (results wrong)
      if (overflow) points(std::true_type{}); else points(std::false_type{});
#endif
      // the four waves hold different points of the same channels: through LDS (the row buffer is dead by now), then
      // one f64 atomic per channel
      __syncthreads();
      float *red = s_rows;
      *reinterpret_cast<float4 *>(red + (0 * 4 + wave) * 256 + lane * 4) = A1;
      *reinterpret_cast<float4 *>(red + (1 * 4 + wave) * 256 + lane * 4) = A2;
      if (MODE == 1) *reinterpret_cast<float4 *>(red + (2 * 4 + wave) * 256 + lane * 4) = A3;
      __syncthreads();
      {
        const size_t ch = (size_t)bi * (a.NS * 256) + sl * 256 + tid;  // this cloud's partial row
#pragma unroll
        for (int k = 0; k < (MODE == 1 ? 3 : 2); ++k) {
          const double v = ((double)red[(k * 4 + 0) * 256 + tid] + red[(k * 4 + 1) * 256 + tid]) +
                           ((double)red[(k * 4 + 2) * 256 + tid] + red[(k * 4 + 3) * 256 + tid]);
#if defined(DH3D_IB_EXP) && (DH3D_IB_EXP & 8)   // timing experiment: no statistics atomics (results wrong)
          if (v == 123.456) a.s0[0] = v;
#else
          unsafeAtomicAdd((k == 0 ? a.s0 : k == 1 ? a.s1 : a.s2) + ch, v);
#endif
        }
      }
      __syncthreads();  // rows and partial sums are overwritten by the next slice
    } else {
      // ---- MODE 2: dh per point into LDS 32 points at a time, scattered onto the staged rows by an MFMA product
      float *s_dh = s_x;
      float *dGs = a.dG + (size_t)sl * a.SS + (size_t)bi * m * a.RS;  // (MODE 3: NS = 1, RS = 256)
      const int nrt = nd > 32 ? 2 : 1;  // 32-slot row tiles in use (block-uniform)
      f32x16 acc[2][2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
      // MODE 3: the gradient rows of the NEXT chunk's points are requested while this chunk is scattered (their L2
      // round trip sat in front of every chunk before: 8 per block)
      f32x16 nx = {};
      auto request_dy = [&](int q) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < kCH / 4; ++j) {
          const int o1 = __builtin_amdgcn_readfirstlane(s_slot[(q * kCH + wave * (kCH / 4) + j) * 4 + 3]);
          // (padding points read row 0 of the cloud and are zeroed below: no load under a branch)
          const float4 v = *reinterpret_cast<const float4 *>(a.dY + ((size_t)bi * n + (o1 ? o1 - 1 : 0)) * 256 + lane * 4);
          nx[4 * j] = o1 ? v.x : 0.f; nx[4 * j + 1] = o1 ? v.y : 0.f; nx[4 * j + 2] = o1 ? v.z : 0.f; nx[4 * j + 3] = o1 ? v.w : 0.f;
        }
      };
      if (MODE == 3) request_dy(0);
      for (int q = 0; q < kP / kCH; ++q) {
        const f32x16 cur = nx;
        if (MODE == 3 && q + 1 < kP / kCH) request_dy(q + 1);
        auto points = [&](auto ovf) __attribute__((always_inline)) {
          constexpr bool OVF = decltype(ovf)::value;
#pragma unroll
          for (int j = 0; j < kCH / 4; ++j) {
            const int pl = wave * (kCH / 4) + j, pt = q * kCH + pl;
            const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
            const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
            const int s0 = __builtin_amdgcn_readfirstlane(si.x), s1 = __builtin_amdgcn_readfirstlane(si.y),
                      s2 = __builtin_amdgcn_readfirstlane(si.z);
            float4 dh;
            if (MODE == 3) {
              dh = make_float4(cur[4 * j], cur[4 * j + 1], cur[4 * j + 2], cur[4 * j + 3]);
            } else {
            const float4 h = mix3(row4<OVF>(s_rows, Gs, s0, lane, a.RS), row4<OVF>(s_rows, Gs, s1, lane, a.RS),
                                  row4<OVF>(s_rows, Gs, s2, lane, a.RS), sw.x, sw.y, sw.z);
            const float dl = sw.w, live = si.w ? 1.f : 0.f;
            if (MODE == 4) {  // dl = q[n] (0 on padding points)
              dh = make_float4(-dl * h.x, -dl * h.y, -dl * h.z, -dl * h.w);
            } else {
            // dh = k1 dz - k2 - k3 h,  dz = dlogit w_fc [h scale + shift > 0]   (0 on padding points)
#define DH3D_IB_DH(X)                                                                           \
  {                                                                                             \
    const float dz = fmaf(h.X, q0.X, q1.X) > 0.f ? dl * wf.X : 0.f;                             \
    dh.X = live * ((fmaf(q0.X, dz, -q2.X)) - q3.X * h.X);                                       \
  }
            DH3D_IB_DH(x) DH3D_IB_DH(y) DH3D_IB_DH(z) DH3D_IB_DH(w)
#undef DH3D_IB_DH
            }
            }
            *reinterpret_cast<float4 *>(s_dh + pl * kLDH + lane * 4) = dh;
            if (OVF) {  // rows that did not fit the staging area: straight to memory
              const int st[3] = {s0, s1, s2};
              const float wt[3] = {sw.x, sw.y, sw.z};
#pragma unroll
              for (int t = 0; t < 3; ++t)
                if (st[t] < 0) {
                  float *dst = dGs + (size_t)(-1 - st[t]) * a.RS + lane * 4;
                  unsafeAtomicAdd(dst, wt[t] * dh.x); unsafeAtomicAdd(dst + 1, wt[t] * dh.y);
                  unsafeAtomicAdd(dst + 2, wt[t] * dh.z); unsafeAtomicAdd(dst + 3, wt[t] * dh.w);
                }
            }
          }
        };
        if (overflow) points(std::true_type{}); else points(std::false_type{});
        __syncthreads();
        // dG_tile[slot, c] += sum_p S[p, slot] dh[p, c]: wave w owns channels 64 w .. 64 w + 63 (two 32-column tiles)
        const int kk = lane >> 5, col = wave * 64 + (lane & 31);
#pragma unroll 4
        for (int st = 0; st < kCH / 2; ++st) {
          const int pl = 2 * st + kk, pt = q * kCH + pl;
          const int4 si = *reinterpret_cast<const int4 *>(s_slot + pt * 4);
          const float4 sw = *reinterpret_cast<const float4 *>(s_w + pt * 4);
          const float b0 = s_dh[pl * kLDH + col], b1 = s_dh[pl * kLDH + col + 32];
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            if (rt < nrt) {
              const int jrow = rt * 32 + (lane & 31);
              const float sv = (si.x == jrow ? sw.x : 0.f) + (si.y == jrow ? sw.y : 0.f) + (si.z == jrow ? sw.z : 0.f);
#if defined(DH3D_IB_EXP) && (DH3D_IB_EXP & 1)   // timing experiment: no MFMAs (results wrong)
              asm volatile("" :: "v"(sv), "v"(b0), "v"(b1));
#else
              acc[rt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, b0, acc[rt][0], 0, 0, 0);
              acc[rt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv, b1, acc[rt][1], 0, 0, 0);
#endif
            }
          }
        }
        __syncthreads();  // the chunk buffer is rewritten
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        if (rt < nrt) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int j = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // 32x32 accumulator layout
#if defined(DH3D_IB_EXP) && (DH3D_IB_EXP & 2)   // timing experiment: plain stores instead of atomics (results wrong)
              if (j < nd) dGs[(size_t)s_row[j] * a.RS + wave * 64 + ct * 32 + (lane & 31)] = acc[rt][ct][r];
#elif defined(DH3D_IB_EXP) && (DH3D_IB_EXP & 4)   // timing experiment: no flush at all
              if (j < nd && acc[rt][ct][r] == 123.456f) dGs[0] = 1.f;
#else
              if (j < nd) unsafeAtomicAdd(dGs + (size_t)s_row[j] * a.RS + wave * 64 + ct * 32 + (lane & 31), acc[rt][ct][r]);
#endif
            }
        }
      }
      // (the next slice's staging writes s_rows only after every wave passed the last barrier above)
    }
  }
}

size_t interp_bn_lds(int mode) {
  const size_t rows = mode == 3 ? 0 : (mode == 2 || mode == 4 ? kCap2 : kCap);
  const size_t base = sizeof(float) * (rows * 256 + kP * 4 * 2 + 32 + 33 + kCap + 3);
  return base + sizeof(float) * (mode >= 2 ? (size_t)kCH * kLDH : 0);  // 70 / 79.6 KB: two workgroups per CU; 22 KB
}

template <int MODE>
int launch(const InterpBnArgs &a, hipStream_t s) {
  auto kern = interp_bn_kernel<MODE>;
  DH3D_ALLOW_BIG_LDS(kern);
  const int per_xcd = dh3d_cdiv(a.B, 8) * a.nblk;
  hipLaunchKernelGGL(kern, dim3(8 * per_xcd), dim3(256), interp_bn_lds(MODE), s, a);
  return dh3d_launch_status();
}

bool shape_ok(int Hd, int m) { return Hd % 256 == 0 && Hd >= 256 && Hd <= 1024 && m <= 1024; }

}  // namespace

// G: the 256-column slices [Hd/256][B*m][256] of coarse @ W + b; idx / dist: three_nn of the fine points [B,n,3];
// order: dh3d_spatial_sort records of the fine cloud [B,n,4] (may be NULL: points in index order, correct but slower);
// mask [B] bytes (may be NULL).  part [2][B][Hd] f64 (zeroed by the CALLER): per-CLOUD partial sums / sums of squares -- their
// sums over B are the column statistics (704 workgroups adding into one row of 1024 doubles cost 27 of 97 us in L2
// atomics on the same addresses; per cloud it is 32 workgroups per address).
DH3D_API int dh3d_interp_bn_colstats(const float *G, int Hd, int row_major, const int32_t *idx, const float *dist,
                                     const float *order, int B, int n, int m, const unsigned char *mask, double *part,
                                     void *stream) {
  DH3D_REQUIRE(G && idx && dist && part && B > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(shape_ok(Hd, m));
  hipStream_t s = (hipStream_t)stream;
  InterpBnArgs a{};
  a.G = G; a.NS = Hd / 256; a.Rc = (long long)B * m; a.idx = idx; a.dist = dist;
  a.SS = row_major ? 256 : a.Rc * 256; a.RS = row_major ? Hd : 256;
  a.order = reinterpret_cast<const float4 *>(order); a.B = B; a.n = n; a.m = m; a.nblk = dh3d_cdiv(n, kP); a.mask = mask;
  a.s0 = part; a.s1 = part + (size_t)B * Hd;
  return launch<0>(a, s);
}

// part [3][B][Hd] f64 (zeroed by the CALLER): per-cloud partials of S1, S2, S3 -- the sums of dh3d_bn_bwd_sums for the rank-one
// gradient dy = dlogit x w_fc on the virtual rows h = interp(G);  dlogit [B*n] by original point index.
DH3D_API int dh3d_interp_bn_bwd_sums(const float *G, int Hd, int row_major, const int32_t *idx, const float *dist,
                                     const float *order, int B, int n, int m, const unsigned char *mask, const float *dlogit,
                                     const float *w_fc, const float *mean, const float *rstd, const float *gamma,
                                     const float *beta, double *part, void *stream) {
  DH3D_REQUIRE(G && idx && dist && dlogit && w_fc && mean && rstd && gamma && beta && part);
  DH3D_REQUIRE(B > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(shape_ok(Hd, m));
  hipStream_t s = (hipStream_t)stream;
  InterpBnArgs a{};
  a.G = G; a.NS = Hd / 256; a.Rc = (long long)B * m; a.idx = idx; a.dist = dist;
  a.SS = row_major ? 256 : a.Rc * 256; a.RS = row_major ? Hd : 256;
  a.order = reinterpret_cast<const float4 *>(order); a.B = B; a.n = n; a.m = m; a.nblk = dh3d_cdiv(n, kP); a.mask = mask;
  a.dlogit = dlogit; a.wfc = w_fc; a.v0 = mean; a.v1 = rstd; a.v2 = gamma; a.v3 = beta;
  a.s0 = part; a.s1 = part + (size_t)B * Hd; a.s2 = part + 2 * (size_t)B * Hd;
  return launch<1>(a, s);
}

// dG [Hd/256][B*m][256] (zeroed by the CALLER) = interp^T(dh),  dh = scale dz - k2 - k3 h  (the coefficients of
// dh3d_bn_bwd_finalize; dz = dlogit w_fc [h scale + shift > 0]); f32 atomics.
DH3D_API int dh3d_interp_bn_bwd_apply(const float *G, int Hd, int row_major, const int32_t *idx, const float *dist,
                                      const float *order, int B, int n, int m, const unsigned char *mask, const float *dlogit,
                                      const float *w_fc, const float *scale, const float *shift, const float *k2,
                                      const float *k3, float *dG, void *stream) {
  DH3D_REQUIRE(G && idx && dist && dlogit && w_fc && scale && shift && k2 && k3 && dG && B > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(shape_ok(Hd, m));
  hipStream_t s = (hipStream_t)stream;
  InterpBnArgs a{};
  a.G = G; a.NS = Hd / 256; a.Rc = (long long)B * m; a.idx = idx; a.dist = dist;
  a.SS = row_major ? 256 : a.Rc * 256; a.RS = row_major ? Hd : 256;
  a.order = reinterpret_cast<const float4 *>(order); a.B = B; a.n = n; a.m = m; a.nblk = dh3d_cdiv(n, kP); a.mask = mask;
  a.dlogit = dlogit; a.wfc = w_fc; a.v0 = scale; a.v1 = shift; a.v2 = k2; a.v3 = k3; a.dG = dG;
  return launch<2>(a, s);
}

// Backward of three_interpolate on the Morton order: grad_points [b, m, c] (zeroed here) += weight * grad_out rows,
// c == 256.  Same result as dh3d_three_interpolate_bwd up to the summation order (f32 atomics on <= 56 staged rows per
// 128-point block instead of one per (point, neighbour, channel): 69 M -> 8 M at 22 x 4096 points).
DH3D_API int dh3d_three_interpolate_bwd_sorted(int b, int n, int c, int m, const float *grad_out, const int32_t *idx,
                                               const float *weight, const float *order, float *grad_points,
                                               void *stream) {
  DH3D_REQUIRE(grad_out && idx && weight && grad_points && b > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(c == 256 && m <= 1024);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * m * c, s) != hipSuccess) return DH3D_ERR_LAUNCH;
  InterpBnArgs a{};
  a.NS = 1; a.Rc = (long long)b * m; a.SS = 0; a.RS = 256; a.idx = idx; a.weight = weight; a.dY = grad_out;
  a.order = reinterpret_cast<const float4 *>(order); a.B = b; a.n = n; a.m = m; a.nblk = dh3d_cdiv(n, kP);
  a.dG = grad_points;
  return launch<3>(a, s);
}

// dc [B*m, 256] += interp^T(-q x),  x = three_interpolate(c) rebuilt in the walk, q [B*n] by original point index: the
// l2-normalisation term of NetVLAD's commuted backward (netvlad_train.hip).  dc is NOT zeroed: it holds the GEMM terms.
DH3D_API int dh3d_interp_scatter_scaled(const float *c, const float *q, const int32_t *idx, const float *dist,
                                        const float *order, int B, int n, int m, const unsigned char *mask, float *dc,
                                        void *stream) {
  DH3D_REQUIRE(c && q && idx && dist && dc && B > 0 && n > 0 && m > 0);
  DH3D_SUPPORTED(m <= 1024);
  InterpBnArgs a{};
  a.G = c; a.NS = 1; a.Rc = (long long)B * m; a.SS = a.Rc * 256; a.RS = 256; a.idx = idx; a.dist = dist;
  a.order = reinterpret_cast<const float4 *>(order); a.B = B; a.n = n; a.m = m; a.nblk = dh3d_cdiv(n, kP); a.mask = mask;
  a.dlogit = q; a.dG = dc;
  return launch<4>(a, (hipStream_t)stream);
}
