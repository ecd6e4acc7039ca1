// flex_conv for the full-resolution layers (32->64 and 64->64 at N points, K = 8), as a persistent,
// wave-specialised pipeline on the bf16 matrix pipe with f32 accuracy ("bf16x6", bf16x3.h).
//
// Same factorisation as flex_pm.hip:  out = [S0|Sx|Sy|Sz] @ [bias; theta_x; theta_y; theta_z].  At these
// shapes the exact-f32 MFMA form needs 15.4 us of matrix pipe per launch (B=8, N=8192, 64->64) against 4.6 us
// of compulsory HBM time, and its gather / GEMM / store phases run back to back inside a workgroup.  Here:
//   * one 768-thread workgroup per CU (three waves per SIMD: two producers + one consumer), looping over 32-point tiles of ONE XCD's contiguous tile range;
//   * waves 0-7 PRODUCE (16 lanes per point): neighbour ids two rounds ahead, neighbour rows one round ahead
//     (the next tile's loads are in flight while this one is reduced), the K-neighbour reduce in f32, the
//     exact 3-way bf16 split, and the S tile into LDS as three bf16 planes (double buffered);
//   * waves 8-11 CONSUME: wave (cb, kh) owns K-half kh x column block cb of the concatenated weight, held in
//     REGISTERS for the lifetime of the workgroup (no weight traffic in the steady state), multiplies with
//     v_mfma_f32_32x32x16_bf16 (6 per K=16, two independent accumulator chains), and the two K-half partial
//     tiles are summed through LDS (double buffered) with feature_bias + BatchNorm + activation applied in
//     the 16-byte store.
// One workgroup barrier per tile; results are deterministic.
//
// What bounds it (measured, tools/coissue_probe.hip): on gfx950 an FP32 VALU instruction does NOT issue while
// another wave's MFMA occupies the same SIMD -- FP VALU time and MFMA time add -- whereas integer VALU, LDS
// and memory instructions do overlap with it.  So the design minimises FP instructions, not just total
// instructions: neighbour offsets are computed ONCE per (point, neighbour) by one lane and broadcast with DPP
// moves (not 16x redundantly), the bf16 split uses packed subtracts, and everything else the producers do
// (addresses, masks, byte permutes, LDS writes, loads) runs in the shadow of the consumers' MFMAs.
#include "bf16x3.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
constexpr int kTM = 32;   // points per tile
constexpr int kPPR = 32;  // points per producer round (16 lanes per point, 8 producer waves)
constexpr int kProducers = 8 * 64, kThreads = kProducers + 256;

template <int DIN, int DOUT>
struct X6Cfg {
  static constexpr int KD = 4 * DIN;                 // GEMM depth
  static constexpr int LD = KD + 8;                  // LDS leading dimension of a bf16 plane (elements)
  static constexpr int KB = KD / 16;                 // k-blocks of 16
  static constexpr int KBH = KB / 2;                 // k-blocks per consumer wave (one K-half)
  static constexpr int VEC = DIN / 16;               // channels per producer lane
  static constexpr int ROUNDS = kTM / kPPR;          // rounds per tile
  static constexpr int PLD = DOUT + 4;               // leading dimension of a partial tile (floats)
  static constexpr int A_ELEMS = 3 * kTM * LD;       // bf16 elements per S buffer
  static constexpr int P_FLOATS = 2 * kTM * PLD;     // floats per partial buffer (two K-halves)
  static constexpr int CV = DOUT / 4;                // float4 per output row
  static constexpr int RP = 256 / CV;                // rows per reduce pass
  static constexpr size_t LDS_BYTES = (size_t)2 * A_ELEMS * 2 + (size_t)2 * P_FLOATS * 4;
  static_assert(DOUT == 64 && (VEC == 2 || VEC == 4) && KBH % 2 == 0 && kTM % RP == 0, "shape");
};

#ifdef DH3D_X6_PROBE  // dev instrumentation (tools/x6_probe.py): cycle stamps of one workgroup's two roles
__device__ long long g_x6probe[2][64][8];
#define XPROBE(role, it, k)                                                                     \
  do {                                                                                          \
    if (blockIdx.x == 8 && (threadIdx.x == 0 || threadIdx.x == kProducers) && (it) < 64) g_x6probe[role][it][k] = clock64(); \
  } while (0)
#else
#define XPROBE(role, it, k) do { } while (0)
#endif

// LDS traffic of this wave done, then workgroup barrier.  Global loads stay in flight across it.
#if defined(DH3D_X6_PROBE) && DH3D_X6_PROBE == 5  // probe 5: arrival / departure time of every wave at every barrier
__device__ long long g_x6bar[12][16][2];
__device__ __forceinline__ void wg_barrier(int k = 0) {
  const int w = threadIdx.x >> 6;
  const bool rec = blockIdx.x == 8 && (threadIdx.x & 63) == 0 && k < 16;
  if (rec) g_x6bar[w][k][0] = clock64();
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (rec) g_x6bar[w][k][1] = clock64();
}
#else
__device__ __forceinline__ void wg_barrier(int = 0) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// value of lane K of the caller's 16-lane row (v_mov_b32_dpp row_newbcast: an integer-class VALU op)
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x150 + K, 0xf, 0xf, false));
}

__device__ __forceinline__ float hi16_of(float a) { return __uint_as_float(__float_as_uint(a) & 0xFFFF0000u); }

// two f32 -> three dwords of two bf16 each; the subtractions are packed (one FP instruction per pair)
__device__ __forceinline__ void split3x2(const f32x2 v, unsigned &c1, unsigned &c2, unsigned &c3) {
  const f32x2 a = {hi16_of(v[0]), hi16_of(v[1])};
  const f32x2 r = v - a;
  const f32x2 b = {hi16_of(r[0]), hi16_of(r[1])};
  const f32x2 t = r - b;
  c1 = pack_hi16(v[0], v[1]);
  c2 = pack_hi16(r[0], r[1]);
  c3 = pack_hi16(t[0], t[1]);
}

template <int VEC> struct LaneVec;
template <> struct LaneVec<4> { typedef float4 type; };
template <> struct LaneVec<2> { typedef float2 type; };

// RAGGED: the last tile is partial (R % 32 != 0): stores are guarded, which costs the reduce its place in the
// MFMA shadow (a guarded store is a branch); full-tile launches use the unguarded instantiation.
template <int DIN, int DOUT, bool RAGGED>
__global__ __launch_bounds__(kThreads) void flex_conv_x6_kernel(
    const float *__restrict__ feat, const float *__restrict__ xyz, const int32_t *__restrict__ nbr,
    const uint4 *__restrict__ wp3, unsigned R, unsigned N, EpilogueArgs ep, float *__restrict__ out, int T) {
  using C = X6Cfg<DIN, DOUT>;
  using FV = typename LaneVec<C::VEC>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  unsigned short *s_A = reinterpret_cast<unsigned short *>(s_raw);           // [2][3][kTM][LD] bf16
  float *s_P = reinterpret_cast<float *>(s_raw + (size_t)2 * C::A_ELEMS * 2);  // [2][2][kTM][PLD] f32
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this workgroup's tiles: XCD x = block % 8 owns the contiguous range [x*Tx, (x+1)*Tx)
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
  const int Tx = (T + 7) >> 3;
  const int tbeg = x * Tx + slot;
  const int tend = (x + 1) * Tx < T ? (x + 1) * Tx : T;
  const int cnt = tbeg < tend ? (tend - tbeg + S - 1) / S : 0;
  if (cnt == 0) return;

  if (wave < kProducers / 64) {
    // ------------------------------------------------------------------ producers
    // FP32 VALU work only issues in the gaps of the consumers' MFMA stream: with the producers ahead in the issue
    // arbitration those gaps open as soon as an FP instruction is ready instead of at the end of a GEMM burst
    // (measured 26.9 -> 25.9 us per launch)
    __builtin_amdgcn_s_setprio(3);
    const int prow = tid >> 4, lj = tid & 15;
    const int c0 = lj * C::VEC;  // first channel of this lane
    const unsigned mrec = (unsigned)(0x100000000ULL / N);
    const int G = cnt * C::ROUNDS;  // rounds of this workgroup
    int nid[2][8], myid[2];
    FV fv[2][8];
    f32x3 qv[2], pc[2];  // whole 12-byte vectors: a loop-carried load result must stay one register tuple

    // rounds at or past G are dummies (row 0 again, into the idle buffer): the pipeline below then has no
    // conditional loads -- a load under a branch is merged through register copies that wait for it
    auto row_of = [&](int g, unsigned &n, bool &ok) {
      const int i = g / C::ROUNDS, r = g % C::ROUNDS;
      n = (unsigned)(tbeg + i * S) * kTM + r * kPPR + prow;
      ok = (int)(g < G) & (int)(n < R);
      n = ok ? n : 0u;
    };
    auto issue_ids = [&](int s, int g) {
      unsigned n; bool ok;
      row_of(g, n, ok);
      const int4 *ip = reinterpret_cast<const int4 *>(nbr + (size_t)n * 8);
      const int4 a = ip[0], b = ip[1];
      nid[s][0] = a.x; nid[s][1] = a.y; nid[s][2] = a.z; nid[s][3] = a.w;
      nid[s][4] = b.x; nid[s][5] = b.y; nid[s][6] = b.z; nid[s][7] = b.w;
      myid[s] = nbr[(size_t)n * 8 + (lj & 7)];  // lane j (and j+8) looks after neighbour j's coordinates
    };
    auto issue_feat = [&](int s, int g) {
      unsigned n; bool ok;
      row_of(g, n, ok);
      unsigned q = __umulhi(n, mrec);  // floor(n / N) or one less
      if (n - q * N >= N) ++q;
      const unsigned cloud0 = q * N;
      __builtin_memcpy(&pc[s], xyz + (size_t)n * 3, 12);
      __builtin_memcpy(&qv[s], reinterpret_cast<const char *>(xyz) + (size_t)((cloud0 + (unsigned)myid[s]) * 12u), 12);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned g2 = cloud0 + (unsigned)nid[s][k];
        fv[s][k] = *reinterpret_cast<const FV *>(reinterpret_cast<const char *>(feat) +
                                                 (size_t)(g2 * (unsigned)(DIN * 4) + (unsigned)(c0 * 4)));
      }
    };
    auto compute = [&](int s, int g) {
      // (rows past R and dummy rounds read point 0: finite values in S rows whose outputs are never stored)
      const int i = g / C::ROUNDS, r = g % C::ROUNDS;
      // neighbour offsets: once per (point, neighbour), then broadcast along the point's 16 lanes
      const float rx = qv[s][0] - pc[s][0], ry = qv[s][1] - pc[s][1], rz = qv[s][2] - pc[s][2];
      f32x2 acc[4][C::VEC / 2];  // [S0, Sx, Sy, Sz][channel pair]
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int h = 0; h < C::VEC / 2; ++h) acc[c][h] = f32x2{0.f, 0.f};
#define X6_ACC(a0, a1, a2, a3, f, dx2, dy2, dz2)                                                                \
  a0 += f;                                                                                                      \
  a1 = __builtin_elementwise_fma(dx2, f, a1);                                                                   \
  a2 = __builtin_elementwise_fma(dy2, f, a2);                                                                   \
  a3 = __builtin_elementwise_fma(dz2, f, a3);
#define DH3D_X6_NEIGHBOUR(KI)                                                                                    \
  {                                                                                                              \
    const float dx = row_bcast<KI>(rx), dy = row_bcast<KI>(ry), dz = row_bcast<KI>(rz);                          \
    const f32x2 dx2 = {dx, dx}, dy2 = {dy, dy}, dz2 = {dz, dz};                                                  \
    const float *fp = reinterpret_cast<const float *>(&fv[s][KI]);                                               \
    _Pragma("unroll") for (int h = 0; h < C::VEC / 2; ++h) {                                                     \
      const f32x2 f = {fp[2 * h], fp[2 * h + 1]};                                                                \
      X6_ACC(acc[0][h], acc[1][h], acc[2][h], acc[3][h], f, dx2, dy2, dz2)                                       \
    }                                                                                                            \
  }
      DH3D_X6_NEIGHBOUR(0) DH3D_X6_NEIGHBOUR(1) DH3D_X6_NEIGHBOUR(2) DH3D_X6_NEIGHBOUR(3)
      DH3D_X6_NEIGHBOUR(4) DH3D_X6_NEIGHBOUR(5) DH3D_X6_NEIGHBOUR(6) DH3D_X6_NEIGHBOUR(7)
#undef DH3D_X6_NEIGHBOUR
#undef X6_ACC
      unsigned short *row = s_A + (size_t)(i & 1) * C::A_ELEMS + (size_t)(r * kPPR + prow) * C::LD + c0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        unsigned c1[C::VEC / 2], c2[C::VEC / 2], c3[C::VEC / 2];
#pragma unroll
        for (int h = 0; h < C::VEC / 2; ++h) {
          split3x2(acc[c][h], c1[h], c2[h], c3[h]);
        }
        if (C::VEC == 4) {
          *reinterpret_cast<uint2 *>(row + c * DIN) = make_uint2(c1[0], c1[C::VEC / 2 - 1]);
          *reinterpret_cast<uint2 *>(row + c * DIN + kTM * C::LD) = make_uint2(c2[0], c2[C::VEC / 2 - 1]);
          *reinterpret_cast<uint2 *>(row + c * DIN + 2 * kTM * C::LD) = make_uint2(c3[0], c3[C::VEC / 2 - 1]);
        } else {
          *reinterpret_cast<unsigned *>(row + c * DIN) = c1[0];
          *reinterpret_cast<unsigned *>(row + c * DIN + kTM * C::LD) = c2[0];
          *reinterpret_cast<unsigned *>(row + c * DIN + 2 * kTM * C::LD) = c3[0];
        }
      }
    };

    issue_ids(0, 0);
    issue_ids(1, 1);
    issue_feat(0, 0);
    issue_ids(0, 2);
    issue_feat(1, 1);
    issue_ids(1, 3);
    for (int g = 0; g < G; g += 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int gg = g + s;
        XPROBE(0, gg, 0);
#if defined(DH3D_X6_PROBE) && DH3D_X6_PROBE == 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        XPROBE(0, gg, 1);
#if (!defined(DH3D_X6_PROBE) || DH3D_X6_PROBE != 3) && !(defined(DH3D_X6_EXP) && (DH3D_X6_EXP & 4))   // probe 3 / exp 4: producers only load (timing experiment)
        compute(s, gg);
#else
        asm volatile("" :: "v"(reinterpret_cast<const float *>(&fv[s][0])[0]), "v"(reinterpret_cast<const float *>(&fv[s][7])[1]),
                     "v"(qv[s][0]), "v"(pc[s][2]));
#endif
        XPROBE(0, gg, 2);
        issue_feat(s, gg + 2);
        issue_ids(s, gg + 4);
        XPROBE(0, gg, 3);
        if (gg % C::ROUNDS == C::ROUNDS - 1 && gg < G) wg_barrier(gg / C::ROUNDS);  // tile gg / ROUNDS is staged
        XPROBE(0, gg, 4);
      }
    }
    wg_barrier(cnt);  // the consumers' last partial tiles
  } else {
    // ------------------------------------------------------------------ consumers
    const int cw = wave - kProducers / 64, lane = tid & 63, ctid = tid - kProducers;
    const int cb = cw & 1, kh = cw >> 1;
    uint4 breg[C::KBH][3];
#pragma unroll
    for (int kb = 0; kb < C::KBH; ++kb)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        breg[kb][p] = wp3[((size_t)(cb * C::KB + kh * C::KBH + kb) * 3 + p) * 64 + lane];
    const int c4 = (ctid % C::CV) * 4, prow = ctid / C::CV;
    float4 pb = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = pb;
    if (ep.pre_bias) pb = *reinterpret_cast<const float4 *>(ep.pre_bias + c4);
    if (ep.scale) sc = *reinterpret_cast<const float4 *>(ep.scale + c4);
    if (ep.shift) sh = *reinterpret_cast<const float4 *>(ep.shift + c4);
    sh.x = fmaf(pb.x, sc.x, sh.x); sh.y = fmaf(pb.y, sc.y, sh.y); sh.z = fmaf(pb.z, sc.z, sh.z); sh.w = fmaf(pb.w, sc.w, sh.w);
    const int lo = ep.act == DH3D_ACT_RELU ? 0 : INT_MIN;
    // GEMM of tile i: this wave's K-half x column block.  The six products of a k-block go to two independent
    // accumulator chains (a dependent MFMA would wait ~12 cycles on its predecessor), small terms first.  The
    // trailing scheduling hints pin the issue order: chains alternating, the A fragments of the next k-block
    // and one slot for an independent VALU instruction / store (the caller's reduce) behind every MFMA.
    auto gemm = [&](int i, f32x16 &acc0, f32x16 &acc1) {
      const unsigned short *abase = s_A + (size_t)(i & 1) * C::A_ELEMS + (size_t)(lane & 31) * C::LD +
                                    8 * (lane >> 5) + kh * C::KBH * 16;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
#if (defined(DH3D_X6_PROBE) && DH3D_X6_PROBE == 4) || (defined(DH3D_X6_EXP) && (DH3D_X6_EXP & 8))     // probe 4 / exp 8: consumers skip the MFMAs (timing experiment)
      for (int kb = 0; kb < 0; ++kb) {
#else
      for (int kb = 0; kb < C::KBH; ++kb) {
#endif
#if defined(DH3D_X6_EXP) && (DH3D_X6_EXP & 1)   // timing experiment: one set of A fragments per tile (results wrong)
        const int kbo = 0;
#else
        const int kbo = kb * 16;
#endif
        const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(abase + kbo);
        const bf16x8 a2 = *reinterpret_cast<const bf16x8 *>(abase + kTM * C::LD + kbo);
        const bf16x8 a3 = *reinterpret_cast<const bf16x8 *>(abase + 2 * kTM * C::LD + kbo);
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, breg[kb][0]), b2 = __builtin_bit_cast(bf16x8, breg[kb][1]),
                     b3 = __builtin_bit_cast(bf16x8, breg[kb][2]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);  // first A fragments (+ the reduce's partial tiles)
#pragma unroll
      for (int m = 0; m < 6 * C::KBH; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        if (m % 6 == 1 && m + 6 < 6 * C::KBH) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
        if (m % 6 == 3) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      }
    };
    auto write_partial = [&](int i, const f32x16 &acc0, const f32x16 &acc1) {
      float *part = s_P + (size_t)(i & 1) * C::P_FLOATS + (size_t)kh * kTM * C::PLD + cb * 32 + (lane & 31);
      // packed adds: an FP32 VALU instruction holds up the SIMD's matrix pipe (see above), so half as many of them
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 sum = f32x2{acc0[r], acc0[r + 1]} + f32x2{acc1[r], acc1[r + 1]};
        part[(size_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * C::PLD] = sum[0];
        part[(size_t)(((r + 1) & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * C::PLD] = sum[1];
      }
    };
    // sum of the two K-half partial tiles of tile i, epilogue, 16-byte stores
    auto reduce_store = [&](int i) {
      const unsigned grow0 = (unsigned)(tbeg + i * S) * kTM;
#pragma unroll
      for (int ps = 0; ps < kTM / C::RP; ++ps) {
        const int p = ps * C::RP + prow;
        const float *q = s_P + (size_t)(i & 1) * C::P_FLOATS + (size_t)p * C::PLD + c4;
        const float4 v0 = *reinterpret_cast<const float4 *>(q);
        const float4 v1 = *reinterpret_cast<const float4 *>(q + kTM * C::PLD);
        // four packed FP instructions per row segment (the pre-bias is folded into the shift); relu / no activation
        // as an INTEGER max on the bit patterns (max(bits, 0) clamps negative floats to +0, max(bits, INT_MIN) is the
        // identity): integer VALU work does overlap the matrix pipe
        const f32x2 t0 = f32x2{v0.x, v0.y} + f32x2{v1.x, v1.y}, t1 = f32x2{v0.z, v0.w} + f32x2{v1.z, v1.w};
        const f32x2 r0 = __builtin_elementwise_fma(t0, f32x2{sc.x, sc.y}, f32x2{sh.x, sh.y});
        const f32x2 r1 = __builtin_elementwise_fma(t1, f32x2{sc.z, sc.w}, f32x2{sh.z, sh.w});
        int4 vi;
        vi.x = max(__float_as_int(r0[0]), lo); vi.y = max(__float_as_int(r0[1]), lo);
        vi.z = max(__float_as_int(r1[0]), lo); vi.w = max(__float_as_int(r1[1]), lo);
        if (!RAGGED || grow0 + p < R) *reinterpret_cast<int4 *>(out + (size_t)(grow0 + p) * DOUT + c4) = vi;
      }
    };
    // The reduce + store of tile i-1 is issued in the shadow of tile i's MFMAs (an MFMA holds the matrix pipe
    // for 32 cycles; the wave issues in order, so independent instructions between two MFMAs are free).
    f32x16 acc0, acc1;
    wg_barrier(0);  // tile 0 staged
    XPROBE(1, 0, 0);
    gemm(0, acc0, acc1);
    write_partial(0, acc0, acc1);
    XPROBE(1, 0, 1);
    wg_barrier(1);  // partials of tile 0 complete, tile 1 staged
    XPROBE(1, 0, 2);
    for (int i = 1; i < cnt; ++i) {
      XPROBE(1, i, 0);
#if defined(DH3D_X6_EXP) && (DH3D_X6_EXP & 2)   // timing experiment: no partial tiles, no reduce + store (results wrong)
      gemm(i, acc0, acc1);
      asm volatile("" :: "v"(acc0), "v"(acc1));
#else
      reduce_store(i - 1);
      gemm(i, acc0, acc1);
      write_partial(i, acc0, acc1);
#endif
      XPROBE(1, i, 1);
      wg_barrier(i + 1);  // partials of tile i complete, tile i+1 staged
      XPROBE(1, i, 2);
    }
    reduce_store(cnt - 1);
  }
}

// [bias; theta_x; theta_y; theta_z] ([4*Din, Dout]) -> the bf16x3 fragment layout of dh3d_pack_weight_x3
__global__ __launch_bounds__(256) void pack_flex_weight_x3_kernel(const float *__restrict__ theta,
                                                                 const float *__restrict__ bias, int Din, int Dout,
                                                                 unsigned short *__restrict__ packed) {
  const int Kd = 4 * Din, KB = Kd / 16;
  const long long total = (long long)Kd * Dout;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int j = (int)(e & 7);
    const int lane = (int)((e >> 3) & 63);
    const long long blk = e >> 9;  // nb*KB + kb
    const int kb = (int)(blk % KB), nb = (int)(blk / KB);
    const int k = kb * 16 + 8 * (lane >> 5) + j, col = nb * 32 + (lane & 31);
    const int comp = k / Din, i = k % Din;
    const float w = comp == 0 ? bias[(size_t)i * Dout + col] : theta[((size_t)(comp - 1) * Din + i) * Dout + col];
    unsigned c1, c2, c3;
    split3(w, c1, c2, c3);
    const size_t base = ((size_t)blk * 3) * 512 + (size_t)lane * 8 + j;
    packed[base] = (unsigned short)c1;
    packed[base + 512] = (unsigned short)c2;
    packed[base + 1024] = (unsigned short)c3;
  }
}

int persistent_grid() {
  static int g = 0;
  if (g == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
    g = cus >= 8 ? (cus & ~7) : 8;
  }
  return g;
}

template <int DIN, int DOUT>
int flex_conv_x6_launch(const float *feat, const float *xyz, const int32_t *nbr, const void *wp3, int B, int N,
                        const EpilogueArgs &ep, float *out, hipStream_t s, int reserve_per_xcd) {
  using C = X6Cfg<DIN, DOUT>;
  const long long R = (long long)B * N;
  const int T = dh3d_cdiv(R, kTM);
  // One workgroup per CU (136 KB of LDS each).  A CU that another stream's kernel holds with a large LDS allocation
  // (the farthest-point sampling keeps ~100 KB for the whole local step) cannot take one of these workgroups until it
  // is released, and the statically split tiles of that workgroup would then run as a second wave after all the
  // others (measured: 26 us alone, 43-48 us beside the FPS).  The caller states how many CUs per XCD are taken and
  // the launch leaves them out, so that every workgroup is resident from the start.
  int grid = persistent_grid() - 8 * (reserve_per_xcd > 0 ? reserve_per_xcd : 0);
  grid = grid < 8 ? 8 : grid;
  if (R % kTM == 0) {
    auto kern = flex_conv_x6_kernel<DIN, DOUT, false>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), C::LDS_BYTES, s, feat, xyz, nbr,
                       static_cast<const uint4 *>(wp3), (unsigned)R, (unsigned)N, ep, out, T);
  } else {
    auto kern = flex_conv_x6_kernel<DIN, DOUT, true>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), C::LDS_BYTES, s, feat, xyz, nbr,
                       static_cast<const uint4 *>(wp3), (unsigned)R, (unsigned)N, ep, out, T);
  }
  return dh3d_launch_status();
}

}  // namespace

#if defined(DH3D_X6_PROBE) && DH3D_X6_PROBE == 5
DH3D_API int dh3d_x6_bar_read(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x6bar), sizeof(long long) * n) == hipSuccess ? 0 : 3;
}
#endif
#ifdef DH3D_X6_PROBE
DH3D_API int dh3d_x6_probe_read(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x6probe), sizeof(long long) * n) == hipSuccess ? 0 : 3;
}
#endif

DH3D_API int dh3d_pack_flex_weight_x3(const float *theta, const float *bias, int Din, int Dout, void *packed,
                                      void *stream) {
  DH3D_REQUIRE(theta && bias && packed && Din > 0 && Dout > 0);
  DH3D_SUPPORTED(Din % 4 == 0 && Dout % 32 == 0);
  hipLaunchKernelGGL(pack_flex_weight_x3_kernel, dim3(dh3d_cdiv((long long)4 * Din * Dout, 256)), dim3(256), 0,
                     (hipStream_t)stream, theta, bias, Din, Dout, static_cast<unsigned short *>(packed));
  return dh3d_launch_status();
}

DH3D_API int dh3d_flex_conv_pm_x6_fwd_r(const float *features, const float *xyz, const int32_t *nbr,
                                        const void *wpacked_x3, int B, int N, int K, int Din, int Dout,
                                        const dh3d_epilogue *ep, int reserve_cus_per_xcd, float *out, void *stream) {
  DH3D_REQUIRE(features && xyz && nbr && wpacked_x3 && out && B > 0 && N > 0 && K > 0);
  DH3D_REQUIRE(reserve_cus_per_xcd >= 0 && reserve_cus_per_xcd < 32);
  const long long R = (long long)B * N;
  // 32-bit byte offsets into the feature map
  DH3D_SUPPORTED((!ep || ep->act != DH3D_ACT_SIGMOID) && K == 8 && N >= 2 && R * Din * 4 < (1LL << 32) && R + kTM < (1LL << 31));
  const EpilogueArgs e = dh3d_ep(ep);
  hipStream_t s = (hipStream_t)stream;
  if (Din == 32 && Dout == 64)
    return flex_conv_x6_launch<32, 64>(features, xyz, nbr, wpacked_x3, B, N, e, out, s, reserve_cus_per_xcd);
  if (Din == 64 && Dout == 64)
    return flex_conv_x6_launch<64, 64>(features, xyz, nbr, wpacked_x3, B, N, e, out, s, reserve_cus_per_xcd);
  return DH3D_ERR_UNSUPPORTED;
}

DH3D_API int dh3d_flex_conv_pm_x6_fwd(const float *features, const float *xyz, const int32_t *nbr,
                                      const void *wpacked_x3, int B, int N, int K, int Din, int Dout,
                                      const dh3d_epilogue *ep, float *out, void *stream) {
  return dh3d_flex_conv_pm_x6_fwd_r(features, xyz, nbr, wpacked_x3, B, N, K, Din, Dout, ep, 0, out, stream);
}
