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
#include <type_traits>

#include "bf16x3.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
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
  static constexpr int A_ELEMS = 3 * kTM * LD;       // bf16 elements per S buffer
  static constexpr int P_FLOATS = 4 * 8 * 64;        // floats per partial buffer: four consumer waves x half an accumulator tile
  static constexpr int B3_VEC = 2 * KB * 64;         // uint4 per third weight plane (two column blocks x KB fragments)
  static constexpr size_t LDS_BYTES = (size_t)2 * A_ELEMS * 2 + (size_t)2 * P_FLOATS * 4 + (size_t)B3_VEC * 16;
  static_assert(DOUT == 64 && (VEC == 2 || VEC == 4) && KBH % 2 == 0 && LDS_BYTES <= 159 * 1024, "shape");
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

// two f32 -> three dwords of two bf16 each.  The subtractions are SCALAR f32 instructions on purpose: on gfx950 a
// packed-f32 VALU instruction (v_pk_add_f32 / v_pk_fma_f32) occupies the matrix pipe -- it serialises with every
// wave's MFMAs on the SIMD -- while scalar f32 VALU work runs beside them (tools/coissue_probe2.hip, profiles/r05_a_*)
__device__ __forceinline__ void split3x2(const f32x2 v, unsigned &c1, unsigned &c2, unsigned &c3) {
  const float r0 = v[0] - hi16_of(v[0]), r1 = v[1] - hi16_of(v[1]);
  const float t0 = r0 - hi16_of(r0), t1 = r1 - hi16_of(r1);
  c1 = pack_hi16(v[0], v[1]);
  c2 = pack_hi16(r0, r1);
  c3 = pack_hi16(t0, t1);
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
#ifndef DH3D_X6_PRIO
#define DH3D_X6_PRIO 1
#endif
    if (DH3D_X6_PRIO & 1) __builtin_amdgcn_s_setprio(3);
    const int prow = tid >> 4, lj = tid & 15;
    const int c0 = lj * C::VEC;  // first channel of this lane
    const unsigned mrec = (unsigned)(0x100000000ULL / N);
    const int G = cnt * C::ROUNDS;  // rounds of this workgroup
    int nid[2][8], myid[2][2];
    FV fv[2][8];
    float qd[2][2], pcd[2];
    // neighbour offsets, one dword per lane (see compute): lane l of a point's 16 lanes is (slot a = (l >> 2) & 3,
    // component i = l & 3) and holds [-, dx, dy, dz][i] of neighbours a and a + 4
    const int da = (tid >> 2) & 3, di = tid & 3;
    const int dcomp = di > 0 ? di - 1 : 0;   // (lanes with i = 0 are never read)

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
      myid[s][0] = nbr[(size_t)n * 8 + da];
      myid[s][1] = nbr[(size_t)n * 8 + 4 + da];
    };
    auto issue_feat = [&](int s, int g) {
      unsigned n; bool ok;
      row_of(g, n, ok);
      unsigned q = __umulhi(n, mrec);  // floor(n / N) or one less
      if (n - q * N >= N) ++q;
      const unsigned cloud0 = q * N;
      pcd[s] = xyz[(size_t)n * 3 + dcomp];
      qd[s][0] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xyz) + (size_t)((cloud0 + (unsigned)myid[s][0]) * 12u + (unsigned)dcomp * 4u));
      qd[s][1] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xyz) + (size_t)((cloud0 + (unsigned)myid[s][1]) * 12u + (unsigned)dcomp * 4u));
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
      // The K-neighbour reduce  S_c[ch] = sum_k [1, dx, dy, dz]_k[c] * f_k[ch]  in SCALAR f32 VALU instructions.  Measured
      // on gfx950 (tools/coissue_probe2.hip, profiles/r05_*_coissue_probe2.txt): scalar-f32 and integer VALU work of
      // one wave issues freely while another wave's MFMAs run; a packed-f32 instruction (v_pk_fma_f32) occupies the
      // matrix pipe; and ANY matrix-pipe instruction of a second wave (packed f32, or this reduce as
      // v_mfma_f32_4x4x1_16b_f32: 4x fewer issue slots) starves behind a wave that streams MFMAs back to back -- the
      // two waves' times add.  So the producers stay off the matrix pipe entirely.
      // Lane (a, i) of a point's 16 lanes holds component i of [1, dx, dy, dz] for neighbours a (d0) and a + 4 (d1);
      // component c of neighbour k is broadcast along the row from lane 4 (k & 3) + c.
      float d0 = qd[s][0] - pcd[s], d1 = qd[s][1] - pcd[s];
      // (a DPP read of a VGPR needs two wait states after the VALU write; the compiler cannot see into the asm below)
      asm volatile("s_nop 1" : "+v"(d0), "+v"(d1));
      f32x2 acc[4][C::VEC / 2];  // [S0, Sx, Sy, Sz][channel pair]
      // acc += bcast(row lane L of d) * f as ONE instruction: v_fmac_f32 with the DPP row broadcast on its first
      // source (the compiler keeps a separate v_mov_b32_dpp per broadcast value: 24 more VALU issue slots per round)
#define DH3D_X6_FMAC_BCAST(ACC, D, F, L) \
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(D), "v"(F))
#define DH3D_X6_MUL_BCAST(ACC, D, F, L) \
  asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "=v"(ACC) : "v"(D), "v"(F))
#define DH3D_X6_NEIGHBOUR_(KI, DS, LX, LY, LZ)                                                                   \
  {                                                                                                              \
    const float *fp = reinterpret_cast<const float *>(&fv[s][KI]);                                               \
    _Pragma("unroll") for (int h = 0; h < C::VEC / 2; ++h) {                                                     \
      _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                            \
        const float f = fp[2 * h + e];                                                                           \
        if ((KI) == 0) {  /* (fma(d, f, 0) == d * f exactly) */                                                  \
          acc[0][h][e] = f;                                                                                      \
          DH3D_X6_MUL_BCAST(acc[1][h][e], DS, f, LX);                                                            \
          DH3D_X6_MUL_BCAST(acc[2][h][e], DS, f, LY);                                                            \
          DH3D_X6_MUL_BCAST(acc[3][h][e], DS, f, LZ);                                                            \
        } else {                                                                                                 \
          acc[0][h][e] += f;                                                                                     \
          DH3D_X6_FMAC_BCAST(acc[1][h][e], DS, f, LX);                                                           \
          DH3D_X6_FMAC_BCAST(acc[2][h][e], DS, f, LY);                                                           \
          DH3D_X6_FMAC_BCAST(acc[3][h][e], DS, f, LZ);                                                           \
        }                                                                                                        \
      }                                                                                                          \
    }                                                                                                            \
  }
#define DH3D_X6_NEIGHBOUR(KI)                                                                                    \
  if (((KI) & 3) == 0) DH3D_X6_NEIGHBOUR_(KI, ((KI) < 4 ? d0 : d1), 1, 2, 3)                                      \
  else if (((KI) & 3) == 1) DH3D_X6_NEIGHBOUR_(KI, ((KI) < 4 ? d0 : d1), 5, 6, 7)                                 \
  else if (((KI) & 3) == 2) DH3D_X6_NEIGHBOUR_(KI, ((KI) < 4 ? d0 : d1), 9, 10, 11)                               \
  else DH3D_X6_NEIGHBOUR_(KI, ((KI) < 4 ? d0 : d1), 13, 14, 15)
      DH3D_X6_NEIGHBOUR(0) DH3D_X6_NEIGHBOUR(1) DH3D_X6_NEIGHBOUR(2) DH3D_X6_NEIGHBOUR(3)
      DH3D_X6_NEIGHBOUR(4) DH3D_X6_NEIGHBOUR(5) DH3D_X6_NEIGHBOUR(6) DH3D_X6_NEIGHBOUR(7)
#undef DH3D_X6_NEIGHBOUR
#undef DH3D_X6_NEIGHBOUR_
#undef DH3D_X6_FMAC_BCAST
#undef DH3D_X6_MUL_BCAST
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

    // Two schedules, one producer of each per SIMD.  Every producer does the same work per round -- reduce + split (VALU and matrix
    // pipe), then the gathers of a later round (address arithmetic, then the CU's one texture-address path) -- and the
    // workgroup barrier keeps the eight of them in step: with one schedule they all want the VALU in the first half of
    // a tile period and the load path in the second.  Odd waves issue their loads FIRST (for the next round, into the
    // slot the previous round freed) and reduce afterwards, so the two resources are used side by side.
    // (Two separate loops, not one loop with a branch: a load under a branch is merged through register copies.)
    if ((wave >> 2) & 1) {  // (waves w and w + 4 share SIMD w % 4: one of each schedule per SIMD)
      issue_ids(0, 0);
      issue_ids(1, 1);
      issue_feat(0, 0);
      issue_ids(0, 2);
      for (int g = 0; g < G; g += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int gg = g + s;
          issue_feat(s ^ 1, gg + 1);
          issue_ids(s ^ 1, gg + 3);
#if !(defined(DH3D_X6_EXP) && (DH3D_X6_EXP & 4))
          compute(s, gg);
#else
          asm volatile("" :: "v"(reinterpret_cast<const float *>(&fv[s][0])[0]), "v"(reinterpret_cast<const float *>(&fv[s][7])[1]),
                       "v"(qd[s][0]), "v"(pcd[s]));
#endif
          if (gg % C::ROUNDS == C::ROUNDS - 1 && gg < G) wg_barrier(gg / C::ROUNDS);
        }
      }
    } else {
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
                       "v"(qd[s][0]), "v"(pcd[s]));
#endif
          XPROBE(0, gg, 2);
          issue_feat(s, gg + 2);
          issue_ids(s, gg + 4);
          XPROBE(0, gg, 3);
          if (gg % C::ROUNDS == C::ROUNDS - 1 && gg < G) wg_barrier(gg / C::ROUNDS);  // tile gg / ROUNDS is staged
          XPROBE(0, gg, 4);
        }
      }
    }
    wg_barrier(cnt);  // the consumers' last partial tiles
  } else {
    // ------------------------------------------------------------------ consumers
    if (DH3D_X6_PRIO & 2) __builtin_amdgcn_s_setprio(3);
    const int cw = wave - kProducers / 64, lane = tid & 63;
    const int cb = cw & 1, kh = cw >> 1;
    // Weights: the two large bf16 planes of this wave's (K-half, column block) quarter live in registers for the
    // lifetime of the workgroup; the third plane (used by one product in six) is parked in LDS in fragment order and
    // read back per k-block beside the A fragments -- with all three in registers (96 + 32 accumulator + A fragments)
    // a 168-register wave has no room to double-buffer the A fragments, and an A fragment requested just before its
    // MFMA costs an LDS round trip per k-block.
    uint4 breg[C::KBH][2];
    uint4 *const s_b3 = reinterpret_cast<uint4 *>(s_raw + (size_t)2 * C::A_ELEMS * 2 + (size_t)2 * C::P_FLOATS * 4) +
                        (size_t)(cb * C::KB + kh * C::KBH) * 64 + lane;
#pragma unroll
    for (int kb = 0; kb < C::KBH; ++kb) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
        breg[kb][p] = wp3[((size_t)(cb * C::KB + kh * C::KBH + kb) * 3 + p) * 64 + lane];
      s_b3[(size_t)kb * 64] = wp3[((size_t)(cb * C::KB + kh * C::KBH + kb) * 3 + 2) * 64 + lane];  // (read back by this wave only)
    }
    // epilogue operands of this lane's output column (the pre-bias folded into the shift)
    const int col = cb * 32 + (lane & 31);
    float sc = 1.f, sh = 0.f;
    if (ep.scale) sc = ep.scale[col];
    if (ep.shift) sh = ep.shift[col];
    if (ep.pre_bias) sh = fmaf(ep.pre_bias[col], sc, sh);
    const int lo = ep.act == DH3D_ACT_RELU ? 0 : INT_MIN;
    // The two K-half waves of a column block each keep HALF of the summed 32x32 tile (kh 0: accumulator registers
    // 0-7 = rows 0-15, kh 1: registers 8-15 = rows 16-31) and hand the other half to the partner through LDS in the
    // accumulator's own layout (two 16-byte writes per lane, no transposition, no bank conflicts).  The partner's
    // half of tile i-1 is read back, added, put through the epilogue and stored in the gaps of tile i's MFMAs:
    // accumulator register r of lane l is output (row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31), so one
    // 4-byte store per register writes two full 128-byte lines.
    float4 *const s_P4 = reinterpret_cast<float4 *>(s_P);
    float4 *const p_mine = s_P4 + (size_t)(cw * 2) * 64 + lane;           // what this wave writes (+ parity * 512, + j * 64)
    const float4 *const p_partner = s_P4 + (size_t)((cw ^ 2) * 2) * 64 + lane;
    const unsigned voff = (unsigned)(((4 * (lane >> 5) + 16 * kh) * DOUT + col) * 4);  // byte offset of (row 0 of this half, col)
    float keep[8];
    const unsigned short *const abase0 = s_A + (size_t)(lane & 31) * C::LD + 8 * (lane >> 5) + kh * C::KBH * 16;

    // One tile: GEMM of this wave's K-half x column block, six bf16 products per k-block on two independent
    // accumulator chains (small terms first).  The A fragments of k-block kb + 1 are requested BEFORE the MFMAs of
    // k-block kb (an LDS round trip is ~100+ cycles, a k-block's MFMAs are 192), and every k-block is a scheduling
    // region of its own, so the order below is the order issued.  PREV: the epilogue of tile i - 1 rides along, 8 / KBH values per k-block.
    auto tile = [&](int i, auto prev_tag) {
      constexpr bool PREV = decltype(prev_tag)::value;
      const unsigned short *abase = abase0 + (size_t)(i & 1) * C::A_ELEMS;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
      bf16x8 a[2][3];
      uint4 b3r[2];
      float4 pp[2];
#pragma unroll
      for (int p = 0; p < 3; ++p) a[0][p] = *reinterpret_cast<const bf16x8 *>(abase + p * kTM * C::LD);
      b3r[0] = s_b3[0];
      if (PREV) {
        pp[0] = p_partner[(size_t)((i - 1) & 1) * 512];
        pp[1] = p_partner[(size_t)((i - 1) & 1) * 512 + 64];
      }
      float *const orow = reinterpret_cast<float *>(reinterpret_cast<char *>(out + (size_t)(tbeg + (i - 1) * S) * kTM * DOUT) + voff);
      const unsigned grow_prev = (unsigned)(tbeg + (i - 1) * S) * kTM + 4 * (lane >> 5) + 16 * kh;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < C::KBH; ++kb) {
        if (kb + 1 < C::KBH) {
#pragma unroll
          for (int p = 0; p < 3; ++p)
            a[(kb + 1) & 1][p] = *reinterpret_cast<const bf16x8 *>(abase + p * kTM * C::LD + (kb + 1) * 16);
          b3r[(kb + 1) & 1] = s_b3[(size_t)(kb + 1) * 64];
        }
#if !((defined(DH3D_X6_PROBE) && DH3D_X6_PROBE == 4) || (defined(DH3D_X6_EXP) && (DH3D_X6_EXP & 8)))  // probe 4 / exp 8: no MFMAs (timing experiment)
        const bf16x8 a1 = a[kb & 1][0], a2 = a[kb & 1][1], a3 = a[kb & 1][2];
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, breg[kb][0]), b2 = __builtin_bit_cast(bf16x8, breg[kb][1]),
                     b3 = __builtin_bit_cast(bf16x8, b3r[kb & 1]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc1, 0, 0, 0);
#endif
#if !(defined(DH3D_X6_EXP) && (DH3D_X6_EXP & 2))   // exp 2: no partial exchange / epilogue / store (timing experiment, results wrong)
        if (PREV) {
#pragma unroll
          for (int jj = 0; jj < 8 / C::KBH; ++jj) {
            const int j = (8 / C::KBH) * kb + jj;
            const float part = reinterpret_cast<const float *>(&pp[j >> 2])[j & 3];
            const int vi = max(__float_as_int(fmaf(keep[j] + part, sc, sh)), lo);
            const int rowoff = (j & 3) + 8 * (j >> 2);
            if (!RAGGED || grow_prev + rowoff < R) reinterpret_cast<int *>(orow)[rowoff * DOUT] = vi;
          }
        }
#endif
        // the order above within the k-block: loads first, then MFMAs with one independent instruction behind each
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int m = 0; m < 6; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          if (m % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#if !(defined(DH3D_X6_EXP) && (DH3D_X6_EXP & 2))
      // K-half sum of the two chains; own half stays in registers, the other half goes to the partner
      float sum[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sum[r] = acc0[r] + acc1[r];
#pragma unroll
      for (int j = 0; j < 8; ++j) keep[j] = kh ? sum[8 + j] : sum[j];
      p_mine[(size_t)(i & 1) * 512] = kh ? make_float4(sum[0], sum[1], sum[2], sum[3]) : make_float4(sum[8], sum[9], sum[10], sum[11]);
      p_mine[(size_t)(i & 1) * 512 + 64] = kh ? make_float4(sum[4], sum[5], sum[6], sum[7]) : make_float4(sum[12], sum[13], sum[14], sum[15]);
#else
      asm volatile("" :: "v"(acc0), "v"(acc1));
#endif
    };

    wg_barrier(0);  // tile 0 staged
    XPROBE(1, 0, 0);
    tile(0, std::false_type{});
    XPROBE(1, 0, 1);
    wg_barrier(1);  // halves of tile 0 exchanged, tile 1 staged
    XPROBE(1, 0, 2);
    for (int i = 1; i < cnt; ++i) {
      XPROBE(1, i, 0);
      tile(i, std::true_type{});
      XPROBE(1, i, 1);
      wg_barrier(i + 1);  // halves of tile i exchanged, tile i+1 staged
      XPROBE(1, i, 2);
    }
#if !(defined(DH3D_X6_EXP) && (DH3D_X6_EXP & 2))
    {  // the last tile's epilogue
      const int i = cnt;
      const float4 pp0 = p_partner[(size_t)((i - 1) & 1) * 512], pp1 = p_partner[(size_t)((i - 1) & 1) * 512 + 64];
      float *const orow = reinterpret_cast<float *>(reinterpret_cast<char *>(out + (size_t)(tbeg + (i - 1) * S) * kTM * DOUT) + voff);
      const unsigned grow_prev = (unsigned)(tbeg + (i - 1) * S) * kTM + 4 * (lane >> 5) + 16 * kh;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float part = j < 4 ? reinterpret_cast<const float *>(&pp0)[j & 3] : reinterpret_cast<const float *>(&pp1)[j & 3];
        const int vi = max(__float_as_int(fmaf(keep[j] + part, sc, sh)), lo);
        const int rowoff = (j & 3) + 8 * (j >> 2);
        if (!RAGGED || grow_prev + rowoff < R) reinterpret_cast<int *>(orow)[rowoff * DOUT] = vi;
      }
    }
#endif
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
