// flex_conv for the full-resolution layers (32->64 and 64->64 at N points, K = 8): persistent, ONE wave per SIMD that
// does everything -- gathers, K-neighbour reduce, bf16x3 split, GEMM, epilogue -- in a single software-pipelined
// instruction stream ("x7"; the wave-specialised predecessor is flex_x6.hip).
//
// Why one wave per SIMD.  Measured on gfx950 (tools/coissue_probe2.hip, profiles/r05_f_coissue_probe2_full.txt):
//   * a SIMD issues one VALU-class instruction (VALU or MFMA) per ~4.7 cycles, all its waves together;
//   * behind a v_mfma_f32_32x32x16_bf16 (32 matrix-pipe cycles) the SAME wave issues five more instructions for free;
//   * a wave that sends anything to the matrix pipe (MFMA, or packed-f32 VALU) while ANOTHER wave streams MFMAs is
//     served one instruction per MFMA of that stream and blocks in order meanwhile: the two waves' times add.
// The factorised flex_conv  out = [S0|Sx|Sy|Sz] @ [bias; theta]  has two matrix-shaped parts -- the reduce
// S_c[ch] = sum_k [1,dx,dy,dz]_k[c] f_k[ch]  (v_mfma_f32_4x4x1_16b_f32: 256 lane-FMAs per issue slot, exact f32) and the
// GEMM (bf16x6: six v_mfma_f32_32x32x16_bf16 per K = 16, f32-accurate) -- and both must therefore come from the same
// wave.  Per 32-point tile and SIMD: 48 + 64 matrix instructions (1536 + ~400 pipe cycles) and ~330 other
// VALU-class instructions, interleaved so that the split / address / epilogue work sits in the MFMAs' shadows.
//
// Workgroup = 4 waves = the four (K-half kh, column block cb) quarters of the concatenated weight, held in REGISTERS for
// the lifetime of the workgroup (one wave per SIMD: 512 registers each).  Per tile t every wave
//   * GEMMs its quarter of tile t from the S planes in LDS (double buffered), two accumulator chains;
//   * produces 8 of the 32 S rows of tile t+1 (two rounds of 4 points, 16 lanes per point): reduce on the matrix pipe,
//     exact 3-way bf16 split, three bf16 planes into the other LDS buffer;
//   * has the neighbour rows of tile t+2 and the neighbour ids of tile t+3 in flight;
//   * exchanges K-half sums with its partner in the accumulator's own layout and stores tile t-1 (flex_x6.hip).
// One workgroup barrier per tile.  Results are bit-identical to flex_x6.hip's (same fma chains, same product order).
#include <type_traits>

#include "bf16x3.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kTM = 32;        // points per tile
constexpr int kThreads = 256;  // four waves: one per SIMD

template <int DIN, int DOUT>
struct X7Cfg {
  static constexpr int KD = 4 * DIN;                 // GEMM depth
  static constexpr int LD = KD + 8;                  // LDS leading dimension of a bf16 plane (elements)
  static constexpr int KB = KD / 16;                 // k-blocks of 16
  static constexpr int KBH = KB / 2;                 // k-blocks per wave (one K-half)
  static constexpr int VEC = DIN / 16;               // channels per lane in the produce part
  static constexpr int A_ELEMS = 3 * kTM * LD;       // bf16 elements per S buffer
  static constexpr int P_FLOATS = 4 * 8 * 64;        // floats per exchange buffer: four waves x half an accumulator tile
  static constexpr size_t LDS_BYTES = (size_t)2 * A_ELEMS * 2 + (size_t)2 * P_FLOATS * 4;
  static_assert(DOUT == 64 && (VEC == 2 || VEC == 4) && (KBH == 8 || KBH == 4), "shape");
};

__device__ __forceinline__ void wg_barrier7() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float hi16_of7(float a) { return __uint_as_float(__float_as_uint(a) & 0xFFFF0000u); }

template <int VEC> struct LaneVec7;
template <> struct LaneVec7<4> { typedef float4 type; };
template <> struct LaneVec7<2> { typedef float2 type; };

template <int P> using par_t = std::integral_constant<int, P>;

#ifdef DH3D_X7_PROBE  // dev instrumentation (tools/x7_check.py): cycle stamps of one wave's stages
__device__ long long g_x7probe[16][12];
#define X7STAMP(it, k)                                                                                      \
  do {                                                                                                      \
    if (blockIdx.x == 8 && threadIdx.x == 0 && (it) >= 0 && (it) < 16) g_x7probe[it][k] = clock64();      \
  } while (0)
#else
#define X7STAMP(it, k) do { } while (0)
#endif

template <int DIN, int DOUT, bool RAGGED>
__global__ __launch_bounds__(kThreads, 1) void flex_conv_x7_kernel(
    const float *__restrict__ feat, const float *__restrict__ xyz, const int32_t *__restrict__ nbr,
    const uint4 *__restrict__ wp3, unsigned R, unsigned N, EpilogueArgs ep, float *__restrict__ out, int T) {
  using C = X7Cfg<DIN, DOUT>;
  using FV = typename LaneVec7<C::VEC>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
  unsigned short *s_A = reinterpret_cast<unsigned short *>(s_raw);             // [2][3][kTM][LD] bf16
  float4 *const s_P4 = reinterpret_cast<float4 *>(s_raw + (size_t)2 * C::A_ELEMS * 2);  // [2][4 waves][2][64] float4
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this workgroup's tiles: XCD x = block % 8 owns the contiguous range [x*Tx, (x+1)*Tx)
  const int x = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
  const int Tx = (T + 7) >> 3;
  const int tbeg = x * Tx + slot;
  const int tend = (x + 1) * Tx < T ? (x + 1) * Tx : T;
  const int cnt = tbeg < tend ? (tend - tbeg + S - 1) / S : 0;
  if (cnt == 0) return;

  // ------------------------------------------------------------------ GEMM side: this wave's weight quarter
  const int cb = wave & 1, kh = wave >> 1;
  uint4 breg[C::KBH][3];
#pragma unroll
  for (int kb = 0; kb < C::KBH; ++kb)
#pragma unroll
    for (int p = 0; p < 3; ++p) breg[kb][p] = wp3[((size_t)(cb * C::KB + kh * C::KBH + kb) * 3 + p) * 64 + lane];
  const int col = cb * 32 + (lane & 31);
  float sc = 1.f, sh = 0.f;
  if (ep.scale) sc = ep.scale[col];
  if (ep.shift) sh = ep.shift[col];
  if (ep.pre_bias) sh = fmaf(ep.pre_bias[col], sc, sh);
  const int lo = ep.act == DH3D_ACT_RELU ? 0 : INT_MIN;
  float4 *const p_mine = s_P4 + (size_t)(wave * 2) * 64 + lane;             // (+ parity * 512, + j * 64)
  const float4 *const p_partner = s_P4 + (size_t)((wave ^ 2) * 2) * 64 + lane;
  const unsigned voff = (unsigned)(((4 * (lane >> 5) + 16 * kh) * DOUT + col) * 4);  // byte offset of (row 0 of this half, col)
  float keep[8];
  const unsigned short *const abase0 = s_A + (size_t)(lane & 31) * C::LD + 8 * (lane >> 5) + kh * C::KBH * 16;

  // ------------------------------------------------------------------ produce side: 8 rows of every tile
  // round r (0, 1) of a tile: point 8 * wave + 4 * r + (lane >> 4), channels c0 .. c0 + VEC - 1 of it
  const int lj = lane & 15, c0 = lj * C::VEC;
  const int da = (lane >> 2) & 3, di = lane & 3;   // offset vectors: (slot a, component i) of the point's 16 lanes
  const int dcomp = di > 0 ? di - 1 : 0;
  const bool done = di == 0;                       // component 0 of [1, dx, dy, dz]
  const unsigned mrec = (unsigned)(0x100000000ULL / N);
  int nid[2][2][8], myid[2][2][2];                 // [tile parity][round][...]
  FV fv[2][2][8];
  float qd[2][2][2], pcd[2][2];

  // tiles at or past cnt are dummies (the last tile again): no conditional loads anywhere in the pipeline
  auto row_of = [&](int i, int r, unsigned &n) {
    // (min, not a test: tiles past the last one re-read it, rows past R re-read row R - 1 -- a branch here would cut the
    //  k-block's scheduling region in two)
    n = min((unsigned)(tbeg + min(i, cnt - 1) * S) * kTM + 8 * wave + 4 * r + (lane >> 4), R - 1u);
  };
  auto issue_ids = [&](auto par, int i) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      unsigned n;
      row_of(i, r, n);
      const int4 *ip = reinterpret_cast<const int4 *>(nbr + (size_t)n * 8);
      const int4 a = ip[0], b = ip[1];
      nid[P][r][0] = a.x; nid[P][r][1] = a.y; nid[P][r][2] = a.z; nid[P][r][3] = a.w;
      nid[P][r][4] = b.x; nid[P][r][5] = b.y; nid[P][r][6] = b.z; nid[P][r][7] = b.w;
      myid[P][r][0] = nbr[(size_t)n * 8 + da];
      myid[P][r][1] = nbr[(size_t)n * 8 + 4 + da];
    }
  };
  auto issue_feat = [&](auto par, int i, int r) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    unsigned n;
    row_of(i, r, n);
    unsigned q = __umulhi(n, mrec);  // floor(n / N) or one less
    if (n - q * N >= N) ++q;
    const unsigned cloud0 = q * N;
    pcd[P][r] = xyz[(size_t)n * 3 + dcomp];
    qd[P][r][0] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xyz) +
                                                   (size_t)((cloud0 + (unsigned)myid[P][r][0]) * 12u + (unsigned)dcomp * 4u));
    qd[P][r][1] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(xyz) +
                                                   (size_t)((cloud0 + (unsigned)myid[P][r][1]) * 12u + (unsigned)dcomp * 4u));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned g2 = cloud0 + (unsigned)nid[P][r][k];
      fv[P][r][k] = *reinterpret_cast<const FV *>(reinterpret_cast<const char *>(feat) +
                                                  (size_t)(g2 * (unsigned)(DIN * 4) + (unsigned)(c0 * 4)));
    }
  };
  // S accumulators of the tile being produced: [round][channel of the lane] x [S0, Sx, Sy, Sz]
  f32x4 sacc[2][C::VEC];
  // neighbours K0 .. K1-1 of round r: B = channel `comp` of the four lanes' feature vectors (the lanes of a block belong
  // to one point), A = the offset vector of neighbour k, taken from block (k & 3) of the point's four blocks by the
  // instruction's own broadcast (cbsz = 2, abid = k & 3); output VGPR c of accumulator `comp` = S_c of channel c0 + comp
  auto reduce_part = [&](auto par, auto rr, auto k0, auto k1) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value, r = decltype(rr)::value, K0 = decltype(k0)::value, K1 = decltype(k1)::value;
    const float d0 = done ? 1.f : qd[P][r][0] - pcd[P][r], d1 = done ? 1.f : qd[P][r][1] - pcd[P][r];
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      const float *fp = reinterpret_cast<const float *>(&fv[P][r][k]);
#pragma unroll
      for (int c = 0; c < C::VEC; ++c) {
        const f32x4 prev = k == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : sacc[r][c];
        if (k == 0) sacc[r][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(d0, fp[c], prev, 2, 0, 0);
        else if (k == 1) sacc[r][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(d0, fp[c], prev, 2, 1, 0);
        else if (k == 2) sacc[r][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(d0, fp[c], prev, 2, 2, 0);
        else if (k == 3) sacc[r][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(d0, fp[c], prev, 2, 3, 0);
        else if (k == 4) sacc[r][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(d1, fp[c], prev, 2, 0, 0);
        else if (k == 5) sacc[r][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(d1, fp[c], prev, 2, 1, 0);
        else if (k == 6) sacc[r][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(d1, fp[c], prev, 2, 2, 0);
        else sacc[r][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(d1, fp[c], prev, 2, 3, 0);
      }
    }
  };
  // component cc (S0 / Sx / Sy / Sz) of round r of tile i: exact 3-way bf16 split (scalar f32 subtractions: a packed-f32
  // instruction holds the matrix pipe), three planes into LDS buffer i & 1
  auto split_part = [&](int i, auto rr, auto ccv) __attribute__((always_inline)) {
    constexpr int r = decltype(rr)::value, cc = decltype(ccv)::value;
    unsigned short *row = s_A + (size_t)(i & 1) * C::A_ELEMS + (size_t)(8 * wave + 4 * r + (lane >> 4)) * C::LD + c0 + cc * DIN;
    unsigned c1[C::VEC / 2], c2[C::VEC / 2], c3[C::VEC / 2];
#pragma unroll
    for (int h = 0; h < C::VEC / 2; ++h) {
      const float v0 = sacc[r][2 * h][cc], v1 = sacc[r][2 * h + 1][cc];
      const float r0 = v0 - hi16_of7(v0), r1 = v1 - hi16_of7(v1);
      const float t0 = r0 - hi16_of7(r0), t1 = r1 - hi16_of7(r1);
      c1[h] = pack_hi16(v0, v1);
      c2[h] = pack_hi16(r0, r1);
      c3[h] = pack_hi16(t0, t1);
    }
    if (C::VEC == 4) {
      *reinterpret_cast<uint2 *>(row) = make_uint2(c1[0], c1[C::VEC / 2 - 1]);
      *reinterpret_cast<uint2 *>(row + kTM * C::LD) = make_uint2(c2[0], c2[C::VEC / 2 - 1]);
      *reinterpret_cast<uint2 *>(row + 2 * kTM * C::LD) = make_uint2(c3[0], c3[C::VEC / 2 - 1]);
    } else {
      *reinterpret_cast<unsigned *>(row) = c1[0];
      *reinterpret_cast<unsigned *>(row + kTM * C::LD) = c2[0];
      *reinterpret_cast<unsigned *>(row + 2 * kTM * C::LD) = c3[0];
    }
  };

  // ------------------------------------------------------------------ one pipeline stage
  // GEMM of tile i (if GEMM) beside the production of tile i + 1, the neighbour rows of tile i + 2 and the ids of tile
  // i + 3.  PI = parity of i: register buffers of tile i + 1 are [1 - PI], of tile i + 2 [PI], ids of tile i + 3 [1 - PI].
  // Every k-block is a scheduling region; the pieces of the other work are dealt over the k-blocks so that each region
  // holds about as many other VALU-class instructions as its MFMAs can shadow.
  auto stage = [&](int i, auto pi, auto gemm_tag, auto prev_tag) __attribute__((always_inline)) {
    constexpr int PI = decltype(pi)::value;
    constexpr bool GEMM = decltype(gemm_tag)::value, PREV = decltype(prev_tag)::value;
    using P1 = par_t<1 - PI>;
    using P2 = par_t<PI>;
    const unsigned short *abase = abase0 + (size_t)(i & 1) * C::A_ELEMS;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    bf16x8 a[2][3];
    float4 pp[2];
    if (GEMM) {
#pragma unroll
      for (int p = 0; p < 3; ++p) a[0][p] = *reinterpret_cast<const bf16x8 *>(abase + p * kTM * C::LD);
    }
    if (GEMM && PREV) {
      pp[0] = p_partner[(size_t)((i - 1) & 1) * 512];
      pp[1] = p_partner[(size_t)((i - 1) & 1) * 512 + 64];
    }
    const int ip = (GEMM && PREV) ? i - 1 : 0;
    float *const orow = reinterpret_cast<float *>(reinterpret_cast<char *>(out + (size_t)(tbeg + ip * S) * kTM * DOUT) + voff);
    const unsigned grow_prev = (unsigned)(tbeg + ip * S) * kTM + 4 * (lane >> 5) + 16 * kh;
    __builtin_amdgcn_sched_barrier(0);
    X7STAMP(i, 0);
#pragma unroll
    for (int kb = 0; kb < C::KBH; ++kb) {
      if (GEMM && kb + 1 < C::KBH) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[(kb + 1) & 1][p] = *reinterpret_cast<const bf16x8 *>(abase + p * kTM * C::LD + (kb + 1) * 16);
      }
      if (GEMM) {
        const bf16x8 a1 = a[kb & 1][0], a2 = a[kb & 1][1], a3 = a[kb & 1][2];
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, breg[kb][0]), b2 = __builtin_bit_cast(bf16x8, breg[kb][1]),
                     b3 = __builtin_bit_cast(bf16x8, breg[kb][2]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc1, 0, 0, 0);
      }
      // ---- the production of tile i + 1, dealt over the k-blocks
      if constexpr (C::KBH == 8) {
        if (kb == 0) reduce_part(P1{}, par_t<0>{}, par_t<0>{}, par_t<4>{});
        if (kb == 1) reduce_part(P1{}, par_t<0>{}, par_t<4>{}, par_t<8>{});
        if (kb == 2) reduce_part(P1{}, par_t<1>{}, par_t<0>{}, par_t<4>{});
        if (kb == 3) reduce_part(P1{}, par_t<1>{}, par_t<4>{}, par_t<8>{});
        if (kb == 2) split_part(i + 1, par_t<0>{}, par_t<0>{});
        if (kb == 3) split_part(i + 1, par_t<0>{}, par_t<1>{});
        if (kb == 4) { split_part(i + 1, par_t<0>{}, par_t<2>{}); split_part(i + 1, par_t<0>{}, par_t<3>{}); }
        if (kb == 5) { split_part(i + 1, par_t<1>{}, par_t<0>{}); split_part(i + 1, par_t<1>{}, par_t<1>{}); }
        if (kb == 6) { split_part(i + 1, par_t<1>{}, par_t<2>{}); split_part(i + 1, par_t<1>{}, par_t<3>{}); }
        // ---- loads: rows of tile i + 2 (ids landed a stage ago) as early as the stage allows -- they are wanted at the
        // start of the next stage and a gather under load takes most of a stage -- then the ids of tile i + 3
        if (kb == 0) issue_feat(P2{}, i + 2, 0);
        if (kb == 1) issue_feat(P2{}, i + 2, 1);
        if (kb == 4) issue_ids(P1{}, i + 3);
      } else {
        if (kb == 0) reduce_part(P1{}, par_t<0>{}, par_t<0>{}, par_t<8>{});
        if (kb == 1) reduce_part(P1{}, par_t<1>{}, par_t<0>{}, par_t<8>{});
        if (kb == 1) { split_part(i + 1, par_t<0>{}, par_t<0>{}); split_part(i + 1, par_t<0>{}, par_t<1>{}); }
        if (kb == 2) { split_part(i + 1, par_t<0>{}, par_t<2>{}); split_part(i + 1, par_t<0>{}, par_t<3>{});
                       split_part(i + 1, par_t<1>{}, par_t<0>{}); split_part(i + 1, par_t<1>{}, par_t<1>{}); }
        if (kb == 3) { split_part(i + 1, par_t<1>{}, par_t<2>{}); split_part(i + 1, par_t<1>{}, par_t<3>{}); }
        if (kb == 0) issue_feat(P2{}, i + 2, 0);
        if (kb == 1) issue_feat(P2{}, i + 2, 1);
        if (kb == 2) issue_ids(P1{}, i + 3);
      }
      // ---- the epilogue of tile i - 1: 8 / KBH values per k-block
      if (GEMM && PREV) {
#pragma unroll
        for (int jj = 0; jj < 8 / C::KBH; ++jj) {
          const int j = (8 / C::KBH) * kb + jj;
          const float part = reinterpret_cast<const float *>(&pp[j >> 2])[j & 3];
          const int vi = max(__float_as_int(fmaf(keep[j] + part, sc, sh)), lo);
          const int rowoff = (j & 3) + 8 * (j >> 2);
          if (!RAGGED || grow_prev + rowoff < R) reinterpret_cast<int *>(orow)[rowoff * DOUT] = vi;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      X7STAMP(i, 1 + kb);
    }
    if (GEMM) {
      // K-half sum of the two chains; own half stays in registers, the other half goes to the partner
      float sum[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sum[r] = acc0[r] + acc1[r];
#pragma unroll
      for (int j = 0; j < 8; ++j) keep[j] = kh ? sum[8 + j] : sum[j];
      p_mine[(size_t)(i & 1) * 512] = kh ? make_float4(sum[0], sum[1], sum[2], sum[3]) : make_float4(sum[8], sum[9], sum[10], sum[11]);
      p_mine[(size_t)(i & 1) * 512 + 64] = kh ? make_float4(sum[4], sum[5], sum[6], sum[7]) : make_float4(sum[12], sum[13], sum[14], sum[15]);
    }
    X7STAMP(i, 9);
    wg_barrier7();  // S rows of tile i + 1 staged, halves of tile i exchanged, buffer i & 1 free for tile i + 2
    X7STAMP(i, 10);
  };

  // ------------------------------------------------------------------ the pipeline
  // prologue: ids of tiles 0, 1, 2; rows of tiles 0, 1; tile 0 produced by stage -1 (no GEMM)
  issue_ids(par_t<0>{}, 0);
  issue_ids(par_t<1>{}, 1);
  issue_feat(par_t<0>{}, 0, 0);
  issue_feat(par_t<0>{}, 0, 1);
  // stage(-1): produces tile 0 from buffers [0], loads rows of tile 1 into [1] (ids [1]) and ids of tile 2 into [0]
  stage(-1, par_t<1>{}, std::false_type{}, std::false_type{});
  stage(0, par_t<0>{}, std::true_type{}, std::false_type{});
  int i = 1;
  for (; i + 1 < cnt; i += 2) {
    stage(i, par_t<1>{}, std::true_type{}, std::true_type{});
    stage(i + 1, par_t<0>{}, std::true_type{}, std::true_type{});
  }
  if (i < cnt) {
    stage(i, par_t<1>{}, std::true_type{}, std::true_type{});
    ++i;
  }
  if (cnt >= 1) {  // the last tile's epilogue (its halves were exchanged before the last barrier)
    const int il = cnt;
    const float4 pp0 = p_partner[(size_t)((il - 1) & 1) * 512], pp1 = p_partner[(size_t)((il - 1) & 1) * 512 + 64];
    float *const orow = reinterpret_cast<float *>(reinterpret_cast<char *>(out + (size_t)(tbeg + (il - 1) * S) * kTM * DOUT) + voff);
    const unsigned grow_prev = (unsigned)(tbeg + (il - 1) * S) * kTM + 4 * (lane >> 5) + 16 * kh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float part = j < 4 ? reinterpret_cast<const float *>(&pp0)[j & 3] : reinterpret_cast<const float *>(&pp1)[j & 3];
      const int vi = max(__float_as_int(fmaf(keep[j] + part, sc, sh)), lo);
      const int rowoff = (j & 3) + 8 * (j >> 2);
      if (!RAGGED || grow_prev + rowoff < R) reinterpret_cast<int *>(orow)[rowoff * DOUT] = vi;
    }
  }
}

int persistent_grid7() {
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
int flex_conv_x7_launch(const float *feat, const float *xyz, const int32_t *nbr, const void *wp3, int B, int N,
                        const EpilogueArgs &ep, float *out, hipStream_t s, int reserve_per_xcd) {
  using C = X7Cfg<DIN, DOUT>;
  const long long R = (long long)B * N;
  const int T = dh3d_cdiv(R, kTM);
  int grid = persistent_grid7() - 8 * (reserve_per_xcd > 0 ? reserve_per_xcd : 0);
  grid = grid < 8 ? 8 : grid;
  if (R % kTM == 0) {
    auto kern = flex_conv_x7_kernel<DIN, DOUT, false>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), C::LDS_BYTES, s, feat, xyz, nbr,
                       static_cast<const uint4 *>(wp3), (unsigned)R, (unsigned)N, ep, out, T);
  } else {
    auto kern = flex_conv_x7_kernel<DIN, DOUT, true>;
    DH3D_ALLOW_BIG_LDS(kern);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), C::LDS_BYTES, s, feat, xyz, nbr,
                       static_cast<const uint4 *>(wp3), (unsigned)R, (unsigned)N, ep, out, T);
  }
  return dh3d_launch_status();
}

}  // namespace

// (internal: dispatched by dh3d_flex_conv_pm_x6_fwd_r in flex_x6.hip)
int dh3d_flex_conv_x7_dispatch(const float *features, const float *xyz, const int32_t *nbr, const void *wpacked_x3,
                               int B, int N, int Din, int Dout, const EpilogueArgs &e, int reserve_cus_per_xcd,
                               float *out, hipStream_t s) {
  if (Din == 32 && Dout == 64)
    return flex_conv_x7_launch<32, 64>(features, xyz, nbr, wpacked_x3, B, N, e, out, s, reserve_cus_per_xcd);
  if (Din == 64 && Dout == 64)
    return flex_conv_x7_launch<64, 64>(features, xyz, nbr, wpacked_x3, B, N, e, out, s, reserve_cus_per_xcd);
  return DH3D_ERR_UNSUPPORTED;
}

#ifdef DH3D_X7_PROBE
DH3D_API int dh3d_x7_probe_read(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x7probe), sizeof(long long) * n) == hipSuccess ? 0 : 3;
}
#endif
#ifdef DH3D_X7_DEV  // dev entry (tools/x7_check.py): the x7 kernel beside the shipped route, same signature as dh3d_flex_conv_pm_x6_fwd
DH3D_API int dh3d_flex_conv_pm_x7_fwd(const float *features, const float *xyz, const int32_t *nbr, const void *wpacked_x3,
                                      int B, int N, int K, int Din, int Dout, const dh3d_epilogue *ep, float *out, void *stream) {
  DH3D_REQUIRE(features && xyz && nbr && wpacked_x3 && out && B > 0 && N > 0 && K == 8);
  return dh3d_flex_conv_x7_dispatch(features, xyz, nbr, wpacked_x3, B, N, Din, Dout, dh3d_ep(ep), 0, out, (hipStream_t)stream);
}
#endif
