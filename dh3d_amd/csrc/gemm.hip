// f32 GEMMs for the BACKWARD passes (gfx950), + the layout helpers the drop-in fast path needs.  Two kernels behind the
// same entries: products of >= 2^26 multiply-adds (K % 32 == 0) run on the bf16 matrix pipe with f32 accuracy
// (gemm_x6_kernel below: both operands split three ways on the fly, six products), small or odd ones on the exact-f32
// pipe (gemm_f32_kernel).
//
//   dh3d_gemm_tn_f32 : C[M,N] (+)= A[K,M]^T * B[K,N]   -- weight gradients dW = X^T dY: the reduction runs over the ROWS
//                      (points) of two row-major activations, split over workgroups and combined with hardware f32
//                      atomics (the reference's theta gradient is atomics too, flex_conv_kernel_gpu.cu.cc:168-248)
//   dh3d_gemm_nn_f32 : C[M,N]    = A[M,K]   * B[K,N]   -- input gradients dX = dY W^T (W^T materialised by the caller)
//   dh3d_transpose32 : [B,R,C] -> [B,C,R] of 32-bit elements (channels-first <-> point-major)
//   dh3d_colsum_f32  : column sums (bias gradients)
//
// gemm_f32_kernel: a workgroup owns a (64*WM) x (64*WN) tile of C, its four waves a 2x2 arrangement of
// (32*WM) x (32*WN) sub-tiles on v_mfma_f32_32x32x2_f32 (a plain f32 fma chain per output -- exact f32, no reduced
// precision).  Operand tiles are staged 16 reduction steps at a time through LDS in "k-major" form [16][tile+4]:
// lane l of a wave then reads A[k = l>>5][i = l&31] and B[k][j] with conflict-free 4-byte LDS reads, one per MFMA
// operand, each reused for WN (WM) MFMAs.  The next chunk's global loads are issued before the current chunk's MFMAs
// (register prefetch, two LDS buffers, one barrier per chunk).
#include "mfma_gemm.h"
#include "bf16x3.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <type_traits>

namespace {

constexpr int kKC = 16;  // reduction steps per LDS stage

// TA: A is stored [K, M] (row = reduction index); otherwise [M, K].
template <bool TA, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, int lda,
                                                      const float *__restrict__ B, int ldb, float *__restrict__ C,
                                                      int ldc, int M, int N, int K, int kchunk, int atomic,
                                                      float *__restrict__ C1, int rows0,
                                                      const float *__restrict__ colbias, int chunks, long long sA,
                                                      long long sB, long long sC, long long sBias) {
  constexpr int BM = 64 * WM, BN = 64 * WN, LA = BM + 4, LB = BN + 4;
  constexpr int NA = kKC * BM / 4 / 256, NBv = kKC * BN / 4 / 256;  // float4 per thread per stage
  static_assert(NA >= 1 && NBv >= 1, "tile too small for 256 threads");
  __shared__ __attribute__((aligned(16))) float s_A[2][kKC * LA];
  __shared__ __attribute__((aligned(16))) float s_B[2][kKC * LB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int bz = blockIdx.z / chunks;  // batch entry (strides sA / sB / sC / sBias elements), then reduction chunk
  const int kbeg = (blockIdx.z - bz * chunks) * kchunk, kend = min(K, kbeg + kchunk);
  if (kbeg >= kend) return;
  A += (size_t)bz * sA; B += (size_t)bz * sB; C += (size_t)bz * sC;
  if (colbias) colbias += (size_t)bz * sBias;

  float4 ra[NA], rb[NBv];
  auto load = [&](int k0) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (TA) {
        const int kr = idx / (BM / 4), c4 = (idx % (BM / 4)) * 4;
        const int k = k0 + kr, m = m0 + c4;
        if (k < kend && m < M) v = *reinterpret_cast<const float4 *>(A + (size_t)k * lda + m);  // M % 4 == 0
      } else {
        const int row = idx / (kKC / 4), kq = idx % (kKC / 4);
        const int m = m0 + row, k = k0 + 4 * kq;
        if (m < M) {
          const float *p = A + (size_t)m * lda + k;
          if (k + 3 < kend) v = *reinterpret_cast<const float4 *>(p);  // lda % 4 == 0, kbeg % 4 == 0
          else {
            if (k < kend) v.x = p[0];
            if (k + 1 < kend) v.y = p[1];
            if (k + 2 < kend) v.z = p[2];
          }
        }
      }
      ra[u] = v;
    }
#pragma unroll
    for (int u = 0; u < NBv; ++u) {
      const int idx = tid + 256 * u;
      const int kr = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
      const int k = k0 + kr, n = n0 + c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k < kend && n < N) v = *reinterpret_cast<const float4 *>(B + (size_t)k * ldb + n);  // N % 4 == 0
      rb[u] = v;
    }
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int idx = tid + 256 * u;
      if (TA) {
        const int kr = idx / (BM / 4), c4 = (idx % (BM / 4)) * 4;
        *reinterpret_cast<float4 *>(&s_A[buf][kr * LA + c4]) = ra[u];
      } else {
        const int row = idx / (kKC / 4), kq = idx % (kKC / 4);
        float *d = &s_A[buf][(4 * kq) * LA + row];
        d[0] = ra[u].x; d[LA] = ra[u].y; d[2 * LA] = ra[u].z; d[3 * LA] = ra[u].w;
      }
    }
#pragma unroll
    for (int u = 0; u < NBv; ++u) {
      const int idx = tid + 256 * u;
      const int kr = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
      *reinterpret_cast<float4 *>(&s_B[buf][kr * LB + c4]) = rb[u];
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int mb = (wave & 1) * 32 * WM + (lane & 31), nb = (wave >> 1) * 32 * WN + (lane & 31), h = lane >> 5;

  load(kbeg);
  stage(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += kKC) {
    const bool more = k0 + kKC < kend;
    if (more) load(k0 + kKC);
    const float *a = &s_A[buf][h * LA + mb], *b = &s_B[buf][h * LB + nb];
#pragma unroll
    for (int kk = 0; kk < kKC; kk += 2) {
      float av[WM], bv[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) av[i] = a[kk * LA + 32 * i];
#pragma unroll
      for (int j = 0; j < WN; ++j) bv[j] = b[kk * LB + 32 * j];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (more) stage(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + (wave >> 1) * 32 * WN + 32 * j + (lane & 31);
      const float cb = (colbias && n < N) ? colbias[n] : 0.f;  // only with a single, non-atomic reduction chunk
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wave & 1) * 32 * WM + 32 * i + mfma_row(r, lane);
        if (m < M && n < N) {
          float *dst = (C1 && m >= rows0) ? C1 + (size_t)(m - rows0) * ldc + n : C + (size_t)m * ldc + n;
          if (atomic) unsafeAtomicAdd(dst, acc[i][j][r]);
          else *dst = acc[i][j][r] + cb;
        }
      }
    }
}

// The same products on the bf16 matrix pipe with f32 accuracy (bf16x6, bf16x3.h) -- BOTH operands change every
// training step, so both are split on the fly: a thread splits each element ONCE where it stages it (exact 8+8+8-bit
// split, three bf16 planes in LDS, row = output index, 32 reduction steps + 8 pad contiguous), and a wave reads its
// fragments with one 16-byte LDS read per plane.  Six v_mfma_f32_32x32x16_bf16 per 16 reduction steps replace eight
// v_mfma_f32_32x32x2_f32 of twice the duration (2.7x less matrix-pipe time); products ordered small terms first and
// interleaved over the wave's 2 x WN accumulator tiles, so no MFMA waits on its predecessor.  Operands stored with the
// reduction index as the ROW ([K,M] of the tn form, every [K,N]) are loaded as dwords down a column (coalesced across
// the lanes) and packed eight to a store.  One LDS buffer + register prefetch: 60 KB, two workgroups per CU.
__device__ __forceinline__ void split3x8(const float *v, uint4 &c1, uint4 &c2, uint4 &c3) {
  uint2 a1, a2, a3, b1, b2, b3;
  split3x4(make_float4(v[0], v[1], v[2], v[3]), a1, a2, a3);
  split3x4(make_float4(v[4], v[5], v[6], v[7]), b1, b2, b3);
  c1 = make_uint4(a1.x, a1.y, b1.x, b1.y);
  c2 = make_uint4(a2.x, a2.y, b2.x, b2.y);
  c3 = make_uint4(a3.x, a3.y, b3.x, b3.y);
}

// KC reduction steps per stage: 16 -> 37 KB of LDS and <= 128 registers, four workgroups (16 waves) per CU hide the
// load latency that one stage of MFMAs (0.3 us) cannot; 32 -> 60 KB, two workgroups.  (A second register set with the
// loads of TWO stages in flight, three workgroups per CU, measured the same to the microsecond: what the waves wait for
// -- PMC: 47 % of their cycles in s_waitcnt / s_barrier, 34 % issue-stalled, MFMA and VALU each ~28 % busy -- is each
// other at the two barriers of a stage, not memory.)
template <bool TA, int WN, int KC>
__global__ __launch_bounds__(256, KC == 16 ? 4 : 2) void gemm_x6_kernel(
    const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc, int M,
    int N, int K, int kchunk, int atomic, float *__restrict__ C1, int rows0, const float *__restrict__ colbias,
    int chunks, long long sA, long long sB, long long sC, long long sBias) {
  constexpr int BM = 128, BN = 64 * WN, LD = KC + 8;
  constexpr int NG = KC / 8;                                  // groups of 8 reduction steps per column
  constexpr int TA_ = 256 / BM, TB_ = 256 / BN;               // threads along the reduction, column-wise operands
  constexpr int GA = (NG + TA_ - 1) / TA_, GB = (NG + TB_ - 1) / TB_;  // groups per thread
  constexpr int NF = BM * KC / 4 / 256;                       // float4 per thread of a row-wise A
  __shared__ __attribute__((aligned(16))) unsigned short s_A[3][BM * LD];
  __shared__ __attribute__((aligned(16))) unsigned short s_B[3][BN * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int bz = blockIdx.z / chunks;
  const int kbeg = (blockIdx.z - bz * chunks) * kchunk, kend = min(K, kbeg + kchunk);
  if (kbeg >= kend) return;
  A += (size_t)bz * sA; B += (size_t)bz * sB; C += (size_t)bz * sC;
  if (colbias) colbias += (size_t)bz * sBias;

  // K % 32 == 0 (dispatch): every stage is whole, no masks or clamps along the reduction.  Every load is unconditional
  // (a load under a branch is merged through register copies that wait for it): rows / columns past the edge are
  // clamped once and zeroed where they are staged; threads without a group of their own (narrow tiles) re-read the last
  // one and only skip the LDS store.  Addresses are a wave-uniform base (scalar unit) + a per-lane 32-bit offset fixed
  // for the kernel -- the vector unit is left to the split (PMC: VALU and MFMA busy were equal, 8 VALU per MFMA).
  float4 ra[TA ? 1 : NF];
  float ca[TA ? GA : 1][8], cb[GB][8];
  const int acol = tid & (BM - 1), akg = __builtin_amdgcn_readfirstlane(tid / BM);  // TA: column of A, its first group
  const int bcol = tid & (BN - 1), bkg = __builtin_amdgcn_readfirstlane(tid / BN);
  const unsigned am = (unsigned)min(m0 + acol, M - 1), bn = (unsigned)min(n0 + bcol, N - 1);
  const bool edge = (m0 + BM > M) | (n0 + BN > N);
  unsigned aoff[TA ? 1 : NF];
  if (!TA) {
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int idx = tid + 256 * u, row = idx / (KC / 4), kq = idx % (KC / 4);
      aoff[u] = (unsigned)min(m0 + row, M - 1) * (unsigned)lda + 4u * kq;  // M * lda < 2^30 (dispatch)
    }
  }
  auto load = [&](int k0) {
    if (!TA) {
      const float *Ak = A + k0;
#pragma unroll
      for (int u = 0; u < NF; ++u) ra[u] = *reinterpret_cast<const float4 *>(Ak + aoff[u]);
    } else {
#pragma unroll
      for (int g = 0; g < GA; ++g) {
        const float *Ak = A + (size_t)(k0 + 8 * min(akg + TA_ * g, NG - 1)) * lda;
#pragma unroll
        for (int j = 0; j < 8; ++j) ca[g][j] = Ak[(size_t)j * lda + am];
      }
    }
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      const float *Bk = B + (size_t)(k0 + 8 * min(bkg + TB_ * g, NG - 1)) * ldb;
#pragma unroll
      for (int j = 0; j < 8; ++j) cb[g][j] = Bk[(size_t)j * ldb + bn];
    }
  };
  auto keep = [](float x, bool ok) { return __uint_as_float(__float_as_uint(x) & (0u - (unsigned)ok)); };
  auto stage = [&](auto ragged_c) {
    constexpr bool RG = decltype(ragged_c)::value;
    if (!TA) {
#pragma unroll
      for (int u = 0; u < NF; ++u) {
        const int idx = tid + 256 * u, row = idx / (KC / 4), kq = idx % (KC / 4);
        float4 v = ra[u];
        if (RG) {
          const bool ok = m0 + row < M;
          v = make_float4(keep(v.x, ok), keep(v.y, ok), keep(v.z, ok), keep(v.w, ok));
        }
        uint2 c1, c2, c3;
        split3x4(v, c1, c2, c3);
        *reinterpret_cast<uint2 *>(&s_A[0][row * LD + 4 * kq]) = c1;
        *reinterpret_cast<uint2 *>(&s_A[1][row * LD + 4 * kq]) = c2;
        *reinterpret_cast<uint2 *>(&s_A[2][row * LD + 4 * kq]) = c3;
      }
    } else {
      const bool mok = m0 + acol < M;
#pragma unroll
      for (int g = 0; g < GA; ++g) {
        const int kg = 8 * (akg + TA_ * g);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = RG ? keep(ca[g][j], mok) : ca[g][j];
        uint4 c1, c2, c3;
        split3x8(v, c1, c2, c3);
        if (NG % TA_ == 0 || kg < KC) {
          *reinterpret_cast<uint4 *>(&s_A[0][acol * LD + kg]) = c1;
          *reinterpret_cast<uint4 *>(&s_A[1][acol * LD + kg]) = c2;
          *reinterpret_cast<uint4 *>(&s_A[2][acol * LD + kg]) = c3;
        }
      }
    }
    const bool nok = n0 + bcol < N;
#pragma unroll
    for (int g = 0; g < GB; ++g) {
      const int kg = 8 * (bkg + TB_ * g);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = RG ? keep(cb[g][j], nok) : cb[g][j];
      uint4 c1, c2, c3;
      split3x8(v, c1, c2, c3);
      if (NG % TB_ == 0 || kg < KC) {
        *reinterpret_cast<uint4 *>(&s_B[0][bcol * LD + kg]) = c1;
        *reinterpret_cast<uint4 *>(&s_B[1][bcol * LD + kg]) = c2;
        *reinterpret_cast<uint4 *>(&s_B[2][bcol * LD + kg]) = c3;
      }
    }
  };

  f32x16 acc[2][WN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int arow = (wave & 1) * 64 + (lane & 31), brow = (wave >> 1) * 32 * WN + (lane & 31), kh8 = 8 * (lane >> 5);

  load(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += KC) {  // (kend - kbeg) % KC == 0 (dispatch)
    if (edge) stage(std::true_type{});
    else stage(std::false_type{});
    __syncthreads();
    load(k0 + KC < kend ? k0 + KC : k0);  // (the last stage re-reads itself: unconditional, in bounds, never staged)
    __builtin_amdgcn_sched_barrier(0);    // ... and in flight BEFORE the products, not wherever the scheduler sinks it
#pragma unroll
    for (int ks = 0; ks < KC; ks += 16) {
      bf16x8 a[2][3], b[WN][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i][p] = *reinterpret_cast<const bf16x8 *>(&s_A[p][(arow + 32 * i) * LD + ks + kh8]);
#pragma unroll
        for (int j = 0; j < WN; ++j)
          b[j][p] = *reinterpret_cast<const bf16x8 *>(&s_B[p][(brow + 32 * j) * LD + ks + kh8]);
      }
#define DH3D_X6_PRODUCT(PA, PB)                                                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < WN; ++j)                  \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], acc[i][j], 0, 0, 0);
      DH3D_X6_PRODUCT(2, 0) DH3D_X6_PRODUCT(0, 2) DH3D_X6_PRODUCT(1, 1)
      DH3D_X6_PRODUCT(1, 0) DH3D_X6_PRODUCT(0, 1) DH3D_X6_PRODUCT(0, 0)
#undef DH3D_X6_PRODUCT
    }
    __syncthreads();
  }

  if (!edge && !C1) {  // interior tile: no guards
    float *c0 = C + (size_t)(m0 + (wave & 1) * 64 + 4 * (lane >> 5)) * ldc + n0 + (wave >> 1) * 32 * WN + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const float bias = colbias ? colbias[n0 + (wave >> 1) * 32 * WN + 32 * j + (lane & 31)] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float *dst = c0 + (size_t)(32 * i + (r & 3) + 8 * (r >> 2)) * ldc + 32 * j;
          if (atomic) unsafeAtomicAdd(dst, acc[i][j][r]);
          else *dst = acc[i][j][r] + bias;
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = n0 + (wave >> 1) * 32 * WN + 32 * j + (lane & 31);
      const float bias = (colbias && n < N) ? colbias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wave & 1) * 64 + 32 * i + mfma_row(r, lane);
        if (m < M && n < N) {
          float *dst = (C1 && m >= rows0) ? C1 + (size_t)(m - rows0) * ldc + n : C + (size_t)m * ldc + n;
          if (atomic) unsafeAtomicAdd(dst, acc[i][j][r]);
          else *dst = acc[i][j][r] + bias;
        }
      }
    }
}

// [Bt][R][Cc] -> [Bt][Cc][R], 32-bit elements, 32x32 tiles through LDS.  The output may be a column block of a wider
// matrix: row stride `ldo` (>= R) and batch stride `obs` elements (dense: ldo = R, obs = R*Cc).
__global__ __launch_bounds__(256) void transpose32_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                         int R, int Cc, long long ldo, long long obs) {
  __shared__ uint32_t tile[32][33];
  const size_t base = (size_t)blockIdx.z * R * Cc, obase = (size_t)blockIdx.z * obs;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = r0 + ty + 8 * u, c = c0 + tx;
    if (r < R && c < Cc) tile[ty + 8 * u][tx] = in[base + (size_t)r * Cc + c];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = c0 + ty + 8 * u, r = r0 + tx;
    if (r < R && c < Cc) out[obase + (size_t)c * ldo + r] = tile[tx][ty + 8 * u];
  }
}

// out[c] (+)= sum_r x[r, c]; grid (ceil(C/64), row chunks); 256 threads = 4 row lanes x 64 columns
__global__ __launch_bounds__(256) void colsum_kernel(const float *__restrict__ x, long long R, int Cc, int rows_per,
                                                    float *__restrict__ out) {
  __shared__ float s[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
  float a = 0.f;
  if (c < Cc)
    for (long long r = r0 + q; r < r1; r += 4) a += x[r * Cc + c];
  s[q][threadIdx.x & 63] = a;
  __syncthreads();
  if (q == 0 && c < Cc) unsafeAtomicAdd(out + c, (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]));
}

// ---- products with one tiny dimension (the [clouds, 256] layers behind NetVLAD: 22 rows at the Oxford batch).  The
// 128-row tiles above run them as one or two workgroups looping over K with 6 of 128 rows in use: 26-28 us for 1.4
// MFLOP.  Plain f32 FMAs here (exact f32, like the matrix-pipe kernel), the small operand in LDS.
//   rows form:  C[M <= 32, N] (+)= A[M, K <= 512] B[K, N] (+ bias): a workgroup owns 64 columns, lane = column, the
//               four waves split the rows; B is read once, coalesced
constexpr int kSkM = 32, kSkK = 512;
template <int RB>  // rows in use, in blocks of eight
__global__ __launch_bounds__(256) void gemm_rows_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                       int ldb, float *__restrict__ C, int ldc, int M, int N, int K,
                                                       int accumulate, const float *__restrict__ colbias) {
  extern __shared__ __attribute__((aligned(16))) float s_a[];  // [32][KP], rows >= M and columns >= K zero
  const int KP = (K + 63) / 64 * 64;
  {  // eight threads per row, 16-byte loads, all of a thread's loads in flight together (K % 4 == 0, lda % 4 == 0)
    const int r = threadIdx.x >> 3, q = (threadIdx.x & 7) * 4, rr = r < M ? r : 0;
    float4 v[kSkK / 32];
#pragma unroll
    for (int j = 0; j < kSkK / 32; ++j) {
      const int k = q + 32 * j, kc = k < K ? k : 0;
      v[j] = *reinterpret_cast<const float4 *>(A + (size_t)rr * lda + kc);
    }
#pragma unroll
    for (int j = 0; j < kSkK / 32; ++j) {
      const int k = q + 32 * j;
      if (k < KP)
        *reinterpret_cast<float4 *>(s_a + r * KP + k) = (r < M && k < K) ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = blockIdx.x * 64 + lane, cc = col < N ? col : N - 1;
  // The four waves split K (a quarter each, all 32 rows): at K = 256 a wave's whole share of B is 64 loads in flight at
  // once -- ONE round trip to L2 / HBM where rows split over the waves made four (the kernel is nothing but latency:
  // 22 of 26 us); rows of A as broadcast 16-byte LDS reads, no test in the loop (the padding is zero).  Partial sums
  // meet in LDS.
  constexpr int NR = 8 * RB;
  float acc[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) acc[i] = 0.f;
  const int kq = KP / 4;  // multiple of 16
  for (int k0 = wave * kq; k0 < (wave + 1) * kq; k0 += 64) {
    const int span = min(64, (wave + 1) * kq - k0);  // 16, 32, 48 or 64
    float b[64];
#pragma unroll
    for (int u = 0; u < 64; ++u) {  // (clamped address, then a select: no load under a branch; 0 beyond K, not 0 * B)
      const float v = B[(size_t)(k0 + u < K ? k0 + u : K - 1) * ldb + cc];
      b[u] = k0 + u < K ? v : 0.f;
    }
#pragma unroll
    for (int u4 = 0; u4 < 16; ++u4) {
      if (4 * u4 < span) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const float4 av = *reinterpret_cast<const float4 *>(s_a + i * KP + k0 + 4 * u4);
          acc[i] = fmaf(av.x, b[4 * u4], acc[i]); acc[i] = fmaf(av.y, b[4 * u4 + 1], acc[i]);
          acc[i] = fmaf(av.z, b[4 * u4 + 2], acc[i]); acc[i] = fmaf(av.w, b[4 * u4 + 3], acc[i]);
        }
      }
    }
  }
  float *s_red = s_a + kSkM * KP;  // [4][32][64]
#pragma unroll
  for (int i = 0; i < NR; ++i) s_red[(wave * kSkM + i) * 64 + lane] = acc[i];
  __syncthreads();
  if (col < N) {
    const float bias = colbias ? colbias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < kSkM / 4; ++i) {
      const int r = wave + 4 * i;
      if (r < M) {
        const float v = (s_red[(0 * kSkM + r) * 64 + lane] + s_red[(1 * kSkM + r) * 64 + lane]) +
                        (s_red[(2 * kSkM + r) * 64 + lane] + s_red[(3 * kSkM + r) * 64 + lane]);
        float *o = C + (size_t)r * ldc + col;
        *o = (accumulate ? *o : bias) + v;
      }
    }
  }
}
//   short-reduction form:  C[M, N] (+)= A[K <= 32, M]^T B[K, N] (weight gradients of those layers: K = clouds): a workgroup
//               owns 64 x 64 outputs, both operand strips in LDS; bound by the store of C
__global__ __launch_bounds__(256) void gemm_shortk_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                         int ldb, float *__restrict__ C, int ldc, int M, int N, int K,
                                                         int accumulate) {
  __shared__ float s_a[kSkM][64], s_b[kSkM][64];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < K * 64; e += 256) {
    const int k = e >> 6, j = e & 63;
    s_a[k][j] = m0 + j < M ? A[(size_t)k * lda + m0 + j] : 0.f;
    s_b[k][j] = n0 + j < N ? B[(size_t)k * ldb + n0 + j] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k = 0; k < K; ++k) {
    const float b = s_b[k][lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = fmaf(s_a[k][wave * 16 + i], b, acc[i]);
  }
  if (n0 + lane < N) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = m0 + wave * 16 + i;
      if (r < M) {
        float *o = C + (size_t)r * ldc + n0 + lane;
        *o = accumulate ? *o + acc[i] : acc[i];
      }
    }
  }
}

struct GemmBatch { int n; long long sA, sB, sC, sBias; };

// reduction chunks of a product (> 1: the split partials meet in f32 atomics on a zeroed C)
int gemm_chunks(bool ta, int M, int N, int K, int batch, int wgs) {
  if (batch == 1 && ((!ta && M <= kSkM && K <= kSkK && K % 4 == 0) || (ta && K <= kSkM))) return 1;  // plain-FMA kernels
  const int BN = N <= 64 ? 64 : 128;
  const int gm = dh3d_cdiv(M, 128), gn = dh3d_cdiv(N, BN);
  int chunks = dh3d_cdiv(wgs, gm * gn * batch);
  const int maxc = dh3d_cdiv(K, ta ? 64 : 256);  // [M,K] operands: only long reductions are worth the atomics
  chunks = chunks > maxc ? maxc : chunks;
  return chunks < 1 ? 1 : chunks;
}

int gemm_launch(bool ta, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int N, int K,
                float *C1, int rows0, bool accumulate, hipStream_t s, const float *colbias = nullptr,
                GemmBatch bt = GemmBatch{1, 0, 0, 0, 0}) {
  if (bt.n == 1 && !C1) {  // one tiny dimension: the plain-FMA kernels above
    if (!ta && M <= kSkM && K <= kSkK && K % 4 == 0 && lda % 4 == 0) {
      const size_t lds = sizeof(float) * (kSkM * ((K + 63) / 64 * 64) + 4 * kSkM * 64);
#define DH3D_ROWS(RBV)                                                                                                 \
  do {                                                                                                                 \
    DH3D_ALLOW_BIG_LDS(gemm_rows_kernel<RBV>);                                                                         \
    hipLaunchKernelGGL(gemm_rows_kernel<RBV>, dim3(dh3d_cdiv(N, 64)), dim3(256), lds, s, A, lda, B, ldb, C, ldc, M, N, \
                       K, accumulate ? 1 : 0, colbias);                                                                \
  } while (0)
      if (M <= 8) DH3D_ROWS(1); else if (M <= 16) DH3D_ROWS(2); else if (M <= 24) DH3D_ROWS(3); else DH3D_ROWS(4);
#undef DH3D_ROWS
      return dh3d_launch_status();
    }
    if (ta && K <= kSkM && !colbias) {
      hipLaunchKernelGGL(gemm_shortk_kernel, dim3(dh3d_cdiv(N, 64), dh3d_cdiv(M, 64)), dim3(256), 0, s, A, lda, B, ldb, C,
                         ldc, M, N, K, accumulate ? 1 : 0);
      return dh3d_launch_status();
    }
  }
  // tile: 128 x 128 (or 128 x 64 for narrow N); the reduction is split so that ~3 workgroups per CU exist
  const bool narrow = N <= 64;
  const int BM = 128, BN = narrow ? 64 : 128;
  const int gm = dh3d_cdiv(M, BM), gn = dh3d_cdiv(N, BN);
  // workgroups aimed at: the split partials meet in f32 atomics (~0.3 T/s chip-wide), so the long-reduction tn form
  // wants fewer, longer chunks (tools/gemm_bench.py wgs: 384 beats 768 by 10-25 % there); a -DDH3D_DEV build reads
  // DH3D_GEMM_WGS / DH3D_GEMM_F32 / DH3D_GEMM_KC from the environment (A/B timing); the shipped library has no such state
#ifdef DH3D_DEV
  static const int wgs_env = [] { const char *e = getenv("DH3D_GEMM_WGS"); return e ? atoi(e) : 0; }();
#else
  constexpr int wgs_env = 0;
#endif
  const int wgs = wgs_env > 0 ? wgs_env : ta ? 384 : 768;
  int chunks = gemm_chunks(ta, M, N, K, bt.n, wgs);
  if (colbias) chunks = 1;
  // products of >= 2^26 multiply-adds with K % 32 == 0 go to the bf16x6 kernel (f32-accurate); small or odd-K ones stay on
  // the exact-f32 pipe.  DH3D_GEMM_F32=1 keeps everything there (A/B timing, bit-level comparisons)
#ifdef DH3D_DEV
  static const bool force_f32 = [] { const char *e = getenv("DH3D_GEMM_F32"); return e && e[0] == '1'; }();
#else
  constexpr bool force_f32 = false;
#endif
  const bool x6 = !force_f32 && K % 32 == 0 && (double)M * N * K * bt.n >= 67108864.0 &&
                  (ta || (lda % 4 == 0 && (double)M * lda < 1073741824.0));
  // stage depth (measured, tools/gemm_bench.py): 16 for the 128-wide tiles (four workgroups per CU), 32 for N <= 64
  // (the A stream dominates: whole 128-byte lines per row); DH3D_GEMM_KC=16|32 forces one (dev)
#ifdef DH3D_DEV
  static const int x6kc = [] { const char *e = getenv("DH3D_GEMM_KC"); return e ? atoi(e) : 0; }();
#else
  constexpr int x6kc = 0;
#endif
  const int kc = !x6 ? kKC : (x6kc == 16 || x6kc == 32) ? x6kc : narrow ? 32 : 16;
  const int kgran = x6 ? 32 : kc;  // chunk granularity: whole stages of either depth
  int kchunk = dh3d_cdiv(K, chunks);
  kchunk = (kchunk + kgran - 1) / kgran * kgran;  // multiple of the stage depth: float4 loads of the [M,K] operand stay aligned
  chunks = dh3d_cdiv(K, kchunk);
  const int atomic = (chunks > 1 || accumulate) ? 1 : 0;
  if (colbias && atomic) return DH3D_ERR_UNSUPPORTED;
  if ((long long)chunks * bt.n > 65535) return DH3D_ERR_UNSUPPORTED;
  if (atomic && !accumulate) {  // the split partials add into zeros
    if (C1) {
      if (hipMemsetAsync(C, 0, sizeof(float) * (size_t)rows0 * ldc, s) != hipSuccess) return DH3D_ERR_LAUNCH;
      if (hipMemsetAsync(C1, 0, sizeof(float) * (size_t)(M - rows0) * ldc, s) != hipSuccess) return DH3D_ERR_LAUNCH;
    } else if (ldc == N && (bt.n == 1 || bt.sC == (long long)M * N)) {
      if (hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * N * bt.n, s) != hipSuccess) return DH3D_ERR_LAUNCH;
    } else if (bt.n == 1) {
      if (hipMemset2DAsync(C, sizeof(float) * ldc, 0, sizeof(float) * N, M, s) != hipSuccess) return DH3D_ERR_LAUNCH;
    } else {
      return DH3D_ERR_UNSUPPORTED;
    }
  }
  const dim3 grid(gn, gm, chunks * bt.n), block(256);
#define DH3D_GEMM(TAV, WNV)                                                                                          \
  hipLaunchKernelGGL((gemm_f32_kernel<TAV, 2, WNV>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, kchunk, atomic, \
                     C1, rows0, colbias, chunks, bt.sA, bt.sB, bt.sC, bt.sBias)
#define DH3D_GEMM6(TAV, WNV, KCV)                                                                                    \
  hipLaunchKernelGGL((gemm_x6_kernel<TAV, WNV, KCV>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, kchunk,       \
                     atomic, C1, rows0, colbias, chunks, bt.sA, bt.sB, bt.sC, bt.sBias)
#define DH3D_GEMM6K(TAV, WNV) do { if (kc == 16) DH3D_GEMM6(TAV, WNV, 16); else DH3D_GEMM6(TAV, WNV, 32); } while (0)
  if (x6) {
    if (ta) { if (narrow) DH3D_GEMM6K(true, 1); else DH3D_GEMM6K(true, 2); }
    else { if (narrow) DH3D_GEMM6K(false, 1); else DH3D_GEMM6K(false, 2); }
  } else if (ta) { if (narrow) DH3D_GEMM(true, 1); else DH3D_GEMM(true, 2); }
  else { if (narrow) DH3D_GEMM(false, 1); else DH3D_GEMM(false, 2); }
#undef DH3D_GEMM
#undef DH3D_GEMM6
#undef DH3D_GEMM6K
  return dh3d_launch_status();
}

}  // namespace

// internal (other translation units of the library)
int dh3d_internal_gemm(bool ta, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int N, int K,
                       float *C1, int rows0, bool accumulate, hipStream_t s) {
  return gemm_launch(ta, A, lda, B, ldb, C, ldc, M, N, K, C1, rows0, accumulate, s);
}
int dh3d_internal_transpose32(const void *in, void *out, int Bt, int R, int Cc, long long ldo, long long obs,
                              hipStream_t s) {
  hipLaunchKernelGGL(transpose32_kernel, dim3(dh3d_cdiv(Cc, 32), dh3d_cdiv(R, 32), Bt), dim3(256), 0, s,
                     static_cast<const uint32_t *>(in), static_cast<uint32_t *>(out), R, Cc, ldo ? ldo : R,
                     ldo ? obs : (long long)R * Cc);
  return dh3d_launch_status();
}

// 1 when the product's reduction is split (the library then zeroes C and adds the partials with atomics -- or, with
// accumulate = 1, adds them onto what the caller put there: a caller that owns a zeroed arena passes accumulate = 1 and
// saves the fill).  ta: 1 = the tn form.  Pure function of the shape.
DH3D_API int dh3d_gemm_is_split(int ta, int M, int N, int K, int batch) {
  return gemm_chunks(ta != 0, M, N, K, batch > 0 ? batch : 1, ta ? 384 : 768) > 1 ? 1 : 0;
}

DH3D_API int dh3d_gemm_tn_f32(const float *A, const float *B, int K, int M, int N, int accumulate, float *C,
                              void *stream) {
  DH3D_REQUIRE(A && B && C && K > 0 && M > 0 && N > 0);
  DH3D_SUPPORTED(M % 4 == 0 && N % 4 == 0 && M / 128 <= 65535);
  return gemm_launch(true, A, M, B, N, C, N, M, N, K, nullptr, 0, accumulate != 0, (hipStream_t)stream);
}

DH3D_API int dh3d_gemm_nn_f32(const float *A, const float *B, const float *colbias, int M, int K, int N,
                              int accumulate, float *C, void *stream) {
  DH3D_REQUIRE(A && B && C && K > 0 && M > 0 && N > 0 && !(colbias && accumulate));
  DH3D_SUPPORTED(K % 4 == 0 && N % 4 == 0 && dh3d_cdiv(M, 128) <= 65535);
  return gemm_launch(false, A, K, B, N, C, N, M, N, K, nullptr, 0, accumulate != 0, (hipStream_t)stream, colbias);
}

// batched forms: `batch` independent products, operands of entry b at A + b*K*M etc. (dense, back to back);
// colbias (nn only) [batch, N]
DH3D_API int dh3d_gemm_tn_f32_batched(const float *A, const float *B, int batch, int K, int M, int N, float *C,
                                      void *stream) {
  DH3D_REQUIRE(A && B && C && batch > 0 && K > 0 && M > 0 && N > 0);
  DH3D_SUPPORTED(M % 4 == 0 && N % 4 == 0 && M / 128 <= 65535);
  return gemm_launch(true, A, M, B, N, C, N, M, N, K, nullptr, 0, false, (hipStream_t)stream, nullptr,
                     GemmBatch{batch, (long long)K * M, (long long)K * N, (long long)M * N, 0});
}

int dh3d_internal_gemm_tn_batched(const float *A, const float *B, int batch, int K, int M, int N, float *C,
                                  bool accumulate, hipStream_t s) {
  return gemm_launch(true, A, M, B, N, C, N, M, N, K, nullptr, 0, accumulate, s, nullptr,
                     GemmBatch{batch, (long long)K * M, (long long)K * N, (long long)M * N, 0});
}

DH3D_API int dh3d_gemm_nn_f32_batched(const float *A, const float *B, const float *colbias, int batch, int M, int K,
                                      int N, float *C, void *stream) {
  DH3D_REQUIRE(A && B && C && batch > 0 && K > 0 && M > 0 && N > 0);
  DH3D_SUPPORTED(K % 4 == 0 && N % 4 == 0 && dh3d_cdiv(M, 128) <= 65535);
  return gemm_launch(false, A, K, B, N, C, N, M, N, K, nullptr, 0, false, (hipStream_t)stream, colbias,
                     GemmBatch{batch, (long long)M * K, (long long)K * N, (long long)M * N, N});
}

DH3D_API int dh3d_transpose32(const void *in, int Bt, int R, int Cc, void *out, void *stream) {
  DH3D_REQUIRE(in && out && Bt > 0 && R > 0 && Cc > 0);
  DH3D_SUPPORTED(Bt <= 65535 && dh3d_cdiv(R, 32) <= 65535);
  return dh3d_internal_transpose32(in, out, Bt, R, Cc, 0, 0, (hipStream_t)stream);
}

DH3D_API int dh3d_colsum_f32(const float *x, long long R, int Cc, int accumulate, float *out, void *stream) {
  DH3D_REQUIRE(x && out && R > 0 && Cc > 0);
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * Cc, s) != hipSuccess) return DH3D_ERR_LAUNCH;
  int chunks = dh3d_cdiv(R, 256);
  chunks = chunks > 512 ? 512 : chunks;
  const int rows_per = dh3d_cdiv(R, chunks);
  hipLaunchKernelGGL(colsum_kernel, dim3(dh3d_cdiv(Cc, 64), dh3d_cdiv(R, rows_per)), dim3(256), 0, s, x, R, Cc, rows_per, out);
  return dh3d_launch_status();
}
