// Exact-f32 MFMA GEMMs for the BACKWARD passes (gfx950), + the layout helpers the drop-in fast path needs.
//
//   dh3d_gemm_tn_f32 : C[M,N] (+)= A[K,M]^T * B[K,N]   -- weight gradients dW = X^T dY: the reduction runs over the ROWS
//                      (points) of two row-major activations, split over workgroups and combined with hardware f32
//                      atomics (the reference's theta gradient is atomics too, flex_conv_kernel_gpu.cu.cc:168-248)
//   dh3d_gemm_nn_f32 : C[M,N]    = A[M,K]   * B[K,N]   -- input gradients dX = dY W^T (W^T materialised by the caller)
//   dh3d_transpose32 : [B,R,C] -> [B,C,R] of 32-bit elements (channels-first <-> point-major)
//   dh3d_colsum_f32  : column sums (bias gradients)
//
// One kernel template: a workgroup owns a (64*WM) x (64*WN) tile of C, its four waves a 2x2 arrangement of
// (32*WM) x (32*WN) sub-tiles on v_mfma_f32_32x32x2_f32 (a plain f32 fma chain per output -- exact f32, no reduced
// precision).  Operand tiles are staged 16 reduction steps at a time through LDS in "k-major" form [16][tile+4]:
// lane l of a wave then reads A[k = l>>5][i = l&31] and B[k][j] with conflict-free 4-byte LDS reads, one per MFMA
// operand, each reused for WN (WM) MFMAs.  The next chunk's global loads are issued before the current chunk's MFMAs
// (register prefetch, two LDS buffers, one barrier per chunk).
#include "mfma_gemm.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

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

struct GemmBatch { int n; long long sA, sB, sC, sBias; };

int gemm_launch(bool ta, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int N, int K,
                float *C1, int rows0, bool accumulate, hipStream_t s, const float *colbias = nullptr,
                GemmBatch bt = GemmBatch{1, 0, 0, 0, 0}) {
  // tile: 128 x 128 (or 128 x 64 for narrow N); the reduction is split so that ~3 workgroups per CU exist
  const bool narrow = N <= 64;
  const int BM = 128, BN = narrow ? 64 : 128;
  const int gm = dh3d_cdiv(M, BM), gn = dh3d_cdiv(N, BN);
  int chunks = dh3d_cdiv(768, gm * gn * bt.n);
  const int maxc = dh3d_cdiv(K, ta ? 64 : 256);  // [M,K] operands: only long reductions are worth the atomics
  chunks = chunks > maxc ? maxc : chunks;
  if (chunks < 1 || colbias) chunks = 1;
  int kchunk = dh3d_cdiv(K, chunks);
  kchunk = (kchunk + kKC - 1) / kKC * kKC;  // multiple of 16: float4 loads of the [M,K] operand stay aligned
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
  if (ta) { if (narrow) DH3D_GEMM(true, 1); else DH3D_GEMM(true, 2); }
  else { if (narrow) DH3D_GEMM(false, 1); else DH3D_GEMM(false, 2); }
#undef DH3D_GEMM
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
