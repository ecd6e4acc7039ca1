// Entry points shared between translation units of libdh3d_hip.so (not exported).
#pragma once
#include "common.h"

// C[M,N] (+)= op(A) * B on the exact-f32 MFMA pipe (gemm.hip).  ta: A is [K,M] (reduction over rows, split + f32
// atomics) else [M,K].  C1 / rows0: rows >= rows0 of C go to C1 (row m - rows0) -- e.g. [bias; theta] gradients.
int dh3d_internal_gemm(bool ta, const float *A, int lda, const float *B, int ldb, float *C, int ldc, int M, int N,
                       int K, float *C1, int rows0, bool accumulate, hipStream_t s);
// [Bt][R][Cc] -> [Bt][Cc][R] (32-bit elements); ldo != 0: output row stride ldo and batch stride obs (elements)
int dh3d_internal_transpose32(const void *in, void *out, int Bt, int R, int Cc, long long ldo, long long obs,
                              hipStream_t s);
// `batch` independent C[b] (+)= A[b]^T B[b]: A [batch,K,M], B [batch,K,N], C [batch,M,N], stored back to back
int dh3d_internal_gemm_tn_batched(const float *A, const float *B, int batch, int K, int M, int N, float *C,
                                  bool accumulate, hipStream_t s);
