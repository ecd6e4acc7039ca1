// Dev probe: what the f32 MFMA pipe delivers for the tile loop shapes used in mfma_gemm.h.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NT, bool LDS_A, bool GLOBAL_B>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ w, float *sink, int KB, int reps) {
  __shared__ __attribute__((aligned(16))) float s_A[64 * 260];
  for (int i = threadIdx.x; i < 64 * 260; i += 256) s_A[i] = i * 1e-4f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[NT];
  for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float *aptr = s_A + ((wave & 1) * 32 + (lane & 31)) * 260 + 4 * (lane >> 5);
  const f32x4 *wp = reinterpret_cast<const f32x4 *>(w) + lane;
  f32x4 b = {1.f, 2.f, 3.f, 4.f}, a = {0.5f, 0.25f, 0.125f, 1.f};
  for (int rep = 0; rep < reps; ++rep) {
    for (int kb = 0; kb < KB; ++kb) {
      if (LDS_A) a = *reinterpret_cast<const f32x4 *>(aptr + (kb & 31) * 8);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (GLOBAL_B) b = wp[(size_t)(((wave >> 1) + 2 * j) * KB + kb) * 64];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc[j], 0, 0, 0);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc[j], 0, 0, 0);
      }
    }
  }
  float s = 0; for (int j = 0; j < NT; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NT, bool LA, bool GB>
void run(const char *name, const float *w, float *sink, int blocks) {
  const int KB = 32, reps = 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NT, LA, GB>), dim3(blocks), dim3(256), 0, 0, w, sink, KB, reps);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NT, LA, GB>), dim3(blocks), dim3(256), 0, 0, w, sink, KB, reps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * NT * KB * reps * 4 * 4096;
  printf("%-44s blocks %5d  %7.1f us  %6.1f TF/s\n", name, blocks, ms * 1e3, flop / ms / 1e9);
}
int main() {
  float *w, *sink; hipMalloc(&w, 64 << 20); hipMemset(w, 0, 64 << 20); hipMalloc(&sink, 4096 * 256 * 4);
  for (int blocks : {256, 512, 1024}) {
    run<1, false, false>("NT=1 registers only", w, sink, blocks);
    run<2, false, false>("NT=2 registers only", w, sink, blocks);
    run<2, true, false>("NT=2 A from LDS", w, sink, blocks);
    run<2, true, true>("NT=2 A from LDS, B from L2 (no prefetch)", w, sink, blocks);
    run<4, true, true>("NT=4 A from LDS, B from L2 (no prefetch)", w, sink, blocks);
  }
  return 0;
}
