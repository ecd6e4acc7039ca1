// Dev probe: cycle cost of the primitives on the FPS per-round chain, at 1 / 4 / 16 waves per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../dh3d_amd/csrc/wave_ops.h"
#define REP 200
template <int TEST>
__global__ void probe(long long *out, float *sink) {
  __shared__ float s_a[64];
  __shared__ int s_i[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += blockDim.x) s_i[i] = (i * 37 + 11) & 4095;
  if (tid < 64) s_a[tid] = tid;
  __syncthreads();
  float v = tid * 0.37f;
  int idx = tid & 4095;
  long long t0 = clock64();
  for (int r = 0; r < REP; ++r) {
    if (TEST == 0) { v = wave_max_f32(v) + lane; }
    if (TEST == 1) { unsigned long long h = __ballot(v > (float)(r & 7)); int l = __builtin_ctzll(h | (1ull << 63)); v += __builtin_amdgcn_readlane(__float_as_int(v), l) * 1e-30f; }
    if (TEST == 2) { if (lane == 0) s_a[tid >> 6] = v; __syncthreads(); v += s_a[(lane & 15) % (blockDim.x >> 6)]; }
    if (TEST == 3) { idx = s_i[idx]; }
    if (TEST == 4) { __syncthreads(); }
    if (TEST == 5) { v = fmaf(v, 1.0001f, 0.5f); v = fmaf(v, 0.9999f, -0.5f); v = fmaf(v, 1.0001f, 0.5f); v = fmaf(v, 0.9999f, -0.5f); }
    if (TEST == 6) { v = row16_max_f32(v) + lane; }
  }
  long long t1 = clock64();
  if (tid == 0) out[TEST] = (t1 - t0) / REP;
  sink[blockIdx.x * blockDim.x + tid] = v + idx;
}
int main() {
  long long *d; float *sink; hipMalloc(&d, 64 * 8); hipMalloc(&sink, 4 * 1024 * 8);
  const char *names[] = {"wave_max_f32 (6 DPP + readlane) + add", "ballot + ctz + readlane", "lds write / barrier / lds read",
                         "dependent ds_read", "s_barrier only", "4 dependent v_fma", "row16_max (4 DPP) + add"};
  for (int threads : {64, 256, 1024}) {
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(probe<2>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(probe<3>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(probe<4>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(probe<5>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(probe<6>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("== %d threads (%d waves)\n", threads, threads / 64);
    for (int i = 0; i < 7; ++i) printf("  %-42s %5lld cycles\n", names[i], h[i]);
  }
  return 0;
}
