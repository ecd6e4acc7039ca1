// Dev probe: throughput of uniform-address (broadcast) LDS reads vs per-lane reads, 1 wave and 4 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP 256
template <int TEST>
__global__ void probe(long long *out, float *sink) {
  __shared__ __attribute__((aligned(16))) float s[4096 * 4];
  for (int i = threadIdx.x; i < 4096 * 4; i += blockDim.x) s[i] = i * 0.001f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc = {0, 0, 0, 0};
  float a1 = 0;
  long long t0 = clock64();
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = (r * 8 + u) & 4095;
      if (TEST == 0) acc += *reinterpret_cast<const f32x4 *>(s + j * 4);                       // uniform b128
      if (TEST == 1) acc += *reinterpret_cast<const f32x4 *>(s + ((j + lane) & 4095) * 4);     // per-lane b128, consecutive
      if (TEST == 2) a1 += s[j];                                                               // uniform b32
      if (TEST == 3) a1 += s[(j + lane) & 4095];                                               // per-lane b32
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[TEST] = (t1 - t0);
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + a1;
}
int main() {
  long long *d; float *sink; hipMalloc(&d, 64 * 8); hipMalloc(&sink, 4 * 1024 * 64);
  const char *names[] = {"uniform-address ds_read_b128", "per-lane ds_read_b128", "uniform-address ds_read_b32", "per-lane ds_read_b32"};
  for (int threads : {64, 256}) {
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(probe<2>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipLaunchKernelGGL(probe<3>, dim3(1), dim3(threads), 0, 0, d, sink);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("== %d threads\n", threads);
    for (int i = 0; i < 4; ++i) printf("  %-34s %6.1f cycles per read instruction\n", names[i], (double)h[i] / (REP * 8));
  }
  return 0;
}
