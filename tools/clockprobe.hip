// Dev probe: effective shader clock of a low-occupancy kernel (s_memtime ticks vs the 100 MHz wall clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(long long *out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) { a = fmaf(a, b, 0.5f); a = fmaf(a, b, -0.5f); }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = c1 - c0; out[blockIdx.x * 3 + 1] = w1 - w0; out[blockIdx.x * 3 + 2] = (long long)a; }
}
int main() {
  long long *d; hipMalloc(&d, 3 * 8 * 4096);
  for (int blocks : {1, 8, 256, 2048}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(spin, dim3(blocks), dim3(1024), 0, 0, d, 200000);
      hipDeviceSynchronize();
    }
    long long h[3]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int wrate = 0; hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
    printf("blocks %4d: %lld shader ticks, %lld wall ticks (wall rate %d kHz) -> %.0f MHz\n", blocks, h[0], h[1], wrate,
           (double)h[0] / ((double)h[1] / (wrate * 1e3)) / 1e6);
  }
  return 0;
}
