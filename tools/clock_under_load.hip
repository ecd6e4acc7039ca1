// Dev probe: real shader clock of a lightly loaded GPU (FPS occupies 8 of 256 CUs): a dependent v_fma chain
// (4 cycles per instruction) timed against the 100 MHz wall clock, with 1 / 8 / 256 / 2048 workgroups resident.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void chain(long long *out, float *sink, int reps) {
  float a = threadIdx.x * 1e-9f, b = 1.000001f;
  const long long w0 = wall_clock64();
  const long long c0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 64; ++u) a = __builtin_fmaf(a, b, 1e-7f);
  }
  const long long c1 = clock64();
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; }
  if (a == 123.f) sink[0] = a;
}
int main() {
  long long *d; float *sink; hipMalloc(&d, 64); hipMalloc(&sink, 64);
  const int reps = 4096;
  for (int threads : {64, 1024})
    for (int blocks : {1, 8, 256, 2048}) {
      for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(chain, dim3(blocks), dim3(threads), 0, 0, d, sink, reps);
      hipDeviceSynchronize();
      long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      const double ns = h[0] * 10.0, n = (double)reps * 64;
      printf("blocks %5d x %4d threads: %.2f ns per dependent fma; clock64 ticks per fma %.2f; clock64 rate %.0f MHz\n", blocks,
             threads, ns / n, h[1] / n, h[1] / ns * 1e3);
    }
  return 0;
}
