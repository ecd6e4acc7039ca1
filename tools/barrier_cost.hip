// Dev probe: cost of a workgroup barrier round (16 waves on one CU), alone and with the LDS traffic of an FPS sync.
// hipcc --offload-arch=gfx950 -O3 tools/barrier_cost.hip -o tools/barrier_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 1000
template <int KIND>
__global__ void probe(long long *out, int *sink) {
  __shared__ __attribute__((aligned(16))) int s[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (i * 7 + 3) & 1023;
  __syncthreads();
  int idx = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long w0 = wall_clock64();
  const long long t0 = clock64();
  for (int r = 0; r < REP; ++r) {
    if (KIND == 0) { __syncthreads(); }
    if (KIND == 1) { __syncthreads(); idx = s[idx]; }                       // barrier + dependent read
    if (KIND == 2) { __syncthreads(); idx = s[idx]; __syncthreads(); idx = s[idx]; }
    if (KIND == 3) { if ((threadIdx.x & 63) == 0) s[wave] = idx + r; __syncthreads(); idx = s[idx & 15] & 1023; }  // publish+read
    if (KIND == 4) { asm volatile("s_barrier"); }                          // raw barrier, no waitcnt
    if (KIND == 5) { idx = s[idx]; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm volatile("" :: "v"(idx)); }  // read only
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
  sink[threadIdx.x] = idx;
}
template <int K> void run(const char *name, int threads, long long *d, int *sink) {
  hipLaunchKernelGGL((probe<K>), dim3(1), dim3(threads), 0, 0, d, sink);
  hipDeviceSynchronize();
  long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("  %-44s threads %4d: %6.1f cycles / iteration  (%.0f ns; clock64 runs at %.0f MHz)\n", name, threads,
         (double)h[0] / REP, h[1] * 10.0 / REP, h[0] / (h[1] * 10.0) * 1000.0);
}
int main() {
  long long *d; int *sink; hipMalloc(&d, 64); hipMalloc(&sink, 4096);
  for (int threads : {64, 256, 512, 1024}) {
    run<0>("__syncthreads()", threads, d, sink);
    run<4>("raw s_barrier", threads, d, sink);
    run<5>("dependent LDS read", threads, d, sink);
    run<1>("__syncthreads + dependent LDS read", threads, d, sink);
    run<2>("2 x (barrier + read)", threads, d, sink);
    run<3>("lane0 write, barrier, read", threads, d, sink);
  }
  return 0;
}
