// Dev probe: what a taken branch costs a wave (forward jumps over code blocks of different sizes), 1 and 16 waves.
// hipcc --offload-arch=gfx950 -O3 tools/branch_cost.hip -o tools/branch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 2000
// one "hop": a uniform branch that skips SKIP filler instructions when taken
#define FILL4 "v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n"
#define FILL16 FILL4 FILL4 FILL4 FILL4
#define FILL64 FILL16 FILL16 FILL16 FILL16
#define FILL256 FILL64 FILL64 FILL64 FILL64
template <int KIND>
__global__ void probe(long long *out, int *sink, int take) {
  int v = threadIdx.x;
  const long long t0 = clock64();
  for (int r = 0; r < REP; ++r) {
    // 8 hops per iteration; `take` is uniform (SGPR), != 0 -> every hop is taken
    if (KIND == 0) {
#pragma unroll
      for (int h = 0; h < 8; ++h)
        asm volatile("s_cmp_lg_u32 %1, 0\n s_cbranch_scc1 1f\n" FILL16 "1:\n v_add_u32 %0, %0, 1\n" : "+v"(v) : "s"(take) : "scc");
    }
    if (KIND == 1) {
#pragma unroll
      for (int h = 0; h < 8; ++h)
        asm volatile("s_cmp_lg_u32 %1, 0\n s_cbranch_scc1 1f\n" FILL64 "1:\n v_add_u32 %0, %0, 1\n" : "+v"(v) : "s"(take) : "scc");
    }
    if (KIND == 2) {
#pragma unroll
      for (int h = 0; h < 8; ++h)
        asm volatile("s_cmp_lg_u32 %1, 0\n s_cbranch_scc1 1f\n" FILL256 "1:\n v_add_u32 %0, %0, 1\n" : "+v"(v) : "s"(take) : "scc");
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) out[0] = t1 - t0;
  sink[threadIdx.x] = v;
}
template <int K> void run(const char *name, int threads, int take, long long *d, int *sink) {
  hipLaunchKernelGGL((probe<K>), dim3(1), dim3(threads), 0, 0, d, sink, take);
  hipLaunchKernelGGL((probe<K>), dim3(1), dim3(threads), 0, 0, d, sink, take);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("  %-26s threads %4d %s: %7.1f cycles per hop\n", name, threads, take ? "taken    " : "not taken", (double)h / REP / 8);
}
int main() {
  long long *d; int *sink; hipMalloc(&d, 64); hipMalloc(&sink, 4096);
  for (int threads : {64, 1024})
    for (int take : {0, 1}) {
      run<0>("skip 16 instrs (128 B)", threads, take, d, sink);
      run<1>("skip 64 instrs (512 B)", threads, take, d, sink);
      run<2>("skip 256 instrs (2 KB)", threads, take, d, sink);
    }
  return 0;
}
