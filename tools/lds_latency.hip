// Dev probe: LDS round-trip latency as the FPS kernels see it (one workgroup, barrier-separated dependent reads).
// hipcc --offload-arch=gfx950 -O3 tools/lds_latency.hip -o tools/lds_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP 300
template <bool BARRIER, bool ALLREAD, int KIND>
__global__ void probe(long long *out, int *sink, int nwords) {
  extern __shared__ __attribute__((aligned(16))) int s_mem[];
  for (int i = threadIdx.x; i < nwords; i += blockDim.x) s_mem[i] = (i * 7 + 3) % nwords;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  int idx = threadIdx.x & 63;
  long long sum = 0;
  for (int r = 0; r < REP; ++r) {
    if (BARRIER) __syncthreads();
    if (ALLREAD || wave == 0) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const long long t0 = clock64();
      if (KIND == 0) idx = s_mem[idx];                                  // dependent ds_read_b32
      if (KIND == 1) idx = atomicMin(&s_mem[(idx & 15) + 64], idx) + idx % nwords;  // returning atomic
      if (KIND == 2) { atomicMin(&s_mem[(idx & 15) + 64], idx); }        // no-return atomic, then wait
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      asm volatile("" :: "v"(idx));
      const long long t1 = clock64();
      sum += t1 - t0;
      idx = ((unsigned)idx) % (unsigned)nwords;
    }
  }
  if (threadIdx.x == 0) out[0] = sum;
  sink[threadIdx.x] = idx;
}
template <bool B, bool A, int K>
void run(const char *name, int threads, int ldsbytes, long long *d, int *sink) {
  hipFuncSetAttribute((const void *)probe<B, A, K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((probe<B, A, K>), dim3(1), dim3(threads), ldsbytes, 0, d, sink, ldsbytes / 4);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("  %-44s threads %4d lds %6d B: %6.1f cycles (incl. ~2 clock reads)\n", name, threads, ldsbytes, (double)h / REP);
}
int main() {
  long long *d; int *sink; hipMalloc(&d, 64); hipMalloc(&sink, 4096);
  for (int lds : {4096, 65536, 143360})
    for (int threads : {64, 256, 1024}) {
      run<false, true, 0>("read, no barrier, all waves", threads, lds, d, sink);
      run<true, true, 0>("read after barrier, all waves", threads, lds, d, sink);
      run<true, false, 0>("read after barrier, wave 0 only", threads, lds, d, sink);
      run<true, true, 1>("returning atomic after barrier, all waves", threads, lds, d, sink);
      run<true, false, 2>("no-return atomic + wait, wave 0 only", threads, lds, d, sink);
    }
  return 0;
}
