// Dev: which SIMD does wave w of a 1024-thread workgroup run on?  (HW_REG_HW_ID: wave_id [3:0], simd_id [5:4], cu_id [11:8])
// hipcc --offload-arch=gfx950 -O2 tools/simd_map.hip -o tools/simd_map && tools/simd_map
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned *d, h[64];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(4), dim3(1024), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) {
    printf("block %d (cu %u): simd of waves 0..15:", b, (h[b * 16] >> 8) & 15);
    for (int w = 0; w < 16; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
    printf("\n");
  }
  return 0;
}
