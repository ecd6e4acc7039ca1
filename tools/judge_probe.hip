// Dev: cycles per pick of the FPS judge loop (one wave, 60-entry pool) in a few formulations.
// hipcc --offload-arch=gfx950 -O3 -Idh3d_amd/csrc -Iinclude tools/judge_probe.hip -o tools/judge_probe && tools/judge_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <limits.h>
#include "wave_ops.h"
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define DH3D_LDS __attribute__((address_space(3)))
__device__ __forceinline__ void lds_vstore(int *p, int v) { *(volatile DH3D_LDS int *)p = v; }
__device__ __forceinline__ void lds_vstore4(f32x4 *p, f32x4 v) { *(volatile DH3D_LDS f32x4 *)p = v; }

// VARIANT: 0 = full loop (reduce, tie check, stores, update); 1 = no LDS stores; 2 = no stores, no tie check;
//          3 = reduce only (+update with lane 0); 4 = like 0 but stores by lane 0 from SGPR copies;
//          5 = every lane stores (losers into a dummy slot of their own), ring entry + head, popcount tie check
template <int VARIANT>
__global__ __launch_bounds__(64) void k(const float4 *pool, int iters, long long *cyc, int *sink) {
  __shared__ f32x4 s_ring[64];
  __shared__ int s_out[4096];
  __shared__ int s_head[4];
  __shared__ f32x4 s_dummy4[64];
  __shared__ int s_dummy[64];
  const int lane = threadIdx.x;
  const float4 e = pool[lane];
  const float cx = e.x, cy = e.y, cz = e.z;
  const int cidx = lane * 7, ckey = lane * 13;
  float cv = lane < 60 ? 10.f + e.w : -2.f;
  int r = 1, rbe = -1;
  const int RB = -1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long t0 = clock64();
  const int rend = r + iters;
  while (true) {
    const int vmax = __float_as_int(wave_max_f32(cv));
    if (!(vmax > rbe)) break;
    rbe = RB;
    int l = 0;
    if (VARIANT != 3) {
      const unsigned long long hit = __ballot(__float_as_int(cv) == vmax);
      l = __builtin_ctzll(hit);
      if (VARIANT != 2 && __builtin_expect(VARIANT == 5 ? __popcll(hit) != 1 : (hit & (hit - 1ull)) != 0ull, 0)) {
        const int kmin = wave_min_i32(__float_as_int(cv) == vmax ? ckey : INT_MAX);
        l = __builtin_ctzll(__ballot(__float_as_int(cv) == vmax && ckey == kmin));
      }
    }
    if (VARIANT == 5) {
      const bool win = lane == l;
      lds_vstore4(win ? &s_ring[r & 63] : &s_dummy4[lane], f32x4{cx, cy, cz, __int_as_float(cidx)});
      lds_vstore(win ? &s_head[0] : &s_dummy[lane], r + 1);
    }
    if (VARIANT == 0) {
      if (lane == l) {
        lds_vstore4(&s_ring[r & 63], f32x4{cx, cy, cz, __int_as_float(cidx)});
        lds_vstore(&s_out[r & 4095], cidx);
        lds_vstore(&s_head[0], r + 1);
      }
    }
    const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), l));
    const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), l));
    const float z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), l));
    if (VARIANT == 4) {
      const int i1 = __builtin_amdgcn_readlane(cidx, l);
      if (lane == 0) {
        lds_vstore4(&s_ring[r & 63], f32x4{x1, y1, z1, __int_as_float(i1)});
        lds_vstore(&s_out[r & 4095], i1);
        lds_vstore(&s_head[0], r + 1);
      }
    }
    const float dx = cx - x1, dy = cy - y1, dz = cz - z1;
    const float d = __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
    // keep the pool alive: the picked entry is re-armed instead of dropping to 0 (timing only)
    cv = __builtin_fminf(d + 9.f, cv) - 1e-3f;
    if (++r == rend) break;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  if (lane == 0) { cyc[VARIANT] = t1 - t0; sink[VARIANT] = r + s_out[5] + s_head[0]; }
  if (cv == 123.f) sink[8] = 1;
}
int main() {
  float4 h[64];
  srand(1);
  for (int i = 0; i < 64; ++i) h[i] = make_float4(rand() / (float)RAND_MAX, rand() / (float)RAND_MAX, rand() / (float)RAND_MAX, rand() / (float)RAND_MAX);
  float4 *d; long long *c; int *s;
  (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&c, 64); (void)hipMalloc(&s, 64);
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d, iters, c, s);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d, iters, c, s);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d, iters, c, s);
    hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d, iters, c, s);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, d, iters, c, s);
    hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, d, iters, c, s);
    (void)hipDeviceSynchronize();
  }
  long long hc[8]; int hs[16];
  (void)hipMemcpy(hc, c, 64, hipMemcpyDeviceToHost); (void)hipMemcpy(hs, s, 64, hipMemcpyDeviceToHost);
  const char *names[6] = {"full (winner lane stores)", "no LDS stores", "no stores, no tie check", "reduce + update only", "full (lane 0 stores)", "all lanes store, popc tie"};
  for (int v = 0; v < 6; ++v) printf("%-28s %7.1f cycles per pick (%d picks)\n", names[v], (double)hc[v] / (hs[v] > 1 ? iters : 1), iters);
  return 0;
}
