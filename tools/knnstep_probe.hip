// Dev probe: cost of the kNN screening step (32 candidates) by stage, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma clang fp contract(off)
#define REP 256
__device__ __forceinline__ void dist8(const float *g0, f32x2 qx2, f32x2 qy2, f32x2 qz2, f32x2 (&s)[4]) {
  const f32x4 a0 = *reinterpret_cast<const f32x4 *>(g0), a1 = *reinterpret_cast<const f32x4 *>(g0 + 4),
              a2 = *reinterpret_cast<const f32x4 *>(g0 + 8), b0 = *reinterpret_cast<const f32x4 *>(g0 + 12),
              b1 = *reinterpret_cast<const f32x4 *>(g0 + 16), b2 = *reinterpret_cast<const f32x4 *>(g0 + 20);
  { const f32x2 dx = f32x2{a0[0], a0[1]} - qx2, dy = f32x2{a0[2], a0[3]} - qy2, dz = f32x2{a1[0], a1[1]} - qz2;
    s[0] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx)); }
  { const f32x2 dx = f32x2{a1[2], a1[3]} - qx2, dy = f32x2{a2[0], a2[1]} - qy2, dz = f32x2{a2[2], a2[3]} - qz2;
    s[1] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx)); }
  { const f32x2 dx = f32x2{b0[0], b0[1]} - qx2, dy = f32x2{b0[2], b0[3]} - qy2, dz = f32x2{b1[0], b1[1]} - qz2;
    s[2] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx)); }
  { const f32x2 dx = f32x2{b1[2], b1[3]} - qx2, dy = f32x2{b2[0], b2[1]} - qy2, dz = f32x2{b2[2], b2[3]} - qz2;
    s[3] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx)); }
}
__device__ __forceinline__ void dist8_scalar(const float *g0, float qx, float qy, float qz, float (&s)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float dx = g0[t * 4] - qx, dy = g0[t * 4 + 1] - qy, dz = g0[t * 4 + 2] - qz;
    s[t] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
  }
}
__global__ void probe_smem(long long *out, float *sink, const float4 *__restrict__ cand) {
  const float q = threadIdx.x * 0.01f;
  const float qx = q, qy = q * 2, qz = q * 3;
  float bound = -1.f;
  int cnt = 0;
  long long t0 = clock64();
  for (int r = 0; r < REP; ++r) {
    const int j = (r * 32) & 1023;
    float m = 1e30f;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
      const float4 c = cand[j + t];  // uniform address -> scalar load
      const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
      m = fminf(m, fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
    }
    if (__any(m <= bound)) { cnt++; bound = m; }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[5] = (t1 - t0);
  sink[blockIdx.x * blockDim.x + threadIdx.x] = cnt + bound;
}
template <int TEST>
__global__ void probe(long long *out, float *sink) {
  __shared__ __attribute__((aligned(16))) float s_c[1024 * 4];
  for (int i = threadIdx.x; i < 1024 * 4; i += blockDim.x) s_c[i] = (i * 37 % 1000) * 0.001f;
  __syncthreads();
  const float q = threadIdx.x * 0.01f;
  const f32x2 qx2 = {q, q}, qy2 = {q * 2, q * 2}, qz2 = {q * 3, q * 3};
  float bound = -1.f, accum = 0.f;
  int cnt = 0;
  long long t0 = clock64();
  for (int r = 0; r < REP; ++r) {
    const int j = (r * 32) & 1023;
    if (TEST <= 3) {
      f32x2 s[4][4]; float mn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        dist8(s_c + ((j >> 2) + 2 * u) * 12 % 3000, qx2, qy2, qz2, s[u]);
        if (TEST >= 1) mn[u] = fminf(fminf(fminf(s[u][0][0], s[u][0][1]), fminf(s[u][1][0], s[u][1][1])), fminf(fminf(s[u][2][0], s[u][2][1]), fminf(s[u][3][0], s[u][3][1])));
        else mn[u] = s[u][0][0] + s[u][1][1] + s[u][2][0] + s[u][3][1] + s[u][0][1] + s[u][1][0] + s[u][2][1] + s[u][3][0];
      }
      const float m = fminf(fminf(mn[0], mn[1]), fminf(mn[2], mn[3]));
      if (TEST == 2) { if (__any(m <= bound)) { cnt++; bound = m; } }
      if (TEST == 3) { if (m <= bound) { cnt++; bound = m; } }
      if (TEST <= 1) accum += m;
    } else {  // scalar (unpacked) math, float4 per candidate layout
      float s[8]; float m = 1e30f;
#pragma unroll
      for (int u = 0; u < 4; ++u) { dist8_scalar(s_c + ((j + 8 * u) & 1023) * 4, q, q * 2, q * 3, s);
#pragma unroll
        for (int t = 0; t < 8; ++t) m = fminf(m, s[t]); }
      if (__any(m <= bound)) { cnt++; bound = m; }
    }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[TEST] = (t1 - t0);
  sink[blockIdx.x * blockDim.x + threadIdx.x] = accum + cnt + bound;
}
int main() {
  long long *d; float *sink; hipMalloc(&d, 64 * 8); hipMalloc(&sink, 4 * 1024 * 64);
  const char *names[] = {"LDS reads + packed math (sum)", "+ min tree", "+ __any branch", "per-lane branch instead of __any", "scalar math + float4 records + __any", "candidates by scalar loads (s_load) + scalar-operand VALU"};
  hipLaunchKernelGGL(probe<0>, dim3(1), dim3(256), 0, 0, d, sink);
  hipLaunchKernelGGL(probe<1>, dim3(1), dim3(256), 0, 0, d, sink);
  hipLaunchKernelGGL(probe<2>, dim3(1), dim3(256), 0, 0, d, sink);
  hipLaunchKernelGGL(probe<3>, dim3(1), dim3(256), 0, 0, d, sink);
  hipLaunchKernelGGL(probe<4>, dim3(1), dim3(256), 0, 0, d, sink);
  float4 *cand; hipMalloc(&cand, 1100 * 16); hipMemset(cand, 0, 1100 * 16);
  hipLaunchKernelGGL(probe_smem, dim3(1), dim3(256), 0, 0, d, sink, cand);
  hipLaunchKernelGGL(probe_smem, dim3(1), dim3(256), 0, 0, d, sink, cand);
  hipDeviceSynchronize();
  long long h[6]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 6; ++i) printf("  %-42s %7.1f cycles per 32 candidates\n", names[i], (double)h[i] / REP);
  return 0;
}
