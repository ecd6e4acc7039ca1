// Dev probe: do VALU instructions of one wave issue while ANOTHER wave on the same SIMD runs MFMAs?
// 8-wave workgroup; waves 0-3 run a VALU loop, waves 4-7 an MFMA loop (or idle).  Prints the SIMD id of every
// wave (HW_ID) and the VALU / MFMA loop times in the four combinations.
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o tools/coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void probe(int valu_on, int mfma_on, int mfma_kind, long long *res, float *sink) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  __syncthreads();
  const long long t0 = clock64();
  if (wave < 4) {
    if (valu_on == 2) {  // integer VALU only
      unsigned b0 = threadIdx.x, b1 = 1, b2 = 2, b3 = 3, b4 = 4, b5 = 5, b6 = 6, b7 = 7;
      for (int i = 0; i < 256; ++i) {
        b0 = (b0 & 0xFFFF0000u) + b1; b1 = __builtin_amdgcn_perm(b1, b2, 0x07060302u); b2 = (b2 ^ b3) + 3; b3 = (b3 << 1) | b4;
        b4 = (b4 & 0xFFFF0000u) + b5; b5 = __builtin_amdgcn_perm(b5, b6, 0x07060302u); b6 = (b6 ^ b7) + 3; b7 = (b7 << 1) | b0;
      }
      if (b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 12345u) sink[0] = b0;
    } else if (valu_on == 3) {  // scalar f32 only
      float b0 = threadIdx.x, b1 = 1.f, b2 = 2.f, b3 = 3.f, b4 = 4.f, b5 = 5.f, b6 = 6.f, b7 = 7.f;
      for (int i = 0; i < 256; ++i) {
        b0 = fmaf(b0, 1.0001f, b1); b1 = fmaf(b1, 0.9999f, b2); b2 = fmaf(b2, 1.0001f, b3); b3 = fmaf(b3, 0.9999f, b4);
        b4 = fmaf(b4, 1.0001f, b5); b5 = fmaf(b5, 0.9999f, b6); b6 = fmaf(b6, 1.0001f, b7); b7 = fmaf(b7, 0.9999f, b0);
      }
      if (b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 1.234f) sink[0] = b0;
    } else if (valu_on == 4) {  // LDS traffic only
      __shared__ float4 buf[512];
      float4 v = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
      for (int i = 0; i < 256; ++i) {
        buf[(threadIdx.x + i) & 255] = v;
        const float4 q = buf[(threadIdx.x + 7 * i + 64) & 255];
        v.x += q.y;
      }
      if (v.x == 1.234f) sink[0] = v.x;
    } else if (valu_on) {
      f32x2 a0 = {1.f, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f};
      float b0 = threadIdx.x, b1 = 1.f, b2 = 2.f, b3 = 3.f;
      const f32x2 m = {1.0001f, 0.9999f};
      for (int i = 0; i < 256; ++i) {  // 8 independent chains, 2048 VALU instrs
        a0 = __builtin_elementwise_fma(a0, m, a1); a1 = __builtin_elementwise_fma(a1, m, a2);
        a2 = __builtin_elementwise_fma(a2, m, a3); a3 = __builtin_elementwise_fma(a3, m, a0);
        b0 = fmaf(b0, 1.0001f, b1); b1 = fmaf(b1, 0.9999f, b2); b2 = fmaf(b2, 1.0001f, b3); b3 = fmaf(b3, 0.9999f, b0);
      }
      if (b0 + b1 + b2 + b3 + a0[0] + a1[1] + a2[0] + a3[1] == 1.234f) sink[0] = b0;
    }
  } else if (mfma_on) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    if (mfma_kind == 0) {
      bf16x8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)1.f; }
      for (int i = 0; i < 64; ++i) {  // 256 MFMAs x 32 cycles
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
    } else {
      float a = threadIdx.x & 7, b = 1.f;
      for (int i = 0; i < 32; ++i) {  // 128 MFMAs x 64 cycles
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
      }
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 1.234f) sink[1] = c0[0];
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
    res[wave * 2] = t1 - t0;
    res[wave * 2 + 1] = hwid;
  }
}

int main() {
  long long *res, h[16];
  float *sink;
  hipMalloc(&res, sizeof(h));
  hipMalloc(&sink, 16);
  for (int kind = 0; kind < 2; ++kind)
    for (int mode = 0; mode < 9; ++mode) {
      const int valu_on = mode == 1 ? 0 : (mode < 3 ? 1 : 2 + (mode - 3) / 2), mfma_on = mode < 3 ? mode != 0 : (mode - 3) % 2;
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, valu_on, mfma_on, kind, res, sink);
      hipMemcpy(h, res, sizeof(h), hipMemcpyDeviceToHost);
      printf("%s valu=%d mfma=%d:", kind ? "f32 32x32x2 " : "bf16 32x32x16", valu_on, mfma_on);
      for (int w = 0; w < 8; ++w) printf("  w%d[simd %lld] %lld", w, (h[w * 2 + 1] >> 4) & 3, h[w * 2]);
      printf("\n");
    }
  return 0;
}
