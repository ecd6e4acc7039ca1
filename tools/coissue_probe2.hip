// Dev probe (round 5): how do FP32 VALU instructions and v_mfma_f32_32x32x16_bf16 share a SIMD on gfx950?
//   A  same wave: an MFMA stream (two accumulator chains) with NF fillers placed in every MFMA gap
//      (scalar v_fma_f32 / packed v_pk_fma_f32 / integer), one wave per SIMD        -> cycles per MFMA
//   B  two waves per SIMD: a bare MFMA stream beside an FP-only wave (round 1's finding: they serialise)
//   C  two / three waves per SIMD, ALL running MFMA + fillers (the symmetric design)  -> cycles per MFMA, per SIMD
//   E  an MFMA + few-fillers wave beside an FP-only wave, with and without s_setprio
// Instruction order is pinned with sched_group_barrier; check with
//   /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading ...   (tools/README)
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_probe2.hip -o tools/coissue_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kIter = 64, kMfmaPerIter = 8;

// KIND 0: scalar f32 fma fillers, 1: packed f32 fma fillers (each does two lanes' worth), 2: integer fillers
template <int NF, int KIND, bool MFMA>
__device__ __forceinline__ void body(float *sink) {
  f32x16 c0 = {0}, c1 = {0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)1.f; }
  float s[8];
  f32x2 p[8];
  unsigned u[8];
  for (int j = 0; j < 8; ++j) { s[j] = threadIdx.x + j; p[j] = f32x2{(float)j, (float)threadIdx.x}; u[j] = threadIdx.x * 7 + j; }
  const float m = 1.0001f;
  const f32x2 m2 = {1.0001f, 0.9999f};
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int q = 0; q < kMfmaPerIter; ++q) {
      if (MFMA) {
        if (q & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int j = (q * NF + f) & 7;
        if (KIND == 0) s[j] = fmaf(s[j], m, 0.5f);
        else if (KIND == 1) p[j] = __builtin_elementwise_fma(p[j], m2, m2);
        else u[j] = __builtin_amdgcn_perm(u[j], u[(j + 1) & 7], 0x07060302u);
      }
    }
#pragma unroll
    for (int q = 0; q < kMfmaPerIter; ++q) {
      if (MFMA) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (NF > 0) __builtin_amdgcn_sched_group_barrier(0x002, NF, 0);
    }
  }
  float t = c0[0] + c1[3];
  for (int j = 0; j < 8; ++j) t += s[j] + p[j][0] + p[j][1] + __uint_as_float(u[j]);
  if (t == 1.2345f) sink[0] = t;
}

// a bare MFMA32 stream on NCH accumulator chains (1: every MFMA waits for its predecessor's result), GAP s_nop states behind each
template <int NCH, int GAP>
__device__ __forceinline__ void mfma_chains(float *sink) {
  f32x16 c[4];
  for (int j = 0; j < 4; ++j) c[j] = f32x16{0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x & 7); b[j] = (__bf16)1.f; }
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int q = 0; q < kMfmaPerIter; ++q) {
      c[q % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[q % NCH], 0, 0, 0);
      if (GAP == 4) asm volatile("s_nop 3");
      if (GAP == 8) asm volatile("s_nop 7");
    }
  }
  float t = 0.f;
  for (int j = 0; j < NCH; ++j) t += c[j][0] + c[j][5];
  if (t == 1.2345f) sink[1] = t;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// a producer round of flex_conv_x6: NM4 x v_mfma_f32_4x4x1 (four chains) then NV VALU (and / sub / perm triples); CL: the
// MFMA4s as one cluster (as compiled) or spread one per (NV / NM4) VALU instructions
template <int NM4, int NV, bool CL>
__device__ __forceinline__ void producer_like(float *sink, int rounds) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float s[8]; unsigned u[8];
  for (int j = 0; j < 8; ++j) { s[j] = threadIdx.x + j; u[j] = threadIdx.x * 7 + j; }
  const float a = threadIdx.x & 3, b = 1.5f;
  for (int r = 0; r < rounds; ++r) {
    if (CL) {
#pragma unroll
      for (int m = 0; m < NM4; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m & 3], 2, 1, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int j = v & 7;
        if (v % 3 == 0) u[j] = u[j] & 0xFFFF0000u;
        else if (v % 3 == 1) s[j] = s[j] - __uint_as_float(u[j]);
        else u[j] = __builtin_amdgcn_perm(u[j], __float_as_uint(s[(j + 1) & 7]), 0x07060302u);
      }
      if (NM4 > 0) __builtin_amdgcn_sched_group_barrier(0x008, NM4 > 0 ? NM4 : 1, 0);
      if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV > 0 ? NV : 1, 0);
    } else {
#pragma unroll
      for (int m = 0; m < NM4; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m & 3], 2, 1, 0);
#pragma unroll
        for (int v = m * (NV / (NM4 > 0 ? NM4 : 1)); v < (m + 1) * (NV / (NM4 > 0 ? NM4 : 1)); ++v) {
          const int j = v & 7;
          if (v % 3 == 0) u[j] = u[j] & 0xFFFF0000u;
          else if (v % 3 == 1) s[j] = s[j] - __uint_as_float(u[j]);
          else u[j] = __builtin_amdgcn_perm(u[j], __float_as_uint(s[(j + 1) & 7]), 0x07060302u);
        }
      }
#pragma unroll
      for (int m = 0; m < NM4; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NM4 > 0 && NV / (NM4 > 0 ? NM4 : 1) > 0 ? NV / NM4 : 1, 0);
      }
    }
  }
  float t = 0.f;
  for (int c = 0; c < 4; ++c) t += acc[c][0] + acc[c][3];
  for (int j = 0; j < 8; ++j) t += s[j] + __uint_as_float(u[j]);
  if (t == 1.2345f) sink[0] = t;
}

// role codes: 0 idle; 700 producer-like clustered (32 MFMA4 + 128 VALU) x 11 rounds; 701 the same, MFMA4s spread; 702 VALU only (0 + 128); 703 MFMA4 only; 100+NF scalar-filler MFMA stream; 200+NF packed; 300+NF integer; 400+NF FP-only scalar (no MFMA);
// 500+NF FP-only packed; 600+NF int-only
#define ROLE_CASES(KINDBASE, KIND, MF)                                                     \
  case KINDBASE + 0: body<0, KIND, MF>(sink); break;                                       \
  case KINDBASE + 1: body<1, KIND, MF>(sink); break;                                       \
  case KINDBASE + 2: body<2, KIND, MF>(sink); break;                                       \
  case KINDBASE + 3: body<3, KIND, MF>(sink); break;                                       \
  case KINDBASE + 4: body<4, KIND, MF>(sink); break;                                       \
  case KINDBASE + 5: body<5, KIND, MF>(sink); break;                                       \
  case KINDBASE + 6: body<6, KIND, MF>(sink); break;                                       \
  case KINDBASE + 8: body<8, KIND, MF>(sink); break;                                       \
  case KINDBASE + 12: body<12, KIND, MF>(sink); break;                                     \
  case KINDBASE + 16: body<16, KIND, MF>(sink); break;

__global__ __launch_bounds__(768) void probe(int r0, int r1, int r2, int prio0, int prio1, int prio2, long long *res, float *sink) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = wave >> 2;  // waves w, w+4, w+8 share SIMD w % 4
  const int role = slot == 0 ? r0 : (slot == 1 ? r1 : r2);
  const int prio = slot == 0 ? prio0 : (slot == 1 ? prio1 : prio2);
  if (prio == 1) __builtin_amdgcn_s_setprio(1);
  if (prio == 2) __builtin_amdgcn_s_setprio(2);
  if (prio == 3) __builtin_amdgcn_s_setprio(3);
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  __syncthreads();
  const long long t0 = clock64();
  switch (role) {
    ROLE_CASES(100, 0, true)
    ROLE_CASES(200, 1, true)
    ROLE_CASES(300, 2, true)
    ROLE_CASES(400, 0, false)
    ROLE_CASES(500, 1, false)
    ROLE_CASES(600, 2, false)
    case 800: mfma_chains<1, 0>(sink); break;
    case 801: mfma_chains<2, 4>(sink); break;
    case 802: mfma_chains<2, 8>(sink); break;
    case 803: mfma_chains<4, 0>(sink); break;
    case 700: producer_like<32, 128, true>(sink, 11); break;
    case 701: producer_like<32, 128, false>(sink, 11); break;
    case 702: producer_like<0, 128, true>(sink, 11); break;
    case 703: producer_like<32, 0, true>(sink, 11); break;
    default: break;
  }
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
    res[wave * 2] = t1 - t0;
    res[wave * 2 + 1] = (hwid >> 4) & 3;
  }
}

static long long *g_res;
static float *g_sink;

// runs one configuration; prints per-slot cycles (max over the slot's four waves) and per-MFMA / per-filler figures
static void run(const char *tag, int waves_per_simd, int r0, int r1, int r2, int p0 = 0, int p1 = 0, int p2 = 0) {
  long long h[24];
  for (int rep = 0; rep < 3; ++rep)
    hipLaunchKernelGGL(probe, dim3(256), dim3(256 * waves_per_simd), 0, 0, r0, r1, r2, p0, p1, p2, g_res, g_sink);
  hipMemcpy(h, g_res, sizeof(h), hipMemcpyDeviceToHost);
  const int roles[3] = {r0, r1, r2};
  printf("%-46s", tag);
  int n_mfma = 0, n_fill = 0;
  long long worst = 0;
  for (int s = 0; s < waves_per_simd; ++s) {
    long long mx = 0;
    for (int w = 0; w < 4; ++w) mx = h[(s * 4 + w) * 2] > mx ? h[(s * 4 + w) * 2] : mx;
    worst = mx > worst ? mx : worst;
    const int kind = roles[s] / 100, nf = roles[s] % 100;
    if (kind >= 1 && kind <= 3) n_mfma += kIter * kMfmaPerIter;
    if (kind >= 1) n_fill += kIter * kMfmaPerIter * nf;
    printf("  slot%d[role %3d] %6lld", s, roles[s], mx);
  }
  printf("  | SIMD: %6lld cyc", worst);
  if (n_mfma) printf(", %5.1f cyc/MFMA", (double)worst / n_mfma);
  if (n_fill) printf(", %4.2f cyc/filler%s", (double)worst / n_fill, n_mfma ? " (if fillers were alone)" : "");
  printf("\n");
}

int main() {
  hipMalloc(&g_res, 24 * sizeof(long long));
  hipMalloc(&g_sink, 16);
  char tag[128];
  printf("== A: one wave per SIMD, MFMA stream (2 chains) + NF fillers per gap ==\n");
  const int nfs[] = {0, 1, 2, 3, 4, 5, 6, 8, 12, 16};
  for (int kind = 1; kind <= 3; ++kind)
    for (int nf : nfs) {
      snprintf(tag, sizeof tag, "A %s NF=%d", kind == 1 ? "scalar-f32" : kind == 2 ? "packed-f32" : "integer", nf);
      run(tag, 1, kind * 100 + nf, 0, 0);
    }
  printf("== fillers alone (no MFMA), one wave per SIMD ==\n");
  for (int kind = 4; kind <= 6; ++kind) {
    snprintf(tag, sizeof tag, "alone %s NF=8", kind == 4 ? "scalar-f32" : kind == 5 ? "packed-f32" : "integer");
    run(tag, 1, kind * 100 + 8, 0, 0);
  }
  printf("== B: bare MFMA wave beside a filler-only wave (two waves per SIMD) ==\n");
  run("B mfma | scalar-f32 only", 2, 100, 408, 0);
  run("B mfma | packed-f32 only", 2, 100, 508, 0);
  run("B mfma | integer only", 2, 100, 608, 0);
  run("B mfma | scalar-f32 only, FP wave prio 3", 2, 100, 408, 0, 0, 3);
  run("B mfma prio 3 | scalar-f32 only", 2, 100, 408, 0, 3, 0);
  run("B scalar only | scalar only (two FP waves)", 2, 408, 408, 0);
  run("B mfma | mfma (two bare MFMA waves)", 2, 100, 100, 0);
  printf("== C: every wave runs MFMA + NF fillers (symmetric) ==\n");
  for (int nf : {2, 4, 6, 8, 12, 16}) {
    snprintf(tag, sizeof tag, "C 2 waves/SIMD scalar NF=%d", nf);
    run(tag, 2, 100 + nf, 100 + nf, 0);
  }
  for (int nf : {2, 4, 6, 8, 12, 16}) {
    snprintf(tag, sizeof tag, "C 3 waves/SIMD scalar NF=%d", nf);
    run(tag, 3, 100 + nf, 100 + nf, 100 + nf);
  }
  for (int nf : {2, 4, 6, 8}) {
    snprintf(tag, sizeof tag, "C 2 waves/SIMD packed NF=%d", nf);
    run(tag, 2, 200 + nf, 200 + nf, 0);
  }
  for (int nf : {4, 8, 16}) {
    snprintf(tag, sizeof tag, "C 2 waves/SIMD integer NF=%d", nf);
    run(tag, 2, 300 + nf, 300 + nf, 0);
  }
  printf("== E: MFMA + few fillers wave beside FP-only waves ==\n");
  run("E mfma+2 | scalar only", 2, 102, 408, 0);
  run("E mfma+4 | scalar only", 2, 104, 408, 0);
  run("E mfma+4 | scalar only prio 3", 2, 104, 408, 0, 0, 3);
  run("E mfma+4 | scalar only | scalar only", 3, 104, 408, 408);
  run("E mfma+4 | integer only | integer only", 3, 104, 608, 608);
  run("E mfma+0 | scalar only | scalar only", 3, 100, 408, 408);
  run("E mfma+0 | scalar only | scalar only, prio 3", 3, 100, 408, 408, 0, 3, 3);
  printf("== F: flex_conv_x6 in miniature: one MFMA32 wave (512 MFMAs = 10.7 tiles) + two producer-like waves (11 rounds each) per SIMD ==\n");
  run("F producer-like alone (clustered)", 1, 700, 0, 0);
  run("F producer-like alone (spread)", 1, 701, 0, 0);
  run("F VALU part alone", 1, 702, 0, 0);
  run("F MFMA4 part alone", 1, 703, 0, 0);
  run("F two producer-like waves", 2, 700, 700, 0);
  run("F mfma32 | producer | producer (clustered)", 3, 100, 700, 700);
  run("F mfma32 | producer | producer (spread)", 3, 100, 701, 701);
  run("F mfma32 | VALU-only | VALU-only", 3, 100, 702, 702);
  run("F mfma32 | MFMA4-only | MFMA4-only", 3, 100, 703, 703);
  run("F mfma32+2 | producer | producer (clustered)", 3, 102, 700, 700);
  run("F mfma32 | producer | producer, producers prio 3", 3, 100, 700, 700, 0, 3, 3);
  run("F mfma32 prio 3 | producer | producer", 3, 100, 700, 700, 3, 0, 0);
  printf("== G: can the MFMA32 wave leave room for another wave's MFMA4s? ==\n");
  run("G 1-chain mfma32 alone", 1, 800, 0, 0);
  run("G 1-chain mfma32 | producer | producer", 3, 800, 700, 700);
  run("G 1-chain mfma32 | producer | producer (spread)", 3, 800, 701, 701);
  run("G 1-chain mfma32 | MFMA4-only | MFMA4-only", 3, 800, 703, 703);
  run("G 2-chain + s_nop 4 alone", 1, 801, 0, 0);
  run("G 2-chain + s_nop 4 | producer | producer", 3, 801, 700, 700);
  run("G 2-chain + s_nop 8 alone", 1, 802, 0, 0);
  run("G 2-chain + s_nop 8 | producer | producer", 3, 802, 700, 700);
  run("G 4-chain mfma32 | producer | producer", 3, 803, 700, 700);
  return 0;
}
