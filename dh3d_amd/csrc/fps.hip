// Farthest point sampling for gfx950 -- replaces farthestpointsamplingKernel
// (tf_ops/sampling/tf_sampling_g.cu:105-170; launcher :203-205, op tf_sampling.cpp:95-123).
//
// FPS is a chain of m dependent rounds; the only outer parallelism is the batch.  One 1024-lane
// workgroup (16 waves, one CU) owns a cloud.  Each lane keeps its PPT points AND their running
// min-distances in registers for the whole kernel (the reference re-reads a [32,n] global scratch
// every round), so a round is: PPT fused distance updates per lane -> wave argmax by DPP-free
// xor-shuffles -> 16 partial winners through LDS (one barrier per round, double-buffered) ->
// every wave re-derives the same winner and reads its coordinates from the LDS copy of the cloud.
//
// Bit-exactness (compiled with -ffp-contract=off):
//   d    = fma(dz,dz, fma(dx,dx, dy*dy))      -- nvcc/LLVM contraction of (:141); see DESIGN.md
//   td   = min(d, td)  starting from 1e38     (:118,143-145)
//   pick = max td; ties -> smallest (k % 512, k / 512): the reference's 512 threads each scan
//          k = tid, tid+512, ... with strict '>' (:146-149) and its tree keeps the lower slot on
//          ties (:158-161).  Encoded here as key(k) = ((k & 511) << 16) | (k >> 9), smaller wins.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;

__device__ __forceinline__ int fps_key(int k) { return ((k & 511) << 16) | (k >> 9); }
__device__ __forceinline__ int fps_unkey(int key) { return ((key & 0xffff) << 9) | (key >> 16); }

// (value, key) argmax over a wave; every lane returns the winner.
__device__ __forceinline__ void wave_argmax(float &val, int &key) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(val, off, 64);
    const int ok = __shfl_xor(key, off, 64);
    const bool take = (ov > val) || (ov == val && ok < key);
    val = take ? ov : val;
    key = take ? ok : key;
  }
}

template <int PPT, bool LDS_COORDS>
__global__ __launch_bounds__(kThreads) void fps_kernel(const float *__restrict__ xyz, int N, int m,
                                                      int32_t *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  // layout: [2][kWaves] vals | [2][kWaves] keys | (LDS_COORDS) x[N] y[N] z[N]
  float *s_val = s_mem;
  int *s_key = reinterpret_cast<int *>(s_mem + 2 * kWaves);
  float *s_x = s_mem + 4 * kWaves;
  float *s_y = s_x + N;
  float *s_z = s_y + N;

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const float *pc = xyz + (size_t)b * N * 3;

  // Lane t owns k_j = (t & 511) + 512*(2j + (t >> 9)): keys increase with j, so a strict '>' scan
  // over j keeps the smallest key among equal values.
  float px[PPT], py[PPT], pz[PPT], md[PPT];
  const int k0 = (tid & 511) + 512 * (tid >> 9);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = k0 + 1024 * j;
    if (k < N) {
      px[j] = pc[(size_t)k * 3];
      py[j] = pc[(size_t)k * 3 + 1];
      pz[j] = pc[(size_t)k * 3 + 2];
      md[j] = 1e38f;
      if (LDS_COORDS) { s_x[k] = px[j]; s_y[k] = py[j]; s_z[k] = pz[j]; }
    } else {
      px[j] = py[j] = pz[j] = 0.f;
      md[j] = -2.f;  // below the reference's initial best = -1: can never be picked
    }
  }
  if (tid == 0) out[(size_t)b * m] = 0;
  __syncthreads();

  int old = 0;
  int buf = 0;
  for (int r = 1; r < m; ++r) {
    float x1, y1, z1;
    if (LDS_COORDS) { x1 = s_x[old]; y1 = s_y[old]; z1 = s_z[old]; }
    else { x1 = pc[(size_t)old * 3]; y1 = pc[(size_t)old * 3 + 1]; z1 = pc[(size_t)old * 3 + 2]; }

    float best = -1.f;
    int bestk = 0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const float dx = px[j] - x1, dy = py[j] - y1, dz = pz[j] - z1;
      const float d = __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
      const float d2 = fminf(d, md[j]);
      md[j] = d2;
      if (d2 > best) { best = d2; bestk = k0 + 1024 * j; }
    }
    int key = fps_key(bestk);
    wave_argmax(best, key);
    if (lane == 0) { s_val[buf * kWaves + wave] = best; s_key[buf * kWaves + wave] = key; }
    __syncthreads();
    float v = s_val[buf * kWaves + (lane & (kWaves - 1))];
    int kk = s_key[buf * kWaves + (lane & (kWaves - 1))];
#pragma unroll
    for (int off = kWaves / 2; off > 0; off >>= 1) {
      const float ov = __shfl_xor(v, off, 64);
      const int ok = __shfl_xor(kk, off, 64);
      const bool take = (ov > v) || (ov == v && ok < kk);
      v = take ? ov : v;
      kk = take ? ok : kk;
    }
    old = fps_unkey(kk);
    if (tid == 0) out[(size_t)b * m + r] = old;
    buf ^= 1;
  }
}

template <int PPT>
int fps_launch(const float *xyz, int B, int N, int m, int32_t *out, hipStream_t s) {
  const size_t red = sizeof(float) * 4 * kWaves;
  const bool lds_coords = (size_t)N * 12 + red <= 128 * 1024;
  if (lds_coords) {
    DH3D_ALLOW_BIG_LDS((fps_kernel<PPT, true>));
    hipLaunchKernelGGL((fps_kernel<PPT, true>), dim3(B), dim3(kThreads), red + (size_t)N * 12, s, xyz, N,
                       m, out);
  } else {
    hipLaunchKernelGGL((fps_kernel<PPT, false>), dim3(B), dim3(kThreads), red, s, xyz, N, m, out);
  }
  return dh3d_launch_status();
}

}  // namespace

DH3D_API int dh3d_farthest_point_sample(int B, int N, int m, const float *inp, float *temp,
                                        int32_t *out, void *stream) {
  (void)temp;
  DH3D_REQUIRE(inp && out && B > 0 && N > 0 && m > 0);  // tf_sampling.cpp:100,105
  DH3D_SUPPORTED(N <= 16384);
  hipStream_t s = (hipStream_t)stream;
  if (N <= 1024) return fps_launch<1>(inp, B, N, m, out, s);
  if (N <= 2048) return fps_launch<2>(inp, B, N, m, out, s);
  if (N <= 4096) return fps_launch<4>(inp, B, N, m, out, s);
  if (N <= 8192) return fps_launch<8>(inp, B, N, m, out, s);
  return fps_launch<16>(inp, B, N, m, out, s);
}
