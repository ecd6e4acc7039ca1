// Spatial (Morton) ordering of a cloud for gfx950 -- a preprocessing step with no counterpart in the
// reference.  It changes no result: kNN and FPS stay exact (bit-identical ids), but both can then skip
// work that provably cannot matter, because 64 consecutive points of the order form a compact group
// with a tight bounding box:
//   * knn_sorted_kernel (knn.hip) visits candidate groups nearest-first and skips a whole group when its
//     box is farther from the query group's box than every lane's current K-th distance;
//   * fps_sorted_kernel (fps.hip) re-evaluates a group's min-distances only when the new sample is
//     closer to the group's box than the group's current maximum.
// One 1024-lane workgroup per cloud: cloud bounding box -> 18-bit Morton cell (6 bits per axis) ->
// in-LDS bitonic sort of 32-bit keys (cell << 14 | original index; N <= 16384) -> sorted float4 records
// (x, y, z, bits(original index)) and one box per group of 64.
#include "common.h"
#include "wave_ops.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = 16;

__device__ __forceinline__ unsigned spread6(unsigned v) {  // 6 bits -> every third bit
  v &= 63u;
  v = (v | (v << 8)) & 0x300Fu;
  v = (v | (v << 4)) & 0x30C3u;
  v = (v | (v << 2)) & 0x9249u;
  return v;
}

template <int PPT>
__global__ __launch_bounds__(kThreads) void spatial_sort_kernel(const float *__restrict__ xyz, int N,
                                                               int npad, float4 *__restrict__ sorted,
                                                               float *__restrict__ gbox) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_keys[];  // [npad] then 6*kWaves floats
  float *s_red = reinterpret_cast<float *>(s_keys + npad);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *pc = xyz + (size_t)b * N * 3;
  const int NG = (N + 63) / 64;

  // ---- cloud bounding box
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float px[PPT], py[PPT], pz[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = tid + kThreads * j;
    px[j] = py[j] = pz[j] = 0.f;
    if (k < N) {
      px[j] = pc[(size_t)k * 3]; py[j] = pc[(size_t)k * 3 + 1]; pz[j] = pc[(size_t)k * 3 + 2];
      lo[0] = fminf(lo[0], px[j]); hi[0] = fmaxf(hi[0], px[j]);
      lo[1] = fminf(lo[1], py[j]); hi[1] = fmaxf(hi[1], py[j]);
      lo[2] = fminf(lo[2], pz[j]); hi[2] = fmaxf(hi[2], pz[j]);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float wl = wave_min_f32(lo[a]), wh = wave_max_f32(hi[a]);
    if (lane == 0) { s_red[wave * 6 + a] = wl; s_red[wave * 6 + 3 + a] = wh; }
  }
  __syncthreads();
  float scale[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = INFINITY, h = -INFINITY;
    for (int w = 0; w < kWaves; ++w) { l = fminf(l, s_red[w * 6 + a]); h = fmaxf(h, s_red[w * 6 + 3 + a]); }
    lo[a] = l;
    scale[a] = 64.f / fmaxf(h - l, 1e-30f);
  }
  // ---- keys
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = tid + kThreads * j;
    if (k < npad) {
      unsigned key = 0xFFFFFFFFu;  // padding sorts last
      if (k < N) {
        const unsigned cx = (unsigned)min(63, max(0, (int)((px[j] - lo[0]) * scale[0])));
        const unsigned cy = (unsigned)min(63, max(0, (int)((py[j] - lo[1]) * scale[1])));
        const unsigned cz = (unsigned)min(63, max(0, (int)((pz[j] - lo[2]) * scale[2])));
        const unsigned cell = spread6(cx) | (spread6(cy) << 1) | (spread6(cz) << 2);
        key = (cell << 14) | (unsigned)k;
      }
      s_keys[k] = key;
    }
  }
  __syncthreads();
  // ---- bitonic sort of npad (power of two) keys
  for (int size = 2; size <= npad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int p = tid; p < (npad >> 1); p += kThreads) {
        const int i = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
        const int q = i | stride;
        const unsigned a = s_keys[i], c = s_keys[q];
        const bool up = ((i & size) == 0);
        if ((a > c) == up) { s_keys[i] = c; s_keys[q] = a; }
      }
      __syncthreads();
    }
  }
  // ---- sorted records + one box per 64: lane l of wave w owns positions (w + 16 j) * 64 + l
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int g = wave + kWaves * j;
    const int i = g * 64 + lane;
    float x = 0.f, y = 0.f, z = 0.f;
    float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
    if (i < N) {
      const int k = (int)(s_keys[i] & 0x3FFFu);
      x = pc[(size_t)k * 3]; y = pc[(size_t)k * 3 + 1]; z = pc[(size_t)k * 3 + 2];
      sorted[(size_t)b * N + i] = make_float4(x, y, z, __int_as_float(k));
      lx = hx = x; ly = hy = y; lz = hz = z;
    }
    if (g < NG) {  // wave-uniform
      lx = wave_min_f32(lx); ly = wave_min_f32(ly); lz = wave_min_f32(lz);
      hx = wave_max_f32(hx); hy = wave_max_f32(hy); hz = wave_max_f32(hz);
      if (lane == 0) {
        float *o = gbox + ((size_t)b * NG + g) * 8;
        o[0] = lx; o[1] = ly; o[2] = lz; o[3] = 0.f; o[4] = hx; o[5] = hy; o[6] = hz; o[7] = 0.f;
      }
    }
  }
}

template <int PPT>
int sort_launch(const float *xyz, int B, int N, float4 *sorted, float *gbox, hipStream_t s) {
  int npad = 2;
  while (npad < N) npad <<= 1;
  const size_t lds = sizeof(unsigned) * npad + sizeof(float) * 6 * kWaves;
  DH3D_ALLOW_BIG_LDS((spatial_sort_kernel<PPT>));
  hipLaunchKernelGGL((spatial_sort_kernel<PPT>), dim3(B), dim3(kThreads), lds, s, xyz, N, npad, sorted, gbox);
  return dh3d_launch_status();
}

}  // namespace

DH3D_API int dh3d_spatial_sort(const float *xyz, int B, int N, float *sorted, float *gbox, void *stream) {
  DH3D_REQUIRE(xyz && sorted && gbox && B > 0 && N > 0);
  DH3D_SUPPORTED(N <= 16384);
  hipStream_t s = (hipStream_t)stream;
  float4 *so = reinterpret_cast<float4 *>(sorted);
  if (N <= 1024) return sort_launch<1>(xyz, B, N, so, gbox, s);
  if (N <= 2048) return sort_launch<2>(xyz, B, N, so, gbox, s);
  if (N <= 4096) return sort_launch<4>(xyz, B, N, so, gbox, s);
  if (N <= 8192) return sort_launch<8>(xyz, B, N, so, gbox, s);
  return sort_launch<16>(xyz, B, N, so, gbox, s);
}
