// Farthest point sampling for gfx950 -- replaces farthestpointsamplingKernel
// (tf_ops/sampling/tf_sampling_g.cu:105-170; launcher :203-205, op tf_sampling.cpp:95-123).
//
// FPS is a chain of m dependent rounds; the only outer parallelism is the batch.  One 1024-lane
// workgroup (16 waves, one CU) owns a cloud.  Each lane keeps its PPT points AND their running
// min-distances in registers for the whole kernel (the reference re-reads a [32,n] global scratch
// every round).  A round is latency-bound, so every step is kept off the LDS where possible:
//   packed-f32 distance update (two points per v_pk_* op) -> wave arg-max on the DPP crossbar
//   (wave_ops.h) -> 16 partial winners through LDS (one barrier per round, double-buffered) ->
//   16-lane DPP row reduce in every wave -> winner coordinates from the LDS copy of the cloud.
// (The first version used ds_bpermute shuffles for both reductions: 1.5 ms for N=8192 -> 1024,
//  profiles/r01_a.)
//
// Bit-exactness (compiled with -ffp-contract=off):
//   d    = fma(dz,dz, fma(dx,dx, dy*dy))      -- nvcc/LLVM contraction of (:141); see DESIGN.md
//   td   = min(d, td)  starting from 1e38     (:118,143-145)
//   pick = max td; ties -> smallest (k % 512, k / 512): the reference's 512 threads each scan
//          k = tid, tid+512, ... with strict '>' (:146-149) and its tree keeps the lower slot on
//          ties (:158-161).  Encoded here as key(k) = ((k & 511) << 16) | (k >> 9), smaller wins.
#include <limits.h>

#include "common.h"
#include "wave_ops.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));


#ifdef DH3D_FPS_PROBE  // dev instrumentation: cycle stamps of one round of wave 0 (tools/fps_probe.py)
__device__ long long g_probe[16];
__device__ unsigned long long g_fps_cnt[2];  // ordered kernel: active (wave, round) pairs, updated groups
#define PROBE(i) do { if (r == 300 && tid == 0 && blockIdx.x == 0) g_probe[i] = clock64(); } while (0)
#else
#define PROBE(i) do { } while (0)
#endif

__device__ __forceinline__ int fps_key(int k) { return ((k & 511) << 16) | (k >> 9); }
__device__ __forceinline__ int fps_unkey(int key) { return ((key & 0xffff) << 9) | (key >> 16); }

// Single-instruction min / max3: the operands are never NaN here, so the canonicalising v_max x,x that
// fminf/fmaxf carry under IEEE mode would only cost issue slots in a loop that is VALU-issue bound.
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int PPT, int WAVES, bool LDS_COORDS>
__global__ __launch_bounds__(64 * WAVES) void fps_kernel(const float *__restrict__ xyz, int N, int m,
                                                       int32_t *__restrict__ out) {
  constexpr int T = 64 * WAVES;
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  // layout: 3 x u64 block-best slots (+pad to 4*WAVES floats) | (LDS_COORDS) x[N] y[N] z[N] picks[m]
  // The block arg-max is ONE 64-bit LDS atomic max per wave on (bits(value) << 32 | ~key): value >= 0 so
  // unsigned order is (value, smaller key wins).  Slots rotate so the reset never races a reader.
  unsigned long long *s_best = reinterpret_cast<unsigned long long *>(s_mem);
  float *s_x = s_mem + 4 * WAVES;
  float *s_y = s_x + N;
  float *s_z = s_y + N;
  int *s_out = reinterpret_cast<int *>(s_z + N);  // picks are buffered here: a global store per round would
                                                  // put its write latency (vmcnt(0) before the barrier) on the chain

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const float *pc = xyz + (size_t)b * N * 3;

  // Lane t owns k_j = t + T*j, held as pairs (j = 2p, 2p+1).  Ties are broken on key(k) explicitly.
  constexpr int NP = (PPT + 1) / 2;
  f32x2 px[NP], py[NP], pz[NP], md[NP];
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) {
    const int k = tid + T * j;
    float x = 0.f, y = 0.f, z = 0.f, d = -2.f;  // -2: below the reference's initial best = -1, never picked
    if (j < PPT && k < N) {
      x = pc[(size_t)k * 3]; y = pc[(size_t)k * 3 + 1]; z = pc[(size_t)k * 3 + 2];
      d = 1e38f;
      if (LDS_COORDS) { s_x[k] = x; s_y[k] = y; s_z[k] = z; }
    }
    px[j >> 1][j & 1] = x; py[j >> 1][j & 1] = y; pz[j >> 1][j & 1] = z; md[j >> 1][j & 1] = d;
  }
  if (tid == 0) { if (LDS_COORDS) s_out[0] = 0; else out[(size_t)b * m] = 0; }
  if (tid < 3) s_best[tid] = 0ull;
  __syncthreads();

  int old = 0;
  int buf = 1;  // slot of round r is r % 3
  for (int r = 1; r < m; ++r) {
    PROBE(0);
    float x1, y1, z1;
    if (LDS_COORDS) { x1 = s_x[old]; y1 = s_y[old]; z1 = s_z[old]; }
    else { x1 = pc[(size_t)old * 3]; y1 = pc[(size_t)old * 3 + 1]; z1 = pc[(size_t)old * 3 + 2]; }
    const f32x2 x2 = {x1, x1}, y2 = {y1, y1}, z2 = {z1, z1};
#ifdef DH3D_FPS_PROBE
    asm volatile("" :: "v"(x1), "v"(y1), "v"(z1));
#endif
    PROBE(1);
    // running min-distance update (two points per packed op) + the lane's best VALUE only
    float best = -1.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const f32x2 dx = px[p] - x2, dy = py[p] - y2, dz = pz[p] - z2;
      const f32x2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
      md[p][0] = vmin(d[0], md[p][0]);
      md[p][1] = vmin(d[1], md[p][1]);
      best = vmax3(best, md[p][0], md[p][1]);
    }
    PROBE(2);
    const float wmax = wave_max_f32(best);
    PROBE(3);
    // smallest key among the lane's points that hold wmax
    int lkey = INT_MAX;
#pragma unroll
    for (int j = 0; j < 2 * NP; ++j) {
      const int kj = fps_key(tid + T * j);
      lkey = (md[j >> 1][j & 1] == wmax) ? min(lkey, kj) : lkey;
    }
    const unsigned long long hit = __ballot(best == wmax);
    int wkey;
    if (__popcll(hit) == 1) wkey = __builtin_amdgcn_readlane(lkey, __builtin_ctzll(hit));
    else wkey = wave_min_i32(lkey);  // several lanes tie: smallest key wins (non-hit lanes hold INT_MAX)
    PROBE(4);
    if (lane == 0) {
      const unsigned long long packed =
          wmax >= 0.f ? (((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned)~wkey) : 0ull;
      atomicMax(&s_best[buf], packed);
      if (wave == 0) s_best[buf == 2 ? 0 : buf + 1] = 0ull;  // next round's slot (last read two rounds ago)
    }
    __syncthreads();
    PROBE(5);
    const int bkey = ~(int)(unsigned)s_best[buf];
    old = fps_unkey(bkey);
    PROBE(6);
    if (tid == 0) { if (LDS_COORDS) s_out[r] = old; else out[(size_t)b * m + r] = old; }
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (LDS_COORDS) {
    __syncthreads();
    for (int r = tid; r < m; r += T) out[(size_t)b * m + r] = s_out[r];
  }
}

// ------------------------------------------------------------------------------------------------
// FPS on a spatially ordered cloud (spatial.hip).  Same results, far less work per round.
//
// Wave w owns the Morton-consecutive groups [w*PPT, (w+1)*PPT) (64 points each: a compact region of the
// cloud); lane l holds point l of each of them.  A point's running min-distance can only drop if the new sample
// is closer to it than that distance, hence -- for a whole group -- only if the sample is closer to the group's
// bounding box than the largest min-distance in the wave.  Per round every wave
//   1. tests its PPT boxes in parallel (lane j <-> group j: ~12 VALU ops + one ballot) against its cached maximum;
//   2. if nothing can change, re-offers its cached (max, key) -- no update, no reduction;
//   3. otherwise updates just the hit groups (one point per lane each) and redoes ONE wave arg-max.
// After the first few dozen samples a round touches one or two waves instead of all N points; the others spend
// ~40 instructions.  (A first version cached a maximum per GROUP and reduced once per hit group: the
// reductions serialised and it was no faster than the plain kernel.)
// The skip test carries a 1e-5 relative margin: it may keep a group that cannot change, never the reverse.
template <int PPT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fps_sorted_kernel(const float4 *__restrict__ sorted,
                                                              const float *__restrict__ gbox, int N, int m,
                                                              int32_t *__restrict__ out) {
  static_assert(PPT <= 64, "one lane per group box");
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  // 3 rotating u64 block-best slots (as fps_kernel) | coordinates by ORIGINAL index | picks
  // (Publishing each wave's candidate coordinates in a per-wave slot and selecting the winner's with v_readlane
  //  -- to save the dependent table lookup -- measured SLOWER: 0.40 vs 0.36 ms of pure sync chain at 8 waves.)
  unsigned long long *s_best = reinterpret_cast<unsigned long long *>(s_mem);
  float *s_x = s_mem + 4 * WAVES;
  float *s_y = s_x + N;
  float *s_z = s_y + N;
  int *s_out = reinterpret_cast<int *>(s_z + N);  // picks, written to global memory once at the end

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NG = (N + 63) / 64;
  const float4 *sc = sorted + (size_t)b * N;

  float px[PPT], py[PPT], pz[PPT], md[PPT];
  int pkey[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = (wave * PPT + j) * 64 + lane;
    px[j] = py[j] = pz[j] = 0.f;
    md[j] = -2.f;  // padding: below the reference's initial best = -1, never picked
    pkey[j] = INT_MAX;
    if (i < N) {
      const float4 r = sc[i];
      const int k = __float_as_int(r.w);
      px[j] = r.x; py[j] = r.y; pz[j] = r.z;
      md[j] = 1e38f;
      pkey[j] = fps_key(k);
      s_x[k] = r.x; s_y[k] = r.y; s_z[k] = r.z;
    }
  }
  // lane j < PPT: bounding box of this wave's group j
  float blx = INFINITY, bly = INFINITY, blz = INFINITY, bhx = -INFINITY, bhy = -INFINITY, bhz = -INFINITY;
  bool has_box = false;
  if (lane < PPT) {
    const int g = wave * PPT + lane;
    if (g < NG) {
      const float *bx = gbox + ((size_t)b * NG + g) * 8;
      blx = bx[0]; bly = bx[1]; blz = bx[2]; bhx = bx[4]; bhy = bx[5]; bhz = bx[6];
      has_box = true;
    }
  }
  if (tid == 0) s_out[0] = 0;
  if (tid < 3) s_best[tid] = 0ull;
  __syncthreads();

  // cached wave maximum (uniform): 1e38 makes the first round update everything; a wave of pure padding
  // offers nothing
  float wmax = __ballot(has_box) != 0ull ? 1e38f : -2.f;
  int wkey = INT_MAX;
  int old = 0;
  int buf = 1;  // slot of round r is r % 3
  for (int r = 1; r < m; ++r) {
    const float x1 = s_x[old], y1 = s_y[old], z1 = s_z[old];
    // 1. which of my groups can change?  squared distance from the sample to each box (0 inside)
    const float ex = fmaxf(fmaxf(blx - x1, x1 - bhx), 0.f);
    const float ey = fmaxf(fmaxf(bly - y1, y1 - bhy), 0.f);
    const float ez = fmaxf(fmaxf(blz - z1, z1 - bhz), 0.f);
    const float bd = (ex * ex + ey * ey + ez * ez) * 0.99999f;
    const unsigned long long need = __ballot(has_box && bd <= wmax);
#if defined(DH3D_FPS_PROBE) && DH3D_FPS_PROBE == 1
    if (lane == 0 && need != 0ull) { atomicAdd(&g_fps_cnt[0], 1ull); atomicAdd(&g_fps_cnt[1], (unsigned long long)__popcll(need)); }
#endif
#if defined(DH3D_FPS_PROBE) && DH3D_FPS_PROBE == 2  // timing experiment: sync chain only (results wrong)
    if (need != 0ull && r < 4) {
#else
    if (need != 0ull) {  // wave-uniform
#endif
      // 2. update the hit groups, then one wave arg-max over everything the wave holds
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        if ((need >> j) & 1ull) {
          const float dx = px[j] - x1, dy = py[j] - y1, dz = pz[j] - z1;
          const float d = __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
          md[j] = vmin(d, md[j]);
        }
      }
      float best = -2.f;
#pragma unroll
      for (int j = 0; j + 1 < PPT; j += 2) best = vmax3(best, md[j], md[j + 1]);
      if (PPT & 1) best = fmaxf(best, md[PPT - 1]);
      wmax = wave_max_f32(best);
      int lkey = INT_MAX;  // smallest key among the lane's points that hold wmax
#pragma unroll
      for (int j = 0; j < PPT; ++j) lkey = (md[j] == wmax) ? min(lkey, pkey[j]) : lkey;
      const unsigned long long hit = __ballot(best == wmax);
      if (__popcll(hit) == 1) wkey = __builtin_amdgcn_readlane(lkey, __builtin_ctzll(hit));
      else wkey = wave_min_i32(lkey);  // several lanes tie: smallest key wins (non-hit lanes hold INT_MAX)
    }
    // 3. block arg-max: one 64-bit LDS atomic max per wave on (bits(value) << 32 | ~key)
    if (lane == 0) {
      const unsigned long long packed =
          wmax >= 0.f ? (((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned)~wkey) : 0ull;
      atomicMax(&s_best[buf], packed);
      if (wave == 0) s_best[buf == 2 ? 0 : buf + 1] = 0ull;  // next round's slot (last read two rounds ago)
    }
    __syncthreads();
    const int bkey = ~(int)(unsigned)s_best[buf];
    old = fps_unkey(bkey);
    if (tid == 0) s_out[r] = old;
    buf = buf == 2 ? 0 : buf + 1;
  }
  __syncthreads();
  for (int r = tid; r < m; r += 64 * WAVES) out[(size_t)b * m + r] = s_out[r];
}

template <int PPT, int WAVES>
int fps_sorted_launch(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out, hipStream_t s) {
  const size_t lds = sizeof(float) * (4 * WAVES + (size_t)3 * N + m);
  if (lds > 159 * 1024) return DH3D_ERR_UNSUPPORTED;
  DH3D_ALLOW_BIG_LDS((fps_sorted_kernel<PPT, WAVES>));
  hipLaunchKernelGGL((fps_sorted_kernel<PPT, WAVES>), dim3(B), dim3(64 * WAVES), lds, s,
                     reinterpret_cast<const float4 *>(sorted), gbox, N, m, out);
  return dh3d_launch_status();
}

template <int PPT, int WAVES>
int fps_launch(const float *xyz, int B, int N, int m, int32_t *out, hipStream_t s) {
  const size_t red = sizeof(float) * 4 * WAVES;
  const bool lds_coords = (size_t)N * 12 + (size_t)m * 4 + red <= 159 * 1024;
  if (lds_coords) {
    DH3D_ALLOW_BIG_LDS((fps_kernel<PPT, WAVES, true>));
    hipLaunchKernelGGL((fps_kernel<PPT, WAVES, true>), dim3(B), dim3(64 * WAVES),
                       red + (size_t)N * 12 + (size_t)m * 4, s, xyz, N, m, out);
  } else {
    hipLaunchKernelGGL((fps_kernel<PPT, WAVES, false>), dim3(B), dim3(64 * WAVES), red, s, xyz, N, m, out);
  }
  return dh3d_launch_status();
}

}  // namespace

// Dev knob (tools/geo_bench.py): waves per cloud; 0 = default.
static int g_fps_waves = 0;
DH3D_API void dh3d_dev_set_fps_waves(int w) { g_fps_waves = w; }

DH3D_API int dh3d_farthest_point_sample(int B, int N, int m, const float *inp, float *temp,
                                        int32_t *out, void *stream) {
  (void)temp;
  DH3D_REQUIRE(inp && out && B > 0 && N > 0 && m > 0);  // tf_sampling.cpp:100,105
  DH3D_SUPPORTED(N <= 16384);
  hipStream_t s = (hipStream_t)stream;
  const int W = g_fps_waves ? g_fps_waves : (N <= 1024 ? 4 : 8);  // measured best on MI355X (tools/geo_bench.py)
#define DH3D_FPS_CASE(WV)                                                              \
  if (W == WV) {                                                                       \
    const int per = 64 * WV;                                                           \
    if (N <= per * 2) return fps_launch<2, WV>(inp, B, N, m, out, s);                  \
    if (N <= per * 4) return fps_launch<4, WV>(inp, B, N, m, out, s);                  \
    if (N <= per * 8) return fps_launch<8, WV>(inp, B, N, m, out, s);                  \
    if (N <= per * 16) return fps_launch<16, WV>(inp, B, N, m, out, s);                \
    if (N <= per * 32) return fps_launch<32, WV>(inp, B, N, m, out, s);                \
    if (N <= per * 64) return fps_launch<64, WV>(inp, B, N, m, out, s);                \
  }
  DH3D_FPS_CASE(4)
  DH3D_FPS_CASE(8)
  DH3D_FPS_CASE(16)
#undef DH3D_FPS_CASE
  return DH3D_ERR_UNSUPPORTED;
}

// Dev knob (tools/geo_bench.py): waves per cloud for the ordered kernel; 0 = default.
static int g_fps_sorted_waves = 0;
DH3D_API void dh3d_dev_set_fps_sorted_waves(int w) { g_fps_sorted_waves = w; }

DH3D_API int dh3d_fps_sorted(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out,
                             void *stream) {
  DH3D_REQUIRE(sorted && gbox && out && B > 0 && N > 0 && m > 0);
  DH3D_SUPPORTED(N <= 12288);  // the by-original-index coordinate table must fit LDS (12 B / point)
  hipStream_t s = (hipStream_t)stream;
  const int W = g_fps_sorted_waves ? g_fps_sorted_waves : 16;  // measured best on MI355X (tools/geo_bench.py)
  const int NG = (N + 63) / 64;
#define DH3D_FPS_CASE(WV)                                                                             \
  if (W == WV) {                                                                                      \
    const int gpw = (NG + WV - 1) / WV; /* groups per wave */                                         \
    if (gpw <= 1) return fps_sorted_launch<1, WV>(sorted, gbox, B, N, m, out, s);                     \
    if (gpw <= 2) return fps_sorted_launch<2, WV>(sorted, gbox, B, N, m, out, s);                     \
    if (gpw <= 4) return fps_sorted_launch<4, WV>(sorted, gbox, B, N, m, out, s);                     \
    if (gpw <= 8) return fps_sorted_launch<8, WV>(sorted, gbox, B, N, m, out, s);                     \
    if (gpw <= 16) return fps_sorted_launch<16, WV>(sorted, gbox, B, N, m, out, s);                   \
    if (gpw <= 32) return fps_sorted_launch<32, WV>(sorted, gbox, B, N, m, out, s);                   \
    if (gpw <= 48) return fps_sorted_launch<48, WV>(sorted, gbox, B, N, m, out, s);                   \
  }
  DH3D_FPS_CASE(4)
  DH3D_FPS_CASE(8)
  DH3D_FPS_CASE(16)
#undef DH3D_FPS_CASE
  return DH3D_ERR_UNSUPPORTED;
}

#ifdef DH3D_FPS_PROBE
DH3D_API int dh3d_fps_cnt_read(unsigned long long *host2, int reset) {
  int rc = hipMemcpyFromSymbol(host2, HIP_SYMBOL(g_fps_cnt), 16) == hipSuccess ? 0 : 3;
  if (reset) { unsigned long long z[2] = {0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fps_cnt), z, 16); }
  return rc;
}
DH3D_API int dh3d_fps_probe_read(long long *host16) {
  return hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_probe), sizeof(long long) * 16) == hipSuccess ? 0 : 3;
}
#endif
