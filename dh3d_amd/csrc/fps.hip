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
__device__ long long g_probe[32];
__device__ unsigned long long g_fps_cnt[2];  // ordered kernel: active (wave, round) pairs, updated groups
#define PROBE(i) do { if (r == 300 && tid == 0 && blockIdx.x == 0) g_probe[i] = clock64(); } while (0)
// batched kernel: per-phase cycle sums of one wave of cloud 0 (DH3D_FPS_PROBE_WAVE, default 0)
#ifndef DH3D_FPS_PROBE_WAVE
#define DH3D_FPS_PROBE_WAVE 0
#endif
#define STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pt[i] = clock64(); } while (0)
#define ASTAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); at[i] = clock64(); } while (0)
#else
#define PROBE(i) do { } while (0)
#define STAMP(i) do { } while (0)
#define ASTAMP(i) do { } while (0)
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
__device__ __forceinline__ float vmaxf(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
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

// ------------------------------------------------------------------------------------------------
// Batched rounds on the ordered cloud: several picks per synchronisation, same picks as the sequential rule.
//
// What bounds fps_sorted_kernel is not its arithmetic but the barrier + LDS chain of a round.  FPS picks are far
// apart by construction, so consecutive picks rarely interact: after a sync every wave holds the arg-max c_w of
// its region (value v_w, tie key) and the second-largest value s_w in the region.  Order the candidates by
// (value, key).  The j-th candidate IS the j-th next pick of the sequential algorithm if no higher-ranked
// candidate c_p (a) lowers its distance (d(c_p, c_j) < v_j, computed with the update's own arithmetic) or
// (b) leaves a better point behind in its region (s_p >= v_j); everything else can only have dropped.  Each wave
// judges its OWN candidate against the 16 published ones (16 lanes, ~10 VALU), the first bad rank is an LDS
// atomicMin, and all candidates ranked before it are taken at once: ~3.7 picks per sync on uniform clouds
// (tools/fps_batch_sim.py).  Two barriers per sync instead of one per pick; the box tests of up to 64/PPT picks
// against the wave's PPT boxes run in ONE pass (lane = pick * PPT + box).
// TABLE: the by-original-index coordinate table lives in LDS (12 B per point: clouds of up to ~12 k points).  Larger
// clouds (TABLE = false) read the winner's coordinates from the cloud itself (`xyz`, L2-resident) instead -- one global
// round trip on the active wave's chain per sync, still far ahead of the one-pick-per-round kernel at 16384 points.
template <int PPT, int WAVES, bool TABLE>
__global__ __launch_bounds__(64 * WAVES) void fps_batched_kernel(const float4 *__restrict__ sorted,
                                                               const float *__restrict__ gbox, int N, int m,
                                                               int32_t *__restrict__ out,
                                                               float *__restrict__ xyz_out,
                                                               const float *__restrict__ xyz) {
  static_assert(PPT <= 64 && WAVES <= 16, "one lane per group box, one lane per candidate");
  constexpr int CAP0 = PPT <= 32 ? 64 / PPT : 1;
  constexpr int CAP = CAP0 < WAVES ? CAP0 : WAVES;  // picks per sync
  constexpr int NP = (PPT + 1) / 2;                 // groups are held and updated in pairs (packed f32)
  constexpr int JW = WAVES < 4 ? WAVES : 4;         // judging waves (one per SIMD), CPJ candidates each
  constexpr int CPJ = (WAVES + JW - 1) / JW;
  constexpr int kBig = 1 << 20;
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  // candidates [WAVES][8]: key lo, key hi, second value, x | y, z, index, -   | picks [16] x,y,z,index | stop[4]
  float4 *s_ent = reinterpret_cast<float4 *>(s_mem);
  float4 *s_pick = s_ent + 2 * WAVES;
  int *s_jthr = reinterpret_cast<int *>(s_pick + 16);  // per judging wave: first rank it holds back
  float *s_x = reinterpret_cast<float *>(s_jthr + 4);
  float *s_y = s_x + (TABLE ? N : 0);
  float *s_z = s_y + (TABLE ? N : 0);
  int *s_out = reinterpret_cast<int *>(s_z + (TABLE ? N : 0));
  const float *pc = TABLE ? nullptr : xyz + (size_t)blockIdx.x * N * 3;

  // the kernel is a dependent chain on 1 CU per cloud while the rest of the step shares the chip: its waves go first
  __builtin_amdgcn_s_setprio(3);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NG = (N + 63) / 64;
  const float4 *sc = sorted + (size_t)b * N;

  f32x2 px[NP], py[NP], pz[NP], md[NP];
  int pkey[2 * NP];
#pragma unroll
  for (int j = 0; j < 2 * NP; ++j) {
    const int i = (wave * PPT + j) * 64 + lane;
    float x = 0.f, y = 0.f, z = 0.f, d = -2.f;  // padding: below the reference's initial best = -1, never picked
    pkey[j] = INT_MAX;
    if (j < PPT && i < N) {
      const float4 r = sc[i];
      const int k = __float_as_int(r.w);
      x = r.x; y = r.y; z = r.z;
      d = 1e38f;
      pkey[j] = fps_key(k);
      if (TABLE) { s_x[k] = r.x; s_y[k] = r.y; s_z[k] = r.z; }
    }
    px[j >> 1][j & 1] = x; py[j >> 1][j & 1] = y; pz[j >> 1][j & 1] = z; md[j >> 1][j & 1] = d;
  }
  // lane = pick * PPT + box: every pick slot sees the wave's PPT boxes
  const int bl = lane % PPT, pk = lane / PPT;
  float blx = INFINITY, bly = INFINITY, blz = INFINITY, bhx = -INFINITY, bhy = -INFINITY, bhz = -INFINITY;
  bool has_box = false;
  if (pk < CAP) {
    const int g = wave * PPT + bl;
    if (g < NG) {
      const float *bx = gbox + ((size_t)b * NG + g) * 8;
      blx = bx[0]; bly = bx[1]; blz = bx[2]; bhx = bx[4]; bhy = bx[5]; bhz = bx[6];
      has_box = true;
    }
  }
  if (tid == 0) s_out[0] = 0;
  if (tid < 4) s_jthr[tid] = kBig;
  if (lane < 2) s_ent[2 * wave + lane] = make_float4(0.f, 0.f, -2.f, 0.f);  // key 0 = nothing to offer
  __syncthreads();

  // A wave executes ~1 instruction per 4.6 cycles whatever its kind (tools/fps_exp.py: the loop below is bound by the
  // instruction count along the chain barrier -> judge -> barrier -> box test -> update -> arg-max, not by data), so
  // every phase is written for few instructions: no per-group branches, no atomics, nothing recomputed.
  float wmax = __ballot(has_box) != 0ull ? 1e38f : -2.f;  // cached wave maximum (uniform)
  float qx, qy, qz;                                         // lane (pk, .): coordinates of pick pk of this sync
  if (TABLE) { qx = s_x[0]; qy = s_y[0]; qz = s_z[0]; }
  else { qx = pc[0]; qy = pc[1]; qz = pc[2]; }
  int npick = 1, r = 1;
#ifdef DH3D_FPS_PROBE
  long long pt[12], at[8];
#endif
  constexpr unsigned long long kPickMask = PPT == 64 ? ~0ull : ((1ull << (PPT & 63)) - 1ull);
  while (true) {
    STAMP(0);
    // 1. every (pick, box) pair at once: can the pick change anything in the group?
    const float ex = fmaxf(fmaxf(blx - qx, qx - bhx), 0.f);
    const float ey = fmaxf(fmaxf(bly - qy, qy - bhy), 0.f);
    const float ez = fmaxf(fmaxf(blz - qz, qz - bhz), 0.f);
    const float bd = (ex * ex + ey * ey + ez * ez) * 0.99999f;
    const unsigned long long need = __ballot(has_box && pk < npick && bd <= wmax);
    STAMP(1);
#if defined(DH3D_FPS_EXP) && (DH3D_FPS_EXP & 1)  // timing experiment: sync chain only (results wrong)
    if (need != 0ull && r < 8) {
#else
    if (need != 0ull) {  // wave-uniform
#endif
      unsigned long long nd = need;
#if defined(DH3D_FPS_EXP) && (DH3D_FPS_EXP & 4)
      if (r < 8)
#endif
      do {  // the picks that reach this wave
        const int p = __builtin_ctzll(nd) / PPT;
        const unsigned long long nb = (nd >> (p * PPT)) & kPickMask;
        nd &= ~(kPickMask << (p * PPT));
        const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), p * PPT));
        const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), p * PPT));
        const float z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), p * PPT));
        const f32x2 x2 = {x1, x1}, y2 = {y1, y1}, z2 = {z1, z1};
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          // small waves: all groups, branch-free (an update that was not needed is harmless and a skipped pair
          // would cost as many scalar instructions as it saves vector ones); large ones: the hit pairs only
          if (PPT <= 8 || ((nb >> (2 * q)) & 3ull)) {
            const f32x2 dx = px[q] - x2, dy = py[q] - y2, dz = pz[q] - z2;
            const f32x2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
            md[q][0] = __builtin_fminf(d[0], md[q][0]);
            md[q][1] = __builtin_fminf(d[1], md[q][1]);
          }
        }
      } while (nd != 0ull);
      ASTAMP(0);
#if defined(DH3D_FPS_EXP) && (DH3D_FPS_EXP & 8)
      if (r < 8) {
#else
      {
#endif
      // one wave arg-max (+ runner-up value) over everything the wave holds
      // (the key of each lane's first maximum rides along: a separate pass of compares after the wave maximum is
      //  known costs ~270 cycles of chain, this ~40)
      float b1 = -2.f, b2 = -2.f;
      int lkey = INT_MAX;
#pragma unroll
      for (int j = 0; j < 2 * NP; ++j) {
        const float x = md[j >> 1][j & 1];
        lkey = x > b1 ? pkey[j] : lkey;
        b2 = __builtin_amdgcn_fmed3f(b1, b2, x);
        b1 = __builtin_fmaxf(b1, x);
      }
      asm volatile("" :: "v"(b1), "v"(b2));
      ASTAMP(1);
      wmax = wave_max_f32(b1);
      ASTAMP(2);
      const unsigned long long hit = __ballot(b1 == wmax);
      int wl = __builtin_ctzll(hit);  // winner lane
      float wsec = wave_max_f32(lane == wl ? b2 : b1);  // runner-up (redone below if wl changes)
      ASTAMP(3);
      int wkey;
      if (__popcll(hit) == 1 && __ballot(b2 == wmax) == 0ull) {  // one point holds the maximum
        wkey = __builtin_amdgcn_readlane(lkey, wl);
      } else {  // ties: the smallest key wins
        lkey = INT_MAX;
#pragma unroll
        for (int j = 0; j < 2 * NP; ++j) lkey = (md[j >> 1][j & 1] == wmax) ? min(lkey, pkey[j]) : lkey;
        wkey = wave_min_i32(lkey);
        wl = __builtin_ctzll(__ballot(lkey == wkey));
        wsec = wave_max_f32(lane == wl ? b2 : b1);
      }
      ASTAMP(4);
      const int widx = fps_unkey(wkey);
      float wx, wy, wz;
      if (TABLE) { wx = s_x[widx]; wy = s_y[widx]; wz = s_z[widx]; }
      else { wx = pc[(size_t)widx * 3]; wy = pc[(size_t)widx * 3 + 1]; wz = pc[(size_t)widx * 3 + 2]; }
      ASTAMP(5);
      // publish the candidate (an untouched wave's entry stays valid)
      if (lane == 0) {
        s_ent[2 * wave] = make_float4(__int_as_float(~wkey), wmax, wsec, wx);
        s_ent[2 * wave + 1] = make_float4(wy, wz, __int_as_float(widx), 0.f);
      }
      ASTAMP(6);
#ifdef DH3D_FPS_PROBE
      if (tid == 64 * DH3D_FPS_PROBE_WAVE && blockIdx.x == 0) {
        g_probe[16] += at[0] - pt[1];
        for (int i = 0; i < 6; ++i) g_probe[17 + i] += at[i + 1] - at[i];
      }
#endif
      }
    }
    STAMP(2);
    if (r >= m) break;
    __syncthreads();
    STAMP(3);
    // 2. judge the candidates: lane (row, p) of a judging wave compares candidate c = wave*CPJ + row with candidate p
#if defined(DH3D_FPS_EXP) && (DH3D_FPS_EXP & 2)  // timing experiment: no judging (results wrong)
    if (wave < JW && r < 8) {
#else
    if (wave < JW) {
#endif
      const int row = lane >> 4, p = lane & 15;
      const int c = wave * CPJ + row;
      const bool valid = row < CPJ && c < WAVES;
      const float4 c0 = s_ent[2 * (c < WAVES ? c : 0)], c1 = s_ent[2 * (c < WAVES ? c : 0) + 1];
      const float4 p0 = s_ent[2 * (p < WAVES ? p : 0)], p1 = s_ent[2 * (p < WAVES ? p : 0) + 1];
      STAMP(7);
      const float vc = c0.y;
      const unsigned long long key_c = ((unsigned long long)__float_as_uint(c0.y) << 32) | __float_as_uint(c0.x);
      const unsigned long long key_p = ((unsigned long long)__float_as_uint(p0.y) << 32) | __float_as_uint(p0.x);
      const bool gt = valid && p < WAVES && key_p > key_c;
      const float dx = c0.w - p0.w, dy = c1.x - p1.x, dz = c1.y - p1.y;
      const float d = __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
      const bool aff = d < vc || p0.z >= vc;
      const unsigned rowg = (unsigned)(__ballot(gt) >> (16 * row)) & 0xffffu;
      const unsigned rowa = (unsigned)(__ballot(gt && aff) >> (16 * row)) & 0xffffu;
      const int rank = __popc(rowg);
      // first rank that must wait for the next sync: my row's, if an outranking candidate interferes with it
      const int stop = valid && rank > 0 && (rowa != 0u || !(vc > 0.f)) ? rank : kBig;
      STAMP(8);
      const int stop4 = min(min(__builtin_amdgcn_readlane(stop, 0), __builtin_amdgcn_readlane(stop, 16)),
                            min(__builtin_amdgcn_readlane(stop, 32), __builtin_amdgcn_readlane(stop, 48)));
      if (lane == 0) s_jthr[wave] = stop4;
      if (p == 0 && valid && key_c != 0ull && rank < CAP) s_pick[rank] = make_float4(c0.w, c1.x, c1.y, c1.z);
    }
    STAMP(4);
    __syncthreads();
    STAMP(5);
    // 3. the accepted picks
    {
      const int4 jt = *reinterpret_cast<const int4 *>(s_jthr);
      const float4 pkv = s_pick[pk < CAP ? pk : 0];
      const int thr = min(min(jt.x, jt.y), min(jt.z, jt.w));
#if defined(DH3D_FPS_EXP)  // timing experiments: one pick per sync whatever the judges said
      npick = min(min(thr, 1), m - r);
#else
      npick = min(min(thr, CAP), m - r);
#endif
      qx = pkv.x; qy = pkv.y; qz = pkv.z;
      if (wave == 0 && bl == 0 && pk < npick) s_out[r + pk] = __float_as_int(pkv.w);
      r += npick;
    }
#ifdef DH3D_FPS_PROBE
    STAMP(6);
    if (tid == 64 * DH3D_FPS_PROBE_WAVE && blockIdx.x == 0) {
      for (int i = 0; i < 6; ++i) g_probe[i] += pt[i + 1] - pt[i];
      if (wave < JW) { g_probe[6] += pt[7] - pt[3]; g_probe[7] += pt[8] - pt[7]; g_probe[8] += pt[4] - pt[8]; }
      g_probe[14] += need != 0ull;
      g_probe[15] += 1;
    }
#endif
  }
  __syncthreads();
  for (int i = tid; i < m; i += 64 * WAVES) out[(size_t)b * m + i] = s_out[i];
  if (xyz_out) {  // the sampled coordinates too (group_point of the xyz, core/tf_utils.py:92-95): they are in LDS
    float *xo = xyz_out + (size_t)b * m * 3;
    for (int e = tid; e < 3 * m; e += 64 * WAVES) {
      const int i = e / 3, c = e - 3 * i, k = s_out[i];
      xo[e] = TABLE ? (c == 0 ? s_x[k] : c == 1 ? s_y[k] : s_z[k]) : pc[(size_t)k * 3 + c];
    }
  }
}

template <int PPT, int WAVES>
int fps_batched_launch(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out, float *xyz_out,
                       const float *xyz, hipStream_t s) {
  const size_t small = sizeof(float) * (8 * WAVES + 64 + 4 + (size_t)m);
  const size_t lds = small + sizeof(float) * (size_t)3 * N;
  if (lds <= 159 * 1024) {
    DH3D_ALLOW_BIG_LDS((fps_batched_kernel<PPT, WAVES, true>));
    hipLaunchKernelGGL((fps_batched_kernel<PPT, WAVES, true>), dim3(B), dim3(64 * WAVES), lds, s,
                       reinterpret_cast<const float4 *>(sorted), gbox, N, m, out, xyz_out, nullptr);
  } else {
    if (!xyz || small > 159 * 1024) return DH3D_ERR_UNSUPPORTED;  // no LDS table: the cloud itself is needed
    DH3D_ALLOW_BIG_LDS((fps_batched_kernel<PPT, WAVES, false>));
    hipLaunchKernelGGL((fps_batched_kernel<PPT, WAVES, false>), dim3(B), dim3(64 * WAVES), small, s,
                       reinterpret_cast<const float4 *>(sorted), gbox, N, m, out, xyz_out, xyz);
  }
  return dh3d_launch_status();
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

// ---------------------------------------------------------------------------------------------------
// Any-N kernel with the running min-distances in the caller's `temp` scratch, as the reference keeps them
// (tf_sampling_g.cu:112-145: temp[blockIdx.x*n+k]); serves (a) clouds of more than 16384 points, which the
// register-resident kernels above do not hold, and (b) the UNCONTRACTED distance arithmetic
// ((dx*dx + dy*dy) + dz*dz, what nvcc -fmad=false would build) for integrators whose reference binary
// was compiled that way -- CONTRACT selects between the two roundings, everything else (1e38 start,
// min, strict '>' in index order per 512-stride lane, lower slot on tree ties) is the same rule.
// One 1024-lane workgroup per cloud; a lane scans k = tid, tid+1024, ...; the block arg-max is one 64-bit
// LDS atomicMax per wave on (bits(value) << 32 | ~key(k)), slots rotating over three rounds.
template <bool CONTRACT>
__global__ __launch_bounds__(1024) void fps_anyn_kernel(const float *__restrict__ xyz, int N, int m,
                                                        float *__restrict__ temp, int32_t *__restrict__ out) {
  __shared__ unsigned long long s_best[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float *pc = xyz + (size_t)b * N * 3;
  float *td = temp + (size_t)b * N;
  for (int k = tid; k < N; k += 1024) td[k] = 1e38f;
  if (tid == 0) out[(size_t)b * m] = 0;
  if (tid < 3) s_best[tid] = 0ull;
  __syncthreads();
  int old = 0;
  for (int r = 1; r < m; ++r) {
    const float x1 = pc[(size_t)old * 3], y1 = pc[(size_t)old * 3 + 1], z1 = pc[(size_t)old * 3 + 2];
    float best = -1.f;
    int bkey = INT_MAX;
    for (int k = tid; k < N; k += 1024) {
      const float dx = pc[(size_t)k * 3] - x1, dy = pc[(size_t)k * 3 + 1] - y1, dz = pc[(size_t)k * 3 + 2] - z1;
      const float d = CONTRACT ? __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy)) : (dx * dx + dy * dy) + dz * dz;
      const float t0 = td[k];
      const float d2 = d < t0 ? d : t0;
      if (d2 != t0) td[k] = d2;
      const int key = fps_key(k);
      if (d2 > best || (d2 == best && key < bkey)) { best = d2; bkey = key; }
    }
    if (best >= 0.f) {
      const unsigned long long v =
          ((unsigned long long)(unsigned)__float_as_int(best) << 32) | (unsigned)(~bkey);
      atomicMax(&s_best[r % 3], v);
    }
    __syncthreads();
    const unsigned long long w = s_best[r % 3];
    // slot (r+2)%3 == (r-1)%3 was last read before this round's barrier and is written again only after the
    // next one: safe to clear here
    if (tid == 0) s_best[(r + 2) % 3] = 0ull;
    old = fps_unkey(~(int)(unsigned)(w & 0xffffffffu));
    if (tid == 0) out[(size_t)b * m + r] = old;
  }
}

}  // namespace

// Dev knob (tools/geo_bench.py): waves per cloud; 0 = default.
static int g_fps_waves = 0;
DH3D_API void dh3d_dev_set_fps_waves(int w) { g_fps_waves = w; }

DH3D_API int dh3d_farthest_point_sample_mode(int B, int N, int m, const float *inp, float *temp, int32_t *out,
                                             int contract, void *stream) {
  DH3D_REQUIRE(inp && out && temp && B > 0 && N > 0 && m > 0 && (contract == 0 || contract == 1));
  hipStream_t s = (hipStream_t)stream;
  if (contract) hipLaunchKernelGGL(fps_anyn_kernel<true>, dim3(B), dim3(1024), 0, s, inp, N, m, temp, out);
  else hipLaunchKernelGGL(fps_anyn_kernel<false>, dim3(B), dim3(1024), 0, s, inp, N, m, temp, out);
  return dh3d_launch_status();
}

DH3D_API int dh3d_farthest_point_sample(int B, int N, int m, const float *inp, float *temp,
                                        int32_t *out, void *stream) {
  DH3D_REQUIRE(inp && out && B > 0 && N > 0 && m > 0);  // tf_sampling.cpp:100,105
  if (N > 16384) {  // beyond the register-resident kernels: distances in the caller's scratch, as upstream
    DH3D_SUPPORTED(temp != nullptr);
    return dh3d_farthest_point_sample_mode(B, N, m, inp, temp, out, 1, stream);
  }
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
static int g_fps_sorted_mode = 0;  // 0 = batched rounds, 1 = one pick per round
DH3D_API void dh3d_dev_set_fps_sorted_waves(int w) { g_fps_sorted_waves = w; }
DH3D_API void dh3d_dev_set_fps_sorted_mode(int v) { g_fps_sorted_mode = v; }

static int fps_sorted_dispatch(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out,
                               float *xyz_out, const float *xyz, void *stream) {
  DH3D_REQUIRE(sorted && gbox && out && B > 0 && N > 0 && m > 0);
  // the by-original-index coordinate table must fit LDS (12 B / point) unless the cloud itself is given
  DH3D_SUPPORTED(N <= 12288 || (xyz && N <= 16384 && g_fps_sorted_mode == 0));
  hipStream_t s = (hipStream_t)stream;
  const int W = g_fps_sorted_waves ? g_fps_sorted_waves : 16;  // measured best on MI355X (tools/geo_bench.py)
  const int NG = (N + 63) / 64;
#define DH3D_FPS_CASE(WV)                                                                             \
  if (W == WV) {                                                                                      \
    const int gpw = (NG + WV - 1) / WV; /* groups per wave */                                         \
    if (g_fps_sorted_mode == 0) {                                                                     \
      if (gpw <= 1) return fps_batched_launch<1, WV>(sorted, gbox, B, N, m, out, xyz_out, xyz, s);         \
      if (gpw <= 2) return fps_batched_launch<2, WV>(sorted, gbox, B, N, m, out, xyz_out, xyz, s);         \
      if (gpw <= 4) return fps_batched_launch<4, WV>(sorted, gbox, B, N, m, out, xyz_out, xyz, s);         \
      if (gpw <= 8) return fps_batched_launch<8, WV>(sorted, gbox, B, N, m, out, xyz_out, xyz, s);         \
      if (gpw <= 16) return fps_batched_launch<16, WV>(sorted, gbox, B, N, m, out, xyz_out, xyz, s);       \
      if (gpw <= 32) return fps_batched_launch<32, WV>(sorted, gbox, B, N, m, out, xyz_out, xyz, s);       \
      if (gpw <= 48) return fps_batched_launch<48, WV>(sorted, gbox, B, N, m, out, xyz_out, xyz, s);       \
    }                                                                                                 \
    if (gpw <= 1) return fps_sorted_launch<1, WV>(sorted, gbox, B, N, m, out, s);                     \
    if (gpw <= 2) return fps_sorted_launch<2, WV>(sorted, gbox, B, N, m, out, s);                     \
    if (gpw <= 4) return fps_sorted_launch<4, WV>(sorted, gbox, B, N, m, out, s);                     \
    if (gpw <= 8) return fps_sorted_launch<8, WV>(sorted, gbox, B, N, m, out, s);                     \
    if (gpw <= 16) return fps_sorted_launch<16, WV>(sorted, gbox, B, N, m, out, s);                   \
    if (gpw <= 32) return fps_sorted_launch<32, WV>(sorted, gbox, B, N, m, out, s);                   \
    if (gpw <= 48) return fps_sorted_launch<48, WV>(sorted, gbox, B, N, m, out, s);                   \
  }
  if (xyz_out && g_fps_sorted_mode != 0) return DH3D_ERR_UNSUPPORTED;  // the one-pick-per-round kernel has no such output
  DH3D_FPS_CASE(4)
  DH3D_FPS_CASE(8)
  DH3D_FPS_CASE(16)
#undef DH3D_FPS_CASE
  return DH3D_ERR_UNSUPPORTED;
}

DH3D_API int dh3d_fps_sorted(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out,
                             void *stream) {
  return fps_sorted_dispatch(sorted, gbox, B, N, m, out, nullptr, nullptr, stream);
}

// + xyz_out [B, m, 3]: the sampled coordinates (what group_point of the cloud by `out` returns), written by the same
// kernel from its LDS copy of the cloud
DH3D_API int dh3d_fps_sorted_xyz(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out,
                                 float *xyz_out, void *stream) {
  DH3D_REQUIRE(xyz_out);
  return fps_sorted_dispatch(sorted, gbox, B, N, m, out, xyz_out, nullptr, stream);
}

// + xyz [B, N, 3]: the cloud the records were sorted from.  Lifts the 12288-point limit of the LDS coordinate table
// to the 16384 of the ordering itself (the winner's coordinates are then read from `xyz`); xyz_out may be NULL.
DH3D_API int dh3d_fps_sorted_cloud(const float *sorted, const float *gbox, const float *xyz, int B, int N, int m,
                                   int32_t *out, float *xyz_out, void *stream) {
  DH3D_REQUIRE(xyz);
  return fps_sorted_dispatch(sorted, gbox, B, N, m, out, xyz_out, xyz, stream);
}

#ifdef DH3D_FPS_PROBE
DH3D_API int dh3d_fps_cnt_read(unsigned long long *host2, int reset) {
  int rc = hipMemcpyFromSymbol(host2, HIP_SYMBOL(g_fps_cnt), 16) == hipSuccess ? 0 : 3;
  if (reset) { unsigned long long z[2] = {0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fps_cnt), z, 16); }
  return rc;
}
DH3D_API int dh3d_fps_probe_read(long long *host16) {
  return hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_probe), sizeof(long long) * 32) == hipSuccess ? 0 : 3;
}
#endif
