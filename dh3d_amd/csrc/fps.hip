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

#ifdef DH3D_FPS_PROBE  // dev instrumentation (tools/fps_list_phases.py): per-phase cycle sums of one wave of cloud 0
__device__ long long g_probe[32];
#ifndef DH3D_FPS_PROBE_WAVE
#define DH3D_FPS_PROBE_WAVE 0
#endif
#define STAMP(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); pt[i] = clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
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
    float x1, y1, z1;
    if (LDS_COORDS) { x1 = s_x[old]; y1 = s_y[old]; z1 = s_z[old]; }
    else { x1 = pc[(size_t)old * 3]; y1 = pc[(size_t)old * 3 + 1]; z1 = pc[(size_t)old * 3 + 2]; }
    const f32x2 x2 = {x1, x1}, y2 = {y1, y1}, z2 = {z1, z1};
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
    const float wmax = wave_max_f32(best);
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
    if (lane == 0) {
      const unsigned long long packed =
          wmax >= 0.f ? (((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned)~wkey) : 0ull;
      atomicMax(&s_best[buf], packed);
      if (wave == 0) s_best[buf == 2 ? 0 : buf + 1] = 0ull;  // next round's slot (last read two rounds ago)
    }
    __syncthreads();
    const int bkey = ~(int)(unsigned)s_best[buf];
    old = fps_unkey(bkey);
    if (tid == 0) { if (LDS_COORDS) s_out[r] = old; else out[(size_t)b * m + r] = old; }
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (LDS_COORDS) {
    __syncthreads();
    for (int r = tid; r < m; r += T) out[(size_t)b * m + r] = s_out[r];
  }
}

// ------------------------------------------------------------------------------------------------
// FPS on a spatially ordered cloud (spatial.hip): candidate LISTS + a sequential judge, ~10 picks per
// synchronisation, same picks as the sequential rule.
//
// Wave w owns the Morton-consecutive groups [w*PPT, (w+1)*PPT) (64 points each: a compact region of the cloud); lane l
// holds point l of each of them, coordinates and running min-distance in registers.  A point's distance can only drop
// if the new sample is closer to it than that distance, hence -- for a whole group -- only if the sample is closer to
// the group's bounding box than the largest distance in the wave: box tests (lane = pick * PPT + box, several picks
// per pass) decide which waves update at all.  (Rounds 1-2: one pick per barrier, 0.68 ms for 8 x 8192 -> 1024; then
// one published candidate per wave judged pairwise, 3.7 picks per sync, 0.39 ms.)
//
// What ended the batches of the one-candidate version was mostly "the region of a picked candidate may hold a better
// point than the next candidate" -- a statement about points nobody had published.  Publish them.  Every touched wave
// lists its arg-max AND the other lanes whose best point lies within a per-wave adaptive margin of it (at most
// L = 64 / WAVES entries: value + tie key), plus ONE bound: the largest running distance among everything it did NOT
// list.  RB = the largest of the WAVES bounds.  The judge (wave 0, one listed candidate per lane, coordinates from the
// LDS table) then simply RUNS the sequential algorithm on the pool: arg-max by (value, smallest key) -> pick -> pool
// values = min(value, distance to the pick, the update's own arithmetic) -> ... and every pick whose value is still
// STRICTLY above RB is the pick the full algorithm would make: an unlisted point started at or below RB and running
// distances only drop.  The first pick of a sync needs no test (the pool holds every wave's exact arg-max).
// Exact under ties: a lane whose two best points tie reports the second one in the bound, so a tied value is never
// accepted past rank 0, and rank 0 is decided on the waves' exact (value, key) winners.
// 8 x 8192 -> 1024: 100 syncs of 10.2 picks, 6.8 k cycles each (update + list 3.2 k on the slowest wave, judge 3.5 k =
// 294 cycles per pick); 0.355 ms against 0.388.  DEADENDS.md: the same lists with picks streamed to worker waves through
// an LDS ring instead of barriers (slower: the workers' instruction issue, four waves per SIMD, is the bottleneck).
// TABLE: the by-original-index coordinate table lives in LDS (12 B per point: clouds of up to ~12 k points).  Larger
// clouds (TABLE = false) read candidate coordinates from the cloud itself (`xyz`, L2-resident).
// The sampled set in the cloud's own Morton order, written by the kernel's tail (round 6): the picks are a SUBSET of an
// already sorted cloud, so a stable compaction of the picked positions is their spatial order -- records (x, y, z, pick
// rank), one box per 64 records and (with the cloud's cell table given) the subset's cell table on the cloud's grid, i.e.
// everything dh3d_spatial_sort_cells(sampled xyz) would hand to three_nn_sorted / knn_grid, without the second sort's
// launch on the chain behind the sampling.  A point picked more than once (a cloud with fewer distinct points than picks)
// yields one record per pick.
struct FpsOrderedOut {
  float4 *sorted_s;    // [B, m] or NULL: nothing below is written
  float *gbox_s;       // [B, ceil(m / 64), 8]
  const int *cells;    // [B, DH3D_CELL_INTS] of the cloud, or NULL
  int *cells_s;        // [B, DH3D_CELL_INTS] of the sampled set (iff cells)
  int occ_min;         // fewer occupied cells than this: cells_s[4106] = 1 (spatial.hip's rule for a set of m points)
};

__device__ __forceinline__ unsigned f32_ordered(float f) {  // monotonic float -> unsigned (for LDS atomicMin / atomicMax)
  const unsigned u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(unsigned u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

template <int PPT, int WAVES, bool TABLE>
__global__ __launch_bounds__(64 * WAVES) void fps_list_kernel(const float4 *__restrict__ sorted,
                                                            const float *__restrict__ gbox, int N, int m,
                                                            int32_t *__restrict__ out, float *__restrict__ xyz_out,
                                                            const float *__restrict__ xyz, FpsOrderedOut oo) {
  static_assert(PPT <= 32 && WAVES <= 16 && 64 % WAVES == 0, "one lane per (pick, box) pair, one lane per pool entry");
  constexpr int L = 64 / WAVES;                       // list entries per wave: the pool is one entry per judge lane
  constexpr int PP = 64 / PPT < 16 ? 64 / PPT : 16;   // picks box-tested per pass
  constexpr int CAP = 32;                             // picks per sync
  constexpr int NP = (PPT + 1) / 2;                   // groups are held and updated in pairs (packed f32)
  constexpr int LO = L >= 4 ? L / 2 : 1, HI = L >= 4 ? L - 1 : L;  // keep the listed count in [LO, HI]
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  // picks [32] x,y,z,index | pool [64] (value, key) | bounds [16] | npick | table | picks
  float4 *s_pick = reinterpret_cast<float4 *>(s_mem);
  float2 *s_list = reinterpret_cast<float2 *>(s_pick + CAP);
  float *s_rb = reinterpret_cast<float *>(s_list + 64);
  int *s_np = reinterpret_cast<int *>(s_rb + 16);
  float *s_x = reinterpret_cast<float *>(s_np + 4);
  float *s_y = s_x + (TABLE ? N : 0);
  float *s_z = s_y + (TABLE ? N : 0);
  int *s_out = reinterpret_cast<int *>(s_z + (TABLE ? N : 0));
  const float *pc = TABLE ? nullptr : xyz + (size_t)blockIdx.x * N * 3;

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
  // lane = pick * PPT + box: every pick slot of a pass sees the wave's PPT boxes
  const int bl = lane % PPT, pk = lane / PPT;
  float blx = INFINITY, bly = INFINITY, blz = INFINITY, bhx = -INFINITY, bhy = -INFINITY, bhz = -INFINITY;
  bool has_box = false;
  if (pk < PP) {
    const int g = wave * PPT + bl;
    if (g < NG) {
      const float *bx = gbox + ((size_t)b * NG + g) * 8;
      blx = bx[0]; bly = bx[1]; blz = bx[2]; bhx = bx[4]; bhy = bx[5]; bhz = bx[6];
      has_box = true;
    }
  }
  if (tid == 0) {
    s_out[0] = 0;
    s_np[0] = 1;
  }
  if (lane < L) s_list[wave * L + lane] = make_float2(-2.f, __int_as_float(INT_MAX));
  if (lane == 0) s_rb[wave] = -2.f;
  __syncthreads();  // the table
  if (tid == 0) {
    float x0, y0, z0;
    if (TABLE) { x0 = s_x[0]; y0 = s_y[0]; z0 = s_z[0]; }
    else { x0 = pc[0]; y0 = pc[1]; z0 = pc[2]; }
    s_pick[0] = make_float4(x0, y0, z0, 0.f);
  }
  __syncthreads();

  float wmax = __ballot(has_box) != 0ull ? 1e38f : -2.f;  // cached wave maximum (uniform)
  float delta = 0.05f;                                      // listing margin, relative to the wave maximum
  int npick = 1, r = 1;
  constexpr unsigned long long kPickMask = PPT == 64 ? ~0ull : ((1ull << (PPT & 63)) - 1ull);
#ifdef DH3D_FPS_PROBE
  long long pt[12];
#endif
  while (r < m) {
    STAMP(0);
    // 1. box tests, PP picks per pass: can pick p change anything in group g?  Then the updates.
    bool touched = false;
    for (int p0 = 0; p0 < npick; p0 += PP) {
      const float4 q = s_pick[min(p0 + pk, CAP - 1)];
      const float ex = fmaxf(fmaxf(blx - q.x, q.x - bhx), 0.f);
      const float ey = fmaxf(fmaxf(bly - q.y, q.y - bhy), 0.f);
      const float ez = fmaxf(fmaxf(blz - q.z, q.z - bhz), 0.f);
      const float bd = (ex * ex + ey * ey + ez * ez) * 0.99999f;
      unsigned long long nd = __ballot(has_box && p0 + pk < npick && bd <= wmax);
      touched |= nd != 0ull;
      while (nd != 0ull) {  // the picks of this pass that reach the wave
        const int p = __builtin_ctzll(nd) / PPT;
        const unsigned long long nb = (nd >> (p * PPT)) & kPickMask;
        nd &= ~(kPickMask << (p * PPT));
        const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.x), p * PPT));
        const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.y), p * PPT));
        const float z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q.z), p * PPT));
        const f32x2 x2 = {x1, x1}, y2 = {y1, y1}, z2 = {z1, z1};
#pragma unroll
        for (int qd = 0; qd < NP; ++qd) {
          // (a pair of groups the pick does not reach is skipped -- a scalar branch per pair against eight packed
          //  instructions at four waves per SIMD: 8 x 8192 353.6 -> 336.6 us, round 6; rounds 3-5 updated all pairs up to PPT = 8)
          if (PPT <= 4 || ((nb >> (2 * qd)) & 3ull)) {
            const f32x2 dx = px[qd] - x2, dy = py[qd] - y2, dz = pz[qd] - z2;
            const f32x2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
            md[qd][0] = __builtin_fminf(d[0], md[qd][0]);
            md[qd][1] = __builtin_fminf(d[1], md[qd][1]);
          }
        }
      }
    }
    STAMP(1);
    if (touched) {  // wave-uniform: new arg-max, list and bound (an untouched wave's published state stays valid)
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
      wmax = wave_max_f32(b1);
      const unsigned long long hit = __ballot(b1 == wmax);
      // the listing margin.  One adaptive margin lists 2-3 lanes on average (it must stay clear of L + 1).  From 8 groups
      // per wave on, three margins are probed at once (independent compares: delta, 1.5 delta, 2.25 delta) and the widest
      // one that still lists at most L lanes, none with two points inside it, is taken: fuller lists, a lower bound, more
      // picks per sync (8 x 8192: 335.4 -> 330.0 us, 4 x 16384: 739 -> 723; at 4 groups per wave 168.5 -> 170.6: off there).
      // `cnt`, which steers delta, stays the NARROW probe's count.
      const float tau0 = wmax - delta * wmax;
      unsigned long long flag = __ballot(b1 > tau0);
      unsigned long long two = __ballot(b2 > tau0);  // two points of a lane inside the margin
      const int cnt = __popcll(flag);
      if (PPT >= 8) {
        const float tau1 = wmax - (1.5f * delta) * wmax, tau2 = wmax - (2.25f * delta) * wmax;
        const unsigned long long f1 = __ballot(b1 > tau1), f2 = __ballot(b1 > tau2);
        const unsigned long long t1 = __ballot(b2 > tau1), t2 = __ballot(b2 > tau2);
        const bool ok2 = __popcll(f2) <= L && t2 == 0ull, ok1 = __popcll(f1) <= L && t1 == 0ull;
        flag = ok2 ? f2 : (ok1 ? f1 : flag);
        two = ok2 ? t2 : (ok1 ? t1 : two);
      }
      const int cntl = __popcll(flag);
      if (__popcll(hit) == 1 && two == 0ull && cntl >= 1 && cntl <= L) {
        // the usual case: one point holds the maximum, every lane inside the margin has ONE point there and they all
        // fit: list them in lane order; everything else is at or below tau
        const int slot = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(flag >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)flag, 0u));
        if (lane < L) s_list[wave * L + lane] = make_float2(-2.f, __int_as_float(INT_MAX));
        if ((flag >> lane) & 1ull) s_list[wave * L + slot] = make_float2(b1, __int_as_float(lkey));
        const float rb = wave_max_f32(((flag >> lane) & 1ull) ? b2 : b1);  // the largest value NOT listed
        if (lane == 0) s_rb[wave] = rb;
      } else {
        int wl = __builtin_ctzll(hit);  // winner lane
        int wkey;
        if (__popcll(hit) == 1 && __ballot(b2 == wmax) == 0ull) {  // one point holds the maximum
          wkey = __builtin_amdgcn_readlane(lkey, wl);
        } else {  // ties: the smallest key wins
          int tkey = INT_MAX;
#pragma unroll
          for (int j = 0; j < 2 * NP; ++j) tkey = (md[j >> 1][j & 1] == wmax) ? min(tkey, pkey[j]) : tkey;
          wkey = wave_min_i32(tkey);
          wl = __builtin_ctzll(__ballot(tkey == wkey));
        }
        // winner in slot 0, then the lanes within the margin in lane order; the rest goes into the bound
        const unsigned long long others = flag & ~(1ull << wl);
        const int slot = lane == wl ? 0 : 1 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(others >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)others, 0u));
        const bool listed = lane == wl || (((others >> lane) & 1ull) && slot < L);
        const float rb = wave_max_f32(listed ? b2 : b1);
        if (wmax >= 0.f) {
          if (lane < L) s_list[wave * L + lane] = make_float2(-2.f, __int_as_float(INT_MAX));
          if (listed) s_list[wave * L + slot] = make_float2(b1, __int_as_float(lane == wl ? wkey : lkey));
          if (lane == 0) s_rb[wave] = rb;
        }
      }
      delta = cnt > HI ? delta * 0.7f : (cnt < LO ? fminf(delta * 1.3f, 0.5f) : delta);
    }
    STAMP(2);
    __syncthreads();
    STAMP(3);
    // 2. the judge: sequential FPS on the pool
    if (wave == 0) {
      const float2 e = s_list[lane];
      float cv = e.x;
      const int ckey = __float_as_int(e.y);
      const int cidx = cv >= 0.f ? fps_unkey(ckey) : 0;
      const float rbl = s_rb[lane & 15];
      float cx, cy, cz;
      if (TABLE) { cx = s_x[cidx]; cy = s_y[cidx]; cz = s_z[cidx]; }
      else { cx = pc[(size_t)cidx * 3]; cy = pc[(size_t)cidx * 3 + 1]; cz = pc[(size_t)cidx * 3 + 2]; }
      // non-negative floats order like their bit patterns: the judge compares on the scalar unit
      const int RB = __builtin_amdgcn_readfirstlane(max(__float_as_int(wave_max_f32(lane < WAVES ? rbl : -2.f)), -1));
      const int cap = __builtin_amdgcn_readfirstlane(min(CAP, m - r));
      int kl = 0, k = 0;  // lane k of kl: the pool lane of pick k
      int rbe = -1;  // the first pick of a sync is unconditional
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(cx), "+v"(cy), "+v"(cz), "+v"(cv));
      STAMP(7);
      while (true) {
        // One pick per trip, hand-scheduled (compiled: 57 instructions and three taken branches, 375 cycles per pick).
        // The pool lane of pick k goes into lane k of kl with v_writelane; its record is fetched after the loop.
        // status: 0 = the pool ran dry (or the sync is full), 1 = the maximum is tied (resolved below, on the keys)
        int st, sv, sl, sx, sy, sz, sm0;
        float t0, t1, t2;
        asm volatile(
            "s_mov_b32 %[sm0], m0\n\t"
            "1:\n\t"
            "v_mov_b32 %[t0], %[cv]\n\t"
            DH3D_DPP_WAVE_N("v_max_f32_dpp", "t0")
            "s_nop 1\n\t"  // VALU write -> v_readlane of the same VGPR
            "v_readlane_b32 %[sv], %[t0], 63\n\t"
            "s_mov_b32 %[st], 0\n\t"
            "s_cmp_gt_i32 %[sv], %[rbe]\n\t"
            "s_cbranch_scc0 9f\n\t"
            "s_mov_b32 %[rbe], %[RB]\n\t"
            "v_cmp_eq_u32 vcc, %[sv], %[cv]\n\t"
            "s_bcnt1_i32_b64 %[sl], vcc\n\t"
            "s_mov_b32 %[st], 1\n\t"
            "s_cmp_eq_u32 %[sl], 1\n\t"
            "s_cbranch_scc0 9f\n\t"
            "s_ff1_i32_b64 %[sl], vcc\n\t"
            "s_mov_b32 %[st], 0\n\t"
            "s_mov_b32 m0, %[k]\n\t"
            "s_nop 1\n\t"  // SALU write of a v_readlane lane select: 4 wait states
            "v_readlane_b32 %[sx], %[cx], %[sl]\n\t"
            "v_readlane_b32 %[sy], %[cy], %[sl]\n\t"
            "v_readlane_b32 %[sz], %[cz], %[sl]\n\t"
            "v_subrev_f32 %[t1], %[sx], %[cx]\n\t"
            "v_subrev_f32 %[t0], %[sy], %[cy]\n\t"
            "v_subrev_f32 %[t2], %[sz], %[cz]\n\t"
            "v_mul_f32 %[t0], %[t0], %[t0]\n\t"
            "v_fmac_f32 %[t0], %[t1], %[t1]\n\t"
            "v_fmac_f32 %[t0], %[t2], %[t2]\n\t"
            "v_min_f32 %[cv], %[t0], %[cv]\n\t"
            "v_writelane_b32 %[kl], %[sl], m0\n\t"  // m0 = k: two SGPR operands would exceed the constant bus
            "s_add_u32 %[k], %[k], 1\n\t"
            "s_cmp_lg_u32 %[k], %[cap]\n\t"
            "s_cbranch_scc1 1b\n\t"
            "9:\n\t"
            "s_mov_b32 m0, %[sm0]\n\t"
            : [cv] "+v"(cv), [k] "+s"(k), [rbe] "+s"(rbe), [kl] "+v"(kl),
              [sm0] "=&s"(sm0), [st] "=&s"(st), [sv] "=&s"(sv), [sl] "=&s"(sl), [sx] "=&s"(sx), [sy] "=&s"(sy), [sz] "=&s"(sz),
              [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
            : [cx] "v"(cx), [cy] "v"(cy), [cz] "v"(cz), [RB] "s"(RB), [cap] "s"(cap)
            : "vcc", "scc");
        if (st == 0) break;
        // tied values in the pool: the smallest key wins (the test against the bound has passed)
        const int kmin = wave_min_i32(__float_as_int(cv) == sv ? ckey : INT_MAX);
        const int l = __builtin_ctzll(__ballot(__float_as_int(cv) == sv && ckey == kmin));
        const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), l));
        const float y1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), l));
        const float z1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), l));
        if (lane == k) kl = l;
        const float dx = cx - x1, dy = cy - y1, dz = cz - z1;
        const float d = __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
        cv = __builtin_fminf(d, cv);
        if (++k == cap) break;
      }
      STAMP(8);
      {  // lane k fetches pick k's record from its pool lane (ds_bpermute: no LDS memory involved)
        const float kx = __int_as_float(__builtin_amdgcn_ds_bpermute(kl << 2, __float_as_int(cx)));
        const float ky = __int_as_float(__builtin_amdgcn_ds_bpermute(kl << 2, __float_as_int(cy)));
        const float kz = __int_as_float(__builtin_amdgcn_ds_bpermute(kl << 2, __float_as_int(cz)));
        const int ki = __builtin_amdgcn_ds_bpermute(kl << 2, cidx);
        if (lane < k) {
          s_pick[lane] = make_float4(kx, ky, kz, __int_as_float(ki));
          s_out[r + lane] = ki;
        }
      }
      if (lane == 0) s_np[0] = k;
    }
    STAMP(4);
    __syncthreads();
    STAMP(5);
    npick = s_np[0];
    r += npick;
#ifdef DH3D_FPS_PROBE
    if (tid == 64 * DH3D_FPS_PROBE_WAVE && blockIdx.x == 0) {
      for (int i = 0; i < 5; ++i) g_probe[i] += pt[i + 1] - pt[i];
      if (wave == 0) { g_probe[6] += pt[7] - pt[3]; g_probe[7] += pt[8] - pt[7]; g_probe[8] += pt[4] - pt[8]; }
      g_probe[13] += npick;
      g_probe[14] += touched;
      g_probe[15] += 1;
    }
#endif
  }
  __syncthreads();
  for (int i = tid; i < m; i += 64 * WAVES) out[(size_t)b * m + i] = s_out[i];
  if (xyz_out) {
    float *xo = xyz_out + (size_t)b * m * 3;
    for (int e = tid; e < 3 * m; e += 64 * WAVES) {
      const int i = e / 3, c = e - 3 * i, k = s_out[i];
      xo[e] = TABLE ? (c == 0 ? s_x[k] : c == 1 ? s_y[k] : s_z[k]) : pc[(size_t)k * 3 + c];
    }
  }
  if constexpr (TABLE) {
    if (oo.sorted_s) {  // (uniform) the sampled set in Morton order: see FpsOrderedOut
      constexpr int T = 64 * WAVES;
      const int NGs = (m + 63) / 64;
      unsigned *s_cnt = reinterpret_cast<unsigned *>(s_out + m);                      // [N + 1] picks per sorted position
      unsigned short *s_pos = reinterpret_cast<unsigned short *>(s_cnt + N + 1);      // [N] original index -> sorted position
      unsigned *s_box = reinterpret_cast<unsigned *>(s_pos + ((N + 1) & ~1));         // [NGs][6] ordered-uint min / max
      unsigned *s_wsum = reinterpret_cast<unsigned *>(s_list);                        // [WAVES + 1] (the pool is dead)
      for (int i = tid; i <= N; i += T) s_cnt[i] = 0u;
      for (int i = tid; i < NGs * 6; i += T) s_box[i] = (i % 6) < 3 ? 0xFFFFFFFFu : 0u;
#pragma unroll
      for (int j = 0; j < 2 * NP; ++j) {
        const int i = (wave * PPT + j) * 64 + lane;
        if (j < PPT && i < N) s_pos[fps_unkey(pkey[j])] = (unsigned short)i;
      }
      if (tid == 0) s_wsum[WAVES] = 0u;
      __syncthreads();
      for (int r = tid; r < m; r += T) atomicAdd(&s_cnt[s_pos[s_out[r]]], 1u);
      __syncthreads();
      // exclusive prefix over the sorted positions (thread t owns the chunk [t * CH, (t + 1) * CH) of [0, N]); s_cnt[N] = m
      const int CH = (N + 1 + T - 1) / T;
      unsigned mine = 0u;
      for (int c = 0; c < CH; ++c) {
        const int i = tid * CH + c;
        if (i <= N) mine += s_cnt[i];
      }
      unsigned inc = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = __shfl_up(inc, off, 64);
        if (lane >= off) inc += up;
      }
      if (lane == 63) s_wsum[wave] = inc;
      __syncthreads();
      unsigned base = inc - mine;
      for (int w = 0; w < wave; ++w) base += s_wsum[w];
      for (int c = 0; c < CH; ++c) {
        const int i = tid * CH + c;
        if (i <= N) {
          const unsigned v = s_cnt[i];
          s_cnt[i] = base;
          base += v;
        }
      }
      __syncthreads();
      if (oo.cells) {  // the subset's cell table on the cloud's grid: cell c opens at the number of picks in front of it
        const int *ct = oo.cells + (size_t)b * DH3D_CELL_INTS;
        int *cs = oo.cells_s + (size_t)b * DH3D_CELL_INTS;
        int occ = 0;
        for (int c = tid; c <= 4096; c += T) {
          const unsigned v = s_cnt[min(max(ct[c], 0), N)];
          cs[c] = (int)v;
          if (c < 4096) occ += (int)(s_cnt[min(max(ct[c + 1], 0), N)] > v);
        }
        for (int c = 4097 + tid; c < DH3D_CELL_INTS; c += T)
          if (c != 4106) cs[c] = ct[c];  // origin, scales, bit schedule: the cloud's grid
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) occ += __shfl_xor(occ, off, 64);
        if (lane == 0) atomicAdd(&s_wsum[WAVES], (unsigned)occ);
      }
      __syncthreads();  // every prefix value has been read: the counters now hand out the slots
      if (oo.cells && tid == 0) oo.cells_s[(size_t)b * DH3D_CELL_INTS + 4106] = (int)s_wsum[WAVES] < oo.occ_min ? 1 : 0;
      float4 *so = oo.sorted_s + (size_t)b * m;
      for (int r = tid; r < m; r += T) {
        const int k = s_out[r];
        const unsigned slot = atomicAdd(&s_cnt[s_pos[k]], 1u);
        const float x = s_x[k], y = s_y[k], z = s_z[k];
        so[slot] = make_float4(x, y, z, __int_as_float(r));
        unsigned *bx = s_box + (slot >> 6) * 6;
        atomicMin(&bx[0], f32_ordered(x)); atomicMin(&bx[1], f32_ordered(y)); atomicMin(&bx[2], f32_ordered(z));
        atomicMax(&bx[3], f32_ordered(x)); atomicMax(&bx[4], f32_ordered(y)); atomicMax(&bx[5], f32_ordered(z));
      }
      __syncthreads();
      for (int g = tid; g < NGs; g += T) {
        float *o = oo.gbox_s + ((size_t)b * NGs + g) * 8;
        const unsigned *bx = s_box + g * 6;
        o[0] = f32_unordered(bx[0]); o[1] = f32_unordered(bx[1]); o[2] = f32_unordered(bx[2]); o[3] = 0.f;
        o[4] = f32_unordered(bx[3]); o[5] = f32_unordered(bx[4]); o[6] = f32_unordered(bx[5]); o[7] = 0.f;
      }
    }
  }
}

template <int PPT, int WAVES>
int fps_list_launch(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out, float *xyz_out,
                    const float *xyz, const FpsOrderedOut &oo, hipStream_t s) {
  const size_t small = sizeof(float) * (4 * 32 + 2 * 64 + 16 + 4 + (size_t)m);
  // (ordered output: picks per sorted position [N + 1] u32, original index -> position [N] u16, boxes [m / 64][6] u32)
  const size_t ordered = oo.sorted_s ? sizeof(unsigned) * ((size_t)N + 1) + sizeof(unsigned short) * (((size_t)N + 1) & ~(size_t)1) +
                                           sizeof(unsigned) * 6 * (((size_t)m + 63) / 64) : 0;
  const size_t lds = small + sizeof(float) * (size_t)3 * N + ordered;
  // The workgroup asks for the WHOLE CU's LDS whatever it needs: this kernel is the step's latency chain (one CU per
  // cloud, every instruction of the judge's chain counts), and a workgroup of another kernel that lands beside it takes
  // issue slots from its sixteen waves.  Same box, steps in flight: local 30.85 k -> 31.17 k clouds/s, one step at a time
  // 0.5053 -> 0.5021 ms (the CUs it keeps to itself were 3 % of the chip per step anyway).
  const size_t whole_cu = (size_t)159 * 1024;
  if (lds <= whole_cu) {
    DH3D_ALLOW_BIG_LDS((fps_list_kernel<PPT, WAVES, true>));
    hipLaunchKernelGGL((fps_list_kernel<PPT, WAVES, true>), dim3(B), dim3(64 * WAVES), whole_cu, s,
                       reinterpret_cast<const float4 *>(sorted), gbox, N, m, out, xyz_out, nullptr, oo);
  } else {
    if (oo.sorted_s) return DH3D_ERR_UNSUPPORTED;  // the ordered output lives beside the LDS coordinate table
    if (!xyz || small > whole_cu) return DH3D_ERR_UNSUPPORTED;  // no LDS table: the cloud itself is needed
    DH3D_ALLOW_BIG_LDS((fps_list_kernel<PPT, WAVES, false>));
    hipLaunchKernelGGL((fps_list_kernel<PPT, WAVES, false>), dim3(B), dim3(64 * WAVES), whole_cu, s,
                       reinterpret_cast<const float4 *>(sorted), gbox, N, m, out, xyz_out, xyz, oo);
  }
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
  // waves per cloud measured best on MI355X: 4 up to 1024 points, 8 above (<= 32 points per lane: no spills)
  if (N <= 512) return fps_launch<2, 4>(inp, B, N, m, out, s);
  if (N <= 1024) return fps_launch<4, 4>(inp, B, N, m, out, s);
  if (N <= 2048) return fps_launch<4, 8>(inp, B, N, m, out, s);
  if (N <= 4096) return fps_launch<8, 8>(inp, B, N, m, out, s);
  if (N <= 8192) return fps_launch<16, 8>(inp, B, N, m, out, s);
  return fps_launch<32, 8>(inp, B, N, m, out, s);
}

static int fps_sorted_dispatch(const float *sorted, const float *gbox, int B, int N, int m, int32_t *out,
                               float *xyz_out, const float *xyz, void *stream, const FpsOrderedOut &oo = FpsOrderedOut{}) {
  DH3D_REQUIRE(sorted && gbox && out && B > 0 && N > 0 && m > 0);
  // the by-original-index coordinate table must fit LDS (12 B / point) unless the cloud itself is given
  DH3D_SUPPORTED(N <= 12288 || (xyz && N <= 16384));
  hipStream_t s = (hipStream_t)stream;
  const int gpw = ((N + 63) / 64 + 15) / 16;  // groups per wave, 16 waves per cloud
  if (gpw <= 1) return fps_list_launch<1, 16>(sorted, gbox, B, N, m, out, xyz_out, xyz, oo, s);
  if (gpw <= 2) return fps_list_launch<2, 16>(sorted, gbox, B, N, m, out, xyz_out, xyz, oo, s);
  if (gpw <= 4) return fps_list_launch<4, 16>(sorted, gbox, B, N, m, out, xyz_out, xyz, oo, s);
  if (gpw <= 8) return fps_list_launch<8, 16>(sorted, gbox, B, N, m, out, xyz_out, xyz, oo, s);
  return fps_list_launch<16, 16>(sorted, gbox, B, N, m, out, xyz_out, xyz, oo, s);
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
// to the 16384 of the ordering itself (candidate coordinates are then read from `xyz`); xyz_out may be NULL.
DH3D_API int dh3d_fps_sorted_cloud(const float *sorted, const float *gbox, const float *xyz, int B, int N, int m,
                                   int32_t *out, float *xyz_out, void *stream) {
  DH3D_REQUIRE(xyz);
  return fps_sorted_dispatch(sorted, gbox, B, N, m, out, xyz_out, xyz, stream);
}

// dh3d_fps_sorted_xyz + the SAMPLED SET IN MORTON ORDER out of the same kernel: sorted_s [B, m, 4] records (x, y, z,
// bits(pick rank)), gbox_s [B, ceil(m / 64), 8], and -- when the cloud's cell table `cells` (dh3d_spatial_sort_cells) is
// given -- cells_s [B, DH3D_CELL_INTS], the subset's table on the cloud's grid: what dh3d_spatial_sort_cells of xyz_out
// would hand to dh3d_three_nn_sorted / dh3d_knn_grid (a valid order + boxes + table, not the identical arrays: the subset
// inherits the cloud's grid and the cloud's order inside a cell).  N <= 8192 (the tail's tables live beside the LDS
// coordinate table); larger clouds: DH3D_ERR_UNSUPPORTED, sort xyz_out instead.
DH3D_API int dh3d_fps_sorted_ordered(const float *sorted, const float *gbox, const int32_t *cells, int B, int N, int m,
                                     int32_t *out, float *xyz_out, float *sorted_s, float *gbox_s, int32_t *cells_s,
                                     void *stream) {
  DH3D_REQUIRE(xyz_out && sorted_s && gbox_s && ((cells == nullptr) == (cells_s == nullptr)));
  DH3D_SUPPORTED(N <= 8192 && N < 65536);
  FpsOrderedOut oo;
  oo.sorted_s = reinterpret_cast<float4 *>(sorted_s);
  oo.gbox_s = gbox_s;
  oo.cells = cells;
  oo.cells_s = cells_s;
  oo.occ_min = (int)(0.6 * 4096.0 * (1.0 - exp(-(double)m / 4096.0)));  // spatial.hip sort_launch's rule for m points
  return fps_sorted_dispatch(sorted, gbox, B, N, m, out, xyz_out, nullptr, stream, oo);
}

#ifdef DH3D_FPS_PROBE
DH3D_API int dh3d_fps_probe_read(long long *host32) {
  return hipMemcpyFromSymbol(host32, HIP_SYMBOL(g_probe), sizeof(long long) * 32) == hipSuccess ? 0 : 3;
}
#endif
