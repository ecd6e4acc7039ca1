// Brute-force exact kNN for gfx950 -- replaces KnnBruteforceFunctor<GPUDevice,float,int>
// (user_ops/kernels/knn_bruteforce_kernel_gpu.cu.cc:36-134,163-228).
//
// The reference sorts all N (distance, id) pairs per query with cub::BlockRadixSort and keeps K.
// Here one lane owns one query and streams every candidate of its cloud from LDS (all 64 lanes read the
// same 16 bytes -> broadcast ds_read_b128).  Per 8 candidates the hot loop is 12 packed-f32 VALU ops
// (v_pk_add/mul/fma: two candidates per instruction) + a min tree + ONE branch: a candidate is screened
// on its squared distance against a conservative bound derived from the lane's current K-th entry.
// Survivors are not inserted on the spot -- with 64 independent queries per wave "some lane has a
// survivor" is true far more often than "this lane has one", and a divergent K-deep insertion per hit
// dominated the first version (profiles/r01_a).  They are appended to a per-lane LDS queue (3 VALU) and
// the whole wave drains its queues only when one of them is half full: then the IEEE sqrt, the tie key
// and the register insertion run back to back for every queued entry.
//
// Bit-exactness (compiled with -ffp-contract=off; every rounding below is explicit):
//   distance  d = sqrt( fma(dz,dz, fma(dy,dy, dx*dx)) ), dx = c.x - q.x   (gpu.cu.cc:102-107 with
//             nvcc's default fma contraction of `sum += val*val`)
//   order     ascending (d, tb), tb(x) = (x % C_THREADS)*C_VPT + x / C_THREADS  -- the rank of
//             point x in CUB's blocked arrangement, which the stable radix sort preserves among
//             equal keys (gpu.cu.cc:98-123); (C_THREADS, C_VPT) from the N ladder (:181-216).
//   The list is kept as 64-bit keys (bits(d) << 32 | tb): d >= 0, so unsigned order == (d, tb) order.
#include <float.h>
#include <limits.h>

#include <type_traits>

#include "common.h"
#include "wave_ops.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

constexpr int kQueriesPerBlock = 256;
constexpr int kChunk = 1024;   // candidates staged per LDS round: 256 groups of 4 x 12 floats = 12 KiB
constexpr int kQueue = 16;     // per-lane survivor queue depth (drained when any lane holds > 8)

struct KnnLadder {
  int log2ct;  // log2(C_THREADS)
  int ctmask;  // C_THREADS-1
  int cv;      // C_VPT
};

static KnnLadder knn_ladder(int N) {
  int t, v;
  if (N <= 32) { t = 32; v = 1; }
  else if (N <= 64) { t = 64; v = 1; }
  else if (N <= 128) { t = 128; v = 1; }
  else if (N <= 256) { t = 128; v = 2; }
  else if (N <= 512) { t = 128; v = 4; }
  else if (N <= 1024) { t = 256; v = 4; }
  else if (N <= 2048) { t = 256; v = 8; }
  else if (N <= 4096) { t = 512; v = 8; }
  else if (N <= 8192) { t = 1024; v = 8; }
  else { t = 1024; v = (N + 1023) / 1024; }  // superset: upstream stops at 8192
  KnnLadder l;
  l.ctmask = t - 1;
  l.cv = v;
  l.log2ct = 0;
  while ((1 << l.log2ct) < t) ++l.log2ct;
  return l;
}

// LDS image of 4 consecutive candidates c0..c3 (12 floats, three 16-byte reads):
//   [x0 x1 y0 y1] [z0 z1 x2 x3] [y2 y3 z2 z3]   -> pairs feed v_pk_* directly
__device__ __forceinline__ int cand_slot(int c, int comp) {
  const int g = c >> 2, r = c & 3;
  const int off = (r < 2) ? (2 * comp + r) : (4 + 2 * comp + r);  // comp 0:x 1:y 2:z
  return g * 12 + off;
}

__device__ __forceinline__ float knn_bcast(float v, int lane) {  // lane is wave-uniform
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// the set bit of m (!= 0) nearest to position gl (0..63), wave-uniform: candidate groups are visited outwards from the
// queries' own group -- Morton neighbours are mostly spatial neighbours, and a K-list fed nearest-first takes ~K
// insertions where one fed in index order takes ~K ln(n/K)
__device__ __forceinline__ int knn_pick_near(unsigned long long m, int gl) {
  const unsigned long long up = m >> gl, dn = m & ((1ull << gl) - 1);
  const int da = up ? __builtin_ctzll(up) : 128;
  const int db = dn ? gl - (63 - __builtin_clzll(dn)) : 128;
  return da <= db ? gl + da : gl - db;
}

template <int KMAX>
struct KnnState {
  u64 keys[KMAX];
  float bound;  // s > bound  =>  sqrt(s) > current K-th distance
};

// KEEP_MIN: the bound may already be tighter than this list's own K-th entry (knn_split_kernel)
template <int KMAX, bool KEEP_MIN = false>
__device__ __forceinline__ void knn_insert_key(KnnState<KMAX> &st, const u64 key) {
  if (key < st.keys[KMAX - 1]) {
    if constexpr (KMAX <= 16) {
      // the list is sorted and keys are unique: entry i becomes its left neighbour where the key goes in further
      // left, the key where it goes in exactly here.  KMAX - 1 independent compares and two selects per entry --
      // no dependent chain (a bubble pass is one compare + four selects per step, each waiting for the last)
      bool lt[KMAX];
#pragma unroll
      for (int i = 0; i < KMAX - 1; ++i) lt[i] = key < st.keys[i];
      lt[KMAX - 1] = true;
#pragma unroll
      for (int i = KMAX - 1; i > 0; --i) {
        const u64 in = lt[i - 1] ? st.keys[i - 1] : key;
        st.keys[i] = lt[i] ? in : st.keys[i];
      }
      st.keys[0] = lt[0] ? key : st.keys[0];
    } else {
      st.keys[KMAX - 1] = key;
#pragma unroll
      for (int i = KMAX - 1; i > 0; --i) {
        const u64 a = st.keys[i - 1], b = st.keys[i];
        const bool lt = b < a;
        st.keys[i - 1] = lt ? b : a;
        st.keys[i] = lt ? a : b;
      }
    }
    const unsigned hb = (unsigned)(st.keys[KMAX - 1] >> 32);
    if (hb <= 0x7f800000u) {
      const float dk = __uint_as_float(hb);
      // (1+2^-20)-inflated square of the K-th distance: any s above it has sqrt(s) > d_K.
      const float nb = __fmul_rn(__fmul_rn(dk, dk), 1.000001f);
      st.bound = KEEP_MIN ? fminf(st.bound, nb) : nb;
    }
  }
}
template <int KMAX, bool KEEP_MIN = false>
__device__ __forceinline__ void knn_offer(KnnState<KMAX> &st, float s, int x, const KnnLadder &lad) {
  if (s <= st.bound) {
    const float d = sqrtf(s);  // IEEE-rounded (llvm.sqrt.f32 under -fhip-fp32-correctly-rounded-divide-sqrt)
    const unsigned tb = (unsigned)((x & lad.ctmask) * lad.cv + (x >> lad.log2ct));
    knn_insert_key<KMAX, KEEP_MIN>(st, ((u64)__float_as_uint(d) << 32) | tb);
  }
}


// Squared distances (reference rounding order) of 8 consecutive candidates of the pair-SoA LDS image.
__device__ __forceinline__ void knn_dist8(const float *g0, const f32x2 qx2, const f32x2 qy2, const f32x2 qz2,
                                          f32x2 (&s)[4]) {
  const f32x4 a0 = *reinterpret_cast<const f32x4 *>(g0), a1 = *reinterpret_cast<const f32x4 *>(g0 + 4),
              a2 = *reinterpret_cast<const f32x4 *>(g0 + 8), b0 = *reinterpret_cast<const f32x4 *>(g0 + 12),
              b1 = *reinterpret_cast<const f32x4 *>(g0 + 16), b2 = *reinterpret_cast<const f32x4 *>(g0 + 20);
  {
    const f32x2 dx = f32x2{a0[0], a0[1]} - qx2, dy = f32x2{a0[2], a0[3]} - qy2, dz = f32x2{a1[0], a1[1]} - qz2;
    s[0] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
  }
  {
    const f32x2 dx = f32x2{a1[2], a1[3]} - qx2, dy = f32x2{a2[0], a2[1]} - qy2, dz = f32x2{a2[2], a2[3]} - qz2;
    s[1] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
  }
  {
    const f32x2 dx = f32x2{b0[0], b0[1]} - qx2, dy = f32x2{b0[2], b0[3]} - qy2, dz = f32x2{b1[0], b1[1]} - qz2;
    s[2] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
  }
  {
    const f32x2 dx = f32x2{b1[2], b1[3]} - qx2, dy = f32x2{b2[0], b2[1]} - qy2, dz = f32x2{b2[2], b2[3]} - qz2;
    s[3] = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
  }
}
__device__ __forceinline__ float knn_min8(const f32x2 (&s)[4]) {
  return fminf(fminf(fminf(s[0][0], s[0][1]), fminf(s[1][0], s[1][1])),
               fminf(fminf(s[2][0], s[2][1]), fminf(s[3][0], s[3][1])));
}

// XYZ_LAYOUT: false = positions [B,3,N] (op layout), true = xyz [B,N,3].
template <int KMAX, bool XYZ_LAYOUT>
__global__ __launch_bounds__(kQueriesPerBlock) void knn_kernel(const float *__restrict__ pos, int N,
                                                              int K, KnnLadder lad,
                                                              int32_t *__restrict__ nn,
                                                              float *__restrict__ dist) {
  __shared__ __attribute__((aligned(16))) float s_c[kChunk * 3];
  __shared__ uint2 s_q[kQueue * kQueriesPerBlock];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int y = blockIdx.x * kQueriesPerBlock + tid;
  const float *pc = pos + (size_t)b * 3 * N;

  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (y < N) {
    if (XYZ_LAYOUT) { qx = pc[(size_t)y * 3]; qy = pc[(size_t)y * 3 + 1]; qz = pc[(size_t)y * 3 + 2]; }
    else { qx = pc[y]; qy = pc[(size_t)N + y]; qz = pc[(size_t)2 * N + y]; }
  }
  const f32x2 qx2 = {qx, qx}, qy2 = {qy, qy}, qz2 = {qz, qz};

  KnnState<KMAX> st;
#pragma unroll
  for (int i = 0; i < KMAX; ++i) st.keys[i] = ~0ull;
  st.bound = INFINITY;
  int cnt = 0;

  for (int base = 0; base < N; base += kChunk) {
    const int len = min(kChunk, N - base);
    const int len8 = (len + 31) & ~31;  // padded (+inf) to the 32-candidate step of the scan loop
    __syncthreads();
    if (XYZ_LAYOUT) {
      for (int e = tid; e < len8 * 3; e += kQueriesPerBlock) {
        const int c = e / 3, comp = e - c * 3;
        s_c[cand_slot(c, comp)] = c < len ? pc[(size_t)base * 3 + e] : INFINITY;
      }
    } else {
      for (int e = tid; e < len8 * 3; e += kQueriesPerBlock) {
        const int comp = e / len8, c = e - comp * len8;
        s_c[cand_slot(c, comp)] = c < len ? pc[(size_t)comp * N + base + c] : INFINITY;
      }
    }
    __syncthreads();
    // 32 candidates per iteration: 24 broadcast LDS reads + 48 packed ops with nothing between them that
    // depends on the lane's list, so they pipeline; ONE wave-uniform test decides whether anybody needs the
    // slow path (a per-8 branch serialised every step behind its own LDS latency: 650-800 cycles/step
    // measured with one wave per SIMD, tools/knn_probe.py).
    const int len32 = (len8 + 31) & ~31;  // the image is padded with +inf up to a multiple of 32
    for (int j = 0; j < len32; j += 32) {
      f32x2 s[4][4];
      float mn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        knn_dist8(s_c + ((j >> 2) + 2 * u) * 12, qx2, qy2, qz2, s[u]);
        mn[u] = knn_min8(s[u]);
      }
      const float m = fminf(fminf(mn[0], mn[1]), fminf(mn[2], mn[3]));
      if (__any(y < N && m <= st.bound)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (y < N && mn[u] <= st.bound) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const float sv = s[u][t >> 1][t & 1];
              const int x = base + j + 8 * u + t;
              if (sv <= st.bound && x < N) {
                s_q[cnt * kQueriesPerBlock + tid] = make_uint2(__float_as_uint(sv), (unsigned)x);
                ++cnt;
              }
            }
          }
          if (__any(cnt > kQueue - 8)) {  // wave-uniform: drain every lane's queue
            for (int i = 0; i < kQueue; ++i) {
              if (!__any(i < cnt)) break;
              if (i < cnt) {
                const uint2 e = s_q[i * kQueriesPerBlock + tid];
                knn_offer<KMAX>(st, __uint_as_float(e.x), (int)e.y, lad);
              }
            }
            cnt = 0;
          }
        }
      }
    }
  }
  for (int i = 0; i < kQueue; ++i) {  // final drain
    if (i < cnt) {
      const uint2 e = s_q[i * kQueriesPerBlock + tid];
      knn_offer<KMAX>(st, __uint_as_float(e.x), (int)e.y, lad);
    }
  }

  if (y < N) {
    int32_t *o_nn = nn + ((size_t)b * N + y) * K;
    float *o_d = dist + ((size_t)b * N + y) * K;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      if (i < K) {
        const unsigned tb = (unsigned)st.keys[i];
        if (st.keys[i] == ~0ull) {  // fewer than K points: reference pads with id -1 / FLT_MAX (:110-111)
          o_nn[i] = -1;
          o_d[i] = FLT_MAX;
        } else {
          o_nn[i] = (int)(((tb % (unsigned)lad.cv) << lad.log2ct) + tb / (unsigned)lad.cv);
          o_d[i] = __uint_as_float((unsigned)(st.keys[i] >> 32));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// kNN on a spatially ordered cloud (spatial.hip).  Still exhaustive in effect -- every candidate is either
// evaluated or PROVEN unable to enter any list of the wave -- but most of the N^2 work disappears:
// a wave owns one group of 64 Morton-consecutive queries and walks the candidate groups nearest-first
// (g, g+1, g-1, g+2, ...).  A candidate group is skipped when the squared distance between the two
// bounding boxes exceeds every lane's current screening bound (which already over-estimates the K-th
// distance); the 1e-5 relative margin on the box distance dwarfs the rounding of the f32 distance chain,
// so a skipped candidate would also have failed the per-candidate screen.  Evaluated groups go through
// exactly the same screen / queue / 64-bit-key insertion as knn_kernel, so ids and distances are
// bit-identical to it (tests).  Waves are independent: no block-level barrier.
constexpr int kSortedWaves = 4;
constexpr int kCellInts = 4112;  // ints per cloud of the cell table (spatial.hip)
constexpr int kCellFlag = 4106;  // ... [kCellFlag]: 1 = not a cloud for cell lists (dense cells), the pruned scan takes it
#ifdef DH3D_KNN_PROBE  // dev instrumentation (tools/knn_probe.py): per-wave cycle / event counters
__device__ long long g_kprobe[8 * 4096];
#endif

template <int KMAX>
__global__ __launch_bounds__(64 * kSortedWaves) void knn_sorted_kernel(const float4 *__restrict__ sorted,
                                                                    const float *__restrict__ gbox, int N,
                                                                    int K, KnnLadder lad,
                                                                    int32_t *__restrict__ nn,
                                                                    float *__restrict__ dist, const int *__restrict__ gate) {
  if (gate && !gate[(size_t)blockIdx.y * kCellInts + kCellFlag]) return;  // (dh3d_knn_grid: only the clouds the cell lists left)
  __shared__ __attribute__((aligned(16))) float s_c[kSortedWaves][64 * 3];  // pair-SoA image per wave
  __shared__ uint2 s_q[kSortedWaves][kQueue * 64];
  __shared__ int s_id[kSortedWaves][64];  // original ids of the staged candidates
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int NG = (N + 63) / 64;
  const int g = blockIdx.x * kSortedWaves + wave;
  if (g >= NG) return;  // whole wave
  const float4 *sc = sorted + (size_t)b * N;
  const float4 *gb = reinterpret_cast<const float4 *>(gbox) + (size_t)b * NG * 2;  // [lo.xyz,_ | hi.xyz,_]
  float *my_c = s_c[wave];
  uint2 *my_q = s_q[wave];
  int *my_id = s_id[wave];

  const int qi = g * 64 + lane;
  const bool valid = qi < N;
  const float4 pad = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(-1));
  float4 qr = valid ? sc[qi] : pad;
  const f32x2 qx2 = {qr.x, qr.x}, qy2 = {qr.y, qr.y}, qz2 = {qr.z, qr.z};
  const float4 qlo = gb[g * 2], qhi = gb[g * 2 + 1];

  KnnState<KMAX> st;
#pragma unroll
  for (int i = 0; i < KMAX; ++i) st.keys[i] = ~0ull;
  st.bound = INFINITY;
  int cnt = 0;
  float wave_bound = INFINITY;  // max over the wave's valid lanes of st.bound (refreshed after each drain)

#ifdef DH3D_KNN_PROBE
  long long pr_t0 = clock64(), pr_drain = 0;
  int pr_ndrain = 0, pr_nslots = 0, pr_ngroups = 0;
#endif
  auto drain = [&]() {
#ifdef DH3D_KNN_PROBE
    const long long d0 = clock64();
    ++pr_ndrain;
#endif
    // ONE copy of the (long) insertion per drain site, the next slot requested from LDS while this one is inserted.
    // Unrolled over the 16 slots the kernel was > 100 KB of code (the drain is inlined at every site of the scan); the
    // instruction cache coped with that in the shipped build (0.2 % misses, profiles/r03_d_pmc_knn.txt) but not in the
    // probe build, and nothing is gained by the unrolling.
    const int deepest = -wave_min_i32(-cnt);
    uint2 nxt = my_q[lane];
#pragma unroll 1
    for (int i = 0; i < deepest; ++i) {  // wave-uniform trip count
      const uint2 e = nxt;
      if (i + 1 < deepest) nxt = my_q[(i + 1) * 64 + lane];
#ifdef DH3D_KNN_PROBE
      ++pr_nslots;
#endif
      if (i < cnt) knn_offer<KMAX>(st, __uint_as_float(e.x), (int)e.y, lad);
    }
    cnt = 0;
    wave_bound = wave_max_f32(valid ? st.bound : 0.f);
#ifdef DH3D_KNN_PROBE
    pr_drain += clock64() - d0;
#endif
  };

  // evaluate the 64 candidates of group gcc (records already in `cr`, one per lane)
  auto scan_group = [&](int gcc, const float4 cr) {
#ifdef DH3D_KNN_PROBE
    ++pr_ngroups;
#endif
    my_c[cand_slot(lane, 0)] = cr.x;
    my_c[cand_slot(lane, 1)] = cr.y;
    my_c[cand_slot(lane, 2)] = cr.z;
    my_id[lane] = __float_as_int(cr.w);
    __builtin_amdgcn_wave_barrier();
    const int clen = min(64, N - gcc * 64);
    // 32 candidates per iteration, one wave-uniform test (see knn_kernel); padding records are +inf
    for (int j = 0; j < clen; j += 32) {
      f32x2 s[4][4];
      float mn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        knn_dist8(my_c + ((j >> 2) + 2 * u) * 12, qx2, qy2, qz2, s[u]);
        mn[u] = knn_min8(s[u]);
      }
      const float m = fminf(fminf(mn[0], mn[1]), fminf(mn[2], mn[3]));
      if (__any(valid && m <= st.bound)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (valid && mn[u] <= st.bound) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const float sv = s[u][t >> 1][t & 1];
              if (sv <= st.bound && j + 8 * u + t < clen) {
                my_q[cnt * 64 + lane] = make_uint2(__float_as_uint(sv), (unsigned)my_id[j + 8 * u + t]);
                ++cnt;
              }
            }
          }
          if (__any(cnt > kQueue - 8)) drain();
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto load_group = [&](int gcc) { const int ci = gcc * 64 + lane; return ci < N ? sc[ci] : pad; };

  // own group first: it fills every list and gives the first finite bounds
  scan_group(g, qr);
  if (__any(cnt > 0)) drain();

  // other groups: lane <-> candidate group box test (64 groups per round), overlapping boxes (distance 0)
  // first, then whatever still lies within the (tightened) bound; next group's records are prefetched.
  const int cg = g >> 6, NC = (NG + 63) >> 6;
  for (int tier = 0; tier < 2; ++tier) {
    for (int k = 0; k < 2 * NC; ++k) {  // chunks of 64 groups outwards from the own one: cg, cg+1, cg-1, cg+2, ...
      const int ci = cg + ((k & 1) ? (k + 1) / 2 : -(k / 2));
      if (ci < 0 || ci >= NC) continue;
      const int c0 = ci * 64, gl = min(max(g - c0, 0), 63);
      const int gi = c0 + lane;
      const bool other = gi < NG && gi != g;
      float bd = INFINITY;
      float4 clo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), chi = clo;
      if (other) {
        clo = gb[gi * 2];
        chi = gb[gi * 2 + 1];
        const float ex = fmaxf(fmaxf(clo.x - qhi.x, qlo.x - chi.x), 0.f);
        const float ey = fmaxf(fmaxf(clo.y - qhi.y, qlo.y - chi.y), 0.f);
        const float ez = fmaxf(fmaxf(clo.z - qhi.z, qlo.z - chi.z), 0.f);
        bd = (ex * ex + ey * ey + ez * ez) * 0.99999f;
      }
      unsigned long long mask =
          tier == 0 ? __ballot(other && bd == 0.f) : __ballot(other && bd > 0.f && bd <= wave_bound);
      if (!mask) continue;
      int l = knn_pick_near(mask, gl);
      mask &= ~(1ull << l);
      float4 nxt = load_group(c0 + l);
      while (true) {
        const int lcur = l;
        const float4 cr = nxt;
        const bool more = mask != 0;
        if (more) {
          l = knn_pick_near(mask, gl);
          mask &= ~(1ull << l);
          nxt = load_group(c0 + l);
        }
        const float bdl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd), lcur));
        if (bdl <= wave_bound) {  // the bound may have tightened since the ballot
          // box against box says "some query of the wave MIGHT reach the group" with the union of the 64 queries and
          // the loosest of their bounds; the same test per query (point to box, own bound) is ~20 instructions
          // against the ~450 of scanning the group, and decides most of the overlapping (tier 0) groups of a query
          // group whose own box is large (Morton order jumps across the cloud inside it)
          const float lx = knn_bcast(clo.x, lcur), ly = knn_bcast(clo.y, lcur), lz = knn_bcast(clo.z, lcur);
          const float hx = knn_bcast(chi.x, lcur), hy = knn_bcast(chi.y, lcur), hz = knn_bcast(chi.z, lcur);
          const float px = fmaxf(fmaxf(lx - qr.x, qr.x - hx), 0.f);
          const float py = fmaxf(fmaxf(ly - qr.y, qr.y - hy), 0.f);
          const float pz = fmaxf(fmaxf(lz - qr.z, qr.z - hz), 0.f);
          const float pd = (px * px + py * py + pz * pz) * 0.99999f;
          if (__any(valid && pd <= st.bound)) scan_group(c0 + lcur, cr);
        }
        if (!more) break;
      }
    }
  }
  if (__any(cnt > 0)) drain();
#ifdef DH3D_KNN_PROBE
  if (lane == 0 && b == 0) {
    long long *o = g_kprobe + (size_t)g * 8;
    o[0] = clock64() - pr_t0; o[1] = pr_drain; o[2] = pr_ndrain; o[3] = pr_nslots; o[4] = pr_ngroups;
  }
#endif

  if (valid) {
    const int y = __float_as_int(qr.w);  // original index of this query
    int32_t *o_nn = nn + ((size_t)b * N + y) * K;
    float *o_d = dist + ((size_t)b * N + y) * K;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      if (i < K) {
        const unsigned tb = (unsigned)st.keys[i];
        if (st.keys[i] == ~0ull) {
          o_nn[i] = -1;
          o_d[i] = FLT_MAX;
        } else {
          o_nn[i] = (int)(((tb % (unsigned)lad.cv) << lad.log2ct) + tb / (unsigned)lad.cv);
          o_d[i] = __uint_as_float((unsigned)(st.keys[i] >> 32));
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same search with S waves per query group.  knn_sorted_kernel has one wave per 64 queries: at the model's sizes
// that is ONE wave per SIMD on the whole chip, each a dependent instruction stream (a wave issues ~1 instruction per
// 4.6 cycles), and the kernel ends with its slowest group.  Here the S waves of a workgroup hold the same 64 queries and
// split the CANDIDATE groups (group gi belongs to wave gi % S), each with its own K-lists; the lists are merged through
// LDS at the end.  A wave alone would prune with the K-th entry of a list that saw 1/S of the candidates; instead every
// wave publishes, per query, the distance of its ceil(K/S)-th entry: S lists hold at least K candidates within the
// largest of those S distances, so max_w d_w bounds the true K-th distance -- as does any single wave's own K-th
// distance; the screen uses the smallest of all of these.  Published values are only ever replaced by smaller ones, so a stale read is still valid
// (no barrier).  Every candidate is evaluated by exactly one wave with the same arithmetic, keys are unique: the merged
// list is the list of the one-wave kernel, bit for bit.
// The body is a device function over a raw LDS block (knn_split_lds_bytes<S>()): knn_split_kernel gives it a block of its
// own, knn_grid_kernel ALIASES it with its candidate lists and serves a crowded cloud in the same launch (round 6: the
// pruned scan as a second launch behind the cell lists was a 4.8 us no-op on every uniform cloud, and vice versa).
template <int S>
__host__ __device__ constexpr int knn_split_lds_bytes() {
  return S * (64 * 3 * 4 + kQueue * 64 * 8 + 64 * 4 + 64 * 4 + 64 * 4);
}
template <int KMAX, int S>
__device__ __forceinline__ void knn_split_body(const float4 *__restrict__ sorted, const float *__restrict__ gbox, int N, int K,
                                               const KnnLadder lad, int32_t *__restrict__ nn, float *__restrict__ dist,
                                               const int b, const int g, unsigned char *lds, const int wave0 = 0) {
  static_assert(KMAX * 64 * sizeof(u64) <= kQueue * 64 * sizeof(uint2), "a K-list fits its wave's survivor queue");
  uint2 (*s_q)[kQueue * 64] = reinterpret_cast<uint2 (*)[kQueue * 64]>(lds);                     // [S] survivor queues
  float (*s_c)[64 * 3] = reinterpret_cast<float (*)[64 * 3]>(lds + S * kQueue * 64 * 8);         // [S] pair-SoA image per wave
  int (*s_id)[64] = reinterpret_cast<int (*)[64]>(lds + S * (kQueue * 64 * 8 + 64 * 3 * 4));     // [S]
  float (*s_share)[64] = reinterpret_cast<float (*)[64]>(lds + S * (kQueue * 64 * 8 + 64 * 3 * 4 + 64 * 4));  // per wave, per query:
                                                                 // inflated square of the distance of its R-th entry
  float (*s_kth)[64] = reinterpret_cast<float (*)[64]>(lds + S * (kQueue * 64 * 8 + 64 * 3 * 4 + 64 * 8));    // ... and of its K-th
                                                                 // entry (its own screening bound)
  constexpr int R = (KMAX + S - 1) / S - 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) - wave0);   // (wave0: the first of this group's S waves)
  const int NG = (N + 63) / 64;
  const float4 *sc = sorted + (size_t)b * N;
  const float4 *gb = reinterpret_cast<const float4 *>(gbox) + (size_t)b * NG * 2;  // [lo.xyz,_ | hi.xyz,_]
  float *my_c = s_c[wave];
  uint2 *my_q = s_q[wave];
  int *my_id = s_id[wave];

  const int qi = g * 64 + lane;
  const bool valid = qi < N;
  const float4 pad = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(-1));
  const float4 qr = valid ? sc[qi] : pad;
  const f32x2 qx2 = {qr.x, qr.x}, qy2 = {qr.y, qr.y}, qz2 = {qr.z, qr.z};
  const float4 qlo = gb[g * 2], qhi = gb[g * 2 + 1];

  KnnState<KMAX> st;
#pragma unroll
  for (int i = 0; i < KMAX; ++i) st.keys[i] = ~0ull;
  st.bound = INFINITY;
  int cnt = 0;
  float wave_bound = INFINITY;
  s_share[wave][lane] = INFINITY;
  s_kth[wave][lane] = INFINITY;
  __syncthreads();
#ifdef DH3D_KNN_PROBE
  long long pr_t0 = clock64(), pr_drain = 0, pr_scan = 0;
  int pr_ndrain = 0, pr_nslots = 0, pr_ngroups = 0, pr_hits = 0, pr_sparse = 0, pr_sparse_slots = 0, pr_entries = 0, pr_halfskip = 0, pr_steps = 0;
#endif

  // everyone's progress -> my screen: the true K-th distance is at most any wave's own K-th, and at most the largest
  // of the S R-th distances
  auto adopt = [&](float mx) {
    float mk = st.bound;
#pragma unroll
    for (int w = 1; w < S; ++w) {
      mx = fmaxf(mx, s_share[(wave + w) % S][lane]);
      mk = fminf(mk, s_kth[(wave + w) % S][lane]);
    }
    st.bound = fminf(mk, mx);
    wave_bound = wave_max_f32(valid ? st.bound : 0.f);
  };

  auto drain = [&]() {
#ifdef DH3D_KNN_PROBE
    const long long d0 = clock64();
    ++pr_ndrain;
#endif
    const int deepest = -wave_min_i32(-cnt);
#ifdef DH3D_KNN_PROBE
    {
      const int deep = __popcll(__ballot(cnt > 4));
      int tot = cnt;
      for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
      pr_entries += tot;
      if (deep <= 3) { ++pr_sparse; pr_sparse_slots += deepest; }
    }
#endif
    uint2 nxt = my_q[lane];
#pragma unroll 1
    for (int i = 0; i < deepest; ++i) {  // one copy of the insertion per site (code size, see knn_sorted_kernel)
      const uint2 e = nxt;
      if (i + 1 < deepest) nxt = my_q[(i + 1) * 64 + lane];
#ifdef DH3D_KNN_PROBE
      ++pr_nslots;
#endif
      if (i < cnt) knn_offer<KMAX, true>(st, __uint_as_float(e.x), (int)e.y, lad);
    }
    cnt = 0;
    // publish my R-th distance, take the largest of everyone's
    const unsigned hb = (unsigned)(st.keys[R] >> 32);
    float mx = INFINITY;
    if (hb <= 0x7f800000u) {
      const float dr = __uint_as_float(hb);
      mx = __fmul_rn(__fmul_rn(dr, dr), 1.000001f);
    }
    s_share[wave][lane] = mx;
    const unsigned hk = (unsigned)(st.keys[KMAX - 1] >> 32);
    if (hk <= 0x7f800000u) {
      const float dk = __uint_as_float(hk);
      s_kth[wave][lane] = __fmul_rn(__fmul_rn(dk, dk), 1.000001f);
    }
    adopt(mx);
#ifdef DH3D_KNN_PROBE
    pr_drain += clock64() - d0;
#endif
  };

  auto scan_group = [&](int gcc, const float4 cr) {
#ifdef DH3D_KNN_PROBE
    ++pr_ngroups;
    const long long s0 = clock64(), dr0 = pr_drain;
#endif
    my_c[cand_slot(lane, 0)] = cr.x;
    my_c[cand_slot(lane, 1)] = cr.y;
    my_c[cand_slot(lane, 2)] = cr.z;
    my_id[lane] = __float_as_int(cr.w);
    __builtin_amdgcn_wave_barrier();
    const int clen = min(64, N - gcc * 64);
    for (int j = 0; j < clen; j += 32) {
#ifdef DH3D_KNN_PROBE  // would a per-query test against the box of these 32 candidates have skipped the step?
      {
        const bool mine = (lane >> 5) == (j >> 5);   // the half-group's candidates sit in lanes 32*(j/32)..
        float lo[3] = {mine ? cr.x : INFINITY, mine ? cr.y : INFINITY, mine ? cr.z : INFINITY};
        float hi[3] = {mine ? cr.x : -INFINITY, mine ? cr.y : -INFINITY, mine ? cr.z : -INFINITY};
        if (cr.x == INFINITY) { hi[0] = hi[1] = hi[2] = -INFINITY; }  // padding records
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3)
          for (int off = 32; off > 0; off >>= 1) {
            lo[c3] = fminf(lo[c3], __shfl_xor(lo[c3], off, 64));
            hi[c3] = fmaxf(hi[c3], __shfl_xor(hi[c3], off, 64));
          }
        const float px = fmaxf(fmaxf(lo[0] - qr.x, qr.x - hi[0]), 0.f), py = fmaxf(fmaxf(lo[1] - qr.y, qr.y - hi[1]), 0.f),
                    pz = fmaxf(fmaxf(lo[2] - qr.z, qr.z - hi[2]), 0.f);
        if (!__any(valid && (px * px + py * py + pz * pz) * 0.99999f <= st.bound)) ++pr_halfskip;
        ++pr_steps;
      }
#endif
      f32x2 sq[4][4];
      float mn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        knn_dist8(my_c + ((j >> 2) + 2 * u) * 12, qx2, qy2, qz2, sq[u]);
        mn[u] = knn_min8(sq[u]);
      }
      const float m = fminf(fminf(mn[0], mn[1]), fminf(mn[2], mn[3]));
      if (__any(valid && m <= st.bound)) {
#ifdef DH3D_KNN_PROBE
        ++pr_hits;
#endif
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (valid && mn[u] <= st.bound) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
              const float sv = sq[u][t >> 1][t & 1];
              if (sv <= st.bound && j + 8 * u + t < clen) {
                my_q[cnt * 64 + lane] = make_uint2(__float_as_uint(sv), (unsigned)my_id[j + 8 * u + t]);
                ++cnt;
              }
            }
          }
          if (__any(cnt > kQueue - 8)) drain();
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#ifdef DH3D_KNN_PROBE
    pr_scan += (clock64() - s0) - (pr_drain - dr0);
#endif
  };
  auto load_group = [&](int gcc) { const int ci = gcc * 64 + lane; return ci < N ? sc[ci] : pad; };

  // lane <-> a candidate group this wave OWNS: the j-th is group j * S + wave.  Box against box in the lanes, 64 owned
  // groups per round; their boxes are requested before the own group is scanned and stay in registers for both tiers.
  const int NO = (NG - wave + S - 1) / S, NCH = (NO + 63) >> 6;  // owned groups, chunks of 64 of them
  const int jg = g / S, cj = min(jg >> 6, max(NCH - 1, 0));      // where the own group sits among them
  bool other = false;
  float bd = INFINITY;
  float4 clo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), chi = clo;
  auto fetch_boxes = [&](int ci) {
    const int j = ci * 64 + lane, gi = j * S + wave;
    other = j < NO && gi != g;
    if (other) {
      clo = gb[gi * 2];
      chi = gb[gi * 2 + 1];
    }
  };
  auto test_boxes = [&]() {
    bd = INFINITY;
    if (other) {
      const float ex = fmaxf(fmaxf(clo.x - qhi.x, qlo.x - chi.x), 0.f);
      const float ey = fmaxf(fmaxf(clo.y - qhi.y, qlo.y - chi.y), 0.f);
      const float ez = fmaxf(fmaxf(clo.z - qhi.z, qlo.z - chi.z), 0.f);
      bd = (ex * ex + ey * ey + ez * ez) * 0.99999f;
    }
  };
  if (NCH == 1) fetch_boxes(0);

  {
    // The queries' own group goes first, its blocks of 8 candidates dealt round-robin to the S waves: every wave enters
    // the other groups with a list of its own and -- after the one barrier -- a bound all 64 candidates contributed
    // to.  (With the group scanned by its owner alone the other waves met their first group with no bound at all and
    // queued every candidate of it.)
    my_c[cand_slot(lane, 0)] = qr.x;
    my_c[cand_slot(lane, 1)] = qr.y;
    my_c[cand_slot(lane, 2)] = qr.z;
    my_id[lane] = __float_as_int(qr.w);
    __builtin_amdgcn_wave_barrier();
    const int clen = min(64, N - g * 64);
    for (int blk = wave; blk * 8 < clen; blk += S) {
      f32x2 sq[4];
      knn_dist8(my_c + 2 * blk * 12, qx2, qy2, qz2, sq);
      if (valid) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float sv = sq[t >> 1][t & 1];
          if (sv <= st.bound && blk * 8 + t < clen) {
            my_q[cnt * 64 + lane] = make_uint2(__float_as_uint(sv), (unsigned)my_id[blk * 8 + t]);
            ++cnt;
          }
        }
      }
      if (__any(cnt > 0)) drain();
    }
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    adopt(s_share[wave][lane]);
  }
  for (int tier = 0; tier < 2; ++tier) {
    for (int k = 0; k < 2 * NCH; ++k) {  // chunks of 64 owned groups outwards from the own one: cj, cj+1, cj-1, ...
      const int ci = cj + ((k & 1) ? (k + 1) / 2 : -(k / 2));
      if (ci < 0 || ci >= NCH) continue;
      if (NCH > 1) fetch_boxes(ci);  // one chunk (N <= 64*64*S points): fetched once, before the own group
      if (NCH > 1 || tier == 0) test_boxes();
      const int gl = min(max(jg - ci * 64, 0), 63);
      if (tier == 1) adopt(s_share[wave][lane]);  // the others have worked since the last look
      unsigned long long mask =
          tier == 0 ? __ballot(other && bd == 0.f) : __ballot(other && bd > 0.f && bd <= wave_bound);
      if (!mask) continue;
      int l = knn_pick_near(mask, gl);
      mask &= ~(1ull << l);
      float4 nxt = load_group((ci * 64 + l) * S + wave);
      while (true) {
        const int lcur = l;
        const float4 cr = nxt;
        const bool more = mask != 0;
        if (more) {
          l = knn_pick_near(mask, gl);
          mask &= ~(1ull << l);
          nxt = load_group((ci * 64 + l) * S + wave);
        }
        const float bdl = knn_bcast(bd, lcur);
        if (bdl <= wave_bound) {  // the bound may have tightened since the ballot
          // box against box says "some query of the wave MIGHT reach the group" with the union of the 64 queries and
          // the loosest of their bounds; the same test per query (point to box, own bound) is ~20 instructions
          // against the ~450 of scanning the group, and decides most of the overlapping (tier 0) groups of a query
          // group whose own box is large (Morton order jumps across the cloud inside it)
          const float lx = knn_bcast(clo.x, lcur), ly = knn_bcast(clo.y, lcur), lz = knn_bcast(clo.z, lcur);
          const float hx = knn_bcast(chi.x, lcur), hy = knn_bcast(chi.y, lcur), hz = knn_bcast(chi.z, lcur);
          const float px = fmaxf(fmaxf(lx - qr.x, qr.x - hx), 0.f);
          const float py = fmaxf(fmaxf(ly - qr.y, qr.y - hy), 0.f);
          const float pz = fmaxf(fmaxf(lz - qr.z, qr.z - hz), 0.f);
          const float pd = (px * px + py * py + pz * pz) * 0.99999f;
          if (__any(valid && pd <= st.bound)) scan_group((ci * 64 + lcur) * S + wave, cr);
        }
        if (!more) break;
      }
    }
  }
  if (__any(cnt > 0)) drain();
#ifdef DH3D_KNN_PROBE
  if (lane == 0 && ((b * NG + g) * S + wave) < 4096) {
    long long *o = g_kprobe + (size_t)((b * NG + g) * S + wave) * 8;
    o[7] = pr_halfskip * 1000 + pr_steps;
    o[0] = clock64() - pr_t0; o[1] = pr_drain; o[2] = pr_ndrain; o[3] = pr_nslots; o[4] = pr_ngroups;
    o[5] = pr_scan; o[6] = pr_hits + 1000 * pr_sparse + 1000000ll * pr_sparse_slots + 1000000000ll * pr_entries;
  }
#endif

  // merge the S lists (tree: w <- w + step), keys are unique so plain insertion
  for (int step = S / 2; step >= 1; step >>= 1) {
    __syncthreads();  // the queues of the waves about to publish are drained
    if (wave >= step && wave < 2 * step) {
#pragma unroll
      for (int i = 0; i < KMAX; ++i)
        my_q[i * 64 + lane] = make_uint2((unsigned)st.keys[i], (unsigned)(st.keys[i] >> 32));
    }
    __syncthreads();
    if (wave < step) {
      const uint2 *oq = s_q[wave + step];
#pragma unroll
      for (int i = 0; i < KMAX; ++i) {
        const uint2 e = oq[i * 64 + lane];
        const u64 key = ((u64)e.y << 32) | e.x;
        if (key < st.keys[KMAX - 1]) {
          st.keys[KMAX - 1] = key;
#pragma unroll
          for (int t = KMAX - 1; t > 0; --t) {
            const u64 a = st.keys[t - 1], c = st.keys[t];
            const bool lt = c < a;
            st.keys[t - 1] = lt ? c : a;
            st.keys[t] = lt ? a : c;
          }
        }
      }
    }
  }

  if (wave == 0 && valid) {
    const int y = __float_as_int(qr.w);  // original index of this query
    int32_t *o_nn = nn + ((size_t)b * N + y) * K;
    float *o_d = dist + ((size_t)b * N + y) * K;
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      if (i < K) {
        const unsigned tb = (unsigned)st.keys[i];
        if (st.keys[i] == ~0ull) {
          o_nn[i] = -1;
          o_d[i] = FLT_MAX;
        } else {
          o_nn[i] = (int)(((tb % (unsigned)lad.cv) << lad.log2ct) + tb / (unsigned)lad.cv);
          o_d[i] = __uint_as_float((unsigned)(st.keys[i] >> 32));
        }
      }
    }
  }
}


template <int KMAX, int S>
__global__ __launch_bounds__(64 * S) void knn_split_kernel(const float4 *__restrict__ sorted,
                                                         const float *__restrict__ gbox, int N, int K,
                                                         KnnLadder lad, int32_t *__restrict__ nn,
                                                         float *__restrict__ dist, const int *__restrict__ gate) {
  if (gate && !gate[(size_t)blockIdx.y * kCellInts + kCellFlag]) return;  // (only the clouds the cell lists left)
  __shared__ __attribute__((aligned(16))) unsigned char s_lds[knn_split_lds_bytes<S>()];
  // (the group is rotated by the cloud: see knn_grid_kernel)
  const unsigned ng = (unsigned)((N + 63) / 64);
  knn_split_body<KMAX, S>(sorted, gbox, N, K, lad, nn, dist, blockIdx.y, (int)((blockIdx.x + 37u * blockIdx.y) % ng), s_lds);
}

// ------------------------------------------------------------------------------------------------
// Small clouds (N <= 2048: the N/8 sampled sets of the model).  The lane-per-query kernels above leave most of the
// chip idle there (B*N/64 waves, each a long dependent scan).  Here a WAVE owns a query: every lane holds
// CPL = N/64 candidates in registers (loaded once, reused for 4 queries), computes their exact keys
// (bits(sqrt d) << 32 | tb), and the K nearest come out of K rounds of "smallest key >= last + 1" -- a per-lane
// scan plus a two-step unsigned wave minimum on the DPP crossbar.  Keys are unique (tb is), so the strict
// threshold replaces any bookkeeping of what was taken.  Same keys, same order, same outputs as knn_kernel.
template <int CPL, bool XYZ>
__global__ __launch_bounds__(256) void knn_small_kernel(const float *__restrict__ pos, int N, int K, KnnLadder lad,
                                                       int32_t *__restrict__ nn, float *__restrict__ dist, int Q) {
  // Q: queries per wave (the candidates are loaded once per wave): 4, or 2 when that still leaves few waves per SIMD --
  // a query is K dependent rounds of two DPP reductions, the kernel is as long as one wave's queries
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q0 = (blockIdx.x * 4 + wave) * Q;
  if (q0 >= N) return;
  const float *base = pos + (size_t)b * N * 3;
  float cx[CPL], cy[CPL], cz[CPL];
  unsigned tbk[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int j = lane + 64 * c, jj = j < N ? j : 0;
    cx[c] = XYZ ? base[(size_t)jj * 3] : base[jj];
    cy[c] = XYZ ? base[(size_t)jj * 3 + 1] : base[(size_t)N + jj];
    cz[c] = XYZ ? base[(size_t)jj * 3 + 2] : base[(size_t)2 * N + jj];
    tbk[c] = j < N ? (unsigned)((j & lad.ctmask) * lad.cv + (j >> lad.log2ct)) : 0xFFFFFFFFu;
  }
#pragma unroll 1
  for (int qi = 0; qi < Q; ++qi) {
    const int q = q0 + qi;
    if (q >= N) break;
    const float qx = XYZ ? base[(size_t)q * 3] : base[q];
    const float qy = XYZ ? base[(size_t)q * 3 + 1] : base[(size_t)N + q];
    const float qz = XYZ ? base[(size_t)q * 3 + 2] : base[(size_t)2 * N + q];
    unsigned kd[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      // same roundings as knn_offer / the reference: sqrt(fma(dz,dz,fma(dy,dy,dx*dx))), :102-107
      const float dx = qx - cx[c], dy = qy - cy[c], dz = qz - cz[c];
      const float d = sqrtf(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)));
      kd[c] = tbk[c] != 0xFFFFFFFFu ? __float_as_uint(d) : 0xFFFFFFFFu;
    }
    // Every lane caches its two smallest admissible keys (m1 < m2): a round is then two wave minima and a pop in the
    // winning lane; the per-lane scan over all CPL keys -- most of a round before -- is redone only when some lane has
    // used up both (it won three of the rounds so far: rare, the K nearest spread over the 64 lanes).
    u64 need = 0;  // smallest key still admissible
    unsigned my_hi = 0xFFFFFFFFu, my_lo = 0xFFFFFFFFu;  // lane r keeps result r
    u64 m1 = ~0ull, m2 = ~0ull;
    bool more = false;  // this lane may hold admissible keys beyond m2
    auto rescan = [&]() {
      m1 = ~0ull; m2 = ~0ull;
      int cnt = 0;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const u64 key = ((u64)kd[c] << 32) | tbk[c];
        const bool ok = key >= need && key != ~0ull;
        const bool c1 = ok && key < m1, c2 = ok && key < m2;
        m2 = c1 ? m1 : (c2 ? key : m2);
        m1 = c1 ? key : m1;
        cnt += ok ? 1 : 0;
      }
      more = cnt > 2;
    };
    rescan();
    for (int r = 0; r < K; ++r) {
      const unsigned hi = wave_min_u32((unsigned)(m1 >> 32));
      const unsigned lo = wave_min_u32((unsigned)(m1 >> 32) == hi ? (unsigned)m1 : 0xFFFFFFFFu);
      if (lane == r) { my_hi = hi; my_lo = lo; }
      const u64 sel = ((u64)hi << 32) | lo;
      if (sel == ~0ull) break;  // fewer than K points: the remaining slots keep the pad value
      need = sel + 1;
      const bool won = m1 == sel;
      const bool dry = won && m2 == ~0ull && more;  // both cached keys used, more behind them
      m1 = won ? m2 : m1;
      m2 = won ? ~0ull : m2;
      if (__ballot(dry) != 0ull) rescan();  // wave-uniform
    }
    if (lane < K) {
      const size_t o = ((size_t)b * N + q) * K + lane;
      if (my_hi == 0xFFFFFFFFu && my_lo == 0xFFFFFFFFu) {  // reference pads with id -1 / FLT_MAX (:110-111)
        nn[o] = -1;
        dist[o] = FLT_MAX;
      } else {
        nn[o] = (int)(((my_lo % (unsigned)lad.cv) << lad.log2ct) + my_lo / (unsigned)lad.cv);
        dist[o] = __uint_as_float(my_hi);
      }
    }
  }
}

template <bool XYZ>
int knn_launch(const float *pos, int B, int N, int K, int32_t *nn, float *dist, hipStream_t s) {
  const KnnLadder lad = knn_ladder(N);
  if (N <= 2048) {  // wave-per-query kernel
    const int Q = (long long)B * N <= 16384 ? 2 : 4;  // <= 4 waves per SIMD with two queries per wave
    dim3 sgrid(dh3d_cdiv(N, 4 * Q), B), sblock(256);
    if (N <= 512) hipLaunchKernelGGL((knn_small_kernel<8, XYZ>), sgrid, sblock, 0, s, pos, N, K, lad, nn, dist, Q);
    else if (N <= 1024) hipLaunchKernelGGL((knn_small_kernel<16, XYZ>), sgrid, sblock, 0, s, pos, N, K, lad, nn, dist, Q);
    else hipLaunchKernelGGL((knn_small_kernel<32, XYZ>), sgrid, sblock, 0, s, pos, N, K, lad, nn, dist, Q);
    return dh3d_launch_status();
  }
  dim3 grid(dh3d_cdiv(N, kQueriesPerBlock), B), block(kQueriesPerBlock);
  if (K <= 4) hipLaunchKernelGGL((knn_kernel<4, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  else if (K <= 8) hipLaunchKernelGGL((knn_kernel<8, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  else if (K <= 16) hipLaunchKernelGGL((knn_kernel<16, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  else if (K <= 32) hipLaunchKernelGGL((knn_kernel<32, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  else hipLaunchKernelGGL((knn_kernel<64, XYZ>), grid, block, 0, s, pos, N, K, lad, nn, dist);
  return dh3d_launch_status();
}

// Any position dimension (the reference loops over Dp, knn_bruteforce_kernel_gpu.cu.cc:98-107; DH3D itself only ever
// passes xyz).  The coverage path: lane = query, candidates staged 256 at a time through LDS ([Dp][256], Dp <= 16), the
// same roundings -- sum = fma(val, val, sum) in dimension order from 0, IEEE sqrt -- and the same (distance, CUB rank)
// order, kept as a sorted list of 64-bit keys per lane (insertion; in scratch for large K).  Unfilled slots (N < K):
// id -1, distance FLT_MAX (:110-111).
template <int KMAX>
__global__ __launch_bounds__(256) void knn_anydp_kernel(const float *__restrict__ pos, int Dp, int N, int K, KnnLadder lad,
                                                        int32_t *__restrict__ nn, float *__restrict__ dist) {
  __shared__ float s_c[16][256];
  const int b = blockIdx.y, y = blockIdx.x * 256 + threadIdx.x;
  const float *pc = pos + (size_t)b * Dp * N;
  float q[16];
#pragma unroll
  for (int dp = 0; dp < 16; ++dp) q[dp] = (dp < Dp && y < N) ? pc[(size_t)dp * N + y] : 0.f;
  u64 keys[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) keys[i] = ~0ull;
  for (int x0 = 0; x0 < N; x0 += 256) {
    __syncthreads();
    for (int dp = 0; dp < Dp; ++dp) s_c[dp][threadIdx.x] = x0 + (int)threadIdx.x < N ? pc[(size_t)dp * N + x0 + threadIdx.x] : 0.f;
    __syncthreads();
    const int cnt = min(256, N - x0);
    for (int j = 0; j < cnt; ++j) {
      float sum = 0.f;
      for (int dp = 0; dp < Dp; ++dp) {
        const float val = s_c[dp][j] - q[dp];
        sum = __builtin_fmaf(val, val, sum);
      }
      const float d = sqrtf(sum);
      const int x = x0 + j;
      const unsigned tb = (unsigned)((x & lad.ctmask) * lad.cv + (x >> lad.log2ct));
      const u64 key = ((u64)__float_as_uint(d) << 32) | tb;
      if (key < keys[K - 1]) {
        int i = K - 1;
        while (i > 0 && keys[i - 1] > key) { keys[i] = keys[i - 1]; --i; }
        keys[i] = key;
      }
    }
  }
  if (y >= N) return;
  int32_t *o_nn = nn + ((size_t)b * N + y) * K;
  float *o_d = dist + ((size_t)b * N + y) * K;
  for (int i = 0; i < K; ++i) {
    if (keys[i] == ~0ull) { o_nn[i] = -1; o_d[i] = 3.402823466e+38f; continue; }
    const unsigned tb = (unsigned)keys[i];
    o_nn[i] = (int)(((tb % (unsigned)lad.cv) << lad.log2ct) + tb / (unsigned)lad.cv);
    o_d[i] = __uint_as_float((unsigned)(keys[i] >> 32));
  }
}

}  // namespace

DH3D_API int dh3d_knn_bruteforce(const float *positions, int B, int Dp, int N, int K, int32_t *nn,
                                 float *dist, void *stream) {
  DH3D_REQUIRE(positions && nn && dist && B > 0 && N > 0 && K > 0 && Dp > 0);
  DH3D_SUPPORTED(K <= 64 && B <= 65535);
  if (Dp == 3) return knn_launch<false>(positions, B, N, K, nn, dist, (hipStream_t)stream);
  DH3D_SUPPORTED(Dp <= 16);  // (any Dp upstream; the staging tile here holds 16 coordinates)
  const KnnLadder lad = knn_ladder(N);
  dim3 grid(dh3d_cdiv(N, 256), B), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (K <= 8) hipLaunchKernelGGL(knn_anydp_kernel<8>, grid, block, 0, s, positions, Dp, N, K, lad, nn, dist);
  else if (K <= 16) hipLaunchKernelGGL(knn_anydp_kernel<16>, grid, block, 0, s, positions, Dp, N, K, lad, nn, dist);
  else hipLaunchKernelGGL(knn_anydp_kernel<64>, grid, block, 0, s, positions, Dp, N, K, lad, nn, dist);
  return dh3d_launch_status();
}

DH3D_API int dh3d_knn_bruteforce_xyz(const float *xyz, int B, int N, int K, int32_t *nn, float *dist,
                                     void *stream) {
  DH3D_REQUIRE(xyz && nn && dist && B > 0 && N > 0 && K > 0);
  DH3D_SUPPORTED(K <= 64 && B <= 65535);
  return knn_launch<true>(xyz, B, N, K, nn, dist, (hipStream_t)stream);
}

#ifdef DH3D_DEV  // dev builds only (tools/geo_bench.py): waves per query group of the ordered search; -1 = default
static int g_knn_split = -1;
DH3D_API void dh3d_dev_set_knn_split(int s) { g_knn_split = s; }
#else
static constexpr int g_knn_split = -1;
#endif

static int knn_sorted_launch(const float *sorted, const float *gbox, int B, int N, int K, int32_t *nn, float *dist,
                             const int *gate, void *stream) {
  DH3D_REQUIRE(sorted && gbox && nn && dist && B > 0 && N > 0 && K > 0);
  DH3D_SUPPORTED(K <= 64 && B <= 65535);
  const KnnLadder lad = knn_ladder(N);
  const int NG = (N + 63) / 64;
  dim3 grid(dh3d_cdiv(NG, kSortedWaves), B), block(64 * kSortedWaves);
  hipStream_t s = (hipStream_t)stream;
  const float4 *so = reinterpret_cast<const float4 *>(sorted);
  // waves per query group (0 = the one-wave kernel): as many as keep the chip at <= ~4 waves per SIMD, where the
  // search turns from latency- to issue-bound (MI355X: 8x8192 0.194 -> 0.109 ms at 4; 32x4096 0.178 -> 0.147 at 2)
  const long long groups = (long long)NG * B;
  const int S = g_knn_split >= 0 ? g_knn_split : groups <= 256 ? 8 : groups <= 1280 ? 4 : groups <= 4096 ? 2 : 0;
  if (S > 0 && K <= 16) {
    dim3 sgrid(NG, B);
#define DH3D_SPLIT_CASE(KM, SS)                                                                               \
  if (K <= KM && S == SS) {                                                                                  \
    hipLaunchKernelGGL((knn_split_kernel<KM, SS>), sgrid, dim3(64 * SS), 0, s, so, gbox, N, K, lad, nn, dist, gate); \
    return dh3d_launch_status();                                                                             \
  }
    DH3D_SPLIT_CASE(4, 2) DH3D_SPLIT_CASE(4, 4)
    DH3D_SPLIT_CASE(8, 2) DH3D_SPLIT_CASE(8, 4) DH3D_SPLIT_CASE(8, 8)
    DH3D_SPLIT_CASE(16, 2) DH3D_SPLIT_CASE(16, 4) DH3D_SPLIT_CASE(16, 8)
#undef DH3D_SPLIT_CASE
  }
  if (K <= 4) hipLaunchKernelGGL((knn_sorted_kernel<4>), grid, block, 0, s, so, gbox, N, K, lad, nn, dist, gate);
  else if (K <= 8) hipLaunchKernelGGL((knn_sorted_kernel<8>), grid, block, 0, s, so, gbox, N, K, lad, nn, dist, gate);
  else if (K <= 16) hipLaunchKernelGGL((knn_sorted_kernel<16>), grid, block, 0, s, so, gbox, N, K, lad, nn, dist, gate);
  else if (K <= 32) hipLaunchKernelGGL((knn_sorted_kernel<32>), grid, block, 0, s, so, gbox, N, K, lad, nn, dist, gate);
  else hipLaunchKernelGGL((knn_sorted_kernel<64>), grid, block, 0, s, so, gbox, N, K, lad, nn, dist, gate);
  return dh3d_launch_status();
}

DH3D_API int dh3d_knn_sorted(const float *sorted, const float *gbox, int B, int N, int K, int32_t *nn,
                             float *dist, void *stream) {
  return knn_sorted_launch(sorted, gbox, B, N, K, nn, dist, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------ cell-list kNN
// The same operator on the grid spatial_sort_kernel lays over a cloud's bounding box (the top 12 bits of the sort key,
// dealt to the axes by extent: 16 x 16 x 16 on a cube, 32 x 32 x 4 on a street scene -- every grid cell is ONE contiguous
// range of the sorted records, cells[] holds the ranges).  Where the pruned scan above shares a candidate stream between 64 queries -- the union of their search balls,
// ~1800 candidates per query at 8 x 8192 -- this one gives every query its own cells.  L lanes per query, two passes:
//   (0) the 27 cells around the query's cell;
//   (1) every other cell of the BOX the ball of the K-th distance seen so far meets (its own radius per axis), skipping a
//       cell whose box lies beyond that distance.
// Each pass pools the surviving cells' records per query in LDS, the query's lanes then take the pooled candidates L apart
// into their own sorted K-lists, and the lists are merged with log2(L) bitonic exchange steps.  Exact after pass 1: the K
// nearest neighbours all lie within the K-th distance seen after pass 0, i.e. inside the box, and a skipped cell provably
// holds nothing closer (conservative box distance, ties included).  What the box pass cannot serve (no K-th distance after
// 27 cells, a ball wider than +-2 cells in x, more than 128 columns) restarts over the sort's 64-point group boxes; clouds
// whose points crowd into few cells (the sort's verdict, cells[kCellFlag]) are left to the pruned scan (dh3d_knn_grid).
// Same distances (reference rounding order, IEEE sqrt) and the same 64-bit (distance, CUB rank) order as every other
// kernel of this file: ids and distance bits are identical.
constexpr int kGridCap = 256;    // pooled candidates per query and pass (knn_grid_kernel)

__device__ __forceinline__ unsigned knn_spread4(unsigned v) {  // 4 bits -> every third bit (the sort's spread6 >> 6)
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}
__device__ __forceinline__ void knn_cswap(u64 &a, u64 &b) {
  const bool lt = b < a;
  const u64 lo = lt ? b : a, hi = lt ? a : b;
  a = lo; b = hi;
}
// the L lanes of a query end up with the same list: the 8 smallest keys of all their lists.  log2(L) exchange steps on
// the DPP crossbar (no LDS round trip): partner = lane ^ 1, lane ^ 2 (quad permutes), then (L = 8) 7 - lane within the
// eight (row_half_mirror: every lane meets one of the other quad, whose four lanes already agree).
template <int CTRL>
__device__ __forceinline__ u64 knn_dpp_u64(u64 v) {
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
  return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}
template <int CTRL>
__device__ __forceinline__ void knn_merge_step(KnnState<8> &st) {
  u64 c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 mine = st.keys[i], theirs = knn_dpp_u64<CTRL>(st.keys[7 - i]);
    c[i] = mine < theirs ? mine : theirs;  // ascending list against the partner's descending one: the 8 smallest, bitonic
  }
  knn_cswap(c[0], c[4]); knn_cswap(c[1], c[5]); knn_cswap(c[2], c[6]); knn_cswap(c[3], c[7]);
  knn_cswap(c[0], c[2]); knn_cswap(c[1], c[3]); knn_cswap(c[4], c[6]); knn_cswap(c[5], c[7]);
  knn_cswap(c[0], c[1]); knn_cswap(c[2], c[3]); knn_cswap(c[4], c[5]); knn_cswap(c[6], c[7]);
#pragma unroll
  for (int i = 0; i < 8; ++i) st.keys[i] = c[i];
}
template <int L>
__device__ __forceinline__ void knn_merge_sublanes(KnnState<8> &st) {
  if (L >= 2) knn_merge_step<0xB1>(st);   // quad_perm [1,0,3,2]
  if (L >= 4) knn_merge_step<0x4E>(st);   // quad_perm [2,3,0,1]
  if (L >= 8) knn_merge_step<0x141>(st);  // row_half_mirror
  const unsigned hb = (unsigned)(st.keys[7] >> 32);
  if (hb <= 0x7f800000u) {
    const float dk = __uint_as_float(hb);
    st.bound = __fmul_rn(__fmul_rn(dk, dk), 1.000001f);
  }
}

#ifdef DH3D_GRID_PROBE  // dev instrumentation (tools/knn_grid_probe.py): cycle stamps of the first wave of 64 workgroups
__device__ long long g_gprobe[64 * 8];
#define GPROBE(i) do { if (threadIdx.x == 0 && blockIdx.x < 64 && blockIdx.y == 0) g_gprobe[blockIdx.x * 8 + (i)] = clock64(); } while (0)
#else
#define GPROBE(i) do { } while (0)
#endif

// L = lanes per query (2, 4 or 8): fewer lanes = longer private candidate streams, but an insertion round serves 64 / L
// queries and the merge has log2(L) steps.
// SF > 0 (L = 4: a workgroup is 64 queries, as in the pruned scan): a cloud the sort flagged as crowded is served HERE by
// the pruned scan with SF waves per query group (knn_split_body on the same LDS block) -- one launch for both kinds of
// cloud; SF = 0: such clouds are left to a second launch (dh3d_knn_grid).
template <int L, int SF>
__global__ __launch_bounds__(256) void knn_grid_kernel(const float4 *__restrict__ sorted, const float *__restrict__ gbox,
                                                      const int *__restrict__ cells, int N, int K, int D, KnnLadder lad,
                                                      int32_t *__restrict__ nn, float *__restrict__ dist) {
  constexpr int QB = 256 / L;  // queries per workgroup
  static_assert(SF == 0 || (L == 4 && SF <= 4), "the merged scan: 64 queries per workgroup, at most four waves");
  constexpr int kGridLds = 3 * 64 * 4 + QB * kGridCap * 2 + QB * 4;
  constexpr int kSplitLds = SF == 2 ? 2 * knn_split_lds_bytes<2>() : SF > 0 ? knn_split_lds_bytes<(SF > 0 ? SF : 1)>() : 0;
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[kGridLds > kSplitLds ? kGridLds : kSplitLds];
  const int b = blockIdx.y, lane = threadIdx.x & 63, sub = lane & (L - 1);
  if (cells[(size_t)b * kCellInts + kCellFlag]) {  // dense cells: the pruned scan takes this cloud
    if constexpr (SF > 0) {
      // (the group a workgroup serves is rotated by the cloud: workgroups x, x + 256, ... share a CU, and with clouds of one
      //  kind in a batch -- the same street, the same sensor -- the same x is the same kind of region in all of them: the
      //  slow far-field groups of every cloud would meet on the same CUs)
      const int ngq = (N + 63) / 64;
      if constexpr (SF == 2) {
        // TWO query groups per workgroup, two waves each (half the workgroups of a crowded cloud leave at once): a CU then
        // carries eight groups instead of four, and the launch ends with its most loaded CU
        const int half = (int)(threadIdx.x >> 7), pairs = (ngq + 1) / 2;
        if ((int)blockIdx.x >= pairs) return;
        const int g0 = (int)blockIdx.x + half * pairs;
        if (g0 >= ngq) return;
        knn_split_body<8, 2>(sorted, gbox, N, K, lad, nn, dist, b, (int)(((unsigned)g0 + 37u * (unsigned)b) % (unsigned)ngq),
                             s_raw + half * knn_split_lds_bytes<2>(), half * 2);
      } else if (threadIdx.x < 64 * SF) {
        knn_split_body<8, (SF > 0 ? SF : 1)>(sorted, gbox, N, K, lad, nn, dist, b, (int)((blockIdx.x + 37u * (unsigned)b) % (unsigned)ngq), s_raw);
      }
    }
    return;
  }
  // (the cell lists too: a query group's work depends on where it sits in the Morton order -- the cloud's border, the big
  //  jumps of the curve -- which is the same place in every cloud of a batch: rotated by the cloud like the scan's)
  const int qi = (int)((blockIdx.x + 37u * (unsigned)b) % gridDim.x) * QB + threadIdx.x / L;
  const bool valid = qi < N;
  const float4 *sc = sorted + (size_t)b * N;
  const int *ct = cells + (size_t)b * kCellInts;
  const float *hd = reinterpret_cast<const float *>(ct) + 4100;
  const float lo[3] = {hd[0], hd[1], hd[2]}, scl[3] = {hd[3], hd[4], hd[5]};
  const float4 qr = sc[valid ? qi : N - 1];
  const float q[3] = {qr.x, qr.y, qr.z};
  // The grid: the sort dealt the cell code's 12 bits to the axes by extent (spatial.hip: `sched`, step 0 = the top bit;
  // a cube: z y x z y x ... = 16 x 16 x 16).  D = low bits of the code left out: a coarser grid whose cells are still
  // contiguous ranges of the order -- 2^D consecutive cells of the table.
  const unsigned sched = (unsigned)ct[4107];
  int nb[3] = {0, 0, 0}, drop[3] = {0, 0, 0};
#pragma unroll
  for (int s = 0; s < 12; ++s) {
    const int a = (int)((sched >> (2 * s)) & 3u);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      nb[c] += (int)(a == c);
      drop[c] += (int)(a == c && s >= 12 - D);
    }
  }
  const int span = 1 << D;
  int cq[3], gmax[3];
  float w[3], eps[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    cq[a] = min((4 << nb[a]) - 1, max(0, (int)((q[a] - lo[a]) * scl[a]))) >> (2 + drop[a]);  // the sort's cell arithmetic
    gmax[a] = ((1 << nb[a]) >> drop[a]) - 1;
    w[a] = (float)(4 << drop[a]) / scl[a];  // cell width
    eps[a] = 2.5e-6f * ((float)(4 << nb[a]) / scl[a]);  // a point may sit outside its cell's nominal interval by a few ulps of the extent
  }
  // the table index of cell (ax, ay, az): every axis' cell number with its bits at their places in the code (per
  // workgroup, in LDS: 64 entries per axis)
  unsigned (*s_ctab)[64] = reinterpret_cast<unsigned (*)[64]>(s_raw);   // [3][64]
  if (threadIdx.x < 192) {
    const int a = threadIdx.x >> 6, v = threadIdx.x & 63;
    unsigned code = 0;
    int rem = nb[a] - drop[a];
    for (int s = 0; s < 12 - D; ++s)
      if ((int)((sched >> (2 * s)) & 3u) == a && rem > 0) {
        --rem;
        code |= (unsigned)((v >> rem) & 1) << (11 - s);
      }
    s_ctab[a][v] = code;
  }
  __syncthreads();
  auto cell_code = [&](int ax, int ay, int az) { return s_ctab[0][ax & 63] | s_ctab[1][ay & 63] | s_ctab[2][az & 63]; };
  KnnState<8> st;
#pragma unroll
  for (int i = 0; i < 8; ++i) st.keys[i] = ~0ull;
  st.bound = valid ? INFINITY : -1.f;

  // conservative squared distance from the query to cells cq[a] + o along one axis (0 inside the cell's own slab)
  auto axis_d2 = [&](int a, int o) {
    const float l = lo[a] + (float)(cq[a] + o) * w[a];
    const float d = fmaxf(fmaxf(fmaxf(l - q[a], q[a] - (l + w[a])), 0.f) - eps[a], 0.f);
    return d * d;
  };
  // the 5 x 5 x 5 block: everything that depends on the x offset alone is computed once (five slabs: distance, cell bits,
  // inside the grid or not); a lane then walks (dy, dz) columns and, per column, the five x offsets with constants
  float d2x[5];
  unsigned bitx[5];
  bool okx[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) {
    const int ax = cq[0] + t - 2;
    okx[t] = (unsigned)ax <= (unsigned)gmax[0];
    d2x[t] = okx[t] ? axis_d2(0, t - 2) : INFINITY;
    bitx[t] = s_ctab[0][ax & 63];
  }
  // is the K-th distance strictly inside the block of radius R around the query's cell?  (faces on the grid's border
  // have nothing behind them)
  auto inside = [&](int R) {
    float G = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (cq[a] - R > 0) G = fminf(G, (q[a] - (lo[a] + (float)(cq[a] - R) * w[a])) - eps[a]);
      if (cq[a] + R < gmax[a]) G = fminf(G, ((lo[a] + (float)(cq[a] + R + 1) * w[a]) - q[a]) - eps[a]);
    }
    if (G == INFINITY) return true;
    G = fmaxf(G, 0.f);
    return st.bound < G * G * 0.99999f;  // bound = the (1 + 2^-20)-inflated square of the K-th distance, inf while the list is short
  };

  // Cells hold 0..~8 points: walking them cell by cell leaves most lanes of the wave idle in every iteration (and the
  // ~45-instruction insertion runs for the whole wave whenever one lane inserts).  So a pass first POOLS: every lane
  // appends the record indices of its surviving cells to its query's list in LDS (one LDS atomic per cell for the
  // place), then the query's lanes take the pooled candidates L apart, four per lane in flight -- every lane busy in
  // every iteration, ~5x fewer iterations of the candidate loop.  The result does not depend on who scans what (keys
  // are unique, the merge sorts).  A list that is full (dense clusters) sends the rest of the cell down the direct path.
  // After a merge all L lanes hold the SAME list: before they scan on, all but one empty theirs (the bound stays), or the
  // next merge would count every entry L times
  auto keep_one_copy = [&]() {
    if (sub != 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) st.keys[i] = ~0ull;
    }
  };
  unsigned short (*s_list)[kGridCap] = reinterpret_cast<unsigned short (*)[kGridCap]>(s_raw + 3 * 64 * 4);   // [QB][kGridCap]
  int *s_cnt = reinterpret_cast<int *>(s_raw + 3 * 64 * 4 + QB * kGridCap * 2);                                // [QB]
  const int qs = threadIdx.x / L;
  auto direct = [&](int j) {
    const float4 r = sc[j];
    const float dx = r.x - q[0], dy = r.y - q[1], dz = r.z - q[2];
    knn_offer<8, false>(st, fmaf(dz, dz, fmaf(dy, dy, dx * dx)), __float_as_int(r.w), lad);
  };
  auto enqueue = [&](int beg, int end) {
    const int len = end - beg;
    if (len <= 0) return;
    const int off = atomicAdd(&s_cnt[qs], len);
    const int fit = min(len, kGridCap - off);
    for (int k = 0; k < fit; ++k) s_list[qs][off + k] = (unsigned short)(beg + k);
    for (int k = max(fit, 0); k < len; ++k) direct(beg + k);
  };
  auto drain = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int total = min(s_cnt[qs], kGridCap);
#pragma unroll 1
    for (int base = sub; base < total; base += 4 * L) {
      float4 r[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * L;
        ok[u] = i < total;
        r[u] = sc[s_list[qs][min(i, total - 1)]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float dx = r[u].x - q[0], dy = r[u].y - q[1], dz = r[u].z - q[2];
        const float s2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));  // the reference's rounding order
        knn_offer<8, false>(st, ok[u] ? s2 : __builtin_nanf(""), __float_as_int(r[u].w), lad);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // One column (dy, dz: offsets + 2) of the 5 x 5 x 5 block, the cells at the x offsets of TMASK: ranges of all of them in
  // flight together (no load under the per-cell branches), then the box tests and the pooling.
  auto column = [&](auto tmask, int dy, int dz) __attribute__((always_inline)) {
    constexpr int TMASK = decltype(tmask)::value;
    const int ay = cq[1] + dy - 2, az = cq[2] + dz - 2;
    if ((unsigned)ay > (unsigned)gmax[1] || (unsigned)az > (unsigned)gmax[2]) return;
    const float d2yz = axis_d2(1, dy - 2) + axis_d2(2, dz - 2);
    const unsigned bityz = cell_code(0, ay, az);
    int beg[5], end[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
      if (TMASK >> t & 1) {
        beg[t] = ct[bityz | bitx[t]];
        end[t] = ct[(bityz | bitx[t]) + span];
      }
    if constexpr (TMASK != 0b01110) {  // shell 2: most cells fail the box test -- the few that pass one by one
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (TMASK >> t & 1) {
          if (!okx[t] || (d2yz + d2x[t]) * 0.99999f > st.bound) continue;
          enqueue(beg[t], end[t]);
        }
    } else {
      // shells 0-1: (nearly) every cell is taken -- they reserve their places in the query's list together (one LDS atomic,
      // not one per cell); cells hold 0..~8 records: the first four places are written without a loop
      int len[5], tot = 0;
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (TMASK >> t & 1) {
          len[t] = okx[t] && !((d2yz + d2x[t]) * 0.99999f > st.bound) ? end[t] - beg[t] : 0;
          tot += len[t];
        }
      if (tot == 0) return;
      int off = atomicAdd(&s_cnt[qs], tot);
#pragma unroll
      for (int t = 0; t < 5; ++t)
        if (TMASK >> t & 1) {
          const int fit = min(len[t], kGridCap - off);  // (a full list sends the rest of the cell down the direct path)
          unsigned short *dst = &s_list[qs][off];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < fit) dst[k] = (unsigned short)(beg[t] + k);
          for (int k = 4; k < fit; ++k) dst[k] = (unsigned short)(beg[t] + k);
          for (int k = max(fit, 0); k < len[t]; ++k) direct(beg[t] + k);
          off += len[t];
        }
    }
  };
  auto begin_pass = [&]() __attribute__((always_inline)) {
    if (sub == 0) s_cnt[qs] = 0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // The columns are dealt DENSELY: walking all 25 with the other pass's columns skipped cost the wave the full body in
  // nearly every iteration (some lane always had a live column) -- clock stamps: the first pass's 27 cells took as long
  // as the second's 98 (tools/knn_grid_probe.py), and the two together two thirds of the kernel.
  GPROBE(0);
  // pass 0: shells 0-1 = the nine inner columns at dx in -1..1
  begin_pass();
#pragma unroll 1
  for (int ci = sub; ci < 9; ci += L) column(std::integral_constant<int, 0b01110>{}, 1 + ci % 3, 1 + ci / 3);
  GPROBE(1);
  drain();
  GPROBE(2);
  knn_merge_sublanes<L>(st);
  GPROBE(3);
  // pass 1: every other cell the ball of the merged bound meets.  The K nearest neighbours all lie within the K-th distance
  // seen so far, so the cells to look at are a BOX with its own radius per axis -- two cells each way in a uniform cloud
  // (the 5 x 5 x 5 block of the first versions of this kernel), one cell in x / y and four thin layers in z on the ground
  // plane of a street scene whose bounding box is a tenth as high as wide.  Exact after this pass: nothing outside the box
  // is within the bound.  (The fixed block + shell-by-shell widening of the first version took 5.9 ms instead of 30 us
  // on such a scene, tools/knn_scene_bench.py.)  Columns (dy, dz) of the box are dealt over the query's lanes, the five x
  // offsets of a column are compile-time; a ball that is wider than that in x, or a box of more than 128 columns, or no
  // K-th distance yet (fewer than K points in the 27 cells) goes to the restart below.
  keep_one_copy();
  begin_pass();
  bool exact = false;
  int ylo = -2, zlo = -2, ny = 5, ncol = valid ? 25 : 0;  // (no K-th distance yet, or a ball too wide for this pass: the 5 x 5 x 5 block)
  if (valid && st.bound < INFINITY) {
    const float r = sqrtf(st.bound) * 1.0001f;
    int dlo[3], dhi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float inv = scl[a] / (float)(4 << drop[a]);  // cells per unit length
      dlo[a] = min(0, max(0, (int)floorf((q[a] - r - eps[a] - lo[a]) * inv)) - cq[a]);
      dhi[a] = max(0, min(gmax[a], (int)floorf((q[a] + r + eps[a] - lo[a]) * inv)) - cq[a]);
    }
    const int by = dhi[1] - dlo[1] + 1, bc = by * (dhi[2] - dlo[2] + 1);
    if (dlo[0] >= -2 && dhi[0] <= 2 && bc <= 128) {
      exact = true;
      ylo = dlo[1]; zlo = dlo[2]; ny = by; ncol = bc;
    }
  }
#pragma unroll 1
  for (int ci = sub; ci < ncol; ci += L) {
    const int iz = ci / ny, dy = ylo + (ci - iz * ny), dz = zlo + iz;
    if (abs(dy) <= 1 && abs(dz) <= 1) column(std::integral_constant<int, 0b10001>{}, dy + 2, dz + 2);  // (dx in -1..1: pass 0)
    else column(std::integral_constant<int, 0b11111>{}, dy + 2, dz + 2);
  }
  GPROBE(4);
  drain();
  GPROBE(5);
  knn_merge_sublanes<L>(st);
  GPROBE(6);
  // What is left -- sparse corners, outliers, clusters that put everything into a few cells -- restarts against the bound
  // it has, over the 64-point groups of the Morton order whose box the search ball meets (the sort's group boxes): exact
  // (the list is emptied first, every point within the bound is offered again, nothing twice) and bounded: N / 64 box
  // tests per query dealt over its lanes.
  const bool done = !valid || exact || inside(2);
  if (__any(!done)) {
    if (done) {
      keep_one_copy();
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) st.keys[i] = ~0ull;  // (the bound stays: it is the K-th distance of a superset of nothing less)
      const int NG = (N + 63) >> 6;
      const float4 *gb = reinterpret_cast<const float4 *>(gbox) + (size_t)b * NG * 2;  // [lo.xyz,_ | hi.xyz,_]
#pragma unroll 1
      for (int g = sub; g < NG; g += L) {
        const float4 lo4 = gb[2 * g], hi4 = gb[2 * g + 1];
        const float ex = fmaxf(fmaxf(lo4.x - q[0], q[0] - hi4.x), 0.f), ey = fmaxf(fmaxf(lo4.y - q[1], q[1] - hi4.y), 0.f),
                    ez = fmaxf(fmaxf(lo4.z - q[2], q[2] - hi4.z), 0.f);
        if ((ex * ex + ey * ey + ez * ez) * 0.99999f > st.bound) continue;
        const int j1 = min(N, g * 64 + 64);
#pragma unroll 1
        for (int j = g * 64; j < j1; j += 8) {  // eight records in flight
          float4 r8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) r8[u] = sc[min(j + u, j1 - 1)];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float dx = r8[u].x - q[0], dy = r8[u].y - q[1], dz = r8[u].z - q[2];
            const float s2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
            knn_offer<8, false>(st, j + u < j1 ? s2 : __builtin_nanf(""), __float_as_int(r8[u].w), lad);
          }
        }
      }
    }
    knn_merge_sublanes<L>(st);
  }
  if (valid) {
    const int y = __float_as_int(qr.w);  // the query's original index
#pragma unroll
    for (int e = 0; e < 8 / L; ++e) {
      const int slot = e * L + sub;
      if (slot >= K) break;
      u64 key = st.keys[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) key = slot == i ? st.keys[i] : key;
      const size_t o = ((size_t)b * N + y) * K + slot;
      if (key == ~0ull) {  // fewer than K points: reference pads with id -1 / FLT_MAX (:110-111)
        nn[o] = -1;
        dist[o] = FLT_MAX;
      } else {
        const unsigned tb = (unsigned)key;
        nn[o] = (int)(((tb % (unsigned)lad.cv) << lad.log2ct) + tb / (unsigned)lad.cv);
        dist[o] = __uint_as_float((unsigned)(key >> 32));
      }
    }
  }
}

DH3D_API int dh3d_knn_grid(const float *sorted, const float *gbox, const int32_t *cells, int B, int N, int K, int32_t *nn,
                           float *dist, void *stream) {
  DH3D_REQUIRE(sorted && gbox && cells && nn && dist && B > 0 && N > 0 && K > 0);
  DH3D_SUPPORTED(K <= 8 && N <= 16384 && B <= 65535);
  const KnnLadder lad = knn_ladder(N);
#ifndef DH3D_GRID_LANES
#define DH3D_GRID_LANES 4
#endif
  constexpr int kLanes = DH3D_GRID_LANES;  // per query
  // one to two points per cell: all 4096 cells down to 4096 points (measured: 32 x 4096 77 us against 83 with half the
  // cells), one bit of the cell code less for every halving below that
  int D = 0;
  while (D < 6 && ((long long)N << D) <= 3072) ++D;
#ifdef DH3D_GRID_DROP_BIAS
  D = D + (DH3D_GRID_DROP_BIAS) < 0 ? 0 : D + (DH3D_GRID_DROP_BIAS);
#endif
  // The clouds whose points crowd into few cells (the sort's verdict, cells[kCellFlag]) go to the pruned scan -- in the SAME
  // launch (four waves per query group = the cell lists' 256-thread workgroup) up to 4096 query groups, as a second launch
  // (whose workgroups leave at once for the other clouds) for the one-wave scan beyond that
  const long long groups = (long long)((N + 63) / 64) * B;
  const dim3 grid(dh3d_cdiv(N, 256 / kLanes), B);
  const float4 *so = reinterpret_cast<const float4 *>(sorted);
  if (kLanes == 4 && groups <= 4096) {
    // (four waves per query group up to 1280 groups, as the scan on its own; beyond that two groups per workgroup with two
    // waves each -- ONE two-wave group per 256-thread workgroup left half its waves unused while it held its LDS: 139.8 us)
    if (groups > 1280)   // many query groups: two groups per workgroup, two waves each (32 x 4096 demo clouds: 97.7 us against 117.9)
      hipLaunchKernelGGL((knn_grid_kernel<4, 2>), grid, dim3(256), 0, (hipStream_t)stream, so, gbox, cells, N, K, D, lad, nn, dist);
    else
      hipLaunchKernelGGL((knn_grid_kernel<4, 4>), grid, dim3(256), 0, (hipStream_t)stream, so, gbox, cells, N, K, D, lad, nn, dist);
    return dh3d_launch_status();
  }
  hipLaunchKernelGGL((knn_grid_kernel<kLanes, 0>), grid, dim3(256), 0, (hipStream_t)stream, so, gbox, cells, N, K, D, lad, nn, dist);
  if (dh3d_launch_status() != DH3D_OK) return DH3D_ERR_LAUNCH;
  return knn_sorted_launch(sorted, gbox, B, N, K, nn, dist, cells, stream);
}

#ifdef DH3D_GRID_PROBE
DH3D_API int dh3d_grid_probe_read(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gprobe), sizeof(long long) * n) == hipSuccess ? 0 : 3;
}
#endif

#ifdef DH3D_KNN_PROBE
DH3D_API int dh3d_knn_probe_read(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_kprobe), sizeof(long long) * n) == hipSuccess ? 0 : 3;
}
#endif
