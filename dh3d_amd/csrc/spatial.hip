// Spatial (Morton) ordering of a cloud for gfx950 -- a preprocessing step with no counterpart in the
// reference.  It changes no result: kNN and FPS stay exact (bit-identical ids), but both can then skip
// work that provably cannot matter, because 64 consecutive points of the order form a compact group
// with a tight bounding box:
//   * knn_sorted_kernel (knn.hip) visits candidate groups nearest-first and skips a whole group when its
//     box is farther from the query group's box than every lane's current K-th distance;
//   * fps_sorted_kernel (fps.hip) re-evaluates a group's min-distances only when the new sample is
//     closer to the group's box than the group's current maximum.
// One 1024-lane workgroup per cloud: cloud bounding box -> 18-bit cell code -> stable in-LDS radix sort of 32-bit
// keys (cell << 14 | original index; N <= 16384) by cell, three 6-bit passes -> sorted float4 records (x, y, z,
// bits(original index)) and one box per group of 64.
// The cell code (round 5) is a Morton code whose bits are dealt to the axes BY EXTENT: the top 12 bits (the grid of the
// cell-list kNN) go one at a time to the axis whose cells are currently the widest (ties: z, y, x -- a cube gets 4 + 4 + 4
// bits interleaved z y x z y x ..., exactly the plain Morton code of rounds 1-4), two more bits per axis follow.  A street
// scene 36 x 36 x 8 m gets 5 + 5 + 2: 32 x 32 x 4 cells of 1.1 x 1.1 x 2 m instead of 16 x 16 x 16 cells of
// 2.25 x 2.25 x 0.5 m -- with the ground plane in one or two layers either way, the 27 cells around a query then hold ~60
// points instead of ~230.  The schedule (which axis every one of the 12 bits belongs to) travels in the cell table's header.
// (The first version was a bitonic network: 91 barrier-separated stages, 70 us for 8 x 8192 -- on the critical
//  path of both the kNN and the FPS.  The radix sort produces the identical order: stable by cell = (cell, index).)
#include <math.h>

#include "common.h"
#include "wave_ops.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = 16;
constexpr int kCellInts = 4112;  // ints per cloud of the cell table (dh3d_spatial_sort_cells)

#ifdef DH3D_SORT_PROBE  // dev instrumentation (tools/sort_probe.py): s_memtime stamps of wave 0 / wave 15 of a few clouds
__device__ long long g_sprobe[8 * 2 * 16];
#define SPROBE(i) do { if ((threadIdx.x == 0 || threadIdx.x == 960) && blockIdx.x < 8) \
  g_sprobe[(blockIdx.x * 2 + (threadIdx.x != 0)) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define SPROBE(i) do { } while (0)
#endif

__device__ __forceinline__ unsigned spread6(unsigned v) {  // 6 bits -> every third bit
  v &= 63u;
  v = (v | (v << 8)) & 0x300Fu;
  v = (v | (v << 4)) & 0x30C3u;
  v = (v | (v << 2)) & 0x9249u;
  return v;
}
constexpr unsigned kSchedCube = 0x186186u;  // z y x z y x ... : 2 1 0 repeated, step 0 in the low bits

// Axis of step s of the 18-step bit schedule (step 0 = the code's most significant bit): the 12 grid steps from `sched`
// (2 bits each), then z y x z y x.
__device__ __forceinline__ int sched_axis(unsigned sched, int s) { return s < 12 ? (int)((sched >> (2 * s)) & 3u) : 2 - (s - 12) % 3; }

// The 12 grid bits dealt by extent: every step goes to the axis whose cells are the widest so far (at most 6 per axis;
// ties -- cell widths within 25 % of each other -- go to the highest axis first, so a (near-)cube gets z y x z y x ...).  nb[a] = grid bits of axis a.
__device__ __forceinline__ unsigned deal_grid_bits(const float ext[3], int nb[3]) {
  // (scalars and selects only: an array indexed by the winning axis would live in scratch memory)
  float c0 = ext[0], c1 = ext[1], c2 = ext[2];
  int n0 = 0, n1 = 0, n2 = 0;
  unsigned sched = 0;
#pragma unroll
  for (int s = 0; s < 12; ++s) {
    int a = -1;
    float best = -1.f;
    if (n2 < 6) { a = 2; best = c2; }
    if (n1 < 6 && (a < 0 || c1 > best * 1.25f)) { best = a < 0 ? c1 : fmaxf(best, c1); a = 1; }
    if (n0 < 6 && (a < 0 || c0 > best * 1.25f)) { best = a < 0 ? c0 : fmaxf(best, c0); a = 0; }
    n0 += (int)(a == 0); n1 += (int)(a == 1); n2 += (int)(a == 2);
    c0 = a == 0 ? c0 * 0.5f : c0; c1 = a == 1 ? c1 * 0.5f : c1; c2 = a == 2 ? c2 * 0.5f : c2;
    sched |= (unsigned)a << (2 * s);
  }
  nb[0] = n0; nb[1] = n1; nb[2] = n2;
  return sched;
}

// cells (may be NULL): per cloud kCellInts ints -- [0, 4096]: first sorted position of every grid cell in the order of
// the cells' 12-bit codes (the top 12 bits of the sort key; an empty cell's entry = the next cell's), [4096] = N;
// [4100..4105] as floats: the grid's origin (x, y, z) and the quantisation scale 2^(nb + 2) / extent per axis (a cell = 4
// quantisation steps); [4107]: the bit schedule (12 x 2 bits, step 0 = the code's top bit) -- the grid the cell-list kNN
// (knn.hip: knn_grid_kernel) searches.
template <int PPT>
__global__ __launch_bounds__(kThreads) void spatial_sort_kernel(const float *__restrict__ xyz, int N,
                                                               int npad, float4 *__restrict__ sorted,
                                                               float *__restrict__ gbox, int *__restrict__ cells,
                                                               int occ_min) {
  extern __shared__ __attribute__((aligned(16))) unsigned s_raw[];  // keys[2][npad] | hist[16][64] | 6*kWaves floats | tab[3][256]
  unsigned *s_hist = s_raw + 2 * npad;
  float *s_red = reinterpret_cast<float *>(s_hist + kWaves * 64);
  unsigned *s_tab = reinterpret_cast<unsigned *>(s_red + 6 * kWaves);  // axis value -> its bits at their places in the code
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *pc = xyz + (size_t)b * N * 3;
  const int NG = (N + 63) / 64;

  // ---- cloud bounding box.  Wave w owns the contiguous index range [w*SEG, (w+1)*SEG), lane-consecutive in
  // steps of 64: the arrangement the stable radix passes below need.
  const int SEG = npad / kWaves;  // = 64 * PPT
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  SPROBE(0);
  float px[PPT], py[PPT], pz[PPT];
  // every load of the thread is requested before the first one is used: no load sits under a branch (a padding lane
  // re-reads the last point) -- with `if (k < N) load` the compiler waited for each of the PPT loads in turn
  // (s_waitcnt vmcnt(0) after every one: PPT dependent round trips to HBM at the head of the step's critical chain)
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = min(wave * SEG + j * 64 + lane, N - 1);
    px[j] = pc[(size_t)k * 3]; py[j] = pc[(size_t)k * 3 + 1]; pz[j] = pc[(size_t)k * 3 + 2];
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) {  // (a re-read point changes no minimum / maximum)
    lo[0] = fminf(lo[0], px[j]); hi[0] = fmaxf(hi[0], px[j]);
    lo[1] = fminf(lo[1], py[j]); hi[1] = fmaxf(hi[1], py[j]);
    lo[2] = fminf(lo[2], pz[j]); hi[2] = fmaxf(hi[2], pz[j]);
  }
  SPROBE(1);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float wl = wave_min_f32(lo[a]), wh = wave_max_f32(hi[a]);
    if (lane == 0) { s_red[wave * 6 + a] = wl; s_red[wave * 6 + 3 + a] = wh; }
  }
  __syncthreads();
  float scale[3], ext[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float l = INFINITY, h = -INFINITY;
    for (int w = 0; w < kWaves; ++w) { l = fminf(l, s_red[w * 6 + a]); h = fmaxf(h, s_red[w * 6 + 3 + a]); }
    lo[a] = l;
    ext[a] = h - l;
  }
  int nb[3];
  const unsigned sched = deal_grid_bits(ext, nb);  // (every thread: 36 compares)
#pragma unroll
  for (int a = 0; a < 3; ++a) scale[a] = (float)(4 << nb[a]) / fmaxf(ext[a], 1e-30f);  // nb + 2 bits per axis
  if (cells && tid == 0) {  // the grid's header, now: origin, scales and schedule need not stay in registers until the end
    float *hd = reinterpret_cast<float *>(cells + (size_t)b * kCellInts) + 4100;
    hd[0] = lo[0]; hd[1] = lo[1]; hd[2] = lo[2];
    hd[3] = scale[0]; hd[4] = scale[1]; hd[5] = scale[2];
    reinterpret_cast<int *>(hd)[7] = (int)sched;
  }
  // (near-)cubic clouds get the plain Morton code of rounds 1-4 from three bit-spreads; any other schedule goes through
  // per-axis deposit tables in LDS (a table build, a barrier and three look-ups per point: ~2.5 us at 8 x 8192)
  const bool cube = sched == kSchedCube;  // workgroup-uniform
  if (!cube) {
    if (tid < 768) {  // the three axes' deposit tables
      const int a = tid >> 8, v = tid & 255;
      unsigned code = 0;
      int rem = nb[a] + 2;
      for (int st = 0; st < 18; ++st)
        if (sched_axis(sched, st) == a && rem > 0) {
          --rem;
          code |= (unsigned)((v >> rem) & 1) << (17 - st);
        }
      s_tab[tid] = code;
    }
    __syncthreads();
  }
  // ---- keys (registers): cell << 14 | original index; padding sorts last.  Four points at a time; the
  // 16-points-per-thread instantiation reads its points AGAIN here (L2 hits) instead of keeping 48 coordinate registers
  // alive across the grid set-up above (it has exactly 128 registers and spilled ~60 of them to scratch otherwise).
  constexpr bool RELOAD = PPT >= 16;
  unsigned key[PPT];
  const int q0 = (4 << nb[0]) - 1, q1 = (4 << nb[1]) - 1, q2 = (4 << nb[2]) - 1;
#pragma unroll
  for (int j0 = 0; j0 < PPT; j0 += 4) {
    float rx[4], ry[4], rz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j0 + u >= PPT) continue;
      if (RELOAD) {
        const int k = min(wave * SEG + (j0 + u) * 64 + lane, N - 1);
        rx[u] = pc[(size_t)k * 3]; ry[u] = pc[(size_t)k * 3 + 1]; rz[u] = pc[(size_t)k * 3 + 2];
      } else {
        rx[u] = px[j0 + u]; ry[u] = py[j0 + u]; rz[u] = pz[j0 + u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (j0 + u >= PPT) continue;
      const int k = wave * SEG + (j0 + u) * 64 + lane;
      const unsigned cx = (unsigned)min(q0, max(0, (int)((rx[u] - lo[0]) * scale[0])));
      const unsigned cy = (unsigned)min(q1, max(0, (int)((ry[u] - lo[1]) * scale[1])));
      const unsigned cz = (unsigned)min(q2, max(0, (int)((rz[u] - lo[2]) * scale[2])));
      const unsigned cell = cube ? spread6(cx) | (spread6(cy) << 1) | (spread6(cz) << 2)   // (workgroup-uniform choice)
                                 : s_tab[cx] | s_tab[256 + cy] | s_tab[512 + cz];
      key[j0 + u] = k < N ? (cell << 14) | (unsigned)k : 0xFFFFFFFFu;
    }
    if (RELOAD) __builtin_amdgcn_sched_barrier(0);
  }
  // ---- three stable counting passes over 6-bit digits of the cell.  Per pass: every wave ranks its keys digit by
  // digit (same-digit lanes found with 6 ballots; the running per-(wave, digit) count lives in LDS), one block scan
  // of the 64 x 16 counts in (digit, wave) order gives the bases, and the keys move to their places.
  SPROBE(2);
  unsigned *src = s_raw, *dst = s_raw + npad;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = 14 + 6 * pass;
    s_hist[tid] = 0u;  // kWaves * 64 == kThreads
    __syncthreads();
    unsigned local[PPT];
    unsigned *whist = s_hist + wave * 64;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const unsigned d = (key[j] >> shift) & 63u;
      unsigned long long same = ~0ull;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const bool bit = (d >> i) & 1u;
        const unsigned long long bal = __ballot(bit);
        same &= bit ? bal : ~bal;
      }
      const unsigned before = whist[d];                      // keys of this digit in the wave's earlier steps
      const unsigned rank = (unsigned)__popcll(same & lt_mask);
      local[j] = before + rank;
      if (rank == 0) whist[d] = before + (unsigned)__popcll(same);  // one lane per digit; reads above precede it
    }
    if (pass == 0) SPROBE(3);
    __syncthreads();
    if (pass == 0) SPROBE(4);
    // exclusive scan over (digit, wave): thread t <-> digit t / 16, wave t % 16
    {
      const int d = tid >> 4, w = tid & 15;
      const unsigned v = s_hist[w * 64 + d];
      unsigned inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = __shfl_up(inc, off, 64);
        if (lane >= off) inc += up;
      }
      __syncthreads();  // every count has been read
      if (lane == 63) s_red[wave] = __uint_as_float(inc);  // wave totals (bit pattern)
      __syncthreads();
      unsigned base = 0;
      for (int ww = 0; ww < wave; ++ww) base += __float_as_uint(s_red[ww]);
      s_hist[w * 64 + d] = base + inc - v;
    }
    __syncthreads();
    if (pass == 0) SPROBE(5);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const unsigned d = (key[j] >> shift) & 63u;
      dst[whist[d] + local[j]] = key[j];
    }
    __syncthreads();
    if (pass == 0) SPROBE(6);
    // reload in the wave-contiguous arrangement for the next pass
#pragma unroll
    for (int j = 0; j < PPT; ++j) key[j] = dst[wave * SEG + j * 64 + lane];
    unsigned *t = src; src = dst; dst = t;
    if (pass == 0) SPROBE(7);
  }
  SPROBE(8);
  const unsigned *s_keys = src;  // sorted
  // ---- sorted records + one box per 64: lane l of wave w owns positions (w + 16 j) * 64 + l.  Gather first (all PPT
  // rows requested together, padding lanes re-read the cloud's last sorted point), then store and reduce.
  float gx[PPT], gy[PPT], gz[PPT];
  int gk[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int i = min((wave + kWaves * j) * 64 + lane, N - 1);
    gk[j] = (int)(s_keys[i] & 0x3FFFu);
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    gx[j] = pc[(size_t)gk[j] * 3]; gy[j] = pc[(size_t)gk[j] * 3 + 1]; gz[j] = pc[(size_t)gk[j] * 3 + 2];
  }
  SPROBE(9);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int g = wave + kWaves * j;
    const int i = g * 64 + lane;
    const float x = gx[j], y = gy[j], z = gz[j];
    float lx = INFINITY, ly = INFINITY, lz = INFINITY, hx = -INFINITY, hy = -INFINITY, hz = -INFINITY;
    if (i < N) {
      sorted[(size_t)b * N + i] = make_float4(x, y, z, __int_as_float(gk[j]));
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
  if (cells) {
    int *ct = cells + (size_t)b * kCellInts;
    // position i opens every cell in (cell(i - 1), cell(i)]; the cells behind the last point open at N
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int i = wave * SEG + j * 64 + lane;
      if (i < N) {
        const int cur = (int)(s_keys[i] >> 20), prev = i > 0 ? (int)(s_keys[i - 1] >> 20) : -1;
        for (int c = prev + 1; c <= cur; ++c) ct[c] = i;
        if (i == N - 1)
          for (int c = cur + 1; c <= 4096; ++c) ct[c] = N;
      }
    }
    // [4106]: is this cloud one for cell lists?  Not when its points crowd into far fewer cells than a uniform cloud of the
    // same size would occupy (street scenes: ground plane + walls in a bounding box a tenth as high as wide; clusters) --
    // dh3d_knn_grid then runs the pruned scan of the Morton order for this cloud (tools/knn_scene_bench.py, 8 x 8192: cell
    // lists 30 us on a uniform cloud, 66 on a uniform 60 x 60 x 8 slab, 410 on a scene where the scan takes 190; occupied
    // cells 3550 / 3535 / 1520 of 4096).
    int occupied = 0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int i = wave * SEG + j * 64 + lane;
      if (i < N) occupied += (int)(i == 0 || (s_keys[i] >> 20) != (s_keys[i - 1] >> 20));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) occupied += __shfl_xor(occupied, off, 64);
    __shared__ int s_occupied;
    if (tid == 0) s_occupied = 0;
    __syncthreads();
    if (lane == 0) atomicAdd(&s_occupied, occupied);
    __syncthreads();
    if (tid == 0) ct[4106] = s_occupied < occ_min ? 1 : 0;   // (the grid's header was written at the top)
  }
  SPROBE(10);
}

template <int PPT>
int sort_launch(const float *xyz, int B, int N, float4 *sorted, float *gbox, int *cells, hipStream_t s) {
  int npad = 2;
  while (npad < N) npad <<= 1;
  if (npad < 64 * kWaves) npad = 64 * kWaves;  // one 64-key step per wave at least
  const size_t lds = sizeof(unsigned) * (2 * (size_t)npad + kWaves * 64 + 3 * 256) + sizeof(float) * 6 * kWaves;
  DH3D_ALLOW_BIG_LDS((spatial_sort_kernel<PPT>));
  // fewer occupied cells than 0.6 x what a uniform cloud of N points leaves non-empty: not a cloud for cell lists (cells[4106])
  const int occ_min = (int)(0.6 * 4096.0 * (1.0 - exp(-(double)N / 4096.0)));
  hipLaunchKernelGGL((spatial_sort_kernel<PPT>), dim3(B), dim3(kThreads), lds, s, xyz, N, npad, sorted, gbox, cells,
                     occ_min);
  return dh3d_launch_status();
}

}  // namespace

#ifdef DH3D_SORT_PROBE
DH3D_API int dh3d_sort_probe_read(long long *host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sprobe), sizeof(long long) * n) == hipSuccess ? 0 : 3;
}
#endif

static int spatial_sort_any(const float *xyz, int B, int N, float *sorted, float *gbox, int *cells, void *stream) {
  DH3D_REQUIRE(xyz && sorted && gbox && B > 0 && N > 0);
  DH3D_SUPPORTED(N <= 16384);
  hipStream_t s = (hipStream_t)stream;
  float4 *so = reinterpret_cast<float4 *>(sorted);
  if (N <= 1024) return sort_launch<1>(xyz, B, N, so, gbox, cells, s);
  if (N <= 2048) return sort_launch<2>(xyz, B, N, so, gbox, cells, s);
  if (N <= 4096) return sort_launch<4>(xyz, B, N, so, gbox, cells, s);
  if (N <= 8192) return sort_launch<8>(xyz, B, N, so, gbox, cells, s);
  return sort_launch<16>(xyz, B, N, so, gbox, cells, s);
}

DH3D_API int dh3d_spatial_sort(const float *xyz, int B, int N, float *sorted, float *gbox, void *stream) {
  return spatial_sort_any(xyz, B, N, sorted, gbox, nullptr, stream);
}

DH3D_API int dh3d_spatial_sort_cells(const float *xyz, int B, int N, float *sorted, float *gbox, int32_t *cells, void *stream) {
  DH3D_REQUIRE(cells);
  return spatial_sort_any(xyz, B, N, sorted, gbox, cells, stream);
}
