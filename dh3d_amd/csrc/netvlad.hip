// Attention-weighted NetVLAD aggregation + context gating for gfx950
// (core/backbones.py:202-320, adopted there from PCAN/loupe).  The reference runs ~15 un-fused TF ops
// with many passes over the [B*N, 256] feature map.  Here:
//   K1 netvlad_assign_accumulate : per chunk of points of one cloud -- row l2-normalise into LDS,
//        assignment GEMM (xn @ Wc, f32 MFMA), BN + softmax + attention weighting in LDS, then the
//        VLAD contraction  vladT[c,d] += sum_p a[p,c] * xn[p,d]  as a second MFMA GEMM whose
//        accumulators stay in registers across all tiles of the chunk.  x is read from HBM once.
//   K2 netvlad_finalize          : sum chunk partials (fixed order -> deterministic), subtract
//        a_sum * W2, intra-normalise per cluster, flatten d-major, L2-normalise.
//   K3 netvlad_hidden_splitk     : [B,16384] x [16384,256] projection, split over K so that the
//        16.8 MB weight streams from HBM exactly once across all CUs.
//   K4 netvlad_gate              : reduce split-K partials, BN, context gating, final L2-normalise.
#include "mfma_gemm.h"

namespace {

constexpr int kD = 256;   // feature size
constexpr int kCl = 64;   // clusters
constexpr int kTM = 64;   // points per tile
constexpr int kLDX = kD + 4;
constexpr int kLDA = kTM + 4;  // actT [cluster][point]

// chunks per cloud: each workgroup keeps a [64 x 256] accumulator over its chunk
static inline int netvlad_chunks(int B, int N) {
  const int tiles = (N + kTM - 1) / kTM;
  int ch = 256 / (B > 0 ? B : 1);
  if (ch < 1) ch = 1;
  if (ch > 16) ch = 16;
  if (ch > tiles) ch = tiles;
  return ch;
}

__global__ __launch_bounds__(256) void netvlad_assign_accumulate(
    const float *__restrict__ x, const float *__restrict__ att, const float *__restrict__ wc_packed,
    const float *__restrict__ bn_scale, const float *__restrict__ bn_shift, int N, int chunks,
    float *__restrict__ part_vlad /*[B][chunks][Cl][D]*/, float *__restrict__ part_asum /*[B][chunks][Cl]*/) {
  extern __shared__ __attribute__((aligned(16))) float s_mem[];
  float *s_x = s_mem;                         // [kTM][kLDX]  l2-normalised rows
  float *s_aT = s_mem + kTM * kLDX;           // [kCl][kLDA]  assignment, transposed
  float *s_asum = s_aT + kCl * kLDA;          // [kCl]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tiles = (N + kTM - 1) / kTM;
  const int t_begin = (int)((long long)tiles * chunk / chunks);
  const int t_end = (int)((long long)tiles * (chunk + 1) / chunks);
  const float *xb = x + (size_t)b * N * kD;
  const float *attb = att + (size_t)b * N;

  if (tid < kCl) s_asum[tid] = 0.f;
  // VLAD accumulators: wave -> cluster block (wave&1), feature blocks (wave>>1) + 2j, j<4
  f32x16 vacc[4];
  zero_acc<4>(vacc);

  // this thread's 64 floats of its row (4 threads per row), loaded one tile AHEAD: the next tile's HBM round trip
  // (64 KB per workgroup, all 256 workgroups at once) hides behind the two GEMMs of the current one
  const int sp = tid >> 2, sq = tid & 3;
  float4 v[16];
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    const int n = t * kTM + sp;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      v[i] = n < N ? *reinterpret_cast<const float4 *>(xb + (size_t)n * kD + (i * 4 + sq) * 4)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  if (t_begin < t_end) load_tile(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    const int p0 = t * kTM;
    __syncthreads();  // previous tile's LDS fully consumed
    // ---- row l2-normalise into LDS
    {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        ss = fmaf(v[i].x, v[i].x, ss); ss = fmaf(v[i].y, v[i].y, ss);
        ss = fmaf(v[i].z, v[i].z, ss); ss = fmaf(v[i].w, v[i].w, ss);
      }
      ss += __shfl_xor(ss, 1, 64);
      ss += __shfl_xor(ss, 2, 64);
      const float inv = rsqrtf(fmaxf(ss, 1e-12f));  // tf.nn.l2_normalize default epsilon
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float4 w = v[i];
        w.x *= inv; w.y *= inv; w.z *= inv; w.w *= inv;
        *reinterpret_cast<float4 *>(s_x + (size_t)sp * kLDX + (i * 4 + sq) * 4) = w;
      }
    }
    if (t + 1 < t_end) load_tile(t + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads here, not at their first use
    __syncthreads();
    // ---- assignment logits = xn @ Wc : [64 x 256] x [256 x 64], one 32x32 tile per wave
    {
      f32x16 acc[1];
      zero_acc<1>(acc);
      const int row0 = (wave & 1) * 32, cb = wave >> 1;
      wave_gemm_f32<1>(s_x, kLDX, row0, wc_packed, kD / 8, cb, 2, acc);
      const int col = cb * 32 + (lane & 31);
      const float sc = bn_scale[col], sh = bn_shift[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) s_aT[(size_t)col * kLDA + row0 + mfma_row(r, lane)] = fmaf(acc[0][r], sc, sh);
    }
    __syncthreads();
    // ---- softmax over the 64 clusters of each point, times attention: 4 threads per point
    {
      const int p = tid >> 2, q = tid & 3;
      const int n = p0 + p;
      float e[16];
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) { e[i] = s_aT[(size_t)(i * 4 + q) * kLDA + p]; mx = fmaxf(mx, e[i]); }
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { e[i] = expf(e[i] - mx); sum += e[i]; }
      sum += __shfl_xor(sum, 1, 64);
      sum += __shfl_xor(sum, 2, 64);
      const float w = n < N ? attb[n] / sum : 0.f;  // rows past N contribute nothing
#pragma unroll
      for (int i = 0; i < 16; ++i) s_aT[(size_t)(i * 4 + q) * kLDA + p] = e[i] * w;
    }
    __syncthreads();
    // ---- a_sum[c] += sum_p a[p,c]: 4 threads per cluster, 16 points each (one thread walking all 64 kept its wave
    //      ~500 cycles behind the others at the next barrier)
    {
      const int c = tid >> 2, part = tid & 3;
      const float4 *row = reinterpret_cast<const float4 *>(s_aT + (size_t)c * kLDA + part * 16);
      const float4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
      float s = ((r0.x + r0.y) + (r0.z + r0.w)) + ((r1.x + r1.y) + (r1.z + r1.w)) +
                (((r2.x + r2.y) + (r2.z + r2.w)) + ((r3.x + r3.y) + (r3.z + r3.w)));
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      if (part == 0) s_asum[c] += s;
    }
    // ---- vladT[c,d] += sum_p aT[c,p] * xn[p,d]: A = aT (LDS, ld kLDA), B = xn read from LDS
    {
      const int row0 = (wave & 1) * 32;  // cluster block
      const float *aptr = s_aT + (size_t)(row0 + (lane & 31)) * kLDA + 4 * (lane >> 5);
#pragma unroll 2
      for (int kb = 0; kb < kTM / 8; ++kb) {
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(aptr + kb * 8);
        const float *brow = s_x + (size_t)(kb * 8 + 4 * (lane >> 5)) * kLDX + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int d0 = ((wave >> 1) + 2 * j) * 32;
          const float b0 = brow[d0], b1 = brow[kLDX + d0], b2 = brow[2 * kLDX + d0], b3 = brow[3 * kLDX + d0];
          vacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b0, vacc[j], 0, 0, 0);
          vacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b1, vacc[j], 0, 0, 0);
          vacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b2, vacc[j], 0, 0, 0);
          vacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b3, vacc[j], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  // ---- write this chunk's partials
  float *pv = part_vlad + ((size_t)b * chunks + chunk) * kCl * kD;
  {
    const int row0 = (wave & 1) * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int d = ((wave >> 1) + 2 * j) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) pv[(size_t)(row0 + mfma_row(r, lane)) * kD + d] = vacc[j][r];
    }
  }
  if (tid < kCl) part_asum[((size_t)b * chunks + chunk) * kCl + tid] = s_asum[tid];
}

__device__ __forceinline__ float block_sum(float v, float *s_red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// grid (8, B), block 1024: workgroup (cg, b) finishes clusters 8*cg .. 8*cg+7 of cloud b; thread (g, d) owns feature
// d of clusters 8*cg + 2g, 2g+1.  (One workgroup per cloud left 224 CUs idle behind a 143 us serial tail:
// profiles/r01_c; 256 threads with 8 clusters each was 64 dependent-ish loads per lane on 4 waves per CU: 25 us.)
constexpr int kCG = 8;  // clusters per workgroup
constexpr int kCT = 2;  // clusters per thread
__global__ __launch_bounds__(1024) void netvlad_finalize(const float *__restrict__ part_vlad,
                                                        const float *__restrict__ part_asum,
                                                        const float *__restrict__ W2 /*[D][Cl]*/,
                                                        int chunks, float *__restrict__ vlad /*[B][D*Cl]*/,
                                                        float *__restrict__ tot /*[B][Cl/kCG]*/) {
  __shared__ float s_csq[16][kCT];
  __shared__ float s_red[16];
  const int b = blockIdx.y, d = threadIdx.x & 255, g = threadIdx.x >> 8, lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;  // waves 4g .. 4g+3 share a cluster pair
  const int c0 = blockIdx.x * kCG + g * kCT;
  float v[kCT];
#pragma unroll
  for (int c = 0; c < kCT; ++c) {
    float s = 0.f, as = 0.f;
    for (int ch = 0; ch < chunks; ++ch) {  // fixed order: deterministic
      s += part_vlad[(((size_t)b * chunks + ch) * kCl + c0 + c) * kD + d];
      as += part_asum[((size_t)b * chunks + ch) * kCl + c0 + c];
    }
    v[c] = s - as * W2[(size_t)d * kCl + c0 + c];  // vlad - a_sum * cluster_weights2 (backbones.py:249-256)
  }
  // intra-normalisation: per cluster over the 256 features (tf.nn.l2_normalize(vlad, 1))
#pragma unroll
  for (int c = 0; c < kCT; ++c) {
    float sq = v[c] * v[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if (lane == 0) s_csq[wave][c] = sq;
  }
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < kCT; ++c) {
    const float csq = (s_csq[4 * g][c] + s_csq[4 * g + 1][c]) + (s_csq[4 * g + 2][c] + s_csq[4 * g + 3][c]);
    v[c] *= rsqrtf(fmaxf(csq, 1e-12f));
    t = fmaf(v[c], v[c], t);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if (lane == 0) s_red[wave] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float all = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) all += s_red[w];
    tot[(size_t)b * (kCl / kCG) + blockIdx.x] = all;
  }
  float *o = vlad + (size_t)b * kD * kCl + (size_t)d * kCl + c0;  // flatten d-major: index d*Cl + c
  *reinterpret_cast<float2 *>(o) = make_float2(v[0], v[1]);
}

// netvlad_finalize fed by the commuted aggregation directly (round 4): V[b] = A'[b]^T c[b] ([64 x m] x [m x 256], m <= 1024
// coarse rows) is formed HERE instead of by a batched split-reduction GEMM launch of its own (19 us for 0.5 GFLOP + 7 us of
// finalize + a dependency gap, on the global step's critical chain).  Workgroup (b, 8 clusters), 16 waves: wave w takes the
// coarse rows of slice w, a lane four features of all eight clusters: per row one 16-byte read of c, two LDS broadcasts
// of the eight assignment values, 32 fma.  The slices meet in LDS in a fixed order (deterministic), then finalize as above.
__global__ __launch_bounds__(1024) void netvlad_assign_finalize(const float *__restrict__ apart /*[B][m][Cl]*/,
                                                               const float *__restrict__ coarse /*[B][m][D]*/,
                                                               const float *__restrict__ asum /*[B][Cl]*/,
                                                               const float *__restrict__ W2 /*[D][Cl]*/, int m, int B,
                                                               float *__restrict__ vlad /*[B][D*Cl]*/,
                                                               float *__restrict__ tot /*[B][Cl/kCG]*/) {
  extern __shared__ __attribute__((aligned(16))) float s_af[];
  float *s_a = s_af;                    // [m][8]   this workgroup's eight columns of A'
  float *s_part = s_af + (size_t)m * 8; // [8 slices][8 clusters][256]
  __shared__ float s_csq[16][kCT];
  __shared__ float s_red[16];
  // XCD-aware (round 6): workgroups go round-robin over the 8 XCDs, and as grid (8, B) the eight cluster groups of a cloud
  // landed on eight DIFFERENT L2s -- every one of them fetched the cloud's 512 KB of coarse rows from the fabric (134 MB per
  // launch for 16.8 MB of rows).  Linear id -> (XCD = id % 8 serves clouds XCD, XCD + 8, ...; the cloud's eight groups in a row).
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int b = xcd + 8 * (seq >> 3), cgi = seq & 7;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0g = cgi * kCG;
  const float *ab = apart + (size_t)b * m * kCl, *cb = coarse + (size_t)b * m * kD;
  for (int e = tid; e < m * 2; e += 1024) {  // rows x two float4
    const int j = e >> 1, h = e & 1;
    *reinterpret_cast<float4 *>(s_a + (size_t)j * 8 + h * 4) = *reinterpret_cast<const float4 *>(ab + (size_t)j * kCl + c0g + h * 4);
  }
  __syncthreads();
  {
    // 16 row slices, one per wave, all eight clusters per lane (32 accumulators): a slice is m / 16 rows = four batches of
    // eight 16-byte reads in flight.  The 16 partial tiles meet in LDS in two rounds (waves 8-15 store, waves 0-7 add
    // theirs on top), so that the buffer stays at 8 tiles.  (Round 6: the first batch requested ahead of the staging and
    // the batches double-buffered by hand -- 8 + 8 rows, 16 + 16 do not fit the 128 VGPRs of a 1024-thread workgroup --
    // compiled to 172 bytes of scratch per lane and 65 us against 17.7.)
    const int d4 = lane * 4;
    const int per = (m + 15) >> 4, j0 = wave * per, j1 = min(m, j0 + per);
    float4 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int j = j0; j < j1; ++j) {
      const float4 x = *reinterpret_cast<const float4 *>(cb + (size_t)j * kD + d4);
      const float4 a0 = *reinterpret_cast<const float4 *>(s_a + (size_t)j * 8), a1 = *reinterpret_cast<const float4 *>(s_a + (size_t)j * 8 + 4);
      const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        acc[c].x = fmaf(av[c], x.x, acc[c].x); acc[c].y = fmaf(av[c], x.y, acc[c].y);
        acc[c].z = fmaf(av[c], x.z, acc[c].z); acc[c].w = fmaf(av[c], x.w, acc[c].w);
      }
    }
    float *slot = s_part + (size_t)(wave & 7) * 8 * kD + d4;
    if (wave >= 8) {
#pragma unroll
      for (int c = 0; c < 8; ++c) *reinterpret_cast<float4 *>(slot + (size_t)c * kD) = acc[c];
    }
    __syncthreads();
    if (wave < 8) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float4 o = *reinterpret_cast<const float4 *>(slot + (size_t)c * kD);
        o.x += acc[c].x; o.y += acc[c].y; o.z += acc[c].z; o.w += acc[c].w;
        *reinterpret_cast<float4 *>(slot + (size_t)c * kD) = o;
      }
    }
  }
  __syncthreads();
  const int d = tid & 255, g = tid >> 8;
  float v[kCT];
#pragma unroll
  for (int c = 0; c < kCT; ++c) {
    const int cl = g * kCT + c;
    float sacc = 0.f;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) sacc += s_part[((size_t)sl * 8 + cl) * kD + d];  // fixed order
    v[c] = sacc - asum[(size_t)b * kCl + c0g + cl] * W2[(size_t)d * kCl + c0g + cl];  // backbones.py:249-256
  }
#pragma unroll
  for (int c = 0; c < kCT; ++c) {
    float sq = v[c] * v[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if (lane == 0) s_csq[wave][c] = sq;
  }
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int c = 0; c < kCT; ++c) {
    const float csq = (s_csq[4 * g][c] + s_csq[4 * g + 1][c]) + (s_csq[4 * g + 2][c] + s_csq[4 * g + 3][c]);
    v[c] *= rsqrtf(fmaxf(csq, 1e-12f));
    t = fmaf(v[c], v[c], t);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if (lane == 0) s_red[wave] = t;
  __syncthreads();
  if (tid == 0) {
    float all = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) all += s_red[w];
    tot[(size_t)b * (kCl / kCG) + cgi] = all;
  }
  float *o = vlad + (size_t)b * kD * kCl + (size_t)d * kCl + c0g + g * kCT;  // flatten d-major: index d*Cl + c
  *reinterpret_cast<float2 *>(o) = make_float2(v[0], v[1]);
}


// whole-vector L2 normalisation (backbones.py:261): grid (8, B); scale = rsqrt(max(sum of the 8 partials, eps))
__global__ __launch_bounds__(256) void netvlad_l2scale(const float *__restrict__ tot, float *__restrict__ vlad) {
  const int b = blockIdx.y;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kCl / kCG; ++i) s += tot[(size_t)b * (kCl / kCG) + i];
  const float inv = rsqrtf(fmaxf(s, 1e-12f));
  float4 *p = reinterpret_cast<float4 *>(vlad + (size_t)b * kD * kCl) + (size_t)blockIdx.x * 512 + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float4 x = p[i * 256];
    x.x *= inv; x.y *= inv; x.z *= inv; x.w *= inv;
    p[i * 256] = x;
  }
}

// Split-K projection on the f32 MFMA pipe: grid (KS, ceil(B/32), O/64); a workgroup takes 32 rows x a 256-deep k
// slice x 64 outputs, its four waves = (32-column block) x (k half), so 4*KS*O/64 waves stream the 16.8 MB weight
// from HBM exactly once between them (the first version put a whole slice x all 256 outputs on one workgroup:
// 64 workgroups, 61 us of mostly exposed load latency).  A = vlad slice staged in LDS; B = Wh rows read in place
// (lanes 0..31 of a fragment are 32 consecutive outputs of one row), one k-block ahead of the MFMAs.
constexpr int kKSlice = 256;
__global__ __launch_bounds__(256) void netvlad_hidden_splitk(const float *__restrict__ vlad,
                                                            const float *__restrict__ Wh, int B, int Kd,
                                                            int O, float *__restrict__ part /*[2*KS][B][O]*/) {
  __shared__ __attribute__((aligned(16))) float s_v[32 * (kKSlice + 4)];
  constexpr int LD = kKSlice + 4;
  const int ks = blockIdx.x, b0 = blockIdx.y * 32;
  const int k0 = ks * kKSlice;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb = min(32, B - b0);
  for (int e = tid; e < 32 * (kKSlice / 4); e += 256) {
    const int bb = e / (kKSlice / 4), k4 = (e % (kKSlice / 4)) * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bb < nb && k0 + k4 < Kd) x = *reinterpret_cast<const float4 *>(vlad + (size_t)(b0 + bb) * Kd + k0 + k4);
    *reinterpret_cast<float4 *>(s_v + (size_t)bb * LD + k4) = x;
  }
  __syncthreads();
  const int kh = wave >> 1;                                  // k half of the slice
  const int o0 = blockIdx.z * 64 + (wave & 1) * 32;          // 32-column block
  const int kbeg = kh * (kKSlice / 2);
  int klen = min(kKSlice, Kd - k0) - kbeg;
  klen = klen < 0 ? 0 : (klen > kKSlice / 2 ? kKSlice / 2 : klen);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float *aptr = s_v + (size_t)(lane & 31) * LD + kbeg + 4 * (lane >> 5);
  const float *wbase = Wh + (size_t)(k0 + kbeg + 4 * (lane >> 5)) * O + o0 + (lane & 31);
  const int nkb = klen / 8;
  // the wave's whole weight panel (16 k-blocks x 4 rows) is requested before the first MFMA: the matrix mostly sits in
  // the 256 MB cache between steps, so this kernel is a chain of load latencies, not of HBM bandwidth (one block of
  // prefetch: 16 us; all in flight: see DESIGN.md)
  constexpr int NKB = kKSlice / 2 / 8;
  float w[NKB][4];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int t = 0; t < 4; ++t) w[kb][t] = kb < nkb ? wbase[(size_t)(kb * 8 + t) * O] : 0.f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    if (kb < nkb) {  // wave-uniform
      const f32x4 a4 = *reinterpret_cast<const f32x4 *>(aptr + kb * 8);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], w[kb][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], w[kb][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], w[kb][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], w[kb][3], acc, 0, 0, 0);
    }
  }
  const int o = o0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int bb = mfma_row(r, lane);
    if (bb < nb) part[((size_t)(ks * 2 + kh) * B + b0 + bb) * O + o] = acc[r];
  }
}

// grid (B), block 1024 (O == 256): reduce split-K, BN, gating, optional final l2-normalise.  Thread (q, o): quarter q
// of the partial sums / of the gating GEMV for output o, combined through LDS in a fixed order (deterministic).  (256
// threads walking all partials and all 256 weight rows one after the other: 27 us of load latency on 32 workgroups.)
__global__ __launch_bounds__(1024) void netvlad_gate(const float *__restrict__ part, int KS, int B, int O,
                                                    const float *__restrict__ bn1_scale,
                                                    const float *__restrict__ bn1_shift,
                                                    const float *__restrict__ Wg,
                                                    const float *__restrict__ bn2_scale,
                                                    const float *__restrict__ bn2_shift, float l2_eps,
                                                    const float *__restrict__ tot, float *__restrict__ out) {
  // tot (may be NULL): the 8 partial square sums of cloud b's intra-normalised VLAD vector, when the caller skipped the
  // whole-vector L2 normalisation kernel (backbones.py:261): (v * inv) @ Wh == inv * (v @ Wh), applied here
  __shared__ float s_q[4][256];
  __shared__ float s_h[256];
  __shared__ float s_red[4];
  const int b = blockIdx.x, o = threadIdx.x & 255, q = threadIdx.x >> 8;
  // the quarter's 64 gating weights are requested ahead of everything else (nothing here depends on them; behind the two
  // barriers below they were four more exposed round trips)
  const int j0 = q * (256 / 4);
  float wv0[32], wv1[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) wv0[i] = Wg ? Wg[(size_t)(j0 + i) * O + o] : 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) wv1[i] = Wg ? Wg[(size_t)(j0 + 32 + i) * O + o] : 0.f;
  {
    const int per = (KS + 3) / 4, k0 = q * per, k1 = min(KS, k0 + per);
    // sixteen partials requested before the first add (two per iteration were 16 dependent-looking round trips of the 32
    // a thread sums at KS = 128: 11.0 us of a launch whose work is a 4 MB read)
    float h0 = 0.f, h1 = 0.f;
    int ks = k0;
    for (; ks + 16 <= k1; ks += 16) {
      float t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] = part[((size_t)(ks + i) * B + b) * O + o];
#pragma unroll
      for (int i = 0; i < 16; i += 2) { h0 += t[i]; h1 += t[i + 1]; }
    }
    for (; ks < k1; ++ks) h0 += part[((size_t)ks * B + b) * O + o];
    s_q[q][o] = h0 + h1;
  }
  __syncthreads();
  float h = (s_q[0][o] + s_q[1][o]) + (s_q[2][o] + s_q[3][o]);
  if (tot) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < kCl / kCG; ++i) t += tot[(size_t)b * (kCl / kCG) + i];
    h *= rsqrtf(fmaxf(t, 1e-12f));
  }
  h = fmaf(h, bn1_scale[o], bn1_shift[o]);
  if (q == 0) s_h[o] = h;
  __syncthreads();
  float v = h;
  if (Wg) {  // context gating (backbones.py:276-277,282-320); Wg == nullptr: gating=False
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i += 4) {
      g0 = fmaf(s_h[j0 + i], wv0[i], g0);
      g1 = fmaf(s_h[j0 + i + 1], wv0[i + 1], g1);
      g2 = fmaf(s_h[j0 + i + 2], wv0[i + 2], g2);
      g3 = fmaf(s_h[j0 + i + 3], wv0[i + 3], g3);
    }
#pragma unroll
    for (int i = 0; i < 32; i += 4) {
      g0 = fmaf(s_h[j0 + 32 + i], wv1[i], g0);
      g1 = fmaf(s_h[j0 + 32 + i + 1], wv1[i + 1], g1);
      g2 = fmaf(s_h[j0 + 32 + i + 2], wv1[i + 2], g2);
      g3 = fmaf(s_h[j0 + 32 + i + 3], wv1[i + 3], g3);
    }
    __syncthreads();  // s_q is reused
    s_q[q][o] = (g0 + g1) + (g2 + g3);
    __syncthreads();
    float g = (s_q[0][o] + s_q[1][o]) + (s_q[2][o] + s_q[3][o]);
    g = fmaf(g, bn2_scale[o], bn2_shift[o]);
    v = h * (1.f / (1.f + expf(-g)));  // every quarter holds the same value
  }
  if (l2_eps > 0.f) {
    float sq = q == 0 ? v * v : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
    if (q == 0 && (o & 63) == 0) s_red[o >> 6] = sq;
    __syncthreads();
    const float all = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    v *= rsqrtf(fmaxf(all, l2_eps));
  }
  if (q != 0) return;
  out[(size_t)b * O + o] = v;
}

}  // namespace

DH3D_API size_t dh3d_netvlad_workspace_bytes(int B, int N, int D, int Cl) {
  if (B <= 0 || N <= 0 || D != kD || Cl != kCl) return 0;
  const int ch = netvlad_chunks(B, N);
  return sizeof(float) * ((size_t)B * ch * kCl * kD + (size_t)B * ch * kCl + (size_t)B * (kCl / kCG));
}

DH3D_API int dh3d_netvlad_aggregate_fwd(const float *x, const float *att, const float *wc_packed,
                                        const float *bn_scale, const float *bn_shift, const float *W2,
                                        int B, int N, int D, int Cl, void *workspace,
                                        size_t workspace_bytes, float *vlad, void *stream) {
  DH3D_REQUIRE(x && att && wc_packed && bn_scale && bn_shift && W2 && workspace && vlad && B > 0 && N > 0);
  DH3D_SUPPORTED(D == kD && Cl == kCl && B <= 65535);
  DH3D_REQUIRE(workspace_bytes >= dh3d_netvlad_workspace_bytes(B, N, D, Cl));
  const int ch = netvlad_chunks(B, N);
  float *part_vlad = static_cast<float *>(workspace);
  float *part_asum = part_vlad + (size_t)B * ch * kCl * kD;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = sizeof(float) * (kTM * kLDX + kCl * kLDA + kCl);
  auto kern = netvlad_assign_accumulate;
  DH3D_ALLOW_BIG_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(ch, B), dim3(256), lds, s, x, att, wc_packed, bn_scale, bn_shift, N, ch,
                     part_vlad, part_asum);
  float *tot = part_asum + (size_t)B * ch * kCl;
  hipLaunchKernelGGL(netvlad_finalize, dim3(kCl / kCG, B), dim3(1024), 0, s, part_vlad, part_asum, W2, ch, vlad, tot);
  hipLaunchKernelGGL(netvlad_l2scale, dim3(kCl / kCG, B), dim3(256), 0, s, tot, vlad);
  return dh3d_launch_status();
}

DH3D_API size_t dh3d_netvlad_head_workspace_bytes(int B, int Kd, int O) {
  if (B <= 0 || Kd <= 0 || O != 256) return 0;
  return sizeof(float) * (size_t)2 * dh3d_cdiv(Kd, kKSlice) * B * O;  // two k-half partials per slice
}

DH3D_API int dh3d_netvlad_head_fwd(const float *vlad, const float *Wh, const float *bn1_scale,
                                   const float *bn1_shift, const float *Wg, const float *bn2_scale,
                                   const float *bn2_shift, int B, int Kd, int O, float l2_eps,
                                   void *workspace, size_t workspace_bytes, float *out, void *stream) {
  DH3D_REQUIRE(vlad && Wh && bn1_scale && bn1_shift && workspace && out);
  DH3D_REQUIRE(!Wg || (bn2_scale && bn2_shift));  // Wg == NULL: no context gating (gating=False, backbones.py:276)
  DH3D_REQUIRE(B > 0 && Kd > 0);
  DH3D_SUPPORTED(O == 256 && B <= 65535);
  DH3D_REQUIRE(workspace_bytes >= dh3d_netvlad_head_workspace_bytes(B, Kd, O));
  const int KS = dh3d_cdiv(Kd, kKSlice);
  float *part = static_cast<float *>(workspace);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(netvlad_hidden_splitk, dim3(KS, dh3d_cdiv(B, 32), O / 64), dim3(256), 0, s, vlad, Wh, B, Kd, O,
                     part);
  hipLaunchKernelGGL(netvlad_gate, dim3(B), dim3(1024), 0, s, part, 2 * KS, B, O, bn1_scale, bn1_shift, Wg,
                     bn2_scale, bn2_shift, l2_eps, static_cast<const float *>(nullptr), out);
  return dh3d_launch_status();
}

// The tail behind the commuted aggregation (dense_x6.hip, dh3d_global_tail_fwd): V [B, Cl, D] = A'^T coarse per cloud and
// asum [B, Cl] are given; subtract asum * W2, intra-normalise, flatten, project, gate.  workspace:
// dh3d_netvlad_tail_workspace_bytes.
DH3D_API size_t dh3d_netvlad_tail_workspace_bytes(int B, int D, int Cl, int O) {
  const size_t h = dh3d_netvlad_head_workspace_bytes(B, D * Cl, O);
  if (!h || D != kD || Cl != kCl) return 0;
  return ((h + 255) & ~(size_t)255) + sizeof(float) * ((size_t)B * D * Cl + (size_t)B * (kCl / kCG));
}

DH3D_API int dh3d_netvlad_tail_fwd(const float *V, const float *asum, const float *W2, const float *Wh,
                                   const float *bn1_scale, const float *bn1_shift, const float *Wg,
                                   const float *bn2_scale, const float *bn2_shift, int B, int D, int Cl, int O,
                                   float l2_eps, void *workspace, size_t workspace_bytes, float *out, void *stream) {
  DH3D_REQUIRE(V && asum && W2 && Wh && bn1_scale && bn1_shift && workspace && out && B > 0);
  DH3D_REQUIRE(!Wg || (bn2_scale && bn2_shift));
  DH3D_SUPPORTED(D == kD && Cl == kCl && O == 256 && B <= 65535);
  DH3D_REQUIRE(workspace_bytes >= dh3d_netvlad_tail_workspace_bytes(B, D, Cl, O));
  const size_t hb = (dh3d_netvlad_head_workspace_bytes(B, D * Cl, O) + 255) & ~(size_t)255;
  char *w = static_cast<char *>(workspace);
  float *part = reinterpret_cast<float *>(w);
  float *vlad = reinterpret_cast<float *>(w + hb);
  float *tot = vlad + (size_t)B * D * Cl;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(netvlad_finalize, dim3(kCl / kCG, B), dim3(1024), 0, s, V, asum, W2, 1, vlad, tot);
  const int Kd = D * Cl, KS = dh3d_cdiv(Kd, kKSlice);
  hipLaunchKernelGGL(netvlad_hidden_splitk, dim3(KS, dh3d_cdiv(B, 32), O / 64), dim3(256), 0, s, vlad, Wh, B, Kd, O,
                     part);
  hipLaunchKernelGGL(netvlad_gate, dim3(B), dim3(1024), 0, s, part, 2 * KS, B, O, bn1_scale, bn1_shift, Wg, bn2_scale,
                     bn2_shift, l2_eps, static_cast<const float *>(tot), out);
  return dh3d_launch_status();
}

// ... and the same tail fed by the assignment itself: apart [B, m, Cl] = A' (the walk's output), coarse [B, m, D], asum [B, Cl];
// V = A'^T coarse is formed inside the finalize kernel (netvlad_assign_finalize).  m <= 1024.
DH3D_API int dh3d_netvlad_tail_assign_fwd(const float *apart, const float *coarse, const float *asum, int m, const float *W2,
                                          const float *Wh, const float *bn1_scale, const float *bn1_shift, const float *Wg,
                                          const float *bn2_scale, const float *bn2_shift, int B, int D, int Cl, int O,
                                          float l2_eps, void *workspace, size_t workspace_bytes, float *out, void *stream) {
  DH3D_REQUIRE(apart && coarse && asum && W2 && Wh && bn1_scale && bn1_shift && workspace && out && B > 0 && m > 0);
  DH3D_REQUIRE(!Wg || (bn2_scale && bn2_shift));
  DH3D_SUPPORTED(D == kD && Cl == kCl && O == 256 && B <= 65535 && m <= 1024);
  DH3D_REQUIRE(workspace_bytes >= dh3d_netvlad_tail_workspace_bytes(B, D, Cl, O));
  const size_t hb = (dh3d_netvlad_head_workspace_bytes(B, D * Cl, O) + 255) & ~(size_t)255;
  char *w = static_cast<char *>(workspace);
  float *part = reinterpret_cast<float *>(w);
  float *vlad = reinterpret_cast<float *>(w + hb);
  float *tot = vlad + (size_t)B * D * Cl;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = sizeof(float) * ((size_t)m * 8 + 8 * 8 * kD);
  DH3D_ALLOW_BIG_LDS(netvlad_assign_finalize);
  hipLaunchKernelGGL(netvlad_assign_finalize, dim3(8 * dh3d_cdiv(B, 8) * (kCl / kCG)), dim3(1024), lds, s, apart, coarse, asum, W2, m, B, vlad, tot);
  const int Kd = D * Cl, KS = dh3d_cdiv(Kd, kKSlice);
  hipLaunchKernelGGL(netvlad_hidden_splitk, dim3(KS, dh3d_cdiv(B, 32), O / 64), dim3(256), 0, s, vlad, Wh, B, Kd, O,
                     part);
  hipLaunchKernelGGL(netvlad_gate, dim3(B), dim3(1024), 0, s, part, 2 * KS, B, O, bn1_scale, bn1_shift, Wg, bn2_scale,
                     bn2_shift, l2_eps, static_cast<const float *>(tot), out);
  return dh3d_launch_status();
}

// Aggregation + projection + gating in one call (the model path): same result as dh3d_netvlad_aggregate_fwd followed by
// dh3d_netvlad_head_fwd up to the rounding of one multiplication per output -- the whole-vector L2 normalisation is not
// a kernel of its own (a pass over [B, 16384] + a dependency gap), its factor multiplies the projected vector instead.
// workspace: dh3d_netvlad_workspace_bytes + dh3d_netvlad_head_workspace_bytes + 4*B*D*Cl bytes.
DH3D_API size_t dh3d_netvlad_fused_workspace_bytes(int B, int N, int D, int Cl, int O) {
  const size_t a = dh3d_netvlad_workspace_bytes(B, N, D, Cl), h = dh3d_netvlad_head_workspace_bytes(B, D * Cl, O);
  if (!a || !h) return 0;
  return ((a + 255) & ~(size_t)255) + ((h + 255) & ~(size_t)255) + sizeof(float) * (size_t)B * D * Cl;
}

DH3D_API int dh3d_netvlad_fused_fwd(const float *x, const float *att, const float *wc_packed, const float *bn_scale,
                                    const float *bn_shift, const float *W2, const float *Wh, const float *bn1_scale,
                                    const float *bn1_shift, const float *Wg, const float *bn2_scale,
                                    const float *bn2_shift, int B, int N, int D, int Cl, int O, float l2_eps,
                                    void *workspace, size_t workspace_bytes, float *out, void *stream) {
  DH3D_REQUIRE(x && att && wc_packed && bn_scale && bn_shift && W2 && Wh && bn1_scale && bn1_shift && workspace && out);
  DH3D_REQUIRE(B > 0 && N > 0 && (!Wg || (bn2_scale && bn2_shift)));
  DH3D_SUPPORTED(D == kD && Cl == kCl && O == 256 && B <= 65535);
  DH3D_REQUIRE(workspace_bytes >= dh3d_netvlad_fused_workspace_bytes(B, N, D, Cl, O));
  const size_t a = (dh3d_netvlad_workspace_bytes(B, N, D, Cl) + 255) & ~(size_t)255;
  const size_t hb = (dh3d_netvlad_head_workspace_bytes(B, D * Cl, O) + 255) & ~(size_t)255;
  char *w = static_cast<char *>(workspace);
  float *vlad = reinterpret_cast<float *>(w + a + hb);
  const int ch = netvlad_chunks(B, N);
  float *part_vlad = reinterpret_cast<float *>(w);
  float *part_asum = part_vlad + (size_t)B * ch * kCl * kD;
  float *tot = part_asum + (size_t)B * ch * kCl;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = sizeof(float) * (kTM * kLDX + kCl * kLDA + kCl);
  auto kern = netvlad_assign_accumulate;
  DH3D_ALLOW_BIG_LDS(kern);
  hipLaunchKernelGGL(kern, dim3(ch, B), dim3(256), lds, s, x, att, wc_packed, bn_scale, bn_shift, N, ch, part_vlad,
                     part_asum);
  hipLaunchKernelGGL(netvlad_finalize, dim3(kCl / kCG, B), dim3(1024), 0, s, part_vlad, part_asum, W2, ch, vlad, tot);
  const int Kd = D * Cl, KS = dh3d_cdiv(Kd, kKSlice);
  float *part = reinterpret_cast<float *>(w + a);
  hipLaunchKernelGGL(netvlad_hidden_splitk, dim3(KS, dh3d_cdiv(B, 32), O / 64), dim3(256), 0, s, vlad, Wh, B, Kd, O,
                     part);
  hipLaunchKernelGGL(netvlad_gate, dim3(B), dim3(1024), 0, s, part, 2 * KS, B, O, bn1_scale, bn1_shift, Wg, bn2_scale,
                     bn2_shift, l2_eps, static_cast<const float *>(tot), out);
  return dh3d_launch_status();
}
