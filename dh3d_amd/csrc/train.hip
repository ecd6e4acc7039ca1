// Training-mode BatchNorm and the attention head's element-wise passes for the Siamese training step (gfx950).
//
// The reference trains with batch statistics (tensorpack BatchNorm / slim batch_norm with is_training,
// core/tf_utils.py:60-63, core/backbones.py:145-173,218-223,271-274): TF runs one fused_batch_norm / moments +
// batchnorm pair per site, forward and backward.  Here every site is two HBM-bound passes per direction:
//   forward   column statistics (sum, sum of squares; f32 per thread, f64 hardware atomics across workgroups, so that
//             var = E[x^2] - E[x]^2 is formed from exact-ish sums)  ->  [host: mean / rstd / folded scale, shift; under
//             sync-BN the all-reduce of (sum, sumsq, count) sits here]  ->  y = act(x * scale + shift)
//   backward  S1 = sum dz, S2 = sum dz * xhat  (dz = dy through the ReLU mask, recomputed from x)  ->  [all-reduce under
//             sync-BN]  ->  dx = gamma * rstd * (dz - S1/n - xhat * S2/n);   dgamma = S2, dbeta = S1.
// The attention head (core/backbones.py:156-173: 256 -> 1024 BNReLU -> 1, sigmoid) never materialises its
// [R, 1024] activation gradient before the BatchNorm: dy[n,c] = dlogit[n] * w_fc[c] is rank one, so both backward passes
// take (dlogit, w_fc) instead of dy, the first one also accumulates the gradient of w_fc, and the second one overwrites
// the saved pre-activation in place with its gradient.
// Rows of padding clouds (sharded batches, dh3d_amd/dist.py) are excluded through a per-cloud mask.
#include "common.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

__device__ __forceinline__ bool row_live(const unsigned char *mask, long long r, int rows_per_cloud) {
  return !mask || mask[r / rows_per_cloud] != 0;
}

// grid (ceil(C/64), chunks); 256 threads = 4 row lanes x 64 columns
__global__ __launch_bounds__(256) void colstats_kernel(const float *__restrict__ x, long long R, int C, int rows_per,
                                                      const unsigned char *__restrict__ mask, int rows_per_cloud,
                                                      double *__restrict__ sum, double *__restrict__ sumsq) {
  __shared__ float s1[4][64], s2[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
  float a = 0.f, b = 0.f;
  if (c < C)
    for (long long r = r0 + q; r < r1; r += 4)
      if (row_live(mask, r, rows_per_cloud)) { const float v = x[r * C + c]; a += v; b = fmaf(v, v, b); }
  s1[q][threadIdx.x & 63] = a; s2[q][threadIdx.x & 63] = b;
  __syncthreads();
  if (q == 0 && c < C) {
    const int t = threadIdx.x;
    unsafeAtomicAdd(sum + c, ((double)s1[0][t] + s1[1][t]) + ((double)s1[2][t] + s1[3][t]));
    unsafeAtomicAdd(sumsq + c, ((double)s2[0][t] + s2[1][t]) + ((double)s2[2][t] + s2[3][t]));
  }
}

// y = act(x * scale[c] + shift[c]), float4 per lane
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const float *__restrict__ x, long long R, int C,
                                                             const float *__restrict__ scale,
                                                             const float *__restrict__ shift, int relu,
                                                             const float *__restrict__ residual, float *__restrict__ y) {
  const int cv = C / 4;
  const long long total = R * cv;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c4 = (int)(e % cv) * 4;
    const float4 v = reinterpret_cast<const float4 *>(x)[e];
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c4), sh = *reinterpret_cast<const float4 *>(shift + c4);
    float4 o;
    o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y); o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    if (residual) {  // added after the activation (the shortcut sum of core/backbones.py:123)
      const float4 r = reinterpret_cast<const float4 *>(residual)[e];
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    reinterpret_cast<float4 *>(y)[e] = o;
  }
}

// att[n] = sigmoid(sum_c relu(h[n,c] * scale[c] + shift[c]) * w[c] + b): one wave per row
__global__ __launch_bounds__(256) void row_logit_kernel(const float *__restrict__ h, long long R, int C,
                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                       const float *__restrict__ w, const float *__restrict__ bptr,
                                                       float *__restrict__ att) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float acc = 0.f;
  for (int c4 = lane * 4; c4 < C; c4 += 256) {
    const float4 v = *reinterpret_cast<const float4 *>(h + row * C + c4);
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c4), sh = *reinterpret_cast<const float4 *>(shift + c4);
    const float4 ww = *reinterpret_cast<const float4 *>(w + c4);
    acc = fmaf(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), ww.x, acc);
    acc = fmaf(fmaxf(fmaf(v.y, sc.y, sh.y), 0.f), ww.y, acc);
    acc = fmaf(fmaxf(fmaf(v.z, sc.z, sh.z), 0.f), ww.z, acc);
    acc = fmaf(fmaxf(fmaf(v.w, sc.w, sh.w), 0.f), ww.w, acc);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) att[row] = 1.f / (1.f + expf(-(acc + bptr[0])));
}

// backward sums.  dy == nullptr: dy[n,c] = rowscale[n] * colvec[c] (attention head) and S3[c] += rowscale[n] * y[n,c].
// xhat = (x - mean) * rstd;  y = xhat * gamma + beta;  dz = relu ? dy * [y > 0] : dy
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                         const float *__restrict__ rowscale,
                                                         const float *__restrict__ colvec, long long R, int C,
                                                         int rows_per, const float *__restrict__ mean,
                                                         const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, int relu,
                                                         const unsigned char *__restrict__ mask, int rows_per_cloud,
                                                         double *__restrict__ S1, double *__restrict__ S2,
                                                         double *__restrict__ S3) {
  __shared__ float s[3][4][64];
  const int t = threadIdx.x & 63, c = blockIdx.x * 64 + t, q = threadIdx.x >> 6;
  const long long r0 = (long long)blockIdx.y * rows_per, r1 = min(R, r0 + rows_per);
  float a = 0.f, b = 0.f, d = 0.f;
  if (c < C) {
    const float mu = mean[c], rs = rstd[c], g = gamma[c], be = beta[c], cvv = colvec ? colvec[c] : 0.f;
    for (long long r = r0 + q; r < r1; r += 4) {
      if (!row_live(mask, r, rows_per_cloud)) continue;
      const float xh = (x[r * C + c] - mu) * rs;
      const float y = fmaf(xh, g, be);
      float dz = dy ? dy[r * C + c] : rowscale[r] * cvv;
      if (relu && !(y > 0.f)) dz = 0.f;
      a += dz; b = fmaf(dz, xh, b);
      if (!dy) d = fmaf(rowscale[r], relu ? fmaxf(y, 0.f) : y, d);
    }
  }
  s[0][q][t] = a; s[1][q][t] = b; s[2][q][t] = d;
  __syncthreads();
  if (q == 0 && c < C) {
    unsafeAtomicAdd(S1 + c, ((double)s[0][0][t] + s[0][1][t]) + ((double)s[0][2][t] + s[0][3][t]));
    unsafeAtomicAdd(S2 + c, ((double)s[1][0][t] + s[1][1][t]) + ((double)s[1][2][t] + s[1][3][t]));
    if (S3) unsafeAtomicAdd(S3 + c, ((double)s[2][0][t] + s[2][1][t]) + ((double)s[2][2][t] + s[2][3][t]));
  }
}

// dx = gamma*rstd*(dz - S1/n - xhat*S2/n) with everything per-column folded by bn_bwd_finalize:
//   dx = k1[c]*dz - k2[c] - k3[c]*x,   dz = relu ? dy*[x*scale[c] + shift[c] > 0] : dy
// (k1 = scale = gamma*rstd, k3 = gamma*rstd^2*S2/n, k2 = gamma*rstd*S1/n - k3*mean).  dx may alias x or dy (in place).
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *x, const float *dy,
                                                          const float *__restrict__ rowscale,
                                                          const float *__restrict__ colvec, long long R, int C,
                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                          const float *__restrict__ k2, const float *__restrict__ k3,
                                                          int relu, const unsigned char *__restrict__ mask,
                                                          int rows_per_cloud, float *dx) {
  const int cv = C / 4;
  // one workgroup walks whole rows: the column of a lane is fixed, its coefficients live in registers
  const int c4 = (threadIdx.x % cv) * 4;
  const int rpb = 256 / cv > 0 ? 256 / cv : 1;      // rows per block pass (cv <= 256)
  const int rl = threadIdx.x / cv;
  if (rl >= rpb) return;
  const float4 sc = *reinterpret_cast<const float4 *>(scale + c4), sh = *reinterpret_cast<const float4 *>(shift + c4);
  const float4 q2 = *reinterpret_cast<const float4 *>(k2 + c4), q3 = *reinterpret_cast<const float4 *>(k3 + c4);
  float4 cvv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!dy) cvv = *reinterpret_cast<const float4 *>(colvec + c4);
  for (long long r = (long long)blockIdx.x * rpb + rl; r < R; r += (long long)gridDim.x * rpb) {
    const size_t e = (size_t)r * cv + c4 / 4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_live(mask, r, rows_per_cloud)) {
      const float4 xv = reinterpret_cast<const float4 *>(x)[e];
      float4 dv;
      if (dy) dv = reinterpret_cast<const float4 *>(dy)[e];
      else { const float rsn = rowscale[r]; dv = make_float4(rsn * cvv.x, rsn * cvv.y, rsn * cvv.z, rsn * cvv.w); }
      if (relu) {
        if (!(fmaf(xv.x, sc.x, sh.x) > 0.f)) dv.x = 0.f;
        if (!(fmaf(xv.y, sc.y, sh.y) > 0.f)) dv.y = 0.f;
        if (!(fmaf(xv.z, sc.z, sh.z) > 0.f)) dv.z = 0.f;
        if (!(fmaf(xv.w, sc.w, sh.w) > 0.f)) dv.w = 0.f;
      }
      o.x = fmaf(sc.x, dv.x, -q2.x) - q3.x * xv.x; o.y = fmaf(sc.y, dv.y, -q2.y) - q3.y * xv.y;
      o.z = fmaf(sc.z, dv.z, -q2.z) - q3.z * xv.z; o.w = fmaf(sc.w, dv.w, -q2.w) - q3.w * xv.w;
    }
    reinterpret_cast<float4 *>(dx)[e] = o;
  }
}

// Training-mode BatchNorm of a SHORT tensor (R <= 64 rows: the [clouds, 256] hidden / gating activations of NetVLAD) in
// one launch per direction instead of fill + statistics + finalize + apply: a workgroup owns 64 channels, thread (c, q)
// walks rows q, q + 4, ...  Same arithmetic as the long form: biased variance from f64 sums, folded scale / shift,
// running buffers with decay `momentum` (variance Bessel-corrected if `unbiased`: what tf.nn.fused_batch_norm feeds the
// moving average); dx = scale dz - k2 - k3 x.  mask [R] bytes (rows_per_cloud == 1) or NULL.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_small_kernel(const float *__restrict__ x, const float *__restrict__ dy, int R,
                                                      int C, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float eps, float momentum,
                                                      int unbiased, int relu, const unsigned char *__restrict__ mask,
                                                      float *run_mean, float *run_var, float *__restrict__ stats,
                                                      float *__restrict__ out, float *__restrict__ dgamma,
                                                      float *__restrict__ dbeta) {
  __shared__ double s_a[4][64], s_b[4][64];
  __shared__ int s_n;
  const int t = threadIdx.x, lc = t & 63, q = t >> 6, c = blockIdx.x * 64 + lc;
  if (t == 0) {
    int n = 0;
    for (int r = 0; r < R; ++r) n += (!mask || mask[r]) ? 1 : 0;
    s_n = n;
  }
  const bool ok = c < C;
  double a = 0.0, b = 0.0;
  if (!BWD) {
    if (ok)
      for (int r = q; r < R; r += 4)
        if (!mask || mask[r]) { const double v = x[(size_t)r * C + c]; a += v; b += v * v; }
  }
  float mu = 0.f, rs = 0.f, sc = 0.f, sh = 0.f;
  if (BWD && ok) { mu = stats[c]; rs = stats[C + c]; sc = stats[2 * C + c]; sh = stats[3 * C + c]; }
  if (BWD && ok) {
    for (int r = q; r < R; r += 4) {
      if (mask && !mask[r]) continue;
      const float xv = x[(size_t)r * C + c];
      float dz = dy[(size_t)r * C + c];
      if (relu && !(fmaf(xv, sc, sh) > 0.f)) dz = 0.f;
      a += dz; b += (double)dz * ((xv - mu) * rs);
    }
  }
  s_a[q][lc] = a; s_b[q][lc] = b;
  __syncthreads();
  const double A = (s_a[0][lc] + s_a[1][lc]) + (s_a[2][lc] + s_a[3][lc]);
  const double Bq = (s_b[0][lc] + s_b[1][lc]) + (s_b[2][lc] + s_b[3][lc]);
  const double n = s_n > 1 ? (double)s_n : 1.0;
  if (!ok) return;
  if (!BWD) {
    const double m = A / n;
    double var = Bq / n - m * m;
    var = var > 0.0 ? var : 0.0;
    rs = (float)(1.0 / sqrt(var + (double)eps));
    sc = gamma[c] * rs; mu = (float)m; sh = beta[c] - mu * sc;
    if (q == 0) {
      stats[c] = mu; stats[C + c] = rs; stats[2 * C + c] = sc; stats[3 * C + c] = sh;
      if (s_n > 0) {
        run_mean[c] = momentum * run_mean[c] + (1.f - momentum) * mu;
        const double uv = unbiased && s_n > 1 ? var * (n / (n - 1.0)) : var;
        run_var[c] = momentum * run_var[c] + (1.f - momentum) * (float)uv;
      }
    }
    for (int r = q; r < R; r += 4) {
      float v = fmaf(x[(size_t)r * C + c], sc, sh);
      if (relu) v = fmaxf(v, 0.f);
      out[(size_t)r * C + c] = v;
    }
  } else {
    const float m1 = (float)(A / n), m2 = (float)(Bq / n);
    const float k3 = gamma[c] * rs * rs * m2, k2 = gamma[c] * rs * m1 - k3 * mu;
    if (q == 0) { dbeta[c] = (float)A; dgamma[c] = (float)Bq; }
    for (int r = q; r < R; r += 4) {
      float o = 0.f;
      if (!mask || mask[r]) {
        const float xv = x[(size_t)r * C + c];
        float dz = dy[(size_t)r * C + c];
        if (relu && !(fmaf(xv, sc, sh) > 0.f)) dz = 0.f;
        o = fmaf(sc, dz, -k2) - k3 * xv;
      }
      out[(size_t)r * C + c] = o;
    }
  }
}

// One launch instead of ~15 tiny tensor ops per BatchNorm site and direction.
// forward: (sum, sumsq, count) f64 -> mean, rstd, folded scale/shift; running buffers updated with decay `momentum`
// (left alone when count == 0: a rank that holds only padding clouds); `unbiased`: the moving variance takes the
// Bessel-corrected batch variance (tf.nn.fused_batch_norm), else the biased one (slim batch_norm with fused=False).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double *__restrict__ sum, const double *__restrict__ sumsq,
                                                         const double *__restrict__ cnt, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float eps, float momentum,
                                                         int unbiased, float *__restrict__ run_mean,
                                                         float *__restrict__ run_var, int C,
                                                         float *__restrict__ mean, float *__restrict__ rstd,
                                                         float *__restrict__ scale, float *__restrict__ shift, int P) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double n = cnt[0] > 1.0 ? cnt[0] : 1.0;
  double s1 = 0.0, s2 = 0.0;  // P > 1: per-cloud partial rows (dh3d_bn_finalize_parts)
  for (int p = 0; p < P; ++p) { s1 += sum[(size_t)p * C + c]; s2 += sumsq[(size_t)p * C + c]; }
  const double mu = s1 / n;
  double var = s2 / n - mu * mu;
  var = var > 0.0 ? var : 0.0;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * rs;
  mean[c] = (float)mu; rstd[c] = rs; scale[c] = sc; shift[c] = beta[c] - (float)mu * sc;
  if (cnt[0] > 0.0) {
    run_mean[c] = momentum * run_mean[c] + (1.f - momentum) * (float)mu;
    const double uv = unbiased && n > 1.0 ? var * (n / (n - 1.0)) : var;
    run_var[c] = momentum * run_var[c] + (1.f - momentum) * (float)uv;
  }
}
// backward: (S1, S2, count) f64 (+ mean, rstd, gamma) -> k2, k3 of bn_bwd_apply
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double *__restrict__ S1, const double *__restrict__ S2,
                                                             const double *__restrict__ cnt, const float *__restrict__ mean,
                                                             const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                             int C, float *__restrict__ k2, float *__restrict__ k3) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double n = cnt[0] > 1.0 ? cnt[0] : 1.0;
  const float m1 = (float)(S1[c] / n), m2 = (float)(S2[c] / n);
  const float q3 = gamma[c] * rstd[c] * rstd[c] * m2;
  k3[c] = q3;
  k2[c] = gamma[c] * rstd[c] * m1 - q3 * mean[c];
}

// The same from per-cloud PARTIAL sums part [nk][P][C] (nk = 2: S1, S2; 3: + S3 -- what the walks of interp_train.hip /
// netvlad_train.hip leave): their sums over P, k2 / k3, and the sums themselves as float32 rows grads [nk][C] (dbeta,
// dgamma, d w_fc) -- one launch instead of a reduction, a conversion and the finalize.
__global__ __launch_bounds__(256) void bn_bwd_finalize_parts_kernel(const double *__restrict__ part, int nk, int P,
                                                                   const double *__restrict__ cnt,
                                                                   const float *__restrict__ mean,
                                                                   const float *__restrict__ rstd,
                                                                   const float *__restrict__ gamma, int C,
                                                                   float *__restrict__ k2, float *__restrict__ k3,
                                                                   float *__restrict__ grads) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double S[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < nk; ++k) {
    double a = 0.0;
    for (int p = 0; p < P; ++p) a += part[((size_t)k * P + p) * C + c];
    S[k] = a;
    grads[(size_t)k * C + c] = (float)a;
  }
  const double n = cnt[0] > 1.0 ? cnt[0] : 1.0;
  const float m1 = (float)(S[0] / n), m2 = (float)(S[1] / n);
  const float q3 = gamma[c] * rstd[c] * rstd[c] * m2;
  k3[c] = q3;
  k2[c] = gamma[c] * rstd[c] * m1 - q3 * mean[c];
}

// dlogit = datt * att * (1 - att) (sigmoid backward of the attention head; 0 on rows of padding clouds) and its sum
// (the gradient of the fc bias; `sum` zeroed by the caller): one launch instead of three element-wise passes and a
// reduction.
__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float *__restrict__ datt, const float *__restrict__ att,
                                                         const unsigned char *__restrict__ mask, int rows_per_cloud,
                                                         long long n, float *__restrict__ dlogit, float *__restrict__ sum) {
  __shared__ float s_p[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float a = att[i];
    const float d = row_live(mask, i, rows_per_cloud) ? datt[i] * a * (1.0f - a) : 0.f;
    dlogit[i] = d;
    acc += d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(sum, (s_p[0] + s_p[1]) + (s_p[2] + s_p[3]));
}

// ---------------------------------------------------------------------------------------------------------------
// NetVLAD soft assignment (core/backbones.py:214-238), one wave per row of s = xn @ Wc (64 clusters = 64 lanes):
//   z = s*scale + shift (training: folded batch statistics);  p = softmax(z);  a = p * att[n]
// backward given da = dL/da:  datt[n] = sum_c da*p;  dp = da*att;  dz = p * (dp - sum_c dp*p)
// forward: a workgroup owns RPW consecutive rows (wave w the rows w, w + 4, ...); with asum != NULL the column sums of a
// over the rows of a cloud (asum [clouds, 64], zeroed by the caller; rows_per_cloud % RPW == 0) ride along -- registers,
// one LDS exchange, 64 atomics per workgroup -- instead of a second pass over a.
template <bool BWD, int RPW>
__global__ __launch_bounds__(256) void netvlad_assign_rows_kernel(const float *__restrict__ sm, long long R,
                                                                 const float *__restrict__ scale,
                                                                 const float *__restrict__ shift,
                                                                 const float *__restrict__ att,
                                                                 const float *__restrict__ da, float *__restrict__ out,
                                                                 float *__restrict__ datt, float *__restrict__ asum,
                                                                 long long rows_per_cloud) {
  __shared__ float s_sum[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long row0 = (long long)blockIdx.x * RPW;
  const float sc = scale[lane], sh = shift[lane];
  float colsum = 0.f;
  for (int i = wave; i < RPW; i += 4) {
    const long long row = row0 + i;
    if (row >= R) break;
    const float z = fmaf(sm[row * 64 + lane], sc, sh);
    float m = z;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    const float e = expf(z - m);
    float den = e;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) den += __shfl_xor(den, off, 64);
    const float p = e / den, w = att[row];
    if (!BWD) {
      const float a = p * w;
      out[row * 64 + lane] = a;
      colsum += a;
      continue;
    }
    const float g = da[row * 64 + lane];
    float d1 = g * p;           // -> datt
    float d2 = g * w * p;       // -> sum_c dp * p
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_xor(d1, off, 64); d2 += __shfl_xor(d2, off, 64); }
    out[row * 64 + lane] = p * (g * w - d2);
    if (lane == 0) datt[row] = d1;
  }
  if (!BWD && asum) {
    s_sum[wave][lane] = colsum;
    __syncthreads();
    if (wave == 0 && row0 < R)
      unsafeAtomicAdd(asum + (row0 / rows_per_cloud) * 64 + lane,
                      (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]));
  }
}

// inverse-distance weights of three_interpolate (core/backbones.py:92-95): d = max(dist, 1e-10), w = (1/d) / sum_t (1/d)
__global__ __launch_bounds__(256) void idw_weights_kernel(const float *__restrict__ dist, long long R,
                                                         float *__restrict__ w) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  const float r0 = 1.0f / fmaxf(dist[3 * r], 1e-10f), r1 = 1.0f / fmaxf(dist[3 * r + 1], 1e-10f),
              r2 = 1.0f / fmaxf(dist[3 * r + 2], 1e-10f);
  const float tot = (r0 + r1) + r2;
  w[3 * r] = r0 / tot; w[3 * r + 1] = r1 / tot; w[3 * r + 2] = r2 / tot;
}

// context gating (core/backbones.py:271-277): y = v * sigmoid(g); backward dv = dy * sig, dg = dy * v * sig * (1 - sig)
template <bool BWD>
__global__ __launch_bounds__(256) void context_gate_kernel(const float *__restrict__ v, const float *__restrict__ g,
                                                          const float *__restrict__ dy, long long n,
                                                          float *__restrict__ o0, float *__restrict__ o1) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float sig = 1.0f / (1.0f + expf(-g[i]));
  if (!BWD) { o0[i] = v[i] * sig; return; }
  const float d = dy[i];
  o0[i] = d * sig;
  o1[i] = d * v[i] * sig * (1.0f - sig);
}

// ---------------------------------------------------------------------------------------------------------------
// NetVLAD between the VLAD contraction and the hidden projection (core/backbones.py:241-262), one workgroup per cloud:
//   u[d,c] = V[c,d] - asum[c] * W2[d,c];  y1 = u * rsqrt(max(sum_d u^2, eps))  (intra-normalisation per cluster);
//   out[d*64 + c] = y1 * rsqrt(max(sum y1^2, eps))                              (L2 over the whole 16384-vector)
// and its backward -- ~12 + ~25 tiny tensor ops of the training graph in two launches.  Thread (c = t & 63, q = t >> 6)
// holds the 64 values d = 64 q .. 64 q + 63 of cluster c in registers.  D == 256, Cl == 64.
template <bool BWD>
__global__ __launch_bounds__(256) void vlad_normalize_kernel(const float *__restrict__ V, const float *__restrict__ asum,
                                                            const float *__restrict__ W2, float eps,
                                                            float *__restrict__ out, float *__restrict__ inv_c,
                                                            float *__restrict__ inv_t, const float *__restrict__ g,
                                                            float *__restrict__ dV, float *__restrict__ dasum,
                                                            float *__restrict__ dW2) {
  __shared__ float s_red[4][64];
  __shared__ float s_tot[4];
  const int b = blockIdx.x, t = threadIdx.x, c = t & 63, q = t >> 6, lane = t & 63;
  const float *Vb = V + ((size_t)b * 64 + c) * 256 + q * 64;
  const float as = asum[(size_t)b * 64 + c];
  auto col_sum = [&](float v) {  // sum over the four d-quarters of cluster c (every thread of the column gets it)
    __syncthreads();
    s_red[q][c] = v;
    __syncthreads();
    return (s_red[0][c] + s_red[1][c]) + (s_red[2][c] + s_red[3][c]);
  };
  auto all_sum = [&](float v) {  // sum over the workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) s_tot[q] = v;
    __syncthreads();
    return (s_tot[0] + s_tot[1]) + (s_tot[2] + s_tot[3]);
  };
  float u[64];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 64; j += 4) {
    const float4 v = *reinterpret_cast<const float4 *>(Vb + j);
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u[j + k] = vv[k] - as * W2[(size_t)(q * 64 + j + k) * 64 + c];
      ss = fmaf(u[j + k], u[j + k], ss);
    }
  }
  const float ssc = col_sum(ss);
  const float ic = rsqrtf(fmaxf(ssc, eps));
  const float tot = all_sum(q == 0 ? ssc * ic * ic : 0.f);  // sum of y1^2 = sum_c ssc * ic^2 (one thread per column)
  const float it = rsqrtf(fmaxf(tot, eps));
  if (!BWD) {
#pragma unroll
    for (int j = 0; j < 64; ++j) out[(size_t)b * 16384 + (size_t)(q * 64 + j) * 64 + c] = u[j] * ic * it;
    if (q == 0) inv_c[(size_t)b * 64 + c] = ic;
    if (t == 0) inv_t[b] = it;
    return;
  }
  // backward.  y = y1 * it:  dy1 = it * (g - y * sum(g y))   [projection only where the norm is not clamped]
  const float *gb = g + (size_t)b * 16384;
  float gv[64];
  float sgy = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    gv[j] = gb[(size_t)(q * 64 + j) * 64 + c];
    sgy = fmaf(gv[j], u[j] * ic * it, sgy);
  }
  const float S = tot > eps ? all_sum(sgy) : 0.f;
  // y1 = u * ic:  du = ic * (dy1 - y1 * sum_d(dy1 y1))
  float sc = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const float y1 = u[j] * ic;
    gv[j] = it * (gv[j] - y1 * it * S);  // dy1
    sc = fmaf(gv[j], y1, sc);
  }
  const float Sc_all = col_sum(sc);  // (barriers inside: every thread calls it, the empty-cluster select comes after)
  const float Sc = ssc > eps ? Sc_all : 0.f;
  float das = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    const float du = ic * (gv[j] - u[j] * ic * Sc);
    u[j] = du;
    const size_t wi = (size_t)(q * 64 + j) * 64 + c;
    das = fmaf(du, W2[wi], das);
    unsafeAtomicAdd(dW2 + wi, -du * as);
  }
  float *dVb = dV + ((size_t)b * 64 + c) * 256 + q * 64;
#pragma unroll
  for (int j = 0; j < 64; j += 4) *reinterpret_cast<float4 *>(dVb + j) = make_float4(u[j], u[j + 1], u[j + 2], u[j + 3]);
  const float dsum = col_sum(das);
  if (q == 0) dasum[(size_t)b * 64 + c] = -dsum;
}

// ---------------------------------------------------------------------------------------------------------------
// Lazy quadruplet loss (core/losses.py:137-200) on the role-ordered descriptors [B | B*P | B*Ng | B] x 256 and its
// gradient in ONE launch (the training graph spends ~60 tiny tensor ops on it):
//   best_pos = min_p |pos_p - q|^2;  trip = max_j relu(m1 + best_pos - |neg_j - q|^2);
//   second = max_j relu(m2 + best_pos - |neg_j - other|^2);  loss = mean_b trip + mean_b second.
// One workgroup per tuple b, thread = descriptor channel (D == 256).  The gradient follows the winning indices (first
// index on ties, like a max / min over a dimension) and vanishes where the hinge is inactive.
__global__ __launch_bounds__(256) void quadruplet_loss_kernel(const float *__restrict__ desc, int B, int P, int Ng,
                                                             float m1, float m2, float *__restrict__ loss,
                                                             float *__restrict__ grad) {
  __shared__ float s_w[4];
  __shared__ float s_d[2 * 64 + 8];  // dneg[Ng], dother[Ng], dpos[P]   (Ng <= 64, P <= 8)
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int D = 256;
  const float *q = desc + (size_t)b * D;
  const float *pos = desc + ((size_t)B + (size_t)b * P) * D;
  const float *neg = desc + ((size_t)B + (size_t)B * P + (size_t)b * Ng) * D;
  const float *oth = desc + ((size_t)B + (size_t)B * P + (size_t)B * Ng + b) * D;
  auto block_sum = [&](float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if (lane == 0) s_w[wave] = v;
    __syncthreads();
    return (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
  };
  const float qv = q[t], ov = oth[t];
  for (int p = 0; p < P; ++p) {
    const float d = pos[(size_t)p * D + t] - qv;
    const float sum = block_sum(d * d);
    if (t == 0) s_d[128 + p] = sum;
  }
  for (int j = 0; j < Ng; ++j) {
    const float nv = neg[(size_t)j * D + t];
    const float d1 = nv - qv, d2 = nv - ov;
    const float a = block_sum(d1 * d1), c = block_sum(d2 * d2);
    if (t == 0) { s_d[j] = a; s_d[64 + j] = c; }
  }
  __syncthreads();
  int ps = 0;
  float best = s_d[128];
  for (int p = 1; p < P; ++p)
    if (s_d[128 + p] < best) { best = s_d[128 + p]; ps = p; }
  int j1 = 0, j2 = 0;
  float t1 = -1.f, t2 = -1.f;  // max over j of the clamped hinges (>= 0)
  float r1 = 0.f, r2 = 0.f;    // the raw hinge at the winner
  for (int j = 0; j < Ng; ++j) {
    const float a = m1 + best - s_d[j], c = m2 + best - s_d[64 + j];
    const float ac = fmaxf(a, 0.f), cc = fmaxf(c, 0.f);
    if (ac > t1) { t1 = ac; j1 = j; r1 = a; }
    if (cc > t2) { t2 = cc; j2 = j; r2 = c; }
  }
  const float invB = 1.f / (float)B;
  if (t == 0) unsafeAtomicAdd(loss, (t1 + t2) * invB);
  // gradient (rows of this tuple only: no atomics needed)
  const float g1 = r1 >= 0.f ? invB : 0.f, g2 = r2 >= 0.f ? invB : 0.f;  // clamp(min=0) passes the gradient at x >= 0
  const float pv = pos[(size_t)ps * D + t];
  const float n1 = neg[(size_t)j1 * D + t], n2 = neg[(size_t)j2 * D + t];
  float *gq = grad + (size_t)b * D, *gp = grad + ((size_t)B + (size_t)b * P) * D;
  float *gn = grad + ((size_t)B + (size_t)B * P + (size_t)b * Ng) * D;
  float *go = grad + ((size_t)B + (size_t)B * P + (size_t)B * Ng + b) * D;
  // d best_pos: q gets -2 (pos - q), pos_ps gets +2 (pos - q), weight g1 + g2;  -d|n1 - q|^2: q gets +2 (n1 - q), n1 gets
  // -2 (n1 - q), weight g1;  -d|n2 - o|^2: n2 gets -2 (n2 - o), o gets +2 (n2 - o), weight g2
  gq[t] = -2.f * (g1 + g2) * (pv - qv) + 2.f * g1 * (n1 - qv);
  for (int p = 0; p < P; ++p) gp[(size_t)p * D + t] = p == ps ? 2.f * (g1 + g2) * (pv - qv) : 0.f;
  for (int j = 0; j < Ng; ++j) {
    float v = 0.f;
    if (j == j1) v -= 2.f * g1 * (n1 - qv);
    if (j == j2) v -= 2.f * g2 * (n2 - ov);
    gn[(size_t)j * D + t] = v;
  }
  go[t] = 2.f * g2 * (n2 - ov);
}

// backward of xn = x * rsqrt(max(sum x^2, eps)) (tf.nn.l2_normalize), one wave per row
__global__ __launch_bounds__(256) void l2norm_rows_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dxn,
                                                             long long R, int C, float eps, float *__restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  float ss = 0.f, dot = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[row * C + c]; ss = fmaf(v, v, ss); dot = fmaf(v, dxn[row * C + c], dot); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { ss += __shfl_xor(ss, off, 64); dot += __shfl_xor(dot, off, 64); }
  const float inv = rsqrtf(fmaxf(ss, eps));
  // d/dx [x * inv]: inv * dxn - x * inv^3 * (x . dxn) where the norm is active, inv * dxn where it is clamped
  const float k = ss > eps ? inv * inv * inv * dot : 0.f;
  for (int c = lane; c < C; c += 64) dx[row * C + c] = fmaf(-x[row * C + c], k, inv * dxn[row * C + c]);
}

// ------------------------------------------------------------------ local-backbone training: element-wise passes
// (stage-1/2 training step, dh3d_amd/training.py LocalTrainer; core/model.py:212-246 with basic_config / detection_config)
// FlexPoolGrad on point-major rows (flex_pool_kernel_gpu.cu.cc:65-93: the gradient of an output goes to the neighbour
// that won the maximum -- argmax holds its point id within the cloud): din[cloud0 + argmax[n,c], c] += dout[n,c].
__global__ __launch_bounds__(256) void flex_pool_pm_bwd_kernel(const float *__restrict__ dout,
                                                              const int32_t *__restrict__ argmax, long long R, int N,
                                                              int C, float *__restrict__ din) {
  const long long total = R * C;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long n = e / C;
    const int c = (int)(e - n * C);
    const long long cloud0 = (n / N) * N;
    atomicAdd(din + (cloud0 + argmax[e]) * C + c, dout[e]);
  }
}

// y = relu(x + x * sigmoid(z))   (se_res_bottleneck's tail, core/backbones.py:52-55), float4 per lane
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(const float4 *__restrict__ x, const float4 *__restrict__ z,
                                                         long long n4, float4 *__restrict__ y) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
    const float4 a = x[e], b = z[e];
    float4 o;
    o.x = fmaxf(a.x + a.x * (1.f / (1.f + __expf(-b.x))), 0.f);
    o.y = fmaxf(a.y + a.y * (1.f / (1.f + __expf(-b.y))), 0.f);
    o.z = fmaxf(a.z + a.z * (1.f / (1.f + __expf(-b.z))), 0.f);
    o.w = fmaxf(a.w + a.w * (1.f / (1.f + __expf(-b.w))), 0.f);
    y[e] = o;
  }
}

__device__ __forceinline__ void se_gate_bwd1(float x, float z, float dy, float &dx, float &dz) {
  const float g = 1.f / (1.f + __expf(-z));
  const float dt = (x + x * g) > 0.f ? dy : 0.f;
  dx = dt * (1.f + g);
  dz = dt * x * g * (1.f - g);
}
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const float4 *__restrict__ x, const float4 *__restrict__ z,
                                                         const float4 *__restrict__ dy, long long n4,
                                                         float4 *__restrict__ dx, float4 *__restrict__ dz) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
    const float4 a = x[e], b = z[e], g = dy[e];
    float4 ox, oz;
    se_gate_bwd1(a.x, b.x, g.x, ox.x, oz.x); se_gate_bwd1(a.y, b.y, g.y, ox.y, oz.y);
    se_gate_bwd1(a.z, b.z, g.z, ox.z, oz.z); se_gate_bwd1(a.w, b.w, g.w, ox.w, oz.w);
    dx[e] = ox; dz[e] = oz;
  }
}

// mode 0: y = relu(x);  mode 1: dx = y > 0 ? dy : 0   (a = x | y, b = unused | dy)
__global__ __launch_bounds__(256) void relu_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, long long n4,
                                                  int mode, float4 *__restrict__ o) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
    const float4 v = a[e];
    float4 r;
    if (mode == 0) {
      r = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    } else {
      const float4 g = b[e];
      r = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
    }
    o[e] = r;
  }
}

// pairwise_dist (core/tf_utils.py:125-136): out[b,i,j] = sum_d (A[b,i,d] - Bm[b,j,d])^2, the differences formed and squared
// as upstream (no |a|^2 + |b|^2 - 2ab expansion: the local losses take sqrt(d + 1e-10) of values near 0).  A 32 x 32 tile of
// pairs per workgroup, both row blocks staged through LDS 32 columns at a time; thread (ti, tj) owns 2 x 2 pairs.
__global__ __launch_bounds__(256) void pairwise_sqdist_kernel(const float *__restrict__ A, const float *__restrict__ Bm, int n,
                                                             int m, int D, float *__restrict__ out) {
  __shared__ float s_a[32][33], s_b[32][33];
  const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
  const float *Ab = A + (size_t)b * n * D, *Bb = Bm + (size_t)b * m * D;
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int d0 = 0; d0 < D; d0 += 32) {
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += 256) {
      const int r = e >> 5, c = e & 31;
      s_a[r][c] = (i0 + r < n && d0 + c < D) ? Ab[(size_t)(i0 + r) * D + d0 + c] : 0.f;
      s_b[r][c] = (j0 + r < m && d0 + c < D) ? Bb[(size_t)(j0 + r) * D + d0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < 32; ++c) {
      const float a0 = s_a[ti][c], a1 = s_a[ti + 16][c], b0 = s_b[tj][c], b1 = s_b[tj + 16][c];
      float t;
      t = a0 - b0; acc[0][0] = fmaf(t, t, acc[0][0]);
      t = a0 - b1; acc[0][1] = fmaf(t, t, acc[0][1]);
      t = a1 - b0; acc[1][0] = fmaf(t, t, acc[1][0]);
      t = a1 - b1; acc[1][1] = fmaf(t, t, acc[1][1]);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int i = i0 + ti + 16 * u, j = j0 + tj + 16 * v;
      if (i < n && j < m) out[((size_t)b * n + i) * m + j] = acc[u][v];
    }
}

inline int flat_grid(long long work) {
  long long g = (work + 255) / 256;
  return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}
inline int row_chunks(long long R, int *rows_per) {
  int chunks = dh3d_cdiv(R, 128);
  chunks = chunks > 1024 ? 1024 : chunks;
  *rows_per = dh3d_cdiv(R, chunks);
  return dh3d_cdiv(R, *rows_per);
}

}  // namespace

DH3D_API int dh3d_bn_colstats(const float *x, long long R, int C, const unsigned char *mask, int rows_per_cloud,
                              double *sum, double *sumsq, void *stream) {
  DH3D_REQUIRE(x && sum && sumsq && R > 0 && C > 0 && (!mask || rows_per_cloud > 0));
  hipStream_t s = (hipStream_t)stream;
  int rows_per;
  const int chunks = row_chunks(R, &rows_per);
  hipLaunchKernelGGL(colstats_kernel, dim3(dh3d_cdiv(C, 64), chunks), dim3(256), 0, s, x, R, C, rows_per, mask,
                     rows_per_cloud > 0 ? rows_per_cloud : 1, sum, sumsq);
  return dh3d_launch_status();
}

DH3D_API int dh3d_pairwise_sqdist(const float *A, const float *Bm, int B, int n, int m, int D, float *out, void *stream) {
  DH3D_REQUIRE(A && Bm && out && B > 0 && n > 0 && m > 0 && D > 0);
  DH3D_SUPPORTED(B <= 65535);
  hipLaunchKernelGGL(pairwise_sqdist_kernel, dim3(dh3d_cdiv(m, 32), dh3d_cdiv(n, 32), B), dim3(256), 0, (hipStream_t)stream, A,
                     Bm, n, m, D, out);
  return dh3d_launch_status();
}

DH3D_API int dh3d_flex_pool_pm_bwd(const float *dout, const int32_t *argmax, int B, int N, int C, float *din, void *stream) {
  DH3D_REQUIRE(dout && argmax && din && B > 0 && N > 0 && C > 0);
  const long long R = (long long)B * N;
  hipLaunchKernelGGL(flex_pool_pm_bwd_kernel, dim3(flat_grid(R * C)), dim3(256), 0, (hipStream_t)stream, dout, argmax, R, N, C,
                     din);
  return dh3d_launch_status();
}

DH3D_API int dh3d_se_gate_fwd(const float *x, const float *z, long long n, float *y, void *stream) {
  DH3D_REQUIRE(x && z && y && n > 0);
  DH3D_SUPPORTED(n % 4 == 0);
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(z), n / 4,
                     reinterpret_cast<float4 *>(y));
  return dh3d_launch_status();
}

DH3D_API int dh3d_se_gate_bwd(const float *x, const float *z, const float *dy, long long n, float *dx, float *dz,
                              void *stream) {
  DH3D_REQUIRE(x && z && dy && dx && dz && n > 0);
  DH3D_SUPPORTED(n % 4 == 0);
  hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(x), reinterpret_cast<const float4 *>(z),
                     reinterpret_cast<const float4 *>(dy), n / 4, reinterpret_cast<float4 *>(dx),
                     reinterpret_cast<float4 *>(dz));
  return dh3d_launch_status();
}

DH3D_API int dh3d_relu_fwd(const float *x, long long n, float *y, void *stream) {
  DH3D_REQUIRE(x && y && n > 0);
  DH3D_SUPPORTED(n % 4 == 0);
  hipLaunchKernelGGL(relu_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(x), (const float4 *)nullptr, n / 4, 0, reinterpret_cast<float4 *>(y));
  return dh3d_launch_status();
}

DH3D_API int dh3d_relu_bwd(const float *y, const float *dy, long long n, float *dx, void *stream) {
  DH3D_REQUIRE(y && dy && dx && n > 0);
  DH3D_SUPPORTED(n % 4 == 0);
  hipLaunchKernelGGL(relu_kernel, dim3(flat_grid(n / 4)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4 *>(y), reinterpret_cast<const float4 *>(dy), n / 4, 1,
                     reinterpret_cast<float4 *>(dx));
  return dh3d_launch_status();
}

DH3D_API int dh3d_scale_shift_act(const float *x, long long R, int C, const float *scale, const float *shift, int relu,
                                  float *y, void *stream) {
  DH3D_REQUIRE(x && scale && shift && y && R > 0 && C > 0);
  DH3D_SUPPORTED(C % 4 == 0);
  hipLaunchKernelGGL(scale_shift_act_kernel, dim3(flat_grid(R * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, R, C,
                     scale, shift, relu, nullptr, y);
  return dh3d_launch_status();
}

DH3D_API int dh3d_scale_shift_act_res(const float *x, long long R, int C, const float *scale, const float *shift, int relu,
                                      const float *residual, float *y, void *stream) {
  DH3D_REQUIRE(x && scale && shift && y && R > 0 && C > 0);
  DH3D_SUPPORTED(C % 4 == 0);
  hipLaunchKernelGGL(scale_shift_act_kernel, dim3(flat_grid(R * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, R, C,
                     scale, shift, relu, residual, y);
  return dh3d_launch_status();
}

DH3D_API int dh3d_row_logit_sigmoid(const float *h, long long R, int C, const float *scale, const float *shift,
                                    const float *w, const float *b, float *att, void *stream) {
  DH3D_REQUIRE(h && scale && shift && w && b && att && R > 0 && C > 0);
  DH3D_SUPPORTED(C % 4 == 0);
  hipLaunchKernelGGL(row_logit_kernel, dim3(dh3d_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, h, R, C, scale, shift,
                     w, b, att);
  return dh3d_launch_status();
}

DH3D_API int dh3d_bn_bwd_sums(const float *x, const float *dy, const float *rowscale, const float *colvec, long long R,
                              int C, const float *mean, const float *rstd, const float *gamma, const float *beta,
                              int relu, const unsigned char *mask, int rows_per_cloud, double *S1, double *S2,
                              double *S3, void *stream) {
  DH3D_REQUIRE(x && mean && rstd && gamma && beta && S1 && S2 && R > 0 && C > 0);
  DH3D_REQUIRE(dy || (rowscale && colvec && S3));
  DH3D_REQUIRE(!mask || rows_per_cloud > 0);
  hipStream_t s = (hipStream_t)stream;
  int rows_per;
  const int chunks = row_chunks(R, &rows_per);
  hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(dh3d_cdiv(C, 64), chunks), dim3(256), 0, s, x, dy, rowscale, colvec, R, C,
                     rows_per, mean, rstd, gamma, beta, relu, mask, rows_per_cloud > 0 ? rows_per_cloud : 1, S1, S2,
                     dy ? nullptr : S3);
  return dh3d_launch_status();
}

DH3D_API int dh3d_bn_bwd_apply(const float *x, const float *dy, const float *rowscale, const float *colvec,
                               long long R, int C, const float *scale, const float *shift, const float *k2,
                               const float *k3, int relu, const unsigned char *mask, int rows_per_cloud, float *dx,
                               void *stream) {
  DH3D_REQUIRE(x && scale && shift && k2 && k3 && dx && R > 0 && C > 0);
  DH3D_REQUIRE(dy || (rowscale && colvec));
  DH3D_REQUIRE(!mask || rows_per_cloud > 0);
  DH3D_SUPPORTED(C % 4 == 0 && C <= 1024 && 256 % (C / 4) == 0);
  const int rpb = 256 / (C / 4);
  long long g = (R + rpb - 1) / rpb;
  g = g > 4096 ? 4096 : g;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, dy, rowscale, colvec, R, C,
                     scale, shift, k2, k3, relu, mask, rows_per_cloud > 0 ? rows_per_cloud : 1, dx);
  return dh3d_launch_status();
}

DH3D_API int dh3d_bn_finalize(const double *sum, const double *sumsq, const double *count, const float *gamma,
                              const float *beta, float eps, float momentum, int unbiased, float *run_mean,
                              float *run_var, int C, float *mean, float *rstd, float *scale, float *shift,
                              void *stream) {
  DH3D_REQUIRE(sum && sumsq && count && gamma && beta && run_mean && run_var && mean && rstd && scale && shift && C > 0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(dh3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sum, sumsq, count,
                     gamma, beta, eps, momentum, unbiased, run_mean, run_var, C, mean, rstd, scale, shift, 1);
  return dh3d_launch_status();
}

// The same from per-cloud partial rows part [2][P][C] (sum | sumsq) as the walks of interp_train.hip / netvlad_train.hip
// leave them: the reduction over P happens here instead of in a launch of its own.
DH3D_API int dh3d_bn_finalize_parts(const double *part, int P, const double *count, const float *gamma, const float *beta,
                                    float eps, float momentum, int unbiased, float *run_mean, float *run_var, int C,
                                    float *mean, float *rstd, float *scale, float *shift, void *stream) {
  DH3D_REQUIRE(part && count && gamma && beta && run_mean && run_var && mean && rstd && scale && shift && C > 0 && P > 0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(dh3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, part,
                     part + (size_t)P * C, count, gamma, beta, eps, momentum, unbiased, run_mean, run_var, C, mean, rstd,
                     scale, shift, P);
  return dh3d_launch_status();
}

DH3D_API int dh3d_bn_bwd_finalize(const double *S1, const double *S2, const double *count, const float *mean,
                                  const float *rstd, const float *gamma, int C, float *k2, float *k3, void *stream) {
  DH3D_REQUIRE(S1 && S2 && count && mean && rstd && gamma && k2 && k3 && C > 0);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(dh3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, S1, S2, count,
                     mean, rstd, gamma, C, k2, k3);
  return dh3d_launch_status();
}

DH3D_API int dh3d_bn_bwd_finalize_parts(const double *part, int nk, int P, const double *count, const float *mean,
                                        const float *rstd, const float *gamma, int C, float *k2, float *k3, float *grads,
                                        void *stream) {
  DH3D_REQUIRE(part && count && mean && rstd && gamma && k2 && k3 && grads && C > 0 && P > 0 && (nk == 2 || nk == 3));
  hipLaunchKernelGGL(bn_bwd_finalize_parts_kernel, dim3(dh3d_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, part, nk, P,
                     count, mean, rstd, gamma, C, k2, k3, grads);
  return dh3d_launch_status();
}

DH3D_API int dh3d_sigmoid_bwd(const float *datt, const float *att, const unsigned char *mask, int rows_per_cloud,
                              long long n, float *dlogit, float *sum, void *stream) {
  DH3D_REQUIRE(datt && att && dlogit && sum && n > 0 && (!mask || rows_per_cloud > 0));
  const int g = (int)(n / 1024 > 1024 ? 1024 : (n + 1023) / 1024);
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, datt, att, mask,
                     rows_per_cloud > 0 ? rows_per_cloud : 1, n, dlogit, sum);
  return dh3d_launch_status();
}

// asum (may be NULL) [R / rows_per_cloud, 64]: column sums of a per cloud, zeroed and accumulated here
DH3D_API int dh3d_netvlad_assign_rows(const float *s, long long R, int Cl, const float *scale, const float *shift,
                                      const float *att, float *a, float *asum, long long rows_per_cloud, void *stream) {
  DH3D_REQUIRE(s && scale && shift && att && a && R > 0);
  DH3D_SUPPORTED(Cl == 64 && (!asum || (rows_per_cloud > 0 && rows_per_cloud % 64 == 0 && R % rows_per_cloud == 0)));
  hipStream_t st = (hipStream_t)stream;
  if (asum && hipMemsetAsync(asum, 0, sizeof(float) * 64 * (size_t)(R / rows_per_cloud), st) != hipSuccess)
    return DH3D_ERR_LAUNCH;
  hipLaunchKernelGGL((netvlad_assign_rows_kernel<false, 64>), dim3(dh3d_cdiv(R, 64)), dim3(256), 0, st, s, R, scale,
                     shift, att, nullptr, a, nullptr, asum, rows_per_cloud);
  return dh3d_launch_status();
}

DH3D_API int dh3d_netvlad_assign_rows_bwd(const float *s, long long R, int Cl, const float *scale, const float *shift,
                                          const float *att, const float *da, float *dz, float *datt, void *stream) {
  DH3D_REQUIRE(s && scale && shift && att && da && dz && datt && R > 0);
  DH3D_SUPPORTED(Cl == 64);
  hipLaunchKernelGGL((netvlad_assign_rows_kernel<true, 4>), dim3(dh3d_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, s, R,
                     scale, shift, att, da, dz, datt, nullptr, 1);
  return dh3d_launch_status();
}

// dist [R, 3] (three_nn) -> w [R, 3] inverse-distance weights
DH3D_API int dh3d_idw_weights(const float *dist, long long R, float *w, void *stream) {
  DH3D_REQUIRE(dist && w && R > 0);
  hipLaunchKernelGGL(idw_weights_kernel, dim3(dh3d_cdiv(R, 256)), dim3(256), 0, (hipStream_t)stream, dist, R, w);
  return dh3d_launch_status();
}

// y = v * sigmoid(g) over n elements, and its backward (dv, dg from dy)
DH3D_API int dh3d_context_gate_fwd(const float *v, const float *g, long long n, float *y, void *stream) {
  DH3D_REQUIRE(v && g && y && n > 0);
  hipLaunchKernelGGL(context_gate_kernel<false>, dim3(dh3d_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, v, g, nullptr,
                     n, y, nullptr);
  return dh3d_launch_status();
}
DH3D_API int dh3d_context_gate_bwd(const float *v, const float *g, const float *dy, long long n, float *dv, float *dg,
                                   void *stream) {
  DH3D_REQUIRE(v && g && dy && dv && dg && n > 0);
  hipLaunchKernelGGL(context_gate_kernel<true>, dim3(dh3d_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, v, g, dy, n, dv,
                     dg);
  return dh3d_launch_status();
}

DH3D_API int dh3d_l2norm_rows_bwd(const float *x, const float *dxn, long long R, int C, float eps, float *dx,
                                  void *stream) {
  DH3D_REQUIRE(x && dxn && dx && R > 0 && C > 0);
  hipLaunchKernelGGL(l2norm_rows_bwd_kernel, dim3(dh3d_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, dxn, R, C, eps,
                     dx);
  return dh3d_launch_status();
}

// V [B, 64, 256] (sum_n a x^T per cloud), asum [B, 64], W2 [256, 64] (cluster_weights2) -> out [B, 16384] in (d, c) order;
// inv_c [B, 64] and inv_t [B] are kept for inspection (the backward recomputes them).
DH3D_API int dh3d_vlad_normalize_fwd(const float *V, const float *asum, const float *W2, int B, int D, int Cl, float eps,
                                     float *out, float *inv_c, float *inv_t, void *stream) {
  DH3D_REQUIRE(V && asum && W2 && out && inv_c && inv_t && B > 0);
  DH3D_SUPPORTED(D == 256 && Cl == 64);
  hipLaunchKernelGGL(vlad_normalize_kernel<false>, dim3(B), dim3(256), 0, (hipStream_t)stream, V, asum, W2, eps, out,
                     inv_c, inv_t, nullptr, nullptr, nullptr, nullptr);
  return dh3d_launch_status();
}

// gradients of the same: dV [B,64,256], dasum [B,64], dW2 [256,64] (zeroed here, f32 atomics over the clouds)
DH3D_API int dh3d_vlad_normalize_bwd(const float *V, const float *asum, const float *W2, const float *grad_out, int B,
                                     int D, int Cl, float eps, float *dV, float *dasum, float *dW2, void *stream) {
  DH3D_REQUIRE(V && asum && W2 && grad_out && dV && dasum && dW2 && B > 0);
  DH3D_SUPPORTED(D == 256 && Cl == 64);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dW2, 0, sizeof(float) * 256 * 64, s) != hipSuccess) return DH3D_ERR_LAUNCH;
  hipLaunchKernelGGL(vlad_normalize_kernel<true>, dim3(B), dim3(256), 0, s, V, asum, W2, eps, nullptr, nullptr, nullptr,
                     grad_out, dV, dasum, dW2);
  return dh3d_launch_status();
}

// desc [B*(2 + P + Ng), 256] role-ordered (queries, positives, negatives, other negatives) -> loss[0] (zeroed here) and
// grad [same shape] = d loss / d desc.  P <= 8, Ng <= 64.
DH3D_API int dh3d_quadruplet_loss(const float *desc, int B, int P, int Ng, int D, float margin, float margin2, float *loss,
                                  float *grad, void *stream) {
  DH3D_REQUIRE(desc && loss && grad && B > 0 && P > 0 && Ng > 0);
  DH3D_SUPPORTED(D == 256 && P <= 8 && Ng <= 64);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(loss, 0, sizeof(float), s) != hipSuccess) return DH3D_ERR_LAUNCH;
  hipLaunchKernelGGL(quadruplet_loss_kernel, dim3(B), dim3(256), 0, s, desc, B, P, Ng, margin, margin2, loss, grad);
  return dh3d_launch_status();
}

// Short tensors (R <= 64): training-mode BatchNorm (+ReLU) forward in ONE launch -- y [R,C], stats [4,C] = mean, rstd,
// scale, shift (for the backward), running buffers updated; mask [R] bytes or NULL (every row its own cloud).
DH3D_API int dh3d_bn_small_fwd(const float *x, int R, int C, const float *gamma, const float *beta, float eps,
                               float momentum, int unbiased, int relu, const unsigned char *mask, float *run_mean,
                               float *run_var, float *stats, float *y, void *stream) {
  DH3D_REQUIRE(x && gamma && beta && run_mean && run_var && stats && y && R > 0 && C > 0);
  DH3D_SUPPORTED(R <= 64);
  hipLaunchKernelGGL(bn_small_kernel<false>, dim3(dh3d_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, x, nullptr, R, C,
                     gamma, beta, eps, momentum, unbiased, relu, mask, run_mean, run_var, stats, y, nullptr, nullptr);
  return dh3d_launch_status();
}

// ... and its backward: dx [R,C], dgamma [C], dbeta [C] from x, dy and the forward's stats
DH3D_API int dh3d_bn_small_bwd(const float *x, const float *dy, int R, int C, const float *gamma, const float *stats,
                               int relu, const unsigned char *mask, float *dx, float *dgamma, float *dbeta,
                               void *stream) {
  DH3D_REQUIRE(x && dy && gamma && stats && dx && dgamma && dbeta && R > 0 && C > 0);
  DH3D_SUPPORTED(R <= 64);
  hipLaunchKernelGGL(bn_small_kernel<true>, dim3(dh3d_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, x, dy, R, C, gamma,
                     nullptr, 0.f, 0.f, 0, relu, mask, nullptr, nullptr, const_cast<float *>(stats), dx, dgamma, dbeta);
  return dh3d_launch_status();
}
