// Shared helpers for the gfx950 kernels of libdh3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dh3d_hip.h"

#define DH3D_API extern "C" __attribute__((visibility("default")))

static inline int dh3d_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Status of the launch just enqueued (no synchronisation).
static inline int dh3d_launch_status() {
  return hipGetLastError() == hipSuccess ? DH3D_OK : DH3D_ERR_LAUNCH;
}

#define DH3D_REQUIRE(cond)                        \
  do {                                            \
    if (!(cond)) return DH3D_ERR_INVALID_ARGUMENT; \
  } while (0)

#define DH3D_SUPPORTED(cond)                 \
  do {                                       \
    if (!(cond)) return DH3D_ERR_UNSUPPORTED; \
  } while (0)

// Raise a kernel's dynamic-LDS cap (gfx950: 160 KiB per workgroup) once per process; idempotent, and kept
// out of the steady state so hipGraph capture never sees it.
#define DH3D_ALLOW_BIG_LDS(kern)                                                              \
  do {                                                                                        \
    static bool dh3d_done_ = false;                                                           \
    if (!dh3d_done_) {                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);      \
      dh3d_done_ = true;                                                                      \
    }                                                                                         \
  } while (0)

// XCD-aware block order.  The dispatcher places block b on XCD b % 8 (observed, used for speed only --
// MI355X_MICROARCH.md "Workgroup dispatch"); each XCD has its own 4 MiB L2.  Tiles are consecutive points of
// a cloud and gather from that cloud's feature map, so give every XCD a CONTIGUOUS range of tiles: the
// map's lines are then fetched into one L2 instead of all eight.  Bijective for any block count.
__device__ __forceinline__ int dh3d_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Epilogue y = act(scale*(x+pre_bias)+shift), by value for kernels.
struct EpilogueArgs {
  const float *pre_bias;
  const float *scale;
  const float *shift;
  int act;
};

static inline EpilogueArgs dh3d_ep(const dh3d_epilogue *ep) {
  EpilogueArgs e{nullptr, nullptr, nullptr, DH3D_ACT_NONE};
  if (ep) {
    e.pre_bias = ep->pre_bias;
    e.scale = ep->scale;
    e.shift = ep->shift;
    e.act = ep->act;
  }
  return e;
}

__device__ __forceinline__ float dh3d_act(float v, int act) {
  if (act == DH3D_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == DH3D_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

__device__ __forceinline__ float dh3d_epilogue_apply(float v, int c, const EpilogueArgs &e) {
  if (e.pre_bias) v += e.pre_bias[c];
  if (e.scale) v *= e.scale[c];
  if (e.shift) v += e.shift[c];
  return dh3d_act(v, e.act);
}

// epilogue coefficients of four consecutive channels in registers (defaults: + 0, x 1, + 0 -- the value is unchanged)
struct Ep4 { float4 pb, sc, sh; };
__device__ __forceinline__ Ep4 ep4_prefetch(const EpilogueArgs &ep, int c4) {
  Ep4 e{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(1.f, 1.f, 1.f, 1.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if (ep.pre_bias) e.pb = *reinterpret_cast<const float4 *>(ep.pre_bias + c4);
  if (ep.scale) e.sc = *reinterpret_cast<const float4 *>(ep.scale + c4);
  if (ep.shift) e.sh = *reinterpret_cast<const float4 *>(ep.shift + c4);
  return e;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
