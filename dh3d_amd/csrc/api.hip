// Library identification for libdh3d_hip.so.
#include "common.h"

namespace {
// Staging copy for the serving loop: either side may be PINNED HOST memory (hipHostMalloc'd memory is device-addressable;
// the kernel reads / writes it over the host link).  A kernel on the slot's own compute queue instead of hipMemcpyAsync:
// the copy engines' hand-over to the compute queue cost 0.16 ms per step of the local pipeline (tools/streaming_probe.py).
__global__ __launch_bounds__(256) void stage_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
                                                         const unsigned char *__restrict__ src_tail,
                                                         unsigned char *__restrict__ dst_tail, int ntail) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
}  // namespace

DH3D_API int dh3d_stage_copy(const void *src, void *dst, size_t bytes, int device_to_device, void *stream) {
  DH3D_REQUIRE(src && dst && bytes > 0);
  DH3D_SUPPORTED((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  const size_t n16 = bytes / 16;
  const int ntail = (int)(bytes - n16 * 16);
  // a few CUs are enough to keep a host link busy (and all a copy should take from the steps in flight); a copy between
  // two device buffers (a consumer's own buffer for a slot's outputs) gets four 16-byte words per thread on up to 2048 blocks
  const size_t want = device_to_device ? (n16 + 1023) / 1024 : (n16 + 255) / 256;
  const size_t cap = device_to_device ? 2048 : 64;
  const int blocks = (int)(want < 1 ? 1 : (want > cap ? cap : want));
  hipLaunchKernelGGL(stage_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const uint4 *>(src),
                     static_cast<uint4 *>(dst), n16, static_cast<const unsigned char *>(src) + n16 * 16,
                     static_cast<unsigned char *>(dst) + n16 * 16, ntail);
  return dh3d_launch_status();
}

DH3D_API int dh3d_version(void) { return 100; }
DH3D_API int dh3d_abi_version(void) { return DH3D_ABI_VERSION; }
DH3D_API const char *dh3d_arch(void) { return "gfx950"; }
DH3D_API const char *dh3d_source_hash(void) {
  return
#include "source_hash.inc"
      ;
}
DH3D_API const char *dh3d_status_string(int st) {
  switch (st) {
    case DH3D_OK: return "ok";
    case DH3D_ERR_INVALID_ARGUMENT: return "invalid argument";
    case DH3D_ERR_UNSUPPORTED: return "unsupported shape";
    case DH3D_ERR_LAUNCH: return "kernel launch failed";
    default: return "unknown status";
  }
}
