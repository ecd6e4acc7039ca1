// Library identification for libdh3d_hip.so.
#include "common.h"

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
