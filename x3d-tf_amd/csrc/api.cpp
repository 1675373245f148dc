// error channel + version of the C ABI
#include <stdarg.h>
#include <stdio.h>

#include "../../include/x3d_hip.h"

static thread_local char g_err[512] = "";

void x3d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct X3dDescribe { char* out; int cap; };
thread_local X3dDescribe x3d_describe = {nullptr, 0};
thread_local int* x3d_parts_query = nullptr;   // x3d_pw_wgrad_dw_parts: the launcher reports its partial-slab count here instead of launching

extern "C" const char* x3d_last_error(void) { return g_err; }
extern "C" int x3d_version(void) { return X3D_ABI_VERSION; }   // history: include/x3d_hip.h

// CRC32C (Castagnoli) for the TF tensor-bundle checkpoint reader/writer (host code; not part of the
// device hot path).  Slicing-by-1 table; 15 MB checkpoints take ~40 ms.
#include <stddef.h>
#include <stdint.h>
extern "C" uint32_t x3d_crc32c(const void* data, size_t n, uint32_t crc) {
  static uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t c = i;
      for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
      table[i] = c;
    }
    init = true;
  }
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = crc ^ 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}
