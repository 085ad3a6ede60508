// Host-only pieces of the C-ABI: error string, device query, radix key compare.
#include <stdarg.h>
#include <string.h>

#include <algorithm>

#include "common.h"

namespace msgl {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int device_cu_count() {
  static int cached = -1;
  if (cached < 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    cached = prop.multiProcessorCount;
  }
  return cached;
}

}  // namespace msgl

extern "C" {

const char* msgl_last_error(void) { return msgl::g_err; }
int msgl_abi_version(void) { return MSGL_ABI_VERSION; }
int msgl_device_cu_count(void) { return msgl::device_cu_count(); }

// Reference semantics: C/src/radix.cpp:19-40 (std::mismatch over the common
// prefix length of two 1-D int32/int64 CPU tensors).
int64_t msgl_fast_compare_key(const void* a, int64_t len_a, const void* b, int64_t len_b,
                              int elem_bytes) {
  if (len_a < 0 || len_b < 0 || (elem_bytes != 4 && elem_bytes != 8) ||
      ((a == nullptr) && len_a > 0) || ((b == nullptr) && len_b > 0)) {
    msgl::set_error("fast_compare_key: need two 1-D int32/int64 host arrays");
    return MSGL_EINVAL;
  }
  const int64_t n = std::min(len_a, len_b);
  if (elem_bytes == 8) {
    const int64_t* x = static_cast<const int64_t*>(a);
    const int64_t* y = static_cast<const int64_t*>(b);
    return std::mismatch(x, x + n, y).first - x;
  }
  const int32_t* x = static_cast<const int32_t*>(a);
  const int32_t* y = static_cast<const int32_t*>(b);
  return std::mismatch(x, x + n, y).first - x;
}

}  // extern "C"
