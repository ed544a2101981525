// Shared by the translation units of libmgcn_hip.so (gfx950 only).
#pragma once
#include <cstdarg>
#include <cstdio>

#include "../../include/mgcn_hip.h"

namespace mgcn {

char *error_buffer();  // thread-local, 512 bytes

inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define MGCN_REQUIRE(cond, ...)                          \
  do {                                                   \
    if (!(cond)) return mgcn::fail(MGCN_EINVAL, __VA_ARGS__); \
  } while (0)

#define MGCN_CHECK_LAUNCH(name)                                                            \
  do {                                                                                     \
    hipError_t e_ = hipGetLastError();                                                     \
    if (e_ != hipSuccess) return mgcn::fail(MGCN_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace mgcn
