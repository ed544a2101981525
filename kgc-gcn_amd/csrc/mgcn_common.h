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

// hub pre-pass (aggregate.hip): chunk sums of the hub destinations' slots -> partial_dev [num_chunks, dim]
int launch_hub_partials(int64_t num_nodes, int32_t dim, int32_t num_rel_rows, const mgcn_edge_rec *rec_dev,
                        const float *x_dev, int64_t ldx, const float *rel_dev, const float *loop_rel_dev,
                        const float *ee_dev, int32_t ee_in_slot_order, int64_t ee_sub_hub, const int32_t *chunks_dev,
                        int64_t chunk_begin, int64_t chunk_end, float *partial_dev, void *stream);

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace mgcn
