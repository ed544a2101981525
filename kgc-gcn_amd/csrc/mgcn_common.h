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

// fused layer, lockstep generation (layer_fused2.hip): D <= 256 and O <= 208 (the shapes whose alternating layers keep
// the packed weights L2-resident only when every workgroup walks them in step)
bool fused2_takes(int32_t dim_in, int32_t dim_out);
size_t fused2_packed_bytes(int32_t dim_in, int32_t dim_out);
int fused2_pack(int32_t dim_in, int32_t dim_out, const float *w_dev, void *wp_dev, void *stream);
int fused2_launch(int64_t num_nodes, int32_t dim_in, int32_t dim_out, int32_t num_rel_rows, const int32_t *rowptr_dev,
                  const mgcn_edge_rec *rec_dev, const float *x_dev, int64_t ldx, const float *rel_dev,
                  const float *loop_rel_dev, const float *ee_dev, const float *loop_edge_dev, const void *wp_dev,
                  const float *bias_dev, const float *bn_mean_dev, const float *bn_var_dev, const float *bn_gamma_dev,
                  const float *bn_beta_dev, float bn_eps, float *out_dev, int64_t ldo, int64_t node_begin,
                  int64_t node_end, int64_t ee_sub_in, int64_t ee_sub_out, const int32_t *hubinfo_dev, int64_t chunk_begin,
                  const float *partial_dev, const float *rels_weight_dev, float *rel_out_dev, void *stream);

// fused layer, elastic generation (layer_fused3.hip): one slot walk for 256 input columns, exact-width LDS buffers, a ring of staging buffers
// coupled by LDS counters, one contiguous run of rows per workgroup
bool fused3_takes(int32_t dim_in, int32_t dim_out);
size_t fused3_packed_bytes(int32_t dim_in, int32_t dim_out);
int fused3_pack(int32_t dim_in, int32_t dim_out, const float *w_dev, void *wp_dev, void *stream);
int fused3_launch(int64_t num_nodes, int32_t dim_in, int32_t dim_out, int32_t num_rel_rows, const int32_t *rowptr_dev,
                  const mgcn_edge_rec *rec_dev, const float *x_dev, int64_t ldx, const float *rel_dev,
                  const float *loop_rel_dev, const float *ee_dev, const float *loop_edge_dev, const void *wp_dev,
                  const float *bias_dev, const float *bn_mean_dev, const float *bn_var_dev, const float *bn_gamma_dev,
                  const float *bn_beta_dev, float bn_eps, float *out_dev, int64_t ldo, int64_t node_begin,
                  int64_t node_end, int64_t ee_sub_in, int64_t ee_sub_out, const int32_t *hubinfo_dev, int64_t chunk_begin,
                  const float *partial_dev, const float *rels_weight_dev, float *rel_out_dev, const int32_t *row_bounds_dev,
                  int32_t num_row_bounds, int32_t tune, uint32_t *status_dev, void *stream);

// fused layer, phase-alternating generation (layer_fused4.hip): D <= 256, O <= 208; all sixteen waves gather a stage of up to 320
// columns of the concatenated K axis into one LDS image, then all sixteen multiply it
bool fused4_takes(int32_t dim_in, int32_t dim_out);
size_t fused4_packed_bytes(int32_t dim_in, int32_t dim_out);
int fused4_pack(int32_t dim_in, int32_t dim_out, const float *w_dev, void *wp_dev, void *stream);
int fused4_launch(int64_t num_nodes, int32_t dim_in, int32_t dim_out, int32_t num_rel_rows, const int32_t *rowptr_dev,
                  const mgcn_edge_rec *rec_dev, const float *x_dev, int64_t ldx, const float *rel_dev,
                  const float *loop_rel_dev, const float *ee_dev, const float *loop_edge_dev, const void *wp_dev,
                  const float *bias_dev, const float *bn_mean_dev, const float *bn_var_dev, const float *bn_gamma_dev,
                  const float *bn_beta_dev, float bn_eps, float *out_dev, int64_t ldo, int64_t node_begin,
                  int64_t node_end, int64_t ee_sub_in, int64_t ee_sub_out, const int32_t *hubinfo_dev, int64_t chunk_begin,
                  const float *partial_dev, const float *rels_weight_dev, float *rel_out_dev, const int32_t *row_bounds_dev,
                  int32_t num_row_bounds, int32_t tune, void *stream);

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace mgcn
