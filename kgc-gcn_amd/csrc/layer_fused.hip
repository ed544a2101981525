// Fused layer forward (eval) — C-ABI entry points: mgcn_pack_weights / mgcn_packed_weights_bytes /
// mgcn_layer_fwd_fused (include/mgcn_hip.h (2)+(4)); replaces model.py:29-30, 99-107, 111-118 in one launch.
// The kernels are layer_fused2.hip (D <= 256, O <= 208), layer_fused3.hip (the rest) and, through `tune` only, layer_fused4.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mgcn_common.h"


namespace {
// Three kernels behind one entry point.
//   generation 2  layer_fused2.hip  D <= 256, O <= 208: gather / multiply roles in lockstep, one workgroup barrier per stage (every
//                 workgroup walks the packed weights in step, which keeps them L2-resident while the step's alternating layers
//                 stream 330 MB through the caches): the benchmark's layers;
//   generation 3  layer_fused3.hip  the rest (D <= 1024, O <= 512), and generation 2's shapes when the caller brings work-balanced
//                 runs for a launch that is short of two tiles per CU (FB15k-237): roles coupled by LDS counters. Generations 2
//                 and 3 share the k order, the six products and (O > 128) the weight packing: their rows are bit-identical
//                 (tests/test_gpu_round4.py holds them to it at the benchmark's sizes);
//   generation 4  layer_fused4.hip  D <= 256, O <= 208, round 4's structural experiment (no role split: sixteen waves gather, then
//                 sixteen multiply; K folded): correct, NOT faster (LAB_NOTES.md) — reachable through `tune` only, own packing.
bool lockstep_shape(int32_t dim_in, int32_t dim_out) { return dim_in <= 256 && mgcn::fused2_takes(dim_in, dim_out); }
int pack_generation(int32_t dim_in, int32_t dim_out) { return lockstep_shape(dim_in, dim_out) ? 2 : 3; }
bool generation_takes(int gen, int32_t dim_in, int32_t dim_out) {
  if (gen == 4) return mgcn::fused4_takes(dim_in, dim_out);
  if (gen == 2) return lockstep_shape(dim_in, dim_out);
  return gen == 3 && mgcn::fused3_takes(dim_in, dim_out);
}
}  // namespace

// generation: 0 = the shape's own; 2 / 3 / 4
extern "C" size_t mgcn_packed_weights_bytes_gen(int32_t generation, int32_t dim_in, int32_t dim_out) {
  const int gen = generation ? generation : pack_generation(dim_in, dim_out);
  if (gen == 4) return mgcn::fused4_packed_bytes(dim_in, dim_out);
  if (gen == 2) return mgcn::fused2_packed_bytes(dim_in, dim_out);
  return mgcn::fused3_packed_bytes(dim_in, dim_out);
}

extern "C" size_t mgcn_packed_weights_bytes(int32_t dim_in, int32_t dim_out) {
  return mgcn_packed_weights_bytes_gen(0, dim_in, dim_out);
}

extern "C" int mgcn_pack_weights_gen(int32_t generation, int32_t dim_in, int32_t dim_out, const float *w_dev, float *wp_dev,
                                     size_t wp_bytes, void *stream) {
  MGCN_REQUIRE(dim_in > 0 && dim_out > 0 && w_dev && wp_dev, "pack_weights: bad arguments");
  const int gen = generation ? generation : pack_generation(dim_in, dim_out);
  MGCN_REQUIRE(gen >= 2 && gen <= 4 && generation_takes(gen, dim_in, dim_out),
               "pack_weights: generation %d does not take D=%d O=%d", gen, dim_in, dim_out);
  MGCN_REQUIRE(wp_bytes >= mgcn_packed_weights_bytes_gen(gen, dim_in, dim_out) && mgcn::aligned16(wp_dev),
               "pack_weights: packed buffer too small or misaligned");
  if (gen == 4) return mgcn::fused4_pack(dim_in, dim_out, w_dev, wp_dev, stream);
  if (gen == 2) return mgcn::fused2_pack(dim_in, dim_out, w_dev, wp_dev, stream);
  return mgcn::fused3_pack(dim_in, dim_out, w_dev, wp_dev, stream);
}

extern "C" int mgcn_pack_weights(int32_t dim_in, int32_t dim_out, const float *w_dev, float *wp_dev, size_t wp_bytes,
                                 void *stream) {
  return mgcn_pack_weights_gen(0, dim_in, dim_out, w_dev, wp_dev, wp_bytes, stream);
}

// Which kernel a launch takes when `tune` forces none (2 = lockstep, 3 = elastic). Work-balanced runs are the elastic kernel's: a
// lockstep shape takes it too when the lockstep tiling would leave the chip short of two tiles per CU (FB15k-237: 228 tiles of
// 64 rows on 256 CUs, the heaviest tile 1.15x the mean) and the two packings coincide (O > 128).
extern "C" int mgcn_fused_kernel_generation(int32_t dim_in, int32_t dim_out, int64_t num_rows, int32_t with_row_bounds) {
  if (!lockstep_shape(dim_in, dim_out)) return 3;
  int cus = 256, dev = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const bool few_tiles = (num_rows + 79) / 80 < 2 * int64_t(cus);
  return (with_row_bounds && few_tiles && dim_out > 128) ? 3 : 2;
}

extern "C" int mgcn_layer_fwd_fused(int64_t num_nodes, int64_t num_edges_half, int32_t dim_in, int32_t dim_out,
                                    int32_t num_rel_rows, const int32_t *rowptr_dev, const mgcn_edge_rec *rec_dev,
                                    const float *x_dev, int64_t ldx, const float *rel_dev, const float *loop_rel_dev,
                                    const float *ee_dev, int32_t ee_in_slot_order, const float *loop_edge_dev,
                                    const float *wp_dev, const float *bias_dev, const float *bn_mean_dev,
                                    const float *bn_var_dev, const float *bn_gamma_dev, const float *bn_beta_dev,
                                    float bn_eps, float *out_dev, int64_t ldo, int64_t node_begin, int64_t node_end,
                                    int64_t ee_sub_in, int64_t ee_sub_out, int64_t ee_sub_hub,
                                    const int32_t *hubinfo_dev, const int32_t *chunks_dev, int64_t chunk_begin,
                                    int64_t chunk_end, float *partial_dev, const float *rels_weight_dev,
                                    float *rel_out_dev, const int32_t *row_bounds_dev, int32_t num_row_bounds,
                                    int32_t tune, uint32_t *status_dev, void *stream) {
  MGCN_REQUIRE(num_nodes >= 0 && num_edges_half >= 0 && dim_in > 0 && dim_out > 0 && num_rel_rows > 0,
               "layer_fwd_fused: bad sizes");
  MGCN_REQUIRE(node_begin >= 0 && node_begin <= node_end && node_end <= num_nodes, "layer_fwd_fused: bad node range");
  MGCN_REQUIRE(num_nodes < (int64_t(1) << 31) - 256 && 2 * num_edges_half < (int64_t(1) << 31) - 1,
               "layer_fwd_fused: sizes exceed int32");
  MGCN_REQUIRE(rowptr_dev && x_dev && loop_rel_dev && loop_edge_dev && wp_dev && bn_mean_dev && bn_var_dev &&
                   bn_gamma_dev && bn_beta_dev && (out_dev || node_end == node_begin) && (rel_dev || num_rel_rows == 1) &&
                   (num_edges_half == 0 || rec_dev), "layer_fwd_fused: null pointer");
  MGCN_REQUIRE(ldx >= dim_in && ldo >= dim_out, "layer_fwd_fused: ldx/ldo too small");
  const bool aligned = mgcn::aligned16(x_dev) && mgcn::aligned16(rel_dev) && mgcn::aligned16(loop_rel_dev) &&
                       mgcn::aligned16(loop_edge_dev) && (!ee_dev || mgcn::aligned16(ee_dev)) &&
                       mgcn::aligned16(out_dev) && mgcn::aligned16(wp_dev) && ldx % 4 == 0 && ldo % 4 == 0;
  if (!aligned || !mgcn::fused3_takes(dim_in, dim_out) || !ee_dev || !ee_in_slot_order || ldx >= (int64_t(1) << 31))
    return mgcn::fail(MGCN_EUNSUPPORTED, "layer_fwd_fused: needs 16-byte aligned operands, a per-edge table in slot order, "
                      "D %% 4 == 0, D <= 1024, O %% 4 == 0, O <= 512 (got D=%d O=%d)", dim_in, dim_out);
  const int64_t num_chunks = chunk_end - chunk_begin;
  MGCN_REQUIRE(chunk_begin >= 0 && num_chunks >= 0 && chunk_end < (int64_t(1) << 31) &&
                   (num_chunks == 0 || (hubinfo_dev && chunks_dev && partial_dev && mgcn::aligned16(partial_dev))),
               "layer_fwd_fused: hub chunks need hubinfo / chunks / a 16-byte aligned partial buffer");
  const bool want_rel = rel_out_dev != nullptr && num_rel_rows > 1;
  MGCN_REQUIRE(!want_rel || (rels_weight_dev && rel_dev), "layer_fwd_fused: the relation projection needs rels_weight and rel");
  if (node_end == node_begin && !want_rel) return MGCN_OK;
  if (num_chunks > 0 && node_end > node_begin) {
    if (int rc = mgcn::launch_hub_partials(num_nodes, dim_in, num_rel_rows, rec_dev, x_dev, ldx, rel_dev, loop_rel_dev,
                                           ee_dev, ee_in_slot_order, ee_sub_hub, chunks_dev, chunk_begin, chunk_end,
                                           partial_dev, stream))
      return rc;
  }
  // tune bits 10-11: 0 = the launch's own kernel; 1 / 2 / 3 force generation 4 / 2 / 3 (A/B runs; wp_dev must be packed for it:
  // mgcn_pack_weights_gen; generations 2 and 3 share one packing for O > 128)
  const int forced = ((tune >> 10) & 3) == 1 ? 4 : ((tune >> 10) & 3);
  const int gen = forced == 0 ? mgcn_fused_kernel_generation(dim_in, dim_out, node_end - node_begin, num_row_bounds > 0) : forced;
  MGCN_REQUIRE(forced != 3 || !lockstep_shape(dim_in, dim_out) || dim_out > 128,
               "layer_fwd_fused: tune %d: generation 3 reads another packing than generation 2 for O <= 128", tune);
  MGCN_REQUIRE(generation_takes(gen, dim_in, dim_out), "layer_fwd_fused: tune %d: generation %d does not take D=%d O=%d", tune, gen,
               dim_in, dim_out);
  MGCN_REQUIRE(num_row_bounds >= 0 && num_row_bounds <= 4096 && (num_row_bounds == 0 || row_bounds_dev),
               "layer_fwd_fused: bad row bounds");
  if (gen == 4)
    return mgcn::fused4_launch(num_nodes, dim_in, dim_out, num_rel_rows, rowptr_dev, rec_dev, x_dev, ldx, rel_dev,
                               loop_rel_dev, ee_dev, loop_edge_dev, wp_dev, bias_dev, bn_mean_dev, bn_var_dev, bn_gamma_dev,
                               bn_beta_dev, bn_eps, out_dev, ldo, node_begin, node_end, ee_sub_in, ee_sub_out,
                               num_chunks > 0 ? hubinfo_dev : nullptr, chunk_begin, partial_dev,
                               want_rel ? rels_weight_dev : nullptr, want_rel ? rel_out_dev : nullptr, row_bounds_dev,
                               num_row_bounds, tune, stream);
  if (gen == 2)
    return mgcn::fused2_launch(num_nodes, dim_in, dim_out, num_rel_rows, rowptr_dev, rec_dev, x_dev, ldx, rel_dev,
                               loop_rel_dev, ee_dev, loop_edge_dev, wp_dev, bias_dev, bn_mean_dev, bn_var_dev, bn_gamma_dev,
                               bn_beta_dev, bn_eps, out_dev, ldo, node_begin, node_end, ee_sub_in, ee_sub_out,
                               num_chunks > 0 ? hubinfo_dev : nullptr, chunk_begin, partial_dev,
                               want_rel ? rels_weight_dev : nullptr, want_rel ? rel_out_dev : nullptr, stream);
  return mgcn::fused3_launch(num_nodes, dim_in, dim_out, num_rel_rows, rowptr_dev, rec_dev, x_dev, ldx, rel_dev,
                               loop_rel_dev, ee_dev, loop_edge_dev, wp_dev, bias_dev, bn_mean_dev, bn_var_dev, bn_gamma_dev,
                               bn_beta_dev, bn_eps, out_dev, ldo, node_begin, node_end, ee_sub_in, ee_sub_out,
                               num_chunks > 0 ? hubinfo_dev : nullptr, chunk_begin, partial_dev,
                               want_rel ? rels_weight_dev : nullptr, want_rel ? rel_out_dev : nullptr, row_bounds_dev,
                               num_row_bounds, tune, status_dev, stream);
}
