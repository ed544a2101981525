/*
 * mgcn_hip.h — C ABI of libmgcn_hip.so: the MI355X (gfx950) implementation of the M-GCN hot path.
 *
 * The reference (weilonghu/KGC-GCN) has no FFI of its own: the hot path is Python calling torch /
 * torch_geometric / torch_scatter tensor ops. Each entry point below therefore replaces a span of
 * reference Python, cited as file:line relative to the reference root. The Python host side
 * (kgc-gcn_amd/_native.py) binds these with ctypes; INTEGRATION.md shows the stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - plain C types only; `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every pointer named *_dev is a BORROWED device pointer (owned by the caller, e.g. a torch
 *     tensor that outlives the call); pointers named *_host are host memory;
 *   - all device work is enqueued asynchronously on `stream`; nothing here allocates, frees or
 *     synchronises the device, so every call may be captured into a hipGraph;
 *   - return value 0 = MGCN_OK; otherwise an MGCN_E* code, and mgcn_last_error() returns a
 *     thread-local, human-readable message;
 *   - no global mutable state besides that thread-local error string (no cached device attributes, no environment
 *     variables); re-entrant per device;
 *   - matrices are row-major f32, indices int32 on the device, int64 on the host side of the feeder.
 */
#ifndef MGCN_HIP_H
#define MGCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGCN_ABI_VERSION 4

enum {
  MGCN_OK = 0,
  MGCN_EINVAL = 1,  /* bad argument: null pointer, negative size, index out of range, misalignment */
  MGCN_ELAUNCH = 2, /* hipGetLastError() after a launch */
  MGCN_EUNSUPPORTED = 3
};

/* One CSR slot = one directed edge in destination order. 16 bytes, read as one dwordx4. */
typedef struct mgcn_edge_rec {
  int32_t src;  /* source node (row of the layer input that is gathered)                          */
  int32_t type; /* row of the relation table [2R+1, D]                                             */
  float norm;   /* deg^-1/2[src] * deg^-1/2[dst], degrees counted by SOURCE within the half (Q2)  */
  int32_t eid;  /* reference edge id (row of the per-edge table in reference order)               */
} mgcn_edge_rec;

int mgcn_abi_version(void);
const char *mgcn_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * (0) Graph ingest on the host (replaces the two Python passes of data_loader.py:61-96 for triple
 * files of 10^7-10^8 lines). mgcn_ingest_open reads the three split files (one `subject relation object` triple per
 * line, any ASCII whitespace between tokens, universal newlines) and assigns entity / relation ids in first-seen
 * order over train, valid, test — per line subject, relation, object — with names lower-cased on insertion
 * (data_loader.py:64-70). Error behaviour follows the reference: a line without exactly three tokens is an error
 * (its tuple unpacking raises ValueError); a token that lower-casing changes is an error naming the token (its raw
 * lookup, data_loader.py:84-86, raises KeyError). A token with a byte >= 0x80 returns MGCN_EUNSUPPORTED (Python's
 * str.lower() is Unicode-aware): use the Python reader for such files.
 * mgcn_ingest_count: what = 0 entities, 1 relations (forward only; reverse relation r + R is implicit), 2 / 3 / 4 =
 * triples of train / valid / test. mgcn_ingest_triples: [n, 3] int64 (s, r, o) ids of a split (0 / 1 / 2).
 * mgcn_ingest_names: the names in id order as one byte string + [n+1] offsets (kind 0 entities, 1 relations).
 */
typedef struct mgcn_ingest mgcn_ingest;
int mgcn_ingest_open(const char *train_path, const char *valid_path, const char *test_path, mgcn_ingest **out);
void mgcn_ingest_close(mgcn_ingest *h);
int64_t mgcn_ingest_count(const mgcn_ingest *h, int32_t what);
int mgcn_ingest_triples(const mgcn_ingest *h, int32_t split, int64_t *triples_host);
int64_t mgcn_ingest_names_bytes(const mgcn_ingest *h, int32_t kind);
int mgcn_ingest_names(const mgcn_ingest *h, int32_t kind, char *bytes_host, int64_t *offsets_host);

/* The loader's known-answer index (data_loader.py:80-96: sr2o over the given triples, both directions — the reverse
 * query of (s, r, o) is (o, r + R, s)) in the form mgcn_filter_mask reads: sorted keys s * 2R + r, CSR pointers,
 * sorted distinct tails. Call with keys_host == NULL to obtain *num_keys / *num_tails, then with buffers of those sizes
 * (keys [num_keys], ptr [num_keys + 1], tails [num_tails]). */
int mgcn_filter_index_build(int64_t num_triples, const int64_t *triples_host, int64_t num_relations, int64_t *keys_host,
                            int64_t *ptr_host, int32_t *tails_host, int64_t *num_keys, int64_t *num_tails);

/* ---------------------------------------------------------------------------------------------
 * (1) Feeder — host side. Replaces data_loader.py:132-157 (`_build_graph`: the bi-directional edge
 * list) as consumed by model.py:88-97 (split into the in-half [0,E) and out-half [E,2E), degree
 * norms per half, `compute_norm` model.py:72-80, hoisted out of the step because the graph is
 * static).
 *   edge_index_host [2, 2E] int64 (row 0 = src, row 1 = dst), edge_type_host [2E] int64.
 * Slot layout produced: [in-half segments by destination | out-half segments by destination | hub segments],
 * every segment in edge-id order (the order a CPU scatter-add visits the edges). A destination with more than
 * hub_threshold slots in a half is a HUB (hub_threshold <= 0: no hubs): its segment moves to the hub region and is
 * cut into chunks of hub_chunk slots, so that no lane group ever walks a long list alone (degree skew);
 * the kernels sum each chunk separately and combine a hub's chunk sums in a fixed order (a strided partition over
 * the lane groups of one workgroup, then the group sums in group order) — still no atomics, still reproducible.
 * Outputs (host, caller-allocated):
 *   rowptr_host [2, N+1] int32   ABSOLUTE slot positions of the non-hub segments (a hub's segment there is empty);
 *   rec_host    [2E]             one record per slot;
 *   perm_host   [2E] int64       slot -> reference edge id (to lay the per-edge table out in slot order once);
 *   hubinfo_host [2, N, 2] int32 (first chunk, chunk count) of destination n in half h, (-1, 0) if not a hub;
 *   chunks_host [max_chunks, 4] int32  {begin, end, first chunk of its hub, chunk count of its hub}: absolute slot
 *                                range [begin, end) of every chunk; *num_chunks_host = chunks in use
 *                                (the three may be NULL when hub_threshold <= 0).
 * Optional outputs for the backward pass (all NULL or all non-NULL):
 *   slot_dst_host [2E] int32     destination node of each slot, bit 31 = half;
 *   mirror_host [2E] int32       slot of the reverse edge ((e + E) mod 2E) of each slot's edge: the edges leaving n
 *                                in half h are the reverses of the edges entering n in half 1-h, so by-SOURCE sums
 *                                walk the destination runs / hub chunks of the other half through this map.
 *                                Needs src[e + E] == dst[e] && dst[e + E] == src[e] for every e < E (the loader's
 *                                list, data_loader.py:143-149); the feeder CHECKS it and, for a list that is not
 *                                mirror-symmetric (the operator seam model.py:82-101 accepts any list), fills mirror
 *                                with -1: everything else stays valid (forward, gee, grel), but gx must not be
 *                                requested from mgcn_aggregate_bwd with such a map (the Python seam raises);
 *   typeptr_host [num_rel_rows+1] int32, typeslots_host [2E] int32
 *                                all slots (ascending) grouped by relation-table row.
 * Fails with MGCN_EINVAL if an endpoint is outside [0,N) or a type outside [0,num_rel_rows-1): the last row of the
 * relation table is the self-loop row (model.py:86), which only the self-loop pass reads.
 */
int mgcn_csr_build_host(int64_t num_nodes, int64_t num_edges_half, int64_t num_rel_rows,
                        const int64_t *edge_index_host, const int64_t *edge_type_host,
                        int64_t hub_threshold, int64_t hub_chunk, int32_t *rowptr_host,
                        mgcn_edge_rec *rec_host, int64_t *perm_host, int32_t *hubinfo_host,
                        int32_t *chunks_host, int64_t max_chunks, int64_t *num_chunks_host,
                        int32_t *slot_dst_host, int32_t *mirror_host, int32_t *typeptr_host,
                        int32_t *typeslots_host);

/* ---------------------------------------------------------------------------------------------
 * (2) Aggregation forward. Replaces the gather / message / scatter-add of the three `propagate`
 * calls, model.py:99-101 + 111-118 (+ the identity gathers model.py:29-30), with the weight
 * multiply moved after the sum:
 *   A[n, 0:D)   = sum over in-half  slots p of n:  norm_p * ((x[src_p] * rel[type_p]) * ee_p)
 *   A[n, D:2D)  = same over the out-half
 *   A[n, 2D:3D) = (x[n] * loop_rel) * loop_edge                       (self loop, no norm)
 * The relation table has num_rel_rows rows: rows [0, num_rel_rows-1) at rel_dev, the last (self-loop)
 * row at loop_rel_dev [D] — two pointers so the caller needs no concatenation (model.py:86); they may
 * point into one contiguous [num_rel_rows, D] tensor.
 * Slots of one destination are summed in slot order by one lane group: no atomics, bitwise
 * reproducible. `ee_dev` is the per-edge table: in SLOT order if ee_in_slot_order != 0 (streamed),
 * else in reference edge-id order (gathered through rec.eid); NULL = no per-edge factor.
 *   x_dev [N, D] (ldx floats between rows), rel_dev [num_rel_rows-1, D], loop_edge_dev [D] or NULL
 *   (then the third block is not written and A needs only 2D columns), a_dev [N, lda].
 * Only destinations [node_begin, node_end) are processed (rows of a_dev indexed by global node id): row chunks
 * can be pipelined against (4) on another stream, and a destination partition is one rank's share.
 * Hubs: [chunk_begin, chunk_end) are the chunks of the hubs among [node_begin, node_end) — one run, because the hub
 * region is in node order (all chunks for the whole graph). When it is not empty (hubinfo_dev / chunks_dev from the
 * feeder) ONE pre-pass launch on the same stream writes the chunk sums to partial_dev [chunk_end - chunk_begin, D] and
 * folds every hub's chunk sums into the row of its first chunk (the last lane group to arrive at a hub's counter adds the
 * rows up in row order: a fixed summation tree), and the main launch adds that one row per hub. partial_dev holds
 * mgcn_hub_partial_floats(chunk_end - chunk_begin, D) floats: the rows, then 2 * (chunk_end - chunk_begin) int32 arrival
 * counters which must be ZERO before the first launch that uses the buffer; every completed launch leaves them zero, so a
 * buffer can be reused by later launches on the same stream without clearing (not by launches that may overlap).
 * Table shard (ABI 2): as in the fused launch below, ee_dev may hold only the rows a destination range needs — its
 * in-half slots, its out-half slots, its hub slots, each a contiguous run — with ee_sub_in / ee_sub_out / ee_sub_hub such
 * that the row of (absolute) slot s is s - ee_sub_{region}; all three are 0 with the whole table. This is what lets a
 * rank of the destination partition hold 1/W of a table that does not fit one GPU (configs[4]: 410 GB).
 */
int64_t mgcn_hub_partial_floats(int64_t num_chunks, int32_t dim);
int mgcn_aggregate_fwd(int64_t num_nodes, int64_t num_edges_half, int32_t dim, int32_t num_rel_rows,
                       const int32_t *rowptr_dev, const mgcn_edge_rec *rec_dev, const float *x_dev,
                       int64_t ldx, const float *rel_dev, const float *loop_rel_dev, const float *ee_dev,
                       int32_t ee_in_slot_order, const float *loop_edge_dev, float *a_dev, int64_t lda,
                       int64_t node_begin, int64_t node_end, const int32_t *hubinfo_dev, const int32_t *chunks_dev,
                       int64_t chunk_begin, int64_t chunk_end, float *partial_dev, int64_t ee_sub_in, int64_t ee_sub_out,
                       int64_t ee_sub_hub, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (3) Aggregation backward (autograd through (2); driven by main.py:66). Given g = dL/dA [N, lda]
 * (first 2D columns used):
 *   gx[s]   = sum over slots p with src_p = s of norm_p * g[dst_p, half] * rel[type_p] * ee_p
 *   gee[p]  = norm_p * g[dst_p, half] * x[src_p] * rel[type_p]                 (slot order)
 *   grel[t] = sum over slots p with type_p = t of norm_p * g[dst_p, half] * x[src_p] * ee_p
 * gx[s] walks the destination runs of s (rowptr, both halves) and maps every slot to its reverse edge through
 * `mirror` — within a run that is edge-id order, the order of the CPU index_add; hubs are summed per chunk by a
 * pre-pass and folded exactly as in (2) (hubinfo / chunks / num_hub_chunks of the whole graph). grel walks
 * typeptr/typeslots (slots grouped by relation row, long lists cut into fixed chunks whose partial sums are added in
 * chunk order); all from mgcn_csr_build_host. Sums in a fixed order: no float atomics, bitwise reproducible.
 * Any of gx/gee/grel may be NULL. gx [N, D], gee [2E, D] in slot order, grel [num_rel_rows, D]. ee_dev is in slot
 * order. workspace_dev: mgcn_aggregate_bwd_workspace bytes, 16-byte aligned.
 */
int mgcn_aggregate_bwd(int64_t num_nodes, int64_t num_edges_half, int32_t dim, int32_t num_rel_rows,
                       const int32_t *rowptr_dev, const mgcn_edge_rec *rec_dev,
                       const int32_t *slot_dst_dev /* bit 31 = half */, const int32_t *mirror_dev,
                       const int32_t *hubinfo_dev, const int32_t *chunks_dev, int64_t num_hub_chunks,
                       const int32_t *typeptr_dev, const int32_t *typeslots_dev, const float *x_dev, int64_t ldx, const float *rel_dev, const float *ee_dev,
                       const float *g_dev, int64_t ldg, float *gx_dev, float *gee_dev, float *grel_dev,
                       float *workspace_dev, size_t workspace_bytes, void *stream);
size_t mgcn_aggregate_bwd_workspace(int64_t num_edges_half, int32_t dim, int32_t num_rel_rows, int64_t num_hub_chunks);

/* ---------------------------------------------------------------------------------------------
 * (4) Dense step + epilogue (f32 MFMA, exact f32). Replaces model.py:116 (moved after the sum) and
 * model.py:103-106 in eval mode:
 *   out = tanh( BN_eval( (A[:,0:D) W_in + A[:,D:2D) W_out + A[:,2D:3D) W_loop) / 3 + bias ) )
 * BN_eval(v) = (v - mean) / sqrt(var + eps) * gamma + beta. bias_dev may be NULL.
 * w_dev [3*dim_in, dim_out] = W_in, W_out, W_loop stacked by rows (one contiguous matrix).
 */
int mgcn_dense_bn_tanh_fwd(int64_t num_nodes, int32_t dim_in, int32_t dim_out, const float *a_dev, int64_t lda,
                           const float *w_dev, const float *bias_dev, const float *bn_mean_dev, const float *bn_var_dev,
                           const float *bn_gamma_dev, const float *bn_beta_dev, float bn_eps,
                           float *out_dev, int64_t ldo, void *stream);

/* (2)+(4) in ONE launch (eval mode): out = tanh(BN_eval((A_in W_in + A_out W_out + A_loop W_loop)/3 + bias)); the
 * aggregates of (2) are built tile by tile in LDS and multiplied there, they never reach HBM. One 1024-thread workgroup
 * per CU owns one contiguous run of destinations; 32-lane groups (16 bytes per lane) sum their destinations' slots in
 * slot order — the sums of (2) —, split each finished row exactly into three bf16 pieces in LDS, and
 * v_mfma_f32_16x16x32_bf16 multiplies them; the epilogue runs on the accumulators. Kernels behind this entry point
 * (mgcn_fused_kernel_generation tells which one a launch takes):
 *   generation 2, dim_in <= 256 and dim_out <= 208  csrc/layer_fused2.hip: eight waves gather, eight multiply; tiles of 80
 *       (or 64) destinations, stages of 128 input columns per mode, two LDS images, one workgroup barrier per stage;
 *   generation 3, otherwise (dim_in <= 1024, dim_out <= 512), and generation 2's shapes when the caller brings row bounds
 *       for a launch short of two tiles per CU  csrc/layer_fused3.hip: one contiguous run of rows per workgroup in tiles of
 *       48-80, stages of 128 or 256 columns, a ring of f32 staging buffers coupled by LDS counters, 13 or 32 column tiles.
 *       Generations 2 and 3 run the same arithmetic (k order, products, and for dim_out > 128 the weight packing): rows
 *       are bit-identical between them;
 *   generation 4, dim_in <= 256 and dim_out <= 208, ONLY through `tune`  csrc/layer_fused4.hip (round 4's experiment: all
 *       sixteen waves gather a stage of up to 320 columns of the concatenated K axis, then all sixteen multiply it; own
 *       packing, own k order: rows differ from generations 2 / 3 in the last bits; not faster, see LAB_NOTES.md).
 * Arguments as in (2) and (4), except that the weights are passed in MFMA fragment order: wp_dev = mgcn_pack_weights() of
 * the stacked [3*dim_in, dim_out] matrix (mgcn_packed_weights_bytes bytes, 16-byte aligned; re-pack whenever a weight
 * changes; the *_gen forms pack for a generation forced through `tune`; 0 = the shape's own, which generations 2 and 3 share for
 * dim_out > 128). Returns MGCN_EUNSUPPORTED (and does nothing) unless all operands are 16-byte aligned, ee_dev is given in
 * slot order, dim_in % 4 == 0, dim_in <= 1024, dim_out % 4 == 0 and dim_out <= 512 — callers then use (2) followed by (4).
 * NUMERIC CONTRACT. The dense step is not the exact-f32 MFMA of (4): every aggregate a and weight w is split EXACTLY
 * into three bf16 pieces (hi = bf16(v) rounded to nearest, mid = bf16(v - hi), lo = v - hi - mid; hi + mid + lo == v bit
 * for bit) and the six products hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid are accumulated in f32; the dropped terms
 * are below 2^-26 |a||w| per product. For finite inputs |out - exact| <= 4 u B + 2e-7 with u = 2^-24 and
 * B = sum_k |a_k||w_k| * |gamma| / (3 sqrt(var + eps)) (tests/test_gpu_round3.py holds both this launch and (2)+(4) to
 * it on rows with 2^40 of dynamic range; on the benchmark's data both are within 5e-7 of float64). Results are
 * bit-identical across launches of this entry point with `tune` bits 10-11 = 0 (whole graph, any destination range, any
 * table shard, any row bounds, either of generations 2 / 3, any tile geometry) but differ from (2)+(4) in the last bits (<= 2e-6 on a tanh output for
 * dim_in <= 256). A non-finite input (inf / NaN) in a gathered row makes that destination's output row NaN, where
 * (2)+(4) may return +-1; other rows are unaffected.
 * Destination partition: only destinations [node_begin, node_end) are computed; out_dev holds THOSE rows
 * (row 0 = node_begin). A rank may hold only its shard of the slot-ordered per-edge table — the rows of the in-half
 * slots [rowptr_in[node_begin], rowptr_in[node_end]) followed by those of the out-half slots of the same nodes — and
 * then the hub slots [chunks[chunk_begin].begin, chunks[chunk_end - 1].end) of the same nodes — and passes
 * ee_sub_in / ee_sub_out / ee_sub_hub such that the row of (absolute) slot s is s - ee_sub_{region}; with the whole
 * table all three are 0. x_dev is always the whole [N, D] layer input.
 * Hubs as in (2): hubinfo_dev / chunks_dev / [chunk_begin, chunk_end) / partial_dev [mgcn_hub_partial_floats(chunk_end -
 * chunk_begin, dim_in)] with its counters zero (one pre-pass launch before the layer's launch).
 * rel_out_dev (optional, [num_rel_rows - 1, dim_out]) = rel_dev @ rels_weight_dev [dim_in, dim_out] (model.py:107, the
 * relations the next layer / the scorer read) computed by the same launch after its last gather stage,
 * with the arithmetic of mgcn_matmul_f32's small-matrix kernel (bit-identical results); NULL = not computed.
 * row_bounds_dev (optional; NULL / 0 = equal runs): num_row_bounds + 1 strictly increasing row offsets from node_begin, first 0,
 * last node_end - node_begin: workgroup i takes destinations [b_i, b_i+1) — the caller's work-balanced runs
 * (slots + a constant per row; one run per CU; no run longer than the equal split ceil(rows / runs) rounded up to 16 rows (up to 80) or to 80 rows (past it), which is what the tile
 * height is chosen for), computed once per graph on the host (GraphCSR.workgroup_bounds). Rows do not
 * depend on the runs (fixed k order per row). With bounds given, a generation-2 shape whose tiling would leave the chip short of
 * two tiles per CU takes generation 3 (dim_out > 128); generation 2 itself ignores them. The bounds are read by the launch, not
 * checked: offsets outside the range are the caller's error.
 * tune: 0 = automatic. For A/B runs only (never needed for correctness): bits 0-3 row tiles per tile (3..5); bits 4-7
 * generation 3: staging buffers (1..4), generation 4: slots per gather batch (2 / 4 / 8); bits 8-9 relation table in LDS
 * (1 = never); bits 10-11 force a generation (1 = generation 4, 2, 3; wp_dev must then come from mgcn_pack_weights_gen for
 * it); bits 12-13 generation 3: input columns per slot walk (1 = 128, 2 = 256).
 * status_dev (optional, one zero-initialised uint32 in device memory): generation 3 couples its roles through LDS counters
 * with BOUNDED spins; a spin that runs out (a wave parked for ~0.1 s by a debugger, a preemption, or a protocol error)
 * lets its wave go on, the rows of that tile are then garbage, and bit 0 of *status_dev is set: callers check the word at
 * their next synchronisation point (kgc-gcn_amd/_native.py check_fused_status). Generations 2 and 4 have no spins. */
/* The kernel a launch of (2b) over num_rows destinations takes with `tune` bits 10-11 = 0: 2 (layer_fused2.hip) or 3
 * (layer_fused3.hip). Informational (profiles, benchmarks name the kernel they measured). */
int mgcn_fused_kernel_generation(int32_t dim_in, int32_t dim_out, int64_t num_rows, int32_t with_row_bounds);
int mgcn_layer_fwd_fused(int64_t num_nodes, int64_t num_edges_half, int32_t dim_in, int32_t dim_out,
                         int32_t num_rel_rows, const int32_t *rowptr_dev, const mgcn_edge_rec *rec_dev,
                         const float *x_dev, int64_t ldx, const float *rel_dev, const float *loop_rel_dev,
                         const float *ee_dev, int32_t ee_in_slot_order, const float *loop_edge_dev,
                         const float *wp_dev, const float *bias_dev, const float *bn_mean_dev,
                         const float *bn_var_dev, const float *bn_gamma_dev, const float *bn_beta_dev,
                         float bn_eps, float *out_dev, int64_t ldo, int64_t node_begin, int64_t node_end,
                         int64_t ee_sub_in, int64_t ee_sub_out, int64_t ee_sub_hub, const int32_t *hubinfo_dev,
                         const int32_t *chunks_dev, int64_t chunk_begin, int64_t chunk_end, float *partial_dev,
                         const float *rels_weight_dev, float *rel_out_dev, const int32_t *row_bounds_dev,
                         int32_t num_row_bounds, int32_t tune, uint32_t *status_dev, void *stream);
int mgcn_pack_weights(int32_t dim_in, int32_t dim_out, const float *w_dev, float *wp_dev, size_t wp_bytes, void *stream);
size_t mgcn_packed_weights_bytes(int32_t dim_in, int32_t dim_out);
int mgcn_pack_weights_gen(int32_t generation, int32_t dim_in, int32_t dim_out, const float *w_dev, float *wp_dev,
                          size_t wp_bytes, void *stream);
size_t mgcn_packed_weights_bytes_gen(int32_t generation, int32_t dim_in, int32_t dim_out);

/* ---------------------------------------------------------------------------------------------
 * (4t) The layer's epilogue in TRAINING mode and its backward (model.py:103-106 under .train(), driven by main.py:61-66):
 *   z = (u_in + u_out + u_loop) / 3 (+ bias)        u_* [N, O] = the three products of model.py:116, dropout already applied
 *   y = tanh((z - mean) * rstd * gamma + beta)      mean / rstd = BATCH statistics over the N rows (biased variance),
 * running_mean / running_var updated as nn.BatchNorm1d does (momentum, unbiased variance); either both NULL or both given.
 * z, y [N, O] contiguous; save_mean / save_rstd [O] are kept for the backward. Reductions over rows are two-stage with fixed
 * row blocks (bitwise reproducible). workspace: mgcn_bn_tanh_train_workspace(N, O) bytes.
 * Backward: given gy = dL/dy, returns gz = dL/dz [N, O], gu = gz / 3 (= dL/du_* before the dropout masks), ggamma, gbeta [O].
 */
size_t mgcn_bn_tanh_train_workspace(int64_t num_rows, int32_t dim_out);
int mgcn_bn_tanh_train_fwd(int64_t num_rows, int32_t dim_out, const float *u_in_dev, const float *u_out_dev,
                           const float *u_loop_dev, int64_t ldu, const float *bias_dev, const float *gamma_dev,
                           const float *beta_dev, float *running_mean_dev, float *running_var_dev, float momentum, float eps,
                           float *z_dev, float *y_dev, float *save_mean_dev, float *save_rstd_dev, float *workspace_dev,
                           size_t workspace_bytes, void *stream);
int mgcn_bn_tanh_train_bwd(int64_t num_rows, int32_t dim_out, const float *z_dev, const float *y_dev, const float *gy_dev,
                           const float *save_mean_dev, const float *save_rstd_dev, const float *gamma_dev, float *gz_dev,
                           float *gu_dev, float *ggamma_dev, float *gbeta_dev, float *workspace_dev, size_t workspace_bytes,
                           void *stream);

/* C[M, N] = A^T B for A [K, M] (lda), B [K, N] (ldb): the weight gradient of model.py:116, dW = aggregate^T g (K = number of
 * nodes). Split over K (fixed ranges, partial products added in range order: reproducible), exact-f32 MFMA. M <= 208, N <= 256,
 * else MGCN_EUNSUPPORTED. workspace: mgcn_matmul_tn_workspace(K, M, N) bytes. */
size_t mgcn_matmul_tn_workspace(int64_t k, int32_t m, int32_t n);
int mgcn_matmul_tn_f32(int64_t k, int32_t m, int32_t n, const float *a_dev, int64_t lda, const float *b_dev, int64_t ldb,
                       float *c_dev, int64_t ldc, float *workspace_dev, size_t workspace_bytes, void *stream);

/* Plain C[M,N] = A[M,K] @ B[K,N] on the same f32 MFMA kernel (model.py:107, the relation projection). */
int mgcn_matmul_f32(int64_t m, int32_t k, int32_t n, const float *a_dev, int64_t lda, const float *b_dev,
                    int64_t ldb, float *c_dev, int64_t ldc, void *stream);

/* ---------------------------------------------------------------------------------------------
 * (5) Full-graph scoring and filtered ranking. Replaces model.py:177-179 and main.py:122-126.
 *   score[b, n] = sigmoid( x[b,:] . ent[n,:] + bias[n] ),  x [B, O], ent [n_local, O].
 * mgcn_score_fwd materialises score [B, n_local] (training / the drop-in forward()).
 * mgcn_score_target computes target[b] = score[b, obj[b]] for the queries whose obj lies in
 *   [ent_row0, ent_row0 + n_local) with the SAME arithmetic as the other two (others untouched).
 * One arithmetic for all three, chosen by shape alone: 16-byte aligned operands with dim % 4 == 0 and dim <= 352 take the
 *   six-product bf16 split on the bf16 MFMA (exact three-way split of both operands, f32 accumulation: the numeric
 *   contract of (2)+(4) in one launch below; 6/16 of the exact-f32 MFMA's time); other shapes the exact-f32 MFMA. A score
 *   is the same f32 value whichever of the three entry points computes it, so counts are exact against a recount over
 *   mgcn_score_fwd's scores.
 * mgcn_score_rank never materialises the scores: for every b it adds to counts[b, 0..2]
 *   gt   = #{n != obj[b], not filtered : score[b,n] >  target[b]}
 *   tl   = #{n != obj[b], not filtered, n <  obj[b] : score[b,n] == target[b]}   (ties_lower)
 *   ties = #{n != obj[b], not filtered : score[b,n] == target[b]}
 * where n is filtered iff label[b, n] != 0 after the reference's label.byte() (f32 rows, stride ldl, the shard's own
 * columns), or — when mask_dev is given instead of label_dev — iff bit (n & 31) of mask[b, n >> 5] is set
 * (uint32 rows, stride ldm words; built on the device by mgcn_filter_mask). Exactly one of label_dev / mask_dev.
 * counts [B,3] int64 must be zeroed by the caller; integer atomics => order independent.
 * rank = 1 + gt (+ tl under the stable tie rule). With an entity shard per GPU the caller sums
 * counts over ranks (RCCL all-reduce); `ent_row0` is the shard's first global entity id.
 */
int mgcn_score_fwd(int32_t batch, int64_t n_local, int32_t dim, const float *x_dev, int64_t ldx,
                   const float *ent_dev, int64_t lde, const float *bias_dev, float *score_dev, int64_t lds,
                   void *stream);
int mgcn_score_target(int32_t batch, int64_t n_local, int64_t ent_row0, int32_t dim, const float *x_dev,
                      int64_t ldx, const float *ent_dev, int64_t lde, const float *bias_dev,
                      const int64_t *obj_dev, float *target_dev, void *stream);
int mgcn_score_rank(int32_t batch, int64_t n_local, int64_t ent_row0, int32_t dim, const float *x_dev,
                    int64_t ldx, const float *ent_dev, int64_t lde, const float *bias_dev,
                    const int64_t *obj_dev, const float *target_dev, const float *label_dev, int64_t ldl,
                    const uint32_t *mask_dev, int64_t ldm, int64_t *counts_dev, void *stream);
/* Training loss fused with the scoring pass (replaces model.py:177-179 + model.py:42-44 and the
 * autograd of both down to the logits, main.py:61-66): for z[b, n] = x[b,:] . ent[n,:] + bias[n], p = sigmoid(z) and
 * targets y[b, n] = hot where bit n of mask[b] is set, cold elsewhere (mgcn_filter_mask over the TRAIN index; hot / cold
 * as in mgcn_label_rows), ONE launch writes
 *   grad_logit_dev [n_local, ldg] (row = entity, column = query): d mean-BCE / d z = (p - y) / max(p(1-p), 1e-12) * p(1-p)
 *                  * inv_count — torch's BCELoss and sigmoid backward formulas multiplied out, inv_count = 1 / (B * N);
 *   loss_partial_dev [mgcn_score_bce_partials(batch, n_local)]: per-workgroup sums of
 *                  (y - 1) * max(log1p(-p), -100) - y * max(log(p), -100), reduced in a fixed order; the loss is their
 *                  sum * inv_count.
 * Neither the scores nor the [B, N] targets are materialised. The caller finishes the backward with plain GEMMs:
 * d ent = G @ x (mgcn_matmul_f32), d x = G^T @ ent, d bias = row sums of G. Returns MGCN_EUNSUPPORTED unless operands
 * are 16-byte aligned, dim % 4 == 0 and batch % 4 == 0 (callers then use mgcn_score_fwd + the framework's loss). */
int64_t mgcn_score_bce_partials(int32_t batch, int64_t n_local);
int mgcn_score_bce_fwd(int32_t batch, int64_t n_local, int32_t dim, const float *x_dev, int64_t ldx, const float *ent_dev,
                       int64_t lde, const float *bias_dev, const uint32_t *mask_dev, int64_t ldm, float hot, float cold,
                       float inv_count, float *grad_logit_dev, int64_t ldg, float *loss_partial_dev, void *stream);

/* Filter bits on the device (replaces building + shipping the dense [B, N] label block of data_loader.py:34-51
 * for evaluation). keys_dev [num_keys] sorted int64 (subject * num_rel_ids + relation), ptr_dev
 * [num_keys+1], tails_dev [ptr[num_keys]] int32: the known tails of each (subject, relation). For query b with key
 * qkey_dev[b] the bits of its tails inside [ent_row0, ent_row0 + n_local) are set in mask_dev[b, :] (zeroed here). */
int mgcn_filter_mask(int32_t batch, const int64_t *qkey_dev, int64_t num_keys, const int64_t *keys_dev,
                     const int64_t *ptr_dev, const int32_t *tails_dev, int64_t ent_row0, int64_t n_local,
                     uint32_t *mask_dev, int64_t ldm, void *stream);

/* Training targets on the device (SURVEY N2 for the train loop; replaces the per-sample dense label rows of
 * data_loader.py:34-51 and their host-to-device copy, main.py:62): out_dev[b, n] = hot if entity ent_row0 + n is a known
 * tail of query b's key, else cold. The caller passes hot = (1 - eps) * 1 + 1/N and cold = (1 - eps) * 0 + 1/N evaluated
 * in f32 as numpy does (data_loader.py:41-43), or 1 and 0 without smoothing. Index arrays as in mgcn_filter_mask
 * (built over the TRAIN split). out_dev [batch, ldo >= n_local]. */
int mgcn_label_rows(int32_t batch, const int64_t *qkey_dev, int64_t num_keys, const int64_t *keys_dev,
                    const int64_t *ptr_dev, const int32_t *tails_dev, int64_t ent_row0, int64_t n_local, float hot,
                    float cold, float *out_dev, int64_t ldo, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MGCN_HIP_H */
