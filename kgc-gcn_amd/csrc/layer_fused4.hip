// Fused layer forward (eval), fourth generation — aggregation + dense step + epilogue in ONE launch (gfx950);
// replaces model.py:29-30, 99-107, 111-118 for one run of destination rows per workgroup. Takes D <= 256, O <= 208.
//
// What differs from layer_fused2.hip / layer_fused3.hip: NO role split. All sixteen waves of the CU's one 1024-thread
// workgroup GATHER a tile, then all sixteen MULTIPLY it, alternating:
//   G  thirty-two 32-lane groups walk the tile's slots (rows dealt by work, sums in slot order: the sums of
//      agg_fwd_kernel) with the whole register file for loads in flight — nothing else competes for the SIMD's issue
//      port — and write each finished row, split exactly into three bf16 pieces, into ONE LDS image that holds the
//      whole stage: up to 320 columns of the CONCATENATED K axis [in-half | self loop | out-half] (a 100-wide layer:
//      all three modes = one stage per tile; a 200-wide layer: two stages, only the cheap self-loop mode is cut);
//   M  four waves per SIMD issue v_mfma_f32_16x16x32_bf16 back to back (the six significant products of the split
//      operands, f32 accumulation: f32-faithful, see layer_fused3.hip / DESIGN.md): wave (simd, j < 3) owns column
//      tile 3 simd + j for all row tiles, wave (simd, 3) the 13th column tile's row tiles simd, simd + 4; weights
//      pre-split, pre-packed in the concatenated K order (pack4_kernel), streamed from L2 one k-block ahead. The
//      three modes' K tails are folded: K = 3 D is padded ONCE (300 -> 320, 600 -> 608), not per mode (3 x 128,
//      3 x 224): 10 instead of 12 and 19 instead of 21 k-blocks per tile.
// Two workgroup barriers per stage (LDS only: vector memory stays in flight). Inside a CU the two phases do not overlap;
// across the chip they do (workgroups drift apart: while one CU multiplies, its neighbours' gathers have the memory
// system), and each phase runs at what its own unit gives: the lockstep / elastic kernels' roles, sharing one issue
// port per SIMD, ran at ~15-25 cycles per instruction per wave and their times ADDED (DESIGN.md / LAB_NOTES.md).
// Rows are bit-identical whichever launch (whole graph, a destination range, a table shard), run or tile computes
// them: fixed slot order, fixed k order.
// STATUS (round 4, LAB_NOTES.md): correct under tools/stress_fused.py and the -m gpu parity tests, and NOT faster than the
// kernels it was meant to replace (WN18RR layers 61 / 118 us against 59 / 96 us): each phase is a chain of memory round trips
// of 2-4 k cycles under the chip-wide bursts that lockstep phases produce, the epilogue's stores sit in front of the next
// phase's loads in the wave's in-order vmcnt queue, and with two tiles per CU a stagger costs a tile of fixed overhead. It
// is therefore reachable through `tune` bits 10-11 = 1 only (with its own weight packing) and dispatched for no shape.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mgcn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int T4 = 1024;
constexpr int NT4 = 13;        // column tiles of the multiply (O <= 208; narrower outputs ride along zero-padded)
constexpr int OP4 = NT4 * 16;
constexpr int SKB4 = 10;       // k-blocks (of 32 columns of the concatenated K axis) per stage image

struct Args4 {
  const int32_t *rowptr;
  const int4 *rec;
  const float *x, *rel, *loop_rel, *ee, *loop_edge;
  const u32x4 *wp;        // packed weights [KB][NT4][3][64] (8 bf16 per lane), pack4_kernel
  const float *bias, *bn_mean, *bn_var, *bn_gamma, *bn_beta;
  float *out;
  int64_t ldx, ldo;
  int32_t n, d, o, rel_rows;
  int32_t node0, node1;   // destinations [node0, node1) are this launch's share; out row 0 = node0
  int32_t ee_sub[2];      // slot-order per-edge table shard: row of (absolute) slot s of half h = s - ee_sub[h]
  const int2 *hubinfo;    // [2][N] (first chunk, chunk count) or null
  const float *partial;   // folded hub totals (pre-pass), row (first chunk - chunk0)
  int32_t chunk0;
  const float *rw;        // relation projection: rels_weight [D, O] (model.py:107) or null
  float *rel_out;         // [rel_rows - 1, O]
  int32_t kb_total, nstage, skb, ncc;   // k-blocks of the K axis, stages per tile, k-blocks per stage, 16-B chunk columns of the image
  int32_t rows_per_wg;
  int32_t nphase;         // phase groups of the stagger (1 = none): group ph's FIRST tile is cut so that its gather phases fall into the others' multiply phases
  const int32_t *bounds;  // [grid + 1] row offsets from node0 of the workgroups' runs (work-balanced), or null: equal runs
  float bn_eps;
#ifdef MGCN_DIAG
  unsigned long long *diag;   // [grid][16 waves][64] s_memtime stamps (tools/fused4_timeline.py)
#endif
};
#ifdef MGCN_DIAG
#define STAMP4(idx)                                                                                                   \
  do {                                                                                                                \
    if (p.diag && lane == 0 && (idx) < 64) p.diag[(int64_t(blockIdx.x) * 16 + wave) * 64 + (idx)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define STAMP4(idx) do {} while (0)
#endif

__device__ __forceinline__ float tanh4_(float v) {   // exp2 + rcp, 7 VALU per value
  const float t = __builtin_amdgcn_exp2f(fabsf(v) * -2.885390081777927f);
  return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), v);
}

// Exact three-way split (layer_fused3.hip): hi = bf16(v) rounded to nearest, mid = bf16(v - hi), lo = v - hi - mid.
__device__ __forceinline__ void split3p(float v0, float v1, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v0, v1}, bf16x2));            // v_cvt_pk_bf16_f32
  const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
  m = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}
__device__ __forceinline__ float4 f4mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4axpy4(float4 s, float4 m, float w) {
  return make_float4(s.x + m.x * w, s.y + m.y * w, s.z + m.z * w, s.w + m.w * w);
}

// Position mi of the concatenated K axis -> mode of the CSR / the stacked weights (0 in-half, 1 out-half, 2 self loop)
__host__ __device__ __forceinline__ int mode_of_pos(int mi) { return mi == 0 ? 0 : (mi == 1 ? 2 : 1); }

// NST1: the tile is ONE stage (3 D <= 320 columns) — known at compile time so that the accumulators are dead while a tile is gathered
template <int NRT, int NCH, int UB, bool RELLDS, bool NST1>
__global__ __launch_bounds__(T4, 4) void layer_fused4_kernel(Args4 p) {
  constexpr int BM = NRT * 16;
  constexpr int CH = 32;                // slots served by one record chunk (lane i: slot cbase + i)
  constexpr int NRL = (BM + 31) / 32;   // self-loop rows per lane group
  extern __shared__ __attribute__((aligned(16))) unsigned char lds4[];
  const int piece = p.ncc * BM * 16;    // bytes of one bf16 piece of the image: [chunk column][row][16 B]
  float *epi = reinterpret_cast<float *>(lds4 + 3 * piece);   // [scale | shift] x OP4: the epilogue as one fma per value
  float *lrle = epi + 2 * OP4;                                // [loop_rel | loop_edge] x D (the self-loop rows' factors)
  float *rel_lds = lrle + 2 * p.d;                            // [rel_rows - 1][D] when RELLDS

  const int bid = int(blockIdx.x), nblk = int(gridDim.x);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  int row_lo = p.node0 + bid * p.rows_per_wg;                              // this workgroup's run of destinations
  int row_hi = row_lo + p.rows_per_wg < p.node1 ? row_lo + p.rows_per_wg : p.node1;
  if (p.bounds) {                                                          // ... or the caller's (work-balanced) run
    row_lo = p.node0 + p.bounds[bid];
    row_hi = p.node0 + p.bounds[bid + 1];
    row_hi = row_hi < p.node1 ? row_hi : p.node1;
  }
  const int myrows = row_hi > row_lo ? row_hi - row_lo : 0;
  const int nstage_ = NST1 ? 1 : p.nstage;
  // Stagger: every workgroup alternates gather (memory system) and multiply (matrix pipes) phases of about equal length, and all
  // start together: left alone, the whole chip gathers at once (measured: a gather phase then runs at the chip's 8 TB/s divided
  // by 256, i.e. it is bandwidth-bound, and the memory system idles while everyone multiplies). Workgroups of phase group ph > 0
  // cut their FIRST tile to ph / nphase of a (gather + multiply) period, so that from then on a group's gather phases meet the
  // other groups' multiply phases. Rows do not depend on the tiles that compute them.
  const int h0 = [&]() {
    const int ph = (bid >> 3) % p.nphase;           // (bid & 7 = XCD under round-robin placement: every XCD gets every group)
    if (ph == 0) return BM;
    int h = (2 * BM * ph + p.nphase * nstage_) / (2 * p.nphase * nstage_);   // BM * ph / (nphase * stages per tile), rounded
    h = (h + 8) / 16 * 16;
    return h < 16 ? 16 : (h > BM ? BM : h);
  }();
  const int my_tiles = myrows <= h0 ? (myrows > 0 ? 1 : 0) : 1 + (myrows - h0 + BM - 1) / BM;
  auto tile_off = [&](int it_) { return it_ == 0 ? 0 : h0 + (it_ - 1) * BM; };
  const int nstage = NST1 ? 1 : p.nstage, d = p.d, k_all = 3 * p.d;

  auto lds_barrier = [] () __attribute__((always_inline)) {   // orders LDS only: vector memory stays in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };

  // ---------------------------------------------------------------------------------------------- gather state
  const int grp = tid >> 5, lig = tid & 31;        // 32 lane groups of 32
  const int glane0 = lane & 32;
  const uint32_t ldx32 = uint32_t(p.ldx), d32 = uint32_t(p.d);
  struct RowPtrs { int a, b, c; };
  auto tile_rows16 = [&](int it_) {
    const int left = myrows - tile_off(it_), h = it_ == 0 ? h0 : BM;
    const int r = left < h ? left : h;
    return (r + 15) & ~15;
  };
  auto rp_of = [&](int it_, int mode_) {           // lane l: the tile's row pointers l, l + 32, l + 64 (clamped)
    const int32_t *rp = p.rowptr + int64_t(mode_) * (p.n + 1);
    const int row0 = row_lo + tile_off(it_);
    const int h = it_ == 0 ? h0 : BM;
    auto at = [&](int i) {
      int node = row0 + (i < h ? i : h);
      node = node < row_hi ? node : row_hi;
      return rp[node];
    };
    RowPtrs r;
    r.a = at(lig); r.b = at(lig + 32); r.c = at(lig + 64);
    return r;
  };
  auto rp_get = [&](const RowPtrs &r, int idx) {   // idx group-uniform, 0..BM: the tile's row pointer idx
    const int from = glane0 + (idx & 31);
    const int va = __shfl(r.a, from), vb = __shfl(r.b, from), vc = __shfl(r.c, from);
    return idx < 32 ? va : (idx < 64 ? vb : vc);
  };
  // The tile's rows are dealt to the 32 lane groups by WORK: group g takes the rows whose work prefix P(i) = slots before
  // row i + c * i falls into [g, g + 1) * P(rows) / 32 (c = cost of an empty row, raised with the tile's slot count so
  // that no group gets more than 31 rows). Every row's slots are summed by ONE group in slot order.
  // ONE partition per stage for both halves: work(row) = its in-half slots + its out-half slots (of the modes the stage holds).
  struct Part { int lo, hi; };                     // rows [lo, hi) of the tile
  auto partition = [&](const RowPtrs &ra, bool has_a, const RowPtrs &rb, bool has_b, int nr) {
    const int base_a = __shfl(ra.a, glane0), base_b = __shfl(rb.a, glane0);
    const int tot = (has_a ? rp_get(ra, nr) - base_a : 0) + (has_b ? rp_get(rb, nr) - base_b : 0);
    const int c = 2 > (tot >> 8) + 1 ? 2 : (tot >> 8) + 1;
    const int ptot = tot + c * nr;
    const int thr_lo = (grp * ptot) >> 5, thr_hi = ((grp + 1) * ptot) >> 5;
    int lo = 0, hi = 0;
    const int va[3] = {ra.a, ra.b, ra.c}, vb[3] = {rb.a, rb.b, rb.c};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = lig + 32 * k;
      const int pw = (has_a ? va[k] - base_a : 0) + (has_b ? vb[k] - base_b : 0) + c * i;
      const unsigned long long blo = __ballot(i < nr && pw < thr_lo), bhi = __ballot(i < nr && pw < thr_hi);
      lo += __popc(uint32_t(blo >> glane0));
      hi += __popc(uint32_t(bhi >> glane0));
    }
    Part q;
    q.lo = lo; q.hi = hi;
    return q;
  };
  auto rp_lane = [&](const RowPtrs &r, const Part &q) {   // lane l: row pointer lo + min(l, hi - lo)
    const int idx = q.lo + (lig < q.hi - q.lo ? lig : q.hi - q.lo);
    const int from = glane0 + (idx & 31);
    const int va = __shfl(r.a, from), vb = __shfl(r.b, from), vc = __shfl(r.c, from);
    return idx < 32 ? va : (idx < 64 ? vb : vc);
  };
  auto rec_chunk = [&](int cbeg, int end) {        // lane i: record of slot cbeg + i (clamped to the range's last slot)
    int4 r = make_int4(0, 0, 0, 0);
    if (end > cbeg) r = p.rec[(cbeg + lig < end) ? cbeg + lig : end - 1];
    return r;
  };
  // Which edge modes (0 in-half at K position 0, 1 out-half at K position 2) a stage holds columns of
  auto stage_has = [&](int s, int mi) {
    const int k0 = s * p.skb * 32, k1 = k0 + p.skb * 32;
    return mi * d < k1 && (mi + 1) * d > k0 && mi * d < k_all;
  };
  // A row of a segment: lane's float4 j covers columns col0 + 128 j + 4 lig of the mode; kq = its quad index in the image
  auto write_row = [&](int row, const float4 (&v)[NCH], const bool (&ok)[NCH], const int (&kq)[NCH]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (ok[j]) {
        uint32_t h[2], m[2], l[2];
        split3p(v[j].x, v[j].y, h[0], m[0], l[0]);
        split3p(v[j].z, v[j].w, h[1], m[1], l[1]);
        const int qc = kq[j] >> 1, frot = (qc >> 1) & 7;
        unsigned char *dst = lds4 + qc * BM * 16 + (kq[j] & 1) * 8 + ((row & ~15) + (((row & 15) + frot) & 15)) * 16;
        *reinterpret_cast<uint2 *>(dst) = make_uint2(h[0], h[1]);
        *reinterpret_cast<uint2 *>(dst + piece) = make_uint2(m[0], m[1]);
        *reinterpret_cast<uint2 *>(dst + 2 * piece) = make_uint2(l[0], l[1]);
      }
    }
  };

  // ---------------------------------------------------------------------------------------------- multiply state
  const int simd = wave & 3, wj = wave >> 2;
  const int r16 = lane & 15, gq = lane >> 4;
  const int ct = wj < 3 ? 3 * simd + wj : 12;
  const bool ct_ok = ct * 16 < p.o;
  auto wload = [&](u32x4 (&wv)[3], int g) __attribute__((always_inline)) {
    g = g < p.kb_total ? g : p.kb_total - 1;       // (the prefetch past the last k-block re-reads it)
    const u32x4 *base = p.wp + ((int64_t(g) * NT4 + ct) * 3) * 64 + lane;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) wv[pc] = base[pc * 64];
  };
  // the six products, small terms first: (w piece, a piece) = (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)
  auto six = [&](f32x4 &accv, const u32x4 (&wv)[3], const bf16x8 (&a)[3]) __attribute__((always_inline)) {
    constexpr int WP[6] = {0, 2, 1, 0, 1, 0}, AP[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int pr = 0; pr < 6; ++pr)
      accv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[WP[pr]]), a[AP[pr]], accv, 0, 0, 0);
  };
  auto frag = [&](bf16x8 (&a)[3], const unsigned char *ap, int rt) __attribute__((always_inline)) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) a[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ap + pc * piece + rt * 256));
  };
  f32x4 acc[NRT];
  u32x4 wn[3][3];          // three NAMES for two live weight triples (the k-block loop is unrolled by three)

  // ---------------------------------------------------------------------------------------------- the pipeline
  // What a gather phase needs before its first row load, fetched AHEAD of it: the row pointers go out before the previous
  // multiply phase's k-loop; the partition, both halves' first slot records and the self-loop rows right after that k-loop,
  // ahead of the epilogue's stores — so that a gather phase starts at its row loads with its records in registers.
  struct Pre {
    Part part;
    int myrp[2];
    int4 rec[2];
    float4 xs[NRL][NCH];
  };
  auto stage_k0 = [&](int s_) { return s_ * p.skb * 32; };
  auto stage_k1 = [&](int s_) { const int k = stage_k0(s_) + p.skb * 32; return k < k_all ? k : k_all; };
  auto loop_cols = [&](int s_, int &lc0, int &lc1) {   // columns [lc0, lc1) of the self-loop mode (K position 1) in stage s_
    const int k0 = stage_k0(s_), k1 = stage_k1(s_);
    lc0 = k0 > d ? k0 - d : 0;
    lc1 = k1 - d < d ? k1 - d : d;
  };
  auto fetch_rowptrs = [&](int it_, int s_, RowPtrs (&nrp)[2]) __attribute__((always_inline)) {
    nrp[0] = rp_of(it_, 0);          // (both halves always: a stage that holds no column of one ignores it)
    nrp[1] = rp_of(it_, 1);
  };
  auto prefetch = [&](int it_, int s_, const RowPtrs (&nrp)[2], Pre &q) __attribute__((always_inline)) {
    const int nr_ = tile_rows16(it_);
    const bool ha = stage_has(s_, 0), hb = stage_has(s_, 2);
    q.part = partition(nrp[0], ha, nrp[1], hb, nr_);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      q.myrp[e] = 0;
      q.rec[e] = make_int4(0, 0, 0, 0);
      if (e == 0 ? ha : hb) {
        q.myrp[e] = rp_lane(nrp[e], q.part);
        q.rec[e] = rec_chunk(__shfl(q.myrp[e], glane0), __shfl(q.myrp[e], glane0 + (q.part.hi - q.part.lo)));
      }
    }
    int lc0, lc1;
    loop_cols(s_, lc0, lc1);
    if (lc1 > lc0) {
      const int r0_ = row_lo + tile_off(it_);
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c_ = lc0 + 128 * j + 4 * lig;
        const int cc = c_ < lc1 ? c_ : lc0;
#pragma unroll
        for (int i = 0; i < NRL; ++i) {
          const int node = (r0_ + grp + 32 * i < row_hi) ? r0_ + grp + 32 * i : row_hi - 1;   // rows past the run: computed, never stored
          q.xs[i][j] = *reinterpret_cast<const float4 *>(p.x + int64_t(node) * p.ldx + cc);
        }
      }
    }
  };

  RowPtrs nrp[2] = {{0, 0, 0}, {0, 0, 0}};
  Pre pre;
  if (my_tiles > 0) fetch_rowptrs(0, 0, nrp);
  {  // one-time LDS set-up
    const int n16 = (3 * piece) >> 4;        // the image starts as zeros: columns past K are read against zero weights
    for (int i = tid; i < n16; i += T4) reinterpret_cast<uint4 *>(lds4)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < OP4) {
      const int c = tid;
      const bool in = c < p.o;
      const float inv = in ? __builtin_amdgcn_rsqf(p.bn_var[c] + p.bn_eps) * p.bn_gamma[c] : 0.f;
      constexpr float third = 1.0f / 3.0f;   // (sum of the three modes) / 3, model.py:103, as a multiplication (<= 1 ulp)
      epi[c] = inv * third;
      epi[OP4 + c] = in ? ((p.bias ? p.bias[c] : 0.f) - p.bn_mean[c]) * inv + p.bn_beta[c] : 0.f;
    }
    for (int i = tid; i < 2 * d; i += T4) lrle[i] = i < d ? p.loop_rel[i] : p.loop_edge[i - d];
    if (RELLDS) {
      const int n4 = ((p.rel_rows - 1) * p.d) >> 2;
      for (int i = tid; i < n4; i += T4) reinterpret_cast<float4 *>(rel_lds)[i] = reinterpret_cast<const float4 *>(p.rel)[i];
    }
  }
  if (my_tiles > 0) prefetch(0, 0, nrp, pre);
  STAMP4(0);
  __syncthreads();
  STAMP4(1);

  for (int it = 0; it < my_tiles; ++it) {
    const int r0 = row_lo + tile_off(it);
    const int nr = tile_rows16(it);
    const int nrt_eff = nr >> 4;
    for (int s = 0; s < nstage; ++s) {
      const int k0 = stage_k0(s);
      const int k1 = stage_k1(s);
      // ========================================================================================== G: gather the stage
      STAMP4(2 + 8 * (it * nstage + s));
      {  // self loop: (x * loop_rel) * loop_edge, model.py:91-94,101; group g owns rows g, g + 32, g + 64 of the tile
        int lc0, lc1;
        loop_cols(s, lc0, lc1);
        if (lc1 > lc0) {
          float4 lrv[NCH], lev[NCH];
          bool lok[NCH];
          int lkq[NCH];
#pragma unroll
          for (int j = 0; j < NCH; ++j) {
            const int c_ = lc0 + 128 * j + 4 * lig;
            lok[j] = c_ < lc1;
            const int cc = lok[j] ? c_ : lc0;
            lkq[j] = (d + c_ - k0) >> 2;
            lrv[j] = *reinterpret_cast<const float4 *>(lrle + cc);
            lev[j] = *reinterpret_cast<const float4 *>(lrle + d + cc);
          }
#pragma unroll
          for (int i = 0; i < NRL; ++i) {
            if (grp + 32 * i < nr) {
              float4 v[NCH];
#pragma unroll
              for (int j = 0; j < NCH; ++j) v[j] = f4mul4(f4mul4(pre.xs[i][j], lrv[j]), lev[j]);
              write_row(grp + 32 * i, v, lok, lkq);
            }
          }
        }
      }
      STAMP4(3 + 8 * (it * nstage + s));
      // edge segments: in-half (K position 0), then out-half (K position 2); this group's rows [e_lo, e_hi) in both
      const int e_lo = pre.part.lo, e_hi = pre.part.hi, e_n = e_hi - e_lo;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int mi = 2 * e, mode = e;
        const int c0 = (k0 > mi * d ? k0 - mi * d : 0), c1 = (k1 - mi * d < d ? k1 - mi * d : d);
        if (c1 <= c0) continue;
        const int myrp = pre.myrp[e];
        const int beg = __shfl(myrp, glane0), end = __shfl(myrp, glane0 + e_n);
        int4 myrec = pre.rec[e];
        const int ee_sub_mode = p.ee_sub[mode];
        int2 myhub = make_int2(-1, 0);                                  // lane i: hub chunks of destination e_lo + i
        {
          const int node = r0 + e_lo + lig;
          if (p.hubinfo && lig < e_n && node < row_hi) myhub = p.hubinfo[int64_t(mode) * p.n + node];
        }
        bool ok[NCH];
        int coff[NCH], kq[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const int c_ = c0 + 128 * j + 4 * lig;
          ok[j] = c_ < c1;
          coff[j] = ok[j] ? c_ : c0;     // lanes past the segment repeat its first columns and store nothing
          kq[j] = (mi * d + c_ - k0) >> 2;
        }
        const float *relbase = RELLDS ? rel_lds : p.rel;
        int row = e_lo, nb = __shfl(myrp, glane0 + 1);
        float4 sum[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) sum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        auto flush = [&]() __attribute__((always_inline)) {   // the run of destination `row` is complete (group-uniform)
          if (p.hubinfo) {     // a hub's own run is empty: its folded total sits in the row of its first chunk
            const int first = __shfl(myhub.x, glane0 + (row - e_lo)), hcnt = __shfl(myhub.y, glane0 + (row - e_lo));
            if (hcnt > 0) {
#pragma unroll
              for (int j = 0; j < NCH; ++j) {
                const float4 ps = *reinterpret_cast<const float4 *>(p.partial + int64_t(first - p.chunk0) * p.d + coff[j]);
                sum[j] = make_float4(sum[j].x + ps.x, sum[j].y + ps.y, sum[j].z + ps.z, sum[j].w + ps.w);
              }
            }
          }
          write_row(row, sum, ok, kq);
#pragma unroll
          for (int j = 0; j < NCH; ++j) sum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          ++row;
        };
        int cbase = beg;                                   // first slot of the record chunk held in myrec
        for (int sl = beg; sl < end; sl += UB) {
          if (sl >= cbase + CH) {                          // group-uniform: next record chunk of a long range
            cbase += CH;
            myrec = rec_chunk(cbase, end);
          }
          // issue: only the row data stays in registers (8 per slot and float4 column); type and norm are shuffled out of the
          // record chunk again when a slot is consumed
          float4 xv[UB][NCH], rv[UB][NCH], ev[UB][NCH];
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            const int slot = (sl + u < end) ? sl + u : end - 1;
            const int from = glane0 + (slot - cbase);
            const uint32_t src = uint32_t(__shfl(myrec.x, from));
            const uint32_t erow = uint32_t(slot - ee_sub_mode);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
              xv[u][j] = *reinterpret_cast<const float4 *>(p.x + coff[j] + uint64_t(src) * ldx32);
              if (!RELLDS) rv[u][j] = *reinterpret_cast<const float4 *>(p.rel + coff[j] + uint64_t(uint32_t(__shfl(myrec.y, from))) * d32);
              ev[u][j] = *reinterpret_cast<const float4 *>(p.ee + coff[j] + uint64_t(erow) * d32);
            }
          }
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            if (sl + u < end) {
              while (sl + u >= nb) {
                flush();
                nb = __shfl(myrp, glane0 + (row - e_lo) + 1);
              }
              const int from = glane0 + (sl + u - cbase);
              const float wgt = __int_as_float(__shfl(myrec.z, from));
              const uint32_t typ = uint32_t(__shfl(myrec.y, from));
#pragma unroll
              for (int j = 0; j < NCH; ++j) {
                const float4 rr = RELLDS ? *reinterpret_cast<const float4 *>(relbase + coff[j] + typ * d32) : rv[u][j];
                sum[j] = f4axpy4(sum[j], f4mul4(f4mul4(xv[u][j], rr), ev[u][j]), wgt);
              }
            }
          }
        }
        while (row < e_hi) flush();  // last run, then zero rows for destinations without slots
        STAMP4(4 + e + 8 * (it * nstage + s));
      }
      // this stage's first weights go out before the barrier (they do not depend on the image)
      const int gbase = s * p.skb;
      const int skb_s = (p.kb_total - gbase < p.skb) ? p.kb_total - gbase : p.skb;
      if (ct_ok) wload(wn[0], gbase);
      // ... and so do the next gather phase's row pointers
      const bool last_stage = s + 1 == nstage;
      // (after the last phase: the same tile once more — unconditional, so that nothing of the previous phase stays live)
      const int nit = (last_stage && it + 1 < my_tiles) ? it + 1 : it, ns = last_stage ? 0 : s + 1;
      fetch_rowptrs(nit, ns, nrp);
      lds_barrier();                 // B1: the stage's image is complete
      STAMP4(6 + 8 * (it * nstage + s));
      if (s == 0) {                  // (zeroed here, not before the gather: the accumulators are dead while a tile's first stage is gathered)
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // ========================================================================================== M: multiply the stage
      if (ct_ok) {
        auto kblock = [&](const u32x4 (&wv)[3], u32x4 (&nx)[3], int kb) __attribute__((always_inline)) {
          wload(nx, gbase + kb + 1);                 // the k-block after this one
          int qc = 4 * kb + gq;
          qc = qc < p.ncc ? qc : p.ncc - 1;          // columns past the image (last k-block): finite values, zero weights
          const unsigned char *ap = lds4 + (qc * BM + ((r16 + ((qc >> 1) & 7)) & 15)) * 16;
          if (wj < 3) {
            if (nrt_eff == NRT) {   // every row tile exists: the fragments of row tile rt + 1 are read before the MFMAs of row tile rt
              bf16x8 a[2][3];
              frag(a[0], ap, 0);
#pragma unroll
              for (int rt = 0; rt < NRT; ++rt) {
                if (rt + 1 < NRT) frag(a[(rt + 1) & 1], ap, rt + 1);
                asm volatile("" ::: "memory");
                six(acc[rt], wv, a[rt & 1]);
              }
            } else {
#pragma unroll
              for (int rt = 0; rt < NRT; ++rt) {
                if (rt < nrt_eff) {
                  bf16x8 a[3];
                  frag(a, ap, rt);
                  six(acc[rt], wv, a);
                }
              }
            }
          } else {                  // the 13th column tile: row tiles simd, simd + 4 (accumulators 0, 1)
#pragma unroll
            for (int u = 0; u < (NRT + 3) / 4; ++u) {
              const int rt = simd + 4 * u;
              if (rt < nrt_eff) {
                bf16x8 a[3];
                frag(a, ap, rt);
                six(acc[u], wv, a);
              }
            }
          }
        };
        int kb = 0;
        for (; kb + 3 <= skb_s; kb += 3) {
          kblock(wn[0], wn[1], kb);
          kblock(wn[1], wn[2], kb + 1);
          kblock(wn[2], wn[0], kb + 2);
        }
        if (kb < skb_s) {
          kblock(wn[0], wn[1], kb);
          if (kb + 1 < skb_s) kblock(wn[1], wn[2], kb + 1);
        }
      }
      STAMP4(7 + 8 * (it * nstage + s));
      // the next gather phase's partition, records and self-loop rows: in flight across the barrier and the epilogue
      prefetch(nit, ns, nrp, pre);
      STAMP4(8 + 8 * (it * nstage + s));
      lds_barrier();                 // B2: the image is free again
      STAMP4(9 + 8 * (it * nstage + s));
    }
    // epilogue tanh(acc * scale + shift) (model.py:103-106); lane holds out[row = 16 rt + r16][16 ct + 4 gq .. + 3]
    if (ct_ok) {
      const int col = ct * 16 + 4 * gq;
      if (col < p.o) {
        const float4 sc = *reinterpret_cast<const float4 *>(epi + col), sh = *reinterpret_cast<const float4 *>(epi + OP4 + col);
        auto store_unit = [&](f32x4 a, int node) __attribute__((always_inline)) {
          if (node < row_hi) {
            const float4 v = make_float4(tanh4_(fmaf(a[0], sc.x, sh.x)), tanh4_(fmaf(a[1], sc.y, sh.y)),
                                         tanh4_(fmaf(a[2], sc.z, sh.z)), tanh4_(fmaf(a[3], sc.w, sh.w)));
            *reinterpret_cast<float4 *>(p.out + int64_t(node - p.node0) * p.ldo + col) = v;
          }
        };
        if (wj < 3) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) store_unit(acc[rt], r0 + rt * 16 + r16);
        } else {
#pragma unroll
          for (int u = 0; u < (NRT + 3) / 4; ++u) store_unit(acc[u], r0 + (simd + 4 * u) * 16 + r16);
        }
      }
    }
  }
  STAMP4(62);
  // all_rel = rel @ rels_weight (model.py:107). One item = one relation row x 16 columns per wave: the four 16-lane
  // groups run the four K quarters of small_matmul_kernel's arithmetic (sequential fmaf chains), the partial sums are
  // added in quarter order — values bit-identical to the separate launch.
  if (p.rel_out) {
    const int rows = p.rel_rows - 1, k = p.d, n = p.o;
    const int ncg = (n + 15) / 16, items = rows * ncg;
    const int kper = (k + 3) / 4;
    const int qd = lane >> 4;
    const int kq0 = qd * kper, kq1 = (kq0 + kper < k) ? kq0 + kper : k;
    for (int item = wave * nblk + bid; item < items; item += nblk * 16) {
      const int row = item / ncg, col = (item - row * ncg) * 16 + (lane & 15);
      const bool ok = col < n;
      const float *ap = p.rel + int64_t(row) * k;
      const float *bp = p.rw + (ok ? col : 0);
      float a = 0.f;
      constexpr int UR = 32;
      for (int i0 = 0; i0 < kper; i0 += UR) {
        float av[UR], bv[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
          const int kk = kq0 + i0 + u;
          const int kc = (i0 + u < kper && kk < kq1) ? kk : 0;
          av[u] = ap[kc];
          bv[u] = bp[int64_t(kc) * n];
        }
#pragma unroll
        for (int u = 0; u < UR; ++u) {
          const int kk = kq0 + i0 + u;
          if (i0 + u < kper && kk < kq1) a = fmaf(av[u], bv[u], a);
        }
      }
      const float q1 = __shfl(a, (lane & 15) + 16), q2 = __shfl(a, (lane & 15) + 32), q3 = __shfl(a, (lane & 15) + 48);
      if (qd == 0 && ok) p.rel_out[int64_t(row) * n + col] = ((a + q1) + q2) + q3;
    }
  }
  STAMP4(63);
}

// wp[((g * NT4 + ct) * 3 + piece) * 64 + lane] = 8 bf16: Wk[32 g + 8 (lane >> 4) + i][16 ct + (lane & 15)], i = 0..7, zero
// outside; Wk = the stacked weights' rows in the kernel's K order [in-half | self loop | out-half] (the stacked matrix the
// caller passes is [W_in; W_out; W_loop], model.py:116 by mode).
__global__ __launch_bounds__(256) void pack4_kernel(const float *__restrict__ w, u32x4 *__restrict__ wp, int d, int o, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, piece = (idx >> 6) % 3, ct = ((idx >> 6) / 3) % NT4, g = (idx >> 6) / (3 * NT4);
  const int col = ct * 16 + (lane & 15), k0 = 32 * g + 8 * (lane >> 4);
  uint32_t bits[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + 2 * i + j;
      const int mi = k / d, c = k - mi * d;
      v[j] = (k < 3 * d && col < o) ? w[(int64_t(mode_of_pos(mi)) * d + c) * o + col] : 0.f;
    }
    uint32_t h, m, l;
    split3p(v[0], v[1], h, m, l);
    bits[i] = piece == 0 ? h : piece == 1 ? m : l;
  }
  wp[idx] = u32x4{bits[0], bits[1], bits[2], bits[3]};
}

struct Shape4 {
  int kb_total, nstage, skb, ncc, nch;
};
Shape4 shape4(int d) {
  Shape4 s;
  s.kb_total = (3 * d + 31) / 32;
  s.nstage = (s.kb_total + SKB4 - 1) / SKB4;
  s.skb = (s.kb_total + s.nstage - 1) / s.nstage;          // stages of equal length (10 + 9 for D = 200, 8 + 8 + 8 for D = 256)
  const int wmax = 3 * d < s.skb * 32 ? 3 * d : s.skb * 32;  // widest stage in columns
  s.ncc = (wmax + 7) / 8;
  s.nch = d > 128 ? 2 : 1;
  return s;
}

constexpr size_t LDS_MAX4 = size_t(160) * 1024;

size_t lds_bytes4(const Shape4 &s, int d, int nrt, size_t rel_bytes) {
  return size_t(3) * s.ncc * (nrt * 16) * 16 + size_t(2) * OP4 * 4 + size_t(2) * d * 4 + rel_bytes;
}

template <int NRT, int NCH, int UB, bool RELLDS, bool NST1>
int launch4(const Args4 &p, int grid, size_t lds, hipStream_t st) {
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused4_kernel<NRT, NCH, UB, RELLDS, NST1>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_MAX4)) != hipSuccess)
    return mgcn::fail(MGCN_ELAUNCH, "layer_fused4: cannot reserve %zu bytes of LDS", LDS_MAX4);
  hipLaunchKernelGGL((layer_fused4_kernel<NRT, NCH, UB, RELLDS, NST1>), dim3(unsigned(grid)), dim3(T4), lds, st, p);
  MGCN_CHECK_LAUNCH("layer_fused4_kernel");
  return MGCN_OK;
}

template <int NRT, int NCH, int UB>
int launch4_rel(const Args4 &p, int grid, size_t lds, bool rel_lds, hipStream_t st) {
  if (NCH == 1 && p.nstage == 1) {   // (one stage needs 3 D <= 320, i.e. one float4 per lane and row)
    if (rel_lds) return launch4<NRT, 1, UB, true, true>(p, grid, lds, st);
    return launch4<NRT, 1, UB, false, true>(p, grid, lds, st);
  }
  if (rel_lds) return launch4<NRT, NCH, UB, true, false>(p, grid, lds, st);
  return launch4<NRT, NCH, UB, false, false>(p, grid, lds, st);
}

template <int NCH, int UB>
int launch4_nrt(const Args4 &p, int nrt, int grid, size_t lds, bool rel_lds, hipStream_t st) {
  if (nrt == 3) return launch4_rel<3, NCH, UB>(p, grid, lds, rel_lds, st);
  if (nrt == 4) return launch4_rel<4, NCH, UB>(p, grid, lds, rel_lds, st);
  return launch4_rel<5, NCH, UB>(p, grid, lds, rel_lds, st);
}

#ifdef MGCN_DIAG
unsigned long long *diag_buf4() {
  static unsigned long long *buf = nullptr;
  if (!buf) {
    if (hipMalloc(&buf, 1024 * 16 * 64 * 8) != hipSuccess) buf = nullptr;
    else (void)hipMemset(buf, 0, 1024 * 16 * 64 * 8);
  }
  return buf;
}
#endif

}  // namespace

#ifdef MGCN_DIAG
extern "C" int mgcn_diag_fused4(unsigned long long *host_out) {   // [1024][16][64] of the LAST generation-4 launch (diagnostics build)
  unsigned long long *b = diag_buf4();
  if (!b) return 1;
  return hipMemcpy(host_out, b, 1024 * 16 * 64 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif

namespace mgcn {

bool fused4_takes(int32_t dim_in, int32_t dim_out) {
  return dim_in > 0 && dim_in % 4 == 0 && dim_in <= 256 && dim_out > 0 && dim_out % 4 == 0 && dim_out <= 208;
}

size_t fused4_packed_bytes(int32_t dim_in, int32_t dim_out) {
  (void)dim_out;
  return size_t(shape4(dim_in).kb_total) * NT4 * 3 * 64 * 16;
}

int fused4_pack(int32_t dim_in, int32_t dim_out, const float *w_dev, void *wp_dev, void *stream) {
  const Shape4 s = shape4(dim_in);
  const int total = s.kb_total * NT4 * 3 * 64;
  hipLaunchKernelGGL(pack4_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w_dev,
                     reinterpret_cast<u32x4 *>(wp_dev), dim_in, dim_out, total);
  MGCN_CHECK_LAUNCH("pack4_kernel");
  return MGCN_OK;
}

// tune: 0 = automatic; bits 0-3 row tiles per tile (3 / 4 / 5), bits 4-7 slots per gather batch (2 or, one float4 per lane and row, 4; 0 = default), bits 8-9
// relation table in LDS (1 = never), bits 12-13 phase groups of the stagger (1 = none, 2 = two, 3 = four; 0 = two): for A/B runs,
// never needed for correctness.
int fused4_launch(int64_t num_nodes, int32_t dim_in, int32_t dim_out, int32_t num_rel_rows, const int32_t *rowptr_dev,
                  const mgcn_edge_rec *rec_dev, const float *x_dev, int64_t ldx, const float *rel_dev,
                  const float *loop_rel_dev, const float *ee_dev, const float *loop_edge_dev, const void *wp_dev,
                  const float *bias_dev, const float *bn_mean_dev, const float *bn_var_dev, const float *bn_gamma_dev,
                  const float *bn_beta_dev, float bn_eps, float *out_dev, int64_t ldo, int64_t node_begin,
                  int64_t node_end, int64_t ee_sub_in, int64_t ee_sub_out, const int32_t *hubinfo_dev, int64_t chunk_begin,
                  const float *partial_dev, const float *rels_weight_dev, float *rel_out_dev, const int32_t *row_bounds_dev,
                  int32_t num_row_bounds, int32_t tune, void *stream) {
  const Shape4 s = shape4(dim_in);
  Args4 p = {};
  p.rowptr = rowptr_dev; p.rec = reinterpret_cast<const int4 *>(rec_dev);
  p.x = x_dev; p.rel = rel_dev; p.loop_rel = loop_rel_dev; p.ee = ee_dev; p.loop_edge = loop_edge_dev;
  p.wp = reinterpret_cast<const u32x4 *>(wp_dev);
  p.bias = bias_dev; p.bn_mean = bn_mean_dev; p.bn_var = bn_var_dev; p.bn_gamma = bn_gamma_dev; p.bn_beta = bn_beta_dev;
  p.out = out_dev; p.ldx = ldx; p.ldo = ldo;
  p.n = int32_t(num_nodes); p.d = dim_in; p.o = dim_out; p.rel_rows = num_rel_rows;
  p.node0 = int32_t(node_begin); p.node1 = int32_t(node_end);
  p.ee_sub[0] = int32_t(ee_sub_in); p.ee_sub[1] = int32_t(ee_sub_out);
  p.hubinfo = reinterpret_cast<const int2 *>(hubinfo_dev); p.partial = partial_dev; p.chunk0 = int32_t(chunk_begin);
  p.rw = rel_out_dev ? rels_weight_dev : nullptr; p.rel_out = rel_out_dev;
  p.kb_total = s.kb_total; p.nstage = s.nstage; p.skb = s.skb; p.ncc = s.ncc;
  p.bn_eps = bn_eps;
#ifdef MGCN_DIAG
  p.diag = diag_buf4();
#endif
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  // one contiguous run of rows per workgroup, a multiple of 16; one workgroup per CU
  const int64_t nrows = node_end - node_begin;
  int64_t rpw = ((nrows + cus - 1) / cus + 15) / 16 * 16;
  if (rpw < 16) rpw = 16;
  int grid = int(nrows > 0 ? (nrows + rpw - 1) / rpw : 1);
  p.rows_per_wg = int32_t(rpw);
  if (row_bounds_dev && num_row_bounds > 0 && nrows > 0) {   // the caller's runs, one workgroup each
    p.bounds = row_bounds_dev;
    grid = num_row_bounds;
    const int64_t per = (nrows + grid - 1) / grid;            // (GraphCSR.workgroup_bounds caps its runs by the same rule)
    rpw = per <= 80 ? (per + 15) / 16 * 16 : (per + 79) / 80 * 80;         // (the longest run the convention allows: whole 80-row tiles)
  }
  const size_t rel_bytes = rel_dev ? size_t(num_rel_rows - 1) * dim_in * 4 : 0;
  const int t_nrt = tune & 15, t_ub = (tune >> 4) & 15, t_rel = (tune >> 8) & 3, t_ph = (tune >> 12) & 3;
  p.nphase = t_ph == 1 ? 1 : (t_ph == 3 ? 4 : 2);
  const int nrt_cap = rpw >= 80 ? 5 : rpw >= 64 ? 4 : 3;     // (a tile taller than the run is pointless)
  const bool rel_wanted = rel_bytes > 0 && rel_bytes <= size_t(32) * 1024 && t_rel != 1;
  auto fits = [&](int a, bool r) { return lds_bytes4(s, dim_in, a, r ? rel_bytes : 0) <= LDS_MAX4; };
  int nrt = 0;
  bool rel_lds = false;
  if (t_nrt) {
    nrt = t_nrt < nrt_cap ? t_nrt : nrt_cap;
    rel_lds = rel_wanted && fits(nrt, true);
    if (nrt < 3 || !fits(nrt, rel_lds)) return mgcn::fail(MGCN_EINVAL, "layer_fwd_fused: tune %d does not fit the LDS", tune);
  } else {
    // the relation table in LDS (a third of the gather's row loads) is worth a row tile; then the tallest tile that fits
    for (int want_rel = rel_wanted ? 1 : 0; want_rel >= 0 && !nrt; --want_rel) {
      for (int a = nrt_cap; a >= (want_rel ? 4 : 3) && a >= 3 && !nrt; --a) {
        if (fits(a, want_rel != 0)) { nrt = a; rel_lds = want_rel != 0; }
      }
    }
    if (!nrt) return mgcn::fail(MGCN_EUNSUPPORTED, "layer_fwd_fused: no tile geometry fits the LDS (D=%d O=%d)", dim_in, dim_out);
  }
  const size_t lds = lds_bytes4(s, dim_in, nrt, rel_lds ? rel_bytes : 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (s.nch == 1) {
    if (t_ub == 2) return launch4_nrt<1, 2>(p, nrt, grid, lds, rel_lds, st);
    return launch4_nrt<1, 4>(p, nrt, grid, lds, rel_lds, st);
  }
  return launch4_nrt<2, 2>(p, nrt, grid, lds, rel_lds, st);
}

}  // namespace mgcn
