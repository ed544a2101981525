// Fused layer forward (eval), third generation — aggregation + dense step + epilogue in ONE launch (gfx950);
// replaces model.py:29-30, 99-107, 111-118 for one run of destination rows per workgroup.
//
// One 1024-thread workgroup per CU (16 waves x 128 VGPRs), each owning ONE CONTIGUOUS run of destination rows
// (rows_per_wg, a multiple of 16: 40 943 rows on 256 CUs = 160 rows each), walked in tiles of BM = 16 * NRT rows; the
// last tile of a run may be shorter (its absent row tiles are skipped by both roles), so no tile height loses rows to
// quantisation. Two roles:
//   waves 8-15  GATHER    sixteen 32-lane groups; a lane holds NCH float4 of a row (columns 4 * lane and, NCH = 2,
//               128 + 4 * lane): ONE slot walk covers 128 * NCH columns of the layer input, so a 200-wide layer is
//               walked once (the second generation walked every slot twice, 512 + 288 bytes of each row). The tile's
//               rows are dealt to the groups by work; a group sums its rows' slots in slot order (the sums of
//               agg_fwd_kernel) and writes each finished row, split exactly into three bf16 pieces, to an LDS image.
//   waves 0-7   MULTIPLY  v_mfma_f32_16x16x32_bf16 on the six significant products of the split operands (f32-faithful:
//               see split3 below / DESIGN.md), weights pre-split and pre-packed, streamed from L2 one k-block ahead;
//               epilogue tanh(acc * scale + shift) (model.py:103-106) on the accumulators.
// Stage = (mode, pass of 128 * NCH columns). The gather role is the launch's critical path (its waves run ~15 cycles per
// instruction beside the MFMA waves' LDS traffic), the multiply role has slack: so the gather only STORES finished f32
// rows into a staging buffer (one ds_write_b128 per lane and row), and the MFMA waves split each staged tile into the
// three bf16 pieces (a branch-free pass, 2 rows per wave-instruction) before multiplying it. Buffers are exactly as
// wide as the pass (ceil(width / 8) 16-byte chunk columns). The roles are coupled by a ring of `nimg` staging buffers
// with two LDS counters each (rows staged / tile converted), not by a workgroup barrier: the gather runs up to nimg
// stages ahead, a fast wave never waits for a slow one of the other role, a tile's epilogue overlaps the next tile's
// gather. The MFMA waves order their own convert / multiply phases with two more counters. Every spin is bounded.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mgcn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int T3 = 1024;
constexpr int SPIN_LIMIT = 1 << 22;   // bounded spins: a protocol error (or a wave parked by a debugger / preemption for ~0.1 s)
                                      // never hangs a wave; it is REPORTED through Args3::status, the caller's status word

struct Args3 {
  const int32_t *rowptr;
  const int4 *rec;
  const float *x, *rel, *loop_rel, *ee, *loop_edge;
  const u32x4 *wp;        // packed weights [G][NT][3][64] (8 bf16 per lane), pack3_kernel
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
  int32_t npass, nkb_last, kbp, kbm, G;   // passes per mode, k-blocks of the last pass / of a full pass / per mode / per tile
  int32_t ncc;            // 16-byte chunk columns of a stage image (8 bf16 each)
  int32_t rows_per_wg, nimg;
  uint32_t *status;        // optional: bit 0 is set when a bounded spin below ran out (the launch's rows are then invalid)
  const int32_t *bounds;   // [grid + 1] row offsets from node0 of the workgroups' runs (work-balanced), or null: equal runs
  float bn_eps;
#ifdef MGCN_DIAG
  unsigned long long *diag;   // [grid][16 waves][4]: cycles in the kernel, cycles waiting for an image, waits, stages
#endif
};
#ifdef MGCN_DIAG
#define DIAG_NOW() __builtin_readcyclecounter()
#define DIAG_LAP(acc) do { const unsigned long long t_ = DIAG_NOW(); acc += t_ - t_last; t_last = t_; } while (0)
#else
#define DIAG_NOW() 0ull
#define DIAG_LAP(acc) do {} while (0)
#endif

__device__ __forceinline__ float tanh3_(float v) {   // exp2 + rcp, 7 VALU per value
  const float t = __builtin_amdgcn_exp2f(fabsf(v) * -2.885390081777927f);
  return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), v);
}

// Exact three-way split of two f32 values into bf16 pieces, packed {even, odd}: hi = bf16(v) (round to nearest even),
// mid = bf16(v - hi), lo = v - hi - mid. Each difference is exact (v - hi has at most 16 significant bits, the next
// one at most 8), so hi + mid + lo == v bit for bit for finite v. Rounding (not truncating) keeps every residual at
// most HALF an ulp of the piece above it, with either sign: the cross terms the multiply drops (mid x lo, lo x mid, lo x
// lo) are below 2^-26 |a| |w| and unbiased, where a truncating split leaves 2^-24 with the sign of the product
// (tests/test_gpu_round3.py feeds rows with 2^40 of dynamic range). A non-finite v gives NaN pieces: the output row is
// NaN where the exact-f32 path may give +-1 (documented in DESIGN.md).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3p(float v0, float v1, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v0, v1}, bf16x2));            // v_cvt_pk_bf16_f32
  const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
  m = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}
__device__ __forceinline__ float4 f4mul3(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4axpy(float4 s, float4 m, float w) {
  return make_float4(s.x + m.x * w, s.y + m.y * w, s.z + m.z * w, s.w + m.w * w);
}

// LDS counters: [0..3] rows staged per staging buffer, [4..7] buffer converted (free again), [8] one-time tables ready,
// [9] MFMA waves that have converted their share of the current stage, [10] MFMA waves done multiplying a stage
__device__ __forceinline__ void wait_ge_(const uint32_t *c, uint32_t need, uint32_t *status) {
  int spins = 0;
#pragma nounroll
  for (; spins < SPIN_LIMIT; ++spins) {
    const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (v >= need) break;
    __builtin_amdgcn_s_sleep(1);
  }
  // A spin that ran out: the wave goes on (it must, or its partners spin out too) and the tile it touches is garbage — say so.
  if (spins == SPIN_LIMIT && status) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void signal_add(uint32_t *c, int lane) {
  asm volatile("" ::: "memory");   // the wave's LDS operations are issued in program order; LDS executes them in order
  if (lane == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

template <int NT, int NRT, int NCH, bool RELLDS>
__global__ __launch_bounds__(T3, 4) void layer_fused3_kernel(Args3 p) {
  constexpr int BM = NRT * 16;
  constexpr int UB = 4 / NCH;           // slots per gather batch: 8 row loads of 16 B per lane in flight either way
  constexpr int CH = 32;                // slots served by one record chunk (lane i: slot cbase + i)
  constexpr int OP = NT * 16;           // padded output width
  extern __shared__ __attribute__((aligned(16))) unsigned char lds3[];
  const int piece = p.ncc * BM * 16;    // bytes of one bf16 piece of the image: [chunk column][row][16 B]
  const int rowb = p.ncc * 32;          // bytes of one staged f32 row (8 floats per chunk column)
  const int sbuf = BM * rowb;           // one staging buffer: [row][column] f32
  unsigned char *stg0 = lds3 + 3 * piece;
  uint32_t *cnt = reinterpret_cast<uint32_t *>(stg0 + p.nimg * sbuf);
  float *epi = reinterpret_cast<float *>(cnt + 16);   // [scale | shift] x OP: the epilogue as one fma per value
  float *rel_lds = epi + 2 * OP;                      // [rel_rows - 1][D] when RELLDS

  const int bid = int(blockIdx.x), nblk = int(gridDim.x);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nimg = p.nimg;
  int row_lo = p.node0 + bid * p.rows_per_wg;                              // this workgroup's run of destinations
  int row_hi = row_lo + p.rows_per_wg < p.node1 ? row_lo + p.rows_per_wg : p.node1;
  if (p.bounds) {                                                          // ... or the caller's (work-balanced) run
    row_lo = p.node0 + p.bounds[bid];
    row_hi = p.node0 + p.bounds[bid + 1];
    row_hi = row_hi < p.node1 ? row_hi : p.node1;
  }
  const int myrows = row_hi > row_lo ? row_hi - row_lo : 0;
  const int my_tiles = (myrows + BM - 1) / BM;
  const int npass = p.npass;

  if (tid < 16) cnt[tid] = 0;
  __syncthreads();   // the only workgroup barrier: nothing is in flight yet
#ifdef MGCN_DIAG
  const unsigned long long t_begin = DIAG_NOW();
  unsigned long long t_wait = 0, n_wait = 0, t_load = 0, n_load = 0, t_last = t_begin, t_a = 0, t_b = 0, t_d = 0, t_o = 0;
  auto wait_ge = [&](const uint32_t *c, uint32_t need) __attribute__((always_inline)) {
    const unsigned long long t0 = DIAG_NOW();
    wait_ge_(c, need, p.status);
    t_wait += DIAG_NOW() - t0;
    ++n_wait;
  };
  auto diag_end = [&]() __attribute__((always_inline)) {
    if (p.diag && lane == 0) {
      unsigned long long *d = p.diag + (int64_t(bid) * 16 + wave) * 8;
      d[0] = DIAG_NOW() - t_begin; d[1] = t_wait; d[2] = t_load; d[3] = n_load; d[4] = t_a; d[5] = t_b; d[6] = t_d; d[7] = t_o;
    }
  };
#else
  auto wait_ge = [&](const uint32_t *c, uint32_t need) __attribute__((always_inline)) { wait_ge_(c, need, p.status); };
  auto diag_end = [] () {};
#endif

  if (wave >= 8) {
    // ------------------------------------------------------------------------------------------ GATHER
    // The gather waves win the issue arbitration over the SIMD's MFMA waves (s_setprio 3: user priority, default 0): their
    // row loads go out as soon as their operands are there instead of queueing behind MFMAs that have a whole stage to
    // complete. WN18RR step 0.157 -> 0.152 ms, FB15k-237 0.325 -> 0.317, configs[4] slice 512 -> 512 5.48 -> 5.30 ms (same-box
    // A/B); priority for the MFMA waves instead: no change; raised only while a batch's loads are issued: half the gain.
    __builtin_amdgcn_s_setprio(3);
    const int gtid = tid - 512;
    const int grp = gtid >> 5, lig = gtid & 31;
    const int glane0 = lane & 32;
    const int qcol = lig >> 1;
    const uint32_t ldx32 = uint32_t(p.ldx), d32 = uint32_t(p.d);
    bool wr[NCH];                                           // this lane's chunk column j exists in the image
#pragma unroll
    for (int j = 0; j < NCH; ++j) wr[j] = 16 * j + qcol < p.ncc;

    auto write_row = [&](unsigned char *stg, int row, const float4 (&v)[NCH], const bool (&ok)[NCH]) __attribute__((always_inline)) {
      unsigned char *dst = stg + row * rowb + lig * 16;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        if (wr[j]) *reinterpret_cast<float4 *>(dst + j * 512) = ok[j] ? v[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    // ring position of the next stage this wave fills
    int s_img = 0;
    uint32_t s_round = 0;
    auto acquire = [&]() __attribute__((always_inline)) {
      if (s_round > 0) {   // all 8 MFMA waves have converted the buffer's last tile (waiting for them at THEIR priority: -1 % on the configs[4] slice)
        __builtin_amdgcn_s_setprio(0);
        wait_ge(cnt + 4 + s_img, 8u * s_round);
        __builtin_amdgcn_s_setprio(3);
      }
      return stg0 + s_img * sbuf;
    };
    auto publish = [&]() __attribute__((always_inline)) {
      signal_add(cnt + s_img, lane);
      if (++s_img == nimg) { s_img = 0; ++s_round; }
    };
    // Edge stages: the tile's rows are dealt to the 16 lane groups by WORK: group g takes the rows whose work prefix
    // P(i) = slots before row i + c * i falls into [g, g + 1) * P(rows) / 16 (c = cost of an empty row, raised with the
    // tile's slot count so that no group gets more than 31 rows). Every row's slots are summed by ONE group in slot
    // order, so sums do not depend on the partition, the tile or the launch. Lane l holds the tile's row pointers
    // l, l + 32, l + 64 (clamped); pointers and the group's first slot records are fetched one (tile, mode) ahead.
    struct RowPtrs { int a, b, c; };
    auto tile_rows16 = [&](int it_) {
      const int left = myrows - it_ * BM;
      const int r = left < BM ? left : BM;
      return (r + 15) & ~15;
    };
    auto rp_of = [&](int it_, int mode_) {
      const int32_t *rp = p.rowptr + int64_t(mode_) * (p.n + 1);
      const int row0 = row_lo + it_ * BM;
      auto at = [&](int i) {
        int node = row0 + (i < BM ? i : BM);
        node = node < row_hi ? node : row_hi;
        return rp[node];
      };
      RowPtrs r;
      r.a = at(lig); r.b = at(lig + 32); r.c = at(lig + 64);
      return r;
    };
    auto rp_get = [&](const RowPtrs &r, int idx) {      // idx group-uniform, 0..BM: the tile's row pointer idx
      const int from = glane0 + (idx & 31);
      const int va = __shfl(r.a, from), vb = __shfl(r.b, from), vc = __shfl(r.c, from);
      return idx < 32 ? va : (idx < 64 ? vb : vc);
    };
    struct Part { int lo, hi, rp; };                     // rows [lo, hi) of the tile; rp: lane l holds row pointer lo + min(l, hi - lo)
    auto partition = [&](const RowPtrs &r, int nr) {     // nr = rows of the tile rounded up to 16
      const int base = __shfl(r.a, glane0);
      const int tot = rp_get(r, nr) - base;
      const int c = 2 > (tot >> 8) + 1 ? 2 : (tot >> 8) + 1;
      const int ptot = tot + c * nr;
      const int thr_lo = (grp * ptot) >> 4, thr_hi = ((grp + 1) * ptot) >> 4;
      int lo = 0, hi = 0;
      const int vals[3] = {r.a, r.b, r.c};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int i = lig + 32 * k;
        const int pw = (vals[k] - base) + c * i;
        const unsigned long long blo = __ballot(i < nr && pw < thr_lo), bhi = __ballot(i < nr && pw < thr_hi);
        lo += __popc(uint32_t(blo >> glane0));
        hi += __popc(uint32_t(bhi >> glane0));
      }
      Part q;
      q.lo = lo; q.hi = hi;
      const int idx = lo + (lig < hi - lo ? lig : hi - lo);
      const int from = glane0 + (idx & 31);
      const int va = __shfl(r.a, from), vb = __shfl(r.b, from), vc = __shfl(r.c, from);
      q.rp = idx < 32 ? va : (idx < 64 ? vb : vc);
      return q;
    };
    auto rec_chunk = [&](int cbeg, int end) {   // lane i: record of slot cbeg + i (clamped to the range's last slot)
      int4 r = make_int4(0, 0, 0, 0);
      if (end > cbeg) r = p.rec[(cbeg + lig < end) ? cbeg + lig : end - 1];
      return r;
    };
    Part cur = {0, 0, 0};
    int4 currec = make_int4(0, 0, 0, 0);
    if (my_tiles > 0) {
      cur = partition(rp_of(0, 0), tile_rows16(0));
      currec = rec_chunk(__shfl(cur.rp, glane0), __shfl(cur.rp, glane0 + (cur.hi - cur.lo)));
    }
    bool tables_ready = !RELLDS;
    for (int it = 0; it < my_tiles; ++it) {
      const int r0 = row_lo + it * BM;
      const int nr = tile_rows16(it);
      // stage order per tile: self loop, in-half, out-half (the multiply walks the k-blocks in this order too)
      for (int mi = 0; mi < 3; ++mi) {
        const int mode = mi == 0 ? 2 : mi - 1;
        if (mode < 2) {
          if (!tables_ready) {          // the relation table is put into LDS by the MFMA waves during the first stage
            wait_ge(cnt + 8, 8u);
            tables_ready = true;
          }
          const int myrp = cur.rp, e_lo = cur.lo, e_hi = cur.hi, e_n = cur.hi - cur.lo;   // this group's rows [e_lo, e_hi)
          const int4 firstrec = currec;
          const int ee_sub_mode = p.ee_sub[mode];
          const bool has_next = mode == 0 || it + 1 < my_tiles;   // next (tile, mode) with records
          RowPtrs nrp = {0, 0, 0};
          if (has_next) nrp = rp_of(mode == 0 ? it : it + 1, mode == 0 ? 1 : 0);
          const int nnr = mode == 0 ? nr : (it + 1 < my_tiles ? tile_rows16(it + 1) : 16);
          Part nxt = {0, 0, 0};
          bool next_recs_issued = false;
          int4 nrec = make_int4(0, 0, 0, 0);
          auto prefetch_next = [&]() __attribute__((always_inline)) {   // the next (tile, mode)'s partition and first records
            if (has_next) {
              nxt = partition(nrp, nnr);
              nrec = rec_chunk(__shfl(nxt.rp, glane0), __shfl(nxt.rp, glane0 + (nxt.hi - nxt.lo)));
            }
          };
          int2 myhub = make_int2(-1, 0);                                  // lane i: hub chunks of destination e_lo + i
          {
            const int node = r0 + e_lo + lig;
            if (p.hubinfo && lig < e_n && node < row_hi) myhub = p.hubinfo[int64_t(mode) * p.n + node];
          }
          const int beg = __shfl(myrp, glane0), end = __shfl(myrp, glane0 + e_n);
          for (int pass = 0; pass < npass; ++pass) {
            unsigned char *img = acquire();
            bool col_ok[NCH];
            int coff[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
              const int c_ = pass * (128 * NCH) + j * 128 + lig * 4;
              col_ok[j] = c_ < p.d;
              coff[j] = col_ok[j] ? c_ : 0;   // lanes past the row width repeat columns 0-3 and store zeros (or nothing)
            }
            const float *relbase = RELLDS ? rel_lds : p.rel;
            int4 myrec = firstrec;
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
              write_row(img, row, sum, col_ok);
#pragma unroll
              for (int j = 0; j < NCH; ++j) sum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
              ++row;
            };
            int cbase = beg;                                   // first slot of the record chunk held in myrec
            for (int s = beg; s < end; s += UB) {
              DIAG_LAP(t_o);
              if (s >= cbase + CH) {                           // group-uniform: next record chunk of a long range
                cbase += CH;
                myrec = rec_chunk(cbase, end);
              }
              int rsrc[UB], rtyp[UB], rnrm[UB];
#pragma unroll
              for (int u = 0; u < UB; ++u) {
                const int from = glane0 + (((s + u < end) ? s + u : end - 1) - cbase);
                rsrc[u] = __shfl(myrec.x, from);
                rtyp[u] = __shfl(myrec.y, from);
                rnrm[u] = __shfl(myrec.z, from);
              }
              float4 xv[UB][NCH], rv[UB][NCH], ev[UB][NCH];
#pragma unroll
              for (int u = 0; u < UB; ++u) {
                const uint32_t erow = uint32_t(((s + u < end) ? s + u : end - 1) - ee_sub_mode);
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                  xv[u][j] = *reinterpret_cast<const float4 *>(p.x + coff[j] + uint64_t(uint32_t(rsrc[u])) * ldx32);
                  if (!RELLDS) rv[u][j] = *reinterpret_cast<const float4 *>(p.rel + coff[j] + uint64_t(uint32_t(rtyp[u])) * d32);
                  ev[u][j] = *reinterpret_cast<const float4 *>(p.ee + coff[j] + uint64_t(erow) * d32);
                }
              }
              DIAG_LAP(t_a);
              if (!next_recs_issued) {   // behind this batch's row loads: the next (tile, mode)'s partition and records
                next_recs_issued = true;
                prefetch_next();
              }
              DIAG_LAP(t_b);
#ifdef MGCN_DIAG
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              ++n_load;
#endif
              DIAG_LAP(t_load);
#pragma unroll
              for (int u = 0; u < UB; ++u) {
                if (s + u < end) {
                  while (s + u >= nb) {
                    flush();
                    nb = __shfl(myrp, glane0 + (row - e_lo) + 1);
                  }
                  const float wgt = __int_as_float(rnrm[u]);
#pragma unroll
                  for (int j = 0; j < NCH; ++j) {
                    const float4 rr = RELLDS ? *reinterpret_cast<const float4 *>(relbase + coff[j] + uint32_t(rtyp[u]) * d32) : rv[u][j];
                    sum[j] = f4axpy(sum[j], f4mul3(f4mul3(xv[u][j], rr), ev[u][j]), wgt);
                  }
                }
              }
              DIAG_LAP(t_d);
            }
            if (!next_recs_issued) {
              next_recs_issued = true;
              prefetch_next();
            }
            while (row < e_hi) flush();  // last run, then zero rows for destinations without slots
            publish();                   // this wave's rows of the image are written
          }
          cur = nxt;
          currec = nrec;
        } else {  // self loop: (x * loop_rel) * loop_edge, model.py:91-94,101; group g owns rows g * NRT .. + NRT of the tile
          for (int pass = 0; pass < npass; ++pass) {
            unsigned char *img = acquire();
            bool col_ok[NCH];
            int coff[NCH];
            float4 lr[NCH], le[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
              const int c_ = pass * (128 * NCH) + j * 128 + lig * 4;
              col_ok[j] = c_ < p.d;
              coff[j] = col_ok[j] ? c_ : 0;
              lr[j] = *reinterpret_cast<const float4 *>(p.loop_rel + coff[j]);
              le[j] = *reinterpret_cast<const float4 *>(p.loop_edge + coff[j]);
            }
            const int g_lo = grp * NRT;
            float4 xs[NRT][NCH];
#pragma unroll
            for (int i = 0; i < NRT; ++i) {
              const int node = (r0 + g_lo + i < row_hi) ? r0 + g_lo + i : row_hi - 1;   // rows past the run: computed, never stored
#pragma unroll
              for (int j = 0; j < NCH; ++j) xs[i][j] = *reinterpret_cast<const float4 *>(p.x + int64_t(node) * p.ldx + coff[j]);
            }
#pragma unroll
            for (int i = 0; i < NRT; ++i) {
              if (g_lo + i < nr) {
                float4 v[NCH];
#pragma unroll
                for (int j = 0; j < NCH; ++j) v[j] = f4mul3(f4mul3(xs[i][j], lr[j]), le[j]);
                write_row(img, g_lo + i, v, col_ok);
              }
            }
            publish();
          }
        }
      }
    }
    // all_rel = rel @ rels_weight (model.py:107), by the gather waves once their last stage is in LDS (the MFMA waves
    // still have stages and the last epilogue to go). One item = one relation row x 16 columns per wave: the four
    // 16-lane groups run the four K quarters of small_matmul_kernel's arithmetic (sequential fmaf chains), the partial
    // sums are added in quarter order — values bit-identical to the separate launch, one load round trip per 32 k.
    if (p.rel_out) {
      const int rows = p.rel_rows - 1, k = p.d, n = p.o;
      const int ncg = (n + 15) / 16, items = rows * ncg;
      const int kper = (k + 3) / 4;
      const int qd = lane >> 4;
      const int k0 = qd * kper, k1 = (k0 + kper < k) ? k0 + kper : k;
      for (int item = (wave - 8) * nblk + bid; item < items; item += nblk * 8) {
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
            const int kk = k0 + i0 + u;
            const int kc = (i0 + u < kper && kk < k1) ? kk : 0;
            av[u] = ap[kc];
            bv[u] = bp[int64_t(kc) * n];
          }
#pragma unroll
          for (int u = 0; u < UR; ++u) {
            const int kk = k0 + i0 + u;
            if (i0 + u < kper && kk < k1) a = fmaf(av[u], bv[u], a);
          }
        }
        const float q1 = __shfl(a, (lane & 15) + 16), q2 = __shfl(a, (lane & 15) + 32), q3 = __shfl(a, (lane & 15) + 48);
        if (qd == 0 && ok) p.rel_out[int64_t(row) * n + col] = ((a + q1) + q2) + q3;
      }
    }
    diag_end();
  } else {
    // ------------------------------------------------------------------------------------------ MULTIPLY
    auto multiply = [&](auto Hc) __attribute__((always_inline)) {
    constexpr int H = decltype(Hc)::value;
    static_assert(NT == 13 || NT == 32, "13 column tiles (O <= 208) or 32 (O <= 512)");
    // The SIMD's two MFMA waves (w = wave & 3, half H) share its column tiles. NT = 13: tiles 3w, 3w + 1 (H = 0) or 3w + 2
    // and a share of the 13th tile, dealt as single (column tile, row tile) units, row tile 4 j + w (H = 1). NT = 32:
    // tiles 8w + 4H + e, e = 0..3. A wave's k-block is E ENTRIES of one column tile each: the weight fragments of an
    // entry (three bf16 pieces, 12 registers) are loaded one entry ahead, so only two triples are live — which leaves
    // registers to read the row fragments of row tile rt + 1 before the MFMAs of row tile rt.
    constexpr int E = NT == 13 ? 2 : 4;
    constexpr bool UNITS = NT == 13 && H == 1;         // the wave's last entry is single units of the 13th column tile
    constexpr int XF = (NRT + 3) / 4;                  // ... at most XF of them
    const int w = wave & 3;
    const int r = lane & 15, gq = lane >> 4;
    auto ct_of = [&](int e) { return NT == 13 ? (e == 0 ? 3 * w + 2 * H : (H == 0 ? 3 * w + 1 : 12)) : 8 * w + 4 * H + e; };
    int xrt[XF];
#pragma unroll
    for (int j = 0; j < XF; ++j) xrt[j] = (UNITS && 4 * j + w < NRT) ? 4 * j + w : NRT;   // NRT = no unit
    auto wload = [&](u32x4 (&wv)[3], int g, int ct) __attribute__((always_inline)) {
      const u32x4 *base = p.wp + ((int64_t(g) * NT + ct) * 3) * 64 + lane;
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) wv[pc] = base[pc * 64];
    };
    // Three NAMES for two live weight triples; entry i of the (3 k-block) loop body uses name i % 3 (G is a multiple of 3)
    u32x4 wn[3][3];
    f32x4 acc[E][NRT];
    auto zero_acc = [&]() {
#pragma unroll
      for (int e = 0; e < E; ++e) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) acc[e][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    // lane holds out[row = 16 rt + r][16 ct + 4 gq .. + 3] (operands swapped: W is the MFMA's A operand)
    auto store_unit = [&](f32x4 a, int node, int col, const float4 &sc, const float4 &sh) {
      if (node < row_hi) {
        const float4 v = make_float4(tanh3_(fmaf(a[0], sc.x, sh.x)), tanh3_(fmaf(a[1], sc.y, sh.y)),
                                     tanh3_(fmaf(a[2], sc.z, sh.z)), tanh3_(fmaf(a[3], sc.w, sh.w)));
        *reinterpret_cast<float4 *>(p.out + int64_t(node - p.node0) * p.ldo + col) = v;
      }
    };
    auto epilogue = [&](int it_) {
      const int node0_ = row_lo + it_ * BM + r;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int col = ct_of(e) * 16 + 4 * gq;
        if (col < p.o) {
          const float4 sc = *reinterpret_cast<const float4 *>(epi + col), sh = *reinterpret_cast<const float4 *>(epi + OP + col);
          if (UNITS && e == E - 1) {
#pragma unroll
            for (int j = 0; j < XF; ++j) {
              if (xrt[j] < NRT) store_unit(acc[e][j], node0_ + xrt[j] * 16, col, sc, sh);
            }
          } else {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) store_unit(acc[e][rt], node0_ + rt * 16, col, sc, sh);
          }
        }
      }
    };

    // (Every workgroup walks the k-blocks in the same order: a row's sum must not depend on which workgroup, tile or
    // launch — whole graph or one rank's destination range — computes it.)
    int kb = 0, pass = 0;
    int nkb_ = 0, npass_ = 0, nmode = 0;          // the k-block after the current one: (mode, pass, ordinal)
    auto advance_next = [&]() {
      const int n = (npass_ == npass - 1) ? p.nkb_last : p.kbp;
      if (++nkb_ == n) {
        nkb_ = 0;
        if (++npass_ == npass) {
          npass_ = 0;
          nmode = nmode == 2 ? 0 : nmode + 1;
        }
      }
    };
    auto gindex = [&]() {   // (packed weights are mode-major in the order in-half, out-half, self loop; stages run loop, in, out)
      const int mode_ = nmode == 0 ? 2 : nmode - 1;
      return mode_ * p.kbm + p.kbp * npass_ + nkb_;
    };
    int gcur = gindex();
    wload(wn[0], gcur, ct_of(0));
    advance_next();
    // once per workgroup, by the MFMA waves while the first stage is gathered: the epilogue's per-column vectors
    // (model.py:103-106 as one fma: tanh(acc * scale + shift)) and, when it fits, the relation table; published through
    // the `ready` counter.
    if (H == 0) {
      for (int c = (wave & 3) * 64 + lane; c < OP; c += 256) {
        const bool in = c < p.o;
        const float inv = in ? __builtin_amdgcn_rsqf(p.bn_var[c] + p.bn_eps) * p.bn_gamma[c] : 0.f;
        constexpr float third = 1.0f / 3.0f;   // (sum of the three modes) / 3, model.py:103, as a multiplication (<= 1 ulp)
        epi[c] = inv * third;
        epi[OP + c] = in ? ((p.bias ? p.bias[c] : 0.f) - p.bn_mean[c]) * inv + p.bn_beta[c] : 0.f;
      }
    }
    if (RELLDS) {
      const int n4 = ((p.rel_rows - 1) * p.d) >> 2;
      for (int i = wave * 64 + lane; i < n4; i += 512)
        reinterpret_cast<float4 *>(rel_lds)[i] = reinterpret_cast<const float4 *>(p.rel)[i];
    }
    signal_add(cnt + 8, lane);
    int m_img = 0;
    uint32_t m_round = 0, m_stage = 0;
    int nrt_eff = NRT;
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
    auto tile_entry = [&](f32x4 (&accv)[NRT], const u32x4 (&wv)[3], const unsigned char *ap) __attribute__((always_inline)) {
      if (nrt_eff == NRT) {   // every row tile exists: the fragments of row tile rt + 1 are read before the MFMAs of row tile rt
        bf16x8 a[2][3];
        frag(a[0], ap, 0);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          if (rt + 1 < NRT) frag(a[(rt + 1) & 1], ap, rt + 1);
          asm volatile("" ::: "memory");   // (the reads of row tile rt + 2 stay below: two fragment sets, not NRT)
          six(accv[rt], wv, a[rt & 1]);
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          if (rt < nrt_eff) {
            bf16x8 a[3];
            frag(a, ap, rt);
            six(accv[rt], wv, a);
          }
        }
      }
    };
    auto unit_entry = [&](f32x4 (&accv)[NRT], const u32x4 (&wv)[3], const unsigned char *ap) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < XF; ++j) {       // (H = 1 only: single units of the 13th column tile)
        if (xrt[j] < nrt_eff) {
          bf16x8 a[3];
          frag(a, ap, xrt[j]);
          six(accv[j], wv, a);
        }
      }
    };
    auto kblock = [&](auto Kc) __attribute__((always_inline)) {
      constexpr int K = decltype(Kc)::value;   // position of the k-block in the loop body (names of its entries: (K E + e) % 3)
      wload(wn[(K * E + 1) % 3], gcur, ct_of(1));   // this k-block's second entry
      const int nkb_c = (pass == npass - 1) ? p.nkb_last : p.kbp;
      if (kb == 0) {
        // the stage's tile: staged f32 rows -> the bf16 image. Wave w converts rows 8 i + w (NCH = 2) or 16 i + 2 w + (lane >> 5)
        // (NCH = 1: two rows per wave-instruction); a lane splits one float4 exactly into three bf16 pieces.
        DIAG_LAP(t_o);
        wait_ge(cnt + m_img, 8u * (m_round + 1));          // all 8 gather waves have staged their rows
        if (m_stage > 0) wait_ge(cnt + 10, 8u * m_stage);   // ... and all MFMA waves are done reading the previous image
        const unsigned char *stg = stg0 + m_img * sbuf;
        constexpr int RPI = NCH == 1 ? 16 : 8;               // rows per iteration of the 8 waves
        const int ql = NCH == 1 ? (lane & 31) : lane;         // quad (float4) of the row
        const int crow0 = NCH == 1 ? 2 * wave + (lane >> 5) : wave;
        const bool qok = ql < 2 * p.ncc;
        const int cc = ql >> 1;
        const int coff = cc * BM * 16 + (ql & 1) * 8;
        const int crot = (cc >> 1) & 7;
        auto convert_row = [&](int row, const float4 &v) __attribute__((always_inline)) {
          uint32_t h[2], m[2], l[2];
          split3p(v.x, v.y, h[0], m[0], l[0]);
          split3p(v.z, v.w, h[1], m[1], l[1]);
          unsigned char *dst = lds3 + coff + ((row & ~15) + (((row & 15) + crot) & 15)) * 16;
          *reinterpret_cast<uint2 *>(dst) = make_uint2(h[0], h[1]);
          *reinterpret_cast<uint2 *>(dst + piece) = make_uint2(m[0], m[1]);
          *reinterpret_cast<uint2 *>(dst + 2 * piece) = make_uint2(l[0], l[1]);
        };
        if (qok) {
          const unsigned char *src = stg + ql * 16;
          const int rows = nrt_eff * 16;
          int row = crow0;
          for (; row + RPI < rows; row += 2 * RPI) {     // two rows per step: the reads first, then the arithmetic
            const float4 v0 = *reinterpret_cast<const float4 *>(src + row * rowb);
            const float4 v1 = *reinterpret_cast<const float4 *>(src + (row + RPI) * rowb);
            convert_row(row, v0);
            convert_row(row + RPI, v1);
          }
          for (; row < rows; row += RPI) convert_row(row, *reinterpret_cast<const float4 *>(src + row * rowb));
        }
        signal_add(cnt + 4 + m_img, lane);                  // this wave's staged rows are consumed
        signal_add(cnt + 9, lane);
        DIAG_LAP(t_a);
        wait_ge(cnt + 9, 8u * (m_stage + 1));                // the whole image is written
        DIAG_LAP(t_b);
      }
      int qc = 4 * kb + gq;
      qc = qc < p.ncc ? qc : p.ncc - 1;   // columns past the image (last k-block): any finite value of the row, their weights are zero
      const unsigned char *ap = lds3 + (qc * BM + ((r + ((qc >> 1) & 7)) & 15)) * 16;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (e > 0) {                       // the entry after this one: the next column tile, or the next k-block's first
          if (e + 1 < E) {
            wload(wn[(K * E + e + 1) % 3], gcur, ct_of(e + 1));
          } else {
            gcur = gindex();               // (wraps into the next tile: same weights)
            wload(wn[(K * E + E) % 3], gcur, ct_of(0));
            advance_next();
          }
        }
        if (NT == 13 || ct_of(e) * 16 < p.o) {    // (NT = 32: column tiles past the output width are skipped)
          if (UNITS && e == E - 1) unit_entry(acc[e], wn[(K * E + e) % 3], ap);
          else tile_entry(acc[e], wn[(K * E + e) % 3], ap);
        }
      }
      if (++kb == nkb_c) {               // this wave is done reading the image
        DIAG_LAP(t_d);
        signal_add(cnt + 10, lane);
        ++m_stage;
        if (++m_img == nimg) { m_img = 0; ++m_round; }
        kb = 0;
        if (++pass == npass) pass = 0;
      }
    };
    wait_ge(cnt + 8, 8u);   // epilogue vectors (written by four of these waves) are in LDS
    for (int it = 0; it < my_tiles; ++it) {
      zero_acc();
      {
        const int left = myrows - it * BM;
        nrt_eff = left < BM ? (left + 15) >> 4 : NRT;
      }
      for (int g0 = 0; g0 < p.G; g0 += 3) {
        kblock(std::integral_constant<int, 0>{});
        kblock(std::integral_constant<int, 1>{});
        kblock(std::integral_constant<int, 2>{});
      }
      epilogue(it);   // overlaps the gather, which is up to nimg stages ahead
    }
    diag_end();
    };
    if (wave < 4) multiply(std::integral_constant<int, 0>{});
    else multiply(std::integral_constant<int, 1>{});
  }
}

int pick_nt3(int o) { return o <= 208 ? 13 : 32; }   // column tiles of the multiply role (narrower outputs ride along zero-padded)

// wp[((g * NT + ct) * 3 + piece) * 64 + lane] = 8 bf16: W[mode * D + 32 kbi + 8 (lane >> 4) + i][16 ct + (lane & 15)],
// i = 0..7, zero outside; g = mode * kbm + kbi (k-block kbi of the mode: 32 consecutive input columns).
__global__ __launch_bounds__(256) void pack3_kernel(const float *__restrict__ w, u32x4 *__restrict__ wp, int d, int o,
                                                    int kbm, int nt, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, piece = (idx >> 6) % 3, ct = ((idx >> 6) / 3) % nt, g = (idx >> 6) / (3 * nt);
  const int mode = g / kbm, kbi = g - mode * kbm;
  const int col = ct * 16 + (lane & 15), k0 = 32 * kbi + 8 * (lane >> 4);
  uint32_t bits[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + 2 * i + j;
      v[j] = (k < d && col < o) ? w[(int64_t(mode) * d + k) * o + col] : 0.f;
    }
    uint32_t h, m, l;
    split3p(v[0], v[1], h, m, l);
    bits[i] = piece == 0 ? h : piece == 1 ? m : l;
  }
  wp[idx] = u32x4{bits[0], bits[1], bits[2], bits[3]};
}

struct Shape3 {
  int nch, npass, nkb_last, kbp, kbm, G, ncc;
};
Shape3 shape3(int d, int nch = 0) {   // nch: float4 per lane and slot walk (0 = by width)
  Shape3 s;
  // 128-column passes by default (one float4 per lane, four slots per batch): round 4's A/B on the product build has the 200-wide
  // layer at 96 / 107 us (warm / cold) against 108 / 118 with one 256-column walk (two float4, two slots per batch) on the WN18RR
  // shape and 187 against 193 on FB15k-237; the 256-column walk stays reachable through `tune` bits 12-13 = 2
  s.nch = nch ? nch : 1;
  const int wpass = 128 * s.nch;
  s.npass = (d + wpass - 1) / wpass;
  const int wlast = d - wpass * (s.npass - 1);
  s.nkb_last = (wlast + 31) / 32;
  s.kbp = 4 * s.nch;
  s.kbm = s.kbp * (s.npass - 1) + s.nkb_last;
  s.G = 3 * s.kbm;
  const int w0 = d < wpass ? d : wpass;
  s.ncc = (w0 + 7) / 8;
  return s;
}

constexpr size_t LDS_MAX = size_t(160) * 1024;

#ifdef MGCN_DIAG
unsigned long long *diag_buf3() {
  static unsigned long long *buf = nullptr;
  if (!buf) {
    if (hipMalloc(&buf, 1024 * 16 * 8 * 8) != hipSuccess) buf = nullptr;
    else (void)hipMemset(buf, 0, 1024 * 16 * 8 * 8);
  }
  return buf;
}
#endif

size_t lds_bytes3(const Shape3 &s, int nrt, int nimg, int nt, size_t rel_bytes) {   // one bf16 image, nimg f32 staging buffers
  return size_t(3) * s.ncc * (nrt * 16) * 16 + size_t(nimg) * (nrt * 16) * s.ncc * 32 + 64 + size_t(2) * nt * 16 * 4 + rel_bytes;
}

template <int NT, int NRT, int NCH, bool RELLDS>
int launch3(const Args3 &p, int grid, size_t lds, hipStream_t st) {
  // (the attribute is sticky per device and raising it costs a few microseconds: set on every launch, no state kept)
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused3_kernel<NT, NRT, NCH, RELLDS>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_MAX)) != hipSuccess)
    return mgcn::fail(MGCN_ELAUNCH, "layer_fused3: cannot reserve %zu bytes of LDS", LDS_MAX);
  hipLaunchKernelGGL((layer_fused3_kernel<NT, NRT, NCH, RELLDS>), dim3(unsigned(grid)), dim3(T3), lds, st, p);
  MGCN_CHECK_LAUNCH("layer_fused3_kernel");
  return MGCN_OK;
}

template <int NT, int NRT, int NCH>
int launch3_rel(const Args3 &p, int grid, size_t lds, bool rel_lds, hipStream_t st) {
  if (rel_lds) return launch3<NT, NRT, NCH, true>(p, grid, lds, st);
  return launch3<NT, NRT, NCH, false>(p, grid, lds, st);
}

template <int NT, int NCH>
int launch3_nrt(const Args3 &p, int nrt, int grid, size_t lds, bool rel_lds, hipStream_t st) {
  if (nrt == 3) return launch3_rel<NT, 3, NCH>(p, grid, lds, rel_lds, st);
  if (nrt == 4) return launch3_rel<NT, 4, NCH>(p, grid, lds, rel_lds, st);
  return launch3_rel<NT, 5, NCH>(p, grid, lds, rel_lds, st);
}

}  // namespace

#ifdef MGCN_DIAG
extern "C" int mgcn_diag_fused3(unsigned long long *host_out) {   // [1024][16][4] of the LAST gen-3 launch (diagnostics build)
  unsigned long long *b = diag_buf3();
  if (!b) return 1;
  return hipMemcpy(host_out, b, 1024 * 16 * 8 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif

namespace mgcn {

bool fused3_takes(int32_t dim_in, int32_t dim_out) {
  return dim_in > 0 && dim_in % 4 == 0 && dim_in <= 1024 && dim_out > 0 && dim_out % 4 == 0 && dim_out <= 512;
}

size_t fused3_packed_bytes(int32_t dim_in, int32_t dim_out) { return size_t(shape3(dim_in).G) * pick_nt3(dim_out) * 3 * 64 * 16; }

int fused3_pack(int32_t dim_in, int32_t dim_out, const float *w_dev, void *wp_dev, void *stream) {
  const Shape3 s = shape3(dim_in);
  const int nt = pick_nt3(dim_out);
  const int total = s.G * nt * 3 * 64;
  hipLaunchKernelGGL(pack3_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w_dev,
                     reinterpret_cast<u32x4 *>(wp_dev), dim_in, dim_out, s.kbm, nt, total);
  MGCN_CHECK_LAUNCH("pack3_kernel");
  return MGCN_OK;
}

// tune: 0 = automatic; bits 0-3 row tiles per tile (3 / 4 / 5), bits 4-7 staging buffers (1..4), bits 8-9 relation table in LDS
// (1 = never, 2 = whenever it fits), bits 12-13 columns per slot walk (1 = 128, 2 = 256): for A/B runs (tools/), never
// needed for correctness (the packed weights do not depend on it).
int fused3_launch(int64_t num_nodes, int32_t dim_in, int32_t dim_out, int32_t num_rel_rows, const int32_t *rowptr_dev,
                  const mgcn_edge_rec *rec_dev, const float *x_dev, int64_t ldx, const float *rel_dev,
                  const float *loop_rel_dev, const float *ee_dev, const float *loop_edge_dev, const void *wp_dev,
                  const float *bias_dev, const float *bn_mean_dev, const float *bn_var_dev, const float *bn_gamma_dev,
                  const float *bn_beta_dev, float bn_eps, float *out_dev, int64_t ldo, int64_t node_begin,
                  int64_t node_end, int64_t ee_sub_in, int64_t ee_sub_out, const int32_t *hubinfo_dev, int64_t chunk_begin,
                  const float *partial_dev, const float *rels_weight_dev, float *rel_out_dev, const int32_t *row_bounds_dev,
                  int32_t num_row_bounds, int32_t tune, uint32_t *status_dev, void *stream) {
  const int t_nch = (tune >> 12) & 3;
  if (t_nch > 2) return mgcn::fail(MGCN_EINVAL, "layer_fwd_fused: bad tune %d", tune);
  const Shape3 s = shape3(dim_in, t_nch);
  Args3 p = {};
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
  p.npass = s.npass; p.nkb_last = s.nkb_last; p.kbp = s.kbp; p.kbm = s.kbm; p.G = s.G; p.ncc = s.ncc;
  p.bn_eps = bn_eps;
  p.status = status_dev;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  // one contiguous run of rows per workgroup, a multiple of 16; one workgroup per CU
  const int64_t nrows = node_end - node_begin;
  int64_t rpw = ((nrows + cus - 1) / cus + 15) / 16 * 16;
  if (rpw < 16) rpw = 16;
  int grid = int(nrows > 0 ? (nrows + rpw - 1) / rpw : 1);
  p.rows_per_wg = int32_t(rpw);
  if (row_bounds_dev && num_row_bounds > 0 && nrows > 0) {   // the caller's runs, one workgroup each (at most one per CU is the point)
    p.bounds = row_bounds_dev;
    grid = num_row_bounds;
    const int64_t per = (nrows + grid - 1) / grid;            // (GraphCSR.workgroup_bounds caps its runs by the same rule)
    rpw = per <= 80 ? (per + 15) / 16 * 16 : (per + 79) / 80 * 80;
  }
  const int nt = pick_nt3(dim_out);
  const size_t rel_bytes = rel_dev ? size_t(num_rel_rows - 1) * dim_in * 4 : 0;
  // Geometry: the tallest tile (weight fragments feed 6 * NRT MFMAs) that leaves room for three staging buffers, else two;
  // the relation table rides in LDS when it fits beside them (a third of the gather's row loads). Tiles taller than
  // the run are pointless.
  const int t_nrt = tune & 15, t_img = (tune >> 4) & 15, t_rel = (tune >> 8) & 3;
  int nrt = 0, nimg = 0;
  bool rel_lds = false;
  const int nrt_cap = nt == 32 ? 3 : rpw >= 80 ? 5 : rpw >= 64 ? 4 : 3;   // (32 column tiles: 48 accumulator registers at 3 row tiles)
  auto fits = [&](int a, int b, bool r) { return lds_bytes3(s, a, b, nt, r ? rel_bytes : 0) <= LDS_MAX; };
  const bool rel_wanted = rel_bytes > 0 && rel_bytes <= size_t(32) * 1024 && t_rel != 1;
  if (t_nrt || t_img || t_nch) {
    nrt = t_nrt ? (t_nrt < nrt_cap ? t_nrt : nrt_cap) : nrt_cap;   // (a tile taller than the run is pointless)
    nimg = t_img ? t_img : 2;
    rel_lds = rel_wanted && fits(nrt, nimg, true);
    if (nrt < 3 || nrt > nrt_cap || nimg < 1 || nimg > 4 || !fits(nrt, nimg, rel_lds))
      return mgcn::fail(MGCN_EINVAL, "layer_fwd_fused: tune %d does not fit the LDS", tune);
  } else {
    for (int want_rel = rel_wanted ? 1 : 0; want_rel >= 0 && !nrt; --want_rel) {
      for (int a = nrt_cap; a >= 3 && !nrt; --a) {
        for (int b = 3; b >= 2 && !nrt; --b) {   // (two staging buffers at least: the gather must be able to run ahead)
          if (fits(a, b, want_rel != 0)) { nrt = a; nimg = b; rel_lds = want_rel != 0; }
        }
      }
    }
    if (!nrt && fits(3, 1, false)) { nrt = 3; nimg = 1; }   // 256-column passes (D > 248): one staging buffer
    if (!nrt) return mgcn::fail(MGCN_EUNSUPPORTED, "layer_fwd_fused: no tile geometry fits the LDS (D=%d O=%d)", dim_in, dim_out);
  }
  p.nimg = nimg;
#ifdef MGCN_DIAG
  p.diag = diag_buf3();
#endif
  const size_t lds = lds_bytes3(s, nrt, nimg, nt, rel_lds ? rel_bytes : 0);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nt == 32) {
    if (s.nch == 1) return launch3_rel<32, 3, 1>(p, grid, lds, rel_lds, st);
    return launch3_rel<32, 3, 2>(p, grid, lds, rel_lds, st);
  }
  if (s.nch == 1) return launch3_nrt<13, 1>(p, nrt, grid, lds, rel_lds, st);
  return launch3_nrt<13, 2>(p, nrt, grid, lds, rel_lds, st);
}

}  // namespace mgcn
