// Fused layer forward (eval), fifth generation — aggregation + dense step + epilogue in ONE launch (gfx950); replaces
// model.py:29-30, 99-107, 111-118 for D <= 256, O <= 208 (the benchmark's layers).
//
// The lockstep kernel's roles (layer_fused2.hip: waves 8-15 gather into one LDS image while waves 0-7 multiply the other, one
// LDS-only workgroup barrier per stage, a tile's epilogue after the next tile's first barrier — so the output stores sit in
// other waves' vmcnt queues than the gather's loads) with the fourth generation's K axis (layer_fused4.hip): the three modes are
// CONCATENATED [in-half | self loop | out-half] and cut into stages of EQUAL width, whatever mode boundaries fall inside:
//   a 100-wide layer:  2 stages of 152 columns  [in 0-99 + loop 0-51] [loop 52-99 + out 0-99]       (lockstep kernel: 3 stages)
//   a 200-wide layer:  4 stages of 152 columns  [in 0-151] [in 152-199 + loop 0-103] [loop 104-199 + out 0-55] [out 56-199]
//                                                                                                     (lockstep kernel: 6 stages)
// so (1) every stage carries an edge walk: none is multiply-bound (the lockstep kernel's self-loop stages gather 2.5 k cycles
// against 7.7 k of multiply, and the gather role idles), (2) a third fewer stage barriers, (3) K = 3 D is padded once per
// stage (5 k-blocks of 32 per 152 columns: 10 / 20 k-blocks per tile instead of 12 / 21). Sums in slot order (the sums of
// agg_fwd_kernel), the exact three-way bf16 split and its six products as in the other generations; the k order is this
// kernel's own, so its rows differ from generations 2 / 3 in the last bits — it is therefore dispatched by SHAPE alone
// (every launch of a shape — whole graph, destination range, table shard — takes it) and rows stay bit-identical across those.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mgcn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int T5 = 1024;       // 8 MFMA waves + 8 gather waves: two of each per SIMD, 128 VGPRs per wave
constexpr int NT5 = 13;        // column tiles of the multiply (O <= 208; narrower outputs ride along zero-padded)
constexpr int OP5 = NT5 * 16;

struct Args5 {
  const int32_t *rowptr;
  const int4 *rec;
  const float *x, *rel, *loop_rel, *ee, *loop_edge;
  const u32x4 *wp;        // packed weights [nstage * nkb][NT5][3][64] (8 bf16 per lane), pack5_kernel
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
  int32_t sw, ncc, nstage, nkb;   // stage width in columns of the K axis, 16-B chunk columns of an image, stages per tile, k-blocks per stage
  float bn_eps;
};

__device__ __forceinline__ float tanh5_(float v) {   // exp2 + rcp, 7 VALU per value
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
__device__ __forceinline__ float4 f4mul5(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4axpy5(float4 s, float4 m, float w) {
  return make_float4(s.x + m.x * w, s.y + m.y * w, s.z + m.z * w, s.w + m.w * w);
}

// Position mi of the concatenated K axis -> mode of the CSR / the stacked weights (0 in-half, 1 out-half, 2 self loop)
__host__ __device__ __forceinline__ int mode_of_pos5(int mi) { return mi == 0 ? 0 : (mi == 1 ? 2 : 1); }

// NCH: float4 per lane and row of a segment (segments are at most 128 * NCH columns wide)
template <int NRT, int NCH, bool RELLDS>
__global__ __launch_bounds__(T5, 4) void layer_fused5_kernel(Args5 p) {
  constexpr int BM = NRT * 16;
  constexpr int RPG = BM / 16;          // self-loop rows per gather group (16 groups of 32 lanes)
  constexpr int UB = 4 / NCH;           // slots per gather batch: 8 row loads of 16 B per lane in flight
  constexpr int CH = 32;                // slots served by one record chunk (lane i: slot cbase + i)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds5[];
  const int piece = p.ncc * BM * 16;    // bytes of one bf16 piece of a stage image: [chunk column][row][16 B]
  const int buf = 3 * piece;
  float *epi = reinterpret_cast<float *>(lds5 + 2 * buf);   // [scale | shift] x OP5: the epilogue as one fma per value
  float *rel_lds = epi + 2 * OP5;                           // [rel_rows - 1][D] when RELLDS

  const int bid = int(blockIdx.x), nblk = int(gridDim.x);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nrows = p.node1 - p.node0;
  const int ntiles = (nrows + BM - 1) / BM;
  const int my_tiles = (ntiles - bid + nblk - 1) / nblk;   // >= 1 (grid <= ntiles)
  const int nstage = p.nstage, d = p.d, k_all = 3 * p.d, sw = p.sw;

  // The per-stage workgroup barrier orders LDS only: vector memory is NOT drained (see layer_fused2.hip).
  auto stage_barrier = [] () __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  {  // once per workgroup, by all sixteen waves (every stage of this kernel walks edges, the first one included, so the tables
     // must be in LDS before the roles split): both images zeroed (columns past a stage's width are read against zero weights),
     // the epilogue's per-column vectors (model.py:103-106 as one fma), the relation table when it fits
    const int n16 = (2 * buf) >> 4;
    for (int i = tid; i < n16; i += T5) reinterpret_cast<uint4 *>(lds5)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < OP5) {
      const int c = tid;
      const bool in = c < p.o;
      const float inv = in ? __builtin_amdgcn_rsqf(p.bn_var[c] + p.bn_eps) * p.bn_gamma[c] : 0.f;
      constexpr float third = 1.0f / 3.0f;   // (sum of the three modes) / 3, model.py:103, as a multiplication (<= 1 ulp)
      epi[c] = inv * third;
      epi[OP5 + c] = in ? ((p.bias ? p.bias[c] : 0.f) - p.bn_mean[c]) * inv + p.bn_beta[c] : 0.f;
    }
    if (RELLDS) {
      const int n4 = ((p.rel_rows - 1) * p.d) >> 2;
      for (int i = tid; i < n4; i += T5) reinterpret_cast<float4 *>(rel_lds)[i] = reinterpret_cast<const float4 *>(p.rel)[i];
    }
  }
  __syncthreads();
  if (wave >= 8) {
    // ------------------------------------------------------------------------------------------ GATHER
    __builtin_amdgcn_s_setprio(3);      // the gather's loads go out ahead of the SIMD's MFMAs (round 3: -3 % on the step)
    const int gtid = tid - 512;
    const int grp = gtid >> 5, lig = gtid & 31;
    const int glane0 = lane & 32;
    const uint32_t ldx32 = uint32_t(p.ldx), d32 = uint32_t(p.d);
    // a row of a segment: lane's float4 j covers the segment's columns 128 j + 4 lig; kq = its quad index in the stage image
    auto write_row = [&](unsigned char *img, int row, const float4 (&v)[NCH], const bool (&ok)[NCH], const int (&kq)[NCH])
        __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        if (ok[j]) {
          uint32_t h[2], m[2], l[2];
          split3p(v[j].x, v[j].y, h[0], m[0], l[0]);
          split3p(v[j].z, v[j].w, h[1], m[1], l[1]);
          const int qc = kq[j] >> 1, frot = (qc >> 1) & 7;
          unsigned char *dst = img + qc * BM * 16 + (kq[j] & 1) * 8 + ((row & ~15) + (((row & 15) + frot) & 15)) * 16;
          *reinterpret_cast<uint2 *>(dst) = make_uint2(h[0], h[1]);
          *reinterpret_cast<uint2 *>(dst + piece) = make_uint2(m[0], m[1]);
          *reinterpret_cast<uint2 *>(dst + 2 * piece) = make_uint2(l[0], l[1]);
        }
      }
    };
    struct RowPtrs { int a, b, c; };
    auto rp_of = [&](int it_, int mode_) {           // lane l: the tile's row pointers l, l + 32, l + 64 (clamped)
      const int32_t *rp = p.rowptr + int64_t(mode_) * (p.n + 1);
      const int row0 = p.node0 + (bid + it_ * nblk) * BM;
      auto at = [&](int i) {
        int node = row0 + (i < BM ? i : BM);
        node = node < p.node1 ? node : p.node1;
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
    // The tile's BM destinations are dealt to the 16 lane groups by WORK (slots of the stage's edge modes + c per row, c raised
    // with the tile's slot count so that no group gets more than 31 rows); every row's slots are summed by ONE group in slot
    // order, so sums do not depend on the partition.
    struct Part { int lo, hi; };
    auto partition = [&](const RowPtrs &ra, bool has_a, const RowPtrs &rb, bool has_b) {
      const int base_a = __shfl(ra.a, glane0), base_b = __shfl(rb.a, glane0);
      const int tot = (has_a ? rp_get(ra, BM) - base_a : 0) + (has_b ? rp_get(rb, BM) - base_b : 0);
      const int c = 2 > (tot >> 8) + 1 ? 2 : (tot >> 8) + 1;
      const int ptot = tot + c * BM;
      const int thr_lo = (grp * ptot) >> 4, thr_hi = ((grp + 1) * ptot) >> 4;
      int lo = 0, hi = 0;
      const int va[3] = {ra.a, ra.b, ra.c}, vb[3] = {rb.a, rb.b, rb.c};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int i = lig + 32 * k;
        const int pw = (has_a ? va[k] - base_a : 0) + (has_b ? vb[k] - base_b : 0) + c * i;
        const unsigned long long blo = __ballot(i < BM && pw < thr_lo), bhi = __ballot(i < BM && pw < thr_hi);
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
    auto stage_k0 = [&](int s_) { return s_ * sw; };
    auto stage_k1 = [&](int s_) { const int k = s_ * sw + sw; return k < k_all ? k : k_all; };
    auto stage_has = [&](int s_, int mi) { return mi * d < stage_k1(s_) && (mi + 1) * d > stage_k0(s_); };
    // What a stage's edge walks need before their first row load — partition, row pointers per lane, first records — is fetched
    // one stage ahead (behind the current stage's first batch of row loads) into variables of its own that the current stage
    // never reads: nothing a group-uniform decision depends on is shared between two stages (LAB_NOTES.md, round 3's hang).
    struct Pre {
      Part part;
      int myrp[2];
      int4 rec[2];
    };
    auto prefetch = [&](int it_, int s_, Pre &q) __attribute__((always_inline)) {
      const bool ha = stage_has(s_, 0), hb = stage_has(s_, 2);
      RowPtrs ra = {0, 0, 0}, rb = {0, 0, 0};
      if (ha) ra = rp_of(it_, 0);
      if (hb) rb = rp_of(it_, 1);
      q.part = partition(ra, ha, rb, hb);
      q.myrp[0] = q.myrp[1] = 0;
      q.rec[0] = q.rec[1] = make_int4(0, 0, 0, 0);
      if (ha) {
        q.myrp[0] = rp_lane(ra, q.part);
        q.rec[0] = rec_chunk(__shfl(q.myrp[0], glane0), __shfl(q.myrp[0], glane0 + (q.part.hi - q.part.lo)));
      }
      if (hb) {
        q.myrp[1] = rp_lane(rb, q.part);
        q.rec[1] = rec_chunk(__shfl(q.myrp[1], glane0), __shfl(q.myrp[1], glane0 + (q.part.hi - q.part.lo)));
      }
    };
    Pre cur, nxt;
    prefetch(0, 0, cur);
    int stage = 0;
    for (int it = 0; it < my_tiles; ++it) {
      const int r0 = p.node0 + (bid + it * nblk) * BM;
      for (int s = 0; s < nstage; ++s, ++stage) {
        unsigned char *img = lds5 + (stage & 1) * buf;
        const int k0 = stage_k0(s), k1 = stage_k1(s);
        const bool last_stage = s + 1 == nstage;
        const bool has_next = !(last_stage && it + 1 == my_tiles);
        const int nit = last_stage ? it + 1 : it, ns = last_stage ? 0 : s + 1;
        bool next_issued = false;
        nxt = cur;                         // (overwritten by prefetch() when there is a next stage)
        // columns of the image past the stage's width (the K axis' last stage may be up to one quad narrower): zeros, so that
        // nothing another tile left there is multiplied (by zero weights: a NaN would survive that)
        {
          const int wq = (k1 - k0) >> 2, tq = (sw >> 2) - wq;     // quads of the stage's width, quads of the tail
          if (lig < tq) {
            const int kqz = wq + lig;
            const int qc = kqz >> 1, frot = (qc >> 1) & 7;
#pragma unroll
            for (int i = 0; i < RPG; ++i) {
              const int row = grp * RPG + i;
              unsigned char *dst = img + qc * BM * 16 + (kqz & 1) * 8 + ((row & ~15) + (((row & 15) + frot) & 15)) * 16;
              *reinterpret_cast<uint2 *>(dst) = make_uint2(0u, 0u);
              *reinterpret_cast<uint2 *>(dst + piece) = make_uint2(0u, 0u);
              *reinterpret_cast<uint2 *>(dst + 2 * piece) = make_uint2(0u, 0u);
            }
          }
        }
        // self loop: (x * loop_rel) * loop_edge, model.py:91-94,101; group g owns rows g * RPG .. + RPG of the tile. The stage's
        // columns [lc0, lc1) of that mode are LOADED here and WRITTEN behind the first edge segment's first batch of row loads,
        // so that their round trip is the batch's (a stage that waits for them alone pays a whole memory latency: + 8 us per layer)
        const int lc0 = (k0 > d ? k0 - d : 0), lc1 = (k1 - d < d ? k1 - d : d);
        const bool has_loop = lc1 > lc0;
        float4 xs[RPG][NCH], lrv[NCH], lev[NCH];
        bool lok[NCH];
        int lkq[NCH];
        if (has_loop) {
#pragma unroll
          for (int j = 0; j < NCH; ++j) {
            const int c_ = lc0 + 128 * j + 4 * lig;
            lok[j] = c_ < lc1;
            const int cc = lok[j] ? c_ : lc0;
            lkq[j] = (d + c_ - k0) >> 2;
#pragma unroll
            for (int i = 0; i < RPG; ++i) {
              const int node = (r0 + grp * RPG + i < p.node1) ? r0 + grp * RPG + i : p.node1 - 1;   // rows past the range: computed, never stored
              xs[i][j] = *reinterpret_cast<const float4 *>(p.x + int64_t(node) * p.ldx + cc);
            }
            lrv[j] = *reinterpret_cast<const float4 *>(p.loop_rel + cc);
            lev[j] = *reinterpret_cast<const float4 *>(p.loop_edge + cc);
          }
        }
        bool loop_written = !has_loop;
        auto write_loop_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
          for (int i = 0; i < RPG; ++i) {
            float4 v[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) v[j] = f4mul5(f4mul5(xs[i][j], lrv[j]), lev[j]);
            write_row(img, grp * RPG + i, v, lok, lkq);
          }
        };
        // edge segments: in-half (K position 0), then out-half (K position 2); this group's rows [e_lo, e_hi) in both
        const int e_lo = cur.part.lo, e_hi = cur.part.hi, e_n = e_hi - e_lo;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int mi = 2 * e, mode = e;
          const int c0 = (k0 > mi * d ? k0 - mi * d : 0), c1 = (k1 - mi * d < d ? k1 - mi * d : d);
          if (c1 <= c0) continue;
          const int myrp = cur.myrp[e];
          const int beg = __shfl(myrp, glane0), end = __shfl(myrp, glane0 + e_n);
          int4 myrec = cur.rec[e];
          const int ee_sub_mode = p.ee_sub[mode];
          int2 myhub = make_int2(-1, 0);                                  // lane i: hub chunks of destination e_lo + i
          {
            const int node = r0 + e_lo + lig;
            if (p.hubinfo && lig < e_n && node < p.node1) myhub = p.hubinfo[int64_t(mode) * p.n + node];
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
            write_row(img, row, sum, ok, kq);
#pragma unroll
            for (int j = 0; j < NCH; ++j) sum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            ++row;
          };
          int cbase = beg;                                   // first slot of the record chunk held in myrec
          // one batch of UB slots: issue the row loads, (FIRST batch of the stage only: the look-ahead of the next stage and the
          // self-loop rows ride behind them), then the arithmetic in slot order
          auto batch = [&](int sl, auto first_c) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_c)::value;
            int rsrc[UB], rtyp[UB], rnrm[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              const int from = glane0 + (((sl + u < end) ? sl + u : end - 1) - cbase);
              rsrc[u] = __shfl(myrec.x, from);
              rtyp[u] = __shfl(myrec.y, from);
              rnrm[u] = __shfl(myrec.z, from);
            }
            float4 xv[UB][NCH], rv[UB][NCH], ev[UB][NCH];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              const uint32_t erow = uint32_t(((sl + u < end) ? sl + u : end - 1) - ee_sub_mode);
#pragma unroll
              for (int j = 0; j < NCH; ++j) {
                xv[u][j] = *reinterpret_cast<const float4 *>(p.x + coff[j] + uint64_t(uint32_t(rsrc[u])) * ldx32);
                if (!RELLDS) rv[u][j] = *reinterpret_cast<const float4 *>(p.rel + coff[j] + uint64_t(uint32_t(rtyp[u])) * d32);
                ev[u][j] = *reinterpret_cast<const float4 *>(p.ee + coff[j] + uint64_t(erow) * d32);
              }
            }
            if (FIRST) {
              if (!next_issued) {   // behind this batch's row loads: the next stage's partition and records
                next_issued = true;
                if (has_next) prefetch(nit, ns, nxt);
              }
              if (!loop_written) {  // ... and the self-loop rows, whose loads are older than the batch's
                loop_written = true;
                write_loop_rows();
              }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              if (sl + u < end) {
                while (sl + u >= nb) {
                  flush();
                  nb = __shfl(myrp, glane0 + (row - e_lo) + 1);
                }
                const float wgt = __int_as_float(rnrm[u]);
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                  const float4 rr = RELLDS ? *reinterpret_cast<const float4 *>(relbase + coff[j] + uint32_t(rtyp[u]) * d32) : rv[u][j];
                  sum[j] = f4axpy5(sum[j], f4mul5(f4mul5(xv[u][j], rr), ev[u][j]), wgt);
                }
              }
            }
          };
          int sl = beg;
          if (sl < end) {
            batch(sl, std::true_type{});
            sl += UB;
          }
          for (; sl < end; sl += UB) {
            if (sl >= cbase + CH) {                          // group-uniform: next record chunk of a long range
              cbase += CH;
              myrec = rec_chunk(cbase, end);
            }
            batch(sl, std::false_type{});
          }
          while (row < e_hi) flush();  // last run, then zero rows for destinations without slots
        }
        if (!loop_written) write_loop_rows();     // (a group without slots in this stage, or a stage without an edge walk)
        if (!next_issued && has_next) prefetch(nit, ns, nxt);
        stage_barrier();               // end of stage: image stage & 1 is complete
        cur = nxt;
      }
    }
    // all_rel = rel @ rels_weight (model.py:107), by the gather waves once their last stage is in LDS. One item = one relation
    // row x 16 columns per wave: the four 16-lane groups run the four K quarters of small_matmul_kernel's arithmetic
    // (sequential fmaf chains), the partial sums are added in quarter order — values bit-identical to the separate launch.
    if (p.rel_out) {
      const int rows = p.rel_rows - 1, k = p.d, n = p.o;
      const int ncg = (n + 15) / 16, items = rows * ncg;
      const int kper = (k + 3) / 4;
      const int qd = lane >> 4;
      const int kq0 = qd * kper, kq1 = (kq0 + kper < k) ? kq0 + kper : k;
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
  } else {
    // ------------------------------------------------------------------------------------------ MULTIPLY
    auto multiply = [&](auto Hc) __attribute__((always_inline)) {
    constexpr int H = decltype(Hc)::value;
    // Column group sg = wave & 3 (the SIMD) owns 3 column tiles; its two waves split them: half H = 0 takes two, half H = 1 the
    // third plus the group's share of the 13th column tile, dealt as single (column tile, row tile) units.
    constexpr int QALL = NT5 / 4, R = NT5 % 4;
    constexpr int QA = (QALL + 1) / 2;
    constexpr int Q = H == 0 ? QA : QALL - QA;       // this wave's whole column tiles
    constexpr int QF = Q > 0 ? Q : 1;
    constexpr int NX = H == 1 ? R * NRT : 0;         // single units shared out round-robin over the four H = 1 waves
    constexpr int XE = (NX + 3) / 4;                 // ... at most XE per wave
    constexpr int XF = XE > 0 ? XE : 1;
    const int w = wave & 3;
    const int r = lane & 15, gq = lane >> 4;
    const int ct0 = w * QALL + (H == 0 ? 0 : QA);
    int xrt[XF];
    bool xok[XF];
#pragma unroll
    for (int j = 0; j < XF; ++j) {
      const int e = 4 * j + w;
      xok[j] = XE > 0 && e < NX;
      xrt[j] = xok[j] ? e % NRT : -1;
    }
    const int G = nstage * p.nkb;
    auto wload = [&](u32x4 (&wq)[QF][3], u32x4 (&wx)[3], int g) {
      const u32x4 *base = p.wp + (int64_t(g) * NT5) * 3 * 64 + lane;
#pragma unroll
      for (int t = 0; t < Q; ++t) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wq[t][pc] = base[((ct0 + t) * 3 + pc) * 64];
      }
      if (XE > 0) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wx[pc] = base[(12 * 3 + pc) * 64];
      }
    };
    // Three NAMES for two live fragment sets (layer_fused2.hip); the k-block loop is flat over the workgroup's whole
    // sequence (tiles x stages x k-blocks) and unrolled by three, so the rotation needs no property of the counts.
    u32x4 wq0[QF][3], wx0[3], wq1[QF][3], wx1[3], wq2[QF][3], wx2[3];
    f32x4 acc[NRT][QF], accx[XF];
    auto zero_acc = [&]() {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
        for (int t = 0; t < QF; ++t) acc[rt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < XF; ++j) accx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // lane holds out[row = 16 rt + r][16 ct + 4 gq .. + 3] (operands swapped: W is the MFMA's A operand)
    auto store_unit = [&](f32x4 a, int prow, int col, const float4 &sc, const float4 &sh) {
      if (prow < nrows) {
        const float4 v = make_float4(tanh5_(fmaf(a[0], sc.x, sh.x)), tanh5_(fmaf(a[1], sc.y, sh.y)),
                                     tanh5_(fmaf(a[2], sc.z, sh.z)), tanh5_(fmaf(a[3], sc.w, sh.w)));
        *reinterpret_cast<float4 *>(p.out + int64_t(prow) * p.ldo + col) = v;
      }
    };
    auto epilogue = [&](int tile) {
      const int prow0 = tile * BM + r;
      auto column_tile = [&](int ct, auto &&body) {
        const int col = ct * 16 + 4 * gq;
        if (col < p.o) body(col, *reinterpret_cast<const float4 *>(epi + col), *reinterpret_cast<const float4 *>(epi + OP5 + col));
      };
#pragma unroll
      for (int t = 0; t < Q; ++t) {
        column_tile(ct0 + t, [&](int col, const float4 &sc, const float4 &sh) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) store_unit(acc[rt][t], prow0 + rt * 16, col, sc, sh);
        });
      }
      if (XE > 0) {
#pragma unroll
        for (int j = 0; j < XF; ++j) {
          if (xok[j])
            column_tile(12, [&](int col, const float4 &sc, const float4 &sh) { store_unit(accx[j], prow0 + xrt[j] * 16, col, sc, sh); });
        }
      }
    };
    int kb = 0, st = 0, stage = 0, it = 0;     // k-block of the stage, stage of the tile, stage of the run, tile of the run
    bool barrier_done = false;
    const int total = my_tiles * G;
    wload(wq0, wx0, 0);
    auto kblock = [&](u32x4 (&wq)[QF][3], u32x4 (&wx)[3], u32x4 (&nq)[QF][3], u32x4 (&nx)[3], int b) {
      {
        int g = st * p.nkb + kb + 1;       // the k-block after this one (wraps into the next tile: same weights)
        g = g < G ? g : 0;
        wload(nq, nx, g);
      }
      if (kb == 0) {
        if (!barrier_done) stage_barrier();   // the stage's image is complete
        barrier_done = false;
        if (st == 0) zero_acc();
      }
      int qc = 4 * kb + gq;
      qc = qc < p.ncc ? qc : p.ncc - 1;    // columns past the image (a stage's last k-block): finite values, zero weights
      const unsigned char *ap = lds5 + (stage & 1) * buf + (qc * BM + ((r + ((qc >> 1) & 7)) & 15)) * 16;
      // the six products, small terms first: (w piece, a piece) = (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)
      constexpr int WP[6] = {0, 2, 1, 0, 1, 0}, AP[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        if (Q > 0) {
          bf16x8 a[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) a[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ap + pc * piece + rt * 256));
#pragma unroll
          for (int pr = 0; pr < 6; ++pr) {
#pragma unroll
            for (int t = 0; t < Q; ++t)
              acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[t][WP[pr]]), a[AP[pr]],
                                                                   acc[rt][t], 0, 0, 0);
          }
        }
      }
      if (XE > 0) {   // this wave's single units of the 13th column tile
#pragma unroll
        for (int j = 0; j < XF; ++j) {
          if (xok[j]) {
            bf16x8 a[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
              a[pc] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ap + pc * piece + xrt[j] * 256));
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
              accx[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wx[WP[pr]]), a[AP[pr]], accx[j], 0, 0, 0);
          }
        }
      }
      if (++kb == p.nkb) {
        kb = 0;
        ++stage;
        if (++st == nstage) {              // the tile's last stage is multiplied
          st = 0;
          // the next tile's first barrier comes BEFORE this tile's epilogue: the images are not read any more, so the gather
          // waves go on with their next stage while these waves finish the rows (the accumulators are theirs alone)
          if (b + 1 < total) {
            stage_barrier();
            barrier_done = true;
          }
          epilogue(bid + it * nblk);
          ++it;
        }
      }
    };
    for (int b = 0; b < total; b += 3) {
      kblock(wq0, wx0, wq1, wx1, b);
      if (b + 1 < total) kblock(wq1, wx1, wq2, wx2, b + 1);
      if (b + 2 < total) kblock(wq2, wx2, wq0, wx0, b + 2);
    }
    };
    if (wave < 4) multiply(std::integral_constant<int, 0>{});
    else multiply(std::integral_constant<int, 1>{});
  }
}

// wp[((g * NT5 + ct) * 3 + piece) * 64 + lane] = 8 bf16: Wk[sw * s + 32 j + 8 (lane >> 4) + i][16 ct + (lane & 15)], i = 0..7,
// g = s * nkb + j, zero for columns past the stage (32 j + ... >= its width) or past K / O; Wk = the stacked weights' rows in
// the kernel's K order [in-half | self loop | out-half] (the stacked matrix the caller passes is [W_in; W_out; W_loop]).
__global__ __launch_bounds__(256) void pack5_kernel(const float *__restrict__ w, u32x4 *__restrict__ wp, int d, int o, int sw, int nkb,
                                                    int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, piece = (idx >> 6) % 3, ct = ((idx >> 6) / 3) % NT5, g = (idx >> 6) / (3 * NT5);
  const int s = g / nkb, j = g - s * nkb;
  const int col = ct * 16 + (lane & 15), c0 = 32 * j + 8 * (lane >> 4);
  uint32_t bits[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int c = c0 + 2 * i + jj;             // column of the stage
      const int k = sw * s + c;
      const int mi = k / d, cm = k - mi * d;
      v[jj] = (c < sw && k < 3 * d && col < o) ? w[(int64_t(mode_of_pos5(mi)) * d + cm) * o + col] : 0.f;
    }
    uint32_t h, m, l;
    split3p(v[0], v[1], h, m, l);
    bits[i] = piece == 0 ? h : piece == 1 ? m : l;
  }
  wp[idx] = u32x4{bits[0], bits[1], bits[2], bits[3]};
}

constexpr size_t LDS_MAX5 = size_t(160) * 1024;
// Widest stage. D <= 128: 152 columns (two 80-row images of 19 chunk columns + WN18RR's relation table fit; a 100-wide layer is
// two stages). Wider inputs: 128 columns, so that a segment is one float4 per lane (4 slots per batch instead of 2) and a
// 200-wide relation table still rides in LDS (a 200-wide layer: five stages of 120 columns, 20 k-blocks).
int sw_max5(int d, int variant) { return variant > 0 ? 8 * (variant + 4) : (d <= 128 ? 152 : 128); }

struct Shape5 {
  int sw, ncc, nstage, nkb, nch;
};
Shape5 shape5(int d, int variant) {
  const int sw_max = sw_max5(d, variant);
  Shape5 s;
  s.nstage = (3 * d + sw_max - 1) / sw_max;
  s.sw = ((3 * d + s.nstage - 1) / s.nstage + 7) / 8 * 8;   // equal widths, a multiple of the 8-column chunk
  s.ncc = s.sw / 8;
  s.nkb = (s.sw + 31) / 32;
  const int wseg = d < s.sw ? d : s.sw;                       // widest segment of one mode inside a stage
  s.nch = wseg > 128 ? 2 : 1;
  return s;
}

size_t lds_bytes5(const Shape5 &s, int d, int nrt, size_t rel_bytes) {
  (void)d;
  return size_t(2) * 3 * s.ncc * (nrt * 16) * 16 + size_t(2) * OP5 * 4 + rel_bytes;
}

template <int NRT, int NCH, bool RELLDS>
int launch5(const Args5 &p, int grid, size_t lds, hipStream_t st) {
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused5_kernel<NRT, NCH, RELLDS>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_MAX5)) != hipSuccess)
    return mgcn::fail(MGCN_ELAUNCH, "layer_fused5: cannot reserve %zu bytes of LDS", LDS_MAX5);
  hipLaunchKernelGGL((layer_fused5_kernel<NRT, NCH, RELLDS>), dim3(unsigned(grid)), dim3(T5), lds, st, p);
  MGCN_CHECK_LAUNCH("layer_fused5_kernel");
  return MGCN_OK;
}

template <int NRT>
int launch5_v(const Args5 &p, int nch, bool rel_lds, int grid, size_t lds, hipStream_t st) {
  if (nch == 1) return rel_lds ? launch5<NRT, 1, true>(p, grid, lds, st) : launch5<NRT, 1, false>(p, grid, lds, st);
  return rel_lds ? launch5<NRT, 2, true>(p, grid, lds, st) : launch5<NRT, 2, false>(p, grid, lds, st);
}

}  // namespace

namespace mgcn {

bool fused5_takes(int32_t dim_in, int32_t dim_out) {
  return dim_in > 0 && dim_in % 4 == 0 && dim_in <= 256 && dim_out > 0 && dim_out % 4 == 0 && dim_out <= 208;
}

// variant: 0 = the shape's own stage width; 1..15 = widest stage 8 * (variant + 4) columns (A/B runs: `tune` bits 4-7 of the launch
// and the packing must agree)
size_t fused5_packed_bytes(int32_t dim_in, int32_t dim_out, int32_t variant) {
  (void)dim_out;
  const Shape5 s = shape5(dim_in, variant);
  return size_t(s.nstage) * s.nkb * NT5 * 3 * 64 * 16;
}

int fused5_pack(int32_t dim_in, int32_t dim_out, int32_t variant, const float *w_dev, void *wp_dev, void *stream) {
  const Shape5 s = shape5(dim_in, variant);
  const int total = s.nstage * s.nkb * NT5 * 3 * 64;
  hipLaunchKernelGGL(pack5_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w_dev,
                     reinterpret_cast<u32x4 *>(wp_dev), dim_in, dim_out, s.sw, s.nkb, total);
  MGCN_CHECK_LAUNCH("pack5_kernel");
  return MGCN_OK;
}

// tune: 0 = automatic; bits 0-3 row tiles per tile (4 / 5), bits 4-7 stage-width variant (as fused5_pack), bits 8-9 relation table
// in LDS (1 = never): for A/B runs.
int fused5_launch(int64_t num_nodes, int32_t dim_in, int32_t dim_out, int32_t num_rel_rows, const int32_t *rowptr_dev,
                  const mgcn_edge_rec *rec_dev, const float *x_dev, int64_t ldx, const float *rel_dev,
                  const float *loop_rel_dev, const float *ee_dev, const float *loop_edge_dev, const void *wp_dev,
                  const float *bias_dev, const float *bn_mean_dev, const float *bn_var_dev, const float *bn_gamma_dev,
                  const float *bn_beta_dev, float bn_eps, float *out_dev, int64_t ldo, int64_t node_begin,
                  int64_t node_end, int64_t ee_sub_in, int64_t ee_sub_out, const int32_t *hubinfo_dev, int64_t chunk_begin,
                  const float *partial_dev, const float *rels_weight_dev, float *rel_out_dev, int32_t tune, void *stream) {
  const Shape5 s = shape5(dim_in, (tune >> 4) & 15);
  Args5 p = {};
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
  p.sw = s.sw; p.ncc = s.ncc; p.nstage = s.nstage; p.nkb = s.nkb;
  p.bn_eps = bn_eps;
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  // Tile height: 80 rows unless 64-row tiles finish the launch in fewer row-steps on this chip (layer_fused2.hip)
  const int64_t nrows = node_end - node_begin;
  auto grid_for = [&](int rt) { const int64_t t = (nrows + rt * 16 - 1) / (rt * 16); return int(t < cus ? (t > 0 ? t : 1) : cus); };
  auto makespan = [&](int bm) { return ((nrows + bm - 1) / bm + cus - 1) / cus * bm; };
  const int t_nrt = tune & 15, t_rel = (tune >> 8) & 3;
  int nrt = t_nrt == 4 || t_nrt == 5 ? t_nrt : (makespan(64) < makespan(80) ? 4 : 5);
  const size_t rel_bytes = rel_dev ? size_t(num_rel_rows - 1) * dim_in * 4 : 0;
  const bool rel_wanted = rel_bytes > 0 && rel_bytes <= size_t(32) * 1024 && t_rel != 1;
  bool rel_lds = rel_wanted && lds_bytes5(s, dim_in, nrt, rel_bytes) <= LDS_MAX5;
  if (lds_bytes5(s, dim_in, nrt, rel_lds ? rel_bytes : 0) > LDS_MAX5) {
    nrt = 4;
    rel_lds = rel_wanted && lds_bytes5(s, dim_in, nrt, rel_bytes) <= LDS_MAX5;
    if (lds_bytes5(s, dim_in, nrt, rel_lds ? rel_bytes : 0) > LDS_MAX5)
      return mgcn::fail(MGCN_EUNSUPPORTED, "layer_fwd_fused: no tile geometry fits the LDS (D=%d O=%d)", dim_in, dim_out);
  }
  const size_t lds = lds_bytes5(s, dim_in, nrt, rel_lds ? rel_bytes : 0);
  const int grid = grid_for(nrt);                          // persistent: one workgroup per CU
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (nrt == 4) return launch5_v<4>(p, s.nch, rel_lds, grid, lds, st);
  return launch5_v<5>(p, s.nch, rel_lds, grid, lds, st);
}

}  // namespace mgcn
