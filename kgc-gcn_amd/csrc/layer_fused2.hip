// Fused layer forward (eval), second generation — aggregation + dense step + epilogue in ONE launch (gfx950);
// replaces model.py:29-30, 99-106, 111-118 for one destination tile per workgroup pass.
//
// One 1024-thread workgroup per CU (16 waves x 128 VGPRs, 133-150 KB of LDS), persistent over tiles of BM = 16 * NRT
// destinations (80; 64 when that finishes the launch in fewer row-steps). Two roles, eight waves each (two of each per SIMD):
//   waves 8-15 GATHER  sixteen 32-lane groups (16 B per lane = 128 columns of the layer input per stage), the tile's rows
//              dealt to the groups by work; a group walks its contiguous slot range 4 slots (8 row loads) at a time in
//              slot order: the same sums as agg_fwd_kernel. A finished row is split EXACTLY into three bf16 pieces
//              (hi + mid + lo = the f32 value) and written to the stage's LDS image.
//   waves 0-7  MULTIPLY  acc += A_tile . W with v_mfma_f32_16x16x32_bf16 on the split operands: the six products
//              hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid (everything down to 2^-16 of |a||w|; what is dropped is
//              below 2^-24, the rounding of one f32 product), f32 accumulation: f32-faithful at 6/16 of the f32
//              MFMA's issue time. A wave owns NT/8 column tiles x ALL row tiles of the block tile, so a weight
//              fragment (three bf16 pieces, pre-split and pre-packed by pack2_kernel, read straight from L2) feeds
//              6 * NRT MFMAs; the operands are swapped (W as the A operand) so that a lane ends up with four
//              consecutive output columns of one row: the epilogue (/3, bias, BN eval, tanh: model.py:103-106)
//              runs on the accumulators and stores 16 bytes per lane.
// Stage = (mode, 128-column chunk of the input): the gather waves fill LDS image (s + 1) & 1 while the MFMA waves
// multiply image s & 1; one workgroup barrier per stage (all workgroups walk the weights in lockstep, which keeps the
// packed weights L2-resident between the step's alternating layers: why this kernel, not layer_fused3.hip, takes the
// shapes with D <= 256, O <= 208 — measured), the pipeline runs across modes and tiles.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "mgcn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int T2 = 1024;   // 8 MFMA waves + 8 gather waves: two of each per SIMD, 128 VGPRs per wave

struct Args2 {
  const int32_t *rowptr;
  const int4 *rec;
  const float *x, *rel, *loop_rel, *ee, *loop_edge;
  const u32x4 *wp;        // packed weights [G][NT][3][64] (8 bf16 per lane)
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
  int32_t nch, nkb_last, kbm, G;   // 128-column chunks per mode, k-blocks of the last chunk, k-blocks per mode / tile
  float bn_eps;
#ifdef MGCN_DIAG
  unsigned long long *stamps;   // [grid][2 roles][128]: s_memtime at stage starts / ends of wave 0 (multiply) and wave 4 (gather)
  int32_t ablate;   // diagnostics build only (tools/fused_ablate.py, tools/ablate_device_time.sh): bit 0 no slots gathered, bit 1 no MFMAs, bit 2 no epilogue
#endif
};
#ifdef MGCN_DIAG
#define MGCN_ABLATE(bit) (p.ablate & (bit))
#define MGCN_STAMP(role, idx)                                                                                      \
  do {                                                                                                             \
    if (p.stamps && lane == 0 && (idx) < 128) p.stamps[(int64_t(blockIdx.x) * 2 + (role)) * 128 + (idx)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define MGCN_ABLATE(bit) 0
#define MGCN_STAMP(role, idx) do {} while (0)
#endif

__device__ __forceinline__ float tanh2_(float v) {   // as layer_fused.hip: exp2 + rcp, 7 VALU per value
  const float t = __builtin_amdgcn_exp2f(fabsf(v) * -2.885390081777927f);
  return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), v);
}

// Exact three-way split of two f32 values into bf16 pieces, packed {even, odd}: hi = bf16(v) (round to nearest even),
// mid = bf16(v - hi), lo = v - hi - mid; every difference is exact, so hi + mid + lo == v bit for bit for finite v (see
// layer_fused3.hip: the same split, the same six products).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3p(float v0, float v1, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{v0, v1}, bf16x2));            // v_cvt_pk_bf16_f32
  const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
  m = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}

// wp[((g * NT + ct) * 3 + piece) * 64 + lane] = 8 bf16: W[mode * D + chunk * 128 + 8 * (4 kb + (lane >> 4)) + i]
// [16 ct + (lane & 15)], i = 0..7, zero outside; g = mode * kbm + 4 * chunk + kb.
__global__ __launch_bounds__(256) void pack2_kernel(const float *__restrict__ w, u32x4 *__restrict__ wp, int d, int o,
                                                    int kbm, int nt, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int lane = idx & 63, piece = (idx >> 6) % 3, ct = ((idx >> 6) / 3) % nt, g = (idx >> 6) / (3 * nt);
  const int mode = g / kbm, kbi = g - mode * kbm, chunk = kbi >> 2, kb = kbi & 3;
  const int col = ct * 16 + (lane & 15), k0 = chunk * 128 + 8 * (4 * kb + (lane >> 4));
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

__device__ __forceinline__ float4 f4mul2(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// LDS: [2 stage images][epilogue vectors 2 x 208 floats][relation table, when RELLDS]
constexpr int EPI_FLOATS = 2 * 208;
constexpr int REL_LDS_MAX_BYTES = 32 * 1024;

template <int NT, int NRT, bool RELLDS, bool HUBS>
__global__ __launch_bounds__(T2, 4) void layer_fused2_kernel(Args2 p) {
  constexpr int BM = NRT * 16;
  constexpr int PIECE = 16 * BM * 16;   // bytes of one bf16 piece of a stage image: 16 chunk columns x BM rows x 16 B
  constexpr int BUF = 3 * PIECE;
  constexpr int RPG = BM / 16;          // destinations per gather group (16 groups of 32 lanes)
  constexpr int UB = 4;                 // slots per gather batch (the role is bound by instruction issue, not by latency)
  constexpr int CH = 32 / UB * UB;      // slots served by one record chunk (lane i: slot cbase + i)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];
  float *epi = reinterpret_cast<float *>(lds2 + 2 * BUF);   // [scale | shift] x 208: the epilogue as one fma per value
  float *rel_lds = epi + EPI_FLOATS;                         // [rel_rows - 1][D] when RELLDS

  const int bid = int(blockIdx.x), nblk = int(gridDim.x);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nrows = p.node1 - p.node0;
  const int ntiles = (nrows + BM - 1) / BM;
  const int my_tiles = (ntiles - bid + nblk - 1) / nblk;   // >= 1 (grid <= ntiles)
  const int nch = p.nch;

#ifdef MGCN_DIAG
  if (p.stamps && lane == 0 && (wave == 0 || wave == 8)) {
    p.stamps[(int64_t(blockIdx.x) * 2 + (wave ? 1 : 0)) * 128 + 120] = __builtin_readcyclecounter();
    p.stamps[(int64_t(blockIdx.x) * 2 + (wave ? 1 : 0)) * 128 + 121] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  // The per-stage workgroup barrier orders LDS only: the waves' own LDS operations are drained (lgkmcnt), vector memory
  // is NOT (__syncthreads() would add s_waitcnt vmcnt(0): every barrier would then wait for the weight prefetch just
  // issued, the next records, and — after an epilogue — for 64 KB of output stores to be acknowledged).
  auto stage_barrier = [] () __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
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
    const int g_lo = grp * RPG, g_hi = g_lo + RPG;
    const int qcol = lig >> 1, frot = (qcol >> 1) & 7;
    const int wbase = qcol * BM * 16 + (lig & 1) * 8;
    const uint32_t ldx32 = uint32_t(p.ldx), d32 = uint32_t(p.d);

    auto write_row = [&](unsigned char *img, int row, float4 v, bool ok) __attribute__((always_inline)) {
      uint32_t h[2], m[2], l[2];
      split3p(ok ? v.x : 0.f, ok ? v.y : 0.f, h[0], m[0], l[0]);
      split3p(ok ? v.z : 0.f, ok ? v.w : 0.f, h[1], m[1], l[1]);
      unsigned char *dst = img + wbase + ((row & ~15) + (((row & 15) + frot) & 15)) * 16;
      *reinterpret_cast<uint2 *>(dst) = make_uint2(h[0], h[1]);
      *reinterpret_cast<uint2 *>(dst + PIECE) = make_uint2(m[0], m[1]);
      *reinterpret_cast<uint2 *>(dst + 2 * PIECE) = make_uint2(l[0], l[1]);
    };
    // Edge stages: the tile's BM destinations are dealt to the 16 lane groups by WORK, not by count: group g takes the
    // rows whose work prefix P(i) = slots before row i + c * i falls into [g, g + 1) * P(BM) / 16 (c = cost of an empty
    // row, raised with the tile's slot count so that no group gets more than 31 rows). A stage ends when its slowest
    // group ends: with 5 rows each the slowest of 16 groups carried ~1.6x the mean slots, by work ~1.15x. Every row's
    // slots are still summed by ONE group in slot order, so sums do not depend on the partition.
    // Per group, lane l holds the tile's row pointers l, l + 32, l + 64 (clamped to BM); both the pointers and the
    // group's first slot records are fetched one (tile, mode) ahead, so a stage starts straight at its row loads.
    struct RowPtrs { int a, b, c; };
    auto rp_of = [&](int it_, int mode_) {
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
    auto rp_get = [&](const RowPtrs &r, int idx) {      // idx group-uniform, 0..BM: the tile's row pointer idx
      const int from = glane0 + (idx & 31);
      const int va = __shfl(r.a, from), vb = __shfl(r.b, from), vc = __shfl(r.c, from);
      return idx < 32 ? va : (idx < 64 ? vb : vc);
    };
    struct Part { int lo, hi, rp; };                     // rows [lo, hi) of the tile; rp: lane l holds row pointer lo + min(l, hi - lo)
    auto partition = [&](const RowPtrs &r) {
      const int base = __shfl(r.a, glane0);
      const int tot = rp_get(r, BM) - base;
      const int c = 2 > (tot >> 8) + 1 ? 2 : (tot >> 8) + 1;
      const int ptot = tot + c * BM;
      const int thr_lo = (grp * ptot) >> 4, thr_hi = ((grp + 1) * ptot) >> 4;
      int lo = 0, hi = 0;
      const int vals[3] = {r.a, r.b, r.c};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int i = lig + 32 * k;
        const int pw = (vals[k] - base) + c * i;
        const unsigned long long blo = __ballot(i < BM && pw < thr_lo), bhi = __ballot(i < BM && pw < thr_hi);
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
    Part cur = partition(rp_of(0, 0));
    int4 currec = rec_chunk(__shfl(cur.rp, glane0), __shfl(cur.rp, glane0 + (cur.hi - cur.lo)));
    int stage = 0;
    for (int it = 0; it < my_tiles; ++it) {
      const int r0 = p.node0 + (bid + it * nblk) * BM;
      // stage order per tile: self loop, in-half, out-half. The first stage of the launch has nothing to overlap with:
      // it is the cheap one (no slot records, no dependent loads), and the relation table / epilogue vectors the MFMA
      // waves put into LDS meanwhile are published by that stage's barrier, before the first edge stage reads them.
      for (int mi = 0; mi < 3; ++mi) {
        const int mode = mi == 0 ? 2 : mi - 1;
        if (mode < 2) {
          const int myrp = cur.rp, e_lo = cur.lo, e_hi = cur.hi, e_n = cur.hi - cur.lo;   // this group's rows [e_lo, e_hi)
          const int4 firstrec = currec;
          const int ee_sub_mode = p.ee_sub[mode];
          const bool has_next = mode == 0 || it + 1 < my_tiles;   // next (tile, mode) with records
          RowPtrs nrp = {0, 0, 0};
          if (has_next) nrp = rp_of(mode == 0 ? it : it + 1, mode == 0 ? 1 : 0);
          Part nxt = {0, 0, 0};
          bool next_recs_issued = false;
          int4 nrec = make_int4(0, 0, 0, 0);
          auto prefetch_next = [&]() __attribute__((always_inline)) {   // the next (tile, mode)'s partition and first records
            if (has_next) {
              nxt = partition(nrp);
              nrec = rec_chunk(__shfl(nxt.rp, glane0), __shfl(nxt.rp, glane0 + (nxt.hi - nxt.lo)));
            }
          };
          int2 myhub = make_int2(-1, 0);                                  // lane i: hub chunks of destination e_lo + i
          {
            const int node = r0 + e_lo + lig;
            if (HUBS && p.hubinfo && lig < e_n && node < p.node1) myhub = p.hubinfo[int64_t(mode) * p.n + node];
          }
          const int beg = __shfl(myrp, glane0), end = MGCN_ABLATE(1) ? beg : __shfl(myrp, glane0 + e_n);
          for (int chunk = 0; chunk < nch; ++chunk, ++stage) {
            if (wave == 8) MGCN_STAMP(1, 2 * stage);
            unsigned char *img = lds2 + (stage & 1) * BUF;
            const int coff_ = chunk * 128 + lig * 4;
            const bool col_ok = coff_ < p.d;
            const int coff = col_ok ? coff_ : 0;   // lanes past the row width repeat columns 0-3 and store zeros
            const float *xb = p.x + coff, *relb = (RELLDS ? rel_lds : p.rel) + coff, *eeb = p.ee + coff;
            int4 myrec = firstrec;
            int row = e_lo, nb = __shfl(myrp, glane0 + 1);
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            auto flush = [&]() __attribute__((always_inline)) {   // the run of destination `row` is complete (group-uniform)
              if (HUBS && p.hubinfo) {     // a hub's own run is empty: its folded total sits in the row of its first chunk
                const int first = __shfl(myhub.x, glane0 + (row - e_lo)), cnt = __shfl(myhub.y, glane0 + (row - e_lo));
                if (cnt > 0) {
                  const float4 ps = *reinterpret_cast<const float4 *>(p.partial + int64_t(first - p.chunk0) * p.d + coff);
                  sum = make_float4(sum.x + ps.x, sum.y + ps.y, sum.z + ps.z, sum.w + ps.w);
                }
              }
              write_row(img, row, sum, col_ok);
              sum = make_float4(0.f, 0.f, 0.f, 0.f);
              ++row;
            };
            int cbase = beg;                                   // first slot of the record chunk held in myrec
#ifdef MGCN_DIAG
            int bstamp = 64;
            if (wave == 8 && stage == 1) { MGCN_STAMP(1, 63); }
#endif
            for (int s = beg; s < end; s += UB) {
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
              float4 xv[UB], rv[UB], ev[UB];
#pragma unroll
              for (int u = 0; u < UB; ++u) {
                if (MGCN_ABLATE(16)) rsrc[u] = 0;    // (diagnostics: every row load hits the same cached lines)
                xv[u] = *reinterpret_cast<const float4 *>(xb + uint64_t(uint32_t(rsrc[u])) * ldx32);
                if (!RELLDS) rv[u] = *reinterpret_cast<const float4 *>(relb + uint64_t(uint32_t(rtyp[u])) * d32);
                const uint32_t erow = MGCN_ABLATE(16) ? 0u : uint32_t(((s + u < end) ? s + u : end - 1) - ee_sub_mode);
                ev[u] = *reinterpret_cast<const float4 *>(eeb + uint64_t(erow) * d32);
              }
#ifdef MGCN_DIAG
              if (wave == 8 && stage == 1 && bstamp < 126) { MGCN_STAMP(1, bstamp); ++bstamp; }   // batch loads issued
#endif
              if (!next_recs_issued) {   // behind this batch's row loads: the next (tile, mode)'s partition and records
                next_recs_issued = true;
                prefetch_next();
              }
#ifdef MGCN_DIAG
              if (wave == 8 && stage == 1 && bstamp < 126) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                MGCN_STAMP(1, bstamp); ++bstamp;                                                    // ... and landed
              }
#endif
#pragma unroll
              for (int u = 0; u < UB; ++u) {
                if (s + u < end) {
                  while (s + u >= nb) {
                    flush();
                    nb = __shfl(myrp, glane0 + (row - e_lo) + 1);
                  }
                  const float4 rr = RELLDS ? *reinterpret_cast<const float4 *>(relb + uint32_t(rtyp[u]) * d32) : rv[u];
                  const float4 m = f4mul2(f4mul2(xv[u], rr), ev[u]);
                  const float wgt = __int_as_float(rnrm[u]);
                  sum = make_float4(sum.x + m.x * wgt, sum.y + m.y * wgt, sum.z + m.z * wgt, sum.w + m.w * wgt);
                }
              }
            }
            if (!next_recs_issued) {
              next_recs_issued = true;
              prefetch_next();
            }
#ifdef MGCN_DIAG
            if (wave == 8 && stage == 1 && bstamp < 126) { MGCN_STAMP(1, bstamp); ++bstamp; }       // batches consumed
#endif
            while (row < e_hi) flush();  // last run, then zero rows for destinations without slots
            if (wave == 8) MGCN_STAMP(1, 2 * stage + 1);
            stage_barrier();             // end of stage: image stage & 1 is complete
          }
          cur = nxt;
          currec = nrec;
        } else {  // self loop: (x * loop_rel) * loop_edge, model.py:91-94,101
          for (int chunk = 0; chunk < nch; ++chunk, ++stage) {
            if (wave == 8) MGCN_STAMP(1, 2 * stage);
            unsigned char *img = lds2 + (stage & 1) * BUF;
            const int coff_ = chunk * 128 + lig * 4;
            const bool col_ok = coff_ < p.d;
            const int coff = col_ok ? coff_ : 0;
            const float4 lr = *reinterpret_cast<const float4 *>(p.loop_rel + coff);
            const float4 le = *reinterpret_cast<const float4 *>(p.loop_edge + coff);
            float4 xs[RPG];
#pragma unroll
            for (int i = 0; i < RPG; ++i) {
              const int node = (r0 + g_lo + i < p.node1) ? r0 + g_lo + i : p.node1 - 1;   // rows past the range: computed, never stored
              xs[i] = *reinterpret_cast<const float4 *>(p.x + int64_t(node) * p.ldx + coff);
            }
#pragma unroll
            for (int i = 0; i < RPG; ++i) write_row(img, g_lo + i, f4mul2(f4mul2(xs[i], lr), le), col_ok);
            if (wave == 8) MGCN_STAMP(1, 2 * stage + 1);
            stage_barrier();
          }
        }
      }
    }
    // all_rel = rel @ rels_weight (model.py:107), by the gather waves once their last stage is in LDS (the MFMA waves
    // still have that stage and the last epilogue to go). One item = one relation row x 16 columns per wave: the four
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
#ifdef MGCN_DIAG
    if (p.stamps && lane == 0 && wave == 8) {
      p.stamps[(int64_t(blockIdx.x) * 2 + 1) * 128 + 122] = __builtin_readcyclecounter();
      p.stamps[(int64_t(blockIdx.x) * 2 + 1) * 128 + 123] = __builtin_amdgcn_s_memrealtime();
    }
#endif
  } else {
    // ------------------------------------------------------------------------------------------ MULTIPLY
    auto multiply = [&](auto Hc) __attribute__((always_inline)) {
    constexpr int H = decltype(Hc)::value;
    // Column group sg = wave & 3 (the SIMD) owns NT/4 column tiles; its two waves split them: half H = 0 takes the
    // first ceil, half H = 1 the rest plus the group's share of the NT % 4 left-over column tiles, dealt as single
    // (column tile, row tile) units. Two MFMA waves per SIMD: one's fragment waits are the other's issue slots.
    constexpr int QALL = NT / 4, R = NT % 4;
    constexpr int QA = (QALL + 1) / 2;
    constexpr int Q = H == 0 ? QA : QALL - QA;       // this wave's whole column tiles
    constexpr int QF = Q > 0 ? Q : 1;
    constexpr int NX = H == 1 ? R * NRT : 0;         // single units shared out round-robin over the four H = 1 waves
    constexpr int XE = (NX + 3) / 4;                 // ... at most XE per wave
    constexpr int XF = XE > 0 ? XE : 1;
    constexpr int XW = (R == 1) ? 1 : XF;            // weight fragments for them (R == 1: all in one column tile)
    const int w = wave & 3;
    const int r = lane & 15, gq = lane >> 4;
    const int ct0 = w * QALL + (H == 0 ? 0 : QA);
    int xct[XF], xrt[XF];
    bool xok[XF];
#pragma unroll
    for (int j = 0; j < XF; ++j) {
      const int e = 4 * j + w;
      xok[j] = XE > 0 && e < NX;
      xct[j] = xok[j] ? 4 * QALL + e / NRT : 0;
      xrt[j] = xok[j] ? e % NRT : -1;
    }
    const int G = p.G;
    auto wload = [&](u32x4 (&wq)[QF][3], u32x4 (&wx)[XW][3], int g) {
      if (MGCN_ABLATE(8)) g = 0;   // (diagnostics: every k-block re-reads the first one's fragments: 39 KB, L1-resident)
      if (MGCN_ABLATE(64) && g != 0) return;   // (diagnostics: no weight loads after the first: the fragments keep their registers)
      const u32x4 *base = p.wp + (int64_t(g) * NT) * 3 * 64 + lane;
#pragma unroll
      for (int t = 0; t < Q; ++t) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wq[t][pc] = base[((ct0 + t) * 3 + pc) * 64];
      }
      if (XE > 0) {
#pragma unroll
        for (int j = 0; j < XW; ++j) {
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) wx[j][pc] = base[(xct[j] * 3 + pc) * 64];
        }
      }
    };
    // Three NAMES for two live fragment sets: a k-block first issues the loads of the NEXT k-block into the set that
    // died one k-block ago, then multiplies with its own (loaded one k-block = ~100 MFMAs earlier). G is a multiple
    // of 3, so the rotation closes per tile with no conditional code between the k-blocks.
    u32x4 wq0[QF][3], wx0[XW][3], wq1[QF][3], wx1[XW][3], wq2[QF][3], wx2[XW][3];

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
        const float4 v = make_float4(tanh2_(fmaf(a[0], sc.x, sh.x)), tanh2_(fmaf(a[1], sc.y, sh.y)),
                                     tanh2_(fmaf(a[2], sc.z, sh.z)), tanh2_(fmaf(a[3], sc.w, sh.w)));
        *reinterpret_cast<float4 *>(p.out + int64_t(prow) * p.ldo + col) = v;
      }
    };
    auto epilogue = [&](int tile) {
      const int prow0 = tile * BM + r;
      auto column_tile = [&](int ct, auto &&body) {
        const int col = ct * 16 + 4 * gq;
        if (col < p.o) body(col, *reinterpret_cast<const float4 *>(epi + col), *reinterpret_cast<const float4 *>(epi + 208 + col));
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
            column_tile(xct[j], [&](int col, const float4 &sc, const float4 &sh) { store_unit(accx[j], prow0 + xrt[j] * 16, col, sc, sh); });
        }
      }
    };

    // (Every workgroup walks the k-blocks in the same order: a row's sum must not depend on which workgroup or which
    // launch — full or one rank's destination range — computes it. Rotating the order per workgroup was tried to
    // spread the weight reads over the L2 channels: no gain, and it breaks that bit-identity.)
    auto rotated = [&](int kb_, int) { return kb_; };
    int kb = 0, chunk = 0, stage = 0;
    bool barrier_done = false;
    int nkb_ = 0, nchunk = 0, nmode = 0;          // the k-block after the current one: (mode, chunk, ordinal)
    auto advance_next = [&]() {
      const int n = (nchunk == nch - 1) ? p.nkb_last : 4;
      if (++nkb_ == n) {
        nkb_ = 0;
        if (++nchunk == nch) {
          nchunk = 0;
          nmode = nmode == 2 ? 0 : nmode + 1;
        }
      }
    };
    auto gindex = [&]() {   // (packed weights are mode-major in the order in-half, out-half, self loop; stages run loop, in, out)
      const int mode_ = nmode == 0 ? 2 : nmode - 1;
      return mode_ * p.kbm + 4 * nchunk + rotated(nkb_, nchunk);
    };
    wload(wq0, wx0, gindex());
    advance_next();
    // once per workgroup, by the MFMA waves while the first stage is gathered: the epilogue's per-column vectors
    // (model.py:103-106 as one fma: tanh(acc * scale + shift)) and, when it fits, the relation table. Both are published
    // to the gather waves by the first stage barrier.
    if (H == 0) {
      const int t8 = (wave & 3) * 64 + lane;
      if (t8 < 208) {
        const int c = t8;
        const bool in = c < p.o;
        const float inv = in ? __builtin_amdgcn_rsqf(p.bn_var[c] + p.bn_eps) * p.bn_gamma[c] : 0.f;
        constexpr float third = 1.0f / 3.0f;   // (sum of the three modes) / 3, model.py:103, as a multiplication (<= 1 ulp)
        epi[c] = inv * third;
        epi[208 + c] = in ? ((p.bias ? p.bias[c] : 0.f) - p.bn_mean[c]) * inv + p.bn_beta[c] : 0.f;
      }
    }
    if (RELLDS) {
      const int n4 = ((p.rel_rows - 1) * p.d) >> 2;
      for (int i = wave * 64 + lane; i < n4; i += 512)
        reinterpret_cast<float4 *>(rel_lds)[i] = reinterpret_cast<const float4 *>(p.rel)[i];
    }
    auto kblock = [&](u32x4 (&wq)[QF][3], u32x4 (&wx)[XW][3], u32x4 (&nq)[QF][3], u32x4 (&nx)[XW][3]) {
      wload(nq, nx, gindex());           // the k-block after this one (wraps into the next tile: same weights)
      advance_next();
      const int nkb_c = (chunk == nch - 1) ? p.nkb_last : 4;
      if (kb == 0) {
        if (!barrier_done) stage_barrier();   // the stage's image is complete
        barrier_done = false;
        if (wave == 0) MGCN_STAMP(0, 2 * stage);
      }
      const int qc = 4 * rotated(kb, chunk) + gq;
      const unsigned char *ap = lds2 + (stage & 1) * BUF + (qc * BM + ((r + ((qc >> 1) & 7)) & 15)) * 16;
      // the six products, small terms first: (w piece, a piece) = (0,2) (2,0) (1,1) (0,1) (1,0) (0,0)
      constexpr int WP[6] = {0, 2, 1, 0, 1, 0}, AP[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        if (Q > 0 && !MGCN_ABLATE(2)) {
          bf16x8 a[3];
#pragma unroll
          for (int pc = 0; pc < 3; ++pc)
            a[pc] = __builtin_bit_cast(bf16x8, MGCN_ABLATE(32) ? wq[0][pc] : *reinterpret_cast<const u32x4 *>(ap + pc * PIECE + rt * 256));   // (bit 32: no LDS fragment reads)
#pragma unroll
          for (int pr = 0; pr < 6; ++pr) {
#pragma unroll
            for (int t = 0; t < Q; ++t)
              acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wq[t][WP[pr]]), a[AP[pr]],
                                                                   acc[rt][t], 0, 0, 0);
          }
        }
      }
      if (XE > 0) {   // this wave's single units: their own fragment reads (row tile = a wave-uniform offset), no branches
#pragma unroll      // inside the row-tile loop above
        for (int j = 0; j < XF; ++j) {
          if (xok[j]) {
            bf16x8 a[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
              a[pc] = __builtin_bit_cast(bf16x8, MGCN_ABLATE(32) ? wx[0][pc] : *reinterpret_cast<const u32x4 *>(ap + pc * PIECE + xrt[j] * 256));
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
              accx[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wx[XW == 1 ? 0 : j][WP[pr]]),
                                                                a[AP[pr]], accx[j], 0, 0, 0);
          }
        }
      }
      if (++kb == nkb_c) {
        if (wave == 0) MGCN_STAMP(0, 2 * stage + 1);
        if (stage < 8) MGCN_STAMP(0, 32 + 8 * stage + wave);   // (diagnostics: when each of the eight MFMA waves ends the stage)
        kb = 0;
        ++stage;
        if (++chunk == nch) chunk = 0;
      }
    };
    for (int it = 0; it < my_tiles; ++it) {
      zero_acc();
      for (int g0 = 0; g0 < G; g0 += 3) {
        kblock(wq0, wx0, wq1, wx1);
        kblock(wq1, wx1, wq2, wx2);
        kblock(wq2, wx2, wq0, wx0);
      }
      // the next tile's first barrier comes BEFORE this tile's epilogue: the images are not read any more, so the gather
      // waves go on with their next stage while these waves finish the rows (the accumulators are theirs alone)
      if (it + 1 < my_tiles) {
        stage_barrier();
        barrier_done = true;
      }
      if (wave == 0) MGCN_STAMP(0, 100 + 2 * it);
      if (!MGCN_ABLATE(4)) epilogue(bid + it * nblk);
      if (wave == 0) MGCN_STAMP(0, 101 + 2 * it);
    }
#ifdef MGCN_DIAG
    if (p.stamps && lane == 0 && wave == 0) {
      p.stamps[int64_t(blockIdx.x) * 2 * 128 + 122] = __builtin_readcyclecounter();
      p.stamps[int64_t(blockIdx.x) * 2 * 128 + 123] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    };
    if (wave < 4) multiply(std::integral_constant<int, 0>{});
    else multiply(std::integral_constant<int, 1>{});
  }
}

template <int NT, int NRT, bool RELLDS, bool HUBS>
int launch2(const Args2 &p, int grid, hipStream_t st) {
  constexpr size_t lds_bytes = size_t(2) * 3 * 16 * (NRT * 16) * 16 + EPI_FLOATS * 4 + (RELLDS ? REL_LDS_MAX_BYTES : 0);
  // (the attribute is sticky per device and setting it costs microseconds: done on every launch, no state kept)
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(&layer_fused2_kernel<NT, NRT, RELLDS, HUBS>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes)) != hipSuccess)
    return mgcn::fail(MGCN_ELAUNCH, "layer_fused2: cannot reserve %zu bytes of LDS", lds_bytes);
  hipLaunchKernelGGL((layer_fused2_kernel<NT, NRT, RELLDS, HUBS>), dim3(unsigned(grid)), dim3(T2), lds_bytes, st, p);
  MGCN_CHECK_LAUNCH("layer_fused2_kernel");
  return MGCN_OK;
}

int pick_nt2(int o) { return o <= 32 ? 2 : o <= 64 ? 4 : o <= 128 ? 8 : 13; }

struct Shape2 {
  int nch, nkb_last, kbm, G;
};
Shape2 shape2(int d) {
  Shape2 s;
  s.nch = (d + 127) / 128;
  const int wlast = d - 128 * (s.nch - 1);
  s.nkb_last = (wlast + 31) / 32;
  s.kbm = 4 * (s.nch - 1) + s.nkb_last;
  s.G = 3 * s.kbm;
  return s;
}

#ifdef MGCN_DIAG
unsigned long long *diag_stamps() {
  static unsigned long long *buf = nullptr;
  if (!buf && getenv("MGCN_FUSED_STAMPS")) {
    if (hipMalloc(&buf, 1024 * 2 * 128 * 8) != hipSuccess) buf = nullptr;
    else (void)hipMemset(buf, 0, 1024 * 2 * 128 * 8);
  }
  return buf;
}
#endif

}  // namespace

#ifdef MGCN_DIAG
extern "C" int mgcn_diag_read_stamps(unsigned long long *host_out) {   // [1024][2][128], of the LAST fused launch
  unsigned long long *b = diag_stamps();
  if (!b) return 1;
  return hipMemcpy(host_out, b, 1024 * 2 * 128 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif

namespace mgcn {

bool fused2_takes(int32_t dim_in, int32_t dim_out) {
  return dim_in > 0 && dim_in % 4 == 0 && dim_in <= 1024 && dim_out > 0 && dim_out % 4 == 0 && dim_out <= 208;
}

size_t fused2_packed_bytes(int32_t dim_in, int32_t dim_out) {
  return size_t(shape2(dim_in).G) * pick_nt2(dim_out) * 3 * 64 * 16;
}

int fused2_pack(int32_t dim_in, int32_t dim_out, const float *w_dev, void *wp_dev, void *stream) {
  const Shape2 s = shape2(dim_in);
  const int nt = pick_nt2(dim_out);
  const int total = s.G * nt * 3 * 64;
  hipLaunchKernelGGL(pack2_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w_dev,
                     reinterpret_cast<u32x4 *>(wp_dev), dim_in, dim_out, s.kbm, nt, total);
  MGCN_CHECK_LAUNCH("pack2_kernel");
  return MGCN_OK;
}

int fused2_launch(int64_t num_nodes, int32_t dim_in, int32_t dim_out, int32_t num_rel_rows, const int32_t *rowptr_dev,
                  const mgcn_edge_rec *rec_dev, const float *x_dev, int64_t ldx, const float *rel_dev,
                  const float *loop_rel_dev, const float *ee_dev, const float *loop_edge_dev, const void *wp_dev,
                  const float *bias_dev, const float *bn_mean_dev, const float *bn_var_dev, const float *bn_gamma_dev,
                  const float *bn_beta_dev, float bn_eps, float *out_dev, int64_t ldo, int64_t node_begin,
                  int64_t node_end, int64_t ee_sub_in, int64_t ee_sub_out, const int32_t *hubinfo_dev, int64_t chunk_begin,
                  const float *partial_dev, const float *rels_weight_dev, float *rel_out_dev, void *stream) {
  const Shape2 s = shape2(dim_in);
  Args2 p = {};
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
  p.nch = s.nch; p.nkb_last = s.nkb_last; p.kbm = s.kbm; p.G = s.G;
  p.bn_eps = bn_eps;
#ifdef MGCN_DIAG
  if (const char *ab = getenv("MGCN_FUSED_ABLATE")) p.ablate = atoi(ab);
  p.stamps = diag_stamps();
#endif
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  // Tile height: 80 rows (5 row tiles per weight fragment) unless 64-row tiles finish the launch in fewer row-steps on
  // this chip (makespan = tiles per CU, rounded up, x rows per tile): FB15k-237's 14 541 rows are one 64-row tile on
  // 228 CUs instead of one 80-row tile on 182.
  const int64_t nrows = node_end - node_begin;
  auto grid_for = [&](int rt) { const int64_t t = (nrows + rt * 16 - 1) / (rt * 16); return int(t < cus ? (t > 0 ? t : 1) : cus); };
  auto makespan = [&](int bm) { return ((nrows + bm - 1) / bm + cus - 1) / cus * bm; };
  int nrt = makespan(64) < makespan(80) ? 4 : 5;
#ifdef MGCN_DIAG
  if (const char *e = getenv("MGCN_FUSED_NRT")) nrt = atoi(e) == 4 ? 4 : 5;
#endif
  const int grid = grid_for(nrt);                          // persistent: one workgroup per CU
  hipStream_t st = static_cast<hipStream_t>(stream);
  // the relation table rides in LDS when it fits beside the stage images (a third of the gather's row loads)
  const bool rel_lds = rel_dev && size_t(num_rel_rows - 1) * dim_in * 4 <= size_t(REL_LDS_MAX_BYTES);
  // NT = 13 (the 200-wide layers) has all variants; narrower outputs take the general one
  const bool hubs = hubinfo_dev != nullptr;
  switch (pick_nt2(dim_out)) {
    case 2: return launch2<2, 5, false, true>(p, grid_for(5), st);
    case 4: return launch2<4, 5, false, true>(p, grid_for(5), st);
    case 8: return launch2<8, 5, false, true>(p, grid_for(5), st);
    default:
      if (nrt == 4) {
        if (rel_lds) return hubs ? launch2<13, 4, true, true>(p, grid, st) : launch2<13, 4, true, false>(p, grid, st);
        return hubs ? launch2<13, 4, false, true>(p, grid, st) : launch2<13, 4, false, false>(p, grid, st);
      }
      if (rel_lds) return hubs ? launch2<13, 5, true, true>(p, grid, st) : launch2<13, 5, true, false>(p, grid, st);
      return hubs ? launch2<13, 5, false, true>(p, grid, st) : launch2<13, 5, false, false>(p, grid, st);
  }
}

}  // namespace mgcn
