// Fused layer forward (eval) — aggregation + dense step + epilogue in ONE launch (gfx950); replaces
// mgcn_aggregate_fwd + mgcn_dense_bn_tanh_fwd (model.py:29-30, 99-106, 111-118) when the shape allows.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mgcn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 32, KS = 16;

// tanh(v) = sign(v) (1 - t) / (1 + t), t = exp(-2|v|), with the hardware exp2 and reciprocal (1 ulp each): this
// kernel is bound by the SIMD's vector issue (VALU work does not overlap v_mfma_f32_16x16x4_f32 issue — measured,
// tools/mfma_coexec.hip), so the epilogue is written for instruction count: 7 VALU per value instead of ~20.
__device__ __forceinline__ float tanhf_(float v) {
  const float t = __builtin_amdgcn_exp2f(fabsf(v) * -2.885390081777927f);   // exp(-2|v|) = 2^(-2 log2(e) |v|)
  return copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), v);
}

int pick_nt(int64_t ncols) { return ncols <= 32 ? 2 : ncols <= 64 ? 4 : ncols <= 128 ? 8 : 13; }

// ---------------------------------------------------------------------------------------------
// Fused layer forward (eval): aggregation + dense step + epilogue in ONE launch; the [N, 3D] aggregate never
// leaves the CU. Block = 8 waves on one 32-destination tile, two roles:
//   waves 4-7  GATHER   lane groups (25 of 32 lanes x dwordx4 for D = 100) each own a run of consecutive
//              destinations; the run's slots are ONE contiguous CSR range walked U at a time (records, then
//              3*U row loads, then the arithmetic in slot order: the same sums as agg_fwd_kernel, bit for bit).
//              Finished rows go to the LDS tile As[mode & 1][32][D+2] (row stride / 2 odd => conflict-free
//              ds_read_b32 of the A fragments).
//   waves 0-3  MFMA     acc += As . W_mode with v_mfma_f32_16x16x4_f32. W comes straight from global memory
//              (L2-resident) in a pre-packed fragment order (pack_w_kernel): one dwordx4 per (k-block, column
//              tile, lane) holds the lane's B values of the block's four MFMA steps; loads run one k-block
//              ahead. No weight slab in LDS, hence no barrier inside a mode.
// The two roles are a 4-stage pipeline over the modes (in-half, out-half, self-loop): while the MFMA waves
// multiply mode m, the gather waves fetch mode m+1; one workgroup barrier per stage. Two blocks per CU.
struct FusedArgs {
  const int32_t *rowptr;
  const int4 *rec;
  const float *x, *rel, *loop_rel, *ee, *loop_edge;
  const float4 *wp;  // packed weights [3][nkb][NT][64] float4
  const float *bias, *bn_mean, *bn_var, *bn_gamma, *bn_beta;
  float *out;
  int64_t ldx, ldo;
  int32_t n, e, d, o, rel_rows, ee_slot_order, gs_log2;
  int32_t node0, node1;   // destinations [node0, node1) are this launch's (this rank's) share; out row 0 = node0
  int64_t ee_sub[2];      // slot-order per-edge table shard: row of (absolute) slot s of half h = s - ee_sub[h]
  const int2 *hubinfo;    // [2][N] (first chunk, chunk count) or null
  const float *partial;   // [chunks in play][D]; row (first chunk - chunk0) of a hub holds its folded total (pre-pass)
  int32_t chunk0;
  const float *rw;        // relation projection: rels_weight [D, O] (model.py:107) or null
  float *rel_out;         // [rel_rows - 1, O] = rel @ rels_weight
  int32_t rel_blocks;     // the first rel_blocks workgroups compute rel_out and leave
  int32_t ablate;  // timing diagnostics only (MGCN_FUSED_ABLATE): bit 0 skips the gather, bit 1 the MFMA loop
  float bn_eps;
};

__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }

// wp[((mode*nkb + kb)*nt + ct)*64 + lane].{x,y,z,w}[i] = W[mode*D + 16kb + 4i + (lane>>4)][16ct + (lane&15)], 0 outside
__global__ __launch_bounds__(256) void pack_w_kernel(const float *__restrict__ w, float4 *__restrict__ wp, int d, int o,
                                                     int nkb, int nt) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 3 * nkb * nt * 64) return;
  const int lane = idx & 63, ct = (idx >> 6) % nt, kb = ((idx >> 6) / nt) % nkb, mode = (idx >> 6) / (nt * nkb);
  const int col = ct * 16 + (lane & 15);
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = kb * KS + 4 * i + (lane >> 4);
    v[i] = (k < d && col < o) ? w[(int64_t(mode) * d + k) * o + col] : 0.f;
  }
  wp[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

constexpr int FUSED_THREADS = 512;

// Epilogue + row stores of a finished tile: /3, bias, BN(eval), tanh (model.py:103-106) on the raw accumulators the
// MFMA waves left in the LDS staging tile `Os`. Run by the 256 GATHER threads (their VALU idles on memory latency,
// while the MFMA waves' time is the kernel's critical path): thread -> one fixed float4 column, rows frow0,
// frow0 + rstep, ... Kept out of line so that its registers do not add to the gather loops' pressure.
__device__ __forceinline__ void finalize_tile(const FusedArgs &p, const float *Os, int ldo_s, int pr0, int gtid) {
  const int c4n = p.o >> 2;
  const int frow0 = gtid / c4n, fcol = (gtid - frow0 * c4n) * 4, rstep = 256 / c4n;
  if (frow0 >= rstep) return;
  const float4 mean = *reinterpret_cast<const float4 *>(p.bn_mean + fcol);
  const float4 var = *reinterpret_cast<const float4 *>(p.bn_var + fcol);
  const float4 gam = *reinterpret_cast<const float4 *>(p.bn_gamma + fcol);
  const float4 bet = *reinterpret_cast<const float4 *>(p.bn_beta + fcol);
  const float4 cb = p.bias ? *reinterpret_cast<const float4 *>(p.bias + fcol) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 inv = make_float4(__builtin_amdgcn_rsqf(var.x + p.bn_eps), __builtin_amdgcn_rsqf(var.y + p.bn_eps),
                                 __builtin_amdgcn_rsqf(var.z + p.bn_eps), __builtin_amdgcn_rsqf(var.w + p.bn_eps));
  for (int lrow = frow0; lrow < BM; lrow += rstep) {
    if (pr0 + lrow >= p.node1 - p.node0) break;
    float4 v = *reinterpret_cast<const float4 *>(Os + lrow * ldo_s + fcol);
    if (!(p.ablate & 8)) {
      constexpr float third = 1.0f / 3.0f;   // (sum of the three modes) / 3, model.py:103, as a multiplication (<= 1 ulp)
      v = make_float4(v.x * third, v.y * third, v.z * third, v.w * third);
      if (p.bias) v = make_float4(v.x + cb.x, v.y + cb.y, v.z + cb.z, v.w + cb.w);
      v = make_float4(tanhf_((v.x - mean.x) * inv.x * gam.x + bet.x), tanhf_((v.y - mean.y) * inv.y * gam.y + bet.y),
                      tanhf_((v.z - mean.z) * inv.z * gam.z + bet.z), tanhf_((v.w - mean.w) * inv.w * gam.w + bet.w));
    }
    *reinterpret_cast<float4 *>(p.out + int64_t(pr0 + lrow) * p.ldo + fcol) = v;
  }
}

// all_rel = rel @ rels_weight (model.py:107; the dropped last row means the loop row is never multiplied). The
// arithmetic is small_matmul_kernel's, item for item (one output row x 64 columns per four waves; the waves split K
// in quarters and run sequential fmaf chains; the four partial sums are added in wave order), so the fused layer
// returns bit-identical relations. Run by the first `rel_blocks` workgroups of the launch, two items at a time.
__device__ __forceinline__ void relation_projection(const FusedArgs &p, float *lds) {
  constexpr int UNR = 8;
  float *part = lds;                                  // [2][4][64]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = wave >> 2, w = wave & 3;
  const int rows = p.rel_rows - 1, k = p.d, n = p.o;
  const int ngrp = (n + 63) / 64, items = rows * ngrp;
  const int kper = (k + 3) / 4, k0 = w * kper, k1 = (k0 + kper < k) ? k0 + kper : k;
  for (int base = int(blockIdx.x) * 2; base < items; base += p.rel_blocks * 2) {   // block-uniform trip count
    const int item = base + half;
    const bool valid = item < items;
    const int row = valid ? item / ngrp : 0, col = (valid ? item - row * ngrp : 0) * 64 + lane;
    const bool ok = valid && col < n;
    const float *ap = p.rel + int64_t(row) * k;
    const float *bp = p.rw + (ok ? col : 0);
    float acc = 0.f;
    int kk = k0;
    for (; kk + UNR <= k1; kk += UNR) {
      float av[UNR], bv[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        av[u] = ap[kk + u];
        bv[u] = bp[int64_t(kk + u) * n];
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc = fmaf(av[u], bv[u], acc);
    }
    for (; kk < k1; ++kk) acc = fmaf(ap[kk], bp[int64_t(kk) * n], acc);
    part[(half * 4 + w) * 64 + lane] = acc;
    __syncthreads();
    if (w == 0 && ok) {
      const float *q = part + half * 256 + lane;
      p.rel_out[int64_t(row) * n + col] = ((q[0] + q[64]) + q[128]) + q[192];
    }
    __syncthreads();
  }
}

// PERSISTENT: a block walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and the gather -> multiply pipeline runs
// straight across tile boundaries: stage s = 3*tile + mode; while the MFMA waves multiply stage s the gather waves
// fetch stage s + 1 (the next tile's in-half when s is a self-loop stage). One workgroup barrier per stage. A
// tile's epilogue values are staged in LDS during its last stage and stored (16 bytes per lane, whole rows) by the
// MFMA waves at the beginning of the next one.
template <int NT>
__global__ __launch_bounds__(FUSED_THREADS, 4) void layer_fused_kernel(FusedArgs p) {
  constexpr int LDO = NT * 16 + 4;     // staging row stride (floats)
  constexpr int NTW = (NT + 3) / 4;    // column tiles per MFMA wave (at most)
  constexpr int U = 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lda = p.d + 2;
  float *As = lds;                     // [2][BM][lda]
  float *Os = lds + 2 * BM * lda;      // [BM][LDO] epilogue staging (2*BM*lda*4 bytes is a multiple of 16)

  if (int(blockIdx.x) < p.rel_blocks) {   // workgroup-uniform: these workgroups only project the relations
    relation_projection(p, lds);
    return;
  }
  const int bid = int(blockIdx.x) - p.rel_blocks, nblk = int(gridDim.x) - p.rel_blocks;   // tile workgroups
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const bool mfma_role = wave < 4;
  const int nkb = (p.d + KS - 1) / KS;
  const int ntiles = (p.node1 - p.node0 + BM - 1) / BM;
  const int my_tiles = (ntiles - bid + nblk - 1) / nblk;  // >= 1 (tile workgroups <= ntiles)

  // The two roles are two separate programs (disjoint live ranges -> each fits the register budget); both
  // execute exactly 3 * my_tiles + 1 workgroup barriers.
  if (!mfma_role) {
    const int gtid = tid - 256;
    const int gs = 1 << p.gs_log2;
    const int grp = gtid >> p.gs_log2, lig = gtid & (gs - 1);
    const int glane0 = lane & ~(gs - 1);             // first lane of this group inside its wave
    const int rpg = (BM * gs) / 256;                 // destinations per group (>= 1, < gs)
    const int g_lo = grp * rpg, g_hi = g_lo + rpg;
    const bool col_ok = lig * 4 < p.d;
    const int coff = col_ok ? lig * 4 : 0;           // lanes past the row width duplicate lane 0 (columns 0-3)
    const float *xb = p.x + coff, *relb = p.rel + coff, *eeb = p.ee + coff;
    const uint32_t ldx32 = uint32_t(p.ldx), d32 = uint32_t(p.d);
    auto finalize = [&](int tile_it) {
      if (!(p.ablate & 4)) finalize_tile(p, Os, LDO, (bid + tile_it * nblk) * BM, gtid);
    };
    // A stage's memory chain is row pointers -> slot records -> rows. The first two links are fetched ONE STAGE
    // AHEAD: lane i of a group holds the row pointer of destination g_lo + i and the record of slot beg + i (the
    // group's slots are one contiguous range), so a stage starts straight at its row loads and the records reach the
    // whole group through ds_bpermute. Runs longer than the group (hub-free runs are <= 64 slots per destination)
    // reload the record chunk on demand.
    auto rp_of = [&](int it_, int mode_) {
      int node = p.node0 + (bid + it_ * nblk) * BM + g_lo + (lig <= rpg ? lig : rpg);
      node = node < p.node1 ? node : p.node1;
      return p.rowptr[int64_t(mode_) * (p.n + 1) + node];   // absolute slot position
    };
    auto rec_chunk = [&](int cbeg, int end) {   // lane i: record of slot cbeg + i (clamped to the range's last slot)
      int4 r = make_int4(0, 0, 0, 0);
      if (end > cbeg) r = p.rec[(cbeg + lig < end) ? cbeg + lig : end - 1];
      return r;
    };
    int currp = (p.ablate & 1) ? 0 : rp_of(0, 0);
    int4 currec = rec_chunk(__shfl(currp, glane0), __shfl(currp, glane0 + rpg));
    int stage = 0;
    for (int it = 0; it < my_tiles; ++it) {
      const int r0 = p.node0 + (bid + it * nblk) * BM;
      for (int mode = 0; mode < 3; ++mode, ++stage) {
        float *at = As + (stage & 1) * BM * lda;
        if (mode == 2 && it > 0) finalize(it - 1);   // Os holds tile it-1 since the barrier two stages back
        if (p.ablate & 1) {
        } else if (mode < 2) {
          const int myrp = currp;
          int4 myrec = currec;
          const int ee_sub_mode = int(p.ee_sub[mode]);   // (a kernel argument indexed by the loop variable: read it once)
          const bool has_next = mode == 0 || it + 1 < my_tiles;   // next stage with records: (it, 1) or (it + 1, 0)
          int nrp = 0;
          if (has_next) nrp = rp_of(mode == 0 ? it : it + 1, mode == 0 ? 1 : 0);
          bool next_recs_issued = false;
          int4 nrec = make_int4(0, 0, 0, 0);
          int node = r0 + g_lo + (lig <= rpg ? lig : rpg);
          int2 myhub = make_int2(-1, 0);                                  // lane i: hub chunks of destination g_lo + i
          if (p.hubinfo && lig < rpg && node < p.node1) myhub = p.hubinfo[int64_t(mode) * p.n + node];
          const int beg = __shfl(myrp, glane0), end = __shfl(myrp, glane0 + rpg);
          int row = g_lo, nb = __shfl(myrp, glane0 + 1);
          float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
          auto add_hub = [&](int r) {   // a hub's own run is empty: its total sits in the row of its first chunk (group-uniform)
            if (!p.hubinfo) return;
            const int first = __shfl(myhub.x, glane0 + (r - g_lo)), cnt = __shfl(myhub.y, glane0 + (r - g_lo));
            if (cnt > 0) {
              const float4 ps = *reinterpret_cast<const float4 *>(p.partial + int64_t(first - p.chunk0) * p.d + coff);
              sum = make_float4(sum.x + ps.x, sum.y + ps.y, sum.z + ps.z, sum.w + ps.w);
            }
          };
          int cbase = beg;                                   // first slot of the record chunk held in myrec
          for (int s = beg; s < end; s += U) {
            if (s >= cbase + gs) {                           // group-uniform: next chunk of a long range
              cbase += gs;
              myrec = rec_chunk(cbase, end);
            }
            int rsrc[U], rtyp[U], rnrm[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
              const int from = glane0 + (((s + u < end) ? s + u : end - 1) - cbase);
              rsrc[u] = __shfl(myrec.x, from);
              rtyp[u] = __shfl(myrec.y, from);
              rnrm[u] = __shfl(myrec.z, from);
            }
            float4 xv[U], rv[U], ev[U];   // row addresses: one unsigned 32 x 32 -> 64 multiply-add each
#pragma unroll
            for (int u = 0; u < U; ++u) {
              xv[u] = *reinterpret_cast<const float4 *>(xb + uint64_t(uint32_t(rsrc[u])) * ldx32);
              rv[u] = *reinterpret_cast<const float4 *>(relb + uint64_t(uint32_t(rtyp[u])) * d32);   // graph edges never use the self-loop row
              const uint32_t erow = uint32_t(((s + u < end) ? s + u : end - 1) - ee_sub_mode);
              ev[u] = *reinterpret_cast<const float4 *>(eeb + uint64_t(erow) * d32);
            }
            if (!next_recs_issued) {   // behind this batch's row loads: the next stage's records (its row pointers are back)
              next_recs_issued = true;
              if (has_next) nrec = rec_chunk(__shfl(nrp, glane0), __shfl(nrp, glane0 + rpg));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
              if (s + u < end) {
                while (s + u >= nb) {  // group-uniform: the run of destination `row` is complete
                  add_hub(row);
                  {   // (lanes past the row width carry lane 0's columns: same address, same value, no branch)
                    float *dst = at + row * lda + coff;
                    *reinterpret_cast<float2 *>(dst) = make_float2(sum.x, sum.y);
                    *reinterpret_cast<float2 *>(dst + 2) = make_float2(sum.z, sum.w);
                  }
                  sum = make_float4(0.f, 0.f, 0.f, 0.f);
                  ++row;
                  nb = __shfl(myrp, glane0 + (row - g_lo) + 1);
                }
                const float4 m = f4mul(f4mul(xv[u], rv[u]), ev[u]);
                const float wgt = __int_as_float(rnrm[u]);
                sum = make_float4(sum.x + m.x * wgt, sum.y + m.y * wgt, sum.z + m.z * wgt, sum.w + m.w * wgt);
              }
            }
          }
          if (!next_recs_issued && has_next) nrec = rec_chunk(__shfl(nrp, glane0), __shfl(nrp, glane0 + rpg));
          currp = nrp;
          currec = nrec;
          for (; row < g_hi; ++row) {  // last run, then zero rows for destinations without slots
            add_hub(row);
            {
              float *dst = at + row * lda + coff;
              *reinterpret_cast<float2 *>(dst) = make_float2(sum.x, sum.y);
              *reinterpret_cast<float2 *>(dst + 2) = make_float2(sum.z, sum.w);
            }
            sum = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        } else {  // self loop: (x * loop_rel) * loop_edge, model.py:91-94,101
          const float4 lr = *reinterpret_cast<const float4 *>(p.loop_rel + coff);
          const float4 le = *reinterpret_cast<const float4 *>(p.loop_edge + coff);
          for (int row = g_lo; row < g_hi; ++row) {
            const int node = (r0 + row < p.node1) ? r0 + row : p.node1 - 1;  // rows past the range are computed, never stored
            const float4 v = f4mul(f4mul(*reinterpret_cast<const float4 *>(p.x + int64_t(node) * p.ldx + coff), lr), le);
            {
              float *dst = at + row * lda + coff;
              *reinterpret_cast<float2 *>(dst) = make_float2(v.x, v.y);
              *reinterpret_cast<float2 *>(dst + 2) = make_float2(v.z, v.w);
            }
          }
        }
        __syncthreads();  // end of stage: As[stage & 1] is complete
      }
    }
    __syncthreads();      // the drain stage (the MFMA waves multiply the last mode and leave its accumulators in Os)
    finalize(my_tiles - 1);
  } else {
    // The block's work per stage is 2 x NT (row tile, column tile) units. Every MFMA wave owns Q4 = NT/4 column tiles
    // with BOTH 16-row tiles (a weight fragment feeds two MFMAs), and the NT%4 left-over column tiles are shared out
    // as single units: NT = 13 -> waves 0,1 take row tile 0 / 1 of the 13th column tile (7,7,6,6 units instead of
    // 8,6,6,6); NT = 2 -> every wave takes one unit. Co-resident blocks (b, b+256 with two blocks per CU) swap the
    // wave pairs so that each SIMD's MFMA pipe sees 13 units per stage pair. Only NTW fragments per k-block are
    // needed, so they are prefetched THREE k-blocks ahead in registers — they come from L2 and one k-block of MFMAs
    // (~0.4 us) does not cover that latency.
    constexpr int Q4 = NT / 4, R4 = NT % 4;
    static_assert(R4 != 3, "column tile counts with NT % 4 == 3 are not instantiated");
    constexpr int QF = Q4 > 0 ? Q4 : 1;            // array extent for the full tiles (Q4 may be 0)
    const int wsel = wave ^ (((bid >> 8) & 1) << 1);
    const int ct0 = wsel * Q4;
    const bool has_half = R4 == 2 || (R4 == 1 && wsel < 2);
    const int hct = 4 * Q4 + (R4 == 2 ? (wsel >> 1) : 0);   // the shared column tile of this wave's single unit
    const bool hrt = (wsel & 1) != 0;                        // ... and its row tile
    const int fr = lane & 15, fq = lane >> 4;
    const int nkb3 = 3 * nkb;                      // k-blocks per tile over the three modes
    auto wload = [&](int g, int t) {               // fragment of k-block g (0 .. nkb3-1, mode-major), register slot t
      const int ct = (t < Q4) ? ct0 + t : (has_half ? hct : 0);
      return p.wp[(int64_t(g) * NT + ct) * 64 + lane];
    };
    float4 w0[NTW], w1[NTW], w2[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {                // in flight across the first barrier
      w0[t] = wload(0, t);
      w1[t] = wload(1 % nkb3, t);
      w2[t] = wload(2 % nkb3, t);
    }
    __syncthreads();      // stage 0 (the gather waves fetch the first mode of the first tile)
    // Loop state kept incrementally (no divisions in the k-block loop: two waves of a SIMD run this program in
    // lockstep, so whatever sits between two k-blocks' MFMAs idles the matrix pipe for both): (mode, kb) = k-block
    // being multiplied, gpre = k-block whose fragments are fetched next (three ahead, wrapping into the next tile).
    int stage = 0, mode = 0, kb = 0, gpre = 3 % nkb3;
    const int tail_steps = (p.d - (nkb - 1) * KS) >> 2;     // MFMA steps of a mode's last k-block (1..4)
    const float *arow = As + fr * lda + fq;
    // A fragments are read ONE k-block ahead into the register set of the next k-block (three sets rotating with
    // the weight sets, so no copies): steps past the row width read the neighbouring row or the staging tile
    // (inside the LDS block) and are not used.
    float aA[2][4], aB[2][4], aC[2][4];
#define MGCN_ALOAD(dst, st, kblock)                                                                          \
    {                                                                                                        \
      const float *ab_ = arow + ((st) & 1) * BM * lda + (kblock) * KS;                                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
        dst[0][i] = ab_[4 * i];                                                                              \
        dst[1][i] = ab_[16 * lda + 4 * i];                                                                   \
      }                                                                                                      \
    }
    MGCN_ALOAD(aA, 0, 0)
    for (int it = 0; it < my_tiles; ++it) {
      f32x4 acc[2][QF], acch = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < QF; ++t) acc[0][t] = acc[1][t] = f32x4{0.f, 0.f, 0.f, 0.f};

      auto accumulators_to_staging = [&]() {  // lane holds rows rt*16 + fq*4 + j of column ct*16 + fr (raw sums)
#pragma unroll
        for (int t = 0; t < Q4; ++t) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Os[(rt * 16 + fq * 4 + j) * LDO + (ct0 + t) * 16 + fr] = acc[rt][t][j];
          }
        }
        if (has_half) {
#pragma unroll
          for (int j = 0; j < 4; ++j) Os[((hrt ? 16 : 0) + fq * 4 + j) * LDO + hct * 16 + fr] = acch[j];
        }
      };

#define MGCN_STEP(wc, ac, i)                                                                                 \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t) {                                                      \
        const float bv_ = (i) == 0 ? wc[t].x : (i) == 1 ? wc[t].y : (i) == 2 ? wc[t].z : wc[t].w;           \
        if (t < Q4) {                                                                                        \
          acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[0][i], bv_, acc[0][t], 0, 0, 0);               \
          acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[1][i], bv_, acc[1][t], 0, 0, 0);               \
        } else if (has_half) {                                                                               \
          if (hrt) acch = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[1][i], bv_, acch, 0, 0, 0);                \
          else acch = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[0][i], bv_, acch, 0, 0, 0);                    \
        }                                                                                                    \
      }
      // one k-block: MFMAs on fragment set `wc`, which is then refilled with the k-block three ahead
#define MGCN_KBLOCK(wc, ac, an)                                                                              \
      {                                                                                                      \
        const bool last_ = kb == nkb - 1;                                                                    \
        const int nsteps_ = (p.ablate & 2) ? 0 : (last_ ? tail_steps : 4);                                   \
        if (!last_) MGCN_ALOAD(an, stage + mode, kb + 1)                                                     \
        if (nsteps_ > 0) { MGCN_STEP(wc, ac, 0) }                                                            \
        if (nsteps_ > 1) { MGCN_STEP(wc, ac, 1) }                                                            \
        if (nsteps_ > 2) { MGCN_STEP(wc, ac, 2) }                                                            \
        if (nsteps_ > 3) { MGCN_STEP(wc, ac, 3) }                                                            \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t) wc[t] = wload(gpre, t);                              \
        gpre = gpre + 1 == nkb3 ? 0 : gpre + 1;                                                              \
        if (last_) {                                                                                         \
          if (mode == 2) accumulators_to_staging();                                                          \
          __syncthreads(); /* end of stage: the next mode's tile is complete */                              \
          kb = 0;                                                                                            \
          mode = mode == 2 ? 0 : mode + 1;                                                                   \
          if (mode == 0) stage += 3;                                                                         \
          MGCN_ALOAD(an, stage + mode, 0)                                                                    \
        } else {                                                                                             \
          ++kb;                                                                                              \
        }                                                                                                    \
      }

      for (int g0 = 0; g0 < nkb3; g0 += 3) {   // nkb3 is a multiple of 3: fragment sets rotate w0 -> w1 -> w2
        MGCN_KBLOCK(w0, aA, aB)
        MGCN_KBLOCK(w1, aB, aC)
        MGCN_KBLOCK(w2, aC, aA)
      }
#undef MGCN_KBLOCK
#undef MGCN_STEP
#undef MGCN_ALOAD
    }
  }
}

}  // namespace

extern "C" size_t mgcn_packed_weights_bytes(int32_t dim_in, int32_t dim_out) {
  const int nkb = (dim_in + KS - 1) / KS, nt = pick_nt(dim_out);
  return size_t(3) * nkb * nt * 64 * sizeof(float4);
}

extern "C" int mgcn_pack_weights(int32_t dim_in, int32_t dim_out, const float *w_dev, float *wp_dev, size_t wp_bytes,
                                 void *stream) {
  MGCN_REQUIRE(dim_in > 0 && dim_out > 0 && w_dev && wp_dev, "pack_weights: bad arguments");
  MGCN_REQUIRE(wp_bytes >= mgcn_packed_weights_bytes(dim_in, dim_out) && mgcn::aligned16(wp_dev),
               "pack_weights: packed buffer too small or misaligned");
  const int nkb = (dim_in + KS - 1) / KS, nt = pick_nt(dim_out);
  const int total = 3 * nkb * nt * 64;
  hipLaunchKernelGGL(pack_w_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     w_dev, reinterpret_cast<float4 *>(wp_dev), dim_in, dim_out, nkb, nt);
  MGCN_CHECK_LAUNCH("pack_w_kernel");
  return MGCN_OK;
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
                                    float *rel_out_dev, void *stream) {
  MGCN_REQUIRE(num_nodes >= 0 && num_edges_half >= 0 && dim_in > 0 && dim_out > 0 && num_rel_rows > 0,
               "layer_fwd_fused: bad sizes");
  MGCN_REQUIRE(node_begin >= 0 && node_begin <= node_end && node_end <= num_nodes, "layer_fwd_fused: bad node range");
  MGCN_REQUIRE(num_nodes < (int64_t(1) << 31) - 64 && 2 * num_edges_half < (int64_t(1) << 31) - 1,
               "layer_fwd_fused: sizes exceed int32");
  MGCN_REQUIRE(rowptr_dev && x_dev && loop_rel_dev && loop_edge_dev && wp_dev && bn_mean_dev && bn_var_dev &&
                   bn_gamma_dev && bn_beta_dev && (out_dev || node_end == node_begin) && (rel_dev || num_rel_rows == 1) &&
                   (num_edges_half == 0 || rec_dev), "layer_fwd_fused: null pointer");
  MGCN_REQUIRE(ldx >= dim_in && ldo >= dim_out, "layer_fwd_fused: ldx/ldo too small");
  const bool aligned = mgcn::aligned16(x_dev) && mgcn::aligned16(rel_dev) && mgcn::aligned16(loop_rel_dev) &&
                       mgcn::aligned16(loop_edge_dev) && (!ee_dev || mgcn::aligned16(ee_dev)) &&
                       mgcn::aligned16(out_dev) && mgcn::aligned16(wp_dev) && ldx % 4 == 0 && ldo % 4 == 0;
  if (!aligned || dim_in % 4 != 0 || dim_in > 256 || dim_out % 4 != 0 || dim_out > 208 || !ee_dev || !ee_in_slot_order ||
      ldx >= (int64_t(1) << 31))
    return mgcn::fail(MGCN_EUNSUPPORTED, "layer_fwd_fused: needs 16-byte aligned operands, a per-edge table in slot order, "
                      "D %% 4 == 0, D <= 256, O %% 4 == 0, O <= 208 (got D=%d O=%d)", dim_in, dim_out);
  const int64_t num_chunks = chunk_end - chunk_begin;
  MGCN_REQUIRE(chunk_begin >= 0 && num_chunks >= 0 && chunk_end < (int64_t(1) << 31) &&
                   (num_chunks == 0 || (hubinfo_dev && chunks_dev && partial_dev && mgcn::aligned16(partial_dev))),
               "layer_fwd_fused: hub chunks need hubinfo / chunks / a 16-byte aligned partial buffer");
  const bool want_rel = rel_out_dev != nullptr && num_rel_rows > 1;
  MGCN_REQUIRE(!want_rel || (rels_weight_dev && rel_dev), "layer_fwd_fused: the relation projection needs rels_weight and rel");
  if (node_end == node_begin && !want_rel) return MGCN_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (num_chunks > 0 && node_end > node_begin) {
    if (int rc = mgcn::launch_hub_partials(num_nodes, dim_in, num_rel_rows, rec_dev, x_dev, ldx, rel_dev, loop_rel_dev,
                                           ee_dev, ee_in_slot_order, ee_sub_hub, chunks_dev, chunk_begin, chunk_end,
                                           partial_dev, stream))
      return rc;
  }
  const int nt = pick_nt(dim_out);
  FusedArgs p = {};
  p.hubinfo = num_chunks > 0 ? reinterpret_cast<const int2 *>(hubinfo_dev) : nullptr;
  p.partial = partial_dev;
  p.chunk0 = int32_t(chunk_begin);
  p.node0 = int32_t(node_begin); p.node1 = int32_t(node_end); p.ee_sub[0] = ee_sub_in; p.ee_sub[1] = ee_sub_out;
  p.rowptr = rowptr_dev; p.rec = reinterpret_cast<const int4 *>(rec_dev);
  p.x = x_dev; p.rel = rel_dev; p.loop_rel = loop_rel_dev; p.ee = ee_dev; p.loop_edge = loop_edge_dev;
  p.wp = reinterpret_cast<const float4 *>(wp_dev);
  p.bias = bias_dev; p.bn_mean = bn_mean_dev; p.bn_var = bn_var_dev; p.bn_gamma = bn_gamma_dev; p.bn_beta = bn_beta_dev;
  p.out = out_dev; p.ldx = ldx; p.ldo = ldo;
  p.n = int32_t(num_nodes); p.e = int32_t(num_edges_half); p.d = dim_in; p.o = dim_out; p.rel_rows = num_rel_rows;
  p.ee_slot_order = ee_in_slot_order; p.bn_eps = bn_eps;
  if (const char *ab = getenv("MGCN_FUSED_ABLATE")) p.ablate = atoi(ab);
  int gl = 3;  // lanes per gather group: smallest power of two >= D/4, at least 8 (so 32 rows cover <= 32 groups)
  while ((1 << gl) * 4 < dim_in) ++gl;
  p.gs_log2 = gl;
  const int ntiles = int((node_end - node_begin + BM - 1) / BM);
  const int lda = dim_in + 2, ldo_s = nt * 16 + 4;
  const size_t lds_bytes = (size_t(2) * BM * lda + size_t(BM) * ldo_s) * 4;
  int grid_i = 2 * 256;   // persistent: two 8-wave blocks per CU (128 VGPRs, <= 80 KiB LDS each)
  if (const char *g = getenv("MGCN_FUSED_GRID")) grid_i = atoi(g);
  const int main_grid = grid_i < ntiles ? grid_i : ntiles;
  if (want_rel) {   // a few extra workgroups project the relations (model.py:107): one or two items each when the tile
    // workgroups fill the chip anyway, else as many as fit beside them
    const int items = (num_rel_rows - 1) * ((dim_out + 63) / 64);
    const int cap = main_grid < 2 * 256 ? 2 * 256 - main_grid : 48;
    int rb = (items + 1) / 2;
    rb = rb < 1 ? 1 : (rb > cap ? cap : rb);
    p.rw = rels_weight_dev; p.rel_out = rel_out_dev; p.rel_blocks = rb;
  }
  const unsigned grid = unsigned(main_grid + p.rel_blocks);
  switch (nt) {
    case 2: hipLaunchKernelGGL((layer_fused_kernel<2>), dim3(grid), dim3(FUSED_THREADS), lds_bytes, st, p); break;
    case 4: hipLaunchKernelGGL((layer_fused_kernel<4>), dim3(grid), dim3(FUSED_THREADS), lds_bytes, st, p); break;
    case 8: hipLaunchKernelGGL((layer_fused_kernel<8>), dim3(grid), dim3(FUSED_THREADS), lds_bytes, st, p); break;
    default: hipLaunchKernelGGL((layer_fused_kernel<13>), dim3(grid), dim3(FUSED_THREADS), lds_bytes, st, p); break;
  }
  MGCN_CHECK_LAUNCH("layer_fused_kernel");
  return MGCN_OK;
}
