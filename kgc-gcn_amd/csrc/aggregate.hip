// Relation- and edge-typed neighbour aggregation over CSR-by-destination slots (gfx950).
//
// Forward: one LANE GROUP per (half, destination) walks that destination's slots in slot order
// and keeps the partial sum in registers; every lane owns VEC consecutive columns of the row, so
// a slot costs one broadcast 16-B record load plus three row loads (x gather, relation row,
// per-edge row — a pure stream when the table is laid out in slot order). One plain store per
// destination row: no atomics, sums reproducible and in the same order as a CPU scatter-add.
// Row width D = 100 f32 = 25 dwordx4: two 32-lane groups per 64-lane wave (SURVEY §7 "Row width").
//
// HBM-bound integer/float streaming work: nothing here is reshaped into a GEMM.
#include <hip/hip_runtime.h>

#include "mgcn_common.h"

namespace {

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  using type = float4;
  static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ float4 load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
  static __device__ __forceinline__ void store(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
  // agent-scope accesses (write-through / cache-bypassing: coherent across the XCDs' L2s inside a launch)
  static __device__ __forceinline__ float4 load_agent(const float *p) {
    return make_float4(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  static __device__ __forceinline__ void store_agent(float *p, float4 v) {
    __hip_atomic_store(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(p + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  static __device__ __forceinline__ float4 mul(float4 a, float4 b) {
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
  }
  static __device__ __forceinline__ float4 muls(float4 a, float s) {
    return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
  }
  static __device__ __forceinline__ float4 add(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
};
template <>
struct Vec<1> {
  using type = float;
  static __device__ __forceinline__ float zero() { return 0.f; }
  static __device__ __forceinline__ float load(const float *p) { return *p; }
  static __device__ __forceinline__ void store(float *p, float v) { *p = v; }
  static __device__ __forceinline__ float load_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  static __device__ __forceinline__ void store_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  static __device__ __forceinline__ float mul(float a, float b) { return a * b; }
  static __device__ __forceinline__ float muls(float a, float s) { return a * s; }
  static __device__ __forceinline__ float add(float a, float b) { return a + b; }
};

struct AggArgs {
  const int32_t *rowptr;  // [2][N+1]
  const int4 *rec;        // [2E] {src, type, norm bits, eid}
  const float *x;
  const float *rel;       // rows [0, rel_rows-1)
  const float *loop_rel;  // row rel_rows-1
  const float *ee;        // may be null
  const float *loop_edge;
  float *a;
  int64_t ldx, lda;
  int32_t n, e, d, rel_rows, ee_slot_order, modes;
  int32_t node0, nodes;  // destinations [node0, node0 + nodes) are processed
  const int2 *hubinfo;   // [2][N] (first chunk, chunk count) or null
  const int4 *chunks;    // [num_chunks] {slot begin, slot end, first chunk of the hub, chunks of the hub}
  float *partial;        // [num_chunks][D] chunk sums; after the fold, row `first chunk` holds the hub's total
  int32_t *hubcnt;       // [2][num_chunks] arrival counters of the fold (zero between launches), index = chunk - chunk0
  int32_t chunk0, nchunks;  // chunks [chunk0, chunk0 + nchunks) are in play; partial row = chunk - chunk0
  int64_t ee_sub_hub;       // table row of hub slot s = s - ee_sub_hub (slot-ordered table shards)
  int64_t ee_sub[2];        // ... and of a slot of half h: s - ee_sub[h] (0 with the whole table)
};

// The slot walk both forward kernels share: acc += sum over slots [first, end) of (x[src] * rel[type] [* ee[slot]]) * norm, in slot
// order, by one lane group. The group's lanes fetch 2 * GS records with two loads and hand them round by lane shuffles; the rows of
// batch b + 1 are in flight while batch b is added up (two register sets). The loop body is free of branches around loads (slot index
// clamped to the run's last slot, column clamped to the row's last chunk, the relation / self-loop row chosen by offset; only the adds
// are predicated), so the compiler counts the outstanding loads (`vmcnt(N)`) instead of draining them at every merge of two paths.
template <int VEC, int CPL, int U, bool ROLL>
__device__ __forceinline__ void walk_slots(const AggArgs &p, int first, int end, int64_t ee_sub, const int (&col)[CPL],
                                           int lane_in_group, int gs, typename Vec<VEC>::type (&acc)[CPL]) {
  using V = Vec<VEC>;
  using T = typename V::type;
  const int last = end - 1;
  const int64_t loop_off = p.loop_rel - p.rel;    // the self-loop row as an offset from the relation table
  const float *ee = p.ee ? p.ee : p.x;            // no per-edge table: a valid address, the value is not used
  const bool has_ee = p.ee != nullptr;
  for (int beg = first; beg < end; beg += 2 * gs) {
    const int stop = min(beg + 2 * gs, end);
    const int nb = (stop - beg + U - 1) / U;
    const int4 win0 = p.rec[min(beg + lane_in_group, last)], win1 = p.rec[min(beg + gs + lane_in_group, last)];
    T xa[U][CPL], ra[U][CPL], ea[U][CPL], xb[U][CPL], rb[U][CPL], eb[U][CPL];
    float wa[U], wb[U];
    auto issue = [&](T (&xv)[U][CPL], T (&rv)[U][CPL], T (&ev)[U][CPL], float (&wt)[U], int b) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = min(beg + b * U + u, last);
        const int idx = min(s - beg, 2 * gs - 1);
        const bool hi = idx >= gs;
        int4 r;
        r.x = __shfl(hi ? win1.x : win0.x, idx, gs); r.y = __shfl(hi ? win1.y : win0.y, idx, gs);
        r.z = __shfl(hi ? win1.z : win0.z, idx, gs); r.w = __shfl(hi ? win1.w : win0.w, idx, gs);
        wt[u] = __int_as_float(r.z);
        const float *xr = p.x + int64_t(r.x) * p.ldx;
        const float *rr = p.rel + ((r.y < p.rel_rows - 1) ? int64_t(r.y) * p.d : loop_off);
        const float *er = ee + (has_ee ? (p.ee_slot_order ? int64_t(s) - ee_sub : int64_t(r.w)) * p.d : int64_t(0));
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          xv[u][c] = V::load(xr + col[c]);
          rv[u][c] = V::load(rr + col[c]);
          ev[u][c] = V::load(er + col[c]);
        }
      }
    };
    auto consume = [&](T (&xv)[U][CPL], T (&rv)[U][CPL], T (&ev)[U][CPL], float (&wt)[U], int b) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (beg + b * U + u < stop) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            T m = V::mul(xv[u][c], rv[u][c]);
            if (has_ee) m = V::mul(m, ev[u][c]);
            acc[c] = V::add(acc[c], V::muls(m, wt[u]));
          }
        }
      }
    };
    if (ROLL) {     // long runs: the next batch's rows are in flight while this one is added up
      issue(xa, ra, ea, wa, 0);
      for (int b = 0; b < nb; b += 2) {
        issue(xb, rb, eb, wb, b + 1);
        consume(xa, ra, ea, wa, b);
        issue(xa, ra, ea, wa, b + 2);
        consume(xb, rb, eb, wb, b + 1);
      }
    } else {        // short runs (a batch or two): no loads past the run's last batch, fewer registers, more waves per SIMD
      for (int b = 0; b < nb; ++b) {
        issue(xa, ra, ea, wa, b);
        consume(xa, ra, ea, wa, b);
      }
    }
  }
}

// GS lanes per group (power of two <= 64), CPL column chunks per lane, U slots per batch (two batches in flight per group): one group
// per (mode, destination); a short run (WN18RR: 2.1 slots on average) costs three dependent memory round trips in total — row
// pointers, records, rows.
template <int VEC, int CPL, int U, bool ROLL = false>
__global__ __launch_bounds__(256) void agg_fwd_kernel(AggArgs p, int gs_log2) {
  using V = Vec<VEC>;
  using T = typename V::type;
  const int gs = 1 << gs_log2;
  const int lane_in_group = threadIdx.x & (gs - 1);
  const int64_t item = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> gs_log2;
  if (item >= int64_t(p.modes) * p.nodes) return;
  const int mode = int(item / p.nodes);
  const int node = p.node0 + int(item - int64_t(mode) * p.nodes);
  const int nchunk = p.d / VEC;

  if (mode == 2) {  // self loop: (x * rel[last]) * loop_edge, model.py:91-94,101
    const float *xr = p.x + int64_t(node) * p.ldx;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int ch = lane_in_group + c * gs;
      if (ch < nchunk)
        V::store(p.a + int64_t(node) * p.lda + 2 * p.d + ch * VEC,
                 V::mul(V::mul(V::load(xr + ch * VEC), V::load(p.loop_rel + ch * VEC)), V::load(p.loop_edge + ch * VEC)));
    }
    return;
  }

  int col[CPL];                                   // this lane's columns (clamped: lanes past the row load its last chunk and store nothing)
#pragma unroll
  for (int c = 0; c < CPL; ++c) col[c] = min(lane_in_group + c * gs, nchunk - 1) * VEC;
  T acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = V::zero();
  const int32_t *rp = p.rowptr + int64_t(mode) * (p.n + 1);   // absolute slot positions
  walk_slots<VEC, CPL, U, ROLL>(p, rp[node], rp[node + 1], p.ee_sub[mode], col, lane_in_group, gs, acc);
  if (p.hubinfo) {  // a hub's own segment above is empty: its total was folded into the row of its first chunk
    const int2 hi = p.hubinfo[int64_t(mode) * p.n + node];
    if (hi.y > 0) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) acc[c] = V::add(acc[c], V::load(p.partial + int64_t(hi.x - p.chunk0) * p.d + col[c]));
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lane_in_group + c * gs;
    if (ch < nchunk) V::store(p.a + int64_t(node) * p.lda + mode * p.d + ch * VEC, acc[c]);
  }
}

constexpr int kFoldSpan = 16;

// The hub fold inside the pre-pass launch. A group that has stored its row arrives at a counter; the LAST group to arrive (whichever it
// is) adds the rows up in row order, so the summation tree depends only on the hub's chunk count: spans of kFoldSpan chunk sums into the
// span's first row, then the span totals (stride kFoldSpan) into the hub's first row. The XCDs' L2s are not coherent with each other
// inside a launch, and agent-scope fences (L2 write-back + invalidate per group) double the launch's time by evicting the gathered rows;
// so the chunk-sum rows themselves are written and read with agent-scope accesses (write-through stores, cache-bypassing loads), a
// group waits for its stores to be acknowledged before its (memory-side) counter increment, and the last arriver's loads depend on the
// increment's result. The last arriver puts the counter back to zero for the next launch.
__device__ __forceinline__ bool hub_arrive(int32_t *counter, int expected, int lane_in_group, int gs) {
  // This group's write-through (sc1) row stores must be ACKNOWLEDGED before the counter moves: the increment travels to another
  // channel than the rows and may otherwise become visible first. A workgroup-scope release fence emits nothing for global memory
  // on gfx950 (checked in the ISA), so the wait is written out; tests/test_host.py greps the device ISA for it.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int old = 0;
  if (lane_in_group == 0) old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __shfl(old, 0, gs);
  if (old != expected - 1) return false;
  if (lane_in_group == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("" ::: "memory");                               // (compiler ordering: the sc1 row loads stay behind the increment,
  return true;                                                 //  whose returned value the branch above depends on)
}

template <int VEC, int CPL>
__device__ __forceinline__ void hub_fold_rows(float *rows, int count, int64_t rstride, const int (&col)[CPL], int lane_in_group, int gs,
                                              int nchunk) {
  using V = Vec<VEC>;
  using T = typename V::type;
  constexpr int UF = 16 / CPL > 0 ? 16 / CPL : 1;
  T tot[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) tot[c] = V::load_agent(rows + col[c]);
  for (int k = 1; k < count; k += UF) {
    T v[UF][CPL];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int kk = min(k + u, count - 1);
#pragma unroll
      for (int c = 0; c < CPL; ++c) v[u][c] = V::load_agent(rows + int64_t(kk) * rstride + col[c]);
    }
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      if (k + u < count) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) tot[c] = V::add(tot[c], v[u][c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c)
    if (lane_in_group + c * gs < nchunk) V::store_agent(rows + col[c], tot[c]);
}

// Hub pre-pass: one lane group per chunk of a hub destination's slots; same arithmetic and slot order (walk_slots). There are few
// groups (~4 waves per CU on the FB15k-237 shape), each a chain of dependent round trips kept short by the walk's record windows and
// its rolling prefetch.
template <int VEC, int CPL, int U, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void agg_hub_kernel(AggArgs p, int gs_log2) {
  using V = Vec<VEC>;
  using T = typename V::type;
  const int gs = 1 << gs_log2;
  const int lane_in_group = threadIdx.x & (gs - 1);
  const int64_t chunk = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> gs_log2;
  if (chunk >= p.nchunks) return;
  const int nchunk = p.d / VEC;
  const int4 range = p.chunks[p.chunk0 + chunk];
  int col[CPL];                                   // this lane's columns (clamped: lanes past the row load its last chunk and store nothing)
#pragma unroll
  for (int c = 0; c < CPL; ++c) col[c] = min(lane_in_group + c * gs, nchunk - 1) * VEC;
  T acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = V::zero();
  walk_slots<VEC, CPL, U, true>(p, range.x, range.y, p.ee_sub_hub, col, lane_in_group, gs, acc);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lane_in_group + c * gs;
    if (ch < nchunk) V::store_agent(p.partial + chunk * p.d + ch * VEC, acc[c]);
  }
  const int w = range.w;                                         // chunks of this hub (the same for a group's lanes)
  if (w < 2) return;
  const int hub0 = range.z - p.chunk0;                           // row of the hub's first chunk
  const int j0 = int(chunk) - hub0;                              // this chunk's index inside its hub
  float *hub_rows = p.partial + int64_t(hub0) * p.d;
  if (w > kFoldSpan) {
    const int sp0 = j0 - j0 % kFoldSpan, spn = min(kFoldSpan, w - sp0);
    if (!hub_arrive(p.hubcnt + hub0 + sp0, spn, lane_in_group, gs)) return;
    hub_fold_rows<VEC, CPL>(hub_rows + int64_t(sp0) * p.d, spn, p.d, col, lane_in_group, gs, nchunk);
    if (!hub_arrive(p.hubcnt + p.nchunks + hub0, (w + kFoldSpan - 1) / kFoldSpan, lane_in_group, gs)) return;
    hub_fold_rows<VEC, CPL>(hub_rows, (w + kFoldSpan - 1) / kFoldSpan, int64_t(kFoldSpan) * p.d, col, lane_in_group, gs, nchunk);
  } else {
    if (!hub_arrive(p.hubcnt + p.nchunks + hub0, w, lane_in_group, gs)) return;
    hub_fold_rows<VEC, CPL>(hub_rows, w, p.d, col, lane_in_group, gs, nchunk);
  }
}

// Hub fold as launches of its own (the backward pre-pass of gx uses it), two levels (a hub of thousands of slots has hundreds of chunk sums: one workgroup adding them all is a chain of
// dependent round trips). Level 1 (span = kFoldSpan, stride = 1): the workgroup of every kFoldSpan-th chunk of a hub adds
// the next kFoldSpan chunk sums into its own row. Level 2 (stride = kFoldSpan): the workgroup of the hub's FIRST chunk adds
// those rows into the first. Lane group j adds rows j, j+J, j+2J, ... of its span in that order (U loads in flight), then
// group 0 adds the J group sums in group order: a fixed summation tree that depends only on the hub's chunk count.
struct FoldArgs {
  const int4 *chunks;
  float *partial;
  int32_t chunk0, d;
  int32_t level;   // 1 or 2
};

template <int VEC, int CPL>
__global__ __launch_bounds__(256) void agg_hub_fold_kernel(FoldArgs p, int gs_log2) {
  using V = Vec<VEC>;
  using T = typename V::type;
  extern __shared__ float red[];  // [J][D]
  const int4 me = p.chunks[p.chunk0 + blockIdx.x];
  const int j0 = p.chunk0 + int(blockIdx.x) - me.z;             // this chunk's index inside its hub (workgroup-uniform)
  if (me.w < 2) return;
  int count, stride;
  if (p.level == 1) {
    if (j0 % kFoldSpan != 0 || me.w <= kFoldSpan) return;       // (a hub of at most kFoldSpan chunks is folded by level 2 alone)
    count = me.w - j0 < kFoldSpan ? me.w - j0 : kFoldSpan;
    stride = 1;
  } else {
    if (j0 != 0) return;
    stride = me.w <= kFoldSpan ? 1 : kFoldSpan;
    count = (me.w + stride - 1) / stride;
  }
  if (count < 2) return;
  float *rows = p.partial + int64_t(me.z - p.chunk0 + j0) * p.d;   // first row of the span
  const int64_t rstride = int64_t(stride) * p.d;
  constexpr int U = 4;
  const int gs = 1 << gs_log2, groups = 256 >> gs_log2;
  const int grp = threadIdx.x >> gs_log2, lane_in_group = threadIdx.x & (gs - 1);
  const int nchunk = p.d / VEC;
  T acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = V::zero();
  for (int k = grp; k < count; k += groups * U) {
    T v[U][CPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k + u * groups;
      if (kk >= count) continue;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int ch = lane_in_group + c * gs;
        if (ch < nchunk) v[u][c] = V::load(rows + int64_t(kk) * rstride + ch * VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k + u * groups >= count) continue;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int ch = lane_in_group + c * gs;
        if (ch < nchunk) acc[c] = V::add(acc[c], v[u][c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lane_in_group + c * gs;
    if (ch < nchunk) V::store(red + grp * p.d + ch * VEC, acc[c]);
  }
  __syncthreads();
  if (grp != 0) return;
  const int used = count < groups ? count : groups;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lane_in_group + c * gs;
    if (ch >= nchunk) continue;
    T tot = V::load(red + ch * VEC);
    for (int j = 1; j < used; ++j) tot = V::add(tot, V::load(red + j * p.d + ch * VEC));
    V::store(rows + ch * VEC, tot);
  }
}

// ---------------------------------------------------------------------------------------------
// Backward
// ---------------------------------------------------------------------------------------------
struct BwdArgs {
  const int4 *rec;
  const int32_t *slot_dst;
  const int32_t *rowptr;     // [2][N+1] destination runs (absolute slots)
  const int32_t *mirror;     // [2E] slot -> slot of the reverse edge
  const int2 *hubinfo;       // [2][N] or null
  const int4 *chunks;        // hub chunks of the whole graph
  float *hub_ws;             // [nchunks_hub][D] chunk sums of gx
  int32_t nchunks_hub;
  const int32_t *typeptr;    // [T+1]
  const int32_t *typeslots;  // [2E]
  const float *x, *rel, *ee, *g;
  float *gx, *gee, *grel, *ws;
  int64_t ldx, ldg;
  int32_t n, e, d, rel_rows, nchunks_type;
};

constexpr int kTypeChunk = 16;  // slots per partial sum of the by-type reduction (short chunks = many lane groups in flight)

// gee[slot] = (g[dst, half] * norm) * x[src] * rel[type]; one group per slot, fully streamed store.
template <int VEC, int CPL>
__global__ __launch_bounds__(256) void agg_bwd_gee_kernel(BwdArgs p, int gs_log2) {
  using V = Vec<VEC>;
  using T = typename V::type;
  const int gs = 1 << gs_log2;
  const int lig = threadIdx.x & (gs - 1);
  const int64_t slot = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> gs_log2;
  if (slot >= 2 * int64_t(p.e)) return;
  const int nchunk = p.d / VEC;
  const int4 r = p.rec[slot];
  const float w = __int_as_float(r.z);
  const int sd = p.slot_dst[slot];
  const int half = (sd >> 31) & 1;
  const float *gr = p.g + int64_t(sd & 0x7fffffff) * p.ldg + half * p.d;
  const float *xr = p.x + int64_t(r.x) * p.ldx;
  const float *rr = p.rel + int64_t(r.y) * p.d;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lig + c * gs;
    if (ch < nchunk) {
      T gm = V::muls(V::load(gr + ch * VEC), w);
      V::store(p.gee + slot * p.d + ch * VEC, V::mul(V::mul(gm, V::load(xr + ch * VEC)), V::load(rr + ch * VEC)));
    }
  }
}

// By-source sums through the mirror map: the edges that leave `node` in half hq are the reverses of the slots p that
// enter it in half 1-hq; reverse slot q = mirror[p] has dst_q = src_p. Contribution of q:
// ((g[dst_q, hq] * norm_q) * rel[type_q]) * ee[q]. U slots per batch: indices, then reverse records, then 3*U rows.
template <int VEC, int CPL, int U = (CPL == 1 ? 4 : (CPL == 2 ? 2 : 1))>
__device__ __forceinline__ void gx_walk(const BwdArgs &p, int beg, int end, int hq, int lig, int gs,
                                        typename Vec<VEC>::type (&acc)[CPL]) {
  using V = Vec<VEC>;
  using T = typename V::type;
  const int nchunk = p.d / VEC;
  const int32_t *src_of = reinterpret_cast<const int32_t *>(p.rec);   // rec[s].src = word 4*s
  for (int s = beg; s < end; s += U) {
    int dq[U], q[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s + u < end) {
        dq[u] = src_of[4 * int64_t(s + u)];
        q[u] = p.mirror[s + u];
      }
    int4 rq[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s + u < end) rq[u] = p.rec[q[u]];
    T gv[U][CPL], rv[U][CPL], ev[U][CPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s + u >= end) continue;
      const float *gr = p.g + int64_t(dq[u]) * p.ldg + hq * p.d;
      const float *rr = p.rel + int64_t(rq[u].y) * p.d;
      const float *er = p.ee ? p.ee + int64_t(q[u]) * p.d : nullptr;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int ch = lig + c * gs;
        if (ch < nchunk) {
          gv[u][c] = V::load(gr + ch * VEC);
          rv[u][c] = V::load(rr + ch * VEC);
          if (er) ev[u][c] = V::load(er + ch * VEC);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (s + u >= end) continue;
      const float w = __int_as_float(rq[u].z);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int ch = lig + c * gs;
        if (ch < nchunk) {
          T m = V::mul(V::muls(gv[u][c], w), rv[u][c]);
          if (p.ee) m = V::mul(m, ev[u][c]);
          acc[c] = V::add(acc[c], m);
        }
      }
    }
  }
}

// gx[node]: reverse edges in half 0 (destination run of half 1), then half 1; a hub's run is empty and its folded
// chunk sums are added instead.
template <int VEC, int CPL, int U>
__device__ __forceinline__ void gx_node(const BwdArgs &p, int gs_log2) {
  using V = Vec<VEC>;
  using T = typename V::type;
  const int gs = 1 << gs_log2;
  const int lig = threadIdx.x & (gs - 1);
  const int64_t node = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> gs_log2;
  if (node >= p.n) return;
  const int nchunk = p.d / VEC;
  T acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = V::zero();
  for (int hq = 0; hq < 2; ++hq) {
    const int h = 1 - hq;
    const int32_t *rp = p.rowptr + int64_t(h) * (p.n + 1);
    gx_walk<VEC, CPL, U>(p, rp[node], rp[node + 1], hq, lig, gs, acc);
    if (p.hubinfo) {
      const int2 hi = p.hubinfo[int64_t(h) * p.n + node];
      if (hi.y > 0) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int ch = lig + c * gs;
          if (ch < nchunk) acc[c] = V::add(acc[c], V::load(p.hub_ws + int64_t(hi.x) * p.d + ch * VEC));
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lig + c * gs;
    if (ch < nchunk) V::store(p.gx + node * p.d + ch * VEC, acc[c]);
  }
}

template <int VEC, int CPL>
__global__ __launch_bounds__(256) void agg_bwd_gx_kernel(BwdArgs p, int gs_log2) {
  gx_node<VEC, CPL, (CPL == 1 ? 4 : (CPL == 2 ? 2 : 1))>(p, gs_log2);
}

// Short runs (fewer than four slots per destination and half on average): two slots in flight and the smaller
// register footprint's occupancy, as in the forward aggregation.
template <int VEC, int CPL>
__global__ __launch_bounds__(256) void agg_bwd_gx_short_kernel(BwdArgs p, int gs_log2) {
  gx_node<VEC, CPL, (CPL == 1 ? 2 : 1)>(p, gs_log2);
}

// Hub pre-pass of gx: one lane group per hub chunk (then agg_hub_fold_kernel).
template <int VEC, int CPL>
__global__ __launch_bounds__(256) void agg_bwd_gx_hub_kernel(BwdArgs p, int gs_log2) {
  using V = Vec<VEC>;
  using T = typename V::type;
  const int gs = 1 << gs_log2;
  const int lig = threadIdx.x & (gs - 1);
  const int64_t chunk = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> gs_log2;
  if (chunk >= p.nchunks_hub) return;
  const int nchunk = p.d / VEC;
  const int4 range = p.chunks[chunk];
  const int hq = 1 - ((p.slot_dst[range.x] >> 31) & 1);   // the chunk's slots enter the hub in half 1-hq
  T acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = V::zero();
  gx_walk<VEC, CPL>(p, range.x, range.y, hq, lig, gs, acc);
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lig + c * gs;
    if (ch < nchunk) V::store(p.hub_ws + chunk * p.d + ch * VEC, acc[c]);
  }
}

// By-type reduction, stage 1: entries [c*CH, (c+1)*CH) of typeslots; every maximal run of one type
// inside the chunk is summed in entry order and stored to a partial row: row `c` if the run starts at
// the chunk boundary (and not at a type boundary), row nchunks + t if it starts at typeptr[t].
template <int VEC, int CPL>
__global__ __launch_bounds__(256) void agg_bwd_grel_partial_kernel(BwdArgs p, int gs_log2) {
  using V = Vec<VEC>;
  using T = typename V::type;
  const int gs = 1 << gs_log2;
  const int lig = threadIdx.x & (gs - 1);
  const int64_t chunk = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> gs_log2;
  if (chunk >= p.nchunks_type) return;
  const int nchunk = p.d / VEC;
  const int64_t e2 = 2 * int64_t(p.e);
  const int64_t lo = chunk * kTypeChunk;
  const int64_t hi = (lo + kTypeChunk < e2) ? lo + kTypeChunk : e2;
  T acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = V::zero();
  int cur_type = p.rec[p.typeslots[lo]].y;
  int64_t out_row = (p.typeptr[cur_type] == lo) ? int64_t(p.nchunks_type) + cur_type : chunk;
  for (int64_t i = lo; i < hi; ++i) {
    const int slot = p.typeslots[i];
    const int4 r = p.rec[slot];
    if (r.y != cur_type) {  // uniform across the group
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int ch = lig + c * gs;
        if (ch < nchunk) V::store(p.ws + out_row * p.d + ch * VEC, acc[c]);
        acc[c] = V::zero();
      }
      cur_type = r.y;
      out_row = int64_t(p.nchunks_type) + cur_type;  // a new type inside a chunk starts at typeptr[type]
    }
    const float w = __int_as_float(r.z);
    const int sd = p.slot_dst[slot];
    const float *gr = p.g + int64_t(sd & 0x7fffffff) * p.ldg + ((sd >> 31) & 1) * p.d;
    const float *xr = p.x + int64_t(r.x) * p.ldx;
    const float *er = p.ee ? p.ee + int64_t(slot) * p.d : nullptr;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int ch = lig + c * gs;
      if (ch < nchunk) {
        T m = V::mul(V::muls(V::load(gr + ch * VEC), w), V::load(xr + ch * VEC));
        if (er) m = V::mul(m, V::load(er + ch * VEC));
        acc[c] = V::add(acc[c], m);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lig + c * gs;
    if (ch < nchunk) V::store(p.ws + out_row * p.d + ch * VEC, acc[c]);
  }
}

// Stage 1 of the by-type reduction AND the per-edge gradient in one pass over the slots in type order (the training step asks
// for both; a third of the backward's bytes are saved: g and x rows are gathered once instead of twice). One lane group per chunk
// of kTypeChunk entries of typeslots, U slots per batch: slot ids, then records + destinations, then the 3 U rows, all in
// flight together (the unfused stage 1 walked its 16 slots as 16 chains of three dependent round trips). Per slot
// P = (g[dst, half] * norm) * x[src]; gee[slot] = P * rel[type] (agg_bwd_gee_kernel's value, bit for bit) and the chunk's partial
// sum takes P * ee[slot] in entry order (agg_bwd_grel_partial_kernel's sum, bit for bit).
template <int VEC, int CPL>
__global__ __launch_bounds__(256) void agg_bwd_gee_grel_kernel(BwdArgs p, int gs_log2) {
  using V = Vec<VEC>;
  using T = typename V::type;
  constexpr int U = CPL == 1 ? 4 : (CPL == 2 ? 2 : 1);
  const int gs = 1 << gs_log2;
  const int lig = threadIdx.x & (gs - 1);
  const int64_t chunk = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> gs_log2;
  if (chunk >= p.nchunks_type) return;
  const int nchunk = p.d / VEC;
  const int64_t e2 = 2 * int64_t(p.e);
  const int64_t lo = chunk * kTypeChunk;
  const int64_t hi = (lo + kTypeChunk < e2) ? lo + kTypeChunk : e2;
  T acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = V::zero();
  int cur_type = p.rec[p.typeslots[lo]].y;
  int64_t out_row = (p.typeptr[cur_type] == lo) ? int64_t(p.nchunks_type) + cur_type : chunk;
  for (int64_t i = lo; i < hi; i += U) {
    int slot[U], sd[U];
    int4 r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) slot[u] = p.typeslots[(i + u < hi) ? i + u : hi - 1];   // (clamped: no branch around a load)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      r[u] = p.rec[slot[u]];
      sd[u] = p.slot_dst[slot[u]];
    }
    T gv[U][CPL], xv[U][CPL], ev[U][CPL], rv[U][CPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float *gr = p.g + int64_t(sd[u] & 0x7fffffff) * p.ldg + ((sd[u] >> 31) & 1) * p.d;
      const float *xr = p.x + int64_t(r[u].x) * p.ldx;
      const float *er = p.ee + int64_t(slot[u]) * p.d;
      const float *rr = p.rel + int64_t(r[u].y) * p.d;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int ch = (lig + c * gs < nchunk) ? lig + c * gs : nchunk - 1;   // (clamped: lanes past the row store nothing)
        gv[u][c] = V::load(gr + ch * VEC);
        xv[u][c] = V::load(xr + ch * VEC);
        ev[u][c] = V::load(er + ch * VEC);
        rv[u][c] = V::load(rr + ch * VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i + u < hi) {
        if (r[u].y != cur_type) {  // uniform across the group: a new type inside the chunk starts at typeptr[type]
#pragma unroll
          for (int c = 0; c < CPL; ++c) {
            const int ch = lig + c * gs;
            if (ch < nchunk) V::store(p.ws + out_row * p.d + ch * VEC, acc[c]);
            acc[c] = V::zero();
          }
          cur_type = r[u].y;
          out_row = int64_t(p.nchunks_type) + cur_type;
        }
        const float w = __int_as_float(r[u].z);
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const int ch = lig + c * gs;
          if (ch < nchunk) {
            const T pm = V::mul(V::muls(gv[u][c], w), xv[u][c]);
            V::store(p.gee + int64_t(slot[u]) * p.d + ch * VEC, V::mul(pm, rv[u][c]));
            acc[c] = V::add(acc[c], V::mul(pm, ev[u][c]));
          }
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lig + c * gs;
    if (ch < nchunk) V::store(p.ws + out_row * p.d + ch * VEC, acc[c]);
  }
}

// Stage 2: grel[t] = partial[nchunks + t] + partial[c] for every chunk boundary strictly inside t's range. One
// workgroup per relation row: lane group j adds the chunk rows j, j + J, j + 2J, ... of the range in that order (four
// loads in flight), then group 0 adds the head row and the J group sums in group order — a fixed summation tree.
template <int VEC, int CPL>
__global__ __launch_bounds__(256) void agg_bwd_grel_final_kernel(BwdArgs p, int gs_log2) {
  using V = Vec<VEC>;
  using T = typename V::type;
  extern __shared__ float red[];  // [J][D]
  constexpr int U = 16 / CPL > 0 ? 16 / CPL : 1;   // rows in flight per lane group (one workgroup per relation row: few workgroups,
                                                   // each a chain of round trips — WN18RR: 22 of them with ~490 chunk rows each)
  const int gs = 1 << gs_log2, groups = 256 >> gs_log2;
  const int grp = threadIdx.x >> gs_log2, lig = threadIdx.x & (gs - 1);
  const int64_t t = blockIdx.x;
  const int nchunk = p.d / VEC;
  const int64_t lo = p.typeptr[t], hi = p.typeptr[t + 1];
  const int64_t cb0 = lo / kTypeChunk + 1;                                   // first chunk boundary inside the range
  const int64_t ncb = hi > lo ? (hi - 1) / kTypeChunk - cb0 + 1 : 0;         // chunk rows to add (may be <= 0)
  T acc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) acc[c] = V::zero();
  for (int64_t k = grp; k < ncb; k += int64_t(groups) * U) {
    T v[U][CPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t kk = k + int64_t(u) * groups;
      if (kk >= ncb) continue;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int ch = lig + c * gs;
        if (ch < nchunk) v[u][c] = V::load(p.ws + (cb0 + kk) * p.d + ch * VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (k + int64_t(u) * groups >= ncb) continue;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int ch = lig + c * gs;
        if (ch < nchunk) acc[c] = V::add(acc[c], v[u][c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lig + c * gs;
    if (ch < nchunk) V::store(red + grp * p.d + ch * VEC, acc[c]);
  }
  __syncthreads();
  if (grp != 0) return;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lig + c * gs;
    if (ch >= nchunk) continue;
    T tot = V::zero();
    if (hi > lo) {
      tot = V::load(p.ws + (int64_t(p.nchunks_type) + t) * p.d + ch * VEC);
      for (int j = 0; j < groups; ++j) tot = V::add(tot, V::load(red + j * p.d + ch * VEC));
    }
    V::store(p.grel + t * p.d + ch * VEC, tot);
  }
}

struct Geometry {
  int vec, cpl, gs_log2;
};

// lanes per group: the smallest power of two covering the row's chunks, capped at one wave.
bool pick_geometry(int d, bool all_aligned, Geometry *g) {
  g->vec = (d % 4 == 0 && all_aligned) ? 4 : 1;
  const int nchunk = d / g->vec;
  int gl = 2;  // at least 4 lanes
  while ((1 << gl) < nchunk && gl < 6) ++gl;
  g->gs_log2 = gl;
  const int cpl = (nchunk + (1 << gl) - 1) >> gl;
  g->cpl = cpl <= 1 ? 1 : cpl <= 2 ? 2 : cpl <= 4 ? 4 : cpl <= 8 ? 8 : 0;
  return g->cpl != 0;
}

#define MGCN_LAUNCH_GEOM(KERNEL, ARGS, ITEMS, GEOM, STREAM)                                               \
  do {                                                                                                    \
    const int64_t threads_ = int64_t(ITEMS) << (GEOM).gs_log2;                                            \
    const unsigned grid_ = unsigned((threads_ + 255) / 256);                                              \
    if (grid_ > 0) {                                                                                      \
      hipStream_t s_ = static_cast<hipStream_t>(STREAM);                                                  \
      if ((GEOM).vec == 4) {                                                                              \
        switch ((GEOM).cpl) {                                                                             \
          case 1: hipLaunchKernelGGL((KERNEL<4, 1>), dim3(grid_), dim3(256), 0, s_, ARGS, (GEOM).gs_log2); break; \
          case 2: hipLaunchKernelGGL((KERNEL<4, 2>), dim3(grid_), dim3(256), 0, s_, ARGS, (GEOM).gs_log2); break; \
          case 4: hipLaunchKernelGGL((KERNEL<4, 4>), dim3(grid_), dim3(256), 0, s_, ARGS, (GEOM).gs_log2); break; \
          default: hipLaunchKernelGGL((KERNEL<4, 8>), dim3(grid_), dim3(256), 0, s_, ARGS, (GEOM).gs_log2); break; \
        }                                                                                                 \
      } else {                                                                                            \
        switch ((GEOM).cpl) {                                                                             \
          case 1: hipLaunchKernelGGL((KERNEL<1, 1>), dim3(grid_), dim3(256), 0, s_, ARGS, (GEOM).gs_log2); break; \
          case 2: hipLaunchKernelGGL((KERNEL<1, 2>), dim3(grid_), dim3(256), 0, s_, ARGS, (GEOM).gs_log2); break; \
          case 4: hipLaunchKernelGGL((KERNEL<1, 4>), dim3(grid_), dim3(256), 0, s_, ARGS, (GEOM).gs_log2); break; \
          default: hipLaunchKernelGGL((KERNEL<1, 8>), dim3(grid_), dim3(256), 0, s_, ARGS, (GEOM).gs_log2); break; \
        }                                                                                                 \
      }                                                                                                   \
    }                                                                                                     \
  } while (0)

void launch_fold(const Geometry &g, const int4 *chunks, float *partial, int32_t chunk0, int32_t dim, int64_t num_chunks,
                 hipStream_t st) {
  FoldArgs f = {chunks, partial, chunk0, dim, 1};
  const size_t lds = size_t(256 >> g.gs_log2) * size_t(dim) * sizeof(float);
#define MGCN_FOLD_CASE(V_, C_)                                                                                          \
  do {                                                                                                                  \
    f.level = 1;                                                                                                        \
    hipLaunchKernelGGL((agg_hub_fold_kernel<V_, C_>), dim3(unsigned(num_chunks)), dim3(256), lds, st, f, g.gs_log2);    \
    f.level = 2;                                                                                                        \
    hipLaunchKernelGGL((agg_hub_fold_kernel<V_, C_>), dim3(unsigned(num_chunks)), dim3(256), lds, st, f, g.gs_log2);    \
  } while (0)
  if (g.vec == 4) {
    switch (g.cpl) {
      case 1: MGCN_FOLD_CASE(4, 1); break;
      case 2: MGCN_FOLD_CASE(4, 2); break;
      case 4: MGCN_FOLD_CASE(4, 4); break;
      default: MGCN_FOLD_CASE(4, 8); break;
    }
  } else {
    switch (g.cpl) {
      case 1: MGCN_FOLD_CASE(1, 1); break;
      case 2: MGCN_FOLD_CASE(1, 2); break;
      case 4: MGCN_FOLD_CASE(1, 4); break;
      default: MGCN_FOLD_CASE(1, 8); break;
    }
  }
#undef MGCN_FOLD_CASE
}

}  // namespace

int mgcn::launch_hub_partials(int64_t num_nodes, int32_t dim, int32_t num_rel_rows, const mgcn_edge_rec *rec_dev,
                              const float *x_dev, int64_t ldx, const float *rel_dev, const float *loop_rel_dev,
                              const float *ee_dev, int32_t ee_in_slot_order, int64_t ee_sub_hub,
                              const int32_t *chunks_dev, int64_t chunk_begin, int64_t chunk_end, float *partial_dev,
                              void *stream) {
  const int64_t num_chunks = chunk_end - chunk_begin;
  const bool aligned = mgcn::aligned16(x_dev) && mgcn::aligned16(rel_dev) && mgcn::aligned16(loop_rel_dev) &&
                       mgcn::aligned16(partial_dev) && (!ee_dev || mgcn::aligned16(ee_dev)) && ldx % 4 == 0;
  Geometry g;
  if (!pick_geometry(dim, aligned, &g)) return mgcn::fail(MGCN_EUNSUPPORTED, "hub partials: dim %d too wide", dim);
  AggArgs p = {};
  p.rec = reinterpret_cast<const int4 *>(rec_dev);
  p.x = x_dev; p.rel = rel_dev; p.loop_rel = loop_rel_dev; p.ee = ee_dev;
  p.ldx = ldx; p.n = int32_t(num_nodes); p.d = dim; p.rel_rows = num_rel_rows; p.ee_slot_order = ee_in_slot_order;
  p.chunks = reinterpret_cast<const int4 *>(chunks_dev);
  p.partial = partial_dev;
  p.hubcnt = reinterpret_cast<int32_t *>(partial_dev + num_chunks * dim);
  p.chunk0 = int32_t(chunk_begin);
  p.nchunks = int32_t(num_chunks);
  p.ee_sub_hub = ee_sub_hub;
  const int64_t threads = num_chunks << g.gs_log2;
  const unsigned grid = unsigned((threads + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define MGCN_HUB_CASE(V_, C_, U_) hipLaunchKernelGGL((agg_hub_kernel<V_, C_, U_>), dim3(grid), dim3(256), 0, st, p, g.gs_log2)
  if (g.vec == 4) {
    switch (g.cpl) {
      case 1: MGCN_HUB_CASE(4, 1, 4); break;
      case 2: MGCN_HUB_CASE(4, 2, 2); break;
      case 4: MGCN_HUB_CASE(4, 4, 1); break;
      default: MGCN_HUB_CASE(4, 8, 1); break;
    }
  } else {
    switch (g.cpl) {
      case 1: MGCN_HUB_CASE(1, 1, 4); break;
      case 2: MGCN_HUB_CASE(1, 2, 2); break;
      case 4: MGCN_HUB_CASE(1, 4, 1); break;
      default: MGCN_HUB_CASE(1, 8, 1); break;
    }
  }
#undef MGCN_HUB_CASE
  MGCN_CHECK_LAUNCH("agg_hub_kernel");
  return MGCN_OK;
}

extern "C" int64_t mgcn_hub_partial_floats(int64_t num_chunks, int32_t dim) {
  return num_chunks > 0 && dim > 0 ? num_chunks * dim + 2 * num_chunks : 0;
}

extern "C" int mgcn_aggregate_fwd(int64_t num_nodes, int64_t num_edges_half, int32_t dim, int32_t num_rel_rows,
                                  const int32_t *rowptr_dev, const mgcn_edge_rec *rec_dev, const float *x_dev,
                                  int64_t ldx, const float *rel_dev, const float *loop_rel_dev, const float *ee_dev,
                                  int32_t ee_in_slot_order, const float *loop_edge_dev, float *a_dev, int64_t lda,
                                  int64_t node_begin, int64_t node_end, const int32_t *hubinfo_dev,
                                  const int32_t *chunks_dev, int64_t chunk_begin, int64_t chunk_end, float *partial_dev,
                                  int64_t ee_sub_in, int64_t ee_sub_out, int64_t ee_sub_hub, void *stream) {
  MGCN_REQUIRE(num_nodes >= 0 && num_edges_half >= 0 && dim > 0 && num_rel_rows > 0, "aggregate_fwd: bad sizes");
  MGCN_REQUIRE((ee_sub_in == 0 && ee_sub_out == 0 && ee_sub_hub == 0) || (ee_dev && ee_in_slot_order),
               "aggregate_fwd: table shard offsets need a per-edge table in slot order");
  MGCN_REQUIRE(num_nodes < (int64_t(1) << 31) - 1 && 2 * num_edges_half < (int64_t(1) << 31) - 1,
               "aggregate_fwd: sizes exceed int32");
  MGCN_REQUIRE(rowptr_dev && x_dev && a_dev && loop_rel_dev && (rel_dev || num_rel_rows == 1),
               "aggregate_fwd: null pointer");
  MGCN_REQUIRE(num_edges_half == 0 || rec_dev, "aggregate_fwd: null rec");
  const int modes = loop_edge_dev ? 3 : 2;
  MGCN_REQUIRE(ldx >= dim && lda >= int64_t(modes) * dim, "aggregate_fwd: ldx/lda too small");
  MGCN_REQUIRE(node_begin >= 0 && node_begin <= node_end && node_end <= num_nodes, "aggregate_fwd: bad node range");
  const int64_t num_chunks = chunk_end - chunk_begin;
  MGCN_REQUIRE(chunk_begin >= 0 && num_chunks >= 0 && chunk_end < (int64_t(1) << 31) &&
                   (num_chunks == 0 || (hubinfo_dev && chunks_dev && partial_dev && mgcn::aligned16(partial_dev))),
               "aggregate_fwd: hub chunks need hubinfo / chunks / a 16-byte aligned partial buffer");
  if (node_end == node_begin) return MGCN_OK;
  const bool aligned = mgcn::aligned16(x_dev) && mgcn::aligned16(rel_dev) && mgcn::aligned16(loop_rel_dev) &&
                       mgcn::aligned16(a_dev) && (!ee_dev || mgcn::aligned16(ee_dev)) &&
                       (!loop_edge_dev || mgcn::aligned16(loop_edge_dev)) && ldx % 4 == 0 && lda % 4 == 0;
  Geometry g;
  if (!pick_geometry(dim, aligned, &g)) return mgcn::fail(MGCN_EUNSUPPORTED, "aggregate_fwd: dim %d too wide", dim);
  AggArgs p = {};
  p.rowptr = rowptr_dev;
  p.rec = reinterpret_cast<const int4 *>(rec_dev);
  p.x = x_dev;
  p.rel = rel_dev;
  p.loop_rel = loop_rel_dev;
  p.ee = ee_dev;
  p.loop_edge = loop_edge_dev;
  p.a = a_dev;
  p.ldx = ldx;
  p.lda = lda;
  p.n = int32_t(num_nodes);
  p.e = int32_t(num_edges_half);
  p.d = dim;
  p.rel_rows = num_rel_rows;
  p.ee_slot_order = ee_in_slot_order;
  p.modes = modes;
  p.node0 = int32_t(node_begin);
  p.nodes = int32_t(node_end - node_begin);
  p.hubinfo = num_chunks > 0 ? reinterpret_cast<const int2 *>(hubinfo_dev) : nullptr;
  p.chunks = reinterpret_cast<const int4 *>(chunks_dev);
  p.partial = partial_dev;
  p.chunk0 = int32_t(chunk_begin);
  p.nchunks = int32_t(num_chunks);
  p.ee_sub_hub = ee_sub_hub;
  p.ee_sub[0] = ee_sub_in; p.ee_sub[1] = ee_sub_out;
  if (num_chunks > 0) {
    if (int rc = mgcn::launch_hub_partials(num_nodes, dim, num_rel_rows, rec_dev, x_dev, ldx, rel_dev, loop_rel_dev, ee_dev,
                                           ee_in_slot_order, ee_sub_hub, chunks_dev, chunk_begin, chunk_end, partial_dev, stream))
      return rc;
  }
  {
    const int64_t threads = (int64_t(modes) * p.nodes) << g.gs_log2;
    const unsigned grid = unsigned((threads + 255) / 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define MGCN_FWD_CASE(V_, C_, U_) hipLaunchKernelGGL((agg_fwd_kernel<V_, C_, U_>), dim3(grid), dim3(256), 0, st, p, g.gs_log2)
    if (g.vec == 4) {
      switch (g.cpl) {
        case 1:
          // slots per batch: short runs (WN18RR: 2.1 slots per destination and half) are covered by two and the smaller
          // build runs more waves per SIMD (44 / 72 us per WN18RR layer against 49 / 84 with four); long runs (FB15k-237:
          // 18.7) take four (96 / 157 us with the hub pre-pass against 98 / 159). A rolling second batch (walk_slots<ROLL>)
          // pays only in the hub pre-pass: on short runs it issues two batches past the run's end (56 / 96 us).
          if (num_edges_half < 4 * num_nodes) { MGCN_FWD_CASE(4, 1, 2); } else { MGCN_FWD_CASE(4, 1, 4); }
          break;
        case 2: MGCN_FWD_CASE(4, 2, 2); break;
        case 4: MGCN_FWD_CASE(4, 4, 1); break;
        default: MGCN_FWD_CASE(4, 8, 1); break;
      }
    } else {
      switch (g.cpl) {
        case 1: MGCN_FWD_CASE(1, 1, 4); break;
        case 2: MGCN_FWD_CASE(1, 2, 2); break;
        case 4: MGCN_FWD_CASE(1, 4, 1); break;
        default: MGCN_FWD_CASE(1, 8, 1); break;
      }
    }
#undef MGCN_FWD_CASE
  }
  MGCN_CHECK_LAUNCH("agg_fwd_kernel");
  return MGCN_OK;
}

extern "C" size_t mgcn_aggregate_bwd_workspace(int64_t num_edges_half, int32_t dim, int32_t num_rel_rows,
                                               int64_t num_hub_chunks) {
  const int64_t nchunks = (2 * num_edges_half + kTypeChunk - 1) / kTypeChunk;
  return size_t(nchunks + num_rel_rows + (num_hub_chunks > 0 ? num_hub_chunks : 0)) * size_t(dim) * sizeof(float);
}

extern "C" int mgcn_aggregate_bwd(int64_t num_nodes, int64_t num_edges_half, int32_t dim, int32_t num_rel_rows,
                                  const int32_t *rowptr_dev, const mgcn_edge_rec *rec_dev,
                                  const int32_t *slot_dst_dev, const int32_t *mirror_dev,
                                  const int32_t *hubinfo_dev, const int32_t *chunks_dev, int64_t num_hub_chunks,
                                  const int32_t *typeptr_dev, const int32_t *typeslots_dev, const float *x_dev,
                                  int64_t ldx, const float *rel_dev, const float *ee_dev, const float *g_dev,
                                  int64_t ldg, float *gx_dev, float *gee_dev, float *grel_dev,
                                  float *workspace_dev, size_t workspace_bytes, void *stream) {
  MGCN_REQUIRE(num_nodes >= 0 && num_edges_half >= 0 && dim > 0 && num_rel_rows > 0, "aggregate_bwd: bad sizes");
  MGCN_REQUIRE(num_nodes < (int64_t(1) << 31) - 1 && 2 * num_edges_half < (int64_t(1) << 31) - 1,
               "aggregate_bwd: sizes exceed int32");
  MGCN_REQUIRE(x_dev && rel_dev && g_dev, "aggregate_bwd: null pointer");
  MGCN_REQUIRE(num_edges_half == 0 || (rec_dev && slot_dst_dev), "aggregate_bwd: null slot arrays");
  MGCN_REQUIRE(ldx >= dim && ldg >= 2 * int64_t(dim), "aggregate_bwd: ldx/ldg too small");
  MGCN_REQUIRE(!gx_dev || (rowptr_dev && (num_edges_half == 0 || mirror_dev)), "aggregate_bwd: gx needs rowptr/mirror");
  MGCN_REQUIRE(num_hub_chunks >= 0 && num_hub_chunks < (int64_t(1) << 31) &&
                   (num_hub_chunks == 0 || !gx_dev || (hubinfo_dev && chunks_dev)),
               "aggregate_bwd: hub chunks need hubinfo / chunks");
  const size_t ws_need = mgcn_aggregate_bwd_workspace(num_edges_half, dim, num_rel_rows, num_hub_chunks);
  MGCN_REQUIRE(!(grel_dev || (gx_dev && num_hub_chunks > 0)) || (workspace_dev && workspace_bytes >= ws_need),
               "aggregate_bwd: workspace too small (%zu bytes needed)", ws_need);
  MGCN_REQUIRE(!grel_dev || (typeptr_dev && (num_edges_half == 0 || typeslots_dev)), "aggregate_bwd: grel needs typeptr/typeslots");
  const bool aligned = mgcn::aligned16(x_dev) && mgcn::aligned16(rel_dev) && mgcn::aligned16(g_dev) &&
                       (!ee_dev || mgcn::aligned16(ee_dev)) && (!gx_dev || mgcn::aligned16(gx_dev)) &&
                       (!gee_dev || mgcn::aligned16(gee_dev)) && (!grel_dev || mgcn::aligned16(grel_dev)) &&
                       (!workspace_dev || mgcn::aligned16(workspace_dev)) && ldx % 4 == 0 && ldg % 4 == 0;
  Geometry g;
  if (!pick_geometry(dim, aligned, &g)) return mgcn::fail(MGCN_EUNSUPPORTED, "aggregate_bwd: dim %d too wide", dim);
  BwdArgs p;
  p.rec = reinterpret_cast<const int4 *>(rec_dev);
  p.slot_dst = slot_dst_dev;
  p.rowptr = rowptr_dev;
  p.mirror = mirror_dev;
  p.hubinfo = num_hub_chunks > 0 ? reinterpret_cast<const int2 *>(hubinfo_dev) : nullptr;
  p.chunks = reinterpret_cast<const int4 *>(chunks_dev);
  p.nchunks_hub = int32_t(num_hub_chunks);
  p.typeptr = typeptr_dev;
  p.typeslots = typeslots_dev;
  p.x = x_dev;
  p.rel = rel_dev;
  p.ee = ee_dev;
  p.g = g_dev;
  p.gx = gx_dev;
  p.gee = gee_dev;
  p.grel = grel_dev;
  p.ws = workspace_dev;
  p.ldx = ldx;
  p.ldg = ldg;
  p.n = int32_t(num_nodes);
  p.e = int32_t(num_edges_half);
  p.d = dim;
  p.rel_rows = num_rel_rows;
  p.nchunks_type = int32_t((2 * num_edges_half + kTypeChunk - 1) / kTypeChunk);
  p.hub_ws = workspace_dev ? workspace_dev + (int64_t(p.nchunks_type) + num_rel_rows) * dim : nullptr;
  // gee and grel together (the training step): one pass over the slots in type order instead of two (LAB_NOTES.md, round 4)
  const bool fused_gee_grel = gee_dev && grel_dev && ee_dev && num_edges_half > 0;
  if (gee_dev && num_edges_half > 0 && !fused_gee_grel) {
    MGCN_LAUNCH_GEOM(agg_bwd_gee_kernel, p, 2 * num_edges_half, g, stream);
    MGCN_CHECK_LAUNCH("agg_bwd_gee_kernel");
  }
  if (gx_dev && num_nodes > 0) {
    if (num_hub_chunks > 0) {
      MGCN_LAUNCH_GEOM(agg_bwd_gx_hub_kernel, p, num_hub_chunks, g, stream);
      MGCN_CHECK_LAUNCH("agg_bwd_gx_hub_kernel");
      launch_fold(g, p.chunks, p.hub_ws, 0, dim, num_hub_chunks, static_cast<hipStream_t>(stream));
      MGCN_CHECK_LAUNCH("agg_hub_fold_kernel");
    }
    if (num_edges_half < 4 * num_nodes) {   // 42.5 vs 55 us on the WN18RR shape
      MGCN_LAUNCH_GEOM(agg_bwd_gx_short_kernel, p, num_nodes, g, stream);
    } else {
      MGCN_LAUNCH_GEOM(agg_bwd_gx_kernel, p, num_nodes, g, stream);
    }
    MGCN_CHECK_LAUNCH("agg_bwd_gx_kernel");
  }
  if (grel_dev) {
    if (p.nchunks_type > 0) {
      if (fused_gee_grel) {
        MGCN_LAUNCH_GEOM(agg_bwd_gee_grel_kernel, p, p.nchunks_type, g, stream);
      } else {
        MGCN_LAUNCH_GEOM(agg_bwd_grel_partial_kernel, p, p.nchunks_type, g, stream);
      }
      MGCN_CHECK_LAUNCH("agg_bwd_grel_partial_kernel");
    }
    {
      const size_t lds = size_t(256 >> g.gs_log2) * size_t(dim) * sizeof(float);
      hipStream_t st = static_cast<hipStream_t>(stream);
#define MGCN_GRELF_CASE(V_, C_) hipLaunchKernelGGL((agg_bwd_grel_final_kernel<V_, C_>), dim3(unsigned(num_rel_rows)), dim3(256), lds, st, p, g.gs_log2)
      if (g.vec == 4) {
        switch (g.cpl) {
          case 1: MGCN_GRELF_CASE(4, 1); break;
          case 2: MGCN_GRELF_CASE(4, 2); break;
          case 4: MGCN_GRELF_CASE(4, 4); break;
          default: MGCN_GRELF_CASE(4, 8); break;
        }
      } else {
        switch (g.cpl) {
          case 1: MGCN_GRELF_CASE(1, 1); break;
          case 2: MGCN_GRELF_CASE(1, 2); break;
          case 4: MGCN_GRELF_CASE(1, 4); break;
          default: MGCN_GRELF_CASE(1, 8); break;
        }
      }
#undef MGCN_GRELF_CASE
    }
    MGCN_CHECK_LAUNCH("agg_bwd_grel_final_kernel");
  }
  return MGCN_OK;
}
