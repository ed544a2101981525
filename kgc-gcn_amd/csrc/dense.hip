// f32 MFMA tile kernel (v_mfma_f32_16x16x4_f32: exact f32, a k-ordered fma chain) with the fused
// epilogues of the M-GCN hot path (gfx950):
//   EPI_NONE      C = A B
//   EPI_BN_TANH   out = tanh(BN_eval((A [W_in;W_out;W_loop]) / 3 + bias))   (model.py:103-106,116)
//   EPI_SIGMOID   score[b, n] = sigmoid(ent[n,:] . x[b,:] + bias[n])        (model.py:177-179)
//   EPI_TARGET    target[b]   = score[b, obj[b]]  (same tile arithmetic, gathered rows, diagonal)
//   EPI_RANK      filtered counts gt / ties_lower / ties per query, scores never stored (main.py:122-126)
//   EPI_BCE       training: BCE(sigmoid(score), target) partial sums + d loss / d logit [M, ncols] (main.py:61-66, N3)
// and, for aligned scoring shapes with K <= 352, score_split_kernel<SIGMOID / TARGET / RANK>: the same three results on
// the bf16 MFMA from exactly split operands (below).
//
// Geometry. The streamed operand (aggregates [N,3D], or the entity table [N,O]) is always the MFMA A
// operand and goes global -> registers directly: lane (r = l&15, q = l>>4) loads the 16 bytes
// A[row r][16t+4q .. 16t+4q+3] of k-block t, so element i of that float4 is the A fragment of MFMA step
// i (k = 16t + 4q + i). The small operand B (the weights, or the query block read transposed) is staged
// through LDS in double-buffered slabs of 16 k-rows shared by the block's two waves; a lane's B fragment
// of step i is Bs[4q+i][16*tile + (l&15)] (row stride == 4 mod 8 floats -> conflict-free ds_read_b32).
// Block = 4 waves = 32 rows x NT column tiles: wave w owns row tile (w & 1) and column tiles
// [ (w>>1)*ceil(NT/2), ... ), i.e. 7 + 6 of the 13 tiles of O = 200, so a wave needs only 28 accumulator
// registers and five blocks (20 waves, 5 per SIMD) are resident per CU; the WN18RR grid is 1280 blocks =
// exactly 5 per CU, 2.5 wave-tiles of MFMA work per SIMD with no tail quantisation.
// Summation order per output element is fixed (k-blocks ascending, step i, then q) and independent of
// the tile position, so a score computed by TARGET, SIGMOID and RANK is the same f32 value.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mgcn_common.h"

namespace {

enum { EPI_NONE = 0, EPI_BN_TANH = 1, EPI_SIGMOID = 2, EPI_TARGET = 3, EPI_RANK = 4, EPI_BCE = 5 };

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TileArgs {
  const float *a;        // [M, K], row stride lda (EPI_TARGET: row i is a[(obj[i]-row0)*lda])
  const float *b;        // NN: [K, ncols] row stride ldb; NT: x [ncols, K], row stride ldb
  float *c;              // output (NONE / BN_TANH: [M, ldc]; SIGMOID: [ncols, ldc] transposed store)
  const float *bias;     // BN_TANH: [ncols] or null; scoring: [M] per entity
  const float *bn_mean, *bn_var, *bn_gamma, *bn_beta;
  const int64_t *obj;    // scoring: [ncols] target entity (global id) per query
  const float *target;   // RANK: [ncols]
  float *target_out;     // TARGET: [ncols]
  const float *label;    // RANK: [ncols, ldl] dense 0/1 rows, or
  const uint32_t *mask;  // RANK: [ncols, ldl] words, bit (n & 31) of word n >> 5 set = entity n filtered
  unsigned long long *counts;  // RANK: [ncols, 3]
  float *loss_partial;   // BCE: one partial sum per block
  float hot, cold, inv_count;  // BCE: target values at / off the known tails, 1 / (B * N)
  int64_t lda, ldb, ldc, ldl, m, row0, n_local;
  int32_t k, ncols, tiles_m;
  int32_t a_vec, b_vec;  // 16-byte loads allowed (alignment + leading dimension checked on the host)
  float bn_eps;
};

constexpr int WAVES = 4, BM = 32, KS = 16, THREADS = 64 * WAVES;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// tanh through one exp and one division: |err| <= ~1e-7 absolute (the layer output feeds a 1e-5-level parity bar);
// libm's tanhf costs ~60 instructions per element and dominated the epilogue.
__device__ __forceinline__ float tanhf_(float v) {
  const float t = __expf(-2.0f * fabsf(v));
  return copysignf((1.0f - t) / (1.0f + t), v);
}

// FAST = every operand 16-byte aligned with leading dimensions, K and the column count multiples of 4:
// one predicated dwordx4 per slot, no scalar tails (the branch-free hot instantiation). !FAST = element-wise
// guarded loads for arbitrary shapes; same arithmetic, same summation order.
template <bool FAST>
__device__ __forceinline__ float4 load4(const float *ptr, bool ok, int first, int limit) {
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FAST) {
    if (ok && first < limit) t = *reinterpret_cast<const float4 *>(ptr);
  } else if (ok) {
    if (first + 0 < limit) t.x = ptr[0];
    if (first + 1 < limit) t.y = ptr[1];
    if (first + 2 < limit) t.z = ptr[2];
    if (first + 3 < limit) t.w = ptr[3];
  }
  return t;
}

template <bool B_NT, int NT, bool FAST>
struct BStage;

// Generic shapes: element-wise guarded loads with zero fill.
template <bool B_NT, int NT>
struct BStage<B_NT, NT, false> {
  static constexpr int BNC = NT * 16;
  static constexpr int LDB = BNC + 4;
  // NN: slots of 4 consecutive columns of one k-row; NT: slots of 4 consecutive k of one column
  static constexpr int SLOTS = B_NT ? BNC * 4 : KS * (BNC / 4);
  static constexpr int PER_THREAD = (SLOTS + THREADS - 1) / THREADS;
  float4 v[PER_THREAD];

  __device__ __forceinline__ void init(const TileArgs &, int, int) {}

  __device__ __forceinline__ void load(const TileArgs &p, int kb, int c0, int tid) {
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      const int s = tid + j * THREADS;
      if (B_NT) {
        const int c = s >> 2, kk = kb * KS + (s & 3) * 4;
        v[j] = load4<false>(p.b + int64_t(c0 + c) * p.ldb + kk, s < SLOTS && c0 + c < p.ncols, kk, p.k);
      } else {
        const int r = s / (BNC / 4), cc = c0 + (s % (BNC / 4)) * 4;
        const int kk = kb * KS + r;
        v[j] = load4<false>(p.b + int64_t(kk) * p.ldb + cc, s < SLOTS && kk < p.k, cc, p.ncols);
      }
    }
  }

  __device__ __forceinline__ void store(float *bs, int tid) const {
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      const int s = tid + j * THREADS;
      if (s < SLOTS) {
        if (B_NT) {
          const int c = s >> 2, r = (s & 3) * 4;
          bs[(r + 0) * LDB + c] = v[j].x;
          bs[(r + 1) * LDB + c] = v[j].y;
          bs[(r + 2) * LDB + c] = v[j].z;
          bs[(r + 3) * LDB + c] = v[j].w;
        } else {
          const int r = s / (BNC / 4), c = (s % (BNC / 4)) * 4;
          *reinterpret_cast<float4 *>(bs + r * LDB + c) = v[j];
        }
      }
    }
  }
};

// Aligned shapes (16-byte aligned operands, K and the column count multiples of 4): every slot is ONE
// unconditional dwordx4 whose address is loop-invariant except for the k-block advance. Nothing is
// zero-filled: a k-row past K is clamped to the last valid row and meets an A fragment that IS zero-filled,
// and a column past ncols is clamped to a valid column whose results the epilogue never stores. Slots past
// the slab (last j of some threads) repeat the last slot (same bytes to the same LDS address).
template <bool B_NT, int NT>
struct BStage<B_NT, NT, true> {
  static constexpr int BNC = NT * 16;
  static constexpr int LDB = BNC + 4;
  static constexpr int SLOTS = B_NT ? BNC * 4 : KS * (BNC / 4);
  static constexpr int PER_THREAD = (SLOTS + THREADS - 1) / THREADS;
  float4 v[PER_THREAD];
  int goff[PER_THREAD];   // NN: column (elements); NT: row offset c*ldb (elements)
  int krow[PER_THREAD];   // NN: k-row inside the slab; NT: first k of the quad inside the slab
  int loff[PER_THREAD];   // LDS element offset of the slot

  __device__ __forceinline__ void init(const TileArgs &p, int c0, int tid) {
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      int s = tid + j * THREADS;
      s = s < SLOTS ? s : SLOTS - 1;
      if (B_NT) {
        const int c = s >> 2;
        int cg = c0 + c;
        cg = cg < p.ncols ? cg : p.ncols - 1;
        goff[j] = cg * int(p.ldb);
        krow[j] = (s & 3) * 4;
        loff[j] = krow[j] * LDB + c;
      } else {
        const int r = s / (BNC / 4), cq = (s % (BNC / 4)) * 4;
        int cc = c0 + cq;
        cc = cc < p.ncols ? cc : p.ncols - 4;
        goff[j] = cc;
        krow[j] = r;
        loff[j] = r * LDB + cq;
      }
    }
  }

  __device__ __forceinline__ void load(const TileArgs &p, int kb, int, int) {
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      int kk = kb * KS + krow[j];
      if (B_NT) {
        kk = kk < p.k ? kk : p.k - 4;
        v[j] = *reinterpret_cast<const float4 *>(p.b + goff[j] + kk);
      } else {
        kk = kk < p.k ? kk : p.k - 1;
        v[j] = *reinterpret_cast<const float4 *>(p.b + int64_t(kk) * p.ldb + goff[j]);
      }
    }
  }

  __device__ __forceinline__ void store(float *bs, int) const {
#pragma unroll
    for (int j = 0; j < PER_THREAD; ++j) {
      if (B_NT) {
        bs[loff[j] + 0 * LDB] = v[j].x;
        bs[loff[j] + 1 * LDB] = v[j].y;
        bs[loff[j] + 2 * LDB] = v[j].z;
        bs[loff[j] + 3 * LDB] = v[j].w;
      } else {
        *reinterpret_cast<float4 *>(bs + loff[j]) = v[j];
      }
    }
  }
};

// NN + aligned shapes: the weight slab goes global -> LDS by LDS-DMA (global_load_lds_dwordx4), 1 KiB per
// wave-instruction, no staging registers and no ds_write pass. The LDS image is the same [KS][LDB] slab;
// a lane's 16 bytes land at piece*1024 + lane*16, so its SOURCE address is computed from that offset
// (row = offset / row bytes, column = remainder); bytes that fall in a row's 16-byte pad or past the slab
// read some valid address and are never used. Clamping rules as in BStage<.., true>.
template <int NT, int PAD = 4>
struct BDma {
  static constexpr int BNC = NT * 16;
  static constexpr int LDB = BNC + PAD;
  static constexpr int ROW_BYTES = LDB * 4;
  static constexpr int PIECES = (KS * ROW_BYTES + 1023) / 1024;
  static constexpr int SLAB_F = PIECES * 256;            // floats per buffer (whole pieces)
  static constexpr int PPW = (PIECES + WAVES - 1) / WAVES;  // pieces per wave
  int rr[PPW], cc[PPW];

  __device__ __forceinline__ void init(int ncols, int c0, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int o = (wave + j * WAVES) * 1024 + lane * 16;
      int r = o / ROW_BYTES, c = (o - r * ROW_BYTES) >> 2;
      r = r < KS ? r : KS - 1;
      c = c < BNC ? c : BNC - 4;
      int cg = c0 + c;
      cg = cg < ncols ? cg : ncols - 4;
      rr[j] = r;
      cc[j] = cg;
    }
  }

  // slab kb of the [k, ncols] matrix `b` (row stride ldb) -> LDS slab
  __device__ __forceinline__ void issue(const float *b, int64_t ldb, int k, int kb, float *slab, int wave) const {
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = wave + j * WAVES;  // wave-uniform
      if (piece < PIECES) {
        int kk = kb * KS + rr[j];
        kk = kk < k ? kk : k - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b + int64_t(kk) * ldb + cc[j]),
                                         (__attribute__((address_space(3))) void *)(slab + piece * 256), 16, 0, 0);
      }
    }
  }
};

// Register budget per instantiation (waves per SIMD the compiler must leave room for). At five waves per SIMD (96
// VGPRs) the wide instantiations spilled: 6-62 registers for the 13-tile dense / scoring kernels, 12-195 for RANK,
// which carries three counters per column tile next to the accumulators (the one-shot evaluation ran 16 % slower
// for it). Every instantiation now compiles spill-free.
constexpr int min_waves(int epi, int nt) {
  return epi == EPI_RANK ? (nt >= 13 ? 2 : nt >= 8 ? 3 : 5) : (nt >= 13 ? 3 : nt >= 8 ? 4 : 5);
}

template <int EPI, bool B_NT, int NT, bool FAST>
__global__ __launch_bounds__(THREADS, min_waves(EPI, NT)) void tile_kernel(TileArgs p) {
  using Stage = BStage<B_NT, NT, FAST>;
  using Dma = BDma<NT>;
  constexpr bool DMA = FAST && !B_NT;
  constexpr int BNC = Stage::BNC, LDB = Stage::LDB;
  constexpr int NTW = (NT + 1) / 2;  // column tiles of a wave in the first half; the second half has NT - NTW
  constexpr int SLAB_F = DMA ? Dma::SLAB_F : KS * LDB;
  // ONE shared array (slabs, output staging, RANK counters): a second __shared__ object next to an LDS-DMA
  // target makes hipcc drain vmcnt before every ds_read (cdna_hip_programming.md, M = 256 GEMM item 4a).
  __shared__ __attribute__((aligned(16))) float Bs[2 * SLAB_F + (EPI == EPI_RANK ? BNC * 3 : (EPI == EPI_BCE ? 4 : 0))];
  static_assert(2 * KS == BM, "the output staging tile [BM][LDB] reuses the two slab buffers");
  unsigned int *cnt = reinterpret_cast<unsigned int *>(Bs + 2 * SLAB_F);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int rt = wave & 1, ch = wave >> 1;
  const int ct0 = ch * NTW;                      // first column tile of this wave
  const int nct = ch == 0 ? NTW : NT - NTW;      // wave-uniform
  const int fr = lane & 15, fq = lane >> 4;
  const int c0 = int(blockIdx.y) * BNC;
  const int nkb = (p.k + KS - 1) / KS;

  if (EPI == EPI_RANK) {
    for (int i = tid; i < BNC * 3; i += THREADS) cnt[i] = 0;
  }
  float bce_sum = 0.f;
  unsigned int my_gt[NTW], my_tl[NTW], my_ti[NTW];
  if (EPI == EPI_RANK) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) my_gt[t] = my_tl[t] = my_ti[t] = 0;
  }

  // RANK walks several row tiles per block (grid-stride) so that its counts stay on chip
  for (int tm = blockIdx.x; tm < p.tiles_m; tm += gridDim.x) {
    const int64_t r0 = int64_t(tm) * BM;
    f32x4 acc[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    int64_t arow = r0 + rt * 16 + fr;
    bool a_ok = arow < p.m;
    if (FAST && EPI != EPI_TARGET && !a_ok) {  // rows past M are never stored: read a valid row instead of predicating
      arow = p.m - 1;
      a_ok = true;
    }
    if (EPI == EPI_TARGET && a_ok) {
      const int64_t o = p.obj[arow] - p.row0;  // the entity row query `arow` must be scored against
      a_ok = o >= 0 && o < p.n_local;
      arow = o;
    }
    const float *aptr = p.a + (a_ok ? arow : 0) * p.lda;

    Stage st;
    Dma dma;
    __syncthreads();  // previous tile's readers are done with buffer 0
    if (DMA) {
      dma.init(p.ncols, c0, wave, lane);
      dma.issue(p.b, p.ldb, p.k, 0, Bs, wave);
    } else {
      st.init(p, c0, tid);
      st.load(p, 0, c0, tid);
    }
    float4 a_cur = load4<FAST>(aptr + 4 * fq, a_ok, 4 * fq, p.k);
    if (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      st.store(Bs, tid);
    }
    __syncthreads();
    int cur = 0;
    for (int kb = 0; kb < nkb; ++kb) {
      float4 a_next = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool more = kb + 1 < nkb;
      if (more) {
        if (DMA) dma.issue(p.b, p.ldb, p.k, kb + 1, Bs + (cur ^ 1) * SLAB_F, wave); else st.load(p, kb + 1, c0, tid);
        a_next = load4<FAST>(aptr + (kb + 1) * KS + 4 * fq, a_ok, (kb + 1) * KS + 4 * fq, p.k);
      }
      const float *bs = Bs + cur * SLAB_F + (4 * fq) * LDB + ct0 * 16 + fr;
      const float av[4] = {a_cur.x, a_cur.y, a_cur.z, a_cur.w};
      // B fragments one MFMA step ahead: the ds_reads of step i+1 are in flight under step i's MFMAs
      float bf[2][NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) bf[0][t] = bs[t * 16];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i + 1 < 4) {
#pragma unroll
          for (int t = 0; t < NTW; ++t) bf[(i + 1) & 1][t] = bs[(i + 1) * LDB + t * 16];
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t)
          if (t < nct) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bf[i & 1][t], acc[t], 0, 0, 0);
      }
      if (DMA) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slab for kb+1 has landed (read it after the barrier)
      } else if (more) {
        st.store(Bs + (cur ^ 1) * SLAB_F, tid);
      }
      a_cur = a_next;
      __syncthreads();
      cur ^= 1;
    }

    // ---- epilogue: lane holds rows r0 + rt*16 + fq*4 + j (j = 0..3), column c0 + (ct0+t)*16 + fr ---
    // Row-major outputs (NONE, BN_TANH) are staged through the (now idle) slab buffers so that every lane
    // stores 16 contiguous bytes of one output row: a dword-per-lane store of the accumulator layout touches
    // 64-byte pieces of four rows per instruction and made the store tail the longest phase of the kernel.
    constexpr bool STAGED = FAST && (EPI == EPI_NONE || EPI == EPI_BN_TANH || EPI == EPI_BCE);
    float *os = Bs;  // [BM][LDB]; 2*KS*LDB == BM*LDB floats
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int lcol = (ct0 + t) * 16 + fr;
      const int col = c0 + lcol;
      if (t >= nct || col >= p.ncols) continue;
      float cb = 0.f, mean = 0.f, inv = 1.f, gam = 1.f, bet = 0.f, tgt = 0.f;
      int64_t ob = -1;
      if (EPI == EPI_BN_TANH) {
        cb = p.bias ? p.bias[col] : 0.f;
        mean = p.bn_mean[col];
        inv = 1.0f / sqrtf(p.bn_var[col] + p.bn_eps);
        gam = p.bn_gamma[col];
        bet = p.bn_beta[col];
      }
      if (EPI == EPI_RANK) {
        tgt = p.target[col];
        ob = p.obj[col] - p.row0;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lrow = rt * 16 + fq * 4 + j;
        const int64_t row = r0 + lrow;
        if (row >= p.m) continue;
        const float v = acc[t][j];
        if (EPI == EPI_NONE) {
          if (STAGED) os[lrow * LDB + lcol] = v; else p.c[row * p.ldc + col] = v;
        } else if (EPI == EPI_BN_TANH) {
          float o = v / 3.0f;
          if (p.bias) o = o + cb;
          o = tanhf_((o - mean) * inv * gam + bet);
          if (STAGED) os[lrow * LDB + lcol] = o; else p.c[row * p.ldc + col] = o;
        } else if (EPI == EPI_SIGMOID) {
          p.c[int64_t(col) * p.ldc + row] = sigmoidf_(v + p.bias[row]);
        } else if (EPI == EPI_TARGET) {
          if (row == col) {
            const int64_t o = p.obj[row] - p.row0;
            if (o >= 0 && o < p.n_local) p.target_out[row] = sigmoidf_(v + p.bias[o]);
          }
        } else if (EPI == EPI_BCE) {
          // torch's BCELoss on sigmoid outputs (main.py:62): loss = -(y log p + (1-y) log(1-p)), logs clamped at -100;
          // d loss / d logit = (p - y) / max(p (1-p), 1e-12) * p (1-p) / (B N), the product of its BCE and sigmoid
          // backward formulas. y = hot at the known tails of the query (bit mask), cold elsewhere (label smoothing).
          const float pz = sigmoidf_(v + p.bias[row]);
          const bool pos = (p.mask[int64_t(col) * p.ldl + (row >> 5)] >> (row & 31)) & 1u;
          const float y = pos ? p.hot : p.cold;
          bce_sum += (y - 1.0f) * fmaxf(log1pf(-pz), -100.0f) - y * fmaxf(logf(pz), -100.0f);
          const float pq = (1.0f - pz) * pz;
          const float gz = (pz - y) / fmaxf(pq, 1e-12f) * pq * p.inv_count;
          if (STAGED) os[lrow * LDB + lcol] = gz; else p.c[row * p.ldc + col] = gz;
        } else if (EPI == EPI_RANK) {
          if (row == ob) continue;                                       // the target itself (main.py:125)
          if (p.mask) {                                                    // bit-packed filter (uniform branch)
            if ((p.mask[int64_t(col) * p.ldl + (row >> 5)] >> (row & 31)) & 1u) continue;
          } else {
            const float lab = p.label[int64_t(col) * p.ldl + row];
            if ((static_cast<int>(lab) & 0xff) != 0) continue;            // label.byte() filter (main.py:124)
          }
          const float s = sigmoidf_(v + p.bias[row]);
          my_gt[t] += s > tgt;
          const bool eq = s == tgt;
          my_ti[t] += eq;
          my_tl[t] += eq && (row < ob);
        }
      }
    }
    if (STAGED) {
      __syncthreads();
      const int c4n = (((p.ncols - c0) < BNC ? (p.ncols - c0) : BNC) + 3) / 4;  // float4 pieces per row (ncols % 4 == 0)
      const bool vec_out = (p.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(p.c) & 15u) == 0);
      for (int s4 = tid; s4 < BM * c4n; s4 += THREADS) {
        const int lrow = s4 / c4n, lc = (s4 - lrow * c4n) * 4;
        const int64_t row = r0 + lrow;
        if (row >= p.m) continue;
        const float4 v4 = *reinterpret_cast<const float4 *>(os + lrow * LDB + lc);
        float *dst = p.c + row * p.ldc + c0 + lc;
        if (vec_out) {
          *reinterpret_cast<float4 *>(dst) = v4;
        } else {
          dst[0] = v4.x; dst[1] = v4.y; dst[2] = v4.z; dst[3] = v4.w;
        }
      }
    }
  }

  if (EPI == EPI_BCE) {   // fixed-order reduction: lanes (xor tree), then the four waves in wave order
    float v = bce_sum;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    float *wsum = Bs + 2 * SLAB_F;
    __syncthreads();
    if (lane == 0) wsum[wave] = v;
    __syncthreads();
    if (tid == 0) p.loss_partial[blockIdx.y * gridDim.x + blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
  }
  if (EPI == EPI_RANK) {
    // lanes fr, fr+16, fr+32, fr+48 hold the same query column: fold, then one LDS add per wave
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      unsigned int g = my_gt[t], l = my_tl[t], e = my_ti[t];
      g += __shfl_xor(g, 16); l += __shfl_xor(l, 16); e += __shfl_xor(e, 16);
      g += __shfl_xor(g, 32); l += __shfl_xor(l, 32); e += __shfl_xor(e, 32);
      if (fq == 0 && t < nct) {
        atomicAdd(&cnt[((ct0 + t) * 16 + fr) * 3 + 0], g);
        atomicAdd(&cnt[((ct0 + t) * 16 + fr) * 3 + 1], l);
        atomicAdd(&cnt[((ct0 + t) * 16 + fr) * 3 + 2], e);
      }
    }
    __syncthreads();
    for (int i = tid; i < BNC * 3; i += THREADS) {
      const int col = c0 + i / 3;
      if (col < p.ncols && cnt[i]) atomicAdd(&p.counts[int64_t(col) * 3 + i % 3], (unsigned long long)cnt[i]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Scoring on the bf16 MFMA (round 3): score / target / filtered rank counts for ALIGNED shapes with K <= 352 — the
// shapes of every evaluation of the benchmark. One kernel, three epilogues, ONE arithmetic: every entity value a and
// query value x is split exactly into three bf16 pieces (hi = bf16(v) rounded, mid = bf16(v - hi), lo = v - hi - mid;
// the split of layer_fused3.hip) and a score is the f32 sum over k-blocks of 32 of the six products
// (a_hi x_lo) (a_lo x_hi) (a_mid x_mid) (a_hi x_mid) (a_mid x_hi) (a_hi x_hi) on v_mfma_f32_16x16x32_bf16 — 6/16 of the
// exact-f32 MFMA's pipe time, everything down to 2^-26 |a||x| per product kept. The order is fixed and does not depend
// on the tile, the strip or the epilogue, so a score computed by SIGMOID, TARGET and RANK is the same f32 value and
// counts stay exact against a recount over the materialised scores. Unaligned shapes and K > 352 keep the exact-f32
// tile kernels for all three entry points alike.
// Block = 512 threads, a strip of 64 queries resident in LDS as three bf16 pieces laid out [piece][k-block][8-k group]
// [query][16 B] (a lane's B fragment of a column tile is one conflict-free ds_read_b128); each wave walks 16-entity
// row tiles: 32 bytes of the entity row per lane and k-block straight from global (the next k-block's in flight),
// split in registers (18 VALU per 8 values), 4 column tiles x 6 MFMAs.
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3p_(float v0, float v1, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2s{v0, v1}, bf16x2s));
  const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
  m = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2s{r0, r1}, bf16x2s));
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2s{s0, s1}, bf16x2s));
}
__device__ __forceinline__ void split8_(const float4 &lo, const float4 &hi, u32x4s &h, u32x4s &m, u32x4s &l) {
  uint32_t a, b, c;
  split3p_(lo.x, lo.y, a, b, c); h[0] = a; m[0] = b; l[0] = c;
  split3p_(lo.z, lo.w, a, b, c); h[1] = a; m[1] = b; l[1] = c;
  split3p_(hi.x, hi.y, a, b, c); h[2] = a; m[2] = b; l[2] = c;
  split3p_(hi.z, hi.w, a, b, c); h[3] = a; m[3] = b; l[3] = c;
}
constexpr int SS_NQT = 4, SS_BQ = SS_NQT * 16, SS_MAX_KB = 11, SS_THREADS = 512;   // 64 queries per strip, K <= 352

template <int EPI>
__global__ __launch_bounds__(SS_THREADS, 2) void score_split_kernel(TileArgs p) {
  constexpr int NQT = SS_NQT, BQ = SS_BQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char ss[];
  const int nkb = (p.k + 31) >> 5;
  const int piece = nkb * 4 * BQ * 16;
  unsigned int *cnt = reinterpret_cast<unsigned int *>(ss + 3 * piece);   // RANK: [BQ][3]
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int fr = lane & 15, fq = lane >> 4;
  const int c0 = int(blockIdx.y) * BQ;
  // the strip's queries -> LDS, split: item = (8-k group, query); consecutive threads take consecutive queries; four items' loads
  // in flight per thread (indices clamped, stores predicated: no branch around a load)
  {
    const int items = nkb * 4 * BQ;
    constexpr int SU = 4;
    for (int it0 = tid; it0 < items; it0 += SU * SS_THREADS) {
      float4 lo[SU], hi[SU];
#pragma unroll
      for (int j = 0; j < SU; ++j) {
        int it = it0 + j * SS_THREADS;
        it = it < items ? it : items - 1;
        const int q = it % BQ, k0 = 8 * (it / BQ);
        int col = c0 + q;
        col = col < p.ncols ? col : p.ncols - 1;       // queries past the batch repeat the last one, never stored or counted
        const float *src = p.b + int64_t(col) * p.ldb;
        const int ka = k0 < p.k ? k0 : 0, kb2 = k0 + 4 < p.k ? k0 + 4 : 0;   // (K % 4 == 0; columns past K read column 0, zeroed below)
        lo[j] = *reinterpret_cast<const float4 *>(src + ka);
        hi[j] = *reinterpret_cast<const float4 *>(src + kb2);
        if (k0 >= p.k) lo[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k0 + 4 >= p.k) hi[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < SU; ++j) {
        const int it = it0 + j * SS_THREADS;
        if (it < items) {
          u32x4s h, m, l;
          split8_(lo[j], hi[j], h, m, l);
          unsigned char *dst = ss + it * 16;             // (= (kg * BQ + q) * 16)
          *reinterpret_cast<u32x4s *>(dst) = h;
          *reinterpret_cast<u32x4s *>(dst + piece) = m;
          *reinterpret_cast<u32x4s *>(dst + 2 * piece) = l;
        }
      }
    }
  }
  if (EPI == EPI_RANK) {
    for (int i = tid; i < BQ * 3; i += SS_THREADS) cnt[i] = 0;
  }
  __syncthreads();

  unsigned int my_gt[NQT], my_tl[NQT], my_ti[NQT];
  float tgt[NQT];
  int64_t ob[NQT];
  if (EPI == EPI_RANK) {
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
      my_gt[t] = my_tl[t] = my_ti[t] = 0;
      int col = c0 + t * 16 + fr;
      col = col < p.ncols ? col : p.ncols - 1;
      tgt[t] = p.target[col];
      ob[t] = p.obj[col] - p.row0;
    }
  }
  const int tiles16 = int((p.m + 15) >> 4);
  // A wave takes PAIRS of 16-entity row tiles (32 rows = one filter word per query): the query fragments it reads from LDS
  // feed both tiles' MFMAs, and 48 MFMAs per k-block cover the latency of the next k-block's row loads.
  // TARGET: row i is query i's target entity and only the diagonal is wanted: row tile (strip, t) meets column tile t.
  constexpr int RT = EPI == EPI_TARGET ? 1 : 2;
  const int units = EPI == EPI_TARGET ? tiles16 : (tiles16 + 1) >> 1;
  int tu = EPI == EPI_TARGET ? int(blockIdx.y) * NQT + wave : int(blockIdx.x) * 8 + wave;
  const int step = EPI == EPI_TARGET ? units : int(gridDim.x) * 8;
  if (EPI == EPI_TARGET && wave >= NQT) tu = units;
  for (; tu < units; tu += step) {
    const float *ap[RT];
    bool a_ok[RT];
#pragma unroll
    for (int h = 0; h < RT; ++h) {
      int64_t arow = (int64_t(tu) * RT + h) * 16 + fr;
      a_ok[h] = true;
      if (EPI == EPI_TARGET) {
        a_ok[h] = arow < p.m;
        if (a_ok[h]) {
          const int64_t o = p.obj[arow] - p.row0;    // the entity row query `arow` must be scored against
          a_ok[h] = o >= 0 && o < p.n_local;
          arow = o;
        }
      } else {
        arow = arow < p.m ? arow : p.m - 1;          // rows past M: a valid row, never stored or counted
      }
      ap[h] = p.a + (a_ok[h] ? arow : 0) * p.lda + 8 * fq;
    }
    auto aload = [&](float4 (&lo)[RT], float4 (&hi)[RT], int kb) {
      const int k0 = 32 * kb + 8 * fq;
#pragma unroll
      for (int h = 0; h < RT; ++h) {
        lo[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        hi[h] = lo[h];
        if (a_ok[h] && k0 < p.k) lo[h] = *reinterpret_cast<const float4 *>(ap[h] + 32 * kb);
        if (a_ok[h] && k0 + 4 < p.k) hi[h] = *reinterpret_cast<const float4 *>(ap[h] + 32 * kb + 4);
      }
    };
    f32x4 acc[RT][NQT];
#pragma unroll
    for (int h = 0; h < RT; ++h) {
#pragma unroll
      for (int t = 0; t < NQT; ++t) acc[h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float4 rlo[2][RT], rhi[2][RT];
    aload(rlo[0], rhi[0], 0);
    uint32_t mw[NQT];
    float bv[RT][4];
    if (EPI == EPI_RANK) {
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        int col = c0 + t * 16 + fr;
        col = col < p.ncols ? col : p.ncols - 1;
        mw[t] = p.mask ? p.mask[int64_t(col) * p.ldl + tu] : 0u;   // the pair's 32 rows = one filter word of the query
      }
    }
    if (EPI == EPI_RANK || EPI == EPI_SIGMOID) {
#pragma unroll
      for (int h = 0; h < RT; ++h) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          int64_t row = (int64_t(tu) * RT + h) * 16 + fq * 4 + j;
          row = row < p.m ? row : p.m - 1;
          bv[h][j] = p.bias[row];
        }
      }
    }
    auto kblock = [&](const float4 (&alo)[RT], const float4 (&ahi)[RT], int kb) __attribute__((always_inline)) {
      u32x4s ah[RT], am[RT], al[RT];
#pragma unroll
      for (int h = 0; h < RT; ++h) split8_(alo[h], ahi[h], ah[h], am[h], al[h]);
      const unsigned char *bp = ss + ((kb * 4 + fq) * BQ + fr) * 16;
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        if (EPI == EPI_TARGET && t != wave) continue;
        const bf16x8s bh = __builtin_bit_cast(bf16x8s, *reinterpret_cast<const u32x4s *>(bp + t * 256));
        const bf16x8s bm = __builtin_bit_cast(bf16x8s, *reinterpret_cast<const u32x4s *>(bp + piece + t * 256));
        const bf16x8s bl = __builtin_bit_cast(bf16x8s, *reinterpret_cast<const u32x4s *>(bp + 2 * piece + t * 256));
#pragma unroll
        for (int h = 0; h < RT; ++h) {
          f32x4 c = acc[h][t];   // the six products, small terms first
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, ah[h]), bl, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, al[h]), bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, am[h]), bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, ah[h]), bm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, am[h]), bh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8s, ah[h]), bh, c, 0, 0, 0);
          acc[h][t] = c;
        }
      }
    };
    for (int kb = 0; kb < nkb; ++kb) {
      if (kb + 1 < nkb) aload(rlo[1], rhi[1], kb + 1);
      kblock(rlo[0], rhi[0], kb);
#pragma unroll
      for (int h = 0; h < RT; ++h) {
        rlo[0][h] = rlo[1][h];
        rhi[0][h] = rhi[1][h];
      }
    }
    // lane holds entities (RT tu + h) * 16 + 4 fq + j (j = 0..3) of query column c0 + 16 t + fr
#pragma unroll
    for (int h = 0; h < RT; ++h) {
      const int64_t r0 = (int64_t(tu) * RT + h) * 16 + fq * 4;
#pragma unroll
      for (int t = 0; t < NQT; ++t) {
        const int col = c0 + t * 16 + fr;
        if (col >= p.ncols) continue;
        if (EPI == EPI_SIGMOID) {
          float sc[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) sc[j] = sigmoidf_(acc[h][t][j] + bv[h][j]);
          float *dst = p.c + int64_t(col) * p.ldc + r0;
          if (r0 + 3 < p.m && (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.c) & 15u) == 0) {
            *reinterpret_cast<float4 *>(dst) = make_float4(sc[0], sc[1], sc[2], sc[3]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (r0 + j < p.m) dst[j] = sc[j];
          }
        } else if (EPI == EPI_TARGET) {
          if (t != wave) continue;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t row = r0 + j;                 // query index of this accumulator row
            if (row == col && row < p.m) {
              const int64_t o = p.obj[row] - p.row0;
              if (o >= 0 && o < p.n_local) p.target_out[row] = sigmoidf_(acc[h][t][j] + p.bias[o]);
            }
          }
        } else if (EPI == EPI_RANK) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t row = r0 + j;
            if (row >= p.m || row == ob[t]) continue;                          // the target itself (main.py:125)
            if (p.mask) {
              if ((mw[t] >> (h * 16 + fq * 4 + j)) & 1u) continue;
            } else {
              const float lab = p.label[int64_t(col) * p.ldl + row];
              if ((static_cast<int>(lab) & 0xff) != 0) continue;                // label.byte() filter (main.py:124)
            }
            const float sc = sigmoidf_(acc[h][t][j] + bv[h][j]);
            my_gt[t] += sc > tgt[t];
            const bool eq = sc == tgt[t];
            my_ti[t] += eq;
            my_tl[t] += eq && (row < ob[t]);
          }
        }
      }
    }
  }
  if (EPI == EPI_RANK) {
#pragma unroll
    for (int t = 0; t < NQT; ++t) {
      unsigned int g = my_gt[t], l = my_tl[t], e = my_ti[t];
      g += __shfl_xor(g, 16); l += __shfl_xor(l, 16); e += __shfl_xor(e, 16);
      g += __shfl_xor(g, 32); l += __shfl_xor(l, 32); e += __shfl_xor(e, 32);
      if (fq == 0) {
        atomicAdd(&cnt[(t * 16 + fr) * 3 + 0], g);
        atomicAdd(&cnt[(t * 16 + fr) * 3 + 1], l);
        atomicAdd(&cnt[(t * 16 + fr) * 3 + 2], e);
      }
    }
    __syncthreads();
    for (int i = tid; i < BQ * 3; i += SS_THREADS) {
      const int col = c0 + i / 3;
      if (col < p.ncols && cnt[i]) atomicAdd(&p.counts[int64_t(col) * 3 + i % 3], (unsigned long long)cnt[i]);
    }
  }
}

// all_rel = rels_embs @ rels_weight (model.py:107 without the dropped last row): [T, K] x [K, O], T tiny, so
// the kernel is pure latency. Block = (one output row, 64 columns); its 4 waves split K and keep UNR
// independent loads in flight per lane; partial sums meet in LDS and are added in wave order.
__global__ __launch_bounds__(256) void small_matmul_kernel(const float *__restrict__ a, int64_t lda,
                                                           const float *__restrict__ b, int64_t ldb,
                                                           float *__restrict__ c, int64_t ldc, int k, int n) {
  constexpr int UNR = 8;
  __shared__ float part[4][64];
  const int row = blockIdx.x, col = blockIdx.y * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  const int kper = (k + 3) / 4, k0 = wave * kper, k1 = (k0 + kper < k) ? k0 + kper : k;
  const bool ok = col < n;
  const float *ap = a + int64_t(row) * lda;
  const float *bp = b + (ok ? col : 0);
  float acc = 0.f;
  int kk = k0;
  for (; kk + UNR <= k1; kk += UNR) {
    float av[UNR], bv[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      av[u] = ap[kk + u];
      bv[u] = bp[int64_t(kk + u) * ldb];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc = fmaf(av[u], bv[u], acc);
  }
  for (; kk < k1; ++kk) acc = fmaf(ap[kk], bp[int64_t(kk) * ldb], acc);
  part[wave][threadIdx.x & 63] = acc;
  __syncthreads();
  if (wave == 0 && ok) {
    const int l = threadIdx.x;
    c[int64_t(row) * ldc + col] = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
  }
}

// Filter construction on the device (replaces the dense [B, N] label block of data_loader.py:34-51 for
// evaluation): known (subject, relation) -> tails lists live on the device as a sorted key array + CSR; one wave
// per query finds its key by binary search and sets the bits of the tails that fall into this entity shard.
__global__ __launch_bounds__(256) void filter_mask_kernel(const int64_t *__restrict__ qkey, int batch,
                                                          const int64_t *__restrict__ keys, int64_t nkeys,
                                                          const int64_t *__restrict__ ptr, const int32_t *__restrict__ tails,
                                                          int64_t row0, int64_t n_local, uint32_t *mask, int64_t ldm) {
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= batch) return;
  const int64_t key = qkey[q];
  int64_t lo = 0, hi = nkeys;
  while (lo < hi) {  // wave-uniform
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  if (lo >= nkeys || keys[lo] != key) return;
  for (int64_t i = ptr[lo] + lane; i < ptr[lo + 1]; i += 64) {
    const int64_t n = int64_t(tails[i]) - row0;
    if (n >= 0 && n < n_local) atomicOr(&mask[int64_t(q) * ldm + (n >> 5)], 1u << (n & 31));
  }
}

// Training targets on the device (replaces building the dense [B, N] label block per sample on the host,
// data_loader.py:34-51, and shipping it over PCIe every step): one workgroup per query fills its row with `cold`
// and then writes `hot` at the known tails of its (subject, relation) key — (1 - eps) * y + 1/N with y in {0, 1}
// evaluated on the host exactly as numpy does (data_loader.py:41-43), so the rows are bit-identical.
__global__ __launch_bounds__(256) void label_rows_kernel(const int64_t *__restrict__ qkey, int batch,
                                                         const int64_t *__restrict__ keys, int64_t nkeys,
                                                         const int64_t *__restrict__ ptr, const int32_t *__restrict__ tails,
                                                         int64_t row0, int64_t n_local, float hot, float cold, float *out,
                                                         int64_t ldo) {
  const int q = blockIdx.x;
  float *row = out + int64_t(q) * ldo;
  for (int64_t n = threadIdx.x; n < n_local; n += 256) row[n] = cold;
  const int64_t key = qkey[q];
  int64_t lo = 0, hi = nkeys;
  while (lo < hi) {  // block-uniform
    const int64_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  if (lo >= nkeys || keys[lo] != key) return;
  __syncthreads();   // the fill above is complete (same workgroup wrote the row)
  for (int64_t i = ptr[lo] + threadIdx.x; i < ptr[lo + 1]; i += 256) {
    const int64_t n = int64_t(tails[i]) - row0;
    if (n >= 0 && n < n_local) row[n] = hot;
  }
}

void set_vec_flags(TileArgs *p) {
  p->a_vec = (p->lda % 4 == 0) && mgcn::aligned16(p->a);
  p->b_vec = (p->ldb % 4 == 0) && mgcn::aligned16(p->b);
}

int check_common(const char *who, int64_t m, int32_t k, int64_t ncols) {
  MGCN_REQUIRE(m >= 0 && k > 0 && ncols >= 0, "%s: bad sizes", who);
  MGCN_REQUIRE(m < (int64_t(1) << 31) - 64 && ncols < (int64_t(1) << 31) - 64, "%s: sizes exceed int32", who);
  return MGCN_OK;
}

// column tiles per wave: the smallest instantiated NT that covers the columns in one block, else 8/16-wide strips
int pick_nt(int64_t ncols) {
  if (ncols <= 32) return 2;
  if (ncols <= 64) return 4;
  if (ncols <= 128) return 8;
  if (ncols <= 208) return 13;
  return (ncols % 208 == 0 || ncols > 1024) ? 13 : 8;
}

template <int EPI, bool B_NT>
int launch(TileArgs &p, int64_t grid_x_cap, hipStream_t stream, const char *name) {
  set_vec_flags(&p);
  p.tiles_m = int32_t((p.m + BM - 1) / BM);
  const bool fast = p.a_vec && p.b_vec && p.k % 4 == 0 && (B_NT || p.ncols % 4 == 0);
  const int nt = fast ? pick_nt(p.ncols) : 4;
  const unsigned gy = unsigned((p.ncols + nt * 16 - 1) / (nt * 16));
  const unsigned gx = unsigned(grid_x_cap > 0 && p.tiles_m > grid_x_cap ? grid_x_cap : p.tiles_m);
  if (!fast) {
    hipLaunchKernelGGL((tile_kernel<EPI, B_NT, 4, false>), dim3(gx, gy), dim3(THREADS), 0, stream, p);
  } else {
    switch (nt) {
      case 2: hipLaunchKernelGGL((tile_kernel<EPI, B_NT, 2, true>), dim3(gx, gy), dim3(THREADS), 0, stream, p); break;
      case 4: hipLaunchKernelGGL((tile_kernel<EPI, B_NT, 4, true>), dim3(gx, gy), dim3(THREADS), 0, stream, p); break;
      case 8: hipLaunchKernelGGL((tile_kernel<EPI, B_NT, 8, true>), dim3(gx, gy), dim3(THREADS), 0, stream, p); break;
      default: hipLaunchKernelGGL((tile_kernel<EPI, B_NT, 13, true>), dim3(gx, gy), dim3(THREADS), 0, stream, p); break;
    }
  }
  MGCN_CHECK_LAUNCH(name);
  return MGCN_OK;
}

}  // namespace

extern "C" int mgcn_dense_bn_tanh_fwd(int64_t num_nodes, int32_t dim_in, int32_t dim_out, const float *a_dev,
                                      int64_t lda, const float *w_dev, const float *bias_dev, const float *bn_mean_dev,
                                      const float *bn_var_dev, const float *bn_gamma_dev, const float *bn_beta_dev,
                                      float bn_eps, float *out_dev, int64_t ldo, void *stream) {
  if (int rc = check_common("dense_bn_tanh_fwd", num_nodes, dim_in, dim_out)) return rc;
  MGCN_REQUIRE(a_dev && w_dev && bn_mean_dev && bn_var_dev && bn_gamma_dev && bn_beta_dev && out_dev,
               "dense_bn_tanh_fwd: null pointer");
  MGCN_REQUIRE(lda >= 3 * int64_t(dim_in) && ldo >= dim_out, "dense_bn_tanh_fwd: lda/ldo too small");
  if (num_nodes == 0 || dim_out == 0) return MGCN_OK;
  TileArgs p = {};
  p.a = a_dev; p.lda = lda;
  p.b = w_dev; p.ldb = dim_out;
  p.c = out_dev; p.ldc = ldo;
  p.bias = bias_dev; p.bn_mean = bn_mean_dev; p.bn_var = bn_var_dev; p.bn_gamma = bn_gamma_dev; p.bn_beta = bn_beta_dev;
  p.bn_eps = bn_eps;
  p.m = num_nodes; p.k = 3 * dim_in; p.ncols = dim_out;
  return launch<EPI_BN_TANH, false>(p, 0, static_cast<hipStream_t>(stream), "tile_kernel<BN_TANH>");
}

extern "C" int mgcn_matmul_f32(int64_t m, int32_t k, int32_t n, const float *a_dev, int64_t lda, const float *b_dev,
                               int64_t ldb, float *c_dev, int64_t ldc, void *stream) {
  if (int rc = check_common("matmul_f32", m, k, n)) return rc;
  MGCN_REQUIRE(a_dev && b_dev && c_dev, "matmul_f32: null pointer");
  MGCN_REQUIRE(lda >= k && ldb >= n && ldc >= n, "matmul_f32: leading dimension too small");
  if (m == 0 || n == 0) return MGCN_OK;
  if (m <= 1024 && n <= 4096 && k <= 2048) {  // relation projection: a few dozen rows, latency bound (a long K is the tile kernel's)
    hipLaunchKernelGGL(small_matmul_kernel, dim3(unsigned(m), unsigned((n + 63) / 64)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a_dev, lda, b_dev, ldb, c_dev, ldc, k, n);
    MGCN_CHECK_LAUNCH("small_matmul_kernel");
    return MGCN_OK;
  }
  TileArgs p = {};
  p.a = a_dev; p.lda = lda;
  p.b = b_dev; p.ldb = ldb;
  p.c = c_dev; p.ldc = ldc;
  p.m = m; p.k = k; p.ncols = n;
  return launch<EPI_NONE, false>(p, 0, static_cast<hipStream_t>(stream), "tile_kernel<NONE>");
}

extern "C" int mgcn_filter_mask(int32_t batch, const int64_t *qkey_dev, int64_t num_keys, const int64_t *keys_dev,
                                const int64_t *ptr_dev, const int32_t *tails_dev, int64_t ent_row0, int64_t n_local,
                                uint32_t *mask_dev, int64_t ldm, void *stream) {
  MGCN_REQUIRE(batch >= 0 && num_keys >= 0 && n_local >= 0 && ent_row0 >= 0, "filter_mask: bad sizes");
  MGCN_REQUIRE(ldm >= (n_local + 31) / 32, "filter_mask: mask rows too short");
  if (batch == 0 || n_local == 0) return MGCN_OK;
  MGCN_REQUIRE(qkey_dev && mask_dev && ptr_dev && (num_keys == 0 || (keys_dev && tails_dev)), "filter_mask: null pointer");
  hipError_t e = hipMemsetAsync(mask_dev, 0, size_t(batch) * size_t(ldm) * sizeof(uint32_t), static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return mgcn::fail(MGCN_ELAUNCH, "filter_mask: memset: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(filter_mask_kernel, dim3(unsigned((batch + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     qkey_dev, batch, keys_dev, num_keys, ptr_dev, tails_dev, ent_row0, n_local, mask_dev, ldm);
  MGCN_CHECK_LAUNCH("filter_mask_kernel");
  return MGCN_OK;
}

extern "C" int mgcn_label_rows(int32_t batch, const int64_t *qkey_dev, int64_t num_keys, const int64_t *keys_dev,
                               const int64_t *ptr_dev, const int32_t *tails_dev, int64_t ent_row0, int64_t n_local,
                               float hot, float cold, float *out_dev, int64_t ldo, void *stream) {
  MGCN_REQUIRE(batch >= 0 && num_keys >= 0 && n_local >= 0 && ent_row0 >= 0 && ldo >= n_local, "label_rows: bad sizes");
  if (batch == 0) return MGCN_OK;
  MGCN_REQUIRE(qkey_dev && out_dev && ptr_dev && (num_keys == 0 || (keys_dev && tails_dev)), "label_rows: null pointer");
  hipLaunchKernelGGL(label_rows_kernel, dim3(unsigned(batch)), dim3(256), 0, static_cast<hipStream_t>(stream), qkey_dev,
                     batch, keys_dev, num_keys, ptr_dev, tails_dev, ent_row0, n_local, hot, cold, out_dev, ldo);
  MGCN_CHECK_LAUNCH("label_rows_kernel");
  return MGCN_OK;
}

// Aligned scoring shapes take score_split_kernel (one arithmetic for score / target / rank counts); the rest keep the
// exact-f32 tile kernels — the SAME choice for all three entry points, so their results stay mutually consistent.
bool split_scoring(TileArgs *p) {
  set_vec_flags(p);
  return p->a_vec && p->b_vec && p->k % 4 == 0 && p->k <= SS_MAX_KB * 32;
}

template <int EPI>
int launch_split(const TileArgs &p, hipStream_t stream, const char *name) {
  const int nkb = (p.k + 31) / 32;
  const size_t lds = size_t(3) * nkb * 4 * SS_BQ * 16 + size_t(SS_BQ) * 3 * sizeof(unsigned int);
  // more than 64 KB of dynamic LDS needs an opt-in (sticky per device; set on every call: no state is kept here)
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(score_split_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024) != hipSuccess)
    return mgcn::fail(MGCN_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", name, lds);
  const unsigned gy = unsigned((p.ncols + SS_BQ - 1) / SS_BQ);
  const int64_t tiles16 = (p.m + 15) / 16;
  // Blocks: the kernel's ~190 VGPRs leave room for ONE block of 8 waves per CU, so blocks beyond 256 run as a second round that
  // pays the block's fixed cost (its strip split into LDS, its first row loads, its counters) again. Up to 32 strips exactly 256
  // blocks (128 queries 39 -> 33 us, 512: 82 -> 71, 2048: 246 -> 242: tools/bench_rank_block.py); beyond, 512 / strips per strip
  // keeps the rounds even (6 268 queries: 754 us against 927 with 256 / strips = 2, which leaves 60 CUs idle).
  int64_t gx = (gy <= 32 ? 256 : 512) / gy;
  const int64_t cap = ((tiles16 + 1) / 2 + 7) / 8;        // (a wave takes pairs of row tiles)
  gx = gx < cap ? gx : cap;
  if (EPI == EPI_TARGET || gx < 1) gx = 1;
  hipLaunchKernelGGL(score_split_kernel<EPI>, dim3(unsigned(gx), gy), dim3(SS_THREADS), lds, stream, p);
  MGCN_CHECK_LAUNCH(name);
  return MGCN_OK;
}

extern "C" int mgcn_score_fwd(int32_t batch, int64_t n_local, int32_t dim, const float *x_dev, int64_t ldx,
                              const float *ent_dev, int64_t lde, const float *bias_dev, float *score_dev,
                              int64_t lds, void *stream) {
  if (int rc = check_common("score_fwd", n_local, dim, batch)) return rc;
  MGCN_REQUIRE(x_dev && ent_dev && bias_dev && score_dev, "score_fwd: null pointer");
  MGCN_REQUIRE(ldx >= dim && lde >= dim && lds >= n_local, "score_fwd: leading dimension too small");
  if (batch == 0 || n_local == 0) return MGCN_OK;
  TileArgs p = {};
  p.a = ent_dev; p.lda = lde;
  p.b = x_dev; p.ldb = ldx;
  p.c = score_dev; p.ldc = lds;
  p.bias = bias_dev;
  p.m = n_local; p.k = dim; p.ncols = batch;
  if (split_scoring(&p)) return launch_split<EPI_SIGMOID>(p, static_cast<hipStream_t>(stream), "score_split_kernel<SIGMOID>");
  return launch<EPI_SIGMOID, true>(p, 0, static_cast<hipStream_t>(stream), "tile_kernel<SIGMOID>");
}

extern "C" int64_t mgcn_score_bce_partials(int32_t batch, int64_t n_local) {
  if (batch <= 0 || n_local <= 0) return 0;
  const int64_t tiles_m = (n_local + BM - 1) / BM;
  const int nt = pick_nt(batch);
  return tiles_m * ((batch + nt * 16 - 1) / (nt * 16));
}

extern "C" int mgcn_score_bce_fwd(int32_t batch, int64_t n_local, int32_t dim, const float *x_dev, int64_t ldx,
                                  const float *ent_dev, int64_t lde, const float *bias_dev, const uint32_t *mask_dev,
                                  int64_t ldm, float hot, float cold, float inv_count, float *grad_logit_dev, int64_t ldg,
                                  float *loss_partial_dev, void *stream) {
  if (int rc = check_common("score_bce_fwd", n_local, dim, batch)) return rc;
  MGCN_REQUIRE(x_dev && ent_dev && bias_dev && mask_dev && grad_logit_dev && loss_partial_dev, "score_bce_fwd: null pointer");
  MGCN_REQUIRE(ldx >= dim && lde >= dim && ldg >= batch && ldm >= (n_local + 31) / 32, "score_bce_fwd: leading dimension too small");
  if (batch == 0 || n_local == 0) return MGCN_OK;
  TileArgs p = {};
  p.a = ent_dev; p.lda = lde;
  p.b = x_dev; p.ldb = ldx;
  p.c = grad_logit_dev; p.ldc = ldg;
  p.bias = bias_dev; p.mask = mask_dev; p.ldl = ldm;
  p.hot = hot; p.cold = cold; p.inv_count = inv_count; p.loss_partial = loss_partial_dev;
  p.m = n_local; p.k = dim; p.ncols = batch;
  // the partial-sum count handed to the caller assumes the aligned (FAST) tiling: refuse other shapes
  if (!(mgcn::aligned16(ent_dev) && mgcn::aligned16(x_dev) && mgcn::aligned16(grad_logit_dev) && lde % 4 == 0 &&
        ldx % 4 == 0 && ldg % 4 == 0 && dim % 4 == 0 && batch % 4 == 0))
    return mgcn::fail(MGCN_EUNSUPPORTED, "score_bce_fwd: needs 16-byte aligned operands, dim %% 4 == 0 and batch %% 4 == 0");
  return launch<EPI_BCE, true>(p, 0, static_cast<hipStream_t>(stream), "tile_kernel<BCE>");
}

extern "C" int mgcn_score_target(int32_t batch, int64_t n_local, int64_t ent_row0, int32_t dim, const float *x_dev,
                                 int64_t ldx, const float *ent_dev, int64_t lde, const float *bias_dev,
                                 const int64_t *obj_dev, float *target_dev, void *stream) {
  if (int rc = check_common("score_target", n_local, dim, batch)) return rc;
  MGCN_REQUIRE(x_dev && ent_dev && bias_dev && obj_dev && target_dev, "score_target: null pointer");
  MGCN_REQUIRE(ldx >= dim && lde >= dim && ent_row0 >= 0, "score_target: bad leading dimension / row0");
  if (batch == 0 || n_local == 0) return MGCN_OK;
  TileArgs p = {};
  p.a = ent_dev; p.lda = lde;
  p.b = x_dev; p.ldb = ldx;
  p.n_local = n_local;
  p.bias = bias_dev; p.obj = obj_dev; p.target_out = target_dev; p.row0 = ent_row0;
  p.m = batch; p.k = dim; p.ncols = batch;
  if (split_scoring(&p)) return launch_split<EPI_TARGET>(p, static_cast<hipStream_t>(stream), "score_split_kernel<TARGET>");
  return launch<EPI_TARGET, true>(p, 0, static_cast<hipStream_t>(stream), "tile_kernel<TARGET>");
}

extern "C" int mgcn_score_rank(int32_t batch, int64_t n_local, int64_t ent_row0, int32_t dim, const float *x_dev,
                               int64_t ldx, const float *ent_dev, int64_t lde, const float *bias_dev,
                               const int64_t *obj_dev, const float *target_dev, const float *label_dev, int64_t ldl,
                               const uint32_t *mask_dev, int64_t ldm, int64_t *counts_dev, void *stream) {
  if (int rc = check_common("score_rank", n_local, dim, batch)) return rc;
  MGCN_REQUIRE(x_dev && ent_dev && bias_dev && obj_dev && target_dev && counts_dev, "score_rank: null pointer");
  MGCN_REQUIRE((label_dev != nullptr) != (mask_dev != nullptr), "score_rank: give exactly one of label / mask");
  MGCN_REQUIRE(ldx >= dim && lde >= dim && ent_row0 >= 0, "score_rank: bad leading dimension / row0");
  MGCN_REQUIRE(label_dev ? ldl >= n_local : ldm >= (n_local + 31) / 32, "score_rank: filter rows too short");
  if (batch == 0 || n_local == 0) return MGCN_OK;
  TileArgs p = {};
  p.a = ent_dev; p.lda = lde;
  p.b = x_dev; p.ldb = ldx;
  p.bias = bias_dev; p.obj = obj_dev; p.target = target_dev; p.label = label_dev; p.mask = mask_dev;
  p.ldl = label_dev ? ldl : ldm;
  p.counts = reinterpret_cast<unsigned long long *>(counts_dev); p.row0 = ent_row0;
  p.m = n_local; p.k = dim; p.ncols = batch;
  if (split_scoring(&p)) return launch_split<EPI_RANK>(p, static_cast<hipStream_t>(stream), "score_split_kernel<RANK>");
  p.tiles_m = int32_t((p.m + BM - 1) / BM);
  return launch<EPI_RANK, true>(p, 1280, static_cast<hipStream_t>(stream), "tile_kernel<RANK>");
}

