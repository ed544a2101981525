// f32 MFMA tile kernel (v_mfma_f32_16x16x4_f32: exact f32, a k-ordered fma chain) with the fused
// epilogues of the M-GCN hot path (gfx950):
//   EPI_NONE      C = A B                                  (relation projection, model.py:107)
//   EPI_BN_TANH   out = tanh(BN_eval((A [W_in;W_out;W_loop]) / 3 + bias))   (model.py:103-106,116)
//   EPI_SIGMOID   score[b, n] = sigmoid(ent[n,:] . x[b,:] + bias[n])        (model.py:177-179)
//   EPI_TARGET    target[b]   = score[b, obj[b]]  (same tile arithmetic, gathered rows, diagonal)
//   EPI_RANK      filtered counts gt / ties_lower / ties per query, scores never stored (main.py:122-126)
// The streamed operand (aggregates [N,3D] or the entity table [N,O]) is always the MFMA A operand; the
// small operand (weights, or the query block x, read transposed) is B. Block = 4 waves, tile 64 x 64,
// K slabs of 16 through LDS; each wave owns a 16-row strip and four 16x16 accumulators.
#include <hip/hip_runtime.h>

#include "mgcn_common.h"

namespace {

enum { EPI_NONE = 0, EPI_BN_TANH = 1, EPI_SIGMOID = 2, EPI_TARGET = 3, EPI_RANK = 4 };

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TileArgs {
  const float *a;        // [M, K], row stride lda (EPI_TARGET: row i is a[(obj[i]-row0)*lda])
  const float *b[3];     // NN: K rows split over up to 3 matrices of `ksplit` rows, row stride ldb
                         // NT: b[0] = x [ncols, K], row stride ldb
  float *c;              // output (NONE / BN_TANH: [M, ldc]; SIGMOID: [ncols, ldc] transposed store)
  const float *bias;     // BN_TANH: [ncols] or null; scoring: [M] per entity
  const float *bn_mean, *bn_var, *bn_gamma, *bn_beta;
  const int64_t *obj;    // scoring: [ncols] target entity (global id) per query
  const float *target;   // RANK: [ncols]
  float *target_out;     // TARGET: [ncols]
  const float *label;    // RANK: [ncols, ldl]
  unsigned long long *counts;  // RANK: [ncols, 3]
  int64_t lda, ldb, ldc, ldl, m, row0, n_local;
  int32_t k, ncols, ksplit, tiles_m, tiles_n;
  int32_t a_vec, b_vec;  // 16-byte loads allowed (alignment + leading dimension checked on the host)
  float bn_eps;
};

constexpr int BM = 64, BN = 64, BK = 16;
constexpr int LDAS = BK + 1;   // A tile [BM][LDAS]
constexpr int LDBS = BN + 16;  // B tile [BK][LDBS]: rows 16 banks apart -> conflict-free fragment reads

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

template <int EPI, bool B_NT>
__global__ __launch_bounds__(256) void tile_kernel(TileArgs p) {
  __shared__ float As[BM * LDAS];
  __shared__ float Bs[BK * LDBS];
  __shared__ unsigned int cnt[EPI == EPI_RANK ? BN * 3 : 1];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fq = lane >> 4;

  // RANK walks several row tiles per block (grid-stride) and keeps its counts on chip.
  const int tn = (EPI == EPI_RANK) ? int(blockIdx.y) : int(blockIdx.x / p.tiles_m);
  const int tm_first = (EPI == EPI_RANK) ? int(blockIdx.x) : int(blockIdx.x % p.tiles_m);
  const int tm_step = (EPI == EPI_RANK) ? int(gridDim.x) : p.tiles_m;
  const int c0 = tn * BN;

  if (EPI == EPI_RANK) {
    for (int i = tid; i < BN * 3; i += 256) cnt[i] = 0;
  }
  unsigned int my_gt[4] = {0, 0, 0, 0}, my_tl[4] = {0, 0, 0, 0}, my_ti[4] = {0, 0, 0, 0};

  const bool a_vec = p.a_vec != 0, b_vec = p.b_vec != 0;

  for (int tm = tm_first; tm < p.tiles_m; tm += tm_step) {
    const int64_t r0 = int64_t(tm) * BM;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the A row this thread stages: row tid/4 of the tile, k quad (tid%4)*4
    const int ar = tid >> 2, akq = (tid & 3) * 4;
    int64_t arow = r0 + ar;
    bool arow_ok = arow < p.m;
    if (EPI == EPI_TARGET && arow_ok) {
      const int64_t o = p.obj[arow] - p.row0;   // gathered entity row of query `arow`
      arow_ok = o >= 0 && o < p.n_local;
      arow = o;
    }
    const float *aptr = p.a + (arow_ok ? arow : 0) * p.lda;

    for (int k0 = 0; k0 < p.k; k0 += BK) {
      // ---- stage A tile ------------------------------------------------------------------
      {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int kk = k0 + akq;
        if (arow_ok) {
          if (a_vec && kk + 3 < p.k) {
            const float4 t = *reinterpret_cast<const float4 *>(aptr + kk);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (kk + i < p.k) v[i] = aptr[kk + i];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) As[ar * LDAS + akq + i] = v[i];
      }
      // ---- stage B tile ------------------------------------------------------------------
      if (B_NT) {  // Bs[k][c] = x[c0 + c][k0 + k]
        const int bc = tid >> 2, bkq = (tid & 3) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int kk = k0 + bkq;
        if (c0 + bc < p.ncols) {
          const float *bp = p.b[0] + int64_t(c0 + bc) * p.ldb;
          if (b_vec && kk + 3 < p.k) {
            const float4 t = *reinterpret_cast<const float4 *>(bp + kk);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (kk + i < p.k) v[i] = bp[kk + i];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) Bs[(bkq + i) * LDBS + bc] = v[i];
      } else {  // Bs[k][c] = W_{(k0+k)/ksplit}[(k0+k)%ksplit][c0 + c]
        const int bk = tid >> 4, bcq = (tid & 15) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int kk = k0 + bk;
        if (kk < p.k) {
          const int seg = kk / p.ksplit;
          const float *bp = p.b[seg] + int64_t(kk - seg * p.ksplit) * p.ldb;
          const int cc = c0 + bcq;
          if (b_vec && cc + 3 < p.ncols) {
            const float4 t = *reinterpret_cast<const float4 *>(bp + cc);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (cc + i < p.ncols) v[i] = bp[cc + i];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) Bs[bk * LDBS + bcq + i] = v[i];
      }
      __syncthreads();
      // ---- 4 k-steps x 4 column tiles of v_mfma_f32_16x16x4_f32 ----------------------------
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        const float a = As[(wave * 16 + fr) * LDAS + ks * 4 + fq];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float b = Bs[(ks * 4 + fq) * LDBS + t * 16 + fr];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
      }
      __syncthreads();
    }

    // ---- epilogue: lane holds rows r0 + wave*16 + fq*4 + j (j = 0..3), column c0 + t*16 + fr --
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = c0 + t * 16 + fr;
      if (col >= p.ncols) continue;
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
        const int64_t row = r0 + wave * 16 + fq * 4 + j;
        if (row >= p.m) continue;
        const float v = acc[t][j];
        if (EPI == EPI_NONE) {
          p.c[row * p.ldc + col] = v;
        } else if (EPI == EPI_BN_TANH) {
          float o = v / 3.0f;
          if (p.bias) o = o + cb;
          o = (o - mean) * inv * gam + bet;
          p.c[row * p.ldc + col] = tanhf(o);
        } else if (EPI == EPI_SIGMOID) {
          p.c[int64_t(col) * p.ldc + row] = sigmoidf_(v + p.bias[row]);
        } else if (EPI == EPI_TARGET) {
          if (row == col) {
            const int64_t o = p.obj[row] - p.row0;
            if (o >= 0 && o < p.n_local) p.target_out[row] = sigmoidf_(v + p.bias[o]);
          }
        } else if (EPI == EPI_RANK) {
          if (row == ob) continue;                                       // the target itself (main.py:125)
          const float lab = p.label[int64_t(col) * p.ldl + row];
          if ((static_cast<int>(lab) & 0xff) != 0) continue;              // label.byte() filter (main.py:124)
          const float s = sigmoidf_(v + p.bias[row]);
          my_gt[t] += s > tgt;
          const bool eq = s == tgt;
          my_ti[t] += eq;
          my_tl[t] += eq && (row < ob);
        }
      }
    }
  }

  if (EPI == EPI_RANK) {
    // lanes fr, fr+16, fr+32, fr+48 hold the same query column: fold, then one LDS add per wave
#pragma unroll
    for (int t = 0; t < 4; ++t) {
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
    for (int i = tid; i < BN * 3; i += 256) {
      const int col = c0 + i / 3;
      if (col < p.ncols && cnt[i]) atomicAdd(&p.counts[int64_t(col) * 3 + i % 3], (unsigned long long)cnt[i]);
    }
  }
}

void set_vec_flags(TileArgs *p) {
  p->a_vec = (p->lda % 4 == 0) && mgcn::aligned16(p->a);
  p->b_vec = (p->ldb % 4 == 0) && mgcn::aligned16(p->b[0]) && (!p->b[1] || mgcn::aligned16(p->b[1])) &&
             (!p->b[2] || mgcn::aligned16(p->b[2]));
}

int check_common(const char *who, int64_t m, int32_t k, int64_t ncols) {
  MGCN_REQUIRE(m >= 0 && k > 0 && ncols >= 0, "%s: bad sizes", who);
  MGCN_REQUIRE(m < (int64_t(1) << 31) - 64 && ncols < (int64_t(1) << 31) - 64, "%s: sizes exceed int32", who);
  return MGCN_OK;
}

}  // namespace

extern "C" int mgcn_dense_bn_tanh_fwd(int64_t num_nodes, int32_t dim_in, int32_t dim_out, const float *a_dev,
                                      int64_t lda, const float *w_in_dev, const float *w_out_dev,
                                      const float *w_loop_dev, const float *bias_dev, const float *bn_mean_dev,
                                      const float *bn_var_dev, const float *bn_gamma_dev, const float *bn_beta_dev,
                                      float bn_eps, float *out_dev, int64_t ldo, void *stream) {
  if (int rc = check_common("dense_bn_tanh_fwd", num_nodes, dim_in, dim_out)) return rc;
  MGCN_REQUIRE(a_dev && w_in_dev && w_out_dev && w_loop_dev && bn_mean_dev && bn_var_dev && bn_gamma_dev &&
                   bn_beta_dev && out_dev, "dense_bn_tanh_fwd: null pointer");
  MGCN_REQUIRE(lda >= 3 * int64_t(dim_in) && ldo >= dim_out, "dense_bn_tanh_fwd: lda/ldo too small");
  if (num_nodes == 0 || dim_out == 0) return MGCN_OK;
  TileArgs p = {};
  p.a = a_dev; p.lda = lda;
  p.b[0] = w_in_dev; p.b[1] = w_out_dev; p.b[2] = w_loop_dev; p.ldb = dim_out; p.ksplit = dim_in;
  p.c = out_dev; p.ldc = ldo;
  p.bias = bias_dev; p.bn_mean = bn_mean_dev; p.bn_var = bn_var_dev; p.bn_gamma = bn_gamma_dev; p.bn_beta = bn_beta_dev;
  p.bn_eps = bn_eps;
  p.m = num_nodes; p.k = 3 * dim_in; p.ncols = dim_out;
  p.tiles_m = int32_t((num_nodes + BM - 1) / BM); p.tiles_n = (dim_out + BN - 1) / BN;
  set_vec_flags(&p);
  hipLaunchKernelGGL((tile_kernel<EPI_BN_TANH, false>), dim3(unsigned(p.tiles_m) * unsigned(p.tiles_n)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  MGCN_CHECK_LAUNCH("tile_kernel<BN_TANH>");
  return MGCN_OK;
}

extern "C" int mgcn_matmul_f32(int64_t m, int32_t k, int32_t n, const float *a_dev, int64_t lda, const float *b_dev,
                               int64_t ldb, float *c_dev, int64_t ldc, void *stream) {
  if (int rc = check_common("matmul_f32", m, k, n)) return rc;
  MGCN_REQUIRE(a_dev && b_dev && c_dev, "matmul_f32: null pointer");
  MGCN_REQUIRE(lda >= k && ldb >= n && ldc >= n, "matmul_f32: leading dimension too small");
  if (m == 0 || n == 0) return MGCN_OK;
  TileArgs p = {};
  p.a = a_dev; p.lda = lda;
  p.b[0] = b_dev; p.ldb = ldb; p.ksplit = k;
  p.c = c_dev; p.ldc = ldc;
  p.m = m; p.k = k; p.ncols = n;
  p.tiles_m = int32_t((m + BM - 1) / BM); p.tiles_n = (n + BN - 1) / BN;
  set_vec_flags(&p);
  hipLaunchKernelGGL((tile_kernel<EPI_NONE, false>), dim3(unsigned(p.tiles_m) * unsigned(p.tiles_n)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  MGCN_CHECK_LAUNCH("tile_kernel<NONE>");
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
  p.b[0] = x_dev; p.ldb = ldx;
  p.c = score_dev; p.ldc = lds;
  p.bias = bias_dev;
  p.m = n_local; p.k = dim; p.ncols = batch;
  p.tiles_m = int32_t((n_local + BM - 1) / BM); p.tiles_n = (batch + BN - 1) / BN;
  set_vec_flags(&p);
  hipLaunchKernelGGL((tile_kernel<EPI_SIGMOID, true>), dim3(unsigned(p.tiles_m) * unsigned(p.tiles_n)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  MGCN_CHECK_LAUNCH("tile_kernel<SIGMOID>");
  return MGCN_OK;
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
  p.b[0] = x_dev; p.ldb = ldx;
  p.n_local = n_local;
  p.bias = bias_dev; p.obj = obj_dev; p.target_out = target_dev; p.row0 = ent_row0;
  p.m = batch; p.k = dim; p.ncols = batch;
  p.tiles_m = (batch + BM - 1) / BM; p.tiles_n = (batch + BN - 1) / BN;
  set_vec_flags(&p);
  hipLaunchKernelGGL((tile_kernel<EPI_TARGET, true>), dim3(unsigned(p.tiles_m) * unsigned(p.tiles_n)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  MGCN_CHECK_LAUNCH("tile_kernel<TARGET>");
  return MGCN_OK;
}

extern "C" int mgcn_score_rank(int32_t batch, int64_t n_local, int64_t ent_row0, int32_t dim, const float *x_dev,
                               int64_t ldx, const float *ent_dev, int64_t lde, const float *bias_dev,
                               const int64_t *obj_dev, const float *target_dev, const float *label_dev, int64_t ldl,
                               int64_t *counts_dev, void *stream) {
  if (int rc = check_common("score_rank", n_local, dim, batch)) return rc;
  MGCN_REQUIRE(x_dev && ent_dev && bias_dev && obj_dev && target_dev && label_dev && counts_dev,
               "score_rank: null pointer");
  MGCN_REQUIRE(ldx >= dim && lde >= dim && ldl >= n_local && ent_row0 >= 0, "score_rank: bad leading dimension / row0");
  if (batch == 0 || n_local == 0) return MGCN_OK;
  TileArgs p = {};
  p.a = ent_dev; p.lda = lde;
  p.b[0] = x_dev; p.ldb = ldx;
  p.bias = bias_dev; p.obj = obj_dev; p.target = target_dev; p.label = label_dev; p.ldl = ldl;
  p.counts = reinterpret_cast<unsigned long long *>(counts_dev); p.row0 = ent_row0;
  p.m = n_local; p.k = dim; p.ncols = batch;
  p.tiles_m = int32_t((n_local + BM - 1) / BM); p.tiles_n = (batch + BN - 1) / BN;
  const unsigned gx = unsigned(p.tiles_m < 512 ? p.tiles_m : 512);
  set_vec_flags(&p);
  hipLaunchKernelGGL((tile_kernel<EPI_RANK, true>), dim3(gx, unsigned(p.tiles_n)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  MGCN_CHECK_LAUNCH("tile_kernel<RANK>");
  return MGCN_OK;
}
