// Training-mode layer epilogue and its backward (gfx950): what model.py:103-106 does in .train() —
//   out = (drop(in_res) + drop(out_res) + loop_res) / 3 (+ bias);  all_ent = tanh(BatchNorm1d(out))  with BATCH statistics —
// and the weight-gradient product of model.py:116's backward, A^T G, as a split-K MFMA kernel. All reductions over the
// N rows are two-stage (fixed row blocks, partial sums added in block order): no atomics, bitwise reproducible.
#include <hip/hip_runtime.h>

#include "mgcn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int RB = 128;          // rows per partial-sum block
constexpr int TPB = 256;

__device__ __forceinline__ float tanh_ref(float v) { return tanhf(v); }

// z = (a + b + c) / 3 (+ bias); partial column sums of z per row block.
__global__ __launch_bounds__(TPB) void combine_sum_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                          const float *__restrict__ c, int64_t ldu, const float *__restrict__ bias,
                                                          float *__restrict__ z, float *__restrict__ part, int64_t n, int o) {
  const int64_t r0 = int64_t(blockIdx.x) * RB, r1 = (r0 + RB < n) ? r0 + RB : n;
  for (int col = threadIdx.x; col < o; col += TPB) {
    const float bc = bias ? bias[col] : 0.f;
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      float v = (a[r * ldu + col] + b[r * ldu + col] + c[r * ldu + col]) / 3.0f;   // model.py:103 (division, as torch)
      if (bias) v += bc;
      z[r * o + col] = v;
      s += v;
    }
    part[int64_t(blockIdx.x) * o + col] = s;
  }
}

// generic partial column sums of f(row) over a row block: MODE 0: (z - mean)^2; MODE 1: g_pre and g_pre * xhat
template <int MODE>
__global__ __launch_bounds__(TPB) void partial_kernel(const float *__restrict__ z, const float *__restrict__ y,
                                                      const float *__restrict__ gy, const float *__restrict__ mean,
                                                      const float *__restrict__ rstd, float *__restrict__ part0,
                                                      float *__restrict__ part1, int64_t n, int o) {
  const int64_t r0 = int64_t(blockIdx.x) * RB, r1 = (r0 + RB < n) ? r0 + RB : n;
  for (int col = threadIdx.x; col < o; col += TPB) {
    const float m = mean[col];
    float s0 = 0.f, s1 = 0.f;
    if (MODE == 0) {
      for (int64_t r = r0; r < r1; ++r) {
        const float d = z[r * o + col] - m;
        s0 += d * d;
      }
      part0[int64_t(blockIdx.x) * o + col] = s0;
    } else {
      const float rs = rstd[col];
      for (int64_t r = r0; r < r1; ++r) {
        const float yy = y[r * o + col];
        const float gp = gy[r * o + col] * (1.0f - yy * yy);       // d tanh
        s0 += gp;
        s1 += gp * ((z[r * o + col] - m) * rs);
      }
      part0[int64_t(blockIdx.x) * o + col] = s0;
      part1[int64_t(blockIdx.x) * o + col] = s1;
    }
  }
}

// out[col] = sum over blocks (in block order) of part[blk][col], times scale
__global__ __launch_bounds__(TPB) void fold_kernel(const float *__restrict__ part, int nblk, int o, float scale, float *__restrict__ out) {
  const int col = blockIdx.x * TPB + threadIdx.x;
  if (col >= o) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[int64_t(b) * o + col];
  out[col] = s * scale;
}

// var (biased) -> rstd; running statistics as nn.BatchNorm1d: unbiased variance, momentum update
__global__ __launch_bounds__(TPB) void stats_finish_kernel(const float *__restrict__ part, int nblk, int o, int64_t n, float eps,
                                                           float momentum, const float *__restrict__ mean, float *__restrict__ rstd,
                                                           float *__restrict__ running_mean, float *__restrict__ running_var) {
  const int col = blockIdx.x * TPB + threadIdx.x;
  if (col >= o) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[int64_t(b) * o + col];
  const float var = s / float(n);
  rstd[col] = 1.0f / sqrtf(var + eps);
  if (running_mean) {
    const float unbiased = n > 1 ? s / float(n - 1) : var;
    running_mean[col] = (1.0f - momentum) * running_mean[col] + momentum * mean[col];
    running_var[col] = (1.0f - momentum) * running_var[col] + momentum * unbiased;
  }
}

__global__ __launch_bounds__(TPB) void apply_fwd_kernel(const float *__restrict__ z, const float *__restrict__ mean,
                                                        const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, float *__restrict__ y, int64_t total, int o) {
  const int64_t i = int64_t(blockIdx.x) * TPB + threadIdx.x;
  if (i >= total) return;
  const int col = int(i % o);
  y[i] = tanh_ref(((z[i] - mean[col]) * rstd[col]) * gamma[col] + beta[col]);
}

// gu = gz / 3 with gz = gamma * rstd * (g_pre - sum_g / n - xhat * sum_gx / n)
__global__ __launch_bounds__(TPB) void apply_bwd_kernel(const float *__restrict__ z, const float *__restrict__ y,
                                                        const float *__restrict__ gy, const float *__restrict__ mean,
                                                        const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                        const float *__restrict__ sum_g, const float *__restrict__ sum_gx,
                                                        float *__restrict__ gz, float *__restrict__ gu, int64_t total, int o, float inv_n) {
  const int64_t i = int64_t(blockIdx.x) * TPB + threadIdx.x;
  if (i >= total) return;
  const int col = int(i % o);
  const float yy = y[i];
  const float gp = gy[i] * (1.0f - yy * yy);
  const float xhat = (z[i] - mean[col]) * rstd[col];
  const float v = (gamma[col] * rstd[col]) * (gp - sum_g[col] * inv_n - xhat * (sum_gx[col] * inv_n));
  gz[i] = v;
  gu[i] = v / 3.0f;
}

// C[M, Nc] = A^T B for A [K, M] (lda), B [K, Nc] (ldb): split over K in gridDim.x row ranges; 8 waves; wave w owns the
// column tiles {w, w + 8} x all row tiles of C (M <= 208, Nc <= 256). v_mfma_f32_16x16x4_f32 (exact f32).
constexpr int TN_THREADS = 512, TN_MT = 13;
__global__ __launch_bounds__(TN_THREADS) void matmul_tn_partial_kernel(const float *__restrict__ a, int64_t lda,
                                                                       const float *__restrict__ b, int64_t ldb,
                                                                       float *__restrict__ part, int64_t k, int m, int nc, int64_t kper) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx = lane & 15, kk = lane >> 4;
  const int mt = (m + 15) / 16, nt = (nc + 15) / 16;
  const int64_t k0 = int64_t(blockIdx.x) * kper, k1 = (k0 + kper < k) ? k0 + kper : k;
  f32x4 acc[TN_MT][2];
#pragma unroll
  for (int i = 0; i < TN_MT; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n0 = wave * 16 + idx, n1 = (wave + 8) * 16 + idx;
  const bool have0 = wave < nt, have1 = wave + 8 < nt;
  for (int64_t r = k0; r < k1; r += 4) {
    const int64_t row = r + kk;
    const bool rok = row < k1;
    const float *ar = a + (rok ? row : k0) * lda;
    const float *br = b + (rok ? row : k0) * ldb;
    const float b0 = (rok && have0 && n0 < nc) ? br[n0] : 0.f;
    const float b1 = (rok && have1 && n1 < nc) ? br[n1] : 0.f;
    float av[TN_MT];
#pragma unroll
    for (int i = 0; i < TN_MT; ++i) {
      const int mm = i * 16 + idx;
      av[i] = (rok && i < mt && mm < m) ? ar[mm] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TN_MT; ++i) {
      if (i < mt) {
        if (have0) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b0, acc[i][0], 0, 0, 0);
        if (have1) acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], b1, acc[i][1], 0, 0, 0);
      }
    }
  }
  // lane holds C rows 16 i + 4 kk + j, column 16 (wave [+8]) + idx
  float *pb = part + int64_t(blockIdx.x) * m * nc;
#pragma unroll
  for (int i = 0; i < TN_MT; ++i) {
    if (i >= mt) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int mm = i * 16 + 4 * kk + j;
      if (mm < m) {
        if (have0 && n0 < nc) pb[int64_t(mm) * nc + n0] = acc[i][0][j];
        if (have1 && n1 < nc) pb[int64_t(mm) * nc + n1] = acc[i][1][j];
      }
    }
  }
}

__global__ __launch_bounds__(TPB) void matmul_tn_fold_kernel(const float *__restrict__ part, int nblk, int64_t mn, float *__restrict__ c,
                                                             int nc, int64_t ldc) {
  const int64_t i = int64_t(blockIdx.x) * TPB + threadIdx.x;
  if (i >= mn) return;
  float s = 0.f;
  for (int b = 0; b < nblk; ++b) s += part[int64_t(b) * mn + i];
  c[(i / nc) * ldc + (i % nc)] = s;
}

int tn_blocks(int64_t k) {
  int64_t nb = (k + 255) / 256;
  return int(nb < 1 ? 1 : (nb > 256 ? 256 : nb));
}

}  // namespace

extern "C" size_t mgcn_bn_tanh_train_workspace(int64_t num_rows, int32_t dim_out) {
  const int64_t nblk = (num_rows + RB - 1) / RB;
  return size_t(2 * nblk * dim_out + 2 * dim_out) * sizeof(float);
}

extern "C" int mgcn_bn_tanh_train_fwd(int64_t num_rows, int32_t dim_out, const float *u_in_dev, const float *u_out_dev,
                                      const float *u_loop_dev, int64_t ldu, const float *bias_dev, const float *gamma_dev,
                                      const float *beta_dev, float *running_mean_dev, float *running_var_dev, float momentum,
                                      float eps, float *z_dev, float *y_dev, float *save_mean_dev, float *save_rstd_dev,
                                      float *workspace_dev, size_t workspace_bytes, void *stream) {
  MGCN_REQUIRE(num_rows > 0 && dim_out > 0, "bn_tanh_train_fwd: bad sizes");
  MGCN_REQUIRE(u_in_dev && u_out_dev && u_loop_dev && gamma_dev && beta_dev && z_dev && y_dev && save_mean_dev && save_rstd_dev &&
                   workspace_dev, "bn_tanh_train_fwd: null pointer");
  MGCN_REQUIRE(ldu >= dim_out && (running_mean_dev != nullptr) == (running_var_dev != nullptr), "bn_tanh_train_fwd: bad arguments");
  MGCN_REQUIRE(workspace_bytes >= mgcn_bn_tanh_train_workspace(num_rows, dim_out), "bn_tanh_train_fwd: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nblk = int((num_rows + RB - 1) / RB), o = dim_out;
  float *part = workspace_dev;
  const unsigned cg = unsigned((o + TPB - 1) / TPB);
  hipLaunchKernelGGL(combine_sum_kernel, dim3(nblk), dim3(TPB), 0, st, u_in_dev, u_out_dev, u_loop_dev, ldu, bias_dev, z_dev, part,
                     num_rows, o);
  hipLaunchKernelGGL(fold_kernel, dim3(cg), dim3(TPB), 0, st, part, nblk, o, 1.0f / float(num_rows), save_mean_dev);
  hipLaunchKernelGGL((partial_kernel<0>), dim3(nblk), dim3(TPB), 0, st, z_dev, nullptr, nullptr, save_mean_dev, nullptr, part, nullptr,
                     num_rows, o);
  hipLaunchKernelGGL(stats_finish_kernel, dim3(cg), dim3(TPB), 0, st, part, nblk, o, num_rows, eps, momentum, save_mean_dev,
                     save_rstd_dev, running_mean_dev, running_var_dev);
  const int64_t total = num_rows * o;
  hipLaunchKernelGGL(apply_fwd_kernel, dim3(unsigned((total + TPB - 1) / TPB)), dim3(TPB), 0, st, z_dev, save_mean_dev, save_rstd_dev,
                     gamma_dev, beta_dev, y_dev, total, o);
  MGCN_CHECK_LAUNCH("bn_tanh_train_fwd");
  return MGCN_OK;
}

extern "C" int mgcn_bn_tanh_train_bwd(int64_t num_rows, int32_t dim_out, const float *z_dev, const float *y_dev, const float *gy_dev,
                                      const float *save_mean_dev, const float *save_rstd_dev, const float *gamma_dev, float *gz_dev,
                                      float *gu_dev, float *ggamma_dev, float *gbeta_dev, float *workspace_dev,
                                      size_t workspace_bytes, void *stream) {
  MGCN_REQUIRE(num_rows > 0 && dim_out > 0, "bn_tanh_train_bwd: bad sizes");
  MGCN_REQUIRE(z_dev && y_dev && gy_dev && save_mean_dev && save_rstd_dev && gamma_dev && gz_dev && gu_dev && ggamma_dev && gbeta_dev &&
                   workspace_dev, "bn_tanh_train_bwd: null pointer");
  MGCN_REQUIRE(workspace_bytes >= mgcn_bn_tanh_train_workspace(num_rows, dim_out), "bn_tanh_train_bwd: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nblk = int((num_rows + RB - 1) / RB), o = dim_out;
  float *p0 = workspace_dev, *p1 = workspace_dev + int64_t(nblk) * o;
  const unsigned cg = unsigned((o + TPB - 1) / TPB);
  hipLaunchKernelGGL((partial_kernel<1>), dim3(nblk), dim3(TPB), 0, st, z_dev, y_dev, gy_dev, save_mean_dev, save_rstd_dev, p0, p1,
                     num_rows, o);
  hipLaunchKernelGGL(fold_kernel, dim3(cg), dim3(TPB), 0, st, p0, nblk, o, 1.0f, gbeta_dev);      // d beta  = sum g_pre
  hipLaunchKernelGGL(fold_kernel, dim3(cg), dim3(TPB), 0, st, p1, nblk, o, 1.0f, ggamma_dev);     // d gamma = sum g_pre * xhat
  const int64_t total = num_rows * o;
  hipLaunchKernelGGL(apply_bwd_kernel, dim3(unsigned((total + TPB - 1) / TPB)), dim3(TPB), 0, st, z_dev, y_dev, gy_dev, save_mean_dev,
                     save_rstd_dev, gamma_dev, gbeta_dev, ggamma_dev, gz_dev, gu_dev, total, o, 1.0f / float(num_rows));
  MGCN_CHECK_LAUNCH("bn_tanh_train_bwd");
  return MGCN_OK;
}

extern "C" size_t mgcn_matmul_tn_workspace(int64_t k, int32_t m, int32_t n) { return size_t(tn_blocks(k)) * m * n * sizeof(float); }

extern "C" int mgcn_matmul_tn_f32(int64_t k, int32_t m, int32_t n, const float *a_dev, int64_t lda, const float *b_dev, int64_t ldb,
                                  float *c_dev, int64_t ldc, float *workspace_dev, size_t workspace_bytes, void *stream) {
  MGCN_REQUIRE(k > 0 && m > 0 && n > 0, "matmul_tn_f32: bad sizes");
  MGCN_REQUIRE(a_dev && b_dev && c_dev && workspace_dev, "matmul_tn_f32: null pointer");
  MGCN_REQUIRE(lda >= m && ldb >= n && ldc >= n, "matmul_tn_f32: leading dimension too small");
  if (m > 16 * TN_MT || n > 256) return mgcn::fail(MGCN_EUNSUPPORTED, "matmul_tn_f32: M <= %d and N <= 256 (got %d, %d)", 16 * TN_MT, m, n);
  MGCN_REQUIRE(workspace_bytes >= mgcn_matmul_tn_workspace(k, m, n), "matmul_tn_f32: workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nblk = tn_blocks(k);
  const int64_t kper = ((k + nblk - 1) / nblk + 3) / 4 * 4;
  hipLaunchKernelGGL(matmul_tn_partial_kernel, dim3(nblk), dim3(TN_THREADS), 0, st, a_dev, lda, b_dev, ldb, workspace_dev, k, m, n, kper);
  const int64_t mn = int64_t(m) * n;
  hipLaunchKernelGGL(matmul_tn_fold_kernel, dim3(unsigned((mn + TPB - 1) / TPB)), dim3(TPB), 0, st, workspace_dev, nblk, mn, c_dev, n, ldc);
  MGCN_CHECK_LAUNCH("matmul_tn_f32");
  return MGCN_OK;
}
