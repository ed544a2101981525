// Micro-benchmark (not product code): cycles per v_mfma_f32_16x16x32_bf16 and wave, registers only, one and two waves per
// SIMD. (1) a wave cycling through C independent accumulators with constant operands; (2) the multiply role's pattern: per
// row tile 12 MFMAs = six products (w piece, a piece) x two column tiles on two accumulators, five row tiles per k-block,
// distinct operand registers. hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(512) void chains(float *out, unsigned long long *cyc, int iters, unsigned seed) {
  f32x4 acc[C];
  for (int c = 0; c < C; ++c) acc[c] = f32x4{0, 0, 0, 0};
  u32x4 wa = {seed + threadIdx.x, seed * 3u, seed * 5u, seed * 7u}, wb = {seed * 11u, seed + 2u * threadIdx.x, seed * 13u, seed * 17u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, wa), b = __builtin_bit_cast(bf16x8, wb);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 12 / C; ++r) {
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int c = 0; c < C; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ORDER 0: the kernel's order (product-major, column tiles inner); 1: column-tile-major (six products of one accumulator
// back to back, then the other's); 2: product-major over BOTH row tiles of a pair (four accumulators cycling)
template <int ORDER>
__global__ __launch_bounds__(512) void role(float *out, unsigned long long *cyc, int iters, unsigned seed) {
  constexpr int NRT = 4, Q = 2;
  f32x4 acc[NRT][Q];
  for (int r = 0; r < NRT; ++r) for (int t = 0; t < Q; ++t) acc[r][t] = f32x4{0, 0, 0, 0};
  bf16x8 w[Q][3], a[NRT][3];
  for (int t = 0; t < Q; ++t) for (int p = 0; p < 3; ++p) { u32x4 v = {seed + threadIdx.x + t, seed * (3u + p), seed * 5u, seed * 7u + t}; w[t][p] = __builtin_bit_cast(bf16x8, v); }
  for (int r = 0; r < NRT; ++r) for (int p = 0; p < 3; ++p) { u32x4 v = {seed * 11u + r, seed + 2u * threadIdx.x, seed * (13u + p), seed * 17u}; a[r][p] = __builtin_bit_cast(bf16x8, v); }
  constexpr int WP[6] = {0, 2, 1, 0, 1, 0}, AP[6] = {2, 0, 1, 1, 0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (ORDER == 0) {
#pragma unroll
      for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
          for (int t = 0; t < Q; ++t) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][WP[pr]], a[r][AP[pr]], acc[r][t], 0, 0, 0);
    } else if (ORDER == 1) {
#pragma unroll
      for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int t = 0; t < Q; ++t)
#pragma unroll
          for (int pr = 0; pr < 6; ++pr) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][WP[pr]], a[r][AP[pr]], acc[r][t], 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < NRT; r += 2)
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
          for (int t = 0; t < Q; ++t) {
            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][WP[pr]], a[r][AP[pr]], acc[r][t], 0, 0, 0);
            acc[r + 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[t][WP[pr]], a[r + 1][AP[pr]], acc[r + 1][t], 0, 0, 0);
          }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int r = 0; r < NRT; ++r) for (int t = 0; t < Q; ++t) s += acc[r][t][0] + acc[r][t][1] + acc[r][t][2] + acc[r][t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <typename K> void run(K kern, float *out, unsigned long long *cyc, int threads, double per_iter, const char *name) {
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 12345u);
    (void)hipDeviceSynchronize();
  }
  unsigned long long h[8];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-64s %s: %.1f cycles per MFMA and wave\n", name, threads == 256 ? "1 wave/SIMD " : "2 waves/SIMD", double(h[0]) / (iters * per_iter));
}

int main() {
  float *out; unsigned long long *cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
  for (int threads : {256, 512}) {
    run(chains<1>, out, cyc, threads, 12, "constant operands, 1 accumulator");
    run(chains<2>, out, cyc, threads, 12, "constant operands, 2 accumulators");
    run(chains<4>, out, cyc, threads, 12, "constant operands, 4 accumulators");
    run(role<0>, out, cyc, threads, 48, "role pattern, product-major (the kernel's order)");
    run(role<1>, out, cyc, threads, 48, "role pattern, accumulator-major (6 products back to back)");
    run(role<2>, out, cyc, threads, 48, "role pattern, product-major over row-tile pairs (4 accumulators)");
  }
  return 0;
}
