// Micro-benchmark (not product code): cycles per v_mfma_f32_16x16x4_f32 in the multiply role's instruction pattern
// (7 accumulators: 3 column tiles x 2 row tiles + 1 single unit; A differs per row tile and k-step, B per column tile
// and k-step), registers only, for 1 and 2 such waves per SIMD, with and without the per-k-step uniform branches.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool BRANCHY, bool BARRIER>
__global__ __launch_bounds__(512) void pattern(float *out, int kblocks, int nsteps, float seed) {
  f32x4 acc[7];
  for (int t = 0; t < 7; ++t) acc[t] = f32x4{0, 0, 0, 0};
  float a0[4], a1[4];
  float4 w[4];
  for (int i = 0; i < 4; ++i) { a0[i] = seed + threadIdx.x + i; a1[i] = seed * 2 + threadIdx.x + i; }
  for (int t = 0; t < 4; ++t) w[t] = make_float4(seed + t, seed + t + 1, seed + t + 2, seed + t + 3);
  for (int kb = 0; kb < kblocks; ++kb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!BRANCHY || i < nsteps) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const float bv = i == 0 ? w[t].x : i == 1 ? w[t].y : i == 2 ? w[t].z : w[t].w;
          acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], bv, acc[2 * t], 0, 0, 0);
          acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], bv, acc[2 * t + 1], 0, 0, 0);
        }
        const float bv = i == 0 ? w[3].x : i == 1 ? w[3].y : i == 2 ? w[3].z : w[3].w;
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i], bv, acc[6], 0, 0, 0);
      }
    }
    if (BARRIER && (kb % 7) == 6) __syncthreads();
    // keep the operands changing a little so that nothing is hoisted (cheap VALU)
    a0[kb & 3] += 1.0f;
  }
  float s = 0;
  for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
  float *out; hipMalloc(&out, 1024 * 512 * 4);
  const int kblocks = 2100;
  // waves per SIMD = blocks per CU * (threads / 256); 256 CUs
  struct Cfg { int blocks, threads; const char *name; } cfgs[] = {
      {256, 256, "1 wave/SIMD "}, {512, 256, "2 waves/SIMD"}, {256, 512, "2 waves/SIMD (one 512 block)"}, {1024, 256, "4 waves/SIMD"}};
  for (auto c : cfgs) {
    const double nmfma = 28.0 * kblocks;   // per wave
    const double waves_per_simd = double(c.blocks) * (c.threads / 64) / (256.0 * 4);
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL((pattern<false, false>), dim3(c.blocks), dim3(c.threads), 0, 0, out, kblocks, 4, 1.f); });
    printf("%-30s plain     %.3f ms  -> %.1f ns per MFMA per SIMD\n", c.name, ms, ms * 1e6 / (nmfma * waves_per_simd));
    ms = timeit([&] { hipLaunchKernelGGL((pattern<true, false>), dim3(c.blocks), dim3(c.threads), 0, 0, out, kblocks, 4, 1.f); });
    printf("%-30s branches  %.3f ms  -> %.1f ns per MFMA per SIMD\n", c.name, ms, ms * 1e6 / (nmfma * waves_per_simd));
    ms = timeit([&] { hipLaunchKernelGGL((pattern<true, true>), dim3(c.blocks), dim3(c.threads), 0, 0, out, kblocks, 4, 1.f); });
    printf("%-30s +barrier  %.3f ms  -> %.1f ns per MFMA per SIMD\n", c.name, ms, ms * 1e6 / (nmfma * waves_per_simd));
  }
  printf("(32 cycles at 2.4 GHz = 13.3 ns)\n");
  return 0;
}
