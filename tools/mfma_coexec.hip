// Micro-benchmark (not product code): does VALU work of OTHER waves on a SIMD slow down a wave that issues
// v_mfma_f32_16x16x4_f32 back to back? 512-thread blocks (one per CU): waves 0-3 run the MFMA pattern, waves 4-7 run
// `valu_per_iter` dependent-free v_fma_f32 (or v_exp_f32 / IEEE divisions) per MFMA-wave k-block.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(512) void coexec(float *out, int kblocks, int valu_iters, float seed) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    f32x4 acc[7];
    for (int t = 0; t < 7; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float a0 = seed + threadIdx.x, a1 = seed * 2 + threadIdx.x, b = seed;
    for (int kb = 0; kb < kblocks; ++kb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[2 * t], 0, 0, 0);
          acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[2 * t + 1], 0, 0, 0);
        }
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[6], 0, 0, 0);
      }
    }
    float s = 0;
    for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j + threadIdx.x;
    for (int it = 0; it < valu_iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (KIND == 0) v[j] = v[j] * 1.0001f + 0.5f;
        else if (KIND == 1) v[j] = __expf(v[j]) * 0.5f;
        else v[j] = v[j] / (3.0f + v[(j + 1) & 7]);
      }
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
  float *out; hipMalloc(&out, 1024 * 512 * 4);
  const int kblocks = 2100;   // 58800 MFMAs per wave: 0.85 ms at 14.5 ns
  for (int blocks : {256, 512}) {
    printf("blocks %d (%d MFMA wave(s) + %d VALU wave(s) per SIMD)\n", blocks, blocks / 256, blocks / 256);
    for (int kind = 0; kind < 3; ++kind) {
      for (int vi : {0, 2000, 8000, 16000, 32000}) {
        float ms;
        if (kind == 0) ms = timeit([&] { hipLaunchKernelGGL((coexec<0>), dim3(blocks), dim3(512), 0, 0, out, kblocks, vi, 1.f); });
        else if (kind == 1) ms = timeit([&] { hipLaunchKernelGGL((coexec<1>), dim3(blocks), dim3(512), 0, 0, out, kblocks, vi, 1.f); });
        else ms = timeit([&] { hipLaunchKernelGGL((coexec<2>), dim3(blocks), dim3(512), 0, 0, out, kblocks, vi, 1.f); });
        float ms_v = 0;
        if (kind == 0) ms_v = timeit([&] { hipLaunchKernelGGL((coexec<0>), dim3(blocks), dim3(512), 0, 0, out, 0, vi, 1.f); });
        else if (kind == 1) ms_v = timeit([&] { hipLaunchKernelGGL((coexec<1>), dim3(blocks), dim3(512), 0, 0, out, 0, vi, 1.f); });
        else ms_v = timeit([&] { hipLaunchKernelGGL((coexec<2>), dim3(blocks), dim3(512), 0, 0, out, 0, vi, 1.f); });
        printf("  %s x %5d iters (8 per iter): both %.3f ms   VALU waves alone %.3f ms\n",
               kind == 0 ? "fma " : kind == 1 ? "exp " : "div ", vi, ms, ms_v);
      }
    }
  }
  return 0;
}
