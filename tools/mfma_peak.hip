// Micro-benchmark (not product code): achievable v_mfma_f32_16x16x4_f32 / 32x32x2 rate on this GPU, registers only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0, 0, 0, 0};
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a0, float b0) {
  f32x16 acc[2];
  for (int t = 0; t < 2; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0;
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0;
  for (int t = 0; t < 2; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  float *out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 4000;
  for (int blocks : {256, 512, 1024, 2048}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    double fl = 2.0 * 16 * 16 * 4 * 8.0 * iters * blocks * 4;
    printf("16x16x4  acc=8 blocks=%4d  %.3f ms  %.1f TF/s\n", blocks, ms, fl / ms / 1e9);
  }
  for (int blocks : {1024}) {
    float ms = timeit([&] { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("16x16x4  acc=2 blocks=%4d  %.3f ms  %.1f TF/s\n", blocks, ms, 2.0 * 16 * 16 * 4 * 2.0 * iters * blocks * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("16x16x4  acc=1 blocks=%4d  %.3f ms  %.1f TF/s\n", blocks, ms, 2.0 * 16 * 16 * 4 * 1.0 * iters * blocks * 4 / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("32x32x2  acc=2 blocks=%4d  %.3f ms  %.1f TF/s\n", blocks, ms, 2.0 * 32 * 32 * 2 * 2.0 * iters * blocks * 4 / ms / 1e9);
  }
  return 0;
}
