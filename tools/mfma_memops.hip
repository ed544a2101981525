// Micro-benchmark (not product code): the multiply role's k-block — 28 x v_mfma_f32_16x16x4_f32 on 7 accumulators —
// with its memory instructions: MODE bit 0 = four global_load_dwordx4 of weight fragments per k-block from an
// L2-resident 240 KB buffer, fetched three k-blocks ahead; bit 1 = four ds_read2_b32 of A fragments per k-block, read
// one k-block ahead. 512-thread blocks whose waves 4-7 leave at once; 256 or 512 blocks (1 or 2 MFMA waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 4) void kblocks(const float4 *__restrict__ wp, float *out, int nkb, int wp_blocks) {
  __shared__ float As[2 * 32 * 102];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2 * 32 * 102; i += 512) As[i] = float(i & 15);
  __syncthreads();
  if (wave >= 4) return;
  const int fr = lane & 15, fq = lane >> 4;
  f32x4 acc[7];
  for (int t = 0; t < 7; ++t) acc[t] = f32x4{0, 0, 0, 0};
  float4 w0[4], w1[4], w2[4];
  auto wload = [&](int g, int t) { return wp[(int64_t(g % wp_blocks) * 13 + wave * 3 + t) * 64 + lane]; };
  for (int t = 0; t < 4; ++t) { w0[t] = wload(0, t); w1[t] = wload(1, t); w2[t] = wload(2, t); }
  float aA[2][4], aB[2][4], aC[2][4];
  const float *arow = As + fr * 102 + fq;
#define ALOAD(dst, kb)                                                           \
  {                                                                              \
    const float *ab_ = arow + ((kb) & 1) * 32 * 102 + ((kb) % 6) * 16;           \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                              \
      dst[0][i] = (MODE & 2) ? ab_[4 * i] : float(i + (kb));                     \
      dst[1][i] = (MODE & 2) ? ab_[16 * 102 + 4 * i] : float(i - (kb));         \
    }                                                                            \
  }
#define STEP(wc, ac, i)                                                                              \
  _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                    \
    const float bv = (i) == 0 ? wc[t].x : (i) == 1 ? wc[t].y : (i) == 2 ? wc[t].z : wc[t].w;         \
    if (t < 3) {                                                                                     \
      acc[2 * t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[0][i], bv, acc[2 * t], 0, 0, 0);          \
      acc[2 * t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[1][i], bv, acc[2 * t + 1], 0, 0, 0);  \
    } else {                                                                                         \
      acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[0][i], bv, acc[6], 0, 0, 0);                  \
    }                                                                                                \
  }
// MODE bit 2: the four weight loads are spread, one behind each MFMA step of the NEXT k-block's predecessor set
// (the set that was consumed one k-block earlier is refilled while this k-block multiplies), instead of clustered
#define KBLOCK(wc, ac, an, kb)                                                                       \
  {                                                                                                  \
    ALOAD(an, (kb) + 1)                                                                              \
    if (MODE & 4) {                                                                                  \
      STEP(wc, ac, 0) if (MODE & 1) wprev[0] = wload((kb) + 2, 0);                                   \
      STEP(wc, ac, 1) if (MODE & 1) wprev[1] = wload((kb) + 2, 1);                                   \
      STEP(wc, ac, 2) if (MODE & 1) wprev[2] = wload((kb) + 2, 2);                                   \
      STEP(wc, ac, 3) if (MODE & 1) wprev[3] = wload((kb) + 2, 3);                                   \
    } else {                                                                                         \
      STEP(wc, ac, 0) STEP(wc, ac, 1) STEP(wc, ac, 2) STEP(wc, ac, 3)                                \
      if (MODE & 1) { _Pragma("unroll") for (int t = 0; t < 4; ++t) wc[t] = wload((kb) + 3, t); }    \
    }                                                                                                \
  }
  ALOAD(aA, 0)
  for (int kb = 0; kb < nkb; kb += 3) {
    { float4 (&wprev)[4] = w2; KBLOCK(w0, aA, aB, kb) }
    { float4 (&wprev)[4] = w0; KBLOCK(w1, aB, aC, kb + 1) }
    { float4 (&wprev)[4] = w1; KBLOCK(w2, aC, aA, kb + 2) }
  }
  float s = 0;
  for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s + w0[0].x + w1[1].y + w2[2].z;
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
  const int wp_blocks = 21;                       // 21 k-blocks x 13 column tiles x 64 lanes x 16 B = 280 KB
  float4 *wp; hipMalloc(&wp, size_t(wp_blocks) * 13 * 64 * sizeof(float4));
  hipMemset(wp, 0, size_t(wp_blocks) * 13 * 64 * sizeof(float4));
  float *out; hipMalloc(&out, 1024 * 256 * 4);
  const int nkb = 2100;
  for (int blocks : {256, 512}) {
    for (int mode : {0, 1, 2, 3, 5, 7}) {
      float ms;
      auto go = [&](auto kern) { return timeit([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, wp, out, nkb, wp_blocks); }); };
      ms = mode == 0 ? go(kblocks<0>) : mode == 1 ? go(kblocks<1>) : mode == 2 ? go(kblocks<2>) : mode == 3 ? go(kblocks<3>)
           : mode == 5 ? go(kblocks<5>) : go(kblocks<7>);
      const double per = ms * 1e6 / (28.0 * nkb * (blocks / 256));
      printf("%d MFMA wave(s)/SIMD  %-34s %.3f ms  -> %.1f ns per MFMA per SIMD\n", blocks / 256,
             mode == 0 ? "registers only" : mode == 1 ? "+ weight loads (L2), clustered" : mode == 2 ? "+ A reads (LDS)"
             : mode == 3 ? "+ both, clustered" : mode == 5 ? "+ weight loads, one per step" : "+ both, loads one per step", ms, per);
    }
  }
  return 0;
}
