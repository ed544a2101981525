// Which workgroups share a CU? 512-thread workgroups with ~78 KB of LDS (two per CU, like layer_fused_kernel): every
// workgroup records its XCC / SE / CU ids and the arrival order, then waits until all have arrived so that the whole
// grid is resident at once.   hipcc --offload-arch=gfx950 -O2 tools/dispatch_probe.hip -o /tmp/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ __launch_bounds__(512) void probe(int *rec, int *counter, int nblocks) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int order = atomicAdd(counter, 1);
    rec[blockIdx.x * 3 + 0] = int(hw);
    rec[blockIdx.x * 3 + 1] = int(xcc);
    rec[blockIdx.x * 3 + 2] = order;
    lds[0] = float(order);
    long long t0 = clock64();
    while (atomicAdd(counter, 0) < nblocks && clock64() - t0 < 200000000LL) {}
  }
  __syncthreads();
}

int main() {
  const int nb = 512;
  int *rec, *counter;
  hipMalloc(&rec, nb * 3 * sizeof(int));
  hipMalloc(&counter, sizeof(int));
  hipMemset(counter, 0, sizeof(int));
  hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
  hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 78000, 0, rec, counter, nb);
  hipDeviceSynchronize();
  std::vector<int> h(nb * 3);
  hipMemcpy(h.data(), rec, nb * 3 * sizeof(int), hipMemcpyDeviceToHost);
  std::map<int, std::vector<int>> by_cu;
  for (int b = 0; b < nb; ++b) {
    const unsigned hw = unsigned(h[b * 3]);
    const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7, xcc = h[b * 3 + 1] & 0xf;
    by_cu[((xcc * 8 + se) * 2 + sh) * 16 + cu].push_back(b);
    if (b < 24) printf("block %3d  xcc %d se %d sh %d cu %2d  arrival %3d\n", b, xcc, se, sh, cu, h[b * 3 + 2]);
  }
  printf("%zu distinct CUs\n", by_cu.size());
  int shown = 0;
  std::map<int, int> diff_hist;
  for (auto &kv : by_cu) {
    if (shown++ < 12) { printf("cu key %5d:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    if (kv.second.size() == 2) diff_hist[kv.second[1] - kv.second[0]]++;
    else diff_hist[-int(kv.second.size())]++;
  }
  for (auto &kv : diff_hist) printf("co-resident block id difference %d: %d CUs\n", kv.first, kv.second);
  return 0;
}
