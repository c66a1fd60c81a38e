// Probe: how many single-wave workgroups (64 threads) does a CU hold at once?  Every block bumps a counter of the CU it
// runs on (HW_ID register), records the largest value it saw, holds its slot for ~20 us (bounded), leaves.
// Also: latency of a dependent chain of global loads (pointer chase through an L2-resident and an HBM-sized table) with
// 1 wave per CU and with the chip full — what a latency-bound search kernel pays per round trip.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 wg_residency.hip -o wg_residency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int VG>
__global__ __launch_bounds__(64) void census(int* cur, int* peak, int lds_bytes_tag) {
  extern __shared__ int dyn[];
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int cu = (int)(((xcc & 0xf) << 8) | (((hw >> 13) & 0x7) << 4) | ((hw >> 8) & 0xf));  // (xcc, se, cu)
  float keep[VG];
  for (int i = 0; i < VG; ++i) keep[i] = (float)(threadIdx.x + i);
  if (threadIdx.x == 0) {
    const int v = atomicAdd(&cur[cu], 1) + 1;
    atomicMax(&peak[cu], v);
    dyn[0] = v;
  }
  const unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 40000ull) {
    for (int i = 0; i < VG; ++i) asm volatile("v_add_f32 %0, %0, %0" : "+v"(keep[i]));
    __builtin_amdgcn_s_sleep(8);
  }
  float s = 0;
  for (int i = 0; i < VG; ++i) s += keep[i];
  if (s == 12345.678f) dyn[1] = 1;
  if (threadIdx.x == 0) atomicSub(&cur[cu], 1);
}

__global__ __launch_bounds__(64) void chase(const int* __restrict__ tab, int steps, int stride, unsigned long long* out, int* sink) {
  int p = (int)((blockIdx.x * 64 + threadIdx.x) * 97) % stride;
  p = tab[p];
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < steps; ++i) p = tab[p];
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) atomicAdd(out, t1 - t0);
  if (p == -1) *sink = 1;
}

int main() {
  int *cur, *peak;
  hipMalloc(&cur, 4096 * 4);
  hipMalloc(&peak, 4096 * 4);
  auto run = [&](auto kern, int lds, const char* what) {
    hipMemset(cur, 0, 4096 * 4);
    hipMemset(peak, 0, 4096 * 4);
    hipLaunchKernelGGL(kern, dim3(256 * 64), dim3(64), lds, 0, cur, peak, lds);
    hipDeviceSynchronize();
    std::vector<int> h(4096);
    hipMemcpy(h.data(), peak, 4096 * 4, hipMemcpyDeviceToHost);
    int mx = 0, cus = 0;
    long long sum = 0;
    for (int v : h) {
      if (v > 0) ++cus, sum += v;
      if (v > mx) mx = v;
    }
    printf("%s: %d CUs seen, peak resident single-wave workgroups per CU: max %d, mean %.1f\n", what, cus, mx, cus ? (double)sum / cus : 0.0);
  };
  run(census<8>, 16, "  ~16 VGPRs, 16 B LDS");
  run(census<8>, 2560, "  ~16 VGPRs, 2.5 KB LDS");
  run(census<56>, 2560, "  ~64 VGPRs, 2.5 KB LDS");
  run(census<72>, 2560, "  ~80 VGPRs, 2.5 KB LDS");
  run(census<88>, 2560, "  ~96 VGPRs, 2.5 KB LDS");
  // dependent-load latency
  for (long long n : {1LL << 18, 1LL << 28}) {  // 1 MB (L2) and 1 GB (HBM) tables of ints
    std::vector<int> h(n);
    unsigned long long x = 88172645463325252ull;
    for (long long i = 0; i < n; ++i) {
      x ^= x << 13, x ^= x >> 7, x ^= x << 17;
      h[i] = (int)(x % (unsigned long long)n);
    }
    int* tab;
    hipMalloc(&tab, n * 4);
    hipMemcpy(tab, h.data(), n * 4, hipMemcpyHostToDevice);
    unsigned long long* out;
    int* sink;
    hipMalloc(&out, 8);
    hipMalloc(&sink, 4);
    for (int blocks : {256, 256 * 8, 256 * 24, 256 * 32}) {
      hipMemset(out, 0, 8);
      const int steps = 200;
      hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, tab, steps, (int)n, out, sink);
      hipDeviceSynchronize();
      unsigned long long t;
      hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
      printf("  table %4lld MB, %5d waves (%2d per CU): %.0f ticks per dependent random 4-byte load (64 distinct lines per wave)\n", n * 4 >> 20, blocks,
             blocks / 256, (double)t / blocks / steps);
    }
    hipFree(tab);
  }
  return 0;
}
