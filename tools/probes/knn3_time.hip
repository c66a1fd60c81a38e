// Timing probe of the C = 3 kNN kernel (dg_knn.h: knn3_kernel) at the benchmark's 353 x 1000 points, for A/B builds of
// its knobs:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../multi_part_assembly_amd/csrc
//             [-DDG_T3=128 -DDG_QN3=24 -DDG_CPC3=8] knn3_time.hip -o knn3_time
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "dg_knn.h"

int main() {
  const int n = 353, N = 1000;
  std::vector<float> h((size_t)n * N * 4);
  srand(2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (i & 3) == 3 ? 0.f : (float)rand() / RAND_MAX - 0.5f;
  float* x; unsigned short* idx; int* hdr;
  hipMalloc(&x, h.size() * 4); hipMalloc(&idx, (size_t)n * N * 20 * 2); hipMalloc(&hdr, 64);
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  int hh[2] = {n, n * N}; hipMemcpy(hdr, hh, 8, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const dim3 g((N + DG_T3 - 1) / DG_T3, DG_KNN_GRID_Y(n));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((dg::knn3_kernel<unsigned short>), g, dim3(DG_T3), 0, 0, x, N, idx, hdr);
  hipEventRecord(a, 0);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((dg::knn3_kernel<unsigned short>), g, dim3(DG_T3), 0, 0, x, N, idx, hdr);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  unsigned long long sum = 0;
  std::vector<unsigned short> out((size_t)n * N * 20);
  hipMemcpy(out.data(), idx, out.size() * 2, hipMemcpyDeviceToHost);
  for (auto v : out) sum = sum * 1315423911ull + v;
  printf("T3=%d QN3=%d CPC3=%d: %.3f ms  (checksum %llx)\n", DG_T3, DG_QN3, DG_CPC3, ms / 10, sum);
  return 0;
}
