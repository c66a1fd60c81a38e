// Timing probe: where does knn_mfma_kernel spend its time?  MODE 0 = full, 1 = Gram tiles only, 2 = gate + pushes.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../multi_part_assembly_amd/csrc knn_time.hip -o knn_time
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "dg_knn.h"

template <int C, int MODE>
float run(const float* x, const float* norm, unsigned short* idx, const int* hdr, int n, int N) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const dim3 g((N + 127) / 128, n);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((dg::knn_mfma_kernel<C, unsigned short, MODE>), g, dim3(256), 0, 0, x, C, norm, N, idx, hdr);
  hipEventRecord(a, 0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((dg::knn_mfma_kernel<C, unsigned short, MODE>), g, dim3(256), 0, 0, x, C, norm, N, idx, hdr);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

template <int C>
void all(int n, int N) {
  std::vector<float> h((size_t)n * N * C);
  srand(1);
  for (auto& v : h) { float u = (float)rand() / RAND_MAX * 2.f - 1.f; v = u > 0 ? u : 0.2f * u; }
  float *x, *norm; unsigned short* idx; int* hdr;
  hipMalloc(&x, h.size() * 4); hipMalloc(&norm, (size_t)n * N * 4); hipMalloc(&idx, (size_t)n * N * 20 * 2); hipMalloc(&hdr, 64);
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  int hh[2] = {n, n * N}; hipMemcpy(hdr, hh, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((dg::rownorm_kernel<C>), dim3((n * N + 255) / 256), dim3(256), 0, 0, x, C, norm, hdr);
  printf("C=%d n=%d N=%d: full %.3f ms | gram only %.3f ms | gram+pushes %.3f ms\n", C, n, N, run<C, 0>(x, norm, idx, hdr, n, N),
         run<C, 1>(x, norm, idx, hdr, n, N), run<C, 2>(x, norm, idx, hdr, n, N));
}

void knn3(int n, int N) {
  std::vector<float> h((size_t)n * N * 4);
  srand(2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (i & 3) == 3 ? 0.f : (float)rand() / RAND_MAX - 0.5f;
  float* x; unsigned short* idx; int* hdr;
  hipMalloc(&x, h.size() * 4); hipMalloc(&idx, (size_t)n * N * 20 * 2); hipMalloc(&hdr, 64);
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  int hh[2] = {n, n * N}; hipMemcpy(hdr, hh, 8, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const dim3 g((N + 255) / 256, n);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((dg::knn3_kernel<unsigned short>), g, dim3(256), 0, 0, x, N, idx, hdr);
  hipEventRecord(a, 0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((dg::knn3_kernel<unsigned short>), g, dim3(256), 0, 0, x, N, idx, hdr);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("C=3 n=%d N=%d: full %.3f ms\n", n, N, ms / 5);
}

int main() {
  knn3(352, 1000);
  all<64>(352, 1000);
  all<128>(352, 1000);
  return 0;
}
