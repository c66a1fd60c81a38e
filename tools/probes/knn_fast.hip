// Probe: the shortlist kNN (dg_knn_fast.h: bf16 bound pass, collect pass, exact rerank) against the exhaustive exact kernel
// (dg_knn.h) — index-for-index comparison, survivor statistics, per-kernel times at the benchmark's part count.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../multi_part_assembly_amd/csrc knn_fast.hip -o knn_fast
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "dg_knn_fast.h"
#ifndef KF_SETS64
#define KF_SETS64 2
#endif

#define TIME(label, reps, ...)                                  \
  do {                                                          \
    hipEvent_t a_, b_;                                          \
    hipEventCreate(&a_); hipEventCreate(&b_);                   \
    __VA_ARGS__;                                                \
    hipEventRecord(a_, 0);                                      \
    for (int w_ = 0; w_ < reps; ++w_) { __VA_ARGS__; }          \
    hipEventRecord(b_, 0);                                      \
    hipEventSynchronize(b_);                                    \
    float ms_; hipEventElapsedTime(&ms_, a_, b_);               \
    printf("  %-28s %.3f ms\n", label, ms_ / reps);             \
  } while (0)

template <int C>
void run(int n, int N, int mode) {
  const size_t R = (size_t)n * N;
  std::vector<float> h(R * C);
  srand(1 + mode);
  for (size_t i = 0; i < h.size(); ++i) {
    float u = (float)rand() / RAND_MAX * 2.f - 1.f;
    if (mode == 1) u = (float)(rand() % 3) * 0.5f;            // lattice: mass ties
    h[i] = u > 0 ? u : 0.2f * u;
  }
  if (mode == 2)  // clustered: points are small perturbations of 8 centres per cloud (close neighbours, large norms)
    for (size_t r = 0; r < R; ++r) {
      const size_t centre = (r / N) * N + (r % 8);
      if (r % N >= 8)
        for (int c = 0; c < C; ++c) h[r * C + c] = h[centre * C + c] * 3.0f + 0.02f * h[r * C + c];
    }
  float *x, *norm, *nl, *nu, *theta; unsigned short *xs, *surv, *idx_ref, *idx_new; unsigned char* scnt; int *hdr, *flags;
  const int Q128 = (N + 127) / 128;
  hipMalloc(&x, R * C * 4); hipMalloc(&norm, R * 4); hipMalloc(&nl, R * 4); hipMalloc(&nu, R * 4); hipMalloc(&theta, R * 4);
  hipMalloc(&xs, R * 2 * C * 2); hipMalloc(&surv, R * 2 * dg::kKfCap * 2); hipMalloc(&scnt, R * 2);
  hipMalloc(&idx_ref, R * 20 * 2); hipMalloc(&idx_new, R * 20 * 2); hipMalloc(&hdr, 64); hipMalloc(&flags, (size_t)(n + 8) * Q128 * 4);
  hipMemcpy(x, h.data(), R * C * 4, hipMemcpyHostToDevice);
  int hh[2] = {n, (int)R}; hipMemcpy(hdr, hh, 8, hipMemcpyHostToDevice);
  hipMemset(idx_new, 0xff, R * 20 * 2);
  const dim3 gold(Q128, DG_KNN_GRID_Y(n)), gnew((N + dg::kKfQB - 1) / dg::kKfQB, DG_KNN_GRID_Y(n)), grr((N + dg::kRrQ - 1) / dg::kRrQ, DG_KNN_GRID_Y(n));
  constexpr int S = C > 64 ? 1 : KF_SETS64, W = 8 / S;
  printf("C=%d n=%d N=%d mode=%d\n", C, n, N, mode);
  TIME("rownorm", 5, hipLaunchKernelGGL((dg::rownorm_kernel<C>), dim3((R + 255) / 256), dim3(256), 0, 0, x, C, norm, hdr));
  TIME("exhaustive exact (old)", 3, hipLaunchKernelGGL((dg::knn_mfma_kernel<C, unsigned short>), gold, dim3(256), 0, 0, x, C, norm, N, idx_ref, hdr));
  TIME("split", 5, hipLaunchKernelGGL((dg::knn_split_kernel<C>), dim3((R * (C / 4) + 255) / 256), dim3(256), 0, 0, x, C, norm, xs, nl, nu, hdr));
  TIME("bound", 5, hipLaunchKernelGGL((dg::knn_gram_kernel<C, false, S, W>), gnew, dim3(64 * W), 0, 0, xs, nl, nl, nu, N, theta, surv, scnt, hdr));
  if (mode == 0 && n > 100) {
    TIME("bound, no epilogue", 5, hipLaunchKernelGGL((dg::knn_gram_kernel<C, false, S, W, 1>), gnew, dim3(64 * W), 0, 0, xs, nl, nl, nu, N, theta, surv, scnt, hdr));
    TIME("bound, no staging", 5, hipLaunchKernelGGL((dg::knn_gram_kernel<C, false, S, W, 2>), gnew, dim3(64 * W), 0, 0, xs, nl, nl, nu, N, theta, surv, scnt, hdr));
    TIME("bound, MFMA only", 5, hipLaunchKernelGGL((dg::knn_gram_kernel<C, false, S, W, 3>), gnew, dim3(64 * W), 0, 0, xs, nl, nl, nu, N, theta, surv, scnt, hdr));
    TIME("bound", 5, hipLaunchKernelGGL((dg::knn_gram_kernel<C, false, S, W>), gnew, dim3(64 * W), 0, 0, xs, nl, nl, nu, N, theta, surv, scnt, hdr));
  }
  TIME("collect", 5, hipLaunchKernelGGL((dg::knn_gram_kernel<C, true, S, W>), gnew, dim3(64 * W), 0, 0, xs, nu, nl, nu, N, theta, surv, scnt, hdr));
  TIME("rerank", 5, hipLaunchKernelGGL((dg::knn_rerank_kernel<C, unsigned short>), grr, dim3(256), 0, 0, x, C, norm, N, surv, scnt, idx_new, hdr));
  std::vector<unsigned short> a(R * 20), b(R * 20);
  std::vector<unsigned char> cnt(R * 2);
  std::vector<int> fl((size_t)n * Q128);
  hipMemcpy(a.data(), idx_ref, R * 40, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), idx_new, R * 40, hipMemcpyDeviceToHost);
  hipMemcpy(cnt.data(), scnt, R * 2, hipMemcpyDeviceToHost);
  hipMemcpy(fl.data(), flags, fl.size() * 4, hipMemcpyDeviceToHost);
  size_t bad_rows = 0, over = 0, total = 0; int mx = 0;
  for (size_t r = 0; r < R; ++r) {
    if (cnt[2 * r] == dg::kKfOverflow || cnt[2 * r + 1] == dg::kKfOverflow) ++over;
    else { const int c = cnt[2 * r] + cnt[2 * r + 1]; total += c; if (c > mx) mx = c; }
    if (memcmp(&a[r * 20], &b[r * 20], 40) != 0) ++bad_rows;
  }
  printf("  survivors per query: mean %.2f max %d | queries scanned exhaustively (list overflow): %zu | MISMATCHED rows: %zu of %zu\n",
         (double)total / (R - over ? R - over : 1), mx, over, bad_rows, R);
  hipFree(x); hipFree(norm); hipFree(nl); hipFree(nu); hipFree(theta); hipFree(xs); hipFree(surv); hipFree(scnt);
  hipFree(idx_ref); hipFree(idx_new); hipFree(hdr); hipFree(flags);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 352, N = argc > 2 ? atoi(argv[2]) : 1000;
  for (int mode = 0; mode < (argc > 3 ? 1 : 3); ++mode) {
    run<64>(mode == 0 ? n : 16, mode == 0 ? N : 300, mode);
    run<128>(mode == 0 ? n : 16, mode == 0 ? N : 300, mode);
  }
  if (argc <= 3) { run<64>(5, 20, 0); run<128>(3, 97, 0); run<64>(2, 1024, 0); }
  return 0;
}
