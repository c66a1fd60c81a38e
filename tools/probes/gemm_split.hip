// Probe: the split-bf16 GEMM kernels (csrc/dg_gemm_split.h) at the DGCNN encoder's shapes — time per launch, TFLOP/s,
// and the largest deviation from the exact-fp32 MFMA kernels (csrc/dg_gemm.h) on the same random operands.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off gemm_split.hip -o gemm_split [-DDG_TN_QUAD=0 ...]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

#include "../../multi_part_assembly_amd/csrc/dg_gemm_split.h"

static float* dev_random(size_t n, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((s >> 8) * (1.0f / 8388608.0f) - 1.0f) * ((s & 7u) == 0 ? 4.0f : 0.5f);
  }
  float* d;
  hipMalloc(&d, n * sizeof(float));
  hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}

template <typename F>
static float time_ms(F launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

static double max_rel(const float* d_a, const float* d_b, size_t n) {
  std::vector<float> a(n), b(n);
  hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (size_t i = 0; i < n; ++i) {
    num = fmax(num, fabs((double)a[i] - b[i]));
    den = fmax(den, fabs((double)b[i]));
  }
  return num / den;
}

// dW [Nout][K] = Y^T X over R rows (Rmax rows allocated), as gemm_tn of dgcnn_enc.hip launches it
template <int BK>
static void run_tn(int R, int Rmax, int Nout, int K, int ldx, int chunks) {
  float* Y = dev_random((size_t)Rmax * Nout, 1);
  float* X = dev_random((size_t)Rmax * ldx, 2);
  float *part, *o1, *o2;
  hipMalloc(&part, (size_t)chunks * Nout * K * 4);
  hipMalloc(&o1, (size_t)Nout * K * 4);
  hipMalloc(&o2, (size_t)Nout * K * 4);
  int hdr_h[2] = {R / 1000, R}, *hdr;
  hipMalloc(&hdr, 8);
  hipMemcpy(hdr, hdr_h, 8, hipMemcpyHostToDevice);
  const dim3 grid((Nout + 127) / 128, K / BK, chunks);
  const int rpc = ((R + chunks - 1) / chunks + 31) / 32 * 32;  // what the kernels derive from hdr when passed 0
  auto split = [&] {
    hipLaunchKernelGGL(dg::gemm_tn_split_kernel<BK>, grid, dim3(dg::kGsT), 0, 0, (const float*)Y, Nout, Nout, (const float*)X,
                       ldx, K, part, rpc, (const int*)hdr, 0);
  };
  auto exact = [&] {
    hipLaunchKernelGGL(dg::gemm_tn_kernel<BK>, grid, dim3(dg::kGT), 0, 0, (const float*)Y, Nout, Nout, (const float*)X, ldx,
                       K, part, rpc, (const int*)hdr, 0);
  };
  const float t_split = time_ms(split, 10);
  dg::launch_tn_reduce(part, chunks, (long long)Nout * K, o1, 0);
  const float t_exact = time_ms(exact, 5);
  dg::launch_tn_reduce(part, chunks, (long long)Nout * K, o2, 0);
  hipDeviceSynchronize();
  const double fl = 2.0 * R * Nout * K;
  printf("tn<%d>  R=%d Nout=%d K=%d chunks=%d: split %.1f us (%.0f TFLOP/s), fp32 mfma %.1f us (%.0f), max dev %.2e of max\n", BK,
         R, Nout, K, chunks, t_split * 1e3, fl / t_split / 1e9, t_exact * 1e3, fl / t_exact / 1e9, max_rel(o1, o2, (size_t)Nout * K));
  hipFree(Y), hipFree(X), hipFree(part), hipFree(o1), hipFree(o2), hipFree(hdr);
}

// C [R][Nout] = A [R][K] W^T
template <int BN>
static void run_nt(int R, int Rmax, int K, int Nout) {
  float* A = dev_random((size_t)Rmax * K, 3);
  float* W = dev_random((size_t)Nout * K, 4);
  float *c1, *c2;
  hipMalloc(&c1, (size_t)Rmax * Nout * 4);
  hipMalloc(&c2, (size_t)Rmax * Nout * 4);
  int hdr_h[2] = {R / 1000, R}, *hdr;
  hipMalloc(&hdr, 8);
  hipMemcpy(hdr, hdr_h, 8, hipMemcpyHostToDevice);
  const unsigned gx = DG_GEMM_GRID_X(Rmax);
  auto split = [&] {
    hipLaunchKernelGGL((dg::gemm_nt_split_kernel<BN, false>), dim3(gx, Nout / BN), dim3(dg::kGsT), 0, 0, (const float*)A, K,
                       (const float*)W, K, c1, Nout, (const int*)hdr, dg::GsEpi{}, 0);
  };
  auto exact = [&] {
    hipLaunchKernelGGL((dg::gemm_nt_kernel<BN, false>), dim3(gx, Nout / BN), dim3(dg::kGT), 0, 0, (const float*)A, K,
                       (const float*)W, K, c2, Nout, (const int*)hdr);
  };
  const float t_split = time_ms(split, 10), t_exact = time_ms(exact, 5);
  const double fl = 2.0 * R * Nout * K, bytes = 4.0 * R * (K + Nout);
  printf("nt<%d>  R=%d K=%d Nout=%d: split %.1f us (%.0f TFLOP/s, %.2f TB/s), fp32 mfma %.1f us (%.0f), max dev %.2e of max\n", BN,
         R, K, Nout, t_split * 1e3, fl / t_split / 1e9, bytes / t_split / 1e9, t_exact * 1e3, fl / t_exact / 1e9,
         max_rel(c1, c2, (size_t)R * Nout));
  hipFree(A), hipFree(W), hipFree(c1), hipFree(c2), hipFree(hdr);
}

int main() {
  const int R = 353000, Rmax = 640000;
  run_tn<128>(R, Rmax, 128, 512, 512, 128);   // tail: dW5 [F = 128][512]
  run_tn<128>(R, Rmax, 512, 128, 512, 128);   // stage 4: d[Wu | Wv] [512][128]
  run_tn<64>(R, Rmax, 256, 64, 512, 128);     // stage 3
  run_tn<64>(R, Rmax, 128, 64, 512, 256);     // stage 2
  run_tn<128>(12800, 12800, 512, 512, 512, 32);  // a pair-MLP layer of the graph networks
  run_nt<128>(R, Rmax, 64, 128);
  run_nt<128>(R, Rmax, 64, 256);
  run_nt<128>(R, Rmax, 128, 512);
  run_nt<128>(R, Rmax, 512, 128);
  run_nt<128>(12800, 12800, 512, 512);
  return 0;
}
