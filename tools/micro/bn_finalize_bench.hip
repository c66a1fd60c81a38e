// Micro-benchmark of the small PointNet reduction kernels in isolation (dev tool, not part of the library).
#include "../../multi_part_assembly_amd/csrc/pointnet.hip"
#include <cstdio>
#include <vector>

namespace {
__global__ __launch_bounds__(1024) void v_empty(float* out) {
  if (threadIdx.x == 0 && out == nullptr) out[0] = 1.0f;
}
template <int MODE>
__global__ __launch_bounds__(1024) void v_loop(const float* __restrict__ partial, const float* __restrict__ valids,
                                               int M, int splits, int C, float* __restrict__ out) {
  __shared__ float sm[16][64];
  const int cl = threadIdx.x & 63, slice = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float a = 0.0f;
  double ad = 0.0;
  const int total = M * splits;
  for (int e = slice; e < total; e += 16) {
    if (MODE >= 1 && valids[MODE >= 2 ? e / splits : e >> 1] == 0.0f) continue;
    const float v = partial[((long long)e * C + c) * 2];
    if (MODE >= 3) ad += (double)v;
    else a += v;
  }
  sm[slice][cl] = a + (float)ad;
  __syncthreads();
  if (slice == 0) {
    float t = 0.0f;
    for (int k = 0; k < 16; ++k) t += sm[k][cl];
    out[c] = t;
  }
}
// reduce_partials alone (batched loads), then variants of the tail
template <int TAIL>
__global__ __launch_bounds__(1024) void v_reduce(const float* __restrict__ partial, const float* __restrict__ valids,
                                                 int M, int splits, int C, const float* __restrict__ count,
                                                 float* __restrict__ out) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, ss;
  reduce_partials(partial, valids, M, splits, C, c, s, ss);
  if (threadIdx.x >= 64) return;
  if (TAIL == 0) {
    out[c] = (float)(s + ss);
  } else {
    const double n = (double)count[0];
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0.0) var = 0.0;
    out[c] = TAIL == 1 ? (float)var : (float)(1.0 / __builtin_sqrt(var + 1e-5));
  }
}
}  // namespace

template <typename F>
void timeit(const char* name, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 1) printf("%-28s %.2f us per launch\n", name, ms * 1000 / 50);
  }
}

int main() {
  const int M = 640, splits = 2, C = 64, N = 1000;
  float *partial, *valids, *count, *gamma, *beta, *rm, *rv, *bn;
  hipMalloc(&partial, sizeof(float) * M * splits * 256 * 2);
  hipMalloc(&valids, sizeof(float) * M);
  hipMalloc(&count, 4);
  hipMalloc(&gamma, 1024);
  hipMalloc(&beta, 1024);
  hipMalloc(&rm, 1024);
  hipMalloc(&rv, 1024);
  hipMalloc(&bn, 4096);
  std::vector<float> v(M, 1.0f), p(M * splits * 256 * 2, 0.5f), g(256, 1.0f);
  hipMemcpy(valids, v.data(), sizeof(float) * M, hipMemcpyHostToDevice);
  hipMemcpy(partial, p.data(), sizeof(float) * p.size(), hipMemcpyHostToDevice);
  hipMemcpy(gamma, g.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(beta, g.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(rm, g.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(rv, g.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(pn_count_kernel, dim3(1), dim3(64), 0, 0, valids, M, N, count);
  timeit("bn_finalize", [&] {
    hipLaunchKernelGGL(pn_bn_finalize_kernel, dim3(C / 64), dim3(64 * kSlices), 0, 0, partial, valids, M, splits, C, count,
                       gamma, beta, rm, rv, 0.1f, 1e-5f, bn);
  });
  timeit("empty 1024 thr", [&] { hipLaunchKernelGGL(v_empty, dim3(1), dim3(1024), 0, 0, bn); });
  timeit("loop fp32 no valids", [&] { hipLaunchKernelGGL((v_loop<0>), dim3(1), dim3(1024), 0, 0, partial, valids, M, splits, C, bn); });
  timeit("loop fp32 valids e>>1", [&] { hipLaunchKernelGGL((v_loop<1>), dim3(1), dim3(1024), 0, 0, partial, valids, M, splits, C, bn); });
  timeit("loop fp32 valids e/splits", [&] { hipLaunchKernelGGL((v_loop<2>), dim3(1), dim3(1024), 0, 0, partial, valids, M, splits, C, bn); });
  timeit("loop fp64 valids e/splits", [&] { hipLaunchKernelGGL((v_loop<3>), dim3(1), dim3(1024), 0, 0, partial, valids, M, splits, C, bn); });
  timeit("reduce_partials only", [&] { hipLaunchKernelGGL((v_reduce<0>), dim3(1), dim3(1024), 0, 0, partial, valids, M, splits, C, count, bn); });
  timeit("reduce + fp64 div", [&] { hipLaunchKernelGGL((v_reduce<1>), dim3(1), dim3(1024), 0, 0, partial, valids, M, splits, C, count, bn); });
  timeit("reduce + fp64 div + rsqrt", [&] { hipLaunchKernelGGL((v_reduce<2>), dim3(1), dim3(1024), 0, 0, partial, valids, M, splits, C, count, bn); });
  return 0;
}
