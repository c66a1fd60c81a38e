// Probe: issue rate of v_mfma_f32_32x32x16_bf16 — NCH independent accumulator chains per wave, W waves per block.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) { a[t] = (__bf16)(float)(threadIdx.x + t); b[t] = (__bf16)(float)(threadIdx.x * 3 + t); }
  f32x16 acc[NCH];
  for (int c = 0; c < NCH; ++c) acc[c] = f32x16{0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < NCH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NCH>
void run(int blocks) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double mf = (double)blocks * 4 * iters * 8 * NCH;
  printf("chains=%d blocks=%d: %.3f ms, %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz (waves per SIMD %d)\n", NCH, blocks, ms,
         mf * 32768 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / mf, (blocks + 255) / 256);
  hipFree(out);
}
int main() {
  run<1>(256); run<2>(256); run<4>(256); run<1>(512); run<2>(512); run<1>(1024); run<2>(1024);
  return 0;
}
