// Probe: issue rate of plain and packed fp32 VALU instructions (wave64), W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a[8];
  f2 p[8];
  for (int t = 0; t < 8; ++t) { a[t] = seed + threadIdx.x + t; p[t] = f2{a[t], a[t] + 1.0f}; }
  const float m = seed * 0.5f;
  const f2 pm = f2{m, m};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) {  // v_mul_f32 + v_add_f32 (two instructions per element)
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = a[t] * m + 1.0f;
      } else if (MODE == 1) {  // v_pk_mul_f32 + v_pk_add_f32 (two instructions per two elements)
#pragma unroll
        for (int t = 0; t < 8; ++t) p[t] = p[t] * pm + f2{1.0f, 1.0f};
      } else {  // v_fma_f32
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = __builtin_fmaf(a[t], m, 1.0f);
      }
    }
  }
  float s = 0;
  for (int t = 0; t < 8; ++t) s += a[t] + p[t][0] + p[t][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(int blocks, const char* name, int instr_per_iter) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  const int iters = 4000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double winstr = (double)blocks * 4 * iters * instr_per_iter;  // wave-instructions
  printf("%-28s blocks=%4d: %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, blocks, ms,
         ms * 1e-3 * 2.4e9 * 1024 / winstr);
  hipFree(out);
}
int main() {
  for (int blocks : {256, 512, 1024}) {
    run<0>(blocks, "v_mul_f32 + v_add_f32", 8 * 8 * 2);
    run<1>(blocks, "v_pk_mul_f32 + v_pk_add_f32", 8 * 8 * 2);
    run<2>(blocks, "v_fma_f32", 8 * 8);
  }
  return 0;
}
