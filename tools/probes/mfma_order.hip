// Probe: how does v_mfma_f32_32x32x2_f32 round its K-sum?  Compares the matrix-core result of a K-long chain with
// host models of the accumulation, bit for bit.  Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma_order.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K>
__global__ void gram(const float* a, const float* b, float* d) {  // a [32][K], b [32][K] (row n of b = column n of B)
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  f32x16 acc = {0};
  for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j * K + k + h], b[j * K + k + h], acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

int main() {
  const int K = 64;
  float *a = (float*)malloc(32 * K * 4), *b = (float*)malloc(32 * K * 4), *d = (float*)malloc(32 * 32 * 4);
  srand(1);
  for (int i = 0; i < 32 * K; ++i) {
    a[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    b[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  }
  float *da, *db, *dd;
  hipMalloc(&da, 32 * K * 4); hipMalloc(&db, 32 * K * 4); hipMalloc(&dd, 32 * 32 * 4);
  hipMemcpy(da, a, 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(db, b, 32 * K * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(gram<K>, dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(d, dd, 32 * 32 * 4, hipMemcpyDeviceToHost);
  int m_seq = 0, m_pair = 0, m_pairfma = 0, m_mulsum = 0, m_rev = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f;
      for (int k = 0; k < K; ++k) s1 = fmaf(a[m * K + k], b[n * K + k], s1);                     // sequential fma chain
      for (int k = 0; k < K; k += 2) s2 = s2 + (a[m * K + k] * b[n * K + k] + a[m * K + k + 1] * b[n * K + k + 1]);
      for (int k = 0; k < K; k += 2) s3 = s3 + fmaf(a[m * K + k + 1], b[n * K + k + 1], a[m * K + k] * b[n * K + k]);
      for (int k = 0; k < K; ++k) s4 = s4 + a[m * K + k] * b[n * K + k];                         // mul, add separately
      for (int k = 0; k < K; k += 2) { s5 = fmaf(a[m * K + k + 1], b[n * K + k + 1], s5); s5 = fmaf(a[m * K + k], b[n * K + k], s5); }
      const float g = d[m * 32 + n];
      m_seq += memcmp(&g, &s1, 4) == 0; m_pair += memcmp(&g, &s2, 4) == 0; m_pairfma += memcmp(&g, &s3, 4) == 0;
      m_mulsum += memcmp(&g, &s4, 4) == 0; m_rev += memcmp(&g, &s5, 4) == 0;
    }
  printf("K=%d of 1024: seq_fma %d  pair(mul+mul)+acc %d  pair(fma)+acc %d  mul_then_add %d  fma_k1_then_k0 %d\n", K, m_seq,
         m_pair, m_pairfma, m_mulsum, m_rev);
  return 0;
}
