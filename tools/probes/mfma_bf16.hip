// Probe: operand layout of v_mfma_f32_32x32x16_bf16 on gfx950.  Hypothesis: lane l supplies A[i = l & 31][k = 8 (l >> 5) .. +7]
// and B[j = l & 31][same k]; C[i][j] lands in lane (j, h), register r with i = (r & 3) + 8 (r >> 2) + 4 h.
// hipcc --offload-arch=gfx950 -O3 mfma_bf16.hip -o mfma_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k(const float* A, const float* B, float* C) {  // A [32][16], B [32][16] (C = A . B^T), C [32][32]
  const int l = threadIdx.x, j = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int t = 0; t < 8; ++t) {
    a[t] = (__bf16)A[j * 16 + 8 * h + t];
    b[t] = (__bf16)B[j * 16 + 8 * h + t];
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

int main() {
  float hA[512], hB[512], hC[1024], *A, *B, *C;
  srand(3);
  for (int i = 0; i < 512; ++i) { hA[i] = (float)(rand() % 17 - 8); hB[i] = (float)(rand() % 13 - 6); }
  hipMalloc(&A, 2048); hipMalloc(&B, 2048); hipMalloc(&C, 4096);
  hipMemcpy(A, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(B, hB, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, C);
  hipMemcpy(hC, C, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float s = 0;
      for (int t = 0; t < 16; ++t) s += hA[i * 16 + t] * hB[j * 16 + t];
      bad += s != hC[i * 32 + j];
    }
  printf("mismatches: %d of 1024\n", bad);
  return bad != 0;
}
