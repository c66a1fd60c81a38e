// Probe of ds_read_b64_tr_b16 (gfx950): what does lane l receive when every lane supplies the address of 4 contiguous b16?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read.hip -o /tmp/tr_read && /tmp/tr_read
// X[r][c] = 64 r + c in a [64][64] u16 image; lane l = 16 g + i supplies &X[4 g + (i >> 2)][4 (i & 3)].
// Expected (guide T10): lane l receives X[4 g + e][i], e = 0..3 — column i of the group's 4 x 16 block.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  const unsigned short* p = lds + (4 * g + (i >> 2)) * 64 + 4 * (i & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d;
  unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) {
      const int r = h[l * 4 + e] / 64, c = h[l * 4 + e] % 64;
      printf(" (%2d,%2d)", r, c);
      bad += !(r == 4 * (l >> 4) + e && c == (l & 15));
    }
    printf("\n");
  }
  printf("%s\n", bad ? "MISMATCH with the expected layout" : "OK: lane l = 16 g + i receives X[4 g + e][i]");
  return 0;
}
