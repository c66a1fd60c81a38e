// Helpers shared by the matrix-core gated searches (gate_nn.hip: nearest neighbour; dg_knn3_gate.h: k = 20): bf16 pieces of
// fp32 values, the operand rows of the K = 16 product, v_min3 trees over a 32 x 32 accumulator, upward rounding.
#pragma once

#include <hip/hip_runtime.h>

namespace mpa {
namespace gate {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// the next float above x (+inf stays +inf; NaN stays NaN)
__device__ __forceinline__ float next_up(float x) {
  const float up = x >= 0.0f ? __uint_as_float(__float_as_uint(x + 0.0f) + 1u) : __uint_as_float(__float_as_uint(x) - 1u);
  return x < __builtin_inff() ? up : x;
}
__device__ __forceinline__ unsigned bf_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf_round(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ unsigned bf_pack(float lo, float hi) { return bf_bits(lo) | (bf_bits(hi) << 16); }
__device__ __forceinline__ bf16x8 as_bf16x8(const uint4 v) { return __builtin_bit_cast(bf16x8, v); }
// three-operand minima only: the two-operand v_min_f32 makes the compiler canonicalise every accumulator register first
__device__ __forceinline__ float min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
// min of the 16 values of a lane's accumulator = min(x, y): 7 x v_min3
__device__ __forceinline__ void min16(const f32x16& a, float& x, float& y) {
  const float m0 = min3(a[0], a[1], a[2]), m1 = min3(a[3], a[4], a[5]), m2 = min3(a[6], a[7], a[8]);
  const float m3 = min3(a[9], a[10], a[11]), m4 = min3(a[12], a[13], a[14]);
  x = min3(m0, m1, m2);
  y = min3(m3, m4, a[15]);
}
// x = p0 + p1 + p2 exactly (three bf16 pieces of a finite fp32 number)
__device__ __forceinline__ void split3(float x, float& p0, float& p1, float& p2) {
  p0 = bf_round(x);
  const float r1 = x - p0;
  p1 = bf_round(r1);
  p2 = r1 - p1;
}
// A point y (centred coordinates) with squared norm n as a ROW of the product (the two 16-byte k-halves):
//   hx hy hz | hx hy hz | lx ly || lz | n0 n1 n2 | 0 0 0 0        h = bf16(y), l = bf16(y - h), n = n0 + n1 + n2 exactly
__device__ __forceinline__ void target_row(float yx, float yy, float yz, float n, uint4& k0, uint4& k1) {
  const float hx = bf_round(yx), hy = bf_round(yy), hz = bf_round(yz);
  const float lx = yx - hx, ly = yy - hy, lz = yz - hz;
  float n0, n1, n2;
  split3(n, n0, n1, n2);
  k0 = uint4{bf_pack(hx, hy), bf_pack(hz, hx), bf_pack(hy, hz), bf_pack(lx, ly)};
  k1 = uint4{bf_bits(lz) | (bf_bits(n0) << 16), bf_pack(n1, n2), 0u, 0u};
}
// ... and as a COLUMN:  -2h | -2l | -2h || . | 1 1 1 | 0 0 0 0   so that row . column = n_row - 2 (h.h + l.h + h.l)
__device__ __forceinline__ void query_column(float yx, float yy, float yz, uint4& k0, uint4& k1) {
  const float hx = bf_round(yx), hy = bf_round(yy), hz = bf_round(yz);
  const float lx = yx - hx, ly = yy - hy, lz = yz - hz;
  const float a = -2.0f;
  const unsigned one = 0x3f80u;
  k0 = uint4{bf_pack(a * hx, a * hy), bf_pack(a * hz, a * lx), bf_pack(a * ly, a * lz), bf_pack(a * hx, a * hy)};
  k1 = uint4{bf_bits(a * hz) | (one << 16), one | (one << 16), 0u, 0u};
}
// the two query tiles of a wave (tile s = queries 32 s + j): lane (j, h) supplies k-half h of column j of either tile
__device__ __forceinline__ void wave_columns(const uint4 k0, const uint4 k1, int j, int h, uint4 (&bq)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int src = 32 * s + j;
    uint4 v0, v1;
    v0.x = __shfl(k0.x, src, 64), v0.y = __shfl(k0.y, src, 64), v0.z = __shfl(k0.z, src, 64), v0.w = __shfl(k0.w, src, 64);
    v1.x = __shfl(k1.x, src, 64), v1.y = __shfl(k1.y, src, 64), v1.z = __shfl(k1.z, src, 64), v1.w = __shfl(k1.w, src, 64);
    bq[s] = uint4{h ? v1.x : v0.x, h ? v1.y : v0.y, h ? v1.z : v0.z, h ? v1.w : v0.w};
  }
}

}  // namespace gate
}  // namespace mpa
