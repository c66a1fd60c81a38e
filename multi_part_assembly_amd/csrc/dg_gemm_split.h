// fp32-GRADE GEMMs on the bf16 matrix cores (gfx950): every fp32 operand is split into three bf16 terms,
//     x = h + m + l,  h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)          (exact: 3 x 8 = 24 significand bits)
// and the product is evaluated as the six terms of order <= 2,  hh + hm + mh + mm + hl + lh,  on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The dropped terms (ml, lm, ll) are below 2^-23 |x||y| per
// element — the size of ONE fp32 rounding of the product — so the result differs from an fp32 GEMM by no more than a
// different summation order does (tests hold the encoder to 1e-4 / 2e-4 of float64 as before).  Six bf16 MFMAs of K = 16
// take 6 x 32 = 192 cycles per SIMD where the eight v_mfma_f32_32x32x2_f32 of the same 16 k-values take 512: the
// large GEMMs of the DGCNN encoder and of the graph networks' MLP layers move from the fp32-MFMA bound (100 TFLOP/s
// measured = 64 % of its peak) to the HBM / LDS bound.  Same tiles, grids and calling conventions as dg_gemm.h.
//
//   gemm_nt_split :  C[r, n] (+)= sum_k A[r, k] * W[n, k]
//   gemm_tn_split :  P[chunk][n, k] = sum_{r in chunk} Y[r, n] * X[r, k]
//
// LDS panels hold the three bf16 planes of a 32-wide k chunk side by side: row = [h(32) | m(32) | l(32)] bf16 = 192
// bytes + 16 of padding (208 = 13 x 16: the 16-byte fragment reads of 16 consecutive rows fall on 16 different bank
// quads).  The split is computed while the operands are staged (13 VALU operations per pair of elements).
#pragma once

#include <hip/hip_runtime.h>

#include "dg_gemm.h"

namespace dg {

typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gs_bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kGsRow = 208;  // bytes per LDS row: 3 planes x 32 bf16 + 16 pad

struct Split4 {
  gs_bf16x4 h, m, l;
};
__device__ __forceinline__ Split4 gs_split(const float4 v) {
  const float f[4] = {v.x, v.y, v.z, v.w};
  Split4 s;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    s.h[u] = (__bf16)f[u];
    const float r1 = f[u] - (float)s.h[u];
    s.m[u] = (__bf16)r1;
    s.l[u] = (__bf16)(r1 - (float)s.m[u]);
  }
  return s;
}
// one thread's float4 (columns 4 c4 .. 4 c4 + 3 of the chunk) -> the three planes of LDS row `row`
__device__ __forceinline__ void gs_stash(unsigned char* panel, int row, int c4, const float4 v) {
  const Split4 s = gs_split(v);
  unsigned char* p = panel + row * kGsRow + 8 * c4;
  *reinterpret_cast<gs_bf16x4*>(p) = s.h;
  *reinterpret_cast<gs_bf16x4*>(p + 64) = s.m;
  *reinterpret_cast<gs_bf16x4*>(p + 128) = s.l;
}
struct Frag3 {
  gs_bf16x8 h, m, l;
};
// fragment of row `row` for k-step s (16 k-values; lane half hh takes the second 8)
__device__ __forceinline__ Frag3 gs_frag(const unsigned char* panel, int row, int s, int hh) {
  const unsigned char* p = panel + row * kGsRow + 32 * s + 16 * hh;
  Frag3 f;
  f.h = *reinterpret_cast<const gs_bf16x8*>(p);
  f.m = *reinterpret_cast<const gs_bf16x8*>(p + 64);
  f.l = *reinterpret_cast<const gs_bf16x8*>(p + 128);
  return f;
}
#define GS_MMA6(acc, a, b)                                                          \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).h, (b).h, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).h, (b).m, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).m, (b).h, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).m, (b).m, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).h, (b).l, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).l, (b).h, acc, 0, 0, 0);

// ---- C[r, n] (+)= A[r, :] . W[n, :] -----------------------------------------------------------------------------------
// Arguments, tiles and the XCD-aware block -> tile mapping exactly as gemm_nt_kernel (dg_gemm.h).
template <int BN, bool ACCUM>
__global__ __launch_bounds__(kGT, 2) void gemm_nt_split_kernel(const float* __restrict__ A, int lda,
                                                               const float* __restrict__ W, int K, float* __restrict__ C,
                                                               int ldc, const int* __restrict__ hdr) {
  constexpr int BM = 128, WN = BN / 2, TN = WN / 32;
  constexpr int B4 = BN * kKC / 4 / kGT;  // float4 per thread and chunk of the W panel: 4 or 2
  __shared__ __attribute__((aligned(16))) unsigned char As[BM * kGsRow];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * kGsRow];
  const int R = hdr[1];
  long long r0;
  int n0;
  {
    const int gy = (int)gridDim.y, L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, xcd = L & 7, k = L >> 3;
    r0 = (long long)((k / gy) * 8 + xcd) * BM;
    n0 = (k % gy) * BN;
  }
  if (r0 >= R) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;
  static_assert(B4 == 4 || B4 == 2, "staging layout");
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2 = {}, rb3 = {};
  const float* ap_[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long r = r0 + rl + 32 * i;
    ap_[i] = A + (r < R ? r : (long long)R - 1) * lda + 4 * c4;
  }
  const float* wp_ = W + (long long)(n0 + rl) * K + 4 * c4;
#define GS_NT_FETCH(kc)                                                        \
  ra0 = *reinterpret_cast<const float4*>(ap_[0] + (kc));                       \
  ra1 = *reinterpret_cast<const float4*>(ap_[1] + (kc));                       \
  ra2 = *reinterpret_cast<const float4*>(ap_[2] + (kc));                       \
  ra3 = *reinterpret_cast<const float4*>(ap_[3] + (kc));                       \
  rb0 = *reinterpret_cast<const float4*>(wp_ + (kc));                          \
  rb1 = *reinterpret_cast<const float4*>(wp_ + 32LL * K + (kc));               \
  if constexpr (B4 == 4) {                                                     \
    rb2 = *reinterpret_cast<const float4*>(wp_ + 64LL * K + (kc));             \
    rb3 = *reinterpret_cast<const float4*>(wp_ + 96LL * K + (kc));             \
  }
#define GS_NT_STASH()                                                          \
  gs_stash(As, rl + 0, c4, ra0);                                               \
  gs_stash(As, rl + 32, c4, ra1);                                              \
  gs_stash(As, rl + 64, c4, ra2);                                              \
  gs_stash(As, rl + 96, c4, ra3);                                              \
  gs_stash(Bs, rl + 0, c4, rb0);                                               \
  gs_stash(Bs, rl + 32, c4, rb1);                                              \
  if constexpr (B4 == 4) {                                                     \
    gs_stash(Bs, rl + 64, c4, rb2);                                            \
    gs_stash(Bs, rl + 96, c4, rb3);                                            \
  }
  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x16{0};
  GS_NT_FETCH(0)
  const int chunks = K / kKC;
  for (int c = 0; c < chunks; ++c) {
    if (c > 0) __syncthreads();  // the previous chunk's fragment reads are done
    GS_NT_STASH()
    __syncthreads();
    if (c + 1 < chunks) {
      GS_NT_FETCH((c + 1) * kKC)
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag3 fa[2], fb[TN];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = gs_frag(As, wr * 64 + a * 32 + j, s, h);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = gs_frag(Bs, wc * WN + b * 32 + j, s, h);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          GS_MMA6(acc[a][b], fa[a], fb[b])
        }
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long row = r0 + wr * 64 + a * 32 + acc_row(r, h);
      if (row < R) {
        float* dst = C + row * ldc + n0 + wc * WN + j;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          if constexpr (ACCUM) dst[32 * b] += acc[a][b][r];
          else dst[32 * b] = acc[a][b][r];
        }
      }
    }
#undef GS_NT_FETCH
#undef GS_NT_STASH
}

// ---- P[chunk][n, k] = sum over the chunk's rows of Y[r, n] * X[r, k] ----------------------------------------------------
// Arguments and grid as gemm_tn_kernel.  The MFMA reduction index is the point row, so both operands are TRANSPOSED while
// they are staged: a thread's float4 (one row, four columns) becomes 3 x 4 two-byte stores into the rows [column][row] of
// the panels.  The four 8-row groups of a panel row are XOR-swizzled by (column >> 4) & 3: the stores of an instruction
// fall on 16 different banks (2-way) instead of 4, and a fragment (8 consecutive rows = one aligned 16-byte group) stays
// contiguous and in order.
template <int BK>
__global__ __launch_bounds__(kGT, 2) void gemm_tn_split_kernel(const float* __restrict__ Y, int ldy, int Nout,
                                                               const float* __restrict__ X, int ldx, int K,
                                                               float* __restrict__ part, int rows_per_chunk,
                                                               const int* __restrict__ hdr) {
  constexpr int BNT = 128, RC = 32;
  constexpr int WK = BK / 2, TK = WK / 32;
  constexpr int Y4 = RC * BNT / 4 / kGT;  // 4
  constexpr int X4 = RC * BK / 4 / kGT;   // 2 or 4
  __shared__ __attribute__((aligned(16))) unsigned char Ys[BNT * kGsRow];
  __shared__ __attribute__((aligned(16))) unsigned char Xs[BK * kGsRow];
  const int R = hdr[1];
  const int n0 = blockIdx.x * BNT, k0 = blockIdx.y * BK;
  const long long rb = (long long)blockIdx.z * rows_per_chunk;
  long long re = rb + rows_per_chunk;
  if (re > R) re = R;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wn = wave >> 1, wk = wave & 1;
  float4 ry[Y4], rx[X4];
  auto fetch = [&](long long r) {
#pragma unroll
    for (int i = 0; i < Y4; ++i) {
      const int e = threadIdx.x + i * kGT, row = e / (BNT / 4), c4 = e % (BNT / 4);
      const bool ok = r + row < re && n0 + 4 * c4 < Nout;
      const float4 t = *reinterpret_cast<const float4*>(Y + (ok ? (r + row) * ldy + n0 + 4 * c4 : 0));
      ry[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
#pragma unroll
    for (int i = 0; i < X4; ++i) {
      const int e = threadIdx.x + i * kGT, row = e / (BK / 4), c4 = e % (BK / 4);
      const bool ok = r + row < re;
      const float4 t = *reinterpret_cast<const float4*>(X + (ok ? (r + row) * ldx + k0 + 4 * c4 : 0));
      rx[i] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
    }
  };
  // element (row r of the step, column col) -> byte offset of its h plane entry
  auto slot = [](int col, int r) { return col * kGsRow + 16 * ((r >> 3) ^ ((col >> 4) & 3)) + 2 * (r & 7); };
  auto put = [&](unsigned char* panel, int col0, int r, const float4 v) {
    const Split4 s = gs_split(v);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned char* p = panel + slot(col0 + u, r);
      *reinterpret_cast<__bf16*>(p) = s.h[u];
      *reinterpret_cast<__bf16*>(p + 64) = s.m[u];
      *reinterpret_cast<__bf16*>(p + 128) = s.l[u];
    }
  };
  auto frag = [&](const unsigned char* panel, int col, int s) {  // rows 16 s + 8 h .. + 7 of column `col`
    const unsigned char* p = panel + col * kGsRow + 16 * ((2 * s + h) ^ ((col >> 4) & 3));
    Frag3 f;
    f.h = *reinterpret_cast<const gs_bf16x8*>(p);
    f.m = *reinterpret_cast<const gs_bf16x8*>(p + 64);
    f.l = *reinterpret_cast<const gs_bf16x8*>(p + 128);
    return f;
  };
  f32x16 acc[2][TK];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = f32x16{0};
  if (rb < re) {
    fetch(rb);
    for (long long r = rb; r < re; r += RC) {
      if (r > rb) __syncthreads();  // the previous step's fragment reads are done
#pragma unroll
      for (int i = 0; i < Y4; ++i) {
        const int e = threadIdx.x + i * kGT;
        put(Ys, 4 * (e % (BNT / 4)), e / (BNT / 4), ry[i]);
      }
#pragma unroll
      for (int i = 0; i < X4; ++i) {
        const int e = threadIdx.x + i * kGT;
        put(Xs, 4 * (e % (BK / 4)), e / (BK / 4), rx[i]);
      }
      __syncthreads();
      if (r + RC < re) fetch(r + RC);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag3 fy[2], fx[TK];
#pragma unroll
        for (int a = 0; a < 2; ++a) fy[a] = frag(Ys, wn * 64 + 32 * a + j, s);
#pragma unroll
        for (int b = 0; b < TK; ++b) fx[b] = frag(Xs, wk * WK + 32 * b + j, s);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < TK; ++b) {
            GS_MMA6(acc[a][b], fy[a], fx[b])
          }
      }
    }
  }
  float* out = part + (long long)blockIdx.z * Nout * K;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * 64 + a * 32 + acc_row(r, h);
      if (n < Nout) {
#pragma unroll
        for (int b = 0; b < TK; ++b) out[(long long)n * K + k0 + wk * WK + 32 * b + j] = acc[a][b][r];
      }
    }
}

}  // namespace dg
