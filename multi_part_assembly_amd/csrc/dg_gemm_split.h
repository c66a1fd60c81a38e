// fp32-GRADE GEMMs on the bf16 matrix cores (gfx950): every fp32 operand is split into three bf16 terms,
//     x = h + m + l,  h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)          (exact: 3 x 8 = 24 significand bits)
// and the product is evaluated as the six terms of order <= 2,  hh + hm + mh + mm + hl + lh,  on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The dropped terms (ml, lm, ll) are below 2^-23 |x||y| per
// element — the size of ONE fp32 rounding of the product — so the result differs from an fp32 GEMM by no more than a
// different summation order does (tests hold the encoder to 1e-4 / 2e-4 of float64 as before).  Six bf16 MFMAs of K = 16
// take 6 x 32 = 192 cycles per SIMD where the eight v_mfma_f32_32x32x2_f32 of the same 16 k-values take 512.
// Same tiles, grids and calling conventions as dg_gemm.h.
//
//   gemm_nt_split :  C[r, n] (+)= sum_k A[r, k] * W[n, k]
//   gemm_tn_split :  P[chunk][n, k] = sum_{r in chunk} Y[r, n] * X[r, k]
//
// LDS panels hold the three bf16 planes of a 32-wide k chunk side by side: row = [h(32) | m(32) | l(32)] bf16 = 192
// bytes + 16 of padding (208 = 13 x 16: the 16-byte fragment reads of 16 consecutive rows fall on 16 different bank
// quads).  The split is computed while the operands are staged (13 VALU operations per pair of elements).
//
// Block = kGsT threads.  tools/probes/gemm_split.hip takes the kernels apart at the encoder's shapes: the three phases of
// a K step — global loads, split + LDS stores, MFMAs — cost about 100 + 145 + 200 us of a 470 us weight gradient and
// add up rather than overlap, with four waves per block (two blocks per CU) or eight (DG_GS_WAVES: 64 x 32 wave tiles,
// four waves per SIMD), with one K step of loads in flight or two: the time is the same within 3 %.  The loads of the
// big shapes move 1.4 GB at ~4.6 TB/s when they run alone; the MFMAs alone run at 240 TFLOP/s fp32-equivalent (the
// bf16 pipe's sustained 1.85 PFLOP/s over six products).  Four waves are the default.
#pragma once

#include <hip/hip_runtime.h>

#include "dg_gemm.h"

namespace dg {

typedef __bf16 gs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gs_bf16x4 __attribute__((ext_vector_type(4)));

#ifndef DG_GS_WAVES  // A/B knob: waves per block (4: the 2 x 2 arrangement of dg_gemm.h; 8: 2 x 4 / 4 x 2)
#define DG_GS_WAVES 4
#endif
#ifndef DG_GS_ROWPERM  // 1: staging threads take panel rows four apart (conflict-free stores); 0: consecutive rows (rounds 3-5a)
#define DG_GS_ROWPERM 1
#endif
constexpr int kGsT = 64 * DG_GS_WAVES;  // threads per block
constexpr int kGsRow = 208;             // bytes per LDS row: 3 planes x 32 bf16 + 16 pad

// probe knobs (tools/probes/gemm_split.hip): which of the three phases of a K step the kernels execute
#ifdef GS_PROBE_NO_STASH
#define GS_STASH_ON (hdr[0] == -12345)
#else
#define GS_STASH_ON true
#endif
#ifdef GS_PROBE_NO_LOAD
#define GS_LOAD_ON (hdr[0] == -12345)
#else
#define GS_LOAD_ON true
#endif

struct Split4 {
  gs_bf16x4 h, m, l;
};
#ifndef DG_GS_SPLIT_PK  // 1: the split on register pairs (v_cvt_pk_bf16_f32 + v_pk_add_f32: ~4.5 VALU operations per element,
#define DG_GS_SPLIT_PK 1  // the same bits); 0: element by element (~8; rounds 3-5)
#endif
typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gs_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Split4 gs_split(const float4 v) {
  Split4 s;
#if DG_GS_SPLIT_PK
  auto split2 = [](gs_f32x2 x, gs_bf16x2& h, gs_bf16x2& m, gs_bf16x2& l) {
    h = __builtin_convertvector(x, gs_bf16x2);
    const gs_f32x2 r = x - __builtin_convertvector(h, gs_f32x2);
    m = __builtin_convertvector(r, gs_bf16x2);
    const gs_f32x2 t = r - __builtin_convertvector(m, gs_f32x2);
    l = __builtin_convertvector(t, gs_bf16x2);
  };
  gs_bf16x2 h0, m0, l0, h1, m1, l1;
  split2(gs_f32x2{v.x, v.y}, h0, m0, l0);
  split2(gs_f32x2{v.z, v.w}, h1, m1, l1);
  s.h = gs_bf16x4{h0[0], h0[1], h1[0], h1[1]};
  s.m = gs_bf16x4{m0[0], m0[1], m1[0], m1[1]};
  s.l = gs_bf16x4{l0[0], l0[1], l1[0], l1[1]};
#else
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    s.h[u] = (__bf16)f[u];
    const float r1 = f[u] - (float)s.h[u];
    s.m[u] = (__bf16)r1;
    s.l[u] = (__bf16)(r1 - (float)s.m[u]);
  }
#endif
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
#ifdef GS_PROBE_NO_MMA  // probe knob: staging without the matrix products
#define GS_MMA6(acc, a, b) acc[0] += (float)(a).h[0] * (float)(b).l[0];
#else
#define GS_MMA6(acc, a, b)                                                          \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).h, (b).h, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).h, (b).m, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).m, (b).h, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).m, (b).m, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).h, (b).l, acc, 0, 0, 0);        \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16((a).l, (b).h, acc, 0, 0, 0);
#endif

// 4 x 4 transpose of bf16 values inside a lane quad: lane t holds (row t; columns 0..3) and receives (column t; rows
// 0..3) — two exchange stages (lane ^ 1, lane ^ 2), each one DPP move and one byte permute / select per dword.
//   sel1 = lane odd ? 0x03020706 : 0x05040100,  low = (lane & 2) == 0
__device__ __forceinline__ uint2 gs_quad_transpose(const gs_bf16x4 x, unsigned sel1, bool low) {
  const uint2 w = __builtin_bit_cast(uint2, x);
  const unsigned p0 = (unsigned)__builtin_amdgcn_mov_dpp((int)w.x, 0xb1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
  const unsigned p1 = (unsigned)__builtin_amdgcn_mov_dpp((int)w.y, 0xb1, 0xf, 0xf, true);
  const unsigned a0 = __builtin_amdgcn_perm(p0, w.x, sel1);  // column (lane & 1), rows of the lane pair
  const unsigned a1 = __builtin_amdgcn_perm(p1, w.y, sel1);  // column 2 + (lane & 1)
  const unsigned send = low ? a1 : a0;
  const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0x4e, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
  return low ? make_uint2(a0, recv) : make_uint2(recv, a1);
}

// waves of a block as WR x WC over a 128 x BN tile: wave tiles of 64 x (BN / 2) with four waves, 64 x 32 or 32 x 32
// with eight
template <int BN>
struct GsWaves {
  static constexpr int WC = DG_GS_WAVES == 4 ? 2 : (BN == 128 ? 4 : 2), WR = DG_GS_WAVES / WC;
  static constexpr int TM = 128 / WR / 32, TN = BN / WC / 32;  // 32 x 32 accumulators per wave
  static_assert(TM >= 1 && TN >= 1, "wave tile");
};

// Optional work of the output pass of gemm_nt_split_kernel (the MLP layers of csrc/mlp.hip):
//   EPI 0: none.   EPI 1: C += bias[n], and stats[(row tile)][n] = (sum, sum of squares) of the tile's valid rows of C —
//   the BatchNorm statistics pass of a layer.   EPI 2: C = relu?(C + bias[n]) — a layer without BatchNorm.
struct GsEpi {
  const float* bias = nullptr;  // [Nout] or NULL
  float* stats = nullptr;       // [ceil(R / 128)][ldstats][2]
  int ldstats = 0;
  int relu = 0;
  int rows = 0;                 // the row count R when `hdr` is NULL (callers that know it on the host)
  unsigned* zero = nullptr;     // 64 words cleared by the first block (the tickets of the layer's cooperative reductions)
};

// ---- C[r, n] (+)= A[r, :] . W[n, :] -----------------------------------------------------------------------------------
// Arguments, tiles and the XCD-aware block -> tile mapping exactly as gemm_nt_kernel (dg_gemm.h); block = kGsT threads.
// WT: the second operand is given TRANSPOSED, Wt [K][Nout] with leading dimension ldwt (the
// input-gradient GEMM dX = dY . W reads the layer's weight as it is stored): its panel is staged through the 4 x 4
// quad transpose of gemm_tn_split_kernel.
template <int BN, bool ACCUM, int EPI = 0, bool WT = false>
__global__ __launch_bounds__(kGsT, DG_GS_WAVES == 8 ? 4 : 2) void gemm_nt_split_kernel(const float* __restrict__ A, int lda,
                                                            const float* __restrict__ W, int K, float* __restrict__ C,
                                                            int ldc, const int* __restrict__ hdr, const GsEpi epi,
                                                            int ldwt) {
  constexpr int BM = 128;
  using WV = GsWaves<BN>;
  constexpr int TM = WV::TM, TN = WV::TN;
  constexpr int RS = kGsT / 8;                 // panel rows staged per pass of the block (8 threads per 32-float row)
  constexpr int A4 = BM / RS, B4 = BN / RS;    // float4 per thread and chunk
  __shared__ __attribute__((aligned(16))) unsigned char As[BM * kGsRow];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[BN * kGsRow];
  const int R = hdr != nullptr ? hdr[1] : epi.rows;
  if (EPI != 0 && epi.zero != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) epi.zero[threadIdx.x] = 0u;
  long long r0;
  int n0;
  {
    const int gy = (int)gridDim.y, L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, xcd = L & 7, k = L >> 3;
    r0 = (long long)((k / gy) * 8 + xcd) * BM;
    n0 = (k % gy) * BN;
  }
  if (r0 >= R) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wr = wave / WV::WC, wc = wave % WV::WC;
  // Panel row of a staging thread.  Eight threads share a row; the 32 lanes of one LDS write cycle hold four rows, and with
  // 208-byte rows (52 dwords) CONSECUTIVE rows start 52 banks apart: their 16-dword runs overlap pairwise, a 2-way conflict
  // on every plane's store (a third of the kernel's LDS cycles: profiles/r05b_c3_pmc_lds_counters.txt).  Rows four apart
  // start 16 banks apart — four disjoint runs — so slot g of a pass takes row 4 (g % 4) + (g / 4) % 4 + 16 (g / 16).
  const int c4 = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const int rl = DG_GS_ROWPERM ? ((rg & 3) << 2) + ((rg >> 2) & 3) + (rg & ~15) : rg;
  float4 ra[A4], rb[B4];
  const float* ap_[A4];
#pragma unroll
  for (int i = 0; i < A4; ++i) {
    const long long r = r0 + rl + RS * i;
    ap_[i] = A + (r < R ? r : (long long)R - 1) * lda + 4 * c4;
  }
  // W panel.  Plain: thread = (output column rl + RS i, k quad c4).  WT: thread = (k row 4 g + qt of the chunk, output
  // column quad wc4) with g = (tid >> 2 & 1) + 2 ((tid >> 3) / (BN / 4)) + WS i — the mapping of gemm_tn_split_kernel.
  constexpr int WC4 = BN / 4, WS = 2 * (kGsT / 8) / WC4;
  static_assert(!WT || 8 / WS == B4, "transposed W panel: float4 per thread");
  const int qt = threadIdx.x & 3, wc4 = (threadIdx.x >> 3) % WC4, wg = ((threadIdx.x >> 2) & 1) + 2 * ((threadIdx.x >> 3) / WC4);
  const unsigned sel1 = (qt & 1) ? 0x03020706u : 0x05040100u;
  const bool qlow = (qt & 2) == 0;
  const float* wp_ = WT ? W + (long long)(4 * wg + qt) * ldwt + n0 + 4 * wc4 : W + (long long)(n0 + rl) * K + 4 * c4;
  auto fetch = [&](int kc) {
#pragma unroll
    for (int i = 0; i < A4; ++i) ra[i] = *reinterpret_cast<const float4*>(ap_[i] + kc);
#pragma unroll
    for (int i = 0; i < B4; ++i) {
      if constexpr (WT) rb[i] = *reinterpret_cast<const float4*>(wp_ + (long long)(kc + 4 * WS * i) * ldwt);
      else rb[i] = *reinterpret_cast<const float4*>(wp_ + (long long)(RS * i) * K + kc);
    }
  };
  auto stash_w = [&](int i) {
    if constexpr (WT) {  // the quad's 4 k rows x 4 columns -> this lane: column 4 wc4 + qt, k = 4 g .. 4 g + 3
      const Split4 sp = gs_split(rb[i]);
      unsigned char* p = Bs + (4 * wc4 + qt) * kGsRow + 8 * (wg + WS * i);
      *reinterpret_cast<uint2*>(p) = gs_quad_transpose(sp.h, sel1, qlow);
      *reinterpret_cast<uint2*>(p + 64) = gs_quad_transpose(sp.m, sel1, qlow);
      *reinterpret_cast<uint2*>(p + 128) = gs_quad_transpose(sp.l, sel1, qlow);
    } else {
      gs_stash(Bs, rl + RS * i, c4, rb[i]);
    }
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x16{0};
  fetch(0);
  const int chunks = K / kKC;
  for (int c = 0; c < chunks; ++c) {
    if (c > 0) __syncthreads();  // the previous chunk's fragment reads are done
    if (GS_STASH_ON) {
#pragma unroll
      for (int i = 0; i < A4; ++i) gs_stash(As, rl + RS * i, c4, ra[i]);
#pragma unroll
      for (int i = 0; i < B4; ++i) stash_w(i);
    }
    __syncthreads();
    if (c + 1 < chunks && GS_LOAD_ON) fetch((c + 1) * kKC);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Frag3 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[a] = gs_frag(As, wr * (32 * TM) + a * 32 + j, s, h);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = gs_frag(Bs, wc * (32 * TN) + b * 32 + j, s, h);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          GS_MMA6(acc[a][b], fa[a], fb[b])
        }
    }
  }
  float bcol[TN], cs[TN], css[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    bcol[b] = (EPI != 0 && epi.bias != nullptr) ? epi.bias[n0 + wc * (32 * TN) + 32 * b + j] : 0.0f;
    cs[b] = css[b] = 0.0f;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long long row = r0 + wr * (32 * TM) + a * 32 + acc_row(r, h);
      if (row < R) {
        float* dst = C + row * ldc + n0 + wc * (32 * TN) + j;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          float v = acc[a][b][r];
          if constexpr (EPI != 0) v += bcol[b];
          if constexpr (EPI == 2) v = epi.relu ? __builtin_fmaxf(v, 0.0f) : v;
          if constexpr (ACCUM) dst[32 * b] += v;
          else dst[32 * b] = v;
          if constexpr (EPI == 1) {
            cs[b] += v;
            css[b] = __builtin_fmaf(v, v, css[b]);
          }
        }
      }
    }
  if constexpr (EPI == 1) {
    // column sums of the tile: lane halves by shuffle, the WR wave rows through LDS in fixed order
    float(*red)[BN][2] = reinterpret_cast<float(*)[BN][2]>(As);  // [WR][BN][2]; the panels are free after the barrier
    __syncthreads();
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      cs[b] += __shfl_xor(cs[b], 32, 64);
      css[b] += __shfl_xor(css[b], 32, 64);
      if (h == 0) {
        red[wr][wc * (32 * TN) + 32 * b + j][0] = cs[b];
        red[wr][wc * (32 * TN) + 32 * b + j][1] = css[b];
      }
    }
    __syncthreads();
    if (threadIdx.x < BN) {
      float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
      for (int q = 0; q < WV::WR; ++q) {
        t0 += red[q][threadIdx.x][0];
        t1 += red[q][threadIdx.x][1];
      }
      float* d = epi.stats + ((r0 / BM) * epi.ldstats + n0 + threadIdx.x) * 2;
      d[0] = t0;
      d[1] = t1;
    }
  }
}

// ---- P[chunk][n, k] = sum over the chunk's rows of Y[r, n] * X[r, k] ----------------------------------------------------
// Arguments and grid as gemm_tn_kernel; block = kGsT threads.  The MFMA reduction index is the point row, so both operands
// are TRANSPOSED while they are staged into the rows [column][row] of the panels.  The four lanes of a quad load four
// consecutive rows of the same four columns, transpose the 4 x 4 block of every bf16 plane in registers
// (gs_quad_transpose) and store 8 bytes (one column, four rows) each: 3 ds_write_b64 per float4 where element-wise
// stores take 12 ds_write_b16.  A 16-lane store group is 8 columns x 2 row groups = 32 different banks.  The four 8-row
// groups of a panel row are XOR-swizzled by (column >> 4) & 3 (a fragment = 8 consecutive rows = one aligned 16-byte
// group).
#ifndef DG_GS_TN_TR  // 1: row-major panels + transposed fragment reads (ds_read_b64_tr_b16, round 6); 0: the quad-transpose staging
#define DG_GS_TN_TR 1
#endif
#if DG_GS_TN_TR
// Round 6: the panels hold the step's 32 rows AS THEY ARE LOADED — row = [h | m | l] planes of the tile's columns — and the
// fragments come out transposed from the LDS (ds_read_b64_tr_b16: two reads per plane and k-step, pn_bwd_q.h's lane map).  The
// 4 x 4 quad transposes of the staging (two DPP moves, two byte permutes and a select per plane and float4, on top of the
// split) are gone: the staging waves' VALU issue is what bounds these kernels (LABBOOK 6.3).
template <int BK>
__global__ __launch_bounds__(kGsT, DG_GS_WAVES == 8 ? 4 : 2) void gemm_tn_split_kernel(const float* __restrict__ Y, int ldy, int Nout,
                                                            const float* __restrict__ X, int ldx, int K,
                                                            float* __restrict__ part, int rows_per_chunk,
                                                            const int* __restrict__ hdr, int rows) {
  constexpr int BNT = 128, RC = 32;
  constexpr int WKW = DG_GS_WAVES == 4 ? 2 : (BK == 128 ? 4 : 2), WNW = DG_GS_WAVES / WKW;
  constexpr int TNn = BNT / WNW / 32, TK = BK / WKW / 32;
  constexpr int YC = BNT / 4, XC = BK / 4;              // column quads per row
  constexpr int YP = kGsT / YC, XP = kGsT / XC;         // rows staged per pass of the block
  constexpr int Y4 = RC / YP, X4 = RC / XP;             // float4 per thread and step
  constexpr int YROW = 6 * BNT + 16, XROW = 6 * BK + 16;  // panel row bytes: three planes + pad (odd multiples of 16)
  static_assert(YP >= 1 && XP >= 1 && RC % YP == 0 && RC % XP == 0, "staging layout");
  __shared__ __attribute__((aligned(16))) unsigned char Ys[RC * YROW];
  __shared__ __attribute__((aligned(16))) unsigned char Xs[RC * XROW];
  const int R = hdr != nullptr ? hdr[1] : rows;
  int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
  if (gridDim.z % 8 == 0) {  // (a chunk's tiles on ONE XCD: see the note on the block -> tile mapping above)
    const int gx = (int)gridDim.x, tiles = gx * (int)gridDim.y, L = (bz * (int)gridDim.y + by) * gx + bx;
    const int q = L >> 3, t = q % tiles;
    bz = (q / tiles) * 8 + (L & 7);
    bx = t % gx;
    by = t / gx;
  }
  const int n0 = bx * BNT, k0 = by * BK;
  const int rpc = rows_per_chunk > 0 ? rows_per_chunk : (int)((((long long)R + gridDim.z - 1) / gridDim.z + 31) / 32 * 32);
  const long long rb = (long long)bz * rpc;
  long long re = rb + rpc;
  if (re > R) re = R;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wn = wave / WKW, wk = wave % WKW;
  float4 ry[Y4], rx[X4];
  // thread -> (row tid / C + P i of the step, column quad tid % C)
  const int yc4 = threadIdx.x % YC, yr = threadIdx.x / YC;
  const int xc4 = threadIdx.x % XC, xr = threadIdx.x / XC;
  const bool ycol_ok = n0 + 4 * yc4 < Nout;
  const float* ybase = Y + (ycol_ok ? n0 + 4 * yc4 : 0);
  const float* xbase = X + k0 + 4 * xc4;
  // loads are unconditional (rows clamped to the chunk's last one); what lies past the chunk is zeroed when it is staged
  auto fetch = [&](long long r) {
#pragma unroll
    for (int i = 0; i < Y4; ++i) {
      const long long row = r + yr + YP * i;
      ry[i] = *reinterpret_cast<const float4*>(ybase + (row < re ? row : re - 1) * ldy);
    }
#pragma unroll
    for (int i = 0; i < X4; ++i) {
      const long long row = r + xr + XP * i;
      rx[i] = *reinterpret_cast<const float4*>(xbase + (row < re ? row : re - 1) * ldx);
    }
  };
  auto put = [&](unsigned char* prow, int plane_bytes, int c4, const float4 v0, bool ok) {
    const float4 v = make_float4(ok ? v0.x : 0.f, ok ? v0.y : 0.f, ok ? v0.z : 0.f, ok ? v0.w : 0.f);
    const Split4 s = gs_split(v);
    unsigned char* p = prow + 8 * c4;
    *reinterpret_cast<gs_bf16x4*>(p) = s.h;
    *reinterpret_cast<gs_bf16x4*>(p + plane_bytes) = s.m;
    *reinterpret_cast<gs_bf16x4*>(p + 2 * plane_bytes) = s.l;
  };
  // transposed fragment of column tile `ct` (32 columns), k-step s (16 rows): lane's address = row 8 (g >> 1) + (i >> 2),
  // column 16 (g & 1) + 4 (i & 3)  (i = lane & 15, g = lane >> 4); the second read 4 rows further
  const int g16 = lane >> 4, i16 = lane & 15;
  const int trow = 8 * (g16 >> 1) + (i16 >> 2), tcol = 16 * (g16 & 1) + 4 * (i16 & 3);
  typedef short gs_s16x4 __attribute__((ext_vector_type(4)));
  typedef short gs_s16x8 __attribute__((ext_vector_type(8)));
  auto tr1 = [&](const unsigned char* p, int rowb) {
    const gs_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs_s16x4*)p);
    const gs_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs_s16x4*)(p + 4 * rowb));
    const gs_s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(gs_bf16x8, v);
  };
  auto frag = [&](const unsigned char* panel, int rowb, int plane_bytes, int ct, int s) {
    const unsigned char* p = panel + (16 * s + trow) * rowb + 2 * (32 * ct + tcol);
    Frag3 f;
    f.h = tr1(p, rowb);
    f.m = tr1(p + plane_bytes, rowb);
    f.l = tr1(p + 2 * plane_bytes, rowb);
    return f;
  };
  f32x16 acc[TNn][TK];
#pragma unroll
  for (int a = 0; a < TNn; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = f32x16{0};
  if (rb < re) {
    fetch(rb);
    for (long long r = rb; r < re; r += RC) {
      if (r > rb) __syncthreads();  // the previous step's fragment reads are done
      if (GS_STASH_ON) {
#pragma unroll
        for (int i = 0; i < Y4; ++i) {
          const int rl = yr + YP * i;
          put(Ys + rl * YROW, 2 * BNT, yc4, ry[i], ycol_ok && r + rl < re);
        }
#pragma unroll
        for (int i = 0; i < X4; ++i) {
          const int rl = xr + XP * i;
          put(Xs + rl * XROW, 2 * BK, xc4, rx[i], r + rl < re);
        }
      }
      __syncthreads();
      if (r + RC < re && GS_LOAD_ON) fetch(r + RC);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag3 fy[TNn], fx[TK];
#pragma unroll
        for (int a = 0; a < TNn; ++a) fy[a] = frag(Ys, YROW, 2 * BNT, wn * TNn + a, s);
#pragma unroll
        for (int b = 0; b < TK; ++b) fx[b] = frag(Xs, XROW, 2 * BK, wk * TK + b, s);
#pragma unroll
        for (int a = 0; a < TNn; ++a)
#pragma unroll
          for (int b = 0; b < TK; ++b) {
            GS_MMA6(acc[a][b], fy[a], fx[b])
          }
      }
    }
  }
  float* out = part + (long long)bz * Nout * K;
#pragma unroll
  for (int a = 0; a < TNn; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * (32 * TNn) + a * 32 + acc_row(r, h);
      if (n < Nout) {
#pragma unroll
        for (int b = 0; b < TK; ++b) out[(long long)n * K + k0 + wk * (32 * TK) + 32 * b + j] = acc[a][b][r];
      }
    }
}
#else
template <int BK>
__global__ __launch_bounds__(kGsT, DG_GS_WAVES == 8 ? 4 : 2) void gemm_tn_split_kernel(const float* __restrict__ Y, int ldy, int Nout,
                                                            const float* __restrict__ X, int ldx, int K,
                                                            float* __restrict__ part, int rows_per_chunk,
                                                            const int* __restrict__ hdr, int rows) {
  constexpr int BNT = 128, RC = 32;
  // waves as WNW x WKW over the 128 x BK tile of the gradient
  constexpr int WKW = DG_GS_WAVES == 4 ? 2 : (BK == 128 ? 4 : 2), WNW = DG_GS_WAVES / WKW;
  constexpr int TNn = BNT / WNW / 32, TK = BK / WKW / 32;
  constexpr int YC = BNT / 4, XC = BK / 4;           // column quads per row
  constexpr int YS = 2 * (kGsT / 8) / YC, XS = 2 * (kGsT / 8) / XC;  // 4-row groups staged per pass of the block
  constexpr int Y4 = 8 / YS, X4 = 8 / XS;            // float4 per thread and step
  static_assert(YS >= 1 && XS >= 1 && YS <= 8 && XS <= 8, "staging layout");
  __shared__ __attribute__((aligned(16))) unsigned char Ys[BNT * kGsRow];
  __shared__ __attribute__((aligned(16))) unsigned char Xs[BK * kGsRow];
  const int R = hdr != nullptr ? hdr[1] : rows;  // `rows`: the row count for callers that know it on the host
  // block -> (n tile, k tile, row chunk).  The tiles of one row chunk read the same rows of Y and X; workgroups go to
  // the 8 XCDs round-robin (linear id L runs on XCD L % 8) and every XCD has its own L2, so a chunk's tiles are given
  // consecutive slots of ONE XCD (chunks a multiple of 8): its rows come from HBM once instead of once per tile.
  int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
  if (gridDim.z % 8 == 0) {
    const int gx = (int)gridDim.x, tiles = gx * (int)gridDim.y, L = (bz * (int)gridDim.y + by) * gx + bx;
    const int q = L >> 3, t = q % tiles;
    bz = (q / tiles) * 8 + (L & 7);
    bx = t % gx;
    by = t / gx;
  }
  const int n0 = bx * BNT, k0 = by * BK;
  // rows_per_chunk == 0: the valid rows (known on the device only) are dealt evenly to the grid's chunks
  const int rpc = rows_per_chunk > 0 ? rows_per_chunk : (int)((((long long)R + gridDim.z - 1) / gridDim.z + 31) / 32 * 32);
  const long long rb = (long long)bz * rpc;
  long long re = rb + rpc;
  if (re > R) re = R;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wn = wave / WKW, wk = wave % WKW;
  float4 ry[Y4], rx[X4];
  // thread -> (row 4 g + qt, column quad c4):  qt = tid & 3, g = (tid >> 2 & 1) + 2 ((tid >> 3) / C) + S i,  c4 = (tid >> 3) % C
  const int qt = threadIdx.x & 3, t8 = threadIdx.x >> 3;
  const int yc4 = t8 % YC, yg = ((threadIdx.x >> 2) & 1) + 2 * (t8 / YC);
  const int xc4 = t8 % XC, xg = ((threadIdx.x >> 2) & 1) + 2 * (t8 / XC);
  const unsigned sel1 = (qt & 1) ? 0x03020706u : 0x05040100u;
  const bool qlow = (qt & 2) == 0;
  const bool ycol_ok = n0 + 4 * yc4 < Nout;
  const float* ybase = Y + (ycol_ok ? n0 + 4 * yc4 : 0);
  const float* xbase = X + k0 + 4 * xc4;
  // loads are unconditional (rows clamped to the chunk's last one); what lies past the chunk is zeroed when it is staged
  auto fetch = [&](long long r) {
#pragma unroll
    for (int i = 0; i < Y4; ++i) {
      const long long row = r + 4 * (yg + YS * i) + qt;
      ry[i] = *reinterpret_cast<const float4*>(ybase + (row < re ? row : re - 1) * ldy);
    }
#pragma unroll
    for (int i = 0; i < X4; ++i) {
      const long long row = r + 4 * (xg + XS * i) + qt;
      rx[i] = *reinterpret_cast<const float4*>(xbase + (row < re ? row : re - 1) * ldx);
    }
  };
  // the quad's 4 x 4 block (rows 4 g .. 4 g + 3, columns 4 c4 .. 4 c4 + 3): this lane stores column 4 c4 + qt
  auto put = [&](unsigned char* panel, int c4, int g, const float4 v0, bool ok) {
    const float4 v = make_float4(ok ? v0.x : 0.f, ok ? v0.y : 0.f, ok ? v0.z : 0.f, ok ? v0.w : 0.f);
    const Split4 s = gs_split(v);
    const int col = 4 * c4 + qt;
    unsigned char* p = panel + col * kGsRow + 16 * ((g >> 1) ^ ((col >> 4) & 3)) + 8 * (g & 1);
    *reinterpret_cast<uint2*>(p) = gs_quad_transpose(s.h, sel1, qlow);
    *reinterpret_cast<uint2*>(p + 64) = gs_quad_transpose(s.m, sel1, qlow);
    *reinterpret_cast<uint2*>(p + 128) = gs_quad_transpose(s.l, sel1, qlow);
  };
  auto frag = [&](const unsigned char* panel, int col, int s) {  // rows 16 s + 8 h .. + 7 of column `col`
    const unsigned char* p = panel + col * kGsRow + 16 * ((2 * s + h) ^ ((col >> 4) & 3));
    Frag3 f;
    f.h = *reinterpret_cast<const gs_bf16x8*>(p);
    f.m = *reinterpret_cast<const gs_bf16x8*>(p + 64);
    f.l = *reinterpret_cast<const gs_bf16x8*>(p + 128);
    return f;
  };
  f32x16 acc[TNn][TK];
#pragma unroll
  for (int a = 0; a < TNn; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = f32x16{0};
  if (rb < re) {
    fetch(rb);
    for (long long r = rb; r < re; r += RC) {
      if (r > rb) __syncthreads();  // the previous step's fragment reads are done
      if (GS_STASH_ON) {
#pragma unroll
        for (int i = 0; i < Y4; ++i) {
          const int g = yg + YS * i;
          put(Ys, yc4, g, ry[i], ycol_ok && r + 4 * g + qt < re);
        }
#pragma unroll
        for (int i = 0; i < X4; ++i) {
          const int g = xg + XS * i;
          put(Xs, xc4, g, rx[i], r + 4 * g + qt < re);
        }
      }
      __syncthreads();
      if (r + RC < re && GS_LOAD_ON) fetch(r + RC);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Frag3 fy[TNn], fx[TK];
#pragma unroll
        for (int a = 0; a < TNn; ++a) fy[a] = frag(Ys, wn * (32 * TNn) + 32 * a + j, s);
#pragma unroll
        for (int b = 0; b < TK; ++b) fx[b] = frag(Xs, wk * (32 * TK) + 32 * b + j, s);
#pragma unroll
        for (int a = 0; a < TNn; ++a)
#pragma unroll
          for (int b = 0; b < TK; ++b) {
            GS_MMA6(acc[a][b], fy[a], fx[b])
          }
      }
    }
  }
  float* out = part + (long long)bz * Nout * K;
#pragma unroll
  for (int a = 0; a < TNn; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + wn * (32 * TNn) + a * 32 + acc_row(r, h);
      if (n < Nout) {
#pragma unroll
        for (int b = 0; b < TK; ++b) out[(long long)n * K + k0 + wk * (32 * TK) + 32 * b + j] = acc[a][b][r];
      }
    }
}

#endif  // DG_GS_TN_TR

}  // namespace dg
