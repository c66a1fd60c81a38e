// Exact-fp32 matrix-core GEMMs over the point rows of the DGCNN encoder (gfx950, v_mfma_f32_32x32x2_f32).
//
// The reference's EdgeConv stages are 1x1 Conv2d over [n, 2C, N, k] edge tensors and a 1x1 Conv1d over the 512-wide
// concatenation (multi_part_assembly/models/modules/encoder/dgcnn.py:57-71,76-100); here they are plain GEMMs over
// R = (valid parts) x N point rows (csrc/dgcnn_enc.hip explains the algebra).  R is only known on the device (the
// number of valid parts is counted there, no host sync), so every kernel takes `hdr` (hdr[1] = R) and is launched
// for the worst case: tiles past R exit at once.
//
//   gemm_nt :  C[r, n] (+)= sum_k A[r, k] * W[n, k]        forward GEMMs and input gradients (W pre-transposed)
//   gemm_tn :  P[chunk][n, k] = sum_{r in chunk} Y[r, n] * X[r, k]   weight gradients, fixed-order second stage
//
// Block = 256 threads (4 waves, 2 x 2), block tile 128 rows x BN columns, K walked in chunks of 32 through an LDS
// panel pair (global loads of chunk c+1 in flight — in registers — during the MFMA chain of chunk c; ONE panel per
// operand and three blocks per CU measured slightly faster than two panels and two blocks); a wave owns a
// 64 x (BN/2) tile = 2 x (BN/64) accumulators of 32x32.  The MFMA K index is split as "lanes 0-31 take the first half
// of the chunk, lanes 32-63 the second", so a lane's operand fragments are contiguous 16-float runs of an LDS row
// (ds_read_b128, rows padded by 4 floats: conflict-free).
#pragma once

#include <hip/hip_runtime.h>

namespace dg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// MFMA 32x32 accumulator layout: lane l holds column (l & 31) and, in register r, row acc_row(r, l >> 5).
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

constexpr int kGT = 256;   // threads per GEMM block
constexpr int kKC = 32;    // K chunk
constexpr int kLD = kKC + 4;
#define DG_GEMM_GRID_X(rows) ((unsigned)((((rows) + 127) / 128 + 7) / 8 * 8))  // row tiles, rounded up to the XCD count

// ---- C[r, n] (+)= A[r, :] . W[n, :] -----------------------------------------------------------------------------------
// A [R, K] row-major with leading dimension lda (a column slice of a wider buffer is fine), W [Nout, K] row-major,
// C [R, Nout] with leading dimension ldc.  K % 32 == 0, Nout % BN == 0.  grid = (ceil(Rmax / 128), Nout / BN).
#ifndef DG_NT_BUFS  // A/B knob: LDS panels per operand (2: double-buffered, 2 blocks per CU; 1: single, 3 blocks)
#define DG_NT_BUFS 1
#endif
template <int BN, bool ACCUM>
__global__ __launch_bounds__(kGT, DG_NT_BUFS == 2 ? 2 : 3) void gemm_nt_kernel(const float* __restrict__ A, int lda,
                                                         const float* __restrict__ W, int K,
                                                         float* __restrict__ C, int ldc, const int* __restrict__ hdr) {
  constexpr int BM = 128;
  constexpr int WN = BN / 2;        // columns per wave
  constexpr int TN = WN / 32;       // 32-wide column tiles per wave (1 or 2)
  constexpr int A4 = BM * kKC / 4 / kGT;  // float4 per thread and chunk (A panel) = 4
  constexpr int B4 = BN * kKC / 4 / kGT;  // (W panel) = 4 or 2
  __shared__ __attribute__((aligned(16))) float As[DG_NT_BUFS][BM * kLD];
  __shared__ __attribute__((aligned(16))) float Bs[DG_NT_BUFS][BN * kLD];
  const int R = hdr[1];
  // block -> (row tile, column tile).  The column tiles of one row tile all read the same A rows; workgroups go to the
  // 8 XCDs round-robin, so they are given consecutive slots of ONE XCD (linear id L: XCD L % 8, row tile
  // (L / 8 / gy) * 8 + L % 8, column tile (L / 8) % gy): the A tile is fetched from HBM once instead of gy times.
  // gridDim.x is a multiple of 8 (DG_GEMM_GRID_X); tiles past R exit.
  long long r0;
  int n0;
  {
    const int gy = (int)gridDim.y, L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, xcd = L & 7, k = L >> 3;
    r0 = (long long)((k / gy) * 8 + xcd) * BM;
    n0 = (k % gy) * BN;
  }
  if (r0 >= R) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;  // wave tile: rows wr*64.., columns wc*WN..
  const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;  // staging role: float4 column c4 of rows rl + 32 i
  // staged operands as named registers (an indexed float4 array here ends up in scratch memory)
  static_assert(A4 == 4 && (B4 == 4 || B4 == 2), "staging layout");
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2 = {}, rb3 = {};
  const float* ap_[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long long r = r0 + rl + 32 * i;  // rows past R: any valid row will do, their results are never stored
    ap_[i] = A + (r < R ? r : (long long)R - 1) * lda + 4 * c4;
  }
  const float* wp_ = W + (long long)(n0 + rl) * K + 4 * c4;
#define DG_NT_FETCH(kc)                                                        \
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
#define DG_NT_STASH(buf)                                                       \
  *reinterpret_cast<float4*>(&As[buf][(rl + 0) * kLD + 4 * c4]) = ra0;         \
  *reinterpret_cast<float4*>(&As[buf][(rl + 32) * kLD + 4 * c4]) = ra1;        \
  *reinterpret_cast<float4*>(&As[buf][(rl + 64) * kLD + 4 * c4]) = ra2;        \
  *reinterpret_cast<float4*>(&As[buf][(rl + 96) * kLD + 4 * c4]) = ra3;        \
  *reinterpret_cast<float4*>(&Bs[buf][(rl + 0) * kLD + 4 * c4]) = rb0;         \
  *reinterpret_cast<float4*>(&Bs[buf][(rl + 32) * kLD + 4 * c4]) = rb1;        \
  if constexpr (B4 == 4) {                                                     \
    *reinterpret_cast<float4*>(&Bs[buf][(rl + 64) * kLD + 4 * c4]) = rb2;      \
    *reinterpret_cast<float4*>(&Bs[buf][(rl + 96) * kLD + 4 * c4]) = rb3;      \
  }
  f32x16 acc[2][TN];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x16{0};
  DG_NT_FETCH(0)
  const int chunks = K / kKC;
  for (int c = 0; c < chunks; ++c) {
    const int buf = DG_NT_BUFS == 2 ? (c & 1) : 0;
    if (DG_NT_BUFS == 1 && c > 0) __syncthreads();  // single panel: its readers of the previous chunk must be done
    DG_NT_STASH(buf)
    __syncthreads();  // also orders the reuse of this buffer: its readers of two chunks ago passed the last barrier
    if (c + 1 < chunks) {
      DG_NT_FETCH((c + 1) * kKC)
    }
    const float* ap = &As[buf][(wr * 64 + j) * kLD + h * 16];
    const float* bp = &Bs[buf][(wc * WN + j) * kLD + h * 16];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float4 fa[2], fb[TN];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = *reinterpret_cast<const float4*>(ap + a * 32 * kLD + 4 * v);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[b] = *reinterpret_cast<const float4*>(bp + b * 32 * kLD + 4 * v);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
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
#undef DG_NT_FETCH
#undef DG_NT_STASH
}

// ---- P[chunk][n, k] = sum over the chunk's rows of Y[r, n] * X[r, k] ----------------------------------------------------
// Y [R, Nout] (ldy), X [R, K] (ldx).  Block tile 128 (n) x BK (k) outputs, the rows of chunk blockIdx.z in steps of
// 32 through LDS; the MFMA reduction index is the row.  grid = (Nout / 128 rounded up, K / BK, row chunks);
// part [chunks][Nout][K].  Chunks entirely past R write zeros (the second stage adds all chunks in order).
#ifndef DG_TN_BUFS  // A/B knob like DG_NT_BUFS
#define DG_TN_BUFS 2
#endif
template <int BK>
__global__ __launch_bounds__(kGT, DG_TN_BUFS == 2 ? 2 : 3) void gemm_tn_kernel(const float* __restrict__ Y, int ldy, int Nout,
                                                         const float* __restrict__ X, int ldx, int K,
                                                         float* __restrict__ part, int rows_per_chunk,
                                                         const int* __restrict__ hdr, int rows) {
  constexpr int BNT = 128;
  constexpr int RC = 32;            // rows per LDS step
  constexpr int WK = BK / 2;        // k columns per wave
  constexpr int TK = WK / 32;       // 1 or 2
  constexpr int Y4 = RC * BNT / 4 / kGT;  // 4
  constexpr int X4 = RC * BK / 4 / kGT;   // 2 or 4
  __shared__ __attribute__((aligned(16))) float Ys[DG_TN_BUFS][RC * BNT];
  __shared__ __attribute__((aligned(16))) float Xs[DG_TN_BUFS][RC * BK];
  const int R = hdr != nullptr ? hdr[1] : rows;
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
  const int wn = wave >> 1, wk = wave & 1;  // wave tile: n rows wn*64.., k columns wk*WK..
  float4 ry[Y4], rx[X4];
  auto fetch = [&](long long r) {
#pragma unroll
    for (int i = 0; i < Y4; ++i) {
      const int e = threadIdx.x + i * kGT, row = e / (BNT / 4), c4 = e % (BNT / 4);
      const bool ok = r + row < re && n0 + 4 * c4 < Nout;  // clamped address + select: no conditional load
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
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < Y4; ++i) reinterpret_cast<float4*>(Ys[buf])[threadIdx.x + i * kGT] = ry[i];
#pragma unroll
    for (int i = 0; i < X4; ++i) reinterpret_cast<float4*>(Xs[buf])[threadIdx.x + i * kGT] = rx[i];
  };
  f32x16 acc[2][TK];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = f32x16{0};
  if (rb < re) {
    fetch(rb);
    int it = 0;
    for (long long r = rb; r < re; r += RC, ++it) {
      const int buf = DG_TN_BUFS == 2 ? (it & 1) : 0;
      if (DG_TN_BUFS == 1 && it > 0) __syncthreads();  // single panel: the previous step's readers must be done
      stash(buf);
      __syncthreads();
      if (r + RC < re) fetch(r + RC);
      const float* yp = &Ys[buf][h * BNT + wn * 64 + j];
      const float* xp = &Xs[buf][h * BK + wk * WK + j];
#pragma unroll
      for (int s = 0; s < RC / 2; ++s) {  // MFMA step s reduces rows 2s (lanes 0-31) and 2s+1 (lanes 32-63)
        float fy[2], fx[TK];
#pragma unroll
        for (int a = 0; a < 2; ++a) fy[a] = yp[2 * s * BNT + 32 * a];
#pragma unroll
        for (int b = 0; b < TK; ++b) fx[b] = xp[2 * s * BK + 32 * b];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < TK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fy[a], fx[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  float* out = part + (long long)bz * Nout * K;
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

// second stage of the weight gradient: out[e] = sum_chunk part[chunk][e], in a FIXED order (deterministic): a block of
// 256 threads owns EPB consecutive elements; 256 / EPB slices of the chunk range are summed side by side (eight
// independent partial sums each, so the loads pipeline) and combined in slice order.
template <int EPB>
__device__ __forceinline__ void tn_reduce_block(const float* __restrict__ part, int chunks, long long elems,
                                                float* __restrict__ out, int block) {
  constexpr int SL = 256 / EPB;
  __shared__ float red[SL][EPB];
  const int el = threadIdx.x % EPB, sl = threadIdx.x / EPB;
  const long long e = (long long)block * EPB + el;
  const int per = (chunks + SL - 1) / SL;
  const int c0 = sl * per, c1 = c0 + per < chunks ? c0 + per : chunks;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (e < elems) {
    const float* p = part + e;
    int c = c0;
    for (; c + 8 <= c1; c += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += p[(long long)(c + u) * elems];
    }
    for (; c < c1; ++c) a[0] += p[(long long)c * elems];
  }
  red[sl][el] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (sl == 0 && e < elems) {
    float t = red[0][el];
#pragma unroll
    for (int q = 1; q < SL; ++q) t += red[q][el];
    out[e] = t;
  }
}
template <int EPB>
static __global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ part, int chunks,
                                                                    long long elems, float* __restrict__ out) {
  tn_reduce_block<EPB>(part, chunks, elems, out, (int)blockIdx.x);
}
// two tables in one launch: blocks [0, blocks_a) reduce table a, the others table b (an MLP layer's weight gradient and
// the per-tile column sums of its bias gradient)
static __global__ __launch_bounds__(256) void gemm_tn_reduce2_kernel(const float* __restrict__ pa, int chunks_a,
                                                                     long long elems_a, float* __restrict__ out_a,
                                                                     int blocks_a, const float* __restrict__ pb,
                                                                     int chunks_b, long long elems_b,
                                                                     float* __restrict__ out_b) {
  if ((int)blockIdx.x < blocks_a) tn_reduce_block<32>(pa, chunks_a, elems_a, out_a, (int)blockIdx.x);
  else tn_reduce_block<32>(pb, chunks_b, elems_b, out_b, (int)blockIdx.x - blocks_a);
}

// few elements and many chunks (the 128 x 4 first-layer gradient: thousands of row tiles) -> more slices per element
static inline void launch_tn_reduce(const float* part, int chunks, long long elems, float* out, hipStream_t s) {
  if (elems <= 4096)
    hipLaunchKernelGGL(gemm_tn_reduce_kernel<8>, dim3((unsigned)((elems + 7) / 8)), dim3(256), 0, s, part, chunks, elems, out);
  else
    hipLaunchKernelGGL(gemm_tn_reduce_kernel<32>, dim3((unsigned)((elems + 31) / 32)), dim3(256), 0, s, part, chunks, elems,
                       out);
}

}  // namespace dg
