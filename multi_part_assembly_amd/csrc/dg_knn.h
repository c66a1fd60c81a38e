// k-nearest-neighbour graph (k = 20) of every part in C-dimensional feature space — gfx950.
//
// Replaces `knn` (multi_part_assembly/models/modules/encoder/dgcnn.py:8-15): the reference materialises the
// [n, N, N] score matrix  -|x_j|^2 + 2 x_i.x_j - |x_i|^2  and calls topk.  Here no score is ever stored.
//
// SCORE ARITHMETIC (pinned op by op; oracle/knn_ref.c restates it; tests require index-exact agreement):
//   C = 3   the reference's own CPU arithmetic, verified bit for bit against torch on the fixture cloud:
//             dot  = fma(x2, y2, fma(x1, y1, x0 * y0))           (the BLAS micro-kernel's k-ascending FMA chain)
//             |x|^2 = (x0*x0 + x1*x1) + x2*x2                    (torch.sum(x**2): products rounded, then added)
//   C >= 64 no reference order exists (a blocked BLAS Gram matrix); defined by the matrix-core chain, which equals
//           a scalar fmaf chain bit for bit (v_mfma_f32_32x32x2_f32 applies its two k values in lane-half order):
//             dot  = fmaf chain over k in the order 0, C/2, 1, C/2+1, ..., C/2-1, C-1, starting from 0
//             |x|^2 = the same chain with y = x
//   score(i, j) = (-|x_j|^2 + 2 * dot(i, j)) - |x_i|^2     every operation rounded to fp32 (dgcnn.py:11-13 order)
//   neighbours of i = the k best (score descending, index ascending among equal scores), best first.
//
// SELECTION.  A lane keeps a sorted (score, index) list of its 20 best in registers.  Inserting costs ~100 VALU
// slots, and 64 independent streams share a wave, so inserting whenever ANY lane needs it would run the insertion
// for almost every candidate.  Instead every lane appends the candidates that beat its current 20th best to a small
// private queue in LDS (slot-major: lane l always hits bank l, conflict-free) and the wave flushes all queues
// together when one of them is about to overflow: the flush loop runs max(queue length) insertions with most lanes
// busy, and the threshold (20th best) is refreshed after every flush.
#pragma once

#include <hip/hip_runtime.h>

#include "dg_gemm.h"

namespace dg {

constexpr int kNbr = 20;   // neighbours per point (the reference's k)
constexpr int kMaxN = 1024;

struct Best {
  float s[kNbr];
  int j[kNbr];
};

__device__ __forceinline__ void best_init(Best& b) {
#pragma unroll
  for (int t = 0; t < kNbr; ++t) {
    b.s[t] = -__builtin_inff();
    b.j[t] = 0;
  }
}

// insert (s0, j0) into the sorted list.  LEX = false: candidates arrive in ascending index order, so a strict `>` on
// the score keeps the earlier index in front among equal scores; LEX = true: arbitrary arrival order (merging lists).
// With p_t = "the new element goes in front of slot t" (monotone along the sorted list):
//     new[t] = p_t ? (p_{t-1} ? old[t-1] : new element) : old[t]
// evaluated from the last slot upwards, in place, with no carried element (a swap-through bubble costs a dependent
// chain and a register shuffle per slot).  For the scores this is the median of (old[t-1], old[t], s0): one
// v_med3_f32 per slot; the indices take one compare and two selects.
template <bool LEX>
__device__ __forceinline__ void best_insert(Best& b, float s0, int j0) {
  auto before = [&](int t) { return LEX ? (s0 > b.s[t] || (s0 == b.s[t] && j0 < b.j[t])) : s0 > b.s[t]; };
  bool pt = before(kNbr - 1);
#pragma unroll
  for (int t = kNbr - 1; t > 0; --t) {
    const bool pm = before(t - 1);
    const int inner = pm ? b.j[t - 1] : j0;
    b.j[t] = pt ? inner : b.j[t];
    b.s[t] = __builtin_amdgcn_fmed3f(b.s[t - 1], b.s[t], s0);
    pt = pm;
  }
  b.j[0] = pt ? j0 : b.j[0];
  b.s[0] = s0 > b.s[0] ? s0 : b.s[0];
}

// the next representable float below x (x = -inf and NaN are returned unchanged)
__device__ __forceinline__ float prev_float(float x) {
  const int b = __float_as_int(x);
  const int stepped = b > 0 ? b - 1 : b + 1;                         // towards -inf in both half-lines
  const int zero_fix = x == 0.0f ? (int)0x80000001u : stepped;       // below +-0: the smallest negative denormal
  const bool keep = x != x || x == -__builtin_inff();
  return __int_as_float(keep ? b : zero_fix);
}

// per-wave queues in LDS: qs[slot][lane], qj[slot][lane]
struct Queue {
  float* qs;
  unsigned short* qj;
  int lane;
  int cnt;
  float thr;
};
// a queue of QN slots takes R candidates per round: flush as soon as fewer than R slots are left
#define DG_QUEUE_FULL_R(q, QN, R) __any((q).cnt > (QN) - (R))
#define DG_QUEUE_FULL(q, QN) DG_QUEUE_FULL_R(q, QN, 16)
// Tuning knobs (tools/probes/knn_time.hip builds variants).  The queues are what limits the blocks per CU: 20 slots
// checked every 8 candidates flush as full as 32 slots checked every 16, and leave room for a third block.
#ifndef DG_QN64   // queue slots per lane, C = 64 Gram kernel
#define DG_QN64 20
#endif
#ifndef DG_QN128
#define DG_QN128 24
#endif
#ifndef DG_GPC64  // candidate groups (of 4) between two queue checks
#define DG_GPC64 2
#endif
#ifndef DG_GPC128
#define DG_GPC128 4
#endif
#ifndef DG_BPC64  // blocks per CU the C = 64 kernel is compiled for
#define DG_BPC64 3
#endif
#ifndef DG_QN3    // C = 3 kernel: queue slots and candidates per check
#define DG_QN3 16
#endif
#ifndef DG_CPC3
#define DG_CPC3 4
#endif
#ifndef DG_T3     // C = 3 kernel: threads (= queries) per block.  tools/probes/knn3_time.hip at 353 x 1000 points, (T3, QN3, CPC3):
#define DG_T3 256 // (256, 16, 4) 0.354 ms, (512, 16, 4) 0.349, (128, 16, 4) 0.466, (64, 16, 4) 0.571, (256, 24, 8) 0.365, (256, 32, 8) 0.458
#endif

__device__ __forceinline__ void queue_push(Queue& q, float s, int idx) {
  if (s > q.thr) {
    q.qs[q.cnt * 64 + q.lane] = s;
    q.qj[q.cnt * 64 + q.lane] = (unsigned short)idx;
    ++q.cnt;
  }
}

// PAIR: lanes l and l + 32 serve the same query, each holding the best 20 of HALF of the candidates seen so far.  The
// 20th best of the union is at least max(a20, b20) and at least min(a10, b10) — a much tighter gate than a lane's own
// 20th best.  Scores strictly below it can be dropped; equal scores must stay (the index decides), hence prev_float.
template <bool PAIR>
__device__ __forceinline__ void queue_flush(Queue& q, Best& b) {
  int n = q.cnt;  // wave-uniform trip count (a data-dependent exit test inside the loop makes the compiler copy the
#pragma unroll    // whole 40-register list twice per pass)
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(n, off, 64);
    n = o > n ? o : n;
  }
  n = __builtin_amdgcn_readfirstlane(n);
  for (int e = 0; e < n; ++e) {
    const float raw = q.qs[e * 64 + q.lane];  // read unconditionally (always inside the queue), then select
    const int cj = q.qj[e * 64 + q.lane];
    const float cs = e < q.cnt ? raw : -__builtin_inff();
    best_insert<false>(b, cs, cj);
  }
  q.cnt = 0;
  q.thr = b.s[kNbr - 1];
  if constexpr (PAIR) {
    const float a20 = b.s[kNbr - 1], a10 = b.s[kNbr / 2 - 1];
    const float o20 = __shfl_xor(a20, 32, 64), o10 = __shfl_xor(a10, 32, 64);
    const float lo = a10 < o10 ? a10 : o10;
    float th = a20 > o20 ? a20 : o20;
    th = prev_float(th > lo ? th : lo);
    q.thr = th > a20 ? th : a20;
  }
}

// Block -> (cloud, query block).  Workgroups go to the 8 XCDs round-robin and every XCD has its own L2; the query blocks
// of one cloud all stream the same candidate rows, so they are given to ONE XCD: linear block id L runs on XCD L % 8 and
// takes cloud (L / 8 / Q) * 8 + L % 8, query block (L / 8) % Q.  grid.y must be a multiple of 8 (DG_KNN_GRID_Y).
#ifndef DG_KNN_XCD
#define DG_KNN_XCD 1
#endif
#define DG_KNN_GRID_Y(n) ((unsigned)(((n) + 7) / 8 * 8))
__device__ __forceinline__ void knn_block(int& v, int& qb) {
#if DG_KNN_XCD
  const int Q = (int)gridDim.x, L = (int)blockIdx.y * Q + (int)blockIdx.x;
  const int xcd = L & 7, k = L >> 3;
  v = (k / Q) * 8 + xcd;
  qb = k % Q;
#else
  v = (int)blockIdx.y;
  qb = (int)blockIdx.x;
#endif
}

// ---- C = 3 -------------------------------------------------------------------------------------------------------------
// x4 [R][4] (xyz0), idx [R][20] u16.  grid = (ceil(N / 256), parts), block 256: lane = query; the part's points
// (+ their norms) sit in LDS and are read as broadcasts.
template <typename IdxT>
__global__ __launch_bounds__(DG_T3) void knn3_kernel(const float* __restrict__ x4, int N, IdxT* __restrict__ idx,
                                                     const int* __restrict__ hdr) {
  __shared__ __attribute__((aligned(16))) float4 pts[kMaxN];  // x, y, z, |p|^2
  constexpr int QN = DG_QN3, CPC = DG_CPC3;
  __shared__ float qs_[DG_T3 / 64][QN * 64];
  __shared__ unsigned short qj_[DG_T3 / 64][QN * 64];
  int v, qb;
  knn_block(v, qb);
  if (v >= hdr[0]) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float4* xp = reinterpret_cast<const float4*>(x4) + (long long)v * N;
  for (int p = threadIdx.x; p < N; p += DG_T3) {
    float4 t = xp[p];
    t.w = (t.x * t.x + t.y * t.y) + t.z * t.z;
    pts[p] = t;
  }
  __syncthreads();
  const int qi = qb * DG_T3 + threadIdx.x, qc = qi < N ? qi : N - 1;
  const float4 me = pts[qc];
  Best b;
  best_init(b);
  Queue q{qs_[wave], qj_[wave], lane, 0, -__builtin_inff()};
  for (int j0 = 0; j0 < N; j0 += CPC) {
#pragma unroll
    for (int u = 0; u < CPC; ++u) {
      const int j = j0 + u;
      if (j < N) {  // wave-uniform
        const float4 t = pts[j];
        const float dot = __builtin_fmaf(me.z, t.z, __builtin_fmaf(me.y, t.y, me.x * t.x));
        queue_push(q, (-t.w + 2.0f * dot) - me.w, j);
      }
    }
    if (DG_QUEUE_FULL_R(q, QN, CPC)) queue_flush<false>(q, b);
  }
  queue_flush<false>(q, b);
  if (qi < N) {
    IdxT* out = idx + ((long long)v * N + qi) * kNbr;
#pragma unroll
    for (int t = 0; t < kNbr; ++t) out[t] = (IdxT)b.j[t];
  }
}

// ---- row norms in the matrix-core chain order -----------------------------------------------------------------------------
// x [R][C] (ld), norm [R].  One thread per row (the chain is sequential by definition); rows are staged through LDS in
// 16 + 16 column slabs so that the global reads stay coalesced.  grid = ceil(Rmax / 256), block 256.
template <int C>
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ x, int ld, float* __restrict__ norm,
                                                      const int* __restrict__ hdr) {
  __shared__ float slab[256][33];
  const int R = hdr[1];
  const long long r0 = (long long)blockIdx.x * 256;
  if (r0 >= R) return;
  float acc = 0.0f;
  const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;  // float4 column (8 per 32-float slab row), rows rl + 32 i
  for (int u = 0; u < C / 32; ++u) {  // chain positions 32u .. 32u+31 = columns 16u+s (even) and C/2+16u+s (odd)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long r = r0 + rl + 32 * i;
      const int col = (c4 < 4 ? 16 * u + 4 * c4 : C / 2 + 16 * u + 4 * (c4 - 4));
      const float4 t = r < R ? *reinterpret_cast<const float4*>(x + r * ld + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      float* d = &slab[rl + 32 * i][4 * c4];
      d[0] = t.x;
      d[1] = t.y;
      d[2] = t.z;
      d[3] = t.w;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float lo = slab[threadIdx.x][s], hi = slab[threadIdx.x][16 + s];
      acc = __builtin_fmaf(lo, lo, acc);
      acc = __builtin_fmaf(hi, hi, acc);
    }
  }
  if (r0 + threadIdx.x < R) norm[r0 + threadIdx.x] = acc;
}

// ---- C = 64 / 128: Gram tiles on the matrix cores ---------------------------------------------------------------------------
// x [R][C] (ld), norm [R], idx [R][20].  grid = (ceil(N / 128), parts), block 256 = 4 waves, a wave owns 32 queries
// (B operand, register-resident); candidate tiles of 32 rows are staged through a double-buffered LDS panel shared by
// the four waves.  Accumulator layout: lane (j, h) holds, for query j, the candidates acc_row(r, h) of the tile — two
// lanes per query, each seeing half of the candidates; their lists are merged at the end.
// MODE (timing probes only, tools/probes/knn_time.hip): 0 = the real kernel; 1 = Gram tiles only (scores summed, no
// selection); 2 = gate + queue pushes but no list maintenance (the queue is simply emptied when full).
template <int C, typename IdxT, int MODE = 0>
__global__ __launch_bounds__(256, C > 64 ? 2 : DG_BPC64) void knn_mfma_kernel(const float* __restrict__ x, int ld,
                                                          const float* __restrict__ norm, int N,
                                                          IdxT* __restrict__ idx, const int* __restrict__ hdr,
                                                          const int* __restrict__ flags = nullptr) {
  constexpr int KH = C / 2, LD = C + 4, T4 = 32 * C / 4 / 256;  // float4 per thread and candidate tile
  constexpr int QN = C > 64 ? DG_QN128 : DG_QN64;  // the blocks of a CU must fit in its 160 KB of LDS
  constexpr int GPC = C > 64 ? DG_GPC128 : DG_GPC64;
  __shared__ __attribute__((aligned(16))) float tile[2][32 * LD];
  __shared__ __attribute__((aligned(16))) float tnorm[2][32];
  __shared__ float qs_[4][QN * 64];
  __shared__ unsigned short qj_[4][QN * 64];
  int v, qb;
  knn_block(v, qb);
  if (v >= hdr[0]) return;
  // as the safety net of the shortlist search (dg_knn_fast.h): only the 128-query blocks it flagged are recomputed
  if (flags != nullptr && flags[v * (int)gridDim.x + qb] == 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const float* xp = x + (long long)v * N * ld;
  const float* np_ = norm + (long long)v * N;
  const int q0 = qb * 128 + wave * 32;
  const int qrow = q0 + j < N ? q0 + j : N - 1;
  float bq[KH];
  {
    const float4* src = reinterpret_cast<const float4*>(xp + (long long)qrow * ld + h * KH);
#pragma unroll
    for (int w = 0; w < KH / 4; ++w) {
      const float4 t = src[w];
      bq[4 * w + 0] = t.x;
      bq[4 * w + 1] = t.y;
      bq[4 * w + 2] = t.z;
      bq[4 * w + 3] = t.w;
    }
  }
  const float qn = np_[qrow];
  Best b;
  best_init(b);
  Queue q{qs_[wave], qj_[wave], lane, 0, -__builtin_inff()};
  const int c4 = threadIdx.x % (C / 4), rl = threadIdx.x / (C / 4);  // staging role
  constexpr int RS = 256 / (C / 4);                                  // rows per staging pass
  float4 raw[T4];
  float rn = 0.0f;
  auto fetch = [&](int t) {
#pragma unroll
    for (int i = 0; i < T4; ++i) {
      const int row = t * 32 + rl + RS * i;
      raw[i] = row < N ? *reinterpret_cast<const float4*>(xp + (long long)row * ld + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (wave == 0 && lane < 32) rn = t * 32 + lane < N ? np_[t * 32 + lane] : 0.0f;
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < T4; ++i) *reinterpret_cast<float4*>(&tile[buf][(rl + RS * i) * LD + 4 * c4]) = raw[i];
    if (wave == 0 && lane < 32) tnorm[buf][lane] = rn;
  };
  const int tiles = (N + 31) / 32;
  fetch(0);
  for (int t = 0; t < tiles; ++t) {
    const int buf = t & 1;
    stash(buf);
    __syncthreads();
    if (t + 1 < tiles) fetch(t + 1);
    const float4* frag = reinterpret_cast<const float4*>(&tile[buf][j * LD + h * KH]);
    f32x16 acc = {0};
#pragma unroll
    for (int w = 0; w < KH / 4; ++w) {
      const float4 a = frag[w];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[4 * w + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[4 * w + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[4 * w + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[4 * w + 3], acc, 0, 0, 0);
    }
    // the lane's 16 candidates are rows 8g + 4h .. 8g + 4h + 3, g = 0..3: their norms come as one 16-byte read per group
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 n4 = *reinterpret_cast<const float4*>(&tnorm[buf][8 * g4 + 4 * h]);
      const float cn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int cand = t * 32 + 8 * g4 + 4 * h + u;
        const float s = cand < N ? (-cn[u] + 2.0f * acc[4 * g4 + u]) - qn : -__builtin_inff();
        if constexpr (MODE == 1) b.s[0] += s;
        else queue_push(q, s, cand);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the groups apart: hoisting all 16 scores costs 30+ registers
      if ((g4 + 1) % GPC == 0) {
        if constexpr (MODE == 2) {
          if (DG_QUEUE_FULL_R(q, QN, 4 * GPC)) {
            q.thr = q.qs[8 * 64 + lane];  // some plausible gate
            q.cnt = 0;
          }
        } else if constexpr (MODE == 0) {
          if (DG_QUEUE_FULL_R(q, QN, 4 * GPC)) queue_flush<(C <= 64)>(q, b);  // C = 128: the shared gate does not fit in 256 registers
        }
      }
    }
  }
  queue_flush<(C <= 64)>(q, b);
  // merge the two halves of every query: lanes 32-63 hand their lists to lanes 0-31 through the queue memory.  The
  // lane's coordinates are derived afresh (mbcnt) so that nothing but the lists has to stay in registers across the
  // tile loop — the C = 64 kernel sits right at the 168-register limit of three blocks per CU.
  const int l2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int j2 = l2 & 31;
  float* qs2 = qs_[wave];
  unsigned short* qj2 = qj_[wave];
  if (l2 >= 32) {
#pragma unroll
    for (int t = 0; t < kNbr; ++t) {
      qs2[t * 64 + j2] = b.s[t];
      qj2[t * 64 + j2] = (unsigned short)b.j[t];
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (l2 < 32) {
#pragma unroll 1
    for (int t = 0; t < kNbr; ++t) best_insert<true>(b, qs2[t * 64 + j2], qj2[t * 64 + j2]);
    const int qi = qb * 128 + wave * 32 + j2;
    if (qi < N) {
      IdxT* out = idx + ((long long)v * N + qi) * kNbr;
#pragma unroll
      for (int t = 0; t < kNbr; ++t) out[t] = (IdxT)b.j[t];
    }
  }
}

}  // namespace dg
