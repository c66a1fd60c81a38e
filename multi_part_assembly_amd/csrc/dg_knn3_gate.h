// k-nearest-neighbour graph of 3-d points (EdgeConv stage 1) with the pair work on the bf16 matrix cores — gfx950.
//
// dg_knn.h's knn3_kernel scores all N^2 pairs of a cloud with the pinned arithmetic and gates a sorted 20-list per
// lane: 0.37-0.40 ms at 353 x 1000 points, ~175 issue cycles per 64 pairs.  Here (the structure of gate_nn.hip, with
// k = 20 instead of 1) ONE v_mfma_f32_32x32x16_bf16 per 32 x 32 pairs and half a VALU operation per pair BOUND the
// scores; the pinned arithmetic and the sorted list see only the cells that can hold a neighbour:
//
//   operands   y = x - c (c = mean of the cloud's first 16 points), y = h + l + r in bf16 pieces.
//              Candidate row (16 bf16):  hx hy hz | hx hy hz | lx ly lz | m (3 pieces) | 0 0 0 0    m = fp32 |y|^2
//              Query column:             -2h      | -2l      | -2h      | 1 1 1        | 0 0 0 0
//              a(i,j) = m_j - 2 (h_i.h_j + l_i.h_j + h_i.l_j)  ~  |x_i - x_j|^2 - M_i.   With D(i,j) = -score(i,j), the
//              PINNED score of dg_knn.h (fma chain on the raw coordinates, norms as sums of rounded squares):
//                  | a(i,j) - (D(i,j) - M_i) |  <=  kappa (M_i + M_j) + kappa_raw (n_i + n_j)  <=  E_i          ... (*)
//              E_i = kappa (M_i + Mmax) + kappa_raw (n_i + nmax); kappa = 6e-5 as derived in gate_nn.hip (its "pinned chain"
//              term is replaced by kappa_raw: the fma chain of three terms is within 3 * 2^-24 sum |x x|, doubled; the two
//              norms 3 * 2^-24 each; forming the score rounds twice on <= 2 (n_i + n_j): 11 * 2^-24 -> kappa_raw = 1e-6).
//   bound      a lane keeps the minimum of a over each of its cells (16 rows of a tile: 8 x v_min3) in a register.  The
//              owner of a query collects its 64 cell minima (32 from the lane that shares its column); c20 = the 20th
//              smallest of them is attained by 20 distinct candidates, so by (*) the 20th smallest D is at most
//              c20 + M_i + E_i, and every true neighbour (boundary ties included) sits in a cell whose minimum is
//              <= c20 + 2 E_i.
//   answer     the owner walks those cells in ascending candidate order, scores their candidates with the pinned
//              arithmetic from fp32 coordinates in LDS, and feeds dg_knn.h's gated queue + sorted list, the gate starting
//              just below -(c20 + M_i + E_i) instead of -inf.  Same list, same tie rule, same output — index for index.
//   Magnitudes beyond 1e30 or non-finite norms switch the bound off for the block (threshold +inf: every cell is walked).
#pragma once

#include <hip/hip_runtime.h>

#include "dg_knn.h"
#include "dg_knn_fast.h"
#include "gate_common.h"

namespace dg {

constexpr float kK3Kappa = 6.0e-5f;
constexpr float kK3KappaRaw = 1.0e-6f;
constexpr int kK3QN = 20;   // queue slots per lane (the queues live in the panel's 32 KB once the bound is done)
constexpr int kK3R = 8;     // candidates between two queue checks

__device__ __forceinline__ float k3_sticky_max(float a, float b) { return b > a || b != b ? b : a; }  // NaN wins

// x4 [R][4] (xyz0), idx [R][20].  grid = (ceil(N / 256), DG_KNN_GRID_Y(parts)), block 256: thread = query.
template <typename IdxT>
__global__ __launch_bounds__(256, 3) void knn3_gate_kernel(const float* __restrict__ x4, int N, IdxT* __restrict__ idx,
                                                           const int* __restrict__ hdr) {
  constexpr int NT = kMaxN / 32;
  // x, y, z, |p|^2 (pinned); NaN past the cloud's end.  Row r sits at r + r / 32: in the answer phase every lane reads its
  // OWN tile, and tiles 512 bytes apart would all fall on the same four banks (a 64-way conflict, measured: 3x the kernel)
  __shared__ __attribute__((aligned(16))) float4 pts[kMaxN + kMaxN / 32];
  __shared__ __attribute__((aligned(16))) uint4 panel[2][kMaxN];  // bf16 rows, plane = k-half; later the waves' queues
  __shared__ float redm[4], redn[4];
  static_assert(kK3QN * 64 * 6 <= 8192 && NT * 64 * 4 <= 8192, "a wave's scratch is a quarter of the panel");
  int v, qb;
  knn_block(v, qb);
  if (v >= hdr[0]) return;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const float4* xp = reinterpret_cast<const float4*>(x4) + (long long)v * N;
  // centre: mean of the first 16 points (N >= 20), the same value in every lane
  float cx = 0.0f, cy = 0.0f, cz = 0.0f;
  for (int t = 0; t < 16; ++t) {
    const float4 p = xp[t];
    cx += p.x, cy += p.y, cz += p.z;
  }
  cx *= 0.0625f, cy *= 0.0625f, cz *= 0.0625f;
  // ---- stage the cloud: pinned rows for the answer, bf16 rows for the bound ----------------------------------------------
  float mmax = 0.0f, nmax = 0.0f;
  const float nanv = __builtin_nanf("");
  for (int r = threadIdx.x; r < kMaxN; r += 256) {
    float yx = 0.0f, yy = 0.0f, yz = 0.0f, m = 3.0e38f;  // rows past the cloud: never a minimum
    float4 rw = {nanv, nanv, nanv, nanv};
    if (r < N) {
      rw = xp[r];
      rw.w = (rw.x * rw.x + rw.y * rw.y) + rw.z * rw.z;
      nmax = k3_sticky_max(nmax, rw.w);
      yx = rw.x - cx, yy = rw.y - cy, yz = rw.z - cz;
      m = (yx * yx + yy * yy) + yz * yz;
      mmax = k3_sticky_max(mmax, m);
    }
    uint4 p0, p1;
    mpa::gate::target_row(yx, yy, yz, m, p0, p1);
    pts[r + (r >> 5)] = rw;
    panel[0][r] = p0;
    panel[1][r] = p1;
  }
  {
    float a = mmax, b = nmax;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a = k3_sticky_max(a, __shfl_xor(a, off, 64));
      b = k3_sticky_max(b, __shfl_xor(b, off, 64));
    }
    if (lane == 0) redm[wave] = a, redn[wave] = b;
  }
  __syncthreads();
  mmax = k3_sticky_max(k3_sticky_max(redm[0], redm[1]), k3_sticky_max(redm[2], redm[3]));
  nmax = k3_sticky_max(k3_sticky_max(redn[0], redn[1]), k3_sticky_max(redn[2], redn[3]));

  // ---- this lane's query (thread = query; lanes past N shadow the last point and never store) ------------------------------
  const int qi = qb * 256 + (int)threadIdx.x, qc = qi < N ? qi : N - 1;
  const float4 me = pts[qc + (qc >> 5)];
  float mq;
  uint4 bq[2];
  {
    const float yx = me.x - cx, yy = me.y - cy, yz = me.z - cz;
    mq = (yx * yx + yy * yy) + yz * yz;
    uint4 k0, k1;
    mpa::gate::query_column(yx, yy, yz, k0, k1);
    mpa::gate::wave_columns(k0, k1, j, h, bq);
  }

  // ---- bound: cell minima.  tmA = cells of THIS lane's own query (tile h of the wave, rows of lane half h), tmB = cells of the
  // partner's query (tile 1 - h, same rows); swapped below -----------------------------------------------------------------------
  float tmA[NT], tmB[NT];
  {
    const int nti = (N + 31) / 32;
    const uint4* pl = &panel[h][j];
#pragma unroll
    for (int t8 = 0; t8 < NT; t8 += 8) {
      if (t8 < nti) {  // (wave-uniform; the panel is padded to kMaxN rows)
        uint4 nx = pl[32 * t8];
#pragma unroll
        for (int t = t8; t < t8 + 8; ++t) {
          const mpa::gate::bf16x8 a = mpa::gate::as_bf16x8(nx);
          if (t + 1 < t8 + 8) nx = pl[32 * (t + 1)];
          const f32x16 z = {0};
          const f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, mpa::gate::as_bf16x8(bq[0]), z, 0, 0, 0);
          const f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, mpa::gate::as_bf16x8(bq[1]), z, 0, 0, 0);
          float x0, y0, x1, y1;
          mpa::gate::min16(acc0, x0, y0);
          mpa::gate::min16(acc1, x1, y1);
          const float c0 = mpa::gate::min3(x0, y0, __builtin_inff()), c1 = mpa::gate::min3(x1, y1, __builtin_inff());
          tmA[t] = h ? c1 : c0;
          tmB[t] = h ? c0 : c1;
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int t = t8; t < t8 + 8; ++t) tmA[t] = tmB[t] = __builtin_inff();
      }
    }
  }
  // the partner lane (same column, other half of the rows) holds the other 32 cells of this lane's query in ITS tmB
#pragma unroll
  for (int t = 0; t < NT; ++t) tmB[t] = __shfl_xor(tmB[t], 32, 64);
  // Every wave is done with the panel: its memory becomes wave-private scratch (8 KB each) — first the received cell
  // minima (the sort below needs their registers), then the queues of the answer phase.
  __syncthreads();
  float* wmem = reinterpret_cast<float*>(&panel[0][0]) + wave * 2048;
#pragma unroll
  for (int t = 0; t < NT; ++t) wmem[t * 64 + lane] = tmB[t];
  // c20: the 20th smallest of the 64 cell minima = -(20th largest of the negated values); Batcher sort of each half,
  // then the 20th of the union as max_i min(a_(i), b_(20 - i)) (dg_knn_fast.h's merge)
  float c20;
  {
    float a[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      a[t] = -tmA[t];
      tmB[t] = -tmB[t];
    }
    kf_sort_desc<NT>(a);
    kf_sort_desc<NT>(tmB);
    float kth = __builtin_fmaxf(a[19], tmB[19]);
#pragma unroll
    for (int i = 1; i <= 19; ++i) kth = __builtin_fmaxf(kth, __builtin_fminf(a[i - 1], tmB[20 - i - 1]));
    c20 = -kth;
  }
  // thresholds, every step rounded towards "keep more"; risky magnitudes: no pruning at all
  const bool safe = mq <= 1e30f && mmax <= 1e30f && nmax <= 1e30f && me.w <= 1e30f;  // (false for NaNs)
  float E = mpa::gate::next_up(mpa::gate::next_up(mpa::gate::next_up(mq + mmax) * kK3Kappa) + mpa::gate::next_up(mpa::gate::next_up(me.w + nmax) * kK3KappaRaw));
  E = mpa::gate::next_up(E + 1e-30f);
  const float thr = safe ? mpa::gate::next_up(c20 + mpa::gate::next_up(2.0f * E)) : __builtin_inff();
  // 20th smallest D <= c20 + M_i + E  =>  20th best score >= -(that); the gate starts strictly below it
  const float dub = mpa::gate::next_up(mpa::gate::next_up(c20 + mpa::gate::next_up(mq * (1.0f + 1e-6f))) + E);
  const float sthr = safe ? prev_float(prev_float(-dub)) : -__builtin_inff();
  unsigned kOwn = 0u, kPar = 0u;
#pragma unroll
  for (int t = NT - 1; t >= 0; --t) {
    kOwn = (kOwn << 1) | (tmA[t] <= thr ? 1u : 0u);
    kPar = (kPar << 1) | (wmem[t * 64 + lane] <= thr ? 1u : 0u);
  }
  {  // only tiles that hold points
    const int nti = (N + 31) / 32;
    const unsigned real = nti >= 32 ? 0xffffffffu : (1u << nti) - 1u;
    kOwn &= real;
    kPar &= real;
  }
  // cells of row half 0 / 1 of every tile (ascending candidate order inside a tile: rows 8g + 0..3 are half 0, 8g + 4..7 half 1)
  const unsigned kLo = h ? kPar : kOwn, kHi = h ? kOwn : kPar;

  // ---- answer: the pinned scores of the qualifying cells through the gated queue + sorted list of dg_knn.h ---------------------
  float* qs_w = wmem;                                                        // [kK3QN][64] floats
  unsigned short* qj_w = reinterpret_cast<unsigned short*>(wmem + kK3QN * 64);  // [kK3QN][64] u16
  Best b;
  best_init(b);
  Queue q{qs_w, qj_w, lane, 0, sthr};
  unsigned tiles = kLo | kHi;
  while (__ballot(tiles != 0u)) {
    const bool act = tiles != 0u;
    const int t = act ? __builtin_ctz(tiles) : 0;
    tiles &= tiles - 1u;  // (0 stays 0)
    const bool lo = act && ((kLo >> t) & 1u), hi = act && ((kHi >> t) & 1u);
    const float ninf = -__builtin_inff();
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int row = 32 * t + 8 * g4;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 p = pts[33 * t + 8 * g4 + u];
        const float dot = __builtin_fmaf(me.z, p.z, __builtin_fmaf(me.y, p.y, me.x * p.x));
        const float s = (-p.w + 2.0f * dot) - me.w;
        queue_push(q, (u < 4 ? lo : hi) ? s : ninf, row + u);  // (rows past N: NaN scores, never pushed)
      }
      if (DG_QUEUE_FULL_R(q, kK3QN, kK3R)) queue_flush<false>(q, b);
    }
  }
  queue_flush<false>(q, b);
  if (qi < N) {
    IdxT* out = idx + ((long long)v * N + qi) * kNbr;
#pragma unroll
    for (int t = 0; t < kNbr; ++t) out[t] = (IdxT)b.j[t];
  }
}

}  // namespace dg
