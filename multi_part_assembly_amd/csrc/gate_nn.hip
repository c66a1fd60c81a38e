// Exact nearest neighbour between two mid-sized clouds with the pair work on the bf16 matrix cores — gfx950.
//
// Serves (a) rot_points_cd_loss of the fused assembly loss (assembly_loss.hip: every valid part against its own
// ground-truth copy, reference utils/loss.py:113-138) and (b) the generic Chamfer operator for clouds of a few hundred
// to a few thousand points (chamfer.hip: the reference's per-part call [B*P, N, 3]^2, utils/chamfer/chamfer.py:9-24).
// Same results as the exhaustive scan of chamfer_core.h, bit for bit: d = (dx*dx + dy*dy) + dz*dz with every operation
// rounded, lowest target index on ties, (1e32, -1) for a query that sees nothing below 1e32.
//
// The exhaustive scan spends ~4.5 VALU operations per (query, target) pair.  Here the pair work is ONE
// v_mfma_f32_32x32x16_bf16 per 32 x 32 pairs plus half a VALU operation per pair (v_min3 trees); the pinned fp32
// arithmetic is spent only on the few CELLS (16 targets of one tile, as a lane's accumulator registers hold them) that
// can contain a query's answer:
//
//   operands   y = x - c (c: any vector, here the mean of the first targets) is cut into bf16 pieces y = h + l + r.
//              Target row (16 bf16):  hx hy hz | hx hy hz | lx ly lz | n (3 pieces) | 0 0 0 0      n = fp32 |y|^2, exact
//              Query column:          -2h      | -2l      | -2h      | 1 1 1        | 0 0 0 0      in three bf16 pieces
//              so one K = 16 product gives  a(i,j) = n_j - 2 (h_i.h_j + l_i.h_j + h_i.l_j)  ~  |y_j|^2 - 2 y_i.y_j
//              = d(i,j) - |y_i|^2, with   | a(i,j) - (d(i,j) - M_i) |  <=  kappa (M_i + M_j)          ... (*)
//              (d: the PINNED distance; M = |y|^2 in real arithmetic).
//   bound      every lane keeps the minimum of each of its cells (8 x v_min3 per 16 accumulator values) in a register,
//              and the running minimum tau_i = min_j a(i,j).  By (*) every target that attains the minimum of the pinned
//              distance has  a(i,j) <= tau_i + 2 kappa (M_i + Mmax)  =: thr_i   (Mmax: the largest target norm).
//   answer     a cell whose minimum is <= thr_i is evaluated exactly: pinned distances of its 16 targets from the fp32
//              coordinates (kept in LDS beside the bf16 panel), lexicographic (distance, index) update as one 64-bit
//              compare.  Typically 1-2 of a query's 64 cells qualify.  The two lanes that share a query (one per half of a
//              tile's rows) merge their keys at the end.
//   Magnitudes beyond 1e30 (overflowing squares) or non-finite norms send the wave to the textbook scan of all targets
//   in index order; NaN / inf coordinates of single points need nothing special (their a is NaN or +inf: never a
//   minimum; their pinned distance is NaN or inf: never below the running best).  The gate only ever DECIDES which
//   pairs get the exact evaluation, it never contributes a digit to a result.  A target cloud whose points all coincide
//   (the zero-padded parts of the reference's per-part call) is answered directly.
//
// kappa (per unit of M_i + M_j).  bf16 keeps 8 significant bits: |y - h| <= 2^-8 |y|, |l| <= 2^-8 |y|, |r| <= 2^-16 |y|
// per coordinate, so y_i.y_j - (hh + lh + hl) = l.l + r_i.y_j + (h + l)_i.r_j is at most
// 3.02 * 2^-16 sum_k |y_ik y_jk| <= 3.02 * 2^-16 (M_i + M_j) / 2; it enters a twice:              4.61e-5
// The 12 bf16 products are exact in fp32; their accumulation inside the matrix core is charged 2^-23 per term of a
// 16-term sum (the internal order and rounding are not documented) on sum |terms| <= 2.03 (M_i + M_j):  3.9e-6
// y = fl(x - c) moves |y_i - y_j|^2 away from |x_i - x_j|^2 by <= 4.04 * 2^-24 (M_i + M_j):          2.4e-7
// the pinned chain is within 6 * 2^-24 of the real |x_i - x_j|^2 <= 2 (M_i + M_j):                    7.2e-7
// the fp32 norms (three squares, two additions) are within 3 * 2^-24 of M, on either side:             3.6e-7
// Sum 5.13e-5; kappa = 6e-5.  The threshold is rounded up at every step and carries 1e-30 of absolute slack for
// products that underflow.
#include "assembly_internal.h"
#include "common.h"
#include "gate_common.h"

namespace mpa {
namespace {

using gate::f32x16;
typedef unsigned long long u64;

constexpr int kGW = 4;                 // waves per block: 256 queries
constexpr int kGQ = 64 * kGW;
constexpr int kGT = 1024;              // targets per LDS panel (32 KB of bf16 rows + 16 KB of fp32 coordinates)
constexpr int kGTiles = kGT / 32;
constexpr float kGKappa = 6.0e-5f;
constexpr float kGMaxNorm = 1e30f;     // beyond: squares may overflow -> the textbook scan

__device__ __forceinline__ float g_dist3(float dx, float dy, float dz) { return (dx * dx + dy * dy) + dz * dz; }
__device__ __forceinline__ u64 g_key(float d, unsigned idx) { return ((u64)__float_as_uint(d) << 32) | idx; }

struct GateArgs {
  const float* a;        // cloud A [M][na][3]
  const float* b;        // cloud B [M][nb][3]
  const float* valids;   // nullable [M]: pairs with valids[m] == 0 are skipped
  int na, nb, tiles;     // tiles = ceil(max(na, nb) / 256)
  // LOSS: idx32[dir] [M][n], tile_sums [2][M][tiles]; else dist[dir] [M][n], idx64[dir] [M][n]
  int* idx32[2];
  float* tile_sums;
  float* dist[2];
  long long* idx64[2];
  int M;
};

// grid = (M * tiles, 2), block 256.  blockIdx.y = direction (0: A's points are the queries).
template <bool LOSS>
__global__ __launch_bounds__(kGQ, 3) void gate_nn_kernel(const GateArgs g) {
  __shared__ uint4 panel[2][kGT];   // plane k-half h: row r -> 8 bf16
  // the same targets' fp32 coordinates (NaN past the cloud's end); row r at r + r / 32: in the answer phase every lane
  // reads its own tile, and tiles 512 bytes apart would all fall on the same banks
  __shared__ float4 raw[kGT + kGT / 32];
  __shared__ float red[kGW], redm[kGW];
  const int m = blockIdx.x / g.tiles, tile = blockIdx.x % g.tiles, dir = blockIdx.y;
  if (g.valids != nullptr && g.valids[m] == 0.0f) return;
  const int nq = dir == 0 ? g.na : g.nb, nt = dir == 0 ? g.nb : g.na;
  const int qbase = tile * kGQ;
  if (qbase >= nq) return;
  const float* qa = (dir == 0 ? g.a : g.b) + 3LL * m * nq;
  const float* tb = (dir == 0 ? g.b : g.a) + 3LL * m * nt;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int j = lane & 31, h = lane >> 5;
  const int qi = qbase + (int)threadIdx.x;
  const bool has = qi < nq;
  const int qc = has ? qi : nq - 1;
  const float X = qa[3LL * qc], Y = qa[3LL * qc + 1], Z = qa[3LL * qc + 2];
  float bd = 1e32f;  // chamfer_kernel.cu:60
  int bi = -1;

  auto store = [&]() {
    if (LOSS) {
      if (has) g.idx32[dir][(long long)m * nq + qi] = bi;
      float s = has ? bd : 0.0f;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) red[wave] = s;
      __syncthreads();
      if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kGW; ++w) t += red[w];
        g.tile_sums[((long long)dir * g.M + m) * g.tiles + tile] = t;
      }
    } else if (has) {
      g.dist[dir][(long long)m * nq + qi] = bd;
      g.idx64[dir][(long long)m * nq + qi] = (long long)bi;
    }
  };
  if (nt == 0) {  // (block-uniform)
    store();
    return;
  }

  // centre: the mean of the first targets (wave-uniform scalar loads, fixed order: the same value in every lane)
  float cx = 0.0f, cy = 0.0f, cz = 0.0f;
  {
    const int nc = nt < 16 ? nt : 16;
    for (int t = 0; t < nc; ++t) {
      cx += tb[3 * t];
      cy += tb[3 * t + 1];
      cz += tb[3 * t + 2];
    }
    const float inv = 1.0f / (float)nc;
    cx *= inv, cy *= inv, cz *= inv;
  }

  // ---- this lane's query as a column of the product, then the two query tiles of the wave --------------------------------
  // (tile s = queries 32 s + j of the wave; lane (j, h) supplies k-half h of column j and keeps that query's coordinates)
  uint4 bq[2];
  float OX, OY, OZ;  // the query of the OTHER tile that this lane serves (tile 1 - h, column j); tile h's is its own
  float mq;
  {
    const float yx = X - cx, yy = Y - cy, yz = Z - cz;
    mq = g_dist3(yx, yy, yz);
    uint4 k0, k1;
    gate::query_column(yx, yy, yz, k0, k1);
    gate::wave_columns(k0, k1, j, h, bq);
    OX = __shfl(X, 32 * (1 - h) + j, 64), OY = __shfl(Y, 32 * (1 - h) + j, 64), OZ = __shfl(Z, 32 * (1 - h) + j, 64);
  }

  // ---- target panel: rows chunk * kGT + r, r < kGT ------------------------------------------------------------------------------
  const float t0x = tb[0], t0y = tb[1], t0z = tb[2];
  bool same = true;
  float mmax = 0.0f;
  const float nanf_ = __builtin_nanf("");
  auto stage = [&](int chunk) {
    for (int r = threadIdx.x; r < kGT; r += kGQ) {
      const int t = chunk * kGT + r;
      float yx = 0.0f, yy = 0.0f, yz = 0.0f, n = 3.0e38f;  // rows past the cloud: never a minimum, never below a threshold
      float4 rw = {nanf_, nanf_, nanf_, 0.0f};
      if (t < nt) {
        const float x = tb[3LL * t], y = tb[3LL * t + 1], z = tb[3LL * t + 2];
        rw = float4{x, y, z, 0.0f};
        same = same && x == t0x && y == t0y && z == t0z;
        yx = x - cx, yy = y - cy, yz = z - cz;
        n = g_dist3(yx, yy, yz);
        mmax = n > mmax || n != n ? n : mmax;  // (a NaN norm sticks: the guard below sees it)
      }
      uint4 p0, p1;
      gate::target_row(yx, yy, yz, n, p0, p1);
      panel[0][r] = p0;
      panel[1][r] = p1;
      raw[r + (r >> 5)] = rw;
    }
  };
  // the cells of the panel in LDS: per lane and query tile the minimum of a over each tile's 16 rows of this lane half
  float tm0[kGTiles], tm1[kGTiles];
  float run0 = __builtin_inff(), run1 = __builtin_inff();
  auto bound = [&](int c) {
    const int rows = nt - c * kGT < kGT ? nt - c * kGT : kGT;
    const int nti = (rows + 31) / 32;
    const uint4* pl = &panel[h][j];
#pragma unroll
    for (int t8 = 0; t8 < kGTiles; t8 += 8) {
      if (t8 < nti) {  // (wave-uniform; the panel is padded to full tiles, so a group of 8 is always safe to process)
        uint4 nx = pl[32 * t8];
#pragma unroll
        for (int t = t8; t < t8 + 8; ++t) {
          const gate::bf16x8 a = gate::as_bf16x8(nx);
          if (t + 1 < t8 + 8) nx = pl[32 * (t + 1)];  // (the next tile's rows are on their way while this one is reduced)
          const f32x16 z = {0};
          const f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, gate::as_bf16x8(bq[0]), z, 0, 0, 0);
          const f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, gate::as_bf16x8(bq[1]), z, 0, 0, 0);
          float x, y;
          gate::min16(acc0, x, y);
          tm0[t] = gate::min3(x, y, __builtin_inff());
          run0 = gate::min3(run0, x, y);
          gate::min16(acc1, x, y);
          tm1[t] = gate::min3(x, y, __builtin_inff());
          run1 = gate::min3(run1, x, y);
          __builtin_amdgcn_sched_barrier(0);  // (tiles kept apart: hoisting the panel reads of many tiles spills registers)
        }
      } else {
#pragma unroll
        for (int t = t8; t < t8 + 8; ++t) tm0[t] = tm1[t] = __builtin_inff();
      }
    }
  };
  const int chunks = (nt + kGT - 1) / kGT;
  stage(0);
  const bool all_same = __syncthreads_and(same ? 1 : 0) != 0;  // (also the barrier behind the panel)
  if (chunks == 1 && all_same) {  // one distinct target: index 0 answers every query that sees it below 1e32
    const float d = g_dist3(X - t0x, Y - t0y, Z - t0z);
    if (d < 1e32f) {
      bd = d;
      bi = 0;
    }
    store();
    return;
  }
  // One panel: bound, thresholds, answer.  Several panels: a first sweep over all of them for tau and Mmax, a second
  // one that bounds each panel again and answers from it.
  const u64 none = g_key(1e32f, 0xffffffffu);
  u64 best0 = none, best1 = none;
  float thr0 = 0.0f, thr1 = 0.0f;
  bool wave_exact = false;
  const int sweeps = chunks == 1 ? 1 : 2;
  for (int sw = 0; sw < sweeps; ++sw) {
    const bool last = sw == sweeps - 1;
    for (int c = 0; c < chunks; ++c) {
      if (sw > 0 || c > 0) {
        __syncthreads();
        stage(c);
        __syncthreads();
      }
      bound(c);
      if (!last) continue;
      if (c == 0) {  // (tau and this thread's share of Mmax are complete)
        float v = mmax;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const float o = __shfl_xor(v, off, 64);
          v = o > v || o != o ? o : v;
        }
        if (lane == 0) redm[wave] = v;
        __syncthreads();
        mmax = redm[0];
#pragma unroll
        for (int w = 1; w < kGW; ++w) mmax = redm[w] > mmax || redm[w] != redm[w] ? redm[w] : mmax;
        // thresholds: the owner of a query (lane 32 h + j holds query 32 h + j = column j of tile h) forms it, rounding up
        const float tau0 = gate::min3(run0, __shfl_xor(run0, 32, 64), __builtin_inff());
        const float tau1 = gate::min3(run1, __shfl_xor(run1, 32, 64), __builtin_inff());
        float sl = gate::next_up(gate::next_up(mq + mmax) * (2.0f * kGKappa));
        sl = gate::next_up(sl + 1e-30f);
        const float thr_own = gate::next_up((h ? tau1 : tau0) + sl);
        thr0 = __shfl(thr_own, j, 64), thr1 = __shfl(thr_own, 32 + j, 64);
        const bool risky = has && !(mq <= kGMaxNorm && mmax <= kGMaxNorm);  // (negated: NaNs are risky)
        wave_exact = __ballot(risky) != 0;
      }
      if (wave_exact) continue;
      // ---- the answer: the qualifying cells, exactly ---------------------------------------------------------------------------
      unsigned k0 = 0u, k1 = 0u;
#pragma unroll
      for (int t = kGTiles - 1; t >= 0; --t) {  // (shifted in from the top: no 32 bit constants in registers)
        k0 = (k0 << 1) | (tm0[t] <= thr0 ? 1u : 0u);
        k1 = (k1 << 1) | (tm1[t] <= thr1 ? 1u : 0u);
      }
      while (__ballot((k0 | k1) != 0u)) {
        const bool act = (k0 | k1) != 0u, from1 = k0 == 0u;
        const unsigned mm = from1 ? k1 : k0;
        const int t = act ? __builtin_ctz(mm) : 0;
        k0 = from1 ? k0 : k0 & (k0 - 1u);
        k1 = from1 && act ? k1 & (k1 - 1u) : k1;
        // (a lane without a cell computes on a NaN query: its distances never beat anything)
        const bool own = from1 == (h == 1);
        const float qx = act ? (own ? X : OX) : nanf_, qy = own ? Y : OY, qz = own ? Z : OZ;
        u64 cur = from1 ? best1 : best0;
        const int row0 = 32 * t + 4 * h;
        const unsigned gidx = (unsigned)(c * kGT + row0);
#pragma unroll
        for (int g2 = 0; g2 < 8; ++g2) {  // rows 8 (g2 / 2) + 2 (g2 % 2) + u, two at a time (registers)
          const int ro = 8 * (g2 >> 1) + 2 * (g2 & 1);
          const float4 pa = raw[row0 + t + ro], pb = raw[row0 + t + ro + 1];
          const u64 ka = g_key(g_dist3(qx - pa.x, qy - pa.y, qz - pa.z), gidx + ro);
          const u64 kb = g_key(g_dist3(qx - pb.x, qy - pb.y, qz - pb.z), gidx + ro + 1);
          cur = ka < cur ? ka : cur;
          cur = kb < cur ? kb : cur;
        }
        best0 = from1 ? best0 : cur;
        best1 = from1 ? cur : best1;
      }
    }
  }
  if (wave_exact) {  // all targets, index order, strict `<`: the textbook loop (magnitudes the bound does not cover)
    for (int t = 0; t < nt; ++t) {
      const float d = g_dist3(X - tb[3LL * t], Y - tb[3LL * t + 1], Z - tb[3LL * t + 2]);
      if (d < bd) {
        bd = d;
        bi = t;
      }
    }
  } else {  // the two lanes of a query merge; the owner keeps the result
    const u64 o0 = __shfl_xor(best0, 32, 64), o1 = __shfl_xor(best1, 32, 64);
    const u64 m0 = o0 < best0 ? o0 : best0, m1 = o1 < best1 ? o1 : best1;
    const u64 mine = h ? m1 : m0;
    const unsigned db = (unsigned)(mine >> 32);
    if (db != __float_as_uint(1e32f)) {  // (a key at exactly 1e32 is the initial one, or a candidate the strict `<` refuses)
      bd = __uint_as_float(db);
      bi = (int)(unsigned)mine;
    }
  }
  store();
}

}  // namespace

int gate_tiles(int64_t na, int64_t nb) { return (int)(((na > nb ? na : nb) + kGQ - 1) / kGQ); }
// (survivor indices are 16-bit; beyond a few thousand points per cloud the grid-pruned search is the better structure anyway)
bool gate_supported(int64_t na, int64_t nb) { return na >= 1 && nb >= 1 && na <= 32768 && nb <= 32768; }

void launch_gate_part_search(const float* valids, const float* C1, const float* C2, int64_t B, int64_t P, int64_t N,
                             int32_t* idx1, int32_t* idx2, float* tile_sums, hipStream_t s) {
  GateArgs g = {};
  g.a = C1, g.b = C2, g.valids = valids;
  g.na = g.nb = (int)N;
  g.tiles = gate_tiles(N, N);
  g.idx32[0] = idx1, g.idx32[1] = idx2;
  g.tile_sums = tile_sums;
  g.M = (int)(B * P);
  hipLaunchKernelGGL((gate_nn_kernel<true>), dim3((unsigned)(B * P * g.tiles), 2), dim3(kGQ), 0, s, g);
}

void launch_gate_cloud_search(const float* xyz1, const float* xyz2, int64_t batch, int64_t n1, int64_t n2, float* dist1,
                              int64_t* idx1, float* dist2, int64_t* idx2, hipStream_t s) {
  GateArgs g = {};
  g.a = xyz1, g.b = xyz2, g.valids = nullptr;
  g.na = (int)n1, g.nb = (int)n2;
  g.tiles = gate_tiles(n1, n2);
  g.dist[0] = dist1, g.dist[1] = dist2;
  g.idx64[0] = reinterpret_cast<long long*>(idx1), g.idx64[1] = reinterpret_cast<long long*>(idx2);
  g.M = (int)batch;
  hipLaunchKernelGGL((gate_nn_kernel<false>), dim3((unsigned)(batch * g.tiles), 2), dim3(kGQ), 0, s, g);
}

}  // namespace mpa
