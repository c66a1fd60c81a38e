// Exact nearest neighbour between two mid-sized clouds with the pair work on the bf16 matrix cores — gfx950.
//
// Serves (a) rot_points_cd_loss of the fused assembly loss (assembly_loss.hip: every valid part against its own
// ground-truth copy, reference utils/loss.py:113-138) and (b) the generic Chamfer operator for clouds of a few hundred
// to a few thousand points (chamfer.hip: the reference's per-part call [B*P, N, 3]^2, utils/chamfer/chamfer.py:9-24).
// Same results as the exhaustive scan of chamfer_core.h, bit for bit: d = (dx*dx + dy*dy) + dz*dz with every operation
// rounded, lowest target index on ties, (1e32, -1) for a query that sees nothing below 1e32.
//
// The exhaustive scan spends ~4.5 VALU operations per (query, target) pair.  Here the pair work is ONE
// v_mfma_f32_32x32x16_bf16 per 32 x 32 pairs and half a VALU operation (pass 1) / one compare (pass 2) per pair; the
// pinned fp32 arithmetic is spent only on the handful of targets that can be the answer:
//
//   operands   y = x - c (c: any vector, here the mean of the first targets) is cut into bf16 pieces y = h + l + r.
//              Target row (16 bf16):  hx hy hz | hx hy hz | lx ly lz | NU (3 pieces) | NL (3 pieces) | 0
//              Query column, pass 1:  -2h      | -2l      | -2h      | 1 1 1         | 0 0 0         | 0
//                            pass 2:  -2h      | -2l      | -2h      | 0 0 0         | 1 1 1         | 0
//              so one K = 16 product gives  a(i,j) = N*_j - 2 (h_i.h_j + l_i.h_j + h_i.l_j)  ~  |y_j|^2 - 2 y_i.y_j
//              = d(i,j) - |y_i|^2.  NU / NL are the target's squared norm scaled up / down by kappa (an fp32 number is
//              exactly three bf16 pieces), so that with the query's own scaled norms QU / QL
//                  a_L(i,j) + QL_i  <=  d(i,j)  <=  a_U(i,j) + QU_i          (d: the PINNED distance)     ... (*)
//   pass 1     tau_i = min_j a_U(i,j): 8 x v_min3 per 16 accumulator values.  d(i, nearest) <= tau_i + QU_i.
//   pass 2     the same tiles with the NL columns; target j survives iff a_L(i,j) <= tau_i + (QU_i - QL_i).  By (*) every
//              target that attains the minimum of the pinned distance survives (ties included); typically 1-2 of a
//              thousand do.  Survivor indices go to a per-lane list in LDS.
//   answer     pinned distance of every survivor from the stored fp32 coordinates, lexicographic (distance, index) update.
//   A wave in which some query ends with no survivor or an overflowed list (non-finite values, coincident or lattice
//   clouds, out-of-range magnitudes) scans all targets in index order with the pinned arithmetic instead: the gate only
//   ever DECIDES which pairs get the exact evaluation, it never contributes a digit to a result.  A target cloud whose
//   points all coincide (the zero-padded parts of the reference's per-part call) is answered directly.
//
// kappa (per unit of M_i + M_j, M = |y|^2 in real arithmetic).  bf16 keeps 8 significant bits: |y - h| <= 2^-8 |y|,
// |l| <= 2^-8 |y|, |r| <= 2^-16 |y| per coordinate, so y_i.y_j - (hh + lh + hl) = l.l + r_i.y_j + (h + l)_i.r_j is at most
// 3.02 * 2^-16 sum_k |y_ik y_jk| <= 3.02 * 2^-16 (M_i + M_j) / 2; it enters a twice:              4.61e-5
// The 16 bf16 products are exact in fp32; their accumulation inside the matrix core is charged 2^-23 per term (the
// internal order and rounding are not documented) on sum |terms| <= 2.03 (M_i + M_j):                 3.9e-6
// y = fl(x - c) moves |y_i - y_j|^2 away from |x_i - x_j|^2 by <= 4.04 * 2^-24 (M_i + M_j):          2.4e-7
// the pinned chain is within 6 * 2^-24 of the real |x_i - x_j|^2 <= 2 (M_i + M_j):                    7.2e-7
// the fp32 norms m (three squares, two additions) are within 3 * 2^-24 of M, on either side:           3.6e-7
// Sum 5.13e-5; kappa = 6e-5.  NU = m (1 + kappa) rounded up, NL = m (1 - kappa) rounded down, QU / QL likewise; the
// threshold is rounded up and carries 1e-30 of absolute slack for products that underflow.
#include "assembly_internal.h"
#include "common.h"

namespace mpa {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kGW = 4;                 // waves per block: 256 queries
constexpr int kGQ = 64 * kGW;
constexpr int kGT = 1024;              // targets per LDS panel (32 KB)
constexpr int kGCap = 6;               // survivor slots per (query, lane half)
constexpr float kGKappa = 6.0e-5f;

__device__ __forceinline__ float g_next(float x) {  // the next float above a finite x (x >= 0 here or anything finite)
  return x >= 0.0f ? __uint_as_float(__float_as_uint(x + 0.0f) + 1u) : __uint_as_float(__float_as_uint(x) - 1u);
}
__device__ __forceinline__ float g_prev(float x) { return -g_next(-x); }
__device__ __forceinline__ float g_dist3(float dx, float dy, float dz) { return (dx * dx + dy * dy) + dz * dz; }
__device__ __forceinline__ unsigned g_bf(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float g_bf_f(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ unsigned g_pk(float lo, float hi) { return g_bf(lo) | (g_bf(hi) << 16); }
__device__ __forceinline__ int g_acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
__device__ __forceinline__ bf16x8 g_as_bf(const uint4 v) { return __builtin_bit_cast(bf16x8, v); }
// min(run, the 16 values): 8 x v_min3_f32 (three-operand minima only: the two-operand v_min_f32 makes the compiler
// canonicalise every accumulator register first, 8 more instructions per tile)
__device__ __forceinline__ float g_min3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float g_min16(float run, const f32x16& a) {
  const float m0 = g_min3(a[0], a[1], a[2]), m1 = g_min3(a[3], a[4], a[5]), m2 = g_min3(a[6], a[7], a[8]);
  const float m3 = g_min3(a[9], a[10], a[11]), m4 = g_min3(a[12], a[13], a[14]);
  return g_min3(run, g_min3(m0, m1, m2), g_min3(m3, m4, a[15]));
}
// x = p0 + p1 + p2 exactly (three bf16 pieces of a finite fp32 number)
__device__ __forceinline__ void g_split3(float x, float& p0, float& p1, float& p2) {
  p0 = g_bf_f(x);
  const float r1 = x - p0;
  p1 = g_bf_f(r1);
  p2 = r1 - p1;
}

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
__global__ __launch_bounds__(kGQ, 4) void gate_nn_kernel(const GateArgs g) {
  __shared__ uint4 panel[2][kGT];                      // plane k-half h: row r -> 8 bf16
  __shared__ unsigned short lst[kGW * 2 * kGCap * 64];  // [(wave * 2 + set) * cap + slot][lane]
  __shared__ unsigned char cntl[kGW * 2 * 64];
  __shared__ float red[kGW];
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
  auto exact_scan = [&]() {  // all targets, index order, strict `<`: the textbook loop (rare)
    bd = 1e32f;
    bi = -1;
    for (int t = 0; t < nt; ++t) {
      const float d = g_dist3(X - tb[3LL * t], Y - tb[3LL * t + 1], Z - tb[3LL * t + 2]);
      if (d < bd) {
        bd = d;
        bi = t;
      }
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
  // (tile s = queries 32 s + j of the wave; lane (j, h) supplies k-half h of column j)
  uint4 b1[2], b2[2];
  float slack;
  {
    const float yx = X - cx, yy = Y - cy, yz = Z - cz;
    const float hx = g_bf_f(yx), hy = g_bf_f(yy), hz = g_bf_f(yz);
    const float lx = yx - hx, ly = yy - hy, lz = yz - hz;
    const float mq = g_dist3(yx, yy, yz);
    const float QU = g_next(__builtin_fmaf(mq, kGKappa, mq)), QL = g_prev(__builtin_fmaf(mq, -kGKappa, mq));
    slack = g_next(g_next(QU - QL) + 1e-30f);
    const float a = -2.0f;
    const uint4 k0 = {g_pk(a * hx, a * hy), g_pk(a * hz, a * lx), g_pk(a * ly, a * lz), g_pk(a * hx, a * hy)};
    const unsigned one = 0x3f80u, z8 = g_bf(a * hz);
    const uint4 k1u = {z8 | (one << 16), one | (one << 16), 0u, 0u};
    const uint4 k1l = {z8, 0u, one | (one << 16), one};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int src = 32 * s + j;
      uint4 v0, vu, vl;
      v0.x = __shfl(k0.x, src, 64), v0.y = __shfl(k0.y, src, 64), v0.z = __shfl(k0.z, src, 64), v0.w = __shfl(k0.w, src, 64);
      vu.x = __shfl(k1u.x, src, 64), vu.y = __shfl(k1u.y, src, 64), vu.z = __shfl(k1u.z, src, 64), vu.w = __shfl(k1u.w, src, 64);
      vl.x = __shfl(k1l.x, src, 64), vl.y = __shfl(k1l.y, src, 64), vl.z = __shfl(k1l.z, src, 64), vl.w = __shfl(k1l.w, src, 64);
      b1[s] = uint4{h ? vu.x : v0.x, h ? vu.y : v0.y, h ? vu.z : v0.z, h ? vu.w : v0.w};
      b2[s] = uint4{h ? vl.x : v0.x, h ? vl.y : v0.y, h ? vl.z : v0.z, h ? vl.w : v0.w};
    }
  }

  // ---- target panel: rows chunk * kGT + r, r < kGT ------------------------------------------------------------------------------
  const float t0x = tb[0], t0y = tb[1], t0z = tb[2];
  bool same = true;
  auto stage = [&](int chunk) {
    for (int r = threadIdx.x; r < kGT; r += kGQ) {
      const int t = chunk * kGT + r;
      uint4 p0 = {0u, 0u, 0u, 0u}, p1;
      float NU = 3.0e38f, NL = 3.0e38f;  // rows past the cloud: never the minimum, never below a threshold
      unsigned lzb = 0u;
      if (t < nt) {
        const float x = tb[3LL * t], y = tb[3LL * t + 1], z = tb[3LL * t + 2];
        same = same && x == t0x && y == t0y && z == t0z;
        const float yx = x - cx, yy = y - cy, yz = z - cz;
        const float hx = g_bf_f(yx), hy = g_bf_f(yy), hz = g_bf_f(yz);
        const float lx = yx - hx, ly = yy - hy, lz = yz - hz;
        const float mt = g_dist3(yx, yy, yz);
        NU = g_next(__builtin_fmaf(mt, kGKappa, mt));
        NL = g_prev(__builtin_fmaf(mt, -kGKappa, mt));
        p0 = uint4{g_pk(hx, hy), g_pk(hz, hx), g_pk(hy, hz), g_pk(lx, ly)};
        lzb = g_bf(lz);
      }
      float u0, u1, u2, l0, l1, l2;
      g_split3(NU, u0, u1, u2);
      g_split3(NL, l0, l1, l2);
      p1 = uint4{lzb | (g_bf(u0) << 16), g_pk(u1, u2), g_pk(l0, l1), g_bf(l2)};
      panel[0][r] = p0;
      panel[1][r] = p1;
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

  // ---- pass 1: tau = min_j a_U --------------------------------------------------------------------------------------------------
  float m0 = __builtin_inff(), m1 = __builtin_inff();
  for (int c = 0; c < chunks; ++c) {
    if (c > 0) {
      __syncthreads();
      stage(c);
      __syncthreads();
    }
    const int rows = nt - c * kGT < kGT ? nt - c * kGT : kGT;
    const int nti = (rows + 31) / 32;
    const uint4* pl = &panel[h][j];

    for (int t = 0; t < nti; ++t) {
      const bf16x8 a = g_as_bf(pl[32 * t]);
      const f32x16 z = {0};
      const f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, g_as_bf(b1[0]), z, 0, 0, 0);
      const f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, g_as_bf(b1[1]), z, 0, 0, 0);
      m0 = g_min16(m0, acc0);
      m1 = g_min16(m1, acc1);
    }
  }
  m0 = __builtin_fminf(m0, __shfl_xor(m0, 32, 64));
  m1 = __builtin_fminf(m1, __shfl_xor(m1, 32, 64));
  // the owner of a query (lane 32 h + j holds query 32 h + j = column j of tile h) forms its threshold
  const float thr_own = g_next((h ? m1 : m0) + slack);
  const float thr0 = __shfl(thr_own, j, 64), thr1 = __shfl(thr_own, 32 + j, 64);

  // ---- pass 2: survivors a_L <= thr ------------------------------------------------------------------------------------------------
  int cnt0 = 0, cnt1 = 0;
  unsigned short* l0p = lst + ((wave * 2 + 0) * kGCap) * 64 + lane;
  unsigned short* l1p = lst + ((wave * 2 + 1) * kGCap) * 64 + lane;
  for (int c = 0; c < chunks; ++c) {
    if (chunks > 1) {  // (one panel: still in LDS)
      __syncthreads();
      stage(c);
      __syncthreads();
    }
    const int rows = nt - c * kGT < kGT ? nt - c * kGT : kGT;
    const int nti = (rows + 31) / 32;
    const uint4* pl = &panel[h][j];

    for (int t = 0; t < nti; ++t) {
      const bf16x8 a = g_as_bf(pl[32 * t]);
      const f32x16 z = {0};
      const f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, g_as_bf(b2[0]), z, 0, 0, 0);
      const f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, g_as_bf(b2[1]), z, 0, 0, 0);
      const int base = c * kGT + 32 * t + 4 * h;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (acc0[r] <= thr0) {
          l0p[(cnt0 < kGCap ? cnt0 : kGCap - 1) * 64] = (unsigned short)(base + (r & 3) + 8 * (r >> 2));
          ++cnt0;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (acc1[r] <= thr1) {
          l1p[(cnt1 < kGCap ? cnt1 : kGCap - 1) * 64] = (unsigned short)(base + (r & 3) + 8 * (r >> 2));
          ++cnt1;
        }
      }
    }
  }
  cntl[(wave * 2 + 0) * 64 + lane] = (unsigned char)(cnt0 < 255 ? cnt0 : 255);
  cntl[(wave * 2 + 1) * 64 + lane] = (unsigned char)(cnt1 < 255 ? cnt1 : 255);
  __syncthreads();

  // ---- the answer: pinned distances of the survivors (owner lane: its query is column j of tile h) -----------------------------
  const int c0 = cntl[(wave * 2 + h) * 64 + j], c1 = cntl[(wave * 2 + h) * 64 + 32 + j];
  const bool bad = has && (c0 + c1 == 0 || c0 > kGCap || c1 > kGCap);
  if (__ballot(bad)) {
    exact_scan();
  } else {
    const int total = has ? c0 + c1 : 0;
    const unsigned short* la = lst + ((wave * 2 + h) * kGCap) * 64 + j;
    int tmax = total;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(tmax, off, 64);
      tmax = o > tmax ? o : tmax;
    }
    for (int e0 = 0; e0 < tmax; e0 += 4) {
      int ti[4];
      float tx[4], ty[4], tz[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u;
        const int ec = e < total ? e : 0;
        ti[u] = total > 0 ? (int)(ec < c0 ? la[ec * 64] : la[(ec - c0) * 64 + 32]) : 0;
        tx[u] = tb[3LL * ti[u]];
        ty[u] = tb[3LL * ti[u] + 1];
        tz[u] = tb[3LL * ti[u] + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float d = g_dist3(X - tx[u], Y - ty[u], Z - tz[u]);
        const bool better = e0 + u < total && (d < bd || (d == bd && bi >= 0 && ti[u] < bi));
        bd = better ? d : bd;
        bi = better ? ti[u] : bi;
      }
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
