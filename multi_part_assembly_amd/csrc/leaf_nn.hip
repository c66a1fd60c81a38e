// Exact nearest-neighbour search over LEAVES of rigid parts — the Chamfer searches of the fused assembly loss
// (assembly_loss.hip): rot_points_cd_loss (every part against its own ground-truth copy, utils/loss.py:113-138) and
// shape_cd_loss (every valid point against the whole other shape, utils/loss.py:141-202).  Same results as the
// brute-force scan, bit for bit — distances with the pinned arithmetic of chamfer_core.h, lowest original index on
// ties — but the spatial structure is not rebuilt per evaluation:
//
//   * Both clouds of every search are rigid images of the SAME source points (part_pcs).  A balanced k-d ordering of
//     each part's N points, computed ONCE per batch on the local coordinates (leaf_order_kernel), therefore cuts every
//     transformed copy of the part into the same compact leaves of 32 points — for the predicted and the ground-truth
//     pose, for every GNN iteration and every min-of-N sample of the step.  The pose kernel of the loss walks the
//     points in that order and leaves, per cloud, cell-free "records" (x, y, z, original index) plus one bounding
//     box per leaf and per part: no sort, no grid, no histogram per loss evaluation.
//   * The twin point: query k of part p has its own image at slot k of part p in the other cloud — one distance
//     evaluation gives every query a valid first candidate (and the wave a first bound) before any search.
//   * One wave = 64 consecutive queries of a part (two leaves: a compact box).  It ranks the target parts by the
//     distance between boxes (lane = part), then the leaves of a part (lane = leaf), nearest first, and scans a leaf
//     only if some lane's own point-to-box distance can still beat (or tie) that lane's best.  The box distances are
//     evaluated with the same expression shape as the point distance — every rounding is monotone — so they are
//     rigorous lower bounds of the pinned fp32 distance without any slack term.
//   * A leaf's records sit in the wave's registers, 16 per row of lanes, and reach the queries through DPP row
//     rotations on the operand of the subtraction (no LDS, no barrier, no scalar-cache traffic); candidates are not
//     visited in index order, so updates are lexicographic (smaller distance, then smaller original index) — the
//     answer of the in-order strict-`<` scan.
//
// hipcc-flags: -fno-slp-vectorize
// (read by _build.py: the SLP vectoriser packs the distance arithmetic into v_pk_*_f32, which cannot take the DPP
// operand — two thirds of the row rotations became separate v_mov_b32_dpp instructions)
#include "assembly_internal.h"
#include "common.h"

namespace mpa {
namespace {

#ifndef MPA_LEAF_GATE  // 1: leaves are screened on the matrix cores (default); 0: every visited leaf is scanned exactly
#define MPA_LEAF_GATE 1
#endif
constexpr int kLeaf = 32;                 // points per leaf
constexpr int kLeafMaxPad = 2048;         // padded points per part (<= 64 leaves: one lane per leaf)
constexpr int kNoIdx = 0x7fffffff;

// ---- wave reductions on the DPP network (non-negative floats compare like their bit patterns) --------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_min_u(unsigned v) {
  v = min(v, dpp_u<0xB1, 0xf>(v));   // quad_perm [1,0,3,2]
  v = min(v, dpp_u<0x4E, 0xf>(v));   // quad_perm [2,3,0,1]
  v = min(v, dpp_u<0x141, 0xf>(v));  // row_half_mirror
  v = min(v, dpp_u<0x140, 0xf>(v));  // row_mirror: every row of 16 holds its minimum
  v = min(v, dpp_u<0x142, 0xa>(v));  // row_bcast15 into rows 1, 3
  v = min(v, dpp_u<0x143, 0xc>(v));  // row_bcast31 into rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_max_u(unsigned v) {
  v = max(v, dpp_u<0xB1, 0xf>(v));
  v = max(v, dpp_u<0x4E, 0xf>(v));
  v = max(v, dpp_u<0x141, 0xf>(v));
  v = max(v, dpp_u<0x140, 0xf>(v));
  v = max(v, dpp_u<0x142, 0xa>(v));
  v = max(v, dpp_u<0x143, 0xc>(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float dist3(float dx, float dy, float dz) { return (dx * dx + dy * dy) + dz * dz; }
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// lower bound of the pinned distance between a point and any point of the box [lo, hi] (same expression shape as
// dist3: subtraction, square, the two additions are monotone under rounding)
__device__ __forceinline__ float lb_point_box(float X, float Y, float Z, const float* __restrict__ bx) {
  const float gx = max3f(bx[0] - X, X - bx[4], 0.0f);
  const float gy = max3f(bx[1] - Y, Y - bx[5], 0.0f);
  const float gz = max3f(bx[2] - Z, Z - bx[6], 0.0f);
  return dist3(gx, gy, gz);
}
// ... and between any point of the box [qlo, qhi] and any point of the box bx
__device__ __forceinline__ float lb_box_box(const float (&qlo)[3], const float (&qhi)[3], const float* __restrict__ bx) {
  const float gx = max3f(bx[0] - qhi[0], qlo[0] - bx[4], 0.0f);
  const float gy = max3f(bx[1] - qhi[1], qlo[1] - bx[5], 0.0f);
  const float gz = max3f(bx[2] - qhi[2], qlo[2] - bx[6], 0.0f);
  return dist3(gx, gy, gz);
}

// ---- 1. the k-d ordering of a part's points (once per batch) ------------------------------------------------------------
// ONE WAVE per part.  Level by level the segment [0, Npad) is halved: every segment is sorted along the widest axis of
// its own bounding box and cut in the middle, down to segments of 32 slots = the leaves.  Everything after the first
// pass over the points is integer: coordinates quantised to 16 bits over the part's box (three u16 arrays in LDS), sort
// keys ONE 32-bit word (coordinate << 11 | point index: a strict total order, so the result is deterministic), segment
// boxes as integer minima / maxima.  Pad slots (beyond N) carry the largest key and stay at the end of the last segment.
// The ordering only steers speed: the search is exact for any permutation.
// Why one wave: the bitonic network is a chain of ~185 dependent stages whatever the block size.  A block of 16 waves
// spends it on barriers and holds 33 KB of LDS for 110 us — enough to push a PointNet block off its CU when the kernel
// runs beside the encoder (pn_fwd_mfma 47 -> 82 us, measured).  A single wave needs no barrier at all (LDS operations of
// one wave execute in order), 10 bytes of LDS per slot, and leaves the chip to whoever runs beside it.
// Output: sorted[m][k] = (x, y, z, original index n as int bits; -1 in pad slots) in LOCAL coordinates.
constexpr unsigned kPadKey = 0xffffffffu;

__global__ __launch_bounds__(64) void leaf_order_kernel(const float* __restrict__ pcs, const float* __restrict__ valids,
                                                        int N, int Npad, float4* __restrict__ sorted) {
  extern __shared__ unsigned order_lds[];
  unsigned* key = order_lds;                                                   // [Npad]
  unsigned short* qc = reinterpret_cast<unsigned short*>(order_lds + Npad);  // [3][Npad] quantised x | y | z
  const int m = blockIdx.x, lane = threadIdx.x;
  if (valids[m] == 0.0f) return;
  const float* src = pcs + 3LL * m * N;
  float4* out = sorted + (long long)m * Npad;
  if (Npad <= kLeaf) {  // one leaf: any order
    if (lane < Npad)
      out[lane] = lane < N ? make_float4(src[3 * lane], src[3 * lane + 1], src[3 * lane + 2], __int_as_float(lane))
                           : make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
    return;
  }
  // the part's box, then the quantised coordinates
  const float inf = __builtin_inff();
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  for (int n = lane; n < N; n += 64) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = src[3 * n + a];
      lo[a] = __builtin_fminf(lo[a], v);
      hi[a] = __builtin_fmaxf(hi[a], v);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = __builtin_fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = __builtin_fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
  }
  float sc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) sc[a] = hi[a] > lo[a] ? 65535.0f / (hi[a] - lo[a]) : 0.0f;  // (degenerate / non-finite: one bucket)
  for (int n = lane; n < Npad; n += 64) {
    const bool in = n < N;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = in ? src[3 * n + a] : 0.0f;
      qc[a * Npad + n] = (unsigned short)__builtin_amdgcn_fmed3f((v - lo[a]) * sc[a], 0.0f, 65535.0f);
    }
    key[n] = in ? (unsigned)n : kPadKey;  // low 11 bits: the point held by this slot
  }
  __builtin_amdgcn_wave_barrier();
  const int PER = Npad / 64;  // consecutive slots per lane in the box / key phases (>= 1; a lane's slots share a segment)
  for (int S = Npad; S > kLeaf; S >>= 1) {
    // integer box of this lane's slots, then of its segment (the G = S / PER lanes that hold it: xor-shuffles below G)
    unsigned blo[3] = {0xffffu, 0xffffu, 0xffffu}, bhi[3] = {0u, 0u, 0u};
    for (int r = 0; r < PER; ++r) {
      const unsigned e = key[lane * PER + r];
      if (e != kPadKey) {
        const int n = (int)(e & 2047u);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const unsigned v = qc[a * Npad + n];
          blo[a] = min(blo[a], v);
          bhi[a] = max(bhi[a], v);
        }
      }
    }
    const int G = S / PER;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      if (off < G) {  // (wave-uniform)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          blo[a] = min(blo[a], (unsigned)__shfl_xor((int)blo[a], off, 64));
          bhi[a] = max(bhi[a], (unsigned)__shfl_xor((int)bhi[a], off, 64));
        }
      }
    }
    const int ex = (int)bhi[0] - (int)blo[0], ey = (int)bhi[1] - (int)blo[1], ez = (int)bhi[2] - (int)blo[2];
    const int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
    for (int r = 0; r < PER; ++r) {
      const unsigned e = key[lane * PER + r];
      if (e != kPadKey) {
        const unsigned n = e & 2047u;
        key[lane * PER + r] = ((unsigned)qc[axis * Npad + n] << 11) | n;  // (always below the pad key)
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ascending bitonic sort of every aligned S-segment: Npad / 2 compare-exchange pairs per stage, pair t of lane
    // t % 64; up to 8 pairs' loads in flight
    const int npairs = Npad / 2;
    auto stage = [&](int lk, int lj) {  // lj < 0: the mirror stage of merge size 2^lk; else partner distance 2^lj
      for (int t0 = lane; t0 < npairs; t0 += 64 * 8) {
        unsigned a[8], b[8];
        int ia[8], ib[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int t = t0 + 64 * u;
          const int tc = t < npairs ? t : t0;  // (clamped: a duplicate of pair t0, rewritten with the same values)
          if (lj < 0) {
            const int k = 1 << lk, base = (tc >> (lk - 1)) << lk, off = tc & ((k >> 1) - 1);
            ia[u] = base + off;
            ib[u] = base + (k - 1 - off);
          } else {
            const int j = 1 << lj;
            ia[u] = ((tc >> lj) << (lj + 1)) + (tc & (j - 1));
            ib[u] = ia[u] + j;
          }
          a[u] = key[ia[u]];
          b[u] = key[ib[u]];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          key[ia[u]] = min(a[u], b[u]);
          key[ib[u]] = max(a[u], b[u]);
        }
      }
      __builtin_amdgcn_wave_barrier();  // (one wave: program order is all the synchronisation LDS needs)
    };
    for (int lk = 1; (1 << lk) <= S; ++lk) {
      stage(lk, -1);
      for (int lj = lk - 2; lj >= 0; --lj) stage(lk, lj);
    }
  }
  for (int k = lane; k < Npad; k += 64) {
    const unsigned e = key[k];
    const bool real = e != kPadKey;
    const int n = real ? (int)(e & 2047u) : 0;
    out[k] = make_float4(src[3 * n], src[3 * n + 1], src[3 * n + 2], __int_as_float(real ? n : -1));
  }
}

#ifdef MPA_LEAF_STATS  // instrumented build for tools/probe_leaf_stats.py only (never in libmpa_hip.so)
// [shape][0] wave searches (both passes), [1] leaf tests, [2] leaf scans, [3] part visits, [4] max scans of one wave search,
// [7] waves deferred to the second pass, [8 + k] wave searches with scans in [2^k, 2^(k+1))
__device__ unsigned long long g_leaf_stats[2][32];
#define MPA_LSTAT(var) ++(var)
#else
#define MPA_LSTAT(var) do { } while (0)
#endif

// ---- 2. the search --------------------------------------------------------------------------------------------------------
// A leaf's 32 records live in the wave's own registers: lane l holds slots (l % 16) and 16 + (l % 16) (two coalesced
// vector loads), and every lane meets the 16 targets of its ROW of 16 lanes through the DPP network — `row_ror:n` on
// the operand of the subtraction, no instruction of its own.  (The records as SGPR operands through the scalar cache
// — the feed of the exhaustive scans in chamfer_core.h, where every wave of a CU streams the same targets — ran this
// search no faster than LDS staging: here every wave walks its own leaves and the scalar cache serves misses one at a
// time.)  Two rotations are evaluated per packed instruction (v_pk_mul_f32 / v_pk_add_f32 on register pairs).
//
// The running best of a lane is ONE 64-bit key, (distance bits << 32) | index: non-negative floats order like their bit
// patterns, so the lexicographic rule (smaller distance, then smaller original index) is a single unsigned 64-bit
// compare per candidate — no chunk minimum, no rare path (with 64 queries per wave SOME lane improves in almost every
// chunk).  NaN distances have the largest bit patterns and never win; a lane without a query holds key 0.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

template <int N>
__device__ __forceinline__ float ror_f(float v) {
  if constexpr (N == 0) return v;
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ unsigned ror_u(unsigned v) {
  if constexpr (N == 0) return v;
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xf, 0xf, true);
}
__device__ __forceinline__ u64 make_key(float d, unsigned idx) { return ((u64)__float_as_uint(d) << 32) | idx; }
__device__ __forceinline__ float key_dist(u64 k) { return __uint_as_float((unsigned)(k >> 32)); }

// the targets at row rotations N and N + 1 of `t` against this lane's query
template <int N>
__device__ __forceinline__ void scan_pair(const float4& t, float X, float Y, float Z, u64& best) {
  const f32x2 dx = {X - ror_f<N>(t.x), X - ror_f<N + 1>(t.x)};
  const f32x2 dy = {Y - ror_f<N>(t.y), Y - ror_f<N + 1>(t.y)};
  const f32x2 dz = {Z - ror_f<N>(t.z), Z - ror_f<N + 1>(t.z)};
  const f32x2 d = (dx * dx + dy * dy) + dz * dz;
  const unsigned w = __float_as_uint(t.w);
  const u64 k0 = make_key(d[0], ror_u<N>(w)), k1 = make_key(d[1], ror_u<N + 1>(w));
  best = k0 < best ? k0 : best;
  best = k1 < best ? k1 : best;
}
__device__ __forceinline__ void scan16(const float4& t, float X, float Y, float Z, u64& best) {
  scan_pair<0>(t, X, Y, Z, best);
  scan_pair<2>(t, X, Y, Z, best);
  scan_pair<4>(t, X, Y, Z, best);
  scan_pair<6>(t, X, Y, Z, best);
  scan_pair<8>(t, X, Y, Z, best);
  scan_pair<10>(t, X, Y, Z, best);
  scan_pair<12>(t, X, Y, Z, best);
  scan_pair<14>(t, X, Y, Z, best);
}

// ---- the matrix-core gate ----------------------------------------------------------------------------------------------------
// Most of a leaf's 32 x 64 (target, query) pairs cannot matter: per query only the leaf's nearest target can, and only
// if it beats the query's running best.  The matrix cores find that target without a single VALU distance evaluation:
//     v(i, j) = (|t_i'|^2 + Q) - 2 q_j' . t_i'  =  |q_j - t_i|^2 + (Q - |q_j'|^2)      (primes: relative to the wave's centre c)
// is a K = 4 product of the rows (t'x, t'y, t'z, |t'|^2 + Q) with the columns (-2 q'x, -2 q'y, -2 q'z, 1): four
// v_mfma_f32_32x32x2_f32 per leaf for the wave's two query tiles.  Q = max_j |q_j'|^2 keeps v positive, so it orders
// like its bit pattern; a lane holds 16 of a query's 32 values per tile, packs the value's register into the low 4 bits,
// and keeps the smallest and second smallest (v_min_f32 / v_med3_f32); one lane-half swap merges the two halves of a query.
// The approximation only DECIDES, the pinned arithmetic ANSWERS (E bounds |v - R - d| for every pair of the leaf):
//   * nearest value - E above the query's best           -> no target of the leaf can win or tie: nothing to do;
//   * else, runner-up clearly above the nearest value    -> the leaf's exact minimum IS that target (strictly: no tie is
//     possible): ONE exact distance (pinned arithmetic on the stored coordinates) and the usual lexicographic update;
//   * else (near-ties, duplicates, non-finite values)    -> the whole wave scans the leaf exactly (scan16, rare).
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GateWave {    // per wave, set up once
  float cx, cy, cz;  // centre of the query box
  float qq, R, Q;    // |q'|^2 of this lane's query, Q - qq, Q
  float b0_01, b0_23, b1_01, b1_23;  // B operands of the two query tiles (queries 0-31 / 32-63)
};

__device__ __forceinline__ void gate_setup(GateWave& w, const float (&qlo)[3], const float (&qhi)[3], float X, float Y,
                                           float Z, bool has) {
  const int lane = threadIdx.x & 63;
  w.cx = 0.5f * (qlo[0] + qhi[0]), w.cy = 0.5f * (qlo[1] + qhi[1]), w.cz = 0.5f * (qlo[2] + qhi[2]);
  const float qx = has ? X - w.cx : 0.0f, qy = has ? Y - w.cy : 0.0f, qz = has ? Z - w.cz : 0.0f;
  w.qq = dist3(qx, qy, qz);
  // (2 % above the largest |q'|^2: R = Q - qq stays far above the rounding of v even where a target coincides with its query)
  w.Q = __uint_as_float(wave_max_u(__float_as_uint(w.qq))) * 1.02f + 1e-30f;
  w.R = w.Q - w.qq;
  const float ax = -2.0f * qx, ay = -2.0f * qy, az = -2.0f * qz;
  const float ox = __shfl_xor(ax, 32, 64), oy = __shfl_xor(ay, 32, 64), oz = __shfl_xor(az, 32, 64);
  const bool lo = lane < 32;
  // lane (j, h): B[k = h][column j]; tile 0 = the queries of lanes 0-31, tile 1 = of lanes 32-63
  w.b0_01 = lo ? ax : oy;
  w.b0_23 = lo ? az : 1.0f;
  w.b1_01 = lo ? ox : ay;
  w.b1_23 = lo ? oz : 1.0f;
}

// smallest and second smallest of a tile's 16 values of this lane, with the value's register 4g + u in the low 4 bits
// (target row 8g + u, + 4 for the upper lane half).  Positive floats: the packed words still order like the values.
// Four independent chains (one per g), merged pairwise.
__device__ __forceinline__ void gate_merge(float& m, float& s, float m2, float s2) {
  s = __builtin_fminf(__builtin_fmaxf(m, m2), __builtin_fminf(s, s2));
  m = __builtin_fminf(m, m2);
}
__device__ __forceinline__ void gate_reduce(const f32x16& acc, float& m, float& s) {
  float mg[4], sg[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    float p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = __int_as_float((__float_as_int(acc[4 * g4 + u]) & ~15) | (4 * g4 + u));
    mg[g4] = __builtin_fminf(p[0], p[1]);
    sg[g4] = __builtin_fmaxf(p[0], p[1]);
    sg[g4] = __builtin_amdgcn_fmed3f(mg[g4], p[2], sg[g4]);
    mg[g4] = __builtin_fminf(mg[g4], p[2]);
    sg[g4] = __builtin_amdgcn_fmed3f(mg[g4], p[3], sg[g4]);
    mg[g4] = __builtin_fminf(mg[g4], p[3]);
  }
  gate_merge(mg[0], sg[0], mg[1], sg[1]);
  gate_merge(mg[2], sg[2], mg[3], sg[3]);
  gate_merge(mg[0], sg[0], mg[2], sg[2]);
  m = mg[0], s = sg[0];
}

// Everything a search launch reads.  Records rec [B*P][Npad] float4 (x, y, z, index: SHAPE p * N + n, else n; pad slots
// (inf, inf, inf, kNoIdx)), leaf boxes [B*P][Npad / 32][8] (lo xyz, -, hi xyz, -), part boxes [B*P][8], the cloud in
// original order [B, P, N, 3] (SHAPE: the padded parts' representatives).  Side 0 / 1 = cloud A / B; direction 0:
// cloud A's points are the queries.
struct LeafArgs {
  const float* valids;
  const float4* rec[2];
  const float* leaf[2];
  const float* part[2];
  const float* orig[2];
  int P, N, Npad, NW;   // NW = waves per part = max(1, Npad / 64)
  int* idx[2];          // per direction
  float* wave_sums;     // [2][B*P][NW]
  // waves that hit the scan cap of the first pass: their count, ids (((m * NW + w) * 2) + dir) and 64 keys each
  int* heavy_count;
  int* heavy_list;
  u64* heavy_keys;
  int total_parts;      // B * P
  const int* route;     // nullable [B]: SHAPE skips samples with route[b] != 0 (the grid search's)
};

#ifndef MPA_LEAF_CAP
#define MPA_LEAF_CAP 40
#endif
constexpr int kScanCap = MPA_LEAF_CAP;   // leaf scans of one wave in the first pass (the slowest wave bounds the launch)
constexpr int kSplit = 8;      // waves that share one deferred wave's leaves in the second pass

// One wave's search: queries = the 64 sorted slots k0 .. k0 + 63 of part m, direction dir.  `best` comes in initialised
// (twin / padded parts / saved state).  Only leaves l with l % split == rem are visited; at most `cap` leaf scans.
// Returns false if the cap stopped the search early.  No lane-divergent control flow between the first and the last DPP
// read: every branch is wave-uniform.
template <bool SHAPE>
__device__ __forceinline__ bool leaf_search_wave(const LeafArgs& g, int m, int k0, int dir, float X, float Y, float Z,
                                                 bool has, u64& best, int split, int rem, int cap) {
  const int P = g.P, Npad = g.Npad, NL = Npad / kLeaf, b = m / P, p = m % P;
  const int lane = threadIdx.x & 63;
  const float4* trec = g.rec[1 - dir];
  const float* qleaf = g.leaf[dir];
  const float* tleaf = g.leaf[1 - dir];
  const float* tpart = g.part[1 - dir];
  // box of this wave's queries: its two leaves
  float qlo[3], qhi[3];
  {
    const int l0 = k0 / kLeaf, l1 = l0 + 1 < NL ? l0 + 1 : l0;
    const float* a = qleaf + ((long long)m * NL + l0) * 8;
    const float* c = qleaf + ((long long)m * NL + l1) * 8;
#pragma unroll
    for (int x = 0; x < 3; ++x) {
      qlo[x] = __builtin_fminf(a[x], c[x]);
      qhi[x] = __builtin_fmaxf(a[4 + x], c[4 + x]);
    }
  }
  unsigned bound = wave_max_u((unsigned)(best >> 32));
  int scans = 0;
  bool done = true;
  [[maybe_unused]] int st_tests = 0, st_parts = 0, st_exact = 0;
#if MPA_LEAF_GATE
  GateWave gw;
  gate_setup(gw, qlo, qhi, X, Y, Z, has);
#endif
  // The leaves of target part tp, nearest first.  Lane l holds leaf l's box and its distance to the query box; the
  // records of the leaf to come are requested while the current one is scanned.
  auto search_part = [&](int tp) {
    const float4* lbx = reinterpret_cast<const float4*>(tleaf + ((long long)(b * P + tp) * NL) * 8);
    const float4 blo = lbx[2 * (lane < NL ? lane : 0)], bhi = lbx[2 * (lane < NL ? lane : 0) + 1];
    unsigned lbl = 0xffffffffu;
    {
      const float bx[8] = {blo.x, blo.y, blo.z, 0.0f, bhi.x, bhi.y, bhi.z, 0.0f};
      if (lane < NL && lane % split == rem) lbl = __float_as_uint(lb_box_box(qlo, qhi, bx));
    }
    const float4* rec = trec + (long long)(b * P + tp) * Npad;
    auto pick = [&](unsigned& lm) {  // nearest unvisited leaf within the bound, or -1
      lm = wave_min_u(lbl);
      if (!(lm <= bound) || lm > 0x7f800000u) return -1;
      const int ll = __builtin_ctzll(__ballot(lbl == lm));
      if (lane == ll) lbl = 0xffffffffu;
      return ll;
    };
    unsigned clm, nlm;
    int cur = pick(clm);
#if MPA_LEAF_GATE
    float4 tG;  // lane l: record l % 32 of the leaf
    if (cur >= 0) tG = rec[cur * kLeaf + (lane & 31)];
#else
    float4 tA, tB;
    if (cur >= 0) {
      tA = rec[cur * kLeaf + (lane & 15)];
      tB = rec[cur * kLeaf + 16 + (lane & 15)];
    }
#endif
    while (cur >= 0) {
      if (scans >= cap) {  // deferred to the second pass (which visits every leaf again, with this wave's best as bound)
        done = false;
        return;
      }
      const int nxt = pick(nlm);
#if MPA_LEAF_GATE
      float4 nG = tG;
      if (nxt >= 0) nG = rec[nxt * kLeaf + (lane & 31)];
#else
      float4 nA = tA, nB = tB;
      if (nxt >= 0) {
        nA = rec[nxt * kLeaf + (lane & 15)];
        nB = rec[nxt * kLeaf + 16 + (lane & 15)];
      }
#endif
      if (clm <= bound) {  // (the bound may have shrunk since this leaf was picked)
        const float bx[8] = {readlane_f(blo.x, cur), readlane_f(blo.y, cur), readlane_f(blo.z, cur), 0.0f,
                             readlane_f(bhi.x, cur), readlane_f(bhi.y, cur), readlane_f(bhi.z, cur), 0.0f};
        const float mine = lb_point_box(X, Y, Z, bx);
        MPA_LSTAT(st_tests);
        if (__ballot(mine <= key_dist(best))) {
          ++scans;
#ifdef MPA_LEAF_EXP  // timing experiments: the scan of a visited leaf 0 (wrong results) or 2 times
          for (int rep_ = 0; rep_ < MPA_LEAF_EXP; ++rep_) {
            asm volatile("" : "+v"(X));
#endif
#if MPA_LEAF_GATE
          // the gate: 4 matrix instructions, ~12 VALU instructions per 16 values
          const bool lo = lane < 32;
          const bool padrec = __float_as_int(tG.w) == kNoIdx;
          const float tx = padrec ? 1e15f : tG.x - gw.cx, ty = padrec ? 1e15f : tG.y - gw.cy, tz = padrec ? 1e15f : tG.z - gw.cz;
          const float tt = dist3(tx, ty, tz) + gw.Q;
          const float a01 = lo ? tx : ty, a23 = lo ? tz : tt;
          float m0, s0, m1, s1;
          {  // one query tile at a time: 16 accumulator registers live, not 32
            f32x16 acc = {0};
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, gw.b0_01, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a23, gw.b0_23, acc, 0, 0, 0);
            gate_reduce(acc, m0, s0);
          }
          __builtin_amdgcn_sched_barrier(0);
          {
            f32x16 acc = {0};
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, gw.b1_01, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a23, gw.b1_23, acc, 0, 0, 0);
            gate_reduce(acc, m1, s1);
          }
          {  // lanes 0-31 keep tile 0 and receive the upper half's tile 0; lanes 32-63 keep tile 1 and receive the lower half's
            const auto ms = __builtin_amdgcn_permlane32_swap(__float_as_int(m0), __float_as_int(m1), false, false);
            const auto ss = __builtin_amdgcn_permlane32_swap(__float_as_int(s0), __float_as_int(s1), false, false);
            m0 = __int_as_float(ms[0]), m1 = __int_as_float(ms[1]), s0 = __int_as_float(ss[0]), s1 = __int_as_float(ss[1]);
          }
          // (now, in every lane, m0 / s0 stem from target rows 8g + u and m1 / s1 from rows 8g + 4 + u of the lane's own query)
          const bool from1 = m1 < m0;
          const float Mp = from1 ? m1 : m0, Sp = __builtin_fminf(__builtin_fmaxf(m0, m1), __builtin_fminf(s0, s1));
          const int c4 = __float_as_int(Mp) & 15;
          const int code = 8 * (c4 >> 2) + (c4 & 3) + (from1 ? 4 : 0);
          const float Mf = __int_as_float(__float_as_int(Mp) & ~15) - gw.R, Sf = __int_as_float(__float_as_int(Sp) & ~15) - gw.R;
          // error bound of the leaf: centre-relative magnitudes of the query and of the leaf's farthest corner
          const float ex = __builtin_fmaxf(__builtin_fabsf(bx[0] - gw.cx), __builtin_fabsf(bx[4] - gw.cx));
          const float ey = __builtin_fmaxf(__builtin_fabsf(bx[1] - gw.cy), __builtin_fabsf(bx[5] - gw.cy));
          const float ez = __builtin_fmaxf(__builtin_fabsf(bx[2] - gw.cz), __builtin_fabsf(bx[6] - gw.cz));
          const float E = 1.5e-5f * ((gw.qq + gw.Q) + dist3(ex, ey, ez));
          const float bd = key_dist(best);
          const bool cand = has && !(Mf - E > bd);             // (negated compares: NaNs take the careful branch)
          const bool clear = Mp > 0.0f && (Sf - Mf > 4.0f * E);
          if (__ballot(cand && !clear)) {  // near-ties / non-finite values: the exact scan of the whole leaf
            MPA_LSTAT(st_exact);
            const float4 tA = rec[cur * kLeaf + (lane & 15)], tB = rec[cur * kLeaf + 16 + (lane & 15)];
            scan16(tA, X, Y, Z, best);
            scan16(tB, X, Y, Z, best);
          } else if (__ballot(cand)) {  // the leaf's nearest target of every query that wants one, exactly
            const int src = 4 * code;
            const float wx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(tG.x)));
            const float wy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(tG.y)));
            const float wz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(tG.z)));
            const unsigned wi = (unsigned)__builtin_amdgcn_ds_bpermute(src, __float_as_int(tG.w));
            const u64 k = make_key(dist3(X - wx, Y - wy, Z - wz), wi);
            best = (cand && k < best) ? k : best;
          }
#else
          scan16(tA, X, Y, Z, best);
          scan16(tB, X, Y, Z, best);
#endif
#ifdef MPA_LEAF_EXP
          }
#endif
          bound = wave_max_u((unsigned)(best >> 32));
        }
      }
      cur = nxt;
      clm = nlm;
#if MPA_LEAF_GATE
      tG = nG;
#else
      tA = nA;
      tB = nB;
#endif
    }
  };
  if constexpr (SHAPE) {
    const float* vb = g.valids + (long long)b * P;
    const bool tvalid = lane < P && vb[lane < P ? lane : 0] != 0.0f;
    const float4* pbx = reinterpret_cast<const float4*>(tpart + (long long)b * P * 8);
    const float4 plo = pbx[2 * (lane < P ? lane : 0)], phi = pbx[2 * (lane < P ? lane : 0) + 1];
    unsigned lbp = 0xffffffffu;
    {
      const float bx[8] = {plo.x, plo.y, plo.z, 0.0f, phi.x, phi.y, phi.z, 0.0f};
      if (tvalid) lbp = __float_as_uint(lb_box_box(qlo, qhi, bx));
    }
    while (done) {
      const unsigned pm = wave_min_u(lbp);
      if (!(pm <= bound) || pm > 0x7f800000u) break;
      const int pl = __builtin_ctzll(__ballot(lbp == pm));
      if (lane == pl) lbp = 0xffffffffu;
      const float bx[8] = {readlane_f(plo.x, pl), readlane_f(plo.y, pl), readlane_f(plo.z, pl), 0.0f,
                           readlane_f(phi.x, pl), readlane_f(phi.y, pl), readlane_f(phi.z, pl), 0.0f};
      const float mine = lb_point_box(X, Y, Z, bx);
      if (__ballot(mine <= key_dist(best)) == 0) continue;
      MPA_LSTAT(st_parts);
      search_part(pl);
    }
  } else {
    search_part(p);
  }
#ifdef MPA_LEAF_STATS
  if (lane == 0) {
    u64* s = g_leaf_stats[SHAPE ? 1 : 0];
    atomicAdd(&s[0], 1ull);
    atomicAdd(&s[1], (u64)st_tests);
    atomicAdd(&s[2], (u64)scans);
    atomicAdd(&s[3], (u64)st_parts);
    atomicMax(&s[4], (u64)scans);
    atomicAdd(&s[7], done ? 0ull : 1ull);
    atomicAdd(&s[5], (u64)st_exact);
    int k = 0;
    while ((2 << k) <= scans) ++k;
    atomicAdd(&s[8 + k], 1ull);
  }
#endif
  return done;
}

// this wave's queries: coordinates (infinitely far for a lane without a query) and the flat output position
__device__ __forceinline__ bool load_queries(const LeafArgs& g, int m, int k0, int dir, float& X, float& Y, float& Z,
                                             int& qidx, float4& twin) {
  const int lane = threadIdx.x & 63, k = k0 + lane;
  const bool in = k < g.Npad;
  const float4 q = g.rec[dir][(long long)m * g.Npad + (in ? k : g.Npad - 1)];
  twin = g.rec[1 - dir][(long long)m * g.Npad + (in ? k : g.Npad - 1)];  // this point's own image in the other cloud
  const bool has = in && __float_as_int(q.w) != kNoIdx;
  const float inf = __builtin_inff();
  X = has ? q.x : inf, Y = has ? q.y : inf, Z = has ? q.z : inf;
  qidx = has ? __float_as_int(q.w) : kNoIdx;
  return has;
}

template <bool SHAPE>
__device__ __forceinline__ void store_result(const LeafArgs& g, int m, int w, int dir, bool has, int qidx, u64 best) {
  const int lane = threadIdx.x & 63;
  if (has) {
    const int b = m / g.P;
    int* iout = g.idx[dir] + (SHAPE ? (long long)b * g.P * g.N : (long long)m * g.N);
    const unsigned bi = (unsigned)best;
    iout[qidx] = bi == (unsigned)kNoIdx ? -1 : (int)bi;
  }
  float s = has ? key_dist(best) : 0.0f;  // the wave's distance sum (fixed tree: deterministic)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) g.wave_sums[((long long)dir * g.total_parts + m) * g.NW + w] = s;
}

// First pass.  grid = (ceil(B*P*NW / 4), 2), block 256: one wave per 64 consecutive sorted slots of a part.
#ifndef MPA_LEAF_WAVES  // waves per SIMD the first pass is compiled for (register budget 512 / waves)
#define MPA_LEAF_WAVES 4
#endif
template <bool SHAPE>
__global__ __launch_bounds__(256, MPA_LEAF_WAVES) void leaf_search_kernel(const LeafArgs g) {
  const int dir = blockIdx.y;
  const int wid = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (wid >= g.total_parts * g.NW) return;
  const int m = wid / g.NW, w = wid % g.NW;
  if (g.valids[m] == 0.0f) return;
  if (SHAPE && g.route != nullptr && g.route[m / g.P] != 0) return;
  const int lane = threadIdx.x & 63, k0 = w * 64;
  float X, Y, Z;
  int qidx;
  float4 twin;
  const bool has = load_queries(g, m, k0, dir, X, Y, Z, qidx, twin);
  u64 best = 0;  // (a lane without a query: key 0 is never beaten)
  {
    const float d = dist3(X - twin.x, Y - twin.y, Z - twin.z);
    const u64 none = make_key(1e32f, (unsigned)kNoIdx), k = make_key(d, __float_as_uint(twin.w));
    if (has) best = k < none ? k : none;
  }
  if (__ballot(has)) {  // (wave-uniform)
    if constexpr (SHAPE) {  // padded parts: one representative target each (index pp * N)
      const int P = g.P, b = m / P;
      const float* vb = g.valids + (long long)b * P;
      const bool pad = lane < P && vb[lane < P ? lane : 0] == 0.0f;
      const float* t = g.orig[1 - dir] + 3LL * ((long long)b * P + (pad ? lane : 0)) * g.N;
      const float rx = t[0], ry = t[1], rz = t[2];
      u64 pm = __ballot(pad);
      while (pm) {
        const int pp = __builtin_ctzll(pm);
        pm &= pm - 1;
        const float d = dist3(X - readlane_f(rx, pp), Y - readlane_f(ry, pp), Z - readlane_f(rz, pp));
        const u64 k = make_key(d, (unsigned)(pp * g.N));
        best = k < best ? k : best;
      }
    }
    const bool done = leaf_search_wave<SHAPE>(g, m, k0, dir, X, Y, Z, has, best, 1, 0, kScanCap);
    if (!done) {  // hand the wave to the second pass with what it has found so far
      int slot = 0;
      if (lane == 0) slot = atomicAdd(g.heavy_count, 1);
      slot = __builtin_amdgcn_readfirstlane(slot);
      if (lane == 0) g.heavy_list[slot] = (m * g.NW + w) * 2 + dir;
      g.heavy_keys[(long long)slot * 64 + lane] = best;
      return;
    }
  }
  store_result<SHAPE>(g, m, w, dir, has, qidx, best);
}

// Second pass: the deferred waves, kSplit waves each.  Every wave starts from the saved keys (a bound that is already
// close to final), takes every kSplit-th leaf of every part it still has to look at, and the block merges the keys.
// grid = any (blocks stride over the list), block 64 * kSplit.
template <bool SHAPE>
__global__ __launch_bounds__(64 * kSplit) void leaf_search_heavy_kernel(const LeafArgs g) {
  __shared__ u64 keys[kSplit][64];
  const int count = *g.heavy_count;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int item = blockIdx.x; item < count; item += gridDim.x) {
    const int code = g.heavy_list[item];
    const int dir = code & 1, mw = code >> 1, m = mw / g.NW, w = mw % g.NW, k0 = w * 64;
    float X, Y, Z;
    int qidx;
    float4 twin;
    const bool has = load_queries(g, m, k0, dir, X, Y, Z, qidx, twin);
    u64 best = g.heavy_keys[(long long)item * 64 + lane];
    (void)leaf_search_wave<SHAPE>(g, m, k0, dir, X, Y, Z, has, best, kSplit, wv, 0x7fffffff);
    __syncthreads();  // (the previous item's readers are done with `keys`)
    keys[wv][lane] = best;
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int v = 1; v < kSplit; ++v) {
        const u64 o = keys[v][lane];
        best = o < best ? o : best;
      }
      store_result<SHAPE>(g, m, w, dir, has, qidx, best);
    }
  }
}

}  // namespace

int leaf_npad(int64_t N) {
  int p = kLeaf;
  while (p < N) p <<= 1;
  return p;
}
bool leaf_supported(int64_t P, int64_t N) { return P >= 1 && P <= 64 && N >= 1 && N <= kLeafMaxPad; }

void launch_leaf_order(const float* part_pcs, const float* valids, int64_t B, int64_t P, int64_t N, float* sorted,
                       hipStream_t s) {
  const int npad = leaf_npad(N);
  hipLaunchKernelGGL(leaf_order_kernel, dim3((unsigned)(B * P)), dim3(64), (size_t)npad * 10, s, part_pcs, valids, (int)N,
                     npad, reinterpret_cast<float4*>(sorted));
}

int64_t leaf_scratch_floats(int64_t B, int64_t P, int64_t N) {  // heavy list + keys + counters, shared by both searches
  const int64_t nw = leaf_npad(N) >= 64 ? leaf_npad(N) / 64 : 1, items = 2 * B * P * nw;
  return 16 + (B + 3) / 4 * 4 + (items + 3) / 4 * 4 + items * 64 * 2;  // counters, route, list, keys
}
int* leaf_heavy_counters(float* scratch) { return reinterpret_cast<int*>(scratch); }
int* leaf_route(float* scratch) { return reinterpret_cast<int*>(scratch) + 16; }

namespace {
constexpr float kRouteFill = 0.075f;  // parts' boxes fill at least this share of the shape's box: the grid answers
__global__ __launch_bounds__(64) void leaf_route_kernel(const float* __restrict__ valids, const float* __restrict__ pbox,
                                                        int P, int force, int* __restrict__ route) {
  const int b = blockIdx.x, p = threadIdx.x;
  if (force >= 0) {
    if (p == 0) route[b] = force;
    return;
  }
  const bool on = p < P && valids[(long long)b * P + (p < P ? p : 0)] != 0.0f;
  const float inf = __builtin_inff();
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf}, vol = 0.0f;
  if (on) {
    const float* bx = pbox + ((long long)b * P + p) * 8;
#pragma unroll
    for (int a = 0; a < 3; ++a) lo[a] = bx[a], hi[a] = bx[4 + a];
    vol = __builtin_fmaxf(hi[0] - lo[0], 0.0f) * __builtin_fmaxf(hi[1] - lo[1], 0.0f) * __builtin_fmaxf(hi[2] - lo[2], 0.0f);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = __builtin_fminf(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = __builtin_fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
    }
    vol += __shfl_xor(vol, off, 64);
  }
  const float all = (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]);
  // (non-finite or empty boxes: the comparison is false -> leaf search, which is exact for anything)
  if (p == 0) route[b] = vol >= kRouteFill * all && all > 0.0f ? 1 : 0;
}
}  // namespace

void launch_leaf_route(const float* valids, const float* pbox, int64_t B, int64_t P, int force, int* route, hipStream_t s) {
  hipLaunchKernelGGL(leaf_route_kernel, dim3((unsigned)B), dim3(64), 0, s, valids, pbox, (int)P, force, route);
}

void launch_leaf_search(bool shape, const float* valids, const LeafCloud& A, const LeafCloud& Bc, int64_t B, int64_t P,
                        int64_t N, int32_t* idx1, int32_t* idx2, float* wave_sums, float* scratch, hipStream_t s,
                        const int* route) {
  LeafArgs g;
  g.route = route;
  g.valids = valids;
  g.rec[0] = reinterpret_cast<const float4*>(A.rec);
  g.rec[1] = reinterpret_cast<const float4*>(Bc.rec);
  g.leaf[0] = A.leaf, g.leaf[1] = Bc.leaf;
  g.part[0] = A.part, g.part[1] = Bc.part;
  g.orig[0] = A.orig, g.orig[1] = Bc.orig;
  g.P = (int)P, g.N = (int)N, g.Npad = leaf_npad(N);
  g.NW = g.Npad >= 64 ? g.Npad / 64 : 1;
  g.idx[0] = idx1, g.idx[1] = idx2;
  g.wave_sums = wave_sums;
  g.total_parts = (int)(B * P);
  const int64_t items = 2 * B * P * g.NW;
  g.heavy_count = leaf_heavy_counters(scratch) + (shape ? 1 : 0);  // (zeroed by the producer of the records)
  g.heavy_list = reinterpret_cast<int*>(scratch) + 16 + (B + 3) / 4 * 4;
  g.heavy_keys = reinterpret_cast<unsigned long long*>(scratch + 16 + (B + 3) / 4 * 4 + (items + 3) / 4 * 4);
  const dim3 grid((unsigned)((B * P * g.NW + 3) / 4), 2);
  const unsigned heavy_blocks = (unsigned)(items < 4096 ? items : 4096);  // (blocks beyond the list's length leave at once)
  if (shape) {
    hipLaunchKernelGGL((leaf_search_kernel<true>), grid, dim3(256), 0, s, g);
    hipLaunchKernelGGL((leaf_search_heavy_kernel<true>), dim3(heavy_blocks), dim3(64 * kSplit), 0, s, g);
  } else {
    hipLaunchKernelGGL((leaf_search_kernel<false>), grid, dim3(256), 0, s, g);
    hipLaunchKernelGGL((leaf_search_heavy_kernel<false>), dim3(heavy_blocks), dim3(64 * kSplit), 0, s, g);
  }
}

}  // namespace mpa

#ifdef MPA_LEAF_STATS
extern "C" int mpa_debug_leaf_stats(unsigned long long* out64, int reset) {
  if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(mpa::g_leaf_stats), 64 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mpa::g_leaf_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
