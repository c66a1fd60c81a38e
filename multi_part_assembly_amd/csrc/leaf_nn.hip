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
//   * Leaf records are walked with wave-uniform addresses: they arrive through the scalar cache as SGPR operands of
//     the distance arithmetic (no LDS, no barrier, no VGPRs for targets); candidates are not visited in index order,
//     so updates are lexicographic (smaller distance, then smaller original index) — the answer of the in-order
//     strict-`<` scan.
#include "assembly_internal.h"
#include "common.h"

namespace mpa {
namespace {

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
// One block per part.  Level by level the segment [0, Npad) is halved: every segment is sorted along the widest axis
// of its own bounding box (bitonic network in LDS, (coordinate, index) keys: a strict total order, so the result is
// deterministic) and cut in the middle, down to segments of 32 slots = the leaves.  Pad slots (beyond N) carry the
// largest key and stay at the end of the last segment.  The ordering only steers speed: the search is exact for any
// permutation.  Output: sorted[m][k] = (x, y, z, original index n as int bits; -1 in pad slots) in LOCAL coordinates.
__device__ __forceinline__ unsigned orderable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void leaf_order_kernel(const float* __restrict__ pcs, const float* __restrict__ valids,
                                                          int N, int Npad, float4* __restrict__ sorted) {
  __shared__ unsigned long long key[kLeafMaxPad];
  __shared__ float px[kLeafMaxPad], py[kLeafMaxPad], pz[kLeafMaxPad];
  __shared__ float cbox[kLeafMaxPad / kLeaf][6];
  const int m = blockIdx.x;
  if (valids[m] == 0.0f) return;
  const float* src = pcs + 3LL * m * N;
  for (int n = threadIdx.x; n < Npad; n += 1024) {
    const bool in = n < N;
    px[n] = in ? src[3 * n] : 0.0f;
    py[n] = in ? src[3 * n + 1] : 0.0f;
    pz[n] = in ? src[3 * n + 2] : 0.0f;
    key[n] = in ? (unsigned long long)n : 0xffffffffffffffffull;  // low word: the point held by this slot
  }
  __syncthreads();
  for (int S = Npad; S > kLeaf; S >>= 1) {
    // boxes of the 32-slot chunks, then of the segments
    for (int k = threadIdx.x; k < Npad; k += 1024) {
      const unsigned long long e = key[k];
      const bool real = e != 0xffffffffffffffffull;
      const int n = real ? (int)(e & 0xffffffffu) : 0;
      const float inf = __builtin_inff();
      float lo[3] = {real ? px[n] : inf, real ? py[n] : inf, real ? pz[n] : inf};
      float hi[3] = {real ? px[n] : -inf, real ? py[n] : -inf, real ? pz[n] : -inf};
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          lo[a] = __builtin_fminf(lo[a], __shfl_xor(lo[a], off, 64));
          hi[a] = __builtin_fmaxf(hi[a], __shfl_xor(hi[a], off, 64));
        }
      }
      if ((k & 31) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          cbox[k >> 5][a] = lo[a];
          cbox[k >> 5][3 + a] = hi[a];
        }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < Npad; k += 1024) {
      const int seg = k / S, c0 = seg * (S / kLeaf), c1 = c0 + S / kLeaf;
      float lo[3] = {cbox[c0][0], cbox[c0][1], cbox[c0][2]}, hi[3] = {cbox[c0][3], cbox[c0][4], cbox[c0][5]};
      for (int c = c0 + 1; c < c1; ++c) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          lo[a] = __builtin_fminf(lo[a], cbox[c][a]);
          hi[a] = __builtin_fmaxf(hi[a], cbox[c][3 + a]);
        }
      }
      const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
      const int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);  // (NaN extents fall through to z: any axis is fine)
      const unsigned long long e = key[k];
      if (e != 0xffffffffffffffffull) {
        const int n = (int)(e & 0xffffffffu);
        const float v = axis == 0 ? px[n] : (axis == 1 ? py[n] : pz[n]);
        key[k] = ((unsigned long long)orderable(v) << 32) | (unsigned)n;
      }
    }
    __syncthreads();
    // ascending bitonic sort of every aligned S-segment; Npad / 2 compare-exchange pairs per stage
    for (int k = 2; k <= S; k <<= 1) {
      for (int t = threadIdx.x; t < Npad / 2; t += 1024) {  // first stage of a merge: mirror pairs
        const int half = k >> 1, blk = t / half, off = t % half;
        const int i = blk * k + off, p = blk * k + (k - 1 - off);
        const unsigned long long a = key[i], b = key[p];
        if (a > b) {
          key[i] = b;
          key[p] = a;
        }
      }
      __syncthreads();
      for (int j = k >> 2; j > 0; j >>= 1) {
        for (int t = threadIdx.x; t < Npad / 2; t += 1024) {
          const int i = (t / j) * 2 * j + (t % j), p = i + j;
          const unsigned long long a = key[i], b = key[p];
          if (a > b) {
            key[i] = b;
            key[p] = a;
          }
        }
        __syncthreads();
      }
    }
  }
  float4* out = sorted + (long long)m * Npad;
  for (int k = threadIdx.x; k < Npad; k += 1024) {
    const unsigned long long e = key[k];
    const bool real = e != 0xffffffffffffffffull;
    const int n = real ? (int)(e & 0xffffffffu) : 0;
    out[k] = make_float4(px[n], py[n], pz[n], __int_as_float(real ? n : -1));
  }
}

// ---- 2. the search --------------------------------------------------------------------------------------------------------
// Scan the 32 records of one leaf (wave-uniform address -> scalar loads), exact arithmetic, lexicographic update.
__device__ __forceinline__ void scan_leaf(const float4* __restrict__ rec, float X, float Y, float Z, float& best,
                                          int& bidx) {
#pragma unroll
  for (int c = 0; c < kLeaf; c += 8) {
    float4 r[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) r[t] = rec[c + t];
    float d[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) d[t] = dist3(X - r[t].x, Y - r[t].y, Z - r[t].z);
    // (fminf ignores NaNs — wanted: a NaN candidate can never win)
    const float a = __builtin_fminf(__builtin_fminf(d[0], d[1]), d[2]);
    const float b = __builtin_fminf(__builtin_fminf(d[3], d[4]), d[5]);
    const float cmin = __builtin_fminf(__builtin_fminf(a, b), __builtin_fminf(d[6], d[7]));
    if (cmin <= best) {  // rare: an improvement, or a tie that may carry a lower index
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int ti = __float_as_int(r[t].w);
        if (d[t] < best || (d[t] == best && ti < bidx)) {
          best = d[t];
          bidx = ti;
        }
      }
    }
  }
}

// Records qrec / trec [B*P][Npad] float4 (x, y, z, index: SHAPE p * N + n, else n; pad slots (inf, inf, inf, kNoIdx)),
// leaf boxes tleaf / qleaf [B*P][Npad / 32][8] (lo xyz, -, hi xyz, -), part boxes tpart [B*P][8]; reps: the target cloud
// in original order [B, P, N, 3] (SHAPE: the padded parts' representatives).  grid = (B*P*tilesq, 2), block 256: a
// block is 256 consecutive sorted slots of part m; dir 0: cloud A queries against cloud B targets.
template <bool SHAPE>
__global__ __launch_bounds__(256) void leaf_search_kernel(
    const float* __restrict__ valids, const float4* __restrict__ recA, const float4* __restrict__ recB,
    const float* __restrict__ leafA, const float* __restrict__ leafB, const float* __restrict__ partA,
    const float* __restrict__ partB, const float* __restrict__ origA, const float* __restrict__ origB, int P, int N,
    int Npad, int tilesq, int* __restrict__ idx1, int* __restrict__ idx2, float* __restrict__ tile_sums) {
  __shared__ float red[4];
  const int bid = blockIdx.x, dir = blockIdx.y;
  const int m = bid / tilesq, tile = bid % tilesq;
  if (valids[m] == 0.0f) return;
  const int b = m / P, p = m % P;
  const int NL = Npad / kLeaf;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float4* qrec = dir == 0 ? recA : recB;
  const float4* trec = dir == 0 ? recB : recA;
  const float* qleaf = dir == 0 ? leafA : leafB;
  const float* tleaf = dir == 0 ? leafB : leafA;
  const float* tpart = dir == 0 ? partB : partA;
  const int k0 = (tile * 4 + wave) * 64;  // first sorted slot of this wave
  float X, Y, Z, best = 0.0f;
  int qidx = kNoIdx, bidx = kNoIdx;
  {
    const float inf = __builtin_inff();
    X = Y = Z = inf;
  }
  bool has = false;
  if (k0 < Npad) {  // (wave-uniform)
    const int k = k0 + lane;
    const bool in = k < Npad;
    const float4 q = qrec[(long long)m * Npad + (in ? k : Npad - 1)];
    has = in && __float_as_int(q.w) != kNoIdx;
    if (has) {
      X = q.x, Y = q.y, Z = q.z;
      qidx = __float_as_int(q.w);
      // the twin: this point's own image in the other cloud
      const float4 t = trec[(long long)m * Npad + k];
      best = 1e32f;
      const float d = dist3(X - t.x, Y - t.y, Z - t.z);
      if (d < best || d == best) {
        best = d;
        bidx = __float_as_int(t.w);
      }
    }
  }
  if (__ballot(has)) {  // (wave-uniform)
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
    unsigned bound = wave_max_u(__float_as_uint(best));
    // the leaves of target part tp, nearest first
    auto search_part = [&](int tp) {
      const float* lbx = tleaf + ((long long)(b * P + tp) * NL) * 8;
      unsigned lbl = 0xffffffffu;
      if (lane < NL) lbl = __float_as_uint(lb_box_box(qlo, qhi, lbx + lane * 8));
      for (;;) {
        const unsigned lm = wave_min_u(lbl);
        if (!(lm <= bound) || lm > 0x7f800000u) break;
        const int ll = __builtin_ctzll(__ballot(lbl == lm));
        if (lane == ll) lbl = 0xffffffffu;
        const float mine = lb_point_box(X, Y, Z, lbx + ll * 8);
        if (__ballot(mine <= best) == 0) continue;
        scan_leaf(trec + ((long long)(b * P + tp) * Npad + ll * kLeaf), X, Y, Z, best, bidx);
        bound = wave_max_u(__float_as_uint(best));
      }
    };
    if constexpr (SHAPE) {
      const float* vb = valids + (long long)b * P;
      const bool tvalid = lane < P && vb[lane < P ? lane : 0] != 0.0f;
      unsigned lbp = 0xffffffffu;
      if (tvalid) lbp = __float_as_uint(lb_box_box(qlo, qhi, tpart + (long long)(b * P + lane) * 8));
      for (;;) {
        const unsigned pm = wave_min_u(lbp);
        if (!(pm <= bound) || pm > 0x7f800000u) break;
        const int pl = __builtin_ctzll(__ballot(lbp == pm));
        if (lane == pl) lbp = 0xffffffffu;
        const float mine = lb_point_box(X, Y, Z, tpart + (long long)(b * P + pl) * 8);
        if (__ballot(mine <= best) == 0) continue;
        search_part(pl);
      }
      // padded parts: one representative target each (index pp * N), in part order
      const float* tcloud = (dir == 0 ? origB : origA) + 3LL * b * P * N;
      float rx = 0.0f, ry = 0.0f, rz = 0.0f;
      const bool pad = lane < P && !tvalid;
      if (pad) {
        const float* t = tcloud + 3LL * lane * N;
        rx = t[0], ry = t[1], rz = t[2];
      }
      unsigned long long pm = __ballot(pad);
      while (pm) {
        const int pp = __builtin_ctzll(pm);
        pm &= pm - 1;
        const float tx = __builtin_amdgcn_readlane(rx, pp), ty = __builtin_amdgcn_readlane(ry, pp),
                    tz = __builtin_amdgcn_readlane(rz, pp);
        const float d = dist3(X - tx, Y - ty, Z - tz);
        const int ti = pp * N;
        if (d < best || (d == best && ti < bidx)) {
          best = d;
          bidx = ti;
        }
      }
    } else {
      search_part(p);
    }
    if (has) {
      int* iout = (dir == 0 ? idx1 : idx2) + (SHAPE ? (long long)b * P * N : (long long)m * N);
      iout[qidx] = bidx == kNoIdx ? -1 : bidx;
    }
  }
  // the block's distance sum (fixed tree: deterministic)
  float s = has ? best : 0.0f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[(long long)dir * gridDim.x + bid] = (red[0] + red[1]) + (red[2] + red[3]);
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
  hipLaunchKernelGGL(leaf_order_kernel, dim3((unsigned)(B * P)), dim3(1024), 0, s, part_pcs, valids, (int)N, leaf_npad(N),
                     reinterpret_cast<float4*>(sorted));
}

void launch_leaf_search(bool shape, const float* valids, const LeafCloud& A, const LeafCloud& Bc, int64_t B, int64_t P,
                        int64_t N, int tilesq, int32_t* idx1, int32_t* idx2, float* tile_sums, hipStream_t s) {
  const int Npad = leaf_npad(N);
  const dim3 grid((unsigned)(B * P * tilesq), 2);
  if (shape)
    hipLaunchKernelGGL((leaf_search_kernel<true>), grid, dim3(256), 0, s, valids, reinterpret_cast<const float4*>(A.rec),
                       reinterpret_cast<const float4*>(Bc.rec), A.leaf, Bc.leaf, A.part, Bc.part, A.orig, Bc.orig, (int)P,
                       (int)N, Npad, tilesq, idx1, idx2, tile_sums);
  else
    hipLaunchKernelGGL((leaf_search_kernel<false>), grid, dim3(256), 0, s, valids, reinterpret_cast<const float4*>(A.rec),
                       reinterpret_cast<const float4*>(Bc.rec), A.leaf, Bc.leaf, A.part, Bc.part, A.orig, Bc.orig, (int)P,
                       (int)N, Npad, tilesq, idx1, idx2, tile_sums);
}

}  // namespace mpa
