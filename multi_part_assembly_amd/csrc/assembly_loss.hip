// Fused geometric-assembly loss (forward + backward) for gfx950.
//
// Replaces, for the geometric datasets, the whole loss half of BaseModel._calc_loss of the reference
// (multi_part_assembly/models/modules/base_model.py:240-314): trans_l2_loss, rot_cosine_loss,
// rot_points_l2_loss, rot_points_cd_loss and shape_cd_loss (utils/loss.py:22-35,59-202) with their
// rot_pc / transform_pc (utils/transforms.py:199-244) and two chamfer_distance calls, and the whole
// autograd graph behind them.  The reference materialises ~20 [B,P,N,3|4] tensors, runs the Chamfer
// kernel over every padded slot and back-propagates through atomics; here:
//
//   pose kernel      one pass over part_pcs writes the four transformed clouds of the VALID parts
//                    (pred/GT rotation, pred/GT rotation+translation), one representative point for
//                    every padded part (all of whose points coincide after the 1e3 fill) and the
//                    per-part sum of |R1 p - R2 p|^2;
//   part-CD kernel   exact NN of each valid part against its own GT copy (chamfer_core.h scan);
//   shape-CD kernel  exact NN of each valid part's points against the whole other shape: the valid
//                    parts' points in index order plus ONE candidate per padded part — identical
//                    arg-mins to scanning all P*N slots (duplicates never beat the first copy under
//                    the strict `<` rule), ~3x fewer pair evaluations at the everyday part counts;
//                    padded QUERY points are skipped: their distances are multiplied by 0 by the loss;
//   finalize kernel  the five [B] loss terms from per-part partial sums;
//   backward kernel  one block per part gathers every contribution to (d/dquat, d/dtrans) of that
//                    part — query side, matched-target side (by scanning the index arrays) — and
//                    reduces through the quaternion Jacobian: no per-point gradient tensors, no
//                    atomics, deterministic.
//
// Loss term order everywhere: 0 trans_loss, 1 rot_pt_cd_loss, 2 transform_pt_cd_loss, 3 rot_loss,
// 4 rot_pt_l2_loss (the names of base_model.py:283-298).
#include <stdlib.h>

#include "assembly_internal.h"
#include "chamfer_core.h"
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMinTile = 64 * 2;            // smallest query tile any scan variant uses (workspace sizing)
constexpr float kPadFill = 1e3f;            // utils/loss.py:175

// XCD-aware block -> (sample, block-within-sample) map.  gfx950 dispatches workgroup i to XCD i % 8 and
// every XCD has a private 4 MiB L2; all blocks that walk the same sample's target cloud (<= 480 KB for
// both shapes) are therefore steered to ONE XCD: XCD x serves samples x, x+8, x+16, ...  Falls back
// to the plain order when the batch is not a multiple of 8 (placement only affects speed).
__device__ __forceinline__ int xcd_remap(int bid, int per_sample, int num_samples) {
  if (num_samples % 8 != 0 || per_sample < 0) return bid;
  const int x = bid & 7, k = bid >> 3;
  return (x + 8 * (k / per_sample)) * per_sample + (k % per_sample);
}

struct Quat {
  float w, x, y, z;
};

__device__ __forceinline__ Quat raw_mul(const Quat a, const Quat b) {
  Quat o;
  o.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  o.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  o.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  o.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return o;
}

// pytorch3d quaternion_apply, term by term (see pose.hip) — bit-identical to the reference CPU path.
__device__ __forceinline__ void quat_apply(const Quat q, float px, float py, float pz, float& ox,
                                           float& oy, float& oz) {
  const Quat p{0.0f, px, py, pz};
  const Quat c{q.w * 1.0f, q.x * -1.0f, q.y * -1.0f, q.z * -1.0f};
  const Quat r = raw_mul(raw_mul(q, p), c);
  ox = r.x;
  oy = r.y;
  oz = r.z;
}

__device__ __forceinline__ Quat load_quat(const float* q) { return Quat{q[0], q[1], q[2], q[3]}; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over the block (fixed tree -> deterministic); result valid in thread 0.  `red` = kThreads/64 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.0f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) s += red[w];
  }
  return s;
}

// ---- pose kernel ----------------------------------------------------------------------------------
// grid = B*P blocks.  partial layout: [B*P][5] = {l2sum, cd1sum, cd2sum, scd1sum, scd2sum}.
__global__ __launch_bounds__(kThreads) void assembly_pose_kernel(
    const float* __restrict__ pcs, const float* __restrict__ valids, const float* __restrict__ q1,
    const float* __restrict__ t1, const float* __restrict__ q2, const float* __restrict__ t2, int N,
    int fill_pads, float* __restrict__ R1, float* __restrict__ R2, float* __restrict__ S1,
    float* __restrict__ S2, float* __restrict__ partial, float* __restrict__ bbox,
    unsigned* __restrict__ ticket) {
  __shared__ float red[kThreads / 64];
  __shared__ float box[kThreads / 64][12];
  const int m = blockIdx.x;
  if (m == 0 && threadIdx.x == 0 && ticket != nullptr) *ticket = 0u;  // of the grid sorts' "last block" election
  const Quat qa = load_quat(q1 + 4 * m), qb = load_quat(q2 + 4 * m);
  const float ta0 = t1[3 * m], ta1 = t1[3 * m + 1], ta2 = t1[3 * m + 2];
  const float tb0 = t2[3 * m], tb1 = t2[3 * m + 1], tb2 = t2[3 * m + 2];
  const long long base = 3LL * m * N;
  if (valids[m] == 0.0f) {
    // every point of a padded part is (fill,fill,fill): one representative suffices for the search
    const int count = fill_pads ? N : 1;
    for (int n = threadIdx.x; n < count; n += kThreads) {
      float ax, ay, az, bx, by, bz;
      quat_apply(qa, kPadFill, kPadFill, kPadFill, ax, ay, az);
      quat_apply(qb, kPadFill, kPadFill, kPadFill, bx, by, bz);
      const long long o = base + 3LL * n;
      S1[o] = ax + ta0; S1[o + 1] = ay + ta1; S1[o + 2] = az + ta2;
      S2[o] = bx + tb0; S2[o + 1] = by + tb1; S2[o + 2] = bz + tb2;
    }
    if (threadIdx.x < 5) partial[5 * m + threadIdx.x] = 0.0f;
    return;
  }
  float l2 = 0.0f;
  // bounding boxes of the part in the two shapes (for the grid of the whole-shape search): lo1, lo2, hi1, hi2
  const float inf = __builtin_inff();
  float bb[12] = {inf, inf, inf, inf, inf, inf, -inf, -inf, -inf, -inf, -inf, -inf};
  for (int n = threadIdx.x; n < N; n += kThreads) {
    const long long o = base + 3LL * n;
    const float px = pcs[o], py = pcs[o + 1], pz = pcs[o + 2];
    float ax, ay, az, bx, by, bz;
    quat_apply(qa, px, py, pz, ax, ay, az);
    quat_apply(qb, px, py, pz, bx, by, bz);
    R1[o] = ax; R1[o + 1] = ay; R1[o + 2] = az;
    R2[o] = bx; R2[o + 1] = by; R2[o + 2] = bz;
    const float s1[3] = {ax + ta0, ay + ta1, az + ta2}, s2[3] = {bx + tb0, by + tb1, bz + tb2};
    S1[o] = s1[0]; S1[o + 1] = s1[1]; S1[o + 2] = s1[2];
    S2[o] = s2[0]; S2[o + 1] = s2[1]; S2[o + 2] = s2[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      bb[k] = __builtin_fminf(bb[k], s1[k]);
      bb[3 + k] = __builtin_fminf(bb[3 + k], s2[k]);
      bb[6 + k] = __builtin_fmaxf(bb[6 + k], s1[k]);
      bb[9 + k] = __builtin_fmaxf(bb[9 + k], s2[k]);
    }
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    l2 += (dx * dx + dy * dy) + dz * dz;
  }
  if (bbox != nullptr) {
#pragma unroll
    for (int k = 0; k < 12; ++k) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(bb[k], off, 64);
        bb[k] = k < 6 ? __builtin_fminf(bb[k], o) : __builtin_fmaxf(bb[k], o);
      }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int k = 0; k < 12; ++k) box[threadIdx.x >> 6][k] = bb[k];
    }
  }
  const float s = block_sum(l2, red);  // (its barrier also publishes `box`)
  if (threadIdx.x == 0) partial[5 * m + 0] = s;
  if (bbox != nullptr && threadIdx.x < 12) {
    float v = box[0][threadIdx.x];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w)
      v = threadIdx.x < 6 ? __builtin_fminf(v, box[w][threadIdx.x]) : __builtin_fmaxf(v, box[w][threadIdx.x]);
    bbox[12LL * m + threadIdx.x] = v;
  }
}

// ---- pose kernel of the leaf search (leaf_nn.hip) ------------------------------------------------------
// Same outputs as assembly_pose_kernel, but the part's points are walked in the k-d order of `sorted`
// ([B*P][Npad] float4: local x, y, z, original index n | -1, leaf_nn.hip) and every cloud additionally leaves what
// the leaf search reads: records (x, y, z, index) in that order — index n for the rotation-only clouds (per-part
// Chamfer), p * N + n for the whole shapes — one box per leaf of 32 slots (a half wave) and one per part.  The boxes
// of the translated clouds are the rotated clouds' boxes plus the translation: fl(a + t) is monotone in a, so this
// IS the box of the stored translated points.
struct LeafOut {
  float4* rec[4];   // R1, R2, S1, S2
  float* leaf[4];
  float* part[4];
};

__global__ __launch_bounds__(kThreads) void assembly_pose_leaf_kernel(
    const float4* __restrict__ sorted, const float* __restrict__ valids, const float* __restrict__ q1,
    const float* __restrict__ t1, const float* __restrict__ q2, const float* __restrict__ t2, int P, int N, int Npad,
    int fill_pads, float* __restrict__ R1, float* __restrict__ R2, float* __restrict__ S1, float* __restrict__ S2,
    float* __restrict__ partial, LeafOut out, int* __restrict__ heavy_counters, float* __restrict__ bbox,
    unsigned* __restrict__ ticket) {
  __shared__ float red[kThreads / 64];
  __shared__ float box[kThreads / 32][12];
  const int m = blockIdx.x, p = m % P;
  if (m == 0 && threadIdx.x < 2) heavy_counters[threadIdx.x] = 0;  // of the two searches' second passes
  if (m == 0 && threadIdx.x == 0 && ticket != nullptr) *ticket = 0u;  // of the grid sorts' "last block" election
  const Quat qa = load_quat(q1 + 4 * m), qb = load_quat(q2 + 4 * m);
  const float ta[3] = {t1[3 * m], t1[3 * m + 1], t1[3 * m + 2]};
  const float tb[3] = {t2[3 * m], t2[3 * m + 1], t2[3 * m + 2]};
  const long long base = 3LL * m * N;
  if (valids[m] == 0.0f) {
    const int count = fill_pads ? N : 1;
    for (int n = threadIdx.x; n < count; n += kThreads) {
      float ax, ay, az, bx, by, bz;
      quat_apply(qa, kPadFill, kPadFill, kPadFill, ax, ay, az);
      quat_apply(qb, kPadFill, kPadFill, kPadFill, bx, by, bz);
      const long long o = base + 3LL * n;
      S1[o] = ax + ta[0]; S1[o + 1] = ay + ta[1]; S1[o + 2] = az + ta[2];
      S2[o] = bx + tb[0]; S2[o + 1] = by + tb[1]; S2[o + 2] = bz + tb[2];
    }
    if (threadIdx.x < 5) partial[5 * m + threadIdx.x] = 0.0f;
    return;
  }
  const float inf = __builtin_inff();
  const int NL = Npad / 32;
  float l2 = 0.0f;
  float pb[12] = {inf, inf, inf, inf, inf, inf, -inf, -inf, -inf, -inf, -inf, -inf};  // lo a, lo b, hi a, hi b (this half wave's leaves)
  for (int k0 = 0; k0 < Npad; k0 += kThreads) {
    const int k = k0 + threadIdx.x;  // (Npad is a power of two >= 32: a half wave is wholly inside or outside)
    const bool in = k < Npad;
    const float4 r = sorted[(long long)m * Npad + (in ? k : 0)];
    const int n = __float_as_int(r.w);
    const bool real = in && n >= 0;
    float ax, ay, az, bx, by, bz;
    quat_apply(qa, r.x, r.y, r.z, ax, ay, az);
    quat_apply(qb, r.x, r.y, r.z, bx, by, bz);
    const float s1[3] = {ax + ta[0], ay + ta[1], az + ta[2]}, s2[3] = {bx + tb[0], by + tb[1], bz + tb[2]};
    if (real) {
      const long long o = base + 3LL * n;
      R1[o] = ax; R1[o + 1] = ay; R1[o + 2] = az;
      R2[o] = bx; R2[o + 1] = by; R2[o + 2] = bz;
      S1[o] = s1[0]; S1[o + 1] = s1[1]; S1[o + 2] = s1[2];
      S2[o] = s2[0]; S2[o + 1] = s2[1]; S2[o + 2] = s2[2];
      const float dx = ax - bx, dy = ay - by, dz = az - bz;
      l2 += (dx * dx + dy * dy) + dz * dz;
    }
    if (in) {
      const long long e = (long long)m * Npad + k;
      const float none = __int_as_float(0x7fffffff);
      out.rec[0][e] = real ? make_float4(ax, ay, az, __int_as_float(n)) : make_float4(inf, inf, inf, none);
      out.rec[1][e] = real ? make_float4(bx, by, bz, __int_as_float(n)) : make_float4(inf, inf, inf, none);
      out.rec[2][e] = real ? make_float4(s1[0], s1[1], s1[2], __int_as_float(p * N + n)) : make_float4(inf, inf, inf, none);
      out.rec[3][e] = real ? make_float4(s2[0], s2[1], s2[2], __int_as_float(p * N + n)) : make_float4(inf, inf, inf, none);
    }
    float bb[12] = {real ? ax : inf,  real ? ay : inf,  real ? az : inf,  real ? bx : inf,  real ? by : inf,  real ? bz : inf,
                    real ? ax : -inf, real ? ay : -inf, real ? az : -inf, real ? bx : -inf, real ? by : -inf, real ? bz : -inf};
#pragma unroll
    for (int c = 0; c < 12; ++c) {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        const float o = __shfl_xor(bb[c], off, 64);
        bb[c] = c < 6 ? __builtin_fminf(bb[c], o) : __builtin_fmaxf(bb[c], o);
      }
      pb[c] = c < 6 ? __builtin_fminf(pb[c], bb[c]) : __builtin_fmaxf(pb[c], bb[c]);
    }
    if (in && (threadIdx.x & 31) == 0) {
      const long long lf = ((long long)m * NL + (k >> 5)) * 8;
      float4* la = reinterpret_cast<float4*>(out.leaf[0] + lf);
      float4* lb = reinterpret_cast<float4*>(out.leaf[1] + lf);
      float4* lc = reinterpret_cast<float4*>(out.leaf[2] + lf);
      float4* ld = reinterpret_cast<float4*>(out.leaf[3] + lf);
      la[0] = make_float4(bb[0], bb[1], bb[2], 0.0f);
      la[1] = make_float4(bb[6], bb[7], bb[8], 0.0f);
      lb[0] = make_float4(bb[3], bb[4], bb[5], 0.0f);
      lb[1] = make_float4(bb[9], bb[10], bb[11], 0.0f);
      lc[0] = make_float4(bb[0] + ta[0], bb[1] + ta[1], bb[2] + ta[2], 0.0f);
      lc[1] = make_float4(bb[6] + ta[0], bb[7] + ta[1], bb[8] + ta[2], 0.0f);
      ld[0] = make_float4(bb[3] + tb[0], bb[4] + tb[1], bb[5] + tb[2], 0.0f);
      ld[1] = make_float4(bb[9] + tb[0], bb[10] + tb[1], bb[11] + tb[2], 0.0f);
    }
  }
  if ((threadIdx.x & 31) == 0) {
#pragma unroll
    for (int c = 0; c < 12; ++c) box[threadIdx.x >> 5][c] = pb[c];
  }
  const float s = block_sum(l2, red);  // (its barrier also publishes `box`)
  if (threadIdx.x == 0) partial[5 * m + 0] = s;
  if (threadIdx.x < 12) {
    const int c = threadIdx.x;
    float v = box[0][c];
#pragma unroll
    for (int w = 1; w < kThreads / 32; ++w) v = c < 6 ? __builtin_fminf(v, box[w][c]) : __builtin_fmaxf(v, box[w][c]);
    // c: 0-2 lo a, 3-5 lo b, 6-8 hi a, 9-11 hi b -> box layout (lo xyz, -, hi xyz, -)
    // (select chains: a runtime index into ta / tb / out.part sends them to scratch memory)
    const int cloud = (c / 3) & 1, hi = c / 6, ax = c % 3;
    const float tA = ax == 0 ? ta[0] : (ax == 1 ? ta[1] : ta[2]), tB = ax == 0 ? tb[0] : (ax == 1 ? tb[1] : tb[2]);
    float* pr = cloud == 0 ? out.part[0] : out.part[1];
    float* ps = cloud == 0 ? out.part[2] : out.part[3];
    pr[8LL * m + 4 * hi + ax] = v;
    ps[8LL * m + 4 * hi + ax] = v + (cloud == 0 ? tA : tB);
    // the grid search's per-part boxes (lo1, lo2, hi1, hi2 of the translated clouds): c is already in that order
    if (bbox != nullptr) bbox[12LL * m + c] = v + (cloud == 0 ? tA : tB);
  }
}

// ---- NN kernels -------------------------------------------------------------------------------------
// Work decomposition: one block = 64*Q QUERY points of one valid part, shared by the block's 4 waves;
// the TARGET set is split evenly between the waves (chunk-aligned), each wave runs the
// chamfer_core.h scan on its share and the four partial (distance, index) minima are merged through
// LDS with the lexicographic rule (smaller distance, then smaller index) — exactly the result of one
// in-order strict-`<` scan.  Compared with "every wave owns its own queries and walks all targets"
// this gives 4x more, 4x shorter blocks, so samples with many parts (long target lists) no longer
// dominate the tail of the launch.
template <int Q, int MODE>
__device__ __forceinline__ void load_queries(mpa::NNScan<Q, MODE>& scan, const float* __restrict__ qpts,
                                             int N, int qbase) {
  scan.init();
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = qbase + q * 64 + lane;
    const int ic = i < N ? i : N - 1;
    scan.set_query(q, qpts[3 * ic], qpts[3 * ic + 1], qpts[3 * ic + 2]);
  }
}

// Scan this wave's share [lo, hi) of the compacted target list made of the valid segments
// [p*N, (p+1)*N), p in [0, P) (vb == nullptr: every segment valid); wave 0 also takes the padded
// parts' representatives.  Positions are counted over valid segments only.
template <int Q, int MODE>
__device__ __forceinline__ void scan_share(mpa::NNScan<Q, MODE>& scan, const float* __restrict__ tb,
                                           const float* __restrict__ vb, int P, int N, int lo, int hi,
                                           bool take_reps) {
  int pos = 0;
  for (int p = 0; p < P; ++p) {  // wave-uniform walk, in index order
    if (vb == nullptr || vb[p] != 0.0f) {
      const int a = lo > pos ? lo - pos : 0;
      const int b = hi - pos < N ? hi - pos : N;
      if (a < b) scan.scan_range(tb, p * N + a, p * N + b, 0);
      pos += N;
    } else if (take_reps) {
      scan.scan_one(tb, p * N, p * N);
    }
  }
}

// Merge the 4 waves' results; thread t < 64*Q ends up owning query qbase + t.  Returns that query's
// distance (0 for threads without a query) and stores its index.
template <int Q, int MODE>
__device__ __forceinline__ float merge_and_store(const mpa::NNScan<Q, MODE>& scan, int N, int qbase,
                                                 int* __restrict__ idx_out, float* sm_d, int* sm_i) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    sm_d[wave * (64 * Q) + q * 64 + lane] = scan.best[q];
    sm_i[wave * (64 * Q) + q * 64 + lane] = scan.bidx[q];
  }
  __syncthreads();
  float out = 0.0f;
  for (int t = threadIdx.x; t < 64 * Q; t += kThreads) {
    float d = sm_d[t];
    int i = sm_i[t];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) {
      const float dw = sm_d[w * (64 * Q) + t];
      const int iw = sm_i[w * (64 * Q) + t];
      // a wave that found nothing keeps (1e32, -1): never preferred over a real candidate
      const bool better = dw < d || (dw == d && iw >= 0 && (i < 0 || iw < i));
      d = better ? dw : d;
      i = better ? iw : i;
    }
    if (qbase + t < N) {
      idx_out[qbase + t] = i;
      out += d;
    }
  }
  return out;
}

// grid = (B*P*tiles, 2) with tiles = ceil(N / (64*Q)).  SHAPE = false: per-part Chamfer (part m of cloud
// A against part m of cloud B).  SHAPE = true: part m's points against the sample's whole other shape.
template <int Q, int MODE, bool SHAPE>
__global__ __launch_bounds__(kThreads) void assembly_nn_kernel(
    const float* __restrict__ valids, const float* __restrict__ C1, const float* __restrict__ C2, int B,
    int P, int N, int tiles, int remap, int* __restrict__ idx1, int* __restrict__ idx2,
    float* __restrict__ tile_sums) {
  __shared__ float sm_d[kThreads / 64 * 64 * Q];
  __shared__ int sm_i[kThreads / 64 * 64 * Q];
  __shared__ float red[kThreads / 64];
  const int bid = xcd_remap(blockIdx.x, remap ? P * tiles : -1, B);
  const int m = bid / tiles, tile = bid % tiles, dir = blockIdx.y;
  if (valids[m] == 0.0f) return;
  const int b = m / P;
  // readfirstlane: the wave index as a scalar, so that this wave's target range — and every target address — is
  // provably wave-uniform and the scan's loads go through the scalar cache (the compiler emits per-lane vector
  // loads of the same address otherwise)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float* qa = (dir == 0 ? C1 : C2) + 3LL * m * N;
  const float* tb = (dir == 0 ? C2 : C1) + (SHAPE ? 3LL * b * P * N : 3LL * m * N);
  const float* vb = SHAPE ? valids + (long long)b * P : nullptr;
  int total = N;
  if (SHAPE) {
    total = 0;
    for (int p = 0; p < P; ++p) total += vb[p] != 0.0f ? N : 0;
  }
  // even, chunk-aligned split of the valid targets between the 4 waves
  int per = (total + kThreads / 64 - 1) / (kThreads / 64);
  per = (per + mpa::kScanChunk - 1) / mpa::kScanChunk * mpa::kScanChunk;
  const int lo = wave * per < total ? wave * per : total;
  const int hi = lo + per < total ? lo + per : total;
  mpa::NNScan<Q, MODE> scan;
  load_queries<Q, MODE>(scan, qa, N, tile * 64 * Q);
  scan_share<Q, MODE>(scan, tb, vb, SHAPE ? P : 1, N, lo, hi, wave == 0);
  const float s = merge_and_store<Q, MODE>(scan, N, tile * 64 * Q, (dir == 0 ? idx1 : idx2) + (long long)m * N,
                                           sm_d, sm_i);
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) tile_sums[((long long)dir * gridDim.x) + bid] = tot;
}

// ---- finalize ---------------------------------------------------------------------------------------
// grid = B blocks of 64 threads.  losses [5][B].
__global__ __launch_bounds__(64) void assembly_finalize_kernel(
    const float* __restrict__ valids, const float* __restrict__ q1, const float* __restrict__ t1,
    const float* __restrict__ q2, const float* __restrict__ t2, const float* __restrict__ partial,
    const float* __restrict__ part_tiles, const float* __restrict__ shape_tiles, int B, int P, int N,
    int tiles_p, int tiles_s, int training, float* __restrict__ losses, const float* __restrict__ shape_tiles_grid,
    int tiles_grid, const int* __restrict__ route) {
  const int b = blockIdx.x, p = threadIdx.x;
  float v = 0.0f, trans = 0.0f, cosine = 0.0f, cd = 0.0f, l2 = 0.0f, scd_slots = 0.0f, scd_parts = 0.0f;
  if (p < P) {
    const int m = b * P + p;
    v = valids[m];
    if (v != 0.0f) {
      const float dx = t1[3 * m] - t2[3 * m], dy = t1[3 * m + 1] - t2[3 * m + 1], dz = t1[3 * m + 2] - t2[3 * m + 2];
      trans = ((dx * dx + dy * dy) + dz * dz) * v;
      const float dot = q1[4 * m] * q2[4 * m] + q1[4 * m + 1] * q2[4 * m + 1] + q1[4 * m + 2] * q2[4 * m + 2] +
                        q1[4 * m + 3] * q2[4 * m + 3];
      cosine = (1.0f - __builtin_fabsf(dot)) * v;
      float c1 = 0.0f, c2 = 0.0f, s1 = 0.0f, s2 = 0.0f;
      // (the two searches may cut a part into different numbers of tiles: brute force / grid vs leaf search)
      const long long nblk_p = (long long)B * P * tiles_p, nblk_s = (long long)B * P * tiles_s;
      for (int t = 0; t < tiles_p; ++t) {
        c1 += part_tiles[(long long)m * tiles_p + t];
        c2 += part_tiles[nblk_p + (long long)m * tiles_p + t];
      }
      if (route != nullptr && route[b] != 0) {  // this sample's whole-shape sums come from the grid search
        const long long nblk_g = (long long)B * P * tiles_grid;
        for (int t = 0; t < tiles_grid; ++t) {
          s1 += shape_tiles_grid[(long long)m * tiles_grid + t];
          s2 += shape_tiles_grid[nblk_g + (long long)m * tiles_grid + t];
        }
      } else {
        for (int t = 0; t < tiles_s; ++t) {
          s1 += shape_tiles[(long long)m * tiles_s + t];
          s2 += shape_tiles[nblk_s + (long long)m * tiles_s + t];
        }
      }
      const float inv_n = 1.0f / (float)N;
      cd = (c1 * inv_n + c2 * inv_n) * v;                  // mean_N d1 + mean_N d2   (loss.py:132)
      l2 = (partial[5 * m] * inv_n) * v;                    // mean_N |R1p - R2p|^2    (loss.py:105)
      scd_slots = (s1 + s2) * v;                            // sum of v*d over the part's slots
      scd_parts = ((s1 + s2) * inv_n) * v;                  // (d1+d2).mean(-1)         (loss.py:197)
    }
  }
  const float nv = wave_sum(v);
  const float a0 = wave_sum(trans), a1 = wave_sum(cd), a3 = wave_sum(cosine), a4 = wave_sum(l2);
  const float a2s = wave_sum(scd_slots), a2p = wave_sum(scd_parts);
  if (threadIdx.x == 0) {
    losses[0 * B + b] = a0 / nv;
    losses[1 * B + b] = a1 / nv;
    losses[2 * B + b] = training ? a2s / (float)(P * N) : a2p / nv;  // loss.py:185-198
    losses[3 * B + b] = a3 / nv;
    losses[4 * B + b] = a4 / nv;
  }
}

// ---- backward ---------------------------------------------------------------------------------------
struct PoseGrad {
  float w, x, y, z, tx, ty, tz;
};

// Accumulate J(p)^T g for out = R(q) p (+ t): see pose.hip for the formulas.
__device__ __forceinline__ void accumulate(PoseGrad& a, float w, float ux, float uy, float uz, float px,
                                           float py, float pz, float gx, float gy, float gz,
                                           bool with_trans) {
  const float cx = uy * pz - uz * py, cy = uz * px - ux * pz, cz = ux * py - uy * px;  // u x p
  const float gp = gx * px + gy * py + gz * pz;
  const float gu = gx * ux + gy * uy + gz * uz;
  const float up = ux * px + uy * py + uz * pz;
  const float dx = py * gz - pz * gy, dy = pz * gx - px * gz, dz = px * gy - py * gx;  // p x g
  a.w += 2.0f * (w * gp + (gx * cx + gy * cy + gz * cz));
  a.x += 2.0f * (-gp * ux + gu * px + up * gx + w * dx);
  a.y += 2.0f * (-gp * uy + gu * py + up * gy + w * dy);
  a.z += 2.0f * (-gp * uz + gu * pz + up * gz + w * dz);
  if (with_trans) {
    a.tx += gx;
    a.ty += gy;
    a.tz += gz;
  }
}

// closed-form terms of pose component k (0..3 quaternion, 4..6 translation) of part m: translation L2, quaternion cosine
__device__ __forceinline__ float pose_closed_form(const float* __restrict__ go, const float* __restrict__ valids,
                                                  const float* __restrict__ q1, const float* __restrict__ t1,
                                                  const float* __restrict__ q2, const float* __restrict__ t2, int B,
                                                  int P, int m, int k) {
  const int b = m / P;
  const float* vb = valids + (long long)b * P;
  if (vb[m % P] == 0.0f) return 0.0f;
  float nv = 0.0f;
  for (int i = 0; i < P; ++i) nv += vb[i];
  if (k >= 4) return (go[0 * B + b] / nv) * 2.0f * (t1[3 * m + (k - 4)] - t2[3 * m + (k - 4)]);
  const float dot = q1[4 * m] * q2[4 * m] + q1[4 * m + 1] * q2[4 * m + 1] + q1[4 * m + 2] * q2[4 * m + 2] +
                    q1[4 * m + 3] * q2[4 * m + 3];
  const float sgn = dot > 0.0f ? 1.0f : (dot < 0.0f ? -1.0f : 0.0f);
  return (go[3 * B + b] / nv) * (-sgn) * q2[4 * m + k];
}

// grid = B*P blocks.  go [5][B] = d(total)/d(loss term).  Writes gq [B*P][4], gt [B*P][3] (pred pose).
__global__ __launch_bounds__(kThreads) void assembly_backward_kernel(
    const float* __restrict__ go, const float* __restrict__ pcs, const float* __restrict__ valids,
    const float* __restrict__ q1, const float* __restrict__ t1, const float* __restrict__ q2,
    const float* __restrict__ t2, const float* __restrict__ R1, const float* __restrict__ R2,
    const float* __restrict__ S1, const float* __restrict__ S2, const int* __restrict__ ip1,
    const int* __restrict__ ip2, const int* __restrict__ is1, const int* __restrict__ is2, int B,
    int P, int N, int training, float* __restrict__ gq, float* __restrict__ gt, float* __restrict__ psum) {
  // gridDim.y = S slices of the point range per part (the scans below are chains of L2 latencies: one 4-wave block
  // per part leaves the chip at ~5 waves per CU); S > 1: the 7 partial sums of a slice go to psum[(m * S + s) * 8 + k]
  // and assembly_backward_finish_kernel adds them in slice order.
  __shared__ float red[kThreads / 64][7];
  const int m = blockIdx.x, b = m / P, p = m % P;
  const int S = (int)gridDim.y, sl = (int)blockIdx.y;
  const int n_lo = (int)((long long)sl * N / S), n_hi = (int)((long long)(sl + 1) * N / S);
  const float* vb = valids + (long long)b * P;
  float nv = 0.0f;
  for (int k = 0; k < P; ++k) nv += vb[k];
  const bool valid = vb[p] != 0.0f;
  const float w = q1[4 * m], ux = q1[4 * m + 1], uy = q1[4 * m + 2], uz = q1[4 * m + 3];
  const float inv_n = 1.0f / (float)N;
  const float c_cd = 2.0f * go[1 * B + b] * inv_n / nv;
  const float c_l2 = 2.0f * go[4 * B + b] * inv_n / nv;
  const float c_s = 2.0f * (training ? go[2 * B + b] / (float)(P * N) : go[2 * B + b] * inv_n / nv);
  const long long base = 3LL * m * N, sbase = 3LL * b * P * N;
  PoseGrad acc{0, 0, 0, 0, 0, 0, 0};

  if (valid) {
    // A. this part's own points as QUERIES (part-CD dir 1, point-wise L2, shape-CD dir 1)
    for (int n = n_lo + threadIdx.x; n < n_hi; n += kThreads) {
      const long long o = base + 3LL * n;
      const float px = pcs[o], py = pcs[o + 1], pz = pcs[o + 2];
      const float ax = R1[o], ay = R1[o + 1], az = R1[o + 2];
      // a stored index of -1 (no candidate below 1e32: NaN / inf poses of a diverged step) contributes nothing,
      // like chamfer_kernel.cu:199-208 for idx = -1 — and must not be used as an address
      const int i1 = ip1[(long long)m * N + n], i2 = is1[(long long)m * N + n];
      const float k_cd = i1 >= 0 ? c_cd : 0.0f, k_s = i2 >= 0 ? c_s : 0.0f;
      const long long j = base + 3LL * (i1 >= 0 ? i1 : 0);
      float gx = k_cd * (ax - R2[j]) + c_l2 * (ax - R2[o]);
      float gy = k_cd * (ay - R2[j + 1]) + c_l2 * (ay - R2[o + 1]);
      float gz = k_cd * (az - R2[j + 2]) + c_l2 * (az - R2[o + 2]);
      accumulate(acc, w, ux, uy, uz, px, py, pz, gx, gy, gz, false);
      const long long js = sbase + 3LL * (i2 >= 0 ? i2 : 0);
      gx = k_s * (S1[o] - S2[js]);
      gy = k_s * (S1[o + 1] - S2[js + 1]);
      gz = k_s * (S1[o + 2] - S2[js + 2]);
      accumulate(acc, w, ux, uy, uz, px, py, pz, gx, gy, gz, true);
    }
    // B. this part's points as matched TARGETS of its GT copy (part-CD dir 2)
    for (int k = n_lo + threadIdx.x; k < n_hi; k += kThreads) {
      const int jraw = ip2[(long long)m * N + k], jn = jraw >= 0 ? jraw : 0;
      const float k_cd = jraw >= 0 ? c_cd : 0.0f;
      const long long o = base + 3LL * k, j = base + 3LL * jn;
      const float gx = -k_cd * (R2[o] - R1[j]), gy = -k_cd * (R2[o + 1] - R1[j + 1]),
                  gz = -k_cd * (R2[o + 2] - R1[j + 2]);
      accumulate(acc, w, ux, uy, uz, pcs[j], pcs[j + 1], pcs[j + 2], gx, gy, gz, false);
    }
  }
  // C. this part's points as matched TARGETS of any valid GT point of the sample (shape-CD dir 2).
  //    Also runs for padded parts: their representative could, in principle, be somebody's nearest.
  //    The scan is a chain of L2 latencies (one index per thread and GT part, a match is rare): the indices of UP parts
  //    are requested together.
  constexpr int UP = 4;
  for (int pp0 = 0; pp0 < P; pp0 += UP) {
    for (int k = n_lo + threadIdx.x; k < n_hi; k += kThreads) {
      int jj[UP];
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const int pp = pp0 + u < P ? pp0 + u : P - 1;
        jj[u] = is2[((long long)b * P + pp) * N + k];
      }
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const int pp = pp0 + u, j = jj[u];
        if (pp >= P || vb[pp] == 0.0f) continue;
        if (j < p * N || j >= (p + 1) * N) continue;  // not a point of this part (a range test, not an integer division)
        const long long o = 3LL * (((long long)b * P + pp) * N + k), jt = sbase + 3LL * j;
        const float gx = -c_s * (S2[o] - S1[jt]), gy = -c_s * (S2[o + 1] - S1[jt + 1]),
                    gz = -c_s * (S2[o + 2] - S1[jt + 2]);
        float px = kPadFill, py = kPadFill, pz = kPadFill;
        if (valid) {
          px = pcs[jt];
          py = pcs[jt + 1];
          pz = pcs[jt + 2];
        }
        accumulate(acc, w, ux, uy, uz, px, py, pz, gx, gy, gz, true);
      }
    }
  }

  float vals[7] = {acc.w, acc.x, acc.y, acc.z, acc.tx, acc.ty, acc.tz};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    const float s = wave_sum(vals[k]);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    float s = 0.0f;
#pragma unroll
    for (int v = 0; v < kThreads / 64; ++v) s += red[v][threadIdx.x];
    const int k = threadIdx.x;
    if (S > 1) {
      psum[((long long)m * S + sl) * 8 + k] = s;
      return;
    }
    s += pose_closed_form(go, valids, q1, t1, q2, t2, B, P, m, k);
    if (k < 4) gq[4 * m + k] = s;
    else gt[3 * m + (k - 4)] = s;
  }
}

// S > 1: one thread per (part, pose component)
__global__ void assembly_backward_finish_kernel(const float* __restrict__ go, const float* __restrict__ valids,
                                                const float* __restrict__ q1, const float* __restrict__ t1,
                                                const float* __restrict__ q2, const float* __restrict__ t2,
                                                const float* __restrict__ psum, int B, int P, int S,
                                                float* __restrict__ gq, float* __restrict__ gt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * P * 8) return;
  const int m = e >> 3, k = e & 7;
  if (k == 7) return;
  float s = 0.0f;
  for (int q = 0; q < S; ++q) s += psum[((long long)m * S + q) * 8 + k];  // slice order: deterministic
  s += pose_closed_form(go, valids, q1, t1, q2, t2, B, P, m, k);
  if (k < 4) gq[4 * m + k] = s;
  else gt[3 * m + (k - 4)] = s;
}

// floats the leaf search adds to the workspace: order + 4 record arrays (float4 per slot), 4 x leaf boxes, 4 x part boxes
int64_t leaf_workspace_floats(int64_t B, int64_t P, int64_t N) {
  if (!mpa::leaf_supported(P, N)) return 0;
  const int64_t npad = mpa::leaf_npad(N), nw = npad >= 64 ? npad / 64 : 1;
  // + per-wave distance sums (2 searches x 2 directions) + the searches' scratch
  return 20 * B * P * npad + 32 * B * P * (npad / 32) + 32 * B * P + (4 * B * P * nw + 3) / 4 * 4 +
         mpa::leaf_scratch_floats(B, P, N);
}

}  // namespace

extern "C" int mpa_assembly_loss_workspace(int64_t B, int64_t P, int64_t N, int64_t* float_elems,
                                           int64_t* int_elems) {
  MPA_REQUIRE(B >= 0 && P >= 0 && N >= 0 && float_elems && int_elems, "assembly_loss_workspace: bad args");
  const int64_t tiles = (N + kMinTile - 1) / kMinTile;
  // 4 clouds + partial[5] + 2 tile-sum arrays (2 directions each), rounded to 16 B, + grid-search scratch
  *float_elems = (4 * B * P * N * 3 + 5 * B * P + 4 * B * P * tiles + 3) / 4 * 4 + mpa::grid_workspace_floats(B, P, N) +
                 leaf_workspace_floats(B, P, N);
  *int_elems = 4 * B * P * N + mpa::grid_workspace_ints(B);
  return MPA_OK;
}

extern "C" int mpa_assembly_order_elems(int64_t B, int64_t P, int64_t N, int64_t* float_elems) {
  MPA_REQUIRE(B >= 0 && P >= 0 && N >= 0 && float_elems, "assembly_order_elems: bad args");
  *float_elems = mpa::leaf_supported(P, N) ? 4 * B * P * (int64_t)mpa::leaf_npad(N) : 0;
  return MPA_OK;
}

extern "C" int mpa_assembly_order(const float* part_pcs, const float* valids, int64_t B, int64_t P, int64_t N,
                                  float* order, void* stream) {
  MPA_REQUIRE(B >= 0 && P >= 0 && N >= 0, "assembly_order: negative size");
  if (B == 0 || !mpa::leaf_supported(P, N)) return MPA_OK;  // (nothing to order: the loss takes its grid search)
  MPA_REQUIRE(part_pcs && valids && order, "assembly_order: null pointer");
  MPA_REQUIRE(B * P * (int64_t)mpa::leaf_npad(N) < (1LL << 31), "assembly_order: problem too large");
  mpa::launch_leaf_order(part_pcs, valids, B, P, N, order, mpa::as_stream(stream));
  return mpa::check_launch("assembly_order");
}

namespace {
struct Workspace {
  float *R1, *R2, *S1, *S2, *partial, *part_tiles, *shape_tiles, *grid_f;
  int *ip1, *ip2, *is1, *is2, *grid_i;
  int tiles;
  // leaf search (leaf_nn.hip): the library's own k-d order (callers may hand one in), records / leaf boxes / part boxes
  // of the four clouds R1, R2, S1, S2
  float *order, *rec[4], *leaf[4], *pbox[4], *wsum_part, *wsum_shape, *scratch;
};

// Which searches answer the two Chamfer terms.  0: brute-force scans; 1 (default): brute-force per-part scan +
// grid-pruned whole-shape search (grid_nn.hip) — the path of rounds 1-4; 2: leaf search for both (leaf_nn.hip); 3 "auto":
// leaf search for the per-part term, and for the whole-shape term of every SAMPLE the search its geometry favours
// (leaf_route_kernel: grid where the parts fill the shape's box, leaves where they are many and small).  Identical
// results.  The caller's `search` argument (>= 0) decides, else MPA_SHAPE_SEARCH = brute | grid | leaf | auto, else 1:
// the leaf structure pays where parts are many and small (artifact-like data: configs that say so ask for "auto"), and
// costs a k-d ordering per batch where they are few and large.
int search_mode(int64_t P, int64_t N, int search) {
  int want = search;
  if (want < 0) {
    const char* e = getenv("MPA_SHAPE_SEARCH");
    want = !e ? 1 : (e[0] == 'b' ? 0 : (e[0] == 'l' ? 2 : (e[0] == 'a' ? 3 : 1)));
  }
  if (want >= 2 && !mpa::leaf_supported(P, N)) want = 1;
  if (want == 1 && P > 64) want = 0;  // (the grid keeps one padded part per lane)
  return want;
}

// Queries per lane of the NN scans: 4 (fewer, fatter blocks) when there is enough work to fill the chip,
// else 2.  MPA_ASSEMBLY_Q=2|4 overrides (tuning only).
int pick_q(int64_t B, int64_t P, int64_t N) {
  if (const char* e = getenv("MPA_ASSEMBLY_Q")) {
    if (e[0] == '2') return 2;
    if (e[0] == '4') return 4;
  }
  return 2;
}

// the per-part Chamfer of the brute / grid modes: matrix-core gated search (default) or the exhaustive scan
bool part_search_gate(int64_t N) {
  const char* e = getenv("MPA_PART_SEARCH");
  if (e != nullptr && e[0] == 's') return false;
  return mpa::gate_supported(N, N) && (N >= 64 || (e != nullptr && e[0] == 'g'));
}

Workspace carve(float* fws, int32_t* iws, int64_t B, int64_t P, int64_t N, int q) {
  Workspace w;
  const int64_t cloud = B * P * N * 3, pn = B * P * N;
  const int64_t tile = 64 * (int64_t)q;
  w.tiles = (int)((N + tile - 1) / tile);
  w.R1 = fws;
  w.R2 = fws + cloud;
  w.S1 = fws + 2 * cloud;
  w.S2 = fws + 3 * cloud;
  w.partial = fws + 4 * cloud;
  w.part_tiles = w.partial + 5 * B * P;
  w.shape_tiles = w.part_tiles + 2 * B * P * w.tiles;  // (capacity was sized for the smallest tile)
  w.ip1 = iws;
  w.ip2 = iws + pn;
  w.is1 = iws + 2 * pn;
  w.is2 = iws + 3 * pn;
  w.grid_i = iws + 4 * pn;
  const int64_t min_tiles = (N + kMinTile - 1) / kMinTile;
  w.grid_f = fws + (4 * cloud + 5 * B * P + 4 * B * P * min_tiles + 3) / 4 * 4;
  float* lf = w.grid_f + mpa::grid_workspace_floats(B, P, N);
  const int64_t npad = mpa::leaf_supported(P, N) ? mpa::leaf_npad(N) : 0;
  w.order = lf;
  lf += 4 * B * P * npad;
  for (int c = 0; c < 4; ++c, lf += 4 * B * P * npad) w.rec[c] = lf;
  for (int c = 0; c < 4; ++c, lf += 8 * B * P * (npad / 32)) w.leaf[c] = lf;
  for (int c = 0; c < 4; ++c, lf += 8 * B * P) w.pbox[c] = lf;
  const int64_t nw = npad >= 64 ? npad / 64 : 1;
  w.wsum_part = lf;
  w.wsum_shape = lf + 2 * B * P * nw;
  w.scratch = lf + (4 * B * P * nw + 3) / 4 * 4;
  return w;
}
}  // namespace

extern "C" int mpa_assembly_loss_forward_ordered(const float* part_pcs, const float* valids,
                                                 const float* quat_pred, const float* trans_pred,
                                                 const float* quat_gt, const float* trans_gt, int64_t B,
                                                 int64_t P, int64_t N, int training, int fill_pad_points,
                                                 const float* order, int search, float* float_ws, int32_t* int_ws,
                                                 float* losses, void* const* events, void* stream);

extern "C" int mpa_assembly_loss_forward(const float* part_pcs, const float* valids,
                                         const float* quat_pred, const float* trans_pred,
                                         const float* quat_gt, const float* trans_gt, int64_t B,
                                         int64_t P, int64_t N, int training, int fill_pad_points,
                                         float* float_ws, int32_t* int_ws, float* losses,
                                         void* stream) {
  return mpa_assembly_loss_forward_ordered(part_pcs, valids, quat_pred, trans_pred, quat_gt, trans_gt, B, P, N,
                                           training, fill_pad_points, nullptr, -1, float_ws, int_ws, losses, nullptr,
                                           stream);
}

extern "C" int mpa_assembly_loss_forward_timed(const float* part_pcs, const float* valids,
                                               const float* quat_pred, const float* trans_pred,
                                               const float* quat_gt, const float* trans_gt, int64_t B,
                                               int64_t P, int64_t N, int training, int fill_pad_points,
                                               float* float_ws, int32_t* int_ws, float* losses,
                                               void* const* events, void* stream) {
  return mpa_assembly_loss_forward_ordered(part_pcs, valids, quat_pred, trans_pred, quat_gt, trans_gt, B, P, N,
                                           training, fill_pad_points, nullptr, -1, float_ws, int_ws, losses, events,
                                           stream);
}

// Same launches; when `events` is non-null it holds 7 hipEvent_t recorded on `stream`: [0] start, [1] after
// the pose kernel (and, when `order` is null, the k-d ordering in front of it), [2] after the per-part Chamfer,
// [3] after the whole-shape Chamfer phase, [4] after the finalize kernel, [5]/[6] immediately before/after the
// whole-shape search kernel itself (leaf or grid search; left untouched by the brute-force path).  bench.py times
// the dominant kernel with them.
// `search`: -1 = MPA_SHAPE_SEARCH or the default (see search_mode), 0 brute, 1 grid, 2 leaf, 3 auto.
// `order` (nullable): the k-d order of this batch's parts from mpa_assembly_order — a function of part_pcs and valids
// only, so one ordering serves every loss evaluation of a step (GNN iterations, min-of-N samples); null: computed here.
extern "C" int mpa_assembly_loss_forward_ordered(const float* part_pcs, const float* valids,
                                                 const float* quat_pred, const float* trans_pred,
                                                 const float* quat_gt, const float* trans_gt, int64_t B,
                                                 int64_t P, int64_t N, int training, int fill_pad_points,
                                                 const float* order, int search, float* float_ws, int32_t* int_ws,
                                                 float* losses, void* const* events, void* stream) {
  auto mark = [&](int k) {
    if (events != nullptr && events[k] != nullptr)  // (entries may be null: time only some of the phases)
      (void)hipEventRecord(reinterpret_cast<hipEvent_t>(events[k]), mpa::as_stream(stream));
  };
  MPA_REQUIRE(B >= 0 && P >= 0 && N >= 0, "assembly_loss_forward: negative size");
  if (B == 0) return MPA_OK;
  MPA_REQUIRE(P >= 1 && P <= 64 && N >= 1, "assembly_loss_forward: need 1 <= P <= 64 and N >= 1");
  MPA_REQUIRE(part_pcs && valids && quat_pred && trans_pred && quat_gt && trans_gt && float_ws &&
                  int_ws && losses, "assembly_loss_forward: null pointer");
  MPA_REQUIRE(P * N < (1LL << 30) && B * P * ((N + kMinTile - 1) / kMinTile) < (1LL << 31),
              "assembly_loss_forward: problem too large");
  hipStream_t s = mpa::as_stream(stream);
  const int q = pick_q(B, P, N);
  const Workspace w = carve(float_ws, int_ws, B, P, N, q);
  const unsigned parts = (unsigned)(B * P);
  const int mode = search_mode(P, N, search);
  // padded parts never write their tile sums: clear them (2 directions x B*P*tiles, both arrays)
  mpa::zero_words_async(w.part_tiles, 4 * B * P * w.tiles, s);
  mark(0);
  if (mode >= 2) {
    // ---- leaf search: k-d order (once per batch), pose kernel in that order, both searches over the leaves ----
    int* route = mpa::leaf_route(w.scratch);
    const int npad = mpa::leaf_npad(N);
    MPA_REQUIRE(B * P * (int64_t)npad < (1LL << 31), "assembly_loss_forward: problem too large");
    if (order == nullptr) {
      mpa::launch_leaf_order(part_pcs, valids, B, P, N, w.order, s);
      order = w.order;
    }
    LeafOut lo;
    for (int c = 0; c < 4; ++c) {
      lo.rec[c] = reinterpret_cast<float4*>(w.rec[c]);
      lo.leaf[c] = w.leaf[c];
      lo.part[c] = w.pbox[c];
    }
    hipLaunchKernelGGL(assembly_pose_leaf_kernel, dim3(parts), dim3(kThreads), 0, s,
                       reinterpret_cast<const float4*>(order), valids, quat_pred, trans_pred, quat_gt, trans_gt, (int)P,
                       (int)N, npad, fill_pad_points, w.R1, w.R2, w.S1, w.S2, w.partial, lo,
                       mpa::leaf_heavy_counters(w.scratch), mode == 3 ? mpa::grid_bbox(w.grid_f, B, P, N) : (float*)nullptr,
                       mode == 3 ? mpa::grid_ticket(w.grid_i, B) : (unsigned*)nullptr);
    mpa::launch_leaf_route(valids, w.pbox[3], B, P, mode == 3 ? -1 : 0, route, s);
    mark(1);
    const mpa::LeafCloud r1{w.rec[0], w.leaf[0], w.pbox[0], w.R1}, r2{w.rec[1], w.leaf[1], w.pbox[1], w.R2};
    const mpa::LeafCloud s1{w.rec[2], w.leaf[2], w.pbox[2], w.S1}, s2{w.rec[3], w.leaf[3], w.pbox[3], w.S2};
    // per-part Chamfer: the matrix-core gated search on the clouds in original order (0.10 vs 0.16 ms on the artifact mix);
    // MPA_PART_SEARCH=scan leaves it to the leaf search as in the first half of round 5
    const bool part_gate = part_search_gate(N);
    if (part_gate)
      mpa::launch_gate_part_search(valids, w.R1, w.R2, B, P, N, w.ip1, w.ip2, w.part_tiles, s);
    else
      mpa::launch_leaf_search(false, valids, r1, r2, B, P, N, w.ip1, w.ip2, w.wsum_part, w.scratch, s);
    mark(2);
    // [5] .. [6] bracket the search kernels proper of BOTH routes (grid search, leaf search + its second pass) — not the
    // grid's build and its distance sums, as in rounds 1-4.  A route no sample takes costs its launches a header each.
    if (mode == 3)
      mpa::launch_grid_shape_search(valids, w.S1, w.S2, B, P, N, w.tiles, w.grid_f, w.grid_i, w.is1, w.is2,
                                    w.shape_tiles, nullptr, nullptr, s, route, 1);
    mark(5);
    if (mode == 3)
      mpa::launch_grid_shape_search(valids, w.S1, w.S2, B, P, N, w.tiles, w.grid_f, w.grid_i, w.is1, w.is2,
                                    w.shape_tiles, nullptr, nullptr, s, route, 2);
    mpa::launch_leaf_search(true, valids, s1, s2, B, P, N, w.is1, w.is2, w.wsum_shape, w.scratch, s, route);
    mark(6);
    if (mode == 3)
      mpa::launch_grid_shape_search(valids, w.S1, w.S2, B, P, N, w.tiles, w.grid_f, w.grid_i, w.is1, w.is2,
                                    w.shape_tiles, nullptr, nullptr, s, route, 4);
    mark(3);
    const int nw = npad >= 64 ? npad / 64 : 1;  // every wave of a valid part leaves its distance sum
    hipLaunchKernelGGL(assembly_finalize_kernel, dim3((unsigned)B), dim3(64), 0, s, valids, quat_pred,
                       trans_pred, quat_gt, trans_gt, w.partial, part_gate ? (const float*)w.part_tiles : (const float*)w.wsum_part,
                       (const float*)w.wsum_shape, (int)B, (int)P, (int)N, part_gate ? mpa::gate_tiles(N, N) : nw, nw, training, losses,
                       (const float*)w.shape_tiles, w.tiles, (const int*)route);
    mark(4);
    return mpa::check_launch("assembly_loss_forward");
  }
  hipLaunchKernelGGL(assembly_pose_kernel, dim3(parts), dim3(kThreads), 0, s, part_pcs, valids,
                     quat_pred, trans_pred, quat_gt, trans_gt, (int)N, fill_pad_points, w.R1, w.R2,
                     w.S1, w.S2, w.partial, mpa::grid_bbox(w.grid_f, B, P, N), mpa::grid_ticket(w.grid_i, B));
  mark(1);
  const dim3 grid(parts * w.tiles, 2, 1);
  // Steering a sample's blocks to one XCD (L2 affinity) loses more to the static load imbalance between
  // XCDs than it gains (measured 2.56 vs 2.21 ms at B=32, P=20, N=1000): off unless MPA_XCD_REMAP=1.
  const char* re = getenv("MPA_XCD_REMAP");
  const int remap = re ? (re[0] != '0') : 0;
  // per-part Chamfer: the matrix-core gated search (gate_nn.hip) unless MPA_PART_SEARCH=scan asks for the exhaustive scan
  // of rounds 1-4 (identical results; the knob exists for A/B timing and the cross-check tests)
  const bool part_gate = part_search_gate(N);
  const int tiles_part = part_gate ? mpa::gate_tiles(N, N) : w.tiles;
  if (part_gate)
    mpa::launch_gate_part_search(valids, w.R1, w.R2, B, P, N, w.ip1, w.ip2, w.part_tiles, s);
  else if (q == 4)
    hipLaunchKernelGGL((assembly_nn_kernel<4, mpa::kChunkMin, false>), grid, dim3(kThreads), 0, s, valids,
                       w.R1, w.R2, (int)B, (int)P, (int)N, w.tiles, remap, w.ip1, w.ip2, w.part_tiles);
  else
    hipLaunchKernelGGL((assembly_nn_kernel<2, mpa::kChunkMin, false>), grid, dim3(kThreads), 0, s, valids,
                       w.R1, w.R2, (int)B, (int)P, (int)N, w.tiles, remap, w.ip1, w.ip2, w.part_tiles);
  mark(2);
  if (mode == 1)
    mpa::launch_grid_shape_search(valids, w.S1, w.S2, B, P, N, w.tiles, w.grid_f, w.grid_i, w.is1, w.is2,
                                  w.shape_tiles, events ? reinterpret_cast<hipEvent_t>(events[5]) : nullptr,
                                  events ? reinterpret_cast<hipEvent_t>(events[6]) : nullptr, s);
  else if (q == 4)
    hipLaunchKernelGGL((assembly_nn_kernel<4, mpa::kChunkMin, true>), grid, dim3(kThreads), 0, s, valids,
                       w.S1, w.S2, (int)B, (int)P, (int)N, w.tiles, remap, w.is1, w.is2, w.shape_tiles);
  else
    hipLaunchKernelGGL((assembly_nn_kernel<2, mpa::kChunkMin, true>), grid, dim3(kThreads), 0, s, valids,
                       w.S1, w.S2, (int)B, (int)P, (int)N, w.tiles, remap, w.is1, w.is2, w.shape_tiles);
  mark(3);
  hipLaunchKernelGGL(assembly_finalize_kernel, dim3((unsigned)B), dim3(64), 0, s, valids, quat_pred,
                     trans_pred, quat_gt, trans_gt, w.partial, w.part_tiles, w.shape_tiles, (int)B,
                     (int)P, (int)N, tiles_part, w.tiles, training, losses, (const float*)nullptr, 0, (const int*)nullptr);
  mark(4);
  return mpa::check_launch("assembly_loss_forward");
}

extern "C" int mpa_assembly_loss_backward(const float* grad_losses, const float* part_pcs,
                                          const float* valids, const float* quat_pred,
                                          const float* trans_pred, const float* quat_gt,
                                          const float* trans_gt, int64_t B, int64_t P, int64_t N,
                                          int training, const float* float_ws, const int32_t* int_ws,
                                          float* grad_quat, float* grad_trans, void* stream) {
  MPA_REQUIRE(B >= 0 && P >= 0 && N >= 0, "assembly_loss_backward: negative size");
  if (B == 0) return MPA_OK;
  MPA_REQUIRE(P >= 1 && P <= 64 && N >= 1, "assembly_loss_backward: need 1 <= P <= 64 and N >= 1");
  MPA_REQUIRE(grad_losses && part_pcs && valids && quat_pred && trans_pred && quat_gt && trans_gt &&
                  float_ws && int_ws && grad_quat && grad_trans, "assembly_loss_backward: null pointer");
  const Workspace w = carve(const_cast<float*>(float_ws), const_cast<int32_t*>(int_ws), B, P, N,
                            pick_q(B, P, N));
  // four slices of the point range per part when the (forward-only) tile-sum area is big enough for their partials
  const int S = w.tiles >= 8 ? 4 : 1;
  hipLaunchKernelGGL(assembly_backward_kernel, dim3((unsigned)(B * P), (unsigned)S), dim3(kThreads), 0,
                     mpa::as_stream(stream), grad_losses, part_pcs, valids, quat_pred, trans_pred,
                     quat_gt, trans_gt, w.R1, w.R2, w.S1, w.S2, w.ip1, w.ip2, w.is1, w.is2, (int)B,
                     (int)P, (int)N, training, grad_quat, grad_trans, w.part_tiles);
  if (S > 1)
    hipLaunchKernelGGL(assembly_backward_finish_kernel, dim3((unsigned)((B * P * 8 + 255) / 256)), dim3(256), 0,
                       mpa::as_stream(stream), grad_losses, valids, quat_pred, trans_pred, quat_gt, trans_gt,
                       (const float*)w.part_tiles, (int)B, (int)P, S, grad_quat, grad_trans);
  return mpa::check_launch("assembly_loss_backward");
}
