// Device-side GT <-> prediction matching of geometrically equivalent parts (semantic datasets).
//
// Replaces BaseModel._match_parts / _linear_sum_assignment (multi_part_assembly/models/modules/base_model.py:150-238):
// per sample and per group of equivalent parts (match_ids == g), the reference sub-samples n = 100 points, applies
// the predicted pose of part i and the GT pose of part j, builds the p x p matrix of Chamfer costs with one
// chamfer_distance call, copies it to the host, runs scipy.optimize.linear_sum_assignment and permutes the GT
// poses — B * groups * sample_iter host round trips per training step.  Here three launches handle every group of
// the batch with no host involvement:
//   match_cost_kernel   one block per (sample, group, i, j): both sub-sampled clouds in LDS, brute-force Chamfer in
//                       the library's pinned arithmetic, cost = sum1/n + sum2/n;
//   lsap_kernel         one block per problem: the shortest-augmenting-path algorithm of scipy's
//                       rectangular_lsap.cpp (scipy 1.15.3; Crouse 2016), restated step by step in float64 so that
//                       ties break the same way — square problems of at most 64 rows, run by one lane out of LDS
//                       (p <= 20 here: a few thousand steps);
//   match_apply_kernel  new_gt[slot] = gt[member[col4row[rank(slot)]]] for grouped slots, identity elsewhere.
#include "common.h"
#include "quat.h"

namespace {

constexpr int kMaxP = 64;      // parts per sample
constexpr int kMaxN = 128;     // sub-sampled points per part
constexpr int kCostT = 128;

// members of group `g` (match_ids == g + 1) of sample b, ascending; returns their number
__device__ __forceinline__ int group_members(const int* __restrict__ ids, int P, int g, int* __restrict__ out) {
  int c = 0;
  for (int p = 0; p < P; ++p)
    if (ids[p] == g + 1) out[c++] = p;
  return c;
}

// grid = (B*G, P*P) blocks of 128 threads; cost [B*G][P][P]
__global__ __launch_bounds__(kCostT) void match_cost_kernel(
    const float* __restrict__ pcs, const float* __restrict__ t1, const float* __restrict__ q1,
    const float* __restrict__ t2, const float* __restrict__ q2, const int* __restrict__ match_ids,
    const int* __restrict__ sample_idx, int P, int N, int G, int n, float* __restrict__ cost) {
  __shared__ int mem[kMaxP];
  __shared__ int count;
  __shared__ float A[kMaxN][3], Bc[kMaxN][3];
  __shared__ float red[2][kCostT];
  const int bg = blockIdx.x, b = bg / G, g = bg % G, i = blockIdx.y / P, j = blockIdx.y % P, t = threadIdx.x;
  if (t == 0) count = group_members(match_ids + (long long)b * P, P, g, mem);
  __syncthreads();
  if (i >= count || j >= count) return;  // block-uniform
  const int pi = mem[i], pj = mem[j];
  if (t < n) {
    const int s = sample_idx[(long long)bg * n + t];
    const long long mi = (long long)b * P + pi, mj = (long long)b * P + pj;
    const float* a = pcs + (mi * N + s) * 3;
    const float* c = pcs + (mj * N + s) * 3;
    float x, y, z;
    mpa::quat_rotate(mpa::Quat{q1[4 * mi], q1[4 * mi + 1], q1[4 * mi + 2], q1[4 * mi + 3]}, a[0], a[1], a[2], x, y, z);
    A[t][0] = x + t1[3 * mi];
    A[t][1] = y + t1[3 * mi + 1];
    A[t][2] = z + t1[3 * mi + 2];
    mpa::quat_rotate(mpa::Quat{q2[4 * mj], q2[4 * mj + 1], q2[4 * mj + 2], q2[4 * mj + 3]}, c[0], c[1], c[2], x, y, z);
    Bc[t][0] = x + t2[3 * mj];
    Bc[t][1] = y + t2[3 * mj + 1];
    Bc[t][2] = z + t2[3 * mj + 2];
  }
  __syncthreads();
  float d1 = 0.0f, d2 = 0.0f;
  if (t < n) {
    float b1 = 1e32f, b2 = 1e32f;
    const float ax = A[t][0], ay = A[t][1], az = A[t][2], bx = Bc[t][0], by = Bc[t][1], bz = Bc[t][2];
    for (int k = 0; k < n; ++k) {
      float dx = ax - Bc[k][0], dy = ay - Bc[k][1], dz = az - Bc[k][2];
      const float da = (dx * dx + dy * dy) + dz * dz;
      b1 = da < b1 ? da : b1;
      dx = bx - A[k][0];
      dy = by - A[k][1];
      dz = bz - A[k][2];
      const float db = (dx * dx + dy * dy) + dz * dz;
      b2 = db < b2 ? db : b2;
    }
    d1 = b1;
    d2 = b2;
  }
  red[0][t] = d1;
  red[1][t] = d2;
  __syncthreads();
  for (int half = kCostT / 2; half >= 1; half >>= 1) {  // fixed-order tree
    if (t < half) {
      red[0][t] += red[0][t + half];
      red[1][t] += red[1][t + half];
    }
    __syncthreads();
  }
  if (t == 0) cost[((long long)bg * P + i) * P + j] = red[0][0] / (float)n + red[1][0] / (float)n;
}

// scipy rectangular_lsap.cpp (square case), one problem; all arrays in LDS, run by one lane.
struct Lsap {
  double u[kMaxP], v[kMaxP], spc[kMaxP];
  int path[kMaxP], col4row[kMaxP], row4col[kMaxP], remaining[kMaxP];
  bool SR[kMaxP], SC[kMaxP];
};

__device__ void lsap_solve(const float* __restrict__ cost, int ld, int nc, Lsap& w) {
  const double inf = __builtin_inf();
  for (int k = 0; k < nc; ++k) {
    w.u[k] = w.v[k] = 0.0;
    w.col4row[k] = w.row4col[k] = -1;
    w.path[k] = -1;
  }
  for (int cur = 0; cur < nc; ++cur) {
    // ---- augmenting path from row `cur`
    double min_val = 0.0;
    int num_remaining = nc;
    for (int it = 0; it < nc; ++it) {
      w.remaining[it] = nc - it - 1;  // filled in reverse, as scipy does
      w.SR[it] = w.SC[it] = false;
      w.spc[it] = inf;
    }
    int sink = -1, i = cur;
    while (sink == -1) {
      int index = -1;
      double lowest = inf;
      w.SR[i] = true;
      for (int it = 0; it < num_remaining; ++it) {
        const int j = w.remaining[it];
        const double r = min_val + (double)cost[(long long)i * ld + j] - w.u[i] - w.v[j];
        if (r < w.spc[j]) {
          w.path[j] = i;
          w.spc[j] = r;
        }
        // among equal costs prefer a column that ends the path (a new sink)
        if (w.spc[j] < lowest || (w.spc[j] == lowest && w.row4col[j] == -1)) {
          lowest = w.spc[j];
          index = it;
        }
      }
      min_val = lowest;
      if (min_val == inf) return;  // infeasible (NaN/inf costs): leave the remaining rows unassigned
      const int j = w.remaining[index];
      if (w.row4col[j] == -1) sink = j;
      else i = w.row4col[j];
      w.SC[j] = true;
      w.remaining[index] = w.remaining[--num_remaining];
    }
    // ---- dual update
    w.u[cur] += min_val;
    for (int r = 0; r < nc; ++r)
      if (w.SR[r] && r != cur) w.u[r] += min_val - w.spc[w.col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (w.SC[j]) w.v[j] -= min_val - w.spc[j];
    // ---- augment
    int j = sink;
    while (true) {
      const int r = w.path[j];
      w.row4col[j] = r;
      const int prev = w.col4row[r];
      w.col4row[r] = j;
      j = prev;
      if (r == cur) break;
    }
  }
}

// grid = problems, block 64.  cost [problems][ld][ld]; sizes[problem] (nullable: derived from match_ids) rows used;
// col4row [problems][ld] (-1 past the size).
__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ cost, const int* __restrict__ sizes,
                                                  const int* __restrict__ match_ids, int P, int G, int ld,
                                                  int* __restrict__ col4row) {
  __shared__ Lsap w;
  __shared__ int mem[kMaxP];
  const int pr = blockIdx.x;
  if (threadIdx.x == 0) {
    int nc = sizes != nullptr ? sizes[pr] : group_members(match_ids + (long long)(pr / G) * P, P, pr % G, mem);
    if (nc > ld) nc = ld;
    if (nc > 0) lsap_solve(cost + (long long)pr * ld * ld, ld, nc, w);
    for (int k = 0; k < ld; ++k) col4row[(long long)pr * ld + k] = k < nc ? w.col4row[k] : -1;
  }
}

// grid = B blocks of 64 threads (thread = slot).  Rows of a group are its members in ascending slot order (the
// reference indexes with the sorted member list), so member rank r receives the GT pose of member col4row[r].
__global__ __launch_bounds__(64) void match_apply_kernel(const float* __restrict__ gt_t, const float* __restrict__ gt_q,
                                                         const int* __restrict__ match_ids,
                                                         const int* __restrict__ col4row, int P, int G,
                                                         float* __restrict__ new_t, float* __restrict__ new_q,
                                                         int* __restrict__ perm) {
  __shared__ int ids[kMaxP];
  const int b = blockIdx.x, p = threadIdx.x;
  if (p < P) ids[p] = match_ids[(long long)b * P + p];
  __syncthreads();
  if (p >= P) return;
  int src = p;
  const int g = ids[p] - 1;
  if (g >= 0 && g < G) {
    int rank = 0;
    for (int k = 0; k < p; ++k) rank += ids[k] == g + 1 ? 1 : 0;
    const int c = col4row[((long long)b * G + g) * P + rank];
    if (c >= 0) {  // member with rank c
      int seen = 0;
      for (int k = 0; k < P; ++k) {
        if (ids[k] == g + 1) {
          if (seen == c) src = k;
          ++seen;
        }
      }
    }
  }
  const long long o = (long long)b * P + p, s = (long long)b * P + src;
  perm[o] = src;
#pragma unroll
  for (int k = 0; k < 3; ++k) new_t[3 * o + k] = gt_t[3 * s + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) new_q[4 * o + k] = gt_q[4 * s + k];
}

}  // namespace

extern "C" int mpa_linear_sum_assignment(const float* cost, const int32_t* sizes, int64_t problems, int64_t ld,
                                         int32_t* col4row, void* stream) {
  MPA_REQUIRE(problems >= 0 && ld >= 1 && ld <= kMaxP, "linear_sum_assignment: need 1 <= ld <= 64");
  if (problems == 0) return MPA_OK;
  MPA_REQUIRE(cost && sizes && col4row, "linear_sum_assignment: null pointer");
  hipLaunchKernelGGL(lsap_kernel, dim3((unsigned)problems), dim3(64), 0, mpa::as_stream(stream), cost, sizes,
                     (const int*)nullptr, 0, 1, (int)ld, col4row);
  return mpa::check_launch("linear_sum_assignment");
}

extern "C" int mpa_match_parts(const float* part_pcs, const float* pred_trans, const float* pred_quat,
                               const float* gt_trans, const float* gt_quat, const int32_t* match_ids,
                               const int32_t* sample_idx, int64_t B, int64_t P, int64_t N, int64_t G, int64_t n,
                               float* cost_ws, int32_t* col4row_ws, float* new_trans, float* new_quat,
                               int32_t* perm, void* stream) {
  MPA_REQUIRE(B >= 0 && P >= 1 && P <= kMaxP && N >= 1 && G >= 1 && G <= P && n >= 1 && n <= kMaxN && n <= N,
              "match_parts: need 1 <= P <= 64, 1 <= G <= P, 1 <= n <= min(N, 128)");
  if (B == 0) return MPA_OK;
  MPA_REQUIRE(part_pcs && pred_trans && pred_quat && gt_trans && gt_quat && match_ids && sample_idx && cost_ws &&
                  col4row_ws && new_trans && new_quat && perm, "match_parts: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  hipLaunchKernelGGL(match_cost_kernel, dim3((unsigned)(B * G), (unsigned)(P * P)), dim3(kCostT), 0, s, part_pcs,
                     pred_trans, pred_quat, gt_trans, gt_quat, match_ids, sample_idx, (int)P, (int)N, (int)G, (int)n,
                     cost_ws);
  hipLaunchKernelGGL(lsap_kernel, dim3((unsigned)(B * G)), dim3(64), 0, s, cost_ws, (const int*)nullptr, match_ids,
                     (int)P, (int)G, (int)P, col4row_ws);
  hipLaunchKernelGGL(match_apply_kernel, dim3((unsigned)B), dim3(64), 0, s, gt_trans, gt_quat, match_ids, col4row_ws,
                     (int)P, (int)G, new_trans, new_quat, perm);
  return mpa::check_launch("match_parts");
}
