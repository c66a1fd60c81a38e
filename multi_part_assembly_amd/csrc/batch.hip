// Batch producer, device side: the per-part transform of GeometryPartDataset.__getitem__
// (multi_part_assembly/datasets/geometry_data.py:74-107,133-146) for a whole batch in one launch.
//
// Per valid part the reference does, in float64 on a CPU worker: centroid = mean of the sampled points, points -=
// centroid, points = (rot_mat @ points^T)^T, points = points[order] (a random permutation), then pads to
// [max_num_part, N, 3] and casts to float32; part_trans = centroid.  Here one block per part slot does the same
// in float64 — fixed-order sums, no FMA (this file is built with -ffp-contract=off) — from the raw sampled
// points, the rotation matrices and the permutations the host drew (same RNG calls as the reference, see
// datasets.py), and writes the float32 batch tensors directly; padded slots are written as zeros.
// HBM-bound: 24 B read + 4 B index + 12 B written per point.
#include "common.h"

namespace {

constexpr int kThreads = 256;

// grid = M part slots, block 256.
__global__ __launch_bounds__(kThreads) void part_batch_transform_kernel(
    const double* __restrict__ raw, const double* __restrict__ rot, const int* __restrict__ perm,
    const float* __restrict__ valids, int N, float* __restrict__ part_pcs, float* __restrict__ part_trans) {
  __shared__ double red[kThreads][3];
  __shared__ double cen[3];
  const int m = blockIdx.x, t = threadIdx.x;
  float* out = part_pcs + 3LL * m * N;
  if (valids[m] == 0.0f) {  // padded slot: zeros (geometry_data.py:102-107)
    for (int i = t; i < 3 * N; i += kThreads) out[i] = 0.0f;
    if (t < 3) part_trans[3 * m + t] = 0.0f;
    return;
  }
  const double* src = raw + 3LL * m * N;
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int i = t; i < N; i += kThreads) {
    sx += src[3 * i + 0];
    sy += src[3 * i + 1];
    sz += src[3 * i + 2];
  }
  red[t][0] = sx;
  red[t][1] = sy;
  red[t][2] = sz;
  __syncthreads();
  for (int half = kThreads / 2; half >= 1; half >>= 1) {  // fixed-order tree
    if (t < half) {
      red[t][0] += red[t + half][0];
      red[t][1] += red[t + half][1];
      red[t][2] += red[t + half][2];
    }
    __syncthreads();
  }
  if (t < 3) {
    const double c = red[0][t] / (double)N;
    cen[t] = c;
    part_trans[3 * m + t] = (float)c;
  }
  __syncthreads();
  const double cx = cen[0], cy = cen[1], cz = cen[2];
  double r[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) r[k] = rot[9LL * m + k];
  const int* ord = perm + (long long)m * N;
  for (int i = t; i < N; i += kThreads) {
    const int s = ord[i];
    const double px = src[3 * s + 0] - cx, py = src[3 * s + 1] - cy, pz = src[3 * s + 2] - cz;
    out[3 * i + 0] = (float)((r[0] * px + r[1] * py) + r[2] * pz);
    out[3 * i + 1] = (float)((r[3] * px + r[4] * py) + r[5] * pz);
    out[3 * i + 2] = (float)((r[6] * px + r[7] * py) + r[8] * pz);
  }
}

}  // namespace

extern "C" int mpa_part_batch_transform(const double* raw, const double* rot, const int32_t* perm,
                                        const float* valids, int64_t M, int64_t N, float* part_pcs,
                                        float* part_trans, void* stream) {
  MPA_REQUIRE(M >= 0 && N >= 1 && N <= (1LL << 24), "part_batch_transform: bad sizes");
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(raw && rot && perm && valids && part_pcs && part_trans, "part_batch_transform: null pointer");
  hipLaunchKernelGGL(part_batch_transform_kernel, dim3((unsigned)M), dim3(kThreads), 0, mpa::as_stream(stream), raw,
                     rot, perm, valids, (int)N, part_pcs, part_trans);
  return mpa::check_launch("part_batch_transform");
}
