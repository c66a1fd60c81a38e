// Chamfer nearest-neighbour forward / backward for gfx950 (MI355X).
//
// Replaces ChamferForwardKernel / ChamferBackwardKernel of the reference
// (multi_part_assembly/utils/chamfer/cuda/chamfer_kernel.cu:32-95, :175-210).  Not a translation:
//
//  * the reference stages 512-point target tiles through shared memory with two block barriers per
//    tile; here the target cloud is walked with WAVE-UNIFORM addresses, so the compiler fetches it
//    through the scalar cache (s_load_dwordx8/x16) into SGPRs and the VALU consumes the coordinates
//    as scalar operands — no LDS traffic, no barriers, no VGPRs spent on targets;
//  * each lane owns Q query points, so one scalar fetch of a target feeds 64*Q distance evaluations;
//  * both directions of the bidirectional search run in ONE launch (blockIdx.y = direction);
//  * the search is exact brute force, and the arithmetic is pinned (see include/mpa_hip.h):
//        d = (dx*dx + dy*dy) + dz*dz   — every op rounded, this file is built with -ffp-contract=off
//    The FILTER variant evaluates the cheaper fused form  f = fma(dz,dz, fma(dy,dy, dx*dx))  on the
//    fast path and falls back to the pinned form only for candidates within 2^-20 (relative) of
//    the running minimum of f.  Both forms round the same positive three-term sum, so they differ
//    by < 8 ulp; any candidate that could win under the pinned form therefore passes the filter and
//    is then compared exactly, in index order, with strict `<` — results are bit-identical to the
//    direct variant (tests/test_chamfer_gpu.py checks that, and both against oracle/).
#include "common.h"

namespace {

constexpr int kThreads = 256;

// Pinned distance (no contraction; see header comment).  S = float (the training path) or double
// (the reference dispatches both, chamfer_kernel.cu:145; its gradcheck runs in double).
template <typename S>
__device__ __forceinline__ S dist_exact(S dx, S dy, S dz) {
  return (dx * dx + dy * dy) + dz * dz;
}

__device__ __forceinline__ float dist_fused(float dx, float dy, float dz) {
  return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ double dist_fused(double dx, double dy, double dz) {
  return __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
}
__device__ __forceinline__ float min_nonan(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double min_nonan(double a, double b) { return __builtin_fmin(a, b); }

// One direction of the search for one block: queries `qa` [na,3] of this batch element against
// targets `tb` [nb,3]; writes dist/idx [na].  T = targets per unrolled chunk.
template <typename S, int Q, bool FILTER>
__device__ __forceinline__ void nn_search(const S* __restrict__ qa, const S* __restrict__ tb,
                                          int na, int nb, int qbase, S* __restrict__ dist,
                                          long long* __restrict__ idx) {
  constexpr int T = 8;
  const S kInitDist = (S)1e32;  // chamfer_kernel.cu:60 (`scalar_t min_dist = 1e32`)
  S x[Q], y[Q], z[Q], best[Q];
  int bidx[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = qbase + q * kThreads + (int)threadIdx.x;
    const int ic = i < na ? i : na - 1;  // clamp: idle lanes shadow the last query, never stored
    x[q] = qa[3 * (long long)ic + 0];
    y[q] = qa[3 * (long long)ic + 1];
    z[q] = qa[3 * (long long)ic + 2];
    best[q] = kInitDist;
    bidx[q] = -1;
  }

  const int nb_main = nb - nb % T;
  if constexpr (!FILTER) {
    for (int j0 = 0; j0 < nb_main; j0 += T) {
      S tx[T], ty[T], tz[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {  // wave-uniform addresses -> scalar loads
        tx[t] = tb[3 * (j0 + t) + 0];
        ty[t] = tb[3 * (j0 + t) + 1];
        tz[t] = tb[3 * (j0 + t) + 2];
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const S d = dist_exact<S>(x[q] - tx[t], y[q] - ty[t], z[q] - tz[t]);
          const bool lt = d < best[q];
          best[q] = lt ? d : best[q];
          bidx[q] = lt ? j0 + t : bidx[q];
        }
      }
    }
  } else {
    // thr = (1 + 2^-20) * (running min of the fused form); +inf until something was seen.
    S fmin_[Q], thr[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) fmin_[q] = thr[q] = (S)__builtin_inff();
    for (int j0 = 0; j0 < nb_main; j0 += T) {
      S tx[T], ty[T], tz[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        tx[t] = tb[3 * (j0 + t) + 0];
        ty[t] = tb[3 * (j0 + t) + 1];
        tz[t] = tb[3 * (j0 + t) + 2];
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        S f[T];
#pragma unroll
        for (int t = 0; t < T; ++t) f[t] = dist_fused(x[q] - tx[t], y[q] - ty[t], z[q] - tz[t]);
        // fminf drops NaNs, which is what we want: a NaN candidate can never win (`<` is false).
        const S m01 = min_nonan(f[0], f[1]), m23 = min_nonan(f[2], f[3]);
        const S m45 = min_nonan(f[4], f[5]), m67 = min_nonan(f[6], f[7]);
        const S cmin = min_nonan(min_nonan(m01, m23), min_nonan(m45, m67));
        if (cmin <= thr[q]) {  // rare: some candidate of this chunk may beat the current best
          const S gate = thr[q];
#pragma unroll
          for (int t = 0; t < T; ++t) {
            if (f[t] <= gate) {
              const S d = dist_exact<S>(x[q] - tx[t], y[q] - ty[t], z[q] - tz[t]);
              if (d < best[q]) {
                best[q] = d;
                bidx[q] = j0 + t;
              }
            }
          }
          fmin_[q] = min_nonan(fmin_[q], cmin);
          thr[q] = fmin_[q] * (S)1.00000095367431640625;  // 1 + 2^-20
        }
      }
    }
  }
  // tail (< T targets), pinned form
  for (int j = nb_main; j < nb; ++j) {
    const S sx = tb[3 * j + 0], sy = tb[3 * j + 1], sz = tb[3 * j + 2];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const S d = dist_exact<S>(x[q] - sx, y[q] - sy, z[q] - sz);
      if (d < best[q]) {
        best[q] = d;
        bidx[q] = j;
      }
    }
  }

#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = qbase + q * kThreads + (int)threadIdx.x;
    if (i < na) {
      dist[i] = best[q];
      idx[i] = (long long)bidx[q];
    }
  }
}

// grid.x = batch * blocks_per_cloud (blocks_per_cloud sized for the LARGER cloud), grid.y = 2.
template <typename S, int Q, bool FILTER>
__global__ __launch_bounds__(kThreads) void chamfer_nn_kernel(
    const S* __restrict__ xyz1, const S* __restrict__ xyz2, int n1, int n2,
    int blocks_per_cloud, S* __restrict__ dist1, long long* __restrict__ idx1,
    S* __restrict__ dist2, long long* __restrict__ idx2) {
  const int b = blockIdx.x / blocks_per_cloud;
  const int qbase = (blockIdx.x % blocks_per_cloud) * (kThreads * Q);
  const long long o1 = (long long)b * n1, o2 = (long long)b * n2;
  if (blockIdx.y == 0) {
    if (qbase >= n1) return;
    nn_search<S, Q, FILTER>(xyz1 + 3 * o1, xyz2 + 3 * o2, n1, n2, qbase, dist1 + o1, idx1 + o1);
  } else {
    if (qbase >= n2) return;
    nn_search<S, Q, FILTER>(xyz2 + 3 * o2, xyz1 + 3 * o1, n2, n1, qbase, dist2 + o2, idx2 + o2);
  }
}

// Backward.  One thread per query point of one direction (blockIdx.y); the gather half is a plain
// accumulate into the query's own gradient row, the scatter half an fp32 atomic into the matched
// point's row (contract: include/mpa_hip.h).
template <typename S>
__global__ __launch_bounds__(kThreads) void chamfer_grad_kernel(
    const S* __restrict__ g1, const S* __restrict__ g2, const S* __restrict__ xyz1,
    const S* __restrict__ xyz2, const long long* __restrict__ idx1,
    const long long* __restrict__ idx2, long long total1, long long total2, int n1, int n2,
    S* gxyz1, S* gxyz2) {
  const bool fwd = blockIdx.y == 0;
  const long long total = fwd ? total1 : total2;
  const int na = fwd ? n1 : n2, nb = fwd ? n2 : n1;
  const S* g = fwd ? g1 : g2;
  const S* pa = fwd ? xyz1 : xyz2;
  const S* pb = fwd ? xyz2 : xyz1;
  const long long* ix = fwd ? idx1 : idx2;
  S* ga = fwd ? gxyz1 : gxyz2;
  S* gb = fwd ? gxyz2 : gxyz1;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long long)gridDim.x * kThreads) {
    const long long j = ix[i];
    if (j < 0 || j >= nb) continue;
    const long long b = i / na;
    const long long t = (b * nb + j) * 3;
    const S s = g[i] * (S)2;
    const S gx = s * (pa[3 * i + 0] - pb[t + 0]);
    const S gy = s * (pa[3 * i + 1] - pb[t + 1]);
    const S gz = s * (pa[3 * i + 2] - pb[t + 2]);
    atomicAdd(ga + 3 * i + 0, gx);
    atomicAdd(ga + 3 * i + 1, gy);
    atomicAdd(ga + 3 * i + 2, gz);
    atomicAdd(gb + t + 0, -gx);
    atomicAdd(gb + t + 1, -gy);
    atomicAdd(gb + t + 2, -gz);
  }
}

template <typename S, int Q, bool FILTER>
void launch_nn(const S* xyz1, const S* xyz2, int64_t batch, int n1, int n2, S* dist1,
               int64_t* idx1, S* dist2, int64_t* idx2, hipStream_t s) {
  const int nmax = n1 > n2 ? n1 : n2;
  const int bpc = (nmax + kThreads * Q - 1) / (kThreads * Q);
  dim3 grid((unsigned)(batch * bpc), 2, 1);
  hipLaunchKernelGGL((chamfer_nn_kernel<S, Q, FILTER>), grid, dim3(kThreads), 0, s, xyz1, xyz2, n1,
                     n2, bpc, dist1, (long long*)idx1, dist2, (long long*)idx2);
}

// variant: 0 = direct pinned arithmetic, 1 = fused-form filter + pinned recheck (bit-identical).
template <typename S>
int chamfer_forward_impl(const S* xyz1, const S* xyz2, int64_t batch, int64_t n1, int64_t n2,
                         S* dist1, int64_t* idx1, S* dist2, int64_t* idx2, int variant,
                         void* stream) {
  MPA_REQUIRE(batch >= 0 && n1 >= 0 && n2 >= 0, "chamfer_forward: negative size");
  MPA_REQUIRE(n1 < (1 << 30) && n2 < (1 << 30), "chamfer_forward: cloud larger than 2^30 points");
  MPA_REQUIRE(variant == 0 || variant == 1, "chamfer_forward: unknown variant %d", variant);
  if (batch == 0 || (n1 == 0 && n2 == 0)) return MPA_OK;
  MPA_REQUIRE((n1 == 0 || (xyz1 && dist1 && idx1)) && (n2 == 0 || (xyz2 && dist2 && idx2)),
              "chamfer_forward: null pointer");
  const int64_t nmax = n1 > n2 ? n1 : n2;
  // Q (queries per lane): 4 for real clouds; 1 for the <=256-point clouds of part matching
  // (reference base_model.py:163-173 sub-samples to 100 points) and for double (register budget).
  const bool q1 = nmax <= kThreads || sizeof(S) == 8;
  const int64_t bpc = (nmax + kThreads * (q1 ? 1 : 4) - 1) / (kThreads * (q1 ? 1 : 4));
  MPA_REQUIRE(batch * bpc < (int64_t)1 << 31, "chamfer_forward: grid too large");
  hipStream_t s = mpa::as_stream(stream);
  const int a = (int)n1, b = (int)n2;
  if (q1) {
    if (variant) launch_nn<S, 1, true>(xyz1, xyz2, batch, a, b, dist1, idx1, dist2, idx2, s);
    else launch_nn<S, 1, false>(xyz1, xyz2, batch, a, b, dist1, idx1, dist2, idx2, s);
  } else {
    if (variant) launch_nn<S, 4, true>(xyz1, xyz2, batch, a, b, dist1, idx1, dist2, idx2, s);
    else launch_nn<S, 4, false>(xyz1, xyz2, batch, a, b, dist1, idx1, dist2, idx2, s);
  }
  return mpa::check_launch("chamfer_forward");
}

template <typename S>
int chamfer_backward_impl(const S* grad_dist1, const S* grad_dist2, const S* xyz1, const S* xyz2,
                          const int64_t* idx1, const int64_t* idx2, int64_t batch, int64_t n1,
                          int64_t n2, S* grad_xyz1, S* grad_xyz2, void* stream) {
  MPA_REQUIRE(batch >= 0 && n1 >= 0 && n2 >= 0, "chamfer_backward: negative size");
  MPA_REQUIRE(n1 < (1 << 30) && n2 < (1 << 30), "chamfer_backward: cloud larger than 2^30 points");
  hipStream_t s = mpa::as_stream(stream);
  const long long total1 = batch * n1, total2 = batch * n2;
  if (total1 == 0 && total2 == 0) return MPA_OK;
  MPA_REQUIRE((total1 == 0 || (grad_dist1 && xyz1 && idx1 && grad_xyz1)) &&
                  (total2 == 0 || (grad_dist2 && xyz2 && idx2 && grad_xyz2)),
              "chamfer_backward: null pointer");
  if (total1 && hipMemsetAsync(grad_xyz1, 0, sizeof(S) * 3 * total1, s) != hipSuccess)
    return mpa::check_launch("chamfer_backward(memset)");
  if (total2 && hipMemsetAsync(grad_xyz2, 0, sizeof(S) * 3 * total2, s) != hipSuccess)
    return mpa::check_launch("chamfer_backward(memset)");
  if (total1 == 0 || total2 == 0) return MPA_OK;  // every idx is -1: nothing to scatter
  const long long tmax = total1 > total2 ? total1 : total2;
  long long blocks = (tmax + kThreads - 1) / kThreads;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(chamfer_grad_kernel<S>, dim3((unsigned)blocks, 2, 1), dim3(kThreads), 0, s,
                     grad_dist1, grad_dist2, xyz1, xyz2, (const long long*)idx1,
                     (const long long*)idx2, total1, total2, (int)n1, (int)n2, grad_xyz1,
                     grad_xyz2);
  return mpa::check_launch("chamfer_backward");
}

}  // namespace

extern "C" int mpa_chamfer_forward_variant(const float* xyz1, const float* xyz2, int64_t batch,
                                           int64_t n1, int64_t n2, float* dist1, int64_t* idx1,
                                           float* dist2, int64_t* idx2, int variant, void* stream) {
  return chamfer_forward_impl<float>(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2, variant,
                                     stream);
}

extern "C" int mpa_chamfer_forward(const float* xyz1, const float* xyz2, int64_t batch, int64_t n1,
                                   int64_t n2, float* dist1, int64_t* idx1, float* dist2,
                                   int64_t* idx2, void* stream) {
  return chamfer_forward_impl<float>(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2, 1,
                                     stream);
}

extern "C" int mpa_chamfer_forward_f64(const double* xyz1, const double* xyz2, int64_t batch,
                                       int64_t n1, int64_t n2, double* dist1, int64_t* idx1,
                                       double* dist2, int64_t* idx2, void* stream) {
  return chamfer_forward_impl<double>(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2, 1,
                                      stream);
}

extern "C" int mpa_chamfer_backward(const float* grad_dist1, const float* grad_dist2,
                                    const float* xyz1, const float* xyz2, const int64_t* idx1,
                                    const int64_t* idx2, int64_t batch, int64_t n1, int64_t n2,
                                    float* grad_xyz1, float* grad_xyz2, void* stream) {
  return chamfer_backward_impl<float>(grad_dist1, grad_dist2, xyz1, xyz2, idx1, idx2, batch, n1, n2,
                                      grad_xyz1, grad_xyz2, stream);
}

extern "C" int mpa_chamfer_backward_f64(const double* grad_dist1, const double* grad_dist2,
                                        const double* xyz1, const double* xyz2,
                                        const int64_t* idx1, const int64_t* idx2, int64_t batch,
                                        int64_t n1, int64_t n2, double* grad_xyz1,
                                        double* grad_xyz2, void* stream) {
  return chamfer_backward_impl<double>(grad_dist1, grad_dist2, xyz1, xyz2, idx1, idx2, batch, n1,
                                       n2, grad_xyz1, grad_xyz2, stream);
}
