// DGCNN building blocks for gfx950: k-nearest-neighbour graph in feature space and the edge-convolution
// aggregation (reference: multi_part_assembly/models/modules/encoder/dgcnn.py:8-38 `knn` / `get_graph_feature`,
// :41-109 `DGCNN`: 4 x [edge features -> Conv2d 1x1 -> BatchNorm2d -> LeakyReLU(0.2) -> max over the k neighbours]).
//
// The reference materialises, per stage, the [n, N, N] score matrix (2.6 GB at n = 640, N = 1000), the gathered
// neighbour tensor and the edge-feature tensor [n, 2C, N, k] (13 GB at C = 128) and then runs the 1x1 convolution
// over all N*k edges.  Here none of them exists:
//   * knn: one thread per query point keeps its k best candidates in registers while the block streams the part's
//     points through LDS (scores in the reference's form  -|x_j|^2 + 2 x_i.x_j - |x_i|^2);
//   * the convolution is linear in the edge feature:  W [x_j - x_i ; x_i] = Wa x_j + (Wb - Wa) x_i = U_j + V_i, so a
//     plain GEMM per POINT (library GEMM: X -> [U | V]) replaces the GEMM per EDGE (20x fewer FLOPs), and
//   * BatchNorm + LeakyReLU being a monotone per-channel map, max_j act(bn(e_ij)) = act(bn(max_j e_ij)) (or min_j
//     for a negative scale): the aggregation kernel gathers the 20 neighbour rows of U (L2-resident, one part at a
//     time), keeps max / min / arg-max / arg-min and the two sums BatchNorm's statistics need, and a second small
//     kernel applies scale/shift/LeakyReLU once the statistics are known.
// Backward: BatchNorm backward is the affine map  de = alpha*dz + gammap*e + betap  with dz nonzero only on the
// selected edge of every (point, channel);  dV_i = sum_j de_ij  and  dU_j = sum_{i: j in nn(i)} de_ij  (coalesced
// fp32 atomics over channels — the reference's own gather backward is an atomic scatter too).
#include "common.h"
#include "coop_reduce.h"

namespace {

using mpa::CoopWs;
using mpa::coop_colsum;
using mpa::kEB;
using mpa::kSlices;

constexpr int kMaxK = 32;

// ---- kNN ------------------------------------------------------------------------------------------------------------
// x [n*N, C] point-major.  grid = (ceil(N / 256), n), block 256: thread = query point.  Candidates are staged in
// LDS tiles of kTile rows and read back as broadcasts.  idx [n*N, K] int32, neighbour indices inside the part,
// best first; equal scores keep the lower index first.
template <int C, int K>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ x, int N, int* __restrict__ idx) {
  constexpr int CP = (C + 3) / 4 * 4;  // padded channel count (C = 3 -> 4)
  constexpr int kTile = 64;
  __shared__ __attribute__((aligned(16))) float tile[kTile][CP];
  __shared__ float tnorm[kTile];
  const int m = blockIdx.y, q = blockIdx.x * 256 + threadIdx.x;
  const float* xp = x + (long long)m * N * C;
  const int qi = q < N ? q : N - 1;
  float xq[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) xq[c] = c < C ? xp[(long long)qi * C + c] : 0.0f;
  // |x|^2: C = 3 follows the reference's CPU arithmetic (torch.sum(x**2): rounded squares added in order, no FMA —
  // bit-equal to torch on the fixture cloud, see csrc/dg_knn.h); wider features use the fmaf chain of the dot product
  float nq = 0.0f;
  if constexpr (C == 3) {
    nq = (xq[0] * xq[0] + xq[1] * xq[1]) + xq[2] * xq[2];
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) nq = __builtin_fmaf(xq[c], xq[c], nq);
  }
  float bs[K];
  int bj[K];
#pragma unroll
  for (int t = 0; t < K; ++t) {
    bs[t] = -__builtin_inff();
    bj[t] = 0;
  }
  for (int j0 = 0; j0 < N; j0 += kTile) {
    __syncthreads();
    for (int e = threadIdx.x; e < kTile * CP; e += 256) {
      const int r = e / CP, c = e % CP, j = j0 + r;
      tile[r][c] = (j < N && c < C) ? xp[(long long)j * C + c] : 0.0f;
    }
    __syncthreads();
    if (threadIdx.x < kTile) {
      float s = 0.0f;
      const float* tr = tile[threadIdx.x];
      if constexpr (C == 3) {
        s = (tr[0] * tr[0] + tr[1] * tr[1]) + tr[2] * tr[2];
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) s = __builtin_fmaf(tr[c], tr[c], s);
      }
      tnorm[threadIdx.x] = s;
    }
    __syncthreads();
    const int cnt = N - j0 < kTile ? N - j0 : kTile;
    for (int r = 0; r < cnt; ++r) {
      float dot = 0.0f;
#pragma unroll
      for (int c4 = 0; c4 < CP / 4; ++c4) {
        const float4 t = *reinterpret_cast<const float4*>(&tile[r][4 * c4]);
        dot = __builtin_fmaf(xq[4 * c4 + 0], t.x, dot);
        dot = __builtin_fmaf(xq[4 * c4 + 1], t.y, dot);
        dot = __builtin_fmaf(xq[4 * c4 + 2], t.z, dot);
        dot = __builtin_fmaf(xq[4 * c4 + 3], t.w, dot);
      }
      const float s = (-tnorm[r] + 2.0f * dot) - nq;  // dgcnn.py:11-13: -xx - (-2 x^T x) - xx^T
      // insert, keeping the list sorted (descending; earlier index first among equals).  64 independent query
      // streams share a wave, so some lane inserts at almost every candidate: the insertion is branch-free
      // (a bubble of selects), and skipped only when no lane of the wave needs it.
      if (__any(s > bs[K - 1])) {
        float cs = s;
        int cj = j0 + r;
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const bool g = s > bs[t];  // the NEW score against every slot: everything behind the insertion point shifts
          const float ts = bs[t];
          const int tj = bj[t];
          bs[t] = g ? cs : ts;
          bj[t] = g ? cj : tj;
          cs = g ? ts : cs;
          cj = g ? tj : cj;
        }
      }
    }
  }
  if (q < N) {
    int* out = idx + ((long long)m * N + q) * K;
#pragma unroll
    for (int t = 0; t < K; ++t) out[t] = bj[t];
  }
}

// ---- edge aggregation, forward ------------------------------------------------------------------------------------------
// uv [n*N, 2*CO] (U | V), idx [n*N, K].  grid = (ceil(N / kRows), n), block = CO threads (thread = channel).
// Per (point, channel): max / min over the K neighbours of U_j (+ V_i), their neighbour slots, and the block's
// partial sums of e and e^2 for the BatchNorm statistics (partial [blocks][CO][2]).
constexpr int kRows = 32;  // points per block

__global__ void edge_gather_kernel(const float* __restrict__ uv, const int* __restrict__ idx, int N, int CO, int K,
                                   float* __restrict__ emax, float* __restrict__ emin,
                                   unsigned char* __restrict__ smax, unsigned char* __restrict__ smin,
                                   float* __restrict__ partial) {
  __shared__ int nbr[kRows][kMaxK];
  const int m = blockIdx.y, r0 = blockIdx.x * kRows, c = threadIdx.x;
  const int rows = N - r0 < kRows ? N - r0 : kRows;
  for (int e = threadIdx.x; e < rows * K; e += blockDim.x)
    nbr[e / K][e % K] = idx[((long long)m * N + r0) * K + e];
  __syncthreads();
  const float* up = uv + (long long)m * N * 2 * CO;
  float s1 = 0.0f, s2 = 0.0f;
  for (int r = 0; r < rows; ++r) {
    const long long row = (long long)m * N + r0 + r;
    const float v = uv[row * 2 * CO + CO + c];
    float mx = -__builtin_inff(), mn = __builtin_inff();
    int ax = 0, an = 0;
    for (int t = 0; t < K; ++t) {
      const float e = up[(long long)nbr[r][t] * 2 * CO + c] + v;
      s1 += e;
      s2 = __builtin_fmaf(e, e, s2);
      if (e > mx) {
        mx = e;
        ax = t;
      }
      if (e < mn) {
        mn = e;
        an = t;
      }
    }
    emax[row * CO + c] = mx;
    emin[row * CO + c] = mn;
    smax[row * CO + c] = (unsigned char)ax;
    smin[row * CO + c] = (unsigned char)an;
  }
  const long long o = (((long long)m * gridDim.x + blockIdx.x) * CO + c) * 2;
  partial[o] = s1;
  partial[o + 1] = s2;
}

// BatchNorm statistics of the edge values -> bn [4][CO] (scale, shift, mean, invstd) + running statistics.
// grid = (CO/64, ceil(rows/kEB)), block 1024.
__global__ __launch_bounds__(64 * kSlices) void edge_bn_finalize_kernel(
    const float* __restrict__ partial, int blocks, int CO, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    float momentum, float eps, float* __restrict__ bn, const CoopWs cw) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, ss;
  const bool last = coop_colsum(blocks, CO, c, cw,
                                [&](int e, bool& ok, double& x, double& y) {
                                  const float2 v = *reinterpret_cast<const float2*>(partial + ((long long)e * CO + c) * 2);
                                  ok = true;
                                  x = (double)v.x;
                                  y = (double)v.y;
                                },
                                s, ss);
  if (!last || threadIdx.x >= 64) return;
  const double mean = s / count;
  double var = ss / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / __builtin_sqrt(var + (double)eps));
  const float scale = gamma[c] * invstd;
  bn[c] = scale;
  bn[CO + c] = beta[c] - (float)mean * scale;
  bn[2 * CO + c] = (float)mean;
  bn[3 * CO + c] = invstd;
  if (running_mean != nullptr) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void edge_bn_from_running_kernel(int CO, const float* __restrict__ gamma, const float* __restrict__ beta,
                                            const float* __restrict__ running_mean,
                                            const float* __restrict__ running_var, float eps,
                                            float* __restrict__ bn) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= CO) return;
  const float invstd = 1.0f / __builtin_sqrtf(running_var[c] + eps);
  const float scale = gamma[c] * invstd;
  bn[c] = scale;
  bn[CO + c] = beta[c] - running_mean[c] * scale;
  bn[2 * CO + c] = running_mean[c];
  bn[3 * CO + c] = invstd;
}

// out = LeakyReLU(bn(selected edge)); keeps the selected edge value and its neighbour slot for backward.
// One thread per (point, channel).
__global__ void edge_apply_kernel(const float* __restrict__ emax, const float* __restrict__ emin,
                                  const unsigned char* __restrict__ smax, const unsigned char* __restrict__ smin,
                                  const float* __restrict__ bn, long long total, int CO, float* __restrict__ out,
                                  float* __restrict__ esel, unsigned char* __restrict__ ssel) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % CO);
  const float scale = bn[c], shift = bn[CO + c];
  const bool hi = scale >= 0.0f;  // scale == 0: every edge maps to `shift`; the first maximal one is as good as any
  const float e = hi ? emax[i] : emin[i];
  const float z = __builtin_fmaf(e, scale, shift);
  out[i] = z > 0.0f ? z : 0.2f * z;
  esel[i] = e;
  ssel[i] = hi ? smax[i] : smin[i];
}

// ---- edge aggregation, backward ---------------------------------------------------------------------------------------------
// dz = grad_out * LeakyReLU'(z) on the selected edges; block partial sums of dz and dz * ehat.
// grid = (ceil(N / kRows), n), block = CO.
__global__ void edge_bwd_sums_kernel(const float* __restrict__ gout, const float* __restrict__ esel,
                                     const float* __restrict__ bn, int N, int CO, float* __restrict__ dz,
                                     float* __restrict__ partial) {
  const int m = blockIdx.y, r0 = blockIdx.x * kRows, c = threadIdx.x;
  const int rows = N - r0 < kRows ? N - r0 : kRows;
  const float scale = bn[c], shift = bn[CO + c], mean = bn[2 * CO + c], invstd = bn[3 * CO + c];
  float s1 = 0.0f, s2 = 0.0f;
  for (int r = 0; r < rows; ++r) {
    const long long o = ((long long)m * N + r0 + r) * CO + c;
    const float e = esel[o], z = __builtin_fmaf(e, scale, shift);
    const float d = gout[o] * (z > 0.0f ? 1.0f : 0.2f);
    dz[o] = d;
    s1 += d;
    s2 = __builtin_fmaf(d, (e - mean) * invstd, s2);
  }
  const long long o = (((long long)m * gridDim.x + blockIdx.x) * CO + c) * 2;
  partial[o] = s1;
  partial[o + 1] = s2;
}

// coefficients [3][CO] (alpha, gammap, betap) of  de = alpha*dz + gammap*e + betap, and dgamma / dbeta.
__global__ __launch_bounds__(64 * kSlices) void edge_bwd_coef_kernel(
    const float* __restrict__ partial, int blocks, int CO, double count, const float* __restrict__ gamma,
    const float* __restrict__ bn, float* __restrict__ coef, float* __restrict__ dgamma,
    float* __restrict__ dbeta, const CoopWs cw) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s1, s2;
  const bool last = coop_colsum(blocks, CO, c, cw,
                                [&](int e, bool& ok, double& x, double& y) {
                                  const float2 v = *reinterpret_cast<const float2*>(partial + ((long long)e * CO + c) * 2);
                                  ok = true;
                                  x = (double)v.x;
                                  y = (double)v.y;
                                },
                                s1, s2);
  if (!last || threadIdx.x >= 64) return;
  const float mean = bn[2 * CO + c], invstd = bn[3 * CO + c];
  const float alpha = gamma[c] * invstd;
  const float gammap = (float)(-(double)alpha * s2 / count * (double)invstd);
  coef[c] = alpha;
  coef[CO + c] = gammap;
  coef[2 * CO + c] = (float)(-(double)alpha * s1 / count - (double)gammap * (double)mean);
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
}

// ---- reverse adjacency ------------------------------------------------------------------------------------------------
// The EdgeConv backward needs, per point j, the sum over the points i that have j among their K neighbours.  A
// scatter with atomics does that in a run-dependent order (and took 12 of the 33 ms of a DGCNN forward+backward at
// 352 parts); instead the kNN graph of every part is transposed once per layer: rptr [n][N+1] / rlist [n][N*K],
// entries (i << 5 | slot) sorted ascending, so the gather below is atomic-free and bit-reproducible.
// grid = n parts, block 1024; N <= kRevMaxN (the in-degree counters live in LDS).
constexpr int kRevMaxN = 16384;

__global__ __launch_bounds__(1024) void edge_reverse_kernel(const int* __restrict__ idx, int N, int K,
                                                            int* __restrict__ rptr, int* __restrict__ rlist) {
  __shared__ int cnt[kRevMaxN];
  __shared__ int wsum[16];
  __shared__ int carry;
  const int m = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const long long E = (long long)N * K;
  const int* id = idx + (long long)m * E;
  int* rp = rptr + (long long)m * (N + 1);
  int* rl = rlist + (long long)m * E;
  for (int j = t; j < N; j += 1024) cnt[j] = 0;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int e = t; e < E; e += 1024) atomicAdd(&cnt[id[e]], 1);  // integer counts: order-independent
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {  // exclusive prefix sum, 1024 rows at a time
    const int j = base + t, v = j < N ? cnt[j] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (j < N) {
      rp[j] = before + inc - v;
      cnt[j] = before + inc - v;  // becomes the row's fill cursor
    }
    __syncthreads();
    if (t == 1023) carry = before + inc;
    __syncthreads();
  }
  if (t == 0) rp[N] = carry;
  for (int e = t; e < E; e += 1024) {
    const int pos = atomicAdd(&cnt[id[e]], 1);
    rl[pos] = ((e / K) << 5) | (e % K);
  }
  __syncthreads();
  for (int j = t; j < N; j += 1024) {  // ascending order inside every row: fixed summation order downstream
    const int b = rp[j], e = cnt[j];
    for (int a = b + 1; a < e; ++a) {
      const int key = rl[a];
      int q = a - 1;
      while (q >= b && rl[q] > key) {
        rl[q + 1] = rl[q];
        --q;
      }
      rl[q + 1] = key;
    }
  }
}

// d(uv) without atomics: with de_ij = gammap*(U_j + V_i) + betap (+ alpha*dz_i for the selected neighbour),
//   dV_i = gammap * sum_t U_{nbr(i,t)} + K*(gammap*V_i + betap) + [a neighbour was selected] alpha*dz_i
//   dU_j = gammap * (deg_j*U_j + sum_{i in in(j)} V_i) + deg_j*betap + sum_{(i,t) in in(j), sel_i == t} alpha*dz_i
// grid = (ceil(N / kRows), n), block = CO (thread = channel); the in-edge list of a row is wave-uniform.
__global__ void edge_bwd_gather_kernel(const float* __restrict__ uv, const int* __restrict__ idx,
                                       const int* __restrict__ rptr, const int* __restrict__ rlist,
                                       const float* __restrict__ dz, const unsigned char* __restrict__ ssel,
                                       const float* __restrict__ coef, int N, int CO, int K,
                                       float* __restrict__ guv) {
  __shared__ int nbr[kRows][kMaxK];
  const int m = blockIdx.y, r0 = blockIdx.x * kRows, c = threadIdx.x;
  const int rows = N - r0 < kRows ? N - r0 : kRows;
  for (int e = threadIdx.x; e < rows * K; e += blockDim.x)
    nbr[e / K][e % K] = idx[((long long)m * N + r0) * K + e];
  __syncthreads();
  const float alpha = coef[c], gammap = coef[CO + c], betap = coef[2 * CO + c];
  const float* up = uv + (long long)m * N * 2 * CO;
  const float* dzp = dz + (long long)m * N * CO;
  const unsigned char* sp = ssel + (long long)m * N * CO;
  const int* rp = rptr + (long long)m * (N + 1);
  const int* rl = rlist + (long long)m * N * K;
  for (int r = 0; r < rows; ++r) {
    const int i = r0 + r;
    const long long row = (long long)m * N + i;
    const float u = up[(long long)i * 2 * CO + c], v = up[(long long)i * 2 * CO + CO + c];
    float su = 0.0f;
    for (int t = 0; t < K; ++t) su += up[(long long)nbr[r][t] * 2 * CO + c];
    const int sel = sp[(long long)i * CO + c];
    float dv = __builtin_fmaf(gammap, su, (float)K * __builtin_fmaf(gammap, v, betap));
    if (sel < K) dv += alpha * dzp[(long long)i * CO + c];
    const int b = rp[i], e = rp[i + 1];
    float sv = 0.0f, sd = 0.0f;
    for (int q = b; q < e; ++q) {
      const int key = rl[q], src = key >> 5, slot = key & 31;
      sv += up[(long long)src * 2 * CO + CO + c];
      if (sp[(long long)src * CO + c] == slot) sd += alpha * dzp[(long long)src * CO + c];
    }
    const float deg = (float)(e - b);
    guv[row * 2 * CO + c] = __builtin_fmaf(gammap, __builtin_fmaf(deg, u, sv), deg * betap) + sd;
    guv[row * 2 * CO + CO + c] = dv;
  }
}

// d(uv): dV_i = sum_j de_ij (own row), dU_j += de_ij (atomics, coalesced over channels).  grad_uv zero-filled.
// grid = (ceil(N / kRows), n), block = CO.
__global__ void edge_bwd_scatter_kernel(const float* __restrict__ uv, const int* __restrict__ idx,
                                        const float* __restrict__ dz, const unsigned char* __restrict__ ssel,
                                        const float* __restrict__ coef, int N, int CO, int K,
                                        float* __restrict__ guv) {
  __shared__ int nbr[kRows][kMaxK];
  const int m = blockIdx.y, r0 = blockIdx.x * kRows, c = threadIdx.x;
  const int rows = N - r0 < kRows ? N - r0 : kRows;
  for (int e = threadIdx.x; e < rows * K; e += blockDim.x)
    nbr[e / K][e % K] = idx[((long long)m * N + r0) * K + e];
  __syncthreads();
  const float alpha = coef[c], gammap = coef[CO + c], betap = coef[2 * CO + c];
  const float* up = uv + (long long)m * N * 2 * CO;
  float* gp = guv + (long long)m * N * 2 * CO;
  for (int r = 0; r < rows; ++r) {
    const long long row = (long long)m * N + r0 + r;
    const float v = uv[row * 2 * CO + CO + c];
    const float d = alpha * dz[row * CO + c];
    const int sel = ssel[row * CO + c];
    float dv = 0.0f;
    for (int t = 0; t < K; ++t) {
      const int j = nbr[r][t];
      const float e = up[(long long)j * 2 * CO + c] + v;
      float de = __builtin_fmaf(gammap, e, betap);
      if (t == sel) de += d;
      dv += de;
      atomicAdd(gp + (long long)j * 2 * CO + c, de);
    }
    guv[row * 2 * CO + CO + c] = dv;
  }
}

struct EdgeWs {
  float *emax, *emin, *esel, *dz, *partial, *bn, *coef;
  unsigned char *smax, *smin, *ssel;
  int *rptr, *rlist;  // reverse adjacency of the kNN graph (backward)
  CoopWs coop;
  int64_t total;  // bytes
};

EdgeWs edge_carve(char* base, int64_t n, int64_t N, int64_t CO, int64_t K) {
  EdgeWs w;
  char* p = base;
  auto take = [&](int64_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  const int64_t R = n * N, blocks = n * ((N + kRows - 1) / kRows);
  w.emax = reinterpret_cast<float*>(take(4 * R * CO));
  w.emin = reinterpret_cast<float*>(take(4 * R * CO));
  w.esel = reinterpret_cast<float*>(take(4 * R * CO));
  w.dz = w.emax;  // backward reuses the forward scratch
  w.smax = reinterpret_cast<unsigned char*>(take(R * CO));
  w.smin = reinterpret_cast<unsigned char*>(take(R * CO));
  w.ssel = reinterpret_cast<unsigned char*>(take(R * CO));
  w.partial = reinterpret_cast<float*>(take(4 * blocks * CO * 2));
  w.bn = reinterpret_cast<float*>(take(4 * 4 * CO));
  w.coef = reinterpret_cast<float*>(take(4 * 4 * CO));
  w.rptr = reinterpret_cast<int*>(take(4 * n * (N + 1)));
  w.rlist = reinterpret_cast<int*>(take(4 * n * N * K));
  w.coop.ticket = reinterpret_cast<unsigned*>(take(64));
  w.coop.stage = reinterpret_cast<double*>(take(8 * 2 * CO * ((blocks + kEB - 1) / kEB)));
  w.total = p - base;
  return w;
}

int edge_check(int64_t n, int64_t N, int64_t CO, int64_t K, const char* who) {
  MPA_REQUIRE(n >= 0 && N >= 1 && N <= 65535, "%s: bad sizes", who);
  MPA_REQUIRE(CO >= 64 && CO <= 1024 && CO % 64 == 0, "%s: output channels must be a multiple of 64 (<= 1024)", who);
  MPA_REQUIRE(K >= 1 && K <= kMaxK && K <= N, "%s: 1 <= k <= 32", who);
  return MPA_OK;
}

}  // namespace

extern "C" int mpa_knn(const float* x, int64_t n, int64_t N, int64_t C, int64_t K, int32_t* idx, void* stream) {
  MPA_REQUIRE(n >= 0 && N >= 1 && K == 20 && K <= N, "knn: k must be 20 (<= N)");
  MPA_REQUIRE(C == 3 || C == 64 || C == 128, "knn: feature width must be 3, 64 or 128");
  if (n == 0) return MPA_OK;
  MPA_REQUIRE(x && idx, "knn: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const dim3 grid((unsigned)((N + 255) / 256), (unsigned)n);
  if (C == 3) hipLaunchKernelGGL((knn_kernel<3, 20>), grid, dim3(256), 0, s, x, (int)N, idx);
  else if (C == 64) hipLaunchKernelGGL((knn_kernel<64, 20>), grid, dim3(256), 0, s, x, (int)N, idx);
  else hipLaunchKernelGGL((knn_kernel<128, 20>), grid, dim3(256), 0, s, x, (int)N, idx);
  return mpa::check_launch("knn");
}

extern "C" int mpa_edge_aggregate_workspace(int64_t n, int64_t N, int64_t CO, int64_t K, int64_t* bytes) {
  if (int st = edge_check(n, N, CO, K, "edge_aggregate_workspace")) return st;
  MPA_REQUIRE(bytes != nullptr, "edge_aggregate_workspace: null pointer");
  *bytes = edge_carve(nullptr, n, N, CO, K).total;
  return MPA_OK;
}

extern "C" int mpa_edge_aggregate_forward(const float* uv, const int32_t* idx, const float* gamma, const float* beta,
                                          float* running_mean, float* running_var, int training, float momentum,
                                          float eps, int64_t n, int64_t N, int64_t CO, int64_t K, void* ws,
                                          float* out, void* stream) {
  if (int st = edge_check(n, N, CO, K, "edge_aggregate_forward")) return st;
  if (n == 0) return MPA_OK;
  MPA_REQUIRE(uv && idx && gamma && beta && running_mean && running_var && ws && out,
              "edge_aggregate_forward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const EdgeWs w = edge_carve(static_cast<char*>(ws), n, N, CO, K);
  const unsigned tiles = (unsigned)((N + kRows - 1) / kRows);
  const int blocks = (int)(n * tiles);
  mpa::zero_words_async(w.coop.ticket, 16, s);
  hipLaunchKernelGGL(edge_gather_kernel, dim3(tiles, (unsigned)n), dim3((unsigned)CO), 0, s, uv, idx, (int)N, (int)CO,
                     (int)K, w.emax, w.emin, w.smax, w.smin, w.partial);
  if (training)
    hipLaunchKernelGGL(edge_bn_finalize_kernel, dim3((unsigned)(CO / 64), (unsigned)((blocks + kEB - 1) / kEB)),
                       dim3(64 * kSlices), 0, s, w.partial, blocks, (int)CO, (double)n * (double)N * (double)K, gamma,
                       beta, running_mean, running_var, momentum, eps, w.bn, w.coop);
  else
    hipLaunchKernelGGL(edge_bn_from_running_kernel, dim3((unsigned)(CO / 64)), dim3(64), 0, s, (int)CO, gamma, beta,
                       running_mean, running_var, eps, w.bn);
  const long long total = (long long)n * N * CO;
  hipLaunchKernelGGL(edge_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w.emax, w.emin, w.smax,
                     w.smin, w.bn, total, (int)CO, out, w.esel, w.ssel);
  return mpa::check_launch("edge_aggregate_forward");
}

extern "C" int mpa_edge_aggregate_backward(const float* grad_out, const float* uv, const int32_t* idx,
                                           const float* gamma, int64_t n, int64_t N, int64_t CO, int64_t K, void* ws,
                                           float* grad_uv, float* grad_gamma, float* grad_beta, void* stream) {
  if (int st = edge_check(n, N, CO, K, "edge_aggregate_backward")) return st;
  if (n == 0) return MPA_OK;
  MPA_REQUIRE(grad_out && uv && idx && gamma && ws && grad_uv && grad_gamma && grad_beta,
              "edge_aggregate_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const EdgeWs w = edge_carve(static_cast<char*>(ws), n, N, CO, K);
  const unsigned tiles = (unsigned)((N + kRows - 1) / kRows);
  const int blocks = (int)(n * tiles);
  hipLaunchKernelGGL(edge_bwd_sums_kernel, dim3(tiles, (unsigned)n), dim3((unsigned)CO), 0, s, grad_out, w.esel, w.bn,
                     (int)N, (int)CO, w.dz, w.partial);
  hipLaunchKernelGGL(edge_bwd_coef_kernel, dim3((unsigned)(CO / 64), (unsigned)((blocks + kEB - 1) / kEB)),
                     dim3(64 * kSlices), 0, s, w.partial, blocks, (int)CO, (double)n * (double)N * (double)K, gamma, w.bn,
                     w.coef, grad_gamma, grad_beta, w.coop);
  if (N <= kRevMaxN) {  // transpose the kNN graph, then gather: no atomics, bit-reproducible
    hipLaunchKernelGGL(edge_reverse_kernel, dim3((unsigned)n), dim3(1024), 0, s, idx, (int)N, (int)K, w.rptr, w.rlist);
    hipLaunchKernelGGL(edge_bwd_gather_kernel, dim3(tiles, (unsigned)n), dim3((unsigned)CO), 0, s, uv, idx, w.rptr,
                       w.rlist, w.dz, w.ssel, w.coef, (int)N, (int)CO, (int)K, grad_uv);
  } else {  // very large parts: the in-degree counters no longer fit in LDS
    mpa::zero_words_async(grad_uv, n * N * 2 * CO, s);
    hipLaunchKernelGGL(edge_bwd_scatter_kernel, dim3(tiles, (unsigned)n), dim3((unsigned)CO), 0, s, uv, idx, w.dz,
                       w.ssel, w.coef, (int)N, (int)CO, (int)K, grad_uv);
  }
  return mpa::check_launch("edge_aggregate_backward");
}
