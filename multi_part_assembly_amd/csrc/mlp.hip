// One layer of the graph networks' MLPs — Linear (+ bias) [-> BatchNorm1d (training or running statistics)] [-> ReLU]
// over R rows, forward and backward — for gfx950.
//
// Replaces the Conv1d(k=1) + BatchNorm1d + ReLU layers of the reference's MLP3 / MLP4 / MLP5 and the Linear + ReLU
// layers of RelationNet (multi_part_assembly/models/dgl/modules.py:5-73, models/rgl_net/modules.py:5-30): the P x P edge
// MLP runs them over B*P*P = 12 800 pair rows three times per training step (dgl/network.py:135-152), the node MLP
// over B*P rows.  The GEMMs are the exact-fp32 matrix-core kernels of dg_gemm.h (weights [Nout, K] as PyTorch stores
// them); BatchNorm statistics are fixed-order two-stage sums in double; the backward is the affine map
//   dY = alpha * dz + gammap * Y + betap,  dz = dOut * [out > 0]
// followed by dW = dY^T X (row-chunked, fixed order), db = column sums of dY, dX = dY W.  No atomics: bit-reproducible.
// Padded pairs are rows like any other (the reference's BatchNorm sees them too, dgl/network.py:139-144).
#include <stdlib.h>

#include "common.h"
#include "coop_reduce.h"
#include "dg_gemm.h"
#include "dg_gemm_split.h"
#include "tf_gemm.h"
#ifndef DG_GEMM_SPLIT  // 1: fp32-grade GEMMs on the bf16 matrix cores (dg_gemm_split.h); 0: v_mfma_f32_32x32x2_f32 (dg_gemm.h)
#define DG_GEMM_SPLIT 1
#endif
#if DG_GEMM_SPLIT
#define DG_NT_KERNEL gemm_nt_split_kernel
#define DG_TN_KERNEL gemm_tn_split_kernel
#define DG_GEMM_THREADS kGsT
#else
#define DG_NT_KERNEL gemm_nt_kernel
#define DG_TN_KERNEL gemm_tn_kernel
#define DG_GEMM_THREADS kGT
#endif

namespace {

using namespace dg;
using mpa::CoopWs;
using mpa::coop_colsum;
using mpa::kEB;
using mpa::kSlices;

constexpr int kRT = 16;          // rows per block of the row-tiled kernels
// Layers of at most this many rows (the node MLPs: B*P = 640) take the GEMMs of tf_gemm.h — one 32 x 32 tile per block,
// the block's 8 waves split K, exact-fp32 matrix-core products: ~11 us a call where the 128-row tiles of
// dg_gemm_split.h leave 5 row tiles to walk K = 512 in 16 dependent steps (22-34 us).  Above it the fp32 matrix-core
// rate (157 TFLOP/s) loses to the split-bf16 kernels.
#ifndef MPA_ML_SMALL_ROWS
#define MPA_ML_SMALL_ROWS 2048
#endif
constexpr int kSmallRows = MPA_ML_SMALL_ROWS;
constexpr int kChunks = 64;      // most row chunks of the weight-gradient GEMM (fewer for few rows: >= 256 rows each)

__global__ void ml_set_hdr_kernel(int* hdr, int R, unsigned* tickets) {
  hdr[0] = 1;
  hdr[1] = R;
  for (int t = threadIdx.x; t < 64; t += blockDim.x) tickets[t] = 0u;
}

// y[r][c] += bias[c] (in place) and the per-tile column sums (sum y, sum y^2).  grid = tiles, block = 256 (channels in
// chunks of 256).
__global__ __launch_bounds__(256) void ml_bias_stats_kernel(float* __restrict__ y, const float* __restrict__ bias, int R,
                                                            int C, float* __restrict__ partial) {
  const long long r0 = (long long)blockIdx.x * kRT;
  const int rows = R - r0 < kRT ? (int)(R - r0) : kRT;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float b = bias != nullptr ? bias[c] : 0.0f;
    float s = 0.0f, ss = 0.0f;
    for (int i = 0; i < rows; ++i) {
      const float t = y[(r0 + i) * C + c] + b;
      y[(r0 + i) * C + c] = t;
      s += t;
      ss = __builtin_fmaf(t, t, ss);
    }
    float* d = partial + ((long long)blockIdx.x * C + c) * 2;
    d[0] = s;
    d[1] = ss;
  }
}

__global__ __launch_bounds__(64 * kSlices) void ml_bn_finalize_kernel(
    const float* __restrict__ partial, int rows, int C, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
    float eps, float* __restrict__ bn, const CoopWs cw) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, ss;
  const bool last = coop_colsum(rows, C, c, cw,
                                [&](int e, bool& ok, double& x, double& y) {
                                  const float2 t = *reinterpret_cast<const float2*>(partial + ((long long)e * C + c) * 2);
                                  ok = true;
                                  x = (double)t.x;
                                  y = (double)t.y;
                                },
                                s, ss);
  if (!last || threadIdx.x >= 64) return;
  const double mean = s / count;
  double var = ss / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / __builtin_sqrt(var + (double)eps));
  const float scale = gamma[c] * invstd;
  bn[c] = scale;
  bn[C + c] = beta[c] - (float)mean * scale;
  bn[2 * C + c] = (float)mean;
  bn[3 * C + c] = invstd;
  if (running_mean != nullptr) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void ml_bn_from_running_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                          const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                          float eps, float* __restrict__ bn) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / __builtin_sqrtf(running_var[c] + eps);
  const float scale = gamma[c] * invstd;
  bn[c] = scale;
  bn[C + c] = beta[c] - running_mean[c] * scale;
  bn[2 * C + c] = running_mean[c];
  bn[3 * C + c] = invstd;
}

// out = act(y * scale + shift) (bn != NULL) or act(y + bias) (bn == NULL); one thread per float4.
__global__ void ml_apply_kernel(const float* __restrict__ y, const float* __restrict__ bn, const float* __restrict__ bias,
                                long long total4, int C, int relu, float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total4) return;
  // column quad of element e = 256 blockIdx.x + threadIdx.x without a 64-bit remainder per thread
  const unsigned q = (unsigned)(C / 4);
  const int c = 4 * (int)(((blockIdx.x % q) * (blockDim.x % q) + threadIdx.x) % q);
  const float4 x = reinterpret_cast<const float4*>(y)[e];
  float4 z;
  if (bn != nullptr) {
    const float4 sc = *reinterpret_cast<const float4*>(bn + c), sh = *reinterpret_cast<const float4*>(bn + C + c);
    z = make_float4(__builtin_fmaf(x.x, sc.x, sh.x), __builtin_fmaf(x.y, sc.y, sh.y), __builtin_fmaf(x.z, sc.z, sh.z),
                    __builtin_fmaf(x.w, sc.w, sh.w));
  } else if (bias != nullptr) {
    const float4 b = *reinterpret_cast<const float4*>(bias + c);
    z = make_float4(x.x + b.x, x.y + b.y, x.z + b.z, x.w + b.w);
  } else {
    z = x;
  }
  if (relu) {
    z.x = z.x > 0.0f ? z.x : 0.0f;
    z.y = z.y > 0.0f ? z.y : 0.0f;
    z.z = z.z > 0.0f ? z.z : 0.0f;
    z.w = z.w > 0.0f ? z.w : 0.0f;
  }
  reinterpret_cast<float4*>(out)[e] = z;
}

// backward, BatchNorm layers: per-tile sums of dz and dz * xhat with dz = g * [out > 0] (relu) or g.
__global__ __launch_bounds__(256) void ml_bwd_sums_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                          const float* __restrict__ y, const float* __restrict__ bn, int R,
                                                          int C, int relu, float* __restrict__ partial) {
  const long long r0 = (long long)blockIdx.x * kRT;
  const int rows = R - r0 < kRT ? (int)(R - r0) : kRT;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mean = bn[2 * C + c], invstd = bn[3 * C + c];
    float s1 = 0.0f, s2 = 0.0f;
    if (rows == kRT) {  // a full tile: all of its loads in flight at once (a rolled loop waits for three loads per row);
      float gv[kRT], ov[kRT], yv[kRT];  // the sums run in the same row order either way
#pragma unroll
      for (int i = 0; i < kRT; ++i) {
        const long long o = (r0 + i) * C + c;
        gv[i] = g[o];
        ov[i] = out[o];  // (unconditional: a load under `relu ?` becomes a branch with a full wait behind it)
        yv[i] = y[o];
      }
#pragma unroll
      for (int i = 0; i < kRT; ++i) {
        const float d = (relu == 0 || ov[i] > 0.0f) ? gv[i] : 0.0f;
        s1 += d;
        s2 = __builtin_fmaf(d, (yv[i] - mean) * invstd, s2);
      }
    } else {
      for (int i = 0; i < rows; ++i) {
        const long long o = (r0 + i) * C + c;
        const float d = relu ? (out[o] > 0.0f ? g[o] : 0.0f) : g[o];
        s1 += d;
        s2 = __builtin_fmaf(d, (y[o] - mean) * invstd, s2);
      }
    }
    float* p = partial + ((long long)blockIdx.x * C + c) * 2;
    p[0] = s1;
    p[1] = s2;
  }
}

__global__ __launch_bounds__(64 * kSlices) void ml_bwd_coef_kernel(const float* __restrict__ partial, int rows, int C,
                                                                   double count, const float* __restrict__ gamma,
                                                                   const float* __restrict__ bn, float* __restrict__ coef,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   const CoopWs cw) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s1, s2;
  const bool last = coop_colsum(rows, C, c, cw,
                                [&](int e, bool& ok, double& x, double& y) {
                                  const float2 t = *reinterpret_cast<const float2*>(partial + ((long long)e * C + c) * 2);
                                  ok = true;
                                  x = (double)t.x;
                                  y = (double)t.y;
                                },
                                s1, s2);
  if (!last || threadIdx.x >= 64) return;
  const float mean = bn[2 * C + c], invstd = bn[3 * C + c];
  const float alpha = gamma[c] * invstd;
  const float gammap = (float)(-(double)alpha * s2 / count * (double)invstd);
  coef[c] = alpha;
  coef[C + c] = gammap;
  coef[2 * C + c] = (float)(-(double)alpha * s1 / count - (double)gammap * (double)mean);
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
}

// dY = alpha * dz + gammap * y + betap (coef != NULL) or dz (coef == NULL); also the per-tile column sums of dY (the
// bias gradient).  grid = tiles, block 256.
__global__ __launch_bounds__(256) void ml_bwd_dy_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                        const float* __restrict__ y, const float* __restrict__ coef, int R,
                                                        int C, int relu, float* __restrict__ dy,
                                                        float* __restrict__ partial) {
  const long long r0 = (long long)blockIdx.x * kRT;
  const int rows = R - r0 < kRT ? (int)(R - r0) : kRT;
  for (int c = threadIdx.x; c < C; c += 256) {
    float alpha = 1.0f, gammap = 0.0f, betap = 0.0f;
    if (coef != nullptr) {
      alpha = coef[c];
      gammap = coef[C + c];
      betap = coef[2 * C + c];
    }
    float s = 0.0f;
    if (rows == kRT) {  // full tile: loads up front, as in ml_bwd_sums_kernel
      float gv[kRT], ov[kRT], yv[kRT];
#pragma unroll
      for (int i = 0; i < kRT; ++i) {
        const long long o = (r0 + i) * C + c;
        gv[i] = g[o];
        ov[i] = out[o];  // (unconditional: a load under `relu ?` becomes a branch with a full wait behind it)
        yv[i] = y[o];
      }
#pragma unroll
      for (int i = 0; i < kRT; ++i) {
        const float d = (relu == 0 || ov[i] > 0.0f) ? gv[i] : 0.0f;
        const float v = coef != nullptr ? __builtin_fmaf(alpha, d, __builtin_fmaf(gammap, yv[i], betap)) : d;
        dy[(r0 + i) * C + c] = v;
        s += v;
      }
    } else {
      for (int i = 0; i < rows; ++i) {
        const long long o = (r0 + i) * C + c;
        const float d = relu ? (out[o] > 0.0f ? g[o] : 0.0f) : g[o];
        const float v = coef != nullptr ? __builtin_fmaf(alpha, d, __builtin_fmaf(gammap, y[o], betap)) : d;
        dy[o] = v;
        s += v;
      }
    }
    partial[(long long)blockIdx.x * C + c] = s;
  }
}

// ---- few rows (R <= kSmallRows: the node MLPs' 640 rows): BatchNorm's reductions inside ONE launch each way ----------------
// A block owns 32 channels and ALL rows (32 row lanes x 32 channel lanes), so the column sums need no second launch: the
// three launches of the row-tiled path (sums -> coefficients -> dY; 8 + 7 + 10 us of launch and dependency latency for
// 0.6 MB of data) become one, the forward's two (statistics -> apply) likewise.  Fixed-order sums: fp32 over a lane's
// rows, double across the 32 row lanes.
constexpr int kSbT = 1024;

// forward: statistics from the GEMM's per-32-row table partial [tiles][C][2] -> bn [4][C] (+ running statistics), then
// out = act(y * scale + shift) for the block's row slab.  grid = (C / 32, slabs): every slab recomputes its channels'
// statistics (a few hundred additions); slab 0 writes them.
__global__ __launch_bounds__(kSbT) void ml_small_bn_apply_kernel(
    const float* __restrict__ partial, int tiles, const float* __restrict__ y, int R, int C, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, float momentum, float eps, int relu, float* __restrict__ bn, float* __restrict__ out) {
  __shared__ double red[32][32][2];
  __shared__ float ss_[2][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  double s = 0.0, ss = 0.0;
  for (int t = rl; t < tiles; t += 32) {
    const float2 v = *reinterpret_cast<const float2*>(partial + ((long long)t * C + c) * 2);
    s += (double)v.x;
    ss += (double)v.y;
  }
  red[rl][cl][0] = s;
  red[rl][cl][1] = ss;
  __syncthreads();
  if (rl == 0) {
    s = ss = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      s += red[k][cl][0];
      ss += red[k][cl][1];
    }
    const double mean = s / count;
    double var = ss / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / __builtin_sqrt(var + (double)eps));
    const float scale = gamma[c] * invstd, shift = beta[c] - (float)mean * scale;
    ss_[0][cl] = scale;
    ss_[1][cl] = shift;
    if (blockIdx.y == 0) {
      bn[c] = scale;
      bn[C + c] = shift;
      bn[2 * C + c] = (float)mean;
      bn[3 * C + c] = invstd;
      if (running_mean != nullptr) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
      }
    }
  }
  __syncthreads();
  const float scale = ss_[0][cl], shift = ss_[1][cl];
  const int rows_per = (R + (int)gridDim.y - 1) / (int)gridDim.y, r0 = blockIdx.y * rows_per;
  const int r1 = r0 + rows_per < R ? r0 + rows_per : R;
#pragma unroll 8
  for (int r = r0 + rl; r < r1; r += 32) {
    float z = __builtin_fmaf(y[(long long)r * C + c], scale, shift);
    if (relu) z = z > 0.0f ? z : 0.0f;
    out[(long long)r * C + c] = z;
  }
}

// backward: column sums of dz and dz * xhat -> coefficients (+ dgamma, dbeta) -> dY = alpha dz + gammap y + betap and its
// column sums (the bias gradient, colsum [C]).  grid = C / 32.
__global__ __launch_bounds__(kSbT) void ml_small_bn_bwd_kernel(
    const float* __restrict__ g, const float* __restrict__ out, const float* __restrict__ y, const float* __restrict__ bn,
    int R, int C, int relu, double count, const float* __restrict__ gamma, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dy, float* __restrict__ colsum) {
  __shared__ double red[32][32][2];
  __shared__ float cf[3][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
  const float mean = bn[2 * C + c], invstd = bn[3 * C + c];
  float s1 = 0.0f, s2 = 0.0f;
  constexpr int U = 8;  // rows in flight per lane (a rolled loop waits for its three loads row by row)
  for (int r0 = rl; r0 < R; r0 += 32 * U) {
    float gv[U], ov[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + 32 * u < R ? r0 + 32 * u : rl;  // (clamped: a valid row, masked below)
      const long long o = (long long)r * C + c;
      gv[u] = g[o], ov[u] = out[o], yv[u] = y[o];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float d = (r0 + 32 * u < R && (relu == 0 || ov[u] > 0.0f)) ? gv[u] : 0.0f;
      s1 += d;
      s2 = __builtin_fmaf(d, (yv[u] - mean) * invstd, s2);
    }
  }
  red[rl][cl][0] = (double)s1;
  red[rl][cl][1] = (double)s2;
  __syncthreads();
  if (rl == 0) {
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      a += red[k][cl][0];
      b += red[k][cl][1];
    }
    const float alpha = gamma[c] * invstd;
    const float gammap = (float)(-(double)alpha * b / count * (double)invstd);
    cf[0][cl] = alpha;
    cf[1][cl] = gammap;
    cf[2][cl] = (float)(-(double)alpha * a / count - (double)gammap * (double)mean);
    dgamma[c] = (float)b;
    dbeta[c] = (float)a;
  }
  __syncthreads();
  const float alpha = cf[0][cl], gammap = cf[1][cl], betap = cf[2][cl];
  float sd = 0.0f;
  for (int r0 = rl; r0 < R; r0 += 32 * U) {
    float gv[U], ov[U], yv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + 32 * u < R ? r0 + 32 * u : rl;
      const long long o = (long long)r * C + c;
      gv[u] = g[o], ov[u] = out[o], yv[u] = y[o];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r0 + 32 * u < R) {
        const float d = (relu == 0 || ov[u] > 0.0f) ? gv[u] : 0.0f;
        const float v = __builtin_fmaf(alpha, d, __builtin_fmaf(gammap, yv[u], betap));
        dy[(long long)(r0 + 32 * u) * C + c] = v;
        sd += v;
      }
    }
  }
  __syncthreads();  // (red is reused)
  red[rl][cl][0] = (double)sd;
  __syncthreads();
  if (rl == 0) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) a += red[k][cl][0];
    colsum[c] = (float)a;
  }
}

__global__ void ml_transpose_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ wt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * cols) return;
  wt[(e % cols) * rows + e / cols] = w[e];
}

// ---- first layer of the P x P edge MLP without the pair tensor --------------------------------------------------------------
// The layer's input row (b, i, j) is [a_i ; b_j], so  x w^T = pa[(b, i)] + pb[(b, j)]  with  pa = a Wa^T + bias,
// pb = b Wb^T  (Wa | Wb = the column halves of w): two GEMMs over the B*P part rows instead of one over the B*P*P pair rows.
// ypre[(b,i,j)][c] = pa[(b,i)][c] + pb[(b,j)][c] and the per-tile column sums (sum, sum of squares; rows ascending) of
// ml_bias_stats_kernel.  grid = tiles of kRT rows, block 256.
__global__ __launch_bounds__(256) void pair_sum_stats_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int P,
                                                             int R, int C, float* __restrict__ ypre,
                                                             float* __restrict__ partial) {
  const long long r0 = (long long)blockIdx.x * kRT;
  const int rows = R - r0 < kRT ? (int)(R - r0) : kRT;
  // (sample s, part i, part j) of the tile's first row, then counted up row by row: three 64-bit divisions per row and
  // thread were most of this kernel's 20 us
  const int bi0 = (int)(r0 / P), j0 = (int)(r0 % P), s0 = bi0 / P, i0 = bi0 % P;
  for (int c = threadIdx.x; c < C; c += 256) {
    float va[kRT], vb[kRT];
    int sidx = s0, ii = i0, j = j0;
#pragma unroll
    for (int i = 0; i < kRT; ++i) {
      va[i] = pa[(long long)(sidx * P + ii) * C + c];
      vb[i] = pb[(long long)(sidx * P + j) * C + c];
      if (i + 1 < rows) {  // (rows behind the tile's end repeat the last one: never stored)
        if (++j == P) {
          j = 0;
          if (++ii == P) {
            ii = 0;
            ++sidx;
          }
        }
      }
    }
    float s = 0.0f, ss = 0.0f;
#pragma unroll
    for (int i = 0; i < kRT; ++i) {
      if (i < rows) {
        const float t = va[i] + vb[i];
        ypre[(r0 + i) * C + c] = t;
        s += t;
        ss = __builtin_fmaf(t, t, ss);
      }
    }
    float* d = partial + ((long long)blockIdx.x * C + c) * 2;
    d[0] = s;
    d[1] = ss;
  }
}

// dpa[(b,i)][c] = sum_j dy[(b,i,j)][c] (blocks [0, M)) and dpb[(b,j)][c] = sum_i dy[(b,i,j)][c] (blocks [M, 2M)), both in
// ascending order of the summed index.  grid = 2 M, block 256.
__global__ __launch_bounds__(256) void pair_reduce_kernel(const float* __restrict__ dy, int P, int M, int C,
                                                          float* __restrict__ dpa, float* __restrict__ dpb) {
  const bool second = (int)blockIdx.x >= M;
  const int m = second ? (int)blockIdx.x - M : (int)blockIdx.x, b = m / P, q = m % P;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.0f;
    mpa::batched_rows<8>(0, 1, P,
                         [&](int t) {
                           const long long row = second ? ((long long)(b * P + t) * P + q) : ((long long)(b * P + q) * P + t);
                           return dy[row * C + c];
                         },
                         [&](int, float v) { acc += v; });
    (second ? dpb : dpa)[(long long)m * C + c] = acc;
  }
}

// out[n * ldo + k] = sum over the chunks of part[chunk][n * K + k] in chunk order: a weight gradient written into a column
// block of a wider matrix.  grid = ceil(N K / 256), block 256.
__global__ __launch_bounds__(256) void ml_reduce_strided_kernel(const float* __restrict__ part, int chunks, int N, int K,
                                                                int ldo, float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x, elems = (long long)N * K;
  if (e >= elems) return;
  float acc = 0.0f;
  mpa::batched_rows<8>(0, 1, chunks, [&](int ch) { return part[(long long)ch * elems + e]; }, [&](int, float v) { acc += v; });
  out[(e / K) * ldo + e % K] = acc;
}

struct MlWs {
  int* hdr;
  unsigned* tickets;
  float *ypre, *bn, *coef, *partial, *wt, *tnpart, *dy;
  double* stage;
  int64_t total;
};

MlWs ml_carve(char* base, int64_t R, int64_t K, int64_t N) {
  MlWs w;
  char* p = base;
  auto take = [&](int64_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  const int64_t tiles = (R + kRT - 1) / kRT;
  w.hdr = reinterpret_cast<int*>(take(64));
  w.tickets = reinterpret_cast<unsigned*>(take(256));
  w.ypre = reinterpret_cast<float*>(take(4 * R * N));
  w.bn = reinterpret_cast<float*>(take(4 * 4 * N));
  w.coef = reinterpret_cast<float*>(take(4 * 3 * N));
  w.partial = reinterpret_cast<float*>(take(4 * tiles * N * 2));
  w.wt = reinterpret_cast<float*>(take(4 * K * N));
  w.tnpart = reinterpret_cast<float*>(take(4 * (int64_t)kChunks * N * K));
  w.dy = reinterpret_cast<float*>(take(4 * R * N));
  w.stage = reinterpret_cast<double*>(take(8 * 2 * N * ((tiles + kEB - 1) / kEB)));
  w.total = p - base;
  return w;
}

int ml_check(int64_t R, int64_t K, int64_t N, const char* who) {
  MPA_REQUIRE(R >= 1 && R <= (1 << 24), "%s: 1 <= rows <= 2^24", who);
  MPA_REQUIRE(K >= 64 && K % 64 == 0 && K <= 4096, "%s: input width must be a multiple of 64 (<= 4096)", who);
  MPA_REQUIRE(N >= 64 && N % 64 == 0 && N <= 4096, "%s: output width must be a multiple of 64 (<= 4096)", who);
  return MPA_OK;
}

// MPA_ML_SMALL=tiled: the row-tiled BatchNorm launches for every row count (A/B timing and the cross-check test)
bool ml_small_fused() {
  const char* e = getenv("MPA_ML_SMALL");
  return !(e != nullptr && e[0] == 't');
}

template <typename Kern, typename... Args>
void launch(Kern kern, dim3 grid, dim3 block, hipStream_t s, Args... args) {
  hipLaunchKernelGGL(kern, grid, block, 0, s, args...);
}

#if DG_GEMM_SPLIT
// C = A W^T with the output pass of `EPI` (dg_gemm_split.h); WT: W is given as [K][Nout].  The row count travels by value.
template <int EPI, bool WT>
void ml_gemm(const float* A, int lda, const float* W, int K, float* C, int ldc, int Nout, int64_t R, GsEpi epi, int ldwt,
             hipStream_t s) {
  const unsigned gx = DG_GEMM_GRID_X(R);
  epi.rows = (int)R;
  if (Nout % 128 == 0)
    launch(gemm_nt_split_kernel<128, false, EPI, WT>, dim3(gx, Nout / 128), dim3(kGsT), s, A, lda, W, K, C, ldc,
           (const int*)nullptr, epi, ldwt);
  else
    launch(gemm_nt_split_kernel<64, false, EPI, WT>, dim3(gx, Nout / 64), dim3(kGsT), s, A, lda, W, K, C, ldc,
           (const int*)nullptr, epi, ldwt);
}
#else
void ml_gemm_nt(const float* A, int lda, const float* W, int K, float* C, int ldc, int Nout, int64_t R, const int* hdr,
                hipStream_t s) {
  const unsigned gx = DG_GEMM_GRID_X(R);
  if (Nout % 128 == 0) launch(DG_NT_KERNEL<128, false>, dim3(gx, Nout / 128), dim3(DG_GEMM_THREADS), s, A, lda, W, K, C, ldc, hdr);
  else launch(DG_NT_KERNEL<64, false>, dim3(gx, Nout / 64), dim3(DG_GEMM_THREADS), s, A, lda, W, K, C, ldc, hdr);
}
#endif

}  // namespace

extern "C" int mpa_mlp_layer_workspace(int64_t R, int64_t K, int64_t N, int64_t* bytes) {
  if (int st = ml_check(R, K, N, "mlp_layer_workspace")) return st;
  MPA_REQUIRE(bytes != nullptr, "mlp_layer_workspace: null pointer");
  *bytes = ml_carve(nullptr, R, K, N).total;
  return MPA_OK;
}

extern "C" int mpa_mlp_layer_forward(const float* x, int64_t ldx, const float* w, const float* bias, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, int training,
                                     float momentum, float eps, int relu, int64_t R, int64_t K, int64_t N, void* ws,
                                     float* out, void* stream) {
  if (int st = ml_check(R, K, N, "mlp_layer_forward")) return st;
  MPA_REQUIRE(x && w && ws && out && ldx >= K && ldx % 4 == 0, "mlp_layer_forward: bad pointer / leading dimension");
  MPA_REQUIRE(gamma == nullptr || (beta && running_mean && running_var), "mlp_layer_forward: incomplete BatchNorm");
  MPA_REQUIRE((uintptr_t)ws % 256 == 0, "mlp_layer_forward: workspace must be 256-byte aligned");
  hipStream_t s = mpa::as_stream(stream);
  const MlWs m = ml_carve(static_cast<char*>(ws), R, K, N);
  const int tiles = (int)((R + kRT - 1) / kRT);
  const long long total4 = R * N / 4;
  const CoopWs cw{m.stage, m.tickets};
  if (R <= kSmallRows) {
    tfg::GemmArgs g{};
    g.A = x;
    g.lda = (int)ldx;
    g.W = w;
    g.bias = bias;
    g.M = (int)R;
    g.N = (int)N;
    g.K = (int)K;
    g.zero = m.tickets;
    if (gamma == nullptr) {
      g.C = out;
      g.relu = relu;
      tfg::launch_gemm<tfg::EPI_BIAS_ACT>(g, s);
      return mpa::check_launch("mlp_layer_forward");
    }
    g.C = m.ypre;
    if (training) {
      const int tiles32 = (int)((R + 31) / 32);
      g.stats = m.partial;
      tfg::launch_gemm<tfg::EPI_STATS>(g, s);
      if (ml_small_fused()) {  // statistics and activation in one launch
        launch(ml_small_bn_apply_kernel, dim3((unsigned)(N / 32), (unsigned)((R + 255) / 256)), dim3(kSbT), s,
               (const float*)m.partial, tiles32, (const float*)m.ypre, (int)R, (int)N, (double)R, gamma, beta, running_mean,
               running_var, momentum, eps, relu, m.bn, out);
        return mpa::check_launch("mlp_layer_forward");
      }
      launch(ml_bn_finalize_kernel, dim3((unsigned)(N / 64), (unsigned)((tiles32 + kEB - 1) / kEB)), dim3(64 * kSlices), s,
             (const float*)m.partial, tiles32, (int)N, (double)R, gamma, beta, running_mean, running_var, momentum, eps,
             m.bn, cw);
    } else {
      tfg::launch_gemm<tfg::EPI_BIAS_ACT>(g, s);  // (g.relu = 0: bias only)
      launch(ml_bn_from_running_kernel, dim3((unsigned)(N / 64)), dim3(64), s, (int)N, gamma, beta,
             (const float*)running_mean, (const float*)running_var, eps, m.bn);
    }
    launch(ml_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), s, (const float*)m.ypre,
           (const float*)m.bn, (const float*)nullptr, total4, (int)N, relu, out);
    return mpa::check_launch("mlp_layer_forward");
  }
#if DG_GEMM_SPLIT
  // bias, BatchNorm statistics / activation in the GEMM's output pass: 3 launches with BatchNorm, 1 without
  GsEpi epi;
  epi.bias = bias;
  epi.zero = m.tickets;
  if (gamma == nullptr) {
    epi.relu = relu;
    ml_gemm<2, false>(x, (int)ldx, w, (int)K, out, (int)N, (int)N, R, epi, 0, s);
    return mpa::check_launch("mlp_layer_forward");
  }
  if (training) {
    const int tiles128 = (int)((R + 127) / 128);
    epi.stats = m.partial;
    epi.ldstats = (int)N;
    ml_gemm<1, false>(x, (int)ldx, w, (int)K, m.ypre, (int)N, (int)N, R, epi, 0, s);
    launch(ml_bn_finalize_kernel, dim3((unsigned)(N / 64), (unsigned)((tiles128 + kEB - 1) / kEB)), dim3(64 * kSlices), s,
           (const float*)m.partial, tiles128, (int)N, (double)R, gamma, beta, running_mean, running_var, momentum, eps, m.bn,
           cw);
  } else {
    ml_gemm<2, false>(x, (int)ldx, w, (int)K, m.ypre, (int)N, (int)N, R, epi, 0, s);
    launch(ml_bn_from_running_kernel, dim3((unsigned)(N / 64)), dim3(64), s, (int)N, gamma, beta,
           (const float*)running_mean, (const float*)running_var, eps, m.bn);
  }
  (void)tiles;
#else
  launch(ml_set_hdr_kernel, dim3(1), dim3(64), s, m.hdr, (int)R, m.tickets);
  if (gamma == nullptr) {
    ml_gemm_nt(x, (int)ldx, w, (int)K, out, (int)N, (int)N, R, m.hdr, s);
    launch(ml_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), s, (const float*)out, (const float*)nullptr,
           bias, total4, (int)N, relu, out);
    return mpa::check_launch("mlp_layer_forward");
  }
  ml_gemm_nt(x, (int)ldx, w, (int)K, m.ypre, (int)N, (int)N, R, m.hdr, s);
  if (training) {
    launch(ml_bias_stats_kernel, dim3((unsigned)tiles), dim3(256), s, m.ypre, bias, (int)R, (int)N, m.partial);
    launch(ml_bn_finalize_kernel, dim3((unsigned)(N / 64), (unsigned)((tiles + kEB - 1) / kEB)), dim3(64 * kSlices), s,
           (const float*)m.partial, tiles, (int)N, (double)R, gamma, beta, running_mean, running_var, momentum, eps, m.bn,
           cw);
  } else {
    if (bias != nullptr)
      launch(ml_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), s, (const float*)m.ypre,
             (const float*)nullptr, bias, total4, (int)N, 0, m.ypre);
    launch(ml_bn_from_running_kernel, dim3((unsigned)(N / 64)), dim3(64), s, (int)N, gamma, beta,
           (const float*)running_mean, (const float*)running_var, eps, m.bn);
  }
#endif
  launch(ml_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), s, (const float*)m.ypre, (const float*)m.bn,
         (const float*)nullptr, total4, (int)N, relu, out);
  return mpa::check_launch("mlp_layer_forward");
}

extern "C" int mpa_mlp_layer_backward(const float* grad_out, const float* x, int64_t ldx, const float* w,
                                      const float* gamma, const float* out, int relu, int64_t R, int64_t K, int64_t N,
                                      void* ws, float* grad_x, float* grad_w, float* grad_b, float* grad_gamma,
                                      float* grad_beta, void* stream) {
  if (int st = ml_check(R, K, N, "mlp_layer_backward")) return st;
  MPA_REQUIRE(grad_out && x && w && out && ws && grad_w, "mlp_layer_backward: null pointer");
  MPA_REQUIRE(gamma == nullptr || (grad_gamma && grad_beta), "mlp_layer_backward: BatchNorm gradients missing");
  hipStream_t s = mpa::as_stream(stream);
  const MlWs m = ml_carve(static_cast<char*>(ws), R, K, N);
  const int tiles = (int)((R + kRT - 1) / kRT);
  const CoopWs cw{m.stage, m.tickets};
  int bias_tiles = tiles;  // rows of the column-sum table the bias gradient is reduced from
  if (gamma != nullptr && R <= kSmallRows && ml_small_fused()) {  // sums, coefficients, dY and its column sums: one launch
    launch(ml_small_bn_bwd_kernel, dim3((unsigned)(N / 32)), dim3(kSbT), s, grad_out, out, (const float*)m.ypre,
           (const float*)m.bn, (int)R, (int)N, relu, (double)R, gamma, grad_gamma, grad_beta, m.dy, m.partial);
    bias_tiles = 1;
  } else {
    if (gamma != nullptr) {
      launch(ml_bwd_sums_kernel, dim3((unsigned)tiles), dim3(256), s, grad_out, out, (const float*)m.ypre,
             (const float*)m.bn, (int)R, (int)N, relu, m.partial);
      launch(ml_bwd_coef_kernel, dim3((unsigned)(N / 64), (unsigned)((tiles + kEB - 1) / kEB)), dim3(64 * kSlices), s,
             (const float*)m.partial, tiles, (int)N, (double)R, gamma, (const float*)m.bn, m.coef, grad_gamma, grad_beta, cw);
    }
    launch(ml_bwd_dy_kernel, dim3((unsigned)tiles), dim3(256), s, grad_out, out, (const float*)m.ypre,
           gamma != nullptr ? (const float*)m.coef : (const float*)nullptr, (int)R, (int)N, relu, m.dy, m.partial);
  }
  // dW [N][K] = dY^T X (row chunks, then a fixed-order sum) and db = the column sums of dY's per-tile table; few rows:
  // one launch of the transformer's weight-gradient kernel (one 32 x 32 tile per block, the block's waves split the rows)
  if (R <= kSmallRows && ml_small_fused()) {
    mpa::launch_small_wgrad(m.dy, x, (int)ldx, grad_w, grad_b, (int)R, (int)N, (int)K, s);
  } else {
    // enough chunks for ~512 blocks in the launch, each at least 64 rows
    const int out_tiles = (int)(((N + 127) / 128) * (K % 128 == 0 ? K / 128 : K / 64));
    int chunks = (512 + out_tiles - 1) / out_tiles;
    if (chunks > (int)(R / 64)) chunks = (int)(R / 64);
    chunks = chunks < 1 ? 1 : (chunks > kChunks ? kChunks : chunks);
    if (chunks >= 8) chunks = chunks / 8 * 8;  // (a multiple of 8 gets the kernel's XCD-aware chunk mapping)
    const int rows_per_chunk = (int)(((R + chunks - 1) / chunks + 31) / 32 * 32);
    const dim3 grid((unsigned)((N + 127) / 128), (unsigned)(K % 128 == 0 ? K / 128 : K / 64), (unsigned)chunks);
    // the row count travels by value in both builds (both kernels take `rows` when hdr is null): the forward's
    // small-row path never writes the header
    const int* hdr = nullptr;
    if (K % 128 == 0)
      launch(DG_TN_KERNEL<128>, grid, dim3(DG_GEMM_THREADS), s, (const float*)m.dy, (int)N, (int)N, x, (int)ldx, (int)K, m.tnpart,
             rows_per_chunk, hdr, (int)R);
    else
      launch(DG_TN_KERNEL<64>, grid, dim3(DG_GEMM_THREADS), s, (const float*)m.dy, (int)N, (int)N, x, (int)ldx, (int)K, m.tnpart,
             rows_per_chunk, hdr, (int)R);
    const long long elems = (long long)N * K;
    if (grad_b != nullptr) {
      const int blocks_w = (int)((elems + 31) / 32);
      launch(dg::gemm_tn_reduce2_kernel, dim3((unsigned)(blocks_w + (N + 31) / 32)), dim3(256), s, (const float*)m.tnpart,
             chunks, elems, grad_w, blocks_w, (const float*)m.partial, bias_tiles, (long long)N, grad_b);
    } else {
      dg::launch_tn_reduce(m.tnpart, chunks, elems, grad_w, s);
    }
  }
  if (grad_x != nullptr && R <= kSmallRows) {  // dX = dY . W with W [N][K] read as the transposed operand [k = N][n = K]
    tfg::GemmArgs g{};
    g.A = m.dy;
    g.W = w;
    g.C = grad_x;
    g.M = (int)R;
    g.N = (int)K;
    g.K = (int)N;
    tfg::launch_gemm<tfg::EPI_NONE, true>(g, s);
  } else if (grad_x != nullptr) {  // dX [R][K] = dY [R][N] . W [N][K]
#if DG_GEMM_SPLIT
    ml_gemm<0, true>(m.dy, (int)N, w, (int)N, grad_x, (int)K, (int)K, R, GsEpi{}, (int)K, s);
#else
    launch(ml_transpose_kernel, dim3((unsigned)((N * K + 255) / 256)), dim3(256), s, w, (int)N, (int)K, m.wt);
    ml_gemm_nt(m.dy, (int)N, m.wt, (int)N, grad_x, (int)K, (int)K, R, m.hdr, s);
#endif
  }
  return mpa::check_launch("mlp_layer_backward");
}

// ---- first layer of the P x P edge MLP on (a_i, b_j) pairs ------------------------------------------------------------------
namespace {

struct PairWs {
  MlWs m;                       // the layer's workspace over the R = B P P pair rows (K = 2 F)
  float *pa, *pb, *dpa, *dpb;   // [B P, N] each
  int64_t total;
};

PairWs pair_carve(char* base, int64_t B, int64_t P, int64_t F, int64_t N) {
  PairWs w;
  w.m = ml_carve(base, B * P * P, 2 * F, N);
  char* p = base + w.m.total;
  auto take = [&](int64_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  const int64_t mn = 4 * B * P * N;
  w.pa = reinterpret_cast<float*>(take(mn));
  w.pb = reinterpret_cast<float*>(take(mn));
  w.dpa = reinterpret_cast<float*>(take(mn));
  w.dpb = reinterpret_cast<float*>(take(mn));
  w.total = p - base;
  return w;
}

int pair_check(int64_t B, int64_t P, int64_t F, int64_t N, const char* who) {
  MPA_REQUIRE(B >= 1 && P >= 1 && P <= 1024 && B * P <= (1 << 20), "%s: 1 <= B, 1 <= P <= 1024, B P <= 2^20", who);
  MPA_REQUIRE(F >= 64 && F % 64 == 0 && F <= 2048, "%s: feature width must be a multiple of 64 (<= 2048)", who);
  return ml_check(B * P * P, 2 * F, N, who);
}

}  // namespace

extern "C" int mpa_pair_layer_workspace(int64_t B, int64_t P, int64_t F, int64_t N, int64_t* bytes) {
  if (int st = pair_check(B, P, F, N, "pair_layer_workspace")) return st;
  MPA_REQUIRE(bytes != nullptr, "pair_layer_workspace: null pointer");
  *bytes = pair_carve(nullptr, B, P, F, N).total;
  return MPA_OK;
}

extern "C" int mpa_pair_layer_forward(const float* a, const float* b, const float* w, const float* bias, const float* gamma,
                                      const float* beta, float* running_mean, float* running_var, int training,
                                      float momentum, float eps, int relu, int64_t B, int64_t P, int64_t F, int64_t N,
                                      void* ws, float* out, void* stream) {
  if (int st = pair_check(B, P, F, N, "pair_layer_forward")) return st;
  MPA_REQUIRE(a && b && w && gamma && beta && running_mean && running_var && ws && out,
              "pair_layer_forward: null pointer (the layer has a BatchNorm)");
  MPA_REQUIRE((uintptr_t)ws % 256 == 0, "pair_layer_forward: workspace must be 256-byte aligned");
  hipStream_t s = mpa::as_stream(stream);
  const PairWs pw = pair_carve(static_cast<char*>(ws), B, P, F, N);
  const MlWs& m = pw.m;
  const int64_t M = B * P, R = M * P;
  const int tiles = (int)((R + kRT - 1) / kRT);
  const long long total4 = R * N / 4;
  const CoopWs cw{m.stage, m.tickets};
  tfg::GemmArgs g{};
  g.A = a;
  g.W = w;
  g.ldw = (int)(2 * F);
  g.bias = bias;
  g.C = pw.pa;
  g.M = (int)M;
  g.N = (int)N;
  g.K = (int)F;
  g.zero = m.tickets;
  tfg::launch_gemm<tfg::EPI_BIAS_ACT>(g, s);  // pa = a Wa^T + bias   (g.relu = 0)
  g.A = b;
  g.W = w + F;
  g.bias = nullptr;
  g.C = pw.pb;
  g.zero = nullptr;
  tfg::launch_gemm<tfg::EPI_BIAS_ACT>(g, s);  // pb = b Wb^T
  launch(pair_sum_stats_kernel, dim3((unsigned)tiles), dim3(256), s, (const float*)pw.pa, (const float*)pw.pb, (int)P, (int)R,
         (int)N, m.ypre, m.partial);
  if (training)
    launch(ml_bn_finalize_kernel, dim3((unsigned)(N / 64), (unsigned)((tiles + kEB - 1) / kEB)), dim3(64 * kSlices), s,
           (const float*)m.partial, tiles, (int)N, (double)R, gamma, beta, running_mean, running_var, momentum, eps, m.bn, cw);
  else
    launch(ml_bn_from_running_kernel, dim3((unsigned)(N / 64)), dim3(64), s, (int)N, gamma, beta, (const float*)running_mean,
           (const float*)running_var, eps, m.bn);
  launch(ml_apply_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), s, (const float*)m.ypre, (const float*)m.bn,
         (const float*)nullptr, total4, (int)N, relu, out);
  return mpa::check_launch("pair_layer_forward");
}

extern "C" int mpa_pair_layer_backward(const float* grad_out, const float* a, const float* b, const float* w,
                                       const float* gamma, const float* out, int relu, int64_t B, int64_t P, int64_t F,
                                       int64_t N, void* ws, float* grad_a, float* grad_b, float* grad_w, float* grad_bias,
                                       float* grad_gamma, float* grad_beta, void* stream) {
  if (int st = pair_check(B, P, F, N, "pair_layer_backward")) return st;
  MPA_REQUIRE(grad_out && a && b && w && gamma && out && ws && grad_w && grad_gamma && grad_beta,
              "pair_layer_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const PairWs pw = pair_carve(static_cast<char*>(ws), B, P, F, N);
  const MlWs& m = pw.m;
  const int64_t M = B * P, R = M * P;
  const int tiles = (int)((R + kRT - 1) / kRT);
  const CoopWs cw{m.stage, m.tickets};
  launch(ml_bwd_sums_kernel, dim3((unsigned)tiles), dim3(256), s, grad_out, out, (const float*)m.ypre, (const float*)m.bn,
         (int)R, (int)N, relu, m.partial);
  launch(ml_bwd_coef_kernel, dim3((unsigned)(N / 64), (unsigned)((tiles + kEB - 1) / kEB)), dim3(64 * kSlices), s,
         (const float*)m.partial, tiles, (int)N, (double)R, gamma, (const float*)m.bn, m.coef, grad_gamma, grad_beta, cw);
  launch(ml_bwd_dy_kernel, dim3((unsigned)tiles), dim3(256), s, grad_out, out, (const float*)m.ypre, (const float*)m.coef,
         (int)R, (int)N, relu, m.dy, m.partial);
  launch(pair_reduce_kernel, dim3((unsigned)(2 * M)), dim3(256), s, (const float*)m.dy, (int)P, (int)M, (int)N, pw.dpa,
         pw.dpb);
  if (grad_bias != nullptr) dg::launch_tn_reduce(m.partial, tiles, (long long)N, grad_bias, s);  // column sums of dY
  // dW[:, 0:F] = dpa^T a,  dW[:, F:2F] = dpb^T b  (row chunks, fixed-order sums into the column halves of grad_w)
  {
    const int out_tiles = (int)(((N + 127) / 128) * (F % 128 == 0 ? F / 128 : F / 64));
    int chunks = (512 + out_tiles - 1) / out_tiles;
    if (chunks > (int)(M / 64)) chunks = (int)(M / 64);
    chunks = chunks < 1 ? 1 : (chunks > kChunks ? kChunks : chunks);
    if (chunks >= 8) chunks = chunks / 8 * 8;
    const int rows_per_chunk = (int)(((M + chunks - 1) / chunks + 31) / 32 * 32);
    const dim3 grid((unsigned)((N + 127) / 128), (unsigned)(F % 128 == 0 ? F / 128 : F / 64), (unsigned)chunks);
    for (int half = 0; half < 2; ++half) {
      const float* dY = half ? pw.dpb : pw.dpa;
      const float* X = half ? b : a;
      if (F % 128 == 0)
        launch(gemm_tn_split_kernel<128>, grid, dim3(kGsT), s, dY, (int)N, (int)N, X, (int)F, (int)F, m.tnpart, rows_per_chunk,
               (const int*)nullptr, (int)M);
      else
        launch(gemm_tn_split_kernel<64>, grid, dim3(kGsT), s, dY, (int)N, (int)N, X, (int)F, (int)F, m.tnpart, rows_per_chunk,
               (const int*)nullptr, (int)M);
      launch(ml_reduce_strided_kernel, dim3((unsigned)((N * F + 255) / 256)), dim3(256), s, (const float*)m.tnpart, chunks,
             (int)N, (int)F, (int)(2 * F), grad_w + half * F);
    }
  }
  for (int half = 0; half < 2; ++half) {  // d a = dpa Wa,  d b = dpb Wb: the weight's column half read as the transposed operand
    float* dst = half ? grad_b : grad_a;
    if (dst == nullptr) continue;
    tfg::GemmArgs g{};
    g.A = half ? pw.dpb : pw.dpa;
    g.W = w + half * F;
    g.ldw = (int)(2 * F);
    g.C = dst;
    g.M = (int)M;
    g.N = (int)F;
    g.K = (int)N;
    if (half == 1 && grad_b == grad_a) {  // one tensor in both roles: its gradient is the sum (added in the second GEMM's
      g.resid = grad_a;                   // output pass: every element is read and written by the same thread)
      tfg::launch_gemm<tfg::EPI_DROP_RESID, true>(g, s);  // (drop.p = 0: resid + value)
    } else {
      tfg::launch_gemm<tfg::EPI_NONE, true>(g, s);
    }
  }
  return mpa::check_launch("pair_layer_backward");
}
