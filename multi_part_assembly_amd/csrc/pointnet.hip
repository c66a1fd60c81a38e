// PointNet part encoder — forward and backward (training-mode BatchNorm) for gfx950.
//
// Replaces the torch module of the reference (multi_part_assembly/models/modules/encoder/pointnet.py:6-41):
// 5 x [1x1 Conv1d (no bias) -> BatchNorm1d -> ReLU (none after the last)] 3-64-64-64-128-F, max over the
// N points of each part; and the compaction around it (models/pn_transformer/network.py:59-68): the
// kernels take ALL B*P part slots plus the validity mask and simply skip padded parts, so launch
// shapes are static and no device->host sync is needed.
//
// Design ("scalar-weight FMA panels", the same scalar-cache idea as chamfer_core.h):
//   * fp32 throughout (parity bar 1e-4; gfx950's fp32 MFMA has the same peak as the fp32 VALU).
//   * activations are point-major  [row = part*N + point][channel]  so one LANE owns one point (row):
//     its 64 output channels live in 64 accumulator VGPRs, the weight W[k][c0..c0+63] of the current
//     input channel k is wave-uniform and arrives through the scalar cache in 64 SGPRs, and the inner
//     loop is 64 independent v_fmac_f32 with a scalar operand — no LDS traffic and no cross-lane
//     shuffles in the main loop.
//   * per 64x64 output panel the wave transposes through LDS once: that gives coalesced row stores
//     AND the per-channel column sums BatchNorm needs (sum, sum of squares) in the same pass.
//   * BatchNorm+ReLU of layer l-1 is applied on the fly when layer l loads its input (the activated
//     values are also written out, they are the scalar operand of the weight-gradient kernel).
//   * backward: dY = alpha*dZ + gamma'*Y + beta' per channel (BatchNorm backward written as an affine
//     map, coefficients from two column sums), input-gradient = the same panel kernel with the
//     untransposed weights, weight-gradient = lane-per-output-channel panels with the activated
//     input as the scalar operand, summed over parts by a deterministic second stage.  No atomics.
#include "common.h"

namespace {

constexpr int kT = 256;           // threads per block (4 waves)
constexpr int kRows = 256;        // rows (points) per block: one row per lane
constexpr int kPanel = 64;        // output channels per accumulator panel
constexpr int kKB = 16;           // input channels per unrolled k-block

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// bnp layout per channel: {scale, shift, mean, invstd}
struct BnP {
  float scale, shift, mean, invstd;
};

// ---- K0: transpose the conv weights, count the valid rows ----------------------------------------------
__global__ void pn_prep_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int cin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout * cin) {
    const int co = i / cin, ci = i % cin;
    wt[ci * cout + co] = w[i];
  }
}

__global__ void pn_count_kernel(const float* __restrict__ valids, int M, int N, float* __restrict__ count) {
  float s = 0.0f;
  for (int m = threadIdx.x; m < M; m += 64) s += valids[m] != 0.0f ? 1.0f : 0.0f;
  s = wave_sum(s);
  if (threadIdx.x == 0) count[0] = s * (float)N;
}

// ---- K1: forward layer ----------------------------------------------------------------------------------
// in: CIN == 3 ? points [M*N][3] : pre-BN output of the previous layer [M*N][CIN]
// grid = (ceil(N/256), M), block 256.  partial [M*tiles][cout][2].
template <int CIN>
__global__ __launch_bounds__(kT) void pn_fwd_layer_kernel(
    const float* __restrict__ in, const BnP* __restrict__ bnp_prev, float* __restrict__ a_out,
    const float* __restrict__ wt, int cout, const float* __restrict__ valids, int N,
    float* __restrict__ y_out, float* __restrict__ partial) {
  __shared__ float tile[kT / 64][64][kPanel + 1];
  __shared__ float red[kT / 64][kPanel][2];
  const int m = blockIdx.y;
  if (valids[m] == 0.0f) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = blockIdx.x * kRows + wave * 64;  // first row of this wave's tile
  const int n = n0 + lane;
  const bool rowok = n < N;
  const long long row = (long long)m * N + (rowok ? n : N - 1);
  const int rows_here = N - n0 < 64 ? (N - n0 < 0 ? 0 : N - n0) : 64;
  const int blk = m * gridDim.x + blockIdx.x;

  for (int c0 = 0; c0 < cout; c0 += kPanel) {
    float acc[kPanel];
#pragma unroll
    for (int c = 0; c < kPanel; ++c) acc[c] = 0.0f;
    if constexpr (CIN == 3) {
      const float a0 = in[row * 3 + 0], a1 = in[row * 3 + 1], a2 = in[row * 3 + 2];
#pragma unroll
      for (int c = 0; c < kPanel; ++c) {
        acc[c] = __builtin_fmaf(a0, wt[0 * cout + c0 + c], acc[c]);
        acc[c] = __builtin_fmaf(a1, wt[1 * cout + c0 + c], acc[c]);
        acc[c] = __builtin_fmaf(a2, wt[2 * cout + c0 + c], acc[c]);
      }
    } else {
      for (int kb = 0; kb < CIN; kb += kKB) {
        float a[kKB];
        const float4* src = reinterpret_cast<const float4*>(in + row * CIN + kb);
#pragma unroll
        for (int v = 0; v < kKB / 4; ++v) {
          const float4 y = src[v];
          a[4 * v + 0] = y.x;
          a[4 * v + 1] = y.y;
          a[4 * v + 2] = y.z;
          a[4 * v + 3] = y.w;
        }
#pragma unroll
        for (int k = 0; k < kKB; ++k) {  // BatchNorm + ReLU of the previous layer, per input channel
          const BnP p = bnp_prev[kb + k];
          a[k] = __builtin_fmaxf(__builtin_fmaf(a[k], p.scale, p.shift), 0.0f);
        }
        if (c0 == 0 && rowok) {  // side output: activated input (scalar operand of the weight gradient)
          float4* dst = reinterpret_cast<float4*>(a_out + row * CIN + kb);
#pragma unroll
          for (int v = 0; v < kKB / 4; ++v) dst[v] = make_float4(a[4 * v], a[4 * v + 1], a[4 * v + 2], a[4 * v + 3]);
        }
#pragma unroll
        for (int k = 0; k < kKB; ++k) {
          const float* wk = wt + (long long)(kb + k) * cout + c0;  // wave-uniform -> SGPRs
#pragma unroll
          for (int c = 0; c < kPanel; ++c) acc[c] = __builtin_fmaf(a[k], wk[c], acc[c]);
        }
      }
    }
    // panel epilogue: transpose through LDS -> coalesced stores + per-channel column sums
#pragma unroll
    for (int c = 0; c < kPanel; ++c) tile[wave][lane][c] = acc[c];
    __builtin_amdgcn_wave_barrier();
    float s = 0.0f, ss = 0.0f;
    float* dst = y_out + ((long long)m * N + n0) * cout + c0 + lane;
    for (int i = 0; i < rows_here; ++i) {
      const float v = tile[wave][i][lane];
      dst[(long long)i * cout] = v;
      s += v;
      ss = __builtin_fmaf(v, v, ss);
    }
    red[wave][lane][0] = s;
    red[wave][lane][1] = ss;
    __syncthreads();
    if (threadIdx.x < kPanel) {
      float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
      for (int w = 0; w < kT / 64; ++w) {
        t0 += red[w][threadIdx.x][0];
        t1 += red[w][threadIdx.x][1];
      }
      partial[((long long)blk * cout + c0 + threadIdx.x) * 2 + 0] = t0;
      partial[((long long)blk * cout + c0 + threadIdx.x) * 2 + 1] = t1;
    }
    __syncthreads();
  }
}

// ---- K2: BatchNorm statistics -> scale/shift (+ running statistics) ---------------------------------------
// grid = ceil(C/64) blocks of 64 threads.
__global__ void pn_bn_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ valids,
                                      int M, int tiles, int C, const float* __restrict__ count,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                      float momentum, float eps, BnP* __restrict__ bnp) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, ss = 0.0;
  for (int m = 0; m < M; ++m) {
    if (valids[m] == 0.0f) continue;
    for (int t = 0; t < tiles; ++t) {
      const long long o = (((long long)m * tiles + t) * C + c) * 2;
      s += (double)partial[o];
      ss += (double)partial[o + 1];
    }
  }
  const double n = (double)count[0];
  const double mean = s / n;
  double var = ss / n - mean * mean;  // biased, as BatchNorm normalises with
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / __builtin_sqrt(var + (double)eps));
  BnP p;
  p.mean = (float)mean;
  p.invstd = invstd;
  p.scale = gamma[c] * invstd;
  p.shift = beta[c] - (float)mean * p.scale;
  bnp[c] = p;
  if (running_mean != nullptr) {  // running_var uses the unbiased estimate
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// eval mode: scale/shift from the running statistics
__global__ void pn_bn_from_running_kernel(int C, const float* __restrict__ gamma,
                                          const float* __restrict__ beta,
                                          const float* __restrict__ running_mean,
                                          const float* __restrict__ running_var, float eps,
                                          BnP* __restrict__ bnp) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  BnP p;
  p.mean = running_mean[c];
  p.invstd = 1.0f / __builtin_sqrtf(running_var[c] + eps);
  p.scale = gamma[c] * p.invstd;
  p.shift = beta[c] - p.mean * p.scale;
  bnp[c] = p;
}

// ---- K3: BatchNorm of the last layer + max over the points of each part -----------------------------------
// grid = (F/64, M), block 256: wave w scans rows n = w, w+4, ...; lane = channel.
__global__ __launch_bounds__(kT) void pn_maxpool_kernel(const float* __restrict__ y5,
                                                        const BnP* __restrict__ bnp,
                                                        const float* __restrict__ valids, int N, int F,
                                                        float* __restrict__ feat, int* __restrict__ argmax) {
  __shared__ float smv[kT / 64][64];
  __shared__ int smi[kT / 64][64];
  const int m = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  if (valids[m] == 0.0f) {
    if (wave == 0) {
      feat[(long long)m * F + c] = 0.0f;  // padded slots hold zeros (network.py:66)
      argmax[(long long)m * F + c] = -1;
    }
    return;
  }
  const BnP p = bnp[c];
  float best = -__builtin_inff();
  int arg = -1;
  const float* src = y5 + (long long)m * N * F + c;
  for (int n = wave; n < N; n += kT / 64) {
    const float z = __builtin_fmaf(src[(long long)n * F], p.scale, p.shift);
    if (z > best) {
      best = z;
      arg = n;
    }
  }
  smv[wave][threadIdx.x & 63] = best;
  smi[wave][threadIdx.x & 63] = arg;
  __syncthreads();
  if (wave == 0) {
    const int l = threadIdx.x;
#pragma unroll
    for (int w = 1; w < kT / 64; ++w) {
      const float v = smv[w][l];
      const int i = smi[w][l];
      if (v > best || (v == best && i >= 0 && i < arg)) {
        best = v;
        arg = i;
      }
    }
    feat[(long long)m * F + c] = best;
    argmax[(long long)m * F + c] = arg;
  }
}

// ---- backward -----------------------------------------------------------------------------------------------
// coef layout per channel: {alpha, gammap, betap, pad}:  dY = alpha*dZ + gammap*Y + betap
struct Coef {
  float alpha, gammap, betap, pad;
};

__device__ __forceinline__ Coef make_coef(float gamma, const BnP p, double s1, double s2, double n) {
  Coef k;
  k.alpha = gamma * p.invstd;
  k.gammap = (float)(-(double)k.alpha * s2 / n * (double)p.invstd);
  k.betap = (float)(-(double)k.alpha * s1 / n - (double)k.gammap * (double)p.mean);
  k.pad = 0.0f;
  return k;
}

// K4: layer-5 coefficients.  dZ5 is sparse: grad_feat[m,c] at row argmax[m,c].  One block per 64 channels.
__global__ void pn_bwd_top_kernel(const float* __restrict__ gfeat, const int* __restrict__ argmax,
                                  const float* __restrict__ y5, const float* __restrict__ valids, int M,
                                  int N, int F, const float* __restrict__ count,
                                  const float* __restrict__ gamma, const BnP* __restrict__ bnp,
                                  Coef* __restrict__ coef, float* __restrict__ dgamma,
                                  float* __restrict__ dbeta) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= F) return;
  const BnP p = bnp[c];
  double s1 = 0.0, s2 = 0.0;
  for (int m = 0; m < M; ++m) {
    if (valids[m] == 0.0f) continue;
    const float g = gfeat[(long long)m * F + c];
    const int n = argmax[(long long)m * F + c];
    if (n < 0) continue;  // all-NaN column: no arg-max, no gradient
    const float y = y5[((long long)m * N + n) * F + c];
    s1 += (double)g;
    s2 += (double)g * (double)((y - p.mean) * p.invstd);
  }
  coef[c] = make_coef(gamma[c], p, s1, s2, (double)count[0]);
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
}

// K7: coefficients of layer l from the (s1, s2) partials written by the input-gradient kernel of layer l+1.
__global__ void pn_bwd_coef_kernel(const float* __restrict__ partial, const float* __restrict__ valids,
                                   int M, int tiles, int C, const float* __restrict__ count,
                                   const float* __restrict__ gamma, const BnP* __restrict__ bnp,
                                   Coef* __restrict__ coef, float* __restrict__ dgamma,
                                   float* __restrict__ dbeta) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int m = 0; m < M; ++m) {
    if (valids[m] == 0.0f) continue;
    for (int t = 0; t < tiles; ++t) {
      const long long o = (((long long)m * tiles + t) * C + c) * 2;
      s1 += (double)partial[o];
      s2 += (double)partial[o + 1];
    }
  }
  coef[c] = make_coef(gamma[c], bnp[c], s1, s2, (double)count[0]);
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
}

// K6: input gradient of layer l (COUT channels -> CIN channels), fused with the ReLU mask and the
// BatchNorm-backward column sums of layer l-1.
//   dY_l[r,co] = alpha*dZ_l + gammap*Y_l + betap   (TOP: dZ_l[r,co] = gfeat[m,co] iff argmax[m,co] == n)
//   dA[r,ci]   = sum_co dY_l[r,co] * W[co][ci]
//   dZ_{l-1}   = dA where BN_{l-1}(Y_{l-1}) > 0 else 0;  s1 += dZ, s2 += dZ * xhat_{l-1}
// grid = (ceil(N/256), M), block 256; lane = row.
template <bool TOP>
__global__ __launch_bounds__(kT) void pn_dgrad_kernel(
    const float* __restrict__ y, const float* __restrict__ dz, const float* __restrict__ gfeat,
    const int* __restrict__ argmax, const Coef* __restrict__ coef, const float* __restrict__ w, int cout,
    int cin, const float* __restrict__ y_prev, const BnP* __restrict__ bnp_prev,
    const float* __restrict__ valids, int N, float* __restrict__ dz_prev, float* __restrict__ partial) {
  __shared__ float tile[kT / 64][64][kPanel + 1];
  __shared__ float red[kT / 64][kPanel][2];
  const int m = blockIdx.y;
  if (valids[m] == 0.0f) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = blockIdx.x * kRows + wave * 64;
  const int n = n0 + lane;
  const bool rowok = n < N;
  const long long row = (long long)m * N + (rowok ? n : N - 1);
  const int rows_here = N - n0 < 64 ? (N - n0 < 0 ? 0 : N - n0) : 64;
  const int blk = m * gridDim.x + blockIdx.x;

  for (int c0 = 0; c0 < cin; c0 += kPanel) {
    float acc[kPanel];
#pragma unroll
    for (int c = 0; c < kPanel; ++c) acc[c] = 0.0f;
    for (int kb = 0; kb < cout; kb += kKB) {
      float dy[kKB];
      const float4* ys = reinterpret_cast<const float4*>(y + row * cout + kb);
#pragma unroll
      for (int v = 0; v < kKB / 4; ++v) {
        const float4 t = ys[v];
        dy[4 * v + 0] = t.x;
        dy[4 * v + 1] = t.y;
        dy[4 * v + 2] = t.z;
        dy[4 * v + 3] = t.w;
      }
      if constexpr (TOP) {
#pragma unroll
        for (int k = 0; k < kKB; ++k) {
          const Coef q = coef[kb + k];
          const int am = argmax[(long long)m * cout + kb + k];   // wave-uniform
          const float g = gfeat[(long long)m * cout + kb + k];   // wave-uniform
          const float dzv = (am == n) ? g : 0.0f;
          dy[k] = __builtin_fmaf(q.alpha, dzv, __builtin_fmaf(q.gammap, dy[k], q.betap));
        }
      } else {
        const float4* zs = reinterpret_cast<const float4*>(dz + row * cout + kb);
        float dzv[kKB];
#pragma unroll
        for (int v = 0; v < kKB / 4; ++v) {
          const float4 t = zs[v];
          dzv[4 * v + 0] = t.x;
          dzv[4 * v + 1] = t.y;
          dzv[4 * v + 2] = t.z;
          dzv[4 * v + 3] = t.w;
        }
#pragma unroll
        for (int k = 0; k < kKB; ++k) {
          const Coef q = coef[kb + k];
          dy[k] = __builtin_fmaf(q.alpha, dzv[k], __builtin_fmaf(q.gammap, dy[k], q.betap));
        }
      }
#pragma unroll
      for (int k = 0; k < kKB; ++k) {
        const float* wk = w + (long long)(kb + k) * cin + c0;  // W[co][ci0..]: wave-uniform -> SGPRs
#pragma unroll
        for (int c = 0; c < kPanel; ++c) acc[c] = __builtin_fmaf(dy[k], wk[c], acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < kPanel; ++c) tile[wave][lane][c] = acc[c];
    __builtin_amdgcn_wave_barrier();
    // lane = input channel ci: ReLU mask of layer l-1, store dZ_{l-1}, column sums for its BN backward
    const BnP p = bnp_prev[c0 + lane];
    float s1 = 0.0f, s2 = 0.0f;
    const long long o = ((long long)m * N + n0) * cin + c0 + lane;
    for (int i = 0; i < rows_here; ++i) {
      const float yp = y_prev[o + (long long)i * cin];
      const float z = __builtin_fmaf(yp, p.scale, p.shift);
      const float d = z > 0.0f ? tile[wave][i][lane] : 0.0f;
      dz_prev[o + (long long)i * cin] = d;
      s1 += d;
      s2 = __builtin_fmaf(d, (yp - p.mean) * p.invstd, s2);
    }
    red[wave][lane][0] = s1;
    red[wave][lane][1] = s2;
    __syncthreads();
    if (threadIdx.x < kPanel) {
      float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
      for (int wv = 0; wv < kT / 64; ++wv) {
        t0 += red[wv][threadIdx.x][0];
        t1 += red[wv][threadIdx.x][1];
      }
      partial[((long long)blk * cin + c0 + threadIdx.x) * 2 + 0] = t0;
      partial[((long long)blk * cin + c0 + threadIdx.x) * 2 + 1] = t1;
    }
    __syncthreads();
  }
}

// K5: weight gradient of layer l, one part per block.column:  dWpart[m][co][ci] = sum_n dY[r,co] * A[r,ci].
// lane = output channel co (64 per wave-panel); the activated input row A[r, ci0..ci0+CI) is the scalar
// operand.  grid = (cout/64 * ceil(cin/64), M), block 256: the 4 waves split the part's rows, LDS-combine.
template <bool TOP, int CI>  // CI = input channels per pass: 64, or 3 for the first layer
__global__ __launch_bounds__(kT) void pn_wgrad_kernel(
    const float* __restrict__ y, const float* __restrict__ dz, const float* __restrict__ gfeat,
    const int* __restrict__ argmax, const Coef* __restrict__ coef, const float* __restrict__ a_prev,
    int cout, int cin, const float* __restrict__ valids, int N, float* __restrict__ dwpart) {
  __shared__ float red[kT / 64][CI][64 + 1];
  const int m = blockIdx.y;
  if (valids[m] == 0.0f) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int ci_groups = (cin + 63) / 64;
  const int co = (blockIdx.x / ci_groups) * 64 + lane;
  const int ci0 = (blockIdx.x % ci_groups) * 64;
  const Coef q = coef[co];
  float g = 0.0f;
  int am = -1;
  if constexpr (TOP) {
    g = gfeat[(long long)m * cout + co];
    am = argmax[(long long)m * cout + co];
  }
  float acc[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) acc[c] = 0.0f;
  const int per = (N + kT / 64 - 1) / (kT / 64);
  const int nb = wave * per, ne = nb + per < N ? nb + per : N;
  for (int n = nb; n < ne; ++n) {
    const long long r = (long long)m * N + n;
    float dzv;
    if constexpr (TOP) dzv = (am == n) ? g : 0.0f;
    else dzv = dz[r * cout + co];
    const float dy = __builtin_fmaf(q.alpha, dzv, __builtin_fmaf(q.gammap, y[r * cout + co], q.betap));
    const float* ar = a_prev + r * cin + ci0;  // wave-uniform -> SGPRs
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = __builtin_fmaf(dy, ar[c], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < CI; ++c) red[wave][c][lane] = acc[c];
  __syncthreads();
  // dwpart[m][co][ci]: lane = ci (contiguous 256 B stores), each wave takes every 4th output channel
  const int co_base = (blockIdx.x / ci_groups) * 64;
  if (lane < CI) {
    for (int cl = wave; cl < 64; cl += kT / 64) {
      float s = 0.0f;
#pragma unroll
      for (int wv = 0; wv < kT / 64; ++wv) s += red[wv][lane][cl];
      dwpart[((long long)m * cout + co_base + cl) * cin + ci0 + lane] = s;
    }
  }
}

// sum over valid parts: dW[i] = sum_m dwpart[m][i]
__global__ void pn_wgrad_reduce_kernel(const float* __restrict__ dwpart, const float* __restrict__ valids,
                                       int M, int elems, float* __restrict__ dw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= elems) return;
  float s = 0.0f;
  for (int m = 0; m < M; ++m)
    if (valids[m] != 0.0f) s += dwpart[(long long)m * elems + i];
  dw[i] = s;
}

// ---- host side ------------------------------------------------------------------------------------------------
struct Dims {
  int64_t M, N, F, rows, tiles;
  int C[6];  // channel widths: C[0] = 3 ... C[5] = F
};

Dims make_dims(int64_t M, int64_t N, int64_t F) {
  Dims d;
  d.M = M;
  d.N = N;
  d.F = F;
  d.rows = M * N;
  d.tiles = (N + kRows - 1) / kRows;
  d.C[0] = 3;
  d.C[1] = 64;
  d.C[2] = 64;
  d.C[3] = 64;
  d.C[4] = 128;
  d.C[5] = (int)F;
  return d;
}

struct PnWs {
  float* Y[6];    // pre-BN outputs, Y[1..5]
  float* A[6];    // activated outputs A[1..4] (A[0] = input points)
  float* dZ[6];   // backward: dZ[1..4]
  float* Wt[6];   // transposed weights [cin][cout]
  BnP* bnp[6];
  Coef* coef[6];
  float* partial;  // [M*tiles][maxC][2]
  float* dwpart;   // [M][128*F]
  float* count;
  int64_t total;
};

PnWs carve(float* base, const Dims& d) {
  PnWs w;
  float* p = base;
  auto take = [&](int64_t n) {
    float* r = p;
    p += (n + 3) / 4 * 4;  // keep 16-byte alignment for float4 accesses
    return r;
  };
  for (int l = 1; l <= 5; ++l) w.Y[l] = take(d.rows * d.C[l]);
  for (int l = 1; l <= 4; ++l) w.A[l] = take(d.rows * d.C[l]);
  for (int l = 1; l <= 4; ++l) w.dZ[l] = take(d.rows * d.C[l]);
  for (int l = 1; l <= 5; ++l) w.Wt[l] = take((int64_t)d.C[l - 1] * d.C[l]);
  for (int l = 1; l <= 5; ++l) w.bnp[l] = reinterpret_cast<BnP*>(take(4LL * d.C[l]));
  for (int l = 1; l <= 5; ++l) w.coef[l] = reinterpret_cast<Coef*>(take(4LL * d.C[l]));
  const int64_t maxc = d.F > 128 ? d.F : 128;
  w.partial = take(d.M * d.tiles * maxc * 2);
  w.dwpart = take(d.M * 128 * maxc);
  w.count = take(4);
  w.total = p - base;
  return w;
}

int check_dims(int64_t M, int64_t N, int64_t F, const char* who) {
  MPA_REQUIRE(M >= 0 && N >= 1 && F >= 64, "%s: bad sizes", who);
  MPA_REQUIRE(F % 64 == 0 && F <= 1024, "%s: feat_dim must be a multiple of 64 (<= 1024)", who);
  MPA_REQUIRE(M <= 65535 && N < (1 << 24), "%s: too many parts / points", who);
  return MPA_OK;
}

}  // namespace

extern "C" int mpa_pointnet_workspace(int64_t M, int64_t N, int64_t F, int64_t* float_elems,
                                      int64_t* int_elems) {
  if (int st = check_dims(M, N, F, "pointnet_workspace")) return st;
  MPA_REQUIRE(float_elems && int_elems, "pointnet_workspace: null pointer");
  const Dims d = make_dims(M, N, F);
  *float_elems = carve(nullptr, d).total;
  *int_elems = M * F;
  return MPA_OK;
}

extern "C" int mpa_pointnet_forward(const float* points, const float* valids, const float* const* conv_w,
                                    const float* const* bn_w, const float* const* bn_b,
                                    float* const* running_mean, float* const* running_var, int training,
                                    float momentum, float eps, int64_t M, int64_t N, int64_t F,
                                    float* float_ws, int32_t* int_ws, float* feat, void* stream) {
  if (int st = check_dims(M, N, F, "pointnet_forward")) return st;
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(points && valids && conv_w && bn_w && bn_b && running_mean && running_var && float_ws &&
                  int_ws && feat, "pointnet_forward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const Dims d = make_dims(M, N, F);
  const PnWs w = carve(float_ws, d);
  for (int l = 1; l <= 5; ++l) {
    const int e = d.C[l] * d.C[l - 1];
    hipLaunchKernelGGL(pn_prep_kernel, dim3((e + 255) / 256), dim3(256), 0, s, conv_w[l - 1], w.Wt[l],
                       d.C[l], d.C[l - 1]);
  }
  hipLaunchKernelGGL(pn_count_kernel, dim3(1), dim3(64), 0, s, valids, (int)M, (int)N, w.count);
  const dim3 grid((unsigned)d.tiles, (unsigned)M), blk(kT);
  for (int l = 1; l <= 5; ++l) {
    const float* in = l == 1 ? points : w.Y[l - 1];
    if (l == 1)
      hipLaunchKernelGGL(pn_fwd_layer_kernel<3>, grid, blk, 0, s, in, (const BnP*)nullptr, (float*)nullptr,
                         w.Wt[l], d.C[l], valids, (int)N, w.Y[l], w.partial);
    else if (d.C[l - 1] == 64)
      hipLaunchKernelGGL(pn_fwd_layer_kernel<64>, grid, blk, 0, s, in, w.bnp[l - 1], w.A[l - 1], w.Wt[l],
                         d.C[l], valids, (int)N, w.Y[l], w.partial);
    else
      hipLaunchKernelGGL(pn_fwd_layer_kernel<128>, grid, blk, 0, s, in, w.bnp[l - 1], w.A[l - 1], w.Wt[l],
                         d.C[l], valids, (int)N, w.Y[l], w.partial);
    const dim3 cg((d.C[l] + 63) / 64);
    if (training)
      hipLaunchKernelGGL(pn_bn_finalize_kernel, cg, dim3(64), 0, s, w.partial, valids, (int)M, (int)d.tiles,
                         d.C[l], w.count, bn_w[l - 1], bn_b[l - 1], running_mean[l - 1], running_var[l - 1],
                         momentum, eps, w.bnp[l]);
    else
      hipLaunchKernelGGL(pn_bn_from_running_kernel, cg, dim3(64), 0, s, d.C[l], bn_w[l - 1], bn_b[l - 1],
                         running_mean[l - 1], running_var[l - 1], eps, w.bnp[l]);
  }
  hipLaunchKernelGGL(pn_maxpool_kernel, dim3((unsigned)(F / 64), (unsigned)M), blk, 0, s, w.Y[5], w.bnp[5],
                     valids, (int)N, (int)F, feat, int_ws);
  return mpa::check_launch("pointnet_forward");
}

extern "C" int mpa_pointnet_backward(const float* grad_feat, const float* points, const float* valids,
                                     const float* const* conv_w, const float* const* bn_w, int64_t M,
                                     int64_t N, int64_t F, float* float_ws, const int32_t* int_ws,
                                     float* const* grad_conv_w, float* const* grad_bn_w,
                                     float* const* grad_bn_b, void* stream) {
  if (int st = check_dims(M, N, F, "pointnet_backward")) return st;
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(grad_feat && points && valids && conv_w && bn_w && float_ws && int_ws && grad_conv_w &&
                  grad_bn_w && grad_bn_b, "pointnet_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const Dims d = make_dims(M, N, F);
  const PnWs w = carve(float_ws, d);
  const dim3 grid((unsigned)d.tiles, (unsigned)M), blk(kT);
  hipLaunchKernelGGL(pn_bwd_top_kernel, dim3((unsigned)(F / 64)), dim3(64), 0, s, grad_feat, int_ws, w.Y[5],
                     valids, (int)M, (int)N, (int)F, w.count, bn_w[4], w.bnp[5], w.coef[5], grad_bn_w[4],
                     grad_bn_b[4]);
  for (int l = 5; l >= 1; --l) {
    const int cout = d.C[l], cin = d.C[l - 1];
    const float* a_prev = l == 1 ? points : w.A[l - 1];
    const dim3 wg((unsigned)((cout / 64) * ((cin + 63) / 64)), (unsigned)M);
    if (l == 5)
      hipLaunchKernelGGL((pn_wgrad_kernel<true, 64>), wg, blk, 0, s, w.Y[l], (const float*)nullptr, grad_feat,
                         int_ws, w.coef[l], a_prev, cout, cin, valids, (int)N, w.dwpart);
    else if (l == 1)
      hipLaunchKernelGGL((pn_wgrad_kernel<false, 3>), wg, blk, 0, s, w.Y[l], w.dZ[l], (const float*)nullptr,
                         (const int*)nullptr, w.coef[l], a_prev, cout, cin, valids, (int)N, w.dwpart);
    else
      hipLaunchKernelGGL((pn_wgrad_kernel<false, 64>), wg, blk, 0, s, w.Y[l], w.dZ[l], (const float*)nullptr,
                         (const int*)nullptr, w.coef[l], a_prev, cout, cin, valids, (int)N, w.dwpart);
    const int elems = cout * cin;
    hipLaunchKernelGGL(pn_wgrad_reduce_kernel, dim3((elems + 255) / 256), dim3(256), 0, s, w.dwpart, valids,
                       (int)M, elems, grad_conv_w[l - 1]);
    if (l == 1) break;
    if (l == 5)
      hipLaunchKernelGGL(pn_dgrad_kernel<true>, grid, blk, 0, s, w.Y[l], (const float*)nullptr, grad_feat,
                         int_ws, w.coef[l], conv_w[l - 1], cout, cin, w.Y[l - 1], w.bnp[l - 1], valids, (int)N,
                         w.dZ[l - 1], w.partial);
    else
      hipLaunchKernelGGL(pn_dgrad_kernel<false>, grid, blk, 0, s, w.Y[l], w.dZ[l], (const float*)nullptr,
                         (const int*)nullptr, w.coef[l], conv_w[l - 1], cout, cin, w.Y[l - 1], w.bnp[l - 1],
                         valids, (int)N, w.dZ[l - 1], w.partial);
    hipLaunchKernelGGL(pn_bwd_coef_kernel, dim3((unsigned)((cin + 63) / 64)), dim3(64), 0, s, w.partial, valids,
                       (int)M, (int)d.tiles, cin, w.count, bn_w[l - 2], w.bnp[l - 1], w.coef[l - 1],
                       grad_bn_w[l - 2], grad_bn_b[l - 2]);
  }
  return mpa::check_launch("pointnet_backward");
}
