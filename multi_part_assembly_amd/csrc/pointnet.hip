// PointNet part encoder — forward and backward (training-mode BatchNorm) for gfx950.
//
// Replaces the torch module of the reference (multi_part_assembly/models/modules/encoder/pointnet.py:6-41):
// 5 x [1x1 Conv1d (no bias) -> BatchNorm1d -> ReLU (none after the last)] 3-64-64-64-128-F, max over the
// N points of each part; and the compaction around it (models/pn_transformer/network.py:59-68): the
// kernels take ALL B*P part slots plus the validity mask and simply skip padded parts, so launch
// shapes are static and no device->host sync is needed.
//
// Design:
//   * fp32 throughout (parity bar 1e-4).  The 1x1 convolutions are GEMMs over [points x channels]
//     and run on the matrix cores with the exact-fp32 MFMA  v_mfma_f32_32x32x2_f32  (same peak as the
//     fp32 VALU on gfx950, but one instruction consumes a 32x2 and a 2x32 operand slice from ONE VGPR
//     each, so operand traffic per FLOP is 32x lower than with scalar-operand FMAs).
//   * activations are point-major  [row = part*N + point][channel].  With the MFMA K index split as
//     "lanes 0-31 take k in [0,K/2), lanes 32-63 take [K/2,K)", every lane's operand fragments are
//     CONTIGUOUS runs of its own row (activations) or of a weight row — plain float4 loads straight
//     from global/L2, no LDS staging, no transposes.  Weights stay register-resident while a wave
//     walks its 32-point tiles.
//   * BatchNorm+ReLU of layer l-1 is applied on the fly wherever layer l needs its input (forward GEMM,
//     weight-gradient GEMM): only the pre-BatchNorm outputs Y_l are ever stored; the
//     per-channel sums BatchNorm needs fall out of the accumulator layout (a lane holds 16 rows of
//     one output channel) and are reduced in a fixed order: deterministic, no atomics.
//   * backward: BatchNorm backward is the per-channel affine map dY = alpha*dZ + gamma'*Y + beta'
//     (coefficients from two column sums), applied on the fly as the operand of the input-gradient
//     GEMM (dA = dY W) and of the weight-gradient GEMM (dW = dY^T A, K = points, one part per block,
//     parts summed by a second deterministic stage).  The 3-channel first layer uses scalar-operand
//     VALU panels (weights through the scalar cache), K = 3 being far too thin for a matrix core.
#include <type_traits>

#include "common.h"
#include "coop_reduce.h"

namespace {

constexpr int kT = 256;      // threads per block (4 waves)
using mpa::CoopWs;
using mpa::coop_colsum;
using mpa::kEB;
using mpa::kSlices;

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// MFMA 32x32 accumulator layout: lane l holds column (l & 31) and, in register r, row acc_row(r, l >> 5).
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// Returns `p` through an opaque asm so that loads from it are NOT hoisted out of the enclosing loop: the
// per-channel tables (BatchNorm scale/shift, backward coefficients) are loop-invariant, and LICM would
// otherwise pin K/2 x 5 values per lane in registers (and spill).  They are L1-resident; re-reading per
// tile is cheap.
template <typename T>
__device__ __forceinline__ const T* opaque(const T* p) {
  asm volatile("" : "+v"(p));
  return p;
}

// Per-layer BatchNorm parameters, struct-of-arrays [4][C]: scale, shift, mean, invstd
//   z = y * scale + shift  ==  gamma * (y - mean) * invstd + beta
// Backward coefficients [3][C]: alpha, gammap, betap  with  dY = alpha*dZ + gammap*Y + betap.

// ---- small kernels ---------------------------------------------------------------------------------------
// number of valid points (BatchNorm's sample count), the compact list of valid parts (vlist[0] = how many, part
// ids from vlist[4] on, ascending) that the persistent backward kernels walk, and the reset of the tickets.
// one block of 1024 threads.
__global__ __launch_bounds__(1024) void pn_count_kernel(const float* __restrict__ valids, int M, int N,
                                                        float* __restrict__ count, unsigned* __restrict__ ticket,
                                                        int* __restrict__ vlist, const float* __restrict__ w1 = nullptr,
                                                        float* __restrict__ wt1 = nullptr) {
  __shared__ int wcnt[16];
  if (threadIdx.x < 4) ticket[threadIdx.x] = 0u;  // the cooperative reductions' counters (reset after every use)
  // (the first layer's 64 x 3 weights transposed for the kernels that recompute that layer: rode in its own launch before)
  if (w1 != nullptr && threadIdx.x < 192) wt1[(threadIdx.x % 3) * 64 + threadIdx.x / 3] = w1[threadIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int base = 0;
  for (int m0 = 0; m0 < M; m0 += 1024) {
    const int m = m0 + threadIdx.x;
    const bool ok = m < M && valids[m] != 0.0f;
    const unsigned long long b = __ballot(ok);
    if (lane == 0) wcnt[wave] = __popcll(b);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int c = wcnt[k];
      before += k < wave ? c : 0;
      total += c;
    }
    if (ok) vlist[4 + base + before + __popcll(b & ((1ull << lane) - 1ull))] = m;
    base += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    count[0] = (float)base * (float)N;
    vlist[0] = base;
  }
}

// the (sum0, sum1) partial tables written by the forward / input-gradient kernels; rows of padded parts hold
// garbage and are skipped (valids == nullptr: one row per persistent block, all of them meaningful)
__device__ __forceinline__ bool reduce_partials(const float* __restrict__ partial,
                                                const float* __restrict__ valids, int M, int splits, int C,
                                                int c, const CoopWs ws, double& s0, double& s1) {
  return coop_colsum(M * splits, C, c, ws,
                     [&](int e, bool& ok, double& x, double& y) {
                       const float2 v = *reinterpret_cast<const float2*>(partial + ((long long)e * C + c) * 2);
                       ok = valids == nullptr || valids[e / splits] != 0.0f;
                       x = (double)v.x;
                       y = (double)v.y;
                     },
                     s0, s1);
}

// BatchNorm statistics -> scale/shift (+ running statistics).  grid = (C/64, ceil(M*splits/kEB)), block 1024.
__global__ __launch_bounds__(64 * kSlices) void pn_bn_finalize_kernel(
    const float* __restrict__ partial, const float* __restrict__ valids, int M, int splits, int C,
    const float* __restrict__ count, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
    float* __restrict__ bn, const CoopWs cw) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, ss;
  if (!reduce_partials(partial, valids, M, splits, C, c, cw, s, ss)) return;
  if (threadIdx.x >= 64) return;
  const double n = (double)count[0];
  if (n == 0.0) {  // no valid part in the whole call (the reference's BatchNorm would refuse an empty batch): a neutral
    bn[c] = 0.0f;  // map, running statistics untouched — every output row is a padded part's zero row anyway, and the
    bn[C + c] = beta[c];  // backward pass then produces zero gradients instead of 0 / 0
    bn[2 * C + c] = 0.0f;
    bn[3 * C + c] = 0.0f;
    return;
  }
  const double mean = s / n;
  double var = ss / n - mean * mean;  // biased: what BatchNorm normalises with
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / __builtin_sqrt(var + (double)eps));
  const float scale = gamma[c] * invstd;
  bn[c] = scale;
  bn[C + c] = beta[c] - (float)mean * scale;
  bn[2 * C + c] = (float)mean;
  bn[3 * C + c] = invstd;
  if (running_mean != nullptr) {  // running_var tracks the unbiased estimate
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
                                          float* __restrict__ bn) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / __builtin_sqrtf(running_var[c] + eps);
  const float scale = gamma[c] * invstd;
  bn[c] = scale;
  bn[C + c] = beta[c] - running_mean[c] * scale;
  bn[2 * C + c] = running_mean[c];
  bn[3 * C + c] = invstd;
}

__device__ __forceinline__ void write_coef(float* __restrict__ coef, int C, int c, float gamma,
                                           const float* __restrict__ bn, double s1, double s2, double n) {
  const float mean = bn[2 * C + c], invstd = bn[3 * C + c];
  const float alpha = gamma * invstd;
  if (n == 0.0) {  // no valid part: nothing to differentiate
    coef[c] = coef[C + c] = coef[2 * C + c] = 0.0f;
    return;
  }
  const float gammap = (float)(-(double)alpha * s2 / n * (double)invstd);
  coef[c] = alpha;
  coef[C + c] = gammap;
  coef[2 * C + c] = (float)(-(double)alpha * s1 / n - (double)gammap * (double)mean);
}

// coefficients of layer l from the (s1, s2) partials written by the input-gradient kernel of layer l+1
__global__ __launch_bounds__(64 * kSlices) void pn_bwd_coef_kernel(
    const float* __restrict__ partial, const float* __restrict__ valids, int M, int splits, int C,
    const float* __restrict__ count, const float* __restrict__ gamma, const float* __restrict__ bn,
    float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta, const CoopWs cw) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s1, s2;
  if (!reduce_partials(partial, valids, M, splits, C, c, cw, s1, s2)) return;
  if (threadIdx.x >= 64) return;
  write_coef(coef, C, c, gamma[c], bn, s1, s2, (double)count[0]);
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
}

// layer-5 coefficients: dZ5 is sparse, grad_feat[m,c] sits at row argmax[m,c] whose pre-BatchNorm value the
// forward saved in ybest[m,c].  grid = (F/64, ceil(M/kEB)), block 1024.
__global__ __launch_bounds__(64 * kSlices) void pn_bwd_top_kernel(
    const float* __restrict__ gfeat, const int* __restrict__ argmax, const float* __restrict__ ybest,
    const float* __restrict__ valids, int M, int F, const float* __restrict__ count,
    const float* __restrict__ gamma, const float* __restrict__ bn, float* __restrict__ coef,
    float* __restrict__ dgamma, float* __restrict__ dbeta, const CoopWs cw) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const float mean = bn[2 * F + c], invstd = bn[3 * F + c];
  double s1, s2;
  const bool last = coop_colsum(M, F, c, cw,
                                [&](int m, bool& ok, double& x, double& y) {
                                  const long long o = (long long)m * F + c;
                                  const float g = gfeat[o];
                                  ok = valids[m] != 0.0f && argmax[o] >= 0;  // all-NaN column: no gradient
                                  x = (double)g;
                                  y = (double)g * (double)((ybest[o] - mean) * invstd);
                                },
                                s1, s2);
  if (!last || threadIdx.x >= 64) return;
  write_coef(coef, F, c, gamma[c], bn, s1, s2, (double)count[0]);
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
}

// ---- top-2 extrema records (last layer) --------------------------------------------------------------------------
// The last layer has no ReLU and feeds a max over the points, so its output tensor is never stored: BatchNorm being
// a per-channel monotone map, max_n z[n] is the image of max_n y[n] (scale > 0) or min_n y[n] (scale < 0).  The
// sign of scale = gamma * invstd is gamma's, known before the statistics are: the forward GEMM keeps, per (part,
// channel), the two largest sign(gamma) * y with their point indices (order: value descending, index ascending;
// gamma == 0 maps every point to `shift` and is handled in the finalize kernel).  Two, because z = fma(y, scale, shift) can round two distinct y to
// the SAME z, and the reference's arg-max then is the lower index of the two — decided once scale/shift are known.
struct Top2 {
  float v1, v2;
  int n1, n2;
};
constexpr int kNoIdx = 0x7fffffff;

__device__ __forceinline__ Top2 top2_empty() { return Top2{-__builtin_inff(), -__builtin_inff(), kNoIdx, kNoIdx}; }
__device__ __forceinline__ bool top2_before(float y, int n, float y2, int n2) {
  return y > y2 || (y == y2 && n < n2);
}
// n is larger than every index pushed before
__device__ __forceinline__ void top2_push(Top2& t, float y, int n) {
  // selects, not branches: this runs once per accumulator element in the forward GEMM's epilogue
  const bool g1 = y > t.v1, g2 = y > t.v2;
  t.v2 = g1 ? t.v1 : (g2 ? y : t.v2);
  t.n2 = g1 ? t.n1 : (g2 ? n : t.n2);
  t.v1 = g1 ? y : t.v1;
  t.n1 = g1 ? n : t.n1;
}
__device__ __forceinline__ Top2 top2_merge(const Top2 a, const Top2 b) {
  Top2 r;
  if (top2_before(b.v1, b.n1, a.v1, a.n1)) {
    r.v1 = b.v1;
    r.n1 = b.n1;
    const bool s = top2_before(a.v1, a.n1, b.v2, b.n2);
    r.v2 = s ? a.v1 : b.v2;
    r.n2 = s ? a.n1 : b.n2;
  } else {
    r.v1 = a.v1;
    r.n1 = a.n1;
    const bool s = top2_before(a.v2, a.n2, b.v1, b.n1);
    r.v2 = s ? a.v2 : b.v1;
    r.n2 = s ? a.n2 : b.n1;
  }
  return r;
}
__device__ __forceinline__ Top2 top2_shfl_xor(const Top2 t, int mask) {
  return Top2{__shfl_xor(t.v1, mask, 64), __shfl_xor(t.v2, mask, 64), __shfl_xor(t.n1, mask, 64),
              __shfl_xor(t.n2, mask, 64)};
}

// BatchNorm of the last layer + max over the points of each part from the per-block top-2 records.
// topv/topn [M*splits][F][2] = the two largest s*y (s = the sign of the channel's gamma) and their indices.
// One thread per (m, c).
__global__ void pn_top_finalize_kernel(const float* __restrict__ topv, const int* __restrict__ topn,
                                       const float* __restrict__ bn, const float* __restrict__ valids,
                                       const float* __restrict__ y4, const float* __restrict__ bn4,
                                       const float* __restrict__ w5, int M, int N, int F, int C4, int splits,
                                       float* __restrict__ feat, int* __restrict__ argmax,
                                       float* __restrict__ ybest) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * F) return;
  const int m = (int)(i / F), c = (int)(i % F);
  if (valids[m] == 0.0f) {
    feat[i] = 0.0f;  // padded slots hold zeros (network.py:66)
    argmax[i] = -1;
    ybest[i] = 0.0f;
    return;
  }
  Top2 t = top2_empty();
  for (int sp = 0; sp < splits; ++sp) {
    const long long o = (((long long)m * splits + sp) * F + c) * 2;
    const float2 v = *reinterpret_cast<const float2*>(topv + o);
    const int2 n = *reinterpret_cast<const int2*>(topn + o);
    t = top2_merge(t, Top2{v.x, v.y, n.x, n.y});
  }
  const float scale = bn[c], shift = bn[F + c];
  float z, y;
  int arg;
  if (scale != 0.0f) {  // scale = gamma * invstd has gamma's sign: the records hold the extrema of sign(gamma) * y
    const float sg = scale > 0.0f ? 1.0f : -1.0f;
    const float y1 = sg * t.v1, y2 = sg * t.v2;
    const float z1 = __builtin_fmaf(y1, scale, shift), z2 = __builtin_fmaf(y2, scale, shift);
    const bool second = t.n2 != kNoIdx && z2 == z1 && t.n2 < t.n1;
    arg = t.n1 == kNoIdx ? -1 : (second ? t.n2 : t.n1);  // no candidate: an all-NaN column
    y = second ? y2 : y1;
    z = z1;
  } else {  // gamma == 0: every point maps to `shift`, the arg-max is the first point; its y is recomputed
    arg = 0;
    z = shift;
    const float* row = y4 + (long long)m * N * C4;
    y = 0.0f;
    for (int k = 0; k < C4; ++k)
      y = __builtin_fmaf(__builtin_fmaxf(__builtin_fmaf(row[k], bn4[k], bn4[C4 + k]), 0.0f), w5[(long long)c * C4 + k], y);
  }
  feat[i] = z;
  argmax[i] = arg;
  ybest[i] = y;
}

// Per part: the arg-max entries (row, channel, alpha*grad) sorted by 32-row tile (ties: channel order), and the
// tile offsets — the sparse operand of the last layer's input gradient.  grid = M, block = F threads.
// Blocks [M, M + C4 + 1) of the same launch compute Q and c0 (pn_top_q below): two small, independent, latency-bound jobs
// that both wait for the coefficients of the last BatchNorm — side by side instead of one after the other.
__device__ __forceinline__ void pn_top_q_row(const float* __restrict__ w5, const float* __restrict__ coef, int F, int C4,
                                             float* __restrict__ q, int k);
__global__ void pn_top_csr_kernel(const int* __restrict__ argmax, const float* __restrict__ gfeat,
                                  const float* __restrict__ coef, const float* __restrict__ valids, int N, int F,
                                  int* __restrict__ erow, int* __restrict__ ech, float* __restrict__ eval,
                                  int* __restrict__ tptr, int M, const float* __restrict__ w5, int C4,
                                  float* __restrict__ q) {
  extern __shared__ int bins[];  // [F]
  if ((int)blockIdx.x >= M) {
    pn_top_q_row(w5, coef, F, C4, q, (int)blockIdx.x - M);
    return;
  }
  const int m = blockIdx.x, c = threadIdx.x, T = (N + 31) / 32;
  if (valids[m] == 0.0f) return;
  const int arg = argmax[(long long)m * F + c];
  const int bin = arg >= 0 ? arg >> 5 : T;  // T: no entry
  bins[c] = bin;
  __syncthreads();
  int below = 0, rank = 0;
  for (int k = 0; k < F; ++k) {
    const int b = bins[k];
    below += b < bin ? 1 : 0;
    rank += (b == bin && k < c) ? 1 : 0;
  }
  if (arg >= 0) {
    const long long o = (long long)m * F + below + rank;
    erow[o] = arg;
    ech[o] = c;
    eval[o] = coef[c] * gfeat[(long long)m * F + c];  // alpha_c * grad_feat[m, c]
  }
  for (int t = c; t <= T; t += blockDim.x) {
    int cnt = 0;
    for (int k = 0; k < F; ++k) cnt += bins[k] < t ? 1 : 0;
    tptr[(long long)m * (T + 1) + t] = cnt;
  }
}

// Q[k][d] = sum_c gammap_c W5[c][k] W5[c][d]  (dA4 = A4 Q + c0 + sparse),  c0[d] = sum_c betap_c W5[c][d].
// grid = C4 + 1 (row k; the extra block writes c0), block = C4 threads (d).
__device__ __forceinline__ void pn_top_q_row(const float* __restrict__ w5, const float* __restrict__ coef, int F, int C4,
                                             float* __restrict__ q, int k) {
  for (int d = threadIdx.x; d < C4; d += blockDim.x) {
    float acc = 0.0f;
    if (k < C4) {
#pragma unroll 8
      for (int c = 0; c < F; ++c)
        acc = __builtin_fmaf(coef[F + c] * w5[(long long)c * C4 + k], w5[(long long)c * C4 + d], acc);
    } else {
#pragma unroll 8
      for (int c = 0; c < F; ++c) acc = __builtin_fmaf(coef[2 * F + c], w5[(long long)c * C4 + d], acc);
    }
    q[(long long)k * C4 + d] = acc;
  }
}

// Weight gradient of the last layer:
//   dW5[c][ci] = alpha_c sum_m g[m,c] A4[m, argmax[m,c], ci]  +  gammap_c (W5 G)[c][ci]  +  betap_c a4sum[ci]
// with G = A4^T A4 and a4sum the column sums of A4 over all valid points (gram[C4][C4] followed by a4sum[C4]).
// grid = F (channel c), block 1024 = C4(=128) columns x 8 part-slices.
__global__ __launch_bounds__(1024) void pn_top_wgrad_kernel(
    const float* __restrict__ gfeat, const int* __restrict__ argmax, const float* __restrict__ valids,
    const float* __restrict__ y4, const float* __restrict__ bn4, const float* __restrict__ w5,
    const float* __restrict__ coef, const float* __restrict__ gram, int M, int N, int F,
    float* __restrict__ dw5) {
  constexpr int C4 = 128, S = 8, U = 80, CH = S * U;  // parts per chunk: the shipped M = 640 is one
  __shared__ float sm[2][S][C4];
  __shared__ int arg_s[CH];    // arg-max row of part m (of this output channel), -1: no contribution
  __shared__ float g_s[CH];
  const int c = blockIdx.x, ci = threadIdx.x & (C4 - 1), slice = threadIdx.x >> 7;
  const float sc = bn4[ci], sh = bn4[C4 + ci];
  // the dense term W5 G of this output channel: every slice takes 16 of the 128 k (requested now, used at the end — on
  // slice 0 alone it was a chain of 128 dependent FMAs behind four batches of loads at the end of a latency-bound kernel)
  float wk[C4 / S], gk[C4 / S];
#pragma unroll
  for (int k = 0; k < C4 / S; ++k) {
    wk[k] = w5[(long long)c * C4 + slice * (C4 / S) + k];
    gk[k] = gram[(slice * (C4 / S) + k) * C4 + ci];
  }
  float acc = 0.0f;
  for (int m0 = 0; m0 < M; m0 += CH) {
    // level 1, once per part instead of once per (part, input channel): arg-max row and gradient of the chunk's parts
    if ((int)threadIdx.x < CH) {
      const int m = m0 + (int)threadIdx.x, mm = m < M ? m : M - 1;
      const int arg = argmax[(long long)mm * F + c];
      arg_s[threadIdx.x] = (m < M && valids[mm] != 0.0f && arg >= 0) ? arg : -1;
      g_s[threadIdx.x] = gfeat[(long long)mm * F + c];
    }
    __syncthreads();
    // level 2: the rows themselves, UB of a thread's requests in flight together (all 80 would need 160 address registers)
    constexpr int UB = 40;
#pragma unroll 1
    for (int u0 = 0; u0 < U; u0 += UB) {
      float yv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i = slice + (u0 + u) * S, arg = arg_s[i], mm = m0 + i < M ? m0 + i : M - 1;
        yv[u] = y4[((long long)mm * N + (arg >= 0 ? arg : 0)) * C4 + ci];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int i = slice + (u0 + u) * S;
        if (arg_s[i] >= 0) acc = __builtin_fmaf(g_s[i], __builtin_fmaxf(__builtin_fmaf(yv[u], sc, sh), 0.0f), acc);
      }
    }
    __syncthreads();
  }
  float wg = 0.0f;
#pragma unroll
  for (int k = 0; k < C4 / S; ++k) wg = __builtin_fmaf(wk[k], gk[k], wg);
  sm[0][slice][ci] = acc;
  sm[1][slice][ci] = wg;
  __syncthreads();
  if (slice != 0) return;
  float sparse = 0.0f;
  wg = 0.0f;
#pragma unroll
  for (int k = 0; k < S; ++k) {
    sparse += sm[0][k][ci];
    wg += sm[1][k][ci];
  }
  dw5[(long long)c * C4 + ci] = coef[c] * sparse + coef[F + c] * wg + coef[2 * F + c] * gram[C4 * C4 + ci];
}

// ---- first layer (3 -> 64): scalar-operand VALU panel ------------------------------------------------------
// lane = point; the 3x64 transposed weights are wave-uniform (scalar cache).  grid = (ceil(N/256), M).
// partial [M*tiles][64][2].
// STORE = false (the shipped path): only the BatchNorm sums leave the kernel.  Y1 — 256 bytes per point, 3 FMAs per
// value from a 12-byte point — is never written: the three kernels that consume it (layer 2 forward, layer 2's fused
// backward, layer 1's weight gradient) recompute their tile from the points with first_layer_y below, the SAME
// operation sequence, so every consumer sees the bits the statistics were taken from.  Saves one write and three reads
// of the [rows x 64] tensor per step (4 x 94 MB at 366 valid parts).
__device__ __forceinline__ float first_layer_y(float a0, float a1, float a2, float w0, float w1, float w2) {
  float v = a0 * w0;
  v = __builtin_fmaf(a1, w1, v);
  return __builtin_fmaf(a2, w2, v);
}

template <bool STORE>
__global__ __launch_bounds__(kT) void pn_fwd_first_kernel(const float* __restrict__ pts,
                                                          const float* __restrict__ wt,
                                                          const float* __restrict__ valids, int N,
                                                          float* __restrict__ y_out,
                                                          float* __restrict__ partial) {
  __shared__ float tile[kT / 64][64][65];
  __shared__ float red[kT / 64][64][2];
  const int m = blockIdx.y;
  if (valids[m] == 0.0f) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = blockIdx.x * kT + wave * 64, n = n0 + lane;
  const long long row = (long long)m * N + (n < N ? n : N - 1);
  const int rows_here = N - n0 < 64 ? (N - n0 < 0 ? 0 : N - n0) : 64;
  const float a0 = pts[row * 3 + 0], a1 = pts[row * 3 + 1], a2 = pts[row * 3 + 2];
#pragma unroll
  for (int c = 0; c < 64; ++c) tile[wave][lane][c] = first_layer_y(a0, a1, a2, wt[c], wt[64 + c], wt[128 + c]);
  __builtin_amdgcn_wave_barrier();
  float s = 0.0f, ss = 0.0f;
  float* dst = y_out + ((long long)m * N + n0) * 64 + lane;
  for (int i = 0; i < rows_here; ++i) {
    const float v = tile[wave][i][lane];
    if constexpr (STORE) dst[(long long)i * 64] = v;
    s += v;
    ss = __builtin_fmaf(v, v, ss);
  }
  red[wave][lane][0] = s;
  red[wave][lane][1] = ss;
  __syncthreads();
  if (threadIdx.x < 64) {
    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
    for (int w = 0; w < kT / 64; ++w) {
      t0 += red[w][threadIdx.x][0];
      t1 += red[w][threadIdx.x][1];
    }
    const long long blk = (long long)m * gridDim.x + blockIdx.x;
    partial[(blk * 64 + threadIdx.x) * 2 + 0] = t0;
    partial[(blk * 64 + threadIdx.x) * 2 + 1] = t1;
  }
}

// ---- MFMA forward layer ---------------------------------------------------------------------------------------
// Y[rows x cout] = relu(bn_prev(Yprev))[rows x CIN] . W[cout x CIN]^T.
// A block owns 64*PANELS output channels and walks block tiles of RB = 32 * (4 / PANELS) rows of its (part, split):
// wave w computes the 32-row sub-tile w / PANELS for the 64-channel panel w % PANELS, its 64 x CIN weight panel
// register-resident (K split by lane half, so every fragment is a contiguous run).  The block tile — one
// contiguous RB*CIN-float run of the point-major input — is fetched by all 256 threads with coalesced 16-byte
// loads ONE TILE AHEAD (registers), gets the previous layer's BatchNorm + ReLU on its way into a double-buffered
// LDS panel, and is read back as MFMA fragments: the input crosses HBM/L2 once per block and the global latency
// hides behind the previous tile's MFMA chain.  Persistent: grid = (min(M*splits, resident blocks), cout /
// (64*PANELS)), block 256; a block keeps its weight panel and walks (valid part, split) units from `vlist`.
// BatchNorm statistics fall out of the accumulator layout (fixed-order reduction, no atomics).
// TOP (last layer): Y is not stored; the block leaves the per-channel top-2 records of its rows instead.
// FIRST (layer 2): `in` holds the raw points [rows][3] and the layer's input Y1 is recomputed from them on the way
// into the panel (wt1 [3][64], see pn_fwd_first_kernel).
template <int CIN, int PANELS, bool TOP, bool FIRST = false>
__global__ __launch_bounds__(kT, 2) void pn_fwd_mfma_kernel(
    const float* __restrict__ in, const float* __restrict__ bn_prev, const float* __restrict__ w, int cout,
    const int* __restrict__ vlist, int N, int splits, float* __restrict__ y_out,
    float* __restrict__ partial, float* __restrict__ topv, int* __restrict__ topn,
    const float* __restrict__ gamma_top, const float* __restrict__ wt1 = nullptr) {
  static_assert(!FIRST || CIN == 64, "the recomputed input is the 64-channel first layer");
  constexpr int KH = CIN / 2;           // K values per lane-half
  constexpr int LD = CIN + 4;           // padded LDS row: conflict-free ds_read_b128 across rows
  constexpr int Q4 = CIN / 4;           // float4 per row
  constexpr int RT = 4 / PANELS;        // 32-row sub-tiles per block tile
  constexpr int RB = 32 * RT;           // rows per block tile
  constexpr int NLD = RB * Q4 / kT;     // float4 per thread and tile
  __shared__ __attribute__((aligned(16))) float buf[2][RB * LD];
  __shared__ float red[kT / 64][64][2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int panel = wave % PANELS, rt = wave / PANELS;
  const int cb = blockIdx.y * 64 * PANELS, c0 = cb + panel * 64;
  // B fragments (weights): tile t covers output channels c0+32t .. c0+32t+31; lane holds W[c][h*KH + s]
  float bw[2][KH];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float4* src = reinterpret_cast<const float4*>(w + (long long)(c0 + 32 * t + j) * CIN + h * KH);
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) {
      const float4 q = src[v];
      bw[t][4 * v + 0] = q.x;
      bw[t][4 * v + 1] = q.y;
      bw[t][4 * v + 2] = q.z;
      bw[t][4 * v + 3] = q.w;
    }
  }
  // staging role of this thread: float4 column c4 of rows rl0, rl0 + kT/Q4, ... (kT % Q4 == 0)
  const int c4 = threadIdx.x % Q4, rl0 = threadIdx.x / Q4;
  const float4 sc = reinterpret_cast<const float4*>(bn_prev)[c4];
  const float4 sh = reinterpret_cast<const float4*>(bn_prev + CIN)[c4];
  const int TB = (N + RB - 1) / RB;
  float4 raw[NLD];  // (FIRST: x, y, z of the row's point in .x .y .z)
  float4 w1a = {}, w1b = {}, w1c = {};  // FIRST: the first layer's weights of this thread's 4 channels
  if constexpr (FIRST) {
    w1a = reinterpret_cast<const float4*>(wt1)[c4];
    w1b = reinterpret_cast<const float4*>(wt1 + 64)[c4];
    w1c = reinterpret_cast<const float4*>(wt1 + 128)[c4];
  }
  int m = 0;
  auto fetch = [&](int tile) {
    if constexpr (FIRST) {
      const float* src = in + ((long long)m * N + (long long)tile * RB) * 3;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int rl = rl0 + i * (kT / Q4);
        const float* p = src + 3 * (tile * RB + rl < N ? rl : 0);
        raw[i] = make_float4(p[0], p[1], p[2], 0.0f);
      }
      return;
    }
    const float4* src = reinterpret_cast<const float4*>(in + ((long long)m * N + (long long)tile * RB) * CIN);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int rl = rl0 + i * (kT / Q4);
      raw[i] = tile * RB + rl < N ? src[i * kT + threadIdx.x] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
  };
  auto stash = [&](int tile, float* dst) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int rl = rl0 + i * (kT / Q4);
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if constexpr (FIRST) {
        const float a0 = raw[i].x, a1 = raw[i].y, a2 = raw[i].z;
        raw[i] = make_float4(first_layer_y(a0, a1, a2, w1a.x, w1b.x, w1c.x), first_layer_y(a0, a1, a2, w1a.y, w1b.y, w1c.y),
                             first_layer_y(a0, a1, a2, w1a.z, w1b.z, w1c.z), first_layer_y(a0, a1, a2, w1a.w, w1b.w, w1c.w));
      }
      if (tile * RB + rl < N) {  // rows past the part's end enter the MFMA as zeros
        v.x = __builtin_fmaxf(__builtin_fmaf(raw[i].x, sc.x, sh.x), 0.0f);
        v.y = __builtin_fmaxf(__builtin_fmaf(raw[i].y, sc.y, sh.y), 0.0f);
        v.z = __builtin_fmaxf(__builtin_fmaf(raw[i].z, sc.z, sh.z), 0.0f);
        v.w = __builtin_fmaxf(__builtin_fmaf(raw[i].w, sc.w, sh.w), 0.0f);
      }
      *reinterpret_cast<float4*>(dst + rl * LD + 4 * c4) = v;
    }
  };
  // TOP: BatchNorm's scale has gamma's sign, so only the extrema of sign(gamma) * y can become the part's maximum
  float sgn[2] = {1.0f, 1.0f};
  if constexpr (TOP) {
#pragma unroll
    for (int t = 0; t < 2; ++t) sgn[t] = gamma_top[c0 + 32 * t + j] < 0.0f ? -1.0f : 1.0f;
  }
  // persistent: the block keeps its weight panel and walks the (valid part, row split) units u, u + gridDim.x, ...
  const int U = vlist[0] * splits;
  int mnext = blockIdx.x < U ? vlist[4 + blockIdx.x / splits] : 0;
  for (int unit = blockIdx.x; unit < U; unit += gridDim.x) {
  m = mnext;
  {
    const int un = unit + gridDim.x;
    mnext = un < U ? vlist[4 + un / splits] : 0;  // needed one unit from now
  }
  const int sp = unit % splits, ob = m * splits + sp;  // ob: the unit's row in the per-unit output tables
  const int t_begin = (int)((long long)sp * TB / splits), t_end = (int)((long long)(sp + 1) * TB / splits);
  float s_[2] = {0.0f, 0.0f}, ss_[2] = {0.0f, 0.0f};
  Top2 hi[2] = {top2_empty(), top2_empty()};
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    float* cur = buf[(tile - t_begin) & 1];
    stash(tile, cur);
    __syncthreads();  // also orders this buffer's reuse: its previous readers finished before the last barrier
    if (tile + 1 < t_end) fetch(tile + 1);  // in flight during the MFMA chain below
    const int r0 = tile * RB + rt * 32;
    const float4* frag = reinterpret_cast<const float4*>(cur + (rt * 32 + j) * LD + h * KH);
    f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) {
      const float4 a = frag[v];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw[0][4 * v + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw[1][4 * v + 0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw[0][4 * v + 1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw[1][4 * v + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw[0][4 * v + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw[1][4 * v + 2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw[0][4 * v + 3], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw[1][4 * v + 3], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gn = r0 + acc_row(r, h);
      if constexpr (TOP) {  // rows past the part's end (all-zero operand rows) must not enter the extrema
        const float ninf = -__builtin_inff();
        const bool ok = gn < N;
        top2_push(hi[0], ok ? sgn[0] * acc0[r] : ninf, gn);
        top2_push(hi[1], ok ? sgn[1] * acc1[r] : ninf, gn);
      } else if (gn < N) {
        float* dst = y_out + ((long long)m * N + gn) * cout + c0 + j;
        dst[0] = acc0[r];
        dst[32] = acc1[r];
      }
      s_[0] += acc0[r];  // zero operand rows give exactly 0: no mask needed for the statistics
      ss_[0] = __builtin_fmaf(acc0[r], acc0[r], ss_[0]);
      s_[1] += acc1[r];
      ss_[1] = __builtin_fmaf(acc1[r], acc1[r], ss_[1]);
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {  // lanes l and l+32 hold the same channel
    s_[t] += __shfl_xor(s_[t], 32, 64);
    ss_[t] += __shfl_xor(ss_[t], 32, 64);
  }
  if (h == 0) {
    red[wave][j][0] = s_[0];
    red[wave][j][1] = ss_[0];
    red[wave][32 + j][0] = s_[1];
    red[wave][32 + j][1] = ss_[1];
  }
  __syncthreads();
  if (threadIdx.x < 64 * PANELS) {  // thread -> (panel, channel); the RT waves of the panel in fixed order
    const int pn = threadIdx.x >> 6, ch = threadIdx.x & 63;
    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      t0 += red[q * PANELS + pn][ch][0];
      t1 += red[q * PANELS + pn][ch][1];
    }
    const long long o = ((long long)ob * cout + cb + threadIdx.x) * 2;
    partial[o] = t0;
    partial[o + 1] = t1;
  }
  if constexpr (TOP) {
    __shared__ Top2 tsm[kT / 64][64];
#pragma unroll
    for (int t = 0; t < 2; ++t) {  // lanes l and l+32 hold the same channel
      hi[t] = top2_merge(hi[t], top2_shfl_xor(hi[t], 32));
    }
    if (h == 0) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        tsm[wave][32 * t + j] = hi[t];
      }
    }
    __syncthreads();
    if (threadIdx.x < 64 * PANELS) {
      const int pn = threadIdx.x >> 6, ch = threadIdx.x & 63;
      Top2 a = tsm[pn][ch];
#pragma unroll
      for (int q = 1; q < RT; ++q) {
        a = top2_merge(a, tsm[q * PANELS + pn][ch]);
      }
      const long long o = ((long long)ob * cout + cb + threadIdx.x) * 2;
      *reinterpret_cast<float2*>(topv + o) = make_float2(a.v1, a.v2);
      *reinterpret_cast<int2*>(topn + o) = make_int2(a.n1, a.n2);
    }
  }
  __syncthreads();  // the reduction scratch is free again before the next unit reaches it
  }  // unit
}

// ---- last layer forward on the bf16 matrix cores, fp32-grade --------------------------------------------------------------
// The 128 -> F layer is the one forward GEMM the matrix cores bind (2.3e10 FLOP against 181 MB of input).  Its operands
// are split into three bf16 terms each, x = h + m + l (exact), and the six products of order <= 2 run on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (csrc/dg_gemm_split.h has the error analysis: the dropped terms are
// one fp32 rounding of the product) — 192 instead of 512 matrix-core cycles per 16 k-values.  A block of NW = F / 32
// waves walks 32-row tiles of its (valid part, split) units like pn_fwd_mfma_kernel; wave w owns output channels
// 32w .. 32w+31 with its split weight slab register-resident (96 VGPRs) and ALL 32 rows of the tile, so the BatchNorm
// sums and the top-2 records of a channel live in one wave (no cross-wave reduction).  The tile is fetched one tile
// ahead, gets the previous layer's BatchNorm + ReLU and the split on its way into a double-buffered LDS panel of three
// bf16 planes (rows of 3 x 256 + 16 bytes: conflict-free 16-byte fragment reads).
typedef __bf16 pn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pn_bf16x4 __attribute__((ext_vector_type(4)));
template <int CIN, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void pn_fwd_split_kernel(
    const float* __restrict__ in, const float* __restrict__ bn_prev, const float* __restrict__ w, int cout,
    const int* __restrict__ vlist, int N, int splits, float* __restrict__ partial, float* __restrict__ topv,
    int* __restrict__ topn, const float* __restrict__ gamma_top) {
  constexpr int NT = 64 * NW, KS = CIN / 16, Q4 = CIN / 4;
  constexpr int ROWB = 3 * CIN * 2 + 16;  // LDS row: h | m | l planes of CIN bf16 each + pad (an odd multiple of 16)
  constexpr int NLD = 32 * Q4 / NT;       // float4 per thread and tile
  static_assert(32 * Q4 % NT == 0 && NT % Q4 == 0, "staging layout");
  __shared__ __attribute__((aligned(16))) unsigned char buf[2][32 * ROWB];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int c0 = 32 * wave;
  // split weight slab: lane (j, h) holds, per k-step, W[c0 + j][16 ks + 8 h .. + 7] as h / m / l
  pn_bf16x8 bh[KS], bm[KS], bl[KS];
  {
    const float* src = w + (long long)(c0 + j) * CIN + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 q0 = *reinterpret_cast<const float4*>(src + 16 * ks), q1 = *reinterpret_cast<const float4*>(src + 16 * ks + 4);
      const float f[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bh[ks][u] = (__bf16)f[u];
        const float r1 = f[u] - (float)bh[ks][u];
        bm[ks][u] = (__bf16)r1;
        bl[ks][u] = (__bf16)(r1 - (float)bm[ks][u]);
      }
    }
  }
  const int c4 = threadIdx.x % Q4, rl0 = threadIdx.x / Q4;
  const float4 sc = reinterpret_cast<const float4*>(bn_prev)[c4];
  const float4 sh = reinterpret_cast<const float4*>(bn_prev + CIN)[c4];
  const int TB = (N + 31) / 32;
  float4 raw[NLD];
  int m = 0;
  auto fetch = [&](int tile) {
    const float4* src = reinterpret_cast<const float4*>(in + ((long long)m * N + (long long)tile * 32) * CIN);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int rl = rl0 + i * (NT / Q4);
      const int rr = tile * 32 + rl < N ? rl : N - 1 - tile * 32;  // rows past the part's end: any row of the part
      raw[i] = src[rr * Q4 + c4];
    }
  };
  auto stash = [&](int tile, unsigned char* dst) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int rl = rl0 + i * (NT / Q4);
      const float4 r = raw[i];
      float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (tile * 32 + rl < N) {  // rows past the part's end enter the MFMA as zeros
        v[0] = __builtin_fmaxf(__builtin_fmaf(r.x, sc.x, sh.x), 0.0f);
        v[1] = __builtin_fmaxf(__builtin_fmaf(r.y, sc.y, sh.y), 0.0f);
        v[2] = __builtin_fmaxf(__builtin_fmaf(r.z, sc.z, sh.z), 0.0f);
        v[3] = __builtin_fmaxf(__builtin_fmaf(r.w, sc.w, sh.w), 0.0f);
      }
      pn_bf16x4 ph, pm, pl;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ph[u] = (__bf16)v[u];
        const float r1 = v[u] - (float)ph[u];
        pm[u] = (__bf16)r1;
        pl[u] = (__bf16)(r1 - (float)pm[u]);
      }
      unsigned char* p = dst + rl * ROWB + 8 * c4;
      *reinterpret_cast<pn_bf16x4*>(p) = ph;
      *reinterpret_cast<pn_bf16x4*>(p + 2 * CIN) = pm;
      *reinterpret_cast<pn_bf16x4*>(p + 4 * CIN) = pl;
    }
  };
  const float sgn = gamma_top[c0 + j] < 0.0f ? -1.0f : 1.0f;  // only the extrema of sign(gamma) * y can become the maximum
  const int U = vlist[0] * splits;
  int mnext = blockIdx.x < U ? vlist[4 + blockIdx.x / splits] : 0;
  for (int unit = blockIdx.x; unit < U; unit += gridDim.x) {
    m = mnext;
    {
      const int un = unit + gridDim.x;
      mnext = un < U ? vlist[4 + un / splits] : 0;
    }
    const int sp = unit % splits, ob = m * splits + sp;
    const int t_begin = (int)((long long)sp * TB / splits), t_end = (int)((long long)(sp + 1) * TB / splits);
    float s_ = 0.0f, ss_ = 0.0f;
    Top2 hi = top2_empty();
    if (t_begin < t_end) fetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
      unsigned char* cur = buf[(tile - t_begin) & 1];
      stash(tile, cur);
      __syncthreads();  // also orders this buffer's reuse: its previous readers finished before the last barrier
      if (tile + 1 < t_end) fetch(tile + 1);
      const unsigned char* arow = cur + j * ROWB + 16 * h;
      f32x16 acc = {0};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const pn_bf16x8 ah = *reinterpret_cast<const pn_bf16x8*>(arow + 32 * ks);
        const pn_bf16x8 am = *reinterpret_cast<const pn_bf16x8*>(arow + 2 * CIN + 32 * ks);
        const pn_bf16x8 al = *reinterpret_cast<const pn_bf16x8*>(arow + 4 * CIN + 32 * ks);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks], acc, 0, 0, 0);
      }
      const int r0 = tile * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gn = r0 + acc_row(r, h);
        top2_push(hi, gn < N ? sgn * acc[r] : -__builtin_inff(), gn);  // rows past the end must not enter the extrema
        s_ += acc[r];  // zero operand rows give exactly 0: no mask needed for the statistics
        ss_ = __builtin_fmaf(acc[r], acc[r], ss_);
      }
    }
    // lanes l and l + 32 hold the same channel (rows 4h .. of every 8): combine, lane half 0 writes the unit's records
    s_ += __shfl_xor(s_, 32, 64);
    ss_ += __shfl_xor(ss_, 32, 64);
    hi = top2_merge(hi, top2_shfl_xor(hi, 32));
    if (h == 0) {
      const long long o = ((long long)ob * cout + c0 + j) * 2;
      partial[o] = s_;
      partial[o + 1] = ss_;
      *reinterpret_cast<float2*>(topv + o) = make_float2(hi.v1, hi.v2);
      *reinterpret_cast<int2*>(topn + o) = make_int2(hi.n1, hi.n2);
    }
    __syncthreads();  // every wave is done with the last tile's panel before the next unit's first stash
  }
}

// ---- MFMA input gradient -----------------------------------------------------------------------------------------
// dA[rows x cin] = dY[rows x K] . W[K x cin] with dY = alpha*dZ + gammap*Y + betap built on the fly,
// then the ReLU mask and the BatchNorm-backward column sums of layer l-1:
//   dZprev = dA where bn_prev(Yprev) > 0;  s1 += dZprev, s2 += dZprev * xhat_prev.
// Same block organisation as the forward GEMM: a block owns 32*NT*PANELS output channels and walks block tiles of
// RB = 32 * (4 / PANELS) rows; the dY tile is built by all 256 threads from Y and dZ fetched ONE TILE AHEAD with
// coalesced 16-byte loads, staged in a double-buffered LDS panel and read back as MFMA fragments; wave w computes
// the 32-row sub-tile w / PANELS for the channel panel w % PANELS with its K x 32*NT weight slab in registers.
// The Yprev values of the epilogue are requested before the MFMA chain.
// TOP (last layer, whose Y was never stored; K = cin = 128):  dA = A Q + c0 + S W5  with A = relu(bn_prev(Yprev))
// staged like the forward operand, w = Q (symmetric 128 x 128, c0 behind it) and S the sparse arg-max gradient:
// per tile a short extra MFMA chain over the part's CSR entries (erow, ech, eval; tptr = tile offsets) with the
// one-hot row selector as A operand and the W5 row of the entry's channel as B operand.
// Persistent like the forward GEMM: grid = (min(M*splits, resident blocks), cin / (32*NT*PANELS)), block 256.
// Only the TOP form is instantiated: layers 2-4 run pn_bwd_fused_kernel, which also produces the weight gradient.
template <int K, int NT, int PANELS, bool TOP>
__global__ __launch_bounds__(kT, 2) void pn_dgrad_mfma_kernel(
    const float* __restrict__ y, const float* __restrict__ dz, const float* __restrict__ coef,
    const float* __restrict__ w, int cin, const float* __restrict__ y_prev, const float* __restrict__ bn_prev,
    const int* __restrict__ vlist, int N, int splits, float* __restrict__ dz_prev,
    float* __restrict__ partial, const int* __restrict__ erow, const int* __restrict__ ech,
    const float* __restrict__ eval, const int* __restrict__ tptr, const float* __restrict__ w5, int F) {
  constexpr int KH = K / 2;            // K values per lane-half
  constexpr int LD = K + 4;
  constexpr int Q4 = K / 4;
  constexpr int RT = 4 / PANELS;       // 32-row sub-tiles per block tile
  constexpr int RB = 32 * RT;          // rows per block tile
  constexpr int NLD = RB * Q4 / kT;    // float4 per thread, tensor and tile
  constexpr int CW = 32 * NT;          // output channels per wave
  __shared__ __attribute__((aligned(16))) float buf[2][RB * LD];
  __shared__ float red[kT / 64][CW][2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int panel = wave % PANELS, rt = wave / PANELS;
  const int db = blockIdx.y * CW * PANELS, d0 = db + panel * CW;
  // B fragments: lane-half h, step s  <->  k = h*KH + s;  bw = W[k][d0 + 32t + j]
  float bw[NT][KH];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int s = 0; s < KH; ++s) bw[t][s] = w[(long long)(h * KH + s) * cin + d0 + 32 * t + j];
  float scp[NT], shp[NT], mnp[NT], isp[NT], c0v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ci = d0 + 32 * t + j;
    scp[t] = bn_prev[ci];
    shp[t] = bn_prev[cin + ci];
    mnp[t] = bn_prev[2 * cin + ci];
    isp[t] = bn_prev[3 * cin + ci];
    c0v[t] = TOP ? w[(long long)K * cin + ci] : 0.0f;
  }
  // staging role of this thread: float4 column c4 of rows rl0, rl0 + kT/Q4, ...; its per-column tables.
  // TOP: the staged operand is relu(bn_prev(Yprev)), tables = scale, shift; else alpha, gammap, betap.
  const int c4 = threadIdx.x % Q4, rl0 = threadIdx.x / Q4;
  const float4 ta = reinterpret_cast<const float4*>(TOP ? bn_prev : coef)[c4];
  const float4 tb = reinterpret_cast<const float4*>(TOP ? bn_prev + K : coef + K)[c4];
  const float4 tc = TOP ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : reinterpret_cast<const float4*>(coef + 2 * K)[c4];
  const int TB = (N + RB - 1) / RB;
  int m = 0;
  // TOP: the part's CSR (<= 256 entries, <= 1025 tile offsets) lives in LDS so that the per-tile sparse chain has
  // a single level of global loads (the W5 rows), issued before the tile's main MFMA chain
  constexpr int kMaxF = 256, kMaxT1 = 1032, kSP = 6;  // tile offsets: N <= 32768 points per part
  __shared__ int s_row[TOP ? kMaxF : 1], s_ch[TOP ? kMaxF : 1], s_ptr[TOP ? kMaxT1 : 1];
  __shared__ float s_val[TOP ? kMaxF : 1];
  float4 ry[NLD], rz[TOP ? 1 : NLD];
  auto fetch = [&](int tile) {
    const long long base = ((long long)m * N + (long long)tile * RB) * Q4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int rl = rl0 + i * (kT / Q4);
      const bool ok = tile * RB + rl < N;
      const long long o = base + i * kT + threadIdx.x;
      if constexpr (TOP) {
        ry[i] = ok ? reinterpret_cast<const float4*>(y_prev)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      } else {
        ry[i] = ok ? reinterpret_cast<const float4*>(y)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        rz[i] = ok ? reinterpret_cast<const float4*>(dz)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
  };
  auto stash = [&](int tile, float* dst) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int rl = rl0 + i * (kT / Q4);
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (tile * RB + rl < N) {  // rows past the part's end enter the MFMA as zeros
        if constexpr (TOP) {
          v.x = __builtin_fmaxf(__builtin_fmaf(ry[i].x, ta.x, tb.x), 0.0f);
          v.y = __builtin_fmaxf(__builtin_fmaf(ry[i].y, ta.y, tb.y), 0.0f);
          v.z = __builtin_fmaxf(__builtin_fmaf(ry[i].z, ta.z, tb.z), 0.0f);
          v.w = __builtin_fmaxf(__builtin_fmaf(ry[i].w, ta.w, tb.w), 0.0f);
        } else {
          v.x = __builtin_fmaf(ta.x, rz[i].x, __builtin_fmaf(tb.x, ry[i].x, tc.x));
          v.y = __builtin_fmaf(ta.y, rz[i].y, __builtin_fmaf(tb.y, ry[i].y, tc.y));
          v.z = __builtin_fmaf(ta.z, rz[i].z, __builtin_fmaf(tb.z, ry[i].z, tc.z));
          v.w = __builtin_fmaf(ta.w, rz[i].w, __builtin_fmaf(tb.w, ry[i].w, tc.w));
        }
      }
      *reinterpret_cast<float4*>(dst + rl * LD + 4 * c4) = v;
    }
  };
  // persistent: the block keeps its weight slab and walks the (valid part, row split) units u, u + gridDim.x, ...
  const int U = vlist[0] * splits;
  int mnext = blockIdx.x < U ? vlist[4 + blockIdx.x / splits] : 0;
  for (int unit = blockIdx.x; unit < U; unit += gridDim.x) {
  m = mnext;
  {
    const int un = unit + gridDim.x;
    mnext = un < U ? vlist[4 + un / splits] : 0;  // needed one unit from now
  }
  const int sp = unit % splits, ob = m * splits + sp;  // ob: the unit's row of `partial`
  const int t_begin = (int)((long long)sp * TB / splits), t_end = (int)((long long)(sp + 1) * TB / splits);
  if constexpr (TOP) {
    const int T1 = (N + 31) / 32 + 1;
    for (int i = threadIdx.x; i < F; i += kT) {
      s_row[i] = erow[(long long)m * F + i];  // slots past the part's entry count hold garbage, never addressed
      s_ch[i] = ech[(long long)m * F + i];
      s_val[i] = eval[(long long)m * F + i];
    }
    for (int i = threadIdx.x; i < T1 && i < kMaxT1; i += kT) s_ptr[i] = tptr[(long long)m * T1 + i];
    // visible after the first barrier of the tile loop
  }
  float s1[NT], s2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0.0f;
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    float* cur = buf[(tile - t_begin) & 1];
    stash(tile, cur);
    __syncthreads();  // also orders this buffer's reuse: its previous readers finished before the last barrier
    if (tile + 1 < t_end) fetch(tile + 1);  // in flight during the MFMA chain below
    const int r0 = tile * RB + rt * 32;
    // Yprev of the epilogue (row of register r, columns d0 + 32t + j); rows past the end read row N-1, unused
    float yp[NT][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gn = r0 + acc_row(r, h);
      const long long o = ((long long)m * N + (gn < N ? gn : N - 1)) * cin + d0 + j;
#pragma unroll
      for (int t = 0; t < NT; ++t) yp[t][r] = y_prev[o + 32 * t];
    }
    float sa[TOP ? kSP : 1], sbv[TOP ? kSP : 1][NT];  // TOP: the first kSP sparse steps (2 entries each)
    int pb = 0, pe = 0;
    if constexpr (TOP) {
      const int st = r0 >> 5;  // this wave's 32-row tile
      pb = s_ptr[st];
      pe = s_ptr[st + 1];
#pragma unroll
      for (int q = 0; q < kSP; ++q) {
        const int ee = pb + 2 * q + h;
        const bool okk = ee < pe;
        const int es = okk ? ee : 0;
        sa[q] = (okk && s_row[es] - r0 == j) ? s_val[es] : 0.0f;
        const float* wrow = w5 + (long long)(okk ? s_ch[es] : 0) * cin + d0 + j;
#pragma unroll
        for (int t = 0; t < NT; ++t) sbv[q][t] = wrow[32 * t];
      }
    }
    const float4* frag = reinterpret_cast<const float4*>(cur + (rt * 32 + j) * LD + h * KH);
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) {
      const float4 a = frag[v];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw[t][4 * v + 0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw[t][4 * v + 1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw[t][4 * v + 2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw[t][4 * v + 3], acc[t], 0, 0, 0);
      }
    }
    if constexpr (TOP) {  // + S W5: the entries whose arg-max row lies in this tile, two per MFMA
#pragma unroll
      for (int q = 0; q < kSP; ++q) {
        if (pb + 2 * q < pe) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[q], sbv[q][t], acc[t], 0, 0, 0);
        }
      }
      for (int e = pb + 2 * kSP; e < pe; e += 2) {  // unusually crowded tile
        const int ee = e + h;
        const bool okk = ee < pe;
        const int es = okk ? ee : 0;
        const float a = (okk && s_row[es] - r0 == j) ? s_val[es] : 0.0f;
        const float* wrow = w5 + (long long)(okk ? s_ch[es] : 0) * cin + d0 + j;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wrow[32 * t], acc[t], 0, 0, 0);
      }
    }
    const bool full = r0 + 32 <= N;  // wave-uniform: only a part's last tile is ragged
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gn = r0 + acc_row(r, h);
      const bool ok = full || gn < N;
      const long long o = ((long long)m * N + gn) * cin + d0 + j;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float zz = __builtin_fmaf(yp[t][r], scp[t], shp[t]);
        const float d = (ok && zz > 0.0f) ? acc[t][r] + c0v[t] : 0.0f;
        if (ok) dz_prev[o + 32 * t] = d;
        s1[t] += d;
        s2[t] = __builtin_fmaf(d, (yp[t][r] - mnp[t]) * isp[t], s2[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    s1[t] += __shfl_xor(s1[t], 32, 64);
    s2[t] += __shfl_xor(s2[t], 32, 64);
  }
  if (h == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      red[wave][32 * t + j][0] = s1[t];
      red[wave][32 * t + j][1] = s2[t];
    }
  }
  __syncthreads();
  if (threadIdx.x < CW * PANELS) {  // thread -> (panel, channel); the RT waves of the panel in fixed order
    const int pn = threadIdx.x / CW, ch = threadIdx.x % CW;
    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      t0 += red[q * PANELS + pn][ch][0];
      t1 += red[q * PANELS + pn][ch][1];
    }
    const long long o = ((long long)ob * cin + db + threadIdx.x) * 2;
    partial[o] = t0;
    partial[o + 1] = t1;
  }
  __syncthreads();  // LDS (CSR copy, reduction scratch) is free again before the next unit rewrites it
  }  // unit
}

// The same input gradient of the never-stored last layer (dA = A Q + c0 + S W5, TOP form above) with the dense product
// A Q on the bf16 matrix cores, fp32-grade: both operands split into three bf16 terms, six products (see
// pn_fwd_split_kernel); the sparse S W5 steps stay exact-fp32 MFMAs on the same accumulator.
template <int K>
__global__ __launch_bounds__(kT, 2) void pn_dgrad_split_kernel(
    const float* __restrict__ y, const float* __restrict__ dz, const float* __restrict__ coef,
    const float* __restrict__ w, int cin, const float* __restrict__ y_prev, const float* __restrict__ bn_prev,
    const int* __restrict__ vlist, int N, int splits, float* __restrict__ dz_prev,
    float* __restrict__ partial, const int* __restrict__ erow, const int* __restrict__ ech,
    const float* __restrict__ eval, const int* __restrict__ tptr, const float* __restrict__ w5, int F) {
  constexpr int NT = 1, PANELS = 4;    // a wave owns 32 output channels, the four waves share one 32-row tile
  constexpr bool TOP = true;
  constexpr int KS = K / 16;           // bf16 MFMA k-steps
  constexpr int ROWB = 3 * K * 2 + 16; // LDS row: h | m | l planes of K bf16 each + pad (an odd multiple of 16 bytes)
  constexpr int Q4 = K / 4;
  constexpr int RT = 4 / PANELS;       // 32-row sub-tiles per block tile
  constexpr int RB = 32 * RT;          // rows per block tile
  constexpr int NLD = RB * Q4 / kT;    // float4 per thread, tensor and tile
  constexpr int CW = 32 * NT;          // output channels per wave
  __shared__ __attribute__((aligned(16))) unsigned char buf[2][RB * ROWB];
  __shared__ float red[kT / 64][CW][2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int panel = wave % PANELS, rt = wave / PANELS;
  const int db = blockIdx.y * CW * PANELS, d0 = db + panel * CW;
  // B fragments, split: lane (j, h) holds, per k-step, Q[k = 16 ks + 8 h .. + 7][d0 + j] as h / m / l
  pn_bf16x8 bh[KS], bm[KS], bl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float f = w[(long long)(16 * ks + 8 * h + u) * cin + d0 + j];
      bh[ks][u] = (__bf16)f;
      const float r1 = f - (float)bh[ks][u];
      bm[ks][u] = (__bf16)r1;
      bl[ks][u] = (__bf16)(r1 - (float)bm[ks][u]);
    }
  float scp[NT], shp[NT], mnp[NT], isp[NT], c0v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ci = d0 + 32 * t + j;
    scp[t] = bn_prev[ci];
    shp[t] = bn_prev[cin + ci];
    mnp[t] = bn_prev[2 * cin + ci];
    isp[t] = bn_prev[3 * cin + ci];
    c0v[t] = TOP ? w[(long long)K * cin + ci] : 0.0f;
  }
  // staging role of this thread: float4 column c4 of rows rl0, rl0 + kT/Q4, ...; its per-column tables.
  // TOP: the staged operand is relu(bn_prev(Yprev)), tables = scale, shift; else alpha, gammap, betap.
  const int c4 = threadIdx.x % Q4, rl0 = threadIdx.x / Q4;
  const float4 ta = reinterpret_cast<const float4*>(TOP ? bn_prev : coef)[c4];
  const float4 tb = reinterpret_cast<const float4*>(TOP ? bn_prev + K : coef + K)[c4];
  const float4 tc = TOP ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : reinterpret_cast<const float4*>(coef + 2 * K)[c4];
  const int TB = (N + RB - 1) / RB;
  int m = 0;
  // TOP: the part's CSR (<= 256 entries, <= 1025 tile offsets) lives in LDS so that the per-tile sparse chain has
  // a single level of global loads (the W5 rows), issued before the tile's main MFMA chain
  constexpr int kMaxF = 256, kMaxT1 = 1032, kSP = 6;  // tile offsets: N <= 32768 points per part
  __shared__ int s_row[TOP ? kMaxF : 1], s_ch[TOP ? kMaxF : 1], s_ptr[TOP ? kMaxT1 : 1];
  __shared__ float s_val[TOP ? kMaxF : 1];
  float4 ry[NLD], rz[TOP ? 1 : NLD];
  auto fetch = [&](int tile) {
    const long long base = ((long long)m * N + (long long)tile * RB) * Q4;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int rl = rl0 + i * (kT / Q4);
      const bool ok = tile * RB + rl < N;
      const long long o = base + i * kT + threadIdx.x;
      if constexpr (TOP) {
        ry[i] = ok ? reinterpret_cast<const float4*>(y_prev)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      } else {
        ry[i] = ok ? reinterpret_cast<const float4*>(y)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        rz[i] = ok ? reinterpret_cast<const float4*>(dz)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
  };
  auto stash = [&](int tile, unsigned char* dst) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int rl = rl0 + i * (kT / Q4);
      float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (tile * RB + rl < N) {  // rows past the part's end enter the MFMA as zeros
        v[0] = __builtin_fmaxf(__builtin_fmaf(ry[i].x, ta.x, tb.x), 0.0f);
        v[1] = __builtin_fmaxf(__builtin_fmaf(ry[i].y, ta.y, tb.y), 0.0f);
        v[2] = __builtin_fmaxf(__builtin_fmaf(ry[i].z, ta.z, tb.z), 0.0f);
        v[3] = __builtin_fmaxf(__builtin_fmaf(ry[i].w, ta.w, tb.w), 0.0f);
      }
      pn_bf16x4 ph, pm, pl;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ph[u] = (__bf16)v[u];
        const float r1 = v[u] - (float)ph[u];
        pm[u] = (__bf16)r1;
        pl[u] = (__bf16)(r1 - (float)pm[u]);
      }
      unsigned char* p = dst + rl * ROWB + 8 * c4;
      *reinterpret_cast<pn_bf16x4*>(p) = ph;
      *reinterpret_cast<pn_bf16x4*>(p + 2 * K) = pm;
      *reinterpret_cast<pn_bf16x4*>(p + 4 * K) = pl;
    }
  };
  // persistent: the block keeps its weight slab and walks the (valid part, row split) units u, u + gridDim.x, ...
  const int U = vlist[0] * splits;
  int mnext = blockIdx.x < U ? vlist[4 + blockIdx.x / splits] : 0;
  for (int unit = blockIdx.x; unit < U; unit += gridDim.x) {
  m = mnext;
  {
    const int un = unit + gridDim.x;
    mnext = un < U ? vlist[4 + un / splits] : 0;  // needed one unit from now
  }
  const int sp = unit % splits, ob = m * splits + sp;  // ob: the unit's row of `partial`
  const int t_begin = (int)((long long)sp * TB / splits), t_end = (int)((long long)(sp + 1) * TB / splits);
  if constexpr (TOP) {
    const int T1 = (N + 31) / 32 + 1;
    for (int i = threadIdx.x; i < F; i += kT) {
      s_row[i] = erow[(long long)m * F + i];  // slots past the part's entry count hold garbage, never addressed
      s_ch[i] = ech[(long long)m * F + i];
      s_val[i] = eval[(long long)m * F + i];
    }
    for (int i = threadIdx.x; i < T1 && i < kMaxT1; i += kT) s_ptr[i] = tptr[(long long)m * T1 + i];
    // visible after the first barrier of the tile loop
  }
  float s1[NT], s2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0.0f;
  if (t_begin < t_end) fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    unsigned char* cur = buf[(tile - t_begin) & 1];
    stash(tile, cur);
    __syncthreads();  // also orders this buffer's reuse: its previous readers finished before the last barrier
    if (tile + 1 < t_end) fetch(tile + 1);  // in flight during the MFMA chain below
    const int r0 = tile * RB + rt * 32;
    // Yprev of the epilogue (row of register r, columns d0 + 32t + j); rows past the end read row N-1, unused
    float yp[NT][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gn = r0 + acc_row(r, h);
      const long long o = ((long long)m * N + (gn < N ? gn : N - 1)) * cin + d0 + j;
#pragma unroll
      for (int t = 0; t < NT; ++t) yp[t][r] = y_prev[o + 32 * t];
    }
    float sa[TOP ? kSP : 1], sbv[TOP ? kSP : 1][NT];  // TOP: the first kSP sparse steps (2 entries each)
    int pb = 0, pe = 0;
    if constexpr (TOP) {
      const int st = r0 >> 5;  // this wave's 32-row tile
      pb = s_ptr[st];
      pe = s_ptr[st + 1];
#pragma unroll
      for (int q = 0; q < kSP; ++q) {
        const int ee = pb + 2 * q + h;
        const bool okk = ee < pe;
        const int es = okk ? ee : 0;
        sa[q] = (okk && s_row[es] - r0 == j) ? s_val[es] : 0.0f;
        const float* wrow = w5 + (long long)(okk ? s_ch[es] : 0) * cin + d0 + j;
#pragma unroll
        for (int t = 0; t < NT; ++t) sbv[q][t] = wrow[32 * t];
      }
    }
    const unsigned char* arow = cur + (rt * 32 + j) * ROWB + 16 * h;
    f32x16 acc[NT];
    acc[0] = f32x16{0};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const pn_bf16x8 ah = *reinterpret_cast<const pn_bf16x8*>(arow + 32 * ks);
      const pn_bf16x8 am = *reinterpret_cast<const pn_bf16x8*>(arow + 2 * K + 32 * ks);
      const pn_bf16x8 al = *reinterpret_cast<const pn_bf16x8*>(arow + 4 * K + 32 * ks);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ks], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[ks], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[ks], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[ks], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ks], acc[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ks], acc[0], 0, 0, 0);
    }
    if constexpr (TOP) {  // + S W5: the entries whose arg-max row lies in this tile, two per MFMA
#pragma unroll
      for (int q = 0; q < kSP; ++q) {
        if (pb + 2 * q < pe) {
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sa[q], sbv[q][t], acc[t], 0, 0, 0);
        }
      }
      for (int e = pb + 2 * kSP; e < pe; e += 2) {  // unusually crowded tile
        const int ee = e + h;
        const bool okk = ee < pe;
        const int es = okk ? ee : 0;
        const float a = (okk && s_row[es] - r0 == j) ? s_val[es] : 0.0f;
        const float* wrow = w5 + (long long)(okk ? s_ch[es] : 0) * cin + d0 + j;
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wrow[32 * t], acc[t], 0, 0, 0);
      }
    }
    const bool full = r0 + 32 <= N;  // wave-uniform: only a part's last tile is ragged
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int gn = r0 + acc_row(r, h);
      const bool ok = full || gn < N;
      const long long o = ((long long)m * N + gn) * cin + d0 + j;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float zz = __builtin_fmaf(yp[t][r], scp[t], shp[t]);
        const float d = (ok && zz > 0.0f) ? acc[t][r] + c0v[t] : 0.0f;
        if (ok) dz_prev[o + 32 * t] = d;
        s1[t] += d;
        s2[t] = __builtin_fmaf(d, (yp[t][r] - mnp[t]) * isp[t], s2[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    s1[t] += __shfl_xor(s1[t], 32, 64);
    s2[t] += __shfl_xor(s2[t], 32, 64);
  }
  if (h == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      red[wave][32 * t + j][0] = s1[t];
      red[wave][32 * t + j][1] = s2[t];
    }
  }
  __syncthreads();
  if (threadIdx.x < CW * PANELS) {  // thread -> (panel, channel); the RT waves of the panel in fixed order
    const int pn = threadIdx.x / CW, ch = threadIdx.x % CW;
    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      t0 += red[q * PANELS + pn][ch][0];
      t1 += red[q * PANELS + pn][ch][1];
    }
    const long long o = ((long long)ob * cin + db + threadIdx.x) * 2;
    partial[o] = t0;
    partial[o + 1] = t1;
  }
  __syncthreads();  // LDS (CSR copy, reduction scratch) is free again before the next unit rewrites it
  }  // unit
}

// ---- MFMA weight gradient --------------------------------------------------------------------------------------
// dW[co][ci] = sum over all valid rows of dY[r,co] * A[r,ci]   (a GEMM whose K dimension is the point rows), with
// dY = alpha*dZ + gammap*Y + betap and A = relu(bn_prev(Yprev)) built on the fly.
// The work is cut into units of RB rows of one part; kWG persistent blocks take the units round-robin (units of
// padded parts are skipped).  Per unit the block builds the dY [RB x COUT] and A [RB x CINP] panels in LDS from
// coalesced 16-byte loads issued ONE UNIT AHEAD (double-buffered), and each wave accumulates its share of the
// (COUT/32) x (CINP/32) output tiles across all units of the block: MFMA A operand = dY^T (lane: channel co,
// k = the row pair 2s + lane half), B operand = A (lane: channel ci), both plain 4-byte LDS reads.  Every block
// leaves one partial dW (deterministic fixed-order sum in pn_wgrad_reduce_kernel).
//   WG_FIRST : A = the raw input points (3 columns, zero-padded to one 32-wide tile) — first layer.
//   WG_GRAM  : dY := A (COUT == CIN): the Gram matrix A^T A plus, in row COUT, the column sums of A — what the
//              weight gradient of the never-stored last layer needs (pn_top_wgrad_kernel).
#ifndef MPA_PN_WG
#define MPA_PN_WG 512
#endif
constexpr int kWG = MPA_PN_WG;  // persistent blocks (2 per CU)
enum { WG_NORMAL = 0, WG_FIRST = 1, WG_GRAM = 2 };  // (WG_NORMAL: layers 2-4, now inside pn_bwd_fused_kernel)

// LDY / co0: the block handles the COUT output channels starting at column co0 of a layer that is LDY wide
// (the 64 -> 128 layer runs as two 64-channel slices, which keeps the double-buffered panels at 64 KB).
template <int COUT, int CIN, int MODE, int LDY = COUT>
__global__ __launch_bounds__(kT) void pn_wgrad_mfma_kernel(
    const float* __restrict__ y, const float* __restrict__ dz, const float* __restrict__ coef,
    const float* __restrict__ y_prev, const float* __restrict__ bn_prev, const int* __restrict__ vlist, int N,
    float* __restrict__ dwpart, int co0, const float* __restrict__ wt1 = nullptr) {
  // WG_FIRST: `y` is not read — the layer's own output Y1 is recomputed from the points in y_prev (wt1 [3][64])
  constexpr int CINP = MODE == WG_FIRST ? 32 : CIN;       // width of the B panel
  constexpr int CT = COUT / 32, IT = CINP / 32, NTILE = CT * IT;
  // GRAM: the matrix is symmetric — only the 10 tiles on or above the diagonal of the 4 x 4 tile grid are computed
  // (3, 3, 2, 2 per wave instead of 4 each) and mirrored on output
  constexpr bool SYM = MODE == WG_GRAM && CT == 4 && IT == 4;
  constexpr int TPW = SYM ? 3 : (NTILE + 3) / 4;
  constexpr int RB = 64;
  constexpr int DYW = MODE == WG_GRAM ? 0 : COUT;         // the dY panel does not exist in GRAM mode
  constexpr int STAGE = RB * (DYW + CINP);
  constexpr int QO = COUT / 4, QI = CIN / 4;
  constexpr int NLO = MODE == WG_GRAM ? 1 : RB * QO / kT;      // float4 per thread: Y and dZ
  constexpr int NLI = MODE == WG_FIRST ? 1 : RB * QI / kT;     // float4 per thread: Yprev
  __shared__ __attribute__((aligned(16))) float buf[2][STAGE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int TB = (N + RB - 1) / RB, U = vlist[0] * TB;
  // output tile (row-of-tiles ct, column-of-tiles it) number i of this wave; -1: none
  auto tile_ct = [&](int i) {
    if constexpr (SYM) return i == 0 ? 0 : (i == 1 ? (wave == 3 ? 2 : 1) : (wave == 0 ? 2 : (wave == 1 ? 3 : -1)));
    else return wave + 4 * i < NTILE ? (wave + 4 * i) / IT : -1;
  };
  auto tile_it = [&](int i) {
    if constexpr (SYM) return i == 0 ? wave : (i == 1 ? (wave == 3 ? 2 : wave + 1) : 3);
    else return (wave + 4 * i) % IT;
  };
  // staging roles and per-column tables
  const int co4 = threadIdx.x % QO, ro0 = threadIdx.x / QO;
  const int ci4 = threadIdx.x % QI, ri0 = threadIdx.x / QI;
  float4 al = {}, gp = {}, bp = {}, sc = {}, sh = {};
  if constexpr (MODE != WG_GRAM) {
    al = reinterpret_cast<const float4*>(coef + co0)[co4];
    gp = reinterpret_cast<const float4*>(coef + LDY + co0)[co4];
    bp = reinterpret_cast<const float4*>(coef + 2 * LDY + co0)[co4];
  }
  if constexpr (MODE != WG_FIRST) {
    sc = reinterpret_cast<const float4*>(bn_prev)[ci4];
    sh = reinterpret_cast<const float4*>(bn_prev + CIN)[ci4];
  } else {  // the zero padding of the point panel (columns 3..31) is written once
    for (int i = threadIdx.x; i < 2 * RB * 32; i += kT) buf[i / (RB * 32)][RB * DYW + i % (RB * 32)] = 0.0f;
    __syncthreads();
  }
  float4 ry[NLO], rz[NLO], rp[NLI];
  float rpt = 0.0f;
  float4 w1a = {}, w1b = {}, w1c = {};
  if constexpr (MODE == WG_FIRST) {
    w1a = reinterpret_cast<const float4*>(wt1)[co4];
    w1b = reinterpret_cast<const float4*>(wt1 + 64)[co4];
    w1c = reinterpret_cast<const float4*>(wt1 + 128)[co4];
  }
  auto fetch = [&](int u, int m) {
    const int n0 = (u % TB) * RB;
    const long long row0 = (long long)m * N + n0;
    if constexpr (MODE != WG_GRAM) {
#pragma unroll
      for (int i = 0; i < NLO; ++i) {
        const int rl = ro0 + i * (kT / QO);
        const bool ok = n0 + rl < N;
        const long long o = ((row0 + rl) * LDY + co0) / 4 + co4;
        if constexpr (MODE == WG_FIRST) {
          const float* p = y_prev + (row0 + (ok ? rl : 0)) * 3;
          ry[i] = make_float4(p[0], p[1], p[2], 0.0f);
        } else {
          ry[i] = ok ? reinterpret_cast<const float4*>(y)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        rz[i] = ok ? reinterpret_cast<const float4*>(dz)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
    if constexpr (MODE == WG_FIRST) {
      const int rl = threadIdx.x / 3;
      rpt = (threadIdx.x < RB * 3 && n0 + rl < N) ? y_prev[row0 * 3 + threadIdx.x] : 0.0f;
    } else {
#pragma unroll
      for (int i = 0; i < NLI; ++i) {
        const bool ok = n0 + ri0 + i * (kT / QI) < N;
        rp[i] = ok ? reinterpret_cast<const float4*>(y_prev)[row0 * QI + i * kT + threadIdx.x]
                   : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
  };
  auto stash = [&](int u, float* dst) {  // rows past the part's end are staged as zeros (both panels)
    const int n0 = (u % TB) * RB;
    if constexpr (MODE != WG_GRAM) {
#pragma unroll
      for (int i = 0; i < NLO; ++i) {
        const int rl = ro0 + i * (kT / QO);
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if constexpr (MODE == WG_FIRST) {
          const float a0 = ry[i].x, a1 = ry[i].y, a2 = ry[i].z;
          ry[i] = make_float4(first_layer_y(a0, a1, a2, w1a.x, w1b.x, w1c.x), first_layer_y(a0, a1, a2, w1a.y, w1b.y, w1c.y),
                              first_layer_y(a0, a1, a2, w1a.z, w1b.z, w1c.z), first_layer_y(a0, a1, a2, w1a.w, w1b.w, w1c.w));
        }
        if (n0 + rl < N) {
          v.x = __builtin_fmaf(al.x, rz[i].x, __builtin_fmaf(gp.x, ry[i].x, bp.x));
          v.y = __builtin_fmaf(al.y, rz[i].y, __builtin_fmaf(gp.y, ry[i].y, bp.y));
          v.z = __builtin_fmaf(al.z, rz[i].z, __builtin_fmaf(gp.z, ry[i].z, bp.z));
          v.w = __builtin_fmaf(al.w, rz[i].w, __builtin_fmaf(gp.w, ry[i].w, bp.w));
        }
        *reinterpret_cast<float4*>(dst + rl * COUT + 4 * co4) = v;
      }
    }
    float* da = dst + RB * DYW;
    if constexpr (MODE == WG_FIRST) {
      if (threadIdx.x < RB * 3) da[(threadIdx.x / 3) * 32 + threadIdx.x % 3] = rpt;
    } else {
#pragma unroll
      for (int i = 0; i < NLI; ++i) {
        const int rl = ri0 + i * (kT / QI);
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (n0 + rl < N) {
          v.x = __builtin_fmaxf(__builtin_fmaf(rp[i].x, sc.x, sh.x), 0.0f);
          v.y = __builtin_fmaxf(__builtin_fmaf(rp[i].y, sc.y, sh.y), 0.0f);
          v.z = __builtin_fmaxf(__builtin_fmaf(rp[i].z, sc.z, sh.z), 0.0f);
          v.w = __builtin_fmaxf(__builtin_fmaf(rp[i].w, sc.w, sh.w), 0.0f);
        }
        *reinterpret_cast<float4*>(da + rl * CINP + 4 * ci4) = v;
      }
    }
  };
  // units u, u + kWG, ... of the valid parts; part ids are looked up two iterations ahead of their use
  auto part_of = [&](int uu) { return uu < U ? vlist[4 + uu / TB] : 0; };
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) acc[i] = f32x16{0};
  float bsum = 0.0f;
  int u = blockIdx.x, un = u + kWG, k = 0;
  int m = part_of(u), mn = part_of(un);
  if (u < U) fetch(u, m);
  while (u < U) {
    float* cur = buf[k];
    stash(u, cur);
    __syncthreads();  // also orders the reuse of this buffer (its readers passed the previous barrier)
    const int unn = un + kWG, mnn = part_of(unn);
    if (un < U) fetch(un, mn);  // in flight during the MFMAs below
    const float* pa = cur + h * (MODE == WG_GRAM ? CINP : COUT) + j;
    const float* pb = cur + RB * DYW + h * CINP + j;
    constexpr int AW = MODE == WG_GRAM ? CINP : COUT;  // row stride of the A-operand panel
#pragma unroll 4
    for (int s = 0; s < RB / 2; ++s) {
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int tct = tile_ct(i), tit = tile_it(i);
        if (tct >= 0) {
          const float a = pa[2 * s * AW + tct * 32];
          const float b = pb[2 * s * CINP + tit * 32];
          if (MODE == WG_GRAM && i == 0) bsum += b;  // tile row 0: its B operand covers columns 32*wave + j
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
      }
    }
    u = un;
    m = mn;
    un = unn;
    mn = mnn;
    k ^= 1;
  }
  constexpr int ELEMS = MODE == WG_FIRST ? COUT * 3 : COUT * CIN + (MODE == WG_GRAM ? CIN : 0);
  float* out = dwpart + (long long)blockIdx.x * ELEMS;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tct = tile_ct(i), tit = tile_it(i);
    if (tct >= 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = tct * 32 + acc_row(r, h), ci = tit * 32 + j;
        if constexpr (MODE == WG_FIRST) {
          if (j < 3) out[co * 3 + j] = acc[i][r];
        } else {
          out[co * CIN + ci] = acc[i][r];
          if (SYM && tct != tit) out[ci * CIN + co] = acc[i][r];  // the mirrored tile
        }
      }
    }
  }
  if constexpr (MODE == WG_GRAM) {
    bsum += __shfl_xor(bsum, 32, 64);
    if (h == 0) out[COUT * CIN + wave * 32 + j] = bsum;
  }
}

// ---- Gram matrix of the last hidden layer on the bf16 matrix cores, fp32-grade ---------------------------------------------
// G = A^T A and the column sums of A = relu(bn_prev(Yprev)) over all valid rows (WG_GRAM above; what the weight gradient of
// the never-stored last layer needs), with the split products of pn_fwd_split_kernel.  The MFMA reduction index is the
// point row, so the operand is TRANSPOSED while it is staged: a thread's float4 (one row, four channels) becomes 3 x 4
// two-byte stores into the channel rows [channel][h | m | l planes of 64 rows] of the panel (rows of 384 + 16 bytes; the
// eight 8-row groups of a plane are XOR-swizzled by (channel >> 2) & 7 so that an instruction's stores spread over the
// banks while a fragment — 8 consecutive rows — stays one aligned 16-byte read).  Same units, persistent blocks, symmetric
// tile assignment (3, 3, 2, 2 tiles per wave, mirrored on output) and output layout as pn_wgrad_mfma_kernel<.., WG_GRAM>.
template <int CIN>
__global__ __launch_bounds__(kT, 2) void pn_gram_split_kernel(const float* __restrict__ y_prev,
                                                              const float* __restrict__ bn_prev,
                                                              const int* __restrict__ vlist, int N,
                                                              float* __restrict__ dwpart) {
  static_assert(CIN == 128, "the symmetric tile assignment is written for a 4 x 4 tile grid");
  constexpr int RB = 64, QI = CIN / 4, NLI = RB * QI / kT, ROWB = 3 * RB * 2 + 16, KS = RB / 16;
  __shared__ __attribute__((aligned(16))) unsigned char pan[CIN * ROWB];
  __shared__ float csum[kT / QI][CIN];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int TB = (N + RB - 1) / RB, U = vlist[0] * TB;
  auto tile_ct = [&](int i) { return i == 0 ? 0 : (i == 1 ? (wave == 3 ? 2 : 1) : (wave == 0 ? 2 : (wave == 1 ? 3 : -1))); };
  auto tile_it = [&](int i) { return i == 0 ? wave : (i == 1 ? (wave == 3 ? 2 : wave + 1) : 3); };
  const int ci4 = threadIdx.x % QI, ri0 = threadIdx.x / QI;
  const float4 sc = reinterpret_cast<const float4*>(bn_prev)[ci4];
  const float4 sh = reinterpret_cast<const float4*>(bn_prev + CIN)[ci4];
  float4 rp[NLI];
  auto fetch = [&](int u, int m) {
    const int n0 = (u % TB) * RB;
    const long long row0 = (long long)m * N + n0;
#pragma unroll
    for (int i = 0; i < NLI; ++i) {
      const int rl = ri0 + i * (kT / QI);
      const long long rr = n0 + rl < N ? row0 + rl : (long long)m * N + N - 1;  // clamped: the value is dropped below
      rp[i] = reinterpret_cast<const float4*>(y_prev)[rr * QI + ci4];
    }
  };
  float4 colsum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // this thread's four channels over its rows of all units
  auto stash = [&](int u) {
    const int n0 = (u % TB) * RB;
#pragma unroll
    for (int i = 0; i < NLI; ++i) {
      const int rl = ri0 + i * (kT / QI);
      float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      if (n0 + rl < N) {  // rows past the part's end are staged as zeros
        v[0] = __builtin_fmaxf(__builtin_fmaf(rp[i].x, sc.x, sh.x), 0.0f);
        v[1] = __builtin_fmaxf(__builtin_fmaf(rp[i].y, sc.y, sh.y), 0.0f);
        v[2] = __builtin_fmaxf(__builtin_fmaf(rp[i].z, sc.z, sh.z), 0.0f);
        v[3] = __builtin_fmaxf(__builtin_fmaf(rp[i].w, sc.w, sh.w), 0.0f);
      }
      colsum.x += v[0];
      colsum.y += v[1];
      colsum.z += v[2];
      colsum.w += v[3];
      const int slot = 16 * ((rl >> 3) ^ (ci4 & 7)) + 2 * (rl & 7);  // (channel >> 2) & 7 == ci4 & 7
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 bh = (__bf16)v[e];
        const float r1 = v[e] - (float)bh;
        const __bf16 bm = (__bf16)r1;
        const __bf16 bl = (__bf16)(r1 - (float)bm);
        unsigned char* p = pan + (4 * ci4 + e) * ROWB + slot;
        *reinterpret_cast<__bf16*>(p) = bh;
        *reinterpret_cast<__bf16*>(p + 2 * RB) = bm;
        *reinterpret_cast<__bf16*>(p + 4 * RB) = bl;
      }
    }
  };
  auto frag = [&](int ch, int ks, pn_bf16x8& fh, pn_bf16x8& fm, pn_bf16x8& fl) {  // rows 16 ks + 8 h .. + 7 of channel ch
    const unsigned char* p = pan + ch * ROWB + 16 * ((2 * ks + h) ^ ((ch >> 2) & 7));
    fh = *reinterpret_cast<const pn_bf16x8*>(p);
    fm = *reinterpret_cast<const pn_bf16x8*>(p + 2 * RB);
    fl = *reinterpret_cast<const pn_bf16x8*>(p + 4 * RB);
  };
  auto part_of = [&](int uu) { return uu < U ? vlist[4 + uu / TB] : 0; };
  f32x16 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = f32x16{0};
  int u = blockIdx.x, un = u + kWG;
  int m = part_of(u), mn = part_of(un);
  if (u < U) fetch(u, m);
  while (u < U) {
    stash(u);
    __syncthreads();
    const int unn = un + kWG, mnn = part_of(unn);
    if (un < U) fetch(un, mn);  // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int tct = tile_ct(i), tit = tile_it(i);
        if (tct >= 0) {  // wave-uniform
          pn_bf16x8 ah, am, al, bh, bm, bl;
          frag(tct * 32 + j, ks, ah, am, al);
          frag(tit * 32 + j, ks, bh, bm, bl);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // single panel: the fragment reads are done before the next unit is staged
    u = un;
    m = mn;
    un = unn;
    mn = mnn;
  }
  float* out = dwpart + (long long)blockIdx.x * (CIN * CIN + CIN);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int tct = tile_ct(i), tit = tile_it(i);
    if (tct >= 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = tct * 32 + acc_row(r, h), ci = tit * 32 + j;
        out[co * CIN + ci] = acc[i][r];
        if (tct != tit) out[ci * CIN + co] = acc[i][r];  // the mirrored tile
      }
    }
  }
  // column sums: the kT / QI threads that share four channels, in a fixed order
  *reinterpret_cast<float4*>(&csum[ri0][4 * ci4]) = colsum;
  __syncthreads();
  if (threadIdx.x < CIN) {
    float t = 0.0f;
#pragma unroll
    for (int q = 0; q < kT / QI; ++q) t += csum[q][threadIdx.x];
    out[CIN * CIN + threadIdx.x] = t;
  }
}

// ---- fused input + weight gradient (64 -> 64 and 64 -> 128 layers) -------------------------------------------------
// Both gradients of a layer consume the same dY = alpha*dZ + gammap*Y + betap tile, and the weight gradient's other
// operand relu(bn_prev(Yprev)) is the tensor the input gradient's epilogue masks with: one kernel reads Y, dZ and
// Yprev ONCE (separate kernels read each of them twice).  2*kWF persistent blocks of NTH threads (two per CU: what
// the double-buffered panels and 256 registers per lane allow) take RB-row units of the valid parts round-robin;
// all threads stage the dY [RB x K] and raw Yprev [RB x 64] panels one unit ahead; the first half of the waves then
// runs the input-gradient MFMA chain (weights register-resident, dZprev = (dY W) masked by bn_prev(Yprev) > 0,
// BatchNorm-backward sums), the second half the weight-gradient chain (output tiles accumulate across the block's
// units) — the same MFMA count per unit on either side.  Per block: one (sum, sum) row of `partial` and one
// partial dW, reduced in fixed order by pn_bwd_coef_kernel / pn_wgrad_reduce_kernel.
// Measured (352 valid parts x 1000 points): 106 us for 64 -> 64 (separate kernels: 174), 153 us for 64 -> 128 (293);
// the fp32 MFMA floor of the two GEMMs is 37 / 74 us, the HBM floor 58 / 86 us.
#ifndef MPA_PN_WF
#define MPA_PN_WF 256
#endif
constexpr int kWF = MPA_PN_WF;
// FIRST (layer 2): `y_prev` holds the raw points [rows][3]; the Y1 panel is recomputed from them (wt1 [3][64]).
template <int K, int NT, int PANELS, int NTH, bool FIRST = false>
__global__ __launch_bounds__(NTH, 2) void pn_bwd_fused_kernel(
    const float* __restrict__ y, const float* __restrict__ dz, const float* __restrict__ coef,
    const float* __restrict__ w, const float* __restrict__ y_prev, const float* __restrict__ bn_prev,
    const int* __restrict__ vlist, int N, float* __restrict__ dz_prev, float* __restrict__ partial,
    float* __restrict__ dwpart, const float* __restrict__ wt1 = nullptr) {
  constexpr int CIN = 64, KH = K / 2, LDY = K + 4, LDP = CIN + 4, QK = K / 4, QC = CIN / 4;
  constexpr int ND = NTH / 128;  // waves of each kind
  constexpr int RT = ND / PANELS, RB = 32 * RT, CW = 32 * NT;
  constexpr int NLY = RB * QK / NTH, NLP = RB * QC / NTH;  // float4 per thread and unit: Y / dZ, Yprev
  constexpr int IT = CIN / 32, NTILE = (K / 32) * IT, TPW = NTILE / ND;
  static_assert(CW * PANELS == CIN && NTILE % ND == 0 && RT >= 1, "tile shapes");
  __shared__ __attribute__((aligned(16))) float bufY[2][RB * LDY];
  __shared__ __attribute__((aligned(16))) float bufP[2][RB * LDP];
  __shared__ float red[ND][CW][2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const bool dwave = wave < ND;
  const int wv = wave % ND, panel = wv % PANELS, rt = wv / PANELS, d0 = panel * CW;
  const int TB = (N + RB - 1) / RB, U = vlist[0] * TB, G = gridDim.x;
  // staging roles and their per-column tables
  const int c4 = threadIdx.x % QK, rl0 = threadIdx.x / QK, p4 = threadIdx.x % QC, rp0 = threadIdx.x / QC;
  const float4 ta = reinterpret_cast<const float4*>(coef)[c4];
  const float4 tb = reinterpret_cast<const float4*>(coef + K)[c4];
  const float4 tc = reinterpret_cast<const float4*>(coef + 2 * K)[c4];
  // input-gradient waves: B fragments (lane-half h, step s <-> k = h*KH + s) and the epilogue's channel tables
  // (the two kinds of waves keep their loop-carried registers in the same array R: weights here, accumulators there)
  constexpr int NR = NT * KH / 16 > TPW ? NT * KH / 16 : TPW;
  f32x16 R[NR];
  float scp[NT], shp[NT], mnp[NT], isp[NT];
  if (dwave) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int s = 0; s < KH; ++s)
        R[(t * KH + s) / 16][(t * KH + s) % 16] = w[(long long)(h * KH + s) * CIN + d0 + 32 * t + j];
      const int ci = d0 + 32 * t + j;
      scp[t] = bn_prev[ci];
      shp[t] = bn_prev[CIN + ci];
      mnp[t] = bn_prev[2 * CIN + ci];
      isp[t] = bn_prev[3 * CIN + ci];
    }
  }
  // weight-gradient waves: scale / shift of their B-operand channels
  float scw[IT], shw[IT];
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    scw[u] = bn_prev[32 * u + j];
    shw[u] = bn_prev[CIN + 32 * u + j];
  }
  float4 ry[NLY], rz[NLY], rp[NLP];
  // FIRST: the first layer's weights sit in LDS and are read where they are used (12 more live registers per lane would
  // spill: the kernel runs at the 256-register limit of two blocks per CU)
  __shared__ __attribute__((aligned(16))) float w1s[FIRST ? 192 : 4];
  if constexpr (FIRST) {
    if (threadIdx.x < 192) w1s[threadIdx.x] = wt1[threadIdx.x];
    __syncthreads();  // the FIRST unit's stash reads them before the loop's own barrier (without this: a race that
                      // tools/exp_race_hunt.py caught as one diverging step in ~300 — wrong Y1 rows in a block's first unit)
  }
  auto fetch = [&](int u, int m) {
    const int n0 = (u % TB) * RB;
    const long long row0 = (long long)m * N + n0;
#pragma unroll
    for (int i = 0; i < NLY; ++i) {
      const bool ok = n0 + rl0 + i * (NTH / QK) < N;
      const long long o = row0 * QK + i * NTH + threadIdx.x;
      ry[i] = ok ? reinterpret_cast<const float4*>(y)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      rz[i] = ok ? reinterpret_cast<const float4*>(dz)[o] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
      const bool ok = n0 + rp0 + i * (NTH / QC) < N;
      if constexpr (FIRST) {
        const float* p = y_prev + (row0 + (ok ? rp0 + i * (NTH / QC) : 0)) * 3;
        rp[i] = make_float4(p[0], p[1], p[2], 0.0f);
      } else {
        rp[i] = ok ? reinterpret_cast<const float4*>(y_prev)[row0 * QC + i * NTH + threadIdx.x]
                   : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    }
  };
  auto stash = [&](int u, float* dy, float* dp) {  // rows past the part's end: dY = 0 (their Yprev is never used)
    const int n0 = (u % TB) * RB;
#pragma unroll
    for (int i = 0; i < NLY; ++i) {
      const int rl = rl0 + i * (NTH / QK);
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (n0 + rl < N) {
        v.x = __builtin_fmaf(ta.x, rz[i].x, __builtin_fmaf(tb.x, ry[i].x, tc.x));
        v.y = __builtin_fmaf(ta.y, rz[i].y, __builtin_fmaf(tb.y, ry[i].y, tc.y));
        v.z = __builtin_fmaf(ta.z, rz[i].z, __builtin_fmaf(tb.z, ry[i].z, tc.z));
        v.w = __builtin_fmaf(ta.w, rz[i].w, __builtin_fmaf(tb.w, ry[i].w, tc.w));
      }
      *reinterpret_cast<float4*>(dy + rl * LDY + 4 * c4) = v;
    }
#pragma unroll
    for (int i = 0; i < NLP; ++i) {
      if constexpr (FIRST) {
        const float4 w1a = reinterpret_cast<const float4*>(w1s)[p4], w1b = reinterpret_cast<const float4*>(w1s + 64)[p4],
                     w1c = reinterpret_cast<const float4*>(w1s + 128)[p4];
        const float a0 = rp[i].x, a1 = rp[i].y, a2 = rp[i].z;
        rp[i] = make_float4(first_layer_y(a0, a1, a2, w1a.x, w1b.x, w1c.x), first_layer_y(a0, a1, a2, w1a.y, w1b.y, w1c.y),
                            first_layer_y(a0, a1, a2, w1a.z, w1b.z, w1c.z), first_layer_y(a0, a1, a2, w1a.w, w1b.w, w1c.w));
      }
      *reinterpret_cast<float4*>(dp + (rp0 + i * (NTH / QC)) * LDP + 4 * p4) = rp[i];
    }
  };
  float s1[NT], s2[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) s1[t] = s2[t] = 0.0f;
  if (!dwave) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) R[i] = f32x16{0};
  }
  // units u, u + G, ... of the valid parts; the part id of a unit is looked up two iterations before its rows are
  // requested, so no load latency sits between the barrier and the next unit's requests
  auto part_of = [&](int uu) { return uu < U ? vlist[4 + uu / TB] : 0; };
  int u = blockIdx.x, un = u + G, kb = 0;
  int m = part_of(u), mn = part_of(un);
  if (u < U) fetch(u, m);
  while (u < U) {
    float* cy = bufY[kb];
    float* cp = bufP[kb];
    stash(u, cy, cp);
    __syncthreads();  // also orders the reuse of these buffers (their readers passed the previous barrier)
    const int unn = un + G, mnn = part_of(unn);
    if (un < U) fetch(un, mn);  // in flight during the MFMA chains below
    if (dwave) {
      const int r0 = (u % TB) * RB + rt * 32;
      const float4* frag = reinterpret_cast<const float4*>(cy + (rt * 32 + j) * LDY + h * KH);
      f32x16 acc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
#pragma unroll
      for (int v = 0; v < KH / 4; ++v) {
        const float4 a = frag[v];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, R[(t * KH + 4 * v + 0) / 16][(t * KH + 4 * v + 0) % 16], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, R[(t * KH + 4 * v + 1) / 16][(t * KH + 4 * v + 1) % 16], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, R[(t * KH + 4 * v + 2) / 16][(t * KH + 4 * v + 2) % 16], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, R[(t * KH + 4 * v + 3) / 16][(t * KH + 4 * v + 3) % 16], acc[t], 0, 0, 0);
        }
      }
      const bool full = r0 + 32 <= N;  // wave-uniform: only a part's last tile is ragged
      // the Yprev values behind the ReLU mask and the BatchNorm sums are read from the panel TWO rows ahead of their use:
      // read where they are used, their 32 LDS round trips per unit stood exposed (12 of the kernel's 102 us by a
      // timing-only build without them; the registers for all 32 at once do not exist: the kernel runs at 256)
      constexpr int LA = 2;
      float ypq[LA][NT];
#pragma unroll
      for (int q = 0; q < LA; ++q)
#pragma unroll
        for (int t = 0; t < NT; ++t) ypq[q][t] = cp[(rt * 32 + acc_row(q, h)) * LDP + d0 + 32 * t + j];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gn = r0 + acc_row(r, h);
        const bool ok = full || gn < N;
        const long long o = ((long long)m * N + gn) * CIN + d0 + j;
        float ypr[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          ypr[t] = ypq[r % LA][t];
          if (r + LA < 16) ypq[r % LA][t] = cp[(rt * 32 + acc_row(r + LA, h)) * LDP + d0 + 32 * t + j];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float ypv = ypr[t];
          const float zz = __builtin_fmaf(ypv, scp[t], shp[t]);
          const float d = (ok && zz > 0.0f) ? acc[t][r] : 0.0f;
          if (ok) dz_prev[o + 32 * t] = d;
          s1[t] += d;
          s2[t] = __builtin_fmaf(d, (ypv - mnp[t]) * isp[t], s2[t]);
        }
      }
    } else {
      const float* pa = cy + h * LDY + j;
      const float* pb = cp + h * LDP + j;
#pragma unroll 4
      for (int s = 0; s < RB / 2; ++s) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int q = wv + ND * i, ct = q / IT, it = q % IT;
          const float a = pa[2 * s * LDY + ct * 32];
          const float b = __builtin_fmaxf(__builtin_fmaf(pb[2 * s * LDP + it * 32], scw[it], shw[it]), 0.0f);
          R[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, R[i], 0, 0, 0);
        }
      }
    }
    u = un;
    m = mn;
    un = unn;
    mn = mnn;
    kb ^= 1;
  }
  if (dwave) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s1[t] += __shfl_xor(s1[t], 32, 64);
      s2[t] += __shfl_xor(s2[t], 32, 64);
    }
    if (h == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        red[wv][32 * t + j][0] = s1[t];
        red[wv][32 * t + j][1] = s2[t];
      }
    }
  } else {
    float* out = dwpart + (long long)blockIdx.x * (K * CIN);
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
      const int q = wv + ND * i, ct = q / IT, it = q % IT;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(ct * 32 + acc_row(r, h)) * CIN + it * 32 + j] = R[i][r];
    }
  }
  __syncthreads();
  if (threadIdx.x < CIN) {  // thread -> (panel, channel); the RT waves of the panel in fixed order
    const int pn = threadIdx.x / CW, ch = threadIdx.x % CW;
    float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
    for (int q = 0; q < RT; ++q) {
      t0 += red[q * PANELS + pn][ch][0];
      t1 += red[q * PANELS + pn][ch][1];
    }
    const long long o = ((long long)blockIdx.x * CIN + threadIdx.x) * 2;
    partial[o] = t0;
    partial[o + 1] = t1;
  }
}

// dW[i] = sum over the blocks' partial dW (valids == nullptr) or over valid parts of dwpart[m][i].
// block 1024 = 64 elements x 16 slices.
__global__ __launch_bounds__(64 * kSlices) void pn_wgrad_reduce_kernel(const float* __restrict__ dwpart,
                                                                      const float* __restrict__ valids,
                                                                      int M, int elems,
                                                                      float* __restrict__ dw) {
  __shared__ float sm[kSlices][64];
  const int el = threadIdx.x & 63, slice = threadIdx.x >> 6, i = blockIdx.x * 64 + el;
  float s = 0.0f;
  if (i < elems) {
    constexpr int U = 8;  // independent loads in flight; padded parts' rows hold garbage and are skipped
    for (int m0 = slice; m0 < M; m0 += kSlices * U) {
      float v[U], ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = m0 + u * kSlices, mm = m < M ? m : M - 1;
        v[u] = dwpart[(long long)mm * elems + i];
        ok[u] = m < M ? (valids != nullptr ? valids[mm] : 1.0f) : 0.0f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (ok[u] != 0.0f) s += v[u];
    }
  }
  sm[slice][el] = s;
  __syncthreads();
  if (slice == 0 && i < elems) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < kSlices; ++k) t += sm[k][el];
    dw[i] = t;
  }
}

// The same reduction for up to four weight gradients in ONE launch (the layers' partial tables wait in their own regions of
// dwpart until the end of the backward pass): blocks [first[k], first[k + 1]) serve problem k.  Same summation order.
struct WgradReduceGroup {
  const float* part[4];
  float* dw[4];
  int rows[4], elems[4], first[5];
};
__global__ __launch_bounds__(64 * kSlices) void pn_wgrad_reduce_group_kernel(const WgradReduceGroup g) {
  __shared__ float sm[kSlices][64];
  int k = 0;
#pragma unroll
  for (int q = 1; q < 4; ++q) k += (int)blockIdx.x >= g.first[q] ? 1 : 0;
  const float* __restrict__ dwpart = g.part[k];
  const int M = g.rows[k], elems = g.elems[k];
  const int el = threadIdx.x & 63, slice = threadIdx.x >> 6, i = ((int)blockIdx.x - g.first[k]) * 64 + el;
  float s = 0.0f;
  if (i < elems) {
    constexpr int U = 8;
    for (int m0 = slice; m0 < M; m0 += kSlices * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = m0 + u * kSlices, mm = m < M ? m : M - 1;
        v[u] = dwpart[(long long)mm * elems + i];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (m0 + u * kSlices < M) s += v[u];
    }
  }
  sm[slice][el] = s;
  __syncthreads();
  if (slice == 0 && i < elems) {
    float t = 0.0f;
#pragma unroll
    for (int q = 0; q < kSlices; ++q) t += sm[q][el];
    g.dw[k][i] = t;
  }
}

#ifndef MPA_PN_QFORM  // 1: conv2..conv4 backward in Q form on the bf16 matrix cores (pn_bwd_q.h); 0: pn_bwd_fused_kernel
#define MPA_PN_QFORM 1
#endif
#include "pn_bwd_q.h"
#include "pn_fwd_ws.h"
#ifndef MPA_PN_FWD4_BLOCKS
#define MPA_PN_FWD4_BLOCKS 512
#endif
#ifndef MPA_PN_FWD_WS  // 1: conv2..conv4 forward wave-specialised on the bf16 matrix cores (pn_fwd_ws.h); 0: pn_fwd_mfma_kernel
#define MPA_PN_FWD_WS 1
#endif
constexpr int kQB = 512;  // persistent blocks of pn_bwd_q_kernel: at most two per CU (MPA_PN_QB2)
#ifndef MPA_PN_QB2  // 1: the 64 -> 64 layers as two 6-wave blocks per CU on 32-row units; 0: one 12-wave block on 64-row units
#define MPA_PN_QB2 0
#endif

// ---- host side ------------------------------------------------------------------------------------------------
struct Dims {
  int64_t M, N, F, rows;
  int splits, tiles1;  // row splits of the MFMA kernels; 256-row tiles of the first-layer kernel
  int splits_top;      // row splits of the last layer's forward GEMM (MFMA-bound: shorter blocks even out the tail)
  int splits_dtop;     // row splits of the last layer's input-gradient GEMM
  int C[6];            // channel widths: C[0] = 3 ... C[5] = F
};

Dims make_dims(int64_t M, int64_t N, int64_t F) {
  Dims d;
  d.M = M;
  d.N = N;
  d.F = F;
  d.rows = M * N;
  d.tiles1 = (int)((N + kT - 1) / kT);
  const int T = (int)((N + 31) / 32);
  d.splits = T >= 32 ? 4 : (T >= 16 ? 2 : 1);  // row splits per part: enough blocks for an even last round
  d.splits_top = T >= 32 ? 8 : d.splits;
  d.splits_dtop = d.splits;  // (a sweep over 2..16 moved the step by < 1 %: the kernels are not tail-bound)
  d.C[0] = 3;
  d.C[1] = 64;
  d.C[2] = 64;
  d.C[3] = 64;
  d.C[4] = 128;
  d.C[5] = (int)F;
  return d;
}

struct PnWs {
  float* Y[6];    // pre-BN outputs Y[1..4] (the last layer's output is never stored)
  float* dZ[6];   // backward: dZ[1..4]
  float* Wt1;     // transposed first-layer weights [3][64]
  float* bn[6];   // [4][C] scale, shift, mean, invstd
  float* coef[6]; // [3][C] alpha, gammap, betap
  float* partial; // per-block column sums
  float* dwpart;  // [kWG][cout*cin] per-block partial weight gradients (last layer: Gram matrix + column sums)
  float* count;
  CoopWs coop;    // fp64 group sums + tickets of the cooperative reductions
  float* topv;    // [M*splits][F][2] top-2 records of the last layer (values)
  float* ybest;   // [M][F] pre-BatchNorm value at the arg-max
  float* eval;    // [M][F] CSR values alpha*grad_feat
  float* q;       // [129][128] Q then c0
  float* gram;    // [129][128] Gram matrix then column sums of A4
  float* ql[5];   // Q form of conv2..conv4: [CIN + 1][CIN] Q then c0 of layer l
  float* red[5];  // ... and the layer's reduced tables [T | G | asum | (S, P^T P, psum)]
  int64_t total;
};

struct PnIws {
  int* argmax;  // [M][F]
  int* topn;    // [M*splits][F][2]
  int* erow;    // [M][F] CSR rows
  int* ech;     // [M][F] CSR channels
  int* tptr;    // [M][T+1] CSR tile offsets
  int* vlist;   // [4 + M] number of valid parts, then (from [4]) their ids
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
  w.Y[1] = nullptr;  // never stored: recomputed from the points by its consumers (pn_fwd_first_kernel)
  for (int l = 2; l <= 4; ++l) w.Y[l] = take(d.rows * d.C[l]);
  w.Y[5] = nullptr;
  for (int l = 1; l <= 4; ++l)  // (Q form: dZ1 never leaves the conv2 kernel)
    w.dZ[l] = (MPA_PN_QFORM && l == 1) ? nullptr : take(d.rows * d.C[l]);
  w.Wt1 = take(192);
  for (int l = 1; l <= 5; ++l) w.bn[l] = take(4LL * d.C[l]);
  for (int l = 1; l <= 5; ++l) w.coef[l] = take(4LL * d.C[l]);
  const int64_t maxc = d.F > 128 ? d.F : 128;
  int64_t smax = d.splits_top > d.splits ? d.splits_top : d.splits;
  smax = d.splits_dtop > smax ? d.splits_dtop : smax;
  int64_t blocks = d.M * (d.tiles1 > smax ? d.tiles1 : smax);
  if (blocks < 2 * kWF) blocks = 2 * kWF;  // the fused backward kernel leaves one row per persistent block
  w.partial = take(blocks * maxc * 2);
  // the Gram partials of the last layer (reduced at once), then — in the same storage — the partial tables of layers 4..1,
  // which wait for ONE grouped reduction at the end of the backward pass
  {
    const int64_t gram = (int64_t)kWG * (128 * 128 + 128);
    int64_t wait = (int64_t)2 * kWF * (128 * 64 + 64 * 64 + 64 * 64) + (int64_t)kWG * (64 * 4);
    if (MPA_PN_QFORM)
      wait = (int64_t)kQB * (pn_bwd_q_elems(128, 64, false) + pn_bwd_q_elems(64, 64, false) + pn_bwd_q_elems(64, 64, true));
    w.dwpart = take(gram > wait ? gram : wait);
  }
  for (int l = 2; l <= 4; ++l) {
    w.ql[l] = take((int64_t)(d.C[l - 1] + 1) * d.C[l - 1]);
    w.red[l] = take(pn_bwd_q_elems(d.C[l], d.C[l - 1], l == 2));
  }
  w.count = take(4);
  w.coop.ticket = reinterpret_cast<unsigned*>(take(4));
  w.coop.stage = reinterpret_cast<double*>(take(2 * 2 * maxc * ((blocks + kEB - 1) / kEB)));
  w.topv = take(d.M * d.splits_top * d.F * 2);
  w.ybest = take(d.M * d.F);
  w.eval = take(d.M * d.F);
  w.q = take(129 * 128);
  w.gram = take(129 * 128);
  w.total = p - base;
  return w;
}

PnIws carve_int(int32_t* base, const Dims& d) {
  PnIws w;
  int32_t* p = base;
  auto take = [&](int64_t n) {
    int32_t* r = p;
    p += (n + 3) / 4 * 4;
    return r;
  };
  w.argmax = take(d.M * d.F);
  w.topn = take(d.M * d.splits_top * d.F * 2);
  w.erow = take(d.M * d.F);
  w.ech = take(d.M * d.F);
  w.tptr = take(d.M * ((d.N + 31) / 32 + 1));
  w.vlist = take(d.M + 4);
  w.total = p - base;
  return w;
}

constexpr int kCUs = 256;  // MI355X
#ifndef MPA_PN_OVERSUB
#define MPA_PN_OVERSUB 1
#endif

// resident blocks per CU of a persistent kernel (registers and LDS decide; asked once per kernel)
template <typename Kern>
int blocks_per_cu(Kern kern, int threads) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, threads, 0) != hipSuccess || n < 1) n = 1;
  return n;
}

int check_dims(int64_t M, int64_t N, int64_t F, const char* who) {
  MPA_REQUIRE(M >= 0 && N >= 1 && F >= 64, "%s: bad sizes", who);
  MPA_REQUIRE(F == 64 || F == 128 || F == 256, "%s: feat_dim must be 64, 128 or 256", who);
  MPA_REQUIRE(M <= 32767 && N <= 32768, "%s: at most 32767 parts of at most 32768 points", who);
  return MPA_OK;
}

}  // namespace

extern "C" int mpa_pointnet_workspace(int64_t M, int64_t N, int64_t F, int64_t* float_elems,
                                      int64_t* int_elems) {
  if (int st = check_dims(M, N, F, "pointnet_workspace")) return st;
  MPA_REQUIRE(float_elems && int_elems, "pointnet_workspace: null pointer");
  const Dims d = make_dims(M, N, F);
  *float_elems = carve(nullptr, d).total;
  *int_elems = carve_int(nullptr, d).total;
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
  const PnIws iw = carve_int(int_ws, d);
  hipLaunchKernelGGL(pn_count_kernel, dim3(1), dim3(1024), 0, s, valids, (int)M, (int)N, w.count, w.coop.ticket,
                     iw.vlist, conv_w[0], w.Wt1);
  for (int l = 1; l <= 5; ++l) {
    int splits;
    int prow_m = (int)M;              // rows of `partial`: (part, split) pairs with the validity mask, or persistent blocks
    const float* prow_valid = valids;
    if (MPA_PN_FWD_WS && l >= 2 && l <= 4) {
      splits = 1;
      prow_m = 256;
      prow_valid = nullptr;
      if (l == 2)  // (64-row units, one 12-wave block per CU; 32-row units in two 10-wave blocks per CU measured 48 vs 38 us)
        hipLaunchKernelGGL((pn_fwd_ws_kernel<64, 64, 64, true>), dim3(256), dim3(768), 0, s, points, w.bn[1], conv_w[1], iw.vlist,
                           (int)N, w.Y[2], w.partial, (const float*)w.Wt1);
      else if (l == 3)
        hipLaunchKernelGGL((pn_fwd_ws_kernel<64, 64, 64, false>), dim3(256), dim3(768), 0, s, w.Y[2], w.bn[2], conv_w[2],
                           iw.vlist, (int)N, w.Y[3], w.partial, (const float*)nullptr);
      else {  // (58 KB of LDS and 76 registers: two blocks per CU)
        prow_m = MPA_PN_FWD4_BLOCKS;
        hipLaunchKernelGGL((pn_fwd_ws_kernel<64, 128, 32, false, 2>), dim3(MPA_PN_FWD4_BLOCKS), dim3(768), 0, s, w.Y[3], w.bn[3],
                           conv_w[3], iw.vlist, (int)N, w.Y[4], w.partial, (const float*)nullptr);
      }
    } else if (l == 1) {
      splits = d.tiles1;
      hipLaunchKernelGGL(pn_fwd_first_kernel<false>, dim3((unsigned)d.tiles1, (unsigned)M), dim3(kT), 0, s, points,
                         w.Wt1, valids, (int)N, (float*)nullptr, w.partial);
    } else {
      splits = l == 5 ? d.splits_top : d.splits;
#define MPA_FWD_(CI, PN, TP, FI, IN, YO, TV, TN)                                                                     \
  {                                                                                                                  \
    static const int occ = blocks_per_cu(pn_fwd_mfma_kernel<CI, PN, TP, FI>, kT);                                    \
    const long long units = (long long)M * splits, cap = (long long)kCUs * occ * MPA_PN_OVERSUB;                     \
    hipLaunchKernelGGL((pn_fwd_mfma_kernel<CI, PN, TP, FI>),                                                         \
                       dim3((unsigned)(units < cap ? units : cap), (unsigned)(d.C[l] / (64 * PN))), dim3(kT), 0, s,  \
                       IN, w.bn[l - 1], conv_w[l - 1], d.C[l], iw.vlist, (int)N, splits, YO, w.partial, TV, TN,      \
                       TP ? bn_w[4] : (const float*)nullptr, (const float*)w.Wt1);                                    \
  }
#define MPA_FWD(CI, PN, TP, IN, YO, TV, TN) MPA_FWD_(CI, PN, TP, false, IN, YO, TV, TN)
#ifndef MPA_PN_SPLIT  // 1: last layer on the bf16 matrix cores (pn_fwd_split_kernel); 0: v_mfma_f32_32x32x2_f32
#define MPA_PN_SPLIT 1
#endif
#define MPA_FWD_SPLIT(NWV)                                                                                           \
  {                                                                                                                  \
    static const int occ = blocks_per_cu(pn_fwd_split_kernel<128, NWV>, 64 * NWV);                                   \
    const long long units = (long long)M * splits, cap = (long long)kCUs * occ * MPA_PN_OVERSUB;                     \
    hipLaunchKernelGGL((pn_fwd_split_kernel<128, NWV>), dim3((unsigned)(units < cap ? units : cap)), dim3(64 * NWV), \
                       0, s, w.Y[4], w.bn[4], conv_w[4], d.C[5], iw.vlist, (int)N, splits, w.partial, w.topv,        \
                       iw.topn, bn_w[4]);                                                                            \
  }
      if (l == 5 && MPA_PN_FWD_WS) {
#define MPA_FWD_TOP_WS(NWV)                                                                                               \
  hipLaunchKernelGGL((pn_fwd_ws_top_kernel<128, NWV>), dim3(256), dim3(64 * (4 + NWV)), 0, s, w.Y[4], w.bn[4], conv_w[4],  \
                     d.C[5], iw.vlist, (int)N, splits, w.partial, w.topv, iw.topn, bn_w[4])
        if (F == 256) MPA_FWD_TOP_WS(8);
        else if (F == 128) MPA_FWD_TOP_WS(4);
        else MPA_FWD_TOP_WS(2);
#undef MPA_FWD_TOP_WS
      } else if (l == 5 && MPA_PN_SPLIT) {
        if (F == 256) MPA_FWD_SPLIT(8)
        else if (F == 128) MPA_FWD_SPLIT(4)
        else MPA_FWD_SPLIT(2)
      } else if (l == 5) {
        if (F == 256) MPA_FWD(128, 4, true, w.Y[4], (float*)nullptr, w.topv, iw.topn)
        else if (F == 128) MPA_FWD(128, 2, true, w.Y[4], (float*)nullptr, w.topv, iw.topn)
        else MPA_FWD(128, 1, true, w.Y[4], (float*)nullptr, w.topv, iw.topn)
      } else if (d.C[l] == 128) {
        MPA_FWD(64, 2, false, w.Y[l - 1], w.Y[l], (float*)nullptr, (int*)nullptr)
      } else if (l == 2) {  // its input, the first layer's output, is recomputed from the points
        MPA_FWD_(64, 1, false, true, points, w.Y[l], (float*)nullptr, (int*)nullptr)
      } else {
        MPA_FWD(64, 1, false, w.Y[l - 1], w.Y[l], (float*)nullptr, (int*)nullptr)
      }
#undef MPA_FWD
#undef MPA_FWD_
#undef MPA_FWD_SPLIT
    }
    const dim3 cg((unsigned)(d.C[l] / 64));
    if (training)
      hipLaunchKernelGGL(pn_bn_finalize_kernel, dim3(cg.x, (unsigned)(((long long)prow_m * splits + kEB - 1) / kEB)),
                         dim3(64 * kSlices), 0, s, w.partial, prow_valid, prow_m, splits, d.C[l], w.count, bn_w[l - 1],
                         bn_b[l - 1], running_mean[l - 1], running_var[l - 1], momentum, eps, w.bn[l], w.coop);
    else
      hipLaunchKernelGGL(pn_bn_from_running_kernel, cg, dim3(64), 0, s, d.C[l], bn_w[l - 1], bn_b[l - 1],
                         running_mean[l - 1], running_var[l - 1], eps, w.bn[l]);
  }
  hipLaunchKernelGGL(pn_top_finalize_kernel, dim3((unsigned)((M * F + 255) / 256)), dim3(256), 0, s, w.topv, iw.topn,
                     w.bn[5], valids, w.Y[4], w.bn[4], conv_w[4], (int)M, (int)N, (int)F, d.C[4], d.splits_top, feat,
                     iw.argmax, w.ybest);
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
  const PnIws iw = carve_int(const_cast<int32_t*>(int_ws), d);
  const int C4 = d.C[4];
  // ---- last layer: its output was never stored (see the Top2 notes); dY5 = alpha*dZ5 + gammap*Y5 + betap with
  //      dZ5 sparse and Y5 = A4 W5^T gives  dA4 = A4 Q + c0 + S W5  and  dW5 = S^T A4 + gammap (W5 G) + betap a4sum
  hipLaunchKernelGGL(pn_bwd_top_kernel, dim3((unsigned)(F / 64), (unsigned)((M + kEB - 1) / kEB)), dim3(64 * kSlices),
                     0, s, grad_feat, iw.argmax, w.ybest, valids, (int)M, (int)F, w.count, bn_w[4], w.bn[5],
                     w.coef[5], grad_bn_w[4], grad_bn_b[4], w.coop);
  hipLaunchKernelGGL(pn_top_csr_kernel, dim3((unsigned)(M + C4 + 1)), dim3((unsigned)F), sizeof(int) * F, s, iw.argmax,
                     grad_feat, w.coef[5], valids, (int)N, (int)F, iw.erow, iw.ech, w.eval, iw.tptr, (int)M, conv_w[4], C4,
                     w.q);
#ifndef MPA_PN_TOPQ  // 1: conv5's input gradient + Gram matrix in one wave-specialised pass (pn_bwd_top_q_kernel); 0: two kernels
#define MPA_PN_TOPQ 1
#endif
  auto reduce_dw = [&](int blocks, int elems, float* dst) {
    hipLaunchKernelGGL(pn_wgrad_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(64 * kSlices), 0, s, w.dwpart,
                       (const float*)nullptr, blocks, elems, dst);
  };
#if MPA_PN_TOPQ
  hipLaunchKernelGGL((pn_bwd_top_q_kernel<128, 4, 4, 4>), dim3(256), dim3(768), 0, s, w.Y[4], w.bn[4], w.q, iw.vlist, (int)N,
                     w.dZ[4], w.partial, w.dwpart, iw.erow, iw.ech, w.eval, iw.tptr, conv_w[4], (int)F);
  hipLaunchKernelGGL(pn_bwd_coef_kernel, dim3((unsigned)(C4 / 64), (unsigned)((256 + kEB - 1) / kEB)), dim3(64 * kSlices), 0, s,
                     w.partial, (const float*)nullptr, 256, 1, C4, w.count, bn_w[3], w.bn[4], w.coef[4], grad_bn_w[3],
                     grad_bn_b[3], w.coop);
  reduce_dw(256, C4 * C4 + C4, w.gram);
#else
  {
#if MPA_PN_SPLIT
#define MPA_DGRAD_TOP pn_dgrad_split_kernel<128>
#else
#define MPA_DGRAD_TOP pn_dgrad_mfma_kernel<128, 1, 4, true>
#endif
    static const int occ = blocks_per_cu(MPA_DGRAD_TOP, kT);
    const long long units = (long long)M * d.splits_dtop, cap = (long long)kCUs * occ * MPA_PN_OVERSUB;
    hipLaunchKernelGGL((MPA_DGRAD_TOP), dim3((unsigned)(units < cap ? units : cap), (unsigned)(C4 / 128)),
                       dim3(kT), 0, s, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, w.q, C4,
                       w.Y[4], w.bn[4], iw.vlist, (int)N, d.splits_dtop, w.dZ[4], w.partial, iw.erow, iw.ech, w.eval,
                       iw.tptr, conv_w[4], (int)F);
  }
  hipLaunchKernelGGL(pn_bwd_coef_kernel, dim3((unsigned)(C4 / 64), (unsigned)((M * d.splits_dtop + kEB - 1) / kEB)),
                     dim3(64 * kSlices), 0, s, w.partial, valids, (int)M, d.splits_dtop, C4, w.count, bn_w[3], w.bn[4],
                     w.coef[4], grad_bn_w[3], grad_bn_b[3], w.coop);
#ifndef MPA_PN_GRAM_SPLIT
#define MPA_PN_GRAM_SPLIT MPA_PN_SPLIT
#endif
#if MPA_PN_GRAM_SPLIT
  hipLaunchKernelGGL((pn_gram_split_kernel<128>), dim3(kWG), dim3(kT), 0, s, w.Y[4], w.bn[4], iw.vlist, (int)N, w.dwpart);
#else
  hipLaunchKernelGGL((pn_wgrad_mfma_kernel<128, 128, WG_GRAM>), dim3(kWG), dim3(kT), 0, s, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, w.Y[4], w.bn[4], iw.vlist, (int)N, w.dwpart, 0);
#endif
  reduce_dw(kWG, C4 * C4 + C4, w.gram);
#endif
  hipLaunchKernelGGL(pn_top_wgrad_kernel, dim3((unsigned)F), dim3(1024), 0, s, grad_feat, iw.argmax, valids, w.Y[4],
                     w.bn[4], conv_w[4], w.coef[5], w.gram, (int)M, (int)N, (int)F, grad_conv_w[4]);
#if MPA_PN_QFORM
  // ---- layers 4..2 in Q form (pn_bwd_q.h): Q / c0 of the layer, then input gradient + weight-gradient tables in one pass,
  //      then the next layer's BatchNorm-backward coefficients; the weight gradients themselves at the very end
  WgradReduceGroup rg{};
  PnFinish fin{};
  long long part_off = 0;
  for (int l = 4, i = 0; l >= 2; --l, ++i) {
    const int cout = d.C[l], cin = d.C[l - 1], elems = pn_bwd_q_elems(cout, cin, l == 2);
    hipLaunchKernelGGL(pn_bwd_q_prep_kernel, dim3((unsigned)(cin + 1)), dim3((unsigned)cin), 0, s, conv_w[l - 1], w.coef[l],
                       cout, cin, w.ql[l]);
    float* const dwl = w.dwpart + part_off;
#define MPA_QK(KK, RBB, NSS, NDD, NWW, FI, BPC, YP)                                                                   \
  hipLaunchKernelGGL((pn_bwd_q_kernel<KK, 64, RBB, NSS, NDD, NWW, FI, (KK == 128 ? 2 : 1)>), dim3(nb),                  \
                     dim3(64 * (NSS + NDD + NWW)), 0,                                                                   \
                     s, w.dZ[l], YP, w.bn[l - 1], conv_w[l - 1], w.coef[l], w.ql[l], iw.vlist, (int)N, w.dZ[l - 1],    \
                     w.partial, dwl, (const float*)w.Wt1)
    const int nb = (cout == 64 && MPA_PN_QB2) ? 512 : 256;
#if MPA_PN_QB2
    if (l == 2) MPA_QK(64, 32, 2, 2, 2, true, 2, points);              // Yprev = conv1's output: recomputed from the points
    else if (cout == 64) MPA_QK(64, 32, 2, 2, 2, false, 2, w.Y[l - 1]);  // 64 -> 64: two blocks per CU, 32-row units
#else
    if (l == 2) MPA_QK(64, 64, 4, 4, 4, true, 1, points);
    else if (cout == 64) MPA_QK(64, 64, 4, 4, 4, false, 1, w.Y[l - 1]);  // 64 -> 64: 64-row units, four input-gradient tiles
#endif
    else MPA_QK(128, 32, 4, 4, 4, false, 1, w.Y[l - 1]);               // 64 -> 128: 32-row units, k-split input-gradient pairs
#undef MPA_QK
    hipLaunchKernelGGL(pn_bwd_coef_kernel, dim3((unsigned)(cin / 64), (unsigned)((nb + kEB - 1) / kEB)),
                       dim3(64 * kSlices), 0, s, w.partial, (const float*)nullptr, nb, 1, cin, w.count, bn_w[l - 2],
                       w.bn[l - 1], w.coef[l - 1], grad_bn_w[l - 2], grad_bn_b[l - 2], w.coop);
    rg.part[i] = dwl;
    rg.dw[i] = w.red[l];
    rg.rows[i] = nb;
    rg.elems[i] = elems;
    fin.red[i] = w.red[l];
    fin.w[i] = conv_w[l - 1];
    fin.coef[i] = w.coef[l];
    fin.dw[i] = grad_conv_w[l - 1];
    fin.K[i] = cout;
    fin.CIN[i] = cin;
    part_off += (long long)nb * elems;
  }
  rg.first[0] = 0;
  fin.first[0] = 0;
  for (int k = 0; k < 4; ++k) rg.first[k + 1] = rg.first[k] + (k < 3 ? (rg.elems[k] + 63) / 64 : 0);
  for (int k = 0; k < 3; ++k) fin.first[k + 1] = fin.first[k] + (fin.K[k] * fin.CIN[k] + 255) / 256;
  fin.w1 = conv_w[0];
  fin.coef1 = w.coef[1];
  fin.dw1 = grad_conv_w[0];
  fin.first_layer = 2;  // conv2's table carries S = dZ1^T P, P^T P and psum
  hipLaunchKernelGGL(pn_wgrad_reduce_group_kernel, dim3((unsigned)rg.first[4]), dim3(64 * kSlices), 0, s, rg);
  hipLaunchKernelGGL(pn_bwd_finish_kernel, dim3((unsigned)(fin.first[3] + 1)), dim3(256), 0, s, fin);
#else
  // ---- layers 4..2: fused input + weight gradient, then the next layer's BatchNorm-backward coefficients
  WgradReduceGroup rg{};
  int n_wait = 0;
  long long wait_off = 0;
  for (int l = 4; l >= 2; --l) {
    const int cout = d.C[l], cin = d.C[l - 1];
#define MPA_FUSED(KK, NT, PN, TH, FI, YP)                                                                              \
  hipLaunchKernelGGL((pn_bwd_fused_kernel<KK, NT, PN, TH, FI>), dim3(nb), dim3(TH), 0, s, w.Y[l], w.dZ[l], w.coef[l],  \
                     conv_w[l - 1], YP, w.bn[l - 1], iw.vlist, (int)N, w.dZ[l - 1], w.partial, dwl,                    \
                     (const float*)w.Wt1)
    const int nb = 2 * kWF;                      // two 4-wave blocks per CU
    float* const dwl = w.dwpart + wait_off;      // this layer's partial table (reduced with the others at the end)
    rg.part[n_wait] = dwl;
    rg.dw[n_wait] = grad_conv_w[l - 1];
    rg.rows[n_wait] = nb;
    rg.elems[n_wait] = cout * cin;
    ++n_wait;
    wait_off += (long long)nb * cout * cin;
    if (l == 2) MPA_FUSED(64, 2, 1, 256, true, points);          // (Yprev = the first layer's output: recomputed)
    else if (cout == 64) MPA_FUSED(64, 2, 1, 256, false, w.Y[l - 1]);  // 64 -> 64: one 64-channel panel, 64-row units
    else MPA_FUSED(128, 1, 2, 256, false, w.Y[l - 1]);                 // 64 -> 128: two 32-channel panels, 32-row units
#undef MPA_FUSED
    hipLaunchKernelGGL(pn_bwd_coef_kernel, dim3((unsigned)(cin / 64), (unsigned)((nb + kEB - 1) / kEB)),
                       dim3(64 * kSlices), 0, s, w.partial, (const float*)nullptr, nb, 1, cin, w.count, bn_w[l - 2],
                       w.bn[l - 1], w.coef[l - 1], grad_bn_w[l - 2], grad_bn_b[l - 2], w.coop);
  }
  hipLaunchKernelGGL((pn_wgrad_mfma_kernel<64, 4, WG_FIRST>), dim3(kWG), dim3(kT), 0, s, (const float*)nullptr, w.dZ[1],
                     w.coef[1], points, (const float*)nullptr, iw.vlist, (int)N, w.dwpart + wait_off, 0, (const float*)w.Wt1);
  rg.part[n_wait] = w.dwpart + wait_off;
  rg.dw[n_wait] = grad_conv_w[0];
  rg.rows[n_wait] = kWG;
  rg.elems[n_wait] = d.C[1] * 3;
  ++n_wait;
  rg.first[0] = 0;
  for (int k = 0; k < 4; ++k) rg.first[k + 1] = rg.first[k] + (k < n_wait ? (rg.elems[k] + 63) / 64 : 0);
  hipLaunchKernelGGL(pn_wgrad_reduce_group_kernel, dim3((unsigned)rg.first[4]), dim3(64 * kSlices), 0, s, rg);
#endif
  return mpa::check_launch("pointnet_backward");
}
