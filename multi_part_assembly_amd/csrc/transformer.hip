// Part-relation transformer encoder + pose head, forward and backward, for gfx950.
//
// Replaces nn.TransformerEncoder as configured by the reference
// (multi_part_assembly/models/pn_transformer/transformer.py:4-79: pre-LN layers, ReLU FFN, key-padding mask,
// final LayerNorm, dropout 0.1) and the pose-head MLP (models/modules/regressor.py:30-68).  PyTorch runs one
// layer as ~25 library launches forward and ~60 backward; here a layer is 7 launches forward / 13 backward:
//
//   * one fp32-MFMA GEMM kernel  C = epilogue(prologue(A) . W^T + b)  (v_mfma_f32_32x32x2_f32, 32x64 tile per
//     wave, A staged through LDS in 64-wide K phases, W fragments straight from L2):
//       prologue  : LayerNorm (row statistics from a 1-wave-per-row pre-pass) or dropout mask;
//       epilogue  : bias, ReLU / LeakyReLU, dropout, residual add, ReLU-gradient mask.
//     Input gradients use the SAME kernel on transposed weight copies; weight gradients use a
//     K = tokens MFMA kernel with the bias gradient folded in.
//   * attention over P <= 64 part tokens: one 64-lane block per (sample, head); scores, masked softmax,
//     dropout and the value product never leave LDS/registers.
//   * dropout masks come from a counter-based hash of (seed, site, element) and are REGENERATED in backward.
// All reductions are fixed-order (no atomics): the step stays bit-reproducible.
#include "common.h"

namespace {

constexpr int kT = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename T>
__device__ __forceinline__ const T* opaque(const T* p) {
  asm volatile("" : "+v"(p));
  return p;
}

// ---- dropout: keep-scale of element `idx` at dropout site `site` -------------------------------------------------
struct Drop {
  unsigned long long seed;
  float p;       // drop probability; 0 disables
  float scale;   // 1 / (1 - p)
};

__device__ __forceinline__ float drop_scale(const Drop d, unsigned site, unsigned long long idx) {
  if (d.p <= 0.0f) return 1.0f;
  unsigned long long x = d.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(site + 1) + idx;
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  const float u = (float)(unsigned)(x >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0, 1)
  return u < d.p ? 0.0f : d.scale;
}

// ---- LayerNorm row statistics: one wave per row -------------------------------------------------------------------
__global__ __launch_bounds__(kT) void ln_stats_kernel(const float* __restrict__ x, int M, int D,
                                                      float eps, float* __restrict__ stats) {
  const int row = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* r = x + (long long)row * D;
  float s = 0.0f;
  for (int k = lane; k < D; k += 64) s += r[k];
  const float mean = wave_sum(s) / (float)D;
  float v = 0.0f;
  for (int k = lane; k < D; k += 64) {
    const float d = r[k] - mean;
    v += d * d;
  }
  const float var = wave_sum(v) / (float)D;
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = 1.0f / __builtin_sqrtf(var + eps);
  }
}

// out = LayerNorm(x) with precomputed stats (final norm).  grid = ceil(M*D / 256).
__global__ void ln_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int M, int D,
                                float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)M * D) return;
  const int row = (int)(i / D), k = (int)(i % D);
  out[i] = (x[i] - stats[2 * row]) * stats[2 * row + 1] * gamma[k] + beta[k];
}

// ---- GEMM  C[M,N] = epi(pro(A)[M,K] . W[N,K]^T + bias) ---------------------------------------------------------------
enum Pro { PRO_NONE = 0, PRO_LN = 1, PRO_DROP = 2 };
enum Epi { EPI_NONE = 0, EPI_RELU_DROP = 1, EPI_DROP_RESID = 2, EPI_LEAKY = 3, EPI_RELU_MASK = 4 };

struct GemmArgs {
  const float* A;      // [M, K]
  const float* W;      // [N, K]
  const float* bias;   // [N] or null
  float* C;            // [M, N]
  int M, N, K;
  // prologue
  const float* stats;  // PRO_LN: [M, 2]
  const float* gamma;  // PRO_LN: [K]
  const float* beta;   // PRO_LN: [K]
  // epilogue
  const float* resid;  // EPI_DROP_RESID: [M, N];  EPI_RELU_MASK: the saved post-ReLU activations [M, N]
  Drop drop;
  unsigned pro_site, epi_site;
};

// grid = (ceil(M/32), N/64), block = ONE wave owning rows [bx*32, +32) and columns [by*64, +64): with
// M = B*P <= a few hundred tokens the launch is latency-bound, so the tile is kept small to spread it over
// as many CUs as possible (640 x 768 -> 240 single-wave blocks).
template <int PRO, int EPI>
__global__ __launch_bounds__(64) void gemm_kernel(const GemmArgs g) {
  constexpr int KP = 64, LD = KP + 4;
  __shared__ __attribute__((aligned(16))) float lds[32 * LD];
  const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
  const int r0 = blockIdx.x * 32, n0 = blockIdx.y * 64;
  f32x16 acc0 = {0}, acc1 = {0};
  for (int k0 = 0; k0 < g.K; k0 += KP) {
    // stage A[r0..r0+32, k0..k0+64) (row segments of 256 B: 16 lanes per row, coalesced)
#pragma unroll 4
    for (int it = 0; it < 8; ++it) {
      const int idx = it * 64 + lane, rl = idx >> 4, c4 = idx & 15;
      const int row = r0 + rl, k = k0 + 4 * c4;
      float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (row < g.M) {
        v = *reinterpret_cast<const float4*>(g.A + (long long)row * g.K + k);
        if constexpr (PRO == PRO_LN) {
          const float mean = g.stats[2 * row], rstd = g.stats[2 * row + 1];
          const float4 ga = *reinterpret_cast<const float4*>(g.gamma + k);
          const float4 be = *reinterpret_cast<const float4*>(g.beta + k);
          v.x = (v.x - mean) * rstd * ga.x + be.x;
          v.y = (v.y - mean) * rstd * ga.y + be.y;
          v.z = (v.z - mean) * rstd * ga.z + be.z;
          v.w = (v.w - mean) * rstd * ga.w + be.w;
        } else if constexpr (PRO == PRO_DROP) {
          const unsigned long long e = (unsigned long long)row * g.K + k;
          v.x *= drop_scale(g.drop, g.pro_site, e);
          v.y *= drop_scale(g.drop, g.pro_site, e + 1);
          v.z *= drop_scale(g.drop, g.pro_site, e + 2);
          v.w *= drop_scale(g.drop, g.pro_site, e + 3);
        }
      }
      *reinterpret_cast<float4*>(lds + rl * LD + 4 * c4) = v;
    }
    __builtin_amdgcn_wave_barrier();
    const float4* fa = reinterpret_cast<const float4*>(lds + j * LD + h * (KP / 2));
    const float4* w0 = reinterpret_cast<const float4*>(g.W + (long long)(n0 + j) * g.K + k0 + h * (KP / 2));
    const float4* w1 = reinterpret_cast<const float4*>(g.W + (long long)(n0 + 32 + j) * g.K + k0 + h * (KP / 2));
#pragma unroll
    for (int v = 0; v < KP / 8; ++v) {
      const float4 a = fa[v], b0 = w0[v], b1 = w1[v];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  const float bias0 = g.bias ? g.bias[n0 + j] : 0.0f, bias1 = g.bias ? g.bias[n0 + 32 + j] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + acc_row(r, h);
    if (row >= g.M) continue;
    const long long o = (long long)row * g.N + n0 + j;
    float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
    if constexpr (EPI == EPI_RELU_DROP) {
      v0 = __builtin_fmaxf(v0, 0.0f) * drop_scale(g.drop, g.epi_site, (unsigned long long)o);
      v1 = __builtin_fmaxf(v1, 0.0f) * drop_scale(g.drop, g.epi_site, (unsigned long long)o + 32);
    } else if constexpr (EPI == EPI_DROP_RESID) {
      v0 = g.resid[o] + v0 * drop_scale(g.drop, g.epi_site, (unsigned long long)o);
      v1 = g.resid[o + 32] + v1 * drop_scale(g.drop, g.epi_site, (unsigned long long)o + 32);
    } else if constexpr (EPI == EPI_LEAKY) {
      v0 = v0 > 0.0f ? v0 : 0.2f * v0;
      v1 = v1 > 0.0f ? v1 : 0.2f * v1;
    } else if constexpr (EPI == EPI_RELU_MASK) {
      // gradient through dropout(relu(z)) given the saved activations a = relu(z) * keep_scale
      v0 = g.resid[o] > 0.0f ? v0 * g.drop.scale : 0.0f;
      v1 = g.resid[o + 32] > 0.0f ? v1 * g.drop.scale : 0.0f;
    }
    g.C[o] = v0;
    g.C[o + 32] = v1;
  }
}

// ---- weight gradient  dW[N,K] = pro_a(dY)[M,N]^T . pro_b(X)[M,K],  db[N] = column sums of pro_a(dY) -------------------
enum WPro { WP_NONE = 0, WP_LN = 1, WP_DROP = 2, WP_LEAKY_MASK = 3 };

struct WgradArgs {
  const float* dY;     // [M, N]
  const float* X;      // [M, K]
  float* dW;           // [N, K]
  float* db;           // [N] or null
  int M, N, K;
  const float* stats;  // WP_LN on X
  const float* gamma;
  const float* beta;
  Drop drop;           // WP_DROP on dY
  unsigned site;
};

// grid = (N/64 * K/64), block 256: the 4 waves split the M rows; 64x64 output tile per block.
template <int APRO, int BPRO>
__global__ __launch_bounds__(kT) void wgrad_kernel(const WgradArgs g) {
  __shared__ float sm[kT / 64][4][16][64];
  __shared__ float sb[kT / 64][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int kg = g.K / 64;
  const int n0 = (blockIdx.x / kg) * 64, k0 = (blockIdx.x % kg) * 64;
  const int per = (g.M + kT / 64 - 1) / (kT / 64);
  const int mb = wave * per, me = mb + per < g.M ? mb + per : g.M;
  const int cnt = me > mb ? me - mb : 0, half = (cnt + 1) / 2;
  float ga[2] = {1.0f, 1.0f}, be[2] = {0.0f, 0.0f};
  if constexpr (BPRO == WP_LN) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ga[u] = g.gamma[k0 + 32 * u + j];
      be[u] = g.beta[k0 + 32 * u + j];
    }
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[t][u] = f32x16{0};
  float bsum[2] = {0.0f, 0.0f};
  for (int s = 0; s < half; ++s) {
    const int row = mb + h * half + s;
    const bool ok = row < me;
    const long long r = ok ? row : mb;
    float a[2], b[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long long o = r * g.N + n0 + 32 * t + j;
      float v = g.dY[o];
      if constexpr (APRO == WP_DROP) v *= drop_scale(g.drop, g.site, (unsigned long long)o);
      a[t] = ok ? v : 0.0f;
      bsum[t] += a[t];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float v = g.X[r * g.K + k0 + 32 * u + j];
      if constexpr (BPRO == WP_LN) v = (v - g.stats[2 * r]) * g.stats[2 * r + 1] * ga[u] + be[u];
      b[u] = v;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) sm[wave][2 * t + u][r][lane] = acc[t][u][r];
#pragma unroll
  for (int t = 0; t < 2; ++t) bsum[t] += __shfl_xor(bsum[t], 32, 64);
  if (h == 0) {
    sb[wave][j] = bsum[0];
    sb[wave][32 + j] = bsum[1];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 4096; e += kT) {
    const int ln = e & 63, r = (e >> 6) & 15, tu = e >> 10;
    float sum = 0.0f;
#pragma unroll
    for (int wv = 0; wv < kT / 64; ++wv) sum += sm[wv][tu][r][ln];
    const int n = n0 + 32 * (tu >> 1) + acc_row(r, ln >> 5);
    const int k = k0 + 32 * (tu & 1) + (ln & 31);
    g.dW[(long long)n * g.K + k] = sum;
  }
  if (g.db != nullptr && k0 == 0 && threadIdx.x < 64) {
    float s = 0.0f;
#pragma unroll
    for (int wv = 0; wv < kT / 64; ++wv) s += sb[wv][threadIdx.x];
    g.db[n0 + threadIdx.x] = s;
  }
}

// ---- transposes ---------------------------------------------------------------------------------------------------------
__global__ void transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int rows, int cols) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = by + i, c = bx + threadIdx.x;
    if (r < rows && c < cols) tile[i][threadIdx.x] = w[(long long)r * cols + c];
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int r = bx + i, c = by + threadIdx.x;  // wt[cols][rows]
    if (r < cols && c < rows) wt[(long long)r * rows + c] = tile[threadIdx.x][i];
  }
}

// ---- attention: one 64-lane block per (sample, head) ----------------------------------------------------------------------
constexpr int kMaxP = 64, kMaxDh = 64;

// qkv [B*P, 3D] (q | k | v), valid [B*P]; probs [B, H, P, P] (post-softmax, pre-dropout); out [B*P, D]
__global__ __launch_bounds__(64) void attn_fwd_kernel(const float* __restrict__ qkv,
                                                      const float* __restrict__ valid, int P, int D, int H,
                                                      Drop drop, unsigned site, float* __restrict__ probs,
                                                      float* __restrict__ out) {
  __shared__ float q[kMaxP][kMaxDh + 1], k[kMaxP][kMaxDh + 1], v[kMaxP][kMaxDh + 1], s[kMaxP][kMaxP + 1];
  const int b = blockIdx.x / H, hd = blockIdx.x % H, dh = D / H, t = threadIdx.x;
  const float scale = 1.0f / __builtin_sqrtf((float)dh);
  for (int e = t; e < P * dh; e += 64) {
    const int i = e / dh, d = e % dh;
    const float* row = qkv + (long long)(b * P + i) * 3 * D + hd * dh + d;
    q[i][d] = row[0] * scale;  // torch scales q before the product
    k[i][d] = row[D];
    v[i][d] = row[2 * D];
  }
  __syncthreads();
  for (int e = t; e < P * P; e += 64) {
    const int i = e / P, jx = e % P;
    float a = 0.0f;
    for (int d = 0; d < dh; ++d) a = __builtin_fmaf(q[i][d], k[jx][d], a);
    s[i][jx] = valid[b * P + jx] != 0.0f ? a : -__builtin_inff();
  }
  __syncthreads();
  if (t < P) {  // softmax of row t
    float m = -__builtin_inff();
    for (int jx = 0; jx < P; ++jx) m = __builtin_fmaxf(m, s[t][jx]);
    float z = 0.0f;
    for (int jx = 0; jx < P; ++jx) {
      const float e = __expf(s[t][jx] - m);
      s[t][jx] = e;
      z += e;
    }
    const float inv = 1.0f / z;
    float* pr = probs + ((long long)(b * H + hd) * P + t) * P;
    for (int jx = 0; jx < P; ++jx) {
      const float p = s[t][jx] * inv;
      pr[jx] = p;
      s[t][jx] = p * drop_scale(drop, site, (unsigned long long)((b * H + hd) * P + t) * P + jx);
    }
  }
  __syncthreads();
  for (int e = t; e < P * dh; e += 64) {
    const int i = e / dh, d = e % dh;
    float a = 0.0f;
    for (int jx = 0; jx < P; ++jx) a = __builtin_fmaf(s[i][jx], v[jx][d], a);
    out[(long long)(b * P + i) * D + hd * dh + d] = a;
  }
}

// dO [B*P, D] -> dqkv [B*P, 3D]
__global__ __launch_bounds__(64) void attn_bwd_kernel(const float* __restrict__ qkv,
                                                      const float* __restrict__ probs,
                                                      const float* __restrict__ dout, int P, int D, int H,
                                                      Drop drop, unsigned site, float* __restrict__ dqkv) {
  __shared__ float q[kMaxP][kMaxDh + 1], k[kMaxP][kMaxDh + 1], v[kMaxP][kMaxDh + 1], go[kMaxP][kMaxDh + 1];
  __shared__ float pd[kMaxP][kMaxP + 1], ds[kMaxP][kMaxP + 1];
  const int b = blockIdx.x / H, hd = blockIdx.x % H, dh = D / H, t = threadIdx.x;
  const float scale = 1.0f / __builtin_sqrtf((float)dh);
  for (int e = t; e < P * dh; e += 64) {
    const int i = e / dh, d = e % dh;
    const float* row = qkv + (long long)(b * P + i) * 3 * D + hd * dh + d;
    q[i][d] = row[0];
    k[i][d] = row[D];
    v[i][d] = row[2 * D];
    go[i][d] = dout[(long long)(b * P + i) * D + hd * dh + d];
  }
  const float* pr = probs + (long long)(b * H + hd) * P * P;
  __syncthreads();
  // pd = dropped probabilities (for dV), ds = dP = dOut . V^T (through the dropout mask)
  for (int e = t; e < P * P; e += 64) {
    const int i = e / P, jx = e % P;
    const float m = drop_scale(drop, site, (unsigned long long)((b * H + hd) * P + i) * P + jx);
    float a = 0.0f;
    for (int d = 0; d < dh; ++d) a = __builtin_fmaf(go[i][d], v[jx][d], a);
    pd[i][jx] = pr[i * P + jx] * m;
    ds[i][jx] = a * m;
  }
  __syncthreads();
  // dV[jx][d] = sum_i pd[i][jx] * go[i][d]
  for (int e = t; e < P * dh; e += 64) {
    const int jx = e / dh, d = e % dh;
    float a = 0.0f;
    for (int i = 0; i < P; ++i) a = __builtin_fmaf(pd[i][jx], go[i][d], a);
    dqkv[(long long)(b * P + jx) * 3 * D + 2 * D + hd * dh + d] = a;
  }
  __syncthreads();
  if (t < P) {  // softmax backward of row t: dS = P * (dP - sum_j dP*P)
    float dot = 0.0f;
    for (int jx = 0; jx < P; ++jx) dot = __builtin_fmaf(ds[t][jx], pr[t * P + jx], dot);
    for (int jx = 0; jx < P; ++jx) ds[t][jx] = pr[t * P + jx] * (ds[t][jx] - dot);
  }
  __syncthreads();
  for (int e = t; e < P * dh; e += 64) {
    const int i = e / dh, d = e % dh;
    float aq = 0.0f, ak = 0.0f;
    for (int jx = 0; jx < P; ++jx) {
      aq = __builtin_fmaf(ds[i][jx], k[jx][d], aq);   // dQ[i] = sum_j dS[i][j] K[j]
      ak = __builtin_fmaf(ds[jx][i], q[jx][d], ak);   // dK[i] = sum_j dS[j][i] Q[j]
    }
    float* row = dqkv + (long long)(b * P + i) * 3 * D + hd * dh + d;
    row[0] = aq * scale;
    row[D] = ak * scale;
  }
}

// ---- LayerNorm backward -----------------------------------------------------------------------------------------------------
// dx[row] = resid[row] + rstd * (dh*gamma - mean(dh*gamma) - xhat * mean(dh*gamma*xhat));  one wave per row.
// Per-block partial sums of dgamma = sum dh*xhat and dbeta = sum dh go to part[block][2][D].
__global__ __launch_bounds__(kT) void ln_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                    const float* __restrict__ stats,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ resid, int M, int D,
                                                    float* __restrict__ dx, float* __restrict__ part) {
  extern __shared__ float sm[];  // [4 waves][2][D]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kT / 64) + wave;
  float* mine = sm + wave * 2 * D;
  if (row < M) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const long long o = (long long)row * D;
    float s1 = 0.0f, s2 = 0.0f;
    for (int k = lane; k < D; k += 64) {
      const float xh = (x[o + k] - mean) * rstd, d = dh[o + k] * gamma[k];
      s1 += d;
      s2 = __builtin_fmaf(d, xh, s2);
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    for (int k = lane; k < D; k += 64) {
      const float xh = (x[o + k] - mean) * rstd, g = dh[o + k];
      dx[o + k] = (resid ? resid[o + k] : 0.0f) + rstd * (g * gamma[k] - s1 - xh * s2);
      mine[k] = g * xh;
      mine[D + k] = g;
    }
  } else {
    for (int k = lane; k < D; k += 64) mine[k] = mine[D + k] = 0.0f;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * D; k += kT) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < kT / 64; ++w) s += sm[w * 2 * D + k];
    part[(long long)blockIdx.x * 2 * D + k] = s;
  }
}

// dgamma[k] = sum_blocks part[b][0][k], dbeta likewise.  grid = ceil(2D/256)
__global__ void ln_reduce_kernel(const float* __restrict__ part, int blocks, int D, float* __restrict__ dgamma,
                                 float* __restrict__ dbeta) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= 2 * D) return;
  float s = 0.0f;
  for (int b = 0; b < blocks; ++b) s += part[(long long)b * 2 * D + k];
  if (k < D) dgamma[k] = s;
  else dbeta[k - D] = s;
}

// y = x * drop_scale (materialised dropout of a gradient); n elements
__global__ void drop_apply_kernel(const float* __restrict__ x, Drop drop, unsigned site, long long n,
                                  float* __restrict__ y) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] * drop_scale(drop, site, (unsigned long long)i);
}

// ---- pose head tail: rot = normalize(h . Wr^T + br), trans = h . Wt^T + bt; one wave per token -------------------------------
__global__ __launch_bounds__(kT) void head_fwd_kernel(const float* __restrict__ hfeat,
                                                      const float* __restrict__ wr, const float* __restrict__ br,
                                                      const float* __restrict__ wt, const float* __restrict__ bt,
                                                      int M, int K, float* __restrict__ rot_raw,
                                                      float* __restrict__ rot, float* __restrict__ trans) {
  const int row = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float a[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int k = lane; k < K; k += 64) {
    const float x = hfeat[(long long)row * K + k];
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = __builtin_fmaf(x, wr[c * K + k], a[c]);
#pragma unroll
    for (int c = 0; c < 3; ++c) a[4 + c] = __builtin_fmaf(x, wt[c * K + k], a[4 + c]);
  }
#pragma unroll
  for (int c = 0; c < 7; ++c) a[c] = wave_sum(a[c]);
  if (lane == 0) {
    float q[4], n2 = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      q[c] = a[c] + br[c];
      n2 += q[c] * q[c];
      rot_raw[4 * row + c] = q[c];
    }
    const float inv = 1.0f / __builtin_fmaxf(__builtin_sqrtf(n2), 1e-12f);  // F.normalize(p=2, eps=1e-12)
#pragma unroll
    for (int c = 0; c < 4; ++c) rot[4 * row + c] = q[c] * inv;
#pragma unroll
    for (int c = 0; c < 3; ++c) trans[3 * row + c] = a[4 + c] + bt[c];
  }
}

// backward of the tail: d(rot_raw) from d(rot) through the normalisation, then dh = dq . Wr + dt . Wt  and the
// raw-gradient rows dqt [M, 8] = (dq0..3, dt0..2, 0) for the weight-gradient reduction.
__global__ __launch_bounds__(kT) void head_bwd_kernel(const float* __restrict__ rot_raw,
                                                      const float* __restrict__ grot,
                                                      const float* __restrict__ gtrans,
                                                      const float* __restrict__ wr, const float* __restrict__ wt,
                                                      int M, int K, float* __restrict__ dqt,
                                                      float* __restrict__ dh) {
  const int row = blockIdx.x * (kT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  float q[4], g[4], n2 = 0.0f, dot = 0.0f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    q[c] = rot_raw[4 * row + c];
    g[c] = grot[4 * row + c];
    n2 += q[c] * q[c];
  }
  const float n = __builtin_sqrtf(n2), nn = __builtin_fmaxf(n, 1e-12f), inv = 1.0f / nn;
#pragma unroll
  for (int c = 0; c < 4; ++c) dot += g[c] * q[c];
  float dq[4], dt[3];
#pragma unroll
  for (int c = 0; c < 4; ++c) dq[c] = n > 1e-12f ? inv * (g[c] - q[c] * dot * inv * inv) : g[c] * inv;
#pragma unroll
  for (int c = 0; c < 3; ++c) dt[c] = gtrans[3 * row + c];
  if (lane < 8) dqt[8 * row + lane] = lane < 4 ? dq[lane] : (lane < 7 ? dt[lane - 4] : 0.0f);
  for (int k = lane; k < K; k += 64) {
    float a = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) a = __builtin_fmaf(dq[c], wr[c * K + k], a);
#pragma unroll
    for (int c = 0; c < 3; ++c) a = __builtin_fmaf(dt[c], wt[c * K + k], a);
    dh[(long long)row * K + k] = a;
  }
}

// dWr[c][k] = sum_rows dqt[row][c] * h[row][k] (c < 4), dWt likewise (c = 4..6), biases = column sums of dqt.
// grid = ceil(K/64) blocks of 256: wave w sums rows w, w+4, ...; lane = k.
__global__ __launch_bounds__(kT) void head_wgrad_kernel(const float* __restrict__ dqt,
                                                        const float* __restrict__ hfeat, int M, int K,
                                                        float* __restrict__ dwr, float* __restrict__ dbr,
                                                        float* __restrict__ dwt, float* __restrict__ dbt) {
  __shared__ float sm[kT / 64][8][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, k = blockIdx.x * 64 + lane;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int row = wave; row < M; row += kT / 64) {
    const float x = k < K ? hfeat[(long long)row * K + k] : 0.0f;
#pragma unroll
    for (int c = 0; c < 7; ++c) a[c] = __builtin_fmaf(dqt[8 * row + c], x, a[c]);
    if (blockIdx.x == 0 && lane < 7) a[7] += dqt[8 * row + lane];  // bias gradients, lane = output
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) sm[wave][c][lane] = a[c];
  __syncthreads();
  if (wave == 0) {
    float s[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = (sm[0][c][lane] + sm[1][c][lane]) + (sm[2][c][lane] + sm[3][c][lane]);
    if (k < K) {
#pragma unroll
      for (int c = 0; c < 4; ++c) dwr[c * K + k] = s[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) dwt[c * K + k] = s[4 + c];
    }
    if (blockIdx.x == 0 && lane < 7) {
      if (lane < 4) dbr[lane] = s[7];
      else dbt[lane - 4] = s[7];
    }
  }
}

// LeakyReLU(0.2) gradient mask: y = dy where act > 0 else 0.2 * dy  (act = saved post-activation values)
__global__ void leaky_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ act, long long n,
                                 float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = act[i] > 0.0f ? dy[i] : 0.2f * dy[i];
}

// ---- host helpers ---------------------------------------------------------------------------------------------------------------
template <int PRO, int EPI>
void launch_gemm(const GemmArgs& g, hipStream_t s) {
  hipLaunchKernelGGL((gemm_kernel<PRO, EPI>), dim3((g.M + 31) / 32, g.N / 64), dim3(64), 0, s, g);
}

template <int APRO, int BPRO>
void launch_wgrad(const WgradArgs& g, hipStream_t s) {
  hipLaunchKernelGGL((wgrad_kernel<APRO, BPRO>), dim3((g.N / 64) * (g.K / 64)), dim3(kT), 0, s, g);
}

void launch_transpose(const float* w, float* wt, int rows, int cols, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, s, w, wt, rows, cols);
}

GemmArgs gemm_args(const float* A, const float* W, const float* bias, float* C, int M, int N, int K) {
  GemmArgs g{};
  g.A = A;
  g.W = W;
  g.bias = bias;
  g.C = C;
  g.M = M;
  g.N = N;
  g.K = K;
  g.drop = Drop{0, 0.0f, 1.0f};
  return g;
}

WgradArgs wgrad_args(const float* dY, const float* X, float* dW, float* db, int M, int N, int K) {
  WgradArgs g{};
  g.dY = dY;
  g.X = X;
  g.dW = dW;
  g.db = db;
  g.M = M;
  g.N = N;
  g.K = K;
  g.drop = Drop{0, 0.0f, 1.0f};
  return g;
}

// parameter slots of one encoder layer (order of nn.TransformerEncoderLayer.named_parameters())
enum { P_WQKV = 0, P_BQKV, P_WO, P_BO, P_W1, P_B1, P_W2, P_B2, P_G1, P_BE1, P_G2, P_BE2, P_PER_LAYER };
enum { S_ATTN = 0, S_SA_OUT = 1, S_FFN = 2, S_FFN_OUT = 3, S_PER_LAYER = 4 };  // dropout sites

struct TfDims {
  int64_t B, P, D, H, FF, L, M;
};

struct TfWs {  // per-layer saved tensors + scratch
  float *x_in, *stats1, *qkv, *probs, *o, *x_mid, *stats2, *f;
};

struct TfLayout {
  TfWs layer[16];
  float *x_final, *stats_f;
  // backward scratch
  float *g_a, *g_b, *g_c, *dz, *dqkv, *wt_a, *lnpart;
  int64_t total;
};

TfLayout tf_carve(float* base, const TfDims& d) {
  TfLayout w;
  float* p = base;
  auto take = [&](int64_t n) {
    float* r = p;
    p += (n + 3) / 4 * 4;
    return r;
  };
  for (int l = 0; l < d.L; ++l) {
    w.layer[l].x_in = take(d.M * d.D);
    w.layer[l].stats1 = take(2 * d.M);
    w.layer[l].qkv = take(d.M * 3 * d.D);
    w.layer[l].probs = take(d.B * d.H * d.P * d.P);
    w.layer[l].o = take(d.M * d.D);
    w.layer[l].x_mid = take(d.M * d.D);
    w.layer[l].stats2 = take(2 * d.M);
    w.layer[l].f = take(d.M * d.FF);
  }
  w.x_final = take(d.M * d.D);
  w.stats_f = take(2 * d.M);
  w.g_a = take(d.M * d.D);
  w.g_b = take(d.M * d.D);
  w.g_c = take(d.M * d.D);
  w.dz = take(d.M * d.FF);
  w.dqkv = take(d.M * 3 * d.D);
  const int64_t wmax = d.D * (d.FF > 3 * d.D ? d.FF : 3 * d.D);
  w.wt_a = take(wmax);
  w.lnpart = take(((d.M + 3) / 4) * 2 * d.D);
  w.total = p - base;
  return w;
}

int tf_check(const TfDims& d, const char* who) {
  MPA_REQUIRE(d.B >= 0 && d.P >= 1 && d.P <= kMaxP, "%s: need 1 <= P <= 64", who);
  MPA_REQUIRE(d.D % 64 == 0 && d.FF % 64 == 0 && d.H >= 1 && d.D % d.H == 0 && d.D / d.H <= kMaxDh,
              "%s: need D, FF multiples of 64 and head dim <= 64", who);
  MPA_REQUIRE(d.L >= 1 && d.L <= 16, "%s: 1..16 layers", who);
  return MPA_OK;
}

}  // namespace

extern "C" int mpa_transformer_workspace(int64_t B, int64_t P, int64_t D, int64_t H, int64_t FF, int64_t L,
                                         int64_t* float_elems) {
  const TfDims d{B, P, D, H, FF, L, B * P};
  if (int st = tf_check(d, "transformer_workspace")) return st;
  MPA_REQUIRE(float_elems != nullptr, "transformer_workspace: null pointer");
  *float_elems = tf_carve(nullptr, d).total;
  return MPA_OK;
}

extern "C" int mpa_transformer_forward(const float* tokens, const float* valid, const float* const* params,
                                       int64_t B, int64_t P, int64_t D, int64_t H, int64_t FF, int64_t L,
                                       float dropout_p, uint64_t seed, float* ws, float* out, void* stream) {
  const TfDims d{B, P, D, H, FF, L, B * P};
  if (int st = tf_check(d, "transformer_forward")) return st;
  if (B == 0) return MPA_OK;
  MPA_REQUIRE(tokens && valid && params && ws && out, "transformer_forward: null pointer");
  MPA_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, "transformer_forward: dropout must be in [0, 1)");
  hipStream_t s = mpa::as_stream(stream);
  const TfLayout w = tf_carve(ws, d);
  const Drop drop{seed, dropout_p, 1.0f / (1.0f - dropout_p)};
  const int M = (int)d.M, Di = (int)D, FFi = (int)FF;
  const float eps = 1e-5f;
  const dim3 rows((M + 3) / 4);
  for (int l = 0; l < L; ++l) {  // layer l + 1 finds its input already in its own x_in slot
    const float* const* pp = params + l * P_PER_LAYER;
    const TfWs& t = w.layer[l];
    const unsigned site0 = (unsigned)(l * S_PER_LAYER);
    if (l == 0 && hipMemcpyAsync(t.x_in, tokens, sizeof(float) * d.M * D, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return mpa::check_launch("transformer_forward(copy)");
    hipLaunchKernelGGL(ln_stats_kernel, rows, dim3(kT), 0, s, t.x_in, M, Di, eps, t.stats1);
    GemmArgs g = gemm_args(t.x_in, pp[P_WQKV], pp[P_BQKV], t.qkv, M, 3 * Di, Di);
    g.stats = t.stats1;
    g.gamma = pp[P_G1];
    g.beta = pp[P_BE1];
    launch_gemm<PRO_LN, EPI_NONE>(g, s);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(B * H)), dim3(64), 0, s, t.qkv, valid, (int)P, Di, (int)H,
                       drop, site0 + S_ATTN, t.probs, t.o);
    g = gemm_args(t.o, pp[P_WO], pp[P_BO], t.x_mid, M, Di, Di);
    g.resid = t.x_in;
    g.drop = drop;
    g.epi_site = site0 + S_SA_OUT;
    launch_gemm<PRO_NONE, EPI_DROP_RESID>(g, s);
    hipLaunchKernelGGL(ln_stats_kernel, rows, dim3(kT), 0, s, t.x_mid, M, Di, eps, t.stats2);
    g = gemm_args(t.x_mid, pp[P_W1], pp[P_B1], t.f, M, FFi, Di);
    g.stats = t.stats2;
    g.gamma = pp[P_G2];
    g.beta = pp[P_BE2];
    g.drop = drop;
    g.epi_site = site0 + S_FFN;
    launch_gemm<PRO_LN, EPI_RELU_DROP>(g, s);
    float* x_out = l + 1 < L ? w.layer[l + 1].x_in : w.x_final;
    g = gemm_args(t.f, pp[P_W2], pp[P_B2], x_out, M, Di, FFi);
    g.resid = t.x_mid;
    g.drop = drop;
    g.epi_site = site0 + S_FFN_OUT;
    launch_gemm<PRO_NONE, EPI_DROP_RESID>(g, s);
  }
  hipLaunchKernelGGL(ln_stats_kernel, rows, dim3(kT), 0, s, w.x_final, M, Di, eps, w.stats_f);
  const float* const* fin = params + L * P_PER_LAYER;
  hipLaunchKernelGGL(ln_apply_kernel, dim3((unsigned)((d.M * D + 255) / 256)), dim3(256), 0, s, w.x_final, w.stats_f,
                     fin[0], fin[1], M, Di, out);
  return mpa::check_launch("transformer_forward");
}

extern "C" int mpa_transformer_backward(const float* grad_out, const float* valid, const float* const* params,
                                        int64_t B, int64_t P, int64_t D, int64_t H, int64_t FF, int64_t L,
                                        float dropout_p, uint64_t seed, float* ws, float* grad_tokens,
                                        float* const* grad_params, void* stream) {
  const TfDims d{B, P, D, H, FF, L, B * P};
  if (int st = tf_check(d, "transformer_backward")) return st;
  if (B == 0) return MPA_OK;
  MPA_REQUIRE(grad_out && valid && params && ws && grad_tokens && grad_params, "transformer_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const TfLayout w = tf_carve(ws, d);
  const Drop drop{seed, dropout_p, 1.0f / (1.0f - dropout_p)};
  const Drop nodrop{0, 0.0f, 1.0f};
  const int M = (int)d.M, Di = (int)D, FFi = (int)FF;
  const dim3 rows((M + 3) / 4);
  const unsigned lnblocks = (unsigned)((M + 3) / 4);
  const size_t ln_smem = sizeof(float) * 4 * 2 * D;
  const dim3 red((unsigned)((2 * D + 255) / 256));
  const float* const* fin = params + L * P_PER_LAYER;
  float* const* gfin = grad_params + L * P_PER_LAYER;
  // final LayerNorm backward -> g_a = d x_final
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(lnblocks), dim3(kT), ln_smem, s, grad_out, w.x_final, w.stats_f, fin[0],
                     (const float*)nullptr, M, Di, w.g_a, w.lnpart);
  hipLaunchKernelGGL(ln_reduce_kernel, red, dim3(256), 0, s, w.lnpart, (int)lnblocks, Di, gfin[0], gfin[1]);
  float* g = w.g_a;      // gradient w.r.t. the current layer's output
  float* spare = w.g_b;  // rotating buffers
  float* spare2 = w.g_c;
  for (int l = (int)L - 1; l >= 0; --l) {
    const float* const* pp = params + l * P_PER_LAYER;
    float* const* gp = grad_params + l * P_PER_LAYER;
    const TfWs& t = w.layer[l];
    const unsigned site0 = (unsigned)(l * S_PER_LAYER);
    // ---- FFN: x_out = x_mid + drop(f . W2^T + b2),  f = drop(relu(LN2(x_mid) . W1^T + b1))
    const float* gd = g;  // g with the output-dropout mask applied
    if (dropout_p > 0.0f) {
      hipLaunchKernelGGL(drop_apply_kernel, dim3((unsigned)((d.M * D + 255) / 256)), dim3(256), 0, s, g, drop,
                         site0 + S_FFN_OUT, (long long)(d.M * D), spare);
      gd = spare;
    }
    launch_wgrad<WP_NONE, WP_NONE>(wgrad_args(gd, t.f, gp[P_W2], gp[P_B2], M, Di, FFi), s);
    launch_transpose(pp[P_W2], w.wt_a, Di, FFi, s);  // W2 [D, FF] -> [FF, D]
    GemmArgs ga = gemm_args(gd, w.wt_a, nullptr, w.dz, M, FFi, Di);
    ga.resid = t.f;
    ga.drop = dropout_p > 0.0f ? drop : nodrop;
    launch_gemm<PRO_NONE, EPI_RELU_MASK>(ga, s);  // dz = d(pre-activation)
    WgradArgs wa = wgrad_args(w.dz, t.x_mid, gp[P_W1], gp[P_B1], M, FFi, Di);
    wa.stats = t.stats2;
    wa.gamma = pp[P_G2];
    wa.beta = pp[P_BE2];
    launch_wgrad<WP_NONE, WP_LN>(wa, s);
    launch_transpose(pp[P_W1], w.wt_a, FFi, Di, s);  // W1 [FF, D] -> [D, FF]
    launch_gemm<PRO_NONE, EPI_NONE>(gemm_args(w.dz, w.wt_a, nullptr, spare2, M, Di, FFi), s);  // d LN2 output
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(lnblocks), dim3(kT), ln_smem, s, spare2, t.x_mid, t.stats2, pp[P_G2], g,
                       M, Di, spare, w.lnpart);  // spare = d x_mid
    hipLaunchKernelGGL(ln_reduce_kernel, red, dim3(256), 0, s, w.lnpart, (int)lnblocks, Di, gp[P_G2], gp[P_BE2]);
    // rotate: g_mid lives in `spare`
    float* g_mid = spare;
    spare = g;
    // ---- attention block: x_mid = x_in + drop(o . Wo^T + bo)
    const float* gmd = g_mid;
    if (dropout_p > 0.0f) {
      hipLaunchKernelGGL(drop_apply_kernel, dim3((unsigned)((d.M * D + 255) / 256)), dim3(256), 0, s, g_mid, drop,
                         site0 + S_SA_OUT, (long long)(d.M * D), spare);
      gmd = spare;
    }
    launch_wgrad<WP_NONE, WP_NONE>(wgrad_args(gmd, t.o, gp[P_WO], gp[P_BO], M, Di, Di), s);
    launch_transpose(pp[P_WO], w.wt_a, Di, Di, s);
    launch_gemm<PRO_NONE, EPI_NONE>(gemm_args(gmd, w.wt_a, nullptr, spare2, M, Di, Di), s);  // d o
    hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)(B * H)), dim3(64), 0, s, t.qkv, t.probs, spare2, (int)P, Di,
                       (int)H, dropout_p > 0.0f ? drop : nodrop, site0 + S_ATTN, w.dqkv);
    wa = wgrad_args(w.dqkv, t.x_in, gp[P_WQKV], gp[P_BQKV], M, 3 * Di, Di);
    wa.stats = t.stats1;
    wa.gamma = pp[P_G1];
    wa.beta = pp[P_BE1];
    launch_wgrad<WP_NONE, WP_LN>(wa, s);
    launch_transpose(pp[P_WQKV], w.wt_a, 3 * Di, Di, s);  // [3D, D] -> [D, 3D]
    launch_gemm<PRO_NONE, EPI_NONE>(gemm_args(w.dqkv, w.wt_a, nullptr, spare2, M, Di, 3 * Di), s);  // d LN1 output
    float* g_in = l == 0 ? grad_tokens : spare;
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(lnblocks), dim3(kT), ln_smem, s, spare2, t.x_in, t.stats1, pp[P_G1], g_mid,
                       M, Di, g_in, w.lnpart);
    hipLaunchKernelGGL(ln_reduce_kernel, red, dim3(256), 0, s, w.lnpart, (int)lnblocks, Di, gp[P_G1], gp[P_BE1]);
    // rotate buffers: next g = g_in (in `spare`), free ones: g_mid's buffer
    if (l > 0) {
      float* old_gmid = g_mid;
      g = spare;
      spare = old_gmid;
    }
  }
  return mpa::check_launch("transformer_backward");
}

// ---- pose head -------------------------------------------------------------------------------------------------------------------
// params: fc1.w [256,F], fc1.b, fc2.w [128,256], fc2.b, rot.w [4,128], rot.b, trans.w [3,128], trans.b
// ws: h1 [M,256] | h2 [M,128] | rot_raw [M,4] | dqt [M,8] | d2 [M,128] | d1 [M,256] | wt [max]
extern "C" int mpa_pose_head_workspace(int64_t M, int64_t F, int64_t* float_elems) {
  MPA_REQUIRE(M >= 0 && F >= 64 && F % 64 == 0 && float_elems, "pose_head_workspace: need F multiple of 64");
  const int64_t wmax = 256 * (F > 256 ? F : 256);
  *float_elems = M * (256 + 128 + 4 + 8 + 128 + 256) + wmax + 64;
  return MPA_OK;
}

extern "C" int mpa_pose_head_forward(const float* x, const float* const* params, int64_t M, int64_t F, float* ws,
                                     float* rot, float* trans, void* stream) {
  MPA_REQUIRE(M >= 0 && F >= 64 && F % 64 == 0, "pose_head_forward: need F multiple of 64");
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(x && params && ws && rot && trans, "pose_head_forward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  float* h1 = ws;
  float* h2 = h1 + M * 256;
  float* rot_raw = h2 + M * 128;
  launch_gemm<PRO_NONE, EPI_LEAKY>(gemm_args(x, params[0], params[1], h1, (int)M, 256, (int)F), s);
  launch_gemm<PRO_NONE, EPI_LEAKY>(gemm_args(h1, params[2], params[3], h2, (int)M, 128, 256), s);
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(kT), 0, s, h2, params[4], params[5],
                     params[6], params[7], (int)M, 128, rot_raw, rot, trans);
  return mpa::check_launch("pose_head_forward");
}

extern "C" int mpa_pose_head_backward(const float* grad_rot, const float* grad_trans, const float* x,
                                      const float* const* params, int64_t M, int64_t F, float* ws, float* grad_x,
                                      float* const* grad_params, void* stream) {
  MPA_REQUIRE(M >= 0 && F >= 64 && F % 64 == 0, "pose_head_backward: need F multiple of 64");
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(grad_rot && grad_trans && x && params && ws && grad_x && grad_params, "pose_head_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  float* h1 = ws;
  float* h2 = h1 + M * 256;
  float* rot_raw = h2 + M * 128;
  float* dqt = rot_raw + M * 4;
  float* d2 = dqt + M * 8;
  float* d1 = d2 + M * 128;
  float* wt = d1 + M * 256;
  const int Mi = (int)M;
  hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(kT), 0, s, rot_raw, grad_rot, grad_trans,
                     params[4], params[6], Mi, 128, dqt, d2);
  hipLaunchKernelGGL(head_wgrad_kernel, dim3(2), dim3(kT), 0, s, dqt, h2, Mi, 128, grad_params[4], grad_params[5],
                     grad_params[6], grad_params[7]);
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3((unsigned)((M * 128 + 255) / 256)), dim3(256), 0, s, d2, h2,
                     (long long)(M * 128), d2);
  launch_wgrad<WP_NONE, WP_NONE>(wgrad_args(d2, h1, grad_params[2], grad_params[3], Mi, 128, 256), s);
  launch_transpose(params[2], wt, 128, 256, s);  // [128,256] -> [256,128]
  launch_gemm<PRO_NONE, EPI_NONE>(gemm_args(d2, wt, nullptr, d1, Mi, 256, 128), s);
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3((unsigned)((M * 256 + 255) / 256)), dim3(256), 0, s, d1, h1,
                     (long long)(M * 256), d1);
  launch_wgrad<WP_NONE, WP_NONE>(wgrad_args(d1, x, grad_params[0], grad_params[1], Mi, 256, (int)F), s);
  launch_transpose(params[0], wt, 256, (int)F, s);  // [256,F] -> [F,256]
  launch_gemm<PRO_NONE, EPI_NONE>(gemm_args(d1, wt, nullptr, grad_x, Mi, (int)F, 256), s);
  return mpa::check_launch("pose_head_backward");
}
