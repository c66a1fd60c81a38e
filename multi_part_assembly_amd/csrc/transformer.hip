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
#include "tf_gemm.h"

namespace {

using namespace tfg;

constexpr int kT = 256;

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ float wave_sum(float v) { return mpa::wave_sum_dpp(v); }  // (common.h: DPP, no LDS permutes)

using mpa::wave_sum_dpp;
template <typename T>
__device__ __forceinline__ const T* opaque(const T* p) {
  asm volatile("" : "+v"(p));
  return p;
}

// ---- LayerNorm forward: one wave per row; writes the row statistics (for backward) and y = LN(x) ---------------------
// x_copy (nullable): also saves x itself (the first layer keeps its input for backward without a memcpy node).
__global__ __launch_bounds__(kT) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, int M, int D, float eps,
                                                    float* __restrict__ stats, float* __restrict__ y,
                                                    float* __restrict__ x_copy) {
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
  const float rstd = 1.0f / __builtin_sqrtf(wave_sum(v) / (float)D + eps);
  if (lane == 0) {
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
  for (int k = lane; k < D; k += 64) {
    const float v = r[k];
    y[(long long)row * D + k] = (v - mean) * rstd * gamma[k] + beta[k];
    if (x_copy != nullptr) x_copy[(long long)row * D + k] = v;
  }
}

// ---- weight gradient  dW[N,K] = dY[M,N]^T . X[M,K],  db[N] = column sums of dY ---------------------------------------
struct WgradArgs {
  const float* dY;     // [M, N]
  const float* X;      // [M, K]
  float* dW;           // [N, K]
  float* db;           // [N] or null
  int M, N, K;
  int ldx;             // row stride of X (0: K)
};

// Up to kGroup independent weight gradients share one launch (the four of a transformer layer's backward: each
// alone is a ~10 us latency-bound kernel); blocks [first[i], first[i+1]) work on problem i.
constexpr int kGroup = 4;
struct WgradGroup {
  WgradArgs p[kGroup];
  int first[kGroup + 1];
};

// grid = (N/32 * K/32) blocks per problem, block = 4 waves that split the M rows (640 tokens -> 160 rows = 80 MFMA
// k-steps each); ONE 32 x 32 output tile per block: a layer's four problems are 768 small blocks, all resident at
// once (3 per CU), so that one wave's loads overlap the other waves' MFMA chains on the same SIMD — a CU that runs
// a whole 16-wave block in lockstep instead alternates between waiting for loads and queueing on the MFMA pipe.
// Lane (j, h) reads, per step, dY[row 2s+h][n0+j] and X[row 2s+h][k0+j] (128 B per half-wave each), twenty steps'
// loads in flight at a time.  The 4 partial tiles are summed in fixed order through LDS (deterministic), the bias
// gradient likewise.
#ifndef MPA_TF_WT
#define MPA_TF_WT 256
#endif
constexpr int kWT = MPA_TF_WT, kWW = kWT / 64;

__global__ __launch_bounds__(kWT) void wgrad_kernel(const WgradGroup G) {
  __shared__ float sm[kWW - 1][16][64];
  __shared__ float sb[kWW][32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < kGroup; ++i) pi += (int)blockIdx.x >= G.first[i] ? 1 : 0;  // first[] is non-decreasing
  const WgradArgs& g = G.p[pi];
  const int blk = (int)blockIdx.x - G.first[pi];
  // tile of this block.  Blocks go to the 8 XCDs round-robin and every XCD has its own L2: the blocks of one XCD
  // take a contiguous eighth of the tiles along the LARGER operand's dimension, so that operand is pulled into
  // exactly one L2 (plain row-major tile order makes all 8 L2s fetch both operands in full: measured 42 MB of
  // fetches per layer for ~6 MB of operands).
  const int tn = g.N / 32, tk = g.K / 32;
  int nt, kt;
  const bool big_k = tk >= tn;
  const int tbig = big_k ? tk : tn;
  if (tbig % 8 == 0 && G.first[pi] % 8 == 0) {
    const int xcd = blk & 7, slot = blk >> 3, per_xcd = tbig / 8;
    const int big = xcd * per_xcd + slot % per_xcd, small = slot / per_xcd;
    nt = big_k ? small : big;
    kt = big_k ? big : small;
  } else {
    nt = blk / tk;
    kt = blk % tk;
  }
  const int n0 = nt * 32, k0 = kt * 32;
  const int per = ((g.M + kWW - 1) / kWW + 1) & ~1;  // rows per wave, even
  const int mb = wave * per, me = mb + per < g.M ? mb + per : g.M, ns = per / 2;
  f32x16 acc = {0};
  float bsum = 0.0f;
  struct Frag {
    float a, b;
  };
  auto load = [&](int s, Frag& f) {
    const int row = mb + 2 * s + h;
    const long long r = row < me ? row : 0;  // out-of-range steps read row 0 and are zeroed in `step`
    f.a = g.dY[r * g.N + n0 + j];
    f.b = g.X[r * (g.ldx > 0 ? g.ldx : g.K) + k0 + j];
  };
  auto step = [&](int s, const Frag& f) {
    if (s >= ns) return;
    const int row = mb + 2 * s + h;
    float a = f.a, b = f.b;
    if (row >= me) a = b = 0.0f;
    bsum += a;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  };
  constexpr int U = 20;
  for (int s0 = 0; s0 < ns; s0 += U) {
    Frag f[U];
#pragma unroll
    for (int u = 0; u < U; ++u) load(s0 + u, f[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) step(s0 + u, f[u]);
  }
  bsum += __shfl_xor(bsum, 32, 64);
  if (h == 0) sb[wave][j] = bsum;
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sm[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
#pragma unroll
      for (int wv = 0; wv < kWW - 1; ++wv) v += sm[wv][r][lane];  // fixed order
      g.dW[(long long)(n0 + acc_row(r, h)) * g.K + k0 + j] = v;
    }
  } else if (wave == 1 && g.db != nullptr && k0 == 0 && lane < 32) {
    float s = 0.0f;
#pragma unroll
    for (int wv = 0; wv < kWW; ++wv) s += sb[wv][lane];
    g.db[n0 + lane] = s;
  }
}

// ---- attention: one 256-thread block per (sample, head) ----------------------------------------------------------------------
constexpr int kMaxP = 64, kMaxDh = 64, kAT = 256;

// qkv [B*P, 3D] (q | k | v), valid [B*P]; probs [B, H, P, P] (post-softmax, pre-dropout); out [B*P, D]
__global__ __launch_bounds__(kAT) void attn_fwd_kernel(const float* __restrict__ qkv,
                                                      const float* __restrict__ valid, int P, int D, int H,
                                                      Drop drop_in, unsigned site, float* __restrict__ probs,
                                                      float* __restrict__ out) {
  const Drop drop = resolve_seed(drop_in);
  __shared__ float q[kMaxP][kMaxDh + 1], k[kMaxP][kMaxDh + 1], v[kMaxP][kMaxDh + 1], s[kMaxP][kMaxP + 1];
  const int b = blockIdx.x / H, hd = blockIdx.x % H, dh = D / H, t = threadIdx.x;
  const float scale = 1.0f / __builtin_sqrtf((float)dh);
  for (int e = t; e < P * dh; e += kAT) {
    const int i = e / dh, d = e % dh;
    const float* row = qkv + (long long)(b * P + i) * 3 * D + hd * dh + d;
    q[i][d] = row[0] * scale;  // torch scales q before the product
    k[i][d] = row[D];
    v[i][d] = row[2 * D];
  }
  __syncthreads();
  for (int e = t; e < P * P; e += kAT) {
    const int i = e / P, jx = e % P;
    float a = 0.0f;
    for (int d = 0; d < dh; ++d) a = __builtin_fmaf(q[i][d], k[jx][d], a);
    s[i][jx] = valid[b * P + jx] == 1.0f ? a : -__builtin_inff();  // a real part iff == 1 (network.py: part_valids == 1)
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
  for (int e = t; e < P * dh; e += kAT) {
    const int i = e / dh, d = e % dh;
    float a = 0.0f;
    for (int jx = 0; jx < P; ++jx) a = __builtin_fmaf(s[i][jx], v[jx][d], a);
    out[(long long)(b * P + i) * D + hd * dh + d] = a;
  }
}

// dO [B*P, D] -> dqkv [B*P, 3D]
__global__ __launch_bounds__(kAT) void attn_bwd_kernel(const float* __restrict__ qkv,
                                                      const float* __restrict__ probs,
                                                      const float* __restrict__ dout, int P, int D, int H,
                                                      Drop drop_in, unsigned site, float* __restrict__ dqkv) {
  const Drop drop = resolve_seed(drop_in);
  __shared__ float q[kMaxP][kMaxDh + 1], k[kMaxP][kMaxDh + 1], v[kMaxP][kMaxDh + 1], go[kMaxP][kMaxDh + 1];
  __shared__ float pd[kMaxP][kMaxP + 1], ds[kMaxP][kMaxP + 1];
  const int b = blockIdx.x / H, hd = blockIdx.x % H, dh = D / H, t = threadIdx.x;
  const float scale = 1.0f / __builtin_sqrtf((float)dh);
  for (int e = t; e < P * dh; e += kAT) {
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
  for (int e = t; e < P * P; e += kAT) {
    const int i = e / P, jx = e % P;
    const float m = drop_scale(drop, site, (unsigned long long)((b * H + hd) * P + i) * P + jx);
    float a = 0.0f;
    for (int d = 0; d < dh; ++d) a = __builtin_fmaf(go[i][d], v[jx][d], a);
    pd[i][jx] = pr[i * P + jx] * m;
    ds[i][jx] = a * m;
  }
  __syncthreads();
  // dV[jx][d] = sum_i pd[i][jx] * go[i][d]
  for (int e = t; e < P * dh; e += kAT) {
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
  for (int e = t; e < P * dh; e += kAT) {
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

// ---- attention on the matrix cores: one wave per (sample, head), P <= 32 tokens, head dim 32 * NT -------------------------
// q k^T and attn . v are the two dense contractions of the layer (reference: nn.MultiheadAttention inside
// models/pn_transformer/transformer.py:23-33); they run on v_mfma_f32_32x32x2_f32 over ONE 32-padded token tile, with
// everything between them in registers and NO LDS:
//   T = K (q scale)^T — computed TRANSPOSED: the accumulator lane (i = query, half h) then holds T[j][i] for the 16 keys
//   j = acc_row(r, h), so the softmax over the keys is a reduction over the lane's own 16 registers plus one exchange with
//   lane i + 32;
//   O = S V with S = dropout(softmax): the MFMA's reduction index k = key, enumerated as (step t, half h) <-> j =
//   acc_row(t, h) — so the A operand of step t IS accumulator register t of the first product (no data movement), and the
//   B operand is the V row of that key, loaded straight from global memory (coalesced over the head dimension).
// Rows / keys >= P of the padded tile are zeros / masked; padded keys (valid == 0) are masked as in the reference.
template <int NT>
__global__ __launch_bounds__(64) void attn_fwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ valid,
                                                           int P, int D, int H, Drop drop_in, unsigned site,
                                                           float* __restrict__ probs, float* __restrict__ out) {
  const Drop drop = resolve_seed(drop_in);
  constexpr int DH = 32 * NT, KH = DH / 2;
  const int b = blockIdx.x / H, hd = blockIdx.x % H, lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  const float scale = 1.0f / __builtin_sqrtf((float)DH);
  // lane (c, h): columns h * KH .. of row c of K (A operand) and of the scaled Q (B operand)
  float ka[KH], qb[KH];
  {
    const float* row = qkv + (long long)(b * P + (c < P ? c : 0)) * 3 * D + hd * DH + h * KH;
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) {
      const float4 q4 = *reinterpret_cast<const float4*>(row + 4 * v);
      const float4 k4 = *reinterpret_cast<const float4*>(row + D + 4 * v);
      const float on = c < P ? 1.0f : 0.0f;
      qb[4 * v + 0] = q4.x * scale * on, qb[4 * v + 1] = q4.y * scale * on;  // torch scales q before the product
      qb[4 * v + 2] = q4.z * scale * on, qb[4 * v + 3] = q4.w * scale * on;
      ka[4 * v + 0] = k4.x * on, ka[4 * v + 1] = k4.y * on, ka[4 * v + 2] = k4.z * on, ka[4 * v + 3] = k4.w * on;
    }
  }
  // everything else the wave will need is requested NOW, in one memory round trip with the operands above: the key mask
  // and the V rows of the second product (loaded where they are used, each was one more dependent round trip of a
  // kernel that is nothing but latency)
  float kv[16], vb[NT][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = acc_row(r, h);
    // a real part iff == 1 (network.py: part_valids == 1).  An unconditional load of a clamped position: behind `j < P &&`
    // each of the sixteen was a branch with a full wait of its own
    kv[r] = (valid[b * P + (j < P ? j : 0)] == 1.0f && j < P) ? 1.0f : 0.0f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      vb[nt][r] = j < P ? qkv[(long long)(b * P + j) * 3 * D + 2 * D + hd * DH + 32 * nt + c] : 0.0f;
  }
  f32x16 acc = {0};
#pragma unroll
  for (int s2 = 0; s2 < KH; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s2], qb[s2], acc, 0, 0, 0);
  // mask + softmax over the keys of query c
  float t[16], m = -__builtin_inff();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    t[r] = kv[r] != 0.0f ? acc[r] : -__builtin_inff();
    m = __builtin_fmaxf(m, t[r]);
  }
  m = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
  float z = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    t[r] = __expf(t[r] - m);
    z += t[r];
  }
  z += __shfl_xor(z, 32, 64);
  const float inv = 1.0f / z;
  const long long prow = ((long long)(b * H + hd) * P + c) * P;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = acc_row(r, h);
    const bool in = c < P && j < P;
    const float pj = t[r] * inv;
    const float keep = in ? drop_scale(drop, site, (unsigned long long)(prow + j)) : 0.0f;
    // the saved probability carries the dropout decision in its sign bit (p >= 0, so the bit is free; -0.0 for a dropped
    // zero): the backward kernel needs the mask in two orientations and would hash every element twice (2.5 of its 10.6 us)
    if (in) probs[prow + j] = keep != 0.0f ? pj : -pj;
    t[r] = pj * keep;
  }
  // O[i][d] = sum_j S[i][j] V[j][d]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x16 o = {0};
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) o = __builtin_amdgcn_mfma_f32_32x32x2f32(t[tt], vb[nt][tt], o, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = acc_row(r, h);
      if (i < P) out[(long long)(b * P + i) * D + hd * DH + 32 * nt + c] = o[r];
    }
  }
}

// ---- LayerNorm + q / k / v projection + attention of ONE (sample, head) in one block (D = 256, head dim 32, P <= 32) -------------
// The qkv GEMM and the attention kernel are two launches of ~9 us each that are mostly latency, and a (sample, head) needs
// only 96 of the 768 qkv columns of its own <= 32 tokens: a 32 x 96 x 256 product.  One block of 8 waves:
//   staging   the sample's token rows (a wave holds a whole 1 KB row per load: LayerNorm is two DPP sums) and the head's 96
//             weight rows (global_load_lds: straight into the LDS panel, one row per instruction) — everything requested at
//             once, the key mask, the biases and the dropout seed of a graph replay included;
//   chains    the waves split K = 256 (32 each); per wave three MFMA chains on lane-per-row fragments of the padded panel:
//             Q~, K~ TRANSPOSED (A = weight row, B = token row): lane (c, h) ends up with features acc_row(r, h) of token c —
//             exactly the operand registers of T = K Q^T (that product's k index is enumerated (step, half) <-> feature
//             acc_row(step, half)); V straight (A = token row, B = weight row): V[token acc_row(r, h)][feature c], the B
//             operand of S V;
//   reduction the 8 partial tiles meet in LDS (the panels' memory); waves 0..2 sum one tile each in wave order, add the bias
//             and write it to qkv (for backward); waves 3..6 hash the dropout keep-scales of the probabilities meanwhile;
//   attention wave 0, as attn_fwd_mfma_kernel, on registers.
// The normalised rows and their statistics are written by head 0's block as the standalone kernels wrote them.
// 13.7 us against 9.5 + 9.0 us and a launch boundary (s_memtime stamps of a block, -DMPA_QKV_EXP=9: loads 2.4 us, LayerNorm +
// panel 2.0, chains 1.9, reduction 2.9, attention 2.9).
#ifndef MPA_QKV_EXP  // timing experiments: 1 stop after the staging, 2 after the chains, 3 after the reduction (wrong results)
#define MPA_QKV_EXP 0
#endif
constexpr int kQT = 512, kQLD = 256 + 4;  // (padded rows: the lane-per-row fragment reads are conflict-free)
__global__ __launch_bounds__(kQT, 2) void attn_qkv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wqkv,
                                                           const float* __restrict__ bqkv, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           const float* __restrict__ valid, int P, int H, Drop drop,
                                                           unsigned site, float* __restrict__ qkv, float* __restrict__ ln_stats,
                                                           float* __restrict__ ln_h, float* __restrict__ ln_xcopy,
                                                           float* __restrict__ probs, float* __restrict__ out,
                                                           unsigned* __restrict__ zero, int zero_n) {
  constexpr int D = 256, DH = 32;
  // rows 0..31: the normalised token rows, 32..127: the q / k / v weight rows of this head; later (aliased) the 8 waves'
  // partial tiles: part[tile][register group][wave][lane]
  __shared__ __attribute__((aligned(16))) float panel[128 * kQLD];
  __shared__ float4 hand[2][4][64];  // the reduced k~ and v tiles on their way to wave 0
  __shared__ float keepm[16][64];    // dropout keep-scales of the probabilities, register r of lane
  float4(*part)[4][8][64] = reinterpret_cast<float4(*)[4][8][64]>(panel);
  static_assert(sizeof(float4) * 3 * 4 * 8 * 64 <= sizeof(float) * 128 * kQLD, "the partial tiles fit the panels");
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  if (zero != nullptr && blockIdx.x == 0 && (int)threadIdx.x < zero_n) zero[threadIdx.x] = 0u;
#if MPA_QKV_EXP == 9
  unsigned long long ts[8];
  int nts = 0;
#define QKV_STAMP() do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ts[nts++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define QKV_STAMP() do { } while (0)
#endif
  QKV_STAMP();
  // ---- loads, all in flight at once and fully coalesced: a wave reads one 1 KB row per instruction
  float4 xr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // token rows wave + 8 i (rows >= P: the sample's first row, dropped below)
    const int row = wave + 8 * i;
    xr[i] = *reinterpret_cast<const float4*>(x + ((long long)b * P + (row < P ? row : 0)) * D + 4 * lane);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) {  // weight rows: panel row 32 + wave + 8 i = tile (i / 4), feature wave + 8 (i % 4)
    // straight into LDS (global_load_lds_dwordx4: wave-uniform LDS row + 16 B per lane — one 1 KB row per instruction, no
    // registers in between; they count in vmcnt, which the barrier below drains)
    const int pr = wave + 8 * i, tile = pr >> 5, f = pr & 31;
    __builtin_amdgcn_global_load_lds(wqkv + ((long long)tile * D + hd * DH + f) * D + 4 * lane,
                                     (__attribute__((address_space(3))) void*)(panel + (32 + pr) * kQLD), 16, 0, 0);
  }
  const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * lane), bt = *reinterpret_cast<const float4*>(beta + 4 * lane);
  // (the dropout seed of a graph replay lives in device memory: requested here, with everything else — read where the hash
  // needs it, it is a memory round trip in the middle of the reduction)
  const Drop dl = resolve_seed(drop);
  // wave 0 also requests what its attention needs later: the key mask and the three bias runs
  // (every wave, unconditional loads of clamped positions: behind `wave == 0` and `j < P &&` the compiler made each of the
  // sixteen mask loads a branch of its own with a full wait — 9 us of dependent round trips)
  float kv[16], bq[16], bk[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = acc_row(r, h);
    kv[r] = valid[b * P + (j < P ? j : 0)];
    bq[r] = bqkv[hd * DH + j];
    bk[r] = bqkv[D + hd * DH + j];
  }
  const float bv = bqkv[2 * D + hd * DH + c];
#pragma unroll
  for (int r = 0; r < 16; ++r) kv[r] = (acc_row(r, h) < P && kv[r] == 1.0f) ? 1.0f : 0.0f;
  QKV_STAMP();
  // ---- LayerNorm: a wave holds a whole row (two passes, as ln_fwd_kernel: mean, then the variance of the deviations)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave + 8 * i;
    const float mean = wave_sum_dpp((xr[i].x + xr[i].y) + (xr[i].z + xr[i].w)) * (1.0f / (float)D);
    const float dx = xr[i].x - mean, dy = xr[i].y - mean, dz = xr[i].z - mean, dw = xr[i].w - mean;
    const float rstd = 1.0f / __builtin_sqrtf(wave_sum_dpp((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / (float)D) + eps);
    const float4 hn = make_float4(dx * rstd * gm.x + bt.x, dy * rstd * gm.y + bt.y, dz * rstd * gm.z + bt.z, dw * rstd * gm.w + bt.w);
    *reinterpret_cast<float4*>(panel + row * kQLD + 4 * lane) = hn;
    if (hd == 0 && row < P) {  // what the standalone LayerNorm would have written
      const long long tok = (long long)b * P + row;
      if (lane == 0) {
        ln_stats[2 * tok] = mean;
        ln_stats[2 * tok + 1] = rstd;
      }
      *reinterpret_cast<float4*>(ln_h + tok * D + 4 * lane) = hn;
      if (ln_xcopy != nullptr) *reinterpret_cast<float4*>(ln_xcopy + tok * D + 4 * lane) = xr[i];
    }
  }
  __syncthreads();
  QKV_STAMP();
#if MPA_QKV_EXP == 1
  return;
#endif
  // ---- three chains per wave over its eighth of K: k = 32 wave + 8 v + 4 h + e
  f32x16 aq = {0}, ak = {0}, av = {0};
  {
    const float* ph = panel + c * kQLD + 32 * wave + 4 * h;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 hv = *reinterpret_cast<const float4*>(ph + 8 * v);
      const float4 q4 = *reinterpret_cast<const float4*>(ph + 32 * kQLD + 8 * v);
      const float4 k4 = *reinterpret_cast<const float4*>(ph + 64 * kQLD + 8 * v);
      const float4 v4 = *reinterpret_cast<const float4*>(ph + 96 * kQLD + 8 * v);
      aq = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.x, hv.x, aq, 0, 0, 0);
      ak = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.x, hv.x, ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_32x32x2f32(hv.x, v4.x, av, 0, 0, 0);
      aq = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.y, hv.y, aq, 0, 0, 0);
      ak = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.y, hv.y, ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_32x32x2f32(hv.y, v4.y, av, 0, 0, 0);
      aq = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.z, hv.z, aq, 0, 0, 0);
      ak = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.z, hv.z, ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_32x32x2f32(hv.z, v4.z, av, 0, 0, 0);
      aq = __builtin_amdgcn_mfma_f32_32x32x2f32(q4.w, hv.w, aq, 0, 0, 0);
      ak = __builtin_amdgcn_mfma_f32_32x32x2f32(k4.w, hv.w, ak, 0, 0, 0);
      av = __builtin_amdgcn_mfma_f32_32x32x2f32(hv.w, v4.w, av, 0, 0, 0);
    }
  }
  QKV_STAMP();
#if MPA_QKV_EXP == 2
  if (aq[0] + ak[1] + av[2] != 12345.0f) return;
#endif
  __syncthreads();  // every wave is done with the panels: their memory takes the partial tiles
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    part[0][g][wave][lane] = make_float4(aq[4 * g], aq[4 * g + 1], aq[4 * g + 2], aq[4 * g + 3]);
    part[1][g][wave][lane] = make_float4(ak[4 * g], ak[4 * g + 1], ak[4 * g + 2], ak[4 * g + 3]);
    part[2][g][wave][lane] = make_float4(av[4 * g], av[4 * g + 1], av[4 * g + 2], av[4 * g + 3]);
  }
  __syncthreads();
  if (wave >= 7) return;  // (the barrier below is reached by the seven waves that are left: finished waves do not count)
  const bool on = c < P;
  const long long tok = (long long)b * P + (on ? c : 0);
  const long long prow = ((long long)(b * H + hd) * P + c) * P;
  float tl[16];
  if (wave >= 3) {
    // the dropout keep-scales of the probabilities (a 64-bit hash per element), four registers' worth per wave 3..6 —
    // beside the reduction instead of inside wave 0's chain
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * (wave - 3) + e, j = acc_row(r, h);
      keepm[r][lane] = (c < P && j < P) ? drop_scale(dl, site, (unsigned long long)(prow + j)) : 0.0f;
    }
  } else {
    // wave t sums tile t over the 8 waves, in wave order; waves 1 and 2 write their tile (k~, v) out and hand it to wave 0
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 a = part[wave][g][0][lane];
#pragma unroll
      for (int w = 1; w < 8; ++w) {
        const float4 p4 = part[wave][g][w][lane];
        a.x += p4.x, a.y += p4.y, a.z += p4.z, a.w += p4.w;
      }
      if (wave == 0) {
        a.x += bq[4 * g], a.y += bq[4 * g + 1], a.z += bq[4 * g + 2], a.w += bq[4 * g + 3];
      } else if (wave == 1) {
        a.x += bk[4 * g], a.y += bk[4 * g + 1], a.z += bk[4 * g + 2], a.w += bk[4 * g + 3];
      } else {
        a.x += bv, a.y += bv, a.z += bv, a.w += bv;
      }
      tl[4 * g] = a.x, tl[4 * g + 1] = a.y, tl[4 * g + 2] = a.z, tl[4 * g + 3] = a.w;
      if (wave > 0) hand[wave - 1][g][lane] = a;
    }
  }
  __syncthreads();
  if (wave < 3) {  // the tiles go out behind the barrier (in front of it, it would wait for the stores to be acknowledged)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (wave < 2) {  // q~ / k~: four consecutive features 8 g + 4 h .. of token c
        if (on)
          *reinterpret_cast<float4*>(qkv + tok * 3 * D + wave * D + hd * DH + 8 * g + 4 * h) =
              make_float4(tl[4 * g], tl[4 * g + 1], tl[4 * g + 2], tl[4 * g + 3]);
      } else {  // v: row acc_row(r, h), feature c
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = acc_row(4 * g + e, h);
          if (j < P) qkv[((long long)b * P + j) * 3 * D + 2 * D + hd * DH + c] = tl[4 * g + e];
        }
      }
    }
  }
  if (wave != 0) return;
  QKV_STAMP();
#if MPA_QKV_EXP == 3
  if (tl[0] != 12345.0f) return;
#endif
  float ka[16], qb[16], vb[16];
  const float scale = 1.0f / __builtin_sqrtf((float)DH);
  const float onf = on ? 1.0f : 0.0f;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 k4 = hand[0][g][lane], v4 = hand[1][g][lane];
    const float kk[4] = {k4.x, k4.y, k4.z, k4.w}, vq[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = 4 * g + e;
      qb[r] = tl[r] * scale * onf;  // torch scales q before the product
      ka[r] = kk[e] * onf;
      vb[r] = acc_row(r, h) < P ? vq[e] : 0.0f;
    }
  }
  f32x16 acc = {0};
#pragma unroll
  for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[s2], qb[s2], acc, 0, 0, 0);
  // mask + softmax over the keys of query c (as attn_fwd_mfma_kernel)
  float t[16], m = -__builtin_inff();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    t[r] = kv[r] != 0.0f ? acc[r] : -__builtin_inff();
    m = __builtin_fmaxf(m, t[r]);
  }
  m = __builtin_fmaxf(m, __shfl_xor(m, 32, 64));
  float z = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    t[r] = __expf(t[r] - m);
    z += t[r];
  }
  z += __shfl_xor(z, 32, 64);
  const float inv = 1.0f / z;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int j = acc_row(r, h);
    const bool in = c < P && j < P;
    const float pj = t[r] * inv;
    const float keep = keepm[r][lane];
    if (in) probs[prow + j] = keep != 0.0f ? pj : -pj;  // (the dropout decision rides in the sign bit)
    t[r] = pj * keep;
  }
  f32x16 o = {0};
#pragma unroll
  for (int tt = 0; tt < 16; ++tt) o = __builtin_amdgcn_mfma_f32_32x32x2f32(t[tt], vb[tt], o, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = acc_row(r, h);
    if (i < P) out[((long long)b * P + i) * D + hd * DH + c] = o[r];
  }
#if MPA_QKV_EXP == 9
  QKV_STAMP();
  if (blockIdx.x == 100 && lane == 0)
    printf("attn_qkv stamps (s_memtime ticks): loads %llu, LN+stage %llu, chains %llu, reduce %llu, attention %llu\n", ts[1] - ts[0],
           ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4]);
#endif
}

// Backward of the same.  dP = dO V^T is taken in BOTH orientations (two chains over the head dimension): lane = query i
// (softmax backward needs a row's keys together) and lane = key j (dK and dV reduce over the queries).  With
// dS = P (dP m - <dP m, P>_row):  dQ = scale dS K,  dK = scale dS^T Q,  dV = (P m)^T dO — three chains over the token
// index whose A operands are accumulator registers again; a row's inner product travels from the query orientation to the
// key orientation with one cross-lane read per register.
template <int NT>
__global__ __launch_bounds__(64) void attn_bwd_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                           const float* __restrict__ dout, int P, int D, int H, Drop drop,
                                                           unsigned site, float* __restrict__ dqkv) {
  constexpr int DH = 32 * NT, KH = DH / 2;
  const int b = blockIdx.x / H, hd = blockIdx.x % H, lane = threadIdx.x, c = lane & 31, h = lane >> 5;
  const float scale = 1.0f / __builtin_sqrtf((float)DH);
  float va[KH], ga[KH];  // lane (c, h): columns h * KH .. of row c of V and of dO
  {
    const long long tok = b * P + (c < P ? c : 0);
    const float* vrow = qkv + tok * 3 * D + 2 * D + hd * DH + h * KH;
    const float* grow = dout + tok * D + hd * DH + h * KH;
    const float on = c < P ? 1.0f : 0.0f;
#pragma unroll
    for (int v = 0; v < KH / 4; ++v) {
      const float4 v4 = *reinterpret_cast<const float4*>(vrow + 4 * v);
      const float4 g4 = *reinterpret_cast<const float4*>(grow + 4 * v);
      va[4 * v + 0] = v4.x * on, va[4 * v + 1] = v4.y * on, va[4 * v + 2] = v4.z * on, va[4 * v + 3] = v4.w * on;
      ga[4 * v + 0] = g4.x * on, ga[4 * v + 1] = g4.y * on, ga[4 * v + 2] = g4.z * on, ga[4 * v + 3] = g4.w * on;
    }
  }
  // every other operand is requested now, in the same memory round trip (see the forward kernel): the probabilities in
  // both orientations and, for the first 32 columns of the head, the K / Q / dO rows of the three token-index chains
  const long long pbase = (long long)(b * H + hd) * P * P;
  const long long row0 = (long long)b * P;
  float pa_[16], pb_[16], kb0[16], qb0[16], gb0[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int x = acc_row(r, h);
    const bool in = c < P && x < P;
    pa_[r] = in ? probs[pbase + (long long)c * P + x] : -0.0f;  // sign bit set: dropped (see the forward kernel)
    pb_[r] = in ? probs[pbase + (long long)x * P + c] : -0.0f;
    const long long tok = row0 + (x < P ? x : 0);
    const float on = x < P ? 1.0f : 0.0f;
    qb0[r] = qkv[tok * 3 * D + hd * DH + c] * on;
    kb0[r] = qkv[tok * 3 * D + D + hd * DH + c] * on;
    gb0[r] = dout[tok * D + hd * DH + c] * on;
  }
  const float keep = drop.p > 0.0f ? drop.scale : 1.0f;  // keep-scale of the elements the forward pass did not drop
  f32x16 ga_acc = {0}, gb_acc = {0};  // lane = query: dP[i][j], j = acc_row(r, h);  lane = key: dP[i][j], i = acc_row(r, h)
#pragma unroll
  for (int s2 = 0; s2 < KH; ++s2) {
    ga_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s2], ga[s2], ga_acc, 0, 0, 0);
    gb_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[s2], va[s2], gb_acc, 0, 0, 0);
  }
  // query orientation: dS[c][j] and the row's inner product
  float dsa[16], rowdot = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float pa = __builtin_fabsf(pa_[r]);
    const float dp = ga_acc[r] * (__float_as_uint(pa_[r]) >> 31 ? 0.0f : keep);
    rowdot = __builtin_fmaf(dp, pa, rowdot);
    dsa[r] = dp;
    ga_acc[r] = pa;
  }
  rowdot += __shfl_xor(rowdot, 32, 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) dsa[r] = ga_acc[r] * (dsa[r] - rowdot);
  // key orientation: dS[i][c] and (P m)[i][c]
  float dsb[16], pdb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = acc_row(r, h);
    const float pb = __builtin_fabsf(pb_[r]);
    const float mk = __float_as_uint(pb_[r]) >> 31 ? 0.0f : keep;
    const float rd = __shfl(rowdot, i, 64);  // (lane i holds query i's inner product, both halves)
    dsb[r] = pb * (gb_acc[r] * mk - rd);
    pdb[r] = pb * mk;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = hd * DH + 32 * nt + c;
    float kb[16], qb[16], gb[16];
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      if (nt == 0) {
        qb[tt] = qb0[tt], kb[tt] = kb0[tt], gb[tt] = gb0[tt];
      } else {
        const int x = acc_row(tt, h);
        const long long tok = row0 + (x < P ? x : 0);
        const float on = x < P ? 1.0f : 0.0f;
        qb[tt] = qkv[tok * 3 * D + col] * on;
        kb[tt] = qkv[tok * 3 * D + D + col] * on;
        gb[tt] = dout[tok * D + col] * on;
      }
    }
    f32x16 dq = {0}, dk = {0}, dv = {0};
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      dq = __builtin_amdgcn_mfma_f32_32x32x2f32(dsa[tt], kb[tt], dq, 0, 0, 0);  // dQ[i] = sum_j dS[i][j] K[j]
      dk = __builtin_amdgcn_mfma_f32_32x32x2f32(dsb[tt], qb[tt], dk, 0, 0, 0);  // dK[j] = sum_i dS[i][j] Q[i]
      dv = __builtin_amdgcn_mfma_f32_32x32x2f32(pdb[tt], gb[tt], dv, 0, 0, 0);  // dV[j] = sum_i (P m)[i][j] dO[i]
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int x = acc_row(r, h);
      if (x < P) {
        float* dst = dqkv + (row0 + x) * 3 * D + col;
        dst[0] = dq[r] * scale;
        dst[D] = dk[r] * scale;
        dst[2 * D] = dv[r];
      }
    }
  }
}

// ---- LayerNorm-2 backward + d o product + attention backward of ONE (sample, head) in one block (D = 256, head dim 32, P <= 32) ----
// The backward twin of attn_qkv_fwd_kernel.  The attention's gradient needs d o [tokens of the sample][32 columns of the head]
// = (masked d x_mid) [tokens][256] . Wo[256][head's columns], and d x_mid is the LayerNorm backward of rows the block can
// normalise itself.  Eight waves: rows by coalesced loads (a wave holds a whole row: the two row means are DPP sums), the
// dropout mask regenerated per element (as the LayerNorm-backward prologue of the d o GEMM did: the same redundancy, 8 heads
// per sample against 8 column tiles per row tile), the masked rows and the head's Wo slice (transposed while it is stored) in
// padded LDS panels, K = 256 split over the waves, TWO MFMA chains per wave: d o straight (A = token row, B = weight row:
// lane (c, h) ends with d o[token acc_row(r, h)][feature c], the B operand of the three token-index chains) and d o
// TRANSPOSED (lane = token c, features acc_row(r, h): the operand of the two head-dimension chains, whose k index is
// enumerated in that order on both operands).  Partial tiles meet in LDS; wave 0 runs attn_bwd_mfma_kernel's arithmetic.
// Head 0's block writes what the standalone LayerNorm backward would have written: d x_mid, its masked copy, and the
// sample's dgamma / dbeta partial row (ln_part[sample]: the reduction kernel is told B rows for this site).
// 14.6 us against 9.0 + 7.8 us and a launch boundary (-DMPA_QKV_EXP=9 stamps of a block, in us: requests 3.7 — the saved
// probabilities come from HBM —, LayerNorm backward + masks + panels 2.6, chains 0.6, reduction 2.2, attention 3.9).
// CHAIN: the kernel goes on with what follows the attention's gradient on the way down — rows of a sample again:
//   d(LN1 output)[tokens][256] = dqkv[tokens][768] . Wqkv[768][256] is a sum over the heads, so every (sample, head) block adds
//   its 96-column slice's product (its d q | d k | d v tiles go back through LDS as lane-per-row operands, the weight rows
//   arrive as coalesced B operands requested at the top of the kernel; wave w owns output columns 32 w ..) as a partial tile in
//   `hp`, written through to agent scope and ordered by a per-sample ticket (tf_gemm.h's split-K hand-over); the block that
//   takes the sample's LAST ticket adds the eight partial tiles in head order and runs LN1's backward on the rows — the
//   residual is the d x_mid this block computed itself — writing the layer's input gradient, its masked copy for the layer
//   below and the sample's dgamma / dbeta partial row.  Replaces the split-K GEMM and the LayerNorm-backward launch behind the
//   attention (8.8 + 5.6 us and two boundaries per layer).
struct ChainArgs {
  const float* wqkv;    // [3 D][D]
  const float* x_in;    // [M][D] LN1's input
  const float* stats1;  // [M][2]
  const float* gamma1;  // [D]
  float* g_in;          // [M][D] gradient at the layer's input
  float* gd_next;       // nullable [M][D]: g_in under the dropout mask of site_next (the layer below's FFN output)
  unsigned site_next;
  float* ln1_part;      // [B][2 D]
  float* hp;            // [B][H][32][D] partial tiles of d(LN1 output)
  unsigned* ticket;     // [B], zero between launches
};
typedef float tf_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tf_f4 ld_agent4(const float* p) {  // past this XCD's L2 (the other blocks' write-through stores)
  tf_f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// the loads above are invisible to the compiler's own wait insertion: this wait names the registers, so that no use of them
// is scheduled in front of it
__device__ __forceinline__ void wait_agent8(tf_f4 (&v)[8]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
               :
               : "memory");
}

constexpr int kBT = 512;
template <bool CHAIN>
__global__ __launch_bounds__(kBT, 2) void attn_do_bwd_kernel(
    const float* __restrict__ dh, const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ resid, float* __restrict__ ln_dx, float* __restrict__ ln_dxdrop, float* __restrict__ ln_part,
    unsigned ln_site, const float* __restrict__ wo, const float* __restrict__ qkv, const float* __restrict__ probs, int P, int H,
    Drop drop_in, unsigned site, float* __restrict__ dqkv, const ChainArgs ch) {
  constexpr int D = 256, DH = 32;
  __shared__ float dq_s[CHAIN ? 3 : 1][32][36];  // CHAIN: the head's d q | d k | d v tiles, row = token
  __shared__ int last_s;
  __shared__ float wq0[CHAIN ? 48 : 1][64];  // CHAIN: wave 0's weight operands, fetched by wave 7
  __shared__ __attribute__((aligned(16))) float panel[64 * kQLD];  // rows 0..31: masked d x rows; 32..63: Wo^T slice [feature][k]
  __shared__ float4 hand[4][64];
  __shared__ float colp[8][2][D];  // head 0: the waves' column partials of dh xhat | dh
  float4(*part)[4][8][64] = reinterpret_cast<float4(*)[4][8][64]>(panel);
  static_assert(sizeof(float4) * 2 * 4 * 8 * 64 <= sizeof(float) * 64 * kQLD, "the partial tiles fit the panels");
  const Drop drop = resolve_seed(drop_in);
#if MPA_QKV_EXP == 9
  unsigned long long ts[8];
  int nts = 0;
#endif
  QKV_STAMP();
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const long long row0 = (long long)b * P;
  // ---- requests: the block's rows (wave + 8 i), the weight slice, and what wave 0's attention needs
  float4 dh4[4], x4[4], r4[4];
  float mean[4], rstd[4];
  const float* rsrc = resid != nullptr ? resid : x;  // (no residual: a valid address, the value is dropped — a conditional load is a
                                                     // branch with a full wait per row)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave + 8 * i;
    const long long tok = row0 + (row < P ? row : 0);
    dh4[i] = *reinterpret_cast<const float4*>(dh + tok * D + 4 * lane);
    x4[i] = *reinterpret_cast<const float4*>(x + tok * D + 4 * lane);
    r4[i] = *reinterpret_cast<const float4*>(rsrc + tok * D + 4 * lane);
    mean[i] = stats[2 * tok];
    rstd[i] = stats[2 * tok + 1];
  }
  const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * lane);
  float4 wv[4];  // Wo rows k = (t + 512 i) / 8, columns hd DH + 4 ((t + 512 i) % 8) .. + 3
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = threadIdx.x + kBT * i;
    wv[i] = *reinterpret_cast<const float4*>(wo + (long long)(f >> 3) * D + hd * DH + 4 * (f & 7));
  }
  // CHAIN: B operands of the chain product, Wqkv[t D + hd DH + k][32 wv + c] with k = 8 v + 4 h + e (the enumeration of the A
  // reads).  They are requested behind the hand-over barrier, by the seven waves that idle through wave 0's attention — wave 7
  // also fetches wave 0's tile and parks it in LDS (held from the top of the kernel they cost wave 0 its registers: spills)
  float wq[CHAIN ? 48 : 1];
  auto load_wq = [&](int wvt, float (&dst)[CHAIN ? 48 : 1]) {
    if constexpr (CHAIN) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            dst[16 * t + 4 * v + e] = ch.wqkv[(long long)(t * D + hd * DH + 8 * v + 4 * h + e) * D + 32 * wvt + c];
    }
  };
  const bool on = c < P;
  const long long tokc = row0 + (on ? c : 0);
  const long long pbase = (long long)(b * H + hd) * P * P;
  float va[16], pa_[16], pb_[16], kb0[16], qb0[16];
  if (wave == 0) {  // (wave-uniform; unconditional loads of clamped positions inside)
    const float onf = on ? 1.0f : 0.0f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // V[token c][features 8 g + 4 h ..]: the enumeration of the transposed d o tile
      const float4 v4 = *reinterpret_cast<const float4*>(qkv + tokc * 3 * D + 2 * D + hd * DH + 8 * g + 4 * h);
      va[4 * g] = v4.x * onf, va[4 * g + 1] = v4.y * onf, va[4 * g + 2] = v4.z * onf, va[4 * g + 3] = v4.w * onf;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int xr = acc_row(r, h), xc = xr < P ? xr : 0;
      const bool in = c < P && xr < P;
      const float p1 = probs[pbase + (long long)(on ? c : 0) * P + xc], p2 = probs[pbase + (long long)xc * P + (on ? c : 0)];
      pa_[r] = in ? p1 : -0.0f;  // sign bit set: dropped (see the forward kernel)
      pb_[r] = in ? p2 : -0.0f;
      const long long tk = row0 + xc;
      const float o2 = xr < P ? 1.0f : 0.0f;
      qb0[r] = qkv[tk * 3 * D + hd * DH + c] * o2;
      kb0[r] = qkv[tk * 3 * D + D + hd * DH + c] * o2;
    }
  }
  QKV_STAMP();
  // ---- LayerNorm backward of the rows, dropout mask, panel
  float4 pxh = make_float4(0.f, 0.f, 0.f, 0.f), pg = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 vrow[4];  // d x_mid of the block's rows (CHAIN: the residual of LN1's backward)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave + 8 * i;
    const bool live = row < P;
    const float4 d4 = dh4[i];
    const float4 xh = make_float4((x4[i].x - mean[i]) * rstd[i], (x4[i].y - mean[i]) * rstd[i], (x4[i].z - mean[i]) * rstd[i],
                                  (x4[i].w - mean[i]) * rstd[i]);
    const float dx = d4.x * gm.x, dy = d4.y * gm.y, dz = d4.z * gm.z, dw = d4.w * gm.w;
    const float s1 = wave_sum_dpp((dx + dy) + (dz + dw)) * (1.0f / (float)D);
    const float s2 = wave_sum_dpp((dx * xh.x + dy * xh.y) + (dz * xh.z + dw * xh.w)) * (1.0f / (float)D);
    float4 v = resid != nullptr ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    v.x += rstd[i] * (dx - s1 - xh.x * s2);
    v.y += rstd[i] * (dy - s1 - xh.y * s2);
    v.z += rstd[i] * (dz - s1 - xh.z * s2);
    v.w += rstd[i] * (dw - s1 - xh.w * s2);
    vrow[i] = v;
    float4 m = v;
    const long long o = (row0 + (live ? row : 0)) * D + 4 * lane;
    if (ln_dxdrop != nullptr) {
      m.x *= drop_scale(drop, ln_site, (unsigned long long)o);
      m.y *= drop_scale(drop, ln_site, (unsigned long long)o + 1);
      m.z *= drop_scale(drop, ln_site, (unsigned long long)o + 2);
      m.w *= drop_scale(drop, ln_site, (unsigned long long)o + 3);
    }
    *reinterpret_cast<float4*>(panel + row * kQLD + 4 * lane) = live ? m : make_float4(0.f, 0.f, 0.f, 0.f);
    if (hd == 0 && live) {
      *reinterpret_cast<float4*>(ln_dx + o) = v;
      if (ln_dxdrop != nullptr) *reinterpret_cast<float4*>(ln_dxdrop + o) = m;
      pxh.x += d4.x * xh.x, pxh.y += d4.y * xh.y, pxh.z += d4.z * xh.z, pxh.w += d4.w * xh.w;
      pg.x += d4.x, pg.y += d4.y, pg.z += d4.z, pg.w += d4.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // Wo^T slice: panel row 32 + feature, column k
    const int f = threadIdx.x + kBT * i, k = f >> 3, n = 4 * (f & 7);
    panel[(32 + n) * kQLD + k] = wv[i].x;
    panel[(33 + n) * kQLD + k] = wv[i].y;
    panel[(34 + n) * kQLD + k] = wv[i].z;
    panel[(35 + n) * kQLD + k] = wv[i].w;
  }
  if (hd == 0) {
    *reinterpret_cast<float4*>(&colp[wave][0][4 * lane]) = pxh;
    *reinterpret_cast<float4*>(&colp[wave][1][4 * lane]) = pg;
  }
  __syncthreads();
  if (hd == 0) {  // the sample's partial row of dgamma | dbeta: the eight waves in order (thread = column of [2 D])
    const int k = threadIdx.x;
    float sm = 0.0f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sm += colp[w][k >> 8][k & 255];
    ln_part[(long long)b * 2 * D + k] = sm;
  }
  QKV_STAMP();
  // ---- two chains per wave over its eighth of K: k = 32 wave + 8 v + 4 h + e
  f32x16 as = {0}, at = {0};
  {
    const float* pm = panel + c * kQLD + 32 * wave + 4 * h;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 m4 = *reinterpret_cast<const float4*>(pm + 8 * v);
      const float4 w4 = *reinterpret_cast<const float4*>(pm + 32 * kQLD + 8 * v);
      as = __builtin_amdgcn_mfma_f32_32x32x2f32(m4.x, w4.x, as, 0, 0, 0);
      at = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.x, m4.x, at, 0, 0, 0);
      as = __builtin_amdgcn_mfma_f32_32x32x2f32(m4.y, w4.y, as, 0, 0, 0);
      at = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.y, m4.y, at, 0, 0, 0);
      as = __builtin_amdgcn_mfma_f32_32x32x2f32(m4.z, w4.z, as, 0, 0, 0);
      at = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.z, m4.z, at, 0, 0, 0);
      as = __builtin_amdgcn_mfma_f32_32x32x2f32(m4.w, w4.w, as, 0, 0, 0);
      at = __builtin_amdgcn_mfma_f32_32x32x2f32(w4.w, m4.w, at, 0, 0, 0);
    }
  }
  QKV_STAMP();
  __syncthreads();  // every wave is done with the panels: their memory takes the partial tiles
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    part[0][g][wave][lane] = make_float4(as[4 * g], as[4 * g + 1], as[4 * g + 2], as[4 * g + 3]);
    part[1][g][wave][lane] = make_float4(at[4 * g], at[4 * g + 1], at[4 * g + 2], at[4 * g + 3]);
  }
  __syncthreads();
  if constexpr (!CHAIN) {
    if (wave >= 2) return;  // (the barrier below is reached by the two waves that are left)
  }
  float tl[16];  // wave 0: d o transposed (the head-dimension chains' operand); wave 1: d o straight, handed over
  if (wave < 2) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 a = part[1 - wave][g][0][lane];
#pragma unroll
      for (int w = 1; w < 8; ++w) {
        const float4 p4 = part[1 - wave][g][w][lane];
        a.x += p4.x, a.y += p4.y, a.z += p4.z, a.w += p4.w;
      }
      tl[4 * g] = a.x, tl[4 * g + 1] = a.y, tl[4 * g + 2] = a.z, tl[4 * g + 3] = a.w;
      if (wave == 1) hand[g][lane] = a;
    }
  }
  __syncthreads();
  if constexpr (!CHAIN) {
    if (wave != 0) return;
  }
  if constexpr (CHAIN) {
    if (wave != 0) load_wq(wave, wq);
    if (wave == 7) {
      float w0[48];
      load_wq(0, w0);
#pragma unroll
      for (int k = 0; k < 48; ++k) wq0[k][lane] = w0[k];
    }
  }
  if (wave == 0) {
  QKV_STAMP();
  float gb0[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 t4 = hand[g][lane];
    gb0[4 * g] = t4.x, gb0[4 * g + 1] = t4.y, gb0[4 * g + 2] = t4.z, gb0[4 * g + 3] = t4.w;
  }
  // ---- attention backward (attn_bwd_mfma_kernel<1>, with the head-dimension index enumerated in accumulator order)
  const float scale = 1.0f / __builtin_sqrtf((float)DH);
  const float keep = drop.p > 0.0f ? drop.scale : 1.0f;
  f32x16 ga_acc = {0}, gb_acc = {0};
#pragma unroll
  for (int s2 = 0; s2 < 16; ++s2) {
    ga_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[s2], tl[s2], ga_acc, 0, 0, 0);
    gb_acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tl[s2], va[s2], gb_acc, 0, 0, 0);
  }
  float dsa[16], rowdot = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float pa = __builtin_fabsf(pa_[r]);
    const float dp = ga_acc[r] * (__float_as_uint(pa_[r]) >> 31 ? 0.0f : keep);
    rowdot = __builtin_fmaf(dp, pa, rowdot);
    dsa[r] = dp;
    ga_acc[r] = pa;
  }
  rowdot += __shfl_xor(rowdot, 32, 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) dsa[r] = ga_acc[r] * (dsa[r] - rowdot);
  float dsb[16], pdb[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = acc_row(r, h);
    const float pb = __builtin_fabsf(pb_[r]);
    const float mk = __float_as_uint(pb_[r]) >> 31 ? 0.0f : keep;
    const float rd = __shfl(rowdot, i, 64);
    dsb[r] = pb * (gb_acc[r] * mk - rd);
    pdb[r] = pb * mk;
  }
  f32x16 dq = {0}, dk = {0}, dv = {0};
#pragma unroll
  for (int tt = 0; tt < 16; ++tt) {
    dq = __builtin_amdgcn_mfma_f32_32x32x2f32(dsa[tt], kb0[tt], dq, 0, 0, 0);
    dk = __builtin_amdgcn_mfma_f32_32x32x2f32(dsb[tt], qb0[tt], dk, 0, 0, 0);
    dv = __builtin_amdgcn_mfma_f32_32x32x2f32(pdb[tt], gb0[tt], dv, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int xr = acc_row(r, h);
    const float q_ = dq[r] * scale, k_ = dk[r] * scale, v_ = dv[r];
    if (xr < P) {
      float* dst = dqkv + (row0 + xr) * 3 * D + hd * DH + c;
      dst[0] = q_;
      dst[D] = k_;
      dst[2 * D] = v_;
    }
    if constexpr (CHAIN) {  // (rows >= P are zeros: their d o rows were)
      dq_s[0][xr][c] = q_;
      dq_s[1][xr][c] = k_;
      dq_s[2][xr][c] = v_;
    }
  }
  }  // wave 0
#if MPA_QKV_EXP == 9
  if (wave == 0) QKV_STAMP();
  if (blockIdx.x == 100 && lane == 0 && wave == 0)
    printf("attn_do_bwd stamps (s_memtime ticks): loads %llu, LN backward + panels %llu, chains %llu, reduce %llu, attention %llu\n",
           ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4]);
#endif
  if constexpr (CHAIN) {
    __syncthreads();  // the head's d q | d k | d v tiles are in LDS
    // what the sample's last block needs for LN1's backward (requested by every block: one of eight uses it, none waits —
    // the product and the ticket below cover the latency)
    float4 x1[4];
    float mean1[4], rstd1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave + 8 * i;
      const long long tok = row0 + (row < P ? row : 0);
      x1[i] = *reinterpret_cast<const float4*>(ch.x_in + tok * D + 4 * lane);
      mean1[i] = ch.stats1[2 * tok];
      rstd1[i] = ch.stats1[2 * tok + 1];
    }
    const float4 g1 = *reinterpret_cast<const float4*>(ch.gamma1 + 4 * lane);
    if (wave == 0) {
#pragma unroll
      for (int k = 0; k < 48; ++k) wq[k] = wq0[k][lane];
    }
#if MPA_QKV_EXP == 9
    unsigned long long tc[6];
    tc[0] = __builtin_amdgcn_s_memtime();
#endif
    f32x16 acc = {0};
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float4 a4 = *reinterpret_cast<const float4*>(&dq_s[t][c][8 * v + 4 * h]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, wq[16 * t + 4 * v], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, wq[16 * t + 4 * v + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, wq[16 * t + 4 * v + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, wq[16 * t + 4 * v + 3], acc, 0, 0, 0);
      }
    float* mine = ch.hp + ((long long)(b * H + hd) * 32) * D + 32 * wave + c;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int xr = acc_row(r, h);
      if (xr < P) __hip_atomic_store(mine + (long long)xr * D, acc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#if MPA_QKV_EXP == 9
    tc[1] = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // written through and acknowledged before the ticket is taken
#if MPA_QKV_EXP == 9
    tc[2] = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = atomicAdd(ch.ticket + b, 1u);
      last_s = old == (unsigned)(H - 1);
      if (old == (unsigned)(H - 1)) ch.ticket[b] = 0u;  // ready for the next launch
    }
    __syncthreads();
#if MPA_QKV_EXP == 9
    tc[3] = __builtin_amdgcn_s_memtime();
#endif
    if (!last_s) return;
    // ---- the sample's last block: the eight heads' partial tiles in head order, then LN1's backward on the rows
    float4 dl[4];
    {
      tf_f4 hv[4][8];  // all 32 requests of the thread in flight together (row by row it was four dependent round trips)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wave + 8 * i, rr = row < P ? row : 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) hv[i][q] = ld_agent4(ch.hp + ((long long)(b * H + (q < H ? q : 0)) * 32 + rr) * D + 4 * lane);
      }
      wait_agent8(hv[0]);
      wait_agent8(hv[1]);  // (vmcnt is already zero: these only name the registers)
      wait_agent8(hv[2]);
      wait_agent8(hv[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tf_f4 a = hv[i][0];
#pragma unroll
        for (int q = 1; q < 8; ++q)
          if (q < H) a += hv[i][q];
        dl[i] = make_float4(a.x, a.y, a.z, a.w);
      }
    }
#if MPA_QKV_EXP == 9
    tc[4] = __builtin_amdgcn_s_memtime();
#endif
    float4 qxh = make_float4(0.f, 0.f, 0.f, 0.f), qg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wave + 8 * i;
      if (row >= P) continue;  // (wave-uniform)
      const float4 d4 = dl[i];
      const float4 xh = make_float4((x1[i].x - mean1[i]) * rstd1[i], (x1[i].y - mean1[i]) * rstd1[i],
                                    (x1[i].z - mean1[i]) * rstd1[i], (x1[i].w - mean1[i]) * rstd1[i]);
      const float dx = d4.x * g1.x, dy = d4.y * g1.y, dz = d4.z * g1.z, dw = d4.w * g1.w;
      const float s1 = wave_sum_dpp((dx + dy) + (dz + dw)) * (1.0f / (float)D);
      const float s2 = wave_sum_dpp((dx * xh.x + dy * xh.y) + (dz * xh.z + dw * xh.w)) * (1.0f / (float)D);
      float4 v = vrow[i];
      v.x += rstd1[i] * (dx - s1 - xh.x * s2);
      v.y += rstd1[i] * (dy - s1 - xh.y * s2);
      v.z += rstd1[i] * (dz - s1 - xh.z * s2);
      v.w += rstd1[i] * (dw - s1 - xh.w * s2);
      const long long o = (row0 + row) * D + 4 * lane;
      *reinterpret_cast<float4*>(ch.g_in + o) = v;
      if (ch.gd_next != nullptr) {
        float4 m = v;
        m.x *= drop_scale(drop, ch.site_next, (unsigned long long)o);
        m.y *= drop_scale(drop, ch.site_next, (unsigned long long)o + 1);
        m.z *= drop_scale(drop, ch.site_next, (unsigned long long)o + 2);
        m.w *= drop_scale(drop, ch.site_next, (unsigned long long)o + 3);
        *reinterpret_cast<float4*>(ch.gd_next + o) = m;
      }
      qxh.x += d4.x * xh.x, qxh.y += d4.y * xh.y, qxh.z += d4.z * xh.z, qxh.w += d4.w * xh.w;
      qg.x += d4.x, qg.y += d4.y, qg.z += d4.z, qg.w += d4.w;
    }
    *reinterpret_cast<float4*>(&colp[wave][0][4 * lane]) = qxh;  // (colp is free: head 0's sums left it two barriers ago)
    *reinterpret_cast<float4*>(&colp[wave][1][4 * lane]) = qg;
    __syncthreads();
    {
      const int k = threadIdx.x;
      float sm = 0.0f;
#pragma unroll
      for (int w = 0; w < 8; ++w) sm += colp[w][k >> 8][k & 255];
      ch.ln1_part[(long long)b * 2 * D + k] = sm;
    }
#if MPA_QKV_EXP == 9
    if (b == 12 && threadIdx.x == 0)
      printf("attn chain stamps (ticks): product + stores %llu, store acknowledge %llu, ticket %llu, partial tiles %llu, LN1 backward %llu\n",
             tc[1] - tc[0], tc[2] - tc[1], tc[3] - tc[2], tc[4] - tc[3], __builtin_amdgcn_s_memtime() - tc[4]);
#endif
  }
}

// ---- LayerNorm backward -----------------------------------------------------------------------------------------------------
// dx[row] = resid[row] + rstd * (dh*gamma - mean(dh*gamma) - xhat * mean(dh*gamma*xhat));  one wave per row.
// Per-block partial sums of dgamma = sum dh*xhat and dbeta = sum dh go to part[block][2][D].
// dx_drop (nullable): also writes dx with the dropout mask of `site` applied — dx is the gradient at the output of a
// residual branch that ended in dropout, and BOTH consumers of the masked gradient (the branch's weight gradient
// and its input-gradient GEMM) would otherwise regenerate the mask once per output tile (a 64-bit hash per
// element and tile: 14 of the 32 us of a layer's weight-gradient launch).
__global__ __launch_bounds__(kT) void ln_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ x,
                                                    const float* __restrict__ stats,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ resid, int M, int D,
                                                    float* __restrict__ dx, float* __restrict__ part,
                                                    const Drop drop_in, unsigned site, float* __restrict__ dx_drop) {
  const Drop drop = resolve_seed(drop_in);
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
      const float v = (resid ? resid[o + k] : 0.0f) + rstd * (g * gamma[k] - s1 - xh * s2);
      dx[o + k] = v;
      if (dx_drop != nullptr) dx_drop[o + k] = v * drop_scale(drop, site, (unsigned long long)(o + k));
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

// dgamma[k] = sum_blocks part[b][0][k], dbeta likewise, for EVERY LayerNorm of the encoder in one launch (the 2L + 1
// partial tables sit side by side; nine launches of 4.5 us were 7 % of the backward).  grid = (2D/64, sites), block 1024:
// 16 groups of 64 columns, group q sums blocks q, q+16, ... and the groups meet in LDS (fixed order).
struct LnSites {
  float* dgamma[2 * 16 + 1];
  float* dbeta[2 * 16 + 1];
  int blocks[2 * 16 + 1];  // rows of the site's partial table (the fused backward leaves one per 32-row tile)
};
__global__ __launch_bounds__(1024) void ln_reduce_kernel(const float* __restrict__ part, long long site_stride, int D,
                                                         const LnSites out) {
  __shared__ float sm[16][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6, k = blockIdx.x * 64 + c, blocks = out.blocks[blockIdx.y];
  const float* p = part + (long long)blockIdx.y * site_stride;
  float s = 0.0f;
#pragma unroll 4
  for (int b = q; b < blocks; b += 16) s += p[(long long)b * 2 * D + k];
  sm[q][c] = s;
  __syncthreads();
  if (q == 0) {
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += sm[i][c];
    if (k < D) out.dgamma[blockIdx.y][k] = t;
    else out.dbeta[blockIdx.y][k - D] = t;
  }
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
                                                      const float* __restrict__ act, int M, int K,
                                                      float* __restrict__ dqt, float* __restrict__ dh) {
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
    dh[(long long)row * K + k] = act[(long long)row * K + k] > 0.0f ? a : 0.2f * a;  // LeakyReLU(0.2) gate
  }
}

// dWr[c][k] = sum_rows dqt[row][c] * h[row][k] (c < 4), dWt likewise (c = 4..6), biases = column sums of dqt.
// grid = ceil(K/64) blocks of 1024: wave w sums rows w, w+16, ...; lane = k.  Rows in tiles of 512: the tile's dqt rows
// are staged in LDS (through the scalar cache every row was a dependent ~130 ns round trip: 21 us for 640 rows).
__global__ __launch_bounds__(1024) void head_wgrad_kernel(const float* __restrict__ dqt,
                                                          const float* __restrict__ hfeat, int M, int K,
                                                          float* __restrict__ dwr, float* __restrict__ dbr,
                                                          float* __restrict__ dwt, float* __restrict__ dbt) {
  __shared__ float sm[16][8][64];
  __shared__ float4 dq[512][2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, k = blockIdx.x * 64 + lane;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int U = 16;  // 512 rows per tile -> 32 rows per wave: two batches of loads, each fully in flight
  for (int t0 = 0; t0 < M; t0 += 512) {
    const int trows = M - t0 < 512 ? M - t0 : 512;
    __syncthreads();
    if ((int)threadIdx.x < 2 * trows)
      dq[threadIdx.x >> 1][threadIdx.x & 1] = reinterpret_cast<const float4*>(dqt + 8LL * t0)[threadIdx.x];
    __syncthreads();
    for (int r0 = wave; r0 < trows; r0 += 16 * U) {
      float xs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = r0 + 16 * u;
        xs[u] = (row < trows && k < K) ? hfeat[(long long)(t0 + row) * K + k] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int row = r0 + 16 * u;
        if (row < trows) {  // wave-uniform: broadcast LDS reads
          const float x = xs[u];
          const float4 d0 = dq[row][0], d1 = dq[row][1];
          a[0] = __builtin_fmaf(d0.x, x, a[0]);
          a[1] = __builtin_fmaf(d0.y, x, a[1]);
          a[2] = __builtin_fmaf(d0.z, x, a[2]);
          a[3] = __builtin_fmaf(d0.w, x, a[3]);
          a[4] = __builtin_fmaf(d1.x, x, a[4]);
          a[5] = __builtin_fmaf(d1.y, x, a[5]);
          a[6] = __builtin_fmaf(d1.z, x, a[6]);
          if (lane < 7) a[7] += reinterpret_cast<const float*>(&dq[row][0])[lane];  // bias gradients, lane = output
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) sm[wave][c][lane] = a[c];
  __syncthreads();
  if (wave < 8) {  // wave c reduces channel c over the 16 partials
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += sm[w][wave][lane];
    if (wave < 4) {
      if (k < K) dwr[wave * K + k] = s;
    } else if (wave < 7) {
      if (k < K) dwt[(wave - 4) * K + k] = s;
    } else if (blockIdx.x == 0 && lane < 7) {
      if (lane < 4) dbr[lane] = s;
      else dbt[lane - 4] = s;
    }
  }
}

// ---- host helpers ---------------------------------------------------------------------------------------------------------------
// C = epi(LayerNorm(x) . W^T + bias) for K = 2 x kKP: the LayerNorm rides in the GEMM's operand load (g.ln_* set)
template <int EPI>
void launch_gemm_ln(const GemmArgs& g, hipStream_t s) {
  const dim3 grid((g.M + 31) / 32, g.N / 32), block(kGT);
  hipLaunchKernelGGL((gemm_kernel<EPI, false, 2, true>), grid, block, 0, s, g);
}

// C = epi(m(d x) . W + ...) for K = 2 x kKP with W given as [K, N]: the LayerNorm BACKWARD that produces d x rides in the
// GEMM's operand load (g.ln_* set, see tf_gemm.h)
template <int EPI>
void launch_gemm_lnb(const GemmArgs& g, hipStream_t s) {
  const dim3 grid((g.M + 31) / 32, g.N / 32), block(kGT);
  hipLaunchKernelGGL((gemm_kernel<EPI, true, 2, 2>), grid, block, 0, s, g);
}

// MPA_TF_LNB=0: every LayerNorm backward as its own launch (the A/B switch of the fusion above)
bool lnb_fused() {
  static const bool on = [] {
    const char* e = getenv("MPA_TF_LNB");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}

// MPA_TF_QKVATTN=0: the qkv GEMM and the attention as two launches (the A/B switch of attn_qkv_fwd_kernel)
bool qkv_attn_fused() {
  static const bool on = [] {
    const char* e = getenv("MPA_TF_QKVATTN");
    return e == nullptr || e[0] != '0';
  }();
  return on;
}

// MPA_TF_CHAIN=0: the d(LN1 output) GEMM and LN1's backward as launches of their own (attn_do_bwd_kernel<true>'s switch)
bool chain_fused() {
  static const bool on = [] {
    const char* e = getenv("MPA_TF_CHAIN");
    return e == nullptr || e[0] != '0';
  }();
  return on;
}

// MPA_TF_DOATTN=0: the d o GEMM (with LN2's backward) and the attention backward as two launches (attn_do_bwd_kernel's switch)
bool do_attn_fused() {
  static const bool on = [] {
    const char* e = getenv("MPA_TF_DOATTN");
    return e == nullptr || e[0] != '0';
  }();
  return on;
}

// MPA_TF_SPLITK=0: the K >= 768 GEMMs as one block per tile (the A/B switch of tf_gemm.h's split-K)
bool splitk_on() {
  static const bool on = [] {
    const char* e = getenv("MPA_TF_SPLITK");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
}

// one launch for n <= kGroup weight gradients
void launch_wgrad_group(const WgradArgs* list, int n, hipStream_t s) {
  WgradGroup G{};
  int blocks = 0;
  for (int i = 0; i < kGroup; ++i) {
    G.first[i] = blocks;
    if (i < n) {
      G.p[i] = list[i];
      blocks += (list[i].N / 32) * (list[i].K / 32);
    }
  }
  G.first[kGroup] = blocks;
  hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)blocks), dim3(kWT), 0, s, G);
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
  g.drop = Drop{0, 0.0f, 1.0f, nullptr};
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
  return g;
}

// parameter slots of one encoder layer (order of nn.TransformerEncoderLayer.named_parameters())
enum { P_WQKV = 0, P_BQKV, P_WO, P_BO, P_W1, P_B1, P_W2, P_B2, P_G1, P_BE1, P_G2, P_BE2, P_PER_LAYER };
enum { S_ATTN = 0, S_SA_OUT = 1, S_FFN = 2, S_FFN_OUT = 3, S_PER_LAYER = 4 };  // dropout sites

struct TfDims {
  int64_t B, P, D, H, FF, L, M;
};

struct TfWs {  // per-layer saved tensors + scratch
  float *x_in, *stats1, *h1, *qkv, *probs, *o, *x_mid, *stats2, *h2, *f;
};

struct TfLayout {
  TfWs layer[16];
  float *x_final, *stats_f;
  // backward scratch
  float *g_a, *g_b, *g_c, *g_d, *gd_out, *gd_mid, *dz, *dqkv, *lnpart;
  float *gd_alt, *hp;   // attn_do_bwd_kernel<CHAIN>: the second masked-gradient buffer, the heads' partial tiles [B][H][32][D]
  float* sk_buf;        // split-K half tiles of the K >= 768 GEMMs (tf_gemm.h)
  unsigned* sk_ticket;  // their tickets: cleared by the first GEMM of every forward / backward call, reset after each use
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
    w.layer[l].h1 = take(d.M * d.D);
    w.layer[l].qkv = take(d.M * 3 * d.D);
    w.layer[l].probs = take(d.B * d.H * d.P * d.P);
    w.layer[l].o = take(d.M * d.D);
    w.layer[l].x_mid = take(d.M * d.D);
    w.layer[l].stats2 = take(2 * d.M);
    w.layer[l].h2 = take(d.M * d.D);
    w.layer[l].f = take(d.M * d.FF);
  }
  w.x_final = take(d.M * d.D);
  w.stats_f = take(2 * d.M);
  w.g_a = take(d.M * d.D);
  w.g_b = take(d.M * d.D);
  w.g_c = take(d.M * d.D);
  w.g_d = take(d.M * d.D);  // d(attention output): the fused LayerNorm backward reads g_c while this is written
  w.gd_out = take(d.M * d.D);  // dropout-masked copies of the gradients at the two residual branches' outputs
  w.gd_mid = take(d.M * d.D);
  w.dz = take(d.M * d.FF);
  w.dqkv = take(d.M * 3 * d.D);
  w.lnpart = take((2 * d.L + 1) * ((d.M + 3) / 4) * 2 * d.D);  // one partial table per LayerNorm
  w.gd_alt = take(d.M * d.D);
  w.hp = take(d.B * d.H * 32 * d.D);
  w.sk_buf = take((int64_t)tfg::kSkTiles * 2048);
  w.sk_ticket = reinterpret_cast<unsigned*>(take(tfg::kSkTiles));
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

namespace mpa {
// dW [N][K] = dY [M][N]^T . X [M][K] (row stride ldx) and db [N] = column sums of dY (nullable), exact-fp32 matrix-core
// products in a fixed order: the weight gradient of a layer over few rows (mlp.hip: the node MLPs' 640) in one launch
void launch_small_wgrad(const float* dY, const float* X, int ldx, float* dW, float* db, int M, int N, int K, hipStream_t s) {
  WgradArgs a = wgrad_args(dY, X, dW, db, M, N, K);
  a.ldx = ldx;
  launch_wgrad_group(&a, 1, s);
}
}  // namespace mpa

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
                                       float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* ws, float* out,
                                       void* stream) {
  const TfDims d{B, P, D, H, FF, L, B * P};
  if (int st = tf_check(d, "transformer_forward")) return st;
  if (B == 0) return MPA_OK;
  MPA_REQUIRE(tokens && valid && params && ws && out, "transformer_forward: null pointer");
  MPA_REQUIRE(dropout_p >= 0.0f && dropout_p < 1.0f, "transformer_forward: dropout must be in [0, 1)");
  hipStream_t s = mpa::as_stream(stream);
  const TfLayout w = tf_carve(ws, d);
  const Drop drop{seed, dropout_p, 1.0f / (1.0f - dropout_p), reinterpret_cast<const unsigned long long*>(seed_dev)};
  const int M = (int)d.M, Di = (int)D, FFi = (int)FF;
  const float eps = 1e-5f;
  const dim3 rows((M + 3) / 4);
  const bool sk = splitk_on();
  const bool fuse_ln = Di == 2 * kKP;  // D = 256: a block's 32 rows fit its registers, LayerNorm rides in the next GEMM
  for (int l = 0; l < L; ++l) {  // layer l + 1 finds its input already in its own x_in slot
    const float* const* pp = params + l * P_PER_LAYER;
    const TfWs& t = w.layer[l];
    const unsigned site0 = (unsigned)(l * S_PER_LAYER);
    GemmArgs g;
    const bool qkv_attn = fuse_ln && P <= 32 && Di / (int)H == 32 && qkv_attn_fused();
    if (qkv_attn) {  // LN1, the q / k / v projection and the attention of a (sample, head) in one block
      hipLaunchKernelGGL(attn_qkv_fwd_kernel, dim3((unsigned)(B * H)), dim3(kQT), 0, s, l == 0 ? tokens : t.x_in, pp[P_WQKV],
                         pp[P_BQKV], pp[P_G1], pp[P_BE1], eps, valid, (int)P, (int)H, drop, site0 + S_ATTN, t.qkv, t.stats1, t.h1,
                         l == 0 ? t.x_in : (float*)nullptr, t.probs, t.o, l == 0 ? w.sk_ticket : (unsigned*)nullptr,
                         (int)tfg::kSkTiles);
    } else if (fuse_ln) {  // LN1 inside the qkv GEMM
      g = gemm_args(l == 0 ? tokens : t.x_in, pp[P_WQKV], pp[P_BQKV], t.qkv, M, 3 * Di, Di);
      g.ln_gamma = pp[P_G1];
      g.ln_beta = pp[P_BE1];
      g.ln_eps = eps;
      g.ln_stats = t.stats1;
      g.ln_h = t.h1;
      g.ln_xcopy = l == 0 ? t.x_in : (float*)nullptr;
      if (l == 0) g.zero = w.sk_ticket, g.zero_n = tfg::kSkTiles;  // (the call's first GEMM: split-K tickets)
      launch_gemm_ln<EPI_NONE>(g, s);
    } else {
      hipLaunchKernelGGL(ln_fwd_kernel, rows, dim3(kT), 0, s, l == 0 ? tokens : t.x_in, pp[P_G1], pp[P_BE1], M, Di, eps,
                         t.stats1, t.h1, l == 0 ? t.x_in : (float*)nullptr);
      g = gemm_args(t.h1, pp[P_WQKV], pp[P_BQKV], t.qkv, M, 3 * Di, Di);
      if (l == 0) g.zero = w.sk_ticket, g.zero_n = tfg::kSkTiles;
      launch_gemm<EPI_NONE>(g, s);
    }
    // q k^T and attn . v on the matrix cores whenever the tokens fit one 32-row tile (the shipped configs: P = 20, head
    // dim 32); the scalar kernel covers the rest of the envelope (<= 64 tokens, any head dim <= 64)
    if (qkv_attn) {
    } else if (P <= 32 && Di / (int)H == 32)
      hipLaunchKernelGGL(attn_fwd_mfma_kernel<1>, dim3((unsigned)(B * H)), dim3(64), 0, s, t.qkv, valid, (int)P, Di, (int)H,
                         drop, site0 + S_ATTN, t.probs, t.o);
    else if (P <= 32 && Di / (int)H == 64)
      hipLaunchKernelGGL(attn_fwd_mfma_kernel<2>, dim3((unsigned)(B * H)), dim3(64), 0, s, t.qkv, valid, (int)P, Di, (int)H,
                         drop, site0 + S_ATTN, t.probs, t.o);
    else
      hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)(B * H)), dim3(kAT), 0, s, t.qkv, valid, (int)P, Di, (int)H,
                         drop, site0 + S_ATTN, t.probs, t.o);
    g = gemm_args(t.o, pp[P_WO], pp[P_BO], t.x_mid, M, Di, Di);
    g.resid = t.x_in;
    g.drop = drop;
    g.epi_site = site0 + S_SA_OUT;
    launch_gemm<EPI_DROP_RESID>(g, s);
    if (fuse_ln) {  // LN2 inside the first FFN GEMM
      g = gemm_args(t.x_mid, pp[P_W1], pp[P_B1], t.f, M, FFi, Di);
      g.ln_gamma = pp[P_G2];
      g.ln_beta = pp[P_BE2];
      g.ln_eps = eps;
      g.ln_stats = t.stats2;
      g.ln_h = t.h2;
      g.drop = drop;
      g.epi_site = site0 + S_FFN;
      launch_gemm_ln<EPI_RELU_DROP>(g, s);
    } else {
      hipLaunchKernelGGL(ln_fwd_kernel, rows, dim3(kT), 0, s, t.x_mid, pp[P_G2], pp[P_BE2], M, Di, eps, t.stats2, t.h2,
                         (float*)nullptr);
      g = gemm_args(t.h2, pp[P_W1], pp[P_B1], t.f, M, FFi, Di);
      g.drop = drop;
      g.epi_site = site0 + S_FFN;
      launch_gemm<EPI_RELU_DROP>(g, s);
    }
    float* x_out = l + 1 < L ? w.layer[l + 1].x_in : w.x_final;
    g = gemm_args(t.f, pp[P_W2], pp[P_B2], x_out, M, Di, FFi);
    g.resid = t.x_mid;
    g.drop = drop;
    g.epi_site = site0 + S_FFN_OUT;
    if (sk) g.sk_buf = w.sk_buf, g.sk_ticket = w.sk_ticket;
    launch_gemm_sk<EPI_DROP_RESID>(g, s);
  }
  const float* const* fin = params + L * P_PER_LAYER;
  hipLaunchKernelGGL(ln_fwd_kernel, rows, dim3(kT), 0, s, w.x_final, fin[0], fin[1], M, Di, eps, w.stats_f, out,
                     (float*)nullptr);
  return mpa::check_launch("transformer_forward");
}

extern "C" int mpa_transformer_backward(const float* grad_out, const float* valid, const float* const* params,
                                        int64_t B, int64_t P, int64_t D, int64_t H, int64_t FF, int64_t L,
                                        float dropout_p, uint64_t seed, const uint64_t* seed_dev, float* ws,
                                        float* grad_tokens, float* const* grad_params, void* stream) {
  const TfDims d{B, P, D, H, FF, L, B * P};
  if (int st = tf_check(d, "transformer_backward")) return st;
  if (B == 0) return MPA_OK;
  MPA_REQUIRE(grad_out && valid && params && ws && grad_tokens && grad_params, "transformer_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const TfLayout w = tf_carve(ws, d);
  const Drop drop{seed, dropout_p, 1.0f / (1.0f - dropout_p), reinterpret_cast<const unsigned long long*>(seed_dev)};
  const int M = (int)d.M, Di = (int)D, FFi = (int)FF;
  const dim3 rows((M + 3) / 4);
  const unsigned lnblocks = (unsigned)((M + 3) / 4);
  const size_t ln_smem = sizeof(float) * 4 * 2 * D;
  const long long ln_stride = (long long)lnblocks * 2 * D;  // partial table of one LayerNorm
  LnSites sites{};
  // D = 256: LN2's backward rides in the operand load of the d o GEMM that consumes its d x (one partial row per 32-row
  // tile).  The other LayerNorm backwards stay launches of their own: the GEMM below them has 32 column tiles, and 32
  // blocks regenerating the same row tile's dropout mask cost more than the launch (19.3 vs 9.9 + 5.8 us, LABBOOK 5.3)
  const bool sk = splitk_on();
  const bool fuse = Di == 2 * kKP && lnb_fused();
  // d o product + LN2 backward + attention backward per (sample, head) in one block (attn_do_bwd_kernel)
  const bool do_attn = fuse && P <= 32 && Di / (int)H == 32 && do_attn_fused() && B <= (int64_t)lnblocks;
  auto ln_site = [&](int idx, float* dgamma, float* dbeta, bool fused) {  // 2l: LN1 of layer l, 2l + 1: LN2, 2L: the final one
    sites.dgamma[idx] = dgamma;
    sites.dbeta[idx] = dbeta;
    sites.blocks[idx] = fused ? (do_attn ? (int)B : (M + 31) / 32) : (int)lnblocks;  // partial rows of the site
    return w.lnpart + idx * ln_stride;
  };
  struct LnB {  // one LayerNorm backward: d x = resid + J^T dh, its masked copy (site) and the dgamma / dbeta partials
    const float *dh, *x, *stats, *gamma, *resid;
    float *dx, *part, *dx_drop;
    unsigned site;
  };
  auto ln_alone = [&](const LnB& b) {
    hipLaunchKernelGGL(ln_bwd_kernel, dim3(lnblocks), dim3(kT), ln_smem, s, b.dh, b.x, b.stats, b.gamma, b.resid, M, Di, b.dx,
                       b.part, drop, b.site, b.dx_drop);
  };
  auto ln_ride = [&](GemmArgs& ga, const LnB& b) {  // ga.A becomes dh; the GEMM multiplies the masked d x
    ga.A = b.dh;
    ga.ln_x = b.x;
    ga.ln_stats = const_cast<float*>(b.stats);
    ga.ln_gamma = b.gamma;
    ga.ln_resid = b.resid;
    ga.ln_dx = b.dx;
    ga.ln_dxdrop = b.dx_drop;
    ga.ln_part = b.part;
    ga.ln_site = b.site;
    ga.drop = drop;
  };
  const float* const* fin = params + L * P_PER_LAYER;
  float* const* gfin = grad_params + L * P_PER_LAYER;
  // final LayerNorm backward -> g_a = d x_final
  // every LayerNorm backward below also leaves the dropout-masked copy its consumers need (see ln_bwd_kernel)
  const bool dr = dropout_p > 0.0f;
  ln_alone(LnB{grad_out, w.x_final, w.stats_f, fin[0], nullptr, w.g_a, ln_site((int)(2 * L), gfin[0], gfin[1], false),
                 dr ? w.gd_out : (float*)nullptr, (unsigned)((L - 1) * S_PER_LAYER + S_FFN_OUT)});
  float* g = w.g_a;      // gradient w.r.t. the current layer's output
  float* spare = w.g_b;  // rotating buffers
  float* spare2 = w.g_c;
  // attn_do_bwd_kernel<true>: the d(LN1 output) product and LN1's backward ride behind the attention's gradient; the masked
  // gradient for the layer below then goes to the OTHER of two buffers (this layer's weight gradients still read the first)
  const bool chain = do_attn && chain_fused() && B <= (int64_t)tfg::kSkTiles && H <= 8;
  float* gd_cur = w.gd_out;
  float* gd_nxt = w.gd_alt;
  for (int l = (int)L - 1; l >= 0; --l) {
    const float* const* pp = params + l * P_PER_LAYER;
    float* const* gp = grad_params + l * P_PER_LAYER;
    const TfWs& t = w.layer[l];
    const unsigned site0 = (unsigned)(l * S_PER_LAYER);
    // ---- FFN: x_out = x_mid + drop(f . W2^T + b2),  f = drop(relu(LN2(x_mid) . W1^T + b1))
    // (the layer's four weight gradients are off the critical path and all their operands stay intact until the
    // layer's last LayerNorm backward: they go out as ONE launch just before it)
    WgradArgs wl[4];
    const float* gdo = dr ? gd_cur : g;  // drop-masked g (site S_FFN_OUT of this layer)
    wl[0] = wgrad_args(gdo, t.f, gp[P_W2], gp[P_B2], M, Di, FFi);
    GemmArgs ga = gemm_args(gdo, pp[P_W2], nullptr, w.dz, M, FFi, Di);  // W2 is [D, FF] = [K, N]
    ga.resid = t.f;
    ga.drop = drop;  // the epilogue's keep-scale of the hidden layer's dropout (f > 0 <=> kept and active)
    if (l == (int)L - 1) ga.zero = w.sk_ticket, ga.zero_n = tfg::kSkTiles;  // (the call's first GEMM: split-K tickets)
    launch_gemm<EPI_RELU_MASK, true>(ga, s);  // dz = d(pre-activation)
    wl[1] = wgrad_args(w.dz, t.h2, gp[P_W1], gp[P_B1], M, FFi, Di);
    ga = gemm_args(w.dz, pp[P_W1], nullptr, spare2, M, Di, FFi);
    if (sk) ga.sk_buf = w.sk_buf, ga.sk_ticket = w.sk_ticket;
    launch_gemm_sk<EPI_NONE, true>(ga, s);  // d LN2 output
    const LnB ln2{spare2, t.x_mid, t.stats2, pp[P_G2], g, spare, ln_site(2 * l + 1, gp[P_G2], gp[P_BE2], fuse),
                  dr ? w.gd_mid : (float*)nullptr, site0 + S_SA_OUT};  // spare = d x_mid
    float* g_mid = spare;
    spare = g;
    // ---- attention block: x_mid = x_in + drop(o . Wo^T + bo)
    const float* gdm = dr ? w.gd_mid : g_mid;  // drop-masked g_mid (site S_SA_OUT)
    wl[2] = wgrad_args(gdm, t.o, gp[P_WO], gp[P_BO], M, Di, Di);
    float* d_o = spare2;
    if (chain) {
      // the layer's input gradient lands in the buffer of d(LN2 output): a sample's rows of it are read only by that sample's
      // blocks, all of them before its last ticket
      ChainArgs ch{};
      ch.wqkv = pp[P_WQKV];
      ch.x_in = t.x_in;
      ch.stats1 = t.stats1;
      ch.gamma1 = pp[P_G1];
      ch.g_in = l == 0 ? grad_tokens : spare2;
      ch.gd_next = dr && l > 0 ? gd_nxt : (float*)nullptr;
      ch.site_next = (unsigned)((l - 1) * S_PER_LAYER + S_FFN_OUT);
      ch.ln1_part = ln_site(2 * l, gp[P_G1], gp[P_BE1], true);
      ch.hp = w.hp;
      ch.ticket = w.sk_ticket;
      hipLaunchKernelGGL(attn_do_bwd_kernel<true>, dim3((unsigned)(B * H)), dim3(kBT), 0, s, ln2.dh, ln2.x, ln2.stats, ln2.gamma,
                         ln2.resid, ln2.dx, ln2.dx_drop, ln2.part, ln2.site, pp[P_WO], t.qkv, t.probs, (int)P, (int)H, drop,
                         site0 + S_ATTN, w.dqkv, ch);
      wl[3] = wgrad_args(w.dqkv, t.h1, gp[P_WQKV], gp[P_BQKV], M, 3 * Di, Di);
      launch_wgrad_group(wl, 4, s);
      if (l > 0) {  // next layer down: g = the buffer just written; the old g (now `spare`) and g_mid's buffer are free
        float* g_new = spare2;
        spare2 = g_mid;
        g = g_new;
        float* tmp = gd_cur;
        gd_cur = gd_nxt;
        gd_nxt = tmp;
      }
      continue;
    }
    if (do_attn) {
      hipLaunchKernelGGL(attn_do_bwd_kernel<false>, dim3((unsigned)(B * H)), dim3(kBT), 0, s, ln2.dh, ln2.x, ln2.stats, ln2.gamma,
                         ln2.resid, ln2.dx, ln2.dx_drop, ln2.part, ln2.site, pp[P_WO], t.qkv, t.probs, (int)P, (int)H, drop,
                         site0 + S_ATTN, w.dqkv, ChainArgs{});
    } else if (fuse) {  // LN2's backward inside the d o GEMM: its operand spare2 is still being read, so d o gets its own buffer
      d_o = w.g_d;
      ga = gemm_args(gdm, pp[P_WO], nullptr, d_o, M, Di, Di);
      ln_ride(ga, ln2);
      launch_gemm_lnb<EPI_NONE>(ga, s);
    } else {
      ln_alone(ln2);
      launch_gemm<EPI_NONE, true>(gemm_args(gdm, pp[P_WO], nullptr, d_o, M, Di, Di), s);
    }
    if (do_attn) {
    } else if (P <= 32 && Di / (int)H == 32)
      hipLaunchKernelGGL(attn_bwd_mfma_kernel<1>, dim3((unsigned)(B * H)), dim3(64), 0, s, t.qkv, t.probs, d_o, (int)P, Di,
                         (int)H, drop, site0 + S_ATTN, w.dqkv);
    else if (P <= 32 && Di / (int)H == 64)
      hipLaunchKernelGGL(attn_bwd_mfma_kernel<2>, dim3((unsigned)(B * H)), dim3(64), 0, s, t.qkv, t.probs, d_o, (int)P, Di,
                         (int)H, drop, site0 + S_ATTN, w.dqkv);
    else
      hipLaunchKernelGGL(attn_bwd_kernel, dim3((unsigned)(B * H)), dim3(kAT), 0, s, t.qkv, t.probs, d_o, (int)P, Di,
                         (int)H, drop, site0 + S_ATTN, w.dqkv);
    wl[3] = wgrad_args(w.dqkv, t.h1, gp[P_WQKV], gp[P_BQKV], M, 3 * Di, Di);
    ga = gemm_args(w.dqkv, pp[P_WQKV], nullptr, spare2, M, Di, 3 * Di);
    if (sk) ga.sk_buf = w.sk_buf, ga.sk_ticket = w.sk_ticket;
    launch_gemm_sk<EPI_NONE, true>(ga, s);  // d LN1 out
    launch_wgrad_group(wl, 4, s);  // before LN1's backward overwrites g's buffer and gd_out
    float* g_in = l == 0 ? grad_tokens : spare;
    ln_alone(LnB{spare2, t.x_in, t.stats1, pp[P_G1], g_mid, g_in, ln_site(2 * l, gp[P_G1], gp[P_BE1], false),
                 dr && l > 0 ? w.gd_out : (float*)nullptr, (unsigned)((l - 1) * S_PER_LAYER + S_FFN_OUT)});
    if (l > 0) {  // next layer down: its output gradient is g_in; g_mid's buffer is free again
      g = spare;
      spare = g_mid;
    }
  }
  hipLaunchKernelGGL(ln_reduce_kernel, dim3((unsigned)(2 * D / 64), (unsigned)(2 * L + 1)), dim3(1024), 0, s,
                     (const float*)w.lnpart, ln_stride, Di, sites);
  return mpa::check_launch("transformer_backward");
}

// ---- pose head -------------------------------------------------------------------------------------------------------------------
// params: fc1.w [256,F], fc1.b, fc2.w [128,256], fc2.b, rot.w [4,128], rot.b, trans.w [3,128], trans.b
// ws: h1 [M,256] | h2 [M,128] | rot_raw [M,4] | dqt [M,8] | d2 [M,128] | d1 [M,256]
//     and, when F is not a multiple of the GEMM panels' 64 columns (labels / noise appended to the features), zero-padded
//     copies of x and fc1.w and the padded gradients of both: x' [M,F'] | w' [256,F'] | dx' [M,F'] | dw' [256,F']
struct HeadPad {
  int64_t Fp;  // F rounded up to a multiple of 64
  float *x, *w, *dx, *dw;
};

static HeadPad head_pad(float* ws, int64_t M, int64_t F) {
  HeadPad h;
  h.Fp = (F + 63) / 64 * 64;
  h.x = ws + M * (256 + 128 + 4 + 8 + 128 + 256) + 64;
  h.w = h.x + M * h.Fp;
  h.dx = h.w + 256 * h.Fp;
  h.dw = h.dx + M * h.Fp;
  return h;
}

// two row-major tables in one launch: dst [rows][ldd] = src [rows][lds] in the first min(lds, ldd) columns, zero beyond
// (ldd > lds pads, ldd < lds drops the padding columns).  blocks [0, nb0) work on table 0, the rest on table 1.
struct Pad2 {
  const float* src[2];
  float* dst[2];
  int rows[2];
};
__global__ __launch_bounds__(256) void head_pad2_kernel(Pad2 a, int lds, int ldd, int nb0) {
  const int t = (int)blockIdx.x < nb0 ? 0 : 1;
  const long long i = ((long long)blockIdx.x - (t ? nb0 : 0)) * 256 + threadIdx.x;
  if (i >= (long long)a.rows[t] * ldd) return;
  const long long r = i / ldd;
  const int c = (int)(i - r * ldd);
  a.dst[t][i] = c < lds ? a.src[t][r * lds + c] : 0.0f;
}

static void launch_pad2(const float* s0, float* d0, int64_t rows0, const float* s1, float* d1, int64_t rows1, int64_t lds,
                        int64_t ldd, hipStream_t s) {
  Pad2 a;
  a.src[0] = s0, a.dst[0] = d0, a.rows[0] = (int)rows0;
  a.src[1] = s1, a.dst[1] = d1, a.rows[1] = (int)rows1;
  const int nb0 = (int)((rows0 * ldd + 255) / 256), nb1 = (int)((rows1 * ldd + 255) / 256);
  hipLaunchKernelGGL(head_pad2_kernel, dim3((unsigned)(nb0 + nb1)), dim3(256), 0, s, a, (int)lds, (int)ldd, nb0);
}

extern "C" int mpa_pose_head_workspace(int64_t M, int64_t F, int64_t* float_elems) {
  MPA_REQUIRE(M >= 0 && F >= 1 && F <= 4096 && float_elems, "pose_head_workspace: need 1 <= F <= 4096");
  *float_elems = M * (256 + 128 + 4 + 8 + 128 + 256) + 64;
  if (F % 64 != 0) *float_elems += 2 * (M + 256) * ((F + 63) / 64 * 64);
  return MPA_OK;
}

extern "C" int mpa_pose_head_forward(const float* x, const float* const* params, int64_t M, int64_t F, float* ws,
                                     float* rot, float* trans, void* stream) {
  MPA_REQUIRE(M >= 0 && F >= 1 && F <= 4096, "pose_head_forward: need 1 <= F <= 4096");
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(x && params && ws && rot && trans, "pose_head_forward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  float* h1 = ws;
  float* h2 = h1 + M * 256;
  float* rot_raw = h2 + M * 128;
  const float* w1 = params[0];
  if (F % 64 != 0) {
    const HeadPad hp = head_pad(ws, M, F);
    launch_pad2(x, hp.x, M, w1, hp.w, 256, F, hp.Fp, s);
    x = hp.x, w1 = hp.w, F = hp.Fp;
  }
  launch_gemm<EPI_LEAKY>(gemm_args(x, w1, params[1], h1, (int)M, 256, (int)F), s);
  launch_gemm<EPI_LEAKY>(gemm_args(h1, params[2], params[3], h2, (int)M, 128, 256), s);
  hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(kT), 0, s, h2, params[4], params[5],
                     params[6], params[7], (int)M, 128, rot_raw, rot, trans);
  return mpa::check_launch("pose_head_forward");
}

extern "C" int mpa_pose_head_backward(const float* grad_rot, const float* grad_trans, const float* x,
                                      const float* const* params, int64_t M, int64_t F, float* ws, float* grad_x,
                                      float* const* grad_params, void* stream) {
  MPA_REQUIRE(M >= 0 && F >= 1 && F <= 4096, "pose_head_backward: need 1 <= F <= 4096");
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(grad_rot && grad_trans && x && params && ws && grad_x && grad_params, "pose_head_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  float* h1 = ws;
  float* h2 = h1 + M * 256;
  float* rot_raw = h2 + M * 128;
  float* dqt = rot_raw + M * 4;
  float* d2 = dqt + M * 8;
  float* d1 = d2 + M * 128;
  const int Mi = (int)M;
  const bool padded = F % 64 != 0;
  const HeadPad hp = head_pad(ws, M, F);  // (the forward call left x' and w' there; fc1.w has not changed since)
  const float* w1 = padded ? hp.w : params[0];
  const float* xin = padded ? hp.x : x;
  float* gx = padded ? hp.dx : grad_x;
  float* gw1 = padded ? hp.dw : grad_params[0];
  const int Fi = (int)(padded ? hp.Fp : F);
  hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(kT), 0, s, rot_raw, grad_rot, grad_trans,
                     params[4], params[6], h2, Mi, 128, dqt, d2);  // d2 = gradient at fc2's pre-activation
  hipLaunchKernelGGL(head_wgrad_kernel, dim3(2), dim3(1024), 0, s, dqt, h2, Mi, 128, grad_params[4], grad_params[5],
                     grad_params[6], grad_params[7]);
  GemmArgs ga = gemm_args(d2, params[2], nullptr, d1, Mi, 256, 128);  // fc2.weight is [128, 256] = [K, N]
  ga.resid = h1;
  launch_gemm<EPI_LEAKY_MASK, true>(ga, s);
  launch_gemm<EPI_NONE, true>(gemm_args(d1, w1, nullptr, gx, Mi, Fi, 256), s);
  const WgradArgs wl[2] = {wgrad_args(d2, h1, grad_params[2], grad_params[3], Mi, 128, 256),
                           wgrad_args(d1, xin, gw1, grad_params[1], Mi, 256, Fi)};
  launch_wgrad_group(wl, 2, s);  // both weight gradients in one launch, off the path to grad_x
  if (padded) launch_pad2(gx, grad_x, M, gw1, grad_params[0], 256, hp.Fp, F, s);  // drop the padding columns
  return mpa::check_launch("pose_head_backward");
}
