// The small per-iteration pieces of the graph networks that sit between the MLP layers (csrc/mlp.hip), for gfx950:
//
//   narrow linear + ReLU      PoseEncoder.mlp1 (7 -> 256)            multi_part_assembly/models/dgl/modules.py:76-86
//   relation head             RelationNet.mlp3 (512 -> 1) + sigmoid, times the valid-pair mask
//                                                                    dgl/modules.py:61-73, dgl/network.py:121-133
//   relation-weighted mean    sum_j edge_ij * rel_ij / (sum_j rel_ij + 1e-6)      dgl/network.py:135-152
//   pair rows                 [a_i ; b_j] for every part pair (i, j) of a sample  dgl/network.py:121-125, 135-141
//
// Each is a few hundred kilobytes to tens of megabytes of traffic per GNN iteration; as library ops they were ~35
// launches per iteration, forward and backward (element-wise products, reductions, 1-column GEMMs, their transposes).
// Here every piece is one launch forward and one or two backward.  All reductions run in a fixed order (no atomics): the
// results are bit-reproducible.
#include "common.h"

namespace {

constexpr int kMaxK = 16;  // input width of the narrow linear layer

template <typename Kern, typename... Args>
void launch(Kern kern, dim3 grid, dim3 block, hipStream_t s, Args... args) {
  hipLaunchKernelGGL(kern, grid, block, 0, s, args...);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- narrow linear + ReLU ---------------------------------------------------------------------------------------------
// out[r][n] = relu(b[n] + sum_k x[r][k] w[n][k]).  grid = R, block = 256 (channels n, n + 256, ...).
__global__ __launch_bounds__(256) void nl_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, int K, int N, float* __restrict__ out) {
  const long long r = blockIdx.x;
  float xr[kMaxK];
#pragma unroll
  for (int k = 0; k < kMaxK; ++k) xr[k] = k < K ? x[r * K + k] : 0.0f;  // (uniform address: scalar loads)
  for (int n = threadIdx.x; n < N; n += 256) {
    float acc = b != nullptr ? b[n] : 0.0f;
#pragma unroll
    for (int k = 0; k < kMaxK; ++k)
      if (k < K) acc = __builtin_fmaf(xr[k], w[(long long)n * K + k], acc);
    out[r * N + n] = acc > 0.0f ? acc : 0.0f;
  }
}

// Backward, two kinds of block in one launch (dy = grad_out where out > 0):
//   blocks [0, R): row r — grad_x[r][k] = sum_n dy[r][n] w[n][k]  (skipped when grad_x is null)
//   then ceil(N / 64) x ceil(R / 128) blocks: 64 channels x 128 rows — the chunk's part of grad_w[n][k] = sum_r dy[r][n] x[r][k]
//   and grad_b[n] = sum_r dy[r][n] into part[chunk][n][kMaxK + 1]; 8 row groups of 64 lanes, 16 rows each with all their
//   loads in flight at once (a block per 64 channels walking ALL rows was a chain of dependent round trips: 63 us at 640
//   rows), the groups' sums added in group order.  nl_reduce_kernel adds the chunks in order.
constexpr int kNlRows = 128;
__global__ __launch_bounds__(512) void nl_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                     const float* __restrict__ x, const float* __restrict__ w, int R,
                                                     int K, int N, float* __restrict__ grad_x, float* __restrict__ part) {
  __shared__ float red[8][64][kMaxK + 1];
  __shared__ float xs[kNlRows][kMaxK];
  const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
  if ((int)blockIdx.x < R) {
    if (grad_x == nullptr) return;
    const long long r = blockIdx.x;
    float acc[kMaxK];
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) acc[k] = 0.0f;
    for (int n = tid; n < N; n += 512) {
      const float dy = out[r * N + n] > 0.0f ? g[r * N + n] : 0.0f;
#pragma unroll
      for (int k = 0; k < kMaxK; ++k)
        if (k < K) acc[k] = __builtin_fmaf(dy, w[(long long)n * K + k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < kMaxK; ++k) {
      if (k < K) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[grp][0][k] = s;
      }
    }
    __syncthreads();
    if (tid < K) {
      float s = 0.0f;
      for (int q = 0; q < 8; ++q) s += red[q][0][tid];
      grad_x[r * K + tid] = s;
    }
    return;
  }
  const int nb = (N + 63) / 64, bb = (int)blockIdx.x - R, chunk = bb / nb;
  const int n = (bb - chunk * nb) * 64 + lane;
  const int t0 = chunk * kNlRows, trows = R - t0 < kNlRows ? R - t0 : kNlRows;
  for (int e = tid; e < trows * K; e += 512) xs[e / K][e % K] = x[(long long)t0 * K + e];
  __syncthreads();
  float acc[kMaxK + 1];
#pragma unroll
  for (int k = 0; k <= kMaxK; ++k) acc[k] = 0.0f;
  if (n < N) {
    constexpr int U = kNlRows / 8;
    float dy[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = grp + 8 * u;
      const long long r = t0 + (rr < trows ? rr : 0);  // (clamped: the value is dropped below)
      dy[u] = out[r * N + n] > 0.0f ? g[r * N + n] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int rr = grp + 8 * u;
      if (rr < trows) {
#pragma unroll
        for (int k = 0; k < kMaxK; ++k)
          if (k < K) acc[k] = __builtin_fmaf(dy[u], xs[rr][k], acc[k]);
        acc[kMaxK] += dy[u];
      }
    }
  }
#pragma unroll
  for (int k = 0; k <= kMaxK; ++k) red[grp][lane][k] = acc[k];
  __syncthreads();
  if (n < N) {
    for (int k = grp; k <= kMaxK; k += 8) {
      float s = 0.0f;
      for (int q = 0; q < 8; ++q) s += red[q][lane][k];
      part[((long long)chunk * N + n) * (kMaxK + 1) + k] = s;
    }
  }
}

// grad_w[n][k] (k < K) and grad_b[n] (column kMaxK of the table) = the row chunks' partials added in chunk order.
__global__ __launch_bounds__(256) void nl_reduce_kernel(const float* __restrict__ part, int chunks, int K, int N,
                                                        float* __restrict__ grad_w, float* __restrict__ grad_b) {
  const int e = (int)blockIdx.x * 256 + threadIdx.x;
  if (e >= N * (K + 1)) return;
  const int n = e / (K + 1), k = e - n * (K + 1);
  const int col = k < K ? k : kMaxK;
  float s = 0.0f;
  for (int c = 0; c < chunks; ++c) s += part[((long long)c * N + n) * (kMaxK + 1) + col];
  if (k < K) grad_w[(long long)n * K + k] = s;
  else if (grad_b != nullptr) grad_b[n] = s;
}

// ---- relation head ------------------------------------------------------------------------------------------------
// sig[r] = sigmoid(b + h[r] . w), out[r] = sig[r] * mask[r].  One wave per row, 4 rows per block.
__global__ __launch_bounds__(256) void rh_fwd_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                     const float* __restrict__ b, const float* __restrict__ mask, int R,
                                                     int K, float* __restrict__ sig, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float acc = 0.0f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 a = *reinterpret_cast<const float4*>(h + r * K + k);
    const float4 c = *reinterpret_cast<const float4*>(w + k);
    acc = __builtin_fmaf(a.x, c.x, acc);
    acc = __builtin_fmaf(a.y, c.y, acc);
    acc = __builtin_fmaf(a.z, c.z, acc);
    acc = __builtin_fmaf(a.w, c.w, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float z = acc + (b != nullptr ? b[0] : 0.0f);
    const float s = 1.0f / (1.0f + expf(-z));
    sig[r] = s;
    out[r] = mask != nullptr ? s * mask[r] : s;
  }
}

constexpr int kRhRows = 16;  // rows per block of the backward (12 800 pair rows: 800 blocks)

// dz[r] = g[r] mask[r] sig[r] (1 - sig[r]);  grad_h[r][k] = dz[r] w[k];  the block's partial of grad_w[k] = sum_r dz[r] h[r][k]
// and of grad_b = sum_r dz[r] into part[block][K + 1].  grid = ceil(R / 16), block = 256.
__global__ __launch_bounds__(256) void rh_bwd_kernel(const float* __restrict__ g, const float* __restrict__ h,
                                                     const float* __restrict__ w, const float* __restrict__ mask,
                                                     const float* __restrict__ sig, int R, int K,
                                                     float* __restrict__ grad_h, float* __restrict__ part) {
  __shared__ float dz[kRhRows];
  const long long r0 = (long long)blockIdx.x * kRhRows;
  const int rows = R - r0 < kRhRows ? (int)(R - r0) : kRhRows;
  if ((int)threadIdx.x < kRhRows) {
    float v = 0.0f;
    if ((int)threadIdx.x < rows) {
      const long long r = r0 + threadIdx.x;
      const float s = sig[r];
      v = g[r] * (mask != nullptr ? mask[r] : 1.0f) * (s * (1.0f - s));
    }
    dz[threadIdx.x] = v;
  }
  __syncthreads();
  float* prow = part + (long long)blockIdx.x * (K + 1);
  for (int k = threadIdx.x; k < K; k += 256) {
    const float wk = w[k];
    float hv[kRhRows];
#pragma unroll
    for (int i = 0; i < kRhRows; ++i) hv[i] = h[(r0 + (i < rows ? i : rows - 1)) * K + k];  // all loads in flight
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < kRhRows; ++i) {
      if (i < rows) {
        const float d = dz[i];
        acc = __builtin_fmaf(d, hv[i], acc);
        if (grad_h != nullptr) grad_h[(r0 + i) * K + k] = d * wk;
      }
    }
    prow[k] = acc;
  }
  if (threadIdx.x == 0) {
    float s = 0.0f;
    for (int i = 0; i < rows; ++i) s += dz[i];
    prow[K] = s;
  }
}

// grad_w[k] (k < K) and grad_b (k == K) = the blocks' partials added in block order.  grid = ceil((K + 1) / 4), block 256:
// one wave per output, lanes over the blocks.
__global__ __launch_bounds__(256) void rh_reduce_kernel(const float* __restrict__ part, int nblk, int K,
                                                        float* __restrict__ grad_w, float* __restrict__ grad_b) {
  const int lane = threadIdx.x & 63, k = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k > K) return;
  float acc = 0.0f;
  for (int q = lane; q < nblk; q += 64) acc += part[(long long)q * (K + 1) + k];
  acc = wave_sum(acc);
  if (lane == 0) {
    if (k < K) grad_w[k] = acc;
    else if (grad_b != nullptr) grad_b[0] = acc;
  }
}

// ---- relation-weighted mean ---------------------------------------------------------------------------------------
// out[g][c] = sum_j edge[g][j][c] rel[g][j] / (sum_j rel[g][j] + 1e-6).  grid = G (rows (sample, part i)), block = 128.
__global__ __launch_bounds__(128) void rm_fwd_kernel(const float* __restrict__ edge, const float* __restrict__ rel, int P,
                                                     int C, float* __restrict__ out) {
  __shared__ float rl[64];  // the row's weights (as scalar-cache operands every one is a dependent round trip)
  const long long g = blockIdx.x;
  if ((int)threadIdx.x < P) rl[threadIdx.x] = rel[g * P + threadIdx.x];
  __syncthreads();
  float den = 0.0f;
  for (int j = 0; j < P; ++j) den += rl[j];
  den += 1e-6f;
  for (int c = threadIdx.x; c < C; c += 128) {
    float acc = 0.0f;
    for (int j = 0; j < P; ++j) acc = __builtin_fmaf(edge[(g * P + j) * C + c], rl[j], acc);
    out[g * C + c] = acc / den;
  }
}

// grad_edge[g][j][c] = go[g][c] rel[g][j] / den;  grad_rel[g][j] = sum_c go[g][c] (edge[g][j][c] - out[g][c]) / den.
__global__ __launch_bounds__(128) void rm_bwd_kernel(const float* __restrict__ go, const float* __restrict__ edge,
                                                     const float* __restrict__ rel, const float* __restrict__ out, int P,
                                                     int C, float* __restrict__ grad_edge, float* __restrict__ grad_rel) {
  __shared__ float red[2][64];
  __shared__ float rl[64];
  const long long g = blockIdx.x;
  if ((int)threadIdx.x < P) rl[threadIdx.x] = rel[g * P + threadIdx.x];
  __syncthreads();
  float den = 0.0f;
  for (int j = 0; j < P; ++j) den += rl[j];
  den += 1e-6f;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if ((int)threadIdx.x < 64) red[0][threadIdx.x] = red[1][threadIdx.x] = 0.0f;
  __syncthreads();
  for (int c0 = 0; c0 < C; c0 += 128) {  // channel chunks of the block's width (one for C <= 128)
    const int c = c0 + threadIdx.x;
    const bool on = c < C;
    const float gd = on ? go[g * C + c] / den : 0.0f, oc = on ? out[g * C + c] : 0.0f;
    for (int j0 = 0; j0 < P; j0 += 4) {  // four rows' loads in flight
      float ev[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        ev[u] = (on && grad_rel != nullptr && j0 + u < P) ? edge[(g * P + j0 + u) * C + c] : oc;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u;
        if (j < P) {  // (uniform)
          if (on && grad_edge != nullptr) grad_edge[(g * P + j) * C + c] = gd * rl[j];
          if (grad_rel != nullptr) {
            const float acc = wave_sum(gd * (ev[u] - oc));
            if (lane == 0) red[wv][j] += acc;  // (chunks in order: fixed summation order)
          }
        }
      }
    }
  }
  if (grad_rel != nullptr) {
    __syncthreads();
    if ((int)threadIdx.x < P) grad_rel[g * P + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x];
  }
}

// ---- pair rows --------------------------------------------------------------------------------------------------------
// out[s][i][j] = [a[s][i] ; b[s][j]]  ([S, P, P, 2F]; swap: [b[s][j] ; a[s][i]]).  grid = S * P (sample, i), block = 256;
// float4 columns.
__global__ __launch_bounds__(256) void pr_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int P, int F,
                                                     int swap, float* __restrict__ out) {
  const long long si = blockIdx.x, s0 = si / P;
  const int F4 = F / 4;
  const float4* ai = reinterpret_cast<const float4*>(a + si * F);
  float4* o = reinterpret_cast<float4*>(out + si * P * 2 * F);
  for (int t = threadIdx.x; t < P * 2 * F4; t += 256) {
    const int j = t / (2 * F4), c = t - j * 2 * F4;
    const bool from_a = (c < F4) != (swap != 0);
    const int cc = c < F4 ? c : c - F4;
    o[t] = from_a ? ai[cc] : reinterpret_cast<const float4*>(b + (s0 * P + j) * F)[cc];
  }
}

// grad_a[s][i] = sum_j g[s][i][j][:F],  grad_b[s][j] = sum_i g[s][i][j][F:]  (swap: the other halves; sums in index order).
// grid = 2 * S * P: the first half of the blocks own a row (s, i) of grad_a, the second half a row (s, j) of grad_b;
// block = 128.
__global__ __launch_bounds__(128) void pr_bwd_kernel(const float* __restrict__ g, int SP, int P, int F, int swap,
                                                     float* __restrict__ grad_a, float* __restrict__ grad_b) {
  const bool second = (int)blockIdx.x >= SP;
  const long long row = second ? (long long)blockIdx.x - SP : (long long)blockIdx.x;
  const long long s0 = row / P, q = row - s0 * P;
  float* dst = second ? grad_b : grad_a;
  if (dst == nullptr) return;
  // first half: q = i, walk j (stride 2F); second half: q = j, walk i (stride P * 2F); a's columns come first unless swapped
  const int col = (second != (swap != 0)) ? F : 0;
  const float* base = (second ? g + ((s0 * P) * P + q) * 2 * F : g + (s0 * P + q) * P * 2 * F) + col;
  const long long step = second ? (long long)P * 2 * F : 2LL * F;
  for (int c = threadIdx.x; c < F; c += 128) {
    float acc = 0.0f;
    for (int t = 0; t < P; ++t) acc += base[t * step + c];
    dst[row * F + c] = acc;
  }
}

}  // namespace

extern "C" int mpa_narrow_linear_relu_forward(const float* x, const float* w, const float* bias, int64_t R, int64_t K,
                                              int64_t N, float* out, void* stream) {
  MPA_REQUIRE(R >= 1 && R <= (1 << 24) && K >= 1 && K <= kMaxK && N >= 1 && N <= (1 << 16),
              "narrow_linear_relu_forward: R=%lld K=%lld (1..%d) N=%lld out of range", (long long)R, (long long)K, kMaxK,
              (long long)N);
  MPA_REQUIRE(x && w && out, "narrow_linear_relu_forward: null pointer");
  launch(nl_fwd_kernel, dim3((unsigned)R), dim3(256), mpa::as_stream(stream), x, w, bias, (int)K, (int)N, out);
  return mpa::check_launch("narrow_linear_relu_forward");
}

extern "C" int mpa_narrow_linear_relu_workspace(int64_t R, int64_t K, int64_t N, int64_t* float_elems) {
  MPA_REQUIRE(float_elems != nullptr, "narrow_linear_relu_workspace: null pointer");
  MPA_REQUIRE(R >= 1 && R <= (1 << 24) && K >= 1 && K <= kMaxK && N >= 1 && N <= (1 << 16),
              "narrow_linear_relu: R=%lld K=%lld (1..%d) N=%lld out of range", (long long)R, (long long)K, kMaxK, (long long)N);
  *float_elems = ((R + kNlRows - 1) / kNlRows) * N * (kMaxK + 1);  // the backward's per-chunk partial sums
  return MPA_OK;
}

extern "C" int mpa_narrow_linear_relu_backward(const float* grad_out, const float* out, const float* x, const float* w,
                                               int64_t R, int64_t K, int64_t N, float* ws, float* grad_x, float* grad_w,
                                               float* grad_b, void* stream) {
  int64_t nf;
  if (int st = mpa_narrow_linear_relu_workspace(R, K, N, &nf)) return st;
  MPA_REQUIRE(grad_out && out && x && w && ws && grad_w, "narrow_linear_relu_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const int chunks = (int)((R + kNlRows - 1) / kNlRows);
  launch(nl_bwd_kernel, dim3((unsigned)(R + ((N + 63) / 64) * chunks)), dim3(512), s, grad_out, out, x, w, (int)R, (int)K,
         (int)N, grad_x, ws);
  launch(nl_reduce_kernel, dim3((unsigned)((N * (K + 1) + 255) / 256)), dim3(256), s, (const float*)ws, chunks, (int)K, (int)N,
         grad_w, grad_b);
  return mpa::check_launch("narrow_linear_relu_backward");
}

extern "C" int mpa_relation_head_workspace(int64_t R, int64_t K, int64_t* float_elems) {
  MPA_REQUIRE(float_elems != nullptr, "relation_head_workspace: null pointer");
  MPA_REQUIRE(R >= 1 && R <= (1 << 24) && K >= 4 && K % 4 == 0 && K <= 4096,
              "relation_head: R=%lld K=%lld (a multiple of 4, <= 4096) out of range", (long long)R, (long long)K);
  *float_elems = R + ((R + kRhRows - 1) / kRhRows) * (K + 1);  // the sigmoids, then the backward's per-block partials
  return MPA_OK;
}

extern "C" int mpa_relation_head_forward(const float* h, const float* w, const float* bias, const float* mask, int64_t R,
                                         int64_t K, float* ws, float* out, void* stream) {
  int64_t n;
  if (int st = mpa_relation_head_workspace(R, K, &n)) return st;
  MPA_REQUIRE(h && w && ws && out, "relation_head_forward: null pointer");
  launch(rh_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), mpa::as_stream(stream), h, w, bias, mask, (int)R, (int)K, ws,
         out);
  return mpa::check_launch("relation_head_forward");
}

extern "C" int mpa_relation_head_backward(const float* grad_out, const float* h, const float* w, const float* mask,
                                          int64_t R, int64_t K, float* ws, float* grad_h, float* grad_w, float* grad_b,
                                          void* stream) {
  int64_t n;
  if (int st = mpa_relation_head_workspace(R, K, &n)) return st;
  MPA_REQUIRE(grad_out && h && w && ws && grad_w, "relation_head_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const int nblk = (int)((R + kRhRows - 1) / kRhRows);
  float* part = ws + R;
  launch(rh_bwd_kernel, dim3((unsigned)nblk), dim3(256), s, grad_out, h, w, mask, (const float*)ws, (int)R, (int)K, grad_h,
         part);
  launch(rh_reduce_kernel, dim3((unsigned)((K + 1 + 3) / 4)), dim3(256), s, (const float*)part, nblk, (int)K, grad_w, grad_b);
  return mpa::check_launch("relation_head_backward");
}

static int rm_check(int64_t G, int64_t P, int64_t C, const char* who) {
  MPA_REQUIRE(G >= 1 && G <= (1 << 24) && P >= 1 && P <= 64 && C >= 1 && C <= (1 << 16),
              "%s: G=%lld P=%lld (<= 64) C=%lld out of range", who, (long long)G, (long long)P, (long long)C);
  return MPA_OK;
}

extern "C" int mpa_relation_mean_forward(const float* edge, const float* rel, int64_t G, int64_t P, int64_t C, float* out,
                                         void* stream) {
  if (int st = rm_check(G, P, C, "relation_mean_forward")) return st;
  MPA_REQUIRE(edge && rel && out, "relation_mean_forward: null pointer");
  launch(rm_fwd_kernel, dim3((unsigned)G), dim3(128), mpa::as_stream(stream), edge, rel, (int)P, (int)C, out);
  return mpa::check_launch("relation_mean_forward");
}

extern "C" int mpa_relation_mean_backward(const float* grad_out, const float* edge, const float* rel, const float* out,
                                          int64_t G, int64_t P, int64_t C, float* grad_edge, float* grad_rel,
                                          void* stream) {
  if (int st = rm_check(G, P, C, "relation_mean_backward")) return st;
  MPA_REQUIRE(grad_out && edge && rel && out, "relation_mean_backward: null pointer");
  launch(rm_bwd_kernel, dim3((unsigned)G), dim3(128), mpa::as_stream(stream), grad_out, edge, rel, out, (int)P, (int)C,
         grad_edge, grad_rel);
  return mpa::check_launch("relation_mean_backward");
}

static int pr_check(int64_t S, int64_t P, int64_t F, const char* who) {
  MPA_REQUIRE(S >= 1 && P >= 1 && S * P <= (1 << 24) && P <= 4096 && F >= 4 && F % 4 == 0 && F <= (1 << 16),
              "%s: S=%lld P=%lld F=%lld (a multiple of 4) out of range", who, (long long)S, (long long)P, (long long)F);
  return MPA_OK;
}

extern "C" int mpa_pair_rows_forward(const float* a, const float* b, int64_t S, int64_t P, int64_t F, int swap,
                                     float* out, void* stream) {
  if (int st = pr_check(S, P, F, "pair_rows_forward")) return st;
  MPA_REQUIRE(a && b && out, "pair_rows_forward: null pointer");
  launch(pr_fwd_kernel, dim3((unsigned)(S * P)), dim3(256), mpa::as_stream(stream), a, b, (int)P, (int)F, swap, out);
  return mpa::check_launch("pair_rows_forward");
}

extern "C" int mpa_pair_rows_backward(const float* grad_out, int64_t S, int64_t P, int64_t F, int swap, float* grad_a,
                                      float* grad_b, void* stream) {
  if (int st = pr_check(S, P, F, "pair_rows_backward")) return st;
  MPA_REQUIRE(grad_out != nullptr, "pair_rows_backward: null pointer");
  launch(pr_bwd_kernel, dim3((unsigned)(2 * S * P)), dim3(128), mpa::as_stream(stream), grad_out, (int)(S * P), (int)P, (int)F,
         swap, grad_a, grad_b);
  return mpa::check_launch("pair_rows_backward");
}
