// DGCNN part encoder — whole forward and backward (training-mode BatchNorm) as hand-written gfx950 kernels.
//
// Replaces the torch module of the reference (multi_part_assembly/models/modules/encoder/dgcnn.py:41-109):
//   4 x [kNN graph (k = 20) in feature space -> edge features [x_j - x_i ; x_i] -> Conv2d 1x1 -> BatchNorm2d ->
//        LeakyReLU(0.2) -> max over the k neighbours]   widths 3-64-64-128-256
//   concat 512 -> Conv1d 1x1 -> BatchNorm1d -> LeakyReLU(0.2) -> [max ; mean] over the N points -> Linear(2F -> F)
// and the valid-part compaction around it (models/dgl/network.py:90-99): the kernels take ALL part slots plus the
// validity mask, count and compact the valid parts on the device (no host sync) and write zeros for padded slots.
//
// The reference materialises per stage an [n, N, N] score matrix, the gathered neighbours and the [n, 2C, N, 20] edge
// tensor (13 GB at C = 128, n = 640).  None of them exists here:
//   * kNN: Gram tiles on the matrix cores, top-20 kept in registers (dg_knn.h; index-exact, arithmetic pinned);
//   * the 1x1 convolution is linear in the edge feature:  W [x_j - x_i ; x_i] = Wa x_j + (Wb - Wa) x_i = U_j + V_i, so
//     ONE exact-fp32 MFMA GEMM per POINT (X -> [U | V], dg_gemm.h) replaces the GEMM per EDGE (20x fewer FLOPs);
//   * BatchNorm + LeakyReLU is a monotone per-channel map whose direction is the sign of gamma, so
//     max_j act(bn(U_j + V_i)) = act(bn(ext_j U_j + V_i)) with ext = max (gamma >= 0) or min: the aggregation kernel
//     keeps a 32-channel slice of the part's U in LDS (128 KB), gathers the 20 neighbour rows from there (LDS, not
//     L2), and leaves the selected edge value, its slot, the neighbour sum (for backward) and the BatchNorm sums
//     sum(e), sum(e^2) over all 20 edges — computed from sum_j U_j, sum_j U_j^2 and V_i without forming the edges.
// Backward: BatchNorm backward is the affine map  de = alpha*dz + gammap*e + betap  with dz nonzero only on the selected
// edge of every (point, channel):
//   dV_i = alpha dz_i + gammap (sum_j U_j + k V_i) + k betap                      (neighbour sum saved by forward)
//   dU_j = gammap (deg_j U_j + sum_{i: j in nn(i)} V_i) + deg_j betap  +  sum_{i: sel_i = j} alpha dz_i
// Both sums run over the TRANSPOSED kNN graph (built per stage in LDS, in-edges sorted by source) from LDS-resident
// 16-channel panels of V, alpha*dz and the selected slots: an in-edge (i -> j, slot t) contributes alpha dz_i[c] exactly
// for the channels whose selected slot at i is t.  No atomics anywhere, fixed summation order: bit-reproducible.
#include "common.h"
#include "coop_reduce.h"
#include "dg_gemm.h"
#include "dg_gemm_split.h"
#ifndef DG_GEMM_SPLIT  // 1: fp32-grade GEMMs on the bf16 matrix cores (dg_gemm_split.h); 0: v_mfma_f32_32x32x2_f32 (dg_gemm.h)
#define DG_GEMM_SPLIT 1
#endif
#if DG_GEMM_SPLIT
#define DG_NT_KERNEL gemm_nt_split_kernel
#define DG_TN_KERNEL gemm_tn_split_kernel
#define DG_GEMM_THREADS kGsT
#define DG_NT_TAIL , GsEpi{}, 0
#else
#define DG_NT_KERNEL gemm_nt_kernel
#define DG_TN_KERNEL gemm_tn_kernel
#define DG_GEMM_THREADS kGT
#define DG_NT_TAIL
#endif
#include <cstdio>
#include <type_traits>
#include <vector>

#include "dg_knn.h"
#include <stdlib.h>

#include "dg_knn_fast.h"
#include "dg_knn3_gate.h"

namespace {

using namespace dg;

// EdgeConv stage 1's graph (C = 3): the matrix-core gated search (dg_knn3_gate.h) unless MPA_KNN3=scan asks for the
// exhaustive knn3_kernel of rounds 2-5 (identical indices; the knob exists for A/B timing and the cross-check test)
// the kernel that writes a stage's output also leaves the next stage's kNN operands (MPA_KNN_PRODUCER=0: the separate
// rownorm / centre / split kernels of rounds 3-5a; identical operands either way)
bool knn_producer() {
  const char* e = getenv("MPA_KNN_PRODUCER");
  return !(e != nullptr && e[0] == '0');
}
bool knn3_gate() {
  const char* e = getenv("MPA_KNN3");
  return !(e != nullptr && e[0] == 's');
}
using mpa::CoopWs;
using mpa::coop_colsum;
using mpa::kEB;
using mpa::batched_rows;
using mpa::kSlices;

constexpr int kCat = 512;                     // 64 + 64 + 128 + 256
constexpr int kCinP[4] = {4, 64, 64, 128};    // input width of stage l (3 padded to 4)
constexpr int kCin[4] = {3, 64, 64, 128};
constexpr int kCO[4] = {64, 64, 128, 256};
constexpr int kOff[4] = {0, 64, 128, 256};    // column of stage l's output inside the concatenation
constexpr int kTile = 128;                    // rows per block of the row-tiled elementwise kernels
#ifndef DG_TN_CHUNKS
#define DG_TN_CHUNKS 128
#endif
constexpr int kTnChunks = DG_TN_CHUNKS;       // row chunks of the weight-gradient GEMMs
constexpr float kSlope = 0.2f;

// ---- bookkeeping --------------------------------------------------------------------------------------------------------
// hdr[0] = number of valid parts nv, hdr[1] = nv * N; vlist[v] = part slot of the v-th valid part; rank[m] = v or -1.
__global__ void dg_prepare_kernel(const float* __restrict__ valids, int M, int N, int* __restrict__ hdr,
                                  int* __restrict__ vlist, int* __restrict__ rank, unsigned* __restrict__ tickets,
                                  int ntickets) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int m0 = 0; m0 < M; m0 += 64) {
    const int m = m0 + lane;
    const bool flag = m < M && valids[m] != 0.0f;
    const unsigned long long mask = __ballot(flag);
    const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    if (flag) {
      vlist[pos] = m;
      rank[m] = pos;
    } else if (m < M) {
      rank[m] = -1;
    }
    base += __popcll(mask);
  }
  if (lane == 0) {
    hdr[0] = base;
    hdr[1] = base * N;
  }
  for (int t = lane; t < ntickets; t += 64) tickets[t] = 0u;
}

__global__ void dg_gather_points_kernel(const float* __restrict__ points, const int* __restrict__ vlist, int N,
                                        float4* __restrict__ x0, const int* __restrict__ hdr) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= hdr[1]) return;
  const int v = (int)(r / N), p = (int)(r % N);
  const float* s = points + ((long long)vlist[v] * N + p) * 3;
  x0[r] = make_float4(s[0], s[1], s[2], 0.0f);
}

// conv weight w [CO][2C] -> stacked ws [2CO][CP] = [Wa ; Wb - Wa] (U = X Wa^T, V = X (Wb - Wa)^T) and its
// transpose wst [CP][2CO] (operand of the input-gradient GEMM); CP = C padded (3 -> 4).
__global__ void dg_wstack_kernel(const float* __restrict__ w, int CO, int C, int CP, float* __restrict__ ws,
                                 float* __restrict__ wst) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * CO * CP) return;
  const int o = e / CP, c = e % CP;
  float val = 0.0f;
  if (c < C) val = o < CO ? w[o * 2 * C + c] : w[(o - CO) * 2 * C + C + c] - w[(o - CO) * 2 * C + c];
  ws[e] = val;
  wst[c * 2 * CO + o] = val;
}

__global__ void dg_transpose_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ wt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * cols) return;
  wt[(e % cols) * rows + e / cols] = w[e];
}

// d(conv weight) from d(stacked weight) ds [2CO][CP]:  dWa = dS_top - dS_bottom, dWb = dS_bottom
__global__ void dg_wunstack_kernel(const float* __restrict__ ds, int CO, int C, int CP, float* __restrict__ dw) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= CO * 2 * C) return;
  const int o = e / (2 * C), c = e % (2 * C);
  dw[e] = c < C ? ds[o * CP + c] - ds[(CO + o) * CP + c] : ds[(CO + o) * CP + (c - C)];
}

// ---- first stage: K = 3 is far too thin for a matrix core ------------------------------------------------------------------
// uv [R][128] = x0 [R][4] . ws [128][4]^T; thread = (row, 4 outputs).  grid = ceil(Rmax / 8), block 256.
__global__ __launch_bounds__(256) void dg_first_uv_kernel(const float4* __restrict__ x0, const float* __restrict__ ws,
                                                          float* __restrict__ uv, const int* __restrict__ hdr) {
  const long long r = (long long)blockIdx.x * 8 + (threadIdx.x >> 5);
  if (r >= hdr[1]) return;
  const int o4 = threadIdx.x & 31;
  const float4 x = x0[r];
  float out[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float4 w = reinterpret_cast<const float4*>(ws)[4 * o4 + u];
    out[u] = (x.x * w.x + x.y * w.y) + x.z * w.z;
  }
  reinterpret_cast<float4*>(uv + r * 128)[o4] = make_float4(out[0], out[1], out[2], out[3]);
}

// d(ws1) partials: part [tiles][128][4] = sum over the tile's rows of duv[r][o] * x0[r][k].  grid = tiles of 1024 rows,
// block 512 = 4 row groups x 128 outputs, groups merged in fixed order.
constexpr int kFirstTile = 256;  // (1024: 128 us for the 180 MB pass — 350 blocks of 256-row serial loops; 256: ~1400 blocks)
__global__ __launch_bounds__(512) void dg_first_wgrad_kernel(const float* __restrict__ duv, const float4* __restrict__ x0,
                                                             float* __restrict__ part, const int* __restrict__ hdr) {
  __shared__ float4 red[4][128];
  const int R = hdr[1];
  const int g = threadIdx.x >> 7, o = threadIdx.x & 127;
  const long long r0 = (long long)blockIdx.x * kFirstTile + g * (kFirstTile / 4);
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
  const int rows = R - r0 < kFirstTile / 4 ? (int)(R - r0) : kFirstTile / 4;  // (<= 0: nothing)
  struct Row {
    float g;
    float4 x;
  };
  batched_rows<8>(0, 1, rows, [&](int i) { return Row{duv[(r0 + i) * 128 + o], x0[r0 + i]}; },
                  [&](int, const Row t) {
                    a0 = __builtin_fmaf(t.g, t.x.x, a0);
                    a1 = __builtin_fmaf(t.g, t.x.y, a1);
                    a2 = __builtin_fmaf(t.g, t.x.z, a2);
                  });
  red[g][o] = make_float4(a0, a1, a2, 0.0f);
  __syncthreads();
  if (g == 0) {
    float4 t = red[0][o];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      t.x += red[k][o].x;
      t.y += red[k][o].y;
      t.z += red[k][o].z;
    }
    reinterpret_cast<float4*>(part)[(long long)blockIdx.x * 128 + o] = t;
  }
}

// optional gradient w.r.t. the input points: gp [M][N][3] (pre-zeroed), rows of valid parts only
__global__ void dg_first_dgrad_kernel(const float* __restrict__ duv, const float* __restrict__ ws,
                                      const int* __restrict__ vlist, int N, float* __restrict__ gp,
                                      const int* __restrict__ hdr) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= hdr[1]) return;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
  for (int o = 0; o < 128; ++o) {
    const float g = duv[r * 128 + o];
    a0 = __builtin_fmaf(g, ws[4 * o], a0);
    a1 = __builtin_fmaf(g, ws[4 * o + 1], a1);
    a2 = __builtin_fmaf(g, ws[4 * o + 2], a2);
  }
  const int v = (int)(r / N), p = (int)(r % N);
  float* d = gp + ((long long)vlist[v] * N + p) * 3;
  d[0] = a0;
  d[1] = a1;
  d[2] = a2;
}

// ---- edge aggregation, forward ------------------------------------------------------------------------------------------------
// grid = (CO / 32, M parts), block AT = 1024 (512: the A/B baseline).  LDS (dynamic): the part's U slice [N][32] and the
// BatchNorm partial sums of the block's point slots.  Lane = (point slot q = lane / 8, channel quad cq = lane % 8): a wave
// gathers for 8 points at once with ds_read_b128, 16 waves -> 128 points per pass.
#ifndef MPA_AGG_FWD_AT
#define MPA_AGG_FWD_AT 1024
#endif
constexpr size_t agg_fwd_lds(int AT, int N) { return (size_t)N * 32 * sizeof(float) + (size_t)(AT / 8) * 32 * 2 * sizeof(float); }

// Copy a 32-channel column slice (row stride `ld` floats, N rows) between global memory and an LDS panel [N][32] with
// four independent 16-byte requests in flight per thread (a plain loop exposes one L2 round trip per pass).
// `gamma` [32] (or NULL): the panel holds sign(gamma) * value per channel (exact).
template <int AT>
__device__ __forceinline__ void dg_load_slice(const float* __restrict__ src, int ld, int N, float* __restrict__ dst,
                                              const float* __restrict__ gamma = nullptr) {
  for (int e0 = threadIdx.x; e0 < N * 8; e0 += 4 * AT) {
    float4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * AT, ec = e < N * 8 ? e : N * 8 - 1;
      t[u] = *reinterpret_cast<const float4*>(src + (long long)(ec >> 3) * ld + 4 * (ec & 7));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * AT;
      if (e < N * 8) {
        if (gamma != nullptr) {
          const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * (e & 7));
          t[u] = make_float4(gm.x < 0.0f ? -t[u].x : t[u].x, gm.y < 0.0f ? -t[u].y : t[u].y, gm.z < 0.0f ? -t[u].z : t[u].z,
                             gm.w < 0.0f ? -t[u].w : t[u].w);
        }
        *reinterpret_cast<float4*>(&dst[4 * e]) = t[u];
      }
    }
  }
}

template <int AT>
__global__ __launch_bounds__(AT) void dg_agg_fwd_kernel(const float* __restrict__ uv, int CO,
                                                        const unsigned short* __restrict__ idx,
                                                        const float* __restrict__ gamma, int N,
                                                        float* __restrict__ esel, unsigned char* __restrict__ ssel,
                                                        float* __restrict__ s1out, float* __restrict__ partial,
                                                        const int* __restrict__ hdr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char agg_lds[];
  constexpr int kSlots = AT / 8;  // points in flight per pass of the block
  float* Us = reinterpret_cast<float*>(agg_lds);                                    // [N][32]
  float(*red)[32][2] = reinterpret_cast<float(*)[32][2]>(Us + (size_t)N * 32);      // [kSlots][32][2]
  // the channel slices of one part run on ONE XCD (dg::knn_block: grid.y is the part count rounded up to 8): they read
  // the same neighbour lists and neighbouring pieces of the same cache lines, and every XCD has its own L2
  int v, sl;
  dg::knn_block(v, sl);
  if (v >= hdr[0]) return;
  const int c0 = sl * 32;
  const float* up = uv + (long long)v * N * 2 * CO;
  dg_load_slice<AT>(up + c0, 2 * CO, N, Us, gamma + c0);  // the panel holds sign(gamma) * U: the maximum of that is tracked
  __syncthreads();
  // lane = (point slot q of 8, channel quad cq of 8): one ds_read_b128 per neighbour feeds four channels, so the index
  // unpacking and address arithmetic are paid once per four channels (the kernel is bound by VALU issue)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cq = lane & 7, q = lane >> 3;
  float sg[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sg[k] = gamma[c0 + 4 * cq + k] < 0.0f ? -1.0f : 1.0f;
  float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  // the neighbour list (5 x 8 bytes) and V of the NEXT point are requested before the current point's 20 LDS gathers
  unsigned wv[10], wn[10];
  float4 vv, vn = make_float4(0.f, 0.f, 0.f, 0.f);
  auto request = [&](int i, unsigned (&w)[10], float4& vout) {
    const int ic = i < N ? i : N - 1;
    const uint2* ip = reinterpret_cast<const uint2*>(idx + ((long long)v * N + ic) * kNbr);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const uint2 t = ip[u];
      w[2 * u] = t.x;
      w[2 * u + 1] = t.y;
    }
    vout = *reinterpret_cast<const float4*>(up + (long long)ic * 2 * CO + CO + c0 + 4 * cq);
  };
  request(wave * 8 + q, wn, vn);
  for (int i = wave * 8 + q; i < N; i += kSlots) {
    const long long row = (long long)v * N + i;
#pragma unroll
    for (int u = 0; u < 10; ++u) wv[u] = wn[u];
    vv = vn;
    request(i + kSlots, wn, vn);
    float best[4], su[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    int at[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) best[k] = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < kNbr; ++t) {
      const int j = (wv[t >> 1] >> (16 * (t & 1))) & 0xffff;
      const float4 u4 = *reinterpret_cast<const float4*>(&Us[j * 32 + 4 * cq]);
      const float u[4] = {u4.x, u4.y, u4.z, u4.w};  // sign(gamma) * U_j
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (u[k] > best[k]) {
          best[k] = u[k];
          at[k] = t;
        }
        su[k] += u[k];
        sq[k] = __builtin_fmaf(u[k], u[k], sq[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) su[k] *= sg[k];  // (exact) the sum of the unsigned values
    const float vk[4] = {vv.x, vv.y, vv.z, vv.w};
    const long long o = row * CO + c0 + 4 * cq;
    *reinterpret_cast<float4*>(esel + o) = make_float4(sg[0] * best[0] + vk[0], sg[1] * best[1] + vk[1],
                                                       sg[2] * best[2] + vk[2], sg[3] * best[3] + vk[3]);
    *reinterpret_cast<uchar4*>(ssel + o) =
        make_uchar4((unsigned char)at[0], (unsigned char)at[1], (unsigned char)at[2], (unsigned char)at[3]);
    *reinterpret_cast<float4*>(s1out + o) = make_float4(su[0], su[1], su[2], su[3]);
    // BatchNorm sums over the 20 edges e = U_j + V_i:  sum e = S + k V,  sum e^2 = Q + 2 V S + k V^2
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a1[k] += su[k] + (float)kNbr * vk[k];
      a2[k] += sq[k] + 2.0f * vk[k] * su[k] + (float)kNbr * vk[k] * vk[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[wave * 8 + q][4 * cq + k][0] = a1[k];
    red[wave * 8 + q][4 * cq + k][1] = a2[k];
  }
  __syncthreads();
  if (threadIdx.x < 32) {  // thread = channel of the slice; the point slots in fixed order
    const int c = threadIdx.x;
    float s = 0.0f, ss = 0.0f;
    for (int k = 0; k < kSlots; ++k) {
      s += red[k][c][0];
      ss += red[k][c][1];
    }
    float* d = partial + ((long long)v * CO + c0 + c) * 2;
    d[0] = s;
    d[1] = ss;
  }
}

// ---- BatchNorm statistics from a partial table [rows][C][2] -> bn [4][C] (scale, shift, mean, invstd) ------------------------
// rows_are_parts != 0: table row e belongs to valid part e (valid iff e < nv); else to the row tile e (valid iff
// e * kTile < nv * N).  count = nv * N * kmul.  grid = (C / 64, ceil(rows / kEB)), block 1024.
__device__ __forceinline__ int dg_valid_rows(const int* hdr, int rows_are_parts) {
  return rows_are_parts ? hdr[0] : (hdr[1] + kTile - 1) / kTile;
}

__global__ __launch_bounds__(64 * kSlices) void dg_bn_finalize_kernel(
    const float* __restrict__ partial, int rows, int C, int rows_are_parts, int kmul, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
    float eps, float* __restrict__ bn, const CoopWs cw, const int* __restrict__ hdr) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int vrows = dg_valid_rows(hdr, rows_are_parts);
  double s, ss;
  const bool last = coop_colsum(rows, C, c, cw,
                                [&](int e, bool& ok, double& x, double& y) {
                                  ok = e < vrows;
                                  const float2 t = ok ? *reinterpret_cast<const float2*>(partial + ((long long)e * C + c) * 2)
                                                      : make_float2(0.0f, 0.0f);
                                  x = (double)t.x;
                                  y = (double)t.y;
                                },
                                s, ss);
  if (!last || threadIdx.x >= 64) return;
  const double count = (double)hdr[1] * (double)kmul;
  const double mean = count > 0.0 ? s / count : 0.0;
  double var = count > 0.0 ? ss / count - mean * mean : 0.0;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / __builtin_sqrt(var + (double)eps));
  const float scale = gamma[c] * invstd;
  bn[c] = scale;
  bn[C + c] = beta[c] - (float)mean * scale;
  bn[2 * C + c] = (float)mean;
  bn[3 * C + c] = invstd;
  if (running_mean != nullptr && count > 0.0) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void dg_bn_from_running_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
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

// backward coefficients coef [3][C] (alpha, gammap, betap) of  dY = alpha*dZ + gammap*Y + betap  from the two column
// sums (sum dz, sum dz*xhat), and dgamma / dbeta.  Same table conventions as dg_bn_finalize_kernel.
__global__ __launch_bounds__(64 * kSlices) void dg_bwd_coef_kernel(
    const float* __restrict__ partial, int rows, int C, int kmul, const float* __restrict__ gamma,
    const float* __restrict__ bn, float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta,
    const CoopWs cw, const int* __restrict__ hdr) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int vrows = dg_valid_rows(hdr, 0);
  double s1, s2;
  const bool last = coop_colsum(rows, C, c, cw,
                                [&](int e, bool& ok, double& x, double& y) {
                                  ok = e < vrows;
                                  const float2 t = ok ? *reinterpret_cast<const float2*>(partial + ((long long)e * C + c) * 2)
                                                      : make_float2(0.0f, 0.0f);
                                  x = (double)t.x;
                                  y = (double)t.y;
                                },
                                s1, s2);
  if (!last || threadIdx.x >= 64) return;
  const double count = (double)hdr[1] * (double)kmul;
  const float mean = bn[2 * C + c], invstd = bn[3 * C + c];
  const float alpha = gamma[c] * invstd;
  const float gammap = count > 0.0 ? (float)(-(double)alpha * s2 / count * (double)invstd) : 0.0f;
  coef[c] = alpha;
  coef[C + c] = gammap;
  coef[2 * C + c] = count > 0.0 ? (float)(-(double)alpha * s1 / count - (double)gammap * (double)mean) : 0.0f;
  dgamma[c] = (float)s2;
  dbeta[c] = (float)s1;
}

// H[r][off + c] = LeakyReLU(bn(esel[r][c]));  one thread per float4.
__global__ void dg_apply_kernel(const float* __restrict__ esel, int CO, const float* __restrict__ bn,
                                float* __restrict__ hcat, int off, const int* __restrict__ hdr) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int q4 = CO / 4, qshift = __ffs(q4) - 1;  // CO is 64, 128 or 256: shifts instead of a 64-bit division per thread
  const long long r = e >> qshift;
  if (r >= hdr[1]) return;
  const int c = 4 * (int)(e & (q4 - 1));
  const float4 x = *reinterpret_cast<const float4*>(esel + r * CO + c);
  const float4 sc = *reinterpret_cast<const float4*>(bn + c), sh = *reinterpret_cast<const float4*>(bn + CO + c);
  float4 z = make_float4(__builtin_fmaf(x.x, sc.x, sh.x), __builtin_fmaf(x.y, sc.y, sh.y),
                         __builtin_fmaf(x.z, sc.z, sh.z), __builtin_fmaf(x.w, sc.w, sh.w));
  z.x = z.x > 0.0f ? z.x : kSlope * z.x;
  z.y = z.y > 0.0f ? z.y : kSlope * z.y;
  z.z = z.z > 0.0f ? z.z : kSlope * z.z;
  z.w = z.w > 0.0f ? z.w : kSlope * z.w;
  *reinterpret_cast<float4*>(hcat + r * kCat + off + c) = z;
}

// ---- the same pass as the PRODUCER of the next stage's kNN operands (stages whose output is 64 or 128 wide) -------------------
// knn_wide (dg_knn_fast.h) needs, per row of the next stage's input x: the pinned norm n (fmaf chain in matrix-core
// order), the bf16 row hi(x - mu), and the scaled centred norms nl / nu.  rownorm_kernel + knn_centre_kernel +
// knn_split1_kernel read x twice more for that (171 + 92 MB at C = 128).  Here the kernel that WRITES x produces them
// from the registers it holds: a row is owned by C / 4 consecutive lanes (full-line stores of x and of the bf16 row), the
// centred norm is the split kernel's own shuffle tree, and the sequential norm chain is walked by one thread per row
// over the block's rows staged in LDS (odd row stride: conflict-free).  Same values as the three kernels, bit for bit
// (tests/test_dgcnn_gpu.py compares the graphs the encoder builds with mpa_knn_exact's on the same rows).
// mu [clouds][C] first: the mean of the cloud's first 16 OUTPUT rows, from the pre-activation rows (knn_centre_kernel's sum).
template <int C>
__global__ __launch_bounds__(C) void dg_apply_centre_kernel(const float* __restrict__ esel, const float* __restrict__ bn,
                                                            int N, float* __restrict__ mu, const int* __restrict__ hdr) {
  const int v = blockIdx.x, c = threadIdx.x;
  if (v >= hdr[0]) return;
  const float sc = bn[c], sh = bn[C + c];
  const float* xp = esel + (long long)v * N * C + c;
  float a = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {  // fixed order
    float z = __builtin_fmaf(xp[(long long)r * C], sc, sh);
    z = z > 0.0f ? z : kSlope * z;
    a += z;
  }
  mu[v * C + c] = a * 0.0625f;
}

template <int C>
__global__ __launch_bounds__(256) void dg_apply_knn_kernel(const float* __restrict__ esel, const float* __restrict__ bn,
                                                           float* __restrict__ hcat, int off, const float* __restrict__ mu,
                                                           int N, float* __restrict__ norm, unsigned short* __restrict__ xs,
                                                           float* __restrict__ nl, float* __restrict__ nu,
                                                           const int* __restrict__ hdr) {
  constexpr int TPR = C / 4, RPP = 256 / TPR, PASSES = 4, RB = RPP * PASSES, LD = C + 1;
  __shared__ float rowbuf[RB * LD];
  __shared__ float msum[RB], esum[RB];
  const long long R = hdr[1];
  const long long r0 = (long long)blockIdx.x * RB;
  if (r0 >= R) return;
  const int c4 = threadIdx.x % TPR, rl = threadIdx.x / TPR, c = 4 * c4;
  const float4 sc = *reinterpret_cast<const float4*>(bn + c), sh = *reinterpret_cast<const float4*>(bn + C + c);
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int row = ps * RPP + rl;
    const long long r = r0 + row, rc = r < R ? r : R - 1;  // (rows past the end shadow the last one: the shuffles need them)
    const float4 x = *reinterpret_cast<const float4*>(esel + rc * C + c);
    float4 z = make_float4(__builtin_fmaf(x.x, sc.x, sh.x), __builtin_fmaf(x.y, sc.y, sh.y),
                           __builtin_fmaf(x.z, sc.z, sh.z), __builtin_fmaf(x.w, sc.w, sh.w));
    z.x = z.x > 0.0f ? z.x : kSlope * z.x;
    z.y = z.y > 0.0f ? z.y : kSlope * z.y;
    z.z = z.z > 0.0f ? z.z : kSlope * z.z;
    z.w = z.w > 0.0f ? z.w : kSlope * z.w;
    const float4 m4 = *reinterpret_cast<const float4*>(mu + (rc / N) * C + c);
    const float y[4] = {z.x - m4.x, z.y - m4.y, z.z - m4.z, z.w - m4.w};
    kf_bf16x4 hi;
#pragma unroll
    for (int u = 0; u < 4; ++u) hi[u] = (__bf16)y[u];
    float m = (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);  // (knn_split1_kernel's trees)
    const float e[4] = {y[0] - (float)hi[0], y[1] - (float)hi[1], y[2] - (float)hi[2], y[3] - (float)hi[3]};
    float e2 = (e[0] * e[0] + e[1] * e[1]) + (e[2] * e[2] + e[3] * e[3]);
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) {
      m += __shfl_xor(m, o, 64);
      e2 += __shfl_xor(e2, o, 64);
    }
    float* d = &rowbuf[row * LD + c];
    d[0] = z.x, d[1] = z.y, d[2] = z.z, d[3] = z.w;
    if (c4 == 0) msum[row] = m, esum[row] = e2;
    if (r < R) {
      *reinterpret_cast<float4*>(hcat + r * kCat + off + c) = z;
      *reinterpret_cast<kf_bf16x4*>(xs + r * C + c) = hi;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < RB && r0 + threadIdx.x < R) {  // the pinned chain: positions i, C/2 + i for i = 0 .. C/2 - 1
    const float* x = &rowbuf[threadIdx.x * LD];
    float n = 0.0f;
#pragma unroll 16
    for (int i = 0; i < C / 2; ++i) {
      const float lo = x[i], hh = x[C / 2 + i];
      n = __builtin_fmaf(lo, lo, n);
      n = __builtin_fmaf(hh, hh, n);
    }
    const long long r = r0 + threadIdx.x;
    norm[r] = n;
    kf_scaled_norms<C>(msum[threadIdx.x], esum[threadIdx.x], n, nl[r], nu[r]);
  }
}

// ---- tail: column statistics of y5, pooling, linear ----------------------------------------------------------------------------
// The row-tiled passes below run 512 threads per tile of kTile rows: thread = (row group, channel), the 512 / F row groups
// walk every (512 / F)-th row (F threads per tile left 128 dependent-address row loads to two waves).
constexpr int kTileT = 512;

// the row groups' (a, b) added in group order by group 0 -> partial[tile][F][2]
__device__ __forceinline__ void tile_pair_out(float a, float b, int F, int g, int c, float* __restrict__ partial) {
  __shared__ float red[kTileT][2];
  red[g * F + c][0] = a;
  red[g * F + c][1] = b;
  __syncthreads();
  if (g == 0) {
    for (int q = 1; q < kTileT / F; ++q) {
      a += red[q * F + c][0];
      b += red[q * F + c][1];
    }
    float* d = partial + ((long long)blockIdx.x * F + c) * 2;
    d[0] = a;
    d[1] = b;
  }
}

// partial [tiles][F][2] = (sum y, sum y^2) over the tile's rows.  grid = ceil(Rmax / kTile), block = kTileT.
__global__ __launch_bounds__(kTileT) void dg_colstats_kernel(const float* __restrict__ y, int F, float* __restrict__ partial,
                                                             const int* __restrict__ hdr) {
  const int R = hdr[1];
  const long long r0 = (long long)blockIdx.x * kTile;
  if (r0 >= R) return;
  const int c = threadIdx.x % F, g = threadIdx.x / F, G = kTileT / F;
  float s = 0.0f, ss = 0.0f;
  const int rows = R - r0 < kTile ? (int)(R - r0) : kTile;
  batched_rows<8>(g, G, rows, [&](int i) { return y[(r0 + i) * F + c]; },
                  [&](int, float t) {
                    s += t;
                    ss = __builtin_fmaf(t, t, ss);
                  });
  tile_pair_out(s, ss, F, g, c, partial);
}

// pooled [nv][2F] = [max_n a ; mean_n a] with a = LeakyReLU(bn(y)); arg [nv][F] = row of the (first) maximum.
// grid = M parts, block 1024 = (1024 / F) row groups x F channels; groups merged in fixed order.
__global__ __launch_bounds__(1024) void dg_pool_kernel(const float* __restrict__ y, int F, int N,
                                                      const float* __restrict__ bn, float* __restrict__ pooled,
                                                      int* __restrict__ arg, const int* __restrict__ hdr) {
  __shared__ float smax[1024], ssum[1024];
  __shared__ int sarg[1024];
  const int v = blockIdx.x;
  if (v >= hdr[0]) return;
  const int G = 1024 / F, g = threadIdx.x / F, c = threadIdx.x % F;
  const float scale = bn[c], shift = bn[F + c];
  const float* yp = y + (long long)v * N * F;
  float best = -__builtin_inff(), sum = 0.0f;
  int at = 0;
  batched_rows<8>(g, G, N, [&](int n) { return yp[(long long)n * F + c]; },
                  [&](int n, float t) {
                    const float z = __builtin_fmaf(t, scale, shift);
                    const float a = z > 0.0f ? z : kSlope * z;
                    sum += a;
                    if (a > best) {
                      best = a;
                      at = n;
                    }
                  });
  smax[threadIdx.x] = best;
  ssum[threadIdx.x] = sum;
  sarg[threadIdx.x] = at;
  __syncthreads();
  if (g == 0) {
    for (int k = 1; k < G; ++k) {
      const float b2 = smax[k * F + c];
      const int a2 = sarg[k * F + c];
      if (b2 > best || (b2 == best && a2 < at)) {
        best = b2;
        at = a2;
      }
      sum += ssum[k * F + c];
    }
    pooled[(long long)v * 2 * F + c] = best;
    pooled[(long long)v * 2 * F + F + c] = sum / (float)N;
    arg[(long long)v * F + c] = at;
  }
}

// feat [M][F] = pooled[rank[m]] . W^T + b (zeros for padded parts).  grid = M, block = F.
__global__ void dg_fc_kernel(const float* __restrict__ pooled, const int* __restrict__ rank, int F,
                             const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ feat) {
  __shared__ float row[512];
  const int m = blockIdx.x, f = threadIdx.x, v = rank[m];
  if (v < 0) {
    feat[(long long)m * F + f] = 0.0f;
    return;
  }
  for (int k = f; k < 2 * F; k += F) row[k] = pooled[(long long)v * 2 * F + k];
  __syncthreads();
  float acc = 0.0f;
  const float4* wr = reinterpret_cast<const float4*>(w + (long long)f * 2 * F);
  for (int k4 = 0; k4 < F / 2; ++k4) {
    const float4 t = wr[k4];
    acc = __builtin_fmaf(row[4 * k4], t.x, acc);
    acc = __builtin_fmaf(row[4 * k4 + 1], t.y, acc);
    acc = __builtin_fmaf(row[4 * k4 + 2], t.z, acc);
    acc = __builtin_fmaf(row[4 * k4 + 3], t.w, acc);
  }
  feat[(long long)m * F + f] = acc + b[f];
}

// dpooled [nv][2F] = gfeat[vlist[v]] . W.  grid = M (v < nv), block = 2F.
__global__ void dg_fc_bwd_in_kernel(const float* __restrict__ gfeat, const int* __restrict__ vlist, int F,
                                    const float* __restrict__ w, float* __restrict__ dpooled,
                                    const int* __restrict__ hdr) {
  __shared__ float g[256];
  const int v = blockIdx.x;
  if (v >= hdr[0]) return;
  const int k = threadIdx.x;
  if (k < F) g[k] = gfeat[(long long)vlist[v] * F + k];
  __syncthreads();
  float acc = 0.0f;
  for (int f = 0; f < F; ++f) acc = __builtin_fmaf(g[f], w[(long long)f * 2 * F + k], acc);
  dpooled[(long long)v * 2 * F + k] = acc;
}

// dW [F][2F], db [F]: block f, 1024 threads = (1024 / 2F) part groups x 2F columns; groups merged in fixed order.
__global__ __launch_bounds__(1024) void dg_fc_bwd_w_kernel(const float* __restrict__ gfeat, const int* __restrict__ vlist,
                                                           int F, const float* __restrict__ pooled,
                                                           float* __restrict__ dw, float* __restrict__ db,
                                                           const int* __restrict__ hdr) {
  __shared__ float red[1024], redb[8];
  const int f = blockIdx.x, W = 2 * F, G = 1024 / W, g = threadIdx.x / W, k = threadIdx.x % W, nv = hdr[0];
  float acc = 0.0f, sb = 0.0f;
  batched_rows<4>(g, G, nv,
                  [&](int v) { return make_float2(gfeat[(long long)vlist[v] * F + f], pooled[(long long)v * W + k]); },
                  [&](int, const float2 t) {
                    acc = __builtin_fmaf(t.x, t.y, acc);
                    sb += t.x;
                  });
  red[threadIdx.x] = acc;
  if (k == 0) redb[g] = sb;
  __syncthreads();
  if (g == 0) {
    for (int q = 1; q < G; ++q) acc += red[q * W + k];
    dw[(long long)f * W + k] = acc;
    if (k == 0) {
      for (int q = 1; q < G; ++q) sb += redb[q];
      db[f] = sb;
    }
  }
}

// gradient reaching a = LeakyReLU(bn(y5)) from the pooling: dmean / N everywhere + dmax at the arg-max row
__device__ __forceinline__ float dg_tail_dz(float y, float scale, float shift, float dmax, float dmean_n, bool is_arg) {
  const float z = __builtin_fmaf(y, scale, shift);
  return (z > 0.0f ? 1.0f : kSlope) * (dmean_n + (is_arg ? dmax : 0.0f));
}

// what one (row, channel) of the tail's backward passes reads
struct DgTailRow {
  float y, dmax, dmean;
  bool is_arg;
};
__device__ __forceinline__ DgTailRow dg_tail_load(const float* y, const float* __restrict__ dpooled,
                                                  const int* __restrict__ arg, long long r, int N, int F, int c) {
  const int ri = (int)r, v = ri / N, p = ri - v * N;  // (rows < 2^31: 32-bit division, a fifth of the 64-bit one's instructions)
  return DgTailRow{y[r * F + c], dpooled[(long long)v * 2 * F + c], dpooled[(long long)v * 2 * F + F + c],
                   arg[(long long)v * F + c] == p};
}

// partial [tiles][F][2] = (sum dz, sum dz*xhat).  grid = ceil(Rmax / kTile), block = kTileT.
__global__ __launch_bounds__(kTileT) void dg_tail_bwd_sums_kernel(const float* __restrict__ y, int F, int N,
                                                                  const float* __restrict__ bn,
                                                                  const float* __restrict__ dpooled,
                                                                  const int* __restrict__ arg, float* __restrict__ partial,
                                                                  const int* __restrict__ hdr) {
  const int R = hdr[1];
  const long long r0 = (long long)blockIdx.x * kTile;
  if (r0 >= R) return;
  const int c = threadIdx.x % F, g = threadIdx.x / F, G = kTileT / F;
  const float scale = bn[c], shift = bn[F + c], mean = bn[2 * F + c], invstd = bn[3 * F + c];
  const int rows = R - r0 < kTile ? (int)(R - r0) : kTile;
  float s1 = 0.0f, s2 = 0.0f;
  batched_rows<4>(g, G, rows, [&](int i) { return dg_tail_load(y, dpooled, arg, r0 + i, N, F, c); },
                  [&](int, const DgTailRow t) {
                    const float dz = dg_tail_dz(t.y, scale, shift, t.dmax, t.dmean / (float)N, t.is_arg);
                    s1 += dz;
                    s2 = __builtin_fmaf(dz, (t.y - mean) * invstd, s2);
                  });
  tile_pair_out(s1, s2, F, g, c, partial);
}

// y5 <- dY5 = alpha*dz + gammap*y + betap, in place.  grid = ceil(Rmax / kTile), block = kTileT.
__global__ __launch_bounds__(kTileT) void dg_tail_bwd_apply_kernel(float* __restrict__ y, int F, int N,
                                                                   const float* __restrict__ bn,
                                                                   const float* __restrict__ coef,
                                                                   const float* __restrict__ dpooled,
                                                                   const int* __restrict__ arg,
                                                                   const int* __restrict__ hdr) {
  const int R = hdr[1];
  const long long r0 = (long long)blockIdx.x * kTile;
  if (r0 >= R) return;
  const int c = threadIdx.x % F, g = threadIdx.x / F, G = kTileT / F;
  const float scale = bn[c], shift = bn[F + c];
  const float alpha = coef[c], gammap = coef[F + c], betap = coef[2 * F + c];
  const int rows = R - r0 < kTile ? (int)(R - r0) : kTile;
  batched_rows<4>(g, G, rows, [&](int i) { return dg_tail_load(y, dpooled, arg, r0 + i, N, F, c); },
                  [&](int i, const DgTailRow t) {
                    const float dz = dg_tail_dz(t.y, scale, shift, t.dmax, t.dmean / (float)N, t.is_arg);
                    y[(r0 + i) * F + c] = __builtin_fmaf(alpha, dz, __builtin_fmaf(gammap, t.y, betap));
                  });
}

// ---- edge aggregation, backward ----------------------------------------------------------------------------------------------
// dz = dH * LeakyReLU'(z) (z > 0 <=> H > 0) and the two BatchNorm-backward sums over the selected edges.
// grid = ceil(Rmax / kTile), block = 512: thread = (row group, channel), the 512 / CO row groups of a tile walk every
// (512 / CO)-th row and are added in group order (one wave per tile left the loads of 128 rows to a single wave: 167 us
// per launch where the 270-540 MB of the pass stream in 80-160).
__global__ __launch_bounds__(512) void dg_agg_bwd_sums_kernel(const float* __restrict__ hcat,
                                                              const float* __restrict__ dhcat, int off, int CO,
                                                              const float* __restrict__ esel, const float* __restrict__ bn,
                                                              float* __restrict__ dz, float* __restrict__ partial,
                                                              const int* __restrict__ hdr) {
  __shared__ float red[512][2];  // [row group][channel]
  const int R = hdr[1];
  const long long r0 = (long long)blockIdx.x * kTile;
  if (r0 >= R) return;
  const int c = threadIdx.x % CO, g = threadIdx.x / CO, G = 512 / CO;
  const float mean = bn[2 * CO + c], invstd = bn[3 * CO + c];
  const int rows = R - r0 < kTile ? (int)(R - r0) : kTile;
  float s1 = 0.0f, s2 = 0.0f;
  batched_rows<4>(g, G, rows,
                  [&](int i) {
                    const long long r = r0 + i;
                    return make_float3(hcat[r * kCat + off + c], dhcat[r * kCat + off + c], esel[r * CO + c]);
                  },
                  [&](int i, const float3 t) {
                    const float d = t.y * (t.x > 0.0f ? 1.0f : kSlope);
                    dz[(r0 + i) * CO + c] = d;
                    s1 += d;
                    s2 = __builtin_fmaf(d, (t.z - mean) * invstd, s2);
                  });
  red[g * CO + c][0] = s1;
  red[g * CO + c][1] = s2;
  __syncthreads();
  if (g == 0) {
    for (int q = 1; q < G; ++q) {
      s1 += red[q * CO + c][0];
      s2 += red[q * CO + c][1];
    }
    float* p = partial + ((long long)blockIdx.x * CO + c) * 2;
    p[0] = s1;
    p[1] = s2;
  }
}

// Transposed kNN graph of every part, points in the order of descending in-degree: order [M][N] (rank -> point),
// rptr [M][N + 1] (rank -> offset), rlist [R][20] = the in-edges of every point as (source point * 32 + neighbour slot
// of the target in the source's list), ascending (fixed summation order downstream).  grid = M parts, block 1024; the part's whole list (N * 20 sources,
// 40 KB) is built and sorted in LDS and written out once, coalesced.
// (a device function: the blocks that run it are part of dg_bwd_head_kernel's grid)
__device__ __forceinline__ void dg_reverse_part(const unsigned short* __restrict__ idx, int N, int* __restrict__ rptr,
                                                int* __restrict__ order, unsigned short* __restrict__ rlist, int v) {
  __shared__ int cnt[kMaxN];
  __shared__ int beg[kMaxN + 1];
  __shared__ unsigned short lst[kMaxN * kNbr];
  __shared__ int wsum[16];
  __shared__ int carry;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int E = N * kNbr;
  const unsigned short* id = idx + (long long)v * E;
  for (int j = t; j < N; j += 1024) cnt[j] = 0;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int e = t; e < E; e += 1024) atomicAdd(&cnt[id[e]], 1);  // integer counts: order-independent
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {  // exclusive prefix sum, 1024 rows at a time
    const int j = base + t, val = j < N ? cnt[j] : 0;
    int inc = val;
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
      beg[j] = before + inc - val;
      cnt[j] = before + inc - val;  // becomes the row's fill cursor
    }
    __syncthreads();
    if (t == 1023) carry = before + inc;
    __syncthreads();
  }
  if (t == 0) beg[N] = carry;
  for (int e = t; e < E; e += 1024) {
    const int pos = atomicAdd(&cnt[id[e]], 1);
    lst[pos] = (unsigned short)((e / kNbr) * 32 + e % kNbr);  // source point and its neighbour slot
  }
  __syncthreads();
  for (int j = t; j < N; j += 1024) {  // ascending order inside every row (insertion sort in LDS: ~20 entries)
    const int b = beg[j], e = cnt[j];
    for (int a = b + 1; a < e; ++a) {
      const unsigned short key = lst[a];
      int q = a - 1;
      while (q >= b && lst[q] > key) {
        lst[q + 1] = lst[q];
        --q;
      }
      lst[q + 1] = key;
    }
  }
  __syncthreads();
  // Points in the order of DESCENDING in-degree (ties: ascending index): the backward gathers 16 points per wave pass
  // and a pass takes as many rounds as its longest list — with the points in index order that is ~35 rounds for a
  // mean of 20 in-edges; in degree order the lists of a pass are equally long.  Bitonic sort of (511 - degree) << 10 |
  // point in LDS (1024 keys, one per thread), then the lists are written out in that order.
  __shared__ unsigned key[kMaxN];
  {
    const int deg = t < N ? cnt[t] - beg[t] : 0;
    key[t] = t < N ? ((unsigned)(511 - (deg > 511 ? 511 : deg)) << 10) | (unsigned)t : 0xffffffffu;
  }
  __syncthreads();
  for (int k = 2; k <= kMaxN; k <<= 1) {
    for (int jj = k >> 1; jj > 0; jj >>= 1) {
      const int partner = t ^ jj;
      if (partner > t) {
        const unsigned a = key[t], b = key[partner];
        const bool up = (t & k) == 0;
        if ((a > b) == up) {
          key[t] = b;
          key[partner] = a;
        }
      }
      __syncthreads();
    }
  }
  // rank r -> point order[r]; offsets of the lists in rank order (exclusive scan of the degrees in rank order)
  const int pj = t < N ? (int)(key[t] & 1023u) : 0;
  const int dg_ = t < N ? cnt[pj] - beg[pj] : 0;
  int inc = dg_;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < wave; ++w) before += wsum[w];
  const int start = before + inc - dg_;
  int* rp = rptr + (long long)v * (N + 1);
  int* ord = order + (long long)v * N;
  unsigned short* rl = rlist + (long long)v * E;
  if (t < N) {
    rp[t] = start;
    ord[t] = pj;
    const int src0 = beg[pj];
    for (int a = 0; a < dg_; ++a) rl[start + a] = lst[src0 + a];
  }
  if (t == N - 1) rp[N] = start + dg_;
}

// First launch of the EdgeConv backward: the transposed graphs of ALL FOUR stages (they depend on the forward's
// neighbour lists only) and the BatchNorm-backward sums of the last stage in ONE grid of 1024-thread blocks.  The graph
// blocks are latency chains in LDS (count, scan, insertion sorts, a 1024-key bitonic sort: ~80 us a block, 0.33 ms as four
// launches of one block per part), the sums blocks stream 2 GB: together the CUs hold both kinds and the graph build hides
// behind the stream.  Blocks [0, 4 M): (stage, part) = (b / M, b % M); blocks [4 M, ...): 128-row tiles of the sums.
struct DgRevArgs {
  const unsigned short* idx[4];
  int* rptr[4];
  int* order[4];
  unsigned short* rlist[4];
};
__global__ __launch_bounds__(1024) void dg_bwd_head_kernel(const DgRevArgs ra, int M, int N, const float* __restrict__ hcat,
                                                           const float* __restrict__ dhcat, int off, int CO,
                                                           const float* __restrict__ esel, const float* __restrict__ bn,
                                                           float* __restrict__ dz, float* __restrict__ partial,
                                                           const int* __restrict__ hdr) {
  const int b = (int)blockIdx.x;
  if (b < 4 * M) {
    const int st = b / M, v = b % M;
    if (v >= hdr[0]) return;
    dg_reverse_part(ra.idx[st], N, ra.rptr[st], ra.order[st], ra.rlist[st], v);
    return;
  }
  // dz = dH * LeakyReLU'(z) and the two sums of dg_agg_bwd_sums_kernel for tile b - 4 M: thread = (row group, channel)
  __shared__ float red[1024][2];  // [row group][channel]
  const int R = hdr[1];
  const long long r0 = (long long)(b - 4 * M) * kTile;
  if (r0 >= R) return;
  const int c = threadIdx.x % CO, g = threadIdx.x / CO, G = 1024 / CO;
  const float mean = bn[2 * CO + c], invstd = bn[3 * CO + c];
  const int rows = R - r0 < kTile ? (int)(R - r0) : kTile;
  float s1 = 0.0f, s2 = 0.0f;
  batched_rows<4>(g, G, rows,
                  [&](int i) {
                    const long long r = r0 + i;
                    return make_float3(hcat[r * kCat + off + c], dhcat[r * kCat + off + c], esel[r * CO + c]);
                  },
                  [&](int i, const float3 t) {
                    const float d = t.y * (t.x > 0.0f ? 1.0f : kSlope);
                    dz[(r0 + i) * CO + c] = d;
                    s1 += d;
                    s2 = __builtin_fmaf(d, (t.z - mean) * invstd, s2);
                  });
  red[g * CO + c][0] = s1;
  red[g * CO + c][1] = s2;
  __syncthreads();
  if (g == 0) {  // the row groups in fixed order
    for (int q = 1; q < G; ++q) {
      s1 += red[q * CO + c][0];
      s2 += red[q * CO + c][1];
    }
    float* p = partial + ((long long)(b - 4 * M) * CO + c) * 2;
    p[0] = s1;
    p[1] = s2;
  }
}

// d(uv) [R][2CO]:  dU_j = gammap (deg_j U_j + sum V_i) + deg_j betap + sum_sel W_i,   dV_j = W_j + gammap (S1_j + k V_j) + k betap
// with W = alpha * dz, `sum V_i` over the in-edges (i -> j) of point j and `sum_sel W_i[c]` over the sources whose
// SELECTED neighbour at channel c is j.  grid = (CO / 16, M), block 1024, one 16-channel slice of one part.
//   Phase A, from the source side: every (source, channel) has exactly one selected neighbour, so the selected-edge sum
//   is N x 16 contributions per block — a twentieth of the (in-edge, channel) tests the transposed-graph walk of rounds
//   2-4 spent on it.  They are scattered with LDS atomics on 64-bit FIXED-POINT words (integer addition commutes: the
//   result does not depend on the order the atomics land in, so the pass stays bit-reproducible; no float atomics).  The
//   scale is a power of two per channel, 2^40 / (the next power of two above the part's largest |W|): every addend is
//   exact to 2^-40 of the channel's largest value (truncated there), the sum is rounded to fp32 once.  A non-finite W turns the channel's
//   sums of the part into NaN (the float chain would have propagated it to some rows).
//   Phase B, over the transposed graph: lane = (point of 16, channel quad) adds V_i of its point's in-edges in ascending
//   source order (one 16-byte LDS read and four additions per in-edge), then writes both gradients.
#ifdef MPA_AGG_STATS  // instrumented build for tools/probe_agg_stats.py only (never in libmpa_hip.so)
constexpr int kAggRec = 1 << 16;
__device__ unsigned long long g_agg_rec[kAggRec][8];  // one record per block (plain stores: shared counters would congest the memory system and distort what they measure)
__device__ unsigned long long g_agg_stats[8];  // blocks, ticks: phase A, passes (sum over waves), total; wave passes, loop iterations; ticks to phase A's first / second barrier
#define AGG_TICK() __builtin_readcyclecounter()
#else
#define AGG_TICK() 0ull
#endif
#ifndef MPA_AGG_BWD_UNROLL
#define MPA_AGG_BWD_UNROLL 4  // in-edges per trip of the walk: their scratch reads, then their panel reads, go out together
#endif
#ifndef MPA_AGG_BS
#define MPA_AGG_BS 16
#endif
constexpr int kBS = MPA_AGG_BS;   // channels per slice (16 | 8)
constexpr int kCQ = kBS / 4;      // lanes per point (a lane owns four channels)
constexpr int kPts = 64 / kCQ;    // points per wave pass
constexpr int kRun = kBS == 16 ? 512 : 768;  // scratch entries per wave (a pass's points x ~20 in-edges, with room; hubs overflow to global reads)
constexpr int kAggT = 64 * kBS;   // threads: N * kCQ <= 4 * kAggT elements in phase A
// dynamic LDS for parts of N points: phase A's accumulators [N][16] u64; phase B's panels Sd | V [(N + 1)][16] float, a
// scratch run per wave and the in-edge offsets
constexpr size_t agg_bwd_lds(int N) {
  const size_t a = (size_t)N * kBS * 8;
  const size_t b = (size_t)(N + 1) * kBS * 2 * sizeof(float) + (size_t)(kAggT / 64) * kRun * 2 + (size_t)(2 * N + 2) * 2;
  return a > b ? a : b;
}
// Block barrier for exchanges through LDS only: waits for the wave's LDS traffic, NOT for its global loads in flight
// (__syncthreads() drains vmcnt too, which would stall every prefetch of the kernel below at the next barrier).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__global__ __launch_bounds__(kAggT) void dg_agg_bwd_kernel(const float* __restrict__ uv, int CO,
                                                           const int* __restrict__ rptr, const int* __restrict__ order,
                                                           const unsigned short* __restrict__ rlist,
                                                           const unsigned short* __restrict__ idx,
                                                           const float* __restrict__ dz,
                                                           const unsigned char* __restrict__ ssel,
                                                           const float* __restrict__ s1in, const float* __restrict__ coef,
                                                           int M, int N, float* __restrict__ guv,
                                                           const int* __restrict__ hdr) {
  constexpr int AT = kAggT;
  extern __shared__ __attribute__((aligned(16))) unsigned char agg_lds[];
  __shared__ int pass_ctr;
  __shared__ __attribute__((aligned(16))) unsigned amax_s[AT / 64][kBS];
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(agg_lds);         // phase A: [N][kBS]
  float* Sd = reinterpret_cast<float*>(agg_lds);                                   // phase B: [(N + 1)][kBS]
  float* Vp = Sd + (N + 1) * kBS;                                                  // [(N + 1)][kBS]
  unsigned short* scr_all = reinterpret_cast<unsigned short*>(Vp + (N + 1) * kBS);  // [AT / 64][kRun]
  unsigned short* rps = scr_all + (AT / 64) * kRun;                                // [N + 2]: offsets < 20 N <= 20480
  unsigned short* ordl = rps + (N + 2);                                            // [N]: the point of every degree rank
  // the 16-channel slices of one part run on ONE XCD (dg::knn_block): a slice's rows are 64-byte halves (V, dz) and
  // 16-byte eighths (selected slots) of cache lines whose other parts the neighbouring slices read
  // Everything the block reads from global memory before its first pass is requested in TWO rounds, up front: with all
  // 160 KB of LDS taken, a CU holds one block and nothing else hides a round trip (three in a row per phase cost a
  // third of the kernel).  Round 1 goes out before the valid-part count arrives (a part slot behind it is allocated:
  // the loads are harmless), round 2 (the selected neighbours, the first pass's lists and rows) as soon as round 1 is in.
  int v, sl;
  dg::knn_block(v, sl);
  const int nv = hdr[0];
  const int vc = v < M ? v : M - 1;
  const unsigned long long tk0 = AGG_TICK();
  unsigned long long n_pass = 0, n_iter = 0;
  const int c0 = sl * kBS;
  const float* up = uv + (long long)vc * N * 2 * CO;
  float* gp = guv + (long long)vc * N * 2 * CO;
  const float* dzp = dz + (long long)vc * N * CO;
  const unsigned char* sp = ssel + (long long)vc * N * CO;
  const unsigned short* ip = idx + (long long)vc * N * kNbr;
  const int* rpg = rptr + (long long)vc * (N + 1);
  const int* ord = order + (long long)vc * N;
  const unsigned short* rl = rlist + (long long)vc * N * kNbr;
  const float* s1p = s1in + (long long)vc * N * CO;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, cq = lane % kCQ, q = lane / kCQ;
  // ---- phase A.  Element e = (source e / kCQ, channel quad e % kCQ = cq); N <= 1024: at most four per thread.
  // Loads return in issue order: the chain the scatter waits for (selected slots -> selected neighbours) goes first, what
  // is needed later (V, the neighbour sums, the first pass's lists) behind it.
  float4 tw[4];
  unsigned ts[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = threadIdx.x + u * AT, ec = e < N * kCQ ? e : N * kCQ - 1;
    ts[u] = *reinterpret_cast<const unsigned*>(sp + (long long)(ec / kCQ) * CO + c0 + 4 * cq);
  }
  // the first pass of every wave is fixed (pass = wave), the others are handed out by a counter: its lists and rows
  const int r_first = kPts * wave;
  const int base_f = rpg[r_first < N ? r_first : N];
  int jn = ord[r_first + q < N ? r_first + q : N - 1];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = threadIdx.x + u * AT, ec = e < N * kCQ ? e : N * kCQ - 1;
    tw[u] = *reinterpret_cast<const float4*>(dzp + (long long)(ec / kCQ) * CO + c0 + 4 * cq);
  }
  const float4 alpha = *reinterpret_cast<const float4*>(coef + c0 + 4 * cq);
  const float4 gammap = *reinterpret_cast<const float4*>(coef + CO + c0 + 4 * cq);
  const float4 betap = *reinterpret_cast<const float4*>(coef + 2 * CO + c0 + 4 * cq);
  const int rp_a = rpg[threadIdx.x < N ? threadIdx.x : N];      // the in-edge offsets and the rank order, for the LDS tables
  const int rp_b = threadIdx.x == 0 ? rpg[N] : 0;
  const int or_a = ord[threadIdx.x < N ? threadIdx.x : N - 1];
  if (v >= nv) return;
  unsigned short tj[4][4];  // the selected neighbours
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = threadIdx.x + u * AT, ec = e < N * kCQ ? e : N * kCQ - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned slot = (ts[u] >> (8 * k)) & 0xffu;
      tj[u][k] = ip[(ec / kCQ) * kNbr + (slot < (unsigned)kNbr ? slot : 0u)];
    }
  }
  unsigned tjp[4][2];  // (two to a register)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    tjp[u][0] = (unsigned)tj[u][0] | ((unsigned)tj[u][1] << 16);
    tjp[u][1] = (unsigned)tj[u][2] | ((unsigned)tj[u][3] << 16);
  }
  const int last = N * kNbr - 1;
  unsigned short pre[kRun / 64];
  float4 un;
#pragma unroll
  for (int u = 0; u < kRun / 64; ++u) pre[u] = rl[base_f + 64 * u + lane < last ? base_f + 64 * u + lane : last];
  auto request_rows = [&]() { un = *reinterpret_cast<const float4*>(up + (long long)jn * 2 * CO + c0 + 4 * cq); };
  request_rows();
  auto load_v = [&](int u) {  // V of the same elements
    const int e = threadIdx.x + u * AT, ec = e < N * kCQ ? e : N * kCQ - 1;
    return *reinterpret_cast<const float4*>(up + (long long)(ec / kCQ) * 2 * CO + CO + c0 + 4 * cq);
  };
  const float4 tv0 = load_v(0), tv1 = load_v(1), tv2 = load_v(2), tv3 = load_v(3);
  auto load_s1 = [&](int u) {  // the forward's neighbour sum of the same elements
    const int e = threadIdx.x + u * AT, ec = e < N * kCQ ? e : N * kCQ - 1;
    return *reinterpret_cast<const float4*>(s1p + (long long)(ec / kCQ) * CO + c0 + 4 * cq);
  };
  const float4 ts0 = load_s1(0), ts1 = load_s1(1), ts2 = load_s1(2), ts3 = load_s1(3);
  for (int e = threadIdx.x; e < N * kBS / 2; e += AT) reinterpret_cast<uint4*>(acc)[e] = make_uint4(0u, 0u, 0u, 0u);
  unsigned mx[4] = {0u, 0u, 0u, 0u};  // the bit pattern of |W| orders like its value, NaN above infinity
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    tw[u] = make_float4(alpha.x * tw[u].x, alpha.y * tw[u].y, alpha.z * tw[u].z, alpha.w * tw[u].w);
    if (threadIdx.x + u * AT < N * kCQ) {
      const unsigned b[4] = {__float_as_uint(tw[u].x) & 0x7fffffffu, __float_as_uint(tw[u].y) & 0x7fffffffu,
                             __float_as_uint(tw[u].z) & 0x7fffffffu, __float_as_uint(tw[u].w) & 0x7fffffffu};
#pragma unroll
      for (int k = 0; k < 4; ++k) mx[k] = b[k] > mx[k] ? b[k] : mx[k];
    }
  }
  // dV needs nothing of the graph: written here, rows in storage order, from the operands phase A holds anyway
  auto put_dv = [&](int u, const float4 w, const float4 vv, const float4 s4) {
    const int e = threadIdx.x + u * AT;
    if (e < N * kCQ) {
      const float kf = (float)kNbr;
      float4 dv;
      dv.x = w.x + __builtin_fmaf(gammap.x, s4.x + kf * vv.x, kf * betap.x);
      dv.y = w.y + __builtin_fmaf(gammap.y, s4.y + kf * vv.y, kf * betap.y);
      dv.z = w.z + __builtin_fmaf(gammap.z, s4.z + kf * vv.z, kf * betap.z);
      dv.w = w.w + __builtin_fmaf(gammap.w, s4.w + kf * vv.w, kf * betap.w);
      *reinterpret_cast<float4*>(gp + (long long)(e / kCQ) * 2 * CO + CO + c0 + 4 * cq) = dv;
    }
  };
#pragma unroll
  for (int off = kCQ; off < 64; off <<= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned t = (unsigned)__shfl_xor((int)mx[k], off, 64);
      mx[k] = t > mx[k] ? t : mx[k];
    }
  if (lane < kCQ) *reinterpret_cast<uint4*>(&amax_s[wave][4 * lane]) = make_uint4(mx[0], mx[1], mx[2], mx[3]);
  lds_barrier();  // the accumulators are zero, the waves' maxima are in
  const unsigned long long tka = AGG_TICK();
  // |W| < 2^(ex - 126) for every W of the channel (ex: the biased exponent of the largest |W|, a denormal counts as 1):
  // addend = W * 2^(166 - ex), truncated towards zero — integer arithmetic on the float's fields, |addend| < 2^40
  int ex0, ex1, ex2, ex3;
  {
    uint4 m = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int w = 0; w < AT / 64; ++w) {
      const uint4 t = *reinterpret_cast<const uint4*>(&amax_s[w][4 * cq]);
      m.x = t.x > m.x ? t.x : m.x;
      m.y = t.y > m.y ? t.y : m.y;
      m.z = t.z > m.z ? t.z : m.z;
      m.w = t.w > m.w ? t.w : m.w;
    }
    auto fix = [](unsigned mb) {
      const int ex = (int)(mb >> 23);
      return ex >= 255 ? 1 << 20 : (ex < 1 ? 1 : ex);  // a channel holding a NaN / infinity: every addend shifts out
    };
    ex0 = fix(m.x), ex1 = fix(m.y), ex2 = fix(m.z), ex3 = fix(m.w);
  }
  auto unfix = [](unsigned long long a, int ex) {  // the sum back in fp32 (one rounding); NaN for a poisoned channel
    const double iv = __longlong_as_double((long long)(1023 - 166 + (ex > 255 ? 1 : ex)) << 52);
    return (float)((double)(long long)a * iv) + (ex > 255 ? __builtin_nanf("") : 0.0f);
  };
  auto add = [&](int j, int k, float w, int exm) {
    const unsigned bits = __float_as_uint(w);
    const int e8 = (int)((bits >> 23) & 0xffu);
    const unsigned mant = (bits & 0x7fffffu) | (e8 ? 0x800000u : 0u);
    const int sh = (e8 ? e8 : 1) + 16 - exm;  // W = mant * 2^(e8 - 150);  <= 16
    const int down = -sh < 31 ? -sh : 31;
    const unsigned long long mag = sh >= 0 ? (unsigned long long)mant << sh : (unsigned long long)(mant >> down);
    const long long x = (bits >> 31) ? -(long long)mag : (long long)mag;
    __hip_atomic_fetch_add(&acc[j * kBS + 4 * cq + k], (unsigned long long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (threadIdx.x + u * AT < N * kCQ) {
      add((int)(tjp[u][0] & 0xffffu), 0, tw[u].x, ex0);
      add((int)(tjp[u][0] >> 16), 1, tw[u].y, ex1);
      add((int)(tjp[u][1] & 0xffffu), 2, tw[u].z, ex2);
      add((int)(tjp[u][1] >> 16), 3, tw[u].w, ex3);
    }
  }
  put_dv(0, tw[0], tv0, ts0), put_dv(1, tw[1], tv1, ts1), put_dv(2, tw[2], tv2, ts2), put_dv(3, tw[3], tv3, ts3);
  lds_barrier();
  const unsigned long long tkb = AGG_TICK();
  auto read_sd = [&](int u) {
    const int e = threadIdx.x + u * AT, ec = e < N * kCQ ? e : N * kCQ - 1;
    const ulonglong2 a0 = *reinterpret_cast<const ulonglong2*>(&acc[ec * 4]);
    const ulonglong2 a1 = *reinterpret_cast<const ulonglong2*>(&acc[ec * 4 + 2]);
    return make_float4(unfix(a0.x, ex0), unfix(a0.y, ex1), unfix(a1.x, ex2), unfix(a1.y, ex3));
  };
  const float4 sd0 = read_sd(0), sd1 = read_sd(1), sd2 = read_sd(2), sd3 = read_sd(3);
  lds_barrier();  // every accumulator has been read: the panels of phase B take their place
  auto stash = [&](int u, const float4 sdv, const float4 tv) {
    const int e = threadIdx.x + u * AT;
    if (e < N * kCQ) {
      *reinterpret_cast<float4*>(&Sd[4 * e]) = sdv;
      *reinterpret_cast<float4*>(&Vp[4 * e]) = tv;
    }
  };
  stash(0, sd0, tv0), stash(1, sd1, tv1), stash(2, sd2, tv2), stash(3, sd3, tv3);
  if (threadIdx.x < kBS) Vp[N * kBS + threadIdx.x] = 0.0f;  // the neutral row (padding in-edges)
  if (threadIdx.x < N) {
    rps[threadIdx.x] = (unsigned short)rp_a;
    ordl[threadIdx.x] = (unsigned short)or_a;
  }
  if (threadIdx.x == 0) {
    rps[N] = (unsigned short)rp_b;
    pass_ctr = AT / 64;
  }
  lds_barrier();
  const unsigned long long tk1 = AGG_TICK();
  // ---- phase B
  unsigned short* sc = scr_all + wave * kRun;
  const unsigned short kNeutral = (unsigned short)(N * 32 + 31);
  // the in-edge lists of a wave pass's consecutive ranks are ONE contiguous run of rlist: requested a pass ahead with
  // coalesced loads, copied to the wave's scratch, read from there by every lane.  A pass = kPts consecutive RANKS of the
  // degree order (equally long lists), longest lists first.  The rank -> point table sits in LDS, so the rows of the next
  // pass's points go out in the same round as its lists (through a global table they waited a round trip for it).
  auto request = [&](int r0) {  // r0: first rank of the pass (wave-uniform)
    const int base = rps[r0 < N ? r0 : N];
#pragma unroll
    for (int u = 0; u < kRun / 64; ++u) pre[u] = rl[base + 64 * u + lane < last ? base + 64 * u + lane : last];
    jn = ordl[r0 + q < N ? r0 + q : N - 1];
    request_rows();
  };
  auto next_pass = [&]() {
    int t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(&pass_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return kPts * __builtin_amdgcn_readfirstlane(t);
  };
  int j0 = r_first;
  while (j0 < N) {
    const int j0n = next_pass();
    const int rk = j0 + q, j = jn;
    const int base = rps[j0];
    const int b = rk < N ? rps[rk] : base, e = rk < N ? rps[rk + 1] : base;
#pragma unroll
    for (int u = 0; u < kRun / 64; ++u) sc[64 * u + lane] = pre[u];
    const float4 u4 = un;
    request(j0n);
    __builtin_amdgcn_wave_barrier();  // the scratch is private to the wave; LDS keeps a wave's accesses in order
    int kmax = e - b;
#pragma unroll
    for (int o = kCQ; o < 64; o <<= 1) {
      const int t = __shfl_xor(kmax, o, 64);
      kmax = t > kmax ? t : kmax;
    }
    kmax = __builtin_amdgcn_readfirstlane(kmax);
    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool hub = __any(e - base > kRun);  // a point whose in-edges reach past the staged run (rare)
    // Two copies of the loop, chosen per pass: only the hub one loads entries from global memory (with the load in the
    // one loop, every iteration waited for the NEXT pass's prefetch issued just above).
    auto scan = [&](auto with_hub) {
      for (int k = 0; k < kmax; k += MPA_AGG_BWD_UNROLL) {  // ascending sources: fixed summation order
        unsigned ent[MPA_AGG_BWD_UNROLL];
#pragma unroll
        for (int u = 0; u < MPA_AGG_BWD_UNROLL; ++u) {
          const int off = b + k + u - base;
          ent[u] = sc[off < kRun ? off : kRun - 1];
        }
        if constexpr (decltype(with_hub)::value) {
#pragma unroll
          for (int u = 0; u < MPA_AGG_BWD_UNROLL; ++u) {
            const int a = b + k + u;
            if (a < e && a - base >= kRun) ent[u] = rl[a];
          }
        }
        float4 t4[MPA_AGG_BWD_UNROLL];
#pragma unroll
        for (int u = 0; u < MPA_AGG_BWD_UNROLL; ++u) {
          const unsigned en = b + k + u < e ? ent[u] : (unsigned)kNeutral;
          t4[u] = *reinterpret_cast<const float4*>(&Vp[(en >> 5) * kBS + 4 * cq]);
        }
#pragma unroll
        for (int u = 0; u < MPA_AGG_BWD_UNROLL; ++u) {
          sv.x += t4[u].x;
          sv.y += t4[u].y;
          sv.z += t4[u].z;
          sv.w += t4[u].w;
        }
      }
    };
    if (__builtin_expect(hub, 0)) scan(std::true_type{});
    else scan(std::false_type{});
#ifdef MPA_AGG_STATS
    ++n_pass;
    n_iter += (kmax + MPA_AGG_BWD_UNROLL - 1) / MPA_AGG_BWD_UNROLL;
#endif
    __builtin_amdgcn_wave_barrier();
    if (rk < N) {
      const float deg = (float)(e - b);
      const float4 sd = *reinterpret_cast<const float4*>(&Sd[j * kBS + 4 * cq]);
      float4 du;
      du.x = __builtin_fmaf(gammap.x, __builtin_fmaf(deg, u4.x, sv.x), deg * betap.x) + sd.x;
      du.y = __builtin_fmaf(gammap.y, __builtin_fmaf(deg, u4.y, sv.y), deg * betap.y) + sd.y;
      du.z = __builtin_fmaf(gammap.z, __builtin_fmaf(deg, u4.z, sv.z), deg * betap.z) + sd.z;
      du.w = __builtin_fmaf(gammap.w, __builtin_fmaf(deg, u4.w, sv.w), deg * betap.w) + sd.w;
      *reinterpret_cast<float4*>(gp + (long long)j * 2 * CO + c0 + 4 * cq) = du;
    }
    j0 = j0n;
  }
#ifdef MPA_AGG_STATS
  {
    __shared__ unsigned long long st_s[4];
    const unsigned long long tk2 = AGG_TICK();
    if (threadIdx.x < 4) st_s[threadIdx.x] = 0ull;
    __syncthreads();
    if (lane == 0) {
      atomicAdd(&st_s[0], tk2 - tk1);  // per wave: time in the passes
      atomicAdd(&st_s[1], n_pass);
      atomicAdd(&st_s[2], n_iter);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long* r = g_agg_rec[((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) % kAggRec];
      r[0] += 1ull, r[1] += tk1 - tk0, r[2] += st_s[0], r[3] += AGG_TICK() - tk0, r[4] += st_s[1], r[5] += st_s[2];
      r[6] += tka - tk0, r[7] += tkb - tk0;
    }
  }
#endif
}

// ---- workspace ----------------------------------------------------------------------------------------------------------------
// scratch of the shortlist kNN search (dg_knn_fast.h) for clouds of N points, R rows in total, width <= 128
struct KnnWs {
  unsigned short *xs, *surv;
  float *nl, *nu, *theta, *mu;
  unsigned char* scnt;
};

template <typename Take>
KnnWs knn_carve(Take&& take, int64_t M, int64_t N) {
  KnnWs k;
  const int64_t R = M * N;
  k.xs = reinterpret_cast<unsigned short*>(take(2 * R * 2 * 128));
  k.surv = reinterpret_cast<unsigned short*>(take(2 * R * 2 * kKfCap));
  k.nl = reinterpret_cast<float*>(take(4 * R));
  k.nu = reinterpret_cast<float*>(take(4 * R));
  k.theta = reinterpret_cast<float*>(take(4 * R));
  k.scnt = reinterpret_cast<unsigned char*>(take(2 * R));
  k.mu = reinterpret_cast<float*>(take(4 * M * 128));  // one-product form: the clouds' centres
  return k;
}

// kNN graph of n (<= M: launch bound) clouds in C = 64 / 128-d feature space: row norms, bf16 split, bound pass, collect
// pass, exact rerank of the survivors (all candidates for the rare query whose survivor list overflowed).  Bit-identical
// to the exhaustive knn_mfma_kernel of dg_knn.h (which is what this build ran before; 0.93 / 1.55 ms vs 0.55 / 0.85 ms
// at 353 x 1000; tools/probes/knn_fast.hip compares the two index for index).
template <int C, typename IdxT>
void knn_wide(const float* x, int ld, float* norm, const KnnWs& k, int64_t M, int64_t N, IdxT* idx, const int* hdr,
              hipStream_t s, bool prepared = false) {
  const int64_t R = M * N;
  constexpr int SETS = 1, WAVES = 8;
  const dim3 ggram((unsigned)((N + kKfQB - 1) / kKfQB), DG_KNN_GRID_Y(M));
  if (!(prepared && kKfProducts == 1)) {  // prepared: the producer of x left norm, xs, nl, nu (dg_apply_knn_kernel)
  // (round 5: one fused pass — norm chain + split from the same staged float4 — was built and measured 0.03-0.05 ms SLOWER
  // per C = 128 search on one box, three alternations: its hi / lo stores are 32-byte segments per row and slab, where the
  // split kernels write full lines; LABBOOK 5.2)
  hipLaunchKernelGGL(rownorm_kernel<C>, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, x, ld, norm, hdr);
  if constexpr (kKfProducts == 1) {
    hipLaunchKernelGGL(knn_centre_kernel<C>, dim3((unsigned)M), dim3(C), 0, s, x, ld, (int)N, k.mu, hdr);
    hipLaunchKernelGGL(knn_split1_kernel<C>, dim3((unsigned)((R * (C / 4) + 255) / 256)), dim3(256), 0, s, x, ld,
                       (const float*)norm, (const float*)k.mu, (int)N, k.xs, k.nl, k.nu, hdr);
  } else {
    hipLaunchKernelGGL(knn_split_kernel<C>, dim3((unsigned)((R * (C / 4) + 255) / 256)), dim3(256), 0, s, x, ld,
                       (const float*)norm, k.xs, k.nl, k.nu, hdr);
  }
  }
#ifndef KF_BOUND_MODE  // timing probes only (tools/build_variant.sh): see knn_gram_kernel's MODE
#define KF_BOUND_MODE 0
#endif
  hipLaunchKernelGGL((knn_gram_kernel<C, false, SETS, WAVES, KF_BOUND_MODE>), ggram, dim3(64 * WAVES), 0, s, (const unsigned short*)k.xs,
                     (const float*)k.nl, (const float*)k.nl, (const float*)k.nu, (int)N, k.theta, k.surv, k.scnt, hdr);
  hipLaunchKernelGGL((knn_gram_kernel<C, true, SETS, WAVES>), ggram, dim3(64 * WAVES), 0, s, (const unsigned short*)k.xs,
                     (const float*)k.nu, (const float*)k.nl, (const float*)k.nu, (int)N, k.theta, k.surv, k.scnt, hdr);
  hipLaunchKernelGGL((knn_rerank_kernel<C, IdxT>), dim3((unsigned)((N + kRrQ - 1) / kRrQ), DG_KNN_GRID_Y(M)), dim3(256), 0,
                     s, x, ld, (const float*)norm, (int)N, (const unsigned short*)k.surv, (const unsigned char*)k.scnt, idx,
                     hdr, (const float*)k.theta, (const float*)k.nu);
#ifdef MPA_KNN_STATS  // instrumentation build (tools/build_variant.sh ... -DMPA_KNN_STATS=1): survivor statistics of this search
  {
    hipStreamSynchronize(s);
    std::vector<unsigned char> h((size_t)(2 * R));
    int hh[2] = {0, 0};
    hipMemcpy(hh, hdr, 8, hipMemcpyDeviceToHost);
    hipMemcpy(h.data(), k.scnt, h.size(), hipMemcpyDeviceToHost);
    long long rows = hh[1], over = 0, total = 0, hist[8] = {0};
    int mx = 0;
    for (long long r = 0; r < rows; ++r) {
      const int a = h[2 * r], b = h[2 * r + 1];
      if (a == kKfOverflow || b == kKfOverflow) { ++over; continue; }
      total += a + b;
      mx = a + b > mx ? a + b : mx;
      ++hist[(a > b ? a : b) / 4 < 7 ? (a > b ? a : b) / 4 : 7];
    }
    fprintf(stderr, "[knn stats] C=%d rows=%lld overflow queries=%lld mean survivors=%.2f max=%d | larger half-list histogram (bins of 4):",
            C, rows, over, (double)total / (double)(rows - over > 0 ? rows - over : 1), mx);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %lld", hist[i]);
    fprintf(stderr, "\n");
  }
#endif
}

struct Ws {
  int *hdr, *vlist, *rank, *arg5, *rptr[4], *order[4];
  unsigned* tickets;
  float4* x0;
  float *hcat, *uv[4], *esel[4], *s1[4], *y5, *norm, *bn[5], *coef, *partial, *wstk[4], *wstt[4], *w5t, *pooled,
      *dpooled, *tnpart, *dhcat, *duv, *dz, *gstk;
  unsigned short *idx[4], *rlist[4];
  unsigned char* ssel[4];
  double* stage;
  KnnWs knn;
  int64_t total;
};

Ws dg_carve(char* base, int64_t M, int64_t N, int64_t F) {
  Ws w;
  char* p = base;
  auto take = [&](int64_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  const int64_t R = M * N, tiles = (R + kTile - 1) / kTile;
  w.hdr = reinterpret_cast<int*>(take(64));
  w.vlist = reinterpret_cast<int*>(take(4 * M));
  w.rank = reinterpret_cast<int*>(take(4 * M));
  w.tickets = reinterpret_cast<unsigned*>(take(64));
  w.x0 = reinterpret_cast<float4*>(take(16 * R));
  w.hcat = reinterpret_cast<float*>(take(4 * R * kCat));
  for (int l = 0; l < 4; ++l) {
    w.uv[l] = reinterpret_cast<float*>(take(4 * R * 2 * kCO[l]));
    w.esel[l] = reinterpret_cast<float*>(take(4 * R * kCO[l]));
    w.s1[l] = reinterpret_cast<float*>(take(4 * R * kCO[l]));
    w.ssel[l] = reinterpret_cast<unsigned char*>(take(R * kCO[l]));
    w.idx[l] = reinterpret_cast<unsigned short*>(take(2 * R * kNbr));
    w.bn[l] = reinterpret_cast<float*>(take(4 * 4 * kCO[l]));
    w.wstk[l] = reinterpret_cast<float*>(take(4 * 2 * kCO[l] * kCinP[l]));
    w.wstt[l] = reinterpret_cast<float*>(take(4 * 2 * kCO[l] * kCinP[l]));
  }
  w.y5 = reinterpret_cast<float*>(take(4 * R * F));
  w.norm = reinterpret_cast<float*>(take(4 * R));
  w.bn[4] = reinterpret_cast<float*>(take(4 * 4 * F));
  w.coef = reinterpret_cast<float*>(take(4 * 3 * kCat));
  const int64_t prow = tiles > M ? tiles : M;  // partial table rows: row tiles or parts
  w.partial = reinterpret_cast<float*>(take(4 * prow * kCat * 2));
  w.w5t = reinterpret_cast<float*>(take(4 * kCat * F));
  w.pooled = reinterpret_cast<float*>(take(4 * M * 2 * F));
  w.dpooled = reinterpret_cast<float*>(take(4 * M * 2 * F));
  w.arg5 = reinterpret_cast<int*>(take(4 * M * F));
  int64_t tn = (int64_t)kTnChunks * kCat * 128;  // largest weight gradient: 512 x 128 (or F x 512)
  if ((int64_t)kTnChunks * F * kCat > tn) tn = (int64_t)kTnChunks * F * kCat;
  const int64_t first = ((R + kFirstTile - 1) / kFirstTile) * 128 * 4;  // first-stage weight-gradient partials
  if (first > tn) tn = first;
  w.tnpart = reinterpret_cast<float*>(take(4 * tn));
  w.dhcat = reinterpret_cast<float*>(take(4 * R * kCat));
  w.duv = reinterpret_cast<float*>(take(4 * R * 2 * kCO[3]));
  w.dz = reinterpret_cast<float*>(take(4 * R * kCO[3]));
  w.gstk = reinterpret_cast<float*>(take(4 * kCat * 128));
  for (int l = 0; l < 4; ++l) {
    w.rptr[l] = reinterpret_cast<int*>(take(4 * M * (N + 1)));
    w.order[l] = reinterpret_cast<int*>(take(4 * M * N));
  }
  for (int l = 0; l < 4; ++l) w.rlist[l] = reinterpret_cast<unsigned short*>(take(2 * R * kNbr));
  w.stage = reinterpret_cast<double*>(take(8 * 2 * kCat * ((prow + kEB - 1) / kEB)));
  w.knn = knn_carve(take, M, N);
  w.total = p - base;
  return w;
}

int dg_check(int64_t M, int64_t N, int64_t F, const char* who) {
  MPA_REQUIRE(M >= 0 && M <= 65528, "%s: 0 <= parts <= 65528", who);
  MPA_REQUIRE(N >= kNbr && N <= kMaxN, "%s: %d <= points per part <= %d", who, kNbr, kMaxN);
  MPA_REQUIRE(F == 64 || F == 128 || F == 256, "%s: feat_dim must be 64, 128 or 256", who);
  return MPA_OK;
}

template <typename Kern, typename... Args>
void launch(Kern kern, dim3 grid, dim3 block, hipStream_t s, Args... args) {
  hipLaunchKernelGGL(kern, grid, block, 0, s, args...);
}

inline void launch_agg_bwd(dim3 grid, hipStream_t s, const float* uv, int CO, const int* rptr, const int* order,
                    const unsigned short* rlist, const unsigned short* idx, const float* dz, const unsigned char* ssel,
                    const float* s1, const float* coef, int M, int N, float* guv, const int* hdr) {
  static_assert(agg_bwd_lds(kMaxN) + 2048 <= 160 * 1024, "dg_agg_bwd_kernel: the panels of the largest part must fit the LDS");
  static bool reserved = false;  // the opt-in to more than 64 KB of dynamic LDS is per kernel, once
  if (!reserved) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(dg_agg_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)agg_bwd_lds(kMaxN));
    reserved = true;
  }
  hipLaunchKernelGGL(dg_agg_bwd_kernel, grid, dim3(kAggT), agg_bwd_lds(N), s, uv, CO, rptr, order, rlist, idx, dz, ssel, s1,
                     coef, M, N, guv, hdr);
}

// C (+)= A . W^T on the matrix cores (see dg_gemm.h); Nout a multiple of 64
void gemm_nt(const float* A, int lda, const float* W, int K, float* C, int ldc, int Nout, bool accum, int64_t Rmax,
             const int* hdr, hipStream_t s) {
  const unsigned gx = DG_GEMM_GRID_X(Rmax);
  if (Nout % 128 == 0) {
    if (accum) launch(DG_NT_KERNEL<128, true>, dim3(gx, Nout / 128), dim3(DG_GEMM_THREADS), s, A, lda, W, K, C, ldc, hdr DG_NT_TAIL);
    else launch(DG_NT_KERNEL<128, false>, dim3(gx, Nout / 128), dim3(DG_GEMM_THREADS), s, A, lda, W, K, C, ldc, hdr DG_NT_TAIL);
  } else {
    if (accum) launch(DG_NT_KERNEL<64, true>, dim3(gx, Nout / 64), dim3(DG_GEMM_THREADS), s, A, lda, W, K, C, ldc, hdr DG_NT_TAIL);
    else launch(DG_NT_KERNEL<64, false>, dim3(gx, Nout / 64), dim3(DG_GEMM_THREADS), s, A, lda, W, K, C, ldc, hdr DG_NT_TAIL);
  }
}

// out [Nout][K] = Y^T . X over the valid rows (two deterministic stages); K a multiple of 64.  The rows are cut into
// as many chunks as it takes to put 512 blocks on the chip (two per CU): a 128 x 64 gradient has ONE output tile.
void gemm_tn(const float* Y, int ldy, int Nout, const float* X, int ldx, int K, float* part, float* out, int64_t Rmax,
             const int* hdr, hipStream_t s) {
  const unsigned tn = (unsigned)((Nout + 127) / 128), tk = (unsigned)(K % 128 == 0 ? K / 128 : K / 64);
  const int chunks = tn * tk >= 4 ? kTnChunks : (int)(4 * kTnChunks / (tn * tk));
  const int rows_per_chunk = 0;  // cut the VALID rows (hdr[1], at most Rmax) evenly: every chunk has work
  const dim3 grid(tn, tk, (unsigned)chunks);
  if (K % 128 == 0) launch(DG_TN_KERNEL<128>, grid, dim3(DG_GEMM_THREADS), s, Y, ldy, Nout, X, ldx, K, part, rows_per_chunk, hdr, 0);
  else launch(DG_TN_KERNEL<64>, grid, dim3(DG_GEMM_THREADS), s, Y, ldy, Nout, X, ldx, K, part, rows_per_chunk, hdr, 0);
  const long long elems = (long long)Nout * K;
  launch_tn_reduce(part, chunks, elems, out, s);
}

void record(void* const* events, int i, hipStream_t s) {
  if (events != nullptr && events[i] != nullptr) hipEventRecord(static_cast<hipEvent_t>(events[i]), s);
}

}  // namespace

extern "C" int mpa_dgcnn_workspace(int64_t M, int64_t N, int64_t F, int64_t* bytes) {
  if (int st = dg_check(M, N, F, "dgcnn_workspace")) return st;
  MPA_REQUIRE(bytes != nullptr, "dgcnn_workspace: null pointer");
  *bytes = dg_carve(nullptr, M, N, F).total;
  return MPA_OK;
}

namespace {
// caller-supplied graph of the COMPACTED valid parts -> the workspace's u16 lists (rows past the valid parts untouched)
__global__ void dg_import_graph_kernel(const int32_t* __restrict__ src, unsigned short* __restrict__ dst,
                                       const int* __restrict__ hdr) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long long)hdr[1] * kNbr) dst[i] = (unsigned short)src[i];
}
__global__ void dg_export_graph_kernel(const unsigned short* __restrict__ src, int32_t* __restrict__ dst,
                                       const int* __restrict__ hdr, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) dst[i] = i < (long long)hdr[1] * kNbr ? (int32_t)src[i] : -1;
}

int dgcnn_forward_impl(const float* points, const float* valids, const float* const* conv_w,
                       const float* const* bn_w, const float* const* bn_b, float* const* running_mean,
                       float* const* running_var, const float* fc_w, const float* fc_b, int training,
                       float momentum, float eps, int64_t M, int64_t N, int64_t F, void* ws, float* feat,
                       void* const* events, const int32_t* const* graphs, void* stream) {
  if (int st = dg_check(M, N, F, "dgcnn_forward")) return st;
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(points && valids && conv_w && bn_w && bn_b && running_mean && running_var && fc_w && fc_b && ws && feat,
              "dgcnn_forward: null pointer");
  MPA_REQUIRE((uintptr_t)ws % 256 == 0, "dgcnn_forward: workspace must be 256-byte aligned");
  hipStream_t s = mpa::as_stream(stream);
  const Ws w = dg_carve(static_cast<char*>(ws), M, N, F);
  const int64_t R = M * N, tiles = (R + kTile - 1) / kTile;
  const CoopWs cw{w.stage, w.tickets};
  launch(dg_prepare_kernel, dim3(1), dim3(64), s, valids, (int)M, (int)N, w.hdr, w.vlist, w.rank, w.tickets, 16);
  launch(dg_gather_points_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), s, points, (const int*)w.vlist, (int)N,
         w.x0, (const int*)w.hdr);
  for (int l = 0; l < 4; ++l) {
    const int CO = kCO[l], C = kCin[l], CP = kCinP[l];
    launch(dg_wstack_kernel, dim3((unsigned)((2 * CO * CP + 255) / 256)), dim3(256), s, conv_w[l], CO, C, CP, w.wstk[l],
           w.wstt[l]);
    // kNN graph in the stage's input space
    record(events, 2 * l, s);
    if (graphs != nullptr && graphs[l] != nullptr) {
      launch(dg_import_graph_kernel, dim3((unsigned)((R * kNbr + 255) / 256)), dim3(256), s, graphs[l], w.idx[l],
             (const int*)w.hdr);
    } else if (l == 0) {
      if (knn3_gate())
        launch(knn3_gate_kernel<unsigned short>, dim3((unsigned)((N + 255) / 256), DG_KNN_GRID_Y(M)), dim3(256), s,
               reinterpret_cast<const float*>(w.x0), (int)N, w.idx[0], (const int*)w.hdr);
      else
        launch(knn3_kernel<unsigned short>, dim3((unsigned)((N + DG_T3 - 1) / DG_T3), DG_KNN_GRID_Y(M)), dim3(DG_T3), s,
               reinterpret_cast<const float*>(w.x0), (int)N, w.idx[0], (const int*)w.hdr);
    } else {
      const float* x = w.hcat + kOff[l - 1];
      // (the operands were left by stage l - 1's apply pass below unless MPA_KNN_PRODUCER=0)
      if (C == 64) knn_wide<64, unsigned short>(x, kCat, w.norm, w.knn, M, N, w.idx[l], (const int*)w.hdr, s, knn_producer());
      else knn_wide<128, unsigned short>(x, kCat, w.norm, w.knn, M, N, w.idx[l], (const int*)w.hdr, s, knn_producer());
    }
    record(events, 2 * l + 1, s);
    // [U | V] = X . [Wa ; Wb - Wa]^T
    if (l == 0) {
      launch(dg_first_uv_kernel, dim3((unsigned)((R + 7) / 8)), dim3(256), s, (const float4*)w.x0,
             (const float*)w.wstk[0], w.uv[0], (const int*)w.hdr);
    } else {
      gemm_nt(w.hcat + kOff[l - 1], kCat, w.wstk[l], C, w.uv[l], 2 * CO, 2 * CO, false, R, w.hdr, s);
    }
    {
      static bool reserved = false;  // opt-in to more than 64 KB of dynamic LDS, once
      if (!reserved) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(dg_agg_fwd_kernel<MPA_AGG_FWD_AT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)agg_fwd_lds(MPA_AGG_FWD_AT, kMaxN));
        reserved = true;
      }
      hipLaunchKernelGGL(dg_agg_fwd_kernel<MPA_AGG_FWD_AT>, dim3((unsigned)(CO / 32), DG_KNN_GRID_Y(M)), dim3(MPA_AGG_FWD_AT),
                         agg_fwd_lds(MPA_AGG_FWD_AT, (int)N), s, (const float*)w.uv[l], CO,
                         (const unsigned short*)w.idx[l], bn_w[l], (int)N, w.esel[l], w.ssel[l], w.s1[l], w.partial,
                         (const int*)w.hdr);
    }
    if (training) {
      launch(dg_bn_finalize_kernel, dim3((unsigned)(CO / 64), (unsigned)((M + kEB - 1) / kEB)), dim3(64 * kSlices), s,
             (const float*)w.partial, (int)M, CO, 1, kNbr, bn_w[l], bn_b[l], running_mean[l], running_var[l], momentum,
             eps, w.bn[l], cw, (const int*)w.hdr);
    } else {
      launch(dg_bn_from_running_kernel, dim3((unsigned)(CO / 64)), dim3(64), s, CO, bn_w[l], bn_b[l],
             (const float*)running_mean[l], (const float*)running_var[l], eps, w.bn[l]);
    }
    // BatchNorm + LeakyReLU into the concatenation; stages 1-3 also leave the next stage's kNN operands
    const bool feeds_knn = l < 3 && kKfProducts == 1 && knn_producer() && !(graphs != nullptr && graphs[l + 1] != nullptr);
    if (feeds_knn && CO == 64) {
      launch(dg_apply_centre_kernel<64>, dim3((unsigned)M), dim3(64), s, (const float*)w.esel[l], (const float*)w.bn[l],
             (int)N, w.knn.mu, (const int*)w.hdr);
      launch(dg_apply_knn_kernel<64>, dim3((unsigned)((R + 63) / 64)), dim3(256), s, (const float*)w.esel[l],
             (const float*)w.bn[l], w.hcat, kOff[l], (const float*)w.knn.mu, (int)N, w.norm, w.knn.xs, w.knn.nl, w.knn.nu,
             (const int*)w.hdr);
    } else if (feeds_knn && CO == 128) {
      launch(dg_apply_centre_kernel<128>, dim3((unsigned)M), dim3(128), s, (const float*)w.esel[l], (const float*)w.bn[l],
             (int)N, w.knn.mu, (const int*)w.hdr);
      launch(dg_apply_knn_kernel<128>, dim3((unsigned)((R + 31) / 32)), dim3(256), s, (const float*)w.esel[l],
             (const float*)w.bn[l], w.hcat, kOff[l], (const float*)w.knn.mu, (int)N, w.norm, w.knn.xs, w.knn.nl, w.knn.nu,
             (const int*)w.hdr);
    } else {
      launch(dg_apply_kernel, dim3((unsigned)((R * (CO / 4) + 255) / 256)), dim3(256), s, (const float*)w.esel[l], CO,
             (const float*)w.bn[l], w.hcat, kOff[l], (const int*)w.hdr);
    }
  }
  // tail: 512 -> F convolution, BatchNorm1d, LeakyReLU, [max ; mean] over the points, Linear
  gemm_nt(w.hcat, kCat, conv_w[4], kCat, w.y5, (int)F, (int)F, false, R, w.hdr, s);
  if (training) {
    launch(dg_colstats_kernel, dim3((unsigned)tiles), dim3(kTileT), s, (const float*)w.y5, (int)F, w.partial,
           (const int*)w.hdr);
    launch(dg_bn_finalize_kernel, dim3((unsigned)(F / 64), (unsigned)((tiles + kEB - 1) / kEB)), dim3(64 * kSlices), s,
           (const float*)w.partial, (int)tiles, (int)F, 0, 1, bn_w[4], bn_b[4], running_mean[4], running_var[4], momentum,
           eps, w.bn[4], cw, (const int*)w.hdr);
  } else {
    launch(dg_bn_from_running_kernel, dim3((unsigned)(F / 64)), dim3(64), s, (int)F, bn_w[4], bn_b[4],
           (const float*)running_mean[4], (const float*)running_var[4], eps, w.bn[4]);
  }
  launch(dg_pool_kernel, dim3((unsigned)M), dim3(1024), s, (const float*)w.y5, (int)F, (int)N, (const float*)w.bn[4],
         w.pooled, w.arg5, (const int*)w.hdr);
  launch(dg_fc_kernel, dim3((unsigned)M), dim3((unsigned)F), s, (const float*)w.pooled, (const int*)w.rank, (int)F, fc_w,
         fc_b, feat);
  return mpa::check_launch("dgcnn_forward");
}
}  // namespace

extern "C" int mpa_dgcnn_forward(const float* points, const float* valids, const float* const* conv_w,
                                 const float* const* bn_w, const float* const* bn_b, float* const* running_mean,
                                 float* const* running_var, const float* fc_w, const float* fc_b, int training,
                                 float momentum, float eps, int64_t M, int64_t N, int64_t F, void* ws, float* feat,
                                 void* const* events, void* stream) {
  return dgcnn_forward_impl(points, valids, conv_w, bn_w, bn_b, running_mean, running_var, fc_w, fc_b, training, momentum,
                            eps, M, N, F, ws, feat, events, nullptr, stream);
}

extern "C" int mpa_dgcnn_forward_graphs(const float* points, const float* valids, const float* const* conv_w,
                                        const float* const* bn_w, const float* const* bn_b, float* const* running_mean,
                                        float* const* running_var, const float* fc_w, const float* fc_b, int training,
                                        float momentum, float eps, int64_t M, int64_t N, int64_t F, void* ws,
                                        float* feat, const int32_t* const* graphs, void* stream) {
  MPA_REQUIRE(graphs != nullptr, "dgcnn_forward_graphs: null pointer");
  return dgcnn_forward_impl(points, valids, conv_w, bn_w, bn_b, running_mean, running_var, fc_w, fc_b, training, momentum,
                            eps, M, N, F, ws, feat, nullptr, graphs, stream);
}

extern "C" int mpa_dgcnn_export_graph(const void* ws, int64_t M, int64_t N, int64_t F, int64_t stage, int32_t* idx,
                                      void* stream) {
  if (int st = dg_check(M, N, F, "dgcnn_export_graph")) return st;
  MPA_REQUIRE(stage >= 0 && stage < 4, "dgcnn_export_graph: stage must be 0..3");
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(ws && idx, "dgcnn_export_graph: null pointer");
  const Ws w = dg_carve(static_cast<char*>(const_cast<void*>(ws)), M, N, F);
  const long long total = (long long)M * N * kNbr;
  launch(dg_export_graph_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), mpa::as_stream(stream),
         (const unsigned short*)w.idx[stage], idx, (const int*)w.hdr, total);
  return mpa::check_launch("dgcnn_export_graph");
}

extern "C" int mpa_dgcnn_backward(const float* grad_feat, const float* const* conv_w, const float* const* bn_w,
                                  const float* fc_w, int64_t M, int64_t N, int64_t F, void* ws,
                                  float* const* grad_conv_w, float* const* grad_bn_w, float* const* grad_bn_b,
                                  float* grad_fc_w, float* grad_fc_b, float* grad_points, void* stream) {
  if (int st = dg_check(M, N, F, "dgcnn_backward")) return st;
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(grad_feat && conv_w && bn_w && fc_w && ws && grad_conv_w && grad_bn_w && grad_bn_b && grad_fc_w &&
                  grad_fc_b,
              "dgcnn_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const Ws w = dg_carve(static_cast<char*>(ws), M, N, F);
  const int64_t R = M * N, tiles = (R + kTile - 1) / kTile;
  const CoopWs cw{w.stage, w.tickets};
  const int* hdr = w.hdr;
  // Linear and pooling
  launch(dg_fc_bwd_in_kernel, dim3((unsigned)M), dim3((unsigned)(2 * F)), s, grad_feat, (const int*)w.vlist, (int)F,
         fc_w, w.dpooled, hdr);
  launch(dg_fc_bwd_w_kernel, dim3((unsigned)F), dim3(1024), s, grad_feat, (const int*)w.vlist, (int)F,
         (const float*)w.pooled, grad_fc_w, grad_fc_b, hdr);
  // BatchNorm1d backward of the tail; y5 becomes dY5
  launch(dg_tail_bwd_sums_kernel, dim3((unsigned)tiles), dim3(kTileT), s, (const float*)w.y5, (int)F, (int)N,
         (const float*)w.bn[4], (const float*)w.dpooled, (const int*)w.arg5, w.partial, hdr);
  launch(dg_bwd_coef_kernel, dim3((unsigned)(F / 64), (unsigned)((tiles + kEB - 1) / kEB)), dim3(64 * kSlices), s,
         (const float*)w.partial, (int)tiles, (int)F, 1, bn_w[4], (const float*)w.bn[4], w.coef, grad_bn_w[4],
         grad_bn_b[4], cw, hdr);
  launch(dg_tail_bwd_apply_kernel, dim3((unsigned)tiles), dim3(kTileT), s, w.y5, (int)F, (int)N,
         (const float*)w.bn[4], (const float*)w.coef, (const float*)w.dpooled, (const int*)w.arg5, hdr);
  gemm_tn(w.y5, (int)F, (int)F, w.hcat, kCat, kCat, w.tnpart, grad_conv_w[4], R, hdr, s);
  launch(dg_transpose_kernel, dim3((unsigned)((F * kCat + 255) / 256)), dim3(256), s, conv_w[4], (int)F, kCat, w.w5t);
  gemm_nt(w.y5, (int)F, w.w5t, (int)F, w.dhcat, kCat, kCat, false, R, hdr, s);
  if (grad_points != nullptr) mpa::zero_words_async(grad_points, M * N * 3, s);
  for (int l = 3; l >= 0; --l) {
    const int CO = kCO[l], C = kCin[l], CP = kCinP[l];
    if (l == 3) {  // + the transposed graphs of all four stages (dg_bwd_head_kernel)
      DgRevArgs ra;
      for (int q = 0; q < 4; ++q) {
        ra.idx[q] = w.idx[q];
        ra.rptr[q] = w.rptr[q];
        ra.order[q] = w.order[q];
        ra.rlist[q] = w.rlist[q];
      }
      launch(dg_bwd_head_kernel, dim3((unsigned)(4 * M + tiles)), dim3(1024), s, ra, (int)M, (int)N, (const float*)w.hcat,
             (const float*)w.dhcat, kOff[l], CO, (const float*)w.esel[l], (const float*)w.bn[l], w.dz, w.partial, hdr);
    } else {
      launch(dg_agg_bwd_sums_kernel, dim3((unsigned)tiles), dim3(512), s, (const float*)w.hcat,
             (const float*)w.dhcat, kOff[l], CO, (const float*)w.esel[l], (const float*)w.bn[l], w.dz, w.partial, hdr);
    }
    launch(dg_bwd_coef_kernel, dim3((unsigned)(CO / 64), (unsigned)((tiles + kEB - 1) / kEB)), dim3(64 * kSlices), s,
           (const float*)w.partial, (int)tiles, CO, kNbr, bn_w[l], (const float*)w.bn[l], w.coef, grad_bn_w[l],
           grad_bn_b[l], cw, hdr);
    launch_agg_bwd(dim3((unsigned)(CO / kBS), DG_KNN_GRID_Y(M)), s, (const float*)w.uv[l], CO, (const int*)w.rptr[l],
                   (const int*)w.order[l], (const unsigned short*)w.rlist[l], (const unsigned short*)w.idx[l],
                   (const float*)w.dz,
                   (const unsigned char*)w.ssel[l], (const float*)w.s1[l], (const float*)w.coef, (int)M, (int)N, w.duv, hdr);
    if (l == 0) {
      const int t1 = (int)((R + kFirstTile - 1) / kFirstTile);
      launch(dg_first_wgrad_kernel, dim3((unsigned)t1), dim3(512), s, (const float*)w.duv, (const float4*)w.x0, w.tnpart,
             hdr);
      launch_tn_reduce(w.tnpart, t1, (long long)(128 * 4), w.gstk, s);
      if (grad_points != nullptr)
        launch(dg_first_dgrad_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), s, (const float*)w.duv,
               (const float*)w.wstk[0], (const int*)w.vlist, (int)N, grad_points, hdr);
    } else {
      gemm_tn(w.duv, 2 * CO, 2 * CO, w.hcat + kOff[l - 1], kCat, C, w.tnpart, w.gstk, R, hdr, s);
      gemm_nt(w.duv, 2 * CO, w.wstt[l], 2 * CO, w.dhcat + kOff[l - 1], kCat, C, true, R, hdr, s);
    }
    launch(dg_wunstack_kernel, dim3((unsigned)((CO * 2 * C + 255) / 256)), dim3(256), s, (const float*)w.gstk, CO, C, CP,
           grad_conv_w[l]);
  }
  return mpa::check_launch("dgcnn_backward");
}

namespace {
__global__ void dg_set_hdr_kernel(int* hdr, int n, int N) {
  hdr[0] = n;
  hdr[1] = n * N;
}
}  // namespace

namespace {
struct KnnExactWs {
  int* hdr;
  float* norm;
  KnnWs knn;
  int64_t total;
};
KnnExactWs knn_exact_carve(char* base, int64_t n, int64_t N) {
  KnnExactWs w;
  char* p = base;
  auto take = [&](int64_t bytes) {
    char* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  w.hdr = reinterpret_cast<int*>(take(64));
  w.norm = reinterpret_cast<float*>(take(4 * n * N));
  w.knn = knn_carve(take, n, N);
  w.total = p - base;
  return w;
}
}  // namespace

extern "C" int mpa_knn_exact_workspace(int64_t n, int64_t N, int64_t* bytes) {
  MPA_REQUIRE(n >= 0 && n <= 65528 && N >= kNbr && N <= kMaxN, "knn_exact_workspace: %d <= N <= %d, n <= 65528", kNbr, kMaxN);
  MPA_REQUIRE(bytes != nullptr, "knn_exact_workspace: null pointer");
  *bytes = knn_exact_carve(nullptr, n, N).total;
  return MPA_OK;
}

extern "C" int mpa_knn_exact(const float* x, int64_t ld, int64_t n, int64_t N, int64_t C, void* ws, int32_t* idx,
                             void* stream) {
  MPA_REQUIRE(n >= 0 && n <= 65528 && N >= kNbr && N <= kMaxN, "knn_exact: %d <= N <= %d, n <= 65528", kNbr, kMaxN);
  MPA_REQUIRE(C == 3 || C == 64 || C == 128, "knn_exact: feature width must be 3, 64 or 128");
  MPA_REQUIRE(ld % 4 == 0 && ld >= (C == 3 ? 4 : C), "knn_exact: bad leading dimension");
  MPA_REQUIRE(C != 3 || ld == 4, "knn_exact: C = 3 takes [n*N, 4] rows (x, y, z, 0)");
  if (n == 0) return MPA_OK;
  MPA_REQUIRE(x && idx && ws, "knn_exact: null pointer");
  MPA_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)x % 16 == 0, "knn_exact: workspace 256-byte, x 16-byte aligned");
  hipStream_t s = mpa::as_stream(stream);
  const KnnExactWs w = knn_exact_carve(static_cast<char*>(ws), n, N);  // hdr = {n, n*N}: every cloud is valid here
  launch(dg_set_hdr_kernel, dim3(1), dim3(1), s, w.hdr, (int)n, (int)N);
  if (C == 3) {
    if (knn3_gate())
      launch(knn3_gate_kernel<int>, dim3((unsigned)((N + 255) / 256), DG_KNN_GRID_Y(n)), dim3(256), s, x, (int)N, idx,
             (const int*)w.hdr);
    else
      launch(knn3_kernel<int>, dim3((unsigned)((N + DG_T3 - 1) / DG_T3), DG_KNN_GRID_Y(n)), dim3(DG_T3), s, x, (int)N, idx,
             (const int*)w.hdr);
  } else if (C == 64) {
    knn_wide<64, int>(x, (int)ld, w.norm, w.knn, n, N, idx, (const int*)w.hdr, s);
  } else {
    knn_wide<128, int>(x, (int)ld, w.norm, w.knn, n, N, idx, (const int*)w.hdr, s);
  }
  return mpa::check_launch("knn_exact");
}

#ifdef MPA_AGG_STATS
__global__ void agg_stats_fold_kernel(int reset) {  // <<<1, 8>>>: sums the records (and clears them)
  unsigned long long t = 0;
  for (int b = 0; b < kAggRec; ++b) {
    t += g_agg_rec[b][threadIdx.x];
    if (reset) g_agg_rec[b][threadIdx.x] = 0ull;
  }
  g_agg_stats[threadIdx.x] = t;
}
extern "C" int mpa_debug_agg_stats(unsigned long long* out8, int reset) {
  hipLaunchKernelGGL(agg_stats_fold_kernel, dim3(1), dim3(8), 0, 0, reset);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_agg_stats), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  return 0;
}
#endif
