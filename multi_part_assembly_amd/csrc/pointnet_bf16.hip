// PointNet part encoder, bf16 PERFORMANCE VARIANT (gfx950) — separately named entry points, never the default path.
//
// Same network as csrc/pointnet.hip (multi_part_assembly/models/modules/encoder/pointnet.py:6-41: five 1x1 convolutions
// 3-64-64-64-128-F without bias, BatchNorm1d in training mode, ReLU except after the last, max over the N points), same
// arguments and the same masking of padded parts, but the arithmetic the reference runs under AMP (chamfer.py:14 keeps
// the LOSS in fp32; the encoder's convolutions run in half precision):
//   * every convolution output y_l is STORED in bf16 ([rows, C_l] row-major) — half the HBM traffic of the fp32 path,
//     which is what bounds every layer once the matrix cores run at the bf16 rate;
//   * the GEMMs are v_mfma_f32_32x32x16_bf16 with fp32 accumulators; the BatchNorm affine + ReLU of layer l is applied
//     in fp32 while layer l+1 loads its operand, and rounded to bf16 for the matrix core;
//   * BatchNorm statistics, their finalisation, all reductions and all parameter gradients are fp32 (sums in a fixed
//     order: deterministic), taken from the values as stored (so the normalised tensor has exactly zero mean);
//   * backward: dy_l = a_l G_l + P_l + Q_l y_l per channel (G_l: the masked upstream gradient, stored in bf16; for the last
//     layer G is the pooled gradient at the arg-max rows and never materialised), one kernel for the input gradient and
//     one for the weight gradient per layer, both forming dy_l on the fly.
// Rows of padded parts never enter: the valid parts are counted and compacted on the device (hdr = {parts, rows}).
//
// Tolerances against the fp32 path are those of bf16 (8 mantissa bits): tests/test_pointnet_bf16_gpu.py holds features to a few
// 1e-2 of their scale and gradients to a cosine similarity; parity claims are made for the fp32 path only.
#include "common.h"
#include "dg_gemm.h"

namespace {

using dg::f32x16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int kT = 256;      // threads per block
constexpr int kRows = 128;   // rows per tile
constexpr int kPad = 8;      // bf16 elements of padding per LDS row (16 bytes: rows stay 16-byte aligned)
constexpr int kChunks = 512; // row chunks of the weight-gradient kernels (two blocks per CU)
constexpr int kWS = 64;      // rows per weight-gradient step
constexpr int kWLD = kWS + kPad;

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned short pack1(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float unpack1(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ void unpack8(const uint4 u, float* f) {
  f[0] = __uint_as_float(u.x << 16);
  f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16);
  f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16);
  f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16);
  f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}
__device__ __forceinline__ void load8f(const float* p, float* f) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
  f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// ---- valid parts: hdr = {nv, nv * N}; vlist[v] = part slot of the v-th valid part; rank[m] = v or -1 -----------------
__global__ __launch_bounds__(1024) void pb_prepare_kernel(const float* __restrict__ valids, int M, int N,
                                                          int* __restrict__ hdr, int* __restrict__ vlist,
                                                          int* __restrict__ rank) {
  __shared__ int wcnt[16];
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
    const int v = base + before + __popcll(b & ((1ull << lane) - 1ull));
    if (ok) vlist[v] = m;
    if (m < M) rank[m] = ok ? v : -1;
    base += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hdr[0] = base;
    hdr[1] = base * N;
  }
}

// W [CO][CI] fp32 -> Wb [CO][CI] bf16 and Wt [CI][CO] bf16
__global__ void pb_weights_kernel(const float* __restrict__ W, int CO, int CI, unsigned short* __restrict__ Wb,
                                  unsigned short* __restrict__ Wt) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= CO * CI) return;
  const int co = e / CI, ci = e % CI;
  const unsigned short b = pack1(W[e]);
  Wb[e] = b;
  Wt[ci * CO + co] = b;
}

// column sums of a bf16 tile in LDS: tile [kRows][ld] (bf16), columns [0, CT) -> part[c] = sum, part[off2 + c] = sum of
// squares.  256 threads: column c = t % CT, 256 / CT row slices; `red` needs 2 * 256 floats.
template <int CT>
__device__ __forceinline__ void tile_col_stats(const unsigned short* tile, int ld, float* red, float* part, int off2) {
  constexpr int SL = kT / CT, RPS = kRows / SL;
  const int c = threadIdx.x % CT, sl = threadIdx.x / CT;
  float s = 0.0f, q = 0.0f;
#pragma unroll 8
  for (int r = 0; r < RPS; ++r) {
    const float v = unpack1(tile[(sl * RPS + r) * ld + c]);
    s += v;
    q += v * v;
  }
  red[threadIdx.x] = s;
  red[kT + threadIdx.x] = q;
  __syncthreads();
  if (threadIdx.x < CT) {
    float ts = 0.0f, tq = 0.0f;
#pragma unroll
    for (int k = 0; k < SL; ++k) {
      ts += red[k * CT + c];
      tq += red[kT + k * CT + c];
    }
    part[c] = ts;
    part[off2 + c] = tq;
  }
}

// ---- layer 1: y1 = W1 x (fp32 FMA chain), stored bf16; statistics of the stored values ---------------------------------
// grid = row tiles; points [M][N][3]; y1 [rows][64]; part [tiles][128].
__global__ __launch_bounds__(kT) void pb_first_fwd_kernel(const float* __restrict__ points, const int* __restrict__ vlist,
                                                          int N, const float* __restrict__ W1,
                                                          unsigned short* __restrict__ y1, float* __restrict__ part,
                                                          const int* __restrict__ hdr) {
  constexpr int LD = 64 + kPad;
  __shared__ float w[64 * 3];
  __shared__ __attribute__((aligned(16))) unsigned short tile[kRows * LD];
  __shared__ float red[2 * kT];
  const int R = hdr[1];
  const int r0 = blockIdx.x * kRows;
  if (r0 >= R) return;
  if (threadIdx.x < 192) w[threadIdx.x] = W1[threadIdx.x];
  __syncthreads();
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1, r = r0 + row;
  float x = 0.0f, y = 0.0f, z = 0.0f;
  const bool ok = r < R;
  if (ok) {
    const int v = r / N, i = r - v * N;
    const float* p = points + ((long long)vlist[v] * N + i) * 3;
    x = p[0];
    y = p[1];
    z = p[2];
  }
  float o[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    const float* wc = &w[(half * 32 + c) * 3];
    o[c] = __builtin_fmaf(wc[2], z, __builtin_fmaf(wc[1], y, wc[0] * x));
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const uint4 u = pack8(&o[8 * g]);
    *reinterpret_cast<uint4*>(&tile[row * LD + half * 32 + 8 * g]) = u;
    if (ok) *reinterpret_cast<uint4*>(&y1[(long long)r * 64 + half * 32 + 8 * g]) = u;
  }
  __syncthreads();
  tile_col_stats<64>(tile, LD, red, part + (long long)blockIdx.x * 128, 64);
}

// ---- layers 2..5: y_out = act(y_in) . W^T,  act = ReLU(a y + b) of the previous layer -----------------------------------
// grid = row tiles, block 256 = 2 x 2 waves, a wave owns 64 rows x CT/2 columns.  The whole K = CI panel of
// both operands sits in LDS (bf16, rows padded by 16 bytes); the output tile goes back through LDS (aliasing the
// panels) for 16-byte row-segment stores and the column statistics.
// POOL (last layer, N >= 128: a tile touches at most two parts): the tile's per-column extrema and their rows go to
// `pool` ([tiles][2][CO] records {max, min, row of max, row of min}, rows counted inside the part) — pb_pool_merge_kernel
// picks by the sign of the BatchNorm scale once the statistics are final, so y5 is never read back for the pooling.
template <int CI, int CT, bool POOL>
__global__ __launch_bounds__(kT) void pb_fwd_kernel(const unsigned short* __restrict__ yin,
                                                    const float* __restrict__ coef_in,  // a [CI] | b [CI]
                                                    const unsigned short* __restrict__ Wb, int CO,
                                                    unsigned short* __restrict__ yout, float* __restrict__ part,
                                                    const int* __restrict__ hdr, int N, float4* __restrict__ pool) {
  constexpr int LDK = CI + kPad, LDC = CT + kPad, TB = CT / 64;
  constexpr int kB = CT * LDK > kRows * LDC ? CT * LDK : kRows * LDC;  // the W panel, later the output tile
  __shared__ __attribute__((aligned(16))) unsigned short As[kRows * LDK];
  __shared__ __attribute__((aligned(16))) unsigned short Bs[kB];
  __shared__ float red[2 * kT];
  const int R = hdr[1];
  const int r0 = blockIdx.x * kRows;
  if (r0 >= R) return;
  constexpr int G = CI / 8;
  // A panel: affine + ReLU in fp32, rounded to bf16 — loaded ONCE, all CO / CT column tiles are computed from it.
  // All loads are issued before the first value is used (clamped addresses, rows past the end are zeroed afterwards).
  {
    constexpr int NI = kRows * G / kT;
    uint4 raw[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = threadIdx.x + kT * i, row = idx / G, g = idx % G;
      const int rr = r0 + row < R ? r0 + row : R - 1;
      raw[i] = *reinterpret_cast<const uint4*>(&yin[(long long)rr * CI + 8 * g]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int idx = threadIdx.x + kT * i, row = idx / G, g = idx % G;
      float f[8], a[8], b[8];
      unpack8(raw[i], f);
      load8f(coef_in + 8 * g, a);
      load8f(coef_in + CI + 8 * g, b);
      const bool ok = r0 + row < R;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = ok ? __builtin_fmaxf(__builtin_fmaf(a[e], f[e], b[e]), 0.0f) : 0.0f;
      *reinterpret_cast<uint4*>(&As[row * LDK + 8 * g]) = pack8(f);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  for (int n0 = 0; n0 < CO; n0 += CT) {
#pragma unroll
    for (int i = 0; i < CT * G / kT; ++i) {
      const int idx = threadIdx.x + kT * i, row = idx / G, g = idx % G;
      *reinterpret_cast<uint4*>(&Bs[row * LDK + 8 * g]) = *reinterpret_cast<const uint4*>(&Wb[(long long)(n0 + row) * CI + 8 * g]);
    }
    __syncthreads();
    f32x16 acc[2][TB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) acc[a][b] = f32x16{0};
#pragma unroll
    for (int kk = 0; kk < CI / 16; ++kk) {
      bf16x8 fa[2], fb[TB];
#pragma unroll
      for (int a = 0; a < 2; ++a)
        fa[a] = *reinterpret_cast<const bf16x8*>(&As[(wr * 64 + a * 32 + j) * LDK + kk * 16 + 8 * h]);
#pragma unroll
      for (int b = 0; b < TB; ++b)
        fb[b] = *reinterpret_cast<const bf16x8*>(&Bs[(wc * (CT / 2) + b * 32 + j) * LDK + kk * 16 + 8 * h]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
    __syncthreads();  // the W panel is dead: the output tile takes its place
    unsigned short* Cs = Bs;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          Cs[(wr * 64 + a * 32 + dg::acc_row(r, h)) * LDC + wc * (CT / 2) + b * 32 + j] = pack1(acc[a][b][r]);
    __syncthreads();
    constexpr int GC = CT / 8;
#pragma unroll
    for (int i = 0; i < kRows * GC / kT; ++i) {
      const int idx = threadIdx.x + kT * i, row = idx / GC, g = idx % GC;
      if (r0 + row < R)
        *reinterpret_cast<uint4*>(&yout[(long long)(r0 + row) * CO + n0 + 8 * g]) = *reinterpret_cast<const uint4*>(&Cs[row * LDC + 8 * g]);
    }
    tile_col_stats<CT>(Cs, LDC, red, part + (long long)blockIdx.x * 2 * CO + n0, CO);
    if constexpr (POOL) {
      constexpr int SLP = kT / CT, RPS = kRows / SLP;
      __shared__ float4 pm[SLP][2][CT];
      const int c = threadIdx.x % CT, sl = threadIdx.x / CT;
      const int v0 = r0 / N, bnd = (v0 + 1) * N - r0;  // tile-local row where the next part starts
      const int lim = R - r0 < kRows ? R - r0 : kRows;
      float mxa = -__builtin_inff(), mna = __builtin_inff(), mxb = -__builtin_inff(), mnb = __builtin_inff();
      int xa = 0, na = 0, xb = 0, nb = 0;
      for (int i0 = sl * RPS; i0 < (sl + 1) * RPS; i0 += 8) {  // ascending rows, strict compares: lowest row on ties
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = unpack1(Cs[(i0 + u) * LDC + c]);  // eight LDS reads in flight
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u;
          const float v = v8[u];
          const bool ina = i < bnd && i < lim, inb = i >= bnd && i < lim;
          const int rpa = r0 + i - v0 * N, rpb = i - bnd;
          if (ina && v > mxa) { mxa = v; xa = rpa; }
          if (ina && v < mna) { mna = v; na = rpa; }
          if (inb && v > mxb) { mxb = v; xb = rpb; }
          if (inb && v < mnb) { mnb = v; nb = rpb; }
        }
      }
      pm[sl][0][c] = make_float4(mxa, mna, __int_as_float(xa), __int_as_float(na));
      pm[sl][1][c] = make_float4(mxb, mnb, __int_as_float(xb), __int_as_float(nb));
      __syncthreads();
      if (threadIdx.x < 2 * CT) {
        const int slot = threadIdx.x / CT;
        float4 best = pm[0][slot][c];
#pragma unroll
        for (int k = 1; k < SLP; ++k) {
          const float4 o = pm[k][slot][c];
          if (o.x > best.x) { best.x = o.x; best.z = o.z; }
          if (o.y < best.y) { best.y = o.y; best.w = o.w; }
        }
        pool[((long long)blockIdx.x * 2 + slot) * CO + n0 + c] = best;
      }
    }
    __syncthreads();  // before the next W panel overwrites the tile
  }
}

// feat = a ext + b from the per-tile extrema of pb_fwd_kernel<.., POOL>; grid = part slots, block F.
__global__ void pb_pool_merge_kernel(const float4* __restrict__ pool, const float* __restrict__ coef, int N, int F,
                                     const int* __restrict__ rank, float* __restrict__ feat, int* __restrict__ arg) {
  const int m = blockIdx.x, c = threadIdx.x, v = rank[m];
  if (v < 0) {
    feat[(long long)m * F + c] = 0.0f;
    return;
  }
  const float a = coef[c], b = coef[F + c];
  const int t0 = (int)(((long long)v * N) / kRows), t1 = (int)((((long long)v + 1) * N - 1) / kRows);
  float best = a >= 0.0f ? -__builtin_inff() : __builtin_inff();
  int row = 0;
  for (int t = t0; t <= t1; ++t) {  // ascending tiles, strict compares: lowest row on ties
    const int slot = v - (int)(((long long)t * kRows) / N);
    const float4 r = pool[((long long)t * 2 + slot) * F + c];
    if (a >= 0.0f) {
      if (r.x > best) { best = r.x; row = __float_as_int(r.z); }
    } else {
      if (r.y < best) { best = r.y; row = __float_as_int(r.w); }
    }
  }
  feat[(long long)m * F + c] = __builtin_fmaf(a, best, b);
  arg[(long long)v * F + c] = row;
}

constexpr int kRS = 64;  // slices of the partial tables in the reduction kernels (1024 threads = 16 channels x 64 slices)
// slice `sl` of kRS of the per-tile partials of channel c: part[t][c] and part[t][C + c], eight loads in flight
__device__ __forceinline__ void sum_partials(const float* __restrict__ part, int C, int c, int sl, int tiles, double& s,
                                             double& q) {
  int t = sl;
  for (; t + kRS * 7 < tiles; t += kRS * 8) {  // 16 loads in flight
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = part[(long long)(t + kRS * u) * 2 * C + c];
      b[u] = part[(long long)(t + kRS * u) * 2 * C + C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s += (double)a[u];
      q += (double)b[u];
    }
  }
  for (; t < tiles; t += kRS) {
    s += (double)part[(long long)t * 2 * C + c];
    q += (double)part[(long long)t * 2 * C + C + c];
  }
}

// ---- BatchNorm finalisation: partial sums -> coef = a | b | mean | rstd; running statistics --------------------------------
// grid = C / 16, block 1024: 16 channels x 64 slices of the tile range; double accumulation, fixed order.
__global__ __launch_bounds__(1024) void pb_finalize_kernel(const float* __restrict__ part, int C, const int* __restrict__ hdr,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ rmean, float* __restrict__ rvar,
                                                         int training, float momentum, float eps,
                                                         float* __restrict__ coef) {
  __shared__ double red[2][kRS / 4][16];
  const int R = hdr[1];
  const int tiles = (R + kRows - 1) / kRows;
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  double s = 0.0, q = 0.0;
  if (training) sum_partials(part, C, c, sl, tiles, s, q);
  // a wave holds 4 slices of the 16 channels: fold them with two butterflies, then the 16 waves meet in LDS
  s += __shfl_xor(s, 16, 64);
  q += __shfl_xor(q, 16, 64);
  s += __shfl_xor(s, 32, 64);
  q += __shfl_xor(q, 32, 64);
  if ((threadIdx.x & 63) < 16) {
    red[0][threadIdx.x >> 6][cl] = s;
    red[1][threadIdx.x >> 6][cl] = q;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double ts = 0.0, tq = 0.0;
#pragma unroll 4
    for (int k = 0; k < kRS / 4; ++k) {
      ts += red[0][k][cl];
      tq += red[1][k][cl];
    }
    float mean, var;
    if (training) {
      if (R == 0) {  // no valid part: nothing to normalise, statistics untouched
        coef[c] = 0.0f;
        coef[C + c] = 0.0f;
        coef[2 * C + c] = 0.0f;
        coef[3 * C + c] = 0.0f;
        return;
      }
      const double m = ts / (double)R;
      double v = tq / (double)R - m * m;
      v = v > 0.0 ? v : 0.0;
      mean = (float)m;
      var = (float)v;
      const float unb = R > 1 ? (float)(v * (double)R / (double)(R - 1)) : var;
      rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.0f - momentum) * rvar[c] + momentum * unb;
    } else {
      mean = rmean[c];
      var = rvar[c];
    }
    const float rstd = 1.0f / __builtin_sqrtf(var + eps);
    const float a = gamma[c] * rstd;
    coef[c] = a;
    coef[C + c] = beta[c] - mean * a;
    coef[2 * C + c] = mean;
    coef[3 * C + c] = rstd;
  }
}

// ---- max over the points of a part: feat = a ext(y5) + b, ext = max or min by the sign of a; arg = row inside the part ----
// grid = part slots, block 256 = (F / 8 column groups) x row slices.
__global__ __launch_bounds__(kT) void pb_pool_kernel(const unsigned short* __restrict__ y5, const float* __restrict__ coef,
                                                     int N, int F, const int* __restrict__ rank,
                                                     float* __restrict__ feat, int* __restrict__ arg) {
  __shared__ float bk[kT * 8];
  __shared__ int bi[kT * 8];
  const int m = blockIdx.x, v = rank[m];
  if (v < 0) {
    for (int c = threadIdx.x; c < F; c += kT) feat[(long long)m * F + c] = 0.0f;
    return;
  }
  const int G = F / 8, S = kT / G;
  const int g = threadIdx.x % G, sl = threadIdx.x / G;
  float a[8], b[8], key[8];
  int idx[8];
  load8f(coef + 8 * g, a);
  load8f(coef + F + 8 * g, b);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    key[e] = -__builtin_inff();
    idx[e] = 0;
  }
  const unsigned short* base = y5 + (long long)v * N * F + 8 * g;
  for (int i0 = sl; i0 < N; i0 += 4 * S) {  // four rows in flight per thread
    uint4 u[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * S;
      u[q] = *reinterpret_cast<const uint4*>(base + (long long)(i < N ? i : sl) * F);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + q * S;
      float f[8];
      unpack8(u[q], f);
      if (i < N) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float k = a[e] >= 0.0f ? f[e] : -f[e];
          if (k > key[e]) {
            key[e] = k;
            idx[e] = i;
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    bk[threadIdx.x * 8 + e] = key[e];
    bi[threadIdx.x * 8 + e] = idx[e];
  }
  __syncthreads();
  if (threadIdx.x < G) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float k = key[e];
      int ix = idx[e];
      for (int s2 = 1; s2 < S; ++s2) {  // slices hold interleaved rows: ties go to the lower row
        const float ok = bk[(s2 * G + g) * 8 + e];
        const int oi = bi[(s2 * G + g) * 8 + e];
        if (ok > k || (ok == k && oi < ix)) {
          k = ok;
          ix = oi;
        }
      }
      const float yv = a[e] >= 0.0f ? k : -k;
      feat[(long long)m * F + 8 * g + e] = __builtin_fmaf(a[e], yv, b[e]);
      arg[(long long)v * F + 8 * g + e] = ix;
    }
  }
}

// ================================================ backward ================================================================
// Per layer: dy[r][c] = a[c] G[r][c] + P[c] + Q[c] y[r][c]   (BatchNorm backward with the batch sums folded into P, Q):
//   s1 = sum_r G, s2 = sum_r G yhat, yhat = (y - mean) rstd;  dgamma = s2, dbeta = s1,
//   P = -a s1 / R + a s2 mean rstd / R,  Q = -a s2 rstd / R.            bwd coefficient block: a | P | Q  (3 C floats)

// last layer: G = grad_feat at the arg-max rows.  One thread per (valid part, channel): writes the part's terms of
// s1 = sum G and s2 = sum G yhat into the partial table (row = part) and marks the arg-max position in `bitmap` (one bit
// per (row, 8-channel group), cleared beforehand): the kernels that form dy on the fly test one bit instead of comparing
// 8 arg-max rows per 16 bytes.  grid = part slots (worst case), block F.
__global__ void pb_top_mark_kernel(const float* __restrict__ grad_feat, const int* __restrict__ vlist,
                                   const int* __restrict__ arg, const unsigned short* __restrict__ y5,
                                   const float* __restrict__ coef, int N, int F, const int* __restrict__ hdr,
                                   float* __restrict__ part, unsigned* __restrict__ bitmap) {
  const int v = blockIdx.x, c = threadIdx.x;
  if (v >= hdr[0]) return;
  const long long r = (long long)v * N + arg[(long long)v * F + c];
  const float g = grad_feat[(long long)vlist[v] * F + c];
  const float y = unpack1(y5[r * F + c]);
  part[(long long)v * 2 * F + c] = g;
  part[(long long)v * 2 * F + F + c] = g * ((y - coef[2 * F + c]) * coef[3 * F + c]);
  atomicOr(&bitmap[r], 1u << (c >> 3));  // one word per row: F / 8 <= 32 groups (an OR: the result has no order)
}

// s1, s2 from the partial table -> dgamma, dbeta and the coefficients a | P | Q.  The table has one row per row tile
// (layers 4..1: written by the input-gradient kernel) or per valid part (last layer: pb_top_mark_kernel).
// grid = C / 16, block 1024.
__global__ __launch_bounds__(1024) void pb_bwd_coef_kernel(const float* __restrict__ part, int C, const int* __restrict__ hdr,
                                                           int per_part, const float* __restrict__ coef,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           float* __restrict__ bc) {
  __shared__ double red[2][kRS / 4][16];
  const int R = hdr[1];
  const int tiles = per_part ? hdr[0] : (R + kRows - 1) / kRows;
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
  double s = 0.0, q = 0.0;
  sum_partials(part, C, c, sl, tiles, s, q);
  // a wave holds 4 slices of the 16 channels: fold them with two butterflies, then the 16 waves meet in LDS
  s += __shfl_xor(s, 16, 64);
  q += __shfl_xor(q, 16, 64);
  s += __shfl_xor(s, 32, 64);
  q += __shfl_xor(q, 32, 64);
  if ((threadIdx.x & 63) < 16) {
    red[0][threadIdx.x >> 6][cl] = s;
    red[1][threadIdx.x >> 6][cl] = q;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll 4
    for (int k = 0; k < kRS / 4; ++k) {
      t1 += red[0][k][cl];
      t2 += red[1][k][cl];
    }
    const float a = coef[c], mean = coef[2 * C + c], rstd = coef[3 * C + c];
    const float inv = R > 0 ? 1.0f / (float)R : 0.0f, f1 = (float)t1, f2 = (float)t2;
    dgamma[c] = f2;
    dbeta[c] = f1;
    bc[c] = a;
    bc[C + c] = (-a * f1 + a * f2 * mean * rstd) * inv;
    bc[2 * C + c] = -a * f2 * rstd * inv;
  }
}

struct TopSrc {
  const float* grad_feat;  // [M][F]
  const int* vlist;
  const int* arg;          // [nv][F]
  const unsigned* bitmap;  // bit (row, 8-channel group): some channel of the group has its arg-max in this row
  int N;
  float invN;
};
// part of row r (r < 2^31): float estimate + one correction step instead of an integer division per 8 values
__device__ __forceinline__ int part_of(long long r, int N, float invN) {
  int v = (int)((float)r * invN);
  const long long lo = (long long)v * N;
  v += lo > r ? -1 : (lo + N <= r ? 1 : 0);
  return v;
}
// the pooled gradient of 8 channels of row r (the rare rows that hold an arg-max of the group)
template <int CO>
__device__ __forceinline__ void top_g(const TopSrc& top, long long r, int c8, float* fg) {
  const int v = part_of(r, top.N, top.invN), li = (int)(r - (long long)v * top.N);
  const int4 a0 = *reinterpret_cast<const int4*>(&top.arg[(long long)v * CO + c8]);
  const int4 a1 = *reinterpret_cast<const int4*>(&top.arg[(long long)v * CO + c8 + 4]);
  const int ai[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  float gf[8];
  load8f(top.grad_feat + (long long)top.vlist[v] * CO + c8, gf);
#pragma unroll
  for (int e = 0; e < 8; ++e) fg[e] = ai[e] == li ? gf[e] : 0.0f;
}

// dy of 8 consecutive channels of one row.  TOP: G comes from the pooled gradient (row == arg row of its part).
template <bool TOP, int CO>
__device__ __forceinline__ void dy8(const unsigned short* __restrict__ G, const unsigned short* __restrict__ y,
                                    const TopSrc& top, const float* __restrict__ bc_lds, long long r, int c8, float* dy) {
  float fy[8], fg[8];
  unpack8(*reinterpret_cast<const uint4*>(&y[r * CO + c8]), fy);
  if constexpr (TOP) {
#pragma unroll
    for (int e = 0; e < 8; ++e) fg[e] = 0.0f;
    if ((top.bitmap[r] >> (c8 >> 3)) & 1u) top_g<CO>(top, r, c8, fg);  // 8 rows in N per column group
  } else {
    unpack8(*reinterpret_cast<const uint4*>(&G[r * CO + c8]), fg);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
    dy[e] = __builtin_fmaf(bc_lds[c8 + e], fg[e], __builtin_fmaf(bc_lds[2 * CO + c8 + e], fy[e], bc_lds[CO + c8 + e]));
}

// ---- input gradient of layer l:  g_prev = (dy_l . W_l) * [a_prev y_prev + b_prev > 0], stored bf16, with the per-tile
// partial sums s1 = sum g_prev, s2 = sum g_prev yhat_prev.  grid = row tiles, block 256 = 2 x 2 waves (64 rows x CI / 2).
// K = CO is walked in panels of KP <= 128.
template <int CI, int CO, bool TOP>
__global__ __launch_bounds__(kT) void pb_dgrad_kernel(const unsigned short* __restrict__ G, const unsigned short* __restrict__ y,
                                                      const TopSrc top, const float* __restrict__ bc,  // a | P | Q of layer l
                                                      const unsigned short* __restrict__ Wt,            // [CI][CO] bf16
                                                      const unsigned short* __restrict__ yprev,
                                                      const float* __restrict__ coef_prev,              // a | b | mean | rstd
                                                      unsigned short* __restrict__ gprev, float* __restrict__ part,
                                                      const int* __restrict__ hdr) {
  constexpr int KP = CO < 128 ? CO : 128, NPH = CO / KP, LDK = KP + kPad, LDC = CI + 4, TB = CI / 64;
  constexpr int kPanelBytes = (kRows + CI) * LDK * 2, kOutBytes = kRows * LDC * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kPanelBytes > kOutBytes ? kPanelBytes : kOutBytes];
  __shared__ float bcs[3 * CO];
  __shared__ float red[4][2][CI];
  unsigned short* As = reinterpret_cast<unsigned short*>(lds);
  unsigned short* Bs = As + kRows * LDK;
  const int R = hdr[1];
  const int r0 = blockIdx.x * kRows;
  if (r0 >= R) return;
  for (int i = threadIdx.x; i < 3 * CO; i += kT) bcs[i] = bc[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc[2][TB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = f32x16{0};
  constexpr int GK = KP / 8;
#pragma unroll 1
  for (int ph = 0; ph < NPH; ++ph) {
    if (ph > 0) __syncthreads();
    {  // every load of the phase is issued before the first value is used
      constexpr int NA_ = kRows * GK / kT, NB_ = CI * GK / kT;
      uint4 ry[NA_], rg[TOP ? 1 : NA_], rw[NB_];
#pragma unroll
      for (int i = 0; i < NA_; ++i) {
        const int idx = threadIdx.x + kT * i, row = idx / GK, g = idx % GK;
        const long long rr = r0 + row < R ? r0 + row : R - 1;
        ry[i] = *reinterpret_cast<const uint4*>(&y[rr * CO + ph * KP + 8 * g]);
        if constexpr (!TOP) rg[i] = *reinterpret_cast<const uint4*>(&G[rr * CO + ph * KP + 8 * g]);
      }
#pragma unroll
      for (int i = 0; i < NB_; ++i) {
        const int idx = threadIdx.x + kT * i, row = idx / GK, g = idx % GK;
        rw[i] = *reinterpret_cast<const uint4*>(&Wt[(long long)row * CO + ph * KP + 8 * g]);
      }
#pragma unroll
      for (int i = 0; i < NA_; ++i) {
        const int idx = threadIdx.x + kT * i, row = idx / GK, g = idx % GK, c8 = ph * KP + 8 * g;
        const bool ok = r0 + row < R;
        float fy[8], fg[8], d[8];
        unpack8(ry[i], fy);
        if constexpr (TOP) {
#pragma unroll
          for (int e = 0; e < 8; ++e) fg[e] = 0.0f;  // the few arg-max positions are patched below
        } else {
          unpack8(rg[i], fg);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
          d[e] = ok ? __builtin_fmaf(bcs[c8 + e], fg[e], __builtin_fmaf(bcs[2 * CO + c8 + e], fy[e], bcs[CO + c8 + e])) : 0.0f;
        *reinterpret_cast<uint4*>(&As[row * LDK + 8 * g]) = pack8(d);
      }
#pragma unroll
      for (int i = 0; i < NB_; ++i) {
        const int idx = threadIdx.x + kT * i, row = idx / GK, g = idx % GK;
        *reinterpret_cast<uint4*>(&Bs[row * LDK + 8 * g]) = rw[i];
      }
      if constexpr (TOP) {
        // one item in ~N / 8 holds an arg-max of its 8 channels: those are recomputed with the pooled gradient (the
        // thread rewrites its own LDS slot; a rolled loop keeps the common path free of this code's registers)
#pragma unroll 1
        for (int i = 0; i < NA_; ++i) {
          const int idx = threadIdx.x + kT * i, row = idx / GK, g = idx % GK, c8 = ph * KP + 8 * g;
          if (r0 + row >= R) continue;
          const long long rr = r0 + row;
          if (!((top.bitmap[rr] >> (c8 >> 3)) & 1u)) continue;
          float d[8];
          dy8<true, CO>(G, y, top, bcs, rr, c8, d);
          *reinterpret_cast<uint4*>(&As[row * LDK + 8 * g]) = pack8(d);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KP / 16; ++kk) {
      bf16x8 fa[2], fb[TB];
#pragma unroll
      for (int a = 0; a < 2; ++a)
        fa[a] = *reinterpret_cast<const bf16x8*>(&As[(wr * 64 + a * 32 + j) * LDK + kk * 16 + 8 * h]);
#pragma unroll
      for (int b = 0; b < TB; ++b)
        fb[b] = *reinterpret_cast<const bf16x8*>(&Bs[(wc * (CI / 2) + b * 32 + j) * LDK + kk * 16 + 8 * h]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  }
  __syncthreads();
  float* Cs = reinterpret_cast<float*>(lds);  // fp32 output tile over the dead panels
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(wr * 64 + a * 32 + dg::acc_row(r, h)) * LDC + wc * (CI / 2) + b * 32 + j] = acc[a][b][r];
  __syncthreads();
  // epilogue: a thread owns ONE 8-channel group (t % GC) and rows t / GC + (256 / GC) i
  constexpr int GC = CI / 8, RS = kT / GC;
  const int g = threadIdx.x % GC, rb = threadIdx.x / GC;
  float pa[8], pb_[8], pm[8], pr[8], s1[8], s2[8];
  load8f(coef_prev + 8 * g, pa);
  load8f(coef_prev + CI + 8 * g, pb_);
  load8f(coef_prev + 2 * CI + 8 * g, pm);
  load8f(coef_prev + 3 * CI + 8 * g, pr);
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.0f;
  {
    constexpr int NE = kRows / RS;
    uint4 ryp[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int row = rb + RS * i;
      const long long rr = r0 + row < R ? r0 + row : R - 1;
      ryp[i] = *reinterpret_cast<const uint4*>(&yprev[rr * CI + 8 * g]);
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int row = rb + RS * i;
      if (r0 + row < R) {
        float fy[8], o[8];
        unpack8(ryp[i], fy);
        const float4 c0 = *reinterpret_cast<const float4*>(&Cs[row * LDC + 8 * g]);
        const float4 c1 = *reinterpret_cast<const float4*>(&Cs[row * LDC + 8 * g + 4]);
        const float d[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(pa[e], fy[e], pb_[e]) > 0.0f ? d[e] : 0.0f;
        const uint4 u = pack8(o);
        *reinterpret_cast<uint4*>(&gprev[(long long)(r0 + row) * CI + 8 * g]) = u;
        float rq[8];
        unpack8(u, rq);  // the sums see the stored (rounded) values
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += rq[e];
          s2[e] += rq[e] * ((fy[e] - pm[e]) * pr[e]);
        }
      }
    }
  }
  // lanes with equal t % GC hold the same channels: butterfly over the other lane bits, then the four waves through LDS
#pragma unroll
  for (int off = GC; off < 64; off <<= 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += __shfl_xor(s1[e], off, 64);
      s2[e] += __shfl_xor(s2[e], off, 64);
    }
  }
  if (lane < GC) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[wave][0][8 * lane + e] = s1[e];
      red[wave][1][8 * lane + e] = s2[e];
    }
  }
  __syncthreads();
  if (threadIdx.x < CI) {
    const int c = threadIdx.x;
    part[(long long)blockIdx.x * 2 * CI + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
    part[(long long)blockIdx.x * 2 * CI + CI + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
  }
}

// ---- weight gradient of layer l:  dW[co][ci] = sum_r dy_l[r][co] act_prev[r][ci] ----------------------------------------
// grid = row chunks (an even split of the VALID rows, from hdr), block 256; the block owns the whole CO x CI gradient of
// its rows ((CO/32)(CI/32) MFMA tiles dealt round-robin to the four waves).  The reduction index of the MFMA is the
// row, so both operands are transposed while they are staged: a lane loads 16 bytes of one row (8 lanes = one 128-byte
// line) and scatters its 8 values to 8 LDS rows.  The global loads of step s + 1 are in flight during the MFMAs of
// step s.
template <int CI, int CO>
constexpr int wgrad_threads() { return (CO / 32) * (CI / 32) >= 32 ? 512 : 256; }  // 8 waves for the 256 x 128 gradient

template <int CI, int CO, bool TOP>
__global__ __launch_bounds__((wgrad_threads<CI, CO>())) void pb_wgrad_kernel(
    const unsigned short* __restrict__ G, const unsigned short* __restrict__ y, const TopSrc top,
    const float* __restrict__ bc, const unsigned short* __restrict__ yprev, const float* __restrict__ coef_prev,
    float* __restrict__ partw, const int* __restrict__ hdr) {
  constexpr int NT = wgrad_threads<CI, CO>(), NW = NT / 64;
  constexpr int TM = CO / 32, TN = CI / 32, TPW = TM * TN / NW;  // MFMA tiles per wave
  constexpr int ND = CO / 8 / NW, NA = CI / 8 / NW;              // 16-byte items per thread and step
  static_assert(TM * TN % NW == 0 && ND >= 1 && NA >= 1, "work split");
  __shared__ __attribute__((aligned(16))) unsigned short Td[CO * kWLD];
  __shared__ __attribute__((aligned(16))) unsigned short Ta[CI * kWLD];
  __shared__ float bcs[3 * CO];
  __shared__ float cps[2 * CI];
  const int R = hdr[1];
  const int rpc = ((R + (int)gridDim.x - 1) / (int)gridDim.x + kWS - 1) / kWS * kWS;
  const long long rb = (long long)blockIdx.x * rpc;
  long long re = rb + rpc;
  if (re > R) re = R;
  for (int i = threadIdx.x; i < 3 * CO; i += NT) bcs[i] = bc[i];
  for (int i = threadIdx.x; i < 2 * CI; i += NT) cps[i] = coef_prev[i];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  f32x16 acc[TPW];
#pragma unroll
  for (int u = 0; u < TPW; ++u) acc[u] = f32x16{0};
  // staging: load instruction `id` of a step covers 8 rows x 8 column groups (one 128-byte line per row); wave w issues
  // ids w, w + NW, ...: id % 8 = row block, id / 8 = 64-channel block.  The 8-row blocks of an LDS row are XOR-swizzled
  // by the column group so that the 64 two-byte stores of an instruction fall on 32 different banks.
  const int rs = lane >> 3, gl = lane & 7;
  constexpr int NWD = NW < 8 ? 8 / NW : 1;  // distinct row blocks (bitmap words) per thread
  uint4 ry[ND], rg[TOP ? 1 : ND], rp[NA];
  unsigned word[NWD];
  auto fetch = [&](long long r) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int id = wave + NW * i, row = 8 * (id & 7) + rs, c8 = 64 * (id >> 3) + 8 * gl;
      const long long rr = r + row < re ? r + row : re - 1;  // clamped: the values of rows past the end are dropped
      ry[i] = *reinterpret_cast<const uint4*>(&y[rr * CO + c8]);
      if constexpr (!TOP) rg[i] = *reinterpret_cast<const uint4*>(&G[rr * CO + c8]);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = wave + NW * i, row = 8 * (id & 7) + rs, c8 = 64 * (id >> 3) + 8 * gl;
      const long long rr = r + row < re ? r + row : re - 1;
      rp[i] = *reinterpret_cast<const uint4*>(&yprev[rr * CI + c8]);
    }
    if constexpr (TOP) {
#pragma unroll
      for (int u = 0; u < NWD; ++u) {
        const long long rr = r + 8 * ((wave + NW * u) & 7) + rs;
        word[u] = rr < re ? top.bitmap[rr] : 0u;
      }
    }
  };
  if (rb < re) fetch(rb);
  for (long long r = rb; r < re; r += kWS) {
    __syncthreads();  // (first pass: the coefficient tables; later: the previous step's fragments have been read)
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int id = wave + NW * i, rbk = id & 7, row = 8 * rbk + rs, c8 = 64 * (id >> 3) + 8 * gl;
      const bool ok = r + row < re;
      float fy[8], fg[8];
      unpack8(ry[i], fy);
      if constexpr (TOP) {
#pragma unroll
        for (int e = 0; e < 8; ++e) fg[e] = 0.0f;
        if ((word[i % NWD] >> (c8 >> 3)) & 1u) top_g<CO>(top, r + row, c8, fg);
      } else {
        unpack8(rg[i], fg);
      }
      const int slot = 8 * (rbk ^ gl) + rs;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = __builtin_fmaf(bcs[c8 + e], fg[e], __builtin_fmaf(bcs[2 * CO + c8 + e], fy[e], bcs[CO + c8 + e]));
        Td[(c8 + e) * kWLD + slot] = ok ? pack1(d) : (unsigned short)0;
      }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = wave + NW * i, rbk = id & 7, row = 8 * rbk + rs, c8 = 64 * (id >> 3) + 8 * gl;
      const bool ok = r + row < re;
      float f[8];
      unpack8(rp[i], f);
      const int slot = 8 * (rbk ^ gl) + rs;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = __builtin_fmaxf(__builtin_fmaf(cps[c8 + e], f[e], cps[CI + c8 + e]), 0.0f);
        Ta[(c8 + e) * kWLD + slot] = ok ? pack1(a) : (unsigned short)0;
      }
    }
    __syncthreads();
    if (r + kWS < re) fetch(r + kWS);
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int q = wave + NW * u, tm = q / TN, tn = q % TN;
      const int sa = ((tm * 32 + j) >> 3) & 7, sb = ((tn * 32 + j) >> 3) & 7;  // the rows' swizzle keys
#pragma unroll
      for (int ks = 0; ks < kWS / 16; ++ks) {
        const int kb = 2 * ks + h;  // 8-row block of the reduction index
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(&Td[(tm * 32 + j) * kWLD + 8 * (kb ^ sa)]);
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(&Ta[(tn * 32 + j) * kWLD + 8 * (kb ^ sb)]);
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[u], 0, 0, 0);
      }
    }
  }
  float* out = partw + (long long)blockIdx.x * CO * CI;
#pragma unroll
  for (int u = 0; u < TPW; ++u) {
    const int q = wave + NW * u, tm = q / TN, tn = q % TN;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(tm * 32 + dg::acc_row(r, h)) * CI + tn * 32 + j] = acc[u][r];
  }
}

// ---- first layer's weight gradient: dW1[c][k] = sum_r dy_1[r][c] x[r][k]; grid = row tiles, partw [tiles][192] -------------
__global__ __launch_bounds__(kT) void pb_first_wgrad_kernel(const unsigned short* __restrict__ g1,
                                                            const unsigned short* __restrict__ y1,
                                                            const float* __restrict__ bc, const float* __restrict__ points,
                                                            const int* __restrict__ vlist, int N, float* __restrict__ partw,
                                                            const int* __restrict__ hdr) {
  __shared__ float xs[kRows * 3];
  __shared__ float red[4][192];
  const int R = hdr[1];
  const int r0 = blockIdx.x * kRows;
  if (r0 >= R) {  // (the reduction reads every tile up to the worst case)
    if (threadIdx.x < 192) partw[(long long)blockIdx.x * 192 + threadIdx.x] = 0.0f;
    return;
  }
  for (int i = threadIdx.x; i < kRows * 3; i += kT) {
    const int row = i / 3, k = i - 3 * row, r = r0 + row;
    float v = 0.0f;
    if (r < R) {
      const int p = r / N, li = r - p * N;
      v = points[((long long)vlist[p] * N + li) * 3 + k];
    }
    xs[i] = v;
  }
  __syncthreads();
  const int c = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const float a = bc[c], P = bc[64 + c], Q = bc[128 + c];
  float w0 = 0.0f, w1 = 0.0f, w2 = 0.0f;
  for (int rr = 0; rr < 32; ++rr) {
    const int row = sl * 32 + rr, r = r0 + row;
    if (r < R) {
      const float g = unpack1(g1[(long long)r * 64 + c]), y = unpack1(y1[(long long)r * 64 + c]);
      const float d = __builtin_fmaf(a, g, __builtin_fmaf(Q, y, P));
      w0 = __builtin_fmaf(d, xs[row * 3 + 0], w0);
      w1 = __builtin_fmaf(d, xs[row * 3 + 1], w1);
      w2 = __builtin_fmaf(d, xs[row * 3 + 2], w2);
    }
  }
  red[sl][c * 3 + 0] = w0;
  red[sl][c * 3 + 1] = w1;
  red[sl][c * 3 + 2] = w2;
  __syncthreads();
  if (threadIdx.x < 192)
    partw[(long long)blockIdx.x * 192 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ================================================ host side ================================================================
struct Ws {
  int* hdr;
  int* vlist;
  int* rank;
  int* arg;
  float* coef[5];   // a | b | mean | rstd
  float* bc[5];     // a | P | Q
  unsigned short* wb[5];  // [CO][CI] bf16 (layers 2..5)
  unsigned short* wt[5];  // [CI][CO]
  unsigned short* y[5];
  unsigned short* g[5];  // g[l]: gradient w.r.t. the BatchNorm output of layer l + 1, l < 4 (the last layer's is never stored)
  float* part;
  float* partw;
  float4* pool;      // [tiles][2][F] extrema records of the last layer's tiles
  unsigned* bitmap;  // one word per row (backward of the last layer)
};

constexpr int kC[6] = {3, 64, 64, 64, 128, 0};

char* take(char*& p, int64_t bytes) {
  char* r = p;
  p += (bytes + 255) / 256 * 256;
  return r;
}

Ws carve(void* base, int64_t M, int64_t N, int64_t F, int64_t* total) {
  Ws w{};
  char* p = static_cast<char*>(base);
  const int64_t R = M * N, tiles = (R + kRows - 1) / kRows;
  int C[6];
  for (int l = 0; l < 5; ++l) C[l] = kC[l];
  C[5] = (int)F;
  w.hdr = reinterpret_cast<int*>(take(p, 256));
  w.vlist = reinterpret_cast<int*>(take(p, 4 * M));
  w.rank = reinterpret_cast<int*>(take(p, 4 * M));
  w.arg = reinterpret_cast<int*>(take(p, 4 * M * F));
  for (int l = 0; l < 5; ++l) {
    w.coef[l] = reinterpret_cast<float*>(take(p, 4 * 4 * C[l + 1]));
    w.bc[l] = reinterpret_cast<float*>(take(p, 4 * 3 * C[l + 1]));
    w.wb[l] = reinterpret_cast<unsigned short*>(take(p, 2 * (int64_t)C[l + 1] * (l == 0 ? 64 : C[l])));
    w.wt[l] = reinterpret_cast<unsigned short*>(take(p, 2 * (int64_t)C[l + 1] * (l == 0 ? 64 : C[l])));
    w.y[l] = reinterpret_cast<unsigned short*>(take(p, 2 * R * C[l + 1]));
    if (l < 4) w.g[l] = reinterpret_cast<unsigned short*>(take(p, 2 * R * C[l + 1]));
  }
  w.part = reinterpret_cast<float*>(take(p, 4 * (tiles > M ? tiles : M) * 2 * (F > 128 ? F : 128)));  // rows: row tiles or parts
  int64_t pw = (int64_t)kChunks * F * 128;
  if (tiles * 192 > pw) pw = tiles * 192;
  w.partw = reinterpret_cast<float*>(take(p, 4 * pw));
  w.bitmap = reinterpret_cast<unsigned*>(take(p, 4 * R));
  w.pool = reinterpret_cast<float4*>(take(p, 16 * tiles * 2 * F));
  if (total) *total = p - static_cast<char*>(base);
  return w;
}

int check_dims(int64_t M, int64_t N, int64_t F, const char* who) {
  MPA_REQUIRE(M >= 0 && N >= 1 && N <= 32768, "%s: bad part / point count", who);
  MPA_REQUIRE(F == 64 || F == 128 || F == 256, "%s: feature width must be 64, 128 or 256", who);
  MPA_REQUIRE(M * N < (1LL << 31) - 65536, "%s: more than 2^31 point rows", who);
  return MPA_OK;
}

template <typename K, typename... A>
void launch(K k, dim3 g, dim3 b, hipStream_t s, A... a) {
  hipLaunchKernelGGL(k, g, b, 0, s, a...);
}

}  // namespace

extern "C" int mpa_pointnet_workspace_bf16(int64_t M, int64_t N, int64_t F, int64_t* bytes) {
  if (int st = check_dims(M, N, F, "pointnet_workspace_bf16")) return st;
  MPA_REQUIRE(bytes != nullptr, "pointnet_workspace_bf16: null pointer");
  int64_t total = 0;
  carve(nullptr, M, N, F, &total);
  *bytes = total + 256;
  return MPA_OK;
}

extern "C" int mpa_pointnet_forward_bf16(const float* points, const float* valids, const float* const* conv_w,
                                         const float* const* bn_w, const float* const* bn_b,
                                         float* const* running_mean, float* const* running_var, int training,
                                         float momentum, float eps, int64_t M, int64_t N, int64_t F, void* ws, float* feat,
                                         void* stream) {
  if (int st = check_dims(M, N, F, "pointnet_forward_bf16")) return st;
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(points && valids && conv_w && bn_w && bn_b && running_mean && running_var && ws && feat,
              "pointnet_forward_bf16: null pointer");
  MPA_REQUIRE((uintptr_t)ws % 256 == 0, "pointnet_forward_bf16: workspace must be 256-byte aligned");
  hipStream_t s = mpa::as_stream(stream);
  const Ws w = carve(ws, M, N, F, nullptr);
  const int64_t R = M * N;
  const unsigned tiles = (unsigned)((R + kRows - 1) / kRows);
  const int C[6] = {3, 64, 64, 64, 128, (int)F};
  const int* hdr = w.hdr;
  launch(pb_prepare_kernel, dim3(1), dim3(1024), s, valids, (int)M, (int)N, w.hdr, w.vlist, w.rank);
  for (int l = 1; l < 5; ++l)
    launch(pb_weights_kernel, dim3((unsigned)((C[l + 1] * C[l] + 255) / 256)), dim3(256), s, conv_w[l], C[l + 1], C[l],
           w.wb[l], w.wt[l]);
  launch(pb_first_fwd_kernel, dim3(tiles), dim3(kT), s, points, (const int*)w.vlist, (int)N, conv_w[0], w.y[0], w.part, hdr);
  auto finalize = [&](int l) {
    launch(pb_finalize_kernel, dim3((unsigned)(C[l + 1] / 16)), dim3(1024), s, (const float*)w.part, C[l + 1], hdr, bn_w[l],
           bn_b[l], running_mean[l], running_var[l], training, momentum, eps, w.coef[l]);
  };
  finalize(0);
  for (int l = 1; l < 5; ++l) {
    const int CI = C[l], CO = C[l + 1];
    const unsigned short* yin = w.y[l - 1];
    const float* cin = w.coef[l - 1];
    const unsigned short* wb = w.wb[l];
    const bool fuse_pool = l == 4 && N >= kRows;  // (shorter parts: a tile could touch three of them)
    float4* none = nullptr;
    if (CI == 64 && CO == 64)
      launch(pb_fwd_kernel<64, 64, false>, dim3(tiles), dim3(kT), s, yin, cin, wb, CO, w.y[l], w.part, hdr, (int)N, none);
    else if (CI == 64)
      launch(pb_fwd_kernel<64, 128, false>, dim3(tiles), dim3(kT), s, yin, cin, wb, CO, w.y[l], w.part, hdr, (int)N, none);
    else if (CO == 64 && fuse_pool)
      launch(pb_fwd_kernel<128, 64, true>, dim3(tiles), dim3(kT), s, yin, cin, wb, CO, w.y[l], w.part, hdr, (int)N, w.pool);
    else if (CO == 64)
      launch(pb_fwd_kernel<128, 64, false>, dim3(tiles), dim3(kT), s, yin, cin, wb, CO, w.y[l], w.part, hdr, (int)N, none);
    else if (fuse_pool)
      launch(pb_fwd_kernel<128, 128, true>, dim3(tiles), dim3(kT), s, yin, cin, wb, CO, w.y[l], w.part, hdr, (int)N, w.pool);
    else
      launch(pb_fwd_kernel<128, 128, false>, dim3(tiles), dim3(kT), s, yin, cin, wb, CO, w.y[l], w.part, hdr, (int)N, none);
    finalize(l);
  }
  if (N >= kRows)
    launch(pb_pool_merge_kernel, dim3((unsigned)M), dim3((unsigned)F), s, (const float4*)w.pool, (const float*)w.coef[4],
           (int)N, (int)F, (const int*)w.rank, feat, w.arg);
  else
    launch(pb_pool_kernel, dim3((unsigned)M), dim3(kT), s, (const unsigned short*)w.y[4], (const float*)w.coef[4], (int)N,
           (int)F, (const int*)w.rank, feat, w.arg);
  return mpa::check_launch("pointnet_forward_bf16");
}

extern "C" int mpa_pointnet_backward_bf16(const float* grad_feat, const float* points, const float* valids,
                                          const float* const* conv_w, const float* const* bn_w, int64_t M, int64_t N,
                                          int64_t F, void* ws, float* const* grad_conv_w, float* const* grad_bn_w,
                                          float* const* grad_bn_b, void* stream) {
  (void)valids;
  (void)conv_w;
  (void)bn_w;
  if (int st = check_dims(M, N, F, "pointnet_backward_bf16")) return st;
  if (M == 0) return MPA_OK;
  MPA_REQUIRE(grad_feat && points && ws && grad_conv_w && grad_bn_w && grad_bn_b, "pointnet_backward_bf16: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const Ws w = carve(ws, M, N, F, nullptr);
  const int64_t R = M * N;
  const unsigned tiles = (unsigned)((R + kRows - 1) / kRows);
  const int* hdr = w.hdr;
  const TopSrc top{grad_feat, w.vlist, w.arg, w.bitmap, (int)N, 1.0f / (float)N};
  const TopSrc none{nullptr, nullptr, nullptr, nullptr, (int)N, 0.0f};
  mpa::zero_words_async(w.bitmap, R, s);
  launch(pb_top_mark_kernel, dim3((unsigned)M), dim3((unsigned)F), s, grad_feat, (const int*)w.vlist, (const int*)w.arg,
         (const unsigned short*)w.y[4], (const float*)w.coef[4], (int)N, (int)F, hdr, w.part, w.bitmap);
  launch(pb_bwd_coef_kernel, dim3((unsigned)(F / 16)), dim3(1024), s, (const float*)w.part, (int)F, hdr, 1,
         (const float*)w.coef[4], grad_bn_w[4], grad_bn_b[4], w.bc[4]);
#define PB_BWD(CI, CO, TOP, l)                                                                                            \
  {                                                                                                                       \
    launch(pb_wgrad_kernel<CI, CO, TOP>, dim3(kChunks), dim3((wgrad_threads<CI, CO>())), s, (const unsigned short*)(TOP ? nullptr : w.g[l]),    \
           (const unsigned short*)w.y[l], TOP ? top : none, (const float*)w.bc[l], (const unsigned short*)w.y[l - 1],      \
           (const float*)w.coef[l - 1], w.partw, hdr);                                                                \
    dg::launch_tn_reduce(w.partw, kChunks, (long long)CO * CI, grad_conv_w[l], s);                                         \
    launch(pb_dgrad_kernel<CI, CO, TOP>, dim3(tiles), dim3(kT), s, (const unsigned short*)(TOP ? nullptr : w.g[l]),        \
           (const unsigned short*)w.y[l], TOP ? top : none, (const float*)w.bc[l], (const unsigned short*)w.wt[l],         \
           (const unsigned short*)w.y[l - 1], (const float*)w.coef[l - 1], w.g[l - 1], w.part, hdr);                       \
    launch(pb_bwd_coef_kernel, dim3((unsigned)(CI / 16)), dim3(1024), s, (const float*)w.part, CI, hdr, 0,                 \
           (const float*)w.coef[l - 1], grad_bn_w[l - 1], grad_bn_b[l - 1], w.bc[l - 1]);                                  \
  }
  if (F == 256) PB_BWD(128, 256, true, 4)
  else if (F == 128) PB_BWD(128, 128, true, 4)
  else PB_BWD(128, 64, true, 4)
  PB_BWD(64, 128, false, 3)
  PB_BWD(64, 64, false, 2)
  PB_BWD(64, 64, false, 1)
#undef PB_BWD
  launch(pb_first_wgrad_kernel, dim3(tiles), dim3(kT), s, (const unsigned short*)w.g[0], (const unsigned short*)w.y[0],
         (const float*)w.bc[0], points, (const int*)w.vlist, (int)N, w.partw, hdr);
  dg::launch_tn_reduce(w.partw, (int)tiles, 192LL, grad_conv_w[0], s);
  return mpa::check_launch("pointnet_backward_bf16");
}
