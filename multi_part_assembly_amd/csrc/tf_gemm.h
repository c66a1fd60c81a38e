// The small-M GEMM of the transformer layers and of the few-hundred-row MLP layers of the graph networks (gfx950):
//     C[M,N] = epilogue(prologue(A[M,K]) . W[N,K]^T + bias)
// on v_mfma_f32_32x32x2_f32 — one 32 x 32 output tile per block, the block's 8 waves split K.  Shared by
// transformer.hip and mlp.hip (both include it inside their own translation unit; nothing here has external linkage).
#pragma once

#include <hip/hip_runtime.h>

namespace tfg {

typedef float f32x16 __attribute__((ext_vector_type(16)));


// ---- dropout: keep-scale of element `idx` at dropout site `site` -------------------------------------------------
struct Drop {
  unsigned long long seed;
  float p;       // drop probability; 0 disables
  float scale;   // 1 / (1 - p)
  const unsigned long long* seed_dev;  // if set, the seed is read from device memory (HIP-graph replays)
};

// The seed of a graph replay lives in device memory.  Kernels resolve it ONCE, at their very top, with this: read inside
// drop_scale — behind its `p <= 0` test, in unrolled per-element code — every call became a vector load of its own followed by a
// full wait: sixteen dependent memory round trips in the LayerNorm-backward prologue of the d o GEMM (tools/isa_latency_scan.py).
__device__ __forceinline__ Drop resolve_seed(Drop d) {
  if (d.p > 0.0f && d.seed_dev != nullptr) d.seed = *d.seed_dev;  // (wave-uniform address)
  d.seed_dev = nullptr;
  return d;
}

__device__ __forceinline__ float drop_scale(const Drop d, unsigned site, unsigned long long idx) {
  if (d.p <= 0.0f) return 1.0f;
  const unsigned long long seed = d.seed_dev != nullptr ? *d.seed_dev : d.seed;  // wave-uniform scalar load
  unsigned long long x = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(site + 1) + idx;
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  const float u = (float)(unsigned)(x >> 40) * (1.0f / 16777216.0f);  // 24 random bits -> [0, 1)
  return u < d.p ? 0.0f : d.scale;
}

// Sum over the 32 lanes of a half-wave, returned to all of them: four DPP additions inside the rows of 16 and ONE exchange
// between the two rows (the five-step `__shfl_xor` butterfly is five dependent LDS permutes, ~100 cycles each — the fused
// LayerNorm prologues run four of those chains per row pair).
__device__ __forceinline__ float half_sum(float v) {
  auto add = [](float a, int bits) { return a + __int_as_float(bits); };
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v + __shfl_xor(v, 16, 64);
}

// ---- GEMM  C[M,N] = epi(A[M,K] . W[N,K]^T + bias) ---------------------------------------------------------------------
enum Epi {
  EPI_NONE = 0, EPI_RELU_DROP = 1, EPI_DROP_RESID = 2, EPI_LEAKY = 3, EPI_RELU_MASK = 4, EPI_LEAKY_MASK = 5,
  EPI_STATS = 6,  // C = acc + bias and stats[row tile][n] = (sum, sum of squares) over the tile's valid rows (fixed order):
                  // the BatchNorm statistics pass of an MLP layer (mlp.hip)
  EPI_BIAS_ACT = 7  // C = relu?(acc + bias) with the ReLU decided at run time (g.relu)
};

struct GemmArgs {
  const float* A;      // [M, K]
  const float* W;      // [N, K]  (WT: [K, N])
  const float* bias;   // [N] or null
  float* C;            // [M, N]
  int M, N, K;
  int lda;             // row stride of A (0: K)
  int ldw;             // row stride of W (0: K, or N with WT): a column block of a wider weight matrix
  int relu;            // EPI_BIAS_ACT
  float* stats;        // EPI_STATS: [ceil(M / 32)][N][2]
  unsigned* zero;      // nullable: 64 words cleared by block (0, 0) (the tickets of the caller's cooperative reductions)
  const float* resid;  // EPI_DROP_RESID: [M, N];  EPI_*_MASK: the saved activations [M, N]
  Drop drop;
  unsigned epi_site;
  // LNF (LayerNorm fused into the operand load, K = 256 only): A is the un-normalised x; the block normalises its 32
  // rows in registers; the blocks of column 0 also write what the standalone LayerNorm kernel would have written
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  float* ln_stats;   // [M][2] mean, rstd
  float* ln_h;       // [M, K] the normalised rows (the weight gradient's operand)
  float* ln_xcopy;   // nullable: x itself (the first layer keeps its input)
  // LNM == 2 (LayerNorm BACKWARD fused into the operand load, K = 256 only): A is d(LayerNorm output); the block
  // turns its 32 rows into d x = resid + rstd (dh gamma - mean(dh gamma) - xhat mean(dh gamma xhat)) in registers
  // (ln_stats is read here) and multiplies the dropout-masked d x; the blocks of column 0 also write what the
  // standalone kernel would have written: d x, its masked copy and the tile's dgamma / dbeta partial sums
  const float* ln_x;      // [M, K] the LayerNorm's input
  const float* ln_resid;  // nullable [M, K]: the gradient arriving over the residual connection
  float* ln_dx;           // [M, K]
  float* ln_dxdrop;       // nullable [M, K]: d x under the dropout mask of ln_site (null: the operand is d x itself)
  float* ln_part;         // [row tiles][2 K]: sum dh xhat | sum dh over the tile's rows
  unsigned ln_site;
  // SK (split-K over two blocks per output tile, gridDim.z = 2): each block multiplies one half of K, leaves its 32 x 32
  // partial tile in sk_buf and takes the tile's ticket; the second block to arrive adds the two halves (lower half first,
  // whichever arrived first) and runs the epilogue.  The half tiles cross XCDs as agent-scope atomic stores / loads ordered
  // by the ticket (coop_reduce.h's hand-over: a release FENCE per block is a write-back of the XCD's L2 and made this
  // kernel 3 x slower, LABBOOK 5.3 xvii; -DMPA_TF_SK_FENCE=1 builds that textbook form: the fallback should a part or a
  // compiler ever stop acknowledging sc1 stores at agent scope — tests/test_model_gpu.py holds the hand-over to bit-equal
  // results over hundreds of launches).  Tickets are zero between launches (reset after use).
  float* sk_buf;        // [tiles][2][1024]
  unsigned* sk_ticket;  // [tiles]
  int zero_n;           // words of `zero` to clear (0: 64)
};

// grid = (ceil(M/32), N/32), block = 8 waves: the block owns ONE 32 x 32 MFMA tile and the waves split K.
// Why so small a tile: M = B*P is a few hundred tokens, so the whole GEMM is ~0.3 GFLOP; the fp32 MFMA pipe of
// one CU retires 256 FLOP/clk, and only many small blocks (160..640 here) put all 256 CUs to work.
// K is walked in phases of kp <= 128 (34 KB of LDS per block, so four blocks share a CU and hide each other's
// latencies): the block copies the 32 x kp panels of A and W into LDS with fully
// coalesced 16-byte loads (a lane-per-row fragment load would touch 64 cache lines per instruction and thrash the
// 32 KB L1), the next phase's global loads are issued before the MFMAs of the current one, and inside a phase
// wave w / lane half h owns the k-run [(2w+h) kp/16, +kp/16) which it reads from LDS as float4.  The eight
// partial tiles meet in LDS (fixed order) and the epilogue writes 128-byte row segments.
// Requires N % 32 == 0 and K % 64 == 0.
// WT: the weight is given as W^T, i.e. [K, N] row-major (input gradients reuse the forward weights untransposed).
#ifndef MPA_TF_SK_FENCE
#define MPA_TF_SK_FENCE 0
#endif
constexpr int kGW = 8, kGT = kGW * 64, kKP = 128, kLD = kKP + 4;

__host__ __device__ inline int gemm_phase(int K) {
  if (K <= kKP) return K;
  for (int kp = kKP; kp > 64; kp -= 64)  // kKP, kKP - 64, ...
    if (K % kp == 0) return kp;
  return 64;
}

// NPH > 0: K is exactly NPH phases and ALL global loads of the block are issued up front (NPH * 16 B * 2 per
// thread in registers), so the block pays the L2/HBM latency once instead of once per phase; with NPH >= 3 the LDS
// panels are double-buffered (one barrier per phase).  NPH == 0: any K, loads one phase ahead.
// LNM: 0 plain operand, 1 LayerNorm forward in the operand load, 2 LayerNorm backward in the operand load.
template <int EPI, bool WT, int NPH, int LNM = 0, bool SK = false>
__global__ __launch_bounds__(kGT) void gemm_kernel(const GemmArgs g) {
  static_assert(!SK || (NPH > 0 && LNM == 0 && EPI != EPI_STATS), "split-K: whole phases, plain operands");
  const Drop drop = (EPI == EPI_RELU_DROP || EPI == EPI_DROP_RESID || LNM == 2) ? resolve_seed(g.drop) : g.drop;
#ifdef MPA_GEMM_STAMPS  // s_memtime stamps of one block of the FFN-up GEMM (EPI_RELU_DROP with the fused LayerNorm), printed
  constexpr bool kStamp = EPI == EPI_RELU_DROP && LNM == 1;
  unsigned long long ts[8];
  int nts = 0;
#define GEMM_STAMP() do { if (kStamp) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ts[nts++] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define GEMM_STAMP() do { } while (0)
#endif
  GEMM_STAMP();
  const int kz = SK ? (int)blockIdx.z * NPH : 0;  // first K phase of this block
  constexpr bool LNF = LNM == 1;
  static_assert(LNM == 0 || NPH == 2, "the fused LayerNorm needs the whole K = 2 x 128 row in registers");
  constexpr int kBuf = NPH >= 3 ? 2 : 1, kPanel = 2 * 32 * kLD;
  __shared__ __attribute__((aligned(16))) float lds[kBuf * kPanel];  // A panel | W panel; later the 8 partial tiles
  float* la = lds;
  float* lw = lds + 32 * kLD;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
  const int r0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int kp = SK ? kKP : gemm_phase(g.K), nph = g.K / kp, ld = kp + 4, q4 = kp / 4, cnt = 32 * q4;
  const int lda = g.lda > 0 ? g.lda : g.K;
  const int ldw = g.ldw > 0 ? g.ldw : (WT ? g.N : g.K);
  if (g.zero != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (int)threadIdx.x < (g.zero_n > 0 ? g.zero_n : 64))
    g.zero[threadIdx.x] = 0u;
  constexpr int kFI = 32 * (kKP / 4) / kGT;  // float4 per thread, operand and phase
  struct Stage {
    float4 a[kFI], w[kFI];
  };
  auto fetch = [&](int ph, Stage& st) {
#pragma unroll
    for (int i = 0; i < kFI; ++i) {
      const int idx = threadIdx.x + kGT * i;
      if (idx < cnt) {
        const int row = idx / q4, c4 = idx % q4;
        const int ar = r0 + row < g.M ? r0 + row : g.M - 1;
        st.a[i] = *reinterpret_cast<const float4*>(g.A + (long long)ar * lda + (kz + ph) * kp + 4 * c4);
        if constexpr (WT) {  // [kp, 32] slab of W^T: 8 lanes per 128-byte row
          st.w[i] = *reinterpret_cast<const float4*>(g.W + (long long)((kz + ph) * kp + (idx >> 3)) * ldw + n0 + 4 * (idx & 7));
        } else {
          st.w[i] = *reinterpret_cast<const float4*>(g.W + (long long)(n0 + row) * ldw + (kz + ph) * kp + 4 * c4);
        }
      }
    }
  };
  auto stash = [&](int ph, const Stage& st, int buf) {
#pragma unroll
    for (int i = 0; i < kFI; ++i) {
      const int idx = threadIdx.x + kGT * i;
      if (idx < cnt) {
        const int row = idx / q4, c4 = idx % q4;
        const float4 a = st.a[i];  // (a by-value copy: storing st.a[i] directly sends the staging arrays to scratch)
        *reinterpret_cast<float4*>(la + buf * kPanel + row * ld + 4 * c4) = a;
        if constexpr (WT) *reinterpret_cast<float4*>(lw + buf * kPanel + (idx >> 3) * 32 + 4 * (idx & 7)) = st.w[i];
        else *reinterpret_cast<float4*>(lw + buf * kPanel + row * ld + 4 * c4) = st.w[i];
      }
    }
  };
  const int kh = kp / (2 * kGW), kb = (wave * 2 + h) * kh, nv = kh / 4;
  f32x16 acc = {0};
  auto compute = [&](int buf) {
    const float* pa = la + buf * kPanel + j * ld + kb;
    const float* pw = lw + buf * kPanel + (WT ? kb * 32 + j : j * ld + kb);
    for (int v = 0; v < nv; ++v) {
      const float4 a = *reinterpret_cast<const float4*>(pa + 4 * v);
      float4 b;
      if constexpr (WT) b = make_float4(pw[128 * v], pw[128 * v + 32], pw[128 * v + 64], pw[128 * v + 96]);  // [k][n]
      else b = *reinterpret_cast<const float4*>(pw + 4 * v);                                                 // [n][k]
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  };
  if constexpr (NPH > 0) {
    Stage st[NPH];
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) fetch(ph, st[ph]);
    GEMM_STAMP();
    if constexpr (LNF) {
      // thread t holds, per phase, columns 4 (t % 32) .. + 3 of rows t / 32 and 16 + t / 32: a row lives in one half-wave
      const int c4 = threadIdx.x & 31;
      float4 gm[NPH], bt[NPH];
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        gm[ph] = *reinterpret_cast<const float4*>(g.ln_gamma + ph * kKP + 4 * c4);
        bt[ph] = *reinterpret_cast<const float4*>(g.ln_beta + ph * kKP + 4 * c4);
      }
#pragma unroll
      for (int i = 0; i < kFI; ++i) {
        float sum = 0.0f;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) sum += (st[ph].a[i].x + st[ph].a[i].y) + (st[ph].a[i].z + st[ph].a[i].w);
        sum = half_sum(sum);
        const float mean = sum / (float)(NPH * kKP);
        float var = 0.0f;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
          const float dx = st[ph].a[i].x - mean, dy = st[ph].a[i].y - mean, dz = st[ph].a[i].z - mean,
                      dw = st[ph].a[i].w - mean;
          var += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        var = half_sum(var);
        const float rstd = 1.0f / __builtin_sqrtf(var / (float)(NPH * kKP) + g.ln_eps);
        const int row = r0 + (threadIdx.x >> 5) + 16 * i;
        const bool save = blockIdx.y == 0 && row < g.M;
        if (save && c4 == 0) {
          g.ln_stats[2 * row] = mean;
          g.ln_stats[2 * row + 1] = rstd;
        }
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
          const float4 x = st[ph].a[i];
          float4 hn;
          hn.x = (x.x - mean) * rstd * gm[ph].x + bt[ph].x;
          hn.y = (x.y - mean) * rstd * gm[ph].y + bt[ph].y;
          hn.z = (x.z - mean) * rstd * gm[ph].z + bt[ph].z;
          hn.w = (x.w - mean) * rstd * gm[ph].w + bt[ph].w;
          st[ph].a[i] = hn;
          if (save) {
            const long long o = (long long)row * g.K + ph * kKP + 4 * c4;
            *reinterpret_cast<float4*>(g.ln_h + o) = hn;
            if (g.ln_xcopy != nullptr) *reinterpret_cast<float4*>(g.ln_xcopy + o) = x;
          }
        }
      }
    }
    if constexpr (LNM == 2) {
      // same ownership as above: thread t holds columns 4 (t % 32) .. + 3 of rows t / 32 and 16 + t / 32 per phase
      const int c4 = threadIdx.x & 31;
      const bool col0 = blockIdx.y == 0;
      float4 gm[NPH], pxh[NPH], pg[NPH];  // gamma; this thread's column partials of dh xhat and dh
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        gm[ph] = *reinterpret_cast<const float4*>(g.ln_gamma + ph * kKP + 4 * c4);
        pxh[ph] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        pg[ph] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
      // every operand of the prologue is requested here, in one round trip (read where they are used — the residual behind
      // its null test, inside the loops — they were a dozen dependent waits in front of the block's MFMA chain)
      float4 lx[kFI][NPH], lr[kFI][NPH];
      float lmean[kFI], lrstd[kFI];
      const float* rsrc = g.ln_resid != nullptr ? g.ln_resid : g.ln_x;  // (no residual: a valid address, the value is dropped)
#pragma unroll
      for (int i = 0; i < kFI; ++i) {
        const int row = r0 + (threadIdx.x >> 5) + 16 * i, rr = row < g.M ? row : g.M - 1;
        const long long o = (long long)rr * g.K + 4 * c4;
        lmean[i] = g.ln_stats[2 * rr];
        lrstd[i] = g.ln_stats[2 * rr + 1];
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
          lx[i][ph] = *reinterpret_cast<const float4*>(g.ln_x + o + ph * kKP);
          lr[i][ph] = *reinterpret_cast<const float4*>(rsrc + o + ph * kKP);
        }
      }
#pragma unroll
      for (int i = 0; i < kFI; ++i) {
        const int row = r0 + (threadIdx.x >> 5) + 16 * i, rr = row < g.M ? row : g.M - 1;
        const float mean = lmean[i], rstd = lrstd[i];
        const long long o = (long long)rr * g.K + 4 * c4;
        float4 xh[NPH];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
          const float4 x = lx[i][ph];
          const float4 dh = st[ph].a[i];
          xh[ph] = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
          const float dx = dh.x * gm[ph].x, dy = dh.y * gm[ph].y, dz = dh.z * gm[ph].z, dw = dh.w * gm[ph].w;
          s1 += (dx + dy) + (dz + dw);
          s2 += (dx * xh[ph].x + dy * xh[ph].y) + (dz * xh[ph].z + dw * xh[ph].w);
        }
        s1 = half_sum(s1);
        s2 = half_sum(s2);
        s1 /= (float)(NPH * kKP);
        s2 /= (float)(NPH * kKP);
        const bool save = col0 && row < g.M;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
          const float4 dh = st[ph].a[i];
          float4 v = g.ln_resid != nullptr ? lr[i][ph] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          v.x += rstd * (dh.x * gm[ph].x - s1 - xh[ph].x * s2);
          v.y += rstd * (dh.y * gm[ph].y - s1 - xh[ph].y * s2);
          v.z += rstd * (dh.z * gm[ph].z - s1 - xh[ph].z * s2);
          v.w += rstd * (dh.w * gm[ph].w - s1 - xh[ph].w * s2);
          float4 m = v;
          if (g.ln_dxdrop != nullptr) {
            const unsigned long long e = (unsigned long long)(o + ph * kKP);
            m.x *= drop_scale(drop, g.ln_site, e);
            m.y *= drop_scale(drop, g.ln_site, e + 1);
            m.z *= drop_scale(drop, g.ln_site, e + 2);
            m.w *= drop_scale(drop, g.ln_site, e + 3);
          }
          st[ph].a[i] = m;
          if (save) {
            *reinterpret_cast<float4*>(g.ln_dx + o + ph * kKP) = v;
            if (g.ln_dxdrop != nullptr) *reinterpret_cast<float4*>(g.ln_dxdrop + o + ph * kKP) = m;
            pxh[ph].x += dh.x * xh[ph].x;
            pxh[ph].y += dh.y * xh[ph].y;
            pxh[ph].z += dh.z * xh[ph].z;
            pxh[ph].w += dh.w * xh[ph].w;
            pg[ph].x += dh.x;
            pg[ph].y += dh.y;
            pg[ph].z += dh.z;
            pg[ph].w += dh.w;
          }
        }
      }
      if (col0) {  // (block-uniform) the 16 row groups meet in LDS in a fixed order: [group][2 K]
        constexpr int kW = 2 * NPH * kKP;
        static_assert(16 * kW <= kBuf * kPanel && kW == kGT, "the partial table fits the panels, one column per thread");
        float* pl = lds + (threadIdx.x >> 5) * kW + 4 * c4;
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
          *reinterpret_cast<float4*>(pl + ph * kKP) = pxh[ph];
          *reinterpret_cast<float4*>(pl + NPH * kKP + ph * kKP) = pg[ph];
        }
        __syncthreads();
        float t = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += lds[q * kW + threadIdx.x];
        g.ln_part[(long long)blockIdx.x * kW + threadIdx.x] = t;
        __syncthreads();
      }
    }
    GEMM_STAMP();
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
      stash(ph, st[ph], ph % kBuf);
      __syncthreads();
      compute(ph % kBuf);
      if (kBuf == 1) __syncthreads();  // double-buffered: the next stash goes to the other panel pair
    }
    if (kBuf == 2) __syncthreads();
  } else {
    Stage st;
    fetch(0, st);
    for (int ph = 0; ph < nph; ++ph) {
      stash(ph, st, 0);
      __syncthreads();
      if (ph + 1 < nph) fetch(ph + 1, st);  // in flight while this phase computes
      compute(0);
      __syncthreads();
    }
  }
  GEMM_STAMP();
  float(*red)[16][64] = reinterpret_cast<float(*)[16][64]>(lds);  // 32 KB, the panels are dead
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  // thread -> (row, column) of the 32 x 32 tile: 32 consecutive columns per half-wave
  const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const float bias = g.bias ? g.bias[n0 + col] : 0.0f;
  float sv[2] = {0.0f, 0.0f};  // EPI_STATS: this thread's two output values (0 for rows behind M)
  float tot[2];                // the tile's two values of this thread, summed over the waves
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int rl = it * 16 + rg;
    const int reg = (rl & 3) + 4 * (rl >> 3), src = ((rl >> 2) & 1) * 32 + col;
    float v = 0.0f;
#pragma unroll
    for (int w = 0; w < kGW; ++w) v += red[w][reg][src];
    tot[it] = v;
  }
  if constexpr (SK) {
    __shared__ int sk_last;
    const long long tile = (long long)blockIdx.x * gridDim.y + blockIdx.y;
    float* mine = g.sk_buf + (tile * 2 + blockIdx.z) * 1024;
#if MPA_TF_SK_FENCE  // the textbook form (A/B and fallback builds): plain stores, agent-scope release / acquire fences
    mine[threadIdx.x] = tot[0];
    mine[kGT + threadIdx.x] = tot[1];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#else
    __hip_atomic_store(mine + threadIdx.x, tot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + kGT + threadIdx.x, tot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // written through and acknowledged before the ticket is taken
#endif
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = atomicAdd(g.sk_ticket + tile, 1u);
      sk_last = old == 1u;
      if (old == 1u) g.sk_ticket[tile] = 0u;  // ready for the next launch
    }
    __syncthreads();
    if (!sk_last) return;
    const float* other = g.sk_buf + (tile * 2 + (1 - blockIdx.z)) * 1024;
#if MPA_TF_SK_FENCE
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float o0 = other[threadIdx.x], o1 = other[kGT + threadIdx.x];
#else
    const float o0 = __hip_atomic_load(other + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float o1 = __hip_atomic_load(other + kGT + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    tot[0] = blockIdx.z == 0 ? tot[0] + o0 : o0 + tot[0];  // lower half of K first, whoever finishes
    tot[1] = blockIdx.z == 0 ? tot[1] + o1 : o1 + tot[1];
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int rl = it * 16 + rg, orow = r0 + rl;
    if (orow >= g.M) continue;
    float v = tot[it] + bias;
    const long long o = (long long)orow * g.N + n0 + col;
    if constexpr (EPI == EPI_RELU_DROP) {
      v = __builtin_fmaxf(v, 0.0f) * drop_scale(drop, g.epi_site, (unsigned long long)o);
    } else if constexpr (EPI == EPI_DROP_RESID) {
      v = g.resid[o] + v * drop_scale(drop, g.epi_site, (unsigned long long)o);
    } else if constexpr (EPI == EPI_LEAKY) {
      v = v > 0.0f ? v : 0.2f * v;
    } else if constexpr (EPI == EPI_RELU_MASK) {
      // gradient through dropout(relu(z)) given the saved activations a = relu(z) * keep_scale
      v = g.resid[o] > 0.0f ? v * g.drop.scale : 0.0f;
    } else if constexpr (EPI == EPI_LEAKY_MASK) {
      v = g.resid[o] > 0.0f ? v : 0.2f * v;  // gradient through LeakyReLU(0.2), resid = saved activations
    } else if constexpr (EPI == EPI_BIAS_ACT) {
      v = g.relu ? __builtin_fmaxf(v, 0.0f) : v;
    } else if constexpr (EPI == EPI_STATS) {
      sv[it] = v;
    }
    g.C[o] = v;
  }
#ifdef MPA_GEMM_STAMPS
  GEMM_STAMP();
  if (kStamp && ((blockIdx.x == 5 && blockIdx.y == 7) || (blockIdx.x + blockIdx.y * gridDim.x) % 97 == 0) && threadIdx.x == 0)
    printf("gemm stamps block (%d, %d) start %llu (s_memtime ticks): loads %llu, prologue %llu, stash + chains %llu, reduce + epilogue %llu\n",
           (int)blockIdx.x, (int)blockIdx.y, ts[0] % 10000000ull, ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], ts[4] - ts[3]);
#endif
  if constexpr (EPI == EPI_STATS) {  // column sums of the 32 x 32 tile, rows in ascending order
    __syncthreads();                 // every partial tile has been read
    float(*tile)[33] = reinterpret_cast<float(*)[33]>(lds);
    tile[rg][col] = sv[0];
    tile[16 + rg][col] = sv[1];
    __syncthreads();
    if (threadIdx.x < 32) {
      float t0 = 0.0f, t1 = 0.0f;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const float t = tile[r][col];
        t0 += t;
        t1 = __builtin_fmaf(t, t, t1);
      }
      float* d = g.stats + ((long long)blockIdx.x * g.N + n0 + col) * 2;
      d[0] = t0;
      d[1] = t1;
    }
  }
}

template <int EPI, bool WT = false>
void launch_gemm(const GemmArgs& g, hipStream_t s) {
  const dim3 grid((g.M + 31) / 32, g.N / 32), block(kGT);
  const int kp = gemm_phase(g.K);
  switch (kp == kKP || g.K <= kKP ? g.K / kp : 0) {  // full-size phases only
    case 1: hipLaunchKernelGGL((gemm_kernel<EPI, WT, 1>), grid, block, 0, s, g); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<EPI, WT, 2>), grid, block, 0, s, g); break;
    case 4: hipLaunchKernelGGL((gemm_kernel<EPI, WT, 4>), grid, block, 0, s, g); break;
    case 6: hipLaunchKernelGGL((gemm_kernel<EPI, WT, 6>), grid, block, 0, s, g); break;
    case 8: hipLaunchKernelGGL((gemm_kernel<EPI, WT, 8>), grid, block, 0, s, g); break;
    default: hipLaunchKernelGGL((gemm_kernel<EPI, WT, 0>), grid, block, 0, s, g); break;
  }
}

// The same product with K split over two blocks per tile when the tiles alone cannot fill the chip (g.sk_buf set, at
// most kSkTiles tiles, K = 6 or 8 full phases): twice the blocks, half the phases each — a block's phases run one
// after the other (loads -> LDS -> MFMA chain), and with one block per CU nothing else hides them.
constexpr int kSkTiles = 256;
template <int EPI, bool WT = false>
void launch_gemm_sk(const GemmArgs& g, hipStream_t s) {
  const int tiles = ((g.M + 31) / 32) * (g.N / 32), nph = g.K % kKP == 0 ? g.K / kKP : 0;
  if (g.sk_buf == nullptr || tiles > kSkTiles || (nph != 6 && nph != 8)) return launch_gemm<EPI, WT>(g, s);
  const dim3 grid((g.M + 31) / 32, g.N / 32, 2), block(kGT);
  if (nph == 8) hipLaunchKernelGGL((gemm_kernel<EPI, WT, 4, 0, true>), grid, block, 0, s, g);
  else hipLaunchKernelGGL((gemm_kernel<EPI, WT, 3, 0, true>), grid, block, 0, s, g);
}

}  // namespace tfg
