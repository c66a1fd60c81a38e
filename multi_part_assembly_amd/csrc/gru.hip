// Recurrent half of a (bi)directional single-layer GRU over short sequences — forward and backward through time — for
// gfx950: replaces the per-step cuDNN/MIOpen launches behind
//   RNNWrapper(nn.GRU(bidirectional))   multi_part_assembly/models/modules/rnn.py:6-46
//   RGLNet.forward                      multi_part_assembly/models/rgl_net/network.py:50-68,118-127
// (B = 32 shapes, T = P = 20 parts, hidden 2F = 256, three GRUs per training step, two directions each: MIOpen runs it
// as ~2500 tiny launches per step, which makes the whole RGL-NET step host-bound).
//
// The input projections  gi[t] = W_ih x_t + b_ih  of all steps are ONE GEMM outside (library op, autograd handles it);
// this file does what is sequential:
//   r = sigmoid(gi_r + W_hr h + b_hr),  z = sigmoid(gi_z + W_hz h + b_hz),  n = tanh(gi_n + r * (W_hn h + b_hn)),
//   h' = (1 - z) * n + z * h                                                     (torch.nn.GRU's equations and gate order)
// ONE launch per pass for both directions: grid = (H / 8, directions).  A block owns 8 hidden units (A/B: 16 units per
// block 0.33 + 0.73 ms forward + backward, 8 units 0.28 + 0.56, 4 units 0.38 + 0.70) — its 24 rows of
// W_hh stay in LDS for the whole sequence — and the blocks of a direction hand each other the new hidden state once per
// step as tagged 8-byte words in global memory (tagged_store below: no grid barrier).  Backward through time keeps its
// 24 x H slice of dW_hh in registers, hands the partial dL/dh_{t-1} of its rows to the other blocks the same way and sums
// the partials in block order: no atomics on floats, bit-reproducible.  All blocks must be co-resident (2 * H/8 <= 256
// CUs): a consumer polls for words only a running producer can write.
#include <stdlib.h>

#include "common.h"

namespace {

#ifndef MPA_GRU_U
#define MPA_GRU_U 8
#endif
constexpr int kU = MPA_GRU_U;  // hidden units per block
constexpr int kGT = 256;     // threads per block
constexpr int kMaxH = 512;
constexpr int kMaxB = 64;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Exchange between the blocks of a direction, once per step, WITHOUT a grid barrier: every value travels as one
// naturally aligned 8-byte {value, tag} word written by a single device-scope store (tag = the step's number), and a
// consumer polls the word itself until the tag is the one it waits for — data and "ready" arrive in the same store, so
// no fence, no counter and no arrival skew sit between a producer and its consumers (a counter barrier with its release /
// acquire fences measured ~7 us per step, most of a step's time).  The exchange buffers are zeroed before the launch
// (tags start at 1); two buffers by step parity suffice: a block can only be one step ahead of the slowest reader of
// its previous values.  All blocks of the grid must be co-resident (checked by gru_resident).
typedef unsigned long long tagged_t;
constexpr int kPollBudget = 1 << 22;
__device__ __forceinline__ void tagged_store(tagged_t* p, float v, unsigned tag) {
  __hip_atomic_store(p, ((tagged_t)tag << 32) | (tagged_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ tagged_t tagged_peek(const tagged_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the value of *p once it carries `tag` (first poll result given).  A producer that never shows up (a grid that is not
// co-resident after all — e.g. another stream's kernels hold the CUs the missing blocks need) exhausts the wait's poll
// budget after a few seconds.  Then the launch gives up CLEANLY: the waiting thread raises the launch's status word (device
// memory of the caller), every other wait of every block sees it within 1024 polls and stops too, the kernel runs to its
// end on whatever values it has, and the host side turns the status word into an error (gru.py) — no trap (which
// takes the process's HIP context with it), no hang.  Without a status word (NULL) the kernel traps as before.
struct Poll {
  int budget, limit;
  int* status;
  bool dead;
};
__device__ __forceinline__ float tagged_wait(const tagged_t* p, tagged_t first, unsigned tag, Poll& pl) {
  tagged_t v = first;
  while ((unsigned)(v >> 32) != tag && !pl.dead) {
    --pl.budget;
    if (pl.budget < 0 || ((pl.budget & 1023) == 0 && pl.status != nullptr &&
                          __hip_atomic_load(pl.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      if (pl.status == nullptr) __builtin_trap();
      __hip_atomic_store(pl.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pl.dead = true;
      break;
    }
    __builtin_amdgcn_s_sleep(2);
    v = tagged_peek(p);
  }
  pl.budget = pl.limit;  // the budget bounds ONE wait (a few seconds), not the launch's total polling
  return __uint_as_float((unsigned)v);
}
// PF words p[q * stride] (q < n valid): wait for the first one alone (ONE polled word per thread while the producers
// are still busy: the pollers' traffic delays the very stores they wait for — re-loading whole batches until every tag
// had arrived was 4x slower), then load the others together, and wait singly for a straggler.
template <int PF>
__device__ __forceinline__ void tagged_wait_all(const tagged_t* p, long long stride, int n, unsigned tag, Poll& budget,
                                                float (&out)[PF]) {
  out[0] = tagged_wait(p, tagged_peek(p), tag, budget);
  tagged_t v[PF];
#pragma unroll
  for (int q = 1; q < PF; ++q) v[q] = tagged_peek(p + (q < n ? q : 0) * stride);
#pragma unroll
  for (int q = 1; q < PF; ++q) out[q] = q < n ? tagged_wait(p + q * stride, v[q], tag, budget) : 0.0f;
}

// gi [D][B][T][3H], h0 [D][B][H], whh [D][3H][H], bhh [D][3H] -> out [D][B][T][H]; saved [D][B][T][4][H] = r, z, n, hn.
template <int H>
__global__ __launch_bounds__(kGT) void gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ h0,
                                                      const float* __restrict__ whh, const float* __restrict__ bhh,
                                                      int B, int T, float* __restrict__ out,
                                                      float* __restrict__ saved, tagged_t* __restrict__ xch,
                                                      int* __restrict__ status, int poll_limit) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = H + 4;              // rows padded by one 16-byte slot: the rows a ds_read_b128 lane group touches
  float* W = smem;                       // (4 samples, 8 units) start 4 banks apart  ([3 * kU][LD]: rows r(u0..), z(..), n(..))
  float* hp = smem + 3 * kU * LD;        // [B][LD] previous hidden state
  const int d = blockIdx.y, u0 = blockIdx.x * kU, nblk = gridDim.x;
  const float* wd = whh + (long long)d * 3 * H * H;
  for (int e = threadIdx.x; e < 3 * kU * H; e += kGT) {
    const int row = e / H, k = e % H, gate = row / kU, u = row % kU;
    W[row * LD + k] = wd[(long long)(gate * H + u0 + u) * H + k];
  }
  const float* gid = gi + (long long)d * B * T * 3 * H;
  float* od = out + (long long)d * B * T * H;
  float* sd = saved + (long long)d * B * T * 4 * H;
  tagged_t* xd = xch + (long long)d * 2 * B * H;  // [2 (step parity)][B][H]
  Poll budget{poll_limit, poll_limit, status, false};
  // the own outputs' biases (constant) and, per step, their input projections: requested BEFORE the wait for the other
  // blocks' hidden state, on which they do not depend (one exposed L2 round trip less per step)
  constexpr int OPT = (kMaxB * kU + kGT - 1) / kGT;
  const float* bh = bhh + (long long)d * 3 * H;
  float bhr[OPT], bhz[OPT], bhn[OPT];
#pragma unroll
  for (int i = 0; i < OPT; ++i) {
    const int e = threadIdx.x + i * kGT, c = u0 + (e < B * kU ? e % kU : 0);
    bhr[i] = bh[c], bhz[i] = bh[H + c], bhn[i] = bh[2 * H + c];
  }
  for (int t = 0; t < T; ++t) {
    float gr[OPT], gz[OPT], gn[OPT];
#pragma unroll
    for (int i = 0; i < OPT; ++i) {
      const int e = threadIdx.x + i * kGT, ec = e < B * kU ? e : 0;
      const float* g = gid + ((long long)(ec / kU) * T + t) * 3 * H + u0 + ec % kU;
      gr[i] = g[0], gz[i] = g[H], gn[i] = g[2 * H];
    }
    // previous hidden state of ALL units: h0, or the tagged words the blocks of this direction published in step t - 1
    if (t == 0) {
      for (int e = threadIdx.x; e < B * (H / 4); e += kGT) {
        const int b = e / (H / 4), k = 4 * (e % (H / 4));
        *reinterpret_cast<float4*>(&hp[b * LD + k]) = *reinterpret_cast<const float4*>(h0 + ((long long)d * B + b) * H + k);
      }
    } else {
      const tagged_t* xr = xd + (long long)((t - 1) & 1) * B * H;
      constexpr int PF = 32;  // words per thread and batch (B x H = 32 x 256: all of a thread's words at once)
      for (int e0 = threadIdx.x; e0 < B * H; e0 += PF * kGT) {
        float hv[PF];
        const int n = (B * H - e0 + kGT - 1) / kGT;
        tagged_wait_all<PF>(xr + e0, kGT, n < PF ? n : PF, (unsigned)t, budget, hv);
#pragma unroll
        for (int q = 0; q < PF; ++q) {
          const int e = e0 + q * kGT;
          if (e < B * H) hp[(e / H) * LD + e % H] = hv[q];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < OPT; ++i) {  // (sample, own unit)
      const int e = threadIdx.x + i * kGT;
      if (e >= B * kU) break;
      const int b = e / kU, u = e % kU;
      const float* hb = hp + b * LD;
      const float *wr = W + (0 * kU + u) * LD, *wz = W + (1 * kU + u) * LD, *wn = W + (2 * kU + u) * LD;
      float ar = 0.0f, az = 0.0f, an = 0.0f;
#pragma unroll 4
      for (int k = 0; k < H; k += 4) {  // 16-byte LDS reads; the three chains run over k in ascending order
        const float4 h4 = *reinterpret_cast<const float4*>(hb + k);
        const float4 r4 = *reinterpret_cast<const float4*>(wr + k);
        const float4 z4 = *reinterpret_cast<const float4*>(wz + k);
        const float4 n4 = *reinterpret_cast<const float4*>(wn + k);
        ar = __builtin_fmaf(h4.x, r4.x, ar);
        az = __builtin_fmaf(h4.x, z4.x, az);
        an = __builtin_fmaf(h4.x, n4.x, an);
        ar = __builtin_fmaf(h4.y, r4.y, ar);
        az = __builtin_fmaf(h4.y, z4.y, az);
        an = __builtin_fmaf(h4.y, n4.y, an);
        ar = __builtin_fmaf(h4.z, r4.z, ar);
        az = __builtin_fmaf(h4.z, z4.z, az);
        an = __builtin_fmaf(h4.z, n4.z, an);
        ar = __builtin_fmaf(h4.w, r4.w, ar);
        az = __builtin_fmaf(h4.w, z4.w, az);
        an = __builtin_fmaf(h4.w, n4.w, an);
      }
      const int c = u0 + u;
      const float r = sigmoidf_(gr[i] + ar + bhr[i]);
      const float z = sigmoidf_(gz[i] + az + bhz[i]);
      const float hn = an + bhn[i];
      const float n = tanhf(gn[i] + r * hn);
      const float hnew = (1.0f - z) * n + z * hb[c];
      od[((long long)b * T + t) * H + c] = hnew;
      tagged_store(xd + (long long)(t & 1) * B * H + b * H + c, hnew, (unsigned)(t + 1));
      float* sv = sd + ((long long)b * T + t) * 4 * H;
      sv[c] = r;
      sv[H + c] = z;
      sv[2 * H + c] = n;
      sv[3 * H + c] = hn;
    }
    __syncthreads();  // hp is rewritten by the next step
  }
}

// backward through time.  gout [D][B][T][H] -> dgi [D][B][T][3H], dwhh [D][3H][H], dbhh [D][3H];
// part [D][2][nblk][B][H]: per-step partial dL/dh_{t-1} of every block's rows (double-buffered by step parity).
template <int H>
__global__ __launch_bounds__(kGT) void gru_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ h0,
                                                      const float* __restrict__ whh, const float* __restrict__ out,
                                                      const float* __restrict__ saved, int B, int T,
                                                      float* __restrict__ dgi, float* __restrict__ dwhh,
                                                      float* __restrict__ dbhh, tagged_t* __restrict__ part,
                                                      int* __restrict__ status, int poll_limit) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W = smem;                        // [3 * kU][H]
  float* hp = smem + 3 * kU * H;          // [B][H] h_{t-1}
  float* dg = hp + B * H;                 // [B][3 * kU] gate gradients of the own rows at this step (r, z, n-hidden)
  float* dhc = dg + B * 3 * kU;           // [B][kU] carried dL/dh of the own units
  const int d = blockIdx.y, u0 = blockIdx.x * kU, nblk = gridDim.x, blk = blockIdx.x;
  const float* wd = whh + (long long)d * 3 * H * H;
  for (int e = threadIdx.x; e < 3 * kU * H; e += kGT) {
    const int row = e / H, k = e % H, gate = row / kU, u = row % kU;
    W[e] = wd[(long long)(gate * H + u0 + u) * H + k];
  }
  for (int e = threadIdx.x; e < B * kU; e += kGT) dhc[e] = 0.0f;
  const float* god = gout + (long long)d * B * T * H;
  const float* od = out + (long long)d * B * T * H;
  const float* sd = saved + (long long)d * B * T * 4 * H;
  float* dgid = dgi + (long long)d * B * T * 3 * H;
  tagged_t* pd = part + (long long)d * 2 * nblk * B * H;  // tagged words, see tagged_store
  Poll budget{poll_limit, poll_limit, status, false};
  // own slice of dW_hh (3 * kU rows x H columns) and of W_hh, column-wise in registers: thread -> column(s) k0 + 256 c of
  // the rows rbase .. rbase + RPT - 1, so that per sample one h value and RPT/4 broadcast 16-byte reads of the gate
  // gradients feed RPT FMAs (a row-major spread cost two LDS reads per FMA)
  constexpr int CPT = H >= kGT ? H / kGT : 1;          // columns per thread
  constexpr int RG = H >= kGT ? 1 : kGT / H;           // row groups (H < 256: the threads split the rows)
  constexpr int RPT = 3 * kU / RG;                     // rows per thread
  const int k0 = threadIdx.x % H, rbase = (threadIdx.x / H) * RPT;
  float dw[RPT][CPT], wreg[RPT][CPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int c = 0; c < CPT; ++c) dw[r][c] = 0.0f;
  float db_acc = 0.0f;  // threads 0 .. 3*kU-1: bias gradient of row threadIdx.x
  __shared__ float red[kGT];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int c = 0; c < CPT; ++c) wreg[r][c] = W[(rbase + r) * H + k0 + kGT * c];
  for (int t = T - 1; t >= 0; --t) {
    for (int e = threadIdx.x; e < B * (H / 4); e += kGT) {
      const int b = e / (H / 4), k = 4 * (e % (H / 4));
      const float* src = t == 0 ? h0 + ((long long)d * B + b) * H + k : od + ((long long)b * T + (t - 1)) * H + k;
      *reinterpret_cast<float4*>(&hp[b * H + k]) = *reinterpret_cast<const float4*>(src);
    }
    __syncthreads();
    // gate gradients of the own units
    for (int e = threadIdx.x; e < B * kU; e += kGT) {
      const int b = e / kU, u = e % kU, c = u0 + u;
      float dh = god[((long long)b * T + t) * H + c] + dhc[e];
      if (t < T - 1) {  // + what the other rows sent back through W_hh in step t + 1 (fixed block order)
        const tagged_t* pp = pd + (long long)((t + 1) & 1) * nblk * B * H + (long long)b * H + c;
        const unsigned tag = (unsigned)(T - (t + 1));
        constexpr int PF = 32;  // words per batch (all blocks of H = 256 at once); the additions in block order
        for (int k0 = 0; k0 < nblk; k0 += PF) {
          float pv[PF];
          const int n = nblk - k0 < PF ? nblk - k0 : PF;
          tagged_wait_all<PF>(pp + (long long)k0 * B * H, (long long)B * H, n, tag, budget, pv);
#pragma unroll
          for (int q = 0; q < PF; ++q)
            if (q < n) dh += pv[q];
        }
      }
      const float* sv = sd + ((long long)b * T + t) * 4 * H;
      const float r = sv[c], z = sv[H + c], n = sv[2 * H + c], hn = sv[3 * H + c];
      const float hprev = hp[b * H + c];
      const float dn = dh * (1.0f - z), dz = dh * (hprev - n);
      const float dn_pre = dn * (1.0f - n * n);
      const float dr_pre = dn_pre * hn * r * (1.0f - r);
      const float dz_pre = dz * z * (1.0f - z);
      dhc[e] = dh * z;  // the direct path to h_{t-1}
      float* go = dgid + ((long long)b * T + t) * 3 * H;
      go[c] = dr_pre;
      go[H + c] = dz_pre;
      go[2 * H + c] = dn_pre;
      dg[b * 3 * kU + u] = dr_pre;
      dg[b * 3 * kU + kU + u] = dz_pre;
      dg[b * 3 * kU + 2 * kU + u] = dn_pre * r;  // gradient w.r.t. (W_hn h + b_hn)
    }
    __syncthreads();
    // dW_hh[row][k] += sum_b dg[b][row] * h_{t-1}[b][k];  db_hh[row] += sum_b dg[b][row]
    for (int b = 0; b < B; ++b) {
      float hk[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) hk[c] = hp[b * H + k0 + kGT * c];
#pragma unroll
      for (int r4 = 0; r4 < RPT / 4; ++r4) {
        const float4 g4 = *reinterpret_cast<const float4*>(&dg[b * 3 * kU + rbase + 4 * r4]);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < CPT; ++c) dw[4 * r4 + u][c] = __builtin_fmaf(gv[u], hk[c], dw[4 * r4 + u][c]);
      }
    }
    if (threadIdx.x < 3 * kU) {
      float a = db_acc;
      for (int b = 0; b < B; ++b) a += dg[b * 3 * kU + threadIdx.x];
      db_acc = a;
    }
    // partial dL/dh_{t-1}[b][k] = sum over the rows of dg[b][row] * W[row][k]: the thread's rows of its column(s);
    // H < 256: the row groups of a column are added in group order through LDS (fixed order)
    tagged_t* po = pd + (long long)(t & 1) * nblk * B * H + (long long)blk * B * H;
    const unsigned otag = (unsigned)(T - t);
    for (int b = 0; b < B; ++b) {
      float a[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) a[c] = 0.0f;
#pragma unroll
      for (int r4 = 0; r4 < RPT / 4; ++r4) {
        const float4 g4 = *reinterpret_cast<const float4*>(&dg[b * 3 * kU + rbase + 4 * r4]);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int c = 0; c < CPT; ++c) a[c] = __builtin_fmaf(gv[u], wreg[4 * r4 + u][c], a[c]);
      }
      if constexpr (RG == 1) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) tagged_store(po + (long long)b * H + k0 + kGT * c, a[c], otag);
      } else {
        red[threadIdx.x] = a[0];
        __syncthreads();
        if (threadIdx.x < H) {
          float sum = 0.0f;
#pragma unroll
          for (int g = 0; g < RG; ++g) sum += red[g * H + threadIdx.x];
          tagged_store(po + (long long)b * H + threadIdx.x, sum, otag);
        }
        __syncthreads();
      }
    }
    __syncthreads();  // hp, dg and dhc are rewritten by the next step
  }
  float* dwd = dwhh + (long long)d * 3 * H * H;
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int row = rbase + r, gate = row / kU, u = row % kU;
#pragma unroll
    for (int c = 0; c < CPT; ++c) dwd[(long long)(gate * H + u0 + u) * H + k0 + kGT * c] = dw[r][c];
  }
  if (threadIdx.x < 3 * kU) {
    const int gate = threadIdx.x / kU, u = threadIdx.x % kU;
    dbhh[(long long)d * 3 * H + gate * H + u0 + u] = db_acc;
  }
}

int gru_check(int64_t D, int64_t B, int64_t T, int64_t H, const char* who) {
  MPA_REQUIRE(D == 1 || D == 2, "%s: 1 or 2 directions", who);
  MPA_REQUIRE(B >= 1 && B <= kMaxB && T >= 1 && T <= 4096, "%s: 1 <= batch <= %d, 1 <= steps <= 4096", who, kMaxB);
  MPA_REQUIRE(H == 128 || H == 256, "%s: hidden size must be 128 or 256 (2 x pc_feat_dim of the shipped configs)", who);
  return MPA_OK;
}

}  // namespace

namespace {
// Every block polls for the words the other blocks of its direction publish in the same step: all H/8 x D blocks must be
// RESIDENT at once (a block that was never dispatched — e.g. on a partitioned device with fewer CUs, or with an LDS
// footprint that leaves one block per CU — would stall the others until their poll budget runs out).  Checked against the
// occupancy the runtime reports.
template <typename Kern>
int gru_resident(Kern kern, size_t smem, int blocks, const char* who) {
  if (smem > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
          hipSuccess)
    return mpa::fail(MPA_EINVAL, "%s: cannot reserve %zu bytes of LDS", who, smem);
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kGT, smem) != hipSuccess)
    return mpa::fail(MPA_EINVAL, "%s: cannot query the device's occupancy", who);
  if ((long long)per_cu * cus < blocks)
    return mpa::fail(MPA_EINVAL, "%s: the per-step exchange needs all %d blocks resident, this device holds %d x %d", who, blocks,
                     per_cu, cus);
  return MPA_OK;
}
}  // namespace

namespace {
// polls one wait may spend (~1 us each) before the launch gives up; MPA_GRU_POLL_BUDGET overrides (tests shorten it)
int poll_limit() {
  if (const char* e = getenv("MPA_GRU_POLL_BUDGET")) {
    const long v = strtol(e, nullptr, 10);
    if (v >= 1024 && v <= (1L << 30)) return (int)v;
  }
  return kPollBudget;
}

// test support: `blocks` blocks that each hold `lds_bytes` of LDS and spin for `usec` microseconds of the constant
// 100 MHz clock — CUs that another stream cannot use meanwhile (tests/test_gru_gpu.py shows the GRU's co-residency failure
// as a Python error with it)
__global__ __launch_bounds__(64) void occupy_kernel(long long ticks, int* sink) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = (int)threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(32);
  if (lds[threadIdx.x] < 0) *sink = 1;
}
}  // namespace

extern "C" int mpa_debug_occupy(int64_t blocks, int64_t lds_bytes, int64_t usec, void* stream) {
  MPA_REQUIRE(blocks >= 1 && blocks <= 4096 && lds_bytes >= 256 && lds_bytes <= 160 * 1024 && usec >= 0 && usec <= 5000000,
              "debug_occupy: bad args");
  if (lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_bytes) != hipSuccess)
    return mpa::fail(MPA_EINVAL, "debug_occupy: cannot reserve %lld bytes of LDS", (long long)lds_bytes);
  hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)blocks), dim3(64), (size_t)lds_bytes, mpa::as_stream(stream),
                     (long long)usec * 100, (int*)nullptr);
  return mpa::check_launch("debug_occupy");
}

extern "C" int mpa_gru_resident(int64_t D, int64_t B, int64_t H, int* ok) {
  MPA_REQUIRE(ok != nullptr, "gru_resident: null pointer");
  *ok = 0;
  if (!((H == 128 || H == 256) && B >= 1 && B <= 64 && (D == 1 || D == 2))) return MPA_OK;
  const size_t fwd = sizeof(float) * (3 * kU * (H + 4) + B * (H + 4)), bwd = sizeof(float) * (3 * kU * H + B * H + B * 3 * kU + B * kU);
  if (fwd > 160 * 1024 || bwd > 160 * 1024) return MPA_OK;
  const int blocks = (int)(H / kU * D);
  int st = H == 128 ? gru_resident(gru_fwd_kernel<128>, fwd, blocks, "gru") : gru_resident(gru_fwd_kernel<256>, fwd, blocks, "gru");
  if (st == MPA_OK)
    st = H == 128 ? gru_resident(gru_bwd_kernel<128>, bwd, blocks, "gru") : gru_resident(gru_bwd_kernel<256>, bwd, blocks, "gru");
  *ok = st == MPA_OK;
  return MPA_OK;
}

extern "C" int mpa_gru_workspace(int64_t D, int64_t B, int64_t T, int64_t H, int64_t* float_elems) {
  if (int st = gru_check(D, B, T, H, "gru_workspace")) return st;
  MPA_REQUIRE(float_elems != nullptr, "gru_workspace: null pointer");
  // saved gates [D][B][T][4][H] | exchange words [D][2][H/kU][B][H] of 8 bytes (the backward's partials; the forward
  // uses the first [D][2][B][H] of them) | padding
  *float_elems = D * B * T * 4 * H + 2 * D * 2 * (H / kU) * B * H + 64;
  return MPA_OK;
}

extern "C" int mpa_gru_forward(const float* gi, const float* h0, const float* whh, const float* bhh, int64_t D, int64_t B,
                               int64_t T, int64_t H, float* ws, float* out, int32_t* status, void* stream) {
  if (int st = gru_check(D, B, T, H, "gru_forward")) return st;
  MPA_REQUIRE(gi && h0 && whh && bhh && ws && out, "gru_forward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  float* saved = ws;
  tagged_t* xch = reinterpret_cast<tagged_t*>(ws + D * B * T * 4 * H);
  MPA_REQUIRE((uintptr_t)xch % 8 == 0, "gru_forward: workspace must be 8-byte aligned");
  const size_t smem = sizeof(float) * (3 * kU * (H + 4) + B * (H + 4));
  MPA_REQUIRE(smem <= 160 * 1024, "gru_forward: batch x hidden size does not fit the 160 KB of LDS");
  // tags run 1..T in EVERY launch: a word left by the previous launch (or graph replay) already carries the tag a
  // consumer waits for, so the clear is load-bearing — and it is a kernel, not hipMemsetAsync (common.h).
  mpa::zero_words_async(xch, 2 * D * 2 * B * H, s);
  const dim3 grid((unsigned)(H / kU), (unsigned)D);
#define MPA_GRU_FWD(HH)                                                                                               \
  {                                                                                                                   \
    static size_t checked = 0; /* LDS request + co-residency of the whole grid, once per footprint */                  \
    if (smem != checked) {                                                                                            \
      if (int st = gru_resident(gru_fwd_kernel<HH>, smem, (int)(grid.x * grid.y), "gru_forward")) return st;          \
      checked = smem;                                                                                                 \
    }                                                                                                                 \
    hipLaunchKernelGGL(gru_fwd_kernel<HH>, grid, dim3(kGT), smem, s, gi, h0, whh, bhh, (int)B, (int)T, out, saved,     \
                       xch, (int*)status, poll_limit());                                                              \
  }
  if (H == 128) MPA_GRU_FWD(128) else MPA_GRU_FWD(256)
#undef MPA_GRU_FWD
  return mpa::check_launch("gru_forward");
}

extern "C" int mpa_gru_backward(const float* grad_out, const float* h0, const float* whh, const float* out, int64_t D,
                                int64_t B, int64_t T, int64_t H, float* ws, float* grad_gi, float* grad_whh,
                                float* grad_bhh, int32_t* status, void* stream) {
  if (int st = gru_check(D, B, T, H, "gru_backward")) return st;
  MPA_REQUIRE(grad_out && h0 && whh && out && ws && grad_gi && grad_whh && grad_bhh, "gru_backward: null pointer");
  hipStream_t s = mpa::as_stream(stream);
  const float* saved = ws;
  tagged_t* part = reinterpret_cast<tagged_t*>(ws + D * B * T * 4 * H);
  MPA_REQUIRE((uintptr_t)part % 8 == 0, "gru_backward: workspace must be 8-byte aligned");
  const size_t smem = sizeof(float) * (3 * kU * H + B * H + B * 3 * kU + B * kU);
  MPA_REQUIRE(smem <= 160 * 1024, "gru_backward: batch x hidden size does not fit the 160 KB of LDS");
  mpa::zero_words_async(part, 2 * D * 2 * (H / kU) * B * H, s);  // load-bearing, see gru_forward
  const dim3 grid((unsigned)(H / kU), (unsigned)D);
#define MPA_GRU_BWD(HH)                                                                                               \
  {                                                                                                                   \
    static size_t checked = 0;                                                                                        \
    if (smem != checked) {                                                                                            \
      if (int st = gru_resident(gru_bwd_kernel<HH>, smem, (int)(grid.x * grid.y), "gru_backward")) return st;         \
      checked = smem;                                                                                                 \
    }                                                                                                                 \
    hipLaunchKernelGGL(gru_bwd_kernel<HH>, grid, dim3(kGT), smem, s, grad_out, h0, whh, out, saved, (int)B, (int)T,    \
                       grad_gi, grad_whh, grad_bhh, part, (int*)status, poll_limit());                                \
  }
  if (H == 128) MPA_GRU_BWD(128) else MPA_GRU_BWD(256)
#undef MPA_GRU_BWD
  return mpa::check_launch("gru_backward");
}
