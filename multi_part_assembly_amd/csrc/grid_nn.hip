// Exact nearest-neighbour search with spatial pruning for the whole-shape Chamfer term of the fused
// assembly loss (assembly_loss.hip).  Same results as the brute-force scan, bit for bit — distances
// computed with the pinned arithmetic of chamfer_core.h, lowest original index on ties — but each
// query only visits the targets that can matter.
//
// Per sample (both transformed shapes, <= 20 000 valid points each):
//   1. grid parameters (grid_params_from_boxes, evaluated by every sort block): bounding box of BOTH shapes' valid points -> one uniform grid per sample, <= 32768
//                    cells (~0.6 points of each shape per cell: finer beats coarser, most of all on clumpy artifact-like shapes), <= 64 cells per axis.  Covering the union
//                    means no query is ever outside the grid.
//   2. grid_sort   : counting sort of each shape, one 1024-thread block per (sample, shape) with both histograms
//                    in LDS (count, scan, scatter in one launch),
//                      role TARGET : by fine cell -> (x, y, z, original index) records + cell start offsets;
//                                    cells are x-fastest, so a row of cells is one contiguous run of records;
//                      role QUERY  : by SUPER-cell (kS^3 fine cells) + a prefix of 64-query batches and the
//                                    work list (work item -> super-cell).
//   3. grid_search : one wave per (super-cell, batch of <= 64 queries, 1 per lane).  All queries of the wave
//                    lie inside one known box (the super-cell), so the candidate set is wave-uniform:
//                      seed  — the super-cell grown by one fine cell per side;
//                      sweep — with B = the largest best-distance in the wave, every cell row (y, z) whose
//                              box distance to the super-cell box is below B is visited over exactly the
//                              x-interval of cells that are closer than B; B shrinks as the sweep goes.
//                    Every cell that intersects ANY lane's search ball is visited, so the search is exact.
//                    The rows of the seed / of a ring are mapped to LANES: every lane copies its row's record
//                    range into one LDS candidate list (all rows in flight at once) and the wave scans the list
//                    with broadcast reads; long single ranges use the scalar-operand scan (records in SGPRs via
//                    the scalar cache, 8 per chunk, one compare per chunk).  Candidates are not visited in index
//                    order, so the update rule is lexicographic: smaller d, then smaller original index — exactly
//                    the answer of the strict-`<` in-order scan.  The <= P padded parts' representatives are
//                    checked last.
//
// Rounding safety: a record may be binned one ulp across a cell face; cell-box distances are therefore
// shrunk by 1e-3 of a cell before they are compared with B, and B is inflated by 1e-5.
#include "assembly_internal.h"
#include "chamfer_core.h"
#include "common.h"

namespace mpa {
namespace {

constexpr int kMaxCells = 32768;
constexpr int kMaxAxis = 64;
constexpr int kStartStride = kMaxCells + 8;   // ints per (sample, shape, role) slot
#ifndef MPA_GRID_KS
#define MPA_GRID_KS 2
#endif
#ifndef MPA_GRID_XCD  // 1: XCD-aware, work-proportional wave table (grid_assign_plan); 0: waves x of pair y
#define MPA_GRID_XCD 1
#endif
#ifndef MPA_GRID_WAVES
#define MPA_GRID_WAVES 768
#endif
constexpr int kS = MPA_GRID_KS;                         // super-cell edge in fine cells (the query bins of the search): 2 is best
                                              // while predictions are far from the ground truth (a wave's search radius is
                                              // its worst query's), 3 once they are close (0.62 vs 0.74 ms loss forward)
constexpr int kMaxSuper = 4096;               // super-cells per sample (<= 22^3 would need more: see grid_params_from_boxes)
constexpr int kWorkStride = kMaxSuper + 20000 / 64 + 64;  // search work items per slot (<= nsuper + points/batch)
constexpr int kBatch = 64;                    // queries per search wave (1 per lane)
#ifndef MPA_GRID_DENSITY
#define MPA_GRID_DENSITY 0.6f
#endif
constexpr float kPointsPerCell = MPA_GRID_DENSITY;  // target points of one shape per fine cell (tools/variant_bench.sh)

struct __attribute__((aligned(16))) GridParams {
  float ox, oy, oz, h;
  float inv_h;
  int gx, gy, gz;
  int sgx, sgy, sgz, ncells;
  int nsuper, nvalid, pad0, pad1;
  int tb[2][6];  // per shape: cell bounding box of its valid points (x0, x1, y0, y1, z0, z1), inclusive
  int pad2[4];
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ void cell_of(const GridParams& g, float x, float y, float z, int& ix, int& iy,
                                        int& iz) {
  // (clamped as floats first: the generic operator bins far outliers, and a float -> int conversion must stay in range)
  ix = clampi((int)__builtin_amdgcn_fmed3f((x - g.ox) * g.inv_h, 0.0f, 1024.0f), 0, g.gx - 1);
  iy = clampi((int)__builtin_amdgcn_fmed3f((y - g.oy) * g.inv_h, 0.0f, 1024.0f), 0, g.gy - 1);
  iz = clampi((int)__builtin_amdgcn_fmed3f((z - g.oz) * g.inv_h, 0.0f, 1024.0f), 0, g.gz - 1);
}

__device__ __forceinline__ int key_of(const GridParams& g, int role, float x, float y, float z) {
  int ix, iy, iz;
  cell_of(g, x, y, z, ix, iy, iz);
  if (role == 0) return (iz * g.gy + iy) * g.gx + ix;
  return ((iz / kS) * g.sgy + (iy / kS)) * g.sgx + (ix / kS);
}

// ---- 1. grid parameters ----------------------------------------------------------------------------------------------
// From the bounding boxes of a sample's two shapes: lo / hi [6] = shape c's box in [3c .. 3c + 2] (valid parts only).
// The pose kernel of the loss leaves one box per PART (assembly_loss.hip: 12 floats, lo1 lo2 hi1 hi2); every sort block
// reduces its sample's boxes and evaluates this itself (a separate one-block-per-sample kernel in front of the sorts
// was 15 us of an almost empty chip).
// tlo / thi [6]: boxes that hold ALL points of each shape (what tb, the shapes' cell boxes, is taken from) — the same
// boxes for the fused loss; the generic operator lays the grid over outlier-trimmed boxes and passes the full ones here.
__device__ __forceinline__ void grid_params_from_boxes(const float* lo, const float* hi, int nvalid, GridParams& g,
                                                       const float* tlo, const float* thi) {
  const float ulo[3] = {__builtin_fminf(lo[0], lo[3]), __builtin_fminf(lo[1], lo[4]), __builtin_fminf(lo[2], lo[5])};
  const float uhi[3] = {__builtin_fmaxf(hi[0], hi[3]), __builtin_fmaxf(hi[1], hi[4]), __builtin_fmaxf(hi[2], hi[5])};
  float ex = uhi[0] - ulo[0], ey = uhi[1] - ulo[1], ez = uhi[2] - ulo[2];
  const float emax = __builtin_fmaxf(__builtin_fmaxf(ex, ey), __builtin_fmaxf(ez, 1e-12f));
  if (!(emax < 1e30f) || nvalid == 0) {  // non-finite coordinates or nothing to index: one cell
    g.ox = g.oy = g.oz = 0.0f;
    g.h = 1.0f;
    g.inv_h = 0.0f;
    g.gx = g.gy = g.gz = 2;
  } else {
    const float floor_e = emax / (float)kMaxAxis;  // flat clouds: no axis thinner than this
    ex = __builtin_fmaxf(ex, floor_e);
    ey = __builtin_fmaxf(ey, floor_e);
    ez = __builtin_fmaxf(ez, floor_e);
    float want = (float)nvalid / kPointsPerCell;
    want = want < 8.0f ? 8.0f : (want > (float)kMaxCells ? (float)kMaxCells : want);
    float h = cbrtf(ex * ey * ez / want);
    h = __builtin_fmaxf(h, emax / (float)kMaxAxis);
    for (int it = 0; it < 64; ++it) {
      g.gx = clampi((int)__builtin_ceilf(ex / h), 1, kMaxAxis);
      g.gy = clampi((int)__builtin_ceilf(ey / h), 1, kMaxAxis);
      g.gz = clampi((int)__builtin_ceilf(ez / h), 1, kMaxAxis);
      const int ns = ((g.gx + kS - 1) / kS) * ((g.gy + kS - 1) / kS) * ((g.gz + kS - 1) / kS);
      if (g.gx * g.gy * g.gz <= kMaxCells && ns <= kMaxSuper) break;
      h *= 1.1f;
    }
    g.ox = ulo[0];
    g.oy = ulo[1];
    g.oz = ulo[2];
    g.h = h;
    g.inv_h = 1.0f / h;
  }
  g.sgx = (g.gx + kS - 1) / kS;
  g.sgy = (g.gy + kS - 1) / kS;
  g.sgz = (g.gz + kS - 1) / kS;
  g.ncells = g.gx * g.gy * g.gz;
  g.nsuper = g.sgx * g.sgy * g.sgz;
  g.nvalid = nvalid;
  g.pad0 = g.pad1 = 0;
  g.pad2[0] = g.pad2[1] = g.pad2[2] = g.pad2[3] = 0;
  for (int c = 0; c < 2; ++c) {  // where each shape's points can be found, in cells (binning is monotone)
    if (nvalid == 0 || g.inv_h == 0.0f) {
      g.tb[c][0] = g.tb[c][2] = g.tb[c][4] = 0;
      g.tb[c][1] = g.gx - 1;
      g.tb[c][3] = g.gy - 1;
      g.tb[c][5] = g.gz - 1;
    } else {
      cell_of(g, tlo[3 * c], tlo[3 * c + 1], tlo[3 * c + 2], g.tb[c][0], g.tb[c][2], g.tb[c][4]);
      cell_of(g, thi[3 * c], thi[3 * c + 1], thi[3 * c + 2], g.tb[c][1], g.tb[c][3], g.tb[c][5]);
    }
  }
}

// ---- 2b. waves -> (sample, direction) with XCD locality ---------------------------------------------------------------
// Workgroups go to the 8 XCDs round-robin (linear id L runs on XCD L % 8) and every XCD has its own L2.  All waves
// that search one (sample, direction) pair therefore get ids of ONE residue class, so the pair's records are pulled
// into one L2 instead of eight — but samples differ 10x in size, so the pairs are dealt to the XCDs by descending
// work (snake order) and every pair gets a share of its XCD's wave slots proportional to its work.
// The plan (per XCD: its pairs and the prefix of their wave slots) is tiny; every search wave looks its slot up itself.
constexpr int kMaxPairs = 1024;            // 2 * B (larger batches fall back to the plain mapping)
constexpr int kPlanCap = kMaxPairs / 8;    // pairs per XCD
struct XcdPlan {
  int cnt[8];
  int lst[8][kPlanCap];
  int first[8][kPlanCap + 1];
  int work[kMaxPairs];  // batches of every (sample, direction): published by its query-side sort block (see sort_hand_over)
};
// one block of 1024 threads: the LAST sort block of the launch (grid_sort_kernel), once every pair's batch count is there
__device__ __forceinline__ void grid_assign_plan(int npairs, int nwaves, XcdPlan* __restrict__ plan) {
  __shared__ int work[kMaxPairs];
  __shared__ int lst[8][kPlanCap], cnt[8];
  const int t = threadIdx.x;
  for (int p = t; p < npairs; p += 1024)  // (agent-scope loads: the counts were written through by other XCDs' blocks)
    work[p] = __hip_atomic_load(plan->work + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t < 8) cnt[t] = 0;
  __syncthreads();
  for (int p = t; p < npairs; p += 1024) {  // rank by descending work (ties: lower pair first), snake over the XCDs
    int r = 0;
    for (int q = 0; q < npairs; ++q) r += work[q] > work[p] || (work[q] == work[p] && q < p);
    const int cyc = r >> 3, pos = r & 7, xcd = (cyc & 1) ? 7 - pos : pos;
    lst[xcd][cyc] = p;
    atomicAdd(&cnt[xcd], 1);  // (integer count: order-free)
  }
  __syncthreads();
  const int per = nwaves / 8;  // wave slots of one XCD
  if (t < 8) {  // proportional shares, at least one wave for every pair that has work
    long long rem_work = 0;
    int busy = 0;
    for (int i = 0; i < cnt[t]; ++i) {
      rem_work += work[lst[t][i]];
      busy += work[lst[t][i]] > 0;
    }
    int rem = per, at = 0;
    plan->cnt[t] = cnt[t];
    for (int i = 0; i < cnt[t]; ++i) {
      const int w = work[lst[t][i]];
      plan->lst[t][i] = lst[t][i];
      plan->first[t][i] = at;
      if (w > 0) {
        int n = (int)(((long long)rem * w + rem_work / 2) / rem_work);
        const int keep = busy - 1;  // one slot for each pair still to come
        n = n < 1 ? 1 : (n > rem - keep ? rem - keep : n);
        at += n;
        rem -= n;
        rem_work -= w;
        --busy;
      }
    }
    plan->first[t][cnt[t]] = at;
  }
}

// ---- 2. counting sort.  slot = (b*2 + shape)*2 + role; role 0 = TARGET (fine cells), 1 = QUERY (super-cells) ------
// One block of 1024 threads per (sample, shape) does the whole sort of that shape's <= 20 000 points with its two
// histograms (<= 32768 fine cells, <= 4096 super-cells: 144 KB) in LDS: count with LDS atomics, exclusive scan in
// place (the starts go to global memory for the search kernel, together with the work list of the QUERY role),
// scatter with the scanned array as cursor.  The order of the records inside a cell depends on the atomics; the
// search result does not (lexicographic updates).

// exclusive scan of cnt[0..nkeys) in place, 1024 threads, PER consecutive keys per thread; returns through st_out
// (nkeys + 1 entries).  WORK: also the prefix of ceil(count / kBatch) (ba_out) and the work list (wl_out).
// The histogram lives in LDS with one pad word per 32 keys: the scan below has thread t walk keys 32 t .. 32 t + 31, which
// unpadded puts all 64 lanes of every access on two banks.
__device__ __forceinline__ int padk(int k) { return k + (k >> 5); }

template <int PER, bool WORK>
__device__ __forceinline__ void block_scan(int* __restrict__ cnt, int nkeys, int* __restrict__ st_out,
                                           int* __restrict__ ba_out, int* __restrict__ wl_out, int (*wsum)[2]) {
  const int base = threadIdx.x * PER;
  int sum = 0, bsum = 0;
  for (int k = 0; k < PER; ++k) {
    const int v = base + k < nkeys ? cnt[padk(base + k)] : 0;
    sum += v;
    bsum += (v + kBatch - 1) / kBatch;
  }
  int incl = sum, bincl = bsum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off, 64), u = __shfl_up(bincl, off, 64);
    if ((int)(threadIdx.x & 63) >= off) {
      incl += t;
      bincl += u;
    }
  }
  __syncthreads();  // wsum may still be read by the previous scan
  if ((threadIdx.x & 63) == 63) {
    wsum[threadIdx.x >> 6][0] = incl;
    wsum[threadIdx.x >> 6][1] = bincl;
  }
  __syncthreads();
  int run = incl - sum, brun = bincl - bsum;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) {
    run += wsum[w][0];
    brun += wsum[w][1];
  }
  for (int k = 0; k < PER; ++k) {
    if (base + k < nkeys) {
      const int v = cnt[padk(base + k)];
      cnt[padk(base + k)] = run;  // becomes the scatter cursor
      if (WORK) {
        ba_out[base + k] = brun;
        for (int i = 0; i < (v + kBatch - 1) / kBatch; ++i) wl_out[brun + i] = base + k;  // work item -> super-cell
      }
      run += v;
      brun += (v + kBatch - 1) / kBatch;
    }
  }
  if (threadIdx.x == 1023) {
    st_out[nkeys] = run;
    if (WORK) ba_out[nkeys] = brun;
    if (WORK) wsum[16][0] = brun;  // the total, for sort_hand_over (a row of its own: rows 0..15 are still being read)
  }
  __syncthreads();
  // the starts, coalesced (written from the loop above every store instruction touched 64 different cache lines)
  for (int i = threadIdx.x; i < nkeys; i += 1024) st_out[i] = cnt[padk(i)];
}

// grid = 4*B (blockIdx.x = (b*2 + shape)*2 + role), block 1024: the two roles of a shape are sorted by different
// blocks (each reads the shape's points itself: 240 KB from L2) — 128 instead of 64 blocks on 256 CUs, and neither
// waits for the other's scan.
constexpr int kSortU = 4;
constexpr int kSortKeep = 20;  // points per thread held in registers by the sort blocks (1024 x 20 slots: the shipped P N)

// Where a sort block's points come from.  fetch(i, x, y, z): the point in flat slot i (unconditional load of a clamped
// position: conditional loads send the unrolled arrays to scratch) and whether it takes part.
struct PartsSource {  // the fused loss: P parts of N points, padded parts out
  static constexpr bool kKeep = true;  // the sort holds a thread's points in registers (grid_sort_role)
  const float* vsm;   // the sample's valid flags (LDS)
  const float* shape;
  int N, total;
  __device__ __forceinline__ bool fetch(int i, float& x, float& y, float& z) const {
    const bool ok = i < total && vsm[i / N] != 0.0f;
    const float* q = shape + 3LL * (ok ? i : 0);
    x = q[0], y = q[1], z = q[2];
    return ok;
  }
};
// the generic operator: one cloud of `total` points.  As TARGETS, a point equal to its predecessor is dropped: the
// strict-`<` in-order scan can never prefer it (same distance to every query, higher index), so the answer is unchanged
// — and the 1e3-filled padded parts of shape_cd_loss (utils/loss.py:173-175: N identical points per padded part) shrink
// to one record each.  As QUERIES the same points are dropped — a point equal to its predecessor has its predecessor's
// answer, copied behind the search (cloud_copy_runs_kernel) from the head of its run: the far-away padded points are the
// most expensive queries a grid can get (everything is "near" from 1e3 away), and there are thousands of copies of each.
// A sample routed to the exhaustive scan sorts nothing.
template <bool DEDUPE>
struct CloudSource {
  static constexpr bool kKeep = false;  // (its fetch reads the predecessor too: the register form spills)
  const float* cloud;
  int total;
  bool off;
  __device__ __forceinline__ bool fetch(int i, float& x, float& y, float& z) const {
    const bool in = i < total && !off;
    const float* q = cloud + 3LL * (in ? i : 0);
    x = q[0], y = q[1], z = q[2];
    if (!DEDUPE) return in;
    const float* r = cloud + 3LL * (in && i > 0 ? i - 1 : 0);
    const float px = r[0], py = r[1], pz = r[2];
    return in && (i == 0 || !(px == x && py == y && pz == z));
  }
};

template <int ROLE, bool GENERIC, class Src>
__device__ __forceinline__ void grid_sort_role(const Src& src, const GridParams& g, int slot, int* __restrict__ starts,
                                               int* __restrict__ batches, int* __restrict__ worklist, int work_stride,
                                               float4* __restrict__ records, int rec_stride, int* cnt, int (*wsum)[2]) {
  const int nkeys = ROLE == 0 ? g.ncells : g.nsuper;
  for (int i = threadIdx.x; i < padk(nkeys) + 1; i += 1024) cnt[i] = 0;
  __syncthreads();
  float4* out = records + (long long)slot * rec_stride;
  auto scan_keys = [&]() {
    if (ROLE == 0)
      block_scan<kMaxCells / 1024, false>(cnt, nkeys, starts + (long long)slot * kStartStride, nullptr, nullptr, wsum);
    else
      block_scan<kMaxSuper / 1024, true>(cnt, nkeys, starts + (long long)slot * kStartStride,
                                         batches + (long long)slot * kStartStride, worklist + (long long)slot * work_stride,
                                         wsum);
    __syncthreads();
    if (ROLE == 0 && threadIdx.x < 8) {  // sentinels: chunked reads may run past the last record
      const float inf = __builtin_inff();
      // (generic: the record count is what the scan left behind the last key — written by this block, in front of a barrier)
      const int nrec = GENERIC ? starts[(long long)slot * kStartStride + nkeys] : g.nvalid;
      out[nrec + threadIdx.x] = make_float4(inf, inf, inf, __int_as_float(0x7fffffff));
    }
  };
  if (Src::kKeep && src.total <= 1024 * kSortKeep) {
    // the shipped sizes (P N = 20 000 slots): a thread's <= 20 points stay in registers between the histogram and the scatter,
    // and all of their loads are in flight at once — the two passes of the loop below are ten dependent L2 round trips
    float x[kSortKeep], y[kSortKeep], z[kSortKeep];
    unsigned okm = 0u;
#pragma unroll
    for (int u = 0; u < kSortKeep; ++u) okm |= src.fetch(threadIdx.x + 1024 * u, x[u], y[u], z[u]) ? 1u << u : 0u;
#pragma unroll
    for (int u = 0; u < kSortKeep; ++u)
      if ((okm >> u) & 1u) atomicAdd(&cnt[padk(key_of(g, ROLE, x[u], y[u], z[u]))], 1);
    __syncthreads();
    scan_keys();
#pragma unroll
    for (int u = 0; u < kSortKeep; ++u)
      if ((okm >> u) & 1u)
        out[atomicAdd(&cnt[padk(key_of(g, ROLE, x[u], y[u], z[u]))], 1)] =
            make_float4(x[u], y[u], z[u], __int_as_float((int)threadIdx.x + 1024 * u));  // (the flat slot p * N + n)
    return;
  }
  // all point slots in one flat loop, kSortU loads per thread in flight (a loop over the parts was one dependent
  // memory round trip per valid part and pass)
  const int total = src.total;
  for (int i0 = threadIdx.x; i0 < total; i0 += 1024 * kSortU) {
    float x[kSortU], y[kSortU], z[kSortU];
    bool ok[kSortU];
#pragma unroll
    for (int u = 0; u < kSortU; ++u) ok[u] = src.fetch(i0 + 1024 * u, x[u], y[u], z[u]);
#pragma unroll
    for (int u = 0; u < kSortU; ++u)
      if (ok[u]) atomicAdd(&cnt[padk(key_of(g, ROLE, x[u], y[u], z[u]))], 1);
  }
  __syncthreads();
  scan_keys();
  for (int i0 = threadIdx.x; i0 < total; i0 += 1024 * kSortU) {
    float x[kSortU], y[kSortU], z[kSortU];
    bool ok[kSortU];
#pragma unroll
    for (int u = 0; u < kSortU; ++u) ok[u] = src.fetch(i0 + 1024 * u, x[u], y[u], z[u]);
#pragma unroll
    for (int u = 0; u < kSortU; ++u)
      if (ok[u])
        out[atomicAdd(&cnt[padk(key_of(g, ROLE, x[u], y[u], z[u]))], 1)] =
            make_float4(x[u], y[u], z[u], __int_as_float(i0 + 1024 * u));  // (the flat slot p * N + n)
  }
}

// The end of a sort block when the launch also plans the search's waves: the query-side block of pair (b, c) publishes
// its batch count, every block takes a ticket, and the last one plans.  The count goes out as ONE agent-scope atomic
// store (written through, acknowledged before the ticket) and is read with agent-scope loads — an agent-scope release
// fence here is a write-back of the XCD's whole L2 behind 320 KB of freshly sorted records per block: 10 of the
// launch's 58 us (LABBOOK 5.3 xviii).  Returns true in the last block.
__device__ __forceinline__ bool sort_hand_over(int role, int pair, int total, unsigned* __restrict__ ticket,
                                               XcdPlan* __restrict__ plan, bool* last) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (role == 1) {
      __hip_atomic_store(plan->work + pair, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  return *last;
}

// bbox [B*P][12]: per part lo1[3], lo2[3], hi1[3], hi2[3] of its points in the two shapes (valid parts; written by the
// pose kernel of the loss).  ticket: one word, zero at launch (the pose kernel clears it).  plan != NULL: the block that
// finishes last builds the search's wave plan (grid_assign_plan).
__global__ __launch_bounds__(1024) void grid_sort_kernel(const float* __restrict__ valids,
                                                         const float* __restrict__ S1,
                                                         const float* __restrict__ S2, int P, int N,
                                                         const float* __restrict__ bbox, GridParams* __restrict__ params,
                                                         int* __restrict__ starts, int* __restrict__ batches,
                                                         int* __restrict__ worklist, float4* __restrict__ records,
                                                         int rec_stride, unsigned* __restrict__ ticket,
                                                         XcdPlan* __restrict__ plan, int nwaves,
                                                         const int* __restrict__ route) {
  __shared__ int cnt[kMaxCells + kMaxCells / 32 + 1];
  __shared__ int wsum[17][2];  // 16 wave totals of the scans + the query side's batch total
  __shared__ GridParams gsm;
  __shared__ bool last;
  __shared__ float vsm[64];  // the sample's valid flags
  const int role = blockIdx.x & 1, c = (blockIdx.x >> 1) & 1, b = blockIdx.x >> 2;
  const float* vb = valids + (long long)b * P;
  if (threadIdx.x < 64) vsm[threadIdx.x] = (int)threadIdx.x < P ? vb[threadIdx.x] : 0.0f;
  // the sample's grid: boxes of its valid parts (one lane per part, P <= 64), then the closed-form parameters
  if (threadIdx.x < 64) {
    const int p = threadIdx.x;
    const bool on = p < P && vb[p] != 0.0f;
    float lo[6], hi[6];
    const float* bp = bbox + ((long long)b * P + (p < P ? p : 0)) * 12;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      lo[k] = on ? bp[k] : __builtin_inff();
      hi[k] = on ? bp[6 + k] : -__builtin_inff();
    }
    int nval = on ? N : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        lo[k] = __builtin_fminf(lo[k], __shfl_xor(lo[k], off, 64));
        hi[k] = __builtin_fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
      }
      nval += __shfl_xor(nval, off, 64);
    }
    if (threadIdx.x == 0) {
      GridParams gl;
      grid_params_from_boxes(lo, hi, nval, gl, lo, hi);
      gsm = gl;
      if (role == 0 && c == 0) params[b] = gl;  // for the search kernel
    }
  }
  __syncthreads();
  // a uniform copy in scalar registers (the binning reads half a dozen fields per point)
  GridParams g;
  {
    const int* src = reinterpret_cast<const int*>(&gsm);
    int* dst = reinterpret_cast<int*>(&g);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(GridParams) / 4); ++k) dst[k] = __builtin_amdgcn_readfirstlane(src[k]);
  }
  const int slot = (b * 2 + c) * 2 + role;
  const float* shape = (c == 0 ? S1 : S2) + 3LL * b * P * N;
  const PartsSource src{vsm, shape, N, P * N};
  if (route != nullptr && route[b] == 0) {
    // this sample's whole-shape search is the leaf search's (leaf_nn.hip): no records, and a work list of length zero —
    // the search waves of its two pairs find nothing to do (bst[nsuper] = 0)
    if (role == 1 && threadIdx.x == 0) batches[(long long)slot * kStartStride + g.nsuper] = 0;
  } else if (role == 0) {
    grid_sort_role<0, false>(src, g, slot, starts, batches, worklist, kWorkStride, records, rec_stride, cnt, wsum);
  } else {
    grid_sort_role<1, false>(src, g, slot, starts, batches, worklist, kWorkStride, records, rec_stride, cnt, wsum);
  }
  if (plan == nullptr) return;
  // the last block of the launch plans the search's waves
  const bool routed_away = route != nullptr && route[b] == 0;
  if (!sort_hand_over(role, b * 2 + c, routed_away ? 0 : wsum[16][0], ticket, plan, &last)) return;
  grid_assign_plan((int)(gridDim.x / 2), nwaves, plan);
}

// ---- 2c. the generic operator's sort: two plain clouds per sample (chamfer.hip, mpa_chamfer_forward) -------------------
// Same launch shape and outputs as grid_sort_kernel; the grid comes from the clouds themselves.  A uniform grid over the
// bounding box dies on outliers — shape_cd_loss hands over clouds whose padded parts sit at (1e3, 1e3, 1e3) while the
// shape itself spans ~1 — so the grid is laid over an OUTLIER-TRIMMED box (two rounds of 3-sigma clipping of the
// distinct points, in double) and everything outside is binned into the border cells, which the search treats as
// unbounded outwards.  The trimming only steers speed: the search is exact for ANY box.
// Samples holding a non-finite or huge (> 1e15: squares would overflow) coordinate are flagged instead (fallback[b] = 1):
// they sort nothing here and are answered by the exhaustive scan behind the search.
struct CloudStats {
  double sum[3], sq[3];
  float lo[3], hi[3];
  int cnt, bad;
};

// block-wide (1024 threads) reduction of one CloudStats; the result lands in every thread.  Fixed order: all four sort
// blocks of a sample must arrive at the SAME grid, bit for bit.
__device__ __forceinline__ void block_reduce_stats(CloudStats& st, CloudStats* sm /* [16] in LDS */, CloudStats* out_sm) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      st.sum[k] += __shfl_xor(st.sum[k], off, 64);
      st.sq[k] += __shfl_xor(st.sq[k], off, 64);
      st.lo[k] = __builtin_fminf(st.lo[k], __shfl_xor(st.lo[k], off, 64));
      st.hi[k] = __builtin_fmaxf(st.hi[k], __shfl_xor(st.hi[k], off, 64));
    }
    st.cnt += __shfl_xor(st.cnt, off, 64);
    st.bad |= __shfl_xor(st.bad, off, 64);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = st;
  __syncthreads();
  if (threadIdx.x == 0) {
    CloudStats t = sm[0];
#pragma unroll 1
    for (int w = 1; w < 16; ++w) {
      for (int k = 0; k < 3; ++k) {
        t.sum[k] += sm[w].sum[k];
        t.sq[k] += sm[w].sq[k];
        t.lo[k] = __builtin_fminf(t.lo[k], sm[w].lo[k]);
        t.hi[k] = __builtin_fmaxf(t.hi[k], sm[w].hi[k]);
      }
      t.cnt += sm[w].cnt;
      t.bad |= sm[w].bad;
    }
    *out_sm = t;
  }
  __syncthreads();
  st = *out_sm;
}

// statistics of the distinct points of one cloud that lie inside [clo, chi] (all three axes)
__device__ __forceinline__ CloudStats cloud_stats(const float* __restrict__ cloud, int n, const float* clo, const float* chi) {
  CloudStats st;
  for (int k = 0; k < 3; ++k) {
    st.sum[k] = st.sq[k] = 0.0;
    st.lo[k] = __builtin_inff();
    st.hi[k] = -__builtin_inff();
  }
  st.cnt = st.bad = 0;
  constexpr int U = 8;  // points per thread in flight (one dependent memory round trip per iteration otherwise)
#pragma unroll 1
  for (int i0 = threadIdx.x; i0 < n; i0 += 1024 * U) {
    float v[U][3], r[U][3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 1024 * u < n ? i0 + 1024 * u : n - 1;
      const float* q = cloud + 3LL * i;
      const float* pr = cloud + 3LL * (i > 0 ? i - 1 : 0);
      v[u][0] = q[0], v[u][1] = q[1], v[u][2] = q[2];
      r[u][0] = pr[0], r[u][1] = pr[1], r[u][2] = pr[2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 1024 * u;
      const bool dup = i > 0 && r[u][0] == v[u][0] && r[u][1] == v[u][1] && r[u][2] == v[u][2];
      bool in = i < n && !dup;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (i < n && !(__builtin_fabsf(v[u][k]) <= 1e15f)) st.bad = 1;  // (NaN fails the comparison too)
        in = in && v[u][k] >= clo[k] && v[u][k] <= chi[k];
      }
      if (in) {
        ++st.cnt;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          st.sum[k] += (double)v[u][k];
          st.sq[k] += (double)v[u][k] * (double)v[u][k];
          st.lo[k] = __builtin_fminf(st.lo[k], v[u][k]);
          st.hi[k] = __builtin_fmaxf(st.hi[k], v[u][k]);
        }
      }
    }
  }
  return st;
}

// mean +- 3 sigma of `st`, rounded outwards, into clo / chi; returns whether that interval cuts anything off the box
__device__ __forceinline__ bool clip_interval(const CloudStats& st, float* clo, float* chi) {
  bool cuts = false;
  for (int k = 0; k < 3; ++k) {
    const double mean = st.sum[k] / (double)st.cnt;
    double var = st.sq[k] / (double)st.cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double sd = __builtin_sqrt(var), a = mean - 3.0 * sd, b = mean + 3.0 * sd;
    clo[k] = (float)(a - 1e-6 * __builtin_fabs(a) - 1e-30);
    chi[k] = (float)(b + 1e-6 * __builtin_fabs(b) + 1e-30);
    cuts = cuts || clo[k] > st.lo[k] || chi[k] < st.hi[k];
  }
  return cuts;
}

// What the sort needs to know about one cloud of one sample: written by cloud_stats_kernel, [b * 2 + cloud].
struct __attribute__((aligned(16))) CloudBox {
  float lo[3], hi[3];    // trimmed box of its distinct points (what the grid should cover)
  float tlo[3], thi[3];  // box of all its points
  int cnt, bad, pad0, pad1;
};

// grid = (2, B), block 1024: one block per (cloud, sample) — the box of the cloud (moments of its distinct points, up to
// two rounds of 3-sigma clipping if that cuts anything off) and the head of every point's run of identical points
// (itself for a point that differs from its predecessor).
__global__ __launch_bounds__(1024) void cloud_stats_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                           int n1, int n2, int hstride, CloudBox* __restrict__ boxes,
                                                           int* __restrict__ heads) {
  __shared__ CloudStats ssm[17];
  __shared__ int wave_head[2][16];
  const int c = blockIdx.x, b = blockIdx.y, n = c == 0 ? n1 : n2;
  const float* const cloud = (c == 0 ? xyz1 + 3LL * b * n1 : xyz2 + 3LL * b * n2);
  float clo[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  float chi[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  CloudStats st = cloud_stats(cloud, n, clo, chi);
  block_reduce_stats(st, ssm, ssm + 16);
  CloudBox out;
  for (int a = 0; a < 3; ++a) {
    out.tlo[a] = out.lo[a] = st.lo[a];
    out.thi[a] = out.hi[a] = st.hi[a];
  }
  out.cnt = st.cnt;
  out.bad = st.bad;
  out.pad0 = out.pad1 = 0;
  if (!st.bad) {
#pragma unroll 1
    for (int round = 0; round < 2 && st.cnt > 0; ++round) {  // (uniform condition: st is the same in every thread)
      if (!clip_interval(st, clo, chi)) break;
      st = cloud_stats(cloud, n, clo, chi);
      block_reduce_stats(st, ssm, ssm + 16);
    }
    if (st.cnt > 0) {  // (nothing survived the clipping: cannot happen, but then the full box stays)
      for (int a = 0; a < 3; ++a) {
        out.lo[a] = st.lo[a];
        out.hi[a] = st.hi[a];
      }
    }
  }
  if (threadIdx.x == 0) boxes[b * 2 + c] = out;
  // run heads: tiles of 1024 consecutive points, the last head of a tile carried into the next (one barrier per tile:
  // the per-wave results are double-buffered, and every thread derives the carry itself)
  int* hd = heads + (long long)(b * 2 + c) * hstride;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int carry = 0, buf = 0;
  constexpr int HU = 4;  // tiles whose points are loaded together (one memory round trip per HU tiles)
  for (int g0 = 0; g0 < n; g0 += 1024 * HU) {
    bool uq[HU];
#pragma unroll
    for (int u = 0; u < HU; ++u) {
      const int i = g0 + 1024 * u + (int)threadIdx.x;
      const float* q = cloud + 3LL * (i < n ? i : 0);
      const float* r = cloud + 3LL * (i < n && i > 0 ? i - 1 : 0);
      uq[u] = i < n && (i == 0 || !(r[0] == q[0] && r[1] == q[1] && r[2] == q[2]));
    }
#pragma unroll
    for (int u = 0; u < HU; ++u, buf ^= 1) {
      const int t0 = g0 + 1024 * u, i = t0 + (int)threadIdx.x;
      if (t0 >= n) break;  // (uniform)
      const unsigned long long below = __ballot(uq[u]) & (~0ull >> (63 - lane));
      int h = below ? t0 + w * 64 + 63 - __builtin_clzll(below) : -1;
      if (lane == 63) wave_head[buf][w] = h;
      __syncthreads();
      int next = carry;
      for (int ww = 0; ww < 16; ++ww) {
        const int v = wave_head[buf][ww];
        if (ww < w && h < 0 && v >= 0) next = v;  // (ascending: the last hit below this wave wins)
        carry = v >= 0 ? v : carry;
      }
      if (h < 0) h = next;
      if (i < n) hd[i] = h;
    }
  }
}

__global__ __launch_bounds__(1024) void cloud_sort_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                          int n1, int n2, const CloudBox* __restrict__ boxes,
                                                          GridParams* __restrict__ params, int* __restrict__ fallback,
                                                          int* __restrict__ starts, int* __restrict__ batches,
                                                          int* __restrict__ worklist, int work_stride,
                                                          float4* __restrict__ records, int rec_stride,
                                                          unsigned* __restrict__ ticket, XcdPlan* __restrict__ plan,
                                                          int nwaves) {
  __shared__ int cnt[kMaxCells + kMaxCells / 32 + 1];
  __shared__ int wsum[17][2];  // 16 wave totals of the scans + the query side's batch total
  __shared__ GridParams gsm;
  __shared__ bool last;
  const int role = blockIdx.x & 1, c = (blockIdx.x >> 1) & 1, b = blockIdx.x >> 2;
  const float* const cl0 = xyz1 + 3LL * b * n1;
  const float* const cl1 = xyz2 + 3LL * b * n2;
  if (threadIdx.x == 0) {  // the sample's grid, from the boxes of its two clouds
    const CloudBox b0 = boxes[b * 2], b1 = boxes[b * 2 + 1];
    float box[24];  // lo[6] hi[6] (trimmed: what the grid covers) tlo[6] thi[6] (full) — [3 * cloud + axis]
    for (int a = 0; a < 3; ++a) {
      box[a] = b0.lo[a], box[3 + a] = b1.lo[a], box[6 + a] = b0.hi[a], box[9 + a] = b1.hi[a];
      box[12 + a] = b0.tlo[a], box[15 + a] = b1.tlo[a], box[18 + a] = b0.thi[a], box[21 + a] = b1.thi[a];
    }
    int bad = b0.bad | b1.bad;
    // a trimmed box too large for fp32 cell arithmetic (its volume must not overflow): the exhaustive scan as well
    float e = 0.0f;
    for (int a = 0; a < 12; ++a) e = __builtin_fmaxf(e, __builtin_fabsf(box[a]));
    if (!(e < 1e10f)) bad = 1;
    GridParams gl;
    grid_params_from_boxes(box, box + 6, bad ? 0 : (b0.cnt > b1.cnt ? b0.cnt : b1.cnt), gl, box + 12, box + 18);
    gl.pad0 = bad;
    gsm = gl;
    if (role == 0 && c == 0) {
      params[b] = gl;
      fallback[b] = bad;
    }
  }
  __syncthreads();
  GridParams g;
  {
    const int* src = reinterpret_cast<const int*>(&gsm);
    int* dst = reinterpret_cast<int*>(&g);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(GridParams) / 4); ++k) dst[k] = __builtin_amdgcn_readfirstlane(src[k]);
  }
  const float* const cloud_c = c == 0 ? cl0 : cl1;
  const int n_c = c == 0 ? n1 : n2;
  const int slot = (b * 2 + c) * 2 + role;
  const CloudSource<true> src{cloud_c, n_c, g.pad0 != 0};
  if (role == 0)
    grid_sort_role<0, true>(src, g, slot, starts, batches, worklist, work_stride, records, rec_stride, cnt, wsum);
  else
    grid_sort_role<1, true>(src, g, slot, starts, batches, worklist, work_stride, records, rec_stride, cnt, wsum);
  if (plan == nullptr) return;
  if (!sort_hand_over(role, b * 2 + c, wsum[16][0], ticket, plan, &last)) return;
  grid_assign_plan((int)(gridDim.x / 2), nwaves, plan);
}

// behind the search (and the hand-back scan): every point that repeats its predecessor takes the answer of its run's head
__global__ __launch_bounds__(256) void cloud_copy_runs_kernel(const int* __restrict__ heads, int n1, int n2, int hstride,
                                                              float* __restrict__ dist1, float* __restrict__ dist2,
                                                              long long* __restrict__ idx1, long long* __restrict__ idx2) {
  const int b = blockIdx.y >> 1, c = blockIdx.y & 1, n = c == 0 ? n1 : n2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int h = heads[(long long)(b * 2 + c) * hstride + i];
  if (h == i) return;
  float* d = (c == 0 ? dist1 : dist2) + (long long)b * n;
  long long* ix = (c == 0 ? idx1 : idx2) + (long long)b * n;
  d[i] = d[h];
  ix[i] = ix[h];
}

// ---- 3. search ----------------------------------------------------------------------------------------------------
// One query per lane: a super-cell holds a few dozen queries, rarely more than 64, and a packed two-query
// `v_pk_*_f32` costs two issue slots anyway — so a second query slot per lane would mostly double the VALU time
// of the scan for nothing.  (Super-cells with more than 64 queries simply appear as several work items.)
struct LaneState {
  float X, Y, Z;
  float best;
  int bidx;
  int split;  // wave-uniform, 1 / 2 / 4: with <= 32 (16) queries in the work item, 2 (4) groups of lanes hold the SAME
              // queries and each group scans every 2nd (4th) candidate; the groups are merged, lexicographically,
              // wherever the best is used
  float qnlo;  // a lower bound of |q|^2 (scan_cand's gate)
};

// ---- the scan's gate ------------------------------------------------------------------------------------------------------------
// The pinned distance costs 8 VALU operations per pair (3 sub, 3 mul, 2 add: no FMA, by contract), and the scan is where the
// kernel's VALU time goes.  Almost every candidate loses, and THAT can be decided with three FMAs: a staged candidate
// carries  tn' = fl(|t|^2) (1 - c)  in its fourth word, the lane computes
//     f = fma(qz, -2 tz, fma(qy, -2 ty, fma(qx, -2 tx, tn')))  ~  |t|^2 - 2 q.t  =  D - |q|^2        (D = |q - t|^2)
// (the staged record holds -2 t: exact, and the pinned path gets fl(qx - tx) back as fma(0.5, -2 tx, qx))
// and only candidates with  f <= thr = best - qnlo  (qnlo = fl(|q|^2) (1 - c), thr rounded up) get the pinned arithmetic and the
// lexicographic update.  The gate never contributes a digit to a result; it must only never reject a candidate whose pinned
// distance d is <= best.  With u = 2^-24, Qn = |q|^2, Tn = |t|^2 in real arithmetic:
//     tn' <= Tn (1 - c + 4 u);  three FMA roundings of partial sums bounded by 2 (Qn + Tn):  f <= Tn - 2 q.t - c Tn + 10 u (Qn + Tn);
//     the pinned chain is within 6 u of D <= 2 (Qn + Tn):  D <= d + 12 u (Qn + Tn);
//     so d <= best implies  f <= best - Qn + 22 u (Qn + Tn) - c Tn  <=  best - Qn (1 - 22 u)   for c >= 22 u,
//     and qnlo = fl(Qn)(1 - c) <= Qn (1 + 3 u)(1 - c)(1 + u) <= Qn (1 - 22 u)   for c >= 27 u = 1.6e-6.        c = 4e-6.
// Magnitudes whose squares may overflow take the pinned path unconditionally (tn' = -inf / qnlo = -inf); a NaN gate value
// (inf - inf, 0 * inf) belongs to a pair whose pinned distance is inf or NaN, which never wins; 1e-30 of absolute slack covers
// underflowing products.  The sentinel records (inf, inf, inf, +inf) give f = +inf or NaN.
#ifndef MPA_GRID_EXP  // 3: every scan runs twice (same results) — the time difference is what the scans cost (LABBOOK 6.3)
#define MPA_GRID_EXP 0
#endif
#ifndef MPA_GRID_FMA_GATE
#define MPA_GRID_FMA_GATE 1  // 0: the pinned distance for every pair (the scan of rounds 2-5; A/B builds)
#endif
constexpr float kGateC = 4e-6f;
__device__ __forceinline__ float gate_norm_lo(float x, float y, float z) {
  const float n = (x * x + y * y) + z * z;
  return n < 1e30f ? n * (1.0f - kGateC) : -__builtin_inff();  // (NaN: -inf as well — the pinned path decides)
}
// the LDS form of a target record: coordinates and tn'; the original point index goes to a list of its own (read only by the
// few pairs that pass the gate)
__device__ __forceinline__ float4 gate_record(const float4 t) {
#if MPA_GRID_FMA_GATE
  return make_float4(-2.0f * t.x, -2.0f * t.y, -2.0f * t.z, gate_norm_lo(t.x, t.y, t.z));
#else
  return t;
#endif
}

__device__ __forceinline__ float dist_exact_s(float dx, float dy, float dz) { return (dx * dx + dy * dy) + dz * dz; }

// lexicographic update with one candidate given as scalars
__device__ __forceinline__ void consider(LaneState& s, float tx, float ty, float tz, int tidx) {
  const float d = dist_exact_s(s.X - tx, s.Y - ty, s.Z - tz);
  if (d < s.best || (d == s.best && tidx < s.bidx)) {
    s.best = d;
    s.bidx = tidx;
  }
}

// Wave-wide reductions and the prefix sum on the DPP network (quad permutes, row mirrors, row broadcasts): ~7 VALU
// instructions each, no LDS.  As `__shfl_xor` / `__shfl_up` loops (ds_bpermute + select + op per step) they were a
// fifth of the search kernel's VALU instructions — and that kernel's VALU pipe is 90 % busy
// (profiles/r05_c2_grid_search_sq_inst_mix.txt).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_self(int v) {  // rows outside ROW_MASK (and lanes without a source) keep v
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_zero(int v) {  // ... read 0
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ float wave_max(float v) {
  auto mx = [](float a, int b) { return __builtin_fmaxf(a, __int_as_float(b)); };
  v = mx(v, dpp_self<0xB1, 0xf>(__float_as_int(v)));   // quad_perm [1,0,3,2]
  v = mx(v, dpp_self<0x4E, 0xf>(__float_as_int(v)));   // quad_perm [2,3,0,1]
  v = mx(v, dpp_self<0x141, 0xf>(__float_as_int(v)));  // row_half_mirror
  v = mx(v, dpp_self<0x140, 0xf>(__float_as_int(v)));  // row_mirror: every row of 16 holds its maximum
  v = mx(v, dpp_self<0x142, 0xa>(__float_as_int(v)));  // row_bcast15 into rows 1, 3
  v = mx(v, dpp_self<0x143, 0xc>(__float_as_int(v)));  // row_bcast31 into rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_max_i(int v) {
  v = max(v, dpp_self<0xB1, 0xf>(v));
  v = max(v, dpp_self<0x4E, 0xf>(v));
  v = max(v, dpp_self<0x141, 0xf>(v));
  v = max(v, dpp_self<0x140, 0xf>(v));
  v = max(v, dpp_self<0x142, 0xa>(v));
  v = max(v, dpp_self<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
// inclusive prefix sum over the lanes
__device__ __forceinline__ int wave_prefix_sum(int v) {
  v += dpp_zero<0x111, 0xf>(v);  // row_shr:1
  v += dpp_zero<0x112, 0xf>(v);  // row_shr:2
  v += dpp_zero<0x114, 0xf>(v);  // row_shr:4
  v += dpp_zero<0x118, 0xf>(v);  // row_shr:8: every row of 16 holds its own prefix sums
  v += dpp_zero<0x142, 0xa>(v);  // row_bcast15: rows 1, 3 += the total of the row before
  v += dpp_zero<0x143, 0xc>(v);  // row_bcast31: rows 2, 3 += the total of rows 0-1
  return v;
}

// squared distance between the intervals [a0, a1] and [b0, b1] on one axis
__device__ __forceinline__ float gap(float a0, float a1, float b0, float b1) {
  const float g = __builtin_fmaxf(__builtin_fmaxf(b0 - a1, a0 - b1), 0.0f);
  return g;
}

// A batch of up to 64 record ranges, one per lane ([rb, re); empty for idle lanes), scanned by the whole wave.
// Walking the ranges one after the other costs two dependent memory latencies per range for a handful of records
// each.  Instead every lane copies ITS range into one LDS candidate list (vector loads, all ranges in flight at
// once), and then all lanes scan that list with broadcast LDS reads: two latencies per BATCH.  Long lists are
// processed in windows of the LDS buffer; single long ranges go to the scalar scan, whose long runs amortise the
// latency by themselves.
#ifndef MPA_GRID_GATHER_U  // records per lane in flight in a window's gather.  A window holds kCand - 16 = 112 records, so 2 covers
#define MPA_GRID_GATHER_U 1  // it in one round — and costs 8 registers = the sixth wave per SIMD: 226 vs 212 us with the gated scan (round 6)
#endif
// Pins a loaded record in registers at this point of the program.  Without it the compiler sinks the second load of a lane
// into the `if` that stores it — the ISA of rounds 2-3 read  load, s_waitcnt vmcnt(0), store, branch, load, s_waitcnt
// vmcnt(0), store : the two gathers of a window, meant to be in flight together, were two dependent memory round trips.
__device__ __forceinline__ void pin_record(float4& t) { asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w)); }
#ifndef MPA_GRID_CAND_CHUNK  // candidates per step of the LDS scan: 4 instead of 8 frees 16 registers (94 -> 78: 6 waves per
#define MPA_GRID_CAND_CHUNK 4  // SIMD instead of 5; 0.279 -> 0.265 ms); 2 costs more LDS instructions than the 7th wave gains
#endif
#ifndef MPA_GRID_CAND_CHUNK_GENERIC
#define MPA_GRID_CAND_CHUNK_GENERIC 3
#endif
#ifndef MPA_GRID_CAND
#define MPA_GRID_CAND 128
#endif
constexpr int kCand = MPA_GRID_CAND;  // candidate records per LDS window (2.5 KB per wave: the window is what limits the
                                      // waves per CU, and this latency-bound search wants all of them — 512 -> 128 records
                                      // together with 768 instead of 128 persistent waves per (sample, direction) took
                                      // the kernel from 0.44 to 0.28 ms; tools/variant_bench.sh)
#ifndef MPA_GRID_LONG
#define MPA_GRID_LONG 32
#endif
constexpr int kLongRange = MPA_GRID_LONG;  // ranges longer than this are fetched by the whole wave, one range at a time

#ifdef MPA_GRID_STATS  // instrumented build for tools/probe_grid_stats.py only (never in libmpa_hip.so)
__device__ unsigned long long g_grid_stats[24];  // 0-7: items, active lanes, scan_batch calls, candidates, long-range candidates,
                                                 // outer-ring batches, items without a bound after the seed; 8-15: shader-clock ticks of
                                                 // a wave's phases: prologue, item header, seed, rings 0-1, outer rings, pads + store,
                                                 // whole wave, waves; 16-18: scan chunks, chunks with a lane past the gate, pinned
                                                 // evaluations (wave-level: a candidate slot with such a lane)
#ifdef MPA_GRID_TIMING  // phase timing: no counters inside the timed phases, the per-wave sums leave in one batch at the end
#define MPA_STAT(i, v) do { } while (0)
#define MPA_TICK(i)                                                        \
  do {                                                                     \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            \
    const unsigned long long now_ = __builtin_readcyclecounter();          \
    tacc_[(i) - 8] += now_ - tick_;                                        \
    tick_ = now_;                                                          \
  } while (0)
#define MPA_TICK_INIT()                                                    \
  unsigned long long tick_ = __builtin_readcyclecounter(), tacc_[6] = {0, 0, 0, 0, 0, 0}; \
  const unsigned long long tick0_ = tick_
#define MPA_TICK_END()                                                                                        \
  do {                                                                                                        \
    if (threadIdx.x == 0) {                                                                                   \
      const unsigned long long end_ = __builtin_readcyclecounter();                                           \
      for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_grid_stats[8 + k_], tacc_[k_]);                              \
      atomicAdd(&g_grid_stats[14], end_ - tick0_);                                                            \
      atomicAdd(&g_grid_stats[15], 1ull);                                                                     \
    }                                                                                                         \
  } while (0)
#else
#define MPA_STAT(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_grid_stats[i], (unsigned long long)(v)); } while (0)
#define MPA_TICK(i) do { } while (0)
#define MPA_TICK_INIT() do { } while (0)
#define MPA_TICK_END() do { } while (0)
#endif
#else
#define MPA_STAT(i, v) do { } while (0)
#define MPA_TICK(i) do { } while (0)
#define MPA_TICK_INIT() do { } while (0)
#define MPA_TICK_END() do { } while (0)
#endif

// every lane scans the wn candidate records staged in LDS (padded to a multiple of 8 with sentinels)
#if MPA_GRID_FMA_GATE
// STEP = s.split as a constant: the T reads of a chunk are one address register plus immediate offsets
template <int T, int STEP>
__device__ __forceinline__ void scan_cand_step(LaneState& s, const float4* __restrict__ cand, const int* __restrict__ cidx,
                                               int wn) {
  const int sub = (int)threadIdx.x / (64 / STEP);
  auto threshold = [&]() {  // best - qnlo, rounded up
    const float t0 = s.best - s.qnlo;
    return __builtin_fmaf(__builtin_fabsf(t0), 2.4e-7f, t0) + 1e-30f;
  };
  float thr = threshold();
  const float4* p = cand + sub;
  for (int j0 = 0; j0 < wn; j0 += T * STEP, p += T * STEP) {
    float4 cur[T];
#pragma unroll
    for (int t = 0; t < T; ++t) cur[t] = p[t * STEP];
    float f[T];  // (the staged record holds -2 t and tn')
#pragma unroll
    for (int t = 0; t < T; ++t)
      f[t] = __builtin_fmaf(s.Z, cur[t].z, __builtin_fmaf(s.Y, cur[t].y, __builtin_fmaf(s.X, cur[t].x, cur[t].w)));
    float fmin = f[0];
#pragma unroll
    for (int t = 1; t < T; ++t) fmin = __builtin_fminf(fmin, f[t]);
    MPA_STAT(16, 1);
    MPA_STAT(17, __ballot(fmin <= thr) != 0 ? 1 : 0);
    if (fmin <= thr) {  // a candidate that may improve on the best, or tie with it
#pragma unroll
      for (int t = 0; t < T; ++t) {
        MPA_STAT(18, __ballot(f[t] <= thr) != 0 ? 1 : 0);
        if (f[t] <= thr) {
          // X - t with t = -cur / 2: the product is exact, the sum rounds once — fl(X - tx)
          const float d = dist_exact_s(__builtin_fmaf(0.5f, cur[t].x, s.X), __builtin_fmaf(0.5f, cur[t].y, s.Y),
                                       __builtin_fmaf(0.5f, cur[t].z, s.Z));
          const int ti = cidx[j0 + t * STEP + sub];
          if (d < s.best || (d == s.best && ti < s.bidx)) {
            s.best = d;
            s.bidx = ti;
            thr = threshold();
          }
        }
      }
    }
  }
}
template <int T>
__device__ __forceinline__ void scan_cand(LaneState& s, const float4* __restrict__ cand, const int* __restrict__ cidx, int wn) {
#if MPA_GRID_EXP == 3  // (timing experiment: every scan twice)
  if (s.split == 1) scan_cand_step<T, 1>(s, cand, cidx, wn);
  else if (s.split == 2) scan_cand_step<T, 2>(s, cand, cidx, wn);
  else scan_cand_step<T, 4>(s, cand, cidx, wn);
  asm volatile("" : "+v"(s.best), "+v"(s.bidx));
#endif
  if (s.split == 1) scan_cand_step<T, 1>(s, cand, cidx, wn);  // (wave-uniform)
  else if (s.split == 2) scan_cand_step<T, 2>(s, cand, cidx, wn);
  else scan_cand_step<T, 4>(s, cand, cidx, wn);
}
#else
template <int T>
__device__ __forceinline__ void scan_cand(LaneState& s, const float4* __restrict__ cand, const int* __restrict__ cidx, int wn) {
  // split > 1: this group of lanes takes candidates sub, sub + split, ... (split LDS addresses per read, not one)
  const int step = s.split, sub = (int)threadIdx.x / (64 / s.split);
  for (int rep = 0; rep < (MPA_GRID_EXP == 3 ? 2 : 1); ++rep)
  for (int j0 = 0; j0 < wn; j0 += T * step) {
    float4 cur[T];
#pragma unroll
    for (int t = 0; t < T; ++t) cur[t] = cand[j0 + t * step + sub];
    float d[T];
#pragma unroll
    for (int t = 0; t < T; ++t) d[t] = dist_exact_s(s.X - cur[t].x, s.Y - cur[t].y, s.Z - cur[t].z);
    float cmin = d[0];
#pragma unroll
    for (int t = 1; t < T; ++t) cmin = __builtin_fminf(cmin, d[t]);
    if (cmin <= s.best) {  // rare: an improvement, or a tie that may carry a lower index
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int ti = __float_as_int(cur[t].w);
        if (d[t] < s.best || (d[t] == s.best && ti < s.bidx)) {
          s.best = d[t];
          s.bidx = ti;
        }
      }
    }
  }
}
#endif

// split mode: every group ends up with the best (distance, index) of all groups
__device__ __forceinline__ void merge_halves(LaneState& s) {
  for (int m = 32; m >= 64 / s.split && m >= 16; m >>= 1) {  // split 2: lanes l ^ 32; split 4: also l ^ 16
    const float ob = __shfl_xor(s.best, m, 64);
    const int oi = __shfl_xor(s.bidx, m, 64);
    if (ob < s.best || (ob == s.best && oi < s.bidx)) {
      s.best = ob;
      s.bidx = oi;
    }
  }
}

// One long contiguous range [begin, end) of the target records (wave-uniform): ALL lanes fetch it together — 64
// records per memory round trip and instruction, several in flight — into the LDS window, then scan it.  (A lane
// copying its own long range alone moves 2 records per round trip; the scalar-operand scan moves 8.)
template <int TC>
__device__ __forceinline__ void scan_range_coop(LaneState& s, const float4* __restrict__ trec, int begin, int end,
                                                float4* __restrict__ cand, int* __restrict__ cidx) {
  constexpr int T = 4 * TC, kCap = kCand - T;  // (4x: a split-4 scan reads up to that far past the end)
  const int lane = threadIdx.x;
  for (int w0 = begin; w0 < end; w0 += kCap) {
    const int wn = end - w0 < kCap ? end - w0 : kCap;
    __syncthreads();  // the previous readers are done with `cand`
    constexpr int U = MPA_GRID_GATHER_U;
    for (int j0 = lane; j0 < wn; j0 += 64 * U) {
      float4 t[U];  // (unconditional loads of a clamped position: conditional ones send the array to scratch)
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = trec[w0 + (j0 + 64 * u < wn ? j0 + 64 * u : wn - 1)];
#pragma unroll
      for (int u = 0; u < U; ++u) pin_record(t[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (j0 + 64 * u < wn) {
          cand[j0 + 64 * u] = gate_record(t[u]);
          if (MPA_GRID_FMA_GATE) cidx[j0 + 64 * u] = __float_as_int(t[u].w);
        }
    }
    if (lane < T) {
      const float inf = __builtin_inff();
      cand[wn + lane] = make_float4(inf, inf, inf, MPA_GRID_FMA_GATE ? inf : __int_as_float(0x7fffffff));
      if (MPA_GRID_FMA_GATE) cidx[wn + lane] = 0x7fffffff;
    }
    __syncthreads();
    scan_cand<TC>(s, cand, cidx, wn);
  }
}

template <int TC>
__device__ __forceinline__ void scan_batch(LaneState& s, const float4* __restrict__ trec, int rb, int re,
                                           float4* __restrict__ cand, int* __restrict__ sidx, int* __restrict__ cidx) {
  const int lane = threadIdx.x;
  int len = re > rb ? re - rb : 0;
#ifdef MPA_GRID_STATS
  {
    int tot = len;
    for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off, 64);
    int lng = len > kLongRange ? len : 0;
    for (int off = 32; off >= 1; off >>= 1) lng += __shfl_xor(lng, off, 64);
    MPA_STAT(2, 1);
    MPA_STAT(3, tot);
    MPA_STAT(4, lng);
  }
#endif
  // long ranges are fetched by the whole wave, one after the other; in the gather below they would unbalance the copy
  unsigned long long big = __ballot(len > kLongRange);
  while (big) {
    const int l = __builtin_ctzll(big);
    big &= big - 1;
    scan_range_coop<TC>(s, trec, __builtin_amdgcn_readlane(rb, l), __builtin_amdgcn_readlane(re, l), cand, cidx);
  }
  if (len > kLongRange) len = 0;
  const int incl = wave_prefix_sum(len);
  const int total = __builtin_amdgcn_readlane(incl, 63), off0 = incl - len;
  constexpr int T = 4 * TC, kCap = kCand - T;  // (4x: a split-4 scan reads up to that far past the end)
  for (int w0 = 0; w0 < total; w0 += kCap) {  // windows of the concatenated list that fit the LDS buffer
    const int wn = total - w0 < kCap ? total - w0 : kCap;
    const int lo = off0 > w0 ? off0 : w0, hi = off0 + len < w0 + wn ? off0 + len : w0 + wn;
    const int cnt = hi > lo ? hi - lo : 0;
    __syncthreads();  // the previous window's readers are done with `cand` / `sidx` (one wave per block: cheap)
    // balanced gather: every lane first lists the record indices of its own (short) range in LDS, then the wave
    // fetches the concatenated list position by position — 64 records per instruction, several in flight — instead
    // of each lane walking its own range two records per memory round trip (round 5, again: per-lane copies with four
    // records in flight per round and wave-uniform rounds: 0.253 vs 0.230 ms)
    for (int k = 0; k < cnt; ++k) sidx[lo - w0 + k] = rb + (lo - off0) + k;
    __syncthreads();
    constexpr int U = MPA_GRID_GATHER_U;
    for (int j0 = lane; j0 < wn; j0 += 64 * U) {
      float4 t[U];
#pragma unroll
      for (int u = 0; u < U; ++u) t[u] = trec[sidx[j0 + 64 * u < wn ? j0 + 64 * u : wn - 1]];
#pragma unroll
      for (int u = 0; u < U; ++u) pin_record(t[u]);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (j0 + 64 * u < wn) {
          cand[j0 + 64 * u] = gate_record(t[u]);
          if (MPA_GRID_FMA_GATE) cidx[j0 + 64 * u] = __float_as_int(t[u].w);
        }
    }
    if (lane < T) {  // pad the last chunk of 8 with records that can never win
      const float inf = __builtin_inff();
      cand[wn + lane] = make_float4(inf, inf, inf, MPA_GRID_FMA_GATE ? inf : __int_as_float(0x7fffffff));
      if (MPA_GRID_FMA_GATE) cidx[wn + lane] = 0x7fffffff;
    }
    __syncthreads();
    scan_cand<TC>(s, cand, cidx, wn);
  }
}

// grid = (768 persistent waves per (sample, dir) on average), block 64.  blockIdx.y = b*2 + dir; dir 0: shape 1 queries
// against shape 2 targets.
// GENERIC (the operator of chamfer.hip: two plain clouds [B, P, 3] and [B, N, 3] per sample — P, N are the point counts
// there, no valid flags, int64 indices out) differs from the fused loss's search in what the pruning may assume:
//   * points outside the (outlier-trimmed) grid sit in its border cells, so a border cell / row is unbounded outwards;
//   * the queries of a wave are boxed by their actual coordinates (a wave-wide min / max), not by their super-cell;
//   * all geometry is relative to the grid origin (the origin may be large against the cell size).
// (GENERIC compiles to 84 VGPRs = 5 waves per SIMD against the fused loss's 78 = 6; forcing 6 costs a spill and measured
// the same: 0.526 vs 0.529 ms per [32, 20000, 3]^2 call)
template <bool GENERIC, typename IdxT>
__global__ __launch_bounds__(64) void grid_search_kernel(
    const float* __restrict__ valids, const float* __restrict__ S1, const float* __restrict__ S2, int P,
    int N, const GridParams* __restrict__ params, const float4* __restrict__ records,
    const int* __restrict__ starts, const int* __restrict__ batches, const int* __restrict__ worklist,
    int work_stride, int rec_stride, float* __restrict__ dist1, float* __restrict__ dist2, IdxT* __restrict__ idx1,
    IdxT* __restrict__ idx2, const XcdPlan* __restrict__ plan) {
  __shared__ float4 cand[kCand];
  __shared__ int sidx[kCand];  // record index of every position of the current window
  __shared__ int cidx[MPA_GRID_FMA_GATE ? kCand : 1];  // original point index of every candidate of the window
  // candidates per step of the LDS scan: the operator's instantiation carries a few more registers (relative geometry, border
  // cells) and reaches six waves per SIMD with 3 (77 registers; 4: 81)
  constexpr int TC = GENERIC ? MPA_GRID_CAND_CHUNK_GENERIC : MPA_GRID_CAND_CHUNK;
  MPA_TICK_INIT();
  // block -> (sample, direction, wave): from the XCD-aware plan (grid_assign_plan) or, without one, waves
  // blockIdx.x of pair blockIdx.y
  int pair = (int)blockIdx.y, wid = (int)blockIdx.x, wstride = (int)gridDim.x;
  if (plan != nullptr) {
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, n = plan->cnt[xcd];
    pair = -1;
    for (int i = 0; i < n; ++i) {  // (wave-uniform scalar loads; <= 2B / 8 entries)
      const int f0 = plan->first[xcd][i], f1 = plan->first[xcd][i + 1];
      if (slot >= f0 && slot < f1) {
        pair = plan->lst[xcd][i];
        wid = slot - f0;
        wstride = f1 - f0;
      }
    }
    if (pair < 0) return;
  }
  const int b = pair >> 1, dir = pair & 1;
  const int qc = dir, tc = 1 - dir;  // query / target shape
  const GridParams& g = params[b];  // by reference: uniform address -> scalar loads (a by-value copy indexed with a
                                    // runtime shape index lands in scratch memory)
  const int qslot = (b * 2 + qc) * 2 + 1, tslot = (b * 2 + tc) * 2 + 0;
  const float4* qrec = records + (long long)qslot * rec_stride;
  const float4* trec = records + (long long)tslot * rec_stride;
  const int* tst = starts + (long long)tslot * kStartStride;
  const int* qst = starts + (long long)qslot * kStartStride;
  const long long qstride = GENERIC ? (long long)(dir == 0 ? P : N) : (long long)P * N;  // points per sample, query side
  float* dout = (dir == 0 ? dist1 : dist2) + (long long)b * qstride;
  IdxT* iout = (dir == 0 ? idx1 : idx2) + (long long)b * qstride;
  const int* bst = batches + (long long)qslot * kStartStride;
  const int total_work = bst[g.nsuper];
  const int lane = threadIdx.x;
  // origin of the geometry below: absolute coordinates for the fused loss, relative to the grid origin otherwise
  const float OX = GENERIC ? 0.0f : g.ox, OY = GENERIC ? 0.0f : g.oy, OZ = GENERIC ? 0.0f : g.oz;
  // the target shape's own cell bounding box: rows and cells outside it are empty, and a query far from a compact
  // target would otherwise walk hundreds of empty rows before reaching it
  const int* tbox = g.tb[tc];
  const int tx0 = tbox[0], tx1 = tbox[1], ty0 = tbox[2], ty1 = tbox[3], tz0 = tbox[4], tz1 = tbox[5];
  const float slack = 1e-3f * g.h;
  // padded parts: one representative target each (index p*N), the same for every work item; lane p holds part p's
  float px = 0.0f, py = 0.0f, pz = 0.0f;
  bool pad = false;
  if constexpr (!GENERIC) {
    const float* tcloud = (tc == 0 ? S1 : S2) + 3LL * b * P * N;
    const float* vb = valids + (long long)b * P;
    if (lane < P && vb[lane] == 0.0f) {
      pad = true;
      const float* t = tcloud + 3LL * lane * N;
      px = t[0];
      py = t[1];
      pz = t[2];
    }
  }
  const unsigned long long padmask = GENERIC ? 0ull : __ballot(pad);
  MPA_TICK(8);

  for (int work = wid; work < total_work; work += wstride) {  // persistent walk over the work list
    const int sc = worklist[(long long)qslot * work_stride + work];  // super-cell with bst[sc] <= work < bst[sc+1]
    const int qb = qst[sc] + (work - bst[sc]) * kBatch, q_end = qst[sc + 1];
    const int sx = sc % g.sgx, sy = (sc / g.sgx) % g.sgy, sz = sc / (g.sgx * g.sgy);
    LaneState s;
    s.split = q_end - qb <= 16 ? 4 : (q_end - qb <= 32 ? 2 : 1);
    const int qi = qb + (lane & (64 / s.split - 1));
    const bool has = qi < q_end;
    MPA_STAT(0, 1);
    { const int nact = __popcll(__ballot(has)); MPA_STAT(1, nact); }
    const float4 qr = qrec[has ? qi : q_end - 1];
    s.X = qr.x;
    s.Y = qr.y;
    s.Z = qr.z;
    const int qflat = __float_as_int(qr.w);
    s.best = 1e32f;
    s.bidx = 0x7fffffff;
    s.qnlo = gate_norm_lo(s.X, s.Y, s.Z);
    MPA_TICK(9);
    // the queries lie in the super-cell box (inflated by the binning slack)
    float bx0 = OX + (float)(kS * sx) * g.h - slack, bx1 = OX + (float)(kS * sx + kS) * g.h + slack;
    float by0 = OY + (float)(kS * sy) * g.h - slack, by1 = OY + (float)(kS * sy + kS) * g.h + slack;
    float bz0 = OZ + (float)(kS * sz) * g.h - slack, bz1 = OZ + (float)(kS * sz + kS) * g.h + slack;
    if constexpr (GENERIC) {  // ... unless they were binned from outside the grid: box them by their coordinates (idle
                              // lanes shadow a real query)
      const float rx = s.X - g.ox, ry = s.Y - g.oy, rz = s.Z - g.oz;
      bx0 = -wave_max(-rx) - slack, bx1 = wave_max(rx) + slack;
      by0 = -wave_max(-ry) - slack, by1 = wave_max(ry) + slack;
      bz0 = -wave_max(-rz) - slack, bz1 = wave_max(rz) + slack;
    }
    // seed: the super-cell grown by one fine cell per side ((kS+2)^2 rows, lane = row), clipped to the target box
    const int x0 = clampi(kS * sx - 1, 0, g.gx - 1), x1 = clampi(kS * sx + kS, 0, g.gx - 1);
    const int y0 = clampi(kS * sy - 1, 0, g.gy - 1), y1 = clampi(kS * sy + kS, 0, g.gy - 1);
    const int z0 = clampi(kS * sz - 1, 0, g.gz - 1), z1 = clampi(kS * sz + kS, 0, g.gz - 1);
    {
      constexpr int kSeedW = kS + 2;  // rows per axis of the seed region
      static_assert(kSeedW * kSeedW <= 64, "one lane per seed row");
      const int z = z0 + lane / kSeedW, y = y0 + lane % kSeedW;
      const int xa = x0 < tx0 ? tx0 : x0, xb = x1 > tx1 ? tx1 : x1;
      const bool ok = lane < kSeedW * kSeedW && z <= z1 && y <= y1 && z >= tz0 && z <= tz1 && y >= ty0 && y <= ty1 && xa <= xb;
      const int row = (z * g.gy + y) * g.gx;
      const int rb = ok ? tst[row + xa] : 0, re = ok ? tst[row + xb + 1] : 0;
      scan_batch<TC>(s, trec, rb, re, cand, sidx, cidx);
    }
    // sweep: every cell whose box can hold a point closer than the worst best-distance of this wave.  Rows (y, z)
    // are visited nearest-first, as square rings around the super-cell's own kS x kS rows (lane = row of a ring), so
    // the bound tightens early; a ring whose nearest row is already farther than the bound ends the sweep.  Rings 0
    // and 1 overlap the seed: only the cells left and right of it are new there (two ranges per row).  Every batch
    // costs two dependent memory round trips (row offsets, then records), so both flanks of a seeded ring share one
    // batch.
    merge_halves(s);
    float bound = wave_max(s.best) * 1.00001f;
    MPA_STAT(6, bound > 1e31f ? 1 : 0);
    MPA_TICK(10);
    const int yc0 = kS * sy, yc1 = kS * sy + kS - 1, zc0 = kS * sz, zc1 = kS * sz + kS - 1;
    const int rmax = max(max(yc0, g.gy - 1 - yc1), max(zc0, g.gz - 1 - zc1));
    auto ring_rows = [&](int r) { return r == 0 ? kS * kS : 2 * (kS + 2 * r) + 2 * (kS + 2 * r - 2); };
    // record range of row i of ring r (side: which flank of the seed, rings 0 and 1 only); empty if out of reach
    auto row_range = [&](int r, int side, int i, bool live, int& rb, int& re) {
      rb = re = 0;
      const int zlo = zc0 - r, zhi = zc1 + r, ylo = yc0 - r, yhi = yc1 + r;
      const int W = yhi - ylo + 1;
      int z, y;
      if (r == 0) {
        z = zlo + i / W;
        y = ylo + i % W;
      } else if (i < W) {
        z = zlo;
        y = ylo + i;
      } else if (i < 2 * W) {
        z = zhi;
        y = ylo + i - W;
      } else {
        z = zlo + 1 + ((i - 2 * W) >> 1);
        y = ((i - 2 * W) & 1) ? yhi : ylo;
      }
      if (!live || z < tz0 || z > tz1 || y < ty0 || y > ty1) return;
      float cz0 = OZ + (float)z * g.h - slack, cz1 = OZ + (float)(z + 1) * g.h + slack;
      float cy0 = OY + (float)y * g.h - slack, cy1 = OY + (float)(y + 1) * g.h + slack;
      if constexpr (GENERIC) {  // border rows hold whatever lies beyond them
        const float inf = __builtin_inff();
        cz0 = z == 0 ? -inf : cz0, cz1 = z == g.gz - 1 ? inf : cz1;
        cy0 = y == 0 ? -inf : cy0, cy1 = y == g.gy - 1 ? inf : cy1;
      }
      const float dz = gap(bz0, bz1, cz0, cz1), dy = gap(by0, by1, cy0, cy1);
      const float rem = bound - dz * dz - dy * dy;
      if (rem <= 0.0f) return;
      // cells x with gap_x(x)^2 < rem: an interval around the super-cell
      const float reach = __builtin_sqrtf(rem) + slack;
      int xa, xb;
      if constexpr (GENERIC) {  // (clamped as floats: far queries put these beyond the int range; the border cells, which
                                // hold everything beyond the grid, are reached exactly when the clamped index says so)
        xa = clampi((int)__builtin_amdgcn_fmed3f(__builtin_floorf((bx0 - reach) * g.inv_h), -1.0f, 1024.0f), 0, g.gx - 1);
        xb = clampi((int)__builtin_amdgcn_fmed3f(__builtin_floorf((bx1 + reach) * g.inv_h), -1.0f, 1024.0f), 0, g.gx - 1);
      } else {
        xa = clampi((int)__builtin_floorf((bx0 - reach - g.ox) * g.inv_h), 0, g.gx - 1);
        xb = clampi((int)__builtin_floorf((bx1 + reach - g.ox) * g.inv_h), 0, g.gx - 1);
      }
      if (bound > 1e31f) {  // nothing found yet: the whole row
        xa = 0;
        xb = g.gx - 1;
      }
      xa = xa < tx0 ? tx0 : xa;
      xb = xb > tx1 ? tx1 : xb;
      const bool seeded = r <= 1 && z >= z0 && z <= z1 && y >= y0 && y <= y1;
      if (seeded) {  // the seed covered [x0, x1] of this row
        if (side == 0) xb = xb < x0 - 1 ? xb : x0 - 1;
        else xa = xa > x1 + 1 ? xa : x1 + 1;
      } else if (side == 1) {
        return;
      }
      if (xa <= xb) {
        const int row = (z * g.gy + y) * g.gx;
        rb = tst[row + xa];
        re = tst[row + xb + 1];
      }
    };
    static_assert(2 * kS * kS + 2 * (2 * (kS + 2) + 2 * kS) <= 64, "both flanks of rings 0 and 1 fit one batch");
    if (bound <= 1e31f && rmax >= 1) {
      // the seed left a bound: rings 0 and 1, both flanks of every row, are ONE batch under it (lanes [0, 8): ring 0,
      // [8, 32): ring 1) — a batch is two dependent memory round trips and ~260 VALU instructions of bookkeeping,
      // which ring 1's slightly staler bound does not cost
      constexpr int kL0 = 2 * kS * kS;
      const int rr = lane >= kL0 ? 1 : 0, li = lane - rr * kL0;
      int rb, re;
      row_range(rr, li & 1, li >> 1, li < 2 * ring_rows(rr), rb, re);
      scan_batch<TC>(s, trec, rb, re, cand, sidx, cidx);
      merge_halves(s);
      bound = wave_max(s.best) * 1.00001f;
    } else {
      for (int r = 0; r <= 1 && r <= rmax; ++r) {  // rings 0 and 1: both flanks of every row in one batch
        int rb, re;
        row_range(r, lane & 1, lane >> 1, lane < 2 * ring_rows(r), rb, re);
        scan_batch<TC>(s, trec, rb, re, cand, sidx, cidx);
        merge_halves(s);
        bound = wave_max(s.best) * 1.00001f;  // (ring 1 must see the bound ring 0 found: without one, rows are whole)
      }
    }
    MPA_TICK(11);
    for (int r = 2; r <= rmax; ++r) {
      const float ring_gap = (float)(r - 1) * g.h - 2.0f * slack;
      if (ring_gap * ring_gap >= bound) break;
      const int nrows = ring_rows(r);
      for (int i0 = 0; i0 < nrows; i0 += 64) {
        int rb, re;
        row_range(r, 0, i0 + lane, i0 + lane < nrows, rb, re);
        MPA_STAT(5, 1);
        scan_batch<TC>(s, trec, rb, re, cand, sidx, cidx);
      }
      merge_halves(s);
      bound = wave_max(s.best) * 1.00001f;  // (two rings per batch were tried: the staler bound costs what the saved
                                            // round trips gain; round 4: so does requesting batch k + 1's row offsets
                                            // before batch k is scanned — a software pipeline of the sweep with the stale
                                            // bound's superset intervals: 0.297 vs 0.266 ms, and 84 registers = 5 waves)
    }
    MPA_TICK(12);
    {  // padded parts' representatives, in part order (uniform loop over the set bits)
      unsigned long long m = padmask;
      while (m) {
        const int p = __builtin_ctzll(m);
        m &= m - 1;
        auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
        consider(s, rl(px, p), rl(py, p), rl(pz, p), p * N);  // (p is wave-uniform: a register read, not an LDS permute)
      }
    }
    if (has && lane < 64 / s.split) {  // (every group holds the merged result; the padded parts' representatives
                                       // were considered by all of them)
      dout[qflat] = s.best;
      iout[qflat] = s.bidx == 0x7fffffff ? (IdxT)-1 : (IdxT)s.bidx;
    }
    MPA_TICK(13);
  }  // work loop
  MPA_TICK_END();
}

// per-part sums of the distances, written where the finalize kernel expects the tile sums (tile 0 of each part;
// the other tiles were cleared).  grid = (B*P, 2), block 256; fixed-order reduction.
__global__ __launch_bounds__(256) void grid_part_sum_kernel(const float* __restrict__ valids,
                                                            const float* __restrict__ dist1,
                                                            const float* __restrict__ dist2, int N, int tiles,
                                                            float* __restrict__ tile_sums, int P,
                                                            const int* __restrict__ route) {
  __shared__ float red[4];
  const int m = blockIdx.x, dir = blockIdx.y;
  if (valids[m] == 0.0f) return;
  if (route != nullptr && route[m / P] == 0) return;  // (searched by the leaf search)
  const float* d = (dir == 0 ? dist1 : dist2) + (long long)m * N;
  float s = 0.0f;
  for (int n = threadIdx.x; n < N; n += 256) s += d[n];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    tile_sums[(long long)dir * gridDim.x * tiles + (long long)m * tiles] = (red[0] + red[1]) + (red[2] + red[3]);
}

}  // namespace

int64_t grid_workspace_floats(int64_t B, int64_t P, int64_t N) {
  const int64_t rec = (P * N + 8) * 4;  // float4 records (+ sentinels) per slot
  // + 2 distance arrays + boxes; a multiple of 4 floats: the leaf search's float4 arrays start right behind this region
  // (odd B * P * N used to leave them 8-byte aligned)
  return (4 * B * rec + B * (int64_t)(sizeof(GridParams) / 4) + 2 * B * P * N + 12 * B * P + 3) / 4 * 4;
}
int64_t grid_workspace_ints(int64_t B) {  // starts, batches, work list, wave table, ticket
  return 2 * 4 * B * (int64_t)kStartStride + 4 * B * (int64_t)kWorkStride + (int64_t)(sizeof(XcdPlan) / 4) + 16;
}
float* grid_bbox(float* fws, int64_t B, int64_t P, int64_t N) {
  const int64_t rec = (P * N + 8) * 4;
  return fws + 4 * B * rec + B * (int64_t)(sizeof(GridParams) / 4) + 2 * B * P * N;
}
unsigned* grid_ticket(int32_t* iws, int64_t B) {
  return reinterpret_cast<unsigned*>(iws + 2 * 4 * B * (int64_t)kStartStride + 4 * B * (int64_t)kWorkStride +
                                     (int64_t)(sizeof(XcdPlan) / 4));
}

int launch_grid_shape_search(const float* valids, const float* S1, const float* S2, int64_t B, int64_t P,
                             int64_t N, int tiles, float* fws, int32_t* iws, int32_t* idx1, int32_t* idx2,
                             float* tile_sums, hipEvent_t before_search, hipEvent_t after_search, hipStream_t s,
                             const int* route, int phases) {
  const int rec_stride = (int)(P * N + 8);
  float4* records = reinterpret_cast<float4*>(fws);
  GridParams* params = reinterpret_cast<GridParams*>(fws + 4 * B * (int64_t)rec_stride * 4);
  float* dist1 = reinterpret_cast<float*>(params + B);
  float* dist2 = dist1 + B * P * N;
  int* starts = iws;
  int* batches = iws + 4 * B * (int64_t)kStartStride;
  int* worklist = batches + 4 * B * (int64_t)kStartStride;
  // persistent waves (768 per (sample, direction) on average) walk a pair's work list of (super-cell, 64-query batch) items
  XcdPlan* plan = reinterpret_cast<XcdPlan*>(worklist + 4 * B * (int64_t)kWorkStride);
  const bool xcd_table = MPA_GRID_XCD && 2 * B >= 8 && 2 * B <= kMaxPairs;
  const int nwaves = (int)(MPA_GRID_WAVES * 2 * B);
  if (phases & 1)
    hipLaunchKernelGGL(grid_sort_kernel, dim3((unsigned)(4 * B)), dim3(1024), 0, s, valids, S1, S2, (int)P, (int)N,
                       (const float*)grid_bbox(fws, B, P, N), params, starts, batches, worklist, records, rec_stride,
                       grid_ticket(iws, B), xcd_table ? plan : (XcdPlan*)nullptr, nwaves, route);
  if (phases & 2) {
    if (before_search != nullptr) (void)hipEventRecord(before_search, s);
    if (xcd_table)
      hipLaunchKernelGGL((grid_search_kernel<false, int>), dim3((unsigned)nwaves), dim3(64), 0, s, valids, S1, S2, (int)P,
                         (int)N, params, records, starts, batches, worklist, kWorkStride, rec_stride, dist1, dist2, idx1, idx2,
                         (const XcdPlan*)plan);
    else
      hipLaunchKernelGGL((grid_search_kernel<false, int>), dim3(MPA_GRID_WAVES, (unsigned)(2 * B)), dim3(64), 0, s, valids, S1,
                         S2, (int)P, (int)N, params, records, starts, batches, worklist, kWorkStride, rec_stride, dist1, dist2,
                         idx1, idx2, (const XcdPlan*)nullptr);
    if (after_search != nullptr) (void)hipEventRecord(after_search, s);
  }
  if (phases & 4)
    hipLaunchKernelGGL(grid_part_sum_kernel, dim3((unsigned)(B * P), 2), dim3(256), 0, s, valids, dist1, dist2, (int)N,
                       tiles, tile_sums, (int)P, route);
  return MPA_OK;
}

// ---- the generic operator's entry (chamfer.hip) ---------------------------------------------------------------------------
namespace {
struct CloudWs {
  float4* records;
  GridParams* params;
  int *starts, *batches, *worklist, *fallback, *heads;
  CloudBox* boxes;
  XcdPlan* plan;
  unsigned* ticket;
  int rec_stride, work_stride;
  int64_t bytes;
};
CloudWs cloud_ws(void* base, int64_t B, int64_t n1, int64_t n2) {
  const int64_t nmax = n1 > n2 ? n1 : n2;
  CloudWs w;
  w.rec_stride = (int)(nmax + 8);
  w.work_stride = (int)(kMaxSuper + nmax / kBatch + 64);
  char* p = static_cast<char*>(base);
  auto take = [&p](int64_t bytes) {
    char* q = p;
    p += (bytes + 255) / 256 * 256;
    return q;
  };
  w.records = reinterpret_cast<float4*>(take(4 * B * (int64_t)w.rec_stride * 16));
  w.params = reinterpret_cast<GridParams*>(take(B * (int64_t)sizeof(GridParams)));
  w.starts = reinterpret_cast<int*>(take(4 * B * (int64_t)kStartStride * 4));
  w.batches = reinterpret_cast<int*>(take(4 * B * (int64_t)kStartStride * 4));
  w.worklist = reinterpret_cast<int*>(take(4 * B * (int64_t)w.work_stride * 4));
  w.plan = reinterpret_cast<XcdPlan*>(take(sizeof(XcdPlan)));
  w.ticket = reinterpret_cast<unsigned*>(take(64));
  w.fallback = reinterpret_cast<int*>(take(B * 4));
  w.heads = reinterpret_cast<int*>(take(2 * B * nmax * 4));
  w.boxes = reinterpret_cast<CloudBox*>(take(2 * B * (int64_t)sizeof(CloudBox)));
  w.bytes = p - static_cast<char*>(base);
  return w;
}
}  // namespace

int64_t cloud_grid_workspace_bytes(int64_t B, int64_t n1, int64_t n2) { return cloud_ws(nullptr, B, n1, n2).bytes; }

bool cloud_grid_supported(int64_t B, int64_t n1, int64_t n2) {
  // the sort block indexes points with ints and keeps one histogram in LDS; 2^22 points per cloud is far beyond what the
  // 32768-cell grid prunes well, but stays correct
  return B >= 1 && n1 >= 1 && n2 >= 1 && n1 <= (1 << 22) && n2 <= (1 << 22) && 4 * B < (1 << 30) / kStartStride;
}

int launch_cloud_grid_search(const float* xyz1, const float* xyz2, int64_t B, int64_t n1, int64_t n2, float* dist1,
                             int64_t* idx1, float* dist2, int64_t* idx2, void* workspace, const int** fallback,
                             hipStream_t s) {
  const CloudWs w = cloud_ws(workspace, B, n1, n2);
  const bool xcd_table = MPA_GRID_XCD && 2 * B >= 8 && 2 * B <= kMaxPairs;
  const int nwaves = (int)(MPA_GRID_WAVES * 2 * B);
  zero_words_async(w.ticket, 16, s);
  hipLaunchKernelGGL(cloud_stats_kernel, dim3(2, (unsigned)B), dim3(1024), 0, s, xyz1, xyz2, (int)n1, (int)n2,
                     w.rec_stride - 8, w.boxes, w.heads);
  hipLaunchKernelGGL(cloud_sort_kernel, dim3((unsigned)(4 * B)), dim3(1024), 0, s, xyz1, xyz2, (int)n1, (int)n2,
                     (const CloudBox*)w.boxes, w.params, w.fallback, w.starts, w.batches, w.worklist, w.work_stride,
                     w.records, w.rec_stride, w.ticket, xcd_table ? w.plan : (XcdPlan*)nullptr, nwaves);
  long long* i1 = reinterpret_cast<long long*>(idx1);
  long long* i2 = reinterpret_cast<long long*>(idx2);
  if (xcd_table)
    hipLaunchKernelGGL((grid_search_kernel<true, long long>), dim3((unsigned)nwaves), dim3(64), 0, s, (const float*)nullptr,
                       xyz1, xyz2, (int)n1, (int)n2, w.params, w.records, w.starts, w.batches, w.worklist, w.work_stride,
                       w.rec_stride, dist1, dist2, i1, i2, (const XcdPlan*)w.plan);
  else
    hipLaunchKernelGGL((grid_search_kernel<true, long long>), dim3(MPA_GRID_WAVES, (unsigned)(2 * B)), dim3(64), 0, s,
                       (const float*)nullptr, xyz1, xyz2, (int)n1, (int)n2, w.params, w.records, w.starts, w.batches,
                       w.worklist, w.work_stride, w.rec_stride, dist1, dist2, i1, i2, (const XcdPlan*)nullptr);
  *fallback = w.fallback;
  return MPA_OK;
}

void launch_cloud_copy_runs(int64_t B, int64_t n1, int64_t n2, float* dist1, int64_t* idx1, float* dist2, int64_t* idx2,
                            void* workspace, hipStream_t s) {
  const CloudWs w = cloud_ws(workspace, B, n1, n2);
  const int64_t nmax = n1 > n2 ? n1 : n2;
  hipLaunchKernelGGL(cloud_copy_runs_kernel, dim3((unsigned)((nmax + 255) / 256), (unsigned)(2 * B)), dim3(256), 0, s,
                     (const int*)w.heads, (int)n1, (int)n2, (int)nmax, dist1, dist2, reinterpret_cast<long long*>(idx1),
                     reinterpret_cast<long long*>(idx2));
}

}  // namespace mpa

#ifdef MPA_GRID_STATS
extern "C" int mpa_debug_grid_stats(unsigned long long* out24, int reset) {
  if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(mpa::g_grid_stats), 24 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[24] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(mpa::g_grid_stats), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
