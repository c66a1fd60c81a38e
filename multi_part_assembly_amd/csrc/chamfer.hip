// Chamfer nearest-neighbour forward / backward for gfx950 (MI355X) — the generic operator behind
// mpa_chamfer_forward / mpa_chamfer_backward (include/mpa_hip.h).
//
// Replaces ChamferForwardKernel / ChamferBackwardKernel of the reference
// (multi_part_assembly/utils/chamfer/cuda/chamfer_kernel.cu:32-95, :175-210).  Not a translation:
// the reference stages 512-point target tiles through shared memory with two block barriers per
// tile and one query per thread; here the scan is the scalar-cache / packed-fp32 design of
// chamfer_core.h (Q queries per lane, targets in SGPRs, no LDS, no barriers), both directions of the
// bidirectional search run in ONE launch (blockIdx.y = direction), and the arithmetic is pinned so
// that results are bit-identical to the reference's CPU ground truth.
#include "assembly_internal.h"
#include "chamfer_core.h"
#include "common.h"

namespace {

constexpr int kThreads = 256;

// ---- fp32: NNScan core ---------------------------------------------------------------------------
template <int Q, int MODE, int THREADS>
__device__ __forceinline__ void nn_search_f32(const float* __restrict__ qa,
                                              const float* __restrict__ tb, int na, int nb,
                                              int qbase, float* __restrict__ dist,
                                              long long* __restrict__ idx) {
  mpa::NNScan<Q, MODE> scan;
  scan.init();
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = qbase + q * THREADS + (int)threadIdx.x;
    const int ic = i < na ? i : na - 1;  // idle lanes shadow the last query, never stored
    scan.set_query(q, qa[3 * (long long)ic + 0], qa[3 * (long long)ic + 1], qa[3 * (long long)ic + 2]);
  }
  scan.scan_range(tb, 0, nb, 0);
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int i = qbase + q * THREADS + (int)threadIdx.x;
    if (i < na) {
      dist[i] = scan.best[q];
      idx[i] = (long long)scan.bidx[q];
    }
  }
}

// grid.x = batch * blocks_per_cloud (blocks_per_cloud sized for the LARGER cloud), grid.y = 2.
template <int Q, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void chamfer_nn_kernel(
    const float* __restrict__ xyz1, const float* __restrict__ xyz2, int n1, int n2,
    int blocks_per_cloud, float* __restrict__ dist1, long long* __restrict__ idx1,
    float* __restrict__ dist2, long long* __restrict__ idx2, const int* __restrict__ only) {
  const int b = blockIdx.x / blocks_per_cloud;
  if (only != nullptr && only[b] == 0) return;  // (behind the pruned search: only the samples it handed back)
  const int qbase = (blockIdx.x % blocks_per_cloud) * (THREADS * Q);
  const long long o1 = (long long)b * n1, o2 = (long long)b * n2;
  if (blockIdx.y == 0) {
    if (qbase >= n1) return;
    nn_search_f32<Q, MODE, THREADS>(xyz1 + 3 * o1, xyz2 + 3 * o2, n1, n2, qbase, dist1 + o1, idx1 + o1);
  } else {
    if (qbase >= n2) return;
    nn_search_f32<Q, MODE, THREADS>(xyz2 + 3 * o2, xyz1 + 3 * o1, n2, n1, qbase, dist2 + o2, idx2 + o2);
  }
}

// ---- fp64: plain one-query-per-lane scan (gradcheck path; not performance relevant) ----------------
__global__ __launch_bounds__(kThreads) void chamfer_nn_kernel_f64(
    const double* __restrict__ xyz1, const double* __restrict__ xyz2, int n1, int n2,
    int blocks_per_cloud, double* __restrict__ dist1, long long* __restrict__ idx1,
    double* __restrict__ dist2, long long* __restrict__ idx2) {
  const int b = blockIdx.x / blocks_per_cloud;
  const int i = (blockIdx.x % blocks_per_cloud) * kThreads + threadIdx.x;
  const bool fwd = blockIdx.y == 0;
  const int na = fwd ? n1 : n2, nb = fwd ? n2 : n1;
  if (i >= na) return;
  const double* qa = (fwd ? xyz1 : xyz2) + 3 * (long long)b * na;
  const double* tb = (fwd ? xyz2 : xyz1) + 3 * (long long)b * nb;
  const double x = qa[3 * i], y = qa[3 * i + 1], z = qa[3 * i + 2];
  double best = 1e32;
  long long arg = -1;
  for (int j = 0; j < nb; ++j) {
    const double dx = x - tb[3 * j], dy = y - tb[3 * j + 1], dz = z - tb[3 * j + 2];
    const double d = (dx * dx + dy * dy) + dz * dz;
    if (d < best) {
      best = d;
      arg = j;
    }
  }
  (fwd ? dist1 : dist2)[(long long)b * na + i] = best;
  (fwd ? idx1 : idx2)[(long long)b * na + i] = arg;
}

// Backward.  One thread per query point of one direction (blockIdx.y); the gather half is a plain
// accumulate into the query's own gradient row, the scatter half an fp32 atomic into the matched
// point's row (contract: include/mpa_hip.h).
template <typename S>
__global__ __launch_bounds__(kThreads) void chamfer_grad_kernel(
    const S* __restrict__ g1, const S* __restrict__ g2, const S* __restrict__ xyz1,
    const S* __restrict__ xyz2, const long long* __restrict__ idx1,
    const long long* __restrict__ idx2, long long total1, long long total2, int n1, int n2,
    S* gxyz1, S* gxyz2) {
  const bool fwd = blockIdx.y == 0;
  const long long total = fwd ? total1 : total2;
  const int na = fwd ? n1 : n2, nb = fwd ? n2 : n1;
  const S* g = fwd ? g1 : g2;
  const S* pa = fwd ? xyz1 : xyz2;
  const S* pb = fwd ? xyz2 : xyz1;
  const long long* ix = fwd ? idx1 : idx2;
  S* ga = fwd ? gxyz1 : gxyz2;
  S* gb = fwd ? gxyz2 : gxyz1;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (long long)gridDim.x * kThreads) {
    const long long j = ix[i];
    if (j < 0 || j >= nb) continue;
    const long long b = i / na;
    const long long t = (b * nb + j) * 3;
    const S s = g[i] * (S)2;
    const S gx = s * (pa[3 * i + 0] - pb[t + 0]);
    const S gy = s * (pa[3 * i + 1] - pb[t + 1]);
    const S gz = s * (pa[3 * i + 2] - pb[t + 2]);
    atomicAdd(ga + 3 * i + 0, gx);
    atomicAdd(ga + 3 * i + 1, gy);
    atomicAdd(ga + 3 * i + 2, gz);
    atomicAdd(gb + t + 0, -gx);
    atomicAdd(gb + t + 1, -gy);
    atomicAdd(gb + t + 2, -gz);
  }
}

template <int Q, int MODE, int THREADS>
void launch_nn(const float* xyz1, const float* xyz2, int64_t batch, int n1, int n2, float* dist1,
               int64_t* idx1, float* dist2, int64_t* idx2, const int* only, hipStream_t s) {
  const int nmax = n1 > n2 ? n1 : n2;
  const int bpc = (nmax + THREADS * Q - 1) / (THREADS * Q);
  dim3 grid((unsigned)(batch * bpc), 2, 1);
  hipLaunchKernelGGL((chamfer_nn_kernel<Q, MODE, THREADS>), grid, dim3(THREADS), 0, s, xyz1, xyz2,
                     n1, n2, bpc, dist1, (long long*)idx1, dist2, (long long*)idx2, only);
}

template <int MODE>
void launch_nn_sized(const float* xyz1, const float* xyz2, int64_t batch, int n1, int n2,
                     float* dist1, int64_t* idx1, float* dist2, int64_t* idx2, hipStream_t s,
                     const int* only = nullptr) {
  const int nmax = n1 > n2 ? n1 : n2;
  // 4 queries per lane x 256 lanes for real clouds; 2 x 64 for the <=256-point clouds of part
  // matching (reference base_model.py:163-173 sub-samples to 100 points).
  if (nmax <= 256) launch_nn<2, MODE, 64>(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2, only, s);
  else launch_nn<4, MODE, 256>(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2, only, s);
}

// Which search answers a call: the exhaustive scan costs n1 * n2 pair evaluations per sample and direction, the
// grid-pruned one a sort plus a few dozen candidates per query — but three launches and a 1024-thread sort block per
// (sample, cloud, role) whatever the size.  Crossover measured on MI355X (DESIGN.md "Generic Chamfer: search choice").
#ifndef MPA_CHAMFER_GRID_MIN_PAIRS
#define MPA_CHAMFER_GRID_MIN_PAIRS (int64_t)(3000 * 3000)
#endif
bool grid_pays(int64_t batch, int64_t n1, int64_t n2) {
  const int64_t nmin = n1 < n2 ? n1 : n2;
  return nmin >= 512 && n1 * n2 >= MPA_CHAMFER_GRID_MIN_PAIRS && mpa::cloud_grid_supported(batch, n1, n2);
}

// ... and between the two, the matrix-core gated search (gate_nn.hip): one bf16 MFMA per 32 x 32 pairs decides which
// few targets get the pinned arithmetic.  It needs no workspace; below a few hundred points the scan's blocks are too
// short for the gate's set-up (panel staging, two passes) to pay.
#ifndef MPA_CHAMFER_GATE_MIN
#define MPA_CHAMFER_GATE_MIN 192
#endif
bool gate_pays(int64_t n1, int64_t n2) {
  const int64_t nmin = n1 < n2 ? n1 : n2;
  return nmin >= MPA_CHAMFER_GATE_MIN && mpa::gate_supported(n1, n2);
}

int check_forward_args(const void* xyz1, const void* xyz2, int64_t batch, int64_t n1, int64_t n2,
                       const void* dist1, const void* idx1, const void* dist2, const void* idx2) {
  MPA_REQUIRE(batch >= 0 && n1 >= 0 && n2 >= 0, "chamfer_forward: negative size");
  MPA_REQUIRE(n1 < (1 << 30) && n2 < (1 << 30), "chamfer_forward: cloud larger than 2^30 points");
  if (batch == 0 || (n1 == 0 && n2 == 0)) return 1;  // nothing to do
  MPA_REQUIRE((n1 == 0 || (xyz1 && dist1 && idx1)) && (n2 == 0 || (xyz2 && dist2 && idx2)),
              "chamfer_forward: null pointer");
  const int64_t nmax = n1 > n2 ? n1 : n2;
  MPA_REQUIRE(batch * ((nmax + 127) / 128) < (int64_t)1 << 31, "chamfer_forward: grid too large");
  return MPA_OK;
}

template <typename S>
int chamfer_backward_impl(const S* grad_dist1, const S* grad_dist2, const S* xyz1, const S* xyz2,
                          const int64_t* idx1, const int64_t* idx2, int64_t batch, int64_t n1,
                          int64_t n2, S* grad_xyz1, S* grad_xyz2, void* stream) {
  MPA_REQUIRE(batch >= 0 && n1 >= 0 && n2 >= 0, "chamfer_backward: negative size");
  MPA_REQUIRE(n1 < (1 << 30) && n2 < (1 << 30), "chamfer_backward: cloud larger than 2^30 points");
  hipStream_t s = mpa::as_stream(stream);
  const long long total1 = batch * n1, total2 = batch * n2;
  if (total1 == 0 && total2 == 0) return MPA_OK;
  MPA_REQUIRE((total1 == 0 || (grad_dist1 && xyz1 && idx1 && grad_xyz1)) &&
                  (total2 == 0 || (grad_dist2 && xyz2 && idx2 && grad_xyz2)),
              "chamfer_backward: null pointer");
  mpa::zero_words_async(grad_xyz1, (int64_t)(sizeof(S) / 4) * 3 * total1, s);
  mpa::zero_words_async(grad_xyz2, (int64_t)(sizeof(S) / 4) * 3 * total2, s);
  if (total1 == 0 || total2 == 0) return MPA_OK;  // every idx is -1: nothing to scatter
  const long long tmax = total1 > total2 ? total1 : total2;
  long long blocks = (tmax + kThreads - 1) / kThreads;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(chamfer_grad_kernel<S>, dim3((unsigned)blocks, 2, 1), dim3(kThreads), 0, s,
                     grad_dist1, grad_dist2, xyz1, xyz2, (const long long*)idx1,
                     (const long long*)idx2, total1, total2, (int)n1, (int)n2, grad_xyz1,
                     grad_xyz2);
  return mpa::check_launch("chamfer_backward");
}

}  // namespace

// variant: 0 = direct, 1 = fused-form gate, 2 = exact chunk-min, 3 = grid-pruned (needs the workspace), 4 = matrix-core
// gated (gate_nn.hip), -1 = by size — see chamfer_core.h / grid_nn.hip / gate_nn.hip.
extern "C" int mpa_chamfer_forward_variant(const float* xyz1, const float* xyz2, int64_t batch,
                                           int64_t n1, int64_t n2, float* dist1, int64_t* idx1,
                                           float* dist2, int64_t* idx2, int variant, void* workspace,
                                           int64_t workspace_bytes, void* stream) {
  MPA_REQUIRE(variant >= -1 && variant <= 4, "chamfer_forward: unknown variant %d", variant);
  const int st = check_forward_args(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2);
  if (st != MPA_OK) return st < 0 ? st : MPA_OK;
  hipStream_t s = mpa::as_stream(stream);
  const int a = (int)n1, b = (int)n2;
  if (variant == 3)
    MPA_REQUIRE(mpa::cloud_grid_supported(batch, n1, n2), "chamfer_forward: the grid-pruned search needs two non-empty clouds");
  if (variant == 4)
    MPA_REQUIRE(mpa::gate_supported(n1, n2), "chamfer_forward: the matrix-core gated search needs two non-empty clouds of <= 32768 points");
  if (variant == -1) variant = grid_pays(batch, n1, n2) && workspace != nullptr ? 3 : (gate_pays(n1, n2) ? 4 : 2);
  if (variant == 3) {
    MPA_REQUIRE(workspace != nullptr && workspace_bytes >= mpa::cloud_grid_workspace_bytes(batch, n1, n2),
                "chamfer_forward: the grid-pruned search needs mpa_chamfer_workspace() bytes of workspace");
    MPA_REQUIRE((uintptr_t)workspace % 16 == 0, "chamfer_forward: workspace must be 16-byte aligned");
    const int* flagged = nullptr;
    if (int e = mpa::launch_cloud_grid_search(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2, workspace, &flagged, s))
      return e;
    // samples the pruned search handed back (non-finite / huge coordinates): the exhaustive scan, for those only
    launch_nn_sized<mpa::kChunkMin>(xyz1, xyz2, batch, a, b, dist1, idx1, dist2, idx2, s, flagged);
    mpa::launch_cloud_copy_runs(batch, n1, n2, dist1, idx1, dist2, idx2, workspace, s);
  } else if (variant == 4) {
    mpa::launch_gate_cloud_search(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2, s);
  } else if (variant == 0) {
    launch_nn_sized<mpa::kDirect>(xyz1, xyz2, batch, a, b, dist1, idx1, dist2, idx2, s);
  } else if (variant == 1) {
    launch_nn_sized<mpa::kFusedGate>(xyz1, xyz2, batch, a, b, dist1, idx1, dist2, idx2, s);
  } else {
    launch_nn_sized<mpa::kChunkMin>(xyz1, xyz2, batch, a, b, dist1, idx1, dist2, idx2, s);
  }
  return mpa::check_launch("chamfer_forward");
}

extern "C" int mpa_chamfer_workspace(int64_t batch, int64_t n1, int64_t n2, int64_t* bytes) {
  MPA_REQUIRE(bytes != nullptr, "chamfer_workspace: null pointer");
  MPA_REQUIRE(batch >= 0 && n1 >= 0 && n2 >= 0, "chamfer_workspace: negative size");
  // what the DEFAULT call will use: nothing where the size rule sends the call to the exhaustive scan (the per-part
  // call [640, 1000, 3]^2 would otherwise reserve 725 MiB of cell tables it never touches)
  *bytes = grid_pays(batch, n1, n2) ? mpa::cloud_grid_workspace_bytes(batch, n1, n2) : 0;
  return MPA_OK;
}

extern "C" int mpa_chamfer_workspace_variant(int64_t batch, int64_t n1, int64_t n2, int variant, int64_t* bytes) {
  MPA_REQUIRE(bytes != nullptr, "chamfer_workspace_variant: null pointer");
  MPA_REQUIRE(batch >= 0 && n1 >= 0 && n2 >= 0, "chamfer_workspace_variant: negative size");
  MPA_REQUIRE(variant >= -1 && variant <= 4, "chamfer_workspace_variant: unknown variant %d", variant);
  if (variant == -1) return mpa_chamfer_workspace(batch, n1, n2, bytes);
  *bytes = variant == 3 && mpa::cloud_grid_supported(batch, n1, n2) ? mpa::cloud_grid_workspace_bytes(batch, n1, n2) : 0;
  return MPA_OK;
}

extern "C" int mpa_chamfer_forward(const float* xyz1, const float* xyz2, int64_t batch, int64_t n1,
                                   int64_t n2, float* dist1, int64_t* idx1, float* dist2,
                                   int64_t* idx2, void* workspace, int64_t workspace_bytes, void* stream) {
  // without (enough) workspace the exhaustive scan answers every size: same results, n1 * n2 work
  const bool ws_ok = workspace != nullptr && mpa::cloud_grid_supported(batch, n1, n2) &&
                     workspace_bytes >= mpa::cloud_grid_workspace_bytes(batch, n1, n2);
  return mpa_chamfer_forward_variant(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2, -1, ws_ok ? workspace : nullptr,
                                     workspace_bytes, stream);
}

extern "C" int mpa_chamfer_forward_f64(const double* xyz1, const double* xyz2, int64_t batch,
                                       int64_t n1, int64_t n2, double* dist1, int64_t* idx1,
                                       double* dist2, int64_t* idx2, void* stream) {
  const int st = check_forward_args(xyz1, xyz2, batch, n1, n2, dist1, idx1, dist2, idx2);
  if (st != MPA_OK) return st < 0 ? st : MPA_OK;
  const int64_t nmax = n1 > n2 ? n1 : n2;
  const int bpc = (int)((nmax + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(chamfer_nn_kernel_f64, dim3((unsigned)(batch * bpc), 2, 1), dim3(kThreads), 0,
                     mpa::as_stream(stream), xyz1, xyz2, (int)n1, (int)n2, bpc, dist1,
                     (long long*)idx1, dist2, (long long*)idx2);
  return mpa::check_launch("chamfer_forward_f64");
}

extern "C" int mpa_chamfer_backward(const float* grad_dist1, const float* grad_dist2,
                                    const float* xyz1, const float* xyz2, const int64_t* idx1,
                                    const int64_t* idx2, int64_t batch, int64_t n1, int64_t n2,
                                    float* grad_xyz1, float* grad_xyz2, void* stream) {
  return chamfer_backward_impl<float>(grad_dist1, grad_dist2, xyz1, xyz2, idx1, idx2, batch, n1, n2,
                                      grad_xyz1, grad_xyz2, stream);
}

extern "C" int mpa_chamfer_backward_f64(const double* grad_dist1, const double* grad_dist2,
                                        const double* xyz1, const double* xyz2,
                                        const int64_t* idx1, const int64_t* idx2, int64_t batch,
                                        int64_t n1, int64_t n2, double* grad_xyz1,
                                        double* grad_xyz2, void* stream) {
  return chamfer_backward_impl<double>(grad_dist1, grad_dist2, xyz1, xyz2, idx1, idx2, batch, n1,
                                       n2, grad_xyz1, grad_xyz2, stream);
}
