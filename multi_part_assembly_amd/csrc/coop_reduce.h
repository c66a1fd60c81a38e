// Cooperative fixed-order column reduction shared by the PointNet and DGCNN kernels (BatchNorm statistics and
// BatchNorm-backward coefficients are sums over per-block partial tables).
#pragma once

#include <hip/hip_runtime.h>

namespace mpa {

// Row loop of the row-tiled streaming kernels: thread (row group g of G) visits rows g, g + G, ... < rows.  `load(i)`
// returns what row i needs from memory, `use(i, t)` consumes it — U rows' loads are issued before the first is used.
// (A rolled `for (i = g; i < rows; i += G)` loop waits for every trip's loads before it issues the next trip's: one row
// in flight per wave, ~2 TB/s for the whole chip however many waves are resident — tools/isa_serial_loops.py lists
// such loops.)  The rows are consumed in the same ascending order either way: sums do not change.
template <int U, typename Load, typename Use>
__device__ __forceinline__ void batched_rows(int g, int G, int rows, Load load, Use use) {
  using T = decltype(load(0));
  int i = g;
  for (; i + (U - 1) * G < rows; i += U * G) {
    T t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = load(i + u * G);
#pragma unroll
    for (int u = 0; u < U; ++u) use(i + u * G, t[u]);
  }
  for (; i < rows; i += G) use(i, load(i));
}


constexpr int kSlices = 16;  // row slices per block of the reduction kernels (64 channels x 16 = 1024 threads)

// Sum per-block (sum0, sum1) partial tables over their rows for 64 channels — cooperatively: a single CU
// streaming the M*splits x 64 x 2 table takes ~25-50 us, so the table is cut into groups of kEB rows, one
// block (64 channels x 16 slices) per group and channel panel (grid = (C/64, G)).  Every block leaves its
// fp64 group sums in `stage`, takes a ticket, and the LAST block of the panel adds the G group sums in fixed
// order: deterministic, one launch.  Returns true in that block only (totals valid for threads < 64).
constexpr int kEB = 64;  // table rows per block

#ifndef MPA_COOP_FENCE
#define MPA_COOP_FENCE 0  // 1: plain stores + agent-scope release / acquire fences (the form of rounds 2-5a; A/B builds)
#endif
__device__ __forceinline__ void coop_store(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double coop_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}

struct CoopWs {
  double* stage;       // [G][C][2]
  unsigned* ticket;    // [C/64], zero between launches
};

// row(e, ok, x, y): the two addends of table row e (ok = false: skip the row)
template <typename Row>
__device__ __forceinline__ bool coop_colsum(int total, int C, int c, const CoopWs ws, Row row, double& s0,
                                            double& s1) {
  __shared__ double sm[kSlices][64][2];
  __shared__ bool last;
  const int cl = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int g = blockIdx.y, G = gridDim.y;
  constexpr int U = kEB / kSlices;
  double x[U], y[U];
  bool ok[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {  // independent loads
    const int e = g * kEB + slice + u * kSlices;
    row(e < total ? e : total - 1, ok[u], x[u], y[u]);
    ok[u] = ok[u] && e < total;
  }
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (ok[u]) {
      a += x[u];
      b += y[u];
    }
  }
  sm[slice][cl][0] = a;
  sm[slice][cl][1] = b;
  __syncthreads();
  if (slice == 0) {
    a = b = 0.0;
#pragma unroll
    for (int k = 0; k < kSlices; ++k) {
      a += sm[k][cl][0];
      b += sm[k][cl][1];
    }
#if MPA_COOP_FENCE
    ws.stage[((long long)g * C + c) * 2] = a;
    ws.stage[((long long)g * C + c) * 2 + 1] = b;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // the group sums are visible device-wide before the ticket is
                                                        // taken (release only: no invalidate needed on this side)
#else
    // the group sums go out as agent-scope atomic stores (written through to the level every XCD sees) and are
    // acknowledged before the ticket is taken.  An agent-scope release FENCE would do the same by writing back the
    // XCD's whole L2 (buffer_wbl2) once per block — measured at ~30 us per launch of 320 blocks (LABBOOK 5.3 xvii)
    coop_store(ws.stage + ((long long)g * C + c) * 2, a);
    coop_store(ws.stage + ((long long)g * C + c) * 2 + 1, b);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  }
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(ws.ticket + blockIdx.x, 1u) == (unsigned)(G - 1);
  __syncthreads();
  if (!last) return false;
#if MPA_COOP_FENCE
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (acquire only)
#endif
  a = b = 0.0;
  batched_rows<4>(slice, kSlices, G,
                  [&](int gg) {
#if MPA_COOP_FENCE
                    return *reinterpret_cast<const double2*>(ws.stage + ((long long)gg * C + c) * 2);
#else
                    const double* p = ws.stage + ((long long)gg * C + c) * 2;  // (agent-scope loads: past this XCD's L2)
                    return make_double2(coop_load(p), coop_load(p + 1));
#endif
                  },
                  [&](int, const double2 t) {
                    a += t.x;
                    b += t.y;
                  });
  sm[slice][cl][0] = a;
  sm[slice][cl][1] = b;
  __syncthreads();
  s0 = s1 = 0.0;
  if (slice == 0) {
#pragma unroll
    for (int k = 0; k < kSlices; ++k) {
      s0 += sm[k][cl][0];
      s1 += sm[k][cl][1];
    }
  }
  if (threadIdx.x == 0) ws.ticket[blockIdx.x] = 0u;  // ready for the next launch
  return true;
}

}  // namespace mpa
