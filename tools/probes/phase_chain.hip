// What does a dependent phase cost — as a kernel of its own in a captured graph, or as a phase of ONE persistent launch
// with a grid-wide ticket barrier and agent-scope (sc1: past the XCD's L2) loads / stores for the data that crosses blocks?
// The phase is shaped like a transformer GEMM tile: a block reads a 32 x 256 fp32 operand panel written by OTHER blocks in the
// previous phase, a 32 x 256 weight panel (constant), does ~1.5 us of arithmetic and writes a 32 x 32 tile.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/phase_chain.hip -o tools/probes/phase_chain.bin && tools/probes/phase_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kM = 640, kN = 256, kK = 256, kT = 512, kTiles = (kM / 32) * (kN / 32);  // 160 tiles per phase
constexpr int kWork = 40;  // dependent FMA rounds of the stand-in arithmetic

__device__ __forceinline__ f4 ld_plain(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ f4 ld_coh(const float* p) {
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_coh(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// one tile: C[r0.., n0..] = f(A[r0.. r0+31][0..255], W[n0.. n0+31][0..255])
template <bool COH>
__device__ __forceinline__ void tile(const float* A, const float* W, float* C, int t, float* lds) {
  const int r0 = (t % (kM / 32)) * 32, n0 = (t / (kM / 32)) * 32;
  f4 a[4], w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + kT * i, row = idx / 64, c4 = idx % 64;
    a[i] = COH ? ld_coh(A + (long long)(r0 + row) * kK + 4 * c4) : ld_plain(A + (long long)(r0 + row) * kK + 4 * c4);
    w[i] = ld_plain(W + (long long)(n0 + row) * kK + 4 * c4);
  }
  if (COH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + kT * i;
    *reinterpret_cast<f4*>(lds + 4 * idx) = a[i];
    *reinterpret_cast<f4*>(lds + 8192 + 4 * idx) = w[i];
  }
  __syncthreads();
  // thread -> outputs (row rl, col) and (row rl + 16, col): a K = 256 dot product each, then the stand-in chain
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  float s0 = 0.0f, s1 = 0.0f;
  for (int k = 0; k < kK; k += 4) {
    const f4 x0 = *reinterpret_cast<const f4*>(lds + rl * kK + k), x1 = *reinterpret_cast<const f4*>(lds + (rl + 16) * kK + k);
    const f4 y = *reinterpret_cast<const f4*>(lds + 8192 + col * kK + ((k + 4 * col) & (kK - 1)));
    s0 += x0.x * y.x + x0.y * y.y + x0.z * y.z + x0.w * y.w;
    s1 += x1.x * y.x + x1.y * y.y + x1.z * y.z + x1.w * y.w;
  }
  for (int i = 0; i < kWork; ++i) {
    s0 = __builtin_fmaf(s0, 0.999f, 1e-3f);
    s1 = __builtin_fmaf(s1, 0.999f, 1e-3f);
  }
  s0 *= 1.0f / 64.0f;
  s1 *= 1.0f / 64.0f;
  if (COH) {
    st_coh(C + (long long)(r0 + rl) * kN + n0 + col, s0);
    st_coh(C + (long long)(r0 + rl + 16) * kN + n0 + col, s1);
  } else {
    C[(long long)(r0 + rl) * kN + n0 + col] = s0;
    C[(long long)(r0 + rl + 16) * kN + n0 + col] = s1;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kT) void phase_kernel(const float* A, const float* W, float* C) {
  __shared__ float lds[16384];
  tile<false>(A, W, C, blockIdx.x, lds);
}

// all blocks resident (grid <= CUs); `bar` counts arrivals monotonically: phase p is complete at (p + 1) * gridDim.x
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's write-through stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

// two-level form: the blocks of an XCD (blockIdx % 8) meet on their own counter, the last of them reports to the global one
__device__ __forceinline__ void grid_barrier2(unsigned* bar, unsigned phase) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x & 7, nx = (gridDim.x + 7 - x) / 8;
    const unsigned old = __hip_atomic_fetch_add(bar + 16 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == (phase + 1) * nx - 1) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (phase + 1) * 8) {}
  }
  __syncthreads();
}

template <bool COH, int BAR>
__global__ __launch_bounds__(kT) void chain_kernel(float* act0, float* act1, const float* W, int phases, unsigned* bar) {
  __shared__ float lds[16384];
  for (int p = 0; p < phases; ++p) {
    const float* A = (p & 1) ? act1 : act0;
    float* C = (p & 1) ? act0 : act1;
    for (int t = blockIdx.x; t < kTiles; t += gridDim.x) tile<COH>(A, W + (long long)(p % 4) * kN * kK, C, t, lds);
    if (BAR == 1) grid_barrier(bar, (unsigned)(p + 1) * gridDim.x);
    if (BAR == 2) grid_barrier2(bar, (unsigned)p);
  }
}

// barrier cost alone
__global__ __launch_bounds__(kT) void barrier_only_kernel(int phases, unsigned* bar) {
  for (int p = 0; p < phases; ++p) grid_barrier(bar, (unsigned)(p + 1) * gridDim.x);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const int phases = 48, reps = 20;
  float *act0, *act1, *W, *ref;
  unsigned* bar;
  CK(hipMalloc(&act0, kM * kK * 4));
  CK(hipMalloc(&act1, kM * kK * 4));
  CK(hipMalloc(&ref, kM * kK * 4));
  CK(hipMalloc(&W, 4 * kN * kK * 4));
  CK(hipMalloc(&bar, 1024));
  std::vector<float> h(kM * kK), hw(4 * kN * kK);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) / 1024.0f - 0.5f;
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 40503u) >> 8 & 1023) / 1024.0f - 0.5f;
  CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // --- one kernel per phase, captured
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int p = 0; p < phases; ++p)
    hipLaunchKernelGGL(phase_kernel, dim3(kTiles), dim3(kT), 0, s, (p & 1) ? act1 : act0, W + (long long)(p % 4) * kN * kK,
                       (p & 1) ? act0 : act1);
  CK(hipStreamEndCapture(s, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CK(hipMemcpyAsync(act0, h.data(), h.size() * 4, hipMemcpyHostToDevice, s));
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(exec, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  printf("graph of %d kernels:        %.2f us per phase\n", phases, best * 1000.0f / phases);
  CK(hipMemcpy(ref, act0, kM * kK * 4, hipMemcpyDeviceToDevice));
  std::vector<float> r0(kM * kK), r1(kM * kK);
  CK(hipMemcpy(r0.data(), ref, r0.size() * 4, hipMemcpyDeviceToHost));
  // --- persistent chain
  for (int variant = 0; variant < 5; ++variant) {
    const int blocks = 160;
    const char* names[5] = {"sc1 data, flat barrier", "sc1 data, two-level barrier", "plain data (WRONG results), flat barrier",
                            "sc1 data, NO barrier (WRONG)", "plain data, NO barrier (WRONG)"};
    best = 1e9f;
    int bad = 0;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemcpyAsync(act0, h.data(), h.size() * 4, hipMemcpyHostToDevice, s));
      CK(hipMemsetAsync(bar, 0, 1024, s));
      CK(hipEventRecord(e0, s));
      if (variant == 0) hipLaunchKernelGGL((chain_kernel<true, 1>), dim3(blocks), dim3(kT), 0, s, act0, act1, W, phases, bar);
      if (variant == 1) hipLaunchKernelGGL((chain_kernel<true, 2>), dim3(blocks), dim3(kT), 0, s, act0, act1, W, phases, bar);
      if (variant == 2) hipLaunchKernelGGL((chain_kernel<false, 1>), dim3(blocks), dim3(kT), 0, s, act0, act1, W, phases, bar);
      if (variant == 3) hipLaunchKernelGGL((chain_kernel<true, 0>), dim3(blocks), dim3(kT), 0, s, act0, act1, W, phases, bar);
      if (variant == 4) hipLaunchKernelGGL((chain_kernel<false, 0>), dim3(blocks), dim3(kT), 0, s, act0, act1, W, phases, bar);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      CK(hipMemcpy(r1.data(), act0, r1.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
    }
    printf("persistent, %3d blocks, %-42s %.2f us per phase   (elements differing from the kernel chain over %d runs: %d)\n",
           blocks, names[variant], best * 1000.0f / phases, reps, bad);
  }
  for (int blocks : {160, 256}) {
    best = 1e9f;
    for (int r = 0; r < reps; ++r) {
      CK(hipMemsetAsync(bar, 0, 1024, s));
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(barrier_only_kernel, dim3(blocks), dim3(kT), 0, s, 1000, bar);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("grid barrier alone, %3d blocks: %.2f us each\n", blocks, best);
  }
  // one phase kernel alone
  best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(phase_kernel, dim3(kTiles), dim3(kT), 0, s, act0, W, act1);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  printf("one phase kernel, events around it: %.2f us\n", best * 1000.0f);
  return 0;
}
