// Calibration of rocprofv3's FETCH_SIZE for the access pattern of the search kernels: every lane gathers 16-byte records at
// pseudo-random positions of a table far larger than the caches (1 GiB, each record read at most once), and a streaming
// kernel reads the same number of bytes with coalesced 16-byte loads.  Known bytes: `n` records x 16 B requested; the memory
// system moves whole 32- / 64- / 128-byte sectors, so FETCH_SIZE / (n x 16 B) of the gather is the factor to apply to the
// counter for gathers, and of the stream the factor for wide coalesced reads (MI355X_MICROARCH.md: 1/2 reported).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/gather16.hip -o tools/probes/gather16.bin
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc -o pm -- tools/probes/gather16.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void gather16_kernel(const float4* __restrict__ tab, unsigned long long nrec, unsigned long long n, float* out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // a permutation of [0, nrec): odd multiplier modulo a power of two
  const unsigned long long idx = (i * 0x9E3779B97F4A7C15ull + 12345ull) & (nrec - 1);
  const float4 v = tab[idx];
  if (v.x == 123456.0f) out[0] = v.y;  // (never true: keeps the load)
}
__global__ void stream16_kernel(const float4* __restrict__ tab, unsigned long long n, float* out) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = tab[i];
  if (v.x == 123456.0f) out[0] = v.y;
}
int main() {
  const unsigned long long nrec = 1ull << 26;  // 64 Mi records = 1 GiB
  const unsigned long long n = 1ull << 22;     // 4 Mi gathers = 64 MiB requested, 1 / 16 of the table
  float4* tab;
  float* out;
  if (hipMalloc(&tab, nrec * 16) != hipSuccess || hipMalloc(&out, 16) != hipSuccess) return 1;
  (void)hipMemset(tab, 0, nrec * 16);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(gather16_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, tab, nrec, n, out);
    hipLaunchKernelGGL(stream16_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, 0, tab + (nrec / 2) + rep * n, n, out);
  }
  (void)hipDeviceSynchronize();
  printf("requested bytes per launch: gather %llu, stream %llu\n", n * 16, n * 16);
  return 0;
}
