// Error plumbing of the C ABI (include/mpa_hip.h): thread-local last-error string.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>

namespace mpa {

static thread_local char g_last_error[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return MPA_OK;
  return fail(MPA_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
}

namespace {
__global__ void zero_words_kernel(uint32_t* __restrict__ p, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    p[i] = 0u;
}
}  // namespace

void zero_words_async(void* p, int64_t words, hipStream_t s) {
  if (words <= 0) return;
  const int64_t blocks = (words + 255) / 256;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s,
                     static_cast<uint32_t*>(p), (long long)words);
}

}  // namespace mpa

extern "C" int mpa_abi_version(void) { return MPA_ABI_VERSION; }

extern "C" const char* mpa_last_error(void) { return mpa::g_last_error; }
