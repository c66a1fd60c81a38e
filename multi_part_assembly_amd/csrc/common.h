// Shared host-side helpers for libmpa_hip.so (gfx950 only; see include/mpa_hip.h for the ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mpa_hip.h"

namespace mpa {

// Records `msg` as this thread's last error and returns `code` (so callers can `return fail(...)`).
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Checks hipGetLastError() after a launch; returns MPA_OK or MPA_ELAUNCH (with the error recorded).
int check_launch(const char* what);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Zero-fills `words` 32-bit words with a KERNEL.  hipMemsetAsync is deliberately not used anywhere in the library:
// as a node of a captured HIP graph it misbehaved on replay (ROCm 7.0/7.2: faults on large fills, stale data on
// small ones), and every entry point must stay graph-capturable.
void zero_words_async(void* p, int64_t words, hipStream_t s);
// transformer.hip: dW [N][K] = dY^T X over few rows M (N, K multiples of 32), db = column sums of dY (nullable)
void launch_small_wgrad(const float* dY, const float* X, int ldx, float* dW, float* db, int M, int N, int K, hipStream_t s);

constexpr int kWave = 64;  // gfx950 wavefront width

}  // namespace mpa

#define MPA_REQUIRE(cond, ...)                                \
  do {                                                        \
    if (!(cond)) return ::mpa::fail(MPA_EINVAL, __VA_ARGS__); \
  } while (0)
