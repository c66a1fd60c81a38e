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

// Sum over the 64 lanes, returned to all of them, on the DPP network: quad permutes, row mirrors, row broadcasts — six
// VALU additions and a lane read.  The `__shfl_xor` butterfly is six LDS permutes (`ds_bpermute`, ~100 cycles each) in a
// dependent chain; in the launch-bound kernels that end in a handful of such sums that chain was a microsecond.  Fixed
// order (so deterministic), but not the butterfly's: lanes are added pairwise inside quads, then mirrored halves, then rows.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto add = [](float a, int bits) { return a + __int_as_float(bits); };
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, true));   // row_mirror: a row's 16 lanes hold its sum
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));  // row_bcast15 into rows 1, 3
  v = add(v, __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));  // row_bcast31 into rows 2, 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

}  // namespace mpa

#define MPA_REQUIRE(cond, ...)                                \
  do {                                                        \
    if (!(cond)) return ::mpa::fail(MPA_EINVAL, __VA_ARGS__); \
  } while (0)
