// Exact brute-force nearest-neighbour scan shared by the generic Chamfer operator (chamfer.hip) and
// the fused assembly-loss kernels (assembly_loss.hip).
//
// One wave = 64 lanes x Q query points held in registers; the target cloud is walked with
// wave-uniform addresses so it arrives through the scalar cache in SGPRs (s_load_dwordx8/x16) and
// feeds the VALU as scalar operands: no LDS, no barriers, no VGPRs for targets.  Queries are packed
// in pairs (ext_vector float2) so the subtract / multiply / fma run as v_pk_*_f32, two distance
// evaluations per instruction.
//
// Arithmetic contract (include/mpa_hip.h): d = (dx*dx + dy*dy) + dz*dz, every op rounded (files are
// built with -ffp-contract=off), lowest target index wins ties (strict `<` in index order), a query
// that never sees d < 1e32 keeps (1e32, -1).
//
// Three scan modes, all bit-identical in their results:
//   kDirect    : compare + 2 selects per pair (the textbook loop).
//   kChunkMin  : exact d for a chunk of 8 targets, v_min3 tree, ONE compare per chunk; the rare
//                chunk whose minimum beats the running best is rescanned for the first index
//                attaining it.  Ties never trigger (strict `<`), so duplicate-heavy clouds stay fast.
//   kFusedGate : as kChunkMin but the chunk minimum is taken over the cheaper fused form
//                f = fma(dz,dz, fma(dy,dy, dx*dx)); chunks with min f <= (1+2^-20) * (running min of f)
//                are rescanned with the exact form.  f and d round the same positive 3-term sum and
//                differ by < 8 ulp, so no candidate that could win under d is ever skipped.  Fastest
//                on tie-free data; exact ties (duplicated points) hit the rescan every chunk.
#pragma once

#include <hip/hip_runtime.h>

namespace mpa {

enum ScanMode { kDirect = 0, kFusedGate = 1, kChunkMin = 2 };

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kScanChunk = 8;  // targets per unrolled chunk (24 floats = s_load_dwordx16 + x8)

__device__ __forceinline__ float dist_exact_f(float dx, float dy, float dz) {
  return (dx * dx + dy * dy) + dz * dz;
}
__device__ __forceinline__ f32x2 dist_exact_v(f32x2 dx, f32x2 dy, f32x2 dz) {
  return (dx * dx + dy * dy) + dz * dz;
}
__device__ __forceinline__ f32x2 dist_fused_v(f32x2 dx, f32x2 dy, f32x2 dz) {
  return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
}
__device__ __forceinline__ float min8(const float (&v)[kScanChunk]) {
  // fminf ignores NaNs — wanted: a NaN candidate can never win a strict `<`
  const float a = __builtin_fminf(__builtin_fminf(v[0], v[1]), v[2]);
  const float b = __builtin_fminf(__builtin_fminf(v[3], v[4]), v[5]);
  return __builtin_fminf(__builtin_fminf(a, b), __builtin_fminf(v[6], v[7]));
}

// Running nearest-neighbour state of Q (even) queries per lane, fp32.
template <int Q, int MODE>
struct NNScan {
  static_assert(Q % 2 == 0, "queries are packed in pairs");
  static constexpr int H = Q / 2;
  f32x2 X[H], Y[H], Z[H];
  float best[Q];
  int bidx[Q];
  float fmin_[Q], gate[Q];  // kFusedGate only

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      best[q] = 1e32f;  // chamfer_kernel.cu:60
      bidx[q] = -1;
      fmin_[q] = gate[q] = __builtin_inff();
    }
  }
  __device__ __forceinline__ void set_query(int q, float x, float y, float z) {
    X[q >> 1][q & 1] = x;
    Y[q >> 1][q & 1] = y;
    Z[q >> 1][q & 1] = z;
  }

  // One candidate, exact form (tails, pad representatives).  `tb` is the cloud base, j the index.
  __device__ __forceinline__ void scan_one(const float* __restrict__ tb, int j, int reported) {
    const float sx = tb[3 * j + 0], sy = tb[3 * j + 1], sz = tb[3 * j + 2];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const f32x2 d = dist_exact_v(X[h] - sx, Y[h] - sy, Z[h] - sz);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = 2 * h + e;
        if (d[e] < best[q]) {
          best[q] = d[e];
          bidx[q] = reported;
        }
        if (MODE == kFusedGate) {  // keep the gate consistent: f of this candidate is within 8 ulp of d
          fmin_[q] = __builtin_fminf(fmin_[q], d[e]);
          gate[q] = fmin_[q] * 1.00000095367431640625f;
        }
      }
    }
  }

  // Targets tb[j_begin .. j_end) (indices relative to `tb`); `index_offset` is added to the reported
  // index (so a sub-range of a larger cloud reports global indices).
  __device__ __forceinline__ void scan_range(const float* __restrict__ tb, int j_begin, int j_end,
                                             int index_offset) {
    constexpr int T = kScanChunk;
    const int n_main = (j_end - j_begin) / T * T;
    // Software prefetch: the scalar loads of chunk c+1 are issued before the VALU work on chunk c, so
    // their (scalar-cache / L2) latency overlaps ~300 VALU cycles instead of stalling the wave.
    float nx[T], ny[T], nz[T];
    if (n_main > 0) {
#pragma unroll
      for (int t = 0; t < T; ++t) {  // wave-uniform addresses -> scalar loads
        nx[t] = tb[3 * (j_begin + t) + 0];
        ny[t] = tb[3 * (j_begin + t) + 1];
        nz[t] = tb[3 * (j_begin + t) + 2];
      }
    }
    for (int c = 0; c < n_main; c += T) {
      const int j0 = j_begin + c;
      float tx[T], ty[T], tz[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        tx[t] = nx[t];
        ty[t] = ny[t];
        tz[t] = nz[t];
      }
      {
        // clamp instead of branching: the last iteration harmlessly re-reads its own chunk
        const int jn = (c + T < n_main) ? j0 + T : j0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          nx[t] = tb[3 * (jn + t) + 0];
          ny[t] = tb[3 * (jn + t) + 1];
          nz[t] = tb[3 * (jn + t) + 2];
        }
      }
      if constexpr (MODE == kDirect) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
          for (int h = 0; h < H; ++h) {
            const f32x2 d = dist_exact_v(X[h] - tx[t], Y[h] - ty[t], Z[h] - tz[t]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int q = 2 * h + e;
              const bool lt = d[e] < best[q];
              best[q] = lt ? d[e] : best[q];
              bidx[q] = lt ? j0 + t + index_offset : bidx[q];
            }
          }
        }
      } else {
#pragma unroll
        for (int h = 0; h < H; ++h) {
          f32x2 v[T];
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const f32x2 dx = X[h] - tx[t], dy = Y[h] - ty[t], dz = Z[h] - tz[t];
            v[t] = (MODE == kFusedGate) ? dist_fused_v(dx, dy, dz) : dist_exact_v(dx, dy, dz);
          }
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int q = 2 * h + e;
            float s[T];
#pragma unroll
            for (int t = 0; t < T; ++t) s[t] = v[t][e];
            const float cmin = min8(s);
            if constexpr (MODE == kChunkMin) {
              if (cmin < best[q]) {  // rare; first index attaining the chunk minimum
                best[q] = cmin;
                int first = T - 1;
#pragma unroll
                for (int t = T - 2; t >= 0; --t) first = (s[t] == cmin) ? t : first;
                bidx[q] = j0 + first + index_offset;
              }
            } else {
              if (cmin <= gate[q]) {  // rare on tie-free data: exact recheck of the near-minimal ones
                const float g = gate[q];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                  if (s[t] <= g) {
                    const float d = dist_exact_f(X[h][e] - tx[t], Y[h][e] - ty[t], Z[h][e] - tz[t]);
                    if (d < best[q]) {
                      best[q] = d;
                      bidx[q] = j0 + t + index_offset;
                    }
                  }
                }
                fmin_[q] = __builtin_fminf(fmin_[q], cmin);
                gate[q] = fmin_[q] * 1.00000095367431640625f;  // 1 + 2^-20
              }
            }
          }
        }
      }
    }
    for (int j = j_begin + n_main; j < j_end; ++j) scan_one(tb, j, j + index_offset);
  }
};

}  // namespace mpa
