/*
 * oracle/chamfer_ref.c — CPU restatement of the reference's Chamfer operator.  TEST INFRASTRUCTURE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file; the
 * product path (multi_part_assembly_amd/) never does.
 *
 * What it restates
 *   forward : ChamferForwardKernel, multi_part_assembly/utils/chamfer/cuda/chamfer_kernel.cu:32-95
 *             (per query point: scan the other cloud in index order, `d < min_dist` strict, start
 *             from min_dist = 1e32 / min_idx = -1, :60-61,:81-85), called once per direction by
 *             ChamferForward :116-168.  The arithmetic of `d` follows the brute-force definition the
 *             reference's own test uses as ground truth (utils/chamfer/test_chamfer.py:8-31:
 *             sum((a-b)**2, -1) then min) — three rounded squares added left to right, NO fused
 *             multiply-add; this file must be compiled with -ffp-contract=off (oracle/Makefile).
 *   backward: ChamferBackwardKernel :175-210 (g = 2*grad_dist; +g*(p-q) into grad_xyz1[p],
 *             -g*(p-q) into grad_xyz2[idx]), both directions as in ChamferBackward :262-285; the
 *             reference accumulates with atomics in unspecified order, here the order is the plain
 *             sequential one (direction 1 then direction 2, points in index order).
 *
 * Pinning: tests/test_oracle_golden.py checks these functions against tests/golden/chamfer_*.npz,
 * produced by tests/golden/make_golden.py from the reference's test_chamfer.py definitions.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define DEFINE_CHAMFER(SUFFIX, S)                                                                  \
  /* one direction: queries a[na,3] vs targets b[nb,3] */                                          \
  static void nn_one_##SUFFIX(const S* a, const S* b, int64_t na, int64_t nb, S* dist,             \
                              int64_t* idx) {                                                      \
    for (int64_t i = 0; i < na; ++i) {                                                             \
      const S x1 = a[3 * i], y1 = a[3 * i + 1], z1 = a[3 * i + 2];                                 \
      S best = (S)1e32;                                                                            \
      int64_t arg = -1;                                                                            \
      for (int64_t j = 0; j < nb; ++j) {                                                           \
        const S dx = x1 - b[3 * j], dy = y1 - b[3 * j + 1], dz = z1 - b[3 * j + 2];                \
        const S d = (dx * dx + dy * dy) + dz * dz;                                                 \
        if (d < best) {                                                                            \
          best = d;                                                                                \
          arg = j;                                                                                 \
        }                                                                                          \
      }                                                                                            \
      dist[i] = best;                                                                              \
      idx[i] = arg;                                                                                \
    }                                                                                              \
  }                                                                                                \
                                                                                                   \
  void oracle_chamfer_forward_##SUFFIX(const S* xyz1, const S* xyz2, int64_t batch, int64_t n1,    \
                                       int64_t n2, S* dist1, int64_t* idx1, S* dist2,              \
                                       int64_t* idx2) {                                            \
    _Pragma("omp parallel for schedule(dynamic, 1)") for (int64_t t = 0; t < 2 * batch; ++t) {     \
      const int64_t b = t >> 1;                                                                    \
      if ((t & 1) == 0)                                                                            \
        nn_one_##SUFFIX(xyz1 + 3 * b * n1, xyz2 + 3 * b * n2, n1, n2, dist1 + b * n1,              \
                        idx1 + b * n1);                                                            \
      else                                                                                         \
        nn_one_##SUFFIX(xyz2 + 3 * b * n2, xyz1 + 3 * b * n1, n2, n1, dist2 + b * n2,              \
                        idx2 + b * n2);                                                            \
    }                                                                                              \
  }                                                                                                \
                                                                                                   \
  static void grad_one_##SUFFIX(const S* g, const int64_t* idx, const S* a, const S* b,            \
                                int64_t na, int64_t nb, S* ga, S* gb) {                            \
    for (int64_t i = 0; i < na; ++i) {                                                             \
      const int64_t j = idx[i];                                                                    \
      if (j < 0 || j >= nb) continue;                                                              \
      const S s = g[i] * (S)2;                                                                     \
      for (int c = 0; c < 3; ++c) {                                                                \
        const S v = s * (a[3 * i + c] - b[3 * j + c]);                                             \
        ga[3 * i + c] += v;                                                                        \
        gb[3 * j + c] -= v;                                                                        \
      }                                                                                            \
    }                                                                                              \
  }                                                                                                \
                                                                                                   \
  void oracle_chamfer_backward_##SUFFIX(const S* g1, const S* g2, const S* xyz1, const S* xyz2,    \
                                        const int64_t* idx1, const int64_t* idx2, int64_t batch,   \
                                        int64_t n1, int64_t n2, S* gxyz1, S* gxyz2) {              \
    memset(gxyz1, 0, sizeof(S) * 3 * batch * n1);                                                  \
    memset(gxyz2, 0, sizeof(S) * 3 * batch * n2);                                                  \
    _Pragma("omp parallel for schedule(static)") for (int64_t b = 0; b < batch; ++b) {             \
      const S *p1 = xyz1 + 3 * b * n1, *p2 = xyz2 + 3 * b * n2;                                    \
      S *o1 = gxyz1 + 3 * b * n1, *o2 = gxyz2 + 3 * b * n2;                                        \
      grad_one_##SUFFIX(g1 + b * n1, idx1 + b * n1, p1, p2, n1, n2, o1, o2);                       \
      grad_one_##SUFFIX(g2 + b * n2, idx2 + b * n2, p2, p1, n2, n1, o2, o1);                       \
    }                                                                                              \
  }

DEFINE_CHAMFER(f32, float)
DEFINE_CHAMFER(f64, double)
