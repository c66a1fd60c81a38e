/*
 * oracle/knn_ref.c — CPU restatement of the reference's kNN graph with the score arithmetic pinned op by op.
 * TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this file.
 *
 * What it restates: `knn` (multi_part_assembly/models/modules/encoder/dgcnn.py:8-15)
 *     inner = -2 * x^T x ; xx = sum(x**2, dim=1) ; pairwise = -xx - inner - xx^T ; idx = pairwise.topk(k)[1]
 * i.e. score(i, j) = (-|x_j|^2 + 2 * dot(x_i, x_j)) - |x_i|^2 with every operation rounded to fp32, and the k best
 * scores per point.  torch.topk leaves the order among EQUAL scores unspecified; here (and in csrc/dg_knn.h) ties are
 * broken by the lower index, and the neighbours are listed best first.
 *
 *   mode 0 (C = 3, the first EdgeConv stage): the arithmetic of the reference's own CPU path, verified bit for bit
 *           against torch (tests/test_oracle_golden.py):  dot = fma(x2,y2, fma(x1,y1, x0*y0))  (the BLAS micro-kernel's
 *           FMA chain), |x|^2 = (x0*x0 + x1*x1) + x2*x2  (torch.sum of separately rounded squares).
 *   mode 1 (C >= 64): the reference's Gram matrix comes out of a blocked BLAS whose summation order is not defined, so
 *           the order is defined by the build's matrix-core kernel (which equals this scalar chain bit for bit):
 *           dot = fmaf chain over k in the order 0, C/2, 1, C/2+1, ..., starting from 0;  |x|^2 = the same chain.
 * Compiled with -ffp-contract=off (oracle/Makefile): the only fused operations are the explicit fmaf() calls.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float dot_mode0(const float* a, const float* b) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }

static float dot_mode1(const float* a, const float* b, int64_t C) {
  float acc = 0.0f;
  const int64_t h = C / 2;
  for (int64_t s = 0; s < h; ++s) {
    acc = fmaf(a[s], b[s], acc);
    acc = fmaf(a[h + s], b[h + s], acc);
  }
  return acc;
}

/* x [n][N][ld] (first C columns used), idx [n][N][k] int32 */
void oracle_knn(const float* x, int64_t n, int64_t N, int64_t C, int64_t ld, int64_t k, int32_t mode, int32_t* idx) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t c = 0; c < n; ++c) {
    const float* xc = x + c * N * ld;
    float* norm = (float*)malloc(sizeof(float) * (size_t)N);
    float* bs = (float*)malloc(sizeof(float) * (size_t)k);
    int32_t* bj = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
    for (int64_t j = 0; j < N; ++j) {
      const float* p = xc + j * ld;
      norm[j] = mode == 0 ? (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2] : dot_mode1(p, p, C);
    }
    for (int64_t i = 0; i < N; ++i) {
      int64_t cnt = 0;
      for (int64_t j = 0; j < N; ++j) {
        const float dot = mode == 0 ? dot_mode0(xc + i * ld, xc + j * ld) : dot_mode1(xc + i * ld, xc + j * ld, C);
        const float s = (-norm[j] + 2.0f * dot) - norm[i];
        /* sorted insertion: score descending; candidates arrive in ascending index, so strict > keeps the lower
         * index in front among equal scores */
        int64_t pos = cnt < k ? cnt : k;
        while (pos > 0 && s > bs[pos - 1]) --pos;
        if (pos < k) {
          const int64_t last = cnt < k ? cnt : k - 1;
          for (int64_t t = last; t > pos; --t) {
            bs[t] = bs[t - 1];
            bj[t] = bj[t - 1];
          }
          bs[pos] = s;
          bj[pos] = (int32_t)j;
          if (cnt < k) ++cnt;
        }
      }
      for (int64_t t = 0; t < k; ++t) idx[(c * N + i) * k + t] = bj[t];
    }
    free(norm);
    free(bs);
    free(bj);
  }
}
