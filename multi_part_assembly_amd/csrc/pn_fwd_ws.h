// PointNet forward layers conv2..conv5 — wave-specialised, fp32-grade split-bf16 products.
// Included by pointnet.hip inside its anonymous namespace, after pn_bwd_q.h (shares its helpers: pn_split4v, PN_MFMA6, ...).
//
//   Y[rows x COUT] = relu(bn_prev(Yprev))[rows x CIN] . W[COUT x CIN]^T      (FIRST: Yprev = conv1's output, recomputed from the points)
//
// One block per CU, persistent over RB-row units of the valid parts, ONE barrier per unit (the structure of pn_bwd_q_kernel):
//   * 4 stager waves: request the unit after next, apply the previous layer's BatchNorm + ReLU, split into three bf16 planes,
//     write the NEXT unit's A panel (double-buffered LDS);
//   * ND GEMM waves: one 32 x 32 output tile each, the wave's 32 weight rows register-resident as split planes (a weight row
//     IS a B fragment: lane = output channel, eight consecutive k), A fragments = 16-byte row reads requested one k-step ahead;
//     hidden layers leave the tile raw in LDS;
//   * hidden layers, 4 store waves: one barrier later they write the previous unit's tile as coalesced 16-byte stores and
//     take BatchNorm's column sums (sum, sum of squares) from it;
//   * last layer (TOP: Y5 is never stored): the GEMM waves keep, per channel, the top-2 records of sign(gamma) * y and the
//     column sums in their accumulator layout (pn_fwd_split_kernel's epilogue).
// Per block: one (sum, sum) row of `partial` per channel (fixed order), reduced by pn_bn_finalize_kernel.

template <int CIN, int COUT, int RB, bool FIRST, int BPC = 1>
__global__ __launch_bounds__(64 * (8 + (RB / 32) * (COUT / 32)), ((8 + (RB / 32) * (COUT / 32)) * BPC + 3) / 4) void pn_fwd_ws_kernel(const float* __restrict__ in, const float* __restrict__ bn_prev,
                                                          const float* __restrict__ w, const int* __restrict__ vlist, int N,
                                                          float* __restrict__ y_out, float* __restrict__ partial,
                                                          const float* __restrict__ wt1) {
  constexpr int NS = 4, SA = 6 * CIN + 16, RT = RB / 32, CT = COUT / 32, ND = RT * CT, KA = CIN / 16;  // (BPC blocks per CU)
  constexpr int NTS = 64 * NS, QC = CIN / 4, RG = NTS / QC, NLY = RB * QC / NTS;  // stagers: float4 per thread and unit
  constexpr int QO = COUT / 4, RGO = NTS / QO, NLO = RB * QO / NTS;               // store waves: float4 per thread and unit
  static_assert(RB * QC % NTS == 0 && RB * QO % NTS == 0 && NTS % QC == 0 && NTS % QO == 0, "shapes");
  static_assert(!FIRST || CIN == 64, "the recomputed input is the 64-channel first layer");
  __shared__ __attribute__((aligned(16))) unsigned char pa[2][RB * SA];  // A planes h | m | l
  __shared__ __attribute__((aligned(16))) float outp[2][RB * COUT];      // a unit's output tiles, row-major
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int TB = (N + RB - 1) / RB, U = vlist[0] * TB, G = gridDim.x;
  const int n_it = (int)blockIdx.x < U ? (U - (int)blockIdx.x + G - 1) / G : 0;
  auto part_of = [&](int it) {
    const int u = (int)blockIdx.x + it * G;
    return u < U && it >= 0 ? vlist[4 + u / TB] : 0;
  };
  auto n0_of = [&](int it) { return (((int)blockIdx.x + it * G) % TB) * RB; };

  if (wave < NS) {
    // ================================================ stager waves ==========================================================
    const int t = threadIdx.x, cy4 = t % QC, ry0 = t / QC;
    const float4 sc = reinterpret_cast<const float4*>(bn_prev)[cy4];
    const float4 sh = reinterpret_cast<const float4*>(bn_prev + CIN)[cy4];
    const pn_f32x2 sc01 = {sc.x, sc.y}, sc23 = {sc.z, sc.w}, sh01 = {sh.x, sh.y}, sh23 = {sh.z, sh.w}, zero2 = {0.0f, 0.0f};
    float4 w1a = {}, w1b = {}, w1c = {};
    if constexpr (FIRST) {
      w1a = reinterpret_cast<const float4*>(wt1)[cy4];
      w1b = reinterpret_cast<const float4*>(wt1 + 64)[cy4];
      w1c = reinterpret_cast<const float4*>(wt1 + 128)[cy4];
    }
    float4 ry[NLY];
    auto fetch = [&](int it, int m) {  // rows past the part's end: any row of the part (zeroed below)
      const int n0 = n0_of(it);
      const long long row0 = (long long)m * N + n0;
#pragma unroll
      for (int i = 0; i < NLY; ++i) {
        const int rl = ry0 + i * RG;
        const int rr = n0 + rl < N ? rl : N - 1 - n0;
        if constexpr (FIRST) {
          const float* p = in + (row0 + rr) * 3;
          ry[i] = make_float4(p[0], p[1], p[2], 0.0f);
        } else {
          ry[i] = reinterpret_cast<const float4*>(in)[(row0 + rr) * QC + cy4];
        }
      }
    };
    auto stash = [&](int it, int b) {
      const int n0 = n0_of(it);
#pragma unroll
      for (int i = 0; i < NLY; ++i) {
        const int rl = ry0 + i * RG;
        float4 yv = ry[i];
        if constexpr (FIRST) {
          const float a0 = yv.x, a1 = yv.y, a2 = yv.z;
          yv = make_float4(first_layer_y(a0, a1, a2, w1a.x, w1b.x, w1c.x), first_layer_y(a0, a1, a2, w1a.y, w1b.y, w1c.y),
                           first_layer_y(a0, a1, a2, w1a.z, w1b.z, w1c.z), first_layer_y(a0, a1, a2, w1a.w, w1b.w, w1c.w));
        }
        pn_f32x2 v01 = __builtin_elementwise_max(__builtin_elementwise_fma(pn_f32x2{yv.x, yv.y}, sc01, sh01), zero2);
        pn_f32x2 v23 = __builtin_elementwise_max(__builtin_elementwise_fma(pn_f32x2{yv.z, yv.w}, sc23, sh23), zero2);
        if (n0 + rl >= N) v01 = v23 = zero2;  // rows past the part's end enter the products as zeros: exact 0 outputs
        pn_bf16x4 ph, pm, pl;
        pn_split4v(v01, v23, ph, pm, pl);
        unsigned char* p = pa[b] + rl * SA + 8 * cy4;
        *reinterpret_cast<pn_bf16x4*>(p) = ph;
        *reinterpret_cast<pn_bf16x4*>(p + 2 * CIN) = pm;
        *reinterpret_cast<pn_bf16x4*>(p + 4 * CIN) = pl;
      }
    };
    int m2 = part_of(2);
    if (n_it > 0) {
      fetch(0, part_of(0));
      stash(0, 0);
      if (n_it > 1) fetch(1, part_of(1));
    }
    __syncthreads();  // panel 0 is complete
    for (int it = 0; it < n_it; ++it) {
      const int m3 = part_of(it + 3);  // (the part id of a unit is looked up two iterations before its rows are requested)
      if (it + 1 < n_it) stash(it + 1, (it + 1) & 1);
      if (it + 2 < n_it) fetch(it + 2, m2);
      m2 = m3;
      __syncthreads();
    }
  } else if (wave < NS + ND) {
    // ================================================= GEMM waves ===========================================================
    const int dw = wave - NS, rt = dw / CT, ct = dw % CT, c0 = 32 * ct;
    pn_bf16x8 bh[KA], bm[KA], bl[KA];  // W[c0 + j][16 ks + 8 hh + u] as h / m / l
    {
      const float* src = w + (long long)(c0 + j) * CIN + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < KA; ++ks) {
        const float4 q0 = *reinterpret_cast<const float4*>(src + 16 * ks), q1 = *reinterpret_cast<const float4*>(src + 16 * ks + 4);
        const float f[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          bh[ks][u] = (__bf16)f[u];
          const float r1 = f[u] - (float)bh[ks][u];
          bm[ks][u] = (__bf16)r1;
          bl[ks][u] = (__bf16)(r1 - (float)bm[ks][u]);
        }
      }
    }
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
      const int b = it & 1;
      const unsigned char* arow = pa[b] + (32 * rt + j) * SA + 16 * hh;
      f32x16 acc = {0};
      pn_bf16x8 fa[2][3];
#pragma unroll
      for (int i = 0; i <= KA; ++i) {
        if (i < KA) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fa[i & 1][pl] = *reinterpret_cast<const pn_bf16x8*>(arow + 2 * CIN * pl + 32 * i);
        }
        if (i > 0) {
          const int sb = (i - 1) & 1;
          PN_MFMA6(acc, fa[sb][0], fa[sb][1], fa[sb][2], bh[i - 1], bm[i - 1], bl[i - 1])
        }
      }
      float* o = &outp[b][0] + (32 * rt) * COUT + c0 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[acc_row(r, hh) * COUT] = acc[r];
      __syncthreads();
    }
  } else {
    // ================================================= store waves ==========================================================
    const int t = threadIdx.x - 64 * (NS + ND), co4 = t % QO, ro0 = t / QO;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f), ss = s;
    auto flush = [&](int it, int m) {  // unit `it`: its tiles sit in outp[it & 1] since the last barrier
      const int n0 = n0_of(it);
      const long long row0 = (long long)m * N + n0;
      const float* ob = &outp[it & 1][0];
#pragma unroll
      for (int i = 0; i < NLO; ++i) {
        const int rl = ro0 + i * RGO;
        const float4 v = *reinterpret_cast<const float4*>(ob + rl * COUT + 4 * co4);
        if (n0 + rl < N) reinterpret_cast<float4*>(y_out)[(row0 + rl) * QO + co4] = v;
        s.x += v.x;  // rows past the part's end are exact zeros: no mask needed for the statistics
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
        ss.x = __builtin_fmaf(v.x, v.x, ss.x);
        ss.y = __builtin_fmaf(v.y, v.y, ss.y);
        ss.z = __builtin_fmaf(v.z, v.z, ss.z);
        ss.w = __builtin_fmaf(v.w, v.w, ss.w);
      }
    };
    int m_prev = 0, m_cur = part_of(0);
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
      const int m_next = part_of(it + 1);
      if (it > 0) flush(it - 1, m_prev);
      m_prev = m_cur;
      m_cur = m_next;
      __syncthreads();
    }
    if (n_it > 0) flush(n_it - 1, m_prev);
    // block totals: the RGO row groups of a channel, through LDS, in a fixed order
    float* scr = reinterpret_cast<float*>(&pa[0][0]);
    *reinterpret_cast<float4*>(scr + (0 * RGO + ro0) * COUT + 4 * co4) = s;
    *reinterpret_cast<float4*>(scr + (1 * RGO + ro0) * COUT + 4 * co4) = ss;
  }
  __syncthreads();  // the end-of-block scratch (aliasing the panels, which nobody reads any more) is complete
  if ((int)threadIdx.x < 2 * COUT) {
    const float* scr = reinterpret_cast<const float*>(&pa[0][0]);
    const int qq = (int)threadIdx.x / COUT, c = (int)threadIdx.x % COUT;
    float v = 0.0f;
#pragma unroll
    for (int g = 0; g < RGO; ++g) v += scr[(qq * RGO + g) * COUT + c];
    partial[((long long)blockIdx.x * COUT + c) * 2 + qq] = v;
  }
}

// The last layer (conv5, CIN = 128 -> F): Y5 is never stored.  NWD = F / 32 GEMM waves, each with its 32 weight rows in
// registers (96), keep — per channel and (part, row split) group — the top-2 records of sign(gamma) * y and the column
// sums in their accumulator layout.  A block walks GROUPS (valid part, split) round-robin and the TG 32-row tiles of a group
// back to back, so a group's running records stay in registers and leave once (the layout pn_top_finalize_kernel and
// pn_bn_finalize_kernel always read: row m * splits + sp).
template <int CIN, int NWD>
__global__ __launch_bounds__(64 * (4 + NWD), (4 + NWD + 3) / 4) void pn_fwd_ws_top_kernel(
    const float* __restrict__ in, const float* __restrict__ bn_prev, const float* __restrict__ w, int cout,
    const int* __restrict__ vlist, int N, int splits, float* __restrict__ partial, float* __restrict__ topv,
    int* __restrict__ topn, const float* __restrict__ gamma_top) {
  constexpr int NS = 4, RB = 32, SA = 6 * CIN + 16, KA = CIN / 16;
  constexpr int NTS = 64 * NS, QC = CIN / 4, RG = NTS / QC, NLY = RB * QC / NTS;
  __shared__ __attribute__((aligned(16))) unsigned char pa[2][RB * SA];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int TB = (N + RB - 1) / RB, TG = (TB + splits - 1) / splits;  // tiles per part, per group
  const int NG = vlist[0] * splits, G = gridDim.x;
  const int n_grp = (int)blockIdx.x < NG ? (NG - (int)blockIdx.x + G - 1) / G : 0, n_it = n_grp * TG;
  auto group_of = [&](int it) { return (int)blockIdx.x + (it / TG) * G; };
  auto part_of = [&](int it) {
    const int g = group_of(it);
    return g < NG && it >= 0 ? vlist[4 + g / splits] : 0;
  };
  auto n0_of = [&](int it) { return ((group_of(it) % splits) * TG + it % TG) * RB; };  // (may lie past the part's end: an empty tile)

  if (wave < NS) {
    const int t = threadIdx.x, cy4 = t % QC, ry0 = t / QC;
    const float4 sc = reinterpret_cast<const float4*>(bn_prev)[cy4];
    const float4 sh = reinterpret_cast<const float4*>(bn_prev + CIN)[cy4];
    const pn_f32x2 sc01 = {sc.x, sc.y}, sc23 = {sc.z, sc.w}, sh01 = {sh.x, sh.y}, sh23 = {sh.z, sh.w}, zero2 = {0.0f, 0.0f};
    float4 ry[NLY];
    auto fetch = [&](int it, int m) {
      const int n0 = n0_of(it);
#pragma unroll
      for (int i = 0; i < NLY; ++i) {
        const int gn = n0 + ry0 + i * RG;
        ry[i] = reinterpret_cast<const float4*>(in)[((long long)m * N + (gn < N ? gn : N - 1)) * QC + cy4];
      }
    };
    auto stash = [&](int it, int b) {
      const int n0 = n0_of(it);
#pragma unroll
      for (int i = 0; i < NLY; ++i) {
        const int rl = ry0 + i * RG;
        pn_f32x2 v01 = __builtin_elementwise_max(__builtin_elementwise_fma(pn_f32x2{ry[i].x, ry[i].y}, sc01, sh01), zero2);
        pn_f32x2 v23 = __builtin_elementwise_max(__builtin_elementwise_fma(pn_f32x2{ry[i].z, ry[i].w}, sc23, sh23), zero2);
        if (n0 + rl >= N) v01 = v23 = zero2;  // rows past the part's end enter the products as zeros
        pn_bf16x4 ph, pm, pl;
        pn_split4v(v01, v23, ph, pm, pl);
        unsigned char* p = pa[b] + rl * SA + 8 * cy4;
        *reinterpret_cast<pn_bf16x4*>(p) = ph;
        *reinterpret_cast<pn_bf16x4*>(p + 2 * CIN) = pm;
        *reinterpret_cast<pn_bf16x4*>(p + 4 * CIN) = pl;
      }
    };
    int m2 = part_of(2);
    if (n_it > 0) {
      fetch(0, part_of(0));
      stash(0, 0);
      if (n_it > 1) fetch(1, part_of(1));
    }
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
      const int m3 = part_of(it + 3);
      if (it + 1 < n_it) stash(it + 1, (it + 1) & 1);
      if (it + 2 < n_it) fetch(it + 2, m2);
      m2 = m3;
      __syncthreads();
    }
  } else {
    const int c0 = 32 * (wave - NS);
    pn_bf16x8 bh[KA], bm[KA], bl[KA];  // W[c0 + j][16 ks + 8 hh + u] as h / m / l
    {
      const float* src = w + (long long)(c0 + j) * CIN + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < KA; ++ks) {
        const float4 q0 = *reinterpret_cast<const float4*>(src + 16 * ks), q1 = *reinterpret_cast<const float4*>(src + 16 * ks + 4);
        const float f[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          bh[ks][u] = (__bf16)f[u];
          const float r1 = f[u] - (float)bh[ks][u];
          bm[ks][u] = (__bf16)r1;
          bl[ks][u] = (__bf16)(r1 - (float)bm[ks][u]);
        }
      }
    }
    const float sgn = gamma_top[c0 + j] < 0.0f ? -1.0f : 1.0f;  // only the extrema of sign(gamma) * y can become the maximum
    float s_ = 0.0f, ss_ = 0.0f;
    Top2 hi = top2_empty();
    int m_cur = part_of(0);
    __syncthreads();
    for (int it = 0; it < n_it; ++it) {
      const int b = it & 1;
      const int m_next = part_of(it + 1);
      const unsigned char* arow = pa[b] + j * SA + 16 * hh;
      f32x16 acc = {0};
      pn_bf16x8 fa[2][3];
#pragma unroll
      for (int i = 0; i <= KA; ++i) {
        if (i < KA) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fa[i & 1][pl] = *reinterpret_cast<const pn_bf16x8*>(arow + 2 * CIN * pl + 32 * i);
        }
        if (i > 0) {
          const int sb = (i - 1) & 1;
          PN_MFMA6(acc, fa[sb][0], fa[sb][1], fa[sb][2], bh[i - 1], bm[i - 1], bl[i - 1])
        }
      }
      const int r0 = n0_of(it);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gn = r0 + acc_row(r, hh);
        top2_push(hi, gn < N ? sgn * acc[r] : -__builtin_inff(), gn);  // rows past the end must not enter the extrema
        s_ += acc[r];  // zero operand rows give exactly 0: no mask needed for the statistics
        ss_ = __builtin_fmaf(acc[r], acc[r], ss_);
      }
      if (it % TG == TG - 1) {  // the group is complete: lanes l and l + 32 hold the same channel (rows 4 hh .. of every 8)
        s_ += __shfl_xor(s_, 32, 64);
        ss_ += __shfl_xor(ss_, 32, 64);
        hi = top2_merge(hi, top2_shfl_xor(hi, 32));
        if (hh == 0) {
          const int ob = m_cur * splits + group_of(it) % splits;
          const long long o = ((long long)ob * cout + c0 + j) * 2;
          partial[o] = s_;
          partial[o + 1] = ss_;
          *reinterpret_cast<float2*>(topv + o) = make_float2(hi.v1, hi.v2);
          *reinterpret_cast<int2*>(topn + o) = make_int2(hi.n1, hi.n2);
        }
        s_ = ss_ = 0.0f;
        hi = top2_empty();
      }
      m_cur = m_next;
      __syncthreads();
    }
  }
}
