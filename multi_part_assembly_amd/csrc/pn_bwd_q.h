// PointNet backward of the hidden layers (conv2..conv4) — "Q form", fp32-grade split-bf16 products, wave-specialised.
// Included by pointnet.hip inside its anonymous namespace (uses acc_row, first_layer_y, pn_bf16x8, pn_bf16x4, f32x16).
//
// BatchNorm backward of layer l is the per-channel affine map  dY = alpha*dZ + gammap*Y + betap  and Y = A W^T with
// A = relu(bn_prev(Yprev)) the layer's input, so both gradients can be written WITHOUT the layer's own output Y:
//     dA      = dZ (alpha.W) + A Q + c0,          Q = W^T diag(gammap) W  [CIN x CIN],  c0 = betap W
//     dW      = alpha.(dZ^T A) + gammap.(W G) + betap (x) asum,      G = A^T A,  asum = column sums of A
// (the form the never-stored last layer always used: pn_dgrad_split_kernel).  The kernel therefore reads dZ_l and
// Yprev ONCE and writes dZ_{l-1}: 360 / 270 MB per launch for the 64 -> 128 / 64 -> 64 layers instead of the 540 / 360 MB
// of pn_bwd_fused_kernel (which also read Y_l), and every product runs on v_mfma_f32_32x32x16_bf16 with both operands
// split into three bf16 terms (six products of order <= 2; csrc/dg_gemm_split.h has the error analysis) instead of the
// 2.67 x slower exact-fp32 MFMA.  T = dZ^T A, G and asum leave the kernel as per-block partial tables; the final
// combination with the coefficients is a tiny kernel at the end of the pass (pn_bwd_finish_kernel).
//
// FIRST (conv2; Yprev = conv1's output, never stored): Yprev is recomputed from the 12-byte points as everywhere else,
// and dZ_1 is not written either: the only consumers of dZ_1 are conv1's weight gradient and bn1's coefficients, and
//     dW1 = alpha1.(dZ1^T P) + gammap1.(W1 P^T P) + betap1 (x) psum
// needs dZ1 only through S = dZ1^T P [64 x 3], which the input-gradient epilogue accumulates in registers (3 FMAs per
// element) — 90 MB less written, 90 MB less read and one kernel (pn_wgrad_mfma_kernel<WG_FIRST>, 47 us) less per step.
//
// One block per CU, persistent over RB-row units of the valid parts, three kinds of waves that meet at ONE barrier per unit:
//   * NS stager waves: fetch the unit after next (coalesced 16-byte loads, a whole unit in flight per CU), apply the
//     previous layer's BatchNorm + ReLU, split into three bf16 planes and write the NEXT unit's panels (double-buffered LDS);
//   * ND input-gradient waves: one 32 x 32 output tile each; A fragments = 16-byte row reads of the dZ / A panels, B
//     fragments = alpha.W^T from an LDS image built once per block, Q register-resident; epilogue = ReLU mask, dZ_{l-1}
//     store, BatchNorm-backward sums;
//   * NW weight-gradient waves: the tiles of T and of the upper triangle of G accumulate across all units of the block;
//     their reduction index is the point row, so both operands are TRANSPOSED reads of the row-major panels —
//     ds_read_b64_tr_b16, two per fragment and plane (tools/probes/tr_read.hip pins the lane layout).
// So staging (VALU + memory), the matrix pipe and the epilogues of a CU overlap by construction instead of taking turns
// within every wave (the stash -> barrier -> MFMA -> epilogue cycle that bound pn_bwd_fused_kernel: LABBOOK round 4).

// -DPN_TIMING: every wave of block 5 adds up the cycles it spends between barriers (busy) and prints them at the end
#ifdef PN_TIMING
#define PN_T_DECL unsigned long long t_busy = 0, t_mark = __builtin_amdgcn_s_memtime(), t_start = t_mark;
#define PN_BAR                                         \
  {                                                    \
    t_busy += __builtin_amdgcn_s_memtime() - t_mark;   \
    __syncthreads();                                   \
    t_mark = __builtin_amdgcn_s_memtime();             \
  }
#define PN_T_REPORT(ROLE)                                                                                              \
  if (blockIdx.x == 5 && lane == 0)                                                                                    \
    printf("K=%d FIRST=%d %s wave %d: units %d busy %llu of %llu cycles\n", K, (int)FIRST, ROLE, wave, n_it, t_busy, \
           __builtin_amdgcn_s_memtime() - t_start);
#else
#define PN_T_DECL
#define PN_BAR __syncthreads();
#define PN_T_REPORT(ROLE)
#endif
#define PN_LDS __attribute__((address_space(3)))
typedef short pn_s16x4 __attribute__((ext_vector_type(4)));
typedef short pn_s16x8 __attribute__((ext_vector_type(8)));

// transposed MFMA fragment: this lane's address = row (8 * lane-half + (i >> 2)), column (16 * (group & 1) + 4 * (i & 3))
// of the 16 k-rows x 32 columns the fragment covers (i = lane & 15, group = lane >> 4); the second read 4 rows further.
__device__ __forceinline__ pn_bf16x8 pn_tr_frag(const unsigned char* p, int stride) {
  const pn_s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((PN_LDS pn_s16x4*)p);
  const pn_s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((PN_LDS pn_s16x4*)(p + 4 * stride));
  const pn_s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(pn_bf16x8, v);
}

__device__ __forceinline__ void pn_split4(const float v[4], pn_bf16x4& ph, pn_bf16x4& pm, pn_bf16x4& pl) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    ph[u] = (__bf16)v[u];
    const float r1 = v[u] - (float)ph[u];
    pm[u] = (__bf16)r1;
    pl[u] = (__bf16)(r1 - (float)pm[u]);
  }
}

// the same split on register pairs: v_cvt_pk_bf16_f32 yields the stored pair directly, the residuals are one v_pk_add_f32 per pair —
// 4.5 VALU instructions per element instead of ~8 (the staging waves' instruction issue is what bounds these kernels)
typedef float pn_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pn_split2v(pn_f32x2 x, pn_bf16x2& h, pn_bf16x2& m, pn_bf16x2& l) {
  h = __builtin_convertvector(x, pn_bf16x2);
  const pn_f32x2 r = x - __builtin_convertvector(h, pn_f32x2);
  m = __builtin_convertvector(r, pn_bf16x2);
  const pn_f32x2 t = r - __builtin_convertvector(m, pn_f32x2);
  l = __builtin_convertvector(t, pn_bf16x2);
}
__device__ __forceinline__ void pn_split4v(pn_f32x2 a, pn_f32x2 b, pn_bf16x4& ph, pn_bf16x4& pm, pn_bf16x4& pl) {
  pn_bf16x2 h0, m0, l0, h1, m1, l1;
  pn_split2v(a, h0, m0, l0);
  pn_split2v(b, h1, m1, l1);
  ph = pn_bf16x4{h0[0], h0[1], h1[0], h1[1]};
  pm = pn_bf16x4{m0[0], m0[1], m1[0], m1[1]};
  pl = pn_bf16x4{l0[0], l0[1], l1[0], l1[1]};
}

__device__ __forceinline__ void pn_fmac(float& acc, float a, float b) {
  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
}

#define PN_MFMA6(ACC, AH, AM, AL, BH, BM, BL)                          \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BH, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BM, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BH, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BM, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BL, ACC, 0, 0, 0); \
  ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, BH, ACC, 0, 0, 0);

// Q[k][d] = sum_c gammap_c W[c][k] W[c][d],  c0[d] = sum_c betap_c W[c][d]  (row CIN of q).  grid = CIN + 1, block = CIN.
__global__ void pn_bwd_q_prep_kernel(const float* __restrict__ w, const float* __restrict__ coef, int K, int CIN,
                                     float* __restrict__ q) {
  const int k = blockIdx.x, d = threadIdx.x;
  float acc = 0.0f;
  if (k < CIN) {
#pragma unroll 8
    for (int c = 0; c < K; ++c) acc = __builtin_fmaf(coef[K + c] * w[(long long)c * CIN + k], w[(long long)c * CIN + d], acc);
  } else {
#pragma unroll 8
    for (int c = 0; c < K; ++c) acc = __builtin_fmaf(coef[2 * K + c], w[(long long)c * CIN + d], acc);
  }
  q[(long long)k * CIN + d] = acc;
}

// floats of one block's partial table: T [K][CIN] | G [CIN][CIN] (upper 32 x 32 tiles only) | asum [CIN] | FIRST: S [64][3], P^T P + psum [12]
__host__ __device__ constexpr int pn_bwd_q_elems(int K, int CIN, bool first) {
  return K * CIN + CIN * CIN + CIN + (first ? 192 + 12 : 0);
}

template <int N, typename F>
__device__ __forceinline__ void pn_static_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
  if constexpr (N > 0) {
    pn_static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Output tiles of the weight-gradient waves.  Operand codes: 0..3 = column tile of dZ, 8 / 9 = column tile 0 / 1 of A;
// kind 0 = T (row tile = dZ column tile, column tile = A column tile), kind 1 = G's upper triangle.
struct PnWTile {
  signed char a, b, kind, orow, ocol;
};
template <int K>
struct PnWPlan;
template <>
struct PnWPlan<64> {  // T 2 x 2 + G (0,0) (0,1) (1,1)
  static constexpr int TPW = 2;
  static constexpr PnWTile t[4][2] = {{{0, 8, 0, 0, 0}, {8, 8, 1, 0, 0}},
                                      {{0, 9, 0, 0, 1}, {8, 9, 1, 0, 1}},
                                      {{1, 8, 0, 1, 0}, {-1, -1, 0, 0, 0}},
                                      {{1, 9, 0, 1, 1}, {9, 9, 1, 1, 1}}};
  static constexpr int zc[4] = {0, 0, 1, 1};
  static constexpr bool uses(int w, int code) {
    for (int i = 0; i < TPW; ++i)
      if (t[w][i].a >= 0 && (t[w][i].a == code || t[w][i].b == code)) return true;
    return false;
  }
};
template <>
struct PnWPlan<128> {  // T 4 x 2 + G
  static constexpr int TPW = 3;
  static constexpr PnWTile t[4][3] = {{{0, 8, 0, 0, 0}, {0, 9, 0, 0, 1}, {8, 8, 1, 0, 0}},
                                      {{1, 8, 0, 1, 0}, {1, 9, 0, 1, 1}, {8, 9, 1, 0, 1}},
                                      {{2, 8, 0, 2, 0}, {2, 9, 0, 2, 1}, {9, 9, 1, 1, 1}},
                                      {{3, 8, 0, 3, 0}, {3, 9, 0, 3, 1}, {-1, -1, 0, 0, 0}}};
  static constexpr int zc[4] = {0, 1, 2, 3};
  static constexpr bool uses(int w, int code) {
    for (int i = 0; i < TPW; ++i)
      if (t[w][i].a >= 0 && (t[w][i].a == code || t[w][i].b == code)) return true;
    return false;
  }
};

template <int K, int CIN, int RB, int NS, int ND, int NW, bool FIRST, int KPN = 1>
__global__ __launch_bounds__(64 * (NS + ND + NW), (NS + ND + NW + 3) / 4) void pn_bwd_q_kernel(  // (waves per SIMD)
    const float* __restrict__ dz, const float* __restrict__ y_prev, const float* __restrict__ bn_prev,
    const float* __restrict__ w, const float* __restrict__ coef, const float* __restrict__ q,
    const int* __restrict__ vlist, int N, float* __restrict__ dz_prev, float* __restrict__ partial,
    float* __restrict__ dwpart, const float* __restrict__ wt1) {
  constexpr int SZ = 6 * K + 16, SA = 6 * CIN + 16;      // panel row strides in bytes: odd multiples of 16 (conflict-free b128 row reads)
  constexpr int RT = RB / 32, CT = CIN / 32;
  constexpr int KZ = K / 16, KA = CIN / 16, KW = RB / 16;  // k-steps: dZ . (alpha W), A . Q, weight gradient
  constexpr int NTS = 64 * NS, QK = K / 4, QC = CIN / 4, RG = NTS / QC;
  constexpr int NLZ = RB * QK / NTS, NLY = RB * QC / NTS;  // float4 per stager thread and unit
  constexpr int ELEMS = pn_bwd_q_elems(K, CIN, FIRST);
  constexpr int OUTP = RB * CIN;                           // floats of one partial output tile set (all tiles of a unit)
  static_assert(ND == RT * CT * KPN && RB * QK % NTS == 0 && RB * QC % NTS == 0 && NTS % QK == 0 && NTS % QC == 0, "shapes");
  static_assert(!FIRST || CIN == 64, "the recomputed input is the 64-channel first layer");
  __shared__ __attribute__((aligned(16))) unsigned char pz[2][RB * SZ];  // dZ planes h | m | l
  __shared__ __attribute__((aligned(16))) unsigned char pa[2][RB * SA];  // A planes
  __shared__ __attribute__((aligned(16))) unsigned char wl[CIN * SZ];    // (alpha W)^T planes: row d, column k
  __shared__ __attribute__((aligned(16))) float outp[2][KPN][OUTP];      // dA tiles of a unit (row-major [RB][CIN]), per k-split share
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int TB = (N + RB - 1) / RB, U = vlist[0] * TB, G = gridDim.x;
  const int n_it = (int)blockIdx.x < U ? (U - (int)blockIdx.x + G - 1) / G : 0;
  auto part_of = [&](int it) {
    const int u = (int)blockIdx.x + it * G;
    return u < U && it >= 0 ? vlist[4 + u / TB] : 0;
  };
  auto n0_of = [&](int it) { return (((int)blockIdx.x + it * G) % TB) * RB; };

  // ---- the block's (alpha W)^T image: every thread converts a few (d, 4 k) groups -------------------------------------------
  for (int e = threadIdx.x; e < CIN * (K / 4); e += blockDim.x) {
    const int d = e % CIN, k4 = e / CIN;  // consecutive threads: consecutive d (coalesced reads of W's rows)
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = coef[4 * k4 + u] * w[(long long)(4 * k4 + u) * CIN + d];
    pn_bf16x4 ph, pm, pl;
    pn_split4(v, ph, pm, pl);
    unsigned char* p = wl + d * SZ + 8 * k4;
    *reinterpret_cast<pn_bf16x4*>(p) = ph;
    *reinterpret_cast<pn_bf16x4*>(p + 2 * K) = pm;
    *reinterpret_cast<pn_bf16x4*>(p + 4 * K) = pl;
  }

  // ---- the input gradient's epilogue, shared by the two kinds of matrix waves (each is half idle between its LDS round trips) ----
  // The input-gradient waves leave a unit's dA tiles raw in LDS; one barrier later, at the top of the next iteration, the
  // input-gradient waves finish the first half of the unit's rows and the weight-gradient waves the second: ReLU mask from the
  // re-fetched Yprev rows (requested an iteration earlier), dZ_{l-1} as coalesced 16-byte stores, BatchNorm-backward sums.
  // Thread -> (row group, 4 columns) as in the stagers.
  static_assert(NW == NS && ND == NS && NLY % 2 == 0, "the epilogue uses the stagers' thread -> (row, 4 columns) map in both roles");
  constexpr int EH = NLY / 2;  // float4 rows per thread, role and unit
    const int te = threadIdx.x & (NTS - 1);
    const int cy4 = te % QC, ry0 = te / QC;
    const float4 sc = reinterpret_cast<const float4*>(bn_prev)[cy4];
    const float4 sh = reinterpret_cast<const float4*>(bn_prev + CIN)[cy4];
    const float4 mn = reinterpret_cast<const float4*>(bn_prev + 2 * CIN)[cy4];
    float4 w1a = {}, w1b = {}, w1c = {};
    if constexpr (FIRST) {
      w1a = reinterpret_cast<const float4*>(wt1)[cy4];
      w1b = reinterpret_cast<const float4*>(wt1 + 64)[cy4];
      w1c = reinterpret_cast<const float4*>(wt1 + 128)[cy4];
    }
    auto first4 = [&](float4 p) {
      return make_float4(first_layer_y(p.x, p.y, p.z, w1a.x, w1b.x, w1c.x), first_layer_y(p.x, p.y, p.z, w1a.y, w1b.y, w1c.y),
                         first_layer_y(p.x, p.y, p.z, w1a.z, w1b.z, w1c.z), first_layer_y(p.x, p.y, p.z, w1a.w, w1b.w, w1c.w));
    };
    float4 ye[EH];
    float4 s1v = make_float4(0.0f, 0.0f, 0.0f, 0.0f), t2v = s1v;
    float4 sax = s1v, say = s1v, saz = s1v;  // FIRST: S[c][k] = sum dZ1[., c] p_k of this thread's 4 channels
    auto load_y = [&](int it, int m, auto half_tag) {  // rows past the part's end: any row of the part (dropped below)
      constexpr int HALF = decltype(half_tag)::value;
      const int n0 = n0_of(it);
      const long long row0 = (long long)m * N + n0;
#pragma unroll
      for (int i = 0; i < EH; ++i) {
        const int rl = ry0 + (HALF * EH + i) * RG;
        const int rr = n0 + rl < N ? rl : N - 1 - n0;
        if constexpr (FIRST) {
          const float* p = y_prev + (row0 + rr) * 3;
          ye[i] = make_float4(p[0], p[1], p[2], 0.0f);
        } else {
          ye[i] = reinterpret_cast<const float4*>(y_prev)[(row0 + rr) * QC + cy4];
        }
      }
    };
    // the epilogue of unit `it` (its dA tiles in outp[it & 1], its Yprev rows / points in `ye`)
    auto epilogue = [&](int it, int m, auto half_tag) {
      constexpr int HALF = decltype(half_tag)::value;
      const int n0 = n0_of(it);
      const long long row0 = (long long)m * N + n0;
      const float* ob = &outp[it & 1][0][0];
#pragma unroll
      for (int i = 0; i < EH; ++i) {
        const int rl = ry0 + (HALF * EH + i) * RG;
        const bool ok = n0 + rl < N;
        float4 o = *reinterpret_cast<const float4*>(ob + rl * CIN + 4 * cy4);
        if constexpr (KPN == 2) {
          const float4 o2 = *reinterpret_cast<const float4*>(ob + OUTP + rl * CIN + 4 * cy4);
          o.x += o2.x;
          o.y += o2.y;
          o.z += o2.z;
          o.w += o2.w;
        }
        float4 yv = ye[i];
        float px = 0.0f, py = 0.0f, pzc = 0.0f;
        if constexpr (FIRST) {
          px = yv.x;
          py = yv.y;
          pzc = yv.z;
          yv = first4(yv);
        }
        float4 d;  // the mask is the forward's own expression: relu(fma(y, scale, shift)) > 0
        d.x = (ok && __builtin_fmaf(yv.x, sc.x, sh.x) > 0.0f) ? o.x : 0.0f;
        d.y = (ok && __builtin_fmaf(yv.y, sc.y, sh.y) > 0.0f) ? o.y : 0.0f;
        d.z = (ok && __builtin_fmaf(yv.z, sc.z, sh.z) > 0.0f) ? o.z : 0.0f;
        d.w = (ok && __builtin_fmaf(yv.w, sc.w, sh.w) > 0.0f) ? o.w : 0.0f;
        if constexpr (FIRST) {
          sax.x = __builtin_fmaf(d.x, px, sax.x);
          sax.y = __builtin_fmaf(d.y, px, sax.y);
          sax.z = __builtin_fmaf(d.z, px, sax.z);
          sax.w = __builtin_fmaf(d.w, px, sax.w);
          say.x = __builtin_fmaf(d.x, py, say.x);
          say.y = __builtin_fmaf(d.y, py, say.y);
          say.z = __builtin_fmaf(d.z, py, say.z);
          say.w = __builtin_fmaf(d.w, py, say.w);
          saz.x = __builtin_fmaf(d.x, pzc, saz.x);
          saz.y = __builtin_fmaf(d.y, pzc, saz.y);
          saz.z = __builtin_fmaf(d.z, pzc, saz.z);
          saz.w = __builtin_fmaf(d.w, pzc, saz.w);
        } else {
          if (ok) reinterpret_cast<float4*>(dz_prev)[(row0 + rl) * QC + cy4] = d;
        }
        s1v.x += d.x;
        s1v.y += d.y;
        s1v.z += d.z;
        s1v.w += d.w;
        t2v.x = __builtin_fmaf(d.x, yv.x - mn.x, t2v.x);  // sum d (yprev - mean): scaled by invstd at the end
        t2v.y = __builtin_fmaf(d.y, yv.y - mn.y, t2v.y);
        t2v.z = __builtin_fmaf(d.z, yv.z - mn.z, t2v.z);
        t2v.w = __builtin_fmaf(d.w, yv.w - mn.w, t2v.w);
      }
    };
    auto put_sums = [&](int role) {  // scratch rows [(q * 2 + role) * RG + row group]: q = 1 s1, 2 s2, 3..5 S's columns
      float* scr = reinterpret_cast<float*>(&pz[0][0]);
      const float* isd = bn_prev + 3 * CIN;
      asm volatile("" : "+s"(isd));  // rebuilt here from the scalar base: not a lane address kept alive across the unit loop
      const float4 is4 = reinterpret_cast<const float4*>(isd)[cy4];
      t2v.x *= is4.x;
      t2v.y *= is4.y;
      t2v.z *= is4.z;
      t2v.w *= is4.w;
      *reinterpret_cast<float4*>(scr + ((1 * 2 + role) * RG + ry0) * CIN + 4 * cy4) = s1v;
      *reinterpret_cast<float4*>(scr + ((2 * 2 + role) * RG + ry0) * CIN + 4 * cy4) = t2v;
      if constexpr (FIRST) {
        *reinterpret_cast<float4*>(scr + ((3 * 2 + role) * RG + ry0) * CIN + 4 * cy4) = sax;
        *reinterpret_cast<float4*>(scr + ((4 * 2 + role) * RG + ry0) * CIN + 4 * cy4) = say;
        *reinterpret_cast<float4*>(scr + ((5 * 2 + role) * RG + ry0) * CIN + 4 * cy4) = saz;
      }
    };
    using Half0 = std::integral_constant<int, 0>;
    using Half1 = std::integral_constant<int, 1>;

  if (wave < NS) {
    // ================================================ stager waves ==========================================================
    // per iteration `it`: conversion of unit it + 1 into the other panel (its rows were requested a whole iteration ago), then
    // the requests for unit it + 2.
    const int t = threadIdx.x;
    const int cz4 = t % QK, rz0 = t / QK;
    float4 rz[NLZ], ry[NLY];
    float4 colsum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float pp[9] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};  // FIRST: xx xy xz yy yz zz, x y z
    auto load_rows = [&](int it, int m, float4 (&dst)[NLY]) {  // rows past the part's end: any row of the part (dropped by the users)
      const int n0 = n0_of(it);
      const long long row0 = (long long)m * N + n0;
#pragma unroll
      for (int i = 0; i < NLY; ++i) {
        const int rl = ry0 + i * RG;
        const int rr = n0 + rl < N ? rl : N - 1 - n0;
        if constexpr (FIRST) {
          const float* p = y_prev + (row0 + rr) * 3;
          dst[i] = make_float4(p[0], p[1], p[2], 0.0f);
        } else {
          dst[i] = reinterpret_cast<const float4*>(y_prev)[(row0 + rr) * QC + cy4];
        }
      }
    };
    auto fetch = [&](int it, int m) {
      const int n0 = n0_of(it);
      const long long row0 = (long long)m * N + n0;
#pragma unroll
      for (int i = 0; i < NLZ; ++i) {
        const int rl = rz0 + i * (NTS / QK);
        const int rr = n0 + rl < N ? rl : N - 1 - n0;
        rz[i] = reinterpret_cast<const float4*>(dz)[(row0 + rr) * QK + cz4];
      }
      load_rows(it, m, ry);
    };
    const pn_f32x2 sc01 = {sc.x, sc.y}, sc23 = {sc.z, sc.w}, sh01 = {sh.x, sh.y}, sh23 = {sh.z, sh.w}, zero2 = {0.0f, 0.0f};
    auto stash_t = [&](int it, int b, auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;  // every row of the unit lies inside the part (all but a part's last unit)
      const int n0 = n0_of(it);
#pragma unroll
      for (int i = 0; i < NLZ; ++i) {
        const int rl = rz0 + i * (NTS / QK);
        const bool ok = FULL || n0 + rl < N;
        const pn_f32x2 v01 = {ok ? rz[i].x : 0.0f, ok ? rz[i].y : 0.0f}, v23 = {ok ? rz[i].z : 0.0f, ok ? rz[i].w : 0.0f};
        pn_bf16x4 ph, pm, pl;
        pn_split4v(v01, v23, ph, pm, pl);
        unsigned char* p = pz[b] + rl * SZ + 8 * cz4;
        *reinterpret_cast<pn_bf16x4*>(p) = ph;
        *reinterpret_cast<pn_bf16x4*>(p + 2 * K) = pm;
        *reinterpret_cast<pn_bf16x4*>(p + 4 * K) = pl;
      }
#pragma unroll
      for (int i = 0; i < NLY; ++i) {
        const int rl = ry0 + i * RG;
        const bool ok = FULL || n0 + rl < N;
        float4 yv = ry[i];
        if constexpr (FIRST) {
          const float a0 = yv.x, a1 = yv.y, a2 = yv.z;
          if (cy4 == 0 && ok) {
            // explicit scalar instructions: the SLP-packed form the compiler makes of these nine updates (v_pk_fma_f32 with
            // op_sel operands) gave a run-to-run different x.z sum on gfx950 (one row's product per launch; the same product
            // accumulated by a plain v_fma beside it was exact and stable) — tools/exp_pn_determinism.py
            pn_fmac(pp[0], a0, a0);
            pn_fmac(pp[1], a0, a1);
            pn_fmac(pp[2], a0, a2);
            pn_fmac(pp[3], a1, a1);
            pn_fmac(pp[4], a1, a2);
            pn_fmac(pp[5], a2, a2);
            pn_fmac(pp[6], a0, 1.0f);
            pn_fmac(pp[7], a1, 1.0f);
            pn_fmac(pp[8], a2, 1.0f);
          }
          yv = first4(yv);
        }
        pn_f32x2 v01 = __builtin_elementwise_max(__builtin_elementwise_fma(pn_f32x2{yv.x, yv.y}, sc01, sh01), zero2);
        pn_f32x2 v23 = __builtin_elementwise_max(__builtin_elementwise_fma(pn_f32x2{yv.z, yv.w}, sc23, sh23), zero2);
        if (!ok) v01 = v23 = zero2;  // rows past the part's end enter every product as zeros
        colsum.x += v01[0];
        colsum.y += v01[1];
        colsum.z += v23[0];
        colsum.w += v23[1];
        pn_bf16x4 ph, pm, pl;
        pn_split4v(v01, v23, ph, pm, pl);
        unsigned char* p = pa[b] + rl * SA + 8 * cy4;
        *reinterpret_cast<pn_bf16x4*>(p) = ph;
        *reinterpret_cast<pn_bf16x4*>(p + 2 * CIN) = pm;
        *reinterpret_cast<pn_bf16x4*>(p + 4 * CIN) = pl;
      }
    };
    auto stash = [&](int it, int b) {
      if (n0_of(it) + RB <= N) stash_t(it, b, std::true_type{});
      else stash_t(it, b, std::false_type{});
    };
    int m2 = part_of(2);
    if (n_it > 0) {
      fetch(0, part_of(0));
      stash(0, 0);
      if (n_it > 1) fetch(1, part_of(1));
    }
    __syncthreads();  // panel 0 and the weight image are complete
    PN_T_DECL
    for (int it = 0; it < n_it; ++it) {
      const int m3 = part_of(it + 3);  // (the part id of a unit is looked up two iterations before its rows are requested)
      if (it + 1 < n_it) stash(it + 1, (it + 1) & 1);
      if (it + 2 < n_it) fetch(it + 2, m2);
      m2 = m3;
      PN_BAR
    }
    PN_T_REPORT("stager")
    // ---- block totals (fixed order): column sums of A, BatchNorm-backward sums (FIRST: S, P^T P, psum) ----------------------
    float* scr = reinterpret_cast<float*>(&pz[0][0]);
    *reinterpret_cast<float4*>(scr + (0 * RG + ry0) * CIN + 4 * cy4) = colsum;
    if constexpr (FIRST) {
      if (cy4 == 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) scr[12 * RG * CIN + ry0 * 12 + e] = pp[e];
      }
    }
  } else if (wave < NS + ND) {
    // ============================================ input-gradient waves ========================================================
    // KPN waves share one 32 x 32 output tile of a unit: wave kp runs its share of the 12 / 8 k-steps (dZ . (alpha W) first, then
    // A . Q) and leaves its partial tile, raw, in outp[unit parity][kp] — the stagers finish it one barrier later.
    const int dw = wave - NS, kp = dw % KPN, tile = dw / KPN, rt = tile / CT, ct = tile % CT, d0 = 32 * ct;
    constexpr int KSTEPS = KZ + KA, KPER = KSTEPS / KPN;
    static_assert(KSTEPS % KPN == 0 && (KPN == 1 || KPN == 2), "k-split");
    pn_bf16x8 qh[KA], qm[KA], ql[KA];  // Q[16 ks + 8 hh + u][d0 + j] as h / m / l (only the steps this wave runs stay live)
#pragma unroll
    for (int ks = 0; ks < KA; ++ks)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float f = q[(long long)(16 * ks + 8 * hh + u) * CIN + d0 + j];
        qh[ks][u] = (__bf16)f;
        const float r1 = f - (float)qh[ks][u];
        qm[ks][u] = (__bf16)r1;
        ql[ks][u] = (__bf16)(r1 - (float)qm[ks][u]);
      }
    const float c0v = kp == KPN - 1 ? q[(long long)CIN * CIN + d0 + j] : 0.0f;  // c0 rides in one wave's accumulator
    __syncthreads();
    PN_T_DECL
    int m_prev = 0, m_cur = part_of(0);
    for (int it = 0; it < n_it; ++it) {
      const int b = it & 1;
      const int m_next = part_of(it + 1);
      if (it > 0) epilogue(it - 1, m_prev, Half0{});  // (the first half of the previous unit's rows)
      load_y(it, m_cur, Half0{});
      m_prev = m_cur;
      m_cur = m_next;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = c0v;
      const unsigned char* arow = pz[b] + (32 * rt + j) * SZ + 16 * hh;
      const unsigned char* brow = wl + (d0 + j) * SZ + 16 * hh;
      const unsigned char* qrow = pa[b] + (32 * rt + j) * SA + 16 * hh;
      // this wave's k-steps LO .. LO + KPER - 1 of the concatenated range; the fragments of step i + 1 are requested before
      // the products of step i are issued (written out per k-split role so that every index is a compile-time constant:
      // left to itself the compiler waited for each step's LDS reads in front of its six products)
      auto chain = [&](auto kp_tag) {
        constexpr int LO = decltype(kp_tag)::value * KPER;
        pn_bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int i = 0; i <= KPER; ++i) {
          if (i < KPER) {
            const int ks = LO + i;
            if (ks < KZ) {
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) {
                fa[i & 1][pl] = *reinterpret_cast<const pn_bf16x8*>(arow + 2 * K * pl + 32 * ks);
                fb[i & 1][pl] = *reinterpret_cast<const pn_bf16x8*>(brow + 2 * K * pl + 32 * ks);
              }
            } else {
#pragma unroll
              for (int pl = 0; pl < 3; ++pl)
                fa[i & 1][pl] = *reinterpret_cast<const pn_bf16x8*>(qrow + 2 * CIN * pl + 32 * (ks - KZ));
            }
          }
          if (i > 0) {
            const int ks = LO + i - 1, sb = (i - 1) & 1;
            if (ks < KZ) {
              PN_MFMA6(acc, fa[sb][0], fa[sb][1], fa[sb][2], fb[sb][0], fb[sb][1], fb[sb][2])
            } else {
              PN_MFMA6(acc, fa[sb][0], fa[sb][1], fa[sb][2], qh[ks - KZ], qm[ks - KZ], ql[ks - KZ])
            }
          }
        }
      };
      if (KPN == 1 || kp == 0) chain(std::integral_constant<int, 0>{});
      else chain(std::integral_constant<int, KPN - 1>{});
      float* o = &outp[b][0][0] + kp * OUTP + (32 * rt) * CIN + d0 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[acc_row(r, hh) * CIN] = acc[r];
      PN_BAR
    }
    if (n_it > 0) epilogue(n_it - 1, m_prev, Half0{});
    PN_T_REPORT("dgrad")
    put_sums(0);
  } else {
    // ============================================ weight-gradient waves =======================================================
    // Wave ww owns the output tiles PnWPlan<K>::t[ww][*] (compile-time lists chosen so that a wave's tiles share operand
    // fragments: one dZ column tile and the two A column tiles at most).  Per k-step (16 point rows) it requests the NEXT
    // step's fragments — ds_read_b64_tr_b16, two per plane — before it issues the products of the current one.
    const int ww = wave - NS - ND;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int trow = 8 * (g16 >> 1) + (i16 >> 2), tcol = 16 * (g16 & 1) + 4 * (i16 & 3);
    const int offz = trow * SZ + 2 * tcol, offa = trow * SA + 2 * tcol;
    using Plan = PnWPlan<K>;
    static_assert(NW == 4 && CT == 2, "the tile plans are written for four weight-gradient waves and a 64-wide input");
    constexpr int TPW = Plan::TPW;
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = f32x16{0};
    __syncthreads();
    PN_T_DECL
    auto run = [&](auto w_tag) {
      constexpr int W = decltype(w_tag)::value;
      constexpr int ZC = Plan::zc[W];                                  // the wave's dZ column tile
      constexpr bool A0 = Plan::uses(W, 8), A1 = Plan::uses(W, 9);     // which A column tiles it reads
      int m_prev = 0, m_cur = part_of(0);
      for (int it = 0; it < n_it; ++it) {
        const int b = it & 1;
        const int m_next = part_of(it + 1);
        if (it > 0) epilogue(it - 1, m_prev, Half1{});  // (the second half of the previous unit's rows)
        load_y(it, m_cur, Half1{});
        m_prev = m_cur;
        m_cur = m_next;
        const unsigned char* bz = pz[b] + offz + 64 * ZC;
        const unsigned char* ba = pa[b] + offa;
        pn_bf16x8 fz[2][3], f0[2][3], f1[2][3];
#pragma unroll
        for (int ks = 0; ks <= KW; ++ks) {
          if (ks < KW) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
              fz[ks & 1][pl] = pn_tr_frag(bz + 16 * ks * SZ + 2 * K * pl, SZ);
              if constexpr (A0) f0[ks & 1][pl] = pn_tr_frag(ba + 16 * ks * SA + 2 * CIN * pl, SA);
              if constexpr (A1) f1[ks & 1][pl] = pn_tr_frag(ba + 16 * ks * SA + 2 * CIN * pl + 64, SA);
            }
          }
          if (ks > 0) {
            const int sb = (ks - 1) & 1;
            pn_static_for<TPW>([&](auto i_tag) {
              constexpr int i = decltype(i_tag)::value;
              constexpr PnWTile t = Plan::t[W][i];
              if constexpr (t.a >= 0) {
                const pn_bf16x8(&fa)[3] = *(t.a < 8 ? &fz[sb] : (t.a == 8 ? &f0[sb] : &f1[sb]));
                const pn_bf16x8(&fb)[3] = *(t.b == 8 ? &f0[sb] : &f1[sb]);
                PN_MFMA6(acc[i], fa[0], fa[1], fa[2], fb[0], fb[1], fb[2])
              }
            });
          }
        }
        PN_BAR
      }
      if (n_it > 0) epilogue(n_it - 1, m_prev, Half1{});
      float* out = dwpart + (long long)blockIdx.x * ELEMS;
      pn_static_for<TPW>([&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        constexpr PnWTile t = Plan::t[W][i];
        if constexpr (t.a >= 0) {
          float* o = out + (t.kind == 0 ? 0 : K * CIN);
#pragma unroll
          for (int r = 0; r < 16; ++r) o[(32 * t.orow + acc_row(r, hh)) * CIN + 32 * t.ocol + j] = acc[i][r];
        }
      });
    };
    if (ww == 0) run(std::integral_constant<int, 0>{});
    else if (ww == 1) run(std::integral_constant<int, 1>{});
    else if (ww == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
    PN_T_REPORT("wgrad")
    put_sums(1);
  }
  __syncthreads();  // the end-of-block scratch (aliasing panel 0, which nobody reads any more) is complete
  {
    // scratch rows [(q * 2 + role) * RG + g][CIN]: q = 0 column sums of A (stagers: role 0 only), 1 s1, 2 s2, FIRST: 3..5 S's three
    // columns (role 0 input-gradient, role 1 weight-gradient waves), RG row groups g — added in this fixed order
    const float* scr = reinterpret_cast<const float*>(&pz[0][0]);
    float* out = dwpart + (long long)blockIdx.x * ELEMS;
    const int t = threadIdx.x;
    constexpr int NQ = FIRST ? 6 : 3;
    if (t < NQ * CIN) {
      const int qq = t / CIN, c = t % CIN;
      float s = 0.0f;
#pragma unroll
      for (int g = 0; g < (qq == 0 ? 1 : 2) * RG; ++g) s += scr[(qq * 2 * RG + g) * CIN + c];
      if (qq == 0) out[K * CIN + CIN * CIN + c] = s;
      else if (qq <= 2) partial[((long long)blockIdx.x * CIN + c) * 2 + (qq - 1)] = s;
      else out[K * CIN + CIN * CIN + CIN + 3 * c + (qq - 3)] = s;
    } else if (FIRST && t < NQ * CIN + 9) {
      const int e = t - NQ * CIN;
      float s = 0.0f;
#pragma unroll
      for (int g = 0; g < RG; ++g) s += scr[12 * RG * CIN + g * 12 + e];
      out[K * CIN + CIN * CIN + CIN + 192 + e] = s;
    }
  }
}

// ---- the never-stored last layer in the same form ---------------------------------------------------------------------------
// conv5's output Y5 is never stored (pointnet.hip: top-2 records), so its backward always was  dA4 = A4 Q + c0 + S W5  with S the
// sparse arg-max gradient (CSR by 32-row tile: erow, ech, eval, tptr) and  dW5  from the Gram matrix G = A4^T A4.  This kernel
// does the input gradient AND the Gram matrix in one pass over Y4 (pn_dgrad_split_kernel + pn_gram_split_kernel read it once
// each: 139 + 79 us) with the wave roles of pn_bwd_q_kernel: stagers convert Y4 rows into the A panel and stage the unit's
// CSR entries (three-stage request pipeline: tile offsets, entries, LDS); four input-gradient waves (one 32-column tile each,
// Q register-resident) leave A Q + c0 raw in LDS; four weight-gradient waves accumulate the ten upper tiles of G; both kinds
// then finish the previous unit: sparse rows added from the staged entries, ReLU mask, dZ4 stores, BatchNorm-backward sums.
struct PnGTile {
  signed char a, b;  // column tiles of A: row tile and column tile of G (a <= b)
};
struct PnGPlan {
  static constexpr int TPW = 3;
  static constexpr PnGTile t[4][3] = {{{0, 0}, {0, 1}, {0, 2}}, {{1, 1}, {1, 2}, {1, 3}}, {{0, 3}, {3, 3}, {-1, -1}}, {{2, 2}, {2, 3}, {-1, -1}}};
  static constexpr bool uses(int w, int tile) {
    for (int i = 0; i < TPW; ++i)
      if (t[w][i].a >= 0 && (t[w][i].a == tile || t[w][i].b == tile)) return true;
    return false;
  }
};
constexpr int kTopES = 32;  // CSR entries of a 32-row tile staged in LDS (more: read from memory by the epilogue, rare)

template <int CIN, int NS, int ND, int NW>
__global__ __launch_bounds__(64 * (NS + ND + NW), (NS + ND + NW + 3) / 4) void pn_bwd_top_q_kernel(
    const float* __restrict__ y_prev, const float* __restrict__ bn_prev, const float* __restrict__ q,
    const int* __restrict__ vlist, int N, float* __restrict__ dz_prev, float* __restrict__ partial,
    float* __restrict__ dwpart, const int* __restrict__ erow, const int* __restrict__ ech,
    const float* __restrict__ eval, const int* __restrict__ tptr, const float* __restrict__ w5, int F) {
  constexpr int K = 0;  // (PN_T_REPORT prints them)
  constexpr bool FIRST = false;
  constexpr int RB = 32, SA = 6 * CIN + 16, CT = CIN / 32, KA = CIN / 16, KW = RB / 16;
  constexpr int NTS = 64 * NS, QC = CIN / 4, RG = NTS / QC, NLY = RB * QC / NTS, EH = NLY;  // (the Gram waves run the whole epilogue: the
  // input-gradient waves hold Q — 96 registers — and spilled with the epilogue's state on top)
  constexpr int ELEMS = CIN * CIN + CIN;
  static_assert(CIN == 128 && ND == CT && NW == 4 && NS == 4 && NLY % 2 == 0, "shapes");
  __shared__ __attribute__((aligned(16))) unsigned char pa[2][RB * SA];  // A planes h | m | l
  __shared__ __attribute__((aligned(16))) float outp[2][RB * CIN];       // A Q + c0 of a unit, row-major
  __shared__ __attribute__((aligned(16))) float rowsum[3][RB * CIN];     // S W5 of a unit, row-major: [unit % 3] (built by the stagers)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, j = lane & 31, hh = lane >> 5;
  const int TB = (N + RB - 1) / RB, U = vlist[0] * TB, G = gridDim.x;
  const int n_it = (int)blockIdx.x < U ? (U - (int)blockIdx.x + G - 1) / G : 0;
  auto part_of = [&](int it) {
    const int u = (int)blockIdx.x + it * G;
    return u < U && it >= 0 ? vlist[4 + u / TB] : 0;
  };
  auto n0_of = [&](int it) { return (((int)blockIdx.x + it * G) % TB) * RB; };
  // ---- the input gradient's epilogue (see pn_bwd_q_kernel), run by the Gram waves ---------------------------------------------
  const int te = threadIdx.x & (NTS - 1);
  const int cy4 = te % QC, ry0 = te / QC;
  const float4 sc = reinterpret_cast<const float4*>(bn_prev)[cy4];
  const float4 sh = reinterpret_cast<const float4*>(bn_prev + CIN)[cy4];
  const float4 mn = reinterpret_cast<const float4*>(bn_prev + 2 * CIN)[cy4];
  float4 ye[EH];
  float4 s1v = make_float4(0.0f, 0.0f, 0.0f, 0.0f), t2v = s1v;
  auto load_y = [&](int it, int m, auto half_tag) {
    constexpr int HALF = decltype(half_tag)::value;
    const int n0 = n0_of(it);
    const long long row0 = (long long)m * N + n0;
#pragma unroll
    for (int i = 0; i < EH; ++i) {
      const int rl = ry0 + (HALF * EH + i) * RG;
      const int rr = n0 + rl < N ? rl : N - 1 - n0;
      ye[i] = reinterpret_cast<const float4*>(y_prev)[(row0 + rr) * QC + cy4];
    }
  };
  auto epilogue = [&](int it, int m, auto half_tag) {
    constexpr int HALF = decltype(half_tag)::value;
    const int n0 = n0_of(it), slot = it % 3;
    const long long row0 = (long long)m * N + n0;
    const float* ob = &outp[it & 1][0];
#pragma unroll
    for (int i = 0; i < EH; ++i) {  // one row at a time: its dense part, + S W5 (summed per row by the stagers an iteration ago)
      const int rl = ry0 + (HALF * EH + i) * RG;
      const bool ok = n0 + rl < N;
      float4 o = *reinterpret_cast<const float4*>(ob + rl * CIN + 4 * cy4);
      const float4 sp = *reinterpret_cast<const float4*>(&rowsum[slot][rl * CIN + 4 * cy4]);
      o.x += sp.x;
      o.y += sp.y;
      o.z += sp.z;
      o.w += sp.w;
      const float4 yv = ye[i];
      float4 d;
      d.x = (ok && __builtin_fmaf(yv.x, sc.x, sh.x) > 0.0f) ? o.x : 0.0f;
      d.y = (ok && __builtin_fmaf(yv.y, sc.y, sh.y) > 0.0f) ? o.y : 0.0f;
      d.z = (ok && __builtin_fmaf(yv.z, sc.z, sh.z) > 0.0f) ? o.z : 0.0f;
      d.w = (ok && __builtin_fmaf(yv.w, sc.w, sh.w) > 0.0f) ? o.w : 0.0f;
      if (ok) reinterpret_cast<float4*>(dz_prev)[(row0 + rl) * QC + cy4] = d;
      s1v.x += d.x;
      s1v.y += d.y;
      s1v.z += d.z;
      s1v.w += d.w;
      t2v.x = __builtin_fmaf(d.x, yv.x - mn.x, t2v.x);
      t2v.y = __builtin_fmaf(d.y, yv.y - mn.y, t2v.y);
      t2v.z = __builtin_fmaf(d.z, yv.z - mn.z, t2v.z);
      t2v.w = __builtin_fmaf(d.w, yv.w - mn.w, t2v.w);
    }
  };
  auto put_sums = [&]() {  // scratch rows [q * RG + row group]: q = 1 s1, 2 s2
    float* scr = reinterpret_cast<float*>(&pa[0][0]);
    const float4 is4 = reinterpret_cast<const float4*>(bn_prev + 3 * CIN)[cy4];
    t2v.x *= is4.x;
    t2v.y *= is4.y;
    t2v.z *= is4.z;
    t2v.w *= is4.w;
    *reinterpret_cast<float4*>(scr + (1 * RG + ry0) * CIN + 4 * cy4) = s1v;
    *reinterpret_cast<float4*>(scr + (2 * RG + ry0) * CIN + 4 * cy4) = t2v;
  };
  using Half0 = std::integral_constant<int, 0>;
  using Half1 = std::integral_constant<int, 1>;

  if (wave < NS) {
    // ================================================ stager waves ==========================================================
    const int t = threadIdx.x;
    const int T1 = (N + 31) / 32 + 1;
    float4 ry[NLY];
    float4 colsum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const pn_f32x2 sc01 = {sc.x, sc.y}, sc23 = {sc.z, sc.w}, sh01 = {sh.x, sh.y}, sh23 = {sh.z, sh.w}, zero2 = {0.0f, 0.0f};
    auto fetch = [&](int it, int m) {
      const int n0 = n0_of(it);
      const long long row0 = (long long)m * N + n0;
#pragma unroll
      for (int i = 0; i < NLY; ++i) {
        const int rl = ry0 + i * RG;
        const int rr = n0 + rl < N ? rl : N - 1 - n0;
        ry[i] = reinterpret_cast<const float4*>(y_prev)[(row0 + rr) * QC + cy4];
      }
    };
    auto stash = [&](int it, int b) {
      const int n0 = n0_of(it);
#pragma unroll
      for (int i = 0; i < NLY; ++i) {
        const int rl = ry0 + i * RG;
        const bool ok = n0 + rl < N;
        pn_f32x2 v01 = __builtin_elementwise_max(__builtin_elementwise_fma(pn_f32x2{ry[i].x, ry[i].y}, sc01, sh01), zero2);
        pn_f32x2 v23 = __builtin_elementwise_max(__builtin_elementwise_fma(pn_f32x2{ry[i].z, ry[i].w}, sc23, sh23), zero2);
        if (!ok) v01 = v23 = zero2;  // rows past the part's end enter every product as zeros
        colsum.x += v01[0];
        colsum.y += v01[1];
        colsum.z += v23[0];
        colsum.w += v23[1];
        pn_bf16x4 ph, pm, pl;
        pn_split4v(v01, v23, ph, pm, pl);
        unsigned char* p = pa[b] + rl * SA + 8 * cy4;
        *reinterpret_cast<pn_bf16x4*>(p) = ph;
        *reinterpret_cast<pn_bf16x4*>(p + 2 * CIN) = pm;
        *reinterpret_cast<pn_bf16x4*>(p + 4 * CIN) = pl;
      }
    };
    // S W5 of a unit, per row: request pipeline inside every stager wave (no cross-wave hand-over): tile offsets of unit u at
    // iteration u - 3, its entries at u - 2 (lane l holds entry l & 31: row, channel, alpha * grad), the row sums at u - 1 —
    // entry e is broadcast with v_readlane, a matching row costs one weight-row load from the L2, here, in the waves that wait
    // for memory anyway.  Entry order = channel order, fixed.  The epilogue reads the sums at u + 1: three slots.
    int tp0 = 0, tp1 = 0, e_row = 0, e_ch = 0, e_cnt = 0, e_pb = 0;
    float e_val = 0.0f;
    auto tp_fetch = [&](int it, int m) {
      const int st = n0_of(it) >> 5;
      tp0 = tptr[(long long)m * T1 + st];
      tp1 = tptr[(long long)m * T1 + st + 1];
    };
    auto ent_fetch = [&](int it, int m) {  // uses the offsets of this unit (tp0 / tp1 hold them now)
      e_cnt = tp1 - tp0;
      e_pb = tp0;
      const int l = lane & 31, e = l < e_cnt ? l : 0;
      const long long g = (long long)m * F + tp0 + e;
      const bool has = e_cnt > 0;
      e_row = has ? erow[g] - n0_of(it) : -1;
      e_ch = has ? ech[g] : 0;
      e_val = has ? eval[g] : 0.0f;
    };
    auto rowsum_build = [&](int it, int m) {  // this thread's NLY rows x 4 columns (e_* hold this unit's entries now)
      const int slot = it % 3, n0 = n0_of(it);
      const int cnt = __builtin_amdgcn_readfirstlane(e_cnt), pb = __builtin_amdgcn_readfirstlane(e_pb);
      float4 acc[NLY];
#pragma unroll
      for (int i = 0; i < NLY; ++i) acc[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      for (int e = 0; e < cnt; ++e) {
        int er, ec;
        float ev;
        if (e < kTopES) {
          er = __builtin_amdgcn_readlane(e_row, e);
          ec = __builtin_amdgcn_readlane(e_ch, e);
          ev = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e_val), e));
        } else {  // unusually crowded tile
          const long long g = (long long)m * F + pb + e;
          er = erow[g] - n0;
          ec = ech[g];
          ev = eval[g];
        }
#pragma unroll
        for (int i = 0; i < NLY; ++i)
          if (er == ry0 + i * RG) {
            const float4 wv = reinterpret_cast<const float4*>(w5 + (long long)ec * CIN)[cy4];
            acc[i].x = __builtin_fmaf(ev, wv.x, acc[i].x);
            acc[i].y = __builtin_fmaf(ev, wv.y, acc[i].y);
            acc[i].z = __builtin_fmaf(ev, wv.z, acc[i].z);
            acc[i].w = __builtin_fmaf(ev, wv.w, acc[i].w);
          }
      }
#pragma unroll
      for (int i = 0; i < NLY; ++i) *reinterpret_cast<float4*>(&rowsum[slot][(ry0 + i * RG) * CIN + 4 * cy4]) = acc[i];
    };
    int m1 = part_of(1), m2 = part_of(2), m3 = part_of(3);
    if (n_it > 0) {
      const int m0 = part_of(0);
      tp_fetch(0, m0);
      fetch(0, m0);
      ent_fetch(0, m0);
      if (n_it > 1) tp_fetch(1, m1);
      rowsum_build(0, m0);
      if (n_it > 1) ent_fetch(1, m1);
      if (n_it > 2) tp_fetch(2, m2);
      stash(0, 0);
      if (n_it > 1) fetch(1, m1);
    }
    __syncthreads();  // panel 0 and unit 0's sparse rows are complete
    PN_T_DECL
    for (int it = 0; it < n_it; ++it) {
      const int m4 = part_of(it + 4);
      if (it + 1 < n_it) {
        rowsum_build(it + 1, m1);  // (its entries were requested an iteration ago, its offsets two)
        stash(it + 1, (it + 1) & 1);
      }
      if (it + 2 < n_it) {
        ent_fetch(it + 2, m2);
        fetch(it + 2, m2);
      }
      if (it + 3 < n_it) tp_fetch(it + 3, m3);
      m1 = m2;
      m2 = m3;
      m3 = m4;
      PN_BAR
    }
    PN_T_REPORT("stager")
    float* scr = reinterpret_cast<float*>(&pa[0][0]);
    *reinterpret_cast<float4*>(scr + (0 * RG + ry0) * CIN + 4 * cy4) = colsum;
  } else if (wave < NS + ND) {
    // ============================================ input-gradient waves ========================================================
    const int ct = wave - NS, d0 = 32 * ct;
    pn_bf16x8 qh[KA], qm[KA], ql[KA];  // Q[16 ks + 8 hh + u][d0 + j] as h / m / l
#pragma unroll
    for (int ks = 0; ks < KA; ++ks)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float f = q[(long long)(16 * ks + 8 * hh + u) * CIN + d0 + j];
        qh[ks][u] = (__bf16)f;
        const float r1 = f - (float)qh[ks][u];
        qm[ks][u] = (__bf16)r1;
        ql[ks][u] = (__bf16)(r1 - (float)qm[ks][u]);
      }
    const float c0v = q[(long long)CIN * CIN + d0 + j];
    __syncthreads();
    PN_T_DECL
    for (int it = 0; it < n_it; ++it) {
      const int b = it & 1;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = c0v;
      const unsigned char* qrow = pa[b] + j * SA + 16 * hh;
      pn_bf16x8 fa[2][3];
#pragma unroll
      for (int i = 0; i <= KA; ++i) {
        if (i < KA) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fa[i & 1][pl] = *reinterpret_cast<const pn_bf16x8*>(qrow + 2 * CIN * pl + 32 * i);
        }
        if (i > 0) {
          const int sb = (i - 1) & 1;
          PN_MFMA6(acc, fa[sb][0], fa[sb][1], fa[sb][2], qh[i - 1], qm[i - 1], ql[i - 1])
        }
      }
      float* o = &outp[b][0] + d0 + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[acc_row(r, hh) * CIN] = acc[r];
      PN_BAR
    }
    PN_T_REPORT("dgrad")
  } else {
    // ============================================ Gram waves ==================================================================
    const int ww = wave - NS - ND;
    const int g16 = lane >> 4, i16 = lane & 15;
    const int trow = 8 * (g16 >> 1) + (i16 >> 2), tcol = 16 * (g16 & 1) + 4 * (i16 & 3);
    const int offa = trow * SA + 2 * tcol;
    constexpr int TPW = PnGPlan::TPW;
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = f32x16{0};
    __syncthreads();
    PN_T_DECL
    auto run = [&](auto w_tag) {
      constexpr int W = decltype(w_tag)::value;
      int m_prev = 0, m_cur = part_of(0);
      for (int it = 0; it < n_it; ++it) {
        const int b = it & 1;
        const int m_next = part_of(it + 1);
        if (it > 0) epilogue(it - 1, m_prev, Half0{});
        load_y(it, m_cur, Half0{});
        m_prev = m_cur;
        m_cur = m_next;
        const unsigned char* ba = pa[b] + offa;
        pn_bf16x8 f[4][3];  // fragments of A's four column tiles (only the ones this wave's tiles use are loaded; one
        // k-step at a time: these waves have the slack, and the registers go to the epilogue's state)
#pragma unroll
        for (int ks = 0; ks < KW; ++ks) {
          pn_static_for<4>([&](auto c_tag) {
            constexpr int c = decltype(c_tag)::value;
            if constexpr (PnGPlan::uses(W, c)) {
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) f[c][pl] = pn_tr_frag(ba + 16 * ks * SA + 2 * CIN * pl + 64 * c, SA);
            }
          });
          pn_static_for<TPW>([&](auto i_tag) {
            constexpr int i = decltype(i_tag)::value;
            constexpr PnGTile tl = PnGPlan::t[W][i];
            if constexpr (tl.a >= 0) {
              PN_MFMA6(acc[i], f[tl.a][0], f[tl.a][1], f[tl.a][2], f[tl.b][0], f[tl.b][1], f[tl.b][2])
            }
          });
        }
        PN_BAR
      }
      if (n_it > 0) epilogue(n_it - 1, m_prev, Half0{});
      float* out = dwpart + (long long)blockIdx.x * ELEMS;  // the full matrix: upper tiles and their mirror images
      pn_static_for<TPW>([&](auto i_tag) {
        constexpr int i = decltype(i_tag)::value;
        constexpr PnGTile tl = PnGPlan::t[W][i];
        if constexpr (tl.a >= 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ra = 32 * tl.a + acc_row(r, hh), cb = 32 * tl.b + j;
            out[ra * CIN + cb] = acc[i][r];
            if (tl.a != tl.b) out[cb * CIN + ra] = acc[i][r];
          }
        }
      });
    };
    if (ww == 0) run(std::integral_constant<int, 0>{});
    else if (ww == 1) run(std::integral_constant<int, 1>{});
    else if (ww == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
    PN_T_REPORT("gram")
    put_sums();
  }
  __syncthreads();  // the end-of-block scratch (aliasing the panels, which nobody reads any more) is complete
  {
    const float* scr = reinterpret_cast<const float*>(&pa[0][0]);
    float* out = dwpart + (long long)blockIdx.x * ELEMS;
    const int t = threadIdx.x;
    if (t < 3 * CIN) {
      const int qq = t / CIN, c = t % CIN;
      float s = 0.0f;
#pragma unroll
      for (int g = 0; g < RG; ++g) s += scr[(qq * RG + g) * CIN + c];
      if (qq == 0) out[CIN * CIN + c] = s;
      else partial[((long long)blockIdx.x * CIN + c) * 2 + (qq - 1)] = s;
    }
  }
}

// The end of the pass: weight gradients of conv4..conv2 (and conv1) from the reduced tables of their layers,
//     dW_l[c][d] = alpha_c T[c][d] + gammap_c sum_k W_l[c][k] G[k][d] + betap_c asum[d]
//     dW_1[c][k] = alpha_c S[c][k]  + gammap_c sum_j W_1[c][j] (P^T P)[j][k] + betap_c psum[k].
// blocks [first[i], first[i + 1]) serve layer i (256 output elements per block); the last block serves conv1.
struct PnFinish {
  const float* red[3];   // reduced tables [T | G | asum | (S, P^T P, psum)]
  const float* w[3];     // the layers' weights [K][CIN]
  const float* coef[3];  // [3][K]
  float* dw[3];
  int K[3], CIN[3], first[4];
  const float* w1;       // conv1 [64][3]
  const float* coef1;    // [3][64]
  float* dw1;
  int first_layer;       // index of the layer whose table carries S / P^T P (conv2)
};
__global__ __launch_bounds__(256) void pn_bwd_finish_kernel(const PnFinish f) {
  if ((int)blockIdx.x == f.first[3]) {  // conv1
    const int t = threadIdx.x;
    if (t >= 192) return;
    const int c = t / 3, k = t % 3, i = f.first_layer;
    const float* x = f.red[i] + f.K[i] * f.CIN[i] + f.CIN[i] * f.CIN[i] + f.CIN[i];
    const float* pp = x + 192;  // xx xy xz yy yz zz, x y z
    float wg = 0.0f;
#pragma unroll
    for (int jx = 0; jx < 3; ++jx) {  // (a, b), a <= b, in the order xx xy xz yy yz zz
      const int a = jx < k ? jx : k, b = jx < k ? k : jx;
      wg = __builtin_fmaf(f.w1[c * 3 + jx], pp[a == 0 ? b : (a == 1 ? 2 + b : 5)], wg);
    }
    f.dw1[t] = f.coef1[c] * x[t] + f.coef1[64 + c] * wg + f.coef1[128 + c] * pp[6 + k];
    return;
  }
  int i = 0;
#pragma unroll
  for (int qq = 1; qq < 3; ++qq) i += (int)blockIdx.x >= f.first[qq] ? 1 : 0;
  const int K = f.K[i], CIN = f.CIN[i];
  const int e = ((int)blockIdx.x - f.first[i]) * 256 + threadIdx.x;
  if (e >= K * CIN) return;
  const int c = e / CIN, d = e % CIN;
  const float* T = f.red[i];
  const float* Gm = T + K * CIN;
  const float* asum = Gm + CIN * CIN;
  const float* wr = f.w[i] + (long long)c * CIN;
  float wg = 0.0f;
#pragma unroll 8
  for (int k = 0; k < CIN; ++k) {  // G is stored by its upper 32 x 32 tiles only (symmetric)
    const float g = (k >> 5) <= (d >> 5) ? Gm[k * CIN + d] : Gm[d * CIN + k];
    wg = __builtin_fmaf(wr[k], g, wg);
  }
  f.dw[i][e] = f.coef[i][c] * T[e] + f.coef[i][K + c] * wg + f.coef[i][2 * K + c] * asum[d];
}
