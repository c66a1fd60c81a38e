// Probe: issue rate of the instructions the leaf search's inner loop is made of (wave64, 8 waves per SIMD): plain VALU,
// DPP row rotations, 64-bit compares, v_cndmask on VCC, v_readlane, packed fp32.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 dpp_rate.hip -o dpp_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = seed * 0.5f;
  unsigned long long k0 = threadIdx.x * 77ull, k1 = threadIdx.x * 131ull + 5;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 8 independent v_sub_f32
      REP8(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                        "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 1) {  // v_subrev_f32_dpp row_ror (source b is never written)
      REP8(asm volatile("v_subrev_f32_dpp %0, %8, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n v_subrev_f32_dpp %1, %8, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n"
                        "v_subrev_f32_dpp %2, %8, %2 row_ror:3 row_mask:0xf bank_mask:0xf\n v_subrev_f32_dpp %3, %8, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                        "v_subrev_f32_dpp %4, %8, %4 row_ror:5 row_mask:0xf bank_mask:0xf\n v_subrev_f32_dpp %5, %8, %5 row_ror:6 row_mask:0xf bank_mask:0xf\n"
                        "v_subrev_f32_dpp %6, %8, %6 row_ror:7 row_mask:0xf bank_mask:0xf\n v_subrev_f32_dpp %7, %8, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 2) {  // v_mov_b32_dpp row_ror
      REP8(asm volatile("v_mov_b32_dpp %0, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_ror:2 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %2, %8 row_ror:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %4, %8 row_ror:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_ror:6 row_mask:0xf bank_mask:0xf\n"
                        "v_mov_b32_dpp %6, %8 row_ror:7 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 3) {  // v_cmp_lt_u64 (8 compares)
      REP8(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %0\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %0\n"
                        "v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %0\n v_cmp_lt_u64 vcc, %0, %1\n v_cmp_lt_u64 vcc, %1, %0\n"
                        : : "v"(k0), "v"(k1) : "vcc");)
    } else if (MODE == 4) {  // v_cmp_lt_f32 (8 compares)
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0\n v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0\n"
                        "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0\n v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %0\n"
                        : : "v"(a0), "v"(a1) : "vcc");)
    } else if (MODE == 5) {  // cmp + 2 cndmask dependent (the update chain): 8 x (1 + 2)
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %2, %2, %0, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
    } else if (MODE == 6) {  // v_readlane_b32 x8
      REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 5\n v_readlane_b32 s22, %0, 7\n v_readlane_b32 s23, %0, 9\n"
                        "v_readlane_b32 s24, %0, 11\n v_readlane_b32 s25, %0, 13\n v_readlane_b32 s26, %0, 15\n v_readlane_b32 s27, %0, 17\n"
                        : : "v"(a0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (MODE == 7) {  // v_min3_f32 x8
      REP8(asm volatile("v_min3_f32 %0, %0, %8, %1\n v_min3_f32 %1, %1, %8, %2\n v_min3_f32 %2, %2, %8, %3\n v_min3_f32 %3, %3, %8, %4\n"
                        "v_min3_f32 %4, %4, %8, %5\n v_min3_f32 %5, %5, %8, %6\n v_min3_f32 %6, %6, %8, %7\n v_min3_f32 %7, %7, %8, %0\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 8) {  // v_mul_f32 with DPP quad_perm
      REP8(asm volatile("v_mul_f32_dpp %0, %8, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %8, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_mul_f32_dpp %2, %8, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %8, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_mul_f32_dpp %4, %8, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %8, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_mul_f32_dpp %6, %8, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %7, %8, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (MODE == 9) {  // s_load_dwordx16 stream? no: ds_bpermute x8
      REP8(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                        "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(k0 + k1);
}

template <int MODE>
void run(int blocks, const char* name, int instr_per_iter) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
  hipEventRecord(a, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double winstr = (double)blocks * 4 * iters * instr_per_iter;  // wave-instructions
  printf("%-34s blocks=%4d: %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, blocks, ms,
         ms * 1e-3 * 2.4e9 * 1024 / winstr);
  hipFree(out);
}
int main() {
  for (int blocks : {512, 2048}) {
    run<0>(blocks, "v_sub_f32", 64);
    run<1>(blocks, "v_subrev_f32_dpp row_ror", 64);
    run<2>(blocks, "v_mov_b32_dpp row_ror", 64);
    run<8>(blocks, "v_mul_f32_dpp quad_perm", 64);
    run<3>(blocks, "v_cmp_lt_u64", 64);
    run<4>(blocks, "v_cmp_lt_f32", 64);
    run<5>(blocks, "cmp + nop + 2 cndmask (3 instr)", 24);
    run<6>(blocks, "v_readlane_b32", 64);
    run<7>(blocks, "v_min3_f32", 64);
    run<9>(blocks, "ds_bpermute_b32", 64);
  }
  return 0;
}
