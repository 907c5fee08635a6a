// valu_rate.hip -- measures VALU issue cost (shader cycles per wave64 instruction per SIMD)
// for the instruction kinds the demod kernel uses, at 8 waves per SIMD on every CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define REP4(X) X X X X

template <int KIND>
__global__ __launch_bounds__(256, 8) void k(float *out, unsigned long long *cyc, unsigned long long *rt, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    int i0 = threadIdx.x * 0x01010101;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        unsigned long long msk = 0x5555555555555555ull + blockIdx.x;
        if (KIND == 0) { asm volatile(REP4("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) ); }
        else if (KIND == 2) { asm volatile(REP4("v_cvt_f32_i32_sdwa %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %1, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %2, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %3, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %4, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %5, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %6, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_cvt_f32_i32_sdwa %7, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0) ); }
        else if (KIND == 3) { asm volatile(REP4("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) ); }
        else if (KIND == 4) { asm volatile(REP4("v_mov_b32_dpp %0, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) ); }
        else if (KIND == 5) { asm volatile(REP4("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc"); }
        else if (KIND == 6) { asm volatile(REP4("v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(b) ); }
        else if (KIND == 7) { asm volatile(REP4("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) ); }
        else if (KIND == 8) { asm volatile(REP4("v_max3_f32 %0, |%0|, |%8|, %9\n v_max3_f32 %1, |%1|, |%8|, %9\n v_max3_f32 %2, |%2|, |%8|, %9\n v_max3_f32 %3, |%3|, |%8|, %9\n v_max3_f32 %4, |%4|, |%8|, %9\n v_max3_f32 %5, |%5|, |%8|, %9\n v_max3_f32 %6, |%6|, |%8|, %9\n v_max3_f32 %7, |%7|, |%8|, %9\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) ); }
        else if (KIND == 9) { asm volatile(REP4("v_fmaak_f32 %0, %0, %8, 0x3dc578e2\n v_fmaak_f32 %1, %1, %8, 0x3dc578e2\n v_fmaak_f32 %2, %2, %8, 0x3dc578e2\n v_fmaak_f32 %3, %3, %8, 0x3dc578e2\n v_fmaak_f32 %4, %4, %8, 0x3dc578e2\n v_fmaak_f32 %5, %5, %8, 0x3dc578e2\n v_fmaak_f32 %6, %6, %8, 0x3dc578e2\n v_fmaak_f32 %7, %7, %8, 0x3dc578e2\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) ); }
        else if (KIND == 10) { asm volatile(REP4("v_cmp_gt_f32 vcc, |%0|, |%1|\n v_cmp_gt_f32 vcc, |%1|, |%0|\n v_cmp_gt_f32 vcc, |%0|, |%1|\n v_cmp_gt_f32 vcc, |%1|, |%0|\n v_cmp_gt_f32 vcc, |%0|, |%1|\n v_cmp_gt_f32 vcc, |%1|, |%0|\n v_cmp_gt_f32 vcc, |%0|, |%1|\n v_cmp_gt_f32 vcc, |%1|, |%0|\n") :: "v"(a0), "v"(a1) : "vcc"); }
        else if (KIND == 11) { asm volatile(REP4("v_bitop3_b32 %0, %0, %8, %9 bitop3:0xa6\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0xa6\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0xa6\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0xa6\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0xa6\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0xa6\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0xa6\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0xa6\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(b) ); }
        else if (KIND == 13) { asm volatile(REP4("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(msk) ); }
        else if (KIND == 14) { asm volatile(REP4("v_sub_f32 %0, 0x40490fdb, %0\n v_sub_f32 %1, 0x40490fdb, %1\n v_sub_f32 %2, 0x40490fdb, %2\n v_sub_f32 %3, 0x40490fdb, %3\n v_sub_f32 %4, 0x40490fdb, %4\n v_sub_f32 %5, 0x40490fdb, %5\n v_sub_f32 %6, 0x40490fdb, %6\n v_sub_f32 %7, 0x40490fdb, %7\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) ); }
        else if (KIND == 15) { asm volatile(REP4("v_min_f32_e64 %0, |%0|, |%8|\n v_min_f32_e64 %1, |%1|, |%8|\n v_min_f32_e64 %2, |%2|, |%8|\n v_min_f32_e64 %3, |%3|, |%8|\n v_min_f32_e64 %4, |%4|, |%8|\n v_min_f32_e64 %5, |%5|, |%8|\n v_min_f32_e64 %6, |%6|, |%8|\n v_min_f32_e64 %7, |%7|, |%8|\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) ); }
        else if (KIND == 16) { asm volatile(REP4("v_cmp_gt_i32 vcc, 0, %0\n v_cmp_gt_i32 vcc, 0, %1\n v_cmp_gt_i32 vcc, 0, %2\n v_cmp_gt_i32 vcc, 0, %3\n v_cmp_gt_i32 vcc, 0, %4\n v_cmp_gt_i32 vcc, 0, %5\n v_cmp_gt_i32 vcc, 0, %6\n v_cmp_gt_i32 vcc, 0, %7\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc"); }
        else if (KIND == 17) { asm volatile(REP4("v_add_u32 %0, 0x80808081, %0\n v_add_u32 %1, 0x80808081, %1\n v_add_u32 %2, 0x80808081, %2\n v_add_u32 %3, 0x80808081, %3\n v_add_u32 %4, 0x80808081, %4\n v_add_u32 %5, 0x80808081, %5\n v_add_u32 %6, 0x80808081, %6\n v_add_u32 %7, 0x80808081, %7\n ") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) ); }
        else if (KIND == 20) { asm volatile(REP4("v_cmp_gt_f32 vcc, |%0|, |%8|\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, |%2|, |%8|\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_gt_f32 vcc, |%4|, |%8|\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, |%6|, |%8|\n v_cndmask_b32 %7, %7, %8, vcc\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc"); }
        else if (KIND == 21) { unsigned long long m0, m1; asm volatile(REP4("v_cmp_gt_f32 %8, |%0|, |%10|\n v_cndmask_b32_e64 %1, %1, %10, %8\n v_cmp_gt_f32 %9, |%2|, |%10|\n v_cndmask_b32_e64 %3, %3, %10, %9\n v_cmp_gt_f32 %8, |%4|, |%10|\n v_cndmask_b32_e64 %5, %5, %10, %8\n v_cmp_gt_f32 %9, |%6|, |%10|\n v_cndmask_b32_e64 %7, %7, %10, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(m0), "=&s"(m1) : "v"(b)); }
        else if (KIND == 22) { asm volatile(REP4("v_ashrrev_i32 %0, 31, %1\n v_and_b32 %0, 0x3fc90fdb, %0\n v_sub_f32 %1, %1, %0\n v_ashrrev_i32 %2, 31, %3\n v_and_b32 %2, 0x3fc90fdb, %2\n v_sub_f32 %3, %3, %2\n v_ashrrev_i32 %4, 31, %5\n v_and_b32 %4, 0x3fc90fdb, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        else if (KIND == 23) { asm volatile(REP4("v_cndmask_b32 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc"); }
        else if (KIND == 24) { asm volatile(REP4("v_sub_f32_e64 %0, |%0|, |%8|\n v_sub_f32_e64 %1, |%1|, |%8|\n v_sub_f32_e64 %2, |%2|, |%8|\n v_sub_f32_e64 %3, |%3|, |%8|\n v_sub_f32_e64 %4, |%4|, |%8|\n v_sub_f32_e64 %5, |%5|, |%8|\n v_sub_f32_e64 %6, |%6|, |%8|\n v_sub_f32_e64 %7, |%7|, |%8|\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        else if (KIND == 25) { asm volatile(REP4("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, |%1|, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, |%3|, %8, %3\n v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, |%5|, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, |%7|, %8, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        else if (KIND == 30) { asm volatile(REP4("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n v_cvt_f32_ubyte0 %4, %8\n v_cvt_f32_ubyte1 %5, %8\n v_cvt_f32_ubyte2 %6, %8\n v_cvt_f32_ubyte3 %7, %8\n") : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(i0)); }
        else if (KIND == 31) { asm volatile(REP4("v_mul_legacy_f32 %0, |%0|, |%8|\n v_mul_legacy_f32 %1, |%1|, |%8|\n v_mul_legacy_f32 %2, |%2|, |%8|\n v_mul_legacy_f32 %3, |%3|, |%8|\n v_mul_legacy_f32 %4, |%4|, |%8|\n v_mul_legacy_f32 %5, |%5|, |%8|\n v_mul_legacy_f32 %6, |%6|, |%8|\n v_mul_legacy_f32 %7, |%7|, |%8|\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        else if (KIND == 32) { asm volatile(REP4("v_perm_b32 %0, %8, %9, %0\n v_perm_b32 %1, %8, %9, %1\n v_perm_b32 %2, %8, %9, %2\n v_perm_b32 %3, %8, %9, %3\n v_perm_b32 %4, %8, %9, %4\n v_perm_b32 %5, %8, %9, %5\n v_perm_b32 %6, %8, %9, %6\n v_perm_b32 %7, %8, %9, %7\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(b)); }
        else if (KIND == 33) { asm volatile(REP4("v_cvt_f32_i32 %0, %8\n v_cvt_f32_i32 %1, %8\n v_cvt_f32_i32 %2, %8\n v_cvt_f32_i32 %3, %8\n v_cvt_f32_i32 %4, %8\n v_cvt_f32_i32 %5, %8\n v_cvt_f32_i32 %6, %8\n v_cvt_f32_i32 %7, %8\n") : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(i0)); }
        else if (KIND == 34) { asm volatile(REP4("v_bfe_i32 %0, %8, 8, 8\n v_bfe_i32 %1, %8, 16, 8\n v_bfe_i32 %2, %8, 8, 8\n v_bfe_i32 %3, %8, 16, 8\n v_bfe_i32 %4, %8, 8, 8\n v_bfe_i32 %5, %8, 16, 8\n v_bfe_i32 %6, %8, 8, 8\n v_bfe_i32 %7, %8, 16, 8\n") : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(i0)); }
        else if (KIND == 35) { asm volatile(REP4("v_rcp_f32 %0, %0\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)); }
        else if (KIND == 36) { asm volatile(REP4("v_cvt_f32_i32_sdwa %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_fma_f32 %1, %1, %9, %9\n v_fma_f32 %2, %2, %9, %9\n v_fma_f32 %3, %3, %9, %9\n v_cvt_f32_i32_sdwa %4, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_fma_f32 %5, %5, %9, %9\n v_fma_f32 %6, %6, %9, %9\n v_fma_f32 %7, %7, %9, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(b)); }
        else if (KIND == 37) { asm volatile(REP4("v_cvt_f32_ubyte1 %0, %8\n v_fma_f32 %1, %1, %9, %9\n v_fma_f32 %2, %2, %9, %9\n v_fma_f32 %3, %3, %9, %9\n v_cvt_f32_ubyte2 %4, %8\n v_fma_f32 %5, %5, %9, %9\n v_fma_f32 %6, %6, %9, %9\n v_fma_f32 %7, %7, %9, %9\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(b)); }
        else if (KIND == 1) { asm volatile(REP4("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc)); }
        else if (KIND == 12) { int s0, s1; asm volatile(REP4("v_readfirstlane_b32 %0, %2\n v_readfirstlane_b32 %1, %3\n v_readfirstlane_b32 %0, %2\n v_readfirstlane_b32 %1, %3\n v_readfirstlane_b32 %0, %2\n v_readfirstlane_b32 %1, %3\n v_readfirstlane_b32 %0, %2\n v_readfirstlane_b32 %1, %3\n") : "=s"(s0), "=s"(s1) : "v"(a0), "v"(a1)); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
        rt[blockIdx.x * 4 + threadIdx.x / 64] = r1 - r0;
    }
}

template <int KIND>
void run(const char *name, int blocks, int iters) {
    float *out; unsigned long long *cyc, *rt;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&cyc, blocks * 4 * 8); hipMalloc(&rt, blocks * 4 * 8);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, rt, iters);  // warm
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, rt, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks * 4), r(blocks * 4);
    hipMemcpy(c.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), rt, blocks * 4 * 8, hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
    double med_c = (double) c[c.size() / 2], med_r = (double) r[r.size() / 2];
    const double instr_per_wave = 32.0 * iters;
    // 8 waves share a SIMD (blocks = 8 per CU * 256 CUs, 4 waves per block -> 32 waves/CU)
    const double cycles_per_instr_per_simd = med_c / (instr_per_wave * 8.0);
    const double clk_mhz = med_c / med_r * 100.0;
    printf("%-28s wall %.3f ms  wave cycles %.0f  shader clock %.0f MHz  -> %.2f cycles per wave64 instr per SIMD\n",
           name, ms, med_c, clk_mhz, cycles_per_instr_per_simd);
    hipFree(out); hipFree(cyc); hipFree(rt);
}

int main() {
    const int blocks = 256 * 8, iters = 2000;
    run<0>("v_fma_f32", blocks, iters);
    run<1>("v_pk_fma_f32", blocks, iters);
    run<2>("v_cvt_f32_i32_sdwa", blocks, iters);
    run<3>("v_rcp_f32", blocks, iters);
    run<4>("v_mov_b32_dpp wave_shl", blocks, iters);
    run<5>("v_cndmask_b32", blocks, iters);
    run<6>("v_bfi_b32", blocks, iters);
    run<7>("v_mul_f32", blocks, iters);
    run<8>("v_max3_f32 |a|,|b|,c", blocks, iters);
    run<9>("v_fmaak_f32", blocks, iters);
    run<10>("v_cmp_gt_f32 e64", blocks, iters);
    run<11>("v_bitop3_b32", blocks, iters);
    run<12>("v_readfirstlane_b32", blocks, iters);
    run<20>("pair: v_cmp vcc + v_cndmask vcc", blocks, iters);
    run<21>("pair: v_cmp sgpr + v_cndmask_e64", blocks, iters);
    run<22>("ashr+and+sub (x2.67 groups)", blocks, iters);
    run<23>("2 cndmask vcc + 6 fma", blocks, iters);
    run<24>("v_sub_f32_e64 |a|,|b|", blocks, iters);
    run<25>("v_fma_f32 (half with |abs|)", blocks, iters);
    run<30>("v_cvt_f32_ubyte0-3", blocks, iters);
    run<31>("v_mul_legacy_f32 abs", blocks, iters);
    run<32>("v_perm_b32", blocks, iters);
    run<33>("v_cvt_f32_i32", blocks, iters);
    run<34>("v_bfe_i32", blocks, iters);
    run<35>("1 rcp + 7 fma", blocks, iters);
    run<36>("2 sdwa cvt + 6 fma", blocks, iters);
    run<37>("2 ubyte cvt + 6 fma", blocks, iters);
    run<13>("v_cndmask_b32_e64 sgpr mask", blocks, iters);
    run<14>("v_sub_f32 literal", blocks, iters);
    run<15>("v_min_f32_e64 abs", blocks, iters);
    run<16>("v_cmp_gt_i32 vcc", blocks, iters);
    run<17>("v_add_u32 literal", blocks, iters);
    return 0;
}
