// VALU issue cost in SHADER CYCLES (s_memtime), independent of the clock the chip happens to run at: one workgroup of 4*k waves on
// one CU (k waves per SIMD), every wave executes the same stream of N vector instructions and stamps its own start and end.
// cycles per wave-instruction per SIMD = (last end - first start of the workgroup) / (N * k).
// The spec figure (MI355X_MICROARCH.md: SIMD-32, a wave64 v_fma_f32 occupies the ALU for 2 cycles; one wave alone issues every 4)
// is the floor this measures against; `tools/roofline.py` uses the measured floor of the kernel's own instruction mix.
// build: hipcc -O3 --offload-arch=gfx950 -o valu_issue_cycles valu_issue_cycles.hip ; run: ./valu_issue_cycles [grid]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ void k(float* out, unsigned long long* stamps, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (KIND == 0) {        // 8 independent v_fma_f32 (3 register sources)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 1) { // 8 independent v_add_f32 (2 sources)
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if (KIND == 2) { // the node test's mix per child: 6 cvt_f32_ubyte, 6 fma, max3, min3, 2 max/min, cmp, cndmask-ish (16 instr)
                asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4\n"
                             "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_max3_f32 %5, %0, %1, %2\n v_min3_f32 %6, %1, %2, %3\n v_max_f32 %5, %5, %9\n v_min_f32 %6, %6, %8\n"
                             "v_cmp_le_f32 vcc, %5, %6\n v_cndmask_b32 %7, %7, %4, vcc\n v_lshlrev_b32 %4, 1, %4\n v_or_b32 %7, %7, %4\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            } else if (KIND == 3) { // dependent chain of 8 v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(b), "v"(c));
            } else if (KIND == 5) { // 8 v_cvt_f32_ubyteN
                asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n"
                             "v_cvt_f32_ubyte0 %4, %9\n v_cvt_f32_ubyte1 %5, %9\n v_cvt_f32_ubyte2 %6, %9\n v_cvt_f32_ubyte3 %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 6) { // 8 v_max3_f32 / v_min3_f32 (3 register sources)
                asm volatile("v_max3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                             "v_max3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 7) { // 4 x (v_cmp_le_f32 vcc + v_cndmask_b32)
                asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_le_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             "v_cmp_le_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_le_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
            } else if (KIND == 8) { // 8 integer ops: and, lshl, or, lshl_or, and_or, bfe, add, xor
                asm volatile("v_and_b32 %0, %0, %8\n v_lshlrev_b32 %1, 3, %1\n v_or_b32 %2, %2, %8\n v_lshl_or_b32 %3, %3, 3, %8\n"
                             "v_and_or_b32 %4, %4, %8, %9\n v_bfe_u32 %5, %5, 3, 8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 9) { // 8 v_fma_f32 with two SGPR/constant operands (1 register source)
                asm volatile("v_fma_f32 %0, %0, 2.0, 0.5\n v_fma_f32 %1, %1, 2.0, 0.5\n v_fma_f32 %2, %2, 2.0, 0.5\n v_fma_f32 %3, %3, 2.0, 0.5\n"
                             "v_fma_f32 %4, %4, 2.0, 0.5\n v_fma_f32 %5, %5, 2.0, 0.5\n v_fma_f32 %6, %6, 2.0, 0.5\n v_fma_f32 %7, %7, 2.0, 0.5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 10) { // 4 x (v_cmp_le_f32 into an SGPR pair + s_and/or use): compare results kept as scalar masks
                asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n v_cmp_le_f32 s[22:23], %2, %3\n v_cmp_le_f32 s[24:25], %4, %5\n v_cmp_le_f32 s[26:27], %6, %7\n"
                             "v_cmp_gt_f32 s[20:21], %1, %0\n v_cmp_gt_f32 s[22:23], %3, %2\n v_cmp_gt_f32 s[24:25], %5, %4\n v_cmp_gt_f32 s[26:27], %7, %6\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
            } else if (KIND == 11) { // 8 v_max_f32 / v_min_f32 (2 sources)
                asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                             "v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if (KIND == 12) { // 8 v_mul_f32 / v_sub_f32 (the Moller-Trumbore block's bulk)
                asm volatile("v_mul_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if (KIND == 13) { // 8 v_fma_mix_f32: src0 an f16 half of a packed register (converted inside the fma), src1/src2 f32
                asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %2, %8, %9, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %4, %8, %9, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %6, %8, %9, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 14) { // 8 v_cvt_f32_f16
                asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n"
                             "v_cvt_f32_f16 %4, %9\n v_cvt_f32_f16 %5, %9\n v_cvt_f32_f16 %6, %9\n v_cvt_f32_f16 %7, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 15) { // 8 three-operand integer ops: perm, or3, bfi, lshl_add, and_or, lshl_or, mad_u32_u24, add3
                asm volatile("v_perm_b32 %0, %0, %8, %9\n v_or3_b32 %1, %1, %8, %9\n v_bfi_b32 %2, %2, %8, %9\n v_lshl_add_u32 %3, %3, 3, %8\n"
                             "v_and_or_b32 %4, %4, %8, %9\n v_lshl_or_b32 %5, %5, 3, %8\n v_mad_u32_u24 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 16) { // 8 v_cndmask_b32 on an SGPR-pair mask
                asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n"
                             "v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21");
            } else if (KIND == 17) { // 8 v_pk_mul_f32 / v_pk_add_f32 on register pairs (16 flops per 8 instructions... counted as 4 instructions below)
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(const double*)&b));
            } else if (KIND == 18) { // the node test per child with f16 planes: 6 v_fma_mix_f32, max3, min3, max, min, cmp, cndmask, or (13 instr)
                asm volatile("v_fma_mix_f32 %0, %4, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %2, %5, %8, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %5, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_fma_mix_f32 %6, %4, %9, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %5, %9, %8 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
                             "v_max3_f32 %0, %0, %1, %2\n v_min3_f32 %1, %3, %6, %7\n v_max_f32 %0, %0, %9\n v_min_f32 %1, %1, %8\n"
                             "v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %4, vcc\n v_or_b32 %3, %3, %2\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            } else if (KIND == 19) { // 8 v_mul_lo_u32 (the node test has two, the address arithmetic more)
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if (KIND == 20) { // 4 v_mad_u64_u32 (index * stride + 64-bit base: how hipcc forms every global address)
                asm volatile("v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n v_mad_u64_u32 %1, s[20:21], %4, %5, %1\n v_mad_u64_u32 %2, s[20:21], %4, %5, %2\n v_mad_u64_u32 %3, s[20:21], %4, %5, %3\n"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(b), "v"(c) : "s20", "s21");
            } else if (KIND == 21) { // 8 v_lshlrev_b32 with SDWA byte selects on both operands
                asm volatile("v_lshlrev_b32_sdwa %0, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n v_lshlrev_b32_sdwa %1, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                             "v_lshlrev_b32_sdwa %2, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2\n v_lshlrev_b32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n"
                             "v_lshlrev_b32_sdwa %4, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n v_lshlrev_b32_sdwa %5, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                             "v_lshlrev_b32_sdwa %6, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2\n v_lshlrev_b32_sdwa %7, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 22) { // 8 of: bitop3, bfe, lshrrev, variable lshlrev, bcnt, ffbh, mbcnt_lo, mbcnt_hi (the loop's bookkeeping)
                asm volatile("v_bitop3_b32 %0, %0, %8, %9 bitop3:0x6c\n v_bfe_u32 %1, %1, 5, 3\n v_lshrrev_b32 %2, 8, %2\n v_lshlrev_b32 %3, %8, %3\n"
                             "v_bcnt_u32_b32 %4, %4, %8\n v_ffbh_u32 %5, %5\n v_mbcnt_lo_u32_b32 %6, %8, %6\n v_mbcnt_hi_u32_b32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 23) { // 8 v_mul_u32_u24 / v_mad_u32_u24
                asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mad_u32_u24 %1, %1, %8, %9\n v_mul_u32_u24 %2, %2, %8\n v_mad_u32_u24 %3, %3, %8, %9\n"
                             "v_mul_u32_u24 %4, %4, %8\n v_mad_u32_u24 %5, %5, %8, %9\n v_mul_u32_u24 %6, %6, %8\n v_mad_u32_u24 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 24) { // 8 of the division / square-root helpers: rcp, sqrt, div_scale, div_fmas-free fma, div_fixup, rcp, sqrt, rsq
                asm volatile("v_rcp_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_div_scale_f32 %2, vcc, %2, %8, %2\n v_div_fixup_f32 %3, %3, %8, %9\n"
                             "v_rcp_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_rsq_f32 %6, %6\n v_div_fixup_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
            } else if (KIND == 25) { // 8 v_fma_f64 (the pinned sine)
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(const double*)&b), "v"(*(const double*)&b));
            } else if (KIND == 4) { // 8 independent v_mov_b32 (1 source)
                asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                             "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { stamps[2 * w] = t0; stamps[2 * w + 1] = t1; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char* name, int instr_per_group, int grid, float* d_out, unsigned long long* d_st) {
    const int iters = 4000;
    for (int kw : {1, 2, 3, 4}) {                     // waves per SIMD (1024 threads per workgroup at most: 16 waves = 4 per SIMD)
        const int threads = 4 * kw * 64;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, d_out, d_st, 50, 1.0f);
        CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(threads), 0, 0, d_out, d_st, iters, 1.0f);
        CHK(hipDeviceSynchronize());
        const int waves = grid * 4 * kw;
        std::vector<unsigned long long> st(2 * waves);
        CHK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        // per workgroup: span from its first start to its last end
        double worst = 0, mean = 0;
        for (int g = 0; g < grid; ++g) {
            unsigned long long lo = ~0ull, hi = 0;
            for (int w = 0; w < 4 * kw; ++w) { lo = std::min(lo, st[2 * (g * 4 * kw + w)]); hi = std::max(hi, st[2 * (g * 4 * kw + w) + 1]); }
            const double cyc = (double)(hi - lo) / ((double)iters * 4 * instr_per_group * kw);
            worst = std::max(worst, cyc); mean += cyc / grid;
        }
        printf("%-34s grid %4d, %d waves/SIMD: %.3f cycles per wave-instruction per SIMD (worst workgroup %.3f)\n", name, grid, kw, mean, worst);
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs, nominal clock %d kHz; cycles are s_memtime shader cycles\n", p.name, p.multiProcessorCount, p.clockRate);
    float* d_out; unsigned long long* d_st;
    CHK(hipMalloc(&d_out, (size_t)1024 * 1024 * sizeof(float)));
    CHK(hipMalloc(&d_st, (size_t)1024 * 16 * 2 * 8));
    for (int grid : {1, argc > 1 ? atoi(argv[1]) : p.multiProcessorCount}) {
        run<0>("v_fma_f32 x8 independent", 8, grid, d_out, d_st);
        run<1>("v_add_f32 x8 independent", 8, grid, d_out, d_st);
        run<4>("v_mov_b32 x8 independent", 8, grid, d_out, d_st);
        run<2>("node-test mix (16 per child)", 16, grid, d_out, d_st);
        run<3>("v_fma_f32 dependent chain", 8, grid, d_out, d_st);
        if (grid == 1) {
            run<5>("v_cvt_f32_ubyte0..3 x8", 8, grid, d_out, d_st);
            run<6>("v_max3/min3_f32 x8", 8, grid, d_out, d_st);
            run<11>("v_max/min_f32 x8 (2 sources)", 8, grid, d_out, d_st);
            run<12>("v_mul/sub_f32 x8", 8, grid, d_out, d_st);
            run<7>("v_cmp vcc + v_cndmask x4", 8, grid, d_out, d_st);
            run<10>("v_cmp into SGPR pairs x8", 8, grid, d_out, d_st);
            run<8>("integer and/shift/or/bfe/add x8", 8, grid, d_out, d_st);
            run<9>("v_fma_f32 x8, constant operands", 8, grid, d_out, d_st);
            run<13>("v_fma_mix_f32 x8 (f16 src0)", 8, grid, d_out, d_st);
            run<14>("v_cvt_f32_f16 x8", 8, grid, d_out, d_st);
            run<15>("3-operand integer ops x8", 8, grid, d_out, d_st);
            run<16>("v_cndmask_b32 x8 (SGPR mask)", 8, grid, d_out, d_st);
            run<17>("v_pk_mul/add_f32 x4", 4, grid, d_out, d_st);
            run<18>("node test per child, f16 planes (13)", 13, grid, d_out, d_st);
            run<19>("v_mul_lo_u32 x8", 8, grid, d_out, d_st);
            run<20>("v_mad_u64_u32 x4", 4, grid, d_out, d_st);
            run<21>("v_lshlrev_b32_sdwa x8 (byte selects)", 8, grid, d_out, d_st);
            run<22>("bitop3/bfe/shifts/bcnt/ffbh/mbcnt x8", 8, grid, d_out, d_st);
            run<23>("v_mul/mad_u32_u24 x8", 8, grid, d_out, d_st);
            run<24>("rcp/sqrt/rsq/div_scale/div_fixup x8", 8, grid, d_out, d_st);
            run<25>("v_fma_f64 x8", 8, grid, d_out, d_st);
        }
    }
    return 0;
}
