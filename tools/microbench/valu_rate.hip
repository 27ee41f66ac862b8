// valu_rate.hip — issue rate of wave64 vector instructions on one SIMD of gfx950 (MI355X), measured: cycles per instruction
// per SIMD with n = 1, 2, 4, 8 waves per SIMD, each wave issuing eight independent chains.  The n = 8 column is the
// throughput that bounds a VALU-bound kernel; it decides the denominator of the VALU roofline in bench.py / DESIGN.md.
// build + run: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/microbench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define REP 64
#define ITERS 1000

// eight copies of one instruction on eight different destination registers
#define OP8_2(ins) asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2))
#define OP8_3(ins) asm volatile(ins " %0, %0, %8, %9\n" ins " %1, %1, %8, %9\n" ins " %2, %2, %8, %9\n" ins " %3, %3, %8, %9\n" ins " %4, %4, %8, %9\n" ins " %5, %5, %8, %9\n" ins " %6, %6, %8, %9\n" ins " %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2))
#define OP8_1(ins) asm volatile(ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3\n" ins " %4, %4\n" ins " %5, %5\n" ins " %6, %6\n" ins " %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2))
// compares write an SGPR pair each
#define OP8_C(ins) asm volatile(ins " s[20:21], %0, %8\n" ins " s[22:23], %1, %8\n" ins " s[24:25], %2, %8\n" ins " s[26:27], %3, %8\n" ins " s[28:29], %4, %8\n" ins " s[30:31], %5, %8\n" ins " s[32:33], %6, %8\n" ins " s[34:35], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35")
#define OP8_CND() asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2) : "s20", "s21")
#define OP8_MB() asm volatile("v_mbcnt_lo_u32_b32 %0, %8, %0\n v_mbcnt_lo_u32_b32 %1, %8, %1\n v_mbcnt_lo_u32_b32 %2, %8, %2\n v_mbcnt_lo_u32_b32 %3, %8, %3\n v_mbcnt_lo_u32_b32 %4, %8, %4\n v_mbcnt_lo_u32_b32 %5, %8, %5\n v_mbcnt_lo_u32_b32 %6, %8, %6\n v_mbcnt_lo_u32_b32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2))
#define OP8_SDWA() asm volatile("v_sub_u32_sdwa %0, %0, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u32_sdwa %1, %1, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u32_sdwa %2, %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u32_sdwa %3, %3, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u32_sdwa %4, %4, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u32_sdwa %5, %5, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u32_sdwa %6, %6, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2))

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, unsigned long long* cyc, int iters) {
  extern __shared__ uint32_t dummyLds[];
  if (iters < 0) dummyLds[threadIdx.x] = 1;
  uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 + 11, a6 = a0 ^ 0x33, a7 = a0 * 7 + 3;
  const uint32_t k1 = out[0], k2 = out[1];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
      if (OP == 0) OP8_2("v_add_u32");
      if (OP == 1) OP8_3("v_min3_i32");
      if (OP == 2) OP8_3("v_perm_b32");
      if (OP == 3) OP8_2("v_pk_max_u16");
      if (OP == 4) OP8_3("v_fma_f32");
      if (OP == 5) OP8_3("v_dot4_u32_u8");
      if (OP == 6) OP8_2("v_mul_lo_u32");
      if (OP == 7) OP8_2("v_min_i32");
      if (OP == 8) OP8_2("v_max_u32");
      if (OP == 9) OP8_2("v_and_b32");
      if (OP == 10) OP8_2("v_lshrrev_b32");
      if (OP == 11) OP8_3("v_bfe_u32");
      if (OP == 12) OP8_3("v_alignbyte_b32");
      if (OP == 13) OP8_3("v_lshl_add_u32");
      if (OP == 14) OP8_3("v_add3_u32");
      if (OP == 15) OP8_3("v_mad_u32_u24");
      if (OP == 16) OP8_2("v_mul_u32_u24");
      if (OP == 17) OP8_C("v_cmp_gt_i32");
      if (OP == 18) OP8_CND();
      if (OP == 19) OP8_MB();
      if (OP == 20) OP8_SDWA();
      if (OP == 21) OP8_2("v_pk_add_u16");
      if (OP == 22) OP8_2("v_pk_sub_i16");
      if (OP == 23) OP8_3("v_dot2_u32_u16");
      if (OP == 24) OP8_3("v_or3_b32");
      if (OP == 25) OP8_3("v_sad_u8");
      if (OP == 26) OP8_3("v_max3_u32");
      if (OP == 27) OP8_3("v_med3_i32");
      if (OP == 28) OP8_1("v_mov_b32");
      if (OP == 29) OP8_1("v_cvt_f32_u32");
      if (OP == 30) OP8_2("v_sub_u32");
      if (OP == 31) OP8_2("v_xor_b32");
      if (OP == 32) OP8_2("v_mul_f32");
      if (OP == 33) OP8_2("v_add_f32");
      if (OP == 34) OP8_1("v_floor_f32");
      if (OP == 35) OP8_1("v_cvt_u32_f32");
      if (OP == 36) OP8_1("v_cvt_f32_ubyte0");
      if (OP == 37) OP8_1("v_cvt_f32_ubyte2");
      if (OP == 38) OP8_2("v_mul_hi_u32_u24");
      if (OP == 39) OP8_3("v_lshl_or_b32");
      if (OP == 40) OP8_3("v_and_or_b32");
      if (OP == 41) OP8_3("v_cvt_pk_u8_f32");
      if (OP == 42) OP8_1("v_rndne_f32");
      if (OP == 43) OP8_2("v_max_f32");
      if (OP == 44) OP8_3("v_bfi_b32");
      if (OP == 45) OP8_3("v_mad_i32_i24");
      if (OP == 46) OP8_2("v_ashrrev_i32");
      if (OP == 47) OP8_2("v_lshlrev_b32");
      if (OP == 48) OP8_2("v_or_b32");
      if (OP == 49) OP8_2("v_bcnt_u32_b32");
      if (OP == 50) OP8_3("v_xad_u32");
      if (OP == 51) OP8_3("v_add_lshl_u32");
      if (OP == 52) OP8_2("v_mul_hi_u32");
      if (OP == 53) OP8_1("v_trunc_f32");
      if (OP == 54) OP8_1("v_cvt_i32_f32");
      if (OP == 55) OP8_1("v_rcp_f32");
      if (OP == 56) OP8_2("v_pk_mul_lo_u16");
      if (OP == 57) OP8_3("v_pk_mad_u16");
      if (OP == 58) OP8_2("v_pk_lshrrev_b16");
      if (OP == 59) OP8_1("v_cvt_f32_i32");
      if (OP == 60) OP8_2("v_ldexp_f32");
      if (OP == 61) OP8_1("v_not_b32");
      if (OP == 62) OP8_1("v_fract_f32");
      if (OP == 63) OP8_2("v_sub_f32");
      if (OP == 64) OP8_2("v_min_f32");
      if (OP == 65) OP8_2("v_mul_i32_i24");
      if (OP == 66) OP8_3("v_fma_f32");
      if (OP == 100) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2));
      if (OP == 101) asm volatile("v_add_u32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[2 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char* name) {
  uint32_t* out; unsigned long long* cyc;
  const int maxBlocks = 256 * 8 * 4;
  hipMalloc(&out, (2 + (size_t)maxBlocks * 256) * 4); hipMemset(out, 0, 64); hipMalloc(&cyc, (size_t)maxBlocks * 4 * 8);
  printf("%-18s", name);
  for (int n : {1, 2, 4, 8}) {  // co-resident 256-thread workgroups per CU = waves per SIMD, enforced through the LDS size
    const int lds = 160 * 1024 / n - (n == 1 ? 0 : 2048);
    const int blocks = 256 * n * 4;  // four rounds: most waves run in the steady state
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    k<OP><<<blocks, 256, lds>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<OP><<<blocks, 256, lds>>>(out, cyc, ITERS);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double instr = (double)blocks * 4 * ITERS * REP;
    printf("  n=%d %5.2f cyc (%.2f T/s)", n, med / ((double)ITERS * REP * n), instr / (ms * 1e-3) / 1e12);
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  printf("cycles per wave64 instruction per SIMD with n waves per SIMD (chip-wide wave-instructions per second)\n");
  run<0>("v_add_u32"); run<30>("v_sub_u32"); run<9>("v_and_b32"); run<31>("v_xor_b32"); run<10>("v_lshrrev_b32"); run<7>("v_min_i32"); run<8>("v_max_u32");
  run<28>("v_mov_b32"); run<29>("v_cvt_f32_u32"); run<16>("v_mul_u32_u24"); run<4>("v_fma_f32");
  run<17>("v_cmp_gt_i32 (sgpr)"); run<18>("v_cndmask (sgpr)"); run<19>("v_mbcnt_lo"); run<20>("v_sub_u32_sdwa");
  run<1>("v_min3_i32"); run<26>("v_max3_u32"); run<27>("v_med3_i32"); run<2>("v_perm_b32"); run<11>("v_bfe_u32"); run<12>("v_alignbyte_b32");
  run<13>("v_lshl_add_u32"); run<14>("v_add3_u32"); run<24>("v_or3_b32"); run<15>("v_mad_u32_u24"); run<25>("v_sad_u8"); run<6>("v_mul_lo_u32");
  run<3>("v_pk_max_u16"); run<21>("v_pk_add_u16"); run<22>("v_pk_sub_i16"); run<5>("v_dot4_u32_u8"); run<23>("v_dot2_u32_u16");
  if (getenv("VALU_RATE_MORE")) {
    run<32>("v_mul_f32"); run<33>("v_add_f32"); run<34>("v_floor_f32"); run<35>("v_cvt_u32_f32"); run<36>("v_cvt_f32_ubyte0"); run<37>("v_cvt_f32_ubyte2");
    run<38>("v_mul_hi_u32_u24"); run<39>("v_lshl_or_b32"); run<40>("v_and_or_b32"); run<41>("v_cvt_pk_u8_f32"); run<42>("v_rndne_f32"); run<43>("v_max_f32");
    run<44>("v_bfi_b32"); run<45>("v_mad_i32_i24"); run<46>("v_ashrrev_i32"); run<47>("v_lshlrev_b32"); run<48>("v_or_b32"); run<49>("v_bcnt_u32_b32");
    run<50>("v_xad_u32"); run<51>("v_add_lshl_u32"); run<52>("v_mul_hi_u32"); run<53>("v_trunc_f32"); run<54>("v_cvt_i32_f32"); run<55>("v_rcp_f32");
    run<56>("v_pk_mul_lo_u16"); run<57>("v_pk_mad_u16"); run<58>("v_pk_lshrrev_b16"); run<59>("v_cvt_f32_i32"); run<60>("v_ldexp_f32"); run<61>("v_not_b32");
    run<62>("v_fract_f32"); run<63>("v_sub_f32"); run<64>("v_min_f32"); run<65>("v_mul_i32_i24"); run<66>("v_fma_f32 (again)"); run<100>("v_mov_b32_dpp row_shr");
    run<101>("v_add_u32_dpp row_shr");
  }
  return 0;
}
