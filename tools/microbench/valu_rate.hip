// valu_rate.hip — issue rate of wave64 vector instructions on one SIMD of gfx950, measured (cycles per instruction per SIMD
// with 1..8 waves per SIMD issuing independent chains).  Decides the denominator of the VALU roofline in bench.py / DESIGN.md.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 64  // unrolled instructions per loop iteration and chain
#define ITERS 2000

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, unsigned long long* cyc, int iters) {
  uint32_t a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 ^ 0x55, a3 = a0 + 7, a4 = a0 * 5, a5 = a0 + 11, a6 = a0 ^ 0x33, a7 = a0 * 7 + 3;
  const uint32_t k1 = out[0], k2 = out[1];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
      // 8 independent chains per wave: no dependency stalls at 4+ cycle latencies
      if (OP == 0) { asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1)); }
      if (OP == 1) { asm volatile("v_min3_i32 %0, %0, %8, %9\n v_min3_i32 %1, %1, %8, %9\n v_min3_i32 %2, %2, %8, %9\n v_min3_i32 %3, %3, %8, %9\n v_min3_i32 %4, %4, %8, %9\n v_min3_i32 %5, %5, %8, %9\n v_min3_i32 %6, %6, %8, %9\n v_min3_i32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2)); }
      if (OP == 2) { asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2)); }
      if (OP == 3) { asm volatile("v_pk_max_u16 %0, %0, %8\n v_pk_max_u16 %1, %1, %8\n v_pk_max_u16 %2, %2, %8\n v_pk_max_u16 %3, %3, %8\n v_pk_max_u16 %4, %4, %8\n v_pk_max_u16 %5, %5, %8\n v_pk_max_u16 %6, %6, %8\n v_pk_max_u16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1)); }
      if (OP == 4) { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2)); }
      if (OP == 5) { asm volatile("v_dot4_u32_u8 %0, %0, %8, %9\n v_dot4_u32_u8 %1, %1, %8, %9\n v_dot4_u32_u8 %2, %2, %8, %9\n v_dot4_u32_u8 %3, %3, %8, %9\n v_dot4_u32_u8 %4, %4, %8, %9\n v_dot4_u32_u8 %5, %5, %8, %9\n v_dot4_u32_u8 %6, %6, %8, %9\n v_dot4_u32_u8 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1), "v"(k2)); }
      if (OP == 6) { asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1)); }
      if (OP == 7) { asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(*(unsigned long long*)&a0), "+v"(*(unsigned long long*)&a2) : "v"(*(unsigned long long*)&a4)); }
      if (OP == 8) { asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k1) : "vcc"); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[2 + blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char* name) {
  uint32_t* out; unsigned long long* cyc;
  hipMalloc(&out, (2 + 1024 * 1024) * 4); hipMemset(out, 0, 64); hipMalloc(&cyc, 8 * 4096);
  printf("%-16s", name);
  for (int wavesPerSimd : {1, 2, 4, 8}) {
    const int threads = 64 * 4 * wavesPerSimd > 1024 ? 1024 : 64 * 4 * wavesPerSimd;  // one workgroup per CU fills every SIMD
    const int blocksPerCu = (64 * 4 * wavesPerSimd) / threads;
    const int blocks = 256 * blocksPerCu;
    k<OP><<<blocks, threads>>>(out, cyc, 10);
    hipDeviceSynchronize();
    k<OP><<<blocks, threads>>>(out, cyc, ITERS);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v; s /= h.size();
    // every wave issued ITERS * REP instructions; a SIMD holds wavesPerSimd of them
    printf("  %dw/SIMD: %.2f cyc/instr/SIMD", wavesPerSimd, s / ((double)ITERS * REP * wavesPerSimd));
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_add_u32");
  run<1>("v_min3_i32");
  run<2>("v_perm_b32");
  run<3>("v_pk_max_u16");
  run<4>("v_fma_f32");
  run<5>("v_dot4_u32_u8");
  run<6>("v_mul_lo_u32");
  run<7>("v_pk_fma_f32");
  run<8>("v_cndmask_b32");
  return 0;
}
