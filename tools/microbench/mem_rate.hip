// mem_rate.hip — cost of one wave64 memory instruction on gfx950 for the access shapes the pyramid kernel can choose from:
// per-lane 8-byte windows at a 4.8-byte pitch (the taps of 4 output pixels at scale 1.2), read from global memory (L1/L2 hits)
// or from LDS, byte-aligned / 4-byte aligned / 8-byte aligned / fully coalesced.  Prints cycles per instruction per CU with
// 8 waves per SIMD issuing back to back (throughput, not latency).
// build + run: hipcc -O3 --offload-arch=gfx950 tools/microbench/mem_rate.hip -o /tmp/mem_rate && /tmp/mem_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 512

__device__ __forceinline__ uint2 ldg8(const uint8_t* p) { uint2 v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint4 ldg16(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }

// MODE: 0 global 8 B byte-aligned pitch 4.8   1 global 8 B at (x & ~3)   2 global 8 B at (x & ~7)   3 global 8 B coalesced (lane * 8)
//       4 global 16 B byte-aligned pitch 9.6  5 global 16 B coalesced    6 global 3 x dword at (x & ~3) + 0, 4, 8
//       10 LDS 8 B byte-aligned pitch 4.8     11 LDS 8 B at (x & ~3)      12 LDS 8 B coalesced        13 LDS 3 x b32 at (x & ~3)
//       14 LDS 16 B byte-aligned pitch 9.6
template <int MODE>
__global__ __launch_bounds__(256) void k(const uint8_t* __restrict__ src, uint32_t* out, unsigned long long* cyc, int rowStride, int iters) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[16384];
  for (int i = threadIdx.x; i < 16384 / 4; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = reinterpret_cast<const uint32_t*>(src)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x48 = (lane * 24) / 5 + 1, x96 = (lane * 48) / 5 + 1;
  uint32_t acc = 0;
  const uint8_t* base = src + (size_t)blockIdx.x * 4096 + wave * rowStride;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    const int row = (it & 15) * 640;   // stay inside 16 KB so that the loads hit in L1 / LDS
    if (MODE == 0) { uint2 v = ldg8(base + row + x48); acc += v.x ^ v.y; }
    if (MODE == 1) { uint2 v = ldg8(base + row + (x48 & ~3)); acc += v.x ^ v.y; }
    if (MODE == 2) { uint2 v = ldg8(base + row + (x48 & ~7)); acc += v.x ^ v.y; }
    if (MODE == 3) { uint2 v = ldg8(base + row + lane * 8); acc += v.x ^ v.y; }
    if (MODE == 4) { uint4 v = ldg16(base + row + x96); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (MODE == 5) { uint4 v = ldg16(base + row + lane * 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (MODE == 6) { const uint32_t* p = reinterpret_cast<const uint32_t*>(base + row + (x48 & ~3)); acc += p[0] ^ p[1] ^ p[2]; }
    if (MODE == 10) { uint2 v; __builtin_memcpy(&v, lds + row + x48, 8); acc += v.x ^ v.y; }
    if (MODE == 11) { uint2 v; __builtin_memcpy(&v, lds + row + (x48 & ~3), 8); acc += v.x ^ v.y; }
    if (MODE == 12) { uint2 v; __builtin_memcpy(&v, lds + row + lane * 8, 8); acc += v.x ^ v.y; }
    if (MODE == 13) { const uint32_t* p = reinterpret_cast<const uint32_t*>(lds + row + (x48 & ~3)); acc += p[0] ^ p[1] ^ p[2]; }
    if (MODE == 14) { uint4 v; __builtin_memcpy(&v, lds + row + x96, 16); acc += v.x ^ v.y ^ v.z ^ v.w; }
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int instrPerIter) {
  uint8_t* src; uint32_t* out; unsigned long long* cyc;
  const int blocks = 256 * 8 * 2;  // 8 workgroups of 4 waves per CU resident, two rounds
  hipMalloc(&src, (size_t)blocks * 4096 + (1 << 20)); hipMemset(src, 7, (size_t)blocks * 4096 + (1 << 20));
  hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
  k<MODE><<<blocks, 256>>>(src, out, cyc, 2560, 16);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(src, out, cyc, 2560, ITERS);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  // chip-wide: blocks * 4 waves * ITERS iterations in ms -> cycles per wave-iteration per CU at 2.4 GHz
  const double waveIters = (double)blocks * 4 * ITERS;
  const double cycPerCU = ms * 1e-3 * 2.4e9 / (waveIters / 256.0);
  printf("%-46s %6.1f cycles per wave-iteration per CU  (%d instr: %5.1f each)   %.3f ms\n", name, cycPerCU, instrPerIter, cycPerCU / instrPerIter, ms);
  hipFree(src); hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("global 8 B, byte-aligned, pitch 4.8", 1);
  run<1>("global 8 B, 4-aligned, pitch 4.8", 1);
  run<2>("global 8 B, 8-aligned, pitch 4.8", 1);
  run<3>("global 8 B, coalesced", 1);
  run<4>("global 16 B, byte-aligned, pitch 9.6", 1);
  run<5>("global 16 B, coalesced", 1);
  run<6>("global 3 x dword, 4-aligned, pitch 4.8", 3);
  run<10>("LDS 8 B, byte-aligned, pitch 4.8", 1);
  run<11>("LDS 8 B, 4-aligned, pitch 4.8", 1);
  run<12>("LDS 8 B, coalesced", 1);
  run<13>("LDS 3 x b32, 4-aligned, pitch 4.8", 3);
  run<14>("LDS 16 B, byte-aligned, pitch 9.6", 1);
  return 0;
}
