// scatter_read.hip — what the selection units' first phase pays for the layout of the candidate area: 1280 workgroups of 256 threads
// start together (five per CU) and every thread reads 8 words, 10 consecutive words per "cell segment"; the segments lie
// `pitch` bytes apart (1376 = one 640x480 cell's worst-case segment, 40 = packed) inside an area another kernel has just written
// (as k_fast_wave does).  Prints the kernel time and the mean / max cycles a workgroup waits for its loads.
// build + run: hipcc -O3 --offload-arch=gfx950 tools/microbench/scatter_read.hip -o /tmp/scatter_read && /tmp/scatter_read
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int WG = 1280, T = 256, PER = 8, SEG = 10;

__global__ __launch_bounds__(256) void writer(uint32_t* a, long long pitchW, long long unitW) {
  // unit u owns segments [u * segsPerUnit ..): writes the 10 live words of each
  const int segsPerUnit = (T * PER + SEG - 1) / SEG;
  for (int s = threadIdx.x; s < segsPerUnit; s += T)
    for (int k = 0; k < SEG; k++) a[blockIdx.x * unitW + s * pitchW + k] = (uint32_t)(s * 16 + k);
}

__global__ __launch_bounds__(256) void reader(const uint32_t* __restrict__ a, long long pitchW, long long unitW, uint32_t* out,
                                              unsigned long long* cyc, int perm) {
  // unit = a permutation of the block index, so that a unit is read on another XCD than the one that wrote it
  const int u = (int)(((long long)blockIdx.x * perm) % WG);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  uint32_t v[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int i = threadIdx.x + j * T;  // candidate i of the unit: segment i / 10, word i % 10
    v[j] = a[u * unitW + (i / SEG) * pitchW + (i % SEG)];
  }
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) acc += v[j];
  __syncthreads();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (acc == 0xdeadbeefu) out[0] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int pitches[] = {1376, 512, 128, 40};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  uint32_t* out;
  unsigned long long* cyc;
  hipMalloc(&out, 4);
  hipMalloc(&cyc, WG * 8);
  for (int pitch : pitches) {
    const long long pitchW = pitch / 4, segs = (T * PER + SEG - 1) / SEG, unitW = segs * pitchW + 64;
    uint32_t* a;
    if (hipMalloc(&a, (size_t)WG * unitW * 4) != hipSuccess) return 1;
    float best = 1e9f;
    std::vector<unsigned long long> h(WG);
    double mean = 0, mx = 0;
    for (int rep = 0; rep < 5; rep++) {
      hipLaunchKernelGGL(writer, dim3(WG), dim3(T), 0, 0, a, pitchW, unitW);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(reader, dim3(WG), dim3(T), 0, 0, a, pitchW, unitW, out, cyc, 171);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) {
        best = ms;
        hipMemcpy(h.data(), cyc, WG * 8, hipMemcpyDeviceToHost);
        mean = 0; mx = 0;
        for (auto c : h) { mean += (double)c / WG; mx = std::max(mx, (double)c); }
      }
    }
    printf("pitch %5d B  area %7.1f MB  reader %7.1f us  load wait per workgroup: mean %7.0f  max %7.0f cycles\n", pitch,
           WG * unitW * 4 / 1e6, best * 1e3, mean, mx);
    hipFree(a);
  }
  return 0;
}
