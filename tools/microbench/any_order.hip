// any_order.hip — does hipExtAnyOrderLaunch let consecutive kernels of ONE stream overlap on gfx950?  (hip_ext.h says the flag is
// "not supported on AMD GFX9xx boards" for hipExtModuleLaunchKernel.)  Three one-workgroup spin kernels of ~200 us each: plain
// launches, launches with the flag, and three streams for comparison.
// build + run: hipcc -O3 --offload-arch=gfx950 tools/microbench/any_order.hip -o /tmp/any_order && /tmp/any_order
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdio>

__global__ void spin(unsigned long long cycles, int* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) {}
  if (out && threadIdx.x == 0) atomicAdd(out, 1);
}

int main() {
  hipStream_t s[3];
  for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
  int* d;
  hipMalloc(&d, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const unsigned long long cyc = 20000;  // 200 us at 100 MHz
  for (int mode = 0; mode < 4; mode++) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipDeviceSynchronize();
      hipEventRecord(e0, s[0]);
      for (int k = 0; k < 3; k++) {
        if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[0], cyc, d);
        if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[0], nullptr, nullptr, k ? hipExtAnyOrderLaunch : 0, cyc, d);
        if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[0], nullptr, nullptr, hipExtAnyOrderLaunch, cyc, d);
        if (mode == 3) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s[k], cyc, d);
      }
      if (mode == 3) { hipStreamSynchronize(s[1]); hipStreamSynchronize(s[2]); }
      hipEventRecord(e1, s[0]);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const char* names[] = {"plain launches, one stream", "flag on kernels 2 and 3, one stream", "flag on all three, one stream", "three streams"};
    printf("%-40s %.3f ms for 3 x 0.2 ms kernels\n", names[mode], best);
  }
  return 0;
}
