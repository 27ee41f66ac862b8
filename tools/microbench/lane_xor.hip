// the value of lane (l ^ 2^k) by DPP selects / v_permlane16_swap / v_permlane32_swap (bitonic sorts of the selection kernels): check on the device
// build + run: hipcc -O3 --offload-arch=gfx950 -o /tmp/lane_xor tools/microbench/lane_xor.hip && /tmp/lane_xor
#include <hip/hip_runtime.h>
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
template <int LM> __device__ __forceinline__ unsigned laneXorT(unsigned v) {
  if constexpr (LM == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
  else if constexpr (LM == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
  else if constexpr (LM == 4) {
    unsigned r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);
    return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false);
  } else if constexpr (LM == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);
  else if constexpr (LM == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (__lane_id() & 16) ? r[0] : r[1];
  } else {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (__lane_id() & 32) ? r[0] : r[1];
  }
}
__global__ void k(unsigned* o) {
  unsigned v = threadIdx.x * 3 + 1;
  o[threadIdx.x] = laneXorT<1>(v); o[64 + threadIdx.x] = laneXorT<2>(v); o[128 + threadIdx.x] = laneXorT<4>(v);
  o[192 + threadIdx.x] = laneXorT<8>(v); o[256 + threadIdx.x] = laneXorT<16>(v); o[320 + threadIdx.x] = laneXorT<32>(v);
}
int main() {
  unsigned* d; hipMalloc(&d, 384 * 4); k<<<1, 64>>>(d); unsigned h[384]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0; int lm[6] = {1, 2, 4, 8, 16, 32};
  for (int t = 0; t < 6; t++) for (int i = 0; i < 64; i++) if (h[t * 64 + i] != (unsigned)((i ^ lm[t]) * 3 + 1)) bad++;
  printf("bad %d\n", bad); return bad != 0;
}
