// Do the DPP / permlane-swap forms of "value of lane ^ J" agree with ds_bpermute on gfx950?  Prints mismatches per J.
//   hipcc --offload-arch=gfx950 -O3 dpp_xor_test.hip -o /tmp/dpp_xor_test && /tmp/dpp_xor_test
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ unsigned dpp(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }
template <int J>
__device__ __forceinline__ unsigned xor_lane(unsigned v, int lane) {
  if constexpr (J == 1) return dpp<0xB1>(v);
  else if constexpr (J == 2) return dpp<0x4E>(v);
  else if constexpr (J == 4) { const unsigned a = dpp<0x124>(v), b = dpp<0x12C>(v); return (lane & 4) ? a : b; }
  else if constexpr (J == 8) return dpp<0x128>(v);
  else if constexpr (J == 16) { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? r[0] : r[1]; }
  else { auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? r[0] : r[1]; }
}
template <int J>
__device__ void check(unsigned* bad) {
  const int lane = threadIdx.x & 63;
  const unsigned v = lane * 2654435761u + 12345u;
  const unsigned want = __shfl_xor(v, J, 64), got = xor_lane<J>(v, lane);
  if (want != got) atomicAdd(&bad[J == 1 ? 0 : J == 2 ? 1 : J == 4 ? 2 : J == 8 ? 3 : J == 16 ? 4 : 5], 1u);
}
__global__ void k(unsigned* bad) { check<1>(bad); check<2>(bad); check<4>(bad); check<8>(bad); check<16>(bad); check<32>(bad); }
int main() {
  unsigned* d; hipMalloc(&d, 32); hipMemset(d, 0, 32);
  k<<<1, 64>>>(d);
  unsigned h[6]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("mismatching lanes for xor 1 2 4 8 16 32: %u %u %u %u %u %u\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  return 0;
}
