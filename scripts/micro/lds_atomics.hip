// LDS atomic throughput on gfx950: lane-operations per clock per CU for ds_add_u32 (no return), ds_add_rtn_u32,
// ds_write_b32, ds_read_b32 to pseudo-random addresses over `nbins` dwords, one workgroup of `threads` per slot.
//   hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ void k(int iters, int nbins, unsigned* out) {
  extern __shared__ unsigned s[];
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) s[i] = 0;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u, acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      x = x * 1664525u + 1013904223u;
      const unsigned a = (x >> 10) % (unsigned)nbins;
      if (MODE == 0) atomicAdd(&s[a], 1u);
      else if (MODE == 1) acc += atomicAdd(&s[a], 1u);
      else if (MODE == 2) s[a] = x;
      else acc += s[a];
    }
  }
  __syncthreads();
  if (acc == 0x12345678u || threadIdx.x == 0) out[blockIdx.x] = acc + s[threadIdx.x % nbins];
}
int main() {
  unsigned* out; hipMalloc(&out, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"ds_add_u32", "ds_add_rtn_u32", "ds_write_b32", "ds_read_b32"};
  for (int threads : {256, 1024}) for (int nbins : {256, 1024, 8192}) for (int mode = 0; mode < 4; mode++) {
    const int iters = 2000, blocks = 256 * (2048 / threads);
    auto launch = [&]() {
      if (mode == 0) k<0><<<blocks, threads, nbins * 4>>>(iters, nbins, out);
      if (mode == 1) k<1><<<blocks, threads, nbins * 4>>>(iters, nbins, out);
      if (mode == 2) k<2><<<blocks, threads, nbins * 4>>>(iters, nbins, out);
      if (mode == 3) k<3><<<blocks, threads, nbins * 4>>>(iters, nbins, out);
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * threads * iters * 8;
    printf("%-16s threads %4d bins %5d: %7.3f ms  %6.1f Gops/s  %5.2f lanes/clk/CU (2.4 GHz)\n", names[mode], threads, nbins, ms, ops / ms / 1e6,
           ops / (ms * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
