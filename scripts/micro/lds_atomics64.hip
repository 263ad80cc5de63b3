// LDS throughput on gfx950 for the group-by table: lane-operations per clock per CU to pseudo-random slots of an
// `nbins`-entry table in LDS.  Modes: 0 ds_add_u64, 1 ds_add_rtn_u64, 2 ds_read_b64, 3 ds_cmpst_rtn_b64 (fails: slot
// occupied), 4 ds_add_u32, 5 ds_add_rtn_u32, 6 ds_min_u32, 7 "group-by row" = read64 + add_rtn64 + add32 + read32.
//   hipcc --offload-arch=gfx950 -O3 lds_atomics64.hip -o /tmp/lds64 && /tmp/lds64
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(int iters, int nbins, unsigned long long* out) {
  extern __shared__ unsigned long long s[];
  unsigned* s32 = reinterpret_cast<unsigned*>(s + nbins);
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) { s[i] = i; s32[i] = 0; s32[nbins + i] = ~0u; }
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
  unsigned long long acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      x = x * 1664525u + 1013904223u;
      const unsigned a = (x >> 10) % (unsigned)nbins;
      if (MODE == 0) atomicAdd(&s[a], (unsigned long long)x);
      else if (MODE == 1) acc += atomicAdd(&s[a], (unsigned long long)x);
      else if (MODE == 2) acc += s[a];
      else if (MODE == 3) acc += atomicCAS(&s[a], ~0ull, (unsigned long long)x);
      else if (MODE == 4) atomicAdd(&s32[a], 1u);
      else if (MODE == 5) acc += atomicAdd(&s32[a], 1u);
      else if (MODE == 6) atomicMin(&s32[nbins + a], x);
      else {
        acc += s[a];
        const unsigned long long old = atomicAdd(&s[a], (unsigned long long)x);
        acc += old + x < old;
        atomicAdd(&s32[a], 1u);
        acc += s32[nbins + a];
      }
    }
  }
  __syncthreads();
  if (acc == 0x12345678u || threadIdx.x == 0) out[blockIdx.x] = acc + s[threadIdx.x % nbins];
}
int main() {
  unsigned long long* out; hipMalloc(&out, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const char* names[] = {"ds_add_u64", "ds_add_rtn_u64", "ds_read_b64", "ds_cmpst_rtn_b64", "ds_add_u32", "ds_add_rtn_u32", "ds_min_u32", "groupby_row(4 ops)"};
  for (int threads : {256, 1024}) for (int nbins : {64, 4096}) for (int mode = 0; mode < 8; mode++) {
    const int iters = 1000, blocks = 256 * (1024 / threads);
    const size_t lds = (size_t)nbins * 16;
    auto launch = [&]() {
      switch (mode) {
        case 0: k<0><<<blocks, threads, lds>>>(iters, nbins, out); break;
        case 1: k<1><<<blocks, threads, lds>>>(iters, nbins, out); break;
        case 2: k<2><<<blocks, threads, lds>>>(iters, nbins, out); break;
        case 3: k<3><<<blocks, threads, lds>>>(iters, nbins, out); break;
        case 4: k<4><<<blocks, threads, lds>>>(iters, nbins, out); break;
        case 5: k<5><<<blocks, threads, lds>>>(iters, nbins, out); break;
        case 6: k<6><<<blocks, threads, lds>>>(iters, nbins, out); break;
        default: k<7><<<blocks, threads, lds>>>(iters, nbins, out); break;
      }
    };
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * threads * iters * 8;
    printf("%-20s threads %4d bins %5d: %8.3f ms  %7.1f G/s  %5.2f per clk per CU (2.4 GHz)\n", names[mode], threads, nbins, ms, ops / ms / 1e6,
           ops / (ms * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
