// Microbenchmark: variants of the streaming Int64 Add (c = a + b, 1 GiB columns) to pick the
// launch geometry / cache policy of arrow_go_amd/csrc/ah_arith.hip.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 add_variants.hip -o /tmp/add_variants && /tmp/add_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned long long u64;
typedef u64 v2 __attribute__((ext_vector_type(2)));

template <int UNROLL, int LD /*0 plain,1 nt*/, int ST>
__global__ void add_gs(const v2* __restrict__ a, const v2* __restrict__ b, v2* __restrict__ c, int64_t nvec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * UNROLL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x; i + (int64_t)(UNROLL - 1) * blockDim.x < nvec; i += stride) {
    v2 x[UNROLL], y[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; k++) {
      x[k] = LD ? __builtin_nontemporal_load(&a[i + (int64_t)k * blockDim.x]) : a[i + (int64_t)k * blockDim.x];
      y[k] = LD ? __builtin_nontemporal_load(&b[i + (int64_t)k * blockDim.x]) : b[i + (int64_t)k * blockDim.x];
    }
#pragma unroll
    for (int k = 0; k < UNROLL; k++) {
      v2 r = x[k] + y[k];
      if (ST) __builtin_nontemporal_store(r, &c[i + (int64_t)k * blockDim.x]); else c[i + (int64_t)k * blockDim.x] = r;
    }
  }
}
// contiguous chunk per block (no grid stride): block b handles vectors [b*CH, (b+1)*CH)
template <int UNROLL, int LD, int ST>
__global__ void add_chunk(const v2* __restrict__ a, const v2* __restrict__ b, v2* __restrict__ c, int64_t nvec, int64_t chunk) {
  int64_t base = (int64_t)blockIdx.x * chunk;
  for (int64_t i = base + threadIdx.x; i + (int64_t)(UNROLL - 1) * blockDim.x < base + chunk; i += (int64_t)blockDim.x * UNROLL) {
    v2 x[UNROLL], y[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; k++) {
      x[k] = LD ? __builtin_nontemporal_load(&a[i + (int64_t)k * blockDim.x]) : a[i + (int64_t)k * blockDim.x];
      y[k] = LD ? __builtin_nontemporal_load(&b[i + (int64_t)k * blockDim.x]) : b[i + (int64_t)k * blockDim.x];
    }
#pragma unroll
    for (int k = 0; k < UNROLL; k++) {
      v2 r = x[k] + y[k];
      if (ST) __builtin_nontemporal_store(r, &c[i + (int64_t)k * blockDim.x]); else c[i + (int64_t)k * blockDim.x] = r;
    }
  }
}
template <typename F> float timeit(F f, int reps = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
  const int64_t rows = 1ll << 27, nvec = rows / 2;
  v2 *a, *b, *c; hipMalloc(&a, rows * 8); hipMalloc(&b, rows * 8); hipMalloc(&c, rows * 8);
  std::vector<u64> h(1 << 22); for (size_t i = 0; i < h.size(); i++) h[i] = i * 0x9E3779B97F4A7C15ull;
  for (int64_t off = 0; off < rows; off += (1 << 22)) { hipMemcpy((u64*)a + off, h.data(), (1 << 22) * 8, hipMemcpyHostToDevice); hipMemcpy((u64*)b + off, h.data(), (1 << 22) * 8, hipMemcpyHostToDevice); }
  auto gbs = [&](float ms) { return 24.0 * rows / ms / 1e6; };
#define GS(U, LD, ST, BLK, GRID) printf("gridstride unroll=%d ld_nt=%d st_nt=%d block=%4d grid=%7d : %7.1f GB/s\n", U, LD, ST, BLK, GRID, gbs(timeit([&] { add_gs<U, LD, ST><<<GRID, BLK>>>(a, b, c, nvec); })));
  for (int grid : {256 * 2, 256 * 4, 256 * 8, 256 * 16, 256 * 32}) { GS(4, 1, 1, 256, grid) }
  for (int grid : {256 * 2, 256 * 4, 256 * 8}) { GS(4, 1, 1, 512, grid) GS(4, 1, 1, 1024, grid) }
  GS(1, 1, 1, 256, 256 * 32) GS(2, 1, 1, 256, 256 * 16) GS(8, 1, 1, 256, 256 * 4) GS(8, 1, 1, 256, 256 * 8)
  GS(4, 1, 0, 256, 256 * 8) GS(4, 0, 1, 256, 256 * 8) GS(4, 0, 0, 256, 256 * 8)
  GS(2, 1, 1, 512, 256 * 8) GS(8, 1, 1, 512, 256 * 4)
  // exact grid: every block one iteration of UNROLL*block vectors
  { int g = (int)(nvec / (256 * 4)); GS(4, 1, 1, 256, g) }
  { int g = (int)(nvec / (256 * 8)); GS(8, 1, 1, 256, g) }
  { int g = (int)(nvec / (1024 * 4)); GS(4, 1, 1, 1024, g) }
#define CH(U, LD, ST, BLK, GRID) printf("chunked    unroll=%d ld_nt=%d st_nt=%d block=%4d grid=%7d : %7.1f GB/s\n", U, LD, ST, BLK, GRID, gbs(timeit([&] { add_chunk<U, LD, ST><<<GRID, BLK>>>(a, b, c, nvec, nvec / GRID); })));
  for (int grid : {256 * 4, 256 * 8, 256 * 16, 256 * 64, 256 * 256}) { CH(4, 1, 1, 256, grid) }
  // plain hipMemcpy D2D for reference (2 B moved per byte copied)
  printf("hipMemcpyDtoD 1 GiB: %7.1f GB/s (read+write)\n", 2.0 * rows * 8 / timeit([&] { hipMemcpyAsync(c, a, rows * 8, hipMemcpyDeviceToDevice, 0); }) / 1e6);
  return 0;
}
