// ah_minmax.hip — min and max of an integer column in one pass (row §8(f)-2).
//
// Replaces utils.GetMinMax{Int8…Uint64} (internal/utils/min_max.go:161-215; AVX2 leaves
// _int64_max_min_avx2 …, internal/utils/_lib/min_max.c:23-126; pure Go :30-148): min starts at the
// type's maximum and max at its minimum, so an EMPTY slice returns (MaxOf, MinOf).  The reference
// uses it for Parquet statistics and dictionary-index validation; validity is not consulted.
// Streaming skeleton of ah_sum.hip: 16 B/lane nontemporal loads, 4 in flight, 2 workgroups per CU,
// per-workgroup partials → one finishing workgroup.  w bytes per row.
#include <limits>
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;

template <typename T>
struct MM {
  T lo, hi;
  __device__ __forceinline__ void init() { lo = std::numeric_limits<T>::max(); hi = std::numeric_limits<T>::min(); }
  __device__ __forceinline__ void add(T v) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
  __device__ __forceinline__ void merge(const MM& o) { lo = o.lo < lo ? o.lo : lo; hi = o.hi > hi ? o.hi : hi; }
};

template <typename T>
__device__ __forceinline__ MM<T> block_reduce(MM<T> a) {
  using W = typename std::conditional<sizeof(T) == 8, long long, int>::type;  // shuffle carrier
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MM<T> b;
    b.lo = (T)__shfl_down((W)a.lo, o, 64);
    b.hi = (T)__shfl_down((W)a.hi, o, 64);
    a.merge(b);
  }
  __shared__ MM<T> sm[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0)
    for (int w = 1; w < kBlock / 64; w++) a.merge(sm[w]);
  return a;  // valid in thread 0
}

template <typename T, bool NT>
__global__ __launch_bounds__(kBlock) void minmax_partials_kernel(const T* __restrict__ values, int64_t n, MM<T>* __restrict__ partials) {
  constexpr int V = 16 / sizeof(T);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  MM<T> a;
  a.init();
  // 16-byte aligned body + element head / tail
  const uintptr_t addr = (uintptr_t)values;
  int64_t head = ((16 - (addr & 15)) & 15) / sizeof(T);
  if (head > n) head = n;
  const int64_t nvec = (n - head) / V;
  const u32x4* body = (const u32x4*)(values + head);
  const int64_t stride = (int64_t)gridDim.x * kBlock * kUnroll;
  for (int64_t i = (int64_t)blockIdx.x * kBlock * kUnroll + threadIdx.x; i < nvec; i += stride) {
    u32x4 v[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      const int64_t j = i + (int64_t)k * kBlock;
      if (j < nvec) v[k] = NT ? __builtin_nontemporal_load(&body[j]) : body[j];
    }
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      if (i + (int64_t)k * kBlock < nvec) {
        const ah_vec16<T> e = __builtin_bit_cast(ah_vec16<T>, v[k]);
#pragma unroll
        for (int j = 0; j < V; j++) a.add(e.v[j]);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int64_t k = 0; k < head; k++) a.add(values[k]);
    for (int64_t k = head + nvec * V; k < n; k++) a.add(values[k]);
  }
  a = block_reduce<T>(a);
  if (threadIdx.x == 0) partials[blockIdx.x] = a;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void minmax_final_kernel(const MM<T>* __restrict__ partials, int np, T* __restrict__ out /*[min, max]*/) {
  MM<T> a;
  a.init();
  for (int i = threadIdx.x; i < np; i += kBlock) a.merge(partials[i]);
  a = block_reduce<T>(a);
  if (threadIdx.x == 0) { out[0] = a.lo; out[1] = a.hi; }
}

template <typename T>
int run_minmax(ah_ctx* c, const void* values, int64_t n, void* out_min_host, void* out_max_host) {
  const unsigned grid = ah_stream_grid(c, ah_ceil_div(ah_ceil_div(n, 16 / sizeof(T)), (int64_t)kBlock * kUnroll), 2);
  void* scratch;
  int rc = ah_scratch_reserve(c, (size_t)grid * sizeof(MM<T>) + 64, &scratch);
  if (rc != AH_OK) return rc;
  MM<T>* partials = (MM<T>*)scratch;
  T* res = (T*)&c->dscalars[12];
  if (c->tune_nt) minmax_partials_kernel<T, true><<<grid, kBlock, 0, c->stream>>>((const T*)values, n, partials);
  else minmax_partials_kernel<T, false><<<grid, kBlock, 0, c->stream>>>((const T*)values, n, partials);
  AH_LAUNCH_CHECK(c);
  minmax_final_kernel<T><<<1, kBlock, 0, c->stream>>>(partials, (int)grid, res);
  AH_LAUNCH_CHECK(c);
  AH_HIP(c, hipMemcpyAsync(c->pinned, res, 2 * sizeof(T), hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(out_min_host, (const void*)c->pinned, sizeof(T));
  memcpy(out_max_host, (const uint8_t*)c->pinned + sizeof(T), sizeof(T));
  return AH_OK;
}

}  // namespace

AH_EXPORT int ah_min_max(ah_ctx* c, int type, const void* values, int64_t n, void* out_min_host, void* out_max_host) {
  AH_ENTER(c);
  if (n < 0) return ah_fail(c, AH_EINVALID, "min_max: negative length");
  if (!out_min_host || !out_max_host) return ah_fail(c, AH_EINVALID, "min_max: null result pointer");
  const int w = ah_type_width(type);
  if (!w || type == AH_FLOAT32 || type == AH_FLOAT64) return ah_fail(c, AH_ENOTIMPL, "min_max: integer types only (got %d)", type);
  if (n > 0 && (!values || ((uintptr_t)values & (uintptr_t)(w - 1)))) return ah_fail(c, AH_EINVALID, "min_max: null or misaligned buffer");
  switch (type) {
#define AH_MM(ID, T)                                                                                              \
  case ID: {                                                                                                      \
    if (n == 0) { T lo = std::numeric_limits<T>::max(), hi = std::numeric_limits<T>::min();                     \
                  memcpy(out_min_host, &lo, sizeof(T)); memcpy(out_max_host, &hi, sizeof(T)); return AH_OK; }     \
    return run_minmax<T>(c, values, n, out_min_host, out_max_host);                                               \
  }
    AH_MM(AH_UINT8, uint8_t) AH_MM(AH_INT8, int8_t) AH_MM(AH_UINT16, uint16_t) AH_MM(AH_INT16, int16_t)
    AH_MM(AH_UINT32, uint32_t) AH_MM(AH_INT32, int32_t) AH_MM(AH_UINT64, uint64_t) AH_MM(AH_INT64, int64_t)
#undef AH_MM
  }
  return ah_fail(c, AH_ENOTIMPL, "min_max: integer types only (got %d)", type);
}
