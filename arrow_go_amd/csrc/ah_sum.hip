// ah_sum.hip — arrow/math Sum on gfx950.
//
// Replaces: Float64Funcs.Sum / Int64Funcs.Sum / Uint64Funcs.Sum
//   arrow/math/float64.go:34-47, int64.go:34-47, uint64.go:34-47
//   → _sum_float64_avx2 (arrow/math/float64_avx2_amd64.go:33-42; C truth
//     arrow/math/_lib/float64.c:20-26), _sum_int64_avx2, _sum_uint64_avx2.
// Validity is ignored, exactly like the reference (float64.go:41-46).
//
// Roofline: HBM read, 8 algorithmic bytes per row, no reuse → one pass of
// 16 B/lane global_load_dwordx4 (nontemporal: the column is read once and is
// larger than the 256 MiB Infinity Cache at the benchmark size), UNROLL loads in
// flight per lane, grid-stride over ≈8 workgroups per CU.
//
// Float64 numerics: each lane keeps a double-double (s, e) accumulator (plus a second, scaled one for rows ≥ 2^960, ±inf
// and NaN: ah_ddsum.h — the result follows the extended reals) updated
// with Knuth's TwoSum (6 flops/elem ≈ 7 TFLOP/s at 8 TB/s — an order of magnitude
// under the fp64 VALU peak, so it is free under the memory bound).  Lanes, waves
// and workgroups are merged in double-double as well, so the result is the exact
// sum rounded once at the end: independent of grid geometry and within 1 ULP of
// the true sum.  The reference's two paths (sequential vs 32 strided partials)
// differ from EACH OTHER by more than that on general data.
#include "ah_common.h"
#include "ah_ddsum.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;  // 16-byte loads in flight per lane

// (s, e) for the rows below 2^960, (bs, be) — scaled by 2^-128 — for the rest: ah_ddsum.h
struct AccDD {
  ah_ddx a;
  __device__ __forceinline__ void init() { ah_ddx_init(a); }
  __device__ __forceinline__ void add(double x) { ah_ddx_add(a, x); }
  // rows known to be below 2^960 (the caller looked at the whole group's high words)
  __device__ __forceinline__ void add_small(double x) { ah_dd_add(a.s, a.e, x); }
  __device__ __forceinline__ void merge(const AccDD& o) { ah_ddx_merge(a, o.a); }
  __device__ __forceinline__ void wave_reduce() {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ah_ddx t;
      t.s = __shfl_down(a.s, o, 64);
      t.e = __shfl_down(a.e, o, 64);
      t.bs = __shfl_down(a.bs, o, 64);
      t.be = __shfl_down(a.be, o, 64);
      ah_ddx_merge(a, t);
    }
  }
  __device__ __forceinline__ double result() const { return ah_ddx_result(a); }
  static __device__ __forceinline__ unsigned hi_abs(double x) { return ah_dd_hi_abs(x); }
  static constexpr unsigned kBigHi = AH_DDX_BIG_HI;
  static constexpr bool kClassed = true;
};

struct AccU64 {
  uint64_t s;
  __device__ __forceinline__ void init() { s = 0; }
  __device__ __forceinline__ void add(uint64_t x) { s += x; }
  __device__ __forceinline__ void add_small(uint64_t x) { s += x; }
  __device__ __forceinline__ void merge(const AccU64& o) { s += o.s; }
  __device__ __forceinline__ void wave_reduce() {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  }
  __device__ __forceinline__ uint64_t result() const { return s; }
  static __device__ __forceinline__ unsigned hi_abs(uint64_t) { return 0; }
  static constexpr unsigned kBigHi = 1;
  static constexpr bool kClassed = false;   // integers have one class
};

template <typename T>
using Vec2 = T __attribute__((ext_vector_type(2)));  // 16-byte aligned, one global_load_dwordx4

template <typename Acc>
__device__ __forceinline__ void block_reduce_store(Acc acc, Acc* out) {
  __shared__ Acc sm[kBlock / 64];
  acc.wave_reduce();
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sm[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    Acc a = sm[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; w++) a.merge(sm[w]);
    *out = a;
  }
}

// One lane's walk over its share of the body.  CAREFUL = false: every row goes through the unguarded TwoSum (the loop of an
// ordinary column: loads and additions overlap, nothing branches) and the largest high word met is returned; CAREFUL = true: every
// row is classed first (ah_ddx_add).
template <typename T, typename Acc, bool NT, bool CAREFUL>
__device__ __forceinline__ unsigned sum_walk(const Vec2<T>* __restrict__ body, int64_t nvec, Acc& a0, Acc& a1) {
  unsigned top = 0;
  const int64_t stride = (int64_t)gridDim.x * kBlock * kUnroll;
  int64_t i = (int64_t)blockIdx.x * kBlock * kUnroll + threadIdx.x;
  // full iterations: all kUnroll loads issued before any use
  for (; i + (int64_t)(kUnroll - 1) * kBlock < nvec; i += stride) {
    Vec2<T> v[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      if (NT) v[k] = __builtin_nontemporal_load(&body[i + (int64_t)k * kBlock]);
      else v[k] = body[i + (int64_t)k * kBlock];
    }
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      if (CAREFUL) {
        a0.add(v[k].x);
        a1.add(v[k].y);
      } else {
        a0.add_small(v[k].x);
        a1.add_small(v[k].y);
        top = max(top, max(Acc::hi_abs(v[k].x), Acc::hi_abs(v[k].y)));
      }
    }
  }
  // ragged last iteration
#pragma unroll
  for (int k = 0; k < kUnroll; k++) {
    int64_t j = i + (int64_t)k * kBlock;
    if (j < nvec) {
      Vec2<T> v = body[j];
      a0.add(v.x);
      a1.add(v.y);
    }
  }
  return top;
}

// One partial per workgroup. body points at the 16-byte-aligned region (nvec
// 2-element vectors); head/tail (≤ 1 element each) are folded in by block 0.
//
// Float64: the first walk treats every row as an ordinary one and only REMEMBERS the largest high word it met (one AND and
// one MAX per row beside the seven additions).  A wave in which some lane met a row ≥ 2^960, ±inf or NaN throws its sums away
// and walks its share again with every row classed (ah_ddsum.h): an ordinary column never takes a branch inside the loop, a
// column with a few special rows re-reads the shares of the few waves that met them, a column full of them costs two reads.
template <typename T, typename Acc, bool NT>
__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const Vec2<T>* __restrict__ body, int64_t nvec,
                                                               const T* __restrict__ head, int nhead,
                                                               const T* __restrict__ tail, int ntail,
                                                               Acc* __restrict__ partials) {
  Acc a0, a1;
  a0.init();
  a1.init();
  const unsigned top = sum_walk<T, Acc, NT, false>(body, nvec, a0, a1);
  if (Acc::kClassed && __any(top >= Acc::kBigHi)) {   // wave-uniform
    a0.init();
    a1.init();
    (void)sum_walk<T, Acc, NT, true>(body, nvec, a0, a1);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int k = 0; k < nhead; k++) a0.add(head[k]);
    for (int k = 0; k < ntail; k++) a1.add(tail[k]);
  }
  a0.merge(a1);
  block_reduce_store(a0, &partials[blockIdx.x]);
}

template <typename T, typename Acc>
__global__ __launch_bounds__(kBlock) void sum_final_kernel(const Acc* __restrict__ partials, int n, T* __restrict__ out) {
  Acc a;
  a.init();
  for (int i = threadIdx.x; i < n; i += kBlock) a.merge(partials[i]);
  __shared__ Acc res;
  block_reduce_store(a, &res);
  __syncthreads();
  if (threadIdx.x == 0) *out = (T)res.result();
}

// Float64 columns of at most 31 rows: ONE lane, acc = +0.0, acc += x left to right.  Up to 31 rows the reference's AVX2 kernel
// is this very loop (arrow/math/_lib/float64_avx2.s:16-17: cmp rsi, 31 ; jbe → the scalar vaddsd loop at .LBB0_4) and so is the
// pure-Go path (arrow/math/float64.go:41-47): there is exactly one reference answer, intermediate overflow included
// ([1e308, 1e308, -1e308] → +inf), and this kernel returns its bytes.  From 32 rows on the two reference orders differ from each
// other and the order-free accumulator above answers.
constexpr size_t kSeqRows = 31;
__global__ void sum_seq_f64_kernel(const double* __restrict__ buf, int n, double* __restrict__ out) {
  double acc = 0.0;
  for (int i = 0; i < n; i++) acc += buf[i];
  *out = acc;
}

template <typename T, typename Acc>
int sum_dev(ah_ctx* c, const T* buf, size_t len, T* res_dev) {
  if (len == 0) {
    AH_HIP(c, hipMemsetAsync(res_dev, 0, sizeof(T), c->stream));
    return AH_OK;
  }
  if (((uintptr_t)buf & (sizeof(T) - 1)) != 0) return ah_fail(c, AH_EINVALID, "sum: buffer not element-aligned");
  if constexpr (Acc::kClassed) {
    if (len <= kSeqRows) {
      sum_seq_f64_kernel<<<1, 1, 0, c->stream>>>((const double*)buf, (int)len, (double*)res_dev);
      AH_LAUNCH_CHECK(c);
      return AH_OK;
    }
  }
  // peel to 16-byte alignment (≤ 1 element), vector body, ≤ 1 element tail
  int nhead = (int)((((uintptr_t)buf & 15) != 0) ? 1 : 0);
  if ((size_t)nhead > len) nhead = (int)len;
  const T* body = buf + nhead;
  int64_t nvec = (int64_t)((len - nhead) / 2);
  const T* tail = body + nvec * 2;
  int ntail = (int)(len - nhead - (size_t)nvec * 2);
  int64_t iters = ah_ceil_div(nvec, (int64_t)kBlock * kUnroll);
  unsigned grid = ah_stream_grid(c, iters, /*default_bpc=*/2);  // reductions: fewest partials, 7.2 TB/s at 2/CU
  void* scratch;
  int rc = ah_scratch_reserve(c, (size_t)grid * sizeof(Acc), &scratch);
  if (rc != AH_OK) return rc;
  Acc* partials = (Acc*)scratch;
  if (c->tune_nt)
    sum_partials_kernel<T, Acc, true><<<grid, kBlock, 0, c->stream>>>((const Vec2<T>*)body, nvec, buf, nhead, tail, ntail, partials);
  else
    sum_partials_kernel<T, Acc, false><<<grid, kBlock, 0, c->stream>>>((const Vec2<T>*)body, nvec, buf, nhead, tail, ntail, partials);
  AH_LAUNCH_CHECK(c);
  sum_final_kernel<T, Acc><<<1, kBlock, 0, c->stream>>>(partials, (int)grid, res_dev);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

// the partials of one chunk, appended to a caller-owned array (the chunked ingest: every chunk's double-double partials meet in ONE
// final reduction, so a sum that arrives in pieces is rounded once, like a sum over the whole column)
template <typename T, typename Acc>
int sum_chunk(ah_ctx* c, const T* buf, size_t len, Acc* partials, int max_partials, int* n_written) {
  *n_written = 0;
  if (len == 0) return AH_OK;
  if (((uintptr_t)buf & (sizeof(T) - 1)) != 0) return ah_fail(c, AH_EINVALID, "sum: buffer not element-aligned");
  int nhead = (int)((((uintptr_t)buf & 15) != 0) ? 1 : 0);
  if ((size_t)nhead > len) nhead = (int)len;
  const T* body = buf + nhead;
  const int64_t nvec = (int64_t)((len - nhead) / 2);
  const T* tail = body + nvec * 2;
  const int ntail = (int)(len - nhead - (size_t)nvec * 2);
  unsigned grid = ah_stream_grid(c, ah_ceil_div(nvec, (int64_t)kBlock * kUnroll), /*default_bpc=*/2);
  if ((int)grid > max_partials) grid = (unsigned)max_partials;
  if (c->tune_nt)
    sum_partials_kernel<T, Acc, true><<<grid, kBlock, 0, c->stream>>>((const Vec2<T>*)body, nvec, buf, nhead, tail, ntail, partials);
  else
    sum_partials_kernel<T, Acc, false><<<grid, kBlock, 0, c->stream>>>((const Vec2<T>*)body, nvec, buf, nhead, tail, ntail, partials);
  AH_LAUNCH_CHECK(c);
  *n_written = (int)grid;
  return AH_OK;
}

template <typename T, typename Acc>
int sum_host(ah_ctx* c, const T* buf, size_t len, T* res_host) {
  if (!res_host) return ah_fail(c, AH_EINVALID, "sum: null result pointer");
  if (len == 0) { *res_host = 0; return AH_OK; }  // float64.go:35-37
  T* dres = (T*)c->dscalars;
  int rc = sum_dev<T, Acc>(c, buf, len, dres);
  if (rc != AH_OK) return rc;
  static_assert(sizeof(T) == 8, "one mailbox word");
  if ((rc = ah_mailbox_read(c, (const unsigned long long*)dres, 1, (unsigned long long*)c->pinned)) != AH_OK) return rc;
  memcpy(res_host, c->pinned, sizeof(T));
  return AH_OK;
}

}  // namespace

// internal (ah_ingest.hip): ah_sum_partial_bytes(is_f64) bytes per partial
size_t ah_sum_partial_bytes(int is_f64) { return is_f64 ? sizeof(AccDD) : sizeof(AccU64); }
int ah_sum_chunk_partials(ah_ctx* c, int is_f64, const void* buf, size_t len, void* partials, int max_partials, int* n_written) {
  if (is_f64) return sum_chunk<double, AccDD>(c, (const double*)buf, len, (AccDD*)partials, max_partials, n_written);
  return sum_chunk<uint64_t, AccU64>(c, (const uint64_t*)buf, len, (AccU64*)partials, max_partials, n_written);
}
// internal (ah_ingest.hip): a whole Float64 column of ≤ 31 rows that arrived in one chunk — the reference's sequential order
int ah_sum_short_f64(ah_ctx* c, const void* buf, size_t len, void* res_dev) {
  return sum_dev<double, AccDD>(c, (const double*)buf, len, (double*)res_dev);
}
int ah_sum_finish_partials(ah_ctx* c, int is_f64, const void* partials, int n, void* res_dev) {
  if (n <= 0) { AH_HIP(c, hipMemsetAsync(res_dev, 0, 8, c->stream)); return AH_OK; }
  if (is_f64) sum_final_kernel<double, AccDD><<<1, kBlock, 0, c->stream>>>((const AccDD*)partials, n, (double*)res_dev);
  else sum_final_kernel<uint64_t, AccU64><<<1, kBlock, 0, c->stream>>>((const AccU64*)partials, n, (uint64_t*)res_dev);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

AH_EXPORT int ah_sum_float64(ah_ctx* c, const double* buf, size_t len, double* res_host) {
  AH_ENTER(c);
  return sum_host<double, AccDD>(c, buf, len, res_host);
}
AH_EXPORT int ah_sum_int64(ah_ctx* c, const int64_t* buf, size_t len, int64_t* res_host) {
  AH_ENTER(c);
  return sum_host<uint64_t, AccU64>(c, (const uint64_t*)buf, len, (uint64_t*)res_host);
}
AH_EXPORT int ah_sum_uint64(ah_ctx* c, const uint64_t* buf, size_t len, uint64_t* res_host) {
  AH_ENTER(c);
  return sum_host<uint64_t, AccU64>(c, buf, len, res_host);
}
AH_EXPORT int ah_sum_float64_dev(ah_ctx* c, const double* buf, size_t len, double* res_dev) {
  AH_ENTER(c);
  return sum_dev<double, AccDD>(c, buf, len, res_dev);
}
AH_EXPORT int ah_sum_int64_dev(ah_ctx* c, const int64_t* buf, size_t len, int64_t* res_dev) {
  AH_ENTER(c);
  return sum_dev<uint64_t, AccU64>(c, (const uint64_t*)buf, len, (uint64_t*)res_dev);
}
