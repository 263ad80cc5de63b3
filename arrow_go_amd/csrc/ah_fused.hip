// ah_fused.hip — Compare(x OP scalar) → Filter(DropNulls) → Sum in one pass.
//
// NEW entry point (no single reference function): it computes, in one read of the
// column, exactly what the reference computes with three calls —
//   compute.CallFunction("greater", x, t)     (compute/scalar_compare.go:33,
//                                              kernels/scalar_comparisons.go:199-218;
//                                              mask validity = x validity,
//                                              compute/executor.go:237-349)
//   compute.Filter(x, mask, DropNulls)        (kernels/vector_selection.go:267-395)
//   math.Int64.Sum / math.Float64.Sum          (arrow/math/float64.go:34-47)
// i.e. Σ x[i] over { i : valid[i] ∧ x[i] OP t } and the survivor count.  Neither the
// mask nor the filtered column is materialised: 8 (+1/8) algorithmic bytes per row
// instead of 16.25 + 16·s for the unfused chain.
//
// Same streaming skeleton as ah_sum.hip: 16-byte nontemporal loads, kUnroll in
// flight per lane, grid-stride, double-double accumulation for f64, wrapping
// uint64 for i64, two-launch finish ({sum, count} per workgroup → one block).
#include "ah_common.h"
#include "ah_ddsum.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;

template <typename T>
using Vec2 = T __attribute__((ext_vector_type(2)));

template <typename T, int OP>
__device__ __forceinline__ bool pred(T a, T t) {
  if (OP == AH_CMP_EQ) return a == t;
  if (OP == AH_CMP_NE) return a != t;
  if (OP == AH_CMP_GT) return a > t;
  return a >= t;
}

struct PartF64 { ah_ddx a; unsigned long long n; };   // ah_ddsum.h: the sum follows the extended reals, like ah_sum_float64
struct PartI64 { unsigned long long s; unsigned long long n; };

template <typename T> struct Acc;
template <> struct Acc<double> {
  using Part = PartF64;
  ah_ddx a = {0, 0, 0, 0}; unsigned long long n = 0;
  static constexpr bool kClassed = true;
  __device__ __forceinline__ void add(double x) { ah_ddx_add(a, x); n++; }
  // a row taken for an ordinary one; → its high word, sign cleared (the caller remembers the largest)
  __device__ __forceinline__ unsigned add_small(double x) { ah_dd_add(a.s, a.e, x); n++; return ah_dd_hi_abs(x); }
  __device__ __forceinline__ void merge(const Part& p) { ah_ddx_merge(a, p.a); n += p.n; }
  __device__ __forceinline__ void wave_reduce() {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ah_ddx t;
      t.s = __shfl_down(a.s, o, 64);
      t.e = __shfl_down(a.e, o, 64);
      t.bs = __shfl_down(a.bs, o, 64);
      t.be = __shfl_down(a.be, o, 64);
      unsigned long long on = __shfl_down(n, o, 64);
      ah_ddx_merge(a, t);
      n += on;
    }
  }
  __device__ __forceinline__ Part part() const { return Part{a, n}; }
};
template <> struct Acc<int64_t> {
  using Part = PartI64;
  unsigned long long s = 0, n = 0;
  static constexpr bool kClassed = false;
  __device__ __forceinline__ void add(int64_t x) { s += (unsigned long long)x; n++; }
  __device__ __forceinline__ unsigned add_small(int64_t x) { add(x); return 0; }
  __device__ __forceinline__ void merge(const Part& p) { s += p.s; n += p.n; }
  __device__ __forceinline__ void wave_reduce() {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_down(s, o, 64); n += __shfl_down(n, o, 64); }
  }
  __device__ __forceinline__ Part part() const { return Part{s, n}; }
};

template <typename T>
__device__ __forceinline__ void block_reduce(Acc<T>& a, typename Acc<T>::Part* out) {
  __shared__ typename Acc<T>::Part sm[kBlock / 64];
  a.wave_reduce();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a.part();
  __syncthreads();
  if (threadIdx.x == 0) {
    Acc<T> r;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) r.merge(sm[w]);
    *out = r.part();
  }
}

// one workgroup's walk over its iterations of the body.  CAREFUL = false (Float64): every kept row goes through the unguarded
// TwoSum and the largest high word among the kept rows is returned; CAREFUL = true: every kept row is classed (ah_ddsum.h).
template <typename T, int OP, bool HAS_VALID, bool NT, bool CAREFUL>
__device__ __forceinline__ unsigned fused_walk(const Vec2<T>* __restrict__ body, int64_t row0, int64_t nvec,
                                               const uint8_t* __restrict__ valid, int64_t off, T thr, Acc<T>& a) {
  unsigned top = 0;
  const int64_t n_iters = (nvec + (int64_t)kBlock * kUnroll - 1) / ((int64_t)kBlock * kUnroll);
  for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
    const int64_t base = it * kBlock * kUnroll + threadIdx.x;
    const bool full = (it + 1) * (int64_t)kBlock * kUnroll <= nvec;
    Vec2<T> v[kUnroll];
    if (full) {
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        if (NT) v[k] = __builtin_nontemporal_load(&body[base + (int64_t)k * kBlock]);
        else v[k] = body[base + (int64_t)k * kBlock];
      }
    }
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      const int64_t j = base + (int64_t)k * kBlock;
      if (!full) {
        if (j >= nvec) continue;
        v[k] = body[j];
      }
      unsigned vb = 3;
      if (HAS_VALID) {
        const int64_t bit = off + row0 + 2 * j;  // two consecutive validity bits
        vb = (unsigned)((valid[bit >> 3] >> (bit & 7)) & 1) | ((unsigned)((valid[(bit + 1) >> 3] >> ((bit + 1) & 7)) & 1) << 1);
      }
      if ((vb & 1) && pred<T, OP>(v[k].x, thr)) { if (CAREFUL) a.add(v[k].x); else top = max(top, a.add_small(v[k].x)); }
      if ((vb & 2) && pred<T, OP>(v[k].y, thr)) { if (CAREFUL) a.add(v[k].y); else top = max(top, a.add_small(v[k].y)); }
    }
  }
  return top;
}

// body = 16-byte aligned region of nvec 2-element vectors starting at row `row0`.  Float64: like sum_partials_kernel
// (ah_sum.hip) the first walk only remembers whether a kept row was ≥ 2^960, ±inf or NaN; a wave that met one walks again
// with every kept row classed.
template <typename T, int OP, bool HAS_VALID, bool NT>
__global__ __launch_bounds__(kBlock) void fused_kernel(const T* __restrict__ x, int64_t n, int64_t row0, int64_t nvec,
                                                        const uint8_t* __restrict__ valid, int64_t off, T thr,
                                                        typename Acc<T>::Part* __restrict__ partials) {
  Acc<T> a;
  const Vec2<T>* body = (const Vec2<T>*)(x + row0);
  const unsigned top = fused_walk<T, OP, HAS_VALID, NT, !Acc<T>::kClassed>(body, row0, nvec, valid, off, thr, a);
  if (Acc<T>::kClassed && __any(top >= AH_DDX_BIG_HI)) {   // wave-uniform
    a = Acc<T>();
    (void)fused_walk<T, OP, HAS_VALID, NT, true>(body, row0, nvec, valid, off, thr, a);
  }
  // unaligned head (< row0 rows) and odd tail — block 0, lane 0
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int64_t i = 0; i < row0; i++)
      if ((!HAS_VALID || ah_bit(valid, off + i)) && pred<T, OP>(x[i], thr)) a.add(x[i]);
    for (int64_t i = row0 + 2 * nvec; i < n; i++)
      if ((!HAS_VALID || ah_bit(valid, off + i)) && pred<T, OP>(x[i], thr)) a.add(x[i]);
  }
  block_reduce<T>(a, &partials[blockIdx.x]);
}

// out_parts (Float64 only): the un-rounded accumulator, for a caller that merges several of them (ah_comm.hip) and rounds once
template <typename T>
__global__ __launch_bounds__(kBlock) void fused_final_kernel(const typename Acc<T>::Part* __restrict__ partials, int np,
                                                              T* __restrict__ out_sum, int64_t* __restrict__ out_count,
                                                              double* __restrict__ out_parts) {
  Acc<T> a;
  for (int i = threadIdx.x; i < np; i += kBlock) a.merge(partials[i]);
  __shared__ typename Acc<T>::Part res;
  block_reduce<T>(a, &res);
  __syncthreads();
  if (threadIdx.x == 0) {
    if constexpr (__is_floating_point(T)) {
      *out_sum = ah_ddx_result(res.a);
      if (out_parts) { out_parts[0] = res.a.s; out_parts[1] = res.a.e; out_parts[2] = res.a.bs; out_parts[3] = res.a.be; }
    } else {
      *out_sum = (T)res.s;
    }
    *out_count = (int64_t)res.n;
  }
}

template <typename T>
int fused_dev(ah_ctx* c, int cmpop, const T* x, const uint8_t* valid, int64_t off, int64_t n, T thr, T* out_sum_dev,
              int64_t* out_count_dev, double* out_parts_dev = nullptr) {
  if (n < 0 || off < 0) return ah_fail(c, AH_EINVALID, "cmp_filter_sum: negative length/offset");
  if (n == 0) {
    AH_HIP(c, hipMemsetAsync(out_sum_dev, 0, sizeof(T), c->stream));
    AH_HIP(c, hipMemsetAsync(out_count_dev, 0, sizeof(int64_t), c->stream));
    if (out_parts_dev) AH_HIP(c, hipMemsetAsync(out_parts_dev, 0, 4 * sizeof(double), c->stream));
    return AH_OK;
  }
  if (!x) return ah_fail(c, AH_EINVALID, "cmp_filter_sum: null values");
  if ((uintptr_t)x & 7) return ah_fail(c, AH_EINVALID, "cmp_filter_sum: buffer not element-aligned");
  int64_t row0 = ((uintptr_t)x & 15) ? 1 : 0;
  if (row0 > n) row0 = n;
  int64_t nvec = (n - row0) / 2;
  // validity is read with byte loads: more waves in flight pay for the extra latency (0.17 ms at
  // 8/CU vs 0.29 ms at 2/CU with 10 % nulls); without validity the reduction likes few partials
  unsigned grid = ah_stream_grid(c, ah_ceil_div(nvec > 0 ? nvec : 1, (int64_t)kBlock * kUnroll), /*default_bpc=*/valid ? 8 : 2);
  using Part = typename Acc<T>::Part;
  void* scratch;
  int rc = ah_scratch_reserve(c, (size_t)grid * sizeof(Part), &scratch);
  if (rc != AH_OK) return rc;
  Part* partials = (Part*)scratch;
#define AH_FUSED(OPC)                                                                                                   \
  if (valid) {                                                                                                          \
    if (c->tune_nt) fused_kernel<T, OPC, true, true><<<grid, kBlock, 0, c->stream>>>(x, n, row0, nvec, valid, off, thr, partials);   \
    else fused_kernel<T, OPC, true, false><<<grid, kBlock, 0, c->stream>>>(x, n, row0, nvec, valid, off, thr, partials);             \
  } else {                                                                                                              \
    if (c->tune_nt) fused_kernel<T, OPC, false, true><<<grid, kBlock, 0, c->stream>>>(x, n, row0, nvec, valid, off, thr, partials);  \
    else fused_kernel<T, OPC, false, false><<<grid, kBlock, 0, c->stream>>>(x, n, row0, nvec, valid, off, thr, partials);            \
  }
  switch (cmpop) {
    case AH_CMP_EQ: AH_FUSED(AH_CMP_EQ) break;
    case AH_CMP_NE: AH_FUSED(AH_CMP_NE) break;
    case AH_CMP_GT: AH_FUSED(AH_CMP_GT) break;
    case AH_CMP_GE: AH_FUSED(AH_CMP_GE) break;
    default: return ah_fail(c, AH_EINVALID, "cmp_filter_sum: bad op %d", cmpop);
  }
#undef AH_FUSED
  AH_LAUNCH_CHECK(c);
  fused_final_kernel<T><<<1, kBlock, 0, c->stream>>>(partials, (int)grid, out_sum_dev, out_count_dev, out_parts_dev);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <typename T>
int fused_host(ah_ctx* c, int cmpop, const T* x, const uint8_t* valid, int64_t off, int64_t n, T thr, T* out_sum_host,
               int64_t* out_count_host) {
  T* dsum = (T*)&c->dscalars[8];
  int64_t* dcnt = (int64_t*)&c->dscalars[9];
  int rc = fused_dev<T>(c, cmpop, x, valid, off, n, thr, dsum, dcnt);
  if (rc != AH_OK) return rc;
  { int mrc = ah_mailbox_read(c, (const unsigned long long*)dsum, 2, (unsigned long long*)c->pinned); if (mrc != AH_OK) return mrc; }
  if (out_sum_host) memcpy(out_sum_host, (const void*)&c->pinned[0], sizeof(T));
  if (out_count_host) memcpy(out_count_host, (const void*)&c->pinned[1], sizeof(int64_t));
  return AH_OK;
}

}  // namespace

AH_EXPORT int ah_cmp_filter_sum_i64(ah_ctx* c, int cmpop, const int64_t* x, const uint8_t* valid, int64_t off, int64_t n,
                                    int64_t threshold, int64_t* out_sum_host, int64_t* out_count_host) {
  AH_ENTER(c);
  return fused_host<int64_t>(c, cmpop, x, valid, off, n, threshold, out_sum_host, out_count_host);
}
AH_EXPORT int ah_cmp_filter_sum_f64(ah_ctx* c, int cmpop, const double* x, const uint8_t* valid, int64_t off, int64_t n,
                                    double threshold, double* out_sum_host, int64_t* out_count_host) {
  AH_ENTER(c);
  return fused_host<double>(c, cmpop, x, valid, off, n, threshold, out_sum_host, out_count_host);
}
AH_EXPORT int ah_cmp_filter_sum_i64_dev(ah_ctx* c, int cmpop, const int64_t* x, const uint8_t* valid, int64_t off,
                                        int64_t n, int64_t threshold, int64_t* out_sum_count_dev) {
  AH_ENTER(c);
  if (!out_sum_count_dev) return ah_fail(c, AH_EINVALID, "cmp_filter_sum: null output");
  return fused_dev<int64_t>(c, cmpop, x, valid, off, n, threshold, out_sum_count_dev, out_sum_count_dev + 1);
}
AH_EXPORT int ah_cmp_filter_sum_f64_dev(ah_ctx* c, int cmpop, const double* x, const uint8_t* valid, int64_t off,
                                        int64_t n, double threshold, double* out_sum_dev, int64_t* out_count_dev) {
  AH_ENTER(c);
  if (!out_sum_dev || !out_count_dev) return ah_fail(c, AH_EINVALID, "cmp_filter_sum: null output");
  return fused_dev<double>(c, cmpop, x, valid, off, n, threshold, out_sum_dev, out_count_dev);
}
// internal (ah_comm.hip): the sum, the count and the un-rounded four-word accumulator {s, e, bs, be} of this rank's rows
int ah_fused_f64_parts_dev(ah_ctx* c, int cmpop, const double* x, const uint8_t* valid, int64_t off, int64_t n, double threshold,
                           double* out_sum_dev, int64_t* out_count_dev, double* out_parts_dev) {
  return fused_dev<double>(c, cmpop, x, valid, off, n, threshold, out_sum_dev, out_count_dev, out_parts_dev);
}
