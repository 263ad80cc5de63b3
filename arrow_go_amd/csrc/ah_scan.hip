// ah_scan.hip — cumulative_sum / cumulative_sum_checked: single-pass prefix scan.
//
// Row §8(f)-2 (same registry, same boundary).  Replaces cumulativeSumExec →
//   cumulativeSumSpans → cumulativeSum{NoNulls,WithNulls}[Checked]
//   (arrow/compute/internal/kernels/vector_cumulative.go:228-360) behind compute's
//   "cumulative_sum" / "cumulative_sum_checked" (arrow/compute/vector_cumulative.go:76-96):
//     out[i] = start + Σ_{j ≤ i, j valid} in[j]
//   null input → null output with payload 0; with skip_nulls = false every row from the first
//   null on is null (state.encounteredNull, :270-284); checked: "overflow" as soon as a running
//   sum leaves the type's range (checkedAddSigned/Unsigned :147-160 — the textbook test, unlike
//   the checked Add kernel's carry quirk).
//
// The reference is a sequential loop; here it is ONE pass over HBM (8 B read + 8 B written per
// Int64 row) with the decoupled look-back scan: workgroups take tiles in ticket order, publish
// their tile aggregate, and a wave looks back over predecessors' {aggregate | inclusive prefix}
// records until it meets an inclusive one.  Cross-workgroup hand-off uses self-validating 64-bit
// words {marker | 32 payload bits} moved with relaxed agent-scope atomics — no fences (see the
// note at `Words`).  Ticket order makes it deadlock-free under any dispatch order: a tile only
// waits for tiles whose workgroups are already running.
//
// Accumulator: unchecked ints scan in uint64 (wraparound commutes with truncation); checked ints
// scan EXACTLY in int64 (≤ 32-bit types) or __int128 (64-bit types) so "some running sum left the
// range" is decided exactly; floats scan in double (parallel order — the reference's sequential
// order cannot be reproduced; tolerance in DESIGN.md §4).
#include <limits>
#include <type_traits>
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kVecPerThread = 4;  // 16-byte vectors per lane per tile

typedef __int128 i128;

template <typename A> __device__ __forceinline__ A shfl_up_any(A v, int o) { return __shfl_up(v, o, 64); }
template <> __device__ __forceinline__ i128 shfl_up_any<i128>(i128 v, int o) {
  unsigned long long lo = (unsigned long long)v, hi = (unsigned long long)(v >> 64);
  lo = __shfl_up(lo, o, 64);
  hi = __shfl_up(hi, o, 64);
  return (i128)(((unsigned __int128)hi << 64) | lo);
}
template <typename A> __device__ __forceinline__ A shfl_down_any(A v, int o) { return __shfl_down(v, o, 64); }
template <> __device__ __forceinline__ i128 shfl_down_any<i128>(i128 v, int o) {
  unsigned long long lo = (unsigned long long)v, hi = (unsigned long long)(v >> 64);
  lo = __shfl_down(lo, o, 64);
  hi = __shfl_down(hi, o, 64);
  return (i128)(((unsigned __int128)hi << 64) | lo);
}
template <typename A> __device__ __forceinline__ A shfl_any(A v, int l) { return __shfl(v, l, 64); }
template <> __device__ __forceinline__ i128 shfl_any<i128>(i128 v, int l) {
  unsigned long long lo = (unsigned long long)v, hi = (unsigned long long)(v >> 64);
  lo = __shfl(lo, l, 64);
  hi = __shfl(hi, l, 64);
  return (i128)(((unsigned __int128)hi << 64) | lo);
}

// exclusive scan of one A per thread across the block; *total = block sum.  No subtraction
// anywhere (inclusive − own would turn ±inf into NaN and lose bits for floats).
template <typename A>
__device__ __forceinline__ A block_exclusive_scan(A v, A* total, A* sm /*kBlock/64 + 1 entries*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  A inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    A t = shfl_up_any<A>(inc, o);
    if (lane >= o) inc += t;
  }
  __syncthreads();  // previous use of sm is finished
  if (lane == 63) sm[wave] = inc;
  __syncthreads();
  A base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kBlock / 64; k++) {
    A t = sm[k];
    if (k < wave) base += t;
    tot += t;
  }
  *total = tot;
  A prev = shfl_up_any<A>(inc, 1);
  if (lane == 0) prev = 0;
  return base + prev;
}

// Cross-workgroup records.  Every 64-bit word is self-validating: {marker:32 | payload:32}, written
// and read with relaxed agent-scope atomics (sc1 accesses, coherent across the 8 XCD L2s).  A
// record of sizeof(A) bytes is sizeof(A)/4 such words and is complete when every word carries the
// marker — no release/acquire fence at all.  (The fence form — plain store, agent release fence,
// flag — costs a `buffer_wbl2 sc1` per publish, i.e. a write-back of every dirty line the output
// stream left in that XCD's L2: measured 90 ns per tile, serialised, 6.2 ms for 2^27 Int64 rows.)
template <typename A> struct Words { static constexpr int N = sizeof(A) / 4; };

template <typename A> __device__ __forceinline__ void to_words(A v, unsigned (&w)[Words<A>::N]) {
  if constexpr (sizeof(A) == 8) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    w[0] = (unsigned)u; w[1] = (unsigned)(u >> 32);
  } else {
    unsigned __int128 u = (unsigned __int128)v;
    w[0] = (unsigned)u; w[1] = (unsigned)(u >> 32); w[2] = (unsigned)(u >> 64); w[3] = (unsigned)(u >> 96);
  }
}
template <typename A> __device__ __forceinline__ A from_words(const unsigned (&w)[Words<A>::N]) {
  if constexpr (sizeof(A) == 8) {
    unsigned long long u = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
    return __builtin_bit_cast(A, u);
  } else {
    unsigned __int128 u = (unsigned __int128)w[0] | ((unsigned __int128)w[1] << 32) | ((unsigned __int128)w[2] << 64) | ((unsigned __int128)w[3] << 96);
    return (A)u;
  }
}

constexpr unsigned long long kMarker = 1ull << 32;

template <typename A>
struct ScanState {
  unsigned long long* agg;   // [ntiles][Words<A>::N]  tile aggregate
  unsigned long long* incl;  // [ntiles][Words<A>::N]  inclusive prefix (start included)
  unsigned* ticket;
  unsigned* overflow;
};

template <typename A>
__device__ __forceinline__ void publish(unsigned long long* slot, A value) {
  unsigned w[Words<A>::N];
  to_words<A>(value, w);
#pragma unroll
  for (int k = 0; k < Words<A>::N; k++) __hip_atomic_store(slot + k, kMarker | w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename A>
__device__ __forceinline__ bool try_read(const unsigned long long* slot, A* value) {
  unsigned w[Words<A>::N];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < Words<A>::N; k++) {
    unsigned long long x = __hip_atomic_load(slot + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = ok && (x >> 32) == 1ull;
    w[k] = (unsigned)x;
  }
  *value = from_words<A>(w);
  return ok;
}

// T: element type; A: accumulator; CHECKED: range test of every running sum at a valid row
template <typename T, typename A, bool CHECKED>
__global__ __launch_bounds__(kBlock) void scan_kernel(const T* __restrict__ in, const uint8_t* __restrict__ valid, int64_t off,
                                                       int64_t n, int64_t limit /*rows ≥ limit are null (first null, !skip_nulls)*/,
                                                       A start, T* __restrict__ out, ScanState<A> st) {
  constexpr int V = 16 / sizeof(T);
  constexpr int TILE = kBlock * kVecPerThread * V;
  __shared__ A sm[kBlock / 64 + 1];
  __shared__ unsigned s_tile;
  __shared__ A s_excl;
  if (threadIdx.x == 0) s_tile = atomicAdd(st.ticket, 1u);
  __syncthreads();
  const int64_t tile = s_tile;
  const int64_t base = tile * TILE;
  const int lane = threadIdx.x & 63;

  // ---- load the tile (16-byte vectors, element-aligned), per-lane element sums
  T x[kVecPerThread][V];
  bool ok[kVecPerThread][V];
  A vsum[kVecPerThread];
#pragma unroll
  for (int k = 0; k < kVecPerThread; k++) {
    const int64_t e0 = base + ((int64_t)k * kBlock + threadIdx.x) * V;
    if (e0 + V <= n) {
      ah_vec16<T> v = *(const ah_vec16<T>*)(in + e0);
#pragma unroll
      for (int j = 0; j < V; j++) x[k][j] = v.v[j];
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) x[k][j] = (e0 + j < n) ? in[e0 + j] : (T)0;
    }
    unsigned vb = e0 < n ? (valid ? (unsigned)ah_load_bits64(valid, off + e0, (n - e0) >= V ? V : (int)(n - e0)) : ~0u) : 0u;
    A s = 0;
#pragma unroll
    for (int j = 0; j < V; j++) {
      ok[k][j] = ((vb >> j) & 1) && (e0 + j < limit) && (e0 + j < n);
      s += ok[k][j] ? (A)x[k][j] : (A)0;
    }
    vsum[k] = s;
  }
  // ---- tile-local scan: kVecPerThread block scans chained by a running carry
  A vexcl[kVecPerThread];  // exclusive prefix (within the tile) at the START of this lane's vector k
  A carry = 0;
#pragma unroll
  for (int k = 0; k < kVecPerThread; k++) {
    A tot;
    A exc = block_exclusive_scan<A>(vsum[k], &tot, sm);
    vexcl[k] = carry + exc;
    carry += tot;
  }
  const A tile_total = carry;

  // ---- publish the aggregate, look back for the exclusive prefix
  constexpr int NW = Words<A>::N;
  if (tile == 0) {
    if (threadIdx.x == 0) {
      s_excl = start;
      publish<A>(st.incl, start + tile_total);
    }
  } else {
    if (threadIdx.x == 0) publish<A>(st.agg + tile * NW, tile_total);
    if (threadIdx.x < 64) {
      A excl = 0;
      int64_t look = tile - 1;
      for (;;) {
        const int64_t t = look - lane;
        bool inclusive = true;
        A v = 0;
        if (t >= 0) {
          for (;;) {
            if (try_read<A>(st.incl + t * NW, &v)) break;
            if (try_read<A>(st.agg + t * NW, &v)) { inclusive = false; break; }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        const unsigned long long done = __ballot(inclusive);
        const int first = __ffsll((long long)done) - 1;  // nearest predecessor with an inclusive prefix
        A part = (first < 0 || lane <= first) ? v : (A)0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += shfl_down_any<A>(part, o);
        excl += shfl_any<A>(part, 0);
        if (first >= 0) break;
        look -= 64;
      }
      if (threadIdx.x == 0) {
        s_excl = excl;
        publish<A>(st.incl + tile * NW, excl + tile_total);
      }
    }
  }
  __syncthreads();
  const A excl = s_excl;

  // ---- outputs
  constexpr bool kIsInt = std::is_integral<T>::value;
  bool ovf = false;
#pragma unroll
  for (int k = 0; k < kVecPerThread; k++) {
    const int64_t e0 = base + ((int64_t)k * kBlock + threadIdx.x) * V;
    A run = excl + vexcl[k];  // exclusive prefix at this vector's first element
    ah_vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; j++) {
      if (ok[k][j]) {
        run += (A)x[k][j];
        if (CHECKED && kIsInt) {
          using L = std::numeric_limits<T>;
          if (run > (A)L::max() || run < (A)L::min()) ovf = true;
        }
        o.v[j] = (T)run;
      } else {
        o.v[j] = (T)0;  // null rows keep the zero of the fresh buffer
      }
    }
    if (e0 + V <= n) {
      *(ah_vec16<T>*)(out + e0) = o;
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) if (e0 + j < n) out[e0 + j] = o.v[j];
    }
  }
  if (CHECKED && __any(ovf) && lane == 0) atomicOr(st.overflow, 1u);
}

// first zero bit of a validity bitmap (or n): where encounteredNull turns on
__global__ __launch_bounds__(kBlock) void first_null_kernel(const uint8_t* __restrict__ valid, int64_t off, int64_t n,
                                                             unsigned long long* __restrict__ result) {
  const int64_t nwords = (n + 63) / 64;
  unsigned long long best = ~0ull;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < nwords; w += stride) {
    int cnt = n - w * 64 >= 64 ? 64 : (int)(n - w * 64);
    uint64_t bits = ah_load_bits64(valid, off + w * 64, cnt);
    uint64_t zeros = ~bits & (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1));
    if (zeros) {
      unsigned long long pos = (unsigned long long)(w * 64 + __ffsll((long long)zeros) - 1);
      if (pos < best) best = pos;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long t = __shfl_down(best, o, 64);
    if (t < best) best = t;
  }
  __shared__ unsigned long long sm[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kBlock / 64; k++) if (sm[k] < best) best = sm[k];
    if (best != ~0ull) atomicMin(result, best);
  }
}

template <typename T, typename A, bool CHECKED>
int run_scan(ah_ctx* c, const void* values, const uint8_t* valid, int64_t off, int64_t n, int64_t limit, A start, void* out) {
  constexpr int V = 16 / sizeof(T);
  constexpr int TILE = kBlock * kVecPerThread * V;
  const int64_t ntiles = ah_ceil_div(n, TILE);
  // scratch: agg[ntiles][NW] | incl[ntiles][NW] (64-bit words) ; ticket + overflow in dscalars[12], [13]
  size_t rec_bytes = (((size_t)ntiles * Words<A>::N * 8) + 63) & ~(size_t)63;
  void* scratch;
  int rc = ah_scratch_reserve(c, 2 * rec_bytes + 64, &scratch);
  if (rc != AH_OK) return rc;
  ScanState<A> st;
  st.agg = (unsigned long long*)scratch;
  st.incl = (unsigned long long*)((uint8_t*)scratch + rec_bytes);
  st.ticket = (unsigned*)&c->dscalars[12];
  st.overflow = (unsigned*)&c->dscalars[13];
  AH_HIP(c, hipMemsetAsync(scratch, 0, 2 * rec_bytes, c->stream));
  AH_HIP(c, hipMemsetAsync(&c->dscalars[12], 0, 2 * sizeof(uint64_t), c->stream));
  scan_kernel<T, A, CHECKED><<<(unsigned)ntiles, kBlock, 0, c->stream>>>((const T*)values, valid, off, n, limit, start, (T*)out, st);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <typename T>
int dispatch_scan(ah_ctx* c, const void* values, const uint8_t* valid, int64_t off, int64_t n, int64_t limit, const void* start_host,
                  int checked, void* out) {
  T start = 0;
  if (start_host) memcpy(&start, start_host, sizeof(T));
  if constexpr (std::is_floating_point<T>::value) {
    return run_scan<T, double, false>(c, values, valid, off, n, limit, (double)start, out);
  } else {
    if (!checked) return run_scan<T, unsigned long long, false>(c, values, valid, off, n, limit, (unsigned long long)start, out);
    if constexpr (sizeof(T) == 8) return run_scan<T, i128, true>(c, values, valid, off, n, limit, (i128)start, out);
    else return run_scan<T, long long, true>(c, values, valid, off, n, limit, (long long)start, out);
  }
}

}  // namespace

AH_EXPORT int ah_cumulative_sum(ah_ctx* c, int type, const void* values, const uint8_t* valid, int64_t off, int64_t n,
                                const void* start_host, int skip_nulls, int checked, void* out_values, uint8_t* out_valid,
                                int64_t* out_null_count_host) {
  AH_ENTER(c);
  if (n < 0 || off < 0) return ah_fail(c, AH_EINVALID, "cumulative_sum: negative length/offset");
  if (out_null_count_host) *out_null_count_host = 0;
  if (n == 0) return AH_OK;
  if (!values || !out_values) return ah_fail(c, AH_EINVALID, "cumulative_sum: null buffer");
  if (valid && !out_valid) return ah_fail(c, AH_EINVALID, "cumulative_sum: input has a validity bitmap but no output validity was given");
  int w = ah_type_width(type);
  if (!w) return ah_fail(c, AH_ENOTIMPL, "cumulative sum input type must be numeric");  // vector_cumulative.go:72-75
  if (((uintptr_t)values | (uintptr_t)out_values) & (uintptr_t)(w - 1)) return ah_fail(c, AH_EINVALID, "cumulative_sum: buffer not element-aligned");
  int64_t limit = n;
  if (valid && !skip_nulls) {
    // rows from the first null on are all null (state.encounteredNull, :270-284)
    unsigned long long* res = (unsigned long long*)&c->dscalars[14];
    AH_HIP(c, hipMemsetAsync(res, 0xFF, sizeof(*res), c->stream));
    first_null_kernel<<<ah_stream_grid(c, ah_ceil_div(ah_ceil_div(n, 64), kBlock), 2), kBlock, 0, c->stream>>>(valid, off, n, res);
    AH_LAUNCH_CHECK(c);
    AH_HIP(c, hipMemcpyAsync(c->pinned, res, sizeof(*res), hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipStreamSynchronize(c->stream));
    unsigned long long fn = *(volatile unsigned long long*)c->pinned;
    if (fn != ~0ull) limit = (int64_t)fn;
  }
  int rc;
  switch (type) {
    case AH_UINT8: rc = dispatch_scan<uint8_t>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    case AH_INT8: rc = dispatch_scan<int8_t>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    case AH_UINT16: rc = dispatch_scan<uint16_t>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    case AH_INT16: rc = dispatch_scan<int16_t>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    case AH_UINT32: rc = dispatch_scan<uint32_t>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    case AH_INT32: rc = dispatch_scan<int32_t>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    case AH_UINT64: rc = dispatch_scan<uint64_t>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    case AH_INT64: rc = dispatch_scan<int64_t>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    case AH_FLOAT32: rc = dispatch_scan<float>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    case AH_FLOAT64: rc = dispatch_scan<double>(c, values, valid, off, n, limit, start_host, checked, out_values); break;
    default: return ah_fail(c, AH_ENOTIMPL, "cumulative sum input type must be numeric");
  }
  if (rc != AH_OK) return rc;
  if (out_valid) {
    // validity: skip_nulls → the input's validity; otherwise ones up to the first null, zeros after
    if (valid && skip_nulls) rc = ah_copy_bitmap(c, valid, off, n, out_valid, 0, 0);
    else {
      rc = ah_set_bits_to(c, out_valid, 0, limit, 1);
      if (rc == AH_OK && limit < n) rc = ah_set_bits_to(c, out_valid, limit, n - limit, 0);
    }
    if (rc != AH_OK) return rc;
  }
  bool need_sync = checked || out_null_count_host;
  if (out_null_count_host && out_valid) {
    rc = ah_popcount_async(c, out_valid, 0, n, (unsigned long long*)&c->dscalars[15]);
    if (rc != AH_OK) return rc;
  }
  if (need_sync) {
    AH_HIP(c, hipMemcpyAsync(c->pinned, &c->dscalars[13], 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if (checked && (*(volatile unsigned*)&c->pinned[0] & 1u)) return ah_fail(c, AH_EOVERFLOW, "overflow");
    if (out_null_count_host) *out_null_count_host = out_valid ? n - (int64_t) * (volatile uint64_t*)&c->pinned[2] : 0;
  }
  return AH_OK;
}
