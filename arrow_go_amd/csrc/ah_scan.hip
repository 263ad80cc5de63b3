// ah_scan.hip — cumulative_sum / cumulative_sum_checked: reduce-then-scan prefix sums.
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
// The reference is one sequential loop.  Here: three kernels, no inter-workgroup waiting —
//   1. tile_sums_kernel     Σ of the valid rows of every tile            (reads  w B/row)
//   2. super_prefix_kernel  exclusive scan of per-8-tile sums (+ start)  (tiny)
//   3. tile_scan_kernel     in-tile scan + tile prefix → out             (reads w, writes w B/row)
// i.e. 3w bytes of traffic per row against 2w algorithmic.  The one-pass alternative (decoupled
// look-back, kept as scripts/micro/scan_lookback_variant.hip.txt) was built and measured first:
// with fence-free self-validating records it reached 0.69–0.88 ms for 2^27 Int64 rows, because
// every tile's prefix has to cross XCDs through sc1 (L2-bypassing) accesses whose round trip is
// ≈3 µs behind the streaming traffic, and twice that sits on every tile's critical path; it
// also needs spin-waits (a co-residency deadlock was observed with an over-subscribed persistent
// grid) and its float results depend on timing.  This version is 0.49 ms, deterministic — the
// summation tree is a fixed function of n — and cannot hang.
//
// Accumulator: unchecked ints scan in uint32 / uint64 (wraparound commutes with truncation); checked
// ints scan EXACTLY in int64 (≤ 32-bit types) or __int128 (64-bit types) so "some running sum
// left the range" is decided exactly; floats scan in double (parallel order — the reference's
// sequential order cannot be reproduced; tolerance in DESIGN.md §4).
#include <limits>
#include <type_traits>
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;

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

// Tile geometry: a tile is 4 wave chunks; a wave's chunk is contiguous (vector k of lane l sits at
// chunk + (k·64 + l)·16 bytes: coalesced), so the in-tile scan is shuffles within a wave plus ONE
// LDS exchange of four wave totals.  VPT = 16-byte vectors per lane per tile.
template <typename T, int VPT>
struct Geom {
  static constexpr int V = 16 / sizeof(T);
  static constexpr int WCHUNK = 64 * VPT * V;
  static constexpr int TILE = (kBlock / 64) * WCHUNK;
};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// rows of one wave chunk + their validity bits (bit j = element j of the vector), with rows at or
// beyond `limit` (first null, !skip_nulls) and beyond n masked off.  ALN: the buffer is 16-byte
// aligned → nontemporal ext-vector loads (the column is streamed, not reused).
template <typename T, int VPT, bool ALN>
__device__ __forceinline__ void load_chunk(const T* __restrict__ in, const uint8_t* __restrict__ valid, int64_t off, int64_t n,
                                           int64_t limit, int64_t wbase, int lane, T (&x)[VPT][16 / sizeof(T)], unsigned (&vb)[VPT]) {
  constexpr int V = 16 / sizeof(T);
#pragma unroll
  for (int k = 0; k < VPT; k++) {
    const int64_t e0 = wbase + ((int64_t)k * 64 + lane) * V;
    if (e0 + V <= n) {
      ah_vec16<T> v;
      if (ALN) v = __builtin_bit_cast(ah_vec16<T>, __builtin_nontemporal_load((const u32x4*)(in + e0)));
      else v = ah_ld16<T>(in + e0);
#pragma unroll
      for (int j = 0; j < V; j++) x[k][j] = v.v[j];
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) x[k][j] = (e0 + j < n) ? in[e0 + j] : (T)0;
    }
    const int cnt = e0 >= n ? 0 : ((n - e0) >= V ? V : (int)(n - e0));
    unsigned b = cnt == 0 ? 0u : (valid ? (unsigned)ah_load_bits64(valid, off + e0, cnt) : ~0u);
    if (cnt < V) b &= (1u << cnt) - 1u;
    if (e0 + V > limit) b &= e0 >= limit ? 0u : ((1u << (int)(limit - e0)) - 1u);
    vb[k] = b;
  }
}

constexpr int kSuper = 8;  // tiles per workgroup in the sums pass = tiles per thread-step in the prefix pass

// 1. tile sums: a workgroup streams kSuper consecutive tiles, one barrier at the end
template <typename T, typename A, int VPT, bool ALN>
__global__ __launch_bounds__(kBlock) void tile_sums_kernel(const T* __restrict__ in, const uint8_t* __restrict__ valid, int64_t off,
                                                            int64_t n, int64_t limit, A* __restrict__ tile_sum, A* __restrict__ super_sum,
                                                            int64_t ntiles) {
  using G = Geom<T, VPT>;
  __shared__ A s_p[kSuper][kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t t0 = (int64_t)blockIdx.x * kSuper;
#pragma unroll 2
  for (int s = 0; s < kSuper; s++) {
    const int64_t tile = t0 + s;
    if (tile >= ntiles) break;
    T x[VPT][G::V];
    unsigned vb[VPT];
    load_chunk<T, VPT, ALN>(in, valid, off, n, limit, tile * G::TILE + (int64_t)wave * G::WCHUNK, lane, x, vb);
    A acc = 0;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
#pragma unroll
      for (int j = 0; j < G::V; j++) acc += ((vb[k] >> j) & 1u) ? (A)x[k][j] : (A)0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += shfl_down_any<A>(acc, o);
    if (lane == 0) s_p[s][wave] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // lanes 0..kSuper-1 hold the tile sums; their sum (in tile order) is the super sum
    A t = 0;
    if (threadIdx.x < kSuper && t0 + threadIdx.x < ntiles) {
#pragma unroll
      for (int w = 0; w < kBlock / 64; w++) t += s_p[threadIdx.x][w];
      tile_sum[t0 + threadIdx.x] = t;
    }
    A sup = 0;
#pragma unroll
    for (int k = 0; k < kSuper; k++) sup += shfl_any<A>(t, k);
    if (threadIdx.x == 0) super_sum[blockIdx.x] = sup;
  }
}

// 2. exclusive scan of the SUPER sums, `start` folded in: one workgroup, each thread a contiguous run.
// (Scanning all tile sums here cost 56 µs for 32768 tiles — 8-byte accesses at a 256-byte lane
// stride; the scan pass instead adds the ≤ 7 tile sums in front of its tile with scalar loads.)
constexpr int kPrefixBlock = 1024;
// carry_in (nullable): the running total of the segments in front of this one (device memory) replaces `start`;
// carry_out (nullable) receives carry + this segment's total, for the next segment.
template <typename A>
__global__ __launch_bounds__(kPrefixBlock) void super_prefix_kernel(const A* __restrict__ super_sum, A* __restrict__ super_excl, int64_t nsuper,
                                                                     A start, const A* __restrict__ carry_in, A* __restrict__ carry_out) {
  if (carry_in) start = *carry_in;
  __shared__ A s_w[kPrefixBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t per = (nsuper + kPrefixBlock - 1) / kPrefixBlock;
  const int64_t lo = (int64_t)threadIdx.x * per;
  const int64_t hi = lo + per < nsuper ? lo + per : nsuper;
  A local = 0;
#pragma unroll 4
  for (int64_t q = lo; q < hi; q++) local += super_sum[q];
  A inc = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    A t = shfl_up_any<A>(inc, o);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  A base = start;
  for (int w = 0; w < wave; w++) base += s_w[w];
  A prev = shfl_up_any<A>(inc, 1);
  if (lane == 0) prev = 0;
  A run = base + prev;
#pragma unroll 4
  for (int64_t q = lo; q < hi; q++) {
    A t = super_sum[q];
    super_excl[q] = run;
    run += t;
  }
  if (carry_out && threadIdx.x == kPrefixBlock - 1) *carry_out = run;  // the last thread's run ends at nsuper (empty runs pass it through)
}

// 3. in-tile scan + tile prefix (= super prefix + the tile sums in front of this tile within its
// super tile: workgroup-uniform, scalar loads).  super_excl == nullptr: a single tile, prefix = start.
template <typename T, typename A, bool CHECKED, int VPT, bool ALN>
__global__ __launch_bounds__(kBlock) void tile_scan_kernel(const T* __restrict__ in, const uint8_t* __restrict__ valid, int64_t off,
                                                            int64_t n, int64_t limit, A start, const A* __restrict__ super_excl,
                                                            const A* __restrict__ tile_sum, T* __restrict__ out,
                                                            unsigned* __restrict__ overflow, int64_t row0, unsigned long long* __restrict__ first_nf) {
  using G = Geom<T, VPT>;
  constexpr int V = G::V;
  constexpr bool kIsInt = std::is_integral<T>::value;
  __shared__ A s_w[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t tile = blockIdx.x;
  const int64_t wb = tile * G::TILE + (int64_t)wave * G::WCHUNK;
  T x[VPT][V];
  unsigned vb[VPT];
  load_chunk<T, VPT, ALN>(in, valid, off, n, limit, wb, lane, x, vb);
  A excl = start;
  if (super_excl) {
    const int64_t sup = tile / kSuper;
    const int in_super = (int)(tile - sup * kSuper);
    excl = super_excl[sup];
    for (int k = 0; k < in_super; k++) excl += tile_sum[sup * kSuper + k];
  }

  // per-lane vector sums, wave scans chained over the VPT vectors
  A inc[VPT];
#pragma unroll
  for (int k = 0; k < VPT; k++) {
    A sacc = 0;
#pragma unroll
    for (int j = 0; j < V; j++) sacc += ((vb[k] >> j) & 1u) ? (A)x[k][j] : (A)0;
    inc[k] = sacc;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      A t = shfl_up_any<A>(inc[k], o);
      if (lane >= o) inc[k] += t;
    }
  }
  A vexcl[VPT];  // exclusive prefix within the wave chunk at the START of this lane's vector k (no subtraction:
  A carry = 0;   // inclusive − own would turn ±inf into NaN and lose bits for floats)
#pragma unroll
  for (int k = 0; k < VPT; k++) {
    A prev = shfl_up_any<A>(inc[k], 1);
    if (lane == 0) prev = 0;
    vexcl[k] = carry + prev;
    carry += shfl_any<A>(inc[k], 63);
  }
  if (lane == 0) s_w[wave] = carry;
  __syncthreads();
  A pre = excl;
#pragma unroll
  for (int w = 0; w < kBlock / 64; w++)
    if (w < wave) pre += s_w[w];

  bool ovf = false;
  unsigned long long nf = ~0ull;  // floats: first row (of the whole column: row0 = this segment's first row) whose running sum is not finite
#pragma unroll
  for (int k = 0; k < VPT; k++) {
    const int64_t e0 = wb + ((int64_t)k * 64 + lane) * V;
    A run = pre + vexcl[k];  // exclusive prefix at this vector's first element
    ah_vec16<T> o;
#pragma unroll
    for (int j = 0; j < V; j++) {
      if ((vb[k] >> j) & 1u) {
        run += (A)x[k][j];
        if (CHECKED && kIsInt) {
          using L = std::numeric_limits<T>;
          if (run > (A)L::max() || run < (A)L::min()) ovf = true;
        }
        o.v[j] = (T)run;
        if (!kIsInt) {
          const T ov = o.v[j];
          if (!(ov - ov == (T)0) && nf == ~0ull) nf = (unsigned long long)(row0 + e0 + j);   // inf − inf and NaN − NaN are NaN
        }
      } else {
        o.v[j] = (T)0;  // null rows keep the zero of the fresh buffer
      }
    }
    if (e0 + V <= n) {
      if (ALN) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), (u32x4*)(out + e0));
      else ah_st16<T>(out + e0, o);
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) if (e0 + j < n) out[e0 + j] = o.v[j];
    }
  }
  if (CHECKED && __any(ovf) && lane == 0) atomicOr(overflow, 1u);
  if (!kIsInt && __any(nf != ~0ull)) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_down(nf, o, 64);
      nf = t < nf ? t : nf;
    }
    // (look before the atomic: in a column that is non-finite from an early row on EVERY wave comes here, and 2^17 same-address
    // atomicMin were 1.4 of that call's 1.95 ms; a wave whose row cannot lower the word leaves it alone)
    if (lane == 0 && nf < __hip_atomic_load(first_nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(first_nf, nf);
  }
}

// ---- floats: a running sum that has become ±inf / NaN stays there -----------------------------------------------
// The reference adds row after row in T (vector_cumulative.go:228-318): once `current` is ±inf it can only stay ±inf or
// turn NaN (an addend of the opposite infinity, or a NaN), and NaN is final — whatever finite values follow.  The tree
// above would come back to finite values.  So: r = first row with a non-finite running sum (found by the scan itself);
// r2 = first valid row after r whose addend turns state s0 = out[r] into NaN (r2 = r if s0 is NaN); rows (r, r2) ← s0,
// rows ≥ r2 ← NaN.  Both kernels return at once in the normal case (no non-finite sum anywhere).
template <typename T>
__global__ __launch_bounds__(kBlock) void sticky_find_kernel(const T* __restrict__ in, const uint8_t* __restrict__ valid, int64_t off,
                                                              int64_t limit, const T* __restrict__ out, const unsigned long long* __restrict__ first_nf,
                                                              unsigned long long* __restrict__ nan_from) {
  const unsigned long long r = *first_nf;
  if (r == ~0ull) return;
  const T s0 = out[r];
  if (s0 != s0) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicMin(nan_from, r); return; }
  unsigned long long best = ~0ull;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)r + 1 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < limit; i += stride) {
    if (!ah_bit(valid, off + i)) continue;
    const T x = in[i];
    if (x != x || x == -s0) { best = (unsigned long long)i; break; }   // NaN, or the opposite infinity (s0 is ±inf here)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_down(best, o, 64);
    best = t < best ? t : best;
  }
  if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(nan_from, best);
}
template <typename T>
__global__ __launch_bounds__(kBlock) void sticky_fill_kernel(const uint8_t* __restrict__ valid, int64_t off, int64_t limit, T* __restrict__ out,
                                                              const unsigned long long* __restrict__ first_nf,
                                                              const unsigned long long* __restrict__ nan_from) {
  const unsigned long long r = *first_nf;
  if (r == ~0ull) return;
  const unsigned long long r2 = *nan_from;
  const T s0 = out[r];   // row r itself is never rewritten
  const T qnan = s0 - s0;  // inf − inf: the NaN this device's adder produces
  // 16 bytes per lane and store where a whole aligned group lies behind r (and, with a validity bitmap, all its rows are valid);
  // row by row at the edges and around nulls — a column that turns NaN early is rewritten once more at the streaming rate
  constexpr int V = 16 / (int)sizeof(T);
  typedef T TV __attribute__((ext_vector_type(V)));
  const int64_t first = (int64_t)r + 1;
  const int64_t g0 = ((uintptr_t)out & 15) == 0 ? first / V : limit;   // (an output that is not 16-byte aligned: row by row throughout)
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t g = g0 + (int64_t)blockIdx.x * kBlock + threadIdx.x; g * V < limit; g += stride) {
    const int64_t i0 = g * V;
    bool whole = i0 >= first && i0 + V <= limit;
    if (whole && valid) {
#pragma unroll
      for (int j = 0; j < V; j++) whole = whole && ah_bit(valid, off + i0 + j);
    }
    if (whole) {
      TV v;
#pragma unroll
      for (int j = 0; j < V; j++) v[j] = (unsigned long long)(i0 + j) < r2 ? s0 : qnan;
      __builtin_nontemporal_store(v, (TV*)(out + i0));
    } else {
#pragma unroll
      for (int j = 0; j < V; j++) {
        const int64_t i = i0 + j;
        if (i >= first && i < limit && ah_bit(valid, off + i)) out[i] = (unsigned long long)i < r2 ? s0 : qnan;
      }
    }
  }
  if (g0 == limit)
    for (int64_t i = first + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < limit; i += stride)
      if (ah_bit(valid, off + i)) out[i] = (unsigned long long)i < r2 ? s0 : qnan;
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

// Segments (OFF by default: c->opt_scan_segment_log2 = 0).  The scan pass re-reads what the sums pass has just read — 24 B/row
// moved for 16 algorithmic.  The idea that segments of ≤ 64–128 MiB would let the 256 MiB Infinity Cache serve the re-read was
// built and measured (scripts/bench_scan_seg.py, profiles/r02_bench_scan_seg.json): 2^27 Int64 rows 0.55 ms in one piece,
// 0.61 / 0.75 / 0.99 ms in 128 / 64 / 32 MiB segments — and a Sum re-reading the SAME 32 … 256 MiB range twenty times runs at
// 5.3 … 6.7 TB/s, below the 7.1 TB/s of a cold 1 GiB stream: on this chip a streamed range that fits the memory-side cache is
// not read back any faster than HBM, so the segments only add 3 launches (and their drain) each.  The running total crosses
// segments through device memory (super_prefix_kernel's carry); kept as a measurement switch and for the tests.
template <typename T, typename A, bool CHECKED, int VPT, bool ALN>
int run_scan(ah_ctx* c, const void* values, const uint8_t* valid, int64_t off, int64_t n, int64_t limit, A start, void* out) {
  using G = Geom<T, VPT>;
  unsigned* overflow = (unsigned*)&c->dscalars[13];
  unsigned long long* first_nf = (unsigned long long*)&c->dscalars[9];
  unsigned long long* nan_from = (unsigned long long*)&c->dscalars[10];
  constexpr bool kFloat = std::is_floating_point<T>::value;
  if (CHECKED) AH_HIP(c, hipMemsetAsync(overflow, 0, sizeof(uint64_t), c->stream));
  if (kFloat) AH_HIP(c, hipMemsetAsync(first_nf, 0xFF, 2 * sizeof(uint64_t), c->stream));
  const int64_t group = (int64_t)G::TILE * kSuper;
  int64_t seg_rows = c->opt_scan_segment_log2 > 0 ? (((int64_t)1 << c->opt_scan_segment_log2) / (int64_t)sizeof(T)) / group * group : 0;
  if (seg_rows <= 0 || n <= seg_rows + seg_rows / 2) seg_rows = n;   // one segment (also: no short tail segment)
  const int64_t nseg = ah_ceil_div(n, seg_rows);
  const int64_t max_tiles = ah_ceil_div(seg_rows < n ? seg_rows : n, G::TILE), max_super = ah_ceil_div(max_tiles, kSuper);
  const size_t tbytes = (((size_t)max_tiles * sizeof(A)) + 127) & ~(size_t)127;
  const size_t sbytes = (((size_t)max_super * sizeof(A)) + 127) & ~(size_t)127;
  uint8_t* scratch = nullptr;
  if (max_tiles > 1 || nseg > 1) {
    int rc = ah_scratch_reserve(c, 2 * (tbytes + 2 * sbytes) + 256, (void**)&scratch);   // two sets: segment k + 1's sums do not wait for k's scan to drain
    if (rc != AH_OK) return rc;
  }
  A* carry = (A*)(scratch + 2 * (tbytes + 2 * sbytes));   // [2]: ping-pong
  for (int64_t sg = 0; sg < nseg; sg++) {
    const int64_t r0 = sg * seg_rows, rn = n - r0 < seg_rows ? n - r0 : seg_rows;
    const int64_t lim = limit <= r0 ? 0 : (limit - r0 < rn ? limit - r0 : rn);
    const T* vin = (const T*)values + r0;
    T* vout = (T*)out + r0;
    const int64_t ntiles = ah_ceil_div(rn, G::TILE);
    const A *super_excl = nullptr, *tile_sum = nullptr;
    if (ntiles > 1 || nseg > 1) {
      const int64_t nsuper = ah_ceil_div(ntiles, kSuper);
      uint8_t* set = scratch + (sg & 1) * (tbytes + 2 * sbytes);
      A* sums = (A*)set;
      A* ssum = (A*)(set + tbytes);
      A* sexcl = (A*)(set + tbytes + sbytes);
      tile_sums_kernel<T, A, VPT, ALN><<<(unsigned)nsuper, kBlock, 0, c->stream>>>(vin, valid, off + r0, rn, lim, sums, ssum, ntiles);
      AH_LAUNCH_CHECK(c);
      super_prefix_kernel<A><<<1, kPrefixBlock, 0, c->stream>>>(ssum, sexcl, nsuper, start, sg > 0 ? &carry[(sg - 1) & 1] : nullptr,
                                                                 nseg > 1 ? &carry[sg & 1] : nullptr);
      AH_LAUNCH_CHECK(c);
      super_excl = sexcl;
      tile_sum = sums;
    }
    tile_scan_kernel<T, A, CHECKED, VPT, ALN><<<(unsigned)ntiles, kBlock, 0, c->stream>>>(vin, valid, off + r0, rn, lim, start, super_excl, tile_sum,
                                                                                          vout, overflow, r0, first_nf);
    AH_LAUNCH_CHECK(c);
  }
  if constexpr (kFloat) {
    const unsigned grid = ah_stream_grid(c, ah_ceil_div(n, kBlock), 8);
    sticky_find_kernel<T><<<grid, kBlock, 0, c->stream>>>((const T*)values, valid, off, limit, (const T*)out, first_nf, nan_from);
    AH_LAUNCH_CHECK(c);
    sticky_fill_kernel<T><<<grid, kBlock, 0, c->stream>>>(valid, off, limit, (T*)out, first_nf, nan_from);
    AH_LAUNCH_CHECK(c);
  }
  return AH_OK;
}

template <typename T, typename A, bool CHECKED, int VPT>
int run_scan_aln(ah_ctx* c, const void* values, const uint8_t* valid, int64_t off, int64_t n, int64_t limit, A start, void* out) {
  const bool aligned = ((((uintptr_t)values) | ((uintptr_t)out)) & 15) == 0 && c->tune_nt;
  return aligned ? run_scan<T, A, CHECKED, VPT, true>(c, values, valid, off, n, limit, start, out)
                 : run_scan<T, A, CHECKED, VPT, false>(c, values, valid, off, n, limit, start, out);
}

template <typename T, typename A, bool CHECKED>
int run_scan_vpt(ah_ctx* c, const void* values, const uint8_t* valid, int64_t off, int64_t n, int64_t limit, A start, void* out) {
  static const int vpt = getenv("ARROWHIP_SCAN_VPT") ? atoi(getenv("ARROWHIP_SCAN_VPT")) : (sizeof(T) >= 4 ? 8 : 4);
  if (vpt == 8) return run_scan_aln<T, A, CHECKED, 8>(c, values, valid, off, n, limit, start, out);
  return run_scan_aln<T, A, CHECKED, 4>(c, values, valid, off, n, limit, start, out);
}

// ---- ONE pass for unchecked integer sums without nulls: decoupled look-back over LARGE tiles ----------------------------------------
// 16 bytes of traffic per Int64 row instead of the 24 of reduce-then-scan.  The look-back's hand-offs cross XCDs (≈ 3 µs each, two on
// every tile's critical path), which sank the 32 KiB-tile version of round 1 (0.69–0.88 ms for 2^27 Int64 rows, §3.4 of DESIGN.md).
// What changed (scripts/micro/scan_onepass.hip, measured step by step): a workgroup of 1024 lanes parks 128 KiB of the column in
// registers, so the 256 resident tiles are 32 MiB of streaming per generation (5 µs in, 5 µs out) around those hand-offs: 0.52 ms;
// the in-tile scan writes its prefixes over the data (no second register array): 0.45 ms; the eight 64-lane scans per wave use DPP
// moves (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31) instead of six ds_bpermute round trips each: 0.40 ms = 5.4 TB/s of the 16
// algorithmic bytes per row.  Smaller tiles with more workgroups per CU are slower (2 × 64 KiB: 0.47 ms, 4 × 32 KiB: 0.63),
// larger ones spill (160 KiB: 0.40, 192 KiB: 0.44); all sixteen waves looking back together (one round for 1024 predecessors): 0.48.
// Tiles are taken in TICKET order (a workgroup only waits for tiles whose workgroups already run: no co-residency assumption);
// records are self-validating words {marker : 32 | payload : 32} moved with relaxed agent-scope atomics, the marker carrying a
// per-context epoch so that the record array is never cleared; integers only: wrap-around sums are associative, so the result does not
// depend on which predecessors' aggregates a look-back happened to find (float sums would).
typedef unsigned long long u64;
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_mov32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned dpp_move(unsigned v) { return dpp_mov32<CTRL, ROW_MASK>(v); }
template <int CTRL, int ROW_MASK> __device__ __forceinline__ u64 dpp_move(u64 v) {
  return ((u64)dpp_mov32<CTRL, ROW_MASK>((unsigned)(v >> 32)) << 32) | dpp_mov32<CTRL, ROW_MASK>((unsigned)v);
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned short dpp_move(unsigned short v) { return (unsigned short)dpp_mov32<CTRL, ROW_MASK>((unsigned)v); }
template <typename E>
__device__ __forceinline__ E wave_incl_scan_dpp(E v) {   // a lane without a source keeps 0
  v += dpp_move<0x111, 0xF>(v);   // row_shr:1
  v += dpp_move<0x112, 0xF>(v);   // row_shr:2
  v += dpp_move<0x114, 0xF>(v);   // row_shr:4
  v += dpp_move<0x118, 0xF>(v);   // row_shr:8
  v += dpp_move<0x142, 0xA>(v);   // row_bcast:15 → rows 1 and 3
  v += dpp_move<0x143, 0xC>(v);   // row_bcast:31 → rows 2 and 3
  return v;
}
__device__ __forceinline__ unsigned read_lane63(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ u64 read_lane63(u64 v) { return ((u64)read_lane63((unsigned)(v >> 32)) << 32) | read_lane63((unsigned)v); }
__device__ __forceinline__ unsigned short read_lane63(unsigned short v) { return (unsigned short)read_lane63((unsigned)v); }

// all ones where bit `pos` of `bits` is set (v_bfe_i32: the one-bit field sign-extended), else 0 — the AND mask of a row's payload
template <typename E> __device__ __forceinline__ E op_bit_mask(unsigned bits, int pos) {
  const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)bits, (unsigned)pos, 1u);
  if constexpr (sizeof(E) == 8) return ((u64)m << 32) | m;
  else return (E)m;
}

constexpr int kOpThreads = 1024;
constexpr int kOpVpt = 8;                 // 16-byte vectors per lane: 128 KiB tiles
constexpr int kOpRecWords = 4;            // per tile: {aggregate lo, hi, inclusive lo, hi}
constexpr unsigned kOpSpinLimit = 1u << 22;   // polls of one record (each ≥ 64 clocks of s_sleep + two L2-bypassing loads ≈ 0.3 µs): about a second
// a look-back that gave up: one word of host-coherent memory (ah_ctx::mailbox[16]) the host reads at its next synchronisation
__device__ __forceinline__ void op_report_stall(unsigned long long* stall) {
  if (stall) __hip_atomic_store(stall, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void op_rec_store(u64* rec, unsigned marker, u64 v) {
  __hip_atomic_store(&rec[0], ((u64)marker << 32) | (v & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&rec[1], ((u64)marker << 32) | (v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool op_rec_load(const u64* rec, unsigned marker, u64* v) {
  const u64 a = __hip_atomic_load(&rec[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 b = __hip_atomic_load(&rec[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((unsigned)(a >> 32) != marker || (unsigned)(b >> 32) != marker) return false;
  *v = (a & 0xffffffffull) | (b << 32);
  return true;
}

// The validity bits of the lane's kOpVpt vectors, packed (bit k·V + j = element j of vector k).  The wave's chunk is 512·V rows =
// 8·V bitmap words: lane w < 8·V reads word w (any bit offset; rows at or beyond `end` — the first null when nulls are not skipped,
// or n — read as null) and leaves it in the wave's 8·V words of LDS; every lane then reads the BYTE that holds its V bits of vector k
// — word k·V + (lane·V >> 6), byte (lane·V >> 3) & 7: one base address, eight immediate offsets — and cuts them out with a bit-field
// extract.  A workgroup is alone on its CU and runs its tiles back to back, so the tile's time is the sum of its phases and every
// vector instruction in front of the in-tile scan costs 16 cycles (4 waves per SIMD × 4) × 32 tiles ≈ 0.24 µs of the call: the first
// version (a ds_bpermute pair per vector + 64-bit shift / mask) and the second (v_readlane pairs + selects) spent ≈ 100 of them here,
// this one ≈ 20.  ONE register per lane through the look-back (eight words would not fit: scalar loads per vector spilled).
// The same words ARE the output's validity (skip_nulls: the input's bits; otherwise ones up to the first null, `end`, and zeros from
// there on — the input's bits below `end` are all ones then): with out_valid the loading lanes store them (the last word byte by
// byte: the bitmap ends with the byte that holds row n − 1, whose bits from row n on stay the caller's) and add their set bits to the
// workgroup's count.
template <int V>
__device__ __forceinline__ unsigned op_lane_bits(const uint8_t* __restrict__ valid, int64_t off, int64_t wbase, int64_t end, int lane,
                                                 uint8_t* __restrict__ out_valid, int64_t n, unsigned long long* s_count, u64* s_words /* the wave's 8·V */) {
  static_assert(kOpVpt == 8 && (V == 2 || V == 4), "op_lane_bits: eight vectors of two or four rows");
  if (lane < kOpVpt * V) {
    const int64_t wpos = wbase + 64 * (int64_t)lane, left = end - wpos;
    const u64 myword = ah_load_bits64(valid, off + wpos, left >= 64 ? 64 : (left < 0 ? 0 : (int)left));
    s_words[lane] = myword;
    if (out_valid && wpos < n) {
      if (n - wpos >= 64) *(u64*)(out_valid + (wpos >> 3)) = myword;
      else {
        for (int64_t b = 0; b * 8 < n - wpos; b++) {
          uint8_t v = (uint8_t)(myword >> (8 * b));
          const int64_t left_bits = n - wpos - 8 * b;   // the bitmap's last byte: the bits from row n on are the caller's (a pre-filled buffer keeps them)
          if (left_bits < 8) v = (uint8_t)((v & ((1u << left_bits) - 1u)) | (out_valid[(wpos >> 3) + b] & ~((1u << left_bits) - 1u)));
          out_valid[(wpos >> 3) + b] = v;
        }
      }
      if (myword) atomicAdd(s_count, (unsigned long long)__popcll(myword));
    }
  }
  // same wave, LDS operations complete in order: a compiler-level ordering point is all the hand-over needs
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint8_t* mybyte = (const uint8_t*)s_words + ((lane * V) >> 6) * 8 + (((lane * V) >> 3) & 7);
  const unsigned sh = (unsigned)(lane * V) & 7u;
  unsigned vbits = 0;
#pragma unroll
  for (int k = 0; k < kOpVpt; k++) vbits |= (((unsigned)mybyte[k * V * 8] >> sh) & ((1u << V) - 1u)) << (k * V);
  return vbits;
}

// E = the accumulator = the element's unsigned twin (uint32 / uint64): rows are summed modulo 2^32 / 2^64.
// NULLS: a null row (validity bit 0, or a row from `limit` on) adds nothing and its output is the zero of a fresh buffer
// (vector_cumulative.go:270-284, 300-318).
// CHECKED (vector_cumulative.go:147-160 checkedAddSigned / Unsigned at every step of the sequential loop): with s_i the running sum
// modulo 2^W, "some running sum left the type's range" ⟺ "some step s_(i−1) + x_i overflows as an addition of W-bit numbers" — the
// first sum to leave the range comes from an exact s_(i−1), so its step's test fires; no sum leaves it ⟹ every s_i is exact and no
// test fires.  Each row's test needs s_(i−1), s_i and x_i = s_i − s_(i−1) only: the prefixes this kernel has anyway.  SIGNED picks
// the test.  The flag is a sticky word all tiles OR into; nothing travels through the look-back records.
template <typename E, bool NULLS = false, bool CHECKED = false, bool SIGNED = false>
__global__ __launch_bounds__(kOpThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void scan_onepass_kernel(
    const E* __restrict__ in, E* __restrict__ out, int64_t n, E start, u64* __restrict__ recs, unsigned* __restrict__ ticket, unsigned ticket_base, unsigned epoch,
    const uint8_t* __restrict__ valid = nullptr, int64_t off = 0, int64_t limit = 0, unsigned* __restrict__ overflow = nullptr,
    uint8_t* __restrict__ out_valid = nullptr, unsigned long long* __restrict__ valid_count = nullptr, unsigned long long* stall = nullptr) {
  constexpr int V = 16 / sizeof(E);
  typedef E EV __attribute__((ext_vector_type(V)));
  constexpr int TILE = kOpThreads * kOpVpt * V;   // rows
  __shared__ E s_wave[kOpThreads / 64];
  __shared__ E s_prefix;
  __shared__ unsigned s_tile;
  __shared__ unsigned long long s_count;   // NULLS with out_valid: set validity bits of this workgroup's tiles
  __shared__ u64 s_bits[NULLS ? (kOpThreads / 64) * kOpVpt * V : 1];   // NULLS: every wave's 8·V validity words of the tile (op_lane_bits)
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);   // (scalar: the wave's chunk is addressed from scalar registers)
  const unsigned m_agg = epoch * 4u + 1u, m_inc = epoch * 4u + 2u;
  const int64_t ntiles = (n + TILE - 1) / TILE;
  if (NULLS && t == 0) s_count = 0;
  for (;;) {
    if (t == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
    const int64_t tile = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)s_tile);
    if (tile >= ntiles) {
      if (NULLS && t == 0 && valid_count && s_count) atomicAdd(valid_count, s_count);   // (the tile's last barrier is behind every LDS add)
      return;
    }
    const int64_t wbase = tile * TILE + (int64_t)wave * (64 * kOpVpt * V);
    EV x[kOpVpt];
    unsigned vbits = 0;
    if constexpr (NULLS) vbits = op_lane_bits<V>(valid, off, wbase, limit < n ? limit : n, lane, out_valid, n, &s_count, &s_bits[NULLS ? wave * kOpVpt * V : 0]);
    // the lane's first kfull vectors lie inside the column (all eight unless this is the column's last chunk): a 32-bit test per vector
    const int lane_rows = (int)(n - wbase < (int64_t)(64 * kOpVpt * V) ? (n - wbase < 0 ? 0 : n - wbase) : (int64_t)(64 * kOpVpt * V)) - lane * V;
    const int kfull = lane_rows >= V ? (lane_rows - V) / (64 * V) + 1 : 0;
    const bool whole = wbase + 64 * kOpVpt * V <= n;   // the wave's chunk lies inside the column (scalar: one branch per wave instead of a bounds test per lane and vector)
    if (whole) {
      const EV* src = (const EV*)(in + wbase) + lane;
#pragma unroll
      for (int k = 0; k < kOpVpt; k++) {
        x[k] = __builtin_nontemporal_load(src + k * 64);
        if (NULLS) {
#pragma unroll
          for (int j = 0; j < V; j++) x[k][j] &= op_bit_mask<E>(vbits, k * V + j);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < kOpVpt; k++) {
        const int64_t e = wbase + ((int64_t)k * 64 + lane) * V;
        if (e + V <= n) x[k] = __builtin_nontemporal_load((const EV*)(in + e));
        else {
#pragma unroll
          for (int j = 0; j < V; j++) x[k][j] = e + j < n ? in[e + j] : (E)0;
        }
        if (NULLS) {
#pragma unroll
          for (int j = 0; j < V; j++) x[k][j] &= op_bit_mask<E>(vbits, k * V + j);
        }
      }
    }
    // x[k] becomes the inclusive prefix of its rows inside the wave's chunk, in place
    E run = 0;
#pragma unroll
    for (int k = 0; k < kOpVpt; k++) {
#pragma unroll
      for (int j = 1; j < V; j++) x[k][j] += x[k][j - 1];
      const E s = x[k][V - 1], inc = wave_incl_scan_dpp<E>(s);
      const E pre = run + inc - s;
      run += read_lane63(inc);
#pragma unroll
      for (int j = 0; j < V; j++) x[k][j] += pre;
    }
    if (lane == 0) s_wave[wave] = run;
    __syncthreads();
    E wpre = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kOpThreads / 64; w++) { const E v = s_wave[w]; if (w < wave) wpre += v; total += v; }
    if (wave == 0) {   // publish the aggregate, look back, publish the inclusive prefix
      u64* mine = recs + (size_t)tile * kOpRecWords;
      E excl = start;
      if (tile == 0) {
        if (lane == 0) op_rec_store(mine + 2, m_inc, (u64)(E)(start + total));
      } else {
        if (lane == 0) op_rec_store(mine, m_agg, (u64)total);
        E acc = 0;
        for (int64_t back = tile - 1;; back -= 64) {
          const int64_t p = back - lane;
          u64 v = 0;
          int kind = 0;   // 0 before tile 0, 1 aggregate, 2 inclusive
          if (p >= 0) {
            const u64* r = recs + (size_t)p * kOpRecWords;
            for (unsigned spins = 0;; spins++) {
              if (op_rec_load(r + 2, m_inc, &v)) { kind = 2; break; }
              if (op_rec_load(r, m_agg, &v)) { kind = 1; break; }
              // Ticket order means the predecessor's workgroup is running: it reports within microseconds.  One that never does (it
              // faulted; the record array was overwritten) must not hang the device: after ≈ a second of polling the stall is reported
              // to the host (checked at the call's own synchronisation and by ah_sync) and the look-back goes on with what it has.
              if (spins >= kOpSpinLimit) { op_report_stall(stall); v = 0; kind = 2; break; }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          const u64 incm = __ballot(kind == 2);
          const int stop = incm ? __builtin_ctzll(incm) : 64;   // the nearest predecessor with an inclusive prefix
          u64 contrib = (lane <= stop && kind != 0) ? v : 0;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) contrib += __shfl_down(contrib, o, 64);
          acc += (E)__shfl(contrib, 0, 64);
          if (incm || back - 63 <= 0) break;
        }
        excl = acc;   // (holds `start` through tile 0's inclusive record)
        if (lane == 0) op_rec_store(mine + 2, m_inc, (u64)(E)(excl + total));
      }
      if (lane == 0) s_prefix = excl;
    }
    __syncthreads();
    const E base = s_prefix + wpre;
    if (NULLS) asm volatile("" : "+v"(vbits));   // a new value to the compiler: the sixteen row masks are cut out of it again, not kept alive through the look-back
    unsigned ovacc = 0;   // CHECKED: bit 31 = some step overflowed
    E carry_in = base;   // CHECKED: the running sum in front of the wave's vector k
#pragma unroll
    for (int k = 0; k < kOpVpt; k++) {
      const int64_t e = wbase + ((int64_t)k * 64 + lane) * V;
#pragma unroll
      for (int j = 0; j < V; j++) x[k][j] += base;
      if (CHECKED) {
        // the running sum in front of the lane's first element: the last element of the lane to the left, or what the wave's vector started from
        E prev = dpp_move<0x138, 0xF>(x[k][V - 1]);   // wave_shr:1 — the left neighbour's last running sum (a ds_bpermute pair per vector before)
        if (lane == 0) prev = carry_in;
        carry_in = read_lane63(x[k][V - 1]);
#pragma unroll
        for (int j = 0; j < V; j++) {
          const E s = x[k][j], xi = s - prev;
          if (SIGNED) ovacc |= (unsigned)(((prev ^ s) & (xi ^ s)) >> (sizeof(E) * 8 - 32));   // bit 31: both operands' signs differ from the sum's
          else ovacc |= s < prev ? 0x80000000u : 0u;                                          // carry out
          prev = s;
        }
      }
      if (NULLS) {   // a null row's output is the zero of a fresh buffer
#pragma unroll
        for (int j = 0; j < V; j++) x[k][j] &= op_bit_mask<E>(vbits, k * V + j);
      }
      if (k < kfull) __builtin_nontemporal_store(x[k], (EV*)(out + wbase) + lane + k * 64);
      else {
#pragma unroll
        for (int j = 0; j < V; j++) if (e + j < n) out[e + j] = x[k][j];
      }
    }
    if (CHECKED && __any((ovacc >> 31) != 0u) && lane == 0) atomicOr(overflow, 1u);
    __syncthreads();   // s_tile / s_wave are taken again
  }
}

// ---- ONE pass for Float64 sums: the same tiles, a look-back whose GROUPING is fixed ------------------------------------------------
// Float additions do not associate, so the integer kernel's look-back — "add whatever aggregates and prefixes the predecessors have
// published so far" — would give results that depend on timing.  Here every tile's prefix is ONE fixed expression of tile totals:
//     tiles come in blocks of 64 and super blocks of 4096;   B_k = tree(T of block k's 64 tiles);   S_j = S_(j−1) + tree(B of super block j);
//     prefix(t) = (S_(super before t) + tree(B of the blocks before t's in its super block)) + tree(T of the tiles before t in its block)
// where tree() is a fixed 64-lane shuffle reduction with +0 in the empty lanes, T a tile's total from the fixed in-tile tree.  A tile
// publishes T as soon as it has it; the last tile of a block publishes B; the last tile of a super block publishes S.  Nobody waits
// for a PREFIX of a neighbour (only the super-block chain does: one hop per 2^26 rows), so the hand-offs are as short as the integer
// kernel's, and the bytes are a function of the input alone.  24 → 16 bytes moved per row.
// Not handled here (they keep reduce-then-scan): nulls, Float32 (its accumulator is double: half a tile's registers more).
constexpr int kOpFBlk = 64, kOpFSup = kOpFBlk * 64;
__device__ __forceinline__ double op_f64_dpp_add(double v, u64 moved) { return v + __builtin_bit_cast(double, moved); }
__device__ __forceinline__ double wave_incl_scan_dpp_f64(double v) {   // lanes without a source add +0
  v = op_f64_dpp_add(v, dpp_move<0x111, 0xF>(__builtin_bit_cast(u64, v)));
  v = op_f64_dpp_add(v, dpp_move<0x112, 0xF>(__builtin_bit_cast(u64, v)));
  v = op_f64_dpp_add(v, dpp_move<0x114, 0xF>(__builtin_bit_cast(u64, v)));
  v = op_f64_dpp_add(v, dpp_move<0x118, 0xF>(__builtin_bit_cast(u64, v)));
  v = op_f64_dpp_add(v, dpp_move<0x142, 0xA>(__builtin_bit_cast(u64, v)));
  v = op_f64_dpp_add(v, dpp_move<0x143, 0xC>(__builtin_bit_cast(u64, v)));
  return v;
}
__device__ __forceinline__ double wave_tree_sum_f64(double v) {   // every lane takes part (empty ones hold +0); the total in lane 0
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return __shfl(v, 0, 64);
}
// the record at `rec` once it carries `marker` (bounded: see the integer kernel)
__device__ __forceinline__ double op_rec_wait_f64(const u64* rec, unsigned marker, unsigned long long* stall) {
  u64 v = 0;
  for (unsigned spins = 0;; spins++) {
    if (op_rec_load(rec, marker, &v)) break;
    if (spins >= kOpSpinLimit) { op_report_stall(stall); v = 0; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  return __builtin_bit_cast(double, v);
}
__global__ __launch_bounds__(kOpThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void scan_onepass_f64_kernel(
    const double* __restrict__ in, double* __restrict__ out, int64_t n, double start, u64* __restrict__ recs, u64* __restrict__ super_recs,
    unsigned* __restrict__ ticket, unsigned epoch, unsigned long long* __restrict__ first_nf, unsigned long long* stall) {
  constexpr int V = 2;
  typedef double DV __attribute__((ext_vector_type(2)));
  constexpr int TILE = kOpThreads * kOpVpt * V;   // rows
  __shared__ double s_wave[kOpThreads / 64];
  __shared__ double s_prefix;
  __shared__ unsigned s_tile;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const unsigned m_agg = epoch * 4u + 1u, m_blk = epoch * 4u + 2u, m_sup = epoch * 4u + 3u;
  const int64_t ntiles = (n + TILE - 1) / TILE;
  for (;;) {
    if (t == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const int64_t tile = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)s_tile);
    if (tile >= ntiles) break;
    const int64_t wbase = tile * TILE + (int64_t)wave * (64 * kOpVpt * V);
    DV x[kOpVpt];
    const int lane_rows = (int)(n - wbase < (int64_t)(64 * kOpVpt * V) ? (n - wbase < 0 ? 0 : n - wbase) : (int64_t)(64 * kOpVpt * V)) - lane * V;
    const int kfull = lane_rows >= V ? (lane_rows - V) / (64 * V) + 1 : 0;
    if (wbase + 64 * kOpVpt * V <= n) {
      const DV* src = (const DV*)(in + wbase) + lane;
#pragma unroll
      for (int k = 0; k < kOpVpt; k++) x[k] = __builtin_nontemporal_load(src + k * 64);
    } else {
#pragma unroll
      for (int k = 0; k < kOpVpt; k++) {
        const int64_t e = wbase + ((int64_t)k * 64 + lane) * V;
        if (e + V <= n) x[k] = __builtin_nontemporal_load((const DV*)(in + e));
        else {
#pragma unroll
          for (int j = 0; j < V; j++) x[k][j] = e + j < n ? in[e + j] : 0.0;
        }
      }
    }
    // x[k] becomes the inclusive prefix of its rows inside the wave's chunk, in place.  Exclusive prefixes come from a shift of the
    // inclusive ones, never from a subtraction (inf − inf)
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < kOpVpt; k++) {
      x[k][1] += x[k][0];
      const double inc = wave_incl_scan_dpp_f64(x[k][1]);
      double ex = __builtin_bit_cast(double, dpp_move<0x138, 0xF>(__builtin_bit_cast(u64, inc)));   // wave_shr:1
      if (lane == 0) ex = 0.0;
      const double pre = run + ex;
      run += __builtin_bit_cast(double, read_lane63(__builtin_bit_cast(u64, inc)));
      x[k][0] += pre;
      x[k][1] += pre;
    }
    if (lane == 0) s_wave[wave] = run;
    __syncthreads();
    double wpre = 0.0, total = 0.0;
#pragma unroll
    for (int w = 0; w < kOpThreads / 64; w++) { const double v = s_wave[w]; if (w < wave) wpre += v; total += v; }
    if (wave == 0) {
      u64* mine = recs + (size_t)tile * kOpRecWords;
      if (lane == 0) op_rec_store(mine, m_agg, __builtin_bit_cast(u64, total));
      const int64_t blk0 = tile & ~(int64_t)(kOpFBlk - 1), sup0 = tile & ~(int64_t)(kOpFSup - 1);
      const int in_blk = (int)(tile - blk0), blks = (int)((blk0 - sup0) / kOpFBlk);
      // the tiles before mine in my block (mine too in lane in_blk: the block's total, should I be its last tile)
      double tv = 0.0;
      if (lane < in_blk) tv = op_rec_wait_f64(recs + (size_t)(blk0 + lane) * kOpRecWords, m_agg, stall);
      const double tiles_before = wave_tree_sum_f64(tv);
      // the blocks before mine in my super block
      double bv = 0.0;
      if (lane < blks) bv = op_rec_wait_f64(recs + (size_t)(sup0 + (int64_t)lane * kOpFBlk + kOpFBlk - 1) * kOpRecWords + 2, m_blk, stall);
      const double blocks_before = wave_tree_sum_f64(bv);
      double sup = start;
      if (sup0 > 0) sup = op_rec_wait_f64(super_recs + (size_t)(sup0 / kOpFSup - 1) * 2, m_sup, stall);   // (uniform: every lane polls the same words)
      const double excl = (sup + blocks_before) + tiles_before;
      if (in_blk == kOpFBlk - 1) {
        const double bsum = wave_tree_sum_f64(lane == in_blk ? total : tv);
        if (lane == 0) op_rec_store(mine + 2, m_blk, __builtin_bit_cast(u64, bsum));
        if (tile - sup0 == kOpFSup - 1) {
          const double ssum = wave_tree_sum_f64(lane == blks ? bsum : bv);
          if (lane == 0) op_rec_store(super_recs + (size_t)(sup0 / kOpFSup) * 2, m_sup, __builtin_bit_cast(u64, sup + ssum));
        }
      }
      if (lane == 0) s_prefix = excl;
    }
    __syncthreads();
    const double base = s_prefix + wpre;
    unsigned bad = 0;   // bit k·V + j: the running sum of that row is not finite
#pragma unroll
    for (int k = 0; k < kOpVpt; k++) {
      const int64_t e = wbase + ((int64_t)k * 64 + lane) * V;
      x[k][0] += base;
      x[k][1] += base;
#pragma unroll
      for (int j = 0; j < V; j++) bad |= (x[k][j] - x[k][j] == 0.0) ? 0u : 1u << (k * V + j);   // inf − inf and NaN − NaN are NaN (rows beyond n hold the last prefix: finite or already counted)
      if (k < kfull) __builtin_nontemporal_store(x[k], (DV*)(out + wbase) + lane + k * 64);
      else {
#pragma unroll
        for (int j = 0; j < V; j++) if (e + j < n) out[e + j] = x[k][j];
      }
    }
    if (__any(bad != 0u)) {   // (never, in a column whose running sums stay finite)
      unsigned long long nf = ~0ull;
      if (bad) {
        const int b = __builtin_ctz(bad);   // the lane's first such row: vector b / V, element b % V
        const int64_t row = wbase + ((int64_t)(b / V) * 64 + lane) * V + (b % V);
        if (row < n) nf = (unsigned long long)row;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long v = __shfl_down(nf, o, 64);
        nf = v < nf ? v : nf;
      }
      if (lane == 0 && nf < __hip_atomic_load(first_nf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(first_nf, nf);
    }
    __syncthreads();   // s_tile / s_wave are taken again
  }
}

// what a one-pass launch starts from, in ONE small launch (three memsets were 14 µs of fills in front of a 0.39 ms kernel): the ticket
// word at zero; `ones` (nullable): two words of all ones (first non-finite row, first NaN row); `zeros` (nullable): three zero words
// (overflow flag, a spare, the validity count)
__global__ void op_prep_kernel(unsigned* __restrict__ ticket, unsigned long long* __restrict__ ones, unsigned long long* __restrict__ zeros) {
  if (threadIdx.x == 0) *ticket = 0u;
  if (ones && threadIdx.x < 2) ones[threadIdx.x] = ~0ull;
  if (zeros && threadIdx.x < 3) zeros[threadIdx.x] = 0ull;
}

// the context's record array for `ntiles` tiles (+ extra_words behind them), a fresh epoch, the ticket word at zero
static int onepass_prepare(ah_ctx* c, int64_t ntiles, size_t extra_words, u64** recs, unsigned** ticket, unsigned long long* ones = nullptr, unsigned long long* zeros = nullptr) {
  const size_t need = ((size_t)ntiles * kOpRecWords + extra_words) * 8 + 64;
  if (need > c->scan_recs_bytes) {   // the context's own record array: only these kernels write it, so every stale word carries an older epoch
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if (c->scan_recs) AH_HIP(c, hipFree(c->scan_recs));
    c->scan_recs = nullptr; c->scan_recs_bytes = 0;
    const size_t want = (need + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    AH_HIP(c, hipMalloc(&c->scan_recs, want));
    AH_HIP(c, hipMemsetAsync(c->scan_recs, 0, want, c->stream));
    c->scan_recs_bytes = want;
    c->scan_epoch = 0;
  }
  if (c->scan_epoch >= (1u << 29)) {   // before a marker wraps
    AH_HIP(c, hipMemsetAsync(c->scan_recs, 0, c->scan_recs_bytes, c->stream));
    c->scan_epoch = 0;
  }
  c->scan_epoch++;
  *recs = (u64*)c->scan_recs;
  *ticket = (unsigned*)((uint8_t*)c->scan_recs + c->scan_recs_bytes - 64);   // the last 64 bytes: the ticket word
  // the ticket starts from zero in EVERY launch (in stream order): a launch that did not run to its end — a fault, an
  // asynchronous error — cannot leave the word out of step with a count the host keeps (the first version kept a running base)
  op_prep_kernel<<<1, 64, 0, c->stream>>>(*ticket, ones, zeros);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

// mode: bit 0 nulls (valid / limit), bit 1 checked, bit 2 signed (checked only)
template <typename E>
int run_onepass(ah_ctx* c, const void* values, int64_t n, E start, void* out, int mode = 0, const uint8_t* valid = nullptr, int64_t off = 0, int64_t limit = 0,
                uint8_t* out_valid = nullptr) {
  constexpr int TILE = kOpThreads * kOpVpt * (16 / (int)sizeof(E));
  const int64_t ntiles = ah_ceil_div(n, (int64_t)TILE);
  u64* recs;
  unsigned* ticket;
  unsigned* overflow = (unsigned*)&c->dscalars[13];
  unsigned long long* vcount = (unsigned long long*)&c->dscalars[15];   // out_valid: the kernel writes the output's validity and counts its set bits
  int prc = onepass_prepare(c, ntiles, 0, &recs, &ticket, nullptr, ((mode & 2) || out_valid) ? (unsigned long long*)overflow : nullptr);   // [13] overflow, [14] (free by now), [15] count
  if (prc != AH_OK) return prc;
  int64_t grid = ntiles < c->num_cu ? ntiles : c->num_cu;   // one workgroup per CU (104 registers × 1024 lanes), tiles by ticket
  const E* vin = (const E*)values;
  E* vout = (E*)out;
  const unsigned g = (unsigned)grid;
#define AH_OP(NULLS, CHECKED, SIGNED) scan_onepass_kernel<E, NULLS, CHECKED, SIGNED><<<g, kOpThreads, 0, c->stream>>>(vin, vout, n, start, recs, ticket, 0u, c->scan_epoch, valid, off, limit, overflow, out_valid, vcount, &c->mailbox[16])
  if constexpr (sizeof(E) < 4) {   // 2-byte columns: unchecked without nulls only
    if (mode != 0) return ah_fail(c, AH_EINVALID, "cumulative_sum: one-pass mode %d for a narrow type", mode);
    AH_OP(false, false, false);
  } else
  switch (mode) {
    case 0: AH_OP(false, false, false); break;
    case 1: AH_OP(true, false, false); break;
    case 2: AH_OP(false, true, false); break;
    case 3: AH_OP(true, true, false); break;
    case 6: AH_OP(false, true, true); break;
    case 7: AH_OP(true, true, true); break;
    default: return ah_fail(c, AH_EINVALID, "cumulative_sum: bad one-pass mode %d", mode);
  }
#undef AH_OP
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

// Float64 without nulls, one pass (scan_onepass_f64_kernel) + the sticky pass of the reduce-then-scan path
static int run_onepass_f64(ah_ctx* c, const void* values, int64_t n, double start, void* out) {
  constexpr int TILE = kOpThreads * kOpVpt * 2;
  const int64_t ntiles = ah_ceil_div(n, (int64_t)TILE);
  u64* recs;
  unsigned* ticket;
  unsigned long long* first_nf = (unsigned long long*)&c->dscalars[9];
  unsigned long long* nan_from = (unsigned long long*)&c->dscalars[10];
  int prc = onepass_prepare(c, ntiles, (size_t)(ntiles / kOpFSup + 2) * 2, &recs, &ticket, first_nf);   // + both sticky words: none
  if (prc != AH_OK) return prc;
  const unsigned grid = (unsigned)(ntiles < c->num_cu ? ntiles : c->num_cu);
  scan_onepass_f64_kernel<<<grid, kOpThreads, 0, c->stream>>>((const double*)values, (double*)out, n, start, recs, recs + (size_t)ntiles * kOpRecWords, ticket,
                                                              c->scan_epoch, first_nf, &c->mailbox[16]);
  AH_LAUNCH_CHECK(c);
  const unsigned sgrid = ah_stream_grid(c, ah_ceil_div(n, kBlock), 8);
  sticky_find_kernel<double><<<sgrid, kBlock, 0, c->stream>>>((const double*)values, nullptr, 0, n, (const double*)out, first_nf, nan_from);
  AH_LAUNCH_CHECK(c);
  sticky_fill_kernel<double><<<sgrid, kBlock, 0, c->stream>>>(nullptr, 0, n, (double*)out, first_nf, nan_from);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <typename T>
int dispatch_scan(ah_ctx* c, const void* values, const uint8_t* valid, int64_t off, int64_t n, int64_t limit, const void* start_host,
                  int checked, void* out, uint8_t* out_valid, bool* validity_done) {
  T start = 0;
  if (start_host) memcpy(&start, start_host, sizeof(T));
  if constexpr (std::is_floating_point<T>::value) {
    // Float64 without nulls: one pass with a fixed-grouping look-back (option scan_onepass 5: reduce-then-scan, for measurement)
    if (sizeof(T) == 8 && c->opt_scan_onepass && c->opt_scan_onepass != 5 && !c->capturing && n >= ((int64_t)1 << 18) && !valid && limit >= n &&
        ((((uintptr_t)values) | ((uintptr_t)out)) & 15) == 0 && c->tune_nt)
      return run_onepass_f64(c, values, n, (double)start, out);
    return run_scan_vpt<T, double, false>(c, values, valid, off, n, limit, (double)start, out);
  } else {
    // unchecked: wraparound commutes with truncation, so the narrowest accumulator ≥ T does
    // (not while a graph records: the epoch and the ticket base are host state a replay would repeat)
    // ONE pass for every 4- / 8-byte integer column — unchecked or checked, with or without nulls (round 5; round 4: unchecked without
    // nulls only).  The narrow types stay on reduce-then-scan (their tiles would be 2^16 … 2^17 rows of 1 … 2 bytes: not measured).
    // 2-byte columns (round 5, last): the same kernel over tiles of 2^16 rows when unchecked and without nulls (options 3 / 4 keep them on
    // reduce-then-scan, for measurement).  1-byte columns stay there: their 128 bytes per lane unpack into as many registers (35 spilled).
    if (sizeof(T) == 2 && c->opt_scan_onepass && c->opt_scan_onepass < 3 && !c->capturing && n >= ((int64_t)1 << 18) && ((((uintptr_t)values) | ((uintptr_t)out)) & 15) == 0 &&
        c->tune_nt && !checked && !valid && limit >= n) {
      if constexpr (sizeof(T) == 2) return run_onepass<unsigned short>(c, values, n, (unsigned short)start, out);
    }
    if (sizeof(T) >= 4 && c->opt_scan_onepass && !c->capturing && n >= ((int64_t)1 << 18) && ((((uintptr_t)values) | ((uintptr_t)out)) & 15) == 0 && c->tune_nt &&
        (c->opt_scan_onepass < 3 || (!checked && !valid))) {   // (3 / 4: round 4's gate — unchecked 4- / 8-byte columns without nulls only; measurement switches)
      const bool nulls = valid != nullptr || limit < n;
      const int mode = (nulls ? 1 : 0) | (checked ? 2 : 0) | (checked && std::is_signed<T>::value ? 4 : 0);
      // the output's validity and its count come out of the same pass (the words the kernel reads are the words to write) — option
      // scan_onepass 2 keeps the bitmap copy + popcount of the first version, for measurement
      uint8_t* ov = (nulls && out_valid && (((uintptr_t)out_valid) & 7) == 0 && c->opt_scan_onepass != 2) ? out_valid : nullptr;
      *validity_done = ov != nullptr;
      if constexpr (sizeof(T) == 8) return run_onepass<unsigned long long>(c, values, n, (unsigned long long)start, out, mode, valid, off, limit, ov);
      else if constexpr (sizeof(T) == 4) return run_onepass<unsigned>(c, values, n, (unsigned)start, out, mode, valid, off, limit, ov);
    }
    if (!checked) {
      if constexpr (sizeof(T) == 8) return run_scan_vpt<T, unsigned long long, false>(c, values, valid, off, n, limit, (unsigned long long)start, out);
      else return run_scan_vpt<T, unsigned, false>(c, values, valid, off, n, limit, (unsigned)start, out);
    }
    if constexpr (sizeof(T) == 8) return run_scan_vpt<T, i128, true>(c, values, valid, off, n, limit, (i128)start, out);
    else return run_scan_vpt<T, long long, true>(c, values, valid, off, n, limit, (long long)start, out);
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
  bool validity_done = false;   // the one-pass kernel wrote out_valid and left its set-bit count in dscalars[15]
  switch (type) {
    case AH_UINT8: rc = dispatch_scan<uint8_t>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    case AH_INT8: rc = dispatch_scan<int8_t>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    case AH_UINT16: rc = dispatch_scan<uint16_t>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    case AH_INT16: rc = dispatch_scan<int16_t>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    case AH_UINT32: rc = dispatch_scan<uint32_t>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    case AH_INT32: rc = dispatch_scan<int32_t>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    case AH_UINT64: rc = dispatch_scan<uint64_t>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    case AH_INT64: rc = dispatch_scan<int64_t>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    case AH_FLOAT32: rc = dispatch_scan<float>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    case AH_FLOAT64: rc = dispatch_scan<double>(c, values, valid, off, n, limit, start_host, checked, out_values, out_valid, &validity_done); break;
    default: return ah_fail(c, AH_ENOTIMPL, "cumulative sum input type must be numeric");
  }
  if (rc != AH_OK) return rc;
  if (out_valid && !validity_done) {
    // validity: skip_nulls → the input's validity; otherwise ones up to the first null, zeros after
    if (valid && skip_nulls) rc = ah_copy_bitmap(c, valid, off, n, out_valid, 0, 0);
    else {
      rc = ah_set_bits_to(c, out_valid, 0, limit, 1);
      if (rc == AH_OK && limit < n) rc = ah_set_bits_to(c, out_valid, limit, n - limit, 0);
    }
    if (rc != AH_OK) return rc;
  }
  bool need_sync = checked || out_null_count_host;
  if (out_null_count_host && out_valid && !validity_done) {
    rc = ah_popcount_async(c, out_valid, 0, n, (unsigned long long*)&c->dscalars[15]);
    if (rc != AH_OK) return rc;
  }
  if (need_sync) {
    AH_HIP(c, hipMemcpyAsync(c->pinned, &c->dscalars[13], 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if ((rc = ah_check_stall(c)) != AH_OK) return rc;
    if (checked && (*(volatile unsigned*)&c->pinned[0] & 1u)) return ah_fail(c, AH_EOVERFLOW, "overflow");
    if (out_null_count_host) *out_null_count_host = out_valid ? n - (int64_t) * (volatile uint64_t*)&c->pinned[2] : 0;
  }
  return AH_OK;
}
