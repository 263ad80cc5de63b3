// ah_sort.hip — sort_indices of one numeric array: stable LSD radix sort (row §8(f)-2).
//
// Replaces kernels.SortIndices for a single key over one array (arrow/compute/internal/kernels/
// vector_sort.go:388-481 → arraySortOneColumnRange, vector_sort_internal.go:252-273) behind
// compute's "sort_indices" (compute/vector_sort.go:42-52):
//   stable; nulls partitioned to the end or the start (key.NullPlacement), NaNs next to them
//   ([rest, NaNs, nulls] / [nulls, NaNs, rest] — partitionNullLikes :90-140: NaN placement follows the
//   null placement, not the order), the rest ordered by cmp.Compare, negated for Descending
//   (vector_sort_support.go:88-94): ties keep their input order in BOTH directions, −0.0 == +0.0.
// The reference is partition + slices.SortStableFunc with a comparator closure per pair.
//
// Here: (1) one stable 3-way partition pass straight off the column (category = rest / NaN / null)
// that also maps every value to an order-preserving unsigned key (sign flip for ints, the IEEE trick
// for floats with −0 → +0, complemented for Descending) and pairs it with its 32-bit row number;
// (2) a bitwise AND / OR reduction of the keys tells which key bytes vary at all; (3) one stable
// 8-bit LSD pass per varying byte over the `rest` range only — per pass: tile histograms →
// exclusive scan → scatter, the in-tile ranks from wave-level match-any (8 ballots) so the pass is
// stable by construction; (4) widening the row numbers to the uint64 output.  Equal keys never
// change relative order in any pass, which is exactly SortStableFunc's contract, and the NaN / null
// groups keep their input order because only pass (1) ever moves them.
// Traffic: 8 (+1/8) B/row for (1) and (2), ≈ 32 B/row per LSD pass ((key, row) read twice, written
// once), 12 B/row for (4): an Int64 column with all 8 bytes varying moves ≈ 290 B/row.
#include <algorithm>
#include <type_traits>
#include <vector>
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kItems = 8;                       // rounds of 64 rows per wave per tile
constexpr int kTile = kBlock * kItems;          // 2048 rows
constexpr int kRadix = 256;
constexpr int kAndOrGrid = 4096;                 // most workgroups the two key-reduction kernels are launched with (their partials live in SortBuffers::andor)
constexpr int kMaxTilesPerBlock = 8;             // tiles one workgroup handles = granularity of the histogram / scan

// small inputs keep one tile per workgroup (parallelism), large ones eight (fewer histogram rows)
static inline int tiles_per_block(int64_t n) { return n >= ((int64_t)1 << 24) ? kMaxTilesPerBlock : 1; }

enum { kCatRest = 0, kCatNaN = 1, kCatNull = 2 };

// order-preserving unsigned key, zero-extended to 64 bits
template <typename T>
__device__ __forceinline__ unsigned long long make_key(T v, bool descending) {
  using U = typename std::make_unsigned<typename std::conditional<std::is_floating_point<T>::value,
                                                                    typename std::conditional<sizeof(T) == 4, int, long long>::type, T>::type>::type;
  constexpr U sign = (U)1 << (sizeof(T) * 8 - 1);
  U k;
  if constexpr (std::is_floating_point<T>::value) {
    if (v == (T)0) v = (T)0;  // −0.0 → +0.0: they tie (cmp.Compare)
    U b = __builtin_bit_cast(U, v);
    k = (b & sign) ? (U)~b : (U)(b | sign);
  } else if constexpr (std::is_signed<T>::value) {
    k = (U)v ^ sign;
  } else {
    k = (U)v;
  }
  if (descending) k = (U)~k;
  return (unsigned long long)k;
}

// ---- the element source of a pass -------------------------------------------------------------
// Column<T>: pass (1) — reads the column, digit = category, emits (key, row).
// Pairs:     an LSD pass — reads (key, row) pairs, digit = a key byte.
template <typename T>
struct Column {
  const T* values;
  const uint8_t* valid;
  int64_t off;
  int descending, nulls_at_start;
  const unsigned* rows_in;  // the order produced by the less significant sort keys; NULL = input order
  __device__ __forceinline__ void load(int64_t i, unsigned long long* key, unsigned* row, unsigned* digit) const {
    const unsigned r = rows_in ? rows_in[i] : (unsigned)i;
    const T v = values[r];
    int cat = kCatRest;
    if (!ah_bit(valid, off + r)) cat = kCatNull;
    else if (std::is_floating_point<T>::value && v != v) cat = kCatNaN;
    *key = cat == kCatRest ? make_key<T>(v, descending) : 0ull;
    *row = r;
    *digit = nulls_at_start ? (unsigned)(2 - cat) : (unsigned)cat;  // [nulls, NaNs, rest] or [rest, NaNs, nulls]
  }
};
struct Pairs {
  const unsigned long long* keys;
  const unsigned* rows;
  int shift;
  __device__ __forceinline__ void load(int64_t i, unsigned long long* key, unsigned* row, unsigned* digit) const {
    *key = keys[i];
    *row = rows[i];
    *digit = (unsigned)(*key >> shift) & 255u;
  }
};

// group-by helper: (value bits, group id) pairs, digit = a byte of the group id
struct IdVal {
  const int32_t* ids;
  const unsigned long long* vals;
  const uint8_t* vvalid;
  int64_t voff;
  int shift;
  __device__ __forceinline__ void load(int64_t i, unsigned long long* key, unsigned* row, unsigned* digit) const {
    const unsigned g = (unsigned)ids[i];
    const bool ok = ah_bit(vvalid, voff + i);
    *key = ok ? vals[i] : 0ull;
    *row = ok ? g : (g | 0x80000000u);
    *digit = (g >> shift) & 255u;
  }
};

// second pass of the group-by partition: pairs already carry the null flag in bit 31 of the id
struct ValId {
  const unsigned long long* vals;
  const unsigned* ids;
  int shift;
  __device__ __forceinline__ void load(int64_t i, unsigned long long* key, unsigned* row, unsigned* digit) const {
    *key = vals[i];
    *row = ids[i];
    *digit = ((*row & 0x7fffffffu) >> shift) & 255u;
  }
};

// block histogram → hist[digit * nblocks + block]; a block = 1 or 8 consecutive tiles handled by one
// workgroup (one histogram row per 16 Ki rows: 8× fewer scattered 4-byte writes and an 8× smaller scan)
template <typename SRC>
__global__ __launch_bounds__(kBlock) void hist_kernel(SRC src, int64_t n, unsigned* __restrict__ hist, int64_t nblocks, int tpb) {
  __shared__ unsigned s_h[kWaves][kRadix];
  for (int i = threadIdx.x; i < kWaves * kRadix; i += kBlock) (&s_h[0][0])[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t block = blockIdx.x;
  for (int t = 0; t < tpb; t++) {
    const int64_t wbase = (block * tpb + t) * kTile + (int64_t)wave * (kItems * 64);
    if (wbase >= n) break;
#pragma unroll
    for (int r = 0; r < kItems; r++) {
      const int64_t i = wbase + r * 64 + lane;
      if (i < n) {
        unsigned long long key; unsigned row, digit;
        src.load(i, &key, &row, &digit);
        atomicAdd(&s_h[wave][digit], 1u);
      }
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < kRadix; d += kBlock) {
    unsigned t = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) t += s_h[w][d];
    hist[(int64_t)d * nblocks + block] = t;
  }
}

// stable scatter: offs = INCLUSIVE scan of hist (digit-major)
template <typename SRC>
__global__ __launch_bounds__(kBlock) void scatter_kernel(SRC src, int64_t n, const unsigned* __restrict__ hist, const unsigned* __restrict__ offs,
                                                          int64_t nblocks, int tpb, unsigned long long* __restrict__ out_keys,
                                                          unsigned* __restrict__ out_rows) {
  __shared__ unsigned s_cnt[kWaves][kRadix];   // per wave: rows of each digit seen in earlier rounds; later: wave bases
  __shared__ unsigned s_start[kRadix], s_goff[kRadix], s_wsum[kWaves];
  __shared__ unsigned long long s_keys[kTile];
  __shared__ unsigned s_rows[kTile];
  __shared__ uint8_t s_dig[kTile];
  static_assert(kBlock == kRadix, "one thread per digit in the prefix step");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // thread d carries the global position of the block's next row of digit d (inclusive scan − own count)
  unsigned run_d = offs[(int64_t)threadIdx.x * nblocks + blockIdx.x] - hist[(int64_t)threadIdx.x * nblocks + blockIdx.x];
  for (int tb = 0; tb < tpb; tb++) {
  const int64_t tile = (int64_t)blockIdx.x * tpb + tb;
  if (tile * kTile >= n) break;  // workgroup-uniform
  for (int i = threadIdx.x; i < kWaves * kRadix; i += kBlock) (&s_cnt[0][0])[i] = 0;
  __syncthreads();
  const int64_t wbase = tile * kTile + (int64_t)wave * (kItems * 64);
  unsigned long long key[kItems];
  unsigned row[kItems], digit[kItems], rank[kItems];
  bool live[kItems];
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    const int64_t i = wbase + r * 64 + lane;
    live[r] = i < n;
    key[r] = 0; row[r] = 0; digit[r] = 0;
    if (live[r]) src.load(i, &key[r], &row[r], &digit[r]);
  }
  // in-wave ranks, round by round: element order inside the tile is (wave, round, lane)
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    // match-any on the 8-bit digit: peers = lanes holding the same digit
    unsigned long long peers = __ballot(live[r]);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const unsigned long long bal = __ballot((digit[r] >> b) & 1u);
      peers &= ((digit[r] >> b) & 1u) ? bal : ~bal;
    }
    if (live[r]) {
      // atomic accesses: ANOTHER lane of this wave advanced the counter in the previous round — a value
      // the per-thread memory model would otherwise let the compiler keep in a register
      unsigned* cnt = &s_cnt[wave][digit[r]];
      const unsigned before = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // every peer reads the same counter …
      rank[r] = before + (unsigned)__popcll(peers & below);
      if ((peers & below) == 0)  // … the lowest one advances it
        __hip_atomic_store(cnt, before + (unsigned)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    // LDS operations of one wave execute in order; keep the compiler from moving them across rounds
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // per digit: exclusive prefix over the waves, the digit's start inside the tile (exclusive scan over
  // the 256 digit totals), and the offset that turns a tile-local sorted position into the global one
  unsigned tot = 0;  // thread d < 256 owns digit d (kBlock == kRadix)
  {
    const int d = threadIdx.x;
#pragma unroll
    for (int w = 0; w < kWaves; w++) {
      const unsigned t = s_cnt[w][d];
      s_cnt[w][d] = tot;
      tot += t;
    }
    unsigned inc = tot;  // inclusive scan of the digit totals: shuffles inside a wave, LDS across waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    unsigned wbase = 0;
#pragma unroll
    for (int w = 0; w < kWaves; w++) if (w < wave) wbase += s_wsum[w];
    const unsigned start = wbase + inc - tot;                       // first tile-local position of digit d
    s_start[d] = start;
    s_goff[d] = run_d - start;     // global = s_goff[d] + local
    run_d += tot;
  }
  __syncthreads();
  // stage the tile in digit order in LDS …
#pragma unroll
  for (int r = 0; r < kItems; r++) {
    if (live[r]) {
      const unsigned lp = s_start[digit[r]] + s_cnt[wave][digit[r]] + rank[r];
      s_keys[lp] = key[r];
      s_rows[lp] = row[r];
      s_dig[lp] = (uint8_t)digit[r];
    }
  }
  __syncthreads();
  // … and write it out: consecutive threads hold consecutive positions of the same digit run, so the
  // stores to HBM are runs of neighbouring addresses instead of one 8-byte store per bucket
  const int64_t tile_n = n - tile * kTile >= kTile ? kTile : n - tile * kTile;
  for (int lp = threadIdx.x; lp < tile_n; lp += kBlock) {
    const unsigned pos = s_goff[s_dig[lp]] + (unsigned)lp;
    out_keys[pos] = s_keys[lp];
    out_rows[pos] = s_rows[lp];
  }
  }  // tiles of this block
}

// which key bits vary at all, and the extreme keys: per workgroup {AND, OR, min, max} of its keys (the host folds the ≤ 1024
// partial results — four same-address atomics per wave would cost more than the pass)
__global__ __launch_bounds__(kBlock) void and_or_kernel(const unsigned long long* __restrict__ keys, int64_t n, unsigned long long* __restrict__ res) {
  __shared__ unsigned long long s_r[kWaves][4];
  unsigned long long a = ~0ull, o = 0ull, mn = ~0ull, mx = 0ull;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const unsigned long long k = keys[i];
    a &= k; o |= k;
    mn = k < mn ? k : mn;
    mx = k > mx ? k : mx;
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    a &= __shfl_down(a, s, 64);
    o |= __shfl_down(o, s, 64);
    const unsigned long long m1 = __shfl_down(mn, s, 64), m2 = __shfl_down(mx, s, 64);
    mn = m1 < mn ? m1 : mn;
    mx = m2 > mx ? m2 : mx;
  }
  if ((threadIdx.x & 63) == 0) { s_r[threadIdx.x >> 6][0] = a; s_r[threadIdx.x >> 6][1] = o; s_r[threadIdx.x >> 6][2] = mn; s_r[threadIdx.x >> 6][3] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kWaves; w++) {
      a &= s_r[w][0]; o |= s_r[w][1];
      mn = s_r[w][2] < mn ? s_r[w][2] : mn;
      mx = s_r[w][3] > mx ? s_r[w][3] : mx;
    }
    res[blockIdx.x * 4 + 0] = a; res[blockIdx.x * 4 + 1] = o; res[blockIdx.x * 4 + 2] = mn; res[blockIdx.x * 4 + 3] = mx;
  }
}

// Pass (1) without the partition, for a column with no validity bitmap in input order: keys and row numbers in one streaming
// pass, the AND / OR / min / max of the keys on the way (per workgroup: res[5 b …] = {AND, OR, min, max, NaN count}).  Valid
// only if the NaN count comes back zero — then every row is `rest` and pass (2) has been done too; otherwise the caller
// runs the partition pass as usual.
template <typename T>
__global__ __launch_bounds__(kBlock) void pairs_kernel(const T* __restrict__ values, int64_t n, int descending, unsigned long long* __restrict__ keys,
                                                        unsigned* __restrict__ rows, unsigned long long* __restrict__ res) {
  __shared__ unsigned long long s_r[kWaves][5];
  unsigned long long a = ~0ull, o = 0ull, mn = ~0ull, mx = 0ull, nans = 0;
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * kBlock * U;
  for (int64_t base = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; base < n; base += stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int64_t i = base + (int64_t)u * kBlock; v[u] = i < n ? __builtin_nontemporal_load(&values[i]) : (T)0; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + (int64_t)u * kBlock;
      if (i >= n) continue;
      if (std::is_floating_point<T>::value && v[u] != v[u]) { nans++; continue; }
      const unsigned long long k = make_key<T>(v[u], descending);
      __builtin_nontemporal_store(k, &keys[i]);
      __builtin_nontemporal_store((unsigned)i, &rows[i]);
      a &= k; o |= k;
      mn = k < mn ? k : mn;
      mx = k > mx ? k : mx;
    }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    a &= __shfl_down(a, s, 64);
    o |= __shfl_down(o, s, 64);
    const unsigned long long m1 = __shfl_down(mn, s, 64), m2 = __shfl_down(mx, s, 64);
    mn = m1 < mn ? m1 : mn;
    mx = m2 > mx ? m2 : mx;
    nans += __shfl_down(nans, s, 64);
  }
  if ((threadIdx.x & 63) == 0) { unsigned long long* r = s_r[threadIdx.x >> 6]; r[0] = a; r[1] = o; r[2] = mn; r[3] = mx; r[4] = nans; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kWaves; w++) {
      a &= s_r[w][0]; o |= s_r[w][1];
      mn = s_r[w][2] < mn ? s_r[w][2] : mn;
      mx = s_r[w][3] > mx ? s_r[w][3] : mx;
      nans += s_r[w][4];
    }
    unsigned long long* r = res + (size_t)blockIdx.x * 5;
    r[0] = a; r[1] = o; r[2] = mn; r[3] = mx; r[4] = nans;
  }
}

__global__ __launch_bounds__(kBlock) void emit_kernel(const unsigned* __restrict__ rows, int64_t n, uint64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = rows[i];
}

// temporaries of one call, carved out of the context's temp arena (the scratch arena is used by the scan this calls)
struct Carver {
  uint8_t* base;
  size_t used = 0;
  static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
  template <typename P>
  void take(size_t bytes, P** out) { *out = (P*)(base + used); used += pad(bytes); }
};

template <typename SRC>
int radix_pass(ah_ctx* c, SRC src, int64_t n, unsigned* hist, unsigned* offs, unsigned long long* out_keys, unsigned* out_rows) {
  const int tpb = tiles_per_block(n);
  const int64_t nblocks = ah_ceil_div(n, (int64_t)kTile * tpb);
  hist_kernel<SRC><<<(unsigned)nblocks, kBlock, 0, c->stream>>>(src, n, hist, nblocks, tpb);
  AH_LAUNCH_CHECK(c);
  int rc = ah_cumulative_sum(c, AH_UINT32, hist, nullptr, 0, (int64_t)kRadix * nblocks, nullptr, 0, 0, offs, nullptr, nullptr);
  if (rc != AH_OK) return rc;
  scatter_kernel<SRC><<<(unsigned)nblocks, kBlock, 0, c->stream>>>(src, n, hist, offs, nblocks, tpb, out_keys, out_rows);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

struct SortBuffers {  // temporaries shared by all keys of one call
  void* msd_tmp = nullptr;  // ah_sort_msd.hip's tables (nullptr: that path is off for this call)
  uint64_t* final_out = nullptr;  // single-key call: where the widened result goes; the MSD path writes it directly when every row is `rest`
  bool emitted = false;
  unsigned long long *ka, *kb, *andor;
  unsigned *ra, *rb, *rc;  // ra / rb: ping-pong of a key's passes; rc: the previous key's result
  unsigned *hist, *offs;
};

// Stable sort of the current row order (rows_in, or the input order) by ONE column.  Leaves the new
// order in b.ra; returns through *result which buffer holds it (always b.ra).
template <typename T>
int sort_by_column(ah_ctx* c, SortBuffers& b, const void* values, const uint8_t* valid, int64_t off, int64_t n, int descending,
                   int nulls_at_start, const unsigned* rows_in) {
  const int64_t ntiles = ah_ceil_div(n, (int64_t)kTile * tiles_per_block(n));  // histogram rows ("blocks")
  int rc;
  // (1) partition by category, keys and row numbers come into being
  Column<T> col{(const T*)values, valid, off, descending, nulls_at_start, rows_in};
  int64_t rest_lo = 0, rest_n = 0;
  unsigned long long k_and = ~0ull, k_or = 0, kmin = ~0ull, kmax = 0;
  bool have_stats = false;
  if (valid == nullptr && rows_in == nullptr && n >= ((int64_t)1 << 20)) {
    // no validity bitmap, input order: nothing to partition unless a float column holds NaNs — one streaming pass, checked after
    // b.andor holds kAndOrGrid × 8 words (sort_buffers): the grid is clamped to that whatever ARROWHIP_BLOCKS_PER_CU asks for
    const unsigned pgrid = std::min(ah_stream_grid(c, ah_ceil_div(n, (int64_t)kBlock * 4), 8), (unsigned)kAndOrGrid);
    pairs_kernel<T><<<pgrid, kBlock, 0, c->stream>>>((const T*)values, n, descending, b.ka, b.ra, b.andor);
    AH_LAUNCH_CHECK(c);
    std::vector<unsigned long long> parts((size_t)pgrid * 5);
    AH_HIP(c, hipMemcpyAsync(parts.data(), b.andor, parts.size() * 8, hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipStreamSynchronize(c->stream));
    unsigned long long nans = 0;
    for (unsigned g = 0; g < pgrid; g++) {
      k_and &= parts[g * 5]; k_or |= parts[g * 5 + 1];
      kmin = parts[g * 5 + 2] < kmin ? parts[g * 5 + 2] : kmin;
      kmax = parts[g * 5 + 3] > kmax ? parts[g * 5 + 3] : kmax;
      nans += parts[g * 5 + 4];
    }
    if (nans == 0) { have_stats = true; rest_lo = 0; rest_n = n; }
  }
  if (!have_stats) {
    if ((rc = radix_pass(c, col, n, b.hist, b.offs, b.ka, b.ra)) != AH_OK) return rc;
    // how many rows of each category?  offs (inclusive scan, digit-major): the last tile's entry of digit d
    unsigned ends[3];
    for (int d = 0; d < 3; d++)
      AH_HIP(c, hipMemcpyAsync(&c->pinned[d], b.offs + ((int64_t)d + 1) * ntiles - 1, 4, hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipStreamSynchronize(c->stream));
    for (int d = 0; d < 3; d++) ends[d] = *(volatile unsigned*)&c->pinned[d];
    // the `rest` category is digit 0 (nulls at end) or digit 2 (nulls at start)
    rest_lo = nulls_at_start ? ends[1] : 0;
    rest_n = nulls_at_start ? (int64_t)ends[2] - ends[1] : ends[0];
  }
  unsigned long long *kcur = b.ka + rest_lo, *kalt = b.kb + rest_lo;
  unsigned *rcur = b.ra + rest_lo, *ralt = b.rb + rest_lo;
  if (rest_n > 1) {
    if (!have_stats) {
      // (2) which key bytes vary; smallest and largest key
      const unsigned agrid = std::min(ah_stream_grid(c, ah_ceil_div(rest_n, kBlock), 4), (unsigned)kAndOrGrid);
      and_or_kernel<<<agrid, kBlock, 0, c->stream>>>(kcur, rest_n, b.andor);
      AH_LAUNCH_CHECK(c);
      std::vector<unsigned long long> parts((size_t)agrid * 4);
      AH_HIP(c, hipMemcpyAsync(parts.data(), b.andor, parts.size() * 8, hipMemcpyDeviceToHost, c->stream));
      AH_HIP(c, hipStreamSynchronize(c->stream));
      for (unsigned g = 0; g < agrid; g++) {
        k_and &= parts[g * 4]; k_or |= parts[g * 4 + 1];
        kmin = parts[g * 4 + 2] < kmin ? parts[g * 4 + 2] : kmin;
        kmax = parts[g * 4 + 3] > kmax ? parts[g * 4 + 3] : kmax;
      }
    }
    const unsigned long long varying = k_and ^ k_or;
    int nvar = 0;
    for (int by = 0; by < (int)sizeof(T); by++) nvar += ((varying >> (8 * by)) & 0xFFull) != 0;
    // (3') large inputs with ≥ 4 varying bytes, in input order (the first key processed): two unstable MSD partition passes
    // + one wave per bucket comparing (key, row) — ah_sort_msd.hip.  Equal keys end up in row order = input order.
    bool done = false;
    if (rows_in == nullptr && nvar >= 4 && c->opt_sort_msd) {
      int used = -1;
      if ((rc = ah_sort_rest_msd(c, kcur, rcur, kalt, ralt, rest_n, varying, kmin, kmax, std::is_floating_point<T>::value ? (int)sizeof(T) : 0, descending, b.msd_tmp,
                                 (unsigned long long*)(rest_n == n ? b.final_out : nullptr), &used)) != AH_OK) return rc;
      if (used > 0) {
        done = true;   // sorted rows are in rcur (each bucket is rewritten in place) — or already widened in the output
        b.emitted = rest_n == n && b.final_out != nullptr;
      } else if (used == 0 && b.msd_tmp && ah_sort_msd_temp_bytes(rest_n) != 0) {
        // it ran and gave up (a bucket too large): the pairs are gone — pass (1) again, then the LSD passes
        if ((rc = radix_pass(c, col, n, b.hist, b.offs, b.ka, b.ra)) != AH_OK) return rc;
      }
    }
    // (3) one stable pass per varying byte, least significant first
    for (int by = 0; !done && by < (int)sizeof(T); by++) {
      if (((varying >> (8 * by)) & 0xFFull) == 0) continue;
      Pairs src{kcur, rcur, 8 * by};
      if ((rc = radix_pass(c, src, rest_n, b.hist, b.offs, kalt, ralt)) != AH_OK) return rc;
      unsigned long long* tk = kcur; kcur = kalt; kalt = tk;
      unsigned* tr = rcur; rcur = ralt; ralt = tr;
    }
    // the rest range must end up in ra, next to the NaN / null groups pass (1) left there
    if (!b.emitted && rcur != b.ra + rest_lo) AH_HIP(c, hipMemcpyAsync(b.ra + rest_lo, rcur, (size_t)rest_n * 4, hipMemcpyDeviceToDevice, c->stream));
  }
  return AH_OK;
}

int sort_dispatch(ah_ctx* c, SortBuffers& b, int type, const void* values, const uint8_t* valid, int64_t off, int64_t n, int descending,
                  int nulls_at_start, const unsigned* rows_in) {
  switch (type) {
#define AH_SORT(ID, T) case ID: return sort_by_column<T>(c, b, values, valid, off, n, descending, nulls_at_start, rows_in);
    AH_SORT(AH_UINT8, uint8_t) AH_SORT(AH_INT8, int8_t) AH_SORT(AH_UINT16, uint16_t) AH_SORT(AH_INT16, int16_t)
    AH_SORT(AH_UINT32, uint32_t) AH_SORT(AH_INT32, int32_t) AH_SORT(AH_UINT64, uint64_t) AH_SORT(AH_INT64, int64_t)
    AH_SORT(AH_FLOAT32, float) AH_SORT(AH_FLOAT64, double)
#undef AH_SORT
  }
  return ah_fail(c, AH_ENOTIMPL, "sorting not supported for type %d", type);  // vector_sort.go:266-268
}

int sort_keys(ah_ctx* c, int nkeys, const int* types, const void* const* values, const uint8_t* const* valids, const int64_t* offs, int64_t n,
              const int* descending, const int* nulls_at_start, uint64_t* out) {
  SortBuffers b;
  const int64_t ntiles = ah_ceil_div(n, kTile);
  int rc;
  const size_t hist_bytes = (size_t)kRadix * ntiles * 4;
  const size_t msd_bytes = c->opt_sort_msd ? ah_sort_msd_temp_bytes(n) : 0;   // sized for rest_n = n
  const size_t total = 2 * Carver::pad((size_t)n * 8) + (nkeys > 1 ? 3 : 2) * Carver::pad((size_t)n * 4) + 2 * Carver::pad(hist_bytes) + Carver::pad(4096 * 8 * 8) + Carver::pad(msd_bytes);
  void* arena;
  if ((rc = ah_temp_reserve(c, total, &arena)) != AH_OK) return rc;
  Carver tmp{(uint8_t*)arena};
  tmp.take((size_t)n * 8, &b.ka);
  tmp.take((size_t)n * 8, &b.kb);
  tmp.take((size_t)n * 4, &b.ra);
  tmp.take((size_t)n * 4, &b.rb);
  b.rc = nullptr;
  if (nkeys > 1) tmp.take((size_t)n * 4, &b.rc);
  tmp.take(hist_bytes, &b.hist);
  tmp.take(hist_bytes, &b.offs);
  tmp.take((size_t)kAndOrGrid * 8 * 8, &b.andor);   // ≤ kAndOrGrid workgroups × {AND, OR, min, max, NaNs} (8 words reserved each)
  if (msd_bytes) { uint8_t* m; tmp.take(msd_bytes, &m); b.msd_tmp = m; }
  // lexicographic order by keys 0..k−1 = stable sorts by key k−1, …, key 0 in turn (every pass is stable)
  const unsigned* rows_in = nullptr;
  if (nkeys == 1) b.final_out = out;
  for (int k = nkeys - 1; k >= 0; k--) {
    if ((rc = sort_dispatch(c, b, types[k], values[k], valids[k], offs[k], n, descending[k], nulls_at_start[k], rows_in)) != AH_OK) return rc;
    if (k > 0) {
      AH_HIP(c, hipMemcpyAsync(b.rc, b.ra, (size_t)n * 4, hipMemcpyDeviceToDevice, c->stream));
      rows_in = b.rc;
    }
  }
  if (!b.emitted) {
    emit_kernel<<<ah_stream_grid(c, ah_ceil_div(n, kBlock), 8), kBlock, 0, c->stream>>>(b.ra, n, out);
    AH_LAUNCH_CHECK(c);
  }
  return AH_OK;
}

}  // namespace

int ah_partition_by_group(ah_ctx* c, const int32_t* ids, const unsigned long long* vals, const uint8_t* vvalid, int64_t voff, int64_t n, int shift,
                          int passes, unsigned* hist, unsigned* offs, unsigned long long* alt_vals, unsigned* alt_ids,
                          unsigned long long* out_vals, unsigned* out_ids) {
  IdVal src{ids, vals, vvalid, voff, shift};
  if (passes == 1) return radix_pass(c, src, n, hist, offs, out_vals, out_ids);
  int rc = radix_pass(c, src, n, hist, offs, alt_vals, alt_ids);
  if (rc != AH_OK) return rc;
  ValId src2{alt_vals, alt_ids, shift + 8};
  return radix_pass(c, src2, n, hist, offs, out_vals, out_ids);
}

int ah_sort_pairs_lsd(ah_ctx* c, unsigned long long* keys, unsigned* rows, unsigned long long* alt_keys, unsigned* alt_rows, int64_t n,
                      unsigned* hist, unsigned* offs, unsigned long long** sorted_keys) {
  for (int by = 0; by < 8; by++) {
    Pairs src{keys, rows, 8 * by};
    int rc = radix_pass(c, src, n, hist, offs, alt_keys, alt_rows);
    if (rc != AH_OK) return rc;
    unsigned long long* tk = keys; keys = alt_keys; alt_keys = tk;
    unsigned* tr = rows; rows = alt_rows; alt_rows = tr;
  }
  *sorted_keys = keys;
  return AH_OK;
}

static int check_column(ah_ctx* c, int type, const void* values) {
  const int w = ah_type_width(type);
  if (!w) return ah_fail(c, AH_ENOTIMPL, "sorting not supported for type %d", type);  // vector_sort.go:266-268
  if (!values) return ah_fail(c, AH_EINVALID, "sort_indices: null buffer");
  if ((uintptr_t)values & (uintptr_t)(w - 1)) return ah_fail(c, AH_EINVALID, "sort_indices: buffer not element-aligned");
  return AH_OK;
}

AH_EXPORT int ah_sort_indices(ah_ctx* c, int type, const void* values, const uint8_t* valid, int64_t off, int64_t n, int descending,
                              int nulls_at_start, uint64_t* out_indices) {
  return ah_sort_indices_multi(c, 1, &type, &values, &valid, &off, n, &descending, &nulls_at_start, out_indices);
}

AH_EXPORT int ah_sort_indices_multi(ah_ctx* c, int nkeys, const int* types, const void* const* values, const uint8_t* const* valids,
                                    const int64_t* offs, int64_t n, const int* descending, const int* nulls_at_start, uint64_t* out_indices) {
  AH_ENTER(c);
  if (nkeys < 1) return ah_fail(c, AH_EINVALID, "must provide at least one sort key");  // compute/vector_sort.go:119-121
  if (n < 0) return ah_fail(c, AH_EINVALID, "sort_indices: negative length");
  for (int k = 0; k < nkeys; k++)
    if (offs[k] < 0) return ah_fail(c, AH_EINVALID, "sort_indices: negative offset");
  if (n == 0) return AH_OK;
  if (!out_indices) return ah_fail(c, AH_EINVALID, "sort_indices: null buffer");
  if (n >= ((int64_t)1 << 32)) return ah_fail(c, AH_ENOTIMPL, "sort_indices: more than 2^32 - 1 rows in one batch");
  for (int k = 0; k < nkeys; k++) {
    int rc = check_column(c, types[k], values[k]);
    if (rc != AH_OK) return rc;
  }
  return sort_keys(c, nkeys, types, values, valids, offs, n, descending, nulls_at_start, out_indices);
}
