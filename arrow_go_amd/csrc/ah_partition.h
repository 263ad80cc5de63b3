// ah_partition.h — "cut the rows by a hash of the key" as shared by the partition-first group-by (ah_groupby.hip) and the
// partition-first unique / dictionary_encode (ah_hash_part.hip): the partition function, the distinct-count sample, the per
// (tile, partition) histogram and the LDS-staged scatter of {key[, value], row | flags} records (tile bookkeeping: ah_bins.h).
#pragma once
#include "ah_common.h"
#include "ah_hashing.h"
#include "ah_bins.h"

namespace {

constexpr int kGbTile = 4096;                       // rows per tile of the hist / scatter passes
constexpr int kGbRows = kGbTile / kThreads;          // 4 per thread in the scatter
constexpr int kGbHistThreads = 256, kGbHistRows = kGbTile / kGbHistThreads;
constexpr unsigned kKeyNull = 0x80000000u, kValNull = 0x40000000u, kRowMask = 0x1fffffffu;
constexpr int64_t kMaxRows = (int64_t)1 << 29;

// Any well-mixed hash will do: results depend on key EQUALITY and row order only.  (hashInt's low bits are fine, but
// its bits above 12 depend on ever fewer key bits, and the partition number must not.)
__device__ __forceinline__ uint64_t gb_mix(uint64_t k) {
  uint64_t x = k * 0x9E3779B97F4A7C15ull;
  x ^= x >> 32;
  x *= 0xD6E8FEB86659FD93ull;
  return x ^ (x >> 32);
}
__device__ __forceinline__ unsigned gb_part(uint64_t m, int lp) { return (unsigned)(m >> (64 - lp)); }   // lp = 3 … 10

// ---- 0: distinct estimate by linear counting over a strided sample ---------------------------------------------------
// Sample group g = 64 consecutive rows at g · stride; the launch covers groups g0, g0 + 2, g0 + 4, … (the host runs the even
// groups, reads the bitmap's popcount, then the odd ones: two points of the distinct-count curve).  A 1024-entry filter
// in LDS drops the keys this workgroup has just seen: without it a hot key sends every sampled row to ONE bitmap word —
// 2^21 same-address atomics = 0.65 ms, and still 0.49 ms with a look-before-set on a Zipf column.
__global__ __launch_bounds__(1024) void gb_sample_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ kvalid, int64_t koff,
                                                          int64_t n, int64_t ngroups, int64_t stride, int g0, int gstep, unsigned* __restrict__ bm, unsigned mmask) {
  __shared__ unsigned long long s_seen[1024];
  s_seen[threadIdx.x] = 0;
  __syncthreads();
  const int64_t g = ((int64_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * gstep + g0;
  const int64_t i = g * stride + (threadIdx.x & 63);
  if (g >= ngroups || i >= n || !ah_bit(kvalid, koff + i)) return;
  const uint64_t m = gb_mix(keys[i]) | 1ull;
  const unsigned f = (unsigned)(m >> 44) & 1023u;
  if (atomicExch(&s_seen[f], m) == m) return;   // an LDS atomic, so that of the lanes holding a hot key at this instant only one goes on
  const unsigned b = (unsigned)(m >> 20) & mmask;
  if (!((__hip_atomic_load(&bm[b >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (b & 31)) & 1u)) atomicOr(&bm[b >> 5], 1u << (b & 31));
}

// ---- 1: per (tile, partition) counts ----------------------------------------------------------------------------------
__global__ __launch_bounds__(kGbHistThreads) void gb_hist_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ kvalid, int64_t koff,
                                                                  int64_t n, int lp, int nb, int64_t ntiles, unsigned* __restrict__ cnt_tm) {
  __shared__ unsigned s_cnt[kMaxBins];
  const int64_t tile = xcd_contiguous_tile(ntiles);
  if (tile < 0) return;
  for (int b = threadIdx.x; b < nb; b += kGbHistThreads) s_cnt[b] = 0;
  __syncthreads();
  const int64_t base = tile * kGbTile;
  unsigned long long k[kGbHistRows];
  if (base + kGbTile <= n) {   // workgroup-uniform: a full tile's loads go out back to back, no per-lane guard between them
#pragma unroll
    for (int u = 0; u < kGbHistRows; u++) k[u] = __builtin_nontemporal_load(&keys[base + u * kGbHistThreads + threadIdx.x]);
  } else {
#pragma unroll
    for (int u = 0; u < kGbHistRows; u++) {
      const int64_t i = base + u * kGbHistThreads + threadIdx.x;
      k[u] = i < n ? __builtin_nontemporal_load(&keys[i]) : 0ull;
    }
  }
#pragma unroll
  for (int u = 0; u < kGbHistRows; u++) {
    const int64_t i = base + u * kGbHistThreads + threadIdx.x;
    const int64_t w0 = base + u * kGbHistThreads + (threadIdx.x & ~63);   // the wave's first row of this step
    // the wave's 64 validity bits in one scalar load; null keys all belong to partition 0 and are counted once per wave —
    // a tenth of a tile's rows on one LDS counter would be served a lane at a time
    const unsigned long long vw = ah_wave_bits64(kvalid, koff + w0, n - w0);   // 0 past the end
    const unsigned long long in = n - w0 >= 64 ? ~0ull : (n > w0 ? (1ull << (n - w0)) - 1ull : 0ull);
    const unsigned long long nulls = in & ~vw;
    if (i >= n) continue;
    if ((vw >> (threadIdx.x & 63)) & 1ull) atomicAdd(&s_cnt[gb_part(gb_mix(k[u]), lp)], 1u);
    else if ((nulls & ((1ull << (threadIdx.x & 63)) - 1ull)) == 0) atomicAdd(&s_cnt[0], (unsigned)__popcll(nulls));
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kGbHistThreads) cnt_tm[tile * nb + b] = s_cnt[b];
}

// ---- 2: records in partition order ------------------------------------------------------------------------------------
// HAS_VALS = false: records are {key, row | null flag} only (unique / dictionary_encode, ah_hash_part.hip)
template <bool HAS_VALS>
__global__ __launch_bounds__(kThreads) void gb_scatter_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ kvalid, int64_t koff,
                                                               const unsigned long long* __restrict__ vals, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                               int64_t n, int lp, int nb, int64_t ntiles, const unsigned* __restrict__ toffs,
                                                               unsigned long long* __restrict__ pkeys, unsigned long long* __restrict__ pvals,
                                                               unsigned* __restrict__ prows, unsigned long long* __restrict__ tile_max) {
  __shared__ unsigned s_cnt[kMaxBins], s_start[kMaxBins], s_goff[kMaxBins], s_wsum[kThreads / 64];
  __shared__ unsigned long long s_stage[kGbTile];
  __shared__ uint16_t s_bin[kGbTile];
  __shared__ unsigned long long s_max[kThreads / 64];
  __shared__ unsigned s_imin[kThreads / 64];
  // consecutive tiles on ONE XCD: the runs they append to a partition meet in that XCD's L2 and leave as whole lines
  const int64_t tile = xcd_contiguous_tile(ntiles);
  if (tile < 0) return;
  s_cnt[threadIdx.x] = 0;
  const int64_t base = tile * kGbTile;
  unsigned long long k[kGbRows], v[kGbRows];
  unsigned rw[kGbRows], bin[kGbRows], rank[kGbRows];
  bool live[kGbRows];
#pragma unroll
  for (int u = 0; u < kGbRows; u++) {
    const int64_t i = base + u * kThreads + threadIdx.x;
    live[u] = i < n;
    k[u] = live[u] ? __builtin_nontemporal_load(&keys[i]) : 0ull;
    v[u] = (HAS_VALS && live[u]) ? __builtin_nontemporal_load(&vals[i]) : 0ull;
  }
  unsigned goff_excl = 0;
  if ((int)threadIdx.x < nb) goff_excl = toffs[tile * nb + threadIdx.x];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kGbRows; u++) {
    const int64_t i = base + u * kThreads + threadIdx.x;
    rw[u] = 0; bin[u] = 0; rank[u] = 0;
    // the wave's validity words by scalar loads (0 past the end)
    const int64_t w0 = base + u * kThreads + (threadIdx.x & ~63);
    const unsigned long long kw64 = ah_wave_bits64(kvalid, koff + w0, n - w0), vw64 = HAS_VALS ? ah_wave_bits64(vvalid, voff + w0, n - w0) : ~0ull;
    const bool kv = (kw64 >> (threadIdx.x & 63)) & 1ull, vv = (vw64 >> (threadIdx.x & 63)) & 1ull;
    // null keys (partition 0) take their ranks from ONE counter update per wave: same-address LDS atomics are served a lane at a time
    const unsigned long long nulls = __ballot(live[u] && !kv);
    if (live[u]) {
      k[u] = kv ? k[u] : 0ull;   // one key for all null rows: the aggregate pass adds consecutive rows of one key in registers
      bin[u] = kv ? gb_part(gb_mix(k[u]), lp) : 0u;
      rw[u] = (unsigned)i | (kv ? 0u : kKeyNull) | (vv ? 0u : kValNull);
      if (kv) rank[u] = atomicAdd(&s_cnt[bin[u]], 1u);
    }
    if (nulls) {   // wave-uniform
      const int leader = __builtin_ctzll(nulls);
      unsigned first = 0;
      if ((int)(threadIdx.x & 63) == leader) first = atomicAdd(&s_cnt[0], (unsigned)__popcll(nulls));
      first = __shfl(first, leader, 64);
      if (live[u] && !kv) rank[u] = first + (unsigned)__popcll(nulls & ((1ull << (threadIdx.x & 63)) - 1ull));
    }
  }
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  if ((int)threadIdx.x < nb) s_goff[threadIdx.x] = goff_excl - s_start[threadIdx.x];
  const int tile_n = n - base >= kGbTile ? kGbTile : (int)(n - base);
  // three rounds through one staging buffer: keys, value bits, row words
#pragma unroll
  for (int u = 0; u < kGbRows; u++)
    if (live[u]) { const unsigned q = s_start[bin[u]] + rank[u]; s_stage[q] = k[u]; s_bin[q] = (uint16_t)bin[u]; }
  __syncthreads();
  int64_t dst[kGbRows];
#pragma unroll
  for (int u = 0; u < kGbRows; u++) {
    const int q = u * kThreads + threadIdx.x;
    dst[u] = q < tile_n ? (int64_t)s_goff[s_bin[q]] + q : -1;
    if (dst[u] >= 0) pkeys[dst[u]] = s_stage[q];   // PLAIN stores: a partition's short runs (32 bytes at 1024 partitions) from consecutive tiles meet in this XCD's L2 and
                                                   // leave as whole lines — with nontemporal hints the pass ran 2× slower at 2^20 groups (2^26 rows: 1.46 → 3.1 ms per call)
  }
  __syncthreads();
  if constexpr (HAS_VALS) {
  // the tile's value range (ah_hashing.h) is taken HERE, where the keys have left the registers: in the ranking loop above the two
  // running extremes cost the second workgroup per CU (66 VGPRs where 64 is the limit: the pass went 0.44 → 0.64 ms)
  unsigned long long vmax = 0;
  unsigned vimin = 0;
#pragma unroll
  for (int u = 0; u < kGbRows; u++) {
    if (live[u]) s_stage[s_start[bin[u]] + rank[u]] = v[u];
    const unsigned long long a = v[u] & 0x7fffffffffffffffull;   // |x| of finite doubles order like their bit patterns
    if (tile_max && live[u] && !(rw[u] & kValNull) && (a >> 52) != 0x7ff && a != 0) {
      vmax = a > vmax ? a : vmax;
      vimin = fx_inv_exp(a) > vimin ? fx_inv_exp(a) : vimin;
    }
  }
  if (tile_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long x = __shfl_down(vmax, o, 64);
      const unsigned xi = __shfl_down(vimin, o, 64);
      vmax = x > vmax ? x : vmax;
      vimin = xi > vimin ? xi : vimin;
    }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = vmax; s_imin[threadIdx.x >> 6] = vimin; }
  }
  __syncthreads();
  if (tile_max && threadIdx.x == 0) {
    unsigned long long x = s_max[0];
    unsigned xi = s_imin[0];
    for (int w = 1; w < kThreads / 64; w++) { x = s_max[w] > x ? s_max[w] : x; xi = s_imin[w] > xi ? s_imin[w] : xi; }
    tile_max[2 * tile] = x;
    tile_max[2 * tile + 1] = xi;
  }
#pragma unroll
  for (int u = 0; u < kGbRows; u++)
    if (dst[u] >= 0) pvals[dst[u]] = s_stage[u * kThreads + threadIdx.x];
  __syncthreads();
  }
  unsigned* s_stage32 = reinterpret_cast<unsigned*>(s_stage);
#pragma unroll
  for (int u = 0; u < kGbRows; u++)
    if (live[u]) s_stage32[s_start[bin[u]] + rank[u]] = rw[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kGbRows; u++)
    if (dst[u] >= 0) prows[dst[u]] = s_stage32[u * kThreads + threadIdx.x];
}

// distinct keys expected among n rows when a sample of p rows held d (uniform-urn model; as in ah_hash.hip)
static double gb_extrapolate(double d, double p, double n) {
  const double r = d / p;
  if (r >= 0.999) return n;
  if (r < 1.0 / 32) return d;   // every key was sampled dozens of times: nothing more to come
  double lo = 1e-9, hi = 64.0;
  for (int it = 0; it < 60; it++) {
    const double mid = 0.5 * (lo + hi);
    if ((1.0 - exp(-mid)) / mid > r) lo = mid; else hi = mid;
  }
  const double C = p / (0.5 * (lo + hi));
  const double e = C * (1.0 - exp(-n / C));
  return e < n ? e : n;
}

}  // namespace
