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
// A workgroup takes kSampleBatches × 16 sample groups, sixteen (one per wave) at a time.
// hist (nullable): [8][1024] counts of the sampled rows by {eighth of the column — the tiles the scatter gives to one XCD
// (xcd_contiguous_tile), xrows rows each —, top 10 bits of gb_mix (a null key: bucket 0, as the scatter puts it)}: what the
// reserving scatter's regions are sized from (gb_layout_kernel).  Counted in LDS for the eighth the workgroup starts in (every
// workgroup but seven lies inside one), one global atomic per touched bucket and workgroup.
// ONE launch covers both halves: the first half of the grid takes the even sample groups and marks bitmap A (bm), the second half the
// odd groups and bitmap B (bm + nwords): |A| and |A ∪ B| are the two points of the curve, counted and posted by gb_sample_finish_kernel.
constexpr int kSampleBatches = 4;
__global__ __launch_bounds__(1024) void gb_sample_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ kvalid, int64_t koff,
                                                          int64_t n, int64_t ngroups, int64_t stride, unsigned* __restrict__ bm, unsigned mmask,
                                                          unsigned* __restrict__ hist, int64_t xrows) {
  __shared__ unsigned long long s_seen[1024];
  __shared__ unsigned s_hist[1024];
  s_seen[threadIdx.x] = 0;
  s_hist[threadIdx.x] = 0;
  __syncthreads();
  const int half = blockIdx.x >= gridDim.x / 2 ? 1 : 0;
  const int64_t blk = (int64_t)blockIdx.x - (half ? gridDim.x / 2 : 0);
  if (bm) bm += half ? ((size_t)mmask + 1) / 32 : 0;   // bm == nullptr: the histogram alone (the partition-first encode sizes its regions from it)
  const int64_t gfirst = (blk * 16 * kSampleBatches) * 2 + half;
  const int x0 = hist ? (int)((gfirst * stride) / xrows) : 0;   // (uniform)
  for (int r = 0; r < kSampleBatches; r++) {
    const int64_t g = ((blk * kSampleBatches + r) * 16 + (threadIdx.x >> 6)) * 2 + half;
    const int64_t i = g * stride + (threadIdx.x & 63);
    const bool in = g < ngroups && i < n;
    const bool kv = in && ah_bit(kvalid, koff + i);
    const uint64_t mix = kv ? gb_mix(keys[i]) : 0ull;
    if (hist && in) {
      const unsigned b = kv ? (unsigned)(mix >> 54) : 0u;
      const int x = (int)(i / xrows);
      if (x == x0) atomicAdd(&s_hist[b], 1u);
      else atomicAdd(&hist[(x < 7 ? x : 7) * 1024 + (int)b], 1u);
    }
    if (!kv || !bm) continue;
    const uint64_t m = mix | 1ull;
    const unsigned f = (unsigned)(m >> 44) & 1023u;
    if (atomicExch(&s_seen[f], m) == m) continue;   // an LDS atomic, so that of the lanes holding a hot key at this instant only one goes on
    const unsigned b = (unsigned)(m >> 20) & mmask;
    if (!((__hip_atomic_load(&bm[b >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (b & 31)) & 1u)) atomicOr(&bm[b >> 5], 1u << (b & 31));
  }
  if (hist) {
    __syncthreads();
    const unsigned cnt = s_hist[threadIdx.x];
    if (cnt) atomicAdd(&hist[(x0 < 7 ? x0 : 7) * 1024 + (int)threadIdx.x], cnt);
  }
}
// |A| and |A ∪ B| of the two bitmaps (nwords64 64-bit words each), posted to the host's mailbox by the last workgroup to finish —
// one launch where two popcounts (two kernels each) and a posting kernel were five.  acc: two words, zero on entry; done: one word, zero.
__global__ __launch_bounds__(256) void gb_sample_finish_kernel(const unsigned long long* __restrict__ bm, int64_t nwords64, unsigned long long* __restrict__ acc,
                                                               unsigned* __restrict__ done, unsigned long long* mb, unsigned long long seq) {
  __shared__ unsigned long long s_a[4], s_u[4];
  __shared__ unsigned s_last;
  unsigned long long a = 0, u = 0;
  for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < nwords64; w += (int64_t)gridDim.x * 256) {
    const unsigned long long x = bm[w], y = bm[nwords64 + w];
    a += (unsigned long long)__popcll(x);
    u += (unsigned long long)__popcll(x | y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); u += __shfl_down(u, o, 64); }
  if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = a; s_u[threadIdx.x >> 6] = u; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&acc[0], s_a[0] + s_a[1] + s_a[2] + s_a[3]);
    atomicAdd(&acc[1], s_u[0] + s_u[1] + s_u[2] + s_u[3]);
    __threadfence();
    s_last = atomicAdd(done, 1u) == gridDim.x - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (s_last && threadIdx.x == 0) {
    __threadfence();
    const unsigned long long w[2] = {__hip_atomic_load(&acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                     __hip_atomic_load(&acc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
    ah_mailbox_post(mb, seq, w, 2);
  }
}

// ---- 1b: regions instead of offsets (the reserving scatter) -----------------------------------------------------------------
// The histogram pass reads the whole key column only to tell every tile where its run of every partition starts.  The
// reserving scatter does without it: every (partition, XCD) pair owns a REGION of the record arrays, a tile reserves its run
// with one returning atomicAdd on the region's cursor, and the passes behind it walk the regions through a table of
// {virtual start, physical − virtual} per region (gb_segments_kernel).  One region per XCD and partition, not one per
// partition: the runs one XCD appends to a partition stay neighbours in THAT XCD's L2 and leave as whole lines, exactly
// as behind the offsets table — one cursor per partition interleaves the eight XCDs' runs line by line and the pass gets
// SLOWER than hist + scatter from 512 partitions on (scripts/micro/scatter_reserve.hip, profiles/r05_scatter_reserve_micro.txt).
// Which tile's run comes first inside a region depends on timing; nothing downstream depends on the order of a partition's
// records (fixed-point sums, counts, minima of first rows; first-occurrence minima in the encode).
// Region capacities come from the 2^21-row sample alone: (sampled rows S of the region + 6·√(S + 1)) × rows per sampled row + 64 —
// six standard deviations of a count that is binomial in the region's true size, so a region of an honestly sampled column is too
// small once in 10^9; all regions together need n + 6·√(regions · sampled rows) · scale ≤ 1.37 n rows at 1024 × 8 regions (the arrays
// hold 1.5 n).  (A floor of 1.25 × the even share, the first version, pushed a Zipf column's total beyond that: its light
// partitions were given far more than they hold.)  A region that turns out too small raises bit 2 of the redo word and the call
// is redone with the histogram — a column whose hot keys the sample did not see in proportion.
constexpr int kGbRegions = 8;   // regions per partition: the XCD (blockIdx & 7) the tile runs on
__global__ __launch_bounds__(1024) void gb_layout_kernel(const unsigned* __restrict__ hist, int lp, int64_t n, int64_t xrows, double scale, int64_t cap_rows,
                                                          unsigned* __restrict__ rstart, unsigned* __restrict__ rcap, unsigned* __restrict__ cursor,
                                                          unsigned* __restrict__ redo) {
  __shared__ unsigned s_tot[kMaxBins], s_start[kMaxBins], s_wsum[kThreads / 64];
  __shared__ unsigned s_sum[kGbRegions][kMaxBins];
  const int p = threadIdx.x, P = 1 << lp;
  // the 1024 buckets of each eighth → P partitions (partition = the bucket's top lp bits): thread t brings bucket t of every eighth
#pragma unroll
  for (int x = 0; x < kGbRegions; x++) s_sum[x][p] = 0;
  __syncthreads();
#pragma unroll
  for (int x = 0; x < kGbRegions; x++) {
    const unsigned h = hist[x * 1024 + p];
    if (h) atomicAdd(&s_sum[x][p >> (10 - lp)], h);
  }
  __syncthreads();
  unsigned cap[kGbRegions];
  unsigned tot = 0;
#pragma unroll
  for (int x = 0; x < kGbRegions; x++) {
    cap[x] = 0;
    if (p < P) {
      const unsigned sc = s_sum[x][p];
      const int64_t rows_x = n - x * xrows < 0 ? 0 : (n - x * xrows < xrows ? n - x * xrows : xrows);   // rows of this eighth of the column
      double want = ((double)sc + 6.0 * sqrt((double)sc + 1.0)) * scale + 64.0;
      if (want > (double)rows_x) want = (double)rows_x;          // a region never needs more than its eighth's rows
      cap[x] = ((unsigned)want + 15u) & ~15u;                    // regions start on whole lines of the 8-byte arrays
      tot += cap[x];
    }
  }
  s_tot[p] = tot;
  __syncthreads();
  block_excl_scan(s_tot, s_start, s_wsum, P);
  if (p < P) {
    unsigned at = s_start[p];
#pragma unroll
    for (int x = 0; x < kGbRegions; x++) {
      rstart[p * kGbRegions + x] = at;
      rcap[p * kGbRegions + x] = cap[x];
      cursor[p * kGbRegions + x] = 0;
      at += cap[x];
    }
    if (p == P - 1 && (int64_t)at > cap_rows) atomicOr(redo, 4u);   // the record arrays cannot hold the regions
  }
}

// after the reserving scatter: the regions as seen by the passes behind it.  vstart[r] = position of region r's first record in
// the DENSE order (partition by partition, XCD by XCD inside a partition; vstart[P·8] = n), delta[r] = physical − dense position,
// binstart[p] = vstart[8p] (binstart[P] = n).  A cursor beyond its capacity or a total that is not n: bit 2 of the redo word.
// tile_rng (nullable): the scatter's per-tile value ranges, reduced HERE to range[0..1] (gb_max_kernel) and checked
// (fx_range_check_kernel) — two launches less on a path that is a chain of small launches.
__global__ __launch_bounds__(1024) void gb_segments_kernel(const unsigned* __restrict__ rstart, const unsigned* __restrict__ rcap, const unsigned* __restrict__ cursor,
                                                            int P, int64_t n, unsigned* __restrict__ vstart, unsigned* __restrict__ delta,
                                                            unsigned* __restrict__ binstart, unsigned* __restrict__ redo,
                                                            const unsigned long long* __restrict__ tile_rng, int64_t ntiles, unsigned long long* __restrict__ range) {
  __shared__ unsigned s_tot[kMaxBins], s_start[kMaxBins], s_wsum[kThreads / 64];
  __shared__ unsigned long long s_max[16], s_imin[16];
  const int p = threadIdx.x;
  unsigned cnt[kGbRegions];
  unsigned tot = 0;
  bool over = (*redo & 4u) != 0;   // void already (gb_layout_kernel: the regions did not fit the arrays)
#pragma unroll
  for (int x = 0; x < kGbRegions; x++) {
    cnt[x] = 0;
    if (p < P) {
      cnt[x] = cursor[p * kGbRegions + x];
      over = over || cnt[x] > rcap[p * kGbRegions + x];
      tot += cnt[x];
    }
  }
  // a void attempt leaves NO records to the passes behind it (cursors beyond their capacities are not positions in the arrays)
  if (__syncthreads_or(over ? 1 : 0)) {
    if (p < P) {
      binstart[p] = 0;
#pragma unroll
      for (int x = 0; x < kGbRegions; x++) { vstart[p * kGbRegions + x] = 0; delta[p * kGbRegions + x] = 0; }
    }
    if (p == 0) { vstart[P * kGbRegions] = 0; binstart[P] = 0; atomicOr(redo, 4u); range[0] = 0; range[1] = 0; }
    return;
  }
  s_tot[p] = tot;
  __syncthreads();
  block_excl_scan(s_tot, s_start, s_wsum, P);
  if (p < P) {
    unsigned at = s_start[p];
    binstart[p] = at;
#pragma unroll
    for (int x = 0; x < kGbRegions; x++) {
      vstart[p * kGbRegions + x] = at;
      delta[p * kGbRegions + x] = rstart[p * kGbRegions + x] - at;
      at += cnt[x];
    }
    if (p == P - 1) {
      vstart[P * kGbRegions] = at;
      binstart[P] = at;
      if ((int64_t)at != n) over = true;
    }
    if (over) atomicOr(redo, 4u);
  }
  if (tile_rng) {
    unsigned long long m = 0, im = 0;
    for (int64_t i = threadIdx.x; i < ntiles; i += 1024) {
      const unsigned long long t = tile_rng[2 * i], ti = tile_rng[2 * i + 1];
      m = t > m ? t : m;
      im = ti > im ? ti : im;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long t = __shfl_down(m, o, 64), ti = __shfl_down(im, o, 64);
      m = t > m ? t : m;
      im = ti > im ? ti : im;
    }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = m; s_imin[threadIdx.x >> 6] = im; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 16; w++) { m = s_max[w] > m ? s_max[w] : m; im = s_imin[w] > im ? s_imin[w] : im; }
      range[0] = m;
      range[1] = im;
      if (fx_wide(m, im)) atomicOr(redo, 2u);
    }
  }
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
// RESERVE = true: no offsets table — the tile reserves its runs in the regions of (partition, blockIdx & 7) (see 1b above);
// toffs is null, rstart / rcap / cursor / redo describe the regions.
template <bool HAS_VALS, bool RESERVE = false>
__global__ __launch_bounds__(kThreads) void gb_scatter_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ kvalid, int64_t koff,
                                                               const unsigned long long* __restrict__ vals, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                               int64_t n, int lp, int nb, int64_t ntiles, const unsigned* __restrict__ toffs,
                                                               unsigned long long* __restrict__ pkeys, unsigned long long* __restrict__ pvals,
                                                               unsigned* __restrict__ prows, unsigned long long* __restrict__ tile_max,
                                                               const unsigned* __restrict__ rstart = nullptr, const unsigned* __restrict__ rcap = nullptr,
                                                               unsigned* __restrict__ cursor = nullptr, unsigned* __restrict__ redo = nullptr,
                                                               unsigned trash_base = 0, unsigned* __restrict__ cnt_out = nullptr,
                                                               unsigned* __restrict__ toffs_out = nullptr) {
  __shared__ unsigned s_cnt[kMaxBins], s_start[kMaxBins], s_goff[kMaxBins], s_wsum[kThreads / 64];
  __shared__ unsigned long long s_stage[kGbTile];
  __shared__ uint16_t s_bin[kGbTile];
  __shared__ unsigned long long s_max[kThreads / 64];
  __shared__ unsigned s_imin[kThreads / 64];
  // consecutive tiles on ONE XCD: the runs they append to a partition meet in that XCD's L2 and leave as whole lines
  const int64_t tile = xcd_contiguous_tile(ntiles);
  if (tile < 0) return;
  if (RESERVE) {   // void already (the regions did not fit the arrays, or an earlier tile's run did not fit its region): nothing to write
    if (threadIdx.x == 0) s_wsum[0] = __hip_atomic_load(redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 4u;
    __syncthreads();
    const unsigned dead = s_wsum[0];
    __syncthreads();
    if (dead) return;
  }
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
  if (!RESERVE && (int)threadIdx.x < nb) goff_excl = toffs[tile * nb + threadIdx.x];
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
  bool over = false;
  if (RESERVE && (int)threadIdx.x < nb) {
    // the tile's run of partition b starts where the region's cursor stood: one returning atomic per (tile, partition with rows),
    // issued as soon as the counts are final and consumed after the scan below
    const unsigned cnt = s_cnt[threadIdx.x];
    if (cnt) {
      const int r = (int)threadIdx.x * kGbRegions + (int)(blockIdx.x & (kGbRegions - 1));
      const unsigned at = atomicAdd(&cursor[r], cnt);
      over = at + cnt > rcap[r];
      goff_excl = rstart[r] + at;
    }
    // (tile, partition) → {rows, where the run went}: what the histogram pass and the offsets table told the passes that send
    // per-record results home tile by tile (the encode's un-permute)
    if (cnt_out) { cnt_out[tile * nb + threadIdx.x] = cnt; toffs_out[tile * nb + threadIdx.x] = goff_excl; }
  }
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  // a run that does not fit its region goes to kGbTile spare rows behind the regions (trash_base; position = the staged position):
  // the call is void and redone with the histogram.  (mod 2^32 below: a region may start below the tile's own prefix)
  if ((int)threadIdx.x < nb) s_goff[threadIdx.x] = over ? trash_base : goff_excl - s_start[threadIdx.x];
  if (RESERVE && over) atomicOr(redo, 4u);
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
    if (RESERVE) dst[u] = q < tile_n ? (int64_t)(unsigned)(s_goff[s_bin[q]] + (unsigned)q) : -1;
    else dst[u] = q < tile_n ? (int64_t)s_goff[s_bin[q]] + q : -1;
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

// one launch instead of seven memsets (each ≈ 5 µs of launch on a path that is a chain of small kernels): up to 8 {address, 16-byte
// words, 32-bit pattern} jobs, every workgroup takes its share of each
struct GbFill { uint4* p[8]; unsigned long long n16[8]; unsigned v[8]; int njobs; unsigned long long* ones; };
__global__ __launch_bounds__(256) void gb_fill_kernel(GbFill f) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int j = 0; j < f.njobs; j++) {
    const uint4 w = {f.v[j], f.v[j], f.v[j], f.v[j]};
    uint4* __restrict__ q = f.p[j];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)f.n16[j]; i += stride) q[i] = w;
  }
  if (f.ones && blockIdx.x == 0 && threadIdx.x == 0) *f.ones = ~0ull;   // (inside a word this thread has just filled: the last job's first)
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
