// ah_bins.h — the tile / bin bookkeeping shared by the two "partition through LDS" pipelines: the binned Take
// (ah_take_binned.hip: bin = window of the values column) and the partition-first group-by (ah_groupby.hip: bin =
// hash partition of the keys).  Both cut the input into tiles, count per (tile, bin), turn the count table into the
// global position of every tile's first record of every bin, and then stage each tile in bin order in LDS.
#pragma once
#include "ah_common.h"

namespace {

constexpr int kThreads = 1024;   // workgroup of the staging kernels: 2 per CU = all 32 wave slots
constexpr int kMaxBins = 1024;   // one thread per bin in the bookkeeping kernels
static_assert(kMaxBins == kThreads, "");

// consecutive tiles on one XCD: block b runs on XCD b & 7 (observed), so XCD x gets tiles [x·tpx, (x+1)·tpx)
__device__ __forceinline__ int64_t xcd_contiguous_tile(int64_t ntiles) {
  const int64_t tpx = (ntiles + 7) >> 3;
  const int64_t t = (int64_t)(blockIdx.x & 7) * tpx + (blockIdx.x >> 3);
  return ((int64_t)(blockIdx.x >> 3) < tpx && t < ntiles) ? t : -1;
}

// exclusive scan of cnt[0..nb) (nb ≤ 1024 = kThreads): thread t owns bin t.  Result in s_start.
__device__ __forceinline__ void block_excl_scan(const unsigned* s_cnt, unsigned* s_start, unsigned* s_wsum, int nb) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned a = t < nb ? s_cnt[t] : 0u;
  unsigned inc = a;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) s_wsum[wave] = inc;
  __syncthreads();
  unsigned base = 0;
#pragma unroll
  for (int w = 0; w < kThreads / 64; w++) if (w < wave) base += s_wsum[w];
  if (t < nb) s_start[t] = base + inc - a;
  __syncthreads();
}

// ---- the offsets table.  toffs[tile][bin] = global position of the tile's first record of that bin = (records of
// smaller bins) + (records of this bin in earlier tiles): a prefix sum DOWN the columns of the tile-major count table,
// done in three small launches with coalesced row accesses only (groups of kGroupTiles tiles):
constexpr int kGroupTiles = 128;
// (a) column sums of each group of tiles
__global__ __launch_bounds__(kMaxBins) void colsum_kernel(const unsigned* __restrict__ cnt_tm, int nb, int64_t ntiles, unsigned* __restrict__ gsum) {
  const int b = threadIdx.x;
  if (b >= nb) return;
  const int64_t t0 = (int64_t)blockIdx.x * kGroupTiles, t1 = t0 + kGroupTiles < ntiles ? t0 + kGroupTiles : ntiles;
  unsigned s = 0;
#pragma unroll 8
  for (int64_t t = t0; t < t1; t++) s += cnt_tm[t * nb + b];
  gsum[(int64_t)blockIdx.x * nb + b] = s;
}
// (b) one workgroup: per bin the exclusive prefix over the groups, then the exclusive prefix over the bins' totals
//     (= binstart; binstart[nb] = nidx); gsum is overwritten with binstart[b] + prefix of the groups before
//     mb (nullable): the largest bin's size is posted to the host's mailbox (ah_mailbox_post) — a caller that wants to see the
//     bins' balance reads it while the passes behind this one already run
__global__ __launch_bounds__(kMaxBins) void bin_prefix_kernel(unsigned* __restrict__ gsum, int nb, int64_t ngroups, int64_t nidx,
                                                               unsigned* __restrict__ binstart, unsigned long long* mb = nullptr, unsigned long long seq = 0) {
  __shared__ unsigned s_tot[kMaxBins], s_start[kMaxBins], s_wsum[kThreads / 64];
  __shared__ unsigned s_largest;
  const int t = threadIdx.x;
  // kMaxBins / nb threads per bin, each walking its own share of the groups (nb = 256, 128 groups: one thread per bin walked them
  // all, twice, with a quarter of the workgroup at work — 22 µs; nb is a power of two wherever it is below kMaxBins)
  const int S = nb < kMaxBins && (nb & (nb - 1)) == 0 ? kMaxBins / nb : 1;
  const int b = S > 1 ? t & (nb - 1) : t, sl = S > 1 ? t / nb : 0;
  const int64_t per = (ngroups + S - 1) / S, g0 = sl * per, g1 = g0 + per < ngroups ? g0 + per : ngroups;
  unsigned part = 0;
  if (b < nb) {
#pragma unroll 8
    for (int64_t g = g0; g < g1; g++) part += gsum[g * nb + b];
  }
  s_tot[t] = b < nb ? part : 0u;   // [share][bin]
  if (t == 0) s_largest = 0;
  __syncthreads();
  unsigned run = 0, before = 0;     // the bin's total; its rows in the shares before mine
  if (b < nb)
    for (int q = 0; q < S; q++) { const unsigned c = s_tot[q * nb + b]; before += q < sl ? c : 0u; run += c; }
  __syncthreads();
  s_tot[t] = t < nb ? run : 0u;
  __syncthreads();
  if (mb) {
    if (t < nb) atomicMax(&s_largest, run);
    __syncthreads();
    if (t == 0) { const unsigned long long w = s_largest; ah_mailbox_post(mb, seq, &w, 1); }
  }
  block_excl_scan(s_tot, s_start, s_wsum, nb);
  if (b < nb) {
    const unsigned start = s_start[b];
    if (sl == 0) binstart[b] = start;
    unsigned acc = start + before;
    for (int64_t g = g0; g < g1; g++) {
      const unsigned c = gsum[g * nb + b];
      gsum[g * nb + b] = acc;
      acc += c;
    }
  }
  if (t == 0) binstart[nb] = (unsigned)nidx;
}
// (c) inside each group: running offsets tile after tile
__global__ __launch_bounds__(kMaxBins) void tile_offs_kernel(const unsigned* __restrict__ cnt_tm, const unsigned* __restrict__ gbase, int nb,
                                                              int64_t ntiles, unsigned* __restrict__ toffs) {
  const int b = threadIdx.x;
  if (b >= nb) return;
  const int64_t t0 = (int64_t)blockIdx.x * kGroupTiles, t1 = t0 + kGroupTiles < ntiles ? t0 + kGroupTiles : ntiles;
  unsigned run = gbase[(int64_t)blockIdx.x * nb + b];
#pragma unroll 8
  for (int64_t t = t0; t < t1; t++) {
    toffs[t * nb + b] = run;
    run += cnt_tm[t * nb + b];
  }
}

}  // namespace
