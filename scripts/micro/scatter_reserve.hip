// The partitioned group-by's scatter WITHOUT the histogram pass in front of it — a measurement for DESIGN.md §3.2 ("what would meet the
// 2^16 bar").  Today: gb_hist_kernel (a whole read of the key column: 113 µs at 2^26 rows) + three offset kernels (40 µs) tell every
// tile where its run of every partition starts, then gb_scatter_kernel (438 µs) moves the records.  Here every partition owns a REGION
// (capacity = its rows × 1.25 + 4096: the library would size it from the 2^21-row sample) and a tile reserves its run with one
// returning atomicAdd per (tile, partition) on the partition's cursor.  Which tile's run comes first inside a region depends on
// timing; the group-by's results do not (fixed-point sums, counts, minima of first rows).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../arrow_go_amd/csrc -I../../include scatter_reserve.hip -o /tmp/scatter_reserve
//   /tmp/scatter_reserve [log2 rows = 26] [log2 groups = 16]
// Prints µs per kernel for both pipelines at 64 and 1024 partitions and checks that every partition holds the same multiset of
// {key, value, row} records either way (an order-independent checksum per partition).
// Round 5: each partition's region can be cut into NX sub-regions, one per XCD (cursor (p, blockIdx & 7)): the runs one XCD appends
// to a partition then stay neighbours in THAT L2 and leave as whole lines, as they do behind the offsets table.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ah_partition.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

// the shipped gb_scatter_kernel (ah_partition.h) with the offsets table replaced by {region start, capacity, cursor} per partition
template <bool HAS_VALS>
__global__ __launch_bounds__(kThreads) void gb_scatter_reserve_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ kvalid, int64_t koff,
                                                               const unsigned long long* __restrict__ vals, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                               int64_t n, int lp, int nb, int nx, int64_t ntiles, const unsigned* __restrict__ rstart, const unsigned* __restrict__ rcap,
                                                               unsigned* __restrict__ cursor, unsigned* __restrict__ overflow,
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
  // RESERVE: the tile's run of partition b starts where the partition's cursor stood — one returning atomic per (tile, partition
  // with rows), issued as soon as the counts are final and consumed after the keys have been staged
  unsigned goff_excl = 0;
  bool over = false;
  if ((int)threadIdx.x < nb) {
    const unsigned cnt = s_cnt[threadIdx.x];
    if (cnt) {
      const int seg = (int)threadIdx.x * nx + (nx > 1 ? (int)(blockIdx.x & 7) % nx : 0);
      const unsigned at = atomicAdd(&cursor[seg], cnt);
      over = at + cnt > rcap[seg];
      goff_excl = rstart[seg] + at;
    }
  }
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  if ((int)threadIdx.x < nb) s_goff[threadIdx.x] = over ? 0xFFFFFFFFu : goff_excl - s_start[threadIdx.x];
  if (over) atomicExch(overflow, 1u);
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
    dst[u] = q < tile_n && s_goff[s_bin[q]] != 0xFFFFFFFFu ? (int64_t)(unsigned)(s_goff[s_bin[q]] + (unsigned)q) : -1;   // (a run that does not fit its region is dropped: the call is void)
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


}  // namespace

namespace {

__global__ void gen_kernel(unsigned long long* keys, unsigned long long* vals, int64_t n, unsigned long long groups) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long h = gb_mix((unsigned long long)i * 0x2545F4914F6CDD1Dull + 17);
  keys[i] = (h % groups) * 0x9E3779B97F4A7C15ull;
  vals[i] = __builtin_bit_cast(unsigned long long, (double)(int)(h >> 40) * 0.125);
}

// order-independent checksum and count of the records in [start[p], start[p] + cnt[p])
__global__ void check_kernel(const unsigned long long* pkeys, const unsigned long long* pvals, const unsigned* prows, const unsigned* start, const unsigned* cnt,
                             int lp, int nx, unsigned long long* sums, unsigned* bad) {
  const int p = blockIdx.x / nx;
  unsigned long long s = 0;
  for (unsigned j = threadIdx.x; j < cnt[blockIdx.x]; j += blockDim.x) {
    const size_t q = (size_t)start[blockIdx.x] + j;
    s += gb_mix(pkeys[q] ^ (pvals[q] * 0xD6E8FEB86659FD93ull) ^ ((unsigned long long)prows[q] << 17));
    if (gb_part(gb_mix(pkeys[q]), lp) != (unsigned)p) atomicAdd(bad, 1u);
  }
  atomicAdd(&sums[p], s);
}

float ms_between(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

}  // namespace

int main(int argc, char** argv) {
  const int lgn = argc > 1 ? atoi(argv[1]) : 26, lgg = argc > 2 ? atoi(argv[2]) : 16;
  const int64_t n = (int64_t)1 << lgn;
  unsigned long long *keys, *vals, *pkeys, *pvals, *pkeys2, *pvals2, *tile_max, *sums;
  unsigned *prows, *prows2, *cnt_tm, *toffs, *gsum, *binstart, *rstart, *rcap, *cursor, *flags;
  const int64_t ntiles = (n + kGbTile - 1) / kGbTile, ngrp = (ntiles + kGroupTiles - 1) / kGroupTiles;
  const size_t cap_rows = (size_t)(n * 5 / 4) + (size_t)kMaxBins * 8 * 1024 + 4096;
  CK(hipMalloc(&keys, n * 8)); CK(hipMalloc(&vals, n * 8));
  CK(hipMalloc(&pkeys, n * 8)); CK(hipMalloc(&pvals, n * 8)); CK(hipMalloc(&prows, n * 4));
  CK(hipMalloc(&pkeys2, cap_rows * 8)); CK(hipMalloc(&pvals2, cap_rows * 8)); CK(hipMalloc(&prows2, cap_rows * 4));
  CK(hipMalloc(&cnt_tm, (size_t)ntiles * kMaxBins * 4)); CK(hipMalloc(&toffs, (size_t)ntiles * kMaxBins * 4)); CK(hipMalloc(&gsum, (size_t)ngrp * kMaxBins * 4));
  CK(hipMalloc(&binstart, (kMaxBins + 1) * 4)); CK(hipMalloc(&rstart, kMaxBins * 8 * 4)); CK(hipMalloc(&rcap, kMaxBins * 8 * 4)); CK(hipMalloc(&cursor, kMaxBins * 8 * 4));
  CK(hipMalloc(&flags, 64)); CK(hipMalloc(&tile_max, (size_t)ntiles * 16)); CK(hipMalloc(&sums, 2 * kMaxBins * 8));
  gen_kernel<<<(unsigned)((n + 255) / 256), 256>>>(keys, vals, n, 1ull << lgg);
  CK(hipDeviceSynchronize());
  hipEvent_t ev[8];
  for (auto& e : ev) CK(hipEventCreate(&e));
  const unsigned tgrid = (unsigned)(((ntiles + 7) / 8) * 8);
  for (int lpnx : {6, 9, 10, 6 + 16, 9 + 16, 10 + 16}) {
    const int lp = lpnx & 15, nx = lpnx >= 16 ? 8 : 1;
    const int P = 1 << lp;
    float best[6] = {1e9f, 1e9f, 1e9f, 1e9f, 1e9f, 1e9f};
    std::vector<unsigned> h_start(P + 1), h_rstart(P * nx), h_rcap(P * nx), h_cnt(P), h_cur(P * nx);
    for (int rep = 0; rep < 5; rep++) {
      // ---- as shipped: hist → offsets (3 kernels) → scatter
      CK(hipEventRecord(ev[0]));
      gb_hist_kernel<<<tgrid, kGbHistThreads>>>(keys, nullptr, 0, n, lp, P, ntiles, cnt_tm);
      CK(hipEventRecord(ev[1]));
      colsum_kernel<<<(unsigned)ngrp, kMaxBins>>>(cnt_tm, P, ntiles, gsum);
      bin_prefix_kernel<<<1, kMaxBins>>>(gsum, P, ngrp, n, binstart);
      tile_offs_kernel<<<(unsigned)ngrp, kMaxBins>>>(cnt_tm, gsum, P, ntiles, toffs);
      CK(hipEventRecord(ev[2]));
      gb_scatter_kernel<true><<<tgrid, kThreads>>>(keys, nullptr, 0, vals, nullptr, 0, n, lp, P, ntiles, toffs, pkeys, pvals, prows, tile_max);
      CK(hipEventRecord(ev[3]));
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h_start.data(), binstart, (P + 1) * 4, hipMemcpyDeviceToHost));
      // ---- regions sized from the true counts (the library: from the sample), one fill, the reserving scatter
      unsigned at = 0;
      for (int p = 0; p < P; p++) {
        h_cnt[p] = h_start[p + 1] - h_start[p];
        for (int x = 0; x < nx; x++) {
          const unsigned share = h_cnt[p] / nx;
          h_rcap[p * nx + x] = (share + share / 4 + (nx > 1 ? 1024 : 4096) + 15u) & ~15u;   // (sub-regions start on 128-byte lines of the key array)
          h_rstart[p * nx + x] = at;
          at += h_rcap[p * nx + x];
        }
      }
      if ((size_t)at > cap_rows) { fprintf(stderr, "regions need %u rows, %zu allocated\n", at, cap_rows); return 1; }
      CK(hipMemcpy(rstart, h_rstart.data(), P * nx * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(rcap, h_rcap.data(), P * nx * 4, hipMemcpyHostToDevice));
      CK(hipEventRecord(ev[4]));
      CK(hipMemsetAsync(cursor, 0, P * nx * 4));
      CK(hipMemsetAsync(flags, 0, 64));
      gb_scatter_reserve_kernel<true><<<tgrid, kThreads>>>(keys, nullptr, 0, vals, nullptr, 0, n, lp, P, nx, ntiles, rstart, rcap, cursor, flags, pkeys2, pvals2, prows2, tile_max);
      CK(hipEventRecord(ev[5]));
      CK(hipDeviceSynchronize());
      const float t[6] = {ms_between(ev[0], ev[1]), ms_between(ev[1], ev[2]), ms_between(ev[2], ev[3]), ms_between(ev[0], ev[3]), ms_between(ev[4], ev[5]), 0};
      for (int k = 0; k < 5; k++) best[k] = t[k] < best[k] ? t[k] : best[k];
    }
    // ---- same records either way?
    unsigned h_flags[2];
    CK(hipMemcpy(h_flags, flags, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_cur.data(), cursor, P * nx * 4, hipMemcpyDeviceToHost));
    bool ok = h_flags[0] == 0;
    for (int p = 0; p < P; p++) { unsigned t = 0; for (int x = 0; x < nx; x++) t += h_cur[p * nx + x]; ok = ok && t == h_cnt[p]; }
    CK(hipMemset(sums, 0, 2 * kMaxBins * 8));
    CK(hipMemset(flags, 0, 64));
    unsigned *d_start_a, *d_cnt;
    CK(hipMalloc(&d_start_a, P * 4)); CK(hipMalloc(&d_cnt, P * 4));
    CK(hipMemcpy(d_start_a, h_start.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cnt, h_cnt.data(), P * 4, hipMemcpyHostToDevice));
    check_kernel<<<P, 1024>>>(pkeys, pvals, prows, d_start_a, d_cnt, lp, 1, sums, flags + 1);
    check_kernel<<<P * nx, 1024>>>(pkeys2, pvals2, prows2, rstart, cursor, lp, nx, sums + kMaxBins, flags + 1);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h_sums(2 * kMaxBins);
    CK(hipMemcpy(h_sums.data(), sums, 2 * kMaxBins * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_flags, flags, 8, hipMemcpyDeviceToHost));
    for (int p = 0; p < P; p++) ok = ok && h_sums[p] == h_sums[kMaxBins + p];
    ok = ok && h_flags[1] == 0;
    CK(hipFree(d_start_a)); CK(hipFree(d_cnt));
    printf("2^%d rows, 2^%d groups, %4d partitions x %d regions: hist %.1f us + offsets %.1f us + scatter %.1f us = %.1f us | fill + reserving scatter %.1f us | %s\n", lgn, lgg, P, nx,
           best[0] * 1e3f, best[1] * 1e3f, best[2] * 1e3f, best[3] * 1e3f, best[4] * 1e3f, ok ? "same records" : "MISMATCH");
    fflush(stdout);
  }
  return 0;
}
