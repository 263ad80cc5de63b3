// ah_msd.h — what the two "two partition levels, then one wave per bucket" pipelines share: the MSD path of sort_indices
// (ah_sort_msd.hip) and the sort-based group-by for very many groups (ah_groupby.hip).  Level 1 cuts the rows into ≤ 1024
// parents, level 2 every parent into ≤ 2048 buckets (its tiles never cross the parent boundary), the last step gives each
// bucket (≈ 64 rows) to one wave that sorts it in registers.
#pragma once
#include "ah_common.h"
#include "ah_bins.h"

namespace {

constexpr int kMsTile = 4096;                  // rows per (virtual) tile
constexpr int kMsRows = kMsTile / kThreads;     // 4 per thread
constexpr int kMaxNb2 = 2048;

// ---- which rows does this workgroup take?  level 1: tile = consecutive 4096 rows; level 2: the tiles of every parent
// start at the parent's first row (the last one is short), numbered parent after parent --------------------------------
struct TileRange { int64_t lo, hi, id; int parent; };   // id = row of the count / offset tables
// Level 1 (pstart = nullptr): plain tiles over [0, n), consecutive tiles on ONE XCD (ah_bins.h) so that the runs they append to
// a bin meet in that XCD's L2.  Level 2: the tiles of a parent start at the parent's first row (the last one is short).  All
// tiles of a parent should run on one XCD too — the parent's 1–2 MB output region then fills up inside one L2 — so tiles are
// numbered class by class (class = parent mod 8: parents 0, 8, 16, … first), block b serves class b mod 8 (the observed
// block → XCD rule; only speed depends on it), and blocks left over in one class take the tiles another class has too many of.
__device__ __forceinline__ int ms_parent_of(int j, int nparents) { const int per = nparents >> 3; return ((j % per) << 3) | (j / per); }
// all threads call
// pend (nullable): parent p's rows are [pstart[p], pend[p]) — parents that are REGIONS with room behind their rows (the reserving
// level-1 scatter of the group-by) — instead of [pstart[p], pstart[p + 1]).
__device__ __forceinline__ TileRange ms_tile(const unsigned* __restrict__ pstart, int nparents, int64_t n, unsigned* s_cnt, unsigned* s_start,
                                             unsigned* s_wsum, int* s_pick, const unsigned* __restrict__ pend = nullptr) {
  TileRange r{0, 0, 0, -1};
  if (!pstart) {
    const int64_t tile = xcd_contiguous_tile((n + kMsTile - 1) / kMsTile);
    if (tile >= 0) { r.lo = tile * kMsTile; r.hi = r.lo + kMsTile < n ? r.lo + kMsTile : n; r.id = tile; r.parent = 0; }
    return r;
  }
  const int t = threadIdx.x, per = nparents >> 3;
  unsigned tiles = 0;
  if (t < nparents) { const int p = ms_parent_of(t, nparents); tiles = ((pend ? pend[p] : pstart[p + 1]) - pstart[p] + kMsTile - 1) / kMsTile; }
  s_cnt[t] = tiles;
  if (t == 0) *s_pick = -1;
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nparents);
  // tile number g this block serves
  const unsigned x = blockIdx.x & 7, q = blockIdx.x >> 3, nblk = gridDim.x >> 3;
  auto class_lo = [&](unsigned c) { return s_start[c * per]; };
  auto class_n = [&](unsigned c) { return (c == 7 ? s_start[nparents - 1] + s_cnt[nparents - 1] : s_start[(c + 1) * per]) - s_start[c * per]; };
  long long g = -1;
  if (q < class_n(x)) {
    g = class_lo(x) + q;
  } else {
    unsigned spare = q - class_n(x);                      // my rank among the idle blocks …
    for (unsigned c = 0; c < x; c++) spare += nblk > class_n(c) ? nblk - class_n(c) : 0u;
    for (unsigned c = 0; c < 8 && g < 0; c++) {           // … = rank of the surplus tile I take
      const unsigned surplus = class_n(c) > nblk ? class_n(c) - nblk : 0u;
      if (spare < surplus) g = class_lo(c) + nblk + spare; else spare -= surplus;
    }
  }
  if (g >= 0 && t < nparents && tiles && s_start[t] <= g && g < s_start[t] + tiles) *s_pick = t;
  __syncthreads();
  const int j = *s_pick;
  if (j < 0) return r;
  const int p = ms_parent_of(j, nparents);
  const int64_t b0 = pstart[p], b1 = pend ? pend[p] : pstart[p + 1];
  r.lo = b0 + (int64_t)(g - s_start[j]) * kMsTile;
  r.hi = r.lo + kMsTile < b1 ? r.lo + kMsTile : b1;
  r.id = g;
  r.parent = p;
  return r;
}

// ms_tile for EVERY block of a launch of `nblocks` workgroups, written to a table the launch reads instead: ms_tile costs each
// workgroup ≈ 2 µs of loads, a block scan and half a dozen barriers before its first row is requested.  A scatter's workgroup lives
// ten times that long and does not notice (measured: profiles/r06_negative_results.txt); a histogram's lives 5 µs.  One thread per
// block; the same arithmetic (blockIdx → b, gridDim → nblocks).
__global__ __launch_bounds__(kThreads) void ms_tile_table_kernel(const unsigned* __restrict__ pstart, const unsigned* __restrict__ pend, int nparents,
                                                                  unsigned nblocks, TileRange* __restrict__ table) {
  __shared__ unsigned s_cnt[kThreads], s_start[kThreads], s_wsum[kThreads / 64];
  const int t = threadIdx.x, per = nparents >> 3;
  unsigned tiles = 0;
  if (t < nparents) { const int p = ms_parent_of(t, nparents); tiles = ((pend ? pend[p] : pstart[p + 1]) - pstart[p] + kMsTile - 1) / kMsTile; }
  s_cnt[t] = tiles;
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nparents);
  const unsigned b = blockIdx.x * (unsigned)kThreads + (unsigned)t;
  if (b >= nblocks) return;
  const unsigned x = b & 7, q = b >> 3, nblk = nblocks >> 3;
  auto class_lo = [&](unsigned c) { return s_start[c * per]; };
  auto class_n = [&](unsigned c) { return (c == 7 ? s_start[nparents - 1] + s_cnt[nparents - 1] : s_start[(c + 1) * per]) - s_start[c * per]; };
  long long g = -1;
  if (q < class_n(x)) {
    g = class_lo(x) + q;
  } else {
    unsigned spare = q - class_n(x);
    for (unsigned c = 0; c < x; c++) spare += nblk > class_n(c) ? nblk - class_n(c) : 0u;
    for (unsigned c = 0; c < 8 && g < 0; c++) {
      const unsigned surplus = class_n(c) > nblk ? class_n(c) - nblk : 0u;
      if (spare < surplus) g = class_lo(c) + nblk + spare; else spare -= surplus;
    }
  }
  TileRange r{0, 0, 0, -1};
  if (g >= 0) {
    int lo = 0, hi = nparents - 1;   // the last j with s_start[j] ≤ g that has tiles (entries without tiles share their successor's start)
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((long long)s_start[mid] <= g) lo = mid; else hi = mid - 1; }
    int j = lo;
    while (j > 0 && !s_cnt[j]) j--;
    if (s_cnt[j] && (long long)s_start[j] <= g && g < (long long)s_start[j] + s_cnt[j]) {
      const int p = ms_parent_of(j, nparents);
      const int64_t b0 = pstart[p], b1 = pend ? pend[p] : pstart[p + 1];
      r.lo = b0 + (int64_t)(g - s_start[j]) * kMsTile;
      r.hi = r.lo + kMsTile < b1 ? r.lo + kMsTile : b1;
      r.id = g;
      r.parent = p;
    }
  }
  table[b] = r;
}

// ---- level 2 offsets: one workgroup per parent -------------------------------------------------------------------------
// toffs[vt][d] = position of virtual tile vt's first row of digit d = parent start + rows of smaller digits in the parent
// + rows of digit d in the parent's earlier tiles;  bstart[parent · nb + d] = first row of bucket (parent, d)
__global__ __launch_bounds__(kThreads) void ms_offs2_kernel(const unsigned* __restrict__ cnt, const unsigned* __restrict__ pstart, int nparents, int nb,
                                                             unsigned* __restrict__ toffs, unsigned* __restrict__ bstart, int64_t n) {
  __shared__ unsigned s_cnt[kThreads], s_start[kThreads], s_wsum[kThreads / 64];
  __shared__ unsigned s_carry;
  const int t = threadIdx.x, p = blockIdx.x;
  unsigned tiles = 0;
  if (t < nparents) { const int q = ms_parent_of(t, nparents); tiles = (pstart[q + 1] - pstart[q] + kMsTile - 1) / kMsTile; }
  s_cnt[t] = tiles;
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nparents);
  const int j = (p & 7) * (nparents >> 3) + (p >> 3);   // this parent's place in the class-major tile numbering (ms_tile)
  const int64_t vt0 = s_start[j], vt1 = vt0 + s_cnt[j];
  __syncthreads();
  const unsigned base = pstart[p];
  if (nb <= kThreads / 2) {
    // few digits (the two-level cuts of the group-by and the encode: 64 … 512): kThreads / nb threads per digit, each walking its own
    // share of the parent's tiles — one thread per digit walked all of them (256 tiles at 2^26 rows and 64 parents: two dependent
    // loops of 256 steps with 128 of the 1024 threads at work, 98 µs per call)
    const int d = t & (nb - 1), g = t / nb, G = kThreads / nb;
    const int64_t per = (vt1 - vt0 + G - 1) / G, a0 = vt0 + g * per, a1 = a0 + per < vt1 ? a0 + per : vt1;
    unsigned part = 0;
    for (int64_t vt = a0; vt < a1; vt++) part += cnt[vt * nb + d];
    s_cnt[t] = part;
    __syncthreads();
    unsigned before = 0, tot = 0;   // rows of digit d in the shares before mine / in the whole parent
    for (int gg = 0; gg < G; gg++) { const unsigned c = s_cnt[gg * nb + d]; before += gg < g ? c : 0u; tot += c; }
    __syncthreads();
    s_cnt[t] = t < nb ? tot : 0u;
    __syncthreads();
    block_excl_scan(s_cnt, s_start, s_wsum, nb);
    unsigned run = base + s_start[d] + before;
    if (g == 0) bstart[(int64_t)p * nb + d] = run;
    for (int64_t vt = a0; vt < a1; vt++) { toffs[vt * nb + d] = run; run += cnt[vt * nb + d]; }
    if (p == nparents - 1 && t == 0) bstart[(int64_t)nparents * nb] = (unsigned)n;
    return;
  }
  // digits t (and t + 1024 when nb = 2048): totals over the parent's tiles
  unsigned tot[2] = {0, 0};
  for (int64_t vt = vt0; vt < vt1; vt++)
    for (int h = 0; h < 2; h++) { const int d = t + h * kThreads; if (d < nb) tot[h] += cnt[vt * nb + d]; }
  unsigned excl[2];
  unsigned carry = 0;
  for (int h = 0; h < 2; h++) {
    if (h * kThreads >= nb) break;
    s_cnt[t] = t + h * kThreads < nb ? tot[h] : 0u;
    __syncthreads();
    block_excl_scan(s_cnt, s_start, s_wsum, kThreads);
    excl[h] = carry + s_start[t];
    if (t == kThreads - 1) s_carry = s_start[t] + s_cnt[t];
    __syncthreads();
    carry += s_carry;
    __syncthreads();
  }
  for (int h = 0; h < 2; h++) {
    const int d = t + h * kThreads;
    if (d >= nb) break;
    unsigned run = base + excl[h];
    bstart[(int64_t)p * nb + d] = run;
    for (int64_t vt = vt0; vt < vt1; vt++) { toffs[vt * nb + d] = run; run += cnt[vt * nb + d]; }
  }
  if (p == nparents - 1 && t == 0) bstart[(int64_t)nparents * nb] = (unsigned)n;
}

// the value lane ^ j holds, without the LDS crossbar (ds_bpermute): DPP within rows of 16 lanes, the gfx950 permlane swaps
// across rows and halves (scripts/micro/dpp_xor_test.hip checks all six forms against __shfl_xor)
template <int CTRL>
__device__ __forceinline__ unsigned ms_dpp(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ unsigned ms_xor_lane(unsigned v, int j, int lane) {   // j is a constant after unrolling
  switch (j) {
    case 1: return ms_dpp<0xB1>(v);                         // quad_perm [1,0,3,2]
    case 2: return ms_dpp<0x4E>(v);                         // quad_perm [2,3,0,1]
    case 4: { const unsigned a = ms_dpp<0x124>(v), b = ms_dpp<0x12C>(v); return (lane & 4) ? a : b; }   // row_ror:4 / row_ror:12
    case 8: return ms_dpp<0x128>(v);                        // row_ror:8
    case 16: { auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return (lane & 16) ? r[0] : r[1]; }
    default: { auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return (lane & 32) ? r[0] : r[1]; }
  }
}
// keep the smaller (keep_min) or the larger of (mine, other)
template <typename E>
__device__ __forceinline__ E ms_pick(const E& mine, const E& other, bool keep_min) {
  const bool other_less = E::less(other, mine);
  return (other_less == keep_min) ? other : mine;
}

// bitonic network over N = 64 · SLOTS pairs held SLOTS per lane: element i = slot · 64 + lane
// E: an element type with static less(a, b) and xchg(a, j, lane) (the element lane ^ j holds)
template <typename E, int SLOTS>
__device__ __forceinline__ void ms_bitonic(E (&x)[SLOTS], int lane) {
#pragma unroll
  for (int k = 2; k <= 64 * SLOTS; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {   // the partner sits in the same lane, slot ^ (j / 64)
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) {
          const int ps = sl ^ (j >> 6);
          if (ps > sl) {
            const bool asc = (sl & (k >> 6)) == 0;   // (i & k) == 0; k = N: always
            const bool swap = E::less(x[ps], x[sl]) == asc;
            const E a = x[sl], bb = x[ps];
            x[sl] = swap ? bb : a;
            x[ps] = swap ? a : bb;
          }
        }
      } else {
        const bool low = (lane & j) == 0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) {
          const bool asc = k < 64 ? (lane & k) == 0 : (sl & (k >> 6)) == 0;
          x[sl] = ms_pick(x[sl], E::xchg(x[sl], j, lane), low == asc);
        }
      }
    }
  }
}
}  // namespace
