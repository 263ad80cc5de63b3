// ah_take_binned.hip — Take for RANDOM indices into a column much larger than the caches.
//
// Same contract as take_kernel (ah_take.hip; primitiveTakeImpl, kernels/vector_selection.go:878-988):
// out[i] = values[idx[i]], null index or null value → payload 0 + validity 0, only valid index slots are
// bounds-checked.  The direct kernel pays one 64-byte line of HBM traffic per 8-byte gather when the indices are
// uniformly random over a 1 GiB column (PMC: ≈ 76 B/row moved for 20 algorithmic, 0.97 TB/s).  What the north star
// calls "LDS staging for the gather / scatter of take" is done here in four streaming passes, so that every line of
// `values` is fetched about once and every other byte moves in runs:
//
//   1 bin_hist     read idx; bounds check; per (bin, tile) counts         bin = idx >> shift: a window of `values`
//                                                                         small enough to stay in one XCD's L2
//   - scan         ah_cumulative_sum over the (bin-major) count table
//   2 bin_scatter  read idx again; ranks from LDS atomics; the tile is staged in bin order in LDS and written as
//                  one 4-byte record per row  { idx & (2^shift − 1) | row-in-tile << shift }
//   3 bin_gather   walks the records bin after bin: the 8-byte gathers hit the bin's window in L2 (workgroups are
//                  mapped so that an XCD works through whole bins), values are written in bin order (streaming)
//   4 unpermute    per tile: its runs (one per bin) are read back, placed at row-in-tile in LDS, and the tile of
//                  output values + its validity words leave with coalesced stores
//
// Traffic: 4 + (4 + 4) + (4 + 8 + 8 window + 8) + (4 + 8 + 8) = 60 B/row, all of it streaming or ≥ 64-byte runs,
// against ≈ 76 B/row of random lines.  Consecutive tiles are given to the SAME XCD (block b → XCD b mod 8 is the
// observed dispatch rule; only speed depends on it) so that the runs neighbouring tiles append to a bin merge into
// whole lines in that XCD's L2 before they are written back.
// The in-run order of records depends on LDS atomic timing; the OUTPUT does not (each record carries its row).
#include <type_traits>
#include "ah_common.h"
#include "ah_bins.h"

namespace {

constexpr int kTileBits = 13;
constexpr int kTile = 1 << kTileBits;   // rows per tile: row-in-tile fits 13 bits of the record
constexpr int kRowsPerThread = kTile / kThreads;  // 8 = two runs of 4 consecutive rows (one 16-byte index load each)
constexpr int kGatherBlock = 256;
constexpr int kGatherPerThread = 8;
constexpr int kGatherChunk = kGatherBlock * kGatherPerThread;  // 2048 records per workgroup

template <int W> struct UIntW;
template <> struct UIntW<1> { using type = uint8_t; };
template <> struct UIntW<2> { using type = uint16_t; };
template <> struct UIntW<4> { using type = uint32_t; };
template <> struct UIntW<8> { using type = uint64_t; };

// the tile's index slots of one thread: kGroups runs of 4 consecutive rows, each run ONE vector load (16 bytes for int32
// indices) — all of them issued before anything is used, so a workgroup has its whole 32 KiB of indices in flight at once.
// Row of (group g, element j) inside the tile: g·4·kThreads + 4·t + j.  Null / out-of-range / past-the-end slots read as 0
// (`live` says whether the row exists); *first_oob = smallest offending row of this thread or ~0.
template <typename IdxT>
struct alignas(sizeof(IdxT)) IdxVec4 { IdxT v[4]; };

// FULL: the tile lies entirely inside [0, nidx) — no per-row range test, every load unconditional (all tiles but the last).
// HAS_IV: an index validity bitmap exists; the 4 bits of a run come from one funnel-shifted word load issued next to the
// index load.  The code is branch-free on purpose: with per-row `if`s the compiler emitted 114 exec-mask branches and a
// rolled atomics loop, and the kernel ran at 2.8 TB/s of a 4-byte-per-row read.
template <typename IdxT, int THREADS, int ROWS, bool FULL, bool HAS_IV>
__device__ __forceinline__ void load_tile_indices(const IdxT* __restrict__ idx, const uint8_t* __restrict__ ivalid, int64_t ioff, int64_t base,
                                                   int64_t nidx, uint64_t nvalues, unsigned (&u)[ROWS], bool (&live)[ROWS],
                                                   unsigned long long* first_oob) {
  using UIdx = typename std::make_unsigned<IdxT>::type;
  constexpr int GROUPS = ROWS / 4;
  IdxVec4<IdxT> raw[GROUPS];
  unsigned vbits[GROUPS];
#pragma unroll
  for (int g = 0; g < GROUPS; g++) {
    const int64_t i0 = base + (int64_t)g * 4 * THREADS + 4 * (int64_t)threadIdx.x;
    if (FULL || i0 + 4 <= nidx) {
      raw[g] = *(const IdxVec4<IdxT>*)(idx + i0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) raw[g].v[j] = i0 + j < nidx ? idx[i0 + j] : (IdxT)0;
    }
    const int left = FULL ? 4 : (int)(nidx - i0 < 4 ? (nidx - i0 < 0 ? 0 : nidx - i0) : 4);
    vbits[g] = HAS_IV ? (unsigned)ah_load_bits64(ivalid, ioff + i0, left) : (left >= 4 ? 0xfu : ((1u << left) - 1u));
  }
  bool any_oob = false;
  // A null index slot gathers nothing that is kept (the un-permute pass clears its row), but it must not be binned through
  // index 0: with 10 % nulls a tenth of every tile's rows would land on ONE bin counter — same-address LDS atomics, served one
  // lane at a time (the histogram pass went 97 → 189 µs).  Null slots get a scattered in-range index instead.
  const int lgv = 63 - __builtin_clzll(nvalues < 0xffffffffull ? (nvalues ? nvalues : 1) : 0xffffffffull);   // 2^lgv ≤ nvalues
#pragma unroll
  for (int g = 0; g < GROUPS; g++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = g * 4 + j;
      const IdxT s = raw[g].v[j];
      const uint64_t w = (uint64_t)(UIdx)s;
      const bool valid = (vbits[g] >> j) & 1u;     // in range of the column AND a non-null index slot
      const bool oob = (std::is_signed<IdxT>::value && s < 0) || w >= nvalues;  // helpers.go:937-939
      const int64_t row = base + (int64_t)g * 4 * THREADS + 4 * (int64_t)threadIdx.x + j;
      live[k] = FULL ? true : (row < nidx);
      u[k] = valid ? (oob ? 0u : (unsigned)w) : (HAS_IV ? (((unsigned)row * 2654435761u) >> 1) >> (31 - lgv) : 0u);
      any_oob |= valid && oob;
    }
  }
  *first_oob = ~0ull;
  if (any_oob) {  // rare: name the first offender of this thread (rows ascend with g, j)
#pragma unroll
    for (int g = GROUPS - 1; g >= 0; g--) {
#pragma unroll
      for (int j = 3; j >= 0; j--) {
        const IdxT s = raw[g].v[j];
        const uint64_t w = (uint64_t)(UIdx)s;
        if (((vbits[g] >> j) & 1u) && ((std::is_signed<IdxT>::value && s < 0) || w >= nvalues))
          *first_oob = (unsigned long long)(base + (int64_t)g * 4 * THREADS + 4 * (int64_t)threadIdx.x + j);
      }
    }
  }
}
__device__ __forceinline__ unsigned row_in_tile(int k) { return (unsigned)((k >> 2) * 4 * kThreads + 4 * (int)threadIdx.x + (k & 3)); }

// ---- 0: are the indices clustered?  (the reference samples 32 points to pick its sorted / reverse loops,
// vector_selection.go:734-809; here the answer picks the direct kernel, which streams at 5 TB/s on such input)
template <typename IdxT>
__global__ __launch_bounds__(256) void sample_kernel(const IdxT* __restrict__ idx, int64_t nidx, int near, unsigned long long* __restrict__ hits) {
  const int64_t span = nidx / gridDim.x;
  const int64_t i = (int64_t)blockIdx.x * span + threadIdx.x;
  bool hit = false, next = false;
  if (threadIdx.x < 255 && i + 1 < nidx) {
    const long long a = (long long)idx[i], b = (long long)idx[i + 1];
    const long long d = a > b ? a - b : b - a;
    hit = d <= near;
    next = d == 1;
  }
  // low word: neighbours within one 128-byte line; high word: neighbours naming ADJACENT values (identity, slices, a reversed column)
  const int c = __syncthreads_count(hit), c1 = __syncthreads_count(next);
  if (threadIdx.x == 0 && c) atomicAdd(hits, (unsigned long long)c | ((unsigned long long)c1 << 32));
}

// ---- 1: per (tile, bin) counts + bounds check.  256 threads × 32 rows: 8 workgroups per CU, each with its whole 32 KiB of
// indices in flight at once (with 2 × 1024 threads × 8 rows the kernel was latency-bound at 2.9 TB/s).
constexpr int kHistThreads = 256, kHistRows = kTile / kHistThreads;
template <typename IdxT, bool FULL, bool HAS_IV>
__device__ __forceinline__ void hist_tile(const IdxT* __restrict__ idx, const uint8_t* __restrict__ ivalid, int64_t ioff, int64_t nidx,
                                          uint64_t nvalues, int shift, int64_t tile, unsigned* s_cnt, unsigned long long* __restrict__ first_bad) {
  unsigned u[kHistRows];
  bool live[kHistRows];
  unsigned long long oob;
  load_tile_indices<IdxT, kHistThreads, kHistRows, FULL, HAS_IV>(idx, ivalid, ioff, tile * kTile, nidx, nvalues, u, live, &oob);
  if (oob != ~0ull) atomicMin(first_bad, oob);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kHistRows; k++)
    if (FULL || live[k]) atomicAdd(&s_cnt[u[k] >> shift], 1u);
}
template <typename IdxT>
__global__ __launch_bounds__(kHistThreads) void bin_hist_kernel(const IdxT* __restrict__ idx, const uint8_t* __restrict__ ivalid, int64_t ioff,
                                                                 int64_t nidx, uint64_t nvalues, int shift, int nb, int64_t ntiles,
                                                                 unsigned* __restrict__ cnt_tm, unsigned long long* __restrict__ first_bad) {
  __shared__ unsigned s_cnt[kMaxBins];
  const int64_t tile = xcd_contiguous_tile(ntiles);
  if (tile < 0) return;
  for (int b = threadIdx.x; b < nb; b += kHistThreads) s_cnt[b] = 0;
  const bool full = (tile + 1) * kTile <= nidx;   // workgroup-uniform
  if (full) {
    if (ivalid) hist_tile<IdxT, true, true>(idx, ivalid, ioff, nidx, nvalues, shift, tile, s_cnt, first_bad);
    else hist_tile<IdxT, true, false>(idx, ivalid, ioff, nidx, nvalues, shift, tile, s_cnt, first_bad);
  } else {
    hist_tile<IdxT, false, true>(idx, ivalid, ioff, nidx, nvalues, shift, tile, s_cnt, first_bad);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kHistThreads) cnt_tm[tile * nb + b] = s_cnt[b];   // one contiguous row per tile
}

// ---- 2: records in bin order
template <typename IdxT>
__global__ __launch_bounds__(kThreads) void bin_scatter_kernel(const IdxT* __restrict__ idx, const uint8_t* __restrict__ ivalid, int64_t ioff,
                                                                int64_t nidx, uint64_t nvalues, int shift, int nb, int64_t ntiles,
                                                                const unsigned* __restrict__ toffs, unsigned* __restrict__ rec) {
  __shared__ unsigned s_cnt[kMaxBins], s_start[kMaxBins], s_goff[kMaxBins], s_wsum[kThreads / 64];
  __shared__ unsigned s_rec[kTile];
  __shared__ uint16_t s_bin[kTile];   // bin of each staged position (the record itself only keeps the index bits below `shift`)
  const int64_t tile = xcd_contiguous_tile(ntiles);
  if (tile < 0) return;
  s_cnt[threadIdx.x] = 0;
  const int64_t base = tile * kTile;
  const unsigned low_mask = (1u << shift) - 1u;
  unsigned u[kRowsPerThread], rank[kRowsPerThread];
  bool live[kRowsPerThread];
  unsigned long long oob;
  if ((tile + 1) * kTile <= nidx) {   // workgroup-uniform
    if (ivalid) load_tile_indices<IdxT, kThreads, kRowsPerThread, true, true>(idx, ivalid, ioff, base, nidx, nvalues, u, live, &oob);
    else load_tile_indices<IdxT, kThreads, kRowsPerThread, true, false>(idx, ivalid, ioff, base, nidx, nvalues, u, live, &oob);
  } else {
    load_tile_indices<IdxT, kThreads, kRowsPerThread, false, true>(idx, ivalid, ioff, base, nidx, nvalues, u, live, &oob);
  }
  unsigned goff_excl = 0;   // this tile's first record of bin t (global), fetched while the ranks are taken
  if ((int)threadIdx.x < nb) goff_excl = toffs[tile * nb + threadIdx.x];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kRowsPerThread; k++) rank[k] = live[k] ? atomicAdd(&s_cnt[u[k] >> shift], 1u) : 0u;
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  if ((int)threadIdx.x < nb) s_goff[threadIdx.x] = goff_excl - s_start[threadIdx.x];   // global record position = s_goff[bin] + staged position
#pragma unroll
  for (int k = 0; k < kRowsPerThread; k++) {
    if (live[k]) {
      const unsigned bin = u[k] >> shift, lp = s_start[bin] + rank[k];
      s_rec[lp] = (u[k] & low_mask) | (row_in_tile(k) << shift);
      s_bin[lp] = (uint16_t)bin;
    }
  }
  __syncthreads();
  const int tile_n = nidx - base >= kTile ? kTile : (int)(nidx - base);
  // consecutive threads hold consecutive staged positions = consecutive records of one bin's run
#pragma unroll
  for (int k = 0; k < kRowsPerThread; k++) {
    const int lp = k * kThreads + threadIdx.x;
    if (lp < tile_n) rec[(int64_t)s_goff[s_bin[lp]] + lp] = s_rec[lp];
  }
}

// ---- 3: gather inside the bins' windows
// (launched with `pad` bytes of dynamic LDS it never touches: that caps the workgroups per CU, i.e. how many windows an
// XCD has open at once — the records in flight on an XCD span that many windows of its 4 MiB L2)
template <int W, bool HAS_VALID>
__global__ __launch_bounds__(kGatherBlock) void bin_gather_kernel(const void* __restrict__ values_v, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                                  const unsigned* __restrict__ rec, const unsigned* __restrict__ binstart, int shift,
                                                                  int nb, int64_t nidx, int64_t nchunks, int seg_chunks, int ldmode,
                                                                  void* __restrict__ gval_v, unsigned long long* __restrict__ gvalid) {
  using T = typename UIntW<W>::type;
  const T* __restrict__ values = (const T*)values_v;
  T* __restrict__ gval = (T*)gval_v;
  __shared__ unsigned s_bs[kMaxBins + 1];
  // XCD x works through segments x, x + 8, … of seg_chunks consecutive chunks (≈ one bin each)
  const int64_t j = blockIdx.x >> 3;
  const int64_t chunk = ((j / seg_chunks) * 8 + (blockIdx.x & 7)) * seg_chunks + j % seg_chunks;
  if (chunk >= nchunks) return;
  for (int b = threadIdx.x; b <= nb; b += kGatherBlock) s_bs[b] = binstart[b];
  __syncthreads();
  const int64_t e0 = chunk * kGatherChunk;
  int bin0 = -1;  // bin of the chunk's first record = (number of bin starts ≤ e0) − 1; the same in every thread
  for (int b0 = 0; b0 < nb; b0 += kGatherBlock) {
    const int b = b0 + threadIdx.x;
    bin0 += __syncthreads_count(b < nb && s_bs[b] <= (unsigned)e0);
  }
  const unsigned low_mask = (1u << shift) - 1u;
  T v[kGatherPerThread];
  bool ok[kGatherPerThread];
  unsigned u[kGatherPerThread];
  // all the records first, then their bins: with the bin walk (a data-dependent LDS loop) between two loads the compiler waited
  // for each record before requesting the next — eight memory round trips in a row in front of every workgroup's gathers
  unsigned r[kGatherPerThread];
#pragma unroll
  for (int k = 0; k < kGatherPerThread; k++) {
    const int64_t e = e0 + k * kGatherBlock + threadIdx.x;
    ok[k] = e < nidx;
    r[k] = ok[k] ? __builtin_nontemporal_load(&rec[e]) : 0u;
  }
#pragma unroll
  for (int k = 0; k < kGatherPerThread; k++) {
    const int64_t e = e0 + k * kGatherBlock + threadIdx.x;
    u[k] = 0;
    if (ok[k]) {
      int bin = bin0;
      while (bin + 1 < nb && s_bs[bin + 1] <= (unsigned)e) bin++;
      u[k] = ((unsigned)bin << shift) | (r[k] & low_mask);
    }
  }
  // value and validity bit are fetched side by side (u is in range for every live record; dead lanes read element 0)
  uint8_t vb[kGatherPerThread];
#pragma unroll
  for (int k = 0; k < kGatherPerThread; k++) {
    if (ldmode == 1) v[k] = __builtin_nontemporal_load(&values[u[k]]);
    else if (ldmode == 2) v[k] = __hip_atomic_load(&values[u[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else v[k] = values[u[k]];
    vb[k] = (HAS_VALID && vvalid != nullptr) ? vvalid[(voff + (int64_t)u[k]) >> 3] : (uint8_t)0xff;
  }
#pragma unroll
  for (int k = 0; k < kGatherPerThread; k++) {
    if (HAS_VALID && !((vb[k] >> ((voff + (int64_t)u[k]) & 7)) & 1)) ok[k] = false;
    if (!ok[k]) v[k] = 0;
  }
#pragma unroll
  for (int k = 0; k < kGatherPerThread; k++) {
    const int64_t e = e0 + k * kGatherBlock + threadIdx.x;
    if (e < nidx) __builtin_nontemporal_store(v[k], &gval[e]);
    if (HAS_VALID) {
      const unsigned long long word = __ballot(ok[k]);
      if ((threadIdx.x & 63) == 0 && e < nidx) gvalid[e >> 6] = word;  // e is a multiple of 64 for lane 0
    }
  }
}

// ---- 3': the same with the window's validity bits in LDS -------------------------------------------------------------
// The value's validity bit doubled the gathers of pass 3 (1.11 → 1.56 ms for 2^27 indices): a second address per lane for one
// bit.  A window of 2^19 values has 64 KiB of validity bits — they fit LDS.  A workgroup (1024 threads) works through 32
// consecutive chunks (64 Ki records, almost always of ONE bin), keeps that bin's bits in LDS and reads them there; the rare
// step that straddles two bins takes the bits from memory as before.
constexpr int kLdsGatherChunks = 4;                      // chunks per workgroup: with more, the workgroups an XCD runs at one time spread over several bins — several 4 MiB windows — and its L2 holds one (32 chunks: 2.7 → 3.2 ms)
constexpr int kLdsStepChunks = kThreads / kGatherBlock;   // 4 chunks = 8192 records per step
template <int W>
__global__ __launch_bounds__(kThreads) void bin_gather_lds_kernel(const void* __restrict__ values_v, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                                   int64_t nvalues, const unsigned* __restrict__ rec,
                                                                   const unsigned* __restrict__ binstart, int shift, int nb, int64_t nidx,
                                                                   int64_t nunits, int seg_units, void* __restrict__ gval_v,
                                                                   unsigned long long* __restrict__ gvalid) {
  using T = typename UIntW<W>::type;
  const T* __restrict__ values = (const T*)values_v;
  T* __restrict__ gval = (T*)gval_v;
  __shared__ unsigned s_bs[kMaxBins + 1];
  __shared__ unsigned s_bits[(1 << 19) / 32 + 2];
  // XCD x works through segments x, x + 8, … of seg_units consecutive units (≈ one bin each)
  const int64_t j = blockIdx.x >> 3;
  const int64_t unit = ((j / seg_units) * 8 + (blockIdx.x & 7)) * seg_units + j % seg_units;
  if (unit >= nunits) return;
  for (int b = threadIdx.x; b <= nb; b += kThreads) s_bs[b] = binstart[b];
  __syncthreads();
  auto bin_of = [&](int64_t e) {   // largest b with s_bs[b] ≤ e (the same in every thread: LDS broadcasts)
    int lo = 0, hi = nb - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_bs[mid] <= (unsigned)e) lo = mid; else hi = mid - 1; }
    return lo;
  };
  const unsigned low_mask = (1u << shift) - 1u;
  int cached = -1;
  int64_t base_bit = 0;
  for (int it = 0; it < kLdsGatherChunks / kLdsStepChunks; it++) {
    const int64_t e0 = (unit * kLdsGatherChunks + (int64_t)it * kLdsStepChunks) * kGatherChunk;
    if (e0 >= nidx) break;
    const int64_t e_last = (e0 + (int64_t)kLdsStepChunks * kGatherChunk < nidx ? e0 + (int64_t)kLdsStepChunks * kGatherChunk : nidx) - 1;
    const int b_first = bin_of(e0), b_last = bin_of(e_last);
    const bool pure = b_first == b_last;
    if (pure && cached != b_first) {
      __syncthreads();   // the previous step's readers are done
      const int64_t v0 = (int64_t)b_first << shift, v1 = v0 + ((int64_t)1 << shift) < nvalues ? v0 + ((int64_t)1 << shift) : nvalues;
      base_bit = (voff + v0) & ~(int64_t)31;
      const int64_t nwords = (voff + v1 - base_bit + 31) >> 5;
      const uint8_t* src = vvalid + (base_bit >> 3);
      const int64_t nbytes = ((voff + v1 + 7) >> 3) - (base_bit >> 3);   // never past the bitmap's last byte
      for (int64_t w = threadIdx.x; w < nwords; w += kThreads) {
        unsigned x = 0;
        if (w * 4 + 4 <= nbytes) memcpy(&x, src + w * 4, 4);
        else for (int q = 0; q < 4 && w * 4 + q < nbytes; q++) x |= (unsigned)src[w * 4 + q] << (8 * q);
        s_bits[w] = x;
      }
      cached = b_first;
      __syncthreads();
    }
    T v[kGatherPerThread];
    bool ok[kGatherPerThread];
    unsigned u[kGatherPerThread];
    unsigned r[kGatherPerThread];   // all the records first, then their bins (see bin_gather_kernel)
#pragma unroll
    for (int k = 0; k < kGatherPerThread; k++) {
      const int64_t e = e0 + k * kThreads + threadIdx.x;
      ok[k] = e < nidx;
      r[k] = ok[k] ? __builtin_nontemporal_load(&rec[e]) : 0u;
    }
#pragma unroll
    for (int k = 0; k < kGatherPerThread; k++) {
      const int64_t e = e0 + k * kThreads + threadIdx.x;
      u[k] = 0;
      if (ok[k]) {
        int bin = b_first;
        if (!pure) while (bin + 1 < nb && s_bs[bin + 1] <= (unsigned)e) bin++;
        u[k] = ((unsigned)bin << shift) | (r[k] & low_mask);
      }
    }
#pragma unroll
    for (int k = 0; k < kGatherPerThread; k++) v[k] = values[u[k]];
#pragma unroll
    for (int k = 0; k < kGatherPerThread; k++) {
      bool bit;
      if (pure) { const int64_t p = voff + (int64_t)u[k] - base_bit; bit = (s_bits[p >> 5] >> (p & 31)) & 1u; }
      else bit = (vvalid[(voff + (int64_t)u[k]) >> 3] >> ((voff + (int64_t)u[k]) & 7)) & 1;
      ok[k] = ok[k] && bit;
      if (!ok[k]) v[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < kGatherPerThread; k++) {
      const int64_t e = e0 + k * kThreads + threadIdx.x;
      if (e < nidx) __builtin_nontemporal_store(v[k], &gval[e]);
      const unsigned long long word = __ballot(ok[k]);
      if ((threadIdx.x & 63) == 0 && e < nidx) gvalid[e >> 6] = word;  // e is a multiple of 64 for lane 0
    }
  }
}

// ---- 4: back to row order
template <int W, bool HAS_VALID>
__global__ __launch_bounds__(kThreads) void unpermute_kernel(const void* __restrict__ gval_v, const unsigned long long* __restrict__ gvalid,
                                                              const unsigned* __restrict__ rec, const unsigned* __restrict__ cnt_tm,
                                                              const unsigned* __restrict__ toffs, int shift, int nb, int64_t ntiles, int64_t nidx,
                                                              const uint8_t* __restrict__ ivalid, int64_t ioff, void* __restrict__ out_v,
                                                              uint8_t* __restrict__ out_valid) {
  using T = typename UIntW<W>::type;
  const T* __restrict__ gval = (const T*)gval_v;
  T* __restrict__ out = (T*)out_v;
  __shared__ unsigned s_cnt[kMaxBins], s_start[kMaxBins], s_goff[kMaxBins], s_wsum[kThreads / 64];
  __shared__ T s_val[kTile];
  __shared__ unsigned s_ok[kTile / 32];
  const int64_t tile = xcd_contiguous_tile(ntiles);
  if (tile < 0) return;
  unsigned excl = 0;
  if ((int)threadIdx.x < nb) {
    s_cnt[threadIdx.x] = cnt_tm[tile * nb + threadIdx.x];
    excl = toffs[tile * nb + threadIdx.x];
  }
  if (HAS_VALID && threadIdx.x < kTile / 32) s_ok[threadIdx.x] = 0;
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  if ((int)threadIdx.x < nb) s_goff[threadIdx.x] = excl - s_start[threadIdx.x];
  __syncthreads();
  const int64_t base = tile * kTile;
  const int tile_n = nidx - base >= kTile ? kTile : (int)(nidx - base);
  // staged position lp → its bin = the LAST b with s_start[b] ≤ lp (empty bins share their start with the next one): the
  // kRowsPerThread searches of a thread are independent, so their LDS reads overlap; then all loads go out together
  int lo[kRowsPerThread], hi[kRowsPerThread];
#pragma unroll
  for (int k = 0; k < kRowsPerThread; k++) { lo[k] = 0; hi[k] = nb - 1; }
#pragma unroll 1
  for (int step = 0; step < 10; step++) {   // 2^10 = kMaxBins
#pragma unroll
    for (int k = 0; k < kRowsPerThread; k++) {
      const int mid = (lo[k] + hi[k] + 1) >> 1;
      const bool le = s_start[mid] <= (unsigned)(k * kThreads + threadIdx.x);
      lo[k] = le ? mid : lo[k];
      hi[k] = le ? hi[k] : mid - 1;
    }
  }
  unsigned r[kRowsPerThread];
  T v[kRowsPerThread];
  bool ok[kRowsPerThread];
#pragma unroll
  for (int k = 0; k < kRowsPerThread; k++) {
    const int lp = k * kThreads + threadIdx.x;
    r[k] = 0; v[k] = 0; ok[k] = false;
    if (lp < tile_n) {
      const int64_t e = (int64_t)s_goff[lo[k]] + lp;
      r[k] = __builtin_nontemporal_load(&rec[e]);
      v[k] = __builtin_nontemporal_load(&gval[e]);
      ok[k] = HAS_VALID ? (bool)((gvalid[e >> 6] >> (e & 63)) & 1ull) : true;
    }
  }
#pragma unroll
  for (int k = 0; k < kRowsPerThread; k++) {
    const int lp = k * kThreads + threadIdx.x;
    if (lp < tile_n) {
      const unsigned row = r[k] >> shift;
      s_val[row] = v[k];
      if (HAS_VALID && ok[k]) atomicOr(&s_ok[row >> 5], 1u << (row & 31));
    }
  }
  __syncthreads();
  if (HAS_VALID) {
    // index validity: a null index slot was gathered through index 0 — clear it, payload 0
    const int wd = threadIdx.x;
    if (wd < (tile_n + 31) / 32) {
      const int nbits = tile_n - wd * 32 >= 32 ? 32 : tile_n - wd * 32;
      unsigned iv = 0xffffffffu;
      if (ivalid != nullptr) iv = (unsigned)ah_load_bits64(ivalid, ioff + base + (int64_t)wd * 32, nbits);
      else if (nbits < 32) iv = (1u << nbits) - 1u;
      const unsigned okw = s_ok[wd] & iv;
      s_ok[wd] = okw;
      uint8_t* p = out_valid + ((base + (int64_t)wd * 32) >> 3);
      const int nbytes = (nbits + 7) >> 3;
      if (nbytes == 4) *(unsigned*)p = okw;
      else for (int bb = 0; bb < nbytes; bb++) p[bb] = (uint8_t)(okw >> (8 * bb));
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < kRowsPerThread; k++) {
    const int i = k * kThreads + threadIdx.x;
    if (i < tile_n) {
      T val = s_val[i];
      if (HAS_VALID && !((s_ok[i >> 5] >> (i & 31)) & 1u)) val = 0;
      __builtin_nontemporal_store(val, &out[base + i]);
    }
  }
}

struct Plan {
  int shift, nb;
  int64_t ntiles;
};

template <typename IdxT>
int run_front(ah_ctx* c, const Plan& p, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, int64_t nvalues, unsigned* cnt_tm,
              unsigned* toffs, unsigned* gsum, unsigned* binstart, unsigned* rec, unsigned long long* first_bad) {
  const unsigned grid = (unsigned)(((p.ntiles + 7) / 8) * 8);
  const int64_t ngroups = ah_ceil_div(p.ntiles, kGroupTiles);
  bin_hist_kernel<IdxT><<<grid, kHistThreads, 0, c->stream>>>((const IdxT*)idx, ivalid, ioff, nidx, (uint64_t)nvalues, p.shift, p.nb, p.ntiles, cnt_tm, first_bad);
  AH_LAUNCH_CHECK(c);
  colsum_kernel<<<(unsigned)ngroups, kMaxBins, 0, c->stream>>>(cnt_tm, p.nb, p.ntiles, gsum);
  AH_LAUNCH_CHECK(c);
  bin_prefix_kernel<<<1, kMaxBins, 0, c->stream>>>(gsum, p.nb, ngroups, nidx, binstart);
  AH_LAUNCH_CHECK(c);
  tile_offs_kernel<<<(unsigned)ngroups, kMaxBins, 0, c->stream>>>(cnt_tm, gsum, p.nb, p.ntiles, toffs);
  AH_LAUNCH_CHECK(c);
  bin_scatter_kernel<IdxT><<<grid, kThreads, 0, c->stream>>>((const IdxT*)idx, ivalid, ioff, nidx, (uint64_t)nvalues, p.shift, p.nb, p.ntiles, toffs, rec);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <int W>
int run_back(ah_ctx* c, const Plan& p, const void* values, const uint8_t* vvalid, int64_t voff, int64_t nvalues, const uint8_t* ivalid, int64_t ioff, int64_t nidx,
             const unsigned* cnt_tm, const unsigned* toffs, const unsigned* binstart, const unsigned* rec, void* gval, unsigned long long* gvalid,
             void* out_values, uint8_t* out_valid) {
  const int64_t nchunks = ah_ceil_div(nidx, kGatherChunk);
  int64_t seg = nchunks / p.nb;  // chunks per bin for evenly spread indices
  if (seg < 1) seg = 1;
  const int64_t rounds = ah_ceil_div(nchunks, 8 * seg);
  const unsigned ggrid = (unsigned)(rounds * 8 * seg);
  const unsigned tgrid = (unsigned)(((p.ntiles + 7) / 8) * 8);
  const int gather_wg_per_cu = c->opt_take_gather_wg;
  const size_t lds_pad = gather_wg_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / (gather_wg_per_cu < 1 ? 1 : gather_wg_per_cu)) - 8192;
  if (out_valid && vvalid && c->opt_take_gather_lds && p.shift <= 19) {
    const int64_t nunits = ah_ceil_div(nchunks, kLdsGatherChunks);
    int64_t seg_units = seg / kLdsGatherChunks;
    if (seg_units < 1) seg_units = 1;
    const unsigned ugrid = (unsigned)(ah_ceil_div(nunits, 8 * seg_units) * 8 * seg_units);
    bin_gather_lds_kernel<W><<<ugrid, kThreads, 0, c->stream>>>(values, vvalid, voff, nvalues, rec, binstart, p.shift, p.nb, nidx, nunits, (int)seg_units, gval, gvalid);
    AH_LAUNCH_CHECK(c);
    unpermute_kernel<W, true><<<tgrid, kThreads, 0, c->stream>>>(gval, gvalid, rec, cnt_tm, toffs, p.shift, p.nb, p.ntiles, nidx, ivalid, ioff, out_values, out_valid);
  } else if (out_valid) {
    bin_gather_kernel<W, true><<<ggrid, kGatherBlock, lds_pad, c->stream>>>(values, vvalid, voff, rec, binstart, p.shift, p.nb, nidx, nchunks, (int)seg, c->opt_take_gather_load, gval, gvalid);
    AH_LAUNCH_CHECK(c);
    unpermute_kernel<W, true><<<tgrid, kThreads, 0, c->stream>>>(gval, gvalid, rec, cnt_tm, toffs, p.shift, p.nb, p.ntiles, nidx, ivalid, ioff, out_values, out_valid);
  } else {
    bin_gather_kernel<W, false><<<ggrid, kGatherBlock, lds_pad, c->stream>>>(values, nullptr, voff, rec, binstart, p.shift, p.nb, nidx, nchunks, (int)seg, c->opt_take_gather_load, gval, gvalid);
    AH_LAUNCH_CHECK(c);
    unpermute_kernel<W, false><<<tgrid, kThreads, 0, c->stream>>>(gval, gvalid, rec, cnt_tm, toffs, p.shift, p.nb, p.ntiles, nidx, nullptr, ioff, out_values, nullptr);
  }
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

}  // namespace

// Called by ah_take_primitive before the direct kernel.  *used = 1: the binned path produced out_values / out_valid
// and published the first out-of-range position in *first_bad exactly like the direct kernel; 0: not applicable
// (small, clustered, or too wide) — nothing was written.
int ah_take_binned_try(ah_ctx* c, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff, int64_t nvalues, int iw,
                       int is_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, void* out_values,
                       uint8_t* out_valid, unsigned long long* first_bad, int* used) {
  *used = 0;
  const int mode = c->opt_take_binned;               // 0 never, 1 auto, 2 whenever legal
  const int window_log2 = c->opt_take_window_log2;   // bytes of `values` per bin (default 4 MiB: an XCD's L2; 1–4 MiB measured within 5 %)
  if (mode == 0 || (byte_width != 1 && byte_width != 2 && byte_width != 4 && byte_width != 8)) return AH_OK;   // (16- / 32-byte and odd-width slots: the plain gather kernels)
  const int64_t vbytes = nvalues * byte_width;
  if (nvalues < 1 || nvalues > ((int64_t)1 << 32) - 1 || nidx > ((int64_t)1 << 32) - 1) return AH_OK;
  if (mode == 1) {
    // only where it pays: a column the caches cannot hold, enough indices that its lines are touched more than once
    if (vbytes < ((int64_t)64 << 20) || nidx < ((int64_t)1 << 20) || nidx * 8 < nvalues) return AH_OK;
  } else if (nidx < kTile) {
    return AH_OK;
  }
  Plan p;
  int lg = 0;
  while (((int64_t)1 << lg) < nvalues) lg++;
  int wl = 0;
  while ((1 << wl) < byte_width) wl++;
  int shift = window_log2 - wl;                       // elements per window
  if (shift > 32 - kTileBits) shift = 32 - kTileBits;  // the record holds shift + 13 bits
  if (shift < 1) shift = 1;
  while (lg - shift > 10) shift++;                     // ≤ 1024 bins
  if (shift > 32 - kTileBits) return AH_OK;            // column too long for 1024 windows of ≤ 2^19 elements
  p.shift = shift;
  p.nb = (int)(((nvalues - 1) >> shift) + 1);
  p.ntiles = ah_ceil_div(nidx, kTile);
  if (mode == 1) {
    // clustered / sorted / reversed indices stream through the direct kernel at the copy rate — sample 64 × 255 neighbours
    unsigned long long* hits = (unsigned long long*)&c->dscalars[3];
    const int near = 128 / byte_width, kind = iw * 2 + (is_signed ? 1 : 0);
    uint64_t word;
    if (c->opt_take_hint_cache && c->take_hint_live && !c->capturing && c->take_hint_idx == idx && c->take_hint_nidx == nidx && c->take_hint_kind == kind && c->take_hint_near == near &&
        c->take_hint_uses < 32) {
      word = c->take_hint_word;   // the same index vector as the last call's (ah_common.h: speed only, never results)
      c->take_hint_uses++;
    } else {
      AH_HIP(c, hipMemsetAsync(hits, 0, sizeof(*hits), c->stream));
#define AH_S(IT) sample_kernel<IT><<<64, 256, 0, c->stream>>>((const IT*)idx, nidx, near, hits); break
      switch (kind) {
        case 2: AH_S(uint8_t); case 3: AH_S(int8_t); case 4: AH_S(uint16_t); case 5: AH_S(int16_t);
        case 8: AH_S(uint32_t); case 9: AH_S(int32_t); case 16: AH_S(uint64_t); case 17: AH_S(int64_t);
        default: return AH_OK;
      }
#undef AH_S
      AH_LAUNCH_CHECK(c);
      { int mrc = ah_mailbox_read(c, hits, 1, (unsigned long long*)&c->pinned[8]); if (mrc != AH_OK) return mrc; }
      word = *(volatile uint64_t*)&c->pinned[8];
      c->take_hint_idx = idx; c->take_hint_nidx = nidx; c->take_hint_kind = kind; c->take_hint_near = near; c->take_hint_word = word; c->take_hint_uses = 0;
    }
    const uint64_t h = word & 0xffffffffull, h1 = word >> 32;
    if (h * 4 > 64 * 255) {  // more than a quarter of the sampled neighbours sit within one 128-byte line
      // … and where most neighbours name ADJACENT values the direct path takes V rows per lane with merged 16-byte accesses
      // (ah_take.hip).  Merely near ones (a sorted random draw: 37 % repeats, 37 % adjacent, 26 % further) keep one row per lane:
      // every wave would run both the merged and the separate gathers (measured 4.35 → 3.97 TB/s)
      if (c->opt_take_vec && h1 * 10 > 64 * 255 * 8) c->take_clustered_hint = 1;
      return AH_OK;
    }
  }
  // temporaries
  const size_t table = (size_t)p.nb * (size_t)p.ntiles * sizeof(unsigned);
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t need = pad(table) * 2 + pad((size_t)ah_ceil_div(p.ntiles, kGroupTiles) * p.nb * 4) + pad((size_t)(p.nb + 1) * 4) + pad((size_t)nidx * 4) + pad((size_t)nidx * byte_width) + pad((size_t)(nidx / 64 + 2) * 8);
  uint8_t* base;
  int rc = ah_temp_reserve(c, need, (void**)&base);
  if (rc != AH_OK) return rc;
  size_t used_b = 0;
  auto take = [&](size_t b) { uint8_t* q = base + used_b; used_b += pad(b); return q; };
  unsigned* cnt_tm = (unsigned*)take(table);
  unsigned* toffs = (unsigned*)take(table);
  unsigned* gsum = (unsigned*)take((size_t)ah_ceil_div(p.ntiles, kGroupTiles) * p.nb * 4);
  unsigned* binstart = (unsigned*)take((size_t)(p.nb + 1) * 4);
  unsigned* rec = (unsigned*)take((size_t)nidx * 4);
  void* gval = take((size_t)nidx * byte_width);
  unsigned long long* gvalid = (unsigned long long*)take((size_t)(nidx / 64 + 2) * 8);
#define AH_F(IT) rc = run_front<IT>(c, p, idx, ivalid, ioff, nidx, nvalues, cnt_tm, toffs, gsum, binstart, rec, first_bad); break
  switch (iw * 2 + (is_signed ? 1 : 0)) {
    case 2: AH_F(uint8_t); case 3: AH_F(int8_t); case 4: AH_F(uint16_t); case 5: AH_F(int16_t);
    case 8: AH_F(uint32_t); case 9: AH_F(int32_t); case 16: AH_F(uint64_t); case 17: AH_F(int64_t);
    default: return AH_OK;
  }
#undef AH_F
  if (rc != AH_OK) return rc;
  switch (byte_width) {
    case 1: rc = run_back<1>(c, p, values, vvalid, voff, nvalues, ivalid, ioff, nidx, cnt_tm, toffs, binstart, rec, gval, gvalid, out_values, out_valid); break;
    case 2: rc = run_back<2>(c, p, values, vvalid, voff, nvalues, ivalid, ioff, nidx, cnt_tm, toffs, binstart, rec, gval, gvalid, out_values, out_valid); break;
    case 4: rc = run_back<4>(c, p, values, vvalid, voff, nvalues, ivalid, ioff, nidx, cnt_tm, toffs, binstart, rec, gval, gvalid, out_values, out_valid); break;
    case 8: rc = run_back<8>(c, p, values, vvalid, voff, nvalues, ivalid, ioff, nidx, cnt_tm, toffs, binstart, rec, gval, gvalid, out_values, out_valid); break;
    default: return AH_OK;
  }
  if (rc != AH_OK) return rc;
  *used = 1;
  return AH_OK;
}
