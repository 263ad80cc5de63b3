// ah_filter.hip — Filter: stable stream compaction of fixed-width values by a
// boolean selection vector, with Arrow null semantics.
//
// Replaces: kernels.PrimitiveFilter (kernels/vector_selection.go:449-520) =
//   getFilterOutputSize (:57-81) + preallocateData (:83-93) + primitiveFilterImpl
//   (:267-395) + filterWriter (:397-421) over the bit-block counters
//   (internal/bitutils/bit_block_counter.go:59-168,357-452), and GetTakeIndices
//   (:102-236), behind compute's "filter"/"array_filter" (compute/selection.go:42-85,
//   618-633).
//
// Payload contract (bit-exact, from reading the reference — SURVEY.md §8a a7):
//   slot selected (filter valid ∧ true)  → value payload copied as is, out validity
//                                           = value validity            (:293-297)
//   filter slot null under EMIT_NULLS     → payload 0, validity 0       (:417-421)
//   otherwise dropped; padding past n_out = 0 (fresh zeroed buffers).
//
// Three launches, all HBM-bound:
//   1. tile_count_kernel  — popcount of the selection word per tile (reads only the
//      bitmaps: n/8 bytes, 1/64 of the values); tile prefixes inside 256-tile super tiles
//   2. super_scan_kernel  — exclusive scan of the (n / 2^16) super-tile totals, one block
//   3. compact_kernel     — one workgroup per tile of 16 KiB of values: the tile's
//      mask words and their running popcounts live in LDS, every lane derives the
//      output rank of its elements with two popcounts (no ballot needed), loads its
//      16-byte vector ONLY if it holds a survivor (so unselected 128-byte lines are
//      never fetched), stages survivors in LDS at their rank, and the workgroup
//      streams the staged run to HBM as aligned 16-byte stores.  The output
//      validity word for 64 input rows is a software bit-compress (pext) of the
//      value-validity word by the selection word, OR-ed into place.
// Algorithmic bytes per input row (w = 8): 8 + 1/8 (+1/8 with value validity) read,
// 8·s (+ s/8) written for selectivity s.
#include <chrono>
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;
#ifndef AH_FILTER_TILE_BYTES
#define AH_FILTER_TILE_BYTES 16384
#endif
constexpr int kTileBytesMax = AH_FILTER_TILE_BYTES;  // values staged per workgroup (W = 1 is capped at 16 KiB: one lane per mask word)
template <int W> constexpr int TileBytes() { return (W == 1 && kTileBytesMax > 16384) ? 16384 : kTileBytesMax; }

template <int W> struct UIntOf;
template <> struct UIntOf<1> { using type = uint8_t; };
template <> struct UIntOf<2> { using type = uint16_t; };
template <> struct UIntOf<4> { using type = uint32_t; };
template <> struct UIntOf<8> { using type = uint64_t; };

// selection word for 64 consecutive rows starting at row `pos` (cnt valid rows):
//   DROP: data ∧ valid          EMIT: data ∨ ¬valid  (vector_selection.go:66-77)
__device__ __forceinline__ uint64_t sel_word(const uint8_t* __restrict__ fdata, const uint8_t* __restrict__ fvalid,
                                             int64_t foff, int64_t pos, int cnt, int null_sel, uint64_t* fv_out) {
  uint64_t mask = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1);
  uint64_t fd = ah_load_bits64(fdata, foff + pos, cnt);
  uint64_t fv = ah_load_bits64(fvalid, foff + pos, cnt);  // NULL → all ones
  if (fv_out) *fv_out = fv;
  return (null_sel == AH_EMIT_NULLS ? (fd | ~fv) : (fd & fv)) & mask;
}

// Hacker's Delight 7-4 "compress" (software pext): gather the bits of x selected
// by m into the low popcount(m) bits, order preserved.
__device__ __forceinline__ uint64_t bit_compress(uint64_t x, uint64_t m) {
  x &= m;
  uint64_t mk = ~m << 1;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    uint64_t mp = mk ^ (mk << 1);
    mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16; mp ^= mp << 32;
    uint64_t mv = mp & m;
    m = (m ^ mv) | (mv >> (1 << i));
    uint64_t t = x & mv;
    x = (x ^ t) | (t >> (1 << i));
    mk &= ~mp;
  }
  return x;
}

// exclusive scan of one int per thread across a 256-thread block
__device__ __forceinline__ int block_exclusive_scan(int v, int* total) {
  __shared__ int wave_tot[kBlock / 64];
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kBlock / 64; k++) {
    int t = wave_tot[k];
    if (k < wave) base += t;
    tot += t;
  }
  *total = tot;
  __syncthreads();
  return base + inc - v;
}

// ---- 1. per-tile survivor counts -------------------------------------------------
// One workgroup covers kSuper = 32 consecutive tiles (a "super tile"; 2048 workgroups for
// 2^27 rows — 256-tile super tiles left one workgroup per CU and made this pass
// latency-bound at 25 µs).  It writes, per
// tile, the number of survivors in the tiles BEFORE it inside the super tile
// (tile_local[tile]) and, per super tile, its total (super_total[s]).  The mask words
// are read coalesced: in iteration j lane t reads word j*256 + t of the super tile.
constexpr int kSuper = 32;

template <int WPT /*mask words per tile*/>
__global__ __launch_bounds__(kBlock) void tile_count_kernel(const uint8_t* __restrict__ fdata, const uint8_t* __restrict__ fvalid,
                                                             int64_t foff, int64_t n, int null_sel, int* __restrict__ tile_local,
                                                             int* __restrict__ super_total, int64_t ntiles) {
  __shared__ int s_cnt[kSuper];
  const int tid = threadIdx.x;
  const int64_t super = blockIdx.x;
  const int64_t word0 = super * (int64_t)kSuper * WPT;  // first mask word of the super tile
  if (tid < kSuper) s_cnt[tid] = 0;
  __syncthreads();
  // tiles are WPT consecutive words: iteration j covers words [j*256, j*256+256) =
  // tiles [j*256/WPT, ...) — each lane adds its popcount into its tile's LDS counter
  constexpr int ITERS = kSuper * WPT / kBlock;
  static_assert(kSuper * WPT % kBlock == 0, "super tile must be a whole number of block passes");
#pragma unroll
  for (int j = 0; j < ITERS; j++) {
    const int wl = j * kBlock + tid;  // word index inside the super tile
    const int64_t pos = (word0 + wl) * 64;
    int v = 0;
    if (pos < n) {
      int cnt = n - pos >= 64 ? 64 : (int)(n - pos);
      v = __popcll(sel_word(fdata, fvalid, foff, pos, cnt, null_sel, nullptr));
    }
    // reduce over the lanes that share a tile (WPT consecutive lanes, WPT ≤ 64 → shuffles;
    // WPT > 64 → whole waves share one tile)
    constexpr int SEG = WPT < 64 ? WPT : 64;
#pragma unroll
    for (int o = SEG / 2; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((tid % SEG) == 0 && v) atomicAdd(&s_cnt[wl / WPT], v);
  }
  __syncthreads();
  int total;
  int c = tid < kSuper ? s_cnt[tid] : 0;
  int excl = block_exclusive_scan(c, &total);
  const int64_t tile = super * kSuper + tid;
  if (tid < kSuper && tile < ntiles) tile_local[tile] = excl;
  if (tid == 0) super_total[super] = total;
}

// ---- 2. exclusive scan of the super-tile totals (one block; ≤ a few thousand entries) -
// mailbox (optional): coherent pinned HOST memory — {total, seq} stored with system scope, so the host that needs the count to go
// on (ah_filter_count) can poll for it instead of waiting for the stream
__global__ __launch_bounds__(1024) void super_scan_kernel(const int* __restrict__ super_total, int64_t nsuper,
                                                           int64_t* __restrict__ super_off, int64_t* __restrict__ total,
                                                           unsigned long long* mailbox, unsigned long long seq) {
  __shared__ int64_t wave_tot[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nsuper; base += 1024) {
    const int64_t i = base + tid;
    int64_t v = i < nsuper ? super_total[i] : 0;
    int64_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int64_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int64_t b = carry_s, tot = 0;
    for (int k = 0; k < 16; k++) {
      int64_t t = wave_tot[k];
      if (k < wave) b += t;
      tot += t;
    }
    if (i < nsuper) super_off[i] = b + inc - v;
    __syncthreads();
    if (tid == 0) carry_s += tot;
    __syncthreads();
  }
  if (tid == 0) {
    *total = carry_s;
    if (mailbox) {
      __hip_atomic_store(&mailbox[0], (unsigned long long)carry_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&mailbox[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- 3. compaction -------------------------------------------------------------------
// W: value byte width.  HAS_VALID: an output validity bitmap exists (some input has
// nulls).  INDICES: payload is the row number (GetTakeIndices), values unused.
// NT: values in and survivors out are streamed once (nontemporal) — chosen for columns the caches cannot hold anyway.
template <int W, bool HAS_VALID, bool INDICES, bool NT>
__global__ __launch_bounds__(kBlock) void compact_kernel(const void* __restrict__ values_v, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                          const uint8_t* __restrict__ fdata, const uint8_t* __restrict__ fvalid,
                                                          int64_t foff, int64_t n, int null_sel,
                                                          const int64_t* __restrict__ super_off, const int* __restrict__ tile_local,
                                                          void* __restrict__ out_v, uint8_t* __restrict__ out_valid,
                                                          unsigned long long* __restrict__ valid_total) {
  using T = typename UIntOf<W>::type;
  constexpr int V = 16 / W;               // elements per 16-byte vector
  constexpr int TILE = TileBytes<W>() / W;    // rows per tile
  constexpr int WPT = TILE / 64;          // mask words per tile
  constexpr int VPT = TILE / V;           // vectors per tile (= 1024)
  constexpr int K = VPT / kBlock;         // vectors per lane (= 4)
  static_assert(WPT <= kBlock, "one lane per mask word");

  __shared__ __attribute__((aligned(16))) T stage[TILE];
  __shared__ uint64_t s_sel[WPT], s_nullsel[WPT], s_obits[WPT + 1];
  __shared__ int s_prefix[WPT];

  const T* __restrict__ values = (const T*)values_v;
  T* __restrict__ out = (T*)out_v;
  const int64_t tile = blockIdx.x;
  const int64_t b = tile * TILE;  // first row of the tile
  const int tid = threadIdx.x;

  // -- A: mask words, running popcounts, output validity bits
  uint64_t sel = 0, vsrc = 0, fvw = ~0ull;
  int cnt = 0;
  if (tid < WPT) {
    int64_t pos = b + (int64_t)tid * 64;
    cnt = pos >= n ? 0 : (n - pos >= 64 ? 64 : (int)(n - pos));
    if (cnt > 0) {
      sel = sel_word(fdata, fvalid, foff, pos, cnt, null_sel, &fvw);
      if (HAS_VALID) vsrc = sel & fvw & ah_load_bits64(INDICES ? nullptr : vvalid, voff + pos, cnt);
    }
    s_sel[tid] = sel;
    s_nullsel[tid] = sel & ~fvw;  // selected only because the filter slot is null (EMIT)
    s_obits[tid] = 0;
  }
  if (tid == 0) s_obits[WPT] = 0;
  int tile_count;
  int prefix = block_exclusive_scan(tid < WPT ? __popcll(sel) : 0, &tile_count);  // contains __syncthreads
  if (tid < WPT) s_prefix[tid] = prefix;
  if (HAS_VALID && tid < WPT && sel) {
    uint64_t ob = bit_compress(vsrc, sel);
    int c = __popcll(sel), w0 = prefix >> 6, sh = prefix & 63;
    if (ob) {
      atomicOr((unsigned long long*)&s_obits[w0], (unsigned long long)(ob << sh));
      if (sh && sh + c > 64) atomicOr((unsigned long long*)&s_obits[w0 + 1], (unsigned long long)(ob >> (64 - sh)));
    }
  }
  __syncthreads();
  if (tile_count == 0) return;

  // -- B: gather survivors into LDS at their rank.  All K (predicated) 16-byte loads of a
  // lane are issued before the first one is consumed: K·16 B in flight per lane.
  const int64_t rows_left = n - b;  // > 0
  ah_vec16<T> xv[K];
  unsigned bits_k[K];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int e0 = (k * kBlock + tid) * V;  // first row of the vector inside the tile
    bits_k[k] = (unsigned)((s_sel[e0 >> 6] >> (e0 & 63)) & ((1u << V) - 1));
    if (!INDICES && bits_k[k] != 0) {  // nothing selected here → the line is never fetched
      if (e0 + V <= rows_left) {
        xv[k] = NT ? ah_ld16_nt<T>(values + b + e0) : ah_ld16<T>(values + b + e0);
      } else {  // ragged end of the column: stay in bounds
#pragma unroll
        for (int e = 0; e < V; e++) xv[k].v[e] = (e0 + e < rows_left) ? values[b + e0 + e] : (T)0;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) {
    const unsigned bits = bits_k[k];
    if (bits == 0) continue;
    const int e0 = (k * kBlock + tid) * V;
    const int sh = e0 & 63;
    const uint64_t word = s_sel[e0 >> 6];
    const unsigned nbits = (unsigned)((s_nullsel[e0 >> 6] >> sh) & ((1u << V) - 1));
    int rank = s_prefix[e0 >> 6] + __popcll(word & ((1ull << sh) - 1));
#pragma unroll
    for (int e = 0; e < V; e++) {
      if (bits & (1u << e)) {
        T val = INDICES ? (T)(b + e0 + e) : xv[k].v[e];
        stage[rank] = (nbits & (1u << e)) ? (T)0 : val;
        rank++;
      }
    }
  }
  __syncthreads();

  // -- C: stream the staged run out (16-byte aligned body, element head/tail)
  const int64_t obase = super_off[tile / kSuper] + tile_local[tile];
  T* dst = out + obase;
  int head = (int)(((16 - ((uintptr_t)dst & 15)) & 15) / W);
  if (head > tile_count) head = tile_count;
  const int nvec = (tile_count - head) / V;
  const int tail0 = head + nvec * V;
  if (tid < head) dst[tid] = stage[tid];
  for (int i = tid; i < nvec; i += kBlock) {
    ah_vec16<T> x;
#pragma unroll
    for (int e = 0; e < V; e++) x.v[e] = stage[head + i * V + e];
    if (NT) ah_st16_nt<T>(dst + head + i * V, x);
    else ah_st16<T>(dst + head + i * V, x);
  }
  if (tid < tile_count - tail0) dst[tail0 + tid] = stage[tail0 + tid];

  if (HAS_VALID) {
    // output validity bits [obase, obase + tile_count): OR 64-bit chunks into place
    uintptr_t vaddr = (uintptr_t)out_valid;
    unsigned long long* vbase = (unsigned long long*)(vaddr & ~(uintptr_t)7);
    const int64_t bit0 = (int64_t)(vaddr & 7) * 8 + obase;
    const int nchunks = (tile_count + 63) >> 6;
    for (int i = tid; i < nchunks; i += kBlock) {
      uint64_t chunk = s_obits[i];
      if (!chunk) continue;
      int64_t g = bit0 + (int64_t)i * 64;
      int64_t w0 = g >> 6;
      int sh = (int)(g & 63);
      atomicOr(&vbase[w0], (unsigned long long)(chunk << sh));
      if (sh) {
        uint64_t hi = chunk >> (64 - sh);
        if (hi) atomicOr(&vbase[w0 + 1], (unsigned long long)hi);
      }
    }
  }
}

template <int W, bool HAS_VALID, bool INDICES>
void launch_compact(ah_ctx* c, int64_t ntiles, const void* values, const uint8_t* vvalid, int64_t voff, const uint8_t* fdata, const uint8_t* fvalid,
                    int64_t foff, int64_t n, int null_sel, const int64_t* super_off, const int* tile_local, void* out_values, uint8_t* out_valid,
                    unsigned long long* valid_total) {
  if (c->tune_nt && n * (int64_t)W >= ((int64_t)8 << 20))
    compact_kernel<W, HAS_VALID, INDICES, true><<<(unsigned)ntiles, kBlock, 0, c->stream>>>(values, vvalid, voff, fdata, fvalid, foff, n, null_sel,
                                                                                            super_off, tile_local, out_values, out_valid, valid_total);
  else
    compact_kernel<W, HAS_VALID, INDICES, false><<<(unsigned)ntiles, kBlock, 0, c->stream>>>(values, vvalid, voff, fdata, fvalid, foff, n, null_sel,
                                                                                             super_off, tile_local, out_values, out_valid, valid_total);
}

// to_cache: the tables go to the context's filter cache (ah_common.h) instead of the scratch arena, and the total is also posted
// to the host mailbox with sequence number `seq`
template <int W>
int run_counts(ah_ctx* c, const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
               int** tile_local_out, int64_t** super_off_out, int64_t** total_out, int64_t* ntiles_out, bool to_cache = false,
               unsigned long long seq = 0) {
  constexpr int TILE = TileBytes<W>() / W;
  constexpr int WPT = TILE / 64;
  int64_t ntiles = ah_ceil_div(n, TILE);
  int64_t nsuper = ah_ceil_div(ntiles, kSuper);
  // layout: super_off[nsuper] (int64) | tile_local[ntiles] (int) | super_total[nsuper] (int) | total (int64, cache only)
  const size_t tables = (size_t)nsuper * sizeof(int64_t) + (size_t)ntiles * sizeof(int) + (size_t)nsuper * sizeof(int);
  size_t bytes = ((tables + 7) & ~(size_t)7) + 64;
  void* scratch;
  if (to_cache) {
    ah_filter_cache& fc = c->fcache;
    if (bytes > fc.bytes) {
      AH_HIP(c, hipStreamSynchronize(c->stream));   // a fill may still be reading the old block
      if (fc.buf) AH_HIP(c, hipFree(fc.buf));
      fc.buf = nullptr; fc.bytes = 0;
      const size_t want = (bytes + 65535) & ~(size_t)65535;
      AH_HIP(c, hipMalloc(&fc.buf, want));
      fc.bytes = want;
    }
    scratch = fc.buf;
  } else {
    int rc = ah_scratch_reserve(c, bytes, &scratch);
    if (rc != AH_OK) return rc;
  }
  int64_t* super_off = (int64_t*)scratch;
  int* tile_local = (int*)(super_off + nsuper);
  int* super_total = tile_local + ntiles;
  int64_t* total = to_cache ? (int64_t*)((uint8_t*)scratch + ((tables + 7) & ~(size_t)7)) : (int64_t*)&c->dscalars[1];
  tile_count_kernel<WPT><<<(unsigned)nsuper, kBlock, 0, c->stream>>>(fdata, fvalid, foff, n, null_sel, tile_local, super_total, ntiles);
  AH_LAUNCH_CHECK(c);
  super_scan_kernel<<<1, 1024, 0, c->stream>>>(super_total, nsuper, super_off, total, to_cache ? c->mailbox : nullptr, seq);
  AH_LAUNCH_CHECK(c);
  *tile_local_out = tile_local; *super_off_out = super_off; *total_out = total; *ntiles_out = ntiles;
  return AH_OK;
}

// *_dev flavour: {rows selected, output null count} stay in device memory
__global__ void filter_status_kernel(const int64_t* __restrict__ total, const unsigned long long* __restrict__ nvalid, int has_valid,
                                     int64_t* __restrict__ status) {
  status[0] = *total;
  status[1] = has_valid ? *total - (int64_t)*nvalid : 0;
}

template <int W, bool INDICES>
int run_filter(ah_ctx* c, const void* values, const uint8_t* vvalid, int64_t voff, const uint8_t* fdata,
               const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel, int64_t n_out, void* out_values,
               uint8_t* out_valid, int64_t* out_null_count_host, int64_t* status_dev = nullptr, bool cached = false) {
  int* tile_local; int64_t* super_off; int64_t* total; int64_t ntiles;
  int rc = AH_OK;
  const ah_filter_cache& fc = c->fcache;
  if (cached && fc.fdata == fdata && fc.fvalid == fvalid && fc.foff == foff && fc.n == n && fc.null_sel == null_sel &&
      fc.tile_rows == TileBytes<W>() / W) {
    // the count call that preceded this fill left the tile prefixes of this very mask: nothing to recount
    tile_local = fc.tile_local; super_off = fc.super_off; total = fc.total; ntiles = fc.ntiles;
  } else {
    rc = run_counts<W>(c, fdata, fvalid, foff, n, null_sel, &tile_local, &super_off, &total, &ntiles);
  }
  if (rc != AH_OK) return rc;
  unsigned long long* valid_total = (unsigned long long*)&c->dscalars[2];
  if (out_valid) {
    if (n_out < 0) return ah_fail(c, AH_EINVALID, "filter: n_out (from ah_filter_count) is required with a validity output");
    if ((rc = ah_zero_bytes(c, out_valid, (size_t)((n_out + 7) / 8))) != AH_OK) return rc;   // (one launch: the stream waits for the host here)
    launch_compact<W, true, INDICES>(c, ntiles, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, super_off, tile_local, out_values, out_valid,
                                     valid_total);
  } else {
    launch_compact<W, false, INDICES>(c, ntiles, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, super_off, tile_local, out_values, nullptr,
                                      valid_total);
  }
  AH_LAUNCH_CHECK(c);
  if (status_dev) {
    // the output was sized for n rows and its validity bytes zeroed up to there: bits past the selection count add nothing
    if (out_valid && (rc = ah_popcount_async(c, out_valid, 0, n_out, valid_total)) != AH_OK) return rc;
    filter_status_kernel<<<1, 1, 0, c->stream>>>(total, valid_total, out_valid != nullptr, status_dev);
    AH_LAUNCH_CHECK(c);
    return AH_OK;
  }
  if (out_null_count_host) {
    // null count = n_out − popcount(out_valid[0, n_out)) — a 2-launch reduction over n_out/8
    // bytes instead of one same-address atomic per tile (~12 ns each, serialised at L2)
    if (out_valid) {
      rc = ah_popcount_async(c, out_valid, 0, n_out, valid_total);
      if (rc != AH_OK) return rc;
    }
    unsigned long long back[2];   // polled mailbox, not a copy + stream synchronisation: the call's tail was ≈ 20 µs of wake-up
    rc = ah_mailbox_read2(c, (const unsigned long long*)total, 1, valid_total, 1, back);
    if (rc != AH_OK) return rc;
    int64_t tot = (int64_t)back[0];
    int64_t nvalid = (int64_t)back[1];
    if (n_out >= 0 && tot != n_out)
      return ah_fail(c, AH_EINVALID, "filter: n_out=%lld does not match the selection count %lld", (long long)n_out, (long long)tot);
    *out_null_count_host = out_valid ? tot - nvalid : 0;
  }
  return AH_OK;
}

}  // namespace

AH_EXPORT int ah_filter_count(ah_ctx* c, const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n,
                              int null_sel, int64_t* n_out_host) {
  AH_ENTER(c);
  if (!n_out_host) return ah_fail(c, AH_EINVALID, "filter_count: null result pointer");
  if (n < 0 || foff < 0) return ah_fail(c, AH_EINVALID, "filter_count: negative length/offset");
  *n_out_host = 0;
  if (n == 0) return AH_OK;
  if (!fdata) return ah_fail(c, AH_EINVALID, "filter_count: null filter data");
  int* tile_local; int64_t* super_off; int64_t* total; int64_t ntiles;
  if (c->capturing) { c->capturing = 2; return ah_fail(c, AH_EINVALID, "filter_count returns a value to the host: it cannot be recorded into a graph"); }
  const unsigned long long seq = ++c->mailbox_seq;
  int rc = run_counts<8>(c, fdata, fvalid, foff, n, null_sel, &tile_local, &super_off, &total, &ntiles, /*to_cache=*/true, seq);
  if (rc != AH_OK) return rc;
  // The host cannot go on without the count (it sizes the output: vector_selection.go:459-475), so it polls the two words the scan
  // kernel posts to coherent pinned memory — a stream synchronisation's wake-up costs more than the two kernels together.  A
  // kernel that never posts (a fault) is found by the synchronisation the poll falls back to.
  bool seen = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0; !seen; spin++) {
    if (__atomic_load_n(&c->mailbox[1], __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
    __builtin_ia32_pause();
    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  if (!seen) {
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if (__atomic_load_n(&c->mailbox[1], __ATOMIC_ACQUIRE) != seq) return ah_fail(c, AH_EHIP, "filter_count: the count kernel did not report");
  }
  *n_out_host = (int64_t)__atomic_load_n(&c->mailbox[0], __ATOMIC_RELAXED);
  ah_filter_cache& fc = c->fcache;
  fc.fdata = fdata; fc.fvalid = fvalid; fc.foff = foff; fc.n = n; fc.null_sel = null_sel; fc.tile_rows = TileBytes<8>() / 8;
  fc.ntiles = ntiles; fc.super_off = super_off; fc.tile_local = tile_local; fc.total = total;
  fc.valid = c->opt_filter_cache != 0;   // never on a shared stream: a foreign kernel may rewrite the mask before the fill
  return AH_OK;
}

AH_EXPORT int ah_filter_primitive(ah_ctx* c, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                                  const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                                  int64_t n_out, void* out_values, uint8_t* out_valid, int64_t* out_null_count_host) {
  const bool cached = c && c->fcache.valid;   // read before AH_ENTER drops it: this fill is the call the cache was left for
  AH_ENTER(c);
  if (n < 0 || foff < 0 || voff < 0) return ah_fail(c, AH_EINVALID, "filter: negative length/offset");
  if (out_null_count_host) *out_null_count_host = 0;
  if (n == 0) return AH_OK;
  if (!fdata || !values) return ah_fail(c, AH_EINVALID, "filter: null input buffer");
  if (((uintptr_t)values | (uintptr_t)out_values) & (uintptr_t)(byte_width - 1))
    return ah_fail(c, AH_EINVALID, "filter: buffer not element-aligned");
  int rc;
  switch (byte_width) {
    case 1: rc = run_filter<1, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n_out, out_values, out_valid, out_null_count_host, nullptr, cached); break;
    case 2: rc = run_filter<2, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n_out, out_values, out_valid, out_null_count_host, nullptr, cached); break;
    case 4: rc = run_filter<4, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n_out, out_values, out_valid, out_null_count_host, nullptr, cached); break;
    case 8: rc = run_filter<8, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n_out, out_values, out_valid, out_null_count_host, nullptr, cached); break;
    default: return ah_fail(c, AH_EINVALID, "filter: invalid values byte width %d", byte_width);  // vector_selection.go:515
  }
  // the same mask may filter the next column: the tables stay valid unless this call's outputs lie on the mask's bytes
  if (rc == AH_OK && cached) {
    c->fcache.valid = true;
    const size_t rows_out = (size_t)(n_out >= 0 ? n_out : n);
    if (ah_fcache_overlaps(c, out_values, rows_out * (size_t)byte_width) || (out_valid && ah_fcache_overlaps(c, out_valid, (rows_out + 7) / 8)))
      c->fcache.valid = false;
  }
  return rc;
}

AH_EXPORT int ah_filter_primitive_dev(ah_ctx* c, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                                      const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                                      void* out_values, uint8_t* out_valid, int64_t* status_dev) {
  AH_ENTER(c);
  if (n < 0 || foff < 0 || voff < 0) return ah_fail(c, AH_EINVALID, "filter: negative length/offset");
  if (!status_dev) return ah_fail(c, AH_EINVALID, "filter: null status pointer");
  if (n == 0) {
    AH_HIP(c, hipMemsetAsync(status_dev, 0, 2 * sizeof(int64_t), c->stream));
    return AH_OK;
  }
  if (!fdata || !values) return ah_fail(c, AH_EINVALID, "filter: null input buffer");
  if (((uintptr_t)values | (uintptr_t)out_values) & (uintptr_t)(byte_width - 1))
    return ah_fail(c, AH_EINVALID, "filter: buffer not element-aligned");
  switch (byte_width) {   // n_out = n: the capacity the caller sized the output for
    case 1: return run_filter<1, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n, out_values, out_valid, nullptr, status_dev);
    case 2: return run_filter<2, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n, out_values, out_valid, nullptr, status_dev);
    case 4: return run_filter<4, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n, out_values, out_valid, nullptr, status_dev);
    case 8: return run_filter<8, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n, out_values, out_valid, nullptr, status_dev);
  }
  return ah_fail(c, AH_EINVALID, "filter: invalid values byte width %d", byte_width);
}

// PrimitiveFilter in ONE call for a caller that sizes its output for the worst case (n rows): no count call, no turnaround between two
// launches — counts and fill run back to back on the stream and the selection count and the output null count come back through the
// polled mailbox.  A host language pays its call overhead once instead of twice (count → allocate → fill: 0.318 ms from Python for
// 2^27 rows at s = 0.5, 0.284 from C; the fill itself is 0.263).
AH_EXPORT int ah_filter_primitive_once(ah_ctx* c, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                                       const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                                       void* out_values, uint8_t* out_valid, int64_t* n_out_host, int64_t* out_null_count_host) {
  AH_ENTER(c);
  if (n < 0 || foff < 0 || voff < 0) return ah_fail(c, AH_EINVALID, "filter: negative length/offset");
  if (!n_out_host) return ah_fail(c, AH_EINVALID, "filter: null result pointer");
  *n_out_host = 0;
  if (out_null_count_host) *out_null_count_host = 0;
  if (n == 0) return AH_OK;
  if (!fdata || !values || !out_values) return ah_fail(c, AH_EINVALID, "filter: null input buffer");
  if (((uintptr_t)values | (uintptr_t)out_values) & (uintptr_t)(byte_width - 1))
    return ah_fail(c, AH_EINVALID, "filter: buffer not element-aligned");
  if (c->capturing) { c->capturing = 2; return ah_fail(c, AH_EINVALID, "filter_primitive_once returns values to the host: it cannot be recorded into a graph"); }
  int rc;
  switch (byte_width) {   // n_out = n: the capacity the caller sized the output for; counts, zeroing and fill are enqueued back to back
    case 1: rc = run_filter<1, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n, out_values, out_valid, nullptr); break;
    case 2: rc = run_filter<2, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n, out_values, out_valid, nullptr); break;
    case 4: rc = run_filter<4, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n, out_values, out_valid, nullptr); break;
    case 8: rc = run_filter<8, false>(c, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, n, out_values, out_valid, nullptr); break;
    default: return ah_fail(c, AH_EINVALID, "filter: invalid values byte width %d", byte_width);
  }
  if (rc != AH_OK) return rc;
  // the selection count sits where run_counts left it (dscalars[1]: this call never takes the cached tables); with a validity output the
  // popcount of its n bits (the bits past the selection are zero) rides along and the popcount's last launch posts both — three small
  // launches less than the first version (popcount ×2, a status kernel, a posting kernel), which made the one-call entry SLOWER than count + fill
  const unsigned long long* total_dev = (const unsigned long long*)&c->dscalars[1];
  unsigned long long back[2] = {0, 0};
  if (out_valid) rc = ah_popcount_post(c, out_valid, 0, n, (unsigned long long*)&c->dscalars[2], total_dev, back);
  else rc = ah_mailbox_read(c, total_dev, 1, back);
  if (rc != AH_OK) return rc;
  *n_out_host = (int64_t)back[0];
  if (out_null_count_host) *out_null_count_host = out_valid ? (int64_t)(back[0] - back[1]) : 0;
  return AH_OK;
}

// internal (ah_hash_part.hip): out_values[j] = values[r], out_rows[j] = r for the set bits r of `bits` (bit 0 of byte 0 = row 0) in ascending
// order — a dictionary in first-occurrence order is the key column compacted by its "first occurrence" bitmap.  No host round trip.
int ah_compact_u64_by_bits(ah_ctx* c, const uint64_t* values, const uint8_t* bits, int64_t n, uint64_t* out_values, int64_t* out_rows) {
  if (n <= 0 || (!out_values && !out_rows)) return AH_OK;
  // the tile prefixes go to the filter cache's block, NOT to the scratch arena: the caller (the encode in progress) keeps a table there
  // that it returns to if the partition-first attempt is abandoned
  int* tile_local; int64_t* super_off; int64_t* total; int64_t ntiles;
  int rc = run_counts<8>(c, bits, nullptr, 0, n, 0, &tile_local, &super_off, &total, &ntiles, /*to_cache=*/true, ++c->mailbox_seq);
  c->fcache.valid = false;   // this call's own tables, not a count left for a fill
  if (rc != AH_OK) return rc;
  unsigned long long* unused = (unsigned long long*)&c->dscalars[2];
  if (out_values) launch_compact<8, false, false>(c, ntiles, values, nullptr, 0, bits, nullptr, 0, n, 0, super_off, tile_local, out_values, nullptr, unused);
  if (out_rows) launch_compact<8, false, true>(c, ntiles, nullptr, nullptr, 0, bits, nullptr, 0, n, 0, super_off, tile_local, out_rows, nullptr, unused);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

AH_EXPORT int ah_filter_to_indices(ah_ctx* c, const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n,
                                   int null_sel, int64_t n_out, uint32_t* out_idx, uint8_t* out_valid,
                                   int64_t* out_null_count_host) {
  AH_ENTER(c);
  if (n < 0 || foff < 0) return ah_fail(c, AH_EINVALID, "filter_to_indices: negative length/offset");
  if (out_null_count_host) *out_null_count_host = 0;
  if (n == 0) return AH_OK;
  if (n >= 0xFFFFFFFFll)  // vector_selection.go:229-235
    return ah_fail(c, AH_ENOTIMPL, "filter length exceeds UINT32_MAX, consider a different strategy for selecting elements");
  if (!fdata) return ah_fail(c, AH_EINVALID, "filter_to_indices: null filter data");
  return run_filter<4, true>(c, nullptr, nullptr, 0, fdata, fvalid, foff, n, null_sel, n_out, out_idx, out_valid, out_null_count_host);
}
