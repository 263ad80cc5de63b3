// ah_common.h — shared host/device helpers for libarrowhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <initializer_list>
#include <stdarg.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/arrowhip.h"

#define AH_EXPORT extern "C" __attribute__((visibility("default")))

// What ah_filter_count leaves for the ah_filter_primitive that follows it (the two-phase protocol of
// vector_selection.go:459/475: count, allocate, fill): the per-tile survivor prefixes of THAT mask, so the fill does not
// count the selection vector a second time.  Dropped by every entry point that may change device memory (AH_ENTER);
// allocation, synchronisation, timers and a memset / upload / copy that stays clear of the mask's bytes keep it (AH_ENTER_KEEP).
struct ah_filter_cache {
  bool valid;
  const uint8_t *fdata, *fvalid;
  int64_t foff, n, ntiles;
  int null_sel, tile_rows;
  void* buf;          // device: super_off[nsuper] (int64) | tile_local[ntiles] (int) | super_total[nsuper] (int) | total (int64)
  size_t bytes;
  int64_t* super_off;
  int* tile_local;
  int64_t* total;
};

struct ah_ctx {
  int device;
  hipStream_t stream;       // compute stream
  hipStream_t copy_stream;  // uploads / downloads
  bool owns_stream;
  hipEvent_t ev_copy;       // orders copy stream <-> compute stream
  hipEvent_t ev_compute;
  hipEvent_t t0, t1;        // ah_timer_*
  hipEvent_t* marks;        // ah_event_record ring (lazily created)
  int n_marks;
  // scratch arena (device): partial sums, tile counts, hash tables ...
  void* scratch;
  size_t scratch_bytes;
  void* temp;              // second grow-only arena: the temporaries of sort / group-by / var-length take (see ah_temp_reserve)
  size_t temp_bytes;
  // small pinned staging block for *_host results (64 x 8 bytes)
  uint64_t* pinned;
  // small device block for scalar results / flags (64 x 8 bytes) followed by a
  // 4096-entry partials area (popcount)
  uint64_t* dscalars;
  int num_cu;
  // tunables (env ARROWHIP_NT / ARROWHIP_BLOCKS_PER_CU, read at ctx creation)
  int tune_nt;             // 1: nontemporal loads/stores on streaming kernels
  int tune_blocks_per_cu;  // grid cap for grid-stride streaming kernels
  // take (ah_take_binned.hip): ARROWHIP_TAKE_BINNED 0 never / 1 auto / 2 whenever legal; _WINDOW_LOG2 bytes of `values` per
  // bin; _GATHER_WG_PER_CU occupancy cap of the gather pass.  Also settable per context: ah_ctx_set_option.
  int opt_take_binned, opt_take_window_log2, opt_take_gather_wg, opt_take_gather_load, opt_take_gather_lds;
  int opt_groupby_keys;        // hash + sum: expected keys per partition the cut aims at (ARROWHIP_GROUPBY_KEYS; the LDS table admits 3584)
  int opt_groupby_partition;   // hash + sum (ah_groupby.hip): 0 never, 1 auto, k ≥ 2 always with 2^(k−2) partitions (ARROWHIP_GROUPBY_PARTITION)
  int opt_encode_partition;    // unique / dictionary_encode (ah_hash_part.hip): 0 never, 1 auto (by the prefix's distinct count), k ≥ 3: always, 2^k partitions (ARROWHIP_ENCODE_PARTITION)
  int opt_encode_part_slots;   // measurement: LDS table size of the one-cut path (8192 default, 4096)
  int opt_encode_byte_map;     // partition-first encode: first occurrences marked by plain byte stores into a byte map that one pass packs into the bitmap (1) or by device-scope atomicOr on the bitmap's words (0)
  int opt_encode_early_look;   // 1 (default): calls of ≥ 2^24 rows count the first 2^16 rows' distinct keys BEFORE the global table is set up (ah_encode_first_look); 0: the look that falls out of the staged inserts
  int opt_encode_part_min;     // auto: smallest expected distinct count that takes the partition-first path (ARROWHIP_ENCODE_PART_MIN)
  int opt_hash_direct;         // unique / dictionary_encode (ah_hash.hip): 0 ids in a separate pass, 1 direct ids, 2 + LDS / re-packed table (default), 3 no re-packed table (ARROWHIP_HASH_DIRECT)
  int opt_sort_msd;            // sort_indices: 0 LSD passes only, 1 auto (ARROWHIP_SORT_MSD)
  int opt_scan_segment_log2;   // cumulative_sum: bytes of input per segment (ARROWHIP_SCAN_SEGMENT_LOG2; 0 = one segment)
  void* expr_cache;        // compiled expression programs (ah_expr.hip)
  int capturing;           // between ah_graph_begin and ah_graph_end: the compute stream records instead of running
  ah_filter_cache fcache;  // ah_filter.hip
  int opt_scan_onepass;    // cumulative_sum of 4- and 8-byte integers (checked or not, nulls or not): 1 (default) one pass with decoupled look-back over 128 KiB tiles, 0 reduce-then-scan, 2 one pass + separate validity copy / popcount, 3 one pass for unchecked columns without nulls only
  void* scan_recs;         // … its tile records (only that kernel writes them: stale words carry older epochs) and, in the last 64 bytes, its ticket word
  size_t scan_recs_bytes;
  unsigned scan_epoch;
  int opt_groupby_seed;    // direct group-by: 1 (default) the workgroups' LDS tables start from the quick look's keys and are added up slot by slot, 0 empty tables merged with atomics
  int opt_groupby_reserve; // partitioned group-by: 1 (default) the scatter reserves its runs in per-(partition, XCD) regions sized from the sample — no histogram pass —, 0 histogram → offsets → scatter
  int opt_groupby_lean;    // direct group-by: 0 always keep a pending group per lane, 1 (default) leave it out when the quick look says neighbouring rows rarely share a key, 2 always leave it out, 3 also in the partitioned path's aggregate pass (an experiment switch: −3 % on evenly spread keys, profiles/r06_gb_lean.json; a hot key would serialise its partition's LDS atomics)
  int opt_filter_cache;    // 1: ah_filter_count leaves its tile prefixes for the fill (default on a stream of the context's own), 0: the fill recounts (default on a shared stream)
  int take_clustered_hint; // ah_take_binned_try → ah_take.hip: this call's indices looked clustered (1), not (0); option take_vec: 0 never, 1 by the sample, 2 always
  int opt_take_vec;
  // The neighbour sample of the last Take (ah_take_binned_try), kept for the next call with the SAME index vector — a record batch's
  // columns are gathered one after the other with one index vector (compute/selection.go:601-677), and the sample + its wait were
  // ≈ 20 µs of every one of those calls.  Kept like the filter cache: valid only for the Take that directly follows (every other
  // compute entry point of this context drops it, an upload / copy / memset into the vector's bytes drops it, a Take whose outputs
  // overlap the vector does not leave one), off by default on a shared stream, sampled again on every 32nd use.  Every Take path
  // returns the same bytes (tests force each of them): an entry gone stale behind the library's back costs speed, never results.
  // Option take_hint_cache.
  bool take_hint_valid, take_hint_live;   // left by the last call / readable by this one (read before AH_ENTER drops it)
  const void* take_hint_idx; int64_t take_hint_nidx; int take_hint_kind, take_hint_near, take_hint_uses, opt_take_hint_cache;
  unsigned long long take_hint_word;
  int opt_arith_xcd_map;        // element-wise binary kernels: every XCD streams one contiguous eighth of the columns (1) or the blocks' natural interleave (0)
  int opt_groupby_scale_guess;  // no-cut Float64 group-by: fixed-point scale from a sample, verified by the aggregate pass (1) or from a pass over all values (0)
  int opt_encode_unperm2_group; // two-cut encode: the level-2 un-permute over 4 consecutive virtual tiles per workgroup (4, the default) or one per workgroup (0)
  int opt_encode_resolve_wgs;   // partition-first encode: workgroups of the resolve pass (partitions are split to reach it)
  int opt_encode_dict_compact;  // partition-first encode: the dictionary = the key column compacted by the first-occurrence bitmap (0 never, 1 from 1024 partitions on, 2 always)
  int opt_encode_table_batch;   // partition-first encode: the table pass probes its four records' first groups together (1) or one by one (0)
  int opt_encode_unperm_group;  // partition-first encode: tiles per workgroup in the final un-permute (1 or 4)
  int opt_take_vec_nt;     // nontemporal hints of the clustered take (7 all, 5 index + output, 4 output, 0 none)
  // 2 × 8 words of coherent (fine-grained) pinned host memory kernels can store to: {value, sequence number} for ah_filter_count, {≤ 7 words, sequence number} for ah_mailbox_read.  A count the host
  // must see before it can go on (ah_filter_count) is polled here instead of paying a stream synchronisation's wake-up.
  unsigned long long* mailbox;
  unsigned long long mailbox_seq;
  char err[512];
};

// grid size for a grid-stride streaming kernel over `work_items` block-iterations.
// default_bpc: the kernel's own default cap in workgroups per CU (0 = no cap: one
// iteration per workgroup, measured fastest for pure element-wise streams — 6.4-6.5 TB/s
// vs 6.0 with 8/CU, scripts/micro/add_variants.hip); ARROWHIP_BLOCKS_PER_CU overrides.
static inline unsigned ah_stream_grid(const ah_ctx* c, int64_t work_items, int default_bpc = 8) {
  int bpc = c->tune_blocks_per_cu > 0 ? c->tune_blocks_per_cu : default_bpc;
  int64_t cap = bpc > 0 ? (int64_t)c->num_cu * bpc : ((int64_t)1 << 30);
  int64_t g = work_items < cap ? work_items : cap;
  return (unsigned)(g < 1 ? 1 : g);
}

// ---- host-side error plumbing ------------------------------------------------
static inline int ah_fail(ah_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
static inline int ah_fail(ah_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define AH_HIP(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e__ = (call);                                                           \
    if (e__ != hipSuccess)                                                             \
      return ah_fail((ctx), AH_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                     __FILE__, __LINE__);                                              \
  } while (0)

#define AH_ENTER_KEEP(ctx)                                             \
  do {                                                                 \
    if (!(ctx)) return AH_EINVALID;                                    \
    (ctx)->err[0] = 0;                                                 \
    AH_HIP((ctx), hipSetDevice((ctx)->device));                        \
  } while (0)

#define AH_ENTER(ctx)                                                  \
  do {                                                                 \
    AH_ENTER_KEEP(ctx);                                                \
    (ctx)->fcache.valid = false;                                       \
    (ctx)->take_hint_valid = false;                                    \
  } while (0)

// does a write to [p, p + nbytes) touch the index vector the Take's path sample was taken from?
static inline bool ah_take_hint_overlaps(const ah_ctx* c, const void* p, size_t nbytes) {
  if (!c->take_hint_valid || !c->take_hint_idx) return false;
  const uintptr_t a = (uintptr_t)p, b = a + nbytes, lo = (uintptr_t)c->take_hint_idx, hi = lo + (size_t)c->take_hint_nidx * (size_t)(c->take_hint_kind / 2);
  return a < hi && lo < b;
}
// does a write to [p, p + nbytes) touch the mask the filter cache was computed from?
static inline bool ah_fcache_overlaps(const ah_ctx* c, const void* p, size_t nbytes) {
  if (!c->fcache.valid) return false;
  const uintptr_t a = (uintptr_t)p, b = a + nbytes;
  const uintptr_t lo = (uintptr_t)(c->fcache.foff >> 3), hi = (uintptr_t)((c->fcache.foff + c->fcache.n + 7) >> 3);
  for (const uint8_t* bm : {c->fcache.fdata, c->fcache.fvalid})
    if (bm && a < (uintptr_t)bm + hi && (uintptr_t)bm + lo < b) return true;
  return false;
}

#define AH_LAUNCH_CHECK(ctx) AH_HIP((ctx), hipGetLastError())

void ah_expr_cache_free(ah_ctx* ctx);  // ah_expr.hip
// internal (ah_sort.hip): stable radix partition of (value bits, group id) pairs for the group-by of
// ah_hash.hip — by (id >> shift) & 255 (passes = 1) or by (id >> shift) & 65535 (passes = 2, LSD; alt_*
// is the intermediate buffer).  A row whose value is null (vvalid bit clear) travels with bit 31 of
// its id set.  hist / offs: 256 · ceil(n / 2048) unsigned each.
int ah_partition_by_group(ah_ctx* ctx, const int32_t* ids, const unsigned long long* vals, const uint8_t* vvalid, int64_t voff, int64_t n,
                          int shift, int passes, unsigned* hist, unsigned* offs, unsigned long long* alt_vals, unsigned* alt_ids,
                          unsigned long long* out_vals, unsigned* out_ids);
// internal (ah_sort_msd.hip): the `rest` range of sort_indices by two MSD partition passes + one wave per bucket; *used = 0: not
// applicable or a bucket came out too large — the pairs are clobbered and the caller regenerates them for the LSD passes
int ah_sort_rest_msd(ah_ctx* ctx, unsigned long long* keys, unsigned* rows, unsigned long long* alt_keys, unsigned* alt_rows, int64_t n,
                     unsigned long long varying, unsigned long long kmin, unsigned long long kmax, int float_bytes, int descending, void* tmp,
                     unsigned long long* out64, int* used);
size_t ah_sort_msd_temp_bytes(int64_t n);   // size of `tmp`; 0: the path does not apply to n pairs
// internal (ah_sort.hip): (key, row) pairs sorted by the full 64-bit key with the stable LSD passes; *sorted_keys = where the keys ended up
int ah_sort_pairs_lsd(ah_ctx* ctx, unsigned long long* keys, unsigned* rows, unsigned long long* alt_keys, unsigned* alt_rows, int64_t n,
                      unsigned* hist, unsigned* offs, unsigned long long** sorted_keys);
// internal (ah_take_binned.hip): Take for random indices into a column the caches cannot hold; *used says whether it ran
int ah_take_binned_try(ah_ctx* ctx, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff, int64_t nvalues, int iw,
                       int is_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, void* out_values,
                       uint8_t* out_valid, unsigned long long* first_bad, int* used);
// internal (ah_groupby.hip): partition-first group-by; *used says whether it produced the result (else: the id-based path)
int ah_groupby_partitioned_try(ah_ctx* ctx, int is_f64, const uint64_t* keys, const uint8_t* kvalid, int64_t koff, const void* vals,
                               const uint8_t* vvalid, int64_t voff, int64_t n, uint64_t* out_keys, void* out_sums, int64_t* out_counts,
                               int64_t* out_first_rows, int64_t* out_ngroups, int32_t* out_null_group, int* used);
// internal (ah_sum.hip): the workgroup partials of one chunk written to a caller-owned array (16 bytes reserved per partial; *n_written
// counts 8-byte partials for the integer sums, 16-byte ones for Float64), and the one final reduction over all of them
int ah_fused_f64_parts_dev(ah_ctx* ctx, int cmpop, const double* x, const uint8_t* valid, int64_t off, int64_t n, double threshold,
                           double* out_sum_dev, int64_t* out_count_dev, double* out_parts_dev);
size_t ah_sum_partial_bytes(int is_f64);
int ah_sum_chunk_partials(ah_ctx* ctx, int is_f64, const void* buf, size_t len, void* partials, int max_partials, int* n_written);
int ah_sum_finish_partials(ah_ctx* ctx, int is_f64, const void* partials, int n, void* res_dev);
int ah_sum_short_f64(ah_ctx* ctx, const void* buf, size_t len, void* res_dev);   // ≤ 31 rows: the reference's sequential order
// internal (ah_hash_part.hip): unique / dictionary_encode of 8-byte keys by partitions of the key hash, 2^lp of them (8 … 10);
// temporaries in the temp arena; *used says whether out_* hold the result
int ah_encode_partitioned_try(ah_ctx* ctx, const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls, int lp, int slots,
                              int32_t* out_ids, uint64_t* out_dict, int64_t* out_first_rows, int64_t* out_ndict, int32_t* out_null_id, int* used);
// … and by two cuts, 64 parents × 2^(lp − 6) partitions (lp = 11 … 13) with LDS tables of `slots` = 4096 or 8192 entries
// distinct valid keys among the first `rows` (≤ 2^16) rows, counted exactly without a table in HBM (ah_hash_part.hip): one launch, one polled wait
int ah_encode_first_look(ah_ctx* ctx, const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t rows, uint64_t* distinct);
int ah_encode_partitioned2_try(ah_ctx* ctx, const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls, int lp, int slots,
                               int32_t* out_ids, uint64_t* out_dict, int64_t* out_first_rows, int64_t* out_ndict, int32_t* out_null_id, int* used);
// internal (ah_ctx.hip): 1..7 device words (8 bytes each, written by work already on the compute stream) → host, through the polled
// mailbox — a cheaper "I need this number before I go on" than a copy + stream synchronisation.  Everything enqueued before it has
// completed when it returns, like a synchronisation of the stream up to that point.
int ah_compact_u64_by_bits(ah_ctx* ctx, const uint64_t* values, const uint8_t* bits, int64_t n, uint64_t* out_values, int64_t* out_rows);   // ah_filter.hip
int ah_check_stall(ah_ctx* ctx);   // AH_EHIP if a kernel reported a stalled look-back since the last check (ah_ctx.hip)
int ah_mailbox_read(ah_ctx* ctx, const unsigned long long* dev_words, int nwords, unsigned long long* out_host);
int ah_mailbox_begin(ah_ctx* ctx, unsigned long long** mb_out, unsigned long long* seq_out);
int ah_mailbox_wait(ah_ctx* ctx, unsigned long long seq, int nwords, unsigned long long* out_host);
#if defined(__HIPCC__)
// ONE thread of a kernel, after everything it reports is final (and visible to it): ≤ 7 words, then the sequence number
__device__ __forceinline__ void ah_mailbox_post(unsigned long long* mb, unsigned long long seq, const unsigned long long* words, int nwords) {
  for (int i = 0; i < nwords; i++) __hip_atomic_store(&mb[i], words[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&mb[7], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif
int ah_mailbox_read2(ah_ctx* ctx, const unsigned long long* dev_words, int nwords, const unsigned long long* dev_words2, int nwords2,
                     unsigned long long* out_host);
// Grow-only scratch arena. Contents are undefined after the call.
int ah_scratch_reserve(ah_ctx* ctx, size_t nbytes, void** out);
// A second grow-only arena for entry points that call a scratch user (the scan) while their own temporaries are
// live: sort, group-by, var-length take.  One reservation per top-level call, carved by the caller; the block is
// reused by the next call in stream order (no hipMalloc / hipFree — each maps and unmaps the whole range — per call).
int ah_temp_reserve(ah_ctx* ctx, size_t nbytes, void** out);
// internal (ah_bitmap.hip): popcount of bits [off, off+nbits) into *total_dev (8 bytes,
// device), enqueued on the compute stream; uses dscalars[16..] as partials — no scratch.
int ah_popcount_async(ah_ctx* ctx, const uint8_t* bits, int64_t off, int64_t nbits, unsigned long long* total_dev);
int ah_popcount_post(ah_ctx* ctx, const uint8_t* bits, int64_t off, int64_t nbits, unsigned long long* total_dev, const unsigned long long* extra_dev,
                     unsigned long long* out_host /* [2]: *extra_dev, the count */);
// zero a device byte range with one launch on the context's stream (any alignment)
int ah_zero_bytes(ah_ctx* ctx, void* dptr, size_t nbytes);

static inline int ah_type_width(int type) {
  switch (type) {
    case AH_UINT8: case AH_INT8: return 1;
    case AH_UINT16: case AH_INT16: return 2;
    case AH_UINT32: case AH_INT32: case AH_FLOAT32: return 4;
    case AH_UINT64: case AH_INT64: case AH_FLOAT64: return 8;
  }
  return 0;
}

static inline int64_t ah_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers ------------------------------------------------------------
#define AH_WAVE 64

// 16-byte vector with only element alignment: lets the backend emit
// global_load_dwordx4 on buffers that are merely element-aligned (Arrow slices:
// &values[offset]); gfx950 runs in unaligned-access mode for global memory.
template <typename T>
struct alignas(sizeof(T)) ah_vec16 {
  T v[16 / sizeof(T)];
};

// 16 bytes at ELEMENT alignment with a nontemporal hint.  Through a native vector type and element-wise moves: a __builtin_bit_cast
// of the loaded vector to the carrier struct compiles to a plain load (seen in the ISA of the first clustered-take kernel), and so
// does a run-time choice between a hinted and a plain access of one address — make the hint a template parameter.
template <typename T>
using ah_raw16 __attribute__((aligned(sizeof(T)))) = T __attribute__((ext_vector_type(16 / sizeof(T))));
template <typename T>
__device__ __forceinline__ ah_vec16<T> ah_ld16_nt(const T* p) {
  const ah_raw16<T> raw = __builtin_nontemporal_load(reinterpret_cast<const ah_raw16<T>*>(p));
  ah_vec16<T> r;
#pragma unroll
  for (int j = 0; j < (int)(16 / sizeof(T)); j++) r.v[j] = raw[j];
  return r;
}
template <typename T>
__device__ __forceinline__ void ah_st16_nt(T* p, const ah_vec16<T>& x) {
  ah_raw16<T> raw;
#pragma unroll
  for (int j = 0; j < (int)(16 / sizeof(T)); j++) raw[j] = x.v[j];
  __builtin_nontemporal_store(raw, reinterpret_cast<ah_raw16<T>*>(p));
}

// the same without the hint: what an element-aligned Arrow slice is read and written with.  `*(const ah_vec16<T>*)p` compiles to
// 2 × 8 bytes (T = 8 bytes), 12 + 4 (4 bytes), 12 + 2 + 1 + 1 (1 byte) — the struct's alignment is its element's — where one
// 16-byte access through the native vector type is legal on gfx950 (unaligned-access mode) and is what the memory pipeline wants.
template <typename T>
__device__ __forceinline__ ah_vec16<T> ah_ld16(const T* p) {
  const ah_raw16<T> raw = *reinterpret_cast<const ah_raw16<T>*>(p);
  ah_vec16<T> r;
#pragma unroll
  for (int j = 0; j < (int)(16 / sizeof(T)); j++) r.v[j] = raw[j];
  return r;
}
template <typename T>
__device__ __forceinline__ void ah_st16(T* p, const ah_vec16<T>& x) {
  ah_raw16<T> raw;
#pragma unroll
  for (int j = 0; j < (int)(16 / sizeof(T)); j++) raw[j] = x.v[j];
  *reinterpret_cast<ah_raw16<T>*>(p) = raw;
}

template <typename T>
__device__ __forceinline__ T ah_wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

__device__ __forceinline__ int ah_lane() { return (int)(threadIdx.x & 63); }

// Read bit i (LSB-first) of a bitmap; NULL bitmap = all ones.
__device__ __forceinline__ int ah_bit(const uint8_t* __restrict__ bm, int64_t i) {
  return bm == nullptr ? 1 : (bm[i >> 3] >> (i & 7)) & 1;
}

// 64 consecutive bits of a bitmap starting at absolute bit position `pos`
// (any alignment), reading only whole 8-byte-aligned words that contain at least
// one bit in [pos, pos + nvalid) — never touches a word outside the caller's
// range, so it cannot fault on the last page of an allocation.  Bits past nvalid
// are returned as 0.  NULL bitmap → all ones (masked to nvalid).
__device__ __forceinline__ uint64_t ah_load_bits64(const uint8_t* __restrict__ bm, int64_t pos, int nvalid);
// The validity word of the 64 consecutive rows a wave is looking at (row `pos0` = the wave's lane 0, the same in every lane):
// the position is moved to scalar registers, so the two 8-byte loads are scalar loads — one per wave instead of a byte load
// (a vector memory instruction with its address arithmetic) per lane and row.
__device__ __forceinline__ uint64_t ah_wave_bits64(const uint8_t* __restrict__ bm, int64_t pos0, int64_t nrows_left) {
  const int64_t p = ((int64_t)__builtin_amdgcn_readfirstlane((int)(pos0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)pos0);
  const int64_t left = ((int64_t)__builtin_amdgcn_readfirstlane((int)(nrows_left >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)nrows_left);
  return ah_load_bits64(bm, p, left >= 64 ? 64 : (int)(left < 0 ? 0 : left));
}
__device__ __forceinline__ uint64_t ah_load_bits64(const uint8_t* __restrict__ bm, int64_t pos, int nvalid) {
  uint64_t mask = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1);
  if (nvalid <= 0) return 0;
  if (bm == nullptr) return mask;
  uintptr_t addr = (uintptr_t)bm + (uintptr_t)(pos >> 3);
  uintptr_t base = addr & ~(uintptr_t)7;
  int shift = (int)((addr - base) * 8 + (pos & 7));  // 0..63
  const uint64_t* w = (const uint64_t*)base;
  uint64_t lo = w[0] >> shift;
  if (shift != 0 && shift + nvalid > 64) lo |= w[1] << (64 - shift);
  return lo & mask;
}
