// ah_groupby.hip — partition-first group-by (hash + sum) for 8-byte keys: config C5 of the baseline.
//
// There is no group-by in the reference: the definition is "dictionary-encode the keys with
// hashing.Table[uint64] (internal/hashing/xxh3_memo_table_types.go:283-294; first-seen order, the null key takes the
// id at which it is first seen, :231-238), then accumulate value[i] into group id[i] in row order".  The id-based path
// of ah_hash.hip follows that literally — ids for every row (one probe of a table that lives in L2 / the Infinity
// Cache / HBM per row), then a partition of (value, id) by id window and an LDS aggregation.  The ids in ROW order are
// the expensive part, and a group-by does not need them: it needs, per distinct key, {sum, count, first row}.
//
// So the rows are first cut by a hash of the key into P partitions small enough that ALL keys of a partition fit a
// table in LDS, and everything after the cut happens at LDS speed:
//
//   0 sample      2^21 rows spread over the column → linear-counting bitmap, read at half and at full sample → distinct
//                 estimate (even-draw extrapolation, × 4 when the two points say "heavy tail") → P = 8 … 1024
//   1 hist        per (tile, partition) counts
//   - offsets     count table → position of every tile's first record of every partition        (ah_bins.h)
//   2 scatter     {key, value bits, row | null flags} staged in partition order in LDS, written as runs; the largest
//                 finite |value| rides along (the fixed-point scale of ah_hashing.h)
//   3 aggregate   a workgroup takes ≤ 2^18 consecutive records of ONE partition: open-addressing table in LDS
//                 {key, sum lo, sum hi, count | inf/nan flags, first row}, LDS atomics per row; the table leaves
//                 as one coalesced copy (a partition handled by one workgroup) or is merged into the partition's
//                 global table with atomics (partitions cut into several chunks: skewed keys)
//   4 rank, emit  first rows → n-bit bitmap → prefix popcount = the sequential memo index of each group
//                 (same argument as ah_hash.hip), groups written to out_*[id]
//
// Up to ≈ 4300 expected groups the id-based path keeps all groups in one LDS table and is as fast; beyond ≈ 2.1 M groups
// 1024 partitions are not enough — both are left to it.
// Nothing here depends on timing: integer sums and the 128-bit fixed-point float sums are associative, first row is
// a minimum, ids are a function of the first rows — two runs give identical bytes, and the same bytes as the id-based
// path (tests/test_gpu_parity.py::test_hash_sum_paths_agree).
// Algorithmic bytes: 16 B/row (key + value).  Moved: 8 (hist) + 16 + 20 (scatter) + 20 (aggregate) = 64 B/row, all
// streaming or ≥ 32-byte runs, against ≈ 90–200 B/row of random lines on the id-based path at 2^16–2^24 groups.
#include <type_traits>
#include "ah_common.h"
#include "ah_hashing.h"
#include "ah_bins.h"
#include "ah_partition.h"
#include "ah_msd.h"

namespace {

constexpr int kSlots = 4096;                         // LDS table: open addressing, linear probing
constexpr int kSoftLimit = 3584;                     // keys admitted to the LDS table; later keys go to the global table
constexpr int kLSlots = kSlots + 2;                  // + the all-ones key (kSlots) and the null key (kSlots + 1)
constexpr int kGSlots = 8192;                        // global table of ONE partition
constexpr int kGStride = kGSlots + 8;                // + the same two special slots at kGSlots, kGSlots + 1
constexpr int kFlatStride = kSlots + 8;              // flat mode (two-level cut): the LDS table as it is + the two special slots at kSlots
constexpr int kChunkLog2 = 18;                       // records per aggregate workgroup
constexpr unsigned kCntMask = 0x1fffffffu;           // count word: bits 29..31 = NaN / +inf / −inf seen
// largest finite |value| — and the smallest exponent, inverted (fx_inv_exp, ah_hashing.h) — of the call from the per-tile pairs the scatter
// pass leaves (plain stores per tile: 65 536 atomicMax on one address cost 0.8 ms — 12 ns each, serialised)
__global__ __launch_bounds__(1024) void gb_max_kernel(const unsigned long long* __restrict__ tile_rng, int64_t ntiles, unsigned long long* __restrict__ range) {
  __shared__ unsigned long long s_max[16], s_imin[16];
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
  }
}

// ---- 3: aggregate -----------------------------------------------------------------------------------------------------
// workgroups a partition of `rows` records is cut into: rows / 2^18 ROUNDED (and ≥ 1 when there are rows), so that evenly
// spread keys with n / P = 2^18 give one chunk per partition — not one full chunk plus a sliver, which doubled the
// number of workgroup rounds and turned every table copy into an atomic merge
__host__ __device__ __forceinline__ unsigned gb_chunks(unsigned rows) {
  const unsigned c = (rows + (1u << (kChunkLog2 - 1))) >> kChunkLog2;
  return rows ? (c ? c : 1u) : 0u;
}
struct GbTable {   // global tables of all partitions, structure of arrays; partition p owns [p·kGStride, (p+1)·kGStride)
  unsigned long long* key;
  unsigned long long* lo;
  unsigned long long* hi;    // floats only
  unsigned* cnt;
  unsigned* first;
};

// The table hashes are 32-bit (two quarter-rate multiplies instead of the eight of gb_mix): inside a partition the keys
// agree in the top bits of gb_mix, so a different function is wanted here anyway.
__device__ __forceinline__ unsigned gb_hash32(unsigned long long k) {
  unsigned h = ((unsigned)k ^ ((unsigned)(k >> 32) * 0x9E3779B1u)) * 0x85EBCA6Bu;
  return h ^ (h >> 15);
}
__device__ __forceinline__ unsigned gb_lslot(unsigned h) { return (h * 0xC2B2AE35u) >> (32 - 12); }
__device__ __forceinline__ unsigned gb_gslot(unsigned h) { return (h * 0x27D4EB2Fu) >> (32 - 13); }
static_assert(kSlots == 1 << 12 && kGSlots == 1 << 13, "slot hashes take the top 12 / 13 bits");

// slot of `key` in partition table `base` (find or claim); −1 = table full → overflow flag, row dropped (the host falls back)
__device__ __noinline__ long long gb_global_slot(unsigned long long* __restrict__ gkey, long long base, unsigned long long key, unsigned* __restrict__ overflow) {
  if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return -1;   // void already: the result is discarded
  unsigned j = gb_gslot(gb_hash32(key));
  for (int probes = 0; probes < kGSlots; probes++) {
    unsigned long long cur = gkey[base + j];
    if (cur == kEmpty) cur = atomicCAS(&gkey[base + j], kEmpty, key);
    if (cur == kEmpty || cur == key) return base + j;
    j = (j + 1) & (kGSlots - 1);
  }
  atomicExch(overflow, 1u);
  return -1;
}

// Find or claim the slot of `key` in an LDS table of kSlots keys (not the all-ones key, not a null).  Linear probing over aligned
// groups of 4 slots, a group per step (32 bytes of LDS, one wait): a wave walks as far as its unluckiest lane, and at load ¼ one
// slot per step meant 3–4 dependent round trips per row for the wave.  The first slot of the probe order that holds the key or is
// empty decides (no deletions: a key never sits behind an empty slot of its own probe order).  The eight comparisons stay
// booleans (lane masks in scalar registers, combined by the scalar unit); per-lane integer masks cost three vector instructions
// per comparison.  Two multiplies: inside a partition the keys agree in the top bits of gb_mix, so a different function is wanted
// anyway.  → slot, or −1 when the table has admitted kSoftLimit keys (tickets are never returned: "full" sticks).
// lkey_base = the LDS byte address of l_key.  Used by the aggregate pass and by the quick look that SEEDS its tables: both must
// place a key the same way.
__device__ __forceinline__ int gb_lds_slot(unsigned lkey_base, unsigned long long* l_key, unsigned* s_used, unsigned long long key) {
  unsigned g = ((((unsigned)key * 0x9E3779B1u) ^ ((unsigned)(key >> 32) * 0x85EBCA6Bu)) >> 20) & (unsigned)(kSlots - 4);
  for (;;) {
    // Two ds_read_b128, spelled out: the compiler splits either 16-byte half into a ds_read2_b64 (it does not see the
    // alignment through the loop-carried slot number), which serves 16 lanes per LDS cycle over 32 banks where
    // ds_read_b128 serves 16 lanes over 64 — and bank conflicts are ¾ of this kernel's LDS time (SQ_LDS_BANK_CONFLICT).
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u64x2 a, c;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(c) : "v"(lkey_base + g * 8u) : "memory");
    const bool h0 = a.x == key, h1 = a.y == key, h2 = c.x == key, h3 = c.y == key;
    const bool e0 = a.x == kEmpty, e1 = a.y == kEmpty, e2 = c.x == kEmpty, e3 = c.y == kEmpty;
    const bool y0 = h0 || e0, y1 = h1 || e1, y2 = h2 || e2, y3 = h3 || e3;
    if (!(y0 || y1 || y2 || y3)) { g = (g + 4) & (kSlots - 1); continue; }
    const int j = (int)g + (y0 ? 0 : y1 ? 1 : y2 ? 2 : 3);
    if (h0 || (!e0 && (h1 || (!e1 && (h2 || (!e2 && h3)))))) return j;   // the key sits in front of the first empty slot (a key is in the table once)
    // first empty slot of the probe order: claim it
    if (atomicAdd(s_used, 1u) >= (unsigned)kSoftLimit) return -1;
    const unsigned long long cur = atomicCAS(&l_key[j], kEmpty, key);
    if (cur == kEmpty || cur == key) return j;
    // another key took it meanwhile: look at the group again
  }
}

// The direct path's per-workgroup results for SEEDED slots (see gb_aggregate_kernel, flat == 2): [workgroup][kLSlots] planes, plain stores
struct GbStaging {
  unsigned long long* lo;
  unsigned long long* hi;
  unsigned* cnt;
  unsigned* first;
};

// ---- group records, binned by first row (the two-level cut's finish) -------------------------------------------------------------------
// With millions of groups the finish used to be the worst part of the call: one bit per group set by a device-scope atomic at a random
// word of the n-bit first-occurrence bitmap (617 µs and 514 MiB of write traffic for an 8 MiB bitmap at 2^24 groups), then four 8-byte
// stores per group to out_*[id] at a random id (1593 µs, 1.5 GiB written for 0.5 GiB of output, every store dirtying its own line).
// A group's id is the RANK of its first row among all first rows, i.e. monotone in the first row: groups binned by first-row range
// are binned by id range.  So the groups leave the LDS tables as 32-byte records (one whole sector per store), binned by first row
// in two reserving scatters (coarse: 2^cshift rows, straight out of the aggregate pass; fine: 4096 rows) — a bin of R rows holds at
// most R first rows, so every bin has a fixed place [bin·R, bin·R + count) and no histogram pass is needed — and one workgroup per
// fine bin ranks its ≤ 4096 records with a 4096-bit bitmap in LDS and writes a contiguous id range of the four output columns.
// The order of the records inside a bin depends on timing; nothing else does (ranks come from the bitmap).
struct __attribute__((aligned(16))) GbRec {
  unsigned long long key, lo, hi;   // hi: Float64 sums only
  unsigned cnt;                     // count | NaN / ±inf flags (kCntMask)
  unsigned first;                   // first row | kKeyNull for the null group
};
static_assert(sizeof(GbRec) == 32, "one record = one 32-byte sector");
constexpr int kRecFineLog2 = 12;    // rows per fine bin: its records' ranks fit a 64-word bitmap
constexpr int kRecMaxBins = 512;    // coarse bins, and fine bins per coarse bin: 2^(9 + 9 + 12) rows
struct GbRecOut {
  GbRec* recs;          // coarse bin b owns [b << cshift, (b + 1) << cshift)
  unsigned* ccursor;    // records in each coarse bin so far
  int cshift, ncoarse;
};
__device__ __forceinline__ void gb_rec_store(GbRec* __restrict__ dst, unsigned long long key, unsigned long long lo, unsigned long long hi, unsigned cnt, unsigned first) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  u64x2 a = {key, lo}, b = {hi, (unsigned long long)cnt | ((unsigned long long)first << 32)};
  u64x2* d = reinterpret_cast<u64x2*>(dst);
  d[0] = a;
  d[1] = b;
}
__device__ __forceinline__ GbRec gb_rec_load(const GbRec* __restrict__ src) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2* p = reinterpret_cast<const u64x2*>(src);
  const u64x2 a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1);
  GbRec r;
  r.key = a.x; r.lo = a.y; r.hi = b.x; r.cnt = (unsigned)b.y; r.first = (unsigned)(b.y >> 32);
  return r;
}

template <bool FX>
__device__ __forceinline__ void gb_global_add(const GbTable& gt, int64_t s, unsigned long long lo, unsigned long long hi, unsigned cntflags, unsigned first) {
  if (FX) { if (lo | hi) fx_add(gt.lo, gt.hi, (size_t)s, lo, hi); }
  else if (lo) atomicAdd(&gt.lo[s], lo);
  if (cntflags & kCntMask) atomicAdd(&gt.cnt[s], cntflags & kCntMask);
  if (cntflags & ~kCntMask) atomicOr(&gt.cnt[s], cntflags & ~kCntMask);
  if (gt.first[s] > first) atomicMin(&gt.first[s], first);
}

// FX: Float64 values summed in 128-bit fixed point; else 64-bit wrapping integer sums.
//
// What bounds this kernel: the vector unit's instruction stream — not LDS, not HBM, and not latency.  PMC at 2^26 rows: 140 VALU +
// 61 SALU + 7 LDS instructions per 64 rows, of the VALU ones ≈ 40 on 64-bit integers (half rate) and 3 quarter-rate multiplies;
// SQ_ACTIVE_INST_ANY = 94 % of the SIMD issue time although every wave is parked half of ITS time (4 waves per SIMD: the table
// takes 136 KiB).  An LDS micro-benchmark retires the row's four LDS operations at 2 rows / clock / CU, eight times the rate
// seen here.  What was tried, read off the ISA and measured (2^16 groups, 2^26 rows):
//   · branch-free addends (fx_split without the shift-direction branch), lane masks kept as booleans for the 4-slot probe,
//     two multiplies for the slot hash instead of three: 160 → 126 vector instructions per row, 441 → 406 µs;
//   · the software pipeline of the record loads never overlapped anything before: entering the loop with the first step's
//     loads pending, the compiler's wait-count pass put "≤ 3 loads outstanding" in front of the row processing, i.e. a wait
//     for the prefetch just issued.  Fixed with an explicit wait before the loop (below) — the time did not move, so the
//     loads were never what the waves waited for;
//   · two pending groups per lane flushed together (both probes read in one go, both returning adds in flight together,
//     wave-uniform loops): half the dependent LDS round trips per row, same instruction count — 6 % SLOWER.  Latency is not it.
// What is left is fewer instructions per row, and the ones that remain are the algorithm: eight 64-bit compares per probe
// step, a 128-bit fixed-point split, two 128-bit adds, five LDS operations.
// SEG (flat == 0 behind the RESERVING scatter, ah_partition.h 1b): the records of a partition lie in kGbRegions regions of the record
// arrays; r0 / r1 / binstart are DENSE positions and seg_vstart / seg_delta turn one into a physical position.  The row loop runs over
// the dense positions of the whole partition as before — a step that lies inside one region (all but seven per partition) adds one
// uniform offset to its addresses, a step across a region boundary looks its regions up lane by lane.
template <bool FX, bool DIRECT = false, bool LEAN = false, bool SEG = false, bool RECS = false>
__global__ __launch_bounds__(kThreads) void gb_aggregate_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ vals,
                                                                 const unsigned* __restrict__ prows, const unsigned* __restrict__ binstart, int nb, GbTable gt,
                                                                 const unsigned long long* __restrict__ absmax, unsigned* __restrict__ overflow, int flat,
                                                                 const uint8_t* __restrict__ kvalid, int64_t koff, const uint8_t* __restrict__ vvalid,
                                                                 int64_t voff, int64_t nrows, int64_t seg_rows, unsigned* __restrict__ tile_range = nullptr,
                                                                 const unsigned long long* __restrict__ seed_keys = nullptr, unsigned seed_used = 0,
                                                                 GbStaging st = GbStaging{nullptr, nullptr, nullptr, nullptr},
                                                                 const unsigned* __restrict__ seg_vstart = nullptr, const unsigned* __restrict__ seg_delta = nullptr,
                                                                 GbRecOut ro = GbRecOut{nullptr, nullptr, 0, 0}) {
  // RECS (flat == 1 only): the table does not leave as a copy — its occupied slots leave as records binned by first row (GbRec above)
  // seed_keys (flat == 2 only): the key plane of an LDS table holding the keys a quick look found — EVERY workgroup starts from
  // it, so a seeded key has the same slot in all of them and their results for it are added up slot by slot afterwards
  // (gd_reduce_kernel) instead of 256 workgroups × groups × 5 atomics on one global table (0.1 ms per 1024 groups, serialised in
  // L2).  Keys the look did not see are placed behind the seeded ones as they come and leave through the atomic merge as before.
  // flat = 2 ("direct", ≤ 2048 expected groups): no cut at all — keys / vals ARE the columns, a workgroup takes 2^18 consecutive rows
  // and the chunks are merged into ONE global table with atomics.
  // flat = 1: one workgroup per partition (blockIdx = partition; thousands of partitions from the two-level cut), tables of
  // kSlots + 8 entries side by side, no merging: a partition that outgrows its LDS table voids the attempt
  __shared__ __attribute__((aligned(16))) unsigned long long l_key[kLSlots];
  __shared__ unsigned long long l_lo[kLSlots];
  __shared__ unsigned long long l_hi[FX ? kLSlots : 1];
  __shared__ unsigned l_cnt[kLSlots];
  __shared__ unsigned l_first[kLSlots];
  __shared__ unsigned s_used, s_direct;
  __shared__ int s_part;
  __shared__ unsigned long long s_snap[2];
  __shared__ unsigned s_vs[SEG ? kGbRegions + 1 : 1], s_dl[SEG ? kGbRegions : 1];   // the current partition's regions: dense starts (+ its end), physical − dense
  __shared__ unsigned s_ccnt[RECS ? kRecMaxBins : 1], s_cbase[RECS ? kRecMaxBins : 1];   // RECS: this table's records per coarse bin, their reserved places
  const int t = threadIdx.x;
  int part;
  bool multi;
  int64_t r0, r1, seg_lo = 0, seg_hi = 0;
  if (flat == 2) {
    part = 0;
    multi = true;
    r0 = (int64_t)blockIdx.x << kChunkLog2;
    r1 = r0 + ((int64_t)1 << kChunkLog2) < nrows ? r0 + ((int64_t)1 << kChunkLog2) : nrows;
    if (r0 >= nrows) return;
  } else if (flat) {
    part = (int)blockIdx.x;
    multi = false;
    if (seg_rows > 0) {   // behind the RESERVING level-2 scatter: the partition's records lie in its own region of seg_rows places, binstart[] holds its count
      r0 = (int64_t)part * seg_rows;
      // a cursor beyond its region: the attempt is void (the host runs the level again behind a histogram).  The runs that did not fit
      // went to the spare rows, the cursor advanced all the same, so the region's tail was never written: nothing of it is read
      const int64_t have = (int64_t)binstart[part];
      r1 = have <= seg_rows ? r0 + have : r0;
    } else {
      r0 = binstart[part];
      r1 = binstart[part + 1];
    }
  } else {
    // which records?  Equal shares of the partition-ordered records, whatever partitions a share spans: workgroup w takes
    // [w·R, (w + 1)·R) and walks the partition segments inside it, one table per segment.  (One workgroup per CU fits, so a
    // launch is ONE round of equal length — cut per partition into 2^18-row chunks, a partition that is larger than the
    // others (null keys all go to partition 0; skewed keys) added a second round for a few workgroups: 10 % null keys cost 35 %.)
    // A share's end that lies within a quarter share of a partition boundary moves there (both neighbours compute the same
    // point): partitions much smaller than a share are then taken whole — their tables leave as copies, not through the
    // atomic merge — and a large partition's end does not leave a sliver to the next workgroup.
    const int64_t total = (int64_t)binstart[nb], x0 = (int64_t)blockIdx.x * seg_rows, x1 = x0 + seg_rows, tol = seg_rows >> 2;
    if (x0 >= total) return;
    if (t < 2) s_snap[t] = ~0ull;
    if (t == 0) s_part = -1;
    __syncthreads();
    if (t < nb) {
      const int64_t b = (int64_t)binstart[t], d0 = b > x0 ? b - x0 : x0 - b, d1 = b > x1 ? b - x1 : x1 - b;
      if (d0 <= tol) atomicMin(&s_snap[0], ((unsigned long long)d0 << 32) | (unsigned long long)b);
      if (d1 <= tol) atomicMin(&s_snap[1], ((unsigned long long)d1 << 32) | (unsigned long long)b);
    }
    __syncthreads();
    seg_lo = s_snap[0] == ~0ull ? x0 : (int64_t)(s_snap[0] & 0xffffffffull);
    seg_hi = x1 >= total ? total : (s_snap[1] == ~0ull ? x1 : (int64_t)(s_snap[1] & 0xffffffffull));
    if (seg_lo >= seg_hi) return;
    if (t < nb && (int64_t)binstart[t] <= seg_lo && seg_lo < (int64_t)binstart[t + 1]) s_part = t;
    __syncthreads();
    part = s_part;
    multi = true;
    r0 = r1 = 0;
  }
  int sh = 0;
  if (FX) sh = fx_shift(*absmax);
  int64_t gbase = (int64_t)part * (flat == 1 ? kFlatStride : kGStride);
  const int gspecial = flat == 1 ? kSlots : kGSlots;   // where the two special slots sit in the partition's global table
  bool went_direct = false;
  // one pending group per lane: {key, 128-bit sum, count | flags, first row}.  A row with the key of the lane's previous row
  // is added in registers: a key that owns most of a chunk (skewed columns) would otherwise put every lane of every wave
  // on ONE LDS address, and same-address LDS atomics are served one lane at a time.
  const unsigned lkey_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned long long*)l_key;   // LDS byte address of the key table
  bool p_live = false;
  unsigned long long p_key = 0, p_lo = 0, p_hi = 0;
  unsigned p_kw = 0, p_cf = 0, p_first = kNoRow;
  // direct mode, Float64: the scale in *absmax is a GUESS from a sample (gb_direct) — the true exponent range of the addends is
  // tracked here, {largest biased exponent, 0x7ff − smallest} as two 16-bit halves of ONE register (the kernel is at its register
  // limit), and checked against the guess afterwards
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  us2 range_pk = {0, 0};
  auto flush_row = [&](unsigned long long key, unsigned kw, unsigned long long lo, unsigned long long hi, unsigned cf, unsigned row) {
    int j;
    if (__builtin_expect(kw != 0 || key == kEmpty, 0)) {
      j = kw ? kSlots + 1 : kSlots;
    } else {
      // gb_lds_slot, spelled out HERE on the __shared__ arrays themselves: through the function's pointer parameters the table accesses
      // were addressed as generic pointers (54 more vector instructions in this kernel), and with the row loop's body in a lambda
      // another 30 — together the 6 % this pass lost between rounds 3 and 4 (422 → 452 µs; read off the ISA: 1093 → 1165 vector
      // instructions, 1081 now)
      unsigned g = ((((unsigned)key * 0x9E3779B1u) ^ ((unsigned)(key >> 32) * 0x85EBCA6Bu)) >> 20) & (unsigned)(kSlots - 4);
      for (;;) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        u64x2 a, c;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(c) : "v"(lkey_base + g * 8u) : "memory");
        const bool h0 = a.x == key, h1 = a.y == key, h2 = c.x == key, h3 = c.y == key;
        const bool e0 = a.x == kEmpty, e1 = a.y == kEmpty, e2 = c.x == kEmpty, e3 = c.y == kEmpty;
        const bool y0 = h0 || e0, y1 = h1 || e1, y2 = h2 || e2, y3 = h3 || e3;
        if (!(y0 || y1 || y2 || y3)) { g = (g + 4) & (kSlots - 1); continue; }
        j = (int)g + (y0 ? 0 : y1 ? 1 : y2 ? 2 : 3);
        if (h0 || (!e0 && (h1 || (!e1 && (h2 || (!e2 && h3)))))) break;
        if (atomicAdd(&s_used, 1u) >= (unsigned)kSoftLimit) { j = -1; break; }
        const unsigned long long cur = atomicCAS(&l_key[j], kEmpty, key);
        if (cur == kEmpty || cur == key) break;
      }
    }
    if (__builtin_expect(j >= 0, 1)) {
      if (FX) {
        // the returning add (its old value yields the carry) and the read of the group's first row travel together: one wait.
        // Rows of a key arrive mostly in ascending order, so the minimum is rarely lowered — and a plain read costs half an atomic.
        const unsigned long long old = atomicAdd(&l_lo[j], lo);
        const unsigned fr = l_first[j];
        atomicAdd(&l_hi[j], hi + (old + lo < old ? 1ull : 0ull));
        atomicAdd(&l_cnt[j], cf & kCntMask);
        if (fr > row) atomicMin(&l_first[j], row);
      } else {
        atomicAdd(&l_lo[j], lo);
        atomicAdd(&l_cnt[j], cf & kCntMask);
        atomicMin(&l_first[j], row);   // nothing here waits for LDS: fire and forget beats look-before-you-lower
      }
      if (__builtin_expect(cf & ~kCntMask, 0)) atomicOr(&l_cnt[j], cf & ~kCntMask);
    } else if (flat == 1) {
      atomicExch(overflow, 1u);   // no global table behind a flat partition
    } else {
      went_direct = true;
      const long long gs = gb_global_slot(gt.key, gbase, key, overflow);
      if (gs >= 0) gb_global_add<FX>(gt, gs, lo, hi, cf, row);
    }
  };
  // software pipeline: the next step's loads (12 per lane) are in flight while this step's rows go through LDS.  The load step
  // only LOADS: anything that consumes a loaded value (the validity bits of the direct mode) would make it wait for memory
  // between the rows of one step, and a step that lies wholly inside the range runs without per-lane guards.
  constexpr int U = 4;   // (direct mode spills 8 registers at 4 rows per step; at 2 it does not and is 5 % slower)
  constexpr int64_t kStep = (int64_t)kThreads * U;
  unsigned long long nk[U], nv[U];
  unsigned nrw[U];                         // the row word — direct mode: the row's two validity BYTES, decoded when the row is processed
  int xs = 0;                              // SEG: the region the loads are in (uniform; only moves forward inside a partition)
  unsigned seg_dl = 0;                     //      its physical − dense offset
  auto load_step = [&](int64_t b, auto full, auto across) {
    constexpr bool kFull = decltype(full)::value;
    constexpr bool kAcross = decltype(across)::value;   // SEG: the step crosses a region boundary — every lane finds its rows' regions
    int64_t phys[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = b + u * kThreads + t;
      phys[u] = i;
      if (SEG) {
        unsigned d = seg_dl;
        if (kAcross) {
          d = s_dl[0];
#pragma unroll
          for (int x = 1; x < kGbRegions; x++) d = i >= (int64_t)s_vs[x] ? s_dl[x] : d;   // the last region starting at or before i (empty ones start where the next does)
        }
        phys[u] = i + (int64_t)d;
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = b + u * kThreads + t;
      const bool in = kFull || i < r1;
      nk[u] = in ? __builtin_nontemporal_load(&keys[phys[u]]) : 0ull;
      nv[u] = in ? __builtin_nontemporal_load(&vals[phys[u]]) : 0ull;
    }
    if (!DIRECT) {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int64_t i = b + u * kThreads + t;
        nrw[u] = (kFull || i < r1) ? __builtin_nontemporal_load(&prows[phys[u]]) : 0u;
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) nrw[u] = 0xFFFFu;
      if (kvalid) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int64_t i = b + u * kThreads + t;
          if (kFull || i < r1) nrw[u] = (nrw[u] & 0xFF00u) | kvalid[(koff + i) >> 3];
        }
      }
      if (vvalid) {
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int64_t i = b + u * kThreads + t;
          if (kFull || i < r1) nrw[u] = (nrw[u] & 0x00FFu) | ((unsigned)vvalid[(voff + i) >> 3] << 8);
        }
      }
    }
  };
  auto load_any = [&](int64_t b) {
    if (SEG) {
      if (b >= r1) return;
      while (xs < kGbRegions - 1 && b >= (int64_t)s_vs[xs + 1]) xs++;   // (uniform)
      seg_dl = s_dl[xs];
      const int64_t region_end = (int64_t)s_vs[xs + 1], lim = region_end < r1 ? region_end : r1;
      if (b + kStep <= lim) load_step(b, std::true_type{}, std::false_type{});
      else load_step(b, std::false_type{}, std::true_type{});
      return;
    }
    if (b + kStep <= r1) load_step(b, std::true_type{}, std::false_type{});
    else if (b < r1) load_step(b, std::false_type{}, std::false_type{});
  };
  for (;;) {   // one pass per partition segment of this workgroup's share (flat modes: exactly one)
  if (flat == 0) {
    const int64_t b0 = binstart[part], b1 = binstart[part + 1];
    r0 = seg_lo > b0 ? seg_lo : b0;
    r1 = seg_hi < b1 ? seg_hi : b1;
    multi = !(r0 == b0 && r1 == b1);   // the whole partition is ours: its table leaves as a copy
    gbase = (int64_t)part * kGStride;
  }
  if (r0 < r1) {
  for (int j = t; j < kLSlots; j += kThreads) { l_key[j] = (DIRECT && seed_keys && j < kSlots) ? seed_keys[j] : kEmpty; l_lo[j] = 0; if (FX) l_hi[j] = 0; l_cnt[j] = 0; l_first[j] = kNoRow; }
  if (t == 0) { s_used = DIRECT && seed_keys ? seed_used : 0u; s_direct = 0; }
  if (RECS) for (int b = t; b < kRecMaxBins; b += kThreads) s_ccnt[b] = 0;
  if (SEG) {
    if (t <= kGbRegions) s_vs[t] = seg_vstart[part * kGbRegions + t];
    if (t < kGbRegions) s_dl[t] = seg_delta[part * kGbRegions + t];
    xs = 0;
  }
  __syncthreads();
  went_direct = false;
  p_live = false;
  load_any(r0);
  // The first step's loads are waited for HERE.  Entering the loop with them pending, the compiler's wait-count pass merges that
  // state with the back edge's and puts "at most 3 loads outstanding" in front of the row processing — which in the steady state
  // means waiting for the prefetch just issued: the pipeline below never overlapped anything (seen in the ISA, not in a profile).
  __builtin_amdgcn_s_waitcnt(0);
  // One step of U rows per lane.  LEAN = false: a row with the key of the lane's previous row joins the pending group in registers
  // (see above: skewed columns).  LEAN = true (chosen by the host from the quick look: rows 1024 apart — a lane's consecutive rows —
  // rarely share a key): every row goes to the table at once; the pending group's bookkeeping, ≈ 20 of a row's ≈ 140 vector
  // instructions and nine registers, would buy nothing (every row flushed the pending group anyway).
  for (int64_t b = r0; b < r1; b += kStep) {
    // the attempt is void already (a partition's global table is full: its estimate was far off): stop feeding tables nobody will
    // read — every step in the direct mode (one table takes all the merges), every 16th step behind the cut (a mis-estimated or
    // heavy-tailed column kept the workgroups probing full tables for the rest of their shares: 0.9 s once)
    if (DIRECT || (flat == 0 && (((b - r0) / kStep) & 15) == 15)) {
      if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    }
    unsigned long long k[U], v[U];
    unsigned rw[U];
#pragma unroll
    for (int u = 0; u < U; u++) { k[u] = nk[u]; v[u] = nv[u]; rw[u] = nrw[u]; }
    load_any(b + kStep);
    const unsigned left = r1 - b < kStep ? (unsigned)(r1 - b) : (unsigned)kStep;   // rows of this step (uniform)
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool live = (unsigned)(u * kThreads + t) < left;   // a slot past the end adds nothing to whatever is pending
      unsigned rwu = rw[u];
      if (DIRECT) {   // in 32 bits: rows are below 2^29 here, and only the low three bits of the bit positions matter
        const unsigned il = (unsigned)b + (unsigned)(u * kThreads + t);
        rwu = il | (((rwu >> (((unsigned)koff + il) & 7u)) & 1u) ? 0u : kKeyNull) | (((rwu >> (8u + (((unsigned)voff + il) & 7u))) & 1u) ? 0u : kValNull);
      }
      const unsigned row = live ? rwu & kRowMask : kNoRow;
      // the addend, without branches: a null value adds nothing and is not counted; ±inf / NaN are counted and flagged
      const bool has = live && !(rwu & kValNull);
      unsigned long long lo, hi = 0;
      unsigned cf = has ? 1u : 0u;
      if (FX) {
        const double x = __builtin_bit_cast(double, v[u]);
        const bool fin = fx_finite(x);
        fx_split(has && fin ? x : 0.0, sh, &lo, &hi);
        cf |= has && !fin ? fx_flag(x) << 29 : 0u;
        if (DIRECT) {
          const unsigned h32 = (unsigned)(v[u] >> 32), e = (h32 >> 20) & 0x7ffu, ee = e ? e : 1u;   // a denormal counts as exponent 1 (fx_split)
          const bool counts = has && fin && ((h32 & 0x7fffffffu) | (unsigned)v[u]) != 0u;
          const us2 pk = {(unsigned short)(counts ? ee : 0u), (unsigned short)(counts ? 0x7ffu - ee : 0u)};
          range_pk = __builtin_elementwise_max(range_pk, pk);
        }
      } else {
        lo = has ? v[u] : 0ull;
      }
      const unsigned kw = rwu & kKeyNull;
      if (DIRECT) k[u] = kw ? 0ull : k[u];   // one key for all null rows (the partition pass has done this for the other modes)
      if (LEAN) {
        if (live) flush_row(k[u], kw, lo, hi, cf, row);
        continue;
      }
      const bool same = !live || (p_live && p_key == k[u] && p_kw == kw);
      if (!same) {
        if (p_live) flush_row(p_key, p_kw, p_lo, p_hi, p_cf, p_first);
        p_key = k[u]; p_kw = kw; p_lo = 0; p_hi = 0; p_cf = 0; p_first = kNoRow;
        p_live = true;
      }
      const unsigned long long nl = p_lo + lo;
      p_hi += hi + (nl < p_lo ? 1ull : 0ull);
      p_lo = nl;
      p_cf = (p_cf + (cf & kCntMask)) | (cf & ~kCntMask);   // < 2^19 rows per chunk: the count cannot reach the flag bits
      p_first = row < p_first ? row : p_first;
    }
  }
  if (!LEAN && p_live) flush_row(p_key, p_kw, p_lo, p_hi, p_cf, p_first);
  if (went_direct) s_direct = 1;
  __syncthreads();
  if (RECS) {
    // (flat == 1: always the only workgroup of its partition, and a full table voids the attempt)
    for (int j = t; j < kLSlots; j += kThreads) {
      const unsigned fr = l_first[j];
      if (fr != kNoRow) atomicAdd(&s_ccnt[fr >> ro.cshift], 1u);
    }
    __syncthreads();
    for (int b = t; b < ro.ncoarse; b += kThreads) {
      const unsigned cn = s_ccnt[b];
      s_cbase[b] = cn ? atomicAdd(&ro.ccursor[b], cn) : 0u;   // ≤ 2^cshift in all: a bin of R rows holds at most R first rows
      s_ccnt[b] = 0;
    }
    __syncthreads();
    for (int j = t; j < kLSlots; j += kThreads) {
      const unsigned fr = l_first[j];
      if (fr == kNoRow) continue;
      const unsigned cb = fr >> ro.cshift;
      const size_t pos = ((size_t)cb << ro.cshift) + s_cbase[cb] + atomicAdd(&s_ccnt[cb], 1u);
      // the all-ones key lives in slot kSlots (its key word is the empty marker's), the null group in kSlots + 1 (key slot: the fresh buffer's zero)
      gb_rec_store(&ro.recs[pos], j == kSlots + 1 ? 0ull : l_key[j], l_lo[j], FX ? l_hi[j] : 0ull, l_cnt[j], fr | (j == kSlots + 1 ? kKeyNull : 0u));
    }
  } else if (!multi && !s_direct) {
    // the only workgroup of this partition and everything is in LDS: the table leaves as it is
    for (int j = t; j < kSlots; j += kThreads) {
      gt.key[gbase + j] = l_key[j];
      gt.lo[gbase + j] = l_lo[j];
      if (FX) gt.hi[gbase + j] = l_hi[j];
      gt.cnt[gbase + j] = l_cnt[j];
      gt.first[gbase + j] = l_first[j];
    }
    if (t < 2) {
      const int j = kSlots + t;
      const int64_t g = gbase + gspecial + t;
      gt.key[g] = l_key[j]; gt.lo[g] = l_lo[j]; if (FX) gt.hi[g] = l_hi[j]; gt.cnt[g] = l_cnt[j]; gt.first[g] = l_first[j];
    } else if (flat == 1 && t < kFlatStride - kSlots) {
      gt.first[gbase + kSlots + t] = kNoRow;   // the stride's padding: nothing was memset in flat mode
    }
  } else {
    for (int j = t; j < kLSlots; j += kThreads) {
      const unsigned fr = l_first[j];
      if (DIRECT && seed_keys && (j >= kSlots || seed_keys[j] != kEmpty)) {
        // a seeded slot (and the two special ones, whose place is fixed): this workgroup's share of it, used or not
        const size_t o = (size_t)blockIdx.x * kLSlots + (size_t)j;
        st.lo[o] = l_lo[j];
        if (FX) st.hi[o] = l_hi[j];
        st.cnt[o] = l_cnt[j];
        st.first[o] = fr;
        continue;
      }
      if (fr == kNoRow) continue;
      long long gs;
      if (j >= kSlots) gs = gbase + kGSlots + (j - kSlots);
      else gs = gb_global_slot(gt.key, gbase, l_key[j], overflow);
      if (gs >= 0) gb_global_add<FX>(gt, gs, l_lo[j], FX ? l_hi[j] : 0ull, l_cnt[j], fr);
    }
  }
  __syncthreads();   // the next segment re-initialises the table
  }
  if (flat != 0 || (int64_t)binstart[part + 1] >= seg_hi) break;
  part++;
  }
  if (DIRECT && FX && tile_range) {
    unsigned emax = range_pk.x, imin = range_pk.y;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned a = __shfl_down(emax, o, 64), b = __shfl_down(imin, o, 64);
      emax = a > emax ? a : emax;
      imin = b > imin ? b : imin;
    }
    if ((t & 63) == 0) {   // 16 atomics per address: a workgroup's own pair of words
      if (emax) atomicMax(&tile_range[2 * blockIdx.x], emax);
      if (imin) atomicMax(&tile_range[2 * blockIdx.x + 1], imin);
    }
  }
}

// ---- 4: first-seen ranks and the output ------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gb_mark_kernel(const unsigned* __restrict__ first, int64_t nslots, unsigned long long* __restrict__ firsts) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < nslots; s += stride) {
    const unsigned fr = first[s];
    if (fr != kNoRow) atomicOr(&firsts[fr >> 6], 1ull << (fr & 63));
  }
}

template <bool FX>
__global__ __launch_bounds__(kBlock) void gb_emit_kernel(GbTable gt, int64_t nslots, const unsigned long long* __restrict__ firsts,
                                                          const unsigned* __restrict__ wordprefix, const int64_t* __restrict__ tileoff,
                                                          const unsigned long long* __restrict__ absmax, unsigned long long* __restrict__ out_keys,
                                                          unsigned long long* __restrict__ out_sums, long long* __restrict__ out_counts,
                                                          long long* __restrict__ out_first_rows, int* __restrict__ null_id, int gstride, int gspecial) {
  int sh = 0;
  if (FX) sh = fx_shift(*absmax);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x; s < nslots; s += stride) {
    const unsigned fr = gt.first[s];
    if (fr == kNoRow) continue;
    const unsigned id = rank_of_row(fr, firsts, wordprefix, tileoff);
    const int in_part = (int)(s % gstride);
    unsigned long long key = gt.key[s];
    if (in_part == gspecial) key = kEmpty;
    if (in_part == gspecial + 1) { key = 0; *null_id = (int)id; }   // the null group's key slot keeps the fresh buffer's zero
    out_keys[id] = key;
    const unsigned cf = gt.cnt[s];
    out_counts[id] = (long long)(cf & kCntMask);
    if (FX) {
      const unsigned f = cf >> 29;
      double r;
      if (f) r = (f & 1u) || (f & 6u) == 6u ? __builtin_nan("") : ((f & 2u) ? __builtin_inf() : -__builtin_inf());
      else r = fx_to_double(gt.lo[s], gt.hi[s], sh);
      out_sums[id] = __builtin_bit_cast(unsigned long long, r);
    } else {
      out_sums[id] = gt.lo[s];
    }
    if (out_first_rows) out_first_rows[id] = (long long)fr;
  }
}

// the second reserving scatter: coarse bin → its fine bins of 4096 rows.  One workgroup per (coarse bin, tile of kRecTile records);
// the grid covers every tile a full bin would have, a workgroup beyond its bin's count leaves at once.
constexpr int kRecTile = 2048;
__global__ __launch_bounds__(kThreads) void gbr_split_kernel(const GbRec* __restrict__ recs1, const unsigned* __restrict__ ccursor, int cshift, int fpc_log2,
                                                             int tiles_per_coarse, GbRec* __restrict__ recs2, unsigned* __restrict__ fcursor) {
  __shared__ unsigned s_cnt[kRecMaxBins], s_base[kRecMaxBins];
  const int t = threadIdx.x;
  const unsigned cb = blockIdx.x / (unsigned)tiles_per_coarse, tile = blockIdx.x % (unsigned)tiles_per_coarse;
  const unsigned cnt = ccursor[cb], lo = tile * (unsigned)kRecTile;
  if (lo >= cnt) return;
  const unsigned hi = cnt - lo < (unsigned)kRecTile ? cnt : lo + (unsigned)kRecTile;
  const int fpc = 1 << fpc_log2;
  for (int b = t; b < fpc; b += kThreads) s_cnt[b] = 0;
  __syncthreads();
  constexpr int U = kRecTile / kThreads;
  GbRec r[U];
  unsigned fl[U], rank[U];
  const GbRec* src = recs1 + ((size_t)cb << cshift);
#pragma unroll
  for (int u = 0; u < U; u++) {
    const unsigned i = lo + (unsigned)(u * kThreads + t);
    fl[u] = ~0u;
    if (i < hi) {
      r[u] = gb_rec_load(src + i);
      fl[u] = ((r[u].first & kRowMask) >> kRecFineLog2) & (unsigned)(fpc - 1);
      rank[u] = atomicAdd(&s_cnt[fl[u]], 1u);
    }
  }
  __syncthreads();
  for (int b = t; b < fpc; b += kThreads) {
    const unsigned cn = s_cnt[b];
    if (cn) s_base[b] = atomicAdd(&fcursor[((size_t)cb << fpc_log2) + b], cn);
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < U; u++) {
    if (fl[u] == ~0u) continue;
    const size_t fb = ((size_t)cb << fpc_log2) + fl[u];
    gb_rec_store(&recs2[(fb << kRecFineLog2) + s_base[fl[u]] + rank[u]], r[u].key, r[u].lo, r[u].hi, r[u].cnt, r[u].first);
  }
}

// one workgroup per fine bin: rank = number of first rows below the record's own inside the bin (bitmap in LDS), id = groups of the
// bins before (fprefix) + rank; the four output columns of the bin are one contiguous id range each.  The columns are put in rank
// order in LDS and leave as whole lines: 8-byte stores straight to out_*[id] were written through sector by sector (PMC: 1.48 GB
// of write traffic for 0.5 GiB of output, 620 µs at 2^24 groups) although a bin's ids span only a few KiB.
constexpr int kRecEmitThreads = 1024, kRecEmitU = (1 << kRecFineLog2) / kRecEmitThreads;
template <bool FX>
__global__ __launch_bounds__(kRecEmitThreads) void gbr_emit_kernel(const GbRec* __restrict__ recs2, const unsigned* __restrict__ fcursor, const int64_t* __restrict__ fprefix,
                                                                    const unsigned long long* __restrict__ absmax, unsigned long long* __restrict__ out_keys,
                                                                    unsigned long long* __restrict__ out_sums, long long* __restrict__ out_counts,
                                                                    long long* __restrict__ out_first_rows, int* __restrict__ null_id) {
  __shared__ unsigned long long s_bits[64];
  __shared__ unsigned s_pre[64];
  __shared__ unsigned long long s_a[1 << kRecFineLog2], s_b[1 << kRecFineLog2];
  const int t = threadIdx.x;
  const unsigned cnt = fcursor[blockIdx.x];
  if (!cnt) return;
  const int64_t base = fprefix[blockIdx.x];
  const GbRec* src = recs2 + ((size_t)blockIdx.x << kRecFineLog2);
  if (t < 64) s_bits[t] = 0;
  GbRec r[kRecEmitU];
#pragma unroll
  for (int u = 0; u < kRecEmitU; u++) {
    const unsigned i = (unsigned)(u * kRecEmitThreads + t);
    r[u].first = kNoRow;
    if (i < cnt) r[u] = gb_rec_load(src + i);
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kRecEmitU; u++)
    if ((unsigned)(u * kRecEmitThreads + t) < cnt) {
      const unsigned fi = r[u].first & (unsigned)((1 << kRecFineLog2) - 1);
      atomicOr(&s_bits[fi >> 6], 1ull << (fi & 63));
    }
  __syncthreads();
  if (t < 64) {
    const unsigned pc = (unsigned)__popcll(s_bits[t]);
    unsigned inc = pc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned v = __shfl_up(inc, o, 64);
      if (t >= o) inc += v;
    }
    s_pre[t] = inc - pc;
  }
  __syncthreads();
  int sh = 0;
  if (FX) sh = fx_shift(*absmax);
  unsigned rank[kRecEmitU];
#pragma unroll
  for (int u = 0; u < kRecEmitU; u++) {
    rank[u] = ~0u;
    if ((unsigned)(u * kRecEmitThreads + t) < cnt) {
      const unsigned fi = r[u].first & (unsigned)((1 << kRecFineLog2) - 1);
      rank[u] = s_pre[fi >> 6] + (unsigned)__popcll(s_bits[fi >> 6] & ((1ull << (fi & 63)) - 1));
      if (r[u].first & kKeyNull) *null_id = (int)(base + rank[u]);
      unsigned long long sum = r[u].lo;
      if (FX) {
        const unsigned f = r[u].cnt >> 29;
        double d;
        if (f) d = (f & 1u) || (f & 6u) == 6u ? __builtin_nan("") : ((f & 2u) ? __builtin_inf() : -__builtin_inf());
        else d = fx_to_double(r[u].lo, r[u].hi, sh);
        sum = __builtin_bit_cast(unsigned long long, d);
      }
      s_a[rank[u]] = r[u].key;
      s_b[rank[u]] = sum;
    }
  }
  __syncthreads();
  for (unsigned i = t; i < cnt; i += kRecEmitThreads) {
    out_keys[base + i] = s_a[i];
    out_sums[base + i] = s_b[i];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kRecEmitU; u++)
    if (rank[u] != ~0u) {
      s_a[rank[u]] = (unsigned long long)(r[u].cnt & kCntMask);
      s_b[rank[u]] = (unsigned long long)(r[u].first & kRowMask);
    }
  __syncthreads();
  for (unsigned i = t; i < cnt; i += kRecEmitThreads) {
    out_counts[base + i] = (long long)s_a[i];
    if (out_first_rows) out_first_rows[base + i] = (long long)s_b[i];
  }
}

// ---- very many groups (beyond 1024 partitions of LDS-table size): sort instead of hash ---------------------------------------
// With millions of groups a partition small enough for an LDS table holds only a few thousand rows — initialising and writing
// out the table would cost more than the rows.  The rows are cut into buckets of ≈ 64 by the top bits of gb_mix(key) instead (the
// two partition levels of ah_msd.h: equal keys share a bucket), and one wave per bucket SORTS its rows by (key, row) in
// registers and walks the runs of equal keys: sum, count, first row = the run's first element.  No table, no probing, no
// atomics on the sums; all rows of a key meet in one bucket, so the group records are final when the wave is done — only
// their order (ids = rank of the first row) is global, and that is the bitmap + prefix popcount of the other paths.
constexpr int kGsLocalMax = 256;   // rows one wave takes; a larger bucket (a key with hundreds of rows among millions of groups) voids the attempt
__device__ __forceinline__ unsigned gs_bucket(unsigned long long key, bool knull, int lb) { return knull ? 0u : (unsigned)(gb_mix(key) >> (64 - lb)); }

struct GsColumns {   // level 1 reads the columns
  const unsigned long long* keys; const uint8_t* kvalid; int64_t koff;
  const unsigned long long* vals; const uint8_t* vvalid; int64_t voff;
  __device__ __forceinline__ void load(int64_t i, unsigned long long* k, unsigned long long* v, unsigned* rw) const {
    const bool kv = ah_bit(kvalid, koff + i), vv = ah_bit(vvalid, voff + i);
    *k = kv ? __builtin_nontemporal_load(&keys[i]) : 0ull;     // a null key travels as 0 + the flag
    *v = __builtin_nontemporal_load(&vals[i]);
    *rw = (unsigned)i | (kv ? 0u : kKeyNull) | (vv ? 0u : kValNull);
  }
};
struct GsRecords {   // level 2 reads level 1's output
  const unsigned long long* keys; const unsigned long long* vals; const unsigned* rows;
  __device__ __forceinline__ void load(int64_t i, unsigned long long* k, unsigned long long* v, unsigned* rw) const {
    *k = __builtin_nontemporal_load(&keys[i]);
    *v = __builtin_nontemporal_load(&vals[i]);
    *rw = __builtin_nontemporal_load(&rows[i]);
  }
};

template <typename SRC>
__global__ __launch_bounds__(kThreads) void gs_hist_kernel(SRC src, int64_t n, const unsigned* __restrict__ pstart, int nparents, int lb, int shift,
                                                            unsigned mask, int nb, unsigned* __restrict__ cnt) {
  __shared__ unsigned s_h[kMaxNb2];
  __shared__ unsigned s_cnt[kThreads], s_start[kThreads], s_wsum[kThreads / 64];
  __shared__ int s_pick;
  const TileRange r = ms_tile(pstart, nparents, n, s_cnt, s_start, s_wsum, &s_pick);
  if (r.parent < 0) return;
  for (int b = threadIdx.x; b < nb; b += kThreads) s_h[b] = 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    const int64_t i = r.lo + u * kThreads + threadIdx.x;
    if (i < r.hi) {
      unsigned long long k, v; unsigned rw;
      src.load(i, &k, &v, &rw);
      atomicAdd(&s_h[(gs_bucket(k, rw & kKeyNull, lb) >> shift) & mask], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kThreads) cnt[r.id * nb + b] = s_h[b];
}

// RES (level 2 of the two-level cut, no histogram pass in front of it): child partition c = parent · nb + digit owns the FIXED region
// [c · cap, (c + 1) · cap) of the record arrays; a tile reserves its run of every digit with one returning atomicAdd on the child's
// cursor (toffs is null).  The children of hash partitions of millions of evenly spread keys differ by a few per cent (cap = 1.25 ×
// the even share: ≥ 8 σ); a child that overflows all the same — a key with tens of thousands of rows — sends the run to kMsTile spare
// rows at trash_base, raises bit 2 of *redo, and the host runs the level again behind a histogram.  All tiles of a parent run on one
// XCD (ms_tile), so the runs a child receives meet in one L2 as they do behind the offsets table.
// (≤ 64 registers: two workgroups per CU.  At 68–71 — the 64-bit destinations of four rows — only one fitted and the pass ran at
// 3.5 TB/s where the one-level scatter, built the same way, runs at 5.0.)
template <typename SRC, bool RES = false, bool RANGE = true>   // RANGE = false: tile_max is null (level 2) — the range bookkeeping and its registers are compiled out
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void gs_scatter_kernel(SRC src, int64_t n, const unsigned* __restrict__ pstart, int nparents, int lb, int shift,
                                                               unsigned mask, int nb, const unsigned* __restrict__ toffs,
                                                               unsigned long long* __restrict__ out_keys, unsigned long long* __restrict__ out_vals,
                                                               unsigned* __restrict__ out_rows, unsigned long long* __restrict__ tile_max,
                                                               unsigned* __restrict__ cursor = nullptr, unsigned cap = 0, unsigned trash_base = 0,
                                                               unsigned* __restrict__ redo = nullptr, const unsigned* __restrict__ pend = nullptr,
                                                               int sub8 = 0) {
  // sub8: the "parents" are the 8 per-XCD regions of each true parent, numbered q = (p >> 3) · 64 + x · 8 + (p & 7) (gs_subregions_kernel):
  // q & 7 = p & 7, so all eight regions of a parent are tiled on ONE XCD (ms_tile's classes) and its children's runs meet in one L2
  __shared__ unsigned s_cnt[kMaxNb2], s_start[kMaxNb2], s_goff[kMaxNb2], s_wsum[kThreads / 64];
  __shared__ unsigned s_a[kThreads], s_b[kThreads];
  __shared__ unsigned long long s_stage[kMsTile];
  __shared__ uint16_t s_bin[kMsTile];
  __shared__ unsigned long long s_max[kThreads / 64];
  __shared__ unsigned s_imin[kThreads / 64];
  __shared__ int s_pick;
  __shared__ unsigned s_carry;
  const TileRange r = ms_tile(pstart, nparents, n, s_a, s_b, s_wsum, &s_pick, pend);
  if (r.parent < 0) return;
  const int t = threadIdx.x;
  for (int b = t; b < nb; b += kThreads) s_cnt[b] = 0;
  unsigned long long k[kMsRows], v[kMsRows], vmax = 0;
  unsigned vimin = 0;
  unsigned rw[kMsRows], dg[kMsRows], rank[kMsRows];
  bool live[kMsRows];
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    const int64_t i = r.lo + u * kThreads + t;
    live[u] = i < r.hi;
    k[u] = 0; v[u] = 0; rw[u] = 0;
    if (live[u]) src.load(i, &k[u], &v[u], &rw[u]);
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    dg[u] = (gs_bucket(k[u], rw[u] & kKeyNull, lb) >> shift) & mask;
    rank[u] = live[u] ? atomicAdd(&s_cnt[dg[u]], 1u) : 0u;
    const unsigned long long a = v[u] & 0x7fffffffffffffffull;
    if (RANGE && tile_max && live[u] && !(rw[u] & kValNull) && (a >> 52) != 0x7ff && a != 0) { vmax = a > vmax ? a : vmax; vimin = fx_inv_exp(a) > vimin ? fx_inv_exp(a) : vimin; }
  }
  if (RANGE && tile_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long x = __shfl_down(vmax, o, 64);
      const unsigned xi = __shfl_down(vimin, o, 64);
      vmax = x > vmax ? x : vmax;
      vimin = xi > vimin ? xi : vimin;
    }
    if ((t & 63) == 0) { s_max[t >> 6] = vmax; s_imin[t >> 6] = vimin; }
  }
  __syncthreads();
  unsigned carry = 0;
  for (int h = 0; h * kThreads < nb; h++) {   // exclusive scan over nb ≤ 2048 digit counts, 1024 at a time
    s_a[t] = t + h * kThreads < nb ? s_cnt[t + h * kThreads] : 0u;
    __syncthreads();
    block_excl_scan(s_a, s_b, s_wsum, kThreads);
    if (t + h * kThreads < nb) {
      const int d = t + h * kThreads;
      const unsigned st = carry + s_b[t];
      s_start[d] = st;
      if (RES) {
        const unsigned cn = s_cnt[d];
        unsigned go = trash_base;
        if (cn) {
          const unsigned parent = sub8 ? (((unsigned)r.parent >> 6) << 3) | ((unsigned)r.parent & 7u) : (unsigned)r.parent;
          const unsigned child = parent * (unsigned)nb + (unsigned)d;
          const unsigned at = atomicAdd(&cursor[child], cn);
          if (at + cn <= cap) go = child * cap + at - st;   // (mod 2^32: a region may start below the tile's own prefix)
          else atomicOr(redo, 4u);
        }
        s_goff[d] = go;
      } else {
        s_goff[d] = toffs[r.id * nb + d] - st;
      }
    }
    if (t == kThreads - 1) s_carry = s_b[t] + s_a[t];
    __syncthreads();
    carry += s_carry;
    __syncthreads();
  }
  if (RANGE && tile_max && t == 0) {
    unsigned long long x = s_max[0];
    unsigned xi = s_imin[0];
    for (int w = 1; w < kThreads / 64; w++) { x = s_max[w] > x ? s_max[w] : x; xi = s_imin[w] > xi ? s_imin[w] : xi; }
    tile_max[2 * r.id] = x;
    tile_max[2 * r.id + 1] = xi;
  }
  const int tile_n = (int)(r.hi - r.lo);
  constexpr unsigned kNone = ~0u;   // positions are below 2^32 − 1 (≤ 2^29 rows, regions within 1.5 n)
  unsigned pos[kMsRows];            // the row's place in the staged tile, once: digit and rank are dead from here on (four registers and
#pragma unroll                      // eight LDS reads less than looking s_start up again in each of the three rounds)
  for (int u = 0; u < kMsRows; u++) {
    pos[u] = live[u] ? s_start[dg[u]] + rank[u] : kNone;
    if (pos[u] != kNone) { s_stage[pos[u]] = k[u]; s_bin[pos[u]] = (uint16_t)dg[u]; }
  }
  __syncthreads();
  unsigned dst[kMsRows];
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    const int q = u * kThreads + t;
    dst[u] = q < tile_n ? s_goff[s_bin[q]] + (unsigned)q : kNone;   // (mod 2^32 in both modes: the offsets table's entries are positions too)
    if (dst[u] != kNone) out_keys[dst[u]] = s_stage[q];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (pos[u] != kNone) s_stage[pos[u]] = v[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (dst[u] != kNone) out_vals[dst[u]] = s_stage[u * kThreads + t];
  __syncthreads();
  unsigned* s_stage32 = reinterpret_cast<unsigned*>(s_stage);
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (pos[u] != kNone) s_stage32[pos[u]] = rw[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (dst[u] != kNone) out_rows[dst[u]] = s_stage32[u * kThreads + t];
}

// sort element: (key, null-key flag | row) decides; li = position in the bucket before the sort (| null-value flag) finds the value
struct GsEl {
  unsigned long long k;
  unsigned r, li;
  static __device__ __forceinline__ bool less(const GsEl& a, const GsEl& b) { return a.k < b.k || (a.k == b.k && a.r < b.r); }
  static __device__ __forceinline__ GsEl xchg(const GsEl& a, int j, int lane) {
    GsEl o;
    o.k = ((unsigned long long)ms_xor_lane((unsigned)(a.k >> 32), j, lane) << 32) | ms_xor_lane((unsigned)a.k, j, lane);
    o.r = ms_xor_lane(a.r, j, lane);
    o.li = ms_xor_lane(a.li, j, lane);
    return o;
  }
};

template <bool FX, int SLOTS>
__device__ __forceinline__ void gs_bucket_groups(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ vals,
                                                 const unsigned* __restrict__ rows, unsigned s, int m, int lane, int sh, unsigned long long* s_val,
                                                 unsigned long long* s_k, unsigned* s_r, unsigned* s_li, unsigned long long* __restrict__ g_key,
                                                 unsigned long long* __restrict__ g_sum, unsigned* __restrict__ g_cnt, unsigned* __restrict__ g_first,
                                                 unsigned long long* __restrict__ firsts) {
  GsEl x[SLOTS];
#pragma unroll
  for (int sl = 0; sl < SLOTS; sl++) {
    const int i = sl * 64 + lane;
    x[sl].k = ~0ull; x[sl].r = ~0u; x[sl].li = 0;   // slots past the bucket hold the largest element
    if (i < m) {
      const unsigned rw = rows[s + i];
      x[sl].k = keys[s + i];
      x[sl].r = (rw & kKeyNull) | (rw & kRowMask);
      x[sl].li = (unsigned)i | ((rw & kValNull) ? 0x80000000u : 0u);
      s_val[i] = vals[s + i];
    }
  }
  ms_bitonic<GsEl, SLOTS>(x, lane);
#pragma unroll
  for (int sl = 0; sl < SLOTS; sl++) { const int i = sl * 64 + lane; s_k[i] = x[sl].k; s_r[i] = x[sl].r; s_li[i] = x[sl].li; }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // the wave's own LDS writes before its reads of other lanes' slots
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < m; i += 64) {
    const unsigned long long k = s_k[i];
    const unsigned ri = s_r[i], nullbit = ri & kKeyNull;
    if (i > 0 && s_k[i - 1] == k && (s_r[i - 1] & kKeyNull) == nullbit) continue;   // not the first row of its key
    unsigned long long lo = 0, hi = 0;
    unsigned cnt = 0, flags = 0;
    for (int j = i; j < m && s_k[j] == k && (s_r[j] & kKeyNull) == nullbit; j++) {
      const unsigned li = s_li[j];
      if (li & 0x80000000u) continue;   // null value: neither summed nor counted
      const unsigned long long vb = s_val[li & 0x7fffffffu];
      cnt++;
      if (FX) {
        const double d = __builtin_bit_cast(double, vb);
        if (fx_finite(d)) {
          unsigned long long l, h;
          fx_split(d, sh, &l, &h);
          const unsigned long long nl = lo + l;
          hi += h + (nl < lo ? 1ull : 0ull);
          lo = nl;
        } else flags |= fx_flag(d);
      } else lo += vb;
    }
    unsigned long long sum = lo;
    if (FX) {
      double rr;
      if (flags) rr = (flags & 1u) || (flags & 6u) == 6u ? __builtin_nan("") : ((flags & 2u) ? __builtin_inf() : -__builtin_inf());
      else rr = fx_to_double(lo, hi, sh);
      sum = __builtin_bit_cast(unsigned long long, rr);
    }
    const unsigned first = ri & kRowMask;
    const int64_t pos = (int64_t)s + i;
    g_key[pos] = nullbit ? 0ull : k;
    g_sum[pos] = sum;
    g_cnt[pos] = cnt | nullbit;
    g_first[pos] = first;
    atomicOr(&firsts[first >> 6], 1ull << (first & 63));
  }
}

template <bool FX>
__global__ __launch_bounds__(256) void gs_local_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ vals,
                                                        const unsigned* __restrict__ rows, const unsigned* __restrict__ bstart, int64_t nbuckets,
                                                        const unsigned long long* __restrict__ absmax, unsigned long long* __restrict__ g_key,
                                                        unsigned long long* __restrict__ g_sum, unsigned* __restrict__ g_cnt,
                                                        unsigned* __restrict__ g_first, unsigned long long* __restrict__ firsts,
                                                        unsigned* __restrict__ oversize) {
  __shared__ unsigned long long s_val[4][kGsLocalMax], s_k[4][kGsLocalMax];
  __shared__ unsigned s_r[4][kGsLocalMax], s_li[4][kGsLocalMax];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + w;
  if (b >= nbuckets) return;
  const unsigned s = bstart[b], e = bstart[b + 1];
  const int m = (int)(e - s);
  if (m <= 0) return;
  if (m > kGsLocalMax) { if (lane == 0) atomicMax(oversize, (unsigned)m); return; }
  int sh = 0;
  if (FX) sh = fx_shift(*absmax);
#define AH_GS(SL) gs_bucket_groups<FX, SL>(keys, vals, rows, s, m, lane, sh, s_val[w], s_k[w], s_r[w], s_li[w], g_key, g_sum, g_cnt, g_first, firsts)
  if (m <= 64) AH_GS(1);
  else if (m <= 128) AH_GS(2);
  else AH_GS(4);
#undef AH_GS
}

template <bool FX>
__global__ __launch_bounds__(kBlock) void gs_emit_kernel(const unsigned long long* __restrict__ g_key, const unsigned long long* __restrict__ g_sum,
                                                          const unsigned* __restrict__ g_cnt, const unsigned* __restrict__ g_first, int64_t n,
                                                          const unsigned long long* __restrict__ firsts, const unsigned* __restrict__ wordprefix,
                                                          const int64_t* __restrict__ tileoff, unsigned long long* __restrict__ out_keys,
                                                          unsigned long long* __restrict__ out_sums, long long* __restrict__ out_counts,
                                                          long long* __restrict__ out_first_rows, int* __restrict__ null_id) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n; p += stride) {
    const unsigned fr = __builtin_nontemporal_load(&g_first[p]);
    if (fr == kNoRow) continue;
    const unsigned id = rank_of_row(fr, firsts, wordprefix, tileoff);
    const unsigned c = g_cnt[p];
    out_keys[id] = g_key[p];
    out_sums[id] = g_sum[p];
    out_counts[id] = (long long)(c & 0x7fffffffu);
    if (out_first_rows) out_first_rows[id] = (long long)fr;
    if (c & kKeyNull) *null_id = (int)id;
  }
}

}  // namespace

// behind the RESERVING level-1 scatter: parent p = the region [p · cap, p · cap + rows), rows = its cursor — or nothing at all when the
// cursor passed the region's end (the attempt is void: bit 2 of *redo; the runs that did not fit lie elsewhere and the region's tail
// was never written).  Float64: the value range of the call (gb_max_kernel has reduced it) is checked here too (fx_range_check_kernel).
__global__ void gs_regions_kernel(const unsigned* __restrict__ cursor, int nparents, unsigned cap, unsigned* __restrict__ pstart, unsigned* __restrict__ pend,
                                  unsigned* __restrict__ redo, const unsigned long long* __restrict__ range) {
  const int p = threadIdx.x;
  if (p < nparents) {
    const unsigned have = cursor[p], lo = (unsigned)p * cap;
    pstart[p] = lo;
    pend[p] = have <= cap ? lo + have : lo;
    if (have > cap) atomicOr(redo, 4u);
  }
  if (p == 0 && range && fx_wide(range[0], range[1])) atomicOr(redo, 2u);
}

// behind gb_scatter_kernel<…, RESERVE> as the FIRST level of the two-level cut: region (p, x) — parent p's rows from the tiles XCD x
// ran — becomes "parent" q = (p >> 3) · 64 + x · 8 + (p & 7) of the second level's tiling: [sstart[q], send[q]).  A cursor beyond its
// region voids the attempt (nothing of any region is read).  The scatter's per-tile value ranges are reduced and checked here
// (what gb_segments_kernel does behind the one-level cut).
__global__ __launch_bounds__(1024) void gs_subregions_kernel(const unsigned* __restrict__ rstart, const unsigned* __restrict__ rcap, const unsigned* __restrict__ cursor,
                                                             int nb1, unsigned* __restrict__ sstart, unsigned* __restrict__ send, unsigned* __restrict__ redo,
                                                             const unsigned long long* __restrict__ tile_rng, int64_t ntiles, unsigned long long* __restrict__ range) {
  __shared__ unsigned long long s_max[16], s_imin[16];
  const int t = threadIdx.x;
  bool over = (*redo & 4u) != 0;   // void already (gb_layout_kernel: the regions did not fit the arrays)
  unsigned have = 0, lo = 0;
  int q = -1;
  if (t < nb1 * kGbRegions) {
    const int p = t / kGbRegions, x = t % kGbRegions;
    q = ((p >> 3) << 6) | (x << 3) | (p & 7);
    have = cursor[t];
    lo = rstart[t];
    over = over || have > rcap[t];
  }
  const bool dead = __syncthreads_or(over ? 1 : 0) != 0;
  if (q >= 0) { sstart[q] = lo; send[q] = dead ? lo : lo + have; }
  if (dead && t == 0) atomicOr(redo, 4u);
  if (tile_rng) {
    unsigned long long m = 0, im = 0;
    for (int64_t i = t; i < ntiles; i += 1024) {
      const unsigned long long a = tile_rng[2 * i], b = tile_rng[2 * i + 1];
      m = a > m ? a : m;
      im = b > im ? b : im;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long a = __shfl_down(m, o, 64), b = __shfl_down(im, o, 64);
      m = a > m ? a : m;
      im = b > im ? b : im;
    }
    if ((t & 63) == 0) { s_max[t >> 6] = m; s_imin[t >> 6] = im; }
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < 16; w++) { m = s_max[w] > m ? s_max[w] : m; im = s_imin[w] > im ? s_imin[w] : im; }
      range[0] = m;
      range[1] = im;
      if (fx_wide(m, im)) atomicOr(redo, 2u);
    }
  }
}

// the sort-based path (gs_* kernels).  *used = 1: out_* hold the result.
static int gs_groupby(ah_ctx* c, int is_f64, const uint64_t* keys, const uint8_t* kvalid, int64_t koff, const void* vals, const uint8_t* vvalid,
                      int64_t voff, int64_t n, uint64_t* out_keys, void* out_sums, int64_t* out_counts, int64_t* out_first_rows, int64_t* out_ngroups,
                      int32_t* out_null_group, int* used) {
  *used = 0;
  if (n < ((int64_t)1 << 22) || n > ((int64_t)1 << 27)) return AH_OK;   // 2^27 rows = 2^21 buckets of 64 = 1024 parents × 2048
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  int lb = 0;
  while (((int64_t)64 << lb) < n) lb++;
  const int lb2 = lb >= 20 ? lb - 10 : lb - lb / 2;
  const int nb2 = 1 << lb2, nb1 = 1 << (lb - lb2);
  const int64_t nbuckets = (int64_t)1 << lb;
  const int64_t ntiles = ah_ceil_div(n, kMsTile), ngrp = ah_ceil_div(ntiles, kGroupTiles), nvt = ((ntiles + nb1 + 7) / 8) * 8;
  const unsigned grid1 = (unsigned)(((ntiles + 7) / 8) * 8);
  const int64_t nwords = ah_ceil_div(n, 64), nrt = rank_tiles(nwords);
  const size_t need = pad((size_t)n * 8) * 4 + pad((size_t)n * 4) * 3 + pad((size_t)ntiles * nb1 * 4) * 2 + pad((size_t)ngrp * nb1 * 4) + pad((size_t)(nb1 + 1) * 4) +
                      pad((size_t)nvt * nb2 * 4) * 2 + pad(((size_t)nbuckets + 1) * 4) + pad((size_t)ntiles * 16) + pad((size_t)nwords * 8) + pad((size_t)nwords * 4) +
                      pad((size_t)nrt * 4) + pad((size_t)nrt * 8);
  uint8_t* base;
  int rc = ah_temp_reserve(c, need, (void**)&base);
  if (rc != AH_OK) { c->err[0] = 0; return AH_OK; }   // these temporaries do not fit: *used stays 0 and the caller's id-based path (a fraction of them) answers
  size_t off = 0;
  auto take = [&](size_t b) { uint8_t* q = base + off; off += pad(b); return q; };
  unsigned long long* pkeys = (unsigned long long*)take((size_t)n * 8);
  unsigned long long* pvals = (unsigned long long*)take((size_t)n * 8);
  unsigned long long* qkeys = (unsigned long long*)take((size_t)n * 8);
  unsigned long long* qvals = (unsigned long long*)take((size_t)n * 8);
  unsigned* prows = (unsigned*)take((size_t)n * 4);
  unsigned* qrows = (unsigned*)take((size_t)n * 4);
  unsigned* g_cnt = (unsigned*)take((size_t)n * 4);
  unsigned* cnt1 = (unsigned*)take((size_t)ntiles * nb1 * 4);
  unsigned* toffs1 = (unsigned*)take((size_t)ntiles * nb1 * 4);
  unsigned* gsum = (unsigned*)take((size_t)ngrp * nb1 * 4);
  unsigned* pstart = (unsigned*)take((size_t)(nb1 + 1) * 4);
  unsigned* cnt2 = (unsigned*)take((size_t)nvt * nb2 * 4);
  unsigned* toffs2 = (unsigned*)take((size_t)nvt * nb2 * 4);
  unsigned* bstart = (unsigned*)take(((size_t)nbuckets + 1) * 4);
  unsigned long long* tile_max = (unsigned long long*)take((size_t)ntiles * 16);
  unsigned long long* firsts = (unsigned long long*)take((size_t)nwords * 8);
  unsigned* wordprefix = (unsigned*)take((size_t)nwords * 4);
  int* tilecnt = (int*)take((size_t)nrt * 4);
  int64_t* tileoff = (int64_t*)take((size_t)nrt * 8);
  unsigned long long* absmax = (unsigned long long*)&c->dscalars[28];   // [28], [29]: the value range (ah_hashing.h)
  unsigned* oversize = (unsigned*)&c->dscalars[21];
  unsigned long long* total = (unsigned long long*)&c->dscalars[22];
  int* null_id = (int*)&c->dscalars[23];
  AH_HIP(c, hipMemsetAsync(&c->dscalars[20], 0, 3 * sizeof(uint64_t), c->stream));
  AH_HIP(c, hipMemsetAsync(&c->dscalars[28], 0, 2 * sizeof(uint64_t), c->stream));   // value range
  AH_HIP(c, hipMemsetAsync(null_id, 0xFF, sizeof(uint64_t), c->stream));
  AH_HIP(c, hipMemsetAsync(firsts, 0, (size_t)nwords * 8, c->stream));
  GsColumns col{(const unsigned long long*)keys, kvalid, koff, (const unsigned long long*)vals, vvalid, voff};
  // level 1
  gs_hist_kernel<GsColumns><<<grid1, kThreads, 0, c->stream>>>(col, n, nullptr, 1, lb, lb2, (unsigned)(nb1 - 1), nb1, cnt1);
  AH_LAUNCH_CHECK(c);
  colsum_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt1, nb1, ntiles, gsum);
  AH_LAUNCH_CHECK(c);
  bin_prefix_kernel<<<1, kMaxBins, 0, c->stream>>>(gsum, nb1, ngrp, n, pstart);
  AH_LAUNCH_CHECK(c);
  tile_offs_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt1, gsum, nb1, ntiles, toffs1);
  AH_LAUNCH_CHECK(c);
  gs_scatter_kernel<GsColumns><<<grid1, kThreads, 0, c->stream>>>(col, n, nullptr, 1, lb, lb2, (unsigned)(nb1 - 1), nb1, toffs1, pkeys, pvals, prows,
                                                                  is_f64 ? tile_max : nullptr);
  AH_LAUNCH_CHECK(c);
  if (is_f64) {
    gb_max_kernel<<<1, 1024, 0, c->stream>>>(tile_max, ntiles, absmax);
    AH_LAUNCH_CHECK(c);
    fx_range_check_kernel<<<1, 1, 0, c->stream>>>(absmax, oversize);   // a wide column goes to the id-based path (per-group scales)
    AH_LAUNCH_CHECK(c);
  }
  // level 2, parent by parent
  GsRecords rec{pkeys, pvals, prows};
  gs_hist_kernel<GsRecords><<<(unsigned)nvt, kThreads, 0, c->stream>>>(rec, n, pstart, nb1, lb, 0, (unsigned)(nb2 - 1), nb2, cnt2);
  AH_LAUNCH_CHECK(c);
  ms_offs2_kernel<<<(unsigned)nb1, kThreads, 0, c->stream>>>(cnt2, pstart, nb1, nb2, toffs2, bstart, n);
  AH_LAUNCH_CHECK(c);
  gs_scatter_kernel<GsRecords, false, false><<<(unsigned)nvt, kThreads, 0, c->stream>>>(rec, n, pstart, nb1, lb, 0, (unsigned)(nb2 - 1), nb2, toffs2, qkeys, qvals, qrows, nullptr);
  AH_LAUNCH_CHECK(c);
  // buckets → group records at the position of each group's first row (p* are free again: they take the records)
  unsigned long long *g_key = pkeys, *g_sum = pvals;
  unsigned* g_first = prows;
  AH_HIP(c, hipMemsetAsync(g_first, 0xFF, (size_t)n * 4, c->stream));
  const unsigned lgrid = (unsigned)ah_ceil_div(nbuckets, 4);
  if (is_f64) gs_local_kernel<true><<<lgrid, 256, 0, c->stream>>>(qkeys, qvals, qrows, bstart, nbuckets, absmax, g_key, g_sum, g_cnt, g_first, firsts, oversize);
  else gs_local_kernel<false><<<lgrid, 256, 0, c->stream>>>(qkeys, qvals, qrows, bstart, nbuckets, absmax, g_key, g_sum, g_cnt, g_first, firsts, oversize);
  AH_LAUNCH_CHECK(c);
  // ids = rank of the first rows
  word_prefix_kernel<<<(unsigned)ah_ceil_div(nwords, kBlock), kBlock, 0, c->stream>>>(firsts, nwords, wordprefix, tilecnt);
  AH_LAUNCH_CHECK(c);
  scan_kernel<<<1, 1024, 0, c->stream>>>(tilecnt, nrt, tileoff, total);
  AH_LAUNCH_CHECK(c);
  const unsigned egrid = ah_stream_grid(c, ah_ceil_div(n, kBlock));
  if (is_f64) gs_emit_kernel<true><<<egrid, kBlock, 0, c->stream>>>(g_key, g_sum, g_cnt, g_first, n, firsts, wordprefix, tileoff, (unsigned long long*)out_keys,
                                                                  (unsigned long long*)out_sums, (long long*)out_counts, (long long*)out_first_rows, null_id);
  else gs_emit_kernel<false><<<egrid, kBlock, 0, c->stream>>>(g_key, g_sum, g_cnt, g_first, n, firsts, wordprefix, tileoff, (unsigned long long*)out_keys,
                                                            (unsigned long long*)out_sums, (long long*)out_counts, (long long*)out_first_rows, null_id);
  AH_LAUNCH_CHECK(c);
  { int mrc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[21], 3, (unsigned long long*)&c->pinned[8]); if (mrc != AH_OK) return mrc; }   // oversize, total, null id
  if (*(volatile unsigned*)&c->pinned[8]) return AH_OK;   // a bucket beyond one wave (a key with hundreds of rows): the id-based path redoes the call
  if (out_ngroups) *out_ngroups = (int64_t) * (volatile uint64_t*)&c->pinned[9];
  if (out_null_group) *out_null_group = *(volatile int32_t*)&c->pinned[10];
  *used = 1;
  return AH_OK;
}

// 2.1 M … 17 M expected groups: the LDS-table aggregation needs 2048 … 8192 partitions — cut in two levels (the gs_* scatter
// kernels: 64 × 64 partitions, runs as long as in the one-level cut of 64), then ONE workgroup per partition, tables dumped
// side by side.  *used = 1: out_* hold the result.
static int gb2_groupby(ah_ctx* c, int is_f64, int lp, const uint64_t* keys, const uint8_t* kvalid, int64_t koff, const void* vals, const uint8_t* vvalid,
                       int64_t voff, int64_t n, uint64_t* out_keys, void* out_sums, int64_t* out_counts, int64_t* out_first_rows, int64_t* out_ngroups,
                       int32_t* out_null_group, int* used, bool reserve2 = true, const unsigned* hist = nullptr, double hist_scale = 0.0) {
  // reserve2: the second level's scatter reserves its runs in fixed regions, one per final partition (gs_scatter_kernel, RES) — no
  // histogram of the first level's output, no offsets table (2^26 rows, 8192 partitions: 254 + 98 µs).  Null keys all go to child 0 of
  // parent 0 and would overflow its region: such columns keep the histogram.
  if (kvalid || c->opt_groupby_reserve == 0) reserve2 = false;
  // hist (the 2^21-row sample's [8][1024] row counts, when the caller ran it): the first level is the one-level cut's own reserving
  // scatter (gb_scatter_kernel<…, RESERVE>: a region per parent and XCD, sized from the sample; 0.48 ms where the two-level machinery's
  // level-1 kernel takes 0.64–0.72) and the second level tiles those regions
  const bool xreg = reserve2 && hist != nullptr;
  *used = 0;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const int lb2 = lp - lp / 2;
  const int nb2 = 1 << lb2, nb1 = 1 << (lp - lb2);
  const int64_t P = (int64_t)1 << lp;
  const int64_t ntiles = ah_ceil_div(n, kMsTile), ngrp = ah_ceil_div(ntiles, kGroupTiles), nvt = ((ntiles + nb1 + 7) / 8) * 8;
  const unsigned grid1 = (unsigned)(((ntiles + 7) / 8) * 8);
  // the finish (GbRec): fine bins of 4096 rows, 2^fpc_log2 of them per coarse bin
  const int64_t nfine_used = ah_ceil_div(n, (int64_t)1 << kRecFineLog2);
  int fbits = 0;
  while (((int64_t)1 << fbits) < nfine_used) fbits++;
  const int fpc_log2 = fbits - fbits / 2, cshift = kRecFineLog2 + fpc_log2;
  const int ncoarse = (int)ah_ceil_div(n, (int64_t)1 << cshift);
  if (ncoarse > kRecMaxBins || (1 << fpc_log2) > kRecMaxBins) return AH_OK;   // (beyond 2^30 rows: not this path's)
  const int64_t nfine = (int64_t)ncoarse << fpc_log2;
  const size_t nrec = (size_t)ncoarse << cshift;                               // record places: every coarse bin whole (≥ n)
  // the records of the coarse scatter lie over the first level's rows (free once the second level is cut), those of the fine scatter
  // over the second level's (free once the aggregate pass has read them): 20 of the 32 bytes per place each
  // first level's record arrays: dense (n rows) behind the offsets table; nb1 regions of cap1 rows (1.125 × the even share + a tile) + kMsTile spare rows otherwise
  const unsigned cap1 = (unsigned)((((n / nb1) * 9 / 8 + kMsTile) + 15) & ~(int64_t)15);
  const int64_t xcap_rows = n + n / 2 + (int64_t)nb1 * kGbRegions * 96;   // gb_layout_kernel's regions (as in gb_cut_aggregate)
  const int64_t prows_n = xreg ? xcap_rows + kGbTile : (reserve2 ? nb1 * (int64_t)cap1 + kMsTile : n);
  const size_t plevel = pad((size_t)prows_n * 8) * 2 + pad((size_t)prows_n * 4);
  const size_t level = plevel > pad(nrec * sizeof(GbRec)) ? plevel : pad(nrec * sizeof(GbRec));   // … and the coarse records lie over them
  const size_t extra = 0;
  // second level's record arrays: dense (n rows) behind the offsets table; P regions of cap2 rows (1.25 × the even share) + kMsTile spare rows otherwise
  const unsigned cap2 = (unsigned)((((n / P) * 5 / 4 + 64) + 15) & ~(int64_t)15);
  const int64_t qrows_n = reserve2 ? P * (int64_t)cap2 + kMsTile : n;
  const size_t qlevel = pad((size_t)qrows_n * 8) * 2 + pad((size_t)qrows_n * 4);
  const size_t qblock = qlevel > level + extra ? qlevel : level + extra;
  const size_t need = (level + extra) + qblock + pad((size_t)ntiles * nb1 * 4) * 2 + pad((size_t)ngrp * nb1 * 4) + pad((size_t)(nb1 + 1) * 4) +
                      pad((size_t)nvt * nb2 * 4) * 2 + pad(((size_t)P + 1) * 4) + pad((size_t)ntiles * 16) + pad((size_t)kRecMaxBins * 4) + pad((size_t)nfine * 4) +
                      pad((size_t)nfine * 8) + pad(((size_t)P + 1) * 4) + pad((size_t)(nb1 * kGbRegions + 1) * 4) * 5;
  uint8_t* base;
  int rc = ah_temp_reserve(c, need, (void**)&base);
  if (rc != AH_OK) { c->err[0] = 0; return AH_OK; }   // these temporaries do not fit: *used stays 0 and the caller's id-based path (a fraction of them) answers
  size_t off = 0;
  auto take = [&](size_t b) { uint8_t* q = base + off; off += pad(b); return q; };
  uint8_t* pbase = take(level);
  unsigned long long* pkeys = (unsigned long long*)pbase;
  unsigned long long* pvals = (unsigned long long*)(pbase + pad((size_t)prows_n * 8));
  unsigned* prows = (unsigned*)(pbase + pad((size_t)prows_n * 8) * 2);
  uint8_t* qbase = take(qblock);
  unsigned long long* qkeys = (unsigned long long*)qbase;
  unsigned long long* qvals = (unsigned long long*)(qbase + pad((size_t)qrows_n * 8));
  unsigned* qrows = (unsigned*)(qbase + pad((size_t)qrows_n * 8) * 2);
  GbRec* recs1 = (GbRec*)pkeys;
  GbRec* recs2 = (GbRec*)qkeys;
  unsigned* cnt1 = (unsigned*)take((size_t)ntiles * nb1 * 4);
  unsigned* toffs1 = (unsigned*)take((size_t)ntiles * nb1 * 4);
  unsigned* gsum = (unsigned*)take((size_t)ngrp * nb1 * 4);
  unsigned* pstart = (unsigned*)take((size_t)(nb1 + 1) * 4);
  unsigned* cnt2 = (unsigned*)take((size_t)nvt * nb2 * 4);
  unsigned* toffs2 = (unsigned*)take((size_t)nvt * nb2 * 4);
  unsigned* bstart = (unsigned*)take(((size_t)P + 1) * 4);
  unsigned long long* tile_max = (unsigned long long*)take((size_t)ntiles * 16);
  unsigned* ccursor = (unsigned*)take((size_t)kRecMaxBins * 4);
  unsigned* fcursor = (unsigned*)take((size_t)nfine * 4);
  int64_t* fprefix = (int64_t*)take((size_t)nfine * 8);
  unsigned* cursor2 = (unsigned*)take(((size_t)P + 1) * 4);
  const int nsub = nb1 * kGbRegions;   // ≤ 512
  unsigned* cursor1 = (unsigned*)take((size_t)(nsub + 1) * 4);
  unsigned* pend = (unsigned*)take((size_t)(nsub + 1) * 4);
  unsigned* sstart = (unsigned*)take((size_t)(nsub + 1) * 4);
  unsigned* rstart1 = (unsigned*)take((size_t)(nsub + 1) * 4);
  unsigned* rcap1 = (unsigned*)take((size_t)(nsub + 1) * 4);
  GbTable gt{nullptr, nullptr, nullptr, nullptr, nullptr};   // no tables: the groups leave the aggregate pass as records
  unsigned long long* absmax = (unsigned long long*)&c->dscalars[28];   // [28], [29]: the value range (ah_hashing.h)
  unsigned* overflow = (unsigned*)&c->dscalars[21];
  unsigned long long* total = (unsigned long long*)&c->dscalars[22];
  int* null_id = (int*)&c->dscalars[23];
  {
    GbFill f;   // one launch: the call's scalars and both cursor arrays (adjacent)
    f.njobs = 5;
    f.p[0] = (uint4*)&c->dscalars[20]; f.n16[0] = 1; f.v[0] = 0u;                 // [20] unused, [21] overflow
    f.p[1] = (uint4*)&c->dscalars[28]; f.n16[1] = 1; f.v[1] = 0u;                 // [28], [29] value range
    f.p[2] = (uint4*)ccursor; f.n16[2] = (pad((size_t)kRecMaxBins * 4) + pad((size_t)nfine * 4)) / 16; f.v[2] = 0u;
    f.p[3] = (uint4*)cursor2; f.n16[3] = (pad(((size_t)P + 1) * 4) + pad((size_t)(nsub + 1) * 4)) / 16; f.v[3] = 0u;   // the children's and the parents' cursors (adjacent)
    f.p[4] = (uint4*)&c->dscalars[22]; f.n16[4] = 1; f.v[4] = 0u;                 // [22] total, [23] …
    f.ones = (unsigned long long*)null_id;                                         // … null id: none (the last job's first word: the same thread)
    gb_fill_kernel<<<64, 256, 0, c->stream>>>(f);
    AH_LAUNCH_CHECK(c);
  }
  GsColumns col{(const unsigned long long*)keys, kvalid, koff, (const unsigned long long*)vals, vvalid, voff};
  const unsigned* l2_start = pstart;
  int l2_parents = nb1, l2_sub8 = 0;
  if (xreg) {
    const int lb1 = lp - lb2;
    const int64_t gtiles = ah_ceil_div(n, kGbTile), xrows = ((gtiles + 7) >> 3) * kGbTile;
    gb_layout_kernel<<<1, 1024, 0, c->stream>>>(hist, lb1, n, xrows, hist_scale, xcap_rows, rstart1, rcap1, cursor1, overflow);
    AH_LAUNCH_CHECK(c);
    gb_scatter_kernel<true, true><<<(unsigned)(((gtiles + 7) / 8) * 8), kThreads, 0, c->stream>>>((const unsigned long long*)keys, kvalid, koff, (const unsigned long long*)vals, vvalid, voff,
                                                                                               n, lb1, nb1, gtiles, nullptr, pkeys, pvals, prows, is_f64 ? tile_max : nullptr,
                                                                                               rstart1, rcap1, cursor1, overflow, (unsigned)xcap_rows);
    AH_LAUNCH_CHECK(c);
    gs_subregions_kernel<<<1, 1024, 0, c->stream>>>(rstart1, rcap1, cursor1, nb1, sstart, pend, overflow, is_f64 ? tile_max : nullptr, gtiles, absmax);
    AH_LAUNCH_CHECK(c);
    l2_start = sstart; l2_parents = nsub; l2_sub8 = 1;
  } else if (reserve2) {
    // level 1 reserves too: parent d owns the region [d · cap1, …); its runs come from all eight XCDs, 64 records (512 bytes of keys) at a
    // time — long enough not to share many lines (one cursor per partition only loses against the offsets table from 512 partitions on)
    gs_scatter_kernel<GsColumns, true><<<grid1, kThreads, 0, c->stream>>>(col, n, nullptr, 1, lp, lb2, (unsigned)(nb1 - 1), nb1, nullptr, pkeys, pvals, prows,
                                                                        is_f64 ? tile_max : nullptr, cursor1, cap1, (unsigned)(nb1 * (int64_t)cap1), overflow);
    AH_LAUNCH_CHECK(c);
    if (is_f64) {
      gb_max_kernel<<<1, 1024, 0, c->stream>>>(tile_max, ntiles, absmax);
      AH_LAUNCH_CHECK(c);
    }
    gs_regions_kernel<<<1, 64, 0, c->stream>>>(cursor1, nb1, cap1, pstart, pend, overflow, is_f64 ? absmax : nullptr);   // nb1 ≤ 64
    AH_LAUNCH_CHECK(c);
  } else {
  gs_hist_kernel<GsColumns><<<grid1, kThreads, 0, c->stream>>>(col, n, nullptr, 1, lp, lb2, (unsigned)(nb1 - 1), nb1, cnt1);
  AH_LAUNCH_CHECK(c);
  colsum_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt1, nb1, ntiles, gsum);
  AH_LAUNCH_CHECK(c);
  bin_prefix_kernel<<<1, kMaxBins, 0, c->stream>>>(gsum, nb1, ngrp, n, pstart);
  AH_LAUNCH_CHECK(c);
  tile_offs_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt1, gsum, nb1, ntiles, toffs1);
  AH_LAUNCH_CHECK(c);
  gs_scatter_kernel<GsColumns><<<grid1, kThreads, 0, c->stream>>>(col, n, nullptr, 1, lp, lb2, (unsigned)(nb1 - 1), nb1, toffs1, pkeys, pvals, prows,
                                                                  is_f64 ? tile_max : nullptr);
  AH_LAUNCH_CHECK(c);
  if (is_f64) {
    gb_max_kernel<<<1, 1024, 0, c->stream>>>(tile_max, ntiles, absmax);
    AH_LAUNCH_CHECK(c);
    fx_range_check_kernel<<<1, 1, 0, c->stream>>>(absmax, overflow);   // a wide column goes to the id-based path (per-group scales)
    AH_LAUNCH_CHECK(c);
  }
  }
  GsRecords rec{pkeys, pvals, prows};
  if (reserve2) {
    const unsigned nvt2 = xreg ? (unsigned)(((ntiles + nsub + 7) / 8) * 8) : (unsigned)nvt;   // every region's last tile may be short
    gs_scatter_kernel<GsRecords, true, false><<<nvt2, kThreads, 0, c->stream>>>(rec, n, l2_start, l2_parents, lp, 0, (unsigned)(nb2 - 1), nb2, nullptr, qkeys, qvals, qrows, nullptr,
                                                                       cursor2, cap2, (unsigned)(P * (int64_t)cap2), overflow, pend, l2_sub8);
    AH_LAUNCH_CHECK(c);
  } else {
    gs_hist_kernel<GsRecords><<<(unsigned)nvt, kThreads, 0, c->stream>>>(rec, n, pstart, nb1, lp, 0, (unsigned)(nb2 - 1), nb2, cnt2);
    AH_LAUNCH_CHECK(c);
    ms_offs2_kernel<<<(unsigned)nb1, kThreads, 0, c->stream>>>(cnt2, pstart, nb1, nb2, toffs2, bstart, n);
    AH_LAUNCH_CHECK(c);
    gs_scatter_kernel<GsRecords, false, false><<<(unsigned)nvt, kThreads, 0, c->stream>>>(rec, n, pstart, nb1, lp, 0, (unsigned)(nb2 - 1), nb2, toffs2, qkeys, qvals, qrows, nullptr);
    AH_LAUNCH_CHECK(c);
  }
  const unsigned* part_rows = reserve2 ? cursor2 : bstart;   // flat aggregate: counts of fixed regions, or the dense starts
  const int64_t part_cap = reserve2 ? (int64_t)cap2 : 0;
  // aggregate: one workgroup per partition; its groups leave as records in the coarse bin of their first row (over p*: free now)
  const GbRecOut ro{recs1, ccursor, cshift, ncoarse};
  const GbStaging nost{nullptr, nullptr, nullptr, nullptr};
  if (is_f64) gb_aggregate_kernel<true, false, false, false, true><<<(unsigned)P, kThreads, 0, c->stream>>>(qkeys, qvals, qrows, part_rows, 0, gt, absmax, overflow, 1, nullptr, 0, nullptr, 0, 0, part_cap,
                                                                                                           nullptr, nullptr, 0, nost, nullptr, nullptr, ro);
  else gb_aggregate_kernel<false, false, false, false, true><<<(unsigned)P, kThreads, 0, c->stream>>>(qkeys, qvals, qrows, part_rows, 0, gt, absmax, overflow, 1, nullptr, 0, nullptr, 0, 0, part_cap,
                                                                                                     nullptr, nullptr, 0, nost, nullptr, nullptr, ro);
  AH_LAUNCH_CHECK(c);
  // coarse → fine bins (over q*: the aggregate pass has read them), ids of each fine bin's first group, the output
  const int tiles_per_coarse = (int)(((int64_t)1 << cshift) / kRecTile);
  gbr_split_kernel<<<(unsigned)(ncoarse * tiles_per_coarse), kThreads, 0, c->stream>>>(recs1, ccursor, cshift, fpc_log2, tiles_per_coarse, recs2, fcursor);
  AH_LAUNCH_CHECK(c);
  scan_kernel<<<1, 1024, 0, c->stream>>>((const int*)fcursor, nfine, fprefix, total);
  AH_LAUNCH_CHECK(c);
  if (is_f64) gbr_emit_kernel<true><<<(unsigned)nfine, kRecEmitThreads, 0, c->stream>>>(recs2, fcursor, fprefix, absmax, (unsigned long long*)out_keys, (unsigned long long*)out_sums,
                                                                         (long long*)out_counts, (long long*)out_first_rows, null_id);
  else gbr_emit_kernel<false><<<(unsigned)nfine, kRecEmitThreads, 0, c->stream>>>(recs2, fcursor, fprefix, absmax, (unsigned long long*)out_keys, (unsigned long long*)out_sums,
                                                                   (long long*)out_counts, (long long*)out_first_rows, null_id);
  AH_LAUNCH_CHECK(c);
  { int mrc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[21], 3, (unsigned long long*)&c->pinned[8]); if (mrc != AH_OK) return mrc; }   // overflow, total, null id
  if (reserve2 && (*(volatile unsigned*)&c->pinned[8] & 4u) && !(*(volatile unsigned*)&c->pinned[8] & 3u))   // a child's region was too small: once more behind a histogram
    return gb2_groupby(c, is_f64, lp, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows, out_ngroups, out_null_group, used, false);
  if (*(volatile unsigned*)&c->pinned[8]) return AH_OK;   // a partition outgrew its LDS table: the id-based path redoes the call
  if (out_ngroups) *out_ngroups = (int64_t) * (volatile uint64_t*)&c->pinned[9];
  if (out_null_group) *out_null_group = *(volatile int32_t*)&c->pinned[10];
  *used = 1;
  return AH_OK;
}

// ≤ 2048 expected groups: no cut — every workgroup aggregates 2^18 consecutive rows of the columns in its LDS table and merges it
// into one global table (the id-based path spends a pass on row-order ids it does not need, and a second one reading them back).
// ---- the direct path's scale: a guess from a sample, checked against the truth the aggregate pass sees ------------------------------
// The fixed-point scale needs the largest |value| BEFORE the first addend is converted — a whole pass over the values (absmax_kernel:
// 0.10–0.12 ms of the direct path's 0.70 at 2^26 rows).  Here: the largest |x| of 2^18 sampled values (4096 groups of 64 consecutive
// rows, evenly spread) + kGuessMargin binades is used as the scale, the aggregate pass — which reads every value anyway — tracks the
// true exponent range, and the call is handed to the id-based path (as a wide column is) when the truth does not fit the guess:
//   · a value more than 2 binades above the guess could overflow the 128-bit accumulator (2^30 rows × 2^(95 + excess));
//   · guess − smallest exponent > 42 would truncate an addend.
// Within those bounds every addend is represented exactly, so the sums are the correctly rounded exact sums whatever the scale was.
constexpr int kGuessMargin = 4;
constexpr int kGuessGroups = 4096;
__global__ __launch_bounds__(256) void fx_sample_max_kernel(const unsigned long long* __restrict__ vals, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                            int64_t n, int64_t groups, int64_t stride, unsigned long long* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  unsigned long long m = 0;
  if (g < groups) {
    const int64_t i = g * stride + (threadIdx.x & 63);
    if (i < n) {
      const unsigned long long b = vals[i] & 0x7fffffffffffffffull;
      if ((b >> 52) != 0x7ff && ah_bit(vvalid, voff + i)) m = b;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long x = __shfl_down(m, o, 64);
    m = x > m ? x : m;
  }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
__global__ void fx_guess_kernel(unsigned long long* __restrict__ range) {
  int e = (int)((range[0] >> 52) & 0x7ff);
  e = (e ? e : 1) + kGuessMargin;
  range[0] = (unsigned long long)(e > 0x7fe ? 0x7fe : e) << 52;
  range[1] = 0;
}
__global__ __launch_bounds__(256) void fx_guess_check_kernel(const unsigned* __restrict__ tile_range, int nblocks, unsigned long long* __restrict__ range,
                                                             unsigned* __restrict__ flag) {
  __shared__ unsigned s_e[256 / 64], s_i[256 / 64];
  unsigned emax = 0, imin = 0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    const unsigned a = tile_range[2 * b], c = tile_range[2 * b + 1];
    emax = a > emax ? a : emax;
    imin = c > imin ? c : imin;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned a = __shfl_down(emax, o, 64), c = __shfl_down(imin, o, 64);
    emax = a > emax ? a : emax;
    imin = c > imin ? c : imin;
  }
  if ((threadIdx.x & 63) == 0) { s_e[threadIdx.x >> 6] = emax; s_i[threadIdx.x >> 6] = imin; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 256 / 64; w++) { emax = s_e[w] > emax ? s_e[w] : emax; imin = s_i[w] > imin ? s_i[w] : imin; }
    const int eg = (int)((range[0] >> 52) & 0x7ff);
    if ((int)emax > eg + 2) atomicOr(flag, 2u);                            // the guess was too low: the accumulators may have wrapped
    if (imin && eg - (0x7ff - (int)imin) > 42) atomicOr(flag, 2u);         // too wide for one scale (as fx_range_check_kernel says of the true range)
    range[1] = imin;
  }
}

// ---- the direct path's prologue and epilogue: one kernel each ---------------------------------------------------------------------
// Around a 0.42 ms aggregate pass the direct path used to launch sixteen small kernels (two sample passes + two popcounts + a post,
// seven memsets, sample max, guess, guess check, mark, word prefix, scan, emit, a post): ≈ 0.16 ms of launches and single-workgroup
// passes for a table of ≤ 2048 groups.  Now: a QUICK LOOK (one workgroup counts the distinct keys of 2^16 spread rows in an LDS
// table and takes the largest sampled |value|; it posts both to the host itself), one kernel that initialises everything, the
// aggregate pass, and one kernel that checks the scale guess, ranks the ≤ 8200 table entries by first row (by counting, in LDS),
// writes the groups out and posts {flags, group count, null group}.
constexpr int kQlGroups = 2048;                      // × 8 consecutive rows = 2^14 sampled rows.  ONE workgroup does the look, i.e. one CU executes all of it:
constexpr int kQlRun = 8;                            // ≈ 2000 instructions per lane for 16 rows = 35 µs (2^16 rows: 56 µs; 2^13 rows: 26 µs, but the aggregate pass
                                                     // then ran 8 % slower at 2^10 / 2^11 groups; 512 runs of 32 rows: 45 µs — it is not address translation).
                                                     // A column SORTED by key shows ≥ 2048 distinct keys this way (beyond the 1800 the direct path is taken for)
                                                     // unless it really has few; the neighbour statistic below catches clustered columns in general
// first row of sample group g: evenly spread, but JITTERED inside its stride — a column that repeats with a period the stride
// divides (a table built by tiling one block, as benchmarks do) showed an equidistant sample the same few rows over and over:
// 65 536 groups looked like 1024 and took the direct path (and its full global table: 0.9 s)
__device__ __forceinline__ int64_t gq_row(int g, int64_t stride) {
  // (multiply-high instead of a remainder: a 64-bit `%` is a hundred instructions, and thirty-two of them per lane were two thirds of
  // the look's 35 µs)
  const unsigned room = stride > kQlRun ? (unsigned)(stride - kQlRun) : 0u;   // strides are below 2^18 (n < 2^29 rows / 2048 groups)
  const unsigned jit = (unsigned)(((unsigned long long)((unsigned)g * 2654435761u) * (unsigned long long)(room + 1u)) >> 32);
  return (int64_t)g * stride + (int64_t)(jit & ~(unsigned)(kQlRun - 1));
}
// gb_lds_slot on a table in GLOBAL memory shared by several workgroups (coherent loads, device-scope CAS): same probe order, same
// "first slot that holds the key or is empty" rule — so the table it leaves is one gb_lds_slot could have built
__device__ __forceinline__ int gq_global_slot(unsigned long long* __restrict__ g_key, unsigned* __restrict__ g_tickets, unsigned long long key) {
  unsigned g = ((((unsigned)key * 0x9E3779B1u) ^ ((unsigned)(key >> 32) * 0x85EBCA6Bu)) >> 20) & (unsigned)(kSlots - 4);
  // A BOUNDED walk: the admission count below is read before the CAS, so lanes that read it together can take the table past
  // kSoftLimit — to completely full, if enough of them are in flight — and an unbounded probe for a key that is not in a full
  // table never ends.  (Sixteen thousand lanes never got there in hundreds of calls; a variant of this kernel with half a million
  // lanes in flight did on its first run and hung the device until the watchdog.)  −1 = "not seeded", which every caller handles.
  // The admission count travels WITH every probe round's four key loads (one wait for five loads): once the table has admitted its
  // kSoftLimit keys the column has "many" groups and nothing else is wanted from the look — a lane leaves at once instead of walking
  // a nearly full table for a key that is not in it (at 2^16 groups fourteen thousand lanes did, thirty rounds of cross-XCD
  // round trips each: the look took 138 µs there, profiles/r04_bench_kernel_stats.csv).  A lane whose key IS seeded may leave
  // without finding it: "not seeded" is a valid answer for every caller, and the host reads the full table as "many".
  for (int probes = 0; probes < kSlots / 4 + 8; probes++) {
    unsigned long long q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = __hip_atomic_load(&g_key[g + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_load(g_tickets, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u >= (unsigned)kSoftLimit) return -1;   // (the word holds count − 1: see qs below)
    int j = -1;
    bool hit = false;
#pragma unroll
    for (int k = 3; k >= 0; k--) {
      if (q[k] == key) { j = (int)g + k; hit = true; }
      else if (q[k] == kEmpty) { j = (int)g + k; hit = false; }
    }
    if (j < 0) { g = (g + 4) & (kSlots - 1); continue; }
    if (hit) return j;
    // (counted on SUCCESS, unlike gb_lds_slot's tickets: sixteen thousand lanes meet an empty table at once here — and counted once
    // per WAVE and round: three and a half thousand single increments of one word are 43 µs of same-address atomics, 12 ns each)
    const unsigned long long cur = atomicCAS(&g_key[j], kEmpty, key);
    const unsigned long long won = __ballot(cur == kEmpty);   // among the lanes that are in this round
    if (won && (int)(threadIdx.x & 63) == __builtin_ctzll(won)) atomicAdd(g_tickets, (unsigned)__popcll(won));
    if (cur == kEmpty || cur == key) return j;
  }
  return -1;
}

// qs: the look's words in device memory, all preset to ones by ONE memset together with the seed table in front of them, so every
// counter reads "stored + 1": [0] tickets, [1] special bits (stored inverted: AND clears), [2] repeats, [3] pairs, [4] neighbours,
// [5] workgroups done, [6..7] ~(largest |value|) as a 64-bit minimum, [8] "many": cleared by a workgroup whose own 1024 rows say so.
constexpr int kQlBlocks = kQlGroups * kQlRun / 1024;   // one row per lane: the look's latency is one row's, not sixteen rows' on one CU (35–50 µs)
// A workgroup first counts the distinct keys of ITS 1024 rows in an LDS table.  More than kQlLocalMany of them: the column has far
// more groups than the direct path takes, whatever the other workgroups see — 1024 rows drawn from G keys show G·(1 − e^(−1024/G))
// distinct ones at most (evenly drawn keys show the most): 866 ± 10 for G = 3000, the direct path's limit, 920 from G ≈ 4300 on.
// Such a workgroup clears word [8] and stays away from the shared table — on a column of 2^16 groups all sixteen do, and the look
// is one launch with no cross-XCD traffic at all instead of fourteen thousand lanes filling a 4096-slot table by CAS, four rounds
// per group of slots (51 µs; 138 µs before the admission count travelled with every probe round).  Otherwise only the lane that
// brought a key into the workgroup's table goes on to the shared one: a column of 16 groups sends 16 keys per workgroup, not 900.
constexpr int kQlLocalSlots = 2048, kQlLocalMany = 920;
__global__ __launch_bounds__(1024) void gq_quicklook_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ kvalid, int64_t koff,
                                                             const unsigned long long* __restrict__ vals, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                             int64_t n, int64_t stride, unsigned long long* __restrict__ seed_keys, unsigned* __restrict__ qs,
                                                             unsigned long long* mb, unsigned long long seq) {
  __shared__ unsigned s_last, s_held, s_local;
  __shared__ unsigned long long s_tab[kQlLocalSlots];
  const int t = threadIdx.x;
  s_tab[t] = kEmpty;
  s_tab[t + 1024] = kEmpty;
  if (t == 0) s_local = 0;
  __syncthreads();
  constexpr int kGpi = 1024 / kQlRun;             // groups per workgroup
  const int64_t i = gq_row((int)blockIdx.x * kGpi + t / kQlRun, stride) + (t % kQlRun);
  const bool in = i < n;
  const unsigned long long key = in ? keys[i] : 0ull;
  const unsigned long long k2 = i + kThreads < n ? keys[i + kThreads] : ~key;   // the row the aggregate pass's lane takes next
  const unsigned long long vv = in && vals ? vals[i] : 0ull;
  const bool kv = in && (!kvalid || ((kvalid[(koff + i) >> 3] >> ((koff + i) & 7)) & 1));
  const bool vok = in && vals && (!vvalid || ((vvalid[(voff + i) >> 3] >> ((voff + i) & 7)) & 1));
  const unsigned long long kprev = __shfl_up(key, 1, 64);   // the row in front (same sample group unless t starts one)
  const bool kprev_ok = __shfl_up((int)kv, 1, 64) != 0;
  unsigned special = 0;
  if (in && !kv) special = 1u;                       // the null group
  if (kv && key == kEmpty) special |= 2u;            // the all-ones key has a slot of its own
  // a lane whose key its left neighbour or the wave's first lane also holds leaves the insert to that lane: a column with one
  // dominant key would otherwise send sixteen thousand CAS to one address (53 µs)
  const unsigned long long key0 = __shfl(key, 0, 64);
  const bool kv0 = __shfl((int)kv, 0, 64) != 0;
  const bool dup = ((t & 63) != 0 && kv0 && key == key0) || ((t & 63) != 0 && kprev_ok && key == kprev);
  bool mine = kv && key != kEmpty && !dup;
  if (mine) {   // the workgroup's own table: the first lane to bring a key keeps it
    unsigned h = (((unsigned)key * 0x9E3779B1u) ^ ((unsigned)(key >> 32) * 0x85EBCA6Bu)) >> (32 - 11);
    for (;;) {
      const unsigned long long cur = atomicCAS(&s_tab[h], kEmpty, key);
      if (cur == kEmpty) { atomicAdd(&s_local, 1u); break; }
      if (cur == key) { mine = false; break; }
      h = (h + 1) & (kQlLocalSlots - 1);
    }
  }
  __syncthreads();
  const bool local_many = s_local > (unsigned)kQlLocalMany;   // (uniform)
  if (local_many) { if (t == 0) atomicAnd(&qs[8], 0u); }
  else if (mine) (void)gq_global_slot(seed_keys, &qs[0], key);
  const unsigned long long rep_m = __ballot(kv && key == k2), pair_m = __ballot(kv), adj_m = __ballot(kv && kprev_ok && (t % kQlRun) != 0 && key == kprev);
  unsigned long long m = 0;
  if (vok) { const unsigned long long b = vv & 0x7fffffffffffffffull; if ((b >> 52) != 0x7ff) m = b; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long x = __shfl_down(m, o, 64);
    m = x > m ? x : m;
    special |= __shfl_down(special, o, 64);
  }
  if ((t & 63) == 0) {   // one update per wave and word
    if (rep_m) atomicAdd(&qs[2], (unsigned)__popcll(rep_m));
    if (pair_m) atomicAdd(&qs[3], (unsigned)__popcll(pair_m));
    if (adj_m) atomicAdd(&qs[4], (unsigned)__popcll(adj_m));
    if (special) atomicAnd(&qs[1], ~special);
    if (m) atomicMin((unsigned long long*)&qs[6], ~m);
  }
  __syncthreads();
  if (t == 0) {
    __threadfence();
    s_last = atomicAdd(&qs[5], 1u) + 1u == (unsigned)kQlBlocks - 1u ? 1u : 0u;   // (stored + 1 = done before this one)
    s_held = 0;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // the last workgroup counts the keys the table holds (tickets over-count: lanes meeting one new key at once each take one)
  unsigned held = 0;
  for (int j = t; j < kSlots; j += 1024) held += __hip_atomic_load(&seed_keys[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kEmpty ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) held += __shfl_down(held, o, 64);
  if ((t & 63) == 0) atomicAdd(&s_held, held);
  __syncthreads();
  if (t == 0) {
    auto rd = [&](int k) { return __hip_atomic_load(&qs[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    const unsigned sp = ~rd(1) & 3u;
    const unsigned used = rd(8) == 0u ? (s_held > (unsigned)kSoftLimit ? s_held : (unsigned)kSoftLimit) : s_held;   // ≥ kSoftLimit − a few: "many" (inserts stop there; a workgroup said so)
    const unsigned long long mx = ~__hip_atomic_load((unsigned long long*)&qs[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // {distinct keys among the sampled rows, largest sampled |value|, pairs | rows sharing the key of the row 1024 further on,
    //  keys in the seed table, neighbouring rows sharing a key}
    const unsigned long long w[5] = {(unsigned long long)used + (sp & 1u) + ((sp >> 1) & 1u), mx,
                                     ((unsigned long long)(rd(3) + 1u) << 32) | (rd(2) + 1u), (unsigned long long)used, (unsigned long long)(rd(4) + 1u)};
    ah_mailbox_post(mb, seq, w, 5);
  }
}

// the seeded slots' shares of all workgroups, added up slot by slot (no atomics, no probing: a seeded key has ONE slot everywhere).
// A block takes 64 slots; its 16 waves each add up a sixteenth of the workgroups (one thread per slot walking all 256 workgroups
// was a chain of 256 dependent latencies: 100 µs), then the 16 partial sums meet in LDS.
constexpr int kRedSlots = 64;
template <bool FX>
__global__ __launch_bounds__(1024) void gd_reduce_kernel(GbStaging st, int nwg, const unsigned long long* __restrict__ seed_keys, GbTable rt) {
  __shared__ unsigned long long s_lo[16][kRedSlots], s_hi[16][kRedSlots];
  __shared__ unsigned s_cnt[16][kRedSlots], s_flags[16][kRedSlots], s_first[16][kRedSlots];
  const int sx = threadIdx.x & (kRedSlots - 1), wy = threadIdx.x >> 6;
  const int j = (int)blockIdx.x * kRedSlots + sx;
  const bool seeded = j < kLSlots && (j >= kSlots || seed_keys[j] != kEmpty);
  unsigned long long lo = 0, hi = 0;
  unsigned cnt = 0, flags = 0, first = kNoRow;
  if (seeded) {
    constexpr int kW = 4;
    for (int w0 = wy; w0 < nwg; w0 += 16 * kW) {
      unsigned long long l[kW], h[kW];
      unsigned c[kW], f[kW];
#pragma unroll
      for (int u = 0; u < kW; u++) {
        const int w = w0 + 16 * u;
        const bool in = w < nwg;
        const size_t o = (size_t)w * kLSlots + (size_t)j;
        l[u] = in ? st.lo[o] : 0ull;
        h[u] = in && FX ? st.hi[o] : 0ull;
        c[u] = in ? st.cnt[o] : 0u;
        f[u] = in ? st.first[o] : kNoRow;
      }
#pragma unroll
      for (int u = 0; u < kW; u++) {
        const unsigned long long nl = lo + l[u];
        hi += h[u] + (nl < lo ? 1ull : 0ull);
        lo = nl;
        cnt += c[u] & kCntMask;
        flags |= c[u] & ~kCntMask;
        first = f[u] < first ? f[u] : first;
      }
    }
  }
  s_lo[wy][sx] = lo; s_hi[wy][sx] = hi; s_cnt[wy][sx] = cnt; s_flags[wy][sx] = flags; s_first[wy][sx] = first;
  __syncthreads();
  if (wy == 0 && j < kLSlots) {
    for (int w = 1; w < 16; w++) {
      const unsigned long long nl = lo + s_lo[w][sx];
      hi += s_hi[w][sx] + (nl < lo ? 1ull : 0ull);
      lo = nl;
      cnt += s_cnt[w][sx];
      flags |= s_flags[w][sx];
      first = s_first[w][sx] < first ? s_first[w][sx] : first;
    }
    rt.key[j] = j < kSlots ? seed_keys[j] : (j == kSlots ? kEmpty : 0ull);
    rt.lo[j] = lo;
    if (FX) rt.hi[j] = hi;
    rt.cnt[j] = cnt | flags;
    rt.first[j] = first;
  }
}

// everything the direct path starts from: the global table empty, the per-workgroup exponent ranges zero, the call's scalars
__global__ __launch_bounds__(256) void gd_prep_kernel(GbTable gt, int64_t nslots, unsigned* __restrict__ tile_range, int ntile_words,
                                                      unsigned long long* __restrict__ dscal20, unsigned long long* __restrict__ range,
                                                      int has_guess, unsigned long long guess_bits) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < nslots; s += stride) {
    gt.key[s] = kEmpty; gt.lo[s] = 0; gt.hi[s] = 0; gt.cnt[s] = 0; gt.first[s] = kNoRow;
  }
  for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < ntile_words; s += stride) tile_range[s] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dscal20[0] = 0; dscal20[1] = 0; dscal20[2] = 0;   // [20] unused, [21] overflow / redo flags, [22] total
    dscal20[3] = ~0ull;                               // [23] null id: none
    if (has_guess) {   // fx_guess_kernel: the sampled maximum's exponent + kGuessMargin binades
      int e = (int)((guess_bits >> 52) & 0x7ff);
      e = (e ? e : 1) + kGuessMargin;
      range[0] = (unsigned long long)(e > 0x7fe ? 0x7fe : e) << 52;
    } else {
      range[0] = 0;
    }
    range[1] = 0;
  }
}

// the epilogue: scale-guess check (fx_guess_check_kernel), first-seen ranks by counting, the groups written out, the post.
// Entries come from two tables: rt (nullable) — the seeded slots added up by gd_reduce_kernel — and gt, the global table the
// unseeded keys (or, without a seed, all keys) were merged into.  A key is in one of them only.
// kFinBlocks workgroups: each gathers ALL used entries into its LDS (12 loads per lane), then ranks its share of them — sixteen
// lanes per entry, each counting a sixteenth of the first rows below the entry's (one workgroup ranking 2048 entries alone was
// 4 M comparisons on one CU: 70 µs).  The last workgroup to finish posts {flags, groups, null group}.
constexpr int kFinMax = kLSlots + kGStride;
constexpr int kFinBlocks = 16;
template <bool FX>
__global__ __launch_bounds__(1024) void gd_finish_kernel(GbTable rt, GbTable gt, const unsigned* __restrict__ tile_range, int nblocks, int check_guess,
                                                          unsigned long long* __restrict__ range, unsigned* __restrict__ flag, unsigned* __restrict__ done,
                                                          int* __restrict__ null_id,
                                                          unsigned long long* __restrict__ out_keys, unsigned long long* __restrict__ out_sums,
                                                          long long* __restrict__ out_counts, long long* __restrict__ out_first_rows,
                                                          unsigned long long* mb, unsigned long long seq) {
  __shared__ __attribute__((aligned(16))) unsigned s_first[kFinMax + 64];
  __shared__ unsigned short s_slot[kFinMax];
  __shared__ unsigned short s_mine[kFinMax / kFinBlocks + 64];   // this workgroup's share: the entries whose table position ≡ blockIdx (mod kFinBlocks) —
  __shared__ unsigned s_e[16], s_i[16];                          // a rule every workgroup agrees on, whatever order its own gathering came out in
  __shared__ unsigned s_used, s_nmine, s_last;
  const int t = threadIdx.x;
  if (t == 0) { s_used = 0; s_nmine = 0; }
  if (check_guess && blockIdx.x == 0) {   // (block-uniform)
    unsigned emax = 0, imin = 0;
    for (int b = t; b < nblocks; b += 1024) {
      const unsigned a = tile_range[2 * b], c = tile_range[2 * b + 1];
      emax = a > emax ? a : emax;
      imin = c > imin ? c : imin;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned a = __shfl_down(emax, o, 64), c = __shfl_down(imin, o, 64);
      emax = a > emax ? a : emax;
      imin = c > imin ? c : imin;
    }
    if ((t & 63) == 0) { s_e[t >> 6] = emax; s_i[t >> 6] = imin; }
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < 16; w++) { emax = s_e[w] > emax ? s_e[w] : emax; imin = s_i[w] > imin ? s_i[w] : imin; }
      const int eg = (int)((range[0] >> 52) & 0x7ff);
      if ((int)emax > eg + 2) atomicOr(flag, 2u);                              // the guess was too low: the accumulators may have wrapped
      if (imin && eg - (0x7ff - (int)imin) > 42) atomicOr(flag, 2u);          // too wide for one scale
      range[1] = imin;
    }
  }
  __syncthreads();
  // the used entries of both tables, in any order (twelve independent loads per lane in flight)
  constexpr int kPer = (kFinMax + 1023) / 1024;
  unsigned fr[kPer];
#pragma unroll
  for (int u = 0; u < kPer; u++) {
    const int sidx = u * 1024 + t;
    fr[u] = kNoRow;
    if (sidx < kLSlots) { if (rt.first) fr[u] = rt.first[sidx]; }
    else if (sidx < kFinMax) fr[u] = gt.first[sidx - kLSlots];
  }
#pragma unroll
  for (int u = 0; u < kPer; u++) {
    if (fr[u] != kNoRow) {
      const unsigned e = atomicAdd(&s_used, 1u);
      s_first[e] = fr[u];
      s_slot[e] = (unsigned short)(u * 1024 + t);
      if ((unsigned)(u * 1024 + t) % (unsigned)kFinBlocks == blockIdx.x) s_mine[atomicAdd(&s_nmine, 1u)] = (unsigned short)e;
    }
  }
  __syncthreads();
  const unsigned used = s_used, nmine = s_nmine;
  if (t < 64) s_first[used + t] = kNoRow;   // padding of the last sweep's reads: larger than every first row, never counted
  __syncthreads();
  int sh = 0;
  if (FX) sh = fx_shift(range[0]);
  const unsigned sub = t & 15u;
  for (unsigned m = (unsigned)t >> 4; m < ((nmine + 63u) & ~63u); m += 64u) {   // (the 16 lanes of an entry stay together)
    const bool live = m < nmine;
    const unsigned e = live ? (unsigned)s_mine[m] : 0u;
    const unsigned first = live ? s_first[e] : 0u;
    unsigned id = 0;
    for (unsigned j = sub * 4u; j < used; j += 64u) {
      const uint4 q = *(const uint4*)&s_first[j];
      id += (q.x < first) + (q.y < first) + (q.z < first) + (q.w < first);
    }
    id += __shfl_xor(id, 8, 64); id += __shfl_xor(id, 4, 64); id += __shfl_xor(id, 2, 64); id += __shfl_xor(id, 1, 64);
    if (!live || sub != 0) continue;
    const int sidx = (int)s_slot[e];
    const bool from_rt = sidx < kLSlots;
    const GbTable& tb = from_rt ? rt : gt;
    const int sl = from_rt ? sidx : sidx - kLSlots;
    const int special = from_rt ? kSlots : kGSlots;
    unsigned long long key = tb.key[sl];
    if (sl == special) key = kEmpty;
    if (sl == special + 1) { key = 0; *null_id = (int)id; }   // the null group's key slot keeps the fresh buffer's zero
    out_keys[id] = key;
    const unsigned cf = tb.cnt[sl];
    out_counts[id] = (long long)(cf & kCntMask);
    if (FX) {
      const unsigned f = cf >> 29;
      double r;
      if (f) r = (f & 1u) || (f & 6u) == 6u ? __builtin_nan("") : ((f & 2u) ? __builtin_inf() : -__builtin_inf());
      else r = fx_to_double(tb.lo[sl], tb.hi[sl], sh);
      out_sums[id] = __builtin_bit_cast(unsigned long long, r);
    } else {
      out_sums[id] = tb.lo[sl];
    }
    if (out_first_rows) out_first_rows[id] = (long long)first;
  }
  __syncthreads();
  if (t == 0) {
    __threadfence();
    s_last = atomicAdd(done, 1u) == (unsigned)kFinBlocks - 1u ? 1u : 0u;
  }
  __syncthreads();
  if (s_last && t == 0) {
    __threadfence();
    const unsigned fl = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int nid = __hip_atomic_load(null_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long w[3] = {(unsigned long long)fl, (unsigned long long)used, (unsigned long long)(long long)nid};
    ah_mailbox_post(mb, seq, w, 3);
  }
}

// guess_bits (nullable): the largest sampled |value| the caller's quick look saw — the scale guess then needs no sample pass here
static int gb_direct(ah_ctx* c, int is_f64, const uint64_t* keys, const uint8_t* kvalid, int64_t koff, const void* vals, const uint8_t* vvalid,
                     int64_t voff, int64_t n, uint64_t* out_keys, void* out_sums, int64_t* out_counts, int64_t* out_first_rows, int64_t* out_ngroups,
                     int32_t* out_null_group, int* used, const unsigned long long* guess_bits = nullptr, bool lean = false,
                     const unsigned long long* seed_keys = nullptr, unsigned seed_used = 0) {
  *used = 0;
  // a seed table at or beyond the admission limit must never reach gb_lds_slot, whose probe loop relies on empty slots being there
  // (the caller's "≤ 2800 distinct" gate already says so; this is the guard at the place that would hang)
  if (seed_keys && seed_used >= (unsigned)kSoftLimit) { seed_keys = nullptr; seed_used = 0; }
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const int64_t nslots = kGStride;
  const unsigned dgrid = (unsigned)ah_ceil_div(n, (int64_t)1 << kChunkLog2);
  const size_t stage_rows = seed_keys ? (size_t)dgrid * kLSlots : 0;
  const size_t need = pad((size_t)nslots * 8) * 3 + pad((size_t)nslots * 4) * 2 + pad((size_t)dgrid * 8) + pad(stage_rows * 8) * 2 + pad(stage_rows * 4) * 2 +
                      pad((size_t)kLSlots * 8) * 3 + pad((size_t)kLSlots * 4) * 2;
  uint8_t* base;
  int rc = ah_temp_reserve(c, need, (void**)&base);
  if (rc != AH_OK) { c->err[0] = 0; return AH_OK; }   // these temporaries do not fit: *used stays 0 and the caller's id-based path (a fraction of them) answers
  size_t off = 0;
  auto take = [&](size_t b) { uint8_t* q = base + off; off += pad(b); return q; };
  GbTable gt;
  gt.key = (unsigned long long*)take((size_t)nslots * 8);
  gt.lo = (unsigned long long*)take((size_t)nslots * 8);
  gt.hi = (unsigned long long*)take((size_t)nslots * 8);
  gt.cnt = (unsigned*)take((size_t)nslots * 4);
  gt.first = (unsigned*)take((size_t)nslots * 4);
  unsigned* tile_range = (unsigned*)take((size_t)dgrid * 8);
  GbStaging st{nullptr, nullptr, nullptr, nullptr};
  GbTable rt{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (seed_keys) {
    st.lo = (unsigned long long*)take(stage_rows * 8);
    st.hi = (unsigned long long*)take(stage_rows * 8);
    st.cnt = (unsigned*)take(stage_rows * 4);
    st.first = (unsigned*)take(stage_rows * 4);
    rt.key = (unsigned long long*)take((size_t)kLSlots * 8);
    rt.lo = (unsigned long long*)take((size_t)kLSlots * 8);
    rt.hi = (unsigned long long*)take((size_t)kLSlots * 8);
    rt.cnt = (unsigned*)take((size_t)kLSlots * 4);
    rt.first = (unsigned*)take((size_t)kLSlots * 4);
  }
  unsigned long long* absmax = (unsigned long long*)&c->dscalars[28];   // [28], [29]: the value range (ah_hashing.h)
  unsigned* overflow = (unsigned*)&c->dscalars[21];
  const unsigned long long* k64 = (const unsigned long long*)keys;
  const unsigned long long* v64 = (const unsigned long long*)vals;
  const bool guess = is_f64 && c->opt_groupby_scale_guess && n >= ((int64_t)1 << 22);
  const bool host_guess = guess && guess_bits != nullptr;
  gd_prep_kernel<<<16, 256, 0, c->stream>>>(gt, nslots, tile_range, (int)(2 * dgrid), (unsigned long long*)&c->dscalars[20], absmax, host_guess ? 1 : 0,
                                            host_guess ? *guess_bits : 0ull);
  AH_LAUNCH_CHECK(c);
  if (is_f64 && !guess) {   // the fixed-point scale needs the largest finite |value| before the first addend is converted
    absmax_kernel<<<ah_stream_grid(c, ah_ceil_div(n, (int64_t)kBlock * 8), 2), kBlock, 0, c->stream>>>(v64, vvalid, voff, n, absmax);
    AH_LAUNCH_CHECK(c);
    fx_range_check_kernel<<<1, 1, 0, c->stream>>>(absmax, overflow);   // a wide column goes to the id-based path (per-group scales)
    AH_LAUNCH_CHECK(c);
  }
  if (guess && !host_guess) {   // … or a guess from a sample of its own (the caller had no quick look), checked like the host's
    const int64_t stride = (n / kGuessGroups) & ~(int64_t)63;
    fx_sample_max_kernel<<<kGuessGroups / 4, 256, 0, c->stream>>>(v64, vvalid, voff, n, kGuessGroups, stride, absmax);
    AH_LAUNCH_CHECK(c);
    fx_guess_kernel<<<1, 1, 0, c->stream>>>(absmax);
    AH_LAUNCH_CHECK(c);
  }
  const unsigned grid = dgrid;
  unsigned* trng = guess ? tile_range : nullptr;
  if (is_f64 && lean) gb_aggregate_kernel<true, true, true><<<grid, kThreads, 0, c->stream>>>(k64, v64, nullptr, nullptr, 1, gt, absmax, overflow, 2, kvalid, koff, vvalid, voff, n, 0, trng, seed_keys, seed_used, st);
  else if (is_f64) gb_aggregate_kernel<true, true><<<grid, kThreads, 0, c->stream>>>(k64, v64, nullptr, nullptr, 1, gt, absmax, overflow, 2, kvalid, koff, vvalid, voff, n, 0, trng, seed_keys, seed_used, st);
  else gb_aggregate_kernel<false, true><<<grid, kThreads, 0, c->stream>>>(k64, v64, nullptr, nullptr, 1, gt, absmax, overflow, 2, kvalid, koff, vvalid, voff, n, 0, nullptr, seed_keys, seed_used, st);
  AH_LAUNCH_CHECK(c);
  if (seed_keys) {   // the seeded slots of all workgroups, slot by slot
    if (is_f64) gd_reduce_kernel<true><<<(unsigned)ah_ceil_div((int64_t)kLSlots, kRedSlots), 1024, 0, c->stream>>>(st, (int)dgrid, seed_keys, rt);
    else gd_reduce_kernel<false><<<(unsigned)ah_ceil_div((int64_t)kLSlots, kRedSlots), 1024, 0, c->stream>>>(st, (int)dgrid, seed_keys, rt);
    AH_LAUNCH_CHECK(c);
  }
  unsigned long long* mb;
  unsigned long long seq, w[3];
  if ((rc = ah_mailbox_begin(c, &mb, &seq)) != AH_OK) return rc;
  unsigned* done = (unsigned*)&c->dscalars[20];   // zeroed by gd_prep_kernel
  int* null_id = (int*)&c->dscalars[23];           // −1 from gd_prep_kernel
  if (is_f64) gd_finish_kernel<true><<<kFinBlocks, 1024, 0, c->stream>>>(rt, gt, tile_range, (int)dgrid, guess ? 1 : 0, absmax, overflow, done, null_id, (unsigned long long*)out_keys,
                                                                        (unsigned long long*)out_sums, (long long*)out_counts, (long long*)out_first_rows, mb, seq);
  else gd_finish_kernel<false><<<kFinBlocks, 1024, 0, c->stream>>>(rt, gt, tile_range, (int)dgrid, 0, absmax, overflow, done, null_id, (unsigned long long*)out_keys,
                                                                  (unsigned long long*)out_sums, (long long*)out_counts, (long long*)out_first_rows, mb, seq);
  AH_LAUNCH_CHECK(c);
  if ((rc = ah_mailbox_wait(c, seq, 3, w)) != AH_OK) return rc;   // {overflow / redo flags, groups, null group}
  if ((unsigned)w[0]) return AH_OK;
  if (out_ngroups) *out_ngroups = (int64_t)w[1];
  if (out_null_group) *out_null_group = (int32_t)(long long)w[2];
  *used = 1;
  return AH_OK;
}

// steps 1 … 4 of the partition-first group-by for 2^lp partitions.  hist (nullable): the sample's [8][1024] row counts — the
// RESERVING scatter is used (no histogram pass; ah_partition.h 1b) and *redo = 4 says a region was too small (the caller runs the
// call again without hist).  *used = 1: out_* hold the result.
static int gb_cut_aggregate(ah_ctx* c, int is_f64, int lp, const unsigned* hist, double sample_scale, const uint64_t* keys, const uint8_t* kvalid, int64_t koff,
                            const void* vals, const uint8_t* vvalid, int64_t voff, int64_t n, uint64_t* out_keys, void* out_sums, int64_t* out_counts,
                            int64_t* out_first_rows, int64_t* out_ngroups, int32_t* out_null_group, int* used, unsigned* redo) {
  *used = 0;
  *redo = 0;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  unsigned long long* absmax = (unsigned long long*)&c->dscalars[28];   // [28], [29]: the value range (ah_hashing.h)
  unsigned* overflow = (unsigned*)&c->dscalars[21];
  unsigned long long* total = (unsigned long long*)&c->dscalars[22];
  int* null_id = (int*)&c->dscalars[23];
  const bool reserve = hist != nullptr;
  const int P = 1 << lp;
  const int64_t ntiles = ah_ceil_div(n, kGbTile), ngrp = ah_ceil_div(ntiles, kGroupTiles);
  const int64_t nslots = (int64_t)P * kGStride;
  const int64_t nwords = ah_ceil_div(n, 64), nrt = rank_tiles(nwords);
  const int64_t xrows = ((ntiles + 7) >> 3) * kGbTile;   // rows of the tiles one XCD takes (xcd_contiguous_tile)
  // record arrays: dense (n rows) behind the offsets table; regions (≤ 1.5 n + slack, gb_layout_kernel checks) + kGbTile spare rows otherwise
  const int64_t cap_rows = reserve ? n + n / 2 + (int64_t)P * kGbRegions * 96 : n;
  const int64_t rec_rows = reserve ? cap_rows + kGbTile : n;
  const int nreg = P * kGbRegions;
  // ---- temporaries (one reservation)
  const size_t table = reserve ? 0 : (size_t)P * (size_t)ntiles * 4;
  const size_t need = pad((size_t)nslots * 8) * 3 + pad((size_t)nslots * 4) * 2 + pad((size_t)nwords * 8) + pad((size_t)nwords * 4) +
                      pad((size_t)nrt * 4) + pad((size_t)nrt * 8) + pad(table) * 2 + pad((size_t)ngrp * P * 4) + pad((size_t)(P + 1) * 4) +
                      pad((size_t)rec_rows * 8) * 2 + pad((size_t)rec_rows * 4) + pad((size_t)ntiles * 16) + pad((size_t)(nreg + 1) * 4) * 5;
  uint8_t* base;
  int rc = ah_temp_reserve(c, need, (void**)&base);
  if (rc != AH_OK && reserve) {
    // the regions want 1.5 n rows of records (≈ 30 B/row) where the dense arrays behind a histogram want n (20 B/row): a call that does
    // not fit THAT way is tried once more the other way (redo = 4: the caller runs the call again without the sample's histogram)
    c->err[0] = 0;
    *redo = 4u;
    return AH_OK;
  }
  if (rc != AH_OK) return rc;
  size_t used_b = 0;
  auto take = [&](size_t b) { uint8_t* q = base + used_b; used_b += pad(b); return q; };
  GbTable gt;
  gt.key = (unsigned long long*)take((size_t)nslots * 8);
  gt.lo = (unsigned long long*)take((size_t)nslots * 8);
  gt.hi = (unsigned long long*)take((size_t)nslots * 8);
  gt.cnt = (unsigned*)take((size_t)nslots * 4);
  gt.first = (unsigned*)take((size_t)nslots * 4);
  unsigned long long* firsts = (unsigned long long*)take((size_t)nwords * 8);
  unsigned* wordprefix = (unsigned*)take((size_t)nwords * 4);
  int* tilecnt = (int*)take((size_t)nrt * 4);
  int64_t* tileoff = (int64_t*)take((size_t)nrt * 8);
  unsigned* cnt_tm = (unsigned*)take(table);
  unsigned* toffs = (unsigned*)take(table);
  unsigned* gsum = (unsigned*)take((size_t)ngrp * P * 4);
  unsigned* binstart = (unsigned*)take((size_t)(P + 1) * 4);
  unsigned long long* pkeys = (unsigned long long*)take((size_t)rec_rows * 8);
  unsigned long long* pvals = (unsigned long long*)take((size_t)rec_rows * 8);
  unsigned* prows = (unsigned*)take((size_t)rec_rows * 4);
  unsigned long long* tile_max = (unsigned long long*)take((size_t)ntiles * 16);
  unsigned* rstart = (unsigned*)take((size_t)(nreg + 1) * 4);
  unsigned* rcap = (unsigned*)take((size_t)(nreg + 1) * 4);
  unsigned* cursor = (unsigned*)take((size_t)(nreg + 1) * 4);
  unsigned* vstart = (unsigned*)take((size_t)(nreg + 1) * 4);
  unsigned* delta = (unsigned*)take((size_t)(nreg + 1) * 4);
  {
    // everything the call starts from, one launch: empty tables (lo, hi, cnt are adjacent), no first rows, the call's scalars
    GbFill f;
    f.njobs = 7;
    f.p[0] = (uint4*)gt.key; f.n16[0] = pad((size_t)nslots * 8) / 16; f.v[0] = 0xFFFFFFFFu;
    f.p[1] = (uint4*)gt.lo; f.n16[1] = (pad((size_t)nslots * 8) * 2 + pad((size_t)nslots * 4)) / 16; f.v[1] = 0u;
    f.p[2] = (uint4*)gt.first; f.n16[2] = pad((size_t)nslots * 4) / 16; f.v[2] = 0xFFFFFFFFu;
    f.p[3] = (uint4*)firsts; f.n16[3] = pad((size_t)nwords * 8) / 16; f.v[3] = 0u;
    f.p[4] = (uint4*)&c->dscalars[20]; f.n16[4] = 1; f.v[4] = 0u;                 // [20] unused, [21] overflow / redo flags
    f.p[5] = (uint4*)&c->dscalars[28]; f.n16[5] = 1; f.v[5] = 0u;                 // [28], [29] value range
    f.p[6] = (uint4*)&c->dscalars[22]; f.n16[6] = 1; f.v[6] = 0u;                 // [22] total, [23] …
    f.ones = (unsigned long long*)null_id;                                         // … null id: none
    gb_fill_kernel<<<(unsigned)(c->num_cu * 4), 256, 0, c->stream>>>(f);
    AH_LAUNCH_CHECK(c);
  }
  const unsigned long long* k64 = (const unsigned long long*)keys;
  const unsigned long long* v64 = (const unsigned long long*)vals;
  // ---- 1, 2: cut
  const unsigned tgrid = (unsigned)(((ntiles + 7) / 8) * 8);
  if (reserve) {
    gb_layout_kernel<<<1, 1024, 0, c->stream>>>(hist, lp, n, xrows, sample_scale, cap_rows, rstart, rcap, cursor, overflow);
    AH_LAUNCH_CHECK(c);
    gb_scatter_kernel<true, true><<<tgrid, kThreads, 0, c->stream>>>(k64, kvalid, koff, v64, vvalid, voff, n, lp, P, ntiles, nullptr, pkeys, pvals, prows,
                                                                     is_f64 ? tile_max : nullptr, rstart, rcap, cursor, overflow, (unsigned)cap_rows);
    AH_LAUNCH_CHECK(c);
    // the regions as the aggregate pass walks them + the value range (gb_max_kernel and fx_range_check_kernel in one)
    gb_segments_kernel<<<1, 1024, 0, c->stream>>>(rstart, rcap, cursor, P, n, vstart, delta, binstart, overflow, is_f64 ? tile_max : nullptr, ntiles, absmax);
    AH_LAUNCH_CHECK(c);
  } else {
  gb_hist_kernel<<<tgrid, kGbHistThreads, 0, c->stream>>>(k64, kvalid, koff, n, lp, P, ntiles, cnt_tm);
  AH_LAUNCH_CHECK(c);
  colsum_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt_tm, P, ntiles, gsum);
  AH_LAUNCH_CHECK(c);
  bin_prefix_kernel<<<1, kMaxBins, 0, c->stream>>>(gsum, P, ngrp, n, binstart);
  AH_LAUNCH_CHECK(c);
  tile_offs_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt_tm, gsum, P, ntiles, toffs);
  AH_LAUNCH_CHECK(c);
  gb_scatter_kernel<true><<<tgrid, kThreads, 0, c->stream>>>(k64, kvalid, koff, v64, vvalid, voff, n, lp, P, ntiles, toffs, pkeys, pvals, prows,
                                                       is_f64 ? tile_max : nullptr);
  AH_LAUNCH_CHECK(c);
  if (is_f64) {   // the fixed-point scale needs the largest finite |value| before the first addend is converted
    gb_max_kernel<<<1, 1024, 0, c->stream>>>(tile_max, ntiles, absmax);
    AH_LAUNCH_CHECK(c);
    fx_range_check_kernel<<<1, 1, 0, c->stream>>>(absmax, overflow);   // a wide column goes to the id-based path (per-group scales)
    AH_LAUNCH_CHECK(c);
  }
  }
  // ---- 3: aggregate
  // One workgroup fits a CU: one round of equal shares (≥ 2^17 records each: a table costs its 136 KiB to set up and to write
  // out).  Shares of the size of a partition where the partitions are smaller — 1024 workgroups for 1024 partitions, handed out
  // by the hardware as CUs come free — measured SLOWER (2^20 groups: 1.89 → 2.12 ms).
  int64_t nwg = n >> 17;
  nwg = nwg < 1 ? 1 : (nwg > c->num_cu ? c->num_cu : nwg);
  const int64_t seg_rows = ah_ceil_div(n, nwg);
  const unsigned grid = (unsigned)ah_ceil_div(n, seg_rows);
  // (the row-by-row LEAN mode of the direct path was measured here too, on evenly spread keys: 1.227 ms against 1.217 at 2^16 groups,
  // 1.546 against 1.563 at 2^20 — nothing, although it frees eight registers: this pass is not bound by its instruction count alone)
  const GbStaging nost{nullptr, nullptr, nullptr, nullptr};
  if (reserve) {
    if (is_f64 && c->opt_groupby_lean == 3) gb_aggregate_kernel<true, false, true, true><<<grid, kThreads, 0, c->stream>>>(pkeys, pvals, prows, binstart, P, gt, absmax, overflow, 0, nullptr, 0, nullptr, 0, 0, seg_rows, nullptr, nullptr, 0, nost, vstart, delta);
    else if (is_f64) gb_aggregate_kernel<true, false, false, true><<<grid, kThreads, 0, c->stream>>>(pkeys, pvals, prows, binstart, P, gt, absmax, overflow, 0, nullptr, 0, nullptr, 0, 0, seg_rows, nullptr, nullptr, 0, nost, vstart, delta);
    else gb_aggregate_kernel<false, false, false, true><<<grid, kThreads, 0, c->stream>>>(pkeys, pvals, prows, binstart, P, gt, absmax, overflow, 0, nullptr, 0, nullptr, 0, 0, seg_rows, nullptr, nullptr, 0, nost, vstart, delta);
  } else {
    if (is_f64) gb_aggregate_kernel<true><<<grid, kThreads, 0, c->stream>>>(pkeys, pvals, prows, binstart, P, gt, absmax, overflow, 0, nullptr, 0, nullptr, 0, 0, seg_rows);
    else gb_aggregate_kernel<false><<<grid, kThreads, 0, c->stream>>>(pkeys, pvals, prows, binstart, P, gt, absmax, overflow, 0, nullptr, 0, nullptr, 0, 0, seg_rows);
  }
  AH_LAUNCH_CHECK(c);
  // ---- 4: rank the groups by first row, write them out
  gb_mark_kernel<<<ah_stream_grid(c, ah_ceil_div(nslots, kBlock)), kBlock, 0, c->stream>>>(gt.first, nslots, firsts);
  AH_LAUNCH_CHECK(c);
  word_prefix_kernel<<<(unsigned)ah_ceil_div(nwords, kBlock), kBlock, 0, c->stream>>>(firsts, nwords, wordprefix, tilecnt);
  AH_LAUNCH_CHECK(c);
  scan_kernel<<<1, 1024, 0, c->stream>>>(tilecnt, nrt, tileoff, total);
  AH_LAUNCH_CHECK(c);
  const unsigned egrid = ah_stream_grid(c, ah_ceil_div(nslots, kBlock));
  if (is_f64) gb_emit_kernel<true><<<egrid, kBlock, 0, c->stream>>>(gt, nslots, firsts, wordprefix, tileoff, absmax, (unsigned long long*)out_keys,
                                                                  (unsigned long long*)out_sums, (long long*)out_counts, (long long*)out_first_rows, null_id, kGStride, kGSlots);
  else gb_emit_kernel<false><<<egrid, kBlock, 0, c->stream>>>(gt, nslots, firsts, wordprefix, tileoff, absmax, (unsigned long long*)out_keys,
                                                            (unsigned long long*)out_sums, (long long*)out_counts, (long long*)out_first_rows, null_id, kGStride, kGSlots);
  AH_LAUNCH_CHECK(c);
  { int mrc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[21], 3, (unsigned long long*)&c->pinned[8]); if (mrc != AH_OK) return mrc; }   // overflow, total, null id
  *redo = *(volatile unsigned*)&c->pinned[8];
  if (*redo) return AH_OK;   // a partition held far more keys than estimated (1), a wide column (2): the caller's path redoes the call; a region too small (4): once more with the histogram
  if (out_ngroups) *out_ngroups = (int64_t) * (volatile uint64_t*)&c->pinned[9];
  if (out_null_group) *out_null_group = *(volatile int32_t*)&c->pinned[10];
  *used = 1;
  return AH_OK;
}

// Called by ah_hash_sum_* before the id-based path.  *used = 1: out_* hold the result; 0: not applicable (small input,
// too many expected groups, or the estimate was so far off that a partition's table overflowed) — the caller runs the
// id-based path, which accepts anything.
int ah_groupby_partitioned_try(ah_ctx* c, int is_f64, const uint64_t* keys, const uint8_t* kvalid, int64_t koff, const void* vals,
                               const uint8_t* vvalid, int64_t voff, int64_t n, uint64_t* out_keys, void* out_sums, int64_t* out_counts,
                               int64_t* out_first_rows, int64_t* out_ngroups, int32_t* out_null_group, int* used) {
  *used = 0;
  const int mode = c->opt_groupby_partition;   // 0 never, 1 auto, 2: always sort-based, 3 / 4: always the two-level cut with 2^11 / 2^13 partitions, k ≥ 5: always LDS tables in 2^(k − 2) partitions (tests, measurements)
  if (mode == 0 || n >= kMaxRows || n < 1 || (mode == 1 && n < ((int64_t)1 << 21))) return AH_OK;
  // the scratch arena holds what must survive the temporaries' reservation further down: the look's key table + words, the sample's
  // two counts + done word, its histogram and its two bitmaps (the last three adjacent: one memset)
  constexpr int kSampleGroups = 1 << 15;            // × 64 consecutive rows = 2^21 sampled rows
  constexpr unsigned kBits = 1u << 24;
  constexpr size_t kSeedBytes = (size_t)kSlots * 8 + 256, kAccBytes = 256, kHistBytes = (size_t)kGbRegions * 1024 * 4, kBmBytes = 2 * (size_t)(kBits / 8);
  uint8_t* sbase = nullptr;
  if (mode == 1 || mode > 4) {
    int src = ah_scratch_reserve(c, kSeedBytes + kAccBytes + kHistBytes + kBmBytes, (void**)&sbase);
    if (src != AH_OK) return src;
  }
  void* seedbuf = sbase;
  unsigned long long* sample_acc = (unsigned long long*)(sbase + kSeedBytes);   // [0] |A|, [1] |A ∪ B|, [2] workgroups done
  unsigned* hist_buf = (unsigned*)(sbase + kSeedBytes + kAccBytes);
  unsigned* sample_bm = (unsigned*)(sbase + kSeedBytes + kAccBytes + kHistBytes);
  const unsigned* sample_hist = nullptr;   // set once the sample has filled hist_buf: the reserving scatter sizes its regions from it
  double sample_scale = 0.0;
  // the 2^21-row sample: two points of the distinct-count curve (linear counting) and, for the reserving scatter, the rows by
  // {eighth of the column, 1024 hash buckets}.  Three launches: one memset, the sample, the count that posts to the mailbox.
  const int64_t sgroups = (n / 64 < kSampleGroups ? n / 64 : kSampleGroups) & ~(int64_t)1;
  const double sampled = (double)sgroups * 64.0;
  auto run_sample = [&](uint64_t* set_half, uint64_t* set_full) -> int {
    const int64_t stride = ((n / sgroups) & ~(int64_t)63) ? ((n / sgroups) & ~(int64_t)63) : 64;
    const int64_t xrows = ((ah_ceil_div(n, kGbTile) + 7) >> 3) * kGbTile;
    const bool want_hist = c->opt_groupby_reserve != 0;
    AH_HIP(c, hipMemsetAsync(sample_acc, 0, kAccBytes + kHistBytes + kBmBytes, c->stream));
    const unsigned half_grid = (unsigned)ah_ceil_div(sgroups / 2, 16 * kSampleBatches);
    gb_sample_kernel<<<2 * half_grid, 1024, 0, c->stream>>>((const unsigned long long*)keys, kvalid, koff, n, sgroups, stride, sample_bm, kBits - 1,
                                                            want_hist ? hist_buf : nullptr, xrows);
    AH_LAUNCH_CHECK(c);
    unsigned long long* mb;
    unsigned long long seq, w[2];
    int rc = ah_mailbox_begin(c, &mb, &seq);
    if (rc != AH_OK) return rc;
    gb_sample_finish_kernel<<<256, 256, 0, c->stream>>>((const unsigned long long*)sample_bm, (int64_t)(kBits / 64), sample_acc, (unsigned*)&sample_acc[2], mb, seq);
    AH_LAUNCH_CHECK(c);
    if ((rc = ah_mailbox_wait(c, seq, 2, w)) != AH_OK) return rc;
    *set_half = w[0];
    *set_full = w[1];
    if (want_hist) { sample_hist = hist_buf; sample_scale = (double)n / sampled; }
    return AH_OK;
  };
  // ---- 0: how many partitions?
  int lp;
  if (mode == -2) return gb_direct(c, is_f64, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows, out_ngroups, out_null_group, used);
  if (mode == 3 || mode == 4) return gb2_groupby(c, is_f64, mode == 3 ? 11 : 13, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows, out_ngroups, out_null_group, used);
  if (mode == 2) return gs_groupby(c, is_f64, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows, out_ngroups, out_null_group, used);
  if (mode > 1) {
    lp = mode - 2 < 3 ? 3 : (mode - 2 > 10 ? 10 : mode - 2);
    if (c->opt_groupby_reserve != 0 && sgroups >= 2) {   // a forced partition count (tests, measurements): the sample runs for its histogram alone
      uint64_t a, b;
      int rc = run_sample(&a, &b);
      if (rc != AH_OK) return rc;
    }
  } else {
    double est = -1.0;
    bool heavy_tail = false;
    if (is_f64) {
      // the quick look: 2^14 spread rows, one workgroup, posted by the kernel itself (≈ 35 µs against ≈ 75 for the two-point sample
      // below).  A column with few enough groups for the direct path is decided here, with the value maximum its scale guess needs.
      unsigned long long* mb;
      unsigned long long seq, w[5];
      // seedbuf: the look's key table, the seed of the direct path's LDS tables (the scratch arena: the direct path's own temporaries are in the other one)
      int qrc;
      AH_HIP(c, hipMemsetAsync(seedbuf, 0xFF, (size_t)kSlots * 8 + 64, c->stream));   // the empty table and the look's words (all "−1")
      if ((qrc = ah_mailbox_begin(c, &mb, &seq)) != AH_OK) return qrc;
      const int64_t qstride = ((n / kQlGroups) & ~(int64_t)(kQlRun - 1)) ? ((n / kQlGroups) & ~(int64_t)(kQlRun - 1)) : kQlRun;
      gq_quicklook_kernel<<<kQlBlocks, 1024, 0, c->stream>>>((const unsigned long long*)keys, kvalid, koff, (const unsigned long long*)vals, vvalid, voff, n, qstride,
                                                             (unsigned long long*)seedbuf, (unsigned*)((uint8_t*)seedbuf + (size_t)kSlots * 8), mb, seq);
      AH_LAUNCH_CHECK(c);
      if ((qrc = ah_mailbox_wait(c, seq, 5, w)) != AH_OK) return qrc;
      const double qrows = (double)(n < (int64_t)kQlGroups * kQlRun ? n : (int64_t)kQlGroups * kQlRun);
      // neighbouring rows that share a key: once in d rows where d keys are drawn evenly; far more often = a clustered (sorted, run-
      // length) column, whose 512 sample runs say little about its distinct count — left to the 2^21-row sample below
      const double adj_rate = (double)w[4] / (qrows > 64.0 ? qrows * (double)(kQlRun - 1) / (double)kQlRun : 1e30);
      const bool clustered = w[0] > 1 && adj_rate > 4.0 / (double)w[0] + 0.02;
      // (up to 3000 expected groups — the LDS table admits 3584 —, where the old rule sent ≤ 2048 here: with seeded tables the
      // merge no longer grows with the group count, 2^11 groups: 0.83 → 0.6 ms)
      if (!clustered && w[0] <= 2800 && gb_extrapolate((double)w[0], qrows, (double)n) <= 3000.0) {
        // rows 1024 apart are a lane's consecutive rows in the aggregate pass: where fewer than one in eight of them share a key the
        // pending-group registers are left out (gb_aggregate_kernel, LEAN)
        const unsigned pairs = (unsigned)(w[2] >> 32), rep = (unsigned)w[2];
        const bool lean = c->opt_groupby_lean != 0 && (c->opt_groupby_lean == 2 || (pairs >= 256 && rep * 8u < pairs));
        const bool seed = c->opt_groupby_seed != 0;
        return gb_direct(c, is_f64, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows, out_ngroups, out_null_group, used, &w[1], lean,
                         seed ? (const unsigned long long*)seedbuf : nullptr, (unsigned)w[3]);
      }
      // (The look's own two points — a 2^16-bit linear-counting bitmap read after half of the sample groups and after all — were
      // tried as the estimate for up to ≈ 10^5 groups too, sparing the 2^21-row sample below: evenly drawn keys came out within 2 %,
      // but a Zipf(1.1) column over 2^20 keys looked like 7000 keys in 2^14 rows, got 32 partitions and took 5.9 ms instead of 1.5.
      // The tail of such a column is only seen by a sample of millions of rows: the look decides the direct path and nothing else.)
    }
    if (est < 0.0) {
    auto distinct = [&](uint64_t set, double rows) {   // linear counting: M·ln(M / zeros)
      const double z = (double)kBits - (double)set;
      const double d = z < 1.0 ? rows : -(double)kBits * log(z / (double)kBits);
      return d > rows ? rows : d;
    };
    uint64_t set_half = 0, set_full = 0;
    int rc = run_sample(&set_half, &set_full);
    if (rc != AH_OK) return rc;
    const double dh = distinct(set_half, sampled / 2), ds = distinct(set_full, sampled);
    est = gb_extrapolate(ds, sampled, (double)n);
    if (est <= 2048.0 && is_f64) return gb_direct(c, is_f64, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows, out_ngroups, out_null_group, used);
    // Keys drawn evenly from C values give a curve that the second half of the sample must follow; a heavy-tailed column
    // (Zipf) keeps bringing new keys long after that curve has flattened, and its full distinct count is several times the
    // even-draw extrapolation — give those columns 4× the partitions rather than let the LDS tables overflow into the
    // global ones (measured: 0.45 → 0.89 ms in the aggregate pass of a Zipf(1.1) column over 2^20 keys)
    const double dh_even = gb_extrapolate(ds, sampled, sampled / 2);
    heavy_tail = dh < 0.93 * dh_even;
    }
    if (est <= 4300.0) return AH_OK;                   // all groups still fit the id-based path's LDS table: ≈ 1 ms there, no better here
    // Expected keys per partition: ≈ 1024–1280 (a quarter of the LDS table) while that needs ≤ 256 partitions; beyond, the
    // scatter's runs get short (4096-row tiles / 1024 partitions = 4 rows) and costs more than fuller tables do — up to 2100
    // keys per partition then (the table admits 3584).  Measured at 2^26 rows (`scripts/bench_gb_keys.py`): 2^19 groups 1.62 →
    // 1.46 ms, 2^20 1.90 → 1.69 ms, 2^21 (one level of 1024 instead of two levels) 2.9 → 2.1 ms; 2^14 … 2^18 are best at ≈ 1024.
    const double kpp = (double)(c->opt_groupby_keys < 256 ? 256 : (c->opt_groupby_keys > 3000 ? 3000 : c->opt_groupby_keys));
    const double kpp_many = kpp > 2100.0 ? kpp : 2100.0;
    if (est > 1024.0 * kpp_many) {
      // beyond 1024 partitions of ≤ 1280 keys.  Evenly spread keys: up to 8192 partitions through the two-level cut, one workgroup
      // each; beyond that (fewer than ≈ 8 rows per group) sort-based buckets.  A heavy tail at this size means keys with thousands
      // of rows — one workgroup per partition or a bucket of 64 cannot take those: the id-based path
      if (heavy_tail || n < ((int64_t)1 << 22) || n > ((int64_t)1 << 27)) return AH_OK;
      if (est <= 8192.0 * kpp_many) {   // 17 M: 8192 partitions of ≤ 2100 expected keys (2^24 groups in 2^26 rows: 5.3 ms this way, 6.3 ms sort-based)
        int lp2 = 11;
        while (lp2 < 13 && est / (double)(1 << lp2) > 1280.0) lp2++;   // (2100 here: 2^22 groups 3 % slower, 2^23 3 % faster)
        return gb2_groupby(c, is_f64, lp2, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows, out_ngroups, out_null_group, used, true, sample_hist, sample_scale);
      }
      return gs_groupby(c, is_f64, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows, out_ngroups, out_null_group, used);
    }
    if (heavy_tail) est *= 4.0;   // up to 1024 partitions (the loop below stops there)
    lp = 3;                                            // ≤ 1280 expected keys per partition (LDS table: 3584), a few partitions at least
    while (lp < 10 && est / (double)(1 << lp) > kpp) lp++;
    if (lp > 8) {
      lp = 8;
      while (lp < 10 && est / (double)(1 << lp) > kpp_many) lp++;
    }
  }
  // the reserving scatter (ah_partition.h 1b) first when the sample left its histogram; a region that turned out too small: once more
  // with the histogram pass
  unsigned redo = 0;
  int rc = gb_cut_aggregate(c, is_f64, lp, sample_hist, sample_scale, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows,
                            out_ngroups, out_null_group, used, &redo);
  if (rc == AH_OK && !*used && sample_hist && redo == 4u)
    rc = gb_cut_aggregate(c, is_f64, lp, nullptr, 0.0, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows, out_ngroups,
                          out_null_group, used, &redo);
  return rc;
}
