// ah_hash.hip — unique / dictionary_encode / group-by-sum over 8-byte keys.
//
// Replaces: doAppendNumeric[uint64] (kernels/vector_hash.go:359-385) driving
//   hashing.Table[uint64].InsertOrGet / GetOrInsertNull
//   (internal/hashing/xxh3_memo_table_types.go:231-238,283-294; open-addressing table
//   :45-187; hashInt internal/hashing/hash_funcs.go:60-67), the
//   dictionaryEncodeAction (vector_hash.go:145-241) and uniqueFinalize (:721-741),
//   behind compute's "unique" / "dictionary_encode" (compute/vector_hash.go:61,79).
//   Int64 and Float64 columns both hash raw 64-bit patterns (vector_hash.go:604-607).
//
// The reference is strictly sequential: memo index = order of first insertion.  The
// same ids are produced in parallel as follows:
//   1. insert_kernel  — every row CASes its key into an open-addressing table keyed
//      by the reference's hashInt (linear probing; 16-byte slots {key, first_row,
//      id}) and atomicMin's its row number into the slot: first_row = first
//      occurrence.  The all-ones key (the table's EMPTY marker) and the null key live
//      in two dedicated slots past the end.  Plain loads short-circuit both atomics
//      once a slot is settled, so low-cardinality columns do not serialise on L2.
//   2. mark_kernel    — each used slot sets bit first_row in an n-bit "first
//      occurrence" bitmap.
//   3. rank kernels   — prefix popcount of that bitmap (per-64-row word prefixes inside
//      2048-row tiles + one scan of tile totals): the number of first occurrences
//      before row r IS the sequential memo index of the key first seen at row r.
//   4. assign_kernel  — each used slot computes its id from (3) and writes its key to
//      dict[id].
//   5. emit_kernel    — out_ids[i] = id of row i's slot (remembered in step 1).
// If more than cap/2 distinct keys show up the attempt is abandoned and repeated
// with a 16× larger table (the reference grows ×4 at the same load factor,
// xxh3_memo_table_types.go:109,169-179); results never depend on the capacity.
//
// HBM: 8 B/row of keys + random 16-byte slot traffic (atomics-bound at high
// cardinality, cache-resident at low cardinality).
#include <type_traits>
#include "ah_common.h"
#include "ah_hashing.h"

namespace {


struct Slot {
  unsigned long long key;
  unsigned first_row;
  unsigned id;
};


// ---- what a "key" is -------------------------------------------------------------------------------
// The table machinery below (insert → rank → emit) only needs, per row: a 64-bit word to CAS into an empty
// slot, a hash to start probing at, and "does this occupied slot hold my key".
struct U64Keys {  // Table[uint64] (xxh3_memo_table_types.go): the word is the key itself
  const unsigned long long* keys;
  static constexpr bool kLdsTable = true;
  // false → the all-ones key, which doubles as the EMPTY marker and lives in its own slot
  __device__ __forceinline__ bool load(int64_t i, unsigned long long* word, uint64_t* h) const {
    const unsigned long long k = keys[i];
    *word = k;
    *h = hash_int(k);
    return k != kEmpty;
  }
  __device__ __forceinline__ bool same(unsigned long long cur, unsigned long long word, int64_t) const { return cur == word; }
};

struct U64u { unsigned long long v; } __attribute__((packed, aligned(1)));
__device__ __forceinline__ unsigned long long load8(const uint8_t* p) { return reinterpret_cast<const U64u*>(p)->v; }
__device__ __forceinline__ unsigned long long load_tail(const uint8_t* p, int nb) {  // 1..7 bytes, little-endian
  unsigned long long w = 0;
  for (int t = 0; t < nb; t++) w |= (unsigned long long)p[t] << (8 * t);
  return w;
}
// Any 64-bit hash will do: ids and dictionary order depend only on which rows are EQUAL and on row order,
// never on hash values (the reference's xxh3 / custom short-string hash, hash_funcs.go:86-124, decides
// only where its own memo table stores an entry).  Multiply-xorshift over 8-byte words.
__device__ __forceinline__ uint64_t hash_bytes(const uint8_t* p, int64_t len) {
  uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)len * 0xC2B2AE3D27D4EB4Full);
  int64_t j = 0;
  for (; j + 8 <= len; j += 8) {
    h = (h ^ load8(p + j)) * 0xFF51AFD7ED558CCDull;
    h ^= h >> 32;
  }
  if (j < len) {
    h = (h ^ load_tail(p + j, (int)(len - j))) * 0xC4CEB9FE1A85EC53ull;
    h ^= h >> 29;
  }
  h *= 0x9FB21C651E98DF25ull;
  return h ^ (h >> 32);
}
__device__ __forceinline__ bool equal_bytes(const uint8_t* a, const uint8_t* b, int64_t len) {
  int64_t j = 0;
  for (; j + 8 <= len; j += 8)
    if (load8(a + j) != load8(b + j)) return false;
  return j == len || load_tail(a + j, (int)(len - j)) == load_tail(b + j, (int)(len - j));
}

// BinaryMemoTable (internal/hashing/xxh3_memo_table.go): the slot word is {hash bits 63..32 | a row that
// holds the value}; a slot is "mine" when the tags agree and the bytes of that row equal mine.  Both
// halves arrive in one CAS, so a reader never sees a tag without its row.
template <typename OffT>
struct BinKeys {
  const OffT* offsets;  // of row 0 of the call (buffer + array offset)
  const uint8_t* data;
  static constexpr bool kLdsTable = false;
  __device__ __forceinline__ bool load(int64_t i, unsigned long long* word, uint64_t* h) const {
    const int64_t b = (int64_t)offsets[i], e = (int64_t)offsets[i + 1];
    *h = hash_bytes(data + b, e - b);
    *word = (*h & 0xFFFFFFFF00000000ull) | (unsigned long long)(unsigned)i;  // never all-ones: i < 2^32 − 1
    return true;
  }
  __device__ __forceinline__ bool same(unsigned long long cur, unsigned long long word, int64_t i) const {
    if ((cur ^ word) >> 32) return false;
    const int64_t r = (int64_t)(unsigned)cur;
    if (r == i) return true;
    const int64_t b = (int64_t)offsets[i], e = (int64_t)offsets[i + 1], rb = (int64_t)offsets[r], re = (int64_t)offsets[r + 1];
    return e - b == re - rb && equal_bytes(data + b, data + rb, e - b);
  }
};

// FixedSizeBinary / Decimal128 / Decimal256 keys (kernels/vector_hash.go:608-609, 698: the same BinaryMemoTable, every value
// `w` bytes): the BinKeys scheme with the offsets implied — value i lives at data + i·w.
struct FixKeys {
  const uint8_t* data;  // of row 0 of the call
  int w;
  static constexpr bool kLdsTable = false;
  __device__ __forceinline__ bool load(int64_t i, unsigned long long* word, uint64_t* h) const {
    *h = hash_bytes(data + i * w, w);
    *word = (*h & 0xFFFFFFFF00000000ull) | (unsigned long long)(unsigned)i;
    return true;
  }
  __device__ __forceinline__ bool same(unsigned long long cur, unsigned long long word, int64_t i) const {
    if ((cur ^ word) >> 32) return false;
    const int64_t r = (int64_t)(unsigned)cur;
    return r == i || equal_bytes(data + i * w, data + r * w, w);
  }
};

// dictionary of fixed-width keys: entry id = the value at its first row (the null entry: zeros, like a fresh builder slot)
__global__ __launch_bounds__(kBlock) void fixed_dict_kernel(const uint8_t* __restrict__ data, int w, const long long* __restrict__ first_rows, int64_t ndict,
                                                             int null_id, uint8_t* __restrict__ dict) {
  const int64_t total = ndict * w;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < total; b += stride) {
    const int64_t id = b / w;
    dict[b] = id == null_id ? (uint8_t)0 : data[first_rows[id] * w + (b - id * w)];
  }
}

// status words in dscalars: [4] distinct count, [5] overflow flag, [6] total ids
// Inserts rows [lo, hi).  Probe chains longer than kProbeLimit mean the table is far beyond
// the load it was sized for (at load ≤ ½ the chance of a 256-long linear-probing cluster is
// ≈ 0.82^256 ≈ 1e-22 per insert) → flag overflow; the host retries with a larger table.
constexpr int kProbeLimit = 256;

constexpr int kInsertRows = 4;  // independent rows per lane per step: their first probes are in flight together

struct SlotView {  // one 16-byte load: the fields of a slot as they were at (about) the same time
  unsigned long long key;
  unsigned first_row;
  unsigned id;
};
__device__ __forceinline__ SlotView load_slot(const Slot* p) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  return SlotView{((unsigned long long)v.y << 32) | v.x, v.z, v.w};
}

template <typename K>
__global__ __launch_bounds__(kBlock) void insert_kernel(const K keys, const uint8_t* __restrict__ valid,
                                                         int64_t off, int64_t lo, int64_t hi, int encode_nulls, Slot* __restrict__ table,
                                                         uint64_t cap, unsigned* __restrict__ row_slot, unsigned flag, unsigned direct_below,
                                                         unsigned long long* __restrict__ distinct, unsigned* __restrict__ overflow,
                                                         unsigned long long* __restrict__ misses) {
  // direct_below > 0: the slots whose first_row lies below it have been ranked already (their id is final,
  // see encode_core) — such a row gets its id right here, every other row a slot number with `flag` set.
  const uint64_t mask = cap - 1;
  const int64_t stride = (int64_t)gridDim.x * kBlock * kInsertRows;
  unsigned nmiss = 0;
  unsigned fresh = 0;  // keys this lane inserted (published once, at the end: same-address atomics cost ~12 ns each)
  for (int64_t base = lo + (int64_t)blockIdx.x * kBlock * kInsertRows + threadIdx.x; base < hi; base += stride) {
    // kind: 0 = nothing to do, 1 = probe the table, 2 = dedicated slot (all-ones key / null), 3 = masked null
    int kind[kInsertRows];
    unsigned long long k[kInsertRows];  // the word this row would CAS into an empty slot
    uint64_t idx[kInsertRows];
    SlotView sv[kInsertRows];
#pragma unroll
    for (int u = 0; u < kInsertRows; u++) {
      const int64_t i = base + (int64_t)u * kBlock;
      kind[u] = 0;
      k[u] = 0;
      idx[u] = 0;
      if (i < hi) {
        if (ah_bit(valid, off + i)) {
          uint64_t h;
          if (!keys.load(i, &k[u], &h)) { kind[u] = 2; idx[u] = cap; }
          else { kind[u] = 1; idx[u] = h & mask; }
        } else if (encode_nulls) { kind[u] = 2; idx[u] = cap + 1; }
        else kind[u] = 3;
      }
    }
#pragma unroll
    for (int u = 0; u < kInsertRows; u++)
      if (kind[u] == 1 || kind[u] == 2) sv[u] = load_slot(&table[idx[u]]);
#pragma unroll
    for (int u = 0; u < kInsertRows; u++) {
      const int64_t i = base + (int64_t)u * kBlock;
      if (kind[u] == 0) continue;
      if (kind[u] == 3) { if (row_slot) row_slot[i] = direct_below ? 0u : kNoRow; continue; }  // masked null → index 0, now or in emit_kernel
      uint64_t s = idx[u];
      unsigned fr = sv[u].first_row;  // may be stale — only ever LARGER than the truth, so the atomicMin below stays correct
      unsigned id = sv[u].id;
      if (kind[u] == 1) {
        unsigned long long cur = sv[u].key;
        int probes = 0;
        for (;;) {
          if (cur == kEmpty) {  // settled slots skip the CAS
            cur = atomicCAS(&table[s].key, kEmpty, k[u]);
            if (cur == kEmpty) { fresh++; break; }
          }
          if (keys.same(cur, k[u], i)) break;
          s = (s + 1) & mask;
          if (++probes > kProbeLimit) { atomicExch(overflow, 1u); return; }
          const SlotView nx = load_slot(&table[s]);
          cur = nx.key;
          fr = nx.first_row;
          id = nx.id;
        }
      }
      if (fr < direct_below) {  // seen in the ranked prefix: nothing to update (i ≥ direct_below > fr)
        if (row_slot) row_slot[i] = id;
        continue;
      }
      if (fr > (unsigned)i) atomicMin(&table[s].first_row, (unsigned)i);
      if (row_slot) row_slot[i] = (unsigned)s | flag;  // flag: bit 31 when emit_kernel must tell slots from final ids
      nmiss++;
    }
  }
  if (fresh) atomicAdd(distinct, (unsigned long long)fresh);
  if (direct_below && nmiss) atomicAdd(misses, (unsigned long long)nmiss);
}

__global__ __launch_bounds__(kBlock) void mark_kernel(const Slot* __restrict__ table, uint64_t nslots,
                                                       unsigned long long* __restrict__ firsts) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (uint64_t s = (uint64_t)blockIdx.x * kBlock + threadIdx.x; s < nslots; s += stride) {
    unsigned fr = table[s].first_row;
    if (fr != kNoRow) atomicOr(&firsts[fr >> 6], 1ull << (fr & 63));
  }
}


__global__ __launch_bounds__(kBlock) void assign_kernel(Slot* __restrict__ table, uint64_t cap, const unsigned long long* __restrict__ firsts,
                                                         const unsigned* __restrict__ wordprefix, const int64_t* __restrict__ tileoff,
                                                         unsigned long long* __restrict__ dict, int* __restrict__ null_id,
                                                         long long* __restrict__ first_rows) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const uint64_t nslots = cap + 2;
  for (uint64_t s = (uint64_t)blockIdx.x * kBlock + threadIdx.x; s < nslots; s += stride) {
    unsigned fr = table[s].first_row;
    if (fr == kNoRow) continue;
    unsigned id = rank_of_row(fr, firsts, wordprefix, tileoff);
    table[s].id = id;
    if (first_rows) first_rows[id] = (long long)fr;
    if (s == cap + 1) {
      *null_id = (int)id;
      if (dict) dict[id] = 0;  // GetDictArrayData: the null slot keeps the fresh buffer's zero
    } else if (dict) {
      dict[id] = table[s].key;  // slot `cap` holds the all-ones key: key field is still kEmpty == that key
    }
  }
}

// ids[i] holds what the insert pass left there: a slot number (→ that slot's id), kNoRow (masked null →
// index 0, vector_hash.go:169-172), or — for rows ≥ direct_from, written by insert_small_kernel — either a
// final id (kept) or a slot number flagged with bit 31.
__global__ __launch_bounds__(kBlock) void emit_kernel(const Slot* __restrict__ table, int32_t* __restrict__ ids, int64_t n, int64_t direct_from) {
  constexpr int U = 4;  // gathers in flight per lane
  const int64_t stride = (int64_t)gridDim.x * kBlock * U;
  for (int64_t base = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; base < n; base += stride) {
    unsigned s[U];
    int32_t id[U];
    bool put[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + (int64_t)u * kBlock;
      s[u] = i < n ? (unsigned)ids[i] : kNoRow;
      put[u] = i < n;
      if (i >= direct_from && s[u] != kNoRow) {
        if (s[u] & 0x80000000u) s[u] &= 0x7fffffffu; else put[u] = false;
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) id[u] = s[u] == kNoRow ? 0 : (put[u] ? (int32_t)table[s[u]].id : 0);
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + (int64_t)u * kBlock;
      if (put[u]) ids[i] = id[u];
    }
  }
}

// ---- low cardinality: the prefix's keys in LDS ---------------------------------------------------
// When the 2^21-row prefix shows ≤ kSmallKeys distinct keys, their ids are already final (an id is the
// number of first occurrences before the key's own, and all of those lie in the prefix too).  The
// (key → id) pairs are copied into a kSmallSlots-entry open-addressing table that every workgroup of
// the main pass keeps in LDS: a hit costs no global access beyond the key load and the id store, so the
// pass runs at HBM speed instead of the L2 gather rate.  A miss (key not seen in the prefix) takes the
// global-table path and leaves a flagged slot number for emit_kernel.
constexpr int kSmallSlots = 8192;
constexpr int kSmallKeys = 4096;
constexpr int kSmallBlock = 1024;
constexpr int kSmallRows = 8;

__global__ __launch_bounds__(kBlock) void small_build_kernel(const Slot* __restrict__ table, uint64_t cap,
                                                              unsigned long long* __restrict__ skeys, unsigned* __restrict__ sids) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (uint64_t s = (uint64_t)blockIdx.x * kBlock + threadIdx.x; s < cap; s += stride) {
    if (table[s].first_row == kNoRow) continue;
    const unsigned long long k = table[s].key;
    unsigned j = (unsigned)hash_int(k) & (kSmallSlots - 1);
    while (atomicCAS(&skeys[j], kEmpty, k) != kEmpty) j = (j + 1) & (kSmallSlots - 1);  // keys are distinct, ≤ half the entries
    sids[j] = table[s].id;
  }
}

__global__ __launch_bounds__(kSmallBlock) void insert_small_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ valid,
                                                                    int64_t off, int64_t lo, int64_t hi, int encode_nulls, Slot* __restrict__ table,
                                                                    uint64_t cap, const unsigned long long* __restrict__ skeys,
                                                                    const unsigned* __restrict__ sids, unsigned* __restrict__ out_ids,
                                                                    unsigned long long* __restrict__ distinct, unsigned* __restrict__ overflow,
                                                                    unsigned long long* __restrict__ misses) {
  __shared__ unsigned long long l_keys[kSmallSlots];
  __shared__ unsigned l_ids[kSmallSlots];
  for (int j = threadIdx.x; j < kSmallSlots; j += kSmallBlock) { l_keys[j] = skeys[j]; l_ids[j] = sids[j]; }
  __syncthreads();
  // the two dedicated slots (all-ones key, null): final ids if the prefix saw them
  const unsigned ones_id = table[cap].first_row != kNoRow ? table[cap].id : kNoRow;
  const unsigned null_id = table[cap + 1].first_row != kNoRow ? table[cap + 1].id : kNoRow;
  const uint64_t mask = cap - 1;
  const int64_t stride = (int64_t)gridDim.x * kSmallBlock * kSmallRows;
  unsigned fresh = 0, nmiss = 0;
  // the next step's keys are requested before this step's are looked up: a lane has 2 × kSmallRows loads in flight and the
  // LDS work of one step hides behind the HBM latency of the next (one workgroup per CU: there is nobody else to hide it)
  unsigned long long kn[kSmallRows];
  bool okn[kSmallRows];
  auto fetch = [&](int64_t b) {
#pragma unroll
    for (int u = 0; u < kSmallRows; u++) {
      const int64_t i = b + (int64_t)u * kSmallBlock;
      okn[u] = i < hi && ah_bit(valid, off + i);
      kn[u] = okn[u] ? __builtin_nontemporal_load(&keys[i]) : 0ull;
    }
  };
  int64_t base = lo + (int64_t)blockIdx.x * kSmallBlock * kSmallRows + threadIdx.x;
  if (base < hi) fetch(base);
  for (; base < hi; base += stride) {
    unsigned long long k[kSmallRows];
    bool ok[kSmallRows];
#pragma unroll
    for (int u = 0; u < kSmallRows; u++) { k[u] = kn[u]; ok[u] = okn[u]; }
    if (base + stride < hi) fetch(base + stride);
    // first probes of all rows together (and the id that goes with a hit): eight independent LDS round trips, not a chain
    unsigned j0[kSmallRows], id0[kSmallRows];
    unsigned long long lk0[kSmallRows];
#pragma unroll
    for (int u = 0; u < kSmallRows; u++) {
      j0[u] = (unsigned)hash_int(k[u]) & (kSmallSlots - 1);
      lk0[u] = l_keys[j0[u]];
      id0[u] = l_ids[j0[u]];
    }
#pragma unroll
    for (int u = 0; u < kSmallRows; u++) {
      const int64_t i = base + (int64_t)u * kSmallBlock;
      if (i >= hi) break;
      unsigned r = kNoRow;       // final id, or kNoRow = "take the global path with slot s"
      uint64_t s = 0;
      bool probe = false;
      if (!ok[u]) {
        if (!encode_nulls) r = 0;  // masked null → index 0
        else { r = null_id; s = cap + 1; }
      } else if (k[u] == kEmpty) {
        r = ones_id; s = cap;
      } else if (lk0[u] == k[u]) {
        r = id0[u];
      } else if (lk0[u] == kEmpty) {
        probe = true;
      } else {
        unsigned j = (j0[u] + 1) & (kSmallSlots - 1);
        for (;;) {
          const unsigned long long lk = l_keys[j];
          if (lk == k[u]) { r = l_ids[j]; break; }
          if (lk == kEmpty) { probe = true; break; }
          j = (j + 1) & (kSmallSlots - 1);
        }
      }
      if (r == kNoRow) {
        if (probe) {
          s = hash_int(k[u]) & mask;
          int probes = 0;
          for (;;) {
            unsigned long long cur = table[s].key;
            if (cur != k[u] && cur == kEmpty) {
              cur = atomicCAS(&table[s].key, kEmpty, k[u]);
              if (cur == kEmpty) { fresh++; cur = k[u]; }
            }
            if (cur == k[u]) break;
            s = (s + 1) & mask;
            if (++probes > kProbeLimit) { atomicExch(overflow, 1u); return; }
          }
        }
        if (table[s].first_row > (unsigned)i) atomicMin(&table[s].first_row, (unsigned)i);
        r = 0x80000000u | (unsigned)s;
        nmiss++;
      }
      if (out_ids) __builtin_nontemporal_store(r, &out_ids[i]);
    }
  }
  if (fresh) atomicAdd(distinct, (unsigned long long)fresh);
  if (nmiss) atomicAdd(misses, (unsigned long long)nmiss);
}

// ---- medium cardinality: the prefix's keys in a table sized for an XCD's L2 ---------------------------------
// The insert table is sized before anything is known about the column (2^22 slots = 64 MiB): 2^16 distinct keys sit
// there one per 64-byte line, 4 MiB of hot lines scattered over 64 MiB — PMC showed the main pass fetching a line
// from the fabric for every probe (35 B/row at 2^16 keys).  Once the prefix is ranked its (key → id) pairs are
// final, so they are re-packed into a read-only table at load ≤ ½ (2^16 keys: 2 MiB, 2^17: 4 MiB — one XCD's L2;
// beyond that still half the footprint in the Infinity Cache) and the main pass probes THAT: a hit costs one
// 16-byte load that stays in L2.  Misses take the insert table like in insert_small_kernel.
struct CSlot {
  unsigned long long key;
  unsigned id;
  unsigned pad;
};
__global__ __launch_bounds__(kBlock) void compact_build_kernel(const Slot* __restrict__ table, uint64_t cap, CSlot* __restrict__ ctab, uint64_t cmask) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (uint64_t s = (uint64_t)blockIdx.x * kBlock + threadIdx.x; s < cap; s += stride) {
    const SlotView v = load_slot(&table[s]);
    if (v.first_row == kNoRow) continue;
    uint64_t j = hash_int(v.key) & cmask;
    while (atomicCAS(&ctab[j].key, kEmpty, v.key) != kEmpty) j = (j + 1) & cmask;  // keys are distinct, load ≤ ½
    ctab[j].id = v.id;
  }
}

constexpr int kCompactRows = 8;  // rows per lane per step: eight independent probes in flight
__global__ __launch_bounds__(kBlock) void insert_compact_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ valid,
                                                                 int64_t off, int64_t lo, int64_t hi, int encode_nulls, Slot* __restrict__ table,
                                                                 uint64_t cap, const CSlot* __restrict__ ctab, uint64_t cmask,
                                                                 unsigned* __restrict__ out_ids, unsigned long long* __restrict__ distinct,
                                                                 unsigned* __restrict__ overflow, unsigned long long* __restrict__ misses) {
  const unsigned ones_id = table[cap].first_row != kNoRow ? table[cap].id : kNoRow;
  const unsigned null_id = table[cap + 1].first_row != kNoRow ? table[cap + 1].id : kNoRow;
  const uint64_t mask = cap - 1;
  const int64_t stride = (int64_t)gridDim.x * kBlock * kCompactRows;
  unsigned fresh = 0, nmiss = 0;
  for (int64_t base = lo + (int64_t)blockIdx.x * kBlock * kCompactRows + threadIdx.x; base < hi; base += stride) {
    unsigned long long k[kCompactRows];
    bool ok[kCompactRows];
    uint4 cs[kCompactRows];
#pragma unroll
    for (int u = 0; u < kCompactRows; u++) {
      const int64_t i = base + (int64_t)u * kBlock;
      ok[u] = i < hi && ah_bit(valid, off + i);
      k[u] = ok[u] ? __builtin_nontemporal_load(&keys[i]) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < kCompactRows; u++) cs[u] = *reinterpret_cast<const uint4*>(&ctab[hash_int(k[u]) & cmask]);
#pragma unroll
    for (int u = 0; u < kCompactRows; u++) {
      const int64_t i = base + (int64_t)u * kBlock;
      if (i >= hi) break;
      unsigned r = kNoRow;  // final id, or kNoRow = "take the insert table with slot s"
      uint64_t s = 0;
      bool probe = false;
      if (!ok[u]) {
        if (!encode_nulls) r = 0;  // masked null → index 0
        else { r = null_id; s = cap + 1; }
      } else if (k[u] == kEmpty) {
        r = ones_id; s = cap;
      } else {
        uint64_t j = hash_int(k[u]) & cmask;
        uint4 c = cs[u];
        for (;;) {
          const unsigned long long ck = ((unsigned long long)c.y << 32) | c.x;
          if (ck == k[u]) { r = c.z; break; }
          if (ck == kEmpty) { probe = true; break; }
          j = (j + 1) & cmask;
          c = *reinterpret_cast<const uint4*>(&ctab[j]);
        }
      }
      if (r == kNoRow) {
        if (probe) {
          s = hash_int(k[u]) & mask;
          int probes = 0;
          for (;;) {
            unsigned long long cur = table[s].key;
            if (cur != k[u] && cur == kEmpty) {
              cur = atomicCAS(&table[s].key, kEmpty, k[u]);
              if (cur == kEmpty) { fresh++; cur = k[u]; }
            }
            if (cur == k[u]) break;
            s = (s + 1) & mask;
            if (++probes > kProbeLimit) { atomicExch(overflow, 1u); return; }
          }
        }
        if (table[s].first_row > (unsigned)i) atomicMin(&table[s].first_row, (unsigned)i);
        r = 0x80000000u | (unsigned)s;
        nmiss++;
      }
      if (out_ids) __builtin_nontemporal_store(r, &out_ids[i]);
    }
  }
  if (fresh) atomicAdd(distinct, (unsigned long long)fresh);
  if (nmiss) atomicAdd(misses, (unsigned long long)nmiss);
}

// Per-group accumulation.  ids are dense and in first-seen order, so a low-cardinality
// column keeps ALL its groups in a per-workgroup LDS table (kLdsGroups × {sum, count}):
// rows hit LDS atomics, and each workgroup flushes every touched group to HBM once —
// instead of two same-address global atomics per row (~12 ns each, serialised at L2),
// which is what made 2^10 groups 20× slower than 2^16 before.
constexpr int kLdsGroups = 4096;

template <typename VT, typename AT, bool USE_LDS>
__global__ __launch_bounds__(kBlock) void group_sum_kernel(const int32_t* __restrict__ ids, const VT* __restrict__ vals,
                                                            const uint8_t* __restrict__ vvalid, int64_t voff, int64_t n,
                                                            AT* __restrict__ sums, unsigned long long* __restrict__ counts, int ngroups, FxAcc fx) {
  constexpr bool kFx = std::is_same<VT, double>::value;   // doubles: 128-bit fixed point (s_sum = low words, s_hi = high words)
  __shared__ unsigned long long s_sum[USE_LDS ? kLdsGroups : 1];
  __shared__ unsigned long long s_hi[USE_LDS && kFx ? kLdsGroups : 1];
  __shared__ unsigned s_cnt[USE_LDS ? kLdsGroups : 1];
  const int nl = ngroups < kLdsGroups ? ngroups : kLdsGroups;
  if (USE_LDS) {
    for (int g = threadIdx.x; g < nl; g += kBlock) { s_sum[g] = 0; s_cnt[g] = 0; if (kFx) s_hi[g] = 0; }
    __syncthreads();
  }
  int sh = 0;
  if constexpr (kFx) sh = fx_shift(*fx.absmax);   // wide columns (fx.gmax): the group's own scale, looked up per row below
  constexpr int U = 8;  // rows per lane per step: 8 id loads + 8 value loads in flight (one row at a time is latency-bound)
  const int64_t stride = (int64_t)gridDim.x * kBlock * U;
  for (int64_t base = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; base < n; base += stride) {
    int32_t g[U];
    VT v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + (int64_t)u * kBlock;
      const bool ok = i < n && ah_bit(vvalid, voff + i);
      g[u] = ok ? __builtin_nontemporal_load(&ids[i]) : -1;
      v[u] = ok ? __builtin_nontemporal_load(&vals[i]) : (VT)0;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (g[u] < 0) continue;
      const bool lds = USE_LDS && g[u] < nl;
      if constexpr (kFx) {
        if (fx_finite(v[u])) {
          unsigned long long lo, hi;
          fx_split(v[u], fx.gmax ? fx_shift(fx.gmax[g[u]]) : sh, &lo, &hi);
          if (lds) fx_add(s_sum, s_hi, (size_t)g[u], lo, hi);
          else fx_add(fx.lo, fx.hi, (size_t)g[u], lo, hi);
        } else {
          atomicOr(&fx.flags[g[u]], fx_flag(v[u]));
        }
      } else {
        if (lds) atomicAdd(&s_sum[g[u]], (unsigned long long)v[u]);
        else atomicAdd(&sums[g[u]], (AT)v[u]);
      }
      if (lds) atomicAdd(&s_cnt[g[u]], 1u);
      else atomicAdd(&counts[g[u]], 1ull);
    }
  }
  if (USE_LDS) {
    __syncthreads();
    for (int g = threadIdx.x; g < nl; g += kBlock) {
      unsigned cnt = s_cnt[g];
      if (cnt) {
        if constexpr (kFx) fx_add(fx.lo, fx.hi, (size_t)g, s_sum[g], s_hi[g]);
        else atomicAdd(&sums[g], (AT)s_sum[g]);
        atomicAdd(&counts[g], (unsigned long long)cnt);
      }
    }
  }
}

// wide columns (ah_hashing.h): largest finite |x| per group = the group's fixed-point scale.  A look before the atomic: once a
// group's maximum has been seen (early, on average) its rows issue none.
__global__ __launch_bounds__(kBlock) void group_max_kernel(const int32_t* __restrict__ ids, const unsigned long long* __restrict__ vals,
                                                            const uint8_t* __restrict__ vvalid, int64_t voff, int64_t n,
                                                            unsigned long long* __restrict__ gmax) {
  constexpr int U = 8;
  const int64_t stride = (int64_t)gridDim.x * kBlock * U;
  for (int64_t base = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; base < n; base += stride) {
    int32_t g[U];
    unsigned long long a[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + (int64_t)u * kBlock;
      const bool ok = i < n && ah_bit(vvalid, voff + i);
      g[u] = ok ? __builtin_nontemporal_load(&ids[i]) : -1;
      a[u] = ok ? __builtin_nontemporal_load(&vals[i]) & 0x7fffffffffffffffull : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (g[u] < 0 || (a[u] >> 52) == 0x7ff || a[u] == 0) continue;
      if (__hip_atomic_load(&gmax[g[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a[u]) atomicMax(&gmax[g[u]], a[u]);
    }
  }
}

// Above 4 Ki groups LDS cannot hold all groups, and two global atomics per row run at ≈ 25 G atomics/s
// device-wide (5.4 ms for 2^26 rows, whatever the number of copies of the sums — measured).  So the
// (value, group id) pairs are first partitioned by id >> 12 with ah_sort.hip's stable radix kernels — one
// 256-way pass up to 1 Mi groups, two passes (65 536 windows) up to 256 Mi — which makes every 64 Ki-row
// chunk a short sequence of runs, each inside one 4096-group window.  A workgroup walks the runs of its
// chunk: aggregate the run in LDS, flush the groups it touched (consecutive addresses), next run.
constexpr int kBucketShift = 12;                     // log2(kLdsGroups)
constexpr int64_t kPartitionOnePass = 1 << 20;       // 256 windows of 4096 groups
constexpr int64_t kPartitionMaxGroups = 1ll << 28;   // 65 536 windows
constexpr int64_t kChunkRows = 1 << 16;
constexpr int64_t kShortRun = 1024;                  // runs shorter than this go straight to global atomics

template <typename AT>
__global__ __launch_bounds__(kBlock) void bucket_sum_kernel(const unsigned long long* __restrict__ vals, const unsigned* __restrict__ ids, int64_t n,
                                                             AT* __restrict__ sums, unsigned long long* __restrict__ counts, FxAcc fx) {
  constexpr bool kFx = std::is_same<AT, double>::value;
  __shared__ unsigned long long s_sum[kLdsGroups];
  __shared__ unsigned long long s_hi[kFx ? kLdsGroups : 1];
  __shared__ unsigned s_cnt[kLdsGroups];
  int sh = 0;
  if constexpr (kFx) sh = fx_shift(*fx.absmax);
  // one row into the LDS window (base = nullptr) or straight into the global accumulators
  auto add_row = [&](unsigned id, unsigned long long bits, bool lds) {
    const size_t g = lds ? (size_t)(id & (kLdsGroups - 1)) : (size_t)id;
    if constexpr (kFx) {
      const double x = __builtin_bit_cast(double, bits);
      if (fx_finite(x)) {
        unsigned long long lo, hi;
        fx_split(x, sh, &lo, &hi);
        if (lds) fx_add(s_sum, s_hi, g, lo, hi);
        else fx_add(fx.lo, fx.hi, g, lo, hi);
      } else {
        atomicOr(&fx.flags[id], fx_flag(x));
      }
    } else {
      if (lds) atomicAdd(&s_sum[g], bits);
      else atomicAdd(&sums[g], (AT)bits);
    }
    if (lds) atomicAdd(&s_cnt[g], 1u);
    else atomicAdd(&counts[g], 1ull);
  };
  const int64_t lo = (int64_t)blockIdx.x * kChunkRows, hi = lo + kChunkRows < n ? lo + kChunkRows : n;
  int64_t pos = lo;
  while (pos < hi) {
    const unsigned bucket = (ids[pos] & 0x7fffffffu) >> kBucketShift;
    // end of this window's run inside the chunk (rows are ordered by window): 256-ary search, every
    // thread probes one sample per round, the samples still inside the window form a prefix
    int64_t a = pos, span = hi - pos;
    while (span > 1) {
      const int64_t step = (span + kBlock - 1) / kBlock;
      const int64_t idx = a + (int64_t)threadIdx.x * step;
      const bool inside = idx < a + span && ((ids[idx] & 0x7fffffffu) >> kBucketShift) == bucket;
      const int cnt = __syncthreads_count(inside);  // ≥ 1: the sample of thread 0 is row a
      const int64_t lim = a + span;
      a += (int64_t)(cnt - 1) * step;
      span = lim - a < step ? lim - a : step;
    }
    const int64_t end = a + 1;
    if (end - pos < kShortRun) {
      for (int64_t i = pos + threadIdx.x; i < end; i += kBlock) {
        const unsigned id = ids[i];
        if (id & 0x80000000u) continue;  // null value: neither summed nor counted
        add_row(id, vals[i], false);
      }
    } else {
      for (int g = threadIdx.x; g < kLdsGroups; g += kBlock) { s_sum[g] = 0; s_cnt[g] = 0; if (kFx) s_hi[g] = 0; }
      __syncthreads();
      constexpr int U = 4;
      for (int64_t b0 = pos + threadIdx.x; b0 < end; b0 += (int64_t)kBlock * U) {
        unsigned id[U];
        unsigned long long v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int64_t i = b0 + (int64_t)u * kBlock;
          id[u] = i < end ? __builtin_nontemporal_load(&ids[i]) : 0x80000000u;
          v[u] = i < end ? __builtin_nontemporal_load(&vals[i]) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (id[u] & 0x80000000u) continue;  // null value (or past the run)
          add_row(id[u], v[u], true);
        }
      }
      __syncthreads();
      for (int g = threadIdx.x; g < kLdsGroups; g += kBlock) {
        const unsigned cnt = s_cnt[g];
        if (cnt) {
          const size_t gg = ((size_t)bucket << kBucketShift) + g;
          if constexpr (kFx) fx_add(fx.lo, fx.hi, gg, s_sum[g], s_hi[g]);
          else atomicAdd(&sums[gg], (AT)s_sum[g]);
          atomicAdd(&counts[gg], (unsigned long long)cnt);
        }
      }
      __syncthreads();
    }
    pos = end;
  }
}

static uint64_t next_pow2_u64(uint64_t x) {
  uint64_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

// distinct keys expected among n rows when a prefix of p rows held d0 (uniform-urn model)
static double estimate_distinct(double d0, double p, double n) {
  const double r = d0 / p;
  if (r >= 0.999) return n;  // (almost) every prefix row was new: cannot tell, assume one key per row
  double lo = 1e-9, hi = 64.0;  // x = p / C;  g(x) = (1 − e^(−x)) / x falls from 1 to 0
  for (int it = 0; it < 60; it++) {
    const double mid = 0.5 * (lo + hi);
    if ((1.0 - exp(-mid)) / mid > r) lo = mid; else hi = mid;
  }
  const double C = p / (0.5 * (lo + hi));
  const double d = C * (1.0 - exp(-n / C));
  return d < n ? d : n;
}

struct EncodeResult {
  int64_t ndict;
  int32_t null_id;
};

// core: ids (optional, n int32), dict (optional), returns sizes.  Device pointers.
template <typename K>
int encode_core(ah_ctx* c, const K keys, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls,
                int32_t* out_ids, uint64_t* out_dict, EncodeResult* res, int64_t* out_first_rows = nullptr, bool allow_partitioned = true) {
  res->ndict = 0;
  res->null_id = -1;
  if (n == 0) return AH_OK;
  // partition-first (ah_hash_part.hip) — forced by the option (tests, measurements); the automatic choice comes after the prefix below.
  // allow_partitioned = false: the caller's own temporaries live in the arena that path would take (the id-based group-by)
  auto try_partitioned = [&](int lp, bool* done, int slots2 = 4096) -> int {
    *done = false;
    if constexpr (K::kLdsTable) {
      int used = 0;
      int64_t nd = 0;
      int32_t nid = -1;
      int prc = lp <= 10 ? ah_encode_partitioned_try(c, (const uint64_t*)keys.keys, valid, off, n, encode_nulls, lp, c->opt_encode_part_slots == 4096 ? 4096 : 8192, out_ids, out_dict, out_first_rows, &nd, &nid, &used)
                         : ah_encode_partitioned2_try(c, (const uint64_t*)keys.keys, valid, off, n, encode_nulls, lp, slots2, out_ids, out_dict, out_first_rows, &nd, &nid, &used);
      if (prc != AH_OK) return prc;
      if (used) { res->ndict = nd; res->null_id = nid; *done = true; }
    }
    return AH_OK;
  };
  if (K::kLdsTable && allow_partitioned && c->opt_encode_partition >= 3) {
    bool done;
    int prc = try_partitioned(c->opt_encode_partition > 13 ? 13 : c->opt_encode_partition, &done);
    if (prc != AH_OK || done) return prc;
  }
  // slot numbers travel through the int32 id column: the largest table (2n slots + 2) must stay below 2^32 − 1
  if (n > ((int64_t)1 << 30)) return ah_fail(c, AH_ENOTIMPL, "hash: more than 2^30 rows per call");
  // the first look's verdict, from the distinct keys d16 among the first 2^16 rows: partitions (and how many), or *undecided (almost every
  // row was new: the 2^21-row prefix will tell), or neither (the global table).  → AH_OK with *done = true when the partitions answered.
  auto decide_after_look = [&](double d16, bool overflowed, bool* undecided, bool* done, bool* attempted) -> int {
    const double hi = (double)((int64_t)1 << 16);
    *done = false;
    *attempted = false;
    const double est = estimate_distinct(d16, hi, (double)n);
    // (almost) every one of the 2^16 rows was new: ≥ 2^25 keys or so, too few repeats to say how many — the prefix below will tell
    *undecided = !overflowed && d16 >= 0.999 * hi;
    if (overflowed || est < (double)c->opt_encode_part_min || est > 8192.0 * 4400.0) return AH_OK;
    int lp = 8;   // 256 … 1024 partitions of ≤ 4400 expected keys in one cut (8192-slot tables); beyond, two cuts into 2048 … 8192 of ≤ 2200 (4096-slot tables)
    const double kpp1 = c->opt_encode_part_slots == 4096 ? 2200.0 : 4400.0;
    while (lp < 10 && est / (double)(1 << lp) > kpp1) lp++;
    int slots2 = 4096;
    if (est / (double)(1 << lp) > kpp1) {
      lp = 11;
      while (lp < 13 && est / (double)(1 << lp) > 2200.0) lp++;
      // 21 … 36 M keys: 8192 partitions with the large tables.  (The small ones take 3072 keys; 2600 expected leaves room for the
      // estimate's ± 9 % at this cardinality — 128 repeats among the 2^16 sampled rows — and the partitions' own spread.)
      if (est / (double)(1 << lp) > 2600.0) slots2 = 8192;
    }
    *attempted = true;
    return try_partitioned(lp, done, slots2);
  };
  // Large calls look BEFORE anything is spent on the global table (ah_encode_first_look: one launch, ≈ 25 µs, against ≈ 90 µs of table
  // fill + staged inserts + polled read): a call the partitions answer never touches the table.  Small calls (n < 2^24) keep the look
  // that falls out of the staged inserts — it costs them nothing, and this one would cost a launch and a wait on every call.
  bool looked = false, look_undecided = false;
  if constexpr (K::kLdsTable) if (allow_partitioned && c->opt_encode_partition == 1 && n >= ((int64_t)1 << 24) && c->opt_encode_early_look) {
    uint64_t d16 = 0;
    int lrc = ah_encode_first_look(c, (const uint64_t*)keys.keys, valid, off, (int64_t)1 << 16, &d16);
    if (lrc != AH_OK) return lrc;
    bool done = false, attempted = false;
    lrc = decide_after_look((double)d16, false, &look_undecided, &done, &attempted);
    if (lrc != AH_OK || done) return lrc;
    looked = true;   // (a void attempt left no slot numbers behind: the table has not been touched yet)
  }
  const int64_t nwords = ah_ceil_div(n, 64);
  const int64_t ntiles = ah_ceil_div(nwords, 32);
  const uint64_t cap_max = next_pow2_u64((uint64_t)n * 2 < 64 ? 64 : (uint64_t)n * 2);
  uint64_t cap = cap_max < ((uint64_t)1 << 22) ? cap_max : ((uint64_t)1 << 22);
  unsigned long long* distinct = (unsigned long long*)&c->dscalars[4];
  unsigned* overflow = (unsigned*)&c->dscalars[5];
  unsigned long long* total = (unsigned long long*)&c->dscalars[6];
  int* null_id = (int*)&c->dscalars[7];
  unsigned long long* misses = (unsigned long long*)&c->dscalars[8];
  // tunable, for measurements: 0 = plain (ids in a separate pass), 1 = direct ids, 2 (default) = direct ids + LDS table
  const int direct_path = c->opt_hash_direct;
  bool resized = false;  // the table was re-planned: it is large and the prefix holds a minority of its keys
  for (;;) {
    // scratch: table | firsts | wordprefix | tilecnt | tileoff
    size_t table_bytes = (size_t)(cap + 2) * sizeof(Slot);
    size_t firsts_bytes = (size_t)nwords * 8;
    size_t wp_bytes = ((size_t)nwords * 4 + 7) & ~(size_t)7;
    size_t tc_bytes = ((size_t)ntiles * 4 + 7) & ~(size_t)7;
    size_t to_bytes = (size_t)ntiles * 8;
    void* scratch;
    const size_t small_bytes = (size_t)kSmallSlots * 12;
    const size_t compact_bytes = K::kLdsTable ? (size_t)(cap / 2) * sizeof(CSlot) : 0;  // ≤ cap/4 prefix keys at load ≤ ½
    int rc = ah_scratch_reserve(c, table_bytes + firsts_bytes + wp_bytes + tc_bytes + to_bytes + small_bytes + compact_bytes + 64, &scratch);
    if (rc != AH_OK) return rc;
    Slot* table = (Slot*)scratch;
    unsigned long long* firsts = (unsigned long long*)((uint8_t*)scratch + table_bytes);
    unsigned* wordprefix = (unsigned*)((uint8_t*)firsts + firsts_bytes);
    int* tilecnt = (int*)((uint8_t*)wordprefix + wp_bytes);
    int64_t* tileoff = (int64_t*)((uint8_t*)tilecnt + tc_bytes);
    unsigned long long* skeys = (unsigned long long*)((uint8_t*)tileoff + to_bytes);
    unsigned* sids = (unsigned*)(skeys + kSmallSlots);
    CSlot* ctab = (CSlot*)(((uintptr_t)(sids + kSmallSlots) + 63) & ~(uintptr_t)63);

    AH_HIP(c, hipMemsetAsync(table, 0xFF, table_bytes, c->stream));
    AH_HIP(c, hipMemsetAsync(&c->dscalars[4], 0, 3 * sizeof(uint64_t), c->stream));
    AH_HIP(c, hipMemsetAsync(null_id, 0xFF, sizeof(uint64_t), c->stream));
    // Capacity planning without a wasted pass: insert a PREFIX of the rows first and look at
    // how many distinct keys it produced.  Low-cardinality columns (the common case) just carry
    // on into the same table; a prefix that already fills a quarter of the table means the
    // column needs a table sized from the extrapolated distinct count — restart with that.
    const int64_t prefix = n < ((int64_t)1 << 21) ? n : ((int64_t)1 << 21);
    unsigned grid = ah_stream_grid(c, ah_ceil_div(n, (int64_t)kBlock * kInsertRows));  // emit_kernel: 4 rows per lane as well
    // staged: with every prefix row in flight at once against an empty table, a low-cardinality column turns
    // into millions of CAS / atomicMin on a few addresses (200 µs for 1024 keys).  A few thousand rows first,
    // then the rest mostly finds settled slots and issues no atomics at all.
    bool part_undecided = false;
    // A partition-first attempt that voids itself (a partition's table overflowed: the head of the column misled the estimate)
    // has already sent ids home — over the slot numbers this path keeps in out_ids for the rows it has inserted.  Put them
    // back: every key of those rows sits in the table, so the second insert only probes (no claim, no counter moves).
    auto restore_slot_numbers = [&](int64_t rows) -> int {
      if (!out_ids || rows <= 0) return AH_OK;
      insert_kernel<K><<<ah_stream_grid(c, ah_ceil_div(rows, (int64_t)kBlock * kInsertRows)), kBlock, 0, c->stream>>>(
          keys, valid, off, 0, rows, encode_nulls, table, cap, (unsigned*)out_ids, 0u, 0u, distinct, overflow, misses);
      AH_LAUNCH_CHECK(c);
      return AH_OK;
    };
    {
      int64_t lo = 0;
      for (int64_t hi : {(int64_t)1 << 12, (int64_t)1 << 16, prefix}) {
        if (hi > prefix) hi = prefix;
        if (hi <= lo) continue;
        unsigned g = ah_stream_grid(c, ah_ceil_div(hi - lo, (int64_t)kBlock * kInsertRows));
        insert_kernel<K><<<g, kBlock, 0, c->stream>>>(keys, valid, off, lo, hi, encode_nulls, table, cap,
                                                   (unsigned*)out_ids, 0u, 0u, distinct, overflow, misses);
        AH_LAUNCH_CHECK(c);
        lo = hi;
        if (K::kLdsTable && allow_partitioned && c->opt_encode_partition == 1 && !resized && !looked && hi == ((int64_t)1 << 16) && n >= ((int64_t)1 << 22)) {
          // A first look after 2^16 rows: ≈ 3·10^5 … 4.5·10^6 expected keys are more than the LDS / re-packed tables and the caches
          // hold — cut the rows by key hash and give every partition a table in LDS (ah_hash_part.hip; ≤ 4400 expected keys per
          // partition of 256 … 1024).  2^16 rows show 2^19 evenly drawn keys with ≈ 3900 repeats (± 62): the urn model's estimate
          // is good to a few per cent up to 2^22 keys; a column whose head misleads (sorted keys) is caught by the partition
          // pass itself (a table that overflows voids the attempt) or simply stays on this path.
          unsigned long long look[2];   // {distinct so far, overflow flag}: polled, not synchronised — this look is on every large call's path
          if ((rc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[4], 2, look)) != AH_OK) return rc;
          c->pinned[0] = look[0]; c->pinned[1] = look[1];
          bool done, attempted;
          int prc = decide_after_look((double)look[0], (unsigned)look[1] != 0, &part_undecided, &done, &attempted);
          if (prc != AH_OK || done) return prc;
          if (attempted && (prc = restore_slot_numbers(hi)) != AH_OK) return prc;
        }
      }
    }
    if (looked && !resized) part_undecided = look_undecided;
    bool restart = false, small = false, direct = false, compact = false;
    uint64_t d0 = 0;
    if (prefix < n && cap < cap_max) {
      if ((rc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[4], 2, (unsigned long long*)c->pinned)) != AH_OK) return rc;
      d0 = *(volatile uint64_t*)&c->pinned[0];
      bool ovf = *(volatile unsigned*)&c->pinned[1] != 0;
      if (part_undecided && !ovf) {   // 2^21 rows hold enough repeats to tell 2^25 keys from 2^26: up to ≈ 36 M expected keys take the two-cut path
        const double est = estimate_distinct((double)d0, (double)prefix, (double)n);
        if (est <= 8192.0 * 4400.0) {
          bool done;
          int prc = try_partitioned(13, &done, est / 8192.0 > 2600.0 ? 8192 : 4096);
          if (prc != AH_OK || done) return prc;
          if ((prc = restore_slot_numbers(prefix)) != AH_OK) return prc;
        }
      }
      if (ovf || d0 > cap / 4) {
        // extrapolate with the urn model (keys drawn uniformly from C values show d0 = C·(1 − e^(−p/C))
        // distinct ones in a prefix of p rows): solve for C, predict the distinct count of all n rows,
        // size for load ≤ ⅓.  A skewed column has more distinct keys than this predicts; the probe
        // limit then trips and the insert is redone with 16× the slots (results never depend on it).
        uint64_t want = next_pow2_u64((uint64_t)(estimate_distinct((double)d0, (double)prefix, (double)n) * 3.0) + 64);
        if (want < cap * 4) want = cap * 4;
        cap = want < cap_max ? want : cap_max;
        restart = true;
        resized = true;
      }
      direct = !restart && !ovf && !resized && direct_path;
      small = K::kLdsTable && direct && d0 <= (uint64_t)kSmallKeys && direct_path > 1;
      compact = K::kLdsTable && direct && !small && direct_path > 1 && direct_path != 3;   // 3: measurement switch, no re-packed table
    }
    if (restart) continue;
    // first-seen ranks of the used slots over the first `rows` rows: table[].id, dict, first_rows, total, null_id
    auto rank_slots = [&](int64_t rows) -> int {
      const int64_t nw = ah_ceil_div(rows, 64), nt = rank_tiles(nw);
      AH_HIP(c, hipMemsetAsync(firsts, 0, (size_t)nw * 8, c->stream));
      mark_kernel<<<ah_stream_grid(c, ah_ceil_div((int64_t)cap + 2, kBlock)), kBlock, 0, c->stream>>>(table, cap + 2, firsts);
      AH_LAUNCH_CHECK(c);
      word_prefix_kernel<<<(unsigned)ah_ceil_div(nw, kBlock), kBlock, 0, c->stream>>>(firsts, nw, wordprefix, tilecnt);
      AH_LAUNCH_CHECK(c);
      scan_kernel<<<1, 1024, 0, c->stream>>>(tilecnt, nt, tileoff, total);
      AH_LAUNCH_CHECK(c);
      assign_kernel<<<ah_stream_grid(c, ah_ceil_div((int64_t)cap + 2, kBlock)), kBlock, 0, c->stream>>>(
          table, cap, firsts, wordprefix, tileoff, (unsigned long long*)out_dict, null_id, (long long*)out_first_rows);
      AH_LAUNCH_CHECK(c);
      return AH_OK;
    };
    // Rows after the prefix: the prefix's keys are ranked first (their ids are final: an id is the number
    // of first occurrences before the key's own, all of which lie in the prefix too), so the main pass can
    // write ids directly — out of LDS when the prefix has ≤ kSmallKeys keys, else out of the slot it probes
    // anyway — and only rows with a key the prefix did not have ("misses") wait for emit_kernel.  Not done
    // after a restart: that table is large (ranking scans it) and most of its keys are still to come.
    int64_t direct_from = n;  // rows from here on hold final ids or flagged slot numbers
    bool ranked = false;      // table[].id, dict, first_rows, total are final
    if (direct) {
      if ((rc = rank_slots(prefix)) != AH_OK) return rc;
      AH_HIP(c, hipMemsetAsync(misses, 0, 8, c->stream));
      direct_from = prefix;
      int64_t lo = prefix;
      if constexpr (K::kLdsTable) if (small) {
        AH_HIP(c, hipMemsetAsync(skeys, 0xFF, (size_t)kSmallSlots * 8, c->stream));
        small_build_kernel<<<ah_stream_grid(c, ah_ceil_div((int64_t)cap, kBlock)), kBlock, 0, c->stream>>>(table, cap, skeys, sids);
        AH_LAUNCH_CHECK(c);
        // a probe segment first: a column whose later rows keep bringing new keys (sorted keys, say)
        // misses the LDS table all the time and is better served by the plain kernel
        const int64_t probe_end = n - prefix > ((int64_t)1 << 23) ? prefix + ((int64_t)1 << 22) : n;
        const unsigned sgrid = (unsigned)c->num_cu;
        insert_small_kernel<<<sgrid, kSmallBlock, 0, c->stream>>>(keys.keys, valid, off, prefix, probe_end, encode_nulls,
                                                                  table, cap, skeys, sids, (unsigned*)out_ids, distinct, overflow, misses);
        AH_LAUNCH_CHECK(c);
        lo = probe_end;
        if (probe_end < n) {
          if ((rc = ah_mailbox_read(c, misses, 1, (unsigned long long*)c->pinned)) != AH_OK) return rc;
          if (*(volatile uint64_t*)c->pinned * 8 < (uint64_t)(probe_end - prefix)) {
            insert_small_kernel<<<sgrid, kSmallBlock, 0, c->stream>>>(keys.keys, valid, off, probe_end, n, encode_nulls,
                                                                      table, cap, skeys, sids, (unsigned*)out_ids, distinct, overflow, misses);
            AH_LAUNCH_CHECK(c);
            lo = n;
          }
        }
      }
      if constexpr (K::kLdsTable) if (compact && lo < n) {
        const uint64_t ccap = next_pow2_u64(d0 * 2 < 64 ? 64 : d0 * 2);
        AH_HIP(c, hipMemsetAsync(ctab, 0xFF, (size_t)ccap * sizeof(CSlot), c->stream));
        compact_build_kernel<<<ah_stream_grid(c, ah_ceil_div((int64_t)cap, kBlock)), kBlock, 0, c->stream>>>(table, cap, ctab, ccap - 1);
        AH_LAUNCH_CHECK(c);
        // a probe segment first, as above: a miss costs a second, dependent walk (re-packed table, then the insert table), so a
        // column that keeps bringing keys the prefix did not hold (2^20 uniform keys: 14 % of the rows; Zipf tails) is
        // better served by the plain kernel, whose first probes are all in flight together — measured 1.4 → 2.1 ms otherwise
        const int64_t seg_lo = lo, probe_end = n - lo > ((int64_t)1 << 23) ? lo + ((int64_t)1 << 22) : n;
        insert_compact_kernel<<<ah_stream_grid(c, ah_ceil_div(probe_end - lo, (int64_t)kBlock * kCompactRows)), kBlock, 0, c->stream>>>(
            keys.keys, valid, off, lo, probe_end, encode_nulls, table, cap, ctab, ccap - 1, (unsigned*)out_ids, distinct, overflow, misses);
        AH_LAUNCH_CHECK(c);
        lo = probe_end;
        if (probe_end < n) {
          if ((rc = ah_mailbox_read(c, misses, 1, (unsigned long long*)c->pinned)) != AH_OK) return rc;
          if (*(volatile uint64_t*)c->pinned * 64 < (uint64_t)(probe_end - seg_lo)) {
            insert_compact_kernel<<<ah_stream_grid(c, ah_ceil_div(n - lo, (int64_t)kBlock * kCompactRows)), kBlock, 0, c->stream>>>(
                keys.keys, valid, off, lo, n, encode_nulls, table, cap, ctab, ccap - 1, (unsigned*)out_ids, distinct, overflow, misses);
            AH_LAUNCH_CHECK(c);
            lo = n;
          }
        }
      }
      if (lo < n) {
        insert_kernel<K><<<grid, kBlock, 0, c->stream>>>(keys, valid, off, lo, n, encode_nulls, table, cap,
                                                      (unsigned*)out_ids, 0x80000000u, (unsigned)prefix, distinct, overflow, misses);
        AH_LAUNCH_CHECK(c);
      }
      {   // dscalars[4 … 8] = distinct, overflow, total, null id, misses: one polled read
        unsigned long long w5[5];
        if ((rc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[4], 5, w5)) != AH_OK) return rc;
        c->pinned[0] = w5[4]; c->pinned[1] = w5[1];
      }
      ranked = *(volatile uint64_t*)&c->pinned[0] == 0 && *(volatile unsigned*)&c->pinned[1] == 0;
      *(volatile uint64_t*)c->pinned = *(volatile uint64_t*)&c->pinned[1];  // overflow flag where the check below reads it
    } else {
      if (prefix < n) {
        insert_kernel<K><<<grid, kBlock, 0, c->stream>>>(keys, valid, off, prefix, n, encode_nulls, table, cap,
                                                      (unsigned*)out_ids, 0u, 0u, distinct, overflow, misses);
        AH_LAUNCH_CHECK(c);
      }
      if ((rc = ah_mailbox_read(c, (const unsigned long long*)overflow, 1, (unsigned long long*)c->pinned)) != AH_OK) return rc;
    }
    if (*(volatile unsigned*)c->pinned) {
      if (cap >= cap_max) return ah_fail(c, AH_EINVALID, "hash: table overflow at maximum capacity (internal error)");
      cap = cap * 16 < cap_max ? cap * 16 : cap_max;
      resized = true;
      continue;
    }
    if (!ranked && (rc = rank_slots(n)) != AH_OK) return rc;
    if (out_ids) {
      // no miss: only the prefix rows still hold slot numbers
      const int64_t rows = ranked ? prefix : n;
      emit_kernel<<<ah_stream_grid(c, ah_ceil_div(rows, (int64_t)kBlock * 4)), kBlock, 0, c->stream>>>(table, out_ids, rows, direct_from);
      AH_LAUNCH_CHECK(c);
    }
    if ((rc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[6], 2, (unsigned long long*)c->pinned)) != AH_OK) return rc;
    res->ndict = (int64_t) * (volatile uint64_t*)&c->pinned[0];
    res->null_id = *(volatile int32_t*)&c->pinned[1];
    return AH_OK;
  }
}

template <typename VT, typename AT>
int hash_sum(ah_ctx* c, const uint64_t* keys, const uint8_t* kvalid, int64_t koff, const VT* vals, const uint8_t* vvalid,
             int64_t voff, int64_t n, uint64_t* out_keys, AT* out_sums, int64_t* out_counts, int64_t* out_first_rows,
             int64_t* out_ngroups_host, int32_t* out_null_group_host) {
  if (n < 0 || koff < 0 || voff < 0) return ah_fail(c, AH_EINVALID, "hash_sum: negative length/offset");
  if (out_ngroups_host) *out_ngroups_host = 0;
  if (out_null_group_host) *out_null_group_host = -1;
  if (n == 0) return AH_OK;
  if (!keys || !vals || !out_keys || !out_sums || !out_counts) return ah_fail(c, AH_EINVALID, "hash_sum: null buffer");
  if (sizeof(VT) == 8) {
    // large inputs with up to ~10^6 groups: cut the rows by key hash first, aggregate each partition in LDS (ah_groupby.hip)
    int used = 0;
    int64_t ng = 0;
    int32_t nullg = -1;
    int prc = ah_groupby_partitioned_try(c, std::is_same<VT, double>::value ? 1 : 0, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums,
                                         out_counts, out_first_rows, &ng, &nullg, &used);
    if (prc != AH_OK) return prc;
    if (used) {
      if (out_ngroups_host) *out_ngroups_host = ng;
      if (out_null_group_host) *out_null_group_host = nullg;
      return AH_OK;
    }
  }
  // temporaries: a dense group id per row and, above 4096 groups, the partitioned (value, id) pairs with their
  // histograms — one reservation in the context's temp arena, sized for the two-pass partition, reused by the next call
  const int64_t nb = ah_ceil_div(n, 2048);
  const size_t pv = (size_t)n * 8, pi = (((size_t)n * 4) + 255) & ~(size_t)255, ph = (size_t)256 * nb * 4;
  constexpr bool kFx = std::is_same<VT, double>::value;
  // doubles: 128-bit fixed-point accumulators + flag word per group (≤ n + 1 groups) and the absmax word
  const size_t fxw = kFx ? (((size_t)(n + 1) * 8) + 255) & ~(size_t)255 : 0, fxf = kFx ? (((size_t)(n + 1) * 4) + 255) & ~(size_t)255 : 0;
  void* arena = nullptr;
  int rc = ah_temp_reserve(c, pi + 2 * (pv + pi) + 2 * ph + 256 + 2 * fxw + fxf + 256, &arena);
  if (rc != AH_OK) return rc;
  int32_t* ids = (int32_t*)arena;
  uint8_t* part = (uint8_t*)arena + pi;
  uint8_t* fxbase = part + 2 * (pv + pi) + 2 * ph + 256;
  FxAcc fx{nullptr, nullptr, nullptr, nullptr, nullptr};
  if (kFx) {
    fx = FxAcc{(unsigned long long*)fxbase, (unsigned long long*)(fxbase + fxw), (unsigned*)(fxbase + 2 * fxw),
               (const unsigned long long*)(fxbase + 2 * fxw + fxf), nullptr};
    AH_HIP(c, hipMemsetAsync((void*)fx.absmax, 0, 16, c->stream));
    absmax_kernel<<<ah_stream_grid(c, ah_ceil_div(n, (int64_t)kBlock * 8), 2), kBlock, 0, c->stream>>>((const unsigned long long*)vals, vvalid, voff, n,
                                                                                                        (unsigned long long*)fx.absmax);
    AH_LAUNCH_CHECK(c);
  }
  EncodeResult res;
  rc = encode_core(c, U64Keys{(const unsigned long long*)keys}, kvalid, koff, n, /*encode_nulls=*/1, ids, out_keys, &res, out_first_rows, /*allow_partitioned=*/false);
  if (rc == AH_OK) {
    hipError_t e1 = hipMemsetAsync(out_sums, 0, (size_t)res.ndict * sizeof(AT), c->stream);
    hipError_t e2 = hipMemsetAsync(out_counts, 0, (size_t)res.ndict * sizeof(int64_t), c->stream);
    if (kFx && e1 == hipSuccess) e1 = hipMemsetAsync(fx.lo, 0, (size_t)res.ndict * 8, c->stream);
    if (kFx && e1 == hipSuccess) e1 = hipMemsetAsync(fx.hi, 0, (size_t)res.ndict * 8, c->stream);
    if (kFx && e1 == hipSuccess) e1 = hipMemsetAsync(fx.flags, 0, (size_t)res.ndict * 4, c->stream);
    if (e1 != hipSuccess || e2 != hipSuccess) rc = ah_fail(c, AH_EHIP, "hash_sum: memset failed");
  }
  bool wide = false;
  if (kFx && rc == AH_OK) {
    // one scale for the call, or one per group?  (ah_hashing.h: a column spanning more than 42 binades)
    hipError_t e = hipMemcpyAsync(&c->pinned[2], fx.absmax, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = ah_fail(c, AH_EHIP, "hash_sum: range read-back failed");
    else wide = fx_wide(*(volatile uint64_t*)&c->pinned[2], *(volatile uint64_t*)&c->pinned[3]);
    if (rc == AH_OK && wide) {
      unsigned long long* gmax = (unsigned long long*)part;   // the partition temporaries are idle on this route
      if (hipMemsetAsync(gmax, 0, (size_t)res.ndict * 8, c->stream) != hipSuccess) rc = ah_fail(c, AH_EHIP, "hash_sum: memset failed");
      if (rc == AH_OK) {
        group_max_kernel<<<ah_stream_grid(c, ah_ceil_div(n, (int64_t)kBlock * 8)), kBlock, 0, c->stream>>>(ids, (const unsigned long long*)vals, vvalid, voff, n, gmax);
        if (hipGetLastError() != hipSuccess) rc = ah_fail(c, AH_EHIP, "hash_sum: launch failed");
      }
      fx.gmax = gmax;
    }
  }
  if (rc == AH_OK) {
    static const int partition_path = getenv("ARROWHIP_HASH_PARTITION") ? atoi(getenv("ARROWHIP_HASH_PARTITION")) : 1;
    if (res.ndict <= kLdsGroups) {
      unsigned grid = ah_stream_grid(c, ah_ceil_div(n, (int64_t)kBlock * 8), /*default_bpc=*/2);
      group_sum_kernel<VT, AT, true><<<grid, kBlock, 0, c->stream>>>(ids, vals, vvalid, voff, n, out_sums,
                                                                     (unsigned long long*)out_counts, (int)res.ndict, fx);
    } else if (partition_path && !wide && res.ndict <= kPartitionMaxGroups && sizeof(VT) == 8) {
      const int passes = res.ndict <= kPartitionOnePass ? 1 : 2;
      unsigned long long* pvals = (unsigned long long*)part;
      unsigned* pids = (unsigned*)(part + pv);
      unsigned* hist = (unsigned*)(part + pv + pi);
      unsigned* offs = (unsigned*)(part + pv + pi + ph);
      unsigned long long* avals = passes == 2 ? (unsigned long long*)(part + pv + pi + 2 * ph) : nullptr;
      unsigned* aids = passes == 2 ? (unsigned*)((uint8_t*)avals + pv) : nullptr;
      rc = ah_partition_by_group(c, ids, (const unsigned long long*)vals, vvalid, voff, n, kBucketShift, passes, hist, offs, avals, aids, pvals, pids);
      if (rc == AH_OK) {
        bucket_sum_kernel<AT><<<(unsigned)ah_ceil_div(n, kChunkRows), kBlock, 0, c->stream>>>(pvals, pids, n, out_sums, (unsigned long long*)out_counts, fx);
        if (hipGetLastError() != hipSuccess) rc = ah_fail(c, AH_EHIP, "hash_sum: launch failed");
      }
    } else {
      unsigned grid = ah_stream_grid(c, ah_ceil_div(n, (int64_t)kBlock * 8));
      group_sum_kernel<VT, AT, false><<<grid, kBlock, 0, c->stream>>>(ids, vals, vvalid, voff, n, out_sums,
                                                                      (unsigned long long*)out_counts, (int)(res.ndict > 0x7fffffff ? 0x7fffffff : res.ndict), fx);
    }
    if (hipGetLastError() != hipSuccess) rc = ah_fail(c, AH_EHIP, "hash_sum: launch failed");
    if constexpr (kFx) {
      if (rc == AH_OK && res.ndict > 0) {
        fx_finalize_kernel<<<(unsigned)ah_ceil_div(res.ndict, kBlock), kBlock, 0, c->stream>>>(fx, res.ndict, (double*)out_sums);
        if (hipGetLastError() != hipSuccess) rc = ah_fail(c, AH_EHIP, "hash_sum: launch failed");
      }
    }
  }
  if (rc != AH_OK) return rc;
  if (out_ngroups_host) *out_ngroups_host = res.ndict;
  if (out_null_group_host) *out_null_group_host = res.null_id;
  return AH_OK;
}

}  // namespace

static int ids_validity(ah_ctx* c, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls, uint8_t* out_ids_valid) {
  // indices validity: all set when nulls are encoded (or there is no validity);
  // otherwise the input validity (NullEncodingMask, vector_hash.go:224-230)
  // the bitmap is this call's to define up to its last byte: the bits behind row n − 1 are zero (ah_copy_bitmap and ah_set_bits_to keep
  // what lies outside their range — right for an executor's slice, wrong for a fresh output whose memory held something else before)
  int rc;
  AH_HIP(c, hipMemsetAsync(out_ids_valid, 0, (size_t)((n + 7) / 8), c->stream));
  if (valid && !encode_nulls) rc = ah_copy_bitmap(c, valid, off, n, out_ids_valid, 0, 0);
  else rc = ah_set_bits_to(c, out_ids_valid, 0, n, 1);
  if (rc != AH_OK) return rc;
  AH_HIP(c, hipStreamSynchronize(c->stream));
  return AH_OK;
}

AH_EXPORT int ah_hash_binary_encode(ah_ctx* c, int offset_width, const void* offsets, const uint8_t* data, const uint8_t* valid, int64_t off,
                                    int64_t n, int encode_nulls, int32_t* out_ids, uint8_t* out_ids_valid, int64_t* out_first_rows,
                                    int64_t* out_ndict_host, int32_t* out_null_id_host) {
  AH_ENTER(c);
  if (n < 0 || off < 0) return ah_fail(c, AH_EINVALID, "hash: negative length/offset");
  if (offset_width != 4 && offset_width != 8) return ah_fail(c, AH_EINVALID, "hash: offset width must be 4 or 8");
  if (out_ndict_host) *out_ndict_host = 0;
  if (out_null_id_host) *out_null_id_host = -1;
  if (n == 0) return AH_OK;
  if (!offsets || !out_first_rows) return ah_fail(c, AH_EINVALID, "hash: null buffer");
  if ((uintptr_t)offsets & (uintptr_t)(offset_width - 1)) return ah_fail(c, AH_EINVALID, "hash: offsets not element-aligned");
  EncodeResult res;
  int rc = offset_width == 4
               ? encode_core(c, BinKeys<int32_t>{(const int32_t*)offsets + off, data}, valid, off, n, encode_nulls, out_ids, nullptr, &res, out_first_rows)
               : encode_core(c, BinKeys<int64_t>{(const int64_t*)offsets + off, data}, valid, off, n, encode_nulls, out_ids, nullptr, &res, out_first_rows);
  if (rc != AH_OK) return rc;
  if (out_ids_valid && (rc = ids_validity(c, valid, off, n, encode_nulls, out_ids_valid)) != AH_OK) return rc;
  if (out_ndict_host) *out_ndict_host = res.ndict;
  if (out_null_id_host) *out_null_id_host = res.null_id;
  return AH_OK;
}

AH_EXPORT int ah_hash_fixed_encode(ah_ctx* c, int byte_width, const uint8_t* data, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls,
                                   int32_t* out_ids, uint8_t* out_ids_valid, int64_t* out_first_rows, uint8_t* out_dict, int64_t* out_ndict_host,
                                   int32_t* out_null_id_host) {
  AH_ENTER(c);
  if (n < 0 || off < 0) return ah_fail(c, AH_EINVALID, "hash: negative length/offset");
  if (byte_width < 1 || byte_width > 4096) return ah_fail(c, AH_EINVALID, "hash: fixed width must be 1..4096 bytes");
  if (out_ndict_host) *out_ndict_host = 0;
  if (out_null_id_host) *out_null_id_host = -1;
  if (n == 0) return AH_OK;
  if (!data || !out_first_rows) return ah_fail(c, AH_EINVALID, "hash: null buffer");
  EncodeResult res;
  int rc = encode_core(c, FixKeys{data + off * (int64_t)byte_width, byte_width}, valid, off, n, encode_nulls, out_ids, nullptr, &res, out_first_rows);
  if (rc != AH_OK) return rc;
  if (out_dict && res.ndict > 0) {
    fixed_dict_kernel<<<ah_stream_grid(c, ah_ceil_div(res.ndict * byte_width, kBlock), 8), kBlock, 0, c->stream>>>(data + off * (int64_t)byte_width, byte_width,
                                                                                                                    (const long long*)out_first_rows, res.ndict, res.null_id, out_dict);
    AH_LAUNCH_CHECK(c);
    AH_HIP(c, hipStreamSynchronize(c->stream));
  }
  if (out_ids_valid && (rc = ids_validity(c, valid, off, n, encode_nulls, out_ids_valid)) != AH_OK) return rc;
  if (out_ndict_host) *out_ndict_host = res.ndict;
  if (out_null_id_host) *out_null_id_host = res.null_id;
  return AH_OK;
}

AH_EXPORT int ah_hash_u64_encode(ah_ctx* c, const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t n,
                                 int encode_nulls, int32_t* out_ids, uint8_t* out_ids_valid, uint64_t* out_dict,
                                 int64_t* out_ndict_host, int32_t* out_null_id_host) {
  AH_ENTER(c);
  if (n < 0 || off < 0) return ah_fail(c, AH_EINVALID, "hash: negative length/offset");
  if (out_ndict_host) *out_ndict_host = 0;
  if (out_null_id_host) *out_null_id_host = -1;
  if (n == 0) return AH_OK;
  if (!keys) return ah_fail(c, AH_EINVALID, "hash: null keys");
  EncodeResult res;
  int rc = encode_core(c, U64Keys{(const unsigned long long*)keys}, valid, off, n, encode_nulls, out_ids, out_dict, &res);
  if (rc != AH_OK) return rc;
  if (out_ids_valid && (rc = ids_validity(c, valid, off, n, encode_nulls, out_ids_valid)) != AH_OK) return rc;
  if (out_ndict_host) *out_ndict_host = res.ndict;
  if (out_null_id_host) *out_null_id_host = res.null_id;
  return AH_OK;
}

AH_EXPORT int ah_hash_sum_f64(ah_ctx* c, const uint64_t* keys, const uint8_t* kvalid, int64_t koff,
                              const double* vals, const uint8_t* vvalid, int64_t voff, int64_t n,
                              uint64_t* out_keys, double* out_sums, int64_t* out_counts, int64_t* out_first_rows,
                              int64_t* out_ngroups_host, int32_t* out_null_group_host) {
  AH_ENTER(c);
  return hash_sum<double, double>(c, keys, kvalid, koff, vals, vvalid, voff, n, out_keys, out_sums, out_counts, out_first_rows,
                                  out_ngroups_host, out_null_group_host);
}

AH_EXPORT int ah_hash_sum_i64(ah_ctx* c, const uint64_t* keys, const uint8_t* kvalid, int64_t koff,
                              const int64_t* vals, const uint8_t* vvalid, int64_t voff, int64_t n,
                              uint64_t* out_keys, int64_t* out_sums, int64_t* out_counts, int64_t* out_first_rows,
                              int64_t* out_ngroups_host, int32_t* out_null_group_host) {
  AH_ENTER(c);
  return hash_sum<unsigned long long, unsigned long long>(c, keys, kvalid, koff, (const unsigned long long*)vals, vvalid, voff, n,
                                                          out_keys, (unsigned long long*)out_sums, out_counts, out_first_rows,
                                                          out_ngroups_host, out_null_group_host);
}

// ---- key → owner partition for the multi-GPU merge (SURVEY.md §8e plan A) ----------------------
// owner = (hashInt(key) >> 40) mod nparts, hashInt as in internal/hashing/hash_funcs.go:60-67: every
// rank computes the same owner for a key, so partial groups of one key all meet on one GPU.
namespace {
__global__ __launch_bounds__(kBlock) void partition_kernel(const unsigned long long* __restrict__ keys, int64_t n, unsigned nparts,
                                                            int32_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) out[i] = (int32_t)((hash_int(keys[i]) >> 40) % nparts);
}
}  // namespace

AH_EXPORT int ah_hash_partition_u64(ah_ctx* c, const uint64_t* keys, int64_t n, int nparts, int32_t* out_part) {
  AH_ENTER(c);
  if (n < 0 || nparts < 1) return ah_fail(c, AH_EINVALID, "hash_partition: bad length / partition count");
  if (n == 0) return AH_OK;
  if (!keys || !out_part) return ah_fail(c, AH_EINVALID, "hash_partition: null buffer");
  partition_kernel<<<ah_stream_grid(c, ah_ceil_div(n, kBlock), 8), kBlock, 0, c->stream>>>((const unsigned long long*)keys, n, (unsigned)nparts, out_part);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}
