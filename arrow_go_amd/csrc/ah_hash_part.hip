// ah_hash_part.hip — partition-first unique / dictionary_encode for 8-byte keys: the path for the cardinalities where the one
// global table of ah_hash.hip is neither in LDS nor in L2 (≈ 4·10^5 … 5·10^6 distinct keys).
//
// Replaces (same results, other order of work): doAppendNumeric[uint64] over hashing.Table[uint64].InsertOrGet
// (kernels/vector_hash.go:359-385, internal/hashing/xxh3_memo_table_types.go:283-294), dictionaryEncodeAction
// (vector_hash.go:145-241) and uniqueFinalize (:721-741): ids = order of first occurrence, the null key (if encoded) takes the
// id at which it is first seen (:231-238), a masked null row gets index 0 (vector_hash.go:169-172).
//
// Why.  With 2^20 keys the global table is 32 MiB: every row's probe is a random 64-byte line from the Infinity Cache or HBM,
// and the ids come from a second walk of the same table — PMC showed 92 B/row at 2^20 keys and 237 at 2^24 for 12 algorithmic
// (profiles/r02_pmc_by_workload.json), all of it random lines at the ≈ 50 G/s this chip serves them.  The group-by of
// ah_groupby.hip already avoids that by cutting the rows by key hash first; this file does the same for the encode, whose
// extra difficulty is that it owes an id to every ROW, in row order:
//
//   1 hist, offsets     per (4096-row tile, partition) counts → where every tile's run of every partition starts   (ah_partition.h, ah_bins.h)
//   2 scatter           {key, row | null flag} staged in partition order in LDS, written as runs (20 B/row moved)
//   3 table             ONE workgroup per partition: its keys in an 8192-slot open-addressing table in LDS {key, first row};
//                       every record learns its SLOT (2 bytes, partition order); the table and the first-row bits leave
//   4 rank              first rows → n-bit bitmap → prefix popcount = the sequential memo index (the argument of ah_hash.hip);
//                       every used slot gets its id, dict[id] = key
//   5 resolve           one workgroup per partition again, the partition's slot → id table in LDS: slot numbers → ids, still
//                       in partition order (6 B/row)
//   6 unpermute         per tile: its runs are read back (id + row), placed at row-in-tile in LDS, 4096 ids leave coalesced
//                       (the way the binned Take sends gathered values home, ah_take_binned.hip)
// `unique` (no ids wanted) stops after step 4.  Everything is streaming or ≥ 64-byte runs: 8 + 20 + 14 + 6 + 12 = 60 B/row.
// Nothing depends on timing: a slot number is arbitrary, an id is a function of first rows only — the bytes are those of the
// global-table path (tests/test_gpu_parity.py::test_hash_encode_partitioned).
//
// One workgroup per partition needs partitions of similar size: a key that owns a large share of the rows (Zipf) makes one
// partition many times the others — seen on the host right after the offsets pass, and such columns stay on the other path.
#include <vector>
#include "ah_common.h"
#include "ah_hashing.h"
#include "ah_bins.h"
#include "ah_partition.h"
#include "ah_msd.h"

namespace {

// LDS table of one partition: S slots (open addressing, aligned groups of 4) + the all-ones key (slot S) and the null key (S + 1);
// ¾ S keys admitted — beyond that the attempt is void (the estimate was far off); a partition's slice of the global copies is
// S + 8 entries.  S = 8192 (96 KiB: one workgroup per CU) for the one-cut path; S = 4096 for the two-cut path, whose partitions
// hold only ≈ 8 Ki rows: three workgroups per CU hide each other's LDS round trips, and a table costs half as much to set up and dump.
constexpr int kESlots = 8192, kESlots2 = 4096;
constexpr unsigned short kMaskedSlot = 0xFFFFu;   // a null row whose nulls are not encoded: index 0, no dictionary entry

template <int S>
__device__ __forceinline__ unsigned enc_group(unsigned long long key) {   // first slot group; inside a partition the keys agree in the top bits of gb_mix
  return ((((unsigned)key * 0x9E3779B1u) ^ ((unsigned)(key >> 32) * 0x85EBCA6Bu)) >> 18) & (unsigned)(S - 4);
}

// ---- 3: one workgroup per partition ------------------------------------------------------------------------------------------
template <int S, bool BATCH>
__global__ __launch_bounds__(kThreads) void enc_table_kernel(const unsigned long long* __restrict__ pkeys, const unsigned* __restrict__ prows,
                                                              const unsigned* __restrict__ binstart, int encode_nulls,
                                                              unsigned long long* __restrict__ tab_key, unsigned* __restrict__ tab_first,
                                                              unsigned short* __restrict__ rec_slot, unsigned long long* __restrict__ firsts,
                                                              unsigned* __restrict__ overflow, uint8_t* __restrict__ fbytes = nullptr) {
  constexpr int kLS = S + 2, kStride = S + 8, kSoft = S / 4 * 3;
  __shared__ __attribute__((aligned(16))) unsigned long long l_key[kLS];
  __shared__ unsigned l_first[kLS];
  __shared__ unsigned s_used;
  const int t = threadIdx.x, part = blockIdx.x;
  const int64_t r0 = binstart[part], r1 = binstart[part + 1];
  for (int j = t; j < kLS; j += kThreads) { l_key[j] = kEmpty; l_first[j] = kNoRow; }
  if (t == 0) s_used = 0;
  __syncthreads();
  const unsigned lkey_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned long long*)l_key;   // LDS byte address of the key table
  // find or claim the slot of `key` (−1: the table is full)
  auto slot_of = [&](unsigned long long key) -> int {
    unsigned g = enc_group<S>(key);
    for (;;) {
      // two ds_read_b128, spelled out (the compiler would split them into ds_read2_b64: half the banks per access, ah_groupby.hip)
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      u64x2 a, c;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(a), "=&v"(c) : "v"(lkey_base + g * 8u) : "memory");
      const bool h0 = a.x == key, h1 = a.y == key, h2 = c.x == key, h3 = c.y == key;
      const bool e0 = a.x == kEmpty, e1 = a.y == kEmpty, e2 = c.x == kEmpty, e3 = c.y == kEmpty;
      const bool y0 = h0 || e0, y1 = h1 || e1, y2 = h2 || e2, y3 = h3 || e3;
      if (!(y0 || y1 || y2 || y3)) { g = (g + 4) & (S - 1); continue; }
      const int j = (int)g + (y0 ? 0 : y1 ? 1 : y2 ? 2 : 3);
      if (h0 || (!e0 && (h1 || (!e1 && (h2 || (!e2 && h3)))))) return j;   // the key sits in front of the first empty slot
      if (atomicAdd(&s_used, 1u) >= (unsigned)kSoft) return -1;             // tickets are never returned: "full" sticks
      const unsigned long long cur = atomicCAS(&l_key[j], kEmpty, key);
      if (cur == kEmpty || cur == key) return j;
      // another key took it meanwhile: look at the group again
    }
  };
  constexpr int U = 4;
  constexpr int64_t kStep = (int64_t)kThreads * U;
  unsigned long long nk[U];
  unsigned nrw[U];
  auto load_step = [&](int64_t b) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = b + u * kThreads + t;
      const bool in = i < r1;
      nk[u] = in ? __builtin_nontemporal_load(&pkeys[i]) : 0ull;
      nrw[u] = in ? __builtin_nontemporal_load(&prows[i]) : 0u;
    }
  };
  bool full = false;
  // the lane's previous row: a key that owns much of the partition would put every lane on one slot group
  unsigned long long p_key = 0;
  int p_slot = -2;
  if (r0 < r1) load_step(r0);
  __builtin_amdgcn_s_waitcnt(0);   // enter the loop with nothing pending: see gb_aggregate_kernel on the compiler's wait counts
  for (int64_t b = r0; b < r1; b += kStep) {
    unsigned long long k[U];
    unsigned rw[U];
#pragma unroll
    for (int u = 0; u < U; u++) { k[u] = nk[u]; rw[u] = nrw[u]; }
    if (b + kStep < r1) load_step(b + kStep);
    if constexpr (BATCH) {
    // Most records find their key where the table already has it (a partition holds ~4096 keys and 64 × as many records), so the
    // U records' first slot groups are read TOGETHER — 2 U ds_read_b128 in flight, one wait — and only a record whose key is not
    // in front of the first empty slot of its group (a new key, or a probe sequence that goes on) takes the serial find-or-claim
    // loop.  One record at a time, the pass spent half its wave cycles parked on LDS round trips (4 waves per SIMD: a partition
    // is one workgroup, 256 partitions one workgroup per CU).
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    unsigned g[U];
    u64x2 qa[U], qc[U];
#pragma unroll
    for (int u = 0; u < U; u++) g[u] = enc_group<S>(k[u]);
    static_assert(U == 4, "the batched probe below is written for four records");
    asm volatile(
        "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\t"
        "ds_read_b128 %2, %9\n\tds_read_b128 %3, %9 offset:16\n\t"
        "ds_read_b128 %4, %10\n\tds_read_b128 %5, %10 offset:16\n\t"
        "ds_read_b128 %6, %11\n\tds_read_b128 %7, %11 offset:16\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(qa[0]), "=&v"(qc[0]), "=&v"(qa[1]), "=&v"(qc[1]), "=&v"(qa[2]), "=&v"(qc[2]), "=&v"(qa[3]), "=&v"(qc[3])
        : "v"(lkey_base + g[0] * 8u), "v"(lkey_base + g[1] * 8u), "v"(lkey_base + g[2] * 8u), "v"(lkey_base + g[3] * 8u)
        : "memory");
    int js[U];
    unsigned rows[U];
    unsigned pend = 0;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = b + u * kThreads + t;
      rows[u] = rw[u] & kRowMask;
      int j = -3;   // −3: not a record of this partition
      if (i < r1) {
        const unsigned long long key = k[u];
        if (rw[u] & kKeyNull) j = encode_nulls ? S + 1 : -2;
        else if (key == kEmpty) j = S;
        else if (p_slot >= 0 && p_key == key) j = p_slot;
        else {
          const bool h0 = qa[u].x == key, h1 = qa[u].y == key, h2 = qc[u].x == key, h3 = qc[u].y == key;
          const bool e0 = qa[u].x == kEmpty, e1 = qa[u].y == kEmpty, e2 = qc[u].x == kEmpty;
          if (h0 || (!e0 && (h1 || (!e1 && (h2 || (!e2 && h3)))))) j = (int)g[u] + (h0 ? 0 : h1 ? 1 : h2 ? 2 : 3);
          else { j = -4; pend |= 1u << u; }   // −4: to the find-or-claim loop below
          p_key = key; p_slot = j;            // (a pending slot is negative: the next record does not reuse it)
        }
      }
      js[u] = j;
    }
    // the stragglers of all U records in ONE divergent phase: some lane of a wave always has one (a twentieth of the records × 64
    // lanes), so a per-record fallback would run the serial loop for every record after all
    while (pend) {
      const int u = __builtin_ctz(pend);
      pend &= pend - 1;
      const unsigned long long key = u == 0 ? k[0] : u == 1 ? k[1] : u == 2 ? k[2] : k[3];
      const int j = slot_of(key);
      if (u == 0) js[0] = j; else if (u == 1) js[1] = j; else if (u == 2) js[2] = j; else js[3] = j;
      if (key == p_key) p_slot = j;
    }
    unsigned fr[U];
#pragma unroll
    for (int u = 0; u < U; u++) fr[u] = js[u] >= 0 ? l_first[js[u]] : 0u;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = b + u * kThreads + t;
      const int j = js[u];
      if (j >= 0) {
        if (fr[u] > rows[u]) atomicMin(&l_first[j], rows[u]);   // rows of a key arrive mostly in ascending order: a read is half an atomic
      } else if (j == -1) {
        full = true;
      }
      if (rec_slot && j != -3) rec_slot[i] = j >= 0 ? (unsigned short)j : kMaskedSlot;
    }
    } else {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = b + u * kThreads + t;
      if (i >= r1) continue;
      const unsigned row = rw[u] & kRowMask;
      int j;
      if (rw[u] & kKeyNull) j = encode_nulls ? S + 1 : -2;
      else if (k[u] == kEmpty) j = S;
      else if (p_slot >= 0 && p_key == k[u]) j = p_slot;
      else { j = slot_of(k[u]); p_key = k[u]; p_slot = j; }
      if (j >= 0) {
        if (l_first[j] > row) atomicMin(&l_first[j], row);   // rows of a key arrive mostly in ascending order: a read is half an atomic
      } else if (j == -1) {
        full = true;
      }
      if (rec_slot) rec_slot[i] = j >= 0 ? (unsigned short)j : kMaskedSlot;
    }
    }
  }
  if (full) atomicExch(overflow, 1u);
  __syncthreads();
  const int64_t gbase = (int64_t)part * kStride;
  for (int j = t; j < kStride; j += kThreads) {
    const unsigned fr = j < kLS ? l_first[j] : kNoRow;
    if (j < kLS && tab_key) tab_key[gbase + j] = l_key[j];   // (not wanted when the dictionary is compacted from the column)
    tab_first[gbase + j] = fr;
    if (fr != kNoRow) {
      if (fbytes) fbytes[fr] = 1;   // a plain byte store into the byte map (enc_bytes_to_bits_kernel packs it): no device-scope read-modify-write per key
      else atomicOr(&firsts[fr >> 6], 1ull << (fr & 63));
    }
  }
}

// byte map (one byte per row, 1 = a first occurrence) → the bitmap: a lane packs 64 bytes into one word
__global__ __launch_bounds__(kBlock) void enc_bytes_to_bits_kernel(const uint8_t* __restrict__ fbytes, int64_t nwords, unsigned long long* __restrict__ firsts) {
  const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (w >= nwords) return;
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  const u64x2* src = reinterpret_cast<const u64x2*>(fbytes + w * 64);
  unsigned long long bits = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const u64x2 v = __builtin_nontemporal_load(src + q);
#pragma unroll
    for (int h = 0; h < 2; h++) {
      // eight bytes of 0 / 1 → eight bits: the bytes' low bits gathered by a multiply (0x0102040810204080 puts byte k's bit at position 56 + k)
      const unsigned long long x = (h ? v.y : v.x) & 0x0101010101010101ull;
      bits |= ((x * 0x0102040810204080ull) >> 56) << (q * 16 + h * 8);
    }
  }
  firsts[w] = bits;
}

// ---- 4: ids of the used slots (tab_first is overwritten with them), the dictionary -----------------------------------------------
__global__ __launch_bounds__(kBlock) void enc_assign_kernel(const unsigned long long* __restrict__ tab_key, unsigned* __restrict__ tab_first, int64_t nslots,
                                                             const unsigned long long* __restrict__ firsts, const unsigned* __restrict__ wordprefix,
                                                             const int64_t* __restrict__ tileoff, unsigned long long* __restrict__ dict,
                                                             long long* __restrict__ first_rows, int* __restrict__ null_id, int slots,
                                                             const ulonglong2* __restrict__ rank_rec) {
  // four slots per lane, their chains (first row → bitmap word, word prefix, tile offset → id → three scattered stores) side by
  // side: one slot per iteration left every wave with a single dependent chain of five round trips (2^24 keys: 1.0 ms for this pass)
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * kBlock * U;
  for (int64_t s0 = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; s0 < nslots; s0 += stride) {
    unsigned fr[U], id[U];
    unsigned long long key[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t s = s0 + (int64_t)u * kBlock;
      fr[u] = s < nslots ? tab_first[s] : kNoRow;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t s = s0 + (int64_t)u * kBlock;
      id[u] = 0; key[u] = 0;
      if (fr[u] != kNoRow) {
        if (rank_rec) {   // {bitmap word, ranks below the word} in ONE 16-byte gather instead of three gathers from three arrays
          const ulonglong2 rr = rank_rec[fr[u] >> 6];
          id[u] = (unsigned)rr.y + (unsigned)__popcll(rr.x & ((1ull << (fr[u] & 63)) - 1));
        } else {
          id[u] = rank_of_row(fr[u], firsts, wordprefix, tileoff);
        }
        if (dict) key[u] = tab_key[s];
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t s = s0 + (int64_t)u * kBlock;
      if (fr[u] == kNoRow) continue;
      tab_first[s] = id[u];
      const int in_part = (int)(s % (slots + 8));
      unsigned long long k = key[u];
      if (in_part == slots) k = kEmpty;
      if (in_part == slots + 1) { k = 0; *null_id = (int)id[u]; }   // GetDictArrayData: the null slot keeps the fresh buffer's zero
      if (dict) dict[id[u]] = k;
      if (first_rows) first_rows[id[u]] = (long long)fr[u];
    }
  }
}

// the null entry: its id (when no assign pass ran: from the null slot of table 0, where all null keys went) and its dictionary value —
// GetDictArrayData leaves the fresh buffer's zero there, whatever bits the first null row holds
__global__ void enc_null_entry_kernel(const unsigned* __restrict__ tab_first, int slots, int need_rank, const unsigned long long* __restrict__ firsts,
                                      const unsigned* __restrict__ wordprefix, const int64_t* __restrict__ tileoff, int* __restrict__ null_id,
                                      unsigned long long* __restrict__ dict) {
  if (need_rank) {
    const unsigned fr = tab_first[slots + 1];
    if (fr != kNoRow) *null_id = (int)rank_of_row(fr, firsts, wordprefix, tileoff);
  }
  const int id = *null_id;
  if (id >= 0 && dict) dict[id] = 0;
}

__global__ __launch_bounds__(kBlock) void rank_rec_kernel(const unsigned long long* __restrict__ firsts, const unsigned* __restrict__ wordprefix,
                                                           const int64_t* __restrict__ tileoff, int64_t nwords, ulonglong2* __restrict__ rec) {
  const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (w < nwords) rec[w] = make_ulonglong2(firsts[w], (unsigned long long)(tileoff[w >> kRankTileLog2] + wordprefix[w]));
}

// ids of the slots + the dictionary (and the first rows) in id order.  compact = false: every used slot writes its key to dict[id] —
// ids follow first rows, so these are scattered 8-byte stores, two per key, behind two scattered reads (2^24 keys: 0.87 ms).
// compact = true: the dictionary is the key column itself compacted by the first-occurrence bitmap (ah_filter.hip: tiles without a
// first occurrence are never fetched, the output leaves as whole lines), and the slots only look their ids up — or nothing at
// all when the caller wants no ids (unique).
int enc_emit(ah_ctx* c, bool compact, bool need_ids, const unsigned long long* tab_key, unsigned* tab_first, int64_t nslots, const unsigned long long* firsts,
             const unsigned* wordprefix, const int64_t* tileoff, const uint64_t* keys, int64_t n, uint64_t* out_dict, int64_t* out_first_rows, int* null_id,
             int slots) {
  const unsigned agrid = ah_stream_grid(c, ah_ceil_div(nslots, (int64_t)kBlock * 4));
  if (!compact) {
    enc_assign_kernel<<<agrid, kBlock, 0, c->stream>>>(tab_key, tab_first, nslots, firsts, wordprefix, tileoff, (unsigned long long*)out_dict,
                                                       (long long*)out_first_rows, null_id, slots, nullptr);
    AH_LAUNCH_CHECK(c);
    return AH_OK;
  }
  int rc = ah_compact_u64_by_bits(c, keys, (const uint8_t*)firsts, n, out_dict, out_first_rows);
  if (rc != AH_OK) return rc;
  if (need_ids) {
    // the key table is not read any more (the dictionary came from the column): its block holds the rank records
    const int64_t nwords = ah_ceil_div(n, 64);
    ulonglong2* rec = nullptr;
    if (nslots * 8 >= nwords * 16) {
      rec = (ulonglong2*)tab_key;
      rank_rec_kernel<<<(unsigned)ah_ceil_div(nwords, kBlock), kBlock, 0, c->stream>>>(firsts, wordprefix, tileoff, nwords, rec);
      AH_LAUNCH_CHECK(c);
    }
    enc_assign_kernel<<<agrid, kBlock, 0, c->stream>>>(tab_key, tab_first, nslots, firsts, wordprefix, tileoff, nullptr, nullptr, null_id, slots, rec);
    AH_LAUNCH_CHECK(c);
  }
  enc_null_entry_kernel<<<1, 1, 0, c->stream>>>(tab_first, slots, need_ids ? 0 : 1, firsts, wordprefix, tileoff, null_id, (unsigned long long*)out_dict);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

// ---- 5: slot numbers → ids, in partition order ---------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(kThreads) void enc_resolve_kernel(const unsigned short* __restrict__ rec_slot, const unsigned* __restrict__ tab_id,
                                                                const unsigned* __restrict__ binstart, int* __restrict__ rec_id, int split) {
  constexpr int kLS = S + 2, kStride = S + 8;
  __shared__ unsigned l_id[kLS];
  // `split` workgroups share a partition's records (each loads the partition's 32 KiB id table): with one workgroup per
  // partition 256 partitions put 16 waves on a CU and the pass waited on memory 90 % of its time (2.4 TB/s)
  const int t = threadIdx.x, part = blockIdx.x / split, piece = blockIdx.x % split;
  const int64_t p0 = binstart[part], p1 = binstart[part + 1];
  const int64_t per = (((p1 - p0 + split - 1) / split) + 63) & ~(int64_t)63;
  const int64_t r0 = p0 + piece * per, r1 = r0 + per < p1 ? r0 + per : p1;
  if (r0 >= r1) return;
  for (int j = t; j < kLS; j += kThreads) l_id[j] = tab_id[(int64_t)part * kStride + j];
  __syncthreads();
  // Eight records per lane: ONE 16-byte load of slot numbers, two 16-byte stores of ids (2-byte loads moved 128 bytes per wave and
  // instruction: 2.4–3.2 TB/s).  The piece's ragged ends — up to the first and from the last multiple of 8 — go one record per lane.
  auto one = [&](int64_t i) {
    const unsigned short sl = rec_slot[i];
    rec_id[i] = sl == kMaskedSlot ? 0 : (int)l_id[sl];
  };
  const int64_t a0 = (r0 + 7) & ~(int64_t)7, a1 = r1 & ~(int64_t)7;
  if (a0 >= a1) {
    for (int64_t i = r0 + t; i < r1; i += kThreads) one(i);
    return;
  }
  if (r0 + t < a0) one(r0 + t);
  if (a1 + t < r1) one(a1 + t);
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef int v4i __attribute__((ext_vector_type(4)));
  const int64_t nvec = (a1 - a0) >> 3;
  const v4u* __restrict__ src = reinterpret_cast<const v4u*>(rec_slot + a0);   // a0 is a multiple of 8 records = 16 bytes; the arrays are 256-byte aligned
  v4i* __restrict__ dst = reinterpret_cast<v4i*>(rec_id + a0);
  constexpr int U = 2;
  for (int64_t b = 0; b < nvec; b += (int64_t)kThreads * U) {
    v4u w[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = b + u * kThreads + t;
      w[u] = i < nvec ? __builtin_nontemporal_load(&src[i]) : (v4u){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = b + u * kThreads + t;
      if (i >= nvec) continue;
      int id[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const unsigned sl = (w[u][e >> 1] >> (16 * (e & 1))) & 0xffffu;
        id[e] = sl == kMaskedSlot ? 0 : (int)l_id[sl];
      }
      __builtin_nontemporal_store((v4i){id[0], id[1], id[2], id[3]}, &dst[2 * i]);
      __builtin_nontemporal_store((v4i){id[4], id[5], id[6], id[7]}, &dst[2 * i + 1]);
    }
  }
}

// ---- 6: ids back to row order, tile by tile -------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void enc_unpermute_kernel(const int* __restrict__ rec_id, const unsigned* __restrict__ prows,
                                                                  const unsigned* __restrict__ cnt_tm, const unsigned* __restrict__ toffs, int nb,
                                                                  int64_t ntiles, int64_t n, int32_t* __restrict__ out_ids) {
  __shared__ unsigned s_cnt[kMaxBins], s_start[kMaxBins], s_goff[kMaxBins], s_wsum[kThreads / 64];
  __shared__ int s_out[kGbTile];
  const int64_t tile = xcd_contiguous_tile(ntiles);
  if (tile < 0) return;
  unsigned excl = 0;
  s_cnt[threadIdx.x] = 0;
  if ((int)threadIdx.x < nb) {
    s_cnt[threadIdx.x] = cnt_tm[tile * nb + threadIdx.x];
    excl = toffs[tile * nb + threadIdx.x];
  }
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  if ((int)threadIdx.x < nb) s_goff[threadIdx.x] = excl - s_start[threadIdx.x];
  __syncthreads();
  const int64_t base = tile * kGbTile;
  const int tile_n = n - base >= kGbTile ? kGbTile : (int)(n - base);
  // staged position lp → its partition = the LAST b with s_start[b] ≤ lp (empty partitions share their start with the next one)
  int lo[kGbRows], hi[kGbRows];
#pragma unroll
  for (int k = 0; k < kGbRows; k++) { lo[k] = 0; hi[k] = nb - 1; }
#pragma unroll 1
  for (int step = 0; step < 10; step++) {   // 2^10 = kMaxBins
#pragma unroll
    for (int k = 0; k < kGbRows; k++) {
      const int mid = (lo[k] + hi[k] + 1) >> 1;
      const bool le = s_start[mid] <= (unsigned)(k * kThreads + threadIdx.x);
      lo[k] = le ? mid : lo[k];
      hi[k] = le ? hi[k] : mid - 1;
    }
  }
  int id[kGbRows];
  unsigned rw[kGbRows];
#pragma unroll
  for (int k = 0; k < kGbRows; k++) {
    const int lp = k * kThreads + threadIdx.x;
    id[k] = 0; rw[k] = 0;
    if (lp < tile_n) {
      const int64_t e = (int64_t)s_goff[lo[k]] + lp;
      id[k] = __builtin_nontemporal_load(&rec_id[e]);
      rw[k] = __builtin_nontemporal_load(&prows[e]);
    }
  }
#pragma unroll
  for (int k = 0; k < kGbRows; k++)
    if (k * kThreads + (int)threadIdx.x < tile_n) s_out[(int64_t)(rw[k] & kRowMask) - base] = id[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kGbRows; k++) {
    const int i = k * kThreads + threadIdx.x;
    if (i < tile_n) __builtin_nontemporal_store(s_out[i], &out_ids[base + i]);
  }
}

// The same for G CONSECUTIVE tiles per workgroup.  Inside a partition the runs of consecutive tiles lie one after the other, so the
// group's records of a partition are ONE run G times as long (256 partitions: 16 records = 64 bytes per tile, 256 bytes per group
// of four) — the per-tile kernel read 64-byte pieces at 2.3 TB/s.  The group's ids are staged in LDS (G · 16 KiB) and leave as whole
// lines.
template <int G>
__global__ __launch_bounds__(kThreads) void enc_unpermute_group_kernel(const int* __restrict__ rec_id, const unsigned* __restrict__ prows,
                                                                       const unsigned* __restrict__ cnt_tm, const unsigned* __restrict__ toffs, int nb,
                                                                       int64_t ntiles, int64_t n, int32_t* __restrict__ out_ids) {
  __shared__ unsigned s_cnt[kMaxBins], s_start[kMaxBins], s_goff[kMaxBins], s_wsum[kThreads / 64];
  __shared__ int s_out[G * kGbTile];
  const int64_t ngroups = (ntiles + G - 1) / G;
  const int64_t grp = xcd_contiguous_tile(ngroups);
  if (grp < 0) return;
  const int64_t tile0 = grp * G;
  unsigned excl = 0, cnt = 0;
  if ((int)threadIdx.x < nb) {
    excl = toffs[tile0 * nb + threadIdx.x];
#pragma unroll
    for (int g = 0; g < G; g++)
      if (tile0 + g < ntiles) cnt += cnt_tm[(tile0 + g) * nb + threadIdx.x];
  }
  s_cnt[threadIdx.x] = cnt;
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  if ((int)threadIdx.x < nb) s_goff[threadIdx.x] = excl - s_start[threadIdx.x];
  __syncthreads();
  const int64_t base = tile0 * kGbTile;
  const int grp_n = n - base >= (int64_t)G * kGbTile ? G * kGbTile : (int)(n - base);
#pragma unroll 1
  for (int c = 0; c < G; c++) {
    int lo[kGbRows], hi[kGbRows];
#pragma unroll
    for (int k = 0; k < kGbRows; k++) { lo[k] = 0; hi[k] = nb - 1; }
#pragma unroll 1
    for (int step = 0; step < 10; step++) {   // 2^10 = kMaxBins
#pragma unroll
      for (int k = 0; k < kGbRows; k++) {
        const int mid = (lo[k] + hi[k] + 1) >> 1;
        const bool le = s_start[mid] <= (unsigned)(c * kGbTile + k * kThreads + threadIdx.x);
        lo[k] = le ? mid : lo[k];
        hi[k] = le ? hi[k] : mid - 1;
      }
    }
    int id[kGbRows];
    unsigned rw[kGbRows];
#pragma unroll
    for (int k = 0; k < kGbRows; k++) {
      const int lp = c * kGbTile + k * kThreads + threadIdx.x;
      id[k] = 0; rw[k] = 0;
      if (lp < grp_n) {
        const int64_t e = (int64_t)s_goff[lo[k]] + lp;
        id[k] = __builtin_nontemporal_load(&rec_id[e]);
        rw[k] = __builtin_nontemporal_load(&prows[e]);
      }
    }
#pragma unroll
    for (int k = 0; k < kGbRows; k++)
      if (c * kGbTile + k * kThreads + (int)threadIdx.x < grp_n) s_out[(int64_t)(rw[k] & kRowMask) - base] = id[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < G * kGbRows; k++) {
    const int i = k * kThreads + threadIdx.x;
    if (i < grp_n) __builtin_nontemporal_store(s_out[i], &out_ids[base + i]);
  }
}

void launch_enc_unpermute(ah_ctx* c, const int* rec_id, const unsigned* prows, const unsigned* cnt_tm, const unsigned* toffs, int nb, int64_t ntiles,
                          int64_t n, int32_t* out_ids) {
  const int G = c->opt_encode_unperm_group;
  if (G > 1) {
    const int g = G >= 8 ? 8 : (G >= 4 ? 4 : 2);
    const int64_t ngroups = (ntiles + g - 1) / g;
    const unsigned grid = (unsigned)(((ngroups + 7) / 8) * 8);
    if (g == 8) enc_unpermute_group_kernel<8><<<grid, kThreads, 0, c->stream>>>(rec_id, prows, cnt_tm, toffs, nb, ntiles, n, out_ids);
    else if (g == 4) enc_unpermute_group_kernel<4><<<grid, kThreads, 0, c->stream>>>(rec_id, prows, cnt_tm, toffs, nb, ntiles, n, out_ids);
    else enc_unpermute_group_kernel<2><<<grid, kThreads, 0, c->stream>>>(rec_id, prows, cnt_tm, toffs, nb, ntiles, n, out_ids);
  } else {
    enc_unpermute_kernel<<<(unsigned)(((ntiles + 7) / 8) * 8), kThreads, 0, c->stream>>>(rec_id, prows, cnt_tm, toffs, nb, ntiles, n, out_ids);
  }
}


// ---- more than 1024 partitions: a second cut ---------------------------------------------------------------------------------------
// Beyond ≈ 4.5 M keys 1024 LDS tables are not enough, and one scatter into thousands of partitions would write runs of one or two
// rows.  So the cut is made twice, as in the MSD sort and the large group-by (ah_msd.h): level 1 = the pass above with 64
// partitions ("parents"); level 2 cuts every parent into 2^lb2 ≤ 128 partitions — its tiles are "virtual tiles" that never cross
// a parent boundary and all run on one XCD.  A level-2 record is {key, row | flag} plus its position in the virtual tile (2 bytes):
// the ids go home in two steps, virtual tile by virtual tile back into level-1 order, then tile by tile into row order.
__device__ __forceinline__ unsigned e2_digit(unsigned long long key, unsigned rw, int lp, unsigned mask) {
  return (rw & kKeyNull) ? 0u : ((unsigned)(gb_mix(key) >> (64 - lp)) & mask);   // level 1 took the top 6 of these lp bits (null keys: partition 0 of parent 0)
}

__global__ __launch_bounds__(kThreads) void e2_hist_kernel(const unsigned long long* __restrict__ pkeys, const unsigned* __restrict__ prows, int64_t n,
                                                            const unsigned* __restrict__ pstart, int nparents, int lp, unsigned mask, int nb,
                                                            unsigned* __restrict__ cnt, const TileRange* __restrict__ tile_table = nullptr) {
  __shared__ unsigned s_h[kThreads];
  __shared__ unsigned s_cnt[kThreads], s_start[kThreads], s_wsum[kThreads / 64];
  __shared__ int s_pick;
  // (tile_table: this launch's tiles precomputed — a histogram's workgroup lives ≈ 5 µs, ms_tile's loads, scan and barriers were 2 of them)
  const TileRange r = tile_table ? tile_table[blockIdx.x] : ms_tile(pstart, nparents, n, s_cnt, s_start, s_wsum, &s_pick);
  if (r.parent < 0) return;
  for (int b = threadIdx.x; b < nb; b += kThreads) s_h[b] = 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    const int64_t i = r.lo + u * kThreads + threadIdx.x;
    // (null keys all sit in parent 0: the other 63 parents' histograms read the keys alone, 8 instead of 12 bytes per record)
    if (i < r.hi) atomicAdd(&s_h[e2_digit(__builtin_nontemporal_load(&pkeys[i]), r.parent == 0 ? __builtin_nontemporal_load(&prows[i]) : 0u, lp, mask)], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kThreads) cnt[r.id * nb + b] = s_h[b];
}

// level-2 offsets for FEW digits (nb ≤ 128): ms_offs2_kernel (ah_msd.h) gives every digit one thread that walks all of the parent's
// virtual tiles — right for the sort's 2048 digits, but here 64 of 1024 threads would walk ≈ 260 tiles twice, one dependent load
// after the other (343 µs at 2^26 rows).  Here thread (slice s, digit d) takes a contiguous slice of the parent's tiles,
// S = 1024 / nb slices per digit; slice totals meet in LDS, one scan, then every thread walks its own slice once more.
__global__ __launch_bounds__(kThreads) void e2_offs2_kernel(const unsigned* __restrict__ cnt, const unsigned* __restrict__ pstart, int nparents, int nb,
                                                             unsigned* __restrict__ toffs, unsigned* __restrict__ bstart, int64_t n,
                                                             unsigned* __restrict__ largest, unsigned* __restrict__ done, unsigned long long* mb, unsigned long long seq) {
  // largest / done: two device words, zero on entry and left zero — the largest final partition's size goes to the host's mailbox
  // from the last workgroup to finish (the host looks at the partitions' balance while the scatter behind this kernel already runs)
  __shared__ unsigned s_cnt[kThreads], s_start[kThreads], s_wsum[kThreads / 64];
  __shared__ unsigned s_part[kThreads];   // [slice][digit] totals
  __shared__ unsigned s_largest;
  const int t = threadIdx.x, p = blockIdx.x;
  if (t == 0) s_largest = 0;
  unsigned tiles = 0;
  if (t < nparents) { const int q = ms_parent_of(t, nparents); tiles = (pstart[q + 1] - pstart[q] + kMsTile - 1) / kMsTile; }
  s_cnt[t] = tiles;
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nparents);
  const int j = (p & 7) * (nparents >> 3) + (p >> 3);   // this parent's place in the class-major tile numbering (ms_tile)
  const int64_t vt0 = s_start[j], nvt = s_cnt[j];
  __syncthreads();
  const int S = kThreads / nb, d = t % nb, sl = t / nb;
  const int64_t per = (nvt + S - 1) / S, a = vt0 + sl * per, b = a + per < vt0 + nvt ? a + per : vt0 + nvt;
  unsigned tot = 0;
  for (int64_t vt = a; vt < b; vt++) tot += cnt[vt * nb + d];
  s_part[sl * nb + d] = sl < S ? tot : 0u;
  __syncthreads();
  // digit totals → exclusive scan over the digits
  unsigned dtot = 0;
  if (t < nb) for (int q = 0; q < S; q++) dtot += s_part[q * nb + t];
  s_cnt[t] = t < nb ? dtot : 0u;
  if (t < nb && dtot) atomicMax(&s_largest, dtot);
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  if (t == 0) {
    atomicMax(largest, s_largest);
    __threadfence();
    if (atomicAdd(done, 1u) == gridDim.x - 1u) {
      __threadfence();
      const unsigned long long w = __hip_atomic_exchange(largest, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ah_mailbox_post(mb, seq, &w, 1);
    }
  }
  const unsigned base = pstart[p];
  if (t < nb) bstart[(int64_t)p * nb + t] = base + s_start[t];
  // this thread's running offset = parent start + smaller digits + this digit in earlier slices
  unsigned run = base + s_start[d];
  for (int q = 0; q < sl; q++) run += s_part[q * nb + d];
  if (sl < S)
    for (int64_t vt = a; vt < b; vt++) { toffs[vt * nb + d] = run; run += cnt[vt * nb + d]; }
  if (p == nparents - 1 && t == 0) bstart[(int64_t)nparents * nb] = (unsigned)n;
}

__global__ __launch_bounds__(kThreads) void e2_scatter_kernel(const unsigned long long* __restrict__ pkeys, const unsigned* __restrict__ prows, int64_t n,
                                                               const unsigned* __restrict__ pstart, int nparents, int lp, unsigned mask, int nb,
                                                               const unsigned* __restrict__ toffs, unsigned long long* __restrict__ out_keys,
                                                               unsigned* __restrict__ out_rows, unsigned short* __restrict__ out_j) {
  __shared__ unsigned s_cnt[kThreads], s_start[kThreads], s_goff[kThreads], s_wsum[kThreads / 64];
  __shared__ unsigned s_a[kThreads], s_b[kThreads];
  __shared__ unsigned long long s_stage[kMsTile];
  __shared__ uint16_t s_bin[kMsTile];
  __shared__ int s_pick;
  const TileRange r = ms_tile(pstart, nparents, n, s_a, s_b, s_wsum, &s_pick);
  if (r.parent < 0) return;
  const int t = threadIdx.x;
  s_cnt[t] = 0;
  unsigned long long k[kMsRows];
  unsigned rw[kMsRows], dg[kMsRows], rank[kMsRows];
  bool live[kMsRows];
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    const int64_t i = r.lo + u * kThreads + t;
    live[u] = i < r.hi;
    k[u] = live[u] ? __builtin_nontemporal_load(&pkeys[i]) : 0ull;
    rw[u] = live[u] ? __builtin_nontemporal_load(&prows[i]) : 0u;
  }
  unsigned goff_excl = 0;
  if (t < nb) goff_excl = toffs[r.id * nb + t];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    dg[u] = e2_digit(k[u], rw[u], lp, mask);
    rank[u] = live[u] ? atomicAdd(&s_cnt[dg[u]], 1u) : 0u;
  }
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nb);   // nb ≤ 128
  if (t < nb) s_goff[t] = goff_excl - s_start[t];
  const int tile_n = (int)(r.hi - r.lo);
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (live[u]) { const unsigned q = s_start[dg[u]] + rank[u]; s_stage[q] = k[u]; s_bin[q] = (uint16_t)dg[u]; }
  __syncthreads();
  int64_t dst[kMsRows];
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    const int q = u * kThreads + t;
    dst[u] = q < tile_n ? (int64_t)s_goff[s_bin[q]] + q : -1;
    if (dst[u] >= 0) out_keys[dst[u]] = s_stage[q];
  }
  __syncthreads();
  // second round: {row word, position in the virtual tile} as one 8-byte staged element
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (live[u]) s_stage[s_start[dg[u]] + rank[u]] = ((unsigned long long)(unsigned)(u * kThreads + t) << 32) | rw[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (dst[u] >= 0) {
      const unsigned long long x = s_stage[u * kThreads + t];
      out_rows[dst[u]] = (unsigned)x;
      out_j[dst[u]] = (unsigned short)(x >> 32);
    }
}

// ids of a virtual tile's records (scattered over the parent's partitions) → back at their level-1 positions
__global__ __launch_bounds__(kThreads) void e2_unpermute_kernel(const int* __restrict__ rec_id, const unsigned short* __restrict__ rec_j,
                                                                 const unsigned* __restrict__ cnt, const unsigned* __restrict__ toffs, int64_t n,
                                                                 const unsigned* __restrict__ pstart, int nparents, int nb, int* __restrict__ rec_id1) {
  __shared__ unsigned s_cnt[kThreads], s_start[kThreads], s_goff[kThreads], s_wsum[kThreads / 64];
  __shared__ unsigned s_a[kThreads], s_b[kThreads];
  __shared__ int s_out[kMsTile];
  __shared__ int s_pick;
  const TileRange r = ms_tile(pstart, nparents, n, s_a, s_b, s_wsum, &s_pick);
  if (r.parent < 0) return;
  const int t = threadIdx.x;
  unsigned excl = 0;
  s_cnt[t] = 0;
  if (t < nb) { s_cnt[t] = cnt[r.id * nb + t]; excl = toffs[r.id * nb + t]; }
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  if (t < nb) s_goff[t] = excl - s_start[t];
  __syncthreads();
  const int tile_n = (int)(r.hi - r.lo);
  int lo[kMsRows], hi[kMsRows];
#pragma unroll
  for (int k = 0; k < kMsRows; k++) { lo[k] = 0; hi[k] = nb - 1; }
#pragma unroll 1
  for (int step = 0; step < 8; step++) {   // 2^7 ≥ nb
#pragma unroll
    for (int k = 0; k < kMsRows; k++) {
      const int mid = (lo[k] + hi[k] + 1) >> 1;
      const bool le = s_start[mid] <= (unsigned)(k * kThreads + t);
      lo[k] = le ? mid : lo[k];
      hi[k] = le ? hi[k] : mid - 1;
    }
  }
  int id[kMsRows];
  unsigned short j[kMsRows];
#pragma unroll
  for (int k = 0; k < kMsRows; k++) {
    const int lp = k * kThreads + t;
    id[k] = 0; j[k] = 0;
    if (lp < tile_n) {
      const int64_t e = (int64_t)s_goff[lo[k]] + lp;
      id[k] = __builtin_nontemporal_load(&rec_id[e]);
      j[k] = __builtin_nontemporal_load(&rec_j[e]);
    }
  }
#pragma unroll
  for (int k = 0; k < kMsRows; k++)
    if (k * kThreads + t < tile_n) s_out[j[k]] = id[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kMsRows; k++) {
    const int i = k * kThreads + t;
    if (i < tile_n) __builtin_nontemporal_store(s_out[i], &rec_id1[r.lo + i]);
  }
}


// (option encode_unperm2_group: 4 = default, 0 = one virtual tile per workgroup.  Built last in round 3: 2^24 keys 2.64 → 2.60 ms)
// the level-2 un-permute over G consecutive virtual tiles of a parent per workgroup, as enc_unpermute_group_kernel does for level 1
// (353 → 205 µs there).  Inside a digit the runs of consecutive virtual tiles lie one after the other (e2_offs2_kernel walks the tiles
// in order), so the group reads ONE run per digit, G times as long (nb2 = 128 digits: 32 records per tile).  A record's tile inside
// the group follows from its offset in the group's run of its digit (cumulative counts per digit in LDS); its 2-byte position is
// relative to that tile.  Needs nparents ≤ 64 (the two-cut path always has 64).
template <int G>
__global__ __launch_bounds__(kThreads) void e2_unpermute_group_kernel(const int* __restrict__ rec_id, const unsigned short* __restrict__ rec_j,
                                                                       const unsigned* __restrict__ cnt, const unsigned* __restrict__ toffs, int64_t n,
                                                                       const unsigned* __restrict__ pstart, int nparents, int nb, int* __restrict__ rec_id1) {
  static_assert(G >= 2 && G <= 4, "cumulative counts of up to three earlier tiles");
  __shared__ unsigned s_tiles[64], s_tstart[64], s_gstart[64];
  __shared__ unsigned s_cnt[kThreads], s_start[kThreads], s_goff[kThreads], s_wsum[kThreads / 64];
  __shared__ unsigned s_cum[G - 1][128];
  __shared__ int s_out[G * kMsTile];
  __shared__ int s_pick;
  const int t = threadIdx.x;
  // tiles and groups per parent, in the class-major parent order the tile numbering of the count / offset tables uses (ms_tile)
  if (t < 64) {
    unsigned tiles = 0;
    if (t < nparents) { const int p = ms_parent_of(t, nparents); tiles = (pstart[p + 1] - pstart[p] + kMsTile - 1) / kMsTile; }
    const unsigned groups = (tiles + G - 1) / G;
    unsigned it = tiles, ig = groups;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned a = __shfl_up(it, o, 64), b = __shfl_up(ig, o, 64);
      if (t >= o) { it += a; ig += b; }
    }
    s_tiles[t] = tiles; s_tstart[t] = it - tiles; s_gstart[t] = ig - groups;
  }
  if (t == 0) s_pick = -1;
  __syncthreads();
  if (t < nparents) {
    const unsigned groups = (s_tiles[t] + G - 1) / G;
    if (groups && s_gstart[t] <= blockIdx.x && blockIdx.x < s_gstart[t] + groups) s_pick = t;
  }
  __syncthreads();
  const int j = s_pick;
  if (j < 0) return;
  const int p = ms_parent_of(j, nparents);
  const unsigned g = blockIdx.x - s_gstart[j];
  const int64_t vt0 = (int64_t)s_tstart[j] + (int64_t)g * G;
  const int gt = (int)(s_tiles[j] - g * G < (unsigned)G ? s_tiles[j] - g * G : (unsigned)G);   // tiles in this group
  const int64_t lo = (int64_t)pstart[p] + (int64_t)g * G * kMsTile;
  const int64_t pend = (int64_t)pstart[p + 1], hi = lo + (int64_t)gt * kMsTile < pend ? lo + (int64_t)gt * kMsTile : pend;
  // per digit: the group's record count, the cumulative counts after each of its tiles, where its run starts
  unsigned excl = 0, c = 0;
  if (t < nb) {
    excl = toffs[vt0 * nb + t];
#pragma unroll
    for (int q = 0; q < G; q++) {
      if (q < gt) c += cnt[(vt0 + q) * nb + t];
      if (q < G - 1) s_cum[q][t] = c;
    }
  }
  s_cnt[t] = t < nb ? c : 0u;
  __syncthreads();
  block_excl_scan(s_cnt, s_start, s_wsum, nb);
  if (t < nb) s_goff[t] = excl - s_start[t];
  __syncthreads();
  const int group_n = (int)(hi - lo);
#pragma unroll 1
  for (int ch = 0; ch < G; ch++) {
    int dlo[kMsRows], dhi[kMsRows];
#pragma unroll
    for (int k = 0; k < kMsRows; k++) { dlo[k] = 0; dhi[k] = nb - 1; }
#pragma unroll 1
    for (int step = 0; step < 8; step++) {   // 2^7 ≥ nb
#pragma unroll
      for (int k = 0; k < kMsRows; k++) {
        const int mid = (dlo[k] + dhi[k] + 1) >> 1;
        const bool le = s_start[mid] <= (unsigned)(ch * kMsTile + k * kThreads + t);
        dlo[k] = le ? mid : dlo[k];
        dhi[k] = le ? dhi[k] : mid - 1;
      }
    }
    int id[kMsRows];
    unsigned short jp[kMsRows];
#pragma unroll
    for (int k = 0; k < kMsRows; k++) {
      const int lp = ch * kMsTile + k * kThreads + t;
      id[k] = 0; jp[k] = 0;
      if (lp < group_n) {
        const int64_t e = (int64_t)s_goff[dlo[k]] + lp;
        id[k] = __builtin_nontemporal_load(&rec_id[e]);
        jp[k] = __builtin_nontemporal_load(&rec_j[e]);
      }
    }
#pragma unroll
    for (int k = 0; k < kMsRows; k++) {
      const int lp = ch * kMsTile + k * kThreads + t;
      if (lp < group_n) {
        const unsigned q = (unsigned)lp - s_start[dlo[k]];   // offset in the group's run of this digit
        int tile = 0;
#pragma unroll
        for (int w = 0; w < G - 1; w++) tile += q >= s_cum[w][dlo[k]] ? 1 : 0;
        s_out[tile * kMsTile + jp[k]] = id[k];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < G * kMsRows; k++) {
    const int i = k * kThreads + t;
    if (i < group_n) __builtin_nontemporal_store(s_out[i], &rec_id1[lo + i]);
  }
}

}  // namespace

// Called by encode_core (ah_hash.hip) for 8-byte keys.  lp = log2 of the number of partitions (8 … 10).  *used = 1: out_* hold the
// result; 0: not applicable or the attempt was void (a partition outgrew its table, or one partition holds several times its
// share of the rows) — nothing the caller owns was touched except out_ids / out_dict / out_first_rows, which it rewrites.
// ---- the first look: distinct keys among the first rows, exactly, in one launch ---------------------------------------------------
// What decides "global table or partitions, and how many" is the number of distinct keys among the first 2^16 rows (the urn model
// turns it into the column's expected cardinality, ah_hash.hip).  The global-table path gets that number as a by-product of its
// staged inserts — behind a table fill of tens of MiB, two insert launches and a polled read, ≈ 90 µs that a call which then goes to
// the partitions has spent for nothing (2^20 keys in 2^26 rows: 1.198 ms with the look, 1.105 with the partition count forced,
// profiles/r05_bench_encode_part.json).  This kernel counts the same thing without a table in HBM: kLookWgs workgroups each read all
// `rows` keys (512 KiB out of L2), keep the 1 / kLookWgs of them whose hash names the workgroup, and count those exactly in an LDS set;
// the last workgroup to finish posts the total to the host's mailbox.  The count must be exact: at 2^24 keys the 2^16 rows hold
// ≈ 128 repeats, and a linear-counting bitmap that fits LDS is off by ± 43.
namespace {
constexpr int kLookWgs = 64, kLookSlots = 4096;   // ≤ 2^16 rows: 1024 ± 32 keys per workgroup; keys that find no room count as new ("too many to tell")
__global__ __launch_bounds__(kThreads) void enc_look_kernel(const unsigned long long* __restrict__ keys, const uint8_t* __restrict__ valid, int64_t off, int64_t rows,
                                                            unsigned long long* __restrict__ acc, unsigned* __restrict__ done, unsigned long long* mb,
                                                            unsigned long long seq) {
  __shared__ unsigned long long l_key[kLookSlots];
  __shared__ unsigned s_new, s_last, s_ones;
  const int t = threadIdx.x;
  for (int j = t; j < kLookSlots; j += kThreads) l_key[j] = kEmpty;
  if (t == 0) { s_new = 0; s_ones = 0; }
  __syncthreads();
  unsigned mine = 0;
  // Only 1 / 64 of the rows are this workgroup's: a lane that probed as soon as it met one held its wave for a whole probe sequence
  // — in two of three steps SOME lane has one — and the pass was forty serial LDS round trips per wave (32 µs).  The wave's rows of
  // interest are queued (ballot + prefix count, LDS) and probed together once the queue holds a wave's worth.
  __shared__ unsigned long long s_queue[kThreads / 64][128];
  const int lane = t & 63, wave = t >> 6;
  unsigned long long* q = s_queue[wave];
  int queued = 0;   // (wave-uniform)
  auto insert = [&](unsigned long long key) {
    if (key == kEmpty) { if (atomicExch(&s_ones, 1u) == 0u) mine++; return; }   // the all-ones key is the set's empty marker: counted on the side
    const unsigned m = ((unsigned)key ^ ((unsigned)(key >> 32) * 0x9E3779B1u)) * 0x85EBCA6Bu;
    unsigned j = (m >> 10) & (unsigned)(kLookSlots - 1);
    bool placed = false;
    for (int probes = 0; probes < kLookSlots && !placed; probes++) {
      unsigned long long cur = l_key[j];
      if (cur == kEmpty) cur = atomicCAS(&l_key[j], kEmpty, key);
      if (cur == kEmpty) { mine++; placed = true; }
      else if (cur == key) placed = true;
      j = (j + 1) & (unsigned)(kLookSlots - 1);
    }
    if (!placed) mine++;
  };
  auto drain = [&](int from) {   // the queue's entries from `from` on, one per lane
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (from + lane < queued) insert(q[from + lane]);
    __builtin_amdgcn_wave_barrier();
  };
  constexpr int U = 16;   // loads in flight per lane: the pass is four round trips to L2 long
  for (int64_t b = 0; b < rows; b += (int64_t)kThreads * U) {
    unsigned long long k[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const int64_t i = b + u * kThreads + t; k[u] = i < rows ? keys[i] : 0ull; }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = b + u * kThreads + t;
      // a 32-bit hash (two quarter-rate multiplies): every workgroup hashes ALL the rows
      const unsigned m = ((unsigned)k[u] ^ ((unsigned)(k[u] >> 32) * 0x9E3779B1u)) * 0x85EBCA6Bu;
      const bool in = i < rows && ah_bit(valid, off + i) && (m >> 26) == blockIdx.x;   // kLookWgs = 64: the top six bits
      const unsigned long long mask = __ballot(in);
      if (in) q[queued + __popcll(mask & ((1ull << lane) - 1ull))] = k[u];
      queued += __popcll(mask);
      if (queued >= 64) {   // (uniform) at most 127 queued: the first 64 go now
        drain(0);
        const unsigned long long rest = lane + 64 < queued ? q[lane + 64] : 0ull;
        __builtin_amdgcn_wave_barrier();
        if (lane + 64 < queued) q[lane] = rest;
        queued -= 64;
      }
    }
  }
  drain(0);
  if (mine) atomicAdd(&s_new, mine);
  __syncthreads();
  if (t == 0) {
    atomicAdd(acc, (unsigned long long)s_new);
    __threadfence();
    s_last = atomicAdd(done, 1u) == gridDim.x - 1u ? 1u : 0u;
    if (s_last) {
      __threadfence();
      const unsigned long long w = __hip_atomic_exchange(acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // both words are left zero for the next call
      __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ah_mailbox_post(mb, seq, &w, 1);
    }
  }
}

}  // namespace
// distinct valid keys among the first `rows` (≤ 2^16) rows → *distinct.  One launch and a polled wait.
int ah_encode_first_look(ah_ctx* c, const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t rows, uint64_t* distinct) {
  unsigned long long* acc = (unsigned long long*)&c->dscalars[33];   // [33] the count, [34] the workgroups that have finished: zero on entry, zeroed again by the last workgroup
  unsigned long long *mb, seq, w;
  int rc = ah_mailbox_begin(c, &mb, &seq);
  if (rc != AH_OK) return rc;
  enc_look_kernel<<<kLookWgs, kThreads, 0, c->stream>>>((const unsigned long long*)keys, valid, off, rows, acc, (unsigned*)&c->dscalars[34], mb, seq);
  AH_LAUNCH_CHECK(c);
  if ((rc = ah_mailbox_wait(c, seq, 1, &w)) != AH_OK) return rc;
  *distinct = (uint64_t)w;
  return AH_OK;
}

int ah_encode_partitioned_try(ah_ctx* c, const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls, int lp, int slots,
                              int32_t* out_ids, uint64_t* out_dict, int64_t* out_first_rows, int64_t* out_ndict, int32_t* out_null_id, int* used) {
  *used = 0;
  if (n < 1 || n >= kMaxRows || lp < 3 || lp > 10 || (slots != kESlots && slots != kESlots2)) return AH_OK;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const int P = 1 << lp;
  const int64_t ntiles = ah_ceil_div(n, kGbTile), ngrp = ah_ceil_div(ntiles, kGroupTiles);
  const int64_t nslots = (int64_t)P * (slots + 8);
  const int64_t nwords = ah_ceil_div(n, 64), nrt = rank_tiles(nwords);
  const size_t table = (size_t)P * (size_t)ntiles * 4;
  const size_t need = pad(table) * 2 + pad((size_t)ngrp * P * 4) + pad((size_t)(P + 1) * 4) + pad((size_t)n * 8) + pad((size_t)n * 4) + pad((size_t)n * 2) +
                      pad((size_t)nslots * 8) + pad((size_t)nslots * 4) + pad((size_t)nwords * 8) + pad((size_t)nwords * 4) + pad((size_t)nrt * 4) + pad((size_t)nrt * 8) + pad((size_t)nwords * 64);
  uint8_t* base;
  int rc = ah_temp_reserve(c, need, (void**)&base);
  if (rc != AH_OK) return rc;
  size_t used_b = 0;
  auto take = [&](size_t b) { uint8_t* q = base + used_b; used_b += pad(b); return q; };
  unsigned* cnt_tm = (unsigned*)take(table);
  unsigned* toffs = (unsigned*)take(table);
  unsigned* gsum = (unsigned*)take((size_t)ngrp * P * 4);
  unsigned* binstart = (unsigned*)take((size_t)(P + 1) * 4);
  unsigned long long* pkeys = (unsigned long long*)take((size_t)n * 8);
  unsigned* prows = (unsigned*)take((size_t)n * 4);
  unsigned short* rec_slot = (unsigned short*)take((size_t)n * 2);
  unsigned long long* tab_key = (unsigned long long*)take((size_t)nslots * 8);
  unsigned* tab_first = (unsigned*)take((size_t)nslots * 4);
  unsigned long long* firsts = (unsigned long long*)take((size_t)nwords * 8);
  // (from 1024 partitions on — ≈ 4·10^6 keys: below, the map's fill and packing cost what the atomics cost)
  uint8_t* fbytes = (c->opt_encode_byte_map == 2 || (c->opt_encode_byte_map == 1 && lp >= 10)) ? (uint8_t*)take((size_t)nwords * 64) : nullptr;
  unsigned* wordprefix = (unsigned*)take((size_t)nwords * 4);
  int* tilecnt = (int*)take((size_t)nrt * 4);
  int64_t* tileoff = (int64_t*)take((size_t)nrt * 8);
  int* rec_id = (int*)pkeys;   // the keys are not needed once every record knows its slot
  unsigned* overflow = (unsigned*)&c->dscalars[30];
  unsigned long long* total = (unsigned long long*)&c->dscalars[31];
  int* null_id = (int*)&c->dscalars[32];
  {
    GbFill f;   // one launch where three memsets were (≈ 4 µs of gap each)
    f.njobs = 2;
    f.p[0] = (uint4*)&c->dscalars[30]; f.n16[0] = 1; f.v[0] = 0u;                 // [30] overflow, [31] total
    f.ones = (unsigned long long*)null_id;                                         // [32] null id: none
    if (fbytes) { f.p[1] = (uint4*)fbytes; f.n16[1] = (size_t)nwords * 4; }   // the byte map (the bitmap is then written whole)
    else { f.p[1] = (uint4*)firsts; f.n16[1] = pad((size_t)nwords * 8) / 16; }
    f.v[1] = 0u;
    gb_fill_kernel<<<(unsigned)(c->num_cu * 2), 256, 0, c->stream>>>(f);
    AH_LAUNCH_CHECK(c);
  }
  const unsigned long long* k64 = (const unsigned long long*)keys;
  // ---- 1, 2: cut
  const unsigned tgrid = (unsigned)(((ntiles + 7) / 8) * 8);
  gb_hist_kernel<<<tgrid, kGbHistThreads, 0, c->stream>>>(k64, valid, off, n, lp, P, ntiles, cnt_tm);
  AH_LAUNCH_CHECK(c);
  colsum_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt_tm, P, ntiles, gsum);
  AH_LAUNCH_CHECK(c);
  // one workgroup per partition is only as fast as the largest partition: the sizes are looked at before the tables are built — through
  // the mailbox, posted by the prefix kernel itself and read while the offsets and the scatter are already running (a copy of
  // binstart + a stream synchronisation left the device idle for ≈ 15 µs on every call; a lopsided column now costs a scatter
  // that nobody reads, and such columns rarely get here: the first look sends them to the global table)
  unsigned long long *mb, seq, largest = 0;
  if ((rc = ah_mailbox_begin(c, &mb, &seq)) != AH_OK) return rc;
  bin_prefix_kernel<<<1, kMaxBins, 0, c->stream>>>(gsum, P, ngrp, n, binstart, mb, seq);
  AH_LAUNCH_CHECK(c);
  tile_offs_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt_tm, gsum, P, ntiles, toffs);
  AH_LAUNCH_CHECK(c);
  gb_scatter_kernel<false><<<tgrid, kThreads, 0, c->stream>>>(k64, valid, off, nullptr, nullptr, 0, n, lp, P, ntiles, toffs, pkeys, nullptr, prows, nullptr);
  AH_LAUNCH_CHECK(c);
  if ((rc = ah_mailbox_wait(c, seq, 1, &largest)) != AH_OK) return rc;
  if ((int64_t)largest * P > 3 * n && largest > (1u << 16)) return AH_OK;
  // ---- 3: tables
  // (measured at 2^26 rows: from 2^22 keys on — 1024 partitions — the compaction wins; below, the slots' few scattered stores are cheaper than its pass)
  const bool compact = c->opt_encode_dict_compact >= 2 || (c->opt_encode_dict_compact == 1 && lp >= 10);
  unsigned long long* tab_key_out = compact ? nullptr : tab_key;
  if (slots == kESlots2) { if (c->opt_encode_table_batch) enc_table_kernel<kESlots2, true><<<(unsigned)P, kThreads, 0, c->stream>>>(pkeys, prows, binstart, encode_nulls, tab_key_out, tab_first, out_ids ? rec_slot : nullptr, firsts, overflow, fbytes); else enc_table_kernel<kESlots2, false><<<(unsigned)P, kThreads, 0, c->stream>>>(pkeys, prows, binstart, encode_nulls, tab_key_out, tab_first, out_ids ? rec_slot : nullptr, firsts, overflow, fbytes); }
  else { if (c->opt_encode_table_batch) enc_table_kernel<kESlots, true><<<(unsigned)P, kThreads, 0, c->stream>>>(pkeys, prows, binstart, encode_nulls, tab_key_out, tab_first, out_ids ? rec_slot : nullptr, firsts, overflow, fbytes); else enc_table_kernel<kESlots, false><<<(unsigned)P, kThreads, 0, c->stream>>>(pkeys, prows, binstart, encode_nulls, tab_key_out, tab_first, out_ids ? rec_slot : nullptr, firsts, overflow, fbytes); }
  AH_LAUNCH_CHECK(c);
  // ---- 4: rank
  if (fbytes) {
    enc_bytes_to_bits_kernel<<<(unsigned)ah_ceil_div(nwords, kBlock), kBlock, 0, c->stream>>>(fbytes, nwords, firsts);
    AH_LAUNCH_CHECK(c);
  }
  word_prefix_kernel<<<(unsigned)ah_ceil_div(nwords, kBlock), kBlock, 0, c->stream>>>(firsts, nwords, wordprefix, tilecnt);
  AH_LAUNCH_CHECK(c);
  scan_kernel<<<1, 1024, 0, c->stream>>>(tilecnt, nrt, tileoff, total);
  AH_LAUNCH_CHECK(c);
  if ((rc = enc_emit(c, compact, out_ids != nullptr, tab_key, tab_first, nslots, firsts, wordprefix, tileoff, (const uint64_t*)k64, n, (uint64_t*)out_dict,
                     (int64_t*)out_first_rows, null_id, slots)) != AH_OK) return rc;
  if (out_ids) {
    // ---- 5, 6: ids per record, then per row
    const int rsplit = P >= c->opt_encode_resolve_wgs ? 1 : (int)(c->opt_encode_resolve_wgs / P);   // ≥ 1024 workgroups: one per partition leaves a CU 16 waves
    if (slots == kESlots2) enc_resolve_kernel<kESlots2><<<(unsigned)(P * rsplit), kThreads, 0, c->stream>>>(rec_slot, tab_first, binstart, rec_id, rsplit);
    else enc_resolve_kernel<kESlots><<<(unsigned)(P * rsplit), kThreads, 0, c->stream>>>(rec_slot, tab_first, binstart, rec_id, rsplit);
    AH_LAUNCH_CHECK(c);
    launch_enc_unpermute(c, rec_id, prows, cnt_tm, toffs, P, ntiles, n, out_ids);
    AH_LAUNCH_CHECK(c);
  }
  if ((rc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[30], 3, (unsigned long long*)&c->pinned[8])) != AH_OK) return rc;   // overflow, total, null id
  if (*(volatile unsigned*)&c->pinned[8]) return AH_OK;   // a partition held more keys than its table admits: the global-table path redoes the call
  if (out_ndict) *out_ndict = (int64_t) * (volatile uint64_t*)&c->pinned[9];
  if (out_null_id) *out_null_id = *(volatile int32_t*)&c->pinned[10];
  *used = 1;
  return AH_OK;
}


// Two cuts: 64 parents × 2^(lp − 6) partitions (lp = 11 … 13), for ≈ 4.5 … 36 M expected keys.  Same contract as above.
int ah_encode_partitioned2_try(ah_ctx* c, const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls, int lp, int slots,
                               int32_t* out_ids, uint64_t* out_dict, int64_t* out_first_rows, int64_t* out_ndict, int32_t* out_null_id, int* used) {
  *used = 0;
  if (n < ((int64_t)1 << 20) || n >= kMaxRows || lp < 7 || lp > 13 || (slots != kESlots && slots != kESlots2)) return AH_OK;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  constexpr int lb1 = 6, nb1 = 1 << lb1;
  const int lb2 = lp - lb1, nb2 = 1 << lb2;
  const int64_t P = (int64_t)1 << lp;
  const int64_t ntiles = ah_ceil_div(n, kGbTile), ngrp = ah_ceil_div(ntiles, kGroupTiles), nvt = ((ntiles + nb1 + 7) / 8) * 8;
  const int64_t nslots = P * (slots + 8);
  const int64_t nwords = ah_ceil_div(n, 64), nrt = rank_tiles(nwords);
  const size_t need = pad((size_t)ntiles * nb1 * 4) * 2 + pad((size_t)ngrp * nb1 * 4) + pad((size_t)(nb1 + 1) * 4) + pad((size_t)nvt * nb2 * 4) * 2 + pad(((size_t)P + 1) * 4) +
                      pad((size_t)n * 8) * 2 + pad((size_t)n * 4) * 2 + pad((size_t)n * 2) * 2 + pad((size_t)nslots * 8) + pad((size_t)nslots * 4) +
                      pad((size_t)nwords * 8) + pad((size_t)nwords * 4) + pad((size_t)nrt * 4) + pad((size_t)nrt * 8) + pad((size_t)nwords * 64) + pad((size_t)nvt * sizeof(TileRange));
  uint8_t* base;
  int rc = ah_temp_reserve(c, need, (void**)&base);
  if (rc != AH_OK) return rc;
  size_t used_b = 0;
  auto take = [&](size_t b) { uint8_t* q = base + used_b; used_b += pad(b); return q; };
  unsigned* cnt1 = (unsigned*)take((size_t)ntiles * nb1 * 4);
  unsigned* toffs1 = (unsigned*)take((size_t)ntiles * nb1 * 4);
  unsigned* gsum = (unsigned*)take((size_t)ngrp * nb1 * 4);
  unsigned* pstart = (unsigned*)take((size_t)(nb1 + 1) * 4);
  unsigned* cnt2 = (unsigned*)take((size_t)nvt * nb2 * 4);
  unsigned* toffs2 = (unsigned*)take((size_t)nvt * nb2 * 4);
  unsigned* bstart = (unsigned*)take(((size_t)P + 1) * 4);
  unsigned long long* pkeys1 = (unsigned long long*)take((size_t)n * 8);
  unsigned long long* pkeys2 = (unsigned long long*)take((size_t)n * 8);
  unsigned* prows1 = (unsigned*)take((size_t)n * 4);
  unsigned* prows2 = (unsigned*)take((size_t)n * 4);
  unsigned short* pj2 = (unsigned short*)take((size_t)n * 2);
  unsigned short* rec_slot = (unsigned short*)take((size_t)n * 2);
  unsigned long long* tab_key = (unsigned long long*)take((size_t)nslots * 8);
  unsigned* tab_first = (unsigned*)take((size_t)nslots * 4);
  unsigned long long* firsts = (unsigned long long*)take((size_t)nwords * 8);
  uint8_t* fbytes = c->opt_encode_byte_map ? (uint8_t*)take((size_t)nwords * 64) : nullptr;
  unsigned* wordprefix = (unsigned*)take((size_t)nwords * 4);
  int* tilecnt = (int*)take((size_t)nrt * 4);
  int64_t* tileoff = (int64_t*)take((size_t)nrt * 8);
  int* rec_id2 = (int*)pkeys2;   // level-2 keys are dead once every record knows its slot
  int* rec_id1 = (int*)pkeys1;   // level-1 keys are dead after the level-2 scatter
  unsigned* overflow = (unsigned*)&c->dscalars[30];
  unsigned long long* total = (unsigned long long*)&c->dscalars[31];
  int* null_id = (int*)&c->dscalars[32];
  {
    GbFill f;   // one launch where three memsets were (≈ 4 µs of gap each)
    f.njobs = 2;
    f.p[0] = (uint4*)&c->dscalars[30]; f.n16[0] = 1; f.v[0] = 0u;                 // [30] overflow, [31] total
    f.ones = (unsigned long long*)null_id;                                         // [32] null id: none
    if (fbytes) { f.p[1] = (uint4*)fbytes; f.n16[1] = (size_t)nwords * 4; }   // the byte map (the bitmap is then written whole)
    else { f.p[1] = (uint4*)firsts; f.n16[1] = pad((size_t)nwords * 8) / 16; }
    f.v[1] = 0u;
    gb_fill_kernel<<<(unsigned)(c->num_cu * 2), 256, 0, c->stream>>>(f);
    AH_LAUNCH_CHECK(c);
  }
  const unsigned long long* k64 = (const unsigned long long*)keys;
  // ---- level 1: 64 parents by the top 6 bits of the key hash
  const unsigned tgrid = (unsigned)(((ntiles + 7) / 8) * 8);
  gb_hist_kernel<<<tgrid, kGbHistThreads, 0, c->stream>>>(k64, valid, off, n, lb1, nb1, ntiles, cnt1);
  AH_LAUNCH_CHECK(c);
  colsum_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt1, nb1, ntiles, gsum);
  AH_LAUNCH_CHECK(c);
  bin_prefix_kernel<<<1, kMaxBins, 0, c->stream>>>(gsum, nb1, ngrp, n, pstart);
  AH_LAUNCH_CHECK(c);
  tile_offs_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt1, gsum, nb1, ntiles, toffs1);
  AH_LAUNCH_CHECK(c);
  gb_scatter_kernel<false><<<tgrid, kThreads, 0, c->stream>>>(k64, valid, off, nullptr, nullptr, 0, n, lb1, nb1, ntiles, toffs1, pkeys1, nullptr, prows1, nullptr);
  AH_LAUNCH_CHECK(c);
  // ---- level 2: every parent into 2^lb2 partitions
  TileRange* tile_table = (TileRange*)take((size_t)nvt * sizeof(TileRange));
  ms_tile_table_kernel<<<(unsigned)ah_ceil_div(nvt, kThreads), kThreads, 0, c->stream>>>(pstart, nullptr, nb1, (unsigned)nvt, tile_table);
  AH_LAUNCH_CHECK(c);
  e2_hist_kernel<<<(unsigned)nvt, kThreads, 0, c->stream>>>(pkeys1, prows1, n, pstart, nb1, lp, (unsigned)(nb2 - 1), nb2, cnt2, tile_table);
  AH_LAUNCH_CHECK(c);
  // (the partitions' balance: posted by the offsets kernel, read while the scatter runs — see ah_encode_partitioned_try)
  unsigned long long *mb, seq, largest = 0;
  if ((rc = ah_mailbox_begin(c, &mb, &seq)) != AH_OK) return rc;
  e2_offs2_kernel<<<(unsigned)nb1, kThreads, 0, c->stream>>>(cnt2, pstart, nb1, nb2, toffs2, bstart, n, (unsigned*)&c->dscalars[35], (unsigned*)&c->dscalars[36], mb, seq);
  AH_LAUNCH_CHECK(c);
  e2_scatter_kernel<<<(unsigned)nvt, kThreads, 0, c->stream>>>(pkeys1, prows1, n, pstart, nb1, lp, (unsigned)(nb2 - 1), nb2, toffs2, pkeys2, prows2, pj2);
  AH_LAUNCH_CHECK(c);
  if ((rc = ah_mailbox_wait(c, seq, 1, &largest)) != AH_OK) return rc;
  if ((int64_t)largest * P > 3 * n && largest > (1u << 16)) return AH_OK;   // a key that owns a large share of the rows: the other path
  // ---- tables, ranks, ids: as in the one-level path, one workgroup per final partition
  const bool compact = c->opt_encode_dict_compact >= 1;
  unsigned long long* tab_key_out = compact ? nullptr : tab_key;
  if (slots == kESlots2) enc_table_kernel<kESlots2, false><<<(unsigned)P, kThreads, 0, c->stream>>>(pkeys2, prows2, bstart, encode_nulls, tab_key_out, tab_first, out_ids ? rec_slot : nullptr, firsts, overflow, fbytes);   // small partitions: the batched probe's registers cost more than its waits
  else enc_table_kernel<kESlots, false><<<(unsigned)P, kThreads, 0, c->stream>>>(pkeys2, prows2, bstart, encode_nulls, tab_key_out, tab_first, out_ids ? rec_slot : nullptr, firsts, overflow, fbytes);   // small partitions: the batched probe's registers cost more than its waits
  AH_LAUNCH_CHECK(c);
  if (fbytes) {
    enc_bytes_to_bits_kernel<<<(unsigned)ah_ceil_div(nwords, kBlock), kBlock, 0, c->stream>>>(fbytes, nwords, firsts);
    AH_LAUNCH_CHECK(c);
  }
  word_prefix_kernel<<<(unsigned)ah_ceil_div(nwords, kBlock), kBlock, 0, c->stream>>>(firsts, nwords, wordprefix, tilecnt);
  AH_LAUNCH_CHECK(c);
  scan_kernel<<<1, 1024, 0, c->stream>>>(tilecnt, nrt, tileoff, total);
  AH_LAUNCH_CHECK(c);
  if ((rc = enc_emit(c, compact, out_ids != nullptr, tab_key, tab_first, nslots, firsts, wordprefix, tileoff, (const uint64_t*)k64, n, (uint64_t*)out_dict,
                     (int64_t*)out_first_rows, null_id, slots)) != AH_OK) return rc;
  if (out_ids) {
    const int rsplit = P >= c->opt_encode_resolve_wgs ? 1 : (int)(c->opt_encode_resolve_wgs / P);   // ≥ 1024 workgroups: one per partition leaves a CU 16 waves
    if (slots == kESlots2) enc_resolve_kernel<kESlots2><<<(unsigned)(P * rsplit), kThreads, 0, c->stream>>>(rec_slot, tab_first, bstart, rec_id2, rsplit);
    else enc_resolve_kernel<kESlots><<<(unsigned)(P * rsplit), kThreads, 0, c->stream>>>(rec_slot, tab_first, bstart, rec_id2, rsplit);
    AH_LAUNCH_CHECK(c);
    if (c->opt_encode_unperm2_group >= 2 && nb1 <= 64 && nb2 <= 128)
      e2_unpermute_group_kernel<4><<<(unsigned)(ah_ceil_div(ntiles + nb1, 4) + nb1), kThreads, 0, c->stream>>>(rec_id2, pj2, cnt2, toffs2, n, pstart, nb1, nb2, rec_id1);
    else
      e2_unpermute_kernel<<<(unsigned)nvt, kThreads, 0, c->stream>>>(rec_id2, pj2, cnt2, toffs2, n, pstart, nb1, nb2, rec_id1);
    AH_LAUNCH_CHECK(c);
    launch_enc_unpermute(c, rec_id1, prows1, cnt1, toffs1, nb1, ntiles, n, out_ids);
    AH_LAUNCH_CHECK(c);
  }
  if ((rc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[30], 3, (unsigned long long*)&c->pinned[8])) != AH_OK) return rc;   // overflow, total, null id
  if (*(volatile unsigned*)&c->pinned[8]) return AH_OK;
  if (out_ndict) *out_ndict = (int64_t) * (volatile uint64_t*)&c->pinned[9];
  if (out_null_id) *out_null_id = *(volatile int32_t*)&c->pinned[10];
  *used = 1;
  return AH_OK;
}
