// ah_setlookup.hip — is_in: membership of every row in a value set (row §8(f)-2, reuses the
// device hash-table idea of ah_hash.hip).
//
// Replaces SetLookupState.Init + isInKernelExec (arrow/compute/internal/kernels/
// scalar_set_lookup.go:192-244, 374-413) behind compute's "is_in" (compute/scalar_set_lookup.go:
// 175-200).  Keys are the RAW BITS of the fixed-width value (uint8/16/32/64 by byte width, :106-133 —
// so +0.0 ≠ −0.0 and NaNs match by payload).  Per row:
//   valid: found → (true, valid) · else Inconclusive ∧ set-has-null → (false, NULL) · else (false, valid)
//   null:  Match ∧ set-has-null → (true, valid) · Skip ∨ (Match ∧ ¬set-has-null) → (false, valid) · else (false, NULL)
//   set-has-null is forced false under Skip (NullIndex stays −1, :239-242).
//
// The reference probes a memo table row by row on one core.  Here the set becomes
//   1- and 2-byte keys: a direct-address BITMAP (≤ 8 KiB) copied into LDS by every workgroup;
//   4- and 8-byte keys: an open-addressing table of the keys themselves (load ≤ ¼, linear probing,
//     a 2-multiply hash — membership needs no particular one), in LDS when it fits 32 KiB, else in HBM;
// and the column streams through once: 64 rows per wave step, results are two wave ballots stored
// as whole 64-bit words (read-modify-write only where the output range is not word-aligned).
// Algorithmic bytes: w + 2/8 per row (+ 1/8 with an input validity bitmap).
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kLdsSlots = 4096;  // 32 KiB of 8-byte slots: sets of up to 1024 keys probe in LDS
constexpr unsigned long long kEmpty = ~0ull;
constexpr int kChunks = 8;  // 64-row chunks per wave step

enum { kFlagSetHasNull = 1u, kFlagAllOnesKey = 2u };

// slot index: membership only needs SOME well-mixed hash; two 32-bit multiplies are ≈ 3× cheaper
// on CDNA4 than the reference's 64-bit hashInt (bswap64(PRIME·v): three v_mul_lo + one v_mul_hi)
__device__ __forceinline__ unsigned slot_of(unsigned long long v, unsigned mask, int shift) {
  const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  const unsigned h = (lo ^ (hi * 0x9E3779B1u)) * 0x85EBCA6Bu;
  return (h >> shift) & mask;  // the high bits of a multiplicative hash are the mixed ones
}

template <int W>
__device__ __forceinline__ unsigned long long load_key(const void* p, int64_t i) {
  if (W == 1) return ((const uint8_t*)p)[i];
  if (W == 2) return ((const uint16_t*)p)[i];
  if (W == 4) return ((const uint32_t*)p)[i];
  return ((const unsigned long long*)p)[i];
}

// ---- build ---------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(kBlock) void build_bitmap_kernel(const void* __restrict__ set_values, const uint8_t* __restrict__ set_valid,
                                                               int64_t set_off, int64_t set_n, unsigned* __restrict__ bits,
                                                               unsigned* __restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < set_n; i += stride) {
    if (!ah_bit(set_valid, set_off + i)) { atomicOr(flags, kFlagSetHasNull); continue; }
    const unsigned k = (unsigned)load_key<W>(set_values, i);
    atomicOr(&bits[k >> 5], 1u << (k & 31));
  }
}

template <int W>
__global__ __launch_bounds__(kBlock) void build_hash_kernel(const void* __restrict__ set_values, const uint8_t* __restrict__ set_valid,
                                                             int64_t set_off, int64_t set_n, unsigned long long* __restrict__ table,
                                                             unsigned mask, int shift, unsigned* __restrict__ flags) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < set_n; i += stride) {
    if (!ah_bit(set_valid, set_off + i)) { atomicOr(flags, kFlagSetHasNull); continue; }
    const unsigned long long k = load_key<W>(set_values, i);
    if (k == kEmpty) { atomicOr(flags, kFlagAllOnesKey); continue; }  // the empty marker's own value gets a flag
    unsigned idx = slot_of(k, mask, shift);
    for (;;) {  // load ≤ ¼: always terminates
      const unsigned long long prev = atomicCAS(&table[idx], kEmpty, k);
      if (prev == kEmpty || prev == k) break;
      idx = (idx + 1) & mask;
    }
  }
}

// ---- output: bits [pos, pos + cnt) := low cnt bits of `word`, every other bit preserved -----
__device__ __forceinline__ void put_bits(uint8_t* __restrict__ bm, int64_t pos, unsigned long long word, int cnt) {
  const uintptr_t addr = (uintptr_t)bm + (uintptr_t)(pos >> 3);
  const int sub = (int)(pos & 7);
  if (cnt == 64 && sub == 0 && (addr & 7) == 0) {
    *(unsigned long long*)addr = word;
    return;
  }
  // general position: up to three aligned 32-bit words, atomics because a neighbouring chunk may
  // own the other bits of the same word
  const uintptr_t base = addr & ~(uintptr_t)3;
  int shift = (int)((addr - base) * 8) + sub;  // 0..31
  const unsigned long long m = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1);
  word &= m;
  unsigned* w = (unsigned*)base;
  // 96-bit window
  const unsigned long long mlo = m << shift, vlo = word << shift;
  const unsigned long long mhi = shift ? (m >> (64 - shift)) : 0ull, vhi = shift ? (word >> (64 - shift)) : 0ull;
  const unsigned m0 = (unsigned)mlo, m1 = (unsigned)(mlo >> 32), m2 = (unsigned)mhi;
  const unsigned v0 = (unsigned)vlo, v1 = (unsigned)(vlo >> 32), v2 = (unsigned)vhi;
  if (m0) { atomicAnd(&w[0], ~m0); if (v0) atomicOr(&w[0], v0); }
  if (m1) { atomicAnd(&w[1], ~m1); if (v1) atomicOr(&w[1], v1); }
  if (m2) { atomicAnd(&w[2], ~m2); if (v2) atomicOr(&w[2], v2); }
}

// ---- probe -----------------------------------------------------------------------------------
// MODE 0: bitmap in LDS (W ≤ 2) · 1: hash table in LDS (≤ 4096 slots, 32 KiB) · 2: hash table in HBM · 3: hash table in
// LDS, 16 384 slots = 128 KiB of gfx950's 160 KiB, one 1024-thread workgroup per CU (sets of up to 4096 keys: 2–4× faster
// than probing the table through L2).
// One row per lane is instruction-bound unless the per-row work is tiny (a wave64 VALU instruction
// costs 4 cycles on a 16-lane SIMD: ≈ 50 instructions per 64-row chunk is the HBM-bound budget for
// 8-byte rows), so: ONE ballot per chunk, the null-behaviour table folded into three wave-uniform
// flags and scalar mask arithmetic, results of a group of chunks gathered with v_writelane and
// stored by one instruction (ALIGNED: output word-aligned; otherwise lane 0 merges bits).
constexpr int kLdsSlotsBig = 16384;
constexpr int kBlockBig = 1024;

template <int W, int MODE, bool HAS_VALID, bool ALIGNED, int BLOCK = (MODE == 3 ? kBlockBig : kBlock)>
__global__ __launch_bounds__(BLOCK) void is_in_kernel(const void* __restrict__ values, const uint8_t* __restrict__ valid, int64_t off,
                                                        int64_t n, const unsigned long long* __restrict__ table, unsigned mask, int shift,
                                                        const unsigned* __restrict__ bits, const unsigned* __restrict__ flags,
                                                        int null_behavior, uint8_t* __restrict__ out_data, uint8_t* __restrict__ out_valid,
                                                        int64_t out_off) {
  __shared__ unsigned long long s_tab[MODE == 1 ? kLdsSlots : (MODE == 3 ? kLdsSlotsBig : 1)];
  __shared__ unsigned s_bits[MODE == 0 ? (W == 1 ? 8 : 2048) : 1];
  if (MODE == 0) {
    for (int i = threadIdx.x; i < (W == 1 ? 8 : 2048); i += BLOCK) s_bits[i] = bits[i];
    __syncthreads();
  } else if (MODE == 1 || MODE == 3) {
    for (unsigned i = threadIdx.x; i <= mask; i += BLOCK) s_tab[i] = table[i];
    __syncthreads();
  }
  const unsigned fl = flags[0];
  const bool set_has_null = (fl & kFlagSetHasNull) && null_behavior != AH_NULL_SKIP;
  // valid row: (found, found ∨ vmiss) · null row: (dnull, vnull)
  const bool vmiss = !(null_behavior == AH_NULL_INCONCLUSIVE && set_has_null);
  const bool dnull = null_behavior == AH_NULL_MATCH && set_has_null;
  const bool vnull = dnull || null_behavior == AH_NULL_SKIP || (!set_has_null && null_behavior == AH_NULL_MATCH);
  const bool all_ones_in_set = fl & kFlagAllOnesKey;
  const int lane = threadIdx.x & 63;
  const int64_t nchunks = (n + 63) >> 6;
  const int64_t ngroups = (nchunks + kChunks - 1) / kChunks;
  const int64_t wave_stride = (int64_t)gridDim.x * (BLOCK / 64);
  unsigned long long* __restrict__ od64 = (unsigned long long*)out_data + (out_off >> 6);   // used when ALIGNED
  unsigned long long* __restrict__ ov64 = (unsigned long long*)out_valid + (out_off >> 6);
  for (int64_t g = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); g < ngroups; g += wave_stride) {
    const int64_t c0 = g * kChunks;
    unsigned long long k[kChunks], vin[kChunks];
#pragma unroll
    for (int u = 0; u < kChunks; u++) {
      const int64_t row = (c0 + u) * 64 + lane;
      k[u] = row < n ? load_key<W>(values, row) : 0ull;
      if (HAS_VALID) {
        const int64_t left = n - (c0 + u) * 64;
        vin[u] = ah_load_bits64(valid, off + (c0 + u) * 64, left >= 64 ? 64 : (left > 0 ? (int)left : 0));
      }
    }
    unsigned dlo = 0, dhi = 0, vlo = 0, vhi = 0;  // lane u holds the group's u-th result words
#pragma unroll
    for (int u = 0; u < kChunks; u++) {
      const int64_t left = n - (c0 + u) * 64;
      if (left <= 0) break;  // wave-uniform
      const unsigned long long range = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
      const unsigned long long in_valid = HAS_VALID ? vin[u] : range;
      bool found = false;
      if ((in_valid >> lane) & 1ull) {
        const unsigned long long key = k[u];
        if (MODE == 0) {
          found = (s_bits[key >> 5] >> (key & 31)) & 1u;
        } else if (W == 8 && key == kEmpty) {
          found = all_ones_in_set;
        } else {
          unsigned idx = slot_of(key, mask, shift);
          for (;;) {  // load ≤ ¼: a miss ends after 1.4 slots on average
            const unsigned long long sl = (MODE == 1 || MODE == 3) ? s_tab[idx] : table[idx];
            if (sl == key) { found = true; break; }
            if (sl == kEmpty) break;
            idx = (idx + 1) & mask;
          }
        }
      }
      const unsigned long long fw = __ballot(found);
      const unsigned long long nulls = ~in_valid & range;
      const unsigned long long dword = fw | (dnull ? nulls : 0ull);
      const unsigned long long vword = fw | (vmiss ? in_valid : 0ull) | (vnull ? nulls : 0ull);
      if (ALIGNED && left >= 64) {
        const bool mine = lane == u;  // the words are wave-uniform (SGPRs): one compare + four selects
        dlo = mine ? (unsigned)dword : dlo;
        dhi = mine ? (unsigned)(dword >> 32) : dhi;
        vlo = mine ? (unsigned)vword : vlo;
        vhi = mine ? (unsigned)(vword >> 32) : vhi;
      } else if (lane == 0) {
        put_bits(out_data, out_off + (c0 + u) * 64, dword, left >= 64 ? 64 : (int)left);
        put_bits(out_valid, out_off + (c0 + u) * 64, vword, left >= 64 ? 64 : (int)left);
      }
    }
    if (ALIGNED) {
      const int64_t full = (n >> 6) - c0;  // whole chunks of this group
      if (lane < kChunks && lane < full) {
        od64[c0 + lane] = ((unsigned long long)dhi << 32) | dlo;
        ov64[c0 + lane] = ((unsigned long long)vhi << 32) | vlo;
      }
    }
  }
}

template <int W, int MODE>
void launch_probe(ah_ctx* c, unsigned grid, const void* values, const uint8_t* valid, int64_t off, int64_t n, const unsigned long long* table,
                  unsigned mask, int shift, const unsigned* bits, const unsigned* flags, int nb, uint8_t* out_data, uint8_t* out_valid,
                  int64_t out_off) {
  const bool aligned = (out_off & 63) == 0 && (((uintptr_t)out_data | (uintptr_t)out_valid) & 7) == 0;
#define AH_PROBE(HV, AL) is_in_kernel<W, MODE, HV, AL><<<grid, (MODE == 3 ? kBlockBig : kBlock), 0, c->stream>>>(values, valid, off, n, table, mask, shift, bits, flags, nb, \
                                                                                  out_data, out_valid, out_off)
  if (valid) { if (aligned) AH_PROBE(true, true); else AH_PROBE(true, false); }
  else { if (aligned) AH_PROBE(false, true); else AH_PROBE(false, false); }
#undef AH_PROBE
}

template <int W>
int run_is_in(ah_ctx* c, const void* values, const uint8_t* valid, int64_t off, int64_t n, const void* set_values, const uint8_t* set_valid,
              int64_t set_off, int64_t set_n, int null_behavior, uint8_t* out_data, uint8_t* out_valid, int64_t out_off) {
  const bool bitmap = W <= 2;
  unsigned long long cap = 64;  // load ≤ ¼: short probe chains matter more than table bytes
  int log2cap = 6;
  while (cap < 4ull * (unsigned long long)set_n) { cap <<= 1; log2cap++; }
  if (cap > (1ull << 31)) return ah_fail(c, AH_ENOTIMPL, "is_in: value set too large (%lld)", (long long)set_n);
  const int shift = 32 - log2cap;
  const size_t table_bytes = bitmap ? 8192 : (size_t)cap * 8;
  void* scratch;
  int rc = ah_scratch_reserve(c, 128 + table_bytes, &scratch);
  if (rc != AH_OK) return rc;
  unsigned* flags = (unsigned*)scratch;
  void* tab = (uint8_t*)scratch + 128;
  AH_HIP(c, hipMemsetAsync(flags, 0, 128, c->stream));
  AH_HIP(c, hipMemsetAsync(tab, bitmap ? 0 : 0xFF, table_bytes, c->stream));
  if (set_n > 0) {
    const unsigned g = ah_stream_grid(c, ah_ceil_div(set_n, kBlock), 8);
    if constexpr (W <= 2) build_bitmap_kernel<W><<<g, kBlock, 0, c->stream>>>(set_values, set_valid, set_off, set_n, (unsigned*)tab, flags);
    else build_hash_kernel<W><<<g, kBlock, 0, c->stream>>>(set_values, set_valid, set_off, set_n, (unsigned long long*)tab, (unsigned)(cap - 1), shift, flags);
    AH_LAUNCH_CHECK(c);
  }
  const int64_t nchunks = ah_ceil_div(n, 64);
  const unsigned grid = ah_stream_grid(c, ah_ceil_div(ah_ceil_div(nchunks, kChunks), kBlock / 64), 8);
  // tunable, for measurements: 0 keeps mid-size sets (1025 … 4096 keys) on the HBM table
  static const bool big_lds = !(getenv("ARROWHIP_ISIN_BIG_LDS") && atoi(getenv("ARROWHIP_ISIN_BIG_LDS")) == 0);
  if constexpr (W <= 2) {
    launch_probe<W, 0>(c, grid, values, valid, off, n, nullptr, 0, 0, (const unsigned*)tab, flags, null_behavior, out_data, out_valid, out_off);
  } else if (cap <= (unsigned long long)kLdsSlots) {
    launch_probe<W, 1>(c, grid, values, valid, off, n, (const unsigned long long*)tab, (unsigned)(cap - 1), shift, nullptr, flags, null_behavior,
                       out_data, out_valid, out_off);
  } else if (cap <= (unsigned long long)kLdsSlotsBig && big_lds) {
    const unsigned per_cu = (unsigned)c->num_cu;  // 128 KiB of LDS each: one workgroup per CU
    const unsigned need = (unsigned)ah_ceil_div(ah_ceil_div(nchunks, kChunks), kBlockBig / 64);
    launch_probe<W, 3>(c, need < per_cu ? need : per_cu, values, valid, off, n, (const unsigned long long*)tab, (unsigned)(cap - 1), shift, nullptr, flags,
                       null_behavior, out_data, out_valid, out_off);
  } else {
    launch_probe<W, 2>(c, grid, values, valid, off, n, (const unsigned long long*)tab, (unsigned)(cap - 1), shift, nullptr, flags, null_behavior,
                       out_data, out_valid, out_off);
  }
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

}  // namespace

AH_EXPORT int ah_is_in(ah_ctx* c, int byte_width, const void* values, const uint8_t* valid, int64_t off, int64_t n, const void* set_values,
                       const uint8_t* set_valid, int64_t set_off, int64_t set_n, int null_behavior, uint8_t* out_data, uint8_t* out_valid,
                       int64_t out_bit_offset) {
  AH_ENTER(c);
  if (n < 0 || off < 0 || set_n < 0 || set_off < 0 || out_bit_offset < 0) return ah_fail(c, AH_EINVALID, "is_in: negative length/offset");
  if (null_behavior < AH_NULL_MATCH || null_behavior > AH_NULL_INCONCLUSIVE) return ah_fail(c, AH_EINVALID, "is_in: bad null matching behavior %d", null_behavior);
  if (n == 0) return AH_OK;
  if (!values || !out_data || !out_valid || (set_n > 0 && !set_values)) return ah_fail(c, AH_EINVALID, "is_in: null buffer");
  if (((uintptr_t)values | (uintptr_t)set_values) & (uintptr_t)(byte_width - 1)) return ah_fail(c, AH_EINVALID, "is_in: buffer not element-aligned");
  switch (byte_width) {
    case 1: return run_is_in<1>(c, values, valid, off, n, set_values, set_valid, set_off, set_n, null_behavior, out_data, out_valid, out_bit_offset);
    case 2: return run_is_in<2>(c, values, valid, off, n, set_values, set_valid, set_off, set_n, null_behavior, out_data, out_valid, out_bit_offset);
    case 4: return run_is_in<4>(c, values, valid, off, n, set_values, set_valid, set_off, set_n, null_behavior, out_data, out_valid, out_bit_offset);
    case 8: return run_is_in<8>(c, values, valid, off, n, set_values, set_valid, set_off, set_n, null_behavior, out_data, out_valid, out_bit_offset);
  }
  return ah_fail(c, AH_ENOTIMPL, "is_in: fixed-width values of 1, 2, 4 or 8 bytes only (got %d)", byte_width);
}
