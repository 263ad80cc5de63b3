/*
 * orc_hash.c — numeric memo table, unique / dictionary_encode, and the
 * group-by-sum definition (TEST INFRASTRUCTURE, see oracle.h).
 *
 * Reference:
 *   hashInt:      internal/hashing/hash_funcs.go:60-67
 *                 h = bswap64(multipliers[alg] * v)
 *   HashTable:    internal/hashing/xxh3_memo_table_types.go:45-187
 *                 open addressing, sentinel h == 0 remapped to 42 (:111-116),
 *                 idx = h & mask, perturb = (h >> 5) + 1,
 *                 idx = (idx + perturb) & mask, perturb = (perturb >> 5) + 1
 *                 (:123-152); capacity power of two ≥ 32 (:53-60); grows ×4 once
 *                 size*2 >= cap (:109,169-179)
 *   Table[T]:     :189-294 (InsertOrGet :283-294, GetOrInsertNull :231-238,
 *                 Size :218-225)
 *   driver:       kernels/vector_hash.go:359-385 (doAppendNumeric),
 *                 :145-241 (dictionaryEncodeAction), :721-741 (uniqueFinalize)
 *   Int64 AND Float64 columns both use Table[uint64] on the raw bit pattern
 *   (vector_hash.go:604-607,690-693) — so keys here are uint64 bit patterns.
 *
 * group-by sum has NO reference implementation; its oracle is
 * the definition in DESIGN.md: dense group ids in first-seen order (exactly the
 * dictionary_encode ids), per-group accumulation in row order.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int bget_opt(const uint8_t* b, int64_t i) { return b == 0 ? 1 : (b[i >> 3] >> (i & 7)) & 1; }
static inline void bset(uint8_t* b, int64_t i) { b[i >> 3] |= (uint8_t)(1u << (i & 7)); }

uint64_t orc_hash_int(uint64_t v, uint64_t alg) {
  static const uint64_t mult[2] = {11400714785074694791ull, 14029467366897019727ull};
  return __builtin_bswap64(mult[alg & 1] * v);
}

typedef struct { uint64_t h; uint64_t val; int32_t memo_idx; } entry_t;
typedef struct { uint64_t cap, mask, size; entry_t* e; int32_t null_idx; } table_t;

static uint64_t next_pow2(uint64_t x) { /* bitutil.NextPowerOf2: 1 << bits.Len(x) */
  uint64_t p = 1; while (x) { p <<= 1; x >>= 1; } return p;
}
static void table_init(table_t* t, uint64_t cap) {
  uint64_t c = next_pow2(cap < 32 ? 32 : cap);
  t->cap = c; t->mask = c - 1; t->size = 0; t->null_idx = -1;
  t->e = (entry_t*)calloc(c, sizeof(entry_t));
}
static uint64_t fix_hash(uint64_t h) { return h == 0 ? 42 : h; }
/* lookup (:123-152); cmp_val < 0 means "never equal" (used by upsize) */
static uint64_t table_lookup(const entry_t* e, uint64_t mask, uint64_t h, int use_cmp, uint64_t val, int* found) {
  h = fix_hash(h);
  uint64_t idx = h & mask, perturb = (h >> 5) + 1;
  for (;;) {
    const entry_t* s = &e[idx];
    if (s->h == h && use_cmp && s->val == val) { *found = 1; return idx; }
    if (s->h == 0) { *found = 0; return idx; }
    idx = (idx + perturb) & mask;
    perturb = (perturb >> 5) + 1;
  }
}
static void table_upsize(table_t* t, uint64_t newcap) {
  entry_t* ne = (entry_t*)calloc(newcap, sizeof(entry_t));
  for (uint64_t i = 0; i < t->cap; i++) {
    if (t->e[i].h != 0) { int f; uint64_t idx = table_lookup(ne, newcap - 1, t->e[i].h, 0, 0, &f); ne[idx] = t->e[i]; }
  }
  free(t->e); t->e = ne; t->cap = newcap; t->mask = newcap - 1;
}
static int table_size(const table_t* t) { return (int)t->size + (t->null_idx >= 0 ? 1 : 0); }
static int table_insert_or_get(table_t* t, uint64_t val, int* found) {
  uint64_t h = orc_hash_int(val, 0);
  uint64_t idx = table_lookup(t->e, t->mask, h, 1, val, found);
  if (*found) return t->e[idx].memo_idx;
  int id = table_size(t);
  t->e[idx].h = fix_hash(h); t->e[idx].val = val; t->e[idx].memo_idx = id;
  t->size++;
  if (t->size * 2 >= t->cap) table_upsize(t, t->cap * 4);
  return id;
}
static int table_get_or_insert_null(table_t* t) {
  if (t->null_idx < 0) t->null_idx = table_size(t);
  return t->null_idx;
}

int orc_hash_u64_encode(const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls,
                        int32_t* out_ids, uint8_t* out_ids_valid, uint64_t* out_dict,
                        int64_t* out_ndict, int32_t* out_null_id) {
  table_t t; table_init(&t, 0);
  if (out_ids_valid) memset(out_ids_valid, 0, (size_t)((n + 7) / 8));
  for (int64_t i = 0; i < n; i++) {
    if (bget_opt(valid, off + i)) {
      int f; int id = table_insert_or_get(&t, keys[i], &f);
      if (out_ids) out_ids[i] = id;
      if (out_ids_valid) bset(out_ids_valid, i);
    } else if (encode_nulls) {
      /* unique / NullEncodingEncode: null occupies the id at which it was first seen */
      int id = table_get_or_insert_null(&t);
      if (out_ids) out_ids[i] = id;
      if (out_ids_valid) bset(out_ids_valid, i);
    } else {
      /* NullEncodingMask: appendIndex(0, false) (vector_hash.go:169-186,224-230) */
      if (out_ids) out_ids[i] = 0;
    }
  }
  int nd = table_size(&t);
  if (out_dict) {
    /* GetDictArrayData (arrow/array/util.go:321-390): values by memo index; the
     * null slot (if any) keeps the zero of the fresh buffer */
    for (int i = 0; i < nd; i++) out_dict[i] = 0;
    for (uint64_t i = 0; i < t.cap; i++) if (t.e[i].h != 0) out_dict[t.e[i].memo_idx] = t.e[i].val;
  }
  if (out_ndict) *out_ndict = nd;
  if (out_null_id) *out_null_id = t.null_idx;
  free(t.e);
  return ORC_OK;
}

/* ---- binary / string keys ---------------------------------------------------------------------
 * doAppendBinary (kernels/vector_hash.go:288-325) over BinaryMemoTable
 * (internal/hashing/xxh3_memo_table.go:248-341): InsertOrGet = lookup by (hash, bytes.Equal against
 * the builder's value), else append to the builder — the memo index is the builder position, i.e. the
 * order of FIRST OCCURRENCE; GetOrInsertNull appends a null (zero length) at Size().
 * Hash (hash_funcs.go:86-124) only decides where an entry sits in the open-addressing table: for
 * ≤ 16 bytes it is restated below; for longer values the reference calls github.com/zeebo/xxh3 v1.1.0
 * (go.mod:47), whose source is NOT in the snapshot — FNV-1a stands in, which cannot change any memo
 * index (equality is decided by the bytes).  The dictionary is not materialised here: out_first_rows[id]
 * = the row whose bytes the builder holds at position id, so dictionary = take(values, first_rows). */
static uint64_t hash_bytes_ref(const uint8_t* b, uint32_t n) {
  if (n <= 16) {
    if (n > 8) {
      uint64_t x, y; memcpy(&x, b + n - 8, 8); memcpy(&y, b, 8);
      return (uint64_t)n ^ orc_hash_int(x, 0) ^ orc_hash_int(y, 1);
    }
    if (n >= 4) {
      uint32_t x, y; memcpy(&x, b + n - 4, 4); memcpy(&y, b, 4);
      return (uint64_t)n ^ orc_hash_int(x, 0) ^ orc_hash_int(y, 1);
    }
    if (n > 0) {
      uint32_t x = (n << 24) ^ ((uint32_t)b[0] << 16) ^ ((uint32_t)b[n / 2] << 8) ^ (uint32_t)b[n - 1];
      return orc_hash_int(x, 0);
    }
    return 1;
  }
  uint64_t h = 14695981039346656037ull;  /* stand-in for xxh3.Hash(b) */
  for (uint32_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h + 1609587929392839161ull;     /* + exprimes[0] (:87) */
}

#define BIN_ENCODE_BODY(OT)                                                                               \
  const OT* o = (const OT*)offsets + off;                                                                 \
  table_t t; table_init(&t, 0);                                                                           \
  int64_t* rows = out_first_rows; /* memo index → row (the "builder") */                                  \
  if (out_ids_valid) memset(out_ids_valid, 0, (size_t)((n + 7) / 8));                                     \
  for (int64_t i = 0; i < n; i++) {                                                                       \
    if (bget_opt(valid, off + i)) {                                                                       \
      const uint8_t* v = data + o[i]; const int64_t len = (int64_t)(o[i + 1] - o[i]);                     \
      uint64_t h = fix_hash(hash_bytes_ref(v, (uint32_t)len));                                            \
      uint64_t idx = h & t.mask, perturb = (h >> 5) + 1; int id = -1;                                     \
      for (;;) {                                                                                          \
        entry_t* s = &t.e[idx];                                                                           \
        if (s->h == h) {                                                                                  \
          const int64_t r = rows[s->memo_idx];                                                            \
          if ((int64_t)(o[r + 1] - o[r]) == len && memcmp(data + o[r], v, (size_t)len) == 0) { id = s->memo_idx; break; } \
        }                                                                                                 \
        if (s->h == 0) break;                                                                             \
        idx = (idx + perturb) & t.mask; perturb = (perturb >> 5) + 1;                                     \
      }                                                                                                   \
      if (id < 0) {                                                                                       \
        id = table_size(&t); rows[id] = i;                                                                \
        t.e[idx].h = h; t.e[idx].val = 0; t.e[idx].memo_idx = id; t.size++;                               \
        if (t.size * 2 >= t.cap) table_upsize(&t, t.cap * 4);                                             \
      }                                                                                                   \
      if (out_ids) out_ids[i] = id;                                                                       \
      if (out_ids_valid) bset(out_ids_valid, i);                                                          \
    } else if (encode_nulls) {                                                                            \
      if (t.null_idx < 0) { t.null_idx = table_size(&t); rows[t.null_idx] = i; }                          \
      if (out_ids) out_ids[i] = t.null_idx;                                                               \
      if (out_ids_valid) bset(out_ids_valid, i);                                                          \
    } else if (out_ids) out_ids[i] = 0;                                                                   \
  }                                                                                                       \
  if (out_ndict) *out_ndict = table_size(&t);                                                             \
  if (out_null_id) *out_null_id = t.null_idx;                                                             \
  free(t.e);                                                                                              \
  return ORC_OK;

int orc_hash_binary_encode(int offset_width, const void* offsets, const uint8_t* data, const uint8_t* valid, int64_t off, int64_t n,
                           int encode_nulls, int32_t* out_ids, uint8_t* out_ids_valid, int64_t* out_first_rows,
                           int64_t* out_ndict, int32_t* out_null_id) {
  if (offset_width == 4) { BIN_ENCODE_BODY(int32_t) }
  BIN_ENCODE_BODY(int64_t)
}

#define HASH_SUM_BODY(VT, ACC_T)                                                                  \
  table_t t; table_init(&t, 0);                                                                   \
  int64_t ng = 0;                                                                                 \
  for (int64_t i = 0; i < n; i++) {                                                               \
    int id;                                                                                       \
    if (bget_opt(kvalid, koff + i)) { int f; id = table_insert_or_get(&t, keys[i], &f); }         \
    else id = table_get_or_insert_null(&t);                                                       \
    if (id >= ng) { out_sums[id] = 0; out_counts[id] = 0; out_keys[id] = bget_opt(kvalid, koff + i) ? keys[i] : 0; if (out_first_rows) out_first_rows[id] = i; ng = id + 1; } \
    if (bget_opt(vvalid, voff + i)) { out_sums[id] = (ACC_T)(out_sums[id] + (ACC_T)vals[i]); out_counts[id]++; } \
  }                                                                                               \
  *out_ngroups = ng;                                                                              \
  if (out_null_group) *out_null_group = t.null_idx;                                               \
  free(t.e);                                                                                      \
  return ORC_OK;

int orc_hash_sum_f64(const uint64_t* keys, const uint8_t* kvalid, int64_t koff,
                     const double* vals, const uint8_t* vvalid, int64_t voff, int64_t n,
                     uint64_t* out_keys, double* out_sums, int64_t* out_counts, int64_t* out_first_rows,
                     int64_t* out_ngroups, int32_t* out_null_group) {
  HASH_SUM_BODY(double, double)
}

int orc_hash_sum_i64(const uint64_t* keys, const uint8_t* kvalid, int64_t koff,
                     const int64_t* vals, const uint8_t* vvalid, int64_t voff, int64_t n,
                     uint64_t* out_keys, int64_t* out_sums, int64_t* out_counts, int64_t* out_first_rows,
                     int64_t* out_ngroups, int32_t* out_null_group) {
  uint64_t* usums = (uint64_t*)out_sums;
  (void)usums;
  HASH_SUM_BODY(int64_t, uint64_t)
}
