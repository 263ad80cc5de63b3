/*
 * orc_sort.c — sort_indices on one numeric array restated (TEST INFRASTRUCTURE).
 *
 * Reference: arrow/compute/internal/kernels
 *   SortIndices :388-481 (single key, one chunk) → arraySortOneColumnRange (vector_sort_internal.go:252-273):
 *     stable partition of the nulls to the end / start (partitionNullsOnly), stable partition of the
 *     NaNs next to them (partitionNullLikes :90-140: NullsAtStart → [nulls, NaNs, rest], NullsAtEnd →
 *     [rest, NaNs, nulls]; NaN placement follows NullPlacement, not Order), then
 *     slices.SortStableFunc of the rest with compareOrdered (vector_sort_support.go:88-94:
 *     cmp.Compare, negated for Descending — ties keep their input order in BOTH orders, −0.0 == +0.0)
 * Restated as: stable merge sort of the finite rows with exactly that comparator.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static inline int bget_opt(const uint8_t* b, int64_t i) { return b == 0 ? 1 : (b[i >> 3] >> (i & 7)) & 1; }

static int g_type, g_desc;
static const void* g_vals;

static int cmp_rows(uint64_t a, uint64_t b) {
  int c;
#define CMP(T) { T x = ((const T*)g_vals)[a], y = ((const T*)g_vals)[b]; c = x < y ? -1 : (x > y ? 1 : 0); }
  switch (g_type) {
    case ORC_UINT8: CMP(uint8_t) break; case ORC_INT8: CMP(int8_t) break;
    case ORC_UINT16: CMP(uint16_t) break; case ORC_INT16: CMP(int16_t) break;
    case ORC_UINT32: CMP(uint32_t) break; case ORC_INT32: CMP(int32_t) break;
    case ORC_UINT64: CMP(uint64_t) break; case ORC_INT64: CMP(int64_t) break;
    case ORC_FLOAT32: CMP(float) break; default: CMP(double) break;
  }
#undef CMP
  return g_desc ? -c : c;
}

static void merge_sort(uint64_t* a, uint64_t* tmp, int64_t lo, int64_t hi) {
  if (hi - lo <= 1) return;
  int64_t mid = lo + (hi - lo) / 2;
  merge_sort(a, tmp, lo, mid);
  merge_sort(a, tmp, mid, hi);
  int64_t i = lo, j = mid, k = lo;
  while (i < mid && j < hi) tmp[k++] = cmp_rows(a[j], a[i]) < 0 ? a[j++] : a[i++];  /* ties: left first */
  while (i < mid) tmp[k++] = a[i++];
  while (j < hi) tmp[k++] = a[j++];
  memcpy(a + lo, tmp + lo, (size_t)(hi - lo) * 8);
}

int orc_sort_indices(int type, const void* values, const uint8_t* valid, int64_t off, int64_t n, int descending,
                     int nulls_at_start, uint64_t* out_indices) {
  if (n == 0) return ORC_OK;
  uint64_t* fin = (uint64_t*)malloc((size_t)n * 8);
  uint64_t* nan = (uint64_t*)malloc((size_t)n * 8);
  uint64_t* nul = (uint64_t*)malloc((size_t)n * 8);
  uint64_t* tmp = (uint64_t*)malloc((size_t)n * 8);
  int64_t nf = 0, nn = 0, nz = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!bget_opt(valid, off + i)) { nul[nz++] = (uint64_t)i; continue; }
    int is_nan = 0;
    if (type == ORC_FLOAT32) { float v = ((const float*)values)[i]; is_nan = v != v; }
    if (type == ORC_FLOAT64) { double v = ((const double*)values)[i]; is_nan = v != v; }
    if (is_nan) nan[nn++] = (uint64_t)i; else fin[nf++] = (uint64_t)i;
  }
  g_type = type; g_desc = descending; g_vals = values;
  merge_sort(fin, tmp, 0, nf);
  int64_t k = 0;
  if (nulls_at_start) {
    memcpy(out_indices + k, nul, (size_t)nz * 8); k += nz;
    memcpy(out_indices + k, nan, (size_t)nn * 8); k += nn;
    memcpy(out_indices + k, fin, (size_t)nf * 8);
  } else {
    memcpy(out_indices + k, fin, (size_t)nf * 8); k += nf;
    memcpy(out_indices + k, nan, (size_t)nn * 8); k += nn;
    memcpy(out_indices + k, nul, (size_t)nz * 8);
  }
  free(fin); free(nan); free(nul); free(tmp);
  return ORC_OK;
}

/* ---- several keys (record batch): radixRecordBatchSortRange, vector_sort_internal.go:167-205 —
 * per key: [rest, NaNs, nulls] or [nulls, NaNs, rest], the rest ordered by value; NaN ties, null ties and
 * value ties are broken by the next key; stable.  Restated as ONE stable merge sort with the
 * lexicographic comparator over (category, value) per key. */
typedef struct { int type; const void* values; const uint8_t* valid; int64_t off; int desc; int nulls_first; } orc_key;
static const orc_key* g_keys;
static int g_nkeys;

static int cat_of(const orc_key* k, uint64_t r) {
  if (!bget_opt(k->valid, k->off + (int64_t)r)) return 2;
  if (k->type == ORC_FLOAT32) { float v = ((const float*)k->values)[r]; if (v != v) return 1; }
  if (k->type == ORC_FLOAT64) { double v = ((const double*)k->values)[r]; if (v != v) return 1; }
  return 0;
}

static int cmp_multi(uint64_t a, uint64_t b) {
  for (int i = 0; i < g_nkeys; i++) {
    const orc_key* k = &g_keys[i];
    int ca = cat_of(k, a), cb = cat_of(k, b);
    if (k->nulls_first) { ca = 2 - ca; cb = 2 - cb; }
    if (ca != cb) return ca < cb ? -1 : 1;
    if (cat_of(k, a) != 0) continue;  /* both NaN or both null: tie on this key */
    g_type = k->type; g_desc = k->desc; g_vals = k->values;
    int c = cmp_rows(a, b);
    if (c) return c;
  }
  return 0;
}

static void merge_sort_multi(uint64_t* a, uint64_t* tmp, int64_t lo, int64_t hi) {
  if (hi - lo <= 1) return;
  int64_t mid = lo + (hi - lo) / 2;
  merge_sort_multi(a, tmp, lo, mid);
  merge_sort_multi(a, tmp, mid, hi);
  int64_t i = lo, j = mid, k = lo;
  while (i < mid && j < hi) tmp[k++] = cmp_multi(a[j], a[i]) < 0 ? a[j++] : a[i++];
  while (i < mid) tmp[k++] = a[i++];
  while (j < hi) tmp[k++] = a[j++];
  memcpy(a + lo, tmp + lo, (size_t)(hi - lo) * 8);
}

int orc_sort_indices_multi(int nkeys, const int* types, const void* const* values, const uint8_t* const* valids, const int64_t* offs,
                           int64_t n, const int* descending, const int* nulls_at_start, uint64_t* out_indices) {
  if (nkeys < 1 || nkeys > 64) return ORC_EINVALID;
  orc_key keys[64];
  for (int i = 0; i < nkeys; i++) keys[i] = (orc_key){types[i], values[i], valids[i], offs[i], descending[i], nulls_at_start[i]};
  g_keys = keys; g_nkeys = nkeys;
  uint64_t* tmp = (uint64_t*)malloc((size_t)(n > 0 ? n : 1) * 8);
  for (int64_t i = 0; i < n; i++) out_indices[i] = (uint64_t)i;
  merge_sort_multi(out_indices, tmp, 0, n);
  free(tmp);
  return ORC_OK;
}
