/*
 * orc_setlookup.c — is_in restated (TEST INFRASTRUCTURE).
 *
 * Reference: arrow/compute/internal/kernels/scalar_set_lookup.go
 *   SetLookupState.Init :192-244 — the value set goes into a memo table keyed on the raw bits of
 *     the fixed-width value (uint8/16/32/64 by byte width, :106-133: floats compare by bit pattern);
 *     NullIndex is set only when the set holds a null AND NullBehavior != Skip (:239-242)
 *   isInKernelExec :374-413 — per row:
 *     valid value: found → (true, valid); else Inconclusive ∧ set-has-null → (false, NULL); else (false, valid)
 *     null value:  Match ∧ set-has-null → (true, valid); Skip ∨ (Match ∧ ¬set-has-null) → (false, valid);
 *                  otherwise (EmitNull, Inconclusive) → (false, NULL)
 * Membership here is a sort + binary search: obviously correct, no hash table to get wrong.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : x > y;
}
static uint64_t load_key(int w, const void* p, int64_t i) {
  switch (w) {
    case 1: return ((const uint8_t*)p)[i];
    case 2: return ((const uint16_t*)p)[i];
    case 4: return ((const uint32_t*)p)[i];
    default: return ((const uint64_t*)p)[i];
  }
}
static inline int bget_opt(const uint8_t* b, int64_t i) { return b == 0 ? 1 : (b[i >> 3] >> (i & 7)) & 1; }
static inline void bput(uint8_t* b, int64_t i, int v) {
  if (v) b[i >> 3] |= (uint8_t)(1u << (i & 7)); else b[i >> 3] &= (uint8_t)~(1u << (i & 7));
}

int orc_is_in(int byte_width, const void* values, const uint8_t* valid, int64_t off, int64_t n,
              const void* set_values, const uint8_t* set_valid, int64_t set_off, int64_t set_n, int null_behavior,
              uint8_t* out_data, uint8_t* out_valid, int64_t out_off) {
  if (byte_width != 1 && byte_width != 2 && byte_width != 4 && byte_width != 8) return ORC_EINVALID;
  uint64_t* keys = (uint64_t*)malloc((size_t)(set_n > 0 ? set_n : 1) * 8);
  int64_t nk = 0;
  int set_has_null = 0;
  for (int64_t i = 0; i < set_n; i++) {
    if (bget_opt(set_valid, set_off + i)) keys[nk++] = load_key(byte_width, set_values, i);
    else set_has_null = 1;
  }
  qsort(keys, (size_t)nk, 8, cmp_u64);
  if (null_behavior == ORC_NULL_SKIP) set_has_null = 0;  /* NullIndex stays -1 (:239-242) */
  for (int64_t i = 0; i < n; i++) {
    int d, v;
    if (bget_opt(valid, off + i)) {
      uint64_t k = load_key(byte_width, values, i);
      int found = nk > 0 && bsearch(&k, keys, (size_t)nk, 8, cmp_u64) != 0;
      if (found) { d = 1; v = 1; }
      else if (null_behavior == ORC_NULL_INCONCLUSIVE && set_has_null) { d = 0; v = 0; }
      else { d = 0; v = 1; }
    } else {
      if (null_behavior == ORC_NULL_MATCH && set_has_null) { d = 1; v = 1; }
      else if (null_behavior == ORC_NULL_SKIP || (!set_has_null && null_behavior == ORC_NULL_MATCH)) { d = 0; v = 1; }
      else { d = 0; v = 0; }
    }
    bput(out_data, out_off + i, d);
    bput(out_valid, out_off + i, v);
  }
  free(keys);
  return ORC_OK;
}
