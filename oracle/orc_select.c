/*
 * orc_select.c — Filter / Take restated per element (TEST INFRASTRUCTURE).
 *
 * Reference:
 *   output size:  kernels/vector_selection.go:57-81 (getFilterOutputSize)
 *   filter:       kernels/vector_selection.go:267-395 (primitiveFilterImpl),
 *                 :397-421 (filterWriter), :449-520 (PrimitiveFilter)
 *   take:         kernels/vector_selection.go:878-988 (primitiveTakeImpl),
 *                 :1144-1192 (takeIdxDispatch, PrimitiveTake)
 *   bounds check: kernels/helpers.go:929-981 (checkIndexBounds)
 *   take indices: kernels/vector_selection.go:102-236 (GetTakeIndices)
 *
 * Payload rules the block state machine implements, restated per element
 * (SURVEY.md §8a a7/a8):
 *   filter: selected slot (filter valid ∧ true) → value payload copied AS IS
 *           (even if the value is null) and out validity = value validity;
 *           filter-null slot under EMIT_NULLS → payload 0, validity 0;
 *           everything else dropped.  Output buffers are fresh and zeroed, so
 *           padding bits past n_out are 0.
 *   take:   out[i] = values[idx[i]] iff idx valid ∧ value valid, else payload 0
 *           and validity 0.
 * Callers decide (like PrimitiveFilter :486-488 / PrimitiveTake :1176) whether
 * a validity buffer exists at all: pass out_valid = NULL when neither input
 * has nulls.
 */
#include "oracle.h"
#include <string.h>

static inline int bget(const uint8_t* b, int64_t i) { return (b[i >> 3] >> (i & 7)) & 1; }
static inline int bget_opt(const uint8_t* b, int64_t i) { return b == 0 ? 1 : bget(b, i); }
static inline void bset(uint8_t* b, int64_t i) { b[i >> 3] |= (uint8_t)(1u << (i & 7)); }

int64_t orc_filter_count(const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel) {
  int64_t c = 0;
  for (int64_t i = 0; i < n; i++) {
    int fv = bget_opt(fvalid, foff + i), fd = bget(fdata, foff + i);
    /* :66-77: EmitNulls counts (data OR NOT valid); DropNulls counts (data AND valid) */
    c += null_sel == ORC_EMIT_NULLS ? (fd | !fv) : (fd & fv);
  }
  return c;
}

int orc_filter_primitive(int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                         const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                         void* out_values, uint8_t* out_valid, int64_t* out_len, int64_t* out_null_count) {
  if (byte_width != 1 && byte_width != 2 && byte_width != 4 && byte_width != 8) return ORC_EINVALID;
  const uint8_t* vin = (const uint8_t*)values;
  uint8_t* vout = (uint8_t*)out_values;
  int64_t n_out = orc_filter_count(fdata, fvalid, foff, n, null_sel);
  /* fresh zeroed buffers (preallocateData :83-93 → ctx.Allocate zero-fills) */
  memset(vout, 0, (size_t)(n_out * byte_width));
  if (out_valid) memset(out_valid, 0, (size_t)((n_out + 7) / 8));
  int64_t pos = 0, nulls = 0;
  for (int64_t i = 0; i < n; i++) {
    int fv = bget_opt(fvalid, foff + i), fd = bget(fdata, foff + i);
    if (fv && fd) {
      /* writeMaybeNull :293-297 — payload copied regardless of value validity */
      memcpy(vout + pos * byte_width, vin + i * byte_width, (size_t)byte_width);
      int valid = bget_opt(vvalid, voff + i);
      if (valid) { if (out_valid) bset(out_valid, pos); } else nulls++;
      pos++;
    } else if (!fv && null_sel == ORC_EMIT_NULLS) {
      /* WriteNull :417-421 — zero payload, validity bit stays 0 */
      nulls++;
      pos++;
    }
  }
  *out_len = n_out;
  if (out_null_count) *out_null_count = nulls;
  return ORC_OK;
}

static inline int load_index(const void* idx, int w, int is_signed, int64_t i, int64_t* sval, uint64_t* uval) {
  switch (w) {
    case 1: { uint8_t u = ((const uint8_t*)idx)[i]; *uval = u; *sval = is_signed ? (int8_t)u : (int64_t)u; return 1; }
    case 2: { uint16_t u = ((const uint16_t*)idx)[i]; *uval = u; *sval = is_signed ? (int16_t)u : (int64_t)u; return 1; }
    case 4: { uint32_t u = ((const uint32_t*)idx)[i]; *uval = u; *sval = is_signed ? (int32_t)u : (int64_t)u; return 1; }
    case 8: { uint64_t u = ((const uint64_t*)idx)[i]; *uval = u; *sval = (int64_t)u; return 1; }
  }
  return 0;
}

int orc_take_primitive(int byte_width, const void* values, const uint8_t* vvalid, int64_t voff, int64_t nvalues,
                       int idx_byte_width, int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff,
                       int64_t nidx, int bounds_check, void* out_values, uint8_t* out_valid,
                       int64_t* out_null_count, int64_t* bad_index) {
  /* 1, 2, 4, 8 = primitiveTakeImpl; any other width = FSBImpl (vector_selection.go:1997-2031: the same visit order, the value copied
   * as valueSize bytes, a null slot left as allocated, i.e. zero) */
  if (byte_width < 1 || byte_width > 4096) return ORC_EINVALID;
  if (idx_byte_width != 1 && idx_byte_width != 2 && idx_byte_width != 4 && idx_byte_width != 8) return ORC_EINDEX;
  const uint8_t* vin = (const uint8_t*)values;
  uint8_t* vout = (uint8_t*)out_values;
  int64_t s = 0; uint64_t u = 0;
  if (bounds_check) {
    /* helpers.go:929-957: only VALID index slots are checked (VisitSetBitRuns
     * over the index validity), error names the first offender in order:
     * (signed ∧ v < 0) ∨ (v ≥ 0 ∧ uint64(v) ≥ len(values)). */
    for (int64_t i = 0; i < nidx; i++) {
      if (!bget_opt(ivalid, ioff + i)) continue;
      load_index(idx, idx_byte_width, idx_signed, i, &s, &u);
      int oob = idx_signed ? (s < 0 || (uint64_t)s >= (uint64_t)nvalues) : (u >= (uint64_t)nvalues);
      if (oob) { if (bad_index) *bad_index = idx_signed ? s : (int64_t)u; return ORC_EINDEX; }
    }
  }
  memset(vout, 0, (size_t)(nidx * byte_width));
  if (out_valid) memset(out_valid, 0, (size_t)((nidx + 7) / 8));
  int64_t nulls = 0;
  for (int64_t i = 0; i < nidx; i++) {
    if (!bget_opt(ivalid, ioff + i)) { nulls++; continue; }
    /* :1147-1158 indices reinterpreted as unsigned of the same width */
    load_index(idx, idx_byte_width, idx_signed, i, &s, &u);
    if (u >= (uint64_t)nvalues) return ORC_EINDEX; /* Go would panic; never reached after a bounds check */
    if (!bget_opt(vvalid, voff + (int64_t)u)) { nulls++; continue; }
    memcpy(vout + i * byte_width, vin + u * byte_width, (size_t)byte_width);
    if (out_valid) bset(out_valid, i);
  }
  if (out_null_count) *out_null_count = nulls;
  return ORC_OK;
}

/*
 * GetTakeIndices (vector_selection.go:102-236), uint32 flavour: positions of
 * the selected filter slots; under EMIT_NULLS a filter-null slot emits a null
 * index (payload 0 — builder AppendNull; validity 0).
 */
int orc_filter_to_indices(const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                          uint32_t* out_idx, uint8_t* out_valid, int64_t* out_len, int64_t* out_null_count) {
  int64_t n_out = orc_filter_count(fdata, fvalid, foff, n, null_sel);
  if (out_valid) memset(out_valid, 0, (size_t)((n_out + 7) / 8));
  int64_t pos = 0, nulls = 0;
  for (int64_t i = 0; i < n; i++) {
    int fv = bget_opt(fvalid, foff + i), fd = bget(fdata, foff + i);
    if (fv && fd) { out_idx[pos] = (uint32_t)i; if (out_valid) bset(out_valid, pos); pos++; }
    else if (!fv && null_sel == ORC_EMIT_NULLS) { out_idx[pos] = 0; nulls++; pos++; }
  }
  *out_len = n_out;
  if (out_null_count) *out_null_count = nulls;
  return ORC_OK;
}

/*
 * booleanTakeImpl (vector_selection.go:990-1074): out data bit i = value bit at voff + idx[i]; a null
 * output (null index or null value) keeps data bit 0 and validity bit 0; bounds as checkIndexBounds.
 */
int orc_take_boolean(const uint8_t* data, const uint8_t* vvalid, int64_t voff, int64_t nvalues, int idx_byte_width, int idx_signed,
                     const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, int bounds_check, uint8_t* out_data,
                     uint8_t* out_valid, int64_t* out_null_count, int64_t* bad_index) {
  int64_t s = 0; uint64_t u = 0;
  if (bounds_check) {
    for (int64_t i = 0; i < nidx; i++) {
      if (!bget_opt(ivalid, ioff + i)) continue;
      load_index(idx, idx_byte_width, idx_signed, i, &s, &u);
      int oob = idx_signed ? (s < 0 || (uint64_t)s >= (uint64_t)nvalues) : (u >= (uint64_t)nvalues);
      if (oob) { if (bad_index) *bad_index = idx_signed ? s : (int64_t)u; return ORC_EINDEX; }
    }
  }
  if (nidx < 0) return ORC_EINVALID;
  memset(out_data, 0, (size_t)((nidx + 7) / 8));
  if (out_valid) memset(out_valid, 0, (size_t)((nidx + 7) / 8));
  int64_t nulls = 0;
  for (int64_t i = 0; i < nidx; i++) {
    if (!bget_opt(ivalid, ioff + i)) { nulls++; continue; }
    load_index(idx, idx_byte_width, idx_signed, i, &s, &u);
    if (!bget_opt(vvalid, voff + (int64_t)u)) { nulls++; continue; }
    if ((data[(voff + (int64_t)u) >> 3] >> ((voff + (int64_t)u) & 7)) & 1) bset(out_data, i);
    if (out_valid) bset(out_valid, i);
  }
  if (out_null_count) *out_null_count = nulls;
  return ORC_OK;
}
