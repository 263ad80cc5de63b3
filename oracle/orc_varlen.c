/*
 * orc_varlen.c — Take and Filter of binary / string columns restated (TEST INFRASTRUCTURE).
 *
 * Reference: arrow/compute/internal/kernels/vector_selection.go
 *   VarBinaryImpl :1925-1992 — a selected VALID slot appends the output offset, then the value's bytes;
 *     every other emitted slot (null value, null index, filter-null under EmitNulls) appends only the
 *     offset (a zero-length null); a final offset closes the column; output offsets start at 0
 *   takeExec / filterExec :1460-1598, 1821-1923 — which slots are emitted and which are valid: the same
 *     rules as the primitive kernels (orc_select.c); checkIndexBounds (helpers.go:929-981)
 *   checkBinaryTakeOffset :1241-1247 — "binary output offset overflow" when the data no longer fits the
 *     offset type
 * offsets: (nvalues + 1) entries of the offset width starting at element `voff` of the offsets buffer
 * (the array's Offset applies to offsets and validity alike); data: the values buffer.
 */
#include "oracle.h"
#include <string.h>

static inline int bget_opt(const uint8_t* b, int64_t i) { return b == 0 ? 1 : (b[i >> 3] >> (i & 7)) & 1; }
static inline void bset(uint8_t* b, int64_t i) { b[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static int64_t off_at(int w, const void* offsets, int64_t i) { return w == 4 ? ((const int32_t*)offsets)[i] : ((const int64_t*)offsets)[i]; }
static void off_put(int w, void* offsets, int64_t i, int64_t v) { if (w == 4) ((int32_t*)offsets)[i] = (int32_t)v; else ((int64_t*)offsets)[i] = v; }

static int append(int w, const void* offsets, const uint8_t* data, int64_t voff, int64_t row, int is_valid, int64_t pos, void* out_offsets,
                  uint8_t* out_data, int64_t* cur) {
  off_put(w, out_offsets, pos, *cur);
  if (!is_valid) return ORC_OK;
  int64_t lo = off_at(w, offsets, voff + row), hi = off_at(w, offsets, voff + row + 1);
  if (w == 4 && *cur + (hi - lo) > INT32_MAX) return ORC_EINVALID;  /* binary output offset overflow */
  if (out_data) memcpy(out_data + *cur, data + lo, (size_t)(hi - lo));
  *cur += hi - lo;
  return ORC_OK;
}

/* out_data may be NULL (size query): *out_total_bytes is always set */
int orc_take_binary(int offset_width, const void* offsets, const uint8_t* data, const uint8_t* vvalid, int64_t voff, int64_t nvalues,
                    int idx_byte_width, int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx,
                    int bounds_check, void* out_offsets, uint8_t* out_data, uint8_t* out_valid, int64_t* out_null_count,
                    int64_t* out_total_bytes, int64_t* bad_index) {
  if (offset_width != 4 && offset_width != 8) return ORC_EINVALID;
  int64_t nulls = 0, cur = 0;
  if (out_valid) memset(out_valid, 0, (size_t)((nidx + 7) / 8));
  for (int pass = bounds_check ? 0 : 1; pass < 2; pass++) {
    for (int64_t i = 0; i < nidx; i++) {
      int iv = bget_opt(ivalid, ioff + i);
      int64_t s = 0; uint64_t u = 0;
      switch (idx_byte_width) {
        case 1: s = ((const int8_t*)idx)[i]; u = ((const uint8_t*)idx)[i]; break;
        case 2: s = ((const int16_t*)idx)[i]; u = ((const uint16_t*)idx)[i]; break;
        case 4: s = ((const int32_t*)idx)[i]; u = ((const uint32_t*)idx)[i]; break;
        default: s = ((const int64_t*)idx)[i]; u = ((const uint64_t*)idx)[i];
      }
      if (pass == 0) {
        if (!iv) continue;
        int oob = idx_signed ? (s < 0 || (uint64_t)s >= (uint64_t)nvalues) : (u >= (uint64_t)nvalues);
        if (oob) { if (bad_index) *bad_index = idx_signed ? s : (int64_t)u; return ORC_EINDEX; }
        continue;
      }
      int ok = iv && bget_opt(vvalid, voff + (int64_t)u);
      int rc = append(offset_width, offsets, data, voff, (int64_t)u, ok, i, out_offsets, out_data, &cur);
      if (rc != ORC_OK) return rc;
      if (ok) { if (out_valid) bset(out_valid, i); } else nulls++;
    }
  }
  off_put(offset_width, out_offsets, nidx, cur);
  if (out_null_count) *out_null_count = nulls;
  if (out_total_bytes) *out_total_bytes = cur;
  return ORC_OK;
}

int orc_filter_binary(int offset_width, const void* offsets, const uint8_t* data, const uint8_t* vvalid, int64_t voff,
                      const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel, void* out_offsets,
                      uint8_t* out_data, uint8_t* out_valid, int64_t* out_len, int64_t* out_null_count, int64_t* out_total_bytes) {
  if (offset_width != 4 && offset_width != 8) return ORC_EINVALID;
  int64_t pos = 0, nulls = 0, cur = 0;
  int64_t n_out = orc_filter_count(fdata, fvalid, foff, n, null_sel);
  if (out_valid) memset(out_valid, 0, (size_t)((n_out + 7) / 8));
  for (int64_t i = 0; i < n; i++) {
    int fv = bget_opt(fvalid, foff + i);
    int fd = (fdata[(foff + i) >> 3] >> ((foff + i) & 7)) & 1;
    int emit, ok;
    if (fv && fd) { emit = 1; ok = bget_opt(vvalid, voff + i); }
    else if (!fv && null_sel == ORC_EMIT_NULLS) { emit = 1; ok = 0; }
    else emit = 0, ok = 0;
    if (!emit) continue;
    int rc = append(offset_width, offsets, data, voff, i, ok, pos, out_offsets, out_data, &cur);
    if (rc != ORC_OK) return rc;
    if (ok) { if (out_valid) bset(out_valid, pos); } else nulls++;
    pos++;
  }
  off_put(offset_width, out_offsets, pos, cur);
  if (out_len) *out_len = pos;
  if (out_null_count) *out_null_count = nulls;
  if (out_total_bytes) *out_total_bytes = cur;
  return ORC_OK;
}
