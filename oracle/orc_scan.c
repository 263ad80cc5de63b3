/*
 * orc_scan.c — cumulative_sum / cumulative_sum_checked restated (TEST INFRASTRUCTURE).
 *
 * Reference: arrow/compute/internal/kernels/vector_cumulative.go
 *   cumulativeSumNoNulls :228-241, cumulativeSumNoNullsChecked :243-262,
 *   cumulativeSumWithNulls :264-290 (null → ClearBit + continue; `encounteredNull` makes every
 *   later row null unless SkipNulls), checked adders :147-160 (textbook range test),
 *   prepareCumulativeOutput :211-226 (fresh zeroed data, validity pre-filled with 0xFF).
 * The running sum is accumulated in T itself, in row order — for floats this IS the reference's
 * rounding order.
 */
#include "oracle.h"
#include <string.h>

static inline int bget_opt(const uint8_t* b, int64_t i) { return b == 0 ? 1 : (b[i >> 3] >> (i & 7)) & 1; }

#define SCAN_INT(ID, T, U, TMIN, TMAX, IS_SIGNED)                                            \
  case ID: {                                                                                  \
    const T* in = (const T*)values; T* out = (T*)out_values;                                  \
    T cur = 0; if (start) memcpy(&cur, start, sizeof(T));                                     \
    for (int64_t i = 0; i < n; i++) {                                                         \
      int ok = bget_opt(valid, off + i);                                                      \
      if (!ok || seen_null) {                                                                 \
        if (out_valid) out_valid[i >> 3] &= (uint8_t)~(1u << (i & 7));                        \
        nulls++;                                                                              \
        if (!ok && !skip_nulls) seen_null = 1;                                                \
        continue;                                                                             \
      }                                                                                       \
      T r = in[i];                                                                            \
      if (checked) {                                                                          \
        if (IS_SIGNED) { if ((r > 0 && cur > (T)(TMAX - r)) || (r < 0 && cur < (T)(TMIN - r))) return ORC_EOVERFLOW; } \
        else { if (cur > (T)(TMAX - r)) return ORC_EOVERFLOW; }                               \
      }                                                                                       \
      cur = (T)((U)cur + (U)r);                                                               \
      out[i] = cur;                                                                           \
    }                                                                                         \
    break;                                                                                    \
  }
#define SCAN_FLT(ID, T)                                                                       \
  case ID: {                                                                                  \
    const T* in = (const T*)values; T* out = (T*)out_values;                                  \
    volatile T cur = 0; if (start) { T s0; memcpy(&s0, start, sizeof(T)); cur = s0; }         \
    for (int64_t i = 0; i < n; i++) {                                                         \
      int ok = bget_opt(valid, off + i);                                                      \
      if (!ok || seen_null) {                                                                 \
        if (out_valid) out_valid[i >> 3] &= (uint8_t)~(1u << (i & 7));                        \
        nulls++;                                                                              \
        if (!ok && !skip_nulls) seen_null = 1;                                                \
        continue;                                                                             \
      }                                                                                       \
      cur = cur + in[i];                                                                      \
      out[i] = cur;                                                                           \
    }                                                                                         \
    break;                                                                                    \
  }

int orc_cumulative_sum(int type, const void* values, const uint8_t* valid, int64_t off, int64_t n,
                       const void* start, int skip_nulls, int checked, void* out_values, uint8_t* out_valid,
                       int64_t* out_null_count) {
  int64_t nulls = 0;
  int seen_null = 0;
  int w = type == ORC_UINT8 || type == ORC_INT8 ? 1 : type == ORC_UINT16 || type == ORC_INT16 ? 2
        : type == ORC_UINT32 || type == ORC_INT32 || type == ORC_FLOAT32 ? 4 : 8;
  memset(out_values, 0, (size_t)(n * w));
  if (out_valid) memset(out_valid, 0xFF, (size_t)((n + 7) / 8));
  switch (type) {
    SCAN_INT(ORC_UINT8, uint8_t, uint8_t, 0, UINT8_MAX, 0) SCAN_INT(ORC_INT8, int8_t, uint8_t, INT8_MIN, INT8_MAX, 1)
    SCAN_INT(ORC_UINT16, uint16_t, uint16_t, 0, UINT16_MAX, 0) SCAN_INT(ORC_INT16, int16_t, uint16_t, INT16_MIN, INT16_MAX, 1)
    SCAN_INT(ORC_UINT32, uint32_t, uint32_t, 0, UINT32_MAX, 0) SCAN_INT(ORC_INT32, int32_t, uint32_t, INT32_MIN, INT32_MAX, 1)
    SCAN_INT(ORC_UINT64, uint64_t, uint64_t, 0, UINT64_MAX, 0) SCAN_INT(ORC_INT64, int64_t, uint64_t, INT64_MIN, INT64_MAX, 1)
    SCAN_FLT(ORC_FLOAT32, float) SCAN_FLT(ORC_FLOAT64, double)
    default: return ORC_EINVALID;
  }
  /* padding bits past n in the last validity byte: the reference fills whole bytes with 0xFF */
  if (out_null_count) *out_null_count = nulls;
  return ORC_OK;
}
