/*
 * orc_minmax.c — utils.GetMinMax* restated (TEST INFRASTRUCTURE).
 * Reference: internal/utils/min_max.go:30-148 (int8MinMax … uint64MinMax): min starts at MaxOf, max at
 * MinOf, one pass with `if min > v` / `if max < v`; an empty slice therefore returns (MaxOf, MinOf).
 * Pinned against the reference's AVX2 machine code (int64_max_min_avx2 …, _lib/min_max.c:23-126).
 */
#include "oracle.h"
#include <string.h>

#define MM(T, TMAX, TMIN)                                                       \
  { const T* p = (const T*)values; T lo = TMAX, hi = TMIN;                      \
    for (int64_t i = 0; i < n; i++) { if (lo > p[i]) lo = p[i]; if (hi < p[i]) hi = p[i]; } \
    memcpy(out_min, &lo, sizeof(T)); memcpy(out_max, &hi, sizeof(T)); return ORC_OK; }

int orc_min_max(int type, const void* values, int64_t n, void* out_min, void* out_max) {
  switch (type) {
    case ORC_UINT8: MM(uint8_t, UINT8_MAX, 0)
    case ORC_INT8: MM(int8_t, INT8_MAX, INT8_MIN)
    case ORC_UINT16: MM(uint16_t, UINT16_MAX, 0)
    case ORC_INT16: MM(int16_t, INT16_MAX, INT16_MIN)
    case ORC_UINT32: MM(uint32_t, UINT32_MAX, 0)
    case ORC_INT32: MM(int32_t, INT32_MAX, INT32_MIN)
    case ORC_UINT64: MM(uint64_t, UINT64_MAX, 0)
    case ORC_INT64: MM(int64_t, INT64_MAX, INT64_MIN)
  }
  return ORC_EINVALID;
}
