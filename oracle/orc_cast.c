/*
 * orc_cast.c — numeric cast restated (TEST INFRASTRUCTURE).
 *
 * Reference (arrow/compute/internal/kernels):
 *   castNumberToNumberUnsafe → castNumericUnsafe (cast_numeric.go:28-131; AVX2 leaf
 *     cast_type_numeric_avx2, _lib/cast_numeric.cc:22-27: out[i] = static_cast<O>(in[i]) for EVERY
 *     slot, valid or not)
 *   CastIntToInt :37-46 → intsCanFit / intsInRange (helpers.go:496-652): only valid slots, bounds in
 *     the INPUT type, "integer value %d not in range: %d to %d" for the first offender
 *   CastIntegerToFloating :62-71 → checkIntToFloatTrunc (numeric_cast.go:698-729): |v| ≤ 2^24 / 2^53
 *   CastFloatingToInteger :53-60 → checkFloatTrunc (:613-660): valid slot with OutT(in) != in →
 *     "float value %f was truncated converting to %s"
 *   boolToNum (numeric_cast.go:555-569)
 *
 * float → int for values the target cannot hold is undefined in C++ ([conv.fpint]) and
 * implementation-specific in Go; with AllowFloatTruncate the reference just stores what the CPU
 * produced.  The rule restated here (and implemented on the GPU) is: truncate toward zero into 64
 * bits, saturating, NaN → 0, then keep the low bits — which is what x86 produces for every
 * |v| < 2^31 (any target) and for every value a 64-bit target can hold.  In safe mode the decision
 * "was truncated" is taken on the input (integral and inside the target's range), which is
 * equivalent to the reference's round-trip test.
 */
#include "oracle.h"
#include <math.h>
#include <stdio.h>
#include <string.h>

static inline int bget_opt(const uint8_t* b, int64_t i) { return b == 0 ? 1 : (b[i >> 3] >> (i & 7)) & 1; }

static int type_width(int t) {
  switch (t) {
    case ORC_UINT8: case ORC_INT8: return 1;
    case ORC_UINT16: case ORC_INT16: return 2;
    case ORC_UINT32: case ORC_INT32: case ORC_FLOAT32: return 4;
    case ORC_UINT64: case ORC_INT64: case ORC_FLOAT64: return 8;
  }
  return 0;
}
static int is_float(int t) { return t == ORC_FLOAT32 || t == ORC_FLOAT64; }
static int is_signed(int t) { return t == ORC_INT8 || t == ORC_INT16 || t == ORC_INT32 || t == ORC_INT64; }
static const char* type_name(int t) {
  switch (t) {
    case ORC_UINT8: return "uint8"; case ORC_INT8: return "int8"; case ORC_UINT16: return "uint16"; case ORC_INT16: return "int16";
    case ORC_UINT32: return "uint32"; case ORC_INT32: return "int32"; case ORC_UINT64: return "uint64"; case ORC_INT64: return "int64";
    case ORC_FLOAT32: return "float32"; case ORC_FLOAT64: return "float64";
  }
  return "?";
}

/* integer payloads travel as (is_negative, magnitude-as-two's-complement uint64) */
static uint64_t load_int(int t, const void* p, int64_t i) {
  switch (t) {
    case ORC_UINT8: return ((const uint8_t*)p)[i];
    case ORC_INT8: return (uint64_t)(int64_t)((const int8_t*)p)[i];
    case ORC_UINT16: return ((const uint16_t*)p)[i];
    case ORC_INT16: return (uint64_t)(int64_t)((const int16_t*)p)[i];
    case ORC_UINT32: return ((const uint32_t*)p)[i];
    case ORC_INT32: return (uint64_t)(int64_t)((const int32_t*)p)[i];
    default: return ((const uint64_t*)p)[i];
  }
}
static void store_int(int t, void* p, int64_t i, uint64_t v) {
  switch (type_width(t)) {
    case 1: ((uint8_t*)p)[i] = (uint8_t)v; break;
    case 2: ((uint16_t*)p)[i] = (uint16_t)v; break;
    case 4: ((uint32_t*)p)[i] = (uint32_t)v; break;
    default: ((uint64_t*)p)[i] = v;
  }
}

/* truncate toward zero into 64 bits: saturating, NaN → 0; values in [2^63, 2^64) keep their
 * unsigned bit pattern when the target is uint64 */
static uint64_t float_to_bits64(double v, int out_unsigned64) {
  if (v != v) return 0;
  if (out_unsigned64 && v >= 9223372036854775808.0) {
    if (v >= 18446744073709551616.0) return UINT64_MAX;
    return (uint64_t)(int64_t)(v - 9223372036854775808.0) + 0x8000000000000000ull;
  }
  if (v >= 9223372036854775808.0) return (uint64_t)INT64_MAX;
  if (v <= -9223372036854775808.0) return (uint64_t)INT64_MIN;
  return (uint64_t)(int64_t)v;
}

static void int_bounds(int in_type, int out_type, int* need, int64_t* lo_s, uint64_t* lo_u, uint64_t* hi_u, int64_t* hi_s) {
  /* getSafeMinMaxSigned / Unsigned (helpers.go:496-543): bounds in the input type */
  int wi = type_width(in_type) * 8, wo = type_width(out_type) * 8;
  int si = is_signed(in_type), so = is_signed(out_type);
  uint64_t max_in = si ? ((1ull << (wi - 1)) - 1) : (wi == 64 ? UINT64_MAX : ((1ull << wi) - 1));
  uint64_t max_out = so ? ((1ull << (wo - 1)) - 1) : (wo == 64 ? UINT64_MAX : ((1ull << wo) - 1));
  int64_t min_in = si ? (wi == 64 ? INT64_MIN : -(int64_t)(1ull << (wi - 1))) : 0;
  int64_t min_out = so ? (wo == 64 ? INT64_MIN : -(int64_t)(1ull << (wo - 1))) : 0;
  *hi_u = max_in < max_out ? max_in : max_out;
  *hi_s = (int64_t)*hi_u;
  *lo_s = min_in > min_out ? min_in : min_out;
  *lo_u = 0;
  *need = !(min_in >= *lo_s && max_in <= *hi_u);
}

int orc_cast_numeric(int in_type, int out_type, const void* in, const uint8_t* valid, int64_t off, int64_t n,
                     int allow_int_overflow, int allow_float_truncate, void* out, int64_t* bad_index, char* msg /*256*/) {
  int wi = type_width(in_type), wo = type_width(out_type);
  if (!wi || !wo) return ORC_EINVALID;
  if (bad_index) *bad_index = -1;
  if (msg) msg[0] = 0;
  const int fi = is_float(in_type), fo = is_float(out_type);
  /* ---- checks that precede the conversion (int → int, int → float) ---- */
  if (!fi) {
    int need = 0; int64_t lo_s = 0, hi_s = 0; uint64_t lo_u = 0, hi_u = 0;
    if (!fo && !allow_int_overflow) int_bounds(in_type, out_type, &need, &lo_s, &lo_u, &hi_u, &hi_s);
    if (fo && !allow_float_truncate && wi >= 4 && !(wi == 4 && out_type == ORC_FLOAT64)) {
      /* checkIntToFloatTrunc */
      uint64_t limit = out_type == ORC_FLOAT32 ? (1ull << 24) : (1ull << 53);
      need = 1; hi_u = limit; hi_s = (int64_t)limit; lo_s = is_signed(in_type) ? -(int64_t)limit : 0;
    }
    if (need) {
      for (int64_t i = 0; i < n; i++) {
        if (!bget_opt(valid, off + i)) continue;
        uint64_t v = load_int(in_type, in, i);
        int bad = is_signed(in_type) ? ((int64_t)v < lo_s || (int64_t)v > hi_s) : (v > hi_u);
        if (bad) {
          if (bad_index) *bad_index = i;
          if (msg) {
            if (is_signed(in_type)) snprintf(msg, 256, "integer value %lld not in range: %lld to %lld", (long long)(int64_t)v, (long long)lo_s, (long long)hi_s);
            else snprintf(msg, 256, "integer value %llu not in range: %llu to %llu", (unsigned long long)v, 0ull, (unsigned long long)hi_u);
          }
          return ORC_EINVALID;
        }
      }
    }
  }
  /* ---- the conversion, every slot ---- */
  for (int64_t i = 0; i < n; i++) {
    if (!fi && !fo) {
      store_int(out_type, out, i, load_int(in_type, in, i));  /* sign-extend to 64 bits, keep the low bits */
    } else if (!fi && fo) {
      uint64_t v = load_int(in_type, in, i);
      if (out_type == ORC_FLOAT32) ((float*)out)[i] = is_signed(in_type) ? (float)(int64_t)v : (float)v;
      else ((double*)out)[i] = is_signed(in_type) ? (double)(int64_t)v : (double)v;
    } else if (fi && fo) {
      if (in_type == ORC_FLOAT32) {
        float v = ((const float*)in)[i];
        if (out_type == ORC_FLOAT32) ((float*)out)[i] = v; else ((double*)out)[i] = (double)v;
      } else {
        double v = ((const double*)in)[i];
        if (out_type == ORC_FLOAT32) ((float*)out)[i] = (float)v; else ((double*)out)[i] = v;
      }
    } else {
      double v = in_type == ORC_FLOAT32 ? (double)((const float*)in)[i] : ((const double*)in)[i];
      store_int(out_type, out, i, float_to_bits64(v, out_type == ORC_UINT64));
    }
  }
  /* ---- float → int: the truncation check follows the conversion (CastFloatingToInteger :53-60) ---- */
  if (fi && !fo && !allow_float_truncate) {
    int wo8 = wo * 8;
    double lo = is_signed(out_type) ? -ldexp(1.0, wo8 - 1) : 0.0;
    double hi_excl = is_signed(out_type) ? ldexp(1.0, wo8 - 1) : ldexp(1.0, wo8);
    for (int64_t i = 0; i < n; i++) {
      if (!bget_opt(valid, off + i)) continue;
      double v = in_type == ORC_FLOAT32 ? (double)((const float*)in)[i] : ((const double*)in)[i];
      if (!(v == trunc(v) && v >= lo && v < hi_excl)) {
        if (bad_index) *bad_index = i;
        if (msg) {
          if (v != v) snprintf(msg, 256, "float value NaN was truncated converting to %s", type_name(out_type));
          else if (isinf(v)) snprintf(msg, 256, "float value %sInf was truncated converting to %s", v > 0 ? "+" : "-", type_name(out_type));
          else snprintf(msg, 256, "float value %f was truncated converting to %s", v, type_name(out_type));
        }
        return ORC_EINVALID;
      }
    }
  }
  return ORC_OK;
}

int orc_cast_bool_to_numeric(int out_type, const uint8_t* bits, int64_t off, int64_t n, void* out) {
  if (!type_width(out_type)) return ORC_EINVALID;
  for (int64_t i = 0; i < n; i++) {
    int b = (bits[(off + i) >> 3] >> ((off + i) & 7)) & 1;
    if (out_type == ORC_FLOAT32) ((float*)out)[i] = b ? 1.0f : 0.0f;
    else if (out_type == ORC_FLOAT64) ((double*)out)[i] = b ? 1.0 : 0.0;
    else store_int(out_type, out, i, (uint64_t)b);
  }
  return ORC_OK;
}

/* ---- ShiftTime (kernels/cast_temporal.go:35-104) --------------------------------------------------
 * The reference's loops restated for the four (InT, OutT) pairs: factor == 1 converts (:41-45); multiply checks the
 * int64 bounds MaxInt64/factor, MinInt64/factor on VALID slots before writing OutT(v)*OutT(factor) (:47-74); divide
 * writes OutT(v / InT(factor)) and then checks InT(out)*InT(factor) != v on VALID slots (:75-102).  The reference
 * returns at the first failing row in index order; *bad_value is that row's input. */
#define SHIFT_LOOP(InT, OutT, UIn, UOut)                                                            \
  do {                                                                                              \
    const InT* src = (const InT*)in;                                                                \
    OutT* dst = (OutT*)out;                                                                         \
    if (factor == 1) {                                                                              \
      for (int64_t i = 0; i < n; i++) dst[i] = (OutT)src[i];                                        \
      return ORC_OK;                                                                                \
    }                                                                                               \
    if (op == 0) {                                                                                  \
      const int64_t max_val = INT64_MAX / factor, min_val = INT64_MIN / factor;                     \
      for (int64_t i = 0; i < n; i++) {                                                             \
        const InT v = src[i];                                                                       \
        const int is_valid = !valid || ((valid[(off + i) >> 3] >> ((off + i) & 7)) & 1);            \
        if (check && is_valid && ((int64_t)v < min_val || (int64_t)v > max_val)) {                  \
          if (bad_value) *bad_value = (int64_t)v;                                                   \
          return ORC_EINVALID;                                                                      \
        }                                                                                           \
        dst[i] = (OutT)((UOut)(OutT)v * (UOut)(OutT)factor);                                        \
      }                                                                                             \
      return ORC_OK;                                                                                \
    }                                                                                               \
    {                                                                                               \
      const InT f = (InT)factor;                                                                    \
      if (f == 0) return ORC_EINVALID;                                                              \
      for (int64_t i = 0; i < n; i++) {                                                             \
        const InT v = src[i];                                                                       \
        const int is_valid = !valid || ((valid[(off + i) >> 3] >> ((off + i) & 7)) & 1);            \
        dst[i] = (OutT)(v / f);                                                                     \
        if (check && is_valid && (InT)((UIn)(InT)dst[i] * (UIn)f) != v) {                           \
          if (bad_value) *bad_value = (int64_t)v;                                                   \
          return ORC_EINVALID;                                                                      \
        }                                                                                           \
      }                                                                                             \
      return ORC_OK;                                                                                \
    }                                                                                               \
  } while (0)

int orc_shift_time(int in_bits, int out_bits, int op, int64_t factor, int check, const void* in, const uint8_t* valid, int64_t off,
                   int64_t n, void* out, int64_t* bad_value) {
  if (factor < 1 || (op != 0 && op != 1)) return ORC_EINVALID;
  if (bad_value) *bad_value = 0;
  if (in_bits == 32 && out_bits == 32) SHIFT_LOOP(int32_t, int32_t, uint32_t, uint32_t);
  else if (in_bits == 32 && out_bits == 64) SHIFT_LOOP(int32_t, int64_t, uint32_t, uint64_t);
  else if (in_bits == 64 && out_bits == 32) SHIFT_LOOP(int64_t, int32_t, uint64_t, uint32_t);
  else if (in_bits == 64 && out_bits == 64) SHIFT_LOOP(int64_t, int64_t, uint64_t, uint64_t);
  return ORC_EINVALID;
}
