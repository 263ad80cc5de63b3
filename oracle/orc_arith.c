/*
 * orc_arith.c — element-wise arithmetic restated (TEST INFRASTRUCTURE).
 *
 * Reference:
 *   unchecked ADD/SUB/MUL/ABS/NEGATE/SIGN, all slots computed (null payloads
 *   included): arrow/compute/internal/kernels/_lib/base_arithmetic.cc:52-273
 *   (the C++ the AVX2 asm is generated from) == the Go loop at
 *   kernels/base_arithmetic.go:110-134.
 *   checked integer ADD/SUB: kernels/base_arithmetic.go:249-278 through
 *   ScalarBinaryNotNull (kernels/helpers.go:284-380): null slots ← 0, and the
 *   reference's carry test, restated bit for bit (see checked_carry below).
 *   checked integer MUL: kernels/base_arithmetic.go:84-106 (mulWithOverflow)
 *   through ScalarBinary (every slot evaluated, null payloads included).
 *   checked float ops are the unchecked SIMD kernels
 *   (kernels/base_arithmetic_amd64.go:109-117).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

static inline int bit_get(const uint8_t* b, int64_t i) { return b == 0 ? 1 : (b[i >> 3] >> (i & 7)) & 1; }

#define FOR_INT_TYPES(X) \
  X(ORC_UINT8, uint8_t, uint8_t) X(ORC_INT8, int8_t, uint8_t) \
  X(ORC_UINT16, uint16_t, uint16_t) X(ORC_INT16, int16_t, uint16_t) \
  X(ORC_UINT32, uint32_t, uint32_t) X(ORC_INT32, int32_t, uint32_t) \
  X(ORC_UINT64, uint64_t, uint64_t) X(ORC_INT64, int64_t, uint64_t)

/* shape 0: l[i] op r[i]; 1: l[i] op r[0]; 2: l[0] op r[i] */
static int binary_impl(int type, int8_t op, int shape, const void* lv, const void* rv, void* ov, int64_t len) {
  int ls = shape == ORC_SHAPE_SA ? 0 : 1, rs = shape == ORC_SHAPE_AS ? 0 : 1;
  if (op == ORC_OP_ADD_CHECKED) op = ORC_OP_ADD; /* base_arithmetic.cc:54-57: "_CHECKED" aliases do not check */
  if (op == ORC_OP_SUB_CHECKED) op = ORC_OP_SUB;
  if (op == ORC_OP_MUL_CHECKED) op = ORC_OP_MUL;
  if (op != ORC_OP_ADD && op != ORC_OP_SUB && op != ORC_OP_MUL) return ORC_EINVALID;
  switch (type) {
#define X(ID, T, U)                                                              \
  case ID: {                                                                     \
    const T* l = (const T*)lv; const T* r = (const T*)rv; T* o = (T*)ov;         \
    for (int64_t i = 0; i < len; i++) {                                          \
      U a = (U)l[i * ls], b = (U)r[i * rs];                                      \
      /* two's-complement wraparound: the C++ does signed MUL in unsigned      \
         (base_arithmetic.cc:107-124); ADD/SUB wrap in the SIMD lanes */        \
      U v = op == ORC_OP_ADD ? (U)(a + b) : op == ORC_OP_SUB ? (U)(a - b) : (U)(a * b); \
      o[i] = (T)v;                                                               \
    }                                                                            \
    return ORC_OK;                                                               \
  }
    FOR_INT_TYPES(X)
#undef X
    case ORC_FLOAT32: {
      const float* l = (const float*)lv; const float* r = (const float*)rv; float* o = (float*)ov;
      for (int64_t i = 0; i < len; i++) {
        float a = l[i * ls], b = r[i * rs];
        o[i] = op == ORC_OP_ADD ? a + b : op == ORC_OP_SUB ? a - b : a * b;
      }
      return ORC_OK;
    }
    case ORC_FLOAT64: {
      const double* l = (const double*)lv; const double* r = (const double*)rv; double* o = (double*)ov;
      for (int64_t i = 0; i < len; i++) {
        double a = l[i * ls], b = r[i * rs];
        o[i] = op == ORC_OP_ADD ? a + b : op == ORC_OP_SUB ? a - b : a * b;
      }
      return ORC_OK;
    }
  }
  return ORC_EINVALID;
}

int orc_arithmetic_binary(int type, int8_t op, const void* l, const void* r, void* out, int64_t len) {
  return binary_impl(type, op, ORC_SHAPE_AA, l, r, out, len);
}
int orc_arithmetic_arr_scalar(int type, int8_t op, const void* l, const void* r, void* out, int64_t len) {
  return binary_impl(type, op, ORC_SHAPE_AS, l, r, out, len);
}
int orc_arithmetic_scalar_arr(int type, int8_t op, const void* l, const void* r, void* out, int64_t len) {
  return binary_impl(type, op, ORC_SHAPE_SA, l, r, out, len);
}

/* base_arithmetic.cc:137-208 (AbsoluteValue, Negate, Sign), same in/out type */
int orc_arithmetic_unary(int type, int8_t op, const void* inv, void* ov, int64_t len) {
  if (op != ORC_OP_ABS && op != ORC_OP_NEGATE && op != ORC_OP_SIGN) return ORC_EINVALID;
  switch (type) {
#define XU(ID, T)                                                                \
  case ID: {                                                                     \
    const T* in = (const T*)inv; T* o = (T*)ov;                                  \
    for (int64_t i = 0; i < len; i++) {                                          \
      T x = in[i];                                                               \
      o[i] = op == ORC_OP_ABS ? x : op == ORC_OP_NEGATE ? (T)(~x + 1) : (T)(x > 0 ? 1 : 0); \
    }                                                                            \
    return ORC_OK;                                                               \
  }
#define XS(ID, T, U)                                                             \
  case ID: {                                                                     \
    const T* in = (const T*)inv; T* o = (T*)ov;                                  \
    for (int64_t i = 0; i < len; i++) {                                          \
      T x = in[i];                                                               \
      if (op == ORC_OP_ABS) {                                                    \
        U m = (U)(x < 0 ? ~(U)0 : 0); /* mask = x >> (bits-1) */                 \
        o[i] = (T)(((U)x + m) ^ m);                                              \
      } else if (op == ORC_OP_NEGATE) {                                          \
        o[i] = (T)(0 - (U)x);                                                    \
      } else {                                                                   \
        o[i] = (T)(x > 0 ? 1 : (x ? -1 : 0));                                    \
      }                                                                          \
    }                                                                            \
    return ORC_OK;                                                               \
  }
    XU(ORC_UINT8, uint8_t) XU(ORC_UINT16, uint16_t) XU(ORC_UINT32, uint32_t) XU(ORC_UINT64, uint64_t)
    XS(ORC_INT8, int8_t, uint8_t) XS(ORC_INT16, int16_t, uint16_t)
    XS(ORC_INT32, int32_t, uint32_t) XS(ORC_INT64, int64_t, uint64_t)
#undef XU
#undef XS
    case ORC_FLOAT32: {
      const float* in = (const float*)inv; float* o = (float*)ov;
      for (int64_t i = 0; i < len; i++) {
        float x = in[i];
        if (op == ORC_OP_ABS) { uint32_t u; memcpy(&u, &x, 4); u &= 0x7fffffffu; memcpy(&o[i], &u, 4); }
        else if (op == ORC_OP_NEGATE) o[i] = -x;
        else o[i] = isnan(x) ? x : (x == 0 ? 0.0f : (signbit(x) ? -1.0f : 1.0f));
      }
      return ORC_OK;
    }
    case ORC_FLOAT64: {
      const double* in = (const double*)inv; double* o = (double*)ov;
      for (int64_t i = 0; i < len; i++) {
        double x = in[i];
        if (op == ORC_OP_ABS) { uint64_t u; memcpy(&u, &x, 8); u &= 0x7fffffffffffffffull; memcpy(&o[i], &u, 8); }
        else if (op == ORC_OP_NEGATE) o[i] = -x;
        else o[i] = isnan(x) ? x : (x == 0 ? 0.0 : (signbit(x) ? -1.0 : 1.0));
      }
      return ORC_OK;
    }
  }
  return ORC_EINVALID;
}

/*
 * Checked integer ops.
 *
 * ADD (base_arithmetic.go:249-263):  out = a + b;
 *     carry := (OutT(a&b) | (OutT(a|b) &^ out)) >> shiftBy ; overflow iff carry > 0
 * SUB (:264-278):                   out = a - b;
 *     carry := (OutT(^a&b) | (^OutT(a^b) & out)) >> shiftBy
 * with shiftBy = bits-1 for unsigned and bits-2 for signed, the shift being
 * ARITHMETIC for signed T.  For signed T, "carry > 0" therefore means: carry
 * vector bit (bits-1) clear AND bit (bits-2) set.  This is the reference's
 * test, not the textbook signed-overflow test — e.g. MinInt64 + MinInt64
 * (carry bit 63 set) is NOT reported, while MaxInt64 + MaxInt64 is.  Restated
 * as is: parity means matching the reference, quirks included.
 */
#define CHECKED_CASE(ID, T, U, IS_SIGNED)                                                      \
  case ID: {                                                                                   \
    const T* l = (const T*)lvp; const T* r = (const T*)rvp; T* o = (T*)ov;                     \
    const int bits = (int)sizeof(T) * 8;                                                       \
    T tmin = IS_SIGNED ? (T)((U)1 << (bits - 1)) : (T)0;                                       \
    T tmax = IS_SIGNED ? (T)(~((U)1 << (bits - 1))) : (T)~(U)0;                                \
    for (int64_t i = 0; i < len; i++) {                                                        \
      int valid = (ls ? bit_get(lvalid, loff + i) : scalar_valid) &&                           \
                  (rs ? bit_get(rvalid, roff + i) : scalar_valid);                             \
      T a = l[i * ls], b = r[i * rs];                                                          \
      if (op == ORC_OP_MUL_CHECKED) {                                                          \
        /* ScalarBinary: every slot, null payloads included (base_arithmetic.go:279-286) */   \
        int ovf = 0;                                                                           \
        if (a > 0) { if (b > 0) { if (a > (T)(tmax / b)) ovf = 1; }                            \
                     else { if (b < (T)(tmin / a)) ovf = 1; } }                                \
        else if (b > 0) { if (a < (T)(tmin / b)) ovf = 1; }                                    \
        else { if (a != 0 && b < (T)(tmax / a)) ovf = 1; }                                     \
        if (ovf) { status = ORC_EOVERFLOW; o[i] = 0; }                                         \
        else o[i] = (T)((U)a * (U)b);                                                          \
        continue;                                                                              \
      }                                                                                        \
      if (!valid) { o[i] = 0; continue; } /* helpers.go:303-306: def */                        \
      U ua = (U)a, ub = (U)b, out, c;                                                          \
      if (op == ORC_OP_ADD_CHECKED) { out = (U)(ua + ub); c = (U)((ua & ub) | ((ua | ub) & (U)~out)); } \
      else { out = (U)(ua - ub); c = (U)(((U)~ua & ub) | ((U) ~(ua ^ ub) & out)); }            \
      int top = (c >> (bits - 1)) & 1, next = bits >= 2 ? (int)((c >> (bits - 2)) & 1) : 0;    \
      int carry_pos = IS_SIGNED ? (!top && next) : top;                                        \
      if (carry_pos) status = ORC_EOVERFLOW;                                                   \
      o[i] = (T)out;                                                                           \
    }                                                                                          \
    return status;                                                                             \
  }

int orc_arithmetic_checked(int type, int8_t op, int shape,
                           const void* lvp, const uint8_t* lvalid, int64_t loff,
                           const void* rvp, const uint8_t* rvalid, int64_t roff,
                           int scalar_valid, void* ov, int64_t len) {
  int ls = shape == ORC_SHAPE_SA ? 0 : 1, rs = shape == ORC_SHAPE_AS ? 0 : 1;
  int status = ORC_OK;
  if (type == ORC_FLOAT32 || type == ORC_FLOAT64) return binary_impl(type, op, shape, lvp, rvp, ov, len);
  if (op != ORC_OP_ADD_CHECKED && op != ORC_OP_SUB_CHECKED && op != ORC_OP_MUL_CHECKED) return ORC_EINVALID;
  if (op != ORC_OP_MUL_CHECKED && shape != ORC_SHAPE_AA && !scalar_valid) {
    /* helpers.go:312-314,341-343: null scalar → output left as allocated (zero) */
    memset(ov, 0, (size_t)len * (type == ORC_UINT8 || type == ORC_INT8 ? 1 : type == ORC_UINT16 || type == ORC_INT16 ? 2
                                 : type == ORC_UINT32 || type == ORC_INT32 ? 4 : 8));
    return ORC_OK;
  }
  switch (type) {
    CHECKED_CASE(ORC_UINT8, uint8_t, uint8_t, 0) CHECKED_CASE(ORC_INT8, int8_t, uint8_t, 1)
    CHECKED_CASE(ORC_UINT16, uint16_t, uint16_t, 0) CHECKED_CASE(ORC_INT16, int16_t, uint16_t, 1)
    CHECKED_CASE(ORC_UINT32, uint32_t, uint32_t, 0) CHECKED_CASE(ORC_INT32, int32_t, uint32_t, 1)
    CHECKED_CASE(ORC_UINT64, uint64_t, uint64_t, 0) CHECKED_CASE(ORC_INT64, int64_t, uint64_t, 1)
  }
  return ORC_EINVALID;
}

/* ---- the exact rest of the arithmetic registry (no SIMD leaf in the reference: pure Go) ----------------
 * divide / divide_unchecked   base_arithmetic.go:154-160,287-294 (ints: zero divisor in a valid slot → errDivByZero
 *                             under both names; Go's truncated quotient; MinInt / −1 wraps), :386-396 (floats:
 *                             unchecked a / b, checked refuses b == 0) — ScalarBinaryNotNull (helpers.go:284-380):
 *                             out = 0 in null slots, the closure runs only on valid ones
 * abs / negate                :295-340: ScalarUnary over the WHOLE value buffer (helpers.go:56-90) → MinInt under a
 *                             null is an overflow too; unsigned abs = copy; floats :398-411
 * bit_wise_and / or / xor     scalar_arithmetic.go:170-245 (bitmap op on the value bytes: every slot)
 * bit_wise_not                :253-268 ScalarUnaryNotNull
 * shift_left / shift_right    :293-378: count outside [0, bits − 2] (signed) / [0, bits − 1] (unsigned) → lhs, and
 *                             errShift for the checked names; ScalarBinaryNotNull
 * floor / ceil / trunc        rounding.go:180-187,748-775 (ScalarUnary: every slot)
 * sqrt_unchecked / sqrt       base_arithmetic.go:412-426
 * msg (≥ 128 bytes, may be NULL) receives the reference's error text. */
#include <math.h>
#include <stdio.h>
enum { X_OK = 0, X_OVERFLOW = 1, X_DIVZERO = 2, X_SHIFT = 3, X_NEGSQRT = 4, X_NEGPOWER = 5 };

static int ext_is_unary(int op) { return op == 71 || op == 25 || op == 26 || op == 6 || op == 27 || (op >= 72 && op <= 74); }
static int ext_every_slot(int op) { return op == 7 || op == 68 || op == 69 || op == 70 || op == 25 || op == 26 || op == 6 || (op >= 72 && op <= 74); }

#define EXT_INT_BODY(T, U, SIGNED)                                                                                 \
  {                                                                                                                \
    const T* l = (const T*)lvp; const T* r = (const T*)rvp; T* o = (T*)ov;                                         \
    const int bits = (int)sizeof(T) * 8;                                                                           \
    for (int64_t i = 0; i < len && err == X_OK; i++) {                                                             \
      const T a = shape == ORC_SHAPE_SA ? l[0] : l[i];                                                             \
      const T b = unary ? (T)0 : (shape == ORC_SHAPE_AS ? r[0] : r[i]);                                            \
      int valid = 1;                                                                                               \
      if (!every) {                                                                                                \
        if (shape != ORC_SHAPE_SA && lvalid && !((lvalid[(loff + i) >> 3] >> ((loff + i) & 7)) & 1)) valid = 0;    \
        if (!unary && shape != ORC_SHAPE_AS && rvalid && !((rvalid[(roff + i) >> 3] >> ((roff + i) & 7)) & 1)) valid = 0; \
      }                                                                                                            \
      if (!valid) { o[i] = 0; continue; }                                                                          \
      switch (op) {                                                                                                \
        case 3: case 24:                                                                                           \
          if (b == 0) { err = X_DIVZERO; o[i] = 0; break; }                                                        \
          if (SIGNED && b == (T)-1) { o[i] = (T)((U)0 - (U)a); break; }                                            \
          o[i] = (T)(a / b); break;                                                                                \
        case 64: case 65: case 66: case 67: {                                                                      \
          int bad = SIGNED ? ((long long)b < 0 || (long long)b >= bits - 1) : ((unsigned long long)b >= (unsigned long long)bits); \
          if (bad) { if (op == 65 || op == 67) err = X_SHIFT; o[i] = a; break; }                                   \
          if (op == 64 || op == 65) o[i] = (T)((U)a << (int)b);                                                    \
          else o[i] = SIGNED ? (T)((long long)a >> (int)b) : (T)((U)a >> (int)b);                                  \
          break;                                                                                                   \
        }                                                                                                          \
        case 71: o[i] = (T)~a; break;                                                                              \
        case 68: o[i] = (T)(a & b); break;                                                                         \
        case 69: o[i] = (T)(a | b); break;                                                                         \
        case 70: o[i] = (T)(a ^ b); break;                                                                         \
        case 7: { /* power_unchecked, base_arithmetic.go:226-248: uint64 right-to-left, narrowed */              \
          if (SIGNED && (long long)b < 0) { err = X_NEGPOWER; break; }                                             \
          unsigned long long base = (unsigned long long)a, e = (unsigned long long)b, pw = 1;                      \
          while (e != 0) { if (e & 1) pw *= base; base *= base; e >>= 1; }                                         \
          o[i] = (T)pw; break; }                                                                                   \
        case 28: { /* power, :342-373: left-to-right with mulWithOverflow (:84-108) */                             \
          if (SIGNED && (long long)b < 0) { err = X_NEGPOWER; break; }                                             \
          if (b == 0) { o[i] = 1; break; }                                                                         \
          unsigned long long ue = (unsigned long long)b, mask = 1ull << (63 - __builtin_clzll(ue));                \
          T pw = 1, t; int of = 0;                                                                                 \
          while (mask != 0) {                                                                                      \
            if (__builtin_mul_overflow(pw, pw, &t)) { of = 1; t = 0; }                                             \
            pw = t;                                                                                                \
            if (ue & mask) { if (__builtin_mul_overflow(pw, a, &t)) { of = 1; t = 0; } pw = t; }                   \
            mask >>= 1;                                                                                            \
          }                                                                                                        \
          if (of) { err = X_OVERFLOW; break; }                                                                     \
          o[i] = pw; break; }                                                                                      \
        case 25: case 26:                                                                                          \
          if (!SIGNED) { if (op == 26) return ORC_EINVALID; o[i] = a; break; }                                     \
          if ((U)a == (U)1 << (bits - 1)) { err = X_OVERFLOW; break; }                                             \
          o[i] = op == 25 ? (T)((long long)a < 0 ? (T)((U)0 - (U)a) : a) : (T)((U)0 - (U)a);                       \
          break;                                                                                                   \
        default: return ORC_EINVALID;                                                                              \
      }                                                                                                            \
    }                                                                                                              \
  }

#define EXT_FLOAT_BODY(T, SQRTF, FABSF)                                                                            \
  {                                                                                                                \
    const T* l = (const T*)lvp; const T* r = (const T*)rvp; T* o = (T*)ov;                                         \
    for (int64_t i = 0; i < len && err == X_OK; i++) {                                                             \
      const T a = shape == ORC_SHAPE_SA ? l[0] : l[i];                                                             \
      const T b = unary ? (T)0 : (shape == ORC_SHAPE_AS ? r[0] : r[i]);                                            \
      int valid = 1;                                                                                               \
      if (!every) {                                                                                                \
        if (shape != ORC_SHAPE_SA && lvalid && !((lvalid[(loff + i) >> 3] >> ((loff + i) & 7)) & 1)) valid = 0;    \
        if (!unary && shape != ORC_SHAPE_AS && rvalid && !((rvalid[(roff + i) >> 3] >> ((roff + i) & 7)) & 1)) valid = 0; \
      }                                                                                                            \
      if (!valid) { o[i] = 0; continue; }                                                                          \
      switch (op) {                                                                                                \
        case 3: o[i] = a / b; break;                                                                               \
        case 7: case 28: o[i] = (T)pow((double)a, (double)b); break;   /* :443-446 */                              \
        case 24: if (b == 0) { err = X_DIVZERO; o[i] = 0; } else o[i] = a / b; break;                              \
        case 25: o[i] = FABSF(a); break;                                                                           \
        case 26: o[i] = -a; break;                                                                                 \
        case 6: o[i] = SQRTF(a); break;                                                                            \
        case 27: if (a < 0) { err = X_NEGSQRT; o[i] = (T)NAN; } else o[i] = SQRTF(a); break;                       \
        case 72: o[i] = (T)floor((double)a); break;   /* rounding.go:183 */                                          \
        case 73: o[i] = (T)ceil((double)a); break;    /* :185 */                                                     \
        case 74: o[i] = (T)trunc((double)a); break;   /* :187 */                                                     \
        default: return ORC_EINVALID;                                                                              \
      }                                                                                                            \
    }                                                                                                              \
  }

int orc_arithmetic_ext(int type, int op, int shape, const void* lvp, const uint8_t* lvalid, int64_t loff, const void* rvp,
                       const uint8_t* rvalid, int64_t roff, int scalar_valid, void* ov, int64_t len, char* msg) {
  const int unary = ext_is_unary(op);
  const int every = ext_every_slot(op) || (op == 28 && (type == ORC_FLOAT32 || type == ORC_FLOAT64));  /* float power: ScalarBinary under both names */
  int err = X_OK;
  if (unary) shape = ORC_SHAPE_AS;
  if (!every && !unary && shape != ORC_SHAPE_AA && !scalar_valid) {
    const int w = type == ORC_UINT8 || type == ORC_INT8 ? 1 : type == ORC_UINT16 || type == ORC_INT16 ? 2
                : type == ORC_UINT32 || type == ORC_INT32 || type == ORC_FLOAT32 ? 4 : 8;
    memset(ov, 0, (size_t)len * w);  /* helpers.go:312-314,341-343 */
    return ORC_OK;
  }
  switch (type) {
    case ORC_UINT8: EXT_INT_BODY(uint8_t, uint8_t, 0) break;
    case ORC_INT8: EXT_INT_BODY(int8_t, uint8_t, 1) break;
    case ORC_UINT16: EXT_INT_BODY(uint16_t, uint16_t, 0) break;
    case ORC_INT16: EXT_INT_BODY(int16_t, uint16_t, 1) break;
    case ORC_UINT32: EXT_INT_BODY(uint32_t, uint32_t, 0) break;
    case ORC_INT32: EXT_INT_BODY(int32_t, uint32_t, 1) break;
    case ORC_UINT64: EXT_INT_BODY(uint64_t, uint64_t, 0) break;
    case ORC_INT64: EXT_INT_BODY(int64_t, uint64_t, 1) break;
    case ORC_FLOAT32: EXT_FLOAT_BODY(float, sqrtf, fabsf) break;
    case ORC_FLOAT64: EXT_FLOAT_BODY(double, sqrt, fabs) break;
    default: return ORC_EINVALID;
  }
  static const char* text[] = {"", "overflow", "divide by zero", "shift amount must be >= 0 and less than precision of type",
                               "square root of negative number", "integers to negative integer powers are not allowed"};
  if (err != X_OK) {
    if (msg) snprintf(msg, 128, "%s", text[err]);
    return err == X_OVERFLOW ? ORC_EOVERFLOW : ORC_EINVALID;
  }
  return ORC_OK;
}

/* ---- round / round_to_multiple (kernels/rounding.go) --------------------------------------------------------------
 * round[T].call :329-370, roundToMultiple[T].call :570-598, getFloatRoundImpl :180-221; ScalarUnaryNotNull.
 * pow10 is math.Pow10(|ndigits|) (InitRoundState :72-91) — orc_pow10 restates Go's table-driven math.Pow10
 * (src/math/pow10.go: pow10tab[n % 32] · pow10postab32[n / 32]), which is what the reference multiplies by. */
double orc_pow10(int n) {
  static const double tab[32] = {1e00, 1e01, 1e02, 1e03, 1e04, 1e05, 1e06, 1e07, 1e08, 1e09, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22, 1e23, 1e24, 1e25, 1e26, 1e27, 1e28, 1e29, 1e30, 1e31};
  static const double pos32[10] = {1e00, 1e32, 1e64, 1e96, 1e128, 1e160, 1e192, 1e224, 1e256, 1e288};
  if (n < 0) return 0;           /* the callers pass |ndigits| */
  if (n > 308) return INFINITY;
  return pos32[n / 32] * tab[n % 32];
}

static double round_mode_impl(double d, int mode) {
  switch (mode) {
    case 0: case 4: return floor(d);
    case 1: case 5: return ceil(d);
    case 2: case 6: return trunc(d);
    case 3: case 7: return signbit(d) ? floor(d) : ceil(d);
    case 8: return nearbyint(d);   /* math.RoundToEven; the default FE_TONEAREST mode */
    default: return floor(d * 0.5) + ceil(d * 0.5);
  }
}

#define ROUND_BODY(T)                                                                                            \
  {                                                                                                              \
    const T* in = (const T*)vp; T* o = (T*)ov;                                                                   \
    const T scale = multiple ? *(const T*)multiple : (T)orc_pow10((int)(ndigits < 0 ? -ndigits : ndigits));      \
    for (int64_t i = 0; i < n; i++) {                                                                            \
      if (valid && !((valid[(off + i) >> 3] >> ((off + i) & 7)) & 1)) { o[i] = 0; continue; }                    \
      const T arg = in[i];                                                                                       \
      o[i] = arg;                                                                                                \
      if (isinf((double)arg) || isnan((double)arg)) continue;                                                    \
      T rv = (multiple || ndigits < 0) ? arg / scale : arg * scale;                                              \
      const T frac = rv - (T)floor((double)rv);                                                                  \
      if (frac == 0) continue;                                                                                   \
      if (mode >= 4 && frac != (T)0.5) rv = (T)round((double)rv);                                                \
      else rv = (T)round_mode_impl((double)rv, mode);                                                            \
      if (multiple) rv *= scale; else if (ndigits > 0) rv /= scale; else rv *= scale;                            \
      if (isinf((double)rv) || isnan((double)rv)) return ORC_EOVERFLOW;                                          \
      o[i] = rv;                                                                                                 \
    }                                                                                                            \
    return ORC_OK;                                                                                               \
  }

int orc_round(int type, const void* vp, const uint8_t* valid, int64_t off, int64_t n, int64_t ndigits, int mode, const void* multiple, void* ov) {
  if (type == ORC_FLOAT32) ROUND_BODY(float)
  if (type == ORC_FLOAT64) ROUND_BODY(double)
  return ORC_EINVALID;
}
