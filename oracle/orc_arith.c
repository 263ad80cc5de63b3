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
