/*
 * orc_sum.c — arrow/math Sum restated (TEST INFRASTRUCTURE, see oracle.h).
 *
 * Reference: arrow/math/float64.go:34-47 (Float64Funcs.Sum → sum_float64_go),
 * arrow/math/_lib/float64.c:20-26 (the C truth the AVX2 asm is generated
 * from), arrow/math/_lib/float64_avx2.s (the vectorised summation order),
 * arrow/math/int64.go:34-47, uint64.go:34-47.
 *
 * Validity bitmaps are ignored by the reference (float64.go:41-46 iterates the
 * raw value slots), so these take only the value buffer.
 * Compile WITHOUT -ffast-math: the orders below are the point.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>

/* noasm path: strict left-to-right `acc += v` (arrow/math/float64.go:41-47). */
void orc_sum_float64_seq(const double* buf, size_t len, double* res) {
  volatile double acc = 0.0; /* volatile: forbid reassociation/vectorisation */
  for (size_t i = 0; i < len; i++) acc = acc + buf[i];
  *res = acc;
}

/*
 * AVX2 path order (arrow/math/_lib/float64_avx2.s): for len >= 32, eight ymm
 * accumulators × 4 lanes = 32 strided partial sums p[j] += buf[32k + j]
 * (.LBB0_9 / .LBB0_12), combined as
 *   t = ((y0+y4)+(y2+y6)) + ((y1+y5)+(y3+y7))        (.LBB0_13, lane-wise)
 *   acc = (t[0]+t[2]) + (t[1]+t[3])                  (vextractf128, vhaddpd)
 * then a sequential scalar tail over the last len % 32 elements (.LBB0_4).
 * For len < 32 the whole sum is the sequential loop starting from 0.0.
 */
void orc_sum_float64_avx2order(const double* buf, size_t len, double* res) {
  double acc = 0.0;
  size_t body = len & ~(size_t)31;
  size_t i = 0;
  if (len > 31 && body != 0) {
    double p[32];
    for (int j = 0; j < 32; j++) p[j] = 0.0;
    for (size_t k = 0; k < body; k += 32)
      for (int j = 0; j < 32; j++) p[j] = p[j] + buf[k + j];
    double t[4];
    for (int q = 0; q < 4; q++) {
      /* y_m lane q == p[4*m + q] */
      double y1 = p[4 + q] + p[20 + q];
      double y3 = p[12 + q] + p[28 + q];
      double y0 = p[0 + q] + p[16 + q];
      double y2 = p[8 + q] + p[24 + q];
      y0 = y0 + y2;
      y1 = y1 + y3;
      t[q] = y0 + y1;
    }
    double u0 = t[0] + t[2], u1 = t[1] + t[3];
    acc = u0 + u1;
    i = body;
  }
  for (; i < len; i++) acc = acc + buf[i];
  *res = acc;
}

/*
 * Correctly rounded sum (Shewchuk's exact expansion, the algorithm behind
 * Python's math.fsum).  Not a reference path: it is the yardstick the float
 * parity rule is stated against ("within 1 ULP of the exact sum"), because the
 * reference's own two paths above disagree with each other on general data.
 * Finite inputs only.
 */
void orc_sum_float64_exact(const double* buf, size_t len, double* res) {
  size_t cap = 64, n = 0;
  double* part = (double*)malloc(cap * sizeof(double));
  for (size_t k = 0; k < len; k++) {
    double x = buf[k];
    size_t i = 0;
    for (size_t j = 0; j < n; j++) {
      double y = part[j];
      if (fabs(x) < fabs(y)) { double t = x; x = y; y = t; }
      volatile double hi = x + y;
      volatile double yr = hi - x;
      double lo = y - yr;
      if (lo != 0.0) part[i++] = lo;
      x = hi;
    }
    n = i;
    if (x != 0.0) {
      if (n >= cap) { cap *= 2; part = (double*)realloc(part, cap * sizeof(double)); }
      part[n++] = x;
    }
  }
  double hi = 0.0;
  if (n > 0) {
    size_t j = n;
    hi = part[--j];
    double lo = 0.0;
    while (j > 0) {
      double x = hi;
      double y = part[--j];
      volatile double h = x + y;
      volatile double yr = h - x;
      hi = h;
      lo = y - yr;
      if (lo != 0.0) break;
    }
    /* round-half-even correction across the remaining partials */
    if (j > 0 && ((lo < 0.0 && part[j - 1] < 0.0) || (lo > 0.0 && part[j - 1] > 0.0))) {
      double y = lo * 2.0;
      volatile double x = hi + y;
      volatile double yr = x - hi;
      if (y == yr) hi = x;
    }
  }
  free(part);
  *res = hi;
}

/* wrapping Σ mod 2^64 (arrow/math/_lib/int64.c:21-27; Go int64 add wraps). */
void orc_sum_int64(const int64_t* buf, size_t len, int64_t* res) {
  uint64_t acc = 0;
  for (size_t i = 0; i < len; i++) acc += (uint64_t)buf[i];
  *res = (int64_t)acc;
}

void orc_sum_uint64(const uint64_t* buf, size_t len, uint64_t* res) {
  uint64_t acc = 0;
  for (size_t i = 0; i < len; i++) acc += buf[i];
  *res = acc;
}

/*
 * The sum over the EXTENDED reals, rounded once — the yardstick for inputs that hold ±inf, NaN, or finite values whose
 * running sum overflows in one summation order and not in another.  Both reference orders above return this value whenever
 * neither overflows on the way (any NaN addend or both infinities → NaN; only +inf / −inf → that infinity; finite addends →
 * the exact sum rounded to nearest even, ±inf only if THAT is beyond DBL_MAX); where they overflow in an intermediate sum
 * they disagree with each other (tests/test_oracle_vs_reference.py shows a vector), so the order-free value is the rule.
 *
 * Method (deliberately unlike the double-double of the HIP kernels): a fixed-point superaccumulator — every finite double is
 * an integer multiple of 2^-1074 below 2^1024, so the sum of < 2^63 of them fits 2098 + 63 bits.  72 limbs of 32 bits kept
 * carry-save in int64 (normalised every 2^20 rows), sign-magnitude at the end, round to nearest even.
 */
#define XR_LIMBS 72
static void xr_normalise(int64_t* l) {
  int64_t carry = 0;
  for (int i = 0; i < XR_LIMBS; i++) {
    int64_t v = l[i] + carry;
    carry = v >> 32;                 /* arithmetic shift: floor division */
    l[i] = v & 0xffffffffLL;
  }
  l[XR_LIMBS - 1] += carry << 32;    /* the top limb keeps the sign */
}

void orc_sum_float64_xreal(const double* buf, size_t len, double* res) {
  int64_t l[XR_LIMBS];
  for (int i = 0; i < XR_LIMBS; i++) l[i] = 0;
  int has_nan = 0, has_pinf = 0, has_ninf = 0;
  size_t since = 0;
  for (size_t k = 0; k < len; k++) {
    uint64_t u;
    __builtin_memcpy(&u, &buf[k], 8);
    const int neg = (int)(u >> 63);
    const int ex = (int)((u >> 52) & 0x7ff);
    uint64_t m = u & 0xfffffffffffffULL;
    if (ex == 0x7ff) {
      if (m) has_nan = 1; else if (neg) has_ninf = 1; else has_pinf = 1;
      continue;
    }
    int p;                           /* value = m · 2^(p − 1074) */
    if (ex == 0) p = 0; else { m |= 1ULL << 52; p = ex - 1; }
    if (m == 0) continue;
    const int w = p >> 5, sh = p & 31;
    /* m << sh spans up to 85 bits: three 32-bit pieces */
    const uint64_t lo = m << sh;                         /* low 64 bits */
    const uint64_t hi = sh ? (m >> (64 - sh)) : 0;       /* bits 64.. */
    const int64_t a0 = (int64_t)(lo & 0xffffffffULL), a1 = (int64_t)(lo >> 32), a2 = (int64_t)hi;
    if (neg) { l[w] -= a0; l[w + 1] -= a1; l[w + 2] -= a2; }
    else { l[w] += a0; l[w + 1] += a1; l[w + 2] += a2; }
    if (++since == (1u << 20)) { xr_normalise(l); since = 0; }
  }
  if (has_nan || (has_pinf && has_ninf)) { *res = NAN; return; }
  if (has_pinf) { *res = INFINITY; return; }
  if (has_ninf) { *res = -INFINITY; return; }
  xr_normalise(l);
  int negative = l[XR_LIMBS - 1] < 0;
  if (negative) {                    /* two's complement → magnitude */
    int64_t carry = 1;
    for (int i = 0; i < XR_LIMBS - 1; i++) {
      int64_t v = ((~l[i]) & 0xffffffffLL) + carry;
      carry = v >> 32;
      l[i] = v & 0xffffffffLL;
    }
    l[XR_LIMBS - 1] = ~l[XR_LIMBS - 1] + carry;
  }
  /* top set bit */
  int top = -1;
  for (int i = XR_LIMBS - 1; i >= 0 && top < 0; i--) {
    uint64_t v = (uint64_t)l[i];
    if (v) top = i * 32 + (63 - __builtin_clzll(v));
  }
  if (top < 0) { *res = 0.0; return; }   /* +0, like 0.0 + (−0.0) */
#define XR_BIT(b) ((b) < 0 ? 0 : (int)(((uint64_t)l[(b) >> 5] >> ((b) & 31)) & 1))
  uint64_t mant = 0;
  double r;
  if (top < 52) {                    /* subnormal or the smallest normals: every bit fits, exact */
    for (int b = top; b >= 0; b--) mant = (mant << 1) | (uint64_t)XR_BIT(b);
    __builtin_memcpy(&r, &mant, 8);  /* m · 2^-1074 has exactly this bit pattern */
  } else {
    for (int b = top; b > top - 53; b--) mant = (mant << 1) | (uint64_t)XR_BIT(b);
    const int guard = XR_BIT(top - 53);
    int sticky = 0;
    for (int b = top - 54; b >= 0 && !sticky; b--) sticky = XR_BIT(b);
    int e = top - 52 + 1;            /* biased exponent of mant · 2^(top−52−1074) */
    if (guard && (sticky || (mant & 1))) { mant++; if (mant >> 53) { mant >>= 1; e++; } }
    if (e >= 0x7ff) r = INFINITY;
    else { uint64_t bits = ((uint64_t)e << 52) | (mant & 0xfffffffffffffULL); __builtin_memcpy(&r, &bits, 8); }
  }
#undef XR_BIT
  *res = negative ? -r : r;
}
