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
