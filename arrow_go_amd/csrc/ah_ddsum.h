// ah_ddsum.h — the Float64 accumulator of Sum (ah_sum.hip), the fused Compare→Filter→Sum (ah_fused.hip), the chunked ingest
// and the cross-rank combine (ah_comm.hip).
//
// Rule: the result is the sum of the addends over the EXTENDED reals, rounded once —
//   any NaN addend, or +inf and −inf together        → NaN
//   only +inf (−inf) among the non-finite addends    → +inf (−inf)
//   finite addends                                   → the exact sum rounded to nearest; ±inf only if THAT exceeds DBL_MAX
// This is what both reference orders return (arrow/math/float64.go:41-47 strict left-to-right; float64_avx2_amd64.s 32 strided
// partials) whenever neither overflows on the way; where a reference order overflows only in an intermediate sum its two paths
// disagree with each other, and this accumulator returns the order-free answer.
//
// How: two double-double accumulators per lane.  Rows with |x| < 2^960 go to (s, e) by Knuth's TwoSum: 2^64 of them cannot
// overflow.  Rows with |x| ≥ 2^960, ±inf and NaN (one integer compare on the high word: exponent field ≥ 0x7BF) go to
// (bs, be) SCALED by 2^-128 (exact: the scaled value stays normal), so finite rows cannot overflow there either; a non-finite
// row lands in bs through the plain addition inside TwoSum, whose IEEE rules (inf + x = inf, inf − inf = NaN, NaN sticky) are
// the rule above; be is garbage from then on and is never read once bs is not finite.  The common case pays two integer
// operations per row and one branch per eight rows over the unguarded TwoSum.
#pragma once
#include <math.h>
#include <stdint.h>

struct ah_ddx {
  double s, e;    // rows below 2^960
  double bs, be;  // rows from 2^960 up, ±inf, NaN — times 2^-128
};

#define AH_DDX_BIG_HI 0x7bf00000u   // high word (sign cleared) of 2^960
#define AH_DDX_DOWN 0x1p-128
#define AH_DDX_UP 0x1p128

#if defined(__HIPCC__)
#define AH_HD __host__ __device__ __forceinline__
#else
#define AH_HD static inline
#endif

AH_HD void ah_dd_add(double& s, double& e, double x) {
  double t = s + x;
  double bp = t - s;
  e += (s - (t - bp)) + (x - bp);
  s = t;
}
AH_HD void ah_dd_merge(double& s, double& e, double os, double oe) {
  double t = s + os;
  double bp = t - s;
  e += ((s - (t - bp)) + (os - bp)) + oe;
  s = t;
}
AH_HD unsigned ah_dd_hi_abs(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return (unsigned)(u >> 32) & 0x7fffffffu;
}
AH_HD void ah_ddx_init(ah_ddx& a) { a.s = a.e = a.bs = a.be = 0.0; }
// one row, any class
AH_HD void ah_ddx_add(ah_ddx& a, double x) {
  if (ah_dd_hi_abs(x) >= AH_DDX_BIG_HI) ah_dd_add(a.bs, a.be, x * AH_DDX_DOWN);
  else ah_dd_add(a.s, a.e, x);
}
AH_HD void ah_ddx_merge(ah_ddx& a, const ah_ddx& o) {
  ah_dd_merge(a.s, a.e, o.s, o.e);
  ah_dd_merge(a.bs, a.be, o.bs, o.be);
}
// rounded once
AH_HD double ah_ddx_result(const ah_ddx& a) {
  if (a.bs == 0.0 && a.be == 0.0) return a.s + a.e;   // no big row was seen (or they were all ±0 — impossible): the plain double-double
  if (!(fabs(a.bs) <= 1.79769313486231570815e308)) return a.bs;   // ±inf or NaN by the IEEE rules of the plain additions
  // normalise the big part: hs = round(bs + be), he the exact remainder
  double hs = a.bs + a.be;
  double bp = hs - a.bs;
  double he = (a.bs - (hs - bp)) + (a.be - bp);
  if (hs == 0.0 && he == 0.0) return a.s + a.e;       // the big rows cancelled exactly
  if (fabs(hs) < 0x1p850) {
    // un-scaling cannot overflow (|hs·2^128| < 2^978, |s| < 2^960 · rows): merge at full scale, nothing of (s, e) is lost
    double U = hs * AH_DDX_UP, V = he * AH_DDX_UP;
    double t = U + a.s;
    double q = t - U;
    double err = (U - (t - q)) + (a.s - q);
    return t + ((err + V) + a.e);
  }
  // the big part dominates (≥ 2^978 at full scale): (s, e) scaled down lose at most bits below 2^-1074, 2000 binades under it
  double s2 = a.s * AH_DDX_DOWN, e2 = a.e * AH_DDX_DOWN;
  double t = hs + s2;
  double q = t - hs;
  double err = (hs - (t - q)) + (s2 - q);
  double r = t + ((err + he) + e2);
  return r * AH_DDX_UP;   // exact, or ±inf when the rounded sum is beyond DBL_MAX
}
