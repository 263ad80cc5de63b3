/*
 * orc_fused.c — Compare → Filter → Sum, restated as the UNFUSED reference chain
 * collapsed per element (TEST INFRASTRUCTURE, see oracle.h).
 *
 * The chain in the reference:
 *   mask = compute.CallFunction("greater", x, scalar)   — mask validity = x validity
 *          (executor.go:237-349 propagateNulls; scalar_comparisons.go:199-218)
 *   y    = compute.Filter(x, mask, DropNulls)            — a mask slot that is
 *          null (x null) or false is dropped (vector_selection.go:267-395)
 *   s    = math.Int64.Sum(y) / math.Float64.Sum(y)        — sums every slot of y
 *          (arrow/math/float64.go:41-47); all slots of y are valid because the
 *          nulls were dropped with the mask.
 * Hence s = Σ x[i] over { i : valid[i] ∧ x[i] OP t }, count = |that set|.
 * LESS / LESS_EQUAL are the reference's operand swap (scalar_compare.go:73-99);
 * callers express them as scalar_arr GT/GE — here only arr OP scalar is needed.
 */
#include "oracle.h"
#include <stdlib.h>

static inline int bget_opt(const uint8_t* b, int64_t i) { return b == 0 ? 1 : (b[i >> 3] >> (i & 7)) & 1; }

#define PRED(a, t) (cmpop == ORC_CMP_EQ ? (a) == (t) : cmpop == ORC_CMP_NE ? (a) != (t) \
                    : cmpop == ORC_CMP_GT ? (a) > (t) : (a) >= (t))

int orc_cmp_filter_sum_i64(int cmpop, const int64_t* x, const uint8_t* valid, int64_t off, int64_t n,
                           int64_t threshold, int64_t* out_sum, int64_t* out_count) {
  uint64_t acc = 0; int64_t cnt = 0;
  for (int64_t i = 0; i < n; i++)
    if (bget_opt(valid, off + i) && PRED(x[i], threshold)) { acc += (uint64_t)x[i]; cnt++; }
  *out_sum = (int64_t)acc; *out_count = cnt;
  return ORC_OK;
}

int orc_cmp_filter_sum_f64(int cmpop, const double* x, const uint8_t* valid, int64_t off, int64_t n,
                           double threshold, double* out_sum_seq, double* out_sum_exact, int64_t* out_count) {
  double* kept = (double*)malloc((size_t)(n > 0 ? n : 1) * sizeof(double));
  int64_t cnt = 0;
  for (int64_t i = 0; i < n; i++)
    if (bget_opt(valid, off + i) && PRED(x[i], threshold)) kept[cnt++] = x[i];
  if (out_sum_seq) orc_sum_float64_seq(kept, (size_t)cnt, out_sum_seq);
  if (out_sum_exact) orc_sum_float64_exact(kept, (size_t)cnt, out_sum_exact);
  *out_count = cnt;
  free(kept);
  return ORC_OK;
}
