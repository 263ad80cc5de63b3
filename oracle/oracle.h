/*
 * oracle.h — CPU restatement of arrow-go's compute/math hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load liboracle.so, and only as the checker / reported baseline.
 * libarrowhip.so (arrow_go_amd/csrc) never links or calls into this file.
 *
 * Every function cites the reference file:line (relative to the arrow-go
 * tree) whose algorithm it restates.  The restatement is deliberately the
 * simplest per-element form that obeys the same payload rules — not a
 * transliteration of the block state machines.
 *
 * Pinning: tests/test_golden.py checks these functions against every known-answer
 * vector the reference's own tests hold for the path, and
 * tests/test_oracle_vs_reference.py checks them bit for bit against the
 * reference's own AVX2 machine code (oracle/_ref/libref_avx2.so, assembled in
 * place by oracle/Makefile) on random inputs.  Not pinned by any reference test:
 * float Sum on non-integer data, payload bytes under nulls, group-by / fused
 * (no reference implementation) — see DESIGN.md §5.
 */
#ifndef ARROWHIP_ORACLE_H
#define ARROWHIP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* arrow.Type ids — arrow/datatype.go:36-72 == _lib/types.h:20-34 */
enum {
  ORC_UINT8 = 2, ORC_INT8 = 3, ORC_UINT16 = 4, ORC_INT16 = 5, ORC_UINT32 = 6,
  ORC_INT32 = 7, ORC_UINT64 = 8, ORC_INT64 = 9, ORC_FLOAT32 = 11, ORC_FLOAT64 = 12
};
/* ArithmeticOp — kernels/base_arithmetic.go:37-82 == _lib/base_arithmetic.cc:31-74 */
enum {
  ORC_OP_ADD = 0, ORC_OP_SUB = 1, ORC_OP_MUL = 2, ORC_OP_ABS = 4, ORC_OP_NEGATE = 5,
  ORC_OP_SIGN = 20, ORC_OP_ADD_CHECKED = 21, ORC_OP_SUB_CHECKED = 22, ORC_OP_MUL_CHECKED = 23
};
/* cmpop — _lib/scalar_comparison.cc:172-178 */
enum { ORC_CMP_EQ = 0, ORC_CMP_NE = 1, ORC_CMP_GT = 2, ORC_CMP_GE = 3 };
enum { ORC_SHAPE_AA = 0, ORC_SHAPE_AS = 1, ORC_SHAPE_SA = 2 };
/* bitOp — arrow/bitutil/bitmaps.go:494-521 */
enum { ORC_BIT_AND = 0, ORC_BIT_OR = 1, ORC_BIT_XOR = 2, ORC_BIT_AND_NOT = 3, ORC_BIT_XNOR = 4 };
/* NullSelectionBehavior — kernels/vector_selection.go:34-39 */
enum { ORC_DROP_NULLS = 0, ORC_EMIT_NULLS = 1 };
/* NullMatchingBehavior — kernels/scalar_set_lookup.go:31-38 */
enum { ORC_NULL_MATCH = 0, ORC_NULL_SKIP = 1, ORC_NULL_EMIT_NULL = 2, ORC_NULL_INCONCLUSIVE = 3 };
/* status classes shared with include/arrowhip.h */
enum { ORC_OK = 0, ORC_EINVALID = 1, ORC_EINDEX = 2, ORC_EOVERFLOW = 3 };

/* ---- arrow/math Sum ------------------------------------------------- */
void orc_sum_float64_seq(const double* buf, size_t len, double* res);
void orc_sum_float64_avx2order(const double* buf, size_t len, double* res);
void orc_sum_float64_exact(const double* buf, size_t len, double* res);
void orc_sum_float64_xreal(const double* buf, size_t len, double* res);   /* extended reals: ±inf, NaN, intermediate overflow */
void orc_sum_int64(const int64_t* buf, size_t len, int64_t* res);
void orc_sum_uint64(const uint64_t* buf, size_t len, uint64_t* res);

/* ---- element-wise arithmetic --------------------------------------- */
int orc_arithmetic_binary(int type, int8_t op, const void* l, const void* r, void* out, int64_t len);
int orc_arithmetic_arr_scalar(int type, int8_t op, const void* l, const void* r, void* out, int64_t len);
int orc_arithmetic_scalar_arr(int type, int8_t op, const void* l, const void* r, void* out, int64_t len);
int orc_arithmetic_unary(int type, int8_t op, const void* in, void* out, int64_t len);
/* shape: AA/AS/SA; validity may be NULL (= all valid); a null scalar is passed
 * as scalar_valid = 0.  Returns ORC_EOVERFLOW if any valid slot overflowed. */
int orc_arithmetic_checked(int type, int8_t op, int shape,
                           const void* l, const uint8_t* lv, int64_t loff,
                           const void* r, const uint8_t* rv, int64_t roff,
                           int scalar_valid, void* out, int64_t len);

/* divide / abs / negate (checked) / bit-wise / shifts / sqrt: ops numbered as in include/arrowhip.h
 * (DIV 3, SQRT 6, DIV_CHECKED 24, ABS_CHECKED 25, NEGATE_CHECKED 26, SQRT_CHECKED 27, SHIFT_LEFT 64 … BIT_NOT 71).
 * ORC_EOVERFLOW / ORC_EINVALID with the reference's text in msg (>= 128 bytes, may be NULL). */
int orc_arithmetic_ext(int type, int op, int shape, const void* l, const uint8_t* lvalid, int64_t loff, const void* r,
                       const uint8_t* rvalid, int64_t roff, int scalar_valid, void* out, int64_t len, char* msg);

/* round (multiple == NULL) / round_to_multiple on float32 / float64; mode = RoundMode 0..9 (rounding.go:40-59) */
int orc_round(int type, const void* values, const uint8_t* valid, int64_t off, int64_t n, int64_t ndigits, int mode, const void* multiple, void* out);
double orc_pow10(int n); /* Go's math.Pow10 for 0 <= n <= 308 */

/* ---- comparisons → packed bitmap ------------------------------------ */
int orc_comparison(int cmpop, int shape, int type, const void* l, const void* r,
                   uint8_t* out_bits, int64_t length, int out_bit_offset);

/* ---- bitmaps --------------------------------------------------------- */
int64_t orc_count_set_bits(const uint8_t* bits, int64_t off, int64_t nbits);
void orc_bitmap_op(int op, const uint8_t* l, int64_t loff, const uint8_t* r, int64_t roff,
                   uint8_t* out, int64_t ooff, int64_t nbits);
void orc_copy_bitmap(const uint8_t* src, int64_t soff, int64_t nbits, uint8_t* dst, int64_t doff, int invert);
void orc_set_bits_to(uint8_t* bits, int64_t off, int64_t nbits, int value);
/* Kleene ops: op 0 = and_kleene, 1 = or_kleene, 2 = and_not_kleene */
void orc_kleene(int op, const uint8_t* lvalid, const uint8_t* ldata, int64_t loff,
                const uint8_t* rvalid, const uint8_t* rdata, int64_t roff,
                uint8_t* ovalid, uint8_t* odata, int64_t ooff, int64_t nbits);

/* ---- selection ------------------------------------------------------- */
int64_t orc_filter_count(const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel);
int orc_filter_primitive(int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                         const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                         void* out_values, uint8_t* out_valid, int64_t* out_len, int64_t* out_null_count);
int orc_take_primitive(int byte_width, const void* values, const uint8_t* vvalid, int64_t voff, int64_t nvalues,
                       int idx_byte_width, int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff,
                       int64_t nidx, int bounds_check, void* out_values, uint8_t* out_valid,
                       int64_t* out_null_count, int64_t* bad_index);
int orc_take_boolean(const uint8_t* data, const uint8_t* vvalid, int64_t voff, int64_t nvalues, int idx_byte_width, int idx_signed,
                     const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, int bounds_check, uint8_t* out_data,
                     uint8_t* out_valid, int64_t* out_null_count, int64_t* bad_index);
int orc_filter_to_indices(const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                          uint32_t* out_idx, uint8_t* out_valid, int64_t* out_len, int64_t* out_null_count);

/* ---- var-length (binary / string) Take and Filter: VarBinaryImpl, kernels/vector_selection.go:1925-1992 ---- */
int orc_take_binary(int offset_width, const void* offsets, const uint8_t* data, const uint8_t* vvalid, int64_t voff, int64_t nvalues,
                    int idx_byte_width, int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx,
                    int bounds_check, void* out_offsets, uint8_t* out_data, uint8_t* out_valid, int64_t* out_null_count,
                    int64_t* out_total_bytes, int64_t* bad_index);
int orc_filter_binary(int offset_width, const void* offsets, const uint8_t* data, const uint8_t* vvalid, int64_t voff,
                      const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel, void* out_offsets,
                      uint8_t* out_data, uint8_t* out_valid, int64_t* out_len, int64_t* out_null_count, int64_t* out_total_bytes);

/* ---- hashing ---------------------------------------------------------- */
uint64_t orc_hash_int(uint64_t v, uint64_t alg);
/* unique / dictionary_encode over 8-byte keys (raw bit patterns).
 * encode_nulls=1 (unique, NullEncodingEncode): null gets a dictionary slot at
 * the position it was first seen; =0 (NullEncodingMask): null → id 0, out
 * validity cleared.  out_ids / out_ids_valid may be NULL (unique). */
int orc_hash_u64_encode(const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls,
                        int32_t* out_ids, uint8_t* out_ids_valid, uint64_t* out_dict,
                        int64_t* out_ndict, int32_t* out_null_id);
/* the same over Binary / String (offset_width 4) and LargeBinary / LargeString (8) values; the
 * dictionary is take(values, out_first_rows[0..ndict)) (a null entry points at the first null row) */
int orc_hash_binary_encode(int offset_width, const void* offsets, const uint8_t* data, const uint8_t* valid, int64_t off, int64_t n,
                           int encode_nulls, int32_t* out_ids, uint8_t* out_ids_valid, int64_t* out_first_rows,
                           int64_t* out_ndict, int32_t* out_null_id);
/* group-by (new functionality, oracle = sequential row-order accumulation) */
int orc_hash_sum_f64(const uint64_t* keys, const uint8_t* kvalid, int64_t koff,
                     const double* vals, const uint8_t* vvalid, int64_t voff, int64_t n,
                     uint64_t* out_keys, double* out_sums, int64_t* out_counts, int64_t* out_first_rows,
                     int64_t* out_ngroups, int32_t* out_null_group);
int orc_hash_sum_i64(const uint64_t* keys, const uint8_t* kvalid, int64_t koff,
                     const int64_t* vals, const uint8_t* vvalid, int64_t voff, int64_t n,
                     uint64_t* out_keys, int64_t* out_sums, int64_t* out_counts, int64_t* out_first_rows,
                     int64_t* out_ngroups, int32_t* out_null_group);

/* ---- cumulative_sum (kernels/vector_cumulative.go:228-360), sequential restatement ---- */
int orc_cumulative_sum(int type, const void* values, const uint8_t* valid, int64_t off, int64_t n,
                       const void* start, int skip_nulls, int checked, void* out_values, uint8_t* out_valid,
                       int64_t* out_null_count);

/* ---- numeric cast (kernels/cast_numeric.go, numeric_cast.go:37-71,613-729, helpers.go:496-652) ----
 * msg: 256-byte buffer receiving the reference's error text on ORC_EINVALID; bad_index: the offending row */
int orc_cast_numeric(int in_type, int out_type, const void* in, const uint8_t* valid, int64_t off, int64_t n,
                     int allow_int_overflow, int allow_float_truncate, void* out, int64_t* bad_index, char* msg);
int orc_cast_bool_to_numeric(int out_type, const uint8_t* bits, int64_t off, int64_t n, void* out);
/* ShiftTime (kernels/cast_temporal.go:35-104): op 0 multiply, 1 divide; ORC_EINVALID + *bad_value = the first failing row's input */
int orc_shift_time(int in_bits, int out_bits, int op, int64_t factor, int check, const void* in, const uint8_t* valid, int64_t off,
                   int64_t n, void* out, int64_t* bad_value);

/* ---- is_in (kernels/scalar_set_lookup.go:192-244,374-413): keys are the raw bits of the fixed-width
 * value; out_data / out_valid bits [out_off, out_off + n) are written, every other bit is preserved */
int orc_is_in(int byte_width, const void* values, const uint8_t* valid, int64_t off, int64_t n,
              const void* set_values, const uint8_t* set_valid, int64_t set_off, int64_t set_n, int null_behavior,
              uint8_t* out_data, uint8_t* out_valid, int64_t out_off);

/* ---- sort_indices of one numeric array (kernels/vector_sort.go:388-481, vector_sort_internal.go:252-273):
 * stable; [rest, NaNs, nulls] (nulls at end) or [nulls, NaNs, rest] (at start); out: n uint64 row numbers */
int orc_sort_indices(int type, const void* values, const uint8_t* valid, int64_t off, int64_t n, int descending,
                     int nulls_at_start, uint64_t* out_indices);

int orc_sort_indices_multi(int nkeys, const int* types, const void* const* values, const uint8_t* const* valids, const int64_t* offs,
                           int64_t n, const int* descending, const int* nulls_at_start, uint64_t* out_indices);

/* ---- utils.GetMinMax* (internal/utils/min_max.go:30-215): empty → (MaxOf, MinOf) ---- */
int orc_min_max(int type, const void* values, int64_t n, void* out_min, void* out_max);

/* ---- fused Compare(>) → Filter → Sum (the unfused chain, restated) ---- */
int orc_cmp_filter_sum_i64(int cmpop, const int64_t* x, const uint8_t* valid, int64_t off, int64_t n,
                           int64_t threshold, int64_t* out_sum, int64_t* out_count);
int orc_cmp_filter_sum_f64(int cmpop, const double* x, const uint8_t* valid, int64_t off, int64_t n,
                           double threshold, double* out_sum_seq, double* out_sum_exact, int64_t* out_count);

#ifdef __cplusplus
}
#endif
#endif
