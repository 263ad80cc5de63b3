// ah_cast.hip — numeric ↔ numeric cast (and bool → numeric), row §8(f)-2.
//
// Replaces, behind compute's "cast" (arrow/compute/cast.go:45-81) for the ten numeric types:
//   castNumberToNumberUnsafe → castNumericUnsafe (kernels/cast_numeric.go:28-131; AVX2 leaf
//     cast_type_numeric_avx2, kernels/_lib/cast_numeric.cc:22-27: out[i] = static_cast<O>(in[i]),
//     every slot, valid or not)
//   and the safe-cast checks the reference runs as SEPARATE passes over the column:
//     CastIntToInt → intsCanFit / intsInRange            (numeric_cast.go:37-46, helpers.go:496-652)
//     CastIntegerToFloating → checkIntToFloatTrunc        (numeric_cast.go:62-71, 698-729)
//     CastFloatingToInteger → checkFloatTrunc             (numeric_cast.go:53-60, 613-660)
//   boolToNum (numeric_cast.go:555-569).
// Here conversion and check are ONE pass: w_in + w_out bytes per row (+ 1/8 when a checked cast
// has a validity bitmap — only valid slots can fail).  A failing cast reports the FIRST offending
// row (atomicMin on its index; failures are the rare path) with the reference's message.
//
// float → int for values the target cannot hold is undefined in C++ and implementation-specific in
// Go; the reference stores whatever the CPU produced (its AVX2 kernel saturates in the vector body and
// wraps in the scalar tail).  The rule here (the CPU checker restates the same one):
// truncate toward zero into 64 bits, saturating, NaN → 0, then keep the low bits — identical to
// x86 for every |v| < 2^31 and for everything a 64-bit target can hold.
#include "ah_cast_impl.h"

using namespace ah_cast_impl;

// 32/64-bit integer inputs are instantiated in ah_cast_wide.hip
int ah_cast_from_wide(ah_ctx* c, int in_type, int out_type, const void* in, const uint8_t* valid, int64_t off, int64_t n, void* out, int aio,
                      int aft);

AH_EXPORT int ah_cast_numeric(ah_ctx* c, int in_type, int out_type, const void* values, const uint8_t* valid, int64_t off, int64_t n,
                              int allow_int_overflow, int allow_float_truncate, void* out_values) {
  AH_ENTER(c);
  if (n < 0 || off < 0) return ah_fail(c, AH_EINVALID, "cast: negative length/offset");
  const int wi = ah_type_width(in_type), wo = ah_type_width(out_type);
  if (!wi || !wo) return ah_fail(c, AH_ENOTIMPL, "cast: numeric types only (got %d → %d)", in_type, out_type);
  if (n == 0) return AH_OK;
  if (!values || !out_values) return ah_fail(c, AH_EINVALID, "cast: null buffer");
  if (((uintptr_t)values & (uintptr_t)(wi - 1)) || ((uintptr_t)out_values & (uintptr_t)(wo - 1)))
    return ah_fail(c, AH_EINVALID, "cast: buffer not element-aligned");
  if (in_type == out_type) {  // the reference's cast of a type to itself is zero-copy (cast.go:52-54); here a plain copy
    AH_HIP(c, hipMemcpyAsync(out_values, values, (size_t)n * wi, hipMemcpyDeviceToDevice, c->stream));
    return AH_OK;
  }
  switch (in_type) {
#define AH_CAST_FROM(ID, IN) case ID: return cast_from<IN>(c, out_type, values, valid, off, n, out_values, allow_int_overflow, allow_float_truncate);
    AH_CAST_FROM(AH_UINT8, uint8_t) AH_CAST_FROM(AH_INT8, int8_t) AH_CAST_FROM(AH_UINT16, uint16_t) AH_CAST_FROM(AH_INT16, int16_t)
    AH_CAST_FROM(AH_FLOAT32, float) AH_CAST_FROM(AH_FLOAT64, double)
#undef AH_CAST_FROM
    case AH_UINT32: case AH_INT32: case AH_UINT64: case AH_INT64:
      return ah_cast_from_wide(c, in_type, out_type, values, valid, off, n, out_values, allow_int_overflow, allow_float_truncate);
  }
  return ah_fail(c, AH_ENOTIMPL, "cast: unsupported input type %d", in_type);
}

AH_EXPORT int ah_cast_bool_to_numeric(ah_ctx* c, int out_type, const uint8_t* bits, int64_t off, int64_t n, void* out_values) {
  AH_ENTER(c);
  if (n < 0 || off < 0) return ah_fail(c, AH_EINVALID, "cast: negative length/offset");
  const int wo = ah_type_width(out_type);
  if (!wo) return ah_fail(c, AH_ENOTIMPL, "cast: numeric target types only (got %d)", out_type);
  if (n == 0) return AH_OK;
  if (!bits || !out_values) return ah_fail(c, AH_EINVALID, "cast: null buffer");
  const unsigned grid = (unsigned)ah_ceil_div(n, (int64_t)kBlock * (16 / wo));
  switch (out_type) {
#define AH_B2N(ID, OUT) case ID: bool_to_num_kernel<OUT><<<grid, kBlock, 0, c->stream>>>(bits, off, n, (OUT*)out_values); break;
    AH_B2N(AH_UINT8, uint8_t) AH_B2N(AH_INT8, int8_t) AH_B2N(AH_UINT16, uint16_t) AH_B2N(AH_INT16, int16_t) AH_B2N(AH_UINT32, uint32_t)
    AH_B2N(AH_INT32, int32_t) AH_B2N(AH_UINT64, uint64_t) AH_B2N(AH_INT64, int64_t) AH_B2N(AH_FLOAT32, float) AH_B2N(AH_FLOAT64, double)
#undef AH_B2N
  }
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}
