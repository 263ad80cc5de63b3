// ah_arith_ext.hip — the exact part of the arithmetic registry beyond + − ×: divide, abs / negate with
// overflow check, bit-wise and / or / xor / not, shifts, sqrt.
//
// Reference (arrow/compute/internal/kernels):
//   divide, divide_unchecked   base_arithmetic.go:154-160, 287-294 (integers: BOTH names refuse a zero divisor in
//                              a valid slot, "divide by zero"; Go's truncated quotient, MinInt / −1 wraps);
//                              :386-396 (floats: unchecked = IEEE a / b, checked refuses b == 0)
//                              — all through ScalarBinaryNotNull (helpers.go:284-380): null slots hold 0
//   abs, negate                :295-340 (signed integers: MinInt → "overflow", tested in EVERY slot — ScalarUnary
//                              walks the value buffer, helpers.go:56-90 — unsigned abs is a copy), floats :398-411
//   bit_wise_and / or / xor    scalar_arithmetic.go:170-245: bitmap ops over the raw value buffers, every slot
//   bit_wise_not               :253-268 ScalarUnaryNotNull: null slots hold 0
//   shift_left / shift_right   :293-378: a shift count outside [0, bits − 2] (signed) / [0, bits − 1] (unsigned)
//                              returns the left operand, and is "shift amount must be >= 0 and less than precision
//                              of type" for the checked names; ScalarBinaryNotNull
//   floor, ceil, trunc         rounding.go:180-187, 748-775: math.Floor / Ceil / Trunc in every slot (ScalarUnary), floats
//   sqrt, sqrt_unchecked       base_arithmetic.go:412-426: unchecked in every slot (ScalarUnary), checked NotNull
//                              with "square root of negative number"
// None of these has an assembly leaf in the reference (base_arithmetic_amd64.go:67-105: "no SIMD for POWER or
// SQRT", NotNull ops stay in Go), so the C signature below is derived from the Go closures.
//
// One kernel shape for all of them: 16 bytes per lane per operand, validity as V bits per lane, one flag word
// of error bits, 16-byte stores.  HBM-bound like ah_arith.hip except 64-bit integer division (≈ 100 VALU
// instructions per element).
#include <type_traits>

#include "ah_common.h"

namespace {

constexpr int kBlock = 256;

template <typename T>
using Vec16 = T __attribute__((ext_vector_type(16 / sizeof(T))));

// 16-byte aligned operands stream through nontemporal vector accesses (+10–15 % on this chip);
// element-aligned Arrow slices fall back to the 16-byte struct access.  `aligned` is wave-uniform.
template <typename ST>
__device__ __forceinline__ ah_vec16<ST> load16(const ST* base, int64_t i, bool aligned) {
  ah_vec16<ST> v;
  if (aligned) {
    const Vec16<ST> t = __builtin_nontemporal_load((const Vec16<ST>*)base + i);
    __builtin_memcpy(&v, &t, 16);
  } else {
    v = ah_ld16<ST>(base + i * (int64_t)(16 / sizeof(ST)));
  }
  return v;
}
template <typename ST>
__device__ __forceinline__ void store16(ST* base, int64_t i, const ah_vec16<ST>& v, bool aligned) {
  if (aligned) {
    Vec16<ST> t;
    __builtin_memcpy(&t, &v, 16);
    __builtin_nontemporal_store(t, (Vec16<ST>*)base + i);
  } else {
    ah_st16<ST>(base + i * (int64_t)(16 / sizeof(ST)), v);
  }
}

enum { ERR_OVERFLOW = 1, ERR_DIV_ZERO = 2, ERR_SHIFT = 4, ERR_NEG_SQRT = 8, ERR_NEG_POWER = 16 };
enum { X_DIV, X_DIV_CHECKED, X_SHL, X_SHL_CHECKED, X_SHR, X_SHR_CHECKED, X_POW_CHECKED, X_BIT_NOT, X_SQRT_CHECKED,  // NotNull
       X_ABS_CHECKED, X_NEG_CHECKED, X_BIT_AND, X_BIT_OR, X_BIT_XOR, X_POW, X_SQRT, X_FLOOR, X_CEIL, X_TRUNC };  // every slot

constexpr bool NotNull(int x) { return x <= X_SQRT_CHECKED; }
constexpr bool Unary(int x) { return x == X_BIT_NOT || x == X_SQRT_CHECKED || x == X_ABS_CHECKED || x == X_NEG_CHECKED || x >= X_SQRT; }

template <typename ST, int X>
__device__ __forceinline__ ST apply(ST a, ST b, unsigned& err) {
  constexpr bool kFloat = std::is_floating_point<ST>::value;
  constexpr bool kSigned = !kFloat && ((ST)-1 < (ST)0);
  constexpr int bits = sizeof(ST) * 8;
  if constexpr (X == X_DIV || X == X_DIV_CHECKED) {
    if constexpr (kFloat) {
      if (X == X_DIV_CHECKED && b == 0) { err |= ERR_DIV_ZERO; return (ST)0; }
      return a / b;
    } else {
      using U = typename std::make_unsigned<ST>::type;
      if (b == 0) { err |= ERR_DIV_ZERO; return (ST)0; }
      if constexpr (kSigned) { if (b == (ST)-1) return (ST)((U)0 - (U)a); }  // MinInt / −1 wraps (Go spec, "Integer overflow")
      return (ST)(a / b);
    }
  } else if constexpr (X == X_SHL || X == X_SHL_CHECKED || X == X_SHR || X == X_SHR_CHECKED) {
    using U = typename std::make_unsigned<ST>::type;
    constexpr ST maxshift = kSigned ? (ST)(bits - 1) : (ST)bits;  // unsigned 8-bit: bits = 8 fits
    const bool bad = kSigned ? (b < 0 || b >= maxshift) : ((unsigned long long)b >= (unsigned long long)bits);
    if (bad) { if (X == X_SHL_CHECKED || X == X_SHR_CHECKED) err |= ERR_SHIFT; return a; }
    if (X == X_SHL || X == X_SHL_CHECKED) return (ST)((U)a << (int)b);
    return (ST)(a >> (int)b);  // arithmetic for signed, logical for unsigned
  } else if constexpr (X == X_BIT_NOT) {
    return (ST)~a;
  } else if constexpr (X == X_BIT_AND) {
    return (ST)(a & b);
  } else if constexpr (X == X_BIT_OR) {
    return (ST)(a | b);
  } else if constexpr (X == X_BIT_XOR) {
    return (ST)(a ^ b);
  } else if constexpr (X == X_ABS_CHECKED || X == X_NEG_CHECKED) {
    if constexpr (kFloat) {
      return X == X_ABS_CHECKED ? (ST)__builtin_fabs(a) : -a;
    } else if constexpr (!kSigned) {
      return a;  // abs of an unsigned value; negate has no unsigned kernel
    } else {
      using U = typename std::make_unsigned<ST>::type;
      constexpr ST tmin = (ST)((U)1 << (bits - 1));
      if (a == tmin) { err |= ERR_OVERFLOW; return (ST)0; }
      return X == X_ABS_CHECKED ? (ST)(a < 0 ? -a : a) : (ST)-a;
    }
  } else if constexpr (X == X_POW || X == X_POW_CHECKED) {
    // power_unchecked / power (base_arithmetic.go:226-248, 342-373, 443-446)
    if constexpr (kFloat) {
      return (ST)::pow((double)a, (double)b);      // OutT(math.Pow(float64(a), float64(b))) under both names
    } else {
      if constexpr (kSigned) {
        if (b < 0) { err |= ERR_NEG_POWER; return (ST)0; }
      }
      if constexpr (X == X_POW) {  // right to left in uint64, narrowed at the end: wraps
        unsigned long long base = (unsigned long long)a, e = (unsigned long long)b, p = 1;
        while (e != 0) {
          if (e & 1) p *= base;
          base *= base;
          e >>= 1;
        }
        return (ST)p;
      } else {  // left to right with mulWithOverflow (:84-108: an overflowing product is 0 and the flag sticks)
        if (b == 0) return (ST)1;
        const unsigned long long ue = (unsigned long long)b;
        unsigned long long mask = 1ull << (63 - __builtin_clzll(ue));
        ST p = (ST)1;
        bool of = false;
        while (mask != 0) {
          ST t;
          if (__builtin_mul_overflow(p, p, &t)) { of = true; t = (ST)0; }
          p = t;
          if (ue & mask) {
            if (__builtin_mul_overflow(p, a, &t)) { of = true; t = (ST)0; }
            p = t;
          }
          mask >>= 1;
        }
        if (of) err |= ERR_OVERFLOW;
        return p;
      }
    }
  } else if constexpr (X == X_FLOOR || X == X_CEIL || X == X_TRUNC) {
    // getFloatRoundImpl (rounding.go:180-187): math.Floor / Ceil / Trunc of the value widened to float64 and narrowed
    // back — exact in the narrow type as well
    if constexpr (kFloat) {
      const double v = (double)a;
      return (ST)(X == X_FLOOR ? __builtin_floor(v) : X == X_CEIL ? __builtin_ceil(v) : __builtin_trunc(v));
    } else {
      return a;
    }
  } else {  // X_SQRT, X_SQRT_CHECKED
    if constexpr (kFloat) {
      if (X == X_SQRT_CHECKED && a < 0) { err |= ERR_NEG_SQRT; return (ST)__builtin_nan(""); }
      return sizeof(ST) == 4 ? (ST)__builtin_sqrtf((float)a) : (ST)__builtin_sqrt((double)a);
    } else {
      return a;
    }
  }
}

// SHAPE: 0 = l[i] ∘ r[i], 1 = l[i] ∘ scalar (and all unary ops), 2 = scalar ∘ r[i]
template <typename ST, int X, int SHAPE>
__global__ __launch_bounds__(kBlock) void ext_kernel(const ST* __restrict__ l, const uint8_t* __restrict__ lv, int64_t loff,
                                                      const ST* __restrict__ r, const uint8_t* __restrict__ rv, int64_t roff,
                                                      ST scalar, ST* __restrict__ out, int64_t len, unsigned* __restrict__ flag, int aligned) {
  constexpr int V = 16 / sizeof(ST);
  using VT = ah_vec16<ST>;
  unsigned err = 0;
  const int64_t nvec = len / V;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    VT a, b, o;
    if (SHAPE != 2) a = load16<ST>(l, i, aligned);
    if (SHAPE != 1) b = load16<ST>(r, i, aligned);
    unsigned vbits = (1u << V) - 1;
    if (NotNull(X)) {
      if (SHAPE != 2 && lv) vbits &= (unsigned)ah_load_bits64(lv, loff + i * V, V);
      if (SHAPE != 1 && rv) vbits &= (unsigned)ah_load_bits64(rv, roff + i * V, V);
    }
#pragma unroll
    for (int e = 0; e < V; e++) {
      unsigned e1 = 0;
      const ST v = apply<ST, X>(SHAPE == 2 ? scalar : a.v[e], SHAPE == 1 ? scalar : b.v[e], e1);
      const bool valid = (vbits >> e) & 1;
      o.v[e] = valid ? v : (ST)0;   // helpers.go:303-306: null slots hold the zero value
      if (valid) err |= e1;
    }
    store16<ST>(out, i, o, aligned);
  }
  if (blockIdx.x == 0) {  // < V trailing elements
    const int64_t j = nvec * V + threadIdx.x;
    if (j < len) {
      const bool valid = !NotNull(X) || ((SHAPE == 2 || ah_bit(lv, loff + j)) && (SHAPE == 1 || ah_bit(rv, roff + j)));
      unsigned e1 = 0;
      const ST v = apply<ST, X>(SHAPE == 2 ? scalar : l[j], SHAPE == 1 ? scalar : r[j], e1);
      out[j] = valid ? v : (ST)0;
      if (valid) err |= e1;
    }
  }
  // one atomic per wave that saw an error, none otherwise
  for (unsigned bit = 1; bit <= ERR_NEG_POWER; bit <<= 1)
    if (__any((err & bit) != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}

template <typename ST, int X>
int launch(ah_ctx* c, int shape, const void* l, const uint8_t* lv, int64_t loff, const void* r, const uint8_t* rv, int64_t roff, void* out,
           int64_t len, unsigned* flag) {
  ST scalar = 0;
  if (!Unary(X)) {
    if (shape == AH_SHAPE_AS) memcpy(&scalar, r, sizeof(ST));
    if (shape == AH_SHAPE_SA) memcpy(&scalar, l, sizeof(ST));
  }
  // exact grid, one vector per lane: same finding as ah_arith.hip (no grid-stride tail, maximum loads in flight)
  const unsigned grid = ah_stream_grid(c, ah_ceil_div(len / (16 / (int64_t)sizeof(ST)) + 1, kBlock), /*default_bpc=*/0);
  const ST* pl = (const ST*)l; const ST* pr = (const ST*)r; ST* po = (ST*)out;
  const bool arr_l = !(shape == AH_SHAPE_SA && !Unary(X)), arr_r = !Unary(X) && shape != AH_SHAPE_AS;
  const int aligned = c->tune_nt && ((((uintptr_t)out) | (arr_l ? (uintptr_t)l : 0) | (arr_r ? (uintptr_t)r : 0)) & 15) == 0;
  if (Unary(X) || shape == AH_SHAPE_AS) ext_kernel<ST, X, 1><<<grid, kBlock, 0, c->stream>>>(pl, lv, loff, nullptr, nullptr, 0, scalar, po, len, flag, aligned);
  else if (shape == AH_SHAPE_AA) ext_kernel<ST, X, 0><<<grid, kBlock, 0, c->stream>>>(pl, lv, loff, pr, rv, roff, scalar, po, len, flag, aligned);
  else ext_kernel<ST, X, 2><<<grid, kBlock, 0, c->stream>>>(nullptr, nullptr, 0, pr, rv, roff, scalar, po, len, flag, aligned);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <typename ST>
int dispatch_int(ah_ctx* c, int op, int shape, const void* l, const uint8_t* lv, int64_t loff, const void* r, const uint8_t* rv, int64_t roff,
                 void* out, int64_t len, unsigned* flag) {
#define AH_X(X) return launch<ST, X>(c, shape, l, lv, loff, r, rv, roff, out, len, flag)
  switch (op) {
    case AH_OP_DIV: AH_X(X_DIV);
    case AH_OP_DIV_CHECKED: AH_X(X_DIV_CHECKED);
    case AH_OP_SHIFT_LEFT: AH_X(X_SHL);
    case AH_OP_SHIFT_LEFT_CHECKED: AH_X(X_SHL_CHECKED);
    case AH_OP_SHIFT_RIGHT: AH_X(X_SHR);
    case AH_OP_SHIFT_RIGHT_CHECKED: AH_X(X_SHR_CHECKED);
    case AH_OP_POWER: AH_X(X_POW);
    case AH_OP_POWER_CHECKED: AH_X(X_POW_CHECKED);
    case AH_OP_BIT_NOT: AH_X(X_BIT_NOT);
    case AH_OP_BIT_AND: AH_X(X_BIT_AND);
    case AH_OP_BIT_OR: AH_X(X_BIT_OR);
    case AH_OP_BIT_XOR: AH_X(X_BIT_XOR);
    case AH_OP_ABS_CHECKED: AH_X(X_ABS_CHECKED);
    case AH_OP_NEGATE_CHECKED:
      if ((ST)-1 > (ST)0) break;  // GetArithmeticUnarySignedKernels: no unsigned negate
      AH_X(X_NEG_CHECKED);
  }
  return ah_fail(c, AH_ENOTIMPL, "arithmetic: op %d is not defined for this integer type", op);
}

template <typename ST>
int dispatch_float(ah_ctx* c, int op, int shape, const void* l, const uint8_t* lv, int64_t loff, const void* r, const uint8_t* rv, int64_t roff,
                   void* out, int64_t len, unsigned* flag) {
  switch (op) {
    case AH_OP_DIV: AH_X(X_DIV);
    case AH_OP_DIV_CHECKED: AH_X(X_DIV_CHECKED);
    case AH_OP_ABS_CHECKED: AH_X(X_ABS_CHECKED);
    case AH_OP_NEGATE_CHECKED: AH_X(X_NEG_CHECKED);
    case AH_OP_POWER: case AH_OP_POWER_CHECKED: AH_X(X_POW);
    case AH_OP_SQRT: AH_X(X_SQRT);
    case AH_OP_SQRT_CHECKED: AH_X(X_SQRT_CHECKED);
    case AH_OP_FLOOR: AH_X(X_FLOOR);
    case AH_OP_CEIL: AH_X(X_CEIL);
    case AH_OP_TRUNC: AH_X(X_TRUNC);
  }
#undef AH_X
  return ah_fail(c, AH_ENOTIMPL, "arithmetic: op %d is not defined for floating point", op);
}

// ---- round / round_to_multiple (kernels/rounding.go:321-370, 562-598) --------------------------------------------
// MODE: RoundMode (rounding.go:40-59).  MULTIPLE: round_to_multiple (scale = the multiple: divide, round, multiply);
// otherwise scale = 10^|ndigits| (multiply first when ndigits ≥ 0, divide first when negative).  Arithmetic in T, the
// rounding primitives in double — as the Go code has it.  Inf / NaN and values that are integral after scaling pass
// through; a non-finite result in a valid slot is "overflow".  ScalarUnaryNotNull: null slots hold 0.
template <typename T>
__device__ __forceinline__ T round_impl(T v, int mode) {
  const double d = (double)v;
  switch (mode) {
    case 0: case 4: return (T)__builtin_floor(d);                                   // RoundDown, HalfDown (tie)
    case 1: case 5: return (T)__builtin_ceil(d);                                    // RoundUp, HalfUp (tie)
    case 2: case 6: return (T)__builtin_trunc(d);                                   // TowardsZero, HalfTowardsZero (tie)
    case 3: case 7: return (T)(__builtin_signbit(d) ? __builtin_floor(d) : __builtin_ceil(d));  // AwayFromZero, HalfAwayFromZero (tie)
    case 8: return (T)__builtin_rint(d);                                            // HalfToEven: math.RoundToEven
    default: return (T)(__builtin_floor(d * 0.5) + __builtin_ceil(d * 0.5));        // HalfToOdd
  }
}

template <typename T, bool MULTIPLE>
__global__ __launch_bounds__(kBlock) void round_kernel(const T* __restrict__ in, const uint8_t* __restrict__ valid, int64_t off, int64_t n,
                                                        T scale, int ndigits_sign, int mode, T* __restrict__ out, unsigned* __restrict__ flag) {
  unsigned err = 0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    const T arg = in[i];
    T res = arg;
    if (!ah_bit(valid, off + i)) {
      res = (T)0;
    } else if (!(__builtin_isinf((double)arg) || __builtin_isnan((double)arg))) {
      T rv = (MULTIPLE || ndigits_sign < 0) ? arg / scale : arg * scale;
      const T frac = rv - (T)__builtin_floor((double)rv);
      if (frac != (T)0) {
        if (mode >= 4 && frac != (T)0.5) rv = (T)__builtin_round((double)rv);  // math.Round: half away from zero (not a tie here)
        else rv = round_impl<T>(rv, mode);
        if (MULTIPLE) rv *= scale;
        else if (ndigits_sign > 0) rv /= scale;
        else rv *= scale;
        if (__builtin_isinf((double)rv) || __builtin_isnan((double)rv)) err |= ERR_OVERFLOW;
        else res = rv;
      }
    }
    out[i] = res;
  }
  if (__any(err != 0) && (threadIdx.x & 63) == 0) atomicOr(flag, (unsigned)ERR_OVERFLOW);
}

bool is_unary_op(int op) {
  return op == AH_OP_BIT_NOT || op == AH_OP_ABS_CHECKED || op == AH_OP_NEGATE_CHECKED || op == AH_OP_SQRT || op == AH_OP_SQRT_CHECKED ||
         op == AH_OP_FLOOR || op == AH_OP_CEIL || op == AH_OP_TRUNC;
}

}  // namespace

AH_EXPORT int ah_arithmetic_ext(ah_ctx* c, int type, int op, int shape, const void* l, const uint8_t* lvalid, int64_t loff, const void* r,
                                const uint8_t* rvalid, int64_t roff, int scalar_valid, void* out, int64_t len) {
  AH_ENTER(c);
  if (len < 0 || loff < 0 || roff < 0) return ah_fail(c, AH_EINVALID, "arithmetic: negative length/offset");
  if (len == 0) return AH_OK;
  const bool unary = is_unary_op(op);
  if (unary) shape = AH_SHAPE_AS;
  if (shape < AH_SHAPE_AA || shape > AH_SHAPE_SA) return ah_fail(c, AH_EINVALID, "arithmetic: bad shape %d", shape);
  const int w = ah_type_width(type);
  if (!w) return ah_fail(c, AH_ENOTIMPL, "arithmetic: unsupported type id %d", type);
  if (!out || (shape != AH_SHAPE_SA && !l) || (!unary && !r) || (shape == AH_SHAPE_SA && !l)) return ah_fail(c, AH_EINVALID, "arithmetic: null buffer");
  const void* arr0 = shape == AH_SHAPE_SA ? r : l;
  if ((((uintptr_t)arr0 | (uintptr_t)out | (shape == AH_SHAPE_AA ? (uintptr_t)r : 0)) & (uintptr_t)(w - 1)) != 0)
    return ah_fail(c, AH_EINVALID, "arithmetic: buffer not element-aligned");
  const bool every_slot = op == AH_OP_BIT_AND || op == AH_OP_BIT_OR || op == AH_OP_BIT_XOR || op == AH_OP_ABS_CHECKED || op == AH_OP_NEGATE_CHECKED ||
                          op == AH_OP_SQRT || op == AH_OP_FLOOR || op == AH_OP_CEIL || op == AH_OP_TRUNC || op == AH_OP_POWER ||
                          (op == AH_OP_POWER_CHECKED && (type == AH_FLOAT32 || type == AH_FLOAT64));
  if (!every_slot && !unary && shape != AH_SHAPE_AA && !scalar_valid) {
    // null scalar: the output stays as allocated = zero (helpers.go:312-314, 341-343)
    AH_HIP(c, hipMemsetAsync(out, 0, (size_t)len * w, c->stream));
    return AH_OK;
  }
  unsigned* flag = (unsigned*)c->dscalars;
  AH_HIP(c, hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream));
  int rc;
  switch (type) {
    case AH_UINT8: rc = dispatch_int<uint8_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_INT8: rc = dispatch_int<int8_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_UINT16: rc = dispatch_int<uint16_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_INT16: rc = dispatch_int<int16_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_UINT32: rc = dispatch_int<uint32_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_INT32: rc = dispatch_int<int32_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_UINT64: rc = dispatch_int<uint64_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_INT64: rc = dispatch_int<int64_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_FLOAT32: rc = dispatch_float<float>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_FLOAT64: rc = dispatch_float<double>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    default: return ah_fail(c, AH_ENOTIMPL, "arithmetic: unsupported type id %d", type);
  }
  if (rc != AH_OK) return rc;
  if (op == AH_OP_DIV && (type == AH_FLOAT32 || type == AH_FLOAT64)) return AH_OK;  // cannot fail: no readback
  if (op == AH_OP_BIT_AND || op == AH_OP_BIT_OR || op == AH_OP_BIT_XOR || op == AH_OP_BIT_NOT || op == AH_OP_SQRT || op == AH_OP_SHIFT_LEFT ||
      op == AH_OP_SHIFT_RIGHT || op == AH_OP_FLOOR || op == AH_OP_CEIL || op == AH_OP_TRUNC)
    return AH_OK;
  AH_HIP(c, hipMemcpyAsync(c->pinned, flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  const unsigned f = *(volatile unsigned*)c->pinned;
  if (f & ERR_OVERFLOW) return ah_fail(c, AH_EOVERFLOW, "overflow");
  if (f & ERR_DIV_ZERO) return ah_fail(c, AH_EINVALID, "divide by zero");
  if (f & ERR_SHIFT) return ah_fail(c, AH_EINVALID, "shift amount must be >= 0 and less than precision of type");
  if (f & ERR_NEG_SQRT) return ah_fail(c, AH_EINVALID, "square root of negative number");
  if (f & ERR_NEG_POWER) return ah_fail(c, AH_EINVALID, "integers to negative integer powers are not allowed");
  return AH_OK;
}

AH_EXPORT int ah_round(ah_ctx* c, int type, const void* values, const uint8_t* valid, int64_t off, int64_t n, int64_t ndigits, int mode,
                       const void* multiple_host, double pow10, void* out) {
  AH_ENTER(c);
  if (n < 0 || off < 0) return ah_fail(c, AH_EINVALID, "round: negative length/offset");
  if (mode < 0 || mode > 9) return ah_fail(c, AH_EINVALID, "round: invalid rounding mode %d", mode);
  if (type != AH_FLOAT32 && type != AH_FLOAT64) return ah_fail(c, AH_ENOTIMPL, "round: unsupported type id %d", type);
  if (n == 0) return AH_OK;
  if (!values || !out) return ah_fail(c, AH_EINVALID, "round: null buffer");
  unsigned* flag = (unsigned*)c->dscalars;
  AH_HIP(c, hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream));
  const unsigned grid = ah_stream_grid(c, ah_ceil_div(n, kBlock), /*default_bpc=*/8);
  const int sgn = ndigits > 0 ? 1 : (ndigits < 0 ? -1 : 0);
  if (type == AH_FLOAT32) {
    float scale = (float)pow10;
    if (multiple_host) { memcpy(&scale, multiple_host, 4); round_kernel<float, true><<<grid, kBlock, 0, c->stream>>>((const float*)values, valid, off, n, scale, 0, mode, (float*)out, flag); }
    else round_kernel<float, false><<<grid, kBlock, 0, c->stream>>>((const float*)values, valid, off, n, scale, sgn, mode, (float*)out, flag);
  } else {
    double scale = pow10;
    if (multiple_host) { memcpy(&scale, multiple_host, 8); round_kernel<double, true><<<grid, kBlock, 0, c->stream>>>((const double*)values, valid, off, n, scale, 0, mode, (double*)out, flag); }
    else round_kernel<double, false><<<grid, kBlock, 0, c->stream>>>((const double*)values, valid, off, n, scale, sgn, mode, (double*)out, flag);
  }
  AH_LAUNCH_CHECK(c);
  AH_HIP(c, hipMemcpyAsync(c->pinned, flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  if (*(volatile unsigned*)c->pinned & ERR_OVERFLOW) return ah_fail(c, AH_EOVERFLOW, "overflow");
  return AH_OK;
}
