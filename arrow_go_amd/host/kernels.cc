// kernels.cc — the kernel library: one ExecFn per function family, each a thin adapter
// from exec.ArraySpan to a leaf of include/arrowhip.h, plus the Register* entry points.
//
// ≙ arrow/compute/internal/kernels/{helpers.go:193-236 (ScalarBinary), :284-380
//   (ScalarBinaryNotNull), scalar_arithmetic.go, scalar_comparisons.go:199-218,
//   scalar_boolean.go:67-347, vector_selection.go:449-520,1162-1192,
//   vector_hash.go:620-741} and arrow/compute/{arithmetic.go, scalar_compare.go,
//   scalar_bool.go:94-110, selection.go:42-114,618-650, vector_hash.go:61-100}.
#include <cmath>
#include "arrowhip_compute.h"

#include <cstring>

namespace arrowhip {
namespace compute {

using exec::ArraySpan;
using exec::ExecResult;
using exec::ExecSpan;
using exec::KernelCtx;

static const Type kNumericTypes[] = {Type::UINT8, Type::INT8,  Type::UINT16, Type::INT16,   Type::UINT32,
                                     Type::INT32, Type::UINT64, Type::INT64, Type::FLOAT32, Type::FLOAT64};

// exec.GetSpanValues (exec/utils.go:38-45): values pointer with the span offset applied
static inline const uint8_t* Values(const ArraySpan& a) { return a.buffers[1].buf + a.offset * (a.type->bit_width / 8); }
static inline uint8_t* Values(ArraySpan* a) { return a->buffers[1].buf + a->offset * (a->type->bit_width / 8); }

static int ShapeOf(const ExecSpan& b) {
  if (b.values[0].IsArray()) return b.values[1].IsArray() ? AH_SHAPE_AA : AH_SHAPE_AS;
  return AH_SHAPE_SA;
}

// ---- arithmetic ---------------------------------------------------------------------------
struct ArithData { int op; bool checked; };

// ScalarBinary (helpers.go:193-236) over the unchecked SIMD leaves
static Status ExecArithUnchecked(KernelCtx* k, const ExecSpan& b, ExecResult* out, int op) {
  Session* s = k->session;
  int type = (int)out->type->id;
  int shape = ShapeOf(b);
  if (out->len == 0) return Status::OK();
  switch (shape) {
    case AH_SHAPE_AA:
      return s->FromStatus(ah_arithmetic_binary(s->ctx(), type, (int8_t)op, Values(b.values[0].array), Values(b.values[1].array), Values(out), out->len));
    case AH_SHAPE_AS:
      return s->FromStatus(ah_arithmetic_arr_scalar(s->ctx(), type, (int8_t)op, Values(b.values[0].array), b.values[1].scalar->value, Values(out), out->len));
    default:
      return s->FromStatus(ah_arithmetic_scalar_arr(s->ctx(), type, (int8_t)op, b.values[0].scalar->value, Values(b.values[1].array), Values(out), out->len));
  }
}

// ScalarBinaryNotNull (helpers.go:284-380) for the checked integer ops
static Status ExecArithChecked(KernelCtx* k, const ExecSpan& b, ExecResult* out, int op) {
  Session* s = k->session;
  int type = (int)out->type->id;
  if (IsFloating(out->type->id)) return ExecArithUnchecked(k, b, out, op);  // base_arithmetic_amd64.go:109-117
  if (out->len == 0) return Status::OK();
  int shape = ShapeOf(b);
  const void *l, *r;
  const uint8_t *lv = nullptr, *rv = nullptr;
  int64_t lo = 0, ro = 0;
  int scalar_valid = 1;
  if (b.values[0].IsArray()) { l = Values(b.values[0].array); lv = b.values[0].array.buffers[0].buf; lo = b.values[0].array.offset; }
  else { l = b.values[0].scalar->value; scalar_valid = b.values[0].scalar->valid; }
  if (b.values[1].IsArray()) { r = Values(b.values[1].array); rv = b.values[1].array.buffers[0].buf; ro = b.values[1].array.offset; }
  else { r = b.values[1].scalar->value; scalar_valid = b.values[1].scalar->valid; }
  return s->FromStatus(ah_arithmetic_checked(s->ctx(), type, (int8_t)op, shape, l, lv, lo, r, rv, ro, scalar_valid, Values(out), out->len));
}

// Go's math.Pow10 (src/math/pow10.go): a product of two table entries — pow10tab[n % 32] · pow10postab32[n / 32] — which
// is what InitRoundState hands the round kernel (rounding.go:86-90); beyond 1e31 it is NOT always the correctly rounded
// literal, so it is restated rather than replaced by pow()
static double GoPow10(int64_t n) {
  static const double tab[32] = {1e00, 1e01, 1e02, 1e03, 1e04, 1e05, 1e06, 1e07, 1e08, 1e09, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22, 1e23, 1e24, 1e25, 1e26, 1e27, 1e28, 1e29, 1e30, 1e31};
  static const double pos32[10] = {1e00, 1e32, 1e64, 1e96, 1e128, 1e160, 1e192, 1e224, 1e256, 1e288};
  if (n < 0) return 0;
  if (n > 308) return HUGE_VAL;
  return pos32[n / 32] * tab[n % 32];
}

// ScalarBinaryNotNull / ScalarUnaryNotNull / whole-buffer closures of the pure-Go arithmetic kernels: divide, abs, negate,
// bit-wise, shifts, sqrt (base_arithmetic.go:154-160,287-340,386-426; scalar_arithmetic.go:170-378)
static Status ExecArithExt(KernelCtx* k, const ExecSpan& b, ExecResult* out, int op) {
  Session* s = k->session;
  if (out->len == 0) return Status::OK();
  const int type = (int)out->type->id;
  const void *l = nullptr, *r = nullptr;
  const uint8_t *lv = nullptr, *rv = nullptr;
  int64_t lo = 0, ro = 0;
  int scalar_valid = 1, shape = AH_SHAPE_AS;
  if (b.values.size() == 2) shape = ShapeOf(b);
  if (b.values[0].IsArray()) { l = Values(b.values[0].array); lv = b.values[0].array.MayHaveNulls() ? b.values[0].array.buffers[0].buf : nullptr; lo = b.values[0].array.offset; }
  else { l = b.values[0].scalar->value; scalar_valid = b.values[0].scalar->valid; }
  if (b.values.size() == 2) {
    if (b.values[1].IsArray()) { r = Values(b.values[1].array); rv = b.values[1].array.MayHaveNulls() ? b.values[1].array.buffers[0].buf : nullptr; ro = b.values[1].array.offset; }
    else { r = b.values[1].scalar->value; scalar_valid = b.values[1].scalar->valid; }
  }
  return s->FromStatus(ah_arithmetic_ext(s->ctx(), type, op, shape, l, lv, lo, r, rv, ro, scalar_valid, Values(out), out->len));
}

static std::shared_ptr<ScalarFunction> MakeArithExt(const std::string& name, int op, int nargs, std::initializer_list<Type> types) {
  auto fn = std::make_shared<ScalarFunction>(name, Arity{nargs, false});
  for (Type t : types) {
    exec::ScalarKernel k;
    k.sig.in_types = nargs == 2 ? std::vector<Type>{t, t} : std::vector<Type>{t};
    k.sig.out_is_first_input = true;
    k.exec_fn = [op](KernelCtx* c, const ExecSpan& b, ExecResult* o) { return ExecArithExt(c, b, o, op); };
    fn->AddKernel(std::move(k));
  }
  fn->promote_numeric = true;  // arithmeticFunction.DispatchBest (arithmetic.go:112-142)
  return fn;
}

static Status ExecUnary(KernelCtx* k, const ExecSpan& b, ExecResult* out, int op) {
  Session* s = k->session;
  if (out->len == 0) return Status::OK();
  return s->FromStatus(ah_arithmetic_unary(s->ctx(), (int)out->type->id, (int8_t)op, Values(b.values[0].array), Values(out), out->len));
}

static std::shared_ptr<ScalarFunction> MakeArith(const std::string& name, int op, bool checked) {
  auto fn = std::make_shared<ScalarFunction>(name, Arity{2, false});
  for (Type t : kNumericTypes) {
    exec::ScalarKernel k;
    k.sig.in_types = {t, t};
    k.sig.out_is_first_input = true;
    k.exec_fn = checked ? exec::ArrayKernelExec([op](KernelCtx* c, const ExecSpan& b, ExecResult* o) { return ExecArithChecked(c, b, o, op); })
                        : exec::ArrayKernelExec([op](KernelCtx* c, const ExecSpan& b, ExecResult* o) { return ExecArithUnchecked(c, b, o, op); });
    fn->AddKernel(std::move(k));
  }
  fn->promote_numeric = true;  // arithmeticFunction.DispatchBest (arithmetic.go:112-142)
  return fn;
}

void RegisterScalarArithmetic(FunctionRegistry* reg) {
  // compute/arithmetic.go: "add" / "subtract" / "multiply" are the CHECKED kernels (the
  // default of compute.Add, :1090-1104), "*_unchecked" the wrapping SIMD ones
  reg->AddFunction(MakeArith("add", AH_OP_ADD_CHECKED, true), false);
  reg->AddFunction(MakeArith("add_unchecked", AH_OP_ADD, false), false);
  reg->AddFunction(MakeArith("subtract", AH_OP_SUB_CHECKED, true), false);
  reg->AddFunction(MakeArith("subtract_unchecked", AH_OP_SUB, false), false);
  // … and under the names compute.Subtract itself calls: impl(ctx, "sub", …) + "_unchecked" with NoCheckOverflow (arithmetic.go:679-682, 1115-1117)
  reg->AddFunction(MakeArith("sub", AH_OP_SUB_CHECKED, true), false);
  reg->AddFunction(MakeArith("sub_unchecked", AH_OP_SUB, false), false);
  reg->AddFunction(MakeArith("multiply", AH_OP_MUL_CHECKED, true), false);
  reg->AddFunction(MakeArith("multiply_unchecked", AH_OP_MUL, false), false);
  // the pure-Go rest of the registry that is exact (arithmetic.go:784-785, 822-840, 855-856, 944-995)
  const auto ints = {Type::UINT8, Type::INT8, Type::UINT16, Type::INT16, Type::UINT32, Type::INT32, Type::UINT64, Type::INT64};
  const auto nums = {Type::UINT8, Type::INT8, Type::UINT16, Type::INT16, Type::UINT32, Type::INT32, Type::UINT64, Type::INT64, Type::FLOAT32, Type::FLOAT64};
  const auto sgn = {Type::INT8, Type::INT16, Type::INT32, Type::INT64, Type::FLOAT32, Type::FLOAT64};
  const auto flt = {Type::FLOAT32, Type::FLOAT64};
  reg->AddFunction(MakeArithExt("divide", AH_OP_DIV_CHECKED, 2, nums), false);
  reg->AddFunction(MakeArithExt("divide_unchecked", AH_OP_DIV, 2, nums), false);
  reg->AddFunction(MakeArithExt("power", AH_OP_POWER_CHECKED, 2, nums), false);             // arithmetic.go:929-930
  reg->AddFunction(MakeArithExt("power_unchecked", AH_OP_POWER, 2, nums), false);
  reg->AddFunction(MakeArithExt("abs", AH_OP_ABS_CHECKED, 1, nums), false);
  reg->AddFunction(MakeArithExt("negate", AH_OP_NEGATE_CHECKED, 1, sgn), false);  // GetArithmeticUnarySignedKernels: no unsigned kernel
  reg->AddFunction(MakeArithExt("bit_wise_and", AH_OP_BIT_AND, 2, ints), false);
  reg->AddFunction(MakeArithExt("bit_wise_or", AH_OP_BIT_OR, 2, ints), false);
  reg->AddFunction(MakeArithExt("bit_wise_xor", AH_OP_BIT_XOR, 2, ints), false);
  reg->AddFunction(MakeArithExt("bit_wise_not", AH_OP_BIT_NOT, 1, ints), false);
  reg->AddFunction(MakeArithExt("shift_left", AH_OP_SHIFT_LEFT_CHECKED, 2, ints), false);
  reg->AddFunction(MakeArithExt("shift_left_unchecked", AH_OP_SHIFT_LEFT, 2, ints), false);
  reg->AddFunction(MakeArithExt("shift_right", AH_OP_SHIFT_RIGHT_CHECKED, 2, ints), false);
  reg->AddFunction(MakeArithExt("shift_right_unchecked", AH_OP_SHIFT_RIGHT, 2, ints), false);
  for (auto p : {std::make_pair("sqrt", AH_OP_SQRT_CHECKED), std::make_pair("sqrt_unchecked", AH_OP_SQRT)}) {
    auto fn = MakeArithExt(p.first, p.second, 1, flt);
    fn->promote_to_float = true;  // arithmeticFloatingPointFunc.DispatchBest (arithmetic.go:144-170): integers go to float64
    reg->AddFunction(fn, false);
  }
  for (auto p : {std::make_pair("floor", AH_OP_FLOOR), std::make_pair("ceil", AH_OP_CEIL), std::make_pair("trunc", AH_OP_TRUNC)}) {
    auto fn = MakeArithExt(p.first, p.second, 1, flt);  // GetSimpleRoundKernels (rounding.go:748-775)
    fn->promote_to_float = true;                         // arithmeticIntegerToFloatingPointFunc (arithmetic.go:172-200)
    reg->AddFunction(fn, false);
  }
  // round / round_to_multiple (arithmetic.go:1036-1056; kernels/rounding.go:321-370, 562-598)
  {
    static const RoundOptions kDefaultRound;                  // {NDigits: 0, Mode: RoundHalfToEven} (arithmetic.go:78)
    static const RoundToMultipleOptions kDefaultRoundMultiple = [] {  // Multiple: float64 1 (:79-80)
      RoundToMultipleOptions o;
      auto one = std::make_shared<Scalar>();
      one->type = GetDataType(Type::FLOAT64); one->valid = true;
      const double v = 1.0; memcpy(one->value, &v, 8);
      o.Multiple = one;
      return o;
    }();
    for (bool multiple : {false, true}) {
      auto fn = std::make_shared<ScalarFunction>(multiple ? "round_to_multiple" : "round", Arity{1, false});
      fn->SetDefaultOptions(multiple ? (const FunctionOptions*)&kDefaultRoundMultiple : (const FunctionOptions*)&kDefaultRound);
      for (Type t : flt) {
        exec::ScalarKernel k;
        k.sig.in_types = {t};
        k.sig.out_is_first_input = true;
        k.exec_fn = [multiple](KernelCtx* c, const ExecSpan& b, ExecResult* out) -> Status {
          Session* s = c->session;
          const ArraySpan& in = b.values[0].array;
          const uint8_t* valid = in.MayHaveNulls() ? in.buffers[0].buf : nullptr;
          const int type = (int)out->type->id;
          if (!multiple) {
            const RoundOptions* o = dynamic_cast<const RoundOptions*>(static_cast<const FunctionOptions*>(c->state));
            if (!o) return Status::Make(StatusCode::Invalid, "attempted to initialize kernel state from invalid function options");  // InitRoundState
            if (out->len == 0) return Status::OK();
            const int64_t ad = o->NDigits < 0 ? -o->NDigits : o->NDigits;
            return s->FromStatus(ah_round(s->ctx(), type, Values(in), valid, in.offset, out->len, o->NDigits, (int)o->Mode, nullptr, GoPow10(ad), Values(out)));
          }
          const RoundToMultipleOptions* o = dynamic_cast<const RoundToMultipleOptions*>(static_cast<const FunctionOptions*>(c->state));
          if (!o) return Status::Make(StatusCode::Invalid, "attempted to initialize kernel state from invalid function options");
          // InitRoundToMultipleState (rounding.go:127-177)
          if (!o->Multiple || !o->Multiple->valid) return Status::Make(StatusCode::Invalid, "rounding multiple must be non-null and valid");
          const Scalar& m = *o->Multiple;
          double mv = 0;
          bool positive = false;
          if (m.type->id == Type::FLOAT64) { memcpy(&mv, m.value, 8); positive = mv > 0; }
          else if (m.type->id == Type::FLOAT32) { float f; memcpy(&f, m.value, 4); mv = f; positive = f > 0; }
          else if (IsSignedInteger(m.type->id)) { long long x = 0; memcpy(&x, m.value, m.type->bit_width / 8); if (m.type->bit_width < 64) x = (x << (64 - m.type->bit_width)) >> (64 - m.type->bit_width); mv = (double)x; positive = x > 0; }
          else if (IsInteger(m.type->id)) { unsigned long long x = 0; memcpy(&x, m.value, m.type->bit_width / 8); mv = (double)x; positive = true; }  // isPositive: unsigned is always "positive"
          else return Status::Make(StatusCode::Invalid, "rounding multiple must be positive");
          if (!positive) return Status::Make(StatusCode::Invalid, "rounding multiple must be positive");
          // an integer multiple is cast to float64, a floating one to the input type (:157-175)
          uint8_t mbuf[8];
          if (type == AH_FLOAT32) { const float f = (float)mv; memcpy(mbuf, &f, 4); } else memcpy(mbuf, &mv, 8);
          if (out->len == 0) return Status::OK();
          return s->FromStatus(ah_round(s->ctx(), type, Values(in), valid, in.offset, out->len, 0, (int)o->Mode, mbuf, 1.0, Values(out)));
        };
        fn->AddKernel(std::move(k));
      }
      fn->promote_to_float = true;  // arithmeticIntegerToFloatingPointFunc
      reg->AddFunction(fn, false);
    }
  }
  struct U { const char* name; int op; };
  for (U u : {U{"abs_unchecked", AH_OP_ABS}, U{"negate_unchecked", AH_OP_NEGATE}, U{"sign", AH_OP_SIGN}}) {
    auto fn = std::make_shared<ScalarFunction>(u.name, Arity{1, false});
    for (Type t : kNumericTypes) {
      exec::ScalarKernel k;
      k.sig.in_types = {t};
      int op = u.op;
      k.exec_fn = [op](KernelCtx* c, const ExecSpan& b, ExecResult* o) { return ExecUnary(c, b, o, op); };
      fn->AddKernel(std::move(k));
    }
    reg->AddFunction(fn, false);
  }
}

// ---- comparisons --------------------------------------------------------------------------
// compareKernel[T] (scalar_comparisons.go:199-218): out bitmap starts at byte out.Offset/8,
// bit prefix out.Offset%8
static Status ExecCompare(KernelCtx* k, const ExecSpan& b, ExecResult* out, int cmpop) {
  Session* s = k->session;
  if (out->len == 0) return Status::OK();
  int shape = ShapeOf(b);
  const DataType* in_type = b.values[0].type();
  const void* l = b.values[0].IsArray() ? (const void*)Values(b.values[0].array) : (const void*)b.values[0].scalar->value;
  const void* r = b.values[1].IsArray() ? (const void*)Values(b.values[1].array) : (const void*)b.values[1].scalar->value;
  return s->FromStatus(ah_comparison(s->ctx(), cmpop, shape, (int)in_type->id, l, r, out->buffers[1].buf + out->offset / 8, out->len, (int)(out->offset % 8)));
}

static Status ExecBoolBinary(KernelCtx* k, const ExecSpan& b, ExecResult* out, int bitop);

void RegisterScalarComparisons(FunctionRegistry* reg) {
  struct C { const char* name; int op; };
  for (C c : {C{"equal", AH_CMP_EQ}, C{"not_equal", AH_CMP_NE}, C{"greater", AH_CMP_GT}, C{"greater_equal", AH_CMP_GE}}) {
    auto fn = std::make_shared<ScalarFunction>(c.name, Arity{2, false});
    if (c.op == AH_CMP_EQ || c.op == AH_CMP_NE) {
      // Boolean × Boolean, the first kernels of CompareKernels(CmpEQ / CmpNE) (kernels/scalar_comparisons.go:612-666: boolEQ / boolNE
      // over the data bitmaps, validity by NullIntersection): equal = XNOR, not_equal = XOR of the data bits
      exec::ScalarKernel k;
      k.sig.in_types = {Type::BOOL, Type::BOOL};
      k.sig.out_is_first_input = false;
      k.sig.out_type = Type::BOOL;
      const int bitop = c.op == AH_CMP_EQ ? AH_BIT_XNOR : AH_BIT_XOR;
      k.exec_fn = [bitop](KernelCtx* kc, const ExecSpan& b, ExecResult* o) { return ExecBoolBinary(kc, b, o, bitop); };
      fn->AddKernel(std::move(k));
    }
    for (Type t : kNumericTypes) {
      exec::ScalarKernel k;
      k.sig.in_types = {t, t};
      k.sig.out_is_first_input = false;
      k.sig.out_type = Type::BOOL;
      int op = c.op;
      k.exec_fn = [op](KernelCtx* kc, const ExecSpan& b, ExecResult* o) { return ExecCompare(kc, b, o, op); };
      fn->AddKernel(std::move(k));
    }
    fn->promote_numeric = true;  // compareFunction.DispatchBest (scalar_compare.go:37-63)
    reg->AddFunction(fn, false);
  }
  // scalar_compare.go:73-99: less / less_equal = flipped greater / greater_equal
  auto less = std::make_shared<ScalarFunction>("less", Arity{2, false});
  less->flipped_of = "greater";
  reg->AddFunction(less, false);
  auto le = std::make_shared<ScalarFunction>("less_equal", Arity{2, false});
  le->flipped_of = "greater_equal";
  reg->AddFunction(le, false);
  // is_null / is_not_null / is_nan (scalar_compare.go:138-160; kernels/scalar_comparisons.go:718-813)
  const std::vector<Type> any_types = {Type::BOOL, Type::UINT8, Type::INT8, Type::UINT16, Type::INT16, Type::UINT32, Type::INT32, Type::UINT64,
                                       Type::INT64, Type::FLOAT32, Type::FLOAT64, Type::STRING, Type::BINARY, Type::LARGE_STRING, Type::LARGE_BINARY};
  auto validity_fn = [&](const char* name, bool want_null) {
    auto fn = std::make_shared<ScalarFunction>(name, Arity{1, false});
    for (Type t : any_types) {
      exec::ScalarKernel k;
      k.sig.in_types = {t};
      k.sig.out_is_first_input = false;
      k.sig.out_type = Type::BOOL;
      k.null_handling = exec::NullHandling::NullComputedNoPrealloc;  // :753-758: the result is never null
      k.mem_alloc = exec::MemAlloc::MemNoPrealloc;
      k.exec_fn = [want_null](KernelCtx* c, const ExecSpan& b, ExecResult* out) -> Status {
        // isNullExec: the inverted validity, or zeros when there is none (:718-730).  isNotNullExec: the validity itself
        // (:732-745) — the reference shares the buffer WITHOUT carrying the input offset, which mis-reads sliced arrays;
        // here the bits are copied from the offset (equal wherever the reference is right)
        Session* s = c->session;
        const ArraySpan& in = b.values[0].array;
        BufferPtr bits;
        AHC_RETURN_NOT_OK(c->AllocateBitmap(in.len, &bits));
        out->nulls = 0;
        out->buffers[1].WrapBuffer(bits);
        if (in.len == 0) return Status::OK();
        AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), bits->dptr, 0, (size_t)((in.len + 7) / 8))));
        if (in.buffers[0].buf)
          return s->FromStatus(ah_copy_bitmap(s->ctx(), in.buffers[0].buf, in.offset, in.len, (uint8_t*)bits->dptr, 0, want_null ? 1 : 0));
        return want_null ? Status::OK() : s->FromStatus(ah_set_bits_to(s->ctx(), (uint8_t*)bits->dptr, 0, in.len, 1));
      };
      fn->AddKernel(std::move(k));
    }
    reg->AddFunction(fn, false);
  };
  validity_fn("is_null", true);
  validity_fn("is_not_null", false);
  {
    auto fn = std::make_shared<ScalarFunction>("is_nan", Arity{1, false});
    for (Type t : kNumericTypes) {
      exec::ScalarKernel k;
      k.sig.in_types = {t};
      k.sig.out_is_first_input = false;
      k.sig.out_type = Type::BOOL;
      k.null_handling = exec::NullHandling::NullNoOutput;  // :788,792,800
      k.exec_fn = [](KernelCtx* c, const ExecSpan& b, ExecResult* out) -> Status {
        Session* s = c->session;
        if (out->len == 0) return Status::OK();
        const ArraySpan& in = b.values[0].array;
        uint8_t* ob = out->buffers[1].buf + out->offset / 8;
        if (IsFloating(in.type->id))  // isNanKernelExec: x != x over every slot (:770-780)
          return s->FromStatus(ah_comparison(s->ctx(), AH_CMP_NE, AH_SHAPE_AA, (int)in.type->id, Values(in), Values(in), ob, out->len, (int)(out->offset % 8)));
        return s->FromStatus(ah_set_bits_to(s->ctx(), out->buffers[1].buf, out->offset, out->len, 0));  // ConstBoolExec(false) :763-768
      };
      fn->AddKernel(std::move(k));
    }
    reg->AddFunction(fn, false);
  }

}

// ---- boolean --------------------------------------------------------------------------------
// a boolean scalar operand is materialised as a constant bitmap (SetBitsTo) so every shape
// runs through the same bitmap kernels — scalar_boolean.go:75-88 does CopyBitmap/SetBitsTo too
static Status BoolOperand(KernelCtx* k, const exec::ExecValue& v, int64_t len, const uint8_t** data, int64_t* off, BufferPtr* keep) {
  if (v.IsArray()) { *data = v.array.buffers[1].buf; *off = v.array.offset; return Status::OK(); }
  AHC_RETURN_NOT_OK(k->AllocateBitmap(len, keep));
  if (v.scalar->value[0] & 1)
    AHC_RETURN_NOT_OK(k->session->FromStatus(ah_set_bits_to(k->session->ctx(), (uint8_t*)(*keep)->dptr, 0, len, 1)));
  *data = (const uint8_t*)(*keep)->dptr;
  *off = 0;
  return Status::OK();
}

static Status ExecBoolBinary(KernelCtx* k, const ExecSpan& b, ExecResult* out, int bitop) {
  Session* s = k->session;
  if (out->len == 0) return Status::OK();
  const uint8_t *l, *r; int64_t lo, ro; BufferPtr kl, kr;
  AHC_RETURN_NOT_OK(BoolOperand(k, b.values[0], out->len, &l, &lo, &kl));
  AHC_RETURN_NOT_OK(BoolOperand(k, b.values[1], out->len, &r, &ro, &kr));
  AHC_RETURN_NOT_OK(s->FromStatus(ah_bitmap_op(s->ctx(), bitop, l, lo, r, ro, out->buffers[1].buf, out->offset, out->len)));
  if (kl || kr) AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));  // temporaries die with this frame
  return Status::OK();
}

static Status ExecKleene(KernelCtx* k, const ExecSpan& b, ExecResult* out, int op) {
  Session* s = k->session;
  if (out->len == 0) return Status::OK();
  const uint8_t *ld, *rd, *lv = nullptr, *rv = nullptr; int64_t lo, ro; BufferPtr kl, kr, klv, krv;
  AHC_RETURN_NOT_OK(BoolOperand(k, b.values[0], out->len, &ld, &lo, &kl));
  AHC_RETURN_NOT_OK(BoolOperand(k, b.values[1], out->len, &rd, &ro, &kr));
  if (b.values[0].IsArray()) lv = b.values[0].array.buffers[0].buf;
  else if (!b.values[0].scalar->valid) { AHC_RETURN_NOT_OK(k->AllocateBitmap(out->len, &klv)); lv = (const uint8_t*)klv->dptr; }
  if (b.values[1].IsArray()) rv = b.values[1].array.buffers[0].buf;
  else if (!b.values[1].scalar->valid) { AHC_RETURN_NOT_OK(k->AllocateBitmap(out->len, &krv)); rv = (const uint8_t*)krv->dptr; }
  // validity bitmaps share the data offset of their array (ArraySpan.Offset)
  AHC_RETURN_NOT_OK(s->FromStatus(ah_kleene(s->ctx(), op, lv, ld, lo, rv, rd, ro, out->buffers[0].buf, out->buffers[1].buf, out->offset, out->len)));
  out->nulls = kUnknownNullCount;
  if (kl || kr || klv || krv) AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));
  return Status::OK();
}

void RegisterScalarBoolean(FunctionRegistry* reg) {
  struct B { const char* name; int op; };
  for (B bo : {B{"and", AH_BIT_AND}, B{"or", AH_BIT_OR}, B{"xor", AH_BIT_XOR}, B{"and_not", AH_BIT_AND_NOT}}) {
    auto fn = std::make_shared<ScalarFunction>(bo.name, Arity{2, false});
    exec::ScalarKernel k;
    k.sig.in_types = {Type::BOOL, Type::BOOL};
    int op = bo.op;
    k.exec_fn = [op](KernelCtx* kc, const ExecSpan& b, ExecResult* o) { return ExecBoolBinary(kc, b, o, op); };
    fn->AddKernel(std::move(k));
    reg->AddFunction(fn, false);
  }
  for (B bo : {B{"and_kleene", AH_KLEENE_AND}, B{"or_kleene", AH_KLEENE_OR}, B{"and_not_kleene", AH_KLEENE_AND_NOT}}) {
    auto fn = std::make_shared<ScalarFunction>(bo.name, Arity{2, false});
    exec::ScalarKernel k;
    k.sig.in_types = {Type::BOOL, Type::BOOL};
    k.null_handling = exec::NullHandling::NullComputedPrealloc;  // scalar_bool.go:101-109
    int op = bo.op;
    k.exec_fn = [op](KernelCtx* kc, const ExecSpan& b, ExecResult* o) { return ExecKleene(kc, b, o, op); };
    fn->AddKernel(std::move(k));
    reg->AddFunction(fn, false);
  }
  auto inv = std::make_shared<ScalarFunction>("invert", Arity{1, false});
  exec::ScalarKernel k;
  k.sig.in_types = {Type::BOOL};
  k.exec_fn = [](KernelCtx* kc, const ExecSpan& b, ExecResult* o) {
    if (o->len == 0) return Status::OK();
    const ArraySpan& a = b.values[0].array;  // NotExecKernel (scalar_boolean.go:334-347)
    return kc->session->FromStatus(ah_copy_bitmap(kc->session->ctx(), a.buffers[1].buf, a.offset, a.len, o->buffers[1].buf, o->offset, 1));
  };
  inv->AddKernel(std::move(k));
  reg->AddFunction(inv, false);
}

// ---- selection --------------------------------------------------------------------------------
// PrimitiveFilter (vector_selection.go:449-520)
static Status ExecFilter(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  Session* s = k->session;
  ArraySpan values = b.values[0].array, filter = b.values[1].array;
  const FilterOptions* opts = static_cast<const FilterOptions*>(k->state);
  int null_sel = opts ? (int)opts->NullSelection : DropNulls;
  AHC_RETURN_NOT_OK(values.UpdateNullCount(s));
  AHC_RETURN_NOT_OK(filter.UpdateNullCount(s));
  const uint8_t* fvalid = filter.MayHaveNulls() ? filter.buffers[0].buf : nullptr;
  if (opts && opts->WorstCaseOutput && values.len > 0 && values.type->bit_width >= 8) {
    // ONE call (ah_filter_primitive_once): the output is sized for the input's length, the kernel counts and fills back to back and the
    // host hears the selection count once, at the end — the reference's count → allocate → fill costs a host language two calls and a
    // turnaround between two launches (0.318 against ≈ 0.29 ms for 2^27 rows at s = 0.5)
    const int w = values.type->bit_width / 8;
    out->nulls = (values.nulls == 0 && (null_sel == DropNulls || filter.nulls == 0)) ? 0 : kUnknownNullCount;
    const bool allocate_validity = values.nulls != 0 || filter.nulls != 0;
    BufferPtr vb, db;
    if (allocate_validity) { AHC_RETURN_NOT_OK(k->AllocateBitmap(values.len, &vb)); out->buffers[0].WrapBuffer(vb); }
    AHC_RETURN_NOT_OK(k->Allocate(values.len * w, &db, /*zero_all=*/false));
    out->buffers[1].WrapBuffer(db);
    int64_t n_sel = 0, nulls = 0;
    const uint8_t* vvalid = values.MayHaveNulls() ? values.buffers[0].buf : nullptr;
    AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_primitive_once(s->ctx(), w, Values(values), vvalid, values.offset, filter.buffers[1].buf, fvalid, filter.offset,
                                                             values.len, null_sel, db->dptr, allocate_validity ? (uint8_t*)vb->dptr : nullptr, &n_sel, &nulls)));
    out->len = n_sel;
    if (allocate_validity) out->nulls = nulls;
    return Status::OK();
  }
  int64_t n_out = 0;
  if (values.len > 0)  // getFilterOutputSize :57-81
    AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_count(s->ctx(), filter.buffers[1].buf, fvalid, filter.offset, filter.len, null_sel, &n_out)));
  // :464-468
  out->nulls = (values.nulls == 0 && (null_sel == DropNulls || filter.nulls == 0)) ? 0 : kUnknownNullCount;
  bool allocate_validity = values.nulls != 0 || filter.nulls != 0;  // :486-488
  int w = values.type->bit_width / 8;
  if (values.type->bit_width == 1)
    return Status::Make(StatusCode::NotImplemented, "boolean-valued filter is outside the accelerated path");
  out->len = n_out;  // preallocateData :83-93
  BufferPtr vb, db;
  if (allocate_validity) { AHC_RETURN_NOT_OK(k->AllocateBitmap(n_out, &vb)); out->buffers[0].WrapBuffer(vb); }
  AHC_RETURN_NOT_OK(k->Allocate(n_out * w, &db, /*zero_all=*/false));
  out->buffers[1].WrapBuffer(db);
  if (values.len == 0) return Status::OK();
  int64_t nulls = 0;
  const uint8_t* vvalid = values.MayHaveNulls() ? values.buffers[0].buf : nullptr;
  AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_primitive(s->ctx(), w, Values(values), vvalid, values.offset, filter.buffers[1].buf, fvalid,
                                                      filter.offset, values.len, null_sel, n_out, db->dptr,
                                                      allocate_validity ? (uint8_t*)vb->dptr : nullptr, allocate_validity ? &nulls : nullptr)));
  if (allocate_validity) out->nulls = nulls;
  return Status::OK();
}

// PrimitiveTake (vector_selection.go:1162-1192)
static Status ExecTake(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  Session* s = k->session;
  ArraySpan values = b.values[0].array, indices = b.values[1].array;
  const TakeOptions* opts = static_cast<const TakeOptions*>(k->state);
  AHC_RETURN_NOT_OK(values.UpdateNullCount(s));
  AHC_RETURN_NOT_OK(indices.UpdateNullCount(s));
  if (values.type->bit_width == 1)
    return Status::Make(StatusCode::NotImplemented, "boolean-valued take is outside the accelerated path");
  int w = values.type->bit_width / 8, iw = indices.type->bit_width / 8;
  bool allocate_validity = values.nulls != 0 || indices.nulls != 0;  // :1176
  out->len = indices.len;
  BufferPtr vb, db;
  if (allocate_validity) { AHC_RETURN_NOT_OK(k->AllocateBitmap(indices.len, &vb)); out->buffers[0].WrapBuffer(vb); }
  AHC_RETURN_NOT_OK(k->Allocate(indices.len * w, &db, /*zero_all=*/false));
  out->buffers[1].WrapBuffer(db);
  if (indices.len == 0) { out->nulls = 0; return Status::OK(); }
  int64_t nulls = 0, bad = 0;
  AHC_RETURN_NOT_OK(s->FromStatus(ah_take_primitive(
      s->ctx(), w, values.buffers[1].buf ? Values(values) : nullptr, values.MayHaveNulls() ? values.buffers[0].buf : nullptr, values.offset,
      values.len, iw, IsSignedInteger(indices.type->id) ? 1 : 0, Values(indices), indices.MayHaveNulls() ? indices.buffers[0].buf : nullptr,
      indices.offset, indices.len, opts ? (int)opts->BoundsCheck : 1, db->dptr, allocate_validity ? (uint8_t*)vb->dptr : nullptr, &nulls, &bad)));
  out->nulls = allocate_validity ? nulls : 0;
  return Status::OK();
}

// FSBImpl (vector_selection.go:1997-2031; registered for FIXED_SIZE_BINARY, DECIMAL128 and DECIMAL256 at :2344-2346 / :2354-2356):
// fixed-width values of any byte width are one slot per row.  The device take moves slots of 1, 2, 4, 8, 16 or 32 bytes (the two
// decimals, UUID-sized and hash-sized binaries) with one access each and any other width byte by byte (the reference's own test
// column is binary(3)); beyond 4096 bytes a slot is refused.
static Status CheckSlotWidth(const ArraySpan& values) {
  const int w = values.type->bit_width / 8;
  if (w >= 1 && w <= 4096) return Status::OK();
  return Status::Make(StatusCode::NotImplemented, "selection of fixed-size binary values: byte widths 1 … 4096, not " + std::to_string(w));
}
static Status ExecTakeFixed(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  AHC_RETURN_NOT_OK(CheckSlotWidth(b.values[0].array));
  return ExecTake(k, b, out);
}
// the filter of wide slots goes through GetTakeIndices (vector_selection.go:102-236) and the take, like the binary and boolean
// filters here: ah_filter_primitive's compaction is built for slots of at most 8 bytes
static Status ExecFilterFixed(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  Session* s = k->session;
  ArraySpan values = b.values[0].array, filter = b.values[1].array;
  AHC_RETURN_NOT_OK(CheckSlotWidth(values));
  const int w = values.type->bit_width / 8;
  if (w == 1 || w == 2 || w == 4 || w == 8) return ExecFilter(k, b, out);
  const FilterOptions* opts = static_cast<const FilterOptions*>(k->state);
  int null_sel = opts ? (int)opts->NullSelection : DropNulls;
  AHC_RETURN_NOT_OK(values.UpdateNullCount(s));
  AHC_RETURN_NOT_OK(filter.UpdateNullCount(s));
  if (values.len >= ((int64_t)1 << 32)) return Status::Make(StatusCode::NotImplemented, "filter of a fixed-size binary column with 2^32 rows or more");
  const uint8_t* fvalid = filter.MayHaveNulls() ? filter.buffers[0].buf : nullptr;
  int64_t n_out = 0;
  if (values.len > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_count(s->ctx(), filter.buffers[1].buf, fvalid, filter.offset, filter.len, null_sel, &n_out)));
  const bool allocate_validity = values.nulls != 0 || filter.nulls != 0;
  out->len = n_out;
  BufferPtr ib, ivb, vb, db;
  AHC_RETURN_NOT_OK(k->Allocate(n_out * 4, &ib));
  AHC_RETURN_NOT_OK(k->AllocateBitmap(n_out, &ivb));
  if (allocate_validity) { AHC_RETURN_NOT_OK(k->AllocateBitmap(n_out, &vb)); out->buffers[0].WrapBuffer(vb); }
  AHC_RETURN_NOT_OK(k->Allocate(n_out * w, &db, /*zero_all=*/false));
  out->buffers[1].WrapBuffer(db);
  out->nulls = 0;
  if (n_out == 0) return Status::OK();
  int64_t idx_nulls = 0, nulls = 0, bad = 0;
  AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_to_indices(s->ctx(), filter.buffers[1].buf, fvalid, filter.offset, values.len, null_sel, n_out,
                                                       (uint32_t*)ib->dptr, (uint8_t*)ivb->dptr, &idx_nulls)));
  AHC_RETURN_NOT_OK(s->FromStatus(ah_take_primitive(s->ctx(), w, Values(values), values.MayHaveNulls() ? values.buffers[0].buf : nullptr, values.offset, values.len,
                                                    4, 0, ib->dptr, idx_nulls ? (const uint8_t*)ivb->dptr : nullptr, 0, n_out, /*bounds_check=*/0, db->dptr,
                                                    allocate_validity ? (uint8_t*)vb->dptr : nullptr, &nulls, &bad)));
  out->nulls = allocate_validity ? nulls : 0;
  return Status::OK();
}

// dictionaryTake / dictionaryFilter (compute/selection.go:497-586): select the INDICES (int32 on the device), keep a pointer
// to the same dictionary
static void CarryDictionary(const ArraySpan& in, ExecResult* out) {
  out->dictionary = in.dictionary;
  out->dict_value_type = in.dict_value_type;
  out->dict_index_type = in.dict_index_type;
}
static Status ExecTakeDictionary(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  AHC_RETURN_NOT_OK(ExecTake(k, b, out));
  CarryDictionary(b.values[0].array, out);
  return Status::OK();
}
static Status ExecFilterDictionary(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  AHC_RETURN_NOT_OK(ExecFilter(k, b, out));
  CarryDictionary(b.values[0].array, out);
  return Status::OK();
}

// VarBinaryImpl (vector_selection.go:1925-1992) under takeExec: offsets + validity + byte count, then the bytes
static Status TakeBinaryCommon(KernelCtx* k, const ArraySpan& values, int idx_width, bool idx_signed, const void* idx, const uint8_t* ivalid,
                               int64_t ioff, int64_t n, bool allocate_validity, ExecResult* out) {
  Session* s = k->session;
  const int ow = values.type->bit_width / 8;
  out->len = n;
  BufferPtr vb, ob, db;
  if (allocate_validity) { AHC_RETURN_NOT_OK(k->AllocateBitmap(n, &vb)); out->buffers[0].WrapBuffer(vb); }
  AHC_RETURN_NOT_OK(k->Allocate((n + 1) * ow, &ob));
  out->buffers[1].WrapBuffer(ob);
  int64_t nulls = 0, total = 0, bad = 0;
  AHC_RETURN_NOT_OK(s->FromStatus(ah_take_binary_offsets(s->ctx(), ow, values.buffers[1].buf, values.MayHaveNulls() ? values.buffers[0].buf : nullptr,
                                                         values.offset, values.len, idx_width, idx_signed, idx, ivalid, ioff, n, 1, ob->dptr,
                                                         allocate_validity ? (uint8_t*)vb->dptr : nullptr, &nulls, &total, &bad)));
  AHC_RETURN_NOT_OK(k->Allocate(total, &db, /*zero_all=*/false));
  out->buffers[2].WrapBuffer(db);
  if (n > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_take_binary_data(s->ctx(), ow, values.buffers[1].buf, values.buffers[2].buf, values.offset, idx_width, idx, n,
                                                        ob->dptr, (uint8_t*)db->dptr)));
  out->nulls = allocate_validity ? nulls : 0;
  return Status::OK();
}

static Status ExecTakeBinary(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  ArraySpan values = b.values[0].array, indices = b.values[1].array;
  AHC_RETURN_NOT_OK(values.UpdateNullCount(k->session));
  AHC_RETURN_NOT_OK(indices.UpdateNullCount(k->session));
  return TakeBinaryCommon(k, values, indices.type->bit_width / 8, IsSignedInteger(indices.type->id), Values(indices),
                          indices.MayHaveNulls() ? indices.buffers[0].buf : nullptr, indices.offset, indices.len,
                          values.nulls != 0 || indices.nulls != 0, out);
}

// binaryFilterImpl (vector_selection.go:1650-1815) = GetTakeIndices (:102-236) + the var-length take
static Status ExecFilterBinary(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  Session* s = k->session;
  ArraySpan values = b.values[0].array, filter = b.values[1].array;
  const FilterOptions* opts = static_cast<const FilterOptions*>(k->state);
  int null_sel = opts ? (int)opts->NullSelection : DropNulls;
  AHC_RETURN_NOT_OK(values.UpdateNullCount(s));
  AHC_RETURN_NOT_OK(filter.UpdateNullCount(s));
  if (values.len >= ((int64_t)1 << 32)) return Status::Make(StatusCode::NotImplemented, "filter of a binary column with 2^32 rows or more");  // :229-235
  const uint8_t* fvalid = filter.MayHaveNulls() ? filter.buffers[0].buf : nullptr;
  int64_t n_out = 0;
  if (values.len > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_count(s->ctx(), filter.buffers[1].buf, fvalid, filter.offset, filter.len, null_sel, &n_out)));
  BufferPtr ib, ivb;
  AHC_RETURN_NOT_OK(k->Allocate(n_out * 4, &ib));
  AHC_RETURN_NOT_OK(k->AllocateBitmap(n_out, &ivb));
  int64_t idx_nulls = 0;
  if (n_out > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_to_indices(s->ctx(), filter.buffers[1].buf, fvalid, filter.offset, values.len, null_sel, n_out,
                                                         (uint32_t*)ib->dptr, (uint8_t*)ivb->dptr, &idx_nulls)));
  bool allocate_validity = values.nulls != 0 || filter.nulls != 0;
  return TakeBinaryCommon(k, values, 4, false, ib->dptr, idx_nulls ? (const uint8_t*)ivb->dptr : nullptr, 0, n_out, allocate_validity, out);
}

// booleanTakeImpl (vector_selection.go:990-1074) and, for filter, GetTakeIndices in front of it
static Status TakeBooleanCommon(KernelCtx* k, const ArraySpan& values, int idx_width, bool idx_signed, const void* idx, const uint8_t* ivalid,
                                int64_t ioff, int64_t n, bool allocate_validity, ExecResult* out) {
  Session* s = k->session;
  out->len = n;
  BufferPtr vb, db;
  if (allocate_validity) { AHC_RETURN_NOT_OK(k->AllocateBitmap(n, &vb)); out->buffers[0].WrapBuffer(vb); }
  AHC_RETURN_NOT_OK(k->AllocateBitmap(n, &db));
  out->buffers[1].WrapBuffer(db);
  int64_t nulls = 0, bad = 0;
  AHC_RETURN_NOT_OK(s->FromStatus(ah_take_boolean(s->ctx(), values.buffers[1].buf, values.MayHaveNulls() ? values.buffers[0].buf : nullptr, values.offset,
                                                  values.len, idx_width, idx_signed, idx, ivalid, ioff, n, 1, (uint8_t*)db->dptr,
                                                  allocate_validity ? (uint8_t*)vb->dptr : nullptr, &nulls, &bad)));
  out->nulls = allocate_validity ? nulls : 0;
  return Status::OK();
}

static Status ExecTakeBoolean(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  ArraySpan values = b.values[0].array, indices = b.values[1].array;
  AHC_RETURN_NOT_OK(values.UpdateNullCount(k->session));
  AHC_RETURN_NOT_OK(indices.UpdateNullCount(k->session));
  return TakeBooleanCommon(k, values, indices.type->bit_width / 8, IsSignedInteger(indices.type->id), Values(indices),
                           indices.MayHaveNulls() ? indices.buffers[0].buf : nullptr, indices.offset, indices.len,
                           values.nulls != 0 || indices.nulls != 0, out);
}

static Status ExecFilterBoolean(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  Session* s = k->session;
  ArraySpan values = b.values[0].array, filter = b.values[1].array;
  const FilterOptions* opts = static_cast<const FilterOptions*>(k->state);
  int null_sel = opts ? (int)opts->NullSelection : DropNulls;
  AHC_RETURN_NOT_OK(values.UpdateNullCount(s));
  AHC_RETURN_NOT_OK(filter.UpdateNullCount(s));
  if (values.len >= ((int64_t)1 << 32)) return Status::Make(StatusCode::NotImplemented, "filter of a boolean column with 2^32 rows or more");
  const uint8_t* fvalid = filter.MayHaveNulls() ? filter.buffers[0].buf : nullptr;
  int64_t n_out = 0;
  if (values.len > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_count(s->ctx(), filter.buffers[1].buf, fvalid, filter.offset, filter.len, null_sel, &n_out)));
  BufferPtr ib, ivb;
  AHC_RETURN_NOT_OK(k->Allocate(n_out * 4, &ib));
  AHC_RETURN_NOT_OK(k->AllocateBitmap(n_out, &ivb));
  int64_t idx_nulls = 0;
  if (n_out > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_to_indices(s->ctx(), filter.buffers[1].buf, fvalid, filter.offset, values.len, null_sel, n_out,
                                                         (uint32_t*)ib->dptr, (uint8_t*)ivb->dptr, &idx_nulls)));
  return TakeBooleanCommon(k, values, 4, false, ib->dptr, idx_nulls ? (const uint8_t*)ivb->dptr : nullptr, 0, n_out,
                           values.nulls != 0 || filter.nulls != 0, out);
}

static const Type kBinaryTypes[] = {Type::STRING, Type::BINARY, Type::LARGE_STRING, Type::LARGE_BINARY};

static const FilterOptions kDefaultFilterOptions;
static const TakeOptions kDefaultTakeOptions;
static const DictionaryEncodeOptions kDefaultDictOptions;

// FilterRecordBatch (compute/selection.go:679-722): the mask becomes ONE index vector (GetTakeIndices,
// kernels/vector_selection.go:102-236) and every column is gathered with it, bounds check off
static Status FilterRecordBatch(ExecCtx* ctx, const Datum& batch, const ArrayData& filter_in, const FilterOptions* opts, Datum* out) {
  Session* s = ctx->session;
  if (batch.num_rows != filter_in.length) return Status::Make(StatusCode::Invalid, "filter inputs must all be the same length");
  if (filter_in.length >= ((int64_t)1 << 32) - 1)
    return Status::Make(StatusCode::NotImplemented, "filter length exceeds UINT32_MAX, consider a different strategy for selecting elements");  // :229-235
  exec::ArraySpan filter;
  filter.SetMembers(filter_in);
  AHC_RETURN_NOT_OK(filter.UpdateNullCount(s));
  const int null_sel = opts ? (int)opts->NullSelection : DropNulls;
  const uint8_t* fvalid = filter.MayHaveNulls() ? filter.buffers[0].buf : nullptr;
  int64_t n_out = 0;
  if (filter.len > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_count(s->ctx(), filter.buffers[1].buf, fvalid, filter.offset, filter.len, null_sel, &n_out)));
  auto idx = std::make_shared<ArrayData>();
  idx->type = GetDataType(Type::UINT32);
  idx->length = n_out;
  AHC_RETURN_NOT_OK(s->Allocate(n_out * 4, &idx->buffers[1]));
  BufferPtr ivb;
  AHC_RETURN_NOT_OK(s->AllocateBitmap(n_out, &ivb));
  int64_t idx_nulls = 0;
  if (n_out > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_to_indices(s->ctx(), filter.buffers[1].buf, fvalid, filter.offset, filter.len, null_sel, n_out,
                                                         (uint32_t*)idx->buffers[1]->dptr, (uint8_t*)ivb->dptr, &idx_nulls)));
  idx->null_count = idx_nulls;
  if (idx_nulls) idx->buffers[0] = ivb;
  static const TakeOptions kNoBoundsCheck = [] { TakeOptions t; t.BoundsCheck = false; return t; }();
  std::vector<ArrayDataPtr> cols;
  for (auto& c : batch.chunks) {
    Datum col;
    AHC_RETURN_NOT_OK(CallFunction(ctx, "array_take", &kNoBoundsCheck, {Datum::Of(c), Datum::Of(idx)}, &col));
    cols.push_back(col.array);
  }
  *out = Datum::OfRecord(batch.names, std::move(cols), n_out);
  return Status::OK();
}

void RegisterVectorSelection(FunctionRegistry* reg) {
  auto af = std::make_shared<VectorFunction>("array_filter", Arity{2, false}, &kDefaultFilterOptions);
  af->chunked = VectorFunction::Chunked::Filter;
  for (Type t : kNumericTypes) {
    exec::VectorKernel k;
    k.sig.in_types = {t, Type::BOOL};
    k.exec_fn = ExecFilter;
    af->AddKernel(std::move(k));
  }
  {
    exec::VectorKernel k;
    k.sig.in_types = {Type::DICTIONARY, Type::BOOL};
    k.exec_fn = ExecFilterDictionary;
    af->AddKernel(std::move(k));
  }
  for (Type t : kBinaryTypes) {
    exec::VectorKernel k;
    k.sig.in_types = {t, Type::BOOL};
    k.exec_fn = ExecFilterBinary;
    af->AddKernel(std::move(k));
  }
  {
    exec::VectorKernel k;
    k.sig.in_types = {Type::BOOL, Type::BOOL};
    k.exec_fn = ExecFilterBoolean;
    af->AddKernel(std::move(k));
  }
  for (Type t : {Type::FIXED_SIZE_BINARY, Type::DECIMAL128, Type::DECIMAL256}) {   // vector_selection.go:2344-2346
    exec::VectorKernel k;
    k.sig.in_types = {t, Type::BOOL};
    k.exec_fn = ExecFilterFixed;
    af->AddKernel(std::move(k));
  }
  reg->AddFunction(af, false);
  auto at = std::make_shared<VectorFunction>("array_take", Arity{2, false}, &kDefaultTakeOptions);
  at->chunked = VectorFunction::Chunked::Take;
  for (Type t : kNumericTypes)
    for (Type it : {Type::INT8, Type::UINT8, Type::INT16, Type::UINT16, Type::INT32, Type::UINT32, Type::INT64, Type::UINT64}) {
      exec::VectorKernel k;
      k.sig.in_types = {t, it};
      k.exec_fn = ExecTake;
      k.can_execute_chunkwise = false;  // selection.go:639
      at->AddKernel(std::move(k));
    }
  for (Type it : {Type::INT8, Type::UINT8, Type::INT16, Type::UINT16, Type::INT32, Type::UINT32, Type::INT64, Type::UINT64}) {
    exec::VectorKernel k;
    k.sig.in_types = {Type::DICTIONARY, it};
    k.exec_fn = ExecTakeDictionary;
    k.can_execute_chunkwise = false;
    at->AddKernel(std::move(k));
  }
  for (Type it : {Type::INT8, Type::UINT8, Type::INT16, Type::UINT16, Type::INT32, Type::UINT32, Type::INT64, Type::UINT64}) {
    exec::VectorKernel k;
    k.sig.in_types = {Type::BOOL, it};
    k.exec_fn = ExecTakeBoolean;
    k.can_execute_chunkwise = false;
    at->AddKernel(std::move(k));
  }
  for (Type t : {Type::FIXED_SIZE_BINARY, Type::DECIMAL128, Type::DECIMAL256})   // vector_selection.go:2354-2356
    for (Type it : {Type::INT8, Type::UINT8, Type::INT16, Type::UINT16, Type::INT32, Type::UINT32, Type::INT64, Type::UINT64}) {
      exec::VectorKernel k;
      k.sig.in_types = {t, it};
      k.exec_fn = ExecTakeFixed;
      k.can_execute_chunkwise = false;
      at->AddKernel(std::move(k));
    }
  for (Type t : kBinaryTypes)
    for (Type it : {Type::INT8, Type::UINT8, Type::INT16, Type::UINT16, Type::INT32, Type::UINT32, Type::INT64, Type::UINT64}) {
      exec::VectorKernel k;
      k.sig.in_types = {t, it};
      k.exec_fn = ExecTakeBinary;
      k.can_execute_chunkwise = false;
      at->AddKernel(std::move(k));
    }
  reg->AddFunction(at, false);
  // filterMetaFunc (selection.go:42-85): validates, then dispatches on the values kind
  reg->AddFunction(std::make_shared<MetaFunction>("filter", Arity{2, false}, &kDefaultFilterOptions,
      [](ExecCtx* ctx, const FunctionOptions* o, const std::vector<Datum>& args, Datum* out) {
        if (!args[1].type() || args[1].type()->id != Type::BOOL)
          return Status::Make(StatusCode::NotImplemented, "filter argument must be boolean type");  // :45-48
        if (args[0].kind == DatumKind::Record) {
          if (args[1].kind != DatumKind::Array)
            return Status::Make(StatusCode::NotImplemented, "record batch filtering only implemented for Array filter");  // :69
          return FilterRecordBatch(ctx, args[0], *args[1].array, static_cast<const FilterOptions*>(o), out);
        }
        if (args[0].kind == DatumKind::Array && args[1].kind == DatumKind::Array && args[0].Len() != args[1].Len())
          return Status::Make(StatusCode::Invalid, "filter inputs must all be the same length");  // FilterArray check
        return CallFunction(ctx, "array_filter", o, args, out);
      }), false);
  // takeMetaFunc (selection.go:93-114)
  reg->AddFunction(std::make_shared<MetaFunction>("take", Arity{2, false}, &kDefaultTakeOptions,
      [](ExecCtx* ctx, const FunctionOptions* o, const std::vector<Datum>& args, Datum* out) {
        if (!args[1].type() || !IsInteger(args[1].type()->id))
          return Status::Make(StatusCode::NotImplemented, "take indices must be an integer type");
        if (args[0].kind == DatumKind::Record) {  // takeRecordImpl (selection.go:160-204): every column through array_take with the same indices
          Datum indices = args[1];
          if (indices.kind == DatumKind::Chunked) {
            if (indices.chunks.empty()) return Status::Make(StatusCode::Invalid, "Must pass at least one array");  // concat.go:43-45
            ArrayDataPtr whole;
            AHC_RETURN_NOT_OK(Concatenate(ctx->session, indices.chunks, indices.chunked_type, &whole));
            indices = Datum::Of(whole);
          }
          std::vector<ArrayDataPtr> cols;
          for (auto& c : args[0].chunks) {
            Datum col;
            AHC_RETURN_NOT_OK(CallFunction(ctx, "array_take", o, {Datum::Of(c), indices}, &col));
            cols.push_back(col.array);
          }
          *out = Datum::OfRecord(args[0].names, std::move(cols), indices.Len());
          return Status::OK();
        }
        return CallFunction(ctx, "array_take", o, args, out);
      }), false);
}

// ---- hashing -----------------------------------------------------------------------------------
// regularHashState over Table[uint64] (vector_hash.go:243-286,359-385,604-607)
static Status ExecHash(KernelCtx* k, const ExecSpan& b, ExecResult* out, bool dict_encode) {
  Session* s = k->session;
  ArraySpan keys = b.values[0].array;
  // 1/2/4-byte keys: the reference memoises them in Table[uint8/16/32] keyed on the raw bits
  // (vector_hash.go:596-607); here the bits are zero-extended to 64 on the device, hashed in the same
  // table, and the dictionary is narrowed back — the ids and the first-seen order are unchanged
  // Boolean keys (doAppendBoolean, vector_hash.go:387-415: the bit becomes the uint8 key 0 / 1 of the same memo table): the bits
  // are widened on the device like the narrow integers, and the ≤ 3 dictionary entries go back to a bitmap through `!= 0`
  const bool is_bool = keys.type->id == Type::BOOL;
  const int kw = is_bool ? 1 : keys.type->bit_width / 8;
  const int raw_type = kw == 1 ? AH_UINT8 : kw == 2 ? AH_UINT16 : AH_UINT32;
  const DictionaryEncodeOptions* opts = static_cast<const DictionaryEncodeOptions*>(k->state);
  int encode_nulls = dict_encode ? (opts && opts->NullEncoding == NullEncodingEncode) : 1;  // uniqueAction.ShouldEncodeNulls() == true
  AHC_RETURN_NOT_OK(keys.UpdateNullCount(s));
  const uint8_t* valid = keys.MayHaveNulls() ? keys.buffers[0].buf : nullptr;
  int64_t n = keys.len;
  BufferPtr ids, ids_valid, dict;
  AHC_RETURN_NOT_OK(k->Allocate((n + 1) * 8, &dict));
  if (dict_encode) {
    AHC_RETURN_NOT_OK(k->Allocate(n * 4, &ids, /*zero_all=*/false));
    if (valid && !encode_nulls) AHC_RETURN_NOT_OK(k->AllocateBitmap(n, &ids_valid));
  }
  int64_t ndict = 0; int32_t null_id = -1;
  const uint64_t* keys64 = is_bool ? nullptr : (const uint64_t*)Values(keys);
  BufferPtr widened;
  if (kw < 8 && n > 0) {
    AHC_RETURN_NOT_OK(k->Allocate(n * 8, &widened, /*zero_all=*/false));
    if (is_bool) AHC_RETURN_NOT_OK(s->FromStatus(ah_cast_bool_to_numeric(s->ctx(), AH_UINT64, keys.buffers[1].buf, keys.offset, n, widened->dptr)));
    else AHC_RETURN_NOT_OK(s->FromStatus(ah_cast_numeric(s->ctx(), raw_type, AH_UINT64, Values(keys), nullptr, 0, n, 1, 1, widened->dptr)));
    keys64 = (const uint64_t*)widened->dptr;
  }
  if (n > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_hash_u64_encode(s->ctx(), keys64, valid, keys.offset, n, encode_nulls,
                                                       ids ? (int32_t*)ids->dptr : nullptr, ids_valid ? (uint8_t*)ids_valid->dptr : nullptr,
                                                       (uint64_t*)dict->dptr, &ndict, &null_id)));
  // GetDictArrayData (arrow/array/util.go:321-390): values by memo index; if the table holds
  // a null, validity = all ones with that bit cleared
  auto d = std::make_shared<ArrayData>();
  d->type = keys.type;
  d->length = ndict;
  d->null_count = null_id >= 0 ? 1 : 0;
  if (is_bool) {
    BufferPtr bits;
    AHC_RETURN_NOT_OK(k->AllocateBitmap(ndict, &bits));
    alignas(8) uint8_t zero[8] = {0};
    if (ndict > 0) AHC_RETURN_NOT_OK(s->FromStatus(ah_comparison(s->ctx(), AH_CMP_NE, AH_SHAPE_AS, AH_UINT64, dict->dptr, zero, (uint8_t*)bits->dptr, ndict, 0)));
    dict = bits;
  } else if (kw < 8) {
    BufferPtr narrow;
    AHC_RETURN_NOT_OK(k->Allocate(ndict * kw, &narrow, /*zero_all=*/false));
    if (ndict > 0)
      AHC_RETURN_NOT_OK(s->FromStatus(ah_cast_numeric(s->ctx(), AH_UINT64, raw_type, dict->dptr, nullptr, 0, ndict, 1, 1, narrow->dptr)));
    dict = narrow;
  }
  d->buffers[1] = dict;
  dict->size = is_bool ? (ndict + 7) / 8 : ndict * kw;
  if (null_id >= 0) {
    BufferPtr dv;
    AHC_RETURN_NOT_OK(k->AllocateBitmap(ndict, &dv));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), dv->dptr, 0xFF, (size_t)((ndict + 7) / 8))));  // memory.Set(…, 0xFF), util.go:381
    AHC_RETURN_NOT_OK(s->FromStatus(ah_set_bits_to(s->ctx(), (uint8_t*)dv->dptr, null_id, 1, 0)));
    d->buffers[0] = dv;
  }
  if (!dict_encode) {  // uniqueFinalize (vector_hash.go:721-741): the result IS the dictionary
    out->type = d->type;
    out->len = d->length;
    out->nulls = d->null_count;
    out->buffers[0].WrapBuffer(d->buffers[0]);
    out->buffers[1].WrapBuffer(d->buffers[1]);
    return Status::OK();
  }
  out->len = n;
  out->nulls = ids_valid ? keys.nulls : 0;  // dictionaryEncodeAction.Flush :188-209
  out->buffers[0].WrapBuffer(ids_valid);
  out->buffers[1].WrapBuffer(ids);
  out->dictionary = d;
  return Status::OK();
}

// regularHashState over BinaryMemoTable (vector_hash.go:288-325,612-618): ids on the device, the dictionary
// = the var-length take of each entry's first row (the memo table's builder holds exactly those bytes)
static Status ExecHashBinary(KernelCtx* k, const ExecSpan& b, ExecResult* out, bool dict_encode) {
  Session* s = k->session;
  ArraySpan keys = b.values[0].array;
  const DictionaryEncodeOptions* opts = static_cast<const DictionaryEncodeOptions*>(k->state);
  int encode_nulls = dict_encode ? (opts && opts->NullEncoding == NullEncodingEncode) : 1;
  AHC_RETURN_NOT_OK(keys.UpdateNullCount(s));
  const uint8_t* valid = keys.MayHaveNulls() ? keys.buffers[0].buf : nullptr;
  const int64_t n = keys.len;
  const int ow = keys.type->bit_width / 8;
  BufferPtr ids, ids_valid, first_rows;
  AHC_RETURN_NOT_OK(k->Allocate((n + 1) * 8, &first_rows));
  if (dict_encode) {
    AHC_RETURN_NOT_OK(k->Allocate(n * 4, &ids, /*zero_all=*/false));
    if (valid && !encode_nulls) AHC_RETURN_NOT_OK(k->AllocateBitmap(n, &ids_valid));
  }
  int64_t ndict = 0; int32_t null_id = -1;
  if (n > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_hash_binary_encode(s->ctx(), ow, keys.buffers[1].buf, keys.buffers[2].buf, valid, keys.offset, n, encode_nulls,
                                                          ids ? (int32_t*)ids->dptr : nullptr, ids_valid ? (uint8_t*)ids_valid->dptr : nullptr,
                                                          (int64_t*)first_rows->dptr, &ndict, &null_id)));
  // GetDictArrayData (arrow/array/util.go:341-366): offsets + values in memo order, a null entry has no
  // bytes; validity = all ones with the null's bit cleared (:375-384)
  ExecResult dres;
  dres.type = keys.type;
  AHC_RETURN_NOT_OK(TakeBinaryCommon(k, keys, 8, true, first_rows->dptr, nullptr, 0, ndict, false, &dres));
  auto d = std::make_shared<ArrayData>();
  d->type = keys.type;
  d->length = ndict;
  d->null_count = null_id >= 0 ? 1 : 0;
  d->buffers[1] = dres.buffers[1].owner;
  d->buffers[2] = dres.buffers[2].owner;
  if (null_id >= 0) {
    BufferPtr dv;
    AHC_RETURN_NOT_OK(k->AllocateBitmap(ndict, &dv));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), dv->dptr, 0xFF, (size_t)((ndict + 7) / 8))));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_set_bits_to(s->ctx(), (uint8_t*)dv->dptr, null_id, 1, 0)));
    d->buffers[0] = dv;
  }
  if (!dict_encode) {  // uniqueFinalize (vector_hash.go:721-741)
    out->type = d->type;
    out->len = d->length;
    out->nulls = d->null_count;
    out->buffers[0].WrapBuffer(d->buffers[0]);
    out->buffers[1].WrapBuffer(d->buffers[1]);
    out->buffers[2].WrapBuffer(d->buffers[2]);
    return Status::OK();
  }
  out->len = n;
  out->nulls = ids_valid ? keys.nulls : 0;
  out->buffers[0].WrapBuffer(ids_valid);
  out->buffers[1].WrapBuffer(ids);
  out->dictionary = d;
  return Status::OK();
}

// FixedSizeBinary / Decimal128 / Decimal256 keys (vector_hash.go:608-609, 698: the BinaryMemoTable over values of one byte width):
// ah_hash_fixed_encode — ids, index validity, null id as for the var-length keys; the dictionary comes back as ndict values
static Status ExecHashFixed(KernelCtx* k, const ExecSpan& b, ExecResult* out, bool dict_encode) {
  Session* s = k->session;
  ArraySpan keys = b.values[0].array;
  const DictionaryEncodeOptions* opts = static_cast<const DictionaryEncodeOptions*>(k->state);
  int encode_nulls = dict_encode ? (opts && opts->NullEncoding == NullEncodingEncode) : 1;
  AHC_RETURN_NOT_OK(keys.UpdateNullCount(s));
  const uint8_t* valid = keys.MayHaveNulls() ? keys.buffers[0].buf : nullptr;
  const int64_t n = keys.len;
  const int w = keys.type->bit_width / 8;
  if (w <= 0) return Status::Make(StatusCode::Invalid, "fixed-size binary keys need a byte width");
  BufferPtr ids, ids_valid, first_rows, dict;
  AHC_RETURN_NOT_OK(k->Allocate((n + 1) * 8, &first_rows));
  AHC_RETURN_NOT_OK(k->Allocate((n + 1) * w, &dict));   // at most n + 1 entries; handed on as the dictionary's value buffer
  if (dict_encode) {
    AHC_RETURN_NOT_OK(k->Allocate(n * 4, &ids, /*zero_all=*/false));
    if (valid && !encode_nulls) AHC_RETURN_NOT_OK(k->AllocateBitmap(n, &ids_valid));
  }
  int64_t ndict = 0; int32_t null_id = -1;
  if (n > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_hash_fixed_encode(s->ctx(), w, keys.buffers[1].buf, valid, keys.offset, n, encode_nulls, ids ? (int32_t*)ids->dptr : nullptr,
                                                         ids_valid ? (uint8_t*)ids_valid->dptr : nullptr, (int64_t*)first_rows->dptr, (uint8_t*)dict->dptr,
                                                         &ndict, &null_id)));
  auto d = std::make_shared<ArrayData>();
  d->type = keys.type;
  d->length = ndict;
  d->null_count = null_id >= 0 ? 1 : 0;
  d->buffers[1] = dict;
  if (null_id >= 0) {   // GetDictArrayData (arrow/array/util.go:375-384): all ones with the null entry's bit cleared
    BufferPtr dv;
    AHC_RETURN_NOT_OK(k->AllocateBitmap(ndict, &dv));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), dv->dptr, 0xFF, (size_t)((ndict + 7) / 8))));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_set_bits_to(s->ctx(), (uint8_t*)dv->dptr, null_id, 1, 0)));
    d->buffers[0] = dv;
  }
  if (!dict_encode) {  // uniqueFinalize (vector_hash.go:721-741)
    out->type = d->type;
    out->len = d->length;
    out->nulls = d->null_count;
    out->buffers[0].WrapBuffer(d->buffers[0]);
    out->buffers[1].WrapBuffer(d->buffers[1]);
    return Status::OK();
  }
  out->len = n;
  out->nulls = ids_valid ? keys.nulls : 0;
  out->buffers[0].WrapBuffer(ids_valid);
  out->buffers[1].WrapBuffer(ids);
  out->dictionary = d;
  return Status::OK();
}

void RegisterVectorHash(FunctionRegistry* reg) {
  auto uq = std::make_shared<VectorFunction>("unique", Arity{1, false});
  uq->chunked = VectorFunction::Chunked::SingleArray;
  auto de = std::make_shared<VectorFunction>("dictionary_encode", Arity{1, false}, &kDefaultDictOptions);
  std::vector<Type> fixed(std::begin(kNumericTypes), std::end(kNumericTypes));
  fixed.push_back(Type::BOOL);
  for (Type t : fixed) {
    exec::VectorKernel ku;
    ku.sig.in_types = {t};
    ku.exec_fn = [](KernelCtx* k, const ExecSpan& b, ExecResult* o) { return ExecHash(k, b, o, false); };
    uq->AddKernel(std::move(ku));
    exec::VectorKernel kd;
    kd.sig.in_types = {t};
    kd.output_is_dictionary = true;
    kd.exec_fn = [](KernelCtx* k, const ExecSpan& b, ExecResult* o) { return ExecHash(k, b, o, true); };
    de->AddKernel(std::move(kd));
  }
  for (Type t : {Type::FIXED_SIZE_BINARY, Type::DECIMAL128, Type::DECIMAL256}) {
    exec::VectorKernel ku;
    ku.sig.in_types = {t};
    ku.exec_fn = [](KernelCtx* k, const ExecSpan& b, ExecResult* o) { return ExecHashFixed(k, b, o, false); };
    uq->AddKernel(std::move(ku));
    exec::VectorKernel kd;
    kd.sig.in_types = {t};
    kd.output_is_dictionary = true;
    kd.exec_fn = [](KernelCtx* k, const ExecSpan& b, ExecResult* o) { return ExecHashFixed(k, b, o, true); };
    de->AddKernel(std::move(kd));
  }
  for (Type t : {Type::BINARY, Type::STRING, Type::LARGE_BINARY, Type::LARGE_STRING}) {
    exec::VectorKernel ku;
    ku.sig.in_types = {t};
    ku.exec_fn = [](KernelCtx* k, const ExecSpan& b, ExecResult* o) { return ExecHashBinary(k, b, o, false); };
    uq->AddKernel(std::move(ku));
    exec::VectorKernel kd;
    kd.sig.in_types = {t};
    kd.output_is_dictionary = true;
    kd.exec_fn = [](KernelCtx* k, const ExecSpan& b, ExecResult* o) { return ExecHashBinary(k, b, o, true); };
    de->AddKernel(std::move(kd));
  }
  {
    // dictionaryHashState (vector_hash.go:505-576, uniqueFinalizeDictionary :757-775): the distinct INDICES in first-seen
    // order, as a dictionary array over the same dictionary; dictionary_encode of a dictionary array is the identity (:473-478)
    exec::VectorKernel ku;
    ku.sig.in_types = {Type::DICTIONARY};
    ku.exec_fn = [](KernelCtx* k, const ExecSpan& b, ExecResult* o) {
      AHC_RETURN_NOT_OK(ExecHash(k, b, o, false));
      CarryDictionary(b.values[0].array, o);
      return Status::OK();
    };
    uq->AddKernel(std::move(ku));
    exec::VectorKernel kd;
    kd.sig.in_types = {Type::DICTIONARY};
    kd.exec_fn = [](KernelCtx*, const ExecSpan& b, ExecResult* o) {
      *o = b.values[0].array;
      return Status::OK();
    };
    de->AddKernel(std::move(kd));
  }
  reg->AddFunction(uq, false);
  reg->AddFunction(de, false);
}

// ---- cast (numeric ↔ numeric, bool ↔ numeric) -------------------------------------------------------
// CastIntToInt / CastFloatingToInteger / CastIntegerToFloating / CastFloatingToFloating
// (kernels/numeric_cast.go:37-71): conversion and safe-cast check in one pass on the device
static Status ExecCastNumeric(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  Session* s = k->session;
  if (out->len == 0) return Status::OK();
  const CastOptions* opts = static_cast<const CastOptions*>(k->state);
  const ArraySpan& in = b.values[0].array;
  return s->FromStatus(ah_cast_numeric(s->ctx(), (int)in.type->id, (int)out->type->id, Values(in), in.MayHaveNulls() ? in.buffers[0].buf : nullptr,
                                       in.offset, in.len, opts && opts->AllowIntOverflow, opts && opts->AllowFloatTruncate, Values(out)));
}
// boolToNum (numeric_cast.go:555-569)
static Status ExecCastBoolToNumeric(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  Session* s = k->session;
  if (out->len == 0) return Status::OK();
  const ArraySpan& in = b.values[0].array;
  return s->FromStatus(ah_cast_bool_to_numeric(s->ctx(), (int)out->type->id, in.buffers[1].buf, in.offset, in.len, Values(out)));
}
// isNonZero (boolean_cast.go:30-36): v != 0 — the array∘scalar "not_equal" kernel
static Status ExecCastNumericToBool(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  Session* s = k->session;
  if (out->len == 0) return Status::OK();
  const ArraySpan& in = b.values[0].array;
  alignas(8) uint8_t zero[8] = {0};
  return s->FromStatus(ah_comparison(s->ctx(), AH_CMP_NE, AH_SHAPE_AS, (int)in.type->id, Values(in), zero, out->buffers[1].buf + out->offset / 8,
                                     out->len, (int)(out->offset % 8)));
}

static const char* CastFunctionName(Type t) {
  switch (t) {
    case Type::BOOL: return "cast_boolean";
    case Type::UINT8: return "cast_uint8"; case Type::INT8: return "cast_int8"; case Type::UINT16: return "cast_uint16";
    case Type::INT16: return "cast_int16"; case Type::UINT32: return "cast_uint32"; case Type::INT32: return "cast_int32";
    case Type::UINT64: return "cast_uint64"; case Type::INT64: return "cast_int64"; case Type::FLOAT32: return "cast_float";
    case Type::FLOAT64: return "cast_double";
    default: return nullptr;
  }
}

// RegisterScalarCast (compute/cast.go:83-85) + the per-target cast functions getCastFunction resolves
// (cast.go:190-260: "cast_int32" …), numeric and boolean targets
void RegisterScalarCast(FunctionRegistry* reg) {
  for (Type to : kNumericTypes) {
    auto fn = std::make_shared<ScalarFunction>(CastFunctionName(to), Arity{1, false});
    for (Type from : kNumericTypes) {
      if (from == to) continue;
      exec::ScalarKernel k;
      k.sig.in_types = {from};
      k.sig.out_is_first_input = false;
      k.sig.out_type = to;
      k.exec_fn = ExecCastNumeric;
      fn->AddKernel(std::move(k));
    }
    exec::ScalarKernel kb;
    kb.sig.in_types = {Type::BOOL};
    kb.sig.out_is_first_input = false;
    kb.sig.out_type = to;
    kb.exec_fn = ExecCastBoolToNumeric;
    fn->AddKernel(std::move(kb));
    reg->AddFunction(fn, false);
  }
  auto fb = std::make_shared<ScalarFunction>(CastFunctionName(Type::BOOL), Arity{1, false});
  for (Type from : kNumericTypes) {
    exec::ScalarKernel k;
    k.sig.in_types = {from};
    k.sig.out_is_first_input = false;
    k.sig.out_type = Type::BOOL;
    k.exec_fn = ExecCastNumericToBool;
    fb->AddKernel(std::move(k));
  }
  reg->AddFunction(fb, false);
  // castMetaFunc (cast.go:45-81)
  reg->AddFunction(std::make_shared<MetaFunction>("cast", Arity{1, false}, nullptr,
      [](ExecCtx* ctx, const FunctionOptions* o, const std::vector<Datum>& args, Datum* out) {
        const CastOptions* opts = dynamic_cast<const CastOptions*>(o);
        if (!opts || !opts->ToType)
          return Status::Make(StatusCode::Invalid, "cast requires that options be passed with a ToType");
        if (args[0].type()->id == opts->ToType->id) { *out = args[0]; return Status::OK(); }  // TypeEqual → the input itself
        const char* name = CastFunctionName(opts->ToType->id);
        if (!name)
          return Status::Make(StatusCode::NotImplemented, std::string("unsupported cast to ") + opts->ToType->name + " from " + args[0].type()->name);
        Status st = CallFunction(ctx, name, o, args, out);
        if (!st.ok() && st.code == StatusCode::NotImplemented)
          return Status::Make(StatusCode::NotImplemented, std::string("unsupported cast to ") + opts->ToType->name + " from " + args[0].type()->name);
        return st;
      }), false);
}

// ---- is_in -------------------------------------------------------------------------------------------
// initSetLookup + execIsIn (compute/scalar_set_lookup.go:67-172): the value set is SAFE-cast to the
// input type when the types differ; the kernel writes data and validity (NullComputedPrealloc)
static Status ExecIsIn(KernelCtx* k, const ExecSpan& b, ExecResult* out) {
  Session* s = k->session;
  const SetOptions* opts = dynamic_cast<const SetOptions*>(static_cast<const FunctionOptions*>(k->state));
  if (!opts) return Status::Make(StatusCode::Invalid, "calling a set lookup function without SetOptions");
  if (!opts->ValueSet) return Status::Make(StatusCode::Invalid, "expected array-like datum, got nil");
  const ArraySpan& in = b.values[0].array;
  ArrayDataPtr vset = opts->ValueSet;
  if (vset->type->id != in.type->id) {
    ExecCtx ectx;
    ectx.session = s;
    Datum casted;
    Status st = CastDatum(&ectx, Datum::Of(vset), CastOptions::Safe(in.type), &casted);
    if (!st.ok()) {
      if (st.code == StatusCode::NotImplemented)
        return Status::Make(StatusCode::Invalid, std::string("array type doesn't match type of values set: ") + in.type->name + " vs " + vset->type->name);
      return st;
    }
    vset = casted.array;
  }
  if (out->len == 0) return Status::OK();
  int w = in.type->bit_width / 8;
  const uint8_t* set_vals = vset->length ? (const uint8_t*)vset->buffers[1]->dptr + vset->offset * w : nullptr;
  const uint8_t* set_valid = vset->buffers[0] ? (const uint8_t*)vset->buffers[0]->dptr : nullptr;
  AHC_RETURN_NOT_OK(s->FromStatus(ah_is_in(s->ctx(), w, Values(in), in.MayHaveNulls() ? in.buffers[0].buf : nullptr, in.offset, in.len, set_vals,
                                           set_valid, vset->offset, vset->length, opts->NullBehavior, out->buffers[1].buf, out->buffers[0].buf,
                                           out->offset)));
  out->nulls = kUnknownNullCount;
  return Status::OK();
}

// RegisterScalarSetLookup (compute/scalar_set_lookup.go:175-232), fixed-width numeric inputs
void RegisterScalarSetLookup(FunctionRegistry* reg) {
  auto fn = std::make_shared<ScalarFunction>("is_in", Arity{1, false});
  for (Type t : kNumericTypes) {
    exec::ScalarKernel k;
    k.sig.in_types = {t};
    k.sig.out_is_first_input = false;
    k.sig.out_type = Type::BOOL;
    k.null_handling = exec::NullHandling::NullComputedPrealloc;  // :186-187
    k.exec_fn = ExecIsIn;
    fn->AddKernel(std::move(k));
  }
  reg->AddFunction(fn, false);
}

// ---- sort_indices / sort ------------------------------------------------------------------------------
// sortIndicesMetaFunc → sortIndicesImpl (compute/vector_sort.go:42-52, 117-190) → kernels.SortIndices
// (kernels/vector_sort.go:388-481), array input: one key, ColumnIndex ignored; uint64 indices, no nulls
static Status SortIndicesImpl(ExecCtx* ctx, const FunctionOptions* o, const std::vector<Datum>& args_in, Datum* out) {
  const SortOptions* opts = dynamic_cast<const SortOptions*>(o);
  if (!opts || opts->Keys.empty()) return Status::Make(StatusCode::Invalid, "must provide at least one sort key");  // :119-121
  Session* s = ctx->session;
  // a chunked column is sorted as its logical concatenation: the indices address the whole column
  // (compute/vector_sort.go:144-148, SortIndicesChunked :226)
  std::vector<Datum> args = args_in;
  if (args.size() == 1 && args[0].kind == DatumKind::Record) {  // KindRecord (:150-166): the keys name columns of the batch
    std::vector<Datum> cols;
    for (auto& c : args_in[0].chunks) cols.push_back(Datum::Of(c));
    if (cols.empty()) return Status::Make(StatusCode::Invalid, "sort_indices: record batch without columns");
    args = cols;
  }
  for (auto& a : args) {
    if (a.kind != DatumKind::Chunked) continue;
    if (a.chunks.empty()) return Status::Make(StatusCode::NotImplemented, "sort_indices of a chunked array without chunks");
    ArrayDataPtr whole;
    AHC_RETURN_NOT_OK(Concatenate(s, a.chunks, a.chunked_type, &whole));
    a = Datum::Of(whole);
  }
  // one array: the first key, ColumnIndex ignored (:130-141); several arrays = the columns of a record batch,
  // every key names its column (:153-166)
  std::vector<SortKey> keys = opts->Keys;
  if (args.size() == 1) { keys.resize(1); keys[0].ColumnIndex = 0; }
  if (keys.size() > 8)  // maxRadixSortKeys (kernels/vector_sort.go:60): beyond it the reference switches comparator
    return Status::Make(StatusCode::NotImplemented, "sort_indices: more than 8 sort keys");
  int64_t length = -1;
  for (auto& a : args) {
    if (a.kind != DatumKind::Array)
      return Status::Make(StatusCode::NotImplemented, "unsupported type for sort_indices operation: the accelerated path sorts arrays");
    if (length < 0) length = a.array->length;
    else if (length != a.array->length) return Status::Make(StatusCode::Invalid, "all columns must have the same length");  // kernels :403-407
  }
  std::vector<int> types, desc, nfirst;
  std::vector<const void*> values;
  std::vector<const uint8_t*> valids;
  std::vector<int64_t> offs;
  for (size_t i = 0; i < keys.size(); i++) {
    const SortKey& key = keys[i];
    if (key.ColumnIndex < 0 || key.ColumnIndex >= (int)args.size())
      return Status::Make(StatusCode::Invalid, "sort key " + std::to_string(i) + " has invalid column index " + std::to_string(key.ColumnIndex));  // :158-160
    const ArrayData& a = *args[key.ColumnIndex].array;
    if (!IsInteger(a.type->id) && !IsFloating(a.type->id))
      return Status::Make(StatusCode::NotImplemented, std::string("sorting not supported for type ") + a.type->name);  // :266-268
    int w = a.type->bit_width / 8;
    types.push_back((int)a.type->id);
    values.push_back(a.length ? (const uint8_t*)a.buffers[1]->dptr + a.offset * w : nullptr);
    valids.push_back((a.buffers[0] && a.null_count != 0) ? (const uint8_t*)a.buffers[0]->dptr : nullptr);
    offs.push_back(a.offset);
    desc.push_back(key.Order == SortOrderDescending);
    nfirst.push_back(key.Placement == SortNullsAtStart);
  }
  auto res = std::make_shared<ArrayData>();
  res->type = GetDataType(Type::UINT64);
  res->length = length;
  res->null_count = 0;
  AHC_RETURN_NOT_OK(s->Allocate(length * 8, &res->buffers[1], /*zero_all=*/false));
  if (length > 0)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_sort_indices_multi(s->ctx(), (int)keys.size(), types.data(), values.data(), valids.data(), offs.data(), length,
                                                          desc.data(), nfirst.data(), (uint64_t*)res->buffers[1]->dptr)));
  *out = Datum::Of(res);
  return Status::OK();
}

void RegisterVectorSort(FunctionRegistry* reg) {
  // Arity: one array, or the columns of a record batch as separate arguments (this layer has no RecordBatch datum)
  reg->AddFunction(std::make_shared<MetaFunction>("sort_indices", Arity{1, true}, nullptr, SortIndicesImpl), false);
  // sortMetaFunc (compute/vector_sort.go:66-85): take(input, sort_indices(input, options))
  reg->AddFunction(std::make_shared<MetaFunction>("sort", Arity{1, false}, nullptr,
      [](ExecCtx* ctx, const FunctionOptions* o, const std::vector<Datum>& args, Datum* out) {
        Datum indices;
        AHC_RETURN_NOT_OK(CallFunction(ctx, "sort_indices", o, args, &indices));
        return CallFunction(ctx, "take", nullptr, {args[0], indices}, out);
      }), false);
}

// ---- cumulative_sum / cumulative_sum_checked ---------------------------------------------------
// Safe numeric cast of the Start scalar to the input type — what safeCastScalar → CastDatum(SafeCastOptions)
// decides (compute/vector_cumulative.go:53-70, kernels/vector_cumulative.go:71-90): integer targets
// need an in-range, integral value; an integer start for a float target must be exactly representable;
// float → float is always allowed.  Failure is arrow.ErrInvalid.
Status SafeCastCumulativeStart(const Scalar& start, const DataType* to, uint8_t out[8]) {
  auto fail = [&](const char* why) {
    return Status::Make(StatusCode::Invalid, std::string("cannot cast cumulative sum start value to ") + to->name + ": " + why);
  };
  memset(out, 0, 8);
  Type from = start.type->id;
  if (from == to->id) { memcpy(out, start.value, 8); return Status::OK(); }
  if (!IsInteger(from) && !IsFloating(from)) return fail("start value is not numeric");
  // widen the source to (sign, magnitude) or a double
  bool src_float = IsFloating(from);
  double f = 0; bool neg = false; uint64_t mag = 0;
  if (src_float) {
    if (from == Type::FLOAT32) { float t; memcpy(&t, start.value, 4); f = t; } else memcpy(&f, start.value, 8);
  } else if (IsSignedInteger(from)) {
    int64_t v = 0;
    switch (start.type->bit_width) {
      case 8: { int8_t t; memcpy(&t, start.value, 1); v = t; break; }
      case 16: { int16_t t; memcpy(&t, start.value, 2); v = t; break; }
      case 32: { int32_t t; memcpy(&t, start.value, 4); v = t; break; }
      default: memcpy(&v, start.value, 8);
    }
    neg = v < 0;
    mag = neg ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
  } else {
    memcpy(&mag, start.value, start.type->bit_width / 8);
  }
  if (IsFloating(to->id)) {
    if (src_float) {  // float64 → float32: plain conversion
      float t = (float)f; memcpy(out, &t, 4);
      return Status::OK();
    }
    const uint64_t lim = to->id == Type::FLOAT32 ? (1ull << 24) : (1ull << 53);  // exactly representable integers
    if (mag > lim) return fail("integer value not exactly representable in the float type");
    double d = neg ? -(double)mag : (double)mag;
    if (to->id == Type::FLOAT32) { float t = (float)d; memcpy(out, &t, 4); } else memcpy(out, &d, 8);
    return Status::OK();
  }
  // integer target
  if (src_float) {
    if (!(f == f) || f != std::trunc(f)) return fail("float value would be truncated");
    if (std::fabs(f) >= 18446744073709551616.0) return fail("value out of range");
    neg = f < 0;
    mag = (uint64_t)std::fabs(f);
  }
  int bits = to->bit_width;
  if (IsSignedInteger(to->id)) {
    uint64_t maxpos = (1ull << (bits - 1)) - 1;
    if (neg ? mag > maxpos + 1 : mag > maxpos) return fail("value out of range");
    int64_t v = neg ? (int64_t)((uint64_t)0 - mag) : (int64_t)mag;
    memcpy(out, &v, bits / 8);
  } else {
    if (neg && mag != 0) return fail("value out of range");
    if (bits < 64 && mag >> bits) return fail("value out of range");
    memcpy(out, &mag, bits / 8);
  }
  return Status::OK();
}

// cumulativeSumExec → cumulativeSumSpans (kernels/vector_cumulative.go:320-360)
static Status ExecCumulativeSum(KernelCtx* k, const ExecSpan& b, ExecResult* out, bool checked) {
  Session* s = k->session;
  ArraySpan in = b.values[0].array;
  const CumulativeOptions* opts = static_cast<const CumulativeOptions*>(k->state);
  // initCumulativeSum :118-146 → cumulativeStartValue :92-116
  uint8_t start[8] = {0};
  bool have_start = false;
  if (opts && opts->Start) {
    if (!opts->Start->valid) return Status::Make(StatusCode::Invalid, "cumulative sum start value must be valid");
    AHC_RETURN_NOT_OK(SafeCastCumulativeStart(*opts->Start, in.type, start));
    have_start = true;
  }
  out->len = in.len;
  if (in.len == 0) return Status::OK();
  AHC_RETURN_NOT_OK(in.UpdateNullCount(s));
  bool needs_validity = in.MayHaveNulls();  // :354
  int w = in.type->bit_width / 8;
  BufferPtr vb, db;
  AHC_RETURN_NOT_OK(k->Allocate(in.len * w, &db, /*zero_all=*/false));  // prepareCumulativeOutput :211-226
  out->buffers[1].WrapBuffer(db);
  if (needs_validity) {
    AHC_RETURN_NOT_OK(k->AllocateBitmap(in.len, &vb));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), vb->dptr, 0xFF, (size_t)((in.len + 7) / 8))));  // memory.Set(validity, 0xFF)
    out->buffers[0].WrapBuffer(vb);
  }
  int64_t nulls = 0;
  AHC_RETURN_NOT_OK(s->FromStatus(ah_cumulative_sum(s->ctx(), (int)in.type->id, Values(in), needs_validity ? in.buffers[0].buf : nullptr, in.offset,
                                                    in.len, have_start ? start : nullptr, opts && opts->SkipNulls, checked ? 1 : 0, db->dptr,
                                                    needs_validity ? (uint8_t*)vb->dptr : nullptr, needs_validity ? &nulls : nullptr)));
  out->nulls = nulls;
  return Status::OK();
}

static const CumulativeOptions kDefaultCumulativeOptions;

// RegisterVectorCumulative (compute/vector_cumulative.go:72-94)
void RegisterVectorCumulative(FunctionRegistry* reg) {
  for (bool checked : {false, true}) {
    auto fn = std::make_shared<VectorFunction>(checked ? "cumulative_sum_checked" : "cumulative_sum", Arity{1, false}, &kDefaultCumulativeOptions);
    fn->chunked = VectorFunction::Chunked::SingleChunk;
    for (Type t : kNumericTypes) {
      exec::VectorKernel k;
      k.sig.in_types = {t};
      k.can_execute_chunkwise = false;  // newCumulativeSumKernel :383-384
      k.exec_fn = [checked](KernelCtx* kc, const ExecSpan& b, ExecResult* o) { return ExecCumulativeSum(kc, b, o, checked); };
      fn->AddKernel(std::move(k));
    }
    reg->AddFunction(fn, false);
  }
}

// ---- fused extension (no reference analogue) --------------------------------
// "greater_filter_sum"(x, scalar) etc.: Σ x over valid slots with x OP scalar, as a 1-element
// array of x's type — what "greater" → "filter" → math.Sum computes in three calls.
void RegisterFusedExtensions(FunctionRegistry* reg) {
  struct F { const char* name; int op; };
  for (F f : {F{"equal_filter_sum", AH_CMP_EQ}, F{"not_equal_filter_sum", AH_CMP_NE}, F{"greater_filter_sum", AH_CMP_GT}, F{"greater_equal_filter_sum", AH_CMP_GE}}) {
    int op = f.op;
    reg->AddFunction(std::make_shared<MetaFunction>(f.name, Arity{2, false}, nullptr,
        [op](ExecCtx* ctx, const FunctionOptions*, const std::vector<Datum>& args, Datum* out) {
          Session* s = ctx->session;
          if (args[0].kind != DatumKind::Array || args[1].kind != DatumKind::Scalar)
            return Status::Make(StatusCode::Invalid, "fused compare-filter-sum needs (array, scalar)");
          const ArrayData& a = *args[0].array;
          if (a.type->id != args[1].scalar->type->id) return Status::Make(StatusCode::TypeError, "operand types differ");
          if (!args[1].scalar->valid) return Status::Make(StatusCode::Invalid, "null threshold");
          auto sc = std::make_shared<Scalar>();
          sc->type = a.type;
          sc->valid = true;
          const uint8_t* valid = a.buffers[0] ? (const uint8_t*)a.buffers[0]->dptr : nullptr;
          const uint8_t* vals = a.length ? (const uint8_t*)a.buffers[1]->dptr + a.offset * 8 : nullptr;
          int64_t cnt = 0;
          if (a.type->id == Type::INT64) {
            int64_t thr, sum = 0; memcpy(&thr, args[1].scalar->value, 8);
            AHC_RETURN_NOT_OK(s->FromStatus(ah_cmp_filter_sum_i64(s->ctx(), op, (const int64_t*)vals, valid, a.offset, a.length, thr, &sum, &cnt)));
            memcpy(sc->value, &sum, 8);
          } else if (a.type->id == Type::FLOAT64) {
            double thr, sum = 0; memcpy(&thr, args[1].scalar->value, 8);
            AHC_RETURN_NOT_OK(s->FromStatus(ah_cmp_filter_sum_f64(s->ctx(), op, (const double*)vals, valid, a.offset, a.length, thr, &sum, &cnt)));
            memcpy(sc->value, &sum, 8);
          } else {
            return Status::Make(StatusCode::NotImplemented, "fused compare-filter-sum is implemented for int64 and float64");
          }
          *out = Datum::Of(sc);
          return Status::OK();
        }), false);
  }
}

}  // namespace compute
}  // namespace arrowhip
