// expression.cc — compute.Expression and its executor, with kernel fusion.
//
// ≙ arrow/compute/expression.go (Literal / Parameter / Call, NewLiteral :596, NewFieldRef :610,
//   NewCall :617, Call.String :292-328) and arrow/compute/exprs/exec.go:542-700
//   (executeScalarBatch: literal → datum, field reference → column, scalar function → one
//   kernel execution per node).  The unfused route below is that algorithm verbatim on top of
//   CallFunction; the fused route flattens the tree to the postfix program of
//   include/arrowhip.h (ah_expr_compile / ah_expr_execute) so the whole tree is one pass
//   over HBM.
#include "arrowhip_compute.h"

#include <cstring>
#include <map>

namespace arrowhip {
namespace compute {

ExprPtr NewLiteral(ScalarPtr s) {
  auto e = std::make_shared<Expression>();
  e->kind = Expression::LITERAL;
  e->literal = Datum::Of(std::move(s));
  return e;
}
ExprPtr NewFieldRef(const std::string& name) {
  auto e = std::make_shared<Expression>();
  e->kind = Expression::FIELD_REF;
  e->field_name = name;
  return e;
}
ExprPtr NewRef(int index) {
  auto e = std::make_shared<Expression>();
  e->kind = Expression::FIELD_REF;
  e->field_index = index;
  return e;
}
ExprPtr NewCall(const std::string& name, std::vector<ExprPtr> args, std::shared_ptr<FunctionOptions> opts) {
  auto e = std::make_shared<Expression>();
  e->kind = Expression::CALL;
  e->function = name;
  e->args = std::move(args);
  e->options = std::move(opts);
  return e;
}

std::string Expression::ToString() const {
  switch (kind) {
    case LITERAL: return literal.scalar && literal.scalar->valid ? std::string("<") + literal.scalar->type->name + " literal>" : "null";
    case FIELD_REF: return field_name.empty() ? "$" + std::to_string(field_index) : field_name;
    default: {
      std::string s = function + "(";
      for (size_t i = 0; i < args.size(); i++) s += (i ? ", " : "") + args[i]->ToString();
      return s + ")";
    }
  }
}

static Status ResolveField(const Expression& e, const ExecBatch& batch, int* index) {
  if (e.field_index >= 0) {
    if (e.field_index >= (int)batch.values.size())
      return Status::Make(StatusCode::Invalid, "field reference $" + std::to_string(e.field_index) + " is out of range");
    *index = e.field_index;
    return Status::OK();
  }
  for (size_t i = 0; i < batch.names.size(); i++)
    if (batch.names[i] == e.field_name) { *index = (int)i; return Status::OK(); }
  return Status::Make(StatusCode::Invalid, "no match for field reference '" + e.field_name + "'");  // FieldRef.FindOne
}

// executeScalarBatch (exprs/exec.go:542-700), one kernel per call
static Status ExecuteUnfused(ExecCtx* ctx, const Expression& e, const ExecBatch& batch, Datum* out) {
  switch (e.kind) {
    case Expression::LITERAL: *out = e.literal; return Status::OK();
    case Expression::FIELD_REF: {
      int idx;
      AHC_RETURN_NOT_OK(ResolveField(e, batch, &idx));
      *out = batch.values[idx];
      return Status::OK();
    }
    default: {
      std::vector<Datum> args(e.args.size());
      for (size_t i = 0; i < e.args.size(); i++) AHC_RETURN_NOT_OK(ExecuteUnfused(ctx, *e.args[i], batch, &args[i]));
      return CallFunction(ctx, e.function, e.options.get(), args, out);
    }
  }
}

static const std::map<std::string, int>& FusibleOps() {
  static const std::map<std::string, int> m = {
      {"add", AH_X_ADD_CHECKED}, {"add_unchecked", AH_X_ADD}, {"subtract", AH_X_SUB_CHECKED}, {"subtract_unchecked", AH_X_SUB}, {"sub", AH_X_SUB_CHECKED}, {"sub_unchecked", AH_X_SUB},
      {"multiply", AH_X_MUL_CHECKED}, {"multiply_unchecked", AH_X_MUL}, {"negate_unchecked", AH_X_NEGATE},
      {"abs_unchecked", AH_X_ABS}, {"sign", AH_X_SIGN}, {"equal", AH_X_EQ}, {"not_equal", AH_X_NE}, {"greater", AH_X_GT},
      {"greater_equal", AH_X_GE}, {"less", AH_X_LT}, {"less_equal", AH_X_LE}, {"and", AH_X_AND}, {"or", AH_X_OR},
      {"xor", AH_X_XOR}, {"and_not", AH_X_AND_NOT}, {"invert", AH_X_INVERT}};
  return m;
}

struct Flat {
  std::vector<ah_expr_node> nodes;
  std::vector<int> col_index;   // program column → batch column
  std::vector<const Scalar*> lits;
  std::vector<ScalarPtr> owned;  // literals re-typed on the host (safe cast) for the program
};

// conversions that keep every value: the implicit casts of DispatchBest that can never fail their safe-cast
// check, hence may run inside the fused kernel (ah_expr.hip applies the same rule to AH_X_CAST)
static bool ValuePreserving(const DataType* from, const DataType* to) {
  if (from->id == to->id) return true;
  const bool fi = IsInteger(from->id), ti = IsInteger(to->id);
  if (fi && ti) return IsSignedInteger(from->id) == IsSignedInteger(to->id) ? to->bit_width >= from->bit_width
                                                                            : (!IsSignedInteger(from->id) && to->bit_width > from->bit_width);
  if (fi) return to->id == Type::FLOAT64 ? from->bit_width <= 32 : (to->id == Type::FLOAT32 && from->bit_width <= 16);
  return from->id == Type::FLOAT32 && to->id == Type::FLOAT64;
}

// Flattens to the postfix program; *type = the type the subtree evaluates to.  Returns false (without error)
// when the tree has something the fused kernel does not cover.
static bool Flatten(ExecCtx* ctx, const Expression& e, const ExecBatch& batch, Flat* f, Status* st, const DataType** type) {
  switch (e.kind) {
    case Expression::LITERAL:
      if (e.literal.kind != DatumKind::Scalar || (int)f->lits.size() >= 16) return false;
      f->nodes.push_back({AH_X_LITERAL, (int32_t)f->lits.size()});
      f->lits.push_back(e.literal.scalar.get());
      *type = e.literal.scalar->type;
      return true;
    case Expression::FIELD_REF: {
      int idx;
      *st = ResolveField(e, batch, &idx);
      if (!st->ok()) return false;
      const Datum& d = batch.values[idx];
      if (d.kind == DatumKind::Scalar) {
        if ((int)f->lits.size() >= 16) return false;
        f->nodes.push_back({AH_X_LITERAL, (int32_t)f->lits.size()});
        f->lits.push_back(d.scalar.get());
        *type = d.scalar->type;
        return true;
      }
      if (d.kind != DatumKind::Array) return false;
      int pos = -1;
      for (size_t i = 0; i < f->col_index.size(); i++) if (f->col_index[i] == idx) pos = (int)i;
      if (pos < 0) {
        if ((int)f->col_index.size() >= 16) return false;
        pos = (int)f->col_index.size();
        f->col_index.push_back(idx);
      }
      f->nodes.push_back({AH_X_FIELD, pos});
      *type = d.array->type;
      return true;
    }
    default: {
      auto it = FusibleOps().find(e.function);
      if (it == FusibleOps().end()) return false;
      const int op = it->second;
      std::vector<const DataType*> types(e.args.size(), nullptr);
      std::vector<size_t> starts, ends;  // program range of each argument; a cast, if any, goes at its end
      for (size_t i = 0; i < e.args.size(); i++) {
        starts.push_back(f->nodes.size());
        if (!Flatten(ctx, *e.args[i], batch, f, st, &types[i])) return false;
        ends.push_back(f->nodes.size());
      }
      const bool arith_or_cmp = op < AH_X_AND && !(op >= AH_X_NEGATE && op <= AH_X_SIGN);
      if (arith_or_cmp && types.size() == 2 && types[0]->id != types[1]->id) {
        // arithmeticFunction / compareFunction.DispatchBest (arithmetic.go:112-142, scalar_compare.go:37-63):
        // both sides to the common numeric type; only casts that cannot fail are fused
        const DataType* common = CommonNumeric(types);
        if (!common) return false;
        for (int i = 1; i >= 0; i--) {
          if (types[i]->id == common->id) continue;
          if (ends[i] - starts[i] == 1 && f->nodes[starts[i]].op == AH_X_LITERAL) {
            // a scalar operand: the safe cast happens once, on the host side of the call (same CastDatum, same error)
            const int slot = f->nodes[ends[i] - 1].arg;
            Datum casted;
            *st = CastDatum(ctx, Datum::Of(std::make_shared<Scalar>(*f->lits[slot])), CastOptions::Safe(common), &casted);
            if (!st->ok() || casted.kind != DatumKind::Scalar) return false;
            f->owned.push_back(casted.scalar);
            f->lits[slot] = casted.scalar.get();
            continue;
          }
          if (!ValuePreserving(types[i], common)) return false;
          f->nodes.insert(f->nodes.begin() + (long)ends[i], ah_expr_node{AH_X_CAST, (int32_t)common->id});
        }
        types[0] = types[1] = common;
      }
      f->nodes.push_back({op, 0});
      *type = (op >= AH_X_EQ && op <= AH_X_INVERT) ? GetDataType(Type::BOOL) : (types.empty() ? nullptr : types[0]);
      return *type != nullptr;
    }
  }
}

// a temporal literal anywhere in the tree
static bool HasTemporalLiteral(const Expression& e) {
  if (e.kind == Expression::LITERAL) return e.literal.kind == DatumKind::Scalar && !e.literal.scalar->logical.empty();
  for (auto& a : e.args) if (a && HasTemporalLiteral(*a)) return true;
  return false;
}

Status ExecuteScalarExpression(ExecCtx* ctx, const ExprPtr& expr, const ExecBatch& batch, Datum* out, bool fuse, bool* fused_out) {
  if (fused_out) *fused_out = false;
  if (!expr) return Status::Make(StatusCode::Invalid, "nil expression");  // exec.go:441-443
  // host-resident columns: the fused kernel and the per-node calls both read device buffers — upload them once, here
  for (auto& v : batch.values)
    if (v.kind == DatumKind::Array && v.array->on_host) {
      ExecBatch dev(batch);
      AHC_RETURN_NOT_OK(MaterializeAllOnDevice(ctx->session, &dev.values));
      return ExecuteScalarExpression(ctx, expr, dev, out, fuse, fused_out);
    }
  // temporal operands: the type rules live in CallFunction, so every node goes through it (the fused kernel would
  // compute on the integers without checking that a timestamp meets a duration of its own unit)
  for (auto& v : batch.values)
    if (v.kind == DatumKind::Array && !v.array->logical.empty()) fuse = false;
  if (fuse && HasTemporalLiteral(*expr)) fuse = false;
  for (auto& v : batch.values)
    if (v.kind == DatumKind::Array && v.array->length != batch.len)
      return Status::Make(StatusCode::Invalid, "all columns of the batch must have the batch length");
  Session* s = ctx->session;
  Flat f;
  Status st;
  const DataType* root_type = nullptr;
  if (fuse && expr->kind == Expression::CALL && Flatten(ctx, *expr, batch, &f, &st, &root_type) && !f.col_index.empty()) {
    std::vector<int> col_types, lit_types;
    for (int ci : f.col_index) col_types.push_back((int)batch.values[ci].array->type->id);
    for (auto* l : f.lits) lit_types.push_back((int)l->type->id);
    ah_expr* prog = nullptr;
    int out_type = 0;
    int rc = ah_expr_compile(s->ctx(), f.nodes.data(), (int)f.nodes.size(), col_types.data(), (int)col_types.size(), lit_types.data(),
                             (int)lit_types.size(), &prog, &out_type);
    if (rc == AH_OK) {
      std::vector<const void*> cv;
      std::vector<const uint8_t*> cvalid;
      std::vector<int64_t> coff;
      bool any_nulls = false;
      for (int ci : f.col_index) {
        const ArrayData& a = *batch.values[ci].array;
        const uint8_t* base = a.buffers[1] ? (const uint8_t*)a.buffers[1]->dptr : nullptr;
        cv.push_back(a.type->bit_width == 1 ? (const void*)base : (const void*)(base + a.offset * (a.type->bit_width / 8)));
        bool has_valid = a.buffers[0] != nullptr && a.null_count != 0;
        cvalid.push_back(has_valid ? (const uint8_t*)a.buffers[0]->dptr : nullptr);
        coff.push_back(a.offset);
        any_nulls |= has_valid;
      }
      std::vector<uint8_t> lv(8 * (f.lits.size() ? f.lits.size() : 1));
      std::vector<int> lvalid(f.lits.size() ? f.lits.size() : 1);
      for (size_t i = 0; i < f.lits.size(); i++) {
        memcpy(&lv[8 * i], f.lits[i]->value, 8);
        lvalid[i] = f.lits[i]->valid;
        any_nulls |= !f.lits[i]->valid;
      }
      auto res = std::make_shared<ArrayData>();
      res->type = GetDataType((Type)out_type);
      res->length = batch.len;
      int64_t nbytes = res->type->bit_width == 1 ? (batch.len + 7) / 8 : batch.len * (res->type->bit_width / 8);
      AHC_RETURN_NOT_OK(s->Allocate(nbytes, &res->buffers[1]));
      if (any_nulls) AHC_RETURN_NOT_OK(s->AllocateBitmap(batch.len, &res->buffers[0]));
      res->null_count = any_nulls ? kUnknownNullCount : 0;
      AHC_RETURN_NOT_OK(s->FromStatus(ah_expr_execute(s->ctx(), prog, cv.data(), cvalid.data(), coff.data(), lv.data(), lvalid.data(), batch.len,
                                                      res->buffers[1]->dptr, any_nulls ? (uint8_t*)res->buffers[0]->dptr : nullptr)));
      if (fused_out) *fused_out = true;
      *out = Datum::Of(res);
      return Status::OK();
    }
    if (rc != AH_ENOTIMPL) return s->FromStatus(rc);  // a real failure; ENOTIMPL → fall back to per-call execution
  }
  if (!st.ok()) return st;
  return ExecuteUnfused(ctx, *expr, batch, out);
}

}  // namespace compute
}  // namespace arrowhip
