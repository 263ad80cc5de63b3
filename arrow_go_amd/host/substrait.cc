// substrait.cc — the front end of the expression executor: a serialized substrait.ExtendedExpression → compute.Expression
// → ExecuteScalarExpression (expression.cc: one generated kernel where the tree allows it).
//
// ≙ exprs.ExecuteScalarSubstrait / ExecuteScalarExpression / executeScalarBatch (arrow/compute/exprs/exec.go:440-700) together
//   with the reference's Substrait ↔ Arrow conventions: ToArrowSchema / FromSubstraitType (exprs/types.go:520-760), the default
//   function mapping (exprs/types.go:82-142, 196-272: lt / gt / lte / gte → less / greater / less_equal / greater_equal,
//   and / or → and_kleene / or_kleene, the `overflow` option of add / subtract / multiply / divide / power / sqrt / abs:
//   SILENT (the default) → "<name>_unchecked", ERROR → "<name>"), literalToDatum (exec.go:118-330) and the Cast rule
//   (exec.go:556-577: THROW_EXCEPTION → compute.UnsafeCastOptions, UNSPECIFIED → ErrInvalid, RETURN_NULL → ErrNotImplemented).
//
// The reference takes the message as a substrait-go object; a C ABI takes its protobuf wire bytes.  substrait-go and protobuf
// are third-party modules absent from /root/reference (go.mod: github.com/substrait-io/substrait-go/v8, google.golang.org/
// protobuf), so this is a restatement of the PUBLISHED substrait.proto field numbers (substrait-io/substrait proto/substrait/
// {extended_expression,algebra,type,extensions/extensions}.proto) read with a 60-line wire-format decoder; the parity anchor is
// Arrow C++'s own producer — tests/test_expressions.py feeds bytes made by pyarrow.substrait.serialize_expressions — next to
// hand-assembled messages for the arrow-go conventions pyarrow does not emit (unsigned integers as type VARIATIONS of i8 … i64,
// exprs/types.go:58-78, where Arrow C++ uses user-defined types; both are read).
#include "arrowhip_compute.h"

#include <cstring>
#include <functional>
#include <map>

namespace arrowhip {
namespace compute {

namespace {

// ---- protobuf wire format ---------------------------------------------------------------------------------------------------
struct Slice {
  const uint8_t* p = nullptr;
  const uint8_t* end = nullptr;
  bool empty() const { return p >= end; }
  std::string str() const { return std::string((const char*)p, (size_t)(end - p)); }
};
struct Field {
  uint32_t number = 0;
  int wire = 0;          // 0 varint, 1 fixed64, 2 length-delimited, 5 fixed32
  uint64_t value = 0;    // varint / fixed
  Slice sub;             // length-delimited
};
struct Reader {
  Slice s;
  bool bad = false;
  explicit Reader(Slice in) : s(in) {}
  bool varint(uint64_t* v) {
    uint64_t r = 0;
    for (int shift = 0; shift < 70; shift += 7) {
      if (s.p >= s.end) return false;
      const uint8_t c = *s.p++;
      r |= (uint64_t)(c & 0x7f) << (shift < 64 ? shift : 63);
      if (!(c & 0x80)) { *v = r; return true; }
    }
    return false;
  }
  // → false at the end of the message or on malformed input (then `bad` is set)
  bool next(Field* f) {
    if (s.empty()) return false;
    uint64_t key;
    if (!varint(&key)) { bad = true; return false; }
    f->number = (uint32_t)(key >> 3);
    f->wire = (int)(key & 7);
    f->sub = Slice();
    switch (f->wire) {
      case 0: if (!varint(&f->value)) { bad = true; return false; } return true;
      case 1: if (s.end - s.p < 8) { bad = true; return false; } memcpy(&f->value, s.p, 8); s.p += 8; return true;
      case 5: { if (s.end - s.p < 4) { bad = true; return false; } uint32_t v; memcpy(&v, s.p, 4); f->value = v; s.p += 4; return true; }
      case 2: {
        uint64_t len;
        if (!varint(&len) || len > (uint64_t)(s.end - s.p)) { bad = true; return false; }
        f->sub.p = s.p; f->sub.end = s.p + len; s.p += len;
        return true;
      }
      default: bad = true; return false;   // groups (3, 4) are not used by substrait
    }
  }
};

Status Malformed(const char* what) { return Status::Make(StatusCode::Invalid, std::string("substrait: malformed ") + what); }
Status NotImpl(const std::string& what) { return Status::Make(StatusCode::NotImplemented, what); }

// ---- extensions: anchors → names ----------------------------------------------------------------------------------------------
struct Extensions {
  std::map<uint32_t, std::string> uris;                               // extension_uri(s) / extension_urn(s) anchor → text
  std::map<uint32_t, std::pair<std::string, std::string>> functions;  // function_anchor → {uri, name}
  std::map<uint32_t, std::string> types;                              // type_anchor → name ("u32" …: Arrow C++'s user-defined types)
  std::map<uint32_t, std::string> variations;                         // type_variation_anchor → name ("u32" …: arrow-go's variations)
};

Status ReadDeclaration(Slice in, Extensions* ext) {
  Reader r(in);
  Field f;
  while (r.next(&f)) {
    if (f.wire != 2 || f.number < 1 || f.number > 3) continue;
    // ExtensionType {1 uri ref, 2 type_anchor, 3 name, 4 urn ref}, ExtensionTypeVariation {1, 2 anchor, 3 name}, ExtensionFunction {1, 2 anchor, 3 name, 4}
    Reader d(f.sub);
    Field g;
    uint32_t uri = 0, urn = 0, anchor = 0;
    std::string name;
    while (d.next(&g)) {
      if (g.number == 1 && g.wire == 0) uri = (uint32_t)g.value;
      else if (g.number == 2 && g.wire == 0) anchor = (uint32_t)g.value;
      else if (g.number == 3 && g.wire == 2) name = g.sub.str();
      else if (g.number == 4 && g.wire == 0) urn = (uint32_t)g.value;
    }
    if (d.bad) return Malformed("extension declaration");
    if (f.number == 1) ext->types[anchor] = name;
    else if (f.number == 2) ext->variations[anchor] = name;
    else {
      auto it = ext->uris.find(urn ? urn : uri);
      ext->functions[anchor] = {it == ext->uris.end() ? std::string() : it->second, name};
    }
  }
  return r.bad ? Malformed("extension declaration") : Status::OK();
}

// ---- types: FromSubstraitType (exprs/types.go:596-760) for the types this layer has ---------------------------------------------
struct SType {
  const DataType* type = nullptr;   // nullptr: a type this layer does not carry (named in `name`)
  std::string name;
};

const DataType* Unsigned(const std::string& name) {
  if (name == "u8") return GetDataType(Type::UINT8);
  if (name == "u16") return GetDataType(Type::UINT16);
  if (name == "u32") return GetDataType(Type::UINT32);
  if (name == "u64") return GetDataType(Type::UINT64);
  return nullptr;
}

// NamedStruct.names lists the field names depth-first, the names of nested struct fields included (substrait/type.proto; substrait-go's
// NamedStruct does the same): how many names FOLLOW a column's own name — one per field of every struct inside it, through lists and
// maps.  The columns of this layer are flat, but a schema may carry a struct column next to the ones an expression uses.
Status CountNestedNames(Slice in, int depth, size_t* extra) {
  if (depth > 64) return Status::Make(StatusCode::Invalid, "substrait: type nested too deeply");
  Reader r(in);
  Field f;
  while (r.next(&f)) {
    if (f.wire != 2 || (f.number != 25 && f.number != 27 && f.number != 28)) continue;   // struct / list / map
    Reader k(f.sub);
    Field g;
    while (k.next(&g)) {
      if (g.wire != 2 || !(g.number == 1 || (f.number == 28 && g.number == 2))) continue;   // struct.types, list.type, map.key / value
      if (f.number == 25) ++*extra;
      AHC_RETURN_NOT_OK(CountNestedNames(g.sub, depth + 1, extra));
    }
    if (k.bad) return Malformed("type");
  }
  if (r.bad) return Malformed("type");
  return Status::OK();
}

Status ReadType(Slice in, const Extensions& ext, SType* out) {
  Reader r(in);
  Field f;
  *out = SType();
  while (r.next(&f)) {
    if (f.wire != 2) continue;
    uint32_t variation = 0, type_ref = 0;
    Reader k(f.sub);
    Field g;
    while (k.next(&g)) {
      if (g.wire != 0) continue;
      if (f.number == 30) { if (g.number == 1) type_ref = (uint32_t)g.value; else if (g.number == 2) variation = (uint32_t)g.value; }
      else if (g.number == 1) variation = (uint32_t)g.value;   // {1 type_variation_reference, 2 nullability}
    }
    if (k.bad) return Malformed("type");
    Type id = Type::NA;
    switch (f.number) {
      case 1: id = Type::BOOL; out->name = "boolean"; break;
      case 2: id = Type::INT8; out->name = "i8"; break;
      case 3: id = Type::INT16; out->name = "i16"; break;
      case 5: id = Type::INT32; out->name = "i32"; break;
      case 7: id = Type::INT64; out->name = "i64"; break;
      case 10: id = Type::FLOAT32; out->name = "fp32"; break;
      case 11: id = Type::FLOAT64; out->name = "fp64"; break;
      case 12: out->name = "string"; return Status::OK();
      case 13: out->name = "binary"; return Status::OK();
      case 14: out->name = "timestamp"; return Status::OK();
      case 16: out->name = "date"; return Status::OK();
      case 17: out->name = "time"; return Status::OK();
      case 24: out->name = "decimal"; return Status::OK();
      case 25: out->name = "struct"; return Status::OK();
      case 27: out->name = "list"; return Status::OK();
      case 32: out->name = "uuid"; return Status::OK();
      case 30: {   // user-defined: Arrow C++ writes u8 … u64 this way
        auto it = ext.types.find(type_ref);
        out->name = it == ext.types.end() ? "user-defined type" : it->second;
        out->type = it == ext.types.end() ? nullptr : Unsigned(it->second);
        return Status::OK();
      }
      default: out->name = "type #" + std::to_string(f.number); return Status::OK();
    }
    out->type = GetDataType(id);
    if (variation != 0) {   // arrow-go writes u8 … u64 as variations of i8 … i64 (exprs/types.go:58-78)
      auto it = ext.variations.find(variation);
      const DataType* u = it == ext.variations.end() ? nullptr : Unsigned(it->second);
      if (!u || u->bit_width != out->type->bit_width) {
        out->type = nullptr;
        out->name += it == ext.variations.end() ? " (unknown type variation)" : " variation " + it->second;
      } else {
        out->type = u;
        out->name = it->second;
      }
    }
    return Status::OK();
  }
  return r.bad ? Malformed("type") : Status::Make(StatusCode::Invalid, "substrait: empty type");
}

ScalarPtr MakeScalar(const DataType* t, bool valid, uint64_t bits) {
  auto s = std::make_shared<Scalar>();
  s->type = t;
  s->valid = valid;
  memcpy(s->value, &bits, 8);
  return s;
}

// ---- expressions ------------------------------------------------------------------------------------------------------------
struct Parser {
  const Extensions& ext;
  int depth = 0;

  // literalToDatum (exec.go:118-330) for the primitive literals
  Status Literal(Slice in, ExprPtr* out) {
    Reader r(in);
    Field f;
    ScalarPtr s;
    uint32_t variation = 0;
    std::string unsupported;
    while (r.next(&f)) {
      switch (f.number) {
        case 1: if (f.wire == 0) s = MakeScalar(GetDataType(Type::BOOL), true, f.value ? 1 : 0); break;
        case 2: if (f.wire == 0) s = MakeScalar(GetDataType(Type::INT8), true, f.value); break;     // int32 on the wire: sign-extended varint
        case 3: if (f.wire == 0) s = MakeScalar(GetDataType(Type::INT16), true, f.value); break;
        case 5: if (f.wire == 0) s = MakeScalar(GetDataType(Type::INT32), true, f.value); break;
        case 7: if (f.wire == 0) s = MakeScalar(GetDataType(Type::INT64), true, f.value); break;
        case 10: if (f.wire == 5) s = MakeScalar(GetDataType(Type::FLOAT32), true, f.value); break;
        case 11: if (f.wire == 1) s = MakeScalar(GetDataType(Type::FLOAT64), true, f.value); break;
        case 29: {   // typed null
          if (f.wire != 2) break;
          SType t;
          AHC_RETURN_NOT_OK(ReadType(f.sub, ext, &t));
          if (!t.type) return NotImpl("substrait: null literal of type " + t.name);
          s = MakeScalar(t.type, false, 0);
          break;
        }
        case 33: {   // user-defined {1 type_reference, 2 value: google.protobuf.Any {1 type_url, 2 value}, 3 type_parameters}: Arrow C++'s unsigned literals
          if (f.wire != 2) break;
          Reader u(f.sub);
          Field g;
          uint32_t type_ref = 0;
          Slice any;
          while (u.next(&g)) {
            if (g.number == 1 && g.wire == 0) type_ref = (uint32_t)g.value;
            else if (g.number == 2 && g.wire == 2) any = g.sub;
          }
          auto it = ext.types.find(type_ref);
          const DataType* t = it == ext.types.end() ? nullptr : Unsigned(it->second);
          if (u.bad || !t || !any.p) return NotImpl("substrait: user-defined literal" + (it == ext.types.end() ? std::string() : " of type " + it->second));
          Reader a(any);
          Slice payload;
          while (a.next(&g)) if (g.number == 2 && g.wire == 2) payload = g.sub;
          uint64_t v = 0;
          Reader w(payload);   // google.protobuf.UInt64Value / UInt32Value {1 value}
          while (w.next(&g)) if (g.number == 1 && g.wire == 0) v = g.value;
          if (a.bad || w.bad) return Malformed("user-defined literal");
          s = MakeScalar(t, true, v);
          break;
        }
        case 50: break;   // nullable: the literal's TYPE nullability, not its value
        case 51: if (f.wire == 0) variation = (uint32_t)f.value; break;
        case 12: unsupported = "string"; break;
        case 13: unsupported = "binary"; break;
        case 14: unsupported = "timestamp"; break;
        case 16: unsupported = "date"; break;
        case 17: unsupported = "time"; break;
        case 24: unsupported = "decimal"; break;
        default: if (f.number < 50) unsupported = "#" + std::to_string(f.number); break;
      }
    }
    if (r.bad) return Malformed("literal");
    if (!s) return NotImpl("substrait: " + (unsupported.empty() ? std::string("empty") : unsupported) + " literal");
    if (variation != 0 && s->valid) {   // arrow-go: an unsigned literal is the signed one with a type variation
      auto it = ext.variations.find(variation);
      const DataType* u = it == ext.variations.end() ? nullptr : Unsigned(it->second);
      if (!u || u->bit_width != s->type->bit_width) return NotImpl("substrait: literal with an unknown type variation");
      s->type = u;
    }
    // narrow the sign-extended 64-bit payload to the type's width (the kernels read bit_width / 8 bytes)
    *out = NewLiteral(s);
    return Status::OK();
  }

  // execFieldRef (exec.go:490-540): root reference, direct struct-field segment, no nesting (this layer has no struct columns)
  Status FieldRef(Slice in, ExprPtr* out) {
    Reader r(in);
    Field f;
    int index = -1;
    bool root = false, other_root = false;
    while (r.next(&f)) {
      if (f.number == 1 && f.wire == 2) {          // direct_reference: ReferenceSegment
        Reader seg(f.sub);
        Field g;
        bool struct_field = false;
        while (seg.next(&g)) {
          if (g.number == 2 && g.wire == 2) {     // struct_field {1 field, 2 child}
            struct_field = true;
            index = 0;                              // proto3: field 0 is not on the wire
            Reader sf(g.sub);
            Field h;
            while (sf.next(&h)) {
              if (h.number == 1 && h.wire == 0) index = (int)(int32_t)h.value;
              else if (h.number == 2) return NotImpl("substrait: nested field references (struct columns are not carried by this layer)");
            }
            if (sf.bad) return Malformed("field reference");
          } else if (g.wire == 2) {
            return NotImpl("substrait: only struct-field reference segments are implemented");
          }
        }
        if (seg.bad || !struct_field) return Malformed("field reference");
      } else if (f.number == 2) {
        return NotImpl("substrait: only direct references are implemented");      // exec.go:496-498
      } else if (f.number == 4) {
        root = true;
      } else if (f.number == 3 || f.number == 5) {
        other_root = true;
      }
    }
    if (r.bad) return Malformed("field reference");
    if (other_root || !root) return NotImpl("substrait: only RootReference is implemented");   // exec.go:491-493
    if (index < 0) return Status::Make(StatusCode::Invalid, "substrait: field reference without a struct field");
    *out = NewRef(index);
    return Status::OK();
  }

  Status ScalarFunction(Slice in, ExprPtr* out) {
    Reader r(in);
    Field f;
    uint32_t ref = 0;
    std::vector<ExprPtr> args;
    std::map<std::string, std::vector<std::string>> options;
    while (r.next(&f)) {
      if (f.number == 1 && f.wire == 0) ref = (uint32_t)f.value;
      else if (f.number == 4 && f.wire == 2) {      // FunctionArgument {1 enum, 2 type, 3 value}
        Reader a(f.sub);
        Field g;
        bool got = false;
        while (a.next(&g)) {
          if (g.number == 3 && g.wire == 2) {
            ExprPtr e;
            AHC_RETURN_NOT_OK(Expr(g.sub, &e));
            args.push_back(e);
            got = true;
          } else if (g.number == 1 || g.number == 2) {
            return NotImpl("substrait: enum / type function arguments");   // (the reference turns an enum into a string scalar: no kernel here takes one)
          }
        }
        if (a.bad || !got) return Malformed("function argument");
      } else if (f.number == 2 && f.wire == 2) {    // deprecated `args`
        ExprPtr e;
        AHC_RETURN_NOT_OK(Expr(f.sub, &e));
        args.push_back(e);
      } else if (f.number == 5 && f.wire == 2) {    // FunctionOption {1 name, 2 preference …}
        Reader o(f.sub);
        Field g;
        std::string name;
        std::vector<std::string> pref;
        while (o.next(&g)) {
          if (g.number == 1 && g.wire == 2) name = g.sub.str();
          else if (g.number == 2 && g.wire == 2) pref.push_back(g.sub.str());
        }
        if (o.bad) return Malformed("function option");
        options[name] = pref;
      }
    }
    if (r.bad) return Malformed("scalar function");
    auto it = ext.functions.find(ref);
    if (it == ext.functions.end()) return NotImpl("substrait: function reference " + std::to_string(ref) + " is not declared");   // DecodeFunction !ok
    const std::string& uri = it->second.first;
    std::string name = it->second.second;
    const size_t colon = name.find(':');   // "add:i32_i32": the signature suffix is not part of the name (strings.Cut, types.go:235)
    if (colon != std::string::npos) name = name.substr(0, colon);
    auto in_set = [&](const char* tag) { return uri.find(tag) != std::string::npos; };
    std::string fname;
    if (in_set("functions_arithmetic") && (name == "add" || name == "subtract" || name == "multiply" || name == "divide" || name == "power" ||
                                           name == "sqrt" || name == "abs")) {
      // decodeOptionlessOverflowableArithmetic (types.go:258-276) + parseOption (:172-192): the first preference that is implemented
      std::string overflow = "SILENT";
      auto oi = options.find("overflow");
      if (oi != options.end() && !oi->second.empty()) {
        overflow.clear();
        for (auto& p : oi->second) {
          if (p != "SILENT" && p != "SATURATE" && p != "ERROR") return Status::Make(StatusCode::Invalid, "substrait: overflow option '" + p + "'");
          if (p == "SILENT" || p == "ERROR") { overflow = p; break; }
        }
        if (overflow.empty()) return NotImpl("substrait: overflow behaviour SATURATE");
      }
      fname = overflow == "SILENT" ? name + "_unchecked" : name;
    } else if (in_set("functions_comparison") && (name == "equal" || name == "not_equal" || name == "lt" || name == "lte" || name == "gt" || name == "gte" ||
                                                  name == "is_null" || name == "is_not_null" || name == "is_nan")) {
      fname = name == "lt" ? "less" : name == "gt" ? "greater" : name == "lte" ? "less_equal" : name == "gte" ? "greater_equal" : name;   // types.go:196-203
    } else if (in_set("functions_boolean") && (name == "and" || name == "or" || name == "not")) {
      fname = name == "and" ? "and_kleene" : name == "or" ? "or_kleene" : "invert";   // the reference's "not" IS its invert kernel (scalar_bool.go:138)
    } else {
      return NotImpl("substrait: " + name + " (" + uri + ")");   // exec.go:606-609
    }
    *out = NewCall(fname, args);
    return Status::OK();
  }

  Status Cast(Slice in, ExprPtr* out) {
    Reader r(in);
    Field f;
    SType t;
    bool have_type = false;
    ExprPtr input;
    uint64_t behavior = 0;
    while (r.next(&f)) {
      if (f.number == 1 && f.wire == 2) { AHC_RETURN_NOT_OK(ReadType(f.sub, ext, &t)); have_type = true; }
      else if (f.number == 2 && f.wire == 2) AHC_RETURN_NOT_OK(Expr(f.sub, &input));
      else if (f.number == 3 && f.wire == 0) behavior = f.value;
    }
    if (r.bad) return Malformed("cast");
    if (!input) return Status::Make(StatusCode::Invalid, "cast without argument to cast");                 // exec.go:557-559
    if (!have_type) return Status::Make(StatusCode::Invalid, "could not determine type for cast");
    if (!t.type) return NotImpl("substrait: cast to " + t.name);
    if (behavior == 0) return Status::Make(StatusCode::Invalid, "cast behavior unspecified");              // :573
    if (behavior == 1) return NotImpl("cast behavior return nil");                                         // :575
    auto opts = std::make_shared<CastOptions>(CastOptions::Unsafe(t.type));                                 // :571 BehaviorThrowException → UnsafeCastOptions
    *out = NewCall("cast", {input}, opts);
    return Status::OK();
  }

  Status Expr(Slice in, ExprPtr* out) {
    if (++depth > 256) return Status::Make(StatusCode::Invalid, "substrait: expression nested too deeply");
    Reader r(in);
    Field f;
    Status st = Status::Make(StatusCode::Invalid, "substrait: empty expression");
    // rex_type is a oneof: of several members on the wire the LAST one counts (protobuf's rule; a writer that merges messages
    // produces exactly that) — every member read replaces what the one before it left
    while (r.next(&f)) {
      if (f.wire != 2) continue;
      switch (f.number) {
        case 1: st = Literal(f.sub, out); break;
        case 2: st = FieldRef(f.sub, out); break;
        case 3: st = ScalarFunction(f.sub, out); break;
        case 11: st = Cast(f.sub, out); break;
        case 5: st = Status::Make(StatusCode::Invalid, "ExecuteScalarExpression cannot execute non-scalar expressions"); break;   // exec.go:545-548 (window function)
        case 6: st = NotImpl("substrait: if-then expressions"); break;     // (the reference's switch ends in ErrNotImplemented, exec.go:699)
        case 7: st = NotImpl("substrait: switch expressions"); break;
        case 8: st = NotImpl("substrait: singular-or-list expressions"); break;
        case 9: st = NotImpl("substrait: multi-or-list expressions"); break;
        case 12: st = NotImpl("substrait: subqueries"); break;
        case 13: st = NotImpl("substrait: nested expressions"); break;
        default: st = NotImpl("substrait: expression kind #" + std::to_string(f.number)); break;
      }
    }
    depth--;
    if (r.bad) return Malformed("expression");
    return st;
  }
};

// the payload of a literal scalar is 8 sign-extended bytes from the wire; the kernels read the type's width — already little-endian
}  // namespace

Status ParseSubstraitExtended(const uint8_t* bytes, int64_t len, SubstraitExtended* out) {
  if (!bytes || len < 0) return Status::Make(StatusCode::Invalid, "nil expression");   // exec.go:466-468
  *out = SubstraitExtended();
  Slice all;
  all.p = bytes; all.end = bytes + len;
  Extensions ext;
  // pass 1: URIs / URNs (declarations refer to them); pass 2: declarations; pass 3: schema and expressions
  for (int pass = 0; pass < 3; pass++) {
    Reader r(all);
    Field f;
    while (r.next(&f)) {
      if (f.wire != 2) continue;
      if (pass == 0 && (f.number == 1 || f.number == 8)) {   // SimpleExtensionURI / SimpleExtensionURN {1 anchor, 2 text}
        Reader u(f.sub);
        Field g;
        uint32_t anchor = 0;
        std::string text;
        while (u.next(&g)) {
          if (g.number == 1 && g.wire == 0) anchor = (uint32_t)g.value;
          else if (g.number == 2 && g.wire == 2) text = g.sub.str();
        }
        if (u.bad) return Malformed("extension uri");
        ext.uris[anchor] = text;
      } else if (pass == 1 && f.number == 2) {
        AHC_RETURN_NOT_OK(ReadDeclaration(f.sub, &ext));
      } else if (pass == 2 && f.number == 4) {              // NamedStruct {1 names …, 2 struct {1 types …}}
        Reader n(f.sub);
        Field g;
        std::vector<std::string> all_names;   // depth-first, nested struct fields included
        std::vector<size_t> nested;           // per column: names that follow its own
        while (n.next(&g)) {
          if (g.number == 1 && g.wire == 2) all_names.push_back(g.sub.str());
          else if (g.number == 2 && g.wire == 2) {
            Reader st(g.sub);
            Field h;
            while (st.next(&h)) {
              if (h.number != 1 || h.wire != 2) continue;
              SType t;
              AHC_RETURN_NOT_OK(ReadType(h.sub, ext, &t));
              size_t extra = 0;
              AHC_RETURN_NOT_OK(CountNestedNames(h.sub, 0, &extra));
              out->types.push_back(t.type);
              out->type_names.push_back(t.name);
              nested.push_back(extra);
            }
            if (st.bad) return Malformed("base schema");
          }
        }
        if (n.bad) return Malformed("base schema");
        size_t pos = 0, nested_total = 0;
        for (size_t i = 0; i < nested.size(); i++) {
          if (pos < all_names.size()) out->names.push_back(all_names[pos]);
          pos += 1 + nested[i];
          nested_total += nested[i];
        }
        if (pos != all_names.size())
          return Status::Make(StatusCode::Invalid, "substrait: base schema has " + std::to_string(all_names.size()) + " names for " + std::to_string(nested.size()) +
                                                       " columns and " + std::to_string(nested_total) + " nested struct fields");
      } else if (pass == 2 && f.number == 3) {              // ExpressionReference {1 expression, 2 measure, 3 output_names …}
        Reader e(f.sub);
        Field g;
        ExprPtr ex;
        Status est = NotImpl("measures not implemented");   // exec.go:477-479
        std::string name;
        while (e.next(&g)) {
          if (g.number == 1 && g.wire == 2) { Parser p{ext}; est = p.Expr(g.sub, &ex); }
          else if (g.number == 3 && g.wire == 2 && name.empty()) name = g.sub.str();
        }
        if (e.bad) return Malformed("expression reference");
        out->exprs.push_back(ex);
        out->expr_status.push_back(est);
        out->out_names.push_back(name);
      }
    }
    if (r.bad) return Malformed("extended expression");
  }
  if (out->names.size() != out->types.size()) return Status::Make(StatusCode::Invalid, "substrait: base schema has " + std::to_string(out->names.size()) +
                                                                                            " names for " + std::to_string(out->types.size()) + " types");
  return Status::OK();
}

Status ExecuteScalarSubstrait(ExecCtx* ctx, const uint8_t* bytes, int64_t len, const std::vector<Datum>& cols, const std::vector<std::string>& col_names,
                              Datum* out, bool fuse, bool* fused_out) {
  SubstraitExtended x;
  AHC_RETURN_NOT_OK(ParseSubstraitExtended(bytes, len, &x));
  if (x.exprs.empty()) return Status::Make(StatusCode::Invalid, "no referred expression to execute");         // exec.go:473
  if (x.exprs.size() > 1) return NotImpl("only single referred expression implemented");                       // :480
  if (!x.expr_status[0].ok()) return x.expr_status[0];
  // makeExecBatch (exec.go:384-438): the input's columns by NAME when names are given (missing fields → null scalars of the
  // field's type, extra columns ignored, any order), else by position; every supplied column must have the schema's type
  ExecBatch batch;
  const size_t nfields = x.names.size();
  if (col_names.empty() && cols.size() > nfields)
    return Status::Make(StatusCode::Invalid, "mismatched length: " + std::to_string(cols.size()) + " columns for a schema of " + std::to_string(nfields) + " fields");
  bool have_len = false;
  for (size_t i = 0; i < nfields; i++) {
    const Datum* d = nullptr;
    if (col_names.empty()) {
      if (i < cols.size()) d = &cols[i];
    } else {
      for (size_t k = 0; k < col_names.size() && k < cols.size(); k++) if (col_names[k] == x.names[i]) { d = &cols[k]; break; }
    }
    batch.names.push_back(x.names[i]);
    if (!d) {
      if (!x.types[i]) { batch.values.push_back(Datum()); continue; }   // an unsupported field that nothing supplies: an error only if referenced
      batch.values.push_back(Datum::Of(MakeScalar(x.types[i], false, 0)));
      continue;
    }
    const DataType* got = d->kind == DatumKind::Array ? d->array->type : d->kind == DatumKind::Scalar ? d->scalar->type : nullptr;
    if (!got) return Status::Make(StatusCode::Invalid, "substrait: input column " + x.names[i] + " must be an array or a scalar");
    if (!x.types[i] || got->id != x.types[i]->id)     // execFieldRef's check (exec.go:524-537), made once per supplied column
      return Status::Make(StatusCode::Invalid, "referenced field " + x.names[i] + " was " + got->name + ", but should have been " +
                                                   (x.types[i] ? std::string(x.types[i]->name) : x.type_names[i]));
    batch.values.push_back(*d);
    if (d->kind == DatumKind::Array) {
      if (have_len && d->array->length != batch.len) return Status::Make(StatusCode::Invalid, "mismatched length");
      batch.len = d->array->length;
      have_len = true;
    }
  }
  if (!have_len) batch.len = 1;   // all scalars: one row (exec.go:660-664)
  // every referenced field must exist (execFieldRef: arrow.ErrInvalid, exec.go:512-514) and be of a type this layer carries
  Status ref_st;
  std::function<void(const Expression&)> check = [&](const Expression& e) {
    if (!ref_st.ok()) return;
    if (e.kind == Expression::FIELD_REF) {
      if (e.field_index < 0 || (size_t)e.field_index >= nfields)
        ref_st = Status::Make(StatusCode::Invalid, "field reference " + std::to_string(e.field_index) + " is outside the base schema's " + std::to_string(nfields) + " fields");
      else if (batch.values[(size_t)e.field_index].kind == DatumKind::None)
        ref_st = NotImpl("substrait: field " + x.names[(size_t)e.field_index] + " has type " + x.type_names[(size_t)e.field_index] + ", which this layer does not carry");
    }
    for (auto& a : e.args) if (a) check(*a);
  };
  check(*x.exprs[0]);
  if (!ref_st.ok()) return ref_st;
  return ExecuteScalarExpression(ctx, x.exprs[0], batch, out, fuse, fused_out);
}

}  // namespace compute
}  // namespace arrowhip
