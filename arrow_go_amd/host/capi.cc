// capi.cc — C entry points of libarrowhip_compute.so: the array-level face of the host
// layer, spoken in the Arrow C Data Interface so that ANY Arrow producer (pyarrow in the
// tests; arrow-go's own arrow/cdata package — cdata.go:72, exports.go — in production)
// can hand over arrays.  Import copies the host buffers into HBM (pinned staging →
// hipMemcpyAsync on the copy stream); export copies results back.
//
//   ahc_call(session, "add", "", 2, args, &out)  ≙  compute.CallFunction(ctx, "add", nil, a, b)
//
// The struct layouts are the public Arrow C Data Interface ABI (arrow/cdata/abi.h:50-79).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "arrowhip_compute.h"
#include "ipc.h"

// the public C header: the Arrow C Data / C Device Data structs and every ahc_* prototype — including it here makes the
// compiler check each definition below against its declaration
#include "../../include/arrowhip_compute.h"

using namespace arrowhip;
using compute::Datum;
using compute::DatumKind;

struct ahc_session {
  std::shared_ptr<Session> session;  // shared with every live buffer (Buffer::keep): destroying the handle never frees a pool a datum still returns memory to
  std::unique_ptr<compute::FunctionRegistry> child_registry;  // per-session registry, like SetExecCtx
  compute::ExecCtx ectx;
  std::string err;
};
struct ahc_datum {
  Datum d;
};

#define AHC_EXPORT extern "C" __attribute__((visibility("default")))

static int Fail(ahc_session* s, const Status& st) {
  if (s) s->err = st.ToString();
  return (int)st.code;
}

AHC_EXPORT int ahc_session_create(int device_id, ahc_session** out) {
  auto* s = new ahc_session();
  Status st = Session::Create(device_id, &s->session);
  if (!st.ok()) { fprintf(stderr, "arrowhip_compute: %s\n", st.ToString().c_str()); delete s; *out = nullptr; return (int)st.code; }
  s->child_registry = compute::NewChildRegistry(compute::GetFunctionRegistry());
  s->ectx.Registry = s->child_registry.get();
  s->ectx.session = s->session.get();
  *out = s;
  return 0;
}
AHC_EXPORT void ahc_session_destroy(ahc_session* s) { delete s; }
AHC_EXPORT const char* ahc_last_error(ahc_session* s) { return s ? s->err.c_str() : "null session"; }
AHC_EXPORT void ahc_datum_release(ahc_datum* d) { delete d; }
AHC_EXPORT int ahc_num_functions(void) { return compute::GetFunctionRegistry()->NumFunctions(); }
AHC_EXPORT int ahc_has_function(const char* name) { return compute::GetFunctionRegistry()->GetFunction(name) != nullptr; }
AHC_EXPORT int ahc_function_num_kernels(const char* name) {
  auto* f = compute::GetFunctionRegistry()->GetFunction(name);
  return f ? f->NumKernels() : -1;
}

// fn.DispatchBest(types...) (ScalarFunction; functions.go:199-218 + arithmeticFunction / compareFunction.DispatchBest) or DispatchExact
// (VectorFunction) WITHOUT executing anything: the argument types the chosen kernel takes go to out_type_ids (the implicit casts the
// executor would insert are type_ids → out_type_ids).  The registry is process-global and needs no device: the reference's
// CheckDispatchBest tables run through this on a machine without a GPU.  Error text into err (NUL-terminated, cap bytes).
AHC_EXPORT int ahc_dispatch_best(const char* function, int nargs, const int* type_ids, int* out_type_ids, char* err, int64_t cap) {
  auto fail = [&](const Status& st) { if (err && cap > 0) snprintf(err, (size_t)cap, "%s", st.ToString().c_str()); return (int)st.code; };
  if (err && cap > 0) err[0] = 0;
  auto* f = compute::GetFunctionRegistry()->GetFunction(function ? function : "");
  if (!f) return fail(Status::Make(StatusCode::KeyError, std::string("function '") + (function ? function : "") + "' not found"));
  Status ar = f->CheckArity((size_t)(nargs < 0 ? 0 : nargs));
  if (!ar.ok()) return fail(ar);
  std::vector<const DataType*> types;
  for (int i = 0; i < nargs; i++) {
    const DataType* t = GetDataType((Type)type_ids[i]);
    if (!t) return fail(Status::Make(StatusCode::Invalid, "unknown type id " + std::to_string(type_ids[i])));
    types.push_back(t);
  }
  Status st;
  if (f->Kind() == compute::FuncKind::Scalar) {
    const exec::ScalarKernel* k = nullptr;
    st = static_cast<compute::ScalarFunction*>(f)->DispatchBest(&types, &k);
  } else if (f->Kind() == compute::FuncKind::Vector) {
    const exec::VectorKernel* k = nullptr;
    st = static_cast<compute::VectorFunction*>(f)->DispatchExact(types, &k);
  } else {
    st = Status::Make(StatusCode::NotImplemented, "meta functions dispatch when they execute");
  }
  if (!st.ok()) return fail(st);
  for (int i = 0; i < nargs; i++) out_type_ids[i] = (int)types[i]->id;
  return 0;
}

static const DataType* TypeFromFormat(const char* f) {
  if (f && f[0] == 't') return TemporalStorage(f);  // timestamps, dates, times, durations: integers with a label
  if (f && (f[0] == 'w' || f[0] == 'd') && f[1] == ':') return FixedWidthBinaryFromFormat(f);   // FixedSizeBinary, Decimal128 / 256
  if (!f || !f[0] || f[1]) return nullptr;
  for (Type bt : {Type::STRING, Type::BINARY, Type::LARGE_STRING, Type::LARGE_BINARY})
    if (GetDataType(bt)->format[0] == f[0]) return GetDataType(bt);
  for (int i = 0; i <= (int)Type::FLOAT64; i++) {
    const DataType* t = GetDataType((Type)i);
    if (t && t->format[0] == f[0] && t->id != Type::NA) return t;
  }
  return nullptr;
}

// ImportCArray-like (arrow/cdata/interface.go:153): host buffers → device.  Does not release the source.
static Status ImportOne(ahc_session* s, ArrowArray* arr, ArrowSchema* schema, ArrayDataPtr* out) {
  const DataType* t = TypeFromFormat(schema->format);
  if (!t) return Status::Make(StatusCode::NotImplemented, std::string("unsupported Arrow format '") + (schema->format ? schema->format : "") + "'");
  Status st;
  auto d = std::make_shared<ArrayData>();
  d->type = t;
  if (schema->format[0] == 't') d->logical = schema->format;
  d->length = arr->length;
  d->null_count = arr->null_count;
  d->offset = arr->offset;
  Session* ss = s->session.get();
  int64_t nbits = arr->offset + arr->length;
  int64_t vbytes = (nbits + 7) / 8;
  int64_t dbytes = t->bit_width == 1 ? vbytes : nbits * (t->bit_width / 8);
  int64_t data_bytes = 0;
  if (IsBaseBinary(t->id)) {  // [validity, offsets (offset + length + 1 entries), data (up to the last offset)]
    dbytes = (nbits + 1) * (t->bit_width / 8);
    if (arr->n_buffers >= 3 && arr->buffers[1] != nullptr)
      data_bytes = t->bit_width == 32 ? (int64_t)((const int32_t*)arr->buffers[1])[nbits] : ((const int64_t*)arr->buffers[1])[nbits];
    AHC_RETURN_NOT_OK(ss->Allocate(data_bytes, &d->buffers[2]));
    if (data_bytes > 0 && arr->buffers[2] != nullptr)
      AHC_RETURN_NOT_OK(ss->FromStatus(ah_upload_async(ss->ctx(), d->buffers[2]->dptr, arr->buffers[2], (size_t)data_bytes)));
  }
  if (arr->n_buffers >= 1 && arr->buffers[0] != nullptr && arr->null_count != 0) {
    AHC_RETURN_NOT_OK(ss->Allocate(vbytes, &d->buffers[0]));
    AHC_RETURN_NOT_OK(ss->FromStatus(ah_upload_async(ss->ctx(), d->buffers[0]->dptr, arr->buffers[0], (size_t)vbytes)));
  } else {
    d->null_count = 0;
  }
  AHC_RETURN_NOT_OK(ss->Allocate(dbytes, &d->buffers[1]));
  if (dbytes > 0 && arr->n_buffers >= 2 && arr->buffers[1] != nullptr)
    AHC_RETURN_NOT_OK(ss->FromStatus(ah_upload_async(ss->ctx(), d->buffers[1]->dptr, arr->buffers[1], (size_t)dbytes)));
  AHC_RETURN_NOT_OK(ss->FromStatus(ah_sync(ss->ctx())));
  if (schema->dictionary && arr->dictionary) {
    // a dictionary array (arrow/cdata/cdata.go importDictionary): the struct describes the INDICES, `dictionary` the values.
    // On the device the indices are always int32 (what dictionary_encode produces); the producer's index type is
    // remembered and restored on export.
    if (!IsInteger(t->id)) return Status::Make(StatusCode::Invalid, "dictionary indices must be integers");
    ArrayDataPtr values;
    AHC_RETURN_NOT_OK(ImportOne(s, arr->dictionary, schema->dictionary, &values));
    if (t->id != Type::INT32) {
      Datum casted;
      AHC_RETURN_NOT_OK(compute::CastDatum(&s->ectx, Datum::Of(d), compute::CastOptions::Safe(GetDataType(Type::INT32)), &casted));
      d = casted.array;
    }
    auto dict = std::make_shared<ArrayData>(*d);
    dict->type = GetDataType(Type::DICTIONARY);
    dict->logical.clear();
    dict->dict_index_type = t;
    dict->dict_value_type = values->type;
    dict->dictionary = values;
    d = dict;
  }
  *out = d;
  return Status::OK();
}

AHC_EXPORT int ahc_import(ahc_session* s, ArrowArray* arr, ArrowSchema* schema, ahc_datum** out) {
  *out = nullptr;
  ArrayDataPtr d;
  Status st = ImportOne(s, arr, schema, &d);
  if (arr->release) arr->release(arr);
  if (schema->release) schema->release(schema);
  if (!st.ok()) return Fail(s, st);
  *out = new ahc_datum{Datum::Of(d)};
  return 0;
}

// a host-resident array datum (ahc_import_host, or a result of the streaming executor) is uploaded whole the first time something
// that cannot stream it needs it; the datum is device-resident from then on
static Status ToDevice(ahc_session* s, ahc_datum* d) {
  if (d->d.kind != DatumKind::Array || !d->d.array->on_host) return Status::OK();
  ArrayDataPtr dev;
  AHC_RETURN_NOT_OK(compute::MaterializeOnDevice(s->session.get(), d->d.array, &dev));
  d->d.array = dev;
  return Status::OK();
}
static Status ToDeviceAll(ahc_session* s, int n, ahc_datum** ds) {
  for (int i = 0; i < n; i++) AHC_RETURN_NOT_OK(ToDevice(s, ds[i]));
  return Status::OK();
}

namespace {
struct HostImportHolder {
  ArrowArray arr;
  ~HostImportHolder() { if (arr.release) arr.release(&arr); }
};
}  // namespace

// ImportCArray (arrow/cdata/interface.go:153) WITHOUT the upload: a flat fixed-width column of at least ExecCtx.HostThresholdBytes
// value bytes stays in the producer's buffers (the ArrowArray is moved into the datum and released with it); smaller ones, and every
// other layout, are imported as ahc_import does.  Pin the buffers (ah_host_alloc_pinned / ah_host_register) for the copies to overlap.
AHC_EXPORT int ahc_import_host(ahc_session* s, ArrowArray* arr, ArrowSchema* schema, ahc_datum** out) {
  *out = nullptr;
  const DataType* t = TypeFromFormat(schema->format);
  const bool flat = t && !IsBaseBinary(t->id) && !schema->dictionary && !arr->dictionary && arr->n_children == 0 && arr->n_buffers == 2 && arr->buffers[1] != nullptr;
  // (a Boolean column — a selection vector — counts as the 8-byte column it selects from: it stays with the values it filters)
  const int64_t as_bytes = flat ? (arr->offset + arr->length) * (int64_t)(t->bit_width == 1 ? 8 : t->bit_width / 8) : 0;
  if (!flat || as_bytes < s->ectx.HostThresholdBytes) return ahc_import(s, arr, schema, out);
  auto holder = std::make_shared<HostImportHolder>();
  holder->arr = *arr;
  arr->release = nullptr;   // moved
  Session* ss = s->session.get();
  auto d = std::make_shared<ArrayData>();
  d->type = t;
  if (schema->format[0] == 't') d->logical = schema->format;
  d->length = holder->arr.length;
  d->offset = holder->arr.offset;
  d->null_count = holder->arr.null_count;
  d->on_host = true;
  auto wrap = [&](const void* p, int64_t size) {
    auto b = std::make_shared<Buffer>();
    ss->Keep(b.get());
    b->owned = false;
    b->owner = holder;
    b->hptr = const_cast<void*>(p);
    b->size = size;
    return b;
  };
  const int64_t nbits = d->offset + d->length;
  if (holder->arr.buffers[0] != nullptr && holder->arr.null_count != 0) d->buffers[0] = wrap(holder->arr.buffers[0], (nbits + 7) / 8);
  else d->null_count = 0;
  // (an unknown null count, −1, stays unknown: "may have nulls" is all the streaming executor asks)
  d->buffers[1] = wrap(holder->arr.buffers[1], t->bit_width == 1 ? (nbits + 7) / 8 : nbits * (t->bit_width / 8));
  if (schema->release) schema->release(schema);
  *out = new ahc_datum{Datum::Of(d)};
  return 0;
}
// 1 while the array's only copy is the host's (an uploaded twin makes it device-resident for every later call)
AHC_EXPORT int ahc_datum_on_host(ahc_datum* d) { return d->d.kind == DatumKind::Array && d->d.array->on_host && !d->d.array->device_twin ? 1 : 0; }

// the ExecCtx fields of this session: "chunk_bytes" (ExecCtx.ChunkSize's role for host-resident arguments: bytes of the widest column
// per span, 0 = 32 MiB), "host_threshold_bytes" (ahc_import_host keeps arrays of at least this many value bytes on the host)
AHC_EXPORT int ahc_session_set_option(ahc_session* s, const char* name, int64_t value) {
  const std::string n = name ? name : "";
  if (value < 0) return Fail(s, Status::Make(StatusCode::Invalid, "option '" + n + "': negative value"));
  if (n == "chunk_bytes") s->ectx.ChunkBytes = value;
  else if (n == "host_threshold_bytes") s->ectx.HostThresholdBytes = value;
  else return Fail(s, Status::Make(StatusCode::KeyError, "unknown session option '" + n + "'"));
  return 0;
}

AHC_EXPORT int ahc_scalar(ahc_session* s, int type_id, int valid, const void* value8, ahc_datum** out) {
  const DataType* t = GetDataType((Type)type_id);
  if (!t) return Fail(s, Status::Make(StatusCode::NotImplemented, "unsupported scalar type"));
  auto sc = std::make_shared<Scalar>();
  sc->type = t;
  sc->valid = valid != 0;
  if (value8) memcpy(sc->value, value8, 8);
  *out = new ahc_datum{Datum::Of(sc)};
  return 0;
}

// the temporal label of a datum ("" for a plain one), and setting it on a scalar built with ahc_scalar
AHC_EXPORT const char* ahc_datum_logical(ahc_datum* d) {
  static thread_local std::string keep;
  keep.clear();
  if (d->d.kind == DatumKind::Array) keep = d->d.array->logical;
  if (d->d.kind == DatumKind::Scalar) keep = d->d.scalar->logical;
  if (d->d.kind == DatumKind::Chunked && !d->d.chunks.empty()) keep = d->d.chunks[0]->logical;
  return keep.c_str();
}
AHC_EXPORT int ahc_scalar_set_logical(ahc_session* s, ahc_datum* d, const char* format) {
  if (d->d.kind != DatumKind::Scalar) return Fail(s, Status::Make(StatusCode::Invalid, "not a scalar datum"));
  const DataType* storage = TemporalStorage(format ? format : "");
  if (!storage || storage->id != d->d.scalar->type->id)
    return Fail(s, Status::Make(StatusCode::TypeError, std::string("'") + (format ? format : "") + "' is not a temporal type stored as " + d->d.scalar->type->name));
  d->d.scalar->logical = format;
  return 0;
}

AHC_EXPORT int ahc_datum_info(ahc_datum* d, int* kind, int* type_id, int64_t* length, int64_t* null_count, int* scalar_valid, void* scalar_value8) {
  *kind = (int)d->d.kind;
  *type_id = d->d.type() ? (int)d->d.type()->id : 0;
  *length = d->d.Len();
  *null_count = d->d.kind == DatumKind::Array ? d->d.array->null_count : 0;
  if (d->d.kind == DatumKind::Scalar) {
    *scalar_valid = d->d.scalar->valid;
    if (scalar_value8) memcpy(scalar_value8, d->d.scalar->value, 8);
  }
  return 0;
}

// options: "key=value;key=value"; keys follow the Go struct tags
//   null_selection_behavior=drop|emit_null   bounds_check=0|1   null_encoding_behavior=mask|encode
//   to_type=<type>   safe=0|1   allow_int_overflow=0|1   allow_float_truncate=0|1       (CastOptions)
//   value_set=@<ahc_datum* in hex>   null_matching_behavior=match|skip|emit_null|inconclusive      (SetOptions)
//   order=ascending|descending   null_placement=at_end|at_start                                   (SortOptions, one key)
//   sort_keys=<col>:<asc|desc>:<at_end|at_start>,…                                                (SortOptions, several keys)
//   ndigits=<n>   round_mode=down|up|towards_zero|towards_infinity|half_down|half_up|half_towards_zero|half_towards_infinity|
//   half_to_even|half_to_odd   multiple=<type>:<value>                                            (RoundOptions / RoundToMultipleOptions)
//   skip_nulls=0|1   start=<type>:<value>|null:<type>   (e.g. start=int64:10, start=double:1.5, start=null:int32)
struct ParsedOptions {
  compute::FilterOptions filter;
  compute::TakeOptions take;
  compute::DictionaryEncodeOptions dict;
  compute::CumulativeOptions cumulative;
  compute::CastOptions cast;
  compute::SetOptions set;
  compute::SortOptions sort;
  compute::RoundOptions round;
  compute::RoundToMultipleOptions round_multiple;
  const compute::FunctionOptions* pick = nullptr;
};
static const struct { const char* name; Type id; } kTypeNames[] = {
    {"uint8", Type::UINT8}, {"int8", Type::INT8}, {"uint16", Type::UINT16}, {"int16", Type::INT16}, {"uint32", Type::UINT32},
    {"int32", Type::INT32}, {"uint64", Type::UINT64}, {"int64", Type::INT64}, {"float", Type::FLOAT32}, {"double", Type::FLOAT64}};
// "<type>:<value>" or "null:<type>" → Scalar (scalar.ParseScalar for the numeric types)
static ScalarPtr ParseScalarText(const std::string& v) {
  size_t c = v.find(':');
  if (c == std::string::npos) return nullptr;
  std::string a = v.substr(0, c), b = v.substr(c + 1);
  bool is_null = a == "null";
  const std::string& tname = is_null ? b : a;
  auto sc = std::make_shared<Scalar>();
  for (auto& tn : kTypeNames)
    if (tname == tn.name) sc->type = GetDataType(tn.id);
  if (!sc->type) return nullptr;
  sc->valid = !is_null;
  if (is_null) return sc;
  if (IsFloating(sc->type->id)) {
    double d = strtod(b.c_str(), nullptr);
    if (sc->type->id == Type::FLOAT32) { float f = (float)d; memcpy(sc->value, &f, 4); } else memcpy(sc->value, &d, 8);
  } else if (IsSignedInteger(sc->type->id)) {
    long long x = strtoll(b.c_str(), nullptr, 10);
    memcpy(sc->value, &x, sc->type->bit_width / 8);
  } else {
    unsigned long long x = strtoull(b.c_str(), nullptr, 10);
    memcpy(sc->value, &x, sc->type->bit_width / 8);
  }
  return sc;
}
static void ParseOptions(const char* text, ParsedOptions* p) {
  std::string t = text ? text : "";
  size_t pos = 0;
  while (pos < t.size()) {
    size_t end = t.find(';', pos);
    if (end == std::string::npos) end = t.size();
    std::string kv = t.substr(pos, end - pos);
    size_t eq = kv.find('=');
    if (eq != std::string::npos) {
      std::string k = kv.substr(0, eq), v = kv.substr(eq + 1);
      if (k == "null_selection_behavior") { p->filter.NullSelection = v == "emit_null" ? compute::EmitNulls : compute::DropNulls; p->pick = &p->filter; }
      else if (k == "output_sizing") { p->filter.WorstCaseOutput = v == "worst_case"; p->pick = &p->filter; }
      if (k == "bounds_check") { p->take.BoundsCheck = v != "0"; p->pick = &p->take; }
      if (k == "to_type") {
        p->cast.ToType = v == "bool" ? GetDataType(Type::BOOL) : nullptr;
        for (auto& tn : kTypeNames) if (v == tn.name) p->cast.ToType = GetDataType(tn.id);
        p->pick = &p->cast;
      }
      if (k == "to_logical") { p->cast.ToLogical = v; p->pick = &p->cast; }
      if (k == "safe" && v == "0") { const DataType* t = p->cast.ToType; std::string l = p->cast.ToLogical; p->cast = compute::CastOptions::Unsafe(t); p->cast.ToLogical = l; p->pick = &p->cast; }
      if (k == "allow_int_overflow") { p->cast.AllowIntOverflow = v != "0"; p->pick = &p->cast; }
      if (k == "allow_time_truncate") { p->cast.AllowTimeTruncate = v != "0"; p->pick = &p->cast; }
      if (k == "allow_time_overflow") { p->cast.AllowTimeOverflow = v != "0"; p->pick = &p->cast; }
      if (k == "allow_float_truncate") { p->cast.AllowFloatTruncate = v != "0"; p->pick = &p->cast; }
      if (k == "value_set" && v.size() > 1 && v[0] == '@') {
        ahc_datum* d = (ahc_datum*)(uintptr_t)strtoull(v.c_str() + 1, nullptr, 16);
        if (d && d->d.kind == DatumKind::Array) p->set.ValueSet = d->d.array;
        p->pick = &p->set;
      }
      if (k == "null_matching_behavior") {
        p->set.NullBehavior = v == "skip" ? compute::NullMatchingSkip : v == "emit_null" ? compute::NullMatchingEmitNull
                            : v == "inconclusive" ? compute::NullMatchingInconclusive : compute::NullMatchingMatch;
        p->pick = &p->set;
      }
      if (k == "sort_keys") {
        p->sort.Keys.clear();
        size_t q = 0;
        while (q < v.size()) {
          size_t e = v.find(',', q);
          if (e == std::string::npos) e = v.size();
          std::string item = v.substr(q, e - q);
          compute::SortKey key;
          size_t c1 = item.find(':'), c2 = c1 == std::string::npos ? c1 : item.find(':', c1 + 1);
          key.ColumnIndex = atoi(item.substr(0, c1).c_str());
          std::string ord = c1 == std::string::npos ? "" : item.substr(c1 + 1, c2 == std::string::npos ? std::string::npos : c2 - c1 - 1);
          std::string npl = c2 == std::string::npos ? "" : item.substr(c2 + 1);
          key.Order = ord == "desc" || ord == "descending" ? compute::SortOrderDescending : compute::SortOrderAscending;
          key.Placement = npl == "at_start" ? compute::SortNullsAtStart : compute::SortNullsAtEnd;
          p->sort.Keys.push_back(key);
          q = e + 1;
        }
        p->pick = &p->sort;
      }
      if (k == "order" || k == "null_placement") {
        if (p->sort.Keys.empty()) p->sort.Keys.push_back(compute::SortKey());
        if (k == "order") p->sort.Keys[0].Order = v == "descending" ? compute::SortOrderDescending : compute::SortOrderAscending;
        else p->sort.Keys[0].Placement = v == "at_start" ? compute::SortNullsAtStart : compute::SortNullsAtEnd;
        p->pick = &p->sort;
      }
      if (k == "ndigits") { p->round.NDigits = strtoll(v.c_str(), nullptr, 10); p->pick = &p->round; }
      if (k == "multiple") { p->round_multiple.Multiple = ParseScalarText(v); p->pick = &p->round_multiple; }
      if (k == "round_mode") {
        static const char* names[] = {"down", "up", "towards_zero", "towards_infinity", "half_down", "half_up", "half_towards_zero",
                                      "half_towards_infinity", "half_to_even", "half_to_odd"};
        for (int m = 0; m < 10; m++)
          if (v == names[m]) { p->round.Mode = (compute::RoundMode)m; p->round_multiple.Mode = (compute::RoundMode)m; }
        if (!p->pick) p->pick = &p->round;
      }
      if (k == "skip_nulls") { p->cumulative.SkipNulls = v != "0"; p->pick = &p->cumulative; }
      if (k == "start") { p->cumulative.Start = ParseScalarText(v); p->pick = &p->cumulative; }
      if (k == "null_encoding_behavior") { p->dict.NullEncoding = v == "encode" ? compute::NullEncodingEncode : compute::NullEncodingMask; p->pick = &p->dict; }
    }
    pos = end + 1;
  }
}

AHC_EXPORT int ahc_call(ahc_session* s, const char* name, const char* options, int nargs, ahc_datum** args, ahc_datum** out) {
  *out = nullptr;
  std::vector<Datum> a;
  for (int i = 0; i < nargs; i++) a.push_back(args[i]->d);
  ParsedOptions po;
  ParseOptions(options, &po);
  Datum res;
  // (host-resident arguments are compute::CallFunction's business: streamed span by span where the function allows it, uploaded
  // whole — once, ArrayData::device_twin — where it does not)
  Status st = compute::CallFunction(&s->ectx, name, po.pick, a, &res);
  if (!st.ok()) return Fail(s, st);
  *out = new ahc_datum{res};
  return 0;
}

AHC_EXPORT int ahc_math_sum(ahc_session* s, ahc_datum* d, double* f64, int64_t* i64, uint64_t* u64) {
  if (d->d.kind != DatumKind::Array) return Fail(s, Status::Make(StatusCode::Invalid, "math.Sum needs an array"));
  const ArrayData& a = d->d.array->on_host && d->d.array->device_twin ? *d->d.array->device_twin : *d->d.array;   // uploaded since: the copy in HBM
  Status st;
  if (!a.logical.empty()) return Fail(s, Status::Make(StatusCode::TypeError, "arrow/math has Float64, Int64 and Uint64 Sum only, not " + a.logical));
  if (a.on_host) {   // chunk by chunk through the device, one final reduction (ah_ingest_sum_*), with the session's ChunkBytes
    st = compute::SumHostResident(&s->ectx, a, f64, i64, u64);
    return st.ok() ? 0 : Fail(s, st);
  }
  switch (a.type->id) {
    case Type::FLOAT64: st = math::Float64.Sum(s->session.get(), a, f64); break;
    case Type::INT64: st = math::Int64.Sum(s->session.get(), a, i64); break;
    case Type::UINT64: st = math::Uint64.Sum(s->session.get(), a, u64); break;
    default: st = Status::Make(StatusCode::TypeError, "arrow/math has Float64, Int64 and Uint64 Sum only");
  }
  return st.ok() ? 0 : Fail(s, st);
}

// registry plumbing, exercised by the tests the way compute/registry_test.go does
AHC_EXPORT int ahc_registry_add_alias(ahc_session* s, const char* alias, const char* existing, int allow_overwrite) {
  auto* f = s->ectx.Registry->GetFunction(existing);
  if (!f) return Fail(s, Status::Make(StatusCode::KeyError, std::string("function '") + existing + "' not found"));
  // a MetaFunction forwarding to `existing`, registered under `alias` in the session's child registry
  std::string target = existing;
  auto fn = std::make_shared<compute::MetaFunction>(alias, f->GetArity(), f->DefaultOptions(),
      [target](compute::ExecCtx* c, const compute::FunctionOptions* o, const std::vector<Datum>& args, Datum* out) {
        return compute::CallFunction(c, target, o, args, out);
      });
  Status st = s->ectx.Registry->AddFunction(fn, allow_overwrite != 0);
  return st.ok() ? 0 : Fail(s, st);
}

// ---- export ------------------------------------------------------------------------------------
struct ExportPriv {
  void* bufs[3] = {nullptr, nullptr, nullptr};
  const void* buffer_ptrs[3] = {nullptr, nullptr, nullptr};
  ArrowArray* dict = nullptr;
  ArrowSchema* dict_schema = nullptr;
};
static void ReleaseArray(ArrowArray* a) {
  auto* p = (ExportPriv*)a->private_data;
  if (p) {
    free(p->bufs[0]);
    free(p->bufs[1]);
    free(p->bufs[2]);
    if (p->dict) { if (p->dict->release) p->dict->release(p->dict); free(p->dict); }
    delete p;
  }
  a->release = nullptr;
}
static void ReleaseSchema(ArrowSchema* sc) {
  if (sc->dictionary) { if (sc->dictionary->release) sc->dictionary->release(sc->dictionary); free(sc->dictionary); }
  free(sc->private_data);
  sc->private_data = nullptr;
  sc->release = nullptr;
}

static Status ExportOne(Session* ss, const ArrayData& a, const DataType* t, ArrowArray* arr, ArrowSchema* schema) {
  memset(arr, 0, sizeof(*arr));
  memset(schema, 0, sizeof(*schema));
  auto* p = new ExportPriv();
  int64_t nbits = a.offset + a.length;
  int64_t vbytes = (nbits + 7) / 8;
  int64_t dbytes = t->bit_width == 1 ? vbytes : nbits * (t->bit_width / 8);
  const bool is_binary = IsBaseBinary(t->id);
  if (is_binary) dbytes = (nbits + 1) * (t->bit_width / 8);
  if (a.buffers[0] && a.null_count != 0) {
    p->bufs[0] = calloc(1, (size_t)vbytes + 64);
    AHC_RETURN_NOT_OK(ss->FromStatus(ah_download_async(ss->ctx(), p->bufs[0], a.buffers[0]->dptr, (size_t)vbytes)));
  }
  p->bufs[1] = calloc(1, (size_t)dbytes + 64);
  if (a.buffers[1] && dbytes > 0)
    AHC_RETURN_NOT_OK(ss->FromStatus(ah_download_async(ss->ctx(), p->bufs[1], a.buffers[1]->dptr, (size_t)dbytes)));
  AHC_RETURN_NOT_OK(ss->FromStatus(ah_sync(ss->ctx())));
  if (is_binary) {  // the data buffer's extent is the last offset, now on the host
    int64_t data_bytes = t->bit_width == 32 ? (int64_t)((const int32_t*)p->bufs[1])[nbits] : ((const int64_t*)p->bufs[1])[nbits];
    p->bufs[2] = calloc(1, (size_t)data_bytes + 64);
    if (a.buffers[2] && data_bytes > 0) {
      AHC_RETURN_NOT_OK(ss->FromStatus(ah_download_async(ss->ctx(), p->bufs[2], a.buffers[2]->dptr, (size_t)data_bytes)));
      AHC_RETURN_NOT_OK(ss->FromStatus(ah_sync(ss->ctx())));
    }
  }
  p->buffer_ptrs[0] = p->bufs[0];
  p->buffer_ptrs[1] = p->bufs[1];
  p->buffer_ptrs[2] = p->bufs[2];
  arr->length = a.length;
  arr->null_count = p->bufs[0] ? a.null_count : 0;
  arr->offset = a.offset;
  arr->n_buffers = is_binary ? 3 : 2;
  arr->buffers = p->buffer_ptrs;
  arr->release = ReleaseArray;
  arr->private_data = p;
  schema->format = t->format;
  schema->private_data = nullptr;
  if (!a.logical.empty()) {  // the temporal type the column came in as; the copy lives until the schema is released
    schema->private_data = strdup(a.logical.c_str());
    schema->format = (const char*)schema->private_data;
  }
  schema->name = "";
  schema->flags = 2;  // ARROW_FLAG_NULLABLE
  schema->release = ReleaseSchema;
  return Status::OK();
}

// ExportArrowArray-like (arrow/cdata/exports.go): device → freshly malloc'ed host buffers
namespace {
struct HostExportPriv {
  ArrayDataPtr keep;
  const void* buffer_ptrs[2] = {nullptr, nullptr};
};
void ReleaseHostArray(ArrowArray* a) {
  delete (HostExportPriv*)a->private_data;
  a->private_data = nullptr;
  a->release = nullptr;
}
}  // namespace

AHC_EXPORT int ahc_export(ahc_session* s, ahc_datum* d, ArrowArray* arr, ArrowSchema* schema) {
  if (d->d.kind != DatumKind::Array) return Fail(s, Status::Make(StatusCode::Invalid, "only array datums can be exported"));
  const ArrayData& a = *d->d.array;
  Session* ss = s->session.get();
  if (a.on_host) {   // ZERO COPY: the consumer gets the host buffers themselves; they live until it releases the array
    memset(arr, 0, sizeof(*arr));
    memset(schema, 0, sizeof(*schema));
    auto* p = new HostExportPriv();
    p->keep = d->d.array;
    p->buffer_ptrs[0] = a.buffers[0] && a.null_count != 0 ? a.buffers[0]->hptr : nullptr;
    p->buffer_ptrs[1] = a.buffers[1] ? a.buffers[1]->hptr : nullptr;
    arr->length = a.length;
    arr->null_count = p->buffer_ptrs[0] ? a.null_count : 0;
    arr->offset = a.offset;
    arr->n_buffers = 2;
    arr->buffers = p->buffer_ptrs;
    arr->release = ReleaseHostArray;
    arr->private_data = p;
    schema->format = a.type->format;
    if (!a.logical.empty()) {
      schema->private_data = strdup(a.logical.c_str());
      schema->format = (const char*)schema->private_data;
    }
    schema->name = "";
    schema->flags = 2;  // ARROW_FLAG_NULLABLE
    schema->release = ReleaseSchema;
    return 0;
  }
  if (a.type->id == Type::DICTIONARY) {
    const DataType* it = a.dict_index_type ? a.dict_index_type : GetDataType(Type::INT32);
    Status st;
    if (it->id != Type::INT32) {  // back to the producer's index type (the values fit: they came from it or are a selection of it)
      auto idx = std::make_shared<ArrayData>(a);
      idx->type = GetDataType(Type::INT32);
      idx->dictionary = nullptr;
      Datum casted;
      st = compute::CastDatum(&s->ectx, Datum::Of(idx), compute::CastOptions::Unsafe(it), &casted);
      if (st.ok()) st = ExportOne(ss, *casted.array, it, arr, schema);
    } else {
      st = ExportOne(ss, a, it, arr, schema);
    }
    if (!st.ok()) return Fail(s, st);
    auto* da = (ArrowArray*)calloc(1, sizeof(ArrowArray));
    auto* ds = (ArrowSchema*)calloc(1, sizeof(ArrowSchema));
    st = ExportOne(ss, *a.dictionary, a.dictionary->type, da, ds);
    if (!st.ok()) return Fail(s, st);
    arr->dictionary = da;
    ((ExportPriv*)arr->private_data)->dict = da;
    schema->dictionary = ds;
    return 0;
  }
  Status st = ExportOne(ss, a, a.type, arr, schema);
  return st.ok() ? 0 : Fail(s, st);
}

// ---- Arrow C Device Data Interface (row §8(f)-3; arrow/cdata/abi.h:66-128, the producer / consumer contract of
// arrow/cdata/interface.go applied to device memory) ------------------------------------------------------------
// Import.  ARROW_DEVICE_ROCM on this session's device: ZERO COPY — the buffers are wrapped, the producer's
// ArrowArray is moved into a holder whose destructor calls its release callback once the last buffer of the
// datum (or of anything computed as a view of it) is gone; the compute stream waits for sync_event.
// ARROW_DEVICE_CPU / ARROW_DEVICE_ROCM_HOST: the host-array path (upload).
namespace {
struct ImportHolder {
  ArrowArray arr;
  ~ImportHolder() { if (arr.release) arr.release(&arr); }
};
}  // namespace

AHC_EXPORT int ahc_import_device(ahc_session* s, ArrowDeviceArray* darr, ArrowSchema* schema, ahc_datum** out) {
  *out = nullptr;
  if (darr->device_type == ARROW_DEVICE_CPU || darr->device_type == ARROW_DEVICE_ROCM_HOST)
    return ahc_import(s, &darr->array, schema, out);
  Session* ss = s->session.get();
  Status st;
  const DataType* t = TypeFromFormat(schema->format);
  if (darr->device_type != ARROW_DEVICE_ROCM)
    st = Status::Make(StatusCode::NotImplemented, "ArrowDeviceArray: device type " + std::to_string(darr->device_type) + " is not ROCm / CPU");
  else if (darr->device_id != ah_device_id(ss->ctx()))
    st = Status::Make(StatusCode::Invalid, "ArrowDeviceArray lives on device " + std::to_string(darr->device_id) + ", the session on device " +
                                               std::to_string(ah_device_id(ss->ctx())));
  else if (!t)
    st = Status::Make(StatusCode::NotImplemented, std::string("unsupported Arrow format '") + (schema->format ? schema->format : "") + "'");
  else if (darr->array.n_children != 0 || darr->array.dictionary)
    st = Status::Make(StatusCode::NotImplemented, "ArrowDeviceArray: flat primitive / binary layouts only (no children, no dictionary)");
  else if (darr->array.n_buffers != (IsBaseBinary(t->id) ? 3 : 2))
    st = Status::Make(StatusCode::Invalid, std::string("ArrowDeviceArray: format '") + schema->format + "' needs " + (IsBaseBinary(t->id) ? "3" : "2") +
                                               " buffers, the array has " + std::to_string(darr->array.n_buffers));
  else if (darr->array.length < 0 || darr->array.offset < 0)
    st = Status::Make(StatusCode::Invalid, "ArrowDeviceArray: negative length / offset");
  if (st.ok()) st = ss->FromStatus(ah_wait_event(ss->ctx(), darr->sync_event));
  if (!st.ok()) {
    if (darr->array.release) darr->array.release(&darr->array);
    if (schema->release) schema->release(schema);
    return Fail(s, st);
  }
  auto holder = std::make_shared<ImportHolder>();
  holder->arr = darr->array;       // move: the consumer owns the struct now …
  darr->array.release = nullptr;   // … and marks the source released (C Data Interface "moving")
  const ArrowArray& a = holder->arr;
  auto d = std::make_shared<ArrayData>();
  d->type = t;
  if (schema->format[0] == 't') d->logical = schema->format;
  d->length = a.length;
  d->null_count = a.null_count;
  d->offset = a.offset;
  const int64_t nbits = a.offset + a.length;
  const int64_t vbytes = (nbits + 7) / 8;
  const int64_t dbytes = t->bit_width == 1 ? vbytes : nbits * (t->bit_width / 8);
  auto wrap = [&](const void* p, int64_t size) {
    auto b = std::make_shared<Buffer>();
    ss->Keep(b.get());
    b->dptr = const_cast<void*>(p);
    b->size = size;
    b->owned = false;
    b->owner = holder;
    return b;
  };
  if (a.buffers[0] != nullptr && a.null_count != 0) d->buffers[0] = wrap(a.buffers[0], vbytes);
  else d->null_count = 0;
  if (IsBaseBinary(t->id)) {
    // [validity, offsets (offset + length + 1 entries), data]: the data buffer's extent is the last offset, which lives on
    // the device — one small read-back (the stream already waits for sync_event), as ImportCArray sizes it from the
    // offsets on the host (arrow/cdata/cdata.go importStringLike)
    const int w = t->bit_width / 8;
    int64_t last = 0;
    if (a.buffers[1] != nullptr) {
      d->buffers[1] = wrap(a.buffers[1], (nbits + 1) * w);
      if (w == 4) { int32_t v = 0; st = ss->FromStatus(ah_download_async(ss->ctx(), &v, (const uint8_t*)a.buffers[1] + nbits * 4, 4)); if (st.ok()) st = ss->FromStatus(ah_sync(ss->ctx())); last = v; }
      else { st = ss->FromStatus(ah_download_async(ss->ctx(), &last, (const uint8_t*)a.buffers[1] + nbits * 8, 8)); if (st.ok()) st = ss->FromStatus(ah_sync(ss->ctx())); }
      if (st.ok() && last < 0) st = Status::Make(StatusCode::Invalid, "ArrowDeviceArray: negative last offset");
      if (st.ok() && last > 0 && a.buffers[2] == nullptr) st = Status::Make(StatusCode::Invalid, "ArrowDeviceArray: offsets reach " + std::to_string(last) + " bytes but there is no data buffer");
    } else if (a.length + a.offset != 0) {
      st = Status::Make(StatusCode::Invalid, "ArrowDeviceArray: binary array without an offsets buffer");
    } else {  // an empty array may carry no offsets: give it the single zero Arrow asks for
      st = ss->Allocate(w, &d->buffers[1]);
    }
    if (st.ok()) {
      if (a.buffers[2] != nullptr) d->buffers[2] = wrap(a.buffers[2], last);
      else st = ss->Allocate(0, &d->buffers[2]);
    }
  } else if (a.buffers[1] != nullptr) {
    d->buffers[1] = wrap(a.buffers[1], dbytes);
  } else {
    st = ss->Allocate(dbytes, &d->buffers[1]);  // zero-length arrays may carry no data buffer
  }
  if (schema->release) schema->release(schema);
  if (!st.ok()) return Fail(s, st);
  *out = new ahc_datum{Datum::Of(d)};
  return 0;
}

// Export.  The datum's device buffers are handed out as they are (no copy); the ArrowDeviceArray keeps the
// ArrayData alive until the consumer calls release.  The compute stream is synchronised first, so
// sync_event = NULL ("no synchronisation needed", abi.h:118-124).
namespace {
struct DeviceExportPriv {
  ArrayDataPtr keep;
  const void* buffer_ptrs[3] = {nullptr, nullptr, nullptr};
};
void ReleaseDeviceArray(ArrowArray* a) {
  delete (DeviceExportPriv*)a->private_data;
  a->release = nullptr;
}
}  // namespace

AHC_EXPORT int ahc_export_device(ahc_session* s, ahc_datum* d, ArrowDeviceArray* out, ArrowSchema* schema) {
  if (d->d.kind != DatumKind::Array) return Fail(s, Status::Make(StatusCode::Invalid, "only array datums can be exported"));
  { Status hs = ToDevice(s, d); if (!hs.ok()) return Fail(s, hs); }
  const ArrayDataPtr& a = d->d.array;
  if (a->type->id == Type::DICTIONARY) return Fail(s, Status::Make(StatusCode::NotImplemented, "device export of dictionary arrays"));
  Session* ss = s->session.get();
  Status st = ss->FromStatus(ah_sync(ss->ctx()));
  if (!st.ok()) return Fail(s, st);
  memset(out, 0, sizeof(*out));
  memset(schema, 0, sizeof(*schema));
  auto* p = new DeviceExportPriv();
  p->keep = a;
  const bool has_nulls = a->buffers[0] && a->null_count != 0;
  p->buffer_ptrs[0] = has_nulls ? a->buffers[0]->dptr : nullptr;
  p->buffer_ptrs[1] = a->buffers[1] ? a->buffers[1]->dptr : nullptr;
  const bool is_binary = IsBaseBinary(a->type->id);  // [validity, offsets, data] (arrow/cdata/cdata.go exportArray: one slot per layout buffer)
  if (is_binary) p->buffer_ptrs[2] = a->buffers[2] ? a->buffers[2]->dptr : nullptr;
  out->array.length = a->length;
  out->array.null_count = has_nulls ? a->null_count : 0;
  out->array.offset = a->offset;
  out->array.n_buffers = is_binary ? 3 : 2;
  out->array.buffers = p->buffer_ptrs;
  out->array.release = ReleaseDeviceArray;
  out->array.private_data = p;
  out->device_id = ah_device_id(ss->ctx());
  out->device_type = ARROW_DEVICE_ROCM;
  out->sync_event = nullptr;
  schema->format = a->type->format;
  schema->private_data = nullptr;
  if (!a->logical.empty()) {
    schema->private_data = strdup(a->logical.c_str());
    schema->format = (const char*)schema->private_data;
  }
  schema->name = "";
  schema->flags = 2;  // ARROW_FLAG_NULLABLE
  schema->release = ReleaseSchema;
  return 0;
}

// test / interop introspection: the device pointers behind an array datum
AHC_EXPORT int ahc_datum_buffers(ahc_datum* d, void** validity, void** data) {
  if (d->d.kind != DatumKind::Array) return 1;
  // a host-resident array: the HOST pointers — unless it has been uploaded since (ahc_datum_on_host says which)
  const ArrayData& a = d->d.array->on_host && d->d.array->device_twin ? *d->d.array->device_twin : *d->d.array;
  const bool h = a.on_host;
  *validity = a.buffers[0] ? (h ? a.buffers[0]->hptr : a.buffers[0]->dptr) : nullptr;
  *data = a.buffers[1] ? (h ? a.buffers[1]->hptr : a.buffers[1]->dptr) : nullptr;
  return 0;
}

// ---- expressions ----------------------------------------------------------------------------------
// text form used by the tests:  call := name '(' arg {',' arg} ')' ;  arg := call | '$'N (column N) | '#'N (literal N)
namespace {
struct ExprParser {
  const char* p;
  std::vector<Datum>* lits;
  std::string err;
  void ws() { while (*p == ' ') p++; }
  compute::ExprPtr parse() {
    ws();
    if (*p == '$' || *p == '#') {
      char k = *p++;
      int n = 0, digits = 0;
      while (*p >= '0' && *p <= '9') { n = n * 10 + (*p++ - '0'); digits++; }
      if (!digits) { err = "expected an index"; return nullptr; }
      if (k == '$') return compute::NewRef(n);
      if (n >= (int)lits->size() || (*lits)[n].kind != DatumKind::Scalar) { err = "bad literal index"; return nullptr; }
      return compute::NewLiteral((*lits)[n].scalar);
    }
    std::string name;
    while ((*p >= 'a' && *p <= 'z') || *p == '_' || (*p >= '0' && *p <= '9')) name += *p++;
    ws();
    if (name.empty() || *p != '(') { err = "expected name("; return nullptr; }
    p++;
    std::vector<compute::ExprPtr> args;
    for (;;) {
      auto a = parse();
      if (!a) return nullptr;
      args.push_back(a);
      ws();
      if (*p == ',') { p++; continue; }
      if (*p == ')') { p++; break; }
      err = "expected , or )";
      return nullptr;
    }
    return compute::NewCall(name, args);
  }
};
}  // namespace

// ExecuteScalarExpression over a batch of columns; fuse = 1 → single JIT kernel when possible
AHC_EXPORT int ahc_expr_eval(ahc_session* s, const char* text, int ncols, ahc_datum** cols, int nlits, ahc_datum** lits, int fuse,
                             ahc_datum** out, int* fused_out) {
  *out = nullptr;
  std::vector<Datum> lit_datums;
  for (int i = 0; i < nlits; i++) lit_datums.push_back(lits[i]->d);
  ExprParser parser{text, &lit_datums, ""};
  compute::ExprPtr e = parser.parse();
  if (!e) return Fail(s, Status::Make(StatusCode::Invalid, "expression syntax: " + parser.err));
  compute::ExecBatch batch;
  for (int i = 0; i < ncols; i++) {
    batch.names.push_back("c" + std::to_string(i));
    batch.values.push_back(cols[i]->d);
    if (cols[i]->d.kind == DatumKind::Array) batch.len = cols[i]->d.array->length;
  }
  Datum res;
  bool fused = false;
  Status st = compute::ExecuteScalarExpression(&s->ectx, e, batch, &res, fuse != 0, &fused);
  if (!st.ok()) return Fail(s, st);
  if (fused_out) *fused_out = fused;
  *out = new ahc_datum{res};
  return 0;
}

// compute.Expression as the reference shapes it (Literal | Parameter | Call, arrow/compute/expression.go:52-78, 278-290), post-order
AHC_EXPORT int ahc_expr_eval_tree(ahc_session* s, const ahc_expr_node* nodes, int n_nodes, int ncols, ahc_datum** cols, const char* const* col_names,
                                  int nlits, ahc_datum** lits, int fuse, ahc_datum** out, int* fused_out) {
  *out = nullptr;
  if (!nodes || n_nodes <= 0) return Fail(s, Status::Make(StatusCode::Invalid, "nil expression"));   // exprs/exec.go:441-443
  std::vector<compute::ExprPtr> built((size_t)n_nodes);
  for (int i = 0; i < n_nodes; i++) {
    const ahc_expr_node& nd = nodes[i];
    switch (nd.kind) {
      case AHC_EXPR_LITERAL:
        if (nd.index < 0 || nd.index >= nlits || lits[nd.index]->d.kind != DatumKind::Scalar)
          return Fail(s, Status::Make(StatusCode::Invalid, "expression node " + std::to_string(i) + ": literal " + std::to_string(nd.index) + " is not a scalar datum of the call"));
        built[(size_t)i] = compute::NewLiteral(lits[nd.index]->d.scalar);
        break;
      case AHC_EXPR_FIELD_REF:
        if (nd.index >= 0) built[(size_t)i] = compute::NewRef(nd.index);
        else if (nd.name && *nd.name) built[(size_t)i] = compute::NewFieldRef(nd.name);
        else return Fail(s, Status::Make(StatusCode::Invalid, "expression node " + std::to_string(i) + ": field reference without index or name"));
        break;
      case AHC_EXPR_CALL: {
        if (!nd.name || !*nd.name) return Fail(s, Status::Make(StatusCode::Invalid, "expression node " + std::to_string(i) + ": call without a function name"));
        if (nd.nargs < 0 || (nd.nargs > 0 && !nd.args)) return Fail(s, Status::Make(StatusCode::Invalid, "expression node " + std::to_string(i) + ": bad argument list"));
        std::vector<compute::ExprPtr> args;
        for (int k = 0; k < nd.nargs; k++) {
          if (nd.args[k] < 0 || nd.args[k] >= i)
            return Fail(s, Status::Make(StatusCode::Invalid, "expression node " + std::to_string(i) + ": argument " + std::to_string(nd.args[k]) + " does not come before its call"));
          args.push_back(built[(size_t)nd.args[k]]);
        }
        std::shared_ptr<compute::FunctionOptions> opts;
        if (nd.options && *nd.options) {
          auto po = std::make_shared<ParsedOptions>();
          ParseOptions(nd.options, po.get());
          if (po->pick) opts = std::shared_ptr<compute::FunctionOptions>(po, const_cast<compute::FunctionOptions*>(po->pick));
        }
        built[(size_t)i] = compute::NewCall(nd.name, args, opts);
        break;
      }
      default:
        return Fail(s, Status::Make(StatusCode::Invalid, "expression node " + std::to_string(i) + ": unknown kind " + std::to_string(nd.kind)));
    }
  }
  compute::ExecBatch batch;
  bool have_len = false;
  for (int i = 0; i < ncols; i++) {
    batch.names.push_back(col_names && col_names[i] ? col_names[i] : "c" + std::to_string(i));
    batch.values.push_back(cols[i]->d);
    if (cols[i]->d.kind == DatumKind::Array) {
      // exprs/exec.go makeExecBatch: every array column has the batch's length (as ExecuteScalarSubstrait checks: substrait.cc)
      if (have_len && cols[i]->d.array->length != batch.len) return Fail(s, Status::Make(StatusCode::Invalid, "mismatched length"));
      batch.len = cols[i]->d.array->length;
      have_len = true;
    }
  }
  if (!have_len) batch.len = 1;
  Datum res;
  bool fused = false;
  Status st = compute::ExecuteScalarExpression(&s->ectx, built.back(), batch, &res, fuse != 0, &fused);
  if (!st.ok()) return Fail(s, st);
  if (fused_out) *fused_out = fused;
  *out = new ahc_datum{res};
  return 0;
}

// exprs.ExecuteScalarSubstrait (arrow/compute/exprs/exec.go:465-488) over a serialized ExtendedExpression: substrait.cc
AHC_EXPORT int ahc_expr_eval_substrait(ahc_session* s, const uint8_t* bytes, int64_t len, int ncols, ahc_datum** cols, const char* const* col_names,
                                       int fuse, ahc_datum** out, int* fused_out) {
  *out = nullptr;
  std::vector<Datum> c;
  std::vector<std::string> names;
  for (int i = 0; i < ncols; i++) {
    c.push_back(cols[i]->d);
    if (col_names) names.push_back(col_names[i] ? col_names[i] : "");
  }
  Datum res;
  bool fused = false;
  Status st = compute::ExecuteScalarSubstrait(&s->ectx, bytes, len, c, names, &res, fuse != 0, &fused);
  if (!st.ok()) return Fail(s, st);
  if (fused_out) *fused_out = fused;
  *out = new ahc_datum{res};
  return 0;
}

// Parses a serialized ExtendedExpression without a device and renders what it understood — the parser's own test entry (runs where
// there is no GPU; the reader takes bytes from outside the process, so truncated and random inputs are part of its tests):
//   "name:type,…|output_name=rendered expression|…"    literals as type(hex of the payload), null literals as type(null),
//   field references as $index, options as the mapped function name shows them (add vs add_unchecked); an expression the
//   executor would refuse renders as "!" + its error text.  A malformed message → the status code, the error text in out.
static std::string RenderExpr(const compute::Expression& e) {
  switch (e.kind) {
    case compute::Expression::LITERAL: {
      const ScalarPtr& sc = e.literal.scalar;
      if (!sc || !sc->type) return "?";
      if (!sc->valid) return std::string(sc->type->name) + "(null)";
      std::string hex;
      const int bytes = sc->type->id == Type::BOOL ? 1 : (sc->type->bit_width + 7) / 8;
      for (int i = bytes - 1; i >= 0; i--) { char b[3]; snprintf(b, sizeof b, "%02x", sc->value[i]); hex += b; }
      return std::string(sc->type->name) + "(" + hex + ")";
    }
    case compute::Expression::FIELD_REF: return e.field_name.empty() ? "$" + std::to_string(e.field_index) : e.field_name;
    default: {
      std::string t = e.function + "(";
      for (size_t i = 0; i < e.args.size(); i++) t += (i ? ", " : "") + (e.args[i] ? RenderExpr(*e.args[i]) : std::string("?"));
      if (auto co = std::dynamic_pointer_cast<compute::CastOptions>(e.options))
        t += std::string(" -> ") + (co->ToType ? co->ToType->name : "?") + (co->AllowIntOverflow && co->AllowFloatTruncate ? " unsafe" : " safe");
      return t + ")";
    }
  }
}
AHC_EXPORT int ahc_substrait_inspect(const uint8_t* bytes, int64_t len, char* out, int64_t cap) {
  auto put = [&](const std::string& t) { if (cap > 0) { snprintf(out, (size_t)cap, "%s", t.c_str()); } };
  compute::SubstraitExtended x;
  Status st = compute::ParseSubstraitExtended(bytes, len, &x);
  if (!st.ok()) { put(st.ToString()); return (int)st.code; }
  std::string text;
  for (size_t i = 0; i < x.names.size(); i++) text += (i ? "," : "") + x.names[i] + ":" + (x.types[i] ? std::string(x.types[i]->name) : "?" + x.type_names[i]);
  for (size_t i = 0; i < x.exprs.size(); i++)
    text += "|" + x.out_names[i] + "=" + (x.expr_status[i].ok() && x.exprs[i] ? RenderExpr(*x.exprs[i]) : "!" + x.expr_status[i].ToString());
  put(text);
  return 0;
}

// ---- chunked datums (compute.ChunkedDatum, datum.go:186-230) ----------------------------------------------------
// A chunked datum is assembled from array datums already on the device (the arrays stay owned by the caller).
AHC_EXPORT int ahc_chunked_from_arrays(ahc_session* s, int type_id, int n, ahc_datum** arrays, ahc_datum** out) {
  *out = nullptr;
  const DataType* t = GetDataType((Type)type_id);
  if (!t) return Fail(s, Status::Make(StatusCode::Invalid, "unknown type id " + std::to_string(type_id)));
  { Status hs = ToDeviceAll(s, n, arrays); if (!hs.ok()) return Fail(s, hs); }
  std::vector<ArrayDataPtr> chunks;
  for (int i = 0; i < n; i++) {
    if (arrays[i]->d.kind != DatumKind::Array) return Fail(s, Status::Make(StatusCode::Invalid, "chunks must be arrays"));
    if (arrays[i]->d.array->type->id != t->id)
      return Fail(s, Status::Make(StatusCode::Invalid, std::string("arrow/array: mismatch data type ") + arrays[i]->d.array->type->name + " vs " + t->name));  // arrow/table.go NewChunked
    if (!chunks.empty() && arrays[i]->d.array->logical != chunks[0]->logical)
      return Fail(s, Status::Make(StatusCode::Invalid, "arrow/array: mismatch data type " + arrays[i]->d.array->logical + " vs " + chunks[0]->logical));
    chunks.push_back(arrays[i]->d.array);
  }
  *out = new ahc_datum{Datum::OfChunks(t, std::move(chunks))};
  return 0;
}
AHC_EXPORT int ahc_datum_num_chunks(ahc_datum* d) { return d->d.kind == DatumKind::Chunked ? (int)d->d.chunks.size() : -1; }
AHC_EXPORT int ahc_datum_chunk(ahc_session* s, ahc_datum* d, int i, ahc_datum** out) {
  *out = nullptr;
  if (d->d.kind != DatumKind::Chunked || i < 0 || i >= (int)d->d.chunks.size()) return Fail(s, Status::Make(StatusCode::Invalid, "chunk index out of range"));
  *out = new ahc_datum{Datum::Of(d->d.chunks[i])};
  return 0;
}

// ---- record batches (compute.RecordDatum, datum.go:232-260): named equal-length columns -------------------------
AHC_EXPORT int ahc_record_from_arrays(ahc_session* s, int n, const char* const* names, ahc_datum** arrays, ahc_datum** out) {
  *out = nullptr;
  { Status hs = ToDeviceAll(s, n, arrays); if (!hs.ok()) return Fail(s, hs); }
  std::vector<ArrayDataPtr> cols;
  std::vector<std::string> nm;
  int64_t rows = 0;
  for (int i = 0; i < n; i++) {
    if (arrays[i]->d.kind != DatumKind::Array) return Fail(s, Status::Make(StatusCode::Invalid, "record batch columns must be arrays"));
    if (i == 0) rows = arrays[i]->d.array->length;
    else if (arrays[i]->d.array->length != rows)  // array.NewRecordBatch validate(): arrow/array/record.go
      return Fail(s, Status::Make(StatusCode::Invalid, std::string("arrow/array: mismatch number of rows in column \"") + names[i] + "\": got=" +
                                                           std::to_string(arrays[i]->d.array->length) + ", want=" + std::to_string(rows)));
    cols.push_back(arrays[i]->d.array);
    nm.push_back(names[i]);
  }
  *out = new ahc_datum{Datum::OfRecord(std::move(nm), std::move(cols), rows)};
  return 0;
}
AHC_EXPORT int ahc_record_num_columns(ahc_datum* d) { return d->d.kind == DatumKind::Record ? (int)d->d.chunks.size() : -1; }
AHC_EXPORT int ahc_record_column(ahc_session* s, ahc_datum* d, int i, const char** name, ahc_datum** out) {
  *out = nullptr;
  if (d->d.kind != DatumKind::Record || i < 0 || i >= (int)d->d.chunks.size()) return Fail(s, Status::Make(StatusCode::Invalid, "column index out of range"));
  if (name) *name = d->d.names[i].c_str();
  *out = new ahc_datum{Datum::Of(d->d.chunks[i])};
  return 0;
}

// ---- Arrow IPC stream → HBM-resident record batches (ipc.NewReader / Reader.Next, arrow/ipc/reader.go:97,202) ----
struct ahc_ipc_reader {
  ahc_session* s;
  std::unique_ptr<ipc::StreamReader> r;
};

// `bytes` must outlive the reader (the body of each batch is copied to the device when it is read)
AHC_EXPORT int ahc_ipc_open(ahc_session* s, const uint8_t* bytes, int64_t len, ahc_ipc_reader** out) {
  *out = nullptr;
  std::unique_ptr<ipc::StreamReader> r;
  Status st = ipc::StreamReader::Open(s->session.get(), bytes, len, &r);
  if (!st.ok()) return Fail(s, st);
  *out = new ahc_ipc_reader{s, std::move(r)};
  return 0;
}
AHC_EXPORT void ahc_ipc_close(ahc_ipc_reader* r) { delete r; }
AHC_EXPORT int ahc_ipc_num_fields(ahc_ipc_reader* r) { return (int)r->r->fields().size(); }
AHC_EXPORT int ahc_ipc_field(ahc_ipc_reader* r, int i, const char** name, int* type_id, int* nullable) {
  if (i < 0 || i >= (int)r->r->fields().size()) return Fail(r->s, Status::Make(StatusCode::Invalid, "field index out of range"));
  const ipc::FieldInfo& f = r->r->fields()[i];
  if (name) *name = f.name.c_str();
  if (type_id) *type_id = f.dict_id >= 0 ? (int)Type::DICTIONARY : (int)f.type->id;
  if (nullable) *nullable = f.nullable;
  return 0;
}
// next batch: columns[0..num_fields) receive array datums (caller releases each), *rows its length;
// *rows = -1 at the end of the stream
AHC_EXPORT int ahc_ipc_next(ahc_ipc_reader* r, ahc_datum** columns, int64_t* rows) {
  bool have = false;
  std::vector<ArrayDataPtr> cols;
  int64_t n = 0;
  Status st = r->r->Next(&have, &cols, &n);
  if (!st.ok()) return Fail(r->s, st);
  if (!have) { *rows = -1; return 0; }
  for (size_t i = 0; i < cols.size(); i++) columns[i] = new ahc_datum{Datum::Of(cols[i])};
  *rows = n;
  return 0;
}
AHC_EXPORT int64_t ahc_ipc_bytes_uploaded(ahc_ipc_reader* r) { return r->r->body_bytes_uploaded(); }

// Walks a stream without a device: "hex(name):type_id:nullable,…|rows,rows,…" into out (NUL-terminated), or
// the error text with the status code returned — the parser's own test entry (runs where there is no GPU).
AHC_EXPORT int ahc_ipc_inspect(const uint8_t* bytes, int64_t len, char* out, int64_t cap) {
  auto put = [&](const std::string& t) { if (cap > 0) { snprintf(out, (size_t)cap, "%s", t.c_str()); } };
  std::unique_ptr<ipc::StreamReader> r;
  Status st = ipc::StreamReader::Open(nullptr, bytes, len, &r);
  std::string text;
  if (st.ok()) {
    for (size_t i = 0; i < r->fields().size(); i++) {
      const ipc::FieldInfo& f = r->fields()[i];
      std::string hex;  // names are arbitrary bytes: hex keeps the separators unambiguous
      for (unsigned char ch : f.name) { char b[3]; snprintf(b, sizeof b, "%02x", ch); hex += b; }
      text += (i ? "," : "") + hex + ":" + std::to_string(f.dict_id >= 0 ? (int)Type::DICTIONARY : (int)f.type->id) + ":" + (f.nullable ? "1" : "0");
      std::string lhex;  // the temporal type of the column (or of the dictionary's values), "" for a plain one
      for (unsigned char ch : f.logical) { char b[3]; snprintf(b, sizeof b, "%02x", ch); lhex += b; }
      text += ":" + lhex;
    }
    text += "|";
    bool have = true, first = true;
    while (st.ok()) {
      int64_t rows = 0;
      st = r->Next(&have, nullptr, &rows);
      if (!st.ok() || !have) break;
      text += (first ? "" : ",") + std::to_string(rows);
      first = false;
    }
  }
  if (!st.ok()) { put(st.ToString()); return (int)st.code; }
  put(text);
  return 0;
}
