// core.cc — data types, device session, function registry and the two executors.
// Mirrors arrow/compute/{registry.go, functions.go, exec.go, executor.go}; every
// non-trivial rule cites the line it restates.  No arithmetic on the host: null
// propagation, popcounts and fills are enqueued on the GPU through include/arrowhip.h.
#include <algorithm>
#include "arrowhip_compute.h"

#include <climits>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>

namespace arrowhip {

// ---- data types -------------------------------------------------------------------------
static const DataType kTypes[] = {
    {Type::NA, 0, "null", "n"},        {Type::BOOL, 1, "bool", "b"},       {Type::UINT8, 8, "uint8", "C"},
    {Type::INT8, 8, "int8", "c"},      {Type::UINT16, 16, "uint16", "S"},  {Type::INT16, 16, "int16", "s"},
    {Type::UINT32, 32, "uint32", "I"}, {Type::INT32, 32, "int32", "i"},    {Type::UINT64, 64, "uint64", "L"},
    {Type::INT64, 64, "int64", "l"},   {Type::FLOAT16, 16, "float16", "e"}, {Type::FLOAT32, 32, "float32", "f"},
    {Type::FLOAT64, 64, "float64", "g"}};
static const DataType kDictType = {Type::DICTIONARY, 32, "dictionary", "i"};
static const DataType kBinaryTypes[] = {{Type::STRING, 32, "utf8", "u"}, {Type::BINARY, 32, "binary", "z"},
                                        {Type::LARGE_STRING, 64, "large_utf8", "U"}, {Type::LARGE_BINARY, 64, "large_binary", "Z"}};

const DataType* TemporalStorage(const std::string& f) {
  if (f.size() < 3 || f[0] != 't') return nullptr;
  auto unit = [&](char c) { return c == 's' || c == 'm' || c == 'u' || c == 'n'; };
  if (f == "tdD") return GetDataType(Type::INT32);                                 // Date32: days
  if (f == "tdm") return GetDataType(Type::INT64);                                 // Date64: milliseconds
  if (f == "tts" || f == "ttm") return GetDataType(Type::INT32);                   // Time32
  if (f == "ttu" || f == "ttn") return GetDataType(Type::INT64);                   // Time64
  if (f[1] == 'D' && f.size() == 3 && unit(f[2])) return GetDataType(Type::INT64);  // Duration
  if (f[1] == 's' && f.size() >= 4 && unit(f[2]) && f[3] == ':') return GetDataType(Type::INT64);  // Timestamp, "tsu:<tz>"
  return nullptr;
}
bool IsBaseBinary(Type id) { return id == Type::STRING || id == Type::BINARY || id == Type::LARGE_STRING || id == Type::LARGE_BINARY; }

static const DataType kFixedPlaceholders[] = {{Type::FIXED_SIZE_BINARY, 0, "fixed_size_binary", "w:0"}, {Type::DECIMAL128, 128, "decimal128", "d:38,0"},
                                              {Type::DECIMAL256, 256, "decimal256", "d:76,0,256"}};
bool IsFixedWidthBinary(Type id) { return id == Type::FIXED_SIZE_BINARY || id == Type::DECIMAL128 || id == Type::DECIMAL256; }
const DataType* FixedWidthBinaryFromFormat(const std::string& f) {
  Type id;
  int bits = 0;
  const char* name;
  if (f.size() > 2 && f[0] == 'w' && f[1] == ':') {   // "w:<bytes>"
    char* end = nullptr;
    const long w = strtol(f.c_str() + 2, &end, 10);
    if (!end || *end || w <= 0 || w > (1 << 20)) return nullptr;
    id = Type::FIXED_SIZE_BINARY; bits = (int)w * 8; name = "fixed_size_binary";
  } else if (f.size() > 2 && f[0] == 'd' && f[1] == ':') {   // "d:<precision>,<scale>[,<bits>]"
    int p = 0, sc = 0, bw = 128;
    const int got = sscanf(f.c_str() + 2, "%d,%d,%d", &p, &sc, &bw);
    if (got < 2 || (bw != 128 && bw != 256)) return nullptr;   // (decimal32 / 64 of newer Arrow versions: not in this reference)
    id = bw == 128 ? Type::DECIMAL128 : Type::DECIMAL256; bits = bw; name = bw == 128 ? "decimal128" : "decimal256";
  } else {
    return nullptr;
  }
  static std::mutex mu;
  static std::map<std::string, std::unique_ptr<DataType>> interned;   // (map nodes do not move: the key's bytes are the format string)
  std::lock_guard<std::mutex> lock(mu);
  auto it = interned.find(f);
  if (it == interned.end()) {
    it = interned.emplace(f, nullptr).first;
    it->second.reset(new DataType{id, bits, name, it->first.c_str()});
  }
  return it->second.get();
}

const DataType* GetDataType(Type id) {
  if (id == Type::DICTIONARY) return &kDictType;
  for (auto& t : kFixedPlaceholders)
    if (t.id == id) return &t;
  for (auto& t : kBinaryTypes)
    if (t.id == id) return &t;
  int i = (int)id;
  if (i < 0 || i > (int)Type::FLOAT64 || id == Type::FLOAT16) return nullptr;
  return &kTypes[i];
}
bool IsSignedInteger(Type id) { return id == Type::INT8 || id == Type::INT16 || id == Type::INT32 || id == Type::INT64; }
bool IsInteger(Type id) { return IsSignedInteger(id) || id == Type::UINT8 || id == Type::UINT16 || id == Type::UINT32 || id == Type::UINT64; }
bool IsFloating(Type id) { return id == Type::FLOAT32 || id == Type::FLOAT64; }
bool IsNumeric(Type id) { return IsInteger(id) || IsFloating(id); }

std::string Status::ToString() const {
  static const char* names[] = {"ok", "invalid", "index error", "not implemented", "type error", "key error", "hip error"};
  if (ok()) return "ok";
  return std::string(names[(int)code]) + ": " + msg;
}

// ---- session / buffers ------------------------------------------------------------------
Buffer::~Buffer() {
  if (owned && dptr && session && session->ctx()) session->Release(dptr, alloc_bytes);
  if (hptr && host_alloc_bytes && session && session->ctx()) session->ReleasePinned(hptr, host_alloc_bytes);
}

Status Session::Create(int device_id, std::shared_ptr<Session>* out) {
  std::shared_ptr<Session> s(new Session());
  int rc = ah_ctx_create(device_id, &s->ctx_);
  if (rc != AH_OK) return Status::Make(StatusCode::Hip, "ah_ctx_create failed (no GPU visible?) — there is no CPU fallback");
  *out = std::move(s);
  return Status::OK();
}
Session::~Session() {
  if (ingest_) ah_ingest_destroy(ingest_);
  TrimPool();
  if (ctx_) ah_ctx_destroy(ctx_);
}
void Session::TrimPool() {
  for (auto& kv : pool_) ah_buf_free(ctx_, kv.second);
  pool_.clear();
  pooled_bytes_ = 0;
  for (auto& kv : pinned_pool_) ah_host_free_pinned(ctx_, kv.second);
  pinned_pool_.clear();
  pinned_pooled_bytes_ = 0;
}
Status Session::AllocatePinned(int64_t nbytes, BufferPtr* out) {
  auto b = std::make_shared<Buffer>();
  Keep(b.get());
  b->size = nbytes;
  b->owned = false;
  const size_t cls = (((size_t)(nbytes > 0 ? nbytes : 1) + 63) & ~(size_t)63) + 64;   // exact sizes: a repeated call asks for the same ones
  auto hit = pinned_pool_.find(cls);
  if (hit != pinned_pool_.end()) {
    b->hptr = hit->second;
    pinned_pooled_bytes_ -= cls;
    pinned_pool_.erase(hit);
  } else {
    AHC_RETURN_NOT_OK(FromStatus(ah_host_alloc_pinned(ctx_, cls, &b->hptr)));
  }
  b->host_alloc_bytes = cls;
  *out = std::move(b);
  return Status::OK();
}
void Session::ReleasePinned(void* hptr, size_t alloc_bytes) {
  if (pinned_pooled_bytes_ + alloc_bytes > ((size_t)8 << 30)) { ah_host_free_pinned(ctx_, hptr); return; }
  pinned_pool_.emplace(alloc_bytes, hptr);
  pinned_pooled_bytes_ += alloc_bytes;
}
Status Session::Ingest(size_t chunk_bytes, ah_ingest** out) {
  if (chunk_bytes == 0) chunk_bytes = (size_t)32 << 20;
  chunk_bytes = (chunk_bytes + 4095) & ~(size_t)4095;
  if (ingest_ && ingest_chunk_ != chunk_bytes) { ah_ingest_destroy(ingest_); ingest_ = nullptr; }
  if (!ingest_) {
    AHC_RETURN_NOT_OK(FromStatus(ah_ingest_create(ctx_, chunk_bytes, 3, &ingest_)));
    ingest_chunk_ = chunk_bytes;
  }
  *out = ingest_;
  return Status::OK();
}
void Session::Release(void* dptr, size_t alloc_bytes) {
  if (alloc_bytes == 0 || pooled_bytes_ + alloc_bytes > pool_cap_) { ah_buf_free(ctx_, dptr); return; }
  pool_.emplace(alloc_bytes, dptr);
  pooled_bytes_ += alloc_bytes;
}
// size classes: powers of two up to 1 MiB, then multiples of 1 MiB — repeated calls on same-shaped batches hit exactly
static size_t SizeClass(size_t nbytes) {
  if (nbytes <= 256) return 256;
  if (nbytes <= ((size_t)1 << 20)) { size_t c = 256; while (c < nbytes) c <<= 1; return c; }
  return (nbytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
}
Status Session::FromStatus(int st) const {
  if (st == AH_OK) return Status::OK();
  StatusCode c = StatusCode::Invalid;
  switch (st) {
    case AH_EINVALID: case AH_EOVERFLOW: c = StatusCode::Invalid; break;  // errOverflow wraps arrow.ErrInvalid
    case AH_EINDEX: c = StatusCode::Index; break;
    case AH_ENOTIMPL: c = StatusCode::NotImplemented; break;
    default: c = StatusCode::Hip; break;
  }
  return Status::Make(c, ah_last_error(ctx_));
}
Status Session::Allocate(int64_t nbytes, BufferPtr* out, bool zero_all) {
  auto b = std::make_shared<Buffer>();
  Keep(b.get());
  b->size = nbytes;
  static const long long cap_env = getenv("ARROWHIP_POOL_BYTES") ? atoll(getenv("ARROWHIP_POOL_BYTES")) : -1;
  if (cap_env >= 0) pool_cap_ = (size_t)cap_env;
  const size_t cls = SizeClass(((size_t)(nbytes > 0 ? nbytes : 1) + 63) & ~(size_t)63);
  auto hit = pool_cap_ ? pool_.find(cls) : pool_.end();
  if (hit != pool_.end()) {
    b->dptr = hit->second;
    pooled_bytes_ -= cls;
    pool_.erase(hit);
  } else {
    int rc = ah_buf_alloc(ctx_, cls, &b->dptr);
    if (rc != AH_OK && !pool_.empty()) {  // out of memory with blocks parked in the pool: give them back and retry
      TrimPool();
      rc = ah_buf_alloc(ctx_, cls, &b->dptr);
    }
    AHC_RETURN_NOT_OK(FromStatus(rc));
  }
  b->alloc_bytes = pool_cap_ ? cls : 0;
  // ctx.Allocate → memory.NewResizableBuffer is zero-filled (executor.go:600 prepareOutput;
  // SURVEY §8a quirk 4): null slots "hold 0" because of this
  size_t padded = ((size_t)(nbytes > 0 ? nbytes : 1) + 63) & ~(size_t)63;
  const size_t keep = zero_all ? 0 : ((size_t)(nbytes > 0 ? nbytes : 0) & ~(size_t)63);
  AHC_RETURN_NOT_OK(FromStatus(ah_memset_async(ctx_, (uint8_t*)b->dptr + keep, 0, padded - keep)));
  *out = std::move(b);
  return Status::OK();
}
Status Session::AllocateBitmap(int64_t nbits, BufferPtr* out) { return Allocate((nbits + 7) / 8, out); }

// ---- spans ------------------------------------------------------------------------------
namespace exec {

void ArraySpan::SetMembers(const ArrayData& d) {
  type = d.type;
  len = d.length;
  nulls = d.null_count;
  offset = d.offset;
  for (int i = 0; i < 3; i++) {
    buffers[i].owner = d.buffers[i];
    buffers[i].buf = d.buffers[i] ? (uint8_t*)d.buffers[i]->dptr : nullptr;
    buffers[i].len = d.buffers[i] ? d.buffers[i]->size : 0;
    buffers[i].self_alloc = false;
  }
  // span.go:322-333: no validity buffer ⇒ null count is 0
  if (buffers[0].buf == nullptr) nulls = 0;
  dictionary = d.dictionary;
  dict_value_type = d.dict_value_type;
  dict_index_type = d.dict_index_type;
}

Status ArraySpan::UpdateNullCount(Session* s, int64_t* out) {
  if (nulls == kUnknownNullCount) {
    if (buffers[0].buf == nullptr) {
      nulls = 0;
    } else {
      int64_t set = 0;
      AHC_RETURN_NOT_OK(s->FromStatus(ah_count_set_bits(s->ctx(), buffers[0].buf, offset, len, &set)));
      nulls = len - set;
    }
  }
  if (out) *out = nulls;
  return Status::OK();
}

ArrayDataPtr ArraySpan::MakeData() const {
  auto d = std::make_shared<ArrayData>();
  d->type = type;
  d->length = len;
  d->null_count = nulls;
  d->offset = offset;
  d->buffers[0] = buffers[0].owner;
  d->buffers[1] = buffers[1].owner;
  d->buffers[2] = buffers[2].owner;
  // span.go:243-247: a known-zero null count drops the validity buffer
  if (nulls == 0) d->buffers[0] = nullptr;
  if (type && type->id == Type::DICTIONARY) {
    d->dictionary = dictionary;
    d->dict_value_type = dict_value_type;
    d->dict_index_type = dict_index_type;
  }
  return d;
}

bool KernelSignature::MatchesInputs(const std::vector<const DataType*>& types) const {
  if (types.size() != in_types.size()) return false;
  for (size_t i = 0; i < types.size(); i++)
    if (!types[i] || types[i]->id != in_types[i]) return false;
  return true;
}

}  // namespace exec

namespace compute {

// ---- Function base ----------------------------------------------------------------------
Status Function::CheckArity(size_t nargs) const {
  // functions.go:130-146
  if (arity_.IsVarArgs && (int)nargs < arity_.NArgs)
    return Status::Make(StatusCode::Invalid, "varargs function '" + name_ + "' needs at least " + std::to_string(arity_.NArgs) +
                                                 " arguments, but only " + std::to_string(nargs) + " passed");
  if (!arity_.IsVarArgs && (int)nargs != arity_.NArgs)
    return Status::Make(StatusCode::Invalid, "function '" + name_ + "' accepts " + std::to_string(arity_.NArgs) +
                                                 " arguments but " + std::to_string(nargs) + " passed");
  return Status::OK();
}

static std::string TypesToString(const std::vector<const DataType*>& types) {
  std::string s = "(";
  for (size_t i = 0; i < types.size(); i++) s += std::string(i ? ", " : "") + (types[i] ? types[i]->name : "?");
  return s + ")";
}

static Status ArgTypes(const std::vector<Datum>& args, std::vector<const DataType*>* types) {
  for (auto& a : args) {
    if (a.kind == DatumKind::None) return Status::Make(StatusCode::Invalid, "invalid datum");
    if (a.kind == DatumKind::Record) return Status::Make(StatusCode::NotImplemented, "record batch arguments are only understood by filter, take, sort_indices and sort");
    types->push_back(a.type());
  }
  return Status::OK();
}

// ---- ScalarFunction ------------------------------------------------------------------------
Status ScalarFunction::AddKernel(exec::ScalarKernel k) {
  if ((int)k.sig.in_types.size() != arity_.NArgs)  // functions.go:246-252 checkArity on the signature
    return Status::Make(StatusCode::Invalid, "kernel signature does not match the arity of function '" + name_ + "'");
  kernels_.push_back(std::move(k));
  return Status::OK();
}
std::vector<exec::ScalarKernel*> ScalarFunction::Kernels() {
  std::vector<exec::ScalarKernel*> out;
  for (auto& k : kernels_) out.push_back(&k);
  return out;
}
Status ScalarFunction::DispatchExact(const std::vector<const DataType*>& types, const exec::ScalarKernel** out) const {
  for (auto& k : kernels_)  // first matching signature wins (functions.go:209-213)
    if (k.sig.MatchesInputs(types)) { *out = &k; return Status::OK(); }
  return Status::Make(StatusCode::NotImplemented, "function '" + name_ + "' has no kernel matching input types " + TypesToString(types));
}

// propagateNulls (executor.go:237-349), preallocated-output branch (the scalar executor here
// always preallocates a validity bitmap unless it can be elided, setupPrealloc :658-702)
static Status PropagateNulls(Session* s, const exec::ExecSpan& batch, exec::ArraySpan* out) {
  ah_ctx* c = s->ctx();
  std::vector<const exec::ArraySpan*> with_nulls;
  bool all_null = false;
  for (auto& v : batch.values) {
    if (v.IsScalar()) {  // getNullGen :190-213
      if (!v.scalar->valid) all_null = true;
      continue;
    }
    const exec::ArraySpan& a = v.array;
    if (a.nulls == 0 || a.buffers[0].buf == nullptr) continue;  // nullGenAllValid
    if (a.nulls == a.len) all_null = true;
    with_nulls.push_back(&a);
  }
  uint8_t* ob = out->buffers[0].buf;
  if (all_null) {
    out->nulls = out->len;
    return s->FromStatus(ah_set_bits_to(c, ob, out->offset, out->len, 0));
  }
  out->nulls = kUnknownNullCount;
  if (with_nulls.empty()) {
    out->nulls = 0;
    return s->FromStatus(ah_set_bits_to(c, ob, out->offset, out->len, 1));
  }
  if (with_nulls.size() == 1) {
    out->nulls = with_nulls[0]->nulls;
    return s->FromStatus(ah_copy_bitmap(c, with_nulls[0]->buffers[0].buf, with_nulls[0]->offset, out->len, ob, out->offset, 0));
  }
  AHC_RETURN_NOT_OK(s->FromStatus(ah_bitmap_op(c, AH_BIT_AND, with_nulls[0]->buffers[0].buf, with_nulls[0]->offset,
                                              with_nulls[1]->buffers[0].buf, with_nulls[1]->offset, ob, out->offset, out->len)));
  for (size_t i = 2; i < with_nulls.size(); i++)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_bitmap_op(c, AH_BIT_AND, ob, out->offset, with_nulls[i]->buffers[0].buf,
                                                with_nulls[i]->offset, ob, out->offset, out->len)));
  return Status::OK();
}

// all-scalar calls: arrow-go promotes scalars to length-1 arrays and unboxes the result
// (executor.go:462-496 execute → haveAllScalars; exec.go:172 WrapResults).  Here: upload 1
// element, run the same array kernel, download 1 element.
static Status ScalarToArray(Session* s, const Scalar& sc, ArrayDataPtr* out) {
  auto d = std::make_shared<ArrayData>();
  d->type = sc.type;
  d->length = 1;
  d->null_count = sc.valid ? 0 : 1;
  AHC_RETURN_NOT_OK(s->Allocate(8, &d->buffers[1]));
  AHC_RETURN_NOT_OK(s->FromStatus(ah_upload_async(s->ctx(), d->buffers[1]->dptr, sc.value, 8)));
  if (!sc.valid) AHC_RETURN_NOT_OK(s->AllocateBitmap(1, &d->buffers[0]));  // zero = null
  AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));
  *out = d;
  return Status::OK();
}
static Status ArrayToScalar(Session* s, const ArrayData& a, ScalarPtr* out) {
  auto sc = std::make_shared<Scalar>();
  sc->type = a.type;
  uint8_t tmp[8] = {0};
  int w = a.type->bit_width == 1 ? 1 : a.type->bit_width / 8;
  AHC_RETURN_NOT_OK(s->FromStatus(ah_download_async(s->ctx(), tmp, a.buffers[1]->dptr, (size_t)w)));
  uint8_t vbyte = 1;
  if (a.buffers[0]) AHC_RETURN_NOT_OK(s->FromStatus(ah_download_async(s->ctx(), &vbyte, a.buffers[0]->dptr, 1)));
  AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));
  sc->valid = (vbyte & 1) != 0 && a.null_count != 1;
  if (a.type->id == Type::BOOL) tmp[0] &= 1;
  memcpy(sc->value, tmp, 8);
  *out = sc;
  return Status::OK();
}

Status ScalarFunction::Execute(ExecCtx* ctx, const FunctionOptions* opts, const std::vector<Datum>& args_in, Datum* out) {
  AHC_RETURN_NOT_OK(CheckArity(args_in.size()));
  if (!flipped_of.empty()) {  // scalar_compare.go:73-99 flippedCompare
    std::vector<Datum> swapped = {args_in[1], args_in[0]};
    return CallFunction(ctx, flipped_of, opts, swapped, out);
  }
  Session* s = ctx->session;
  if (!s) return Status::Make(StatusCode::Invalid, "ExecCtx has no device session");
  std::vector<Datum> args = args_in;
  std::vector<const DataType*> types;
  AHC_RETURN_NOT_OK(ArgTypes(args, &types));
  const exec::ScalarKernel* kernel = nullptr;
  AHC_RETURN_NOT_OK(DispatchBest(&types, &kernel));
  // cast arguments if necessary (execInternal, exec.go:105-114): implicit casts are SAFE casts
  for (size_t i = 0; i < args.size(); i++) {
    if (args[i].type()->id == types[i]->id) continue;
    Datum casted;
    AHC_RETURN_NOT_OK(CastDatum(ctx, args[i], CastOptions::Safe(types[i]), &casted));
    args[i] = casted;
  }

  // inferBatchLength (executor.go:352-390): arrays must agree; all scalars → length 1
  int64_t length = -1;
  bool all_scalar = true;
  for (auto& a : args) {
    if (a.kind != DatumKind::Array) continue;
    all_scalar = false;
    if (length < 0) length = a.array->length;
    else if (length != a.array->length)
      return Status::Make(StatusCode::Invalid, "array arguments must all be the same length");
  }
  if (all_scalar) {
    for (auto& a : args) {
      ArrayDataPtr arr;
      AHC_RETURN_NOT_OK(ScalarToArray(s, *a.scalar, &arr));
      a = Datum::Of(arr);
    }
    length = 1;
  }

  exec::ExecSpan span;
  span.len = length;
  for (auto& a : args) {
    exec::ExecValue v;
    if (a.kind == DatumKind::Array) v.array.SetMembers(*a.array);
    else v.scalar = a.scalar.get();
    span.values.push_back(std::move(v));
  }

  exec::ExecResult res;
  res.type = kernel->sig.out_is_first_input ? types[0] : GetDataType(kernel->sig.out_type);
  res.len = length;
  res.offset = 0;
  // setupPrealloc (executor.go:658-702)
  bool elide_validity = false, prealloc_validity = false;
  switch (kernel->null_handling) {
    case exec::NullHandling::NullComputedPrealloc: prealloc_validity = true; break;
    case exec::NullHandling::NullIntersection: {
      elide_validity = true;
      for (auto& v : span.values) {
        bool all_valid = v.IsScalar() ? v.scalar->valid : (v.array.nulls == 0 || v.array.buffers[0].buf == nullptr);
        elide_validity = elide_validity && all_valid;
      }
      prealloc_validity = !elide_validity;
      break;
    }
    case exec::NullHandling::NullNoOutput: elide_validity = true; break;
    default: break;
  }
  if (prealloc_validity) {
    BufferPtr b;
    AHC_RETURN_NOT_OK(s->AllocateBitmap(length, &b));
    res.buffers[0].WrapBuffer(b);
  }
  if (kernel->mem_alloc == exec::MemAlloc::MemPrealloc) {
    BufferPtr b;
    int64_t nbytes = res.type->bit_width == 1 ? (length + 7) / 8 : length * (res.type->bit_width / 8);
    // every scalar kernel with a byte-addressed output writes all `length` slots (null slots included: the NotNull
    // ops store their zero, a null scalar operand memsets); bitmap outputs keep the full zero-fill, their kernels
    // read-modify-write boundary words
    AHC_RETURN_NOT_OK(s->Allocate(nbytes, &b, /*zero_all=*/res.type->bit_width == 1));
    res.buffers[1].WrapBuffer(b);
  }
  // executeSingleSpan (executor.go:644-656)
  if (kernel->null_handling == exec::NullHandling::NullIntersection) {
    if (!elide_validity) AHC_RETURN_NOT_OK(PropagateNulls(s, span, &res));
    else res.nulls = 0;
  } else if (kernel->null_handling == exec::NullHandling::NullNoOutput) {
    res.nulls = 0;
  }
  exec::KernelCtx kctx;
  kctx.session = s;
  kctx.state = opts ? opts : default_opts_;
  kctx.kernel_data = kernel->data.get();
  AHC_RETURN_NOT_OK(kernel->exec_fn(&kctx, span, &res));

  ArrayDataPtr result = res.MakeData();
  if (all_scalar) {
    ScalarPtr sc;
    AHC_RETURN_NOT_OK(ArrayToScalar(s, *result, &sc));
    *out = Datum::Of(sc);
  } else {
    *out = Datum::Of(result);
  }
  return Status::OK();
}

// commonNumeric (compute/utils.go:178-240)
const DataType* CommonNumeric(const std::vector<const DataType*>& types) {
  for (auto* t : types)
    if (!IsInteger(t->id) && !IsFloating(t->id)) return nullptr;
  for (auto* t : types) if (t->id == Type::FLOAT64) return GetDataType(Type::FLOAT64);
  for (auto* t : types) if (t->id == Type::FLOAT32) return GetDataType(Type::FLOAT32);
  int max_signed = 0, max_unsigned = 0;
  for (auto* t : types) {
    if (IsSignedInteger(t->id)) max_signed = std::max(max_signed, t->bit_width);
    else max_unsigned = std::max(max_unsigned, t->bit_width);
  }
  if (max_signed == 0) {
    if (max_unsigned >= 64) return GetDataType(Type::UINT64);
    if (max_unsigned == 32) return GetDataType(Type::UINT32);
    if (max_unsigned == 16) return GetDataType(Type::UINT16);
    return GetDataType(Type::UINT8);
  }
  if (max_signed <= max_unsigned) {  // bitutil.NextPowerOf2(maxWidthUnsigned + 1)
    int w = 8;
    while (w < max_unsigned + 1) w *= 2;
    max_signed = w;
  }
  if (max_signed >= 64) return GetDataType(Type::INT64);
  if (max_signed == 32) return GetDataType(Type::INT32);
  if (max_signed == 16) return GetDataType(Type::INT16);
  return GetDataType(Type::INT8);
}

int ScalarFunction::NumKernels() const {
  if (!flipped_of.empty() && kernels_.empty())
    if (auto* base = GetFunctionRegistry()->GetFunction(flipped_of)) return base->NumKernels();
  return (int)kernels_.size();
}

Status ScalarFunction::DispatchBest(std::vector<const DataType*>* types, const exec::ScalarKernel** out) const {
  if (!flipped_of.empty() && types->size() == 2) {
    // "less" / "less_equal" hold the kernels of "greater" / "greater_equal" with the arguments exchanged (makeFlippedCompare,
    // scalar_compare.go:73-99, 134-135): what the reference's DispatchBest reports for (l, r) is what the base reports for (r, l)
    auto* base = dynamic_cast<const ScalarFunction*>(GetFunctionRegistry()->GetFunction(flipped_of));
    if (!base) return Status::Make(StatusCode::KeyError, "function '" + flipped_of + "' not found");
    std::vector<const DataType*> sw = {(*types)[1], (*types)[0]};
    Status fs = base->DispatchBest(&sw, out);
    if (!fs.ok()) {   // under this function's name, the arguments in the caller's order
      std::string m = fs.msg;
      const std::string from = "function '" + flipped_of + "'";
      const size_t at = m.find(from);
      if (at != std::string::npos) m.replace(at, from.size(), "function '" + name_ + "'");
      return fs.code == StatusCode::NotImplemented ? Status::Make(StatusCode::NotImplemented, "function '" + name_ + "' has no kernel matching input types " + TypesToString(*types))
                                                   : Status::Make(fs.code, m);
    }
    (*types)[0] = sw[1];
    (*types)[1] = sw[0];
    return fs;
  }
  Status st = DispatchExact(*types, out);
  if (!st.ok() && promote_to_float && types->size() == 1 && IsInteger((*types)[0]->id)) {
    std::vector<const DataType*> promoted{GetDataType(Type::FLOAT64)};
    Status st2 = DispatchExact(promoted, out);
    if (st2.ok()) { *types = promoted; return st2; }
  }
  if (st.ok() || !promote_numeric || types->size() != 2) return st;
  if (const DataType* common = CommonNumeric(*types)) {
    std::vector<const DataType*> promoted(types->size(), common);
    Status st2 = DispatchExact(promoted, out);
    if (st2.ok()) { *types = promoted; return st2; }
  }
  return st;
}

Status CastDatum(ExecCtx* ctx, const Datum& in, const CastOptions& opts, Datum* out) {
  return CallFunction(ctx, "cast", &opts, {in}, out);
}

// ---- VectorFunction ------------------------------------------------------------------------
Status VectorFunction::AddKernel(exec::VectorKernel k) {
  if ((int)k.sig.in_types.size() != arity_.NArgs)
    return Status::Make(StatusCode::Invalid, "kernel signature does not match the arity of function '" + name_ + "'");
  kernels_.push_back(std::move(k));
  return Status::OK();
}
Status VectorFunction::DispatchExact(const std::vector<const DataType*>& types, const exec::VectorKernel** out) const {
  for (auto& k : kernels_)
    if (k.sig.MatchesInputs(types)) { *out = &k; return Status::OK(); }
  return Status::Make(StatusCode::NotImplemented, "function '" + name_ + "' has no kernel matching input types " + TypesToString(types));
}
Status VectorFunction::Execute(ExecCtx* ctx, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out) {
  AHC_RETURN_NOT_OK(CheckArity(args.size()));
  Session* s = ctx->session;
  if (!s) return Status::Make(StatusCode::Invalid, "ExecCtx has no device session");
  std::vector<const DataType*> types;
  AHC_RETURN_NOT_OK(ArgTypes(args, &types));
  const exec::VectorKernel* kernel = nullptr;
  AHC_RETURN_NOT_OK(DispatchExact(types, &kernel));
  // vectorExecutor.Execute (executor.go:896-960): one span (the whole array); nothing
  // preallocated (VectorKernel defaults NullComputedNoPrealloc + MemNoPrealloc)
  exec::ExecSpan span;
  bool all_scalar = !args.empty();
  for (auto& a : args) all_scalar = all_scalar && a.kind == DatumKind::Scalar;
  std::vector<ArrayDataPtr> promoted;  // checkIfAllScalar → PromoteExecSpanScalars (executor.go:951-954)
  for (auto& a : args) {
    if (a.kind != DatumKind::Array && !all_scalar)
      return Status::Make(StatusCode::NotImplemented, "vector function '" + name_ + "' needs array arguments");
    exec::ExecValue v;
    if (all_scalar) {
      ArrayDataPtr arr;
      AHC_RETURN_NOT_OK(ScalarToArray(s, *a.scalar, &arr));
      promoted.push_back(arr);
      v.array.SetMembers(*arr);
    } else
    v.array.SetMembers(*a.array);
    span.values.push_back(std::move(v));
  }
  span.len = span.values.empty() ? 0 : span.values[0].array.len;
  exec::ExecResult res;
  res.type = kernel->sig.out_is_first_input ? types[0] : GetDataType(kernel->sig.out_type);
  exec::KernelCtx kctx;
  kctx.session = s;
  kctx.state = opts ? opts : default_opts_;
  AHC_RETURN_NOT_OK(kernel->exec_fn(&kctx, span, &res));
  ArrayDataPtr result = res.MakeData();
  if (kernel->output_is_dictionary) {
    result->type = GetDataType(Type::DICTIONARY);
    result->dict_value_type = types[0];
    result->dictionary = res.dictionary;
  }
  *out = Datum::Of(result);
  return Status::OK();
}

Status MetaFunction::Execute(ExecCtx* ctx, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out) {
  AHC_RETURN_NOT_OK(CheckArity(args.size()));  // functions.go:417-428
  return impl_(ctx, opts ? opts : default_opts_, args, out);
}

// ---- registry ---------------------------------------------------------------------------
static std::mutex g_reg_mu;

bool FunctionRegistry::CanAddFunction(const std::shared_ptr<Function>& fn, bool allow_overwrite) const {
  if (parent_ && !parent_->CanAddFunction(fn, allow_overwrite)) return false;  // registry.go:83-95
  return allow_overwrite || fns_.find(fn->Name()) == fns_.end();
}
Status FunctionRegistry::AddFunction(std::shared_ptr<Function> fn, bool allow_overwrite) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  if (!CanAddFunction(fn, allow_overwrite))
    return Status::Make(StatusCode::KeyError, "already have a function registered with name: " + fn->Name());  // registry.go:104-111
  fns_[fn->Name()] = std::move(fn);
  return Status::OK();
}
Status FunctionRegistry::AddAlias(const std::string& target, const std::string& source) {
  Function* f = GetFunction(source);
  if (!f) return Status::Make(StatusCode::KeyError, "no function registered with name: " + source);
  std::lock_guard<std::mutex> lk(g_reg_mu);
  for (auto& kv : fns_)
    if (kv.second.get() == f) { fns_[target] = kv.second; return Status::OK(); }
  return Status::Make(StatusCode::KeyError, "alias source lives in a parent registry: " + source);
}
Function* FunctionRegistry::GetFunction(const std::string& name) const {
  auto it = fns_.find(name);
  if (it != fns_.end()) return it->second.get();
  return parent_ ? parent_->GetFunction(name) : nullptr;  // registry.go:120-131
}
std::vector<std::string> FunctionRegistry::GetFunctionNames() const {
  std::vector<std::string> names = parent_ ? parent_->GetFunctionNames() : std::vector<std::string>();
  for (auto& kv : fns_) names.push_back(kv.first);
  return names;
}
int FunctionRegistry::NumFunctions() const { return (int)fns_.size() + (parent_ ? parent_->NumFunctions() : 0); }

FunctionRegistry* GetFunctionRegistry() {
  static FunctionRegistry* reg = [] {
    auto* r = new FunctionRegistry();
    RegisterScalarArithmetic(r);
    RegisterScalarComparisons(r);
    RegisterScalarBoolean(r);
    RegisterVectorSelection(r);
    RegisterVectorHash(r);
    RegisterVectorCumulative(r);
    RegisterScalarCast(r);
    RegisterScalarSetLookup(r);
    RegisterVectorSort(r);
    RegisterFusedExtensions(r);
    return r;
  }();
  return reg;
}
std::unique_ptr<FunctionRegistry> NewChildRegistry(FunctionRegistry* parent) {
  return std::unique_ptr<FunctionRegistry>(new FunctionRegistry(parent));
}

// ---- chunked arguments -------------------------------------------------------------------------
// The reference walks chunked inputs span by span on the host (iterateExecSpans, executor.go:750-870:
// spans end wherever any argument's chunk ends) and calls the kernel once per span.  A launch per small
// chunk is the wrong shape for this device, so the chunks of every chunked argument are laid end to end
// in HBM (Concatenate: device-to-device copies), the kernel runs ONCE, and the result is cut back into
// zero-copy slices at exactly the boundaries the reference's spans would have produced.  Row for row
// the results are the same: every kernel on this path is either element-wise or defined on the
// logical concatenation (unique, dictionary_encode, cumulative_sum, sort_indices carry state across
// chunks in the reference too).
ArrayDataPtr SliceData(const ArrayDataPtr& a, int64_t off, int64_t len) {
  auto d = std::make_shared<ArrayData>(*a);
  d->offset = a->offset + off;
  d->length = len;
  d->null_count = (a->null_count == 0 || !a->buffers[0]) ? 0 : (off == 0 && len == a->length ? a->null_count : kUnknownNullCount);
  return d;
}

// array.Equal for two dictionaries on the device (what dictionaryHashState.Append asks before it unifies, vector_hash.go:527-531):
// same type and length, equal validity, equal values (fixed width) or equal offsets and bytes (binary).  Compares through the
// compare + popcount kernels; only unsliced arrays (offset 0), anything else reports "not equal".
static Status DeviceArrayEqual(Session* s, const ArrayData& a, const ArrayData& b, bool* eq) {
  *eq = false;
  if (a.type->id != b.type->id || a.length != b.length || a.offset != 0 || b.offset != 0) return Status::OK();
  const int64_t n = a.length;
  if (n == 0) { *eq = true; return Status::OK(); }
  const bool av = a.buffers[0] && a.null_count != 0, bv = b.buffers[0] && b.null_count != 0;
  if (av != bv) return Status::OK();
  BufferPtr bits;
  auto same_bytes = [&](int type, const void* x, const void* y, int64_t count, bool* same) -> Status {
    if (count == 0) { *same = true; return Status::OK(); }
    AHC_RETURN_NOT_OK(s->AllocateBitmap(count, &bits));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_comparison(s->ctx(), AH_CMP_NE, AH_SHAPE_AA, type, x, y, (uint8_t*)bits->dptr, count, 0)));
    int64_t diff = 0;
    AHC_RETURN_NOT_OK(s->FromStatus(ah_count_set_bits(s->ctx(), (const uint8_t*)bits->dptr, 0, count, &diff)));
    *same = diff == 0;
    return Status::OK();
  };
  bool same = true;
  if (av) {
    BufferPtr x;
    AHC_RETURN_NOT_OK(s->AllocateBitmap(n, &x));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_bitmap_op(s->ctx(), AH_BIT_XOR, (const uint8_t*)a.buffers[0]->dptr, 0, (const uint8_t*)b.buffers[0]->dptr, 0, (uint8_t*)x->dptr, 0, n)));
    int64_t diff = 0;
    AHC_RETURN_NOT_OK(s->FromStatus(ah_count_set_bits(s->ctx(), (const uint8_t*)x->dptr, 0, n, &diff)));
    if (diff) return Status::OK();
  }
  const int w = a.type->bit_width / 8;
  if (IsBaseBinary(a.type->id)) {
    AHC_RETURN_NOT_OK(same_bytes(w == 4 ? AH_INT32 : AH_INT64, a.buffers[1]->dptr, b.buffers[1]->dptr, n + 1, &same));
    if (!same) return Status::OK();
    uint8_t last[8] = {0};
    AHC_RETURN_NOT_OK(s->FromStatus(ah_download_async(s->ctx(), last, (const uint8_t*)a.buffers[1]->dptr + n * w, (size_t)w)));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));
    int64_t bytes = 0;
    if (w == 4) { int32_t v; memcpy(&v, last, 4); bytes = v; } else memcpy(&bytes, last, 8);
    AHC_RETURN_NOT_OK(same_bytes(AH_UINT8, a.buffers[2]->dptr, b.buffers[2]->dptr, bytes, &same));
  } else if (a.type->bit_width == 1) {
    return Status::OK();  // boolean dictionaries: not compared
  } else {
    // raw bit patterns: compare as unsigned integers of the element width (NaN payloads and ±0 distinguish, as bytes.Equal would)
    const int t = w == 1 ? AH_UINT8 : w == 2 ? AH_UINT16 : w == 4 ? AH_UINT32 : AH_UINT64;
    AHC_RETURN_NOT_OK(same_bytes(t, a.buffers[1]->dptr, b.buffers[1]->dptr, n, &same));
  }
  *eq = same;   // values under nulls may differ between producers; validity already matched
  return Status::OK();
}

Status Concatenate(Session* s, const std::vector<ArrayDataPtr>& chunks, const DataType* type, ArrayDataPtr* out) {
  if (chunks.size() == 1) { *out = chunks[0]; return Status::OK(); }
  auto d = std::make_shared<ArrayData>();
  d->type = type;
  d->logical = chunks.empty() ? std::string() : chunks[0]->logical;
  int64_t total = 0;
  bool nulls = false;
  for (auto& c : chunks) {
    if (c->logical != d->logical) return Status::Make(StatusCode::Invalid, "arrays to be concatenated must be identically typed, but " + d->logical + " and " + c->logical + " were encountered.");
    if (c->type->id != type->id) return Status::Make(StatusCode::Invalid, "arrays to be concatenated must be identically typed, but " + std::string(type->name) + " and " + c->type->name + " were encountered.");  // concat.go:53-56
    total += c->length;
    nulls = nulls || (c->buffers[0] && c->null_count != 0);
  }
  d->length = total;
  d->null_count = nulls ? kUnknownNullCount : 0;
  std::vector<ArrayDataPtr> unified;  // chunks re-indexed against one dictionary, when theirs differ
  const std::vector<ArrayDataPtr>* src = &chunks;
  if (type->id == Type::DICTIONARY) {
    bool same = true;
    for (auto& c : chunks) {
      if (c->dictionary == chunks[0]->dictionary) continue;
      bool eq = false;
      if (c->dictionary && chunks[0]->dictionary) AHC_RETURN_NOT_OK(DeviceArrayEqual(s, *c->dictionary, *chunks[0]->dictionary, &eq));
      same = same && eq;
    }
    ArrayDataPtr dict = chunks[0]->dictionary;
    if (!same) {
      // array.NewDictionaryUnifier + TransposeDictIndices (arrow/array/dictionary.go:1380-1500, concat.go:600-640), on the
      // device: the dictionaries laid end to end and dictionary-encoded with nulls encoded ARE the unifier's memo table —
      // ids = the transposition maps, its dictionary = the unified values in first-seen order; each chunk's indices are
      // then gathered through its slice of the map
      std::vector<ArrayDataPtr> dvals;
      for (auto& c : chunks) {
        if (!c->dictionary || c->dictionary->type->id != chunks[0]->dictionary->type->id)
          return Status::Make(StatusCode::Invalid, "dictionary type different from unifier");  // dictionary.go:1416-1418
        dvals.push_back(c->dictionary);
      }
      ArrayDataPtr all;
      AHC_RETURN_NOT_OK(Concatenate(s, dvals, dvals[0]->type, &all));
      ExecCtx ectx{GetFunctionRegistry(), s};
      DictionaryEncodeOptions eo;
      eo.NullEncoding = NullEncodingEncode;
      Datum enc;
      AHC_RETURN_NOT_OK(CallFunction(&ectx, "dictionary_encode", &eo, {Datum::Of(all)}, &enc));
      dict = enc.array->dictionary;
      auto map_all = std::make_shared<ArrayData>(*enc.array);
      map_all->type = GetDataType(Type::INT32);
      map_all->dictionary = nullptr;
      TakeOptions no_check;
      no_check.BoundsCheck = false;
      int64_t o = 0;
      for (auto& c : chunks) {
        auto idx = std::make_shared<ArrayData>(*c);   // the chunk's indices as a plain int32 array
        idx->type = GetDataType(Type::INT32);
        idx->dictionary = nullptr;
        Datum moved;
        AHC_RETURN_NOT_OK(CallFunction(&ectx, "array_take", &no_check, {Datum::Of(SliceData(map_all, o, c->dictionary->length)), Datum::Of(idx)}, &moved));
        auto nc = std::make_shared<ArrayData>(*moved.array);
        nc->type = type;
        unified.push_back(nc);
        o += c->dictionary->length;
      }
      src = &unified;
    }
    d->dictionary = dict;
    d->dict_value_type = chunks[0]->dict_value_type;
    d->dict_index_type = chunks[0]->dict_index_type;
  }
  const std::vector<ArrayDataPtr>& chunks_ = *src;
  auto bitmap = [&](int which, BufferPtr* dst) -> Status {  // validity (which = 0) or boolean data (which = 1), bit by bit offset
    AHC_RETURN_NOT_OK(s->AllocateBitmap(total, dst));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), (*dst)->dptr, 0, (size_t)((total + 7) / 8))));
    int64_t pos = 0;
    for (auto& c : chunks_) {
      if (c->length == 0) continue;
      if (c->buffers[which] && !(which == 0 && c->null_count == 0))
        AHC_RETURN_NOT_OK(s->FromStatus(ah_copy_bitmap(s->ctx(), (const uint8_t*)c->buffers[which]->dptr, c->offset, c->length, (uint8_t*)(*dst)->dptr, pos, 0)));
      else
        AHC_RETURN_NOT_OK(s->FromStatus(ah_set_bits_to(s->ctx(), (uint8_t*)(*dst)->dptr, pos, c->length, 1)));
      pos += c->length;
    }
    return Status::OK();
  };
  if (nulls) AHC_RETURN_NOT_OK(bitmap(0, &d->buffers[0]));
  const int w = type->bit_width / 8;
  if (type->bit_width == 1) {
    AHC_RETURN_NOT_OK(bitmap(1, &d->buffers[1]));
  } else if (IsBaseBinary(type->id)) {
    // value ranges: first and last offset of every chunk (concat.go:182-200 / 300-322 concatOffsets)
    std::vector<int64_t> first(chunks_.size(), 0), last(chunks_.size(), 0);
    std::vector<uint8_t> host(chunks_.size() * 16, 0);
    for (size_t i = 0; i < chunks_.size(); i++) {
      auto& c = chunks_[i];
      if (c->length == 0 || !c->buffers[1]) continue;
      const uint8_t* o = (const uint8_t*)c->buffers[1]->dptr;
      AHC_RETURN_NOT_OK(s->FromStatus(ah_download_async(s->ctx(), &host[i * 16], o + c->offset * w, (size_t)w)));
      AHC_RETURN_NOT_OK(s->FromStatus(ah_download_async(s->ctx(), &host[i * 16 + 8], o + (c->offset + c->length) * w, (size_t)w)));
    }
    AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));
    int64_t bytes = 0;
    for (size_t i = 0; i < chunks_.size(); i++) {
      if (w == 4) { int32_t a, b; memcpy(&a, &host[i * 16], 4); memcpy(&b, &host[i * 16 + 8], 4); first[i] = a; last[i] = b; }
      else { memcpy(&first[i], &host[i * 16], 8); memcpy(&last[i], &host[i * 16 + 8], 8); }
      bytes += last[i] - first[i];
    }
    if (w == 4 && bytes > INT32_MAX) return Status::Make(StatusCode::Invalid, "offset overflow while concatenating arrays");  // concat.go:197
    AHC_RETURN_NOT_OK(s->Allocate((total + 1) * w, &d->buffers[1]));
    AHC_RETURN_NOT_OK(s->Allocate(bytes, &d->buffers[2]));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), d->buffers[1]->dptr, 0, (size_t)w)));
    int64_t pos = 0, base = 0;
    for (size_t i = 0; i < chunks_.size(); i++) {
      auto& c = chunks_[i];
      if (c->length == 0) continue;
      const int64_t delta = base - first[i];  // out_offsets[pos + 1 + j] = in_offsets[offset + 1 + j] + delta
      const int32_t delta32 = (int32_t)delta;
      AHC_RETURN_NOT_OK(s->FromStatus(ah_arithmetic_arr_scalar(s->ctx(), w == 4 ? AH_INT32 : AH_INT64, AH_OP_ADD,
                                                               (const uint8_t*)c->buffers[1]->dptr + (c->offset + 1) * w,
                                                               w == 4 ? (const void*)&delta32 : (const void*)&delta,
                                                               (uint8_t*)d->buffers[1]->dptr + (pos + 1) * w, c->length)));
      if (last[i] > first[i])
        AHC_RETURN_NOT_OK(s->FromStatus(ah_copy_async(s->ctx(), (uint8_t*)d->buffers[2]->dptr + base, (const uint8_t*)c->buffers[2]->dptr + first[i],
                                                      (size_t)(last[i] - first[i]))));
      pos += c->length;
      base += last[i] - first[i];
    }
  } else {
    AHC_RETURN_NOT_OK(s->Allocate(total * w, &d->buffers[1]));
    int64_t pos = 0;
    for (auto& c : chunks_) {
      if (c->length == 0) continue;
      AHC_RETURN_NOT_OK(s->FromStatus(ah_copy_async(s->ctx(), (uint8_t*)d->buffers[1]->dptr + pos * w, (const uint8_t*)c->buffers[1]->dptr + c->offset * w,
                                                    (size_t)(c->length * w))));
      pos += c->length;
    }
  }
  *out = d;
  return Status::OK();
}

namespace {

// cut `a` at the given row counts; empty pieces are dropped (WrapResults, executor.go:521-582 / 1000-1080)
Datum Rechunk(const ArrayDataPtr& a, const std::vector<int64_t>& lens, bool keep_one_if_empty) {
  std::vector<ArrayDataPtr> pieces;
  int64_t pos = 0;
  for (int64_t n : lens) {
    if (n > 0) pieces.push_back(SliceData(a, pos, n));
    pos += n;
  }
  if (pieces.empty() && keep_one_if_empty) pieces.push_back(SliceData(a, 0, 0));
  Datum d = Datum::OfChunks(a->type, std::move(pieces));
  return d;
}

// span lengths of iterateExecSpans: the union of the chunk boundaries of all chunked arguments
Status SpanLengths(const std::vector<Datum>& args, std::vector<int64_t>* lens, int64_t* length) {
  *length = -1;
  for (auto& a : args) {
    if (!a.IsArrayLike()) continue;
    if (*length < 0) *length = a.Len();
    else if (*length != a.Len()) return Status::Make(StatusCode::Invalid, "array arguments must all be the same length");  // executor.go:352-390
  }
  std::vector<int64_t> cuts;
  for (auto& a : args) {
    if (a.kind != DatumKind::Chunked) continue;
    int64_t pos = 0;
    for (auto& c : a.chunks) { pos += c->length; cuts.push_back(pos); }
  }
  std::sort(cuts.begin(), cuts.end());
  cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
  lens->clear();
  int64_t prev = 0;
  for (int64_t c : cuts) { lens->push_back(c - prev); prev = c; }
  return Status::OK();
}

Status ExecuteChunked(ExecCtx* ctx, Function* fn, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out) {
  Session* s = ctx ? ctx->session : nullptr;
  if (!s) return Status::Make(StatusCode::Invalid, "ExecCtx has no device session");
  if (fn->Kind() == FuncKind::Meta) return fn->Execute(ctx, opts, args, out);  // filter / take / cast / sort forward to their array_* twins
  std::vector<Datum> flat = args;
  for (auto& a : flat) {
    if (a.kind != DatumKind::Chunked) continue;
    ArrayDataPtr whole;
    if (a.chunks.empty()) {  // no chunks at all: a zero-length array of the type (MakeArrayOfNull(…, 0), selection.go:270)
      whole = std::make_shared<ArrayData>();
      whole->type = a.chunked_type;
      whole->null_count = 0;
      AHC_RETURN_NOT_OK(s->Allocate(IsBaseBinary(a.chunked_type->id) ? a.chunked_type->bit_width / 8 : 0, &whole->buffers[1]));
      if (IsBaseBinary(a.chunked_type->id)) {
        AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), whole->buffers[1]->dptr, 0, (size_t)(a.chunked_type->bit_width / 8))));
        AHC_RETURN_NOT_OK(s->Allocate(0, &whole->buffers[2]));
      }
    } else {
      AHC_RETURN_NOT_OK(Concatenate(s, a.chunks, a.chunked_type, &whole));
    }
    a = Datum::Of(whole);
  }
  auto mode = VectorFunction::Chunked::SameLength;
  if (fn->Kind() == FuncKind::Vector) mode = static_cast<VectorFunction*>(fn)->chunked;
  std::vector<int64_t> lens;
  int64_t length = 0;
  if (mode == VectorFunction::Chunked::Take) {
    // selection.go:195-330: a chunked `indices` gives one output chunk per indices chunk; chunked values with
    // array indices give a one-chunk result
    Datum res;
    AHC_RETURN_NOT_OK(fn->Execute(ctx, opts, flat, &res));
    if (args[1].kind == DatumKind::Chunked) {
      for (auto& c : args[1].chunks) lens.push_back(c->length);
      *out = Rechunk(res.array, lens, false);
    } else {
      *out = Datum::OfChunks(res.array->type, {res.array});
    }
    return Status::OK();
  }
  AHC_RETURN_NOT_OK(SpanLengths(args, &lens, &length));
  Datum res;
  AHC_RETURN_NOT_OK(fn->Execute(ctx, opts, flat, &res));
  if (res.kind != DatumKind::Array) { *out = res; return Status::OK(); }
  switch (mode) {
    case VectorFunction::Chunked::SingleArray: *out = res; break;
    case VectorFunction::Chunked::SingleChunk: *out = Datum::OfChunks(res.array->type, {res.array}); break;
    case VectorFunction::Chunked::Filter: {
      // one output chunk per span, as long as that span's selection (getFilterOutputSize, vector_selection.go:57-81)
      const FilterOptions* fo = static_cast<const FilterOptions*>(opts ? opts : fn->DefaultOptions());
      const int null_sel = fo ? (int)fo->NullSelection : 0;
      const ArrayData& f = *flat[1].array;
      const uint8_t* fvalid = (f.buffers[0] && f.null_count != 0) ? (const uint8_t*)f.buffers[0]->dptr : nullptr;
      std::vector<int64_t> outs;
      int64_t pos = 0;
      for (int64_t n : lens) {
        int64_t cnt = 0;
        if (n > 0) AHC_RETURN_NOT_OK(s->FromStatus(ah_filter_count(s->ctx(), (const uint8_t*)f.buffers[1]->dptr, fvalid, f.offset + pos, n, null_sel, &cnt)));
        outs.push_back(cnt);
        pos += n;
      }
      *out = Rechunk(res.array, outs, false);
      break;
    }
    default:
      *out = Rechunk(res.array, lens, fn->Kind() == FuncKind::Scalar);
  }
  return Status::OK();
}

}  // namespace

// ---- temporal front end -------------------------------------------------------------------------------------
// Timestamp / Date / Time / Duration columns are integers with a label (ArrayData::logical).  The reference registers
// the SAME integer kernels for them under temporal input matchers (selection: vector_selection.go:1845-1870 "any
// fixed width"; hashing: vector_hash.go:545-560 by physical type; compare: scalar_comparisons.go:640-690; add /
// subtract: arithmetic.go:630-770), so the work here is the type rule, not a kernel: check the labels the way those
// matchers would, run the call on the bare integers, label the result.  What the reference reaches through implicit
// unit casts (commonTemporalResolution, arithmetic.go:130-131) is refused with the two types named — cast first.
namespace {

struct TemporalType { char kind = 0, unit = 0; bool zoned = false; };  // kind: 's' timestamp, 'D' duration, 'd' date, 't' time
TemporalType ParseTemporal(const std::string& f) {
  TemporalType t;
  if (f.empty()) return t;
  t.kind = f[1];
  t.unit = f[2];
  t.zoned = t.kind == 's' && f.size() > 4;
  return t;
}
std::string DatumLogical(const Datum& d, Status* st) {
  switch (d.kind) {
    case DatumKind::Array: return d.array->logical;
    case DatumKind::Scalar: return d.scalar->logical;
    case DatumKind::Chunked: {
      std::string l = d.chunks.empty() ? std::string() : d.chunks[0]->logical;
      for (auto& c : d.chunks)
        if (c->logical != l) *st = Status::Make(StatusCode::Invalid, "chunks of one column must be identically typed, but " + l + " and " + c->logical + " were encountered");
      return l;
    }
    default: return std::string();
  }
}
Datum WithLogical(const Datum& d, const std::string& l) {  // a relabelled view; buffers are shared
  Datum r = d;
  if (d.kind == DatumKind::Array) { r.array = std::make_shared<ArrayData>(*d.array); r.array->logical = l; }
  if (d.kind == DatumKind::Scalar) { r.scalar = std::make_shared<Scalar>(*d.scalar); r.scalar->logical = l; }
  if (d.kind == DatumKind::Chunked)
    for (auto& c : r.chunks) { c = std::make_shared<ArrayData>(*c); c->logical = l; }
  return r;
}
// equal types for the temporal matchers: identical, or two timestamps of one unit — exec.TimestampTypeUnit (kernel.go) matches
// on the unit alone, so the reference compares and subtracts timestamps of different zones (and zoned with naive ones)
bool SameTemporal(const std::string& a, const std::string& b) {
  if (a == b) return true;
  TemporalType x = ParseTemporal(a), y = ParseTemporal(b);
  return x.kind == 's' && y.kind == 's' && x.unit == y.unit;
}
int UnitRank(char u) { return u == 's' ? 0 : u == 'm' ? 1 : u == 'u' ? 2 : 3; }
std::string WithUnit(const std::string& l, char u) {
  TemporalType t = ParseTemporal(l);
  if (t.kind == 's') { std::string r = l; r[2] = u; return r; }
  if (t.kind == 'D') return std::string("tD") + u;
  if (t.kind == 't') return std::string("tt") + u;
  return l;
}
// arrow.DataType.String() of a temporal type, for messages
std::string TemporalName(const std::string& l) {
  TemporalType t = ParseTemporal(l);
  const std::string unit = t.unit == 's' ? "s" : t.unit == 'm' ? "ms" : t.unit == 'u' ? "us" : "ns";
  if (l == "tdD") return "date32";
  if (l == "tdm") return "date64";
  if (t.kind == 't') return std::string(UnitRank(t.unit) < 2 ? "time32[" : "time64[") + unit + "]";
  if (t.kind == 'D') return "duration[" + unit + "]";
  if (t.kind == 's') return "timestamp[" + unit + (l.size() > 4 ? ", tz=" + l.substr(4) : std::string()) + "]";
  return l;
}

// The temporal → temporal casts that are a unit change (cast_temporal.go:240-420 → ShiftTime :35-104): timestamp → timestamp,
// duration → duration, time32 / time64 among themselves, date32 ↔ date64.  One ah_shift_time pass; the result keeps the
// input's validity buffer and offset.  A scalar is converted on the host by the same rule.
Status CastTemporalUnits(ExecCtx* ctx, const Datum& in, const std::string& from, const std::string& to, const CastOptions& opts, Datum* out) {
  if (from == to) { *out = in; return Status::OK(); }
  TemporalType a = ParseTemporal(from), b = ParseTemporal(to);
  const DataType* st_in = TemporalStorage(from);
  const DataType* st_out = TemporalStorage(to);
  if (!st_in || !st_out || a.kind != b.kind)
    return Status::Make(StatusCode::NotImplemented, "cast from " + TemporalName(from) + " to " + TemporalName(to) + " is not built");
  int64_t factor = 1;
  int op = AH_SHIFT_MULTIPLY;
  if (a.kind == 'd') {
    factor = 86400000;  // days ↔ milliseconds
    op = from == "tdD" ? AH_SHIFT_MULTIPLY : AH_SHIFT_DIVIDE;
  } else {
    const int d = UnitRank(b.unit) - UnitRank(a.unit);
    for (int i = 0; i < (d < 0 ? -d : d); i++) factor *= 1000;
    op = d >= 0 ? AH_SHIFT_MULTIPLY : AH_SHIFT_DIVIDE;
  }
  const bool check = op == AH_SHIFT_MULTIPLY ? !opts.AllowTimeOverflow : !opts.AllowTimeTruncate;
  auto fail = [&](int64_t v) {
    return Status::Make(StatusCode::Invalid, "casting from " + TemporalName(from) + " to " + TemporalName(to) +
                                                 (op == AH_SHIFT_MULTIPLY ? " would result in out of bounds timestamp: " : " would lose data: ") + std::to_string(v));
  };
  const int wi = st_in->bit_width / 8, wo = st_out->bit_width / 8;
  Session* s = ctx->session;
  auto one = [&](const ArrayData& src, ArrayDataPtr* dst) -> Status {
    auto d = std::make_shared<ArrayData>(src);
    d->type = st_out;
    d->logical = to;
    d->buffers[1] = nullptr;
    AHC_RETURN_NOT_OK(s->Allocate((src.offset + src.length) * wo, &d->buffers[1], /*zero_all=*/src.offset != 0));
    if (src.length > 0) {
      const bool nulls = src.buffers[0] && src.null_count != 0;
      int64_t bad = 0;
      int rc = ah_shift_time(s->ctx(), wi * 8, wo * 8, op, factor, check, (const uint8_t*)src.buffers[1]->dptr + src.offset * wi,
                             nulls ? (const uint8_t*)src.buffers[0]->dptr : nullptr, src.offset, src.length,
                             (uint8_t*)d->buffers[1]->dptr + src.offset * wo, &bad);
      if (rc == AH_EINVALID && check) return fail(bad);
      AHC_RETURN_NOT_OK(s->FromStatus(rc));
    }
    *dst = d;
    return Status::OK();
  };
  switch (in.kind) {
    case DatumKind::Array: {
      ArrayDataPtr d;
      AHC_RETURN_NOT_OK(one(*in.array, &d));
      *out = Datum::Of(d);
      return Status::OK();
    }
    case DatumKind::Chunked: {
      std::vector<ArrayDataPtr> chunks;
      for (auto& c : in.chunks) {
        ArrayDataPtr d;
        AHC_RETURN_NOT_OK(one(*c, &d));
        chunks.push_back(d);
      }
      *out = Datum::OfChunks(st_out, std::move(chunks));
      return Status::OK();
    }
    case DatumKind::Scalar: {
      auto sc = std::make_shared<Scalar>(*in.scalar);
      sc->type = st_out;
      sc->logical = to;
      int64_t v = 0;
      if (wi == 4) { int32_t x; memcpy(&x, in.scalar->value, 4); v = x; } else memcpy(&v, in.scalar->value, 8);
      int64_t r;
      if (op == AH_SHIFT_MULTIPLY) {
        if (check && in.scalar->valid && (v < INT64_MIN / factor || v > INT64_MAX / factor)) return fail(v);
        r = wo == 4 ? (int64_t)(int32_t)((uint32_t)(int32_t)v * (uint32_t)(int32_t)factor) : (int64_t)((uint64_t)v * (uint64_t)factor);
      } else {
        r = v / factor;
        if (wo == 4) r = (int32_t)r;
        const int64_t again = wi == 4 ? (int64_t)(int32_t)((uint32_t)(int32_t)r * (uint32_t)(int32_t)factor) : (int64_t)((uint64_t)r * (uint64_t)factor);
        if (check && in.scalar->valid && again != v) return fail(v);
      }
      memset(sc->value, 0, 8);
      if (wo == 4) { int32_t x = (int32_t)r; memcpy(sc->value, &x, 4); } else memcpy(sc->value, &r, 8);
      *out = Datum::Of(sc);
      return Status::OK();
    }
    default: return Status::Make(StatusCode::Invalid, "cast: not an array or scalar");
  }
}
std::string Describe(const std::string& l, const Datum& d) { return l.empty() ? std::string(d.type() ? d.type()->name : "?") : l; }

enum class TemporalRule { Preserve, Plain, Same, Add, Sub, Cast };
const std::map<std::string, TemporalRule>& TemporalRules() {
  static const std::map<std::string, TemporalRule> r = {
      {"take", TemporalRule::Preserve}, {"array_take", TemporalRule::Preserve}, {"filter", TemporalRule::Preserve},
      {"array_filter", TemporalRule::Preserve}, {"unique", TemporalRule::Preserve}, {"sort", TemporalRule::Preserve},
      {"dictionary_encode", TemporalRule::Preserve},
      {"is_null", TemporalRule::Plain}, {"is_not_null", TemporalRule::Plain}, {"sort_indices", TemporalRule::Plain},
      {"equal", TemporalRule::Same}, {"not_equal", TemporalRule::Same}, {"greater", TemporalRule::Same}, {"greater_equal", TemporalRule::Same},
      {"less", TemporalRule::Same}, {"less_equal", TemporalRule::Same}, {"is_in", TemporalRule::Same},
      {"add", TemporalRule::Add}, {"add_unchecked", TemporalRule::Add},
      {"subtract", TemporalRule::Sub}, {"subtract_unchecked", TemporalRule::Sub}, {"sub", TemporalRule::Sub}, {"sub_unchecked", TemporalRule::Sub},
      {"cast", TemporalRule::Cast}};
  return r;
}


// time32 / time64 ± duration of the same unit → the time type (GetArithmeticFunctionTimeDuration, scalar_arithmetic.go:47-65;
// timeDurationOp, base_arithmetic.go:642-700): the duration is narrowed to the time's storage (`OutT(b)`: for time32 the int64
// duration is TRUNCATED to int32, as the generic instantiation does), the integer kernel adds / subtracts (the checked names
// report "overflow" from its carry test), and every result — ScalarBinary walks all slots — must lie in [0, one day):
// otherwise the call fails with the LAST offending value, as the Go loop that keeps overwriting its error does.
Status TimePlusDuration(ExecCtx* ctx, const std::string& name, const FunctionOptions* opts, const Datum& time, const std::string& time_lg,
                        const Datum& dur, Datum* out) {
  const DataType* st = TemporalStorage(time_lg);
  TemporalType t = ParseTemporal(time_lg);
  int64_t day = 86400;
  for (int i = 0; i < UnitRank(t.unit); i++) day *= 1000;
  Datum d = dur;
  if (st->id == Type::INT32) {
    CastOptions narrow = CastOptions::Unsafe(GetDataType(Type::INT32));
    AHC_RETURN_NOT_OK(CastDatum(ctx, dur, narrow, &d));
  }
  Datum r;
  AHC_RETURN_NOT_OK(CallFunction(ctx, name, opts, {time, d}, &r));
  Session* s = ctx->session;
  auto check = [&](const ArrayData& a) -> Status {
    if (a.length == 0) return Status::OK();
    const int w = st->bit_width / 8;
    const uint8_t* vals = (const uint8_t*)a.buffers[1]->dptr + a.offset * w;
    int64_t lo = 0, hi = 0;
    if (w == 4) { int32_t mn, mx; AHC_RETURN_NOT_OK(s->FromStatus(ah_min_max(s->ctx(), AH_INT32, vals, a.length, &mn, &mx))); lo = mn; hi = mx; }
    else AHC_RETURN_NOT_OK(s->FromStatus(ah_min_max(s->ctx(), AH_INT64, vals, a.length, &lo, &hi)));
    if (lo >= 0 && hi < day) return Status::OK();
    std::vector<uint8_t> host((size_t)a.length * w);   // the error path: name the value the sequential loop would report
    AHC_RETURN_NOT_OK(s->FromStatus(ah_download_async(s->ctx(), host.data(), vals, host.size())));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));
    int64_t bad = 0;
    for (int64_t i = a.length - 1; i >= 0; i--) {
      int64_t v;
      if (w == 4) { int32_t x; memcpy(&x, host.data() + i * 4, 4); v = x; } else memcpy(&v, host.data() + i * 8, 8);
      if (v < 0 || v >= day) { bad = v; break; }
    }
    return Status::Make(StatusCode::Invalid, std::to_string(bad) + " is not within acceptable range of [0, " + std::to_string(day) + ") s");
  };
  if (r.kind == DatumKind::Array) AHC_RETURN_NOT_OK(check(*r.array));
  else if (r.kind == DatumKind::Chunked) { for (auto& c : r.chunks) AHC_RETURN_NOT_OK(check(*c)); }
  else if (r.kind == DatumKind::Scalar && r.scalar->valid) {
    int64_t v;
    if (st->id == Type::INT32) { int32_t x; memcpy(&x, r.scalar->value, 4); v = x; } else memcpy(&v, r.scalar->value, 8);
    if (v < 0 || v >= day) return Status::Make(StatusCode::Invalid, std::to_string(v) + " is not within acceptable range of [0, " + std::to_string(day) + ") s");
  }
  *out = WithLogical(r, time_lg);
  return Status::OK();
}

Status CallTemporal(ExecCtx* ctx, const std::string& name, const FunctionOptions* opts, const std::vector<Datum>& args_in,
                    const std::vector<std::string>& lg_in, Datum* out) {
  auto refuse = [&]() {
    std::string types;
    for (size_t i = 0; i < args_in.size(); i++) types += (i ? ", " : "") + Describe(lg_in[i], args_in[i]);
    return Status::Make(StatusCode::NotImplemented, "function '" + name + "' has no kernel matching input types (" + types + ")");
  };
  auto it = TemporalRules().find(name);
  if (it == TemporalRules().end()) return refuse();
  // DispatchBest for two temporal operands whose units differ: both go to the finer unit (commonTemporalResolution +
  // replaceTemporalTypes for add / subtract, arithmetic.go:130-131, utils.go:130-170; commonTemporal for the comparisons,
  // utils.go:329-399) through the safe unit cast.  Dates are promoted only among themselves, for comparisons.
  std::vector<Datum> args = args_in;
  std::vector<std::string> lg = lg_in;
  if (args.size() == 2 && !lg[0].empty() && !lg[1].empty() && it->second != TemporalRule::Preserve && it->second != TemporalRule::Plain &&
      it->second != TemporalRule::Cast && name != "is_in") {
    TemporalType a = ParseTemporal(lg[0]), b = ParseTemporal(lg[1]);
    auto unit_kind = [](char k) { return k == 's' || k == 'D' || k == 't'; };
    std::string to[2] = {lg[0], lg[1]};
    if (unit_kind(a.kind) && unit_kind(b.kind) && a.unit != b.unit && (it->second != TemporalRule::Same || a.kind == b.kind)) {
      const char finest = UnitRank(a.unit) > UnitRank(b.unit) ? a.unit : b.unit;
      to[0] = WithUnit(lg[0], finest);
      to[1] = WithUnit(lg[1], finest);
    } else if (it->second == TemporalRule::Same && a.kind == 'd' && b.kind == 'd' && lg[0] != lg[1]) {
      to[0] = to[1] = "tdm";
    }
    for (int i = 0; i < 2; i++)
      if (to[i] != lg[i]) {
        Datum moved;
        AHC_RETURN_NOT_OK(CastTemporalUnits(ctx, args[i], lg[i], to[i], CastOptions::Safe(nullptr), &moved));
        args[i] = moved;
        lg[i] = to[i];
      }
  }
  std::vector<Datum> bare;
  for (size_t i = 0; i < args.size(); i++) bare.push_back(lg[i].empty() ? args[i] : WithLogical(args[i], ""));
  std::string out_logical;
  const bool checked = name.find("_unchecked") == std::string::npos;
  switch (it->second) {
    case TemporalRule::Preserve:  // the values move, their meaning does not; indices / masks are plain columns
      for (size_t i = 1; i < args.size(); i++) if (!lg[i].empty()) return refuse();
      out_logical = lg[0];
      break;
    case TemporalRule::Plain:
      for (size_t i = 1; i < args.size(); i++) if (!lg[i].empty()) return refuse();
      break;
    case TemporalRule::Same: {
      for (size_t i = 1; i < args.size(); i++) if (!SameTemporal(lg[0], lg[i])) return refuse();
      if (name == "is_in") {
        auto* so = dynamic_cast<const SetOptions*>(opts);
        if (!so || !so->ValueSet || !SameTemporal(lg[0], so->ValueSet->logical))
          return Status::Make(StatusCode::TypeError, "is_in: the value set (" + (so && so->ValueSet ? Describe(so->ValueSet->logical, Datum::Of(so->ValueSet)) : "none") + ") is not of the column's type " + lg[0]);
        SetOptions plain = *so;
        plain.ValueSet = std::make_shared<ArrayData>(*so->ValueSet);
        plain.ValueSet->logical.clear();
        return CallFunction(ctx, name, &plain, bare, out);
      }
      break;
    }
    case TemporalRule::Add: {  // arithmetic.go:648-668
      if (args.size() != 2) return refuse();
      TemporalType a = ParseTemporal(lg[0]), b = ParseTemporal(lg[1]);
      if (a.kind == 's' && b.kind == 'D' && a.unit == b.unit) out_logical = lg[0];        // timestamp + duration → timestamp
      else if (a.kind == 'D' && b.kind == 's' && a.unit == b.unit) out_logical = lg[1];   // duration + timestamp → timestamp
      else if (a.kind == 'D' && b.kind == 'D' && a.unit == b.unit) out_logical = lg[0];   // duration + duration → duration
      else if (a.kind == 't' && b.kind == 'D' && a.unit == b.unit) return TimePlusDuration(ctx, name, opts, bare[0], lg[0], bare[1], out);   // time + duration → time
      else return refuse();
      break;
    }
    case TemporalRule::Sub: {  // arithmetic.go:694-762
      if (args.size() != 2) return refuse();
      TemporalType a = ParseTemporal(lg[0]), b = ParseTemporal(lg[1]);
      const std::string dur = std::string("tD") + a.unit;
      if (a.kind == 's' && b.kind == 's' && SameTemporal(lg[0], lg[1])) out_logical = dur;     // timestamp − timestamp → duration
      else if (a.kind == 's' && b.kind == 'D' && a.unit == b.unit) out_logical = lg[0];          // timestamp − duration → timestamp
      else if (a.kind == 'D' && b.kind == 'D' && a.unit == b.unit) out_logical = lg[0];
      else if (a.kind == 't' && b.kind == 'D' && a.unit == b.unit) return TimePlusDuration(ctx, name, opts, bare[0], lg[0], bare[1], out);   // time − duration → time
      else if (a.kind == 't' && lg[0] == lg[1] && (a.unit == 'u' || a.unit == 'n')) out_logical = dur;  // time64 − time64 → duration
      else if (a.kind == 't' && lg[0] == lg[1]) {
        // time32 − time32 → duration: the int32 kernel, widened afterwards (arithmetic.go:721-741)
        Datum narrow;
        AHC_RETURN_NOT_OK(CallFunction(ctx, name, opts, bare, &narrow));
        AHC_RETURN_NOT_OK(CastDatum(ctx, narrow, CastOptions::Safe(GetDataType(Type::INT64)), out));
        *out = WithLogical(*out, dur);
        return Status::OK();
      } else if (lg[0] == "tdm" && lg[1] == "tdm") out_logical = "tDm";                        // date64 − date64 → duration[ms]
      else if (lg[0] == "tdD" && lg[1] == "tdD") {
        // date32 − date32 → duration[s] (SubtractDate32, base_arithmetic.go:702-720).  Unchecked: the difference AND the ×86400 are
        // int32 arithmetic that wraps, then widened; checked: both in int64 (where neither step can overflow).
        auto day = std::make_shared<Scalar>();
        day->valid = true;
        Datum diff, secs;
        if (checked) {
          Datum w0, w1;
          AHC_RETURN_NOT_OK(CastDatum(ctx, bare[0], CastOptions::Safe(GetDataType(Type::INT64)), &w0));
          AHC_RETURN_NOT_OK(CastDatum(ctx, bare[1], CastOptions::Safe(GetDataType(Type::INT64)), &w1));
          AHC_RETURN_NOT_OK(CallFunction(ctx, "subtract", opts, {w0, w1}, &diff));
          day->type = GetDataType(Type::INT64);
          int64_t v = 86400; memcpy(day->value, &v, 8);
          AHC_RETURN_NOT_OK(CallFunction(ctx, "multiply", opts, {diff, Datum::Of(day)}, &secs));
        } else {
          AHC_RETURN_NOT_OK(CallFunction(ctx, "subtract_unchecked", opts, bare, &diff));
          day->type = GetDataType(Type::INT32);
          int32_t v = 86400; memcpy(day->value, &v, 4);
          Datum narrow;
          AHC_RETURN_NOT_OK(CallFunction(ctx, "multiply_unchecked", opts, {diff, Datum::Of(day)}, &narrow));
          AHC_RETURN_NOT_OK(CastDatum(ctx, narrow, CastOptions::Safe(GetDataType(Type::INT64)), &secs));
        }
        *out = WithLogical(secs, "tDs");
        return Status::OK();
      } else return refuse();
      break;
    }
    case TemporalRule::Cast: {
      // temporal ↔ its storage integer is a relabelling (cast.go "zero copy" casts); temporal → temporal of the same family is a unit change
      auto* co = dynamic_cast<const CastOptions*>(opts);
      if (!co || args.size() != 1) return refuse();
      const DataType* storage = TemporalStorage(lg[0]);
      if (!co->ToLogical.empty()) {
        if (!TemporalStorage(co->ToLogical)) return Status::Make(StatusCode::NotImplemented, "cast to " + co->ToLogical + ": not a temporal type this layer takes");
        return CastTemporalUnits(ctx, args[0], lg[0], co->ToLogical, *co, out);
      }
      if (co->ToType && storage && co->ToType->id == storage->id) { *out = bare[0]; return Status::OK(); }
      return Status::Make(StatusCode::NotImplemented, std::string("cast from ") + lg[0] + " to " + (co->ToType ? co->ToType->name : "?") + " is not built");
    }
  }
  AHC_RETURN_NOT_OK(CallFunction(ctx, name, opts, bare, out));
  if (!out_logical.empty()) {
    if (name == "dictionary_encode") {  // the label belongs to the dictionary's values
      auto relabel = [&](ArrayDataPtr& a) {
        a = std::make_shared<ArrayData>(*a);
        if (a->dictionary) { a->dictionary = std::make_shared<ArrayData>(*a->dictionary); a->dictionary->logical = out_logical; }
      };
      if (out->kind == DatumKind::Array) relabel(out->array);
      if (out->kind == DatumKind::Chunked) for (auto& c : out->chunks) relabel(c);
    } else {
      *out = WithLogical(*out, out_logical);
    }
  }
  return Status::OK();
}

}  // namespace

Status CallFunction(ExecCtx* ctx, const std::string& name, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out) {
  FunctionRegistry* reg = ctx && ctx->Registry ? ctx->Registry : GetFunctionRegistry();
  Function* fn = reg->GetFunction(name);
  if (!fn) return Status::Make(StatusCode::KeyError, "function '" + name + "' not found");  // exec.go:191-199
  // HOST-resident arguments (ArrayData::on_host: their buffers hold no device pointer) never reach a kernel: the call is streamed
  // span by span where the function allows it, and the host arguments are uploaded whole where it does not (hoststream.cc)
  for (auto& a : args)
    if (a.kind == DatumKind::Array && a.array->on_host) {
      if (!ctx || !ctx->session) return Status::Make(StatusCode::Invalid, "host-resident argument: the ExecCtx carries no session to stream it through");
      bool handled = false;
      AHC_RETURN_NOT_OK(CallHostResident(ctx, name, opts, args, out, &handled));
      if (handled) return Status::OK();
      std::vector<Datum> dev_args(args);
      AHC_RETURN_NOT_OK(MaterializeAllOnDevice(ctx->session, &dev_args));
      return CallFunction(ctx, name, opts, dev_args, out);
    }
  {
    std::vector<std::string> lg;
    bool any = false;
    Status st;
    for (auto& a : args) { lg.push_back(DatumLogical(a, &st)); any = any || !lg.back().empty(); }
    AHC_RETURN_NOT_OK(st);
    if (!any)
      if (auto* co = dynamic_cast<const CastOptions*>(opts))
        if (name == "cast" && !co->ToLogical.empty() && args.size() == 1) {  // integer → temporal of that storage: a relabelling
          const DataType* storage = TemporalStorage(co->ToLogical);
          if (!storage) return Status::Make(StatusCode::NotImplemented, "cast to " + co->ToLogical + ": not a temporal type this layer takes");
          if (!args[0].type() || args[0].type()->id != storage->id)
            return Status::Make(StatusCode::NotImplemented, std::string("cast from ") + (args[0].type() ? args[0].type()->name : "?") + " to " + co->ToLogical + " is not built; cast to " + storage->name + " first");
          *out = WithLogical(args[0], co->ToLogical);
          return Status::OK();
        }
    if (any) return CallTemporal(ctx, name, opts, args, lg, out);
  }
  for (auto& a : args)
    if (a.kind == DatumKind::Chunked) return ExecuteChunked(ctx, fn, opts, args, out);
  return fn->Execute(ctx, opts, args, out);
}

}  // namespace compute

// ---- arrow/math ----------------------------------------------------------------------------
namespace math {
const Float64Funcs Float64{};
const Int64Funcs Int64{};
const Uint64Funcs Uint64{};

// a host-resident array: chunk by chunk through the device, one final reduction (hoststream.cc SumHostResident)
static compute::ExecCtx HostCtx(Session* s) { compute::ExecCtx c; c.session = s; return c; }
static const void* ValuesPtr(const ArrayData& a) {
  return (const uint8_t*)a.buffers[1]->dptr + a.offset * (a.type->bit_width / 8);  // Float64Values(): offset applied
}
Status Float64Funcs::Sum(Session* s, const ArrayData& a, double* out) const {
  if (a.type->id != Type::FLOAT64) return Status::Make(StatusCode::TypeError, "math.Float64.Sum needs a float64 array");
  if (a.length == 0) { *out = 0; return Status::OK(); }  // float64.go:35-37
  if (a.on_host) { if (a.device_twin) return Sum(s, *a.device_twin, out); auto c = HostCtx(s); return compute::SumHostResident(&c, a, out, nullptr, nullptr); }
  return s->FromStatus(ah_sum_float64(s->ctx(), (const double*)ValuesPtr(a), (size_t)a.length, out));
}
Status Int64Funcs::Sum(Session* s, const ArrayData& a, int64_t* out) const {
  if (a.type->id != Type::INT64) return Status::Make(StatusCode::TypeError, "math.Int64.Sum needs an int64 array");
  if (a.length == 0) { *out = 0; return Status::OK(); }
  if (a.on_host) { if (a.device_twin) return Sum(s, *a.device_twin, out); auto c = HostCtx(s); return compute::SumHostResident(&c, a, nullptr, out, nullptr); }
  return s->FromStatus(ah_sum_int64(s->ctx(), (const int64_t*)ValuesPtr(a), (size_t)a.length, out));
}
Status Uint64Funcs::Sum(Session* s, const ArrayData& a, uint64_t* out) const {
  if (a.type->id != Type::UINT64) return Status::Make(StatusCode::TypeError, "math.Uint64.Sum needs a uint64 array");
  if (a.length == 0) { *out = 0; return Status::OK(); }
  if (a.on_host) { if (a.device_twin) return Sum(s, *a.device_twin, out); auto c = HostCtx(s); return compute::SumHostResident(&c, a, nullptr, nullptr, out); }
  return s->FromStatus(ah_sum_uint64(s->ctx(), (const uint64_t*)ValuesPtr(a), (size_t)a.length, out));
}
}  // namespace math

}  // namespace arrowhip
