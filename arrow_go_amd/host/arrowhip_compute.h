// arrowhip_compute.h — host-side mirror of arrow-go's compute plugin interface for the
// hot path, above the C ABI of libarrowhip.so.
//
// Why C++: the reference's host side is Go and there is no Go toolchain in this
// image, so the layer a Go maintainer would write with cgo is
// restated here in C++ with the SAME names, argument meaning and error behaviour:
//
//   arrowhip::exec::{ArraySpan, ExecValue, ExecSpan, ExecResult, KernelCtx,
//                    NullHandling, MemAlloc, ScalarKernel, VectorKernel}
//                                   ≙ arrow/compute/exec/{span.go:76-88,548-576,
//                                     kernel.go:78-93,457-499,617-727}
//   arrowhip::compute::{Function, ScalarFunction, VectorFunction, MetaFunction,
//                    FunctionRegistry, GetFunctionRegistry, NewChildRegistry, ExecCtx,
//                    Datum, CallFunction}
//                                   ≙ arrow/compute/{functions.go:30-41,239-420,
//                                     registry.go:30-120, executor.go:46-122,
//                                     datum.go:35-40, exec.go:59-191}
//   the scalar executor (length check, output preallocation, propagateNulls) and the
//   vector executor               ≙ arrow/compute/executor.go:237-349,498-702,896-1100
//   arrowhip::math::{Float64, Int64, Uint64}.Sum
//                                   ≙ arrow/math/{float64,int64,uint64}.go:25-47
//
// Arrays live in HBM: ArrayData buffers are device allocations owned by a Session's
// ah_ctx.  Every kernel ExecFn in kernels.cc is a thin adapter from ArraySpan to one
// leaf of include/arrowhip.h — no arithmetic happens on the host.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/arrowhip.h"

namespace arrowhip {

// ---- arrow.Type / arrow.DataType (arrow/datatype.go:36-72) ---------------------------
enum class Type : int {
  NA = 0, BOOL = 1, UINT8 = 2, INT8 = 3, UINT16 = 4, INT16 = 5, UINT32 = 6, INT32 = 7,
  UINT64 = 8, INT64 = 9, FLOAT16 = 10, FLOAT32 = 11, FLOAT64 = 12, STRING = 13, BINARY = 14,
  FIXED_SIZE_BINARY = 15, DECIMAL128 = 23, DECIMAL256 = 24,   // arrow.FIXED_SIZE_BINARY / DECIMAL128 / DECIMAL256 (arrow/datatype.go)
  LARGE_STRING = 34, LARGE_BINARY = 35, DICTIONARY = 36
};

struct DataType {
  Type id;
  int bit_width;       // arrow.FixedWidthDataType.BitWidth(); base-binary types: the OFFSET width (32 / 64)
  const char* name;    // DataType.Name()
  const char* format;  // Arrow C Data Interface format string
};
const DataType* GetDataType(Type id);  // singletons; nullptr if unsupported
// FixedSizeBinary / Decimal128 / Decimal256 are PARAMETRIC (byte width; precision, scale): one interned DataType per C Data format
// ("w:16", "d:20,3", "d:40,5,256").  bit_width = 8 · the value's bytes.  GetDataType(id) of the three ids is a width-less
// placeholder for dispatch by type id.  Keys of unique / dictionary_encode (vector_hash.go:608-609, 698); no arithmetic on them.
const DataType* FixedWidthBinaryFromFormat(const std::string& format);   // nullptr: not such a format
bool IsFixedWidthBinary(Type id);
bool IsInteger(Type id);
bool IsSignedInteger(Type id);
bool IsFloating(Type id);
bool IsNumeric(Type id);
// Temporal types by their C Data format: the integer type that stores them (nullptr: not a temporal format this layer takes)
const DataType* TemporalStorage(const std::string& format);
bool IsBaseBinary(Type id);  // String, Binary, LargeString, LargeBinary: buffers = [validity, offsets, data]

// ---- errors (arrow/errors.go) ---------------------------------------------------------
enum class StatusCode { OK = 0, Invalid, Index, NotImplemented, TypeError, KeyError, Hip };
struct Status {
  StatusCode code = StatusCode::OK;
  std::string msg;
  bool ok() const { return code == StatusCode::OK; }
  static Status OK() { return Status(); }
  static Status Make(StatusCode c, std::string m) { Status s; s.code = c; s.msg = std::move(m); return s; }
  std::string ToString() const;  // "invalid: …" like fmt.Errorf("%w: …", arrow.ErrInvalid)
};
#define AHC_RETURN_NOT_OK(expr) do { ::arrowhip::Status s__ = (expr); if (!s__.ok()) return s__; } while (0)

// ---- device memory (memory.Allocator / memory.Buffer: arrow/memory/allocator.go:23-27,
//      buffer.go:26) ---------------------------------------------------------------------
class Session;  // owns the ah_ctx
struct Buffer {
  Session* session = nullptr;
  // Keeps the session (its ah_ctx and the block pool ~Buffer returns memory to) alive for as long as any buffer of it
  // is: a datum or an exported ArrowDeviceArray may outlive ahc_session_destroy.  Set by Session::Keep() wherever
  // `session` is set.
  std::shared_ptr<Session> keep;
  void* dptr = nullptr;
  int64_t size = 0;
  // foreign device memory (Arrow C Device Data Interface import): not freed here; `owner` keeps the
  // producer's ArrowArray alive and calls its release callback when the last buffer goes away
  bool owned = true;
  std::shared_ptr<void> owner;
  size_t alloc_bytes = 0;  // size class of the block behind dptr when it came from Session::Allocate (0: not pooled)
  // HOST-resident bytes (dptr is null): a column left where its producer put it (ahc_import_host; `owner` keeps the producer's
  // ArrowArray) or a result the streaming executor wrote into pinned memory of the session's (host_alloc_bytes != 0: returned to
  // the session's pinned pool).  Only ArrayData::on_host arrays hold such buffers (hoststream.cc).
  void* hptr = nullptr;
  size_t host_alloc_bytes = 0;
  ~Buffer();
};
using BufferPtr = std::shared_ptr<Buffer>;

class Session : public std::enable_shared_from_this<Session> {
 public:
  static Status Create(int device_id, std::shared_ptr<Session>* out);
  ~Session();
  // buffers call this so that the session outlives them (Buffer::keep)
  void Keep(Buffer* b) { b->session = this; b->keep = shared_from_this(); }
  ah_ctx* ctx() const { return ctx_; }
  // zero-filled like GoAllocator / Mallocator (quirk 4 in SURVEY.md §8a): kernels rely on it
  // zero_all = false: only the padding behind the last whole 64-byte block is cleared — for outputs whose every slot the
  // kernel writes anyway (the zero-fill of a 1 GiB output costs a third of the Add that fills it)
  Status Allocate(int64_t nbytes, BufferPtr* out, bool zero_all = true);
  Status AllocateBitmap(int64_t nbits, BufferPtr* out);
  Status FromStatus(int ah_status) const;  // AH_* → Status with ah_last_error()
  // Freed output buffers are kept in size classes and handed out again — what a pooling memory.Allocator does for the
  // Go executor.  hipMalloc / hipFree map and unmap the whole range (≈ 0.35 ms per GiB round trip, two thirds of an
  // Int64 Add kernel); reuse is safe because every consumer is ordered on the session's compute stream and both copy
  // directions wait for it (ah_upload_async / ah_download_async).  ARROWHIP_POOL_BYTES caps the cache (default 16 GiB; 0 = off).
  void Release(void* dptr, size_t alloc_bytes);
  void TrimPool();
  // pinned host memory for the results of the streaming executor, pooled like the device blocks (pinning 1 GiB takes longer than
  // moving it across PCIe)
  Status AllocatePinned(int64_t nbytes, BufferPtr* out);
  void ReleasePinned(void* hptr, size_t alloc_bytes);
  // the session's ah_ingest (three slots of `chunk_bytes`; 0: 32 MiB), created on first use, re-created when the chunk size changes
  Status Ingest(size_t chunk_bytes, struct ah_ingest** out);
 private:
  ah_ctx* ctx_ = nullptr;
  std::multimap<size_t, void*> pool_;
  size_t pooled_bytes_ = 0, pool_cap_ = (size_t)16 << 30;
  std::multimap<size_t, void*> pinned_pool_;
  size_t pinned_pooled_bytes_ = 0;
  struct ah_ingest* ingest_ = nullptr;
  size_t ingest_chunk_ = 0;
};

// ---- arrow.ArrayData (arrow/array.go:54-86) / scalar.Scalar ----------------------------
constexpr int64_t kUnknownNullCount = -1;  // array.UnknownNullCount
struct ArrayData {
  const DataType* type = nullptr;
  int64_t length = 0;
  int64_t null_count = kUnknownNullCount;
  int64_t offset = 0;
  BufferPtr buffers[3];                    // [0] validity, [1] values / boolean data / offsets, [2] var-length data
  std::shared_ptr<ArrayData> dictionary;   // DICTIONARY arrays: buffers = the indices, `dictionary` = the values
  const DataType* dict_value_type = nullptr;
  const DataType* dict_index_type = nullptr;  // nullptr = int32 (what dictionary_encode produces)
  // Temporal columns (arrow.TimestampType / Date32 / Date64 / Time32 / Time64 / DurationType, datatype_fixedwidth.go):
  // `type` is the PHYSICAL integer type the kernels run on, `logical` the Arrow C Data format of the
  // temporal type ("tsu:UTC", "tdD", "ttm", "tDn" …; empty = a plain column).  CallFunction checks the
  // reference's type rules on it, runs the integer kernel, and labels the result (core.cc, "temporal front end").
  std::string logical;
  // the buffers are HOST memory (Buffer::hptr, dptr == nullptr): a column imported with ahc_import_host or a result of the
  // streaming executor.  Flat fixed-width columns only.  compute::CallFunction streams such arguments through the device in chunks
  // where the function allows it and uploads them whole where it does not (core.cc → hoststream.cc); math::*.Sum streams them;
  // ExecuteScalarExpression / ExecuteScalarSubstrait upload them before evaluating.  exec::ArraySpan never sees one.
  bool on_host = false;
  // … and the device-resident copy MaterializeOnDevice made the first time a call that cannot stream needed the column: later
  // calls (streamable or not) use it instead of crossing the link again.  Lives and dies with the host array.
  mutable std::shared_ptr<ArrayData> device_twin;
};
using ArrayDataPtr = std::shared_ptr<ArrayData>;

struct Scalar {
  const DataType* type = nullptr;
  bool valid = false;
  alignas(8) uint8_t value[8] = {0};  // little-endian payload of width type->bit_width (bool: value[0])
  std::string logical;                // as ArrayData::logical
};
using ScalarPtr = std::shared_ptr<Scalar>;

namespace exec {

// exec.BufferSpan / exec.ArraySpan (arrow/compute/exec/span.go:43-88)
struct BufferSpan {
  uint8_t* buf = nullptr;  // DEVICE pointer (nil ⇔ no buffer)
  int64_t len = 0;
  BufferPtr owner;
  bool self_alloc = false;
  void WrapBuffer(const BufferPtr& b) { buf = b ? (uint8_t*)b->dptr : nullptr; len = b ? b->size : 0; owner = b; self_alloc = true; }
};
struct ArraySpan {
  const DataType* type = nullptr;
  int64_t len = 0, nulls = kUnknownNullCount, offset = 0;
  BufferSpan buffers[3];
  ArrayDataPtr dictionary;  // DICTIONARY arrays (exec.ArraySpan.Dictionary, span.go:159-168)
  const DataType* dict_value_type = nullptr;
  const DataType* dict_index_type = nullptr;
  void SetMembers(const ArrayData& d);
  bool MayHaveNulls() const { return nulls != 0 && buffers[0].buf != nullptr; }  // span.go:127-129
  // ArraySpan.UpdateNullCount (span.go:112-125): popcount of the validity on the device
  Status UpdateNullCount(Session* s, int64_t* out = nullptr);
  ArrayDataPtr MakeData() const;  // ArraySpan.MakeData (span.go:238-)
};
struct ExecValue {
  ArraySpan array;
  const Scalar* scalar = nullptr;
  bool IsArray() const { return scalar == nullptr; }
  bool IsScalar() const { return scalar != nullptr; }
  const DataType* type() const { return scalar ? scalar->type : array.type; }
};
struct ExecSpan {
  int64_t len = 0;
  std::vector<ExecValue> values;
};
using ExecResult = ArraySpan;

// exec.NullHandling / exec.MemAlloc (arrow/compute/exec/kernel.go:457-499)
enum class NullHandling { NullIntersection, NullComputedPrealloc, NullComputedNoPrealloc, NullNoOutput };
enum class MemAlloc { MemPrealloc, MemNoPrealloc };

// exec.KernelCtx (kernel.go:78-93)
struct KernelCtx {
  Session* session = nullptr;
  const void* state = nullptr;   // KernelState: the FunctionOptions for vector kernels
  const void* kernel_data = nullptr;  // Kernel.Data
  Status Allocate(int64_t nbytes, BufferPtr* out, bool zero_all = true) { return session->Allocate(nbytes, out, zero_all); }
  Status AllocateBitmap(int64_t nbits, BufferPtr* out) { return session->AllocateBitmap(nbits, out); }
};

using ArrayKernelExec = std::function<Status(KernelCtx*, const ExecSpan&, ExecResult*)>;  // kernel.go:617

// exec.KernelSignature with exact-type matching (kernel.go:330-452, reduced to what the
// path needs: fixed input types, output = a fixed type or the first input's type)
struct KernelSignature {
  std::vector<Type> in_types;
  bool out_is_first_input = true;
  Type out_type = Type::NA;
  bool MatchesInputs(const std::vector<const DataType*>& types) const;
};

struct ScalarKernel {  // kernel.go:632-672
  KernelSignature sig;
  ArrayKernelExec exec_fn;
  NullHandling null_handling = NullHandling::NullIntersection;  // defaults :660-661
  MemAlloc mem_alloc = MemAlloc::MemPrealloc;
  std::shared_ptr<void> data;
};
using FinalizeFn = std::function<Status(KernelCtx*, std::vector<ArraySpan>*)>;
struct VectorKernel {  // kernel.go:686-727
  KernelSignature sig;
  ArrayKernelExec exec_fn;
  NullHandling null_handling = NullHandling::NullComputedNoPrealloc;  // defaults :724-725
  MemAlloc mem_alloc = MemAlloc::MemNoPrealloc;
  bool can_execute_chunkwise = true;
  bool output_is_dictionary = false;
};

}  // namespace exec

namespace compute {

// FunctionOptions structs (compute/arithmetic.go ArithmeticOptions, kernels FilterOptions
// vector_selection.go:41-43, TakeOptions :49-51, DictionaryEncodeOptions vector_hash.go:96-99)
struct FunctionOptions { virtual ~FunctionOptions() = default; virtual const char* TypeName() const = 0; };
struct ArithmeticOptions : FunctionOptions { bool NoCheckOverflow = false; const char* TypeName() const override { return "ArithmeticOptions"; } };
enum NullSelectionBehavior { DropNulls = 0, EmitNulls = 1 };
struct FilterOptions : FunctionOptions {
  NullSelectionBehavior NullSelection = DropNulls;
  // not in the reference (its PrimitiveFilter always counts first: vector_selection.go:459-475): true = the output is allocated for the
  // worst case (as many rows as the input) and the kernel runs in ONE call — no count, no second launch after the host has heard back
  bool WorstCaseOutput = false;
  const char* TypeName() const override { return "FilterOptions"; }
};
struct TakeOptions : FunctionOptions { bool BoundsCheck = true; const char* TypeName() const override { return "TakeOptions"; } };
enum NullEncodingBehavior { NullEncodingMask = 0, NullEncodingEncode = 1 };
struct DictionaryEncodeOptions : FunctionOptions { NullEncodingBehavior NullEncoding = NullEncodingMask; const char* TypeName() const override { return "DictionaryEncodeOptions"; } };
// kernels.CastOptions (kernels/cast.go:27-35); NewCastOptions(dt, safe) / SafeCastOptions / UnsafeCastOptions (compute/cast.go)
struct CastOptions : FunctionOptions {
  const DataType* ToType = nullptr;
  std::string ToLogical;  // a temporal target by its C Data format ("tsu:", "tdD" …): see core.cc, temporal front end
  bool AllowIntOverflow = false, AllowTimeTruncate = false, AllowTimeOverflow = false, AllowDecimalTruncate = false,
       AllowFloatTruncate = false, AllowInvalidUtf8 = false;
  const char* TypeName() const override { return "CastOptions"; }
  static CastOptions Safe(const DataType* to) { CastOptions o; o.ToType = to; return o; }
  static CastOptions Unsafe(const DataType* to) {
    CastOptions o; o.ToType = to;
    o.AllowIntOverflow = o.AllowTimeTruncate = o.AllowTimeOverflow = o.AllowDecimalTruncate = o.AllowFloatTruncate = o.AllowInvalidUtf8 = true;
    return o;
  }
};
// compute.SetOptions (scalar_set_lookup.go:62-65) + kernels.NullMatchingBehavior (kernels/scalar_set_lookup.go:31-38)
enum NullMatchingBehavior { NullMatchingMatch = 0, NullMatchingSkip = 1, NullMatchingEmitNull = 2, NullMatchingInconclusive = 3 };
struct SetOptions : FunctionOptions {
  ArrayDataPtr ValueSet;  // device array (ArrayDatum)
  NullMatchingBehavior NullBehavior = NullMatchingMatch;
  const char* TypeName() const override { return "SetOptions"; }
};
// kernels.SortKey / compute.SortOptions (kernels/vector_sort.go:27-58, compute/vector_sort.go:88-104): one key per
// sort column; for a bare array only the first key is used and ColumnIndex is ignored
enum SortOrder { SortOrderAscending = 0, SortOrderDescending = 1 };
enum NullPlacement { SortNullsAtEnd = 0, SortNullsAtStart = 1 };
struct SortKey { int ColumnIndex = 0; SortOrder Order = SortOrderAscending; NullPlacement Placement = SortNullsAtEnd; };
struct SortOptions : FunctionOptions {
  std::vector<SortKey> Keys;
  const char* TypeName() const override { return "SortKeys"; }
};
// kernels.CumulativeOptions (vector_cumulative.go:30-39): nil Start = zero of the input type
struct CumulativeOptions : FunctionOptions { ScalarPtr Start; bool SkipNulls = false; const char* TypeName() const override { return "CumulativeOptions"; } };
// kernels.RoundMode / RoundOptions / RoundToMultipleOptions (rounding.go:37-66, 93-104; defaults arithmetic.go:78-80)
enum RoundMode { RoundDown = 0, RoundUp, RoundTowardsZero, RoundTowardsInfinity, RoundHalfDown, RoundHalfUp, RoundHalfTowardsZero,
                 RoundHalfTowardsInfinity, RoundHalfToEven, RoundHalfToOdd };
struct RoundOptions : FunctionOptions { int64_t NDigits = 0; RoundMode Mode = RoundHalfToEven; const char* TypeName() const override { return "RoundOptions"; } };
struct RoundToMultipleOptions : FunctionOptions { ScalarPtr Multiple; RoundMode Mode = RoundHalfToEven; const char* TypeName() const override { return "RoundToMultipleOptions"; } };
struct CompareFilterSumOptions : FunctionOptions { int cmpop = AH_CMP_GT; const char* TypeName() const override { return "CompareFilterSumOptions"; } };

// compute.Datum (datum.go:35-40): array or scalar
enum class DatumKind { None, Scalar, Array, Chunked, Record };
struct Datum {
  DatumKind kind = DatumKind::None;
  ArrayDataPtr array;
  ScalarPtr scalar;
  std::vector<ArrayDataPtr> chunks;       // ChunkedDatum (datum.go:186-230): arrow.Chunked = type + chunk list
  const DataType* chunked_type = nullptr;
  static Datum Of(ArrayDataPtr a) { Datum d; d.kind = DatumKind::Array; d.array = std::move(a); return d; }
  static Datum Of(ScalarPtr s) { Datum d; d.kind = DatumKind::Scalar; d.scalar = std::move(s); return d; }
  // RecordDatum (datum.go:232-260): equal-length named columns; `chunks` holds the columns
  std::vector<std::string> names;
  int64_t num_rows = 0;
  static Datum OfRecord(std::vector<std::string> n, std::vector<ArrayDataPtr> cols, int64_t rows) {
    Datum d; d.kind = DatumKind::Record; d.names = std::move(n); d.chunks = std::move(cols); d.num_rows = rows; return d;
  }
  static Datum OfChunks(const DataType* t, std::vector<ArrayDataPtr> c) {
    Datum d; d.kind = DatumKind::Chunked; d.chunked_type = t; d.chunks = std::move(c); return d;
  }
  const DataType* type() const {
    return kind == DatumKind::Array ? array->type : kind == DatumKind::Scalar ? scalar->type : kind == DatumKind::Chunked ? chunked_type : nullptr;
  }
  int64_t Len() const {
    if (kind == DatumKind::Chunked) { int64_t n = 0; for (auto& c : chunks) n += c->length; return n; }
    if (kind == DatumKind::Record) return num_rows;
    return kind == DatumKind::Array ? array->length : 1;
  }
  bool IsArrayLike() const { return kind == DatumKind::Array || kind == DatumKind::Chunked; }
};

// array.Concatenate (arrow/array/concat.go:41-110) for the layouts of this path, done in HBM: validity and
// boolean data through the bit-offset bitmap copy, fixed-width values with one device copy per chunk,
// var-length offsets rebased with the arr∘scalar Add kernel.  A single chunk is returned as is.
Status Concatenate(Session* s, const std::vector<ArrayDataPtr>& chunks, const DataType* type, ArrayDataPtr* out);
// array.NewSliceData: zero-copy view
ArrayDataPtr SliceData(const ArrayDataPtr& a, int64_t off, int64_t len);

class FunctionRegistry;
// compute.ExecCtx (executor.go:46-64) — the registry travels with the context, so a child
// registry carrying the GPU kernels is selected per call exactly like SetExecCtx does in Go
struct ExecCtx {
  FunctionRegistry* Registry = nullptr;
  Session* session = nullptr;
  // ExecCtx.ChunkSize (executor.go:47-50: "the maximum length of an ExecSpan") for HOST-resident arguments, in bytes of the widest
  // column of a span: the piece that is uploaded, computed and downloaded while its neighbours are (0: 32 MiB).  Device-resident
  // arguments are never cut: one launch over the whole column is the right shape there.
  int64_t ChunkBytes = 0;
  // ahc_import_host keeps an array on the host from this many value bytes on; smaller ones are uploaded at import
  int64_t HostThresholdBytes = (int64_t)64 << 20;
};

// ---- host-resident arguments (hoststream.cc) --------------------------------------------------------------------------------
// CallFunction for a call with at least one ArrayData::on_host argument.  *handled = true: `out` holds the result (host-resident
// for the streamed scalar kernels and Filter); false: the function or the argument shapes are not streamable — the caller uploads
// the arguments whole (MaterializeOnDevice) and takes the usual path.
Status CallHostResident(ExecCtx* ctx, const std::string& name, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out, bool* handled);
Status MaterializeOnDevice(Session* s, const ArrayDataPtr& host, ArrayDataPtr* out);
// every host-resident array datum of `values` replaced by a device-resident copy (what CallFunction, the expression executors
// and the record / chunked entry points do before anything that cannot stream)
Status MaterializeAllOnDevice(Session* s, std::vector<Datum>* values);
Status SumHostResident(ExecCtx* ctx, const ArrayData& a, double* f64, int64_t* i64, uint64_t* u64);
// cumulativeStartValue (vector_cumulative.go:92-116): the Start option safe-cast to the column's type, little-endian in out[8] (kernels.cc)
Status SafeCastCumulativeStart(const Scalar& start, const DataType* to, uint8_t out[8]);

enum class FuncKind { Scalar, Vector, Meta };  // functions.go:88-100
struct Arity { int NArgs; bool IsVarArgs; };

class Function {  // functions.go:30-41
 public:
  Function(std::string name, Arity arity, FuncKind kind, const FunctionOptions* default_opts = nullptr)
      : name_(std::move(name)), arity_(arity), kind_(kind), default_opts_(default_opts) {}
  virtual ~Function() = default;
  const std::string& Name() const { return name_; }
  FuncKind Kind() const { return kind_; }
  Arity GetArity() const { return arity_; }
  const FunctionOptions* DefaultOptions() const { return default_opts_; }
  void SetDefaultOptions(const FunctionOptions* o) { default_opts_ = o; }
  virtual int NumKernels() const = 0;
  virtual Status Execute(ExecCtx* ctx, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out) = 0;
  Status CheckArity(size_t nargs) const;  // functions.go:130-146 checkArity
 protected:
  std::string name_;
  Arity arity_;
  FuncKind kind_;
  const FunctionOptions* default_opts_;
};

class ScalarFunction : public Function {  // functions.go:239-290
 public:
  ScalarFunction(std::string name, Arity arity) : Function(std::move(name), arity, FuncKind::Scalar) {}
  Status AddKernel(exec::ScalarKernel k);
  int NumKernels() const override;   // a flipped comparison counts the kernels it shares with its base (scalar_compare.go:73-99)
  // funcImpl.Kernels() (functions.go:220-226): live pointers — the in-place swap route
  std::vector<exec::ScalarKernel*> Kernels();
  Status DispatchExact(const std::vector<const DataType*>& types, const exec::ScalarKernel** out) const;  // :199-218
  Status Execute(ExecCtx* ctx, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out) override;
  // set for "less"/"less_equal": swap the two arguments and run `flipped` (scalar_compare.go:73-99)
  std::string flipped_of;
  // arithmeticFunction.DispatchBest (arithmetic.go:112-142) / compareFunction.DispatchBest
  // (scalar_compare.go:37-63): when no kernel matches exactly, both arguments are promoted to
  // commonNumeric (utils.go:178-240) and implicitly SAFE-cast (exec.go:105-114)
  bool promote_numeric = false;
  bool promote_to_float = false;  // unary floating-point functions: integer arguments are cast to float64
  Status DispatchBest(std::vector<const DataType*>* types, const exec::ScalarKernel** out) const;
 private:
  std::vector<exec::ScalarKernel> kernels_;
};

class VectorFunction : public Function {  // functions.go:290-360
 public:
  VectorFunction(std::string name, Arity arity, const FunctionOptions* def = nullptr) : Function(std::move(name), arity, FuncKind::Vector, def) {}
  Status AddKernel(exec::VectorKernel k);
  int NumKernels() const override { return (int)kernels_.size(); }
  Status DispatchExact(const std::vector<const DataType*>& types, const exec::VectorKernel** out) const;
  Status Execute(ExecCtx* ctx, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out) override;
  // What the function returns for chunked input — the visible outcome of VectorKernel.CanExecuteChunkWise /
  // ExecChunked / OutputChunked / Finalize in the reference (exec/kernel.go:697-725, executor.go:896-1080):
  enum class Chunked {
    SameLength,   // one output row per input row, re-chunked like the input (dictionary_encode: vector_hash.go:905)
    SingleArray,  // state accumulated over the chunks, one array out (unique: vector_hash.go:890, Finalize; sort_indices)
    SingleChunk,  // ExecChunked returning one result (cumulative_sum: vector_cumulative.go:368-391)
    Filter,       // chunk-wise with data-dependent lengths (array_filter)
    Take,         // the take meta function's rules (selection.go:195-330)
  };
  Chunked chunked = Chunked::SameLength;
 private:
  std::vector<exec::VectorKernel> kernels_;
};

using MetaImpl = std::function<Status(ExecCtx*, const FunctionOptions*, const std::vector<Datum>&, Datum*)>;
class MetaFunction : public Function {  // functions.go:360-420
 public:
  MetaFunction(std::string name, Arity arity, const FunctionOptions* def, MetaImpl impl)
      : Function(std::move(name), arity, FuncKind::Meta, def), impl_(std::move(impl)) {}
  int NumKernels() const override { return 0; }
  Status Execute(ExecCtx* ctx, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out) override;
 private:
  MetaImpl impl_;
};

class FunctionRegistry {  // registry.go:30-120
 public:
  explicit FunctionRegistry(FunctionRegistry* parent = nullptr) : parent_(parent) {}
  // AddFunction(fn, allowOverwrite): ErrKey "already have a function registered with name …"
  Status AddFunction(std::shared_ptr<Function> fn, bool allow_overwrite);
  Status AddAlias(const std::string& target, const std::string& source);
  Function* GetFunction(const std::string& name) const;  // nullptr ⇔ (nil, false)
  std::vector<std::string> GetFunctionNames() const;
  int NumFunctions() const;
  bool CanAddFunction(const std::shared_ptr<Function>& fn, bool allow_overwrite) const;
 private:
  FunctionRegistry* parent_;
  std::map<std::string, std::shared_ptr<Function>> fns_;
};
FunctionRegistry* GetFunctionRegistry();  // registry.go:47 — process default, GPU kernels pre-registered
std::unique_ptr<FunctionRegistry> NewChildRegistry(FunctionRegistry* parent);  // registry.go:69-73

// compute.CallFunction (exec.go:191): "function '%s' not found" → KeyError
Status CallFunction(ExecCtx* ctx, const std::string& name, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out);

// registration entry points (≙ RegisterScalarArithmetic, RegisterScalarComparisons,
// RegisterScalarBoolean, RegisterVectorSelection, RegisterVectorHash in arrow/compute)
void RegisterScalarArithmetic(FunctionRegistry* reg);
void RegisterScalarComparisons(FunctionRegistry* reg);
void RegisterScalarBoolean(FunctionRegistry* reg);
void RegisterVectorCumulative(FunctionRegistry* reg);
void RegisterScalarCast(FunctionRegistry* reg);
void RegisterScalarSetLookup(FunctionRegistry* reg);
void RegisterVectorSort(FunctionRegistry* reg);
const DataType* CommonNumeric(const std::vector<const DataType*>& types);  // utils.go:178-240; nullptr if none
// compute.CastDatum / CastArray (cast.go:917-935)
Status CastDatum(ExecCtx* ctx, const Datum& in, const CastOptions& opts, Datum* out);
void RegisterVectorSelection(FunctionRegistry* reg);
void RegisterVectorHash(FunctionRegistry* reg);
void RegisterFusedExtensions(FunctionRegistry* reg);

// ---- compute.Expression (arrow/compute/expression.go:59-78,596-620) and its executor
// (arrow/compute/exprs/exec.go:440-700 ExecuteScalarExpression / executeScalarBatch) --------
struct Expression;
using ExprPtr = std::shared_ptr<Expression>;
struct Expression {
  enum Kind { LITERAL, FIELD_REF, CALL } kind = LITERAL;
  Datum literal;                 // LITERAL: a scalar datum
  int field_index = -1;          // FIELD_REF: position in the input batch …
  std::string field_name;        // … or its name (resolved against the batch's names)
  std::string function;          // CALL
  std::vector<ExprPtr> args;
  std::shared_ptr<FunctionOptions> options;
  std::string ToString() const;  // Call.String(): "add(a, multiply(b, 2))"
};
ExprPtr NewLiteral(ScalarPtr s);
ExprPtr NewFieldRef(const std::string& name);   // expression.go:610
ExprPtr NewRef(int index);                      // expression.go:605 (FieldRef by position)
ExprPtr NewCall(const std::string& name, std::vector<ExprPtr> args, std::shared_ptr<FunctionOptions> opts = nullptr);  // :617

// the input record batch: named, equal-length device arrays (compute.ExecBatch)
struct ExecBatch {
  std::vector<std::string> names;
  std::vector<Datum> values;
  int64_t len = 0;
};
// Evaluates `expr` over `batch`.  fuse = true: if the whole tree is made of fusible scalar
// calls (arithmetic, comparisons, plain boolean ops) over same-typed operands, it runs as ONE
// JIT-compiled kernel (ah_expr_*); otherwise — and always with fuse = false — it is evaluated
// exactly like executeScalarBatch: one CallFunction per call node, intermediates materialised.
// Both routes give bit-identical results.  *fused_out (nullable) reports which one ran.
Status ExecuteScalarExpression(ExecCtx* ctx, const ExprPtr& expr, const ExecBatch& batch, Datum* out, bool fuse = true,
                               bool* fused_out = nullptr);

// ---- the Substrait front end (substrait.cc): exprs.ExecuteScalarSubstrait (arrow/compute/exprs/exec.go:465-488) -----------------
// a parsed substrait.ExtendedExpression: the base schema (types this layer does not carry are nullptr + their Substrait name) and
// one Expression per referred expression (or the status that says why it could not be built)
struct SubstraitExtended {
  std::vector<std::string> names;
  std::vector<const DataType*> types;
  std::vector<std::string> type_names;
  std::vector<ExprPtr> exprs;
  std::vector<Status> expr_status;
  std::vector<std::string> out_names;
};
Status ParseSubstraitExtended(const uint8_t* bytes, int64_t len, SubstraitExtended* out);
// cols: the input's columns (arrays or scalars) — by position in the base schema when col_names is empty, else matched by name
// (missing fields become null scalars, makeExecBatch exec.go:384-438)
Status ExecuteScalarSubstrait(ExecCtx* ctx, const uint8_t* bytes, int64_t len, const std::vector<Datum>& cols, const std::vector<std::string>& col_names,
                              Datum* out, bool fuse = true, bool* fused_out = nullptr);

}  // namespace compute

// ---- arrow/math (float64.go:25-39, int64.go, uint64.go) ---------------------------------
namespace math {
struct Float64Funcs { Status Sum(Session* s, const ArrayData& a, double* out) const; };
struct Int64Funcs { Status Sum(Session* s, const ArrayData& a, int64_t* out) const; };
struct Uint64Funcs { Status Sum(Session* s, const ArrayData& a, uint64_t* out) const; };
extern const Float64Funcs Float64;
extern const Int64Funcs Int64;
extern const Uint64Funcs Uint64;
}  // namespace math

}  // namespace arrowhip
