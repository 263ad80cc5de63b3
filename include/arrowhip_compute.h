/*
 * arrowhip_compute.h — C ABI of libarrowhip_compute.so: the ARRAY-LEVEL face of the MI355X execution layer.
 *
 * libarrowhip.so (arrowhip.h) replaces the reference's leaves (pointers + lengths).  This library sits one level up:
 * it carries the part of arrow-go's arrow/compute that a Go host would otherwise keep — function registry, scalar /
 * vector executors (null propagation, preallocation, chunked arguments, DispatchBest casts), the Arrow IPC stream
 * reader — next to the device, so that a chain of calls never leaves HBM.  Whole arrays enter and leave through the
 * Arrow C Data Interface and the Arrow C Device Data Interface, i.e. the structs arrow-go's own arrow/cdata package
 * speaks (arrow/cdata/abi.h:50-128, cdata.go:69-72, interface.go:73-153).
 *
 *   ahc_call(session, "add", "", 2, args, &out)   ≙   compute.CallFunction(ctx, "add", nil, a, b)   compute/exec.go:191
 *
 * A Go binding (INTEGRATION.md §"array-level route") is `#include "arrowhip_compute.h"` + cdata.ExportArrowArray /
 * cdata.ImportCArrayWithType on either side of ahc_import / ahc_export.
 *
 * Conventions
 *   - every int-returning function returns 0 or a status class (AHC_*); the message is ahc_last_error(session) until
 *     the next call on that session.  Functions returning a count return -1 for "not that kind of datum".
 *   - an ahc_datum is an opaque, reference-counted handle (compute.Datum, compute/datum.go:52-60: scalar, array,
 *     chunked array, record batch) whose buffers live in HBM; release every handle you were given with
 *     ahc_datum_release.  A datum may outlive its session handle: device memory is returned when the last datum goes.
 *   - one call at a time per session (it owns one ah_ctx: one compute stream, one copy stream); sessions are
 *     independent and may be used from different threads.
 *   - nothing here retains a host pointer after the call returns.
 */
#ifndef ARROWHIP_COMPUTE_C_H
#define ARROWHIP_COMPUTE_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Arrow C Data Interface / C Device Data Interface (verbatim ABI: arrow/cdata/abi.h:50-79, 95-128) ------------- */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif /* ARROW_C_DATA_INTERFACE */

#ifndef ARROW_C_DEVICE_DATA_INTERFACE
#define ARROW_C_DEVICE_DATA_INTERFACE
typedef int32_t ArrowDeviceType;
#define ARROW_DEVICE_CPU 1
#define ARROW_DEVICE_ROCM 10
#define ARROW_DEVICE_ROCM_HOST 11
struct ArrowDeviceArray {
  struct ArrowArray array;
  int64_t device_id;
  ArrowDeviceType device_type;
  void* sync_event; /* ARROW_DEVICE_ROCM: hipEvent_t* or NULL */
  int64_t reserved[3];
};
#endif /* ARROW_C_DEVICE_DATA_INTERFACE */

/* ---- status classes: the sentinel errors of arrow/errors.go the Go side wraps with fmt.Errorf("%w: …") ------------ */
#define AHC_OK 0
#define AHC_EINVALID 1  /* arrow.ErrInvalid (also "overflow") */
#define AHC_EINDEX 2    /* arrow.ErrIndex */
#define AHC_ENOTIMPL 3  /* arrow.ErrNotImplemented */
#define AHC_ETYPE 4     /* arrow.ErrType — no kernel matching input types */
#define AHC_EKEY 5      /* arrow.ErrKey — unknown function / already registered (compute/registry.go:106-135) */
#define AHC_EHIP 6      /* HIP runtime failure; there is no CPU fallback */

/* datum kinds (ahc_datum_info) — compute.DatumKind, compute/datum.go:35-49 */
#define AHC_KIND_NONE 0
#define AHC_KIND_SCALAR 1
#define AHC_KIND_ARRAY 2
#define AHC_KIND_CHUNKED 3
#define AHC_KIND_RECORD 4

typedef struct ahc_session ahc_session;       /* ah_ctx + compute.ExecCtx with its own child registry (compute/executor.go:46-112) */
typedef struct ahc_datum ahc_datum;           /* compute.Datum */
typedef struct ahc_ipc_reader ahc_ipc_reader; /* ipc.Reader (arrow/ipc/reader.go:45-70) */

/* ---- session / registry ------------------------------------------------------------------------------------------ */
int ahc_session_create(int device_id, ahc_session** out);
void ahc_session_destroy(ahc_session* s);
const char* ahc_last_error(ahc_session* s); /* never NULL; owned by the session */
/* compute.GetFunctionRegistry(): NumFunctions / GetFunction(name) != nil / len(fn.Kernels()) (registry.go:30-40) */
int ahc_num_functions(void);
int ahc_has_function(const char* name);
int ahc_function_num_kernels(const char* name);
/* fn.DispatchBest(types...) of a scalar function (functions.go:199-218; arithmeticFunction / compareFunction.DispatchBest:
 * arithmetic.go:112-142, scalar_compare.go:37-63) or DispatchExact of a vector function, without executing: out_type_ids receives
 * the argument types of the chosen kernel (type ids as in ahc_datum_type_id), i.e. the implicit casts the executor would insert.
 * Needs no session and no GPU.  Error text into err. */
int ahc_dispatch_best(const char* function, int nargs, const int* type_ids, int* out_type_ids, char* err, int64_t cap);
/* AddAlias / AddFunction(allowOverwrite) on the session's child registry (registry.go:69-73, 97-135) */
int ahc_registry_add_alias(ahc_session* s, const char* alias, const char* existing, int allow_overwrite);

/* ---- datums ------------------------------------------------------------------------------------------------------- */
/* cdata.ImportCArrayWithType on a HOST array: buffers are uploaded (pinned staging → hipMemcpyAsync on the copy
 * stream); the producer's structs are released before returning, success or not.  Formats: the ten numeric types,
 * bool, u / z / U / Z (string / binary), tdD tdm tt* ts* tD* (temporal: stored as integers + label), and
 * dictionary-encoded arrays of those. */
int ahc_import(ahc_session* s, struct ArrowArray* arr, struct ArrowSchema* schema, ahc_datum** out);
/* cdata.ExportArrowArray: device → host copy; the caller owns (and must release) arr and schema */
int ahc_export(ahc_session* s, ahc_datum* d, struct ArrowArray* arr, struct ArrowSchema* schema);
/* HOST-RESIDENT arguments — chunked, overlapped execution (arrow_go_amd/host/hoststream.cc).  The reference's executor cuts a call
 * into spans of at most ExecCtx.ChunkSize rows (arrow/compute/executor.go:47-50, :499 iterateExecSpans); for a column that lives in
 * host memory that span loop is what hides the PCIe link.  ahc_import_host is ahc_import WITHOUT the upload for flat fixed-width
 * columns of at least "host_threshold_bytes" (64 MiB) value bytes: the producer's buffers stay where they are (pin them for the copies
 * to overlap) and are released with the datum.  ahc_call then streams such arguments span by span — upload k + 1, kernel k, download
 * k − 1 on three streams — for add / sub / subtract / multiply [+ _unchecked], equal … less_equal (array ∘ array, array ∘ scalar,
 * scalar ∘ array of one numeric type; a null scalar is handled as the whole-array kernels handle it: only checked integer add / sub
 * leave the payload untouched), filter / array_filter, cast (numeric → numeric, safe or not) and cumulative_sum[_checked] of integer
 * columns with nulls skipped or none (the running sum is carried from span to span); ahc_math_sum sums them chunk by chunk with one
 * final reduction.  The bytes of a streamed result — payload under nulls, validity, its last byte's tail bits — are those of the
 * whole-array call.  Results of streamed calls are host-resident too (pinned memory of the session's; ahc_export hands the buffers out
 * without a copy, ahc_datum_buffers returns HOST pointers for them).  Every other function uploads a host-resident argument whole the
 * first time it meets it; the copy stays with the datum, which counts as device-resident from then on (ahc_datum_on_host → 0,
 * ahc_datum_buffers → the device pointers, later calls use the copy instead of streaming).  A column larger than the free HBM can be
 * added, compared, filtered, cast and summed this way: only three spans of it are on the device at any time.
 * ahc_session_set_option: "chunk_bytes" (bytes of the widest column per span: ExecCtx.ChunkSize's role; 0 = 32 MiB),
 * "host_threshold_bytes". */
int ahc_import_host(ahc_session* s, struct ArrowArray* arr, struct ArrowSchema* schema, ahc_datum** out);
int ahc_datum_on_host(ahc_datum* d);
int ahc_session_set_option(ahc_session* s, const char* name, int64_t value);
/* Arrow C Device Data Interface.  ARROW_DEVICE_ROCM on the session's device: ZERO COPY both ways — import wraps the
 * producer's buffers (its release callback runs when the last datum referencing them is gone; the compute stream
 * waits for sync_event), export hands the datum's HBM buffers out (stream synchronised, sync_event = NULL; the export
 * keeps them alive until the consumer calls release).  ARROW_DEVICE_CPU / ROCM_HOST imports take the upload path.
 * Layouts: fixed-width (2 buffers) and string / binary (3 buffers: validity, offsets, data). */
int ahc_import_device(ahc_session* s, struct ArrowDeviceArray* darr, struct ArrowSchema* schema, ahc_datum** out);
int ahc_export_device(ahc_session* s, ahc_datum* d, struct ArrowDeviceArray* out, struct ArrowSchema* schema);
/* scalar.Scalar of arrow.Type `type_id` (ids as in arrowhip.h; 1 = BOOL); value8: little-endian payload, 8 bytes */
int ahc_scalar(ahc_session* s, int type_id, int valid, const void* value8, ahc_datum** out);
/* temporal label of a datum as its C Data format ("tsu:UTC", "tdD", …; "" = plain); valid until the next call on
 * this thread.  ahc_scalar_set_logical labels a scalar made by ahc_scalar (AHC_ETYPE if the storage width differs). */
const char* ahc_datum_logical(ahc_datum* d);
int ahc_scalar_set_logical(ahc_session* s, ahc_datum* d, const char* format);
/* kind, arrow.Type id, Len(), null count (arrays; -1 = unknown), and for scalars validity + payload */
int ahc_datum_info(ahc_datum* d, int* kind, int* type_id, int64_t* length, int64_t* null_count, int* scalar_valid,
                   void* scalar_value8);
/* the device pointers behind an array datum (validity may be NULL; for binary types `data` is the offsets buffer) */
int ahc_datum_buffers(ahc_datum* d, void** validity, void** data);
void ahc_datum_release(ahc_datum* d);
/* compute.ChunkedDatum from array datums of one type (datum.go:186-230); the arrays stay the caller's */
int ahc_chunked_from_arrays(ahc_session* s, int type_id, int n, ahc_datum** arrays, ahc_datum** out);
int ahc_datum_num_chunks(ahc_datum* d);
int ahc_datum_chunk(ahc_session* s, ahc_datum* d, int i, ahc_datum** out);
/* compute.RecordDatum from equal-length array datums (datum.go:232-262) */
int ahc_record_from_arrays(ahc_session* s, int n, const char* const* names, ahc_datum** arrays, ahc_datum** out);
int ahc_record_num_columns(ahc_datum* d);
int ahc_record_column(ahc_session* s, ahc_datum* d, int i, const char** name, ahc_datum** out);

/* ---- calls -------------------------------------------------------------------------------------------------------- */
/* compute.CallFunction(ctx, name, opts, args...) (compute/exec.go:191-199).  `options` is "key=value;key=value" (NULL
 * or "" = the function's defaults) for the option structs of the registered functions:
 *   FilterOptions  null_selection_behavior=drop|emit_null; output_sizing=exact|worst_case (worst_case: the output is allocated
 *                  for the input's length and the kernel runs in ONE call — ah_filter_primitive_once — instead of count → allocate → fill)
 *   TakeOptions    bounds_check=0|1
 *   CastOptions    to_type=<type name>; to_logical=<C Data format>; safe=0; allow_int_overflow= allow_float_truncate=
 *                  allow_time_truncate= allow_time_overflow=0|1
 *   SetOptions     value_set=@<hex address of an array ahc_datum>; null_matching_behavior=match|skip|emit_null|inconclusive
 *   SortOptions    sort_keys=<col>:<asc|desc>:<at_end|at_start>,…    ArraySortOptions  order= null_placement=
 *   CumulativeSumOptions  start=<number>; skip_nulls=0|1           RoundOptions  ndigits= round_mode= multiple=
 *   DictionaryEncodeOptions  null_encoding_behavior=mask|encode
 * Errors carry the reference's texts ("overflow", "%d out of bounds", "no kernel matching input types (…)", …). */
int ahc_call(ahc_session* s, const char* name, const char* options, int nargs, ahc_datum** args, ahc_datum** out);
/* math.Float64.Sum / Int64.Sum / Uint64.Sum (arrow/math/float64.go:34-39 …): the pointer matching the array's type
 * receives the result, validity ignored as in the reference */
int ahc_math_sum(ahc_session* s, ahc_datum* d, double* f64, int64_t* i64, uint64_t* u64);
/* exprs.ExecuteScalarExpression over a batch of columns (compute/exprs/exec.go:542-700).  text: call := name '(' arg
 * {',' arg} ')'; arg := call | '$'N (column N) | '#'N (literal N, a scalar datum).  fuse != 0: one generated kernel for
 * the whole tree where the generator covers it (*fused_out says whether), else call by call — same bytes either way. */
int ahc_expr_eval(ahc_session* s, const char* text, int ncols, ahc_datum** cols, int nlits, ahc_datum** lits, int fuse,
                  ahc_datum** out, int* fused_out);
/* The same executor fed with the reference's own tree shape instead of text: compute.Expression is Literal{datum} |
 * Parameter{field reference} | Call{funcName, args, options} (arrow/compute/expression.go:52-78, 278-290, NewLiteral :596,
 * NewFieldRef / NewRef :610, NewCall :617).  nodes[] holds the tree in post-order — a call's arguments come before it, the root is
 * the LAST node; a field reference is a column position, or index = -1 and a name resolved against col_names (FieldRef.FindOne:
 * "no match for field reference"); a literal is a position in lits[] (scalar datums); a call carries the options text of
 * ahc_call.  Same fusion rule and same bytes as ahc_expr_eval. */
#define AHC_EXPR_LITERAL 0
#define AHC_EXPR_FIELD_REF 1
#define AHC_EXPR_CALL 2
typedef struct ahc_expr_node {
  int32_t kind;
  int32_t index;          /* LITERAL: position in lits[]; FIELD_REF: column position or -1 (by name) */
  const char* name;       /* CALL: function name; FIELD_REF by name */
  const char* options;    /* CALL: options text (nullable) */
  int32_t nargs;          /* CALL */
  const int32_t* args;    /* CALL: positions in nodes[] of the arguments, each smaller than this node's */
} ahc_expr_node;
int ahc_expr_eval_tree(ahc_session* s, const ahc_expr_node* nodes, int n_nodes, int ncols, ahc_datum** cols, const char* const* col_names,
                       int nlits, ahc_datum** lits, int fuse, ahc_datum** out, int* fused_out);
/* exprs.ExecuteScalarSubstrait (arrow/compute/exprs/exec.go:465-488): `bytes` is a serialized substrait.ExtendedExpression with ONE
 * referred expression — scalar functions of the default extension set (add / subtract / multiply / divide / power / sqrt / abs with
 * their `overflow` option, equal / not_equal / lt / lte / gt / gte / is_null / is_not_null / is_nan, and / or / not), root field
 * references, primitive literals (unsigned ones as arrow-go's type variations or Arrow C++'s user-defined types), casts.  cols: the
 * input's columns in the order of the base schema, or — with col_names — matched by name, missing fields being null scalars
 * (makeExecBatch, exec.go:384-438); a column whose type differs from the schema's is arrow.ErrInvalid "referenced field … was …,
 * but should have been …"; what the reference's executor refuses is refused with its error class (ErrNotImplemented: measures,
 * several referred expressions, if-then, nested references, unknown functions, SATURATE, RETURN_NULL casts; ErrInvalid: no
 * expression, unspecified cast behaviour).  All-scalar input gives a scalar datum. */
int ahc_expr_eval_substrait(ahc_session* s, const uint8_t* bytes, int64_t len, int ncols, ahc_datum** cols, const char* const* col_names,
                            int fuse, ahc_datum** out, int* fused_out);

/* The same reader without a device: parses `bytes` and renders what it understood into out (NUL-terminated, truncated to cap) as
 * "name:type,…|output_name=expression|…" — literals as type(hex payload) or type(null), field references as $index, casts as cast(x -> type unsafe), an
 * expression the executor would refuse as "!" + its error text.  A malformed message returns the status code with the error text in out.
 * Needs no session and no GPU: the CPU tests of the wire-format reader (truncated and random bytes included) go through it. */
int ahc_substrait_inspect(const uint8_t* bytes, int64_t len, char* out, int64_t cap);

/* ---- Arrow IPC → HBM (ipc.NewReader / Reader.Next, arrow/ipc/reader.go:97-300) ------------------------------------ */
/* stream or file format, uncompressed bodies; each RecordBatch body goes to the device in one copy and the columns are
 * slices of it.  `bytes` must stay valid until ahc_ipc_close. */
int ahc_ipc_open(ahc_session* s, const uint8_t* bytes, int64_t len, ahc_ipc_reader** out);
void ahc_ipc_close(ahc_ipc_reader* r);
int ahc_ipc_num_fields(ahc_ipc_reader* r);
int ahc_ipc_field(ahc_ipc_reader* r, int i, const char** name, int* type_id, int* nullable);
/* next record batch: `columns` receives ahc_ipc_num_fields() array datums; *rows = -1 at end of stream */
int ahc_ipc_next(ahc_ipc_reader* r, ahc_datum** columns, int64_t* rows);
int64_t ahc_ipc_bytes_uploaded(ahc_ipc_reader* r);
/* walks a stream WITHOUT a device: "hex(name):type_id:nullable:hex(logical),…|rows,rows,…" into out (NUL-terminated),
 * or the error text with the status returned */
int ahc_ipc_inspect(const uint8_t* bytes, int64_t len, char* out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* ARROWHIP_COMPUTE_C_H */
