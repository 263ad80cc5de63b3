/*
 * arrowhip.h — C ABI of libarrowhip.so: MI355X (gfx950) kernels for the hot path of
 * arrow-go's arrow/compute + arrow/math kernel registry.
 *
 * This is the drop-in boundary.  Each entry point replaces one leaf the Go code
 * reaches today through c2goasm Plan9 assembly (or a pure-Go loop); the citation on
 * each declaration is the reference interface it replaces (paths relative to the
 * arrow-go tree).  A Go maintainer binds these with cgo (INTEGRATION.md shows the
 * stubs); nothing here mentions torch, C++ or HIP types.
 *
 * Conventions
 *   - every function returns an int status: AH_OK or an error class; the message is
 *     available from ah_last_error(ctx) until the next call on that ctx.
 *   - pointers named *_host are host memory, read or written before the call
 *     returns (the library never retains a host pointer — the cgo pointer rule);
 *     all other data pointers are DEVICE pointers (from ah_buf_alloc, or any
 *     hipMalloc'd / torch-allocated memory on the ctx's device).
 *   - bitmaps are Arrow validity/boolean bitmaps: LSB-first, bit i of the logical
 *     array is bit ((off + i) & 7) of byte ((off + i) >> 3).  A NULL validity
 *     pointer means "all valid" (ArraySpan.Buffers[0].Buf == nil).
 *   - kernels are enqueued on the ctx's compute stream and the call returns without
 *     waiting, EXCEPT when the signature has a *_host output: those calls
 *     synchronise the stream before returning.  ah_sync() waits for everything (and returns
 *     AH_EHIP if a kernel reported since the last synchronisation that it gave up waiting for
 *     another workgroup — the one-pass cumulative sums' bounded look-back; that call's output is invalid).
 *   - an ah_ctx may be used from any OS thread (every entry point selects the
 *     ctx's device first — Go's executor goroutine may migrate between threads,
 *     compute/exec.go:165) but serves one call at a time.
 *   - buffers must not alias unless stated; sizes are in ELEMENTS unless named
 *     nbytes / nbits.
 */
#ifndef ARROWHIP_H
#define ARROWHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status classes ------------------------------------------------------ */
#define AH_OK 0
#define AH_EINVALID 1   /* arrow.ErrInvalid */
#define AH_EINDEX 2     /* arrow.ErrIndex ("<v> out of bounds") */
#define AH_EOVERFLOW 3  /* arrow.ErrInvalid: "overflow" (kernels/base_arithmetic.go errOverflow) */
#define AH_EHIP 4       /* HIP runtime failure */
#define AH_ENOTIMPL 5   /* arrow.ErrNotImplemented */

/* arrow.Type ids passed verbatim — arrow/datatype.go:36-72 == kernels/_lib/types.h:20-34 */
#define AH_UINT8 2
#define AH_INT8 3
#define AH_UINT16 4
#define AH_INT16 5
#define AH_UINT32 6
#define AH_INT32 7
#define AH_UINT64 8
#define AH_INT64 9
#define AH_FLOAT32 11
#define AH_FLOAT64 12

/* ArithmeticOp — kernels/base_arithmetic.go:37-82 == kernels/_lib/base_arithmetic.cc:31-74 */
#define AH_OP_ADD 0
#define AH_OP_SUB 1
#define AH_OP_MUL 2
#define AH_OP_ABS 4
#define AH_OP_NEGATE 5
#define AH_OP_SIGN 20
#define AH_OP_DIV 3               /* ah_arithmetic_ext ops: the reference's ArithmeticOp numbering where it has one … */
#define AH_OP_SQRT 6
#define AH_OP_DIV_CHECKED 24
#define AH_OP_ABS_CHECKED 25
#define AH_OP_NEGATE_CHECKED 26
#define AH_OP_SQRT_CHECKED 27
#define AH_OP_POWER 7             /* OpPower / OpPowerChecked (base_arithmetic.go:46,71) */
#define AH_OP_POWER_CHECKED 28
#define AH_OP_SHIFT_LEFT 64       /* … and private numbers for the shift / bit-wise kernels (scalar_arithmetic.go) */
#define AH_OP_SHIFT_LEFT_CHECKED 65
#define AH_OP_SHIFT_RIGHT 66
#define AH_OP_SHIFT_RIGHT_CHECKED 67
#define AH_OP_BIT_AND 68
#define AH_OP_BIT_OR 69
#define AH_OP_BIT_XOR 70
#define AH_OP_BIT_NOT 71
#define AH_OP_FLOOR 72             /* floor / ceil / trunc (kernels/rounding.go:180-187, 748-775) */
#define AH_OP_CEIL 73
#define AH_OP_TRUNC 74
#define AH_OP_ADD_CHECKED 21
#define AH_OP_SUB_CHECKED 22
#define AH_OP_MUL_CHECKED 23

/* cmpop — kernels/_lib/scalar_comparison.cc:172-178; LESS/LESS_EQUAL are the
 * caller's operand swap, exactly as compute/scalar_compare.go:73-99 does it. */
#define AH_CMP_EQ 0
#define AH_CMP_NE 1
#define AH_CMP_GT 2
#define AH_CMP_GE 3
/* operand shapes: array∘array, array∘scalar, scalar∘array (kernels/helpers.go:193-236) */
#define AH_SHAPE_AA 0
#define AH_SHAPE_AS 1
#define AH_SHAPE_SA 2
/* bitOp — arrow/bitutil/bitmaps.go:494-521 */
#define AH_BIT_AND 0
#define AH_BIT_OR 1
#define AH_BIT_XOR 2
#define AH_BIT_AND_NOT 3
#define AH_BIT_XNOR 4
/* Kleene ops — compute/scalar_bool.go:94-110 */
#define AH_KLEENE_AND 0
#define AH_KLEENE_OR 1
#define AH_KLEENE_AND_NOT 2
/* NullSelectionBehavior — kernels/vector_selection.go:34-39 */
#define AH_DROP_NULLS 0
#define AH_EMIT_NULLS 1

/* ---- context ------------------------------------------------------------- */
typedef struct ah_ctx ah_ctx; /* device id, compute stream, copy stream, scratch arena */

/* Own streams (compute + copy). */
int ah_ctx_create(int device_id, ah_ctx** out);
/* Borrow an existing hipStream_t as the compute stream (e.g. torch's current
 * stream, so torch.distributed collectives order naturally after our kernels). */
int ah_ctx_create_on_stream(int device_id, void* hip_stream, ah_ctx** out);
void ah_ctx_destroy(ah_ctx* ctx);
const char* ah_last_error(ah_ctx* ctx); /* never NULL; owned by ctx */
/* measurement / test switches of one context; none changes a result.  "nt" (nontemporal streaming accesses 0|1),
 * "blocks_per_cu" (grid cap of grid-stride kernels, 0 = each kernel's default), "take_binned" (0 never, 1 auto, 2 whenever
 * legal), "take_window_log2" (bytes of values per bin), "take_hint_cache" (1, the default of ah_ctx_create: the neighbour sample that picks a Take's path is kept for the
 * Take that directly follows with the same index vector — one vector gathers all columns of a record batch —, dropped like the filter cache by every other
 * compute entry point and by uploads / copies / memsets into the vector, sampled again on every 32nd use; every path returns the same bytes, so an entry
 * gone stale behind the library's back costs speed, never results; 0, the default of ah_ctx_create_on_stream: sample on every call), "take_gather_wg_per_cu", "take_gather_load", "scan_segment_log2",
 * "hash_direct" (unique / dictionary_encode: 0 ids in a separate pass … 2 default, 3 without the re-packed table),
 * "groupby_partition" (hash + sum: 0 id-based path only, 1 auto, k >= 5 always 2^(k-2) partitions, 2 sort-based, 3 / 4 two levels,
 * -2 no cut), "groupby_keys" (expected keys per partition the auto choice aims at, default 1280), "sort_msd" (sort_indices:
 * 0 LSD passes only, 1 auto), "groupby_seed" (hash + sum with few groups: 1 the workgroups' LDS tables start from the keys a quick look
 * found and are added up slot by slot, 0 empty tables merged with atomics), "groupby_lean" (same path: 0 keep a pending group per lane,
 * 1 leave it out when neighbouring rows rarely share a key, 2 always leave it out), "groupby_reserve" (partitioned hash + sum: 1 the
 * scatter reserves its runs in per-(partition, XCD) regions sized from the key sample — no histogram pass —, 0 histogram first), "scan_onepass" (cumulative_sum of 4- /
 * 8-byte integers, and of unchecked 2-byte ones without nulls: 1 one pass with decoupled look-back, the output's validity and null count written by the same kernel, 0 reduce-then-scan, 2 one pass with
 * a separate bitmap copy + popcount, 3 one pass for unchecked columns without nulls only — 2 and 3 are measurement switches), "filter_cache" (1: ah_filter_count leaves its tile prefixes for the ah_filter_primitive that
 * follows, dropped by every entry point of THIS context that may write device memory — the default of ah_ctx_create; 0: the fill
 * always recounts — the default of ah_ctx_create_on_stream, where another producer on the shared stream may rewrite the mask
 * between the two calls; set it to 1 there only if nothing else writes the mask in between).  Defaults come from the ARROWHIP_*
 * environment variables of the same names at context creation. */
int ah_ctx_set_option(ah_ctx* ctx, const char* name, int64_t value);
const char* ah_version(void);
int ah_device_count(int* n_host);

/* ---- buffers (back a Go memory.Allocator: arrow/memory/allocator.go:23-27) - */
int ah_buf_alloc(ah_ctx* ctx, size_t nbytes, void** dptr_host);      /* 256-B aligned, NOT zeroed */
int ah_buf_free(ah_ctx* ctx, void* dptr);
int ah_host_alloc_pinned(ah_ctx* ctx, size_t nbytes, void** hptr_host);
int ah_host_free_pinned(ah_ctx* ctx, void* hptr);
/* copies run on the copy stream; the compute stream is made to wait for an upload,
 * a download waits for the compute work enqueued so far.  hptr should be pinned
 * (pageable memory works but the runtime stages it). */
int ah_upload_async(ah_ctx* ctx, void* dptr, const void* hptr, size_t nbytes);
int ah_download_async(ah_ctx* ctx, void* hptr, const void* dptr, size_t nbytes);
int ah_memset_async(ah_ctx* ctx, void* dptr, int byte_value, size_t nbytes); /* arrow/memory/_lib/memory.c:20-27 */
/* device → device, ordered on the compute stream: the value-range copies of array.Concatenate
 * (arrow/array/concat.go:159-180 concatBuffers) when chunked inputs are laid end to end in HBM */
int ah_copy_async(ah_ctx* ctx, void* dst, const void* src, size_t nbytes);
int ah_sync(ah_ctx* ctx);
/* ---- hipGraph capture: many launch-bound calls, one submission ----------------------------------------------------------------
 * No reference analogue.  Between ah_graph_begin and ah_graph_end the calls on this context are RECORDED, not run; ah_graph_launch
 * replays the sequence (same pointers, same lengths — new contents) with one submission to the compute stream.  Capturable: the
 * calls that neither wait for the device nor allocate — element-wise arithmetic, comparisons, bitmap / Kleene ops, copies and
 * memsets, the *_dev flavours (ah_sum_*_dev, ah_cmp_filter_sum_*_dev, ah_filter_primitive_dev, ah_take_primitive_dev),
 * ah_cumulative_sum without a null count — after ONE eager call of the same sequence (the scratch arenas get their size there).
 * A call with a *_host result invalidates the capture; ah_graph_end then fails and nothing was run.  The recorded launches point into
 * the context's work areas: if a later, larger call grows one of them, ah_graph_launch refuses the stale graph (AH_EINVALID) — record again. */
typedef struct ah_graph ah_graph;
int ah_graph_begin(ah_ctx* ctx);
int ah_graph_end(ah_ctx* ctx, ah_graph** out);
int ah_graph_launch(ah_ctx* ctx, ah_graph* graph);
int ah_graph_destroy(ah_graph* graph);
/* pin / unpin memory the host already owns (hipHostRegister) — e.g. a Go []byte for the duration of one cgo call; copies from
 * pageable memory are staged by the runtime and overlap nothing */
int ah_host_register(ah_ctx* ctx, void* hptr, size_t nbytes);
int ah_host_unregister(ah_ctx* ctx, void* hptr);

/* ---- overlapped host -> HBM ingest (csrc/ah_ingest.hip) ----------------------------------------------------------------------
 * Replaces, for host-resident inputs, the executor's span loop: ExecCtx.ChunkSize / NumParallel (arrow/compute/executor.go:46-64)
 * and the per-span kernel call (:658-702), over buffers from a memory.Allocator (arrow/memory/allocator.go:20-27 — here
 * ah_host_alloc_pinned).  An ah_ingest owns `depth` slots of three `chunk_bytes` device buffers, an upload stream and a download
 * stream; chunk k + 1 uploads while chunk k computes and chunk k - 1 downloads, ordered by per-slot events (not by the
 * stream-wide fences of ah_upload_async).  chunk_bytes = 0 -> 32 MiB, depth = 0 -> 3.  Host buffers must be pinned
 * (ah_host_alloc_pinned / ah_host_register) for the copies to overlap; pageable memory still gives correct results.
 * One call at a time per ah_ingest; it shares its context's compute stream. */
typedef struct ah_ingest ah_ingest;
int ah_ingest_create(ah_ctx* ctx, size_t chunk_bytes, int depth, ah_ingest** out);
int ah_ingest_destroy(ah_ingest* ing);
/* arrow/math Sum of a HOST column (float64.go:34-47 / int64.go:34-47; validity ignored): every chunk's double-double workgroup
 * partials are kept and reduced ONCE, so the result has the rounding of ah_sum_float64 over the whole column */
int ah_ingest_sum_float64(ah_ingest* ing, const double* buf_host, size_t len, double* res_host);
int ah_ingest_sum_int64(ah_ingest* ing, const int64_t* buf_host, size_t len, int64_t* res_host);
/* ah_arithmetic_binary with all three buffers on the host (base_arithmetic.cc:441-475) */
int ah_ingest_arithmetic_binary(ah_ingest* ing, int type, int8_t op, const void* l_host, const void* r_host, void* out_host, int64_t len);
/* PrimitiveFilter (vector_selection.go:449-520) on host buffers, two-phase like ah_filter_count / ah_filter_primitive: the count
 * call uploads the selection vector (n/8 bytes) and KEEPS it on the device; the host allocates the outputs; the fill call
 * streams the values through (chunks with no survivor never cross PCIe) and returns values, validity (out_valid_host nullable:
 * pass NULL when no input has nulls) and null count.  Same payload rules as ah_filter_primitive. */
int ah_ingest_filter_count(ah_ingest* ing, const uint8_t* fdata_host, const uint8_t* fvalid_host, int64_t foff, int64_t n, int null_sel,
                           int64_t* n_out_host);
int ah_ingest_filter_primitive(ah_ingest* ing, int byte_width, const void* values_host, const uint8_t* vvalid_host, int64_t voff, int64_t n,
                               int64_t n_out, void* out_values_host, uint8_t* out_valid_host, int64_t* out_null_count_host);
/* The slot protocol itself, for a host that runs other kernels over its own chunking (go/arrowhip/register.go stage()):
 *   upload(slot, which, …, first_of_chunk=1 for the chunk's first copy)   h2d stream; waits only for the slot's last release
 *   ready(slot, writes_output)                                            compute stream waits for the slot's uploads
 *                                                                          (and, if the kernel writes buffer 2, for its last download)
 *   … any ah_* kernels on ah_ingest_slot_buffer(slot, 0..2) …
 *   release(slot)                                                          marks the kernels done
 *   download(slot, which, …)                                               d2h stream; waits for the release
 *   wait()                                                                 host waits for all three streams */
int ah_ingest_depth(ah_ingest* ing);
size_t ah_ingest_chunk_bytes(ah_ingest* ing);
void* ah_ingest_slot_buffer(ah_ingest* ing, int slot, int which);
int ah_ingest_slot_upload(ah_ingest* ing, int slot, int which, size_t dst_offset, const void* hptr, size_t nbytes, int first_of_chunk);
int ah_ingest_slot_ready(ah_ingest* ing, int slot, int writes_output);
int ah_ingest_slot_release(ah_ingest* ing, int slot);
int ah_ingest_slot_download(ah_ingest* ing, int slot, int which, size_t src_offset, void* hptr, size_t nbytes);
int ah_ingest_wait(ah_ingest* ing);
/* hipEvent pair on the compute stream (what bench.py times kernels with). */
int ah_timer_start(ah_ctx* ctx);
int ah_timer_stop(ah_ctx* ctx, float* ms_host); /* synchronises */
/* numbered event slots (0..65535): bracket single kernels inside a longer timed region
 * without synchronising in between; elapsed_ms waits for slot_b only. */
int ah_event_record(ah_ctx* ctx, int slot);
int ah_event_elapsed_ms(ah_ctx* ctx, int slot_a, int slot_b, float* ms_host);

/* ---- arrow/math Sum -------------------------------------------------------
 * replaces _sum_float64_avx2 / sum_float64_go (arrow/math/float64_avx2_amd64.go:33-42,
 * arrow/math/float64.go:41-47; C truth arrow/math/_lib/float64.c:20-26) and the
 * int64/uint64 twins (arrow/math/_lib/int64.c:21-27, uint64.c).  Validity is
 * ignored exactly as the reference ignores it.  len == 0 → 0.
 * float64: every lane accumulates in double-double (TwoSum), so the result is the
 * exact sum rounded once (≤ 1 ULP), independent of launch geometry; see DESIGN.md
 * for why this — not either of the reference's two mutually inconsistent orders —
 * is the parity rule.  int64/uint64: wrapping, bit-exact. */
int ah_sum_float64(ah_ctx* ctx, const double* buf, size_t len, double* res_host);
int ah_sum_int64(ah_ctx* ctx, const int64_t* buf, size_t len, int64_t* res_host);
int ah_sum_uint64(ah_ctx* ctx, const uint64_t* buf, size_t len, uint64_t* res_host);
/* same, result left in device memory (8 bytes), no synchronisation */
int ah_sum_float64_dev(ah_ctx* ctx, const double* buf, size_t len, double* res_dev);
int ah_sum_int64_dev(ah_ctx* ctx, const int64_t* buf, size_t len, int64_t* res_dev);

/* ---- element-wise arithmetic ------------------------------------------------
 * replace _arithmetic_binary_avx2 / _arithmetic_arr_scalar_avx2 /
 * _arithmetic_scalar_arr_avx2 / _arithmetic_unary_same_types_avx2
 * (kernels/base_arithmetic_avx2_amd64.go:27-60; C truth
 * kernels/_lib/base_arithmetic.cc:465-483).  op ∈ {ADD, SUB, MUL} (the *_CHECKED
 * ids alias the unchecked op, as in the C source :54-57); every slot is computed,
 * null payloads included.  Scalars are read from host memory (one element). */
int ah_arithmetic_binary(ah_ctx* ctx, int type, int8_t op, const void* l, const void* r, void* out, int64_t len);
int ah_arithmetic_arr_scalar(ah_ctx* ctx, int type, int8_t op, const void* l, const void* r_host, void* out, int64_t len);
int ah_arithmetic_scalar_arr(ah_ctx* ctx, int type, int8_t op, const void* l_host, const void* r, void* out, int64_t len);
int ah_arithmetic_unary(ah_ctx* ctx, int type, int8_t op, const void* in, void* out, int64_t len); /* ABS, NEGATE, SIGN */
/* checked integer ADD/SUB/MUL — replaces the pure-Go ScalarBinaryNotNull path
 * (kernels/base_arithmetic.go:249-286, kernels/helpers.go:284-380): ADD/SUB write 0
 * under null slots and test only valid slots; MUL evaluates every slot
 * (mulWithOverflow :84-106).  Returns AH_EOVERFLOW ("overflow") if any tested slot
 * overflowed by the reference's own carry test; `out` is then unspecified, as the
 * reference discards it.  For AS/SA shapes the scalar operand pointer is host
 * memory and scalar_valid says whether the scalar is non-null.  Synchronises. */
int ah_arithmetic_checked(ah_ctx* ctx, int type, int8_t op, int shape,
                          const void* l, const uint8_t* lvalid, int64_t loff,
                          const void* r, const uint8_t* rvalid, int64_t roff,
                          int scalar_valid, void* out, int64_t len);

/* The exact rest of the arithmetic registry — no assembly leaf in the reference (base_arithmetic_amd64.go:67-105
 * keeps these in Go), signature derived from the Go closures.  Same argument meaning as ah_arithmetic_checked;
 * unary ops ignore r / shape.
 *   DIV, DIV_CHECKED (divide_unchecked, divide; base_arithmetic.go:154-160,287-294,386-396): integers refuse a zero
 *       divisor in a valid slot under BOTH names (AH_EINVALID "divide by zero"), truncated quotient, MinInt / −1
 *       wraps; floats: IEEE a / b, the checked name refuses b == 0.  Null slots hold 0 (ScalarBinaryNotNull).
 *   ABS_CHECKED, NEGATE_CHECKED (abs, negate; :295-340,398-411): MinInt in ANY slot, null or not, is
 *       AH_EOVERFLOW "overflow" (ScalarUnary walks the whole value buffer); unsigned abs copies; no unsigned negate.
 *   BIT_AND / OR / XOR (bit_wise_*; scalar_arithmetic.go:170-245): every slot.  BIT_NOT (:253-268): null slots hold 0.
 *   SHIFT_LEFT / RIGHT [_CHECKED] (:293-378): a count outside [0, bits − 2] (signed) / [0, bits − 1] (unsigned)
 *       returns the left operand; the checked names make it AH_EINVALID "shift amount must be >= 0 and less than
 *       precision of type".  Null slots hold 0.
 *   SQRT (every slot), SQRT_CHECKED (null slots 0; a negative valid value is AH_EINVALID "square root of negative
 *       number"; :412-426): correctly rounded IEEE sqrt, floats only.
 *   FLOOR, CEIL, TRUNC (floor, ceil, trunc; rounding.go:180-187,748-775): every slot, floats only.
 *   POWER (power_unchecked; base_arithmetic.go:226-248): integers right-to-left in uint64, narrowed (wraps), EVERY slot;
 *       POWER_CHECKED (power; :342-373): left-to-right with mulWithOverflow → AH_EOVERFLOW "overflow", valid slots only,
 *       null slots 0.  A negative integer exponent is AH_EINVALID "integers to negative integer powers are not allowed"
 *       under both.  Floats (:443-446): pow in double under both names, every slot, never fails. */
int ah_arithmetic_ext(ah_ctx* ctx, int type, int op, int shape, const void* l, const uint8_t* lvalid, int64_t loff, const void* r,
                      const uint8_t* rvalid, int64_t roff, int scalar_valid, void* out, int64_t len);

/* round / round_to_multiple on Float32 / Float64 (kernels/rounding.go:321-370 round[T].call, :562-598
 * roundToMultiple[T].call; ScalarUnaryNotNull → null slots hold 0).  mode = RoundMode (rounding.go:40-59: 0 Down, 1 Up,
 * 2 TowardsZero, 3 AwayFromZero, 4 HalfDown, 5 HalfUp, 6 HalfTowardsZero, 7 HalfAwayFromZero, 8 HalfToEven, 9 HalfToOdd).
 * multiple_host == NULL: round to ndigits, pow10 = math.Pow10(|ndigits|) computed by the caller exactly as
 * InitRoundState does (:72-91); else *multiple_host (one element of `type`, positive) is the rounding multiple and
 * ndigits / pow10 are ignored.  A non-finite result in a valid slot is AH_EOVERFLOW "overflow". */
int ah_round(ah_ctx* ctx, int type, const void* values, const uint8_t* valid, int64_t off, int64_t n, int64_t ndigits, int mode,
             const void* multiple_host, double pow10, void* out);

/* ---- comparisons → packed bitmap ----------------------------------------------
 * replaces the 12 _comparison_<op>_<shape>_avx2 symbols
 * (kernels/scalar_comparison_avx2_amd64.go; C truth
 * kernels/_lib/scalar_comparison.cc:210-256).  out_bits points at the byte holding
 * the first output bit, out_bit_offset % 8 is the bit prefix inside it (the
 * reference passes out.Offset; kernels/scalar_comparisons.go:199-218).  Only bits
 * [prefix, prefix+length) are written. */
int ah_comparison(ah_ctx* ctx, int cmpop, int shape, int type, const void* l, const void* r,
                  uint8_t* out_bits, int64_t length, int out_bit_offset);

/* ---- null-bitmap utilities -----------------------------------------------------
 * replace bitutil.BitmapAnd/Or/Xor/AndNot/Xnor (arrow/bitutil/bitmaps.go:592-637 →
 * _bitmap_aligned_*_avx2, arrow/bitutil/bitmap_ops_avx2_amd64.go:26-52, plus the Go
 * unaligned path :568-582), CountSetBits (arrow/bitutil/bitutil.go:89-130),
 * CopyBitmap/InvertBitmap (bitmaps.go:483-493), SetBitsTo (bitutil.go:158-204) and
 * the Kleene word kernels (kernels/scalar_boolean.go:29-65).  Arbitrary,
 * independent bit offsets; only bits [ooff, ooff+nbits) of the output change. */
int ah_bitmap_op(ah_ctx* ctx, int op, const uint8_t* l, int64_t loff, const uint8_t* r, int64_t roff,
                 uint8_t* out, int64_t ooff, int64_t nbits);
int ah_count_set_bits(ah_ctx* ctx, const uint8_t* bits, int64_t off, int64_t nbits, int64_t* out_host);
int ah_copy_bitmap(ah_ctx* ctx, const uint8_t* src, int64_t soff, int64_t nbits, uint8_t* dst, int64_t doff, int invert);
int ah_set_bits_to(ah_ctx* ctx, uint8_t* bits, int64_t off, int64_t nbits, int value);
int ah_kleene(ah_ctx* ctx, int op, const uint8_t* lvalid, const uint8_t* ldata, int64_t loff,
              const uint8_t* rvalid, const uint8_t* rdata, int64_t roff,
              uint8_t* ovalid, uint8_t* odata, int64_t ooff, int64_t nbits);

/* ---- selection -------------------------------------------------------------------
 * Filter: data-dependent output size and Go owns allocation, so the protocol is the
 * reference's own two steps: ah_filter_count == getFilterOutputSize
 * (kernels/vector_selection.go:57-81), caller allocates n_out*byte_width bytes (+
 * ceil(n_out/8) validity bytes iff either input has nulls — preallocateData :83-93,
 * PrimitiveFilter :486-488), then ah_filter_primitive == primitiveFilterImpl
 * (:267-395).  n_out is the value ah_filter_count returned (the library zero-fills
 * the ceil(n_out/8) bytes of out_valid itself and cross-checks n_out when it
 * synchronises; pass -1 only when out_valid is NULL).  Payload rules in DESIGN.md.
 * byte_width ∈ {1,2,4,8}.  out_null_count_host may be NULL (then no sync).
 * The count call leaves the per-tile survivor prefixes of ITS mask in the context, and an ah_filter_primitive with the same
 * (fdata, fvalid, foff, n, null_sel) uses them instead of counting again — as long as nothing that can change device memory was
 * called on this context in between: allocation, synchronisation, timers, and a memset / upload / copy whose destination
 * does not overlap the mask's bytes keep them; every other entry point drops them.  Writing the mask from OUTSIDE this
 * context between the two calls (another context, another library on another stream) breaks the two-phase contract itself —
 * n_out would no longer be the mask's count either. */
int ah_filter_count(ah_ctx* ctx, const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n,
                    int null_sel, int64_t* n_out_host);
int ah_filter_primitive(ah_ctx* ctx, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                        const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                        int64_t n_out, void* out_values, uint8_t* out_valid, int64_t* out_null_count_host);
/* The same compaction without the count call and without a host round trip, for pipelines that
 * keep going on the stream: out_values holds n·byte_width bytes and out_valid (iff either input has
 * nulls) ceil(n/8) — the worst case — and status_dev[0] = rows selected, status_dev[1] = output
 * null count are left in DEVICE memory (16 bytes, readable after ah_sync or by the next kernel). */
int ah_filter_primitive_dev(ah_ctx* ctx, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                            const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                            void* out_values, uint8_t* out_valid, int64_t* status_dev /* [2] */);
/* … and the same in ONE synchronous call: outputs sized for n rows as above, the rows selected and the output null count returned to
 * the host through the polled mailbox — for callers that can afford the worst-case allocation (vector_selection.go:459-475 sizes
 * exactly, at the price of a count call and a second launch after the host has heard back). */
int ah_filter_primitive_once(ah_ctx* ctx, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                             const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n, int null_sel,
                             void* out_values, uint8_t* out_valid, int64_t* n_out_host, int64_t* out_null_count_host);
/* GetTakeIndices (kernels/vector_selection.go:102-236), uint32 flavour: the mask as
 * an index vector so FilterRecordBatch can gather N columns with one scan. */
int ah_filter_to_indices(ah_ctx* ctx, const uint8_t* fdata, const uint8_t* fvalid, int64_t foff, int64_t n,
                         int null_sel, int64_t n_out, uint32_t* out_idx, uint8_t* out_valid,
                         int64_t* out_null_count_host);
/* Take == PrimitiveTake (kernels/vector_selection.go:1162-1192 → primitiveTakeImpl
 * :878-988): out[i] = values[idx[i]]; null index or null value → payload 0, validity
 * 0.  byte_width ∈ {1,2,4,8} and, for FSBImpl's slots (:1997-2031: Decimal128/256, 16- and 32-byte binaries), {16,32}; any other
 * width up to 4096 is copied byte by byte (a plain kernel: the reference's own test column is binary(3)).
 * idx_byte_width ∈ {1,2,4,8}; idx_signed selects the bounds rule of
 * checkIndexBounds (kernels/helpers.go:929-957; only valid index slots are checked).
 * An out-of-range valid index returns AH_EINDEX with the first offender (in index
 * order) in *bad_index_host and "<v> out of bounds" as the message; the check is
 * fused into the gather and also runs when bounds_check == 0 (where the reference
 * would panic).  out_valid may be NULL when neither input has nulls.  Synchronises. */
int ah_take_primitive(ah_ctx* ctx, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                      int64_t nvalues, int idx_byte_width, int idx_signed, const void* idx,
                      const uint8_t* ivalid, int64_t ioff, int64_t nidx, int bounds_check,
                      void* out_values, uint8_t* out_valid, int64_t* out_null_count_host,
                      int64_t* bad_index_host);
/* Take without the host round trip: status_dev[0] = POSITION (not value) of the first valid index
 * that is out of range, UINT64_MAX when there is none — the output is then unspecified at the
 * offending rows (payload 0, validity 0) and the caller raises checkIndexBounds' error itself;
 * status_dev[1] = output null count (0 without out_valid).  Device memory, 16 bytes.  The one wait
 * left is the path choice for columns ≥ 64 MiB gathered by ≥ 2^20 indices (a 64 × 255-neighbour sample). */
int ah_take_primitive_dev(ah_ctx* ctx, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                          int64_t nvalues, int idx_byte_width, int idx_signed, const void* idx,
                          const uint8_t* ivalid, int64_t ioff, int64_t nidx,
                          void* out_values, uint8_t* out_valid, uint64_t* status_dev /* [2] */);

/* ---- hashing ------------------------------------------------------------------------
 * unique / dictionary_encode over 8-byte keys == doAppendNumeric[uint64] over
 * hashing.Table[uint64] (kernels/vector_hash.go:359-385,
 * internal/hashing/xxh3_memo_table_types.go:189-294; Int64 and Float64 columns both
 * hash their raw bit patterns, vector_hash.go:604-607).  Dense ids are in
 * FIRST-SEEN ORDER, bit-identical to the sequential memo table.  encode_nulls = 1
 * (unique, NullEncodingEncode): null owns the id at which it was first seen;
 * = 0 (NullEncodingMask): null → id 0 + cleared validity bit.  out_dict must hold
 * min(n, distinct)+1 keys — call with out_dict = NULL first to learn the size if
 * unknown, or size it n+1.  out_ids / out_ids_valid may be NULL.  Synchronises.  ≤ 2^30 rows per call
 * (slot numbers travel through the int32 id column; every reference config is ≤ 2^30 rows). */
int ah_hash_u64_encode(ah_ctx* ctx, const uint64_t* keys, const uint8_t* valid, int64_t off, int64_t n,
                       int encode_nulls, int32_t* out_ids, uint8_t* out_ids_valid, uint64_t* out_dict,
                       int64_t* out_ndict_host, int32_t* out_null_id_host);
/* The same for Binary / String (offset_width 4) and LargeBinary / LargeString (8) values ==
 * doAppendBinary (kernels/vector_hash.go:288-325) over hashing.BinaryMemoTable
 * (internal/hashing/xxh3_memo_table.go:248-341: lookup by hash + bytes.Equal, memo index = position
 * in the builder = order of first occurrence; null via GetOrInsertNull :333-341).  `offsets`: the
 * values' offsets buffer (elements off + i, off + i + 1 delimit row i; validity bit off + i).
 * The memo table's hash (hash_funcs.go:86-124, xxh3 for > 16 bytes) only places entries inside the
 * reference's table and cannot influence ids or dictionary order, so the device uses its own 64-bit
 * hash and compares bytes on tag match.  The dictionary is not copied here: out_first_rows[id]
 * (int64, relative to off, n + 1 entries of room) is the row that first held dictionary entry id —
 * dictionary = ah_take_binary_offsets / _data with those rows as indices (the null entry, if any,
 * points at a null row: zero length, as BinaryBuilder.AppendNull leaves it).  ≤ 2^30 rows per call. */
int ah_hash_binary_encode(ah_ctx* ctx, int offset_width, const void* offsets, const uint8_t* data, const uint8_t* valid, int64_t off,
                          int64_t n, int encode_nulls, int32_t* out_ids, uint8_t* out_ids_valid, int64_t* out_first_rows,
                          int64_t* out_ndict_host, int32_t* out_null_id_host);
/* FixedSizeBinary / Decimal128 / Decimal256 keys of unique / dictionary_encode (kernels/vector_hash.go:608-609, 698: the same
 * BinaryMemoTable, every value byte_width bytes; row i of the call = data + (off + i)·byte_width, validity bit off + i).  ids, index
 * validity, first rows, null id as ah_hash_binary_encode; out_dict (nullable, ndict·byte_width bytes) receives the dictionary —
 * the value at each entry's first row, zeros for the null entry.  Synchronises. */
int ah_hash_fixed_encode(ah_ctx* ctx, int byte_width, const uint8_t* data, const uint8_t* valid, int64_t off, int64_t n, int encode_nulls,
                         int32_t* out_ids, uint8_t* out_ids_valid, int64_t* out_first_rows, uint8_t* out_dict, int64_t* out_ndict_host,
                         int32_t* out_null_id_host);
/* group-by sum (NEW — arrow-go has no hash aggregate; definition in DESIGN.md): groups
 * = dictionary_encode(keys, encode_nulls=1) ids; out_sums[g] = Σ valid vals of group
 * g, out_counts[g] = number of valid vals.  i64 sums wrap and are exact; f64 sums are
 * accumulated in 128-bit fixed point with integer atomics (csrc/ah_hashing.h): the same bytes
 * run after run; a column spanning <= 42 binades gives every group its CORRECTLY ROUNDED exact
 * sum, a wider one (an outlier such as 1e300 next to ordinary values) one scale per group and
 * |err| <= ulp/2 + n_g·2^-94·max_g|x| — inside the sequential definition's n_g·ε·Σ_g|x|.
 * out_first_rows (nullable) receives each group's first
 * row index — what a cross-GPU merge needs to restore the global first-seen order.
 * Outputs sized like out_dict above. */
int ah_hash_sum_f64(ah_ctx* ctx, const uint64_t* keys, const uint8_t* kvalid, int64_t koff,
                    const double* vals, const uint8_t* vvalid, int64_t voff, int64_t n,
                    uint64_t* out_keys, double* out_sums, int64_t* out_counts, int64_t* out_first_rows,
                    int64_t* out_ngroups_host, int32_t* out_null_group_host);
int ah_hash_sum_i64(ah_ctx* ctx, const uint64_t* keys, const uint8_t* kvalid, int64_t koff,
                    const int64_t* vals, const uint8_t* vvalid, int64_t voff, int64_t n,
                    uint64_t* out_keys, int64_t* out_sums, int64_t* out_counts, int64_t* out_first_rows,
                    int64_t* out_ngroups_host, int32_t* out_null_group_host);

/* ---- fused Compare(op scalar) → Filter(DropNulls) → Sum ------------------------------
 * NEW entry point (no reference analogue) computing in ONE pass what the reference
 * computes with "greater" → "array_filter" → math.Sum: Σ x[i] over valid slots with
 * x[i] OP threshold, and the survivor count.  8 B/row of HBM traffic instead of
 * 16.25 + 16·s.  *_dev variants leave {sum, count} (16 bytes) in device memory for a
 * following RCCL all-reduce. */
int ah_cmp_filter_sum_i64(ah_ctx* ctx, int cmpop, const int64_t* x, const uint8_t* valid, int64_t off, int64_t n,
                          int64_t threshold, int64_t* out_sum_host, int64_t* out_count_host);
int ah_cmp_filter_sum_f64(ah_ctx* ctx, int cmpop, const double* x, const uint8_t* valid, int64_t off, int64_t n,
                          double threshold, double* out_sum_host, int64_t* out_count_host);
int ah_cmp_filter_sum_i64_dev(ah_ctx* ctx, int cmpop, const int64_t* x, const uint8_t* valid, int64_t off,
                              int64_t n, int64_t threshold, int64_t* out_sum_count_dev /* [2] */);
int ah_cmp_filter_sum_f64_dev(ah_ctx* ctx, int cmpop, const double* x, const uint8_t* valid, int64_t off,
                              int64_t n, double threshold, double* out_sum_dev, int64_t* out_count_dev);

/* ---- cumulative_sum (row §8(f)-2) -----------------------------------------------------------
 * replaces cumulativeSumExec → cumulativeSum{NoNulls,WithNulls}[Checked]
 * (kernels/vector_cumulative.go:228-360) behind "cumulative_sum" / "cumulative_sum_checked"
 * (compute/vector_cumulative.go:76-96): out[i] = start + Σ_{j≤i, valid} in[j]; a null row gives a
 * null output (payload 0) and, unless skip_nulls, turns every later row null as well; checked →
 * AH_EOVERFLOW ("overflow") as soon as a running sum leaves the type's range.  One pass over HBM
 * (decoupled look-back scan).  start_host: one element of `type` or NULL (= 0).  out_valid must be
 * given iff `valid` is (ceil(n/8) bytes, offset 0).  Integers bit-exact; floats are summed in a
 * parallel order (tolerance in DESIGN.md).  Synchronises if checked or out_null_count_host. */
int ah_cumulative_sum(ah_ctx* ctx, int type, const void* values, const uint8_t* valid, int64_t off, int64_t n,
                      const void* start_host, int skip_nulls, int checked, void* out_values, uint8_t* out_valid,
                      int64_t* out_null_count_host);

/* ---- device interchange (row §8(f)-3) ---------------------------------------------------------------
 * ArrowDeviceArray.sync_event (arrow/cdata/abi.h:104-128; for ARROW_DEVICE_ROCM a hipEvent_t*): make
 * the context's compute stream wait for the producer's event before touching imported buffers.
 * NULL = already synchronised.  No host synchronisation. */
int ah_wait_event(ah_ctx* ctx, void* hip_event_ptr);
int ah_device_id(ah_ctx* ctx);

/* ---- min_max (row §8(f)-2) -----------------------------------------------------------------------
 * replaces utils.GetMinMax{Int8…Uint64} (internal/utils/min_max.go:161-215; AVX2 leaves
 * _int64_max_min_avx2(values, len, &min, &max) …, internal/utils/min_max_avx2_amd64.go; C:
 * internal/utils/_lib/min_max.c:23-126).  Integer types; validity is not consulted (as in the
 * reference); an empty input returns (MaxOf, MinOf).  One pass, w bytes per row; synchronises. */
int ah_min_max(ah_ctx* ctx, int type, const void* values, int64_t n, void* out_min_host, void* out_max_host);

/* ---- multi-GPU group merge helper: owner of a key = (hashInt(key) >> 40) mod nparts,
 * hashInt = internal/hashing/hash_funcs.go:60-67.  No reference analogue (the reference is single-process). */
int ah_hash_partition_u64(ah_ctx* ctx, const uint64_t* keys, int64_t n, int nparts, int32_t* out_part);

/* ---- multi-GPU exchange — RCCL over xGMI on the context's compute stream -------------------------------
 * No reference analogue (arrow-go is single-process).  One process per GPU.  Rank 0 calls ah_comm_unique_id and the
 * host distributes the 128 bytes (file, socket, launcher); every rank then calls ah_comm_init on its own ah_ctx.
 * Collectives are enqueued behind the kernels already on the compute stream and return without waiting (ah_sync to
 * wait); buffers are device pointers, counts / offsets host arrays of `world` entries.  The path needs exactly these:
 *   C4  ah_comm_allreduce_sum(AH_INT64, partial, partial, 2) after ah_cmp_filter_sum_i64_dev — 16 bytes;
 *       float64: ah_comm_allgather of the 8-byte partials, added in rank order by the caller (bit-reproducible)
 *   C5  owner merge of group tuples: ah_comm_alltoallv (direct send / recv pairs — all xGMI links at once) of
 *       O(groups) bytes, then ah_comm_allgather of the owners' results
 * RCCL is bound with dlopen at first use (the copy already in the process, e.g. torch's, else the system's;
 * ARROWHIP_RCCL overrides); a machine without it gets AH_EHIP from these calls and loses nothing else. */
typedef struct ah_comm ah_comm;
int ah_comm_unique_id(void* id_host128);
int ah_comm_init(ah_ctx* ctx, int rank, int world, const void* unique_id_host128, ah_comm** out);
int ah_comm_destroy(ah_comm* comm);
int ah_comm_rank(ah_comm* comm);
int ah_comm_world(ah_comm* comm);
/* type: AH_INT32 / UINT32 / INT64 / UINT64 / FLOAT32 / FLOAT64; send may equal recv (in place) */
int ah_comm_allreduce_sum(ah_comm* comm, int type, const void* send, void* recv, int64_t count);
/* recv holds world × nbytes_per_rank bytes, rank r's block at r × nbytes_per_rank */
int ah_comm_allgather(ah_comm* comm, const void* send, void* recv, int64_t nbytes_per_rank);
/* block for rank r: send + send_offs_host[r], send_bytes_host[r] bytes; arrives at recv + recv_offs_host[r] of rank r;
 * recv_bytes_host[r] must be what rank r sends here (exchange the sizes first: an allgather of 8·world bytes) */
int ah_comm_alltoallv(ah_comm* comm, const void* send, const int64_t* send_bytes_host, const int64_t* send_offs_host, void* recv,
                      const int64_t* recv_bytes_host, const int64_t* recv_offs_host);
/* A communicator whose bytes are carried by the HOST's own transport (its launcher's sockets, MPI, a gloo group in the tests)
 * instead of RCCL: the two callbacks work on host memory, return 0 on success, and are called by every rank in the same order.
 * Device blocks are staged through pinned memory.  For hosts without RCCL between their processes — and for several ranks on
 * ONE GPU, which RCCL refuses.  All ah_comm_* entry points work on either flavour. */
typedef struct ah_transport {
  void* user;
  /* every rank contributes nbytes_per_rank bytes; recv_host gets world x nbytes_per_rank, rank order */
  int (*allgather)(void* user, const void* send_host, void* recv_host, int64_t nbytes_per_rank);
  /* block for rank r: send_host + send_offs[r], send_bytes[r] bytes -> recv_host + recv_offs[r] on rank r (the own block included) */
  int (*alltoallv)(void* user, const void* send_host, const int64_t* send_bytes, const int64_t* send_offs, void* recv_host,
                   const int64_t* recv_bytes, const int64_t* recv_offs);
} ah_transport;
int ah_comm_init_transport(ah_ctx* ctx, int rank, int world, const ah_transport* transport, ah_comm** out);
/* ---- configs C4 / C5 as single calls — what a host runs per record-batch shard, nothing in between -------------
 * C4: the fused Compare(op, scalar) -> Filter(DropNulls) -> Sum over THIS rank's shard (ah_cmp_filter_sum_*) and the one exchange
 * step of the path: Int64 — an all-reduce of the 16-byte {sum, count} (wrapping sums are exact in any order); Float64 — an
 * all-gather of the partials, added in RANK order on every rank (the same bytes on every rank and in every run of a given world
 * size; an all-reduce's order is the ring's).  Every rank receives the global result.  Synchronises. */
int ah_comm_cmp_filter_sum_i64(ah_comm* comm, int cmpop, const int64_t* x, const uint8_t* valid, int64_t off, int64_t n_local,
                               int64_t threshold, int64_t* out_sum_host, int64_t* out_count_host);
int ah_comm_cmp_filter_sum_f64(ah_comm* comm, int cmpop, const double* x, const uint8_t* valid, int64_t off, int64_t n_local,
                               double threshold, double* out_sum_host, int64_t* out_count_host);
/* C5: merge of the ranks' LOCAL aggregates (the outputs of ah_hash_sum_* over each rank's row shard; first_rows local, row_offset =
 * the shard's first global row) by key-hash owner: groups bucketed by (hashInt(key) >> 40) mod world on the device -> ragged
 * all-to-all of {key, sum, count, global first row} tuples (O(groups) bytes, never O(rows)) -> the owner re-aggregates (Float64:
 * the reproducible sums of ah_hash_sum_f64) -> ragged all-gather -> ordered by global first row: the groups, in the order, that
 * unique / dictionary_encode over the undivided column gives (kernels/vector_hash.go:359-385, 721-741).  Every rank receives all
 * groups.  Device buffers; out_* hold `capacity` groups; *out_ngroups_host = the global group count — AH_EINVALID with that
 * count set (and nothing written) when capacity is too small.  Synchronises.
 * The NULL group: null_group_local = what ah_hash_sum_* reported in *out_null_group_host for this rank's shard (−1: none).  A null
 * key has no hash owner — its group (key slot 0, like the local aggregate's) is merged on every rank from the ranks' null tuples
 * in rank order, takes its place in the first-seen order, and *out_null_group_host (nullable) = its position, −1 if no rank had
 * one.  A column holding both null keys and the key 0 therefore yields two groups, as over the undivided column. */
int ah_comm_merge_groups(ah_comm* comm, int is_f64, const uint64_t* keys, const void* sums, const int64_t* counts, const int64_t* first_rows,
                         int64_t ngroups_local, int32_t null_group_local, int64_t row_offset, int64_t capacity, uint64_t* out_keys,
                         void* out_sums, int64_t* out_counts, int64_t* out_first_rows, int64_t* out_ngroups_host,
                         int32_t* out_null_group_host);

/* ---- numeric cast (row §8(f)-2) ----------------------------------------------------------------
 * replaces castNumberToNumberUnsafe → castNumericUnsafe (kernels/cast_numeric.go:28-131; AVX2 leaf
 * cast_type_numeric_avx2(itype, otype, in, out, len), kernels/_lib/cast_numeric.cc:62) together with
 * the safe-cast checks the reference runs as separate passes: intsCanFit / intsInRange
 * (helpers.go:496-652), checkIntToFloatTrunc (numeric_cast.go:698-729), checkFloatTrunc (:613-660) —
 * behind compute's "cast" with CastOptions{ToType, AllowIntOverflow, AllowFloatTruncate}
 * (kernels/cast.go:27-35).  Every slot is converted (valid or not); only VALID slots can fail a check
 * (valid = NULL: all valid).  Failure → AH_EINVALID with the reference's text for the first offending
 * row: "integer value %d not in range: %d to %d" / "float value %f was truncated converting to %s".
 * Synchronises only when a check is active.  in_type == out_type is a device copy. */
int ah_cast_numeric(ah_ctx* ctx, int in_type, int out_type, const void* values, const uint8_t* valid, int64_t off, int64_t n,
                    int allow_int_overflow, int allow_float_truncate, void* out_values);
/* ---- temporal unit change (the casts either side of the temporal kernels) ----------------------------
 * replaces ShiftTime[InT, OutT](ctx, op, factor, input, output) (kernels/cast_temporal.go:35-104), the leaf of the
 * timestamp / duration / time32↔time64 / date32↔date64 casts (cast_temporal.go:240-420) and of the implicit unit
 * casts DispatchBest inserts between temporal operands (arithmetic.go:130-131).  in_bits / out_bits: 32 or 64
 * (the storage integers).  op: AH_SHIFT_MULTIPLY out = OutT(v)·OutT(factor) (wrapping), AH_SHIFT_DIVIDE
 * out = OutT(v / InT(factor)) (truncating); factor == 1 only converts the width.  Every slot is converted.
 * check != 0 (CastOptions.AllowTimeOverflow / AllowTimeTruncate false): a VALID value (valid = NULL: all) outside
 * [MinInt64/factor, MaxInt64/factor] (multiply) or not a multiple of factor (divide) → AH_EINVALID, *bad_value =
 * that value of the FIRST such row ("would result in out of bounds timestamp: %d" / "would lose data: %d");
 * the type names of the reference's message are the host's to add.  Synchronises only when checking. */
#define AH_SHIFT_MULTIPLY 0
#define AH_SHIFT_DIVIDE 1
int ah_shift_time(ah_ctx* ctx, int in_bits, int out_bits, int op, int64_t factor, int check, const void* values, const uint8_t* valid,
                  int64_t off, int64_t n, void* out_values, int64_t* bad_value);
/* boolToNum (numeric_cast.go:555-569): out[i] = bit(off + i) ? 1 : 0.  (numeric → bool is
 * isNonZero, boolean_cast.go:30-36 = ah_comparison(AH_CMP_NE, AH_SHAPE_AS, type, values, &zero).) */
int ah_cast_bool_to_numeric(ah_ctx* ctx, int out_type, const uint8_t* bits, int64_t off, int64_t n, void* out_values);

/* ---- is_in (row §8(f)-2) -----------------------------------------------------------------------
 * replaces SetLookupState.Init + isInKernelExec (kernels/scalar_set_lookup.go:192-244, 374-413)
 * behind compute's "is_in" with SetOptions{ValueSet, NullBehavior} (compute/scalar_set_lookup.go:
 * 62-65, 175-200).  Keys are the raw bits of a 1/2/4/8-byte value.  Writes the result's data AND
 * validity bits [out_bit_offset, out_bit_offset + n) (NullComputedPrealloc) and preserves every other
 * bit.  No host synchronisation. */
#define AH_NULL_MATCH 0        /* NullMatchingBehavior, kernels/scalar_set_lookup.go:31-38 */
#define AH_NULL_SKIP 1
#define AH_NULL_EMIT_NULL 2
#define AH_NULL_INCONCLUSIVE 3
int ah_is_in(ah_ctx* ctx, int byte_width, const void* values, const uint8_t* valid, int64_t off, int64_t n,
             const void* set_values, const uint8_t* set_valid, int64_t set_off, int64_t set_n, int null_behavior,
             uint8_t* out_data, uint8_t* out_valid, int64_t out_bit_offset);

/* ---- sort_indices (row §8(f)-2) ----------------------------------------------------------------
 * replaces kernels.SortIndices for one key over one array (kernels/vector_sort.go:388-481 →
 * arraySortOneColumnRange, vector_sort_internal.go:252-273) behind compute's "sort_indices"
 * (compute/vector_sort.go:42-52): a STABLE permutation of [0, n) as uint64 row numbers; nulls at the
 * end (default) or the start, NaNs next to them, the rest ascending or descending with ties in
 * input order.  Stable LSD radix sort; synchronises (two small read-backs). */
int ah_sort_indices(ah_ctx* ctx, int type, const void* values, const uint8_t* valid, int64_t off, int64_t n,
                    int descending, int nulls_at_start, uint64_t* out_indices);
/* several sort keys over equal-length columns (record batch / table: radixRecordBatchSortRange,
 * kernels/vector_sort_internal.go, and multiColumnComparator, vector_sort.go:103-120): lexicographic
 * by key 0, then key 1, …, each with its own order and null placement; stable.  Arrays of nkeys entries. */
int ah_sort_indices_multi(ah_ctx* ctx, int nkeys, const int* types, const void* const* values, const uint8_t* const* valids,
                          const int64_t* offs, int64_t n, const int* descending, const int* nulls_at_start, uint64_t* out_indices);

/* Boolean VALUES (booleanTakeImpl, kernels/vector_selection.go:990-1074): data and validity are bitmaps
 * indexed at voff + idx; a null output keeps data bit 0.  Output bitmaps start at bit 0.  Filter of a
 * boolean column = ah_filter_to_indices + this (the reference's own boolFilterWriter never advances its
 * output position, :433-436 — SURVEY.md quirk 8 — and is not replicated). */
int ah_take_boolean(ah_ctx* ctx, const uint8_t* data, const uint8_t* vvalid, int64_t voff, int64_t nvalues, int idx_byte_width,
                    int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, int bounds_check,
                    uint8_t* out_data, uint8_t* out_valid, int64_t* out_null_count_host, int64_t* bad_index_host);

/* ---- var-length Take / Filter (row §8(f)-4) ------------------------------------------------------
 * replaces VarBinaryImpl (kernels/vector_selection.go:1925-1992) under takeExec / filterExec
 * (:1460-1598, 1821-1923) for Binary / String (offset_width 4) and LargeBinary / LargeString (8).
 * `offsets`: the values' offsets buffer (element voff + i and voff + i + 1 delimit value i; vvalid
 * bit voff + i).  Two calls with the caller's allocation in between:
 *   _offsets: out_offsets[0..nidx] (offset_width each, starting at 0), out_valid (nullable), the null
 *             count, the number of data bytes, AH_EINDEX "<v> out of bounds" for the first offending
 *             valid index, AH_EINVALID "binary output offset overflow" (:1245) — synchronises;
 *   _data:    the bytes, into a buffer of that many bytes.
 * A null output (null value or null index) has zero length.  Filter of a binary column =
 * ah_filter_count → ah_filter_to_indices → these two (GetTakeIndices, as compute/selection.go:687 does
 * for record batches); uint32 indices, so < 2^32 rows. */
int ah_take_binary_offsets(ah_ctx* ctx, int offset_width, const void* offsets, const uint8_t* vvalid, int64_t voff, int64_t nvalues,
                           int idx_byte_width, int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx,
                           int bounds_check, void* out_offsets, uint8_t* out_valid, int64_t* out_null_count_host,
                           int64_t* out_total_bytes_host, int64_t* bad_index_host);
int ah_take_binary_data(ah_ctx* ctx, int offset_width, const void* offsets, const uint8_t* data, int64_t voff, int idx_byte_width,
                        const void* idx, int64_t nidx, const void* out_offsets, uint8_t* out_data);

/* ---- fused scalar-expression evaluation (row §8(f)-1: the expression executor) ---------
 * What compute.Expression trees — NewCall / NewFieldRef / NewLiteral, arrow/compute/
 * expression.go:596-620 — evaluate to through executeScalarBatch (arrow/compute/exprs/
 * exec.go:542-700: one scalar kernel per call node, every intermediate materialised),
 * computed in ONE kernel generated for the tree and JIT-compiled with hiprtc.  The tree is a
 * postfix program; results are bit-identical to executing the calls one by one through the
 * entry points above (same wraparound / IEEE / checked-overflow / null-intersection rules;
 * operand types of a call must match — the caller inserts AH_X_CAST nodes where the reference's
 * DispatchBest would cast — else AH_ENOTIMPL). */
typedef struct { int32_t op; int32_t arg; } ah_expr_node;
typedef struct ah_expr ah_expr;
#define AH_X_FIELD 1        /* push input column `arg` */
#define AH_X_LITERAL 2      /* push literal `arg` (value given at execute time) */
#define AH_X_ADD 10         /* add_unchecked / subtract_unchecked / multiply_unchecked */
#define AH_X_SUB 11
#define AH_X_MUL 12
#define AH_X_ADD_CHECKED 13 /* add / subtract / multiply (checked on integers) */
#define AH_X_SUB_CHECKED 14
#define AH_X_MUL_CHECKED 15
#define AH_X_NEGATE 20      /* negate_unchecked, abs_unchecked, sign */
#define AH_X_ABS 21
#define AH_X_SIGN 22
#define AH_X_EQ 30          /* equal, not_equal, greater, greater_equal, less, less_equal */
#define AH_X_NE 31
#define AH_X_GT 32
#define AH_X_GE 33
#define AH_X_LT 34
#define AH_X_LE 35
#define AH_X_AND 40         /* and, or, xor, and_not, invert (plain, non-Kleene) */
#define AH_X_OR 41
#define AH_X_XOR 42
#define AH_X_AND_NOT 43
#define AH_X_INVERT 44
#define AH_X_CAST 50        /* value-preserving numeric conversion of the top of the stack to type `arg`: integer → wider integer
                             * (unsigned → signed only when wider), ≤ 32-bit integer → double, ≤ 16-bit integer → float, float →
                             * double — the implicit casts DispatchBest inserts (commonNumeric, compute/internal/kernels/
                             * helpers.go) that can never fail a safe-cast check; any other pair is AH_ENOTIMPL */
/* column / literal types are arrow.Type ids (1 = BOOL: bitmap column).  Compiled programs are
 * cached per context by (program, types); the returned handle is owned by the context. */
int ah_expr_compile(ah_ctx* ctx, const ah_expr_node* nodes, int n_nodes, const int* col_types, int n_cols,
                    const int* lit_types, int n_lits, ah_expr** out, int* out_type_host);
/* col_values[i]: element 0 of column i (offset applied) — for BOOL columns the bitmap base;
 * col_valid[i] (nullable) and BOOL data are addressed with bit offset col_offsets[i].
 * lit_values_host: n_lits × 8 bytes (little-endian payload), lit_valid_host: n_lits flags.
 * out_values: len elements (BOOL: ceil(len/8) bytes, 8-byte aligned); out_valid (nullable:
 * pass NULL when no input can be null) receives the AND of all referenced validities.
 * Returns AH_EOVERFLOW ("overflow") like ah_arithmetic_checked when a checked node overflows
 * in a valid slot (synchronises only if the program has checked integer nodes). */
int ah_expr_execute(ah_ctx* ctx, ah_expr* expr, const void* const* col_values, const uint8_t* const* col_valid,
                    const int64_t* col_offsets, const void* lit_values_host, const int* lit_valid_host, int64_t len,
                    void* out_values, uint8_t* out_valid);
const char* ah_expr_source(ah_expr* expr); /* generated HIP source, for inspection */
/* stateless: generate (and optionally hiprtc-compile for gfx950) without a GPU or a context;
 * src_buf/err_buf may be NULL.  Used by the CPU test-suite. */
int ah_expr_codegen(const ah_expr_node* nodes, int n_nodes, const int* col_types, int n_cols, const int* lit_types, int n_lits,
                    int do_compile, char* src_buf, size_t src_cap, char* err_buf, size_t err_cap, int* out_type_host);

#ifdef __cplusplus
}
#endif
#endif /* ARROWHIP_H */
