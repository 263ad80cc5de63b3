// Package arrowhip binds libarrowhip.so (include/arrowhip.h) into arrow-go.
//
// UNVERIFIED IN THIS REPOSITORY'S BUILD ENVIRONMENT: the image has no Go toolchain, so this
// package has never been compiled.  It is the reference-side binding a maintainer would add;
// the same C ABI is exercised end to end by the C++ mirror in arrow_go_amd/host and the pytest
// suite.  Build with:  go build -tags hip ./go/arrowhip
//
//go:build hip

package arrowhip

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../arrow_go_amd -larrowhip -Wl,-rpath,${SRCDIR}/../../arrow_go_amd
#include <stdlib.h>
#include "arrowhip.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"

	"github.com/apache/arrow-go/v18/arrow"
	"github.com/apache/arrow-go/v18/arrow/compute"
)

// Context owns one GPU: ah_ctx (device id, compute stream, copy stream, scratch arena).
// Every C entry point selects the device itself, so a Context may be used from whatever OS
// thread the executor goroutine lands on (arrow/compute/exec.go:165); it serves one call at a
// time — pool Contexts for concurrency, like scalarExecPool does (executor.go:867-873).
type Context struct {
	c   *C.ah_ctx
	ing *Ingest // lazily created by register.go's ExecFns for host-resident operands (ingest.go)
}

func NewContext(device int) (*Context, error) {
	var c *C.ah_ctx
	if st := C.ah_ctx_create(C.int(device), &c); st != C.AH_OK {
		return nil, fmt.Errorf("arrowhip: ah_ctx_create(%d) failed (status %d): no GPU, no fallback", device, int(st))
	}
	ctx := &Context{c: c}
	runtime.SetFinalizer(ctx, func(x *Context) { x.Close() })
	return ctx, nil
}

func (x *Context) Close() {
	if x.ing != nil {
		x.ing.Close()
		x.ing = nil
	}
	if x.c != nil {
		C.ah_ctx_destroy(x.c)
		x.c = nil
	}
}

// err maps the C status classes onto arrow-go's sentinel errors so callers can errors.Is().
func (x *Context) err(st C.int) error {
	if st == C.AH_OK {
		return nil
	}
	msg := C.GoString(C.ah_last_error(x.c))
	switch st {
	case C.AH_EINVALID, C.AH_EOVERFLOW:
		return fmt.Errorf("%w: %s", arrow.ErrInvalid, msg)
	case C.AH_EINDEX:
		return fmt.Errorf("%w: %s", arrow.ErrIndex, msg)
	case C.AH_ENOTIMPL:
		return fmt.Errorf("%w: %s", arrow.ErrNotImplemented, msg)
	}
	return fmt.Errorf("arrowhip: HIP failure: %s", msg)
}

// DeviceBuffer is a device allocation (ah_buf_alloc); Ptr is a DEVICE pointer.
type DeviceBuffer struct {
	ctx  *Context
	Ptr  unsafe.Pointer
	Size int
}

func (x *Context) Alloc(nbytes int) (*DeviceBuffer, error) {
	var p unsafe.Pointer
	if err := x.err(C.ah_buf_alloc(x.c, C.size_t(nbytes), &p)); err != nil {
		return nil, err
	}
	return &DeviceBuffer{ctx: x, Ptr: p, Size: nbytes}, nil
}

func (b *DeviceBuffer) Free() {
	if b.Ptr != nil {
		C.ah_buf_free(b.ctx.c, b.Ptr)
		b.Ptr = nil
	}
}

// Upload copies a Go slice to the device.  The C side never retains the host pointer past
// the call when followed by Sync (cgo pointer rule); pass pinned memory (PinnedAllocator) to
// get true async DMA.
func (b *DeviceBuffer) Upload(src []byte) error {
	if len(src) == 0 {
		return nil
	}
	if err := b.ctx.err(C.ah_upload_async(b.ctx.c, b.Ptr, unsafe.Pointer(&src[0]), C.size_t(len(src)))); err != nil {
		return err
	}
	return b.ctx.Sync()
}

func (b *DeviceBuffer) Download(dst []byte) error {
	if len(dst) == 0 {
		return nil
	}
	if err := b.ctx.err(C.ah_download_async(b.ctx.c, unsafe.Pointer(&dst[0]), b.Ptr, C.size_t(len(dst)))); err != nil {
		return err
	}
	return b.ctx.Sync()
}

func (x *Context) Sync() error { return x.err(C.ah_sync(x.c)) }

// ---- arrow/math ---------------------------------------------------------------------------

// SumFloat64 replaces _sum_float64_avx2 (arrow/math/float64_avx2_amd64.go:33-42) for a
// device-resident value buffer of n float64.
func (x *Context) SumFloat64(values unsafe.Pointer, n int) (float64, error) {
	var r C.double
	err := x.err(C.ah_sum_float64(x.c, (*C.double)(values), C.size_t(n), &r))
	return float64(r), err
}

func (x *Context) SumInt64(values unsafe.Pointer, n int) (int64, error) {
	var r C.int64_t
	err := x.err(C.ah_sum_int64(x.c, (*C.int64_t)(values), C.size_t(n), &r))
	return int64(r), err
}

// ---- element-wise -------------------------------------------------------------------------

// ArithmeticBinary replaces arithmeticAvx2 (kernels/base_arithmetic_avx2_amd64.go:37-39):
// typ is the arrow.Type id, op the kernels.ArithmeticOp value, all pointers device memory.
func (x *Context) ArithmeticBinary(typ arrow.Type, op int8, l, r, out unsafe.Pointer, n int64) error {
	return x.err(C.ah_arithmetic_binary(x.c, C.int(typ), C.int8_t(op), l, r, out, C.int64_t(n)))
}

// ArithmeticArrScalar: the scalar is HOST memory (one element), exactly like the asm leaf.
func (x *Context) ArithmeticArrScalar(typ arrow.Type, op int8, l, rHost, out unsafe.Pointer, n int64) error {
	return x.err(C.ah_arithmetic_arr_scalar(x.c, C.int(typ), C.int8_t(op), l, rHost, out, C.int64_t(n)))
}

// ArithmeticScalarArr: scalar ∘ array (scalar in HOST memory).
func (x *Context) ArithmeticScalarArr(typ arrow.Type, op int8, lHost, r, out unsafe.Pointer, n int64) error {
	return x.err(C.ah_arithmetic_scalar_arr(x.c, C.int(typ), C.int8_t(op), lHost, r, out, C.int64_t(n)))
}

// Comparison replaces the 12 _comparison_*_avx2 leaves (kernels/scalar_comparison_avx2_amd64.go).
func (x *Context) Comparison(cmpop, shape int, typ arrow.Type, l, r, outBits unsafe.Pointer, n int64, outBitOffset int) error {
	return x.err(C.ah_comparison(x.c, C.int(cmpop), C.int(shape), C.int(typ), l, r, (*C.uint8_t)(outBits), C.int64_t(n), C.int(outBitOffset)))
}

// BitmapAnd replaces bitutil.BitmapAnd (arrow/bitutil/bitmaps.go:601) on device bitmaps.
func (x *Context) BitmapAnd(l unsafe.Pointer, lOff int64, r unsafe.Pointer, rOff int64, out unsafe.Pointer, oOff, n int64) error {
	return x.err(C.ah_bitmap_op(x.c, C.AH_BIT_AND, (*C.uint8_t)(l), C.int64_t(lOff), (*C.uint8_t)(r), C.int64_t(rOff), (*C.uint8_t)(out), C.int64_t(oOff), C.int64_t(n)))
}

// BitmapOp is bitutil.BitmapOp (arrow/bitutil/bitmaps.go:494-521): op = AH_BIT_AND / OR / XOR / AND_NOT / XNOR.
func (x *Context) BitmapOp(op int, l unsafe.Pointer, lOff int64, r unsafe.Pointer, rOff int64, out unsafe.Pointer, oOff, n int64) error {
	return x.err(C.ah_bitmap_op(x.c, C.int(op), (*C.uint8_t)(l), C.int64_t(lOff), (*C.uint8_t)(r), C.int64_t(rOff), (*C.uint8_t)(out), C.int64_t(oOff), C.int64_t(n)))
}

// Kleene is computeKleene (kernels/scalar_boolean.go:29-65): op = AH_KLEENE_AND / OR / AND_NOT; writes validity AND data.
func (x *Context) Kleene(op int, lvalid, ldata unsafe.Pointer, lOff int64, rvalid, rdata unsafe.Pointer, rOff int64, ovalid, odata unsafe.Pointer, oOff, n int64) error {
	return x.err(C.ah_kleene(x.c, C.int(op), (*C.uint8_t)(lvalid), (*C.uint8_t)(ldata), C.int64_t(lOff), (*C.uint8_t)(rvalid), (*C.uint8_t)(rdata),
		C.int64_t(rOff), (*C.uint8_t)(ovalid), (*C.uint8_t)(odata), C.int64_t(oOff), C.int64_t(n)))
}

func (x *Context) CountSetBits(bits unsafe.Pointer, off, n int64) (int64, error) {
	var r C.int64_t
	err := x.err(C.ah_count_set_bits(x.c, (*C.uint8_t)(bits), C.int64_t(off), C.int64_t(n), &r))
	return int64(r), err
}

// ---- selection ----------------------------------------------------------------------------

// FilterCount == getFilterOutputSize (kernels/vector_selection.go:57-81).
func (x *Context) FilterCount(fdata, fvalid unsafe.Pointer, foff, n int64, nullSel int) (int64, error) {
	var r C.int64_t
	err := x.err(C.ah_filter_count(x.c, (*C.uint8_t)(fdata), (*C.uint8_t)(fvalid), C.int64_t(foff), C.int64_t(n), C.int(nullSel), &r))
	return int64(r), err
}

// FilterPrimitive == primitiveFilterImpl (vector_selection.go:267-395); nOut from FilterCount.
func (x *Context) FilterPrimitive(byteWidth int, values, vvalid unsafe.Pointer, voff int64, fdata, fvalid unsafe.Pointer,
	foff, n int64, nullSel int, nOut int64, outValues, outValid unsafe.Pointer) (nulls int64, err error) {
	var r C.int64_t
	err = x.err(C.ah_filter_primitive(x.c, C.int(byteWidth), values, (*C.uint8_t)(vvalid), C.int64_t(voff), (*C.uint8_t)(fdata),
		(*C.uint8_t)(fvalid), C.int64_t(foff), C.int64_t(n), C.int(nullSel), C.int64_t(nOut), outValues, (*C.uint8_t)(outValid), &r))
	return int64(r), err
}

// FilterToIndices == GetTakeIndices (vector_selection.go:102-236), uint32 flavour: one index vector gathers every column of
// a record batch (FilterRecordBatch, compute/selection.go:679-722).
func (x *Context) FilterToIndices(fdata, fvalid unsafe.Pointer, foff, n int64, nullSel int, nOut int64, outIdx, outValid unsafe.Pointer) (nulls int64, err error) {
	var r C.int64_t
	err = x.err(C.ah_filter_to_indices(x.c, (*C.uint8_t)(fdata), (*C.uint8_t)(fvalid), C.int64_t(foff), C.int64_t(n), C.int(nullSel), C.int64_t(nOut),
		(*C.uint32_t)(outIdx), (*C.uint8_t)(outValid), &r))
	return int64(r), err
}

// TakePrimitive == PrimitiveTake (vector_selection.go:1162-1192) with the bounds check fused.
func (x *Context) TakePrimitive(byteWidth int, values, vvalid unsafe.Pointer, voff, nvalues int64, idxWidth int, idxSigned bool,
	idx, ivalid unsafe.Pointer, ioff, nidx int64, outValues, outValid unsafe.Pointer) (nulls int64, err error) {
	var r, bad C.int64_t
	s := C.int(0)
	if idxSigned {
		s = 1
	}
	err = x.err(C.ah_take_primitive(x.c, C.int(byteWidth), values, (*C.uint8_t)(vvalid), C.int64_t(voff), C.int64_t(nvalues),
		C.int(idxWidth), s, idx, (*C.uint8_t)(ivalid), C.int64_t(ioff), C.int64_t(nidx), 1, outValues, (*C.uint8_t)(outValid), &r, &bad))
	return int64(r), err
}

// FilterPrimitiveDev / TakePrimitiveDev: the same kernels without the host round trip — outputs sized for the worst case,
// {selected rows, nulls} resp. {position of the first out-of-range index or MaxUint64, nulls} left in 16 bytes of device memory.
func (x *Context) FilterPrimitiveDev(byteWidth int, values, vvalid unsafe.Pointer, voff int64, fdata, fvalid unsafe.Pointer,
	foff, n int64, nullSel int, outValues, outValid, statusDev unsafe.Pointer) error {
	return x.err(C.ah_filter_primitive_dev(x.c, C.int(byteWidth), values, (*C.uint8_t)(vvalid), C.int64_t(voff), (*C.uint8_t)(fdata),
		(*C.uint8_t)(fvalid), C.int64_t(foff), C.int64_t(n), C.int(nullSel), outValues, (*C.uint8_t)(outValid), (*C.int64_t)(statusDev)))
}

// FilterPrimitiveOnce: PrimitiveFilter in ONE synchronous call for a caller that sizes its outputs for n rows — no count call, no
// second launch after the host has heard back; the rows selected and the output null count come back through the mailbox.
func (x *Context) FilterPrimitiveOnce(byteWidth int, values, vvalid unsafe.Pointer, voff int64, fdata, fvalid unsafe.Pointer,
	foff, n int64, nullSel int, outValues, outValid unsafe.Pointer) (nOut, nulls int64, err error) {
	var k, r C.int64_t
	err = x.err(C.ah_filter_primitive_once(x.c, C.int(byteWidth), values, (*C.uint8_t)(vvalid), C.int64_t(voff), (*C.uint8_t)(fdata),
		(*C.uint8_t)(fvalid), C.int64_t(foff), C.int64_t(n), C.int(nullSel), outValues, (*C.uint8_t)(outValid), &k, &r))
	return int64(k), int64(r), err
}

func (x *Context) TakePrimitiveDev(byteWidth int, values, vvalid unsafe.Pointer, voff, nvalues int64, idxWidth int, idxSigned bool,
	idx, ivalid unsafe.Pointer, ioff, nidx int64, outValues, outValid, statusDev unsafe.Pointer) error {
	s := C.int(0)
	if idxSigned {
		s = 1
	}
	return x.err(C.ah_take_primitive_dev(x.c, C.int(byteWidth), values, (*C.uint8_t)(vvalid), C.int64_t(voff), C.int64_t(nvalues),
		C.int(idxWidth), s, idx, (*C.uint8_t)(ivalid), C.int64_t(ioff), C.int64_t(nidx), outValues, (*C.uint8_t)(outValid), (*C.uint64_t)(statusDev)))
}

// CmpFilterSumInt64 is the fused Compare→Filter→Sum (no reference analogue).
func (x *Context) CmpFilterSumInt64(cmpop int, values, valid unsafe.Pointer, off, n, threshold int64) (sum, count int64, err error) {
	var s, c C.int64_t
	err = x.err(C.ah_cmp_filter_sum_i64(x.c, C.int(cmpop), (*C.int64_t)(values), (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), C.int64_t(threshold), &s, &c))
	return int64(s), int64(c), err
}

// CumulativeSum mirrors kernels.cumulativeSumExec (vector_cumulative.go:340-350): start is one
// element of the input type (nil = zero); outValid must be given iff valid is.
func (x *Context) CumulativeSum(typ arrow.Type, values, valid unsafe.Pointer, off, n int64, start unsafe.Pointer, skipNulls, checked bool,
	out, outValid unsafe.Pointer) (nulls int64, err error) {
	var c C.int64_t
	st := C.ah_cumulative_sum(x.c, C.int(typ), values, (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), start,
		boolInt(skipNulls), boolInt(checked), out, (*C.uint8_t)(outValid), &c)
	return int64(c), x.err(st)
}

// CastNumeric mirrors castNumericUnsafe + the safe-cast checks (numeric_cast.go:37-71): the error
// carries the reference's text ("integer value %d not in range: %d to %d", …) as arrow.ErrInvalid.
func (x *Context) CastNumeric(in, out arrow.Type, values, valid unsafe.Pointer, off, n int64, opts compute.CastOptions, dst unsafe.Pointer) error {
	return x.err(C.ah_cast_numeric(x.c, C.int(in), C.int(out), values, (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n),
		boolInt(opts.AllowIntOverflow), boolInt(opts.AllowFloatTruncate), dst))
}

// ShiftTime mirrors kernels.ShiftTime[InT, OutT] (cast_temporal.go:35-104), the leaf of the timestamp / duration /
// time32↔time64 / date32↔date64 unit casts: inBits / outBits are the storage widths, op arrow.ConvMULTIPLY or
// arrow.ConvDIVIDE.  The checks follow opts.AllowTimeOverflow / AllowTimeTruncate; a failure comes back as
// arrow.ErrInvalid with the reference's text built around the first offending value.
func (x *Context) ShiftTime(inBits, outBits int, op arrow.TimestampConvertOp, factor int64, opts compute.CastOptions,
	inType, outType arrow.DataType, values, valid unsafe.Pointer, off, n int64, dst unsafe.Pointer) error {
	cop, check := C.int(C.AH_SHIFT_MULTIPLY), !opts.AllowTimeOverflow
	if op == arrow.ConvDIVIDE {
		cop, check = C.int(C.AH_SHIFT_DIVIDE), !opts.AllowTimeTruncate
	}
	var bad C.int64_t
	rc := C.ah_shift_time(x.c, C.int(inBits), C.int(outBits), cop, C.int64_t(factor), boolInt(check), values, (*C.uint8_t)(valid),
		C.int64_t(off), C.int64_t(n), dst, &bad)
	if rc == C.AH_EINVALID && check {
		what := "would result in out of bounds timestamp"
		if op == arrow.ConvDIVIDE {
			what = "would lose data"
		}
		return fmt.Errorf("%w: casting from %s to %s %s: %v", arrow.ErrInvalid, inType, outType, what, int64(bad))
	}
	return x.err(rc)
}

// ArithmeticExt covers the pure-Go arithmetic kernels that have no assembly leaf: divide, abs, negate,
// bit-wise ops, shifts, sqrt, floor / ceil / trunc (base_arithmetic.go:154-160,287-340,386-426;
// scalar_arithmetic.go:170-378; rounding.go:180-187).  op: AH_OP_* of include/arrowhip.h.  Errors carry
// the reference's text ("divide by zero", "overflow", …) as arrow.ErrInvalid.
func (x *Context) ArithmeticExt(typ arrow.Type, op, shape int, l, lvalid unsafe.Pointer, loff int64, r, rvalid unsafe.Pointer, roff int64,
	scalarValid bool, out unsafe.Pointer, n int64) error {
	return x.err(C.ah_arithmetic_ext(x.c, C.int(typ), C.int(op), C.int(shape), l, (*C.uint8_t)(lvalid), C.int64_t(loff),
		r, (*C.uint8_t)(rvalid), C.int64_t(roff), boolInt(scalarValid), out, C.int64_t(n)))
}

// HashU64Encode mirrors doAppendNumeric[uint64] over hashing.Table[uint64] (vector_hash.go:359-385): int32 ids in FIRST-SEEN
// order + the dictionary; nullID = the id null received (-1: none).  outIDsValid / outDict may be nil.
func (x *Context) HashU64Encode(keys, valid unsafe.Pointer, off, n int64, encodeNulls bool, outIDs, outIDsValid, outDict unsafe.Pointer) (ndict int64, nullID int32, err error) {
	var nd C.int64_t
	var nid C.int32_t
	st := C.ah_hash_u64_encode(x.c, (*C.uint64_t)(keys), (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), boolInt(encodeNulls), (*C.int32_t)(outIDs),
		(*C.uint8_t)(outIDsValid), (*C.uint64_t)(outDict), &nd, &nid)
	return int64(nd), int32(nid), x.err(st)
}

// HashSumFloat64 is the group-by sum (no reference analogue; definition in DESIGN.md §4): groups in first-seen order of the
// keys, per group the sum of the valid values (bit-reproducible: 128-bit fixed point), their count and the first row.
func (x *Context) HashSumFloat64(keys, kvalid unsafe.Pointer, koff int64, vals, vvalid unsafe.Pointer, voff, n int64,
	outKeys, outSums, outCounts, outFirstRows unsafe.Pointer) (ngroups int64, nullGroup int32, err error) {
	var ng C.int64_t
	var nid C.int32_t
	st := C.ah_hash_sum_f64(x.c, (*C.uint64_t)(keys), (*C.uint8_t)(kvalid), C.int64_t(koff), (*C.double)(vals), (*C.uint8_t)(vvalid), C.int64_t(voff), C.int64_t(n),
		(*C.uint64_t)(outKeys), (*C.double)(outSums), (*C.int64_t)(outCounts), (*C.int64_t)(outFirstRows), &ng, &nid)
	return int64(ng), int32(nid), x.err(st)
}

// IsIn mirrors isInKernelExec (kernels/scalar_set_lookup.go:374-413): nullBehavior = AH_NULL_MATCH / SKIP / EMIT_NULL / INCONCLUSIVE.
func (x *Context) IsIn(byteWidth int, values, valid unsafe.Pointer, off, n int64, setValues, setValid unsafe.Pointer, setOff, setN int64, nullBehavior int,
	outData, outValid unsafe.Pointer, outBitOffset int64) error {
	return x.err(C.ah_is_in(x.c, C.int(byteWidth), values, (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), setValues, (*C.uint8_t)(setValid), C.int64_t(setOff),
		C.int64_t(setN), C.int(nullBehavior), (*C.uint8_t)(outData), (*C.uint8_t)(outValid), C.int64_t(outBitOffset)))
}

// HashBinaryEncode mirrors doAppendBinary over BinaryMemoTable (vector_hash.go:288-325): ids in first-seen
// order and, per dictionary entry, the row that first held it; the dictionary itself is
// TakeBinary(values, firstRows[:ndict]).
func (x *Context) HashBinaryEncode(offsetWidth int, offsets, data, valid unsafe.Pointer, off, n int64, encodeNulls bool,
	outIDs, outIDsValid, outFirstRows unsafe.Pointer) (ndict int64, nullID int32, err error) {
	var nd C.int64_t
	var nid C.int32_t
	st := C.ah_hash_binary_encode(x.c, C.int(offsetWidth), offsets, (*C.uint8_t)(data), (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n),
		boolInt(encodeNulls), (*C.int32_t)(outIDs), (*C.uint8_t)(outIDsValid), (*C.int64_t)(outFirstRows), &nd, &nid)
	return int64(nd), int32(nid), x.err(st)
}

// SortIndices mirrors the single-column path of sortIndicesMetaFunc (compute/vector_sort.go:110-215):
// a stable order, uint64 indices.
func (x *Context) SortIndices(typ arrow.Type, values, valid unsafe.Pointer, off, n int64, descending, nullsAtStart bool, outIndices unsafe.Pointer) error {
	return x.err(C.ah_sort_indices(x.c, C.int(typ), values, (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), boolInt(descending),
		boolInt(nullsAtStart), (*C.uint64_t)(outIndices)))
}

// CopyDevice is the device-to-device copy array.Concatenate needs when chunked inputs are laid end to end in HBM.
func (x *Context) CopyDevice(dst, src unsafe.Pointer, nbytes int) error {
	return x.err(C.ah_copy_async(x.c, dst, src, C.size_t(nbytes)))
}

func boolInt(b bool) C.int {
	if b {
		return 1
	}
	return 0
}
