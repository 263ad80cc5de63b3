//go:build hip

package arrowhip

/*
#include <stdlib.h>
#include "arrowhip.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/apache/arrow-go/v18/arrow"
)

// EXPERIMENTAL — never compiled (no Go toolchain in the build image; tests/test_go_shim_static.py checks the declarations and
// every C call against include/arrowhip.h).  The rest of the header's entry points, one thin method each: all pointers are
// DEVICE memory unless the name says Host.

// ---- context plumbing ------------------------------------------------------------------------------------------------------

func Version() string { return C.GoString(C.ah_version()) }

func DeviceCount() (int, error) {
	var n C.int
	if st := C.ah_device_count(&n); st != C.AH_OK {
		return 0, fmt.Errorf("arrowhip: ah_device_count failed (status %d)", int(st))
	}
	return int(n), nil
}

// NewContextOnStream shares an existing hipStream_t (another library's compute stream) instead of creating one: calls of
// both are ordered by the stream.
func NewContextOnStream(device int, hipStream unsafe.Pointer) (*Context, error) {
	var c *C.ah_ctx
	if st := C.ah_ctx_create_on_stream(C.int(device), hipStream, &c); st != C.AH_OK {
		return nil, fmt.Errorf("arrowhip: ah_ctx_create_on_stream(%d) failed (status %d)", device, int(st))
	}
	return &Context{c: c}, nil
}

func (x *Context) DeviceID() int { return int(C.ah_device_id(x.c)) }

// SetOption: the measurement switches of DESIGN.md §8 ("take_binned", "groupby_partition", …).
func (x *Context) SetOption(name string, value int64) error {
	cs := C.CString(name)
	defer C.free(unsafe.Pointer(cs))
	return x.err(C.ah_ctx_set_option(x.c, cs, C.int64_t(value)))
}

// WaitEvent makes the compute stream wait for a hipEvent_t recorded by another library.
func (x *Context) WaitEvent(hipEvent unsafe.Pointer) error { return x.err(C.ah_wait_event(x.c, hipEvent)) }

func (x *Context) Memset(dst unsafe.Pointer, byteValue int, nbytes int) error {
	return x.err(C.ah_memset_async(x.c, dst, C.int(byteValue), C.size_t(nbytes)))
}

func (x *Context) TimerStart() error { return x.err(C.ah_timer_start(x.c)) }

func (x *Context) TimerStop() (ms float32, err error) {
	var v C.float
	err = x.err(C.ah_timer_stop(x.c, &v))
	return float32(v), err
}

func (x *Context) EventRecord(slot int) error { return x.err(C.ah_event_record(x.c, C.int(slot))) }

func (x *Context) EventElapsedMs(slotA, slotB int) (ms float32, err error) {
	var v C.float
	err = x.err(C.ah_event_elapsed_ms(x.c, C.int(slotA), C.int(slotB), &v))
	return float32(v), err
}

// ---- element-wise ----------------------------------------------------------------------------------------------------------

// ArithmeticChecked is the kernel behind the DEFAULT compute.Add / Subtract / Multiply ("add", "subtract", "multiply" are
// OpAddChecked …: compute/arithmetic.go:635-636, kernels/base_arithmetic.go:249-286): op = opAdd / opSub / opMul, shape
// = shapeAA / AS / SA; for AS / SA the scalar operand is HOST memory and scalarValid says whether it is non-null.  An
// overflow in a valid slot comes back as arrow.ErrInvalid "overflow" (AH_EOVERFLOW), like the reference's errOverflow.
func (x *Context) ArithmeticChecked(typ arrow.Type, op int8, shape int, l, lvalid unsafe.Pointer, loff int64, r, rvalid unsafe.Pointer, roff int64,
	scalarValid bool, out unsafe.Pointer, n int64) error {
	return x.err(C.ah_arithmetic_checked(x.c, C.int(typ), C.int8_t(op), C.int(shape), l, (*C.uint8_t)(lvalid), C.int64_t(loff),
		r, (*C.uint8_t)(rvalid), C.int64_t(roff), boolInt(scalarValid), out, C.int64_t(n)))
}

// ArithmeticUnary: abs_unchecked / negate_unchecked / sign over the _arithmetic_unary_*_avx2 leaves
// (kernels/base_arithmetic_avx2_amd64.go:55-77): op = AH_OP_ABS / AH_OP_NEGATE / AH_OP_SIGN.
func (x *Context) ArithmeticUnary(typ arrow.Type, op int8, in, out unsafe.Pointer, n int64) error {
	return x.err(C.ah_arithmetic_unary(x.c, C.int(typ), C.int8_t(op), in, out, C.int64_t(n)))
}

// Round mirrors round / round_to_multiple (kernels/rounding.go:38-178): mode = compute.RoundMode; multipleHost: one element of
// the column's type in HOST memory for round_to_multiple, nil for round; pow10 = 10^|ndigits| as the reference computes it.
func (x *Context) Round(typ arrow.Type, values, valid unsafe.Pointer, off, n, ndigits int64, mode int, multipleHost unsafe.Pointer, pow10 float64, out unsafe.Pointer) error {
	return x.err(C.ah_round(x.c, C.int(typ), values, (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), C.int64_t(ndigits), C.int(mode), multipleHost, C.double(pow10), out))
}

// MinMax mirrors the _int*_max_min_avx2 leaves (internal/utils/min_max_avx2_amd64.go): outMinHost / outMaxHost receive one
// element of the column's type each.
func (x *Context) MinMax(typ arrow.Type, values unsafe.Pointer, n int64, outMinHost, outMaxHost unsafe.Pointer) error {
	return x.err(C.ah_min_max(x.c, C.int(typ), values, C.int64_t(n), outMinHost, outMaxHost))
}

// CastBoolToNumeric: boolean → number (kernels/numeric_cast.go:614-640), 1 for a set bit.
func (x *Context) CastBoolToNumeric(out arrow.Type, bits unsafe.Pointer, off, n int64, outValues unsafe.Pointer) error {
	return x.err(C.ah_cast_bool_to_numeric(x.c, C.int(out), (*C.uint8_t)(bits), C.int64_t(off), C.int64_t(n), outValues))
}

// CopyBitmap == bitutil.CopyBitmap / InvertBitmap (arrow/bitutil/bitmaps.go:523-560): bits outside [doff, doff+n) keep their value.
func (x *Context) CopyBitmap(src unsafe.Pointer, soff, n int64, dst unsafe.Pointer, doff int64, invert bool) error {
	return x.err(C.ah_copy_bitmap(x.c, (*C.uint8_t)(src), C.int64_t(soff), C.int64_t(n), (*C.uint8_t)(dst), C.int64_t(doff), boolInt(invert)))
}

// SetBitsTo == bitutil.SetBitsTo (arrow/bitutil/bitutil.go:106-143).
func (x *Context) SetBitsTo(bits unsafe.Pointer, off, n int64, value bool) error {
	return x.err(C.ah_set_bits_to(x.c, (*C.uint8_t)(bits), C.int64_t(off), C.int64_t(n), boolInt(value)))
}

// ---- selection -------------------------------------------------------------------------------------------------------------

// TakeBoolean == the boolean flavour of Take (kernels/vector_selection.go:1194-1271): `data` is a bitmap column.
func (x *Context) TakeBoolean(data, vvalid unsafe.Pointer, voff, nvalues int64, idxWidth int, idxSigned bool, idx, ivalid unsafe.Pointer, ioff, nidx int64,
	outData, outValid unsafe.Pointer) (nulls int64, err error) {
	var r, bad C.int64_t
	err = x.err(C.ah_take_boolean(x.c, (*C.uint8_t)(data), (*C.uint8_t)(vvalid), C.int64_t(voff), C.int64_t(nvalues), C.int(idxWidth), boolInt(idxSigned),
		idx, (*C.uint8_t)(ivalid), C.int64_t(ioff), C.int64_t(nidx), 1, (*C.uint8_t)(outData), (*C.uint8_t)(outValid), &r, &bad))
	return int64(r), err
}

// TakeBinaryOffsets / TakeBinaryData == the two passes of the var-length Take (kernels/vector_selection.go:1273-1440 VarBinaryImpl):
// first the output offsets, validity and total byte count; the caller allocates the data buffer; then the bytes.
func (x *Context) TakeBinaryOffsets(offsetWidth int, offsets, vvalid unsafe.Pointer, voff, nvalues int64, idxWidth int, idxSigned bool,
	idx, ivalid unsafe.Pointer, ioff, nidx int64, outOffsets, outValid unsafe.Pointer) (nulls, totalBytes int64, err error) {
	var r, tot, bad C.int64_t
	err = x.err(C.ah_take_binary_offsets(x.c, C.int(offsetWidth), offsets, (*C.uint8_t)(vvalid), C.int64_t(voff), C.int64_t(nvalues), C.int(idxWidth), boolInt(idxSigned),
		idx, (*C.uint8_t)(ivalid), C.int64_t(ioff), C.int64_t(nidx), 1, outOffsets, (*C.uint8_t)(outValid), &r, &tot, &bad))
	return int64(r), int64(tot), err
}

func (x *Context) TakeBinaryData(offsetWidth int, offsets, data unsafe.Pointer, voff int64, idxWidth int, idx unsafe.Pointer, nidx int64, outOffsets, outData unsafe.Pointer) error {
	return x.err(C.ah_take_binary_data(x.c, C.int(offsetWidth), offsets, (*C.uint8_t)(data), C.int64_t(voff), C.int(idxWidth), idx, C.int64_t(nidx), outOffsets, (*C.uint8_t)(outData)))
}

// ---- hashing ---------------------------------------------------------------------------------------------------------------

// HashFixedEncode: unique / dictionary_encode over FixedSizeBinary / Decimal128 / Decimal256 keys (kernels/vector_hash.go:608-609,
// 698: byteWidth-byte keys through BinaryMemoTable); ids in first-seen order, first rows, the dictionary's bytes.
func (x *Context) HashFixedEncode(byteWidth int, data, valid unsafe.Pointer, off, n int64, encodeNulls bool, outIDs, outIDsValid, outFirstRows, outDict unsafe.Pointer) (ndict int64, nullID int32, err error) {
	var nd C.int64_t
	var nid C.int32_t
	st := C.ah_hash_fixed_encode(x.c, C.int(byteWidth), (*C.uint8_t)(data), (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), boolInt(encodeNulls), (*C.int32_t)(outIDs),
		(*C.uint8_t)(outIDsValid), (*C.int64_t)(outFirstRows), (*C.uint8_t)(outDict), &nd, &nid)
	return int64(nd), int32(nid), x.err(st)
}

// HashSumInt64: the group-by sum with Int64 values (wrapping, exact in any order).
func (x *Context) HashSumInt64(keys, kvalid unsafe.Pointer, koff int64, vals, vvalid unsafe.Pointer, voff, n int64,
	outKeys, outSums, outCounts, outFirstRows unsafe.Pointer) (ngroups int64, nullGroup int32, err error) {
	var ng C.int64_t
	var nid C.int32_t
	st := C.ah_hash_sum_i64(x.c, (*C.uint64_t)(keys), (*C.uint8_t)(kvalid), C.int64_t(koff), (*C.int64_t)(vals), (*C.uint8_t)(vvalid), C.int64_t(voff), C.int64_t(n),
		(*C.uint64_t)(outKeys), (*C.int64_t)(outSums), (*C.int64_t)(outCounts), (*C.int64_t)(outFirstRows), &ng, &nid)
	return int64(ng), int32(nid), x.err(st)
}

// HashPartition: partition id of every key by the reference's integer hash (internal/hashing/hash_funcs.go:60-67) — the owner
// function of the C5 merge.
func (x *Context) HashPartition(keys unsafe.Pointer, n int64, nparts int, outPart unsafe.Pointer) error {
	return x.err(C.ah_hash_partition_u64(x.c, (*C.uint64_t)(keys), C.int64_t(n), C.int(nparts), (*C.int32_t)(outPart)))
}

// ---- fused / sort ----------------------------------------------------------------------------------------------------------

// CmpFilterSumFloat64 is the fused Compare→Filter→Sum over Float64 (the sum as math.Float64.Sum would give it, DESIGN.md §4).
func (x *Context) CmpFilterSumFloat64(cmpop int, values, valid unsafe.Pointer, off, n int64, threshold float64) (sum float64, count int64, err error) {
	var s C.double
	var c C.int64_t
	err = x.err(C.ah_cmp_filter_sum_f64(x.c, C.int(cmpop), (*C.double)(values), (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), C.double(threshold), &s, &c))
	return float64(s), int64(c), err
}

// CmpFilterSumInt64Dev / CmpFilterSumFloat64Dev leave {sum, count} in device memory (no host round trip).
func (x *Context) CmpFilterSumInt64Dev(cmpop int, values, valid unsafe.Pointer, off, n, threshold int64, outSumCountDev unsafe.Pointer) error {
	return x.err(C.ah_cmp_filter_sum_i64_dev(x.c, C.int(cmpop), (*C.int64_t)(values), (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), C.int64_t(threshold), (*C.int64_t)(outSumCountDev)))
}

func (x *Context) CmpFilterSumFloat64Dev(cmpop int, values, valid unsafe.Pointer, off, n int64, threshold float64, outSumDev, outCountDev unsafe.Pointer) error {
	return x.err(C.ah_cmp_filter_sum_f64_dev(x.c, C.int(cmpop), (*C.double)(values), (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(n), C.double(threshold),
		(*C.double)(outSumDev), (*C.int64_t)(outCountDev)))
}

// SortKey is one column of a multi-key sort (compute.SortKey, compute/vector_sort.go:40-60).
type SortKey struct {
	Type         arrow.Type
	Values       unsafe.Pointer
	Valid        unsafe.Pointer
	Off          int64
	Descending   bool
	NullsAtStart bool
}

// SortIndicesMulti mirrors the multi-column path of sort_indices (compute/vector_sort.go:217-330): a stable lexicographic order.
// The pointer tables are built in C memory: cgo forbids passing Go memory that itself holds pointers.
func (x *Context) SortIndicesMulti(keys []SortKey, n int64, outIndices unsafe.Pointer) error {
	k := len(keys)
	if k == 0 {
		return fmt.Errorf("%w: arrowhip: SortIndicesMulti wants at least one key", arrow.ErrInvalid)
	}
	ptrSize := C.size_t(unsafe.Sizeof(unsafe.Pointer(nil)))
	vals := C.malloc(C.size_t(k) * ptrSize)
	defer C.free(vals)
	valids := C.malloc(C.size_t(k) * ptrSize)
	defer C.free(valids)
	vslice := unsafe.Slice((*unsafe.Pointer)(vals), k)
	mslice := unsafe.Slice((*unsafe.Pointer)(valids), k)
	types := make([]C.int, k)
	desc := make([]C.int, k)
	nulls := make([]C.int, k)
	offs := make([]C.int64_t, k)
	for i, key := range keys {
		vslice[i], mslice[i] = key.Values, key.Valid
		types[i], desc[i], nulls[i], offs[i] = C.int(key.Type), boolInt(key.Descending), boolInt(key.NullsAtStart), C.int64_t(key.Off)
	}
	return x.err(C.ah_sort_indices_multi(x.c, C.int(k), &types[0], (*unsafe.Pointer)(vals), (**C.uint8_t)(valids), &offs[0], C.int64_t(n), &desc[0], &nulls[0], (*C.uint64_t)(outIndices)))
}

// ---- ingest slots: the building blocks of Ingest for callers that run their own kernels on the chunks ------------------------

func (i *Ingest) Depth() int      { return int(C.ah_ingest_depth(i.g)) }
func (i *Ingest) ChunkBytes() int { return int(C.ah_ingest_chunk_bytes(i.g)) }

// SlotBuffer: device buffer `which` (0, 1 inputs; 2 output) of a slot.
func (i *Ingest) SlotBuffer(slot, which int) unsafe.Pointer {
	return C.ah_ingest_slot_buffer(i.g, C.int(slot), C.int(which))
}

func (i *Ingest) SlotUpload(slot, which int, dstOffset int, host []byte, firstOfChunk bool) error {
	if len(host) == 0 {
		return nil
	}
	return i.ctx.err(C.ah_ingest_slot_upload(i.g, C.int(slot), C.int(which), C.size_t(dstOffset), unsafe.Pointer(&host[0]), C.size_t(len(host)), boolInt(firstOfChunk)))
}

func (i *Ingest) SlotReady(slot int, writesOutput bool) error {
	return i.ctx.err(C.ah_ingest_slot_ready(i.g, C.int(slot), boolInt(writesOutput)))
}

func (i *Ingest) SlotRelease(slot int) error { return i.ctx.err(C.ah_ingest_slot_release(i.g, C.int(slot))) }

func (i *Ingest) SlotDownload(slot, which int, srcOffset int, host []byte) error {
	if len(host) == 0 {
		return nil
	}
	return i.ctx.err(C.ah_ingest_slot_download(i.g, C.int(slot), C.int(which), C.size_t(srcOffset), unsafe.Pointer(&host[0]), C.size_t(len(host))))
}

func (i *Ingest) Wait() error { return i.ctx.err(C.ah_ingest_wait(i.g)) }
