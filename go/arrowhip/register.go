//go:build hip

package arrowhip

import (
	"context"
	"fmt"
	"unsafe"

	"github.com/apache/arrow-go/v18/arrow"
	"github.com/apache/arrow-go/v18/arrow/bitutil"
	"github.com/apache/arrow-go/v18/arrow/compute"
	"github.com/apache/arrow-go/v18/arrow/compute/exec"
	"github.com/apache/arrow-go/v18/arrow/scalar"
)

// EXPERIMENTAL — never compiled (no Go toolchain in the build image).  Every ExecFn below is the Go half of a
// contract whose C half is exercised end to end by arrow_go_amd/host/kernels.cc (same executor rules, same
// entry points) and the pytest suite.
//
// Two registration routes (SURVEY.md §8b):
//   Register     — route (i): new functions ("add_hip", "add_unchecked_hip", "greater_hip", "array_filter_hip", …) in a CHILD
//                  registry (registry.go:69-73) carried by the returned context's ExecCtx (executor.go:110-112).
//   SwapInPlace  — route (ii): the stock functions keep their names, dispatch and DispatchBest promotion; only the
//                  ExecFn of the kernels this library covers is replaced, through funcImpl.Kernels()
//                  (functions.go:220-226), which hands out pointers into the live kernel slice.

// ArithmeticOp / CompareOperator values: kernels.* is internal (base_arithmetic.go:37-82, scalar_comparisons.go:33-40).
const (
	opAdd, opSub, opMul      int8 = 0, 1, 2
	cmpEQ, cmpNE, cmpGT, cmpGE    = 0, 1, 2, 3
	shapeAA, shapeAS, shapeSA     = 0, 1, 2
)

var numeric = []arrow.DataType{arrow.PrimitiveTypes.Int8, arrow.PrimitiveTypes.Uint8, arrow.PrimitiveTypes.Int16, arrow.PrimitiveTypes.Uint16,
	arrow.PrimitiveTypes.Int32, arrow.PrimitiveTypes.Uint32, arrow.PrimitiveTypes.Int64, arrow.PrimitiveTypes.Uint64,
	arrow.PrimitiveTypes.Float32, arrow.PrimitiveTypes.Float64}

func width(dt arrow.DataType) int { return dt.(arrow.FixedWidthDataType).BitWidth() / 8 }

// dev is an operand staged in HBM for the duration of one ExecFn: values (element 0 = row 0 of the span) and, when the
// span may have nulls, its validity bytes with the span's bit offset kept (the C ABI takes arbitrary bit offsets).
// Arrays that should stay resident across calls are better kept as DeviceBuffers and fed to the Context methods
// directly: arrow-go has no device-array type to carry them through CallFunction (INTEGRATION.md "where the time goes").
type dev struct {
	vals, valid *DeviceBuffer
	voff        int64 // bit offset of row 0 inside valid
}

func (d *dev) free() {
	if d.vals != nil {
		d.vals.Free()
	}
	if d.valid != nil {
		d.valid.Free()
	}
}
func (d *dev) validPtr() unsafe.Pointer {
	if d.valid == nil {
		return nil
	}
	return d.valid.Ptr
}

// PinUploads: buffers that did NOT come from a PinnedAllocator are pinned (hipHostRegister) for the duration of their upload, so
// the copy is a DMA from the Arrow buffer itself instead of the runtime's staged copy.  Arrays built with
// memory.Allocator = &PinnedAllocator{Ctx: x} need nothing: their buffers are pinned from birth.  Columns too large to wait for
// go through Ingest (ingest.go): chunk k + 1 uploads while chunk k computes.
var PinUploads = false

// IngestThresholdBytes: operands at least this large go through the chunked, overlapped ingest instead of one whole upload.
var IngestThresholdBytes = 64 << 20

// ingest returns the context's pipeline (32 MiB chunks, 3 slots), created on first use.
func (x *Context) ingest() (*Ingest, error) {
	if x.ing == nil {
		g, err := x.NewIngest(0, 0)
		if err != nil {
			return nil, err
		}
		x.ing = g
	}
	return x.ing, nil
}

func (x *Context) upload(b []byte) (*DeviceBuffer, error) {
	d, err := x.Alloc(len(b) + 64)
	if err != nil {
		return nil, err
	}
	if PinUploads && len(b) >= 1<<20 {
		if perr := x.Pin(b); perr == nil {
			defer x.Unpin(b)
		}
	}
	if err := d.Upload(b); err != nil {
		d.Free()
		return nil, err
	}
	return d, nil
}

// stage uploads a fixed-width (w bytes; w == 0: bitmap-valued, i.e. Boolean) ArraySpan.
func (x *Context) stage(a *exec.ArraySpan, w int) (d dev, err error) {
	if w > 0 {
		d.vals, err = x.upload(a.Buffers[1].Buf[int(a.Offset)*w : (int(a.Offset)+int(a.Len))*w])
	} else {
		d.vals, err = x.upload(a.Buffers[1].Buf[a.Offset/8 : bitutil.BytesForBits(a.Offset+a.Len)])
	}
	if err == nil && a.MayHaveNulls() {
		d.valid, err = x.upload(a.Buffers[0].Buf[a.Offset/8 : bitutil.BytesForBits(a.Offset+a.Len)])
	}
	d.voff = a.Offset % 8
	if err != nil {
		d.free()
	}
	return
}

// ---- scalar kernels: NullIntersection + MemPrealloc (exec/kernel.go:660-661) -----------------------------------------
// The executor has allocated and zeroed out.Buffers[1] and computed the validity (propagateNulls); the kernel fills values.

// binaryExec == ScalarBinary over _arithmetic_*_avx2 (kernels/helpers.go:193-236): all three operand shapes.
func binaryExec(x *Context, typ arrow.Type, w int, op int8) exec.ArrayKernelExec {
	return func(_ *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error {
		n := out.Len
		ob := out.Buffers[1].Buf[int(out.Offset)*w : (int(out.Offset)+int(n))*w]
		l, r := &batch.Values[0], &batch.Values[1]
		// A large span of two arrays never sits in HBM whole: the chunked ingest (ingest.go → ah_ingest_arithmetic_binary) uploads
		// chunk k + 1 while chunk k computes and chunk k − 1 downloads — 0.95 of the PCIe link instead of 0.66 for
		// upload-all / compute / download-all (INTEGRATION.md §2a).  Buffers from a PinnedAllocator overlap as they are; others are
		// pinned for the call when PinUploads is set.
		if l.IsArray() && r.IsArray() && int(n)*w >= IngestThresholdBytes {
			ing, err := x.ingest()
			if err != nil {
				return err
			}
			lb := l.Array.Buffers[1].Buf[int(l.Array.Offset)*w : (int(l.Array.Offset)+int(n))*w]
			rb := r.Array.Buffers[1].Buf[int(r.Array.Offset)*w : (int(r.Array.Offset)+int(n))*w]
			if PinUploads {
				for _, b := range [][]byte{lb, rb, ob} {
					if perr := x.Pin(b); perr == nil {
						defer x.Unpin(b)
					}
				}
			}
			return ing.ArithmeticBinary(typ, op, lb, rb, ob, n)
		}
		do, err := x.Alloc(len(ob) + 64) // after the ingest branch: a large span never has its whole output in HBM
		if err != nil {
			return err
		}
		defer do.Free()
		var kerr error
		switch {
		case l.IsArray() && r.IsArray():
			dl, err := x.stage(&l.Array, w)
			if err != nil {
				return err
			}
			defer dl.free()
			dr, err := x.stage(&r.Array, w)
			if err != nil {
				return err
			}
			defer dr.free()
			kerr = x.ArithmeticBinary(typ, op, dl.vals.Ptr, dr.vals.Ptr, do.Ptr, n)
		case l.IsArray(): // array ∘ scalar: the scalar's bytes are host memory, as for the asm leaf
			dl, err := x.stage(&l.Array, w)
			if err != nil {
				return err
			}
			defer dl.free()
			kerr = x.ArithmeticArrScalar(typ, op, dl.vals.Ptr, scalarBytes(r.Scalar), do.Ptr, n)
		default:
			dr, err := x.stage(&r.Array, w)
			if err != nil {
				return err
			}
			defer dr.free()
			kerr = x.ArithmeticScalarArr(typ, op, scalarBytes(l.Scalar), dr.vals.Ptr, do.Ptr, n)
		}
		if kerr != nil {
			return kerr
		}
		return do.Download(ob)
	}
}

// checkedExec == ScalarBinaryNotNull over OpAddChecked / OpSubChecked / OpMulChecked (kernels/base_arithmetic.go:249-286,
// kernels/helpers.go:284-380): the kernels behind the DEFAULT compute.Add / Subtract / Multiply ("add", "subtract",
// "multiply": compute/arithmetic.go:635-636, 1095-1105).  Null slots are skipped (ADD / SUB leave 0 under them), an overflow in a
// valid slot fails the call with arrow.ErrInvalid "overflow" (AH_EOVERFLOW → Context.err); floats have no checked flavour and
// run the unchecked leaf (base_arithmetic_amd64.go:109-117) — the C entry point makes that choice itself.
func checkedExec(x *Context, typ arrow.Type, w int, op int8) exec.ArrayKernelExec {
	return func(_ *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error {
		n := out.Len
		ob := out.Buffers[1].Buf[int(out.Offset)*w : (int(out.Offset)+int(n))*w]
		l, r := &batch.Values[0], &batch.Values[1]
		do, err := x.Alloc(len(ob) + 64)
		if err != nil {
			return err
		}
		defer do.Free()
		var lp, lv, rp, rv unsafe.Pointer
		var lo, ro int64
		shape, scalarValid := shapeAA, true
		if l.IsArray() {
			dl, err := x.stage(&l.Array, w)
			if err != nil {
				return err
			}
			defer dl.free()
			lp, lv, lo = dl.vals.Ptr, dl.validPtr(), dl.voff
		} else {
			lp, shape, scalarValid = scalarBytes(l.Scalar), shapeSA, l.Scalar.IsValid()
		}
		if r.IsArray() {
			dr, err := x.stage(&r.Array, w)
			if err != nil {
				return err
			}
			defer dr.free()
			rp, rv, ro = dr.vals.Ptr, dr.validPtr(), dr.voff
		} else {
			rp, shape, scalarValid = scalarBytes(r.Scalar), shapeAS, r.Scalar.IsValid()
		}
		if err := x.ArithmeticChecked(typ, op, shape, lp, lv, lo, rp, rv, ro, scalarValid, do.Ptr, n); err != nil {
			return err
		}
		return do.Download(ob)
	}
}

// compareExec == compareKernel[T] (kernels/scalar_comparisons.go:199-218): a packed bitmap starting at bit out.Offset;
// only bits [out.Offset%8, +len) of the touched bytes change (the device copy starts from the executor's bytes).
// swap: less / less_equal are greater / greater_equal with the operands exchanged (compute/scalar_compare.go:73-99).
func compareExec(x *Context, typ arrow.Type, w, cmpop int, swap bool) exec.ArrayKernelExec {
	return func(_ *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error {
		l, r := &batch.Values[0], &batch.Values[1]
		if swap {
			l, r = r, l
		}
		ob := out.Buffers[1].Buf[out.Offset/8 : bitutil.BytesForBits(out.Offset+out.Len)]
		do, err := x.upload(ob)
		if err != nil {
			return err
		}
		defer do.Free()
		var lp, rp unsafe.Pointer
		shape := shapeAA
		if l.IsArray() {
			dl, err := x.stage(&l.Array, w)
			if err != nil {
				return err
			}
			defer dl.free()
			lp = dl.vals.Ptr
		} else {
			lp, shape = scalarBytes(l.Scalar), shapeSA
		}
		if r.IsArray() {
			dr, err := x.stage(&r.Array, w)
			if err != nil {
				return err
			}
			defer dr.free()
			rp = dr.vals.Ptr
		} else {
			rp, shape = scalarBytes(r.Scalar), shapeAS
		}
		if err := x.Comparison(cmpop, shape, typ, lp, rp, do.Ptr, out.Len, int(out.Offset%8)); err != nil {
			return err
		}
		return do.Download(ob)
	}
}

// boolExec == SimpleBinary[AndOpKernel …] (kernels/scalar_boolean.go:67-160): the DATA bitmaps combined with
// bitutil.BitmapAnd/Or/Xor/AndNot semantics; validity came from the executor (NullIntersection).
func boolExec(x *Context, bitop int) exec.ArrayKernelExec {
	return func(_ *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error {
		if !batch.Values[0].IsArray() || !batch.Values[1].IsArray() {
			return fmt.Errorf("%w: arrowhip: boolean kernels take arrays (the stock kernel folds scalars)", arrow.ErrNotImplemented)
		}
		dl, err := x.stage(&batch.Values[0].Array, 0)
		if err != nil {
			return err
		}
		defer dl.free()
		dr, err := x.stage(&batch.Values[1].Array, 0)
		if err != nil {
			return err
		}
		defer dr.free()
		ob := out.Buffers[1].Buf[out.Offset/8 : bitutil.BytesForBits(out.Offset+out.Len)]
		do, err := x.upload(ob)
		if err != nil {
			return err
		}
		defer do.Free()
		if err := x.BitmapOp(bitop, dl.vals.Ptr, batch.Values[0].Array.Offset%8, dr.vals.Ptr, batch.Values[1].Array.Offset%8, do.Ptr, out.Offset%8, out.Len); err != nil {
			return err
		}
		return do.Download(ob)
	}
}

// kleeneExec == computeKleene (scalar_boolean.go:29-65) under NullComputedPrealloc: validity AND data are the kernel's.
func kleeneExec(x *Context, op int) exec.ArrayKernelExec {
	return func(_ *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error {
		if !batch.Values[0].IsArray() || !batch.Values[1].IsArray() {
			return fmt.Errorf("%w: arrowhip: kleene kernels take arrays", arrow.ErrNotImplemented)
		}
		l, r := &batch.Values[0].Array, &batch.Values[1].Array
		dl, err := x.stage(l, 0)
		if err != nil {
			return err
		}
		defer dl.free()
		dr, err := x.stage(r, 0)
		if err != nil {
			return err
		}
		defer dr.free()
		vb := out.Buffers[0].Buf[out.Offset/8 : bitutil.BytesForBits(out.Offset+out.Len)]
		ob := out.Buffers[1].Buf[out.Offset/8 : bitutil.BytesForBits(out.Offset+out.Len)]
		dv, err := x.upload(vb)
		if err != nil {
			return err
		}
		defer dv.Free()
		do, err := x.upload(ob)
		if err != nil {
			return err
		}
		defer do.Free()
		if err := x.Kleene(op, dl.validPtr(), dl.vals.Ptr, l.Offset%8, dr.validPtr(), dr.vals.Ptr, r.Offset%8, dv.Ptr, do.Ptr, out.Offset%8, out.Len); err != nil {
			return err
		}
		if err := dv.Download(vb); err != nil {
			return err
		}
		return do.Download(ob)
	}
}

// ---- vector kernels: NullComputedNoPrealloc + MemNoPrealloc (exec/kernel.go:724-725) -----------------------------------
// The kernel sizes its own output: count on the device → ctx.Allocate / AllocateBitmap (the caller's memory.Allocator,
// exec/kernel.go:84-93) → fill → BufferSpan.WrapBuffer (exec/span.go:65-69).

// filterExec == PrimitiveFilter (kernels/vector_selection.go:449-520): getFilterOutputSize, preallocateData,
// primitiveFilterImpl — as ah_filter_count, ctx.Allocate, ah_filter_primitive.
func filterExec(x *Context, w int) exec.ArrayKernelExec {
	return func(ctx *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error {
		values, filter := &batch.Values[0].Array, &batch.Values[1].Array
		nullSel := int(ctx.State.(compute.FilterOptions).NullSelection)
		dv, err := x.stage(values, w)
		if err != nil {
			return err
		}
		defer dv.free()
		df, err := x.stage(filter, 0)
		if err != nil {
			return err
		}
		defer df.free()
		nOut, err := x.FilterCount(df.vals.Ptr, df.validPtr(), df.voff, filter.Len, nullSel)
		if err != nil {
			return err
		}
		withValid := values.MayHaveNulls() || filter.MayHaveNulls() // :486-488
		do, err := x.Alloc(int(nOut)*w + 64)
		if err != nil {
			return err
		}
		defer do.Free()
		var dvo *DeviceBuffer
		var dvoPtr unsafe.Pointer
		if withValid {
			if dvo, err = x.Alloc(int(bitutil.BytesForBits(nOut)) + 64); err != nil {
				return err
			}
			defer dvo.Free()
			dvoPtr = dvo.Ptr
		}
		nulls, err := x.FilterPrimitive(w, dv.vals.Ptr, dv.validPtr(), dv.voff, df.vals.Ptr, df.validPtr(), df.voff, filter.Len, nullSel, nOut, do.Ptr, dvoPtr)
		if err != nil {
			return err
		}
		return finishVector(ctx, out, values.Type, nOut, nulls, w, do, dvo)
	}
}

// takeExec == PrimitiveTake (vector_selection.go:1162-1192); an out-of-range index comes back as arrow.ErrIndex
// "<v> out of bounds" from the fused bounds check (helpers.go:929-981).
func takeExec(x *Context, w int) exec.ArrayKernelExec {
	return func(ctx *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error {
		values, indices := &batch.Values[0].Array, &batch.Values[1].Array
		iw := width(indices.Type)
		dv, err := x.stage(values, w)
		if err != nil {
			return err
		}
		defer dv.free()
		di, err := x.stage(indices, iw)
		if err != nil {
			return err
		}
		defer di.free()
		n := indices.Len
		do, err := x.Alloc(int(n)*w + 64)
		if err != nil {
			return err
		}
		defer do.Free()
		var dvo *DeviceBuffer
		var dvoPtr unsafe.Pointer
		if values.MayHaveNulls() || indices.MayHaveNulls() { // :1176
			if dvo, err = x.Alloc(int(bitutil.BytesForBits(n)) + 64); err != nil {
				return err
			}
			defer dvo.Free()
			dvoPtr = dvo.Ptr
		}
		signed := arrow.IsSignedInteger(indices.Type.ID())
		nulls, err := x.TakePrimitive(w, dv.vals.Ptr, dv.validPtr(), dv.voff, values.Len, iw, signed, di.vals.Ptr, di.validPtr(), di.voff, n, do.Ptr, dvoPtr)
		if err != nil {
			return err
		}
		return finishVector(ctx, out, values.Type, n, nulls, w, do, dvo)
	}
}

// finishVector allocates the result through the caller's allocator and copies the device output into it.
func finishVector(ctx *exec.KernelCtx, out *exec.ExecResult, dt arrow.DataType, n, nulls int64, w int, vals, valid *DeviceBuffer) error {
	out.Type, out.Len, out.Offset, out.Nulls = dt, n, 0, nulls
	data := ctx.Allocate(int(n) * w)
	if err := vals.Download(data.Bytes()); err != nil {
		data.Release()
		return err
	}
	out.Buffers[1].WrapBuffer(data)
	if valid != nil && nulls != 0 {
		bm := ctx.AllocateBitmap(n)
		if err := valid.Download(bm.Bytes()); err != nil {
			bm.Release()
			return err
		}
		out.Buffers[0].WrapBuffer(bm)
	} else {
		out.Nulls = 0
	}
	return nil
}

// hashExec == hashExec + uniqueFinalize / dictionaryEncodeFinalize over hashing.Table[uint64] (kernels/vector_hash.go:
// 359-385, 721-741, 854-873) for 8-byte keys (Int64 / Uint64 / Float64 hash their raw bits, :604-607).  The kernel is
// registered with CanExecuteChunkWise = false: the executor hands it the whole column, as one memo table would see it.
func hashExec(x *Context, encode bool) exec.ArrayKernelExec {
	return func(ctx *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error {
		in := &batch.Values[0].Array
		n := in.Len
		dk, err := x.stage(in, 8)
		if err != nil {
			return err
		}
		defer dk.free()
		encodeNulls := !encode // unique: null owns the id at which it was first seen (GetOrInsertNull :231-238)
		if encode {
			encodeNulls = ctx.State.(compute.DictionaryEncodeOptions).NullEncoding == compute.NullEncodingEncode
		}
		ids, err := x.Alloc(int(n)*4 + 64)
		if err != nil {
			return err
		}
		defer ids.Free()
		dict, err := x.Alloc(int(n+1)*8 + 64)
		if err != nil {
			return err
		}
		defer dict.Free()
		ndict, nullID, err := x.HashU64Encode(dk.vals.Ptr, dk.validPtr(), dk.voff, n, encodeNulls, ids.Ptr, nil, dict.Ptr)
		if err != nil {
			return err
		}
		var dspan exec.ArraySpan // the dictionary: ndict values, the null entry (if any) invalid
		if err := finishVector(ctx, &dspan, in.Type, ndict, 0, 8, dict, nil); err != nil {
			return err
		}
		if nullID >= 0 {
			bm := ctx.AllocateBitmap(ndict)
			bitutil.SetBitsTo(bm.Bytes(), 0, ndict, true)
			bitutil.ClearBit(bm.Bytes(), int(nullID))
			dspan.Buffers[0].WrapBuffer(bm)
			dspan.Nulls = 1
		}
		if !encode {
			*out = dspan
			return nil
		}
		// indices: int32, null where the input was null unless nulls are encoded (dictionaryEncodeAction :224-230)
		if err := finishVector(ctx, out, &arrow.DictionaryType{IndexType: arrow.PrimitiveTypes.Int32, ValueType: in.Type}, n, 0, 4, ids, nil); err != nil {
			return err
		}
		if !encodeNulls && in.MayHaveNulls() {
			bm := ctx.AllocateBitmap(n)
			bitutil.CopyBitmap(in.Buffers[0].Buf, int(in.Offset), int(n), bm.Bytes(), 0)
			out.Buffers[0].WrapBuffer(bm)
			out.Nulls = in.Nulls
		}
		out.SetDictionary(&dspan)
		return nil
	}
}

func scalarBytes(s scalar.Scalar) unsafe.Pointer { // scalar.PrimitiveScalar.Data(): the value's little-endian bytes (host memory)
	return unsafe.Pointer(&s.(scalar.PrimitiveScalar).Data()[0])
}

type funcSpec struct {
	name string
	add  func(fn *compute.ScalarFunction, x *Context) error
}

func perType(types []arrow.DataType, outType func(arrow.DataType) exec.OutputType, mk func(dt arrow.DataType) exec.ArrayKernelExec) func(*compute.ScalarFunction, *Context) error {
	return func(fn *compute.ScalarFunction, _ *Context) error {
		for _, dt := range types {
			in := []exec.InputType{exec.NewExactInput(dt), exec.NewExactInput(dt)}
			if err := fn.AddNewKernel(in, outType(dt), mk(dt), nil); err != nil {
				return err
			}
		}
		return nil
	}
}

// Register installs the GPU kernels in a child registry under "<name>_hip" and returns a context that carries it:
//
//	ctx, _ := arrowhip.Register(context.Background(), gpu)
//	sum, _ := compute.CallFunction(ctx, "add_hip", nil, a, b)            // checked, like compute.Add
//	mask, _ := compute.CallFunction(ctx, "greater_hip", nil, sum, compute.NewDatum(int64(0)))
//	kept, _ := compute.CallFunction(ctx, "array_filter_hip", compute.DefaultFilterOptions(), sum, mask)
//
// New names keep arithmeticFunction.DispatchBest's numeric promotion for the stock functions (a plain ScalarFunction
// registered as "add" would lose it: arithmetic.go:112-142 vs functions.go:260-262); SwapInPlace is the other route.
func Register(parent context.Context, x *Context) (context.Context, error) {
	reg := compute.NewChildRegistry(compute.GetFunctionRegistry())
	same := func(dt arrow.DataType) exec.OutputType { return exec.NewOutputType(dt) }
	boolean := func(arrow.DataType) exec.OutputType { return exec.NewOutputType(arrow.FixedWidthTypes.Boolean) }
	var specs []funcSpec
	for name, op := range map[string]int8{"add_unchecked_hip": opAdd, "subtract_unchecked_hip": opSub, "multiply_unchecked_hip": opMul} {
		op := op
		specs = append(specs, funcSpec{name, perType(numeric, same, func(dt arrow.DataType) exec.ArrayKernelExec { return binaryExec(x, dt.ID(), width(dt), op) })})
	}
	for name, op := range map[string]int8{"add_hip": opAdd, "subtract_hip": opSub, "multiply_hip": opMul} { // the checked defaults
		op := op
		specs = append(specs, funcSpec{name, perType(numeric, same, func(dt arrow.DataType) exec.ArrayKernelExec { return checkedExec(x, dt.ID(), width(dt), op) })})
	}
	for name, c := range map[string]struct {
		op   int
		swap bool
	}{"equal_hip": {cmpEQ, false}, "not_equal_hip": {cmpNE, false}, "greater_hip": {cmpGT, false}, "greater_equal_hip": {cmpGE, false},
		"less_hip": {cmpGT, true}, "less_equal_hip": {cmpGE, true}} {
		c := c
		specs = append(specs, funcSpec{name, perType(numeric, boolean, func(dt arrow.DataType) exec.ArrayKernelExec { return compareExec(x, dt.ID(), width(dt), c.op, c.swap) })})
	}
	bools := []arrow.DataType{arrow.FixedWidthTypes.Boolean}
	for name, bitop := range map[string]int{"and_hip": 0, "or_hip": 1, "xor_hip": 2, "and_not_hip": 3} { // AH_BIT_*
		bitop := bitop
		specs = append(specs, funcSpec{name, perType(bools, boolean, func(arrow.DataType) exec.ArrayKernelExec { return boolExec(x, bitop) })})
	}
	for _, s := range specs {
		fn := compute.NewScalarFunction(s.name, compute.Binary(), compute.FunctionDoc{Summary: "MI355X " + s.name})
		if err := s.add(fn, x); err != nil {
			return nil, err
		}
		if !reg.AddFunction(fn, true) {
			return nil, fmt.Errorf("arrowhip: could not register %s", s.name)
		}
	}
	for name, op := range map[string]int{"and_kleene_hip": 0, "or_kleene_hip": 1, "and_not_kleene_hip": 2} { // AH_KLEENE_*
		fn := compute.NewScalarFunction(name, compute.Binary(), compute.FunctionDoc{Summary: "MI355X " + name})
		k := exec.NewScalarKernel([]exec.InputType{exec.NewExactInput(bools[0]), exec.NewExactInput(bools[0])}, exec.NewOutputType(bools[0]), kleeneExec(x, op), nil)
		k.NullHandling = exec.NullComputedPrealloc // scalar_bool.go:100-110
		if err := fn.AddKernel(k); err != nil {
			return nil, err
		}
		reg.AddFunction(fn, true)
	}
	if err := registerVector(reg, x); err != nil {
		return nil, err
	}
	ectx := compute.DefaultExecCtx()
	ectx.Registry = reg
	return compute.SetExecCtx(parent, ectx), nil
}

func firstType(_ *exec.KernelCtx, args []arrow.DataType) (arrow.DataType, error) { return args[0], nil }

func registerVector(reg compute.FunctionRegistry, x *Context) error {
	filter := compute.NewVectorFunction("array_filter_hip", compute.Binary(), compute.EmptyFuncDoc)
	filter.SetDefaultOptions(compute.DefaultFilterOptions())
	take := compute.NewVectorFunction("array_take_hip", compute.Binary(), compute.EmptyFuncDoc)
	take.SetDefaultOptions(compute.DefaultTakeOptions())
	for _, dt := range numeric {
		fk := exec.NewVectorKernel([]exec.InputType{exec.NewExactInput(dt), exec.NewExactInput(arrow.FixedWidthTypes.Boolean)},
			exec.NewComputedOutputType(firstType), filterExec(x, width(dt)), exec.OptionsInit[compute.FilterOptions])
		if err := filter.AddKernel(fk); err != nil {
			return err
		}
		tk := exec.NewVectorKernel([]exec.InputType{exec.NewExactInput(dt), exec.NewMatchedInput(exec.Integer())},
			exec.NewComputedOutputType(firstType), takeExec(x, width(dt)), exec.OptionsInit[compute.TakeOptions])
		tk.CanExecuteChunkWise = false // selection.go:633
		if err := take.AddKernel(tk); err != nil {
			return err
		}
	}
	unique := compute.NewVectorFunction("unique_hip", compute.Unary(), compute.EmptyFuncDoc)
	encode := compute.NewVectorFunction("dictionary_encode_hip", compute.Unary(), compute.EmptyFuncDoc)
	encode.SetDefaultOptions(&compute.DictionaryEncodeOptions{})
	for _, dt := range []arrow.DataType{arrow.PrimitiveTypes.Int64, arrow.PrimitiveTypes.Uint64, arrow.PrimitiveTypes.Float64} {
		uk := exec.NewVectorKernel([]exec.InputType{exec.NewExactInput(dt)}, exec.NewComputedOutputType(firstType), hashExec(x, false), nil)
		uk.CanExecuteChunkWise, uk.OutputChunked = false, false
		if err := unique.AddKernel(uk); err != nil {
			return err
		}
		dictType := func(_ *exec.KernelCtx, args []arrow.DataType) (arrow.DataType, error) {
			return &arrow.DictionaryType{IndexType: arrow.PrimitiveTypes.Int32, ValueType: args[0]}, nil // outputDictionaryType, vector_hash.go:844-852
		}
		ek := exec.NewVectorKernel([]exec.InputType{exec.NewExactInput(dt)}, exec.NewComputedOutputType(dictType), hashExec(x, true),
			exec.OptionsInit[compute.DictionaryEncodeOptions])
		ek.CanExecuteChunkWise = false // one dictionary for the whole column
		if err := encode.AddKernel(ek); err != nil {
			return err
		}
	}
	for _, fn := range []*compute.VectorFunction{filter, take, unique, encode} {
		if !reg.AddFunction(fn, true) {
			return fmt.Errorf("arrowhip: could not register %s", fn.Name())
		}
	}
	return nil
}

// SwapInPlace is route (ii): replace ONLY the ExecFn of the kernels this library covers inside the process-global
// registry's own functions, so that "add_unchecked", "greater", "array_filter", "array_take" … keep their names,
// arity checks, DispatchBest promotion and option handling.  funcImpl.Kernels() (functions.go:220-226) returns
// pointers into the live kernel slice; appending with AddKernel would not work, dispatch takes the FIRST matching
// signature (functions.go:209-213).  The returned function restores the original ExecFns.
func SwapInPlace(x *Context) (restore func(), err error) {
	reg := compute.GetFunctionRegistry()
	var undo []func()
	swapScalar := func(name string, mk func(dt arrow.DataType) exec.ArrayKernelExec) {
		fn, ok := reg.GetFunction(name)
		if !ok {
			return
		}
		ks, ok := fn.(interface{ Kernels() []*exec.ScalarKernel })
		if !ok {
			return
		}
		for _, k := range ks.Kernels() {
			for _, dt := range numeric {
				if k.Signature.MatchesInputs([]arrow.DataType{dt, dt}) {
					k, old := k, k.ExecFn
					k.ExecFn = mk(dt)
					undo = append(undo, func() { k.ExecFn = old })
					break
				}
			}
		}
	}
	// (compute.Subtract itself calls "sub" / "sub_unchecked", arithmetic.go:1115-1117; "subtract*" are the same kernels under the
	// names the expression layer and Arrow C++ use, arithmetic.go:679-682 — both pairs are swapped)
	for name, op := range map[string]int8{"add_unchecked": opAdd, "subtract_unchecked": opSub, "sub_unchecked": opSub, "multiply_unchecked": opMul} {
		op := op
		swapScalar(name, func(dt arrow.DataType) exec.ArrayKernelExec { return binaryExec(x, dt.ID(), width(dt), op) })
	}
	// "add" / "subtract" / "multiply" — what compute.Add / Subtract / Multiply call unless NoCheckOverflow is set (arithmetic.go:1095-1105)
	for name, op := range map[string]int8{"add": opAdd, "subtract": opSub, "sub": opSub, "multiply": opMul} {
		op := op
		swapScalar(name, func(dt arrow.DataType) exec.ArrayKernelExec { return checkedExec(x, dt.ID(), width(dt), op) })
	}
	for name, c := range map[string]int{"equal": cmpEQ, "not_equal": cmpNE, "greater": cmpGT, "greater_equal": cmpGE} {
		c := c
		swapScalar(name, func(dt arrow.DataType) exec.ArrayKernelExec { return compareExec(x, dt.ID(), width(dt), c, false) })
	}
	// "less" / "less_equal" do NOT follow for free: makeFlippedCompare (scalar_compare.go:84-99) copies each kernel of "greater" /
	// "greater_equal" and captures its ExecFn VALUE in the copy's data (unflippedExec) when the registry is built, so replacing
	// greater's ExecFn afterwards leaves the flipped copies on the CPU loop.  Their own ExecFn (flippedCompare) is replaced here by
	// the same comparison with the operands exchanged.
	for name, c := range map[string]int{"less": cmpGT, "less_equal": cmpGE} {
		c := c
		swapScalar(name, func(dt arrow.DataType) exec.ArrayKernelExec { return compareExec(x, dt.ID(), width(dt), c, true) })
	}
	// The stock selection functions hold ONE kernel for every fixed-width primitive type (exec.Primitive(), vector_selection.go:2340-2360):
	// the wrapper decides per call — integers and float32 / float64 of 1, 2, 4 or 8 bytes go to the HIP kernel of that width, everything
	// else the matcher admits (booleans, float16, temporal and interval types) stays where it was.
	swapVector := func(name string, second arrow.DataType, mk func(w int) exec.ArrayKernelExec) {
		fn, ok := reg.GetFunction(name)
		if !ok {
			return
		}
		ks, ok := fn.(interface{ Kernels() []*exec.VectorKernel })
		if !ok {
			return
		}
		for _, k := range ks.Kernels() {
			matches := false
			for _, dt := range numeric {
				if k.Signature.MatchesInputs([]arrow.DataType{dt, second}) {
					matches = true
					break
				}
			}
			if !matches {
				continue
			}
			k, old := k, k.ExecFn
			hip := map[int]exec.ArrayKernelExec{1: mk(1), 2: mk(2), 4: mk(4), 8: mk(8)}
			k.ExecFn = func(ctx *exec.KernelCtx, b *exec.ExecSpan, o *exec.ExecResult) error {
				t := b.Values[0].Array.Type
				if id := t.ID(); arrow.IsInteger(id) || id == arrow.FLOAT32 || id == arrow.FLOAT64 {
					if h, ok := hip[width(t)]; ok {
						return h(ctx, b, o)
					}
				}
				return old(ctx, b, o)
			}
			undo = append(undo, func() { k.ExecFn = old })
		}
	}
	swapVector("array_filter", arrow.FixedWidthTypes.Boolean, func(w int) exec.ArrayKernelExec { return filterExec(x, w) })
	swapVector("array_take", arrow.PrimitiveTypes.Int32, func(w int) exec.ArrayKernelExec { return takeExec(x, w) })
	return func() {
		for _, u := range undo {
			u()
		}
	}, nil
}
