//go:build hip

package arrowhip

/*
#include "arrowhip.h"
*/
import "C"

import (
	"unsafe"

	"github.com/apache/arrow-go/v18/arrow"
)

// EXPERIMENTAL — never compiled (no Go toolchain in the build image); the C half is exercised by tests/test_ingest.py,
// tests/test_distributed_gpu.py and, from a compiled foreign caller on many threads, tests/cabi_driver.c.
//
// Ingest is ah_ingest: the chunked, overlapped host → HBM pipeline (csrc/ah_ingest.hip).  It plays the role of the executor's
// span loop (ExecCtx.ChunkSize, compute/executor.go:46-64, 658-702) for inputs that live in host memory: chunk k + 1 is on its
// way over PCIe while chunk k is computed and chunk k − 1 goes back.  The host slices handed in must stay alive for the call
// and should come from a PinnedAllocator (allocator.go) — or be pinned for the call with Context.Pin — for the copies to
// overlap; pageable memory gives the same results without the overlap.
type Ingest struct {
	ctx *Context
	g   *C.ah_ingest
}

// NewIngest: chunkBytes = 0 → 32 MiB, depth = 0 → 3 slots.
func (x *Context) NewIngest(chunkBytes, depth int) (*Ingest, error) {
	var g *C.ah_ingest
	if err := x.err(C.ah_ingest_create(x.c, C.size_t(chunkBytes), C.int(depth), &g)); err != nil {
		return nil, err
	}
	return &Ingest{ctx: x, g: g}, nil
}

func (i *Ingest) Close() {
	if i.g != nil {
		C.ah_ingest_destroy(i.g)
		i.g = nil
	}
}

// Pin / Unpin: hipHostRegister over memory the Go side already owns, for the duration of a call (cgo rule: C keeps no
// Go pointer after the call returns — the registration is dropped before it does).
func (x *Context) Pin(b []byte) error {
	if len(b) == 0 {
		return nil
	}
	return x.err(C.ah_host_register(x.c, unsafe.Pointer(&b[0]), C.size_t(len(b))))
}
func (x *Context) Unpin(b []byte) error {
	if len(b) == 0 {
		return nil
	}
	return x.err(C.ah_host_unregister(x.c, unsafe.Pointer(&b[0])))
}

// SumFloat64 == math.Float64.Sum over a host slice (arrow/math/float64.go:34-47): one rounding for the whole column.
func (i *Ingest) SumFloat64(v []float64) (float64, error) {
	var r C.double
	var p *C.double
	if len(v) > 0 {
		p = (*C.double)(unsafe.Pointer(&v[0]))
	}
	err := i.ctx.err(C.ah_ingest_sum_float64(i.g, p, C.size_t(len(v)), &r))
	return float64(r), err
}

// ArithmeticBinary == the unchecked arithmetic leaf (kernels/base_arithmetic_avx2_amd64.go:35-53) with host operands.
func (i *Ingest) ArithmeticBinary(typ arrow.Type, op int8, l, r, out []byte, n int64) error {
	if n == 0 {
		return nil
	}
	return i.ctx.err(C.ah_ingest_arithmetic_binary(i.g, C.int(typ), C.int8_t(op), unsafe.Pointer(&l[0]), unsafe.Pointer(&r[0]), unsafe.Pointer(&out[0]), C.int64_t(n)))
}

// FilterCount / FilterPrimitive == getFilterOutputSize then primitiveFilterImpl (kernels/vector_selection.go:57-81, 267-395) with
// host buffers: the count call leaves the selection vector on the device, the caller allocates (ctx.Allocate), the fill call
// streams the values through.
func (i *Ingest) FilterCount(fdata, fvalid []byte, foff, n int64, nullSel int) (int64, error) {
	var nOut C.int64_t
	var fv *C.uint8_t
	if fvalid != nil {
		fv = (*C.uint8_t)(unsafe.Pointer(&fvalid[0]))
	}
	if n == 0 {
		return 0, nil
	}
	err := i.ctx.err(C.ah_ingest_filter_count(i.g, (*C.uint8_t)(unsafe.Pointer(&fdata[0])), fv, C.int64_t(foff), C.int64_t(n), C.int(nullSel), &nOut))
	return int64(nOut), err
}

func (i *Ingest) FilterPrimitive(w int, values, vvalid []byte, voff, n, nOut int64, outValues, outValid []byte) (nulls int64, err error) {
	if n == 0 {
		return 0, nil
	}
	var vv, ov *C.uint8_t
	if vvalid != nil {
		vv = (*C.uint8_t)(unsafe.Pointer(&vvalid[0]))
	}
	if outValid != nil {
		ov = (*C.uint8_t)(unsafe.Pointer(&outValid[0]))
	}
	var op unsafe.Pointer
	if nOut > 0 {
		op = unsafe.Pointer(&outValues[0])
	}
	var nc C.int64_t
	err = i.ctx.err(C.ah_ingest_filter_primitive(i.g, C.int(w), unsafe.Pointer(&values[0]), vv, C.int64_t(voff), C.int64_t(n), C.int64_t(nOut), op, ov, &nc))
	return int64(nc), err
}
