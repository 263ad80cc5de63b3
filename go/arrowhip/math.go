//go:build hip

package arrowhip

/*
#include "arrowhip.h"
*/
import "C"

import (
	"unsafe"

	"github.com/apache/arrow-go/v18/arrow/array"
)

// EXPERIMENTAL — never compiled (no Go toolchain in the build image; tests/test_go_shim_static.py checks the declarations and
// every C call against include/arrowhip.h).
//
// Math has the shape of package arrow/math: math.Float64.Sum(a), math.Int64.Sum(a), math.Uint64.Sum(a) (arrow/math/float64.go:
// 25-39, int64.go:25-39, uint64.go:25-39) become m.Float64.Sum(a), … on a GPU context.  Like the reference the functions take
// the array, ignore its validity bitmap (float64.go:41-47 walks Float64Values()), return 0 for an empty array and return a bare
// value: a HIP failure panics (there is no CPU fallback to return instead); SumErr is the flavour that returns the error.
//
// Float64: the sum over the extended reals rounded once — within 1 ULP of the exact sum, which is where both of the
// reference's summation orders (sequential, 32 strided partials) aim; ±inf, NaN and overflow as IEEE addition gives them in
// either order (DESIGN.md §4).
type Math struct {
	Float64 Float64Funcs
	Int64   Int64Funcs
	Uint64  Uint64Funcs
}

func NewMath(x *Context) *Math {
	return &Math{Float64: Float64Funcs{x}, Int64: Int64Funcs{x}, Uint64: Uint64Funcs{x}}
}

type Float64Funcs struct{ x *Context }
type Int64Funcs struct{ x *Context }
type Uint64Funcs struct{ x *Context }

func (f Float64Funcs) Sum(a *array.Float64) float64 {
	r, err := f.SumErr(a)
	if err != nil {
		panic(err)
	}
	return r
}

// SumErr: the values live in host memory (an arrow-go array), so they stream through the chunked ingest — chunk k + 1 uploads
// while chunk k is summed, every chunk's partials meet in one final reduction.
func (f Float64Funcs) SumErr(a *array.Float64) (float64, error) {
	if a.Len() == 0 {
		return 0, nil // float64.go:35-37
	}
	ing, err := f.x.ingest()
	if err != nil {
		return 0, err
	}
	return ing.SumFloat64(a.Float64Values())
}

func (f Int64Funcs) Sum(a *array.Int64) int64 {
	r, err := f.SumErr(a)
	if err != nil {
		panic(err)
	}
	return r
}

func (f Int64Funcs) SumErr(a *array.Int64) (int64, error) {
	if a.Len() == 0 {
		return 0, nil
	}
	ing, err := f.x.ingest()
	if err != nil {
		return 0, err
	}
	return ing.SumInt64(a.Int64Values())
}

func (f Uint64Funcs) Sum(a *array.Uint64) uint64 {
	r, err := f.SumErr(a)
	if err != nil {
		panic(err)
	}
	return r
}

// SumErr: a wrapping sum mod 2^64 is the same bits signed or unsigned (uint64.go:41-47).
func (f Uint64Funcs) SumErr(a *array.Uint64) (uint64, error) {
	if a.Len() == 0 {
		return 0, nil
	}
	ing, err := f.x.ingest()
	if err != nil {
		return 0, err
	}
	v := a.Uint64Values()
	r, err := ing.SumInt64(unsafe.Slice((*int64)(unsafe.Pointer(&v[0])), len(v)))
	return uint64(r), err
}

// SumInt64 == math.Int64.Sum over a host slice (arrow/math/int64.go:34-47).
func (i *Ingest) SumInt64(v []int64) (int64, error) {
	var r C.int64_t
	var p *C.int64_t
	if len(v) > 0 {
		p = (*C.int64_t)(unsafe.Pointer(&v[0]))
	}
	err := i.ctx.err(C.ah_ingest_sum_int64(i.g, p, C.size_t(len(v)), &r))
	return int64(r), err
}

// SumUint64 / SumFloat64Dev / SumInt64Dev: device-resident value buffers; the *Dev flavours leave the result in 8 bytes of
// device memory (no host round trip — usable inside Context.Record).
func (x *Context) SumUint64(values unsafe.Pointer, n int) (uint64, error) {
	var r C.uint64_t
	err := x.err(C.ah_sum_uint64(x.c, (*C.uint64_t)(values), C.size_t(n), &r))
	return uint64(r), err
}

func (x *Context) SumFloat64Dev(values unsafe.Pointer, n int, resDev unsafe.Pointer) error {
	return x.err(C.ah_sum_float64_dev(x.c, (*C.double)(values), C.size_t(n), (*C.double)(resDev)))
}

func (x *Context) SumInt64Dev(values unsafe.Pointer, n int, resDev unsafe.Pointer) error {
	return x.err(C.ah_sum_int64_dev(x.c, (*C.int64_t)(values), C.size_t(n), (*C.int64_t)(resDev)))
}
