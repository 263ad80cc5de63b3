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
// every C call against include/arrowhip.h); the C half is exercised by tests/test_expressions.py.
//
// Expr is ah_expr: a scalar expression tree compiled into ONE kernel (csrc/ah_expr.hip) — what
// exprs.ExecuteScalarExpression computes with one kernel and one materialised intermediate per call node
// (arrow/compute/exprs/exec.go:440-700).  ExprNode is one step of the postfix program: Op = AH_X_* of include/arrowhip.h,
// Arg = column index (AH_X_COL), literal index (AH_X_LIT) or target type (AH_X_CAST).
type ExprNode struct{ Op, Arg int32 }

type Expr struct {
	ctx     *Context
	e       *C.ah_expr
	OutType arrow.Type
	ncols   int
	nlits   int
}

func cIntTypes(ts []arrow.Type) []C.int {
	out := make([]C.int, len(ts)+1) // never empty: &out[0] must exist
	for i, t := range ts {
		out[i] = C.int(t)
	}
	return out
}

// ExprCompile: the handle is cached per context by (program, types) and owned by it.
func (x *Context) ExprCompile(nodes []ExprNode, colTypes, litTypes []arrow.Type) (*Expr, error) {
	if len(nodes) == 0 {
		return nil, fmt.Errorf("%w: arrowhip: empty expression program", arrow.ErrInvalid)
	}
	ct, lt := cIntTypes(colTypes), cIntTypes(litTypes)
	var e *C.ah_expr
	var ot C.int
	st := C.ah_expr_compile(x.c, (*C.ah_expr_node)(unsafe.Pointer(&nodes[0])), C.int(len(nodes)), &ct[0], C.int(len(colTypes)), &lt[0], C.int(len(litTypes)), &e, &ot)
	if err := x.err(st); err != nil {
		return nil, err
	}
	return &Expr{ctx: x, e: e, OutType: arrow.Type(ot), ncols: len(colTypes), nlits: len(litTypes)}, nil
}

// Execute: colValues / colValid are device pointers (colValid[i] nil = no nulls), BOOL columns and validities are addressed
// with bit offset colOffsets[i]; litValues are the literals' little-endian payloads in 8 bytes each.  outValid may be nil when
// no input can be null.  A checked node that overflows in a valid slot is arrow.ErrInvalid "overflow".
func (e *Expr) Execute(colValues, colValid []unsafe.Pointer, colOffsets []int64, litValues []uint64, litValid []bool, n int64, outValues, outValid unsafe.Pointer) error {
	if len(colValues) != e.ncols || len(colValid) != e.ncols || len(colOffsets) != e.ncols || len(litValues) != e.nlits || len(litValid) != e.nlits {
		return fmt.Errorf("%w: arrowhip: expression wants %d columns and %d literals", arrow.ErrInvalid, e.ncols, e.nlits)
	}
	ptrSize := C.size_t(unsafe.Sizeof(unsafe.Pointer(nil)))
	vals := C.malloc(C.size_t(e.ncols+1) * ptrSize) // pointer tables live in C memory (cgo pointer rule)
	defer C.free(vals)
	valids := C.malloc(C.size_t(e.ncols+1) * ptrSize)
	defer C.free(valids)
	vslice := unsafe.Slice((*unsafe.Pointer)(vals), e.ncols+1)
	mslice := unsafe.Slice((*unsafe.Pointer)(valids), e.ncols+1)
	offs := make([]C.int64_t, e.ncols+1)
	for i := 0; i < e.ncols; i++ {
		vslice[i], mslice[i], offs[i] = colValues[i], colValid[i], C.int64_t(colOffsets[i])
	}
	lv := make([]C.uint64_t, e.nlits+1)
	lok := make([]C.int, e.nlits+1)
	for i := 0; i < e.nlits; i++ {
		lv[i], lok[i] = C.uint64_t(litValues[i]), boolInt(litValid[i])
	}
	return e.ctx.err(C.ah_expr_execute(e.ctx.c, e.e, (*unsafe.Pointer)(vals), (**C.uint8_t)(valids), &offs[0], unsafe.Pointer(&lv[0]), &lok[0], C.int64_t(n), outValues, (*C.uint8_t)(outValid)))
}

// Source: the generated HIP source, for inspection.
func (e *Expr) Source() string { return C.GoString(C.ah_expr_source(e.e)) }

// ExprCodegen generates (and, with compile, hiprtc-compiles for gfx950) a program without a GPU or a context.
func ExprCodegen(nodes []ExprNode, colTypes, litTypes []arrow.Type, compile bool) (src string, outType arrow.Type, err error) {
	if len(nodes) == 0 {
		return "", 0, fmt.Errorf("%w: arrowhip: empty expression program", arrow.ErrInvalid)
	}
	ct, lt := cIntTypes(colTypes), cIntTypes(litTypes)
	const srcCap, errCap = 1 << 16, 1 << 12
	sb := (*C.char)(C.malloc(srcCap))
	defer C.free(unsafe.Pointer(sb))
	eb := (*C.char)(C.malloc(errCap))
	defer C.free(unsafe.Pointer(eb))
	var ot C.int
	st := C.ah_expr_codegen((*C.ah_expr_node)(unsafe.Pointer(&nodes[0])), C.int(len(nodes)), &ct[0], C.int(len(colTypes)), &lt[0], C.int(len(litTypes)), boolInt(compile),
		sb, C.size_t(srcCap), eb, C.size_t(errCap), &ot)
	if st != C.AH_OK {
		return "", 0, fmt.Errorf("%w: %s", arrow.ErrInvalid, C.GoString(eb))
	}
	return C.GoString(sb), arrow.Type(ot), nil
}
