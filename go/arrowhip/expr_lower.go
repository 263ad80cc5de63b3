//go:build hip

package arrowhip

/*
#include "arrowhip.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/apache/arrow-go/v18/arrow"
)

// EXPERIMENTAL — never compiled (no Go toolchain in the build image; tests/test_go_shim_static.py checks the declarations and
// every C name against include/arrowhip.h).  The C++ twin of this lowering (arrow_go_amd/host/expression.cc Flatten, fed by
// substrait.cc) is what tests/test_expressions.py exercises, with the cases of exprs/exec_test.go.
//
// ExprTree is the scalar-expression shape exprs.ExecuteScalarExpression recurses over (arrow/compute/exprs/exec.go:542-700) after
// the Substrait ↔ Arrow name mapping (exprs/types.go:196-272): a call of an Arrow compute function ("add", "add_unchecked",
// "greater", "and", …), a root field reference, a primitive literal, a cast.  An adapter from substrait-go's expr.Expression fills
// it: *expr.ScalarFunction → Call (name through the extension set's DecodeFunction), *expr.FieldReference → Field,
// expr.Literal → Lit, *expr.Cast → Cast.  (compute.Expression itself, deprecated in the reference, keeps funcName and args
// unexported, expression.go:278-285 — it cannot be walked from outside its package.)
type ExprKind int

const (
	ExprCall ExprKind = iota
	ExprField
	ExprLit
	ExprCast
)

type ExprTree struct {
	Kind     ExprKind
	Func     string      // ExprCall: Arrow function name
	Args     []*ExprTree // ExprCall; ExprCast: the one input
	Field    int         // ExprField: position in the input batch
	LitBits  uint64      // ExprLit: little-endian payload
	LitType  arrow.Type
	LitValid bool
	CastTo   arrow.Type // ExprCast
}

// exprOps: the functions the generated kernel covers (csrc/ah_expr.hip) — anything else makes LowerExpr refuse and the caller
// evaluates the tree node by node through CallFunction, which gives the same bytes.
var exprOps = map[string]int32{
	"add": C.AH_X_ADD_CHECKED, "add_unchecked": C.AH_X_ADD, "subtract": C.AH_X_SUB_CHECKED, "subtract_unchecked": C.AH_X_SUB, "sub": C.AH_X_SUB_CHECKED, "sub_unchecked": C.AH_X_SUB,
	"multiply": C.AH_X_MUL_CHECKED, "multiply_unchecked": C.AH_X_MUL, "negate_unchecked": C.AH_X_NEGATE, "abs_unchecked": C.AH_X_ABS,
	"sign": C.AH_X_SIGN, "equal": C.AH_X_EQ, "not_equal": C.AH_X_NE, "greater": C.AH_X_GT, "greater_equal": C.AH_X_GE,
	"less": C.AH_X_LT, "less_equal": C.AH_X_LE, "and": C.AH_X_AND, "or": C.AH_X_OR, "xor": C.AH_X_XOR, "and_not": C.AH_X_AND_NOT,
	"invert": C.AH_X_INVERT,
}

// Lowered is a postfix program and what it reads: Cols[i] is the batch column behind program column i.
type Lowered struct {
	Nodes    []ExprNode
	Cols     []int
	LitBits  []uint64
	LitTypes []arrow.Type
	LitValid []bool
}

// LowerExpr flattens t into the postfix program of ah_expr_compile.  Operand types of a call must already agree (the caller
// inserts ExprCast nodes where DispatchBest would cast, arithmetic.go:112-142); a cast the kernel cannot do without a check is
// refused by ah_expr_compile with ErrNotImplemented.
func LowerExpr(t *ExprTree) (*Lowered, error) {
	l := &Lowered{}
	if err := l.walk(t); err != nil {
		return nil, err
	}
	return l, nil
}

func (l *Lowered) walk(t *ExprTree) error {
	if t == nil {
		return fmt.Errorf("%w: nil expression", arrow.ErrInvalid) // exec.go:441-443
	}
	switch t.Kind {
	case ExprLit:
		l.Nodes = append(l.Nodes, ExprNode{Op: C.AH_X_LITERAL, Arg: int32(len(l.LitBits))})
		l.LitBits = append(l.LitBits, t.LitBits)
		l.LitTypes = append(l.LitTypes, t.LitType)
		l.LitValid = append(l.LitValid, t.LitValid)
	case ExprField:
		pos := -1
		for i, c := range l.Cols {
			if c == t.Field {
				pos = i
			}
		}
		if pos < 0 {
			pos = len(l.Cols)
			l.Cols = append(l.Cols, t.Field)
		}
		l.Nodes = append(l.Nodes, ExprNode{Op: C.AH_X_FIELD, Arg: int32(pos)})
	case ExprCast:
		if len(t.Args) != 1 {
			return fmt.Errorf("%w: cast without argument to cast", arrow.ErrInvalid) // exec.go:557-559
		}
		if err := l.walk(t.Args[0]); err != nil {
			return err
		}
		l.Nodes = append(l.Nodes, ExprNode{Op: C.AH_X_CAST, Arg: int32(t.CastTo)})
	case ExprCall:
		op, ok := exprOps[t.Func]
		if !ok {
			return fmt.Errorf("%w: %s is not covered by the fused kernel", arrow.ErrNotImplemented, t.Func)
		}
		for _, a := range t.Args {
			if err := l.walk(a); err != nil {
				return err
			}
		}
		l.Nodes = append(l.Nodes, ExprNode{Op: op})
	default:
		return fmt.Errorf("%w: expression kind %d", arrow.ErrNotImplemented, int(t.Kind))
	}
	return nil
}

// Run compiles (cached per context) and executes the program over device-resident columns: colValues / colValid / colOffsets are
// indexed like the BATCH; outValid may be nil when no input can be null.
func (l *Lowered) Run(x *Context, batchTypes []arrow.Type, colValues, colValid []unsafe.Pointer, colOffsets []int64, n int64, outValues, outValid unsafe.Pointer) (arrow.Type, error) {
	ct := make([]arrow.Type, len(l.Cols))
	cv := make([]unsafe.Pointer, len(l.Cols))
	cm := make([]unsafe.Pointer, len(l.Cols))
	co := make([]int64, len(l.Cols))
	for i, c := range l.Cols {
		if c < 0 || c >= len(batchTypes) {
			return 0, fmt.Errorf("%w: field reference %d of a batch of %d columns", arrow.ErrInvalid, c, len(batchTypes)) // exec.go:512-514
		}
		ct[i], cv[i], cm[i], co[i] = batchTypes[c], colValues[c], colValid[c], colOffsets[c]
	}
	e, err := x.ExprCompile(l.Nodes, ct, l.LitTypes)
	if err != nil {
		return 0, err
	}
	return e.OutType, e.Execute(cv, cm, co, l.LitBits, l.LitValid, n, outValues, outValid)
}
