//go:build hip

package arrowhip

import (
	"context"
	"fmt"
	"unsafe"

	"github.com/apache/arrow-go/v18/arrow"
	"github.com/apache/arrow-go/v18/arrow/compute"
	"github.com/apache/arrow-go/v18/arrow/compute/exec"
)

// ArithmeticOp values (kernels.ArithmeticOp is in an internal package: base_arithmetic.go:37-82).
const (
	opAdd int8 = 0
	opSub int8 = 1
	opMul int8 = 2
)

// stage uploads one host ArraySpan buffer to the device (PCIe-bound: ~47 GB/s; arrays that
// should stay resident across calls are better kept as DeviceBuffers and fed to the Context
// methods directly — arrow-go has no device-array type to carry them through CallFunction).
func (x *Context) stage(b []byte) (*DeviceBuffer, error) {
	d, err := x.Alloc(len(b) + 64)
	if err != nil {
		return nil, err
	}
	if err := d.Upload(b); err != nil {
		d.Free()
		return nil, err
	}
	return d, nil
}

// binaryExec builds an exec.ArrayKernelExec (exec/kernel.go:617) for op over one numeric type.
// With the scalar-kernel defaults (NullIntersection + MemPrealloc, kernel.go:660-661) the
// executor has already allocated+zeroed out.Buffers[1] and computed the validity, so the kernel
// only fills the values — exactly what ScalarBinary (kernels/helpers.go:193-236) does.
func binaryExec(x *Context, typ arrow.Type, width int, op int8) exec.ArrayKernelExec {
	return func(_ *exec.KernelCtx, batch *exec.ExecSpan, out *exec.ExecResult) error {
		if !batch.Values[0].IsArray() || !batch.Values[1].IsArray() {
			return fmt.Errorf("%w: arrowhip: scalar operands take the CPU kernel", arrow.ErrNotImplemented)
		}
		l, r := &batch.Values[0].Array, &batch.Values[1].Array
		n := int(out.Len)
		lb := l.Buffers[1].Buf[int(l.Offset)*width : (int(l.Offset)+n)*width]
		rb := r.Buffers[1].Buf[int(r.Offset)*width : (int(r.Offset)+n)*width]
		ob := out.Buffers[1].Buf[int(out.Offset)*width : (int(out.Offset)+n)*width]
		dl, err := x.stage(lb)
		if err != nil {
			return err
		}
		defer dl.Free()
		dr, err := x.stage(rb)
		if err != nil {
			return err
		}
		defer dr.Free()
		do, err := x.Alloc(len(ob) + 64)
		if err != nil {
			return err
		}
		defer do.Free()
		if err := x.ArithmeticBinary(typ, op, dl.Ptr, dr.Ptr, do.Ptr, int64(n)); err != nil {
			return err
		}
		return do.Download(ob)
	}
}

// Register installs the GPU kernels into a CHILD registry (registry.go:69-73) under new names
// ("add_unchecked_hip", …) and returns a context whose ExecCtx carries it (executor.go:110-112):
//
//	ctx, _ := arrowhip.Register(context.Background(), gpu)
//	out, _ := compute.CallFunction(ctx, "add_unchecked_hip", nil, a, b)
//
// New names keep arithmeticFunction.DispatchBest's numeric promotion for the stock functions
// (a plain ScalarFunction registered as "add" would lose it: arithmetic.go:112-142 vs
// functions.go:260-262).  The alternative — swapping ExecFn in place through
// funcImpl.Kernels() (functions.go:220-226) — is shown in INTEGRATION.md.
func Register(parent context.Context, x *Context) (context.Context, error) {
	reg := compute.NewChildRegistry(compute.GetFunctionRegistry())
	for _, f := range []struct {
		name string
		op   int8
	}{{"add_unchecked_hip", opAdd}, {"subtract_unchecked_hip", opSub}, {"multiply_unchecked_hip", opMul}} {
		fn := compute.NewScalarFunction(f.name, compute.Binary(), compute.FunctionDoc{Summary: "MI355X " + f.name})
		for _, t := range []struct {
			dt arrow.DataType
			w  int
		}{{arrow.PrimitiveTypes.Int64, 8}, {arrow.PrimitiveTypes.Uint64, 8}, {arrow.PrimitiveTypes.Float64, 8},
			{arrow.PrimitiveTypes.Int32, 4}, {arrow.PrimitiveTypes.Uint32, 4}, {arrow.PrimitiveTypes.Float32, 4}} {
			in := []exec.InputType{exec.NewExactInput(t.dt), exec.NewExactInput(t.dt)}
			if err := fn.AddNewKernel(in, exec.NewOutputType(t.dt), binaryExec(x, t.dt.ID(), t.w, f.op), nil); err != nil {
				return nil, err
			}
		}
		if !reg.AddFunction(fn, true) {
			return nil, fmt.Errorf("arrowhip: could not register %s", f.name)
		}
	}
	ectx := compute.DefaultExecCtx()
	ectx.Registry = reg
	return compute.SetExecCtx(parent, ectx), nil
}

var _ = unsafe.Pointer(nil)
