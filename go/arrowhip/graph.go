//go:build hip

package arrowhip

/*
#include "arrowhip.h"
*/
import "C"

// EXPERIMENTAL — never compiled (no Go toolchain in the build image); the C half is exercised by
// tests/test_gpu_parity.py::test_graph_capture_replays_a_chain.
//
// Graph is ah_graph: a recorded sequence of kernel calls on device-resident buffers, replayed with one submission.  It is for the
// executor's small-batch case (ExecCtx.ChunkSize-sized spans, compute/executor.go:658-702): a chain of cheap kernels over a
// 64 Ki-row span spends more time between launches than in them.  Record runs fn with the context in capture mode — fn may
// only use calls that leave their result on the device (see include/arrowhip.h, "hipGraph capture") — after one eager fn()
// by the caller, so that the scratch arenas have their size.
type Graph struct {
	ctx *Context
	g   *C.ah_graph
}

func (x *Context) Record(fn func() error) (*Graph, error) {
	if err := x.err(C.ah_graph_begin(x.c)); err != nil {
		return nil, err
	}
	ferr := fn()
	var g *C.ah_graph
	err := x.err(C.ah_graph_end(x.c, &g))
	if ferr != nil {
		if g != nil {
			C.ah_graph_destroy(g)
		}
		return nil, ferr
	}
	if err != nil {
		return nil, err
	}
	return &Graph{ctx: x, g: g}, nil
}

// Launch replays the sequence on the context's compute stream (same buffers, their contents of the moment).
func (g *Graph) Launch() error { return g.ctx.err(C.ah_graph_launch(g.ctx.c, g.g)) }

func (g *Graph) Close() {
	if g.g != nil {
		C.ah_graph_destroy(g.g)
		g.g = nil
	}
}
