//go:build hip

package arrowhip

/*
#include "arrowhip.h"
*/
import "C"

import "unsafe"

// PinnedAllocator implements memory.Allocator (arrow/memory/allocator.go:23-27) on
// hipHostMalloc'ed memory, so Arrow buffers built by ordinary arrow-go builders can be DMA'd to
// HBM by ah_upload_async without a staging copy.  Same contract as memory/mallocator: 64-byte
// aligned (hipHostMalloc is page aligned), zero-filled.
type PinnedAllocator struct{ Ctx *Context }

func (p *PinnedAllocator) Allocate(size int) []byte {
	if size == 0 {
		return []byte{}
	}
	var h unsafe.Pointer
	if st := C.ah_host_alloc_pinned(p.Ctx.c, C.size_t(size), &h); st != C.AH_OK {
		panic("arrowhip: pinned allocation failed")
	}
	b := unsafe.Slice((*byte)(h), size)
	for i := range b {
		b[i] = 0
	}
	return b
}

func (p *PinnedAllocator) Reallocate(size int, b []byte) []byte {
	if size == len(b) {
		return b
	}
	nb := p.Allocate(size)
	copy(nb, b)
	p.Free(b)
	return nb
}

func (p *PinnedAllocator) Free(b []byte) {
	if cap(b) == 0 {
		return
	}
	C.ah_host_free_pinned(p.Ctx.c, unsafe.Pointer(&b[:1][0]))
}
