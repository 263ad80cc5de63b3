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
// every C call against include/arrowhip.h); the C half is exercised by tests/test_distributed_gpu.py with one, two and three ranks.
//
// Comm is ah_comm: this rank's communicator on the Context's compute stream (RCCL over xGMI, one process per GPU;
// SURVEY.md §8e).  Rank 0 makes the id with UniqueID and the launcher ships the 128 bytes to the other ranks.
// (ah_comm_init_transport, the flavour whose bytes travel through caller-supplied C callbacks, is for test rigs that put several
// ranks on one GPU; it is deliberately not bound here: a Go callback table would need //export trampolines and has no production
// use.)
type Comm struct {
	ctx *Context
	m   *C.ah_comm
}

func UniqueID() ([128]byte, error) {
	var id [128]byte
	if st := C.ah_comm_unique_id(unsafe.Pointer(&id[0])); st != C.AH_OK {
		return id, fmt.Errorf("arrowhip: ah_comm_unique_id failed (status %d): is librccl.so loadable?", int(st))
	}
	return id, nil
}

func (x *Context) NewComm(rank, world int, id [128]byte) (*Comm, error) {
	var m *C.ah_comm
	if err := x.err(C.ah_comm_init(x.c, C.int(rank), C.int(world), unsafe.Pointer(&id[0]), &m)); err != nil {
		return nil, err
	}
	return &Comm{ctx: x, m: m}, nil
}

func (c *Comm) Close() {
	if c.m != nil {
		C.ah_comm_destroy(c.m)
		c.m = nil
	}
}

func (c *Comm) Rank() int  { return int(C.ah_comm_rank(c.m)) }
func (c *Comm) World() int { return int(C.ah_comm_world(c.m)) }

// AllReduceSum: in place when send == recv (device pointers); typ = arrow.INT64 / UINT64 / FLOAT64 / INT32 / FLOAT32.
func (c *Comm) AllReduceSum(typ arrow.Type, send, recv unsafe.Pointer, count int64) error {
	return c.ctx.err(C.ah_comm_allreduce_sum(c.m, C.int(typ), send, recv, C.int64_t(count)))
}

func (c *Comm) AllGather(send, recv unsafe.Pointer, nbytesPerRank int64) error {
	return c.ctx.err(C.ah_comm_allgather(c.m, send, recv, C.int64_t(nbytesPerRank)))
}

// AllToAllV: the ragged exchange of group tuples for C5's key-hash-owner merge; sizes and offsets in bytes, one per rank.
func (c *Comm) AllToAllV(send unsafe.Pointer, sendBytes, sendOffs []int64, recv unsafe.Pointer, recvBytes, recvOffs []int64) error {
	w := c.World()
	if len(sendBytes) != w || len(sendOffs) != w || len(recvBytes) != w || len(recvOffs) != w {
		return fmt.Errorf("%w: arrowhip: AllToAllV wants %d sizes and offsets", arrow.ErrInvalid, w)
	}
	sb := (*C.int64_t)(unsafe.Pointer(&sendBytes[0]))
	so := (*C.int64_t)(unsafe.Pointer(&sendOffs[0]))
	rb := (*C.int64_t)(unsafe.Pointer(&recvBytes[0]))
	ro := (*C.int64_t)(unsafe.Pointer(&recvOffs[0]))
	return c.ctx.err(C.ah_comm_alltoallv(c.m, send, sb, so, recv, rb, ro))
}

// CmpFilterSumInt64: config C4 over this rank's shard (device pointers) → the global (sum, count) on every rank.
func (c *Comm) CmpFilterSumInt64(cmpop int, x, valid unsafe.Pointer, off, nLocal, threshold int64) (sum, count int64, err error) {
	var s, n C.int64_t
	err = c.ctx.err(C.ah_comm_cmp_filter_sum_i64(c.m, C.int(cmpop), (*C.int64_t)(x), (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(nLocal), C.int64_t(threshold), &s, &n))
	return int64(s), int64(n), err
}

// CmpFilterSumFloat64: the Float64 flavour — every rank's un-rounded accumulator is gathered and merged in rank order, rounded
// once: the same bytes on every rank; ±inf / NaN / overflow as math.Float64.Sum over the undivided column would give.
func (c *Comm) CmpFilterSumFloat64(cmpop int, x, valid unsafe.Pointer, off, nLocal int64, threshold float64) (sum float64, count int64, err error) {
	var s C.double
	var n C.int64_t
	err = c.ctx.err(C.ah_comm_cmp_filter_sum_f64(c.m, C.int(cmpop), (*C.double)(x), (*C.uint8_t)(valid), C.int64_t(off), C.int64_t(nLocal), C.double(threshold), &s, &n))
	return float64(s), int64(n), err
}

// MergeGroups: config C5 — this rank's local aggregate (outputs of HashSumFloat64 / HashSumInt64, nullGroupLocal = the null group
// they reported, -1 for none) → all groups in global first-seen order on every rank; nullGroup = the merged null group's position
// (-1: no rank had one).
func (c *Comm) MergeGroups(isF64 bool, keys, sums, counts, firstRows unsafe.Pointer, nLocal int64, nullGroupLocal int32, rowOffset, capacity int64,
	outKeys, outSums, outCounts, outFirstRows unsafe.Pointer) (ngroups int64, nullGroup int32, err error) {
	var g C.int64_t
	var ng C.int32_t
	err = c.ctx.err(C.ah_comm_merge_groups(c.m, boolInt(isF64), (*C.uint64_t)(keys), sums, (*C.int64_t)(counts), (*C.int64_t)(firstRows), C.int64_t(nLocal), C.int32_t(nullGroupLocal),
		C.int64_t(rowOffset), C.int64_t(capacity), (*C.uint64_t)(outKeys), outSums, (*C.int64_t)(outCounts), (*C.int64_t)(outFirstRows), &g, &ng))
	return int64(g), int32(ng), err
}
