"""Thin host-side handles over the C ABI: a context, device buffers, and one Python
method per leaf entry point taking DeviceBuffer / raw device pointers.

Harness-level plumbing used by tests/, bench.py and the array-level layer in
arrow_go_amd.compute; it adds no semantics of its own.  Everything executes in
libarrowhip.so on the GPU — there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _native as N
from ._native import lib, check


def _ptr(x) -> Optional[int]:
    """device pointer of a DeviceBuffer / DevicePtr / int / None"""
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    return int(x)


class DeviceBuffer:
    """A device allocation owned by a Context (ah_buf_alloc / ah_buf_free)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(ctx.handle, lib.ah_buf_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value or 0

    def at(self, byte_offset: int) -> int:
        return self.ptr + int(byte_offset)

    def upload(self, arr: np.ndarray, byte_offset: int = 0) -> "DeviceBuffer":
        a = np.ascontiguousarray(arr)
        assert byte_offset + a.nbytes <= max(self.nbytes, 1), "upload out of range"
        if a.nbytes:
            check(self.ctx.handle, lib.ah_upload_async(self.ctx.handle, self.ptr + byte_offset, a.ctypes.data, a.nbytes))
            self.ctx.sync()  # `a` may be a temporary
        return self

    def download(self, dtype, count: int, byte_offset: int = 0) -> np.ndarray:
        out = np.empty(int(count), dtype=dtype)
        if out.nbytes:
            check(self.ctx.handle, lib.ah_download_async(self.ctx.handle, out.ctypes.data, self.ptr + byte_offset, out.nbytes))
            self.ctx.sync()
        return out

    def memset(self, byte_value: int, nbytes: Optional[int] = None, byte_offset: int = 0) -> None:
        check(self.ctx.handle, lib.ah_memset_async(self.ctx.handle, self.ptr + byte_offset, byte_value,
                                                   self.nbytes if nbytes is None else nbytes))

    def free(self) -> None:
        if self.ptr:
            lib.ah_buf_free(self.ctx.handle, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                self.free()
        except Exception:
            pass


class PinnedBuffer:
    """Pinned host memory from ah_host_alloc_pinned (what a Go memory.Allocator backed by go/arrowhip/allocator.go hands out),
    seen as a numpy array: the source / destination of overlapped transfers."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(ctx.handle, lib.ah_host_alloc_pinned(ctx.handle, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p.value or 0
        self._raw = (C.c_uint8 * max(self.nbytes, 1)).from_address(self.ptr)

    def view(self, dtype, count: Optional[int] = None, byte_offset: int = 0) -> np.ndarray:
        dt = np.dtype(dtype)
        n = (self.nbytes - byte_offset) // dt.itemsize if count is None else int(count)
        return np.frombuffer(self._raw, dtype=dt, count=n, offset=byte_offset)

    def free(self) -> None:
        if self.ptr:
            self._raw = None
            lib.ah_host_free_pinned(self.ctx.handle, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                self.free()
        except Exception:
            pass


def _hptr(x) -> Optional[int]:
    """host pointer of a numpy array / PinnedBuffer / int / None"""
    if x is None:
        return None
    if isinstance(x, PinnedBuffer):
        return x.ptr
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data
    return int(x)


class Ingest:
    """ah_ingest: chunked host → HBM pipeline with per-slot events (include/arrowhip.h, "overlapped host -> HBM ingest").
    Host arguments are numpy arrays (ideally views of a PinnedBuffer), PinnedBuffers or raw host addresses."""

    def __init__(self, ctx: "Context", chunk_bytes: int = 0, depth: int = 0):
        h = C.c_void_p()
        check(ctx.handle, lib.ah_ingest_create(ctx.handle, chunk_bytes, depth, C.byref(h)))
        self.ctx, self.handle = ctx, h

    def close(self) -> None:
        if self.handle:
            lib.ah_ingest_destroy(self.handle)
            self.handle = None

    def sum_float64(self, host, n: int) -> float:
        r = C.c_double()
        check(self.ctx.handle, lib.ah_ingest_sum_float64(self.handle, _hptr(host), n, C.byref(r)))
        return r.value

    def sum_int64(self, host, n: int) -> int:
        r = C.c_int64()
        check(self.ctx.handle, lib.ah_ingest_sum_int64(self.handle, _hptr(host), n, C.byref(r)))
        return r.value

    def arithmetic_binary(self, type_id: int, op: int, l_host, r_host, out_host, n: int) -> None:
        check(self.ctx.handle, lib.ah_ingest_arithmetic_binary(self.handle, type_id, op, _hptr(l_host), _hptr(r_host), _hptr(out_host), n))

    def filter_count(self, fdata_host, fvalid_host, foff: int, n: int, null_sel: int) -> int:
        r = C.c_int64()
        check(self.ctx.handle, lib.ah_ingest_filter_count(self.handle, _hptr(fdata_host), _hptr(fvalid_host), foff, n, null_sel, C.byref(r)))
        return r.value

    def filter_primitive(self, byte_width: int, values_host, vvalid_host, voff: int, n: int, n_out: int, out_values_host, out_valid_host) -> int:
        r = C.c_int64()
        check(self.ctx.handle, lib.ah_ingest_filter_primitive(self.handle, byte_width, _hptr(values_host), _hptr(vvalid_host), voff, n, n_out,
                                                              _hptr(out_values_host), _hptr(out_valid_host), C.byref(r)))
        return r.value


class Graph:
    """ah_graph: a captured sequence of calls, replayed with one submission (include/arrowhip.h, "hipGraph capture")"""

    def __init__(self, ctx: "Context", handle):
        self.ctx, self.handle = ctx, handle

    def launch(self) -> None:
        check(self.ctx.handle, lib.ah_graph_launch(self.ctx.handle, self.handle))

    def close(self) -> None:
        if self.handle:
            lib.ah_graph_destroy(self.handle)
            self.handle = None


class Comm:
    """ah_comm: this rank's RCCL communicator on an ah_ctx's compute stream (include/arrowhip.h, "multi-GPU exchange").
    Buffers are device pointers (ints, DeviceBuffers, or anything with .data_ptr())."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        st = lib.ah_comm_unique_id(C.addressof(buf))
        if st != N.AH_OK:
            raise N.ErrHip("ah_comm_unique_id failed: is librccl.so loadable?")
        return bytes(buf)

    def __init__(self, ctx: "Context", rank: int, world: int, unique_id: bytes):
        assert len(unique_id) == 128
        h = C.c_void_p()
        idb = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(ctx.handle, lib.ah_comm_init(ctx.handle, rank, world, C.addressof(idb), C.byref(h)))
        self.ctx, self.handle, self.rank, self.world = ctx, h, rank, world

    @classmethod
    def from_transport(cls, ctx: "Context", rank: int, world: int, transport) -> "Comm":
        """a communicator over a host-supplied transport (ah_comm_init_transport): `transport` has a `.struct` (AhTransport)
        whose callbacks stay alive with it — e.g. arrow_go_amd.distributed.GlooTransport"""
        self = cls.__new__(cls)
        h = C.c_void_p()
        check(ctx.handle, lib.ah_comm_init_transport(ctx.handle, rank, world, C.byref(transport.struct), C.byref(h)))
        self.ctx, self.handle, self.rank, self.world, self._transport = ctx, h, rank, world, transport
        return self

    def close(self) -> None:
        if self.handle:
            lib.ah_comm_destroy(self.handle)
            self.handle = None

    # ---- configs C4 / C5 as single calls -------------------------------------------------------------------------------
    def cmp_filter_sum_i64(self, cmpop: int, x, valid, off: int, n_local: int, threshold: int):
        s, n = C.c_int64(), C.c_int64()
        check(self.ctx.handle, lib.ah_comm_cmp_filter_sum_i64(self.handle, cmpop, _ptr(x), _ptr(valid), off, n_local, threshold, C.byref(s), C.byref(n)))
        return s.value, n.value

    def cmp_filter_sum_f64(self, cmpop: int, x, valid, off: int, n_local: int, threshold: float):
        s, n = C.c_double(), C.c_int64()
        check(self.ctx.handle, lib.ah_comm_cmp_filter_sum_f64(self.handle, cmpop, _ptr(x), _ptr(valid), off, n_local, threshold, C.byref(s), C.byref(n)))
        return s.value, n.value

    def merge_groups(self, is_f64: bool, keys, sums, counts, first_rows, ngroups_local: int, row_offset: int, capacity: int,
                     out_keys, out_sums, out_counts, out_first_rows, null_group_local: int = -1, with_null_group: bool = False):
        """→ the global group count; with_null_group: → (count, position of the merged null group or -1).  null_group_local is what
        hash_sum reported for this rank's shard."""
        g = C.c_int64()
        ng = C.c_int32(-1)
        try:
            check(self.ctx.handle, lib.ah_comm_merge_groups(self.handle, int(is_f64), _ptr(keys), _ptr(sums), _ptr(counts), _ptr(first_rows),
                                                            ngroups_local, int(null_group_local), row_offset, capacity, _ptr(out_keys), _ptr(out_sums),
                                                            _ptr(out_counts), _ptr(out_first_rows), C.byref(g), C.byref(ng)))
        finally:
            self.last_ngroups = g.value      # set even when the capacity was too small
        return (g.value, ng.value) if with_null_group else g.value

    def allreduce_sum(self, type_id: int, send, recv, count: int) -> None:
        check(self.ctx.handle, lib.ah_comm_allreduce_sum(self.handle, type_id, _ptr(send), _ptr(recv), count))

    def allgather(self, send, recv, nbytes_per_rank: int) -> None:
        check(self.ctx.handle, lib.ah_comm_allgather(self.handle, _ptr(send), _ptr(recv), nbytes_per_rank))

    def alltoallv(self, send, send_bytes, send_offs, recv, recv_bytes, recv_offs) -> None:
        arr = lambda v: (C.c_int64 * self.world)(*[int(x) for x in v])
        check(self.ctx.handle, lib.ah_comm_alltoallv(self.handle, _ptr(send), arr(send_bytes), arr(send_offs), _ptr(recv), arr(recv_bytes), arr(recv_offs)))


class Context:
    """ah_ctx: one GPU, a compute stream and a copy stream."""

    def __init__(self, device_id: int = 0, stream: Optional[int] = None):
        h = C.c_void_p()
        if stream is None:
            st = lib.ah_ctx_create(device_id, C.byref(h))
        else:
            st = lib.ah_ctx_create_on_stream(device_id, C.c_void_p(stream), C.byref(h))
        if st != N.AH_OK:
            raise N.ErrHip(f"ah_ctx_create(device={device_id}) failed with status {st}: is a GPU visible?")
        self.handle = h
        self.device_id = device_id

    # ---- lifecycle / memory ---------------------------------------------------------
    def close(self) -> None:
        if self.handle:
            lib.ah_ctx_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self) -> None:
        check(self.handle, lib.ah_sync(self.handle))

    def set_option(self, name: str, value: int) -> None:
        """measurement / test switch of this context (ah_ctx_set_option); never changes a result"""
        check(self.handle, lib.ah_ctx_set_option(self.handle, name.encode(), int(value)))

    # ---- hipGraph capture -------------------------------------------------------------------------------------------------
    def graph_begin(self) -> None:
        check(self.handle, lib.ah_graph_begin(self.handle))

    def graph_end(self) -> "Graph":
        h = C.c_void_p()
        check(self.handle, lib.ah_graph_end(self.handle, C.byref(h)))
        return Graph(self, h)

    def alloc_pinned(self, nbytes: int) -> PinnedBuffer:
        return PinnedBuffer(self, nbytes)

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr: np.ndarray, pad: int = 0) -> DeviceBuffer:
        a = np.ascontiguousarray(arr)
        return DeviceBuffer(self, a.nbytes + pad).upload(a)

    def timer_start(self) -> None:
        check(self.handle, lib.ah_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        check(self.handle, lib.ah_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def event_record(self, slot: int) -> None:
        check(self.handle, lib.ah_event_record(self.handle, slot))

    def event_elapsed_ms(self, slot_a: int, slot_b: int) -> float:
        ms = C.c_float()
        check(self.handle, lib.ah_event_elapsed_ms(self.handle, slot_a, slot_b, C.byref(ms)))
        return ms.value

    # ---- arrow/math Sum -----------------------------------------------------------------
    def sum_float64(self, buf, n: int) -> float:
        r = C.c_double()
        check(self.handle, lib.ah_sum_float64(self.handle, _ptr(buf), n, C.byref(r)))
        return r.value

    def sum_int64(self, buf, n: int) -> int:
        r = C.c_int64()
        check(self.handle, lib.ah_sum_int64(self.handle, _ptr(buf), n, C.byref(r)))
        return r.value

    def sum_uint64(self, buf, n: int) -> int:
        r = C.c_uint64()
        check(self.handle, lib.ah_sum_uint64(self.handle, _ptr(buf), n, C.byref(r)))
        return r.value

    def sum_float64_dev(self, buf, n: int, res_dev) -> None:
        check(self.handle, lib.ah_sum_float64_dev(self.handle, _ptr(buf), n, _ptr(res_dev)))

    def sum_int64_dev(self, buf, n: int, res_dev) -> None:
        check(self.handle, lib.ah_sum_int64_dev(self.handle, _ptr(buf), n, _ptr(res_dev)))

    # ---- arithmetic ---------------------------------------------------------------------
    def arithmetic(self, type_id: int, op: int, shape: int, l, r, out, n: int) -> None:
        """l / r: device pointer, or (for the scalar side of AS / SA) a numpy scalar array."""
        if shape == N.SHAPE_AA:
            st = lib.ah_arithmetic_binary(self.handle, type_id, op, _ptr(l), _ptr(r), _ptr(out), n)
        elif shape == N.SHAPE_AS:
            s = np.ascontiguousarray(r)
            st = lib.ah_arithmetic_arr_scalar(self.handle, type_id, op, _ptr(l), s.ctypes.data, _ptr(out), n)
        else:
            s = np.ascontiguousarray(l)
            st = lib.ah_arithmetic_scalar_arr(self.handle, type_id, op, s.ctypes.data, _ptr(r), _ptr(out), n)
        check(self.handle, st)

    def arithmetic_unary(self, type_id: int, op: int, inp, out, n: int) -> None:
        check(self.handle, lib.ah_arithmetic_unary(self.handle, type_id, op, _ptr(inp), _ptr(out), n))

    def arithmetic_checked(self, type_id: int, op: int, shape: int, l, lvalid, loff, r, rvalid, roff, scalar_valid,
                           out, n: int) -> None:
        keep = []

        def side(x, is_scalar):
            if is_scalar:
                s = np.ascontiguousarray(x)
                keep.append(s)
                return s.ctypes.data
            return _ptr(x)

        st = lib.ah_arithmetic_checked(self.handle, type_id, op, shape, side(l, shape == N.SHAPE_SA), _ptr(lvalid), loff,
                                       side(r, shape == N.SHAPE_AS), _ptr(rvalid), roff, int(scalar_valid), _ptr(out), n)
        check(self.handle, st)

    def arithmetic_ext(self, type_id: int, op: int, shape: int, l, lvalid, loff, r, rvalid, roff, scalar_valid, out, n: int) -> None:
        """divide / abs / negate (checked) / bit-wise / shifts / sqrt — see ah_arithmetic_ext; unary ops: r = None"""
        keep = []

        def side(x, is_scalar):
            if x is None:
                return None
            if is_scalar:
                s = np.ascontiguousarray(x)
                keep.append(s)
                return s.ctypes.data
            return _ptr(x)

        st = lib.ah_arithmetic_ext(self.handle, type_id, op, shape, side(l, shape == N.SHAPE_SA), _ptr(lvalid), loff,
                                   side(r, shape == N.SHAPE_AS), _ptr(rvalid), roff, int(scalar_valid), _ptr(out), n)
        check(self.handle, st)

    def round(self, type_id: int, values, valid, off: int, n: int, ndigits: int, mode: int, multiple, pow10: float, out) -> None:
        """round (multiple=None; pow10 = math.Pow10(|ndigits|)) / round_to_multiple (multiple: numpy scalar array of the type)"""
        m = None if multiple is None else np.ascontiguousarray(multiple)
        check(self.handle, lib.ah_round(self.handle, type_id, _ptr(values), _ptr(valid), off, n, ndigits, mode,
                                        None if m is None else m.ctypes.data, float(pow10), _ptr(out)))

    # ---- compare ------------------------------------------------------------------------
    def comparison(self, cmpop: int, shape: int, type_id: int, l, r, out_bits, n: int, out_bit_offset: int = 0) -> None:
        keep = []

        def side(x, is_scalar):
            if is_scalar:
                s = np.ascontiguousarray(x)
                keep.append(s)
                return s.ctypes.data
            return _ptr(x)

        st = lib.ah_comparison(self.handle, cmpop, shape, type_id, side(l, shape == N.SHAPE_SA),
                               side(r, shape == N.SHAPE_AS), _ptr(out_bits), n, out_bit_offset)
        check(self.handle, st)

    # ---- bitmaps ------------------------------------------------------------------------
    def bitmap_op(self, op: int, l, loff: int, r, roff: int, out, ooff: int, nbits: int) -> None:
        check(self.handle, lib.ah_bitmap_op(self.handle, op, _ptr(l), loff, _ptr(r), roff, _ptr(out), ooff, nbits))

    def count_set_bits(self, bits, off: int, nbits: int) -> int:
        r = C.c_int64()
        check(self.handle, lib.ah_count_set_bits(self.handle, _ptr(bits), off, nbits, C.byref(r)))
        return r.value

    def copy_bitmap(self, src, soff: int, nbits: int, dst, doff: int, invert: bool = False) -> None:
        check(self.handle, lib.ah_copy_bitmap(self.handle, _ptr(src), soff, nbits, _ptr(dst), doff, int(invert)))

    def set_bits_to(self, bits, off: int, nbits: int, value: bool) -> None:
        check(self.handle, lib.ah_set_bits_to(self.handle, _ptr(bits), off, nbits, int(value)))

    def kleene(self, op: int, lvalid, ldata, loff: int, rvalid, rdata, roff: int, ovalid, odata, ooff: int, nbits: int) -> None:
        check(self.handle, lib.ah_kleene(self.handle, op, _ptr(lvalid), _ptr(ldata), loff, _ptr(rvalid), _ptr(rdata), roff,
                                         _ptr(ovalid), _ptr(odata), ooff, nbits))

    # ---- selection ----------------------------------------------------------------------
    def filter_count(self, fdata, fvalid, foff: int, n: int, null_sel: int) -> int:
        r = C.c_int64()
        check(self.handle, lib.ah_filter_count(self.handle, _ptr(fdata), _ptr(fvalid), foff, n, null_sel, C.byref(r)))
        return r.value

    def filter_primitive(self, byte_width: int, values, vvalid, voff: int, fdata, fvalid, foff: int, n: int, null_sel: int,
                         n_out: int, out_values, out_valid, want_null_count: bool = True) -> int:
        r = C.c_int64()
        check(self.handle, lib.ah_filter_primitive(self.handle, byte_width, _ptr(values), _ptr(vvalid), voff, _ptr(fdata),
                                                   _ptr(fvalid), foff, n, null_sel, n_out, _ptr(out_values), _ptr(out_valid),
                                                   C.byref(r) if want_null_count else None))
        return r.value

    def filter_primitive_dev(self, byte_width: int, values, vvalid, voff: int, fdata, fvalid, foff: int, n: int, null_sel: int,
                             out_values, out_valid, status_dev) -> None:
        """No count call, no host round trip: outputs sized for n rows, {selected, null count} left in `status_dev` (16 bytes)."""
        check(self.handle, lib.ah_filter_primitive_dev(self.handle, byte_width, _ptr(values), _ptr(vvalid), voff, _ptr(fdata),
                                                       _ptr(fvalid), foff, n, null_sel, _ptr(out_values), _ptr(out_valid), _ptr(status_dev)))

    def filter_primitive_once(self, byte_width: int, values, vvalid, voff: int, fdata, fvalid, foff: int, n: int, null_sel: int,
                              out_values, out_valid):
        """PrimitiveFilter in ONE call: outputs sized for n rows (the worst case), → (rows selected, output null count)."""
        k, nulls = C.c_int64(), C.c_int64()
        check(self.handle, lib.ah_filter_primitive_once(self.handle, byte_width, _ptr(values), _ptr(vvalid), voff, _ptr(fdata),
                                                        _ptr(fvalid), foff, n, null_sel, _ptr(out_values), _ptr(out_valid),
                                                        C.byref(k), C.byref(nulls)))
        return k.value, nulls.value

    def take_primitive_dev(self, byte_width: int, values, vvalid, voff: int, nvalues: int, idx_byte_width: int, idx_signed: bool,
                           idx, ivalid, ioff: int, nidx: int, out_values, out_valid, status_dev) -> None:
        """No host round trip: {position of the first out-of-range index or 2^64 − 1, null count} left in `status_dev` (16 bytes)."""
        check(self.handle, lib.ah_take_primitive_dev(self.handle, byte_width, _ptr(values), _ptr(vvalid), voff, nvalues,
                                                     idx_byte_width, int(idx_signed), _ptr(idx), _ptr(ivalid), ioff, nidx,
                                                     _ptr(out_values), _ptr(out_valid), _ptr(status_dev)))

    def filter_to_indices(self, fdata, fvalid, foff: int, n: int, null_sel: int, n_out: int, out_idx, out_valid) -> int:
        r = C.c_int64()
        check(self.handle, lib.ah_filter_to_indices(self.handle, _ptr(fdata), _ptr(fvalid), foff, n, null_sel, n_out,
                                                    _ptr(out_idx), _ptr(out_valid), C.byref(r)))
        return r.value

    def take_primitive(self, byte_width: int, values, vvalid, voff: int, nvalues: int, idx_byte_width: int, idx_signed: bool,
                       idx, ivalid, ioff: int, nidx: int, bounds_check: bool, out_values, out_valid) -> int:
        r = C.c_int64()
        bad = C.c_int64()
        check(self.handle, lib.ah_take_primitive(self.handle, byte_width, _ptr(values), _ptr(vvalid), voff, nvalues,
                                                 idx_byte_width, int(idx_signed), _ptr(idx), _ptr(ivalid), ioff, nidx,
                                                 int(bounds_check), _ptr(out_values), _ptr(out_valid), C.byref(r), C.byref(bad)))
        return r.value

    # ---- cumulative_sum -----------------------------------------------------------------
    def cumulative_sum(self, type_id: int, values, valid, off: int, n: int, start: Optional[bytes], skip_nulls: bool, checked: bool,
                       out_values, out_valid) -> int:
        """start: little-endian payload of one element of the type, or None (= 0).  Returns the output null count."""
        nulls = C.c_int64()
        sbuf = None
        if start is not None:
            sbuf = (C.c_uint8 * 8)()
            C.memmove(sbuf, start, len(start))
        check(self.handle, lib.ah_cumulative_sum(self.handle, type_id, _ptr(values), _ptr(valid), off, n,
                                                 C.addressof(sbuf) if sbuf is not None else None, int(skip_nulls), int(checked),
                                                 _ptr(out_values), _ptr(out_valid),
                                                 C.byref(nulls) if out_valid is not None else None))  # no validity → no sync
        return nulls.value

    # ---- cast ---------------------------------------------------------------------------
    def cast_numeric(self, in_type: int, out_type: int, values, valid, off: int, n: int, allow_int_overflow: bool,
                     allow_float_truncate: bool, out_values) -> None:
        check(self.handle, lib.ah_cast_numeric(self.handle, in_type, out_type, _ptr(values), _ptr(valid), off, n,
                                               int(allow_int_overflow), int(allow_float_truncate), _ptr(out_values)))

    def shift_time(self, in_bits: int, out_bits: int, op: int, factor: int, checked: bool, values, valid, off: int, n: int, out_values) -> None:
        """ShiftTime (cast_temporal.go:35-104); op 0 multiply, 1 divide.  A failed check raises ErrInvalid carrying .bad_value"""
        bad = C.c_int64(0)
        try:
            check(self.handle, lib.ah_shift_time(self.handle, in_bits, out_bits, op, factor, int(checked), _ptr(values), _ptr(valid), off, n,
                                                        _ptr(out_values), C.byref(bad)))
        except N.ErrInvalid as e:
            e.bad_value = bad.value
            raise

    def cast_bool_to_numeric(self, out_type: int, bits, off: int, n: int, out_values) -> None:
        check(self.handle, lib.ah_cast_bool_to_numeric(self.handle, out_type, _ptr(bits), off, n, _ptr(out_values)))

    # ---- set lookup ---------------------------------------------------------------------
    def is_in(self, byte_width: int, values, valid, off: int, n: int, set_values, set_valid, set_off: int, set_n: int,
              null_behavior: int, out_data, out_valid, out_bit_offset: int = 0) -> None:
        check(self.handle, lib.ah_is_in(self.handle, byte_width, _ptr(values), _ptr(valid), off, n, _ptr(set_values), _ptr(set_valid),
                                        set_off, set_n, null_behavior, _ptr(out_data), _ptr(out_valid), out_bit_offset))

    # ---- min / max ----------------------------------------------------------------------
    def min_max(self, type_id: int, values, n: int, dtype):
        lo, hi = np.zeros(1, dtype), np.zeros(1, dtype)
        check(self.handle, lib.ah_min_max(self.handle, type_id, _ptr(values), n, lo.ctypes.data, hi.ctypes.data))
        return lo[0], hi[0]

    # ---- sort ---------------------------------------------------------------------------
    def sort_indices(self, type_id: int, values, valid, off: int, n: int, descending: bool, nulls_at_start: bool, out_indices) -> None:
        check(self.handle, lib.ah_sort_indices(self.handle, type_id, _ptr(values), _ptr(valid), off, n, int(descending), int(nulls_at_start),
                                               _ptr(out_indices)))

    def sort_indices_multi(self, keys, n: int, out_indices) -> None:
        """keys: [(type_id, values, valid, off, descending, nulls_at_start), …] — most significant first"""
        k = len(keys)
        types = (C.c_int * k)(*[x[0] for x in keys])
        vals = (C.c_void_p * k)(*[_ptr(x[1]) for x in keys])
        valids = (C.c_void_p * k)(*[_ptr(x[2]) for x in keys])
        offs = (C.c_int64 * k)(*[x[3] for x in keys])
        desc = (C.c_int * k)(*[int(x[4]) for x in keys])
        nfirst = (C.c_int * k)(*[int(x[5]) for x in keys])
        check(self.handle, lib.ah_sort_indices_multi(self.handle, k, types, vals, valids, offs, n, desc, nfirst, _ptr(out_indices)))

    def take_boolean(self, data, vvalid, voff: int, nvalues: int, idx_byte_width: int, idx_signed: bool, idx, ivalid, ioff: int, nidx: int,
                     out_data, out_valid) -> int:
        r, bad = C.c_int64(), C.c_int64()
        check(self.handle, lib.ah_take_boolean(self.handle, _ptr(data), _ptr(vvalid), voff, nvalues, idx_byte_width, int(idx_signed), _ptr(idx),
                                               _ptr(ivalid), ioff, nidx, 1, _ptr(out_data), _ptr(out_valid), C.byref(r), C.byref(bad)))
        return r.value

    # ---- var-length take ------------------------------------------------------------------
    def take_binary_offsets(self, offset_width: int, offsets, vvalid, voff: int, nvalues: int, idx_byte_width: int, idx_signed: bool,
                            idx, ivalid, ioff: int, nidx: int, out_offsets, out_valid):
        """→ (null_count, total_bytes)"""
        nulls, total, bad = C.c_int64(), C.c_int64(), C.c_int64()
        check(self.handle, lib.ah_take_binary_offsets(self.handle, offset_width, _ptr(offsets), _ptr(vvalid), voff, nvalues, idx_byte_width,
                                                      int(idx_signed), _ptr(idx), _ptr(ivalid), ioff, nidx, 1, _ptr(out_offsets),
                                                      _ptr(out_valid), C.byref(nulls), C.byref(total), C.byref(bad)))
        return nulls.value, total.value

    def take_binary_data(self, offset_width: int, offsets, data, voff: int, idx_byte_width: int, idx, nidx: int, out_offsets, out_data) -> None:
        check(self.handle, lib.ah_take_binary_data(self.handle, offset_width, _ptr(offsets), _ptr(data), voff, idx_byte_width, _ptr(idx), nidx,
                                                   _ptr(out_offsets), _ptr(out_data)))

    # ---- hashing ------------------------------------------------------------------------
    def hash_u64_encode(self, keys, valid, off: int, n: int, encode_nulls: bool, out_ids, out_ids_valid, out_dict):
        nd = C.c_int64()
        nid = C.c_int32()
        check(self.handle, lib.ah_hash_u64_encode(self.handle, _ptr(keys), _ptr(valid), off, n, int(encode_nulls), _ptr(out_ids),
                                                  _ptr(out_ids_valid), _ptr(out_dict), C.byref(nd), C.byref(nid)))
        return nd.value, nid.value

    def hash_binary_encode(self, offset_width: int, offsets, data, valid, off: int, n: int, encode_nulls: bool, out_ids, out_ids_valid,
                           out_first_rows):
        nd = C.c_int64()
        nid = C.c_int32()
        check(self.handle, lib.ah_hash_binary_encode(self.handle, offset_width, _ptr(offsets), _ptr(data), _ptr(valid), off, n,
                                                     int(encode_nulls), _ptr(out_ids), _ptr(out_ids_valid), _ptr(out_first_rows),
                                                     C.byref(nd), C.byref(nid)))
        return nd.value, nid.value

    def hash_fixed_encode(self, byte_width: int, data, valid, off: int, n: int, encode_nulls: bool, out_ids, out_ids_valid, out_first_rows, out_dict):
        nd = C.c_int64()
        nid = C.c_int32()
        check(self.handle, lib.ah_hash_fixed_encode(self.handle, byte_width, _ptr(data), _ptr(valid), off, n, int(encode_nulls), _ptr(out_ids),
                                                    _ptr(out_ids_valid), _ptr(out_first_rows), _ptr(out_dict), C.byref(nd), C.byref(nid)))
        return nd.value, nid.value

    def hash_sum(self, kind: str, keys, kvalid, koff: int, vals, vvalid, voff: int, n: int, out_keys, out_sums, out_counts,
                 out_first_rows=None):
        ng = C.c_int64()
        nid = C.c_int32()
        fn = lib.ah_hash_sum_f64 if kind == "f64" else lib.ah_hash_sum_i64
        check(self.handle, fn(self.handle, _ptr(keys), _ptr(kvalid), koff, _ptr(vals), _ptr(vvalid), voff, n, _ptr(out_keys),
                              _ptr(out_sums), _ptr(out_counts), _ptr(out_first_rows), C.byref(ng), C.byref(nid)))
        return ng.value, nid.value

    def hash_partition(self, keys, n: int, nparts: int, out_part) -> None:
        check(self.handle, lib.ah_hash_partition_u64(self.handle, _ptr(keys), n, nparts, _ptr(out_part)))

    # ---- fused expressions --------------------------------------------------------------
    def expr_compile(self, nodes, col_types, lit_types):
        """nodes: [(opcode, arg), …] postfix.  Returns (handle, out_type)."""
        na = _nodes_array(nodes)
        ct = (C.c_int * max(len(col_types), 1))(*col_types)
        lt = (C.c_int * max(len(lit_types), 1))(*lit_types)
        h = C.c_void_p()
        ot = C.c_int()
        check(self.handle, lib.ah_expr_compile(self.handle, na, len(nodes), ct, len(col_types), lt, len(lit_types), C.byref(h), C.byref(ot)))
        return h, ot.value

    def expr_execute(self, handle, col_values, col_valid, col_offsets, lit_values, lit_valid, n: int, out_values, out_valid) -> None:
        nc = max(len(col_values), 1)
        cv = (C.c_void_p * nc)(*[_ptr(x) for x in col_values])
        cvd = (C.c_void_p * nc)(*[_ptr(x) for x in col_valid])
        co = (C.c_int64 * nc)(*col_offsets)
        nl = max(len(lit_values), 1)
        lv = (C.c_uint8 * (8 * nl))()
        for i, raw in enumerate(lit_values):  # raw: bytes (≤ 8) little-endian payload
            C.memmove(C.addressof(lv) + 8 * i, raw, len(raw))
        lvd = (C.c_int * nl)(*[int(x) for x in lit_valid]) if lit_valid else (C.c_int * nl)()
        check(self.handle, lib.ah_expr_execute(self.handle, handle, cv, cvd, co, lv, lvd, n, _ptr(out_values), _ptr(out_valid)))

    def expr_source(self, handle) -> str:
        return lib.ah_expr_source(handle).decode()

    # ---- fused --------------------------------------------------------------------------
    def cmp_filter_sum_i64(self, cmpop: int, x, valid, off: int, n: int, threshold: int):
        s = C.c_int64()
        cnt = C.c_int64()
        check(self.handle, lib.ah_cmp_filter_sum_i64(self.handle, cmpop, _ptr(x), _ptr(valid), off, n, threshold, C.byref(s), C.byref(cnt)))
        return s.value, cnt.value

    def cmp_filter_sum_f64(self, cmpop: int, x, valid, off: int, n: int, threshold: float):
        s = C.c_double()
        cnt = C.c_int64()
        check(self.handle, lib.ah_cmp_filter_sum_f64(self.handle, cmpop, _ptr(x), _ptr(valid), off, n, threshold, C.byref(s), C.byref(cnt)))
        return s.value, cnt.value

    def cmp_filter_sum_i64_dev(self, cmpop: int, x, valid, off: int, n: int, threshold: int, out_sum_count_dev) -> None:
        check(self.handle, lib.ah_cmp_filter_sum_i64_dev(self.handle, cmpop, _ptr(x), _ptr(valid), off, n, threshold,
                                                         _ptr(out_sum_count_dev)))

    def cmp_filter_sum_f64_dev(self, cmpop: int, x, valid, off: int, n: int, threshold: float, out_sum_dev, out_count_dev) -> None:
        check(self.handle, lib.ah_cmp_filter_sum_f64_dev(self.handle, cmpop, _ptr(x), _ptr(valid), off, n, threshold,
                                                         _ptr(out_sum_dev), _ptr(out_count_dev)))


def _nodes_array(nodes):
    arr = (C.c_int32 * (2 * len(nodes)))()
    for i, (op, arg) in enumerate(nodes):
        arr[2 * i], arr[2 * i + 1] = op, arg
    return arr


def expr_codegen(nodes, col_types, lit_types, compile_it: bool = True):
    """Stateless: generate (and hiprtc-compile for gfx950) the fused kernel of a postfix
    program; works without a GPU.  Returns (source, out_type); raises on error."""
    na = _nodes_array(nodes)
    ct = (C.c_int * max(len(col_types), 1))(*col_types)
    lt = (C.c_int * max(len(lit_types), 1))(*lit_types)
    src = C.create_string_buffer(1 << 16)
    err = C.create_string_buffer(1024)
    ot = C.c_int()
    st = lib.ah_expr_codegen(na, len(nodes), ct, len(col_types), lt, len(lit_types), int(compile_it), src, len(src), err, len(err),
                             C.byref(ot))
    if st != N.AH_OK:
        raise N._ERRS.get(st, N.ArrowHipError)(err.value.decode())
    return src.value.decode(), ot.value


def device_count() -> int:
    n = C.c_int()
    st = lib.ah_device_count(C.byref(n))
    return n.value if st == N.AH_OK else 0
