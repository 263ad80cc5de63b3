"""Two implementations of the same leaf interface, so every parity test body runs
unchanged against (a) the CPU oracle and (b) the HIP library through the C ABI.

  OracleBackend — oracle/liboracle.so (checker; `-m "not gpu"` pins it to the
                  reference's golden vectors)
  HipBackend    — libarrowhip.so on cuda:0 through include/arrowhip.h (`-m gpu`)

All inputs/outputs are host numpy arrays; HipBackend uploads, runs, downloads.
`misalign` (elements) shifts the device copy of a value buffer off 16-byte alignment
the way an Arrow slice (&values[offset]) does.
"""
from __future__ import annotations

import numpy as np

from tests import oracle_lib as OL

STATUS_OK, STATUS_EINVALID, STATUS_EINDEX, STATUS_EOVERFLOW = 0, 1, 2, 3


class OracleBackend:
    name = "oracle"

    def __init__(self):
        self.o = OL.load_oracle()

    def sum(self, arr, misalign=0):
        arr = np.ascontiguousarray(arr)
        if arr.dtype == np.float64:
            return float(self.o.sum_float64_exact(arr))
        return int(self.o.sum_int64(arr)) if arr.dtype == np.int64 else int(self.o.sum_uint64(arr))

    def arithmetic(self, op, shape, l, r, misalign=0):
        return self.o.arithmetic(op, shape, l, r)

    def arithmetic_unary(self, op, a, misalign=0):
        return self.o.arithmetic_unary(op, a)

    def arithmetic_checked(self, op, shape, l, lvalid, loff, r, rvalid, roff, scalar_valid=True):
        return self.o.arithmetic_checked(op, shape, l, lvalid, loff, r, rvalid, roff, scalar_valid)

    def arithmetic_ext(self, op, shape, l, lvalid, loff, r, rvalid, roff, scalar_valid=True):
        """→ (status, out, error text)"""
        return self.o.arithmetic_ext(op, shape, l, lvalid, loff, r, rvalid, roff, scalar_valid)

    def round(self, values, valid, off, ndigits, mode, multiple=None):
        return self.o.round(values, valid, off, ndigits, mode, multiple)

    def comparison(self, cmpop, shape, l, r, out_init, out_bit_offset=0, misalign=0):
        out = np.array(out_init, dtype=np.uint8, copy=True)
        return self.o.comparison(cmpop, shape, l, r, out, out_bit_offset)

    def count_set_bits(self, bits, off, n):
        return self.o.count_set_bits(np.ascontiguousarray(bits), off, n)

    def bitmap_op(self, op, l, loff, r, roff, out_init, ooff, n):
        return self.o.bitmap_op(op, l, loff, r, roff, np.array(out_init, dtype=np.uint8, copy=True), ooff, n)

    def copy_bitmap(self, src, soff, n, out_init, doff, invert=False):
        return self.o.copy_bitmap(src, soff, n, np.array(out_init, dtype=np.uint8, copy=True), doff, invert)

    def set_bits_to(self, out_init, off, n, value):
        return self.o.set_bits_to(np.array(out_init, dtype=np.uint8, copy=True), off, n, value)

    def kleene(self, op, lvalid, ldata, loff, rvalid, rdata, roff, ovalid_init, odata_init, ooff, n):
        return self.o.kleene(op, lvalid, ldata, loff, rvalid, rdata, roff, np.array(ovalid_init, dtype=np.uint8, copy=True),
                             np.array(odata_init, dtype=np.uint8, copy=True), ooff, n)

    def filter_count(self, fdata, fvalid, foff, n, null_sel):
        return self.o.filter_count(fdata, fvalid, foff, n, null_sel)

    def filter(self, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid, misalign=0):
        return self.o.filter_primitive(values, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid)

    def filter_to_indices(self, fdata, fvalid, foff, n, null_sel, want_valid):
        return self.o.filter_to_indices(fdata, fvalid, foff, n, null_sel, want_valid)

    def take(self, values, vvalid, voff, idx, ivalid, ioff, bounds_check, want_valid):
        return self.o.take_primitive(values, vvalid, voff, idx, ivalid, ioff, bounds_check, want_valid)

    def cumulative_sum(self, values, valid, off, start=None, skip_nulls=False, checked=False, misalign=0):
        return self.o.cumulative_sum(values, valid, off, start, skip_nulls, checked)

    def cast_numeric(self, values, out_dtype, valid=None, off=0, allow_int_overflow=False, allow_float_truncate=False, misalign=0):
        return self.o.cast_numeric(values, out_dtype, valid, off, allow_int_overflow, allow_float_truncate)

    def cast_bool_to_numeric(self, bits, off, n, out_dtype):
        return self.o.cast_bool_to_numeric(bits, off, n, out_dtype)

    def shift_time(self, values, out_dtype, op, factor, check, valid=None, off=0, misalign=0):
        return self.o.shift_time(values, out_dtype, op, factor, check, valid, off)

    def is_in(self, values, valid, off, set_values, set_valid, set_off, null_behavior, out_off=0, fill=0, misalign=0):
        return self.o.is_in(values, valid, off, set_values, set_valid, set_off, null_behavior, out_off, fill)

    def sort_indices(self, values, valid, off, descending=False, nulls_at_start=False, misalign=0):
        return self.o.sort_indices(values, valid, off, descending, nulls_at_start)

    def min_max(self, values, misalign=0):
        return self.o.min_max(values)

    def sort_indices_multi(self, columns):
        return self.o.sort_indices_multi(columns)

    def take_binary(self, offsets, data, vvalid, voff, nvalues, idx, ivalid, ioff, want_valid):
        return self.o.take_binary(offsets, data, vvalid, voff, nvalues, idx, ivalid, ioff, want_valid)

    def filter_binary(self, offsets, data, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid):
        return self.o.filter_binary(offsets, data, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid)

    def hash_encode(self, keys, valid, off, encode_nulls):
        return self.o.hash_u64_encode(keys, valid, off, encode_nulls)

    def hash_binary_encode(self, offsets, data, valid, off, n, encode_nulls):
        return self.o.hash_binary_encode(offsets, data, valid, off, n, encode_nulls)

    def hash_fixed_encode(self, data, w, valid, off, n, encode_nulls):
        """fixed-width keys = binary keys whose offsets are implied (i·w): the oracle's binary memo table over explicit offsets"""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        offsets = (np.arange(data.size // w + 1, dtype=np.int64) * w)
        ids, idv, first, nid = self.o.hash_binary_encode(offsets, data, valid, off, n, encode_nulls)
        base = data[off * w:].reshape(-1, w)
        dic = np.zeros((first.size, w), np.uint8)
        for i, r in enumerate(first):
            if i != nid:
                dic[i] = base[r]
        return ids, idv, first, nid, dic.reshape(-1)

    def hash_sum(self, kind, keys, kvalid, koff, vals, vvalid, voff):
        return self.o.hash_sum(kind, keys, kvalid, koff, vals, vvalid, voff)

    def cmp_filter_sum_i64(self, cmpop, x, valid, off, thr):
        return self.o.cmp_filter_sum_i64(cmpop, x, valid, off, thr)

    def cmp_filter_sum_f64(self, cmpop, x, valid, off, thr):
        s_seq, s_exact, cnt = self.o.cmp_filter_sum_f64(cmpop, x, valid, off, thr)
        return s_exact, cnt


class HipBackend:
    name = "hip"

    def __init__(self, ctx, dirty_outputs=False):
        self.c = ctx
        self.dirty_outputs = dirty_outputs

    # -- helpers
    def _dirty(self, nbytes):
        """an output (or padded input) buffer; with dirty_outputs its memory held something else before (0xA5 …), so that a kernel which
        defines only part of its last byte — or reads behind its input — shows.  tests/test_gpu_parity.py runs that way."""
        buf = self.c.alloc(nbytes)
        if self.dirty_outputs:
            buf.memset(0xA5)
        return buf

    def _up(self, arr, misalign=0):
        """upload; returns (buffer, device pointer of element 0)"""
        if arr is None:
            return None, None
        a = np.ascontiguousarray(arr)
        lead = misalign * a.dtype.itemsize
        buf = self.c.alloc(lead + a.nbytes + 64)
        buf.upload(a, lead)
        return buf, buf.ptr + lead

    def _upbits(self, bits):
        if bits is None:
            return None, None
        a = np.ascontiguousarray(bits, dtype=np.uint8)
        buf = self._dirty(a.nbytes + 64)
        buf.upload(a)
        return buf, buf.ptr

    def sum(self, arr, misalign=0):
        arr = np.ascontiguousarray(arr)
        b, p = self._up(arr, misalign)
        if arr.dtype == np.float64:
            return self.c.sum_float64(p, arr.size)
        if arr.dtype == np.int64:
            return self.c.sum_int64(p, arr.size)
        return self.c.sum_uint64(p, arr.size)

    def arithmetic(self, op, shape, l, r, misalign=0):
        l = np.ascontiguousarray(l); r = np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        tid = OL.TYPE_IDS[arr.dtype]
        lb, lp = (None, l) if shape == 2 else self._up(l, misalign)
        rb, rp = (None, r) if shape == 1 else self._up(r, misalign)
        ob = self._dirty(arr.nbytes + 128)
        op_ptr = ob.ptr + misalign * arr.dtype.itemsize
        self.c.arithmetic(tid, op, shape, lp, rp, op_ptr, arr.size)
        return ob.download(arr.dtype, arr.size, misalign * arr.dtype.itemsize)

    def arithmetic_unary(self, op, a, misalign=0):
        a = np.ascontiguousarray(a)
        ib, ip = self._up(a, misalign)
        ob = self._dirty(a.nbytes + 128)
        self.c.arithmetic_unary(OL.TYPE_IDS[a.dtype], op, ip, ob.ptr + misalign * a.dtype.itemsize, a.size)
        return ob.download(a.dtype, a.size, misalign * a.dtype.itemsize)

    def arithmetic_checked(self, op, shape, l, lvalid, loff, r, rvalid, roff, scalar_valid=True):
        import arrow_go_amd as ah
        l = np.ascontiguousarray(l); r = np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        lb, lp = (None, l) if shape == 2 else self._up(l)
        rb, rp = (None, r) if shape == 1 else self._up(r)
        lvb, lvp = self._upbits(lvalid)
        rvb, rvp = self._upbits(rvalid)
        ob = self._dirty(arr.nbytes + 64)
        ob.memset(0xCD)
        try:
            self.c.arithmetic_checked(OL.TYPE_IDS[arr.dtype], op, shape, lp, lvp, loff, rp, rvp, roff, scalar_valid, ob, arr.size)
            st = STATUS_OK
        except ah.ErrOverflow as e:
            assert "overflow" in str(e)
            st = STATUS_EOVERFLOW
        return st, ob.download(arr.dtype, arr.size)

    def round(self, values, valid, off, ndigits, mode, multiple=None):
        import arrow_go_amd as ah
        values = np.ascontiguousarray(values)
        vb, vp = self._up(values); vvb, vvp = self._upbits(valid)
        ob = self._dirty(values.nbytes + 64); ob.memset(0xCD)
        m = None if multiple is None else np.array([multiple], values.dtype)
        try:
            self.c.round(OL.TYPE_IDS[values.dtype], vp, vvp, off, values.size, ndigits, mode, m, OL.load_oracle().pow10(abs(ndigits)), ob)
            st = STATUS_OK
        except ah.ErrOverflow:
            st = STATUS_EOVERFLOW
        return st, ob.download(values.dtype, values.size)

    def arithmetic_ext(self, op, shape, l, lvalid, loff, r, rvalid, roff, scalar_valid=True):
        import arrow_go_amd as ah
        l = np.ascontiguousarray(l); r = None if r is None else np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        lb, lp = (None, l) if shape == 2 else self._up(l)
        rb, rp = (None, r) if (shape == 1 or r is None) else self._up(r)
        lvb, lvp = self._upbits(lvalid)
        rvb, rvp = self._upbits(rvalid)
        ob = self._dirty(arr.nbytes + 64)
        ob.memset(0xCD)
        st, msg = STATUS_OK, ""
        try:
            self.c.arithmetic_ext(OL.TYPE_IDS[arr.dtype], op, shape, lp, lvp, loff, rp, rvp, roff, scalar_valid, ob, arr.size)
        except ah.ErrOverflow as e:
            st, msg = STATUS_EOVERFLOW, str(e)
        except ah.ErrInvalid as e:
            st, msg = STATUS_EINVALID, str(e)
        return st, ob.download(arr.dtype, arr.size), msg

    def comparison(self, cmpop, shape, l, r, out_init, out_bit_offset=0, misalign=0):
        l = np.ascontiguousarray(l); r = np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        lb, lp = (None, l) if shape == 2 else self._up(l, misalign)
        rb, rp = (None, r) if shape == 1 else self._up(r, misalign)
        out_init = np.ascontiguousarray(out_init, dtype=np.uint8)
        ob, op_ = self._upbits(out_init)
        self.c.comparison(cmpop, shape, OL.TYPE_IDS[arr.dtype], lp, rp, op_, arr.size, out_bit_offset)
        return ob.download(np.uint8, out_init.size)

    def count_set_bits(self, bits, off, n):
        b, p = self._upbits(bits)
        return self.c.count_set_bits(p, off, n)

    def bitmap_op(self, op, l, loff, r, roff, out_init, ooff, n):
        lb, lp = self._upbits(l); rb, rp = self._upbits(r)
        out_init = np.ascontiguousarray(out_init, dtype=np.uint8)
        ob, op_ = self._upbits(out_init)
        self.c.bitmap_op(op, lp, loff, rp, roff, op_, ooff, n)
        return ob.download(np.uint8, out_init.size)

    def copy_bitmap(self, src, soff, n, out_init, doff, invert=False):
        sb, sp = self._upbits(src)
        out_init = np.ascontiguousarray(out_init, dtype=np.uint8)
        ob, op_ = self._upbits(out_init)
        self.c.copy_bitmap(sp, soff, n, op_, doff, invert)
        return ob.download(np.uint8, out_init.size)

    def set_bits_to(self, out_init, off, n, value):
        out_init = np.ascontiguousarray(out_init, dtype=np.uint8)
        ob, op_ = self._upbits(out_init)
        self.c.set_bits_to(op_, off, n, value)
        return ob.download(np.uint8, out_init.size)

    def kleene(self, op, lvalid, ldata, loff, rvalid, rdata, roff, ovalid_init, odata_init, ooff, n):
        lvb, lvp = self._upbits(lvalid); ldb, ldp = self._upbits(ldata)
        rvb, rvp = self._upbits(rvalid); rdb, rdp = self._upbits(rdata)
        ovalid_init = np.ascontiguousarray(ovalid_init, dtype=np.uint8)
        odata_init = np.ascontiguousarray(odata_init, dtype=np.uint8)
        ovb, ovp = self._upbits(ovalid_init); odb, odp = self._upbits(odata_init)
        self.c.kleene(op, lvp, ldp, loff, rvp, rdp, roff, ovp, odp, ooff, n)
        return ovb.download(np.uint8, ovalid_init.size), odb.download(np.uint8, odata_init.size)

    def filter_count(self, fdata, fvalid, foff, n, null_sel):
        fb, fp = self._upbits(fdata); vb, vp = self._upbits(fvalid)
        return self.c.filter_count(fp, vp, foff, n, null_sel)

    def filter(self, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid, misalign=0):
        values = np.ascontiguousarray(values)
        w = values.dtype.itemsize
        vb, vp = self._up(values, misalign)
        vvb, vvp = self._upbits(vvalid)
        fb, fp = self._upbits(fdata); fvb, fvp = self._upbits(fvalid)
        n_out = self.c.filter_count(fp, fvp, foff, n, null_sel)
        ob = self._dirty(n_out * w + 128)
        ob.memset(0xCD)  # the library must not rely on a pre-zeroed value buffer
        ovb = self._dirty((n_out + 7) // 8 + 64) if want_valid else None
        if ovb is not None:
            ovb.memset(0xCD)
        nulls = self.c.filter_primitive(w, vp, vvp, voff, fp, fvp, foff, n, null_sel, n_out, ob, ovb)
        out = ob.download(values.dtype, n_out)
        ov = ovb.download(np.uint8, (n_out + 7) // 8) if want_valid else None
        # every Filter of the suite also goes through the one-call entry (ah_filter_primitive_once: outputs sized for n rows, count and
        # null count through the mailbox) and must give the same rows, validity bits, count and null count
        ob1 = self._dirty(n * w + 128)
        ob1.memset(0xCD)
        ovb1 = self._dirty((n + 7) // 8 + 64) if want_valid else None
        if ovb1 is not None:
            ovb1.memset(0xCD)
        k1, nulls1 = self.c.filter_primitive_once(w, vp, vvp, voff, fp, fvp, foff, n, null_sel, ob1, ovb1)
        assert k1 == n_out, ("filter_primitive_once: rows selected", k1, n_out)
        assert ob1.download(values.dtype, n_out).tobytes() == out.tobytes(), "filter_primitive_once: values"
        if want_valid:
            assert nulls1 == nulls, ("filter_primitive_once: null count", nulls1, nulls)
            bits = lambda b: np.unpackbits(b, bitorder="little")[:n_out]
            assert (bits(ovb1.download(np.uint8, (n_out + 7) // 8)) == bits(ov)).all(), "filter_primitive_once: validity"
        return out, ov, nulls

    def filter_to_indices(self, fdata, fvalid, foff, n, null_sel, want_valid):
        fb, fp = self._upbits(fdata); fvb, fvp = self._upbits(fvalid)
        n_out = self.c.filter_count(fp, fvp, foff, n, null_sel)
        ob = self._dirty(n_out * 4 + 128)
        ovb = self._dirty((n_out + 7) // 8 + 64) if want_valid else None
        nulls = self.c.filter_to_indices(fp, fvp, foff, n, null_sel, n_out, ob, ovb)
        return ob.download(np.uint32, n_out), (ovb.download(np.uint8, (n_out + 7) // 8) if want_valid else None), nulls

    def take(self, values, vvalid, voff, idx, ivalid, ioff, bounds_check, want_valid):
        import arrow_go_amd as ah
        values = np.ascontiguousarray(values); idx = np.ascontiguousarray(idx)
        vb, vp = self._up(values); vvb, vvp = self._upbits(vvalid)
        ib, ip = self._up(idx); ivb, ivp = self._upbits(ivalid)
        ob = self._dirty(idx.size * values.dtype.itemsize + 64)
        ob.memset(0xCD)
        ovb = self._dirty((idx.size + 7) // 8 + 64) if want_valid else None
        try:
            nulls = self.c.take_primitive(values.dtype.itemsize, vp, vvp, voff, values.size, idx.dtype.itemsize,
                                          idx.dtype.kind == "i", ip, ivp, ioff, idx.size, bounds_check, ob, ovb)
        except ah.ErrIndex as e:
            msg = str(e)
            assert msg.endswith("out of bounds"), msg
            return STATUS_EINDEX, None, None, 0, int(msg.split()[0])
        out = ob.download(values.dtype, idx.size)
        ov = ovb.download(np.uint8, (idx.size + 7) // 8) if want_valid else None
        return STATUS_OK, out, ov, nulls, 0

    # -- the *_dev flavours: worst-case outputs, status words fetched after the fact
    def filter_dev(self, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid, misalign=0):
        values = np.ascontiguousarray(values)
        w = values.dtype.itemsize
        vb, vp = self._up(values, misalign)
        vvb, vvp = self._upbits(vvalid)
        fb, fp = self._upbits(fdata); fvb, fvp = self._upbits(fvalid)
        ob = self._dirty(n * w + 128); ob.memset(0xCD)
        ovb = self._dirty((n + 7) // 8 + 64) if want_valid else None
        if ovb is not None:
            ovb.memset(0xCD)
        st = self._dirty(64); st.memset(0xCD)
        self.c.filter_primitive_dev(w, vp, vvp, voff, fp, fvp, foff, n, null_sel, ob, ovb, st)
        n_out, nulls = (int(v) for v in st.download(np.int64, 2))
        out = ob.download(values.dtype, n_out)
        ov = ovb.download(np.uint8, (n_out + 7) // 8) if want_valid else None
        return out, ov, nulls

    def take_dev(self, values, vvalid, voff, idx, ivalid, ioff, want_valid):
        """→ (values, validity bytes, null count, position of the first out-of-range index or None)"""
        values = np.ascontiguousarray(values); idx = np.ascontiguousarray(idx)
        vb, vp = self._up(values); vvb, vvp = self._upbits(vvalid)
        ib, ip = self._up(idx); ivb, ivp = self._upbits(ivalid)
        ob = self._dirty(idx.size * values.dtype.itemsize + 64); ob.memset(0xCD)
        ovb = self._dirty((idx.size + 7) // 8 + 64) if want_valid else None
        st = self._dirty(64); st.memset(0xCD)
        self.c.take_primitive_dev(values.dtype.itemsize, vp, vvp, voff, values.size, idx.dtype.itemsize, idx.dtype.kind == "i",
                                  ip, ivp, ioff, idx.size, ob, ovb, st)
        bad, nulls = (int(v) for v in st.download(np.uint64, 2))
        out = ob.download(values.dtype, idx.size)
        ov = ovb.download(np.uint8, (idx.size + 7) // 8) if want_valid else None
        return out, ov, nulls, (None if bad == 2**64 - 1 else bad)

    def cumulative_sum(self, values, valid, off, start=None, skip_nulls=False, checked=False, misalign=0):
        import arrow_go_amd as ah
        values = np.ascontiguousarray(values)
        n = values.size
        w = values.dtype.itemsize
        vb, vp = self._up(values, misalign); vvb, vvp = self._upbits(valid)
        ob = self._dirty(n * w + 128)
        ob.memset(0xCD)  # the library must write every row, including the zero payload of null rows
        ovb = self._dirty((n + 7) // 8 + 64) if valid is not None else None
        if ovb is not None:
            ovb.memset(0xFF)  # prepareCumulativeOutput pre-fills the validity with ones; only bits [0, n) are the library's
        sb = np.array([start], dtype=values.dtype).tobytes() if start is not None else None
        try:
            nulls = self.c.cumulative_sum(OL.TYPE_IDS[values.dtype], vp, vvp, off, n, sb, skip_nulls, checked,
                                          ob.ptr + misalign * w, ovb)
        except ah.ErrOverflow as e:
            assert "overflow" in str(e)
            return STATUS_EOVERFLOW, None, None, 0
        out = ob.download(values.dtype, n, misalign * w)
        ov = None
        if ovb is not None:
            ovg = ovb.download(np.uint8, (n + 7) // 8 + 16)
            assert (ovg[(n + 7) // 8:] == 0xFF).all(), "cumulative_sum wrote beyond the validity bitmap's last byte"
            ov = ovg[:(n + 7) // 8]
        return STATUS_OK, out, ov, nulls

    def cast_numeric(self, values, out_dtype, valid=None, off=0, allow_int_overflow=False, allow_float_truncate=False, misalign=0):
        import arrow_go_amd as ah
        values = np.ascontiguousarray(values)
        od = np.dtype(out_dtype)
        vb, vp = self._up(values, misalign); vvb, vvp = self._upbits(valid)
        ob = self._dirty(values.size * od.itemsize + 128)
        ob.memset(0xCD)
        try:
            self.c.cast_numeric(OL.TYPE_IDS[values.dtype], OL.TYPE_IDS[od], vp, vvp, off, values.size, allow_int_overflow,
                                allow_float_truncate, ob.ptr + misalign * od.itemsize)
        except ah.ErrInvalid as e:
            return STATUS_EINVALID, None, str(e)
        return STATUS_OK, ob.download(od, values.size, misalign * od.itemsize), ""

    def shift_time(self, values, out_dtype, op, factor, check, valid=None, off=0, misalign=0):
        import arrow_go_amd as ah
        values = np.ascontiguousarray(values)
        od = np.dtype(out_dtype)
        vb, vp = self._up(values, misalign); vvb, vvp = self._upbits(valid)
        ob = self._dirty(values.size * od.itemsize + 128)
        ob.memset(0xCD)
        try:
            self.c.shift_time(values.dtype.itemsize * 8, od.itemsize * 8, op, factor, check, vp, vvp, off, values.size, ob.ptr + misalign * od.itemsize)
        except ah.ErrInvalid as e:
            return STATUS_EINVALID, None, e.bad_value
        return STATUS_OK, ob.download(od, values.size, misalign * od.itemsize), 0

    def cast_bool_to_numeric(self, bits, off, n, out_dtype):
        od = np.dtype(out_dtype)
        bb, bp = self._upbits(bits)
        ob = self._dirty(n * od.itemsize + 64)
        ob.memset(0xCD)
        self.c.cast_bool_to_numeric(OL.TYPE_IDS[od], bp, off, n, ob)
        return ob.download(od, n)

    def is_in(self, values, valid, off, set_values, set_valid, set_off, null_behavior, out_off=0, fill=0, misalign=0):
        values = np.ascontiguousarray(values); set_values = np.ascontiguousarray(set_values)
        n = values.size
        vb, vp = self._up(values, misalign); vvb, vvp = self._upbits(valid)
        sb, sp = self._up(set_values) if set_values.size else (None, None); svb, svp = self._upbits(set_valid)
        nb = (out_off + n + 7) // 8 + 1
        odb = self._dirty(nb + 64); ovb = self._dirty(nb + 64)
        odb.memset(fill); ovb.memset(fill)
        self.c.is_in(values.dtype.itemsize, vp, vvp, off, n, sp, svp, set_off, set_values.size, null_behavior, odb, ovb, out_off)
        return odb.download(np.uint8, nb), ovb.download(np.uint8, nb)

    def take_binary(self, offsets, data, vvalid, voff, nvalues, idx, ivalid, ioff, want_valid):
        import arrow_go_amd as ah
        offsets = np.ascontiguousarray(offsets); idx = np.ascontiguousarray(idx)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = idx.size
        w = offsets.dtype.itemsize
        ofb, ofp = self._up(offsets); db, dp = self._up(data) if data.size else (None, None)
        vvb, vvp = self._upbits(vvalid); ib, ip = self._up(idx) if n else (None, None); ivb, ivp = self._upbits(ivalid)
        oob = self._dirty((n + 1) * w + 64); oob.memset(0xCD)
        ovb = self._dirty((n + 7) // 8 + 64) if want_valid else None
        try:
            nulls, total = self.c.take_binary_offsets(w, ofp, vvp, voff, nvalues, idx.dtype.itemsize, idx.dtype.kind == "i", ip, ivp, ioff, n, oob, ovb)
        except ah.ErrIndex as e:
            return STATUS_EINDEX, None, None, None, 0, int(str(e).split()[0])
        odb = self._dirty(total + 64); odb.memset(0xCD)
        self.c.take_binary_data(w, ofp, dp, voff, idx.dtype.itemsize, ip, n, oob, odb)
        return (STATUS_OK, oob.download(offsets.dtype, n + 1), odb.download(np.uint8, total),
                (ovb.download(np.uint8, (n + 7) // 8) if want_valid else None), nulls, 0)

    def filter_binary(self, offsets, data, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid):
        """filter = GetTakeIndices + var-length take (ah_filter_count → ah_filter_to_indices → ah_take_binary_*)"""
        offsets = np.ascontiguousarray(offsets); data = np.ascontiguousarray(data, dtype=np.uint8)
        w = offsets.dtype.itemsize
        ofb, ofp = self._up(offsets); db, dp = self._up(data) if data.size else (None, None)
        vvb, vvp = self._upbits(vvalid); fb, fp = self._upbits(fdata); fvb, fvp = self._upbits(fvalid)
        n_out = self.c.filter_count(fp, fvp, foff, n, null_sel)
        ib = self._dirty(n_out * 4 + 64); ivb = self._dirty((n_out + 7) // 8 + 64)
        idx_nulls = self.c.filter_to_indices(fp, fvp, foff, n, null_sel, n_out, ib, ivb)
        oob = self._dirty((n_out + 1) * w + 64)
        ovb = self._dirty((n_out + 7) // 8 + 64) if want_valid else None
        nulls, total = self.c.take_binary_offsets(w, ofp, vvp, voff, n, 4, False, ib, ivb if idx_nulls else None, 0, n_out, oob, ovb)
        odb = self._dirty(total + 64)
        self.c.take_binary_data(w, ofp, dp, voff, 4, ib, n_out, oob, odb)
        return (oob.download(offsets.dtype, n_out + 1), odb.download(np.uint8, total),
                (ovb.download(np.uint8, (n_out + 7) // 8) if want_valid else None), nulls)

    def sort_indices_multi(self, columns):
        keep, keys = [], []
        n = np.asarray(columns[0][0]).size
        for values, valid, off, desc, nfirst in columns:
            values = np.ascontiguousarray(values)
            vb, vp = self._up(values); vvb, vvp = self._upbits(valid)
            keep += [vb, vvb]
            keys.append((OL.TYPE_IDS[values.dtype], vp, vvp, off, desc, nfirst))
        ob = self._dirty(n * 8 + 64)
        self.c.sort_indices_multi(keys, n, ob)
        return ob.download(np.uint64, n)

    def min_max(self, values, misalign=0):
        values = np.ascontiguousarray(values)
        vb, vp = self._up(values, misalign) if values.size else (None, None)
        return self.c.min_max(OL.TYPE_IDS[values.dtype], vp, values.size, values.dtype)

    def sort_indices(self, values, valid, off, descending=False, nulls_at_start=False, misalign=0):
        values = np.ascontiguousarray(values)
        n = values.size
        vb, vp = self._up(values, misalign); vvb, vvp = self._upbits(valid)
        ob = self._dirty(n * 8 + 64)
        ob.memset(0xCD)
        self.c.sort_indices(OL.TYPE_IDS[values.dtype], vp, vvp, off, n, descending, nulls_at_start, ob)
        return ob.download(np.uint64, n)

    def hash_encode(self, keys, valid, off, encode_nulls):
        keys = np.ascontiguousarray(keys).view(np.uint64)
        n = keys.size
        kb, kp = self._up(keys); vb, vp = self._upbits(valid)
        idb = self._dirty(n * 4 + 64); idvb = self._dirty((n + 7) // 8 + 64); db = self._dirty((n + 1) * 8 + 64)
        nd, nid = self.c.hash_u64_encode(kp, vp, off, n, encode_nulls, idb, idvb, db)
        return idb.download(np.int32, n), idvb.download(np.uint8, (n + 7) // 8), db.download(np.uint64, nd), nid

    def hash_binary_encode(self, offsets, data, valid, off, n, encode_nulls):
        offsets = np.ascontiguousarray(offsets); data = np.ascontiguousarray(data, dtype=np.uint8)
        ofb, ofp = self._up(offsets); db, dp = self._up(data) if data.size else (None, None); vb, vp = self._upbits(valid)
        idb = self._dirty(n * 4 + 64); idvb = self._dirty((n + 7) // 8 + 64); frb = self._dirty((n + 1) * 8 + 64)
        nd, nid = self.c.hash_binary_encode(offsets.dtype.itemsize, ofp, dp, vp, off, n, encode_nulls, idb, idvb, frb)
        return idb.download(np.int32, n), idvb.download(np.uint8, (n + 7) // 8), frb.download(np.int64, nd), nid

    def hash_fixed_encode(self, data, w, valid, off, n, encode_nulls):
        data = np.ascontiguousarray(data, dtype=np.uint8)
        db, dp = self._up(data); vb, vp = self._upbits(valid)
        idb = self._dirty(n * 4 + 64); idvb = self._dirty((n + 7) // 8 + 64); frb = self._dirty((n + 1) * 8 + 64); dcb = self._dirty((n + 1) * w + 64)
        nd, nid = self.c.hash_fixed_encode(w, dp, vp, off, n, encode_nulls, idb, idvb, frb, dcb)
        return idb.download(np.int32, n), idvb.download(np.uint8, (n + 7) // 8), frb.download(np.int64, nd), nid, dcb.download(np.uint8, nd * w)

    def hash_sum(self, kind, keys, kvalid, koff, vals, vvalid, voff):
        keys = np.ascontiguousarray(keys).view(np.uint64); vals = np.ascontiguousarray(vals)
        n = keys.size
        kb, kp = self._up(keys); kvb, kvp = self._upbits(kvalid)
        xb, xp = self._up(vals); xvb, xvp = self._upbits(vvalid)
        okb = self._dirty((n + 1) * 8 + 64); osb = self._dirty((n + 1) * 8 + 64); ocb = self._dirty((n + 1) * 8 + 64)
        ofb = self._dirty((n + 1) * 8 + 64)
        ng, nid = self.c.hash_sum(kind, kp, kvp, koff, xp, xvp, voff, n, okb, osb, ocb, ofb)
        return okb.download(np.uint64, ng), osb.download(vals.dtype, ng), ocb.download(np.int64, ng), nid, ofb.download(np.int64, ng)

    def cmp_filter_sum_i64(self, cmpop, x, valid, off, thr, misalign=0):
        x = np.ascontiguousarray(x)
        xb, xp = self._up(x, misalign); vb, vp = self._upbits(valid)
        return self.c.cmp_filter_sum_i64(cmpop, xp, vp, off, x.size, int(thr))

    def cmp_filter_sum_f64(self, cmpop, x, valid, off, thr, misalign=0):
        x = np.ascontiguousarray(x)
        xb, xp = self._up(x, misalign); vb, vp = self._upbits(valid)
        return self.c.cmp_filter_sum_f64(cmpop, xp, vp, off, x.size, float(thr))
