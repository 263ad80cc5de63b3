"""ctypes access to the CPU oracle (oracle/liboracle.so) and, when built, to the
reference's own AVX2 machine code (oracle/_ref/libref_avx2.so, libref_c.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  All arguments are numpy arrays (host memory).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# arrow.Type ids
TYPE_IDS = {np.dtype("uint8"): 2, np.dtype("int8"): 3, np.dtype("uint16"): 4, np.dtype("int16"): 5,
            np.dtype("uint32"): 6, np.dtype("int32"): 7, np.dtype("uint64"): 8, np.dtype("int64"): 9,
            np.dtype("float32"): 11, np.dtype("float64"): 12}
ALL_DTYPES = list(TYPE_IDS.keys())
INT_DTYPES = [d for d in ALL_DTYPES if d.kind in "iu"]


def build_oracle() -> None:
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pack_bits(bools) -> np.ndarray:
    """LSB-first bitmap of a bool sequence (Arrow layout)."""
    return np.packbits(np.asarray(bools, dtype=bool), bitorder="little")


def unpack_bits(buf: np.ndarray, off: int, n: int) -> np.ndarray:
    return np.unpackbits(np.asarray(buf, dtype=np.uint8), bitorder="little")[off:off + n].astype(bool)


class Oracle:
    def __init__(self, lib: C.CDLL):
        self.lib = lib
        vp, i64, i32, i8, it, sz = C.c_void_p, C.c_int64, C.c_int32, C.c_int8, C.c_int, C.c_size_t
        sig = {
            "orc_sum_float64_seq": (None, [vp, sz, vp]),
            "orc_sum_float64_avx2order": (None, [vp, sz, vp]),
            "orc_sum_float64_exact": (None, [vp, sz, vp]),
            "orc_sum_float64_xreal": (None, [vp, sz, vp]),
            "orc_sum_int64": (None, [vp, sz, vp]),
            "orc_sum_uint64": (None, [vp, sz, vp]),
            "orc_arithmetic_binary": (it, [it, i8, vp, vp, vp, i64]),
            "orc_arithmetic_arr_scalar": (it, [it, i8, vp, vp, vp, i64]),
            "orc_arithmetic_scalar_arr": (it, [it, i8, vp, vp, vp, i64]),
            "orc_arithmetic_unary": (it, [it, i8, vp, vp, i64]),
            "orc_arithmetic_checked": (it, [it, i8, it, vp, vp, i64, vp, vp, i64, it, vp, i64]),
            "orc_round": (it, [it, vp, vp, i64, i64, i64, it, vp, vp]),
            "orc_pow10": (C.c_double, [it]),
            "orc_arithmetic_ext": (it, [it, it, it, vp, vp, i64, vp, vp, i64, it, vp, i64, vp]),
            "orc_comparison": (it, [it, it, it, vp, vp, vp, i64, it]),
            "orc_count_set_bits": (i64, [vp, i64, i64]),
            "orc_bitmap_op": (None, [it, vp, i64, vp, i64, vp, i64, i64]),
            "orc_copy_bitmap": (None, [vp, i64, i64, vp, i64, it]),
            "orc_set_bits_to": (None, [vp, i64, i64, it]),
            "orc_kleene": (None, [it, vp, vp, i64, vp, vp, i64, vp, vp, i64, i64]),
            "orc_filter_count": (i64, [vp, vp, i64, i64, it]),
            "orc_filter_primitive": (it, [it, vp, vp, i64, vp, vp, i64, i64, it, vp, vp, vp, vp]),
            "orc_take_primitive": (it, [it, vp, vp, i64, i64, it, it, vp, vp, i64, i64, it, vp, vp, vp, vp]),
            "orc_filter_to_indices": (it, [vp, vp, i64, i64, it, vp, vp, vp, vp]),
            "orc_cumulative_sum": (it, [it, vp, vp, i64, i64, vp, it, it, vp, vp, vp]),
            "orc_cast_numeric": (it, [it, it, vp, vp, i64, i64, it, it, vp, vp, vp]),
            "orc_shift_time": (it, [it, it, it, i64, it, vp, vp, i64, i64, vp, vp]),
            "orc_cast_bool_to_numeric": (it, [it, vp, i64, i64, vp]),
            "orc_is_in": (it, [it, vp, vp, i64, i64, vp, vp, i64, i64, it, vp, vp, i64]),
            "orc_sort_indices": (it, [it, vp, vp, i64, i64, it, it, vp]),
            "orc_min_max": (it, [it, vp, i64, vp, vp]),
            "orc_sort_indices_multi": (it, [it, vp, vp, vp, vp, i64, vp, vp, vp]),
            "orc_take_binary": (it, [it, vp, vp, vp, i64, i64, it, it, vp, vp, i64, i64, it, vp, vp, vp, vp, vp, vp]),
            "orc_filter_binary": (it, [it, vp, vp, vp, i64, vp, vp, i64, i64, it, vp, vp, vp, vp, vp, vp]),
            "orc_take_boolean": (it, [vp, vp, i64, i64, it, it, vp, vp, i64, i64, it, vp, vp, vp, vp]),
            "orc_hash_int": (C.c_uint64, [C.c_uint64, C.c_uint64]),
            "orc_hash_u64_encode": (it, [vp, vp, i64, i64, it, vp, vp, vp, vp, vp]),
            "orc_hash_binary_encode": (it, [it, vp, vp, vp, i64, i64, it, vp, vp, vp, vp, vp]),
            "orc_hash_sum_f64": (it, [vp, vp, i64, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp]),
            "orc_hash_sum_i64": (it, [vp, vp, i64, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp]),
            "orc_cmp_filter_sum_i64": (it, [it, vp, vp, i64, i64, i64, vp, vp]),
            "orc_cmp_filter_sum_f64": (it, [it, vp, vp, i64, i64, C.c_double, vp, vp, vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(lib, name)
            f.restype = res
            f.argtypes = args

    # ---- sums ---------------------------------------------------------------------------
    def _sum(self, fn, a, dtype):
        a = np.ascontiguousarray(a)
        r = np.zeros(1, dtype=dtype)
        getattr(self.lib, fn)(_p(a), a.size, _p(r))
        return r[0]

    def sum_float64_seq(self, a): return self._sum("orc_sum_float64_seq", a, np.float64)
    def sum_float64_avx2order(self, a): return self._sum("orc_sum_float64_avx2order", a, np.float64)
    def sum_float64_exact(self, a): return self._sum("orc_sum_float64_exact", a, np.float64)
    def sum_float64_xreal(self, a): return self._sum("orc_sum_float64_xreal", a, np.float64)
    def sum_int64(self, a): return self._sum("orc_sum_int64", a, np.int64)
    def sum_uint64(self, a): return self._sum("orc_sum_uint64", a, np.uint64)

    # ---- arithmetic ---------------------------------------------------------------------
    def arithmetic(self, op, shape, l, r):
        l = np.ascontiguousarray(l); r = np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        out = np.zeros(arr.size, dtype=arr.dtype)
        fn = [self.lib.orc_arithmetic_binary, self.lib.orc_arithmetic_arr_scalar, self.lib.orc_arithmetic_scalar_arr][shape]
        st = fn(TYPE_IDS[arr.dtype], op, _p(l), _p(r), _p(out), arr.size)
        assert st == 0, st
        return out

    def arithmetic_unary(self, op, a):
        a = np.ascontiguousarray(a)
        out = np.zeros(a.size, dtype=a.dtype)
        st = self.lib.orc_arithmetic_unary(TYPE_IDS[a.dtype], op, _p(a), _p(out), a.size)
        assert st == 0, st
        return out

    def arithmetic_checked(self, op, shape, l, lvalid, loff, r, rvalid, roff, scalar_valid=True):
        l = np.ascontiguousarray(l); r = np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        out = np.zeros(arr.size, dtype=arr.dtype)
        st = self.lib.orc_arithmetic_checked(TYPE_IDS[arr.dtype], op, shape, _p(l), _p(lvalid), loff, _p(r), _p(rvalid), roff,
                                             int(scalar_valid), _p(out), arr.size)
        return st, out

    def arithmetic_ext(self, op, shape, l, lvalid, loff, r, rvalid, roff, scalar_valid=True):
        """→ (status, out, message)"""
        l = np.ascontiguousarray(l); r = None if r is None else np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        out = np.zeros(arr.size, dtype=arr.dtype)
        msg = C.create_string_buffer(256)
        st = self.lib.orc_arithmetic_ext(TYPE_IDS[arr.dtype], op, shape, _p(l), _p(lvalid), loff, _p(r), _p(rvalid), roff,
                                         int(scalar_valid), _p(out), arr.size, msg)
        return st, out, msg.value.decode()

    def pow10(self, n): return float(self.lib.orc_pow10(int(n)))

    def round(self, values, valid, off, ndigits, mode, multiple=None):
        values = np.ascontiguousarray(values)
        out = np.zeros(values.size, values.dtype)
        m = None if multiple is None else np.array([multiple], values.dtype)
        st = self.lib.orc_round(TYPE_IDS[values.dtype], _p(values), _p(valid), off, values.size, ndigits, mode, _p(m), _p(out))
        return st, out

    # ---- compare ------------------------------------------------------------------------
    def comparison(self, cmpop, shape, l, r, out_bits, out_bit_offset=0):
        l = np.ascontiguousarray(l); r = np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        st = self.lib.orc_comparison(cmpop, shape, TYPE_IDS[arr.dtype], _p(l), _p(r), _p(out_bits), arr.size, out_bit_offset)
        assert st == 0, st
        return out_bits

    # ---- bitmaps ------------------------------------------------------------------------
    def count_set_bits(self, bits, off, n): return int(self.lib.orc_count_set_bits(_p(bits), off, n))

    def bitmap_op(self, op, l, loff, r, roff, out, ooff, n):
        self.lib.orc_bitmap_op(op, _p(l), loff, _p(r), roff, _p(out), ooff, n)
        return out

    def copy_bitmap(self, src, soff, n, dst, doff, invert=False):
        self.lib.orc_copy_bitmap(_p(src), soff, n, _p(dst), doff, int(invert))
        return dst

    def set_bits_to(self, bits, off, n, value):
        self.lib.orc_set_bits_to(_p(bits), off, n, int(value))
        return bits

    def kleene(self, op, lvalid, ldata, loff, rvalid, rdata, roff, ovalid, odata, ooff, n):
        self.lib.orc_kleene(op, _p(lvalid), _p(ldata), loff, _p(rvalid), _p(rdata), roff, _p(ovalid), _p(odata), ooff, n)
        return ovalid, odata

    # ---- selection ----------------------------------------------------------------------
    def filter_count(self, fdata, fvalid, foff, n, null_sel):
        return int(self.lib.orc_filter_count(_p(fdata), _p(fvalid), foff, n, null_sel))

    def filter_primitive(self, values, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid):
        """values: numpy array positioned at row 0 of the logical array (offset applied)."""
        values = np.ascontiguousarray(values)
        w = values.dtype.itemsize
        n_out = self.filter_count(fdata, fvalid, foff, n, null_sel)
        out = np.zeros(max(n_out, 1), dtype=values.dtype)
        ov = np.zeros((n_out + 7) // 8 + 1, dtype=np.uint8) if want_valid else None
        olen = np.zeros(1, np.int64); onull = np.zeros(1, np.int64)
        st = self.lib.orc_filter_primitive(w, _p(values), _p(vvalid), voff, _p(fdata), _p(fvalid), foff, n, null_sel,
                                           _p(out), _p(ov), _p(olen), _p(onull))
        assert st == 0, st
        return out[:n_out], (ov[:(n_out + 7) // 8] if want_valid else None), int(onull[0])

    def filter_to_indices(self, fdata, fvalid, foff, n, null_sel, want_valid):
        n_out = self.filter_count(fdata, fvalid, foff, n, null_sel)
        out = np.zeros(max(n_out, 1), dtype=np.uint32)
        ov = np.zeros((n_out + 7) // 8 + 1, dtype=np.uint8) if want_valid else None
        olen = np.zeros(1, np.int64); onull = np.zeros(1, np.int64)
        st = self.lib.orc_filter_to_indices(_p(fdata), _p(fvalid), foff, n, null_sel, _p(out), _p(ov), _p(olen), _p(onull))
        assert st == 0, st
        return out[:n_out], (ov[:(n_out + 7) // 8] if want_valid else None), int(onull[0])

    def take_primitive(self, values, vvalid, voff, idx, ivalid, ioff, bounds_check, want_valid):
        values = np.ascontiguousarray(values); idx = np.ascontiguousarray(idx)
        out = np.zeros(max(idx.size, 1), dtype=values.dtype)
        ov = np.zeros((idx.size + 7) // 8 + 1, dtype=np.uint8) if want_valid else None
        onull = np.zeros(1, np.int64); bad = np.zeros(1, np.int64)
        st = self.lib.orc_take_primitive(values.dtype.itemsize, _p(values), _p(vvalid), voff, values.size, idx.dtype.itemsize,
                                         int(idx.dtype.kind == "i"), _p(idx), _p(ivalid), ioff, idx.size, int(bounds_check),
                                         _p(out), _p(ov), _p(onull), _p(bad))
        return st, out[:idx.size], (ov[:(idx.size + 7) // 8] if want_valid else None), int(onull[0]), int(bad[0])

    # ---- cumulative sum ----------------------------------------------------------------
    def cumulative_sum(self, values, valid, off, start, skip_nulls, checked):
        """→ (status, out, out_valid or None, null_count).  start: numpy scalar of values.dtype or None."""
        values = np.ascontiguousarray(values)
        n = values.size
        out = np.zeros(max(n, 1), dtype=values.dtype)
        ov = np.zeros((n + 7) // 8 + 1, np.uint8) if valid is not None else None
        nulls = np.zeros(1, np.int64)
        sv = np.array([start], dtype=values.dtype) if start is not None else None
        st = self.lib.orc_cumulative_sum(TYPE_IDS[values.dtype], _p(values), _p(valid), off, n, _p(sv), int(skip_nulls), int(checked),
                                         _p(out), _p(ov), _p(nulls))
        return st, out[:n], (ov[:(n + 7) // 8] if ov is not None else None), int(nulls[0])

    # ---- cast ---------------------------------------------------------------------------
    def cast_numeric(self, values, out_dtype, valid=None, off=0, allow_int_overflow=False, allow_float_truncate=False):
        """→ (status, out, message)"""
        values = np.ascontiguousarray(values)
        out = np.zeros(max(values.size, 1), dtype=out_dtype)
        bad = np.zeros(1, np.int64)
        msg = C.create_string_buffer(256)
        st = self.lib.orc_cast_numeric(TYPE_IDS[values.dtype], TYPE_IDS[np.dtype(out_dtype)], _p(values), _p(valid), off, values.size,
                                       int(allow_int_overflow), int(allow_float_truncate), _p(out), _p(bad), msg)
        return st, out[:values.size], msg.value.decode()

    def shift_time(self, values, out_dtype, op, factor, check, valid=None, off=0):
        """→ (status, out, bad value)"""
        values = np.ascontiguousarray(values)
        out = np.zeros(max(values.size, 1), dtype=out_dtype)
        bad = np.zeros(1, np.int64)
        st = self.lib.orc_shift_time(values.dtype.itemsize * 8, np.dtype(out_dtype).itemsize * 8, op, factor, int(check), _p(values), _p(valid),
                                     off, values.size, _p(out), _p(bad))
        return st, out[:values.size], int(bad[0])

    def cast_bool_to_numeric(self, bits, off, n, out_dtype):
        out = np.zeros(max(n, 1), dtype=out_dtype)
        st = self.lib.orc_cast_bool_to_numeric(TYPE_IDS[np.dtype(out_dtype)], _p(bits), off, n, _p(out))
        assert st == 0, st
        return out[:n]

    # ---- set lookup ---------------------------------------------------------------------
    def is_in(self, values, valid, off, set_values, set_valid, set_off, null_behavior, out_off=0, fill=0):
        """→ (data bits, validity bits) covering out_off + n bits, pre-filled with `fill` bytes"""
        values = np.ascontiguousarray(values); set_values = np.ascontiguousarray(set_values)
        n = values.size
        nb = (out_off + n + 7) // 8 + 1
        od = np.full(nb, fill, np.uint8); ov = np.full(nb, fill, np.uint8)
        st = self.lib.orc_is_in(values.dtype.itemsize, _p(values), _p(valid), off, n, _p(set_values), _p(set_valid), set_off,
                                set_values.size, null_behavior, _p(od), _p(ov), out_off)
        assert st == 0, st
        return od, ov

    # ---- min / max ----------------------------------------------------------------------
    def min_max(self, values):
        values = np.ascontiguousarray(values)
        lo, hi = np.zeros(1, values.dtype), np.zeros(1, values.dtype)
        st = self.lib.orc_min_max(TYPE_IDS[values.dtype], _p(values), values.size, _p(lo), _p(hi))
        assert st == 0, st
        return lo[0], hi[0]

    # ---- sort ---------------------------------------------------------------------------
    def sort_indices(self, values, valid, off, descending, nulls_at_start):
        values = np.ascontiguousarray(values)
        out = np.zeros(max(values.size, 1), np.uint64)
        st = self.lib.orc_sort_indices(TYPE_IDS[values.dtype], _p(values), _p(valid), off, values.size, int(descending), int(nulls_at_start), _p(out))
        assert st == 0, st
        return out[:values.size]

    def sort_indices_multi(self, columns):
        """columns: [(values, valid, off, descending, nulls_at_start), …] most significant first"""
        cols = [(np.ascontiguousarray(c[0]),) + tuple(c[1:]) for c in columns]
        k, n = len(cols), cols[0][0].size
        types = (C.c_int * k)(*[TYPE_IDS[c[0].dtype] for c in cols])
        vals = (C.c_void_p * k)(*[c[0].ctypes.data for c in cols])
        valids = (C.c_void_p * k)(*[c[1].ctypes.data if c[1] is not None else None for c in cols])
        offs = (C.c_int64 * k)(*[c[2] for c in cols])
        desc = (C.c_int * k)(*[int(c[3]) for c in cols])
        nfirst = (C.c_int * k)(*[int(c[4]) for c in cols])
        out = np.zeros(max(n, 1), np.uint64)
        st = self.lib.orc_sort_indices_multi(k, types, vals, valids, offs, n, desc, nfirst, _p(out))
        assert st == 0, st
        return out[:n]

    def take_boolean(self, data, vvalid, voff, nvalues, idx, ivalid, ioff, want_valid):
        idx = np.ascontiguousarray(idx)
        n = idx.size
        od = np.zeros((n + 7) // 8 + 1, np.uint8); ov = np.zeros((n + 7) // 8 + 1, np.uint8) if want_valid else None
        nulls, bad = np.zeros(1, np.int64), np.zeros(1, np.int64)
        st = self.lib.orc_take_boolean(_p(data), _p(vvalid), voff, nvalues, idx.dtype.itemsize, int(idx.dtype.kind == "i"), _p(idx), _p(ivalid),
                                       ioff, n, 1, _p(od), _p(ov), _p(nulls), _p(bad))
        return st, od[:(n + 7) // 8], (ov[:(n + 7) // 8] if want_valid else None), int(nulls[0]), int(bad[0])

    # ---- var-length take / filter -------------------------------------------------------
    def take_binary(self, offsets, data, vvalid, voff, nvalues, idx, ivalid, ioff, want_valid, bounds_check=True):
        """→ (status, out_offsets, out_data, out_valid, nulls, bad_index)"""
        offsets = np.ascontiguousarray(offsets); idx = np.ascontiguousarray(idx)
        data = np.ascontiguousarray(data, dtype=np.uint8)
        n = idx.size
        oo = np.zeros(n + 1, offsets.dtype)
        od = np.zeros(max(int(data.size) * max(n, 1), 1), np.uint8) if n * max(data.size, 1) < (1 << 26) else np.zeros(1 << 26, np.uint8)
        ov = np.zeros((n + 7) // 8 + 1, np.uint8) if want_valid else None
        nulls, total, bad = np.zeros(1, np.int64), np.zeros(1, np.int64), np.zeros(1, np.int64)
        st = self.lib.orc_take_binary(offsets.dtype.itemsize, _p(offsets), _p(data), _p(vvalid), voff, nvalues, idx.dtype.itemsize,
                                      int(idx.dtype.kind == "i"), _p(idx), _p(ivalid), ioff, n, int(bounds_check), _p(oo), _p(od), _p(ov),
                                      _p(nulls), _p(total), _p(bad))
        return st, oo, od[:int(total[0])], (ov[:(n + 7) // 8] if want_valid else None), int(nulls[0]), int(bad[0])

    def filter_binary(self, offsets, data, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid):
        """→ (out_offsets[:n_out+1], out_data, out_valid, nulls)"""
        offsets = np.ascontiguousarray(offsets); data = np.ascontiguousarray(data, dtype=np.uint8)
        oo = np.zeros(n + 1, offsets.dtype); od = np.zeros(max(data.size, 1), np.uint8)
        ov = np.zeros((n + 7) // 8 + 1, np.uint8) if want_valid else None
        n_out, nulls, total = np.zeros(1, np.int64), np.zeros(1, np.int64), np.zeros(1, np.int64)
        st = self.lib.orc_filter_binary(offsets.dtype.itemsize, _p(offsets), _p(data), _p(vvalid), voff, _p(fdata), _p(fvalid), foff, n,
                                        null_sel, _p(oo), _p(od), _p(ov), _p(n_out), _p(nulls), _p(total))
        assert st == 0, st
        m = int(n_out[0])
        return oo[:m + 1], od[:int(total[0])], (ov[:(m + 7) // 8] if want_valid else None), int(nulls[0])

    # ---- hashing ------------------------------------------------------------------------
    def hash_int(self, v, alg=0): return int(self.lib.orc_hash_int(int(v) & (2**64 - 1), alg))

    def hash_u64_encode(self, keys, valid, off, encode_nulls):
        keys = np.ascontiguousarray(keys).view(np.uint64)
        n = keys.size
        ids = np.zeros(max(n, 1), np.int32); idv = np.zeros((n + 7) // 8 + 1, np.uint8)
        d = np.zeros(n + 1, np.uint64); nd = np.zeros(1, np.int64); nid = np.zeros(1, np.int32)
        st = self.lib.orc_hash_u64_encode(_p(keys), _p(valid), off, n, int(encode_nulls), _p(ids), _p(idv), _p(d), _p(nd), _p(nid))
        assert st == 0, st
        return ids[:n], idv[:(n + 7) // 8], d[:int(nd[0])], int(nid[0])

    def hash_binary_encode(self, offsets, data, valid, off, n, encode_nulls):
        """→ ids, ids validity, first rows (one per dictionary entry), null id"""
        offsets = np.ascontiguousarray(offsets); data = np.ascontiguousarray(data, dtype=np.uint8)
        ids = np.zeros(max(n, 1), np.int32); idv = np.zeros((n + 7) // 8 + 1, np.uint8)
        fr = np.zeros(n + 1, np.int64); nd = np.zeros(1, np.int64); nid = np.zeros(1, np.int32)
        st = self.lib.orc_hash_binary_encode(offsets.dtype.itemsize, _p(offsets), _p(data) if data.size else None, _p(valid), off, n,
                                             int(encode_nulls), _p(ids), _p(idv), _p(fr), _p(nd), _p(nid))
        assert st == 0, st
        return ids[:n], idv[:(n + 7) // 8], fr[:int(nd[0])], int(nid[0])

    def hash_sum(self, kind, keys, kvalid, koff, vals, vvalid, voff):
        keys = np.ascontiguousarray(keys).view(np.uint64); vals = np.ascontiguousarray(vals)
        n = keys.size
        ok = np.zeros(n + 1, np.uint64); os_ = np.zeros(n + 1, vals.dtype); oc = np.zeros(n + 1, np.int64)
        of = np.zeros(n + 1, np.int64)
        ng = np.zeros(1, np.int64); nid = np.zeros(1, np.int32)
        fn = self.lib.orc_hash_sum_f64 if kind == "f64" else self.lib.orc_hash_sum_i64
        st = fn(_p(keys), _p(kvalid), koff, _p(vals), _p(vvalid), voff, n, _p(ok), _p(os_), _p(oc), _p(of), _p(ng), _p(nid))
        assert st == 0, st
        g = int(ng[0])
        return ok[:g], os_[:g], oc[:g], int(nid[0]), of[:g]

    # ---- fused --------------------------------------------------------------------------
    def cmp_filter_sum_i64(self, cmpop, x, valid, off, thr):
        x = np.ascontiguousarray(x); s = np.zeros(1, np.int64); c = np.zeros(1, np.int64)
        self.lib.orc_cmp_filter_sum_i64(cmpop, _p(x), _p(valid), off, x.size, int(thr), _p(s), _p(c))
        return int(s[0]), int(c[0])

    def cmp_filter_sum_f64(self, cmpop, x, valid, off, thr):
        x = np.ascontiguousarray(x); s = np.zeros(1, np.float64); e = np.zeros(1, np.float64); c = np.zeros(1, np.int64)
        self.lib.orc_cmp_filter_sum_f64(cmpop, _p(x), _p(valid), off, x.size, float(thr), _p(s), _p(e), _p(c))
        return float(s[0]), float(e[0]), int(c[0])


_cached = None


def load_oracle() -> Oracle:
    global _cached
    if _cached is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        _cached = Oracle(C.CDLL(path))
    return _cached


class Reference:
    """The reference's own kernels: AVX2 machine code assembled from its clang output
    (libref_avx2.so) and its C sum sources compiled strict-sequential (libref_c.so)."""

    CMP_NAMES = {0: "equal", 1: "not_equal", 2: "greater", 3: "greater_equal"}
    SHAPE_NAMES = {0: "arr_arr", 1: "arr_scalar", 2: "scalar_arr"}

    def __init__(self, avx2: C.CDLL, cseq: C.CDLL):
        self.avx2, self.cseq = avx2, cseq

    def sum(self, which, a):
        a = np.ascontiguousarray(a)
        r = np.zeros(1, dtype=a.dtype)
        name = {np.dtype("float64"): "float64", np.dtype("int64"): "int64", np.dtype("uint64"): "uint64"}[a.dtype]
        lib, suffix = (self.avx2, "avx2") if which == "avx2" else (self.cseq, "x86")
        f = getattr(lib, f"sum_{name}_{suffix}")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        f(_p(a), a.size, _p(r))
        return r[0]

    def arithmetic(self, op, shape, l, r):
        l = np.ascontiguousarray(l); r = np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        out = np.zeros(arr.size, dtype=arr.dtype)
        f = getattr(self.avx2, ["arithmetic_binary_avx2", "arithmetic_arr_scalar_avx2", "arithmetic_scalar_arr_avx2"][shape])
        f.restype = None
        f.argtypes = [C.c_int, C.c_int8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        f(TYPE_IDS[arr.dtype], op, _p(l), _p(r), _p(out), arr.size)
        return out

    def cast_numeric(self, a, out_dtype):
        """cast_type_numeric_avx2 (kernels/_lib/cast_numeric.cc:62): the unchecked conversion"""
        a = np.ascontiguousarray(a)
        out = np.zeros(a.size, dtype=out_dtype)
        f = self.avx2.cast_type_numeric_avx2
        f.restype = None
        f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        f(TYPE_IDS[a.dtype], TYPE_IDS[np.dtype(out_dtype)], _p(a), _p(out), a.size)
        return out

    def min_max(self, a):
        """int64_max_min_avx2(values, len, &min, &max) … (internal/utils/_lib/min_max.c:23-126)"""
        a = np.ascontiguousarray(a)
        lo, hi = np.zeros(1, a.dtype), np.zeros(1, a.dtype)
        f = getattr(self.avx2, f"{a.dtype.name}_max_min_avx2")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        f(_p(a), a.size, _p(lo), _p(hi))
        return lo[0], hi[0]

    def arithmetic_unary(self, op, a):
        a = np.ascontiguousarray(a)
        out = np.zeros(a.size, dtype=a.dtype)
        f = self.avx2.arithmetic_unary_same_types_avx2
        f.restype = None
        f.argtypes = [C.c_int, C.c_int8, C.c_void_p, C.c_void_p, C.c_int]
        f(TYPE_IDS[a.dtype], op, _p(a), _p(out), a.size)
        return out

    def comparison(self, cmpop, shape, l, r, out_bits, out_bit_offset=0):
        l = np.ascontiguousarray(l); r = np.ascontiguousarray(r)
        arr = r if shape == 2 else l
        f = getattr(self.avx2, f"comparison_{self.CMP_NAMES[cmpop]}_{self.SHAPE_NAMES[shape]}_avx2")
        f.restype = None
        f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        f(TYPE_IDS[arr.dtype], _p(l), _p(r), _p(out_bits), arr.size, out_bit_offset)
        return out_bits

    def bitmap_aligned(self, op, l, r):
        l = np.ascontiguousarray(l); r = np.ascontiguousarray(r)
        out = np.zeros(l.size, np.uint8)
        f = getattr(self.avx2, {0: "bitmap_aligned_and_avx2", 1: "bitmap_aligned_or_avx2", 2: "bitmap_aligned_xor_avx2",
                                3: "bitmap_aligned_and_not_avx2"}[op])
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        f(_p(l), _p(r), _p(out), l.size)
        return out


def load_reference():
    """None when oracle/_ref was never built (no /root/reference and no prebuilt libs)."""
    a = os.path.join(ORACLE_DIR, "_ref", "libref_avx2.so")
    c = os.path.join(ORACLE_DIR, "_ref", "libref_c.so")
    if not (os.path.exists(a) and os.path.exists(c)):
        return None
    return Reference(C.CDLL(a), C.CDLL(c))
