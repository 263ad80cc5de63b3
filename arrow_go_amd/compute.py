"""Array-level access to the C++ host layer (libarrowhip_compute.so), which mirrors
arrow-go's `compute.CallFunction` / function registry / `arrow/math` API on top of the HIP
kernels.  Arrays cross the boundary through the Arrow C Data Interface, exactly the route
arrow-go's own `arrow/cdata` package provides; pyarrow is only the producer/consumer of
those structs in the tests.

    s = Session(0)
    s.call_function("add", [pa.array([3, 2, 6]), pa.array([1, 0, 2])])   # → [4, 2, 8]
    s.call_function("filter", [values, mask], "null_selection_behavior=emit_null")

Errors mirror arrow-go's sentinel errors (arrow.ErrInvalid, ErrIndex, ErrNotImplemented,
"function '…' not found").  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import struct

from . import _native  # loads libarrowhip.so first (RTLD_GLOBAL)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libarrowhip_compute.so")
if not os.path.exists(_LIB):
    raise ImportError(f"{_LIB} is missing — run __graft_entry__.build()")
lib = C.CDLL(_LIB)

_vp = C.c_void_p
lib.ahc_session_create.argtypes = [C.c_int, C.POINTER(_vp)]
lib.ahc_session_destroy.argtypes = [_vp]
lib.ahc_session_destroy.restype = None
lib.ahc_last_error.argtypes = [_vp]
lib.ahc_last_error.restype = C.c_char_p
lib.ahc_datum_release.argtypes = [_vp]
lib.ahc_datum_release.restype = None
lib.ahc_import.argtypes = [_vp, _vp, _vp, C.POINTER(_vp)]
lib.ahc_scalar.argtypes = [_vp, C.c_int, C.c_int, _vp, C.POINTER(_vp)]
lib.ahc_datum_info.argtypes = [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                               C.POINTER(C.c_int), _vp]
lib.ahc_call.argtypes = [_vp, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]
lib.ahc_export.argtypes = [_vp, _vp, _vp, _vp]
lib.ahc_import_host.argtypes = [_vp, _vp, _vp, C.POINTER(_vp)]
lib.ahc_datum_on_host.argtypes = [_vp]
lib.ahc_session_set_option.argtypes = [_vp, C.c_char_p, C.c_int64]
lib.ahc_math_sum.argtypes = [_vp, _vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
lib.ahc_has_function.argtypes = [C.c_char_p]
lib.ahc_function_num_kernels.argtypes = [C.c_char_p]
lib.ahc_registry_add_alias.argtypes = [_vp, C.c_char_p, C.c_char_p, C.c_int]
lib.ahc_import_device.argtypes = [_vp, _vp, _vp, C.POINTER(_vp)]
lib.ahc_export_device.argtypes = [_vp, _vp, _vp, _vp]
lib.ahc_datum_buffers.argtypes = [_vp, C.POINTER(_vp), C.POINTER(_vp)]
lib.ahc_chunked_from_arrays.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]
lib.ahc_datum_num_chunks.argtypes = [_vp]
lib.ahc_datum_chunk.argtypes = [_vp, _vp, C.c_int, C.POINTER(_vp)]
lib.ahc_record_from_arrays.argtypes = [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(_vp), C.POINTER(_vp)]
lib.ahc_record_num_columns.argtypes = [_vp]
lib.ahc_record_column.argtypes = [_vp, _vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(_vp)]
lib.ahc_ipc_open.argtypes = [_vp, _vp, C.c_int64, C.POINTER(_vp)]
lib.ahc_ipc_close.argtypes = [_vp]
lib.ahc_ipc_close.restype = None
lib.ahc_ipc_num_fields.argtypes = [_vp]
lib.ahc_ipc_field.argtypes = [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.ahc_ipc_next.argtypes = [_vp, C.POINTER(_vp), C.POINTER(C.c_int64)]
lib.ahc_datum_logical.argtypes = [_vp]
lib.ahc_datum_logical.restype = C.c_char_p
lib.ahc_scalar_set_logical.argtypes = [_vp, _vp, C.c_char_p]
lib.ahc_ipc_bytes_uploaded.argtypes = [_vp]
lib.ahc_ipc_bytes_uploaded.restype = C.c_int64
lib.ahc_ipc_inspect.argtypes = [_vp, C.c_int64, C.c_char_p, C.c_int64]
lib.ahc_substrait_inspect.argtypes = [_vp, C.c_int64, C.c_char_p, C.c_int64]
lib.ahc_dispatch_best.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int64]
lib.ahc_expr_eval.argtypes = [_vp, C.c_char_p, C.c_int, C.POINTER(_vp), C.c_int, C.POINTER(_vp), C.c_int, C.POINTER(_vp), C.POINTER(C.c_int)]


class _ExprNode(C.Structure):   # ahc_expr_node of include/arrowhip_compute.h
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32), ("name", C.c_char_p), ("options", C.c_char_p), ("nargs", C.c_int32), ("args", C.POINTER(C.c_int32))]


lib.ahc_expr_eval_tree.argtypes = [_vp, C.POINTER(_ExprNode), C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(C.c_char_p), C.c_int, C.POINTER(_vp), C.c_int,
                                   C.POINTER(_vp), C.POINTER(C.c_int)]
lib.ahc_expr_eval_substrait.argtypes = [_vp, C.c_void_p, C.c_int64, C.c_int, C.POINTER(_vp), C.POINTER(C.c_char_p), C.c_int, C.POINTER(_vp), C.POINTER(C.c_int)]


# ---- Arrow C (Device) Data Interface structs (arrow/cdata/abi.h) ------------------------------------
class CArrowSchema(C.Structure):
    pass


CArrowSchema._fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                         ("n_children", C.c_int64), ("children", _vp), ("dictionary", _vp),
                         ("release", C.CFUNCTYPE(None, C.POINTER(CArrowSchema))), ("private_data", _vp)]


class CArrowArray(C.Structure):
    pass


CArrowArray._fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                        ("n_children", C.c_int64), ("buffers", C.POINTER(_vp)), ("children", _vp), ("dictionary", _vp),
                        ("release", C.CFUNCTYPE(None, C.POINTER(CArrowArray))), ("private_data", _vp)]


class CArrowDeviceArray(C.Structure):
    _fields_ = [("array", CArrowArray), ("device_id", C.c_int64), ("device_type", C.c_int32), ("sync_event", _vp),
                ("reserved", C.c_int64 * 3)]


ARROW_DEVICE_CPU, ARROW_DEVICE_ROCM, ARROW_DEVICE_ROCM_HOST = 1, 10, 11
_FORMATS = {"bool": b"b", "int8": b"c", "uint8": b"C", "int16": b"s", "uint16": b"S", "int32": b"i", "uint32": b"I", "int64": b"l",
            "uint64": b"L", "float": b"f", "double": b"g", "string": b"u", "binary": b"z", "large_string": b"U", "large_binary": b"Z"}


class DeviceArray:
    """An array datum that stays in HBM (an ahc_datum handle).  Pass it to call_function like a
    pyarrow array; `to_arrow()` downloads, `export_device()` hands the buffers to another ROCm consumer
    through the Arrow C Device Data Interface without a copy."""

    def __init__(self, session, handle):
        self.session, self.h = session, handle

    def buffers(self):
        v, d = _vp(), _vp()
        lib.ahc_datum_buffers(self.h, C.byref(v), C.byref(d))
        return v.value, d.value

    def to_arrow(self):
        return self.session._export(self.h)

    def on_host(self) -> bool:
        """True for a HOST-resident array: imported with Session.import_host / import_host_buffers, or the result of a call
        that streamed such arguments through the device (include/arrowhip_compute.h, "HOST-RESIDENT arguments")"""
        return bool(lib.ahc_datum_on_host(self.h))

    def export_device(self):
        """→ (CArrowDeviceArray, CArrowSchema); the caller owns them and must call array.release."""
        darr, sch = CArrowDeviceArray(), CArrowSchema()
        self.session._check(lib.ahc_export_device(self.session.h, self.h, C.addressof(darr), C.addressof(sch)))
        return darr, sch

    def release(self):
        if self.h:
            lib.ahc_datum_release(self.h)
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class ArrowError(Exception):
    code = 0


class ErrInvalid(ArrowError):
    code = 1


class ErrIndex(ArrowError):
    code = 2


class ErrNotImplemented(ArrowError):
    code = 3


class ErrType(ArrowError):
    code = 4


class ErrKey(ArrowError):
    code = 5


class ErrHip(ArrowError):
    code = 6


_ERRS = {1: ErrInvalid, 2: ErrIndex, 3: ErrNotImplemented, 4: ErrType, 5: ErrKey, 6: ErrHip}

# arrow.Type ids
_TYPE_IDS = {"dictionary": 36, "string": 13, "binary": 14, "large_string": 34, "large_binary": 35, "bool": 1, "uint8": 2, "int8": 3, "uint16": 4, "int16": 5, "uint32": 6, "int32": 7, "uint64": 8, "int64": 9,
             "float": 11, "double": 12}
_PACK = {1: "<?", 2: "<B", 3: "<b", 4: "<H", 5: "<h", 6: "<I", 7: "<i", 8: "<Q", 9: "<q", 11: "<f", 12: "<d"}


_UNITS = {"s": "s", "m": "ms", "u": "us", "n": "ns"}


def _temporal_format(t):
    """pyarrow temporal type → its Arrow C Data format ("tsu:UTC", "tdD", "ttm", "tDn"), None for any other type"""
    import pyarrow as pa
    short = {v: k for k, v in _UNITS.items()}
    if pa.types.is_timestamp(t):
        return "ts%s:%s" % (short[t.unit], t.tz or "")
    if pa.types.is_duration(t):
        return "tD" + short[t.unit]
    if pa.types.is_date32(t):
        return "tdD"
    if pa.types.is_date64(t):
        return "tdm"
    if pa.types.is_time32(t) or pa.types.is_time64(t):
        return "tt" + short[t.unit]
    return None


def _temporal_type(fmt):
    """the inverse: C Data format → pyarrow type"""
    import pyarrow as pa
    if fmt == "tdD":
        return pa.date32()
    if fmt == "tdm":
        return pa.date64()
    unit = _UNITS[fmt[2]]
    if fmt[1] == "t":
        return pa.time32(unit) if unit in ("s", "ms") else pa.time64(unit)
    if fmt[1] == "D":
        return pa.duration(unit)
    if fmt[1] == "s":
        return pa.timestamp(unit, tz=fmt[4:] or None)
    raise ErrNotImplemented(f"temporal format {fmt!r}")


def _as_bytes_ptr(buf):
    """(keep-alive object, address, length) of a bytes-like / pyarrow.Buffer without copying when possible"""
    mv = memoryview(buf)
    if mv.readonly:
        keep = (C.c_char * mv.nbytes).from_buffer_copy(mv) if not isinstance(buf, bytes) else buf
        addr = C.cast(C.c_char_p(keep), _vp).value if isinstance(keep, bytes) else C.addressof(keep)
        return keep, addr, mv.nbytes
    keep = (C.c_char * mv.nbytes).from_buffer(mv)
    return keep, C.addressof(keep), mv.nbytes



def dispatch_best(function: str, types):
    """fn.DispatchBest(types...) without executing (ahc_dispatch_best): pyarrow types in → the argument types of the chosen kernel
    as pyarrow types; raises the reference's error class (ErrNotImplemented "has no kernel matching input types", KeyError …)."""
    ids = {"bool": 1, "uint8": 2, "int8": 3, "uint16": 4, "int16": 5, "uint32": 6, "int32": 7, "uint64": 8, "int64": 9, "float": 11, "double": 12,
           "string": 13, "binary": 14, "large_string": 34, "large_binary": 35}
    back = {v: k for k, v in ids.items()}
    n = len(types)
    tin = (C.c_int * n)(*[ids[str(t)] for t in types])
    tout = (C.c_int * n)()
    err = C.create_string_buffer(1024)
    rc = lib.ahc_dispatch_best(function.encode(), n, tin, tout, err, len(err))
    if rc != 0:
        raise _ERRS.get(rc, ArrowError)(err.value.decode(errors="replace"))
    import pyarrow as pa
    named = {"float": pa.float32(), "double": pa.float64(), "bool": pa.bool_()}
    return [named.get(back[i], None) or getattr(pa, back[i])() for i in tout]

def inspect_substrait(buf) -> str:
    """What the Substrait reader understood of a serialized ExtendedExpression, without a device (ahc_substrait_inspect):
    "name:type,…|output_name=expression|…"; raises the reference's error class for a message the reader refuses as a whole."""
    keep, addr, n = _as_bytes_ptr(buf)
    out = C.create_string_buffer(1 << 16)
    rc = lib.ahc_substrait_inspect(addr, n, out, len(out))
    text = out.value.decode(errors="replace")
    if rc != 0:
        raise _ERRS.get(rc, ArrowError)(text)
    return text


def ipc_inspect(buf):
    """Walk an Arrow IPC stream on the host only (no GPU needed): → ([(name, type name, nullable)], [rows per batch]).
    Raises the same errors read_ipc would."""
    keep, addr, n = _as_bytes_ptr(buf)
    out = C.create_string_buffer(1 << 20)
    rc = lib.ahc_ipc_inspect(addr, n, out, len(out))
    text = out.value.decode()
    if rc != 0:
        raise _ERRS.get(rc, ArrowError)(text)
    names = {v: k for k, v in _TYPE_IDS.items()}
    fields, rows = text.split("|")
    fl = []
    for f in filter(None, fields.split(",")):
        nm, tid, nullable, logical = f.rsplit(":", 3)
        tname = str(_temporal_type(bytes.fromhex(logical).decode())) if logical and int(tid) != 36 else names[int(tid)]
        fl.append((bytes.fromhex(nm).decode("utf-8", "replace"), tname, nullable == "1"))
    return fl, [int(r) for r in rows.split(",") if r]


def has_function(name: str) -> bool:
    return bool(lib.ahc_has_function(name.encode()))


def function_num_kernels(name: str) -> int:
    return int(lib.ahc_function_num_kernels(name.encode()))


def num_functions() -> int:
    return int(lib.ahc_num_functions())


class Session:
    """A device session: ah_ctx + ExecCtx with its own child registry."""

    def __init__(self, device_id: int = 0):
        h = _vp()
        rc = lib.ahc_session_create(device_id, C.byref(h))
        if rc != 0:
            raise ErrHip("ahc_session_create failed: is a GPU visible? (no CPU fallback)")
        self.h = h
        self.device_id = device_id

    def close(self):
        if self.h:
            lib.ahc_session_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise _ERRS.get(rc, ArrowError)(lib.ahc_last_error(self.h).decode())

    # -- datum plumbing
    def _import(self, arr):
        import pyarrow as pa
        if isinstance(arr, pa.ChunkedArray):  # compute.ChunkedDatum: every chunk uploaded, chunk boundaries kept
            tname = str(arr.type)
            if pa.types.is_dictionary(arr.type):
                tid = 36  # arrow.DICTIONARY
            elif _temporal_format(arr.type):  # stored as integers of the type's width, labelled chunk by chunk
                tid = _TYPE_IDS["int32" if arr.type.bit_width == 32 else "int64"]
            elif tname in _TYPE_IDS:
                tid = _TYPE_IDS[tname]
            else:
                raise ErrNotImplemented(f"unsupported chunked type {tname}")
            parts = [self._import(c) for c in arr.chunks]
            try:
                d = _vp()
                handles = (_vp * max(len(parts), 1))(*parts)
                self._check(lib.ahc_chunked_from_arrays(self.h, tid, len(parts), handles, C.byref(d)))
                return d
            finally:
                for p in parts:
                    lib.ahc_datum_release(p)
        a = (C.c_uint8 * 80)()
        s = (C.c_uint8 * 72)()
        arr._export_to_c(C.addressof(a), C.addressof(s))
        d = _vp()
        self._check(lib.ahc_import(self.h, C.addressof(a), C.addressof(s), C.byref(d)))
        return d

    def _scalar(self, sc):
        import pyarrow as pa
        logical = _temporal_format(sc.type)
        tid = _TYPE_IDS[("int32" if sc.type.bit_width == 32 else "int64") if logical else str(sc.type)]
        valid = sc.is_valid
        buf = (C.c_uint8 * 8)()
        if valid:
            raw = struct.pack(_PACK[tid], sc.value if logical else sc.as_py())
            C.memmove(buf, raw, len(raw))
        d = _vp()
        self._check(lib.ahc_scalar(self.h, tid, int(valid), buf, C.byref(d)))
        if logical:
            self._check(lib.ahc_scalar_set_logical(self.h, d, logical.encode()))
        return d

    def import_device(self, type_name: str, length: int, data_ptr, validity_ptr=None, null_count: int = 0, offset: int = 0,
                      sync_event=None, on_release=None, device_type: int = ARROW_DEVICE_ROCM, device_id=None, var_data_ptr=None,
                      n_buffers=None) -> "DeviceArray":
        """Consumer side of the Arrow C Device Data Interface (arrow/cdata/abi.h:66-128): wrap device
        buffers produced elsewhere WITHOUT copying them.  `on_release` is called when the library lets
        go of the producer's array (its release callback).  String / binary types: `data_ptr` is the offsets
        buffer, `var_data_ptr` the bytes (three buffers)."""
        fmt = _FORMATS[type_name]
        nb = n_buffers if n_buffers is not None else (3 if fmt in (b"u", b"z", b"U", b"Z") else 2)
        bufs = (_vp * 3)(validity_ptr, data_ptr, var_data_ptr)
        keep = {"bufs": bufs, "fmt": fmt}

        def _release_array(ptr):
            ptr.contents.release = C.cast(None, CArrowArray._fields_[8][1])
            if on_release:
                on_release()
            keep.clear()

        def _release_schema(ptr):
            ptr.contents.release = C.cast(None, CArrowSchema._fields_[7][1])

        rel_a = CArrowArray._fields_[8][1](_release_array)
        rel_s = CArrowSchema._fields_[7][1](_release_schema)
        keep["cb"] = (rel_a, rel_s)
        self._device_imports = getattr(self, "_device_imports", [])
        self._device_imports.append(keep)  # callbacks must outlive the C side's use of them
        darr = CArrowDeviceArray()
        darr.array.length, darr.array.null_count, darr.array.offset = length, null_count, offset
        darr.array.n_buffers, darr.array.n_children = nb, 0
        darr.array.buffers = C.cast(bufs, C.POINTER(_vp))
        darr.array.release = rel_a
        darr.device_id = self.device_id if device_id is None else device_id
        darr.device_type = device_type
        darr.sync_event = sync_event
        sch = CArrowSchema()
        sch.format, sch.name, sch.flags = fmt, b"", 2
        sch.release = rel_s
        d = _vp()
        self._check(lib.ahc_import_device(self.h, C.addressof(darr), C.addressof(sch), C.byref(d)))
        return DeviceArray(self, d)

    def set_option(self, name: str, value: int) -> None:
        """ExecCtx fields of this session: 'chunk_bytes' (ExecCtx.ChunkSize's role for host-resident arguments), 'host_threshold_bytes'."""
        self._check(lib.ahc_session_set_option(self.h, name.encode(), int(value)))

    def import_host(self, arr) -> "DeviceArray":
        """ahc_import_host of a pyarrow array: a flat fixed-width column of at least host_threshold_bytes stays in pyarrow's buffers
        (pageable memory: correct, but the copies do not overlap — use import_host_buffers with pinned memory for that)"""
        a = (C.c_uint8 * 80)()
        s = (C.c_uint8 * 72)()
        arr._export_to_c(C.addressof(a), C.addressof(s))
        d = _vp()
        self._check(lib.ahc_import_host(self.h, C.addressof(a), C.addressof(s), C.byref(d)))
        return DeviceArray(self, d)

    def import_host_buffers(self, type_name: str, length: int, data_ptr, validity_ptr=None, null_count: int = 0, offset: int = 0,
                            on_release=None) -> "DeviceArray":
        """ahc_import_host of raw host buffers (e.g. ah.Context.alloc_pinned memory): they stay where they are and must outlive the
        returned array; `on_release` is called when the library lets go of them"""
        fmt = _FORMATS[type_name]
        bufs = (_vp * 2)(validity_ptr, data_ptr)
        keep = {"bufs": bufs, "fmt": fmt}

        def _release_array(ptr):
            ptr.contents.release = C.cast(None, CArrowArray._fields_[8][1])
            if on_release:
                on_release()

        def _release_schema(ptr):
            ptr.contents.release = C.cast(None, CArrowSchema._fields_[7][1])

        rel_a = CArrowArray._fields_[8][1](_release_array)
        rel_s = CArrowSchema._fields_[7][1](_release_schema)
        keep["cb"] = (rel_a, rel_s)
        self._device_imports = getattr(self, "_device_imports", [])
        self._device_imports.append(keep)  # callbacks must outlive the C side's use of them
        arr = CArrowArray()
        arr.length, arr.null_count, arr.offset = length, null_count, offset
        arr.n_buffers, arr.n_children = 2, 0
        arr.buffers = C.cast(bufs, C.POINTER(_vp))
        arr.release = rel_a
        sch = CArrowSchema()
        sch.format, sch.name, sch.flags = fmt, b"", 2
        sch.release = rel_s
        d = _vp()
        self._check(lib.ahc_import_host(self.h, C.addressof(arr), C.addressof(sch), C.byref(d)))
        return DeviceArray(self, d)

    def _to_datum(self, x):
        import pyarrow as pa
        if isinstance(x, DeviceArray):
            return x.h
        if isinstance(x, (pa.Array, pa.ChunkedArray)):
            return self._import(x)
        if isinstance(x, pa.RecordBatch):  # compute.RecordDatum
            parts = [self._import(c) for c in x.columns]
            try:
                d = _vp()
                names = (C.c_char_p * max(len(parts), 1))(*[n.encode() for n in x.schema.names])
                handles = (_vp * max(len(parts), 1))(*parts)
                self._check(lib.ahc_record_from_arrays(self.h, len(parts), names, handles, C.byref(d)))
                return d
            finally:
                for p in parts:
                    lib.ahc_datum_release(p)
        if isinstance(x, pa.Scalar):
            return self._scalar(x)
        raise TypeError(f"unsupported argument {type(x)}: pass a pyarrow Array, ChunkedArray, RecordBatch or Scalar")

    def _export(self, d):
        import pyarrow as pa
        kind, tid, length, nulls, sv = C.c_int(), C.c_int(), C.c_int64(), C.c_int64(), C.c_int()
        val = (C.c_uint8 * 8)()
        lib.ahc_datum_info(d, C.byref(kind), C.byref(tid), C.byref(length), C.byref(nulls), C.byref(sv), val)
        if kind.value == 1:  # scalar
            name = {v: k for k, v in _TYPE_IDS.items()}[tid.value]
            typ = {"float": pa.float32(), "double": pa.float64(), "bool": pa.bool_()}.get(name) or getattr(pa, name)()
            logical = lib.ahc_datum_logical(d).decode()
            if not sv.value:
                return pa.scalar(None, type=_temporal_type(logical) if logical else typ)
            fmt = _PACK[tid.value]
            plain = pa.scalar(struct.unpack(fmt, bytes(val)[:struct.calcsize(fmt)])[0], type=typ)
            return plain.cast(_temporal_type(logical)) if logical else plain
        if kind.value == 4:  # record batch
            cols, names = [], []
            for i in range(lib.ahc_record_num_columns(d)):
                c, nm = _vp(), C.c_char_p()
                self._check(lib.ahc_record_column(self.h, d, i, C.byref(nm), C.byref(c)))
                names.append(nm.value.decode())
                try:
                    cols.append(self._export(c))
                finally:
                    lib.ahc_datum_release(c)
            return pa.RecordBatch.from_arrays(cols, names=names)
        if kind.value == 3:  # chunked
            parts = []
            for i in range(lib.ahc_datum_num_chunks(d)):
                c = _vp()
                self._check(lib.ahc_datum_chunk(self.h, d, i, C.byref(c)))
                try:
                    parts.append(self._export(c))
                finally:
                    lib.ahc_datum_release(c)
            if parts:
                return pa.chunked_array(parts)
            name = {v: k for k, v in _TYPE_IDS.items()}[tid.value]
            typ = {"float": pa.float32(), "double": pa.float64(), "bool": pa.bool_()}.get(name) or getattr(pa, name)()
            return pa.chunked_array([], type=typ)
        a = (C.c_uint8 * 80)()
        s = (C.c_uint8 * 72)()
        self._check(lib.ahc_export(self.h, d, C.addressof(a), C.addressof(s)))
        return pa.Array._import_from_c(C.addressof(a), C.addressof(s))

    # -- compute.CallFunction
    def call_function(self, name: str, args, options: str = "", value_set=None, keep_on_device: bool = False):
        """options: "key=value;…" (Go struct tags).  value_set: the SetOptions.ValueSet array of is_in.
        keep_on_device: return a DeviceArray (no download) — chain calls without leaving HBM."""
        datums = [self._to_datum(x) for x in args]
        borrowed_handles = [x.h.value for x in list(args) + ([value_set] if value_set is not None else []) if isinstance(x, DeviceArray)]
        if value_set is not None:
            vs = self._to_datum(value_set)
            datums.append(vs)  # released with the arguments
            options = (options + ";" if options else "") + "value_set=@%x" % vs.value
            nargs = len(datums) - 1
        else:
            nargs = len(datums)
        try:
            arr = (_vp * len(datums))(*datums)
            out = _vp()
            self._check(lib.ahc_call(self.h, name.encode(), options.encode(), nargs, arr, C.byref(out)))
            if keep_on_device:
                return DeviceArray(self, out)
            try:
                return self._export(out)
            finally:
                lib.ahc_datum_release(out)
        finally:
            for d in datums:
                if d.value not in borrowed_handles:  # DeviceArray arguments stay owned by their wrappers
                    lib.ahc_datum_release(d)

    # -- ipc.NewReader / Reader.Next, with the record-batch bodies landing in HBM
    def read_ipc(self, buf):
        """Iterate the record batches of an Arrow IPC *stream* (bytes / pyarrow.Buffer / mmap): yields
        (field names, [DeviceArray per column], rows).  Each body goes to the device in one copy; the columns
        are slices of it."""
        keep, addr, n = _as_bytes_ptr(buf)
        r = _vp()
        self._check(lib.ahc_ipc_open(self.h, addr, n, C.byref(r)))
        try:
            nf = lib.ahc_ipc_num_fields(r)
            names = []
            for i in range(nf):
                nm = C.c_char_p()
                self._check(lib.ahc_ipc_field(r, i, C.byref(nm), None, None))
                names.append(nm.value.decode())
            while True:
                cols = (_vp * max(nf, 1))()
                rows = C.c_int64()
                self._check(lib.ahc_ipc_next(r, cols, C.byref(rows)))
                if rows.value < 0:
                    return
                yield names, [DeviceArray(self, _vp(cols[i])) for i in range(nf)], rows.value
        finally:
            lib.ahc_ipc_close(r)
            del keep

    # -- compute.Expression / exprs.ExecuteScalarExpression
    def eval_expression(self, text: str, columns, literals=(), fuse: bool = True, raw: bool = False):
        """Evaluate an expression tree over a batch.  `text` is prefix notation with `$i` for
        column i and `#k` for literal k, e.g. "greater(multiply_unchecked(add($0,$1),$2),#0)".
        fuse=True → one JIT-compiled kernel when the tree is fusible; fuse=False → one kernel per
        call node (the reference's executeScalarBatch).  Returns (result, fused)."""
        cols = [self._to_datum(c) for c in columns]
        lits = [self._to_datum(l) for l in literals]
        borrowed = [x.h.value for x in list(columns) + list(literals) if isinstance(x, DeviceArray)]   # stay owned by their wrappers
        try:
            ca = (_vp * max(len(cols), 1))(*cols)
            la = (_vp * max(len(lits), 1))(*lits)
            out = _vp()
            fused = C.c_int()
            self._check(lib.ahc_expr_eval(self.h, text.encode(), len(cols), ca, len(lits), la, int(fuse), C.byref(out), C.byref(fused)))
            try:
                return self._export(out), bool(fused.value)
            finally:
                lib.ahc_datum_release(out)
        finally:
            for d in cols + lits:
                if d.value not in borrowed:
                    lib.ahc_datum_release(d)

    def eval_expression_tree(self, tree, columns, names=None, fuse: bool = True):
        """The reference's own tree shape (compute.Expression: Literal | field reference | Call{name, args, options},
        arrow/compute/expression.go:52-78, 278-290) through ahc_expr_eval_tree.  tree: ("call", name, [subtrees], options text or None) |
        ("field", position or name) | ("lit", pyarrow scalar).  Returns (result, fused)."""
        nodes, lits, keep = [], [], []

        def walk(t):
            if t[0] == "lit":
                lits.append(t[1])
                nodes.append(_ExprNode(0, len(lits) - 1, None, None, 0, None))
            elif t[0] == "field":
                by_name = isinstance(t[1], str)
                nodes.append(_ExprNode(1, -1 if by_name else int(t[1]), t[1].encode() if by_name else None, None, 0, None))
            elif t[0] == "call":
                pos = [walk(a) for a in t[2]]
                arr = (C.c_int32 * max(len(pos), 1))(*pos)
                keep.append(arr)
                opts = t[3].encode() if len(t) > 3 and t[3] else None
                nodes.append(_ExprNode(2, 0, t[1].encode(), opts, len(pos), C.cast(arr, C.POINTER(C.c_int32))))
            else:
                raise ValueError(f"expression node kind {t[0]!r}")
            return len(nodes) - 1

        walk(tree)
        cols = [self._to_datum(c) for c in columns]
        ld = [self._to_datum(l) for l in lits]
        try:
            na = (_ExprNode * len(nodes))(*nodes)
            ca = (_vp * max(len(cols), 1))(*cols)
            la = (_vp * max(len(ld), 1))(*ld)
            nm = (C.c_char_p * max(len(cols), 1))(*[n.encode() for n in names]) if names is not None else None
            out, fused = _vp(), C.c_int()
            self._check(lib.ahc_expr_eval_tree(self.h, na, len(nodes), len(cols), ca, nm, len(ld), la, int(fuse), C.byref(out), C.byref(fused)))
            try:
                return self._export(out), bool(fused.value)
            finally:
                lib.ahc_datum_release(out)
        finally:
            for d in cols + ld:
                lib.ahc_datum_release(d)

    def eval_substrait(self, message: bytes, columns, names=None, fuse: bool = True):
        """exprs.ExecuteScalarSubstrait (arrow/compute/exprs/exec.go:465-488): `message` is a serialized substrait.ExtendedExpression
        with one referred expression; columns (pyarrow arrays / scalars / DeviceArrays) in base-schema order, or matched by `names`.
        Returns (result, fused)."""
        message = bytes(message)
        cols = [self._to_datum(c) for c in columns]
        try:
            ca = (_vp * max(len(cols), 1))(*cols)
            nm = (C.c_char_p * max(len(cols), 1))(*[n.encode() for n in names]) if names is not None else None
            buf = C.create_string_buffer(message, len(message))
            out, fused = _vp(), C.c_int()
            self._check(lib.ahc_expr_eval_substrait(self.h, C.cast(buf, C.c_void_p), len(message), len(cols), ca, nm, int(fuse), C.byref(out), C.byref(fused)))
            try:
                return self._export(out), bool(fused.value)
            finally:
                lib.ahc_datum_release(out)
        finally:
            for d in cols:
                lib.ahc_datum_release(d)

    # -- arrow/math
    def math_sum(self, arr):
        """math.Float64.Sum / Int64.Sum / Uint64.Sum of a pyarrow array or a DeviceArray"""
        resident = isinstance(arr, DeviceArray)
        d = arr.h if resident else self._import(arr)
        try:
            f, i, u = C.c_double(), C.c_int64(), C.c_uint64()
            self._check(lib.ahc_math_sum(self.h, d, C.byref(f), C.byref(i), C.byref(u)))
            if resident:
                kind, tid, length, nulls, sv = C.c_int(), C.c_int(), C.c_int64(), C.c_int64(), C.c_int()
                lib.ahc_datum_info(d, C.byref(kind), C.byref(tid), C.byref(length), C.byref(nulls), C.byref(sv), None)
                t = {v: k for k, v in _TYPE_IDS.items()}[tid.value]
            else:
                t = str(arr.type)
            return f.value if t == "double" else (i.value if t == "int64" else u.value)
        finally:
            if not resident:
                lib.ahc_datum_release(d)

    def add_alias(self, alias: str, existing: str, allow_overwrite: bool = False):
        self._check(lib.ahc_registry_add_alias(self.h, alias.encode(), existing.encode(), int(allow_overwrite)))
