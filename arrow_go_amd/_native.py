"""ctypes binding of libarrowhip.so (the C ABI in include/arrowhip.h).

This is plumbing: it declares the C signatures and turns status codes into Python
exceptions named after the arrow-go error values a Go shim would wrap
(arrow.ErrInvalid / ErrIndex / ErrNotImplemented — arrow/errors.go).  There is NO
fallback: if the HIP library is missing or a symbol the header declares is not
exported, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARROWHIP_LIB") or os.path.join(_HERE, "libarrowhip.so")  # env: kernel-variant experiments only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "arrowhip.h")

AH_OK, AH_EINVALID, AH_EINDEX, AH_EOVERFLOW, AH_EHIP, AH_ENOTIMPL = 0, 1, 2, 3, 4, 5

# arrow.Type ids (arrow/datatype.go:36-72)
UINT8, INT8, UINT16, INT16, UINT32, INT32, UINT64, INT64, FLOAT32, FLOAT64 = 2, 3, 4, 5, 6, 7, 8, 9, 11, 12
OP_ADD, OP_SUB, OP_MUL, OP_ABS, OP_NEGATE, OP_SIGN = 0, 1, 2, 4, 5, 20
OP_ADD_CHECKED, OP_SUB_CHECKED, OP_MUL_CHECKED = 21, 22, 23
CMP_EQ, CMP_NE, CMP_GT, CMP_GE = 0, 1, 2, 3
SHAPE_AA, SHAPE_AS, SHAPE_SA = 0, 1, 2
BIT_AND, BIT_OR, BIT_XOR, BIT_AND_NOT, BIT_XNOR = 0, 1, 2, 3, 4
KLEENE_AND, KLEENE_OR, KLEENE_AND_NOT = 0, 1, 2
DROP_NULLS, EMIT_NULLS = 0, 1
BOOL = 1
X_FIELD, X_LITERAL = 1, 2
X_ADD, X_SUB, X_MUL, X_ADD_CHECKED, X_SUB_CHECKED, X_MUL_CHECKED = 10, 11, 12, 13, 14, 15
X_NEGATE, X_ABS, X_SIGN = 20, 21, 22
X_EQ, X_NE, X_GT, X_GE, X_LT, X_LE = 30, 31, 32, 33, 34, 35
X_AND, X_OR, X_XOR, X_AND_NOT, X_INVERT = 40, 41, 42, 43, 44
X_CAST = 50
OP_DIV, OP_SQRT, OP_DIV_CHECKED, OP_ABS_CHECKED, OP_NEGATE_CHECKED, OP_SQRT_CHECKED = 3, 6, 24, 25, 26, 27
OP_POWER, OP_POWER_CHECKED = 7, 28
OP_SHIFT_LEFT, OP_SHIFT_LEFT_CHECKED, OP_SHIFT_RIGHT, OP_SHIFT_RIGHT_CHECKED, OP_BIT_AND, OP_BIT_OR, OP_BIT_XOR, OP_BIT_NOT = 64, 65, 66, 67, 68, 69, 70, 71
OP_FLOOR, OP_CEIL, OP_TRUNC = 72, 73, 74


class ArrowHipError(Exception):
    """Base class; .status is the C status code."""

    status = -1


class ErrInvalid(ArrowHipError):  # arrow.ErrInvalid
    status = AH_EINVALID


class ErrIndex(ArrowHipError):  # arrow.ErrIndex
    status = AH_EINDEX


class ErrOverflow(ErrInvalid):  # arrow.ErrInvalid: "overflow"
    status = AH_EOVERFLOW


class ErrHip(ArrowHipError):
    status = AH_EHIP


class ErrNotImplemented(ArrowHipError):  # arrow.ErrNotImplemented
    status = AH_ENOTIMPL


_ERRS = {AH_EINVALID: ErrInvalid, AH_EINDEX: ErrIndex, AH_EOVERFLOW: ErrOverflow, AH_EHIP: ErrHip,
         AH_ENOTIMPL: ErrNotImplemented}


def declared_symbols(header_path: str = HEADER_PATH) -> list[str]:
    """Every function name include/arrowhip.h declares."""
    with open(header_path) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(ah_[a-z0-9_]+)\s*\(", text)))


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"libarrowhip.so does not export symbols declared in arrowhip.h: {missing}")
    return lib


lib = _load()

_vp, _i64, _i32, _int, _i8, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_int8, C.c_size_t
_pi64, _pvp, _pf, _pd, _pi32, _pint = (C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_float),
                                       C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int))

_SIGS = {
    "ah_ctx_create": [_int, _pvp],
    "ah_ctx_create_on_stream": [_int, _vp, _pvp],
    "ah_device_count": [_pint],
    "ah_ctx_set_option": [_vp, C.c_char_p, _i64],
    "ah_buf_alloc": [_vp, _sz, _pvp],
    "ah_buf_free": [_vp, _vp],
    "ah_host_alloc_pinned": [_vp, _sz, _pvp],
    "ah_host_free_pinned": [_vp, _vp],
    "ah_upload_async": [_vp, _vp, _vp, _sz],
    "ah_download_async": [_vp, _vp, _vp, _sz],
    "ah_memset_async": [_vp, _vp, _int, _sz],
    "ah_copy_async": [_vp, _vp, _vp, _sz],
    "ah_sync": [_vp],
    "ah_timer_start": [_vp],
    "ah_timer_stop": [_vp, _pf],
    "ah_event_record": [_vp, _int],
    "ah_event_elapsed_ms": [_vp, _int, _int, _pf],
    "ah_sum_float64": [_vp, _vp, _sz, _pd],
    "ah_sum_int64": [_vp, _vp, _sz, _pi64],
    "ah_sum_uint64": [_vp, _vp, _sz, C.POINTER(C.c_uint64)],
    "ah_sum_float64_dev": [_vp, _vp, _sz, _vp],
    "ah_sum_int64_dev": [_vp, _vp, _sz, _vp],
    "ah_arithmetic_binary": [_vp, _int, _i8, _vp, _vp, _vp, _i64],
    "ah_arithmetic_arr_scalar": [_vp, _int, _i8, _vp, _vp, _vp, _i64],
    "ah_arithmetic_scalar_arr": [_vp, _int, _i8, _vp, _vp, _vp, _i64],
    "ah_arithmetic_unary": [_vp, _int, _i8, _vp, _vp, _i64],
    "ah_arithmetic_checked": [_vp, _int, _i8, _int, _vp, _vp, _i64, _vp, _vp, _i64, _int, _vp, _i64],
    "ah_round": [_vp, _int, _vp, _vp, _i64, _i64, _i64, _int, _vp, C.c_double, _vp],
    "ah_arithmetic_ext": [_vp, _int, _int, _int, _vp, _vp, _i64, _vp, _vp, _i64, _int, _vp, _i64],
    "ah_comparison": [_vp, _int, _int, _int, _vp, _vp, _vp, _i64, _int],
    "ah_bitmap_op": [_vp, _int, _vp, _i64, _vp, _i64, _vp, _i64, _i64],
    "ah_count_set_bits": [_vp, _vp, _i64, _i64, _pi64],
    "ah_copy_bitmap": [_vp, _vp, _i64, _i64, _vp, _i64, _int],
    "ah_set_bits_to": [_vp, _vp, _i64, _i64, _int],
    "ah_kleene": [_vp, _int, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i64],
    "ah_filter_count": [_vp, _vp, _vp, _i64, _i64, _int, _pi64],
    "ah_filter_primitive": [_vp, _int, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _int, _i64, _vp, _vp, _pi64],
    "ah_filter_primitive_dev": [_vp, _int, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp],
    "ah_filter_primitive_once": [_vp, _int, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _int, _vp, _vp, _pi64, _pi64],
    "ah_take_primitive_dev": [_vp, _int, _vp, _vp, _i64, _i64, _int, _int, _vp, _vp, _i64, _i64, _vp, _vp, _vp],
    "ah_filter_to_indices": [_vp, _vp, _vp, _i64, _i64, _int, _i64, _vp, _vp, _pi64],
    "ah_take_primitive": [_vp, _int, _vp, _vp, _i64, _i64, _int, _int, _vp, _vp, _i64, _i64, _int, _vp, _vp, _pi64, _pi64],
    "ah_cumulative_sum": [_vp, _int, _vp, _vp, _i64, _i64, _vp, _int, _int, _vp, _vp, _pi64],
    "ah_cast_numeric": [_vp, _int, _int, _vp, _vp, _i64, _i64, _int, _int, _vp],
    "ah_shift_time": [_vp, _int, _int, _int, _i64, _int, _vp, _vp, _i64, _i64, _vp, _vp],
    "ah_cast_bool_to_numeric": [_vp, _int, _vp, _i64, _i64, _vp],
    "ah_is_in": [_vp, _int, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _int, _vp, _vp, _i64],
    "ah_sort_indices": [_vp, _int, _vp, _vp, _i64, _i64, _int, _int, _vp],
    "ah_wait_event": [_vp, _vp],
    "ah_device_id": [_vp],
    "ah_min_max": [_vp, _int, _vp, _i64, _vp, _vp],
    "ah_hash_partition_u64": [_vp, _vp, _i64, _int, _vp],
    "ah_sort_indices_multi": [_vp, _int, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp],
    "ah_take_binary_offsets": [_vp, _int, _vp, _vp, _i64, _i64, _int, _int, _vp, _vp, _i64, _i64, _int, _vp, _vp, _pi64, _pi64, _pi64],
    "ah_take_binary_data": [_vp, _int, _vp, _vp, _i64, _int, _vp, _i64, _vp, _vp],
    "ah_take_boolean": [_vp, _vp, _vp, _i64, _i64, _int, _int, _vp, _vp, _i64, _i64, _int, _vp, _vp, _pi64, _pi64],
    "ah_hash_u64_encode": [_vp, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _pi64, _pi32],
    "ah_hash_binary_encode": [_vp, _int, _vp, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _pi64, _pi32],
    "ah_hash_fixed_encode": [_vp, _int, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _pi64, _pi32],
    "ah_hash_sum_f64": [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _pi64, _pi32],
    "ah_hash_sum_i64": [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _pi64, _pi32],
    "ah_cmp_filter_sum_i64": [_vp, _int, _vp, _vp, _i64, _i64, _i64, _pi64, _pi64],
    "ah_cmp_filter_sum_f64": [_vp, _int, _vp, _vp, _i64, _i64, C.c_double, _pd, _pi64],
    "ah_cmp_filter_sum_i64_dev": [_vp, _int, _vp, _vp, _i64, _i64, _i64, _vp],
    "ah_cmp_filter_sum_f64_dev": [_vp, _int, _vp, _vp, _i64, _i64, C.c_double, _vp, _vp],
}
_SIGS.update({
    "ah_comm_unique_id": [_vp],
    "ah_comm_init": [_vp, _int, _int, _vp, _pvp],
    "ah_comm_destroy": [_vp],
    "ah_comm_rank": [_vp],
    "ah_comm_world": [_vp],
    "ah_comm_allreduce_sum": [_vp, _int, _vp, _vp, _i64],
    "ah_comm_allgather": [_vp, _vp, _vp, _i64],
    "ah_comm_alltoallv": [_vp, _vp, _pi64, _pi64, _vp, _pi64, _pi64],
    "ah_expr_compile": [_vp, _vp, _int, _vp, _int, _vp, _int, _pvp, _pint],
    "ah_expr_execute": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp],
    "ah_expr_codegen": [_vp, _int, _vp, _int, _vp, _int, _int, C.c_char_p, _sz, C.c_char_p, _sz, _pint],
})
TRANSPORT_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
TRANSPORT_ALLTOALLV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p,
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64))


class AhTransport(C.Structure):   # struct ah_transport (include/arrowhip.h)
    _fields_ = [("user", C.c_void_p), ("allgather", TRANSPORT_ALLGATHER), ("alltoallv", TRANSPORT_ALLTOALLV)]


_SIGS.update({
    "ah_comm_init_transport": [_vp, _int, _int, C.POINTER(AhTransport), _pvp],
    "ah_comm_cmp_filter_sum_i64": [_vp, _int, _vp, _vp, _i64, _i64, _i64, _pi64, _pi64],
    "ah_comm_cmp_filter_sum_f64": [_vp, _int, _vp, _vp, _i64, _i64, C.c_double, _pd, _pi64],
    "ah_comm_merge_groups": [_vp, _int, _vp, _vp, _vp, _vp, _i64, C.c_int32, _i64, _i64, _vp, _vp, _vp, _vp, _pi64, _pi32],
    "ah_graph_begin": [_vp],
    "ah_graph_end": [_vp, _pvp],
    "ah_graph_launch": [_vp, _vp],
    "ah_graph_destroy": [_vp],
    "ah_host_register": [_vp, _vp, _sz],
    "ah_host_unregister": [_vp, _vp],
    "ah_ingest_create": [_vp, _sz, _int, _pvp],
    "ah_ingest_destroy": [_vp],
    "ah_ingest_sum_float64": [_vp, _vp, _sz, _pd],
    "ah_ingest_sum_int64": [_vp, _vp, _sz, _pi64],
    "ah_ingest_arithmetic_binary": [_vp, _int, _i8, _vp, _vp, _vp, _i64],
    "ah_ingest_filter_count": [_vp, _vp, _vp, _i64, _i64, _int, _pi64],
    "ah_ingest_filter_primitive": [_vp, _int, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _pi64],
    "ah_ingest_depth": [_vp],
    "ah_ingest_slot_upload": [_vp, _int, _int, _sz, _vp, _sz, _int],
    "ah_ingest_slot_ready": [_vp, _int, _int],
    "ah_ingest_slot_release": [_vp, _int],
    "ah_ingest_slot_download": [_vp, _int, _int, _sz, _vp, _sz],
    "ah_ingest_wait": [_vp],
})
for _name, _args in _SIGS.items():
    _fn = getattr(lib, _name)
    _fn.argtypes = _args
    _fn.restype = _int
lib.ah_ctx_destroy.argtypes = [_vp]
lib.ah_ctx_destroy.restype = None
lib.ah_last_error.argtypes = [_vp]
lib.ah_last_error.restype = C.c_char_p
lib.ah_expr_source.argtypes = [_vp]
lib.ah_expr_source.restype = C.c_char_p
lib.ah_ingest_chunk_bytes.argtypes = [_vp]
lib.ah_ingest_chunk_bytes.restype = _sz
lib.ah_ingest_slot_buffer.argtypes = [_vp, _int, _int]
lib.ah_ingest_slot_buffer.restype = _vp
lib.ah_version.argtypes = []
lib.ah_version.restype = C.c_char_p


def check(ctx_handle, status: int) -> None:
    if status == AH_OK:
        return
    msg = lib.ah_last_error(ctx_handle).decode() if ctx_handle else "arrowhip error"
    raise _ERRS.get(status, ArrowHipError)(msg)
