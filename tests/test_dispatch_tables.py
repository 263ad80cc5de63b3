"""The reference's own DispatchBest tables against the host layer's registry, on a machine without a GPU (ahc_dispatch_best resolves a
kernel without executing it; the registry is process-global).  Transcribed from
  arrow/compute/arithmetic_test.go:715-753       TestBinaryArithmeticDispatchBest
  arrow/compute/scalar_compare_test.go:1258-1318 TestCompareKernelsDispatchBest
  arrow/compute/scalar_compare_test.go:1369-1374 TestCompareGreaterWithImplicitCastUint64EdgeCase
(rows over types this layer does not carry as argument types of these functions — Null, dictionaries, decimals, fixed-size binary,
the String / Binary mixes — are left out and listed below), plus the function NAMES the reference registers for the path.
"""
import pyarrow as pa
import pytest

from arrow_go_amd import compute as ac

I8, I16, I32, I64 = pa.int8(), pa.int16(), pa.int32(), pa.int64()
U8, U16, U32, U64 = pa.uint8(), pa.uint16(), pa.uint32(), pa.uint64()
F32, F64 = pa.float32(), pa.float64()

# {left, right, expected} — arithmetic_test.go:724-745 (both arguments go to `expected`)
ARITHMETIC_ROWS = [
    (I32, I32, I32), (I32, I8, I32), (I32, I16, I32), (I32, I32, I32), (I32, I64, I64),
    (I32, U8, I32), (I32, U16, I32), (I32, U32, I64), (I32, U64, I64),
    (U8, U8, U8), (U8, U16, U16),
    (I32, F32, F32), (F32, I64, F32), (F64, I32, F64),
]
# left out: {Int32, Null}, {Null, Int32} (no Null argument type here), the two dictionary<int8, float64> rows (dictionary arguments are
# decoded by the caller in this layer: DESIGN.md §8)


@pytest.mark.parametrize("base", ["add", "sub", "subtract", "multiply", "divide", "power"])
@pytest.mark.parametrize("suffix", ["", "_unchecked"])
def test_binary_arithmetic_dispatch_best(base, suffix):
    """("sub" is the name compute.Subtract calls, arithmetic.go:1115-1117; "subtract" the same kernels under the expression layer's name,
    :679-682 — the reference's test loops over "sub")"""
    name = base + suffix
    for left, right, expected in ARITHMETIC_ROWS:
        assert ac.dispatch_best(name, [left, right]) == [expected, expected], (name, left, right)


# {origLeft, origRight, expectLeft, expectRight} — scalar_compare_test.go:1263-1281
COMPARE_ROWS = [
    (I32, I32, I32, I32), (I32, I8, I32, I32), (I32, I16, I32, I32), (I32, I64, I64, I64),
    (I32, U8, I32, I32), (I32, U16, I32, I32), (I32, U32, I64, I64), (I32, U64, I64, I64),
    (U8, U8, U8, U8), (U8, U16, U16, U16),
    (I32, F32, F32, F32), (F32, I64, F32, F32), (F64, I32, F64, F64),
]
# left out: Null and dictionary rows (as above); timestamp / date rows (temporal operands reach these functions as their storage
# integers + a label, re-united by the temporal front end: tests/test_temporal.py covers the unit rules on the GPU); String / Binary /
# LargeString / FixedSizeBinary mixes and the decimal rows (no comparison kernels for those types in this layer: DESIGN.md §8)


@pytest.mark.parametrize("name", ["equal", "not_equal", "less", "less_equal", "greater", "greater_equal"])
def test_compare_kernels_dispatch_best(name):
    for l, r, el, er in COMPARE_ROWS:
        assert ac.dispatch_best(name, [l, r]) == [el, er], (name, l, r)


def test_compare_uint64_edge_case():
    """int64 is as wide as the promotion goes (scalar_compare_test.go:1369-1374)"""
    assert ac.dispatch_best("greater", [I8, U64]) == [I64, I64]


def test_dispatch_errors_have_the_reference_class():
    with pytest.raises(ac.ErrNotImplemented, match=r"function 'add' has no kernel matching input types \(bool, int8\)"):   # functions.go:216
        ac.dispatch_best("add", [pa.bool_(), I8])
    with pytest.raises(ac.ErrKey, match="function 'frobnicate' not found"):                                               # registry.go / exec.go:191
        ac.dispatch_best("frobnicate", [I8])
    with pytest.raises(ac.ErrInvalid, match="accepts 2 arguments but 1 passed"):                                           # functions.go:130-146
        ac.dispatch_best("add", [I8])
    # no promotion for the boolean functions or the vector functions: exact dispatch only
    assert ac.dispatch_best("and_kleene", [pa.bool_(), pa.bool_()]) == [pa.bool_(), pa.bool_()]
    with pytest.raises(ac.ErrNotImplemented):
        ac.dispatch_best("and_kleene", [pa.bool_(), I8])


# every function name the reference registers in the files of SURVEY §8(a)'s path (grep over arithmetic.go, scalar_compare.go,
# scalar_bool.go, selection.go, vector_hash.go, vector_cumulative.go, cast.go, scalar_set_lookup.go, vector_sort.go) …
REFERENCE_NAMES = """abs abs_unchecked add add_unchecked array_filter array_take bit_wise_and bit_wise_not bit_wise_or bit_wise_xor cast ceil
cumulative_sum cumulative_sum_checked dictionary_encode divide divide_unchecked equal filter floor greater greater_equal is_in is_nan
is_not_null is_null less less_equal multiply multiply_unchecked negate negate_unchecked not_equal power power_unchecked round
round_to_multiple shift_left shift_left_unchecked shift_right shift_right_unchecked sign sort sort_indices sqrt sqrt_unchecked sub
sub_unchecked subtract subtract_unchecked take trunc unique and or xor and_not and_kleene or_kleene and_not_kleene invert""".split()
# … except what DESIGN.md §8 puts out of scope: results that depend on the math library, calendar arithmetic, run-end encoding
OUT_OF_SCOPE = """acos acos_unchecked asin asin_unchecked atan atan2 cos cos_unchecked sin sin_unchecked tan tan_unchecked ln ln_unchecked log10
log10_unchecked log1p log1p_unchecked log2 log2_unchecked logb logb_unchecked ceil_temporal floor_temporal round_temporal run_end_encode
run_end_decode""".split()


def test_registry_holds_the_reference_names_of_the_path():
    missing = [n for n in REFERENCE_NAMES if not ac.lib.ahc_has_function(n.encode())]
    assert missing == []
    assert [n for n in OUT_OF_SCOPE if ac.lib.ahc_has_function(n.encode())] == []   # (the list above stays honest)
    # an alias pair is ONE set of kernels under two names
    for a, b in (("sub", "subtract"), ("sub_unchecked", "subtract_unchecked")):
        assert ac.lib.ahc_function_num_kernels(a.encode()) == ac.lib.ahc_function_num_kernels(b.encode()) > 0


def test_can_cast_numeric_and_boolean():
    """compute.CanCast(from, to) = "the target's cast function has a kernel for from's id" (cast.go:947-959); the rows of TestCanCast
    (cast_test.go:334-375) over the types this layer casts: every numeric type and Boolean to every numeric type and Boolean.  The
    per-target functions carry the reference's names (cast.go:788, 838-880: cast_boolean, cast_int8 … cast_float, cast_double); a cast to
    the SAME type never reaches them (the `cast` meta function returns its input when the types are equal, cast.go:52-54, here too)."""
    nums = [I8, U8, I16, U16, I32, U32, I64, U64, F32, F64]
    fn = {pa.bool_(): "cast_boolean", F32: "cast_float", F64: "cast_double"}
    for to in nums + [pa.bool_()]:
        name = fn.get(to, f"cast_{to}")
        assert ac.lib.ahc_has_function(name.encode()), name
        for frm in nums + [pa.bool_()]:
            if frm == to:
                continue
            assert ac.dispatch_best(name, [frm]) == [frm], (frm, to)
    for wrong in ("cast_float32", "cast_float64", "cast_bool"):
        assert not ac.lib.ahc_has_function(wrong.encode())


def test_hash_kernels_exist_for_fixed_size_binary_and_decimal_keys():
    """unique / dictionary_encode hold kernels for FixedSizeBinary, Decimal128 and Decimal256 (kernels/vector_hash.go:608-609: the
    fixed-size-binary-like types share the binary memo table; type ids 15 / 23 / 24 as in arrow/datatype.go), resolved by type id
    without a device; arithmetic on them is refused with the reference's dispatch error"""
    import ctypes as C
    for fn in ("unique", "dictionary_encode"):
        for tid in (15, 23, 24):
            tin, tout, err = (C.c_int * 1)(tid), (C.c_int * 1)(), C.create_string_buffer(512)
            assert ac.lib.ahc_dispatch_best(fn.encode(), 1, tin, tout, err, len(err)) == 0, (fn, tid, err.value)
            assert tout[0] == tid
    # FSBImpl's selection kernels (kernels/vector_selection.go:2344-2346, :2354-2356): array_filter by a boolean (id 1), array_take by
    # every integer index type (ids 2 … 9)
    for tid in (15, 23, 24):
        tin, tout, err = (C.c_int * 2)(tid, 1), (C.c_int * 2)(), C.create_string_buffer(512)
        assert ac.lib.ahc_dispatch_best(b"array_filter", 2, tin, tout, err, len(err)) == 0, (tid, err.value)
        for it in range(2, 10):
            tin = (C.c_int * 2)(tid, it)
            assert ac.lib.ahc_dispatch_best(b"array_take", 2, tin, tout, err, len(err)) == 0, (tid, it, err.value)
            assert (tout[0], tout[1]) == (tid, it)
    tin, tout, err = (C.c_int * 2)(23, 23), (C.c_int * 2)(), C.create_string_buffer(512)
    assert ac.lib.ahc_dispatch_best(b"add", 2, tin, tout, err, len(err)) != 0 and b"no kernel matching" in err.value
