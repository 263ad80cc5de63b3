"""Array-level tests of the host layer that mirrors arrow-go's compute API
(arrowhip::compute::CallFunction, the registry, the scalar/vector executors, arrow/math).
They read like the reference's own table-driven tests: inputs and expectations are the
JSON-ish literals of arrow/compute/*_test.go, compared logically (values + validity),
which is what array.ApproxEqual does.  pyarrow.compute (Arrow C++) serves as an
independent semantic cross-check on random data.
"""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

NUMERIC = [pa.uint8(), pa.int8(), pa.uint16(), pa.int16(), pa.uint32(), pa.int32(), pa.uint64(), pa.int64(),
           pa.float32(), pa.float64()]
INTS = NUMERIC[:8]


@pytest.fixture(scope="module")
def sess():
    from arrow_go_amd import compute as ac
    s = ac.Session(0)
    yield s
    s.close()


# ---- registry (no GPU needed) -----------------------------------------------------------------
def test_registry_has_the_path_functions():
    # compute/registry_test.go + the names registered by RegisterScalarArithmetic /
    # RegisterScalarComparisons / RegisterScalarBoolean / RegisterVectorSelection / RegisterVectorHash
    from arrow_go_amd import compute as ac
    for name in ["add", "add_unchecked", "subtract", "subtract_unchecked", "multiply", "multiply_unchecked",
                 "abs_unchecked", "negate_unchecked", "sign", "equal", "not_equal", "greater", "greater_equal", "less",
                 "less_equal", "and", "or", "xor", "and_not", "invert", "and_kleene", "or_kleene", "and_not_kleene",
                 "filter", "array_filter", "take", "array_take", "unique", "dictionary_encode", "greater_filter_sum",
                 "cumulative_sum", "cumulative_sum_checked", "cast", "cast_int32", "cast_double", "cast_boolean", "is_in", "sort_indices", "sort"]:
        assert ac.has_function(name), name
    assert not ac.has_function("no_such_function")
    assert ac.function_num_kernels("add") == 10            # one per numeric type
    assert ac.function_num_kernels("array_take") == 152     # (10 numeric + 4 binary-like + bool + dictionary + 3 fixed-width binary value types) × 8 index types
    assert ac.function_num_kernels("filter") == 0           # MetaFunction
    assert ac.function_num_kernels("cast_int64") == 10       # 9 other numeric types + bool
    assert ac.function_num_kernels("cumulative_sum") == 10 and ac.function_num_kernels("cumulative_sum_checked") == 10
    assert ac.num_functions() >= 30


# ---- arithmetic -----------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("typ", NUMERIC, ids=str)
@pytest.mark.parametrize("fn", ["add", "add_unchecked"])
def test_add(sess, typ, fn):
    # arrow/compute/arithmetic_test.go:325-359 (BinaryArithmeticSuite.TestAdd)
    A = lambda v: pa.array(v, type=typ)
    S = lambda v: pa.scalar(v, type=typ)
    call = lambda l, r: sess.call_function(fn, [l, r]).to_pylist()
    assert call(A([]), A([])) == []
    assert call(A([3, 2, 6]), A([1, 0, 2])) == [4, 2, 8]
    assert call(A([None, 1, None]), A([3, 4, 5])) == [None, 5, None]
    assert call(A([None, 1, 2]), A([3, 4, None])) == [None, 5, None]
    assert call(A([None]), A([None])) == [None]
    assert call(S(3), A([1, 2])) == [4, 5]
    assert call(S(3), A([None, 2])) == [None, 5]
    assert call(S(None), A([1, 2])) == [None, None]
    assert call(A([1, 2]), S(3)) == [4, 5]
    assert call(A([None, 2]), S(3)) == [None, 5]
    # scalar ∘ scalar (arithmetic_test.go:115-132)
    assert sess.call_function(fn, [S(3), S(4)]).as_py() == 7
    assert sess.call_function(fn, [S(None), S(4)]).as_py() is None


@pytest.mark.gpu
@pytest.mark.parametrize("typ", INTS, ids=str)
def test_add_overflow(sess, typ):
    from arrow_go_amd import compute as ac
    info = np.iinfo(typ.to_pandas_dtype())
    mx = pa.array([info.max], type=typ)
    with pytest.raises(ac.ErrInvalid, match="overflow"):      # arithmetic_test.go:352-358
        sess.call_function("add", [mx, mx])
    wrapped = sess.call_function("add_unchecked", [mx, mx]).to_pylist()
    with np.errstate(over="ignore"):
        npm = np.array([info.max], dtype=typ.to_pandas_dtype())
        assert wrapped == (npm + npm).tolist()
    # overflow under a null slot is ignored by the checked kernel
    assert sess.call_function("add", [pa.array([None], type=typ), mx]).to_pylist() == [None]
    with pytest.raises(ac.ErrInvalid, match="overflow"):
        sess.call_function("multiply", [mx, pa.array([2], type=typ)])
    with pytest.raises(ac.ErrInvalid, match="overflow"):
        sess.call_function("subtract", [pa.array([info.min], type=typ), pa.array([1], type=typ)])


@pytest.mark.gpu
@pytest.mark.parametrize("typ", NUMERIC, ids=str)
def test_sub_mul_unary(sess, typ):
    A = lambda v: pa.array(v, type=typ)
    assert sess.call_function("subtract", [A([3, 2, 6]), A([1, 0, 2])]).to_pylist() == [2, 2, 4]
    assert sess.call_function("multiply", [A([3, 2, None]), A([1, 0, 2])]).to_pylist() == [3, 0, None]
    assert sess.call_function("sign", [A([0, 5, None])]).to_pylist() == [0, 1, None]
    if not pa.types.is_unsigned_integer(typ):
        assert sess.call_function("negate_unchecked", [A([1, -2, None])]).to_pylist() == [-1, 2, None]
        assert sess.call_function("abs_unchecked", [A([1, -2, None])]).to_pylist() == [1, 2, None]


@pytest.mark.gpu
def test_add_sliced_inputs(sess):
    # non-zero ArraySpan.Offset on both sides: value pointers and validity bit offsets
    a = pa.array([9, 9, 9, 1, None, 3, 4, None], type=pa.int64()).slice(3, 5)
    b = pa.array([7, 10, 20, None, 40, 50], type=pa.int64()).slice(1, 5)
    assert sess.call_function("add", [a, b]).to_pylist() == [11, None, None, 44, None]
    f = pa.array([0.5, 1.5, None, 2.5], type=pa.float64()).slice(1, 3)
    assert sess.call_function("multiply", [f, pa.scalar(2.0)]).to_pylist() == [3.0, None, 5.0]


@pytest.mark.gpu
def test_dispatch_errors(sess):
    from arrow_go_amd import compute as ac
    with pytest.raises(ac.ErrKey, match="function 'nope' not found"):                # exec.go:191-199
        sess.call_function("nope", [pa.array([1])])
    with pytest.raises(ac.ErrInvalid, match="accepts 2 arguments but 1 passed"):     # functions.go:130-146
        sess.call_function("add", [pa.array([1])])
    with pytest.raises(ac.ErrInvalid, match="same length"):                          # executor.go inferBatchLength
        sess.call_function("add", [pa.array([1, 2]), pa.array([1, 2, 3])])
    with pytest.raises(ac.ErrNotImplemented, match="no kernel matching input types"):
        sess.call_function("abs_unchecked", [pa.array([True])])                       # functions.go:199-218 DispatchExact


@pytest.mark.gpu
def test_sub_is_the_name_compute_subtract_calls(sess):
    """compute.Subtract → impl(ctx, "sub", …), "sub_unchecked" with NoCheckOverflow (arithmetic.go:679-682, 1115-1117): the same kernels as
    "subtract" / "subtract_unchecked", overflow rule included"""
    from arrow_go_amd import compute as ac
    a, b = pa.array([5, None, -7, 2**31 - 1], pa.int32()), pa.array([7, 1, None, 1], pa.int32())
    assert sess.call_function("sub", [a, b]).to_pylist() == sess.call_function("subtract", [a, b]).to_pylist() == [-2, None, None, 2**31 - 2]
    lo = pa.array([-2**31, 3], pa.int32())
    assert sess.call_function("sub_unchecked", [lo, pa.array([1, 1], pa.int32())]).to_pylist() == [2**31 - 1, 2]      # wraps
    with pytest.raises(ac.ErrInvalid, match="overflow"):
        sess.call_function("sub", [lo, pa.array([1, 1], pa.int32())])


@pytest.mark.gpu
def test_child_registry_alias(sess):
    # registry.go:69-73 NewChildRegistry + AddFunction(allowOverwrite)
    from arrow_go_amd import compute as ac
    sess.add_alias("plus", "add_unchecked")
    assert sess.call_function("plus", [pa.array([1, 2]), pa.array([10, 20])]).to_pylist() == [11, 22]
    with pytest.raises(ac.ErrKey, match="already have a function registered"):
        sess.add_alias("plus", "add")
    sess.add_alias("plus", "add", allow_overwrite=True)
    assert not ac.has_function("plus")  # the process-default registry is untouched


# ---- compare ----------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("typ", NUMERIC, ids=str)
def test_compare(sess, typ):
    # scalar_compare_test.go:299-483
    one = pa.scalar(1, type=typ)
    A = lambda v: pa.array(v, type=typ)
    call = lambda fn, l, r: sess.call_function(fn, [l, r]).to_pylist()
    assert call("equal", A([]), one) == []
    assert call("equal", A([None]), one) == [None]
    assert call("equal", A([0, 0, 1, 1, 2, 2]), one) == [False, False, True, True, False, False]
    assert call("equal", A([None, 0, 1, 1]), one) == [None, False, True, True]
    assert call("not_equal", A([5, 4, 3, 2, 1, 0]), one) == [True, True, True, True, False, True]
    assert call("greater", A([0, 1, 2, 3, 4, 5]), one) == [False, False, True, True, True, True]
    assert call("greater_equal", A([None, 0, 1, 1]), one) == [None, False, True, True]
    assert call("less", A([0, 0, 1, 1, 2, 2]), one) == [True, True, False, False, False, False]
    assert call("less_equal", A([None, 0, 1, 1]), one) == [None, True, True, True]
    assert call("equal", one, A([0, 1, 2, 3, 4, 5])) == [False, True, False, False, False, False]
    assert call("greater", A([1, 2, None]), A([0, 2, 5])) == [True, False, None]
    assert sess.call_function("less", [one, pa.scalar(None, type=typ)]).as_py() is None


@pytest.mark.gpu
def test_compare_random_vs_arrow_cpp(sess):
    # scalar_compare_test.go:93-158 randomized compare vs slowCompare — here vs Arrow C++
    rng = np.random.default_rng(0)
    for n in [1, 65, 1000, 70001]:
        for null_p in (0.0, 0.3):
            mask = rng.random(n) < null_p
            l = pa.array(rng.integers(0, 10, n), mask=mask, type=pa.int32())
            r = pa.array(rng.integers(0, 10, n), mask=rng.random(n) < null_p, type=pa.int32())
            for fn in ["equal", "not_equal", "greater", "greater_equal", "less", "less_equal"]:
                assert sess.call_function(fn, [l, r]).equals(getattr(pc, fn)(l, r)), (fn, n)
                assert sess.call_function(fn, [l, pa.scalar(4, pa.int32())]).equals(getattr(pc, fn)(l, pa.scalar(4, pa.int32())))


# ---- boolean ---------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_boolean_vs_arrow_cpp(sess):
    rng = np.random.default_rng(1)
    for n in [1, 9, 64, 1000, 70001]:
        l = pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.2)
        r = pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.2)
        for fn in ["and", "or", "xor", "and_not", "and_kleene", "or_kleene", "and_not_kleene"]:
            got, exp = sess.call_function(fn, [l, r]), getattr(pc, fn)(l, r)
            assert got.equals(exp), (fn, n)
            # sliced (offset) operands
            if n > 20:
                ls, rs = l.slice(3, n - 10), r.slice(7, n - 10)
                assert sess.call_function(fn, [ls, rs]).equals(getattr(pc, fn)(ls, rs)), (fn, n, "sliced")
        assert sess.call_function("invert", [l]).equals(pc.invert(l))
        for sv in (True, False, None):
            sc = pa.scalar(sv, pa.bool_())
            for fn in ["and_kleene", "or_kleene", "and", "or"]:
                assert sess.call_function(fn, [l, sc]).equals(getattr(pc, fn)(l, sc)), (fn, sv)


# ---- filter -------------------------------------------------------------------------------------------
FILTER_CASES = [  # vector_selection_test.go:449-484
    ([], [], "drop", []),
    ([9], [False], "drop", []),
    ([9], [True], "drop", [9]),
    ([9], [None], "drop", []),
    ([9], [None], "emit_null", [None]),
    ([None], [True], "drop", [None]),
    ([7, 8, 9], [False, True, False], "drop", [8]),
    ([7, 8, 9], [True, False, True], "drop", [7, 9]),
    ([None, 8, 9], [False, True, False], "drop", [8]),
    ([7, 8, 9], [None, True, False], "drop", [8]),
    ([7, 8, 9], [None, True, False], "emit_null", [None, 8]),
    ([7, 8, 9], [True, None, True], "drop", [7, 9]),
    ([7, 8, 9], [True, None, True], "emit_null", [7, None, 9]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("typ", NUMERIC, ids=str)
def test_filter(sess, typ):
    for values, filt, sel, exp in FILTER_CASES:
        v, f = pa.array(values, type=typ), pa.array(filt, type=pa.bool_())
        assert sess.call_function("filter", [v, f], f"null_selection_behavior={sel}").to_pylist() == exp
        # vector_selection_test.go:121-144: null fillers / [true,false] prepended, then sliced
        vs = pa.concat_arrays([pa.array([None] * 3, type=typ), v]).slice(3)
        fs = pa.concat_arrays([pa.array([True, False]), f]).slice(2)
        assert sess.call_function("filter", [vs, fs], f"null_selection_behavior={sel}").to_pylist() == exp


@pytest.mark.gpu
def test_filter_errors_and_sliced_filter(sess):
    from arrow_go_amd import compute as ac
    v = pa.array([7, 8, 9], pa.int64())
    f = pa.array([False, True, True, True, False, True]).slice(3, 3)
    assert sess.call_function("filter", [v, f]).to_pylist() == [7, 9]                 # :474-478
    with pytest.raises(ac.ErrInvalid, match="same length"):                            # :480-483
        sess.call_function("filter", [v, pa.array([True, False])])
    with pytest.raises(ac.ErrNotImplemented):
        sess.call_function("filter", [v, pa.array([1, 0, 1])])


@pytest.mark.gpu
def test_filter_random_vs_arrow_cpp(sess):
    # vector_selection_test.go:554-613 (compare-then-filter on random data, lengths 8..512 and beyond)
    rng = np.random.default_rng(2)
    for n in [8, 64, 512, 4099, 70001]:
        for null_p in (0.0, 0.1):
            v = pa.array(rng.integers(-100, 100, n), mask=rng.random(n) < null_p, type=pa.int64())
            mask = sess.call_function("greater", [v, pa.scalar(0, pa.int64())])
            for sel, pcsel in (("drop", "drop"), ("emit_null", "emit_null")):
                got = sess.call_function("filter", [v, mask], f"null_selection_behavior={sel}")
                assert got.equals(pc.filter(v, mask, null_selection_behavior=pcsel)), (n, null_p, sel)


# ---- take ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("typ", NUMERIC, ids=str)
@pytest.mark.parametrize("ityp", [pa.int8(), pa.uint32(), pa.int64()], ids=str)
def test_take(sess, typ, ityp):
    from arrow_go_amd import compute as ac
    A = lambda v: pa.array(v, type=typ)
    I = lambda v: pa.array(v, type=ityp)
    call = lambda v, i: sess.call_function("take", [v, i]).to_pylist()
    assert call(A([7, 8, 9]), I([])) == []                                             # :1127-1141
    assert call(A([7, 8, 9]), I([0, 1, 0])) == [7, 8, 7]
    assert call(A([None, 8, 9]), I([0, 1, 0])) == [None, 8, None]
    assert call(A([7, 8, 9]), I([None, 1, 0])) == [None, 8, 7]
    assert call(A([None, 8, 9]), I([])) == []
    assert call(A([7, 8, 9]), I([0, 0, 0, 0, 0, 0, 2])) == [7, 7, 7, 7, 7, 7, 9]
    with pytest.raises(ac.ErrIndex, match="9 out of bounds"):
        call(A([7, 8, 9]), I([0, 9, 0]))
    if pa.types.is_signed_integer(ityp):
        with pytest.raises(ac.ErrIndex, match="-1 out of bounds"):
            call(A([7, 8, 9]), I([0, -1, 0]))
    # sliced values and sliced indices (:213-253)
    vs = pa.concat_arrays([A([None, None]), A([7, 8, 9])]).slice(2)
    is_ = pa.concat_arrays([I([1]), I([2, None, 0])]).slice(1)
    assert call(vs, is_) == [9, None, 7]


@pytest.mark.gpu
def test_take_random_vs_arrow_cpp(sess):
    rng = np.random.default_rng(3)
    for nv, ni in [(100, 1000), (5000, 70001)]:
        v = pa.array(rng.integers(-1000, 1000, nv), mask=rng.random(nv) < 0.1, type=pa.int64())
        i = pa.array(rng.integers(0, nv, ni), mask=rng.random(ni) < 0.1, type=pa.int32())
        assert sess.call_function("take", [v, i]).equals(pc.take(v, i))


# ---- hash ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("typ", NUMERIC, ids=str)
def test_unique_and_dictionary_encode(sess, typ):
    A = lambda v: pa.array(v, type=typ)
    uq = lambda v: sess.call_function("unique", [v]).to_pylist()
    assert uq(A([2, None, 2, 1])) == [2, None, 1]                                      # vector_hash_test.go:236-254
    assert uq(A([None, None, 3, 1])) == [None, 3, 1]
    assert uq(A([1, 2, None, 3, 2, None]).slice(1, 4)) == [2, None, 3]
    d = sess.call_function("dictionary_encode", [A([10, 20, 10, None, 20, None])])    # :541-591 (numeric twin)
    assert d.indices.to_pylist() == [0, 1, 0, None, 1, None] and d.dictionary.to_pylist() == [10, 20]
    d = sess.call_function("dictionary_encode", [A([10, 20, 10, None, 20, None])], "null_encoding_behavior=encode")
    assert d.indices.to_pylist() == [0, 1, 0, 2, 1, 2] and d.dictionary.to_pylist() == [10, 20, None]
    # random vs Arrow C++ (same first-seen-order contract)
    rng = np.random.default_rng(4)
    v = pa.array(rng.integers(0, 120, 70001), mask=rng.random(70001) < 0.05, type=typ)
    assert sess.call_function("unique", [v]).equals(pc.unique(v))
    got, exp = sess.call_function("dictionary_encode", [v]), pc.dictionary_encode(v)
    assert got.indices.equals(exp.indices) and got.dictionary.equals(exp.dictionary)


@pytest.mark.gpu
def test_unique_and_dictionary_encode_boolean(sess):
    """doAppendBoolean (kernels/vector_hash.go:387-415): the bit is the uint8 key of the same memo table"""
    A = lambda v: pa.array(v, type=pa.bool_())
    uq = lambda v: sess.call_function("unique", [v])
    assert uq(A([False, None, True, False, None, True])).to_pylist() == [False, None, True]   # TestUniqueBoolean, vector_hash_test.go:300-319
    assert uq(A([True, True, False, True])).to_pylist() == [True, False]
    d = sess.call_function("dictionary_encode", [A([False, True, False, None, True])])           # TestDictionaryEncodeBoolean :637-698
    assert d.indices.to_pylist() == [0, 1, 0, None, 1] and d.dictionary.to_pylist() == [False, True]
    d = sess.call_function("dictionary_encode", [A([True, False, True, None, True, False, True]).slice(1, 5)], "null_encoding_behavior=encode")  # :700-760
    assert d.indices.to_pylist() == [0, 1, 2, 1, 0] and d.dictionary.to_pylist() == [False, True, None]
    assert uq(A([True, None, True, None, False])).to_pylist() == [True, None, False]
    assert uq(A([None, None])).to_pylist() == [None]
    assert uq(A([True, False, None, True]).slice(1, 3)).to_pylist() == [False, None, True]
    assert uq(A([])).to_pylist() == [] and uq(A([True])).type == pa.bool_()
    d = sess.call_function("dictionary_encode", [A([True, False, True, None, False, None])])
    assert d.type == pa.dictionary(pa.int32(), pa.bool_())
    assert d.indices.to_pylist() == [0, 1, 0, None, 1, None] and d.dictionary.to_pylist() == [True, False]
    d = sess.call_function("dictionary_encode", [A([True, False, True, None, False, None])], "null_encoding_behavior=encode")
    assert d.indices.to_pylist() == [0, 1, 0, 2, 1, 2] and d.dictionary.to_pylist() == [True, False, None]
    rng = np.random.default_rng(6)
    v = pa.array(rng.random(70003) < 0.3, mask=rng.random(70003) < 0.05, type=pa.bool_()).slice(3)
    assert uq(v).equals(pc.unique(v))
    got, exp = sess.call_function("dictionary_encode", [v]), pc.dictionary_encode(v)
    assert got.indices.equals(exp.indices) and got.dictionary.equals(exp.dictionary)


@pytest.mark.gpu
@pytest.mark.parametrize("typ", [pa.string(), pa.large_string(), pa.binary(), pa.large_binary()], ids=str)
def test_unique_and_dictionary_encode_binary(sess, typ):
    isbin = pa.types.is_binary(typ) or pa.types.is_large_binary(typ)
    conv = (lambda x: x.encode() if isinstance(x, str) and isbin else x)
    A = lambda v: pa.array([conv(x) for x in v], type=typ)
    L = lambda v: [conv(x) for x in v]
    uq = lambda v: sess.call_function("unique", [v])
    assert uq(A(["test", None, "test2", "test"])).to_pylist() == L(["test", None, "test2"])          # vector_hash_test.go:272-279
    assert uq(A(["foo", "bar", "foo", "bar", "baz", "quuux", "foo"])).to_pylist() == L(["foo", "bar", "baz", "quuux"])  # :420-449
    assert uq(A(["test", None, "test2", "test"])).type == typ
    d = sess.call_function("dictionary_encode", [A(["foo", "bar", "foo", None, "bar", None])])      # :541-594
    assert d.type == pa.dictionary(pa.int32(), typ)
    assert d.indices.to_pylist() == [0, 1, 0, None, 1, None] and d.dictionary.to_pylist() == L(["foo", "bar"])
    d = sess.call_function("dictionary_encode", [A(["foo", "bar", "foo", None, "bar", None])], "null_encoding_behavior=encode")
    assert d.indices.to_pylist() == [0, 1, 0, 2, 1, 2] and d.dictionary.to_pylist() == L(["foo", "bar", None])
    assert d.dictionary.null_count == 1 and d.indices.null_count == 0
    d = sess.call_function("dictionary_encode", [A(["ignored", "foo", None, "bar", "foo", "ignored"]).slice(1, 4)])  # :802-825
    assert d.indices.to_pylist() == [0, None, 1, 0] and d.dictionary.to_pylist() == L(["foo", "bar"])
    assert uq(A([])).to_pylist() == []
    # random vs Arrow C++ (same first-seen-order contract)
    rng = np.random.default_rng(5)
    words = ["w%d" % i * int(1 + i % 5) for i in range(300)] + ["", "a" * 40]
    v = pa.array([conv(words[j]) for j in rng.integers(0, len(words), 50001)], mask=rng.random(50001) < 0.05, type=typ)
    assert uq(v).equals(pc.unique(v))
    got, exp = sess.call_function("dictionary_encode", [v]), pc.dictionary_encode(v)
    assert got.indices.equals(exp.indices) and got.dictionary.equals(exp.dictionary)
    got, exp = sess.call_function("dictionary_encode", [v], "null_encoding_behavior=encode"), pc.dictionary_encode(v, null_encoding="encode")
    assert got.indices.equals(exp.indices) and got.dictionary.equals(exp.dictionary)


# ---- cumulative_sum (arrow/compute/vector_cumulative_test.go) -----------------------------------------------
@pytest.mark.gpu
def test_cumulative_sum_reference_tables(sess):
    from arrow_go_amd import compute as ac
    I32 = lambda v: pa.array(v, type=pa.int32())
    cs = lambda a, o="": sess.call_function("cumulative_sum", [a], o).to_pylist()
    csc = lambda a, o="": sess.call_function("cumulative_sum_checked", [a], o).to_pylist()
    assert cs(I32([1, 2, 3, 4])) == [1, 3, 6, 10]                                   # TestCumulativeSum :41
    assert cs(I32([])) == [] and cs(I32([None, None])) == [None, None]               # AdditionalInputs :94
    assert cs(pa.array([1, 2, 3], pa.uint8())) == [1, 3, 6]
    assert cs(pa.array([1.5, 2.5], pa.float32())) == [1.5, 4.0] and cs(pa.array([1.5, 2.5], pa.float64())) == [1.5, 4.0]
    assert cs(I32([0, 1, 2, 3]).slice(1, 2)) == [1, 3]
    assert cs(I32([9, None, 2, 3, 99]).slice(1, 3)) == [None, None, None]
    assert cs(I32([9, None, 2, 3, 99]).slice(1, 3), "skip_nulls=1") == [None, 2, 5]
    r = sess.call_function("cumulative_sum", [pa.scalar(3, pa.int32())])              # scalar input → 1-row array
    assert isinstance(r, pa.Array) and r.to_pylist() == [3]
    v = I32([1, None, 2, None, 3])                                                   # NullsAndStart :238
    assert cs(v) == [1, None, None, None, None]
    assert cs(v, "skip_nulls=1") == [1, None, 3, None, 6]
    assert cs(v, "skip_nulls=1;start=int64:10") == [11, None, 13, None, 16]          # Start is safe-cast to the input type
    out = sess.call_function("cumulative_sum", [v], "skip_nulls=1")
    assert out.null_count == 2 and out.type == pa.int32()
    # TestCumulativeSumNullScalarInput :172
    for typ in NUMERIC:
        for fn in ("cumulative_sum", "cumulative_sum_checked"):
            for o in ("", "skip_nulls=1", "start=%s:10" % typ, "start=%s:10;skip_nulls=1" % typ):
                assert sess.call_function(fn, [pa.scalar(None, typ)], o).to_pylist() == [None]
    # typed null start / unsafe start casts are arrow.ErrInvalid (:277-344)
    for fn in ("cumulative_sum", "cumulative_sum_checked"):
        with pytest.raises(ac.ErrInvalid, match="start value must be valid"):
            sess.call_function(fn, [I32([1])], "start=null:int32")
    for typ, start in ((pa.int8(), "int64:128"), (pa.int8(), "int64:-129"), (pa.uint8(), "int64:-1"), (pa.int32(), "double:1.5")):
        with pytest.raises(ac.ErrInvalid, match="cannot cast cumulative sum start value"):
            sess.call_function("cumulative_sum", [pa.array([0], typ)], "start=" + start)
    assert cs(pa.array([0], pa.int8()), "start=int64:127") == [127]
    assert cs(pa.array([1.0], pa.float64()), "start=int32:2") == [3.0]
    # checked (:750-840)
    assert cs(pa.array([127, 1], pa.int8())) == [127, -128]
    for typ in INTS:
        lo, hi = (0, 2**typ.bit_width - 1) if pa.types.is_unsigned_integer(typ) else (-2**(typ.bit_width - 1), 2**(typ.bit_width - 1) - 1)
        with pytest.raises(ac.ErrInvalid, match="overflow"):
            csc(pa.array([hi, 1], typ))
        assert csc(pa.array([hi - 1, 1], typ)) == [hi - 1, hi]
        if lo < 0:
            with pytest.raises(ac.ErrInvalid, match="overflow"):
                csc(pa.array([lo, -1], typ))
            assert csc(pa.array([lo + 1, -1], typ)) == [lo + 1, lo]
    with pytest.raises(ac.ErrNotImplemented, match="no kernel matching input types"):
        cs(pa.array([True, False]))


@pytest.mark.gpu
@pytest.mark.parametrize("typ", NUMERIC, ids=str)
def test_cumulative_sum_random_vs_arrow_cpp(sess, typ):
    # Arrow C++'s cumulative_sum has the same contract (wraparound, skip_nulls, start)
    rng = np.random.default_rng(typ.bit_width)
    n = 50021
    if pa.types.is_floating(typ):
        vals = rng.integers(-100, 100, n).astype(typ.to_pandas_dtype())   # integer-valued: every summation order is exact
    else:
        info = np.iinfo(typ.to_pandas_dtype())
        vals = rng.integers(info.min, info.max, n, dtype=typ.to_pandas_dtype(), endpoint=True)
    for p_null, skip in ((0.0, False), (0.2, True), (0.0001, False)):
        a = pa.array(vals, mask=rng.random(n) < p_null if p_null else None, type=typ).slice(17, n - 30)
        got = sess.call_function("cumulative_sum", [a], "skip_nulls=%d;start=%s:3" % (skip, typ))
        exp = pc.cumulative_sum(a, start=pa.scalar(3, typ), skip_nulls=skip)
        assert got.equals(exp), (typ, p_null, skip)
        assert got.null_count == exp.null_count


# ---- cast + implicit numeric promotion (arrow/compute/cast_test.go, arithmetic.go DispatchBest) -----------
_PA = {"uint8": pa.uint8(), "int8": pa.int8(), "uint16": pa.uint16(), "int16": pa.int16(), "uint32": pa.uint32(), "int32": pa.int32(),
       "uint64": pa.uint64(), "int64": pa.int64(), "float": pa.float32(), "double": pa.float64(), "bool": pa.bool_()}


def _tname(t):
    return [k for k, v in _PA.items() if v == t][0]


@pytest.mark.gpu
def test_cast_reference_tables(sess):
    from arrow_go_amd import compute as ac
    cast = lambda arr, to, o="": sess.call_function("cast", [arr], "to_type=%s;%s" % (_tname(to), o)).to_pylist()
    A = lambda v, t: pa.array(v, type=t)
    # TestToIntUpcast / TestToIntDowncastSafe / Unsafe (cast_test.go:483-586)
    assert cast(A([0, None, 127, -1, 0], pa.int8()), pa.int32()) == [0, None, 127, -1, 0]
    assert cast(A([0, 100, 200, 255, 0], pa.uint8()), pa.int16()) == [0, 100, 200, 255, 0]
    assert cast(A([0, None, 200, 1, 2], pa.int16()), pa.uint8()) == [0, None, 200, 1, 2]
    with pytest.raises(ac.ErrInvalid, match="integer value 256 not in range: 0 to 255"):
        cast(A([0, None, 256, 0, 0], pa.int16()), pa.uint8())
    with pytest.raises(ac.ErrInvalid, match="integer value -70000 not in range: -32768 to 32767"):
        cast(A([0, None, 2000, -70000, 2], pa.int32()), pa.int16())
    assert cast(A([0, None, 256, 1, 2, -1], pa.int16()), pa.uint8(), "allow_int_overflow=1") == [0, None, 0, 1, 2, 255]
    assert cast(A([0, None, 2000, 70000, -70000], pa.int32()), pa.int16(), "safe=0") == [0, None, 2000, 4464, -4464]
    # TestIntegerSignedToUnsigned / UnsignedToSigned (:520-571)
    i32 = A([-2147483648, None, -1, 65535, 2147483647], pa.int32())
    for to in (pa.uint32(), pa.uint64(), pa.uint16()):
        with pytest.raises(ac.ErrInvalid, match="not in range"):
            cast(i32, to)
    assert cast(i32, pa.uint64(), "allow_int_overflow=1") == [18446744071562067968, None, 18446744073709551615, 65535, 2147483647]
    u32 = A([4294967295, None, 0, 32768], pa.uint32())
    with pytest.raises(ac.ErrInvalid, match="integer value 32768 not in range: 0 to 32767"):
        cast(u32.slice(1), pa.int16())
    assert cast(u32, pa.int16(), "allow_int_overflow=1") == [-1, None, 0, -32768]
    # TestFloatingToInt (:588-603)
    for frm in (pa.float32(), pa.float64()):
        for to in (pa.int32(), pa.int64()):
            assert cast(A([1.0, None, 0.0, -1.0, 5.0], frm), to) == [1, None, 0, -1, 5]
            with pytest.raises(ac.ErrInvalid, match="float value 1.500000 was truncated converting to " + str(to)):
                cast(A([1.5, 0.0, None, 0.5, -1.5, 5.5], frm), to)
            assert cast(A([1.5, 0.0, None, 0.5, -1.5, 5.5], frm), to, "allow_float_truncate=1") == [1, 0, None, 0, -1, 5]
    # TestIntToFloating (:611-629)
    for frm in (pa.uint32(), pa.int32()):
        with pytest.raises(ac.ErrInvalid, match="integer value 16777217 not in range"):
            cast(A([16777216, 16777217], frm), pa.float32())
        assert cast(A([16777216], frm), pa.float32()) == [16777216.0]
    i64 = A([-9223372036854775808, -9223372036854775807, 0, 9223372036854775806, 9223372036854775807], pa.int64())
    with pytest.raises(ac.ErrInvalid, match="not in range: -9007199254740992 to 9007199254740992"):
        cast(i64, pa.float64())
    masked = pa.array(i64.to_pylist(), mask=np.array([True, True, False, True, True]), type=pa.int64())
    assert cast(masked, pa.float64()) == [None, None, 0.0, None, None]
    # TestNumericToBool (:456-468) + bool → numeric
    for t in NUMERIC:
        assert cast(A([0, None, 127, 1, 0], t), pa.bool_()) == [False, None, True, True, False]
        assert cast(pa.array([True, None, False, True]), t) == [1, None, 0, 1]
    assert cast(A([0, None, 127, -1, 0], pa.int8()), pa.bool_()) == [False, None, True, True, False]
    assert cast(A([0.0, None, float("nan"), -1.0, -0.0], pa.float64()), pa.bool_()) == [False, None, True, True, False]
    # identity → the input itself; scalar input → scalar; missing ToType
    same = sess.call_function("cast", [A([1, None], pa.int32())], "to_type=int32")
    assert same.to_pylist() == [1, None] and same.type == pa.int32()
    sc = sess.call_function("cast", [pa.scalar(7, pa.int8())], "to_type=double")
    assert isinstance(sc, pa.Scalar) and sc.as_py() == 7.0 and sc.type == pa.float64()
    with pytest.raises(ac.ErrInvalid, match="cast requires that options be passed with a ToType"):
        sess.call_function("cast", [A([1], pa.int32())], "allow_int_overflow=1")
    out = sess.call_function("cast", [A([1, None, 3], pa.int16()).slice(1)], "to_type=int64")
    assert out.type == pa.int64() and out.null_count == 1 and out.to_pylist() == [None, 3]


@pytest.mark.gpu
@pytest.mark.parametrize("frm", NUMERIC, ids=str)
def test_cast_random_vs_arrow_cpp(sess, frm):
    # Arrow C++ cast: the same safe / unsafe contract; values every target can hold → safe cast succeeds
    rng = np.random.default_rng(frm.bit_width + 7)
    n = 20011
    vals = rng.integers(0, 100, n).astype(frm.to_pandas_dtype())
    a = pa.array(vals, mask=rng.random(n) < 0.1, type=frm).slice(5, n - 11)
    for to in NUMERIC:
        got = sess.call_function("cast", [a], "to_type=" + _tname(to))
        exp = pc.cast(a, to)
        assert got.equals(exp) and got.type == to, (frm, to)
    # unsafe int narrowing wraps like Arrow C++
    if pa.types.is_integer(frm):
        info = np.iinfo(frm.to_pandas_dtype())
        wide = pa.array(rng.integers(info.min, info.max, n, dtype=frm.to_pandas_dtype(), endpoint=True), type=frm)
        for to in INTS:
            got = sess.call_function("cast", [wide], "to_type=%s;safe=0" % _tname(to))
            assert got.equals(pc.cast(wide, to, safe=False)), (frm, to)


@pytest.mark.gpu
def test_implicit_numeric_promotion(sess):
    """DispatchBest: no exact kernel → both sides go to commonNumeric through a SAFE cast
    (arithmetic.go:112-142, scalar_compare.go:37-63, utils.go:178-240, exec.go:105-114)."""
    from arrow_go_amd import compute as ac
    A = lambda v, t: pa.array(v, type=t)
    pairs = [(pa.int32(), pa.int64(), pa.int64()), (pa.int8(), pa.uint8(), pa.int16()), (pa.uint16(), pa.uint32(), pa.uint32()),
             (pa.int16(), pa.uint32(), pa.int64()), (pa.uint64(), pa.int8(), pa.int64()), (pa.int64(), pa.float32(), pa.float32()),
             (pa.float32(), pa.float64(), pa.float64()), (pa.uint8(), pa.float64(), pa.float64()), (pa.uint32(), pa.int32(), pa.int64())]
    for lt, rt, common in pairs:
        l, r = A([1, None, 3, 40], lt), A([5, 6, None, 2], rt)
        for fn in ("add", "subtract_unchecked", "multiply"):
            if fn.startswith("subtract") and pa.types.is_unsigned_integer(common):
                l2, r2 = A([10, None, 30, 40], lt), r
            else:
                l2, r2 = l, r
            got = sess.call_function(fn, [l2, r2])
            exp = getattr(pc, fn.replace("_unchecked", ""))(l2, r2)
            assert got.type == common, (lt, rt, fn, got.type)
            assert got.to_pylist() == exp.to_pylist(), (lt, rt, fn)
        for fn in ("equal", "greater", "less_equal"):
            assert sess.call_function(fn, [l, r]).to_pylist() == getattr(pc, fn)(l, r).to_pylist(), (lt, rt, fn)
        # array ∘ scalar of another type
        assert sess.call_function("add", [l, pa.scalar(2, rt)]).to_pylist() == pc.add(l, pa.scalar(2, rt)).to_pylist()
        assert sess.call_function("greater", [pa.scalar(2, lt), r]).to_pylist() == pc.greater(pa.scalar(2, lt), r).to_pylist()
    # the implicit cast is SAFE: a uint64 beyond int64 cannot be promoted next to a signed operand
    with pytest.raises(ac.ErrInvalid, match="integer value 18446744073709551615 not in range: 0 to 9223372036854775807"):
        sess.call_function("add", [A([2**64 - 1], pa.uint64()), A([1], pa.int8())])
    with pytest.raises(ac.ErrInvalid, match="not in range"):
        sess.call_function("equal", [A([2**53 + 1], pa.int64()), A([1.0], pa.float64())])
    # non-numeric operands are not promoted
    with pytest.raises(ac.ErrNotImplemented, match="no kernel matching input types"):
        sess.call_function("add", [pa.array([True]), A([1], pa.int8())])


# ---- is_in (arrow/compute/scalar_set_lookup_test.go:104-168) -------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("typ", NUMERIC, ids=str)
def test_is_in_primitive(sess, typ):
    T, F, N = True, False, None
    A = lambda v: pa.array(v, type=typ)
    isin = lambda vals, vset, nb: sess.call_function("is_in", [A(vals)], "null_matching_behavior=" + nb, value_set=A(vset)).to_pylist()
    cases = [
        ([0, 1, 2, 3, 2], [2, 1], {"match": [F, T, T, F, T]}),
        ([None, 1, 2, 3, 2], [2, 1], {"match": [F, T, T, F, T], "skip": [F, T, T, F, T], "emit_null": [N, T, T, F, T], "inconclusive": [N, T, T, F, T]}),
        ([0, 1, 2, 3, 2], [2, None, 1], {"match": [F, T, T, F, T], "skip": [F, T, T, F, T], "emit_null": [F, T, T, F, T], "inconclusive": [N, T, T, N, T]}),
        ([None, 1, 2, 3, 2], [2, None, 1], {"match": [T, T, T, F, T], "skip": [F, T, T, F, T], "emit_null": [N, T, T, F, T], "inconclusive": [N, T, T, N, T]}),
        ([None, 1, 2, 3, 2], [None, 2, 2, None, 1, 1], {"match": [T, T, T, F, T], "skip": [F, T, T, F, T], "emit_null": [N, T, T, F, T],
                                                        "inconclusive": [N, T, T, N, T]}),
        ([], [], {"match": []}),
    ]
    for vals, vset, exp in cases:
        for nb, want in exp.items():
            assert isin(vals, vset, nb) == want, (vals, vset, nb)


@pytest.mark.gpu
def test_is_in_options_casts_and_arrow_cpp(sess):
    from arrow_go_amd import compute as ac
    with pytest.raises(ac.ErrInvalid, match="without SetOptions"):
        sess.call_function("is_in", [pa.array([1])])
    # the value set is safe-cast to the input type (initSetLookup, scalar_set_lookup.go:93-112)
    got = sess.call_function("is_in", [pa.array([1, 2, 300], pa.int32())], value_set=pa.array([2, 300], pa.int64()))
    assert got.to_pylist() == [False, True, True]
    with pytest.raises(ac.ErrInvalid, match="not in range"):
        sess.call_function("is_in", [pa.array([1], pa.int8())], value_set=pa.array([1000], pa.int64()))
    # sliced input and sliced value set, random, against Arrow C++ (default = match nulls; skip_nulls=True = skip)
    rng = np.random.default_rng(11)
    n = 50021
    a = pa.array(rng.integers(0, 4000, n), mask=rng.random(n) < 0.1, type=pa.int64()).slice(7, n - 20)
    vs = pa.array(rng.integers(0, 4000, 900), mask=rng.random(900) < 0.05, type=pa.int64()).slice(3, 800)
    assert sess.call_function("is_in", [a], value_set=vs).equals(pc.is_in(a, value_set=vs))
    assert sess.call_function("is_in", [a], "null_matching_behavior=skip", value_set=vs).equals(pc.is_in(a, value_set=vs, skip_nulls=True))
    f = pa.array([0.0, -0.0, 1.5, None], pa.float64())
    assert sess.call_function("is_in", [f], value_set=pa.array([0.0, None], pa.float64())).to_pylist() == [True, False, False, True]


# ---- sort_indices / sort (arrow/compute/vector_sort_test.go) ------------------------------------------------
@pytest.mark.gpu
def test_sort_indices_reference_tables(sess):
    from arrow_go_amd import compute as ac
    si = lambda a, o="order=ascending": sess.call_function("sort_indices", [a], o).to_pylist()
    I32 = lambda v: pa.array(v, type=pa.int32())
    assert si(I32([3, 1, 4, 1, 5, 9, 2, 6])) == [1, 3, 6, 0, 2, 4, 7, 5]                                  # TestSortIndices :40
    assert si(I32([3, 1, 4, 1, 5, 9, 2, 6]), "order=descending") == [5, 7, 4, 2, 0, 6, 1, 3]
    assert si(I32([3, None, 4, 0, 5])) == [3, 0, 2, 4, 1]
    assert si(I32([3, None, 4, 0, 5]), "null_placement=at_start") == [1, 3, 0, 2, 4]
    assert si(pa.array([3.14, float("nan"), 2.71, 1.41, float("nan")])) == [3, 2, 0, 1, 4]
    assert si(I32([])) == [] and si(I32([None, None, None])) == [0, 1, 2]
    assert si(I32([1, 2, 1, 2, 1])) == [0, 2, 4, 1, 3]
    v = pa.array([0, 1, None, -3, None, -42, 5], pa.int16())                                             # CppArrayParity :1186
    assert si(v) == [5, 3, 0, 1, 6, 2, 4]
    assert si(v, "order=descending;null_placement=at_start") == [2, 4, 6, 1, 0, 3, 5]
    out = sess.call_function("sort_indices", [v], "order=ascending")
    assert out.type == pa.uint64() and out.null_count == 0
    # sort = take(input, sort_indices(input))  (TestSortArray :326)
    assert sess.call_function("sort", [v], "order=ascending").to_pylist() == [-42, -3, 0, 1, 5, None, None]
    assert sess.call_function("sort", [v.slice(1, 5)], "order=descending").to_pylist() == [1, -3, -42, None, None]
    with pytest.raises(ac.ErrInvalid, match="at least one sort key"):
        sess.call_function("sort_indices", [v])
    with pytest.raises(ac.ErrNotImplemented, match="sorting not supported"):
        sess.call_function("sort_indices", [pa.array([True, False])], "order=ascending")


@pytest.mark.gpu
def test_sort_indices_record_batch(sess):
    """several sort keys over the columns of a record batch (TestVectorSortIndicesCppRecordBatchParity :1346,
    TestSortRecordBatch :554), and random data against Arrow C++'s table sort"""
    from arrow_go_amd import compute as ac
    nan = float("nan")
    a = pa.array([None, 1, 3, None, nan, nan, nan, 1], pa.float32()); b = pa.array([5, 3, None, None, None, nan, 5, 5], pa.float64())
    si = lambda cols, keys: sess.call_function("sort_indices", cols, "sort_keys=" + keys).to_pylist()
    assert si([a, b], "0:asc:at_end,1:desc:at_end") == [7, 1, 2, 6, 5, 4, 0, 3]
    assert si([a, b], "0:asc:at_start,1:desc:at_start") == [3, 0, 4, 5, 6, 7, 1, 2]
    u = pa.array([3, 1, 3, 0, 2, 1, 1], pa.uint8()); v = pa.array([5, 3, 4, 6, 5, 5, 3], pa.uint32())
    assert si([u, v], "0:asc:at_end,1:desc:at_end") == [3, 5, 1, 6, 4, 0, 2]
    assert si([v, u], "1:asc:at_end,0:desc:at_end") == [3, 5, 1, 6, 4, 0, 2]          # keys name their columns
    with pytest.raises(ac.ErrInvalid, match="sort key 1 has invalid column index 5"):
        si([u, v], "0:asc:at_end,5:desc:at_end")
    with pytest.raises(ac.ErrInvalid, match="same length"):
        si([u, pa.array([1, 2], pa.int8())], "0:asc:at_end,1:asc:at_end")
    rng = np.random.default_rng(2)
    n = 40009
    c0 = pa.array(rng.integers(0, 5, n), mask=rng.random(n) < 0.1, type=pa.int8())
    c1 = pa.array(rng.integers(0, 9, n).astype(np.float64), mask=rng.random(n) < 0.1)
    c2 = pa.array(rng.integers(0, 10**6, n), type=pa.int64())
    tbl = pa.table({"a": c0, "b": c1, "c": c2})
    for npl in ("at_end", "at_start"):
        got = si([c0, c1, c2], "0:asc:%s,1:desc:%s,2:asc:%s" % (npl, npl, npl))
        exp = pc.sort_indices(tbl, sort_keys=[("a", "ascending"), ("b", "descending"), ("c", "ascending")], null_placement=npl)
        assert got == exp.to_pylist(), npl


@pytest.mark.gpu
@pytest.mark.parametrize("typ", NUMERIC, ids=str)
def test_sort_indices_random_vs_arrow_cpp(sess, typ):
    # Arrow C++ sort_indices is stable with the same null / NaN placement rules
    rng = np.random.default_rng(typ.bit_width + 31)
    n = 30011
    vals = rng.integers(0, 50, n).astype(typ.to_pandas_dtype())
    a = pa.array(vals, mask=rng.random(n) < 0.1, type=typ).slice(9, n - 20)
    for order in ("ascending", "descending"):
        for npl in ("at_end", "at_start"):
            got = sess.call_function("sort_indices", [a], "order=%s;null_placement=%s" % (order, npl))
            exp = pc.sort_indices(a, sort_keys=[("x", order)], null_placement=npl) if False else pc.array_sort_indices(a, order=order, null_placement=npl)
            assert got.to_pylist() == exp.to_pylist(), (typ, order, npl)


# ---- var-length take / filter (vector_selection_test.go:698-704, 1193-1210) -----------------------------------
_BIN = [pa.string(), pa.binary(), pa.large_string(), pa.large_binary()]


@pytest.mark.gpu
@pytest.mark.parametrize("typ", _BIN, ids=str)
def test_take_filter_binary(sess, typ):
    from arrow_go_amd import compute as ac
    conv = (lambda v: v) if pa.types.is_string(typ) or pa.types.is_large_string(typ) else (lambda v: None if v is None else v.encode())
    A = lambda v: pa.array([conv(x) for x in v], type=typ)
    take = lambda v, i, it=pa.int32(): sess.call_function("take", [A(v), pa.array(i, type=it)])
    assert take(["a", "b", "c"], [0, 1, 0]).equals(A(["a", "b", "a"]))
    assert take([None, "b", "c"], [0, 1, 0]).equals(A([None, "b", None]))
    assert take(["a", "b", "c"], [None, 1, 0]).equals(A([None, "b", "a"]))
    with pytest.raises(ac.ErrIndex, match="9 out of bounds"):
        take(["a", "b", "c"], [0, 9, 0], pa.int8())
    with pytest.raises(ac.ErrIndex, match="5 out of bounds"):
        take(["a", "b", "c"], [2, 5], pa.int64())
    out = take(["a", "b", "c"], [0, 1, 0])
    assert out.type == typ and out.null_count == 0
    filt = lambda v, f, o="": sess.call_function("filter", [A(v), pa.array(f, type=pa.bool_())], o)
    assert filt(["a", "b", "c"], [False, True, False]).equals(A(["b"]))
    assert filt([None, "b", "c"], [False, True, False]).equals(A(["b"]))
    assert filt(["a", "b", "c"], [None, True, False], "null_selection_behavior=emit_null").equals(A([None, "b"]))
    assert filt(["a", "b", "c"], [None, True, False]).equals(A(["b"]))
    assert filt([], []).equals(A([]))
    # random, sliced, against Arrow C++
    rng = np.random.default_rng(5)
    n = 20011
    words = ["", "x", "hello", "a much longer value that spans many bytes " * 4, "ünïcödé", "zz"]
    vals = [None if rng.random() < 0.1 else words[int(rng.integers(0, len(words)))] + str(int(rng.integers(0, 1000))) for _ in range(n)]
    a = A(vals).slice(11, n - 30)
    idx = pa.array(rng.integers(0, len(a), 30000), mask=rng.random(30000) < 0.05, type=pa.int32())
    assert sess.call_function("take", [a, idx]).equals(pc.take(a, idx))
    m = pa.array(rng.random(len(a)) < 0.4, mask=rng.random(len(a)) < 0.1).slice(3)
    a2 = a.slice(3)
    assert sess.call_function("filter", [a2, m]).equals(pc.filter(a2, m))
    assert sess.call_function("filter", [a2, m], "null_selection_behavior=emit_null").equals(pc.filter(a2, m, null_selection_behavior="emit_null"))
    # sort = take(input, sort_indices(input)) needs a numeric key; a string column rides along through take
    keys = pa.array(rng.integers(0, 100, len(a)), type=pa.int32())
    order = sess.call_function("sort_indices", [keys], "order=ascending")
    assert sess.call_function("take", [a, order]).equals(pc.take(a, pc.array_sort_indices(keys)))


# ---- boolean-valued take / filter (vector_selection_test.go:369-377, TestTakeBoolean) ------------------------
@pytest.mark.gpu
def test_take_filter_boolean(sess):
    from arrow_go_amd import compute as ac
    B = lambda v: pa.array(v, type=pa.bool_())
    filt = lambda v, f, o="": sess.call_function("filter", [B(v), B(f)], o).to_pylist()
    assert filt([], []) == []
    assert filt([True, False, True], [False, True, False]) == [False]
    assert filt([None, False, True], [False, True, False]) == [False]
    assert filt([True, False, True], [None, True, False], "null_selection_behavior=emit_null") == [None, False]
    take = lambda v, i, it=pa.int32(): sess.call_function("take", [B(v), pa.array(i, type=it)]).to_pylist()
    assert take([True, False, True], [0, 1, 0]) == [True, False, True]
    assert take([None, False, True], [0, 1, 0]) == [None, False, None]
    assert take([True, False, True], [None, 1, 0]) == [None, False, True]
    with pytest.raises(ac.ErrIndex, match="out of bounds"):
        take([True, False, True], [0, 9, 0])
    # the case the reference's own boolean filter gets wrong (boolFilterWriter never advances):
    # several individually selected values in a mixed block — here it is simply right, and equals Arrow C++
    rng = np.random.default_rng(8)
    n = 70001
    v = pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1).slice(5)
    m = pa.array(rng.random(n - 5) < 0.3, mask=rng.random(n - 5) < 0.1)
    assert sess.call_function("filter", [v, m]).equals(pc.filter(v, m))
    assert sess.call_function("filter", [v, m], "null_selection_behavior=emit_null").equals(pc.filter(v, m, null_selection_behavior="emit_null"))
    idx = pa.array(rng.integers(0, len(v), 50000), mask=rng.random(50000) < 0.05, type=pa.uint32())
    assert sess.call_function("take", [v, idx]).equals(pc.take(v, idx))


# ---- arrow/math + fused -------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_math_sum(sess):
    # arrow/math/float64_test.go:30-48: Σ 0..9999, empty → 0; validity is ignored (float64.go:41-46)
    assert sess.math_sum(pa.array(np.arange(10000, dtype=np.float64))) == 49995000.0
    assert sess.math_sum(pa.array(np.arange(10000, dtype=np.int64))) == 49995000
    assert sess.math_sum(pa.array(np.arange(10000, dtype=np.uint64))) == 49995000
    assert sess.math_sum(pa.array([], pa.float64())) == 0.0
    assert sess.math_sum(pa.array(np.arange(100, dtype=np.int64)).slice(10, 5)) == 10 + 11 + 12 + 13 + 14


@pytest.mark.gpu
def test_fused_equals_three_calls(sess):
    rng = np.random.default_rng(5)
    n = 100003
    x = pa.array(rng.integers(-10**6, 10**6, n), mask=rng.random(n) < 0.1, type=pa.int64())
    thr = pa.scalar(1234, pa.int64())
    mask = sess.call_function("greater", [x, thr])
    kept = sess.call_function("filter", [x, mask])
    assert kept.null_count == 0
    assert sess.call_function("greater_filter_sum", [x, thr]).as_py() == sess.math_sum(kept) == pc.sum(kept).as_py()


@pytest.mark.gpu
def test_filter_worst_case_output_is_one_call(sess):
    """FilterOptions output_sizing=worst_case: the output is allocated for the input's length and the kernel runs in ONE call
    (ah_filter_primitive_once) instead of count → allocate → fill; same arrays as the default and as Arrow C++"""
    rng = np.random.default_rng(17)
    n = 300_007
    for typ in (pa.int64(), pa.float32(), pa.uint8()):
        for vnulls in (False, True):
            for fnulls in (False, True):
                v = pa.array(rng.integers(0, 200, n), mask=(rng.random(n) < 0.1) if vnulls else None, type=pa.int64()).cast(typ)
                f = pa.array(rng.random(n) < 0.5, mask=(rng.random(n) < 0.1) if fnulls else None)
                for sel in ("drop", "emit_null"):
                    want = pc.filter(v, f, null_selection_behavior=sel)
                    got = sess.call_function("filter", [v, f], options=f"null_selection_behavior={sel};output_sizing=worst_case")
                    base = sess.call_function("filter", [v, f], options=f"null_selection_behavior={sel}")
                    assert got.equals(want) and base.equals(want) and got.null_count == want.null_count, (typ, vnulls, fnulls, sel)


@pytest.mark.gpu
def test_unique_and_dictionary_encode_of_fixed_size_binary_and_decimals(sess):
    """FixedSizeBinary / Decimal128 / Decimal256 keys through the registry (kernels/vector_hash.go:608-609, 698: the BinaryMemoTable
    over values of one byte width → ah_hash_fixed_encode): first-seen order, both null encodings, == Arrow C++"""
    import decimal
    rng = np.random.default_rng(23)
    n = 40_000
    picks = rng.integers(0, 500, n)
    mask = rng.random(n) < 0.1
    cols = []
    for w in (16, 5, 1):
        vals = [int(p).to_bytes(8, "little").ljust(w, b"\x07")[:w] for p in picks]
        cols.append(pa.array([None if m else v for v, m in zip(vals, mask)], type=pa.binary(w)))
    cols.append(pa.array([None if m else decimal.Decimal(int(p)) / 1000 for p, m in zip(picks, mask)], type=pa.decimal128(20, 3)))
    cols.append(pa.array([None if m else decimal.Decimal(int(p) * 10**30) / 100000 for p, m in zip(picks, mask)], type=pa.decimal256(50, 5)))
    cols.append(pa.array([decimal.Decimal(int(p)) for p in picks], type=pa.decimal128(9, 0)))          # no nulls
    for col in cols:
        got = sess.call_function("unique", [col])
        want = pc.unique(col)
        assert got.type == col.type and got.equals(want), col.type
        for enc in ("mask", "encode"):
            g = sess.call_function("dictionary_encode", [col], options=f"null_encoding_behavior={enc}")
            w_ = pc.dictionary_encode(col, null_encoding=enc)
            assert g.type.value_type == col.type
            assert g.indices.equals(w_.indices.cast(pa.int32())) and g.dictionary.equals(w_.dictionary), (col.type, enc)
    # a sliced column (offset ≠ 0) and an empty one
    sl = cols[0].slice(1234, 5000)
    assert sess.call_function("unique", [sl]).equals(pc.unique(sl))
    empty = pa.array([], type=pa.binary(16))
    assert len(sess.call_function("unique", [empty])) == 0
    # nothing else is registered for these types: the reference's dispatch error, not a crash
    with pytest.raises(Exception, match="no kernel matching"):
        sess.call_function("add", [cols[3], cols[3]])


@pytest.mark.gpu
def test_take_and_filter_of_fixed_size_binary_and_decimals(sess):
    """FSBImpl (kernels/vector_selection.go:1997-2031, registered at :2344-2346 and :2354-2356): Decimal128 / Decimal256 and binaries
    of 16 / 32 bytes are 16- and 32-byte slots of the device take; binary(8) and narrower ride the primitive kernels; the filter of
    wide slots is GetTakeIndices + take.  == Arrow C++ with nulls on both sides, sliced inputs, every index type."""
    import decimal
    rng = np.random.default_rng(29)
    n = 30_011
    picks = rng.integers(0, 1 << 40, n)
    mask = rng.random(n) < 0.1
    cols = []
    for w in (32, 16, 8, 2):
        vals = [int(p).to_bytes(8, "little").ljust(w, b"\x05")[:w] for p in picks]
        cols.append(pa.array([None if m else v for v, m in zip(vals, mask)], type=pa.binary(w)))
        cols.append(pa.array(vals, type=pa.binary(w)))
    cols.append(pa.array([None if m else decimal.Decimal(int(p)) / 1000 for p, m in zip(picks, mask)], type=pa.decimal128(20, 3)))
    cols.append(pa.array([None if m else decimal.Decimal(int(p) * 10**30) / 100000 for p, m in zip(picks, mask)], type=pa.decimal256(50, 5)))
    for col in cols:
        for ityp, m in ((pa.int8(), 100), (pa.uint16(), 30_000), (pa.int32(), n), (pa.uint32(), n), (pa.int64(), n), (pa.uint64(), n)):
            idx = rng.integers(0, m, 20_003)
            for inulls in (False, True):
                ia = pa.array(idx, type=ityp, mask=(rng.random(len(idx)) < 0.2) if inulls else None)
                got = sess.call_function("take", [col, ia])
                want = pc.take(col, ia)
                assert got.type == col.type and got.equals(want) and got.null_count == want.null_count, (col.type, ityp, inulls)
        for fnulls in (False, True):
            f = pa.array(rng.random(n) < 0.4, mask=(rng.random(n) < 0.15) if fnulls else None)
            for sel in ("drop", "emit_null"):
                got = sess.call_function("filter", [col, f], options=f"null_selection_behavior={sel}")
                want = pc.filter(col, f, null_selection_behavior=sel)
                assert got.type == col.type and got.equals(want) and got.null_count == want.null_count, (col.type, fnulls, sel)
        sl, fs = col.slice(777, 9000), pa.array(rng.random(9000) < 0.5)
        assert sess.call_function("filter", [sl, fs]).equals(pc.filter(sl, fs))
        assert sess.call_function("take", [sl, pa.array([8999, 0, 5], type=pa.int32())]).equals(pc.take(sl, pa.array([8999, 0, 5])))
        with pytest.raises(Exception, match="out of bounds"):
            sess.call_function("take", [sl, pa.array([9000], type=pa.int32())])
        assert len(sess.call_function("filter", [col.slice(0, 0), pa.array([], type=pa.bool_())])) == 0
    # any other width goes byte by byte
    odd = pa.array([b"abcde", None, b"fghij"], type=pa.binary(5))
    assert sess.call_function("take", [odd, pa.array([2, 1, 0, None, 2], type=pa.int32())]).equals(pc.take(odd, pa.array([2, 1, 0, None, 2])))
    assert sess.call_function("filter", [odd, pa.array([True, None, True])], options="null_selection_behavior=emit_null").equals(
        pc.filter(odd, pa.array([True, None, True]), null_selection_behavior="emit_null"))


@pytest.mark.gpu
def test_take_fixed_size_binary_reference_table(sess):
    """compute/vector_selection_test.go:1173-1187 (TakeKernelTestFSB.TestFixedSizeBinary: binary(3), values "aaa" / "bbb" / "ccc"):
    the three assertTake rows, no validity bitmap without nulls, and the two out-of-bounds index errors (arrow.ErrIndex) — plus random
    columns of widths 3, 7 and 24 against Arrow C++ for take and filter."""
    t = pa.binary(3)
    vals = lambda xs: pa.array(xs, type=t)
    for v, i, want in ((["aaa", "bbb", "ccc"], [0, 1, 0], ["aaa", "bbb", "aaa"]),
                       ([None, "bbb", "ccc"], [0, 1, 0], [None, "bbb", None]),
                       (["aaa", "bbb", "ccc"], [None, 1, 0], [None, "bbb", "aaa"])):
        enc = lambda xs: [None if x is None else x.encode() for x in xs]
        for ityp in (pa.int8(), pa.uint8(), pa.int16(), pa.uint16(), pa.int32(), pa.uint32(), pa.int64(), pa.uint64()):   # assertTake runs every index type
            got = sess.call_function("take", [vals(enc(v)), pa.array(i, type=ityp)])
            assert got.type == t and got.equals(vals(enc(want))), (v, i, ityp)
    got = sess.call_function("take", [vals([b"aaa", b"bbb", b"ccc"]), pa.array([0, 1, 0], type=pa.int16())])
    assert got.null_count == 0 and got.buffers()[0] is None      # assertNoValidityBitmapUnknownNullCountJSON
    with pytest.raises(Exception, match="out of bounds"):
        sess.call_function("take", [vals([b"aaa", b"bbb", b"ccc"]), pa.array([0, 9, 0], type=pa.int8())])
    with pytest.raises(Exception, match="out of bounds"):
        sess.call_function("take", [vals([b"aaa", b"bbb", b"ccc"]), pa.array([2, 5], type=pa.int64())])
    rng = np.random.default_rng(41)
    n = 20_011
    for w in (3, 7, 24):
        raw = rng.integers(0, 256, (n, w), dtype=np.uint8)
        col = pa.array([None if m else bytes(r) for r, m in zip(raw, rng.random(n) < 0.1)], type=pa.binary(w))
        idx = pa.array(rng.integers(0, n, 15_001), type=pa.int32(), mask=rng.random(15_001) < 0.2)
        assert sess.call_function("take", [col, idx]).equals(pc.take(col, idx)), w
        f = pa.array(rng.random(n) < 0.4, mask=rng.random(n) < 0.15)
        for sel in ("drop", "emit_null"):
            got = sess.call_function("filter", [col, f], options=f"null_selection_behavior={sel}")
            assert got.equals(pc.filter(col, f, null_selection_behavior=sel)), (w, sel)
        sl = col.slice(1000, 5000)
        assert sess.call_function("take", [sl, pa.array([4999, 0, 17], type=pa.int32())]).equals(pc.take(sl, pa.array([4999, 0, 17])))


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [128, 256])
def test_filter_decimal_reference_table(sess, bits):
    """compute/vector_selection_test.go:656-689 (FilterKernelWithDecimal.TestFilterDecimalNumeric, run for Decimal128 and Decimal256
    at precision 3, scale 2) through the registry's filter: the table row by row with both null selections, the sliced mask, and
    the mask of another length refused (arrow.ErrInvalid)."""
    import decimal
    from tests.test_oracle_vs_reference import DECIMAL_FILTER_TABLE
    typ = pa.decimal128(3, 2) if bits == 128 else pa.decimal256(3, 2)
    arr = lambda vals: pa.array([None if v is None else decimal.Decimal(v) for v in vals], type=typ)
    for vals, mask, want in DECIMAL_FILTER_TABLE:
        v, f = arr(vals), pa.array(mask, type=pa.bool_())
        got = sess.call_function("filter", [v, f], options="null_selection_behavior=emit_null")
        assert got.type == typ and got.equals(arr(want)), (vals, mask)
        drop = [w for w, m in zip(want, [m for m in mask if m is None or m]) if m is not None]
        got = sess.call_function("filter", [v, f], options="null_selection_behavior=drop")
        assert got.equals(arr(drop)) and got.equals(pc.filter(v, f, null_selection_behavior="drop")), (vals, mask)
    val = arr(["7.12", "8.00", "9.87"])
    sliced = pa.array([False, True, True, True, False, True]).slice(3, 3)
    assert sess.call_function("filter", [val, sliced]).equals(arr(["7.12", "9.87"]))
    for sel in ("emit_null", "drop"):
        with pytest.raises(Exception, match="(?i)invalid|length"):
            sess.call_function("filter", [val, pa.array([], type=pa.bool_())], options=f"null_selection_behavior={sel}")


# ---- divide / abs / negate / bit-wise / shifts / sqrt through the registry ------------------------------------
@pytest.mark.gpu
def test_extended_arithmetic_functions(sess):
    from arrow_go_amd import compute as ac
    rng = np.random.default_rng(77)
    n = 20011
    for typ in NUMERIC:
        np_t = typ.to_pandas_dtype()
        isint = pa.types.is_integer(typ)
        if isint:
            info = np.iinfo(np_t)
            a = pa.array(rng.integers(info.min + 1, info.max, n, dtype=np_t, endpoint=True), mask=rng.random(n) < 0.1, type=typ)
            bnp = rng.integers(info.min + 1, info.max, n, dtype=np_t, endpoint=True); bnp[bnp == 0] = 1
            b = pa.array(bnp, mask=rng.random(n) < 0.1, type=typ)
        else:
            a = pa.array(rng.standard_normal(n).astype(np_t), mask=rng.random(n) < 0.1, type=typ)
            b = pa.array((rng.standard_normal(n) + 3).astype(np_t), mask=rng.random(n) < 0.1, type=typ)
        for name, ref in (("divide", pc.divide_checked), ("divide_unchecked", pc.divide)):
            assert sess.call_function(name, [a, b]).equals(ref(a, b)), (name, typ)
            assert sess.call_function(name, [a, pa.scalar(3, typ)]).equals(ref(a, pa.scalar(3, typ)))
        assert sess.call_function("abs", [a]).equals(pc.abs_checked(a))
        if isint:
            with pytest.raises(ac.ErrInvalid, match="divide by zero"):
                sess.call_function("divide_unchecked", [a, pa.array(np.zeros(n, np_t), type=typ)])
            for name, ref in (("bit_wise_and", pc.bit_wise_and), ("bit_wise_or", pc.bit_wise_or), ("bit_wise_xor", pc.bit_wise_xor)):
                assert sess.call_function(name, [a, b]).equals(ref(a, b)), (name, typ)
            assert sess.call_function("bit_wise_not", [a]).equals(pc.bit_wise_not(a))
            cnt = pa.array(rng.integers(0, info.bits - 1, n).astype(np_t), mask=rng.random(n) < 0.1, type=typ)
            for name, ref in (("shift_left", pc.shift_left_checked), ("shift_right", pc.shift_right_checked),
                              ("shift_left_unchecked", pc.shift_left), ("shift_right_unchecked", pc.shift_right)):
                assert sess.call_function(name, [a, cnt]).equals(ref(a, cnt)), (name, typ)
            with pytest.raises(ac.ErrInvalid, match="shift amount"):
                sess.call_function("shift_left", [a, pa.scalar(info.bits, typ)])
            if info.min < 0:
                assert sess.call_function("negate", [a]).equals(pc.negate_checked(a))
                with pytest.raises(ac.ErrInvalid, match="overflow"):
                    sess.call_function("abs", [pa.array([1, info.min], type=typ)])
            else:
                with pytest.raises(ac.ErrNotImplemented, match="no kernel matching"):
                    sess.call_function("negate", [a])
        else:
            assert sess.call_function("negate", [a]).equals(pc.negate_checked(a))
            p = pc.abs(a)
            assert sess.call_function("sqrt", [p]).equals(pc.sqrt_checked(p))
            got, exp = sess.call_function("sqrt_unchecked", [a]), pc.sqrt(a)
            assert got.is_valid().equals(exp.is_valid())
            assert np.array_equal(got.to_numpy(zero_copy_only=False), exp.to_numpy(zero_copy_only=False), equal_nan=True)
            with pytest.raises(ac.ErrInvalid, match="square root of negative"):
                sess.call_function("sqrt", [pa.array([4, -1], type=typ)])
            with pytest.raises(ac.ErrInvalid, match="divide by zero"):
                sess.call_function("divide", [a, pa.scalar(0, typ)])
    # implicit promotion, as for the other arithmetic functions
    assert sess.call_function("divide", [pa.array([7, 9], pa.int8()), pa.array([2, 3], pa.int32())]).equals(pa.array([3, 3], pa.int32()))
    assert sess.call_function("sqrt", [pa.array([4, 9, None], pa.int32())]).equals(pa.array([2.0, 3.0, None]))


@pytest.mark.gpu
def test_validity_and_rounding_functions(sess):
    """is_null / is_not_null / is_nan (scalar_compare.go:138-160) and floor / ceil / trunc (rounding.go:748-775)"""
    rng = np.random.default_rng(78)
    n = 10007
    cols = [pa.array(rng.integers(-5, 5, n), mask=rng.random(n) < 0.3, type=pa.int32()),
            pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.3, type=pa.float64()),
            pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.3),
            pa.array(["a", "bb", None, ""] * (n // 4) + ["x"] * (n % 4)),
            pa.array(rng.integers(0, 9, n), type=pa.uint8())]          # no validity buffer at all
    for c in cols:
        for v in (c, c.slice(13, 999), c.slice(0, 0)):                  # sliced inputs: bits are taken from the offset
            assert sess.call_function("is_null", [v]).equals(pc.is_null(v))
            assert sess.call_function("is_not_null", [v]).equals(pc.is_valid(v))
    f = pa.array([1.0, float("nan"), None, float("inf"), -0.0], type=pa.float64())
    assert sess.call_function("is_nan", [f]).to_pylist() == [False, True, False, False, False]     # NullNoOutput: no nulls in the result
    assert sess.call_function("is_nan", [f]).null_count == 0
    f32 = pa.array(np.array([np.nan, 1, np.nan], np.float32))
    assert sess.call_function("is_nan", [f32]).to_pylist() == [True, False, True]
    assert sess.call_function("is_nan", [cols[0]]).to_pylist() == [False] * n
    x = pa.array(rng.standard_normal(n) * 100, mask=rng.random(n) < 0.1, type=pa.float64())
    x32 = x.cast(pa.float32())
    for name, ref in (("floor", pc.floor), ("ceil", pc.ceil), ("trunc", pc.trunc)):
        assert sess.call_function(name, [x]).equals(ref(x))
        assert sess.call_function(name, [x32]).equals(ref(x32))
        assert sess.call_function(name, [cols[0]]).equals(ref(cols[0].cast(pa.float64())))        # integers go to float64


@pytest.mark.gpu
def test_round_functions(sess):
    """round / round_to_multiple (arithmetic.go:1036-1056) against Arrow C++ — the Go kernels are its transliteration"""
    from arrow_go_amd import compute as ac
    rng = np.random.default_rng(79)
    n = 20011
    x = pa.array(np.round(rng.standard_normal(n) * 100, 3), mask=rng.random(n) < 0.1, type=pa.float64())
    x32 = pa.array(np.round(rng.standard_normal(n) * 100, 2).astype(np.float32), mask=rng.random(n) < 0.1, type=pa.float32())
    modes = ["down", "up", "towards_zero", "towards_infinity", "half_down", "half_up", "half_towards_zero", "half_towards_infinity",
             "half_to_even", "half_to_odd"]
    assert sess.call_function("round", [x]).equals(pc.round(x))                               # defaults: ndigits 0, half to even
    assert sess.call_function("round_to_multiple", [x]).equals(pc.round_to_multiple(x))       # multiple 1.0
    for mode in modes:
        for nd in (-2, 0, 1, 2):
            for v in (x, x32):
                got = sess.call_function("round", [v], f"ndigits={nd};round_mode={mode}")
                assert got.equals(pc.round(v, ndigits=nd, round_mode=mode)), (mode, nd, v.type)
        for mult in ("double:0.25", "double:2", "int64:5"):
            m = float(mult.split(":")[1])
            got = sess.call_function("round_to_multiple", [x], f"multiple={mult};round_mode={mode}")
            assert got.equals(pc.round_to_multiple(x, multiple=m, round_mode=mode)), (mode, mult)
    ints = pa.array([1, 25, None, -15], type=pa.int32())                                     # integers are rounded as float64
    assert sess.call_function("round", [ints], "ndigits=-1;round_mode=half_up").equals(pa.array([0.0, 30.0, None, -10.0]))
    with pytest.raises(ac.ErrInvalid, match="must be positive"):
        sess.call_function("round_to_multiple", [x], "multiple=double:-1")
    with pytest.raises(ac.ErrInvalid, match="non-null"):
        sess.call_function("round_to_multiple", [x], "multiple=null:double")
    with pytest.raises(ac.ErrInvalid, match="overflow"):
        sess.call_function("round", [pa.array([1.7e308])], "ndigits=-308;round_mode=up")


# ---- dictionary arrays: dictionaryTake / dictionaryFilter (compute/selection.go:497-586), dictionaryHashState (vector_hash.go:505-576) ----
@pytest.mark.gpu
@pytest.mark.parametrize("index_type", [pa.int8(), pa.uint16(), pa.int32(), pa.int64()], ids=str)
def test_dictionary_arrays(sess, index_type):
    rng = np.random.default_rng(80)
    n = 20001
    for dictionary in (pa.array([10, 20, 30, 40], pa.int64()), pa.array(["foo", "bar", None, "quuux", ""])):
        idx = pa.array(rng.integers(0, len(dictionary), n), mask=rng.random(n) < 0.1, type=index_type)
        d = pa.DictionaryArray.from_arrays(idx, dictionary)
        back = sess.call_function("dictionary_encode", [d])                       # identity (vector_hash.go:473-478), and a C Data round trip
        assert back.type == d.type and back.equals(d)
        sel = pa.array(rng.integers(0, n, 5000), mask=rng.random(5000) < 0.1, type=pa.int32())
        got = sess.call_function("take", [d, sel])
        assert got.type == d.type and got.equals(d.take(sel)) and got.dictionary.equals(dictionary)
        mask = pa.array(rng.random(n) < 0.3, mask=rng.random(n) < 0.1)
        for opt, ns in (("", "drop"), ("null_selection_behavior=emit_null", "emit_null")):
            got = sess.call_function("filter", [d, mask], opt)
            assert got.equals(d.filter(mask, null_selection_behavior=ns)) and got.dictionary.equals(dictionary)
        uq = sess.call_function("unique", [d])                                      # vector_hash_test.go:452-500 TestDictionaryUnique
        exp = pc.unique(d)
        assert uq.type == d.type and uq.indices.equals(exp.indices) and uq.dictionary.equals(dictionary)
        # chunks of one dictionary array keep their dictionary
        parts = pa.chunked_array([d.slice(0, 700), d.slice(700)])
        got = sess.call_function("take", [parts, sel])
        assert got.combine_chunks().equals(d.take(sel))
    # vector_hash_test.go:452-500: indices [3,0,0,0,1,1,3,0,1,3,0,1] over [10,20,30,40] → [3,0,1] over the same dictionary
    d = pa.DictionaryArray.from_arrays(pa.array([3, 0, 0, 0, 1, 1, 3, 0, 1, 3, 0, 1], index_type), pa.array([10, 20, 30, 40], pa.int64()))
    uq = sess.call_function("unique", [d])
    assert uq.indices.to_pylist() == [3, 0, 1] and uq.dictionary.to_pylist() == [10, 20, 30, 40]


# ---- power / power_unchecked (arithmetic_test.go:482-510) -----------------------------------------------------------
@pytest.mark.gpu
def test_power_through_the_registry(sess):
    import pyarrow.compute as pc
    from arrow_go_amd import compute as ac
    for t in [pa.int8(), pa.uint8(), pa.int16(), pa.uint16(), pa.int32(), pa.uint32(), pa.int64(), pa.uint64()]:
        a, b = pa.array([None, 2, 3, None, 5, 0, 1], t), pa.array([1, 6, 2, 5, 1, 0, 0], t)   # every result (and every square) fits int8
        for fn, ref in (("power", pc.power_checked), ("power_unchecked", pc.power)):
            assert sess.call_function(fn, [a, b]).equals(ref(a, b)), (fn, t)
            assert sess.call_function(fn, [a, pa.scalar(2, t)]).equals(ref(a, pa.scalar(2, t)))
            assert sess.call_function(fn, [pa.scalar(3, t), pa.array([None, 3, 4, None, 2], t)]).to_pylist() == [None, 27, 81, None, 9]
        mx = pa.array([2 ** (t.bit_width - (1 if pa.types.is_signed_integer(t) else 0)) - 1], t)
        with pytest.raises(ac.ErrInvalid, match="overflow"):
            sess.call_function("power", [mx, pa.array([10], t)])
        assert sess.call_function("power_unchecked", [mx, pa.array([10], t)]).to_pylist() == [1]
        if pa.types.is_signed_integer(t):
            for fn in ("power", "power_unchecked"):
                with pytest.raises(ac.ErrInvalid, match="integers to negative integer powers are not allowed"):
                    sess.call_function(fn, [pa.array([2, 3], t), pa.array([1, -1], t)])
