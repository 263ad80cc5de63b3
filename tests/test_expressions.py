"""Row §8(f)-1: the expression executor as a fusion front-end.

compute.Expression trees (NewCall / NewFieldRef / NewLiteral, arrow/compute/expression.go:
596-620) are evaluated by the reference one kernel per call node (exprs/exec.go:542-700).
arrow_go_amd evaluates the same tree either that way (fuse=False) or as ONE hiprtc-compiled
kernel (fuse=True).  The two routes must agree BYTE FOR BYTE — values, validity, and the
payload under nulls — and agree logically with Arrow C++ (pyarrow.compute).
"""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

import arrow_go_amd as ah

N = ah._native


# ---- CPU: code generation + hiprtc compile for gfx950 (no GPU needed) -------------------------
def test_codegen_compiles_offline():
    prog = [(N.X_FIELD, 0), (N.X_FIELD, 1), (N.X_ADD_CHECKED, 0), (N.X_FIELD, 2), (N.X_MUL, 0), (N.X_LITERAL, 0), (N.X_GT, 0)]
    for t in (N.INT8, N.UINT16, N.INT32, N.UINT64, N.INT64, N.FLOAT32, N.FLOAT64):
        src, out_type = ah.expr_codegen(prog, [t] * 3, [t])
        assert out_type == N.BOOL and "ah_expr_kernel" in src
    # boolean tree over bitmap columns and a unary op
    src, out_type = ah.expr_codegen([(N.X_FIELD, 0), (N.X_FIELD, 1), (N.X_AND_NOT, 0), (N.X_INVERT, 0)], [N.BOOL, N.BOOL], [])
    assert out_type == N.BOOL
    src, out_type = ah.expr_codegen([(N.X_FIELD, 0), (N.X_ABS, 0), (N.X_FIELD, 0), (N.X_SIGN, 0), (N.X_MUL, 0)], [N.INT32], [])
    assert out_type == N.INT32
    # float contraction must stay off: a*b+c rounds twice like two kernels
    src, _ = ah.expr_codegen([(N.X_FIELD, 0), (N.X_FIELD, 1), (N.X_MUL, 0), (N.X_FIELD, 2), (N.X_ADD, 0)], [N.FLOAT64] * 3, [])
    assert "v2 = v0 * v1" in src


def test_codegen_cast_nodes():
    # the value-preserving casts DispatchBest may insert compile into the kernel …
    for frm, to in [(N.INT8, N.INT64), (N.UINT16, N.INT32), (N.UINT32, N.UINT64), (N.INT32, N.FLOAT64), (N.INT16, N.FLOAT32),
                    (N.UINT8, N.FLOAT32), (N.FLOAT32, N.FLOAT64), (N.INT64, N.INT64)]:
        src, out_type = ah.expr_codegen([(N.X_FIELD, 0), (N.X_CAST, to), (N.X_FIELD, 1), (N.X_ADD, 0)], [frm, to], [])
        assert out_type == to
    # … the ones that need the checked cast kernel do not
    for frm, to in [(N.INT64, N.INT32), (N.INT32, N.UINT32), (N.UINT32, N.INT32), (N.INT64, N.FLOAT64), (N.INT32, N.FLOAT32),
                    (N.FLOAT64, N.FLOAT32), (N.FLOAT64, N.INT64), (N.BOOL, N.INT8)]:
        with pytest.raises(ah.ErrNotImplemented):
            ah.expr_codegen([(N.X_FIELD, 0), (N.X_CAST, to)], [frm], [])


def test_codegen_rejects_what_the_reference_would_cast():
    with pytest.raises(ah.ErrNotImplemented, match="AH_X_CAST"):
        ah.expr_codegen([(N.X_FIELD, 0), (N.X_FIELD, 1), (N.X_ADD, 0)], [N.INT64, N.INT32], [])
    with pytest.raises(ah.ErrNotImplemented):
        ah.expr_codegen([(N.X_FIELD, 0), (N.X_FIELD, 1), (N.X_AND, 0)], [N.INT64, N.INT64], [])
    with pytest.raises(ah.ErrInvalid, match="stack"):
        ah.expr_codegen([(N.X_FIELD, 0), (N.X_ADD, 0)], [N.INT64], [])


# ---- GPU ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sess():
    from arrow_go_amd import compute as ac
    s = ac.Session(0)
    yield s
    s.close()


def raw_buffers(arr):
    """(validity bytes or None, data bytes) restricted to the array's logical range"""
    n, off = len(arr), arr.offset
    bufs = arr.buffers()
    w = arr.type.bit_width
    valid = None
    if bufs[0] is not None and arr.null_count:
        valid = np.unpackbits(np.frombuffer(bufs[0], np.uint8), bitorder="little")[off:off + n].tobytes()
    if w == 1:
        data = np.unpackbits(np.frombuffer(bufs[1], np.uint8), bitorder="little")[off:off + n].tobytes()
    else:
        data = np.frombuffer(bufs[1], np.uint8)[off * w // 8:(off + n) * w // 8].tobytes()
    return valid, data


EXPRS = [
    # (text, number of columns, literal values)
    ("greater(multiply_unchecked(add_unchecked($0,$1),$2),#0)", 3, [5]),
    ("and(greater($0,#0),less_equal($1,$2))", 3, [0]),
    ("subtract_unchecked(multiply_unchecked($0,$0),multiply_unchecked($1,#0))", 2, [3]),
    ("xor(equal($0,$1),invert(not_equal($2,#0)))", 3, [1]),
    ("add(abs_unchecked($0),sign($1))", 2, []),
    ("or(and_not(greater_equal($0,$1),less($1,$2)),equal($2,#0))", 3, [2]),
]


@pytest.mark.gpu
@pytest.mark.parametrize("typ", [pa.int8(), pa.uint16(), pa.int32(), pa.int64(), pa.uint64(), pa.float32(), pa.float64()], ids=str)
def test_fused_equals_per_call_execution(sess, typ):
    rng = np.random.default_rng(7)
    for n in [1, 63, 64, 65, 1000, 70001]:
        for null_p in (0.0, 0.2):
            cols = [pa.array(rng.integers(0, 6, n), mask=(rng.random(n) < null_p) if null_p else None, type=typ) for _ in range(3)]
            if n > 10:  # sliced inputs: value pointers and validity bit offsets differ per column
                cols = [pa.concat_arrays([pa.array([1] * (i + 1), type=typ), c]).slice(i + 1) for i, c in enumerate(cols)]
            for text, ncols, lits in EXPRS:
                if pa.types.is_unsigned_integer(typ) and ("abs" in text or "subtract" in text):
                    pass  # still well defined (wraparound) — keep
                literals = [pa.scalar(v, type=typ) for v in lits]
                fused, was_fused = sess.eval_expression(text, cols[:ncols], literals, fuse=True)
                plain, was_fused2 = sess.eval_expression(text, cols[:ncols], literals, fuse=False)
                assert was_fused and not was_fused2
                assert fused.type == plain.type and len(fused) == len(plain) == n
                fv, fd = raw_buffers(fused)
                pv, pd = raw_buffers(plain)
                assert fd == pd, (text, n, null_p, "payload bytes (incl. under nulls)")
                # validity: both absent, or same bits (an absent bitmap == all valid)
                assert (fv or b"\x01" * n) == (pv or b"\x01" * n), (text, n, null_p)


@pytest.mark.gpu
def test_fused_vs_arrow_cpp(sess):
    rng = np.random.default_rng(8)
    n = 50000
    a = pa.array(rng.integers(-100, 100, n), mask=rng.random(n) < 0.1, type=pa.int64())
    b = pa.array(rng.integers(-100, 100, n), mask=rng.random(n) < 0.1, type=pa.int64())
    c = pa.array(rng.integers(-100, 100, n), type=pa.int64())
    got, fused = sess.eval_expression("greater(multiply($0,add($1,$2)),#0)", [a, b, c], [pa.scalar(50, pa.int64())])
    assert fused
    assert got.equals(pc.greater(pc.multiply_checked(a, pc.add_checked(b, c)), 50))
    x = pa.array(rng.uniform(-1, 1, n), mask=rng.random(n) < 0.1)
    y = pa.array(rng.uniform(-1, 1, n))
    got, fused = sess.eval_expression("add(multiply($0,$1),$0)", [x, y])
    assert fused and got.equals(pc.add(pc.multiply(x, y), x))  # bit-equal: no FMA contraction
    got, fused = sess.eval_expression("and(less($0,$1),greater($1,#0))", [x, y], [pa.scalar(0.25)])
    assert got.equals(pc.and_(pc.less(x, y), pc.greater(y, 0.25)))


@pytest.mark.gpu
def test_fused_checked_overflow_and_null_literal(sess):
    from arrow_go_amd import compute as ac
    mx = np.iinfo(np.int64).max
    a = pa.array([1, mx, None, 4], pa.int64())
    b = pa.array([1, 1, mx, None], pa.int64())
    for fuse in (True, False):
        with pytest.raises(ac.ErrInvalid, match="overflow"):
            sess.eval_expression("greater(add($0,$1),#0)", [a, b], [pa.scalar(0, pa.int64())], fuse=fuse)
    # the overflowing pair sits under a null → no error, and unchecked never errors
    a2 = pa.array([1, None, None, 4], pa.int64())
    for fuse in (True, False):
        got, _ = sess.eval_expression("add($0,$1)", [a2, b], fuse=fuse)
        assert got.to_pylist() == [2, None, None, None]
        got, _ = sess.eval_expression("add_unchecked($0,$1)", [a, b], fuse=fuse)
        assert got.to_pylist() == [2, -(2**63), None, None]
        got, _ = sess.eval_expression("greater($0,#0)", [a], [pa.scalar(None, pa.int64())], fuse=fuse)
        assert got.to_pylist() == [None] * 4


@pytest.mark.gpu
def test_fused_implicit_promotion(sess):
    """DispatchBest's implicit casts (arithmetic.go:112-142, scalar_compare.go:37-63) inside the fused kernel: column
    casts that cannot fail run as AH_X_CAST nodes, scalar operands are safe-cast once on the host; fused ==
    per-call, byte for byte, and equal to Arrow C++"""
    from arrow_go_amd import compute as ac
    rng = np.random.default_rng(9)
    n = 30011
    i8 = pa.array(rng.integers(-100, 100, n), mask=rng.random(n) < 0.1, type=pa.int8())
    u16 = pa.array(rng.integers(0, 60000, n), mask=rng.random(n) < 0.1, type=pa.uint16())
    i32 = pa.array(rng.integers(-10**6, 10**6, n), type=pa.int32())
    i64 = pa.array(rng.integers(-10**9, 10**9, n), mask=rng.random(n) < 0.1, type=pa.int64())
    f32 = pa.array(rng.uniform(-1, 1, n).astype(np.float32), type=pa.float32())
    f64 = pa.array(rng.uniform(-1, 1, n), mask=rng.random(n) < 0.1, type=pa.float64())
    cases = [
        ("add($0,$1)", [i8, i64], [], pc.add_checked(i8.cast(pa.int64()), i64)),                       # int8 + int64 → int64
        ("multiply_unchecked($0,$1)", [u16, i32], [], pc.multiply(u16.cast(pa.int32()), i32)),        # uint16 · int32 → int32
        ("add($0,$1)", [i8, u16], [], pc.add_checked(i8.cast(pa.int32()), u16.cast(pa.int32()))),    # int8 + uint16 → int32
        ("greater($0,$1)", [i32, f64], [], pc.greater(i32.cast(pa.float64()), f64)),                  # int32 vs double → double
        ("add($0,$1)", [f32, f64], [], pc.add(f32.cast(pa.float64()), f64)),                          # float + double
        ("subtract($0,$1)", [i8, f32], [], pc.subtract(i8.cast(pa.float32()), f32)),                  # int8 − float → float
        ("greater(add($0,$1),#0)", [i8, i64], [pa.scalar(7, pa.int32())], pc.greater(pc.add_checked(i8.cast(pa.int64()), i64), 7)),
        ("multiply($0,#0)", [f64], [pa.scalar(3, pa.int64())], pc.multiply(f64, 3.0)),                # scalar int64 → double on the host
        ("less($0,#0)", [i32], [pa.scalar(12345678901, pa.int64())], pc.less(i32.cast(pa.int64()), 12345678901)),
    ]
    for text, cols, lits, exp in cases:
        got, fused = sess.eval_expression(text, cols, lits)
        ref, ref_fused = sess.eval_expression(text, cols, lits, fuse=False)
        assert fused and not ref_fused, text
        assert got.type == exp.type and got.equals(exp), text
        assert got.equals(ref) and got.buffers()[1].equals(ref.buffers()[1]), text   # same bytes, null slots included
    # a scalar whose safe cast fails gives the per-call error, fused or not
    for fuse in (True, False):
        with pytest.raises(ac.ErrInvalid):
            sess.eval_expression("add($0,#0)", [f64], [pa.scalar(2**53 + 1, pa.int64())], fuse=fuse)


@pytest.mark.gpu
def test_unfusible_trees_fall_back(sess):
    # a Kleene node is not in the fused op set: the tree runs per call and still works
    a = pa.array([True, None, False]); b = pa.array([None, True, False])
    got, fused = sess.eval_expression("and_kleene($0,invert($1))", [a, b])
    assert not fused and got.equals(pc.and_kleene(a, pc.invert(b)))
    from arrow_go_amd import compute as ac
    # mixed operand types whose promotion could fail its safe-cast check (int64 → double is exact only below
    # 2^53) are not fused → per-call execution, where DispatchBest casts both sides exactly like the reference
    mixed, fused = sess.eval_expression("add($0,$1)", [pa.array([1, None], pa.int64()), pa.array([1.5, 2.0], pa.float64())])
    assert not fused and mixed.type == pa.float64() and mixed.to_pylist() == [2.5, None]
    with pytest.raises(ac.ErrInvalid):
        sess.eval_expression("add($0,$1)", [pa.array([2**53 + 1], pa.int64()), pa.array([1.5], pa.float64())])
    with pytest.raises(ac.ErrKey, match="not found"):
        sess.eval_expression("frobnicate($0)", [pa.array([1])])
    with pytest.raises(ac.ErrInvalid, match="out of range"):
        sess.eval_expression("add($0,$5)", [pa.array([1])])


# ---- the front end: the reference's tree shape and Substrait (arrow/compute/exprs/exec.go:440-700) ----------------------------
from tests.substrait_builder import SB, ARITH, CMP, BOOLF   # the protobuf writer of these tests (shared with tests/test_substrait_reader.py)


@pytest.mark.gpu
def test_substrait_comparisons(sess):
    """exprs/exec_test.go:132-216 TestComparisons: fn(arg1, arg2) over a struct SCALAR of two int32 → a Boolean scalar (the uuid cases
    need an extension type over FixedSizeBinary, which this layer does not carry)"""
    zero, one, two = (pa.scalar(v, pa.int32()) for v in (0, 1, 2))

    def expect(fn, arg1, arg2, res):
        for name in (fn, fn + ":any_any"):                       # substrait-go's compound names carry the signature
            b = SB([("arg1", "i32"), ("arg2", "i32")])
            msg = b.build(b.call(CMP, name, b.field(0), b.field(1)))
            for fuse in (True, False):
                out, _ = sess.eval_substrait(msg, [arg1, arg2], fuse=fuse)
                assert isinstance(out, pa.Scalar) and out.type == pa.bool_() and out.as_py() is res, (fn, arg1, arg2)

    expect("equal", one, one, True)
    expect("equal", one, two, False)
    expect("lt", one, two, True)            # the test writes "less": substrait-go resolves it to the comparison set's `lt`
    expect("lt", one, zero, False)
    expect("gt", one, zero, True)
    expect("gt", one, two, False)
    expect("lte", one, one, True)
    expect("gte", zero, one, False)
    expect("not_equal", one, two, True)


SFC_INPUT = {"a": [6.125, 0.0, -1.0], "b": [3.375, 1.0, 4.75]}


@pytest.mark.gpu
def test_substrait_execute_scalar_func_call(sess):
    """exprs/exec_test.go:360-440 TestExecuteScalarFuncCall: "add" a + 3.5 → [9.625, 3.5, 2.5]; "add sub" a + (3.5 − b) →
    [6.25, 2.5, -2.25]; "add nested" references struct children, which this layer has no column type for → ErrNotImplemented"""
    from arrow_go_amd import compute as ac
    a, bcol = pa.array(SFC_INPUT["a"]), pa.array(SFC_INPUT["b"])
    b = SB([("a", "fp64"), ("b", "fp64")])
    add = b.build(b.call(ARITH, "add", b.field(0), b.lit("fp64", 3.5), out_type="fp64"))
    add_sub = b.build(b.call(ARITH, "add", b.field(0), b.call(ARITH, "subtract", b.lit("fp64", 3.5), b.field(1), out_type="fp64"), out_type="fp64"))
    for fuse in (True, False):
        got, fused = sess.eval_substrait(add, [a, bcol], fuse=fuse)
        assert got.equals(pa.array([9.625, 3.5, 2.5])) and fused == fuse
        got, fused = sess.eval_substrait(add_sub, [a, bcol], fuse=fuse)
        assert got.equals(pa.array([6.25, 2.5, -2.25])) and fused == fuse
    nested = SB([("a", "fp64")])     # (the outer field's real type is a struct; the reference is refused before the schema matters)
    msg = nested.build(nested.call(ARITH, "add", nested.field(0, child=0), nested.field(0, child=1), out_type="fp64"))
    with pytest.raises(ac.ErrNotImplemented, match="nested field references"):
        sess.eval_substrait(msg, [a])
    # the same two trees in the reference's own Expression shape (expression.go:52-78: Call{name, args} / field reference / Literal)
    tree = ("call", "add_unchecked", [("field", "a"), ("lit", pa.scalar(3.5))])
    got, fused = sess.eval_expression_tree(tree, [a, bcol], names=["a", "b"])
    assert fused and got.equals(pa.array([9.625, 3.5, 2.5]))
    tree = ("call", "add_unchecked", [("field", 0), ("call", "subtract_unchecked", [("lit", pa.scalar(3.5)), ("field", "b")])])
    got, fused = sess.eval_expression_tree(tree, [a, bcol], names=["a", "b"], fuse=False)
    assert not fused and got.equals(pa.array([6.25, 2.5, -2.25]))
    with pytest.raises(ac.ErrInvalid, match="no match for field reference 'zz'"):
        sess.eval_expression_tree(("call", "add", [("field", "zz"), ("field", 0)]), [a, bcol], names=["a", "b"])
    with pytest.raises(ac.ErrInvalid, match="mismatched length"):     # makeExecBatch: every array column has the batch's length
        sess.eval_expression_tree(("call", "add_unchecked", [("field", 0), ("field", 1)]), [a, pa.array([1.0, 2.0])], names=["a", "b"])


BORING = [("in", pa.bool_()), ("bool", pa.bool_()), ("i8", pa.int8()), ("i32", pa.int32()), ("u32", pa.uint32()), ("i64", pa.int64()), ("f32", pa.float32()),
          ("f64", pa.float64()), ("date32", pa.date32()), ("str", pa.string()), ("bin", pa.binary())]       # boringArrowSchema + "in" (exec_test.go:73-84, 442-444)
BORING_SB = [("in", "bool"), ("bool", "bool"), ("i8", "i8"), ("i32", "i32"), ("u32", "u32"), ("i64", "i64"), ("f32", "fp32"), ("f64", "fp64"),
             ("date32", "date"), ("str", "string"), ("bin", "binary")]


@pytest.mark.gpu
def test_substrait_generate_mask(sess):
    """exprs/exec_test.go:441-500 TestGenerateMask: the filter expression over the record must equal its "in" column.  The record has
    the eleven fields of the reference's schema; only the referenced ones are supplied (by name) — makeExecBatch fills the others"""
    simple = {"i32": [0, 0, 1, 2, 0, 0, 0], "f32": [-0.1, 0.3, 0.2, -0.1, 0.1, None, 1.0], "in": [True, True, False, False, True, True, True]}
    complex_ = {"f64": [0.3, -0.1, 0.1, 0.0, 1.0, -2.0, 3.0], "f32": [0.1, 0.3, 0.2, -0.1, 0.1, None, 1.0], "in": [True, False, True, False, True, None, True]}
    b = SB(BORING_SB)
    idx = {n: i for i, (n, _) in enumerate(BORING_SB)}
    f_simple = b.build(b.call(CMP, "equal", b.field(idx["i32"]), b.lit("i32", 0)))
    f_complex = b.build(b.call(CMP, "gt", b.call(ARITH, "multiply", b.cast("fp64", b.field(idx["f32"])), b.field(idx["f64"]), out_type="fp64"), b.lit("fp64", 0.0)))
    schema = dict(BORING)
    for rows, msg in ((simple, f_simple), (complex_, f_complex)):
        names = [n for n in rows if n != "in"]
        cols = [pa.array(rows[n], type=schema[n]) for n in names]
        for fuse in (True, False):
            mask, _ = sess.eval_substrait(msg, cols, names=names, fuse=fuse)
            assert mask.equals(pa.array(rows["in"], pa.bool_())), (names, fuse, mask)
    # the same filters serialized by Arrow C++ (an independent producer of the wire format; unsigned types as user-defined types there)
    import pyarrow.substrait as ps
    sch = pa.schema(BORING)
    e_simple = pc.equal(pc.field("i32"), pc.scalar(pa.scalar(0, pa.int32())))
    e_complex = pc.greater(pc.multiply(pc.field("f32").cast(pa.float64(), safe=False), pc.field("f64")), pc.scalar(0.0))
    for rows, e in ((simple, e_simple), (complex_, e_complex)):
        msg = bytes(ps.serialize_expressions([e], ["out"], sch))
        names = [n for n in rows if n != "in"]
        mask, _ = sess.eval_substrait(msg, [pa.array(rows[n], type=schema[n]) for n in names], names=names)
        assert mask.equals(pa.array(rows["in"], pa.bool_()))


@pytest.mark.gpu
def test_substrait_from_arrow_cpp_matches_arrow_cpp(sess):
    """expressions serialized by pyarrow.substrait (Arrow C++'s producer), evaluated here and by Arrow C++ itself over the same random
    columns with nulls: equal results, and the fused route == the per-call route byte for byte"""
    import pyarrow.dataset as ds
    import pyarrow.substrait as ps
    rng = np.random.default_rng(31)
    n = 20011
    tbl = pa.table({
        "i32": pa.array(rng.integers(-1000, 1000, n), mask=rng.random(n) < 0.1, type=pa.int32()),
        "j32": pa.array(rng.integers(-1000, 1000, n), type=pa.int32()),
        "i64": pa.array(rng.integers(-10**9, 10**9, n), mask=rng.random(n) < 0.1, type=pa.int64()),
        "u32": pa.array(rng.integers(0, 4 * 10**9, n), mask=rng.random(n) < 0.1, type=pa.uint32()),
        "f64": pa.array(rng.uniform(-1, 1, n), mask=rng.random(n) < 0.1),
        "g64": pa.array(rng.uniform(-1, 1, n)),
        "b": pa.array(rng.random(n) < 0.5, mask=rng.random(n) < 0.1),
        "c": pa.array(rng.random(n) < 0.5),
    })
    F = pc.field
    exprs = [
        pc.add(F("i32"), F("j32")),                                             # overflow SILENT → add_unchecked
        pc.add_checked(F("i32"), F("j32")),                                     # overflow ERROR → add
        pc.greater(pc.multiply(pc.add(F("f64"), F("g64")), F("g64")), pc.scalar(0.25)),
        pc.and_kleene(pc.less(F("i64"), pc.scalar(pa.scalar(0, pa.int64()))), pc.invert(F("b"))),
        pc.or_kleene(F("b"), pc.greater_equal(F("f64"), F("g64"))),
        pc.less_equal(F("u32"), pc.scalar(pa.scalar(3_000_000_000, pa.uint32()))),   # an unsigned literal (user-defined literal on the wire)
        pc.subtract(F("i64"), pc.scalar(pa.scalar(-5, pa.int64()))),            # a negative literal: ten-byte varint
        pc.is_null(F("f64")),
        pc.not_equal(F("c"), F("b")),
        pc.multiply(F("i32").cast(pa.float64(), safe=False), F("f64")),
        pc.equal(F("i32"), pc.scalar(pa.scalar(None, pa.int32()))),             # a typed null literal
    ]
    for e in exprs:
        msg = bytes(ps.serialize_expressions([e], ["out"], tbl.schema))
        want = ds.dataset(tbl).to_table(columns={"out": e})["out"].combine_chunks()
        got, fused = sess.eval_substrait(msg, [c.combine_chunks() for c in tbl.columns], fuse=True)
        ref, _ = sess.eval_substrait(msg, [c.combine_chunks() for c in tbl.columns], fuse=False)
        assert got.type == want.type and got.equals(want), str(e)
        assert got.equals(ref) and got.buffers()[1].equals(ref.buffers()[1]), str(e)


@pytest.mark.gpu
def test_substrait_what_the_reference_refuses(sess):
    """the error CLASS of every refusal follows exprs/exec.go"""
    from arrow_go_amd import compute as ac
    a = pa.array([1, 2, 3], pa.int32())
    b = SB([("a", "i32"), ("u", "u32")])
    ok = b.call(CMP, "equal", b.field(0), b.lit("i32", 2))
    with pytest.raises(ac.ErrInvalid, match="no referred expression"):                   # exec.go:473
        sess.eval_substrait(b.build(), [a])
    with pytest.raises(ac.ErrNotImplemented, match="only single referred expression"):   # :480
        sess.eval_substrait(b.build(ok, ok), [a])
    with pytest.raises(ac.ErrNotImplemented, match="measures"):                          # :477
        sess.eval_substrait(b.build(ok, measure=True), [a])
    with pytest.raises(ac.ErrInvalid, match="referenced field a was int64, but should have been int32"):   # :533
        sess.eval_substrait(b.build(ok), [pa.array([1, 2, 3], pa.int64())])
    with pytest.raises(ac.ErrNotImplemented, match="modulus"):                            # :606-609: not in the default extension set
        sess.eval_substrait(b.build(b.call(ARITH, "modulus", b.field(0), b.field(0), out_type="i32")), [a])
    with pytest.raises(ac.ErrNotImplemented, match="SATURATE"):                           # types.go:172-192
        sess.eval_substrait(b.build(b.call(ARITH, "add", b.field(0), b.field(0), options={"overflow": ["SATURATE"]}, out_type="i32")), [a])
    with pytest.raises(ac.ErrInvalid, match="cast behavior unspecified"):                 # exec.go:573
        sess.eval_substrait(b.build(b.cast("i64", b.field(0), behavior=0)), [a])
    with pytest.raises(ac.ErrNotImplemented, match="cast behavior return nil"):           # :575
        sess.eval_substrait(b.build(b.cast("i64", b.field(0), behavior=1)), [a])
    with pytest.raises(ac.ErrInvalid, match="outside the base schema"):                   # :512-514
        sess.eval_substrait(b.build(b.call(CMP, "equal", b.field(7), b.field(0))), [a])
    with pytest.raises(ac.ErrInvalid, match="malformed"):
        sess.eval_substrait(b.build(ok)[:-3], [a])
    # overflow: ERROR is the checked kernel, SILENT (and no option at all) the wrapping one; SATURATE, ERROR → the first implemented one
    big = pa.array([2**31 - 1, 1], pa.int32())
    sb = SB([("a", "i32")])
    for opts, checked in (({"overflow": ["ERROR"]}, True), ({"overflow": ["SILENT"]}, False), (None, False), ({"overflow": ["SATURATE", "ERROR"]}, True)):
        msg = sb.build(sb.call(ARITH, "add", sb.field(0), sb.field(0), options=opts, out_type="i32"))
        if checked:
            with pytest.raises(ac.ErrInvalid, match="overflow"):
                sess.eval_substrait(msg, [big])
        else:
            got, _ = sess.eval_substrait(msg, [big])
            assert got.to_pylist() == [-2, 2]
    # arrow-go's unsigned convention: u32 is a type VARIATION of i32 (exprs/types.go:58-78) — column type and literal
    u = pa.array([5, 4_000_000_000, None], pa.uint32())
    got, _ = sess.eval_substrait(b.build(b.call(CMP, "gt", b.field(1), b.lit("u32", 3_000_000_000))), [a, u])
    assert got.to_pylist() == [False, True, None]
    # a cast that THROW_EXCEPTION turns into compute.UnsafeCastOptions (exec.go:571): 3.7 → 3, no error
    sbf = SB([("x", "fp64")])
    got, _ = sess.eval_substrait(sbf.build(sbf.cast("i32", sbf.field(0))), [pa.array([3.7, -1.2, None])])
    assert got.to_pylist() == [3, -1, None]
