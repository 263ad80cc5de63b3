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
