"""Known-answer vectors from the reference's own tests, run against
BOTH the CPU oracle (pins the oracle: `-m "not gpu"`) and the HIP library through the
C ABI (`-m gpu`).  Citations are arrow-go paths.
"""
import itertools

import numpy as np
import pytest

from tests import oracle_lib as OL
from tests.backends import OracleBackend, HipBackend, STATUS_OK, STATUS_EINVALID, STATUS_EINDEX, STATUS_EOVERFLOW

ADD, SUB, MUL, ABS, NEG, SIGN = 0, 1, 2, 4, 5, 20
ADD_C, SUB_C, MUL_C = 21, 22, 23
EQ, NE, GT, GE = 0, 1, 2, 3
AA, AS, SA = 0, 1, 2
AND, OR, XOR, ANDNOT, XNOR = 0, 1, 2, 3, 4
DROP, EMIT = 0, 1


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    if request.param == "oracle":
        return OracleBackend()
    return HipBackend(request.getfixturevalue("ctx"))


def mk(vals, dtype, null_fill=0):
    """python list with None → (values array, validity bitmap or None)"""
    valid = [v is not None for v in vals]
    arr = np.array([null_fill if v is None else v for v in vals], dtype=dtype)
    return arr, (None if all(valid) else OL.pack_bits(valid))


def logical(values, validbits, off, n):
    """values + validity → python list with None (array.ApproxEqual-style compare)"""
    if validbits is None:
        return list(values[:n].tolist())
    v = OL.unpack_bits(validbits, off, n)
    return [x if ok else None for x, ok in zip(values[:n].tolist(), v)]


# ---- arrow/math Sum -----------------------------------------------------------------------
# arrow/math/float64_test.go:30-48, int64_test.go, uint64_test.go: Σ 0..9999 = 49995000; empty → 0
@pytest.mark.parametrize("dtype", [np.float64, np.int64, np.uint64])
def test_sum_known_answer(be, dtype):
    a = np.arange(10000, dtype=dtype)
    assert be.sum(a) == 49995000
    assert be.sum(np.zeros(0, dtype=dtype)) == 0


# README.md:95-141 benchmark shape (BenchmarkFloat64Funcs_Sum_8192 — config C1)
def test_sum_8192(be):
    a = np.arange(8192, dtype=np.float64)
    assert be.sum(a) == 8191 * 8192 / 2


# ---- arithmetic ---------------------------------------------------------------------------
# arrow/compute/arithmetic_test.go:325-359 (BinaryArithmeticSuite.TestAdd), value payloads;
# null propagation is the executor's BitmapAnd, tested below in the bitmap section.
@pytest.mark.parametrize("dtype", OL.ALL_DTYPES, ids=str)
def test_add_vectors(be, dtype):
    t = lambda x: np.array(x, dtype=dtype)
    np.testing.assert_array_equal(be.arithmetic(ADD, AA, t([3, 2, 6]), t([1, 0, 2])), t([4, 2, 8]))
    np.testing.assert_array_equal(be.arithmetic(ADD, SA, t([3]), t([1, 2])), t([4, 5]))
    np.testing.assert_array_equal(be.arithmetic(ADD, AS, t([1, 2]), t([3])), t([4, 5]))
    np.testing.assert_array_equal(be.arithmetic(SUB, AA, t([3, 2, 6]), t([1, 0, 2])), t([2, 2, 4]))
    np.testing.assert_array_equal(be.arithmetic(SUB, SA, t([3]), t([1, 2])), t([2, 1]))
    np.testing.assert_array_equal(be.arithmetic(SUB, AS, t([4, 5]), t([3])), t([1, 2]))
    np.testing.assert_array_equal(be.arithmetic(MUL, AA, t([3, 2, 6]), t([1, 0, 2])), t([3, 0, 12]))
    np.testing.assert_array_equal(be.arithmetic(MUL, SA, t([3]), t([1, 2])), t([3, 6]))
    np.testing.assert_array_equal(be.arithmetic(MUL, AS, t([1, 2]), t([3])), t([3, 6]))


# arithmetic_test.go:325-359: checked add, nulls → payload 0 (helpers.go:303-306); [max]+[max] → "overflow"
@pytest.mark.parametrize("dtype", OL.INT_DTYPES, ids=str)
def test_add_checked_vectors(be, dtype):
    info = np.iinfo(dtype)
    l, lv = mk([None, 1, None], dtype, null_fill=7)
    r, rv = mk([3, 4, 5], dtype)
    st, out = be.arithmetic_checked(ADD_C, AA, l, lv, 0, r, rv, 0)
    assert st == STATUS_OK
    np.testing.assert_array_equal(out, np.array([0, 5, 0], dtype=dtype))
    l, lv = mk([None, 1, 2], dtype, null_fill=9)
    r, rv = mk([3, 4, None], dtype, null_fill=9)
    st, out = be.arithmetic_checked(ADD_C, AA, l, lv, 0, r, rv, 0)
    assert st == STATUS_OK
    np.testing.assert_array_equal(out, np.array([0, 5, 0], dtype=dtype))
    m = np.array([info.max], dtype=dtype)
    st, _ = be.arithmetic_checked(ADD_C, AA, m, None, 0, m, None, 0)
    assert st == STATUS_EOVERFLOW
    # overflow hidden under a null slot is not an error (only valid slots are tested)
    st, out = be.arithmetic_checked(ADD_C, AA, m, OL.pack_bits([False]), 0, m, None, 0)
    assert st == STATUS_OK and out[0] == 0
    # scalar shapes
    st, out = be.arithmetic_checked(ADD_C, SA, np.array([3], dtype=dtype), None, 0, *mk([None, 2], dtype), 0)
    assert st == STATUS_OK
    np.testing.assert_array_equal(out, np.array([0, 5], dtype=dtype))
    st, out = be.arithmetic_checked(SUB_C, AS, np.array([5, 7], dtype=dtype), None, 0, np.array([3], dtype=dtype), None, 0)
    assert st == STATUS_OK
    np.testing.assert_array_equal(out, np.array([2, 4], dtype=dtype))
    st, _ = be.arithmetic_checked(SUB_C, AA, np.array([info.min], dtype=dtype), None, 0, np.array([1], dtype=dtype), None, 0)
    assert st == STATUS_EOVERFLOW
    st, _ = be.arithmetic_checked(MUL_C, AA, m, None, 0, np.array([2], dtype=dtype), None, 0)
    assert st == STATUS_EOVERFLOW
    st, out = be.arithmetic_checked(MUL_C, AA, np.array([3, 2], dtype=dtype), None, 0, np.array([5, 7], dtype=dtype), None, 0)
    assert st == STATUS_OK
    np.testing.assert_array_equal(out, np.array([15, 14], dtype=dtype))


def test_checked_add_reference_carry_quirk(be):
    """kernels/base_arithmetic.go:249-263: `carry > 0` after an arithmetic shift by
    bits-2 — MinInt64+MinInt64 (carry bit 63 set) is NOT reported; restated as is."""
    mn = np.array([np.iinfo(np.int64).min], dtype=np.int64)
    st, out = be.arithmetic_checked(ADD_C, AA, mn, None, 0, mn, None, 0)
    assert st == STATUS_OK and out[0] == 0
    st, _ = be.arithmetic_checked(ADD_C, AA, np.array([-1], np.int64), None, 0, mn, None, 0)
    assert st == STATUS_OK  # same quirk: top carry bit set
    um = np.array([np.iinfo(np.uint64).max], dtype=np.uint64)
    st, _ = be.arithmetic_checked(ADD_C, AA, um, None, 0, np.array([1], np.uint64), None, 0)
    assert st == STATUS_EOVERFLOW


# arithmetic_test.go (UnaryArithmeticSuite): abs / negate / sign value semantics
@pytest.mark.parametrize("dtype", OL.ALL_DTYPES, ids=str)
def test_unary_vectors(be, dtype):
    dt = np.dtype(dtype)
    if dt.kind == "u":
        a = np.array([0, 1, 10, 127], dtype=dtype)
        np.testing.assert_array_equal(be.arithmetic_unary(ABS, a), a)
        np.testing.assert_array_equal(be.arithmetic_unary(SIGN, a), np.array([0, 1, 1, 1], dtype=dtype))
        np.testing.assert_array_equal(be.arithmetic_unary(NEG, a), (np.zeros_like(a) - a).astype(dtype))
    else:
        a = np.array([0, 1, -1, 10, -10, 127, -127], dtype=dtype)
        np.testing.assert_array_equal(be.arithmetic_unary(ABS, a), np.abs(a))
        np.testing.assert_array_equal(be.arithmetic_unary(NEG, a), -a)
        np.testing.assert_array_equal(be.arithmetic_unary(SIGN, a), np.sign(a).astype(dtype))
    if dt.kind == "i":  # abs(min) wraps to min in the unchecked kernel (base_arithmetic.cc:147-149)
        mn = np.array([np.iinfo(dtype).min], dtype=dtype)
        np.testing.assert_array_equal(be.arithmetic_unary(ABS, mn), mn)
    if dt.kind == "f":
        s = be.arithmetic_unary(SIGN, np.array([np.nan, -0.0, 0.0, -np.inf, np.inf], dtype=dtype))
        assert np.isnan(s[0]) and s[1] == 0 and s[2] == 0 and s[3] == -1 and s[4] == 1
        z = be.arithmetic_unary(ABS, np.array([-0.0, -np.inf], dtype=dtype))
        assert not np.signbit(z[0]) and z[1] == np.inf


# ---- compare ------------------------------------------------------------------------------
# arrow/compute/scalar_compare_test.go:299-483 (NumericCompareSuite, values only)
CMP_TABLE = [
    (EQ, [0, 0, 1, 1, 2, 2], [False, False, True, True, False, False]),
    (EQ, [0, 1, 2, 3, 4, 5], [False, True, False, False, False, False]),
    (EQ, [5, 4, 3, 2, 1, 0], [False, False, False, False, True, False]),
    (NE, [0, 0, 1, 1, 2, 2], [True, True, False, False, True, True]),
    (NE, [5, 4, 3, 2, 1, 0], [True, True, True, True, False, True]),
    (GT, [0, 0, 1, 1, 2, 2], [False, False, False, False, True, True]),
    (GT, [0, 1, 2, 3, 4, 5], [False, False, True, True, True, True]),
    (GT, [4, 5, 6, 7, 8, 9], [True, True, True, True, True, True]),
    (GE, [0, 0, 1, 1, 2, 2], [False, False, True, True, True, True]),
    (GE, [0, 1, 2, 3, 4, 5], [False, True, True, True, True, True]),
]


@pytest.mark.parametrize("dtype", OL.ALL_DTYPES, ids=str)
def test_compare_array_scalar(be, dtype):
    one = np.array([1], dtype=dtype)
    for op, vals, exp in CMP_TABLE:
        a = np.array(vals, dtype=dtype)
        out = be.comparison(op, AS, a, one, np.zeros(1, np.uint8))
        assert OL.unpack_bits(out, 0, 6).tolist() == exp, (op, vals)
    # LESS / LESS_EQUAL are the operand swap (compute/scalar_compare.go:73-99):
    # [0,0,1,1,2,2] < 1  ==  1 > [..]  → scalar_arr GT
    a = np.array([0, 0, 1, 1, 2, 2], dtype=dtype)
    out = be.comparison(GT, SA, one, a, np.zeros(1, np.uint8))
    assert OL.unpack_bits(out, 0, 6).tolist() == [True, True, False, False, False, False]
    out = be.comparison(GE, SA, one, a, np.zeros(1, np.uint8))
    assert OL.unpack_bits(out, 0, 6).tolist() == [True, True, True, True, False, False]
    # array ∘ array (scalar_compare_test.go TestSimpleCompareArrayArray shape)
    b = np.array([1, 0, 1, 2, 2, 3], dtype=dtype)
    out = be.comparison(GT, AA, a, b, np.zeros(1, np.uint8))
    assert OL.unpack_bits(out, 0, 6).tolist() == [False, False, False, False, False, False]
    out = be.comparison(EQ, AA, a, b, np.zeros(1, np.uint8))
    assert OL.unpack_bits(out, 0, 6).tolist() == [False, True, True, False, True, False]


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=str)
def test_compare_nan(be, dtype):
    # IEEE: every ordered compare with NaN is false, != is true (scalar_comparison.cc:31-56)
    a = np.array([np.nan, 1.0, np.nan], dtype=dtype)
    b = np.array([np.nan, np.nan, 1.0], dtype=dtype)
    for op, exp in [(EQ, False), (NE, True), (GT, False), (GE, False)]:
        out = be.comparison(op, AA, a, b, np.zeros(1, np.uint8))
        assert OL.unpack_bits(out, 0, 3).tolist() == [exp] * 3


def test_compare_preserves_bits_outside_range(be):
    # scalar_comparison.cc:71-81,91-95: prefix/tail via set_bit_to keep neighbouring bits
    a = np.arange(21, dtype=np.int32)
    init = np.full(5, 0xFF, dtype=np.uint8)
    out = be.comparison(GT, AS, a, np.array([100], np.int32), init, out_bit_offset=3)
    bits = OL.unpack_bits(out, 0, 40)
    assert bits[:3].all() and not bits[3:24].any() and bits[24:].all()


# ---- bitmaps ------------------------------------------------------------------------------
def bbits(*vals):
    """arrow/internal/testing/tools/bits.go:26-40 IntsToBitsLSB: leftmost hex digit = bit 0"""
    out = []
    for v in vals:
        digits = f"{v:08x}"
        out.append(sum((1 << j) for j, ch in enumerate(digits) if ch == "1"))
    return np.array(out, dtype=np.uint8)


# arrow/bitutil/bitutil_test.go:155-189 TestCountSetBits
COUNT_TABLE = [
    (bbits(0x11000000), 0, 3, 2),
    (bbits(0x11000011, 0x01000000), 0, 11, 5),
    (bbits(0x11001010, 0x11110000, 0x00001111, 0x11000011, 0x11001010, 0x11110000, 0x00001111, 0x11000011, 0x10001001), 0, 72, 35),
    (bbits(0x11111110), 0, 8, 7),
    (bbits(0x11100001), 0, 3, 3),
    (bbits(0x11111111, 0x11111111), 0, 11, 11),
    (bbits(*([0x11111111] * 9)), 0, 72, 72),
    (bbits(0x00000001), 0, 3, 0),
    (bbits(0x00000000, 0x00000000), 0, 11, 0),
    (bbits(*([0] * 9)), 0, 72, 0),
    (bbits(0x11000000), 1, 3, 1),
    (bbits(0x11000000), 2, 3, 0),
    (bbits(0x11000011, 0x01000000, 0x00000000), 1, 11, 4),
    (bbits(0x11000011, 0x01000000, 0x00000000), 2, 11, 3),
    (bbits(0x11000011, 0x01000000, 0x00000000), 3, 11, 3),
    (bbits(0x11000011, 0x01000000, 0x00000000), 6, 11, 3),
    (bbits(0x11000011, 0x01000000, 0x00000000), 7, 11, 2),
    (bbits(0x11000011, 0x01000000, 0x00000000), 8, 11, 1),
]


def test_count_set_bits_table(be):
    for buf, off, n, exp in COUNT_TABLE:
        assert be.count_set_bits(buf, off, n) == exp, (buf, off, n)


def test_count_set_bits_offset_sweep(be):
    # bitutil_test.go:191-224: 1000 random bytes, offsets {0..12,16,32,37,63,64,128,n-30,n-64}
    rng = np.random.default_rng(0)
    buf = rng.integers(0, 256, 1000, dtype=np.uint8)
    nbits = 8000
    bits = np.unpackbits(buf, bitorder="little")
    for off in [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 32, 37, 63, 64, 128, nbits - 30, nbits - 64]:
        assert be.count_set_bits(buf, off, nbits - off) == int(bits[off:].sum())


def bitmap_from_slice(vals, offset):
    """arrow/bitutil/bitmaps_test.go bitmapFromSlice: `offset` zero bits, then vals"""
    return OL.pack_bits([False] * offset + [bool(v) for v in vals])


LBITS = [0, 1, 1, 1, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1]
RBITS = [0, 0, 1, 0, 1, 1, 0, 0, 1, 1, 1, 0, 1, 0]
BITMAP_EXPECT = {  # bitmaps_test.go:484-536
    AND: [0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0],
    OR: [0, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1],
    XNOR: [1, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0],
    XOR: [0, 1, 0, 1, 1, 1, 0, 1, 1, 0, 1, 1, 1, 1],
    ANDNOT: [0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1],
}


@pytest.mark.parametrize("op", [AND, OR, XNOR, XOR, ANDNOT])
def test_bitmap_op_offsets(be, op):
    # bitmaps_test.go:364-470: aligned offsets {0,8,16,24,32,...} and unaligned
    # {0,1,3,5,7,8,13,21,38,75,120,65536} for left × right × out
    offsets = [0, 1, 3, 5, 7, 8, 13, 21, 38, 75, 120, 65536]
    combos = list(itertools.product(offsets, repeat=3))
    if be.name == "hip":  # every combo is an upload + launch; keep the mixed-alignment ones
        combos = [c for i, c in enumerate(combos) if i % 23 == 0] + [(8, 8, 8), (16, 8, 24), (65536, 1, 7)]
    n = len(LBITS)
    for lo, ro, oo in combos:
        left, right = bitmap_from_slice(LBITS, lo), bitmap_from_slice(RBITS, ro)
        for fill in (0x00, 0xFF):
            init = np.full((oo + n + 7) // 8 + 1, fill, dtype=np.uint8)
            out = be.bitmap_op(op, left, lo, right, ro, init, oo, n)
            got = OL.unpack_bits(out, oo, n).astype(int).tolist()
            assert got == BITMAP_EXPECT[op], (op, lo, ro, oo)
            # nothing outside [oo, oo+n) may change
            allbits = OL.unpack_bits(out, 0, out.size * 8)
            keep = np.ones(out.size * 8, bool); keep[oo:oo + n] = False
            assert (allbits[keep] == bool(fill)).all(), (op, lo, ro, oo, fill)


def test_small_bitmap_op(be):
    # bitmaps_test.go:542-556 TestSmallBitmapOp
    left, right = np.array([127, 207], np.uint8), np.array([254, 127], np.uint8)
    out = be.bitmap_op(AND, left, 0, right, 0, np.zeros(2, np.uint8), 0, 8)
    assert out[0] == 126
    out = be.bitmap_op(AND, left, 0, right, 0, np.zeros(2, np.uint8), 0, 16)
    assert out.tolist() == [126, 79]


def test_bitmap_op_byte_aligned_nonzero_offset(be):
    """The case arrow/bitutil/bitmaps.go:536 (`endMask := (lOffset + length%8)`) gets
    wrong upstream (last byte left unwritten when lOffset is a non-zero multiple of 8 and
    the range ends on a byte boundary).  We implement the documented contract."""
    left, right = np.array([0, 0xF0, 0xAA], np.uint8), np.array([0, 0x3C, 0xFF], np.uint8)
    out = be.bitmap_op(AND, left, 8, right, 8, np.zeros(3, np.uint8), 8, 16)
    assert out.tolist() == [0, 0x30, 0xAA]


def test_copy_invert_setbits(be):
    # bitmaps.go:483-493 CopyBitmap / InvertBitmap, bitutil.go:158-204 SetBitsTo
    # (bitutil_test.go:226-257 TestSetBitsTo expectations)
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, 40, dtype=np.uint8)
    sbits = np.unpackbits(src, bitorder="little")
    for soff, doff, n in [(0, 0, 64), (3, 0, 100), (0, 5, 77), (13, 21, 200), (8, 8, 16), (1, 7, 1), (5, 9, 0)]:
        for inv in (False, True):
            out = be.copy_bitmap(src, soff, n, np.full(40, 0xA5, np.uint8), doff, inv)
            exp = np.unpackbits(np.full(40, 0xA5, np.uint8), bitorder="little")
            exp[doff:doff + n] = sbits[soff:soff + n] ^ (1 if inv else 0)
            np.testing.assert_array_equal(np.unpackbits(out, bitorder="little"), exp)
    # TestSetBitsTo (bitutil_test.go:226-257)
    for fill in (0x00, 0xFF):
        bm = be.set_bits_to(np.full(8, fill, np.uint8), 0, 0, True)
        assert bm.tolist() == [fill] * 8
        bm = be.set_bits_to(np.full(8, fill, np.uint8), 2, 2, True)
        bm = be.set_bits_to(bm, 4, 2, False)
        assert bm[0] == ((fill & ~0x3C) | 0x0C)
        bm = be.set_bits_to(np.full(8, fill, np.uint8), 4, 8, True)
        bm = be.set_bits_to(bm, 12, 4, False)
        assert bm[0] == ((fill & 0x0F) | 0xF0) and bm[1] == 0x0F
        bm = be.set_bits_to(np.full(8, fill, np.uint8), 0, 64, True)
        assert bm.tolist() == [0xFF] * 8
        bm = be.set_bits_to(np.full(8, fill, np.uint8), 0, 64, False)
        assert bm.tolist() == [0x00] * 8


def test_kleene_truth_tables(be):
    # compute/scalar_bool_test.go (TestAndKleene/OrKleene/AndNotKleene) truth tables
    T, F, N = True, False, None
    vals = [T, F, N]
    pairs = [(a, b) for a in vals for b in vals]
    lv = OL.pack_bits([a is not None for a, _ in pairs]); ld = OL.pack_bits([a is True for a, _ in pairs])
    rv = OL.pack_bits([b is not None for _, b in pairs]); rd = OL.pack_bits([b is True for _, b in pairs])

    def k_and(a, b):
        if a is False or b is False: return False
        if a is None or b is None: return None
        return True

    def k_or(a, b):
        if a is True or b is True: return True
        if a is None or b is None: return None
        return False

    def k_not(b): return None if b is None else (not b)
    for op, fn in [(0, k_and), (1, k_or), (2, lambda a, b: k_and(a, k_not(b)))]:
        for off in (0, 3):
            lvo, ldo = bitmap_from_slice(OL.unpack_bits(lv, 0, 9), off), bitmap_from_slice(OL.unpack_bits(ld, 0, 9), off)
            ov, od = be.kleene(op, lvo, ldo, off, rv, rd, 0, np.zeros(3, np.uint8), np.zeros(3, np.uint8), 5, 9)
            got = [(bool(d) if v else None) for v, d in zip(OL.unpack_bits(ov, 5, 9), OL.unpack_bits(od, 5, 9))]
            assert got == [fn(a, b) for a, b in pairs], op


# ---- filter -------------------------------------------------------------------------------
# arrow/compute/vector_selection_test.go:449-484 (FilterKernelNumeric.TestFilterNumeric)
FILTER_CASES = [
    # values, filter, null_selection, expected
    ([], [], DROP, []),
    ([9], [False], DROP, []),
    ([9], [True], DROP, [9]),
    ([9], [None], DROP, []),
    ([9], [None], EMIT, [None]),
    ([None], [True], DROP, [None]),
    ([7, 8, 9], [False, True, False], DROP, [8]),
    ([7, 8, 9], [True, False, True], DROP, [7, 9]),
    ([None, 8, 9], [False, True, False], DROP, [8]),
    ([7, 8, 9], [None, True, False], DROP, [8]),
    ([7, 8, 9], [None, True, False], EMIT, [None, 8]),
    ([7, 8, 9], [True, None, True], DROP, [7, 9]),
    ([7, 8, 9], [True, None, True], EMIT, [7, None, 9]),
]


def run_filter_case(be, dtype, values, filt, null_sel, sliced):
    vals, vvalid = mk(values, dtype, null_fill=42)
    fd, fvalid = mk(filt, np.uint8)
    fdata = OL.pack_bits(fd.astype(bool)) if len(filt) else np.zeros(1, np.uint8)
    voff = foff = 0
    if sliced:
        # vector_selection_test.go:121-144: 3 null fillers before the values, [true,false]
        # before the filter, then slice → non-zero bit offsets everywhere
        voff, foff = 3, 2
        vbits = [False] * 3 + [v is not None for v in values]
        vvalid = OL.pack_bits(vbits)
        fbits = [True, False] + [bool(x) for x in fd.tolist()]
        fdata = OL.pack_bits(fbits)
        if fvalid is not None or True:
            fvalid = OL.pack_bits([True, True] + [f is not None for f in filt])
    n = len(values)
    has_nulls = (vvalid is not None and not OL.unpack_bits(vvalid, voff, n).all()) or \
                (fvalid is not None and not OL.unpack_bits(fvalid, foff, n).all())
    out, ov, nulls = be.filter(vals, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid=has_nulls)
    return logical(out, ov, 0, len(out)), out, nulls


@pytest.mark.parametrize("dtype", [np.int8, np.uint16, np.int32, np.float32, np.int64, np.float64], ids=str)
@pytest.mark.parametrize("sliced", [False, True])
def test_filter_vectors(be, dtype, sliced):
    for values, filt, null_sel, exp in FILTER_CASES:
        got, raw, nulls = run_filter_case(be, dtype, values, filt, null_sel, sliced)
        assert got == exp, (values, filt, null_sel)
        assert nulls == sum(1 for e in exp if e is None)


def test_filter_sliced_filter(be):
    # vector_selection_test.go:474-478: filter [F,T,T,T,F,T][3:6] on [7,8,9] → [7,9]
    vals = np.array([7, 8, 9], np.int64)
    fdata = OL.pack_bits([False, True, True, True, False, True])
    out, ov, _ = be.filter(vals, None, 0, fdata, None, 3, 3, DROP, want_valid=False)
    assert out.tolist() == [7, 9]


def test_filter_null_payload_rules(be):
    """a7 payload contract: selected-but-null value keeps its payload; filter-null under
    EMIT gets payload 0 (vector_selection.go:293-297,417-421)."""
    vals = np.array([11, 22, 33, 44], np.int64)
    vvalid = OL.pack_bits([True, False, True, True])
    fdata = OL.pack_bits([True, True, False, True])
    fvalid = OL.pack_bits([True, True, True, False])
    out, ov, nulls = be.filter(vals, vvalid, 0, fdata, fvalid, 0, 4, EMIT, want_valid=True)
    assert out.tolist() == [11, 22, 0] and OL.unpack_bits(ov, 0, 3).tolist() == [True, False, False] and nulls == 2
    out, ov, nulls = be.filter(vals, vvalid, 0, fdata, fvalid, 0, 4, DROP, want_valid=True)
    assert out.tolist() == [11, 22] and OL.unpack_bits(ov, 0, 2).tolist() == [True, False] and nulls == 1


def test_filter_to_indices(be):
    # kernels/vector_selection.go:102-236 GetTakeIndices
    fdata = OL.pack_bits([True, False, True, True, False])
    fvalid = OL.pack_bits([True, True, False, True, False])
    idx, iv, nulls = be.filter_to_indices(fdata, None, 0, 5, DROP, want_valid=False)
    assert idx.tolist() == [0, 2, 3]
    idx, iv, nulls = be.filter_to_indices(fdata, fvalid, 0, 5, DROP, want_valid=False)
    assert idx.tolist() == [0, 3]
    idx, iv, nulls = be.filter_to_indices(fdata, fvalid, 0, 5, EMIT, want_valid=True)
    assert idx.tolist() == [0, 0, 3, 0] and OL.unpack_bits(iv, 0, 4).tolist() == [True, False, True, False] and nulls == 2


# ---- take ---------------------------------------------------------------------------------
# arrow/compute/vector_selection_test.go:1127-1141 (TakeKernelTestNumeric) + :213-253
TAKE_CASES = [
    ([7, 8, 9], [], []),
    ([7, 8, 9], [0, 1, 0], [7, 8, 7]),
    ([None, 8, 9], [0, 1, 0], [None, 8, None]),
    ([7, 8, 9], [None, 1, 0], [None, 8, 7]),
    ([None, 8, 9], [], []),
    ([7, 8, 9], [0, 0, 0, 0, 0, 0, 2], [7, 7, 7, 7, 7, 7, 9]),
]


@pytest.mark.parametrize("vdtype", [np.int8, np.uint16, np.float32, np.int64, np.float64], ids=str)
@pytest.mark.parametrize("idtype", [np.int8, np.uint32, np.int32, np.int64, np.uint16], ids=str)
def test_take_vectors(be, vdtype, idtype):
    for values, indices, exp in TAKE_CASES:
        vals, vvalid = mk(values, vdtype, null_fill=55)
        idx, ivalid = mk(indices, idtype, null_fill=1)
        want_valid = vvalid is not None or ivalid is not None
        st, out, ov, nulls, _ = be.take(vals, vvalid, 0, idx, ivalid, 0, True, want_valid)
        assert st == STATUS_OK
        assert logical(out, ov, 0, len(indices)) == exp
        if want_valid:
            assert nulls == sum(1 for e in exp if e is None)
            # null outputs keep payload 0 (fresh zeroed buffer, a8)
            assert all(o == 0 for o, e in zip(out.tolist(), exp) if e is None)


def test_take_bounds_errors(be):
    # vector_selection_test.go:1139-1140: index 9 and -1 → ErrIndex (helpers.go:929-957)
    vals = np.array([7, 8, 9], np.int64)
    st, *_rest, bad = be.take(vals, None, 0, np.array([0, 9, 0], np.int8), None, 0, True, False)
    assert st == STATUS_EINDEX and bad == 9
    st, *_rest, bad = be.take(vals, None, 0, np.array([0, -1, 0], np.int8), None, 0, True, False)
    assert st == STATUS_EINDEX and bad == -1
    # first offender in index order
    st, *_rest, bad = be.take(vals, None, 0, np.array([0, 5, 1, -3], np.int32), None, 0, True, False)
    assert st == STATUS_EINDEX and bad == 5
    # a NULL index slot is never bounds-checked (VisitSetBitRuns over validity)
    idx, ivalid = np.array([0, 99, 2], np.int32), OL.pack_bits([True, False, True])
    st, out, ov, nulls, _ = be.take(vals, None, 0, idx, ivalid, 0, True, True)
    assert st == STATUS_OK and out.tolist() == [7, 0, 9] and nulls == 1


def test_take_sliced(be):
    # vector_selection_test.go:213-253: sliced values (offset 2) and sliced indices (offset 1)
    vfull, vvalid_full = mk([None, None, 7, 8, None], np.int64, null_fill=99)
    ifull = np.array([77, 0, 1, 2], np.int32)
    ivalid_full = OL.pack_bits([False, True, True, True])
    st, out, ov, nulls, _ = be.take(vfull[2:], vvalid_full, 2, ifull[1:], ivalid_full, 1, True, True)
    assert st == STATUS_OK
    assert logical(out, ov, 0, 3) == [7, 8, None] and nulls == 1


# ---- hashing ------------------------------------------------------------------------------
def test_hash_int_function(orc):
    # internal/hashing/hash_funcs.go:60-67: bswap64(0x9E3779B185EBCA87 * v)
    for v in [0, 1, 2, 42, 2**63, 2**64 - 1, 0xDEADBEEF]:
        prod = (11400714785074694791 * v) % 2**64
        assert orc.hash_int(v) == int.from_bytes(prod.to_bytes(8, "little"), "big")


# arrow/compute/vector_hash_test.go:236-254 (PrimitiveHashKernelSuite.TestUnique)
UNIQUE_CASES = [
    ([2, None, 2, 1], [2, None, 1]),
    ([None, None, 3, 1], [None, 3, 1]),
    ([2, 1, 2, 1], [2, 1]),
    ([5, 4, 3, 1, 1], [5, 4, 3, 1]),
]


@pytest.mark.parametrize("dtype", [np.int64, np.uint64, np.float64], ids=str)
def test_unique_vectors(be, dtype):
    for values, exp in UNIQUE_CASES:
        keys, valid = mk(values, dtype, null_fill=123)
        ids, idv, d, null_id = be.hash_encode(keys, valid, 0, True)
        got = [None if i == null_id else x for i, x in enumerate(d.view(dtype).tolist())]
        assert got == exp, values
    # sliced input: [1,2,null,3,2,null][1:5] → [2,null,3]  (vector_hash_test.go:250-253)
    keys, valid = mk([1, 2, None, 3, 2, None], dtype, null_fill=9)
    ids, idv, d, null_id = be.hash_encode(keys[1:5], valid, 1, True)
    got = [None if i == null_id else x for i, x in enumerate(d.view(dtype).tolist())]
    assert got == [2, None, 3]


def test_dictionary_encode_vectors(be):
    # vector_hash_test.go:541-591: ["foo","bar","foo",null,"bar",null] → [0,1,0,null,1,null]
    # (NullEncodingMask) or [0,1,0,2,1,2] + 3-entry dictionary (NullEncodingEncode) — same
    # rule for numeric keys (doAppendNumeric, kernels/vector_hash.go:359-385)
    keys, valid = mk([10, 20, 10, None, 20, None], np.int64, null_fill=77)
    ids, idv, d, null_id = be.hash_encode(keys, valid, 0, False)
    assert ids.tolist() == [0, 1, 0, 0, 1, 0]
    assert OL.unpack_bits(idv, 0, 6).tolist() == [True, True, True, False, True, False]
    assert d.view(np.int64).tolist() == [10, 20] and null_id == -1
    ids, idv, d, null_id = be.hash_encode(keys, valid, 0, True)
    assert ids.tolist() == [0, 1, 0, 2, 1, 2] and null_id == 2 and len(d) == 3
    assert OL.unpack_bits(idv, 0, 6).all()
    assert d.view(np.int64).tolist()[:2] == [10, 20] and d[2] == 0


def test_dictionary_encode_resizes_memo_table(be):
    # vector_hash_test.go:768 TestDictionaryEncodeResizesMemoTable: enough distinct keys
    # to force table growth (initial capacity 32, grows ×4 at load 1/2)
    keys = np.arange(1000, dtype=np.int64) * 7919
    keys = np.concatenate([keys, keys[::-1]])
    ids, idv, d, null_id = be.hash_encode(keys, None, 0, False)
    assert d.view(np.int64).tolist() == (np.arange(1000) * 7919).tolist()
    assert ids.tolist() == list(range(1000)) + list(range(999, -1, -1))


def test_hash_float_bit_patterns(be):
    # vector_hash.go:604-607,690-693: Float64 keys hash raw bits → +0.0 ≠ -0.0, NaN payloads distinct
    nan1 = np.array([0x7FF8000000000001], np.uint64).view(np.float64)[0]
    nan2 = np.array([0x7FF8000000000002], np.uint64).view(np.float64)[0]
    keys = np.array([0.0, -0.0, nan1, nan2, nan1, 0.0], np.float64)
    ids, idv, d, null_id = be.hash_encode(keys, None, 0, False)
    assert ids.tolist() == [0, 1, 2, 3, 2, 0] and len(d) == 4


def test_hash_all_ones_key(be):
    # 0xFFFF...FFFF is a legal key (the GPU table's EMPTY marker must not swallow it)
    keys = np.array([2**64 - 1, 5, 2**64 - 1, 0], np.uint64)
    ids, idv, d, null_id = be.hash_encode(keys, None, 0, False)
    assert ids.tolist() == [0, 1, 0, 2] and d.tolist() == [2**64 - 1, 5, 0]


# ---- fused Compare→Filter→Sum ---------------------------------------------------------------
def test_fused_equals_unfused_chain_small(be):
    # the reference chain: "greater" → Filter(DropNulls) → math.Sum
    x = np.array([5, -3, 10, 7, 2, 9], np.int64)
    valid = OL.pack_bits([True, True, False, True, True, True])
    assert be.cmp_filter_sum_i64(GT, x, None, 0, 4) == (5 + 10 + 7 + 9, 4)
    assert be.cmp_filter_sum_i64(GT, x, valid, 0, 4) == (5 + 7 + 9, 3)
    assert be.cmp_filter_sum_i64(GE, x, valid, 0, 7) == (7 + 9, 2)
    assert be.cmp_filter_sum_i64(EQ, x, valid, 0, 2) == (2, 1)
    assert be.cmp_filter_sum_i64(NE, x, valid, 0, 2) == (5 - 3 + 7 + 9, 4)
    xf = x.astype(np.float64)
    s, c = be.cmp_filter_sum_f64(GT, xf, valid, 0, 4.0)
    assert (s, c) == (21.0, 3)


# ---- cumulative_sum / cumulative_sum_checked ----------------------------------------------
# arrow/compute/vector_cumulative_test.go
def run_cumsum(be, dtype, vals, start=None, skip_nulls=False, checked=False, sl=None):
    """vals: python list with None; sl = (lo, hi) slices the input like array.NewSlice."""
    arr, valid = mk(vals, dtype, null_fill=77)  # junk under nulls: the payload must not leak into the sums
    lo, hi = sl if sl else (0, len(vals))
    st, out, ov, nulls = be.cumulative_sum(arr[lo:hi], valid, lo, None if start is None else np.dtype(dtype).type(start),
                                           skip_nulls, checked)
    if st != STATUS_OK:
        return st, None
    n = hi - lo
    got = logical(out, ov, 0, n)
    assert nulls == sum(g is None for g in got)
    if ov is not None:  # null rows keep the zero of the fresh output buffer (prepareCumulativeOutput :211-226)
        assert all(out[i] == 0 for i in range(n) if got[i] is None)
    return st, got


def test_cumulative_sum_basic(be):
    # TestCumulativeSum :41-55, TestCumulativeSumValueOptions :57-75
    assert run_cumsum(be, np.int32, [1, 2, 3, 4]) == (STATUS_OK, [1, 3, 6, 10])
    assert run_cumsum(be, np.int32, [1, 2, 3]) == (STATUS_OK, [1, 3, 6])


def test_cumulative_sum_additional_inputs(be):
    # TestCumulativeSumAdditionalInputs :94-170
    assert run_cumsum(be, np.int32, []) == (STATUS_OK, [])
    assert run_cumsum(be, np.int32, [None, None]) == (STATUS_OK, [None, None])
    assert run_cumsum(be, np.uint8, [1, 2, 3]) == (STATUS_OK, [1, 3, 6])
    assert run_cumsum(be, np.float32, [1.5, 2.5]) == (STATUS_OK, [1.5, 4.0])
    assert run_cumsum(be, np.float64, [1.5, 2.5]) == (STATUS_OK, [1.5, 4.0])
    assert run_cumsum(be, np.int32, [3]) == (STATUS_OK, [3])                      # scalar input → 1-row array
    assert run_cumsum(be, np.int32, [0, 1, 2, 3], sl=(1, 3)) == (STATUS_OK, [1, 3])  # sliced
    # sliced nulls
    assert run_cumsum(be, np.int32, [9, None, 2, 3, 99], sl=(1, 4)) == (STATUS_OK, [None, None, None])
    assert run_cumsum(be, np.int32, [9, None, 2, 3, 99], sl=(1, 4), skip_nulls=True) == (STATUS_OK, [None, 2, 5])


@pytest.mark.parametrize("dtype", OL.ALL_DTYPES, ids=str)
@pytest.mark.parametrize("checked", [False, True])
def test_cumulative_sum_null_scalar_input(be, dtype, checked):
    # TestCumulativeSumNullScalarInput :172-236 — a null scalar is a 1-row all-null array
    for start in (None, 10):
        for skip in (False, True):
            assert run_cumsum(be, dtype, [None], start=start, skip_nulls=skip, checked=checked) == (STATUS_OK, [None])


def test_cumulative_sum_nulls_and_start(be):
    # TestCumulativeSumNullsAndStart :238-275
    v = [1, None, 2, None, 3]
    assert run_cumsum(be, np.int32, v) == (STATUS_OK, [1, None, None, None, None])
    assert run_cumsum(be, np.int32, v, skip_nulls=True) == (STATUS_OK, [1, None, 3, None, 6])
    assert run_cumsum(be, np.int32, v, start=10, skip_nulls=True) == (STATUS_OK, [11, None, 13, None, 16])


def test_cumulative_sum_state_across_chunks(be):
    # TestCumulativeSumStateAcrossChunks :646-679 — chunks [1,2] [null] [3]: one running state; here one array
    assert run_cumsum(be, np.int32, [1, 2, None, 3]) == (STATUS_OK, [1, 3, None, None])
    assert run_cumsum(be, np.int32, [1, 2, None, 3], skip_nulls=True) == (STATUS_OK, [1, 3, None, 6])


def test_cumulative_sum_checked(be):
    # TestCumulativeSumChecked :750-767 — unchecked wraps, checked reports
    assert run_cumsum(be, np.int8, [127, 1]) == (STATUS_OK, [127, -128])
    assert run_cumsum(be, np.int8, [127, 1], checked=True)[0] == STATUS_EOVERFLOW


_LIM = {np.dtype(d): (np.iinfo(d).min, np.iinfo(d).max) for d in OL.INT_DTYPES}


@pytest.mark.parametrize("dtype", OL.INT_DTYPES, ids=str)
def test_cumulative_sum_checked_integer_overflow(be, dtype):
    # TestCumulativeSumCheckedIntegerOverflow :769-806
    lo, hi = _LIM[np.dtype(dtype)]
    assert run_cumsum(be, dtype, [hi, 1], checked=True)[0] == STATUS_EOVERFLOW
    if lo < 0:
        assert run_cumsum(be, dtype, [lo, -1], checked=True)[0] == STATUS_EOVERFLOW
    # the start value takes part in the running sum (cumulativeSumState.current)
    assert run_cumsum(be, dtype, [1], start=hi, checked=True)[0] == STATUS_EOVERFLOW
    # rows behind a null (propagated) or null rows themselves never overflow
    assert run_cumsum(be, dtype, [hi, None, 1], checked=True) == (STATUS_OK, [hi, None, None])
    assert run_cumsum(be, dtype, [hi, None, 1], checked=True, skip_nulls=True)[0] == STATUS_EOVERFLOW


@pytest.mark.parametrize("dtype", OL.INT_DTYPES, ids=str)
def test_cumulative_sum_checked_integer_boundaries(be, dtype):
    # TestCumulativeSumCheckedIntegerBoundaries :808-840
    lo, hi = _LIM[np.dtype(dtype)]
    assert run_cumsum(be, dtype, [hi - 1, 1], checked=True) == (STATUS_OK, [hi - 1, hi])
    if lo < 0:
        assert run_cumsum(be, dtype, [lo + 1, -1], checked=True) == (STATUS_OK, [lo + 1, lo])
        # leaves the range and comes back: still an overflow (the reference stops at the first one)
        assert run_cumsum(be, dtype, [hi, 1, -5], checked=True)[0] == STATUS_EOVERFLOW
        assert run_cumsum(be, dtype, [hi, -5, 1], checked=True) == (STATUS_OK, [hi, hi - 5, hi - 4])


# ---- numeric cast --------------------------------------------------------------------------
# arrow/compute/cast_test.go
def run_cast(be, in_dtype, out_dtype, vals, safe=True, allow_int_overflow=False, allow_float_truncate=False, sl=None):
    arr, valid = mk(vals, in_dtype, null_fill=0)
    lo, hi = sl if sl else (0, len(vals))
    st, out, msg = be.cast_numeric(arr[lo:hi], out_dtype, valid, lo, allow_int_overflow or not safe, allow_float_truncate or not safe)
    if st != STATUS_OK:
        return st, msg
    return st, logical(out, valid, lo, hi - lo)  # NullIntersection: the output validity IS the input validity


def test_cast_int_upcast(be):
    # TestToIntUpcast :483-489
    assert run_cast(be, np.int8, np.int32, [0, None, 127, -1, 0]) == (STATUS_OK, [0, None, 127, -1, 0])
    assert run_cast(be, np.uint8, np.int16, [0, 100, 200, 255, 0]) == (STATUS_OK, [0, 100, 200, 255, 0])


def test_cast_int_downcast_safe(be):
    # TestToIntDowncastSafe :491-518
    assert run_cast(be, np.int16, np.uint8, [0, None, 200, 1, 2]) == (STATUS_OK, [0, None, 200, 1, 2])
    assert run_cast(be, np.int16, np.uint8, [0, None, 256, 0, 0]) == (STATUS_EINVALID, "integer value 256 not in range: 0 to 255")
    assert run_cast(be, np.int16, np.uint8, [0, None, -1, 0, 0]) == (STATUS_EINVALID, "integer value -1 not in range: 0 to 255")
    assert run_cast(be, np.int32, np.int16, [0, None, 2000, 1, 2]) == (STATUS_OK, [0, None, 2000, 1, 2])
    assert run_cast(be, np.int32, np.int16, [0, None, 2000, 70000, 2]) == (STATUS_EINVALID, "integer value 70000 not in range: -32768 to 32767")
    assert run_cast(be, np.int32, np.int16, [0, None, 2000, -70000, 2])[0] == STATUS_EINVALID
    assert run_cast(be, np.int32, np.uint8, [0, None, 2000, -70000, 2]) == (STATUS_EINVALID, "integer value 2000 not in range: 0 to 255")


def test_cast_integer_signed_to_unsigned(be):
    # TestIntegerSignedToUnsigned :520-552
    i32 = [-2147483648, None, -1, 65535, 2147483647]
    for to in (np.uint32, np.uint64, np.uint16):
        assert run_cast(be, np.int32, to, i32)[0] == STATUS_EINVALID
    assert run_cast(be, np.int32, np.uint32, i32, allow_int_overflow=True) == (STATUS_OK, [2147483648, None, 4294967295, 65535, 2147483647])
    assert run_cast(be, np.int32, np.uint64, i32, allow_int_overflow=True) == \
        (STATUS_OK, [18446744071562067968, None, 18446744073709551615, 65535, 2147483647])
    i32 = [0, None, 0, 65536, 2147483647]
    assert run_cast(be, np.int32, np.uint16, i32) == (STATUS_EINVALID, "integer value 65536 not in range: 0 to 65535")
    assert run_cast(be, np.int32, np.uint16, i32, allow_int_overflow=True) == (STATUS_OK, [0, None, 0, 0, 65535])


def test_cast_integer_unsigned_to_signed(be):
    # TestIntegerUnsignedToSigned :554-571
    u32 = [4294967295, None, 0, 32768]
    assert run_cast(be, np.uint32, np.int32, u32) == (STATUS_EINVALID, "integer value 4294967295 not in range: 0 to 2147483647")
    assert run_cast(be, np.uint32, np.int16, u32)[0] == STATUS_EINVALID
    assert run_cast(be, np.uint32, np.int16, u32, sl=(1, 4)) == (STATUS_EINVALID, "integer value 32768 not in range: 0 to 32767")
    assert run_cast(be, np.uint32, np.int32, u32, allow_int_overflow=True) == (STATUS_OK, [-1, None, 0, 32768])
    assert run_cast(be, np.uint32, np.int64, u32, allow_int_overflow=True) == (STATUS_OK, [4294967295, None, 0, 32768])
    assert run_cast(be, np.uint32, np.int16, u32, allow_int_overflow=True) == (STATUS_OK, [-1, None, 0, -32768])


def test_cast_int_downcast_unsafe(be):
    # TestToIntDowncastUnsafe :573-586
    u = dict(allow_int_overflow=True)
    assert run_cast(be, np.int16, np.uint8, [0, None, 200, 1, 2], **u) == (STATUS_OK, [0, None, 200, 1, 2])
    assert run_cast(be, np.int16, np.uint8, [0, None, 256, 1, 2, -1], **u) == (STATUS_OK, [0, None, 0, 1, 2, 255])
    assert run_cast(be, np.int32, np.int16, [0, None, 2000, 1, 2, -1], **u) == (STATUS_OK, [0, None, 2000, 1, 2, -1])
    assert run_cast(be, np.int32, np.int16, [0, None, 2000, 70000, -70000], **u) == (STATUS_OK, [0, None, 2000, 4464, -4464])


@pytest.mark.parametrize("frm", [np.float32, np.float64], ids=str)
@pytest.mark.parametrize("to", [np.int32, np.int64], ids=str)
def test_cast_floating_to_int(be, frm, to):
    # TestFloatingToInt :588-603
    assert run_cast(be, frm, to, [1.0, None, 0.0, -1.0, 5.0]) == (STATUS_OK, [1, None, 0, -1, 5])
    assert run_cast(be, frm, to, [1.5, 0.0, None, 0.5, -1.5, 5.5]) == \
        (STATUS_EINVALID, "float value 1.500000 was truncated converting to " + np.dtype(to).name)
    assert run_cast(be, frm, to, [1.5, 0.0, None, 0.5, -1.5, 5.5], allow_float_truncate=True) == (STATUS_OK, [1, 0, None, 0, -1, 5])


def test_cast_int_to_floating(be):
    # TestIntToFloating :611-629
    for frm in (np.uint32, np.int32):
        assert run_cast(be, frm, np.float32, [16777216, 16777217])[0] == STATUS_EINVALID
        assert run_cast(be, frm, np.float32, [16777216]) == (STATUS_OK, [16777216.0])
    i64 = [-9223372036854775808, -9223372036854775807, 0, 9223372036854775806, 9223372036854775807]
    assert run_cast(be, np.int64, np.float64, i64) == \
        (STATUS_EINVALID, "integer value -9223372036854775808 not in range: -9007199254740992 to 9007199254740992")
    # masked: the offenders are null → the cast succeeds (maskArrayWithNullsAt {0,1,3,4})
    arr = np.array(i64, np.int64)
    valid = OL.pack_bits([False, False, True, False, False])
    st, out, _ = be.cast_numeric(arr, np.float64, valid, 0, False, False)
    assert st == STATUS_OK and out[2] == 0.0
    assert run_cast(be, np.uint64, np.float64, [9007199254740992, 9007199254740993]) == \
        (STATUS_EINVALID, "integer value 9007199254740993 not in range: 0 to 9007199254740992")
    # small integers never need the check (checkIntToFloatTrunc :700-703), int32 → float64 neither
    assert run_cast(be, np.int16, np.float32, [-32768, 32767]) == (STATUS_OK, [-32768.0, 32767.0])
    assert run_cast(be, np.int32, np.float64, [-2147483648, 2147483647]) == (STATUS_OK, [-2147483648.0, 2147483647.0])


def test_cast_float_to_float_and_bool(be):
    assert run_cast(be, np.float64, np.float32, [1.5, None, 0.1, -2.5e-50, 1e300]) == \
        (STATUS_OK, [1.5, None, float(np.float32(0.1)), -0.0, float("inf")])
    assert run_cast(be, np.float32, np.float64, [1.5, None, 3.25]) == (STATUS_OK, [1.5, None, 3.25])
    # boolToNum (numeric_cast.go:555-569)
    bits = OL.pack_bits([False, False, True, True, False, True, False, False, True, True, True])
    for dt in OL.ALL_DTYPES:
        assert be.cast_bool_to_numeric(bits, 2, 9, dt).tolist() == [1, 1, 0, 1, 0, 0, 1, 1, 1]


# ---- is_in ---------------------------------------------------------------------------------
# arrow/compute/scalar_set_lookup_test.go:104-168 TestIsInPrimitive
MATCH, SKIP, EMIT_NULL, INCONCLUSIVE = 0, 1, 2, 3


def run_is_in(be, dtype, vals, vset, nb, out_off=0, fill=0):
    arr, valid = mk(vals, dtype, null_fill=2)       # a null slot carries a payload that IS in the set: it must not match by value
    sarr, svalid = mk(vset, dtype, null_fill=3)     # and a null set entry carries a payload that is probed for
    od, ov = be.is_in(arr, valid, 0, sarr, svalid, 0, nb, out_off, fill)
    n = len(vals)
    d = OL.unpack_bits(od, out_off, n); v = OL.unpack_bits(ov, out_off, n)
    return [bool(x) if ok else None for x, ok in zip(d, v)], od, ov


@pytest.mark.parametrize("dtype", OL.ALL_DTYPES, ids=str)
def test_is_in_primitive(be, dtype):
    T, F, N = True, False, None
    cases = [
        ([0, 1, 2, 3, 2], [2, 1], {MATCH: [F, T, T, F, T]}),
        ([None, 1, 2, 3, 2], [2, 1], {MATCH: [F, T, T, F, T], SKIP: [F, T, T, F, T], EMIT_NULL: [N, T, T, F, T], INCONCLUSIVE: [N, T, T, F, T]}),
        ([0, 1, 2, 3, 2], [2, None, 1], {MATCH: [F, T, T, F, T], SKIP: [F, T, T, F, T], EMIT_NULL: [F, T, T, F, T], INCONCLUSIVE: [N, T, T, N, T]}),
        ([None, 1, 2, 3, 2], [2, None, 1], {MATCH: [T, T, T, F, T], SKIP: [F, T, T, F, T], EMIT_NULL: [N, T, T, F, T], INCONCLUSIVE: [N, T, T, N, T]}),
        ([None, 1, 2, 3, 2], [None, 2, 2, None, 1, 1], {MATCH: [T, T, T, F, T], SKIP: [F, T, T, F, T], EMIT_NULL: [N, T, T, F, T],
                                                        INCONCLUSIVE: [N, T, T, N, T]}),
        ([], [], {MATCH: []}),
    ]
    for vals, vset, exp in cases:
        for nb, want in exp.items():
            assert run_is_in(be, dtype, vals, vset, nb)[0] == want, (vals, vset, nb)
    # an empty set matches nothing; values are keyed on their bits
    assert run_is_in(be, dtype, [1, None], [], MATCH)[0] == [F, F]
    assert run_is_in(be, dtype, [1, None], [], EMIT_NULL)[0] == [F, N]


def test_is_in_bit_patterns_and_output_range(be):
    # floats are looked up by bit pattern (SetLookupState[uint64], scalar_set_lookup.go:106-133)
    nan1 = np.array([0x7FF8000000000001], np.uint64).view(np.float64)[0]
    vals = np.array([0.0, -0.0, np.nan, nan1, 1.5], np.float64)
    got = be.is_in(vals, None, 0, np.array([0.0, np.nan], np.float64), None, 0, MATCH)[0]
    assert OL.unpack_bits(got, 0, 5).tolist() == [1, 0, 1, 0, 0]
    # the all-ones key and key 0 are ordinary members
    k = np.array([2**64 - 1, 0, 5], np.uint64)
    assert OL.unpack_bits(be.is_in(k, None, 0, np.array([2**64 - 1], np.uint64), None, 0, MATCH)[0], 0, 3).tolist() == [1, 0, 0]
    assert OL.unpack_bits(be.is_in(k, None, 0, np.array([0], np.uint64), None, 0, MATCH)[0], 0, 3).tolist() == [0, 1, 0]
    # bits outside [out_off, out_off + n) are preserved, whatever they were
    for fill in (0x00, 0xFF):
        for out_off in (0, 3, 13, 64, 67):
            exp, od, ov = run_is_in(be, np.int32, [None, 1, 2, 3, 2] * 30, [2, None, 1], INCONCLUSIVE, out_off, fill)
            assert exp == [None, True, True, None, True] * 30
            n = 150
            for buf in (od, ov):
                bits = OL.unpack_bits(buf, 0, len(buf) * 8)
                assert all(b == (fill & 1) for b in bits[:out_off]) and all(b == (fill & 1) for b in bits[out_off + n:])


# ---- sort_indices ----------------------------------------------------------------------------
# arrow/compute/vector_sort_test.go
def run_sort(be, dtype, vals, descending=False, nulls_at_start=False, sl=None):
    arr, valid = mk(vals, dtype, null_fill=0)
    lo, hi = sl if sl else (0, len(vals))
    return be.sort_indices(arr[lo:hi], valid, lo, descending, nulls_at_start).tolist()


def test_sort_indices_reference_table(be):
    # TestSortIndices :40-300
    assert run_sort(be, np.int32, [3, 1, 4, 1, 5, 9, 2, 6]) == [1, 3, 6, 0, 2, 4, 7, 5]
    assert run_sort(be, np.int32, [3, 1, 4, 1, 5, 9, 2, 6], descending=True) == [5, 7, 4, 2, 0, 6, 1, 3]
    assert run_sort(be, np.int32, [3, None, 4, 0, 5]) == [3, 0, 2, 4, 1]
    assert run_sort(be, np.int32, [3, None, 4, 0, 5], nulls_at_start=True) == [1, 3, 0, 2, 4]
    assert run_sort(be, np.float64, [3.14, float("nan"), 2.71, 1.41, float("nan")]) == [3, 2, 0, 1, 4]
    assert run_sort(be, np.int32, []) == []
    assert run_sort(be, np.int32, [None, None, None]) == [0, 1, 2]
    assert run_sort(be, np.int32, [1, 2, 1, 2, 1]) == [0, 2, 4, 1, 3]                       # StableSort
    assert run_sort(be, np.uint64, [100, 50, 200, 25]) == [3, 1, 0, 2]
    assert run_sort(be, np.float32, [3.0, 1.0, 4.0, 1.0, 5.0, 9.0, 2.0, 6.0]) == [1, 3, 6, 0, 2, 4, 7, 5]


def test_sort_indices_cpp_parity_vectors(be):
    # TestVectorSortIndicesCppArrayParity :1186-1240
    v = [0, 1, None, -3, None, -42, 5]
    assert run_sort(be, np.int16, v) == [5, 3, 0, 1, 6, 2, 4]
    assert run_sort(be, np.int16, v, descending=True, nulls_at_start=True) == [2, 4, 6, 1, 0, 3, 5]
    f = [None, 1, 3.3, None, 2, 5.3]
    assert run_sort(be, np.float64, f) == [1, 4, 2, 5, 0, 3]
    assert run_sort(be, np.float64, f, nulls_at_start=True) == [0, 3, 1, 4, 2, 5]
    assert run_sort(be, np.float64, f, descending=True) == [5, 2, 4, 1, 0, 3]
    assert run_sort(be, np.float64, f, descending=True, nulls_at_start=True) == [0, 3, 5, 2, 4, 1]
    u = [255, None, 0, 255, 10, None, 128, 0]
    assert run_sort(be, np.uint8, u) == [2, 7, 4, 6, 0, 3, 1, 5]
    assert run_sort(be, np.uint8, u, nulls_at_start=True) == [1, 5, 2, 7, 4, 6, 0, 3]


@pytest.mark.parametrize("dtype", OL.ALL_DTYPES, ids=str)
def test_sort_indices_rules(be, dtype):
    # ties keep their input order in BOTH directions; NaNs sit next to the nulls whatever the order;
    # −0.0 and +0.0 tie; a slice sorts its own rows (indices relative to the slice)
    assert run_sort(be, dtype, [2, 1, 2, 1], descending=True) == [0, 2, 1, 3]
    assert run_sort(be, dtype, [9, 5, None, 7, 5], sl=(1, 5)) == [0, 3, 2, 1]
    if np.dtype(dtype).kind == "f":
        nan = float("nan")
        v = [1.0, nan, None, -0.0, 0.0, float("-inf"), nan, float("inf")]
        assert run_sort(be, dtype, v) == [5, 3, 4, 0, 7, 1, 6, 2]
        assert run_sort(be, dtype, v, descending=True) == [7, 0, 3, 4, 5, 1, 6, 2]
        assert run_sort(be, dtype, v, nulls_at_start=True) == [2, 1, 6, 5, 3, 4, 0, 7]
        assert run_sort(be, dtype, v, descending=True, nulls_at_start=True) == [2, 1, 6, 7, 0, 3, 4, 5]
    else:
        info = np.iinfo(dtype)
        assert run_sort(be, dtype, [info.max, info.min, 1, info.max]) == [1, 2, 0, 3]
        assert run_sort(be, dtype, [info.max, info.min, 1, info.max], descending=True) == [0, 3, 2, 1]


# ---- min_max ---------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", OL.INT_DTYPES, ids=str)
def test_min_max(be, dtype):
    # internal/utils/min_max.go:30-148; internal/utils/min_max_test.go (random slices vs a plain loop)
    info = np.iinfo(dtype)
    assert be.min_max(np.array([], dtype)) == (info.max, info.min)          # empty: the accumulators' start values
    assert be.min_max(np.array([5], dtype)) == (5, 5)
    assert be.min_max(np.array([3, 1, 4, 1, 5, 9, 2, 6], dtype)) == (1, 9)
    assert be.min_max(np.array([info.max, 7, info.min, 7], dtype)) == (info.min, info.max)
    rng = np.random.default_rng(np.dtype(dtype).itemsize)
    for n in (1, 2, 15, 16, 17, 255, 1000, 4099, 70001):
        a = rng.integers(info.min, info.max, n, dtype=dtype, endpoint=True)
        assert be.min_max(a) == (a.min(), a.max()), n


# ---- sort_indices, several keys (record batch) -----------------------------------------------
def run_sort_multi(be, cols):
    """cols: [(dtype, values-with-None, descending, nulls_at_start), …]"""
    args = []
    for dtype, vals, desc, nfirst in cols:
        arr, valid = mk(vals, dtype, null_fill=0)
        args.append((arr, valid, 0, desc, nfirst))
    return be.sort_indices_multi(args).tolist()


def test_sort_indices_record_batch_cpp_parity(be):
    # TestVectorSortIndicesCppRecordBatchParity :1346-1460 — keys (a ascending, b descending)
    nan = float("nan")
    for nfirst in (False, True):
        assert run_sort_multi(be, [(np.uint8, [3, 1, 3, 0, 2, 1, 1], False, nfirst), (np.uint32, [5, 3, 4, 6, 5, 5, 3], True, nfirst)]) == [3, 5, 1, 6, 4, 0, 2]
    a = [None, 1, 3, None, 2, 1, 3]; b = [5, 3, None, None, 5, 5, 5]
    assert run_sort_multi(be, [(np.uint8, a, False, False), (np.uint32, b, True, False)]) == [5, 1, 4, 6, 2, 0, 3]
    assert run_sort_multi(be, [(np.uint8, a, False, True), (np.uint32, b, True, True)]) == [3, 0, 5, 1, 4, 2, 6]
    a = [3, 1, 3, 0, nan, nan, nan, 1]; b = [5, nan, 4, 6, 5, nan, 5, 5]
    assert run_sort_multi(be, [(np.float32, a, False, False), (np.float64, b, True, False)]) == [3, 7, 1, 0, 2, 4, 6, 5]
    assert run_sort_multi(be, [(np.float32, a, False, True), (np.float64, b, True, True)]) == [5, 4, 6, 3, 1, 7, 0, 2]
    a = [None, 1, 3, None, nan, nan, nan, 1]; b = [5, 3, None, None, None, nan, 5, 5]
    assert run_sort_multi(be, [(np.float32, a, False, False), (np.float64, b, True, False)]) == [7, 1, 2, 6, 5, 4, 0, 3]
    assert run_sort_multi(be, [(np.float32, a, False, True), (np.float64, b, True, True)]) == [3, 0, 4, 5, 6, 7, 1, 2]


def test_sort_indices_record_batch_table(be):
    # TestSortRecordBatch :554-700: (value asc) and (category asc, priority desc)-style keys, three keys, mixed placement
    assert run_sort_multi(be, [(np.int32, [30, 10, 20], False, False)]) == [1, 2, 0]
    cat = [2, 1, 2, 1]; val = [1, 1, 2, 2]; pri = [100, 200, 300, 400]
    assert run_sort_multi(be, [(np.int32, val, False, False), (np.int32, pri, True, True)]) == [1, 0, 3, 2]
    assert run_sort_multi(be, [(np.int8, cat, False, False), (np.int32, val, True, False), (np.int64, pri, False, False)]) == [3, 1, 2, 0]
    assert run_sort_multi(be, [(np.int8, [1, 1, 1], False, False), (np.float64, [None, 2.5, None], False, True), (np.uint16, [9, 8, 7], False, False)]) == [2, 0, 1]


# ---- var-length (binary / string) Take and Filter ----------------------------------------------
def mk_binary(vals, offset_dtype, pad=b""):
    """python list of bytes / None → (offsets, data, validity or None); nulls carry `pad` as payload
    so that a null's bytes must not leak into the output"""
    offs, data, valid = [0], b"", []
    for v in vals:
        b = pad if v is None else v
        data += b
        offs.append(len(data))
        valid.append(v is not None)
    return np.array(offs, offset_dtype), np.frombuffer(data, np.uint8), (None if all(valid) else OL.pack_bits(valid))


def un_binary(offsets, data, valid, n):
    bits = OL.unpack_bits(valid, 0, n) if valid is not None else [1] * n
    out = []
    for i in range(n):
        lo, hi = int(offsets[i]), int(offsets[i + 1])
        if not bits[i]:
            assert lo == hi          # a null output has zero length (VarBinaryImpl :1979-1982)
            out.append(None)
        else:
            out.append(bytes(data[lo:hi]))
    return out


def run_take_binary(be, odt, vals, idtype, idx, sl=None):
    offsets, data, vvalid = mk_binary(vals, odt, pad=b"??")
    ia, ivalid = mk(idx, idtype)
    lo, hi = sl if sl else (0, len(vals))
    st, oo, od, ov, nulls, bad = be.take_binary(offsets, data, vvalid, lo, hi - lo, ia, ivalid, 0, True)
    if st != STATUS_OK:
        return st, bad
    assert oo[0] == 0 and oo[-1] == len(od)
    got = un_binary(oo, od, ov, len(idx))
    assert nulls == sum(g is None for g in got)
    return st, got


@pytest.mark.parametrize("odt", [np.int32, np.int64], ids=["binary", "large_binary"])
def test_take_string(be, odt):
    # TestTakeString :1193-1210 (a, b, c)
    a, b, c = b"a", b"b", b"c"
    assert run_take_binary(be, odt, [a, b, c], np.int32, [0, 1, 0]) == (STATUS_OK, [a, b, a])
    assert run_take_binary(be, odt, [None, b, c], np.int32, [0, 1, 0]) == (STATUS_OK, [None, b, None])
    assert run_take_binary(be, odt, [a, b, c], np.int32, [None, 1, 0]) == (STATUS_OK, [None, b, a])
    assert run_take_binary(be, odt, [a, b, c], np.int8, [0, 9, 0]) == (STATUS_EINDEX, 9)
    assert run_take_binary(be, odt, [a, b, c], np.int64, [2, 5]) == (STATUS_EINDEX, 5)
    # longer values, empty strings, repeats, a sliced values array (Offset applies to offsets and validity)
    v = [b"", b"hello world, this is a longer value " * 3, None, b"x", b"", b"\x00\xff binary"]
    assert run_take_binary(be, odt, v, np.uint16, [5, 1, 1, 0, 4, 2, 3]) == (STATUS_OK, [v[5], v[1], v[1], b"", b"", None, b"x"])
    assert run_take_binary(be, odt, v, np.int32, [0, 3, 1], sl=(2, 6)) == (STATUS_OK, [None, v[5], b"x"])
    assert run_take_binary(be, odt, v, np.int32, []) == (STATUS_OK, [])


@pytest.mark.parametrize("odt", [np.int32, np.int64], ids=["binary", "large_binary"])
def test_filter_string(be, odt):
    # TestFilterString :698-704
    a, b, c = b"a", b"b", b"c"

    def run(vals, filt, null_sel):
        offsets, data, vvalid = mk_binary(vals, odt, pad=b"??")
        fd = OL.pack_bits([bool(x) for x in filt]); fv = None if all(x is not None for x in filt) else OL.pack_bits([x is not None for x in filt])
        oo, od, ov, nulls = be.filter_binary(offsets, data, vvalid, 0, fd, fv, 0, len(vals), null_sel, True)
        got = un_binary(oo, od, ov, len(oo) - 1)
        assert nulls == sum(g is None for g in got)
        return got

    assert run([a, b, c], [False, True, False], DROP) == [b]
    assert run([None, b, c], [False, True, False], DROP) == [b]
    assert run([a, b, c], [None, True, False], EMIT) == [None, b]
    assert run([a, b, c], [None, True, False], DROP) == [b]
    assert run([a, None, c * 40, b""], [True, True, True, True], DROP) == [a, None, c * 40, b""]
    assert run([], [], DROP) == []


# ---- unique / dictionary_encode over String / Binary ------------------------------------------------
def bin_dictionary(be, offsets, data, valid, off, n, first_rows, null_id):
    """dictionary = take(values, first rows) — what GetDictArrayData (arrow/array/util.go:341-366) reads out of
    the memo table's builder; the null entry comes out as None"""
    st, oo, od, ov, nulls, _ = be.take_binary(offsets, data, valid, off, n, first_rows, None, 0, valid is not None)
    assert st == 0
    return un_binary(oo, od, ov, len(first_rows))


@pytest.mark.parametrize("odt", [np.int32, np.int64], ids=["string", "large_string"])
def test_unique_binary_vectors(be, odt):
    # vector_hash_test.go:272-279 (BinaryTypeHashKernelSuite.TestUnique, String / LargeString / Binary / LargeBinary)
    offsets, data, valid = mk_binary([b"test", None, b"test2", b"test"], odt, pad=b"")
    ids, idv, fr, null_id = be.hash_binary_encode(offsets, data, valid, 0, 4, True)
    assert bin_dictionary(be, offsets, data, valid, 0, 4, fr, null_id) == [b"test", None, b"test2"] and null_id == 1
    # :420-449 TestUniqueChunkedArrayInvoke, the two chunks laid end to end
    offsets, data, valid = mk_binary([b"foo", b"bar", b"foo", b"bar", b"baz", b"quuux", b"foo"], odt)
    ids, idv, fr, null_id = be.hash_binary_encode(offsets, data, valid, 0, 7, True)
    assert bin_dictionary(be, offsets, data, valid, 0, 7, fr, null_id) == [b"foo", b"bar", b"baz", b"quuux"] and null_id == -1


@pytest.mark.parametrize("odt", [np.int32, np.int64], ids=["string", "large_string"])
def test_dictionary_encode_binary_vectors(be, odt):
    # vector_hash_test.go:541-594 TestDictionaryEncode
    offsets, data, valid = mk_binary([b"foo", b"bar", b"foo", None, b"bar", None], odt, pad=b"zz")
    ids, idv, fr, null_id = be.hash_binary_encode(offsets, data, valid, 0, 6, False)
    assert ids.tolist() == [0, 1, 0, 0, 1, 0] and null_id == -1
    assert OL.unpack_bits(idv, 0, 6).tolist() == [True, True, True, False, True, False]
    assert bin_dictionary(be, offsets, data, valid, 0, 6, fr, null_id) == [b"foo", b"bar"]
    ids, idv, fr, null_id = be.hash_binary_encode(offsets, data, valid, 0, 6, True)
    assert ids.tolist() == [0, 1, 0, 2, 1, 2] and null_id == 2 and OL.unpack_bits(idv, 0, 6).all()
    assert bin_dictionary(be, offsets, data, valid, 0, 6, fr, null_id) == [b"foo", b"bar", None]
    # :802-825 TestDictionaryEncodeArraySlicedInput: ["ignored","foo",null,"bar","foo","ignored"][1:5]
    offsets, data, valid = mk_binary([b"ignored", b"foo", None, b"bar", b"foo", b"ignored"], odt)
    ids, idv, fr, null_id = be.hash_binary_encode(offsets, data, valid, 1, 4, False)
    assert ids.tolist() == [0, 0, 1, 0] and OL.unpack_bits(idv, 0, 4).tolist() == [True, False, True, True]
    assert bin_dictionary(be, offsets, data, valid, 1, 4, fr, null_id) == [b"foo", b"bar"]
    # a null whose slot still holds bytes equal to a real value is not that value
    offsets, data, valid = mk_binary([b"ab", None, b"ab"], odt, pad=b"ab")
    ids, idv, fr, null_id = be.hash_binary_encode(offsets, data, valid, 0, 3, True)
    assert ids.tolist() == [0, 1, 0] and null_id == 1
    # empty input
    ids, idv, fr, null_id = be.hash_binary_encode(np.zeros(1, odt), np.zeros(0, np.uint8), None, 0, 0, True)
    assert len(ids) == 0 and len(fr) == 0 and null_id == -1


# ---- divide / shifts / bit-wise / abs / negate / sqrt (arrow/compute/arithmetic_test.go) -------------------------
X = dict(DIV=3, SQRT=6, DIV_CHECKED=24, ABS_CHECKED=25, NEGATE_CHECKED=26, SQRT_CHECKED=27, SHL=64, SHL_CHECKED=65, SHR=66,
         SHR_CHECKED=67, AND=68, OR=69, XOR=70, NOT=71)
_INTS = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]
_SIGNED = [np.int8, np.int16, np.int32, np.int64]
_FLOATS = [np.float32, np.float64]


def ext_binop(be, op, l, r, dtype, fill=7):
    """both sides python lists with None → (status, logical result with the null-intersection validity, message, raw)"""
    la, lv = mk(l, dtype, fill); ra, rv = mk(r, dtype, fill)
    st, out, msg = be.arithmetic_ext(op, 0, la, lv, 0, ra, rv, 0)
    valid = [(a is not None) and (b is not None) for a, b in zip(l, r)]
    return st, [x if ok else None for x, ok in zip(out.tolist(), valid)], msg, out


@pytest.mark.parametrize("dtype", _INTS + _FLOATS, ids=str)
def test_divide_vectors(be, dtype):
    # arithmetic_test.go:427-459 TestDiv, for divide (checked) and divide_unchecked alike
    for op in (X["DIV"], X["DIV_CHECKED"]):
        assert ext_binop(be, op, [3, 2, 6], [1, 1, 2], dtype)[:2] == (0, [3, 2, 3])
        st, res, _, raw = ext_binop(be, op, [None, 10, 30, None, 20], [1, 5, 2, 5, 10], dtype)
        assert st == 0 and res == [None, 2, 15, None, 2]
        assert raw[0] == 0 and raw[3] == 0                                   # ScalarBinaryNotNull: null slots hold 0
        arr, v = mk([None, 1, 3, None, 2] if np.dtype(dtype).kind != "f" else [None, 1, 2.5, None, 2], dtype, 9)
        s = np.array([33 if np.dtype(dtype).kind != "f" else 10], dtype)
        st, out, _ = be.arithmetic_ext(op, 2, s, None, 0, arr, v, 0)       # scalar ÷ array
        assert st == 0 and out.tolist() == ([0, 33, 11, 0, 16] if np.dtype(dtype).kind != "f" else [0, 10, 4, 0, 5])
        arr, v = mk([None, 10, 30, None, 2], dtype, 9)
        st, out, _ = be.arithmetic_ext(op, 1, arr, v, 0, np.array([3 if np.dtype(dtype).kind != "f" else 10], dtype), None, 0)   # array ÷ scalar
        assert st == 0 and out.tolist()[1:3] == ([3, 10] if np.dtype(dtype).kind != "f" else [1, 3])
    # :462-481 TestDivideByZero
    if np.dtype(dtype).kind != "f":
        for op in (X["DIV"], X["DIV_CHECKED"]):
            st, _, msg, _ = ext_binop(be, op, [3, 2, 6], [1, 1, 0], dtype)
            assert st == 1 and "divide by zero" in msg
        assert ext_binop(be, X["DIV"], [3, None, 6], [1, 0, 2], dtype)[:2] == (0, [3, None, 3])   # a zero divisor under a null is not looked at
    else:
        for l in ([3, 2, 6], [3, 2, 0], [3, 2, -6]):
            st, _, msg, _ = ext_binop(be, X["DIV_CHECKED"], l, [1, 1, 0], dtype)
            assert st == 1 and "divide by zero" in msg
        with np.errstate(all="ignore"):
            assert ext_binop(be, X["DIV"], [3, 2, 6], [1, 1, 0], dtype)[1] == [3, 2, float("inf")]
            assert ext_binop(be, X["DIV"], [3, 2, -6], [1, 1, 0], dtype)[1] == [3, 2, float("-inf")]
            assert np.isnan(ext_binop(be, X["DIV"], [3, 2, 0], [1, 1, 0], dtype)[1][2])
    if dtype in _SIGNED:   # Go: the quotient of MinInt / −1 wraps, and division truncates toward zero
        mn = np.iinfo(dtype).min
        assert ext_binop(be, X["DIV"], [mn, -7, 7], [-1, 2, -2], dtype)[1] == [mn, -3, -3]


@pytest.mark.parametrize("dtype", _INTS, ids=str)
def test_shift_vectors(be, dtype):
    # :571-611 TestShiftLeft / TestShiftRight
    for op in (X["SHL"], X["SHL_CHECKED"]):
        assert ext_binop(be, op, [0, 1, 2, 3], [2, 3, 4, 5], dtype)[:2] == (0, [0, 8, 32, 96])
        assert ext_binop(be, op, [0, None, 2, 3], [2, 3, None, 5], dtype)[:2] == (0, [0, None, None, 96])
        st, out, _ = be.arithmetic_ext(op, 2, np.array([2], dtype), None, 0, *mk([None, 5], dtype, 1), 0)
        assert st == 0 and out.tolist() == [0, 64]
        st, out, _ = be.arithmetic_ext(op, 1, *mk([None, 5], dtype, 1), 0, np.array([3], dtype), None, 0)
        assert st == 0 and out.tolist() == [0, 40]
        st, out, _ = be.arithmetic_ext(op, 1, *mk([None, 5], dtype, 1), 0, np.array([3], dtype), None, 0, scalar_valid=False)
        assert st == 0 and out.tolist() == [0, 0]                           # null scalar: everything null, output zero
    for op in (X["SHR"], X["SHR_CHECKED"]):
        assert ext_binop(be, op, [0, 1, 4, 8], [1, 1, 1, 4], dtype)[:2] == (0, [0, 0, 2, 0])
        assert ext_binop(be, op, [0, None, 4, 8], [1, 1, None, 4], dtype)[:2] == (0, [0, None, None, 0])
        st, out, _ = be.arithmetic_ext(op, 2, np.array([64], dtype), None, 0, *mk([None, 2, 6], dtype, 1), 0)
        assert st == 0 and out.tolist() == [0, 16, 1]
    # :613-671 the overflow tables
    info = np.iinfo(dtype)
    signed = info.min < 0
    bw = info.bits - (1 if signed else 0)
    err = "shift amount must be >= 0 and less than precision of type"
    assert ext_binop(be, X["SHL_CHECKED"], [1], [bw - 1], dtype)[1] == [int(dtype(1) << dtype(bw - 1))]
    assert ext_binop(be, X["SHL_CHECKED"], [2], [bw - 2], dtype)[1] == [int(dtype(1) << dtype(bw - 1))]
    assert ext_binop(be, X["SHR_CHECKED"], [info.max], [bw - 1], dtype)[1] == [1]
    if not signed:
        assert ext_binop(be, X["SHL_CHECKED"], [2, 4], [bw - 1, bw - 1], dtype)[1] == [0, 0]
        for op in (X["SHL_CHECKED"], X["SHR_CHECKED"]):
            st, _, msg, _ = ext_binop(be, op, [1], [bw], dtype)
            assert st == 1 and err in msg
    else:
        assert ext_binop(be, X["SHL_CHECKED"], [2, 4, info.min], [bw - 1, bw - 1, 1], dtype)[1] == [info.min, 0, 0]
        assert ext_binop(be, X["SHR_CHECKED"], [-1, -1, info.min], [1, 5, 1], dtype)[1] == [-1, -1, info.min // 2]
        for op, un in ((X["SHL_CHECKED"], X["SHL"]), (X["SHR_CHECKED"], X["SHR"])):
            st, _, msg, _ = ext_binop(be, op, [1, 2], [1, -1], dtype)
            assert st == 1 and err in msg
            st, _, msg, _ = ext_binop(be, op, [1], [bw], dtype)
            assert st == 1 and err in msg
            assert ext_binop(be, un, [1, 1], [-1, bw], dtype)[:2] == (0, [1, 1])     # unchecked: the left operand comes back


@pytest.mark.parametrize("dtype", _INTS, ids=str)
def test_bitwise_vectors(be, dtype):
    # :3045-3063 TestBitWiseAnd / Or / Xor (the byte patterns, in every integer width); bit_wise_not :253-268
    w = np.dtype(dtype).itemsize
    pat = lambda bs: np.frombuffer(bytes(b for b in bs for _ in range(w)), dtype)
    a, b = pat([0x00, 0xFF, 0x00, 0xFF]), pat([0x00, 0x00, 0xFF, 0xFF])
    for op, exp in ((X["AND"], [0x00, 0x00, 0x00, 0xFF]), (X["OR"], [0x00, 0xFF, 0xFF, 0xFF]), (X["XOR"], [0x00, 0xFF, 0xFF, 0x00])):
        st, out, _ = be.arithmetic_ext(op, 0, a, None, 0, b, None, 0)
        assert st == 0 and out.tobytes() == pat(exp).tobytes()
        # the value buffers are combined in every slot: validity does not change the bytes
        st, out2, _ = be.arithmetic_ext(op, 0, a, OL.pack_bits([True, False, True, False]), 0, b, None, 0)
        assert out2.tobytes() == out.tobytes()
    arr, v = mk([0, None, 5, -1 if np.iinfo(dtype).min < 0 else np.iinfo(dtype).max], dtype, 3)
    st, out, _ = be.arithmetic_ext(X["NOT"], 1, arr, v, 0, None, None, 0)
    assert st == 0 and out.tolist() == [int(dtype(~dtype(0))), 0, int(dtype(~dtype(5))), 0 if np.iinfo(dtype).min < 0 else 0]


@pytest.mark.parametrize("dtype", _INTS + _FLOATS, ids=str)
def test_abs_negate_checked_vectors(be, dtype):
    # :2631-2760: abs over signed / unsigned / floating, negate over signed / floating
    kind = np.dtype(dtype).kind
    un = lambda op, vals, fill=3: be.arithmetic_ext(op, 1, *mk(vals, dtype, fill), 0, None, None, 0)
    if kind == "u":
        st, out, _ = un(X["ABS_CHECKED"], [0, 1, 10, 127, np.iinfo(dtype).max])
        assert st == 0 and out.tolist() == [0, 1, 10, 127, np.iinfo(dtype).max]
        return
    st, out, _ = un(X["ABS_CHECKED"], [1, None, -10, -1, -127, 0])
    assert st == 0 and out.tolist()[0] == 1 and out.tolist()[2:] == [10, 1, 127, 0]
    st, out, _ = un(X["NEGATE_CHECKED"], [1, None, -10, 127, -127])
    assert st == 0 and out.tolist()[0] == -1 and out.tolist()[2:] == [10, -127, 127]
    if kind == "i":
        mn, mx = np.iinfo(dtype).min, np.iinfo(dtype).max
        assert un(X["ABS_CHECKED"], [mx])[1].tolist() == [mx]
        assert un(X["NEGATE_CHECKED"], [mn + 1, mx])[1].tolist() == [mx, mn + 1]
        for op in (X["ABS_CHECKED"], X["NEGATE_CHECKED"]):
            st, _, msg = un(op, [1, mn])
            assert st == 3 and "overflow" in msg
            # ScalarUnary walks the whole value buffer: MinInt under a NULL slot overflows as well (helpers.go:56-90)
            st, _, msg = un(op, [1, None], fill=mn)
            assert st == 3 and "overflow" in msg
    else:
        st, out, _ = un(X["ABS_CHECKED"], [-0.0, float("-inf"), 1.5])
        assert st == 0 and out.tolist() == [0.0, float("inf"), 1.5] and not np.signbit(out[0])
        st, out, _ = un(X["NEGATE_CHECKED"], [0.0, float("inf")])
        assert np.signbit(out[0]) and out[1] == float("-inf")


@pytest.mark.parametrize("dtype", _FLOATS, ids=str)
def test_sqrt_vectors(be, dtype):
    # base_arithmetic.go:412-426; arithmetic_test.go TestSqrt
    un = lambda op, vals, fill=4: be.arithmetic_ext(op, 1, *mk(vals, dtype, fill), 0, None, None, 0)
    for op in (X["SQRT"], X["SQRT_CHECKED"]):
        st, out, _ = un(op, [0, 1, 4, 2.25, float("inf")])
        assert st == 0 and out.tolist() == [0, 1, 2, 1.5, float("inf")]
        st, out, _ = un(op, [9, -0.0])
        assert st == 0 and out[0] == 3 and np.signbit(out[1])
    with np.errstate(all="ignore"):
        st, out, _ = un(X["SQRT"], [4, -1, None], fill=-4)
        assert st == 0 and out[0] == 2 and np.isnan(out[1]) and np.isnan(out[2])   # unchecked: every slot, null payloads too
    st, _, msg = un(X["SQRT_CHECKED"], [4, -1])
    assert st == 1 and "square root of negative number" in msg
    st, out, _ = un(X["SQRT_CHECKED"], [4, None], fill=-4)
    assert st == 0 and out.tolist() == [2, 0]
    # correctly rounded: equals numpy's IEEE sqrt on random inputs
    x = np.random.default_rng(1).uniform(0, 1e6, 4099).astype(dtype)
    st, out, _ = be.arithmetic_ext(X["SQRT"], 1, x, None, 0, None, None, 0)
    assert out.tobytes() == np.sqrt(x).tobytes()


@pytest.mark.parametrize("dtype", _FLOATS, ids=str)
def test_floor_ceil_trunc_vectors(be, dtype):
    # arithmetic_test.go TestRounding*: floor / ceil / trunc tables (rounding.go:180-187)
    vals = [3.2, 3.5, 3.7, 4.5, -3.2, -3.5, -3.7, 0.0, -0.0, float("inf"), float("-inf")]
    exp = {72: [3, 3, 3, 4, -4, -4, -4, 0, -0.0, float("inf"), float("-inf")],
           73: [4, 4, 4, 5, -3, -3, -3, 0, -0.0, float("inf"), float("-inf")],
           74: [3, 3, 3, 4, -3, -3, -3, 0, -0.0, float("inf"), float("-inf")]}
    arr = np.array(vals, dtype)
    for op, e in exp.items():
        st, out, _ = be.arithmetic_ext(op, 1, arr, None, 0, None, None, 0)
        assert st == 0 and out.tolist() == e and np.signbit(out[8])
        st, out, _ = be.arithmetic_ext(op, 1, np.array([np.nan, 2.5], dtype), OL.pack_bits([True, False]), 0, None, None, 0)
        assert np.isnan(out[0]) and out[1] == {72: 2, 73: 3, 74: 2}[op]       # every slot, nulls included
    x = (np.random.default_rng(2).standard_normal(5003) * 1e3).astype(dtype)
    for op, f in ((72, np.floor), (73, np.ceil), (74, np.trunc)):
        assert be.arithmetic_ext(op, 1, x, None, 0, None, None, 0)[1].tobytes() == f(x).tobytes()


# ---- round / round_to_multiple (arithmetic_test.go:3268-3366) -------------------------------------------------------
_RMODE = dict(DOWN=0, UP=1, TOWARDS_ZERO=2, TOWARDS_INFINITY=3, HALF_DOWN=4, HALF_UP=5, HALF_TOWARDS_ZERO=6, HALF_TOWARDS_INFINITY=7,
              HALF_TO_EVEN=8, HALF_TO_ODD=9)
_ROUND_TABLE = [("DOWN", [3, 3, 3, 4, -4, -4, -4]), ("UP", [4, 4, 4, 5, -3, -3, -3]), ("TOWARDS_ZERO", [3, 3, 3, 4, -3, -3, -3]),
                ("TOWARDS_INFINITY", [4, 4, 4, 5, -4, -4, -4]), ("HALF_DOWN", [3, 3, 4, 4, -3, -4, -4]), ("HALF_UP", [3, 4, 4, 5, -3, -3, -4]),
                ("HALF_TOWARDS_ZERO", [3, 3, 4, 4, -3, -3, -4]), ("HALF_TO_EVEN", [3, 4, 4, 4, -3, -4, -4]), ("HALF_TO_ODD", [3, 3, 4, 5, -3, -3, -4])]


@pytest.mark.parametrize("dtype", _FLOATS, ids=str)
def test_round_vectors(be, dtype):
    vals = np.array([3.2, 3.5, 3.7, 4.5, -3.2, -3.5, -3.7], dtype)
    for mode, exp in _ROUND_TABLE:
        for kw in ({}, {"multiple": 1}):                       # TestRound and TestRoundToMultiple share the table
            st, out = be.round(vals, None, 0, 0, _RMODE[mode], **kw)
            assert st == 0 and out.tolist() == exp, (mode, kw)
            sp, v = mk([None, 0, float("inf"), float("-inf"), float("nan")], dtype, 7.5)
            st, out = be.round(sp, v, 0, 0, _RMODE[mode], **kw)
            assert st == 0 and out[0] == 0 and out[1] == 0 and out[2] == np.inf and out[3] == -np.inf and np.isnan(out[4])
    vals = np.array([320, 3.5, 3.075, 4.5, -3.212, -35.1234, -3.045], dtype)
    close = lambda a, b: np.allclose(a, np.array(b, dtype), rtol=1e-6 if dtype == np.float32 else 1e-12, atol=0)
    for nd, exp in ((-2, [300, 0.0, 0.0, 0.0, -0.0, -0.0, -0.0]), (-1, [320, 0.0, 0.0, 0.0, -0.0, -40, -0.0]), (0, [320, 4, 3, 5, -3, -35, -3]),
                    (1, [320, 3.5, 3.1, 4.5, -3.2, -35.1, -3]), (2, [320, 3.5, 3.08, 4.5, -3.21, -35.12, -3.05])):
        st, out = be.round(vals, None, 0, nd, _RMODE["HALF_TOWARDS_INFINITY"])
        assert st == 0 and close(out, exp), (nd, out)
    for mult, exp in ((0.05, [320, 3.5, 3.1, 4.5, -3.2, -35.1, -3.05]), (0.1, [320, 3.5, 3.1, 4.5, -3.2, -35.1, -3]), (2, [320, 4, 4, 4, -4, -36, -4]),
                      (10, [320, 0.0, 0.0, 0.0, -0.0, -40, -0.0]), (100, [300, 0.0, 0.0, 0.0, -0.0, -0.0, -0.0])):
        st, out = be.round(vals, None, 0, 0, _RMODE["HALF_TOWARDS_INFINITY"], multiple=mult)
        assert st == 0 and close(out, exp), (mult, out)
    # overflow: the rescaled result leaves the type's range
    big = np.array([np.finfo(dtype).max], dtype)
    assert be.round(big, None, 0, -int(np.log10(np.finfo(dtype).max)), _RMODE["UP"])[0] == 3
    assert be.round(big, OL.pack_bits([False]), 0, -int(np.log10(np.finfo(dtype).max)), _RMODE["UP"])[0] == 0   # not under a null


def test_pow10_is_gos_table(orc):
    # math.Pow10: pow10tab[n % 32] · pow10postab32[n / 32] — the product, not the correctly rounded literal, beyond 1e31
    assert orc.pow10(0) == 1.0 and orc.pow10(22) == 1e22 and orc.pow10(31) == 1e31
    assert orc.pow10(40) == 1e32 * 1e8 and orc.pow10(308) == 1e288 * 1e20 and orc.pow10(309) == float("inf")


# ---- ShiftTime: unit changes of temporal columns (arrow/compute/cast_test.go) ----------------------------------
SHIFT_MUL, SHIFT_DIV = 0, 1
_SHIFT_CASES = [  # (coarse dtype, fine dtype, factor): TestTimestampToTimestamp :2650-2693, TestTimeToTime :2964-3038, TestDurationToDuration
    (np.int64, np.int64, 1000), (np.int64, np.int64, 1000000000), (np.int32, np.int32, 1000), (np.int32, np.int64, 1000),
    (np.int32, np.int64, 1000000), (np.int32, np.int64, 1000000000)]


@pytest.mark.parametrize("coarse_t,fine_t,factor", _SHIFT_CASES, ids=lambda v: getattr(v, "__name__", str(v)))
def test_shift_time_reference_vectors(be, coarse_t, fine_t, factor):
    coarse, cv = mk([0, None, 200, 1, 2], coarse_t)
    st, out, _ = be.shift_time(coarse, fine_t, SHIFT_MUL, factor, True, cv)
    assert st == STATUS_OK and logical(out, cv, 0, 5) == [0, None, 200 * factor, factor, 2 * factor]
    lossy, lv = mk([0, None, 200 * factor + 456 * (factor // 1000), factor + 123 * (factor // 1000), 2 * factor + 456 * (factor // 1000)], fine_t)
    st, _, bad = be.shift_time(lossy, coarse_t, SHIFT_DIV, factor, True, lv)      # AllowTimeTruncate = false: "would lose data"
    assert st == STATUS_EINVALID and bad == 200 * factor + 456 * (factor // 1000)
    st, out, _ = be.shift_time(lossy, coarse_t, SHIFT_DIV, factor, False, lv)     # AllowTimeTruncate = true: divide / truncate
    assert st == STATUS_OK and logical(out, lv, 0, 5) == [0, None, 200, 1, 2]


def test_shift_time_dates_and_overflow(be):
    # TestDateToDate :3052-3069
    d32, dv = mk([0, None, 100, 1, 10], np.int32)
    st, out, _ = be.shift_time(d32, np.int64, SHIFT_MUL, 86400000, True, dv)
    assert st == STATUS_OK and logical(out, dv, 0, 5) == [0, None, 8640000000, 86400000, 864000000]
    st, back, _ = be.shift_time(out, np.int32, SHIFT_DIV, 86400000, True, dv)
    assert st == STATUS_OK and logical(back, dv, 0, 5) == [0, None, 100, 1, 10]
    lossy, lv = mk([0, None, 8640000123, 86400456, 864000789], np.int64)
    st, _, bad = be.shift_time(lossy, np.int32, SHIFT_DIV, 86400000, True, lv)
    assert st == STATUS_EINVALID and bad == 8640000123
    st, out, _ = be.shift_time(lossy, np.int32, SHIFT_DIV, 86400000, False, lv)
    assert st == STATUS_OK and logical(out, lv, 0, 5) == [0, None, 100, 1, 10]
    # TestTimestampToTimestampMultiplyOverflow :2703-2707 (years 1000 … 3000 in seconds → ns), TestDurationToDurationMultiplyOverflow :3166-3169
    far = np.array([-30610224000, -5364662400, 946684800, 10413792000, 32503680000], np.int64)
    st, _, bad = be.shift_time(far, np.int64, SHIFT_MUL, 1000000000, True)
    assert st == STATUS_EINVALID and bad == -30610224000
    st, _, bad = be.shift_time(np.array([10000000000, 1, 2, 3, 10000000000], np.int64), np.int64, SHIFT_MUL, 1000000000, True)
    assert st == STATUS_EINVALID and bad == 10000000000
    # AllowTimeOverflow: the product wraps
    st, out, _ = be.shift_time(far, np.int64, SHIFT_MUL, 1000000000, False)
    assert st == STATUS_OK and out.tolist() == (far.astype(np.uint64) * np.uint64(1000000000)).astype(np.int64).tolist()
    # a failing value in a NULL slot does not fail the cast; the first VALID offender is the one reported
    vals, vv = mk([None, 5, 10413792000, 32503680000], np.int64)
    vals[0] = 32503680000
    st, _, bad = be.shift_time(vals, np.int64, SHIFT_MUL, 1000000000, True, vv)
    assert st == STATUS_EINVALID and bad == 10413792000
    # factor 1 converts the width and never fails (cast_temporal.go:41-45)
    st, out, _ = be.shift_time(np.array([1, -2, 3], np.int32), np.int64, SHIFT_MUL, 1, True)
    assert st == STATUS_OK and out.tolist() == [1, -2, 3]
    # a quotient that does not fit the 32-bit output is "lost data" too (:85)
    st, _, bad = be.shift_time(np.array([0, 2**40 * 1000], np.int64), np.int32, SHIFT_DIV, 1000, True)
    assert st == STATUS_EINVALID and bad == 2**40 * 1000


# ---- power / power_unchecked (arithmetic_test.go:482-510 TestPower; base_arithmetic.go:226-248, 342-373, 443-446) --
X.update(POW=7, POW_CHECKED=28)


@pytest.mark.parametrize("dtype", _INTS, ids=str)
def test_power_integer_vectors(be, dtype):
    for op in (X["POW"], X["POW_CHECKED"]):
        assert ext_binop(be, op, [3, 2, 6, 2], [1, 1, 2, 0], dtype)[:2] == (0, [3, 2, 36, 1])
        assert ext_binop(be, op, [None, 2, 3, None, 20], [1, 6, 2, 5, 1], dtype, fill=1)[:2] == (0, [None, 64, 9, None, 20])
        st, out, _ = be.arithmetic_ext(op, 2, np.array([3], dtype), None, 0, *mk([None, 3, 4, None, 2], dtype, 1), 0)   # scalar ^ array
        assert st == 0 and out.tolist()[1:3] + out.tolist()[4:] == [27, 81, 9]
        st, out, _ = be.arithmetic_ext(op, 1, *mk([None, 10, 3, None, 2], dtype, 1), 0, np.array([2], dtype), None, 0)  # array ^ scalar
        assert st == 0 and out.tolist()[1:3] + out.tolist()[4:] == [100, 9, 4]
        assert ext_binop(be, op, [4], [3], dtype)[1] == [64]
        assert ext_binop(be, op, [0, 1, 0], [0, 0, 42], dtype)[:2] == (0, [1, 1, 0])
    mx = int(np.iinfo(dtype).max)
    st, _, msg, _ = ext_binop(be, X["POW_CHECKED"], [mx], [10], dtype)
    assert st == STATUS_EOVERFLOW and "overflow" in msg
    assert ext_binop(be, X["POW"], [mx], [10], dtype)[:2] == (0, [1])                 # wraps: max^10 ≡ 1
    if dtype in _SIGNED:
        err = "integers to negative integer powers are not allowed"
        for op in (X["POW"], X["POW_CHECKED"]):
            st, _, msg, _ = ext_binop(be, op, [2, 3], [1, -1], dtype)
            assert st == STATUS_EINVALID and err in msg
        # ScalarBinaryNotNull (checked) skips null slots, ScalarBinary (unchecked) runs the closure on them too
        assert ext_binop(be, X["POW_CHECKED"], [2, None], [3, -1], dtype)[:2] == (0, [8, None])
        st, _, msg, _ = ext_binop(be, X["POW"], [2, None], [3, -1], dtype)
        assert st == STATUS_EINVALID and err in msg
        mn = int(np.iinfo(dtype).min)
        bits = np.iinfo(dtype).bits
        assert ext_binop(be, X["POW_CHECKED"], [-2, -1, -1], [bits - 1, 7, 8], dtype)[:2] == (0, [mn, -1, 1])   # (−2)^(bits−1) = MinInt fits
        st, _, msg, _ = ext_binop(be, X["POW_CHECKED"], [2], [bits - 1], dtype)
        assert st == STATUS_EOVERFLOW


@pytest.mark.parametrize("dtype", _FLOATS, ids=str)
def test_power_float_vectors(be, dtype):
    # the reference checks these with array.ApproxEqual; the same values here to 4 ulp of the type
    tol = 4 * np.finfo(dtype).eps
    close = lambda got, want: all((g is None and w is None) or (g == w) or (g is not None and w is not None and abs(g - w) <= tol * abs(w)) for g, w in zip(got, want))
    with np.errstate(all="ignore"):
        for op in (X["POW"], X["POW_CHECKED"]):
            assert close(ext_binop(be, op, [3.4, 16, 0.64, 1.2, 0], [1, 0.5, 2, 4, 0], dtype)[1], [float(dtype(3.4)), 4, float(dtype(0.64)) ** 2, float(dtype(1.2)) ** 4, 1])
            assert close(ext_binop(be, op, [None, 1, 3.3, None, 2], [1, 4, 2, 5, 0.1], dtype)[1], [None, 1, float(dtype(3.3)) ** 2, None, 2 ** float(dtype(0.1))])
            inf = float("inf")
            assert ext_binop(be, op, [3.4, inf, -inf, 1.1, 10000], [1, 2, 3, inf, 100000], dtype)[1][1:] == [inf, -inf, inf, inf]
            assert ext_binop(be, op, [0.0, 0.0], [-1.0, -3.0], dtype)[1] == [inf, inf]
            got = ext_binop(be, op, [3.4, float("nan"), 2.0], [1, 2, 2.0], dtype)[1]
            assert np.isnan(got[1]) and got[2] == 4.0


def test_unique_fixed_width_keys_reference_vectors():
    """TestUniqueFixedSizeBinary / TestUniqueDecimal (vector_hash_test.go:342-389) on the oracle: fixed-width keys go through the
    binary memo table (kernels/vector_hash.go:608-609, 698) — ["aaa", null, "bbb", "aaa"] → ["aaa", null, "bbb"]; decimal128 / 256
    [12, null, 11, 12] → [12, null, 11].  (The same vectors run on the device in test_gpu_parity.py::test_hash_fixed_width_keys.)"""
    from tests.backends import OracleBackend
    o = OracleBackend()
    valid = np.array([0b1101], np.uint8)
    ids, idv, first, nid, dic = o.hash_fixed_encode(np.frombuffer(b"aaa" + b"zzz" + b"bbb" + b"aaa", np.uint8), 3, valid, 0, 4, True)
    assert ids.tolist() == [0, 1, 2, 0] and nid == 1 and first.tolist() == [0, 1, 2] and bytes(dic) == b"aaa\0\0\0bbb"
    for w in (16, 32):
        num = lambda v: np.frombuffer(int(v).to_bytes(w, "little", signed=True), np.uint8)
        ids, idv, first, nid, dic = o.hash_fixed_encode(np.concatenate([num(12), num(12), num(11), num(12)]), w, valid, 0, 4, True)
        assert ids.tolist() == [0, 1, 2, 0] and nid == 1 and bytes(dic) == bytes(num(12)) + bytes(w) + bytes(num(11))
    # nulls masked instead of encoded: the null row gets index 0 and no dictionary entry (dictionaryEncodeAction, vector_hash.go:169-172)
    ids, idv, first, nid, dic = o.hash_fixed_encode(np.frombuffer(b"aaa" + b"zzz" + b"bbb" + b"aaa", np.uint8), 3, valid, 0, 4, False)
    assert ids.tolist() == [0, 0, 1, 0] and nid == -1 and bytes(dic) == b"aaabbb" and idv[0] & 0xF == 0b1101
