"""Pins the oracle's C restatement to the REFERENCE'S OWN CODE run here:
oracle/_ref/libref_avx2.so is the reference's AVX2 machine code (its clang-generated
GAS files, assembled in place by oracle/Makefile) and libref_c.so its C sum kernels
compiled strict-sequential (= the noasm order).  Skipped only if oracle/_ref was never
built (no /root/reference and no prebuilt binaries).
"""
import numpy as np
import pytest

from tests import oracle_lib as OL

ref = OL.load_reference()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (reference sources absent)")


@pytest.fixture(scope="module")
def o():
    return OL.load_oracle()


def rand(rng, dtype, n):
    dt = np.dtype(dtype)
    if dt.kind == "f":
        a = rng.standard_normal(n).astype(dt) * 1e3
        if n > 8:
            a[rng.integers(0, n, 4)] = [np.nan, np.inf, -np.inf, -0.0]
        return a
    info = np.iinfo(dt)
    return rng.integers(info.min, info.max, n, dtype=dt, endpoint=True)


@pytest.mark.parametrize("n", [0, 1, 7, 31, 32, 33, 63, 64, 65, 255, 256, 1000, 8192, 100003])
def test_sum_orders_bit_exact(o, n):
    rng = np.random.default_rng(n)
    a = rng.uniform(-1, 1, n)
    # AVX2 path: our restatement of the 32-strided-partials order == the reference's asm, bit for bit
    assert o.sum_float64_avx2order(a).tobytes() == ref.sum("avx2", a).tobytes()
    # noasm path: strict left-to-right == the reference's C compiled without reassociation
    assert o.sum_float64_seq(a).tobytes() == ref.sum("seq", a).tobytes()
    i = rand(rng, np.int64, n)
    assert o.sum_int64(i) == ref.sum("avx2", i) == ref.sum("seq", i)
    u = rand(rng, np.uint64, n)
    assert o.sum_uint64(u) == ref.sum("avx2", u) == ref.sum("seq", u)


def test_sum_exact_is_between_friends(o):
    # the correctly rounded sum is what both reference orders approximate
    rng = np.random.default_rng(7)
    a = rng.uniform(-1, 1, 1 << 16)
    import math
    assert o.sum_float64_exact(a) == math.fsum(a.tolist())
    big = np.array([1e308, 1.0, -1e308, 1e-300] * 5)
    assert o.sum_float64_exact(big) == math.fsum(big.tolist())


@pytest.mark.parametrize("dtype", OL.ALL_DTYPES, ids=str)
@pytest.mark.parametrize("op", [0, 1, 2, 21, 22, 23])
def test_arithmetic_matches_reference_avx2(o, dtype, op):
    rng = np.random.default_rng(op * 100 + OL.TYPE_IDS[np.dtype(dtype)])
    for n in [0, 1, 3, 15, 16, 17, 31, 33, 64, 1000]:
        l, r = rand(rng, dtype, n), rand(rng, dtype, n)
        s = rand(rng, dtype, 1)
        for shape, a, b in [(0, l, r), (1, l, s), (2, s, r)]:
            if n == 0 and shape != 0:
                continue
            got, exp = o.arithmetic(op, shape, a, b), ref.arithmetic(op, shape, a, b)
            assert got.tobytes() == exp.tobytes(), (dtype, op, shape, n)


@pytest.mark.parametrize("dtype", OL.ALL_DTYPES, ids=str)
@pytest.mark.parametrize("op", [4, 5])
def test_unary_matches_reference_avx2(o, dtype, op):
    rng = np.random.default_rng(op)
    for n in [1, 3, 17, 64, 1000]:
        a = rand(rng, dtype, n)
        assert o.arithmetic_unary(op, a).tobytes() == ref.arithmetic_unary(op, a).tobytes(), (dtype, op, n)


@pytest.mark.parametrize("dtype", OL.ALL_DTYPES, ids=str)
@pytest.mark.parametrize("cmpop", [0, 1, 2, 3])
def test_compare_matches_reference_avx2(o, dtype, cmpop):
    rng = np.random.default_rng(cmpop)
    dt = np.dtype(dtype)
    for n in [1, 5, 8, 31, 32, 33, 64, 100, 1000]:
        # small value range so equality actually happens
        l = rng.integers(0, 4, n).astype(dt); r = rng.integers(0, 4, n).astype(dt)
        if dt.kind == "f" and n > 4:
            l[1] = np.nan; r[2] = np.nan
        s = np.array([2], dtype=dt)
        for shape, a, b in [(0, l, r), (1, l, s), (2, s, r)]:
            for offset in [0, 1, 3, 7]:
                for fill in (0x00, 0xFF):
                    nb = (offset + n + 7) // 8 + 1
                    got = o.comparison(cmpop, shape, a, b, np.full(nb, fill, np.uint8), offset)
                    exp = ref.comparison(cmpop, shape, a, b, np.full(nb, fill, np.uint8), offset)
                    assert got.tobytes() == exp.tobytes(), (dtype, cmpop, shape, n, offset, fill)


@pytest.mark.parametrize("op", [0, 1, 2, 3])
def test_bitmap_aligned_matches_reference_avx2(o, op):
    rng = np.random.default_rng(op)
    for nbytes in [1, 7, 31, 32, 33, 100, 4096]:
        l = rng.integers(0, 256, nbytes, dtype=np.uint8); r = rng.integers(0, 256, nbytes, dtype=np.uint8)
        exp = ref.bitmap_aligned(op, l, r)
        got = o.bitmap_op(op, l, 0, r, 0, np.zeros(nbytes, np.uint8), 0, nbytes * 8)
        assert got.tobytes() == exp.tobytes()


@pytest.mark.parametrize("frm", OL.ALL_DTYPES, ids=str)
@pytest.mark.parametrize("to", OL.ALL_DTYPES, ids=str)
def test_cast_matches_reference_avx2(o, frm, to):
    """cast_type_numeric_avx2 is static_cast per element.  Integer inputs and float → float: every
    value, bit for bit.  Float → int: only values the target can hold — beyond that static_cast is
    undefined and the reference's own machine code disagrees with itself (see the next test)."""
    if frm == to:
        pytest.skip("identity")
    rng = np.random.default_rng(OL.TYPE_IDS[np.dtype(frm)] * 16 + OL.TYPE_IDS[np.dtype(to)])
    fd, td = np.dtype(frm), np.dtype(to)
    for n in [0, 1, 3, 15, 16, 17, 33, 64, 1000]:
        if fd.kind == "f" and td.kind != "f":
            info = np.iinfo(td)
            lo, hi = max(float(info.min), -2.0**31 + 1024), min(float(info.max), 2.0**31 - 1024)
            a = rng.uniform(lo, hi, n).astype(fd)
            a = np.clip(a, fd.type(lo), fd.type(hi))
        else:
            a = rand(rng, frm, n)
        st, got, _ = o.cast_numeric(a, to, None, 0, True, True)
        assert st == 0
        exp = ref.cast_numeric(a, to)
        if td.kind == "f":
            nan = np.isnan(exp)
            assert np.array_equal(np.isnan(got), nan)
            assert got[~nan].tobytes() == exp[~nan].tobytes(), (frm, to, n)
        else:
            assert got.tobytes() == exp.tobytes(), (frm, to, n)


def test_reference_float_to_narrow_int_out_of_range_is_inconsistent():
    """Why out-of-range float → int is not pinned: the reference's AVX2 kernel SATURATES in its
    vector body (packus/packss) and WRAPS in its scalar tail — same call, same value, two answers."""
    a = np.full(40, 300.0, np.float32)
    out = ref.cast_numeric(a, np.uint8)
    assert set(out.tolist()) == {255, 44}, out


@pytest.mark.parametrize("dtype", OL.INT_DTYPES, ids=str)
def test_min_max_matches_reference_avx2(o, dtype):
    rng = np.random.default_rng(77 + np.dtype(dtype).itemsize)
    info = np.iinfo(dtype)
    for n in [1, 3, 15, 16, 17, 31, 33, 64, 1000, 100003]:
        a = rng.integers(info.min, info.max, n, dtype=dtype, endpoint=True)
        assert o.min_max(a) == ref.min_max(a), (dtype, n)


def test_oracle_wide_slot_take_vs_arrow_cpp():
    """FSBImpl's take (kernels/vector_selection.go:1997-2031) as the oracle restates it — orc_take_primitive with slots of 16 and 32
    bytes — against Arrow C++'s take of binary(16) / binary(32) / decimal128 / decimal256 columns (the independent semantic
    cross-check of SURVEY §8c: values and validity; a null slot's payload is the fresh buffer's zero in both).  CPU only: this pins
    the restatement the GPU's 16- / 32-byte Take is compared with (tests/test_gpu_parity.py::test_take_wide_slots_bit_exact)."""
    pa = pytest.importorskip("pyarrow")
    import pyarrow.compute as pc
    o = OL.load_oracle()
    rng = np.random.default_rng(97)
    for w in (16, 32, 3, 24):      # (3: the reference's own test column, vector_selection_test.go:1170)
        n, m = 5000, 7001
        raw = rng.integers(0, 256, (n, w), dtype=np.uint8)
        vmask = rng.random(n) < 0.1
        col = pa.array([None if z else bytes(r) for r, z in zip(raw, vmask)], type=pa.binary(w))
        vvalid = np.packbits(~vmask, bitorder="little")
        for ityp, npt in ((pa.int8(), np.int8), (pa.uint16(), np.uint16), (pa.int32(), np.int32), (pa.uint64(), np.uint64)):
            hi = min(n, np.iinfo(npt).max + 1)
            idx = rng.integers(0, hi, m).astype(npt)
            imask = rng.random(m) < 0.2
            want = pc.take(col, pa.array(idx, type=ityp, mask=imask))
            ivalid = np.packbits(~imask, bitorder="little")
            st, out, ov, nulls, _ = o.take_primitive(raw.view(np.dtype(f"V{w}")).reshape(-1), vvalid, 0, idx, ivalid, 0, True, True)
            assert st == 0 and nulls == want.null_count
            got_valid = np.unpackbits(ov, bitorder="little")[:m].astype(bool)
            assert np.array_equal(got_valid, np.array(want.is_valid()))
            got = out.view(np.uint8).reshape(m, w)
            for i in np.flatnonzero(got_valid)[:2000]:
                assert bytes(got[i]) == want[int(i)].as_py()
            assert not got[~got_valid].any()      # null slots: zero payload
    # the same bytes read as decimals: Arrow C++'s take of a decimal128 column == the 16-byte slots moved by the oracle
    import decimal
    vals = [decimal.Decimal(int(v)) / 1000 for v in rng.integers(-10**15, 10**15, 300)]
    dcol = pa.array(vals, type=pa.decimal128(20, 3))
    idx = rng.integers(0, 300, 1000).astype(np.int32)
    want = pc.take(dcol, pa.array(idx))
    slots = np.frombuffer(dcol.buffers()[1], dtype=np.dtype("V16"), count=300)
    st, out, _, _, _ = o.take_primitive(slots, None, 0, idx, None, 0, True, False)
    assert st == 0 and out.tobytes() == bytes(want.buffers()[1])[:1000 * 16]


# the reference's own table for Filter on Decimal128 / Decimal256 (precision 3, scale 2): compute/vector_selection_test.go:656-671
# (FilterKernelWithDecimal.TestFilterDecimalNumeric) — values, mask, expected with DROP … and what EMIT_NULL makes of the same rows
DECIMAL_FILTER_TABLE = [
    ([], [], []),
    (["9.00"], [False], []), (["9.00"], [True], ["9.00"]), (["9.00"], [None], [None]),
    ([None], [False], []), ([None], [True], [None]), ([None], [None], [None]),
    (["7.12", "8.00", "9.87"], [False, True, False], ["8.00"]),
    (["7.12", "8.00", "9.87"], [True, False, True], ["7.12", "9.87"]),
    ([None, "8.00", "9.87"], [False, True, False], ["8.00"]),
    (["7.12", "8.00", "9.87"], [None, True, False], [None, "8.00"]),
    (["7.12", "8.00", "9.87"], [True, None, True], ["7.12", None, "9.87"]),
]


def decimal_slots(vals, width):
    """["7.12", None, …] at scale 2 → (little-endian two's-complement slots of `width` bytes, validity bits or None)"""
    raw = np.zeros((len(vals), width), np.uint8)
    for i, v in enumerate(vals):
        if v is not None:
            raw[i] = np.frombuffer(int(round(float(v) * 100)).to_bytes(width, "little", signed=True), np.uint8)
    valid = OL.pack_bits([v is not None for v in vals]) if any(v is None for v in vals) else None
    return raw.view(np.dtype(f"V{width}")).reshape(-1), valid


@pytest.mark.parametrize("width", [16, 32], ids=["decimal128", "decimal256"])
def test_oracle_decimal_filter_table(width):
    """The reference's TestFilterDecimalNumeric rows through the oracle's composition for wide slots — GetTakeIndices
    (orc_filter_to_indices) + FSBImpl's take (orc_take_primitive, 16- / 32-byte slots) — which is what the device's array_filter of
    Decimal128 / Decimal256 runs (host/kernels.cc ExecFilterFixed).  The table's expectations are EMIT_NULL's for a null mask slot
    (`[null]` stays) and identical under DROP when the mask has no nulls; DROP on a null mask slot removes the row."""
    o = OL.load_oracle()
    for vals, mask, want in DECIMAL_FILTER_TABLE:
        slots, vvalid = decimal_slots(vals, width)
        fdata = OL.pack_bits([bool(m) for m in mask])
        fvalid = OL.pack_bits([m is not None for m in mask]) if any(m is None for m in mask) else None
        for null_sel, expect in ((1, want), (0, [w for w, m in zip(want, [m for m in mask if m is None or m]) if m is not None])):
            n = len(vals)
            if n == 0:
                assert expect == []
                continue
            idx, iv, _ = o.filter_to_indices(fdata, fvalid, 0, n, null_sel, True)
            st, out, ov, nulls, _ = o.take_primitive(slots, vvalid, 0, idx, iv if fvalid is not None and null_sel == 1 else None, 0, True, True)
            assert st == 0 and len(idx) == len(expect), (vals, mask, null_sel)
            got_valid = np.unpackbits(ov, bitorder="little")[:len(idx)].astype(bool)
            exp_slots, _ = decimal_slots(expect, width)
            assert list(got_valid) == [e is not None for e in expect], (vals, mask, null_sel)
            assert out[:len(idx)][got_valid].tobytes() == exp_slots[got_valid].tobytes(), (vals, mask, null_sel)
