"""Timestamp / Date / Time / Duration columns through CallFunction.

The reference registers its integer kernels for these types under temporal input matchers — selection
(kernels/vector_selection.go:1845-1870), hashing (vector_hash.go:545-560), comparisons
(scalar_comparisons.go:640-690), add / subtract (arithmetic.go:630-770) — so the checks here are the TYPE RULES
(which pairs meet, what the result is labelled) and that the integers underneath come out bit for bit.  Values
are compared with Arrow C++ on the same arrays; the subtraction vectors are the ones of
ScalarBinaryTemporalArithmeticSuite (arrow/compute/arithmetic_test.go:2234-2283)."""
import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pyarrow.ipc
import pytest

from arrow_go_amd import compute as ac

TYPES = [pa.timestamp("s"), pa.timestamp("ms", "UTC"), pa.timestamp("us", "America/New_York"), pa.timestamp("ns"),
         pa.duration("s"), pa.duration("ms"), pa.duration("us"), pa.duration("ns"),
         pa.date32(), pa.date64(), pa.time32("s"), pa.time32("ms"), pa.time64("us"), pa.time64("ns")]


# ---- host only -------------------------------------------------------------------------------------------------
def test_formats_round_trip():
    for t in TYPES:
        f = ac._temporal_format(t)
        assert f and f[0] == "t" and ac._temporal_type(f) == t
        # the format is the one the C Data Interface uses for the type
        assert pa.array([], type=t).type == ac._temporal_type(f)
    assert ac._temporal_format(pa.int64()) is None and ac._temporal_format(pa.string()) is None
    assert ac._temporal_format(pa.timestamp("us", "UTC")) == "tsu:UTC" and ac._temporal_format(pa.date32()) == "tdD"


def test_ipc_inspect_names_temporal_fields():
    t = pa.table({"ts": pa.array([1, 2, None], pa.timestamp("us", "UTC")), "d": pa.array([1, 2, 3], pa.date32()),
                  "d64": pa.array([86400000, 0, None], pa.date64()), "t32": pa.array([1, 2, 3], pa.time32("ms")),
                  "t": pa.array([1, 2, 3], pa.time64("ns")), "du": pa.array([1, 2, 3], pa.duration("s")), "plain": pa.array([1, 2, 3], pa.int64())})
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, t.schema) as w:
        w.write_table(t)
    fields, rows = ac.ipc_inspect(sink.getvalue())
    assert fields == [(f.name, str(f.type), True) for f in t.schema] and rows == [3]


# ---- device ----------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sess():
    s = ac.Session(0)
    yield s
    s.close()


def rand(rng, typ, n, p_null=0.15, lo=0, hi=80000):
    """values that are valid for every temporal type (times of day stay below one day in seconds)"""
    vals = rng.integers(lo, hi, n)
    if pa.types.is_date64(typ):
        vals = vals * 86400000
    store = pa.int32() if typ.bit_width == 32 else pa.int64()
    return pa.array(vals, mask=rng.random(n) < p_null, type=store).cast(typ)


def storage(a):
    return a.cast(pa.int32() if a.type.bit_width == 32 else pa.int64())


@pytest.mark.gpu
@pytest.mark.parametrize("typ", TYPES, ids=str)
def test_selection_hashing_sorting_keep_the_type(sess, typ):
    rng = np.random.default_rng(7)
    n = 5003
    a = rand(rng, typ, n, hi=300)           # few distinct values: unique / dictionary_encode have work to do
    rt = sess.call_function("array_take", [a, pa.array(np.arange(n), pa.int64())])
    assert rt.type == typ and rt.equals(a)  # import → kernel → export leaves type and values alone
    idx = pa.array(rng.integers(0, n, 777), mask=rng.random(777) < 0.1, type=pa.int32())
    assert sess.call_function("take", [a, idx]).equals(a.take(idx))
    mask = pa.array(rng.random(n) < 0.4, mask=rng.random(n) < 0.1)
    assert sess.call_function("filter", [a, mask]).equals(a.filter(mask))
    assert sess.call_function("filter", [a, mask], "null_selection_behavior=emit_null").equals(a.filter(mask, null_selection_behavior="emit_null"))
    u = sess.call_function("unique", [a])
    assert u.type == typ and u.equals(a.unique())
    d = sess.call_function("dictionary_encode", [a])
    assert d.type == pa.dictionary(pa.int32(), typ) and d.equals(a.dictionary_encode())
    si = sess.call_function("sort_indices", [a], "order=descending")
    assert si.type == pa.uint64() and len(si) == n
    assert storage(a).take(si).equals(storage(a).take(pc.array_sort_indices(a, order="descending")))
    so = sess.call_function("sort", [a], "order=ascending")
    assert so.type == typ and so.equals(a.take(pc.array_sort_indices(a, order="ascending")))
    assert sess.call_function("is_null", [a]).equals(a.is_null())
    assert sess.call_function("is_not_null", [a]).equals(a.is_valid())


@pytest.mark.gpu
@pytest.mark.parametrize("typ", TYPES, ids=str)
def test_comparisons_need_equal_types(sess, typ):
    rng = np.random.default_rng(11)
    n = 4099
    a, b = rand(rng, typ, n, hi=50), rand(rng, typ, n, hi=50)
    for name, ref in [("equal", pc.equal), ("not_equal", pc.not_equal), ("less", pc.less), ("less_equal", pc.less_equal),
                      ("greater", pc.greater), ("greater_equal", pc.greater_equal)]:
        assert sess.call_function(name, [a, b]).equals(ref(a, b)), name
    s = pa.scalar(25, storage(a).type).cast(typ)
    assert sess.call_function("greater", [a, s]).equals(pc.greater(a, s))
    assert sess.call_function("less_equal", [s, a]).equals(pc.less_equal(s, a))
    assert sess.call_function("equal", [a, pa.scalar(None, typ)]).null_count == n
    # a bare integer is not a point in time
    with pytest.raises(ac.ErrNotImplemented, match="no kernel matching input types"):
        sess.call_function("equal", [a, storage(b)])
    vs = rand(rng, typ, 40, p_null=0, hi=50)
    assert sess.call_function("is_in", [a], value_set=vs).equals(pc.is_in(a, value_set=vs))
    with pytest.raises(ac.ErrType, match="value set"):
        sess.call_function("is_in", [a], value_set=storage(vs))


@pytest.mark.gpu
def test_timestamp_units_and_zones(sess):
    a = pa.array([1, 5, None, 7], pa.timestamp("us", "UTC"))
    # exec.TimestampTypeUnit matches on the unit alone: other zones, and naive timestamps, meet a zoned one
    for other in [pa.array([1, 4, 3, 9], pa.timestamp("us", "Europe/Paris")), pa.array([1, 4, 3, 9], pa.timestamp("us"))]:
        assert sess.call_function("less", [a, other]).to_pylist() == [False, False, None, True]
        d = sess.call_function("subtract", [a, other])
        assert d.type == pa.duration("us") and d.cast(pa.int64()).to_pylist() == [0, 1, None, -2]
    # different units: DispatchBest takes both sides to the finer one (utils.go:130-170, :329-399)
    ms = pa.array([0, 1, 3, -9], pa.timestamp("ms", "UTC"))
    assert sess.call_function("greater", [a, ms]).to_pylist() == [True, False, None, True]
    assert sess.call_function("greater", [ms, a]).to_pylist() == [False, True, None, False]
    d = sess.call_function("subtract", [ms, a])
    assert d.type == pa.duration("us") and d.cast(pa.int64()).to_pylist() == [-1, 995, None, -9007]
    got = sess.call_function("add", [pa.array([10, None], pa.timestamp("s")), pa.array([5, 5], pa.duration("ns"))])
    assert got.type == pa.timestamp("ns") and got.cast(pa.int64()).to_pylist() == [10_000_000_005, None]
    got = sess.call_function("subtract", [pa.array([10, 20], pa.time32("s")), pa.array([5, 7], pa.time64("us"))])
    assert got.type == pa.duration("us") and got.cast(pa.int64()).to_pylist() == [9_999_995, 19_999_993]
    assert sess.call_function("equal", [pa.array([1, 2, None], pa.date32()), pa.array([86400000, 5, 7], pa.date64())]).to_pylist() == [True, False, None]
    assert sess.call_function("less", [pa.array([1, 2000], pa.duration("s")), pa.scalar(1500, pa.int64()).cast(pa.duration("ms"))]).to_pylist() == [True, False]
    # the implicit cast is the safe one: a second count that has no nanosecond representation stops the call
    with pytest.raises(ac.ErrInvalid, match=r"casting from timestamp\[s\] to timestamp\[ns\] would result in out of bounds timestamp: 32503680000"):
        sess.call_function("less", [pa.array([1, 32503680000], pa.timestamp("s")), pa.array([1, 2], pa.timestamp("ns"))])
    # no common type: a point in time and a span, a date and a timestamp
    for args in [[a, pa.array([1, 2, 3, 4], pa.duration("us"))], [a, pa.array([1, 2, 3, 4], pa.date32())]]:
        with pytest.raises(ac.ErrNotImplemented, match="no kernel matching input types"):
            sess.call_function("equal", args)


# arrow/compute/arithmetic_test.go:2234-2268
DATE32 = ([0, 11016, -25932, 23148, 18262, 18261, 18260, 14609, 14610, 14612, 14613, 13149, 13148, 14241, 14242, 15340, None],
          [365, 10650, -25901, 23118, 18263, 18259, 18260, 14609, 14610, 14612, 14613, 13149, 13148, 14240, 13937, 15400, None])
DATE64 = ([0, 951782400000, -2240524800000, 1999987200000, 1577836800000, 1577750400000, 1577664000000, 1262217600000, 1262304000000,
           1262476800000, 1262563200000, 1136073600000, 1135987200000, 1230422400000, 1230508800000, 1325376000000, None],
          [31536000000, 920160000000, -2237846400000, 1997395200000, 1577923200000, 1577577600000, 1577664000000, 1262217600000,
           1262304000000, 1262476800000, 1262563200000, 1136073600000, 1135987200000, 1230336000000, 1204156800000, 1330560000000, None])
TIME_S = ([59, 84203, 3560, 12800, 3905, 7810, 11715, 15620, 19525, 23430, 27335, 31240, 35145, 0, 0, 3723, None],
          [59, 84203, 12642, 7182, 68705, 7390, 915, 16820, 19525, 5430, 84959, 31207, 35145, 0, 0, 3723, None])
TIME_MS = ([59123, 84203999, 3560001, 12800000, 3905001, 7810002, 11715003, 15620004, 19525005, 23430006, 27335000, 31240000, 35145000, 0, 0, 3723000, None],
           [59103, 84203999, 12642001, 7182000, 68705005, 7390000, 915003, 16820004, 19525005, 5430006, 84959000, 31207000, 35145000, 0, 0, 3723000, None])
TIME_US = ([59123456, 84203999999, 3560001001, 12800000000, 3905001000, 7810002000, 11715003000, 15620004132, 19525005321, 23430006163,
            27335000000, 31240000000, 35145000000, 0, 0, 3723000000, None],
           [59103476, 84203999999, 12642001001, 7182000000, 68705005000, 7390000000, 915003000, 16820004432, 19525005021, 5430006163,
            84959000000, 31207000000, 35145000000, 0, 0, 3723000000, None])
TIME_NS = ([59123456789, 84203999999999, 3560001001001, 12800000000000, 3905001000000, 7810002000000, 11715003000000, 15620004132000,
            19525005321000, 23430006163000, 27335000000000, 31240000000000, 35145000000000, 0, 0, 3723000000000, None],
           [59103476799, 84203999999909, 12642001001001, 7182000000000, 68705005000000, 7390000000000, 915003000000, 16820004432000,
            19525005021000, 5430006163000, 84959000000000, 31207000000000, 35145000000000, 0, 0, 3723000000000, None])


@pytest.mark.gpu
@pytest.mark.parametrize("vals,typ,out,scale", [(DATE32, pa.date32(), pa.duration("s"), 86400), (DATE64, pa.date64(), pa.duration("ms"), 1),
                                                (TIME_S, pa.time32("s"), pa.duration("s"), 1), (TIME_MS, pa.time32("ms"), pa.duration("ms"), 1),
                                                (TIME_US, pa.time64("us"), pa.duration("us"), 1), (TIME_NS, pa.time64("ns"), pa.duration("ns"), 1)],
                         ids=["date32", "date64", "time32s", "time32ms", "time64us", "time64ns"])
def test_reference_subtraction_vectors(sess, vals, typ, out, scale):
    """TestTemporalAddSub: the difference of two dates / times of day is a duration of the type's unit"""
    store = pa.int32() if typ.bit_width == 32 else pa.int64()
    a, b = pa.array(vals[0], store).cast(typ), pa.array(vals[1], store).cast(typ)
    want = [None if x is None else (x - y) * scale for x, y in zip(*vals)]
    for fn in ["subtract", "subtract_unchecked"]:
        got = sess.call_function(fn, [a, b])
        assert got.type == out, fn
        assert got.cast(pa.int64()).to_pylist() == want, fn


@pytest.mark.gpu
def test_date32_difference_overflow_rules(sess):
    # SubtractDate32 (base_arithmetic.go:702-720): unchecked multiplies the int32 difference by 86400 in int32 and wraps,
    # checked does both steps in int64
    a, b = pa.array([30000, 0, None], pa.date32()), pa.array([0, 30000, 1], pa.date32())
    wrap32 = lambda v: int(np.array([v], dtype=np.int64).astype(np.int32)[0])
    assert wrap32(30000 * 86400) != 30000 * 86400
    assert sess.call_function("subtract_unchecked", [a, b]).cast(pa.int64()).to_pylist() == [wrap32(30000 * 86400), wrap32(-30000 * 86400), None]
    assert sess.call_function("subtract", [a, b]).cast(pa.int64()).to_pylist() == [30000 * 86400, -30000 * 86400, None]


@pytest.mark.gpu
@pytest.mark.parametrize("typ,unit,day", [(pa.time32("s"), "s", 86400), (pa.time32("ms"), "ms", 86400 * 10**3),
                                          (pa.time64("us"), "us", 86400 * 10**6), (pa.time64("ns"), "ns", 86400 * 10**9)], ids=str)
def test_time_plus_minus_duration(sess, typ, unit, day):
    """GetArithmeticFunctionTimeDuration (scalar_arithmetic.go:47-65) / timeDurationOp (base_arithmetic.go:642-700): time ± duration of
    the same unit → the time type; every result must lie in [0, one day), the error names the last offender"""
    rng = np.random.default_rng(13)
    n = 3001
    store = pa.int32() if pa.types.is_time32(typ) else pa.int64()
    t0 = rng.integers(0, day // 2, n)
    d0 = rng.integers(0, day // 2, n)
    t = pa.array(t0, mask=rng.random(n) < 0.1, type=store).cast(typ)
    d = pa.array(d0, mask=rng.random(n) < 0.1, type=pa.int64()).cast(pa.duration(unit))
    for name in ("add", "add_unchecked"):
        got = sess.call_function(name, [t, d])
        assert got.type == typ
        want = pc.add(t.cast(store).cast(pa.int64()), d.cast(pa.int64())).cast(store).cast(typ)
        assert got.equals(want), name
    big = pa.array(t0 + day // 2, mask=rng.random(n) < 0.1, type=store).cast(typ)
    for name in ("subtract", "subtract_unchecked"):
        got = sess.call_function(name, [big, d])
        want = pc.subtract(big.cast(store).cast(pa.int64()), d.cast(pa.int64())).cast(store).cast(typ)
        assert got.type == typ and got.equals(want), name
    # arr ∘ scalar
    one = pa.scalar(5, pa.int64()).cast(pa.duration(unit))
    assert sess.call_function("add", [t, one]).equals(pc.add(t.cast(store).cast(pa.int64()), 5).cast(store).cast(typ))
    # out of [0, day): the last offending value is reported (the Go loop keeps overwriting its error)
    late = pa.array([10, day - 1, day - 2, 7], type=store).cast(typ)
    step = pa.array([1, 1, 5, 1], type=pa.int64()).cast(pa.duration(unit))
    with pytest.raises(ac.ErrInvalid, match=r"%d is not within acceptable range of \[0, %d\) s" % (day + 3, day)):
        sess.call_function("add", [late, step])
    with pytest.raises(ac.ErrInvalid, match=r"-4 is not within acceptable range"):
        sess.call_function("subtract_unchecked", [pa.array([1, 50], type=store).cast(typ), pa.array([5, 1], type=pa.int64()).cast(pa.duration(unit))])
    # a duration of another unit is refused unless it can be brought to the time's family unit
    with pytest.raises((ac.ErrNotImplemented, ac.ErrInvalid)):
        sess.call_function("multiply", [t, d])


@pytest.mark.gpu
@pytest.mark.parametrize("unit", ["s", "ms", "us", "ns"])
def test_timestamp_duration_arithmetic(sess, unit):
    rng = np.random.default_rng(3)
    n = 3001
    ts, ts2 = rand(rng, pa.timestamp(unit, "UTC"), n, lo=-10**15, hi=10**15), rand(rng, pa.timestamp(unit, "UTC"), n, lo=-10**15, hi=10**15)
    du, du2 = rand(rng, pa.duration(unit), n, lo=-10**12, hi=10**12), rand(rng, pa.duration(unit), n, lo=-10**12, hi=10**12)
    for suffix, padd, psub in [("", pc.add_checked, pc.subtract_checked), ("_unchecked", pc.add, pc.subtract)]:
        for fn, args, ref in [("add", [ts, du], padd), ("add", [du, ts], padd), ("add", [du, du2], padd),
                              ("subtract", [ts, ts2], psub), ("subtract", [ts, du], psub), ("subtract", [du, du2], psub)]:
            got, want = sess.call_function(fn + suffix, args), ref(*args)
            assert got.type == want.type and got.equals(want), (fn + suffix, [str(x.type) for x in args])
    # arr ∘ scalar keeps the rule
    one = pa.scalar(1000, pa.int64()).cast(pa.duration(unit))
    assert sess.call_function("add", [ts, one]).equals(pc.add_checked(ts, one))
    # the checked form reports int64 overflow in a valid slot, the unchecked one wraps
    edge = pa.array([2**63 - 1, 5, None], pa.int64()).cast(pa.timestamp(unit, "UTC"))
    step = pa.array([1, 1, 1], pa.int64()).cast(pa.duration(unit))
    with pytest.raises(ac.ErrInvalid, match="overflow"):
        sess.call_function("add", [edge, step])
    assert sess.call_function("add_unchecked", [edge, step]).cast(pa.int64()).to_pylist() == [-2**63, 6, None]
    # pairs the reference has no kernel for (arithmetic.go:630-770)
    # a duration of another unit: both sides go to the finer one first
    other = {"s": "ms", "ms": "us", "us": "ns", "ns": "s"}[unit]
    finer = "ns" if "ns" in (unit, other) else other
    small_ts, du_o = rand(rng, pa.timestamp(unit, "UTC"), n), rand(rng, pa.duration(other), n)
    got = sess.call_function("add", [small_ts, du_o])
    assert got.equals(pc.add_checked(small_ts.cast(pa.timestamp(finer, "UTC")), du_o.cast(pa.duration(finer))))
    for fn, args in [("add", [ts, ts2]), ("multiply", [du, du2]),
                     ("subtract", [du, ts]), ("add", [ts, storage(du)]), ("negate", [du]), ("cumulative_sum", [du])]:
        with pytest.raises(ac.ErrNotImplemented, match="no kernel matching input types"):
            sess.call_function(fn, args)


@pytest.mark.gpu
def test_casts_between_a_temporal_type_and_its_storage(sess):
    ts = pa.array([1, None, 3], pa.timestamp("us", "UTC"))
    as_int = sess.call_function("cast", [ts], "to_type=int64")
    assert as_int.type == pa.int64() and as_int.to_pylist() == [1, None, 3]
    back = sess.call_function("cast", [as_int], "to_logical=tsu:UTC")
    assert back.equals(ts)
    assert sess.call_function("cast", [ts], "to_logical=tsu:UTC").equals(ts)
    d = sess.call_function("cast", [pa.array([1, 2], pa.int32())], "to_logical=tdD")
    assert d.equals(pa.array([1, 2], pa.date32()))
    for args, opts in [([ts], "to_type=int32"), ([pa.array([1], pa.int32())], "to_logical=tsu:"), ([pa.array([1], pa.int64())], "to_logical=txx"),
                       ([ts], "to_logical=tDu"), ([ts], "to_logical=tdD")]:
        with pytest.raises(ac.ErrNotImplemented):
            sess.call_function("cast", args, opts)


# arrow/compute/cast_test.go: TestTimestampToTimestamp :2650-2693, TestTimeToTime :2964-3038, TestDurationToDuration :3083-3157
_UNIT_CASTS = [(pa.timestamp("s"), pa.timestamp("ms"), 10**3), (pa.timestamp("ms"), pa.timestamp("us"), 10**3), (pa.timestamp("us"), pa.timestamp("ns"), 10**3),
               (pa.timestamp("s"), pa.timestamp("ns"), 10**9), (pa.duration("s"), pa.duration("ms"), 10**3), (pa.duration("ms"), pa.duration("us"), 10**3),
               (pa.duration("us"), pa.duration("ns"), 10**3), (pa.duration("s"), pa.duration("ns"), 10**9),
               (pa.time32("s"), pa.time32("ms"), 10**3), (pa.time32("ms"), pa.time64("us"), 10**3), (pa.time64("us"), pa.time64("ns"), 10**3),
               (pa.time32("s"), pa.time64("us"), 10**6), (pa.time32("ms"), pa.time64("ns"), 10**6), (pa.time32("s"), pa.time64("ns"), 10**9)]


@pytest.mark.gpu
@pytest.mark.parametrize("coarse,fine,factor", _UNIT_CASTS, ids=lambda v: str(v))
def test_unit_casts_reference_vectors(sess, coarse, fine, factor):
    store = lambda t: pa.int32() if t.bit_width == 32 else pa.int64()
    arr = lambda vals, t: pa.array(vals, store(t)).cast(t)
    to = lambda t: "to_logical=" + ac._temporal_format(t)
    got = sess.call_function("cast", [arr([0, None, 200, 1, 2], coarse)], to(fine))
    assert got.type == fine and got.equals(arr([0, None, 200 * factor, factor, 2 * factor], fine))
    k = factor // 1000
    lossy = arr([0, None, 200 * factor + 456 * k, factor + 123 * k, 2 * factor + 456 * k], fine)
    with pytest.raises(ac.ErrInvalid, match="would lose data: %d" % (200 * factor + 456 * k)):
        sess.call_function("cast", [lossy], to(coarse))
    got = sess.call_function("cast", [lossy], to(coarse) + ";allow_time_truncate=1")
    assert got.type == coarse and got.equals(arr([0, None, 200, 1, 2], coarse))
    # a slice keeps its offset; chunked columns go chunk by chunk
    big = arr(list(range(1000)), coarse)
    assert sess.call_function("cast", [big.slice(13, 900)], to(fine)).equals(arr([v * factor for v in range(13, 913)], fine))
    ch = sess.call_function("cast", [pa.chunked_array([big.slice(0, 10), big.slice(10, 990)])], to(fine))
    assert isinstance(ch, pa.ChunkedArray) and ch.type == fine and ch.combine_chunks().equals(arr([v * factor for v in range(1000)], fine))


@pytest.mark.gpu
def test_unit_cast_overflow_and_dates(sess):
    # TestTimestampToTimestampMultiplyOverflow :2703-2707, TestDurationToDurationMultiplyOverflow :3166-3169, TestDateToDate :3052-3069
    far = pa.array([-30610224000, -5364662400, 946684800, 10413792000, 32503680000], pa.timestamp("s"))
    with pytest.raises(ac.ErrInvalid, match=r"casting from timestamp\[s\] to timestamp\[ns\] would result in out of bounds timestamp: -30610224000"):
        sess.call_function("cast", [far], "to_logical=tsn:")
    wrapped = sess.call_function("cast", [far], "to_logical=tsn:;allow_time_overflow=1")
    assert wrapped.cast(pa.int64()).to_pylist() == (np.array(far.cast(pa.int64())).astype(np.uint64) * np.uint64(10**9)).astype(np.int64).tolist()
    with pytest.raises(ac.ErrInvalid, match=r"casting from duration\[s\] to duration\[ns\] would result in out of bounds timestamp: 10000000000"):
        sess.call_function("cast", [pa.array([10000000000, 1, 2, 3, 10000000000], pa.duration("s"))], "to_logical=tDn")
    d32 = pa.array([0, None, 100, 1, 10], pa.date32())
    d64 = sess.call_function("cast", [d32], "to_logical=tdm")
    assert d64.equals(pa.array([0, None, 8640000000, 86400000, 864000000], pa.date64()))
    assert sess.call_function("cast", [d64], "to_logical=tdD").equals(d32)
    lossy = pa.array([0, None, 8640000123, 86400456, 864000789], pa.int64())
    lossy = sess.call_function("cast", [lossy], "to_logical=tdm", keep_on_device=True)   # pyarrow would refuse to build this date64
    with pytest.raises(ac.ErrInvalid, match="casting from date64 to date32 would lose data: 8640000123"):
        sess.call_function("cast", [lossy], "to_logical=tdD")
    assert sess.call_function("cast", [lossy], "to_logical=tdD;allow_time_truncate=1").equals(d32)
    # zones are labels: a cast between them leaves the instants alone (TestTimestampToTimestampSimpleTimezone :2643-2648)
    z = pa.array([1672601100123456, None], pa.timestamp("us", "Etc/UTC"))
    assert sess.call_function("cast", [z], "to_logical=tsu:").equals(z.cast(pa.int64()).cast(pa.timestamp("us")))
    # scalars follow the same rule on the host
    assert sess.call_function("cast", [pa.scalar(7, pa.int64()).cast(pa.duration("s"))], "to_logical=tDm").value == 7000


@pytest.mark.gpu
def test_chunked_columns_record_batches_and_dictionaries(sess):
    rng = np.random.default_rng(5)
    n = 6000
    typ = pa.timestamp("ms", "UTC")
    a = rand(rng, typ, n, hi=400)
    ca = pa.chunked_array([a.slice(0, 1000), a.slice(1000, 0), a.slice(1000, 5000)], type=typ)
    idx = pa.array(rng.integers(0, n, 500), pa.int64())
    out = sess.call_function("take", [ca, idx])
    assert isinstance(out, pa.ChunkedArray) and out.type == typ and out.combine_chunks().equals(a.take(idx))
    u = sess.call_function("unique", [ca])
    assert u.type == typ and u.equals(a.unique())
    g = sess.call_function("greater", [ca, pa.scalar(200, pa.int64()).cast(typ)])
    assert g.combine_chunks().equals(pc.greater(a, pa.scalar(200, pa.int64()).cast(typ)))
    # a record batch: every column keeps its own type through filter / take / sort
    batch = pa.RecordBatch.from_arrays([a, rand(rng, pa.date32(), n), pa.array(rng.integers(0, 9, n), pa.int64()),
                                        rand(rng, pa.duration("ns"), n)], names=["when", "day", "k", "took"])
    mask = pa.array(rng.random(n) < 0.3)
    got = sess.call_function("filter", [batch, mask])
    assert got.schema == batch.schema and got.equals(batch.filter(mask))
    got = sess.call_function("take", [batch, idx])
    assert got.schema == batch.schema and got.equals(batch.take(idx))
    # dictionary-encoded temporal column: the values of the dictionary carry the type
    d = a.dictionary_encode()
    got = sess.call_function("take", [d, idx])
    assert got.type == d.type and got.equals(d.take(idx))
    assert sess.call_function("unique", [d]).type == d.type


@pytest.mark.gpu
def test_read_ipc_temporal_columns(sess):
    rng = np.random.default_rng(9)
    n = 2000
    cols = {str(i): rand(rng, t, n) for i, t in enumerate(TYPES)}
    cols["enc"] = rand(rng, pa.timestamp("s"), n, hi=30).dictionary_encode()
    t = pa.table(cols)
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, t.schema) as w:
        for b in t.to_batches(max_chunksize=700):
            w.write_batch(b)
    want = t.to_batches(max_chunksize=700)
    got = list(sess.read_ipc(sink.getvalue()))
    assert len(got) == len(want)
    for (names, arrays, rows), b in zip(got, want):
        assert names == b.schema.names and rows == b.num_rows
        for dev, col in zip(arrays, b.columns):
            back = dev.to_arrow()
            assert back.type == col.type and back.equals(col), str(col.type)
    # a column read from the stream goes straight into a kernel, type rules included
    names, arrays, rows = got[0]
    d = sess.call_function("subtract", [arrays[1], arrays[1]])
    assert d.type == pa.duration("ms") and set(d.cast(pa.int64()).to_pylist()) <= {0, None}


@pytest.mark.gpu
def test_expressions_over_temporal_columns_run_node_by_node(sess):
    rng = np.random.default_rng(13)
    n = 3000
    ts, du = rand(rng, pa.timestamp("us"), n), rand(rng, pa.duration("us"), n)
    cut = pa.scalar(60000, pa.int64()).cast(pa.timestamp("us"))
    got, fused = sess.eval_expression("greater(add($0,$1),#0)", [ts, du], [cut], fuse=True)
    assert not fused and got.equals(pc.greater(pc.add_checked(ts, du), cut))
    with pytest.raises(ac.ErrNotImplemented, match="no kernel matching input types"):
        sess.eval_expression("add($0,$0)", [ts, du], [], fuse=True)
