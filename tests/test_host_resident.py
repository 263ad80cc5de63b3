"""CallFunction over HOST-resident arguments: chunked, overlapped execution inside the host mirror (arrow_go_amd/host/hoststream.cc).

The reference's executor cuts a call into spans of at most ExecCtx.ChunkSize rows (arrow/compute/executor.go:47-50, :499
iterateExecSpans, :658-702).  Here a column imported with ahc_import_host stays in host memory and add / sub / multiply, the
comparisons, filter and arrow/math Sum stream it through the device span by span (upload k + 1 | kernel k | download k − 1); every
other function uploads it whole first.  Checked against Arrow C++ (pyarrow.compute) — logically — and against this library's own
whole-array path byte for byte (value bytes under nulls, validity bytes with their tail bits, null count: same_bytes); then the rate of a 1 GiB pinned column against the link's, and a call whose arguments do not fit the HBM
that is left."""
import numpy as np
import pytest

pa = pytest.importorskip("pyarrow")
import pyarrow.compute as pc  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    from arrow_go_amd import compute as ac
    s = ac.Session(0)
    s.set_option("host_threshold_bytes", 0)       # every flat column imported with import_host stays on the host
    s.set_option("chunk_bytes", 64 << 10)          # 8192 Int64 rows per span: dozens of spans on small columns
    yield s
    s.close()


def column(rng, n, typ, nulls):
    if pa.types.is_floating(typ):
        v = rng.uniform(-1e6, 1e6, n)
    else:
        v = rng.integers(-10**6, 10**6, n)
    mask = rng.random(n) < 0.1 if nulls else None
    return pa.array(v, type=typ, mask=mask)


def same(a, b):
    """logical equality with Arrow C++'s result"""
    assert a.type == b.type and len(a) == len(b), (a.type, b.type, len(a), len(b))
    assert a.null_count == b.null_count
    assert a.equals(b)


def raw(a):
    """(value bytes of the rows, validity bytes with the last byte's tail bits, null count) of an exported array"""
    assert a.offset == 0
    n = len(a)
    validity, data = a.buffers()[:2]
    width = a.type.bit_width
    nbytes = (n + 7) // 8 if width == 1 else n * (width // 8)
    return (data.to_pybytes()[:nbytes] if data is not None else b"",
            validity.to_pybytes()[:(n + 7) // 8] if validity is not None else None, a.null_count)


def same_bytes(streamed, whole):
    """the contract of hoststream.cc: the streamed result is the whole-array path's byte for byte — payload under nulls, the bits
    behind the last row, and whether a validity bitmap exists at all"""
    assert streamed.type == whole.type and len(streamed) == len(whole)
    sv, sb, sn = raw(streamed)
    wv, wb, wn = raw(whole)
    assert sn == wn, (sn, wn)
    assert (sb is None) == (wb is None), "one path allocated a validity bitmap, the other did not"
    assert sb == wb, "validity bytes differ"
    if sv != wv:
        w = max(streamed.type.bit_width // 8, 1)
        i = next(k for k in range(len(sv)) if sv[k] != wv[k]) // w
        raise AssertionError(f"value bytes differ first at row {i}: streamed {streamed[i:i + 1]} / whole-array {whole[i:i + 1]}")


def both(sess, name, host_args, plain_args, options=""):
    """(streamed result, whole-array result) of one call, as pyarrow arrays"""
    got = sess.call_function(name, host_args, options=options, keep_on_device=True)
    assert got.on_host(), f"{name}: not streamed"
    return got.to_arrow(), sess.call_function(name, plain_args, options=options)


@pytest.mark.parametrize("typ", [pa.int64(), pa.float64(), pa.int32(), pa.uint16()])
@pytest.mark.parametrize("name", ["add", "subtract", "multiply", "add_unchecked", "multiply_unchecked"])
def test_streamed_arithmetic(sess, name, typ):
    rng = np.random.default_rng(hash((name, str(typ))) % 2**31)
    n = 100_003
    pcf = {"add": pc.add_checked, "subtract": pc.subtract_checked, "multiply": pc.multiply_checked, "add_unchecked": pc.add, "multiply_unchecked": pc.multiply}[name]
    small = pa.types.is_integer(typ) and typ.bit_width <= 32
    for lnulls in (False, True):
        for rnulls in (False, True):
            a, b = column(rng, n, pa.int64(), lnulls), column(rng, n, pa.int64(), rnulls)
            if small:     # keep the products inside the narrow type (and a − b ≥ 0 for the unsigned one)
                a, b = pc.bit_wise_and(a, pa.scalar(63, pa.int64())), pc.bit_wise_and(b, pa.scalar(63, pa.int64()))
                if pa.types.is_unsigned_integer(typ):
                    a = pc.bit_wise_or(a, pa.scalar(64, pa.int64()))
            a, b = a.cast(typ), b.cast(typ)
            want = pcf(a, b)
            ha, hb = sess.import_host(a), sess.import_host(b)
            assert ha.on_host() and hb.on_host()
            got, whole = both(sess, name, [ha, hb], [a, b])         # streamed: the result is host-resident
            same(got, want)
            same(whole, want)
            same_bytes(got, whole)                                  # … and the whole-array path's bytes
            # array ∘ scalar and scalar ∘ array
            sc = pa.scalar(3, typ)
            got, whole = both(sess, name, [ha, sc], [a, sc])
            same(got, pcf(a, sc))
            same_bytes(got, whole)
            sc = pa.scalar(127, typ)
            got, whole = both(sess, name, [sc, hb], [sc, b])
            same(got, pcf(sc, b))
            same_bytes(got, whole)
    # a null scalar: every row null
    got = sess.call_function(name, [sess.import_host(a), pa.scalar(None, typ)])
    assert got.null_count == n


@pytest.mark.parametrize("typ", [pa.int64(), pa.float64(), pa.int16(), pa.float32(), pa.uint8()])
@pytest.mark.parametrize("name", ["add", "subtract", "multiply", "add_unchecked", "subtract_unchecked", "multiply_unchecked",
                                  "greater", "equal", "less", "not_equal"])
def test_null_scalar_payload_is_the_whole_array_paths(sess, name, typ):
    """ScalarBinaryNotNull (checked integer add / sub) leaves the output as allocated under a null scalar
    (kernels/helpers.go:311-314,340-343); ScalarBinary — every unchecked op, every float op, checked multiply
    (base_arithmetic.go:273-280) and the comparisons — unboxes the scalar's stored value and computes every row under the
    all-null validity (helpers.go:204-222).  Both paths of this library must hold the same bytes under those nulls."""
    rng = np.random.default_rng(zlib_seed(name, typ))
    n = 40_003                                                       # 5 spans of 8192 Int64 rows; not a multiple of 8
    a = column(rng, n, pa.int64(), True)
    a = pc.bit_wise_and(a, pa.scalar(15, pa.int64())).cast(typ)
    null = pa.scalar(None, typ)
    for args, plain in (([sess.import_host(a), null], [a, null]), ([null, sess.import_host(a)], [null, a])):
        got, whole = both(sess, name, args, plain)
        assert got.null_count == n and whole.null_count == n
        same_bytes(got, whole)
    arithmetic = name.split("_")[0] in ("add", "subtract", "multiply")
    if arithmetic and (name.endswith("_unchecked") or pa.types.is_floating(typ) or name == "multiply"):
        # the payload is l[i] ∘ 0 — not a block of zeros: a + 0 under the nulls
        got = sess.call_function(name, [sess.import_host(a), null], keep_on_device=True).to_arrow()
        vals = np.frombuffer(raw(got)[0], dtype=typ.to_pandas_dtype())
        src = np.frombuffer(raw(a)[0], dtype=typ.to_pandas_dtype())
        want = src * 0 if "multiply" in name else src
        assert np.array_equal(vals, want)


def zlib_seed(*parts):
    import zlib
    return zlib.crc32("/".join(str(p) for p in parts).encode())


def test_checked_add_with_nulls_reuses_slots_without_leaking(sess):
    """three slots serve 13 spans: under a null row the checked kernel must store the zero value (helpers.go:303-306), not what an
    earlier span left in the slot — large payloads early, nulls late, both operands, and the rows under nulls hold values that
    would overflow if they were tested"""
    rng = np.random.default_rng(77)
    n = 100_003
    big = np.iinfo(np.int64).max
    av = rng.integers(2**61, 2**62, n)
    bv = rng.integers(2**61, 2**62 - 1, n)
    am = rng.random(n) < np.linspace(0.0, 0.9, n)                     # ever more nulls towards the end
    bm = rng.random(n) < 0.2
    av[am] = big
    bv[bm] = big                                                      # big + anything overflows: must stay untested
    a, b = pa.array(av, mask=am), pa.array(bv, mask=bm)
    for name in ("add", "subtract"):
        got, whole = both(sess, name, [sess.import_host(a), sess.import_host(b)], [a, b])
        same_bytes(got, whole)
        vals = np.frombuffer(raw(got)[0], np.int64)
        assert not vals[am | bm].any(), "payload under a null row is not the zero value"
    got, whole = both(sess, "add", [sess.import_host(a), pa.scalar(1, pa.int64())], [a, pa.scalar(1, pa.int64())])
    same_bytes(got, whole)


@pytest.mark.parametrize("n", [8192 * 3, 8192 * 3 + 1, 8192 * 2 + 63, 8192 + 7, 5, 64, 65])
def test_last_spans_tail_bits(sess, n):
    """a comparison's data bitmap and the result's validity end inside a byte: the bits behind row n − 1 are the whole-array
    path's (zero), whatever the slot held before"""
    rng = np.random.default_rng(n)
    warm = pa.array(np.full(8192 * 3, -1, np.int64))                  # leaves all-ones bitmaps in every slot
    sess.call_function("equal", [sess.import_host(warm), sess.import_host(warm)], keep_on_device=True).release()
    a, b = column(rng, n, pa.int64(), True), column(rng, n, pa.int64(), True)
    for name in ("greater_equal", "not_equal", "less"):
        got, whole = both(sess, name, [sess.import_host(a), sess.import_host(b)], [a, b])
        same_bytes(got, whole)
        same(got, getattr(pc, name)(a, b))
    got, whole = both(sess, "add", [sess.import_host(a), sess.import_host(b)], [a, b])
    same_bytes(got, whole)


def test_streamed_checked_add_overflows(sess):
    from arrow_go_amd import compute as ac
    n = 50_000
    a = np.zeros(n, np.int64)
    a[n - 7] = np.iinfo(np.int64).max           # in the last span
    ha, hb = sess.import_host(pa.array(a)), sess.import_host(pa.array(np.ones(n, np.int64)))
    with pytest.raises(ac.ArrowError, match="overflow"):
        sess.call_function("add", [ha, hb])
    same(sess.call_function("add_unchecked", [ha, hb]), pc.add(pa.array(a), pa.array(np.ones(n, np.int64))))
    # … unless the offending row is null (checked kernels test valid slots only)
    mask = np.zeros(n, bool)
    mask[n - 7] = True
    hc = sess.import_host(pa.array(a, mask=mask))
    got = sess.call_function("add", [hc, hb])
    assert got.null_count == 1 and got[n - 8].as_py() == 1


@pytest.mark.parametrize("name", ["equal", "not_equal", "greater", "greater_equal", "less", "less_equal"])
def test_streamed_comparisons(sess, name):
    rng = np.random.default_rng(len(name))
    n = 70_001
    for typ in (pa.int64(), pa.float64(), pa.int8()):
        a = column(rng, n, pa.int64(), True)
        b = column(rng, n, pa.int64(), True)
        if typ == pa.int8():
            a, b = pc.bit_wise_and(a, pa.scalar(7, pa.int64())), pc.bit_wise_and(b, pa.scalar(7, pa.int64()))
        a, b = a.cast(typ), b.cast(typ)
        pcf = getattr(pc, name)
        ha, hb = sess.import_host(a), sess.import_host(b)
        got, whole = both(sess, name, [ha, hb], [a, b])
        same(got, pcf(a, b))
        same_bytes(got, whole)
        sc = pa.scalar(2, typ)
        got, whole = both(sess, name, [ha, sc], [a, sc])
        same(got, pcf(a, sc))
        same_bytes(got, whole)
        got, whole = both(sess, name, [sc, hb], [sc, b])
        same(got, pcf(sc, b))
        same_bytes(got, whole)


@pytest.mark.parametrize("null_sel", ["drop", "emit_null"])
def test_streamed_filter(sess, null_sel):
    rng = np.random.default_rng(5)
    n = 200_017
    for typ in (pa.int64(), pa.float32(), pa.int16()):
        for vnulls in (False, True):
            for fnulls in (False, True):
                v = column(rng, n, pa.int64(), vnulls).cast(pa.float32() if typ == pa.float32() else typ, safe=False)
                f = pa.array(rng.random(n) < 0.4, mask=(rng.random(n) < 0.1) if fnulls else None)
                want = pc.filter(v, f, null_selection_behavior=null_sel)
                got, whole = both(sess, "filter", [sess.import_host(v), sess.import_host(f)], [v, f], options=f"null_selection_behavior={null_sel}")
                same(got, want)
                same_bytes(got, whole)


def test_streamed_math_sum(sess):
    rng = np.random.default_rng(9)
    n = 300_001
    x = rng.uniform(-1, 1, n)
    h = sess.import_host(pa.array(x))
    assert abs(sess.math_sum(h) - float(np.sum(x, dtype=np.longdouble))) <= 1e-9
    i = rng.integers(-2**40, 2**40, n)
    assert sess.math_sum(sess.import_host(pa.array(i))) == int(i.sum())
    short = pa.array([1e308, 1e308, -1e308])                 # ≤ 31 rows: the reference's sequential order, also through the ingest
    assert sess.math_sum(sess.import_host(short)) == float("inf")


_TNAME = {pa.int8(): "int8", pa.uint8(): "uint8", pa.int16(): "int16", pa.uint16(): "uint16", pa.int32(): "int32", pa.uint32(): "uint32",
          pa.int64(): "int64", pa.uint64(): "uint64", pa.float32(): "float", pa.float64(): "double"}


@pytest.mark.parametrize("pair", [(pa.int64(), pa.int32()), (pa.int32(), pa.int64()), (pa.float64(), pa.float32()), (pa.int64(), pa.float64()),
                                  (pa.float64(), pa.int64()), (pa.uint16(), pa.int8()), (pa.float32(), pa.uint8())], ids=str)
def test_streamed_cast(sess, pair):
    """numeric → numeric casts of a host-resident column run span by span (CastIntToInt … CastFloatingToFloating,
    kernels/numeric_cast.go:37-71): the whole-array path's bytes — payload under nulls, validity, tail bits —, safe and unsafe; a value
    that does not fit fails the safe cast with the whole-array path's error, wherever its span lies; under a null it is not looked at"""
    from arrow_go_amd import compute as ac
    src, dst = pair
    rng = np.random.default_rng(zlib_seed("cast", src, dst))
    n = 100_003
    for nulls in (False, True):
        a = column(rng, n, pa.int64(), nulls)
        a = pc.bit_wise_and(a, pa.scalar(63, pa.int64())).cast(src)              # 0 … 63: fits every target, integral
        opt = f"to_type={_TNAME[dst]}"
        got, whole = both(sess, "cast", [sess.import_host(a)], [a], options=opt)
        same(got, a.cast(dst))
        same_bytes(got, whole)
        got, whole = both(sess, "cast", [sess.import_host(a)], [a], options=opt + ";safe=0")
        same_bytes(got, whole)
    # a value the target cannot hold (or would truncate), in the last span
    bad = {(pa.int64(), pa.int32()): 2**40, (pa.int64(), pa.float64()): 2**53 + 1, (pa.float64(), pa.int64()): 0.5, (pa.uint16(), pa.int8()): 300,
           (pa.float32(), pa.uint8()): 300.5}.get(pair)
    if bad is None:          # widening and float → float casts cannot fail
        return
    vals = np.zeros(n, src.to_pandas_dtype())
    vals[n - 11] = bad
    errs = []
    for arg in (sess.import_host(pa.array(vals)), pa.array(vals)):
        with pytest.raises(ac.ArrowError) as ei:
            sess.call_function("cast", [arg], options=f"to_type={_TNAME[dst]}")
        errs.append(str(ei.value))
    assert errs[0] == errs[1], errs
    mask = np.zeros(n, bool)
    mask[n - 11] = True
    m = pa.array(vals, mask=mask)
    got, whole = both(sess, "cast", [sess.import_host(m)], [m], options=f"to_type={_TNAME[dst]}")
    same_bytes(got, whole)


@pytest.mark.parametrize("typ", [pa.int64(), pa.uint64(), pa.int32(), pa.uint16(), pa.int8()], ids=str)
def test_streamed_cumulative_sum(sess, typ):
    """cumulative_sum / cumulative_sum_checked of a host-resident INTEGER column: the running sum travels from span to span (13 spans
    here), the bytes are the whole-array path's (integer sums wrap: the cut does not matter) — with a start value, with nulls skipped
    (payload 0 under them, the validity's tail bits ones: prepareCumulativeOutput), with the last rows of a span null; a checked sum
    that leaves the range in a LATE span fails with "overflow"; float columns and nulls that are not skipped take the whole-array path"""
    from arrow_go_amd import compute as ac
    rng = np.random.default_rng(zlib_seed("cumsum", typ))
    n = 100_003
    info = np.iinfo(typ.to_pandas_dtype())
    small = max(int(info.max) // (4 * n), 1)
    tn = _TNAME[typ]
    for nulls in (False, True):
        if info.max < 2**31:      # narrow types: a handful of ones, so that no running sum leaves the range
            v = (rng.random(n) < 20.0 / n).astype(np.int64)
        else:
            v = rng.integers(0 if info.min == 0 else -small, small, n, endpoint=True)
        mask = None
        if nulls:
            mask = rng.random(n) < 0.1
            mask[8192 - 300:8192] = True                     # the first span (64 KiB of Int64) ends in nulls: its running sum is an earlier row's
        a = pa.array(v, type=typ, mask=mask)
        for name in ("cumulative_sum", "cumulative_sum_checked"):
            for opt in ("skip_nulls=1", f"skip_nulls=1;start={tn}:3"):
                got, whole = both(sess, name, [sess.import_host(a)], [a], options=opt)
                same_bytes(got, whole)
                same(got, (pc.cumulative_sum_checked if name.endswith("checked") else pc.cumulative_sum)(a, start=3 if "start" in opt else 0, skip_nulls=True))
        if nulls:   # nulls not skipped: every row behind the first null is null — the whole-array path (the column is uploaded)
            r = sess.call_function("cumulative_sum", [sess.import_host(a)], options="skip_nulls=0", keep_on_device=True)
            assert not r.on_host()
            same(r.to_arrow(), pc.cumulative_sum(a, skip_nulls=False))
    # wrap-around (unchecked) and overflow (checked) in a late span
    wide = pa.array(rng.integers(info.min, info.max, n, endpoint=True, dtype=typ.to_pandas_dtype()), type=typ)
    got, whole = both(sess, "cumulative_sum", [sess.import_host(wide)], [wide], options="skip_nulls=1")
    same_bytes(got, whole)
    late = np.zeros(n, typ.to_pandas_dtype())
    late[5] = info.max
    late[n - 9] = 1
    for arg in (sess.import_host(pa.array(late)), pa.array(late)):
        with pytest.raises(ac.ArrowError, match="overflow"):
            sess.call_function("cumulative_sum_checked", [arg], options="skip_nulls=1")
    # floats: not streamed
    f = pa.array(rng.uniform(-1, 1, n))
    r = sess.call_function("cumulative_sum", [sess.import_host(f)], keep_on_device=True)
    assert not r.on_host()


def test_other_functions_upload_the_column_whole(sess):
    rng = np.random.default_rng(10)
    a = column(rng, 60_000, pa.int64(), True)
    h = sess.import_host(a)
    assert h.on_host()
    same(sess.call_function("negate", [h]), pc.negate_checked(a))      # not streamed: uploaded whole (inside compute::CallFunction) …
    assert not h.on_host()                                               # … and device-resident from then on (ArrayData::device_twin)
    r = sess.call_function("add", [h, h], keep_on_device=True)
    assert not r.on_host()                                               # the copy in HBM is used, nothing crosses the link again
    same(r.to_arrow(), pc.add_checked(a, a))
    assert sess.math_sum(h) == sess.math_sum(a)                        # arrow/math over the uploaded copy
    # the expression executor takes host columns too (fused and node by node)
    for fuse in (True, False):
        h3 = sess.import_host(a)
        got, _ = sess.eval_expression("add($0,$0)", [h3], fuse=fuse)
        same(got, pc.add_checked(a, a))
    # mixed residency: the host side is uploaded
    h2 = sess.import_host(a)
    dev = sess.call_function("negate", [a], keep_on_device=True)
    same(sess.call_function("add", [h2, dev]), pc.add_checked(a, pc.negate_checked(a)))
    # below the threshold nothing stays on the host
    sess.set_option("host_threshold_bytes", 64 << 20)
    try:
        assert not sess.import_host(a).on_host()
    finally:
        sess.set_option("host_threshold_bytes", 0)


def _pinned_column(ctx, sess, rows, dtype, fill):
    pb = ctx.alloc_pinned(rows * 8 + 64)
    v = pb.view(dtype, rows)
    chunk = fill(1 << 22)
    for o in range(0, rows, chunk.size):
        v[o:o + chunk.size] = chunk[:min(chunk.size, rows - o)]
    return pb, v


def test_one_gib_pinned_column_runs_at_the_links_rate(ctx):
    """ahc_call("add") on 1 GiB pinned columns: 2 GiB cross the link towards the device while 1 GiB comes back; the uploads must run at
    ≥ 0.90 of what one plain pinned copy of the same size reaches, and the bytes must be the whole-array path's"""
    import time
    import arrow_go_amd as ah
    from arrow_go_amd import compute as ac
    rows = 1 << 27
    rng = np.random.default_rng(3)
    s = ac.Session(0)
    try:
        pa_buf, va = _pinned_column(ctx, s, rows, np.int64, lambda m: rng.integers(-2**40, 2**40, m))
        pb_buf, vb = _pinned_column(ctx, s, rows, np.int64, lambda m: rng.integers(-2**40, 2**40, m))
        # the link: one plain pinned upload of 1 GiB
        dev = ctx.alloc(rows * 8)
        best = 1e9
        for _ in range(3):
            ctx.sync()
            t0 = time.perf_counter()
            ah._native.check(ctx.handle, ah._native.lib.ah_upload_async(ctx.handle, dev.ptr, pa_buf.ptr, rows * 8))
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        link = rows * 8 / best / 1e9
        dev.free()
        ha = s.import_host_buffers("int64", rows, pa_buf.ptr)
        hb = s.import_host_buffers("int64", rows, pb_buf.ptr)
        assert ha.on_host() and hb.on_host()
        rates = {}
        for name in ("add", "add_unchecked"):
            s.call_function(name, [ha, hb], keep_on_device=True).release()      # the first call pins the output block and creates the ingest
            t_best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                out = s.call_function(name, [ha, hb], keep_on_device=True)
                t_best = min(t_best, time.perf_counter() - t0)
                last = out
                if _ < 2:
                    out.release()
            rates[name] = 2 * rows * 8 / t_best / 1e9
            _, dptr = last.buffers()
            import ctypes
            got = np.ctypeslib.as_array((ctypes.c_int64 * rows).from_address(dptr))
            assert last.on_host() and np.array_equal(got, va + vb), name
            last.release()
        print(f"link {link:.1f} GB/s; add {rates['add']:.1f}, add_unchecked {rates['add_unchecked']:.1f} GB/s towards the device")
        assert rates["add_unchecked"] >= 0.90 * link, (rates, link)
        assert rates["add"] >= 0.85 * link, (rates, link)          # the checked kernel returns a verdict per span
        ha.release(); hb.release()
    finally:
        s.close()
        pa_buf.free(); pb_buf.free()


def test_columns_larger_than_the_free_hbm(ctx):
    """the whole-array path needs both arguments and the output resident; the streamed path needs three spans.  Most of the HBM is
    taken first (never touched), then two 2 GiB host columns are added: the upload-everything path cannot allocate, the streamed
    path runs."""
    import arrow_go_amd as ah
    from arrow_go_amd import compute as ac
    import ctypes as C
    free_b, total_b = C.c_size_t(), C.c_size_t()
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b)) == 0
    rows = 1 << 28                                   # 2 GiB per Int64 column
    leave = 3 << 30                                  # < 3 × 2 GiB: the whole-array path cannot fit
    hogs = []
    s = ac.Session(0)
    try:
        want = int(free_b.value) - leave
        while want > (1 << 30):                      # in 16 GiB pieces; allocated, never touched
            sz = min(want, 16 << 30)
            p = C.c_void_p()
            if hip.hipMalloc(C.byref(p), C.c_size_t(sz)) != 0:
                break
            hogs.append(p)
            want -= sz
        assert hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b)) == 0 and free_b.value < 3 * rows * 8
        rng = np.random.default_rng(4)
        pa_buf, va = _pinned_column(ctx, s, rows, np.int64, lambda m: rng.integers(-2**40, 2**40, m))
        pb_buf, vb = _pinned_column(ctx, s, rows, np.int64, lambda m: rng.integers(-2**40, 2**40, m))
        ha = s.import_host_buffers("int64", rows, pa_buf.ptr)
        hb = s.import_host_buffers("int64", rows, pb_buf.ptr)
        out = s.call_function("add", [ha, hb], keep_on_device=True)
        _, dptr = out.buffers()
        got = np.ctypeslib.as_array((C.c_int64 * rows).from_address(dptr))
        assert out.on_host() and np.array_equal(got, va + vb)
        assert s.math_sum(ha) == int(va.sum())
        out.release(); ha.release(); hb.release()
        pa_buf.free(); pb_buf.free()
    finally:
        for p in hogs:
            hip.hipFree(p)
        s.close()
