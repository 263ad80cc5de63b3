"""Arrow C Device Data Interface (arrow/cdata/abi.h:66-128): device arrays enter and
leave the host layer WITHOUT a copy.  The producer here is a second ah_ctx on the same GPU (its own
stream), standing in for "another ROCm library"."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    from arrow_go_amd import compute as ac
    s = ac.Session(0)
    yield s
    s.close()


@pytest.fixture(scope="module")
def producer():
    import arrow_go_amd as ah
    c = ah.Context(0)
    yield c
    c.close()


def test_import_compute_export_zero_copy(sess, producer):
    from arrow_go_amd import compute as ac
    n = 100003
    rng = np.random.default_rng(1)
    x = rng.integers(-1000, 1000, n, dtype=np.int64)
    valid = rng.random(n) >= 0.2
    vb = producer.to_device(x)
    bb = producer.to_device(np.packbits(valid, bitorder="little"))
    producer.sync()
    released = []
    a = sess.import_device("int64", n, vb.ptr, bb.ptr, null_count=int((~valid).sum()), on_release=lambda: released.append(1))
    assert a.buffers() == (bb.ptr, vb.ptr)                       # the producer's memory, not a copy
    exp = pa.array(x, mask=~valid)
    assert a.to_arrow().equals(exp)
    # chain two calls without leaving the device, then hand the result out through the device interface
    t = sess.call_function("add", [a, pa.scalar(5, pa.int64())], keep_on_device=True)
    m = sess.call_function("greater", [t, pa.scalar(0, pa.int64())], keep_on_device=True)
    kept = sess.call_function("filter", [t, m], keep_on_device=True)
    darr, sch = kept.export_device()
    assert darr.device_type == ac.ARROW_DEVICE_ROCM and darr.device_id == 0 and not darr.sync_event
    assert sch.format == b"l" and darr.array.n_buffers == 2
    kv, kd = kept.buffers()
    assert darr.array.buffers[1] == kd and darr.array.buffers[0] == kv      # again no copy
    want = pc.filter(pc.add(exp, 5), pc.greater(pc.add(exp, 5), 0))
    assert darr.array.length == len(want) and darr.array.null_count == 0
    # the consumer (here: the producer context) reads the exported pointer directly
    import arrow_go_amd as ah
    host = np.zeros(len(want), np.int64)
    ah._native.check(producer.handle, ah._native.lib.ah_download_async(producer.handle, host.ctypes.data, darr.array.buffers[1], host.nbytes))
    producer.sync()
    assert host.tolist() == want.to_pylist()
    # lifetimes: the export keeps the buffers alive after the datum handle is gone; the import's release
    # callback fires exactly once, when the last reference to the producer's array is dropped
    kept.release()
    ah._native.check(producer.handle, ah._native.lib.ah_download_async(producer.handle, host.ctypes.data, darr.array.buffers[1], host.nbytes))
    producer.sync()
    assert host.tolist() == want.to_pylist()
    darr.array.release(C.byref(darr.array))
    assert not darr.array.release
    assert released == []
    same = sess.call_function("cast", [a], "to_type=int64", keep_on_device=True)   # identity cast: a view of the import
    assert same.buffers() == a.buffers()
    a.release()
    assert released == []            # still referenced by `same`
    same.release()
    assert released == [1]
    for d in (t, m):
        d.release()


def test_import_sync_event_slices_and_errors(sess, producer):
    from arrow_go_amd import compute as ac
    hip = C.CDLL("libamdhip64.so")
    ev = C.c_void_p()
    assert hip.hipEventCreate(C.byref(ev)) == 0
    x = np.arange(1000, dtype=np.float64)
    vb = producer.to_device(x)          # upload enqueued on the producer's stream …
    assert hip.hipEventRecord(ev, None) == 0
    a = sess.import_device("double", 900, vb.ptr, offset=50, sync_event=C.addressof(ev))   # … the session's stream waits for the event
    got = sess.call_function("multiply", [a, pa.scalar(2.0)])
    assert got.to_pylist() == (x[50:950] * 2).tolist()
    a.release()
    hip.hipEventDestroy(ev)
    # wrong device / foreign device type: error, and the producer's array is released
    rel = []
    with pytest.raises(ac.ErrInvalid, match="lives on device 3"):
        sess.import_device("double", 10, vb.ptr, device_id=3, on_release=lambda: rel.append(1))
    with pytest.raises(ac.ErrNotImplemented, match="device type 2"):
        sess.import_device("double", 10, vb.ptr, device_type=2, on_release=lambda: rel.append(1))
    assert rel == [1, 1]


def test_import_cpu_device_array_from_pyarrow(sess):
    """ARROW_DEVICE_CPU arrays (what pyarrow exports) take the host path: uploaded, then computed on"""
    from arrow_go_amd import compute as ac
    arr = pa.array([1, None, 3, 4], pa.int32())
    darr, sch = ac.CArrowDeviceArray(), ac.CArrowSchema()
    arr._export_to_c_device(C.addressof(darr), C.addressof(sch))
    assert darr.device_type == ac.ARROW_DEVICE_CPU
    d = C.c_void_p()
    sess._check(ac.lib.ahc_import_device(sess.h, C.addressof(darr), C.addressof(sch), C.byref(d)))
    da = ac.DeviceArray(sess, d)
    assert sess.call_function("add", [da, da]).to_pylist() == [2, None, 6, 8]
    da.release()


def test_string_device_round_trip(sess, producer):
    """String / binary layouts are [validity, offsets, data] (three buffers) on both sides of the device interface:
    import wraps all three without a copy (the data extent comes from the last offset), export hands all three out."""
    from arrow_go_amd import compute as ac
    import arrow_go_amd as ah
    rng = np.random.default_rng(3)
    words = [("w%d" % i) * (i % 5) for i in range(40)]
    for typ, name, odt in ((pa.string(), "string", np.int32), (pa.large_binary(), "large_binary", np.int64)):
        vals = [words[j] for j in rng.integers(0, len(words), 1000)]
        if name == "large_binary":
            vals = [v.encode() for v in vals]
        mask = rng.random(1000) < 0.2
        arr = pa.array(vals, mask=mask, type=typ)
        vbuf, obuf, dbuf = arr.buffers()
        ob = producer.to_device(np.frombuffer(obuf, odt, 1001))
        db = producer.to_device(np.frombuffer(dbuf, np.uint8))
        vb = producer.to_device(np.frombuffer(vbuf, np.uint8))
        producer.sync()
        rel = []
        a = sess.import_device(name, 900, ob.ptr, vb.ptr, null_count=-1, offset=50, var_data_ptr=db.ptr, on_release=lambda: rel.append(1))
        assert a.to_arrow().equals(arr.slice(50, 900))
        idx = pa.array(rng.integers(0, 900, 300), pa.int32())
        assert sess.call_function("take", [a, idx]).equals(pc.take(arr.slice(50, 900), idx))
        darr, sch = a.export_device()
        assert darr.array.n_buffers == 3 and sch.format == {"string": b"u", "large_binary": b"Z"}[name]
        assert (darr.array.buffers[1], darr.array.buffers[2]) == (ob.ptr, db.ptr)    # zero copy both ways
        assert darr.array.offset == 50 and darr.array.length == 900
        darr.array.release(C.byref(darr.array))
        a.release()
        assert rel == [1]
        # a fixed-width array with 3 buffers / a string array with 2: rejected, the producer's array released
        with pytest.raises(ac.ErrInvalid, match="needs 3 buffers"):
            sess.import_device(name, 10, ob.ptr, n_buffers=2, on_release=lambda: rel.append(2))
        with pytest.raises(ac.ErrInvalid, match="needs 2 buffers"):
            sess.import_device("int64", 10, ob.ptr, n_buffers=3, on_release=lambda: rel.append(3))
        assert rel == [1, 2, 3]


def test_datum_outlives_session(producer):
    """a datum (and the pooled blocks behind it) released after its session was destroyed must not touch freed memory:
    buffers keep the session's context and pool alive (Buffer::keep)"""
    from arrow_go_amd import compute as ac
    with ac.Session(0) as s:
        d = s.call_function("add", [pa.array(np.arange(100000)), pa.scalar(1)], keep_on_device=True)
        kept_ptr = d.buffers()[1]
    host = np.zeros(100000, np.int64)
    import arrow_go_amd as ah
    ah._native.check(producer.handle, ah._native.lib.ah_download_async(producer.handle, host.ctypes.data, kept_ptr, host.nbytes))
    producer.sync()
    assert host[-1] == 100000                     # the memory is still the datum's
    d.release()                                   # last reference: block goes back to the (still alive) pool, then the pool is freed
