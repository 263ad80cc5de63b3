"""Arrow IPC stream → HBM: the host-side FlatBuffer / framing reader is checked
against streams written by pyarrow without a GPU; the upload path on the GPU box.
Reference behaviour: arrow/ipc/reader.go:97-300, message.go:207-287, file_reader.go:523-616."""
import struct

import numpy as np
import pyarrow as pa
import pytest

from arrow_go_amd import compute as ac

TYPES = [pa.int8(), pa.uint8(), pa.int16(), pa.uint16(), pa.int32(), pa.uint32(), pa.int64(), pa.uint64(), pa.float32(), pa.float64(),
         pa.bool_(), pa.string(), pa.binary(), pa.large_string(), pa.large_binary()]
NAMES = {"float": "float", "double": "double"}


def random_column(rng, typ, n, p_null):
    mask = rng.random(n) < p_null if p_null else None
    if pa.types.is_boolean(typ):
        return pa.array(rng.random(n) < 0.5, mask=mask, type=typ)
    if pa.types.is_floating(typ):
        return pa.array(rng.normal(size=n), mask=mask, type=typ)
    if pa.types.is_integer(typ):
        info = np.iinfo(typ.to_pandas_dtype())
        return pa.array(rng.integers(info.min, info.max, n, dtype=typ.to_pandas_dtype(), endpoint=True), mask=mask, type=typ)
    words = [("w%d" % i) * (i % 7) for i in range(50)]
    vals = [words[j] for j in rng.integers(0, len(words), n)]
    if pa.types.is_binary(typ) or pa.types.is_large_binary(typ):
        vals = [v.encode() for v in vals]
    return pa.array(vals, mask=mask, type=typ)


def make_stream(batches, schema):
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, schema) as w:
        for b in batches:
            w.write_batch(b)
    return sink.getvalue()


def sample_stream(seed=0, sizes=(1000, 0, 7, 4097)):
    rng = np.random.default_rng(seed)
    schema = pa.schema([pa.field("c%d_%s" % (i, t), t, nullable=(i % 3 != 0)) for i, t in enumerate(TYPES)])
    batches = [pa.record_batch([random_column(rng, t, n, 0.2 if i % 3 else 0.0) for i, t in enumerate(TYPES)], schema=schema) for n in sizes]
    return schema, batches, make_stream(batches, schema)


def test_inspect_schema_and_batches():
    schema, batches, buf = sample_stream()
    fields, rows = ac.ipc_inspect(buf)
    assert [f[0] for f in fields] == schema.names
    assert [f[1] for f in fields] == [{"float": "float", "double": "double"}.get(str(t), str(t)) for t in schema.types]
    assert [f[2] for f in fields] == [f.nullable for f in schema]
    assert rows == [b.num_rows for b in batches]
    # plain bytes and a schema-only stream
    assert ac.ipc_inspect(buf.to_pybytes())[1] == rows
    assert ac.ipc_inspect(make_stream([], schema)) == (fields, [])


def test_inspect_file_format():
    # ipc.NewFileReader's input: magic, the same messages, EOS, footer, magic
    schema, batches, _ = sample_stream(5, sizes=(10, 0, 300))
    sink = pa.BufferOutputStream()
    with pa.ipc.new_file(sink, schema) as w:
        for b in batches:
            w.write_batch(b)
    fields, rows = ac.ipc_inspect(sink.getvalue())
    assert [f[0] for f in fields] == schema.names and rows == [10, 0, 300]


def test_inspect_dictionary_batches():
    # DictionaryBatch messages (reader.go:167-200): the field reports as a dictionary, the batches keep their lengths;
    # a replaced dictionary (same id, second non-delta batch) is accepted like the reference does
    d1 = pa.DictionaryArray.from_arrays(pa.array([0, 1, None, 0], pa.int8()), pa.array(["x", "yy"]))
    d2 = pa.DictionaryArray.from_arrays(pa.array([1, 0], pa.int8()), pa.array(["p", "q", "r"]))
    schema = pa.schema([("k", d1.type), ("v", pa.int32())])
    buf = make_stream([pa.record_batch([d1, pa.array([1, 2, 3, 4], pa.int32())], schema=schema),
                       pa.record_batch([d2, pa.array([5, 6], pa.int32())], schema=schema)], schema)
    fields, rows = ac.ipc_inspect(buf)
    assert fields == [("k", "dictionary", True), ("v", "int32", True)] and rows == [4, 2]


def test_inspect_rejects_what_it_does_not_read():
    l = pa.array([[1, 2], [3]])
    with pytest.raises(ac.ErrNotImplemented, match="nested|flatbuf type"):
        ac.ipc_inspect(make_stream([pa.record_batch([l], names=["l"])], pa.schema([("l", l.type)])))
    t = pa.array([1, 2], type=pa.decimal128(10, 2))
    with pytest.raises(ac.ErrNotImplemented, match="flatbuf type"):
        ac.ipc_inspect(make_stream([pa.record_batch([t], names=["t"])], pa.schema([("t", t.type)])))


def compressed_stream(codec, seed=5, sizes=(1000, 0, 7, 40000), file_format=False):
    rng = np.random.default_rng(seed)
    schema = pa.schema([pa.field("c%d_%s" % (i, t), t, nullable=(i % 3 != 0)) for i, t in enumerate(TYPES)])
    batches = [pa.record_batch([random_column(rng, t, n, 0.2 if i % 3 else 0.0) for i, t in enumerate(TYPES)], schema=schema) for n in sizes]
    sink = pa.BufferOutputStream()
    opts = pa.ipc.IpcWriteOptions(compression=codec)
    with (pa.ipc.new_file if file_format else pa.ipc.new_stream)(sink, schema, options=opts) as w:
        for b in batches:
            w.write_batch(b)
    return schema, batches, sink.getvalue()


@pytest.mark.parametrize("codec", ["lz4", "zstd"])
def test_inspect_compressed_bodies(codec):
    """BodyCompression (ipc/compression.go:25-40, file_reader.go:585-612): every buffer is [int64 length | frame], −1 = stored.
    The walk inflates each buffer and checks the announced sizes against the schema like an uncompressed body."""
    schema, batches, buf = compressed_stream(codec)
    fields, rows = ac.ipc_inspect(buf)
    assert [f[0] for f in fields] == schema.names and rows == [b.num_rows for b in batches]
    # a small incompressible column travels as stored buffers (length prefix −1) next to compressed ones
    rng = np.random.default_rng(0)
    t = pa.table({"r": rng.integers(0, 2**62, 50), "z": np.zeros(50, np.int64)})
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, t.schema, options=pa.ipc.IpcWriteOptions(compression=codec)) as w:
        w.write_table(t)
    assert ac.ipc_inspect(sink.getvalue())[1] == [50]


@pytest.mark.parametrize("codec", ["lz4", "zstd"])
def test_inspect_damaged_compressed_bodies(codec):
    """flipped bytes inside the frames and lying length prefixes end in an error, not in a crash or an oversized allocation"""
    _, _, buf = compressed_stream(codec, sizes=(3000,))
    raw = bytearray(buf.to_pybytes())
    rng = np.random.default_rng(11)
    body_from = len(raw) // 3
    errors = 0
    for _ in range(150):
        b = bytearray(raw)
        pos = int(rng.integers(body_from, len(b) - 16))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            ac.ipc_inspect(bytes(b))
        except (ac.ErrInvalid, ac.ErrNotImplemented):
            errors += 1
    assert errors > 0                     # LZ4 / ZSTD frames carry checksums or structure: most flips are noticed
    # a length prefix that announces 2^50 bytes from a few hundred
    b = bytearray(raw)
    idx = bytes(raw).find(struct.pack("<q", 3000 * 8))   # the int64 column's announced size
    assert idx > 0
    b[idx:idx + 8] = struct.pack("<q", 1 << 50)
    with pytest.raises(ac.ErrInvalid):
        ac.ipc_inspect(bytes(b))


def test_inspect_survives_damaged_streams():
    """truncations and bit flips anywhere in the metadata end in an error (or a shorter stream), never a crash"""
    _, _, buf = sample_stream(1, sizes=(100, 33))
    raw = buf.to_pybytes()
    with pytest.raises(ac.ErrInvalid):
        ac.ipc_inspect(b"")
    with pytest.raises(ac.ErrInvalid):
        ac.ipc_inspect(raw[:3])
    ok = bad = 0
    for cut in list(range(4, 600, 7)) + list(range(600, len(raw), 997)):
        try:
            ac.ipc_inspect(raw[:cut]); ok += 1
        except ac.ArrowError:
            bad += 1
    assert bad > 0
    rng = np.random.default_rng(2)
    meta_end = 8 + struct.unpack("<i", raw[4:8])[0]
    for _ in range(400):
        b = bytearray(raw)
        pos = int(rng.integers(0, min(len(raw), meta_end + 2000)))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            ac.ipc_inspect(bytes(b))
        except ac.ArrowError:
            pass
    # a body shorter than the metadata says
    with pytest.raises(ac.ErrInvalid, match="body"):
        ac.ipc_inspect(raw[:-64])


@pytest.fixture(scope="module")
def sess():
    s = ac.Session(0)
    yield s
    s.close()


@pytest.mark.gpu
def test_read_ipc_round_trip(sess):
    schema, batches, buf = sample_stream(3)
    got = list(sess.read_ipc(buf))
    assert len(got) == len(batches)
    for (names, cols, rows), b in zip(got, batches):
        assert names == schema.names and rows == b.num_rows
        for c, exp in zip(cols, b.columns):
            assert c.to_arrow().equals(exp)
    # the columns are ordinary device arrays: compute on them without leaving HBM
    names, cols, rows = got[0]
    i64 = cols[names.index("c6_int64")]
    out = sess.call_function("add_unchecked", [i64, i64])
    import pyarrow.compute as pc
    assert out.equals(pc.add(batches[0].column(6), batches[0].column(6)))
    s = cols[names.index("c11_string")]
    assert sess.call_function("unique", [s]).equals(pc.unique(batches[0].column(11)))


@pytest.mark.gpu
@pytest.mark.parametrize("codec,file_format", [("lz4", False), ("zstd", False), ("zstd", True)])
def test_read_ipc_compressed(sess, codec, file_format):
    schema, batches, buf = compressed_stream(codec, seed=9, file_format=file_format)
    got = list(sess.read_ipc(buf))
    assert len(got) == len(batches)
    for (names, cols, rows), b in zip(got, batches):
        assert names == schema.names and rows == b.num_rows
        for c, exp in zip(cols, b.columns):
            assert c.to_arrow().equals(exp)


@pytest.mark.gpu
def test_read_ipc_dictionary_columns(sess):
    rng = np.random.default_rng(8)
    n = 5000
    words = pa.array(["w%d" % i for i in range(37)] + [None])
    nums = pa.array(np.arange(100, 120), pa.int64())
    for index_type in (pa.int8(), pa.uint16(), pa.int32(), pa.int64()):
        k1 = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 38, n), mask=rng.random(n) < 0.1, type=index_type), words)
        k2 = pa.DictionaryArray.from_arrays(pa.array(rng.integers(0, 20, n), type=index_type), nums)
        schema = pa.schema([("s", k1.type), ("x", pa.float64()), ("i", k2.type)])
        b = pa.record_batch([k1, pa.array(rng.standard_normal(n)), k2], schema=schema)
        got = list(sess.read_ipc(make_stream([b, b.slice(10, 100)], schema)))
        assert [g[2] for g in got] == [n, 100]
        for (names, cols, rows), exp in zip(got, (b, b.slice(10, 100))):
            for c, e in zip(cols, exp.columns):
                a = c.to_arrow()
                assert a.type == e.type and a.equals(e), (index_type, a.type)
        # the columns are ordinary dictionary arrays on the device
        s_col = got[0][1][0]
        sel = pa.array(rng.integers(0, n, 777), type=pa.int32())
        assert sess.call_function("take", [s_col, sel]).equals(k1.take(sel))
        import pyarrow.compute as pc
        assert sess.call_function("unique", [s_col]).indices.equals(pc.unique(k1).indices)


@pytest.mark.gpu
def test_read_ipc_delta_dictionaries(sess):
    # reader.go:186-196: a delta batch appends to the dictionary the earlier batches were read with
    d1, d2, d3 = (pa.array(v).dictionary_encode() for v in (["a", "b", "a"], ["a", "b", "c", "c"], ["a", "b", "c", "d", None]))
    sink = pa.BufferOutputStream()
    with pa.ipc.new_stream(sink, pa.schema([("d", d1.type)]), options=pa.ipc.IpcWriteOptions(emit_dictionary_deltas=True)) as w:
        for d in (d1, d2, d3):
            w.write_batch(pa.record_batch([d], names=["d"]))
    assert ac.ipc_inspect(sink.getvalue())[1] == [3, 4, 5]
    got = [cols[0].to_arrow() for _, cols, _ in sess.read_ipc(sink.getvalue())]
    assert [g.to_pylist() for g in got] == [["a", "b", "a"], ["a", "b", "c", "c"], ["a", "b", "c", "d", None]]
    assert got[1].dictionary.to_pylist() == ["a", "b", "c"] and got[2].dictionary.to_pylist() == ["a", "b", "c", "d"]


@pytest.mark.gpu
def test_read_ipc_file_format(sess):
    schema, batches, _ = sample_stream(6, sizes=(500, 64))
    sink = pa.BufferOutputStream()
    with pa.ipc.new_file(sink, schema) as w:
        for b in batches:
            w.write_batch(b)
    got = list(sess.read_ipc(sink.getvalue()))
    assert len(got) == 2
    for (names, cols, rows), b in zip(got, batches):
        assert rows == b.num_rows and all(c.to_arrow().equals(e) for c, e in zip(cols, b.columns))


@pytest.mark.gpu
def test_read_ipc_large_batch(sess):
    rng = np.random.default_rng(4)
    n = 1 << 22
    schema = pa.schema([("k", pa.int64()), ("v", pa.float64())])
    b = pa.record_batch([pa.array(rng.integers(0, 1000, n)), pa.array(rng.normal(size=n))], schema=schema)
    buf = make_stream([b, b.slice(5, 1000)], schema)
    got = list(sess.read_ipc(buf))
    assert [g[2] for g in got] == [n, 1000]
    assert sess.math_sum(got[0][1][0]) == int(np.asarray(b.column(0)).sum())
    assert got[1][1][1].to_arrow().equals(b.column(1).slice(5, 1000))


def test_inspect_rejects_row_counts_that_would_wrap_the_size_checks():
    """a crafted RecordBatch.length / FieldNode.length ≥ 2^60 would wrap `rows × width` to a small number and pass the
    "buffer holds enough bytes" tests; the reader bounds the row count by the body size first (Go would panic on the slice)"""
    schema = pa.schema([("x", pa.int64())])
    n = 12345
    raw = bytearray(make_stream([pa.record_batch([pa.array(np.arange(n))], schema=schema)], schema).to_pybytes())
    pat = struct.pack("<q", n)
    hits = [i for i in range(len(raw) - 8) if raw[i:i + 8] == pat and i < len(raw) - 8 * n]   # metadata only, not the body
    assert len(hits) >= 2                       # RecordBatch.length and the FieldNode's length
    for big in (1 << 61, (1 << 61) + 1, 1 << 62):
        b = bytearray(raw)
        for i in hits:
            b[i:i + 8] = struct.pack("<q", big)
        with pytest.raises(ac.ErrInvalid, match="cannot fit|rows"):
            ac.ipc_inspect(bytes(b))
